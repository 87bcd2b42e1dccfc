// Monthly averaging and error propagation: averaging.py:11-24 (error_averager) and :97-108 (the
// np.nanmean(axis=0) reductions) of the reference.
//
// HBM layout: a stack of k regridded granules, k*n contiguous elements of T (granule-major), read
// exactly once; one n-element field out.  Thread per cell (x VEC cells), k-loop innermost: for
// every k the wave reads 64*VEC consecutive cells -> fully coalesced 16-byte-per-lane loads.
// The sum runs over k in order, exactly the order NumPy's axis-0 reduction uses, so float64
// results are bit-identical to np.nanmean for finite data.
#include "oisat_common.h"

namespace {

template <typename T>
struct Vec;
template <>
struct Vec<float> {
    using type = float4;
    static constexpr int N = 4;
};
template <>
struct Vec<double> {
    using type = double2;
    static constexpr int N = 2;
};

template <typename T>
__device__ __forceinline__ bool is_inf(T v) { return v == __builtin_inf() || v == -__builtin_inf(); }

template <typename T, bool ERR>
__device__ __forceinline__ void accum(T v, bool flagA, T& sum, unsigned& cnt) {
    // ERR:  flagA = square_input;  drop NaN and inf (averaging.py:19-20)
    // mean: flagA = inf_to_nan;    drop NaN, and inf too when flagged (averaging.py:92)
    if (ERR) {
        if (flagA) v = v * v;
        if (v == v && !is_inf(v)) { sum += v; ++cnt; }
    } else {
        if (v == v && !(flagA && is_inf(v))) { sum += v; ++cnt; }
    }
}

template <typename T, bool ERR>
__device__ __forceinline__ T finish(T sum, unsigned cnt) {
    if (ERR) {
        const T c = (T)cnt;
        return sqrt(sum / (c * c));        // sqrt(sum/size**2); 0/0 -> NaN
    }
    return sum / (T)cnt;                   // nanmean: 0/0 -> NaN
}

template <typename T, bool ERR>
__global__ __launch_bounds__(256) void stack_reduce_kernel(const T* __restrict__ stack, int k, int64_t n, bool flagA,
                                                            bool aligned, T* __restrict__ out) {
    using V = typename Vec<T>::type;
    constexpr int N = Vec<T>::N;
    const int64_t nvec = n / N;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    if (aligned) {
        for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += stride) {
            T sum[N];
            unsigned cnt[N];
#pragma unroll
            for (int j = 0; j < N; ++j) { sum[j] = T(0); cnt[j] = 0; }
            const V* p = reinterpret_cast<const V*>(stack) + v;
            for (int g = 0; g < k; ++g) {
                const V x = p[(int64_t)g * nvec];
                const T* xs = reinterpret_cast<const T*>(&x);
#pragma unroll
                for (int j = 0; j < N; ++j) accum<T, ERR>(xs[j], flagA, sum[j], cnt[j]);
            }
            V r;
            T* rs = reinterpret_cast<T*>(&r);
#pragma unroll
            for (int j = 0; j < N; ++j) rs[j] = finish<T, ERR>(sum[j], cnt[j]);
            reinterpret_cast<V*>(out)[v] = r;
        }
    } else {
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
            T sum = T(0);
            unsigned cnt = 0;
            for (int g = 0; g < k; ++g) accum<T, ERR>(stack[(int64_t)g * n + i], flagA, sum, cnt);
            out[i] = finish<T, ERR>(sum, cnt);
        }
    }
}

template <typename T, bool ERR>
int launch(oisat_ctx* h, const char* name, const void* stack, int k, int64_t n, int flag, void* out) {
    // vector path: every granule row (and the output) must start 16-byte aligned
    const bool aligned = (n % Vec<T>::N) == 0 && ((uintptr_t)stack % 16) == 0 && ((uintptr_t)out % 16) == 0;
    const int64_t items = aligned ? n / Vec<T>::N : n;
    const int grid = stream_grid(items, 256);
    OISAT_LAUNCH(h, name, (stack_reduce_kernel<T, ERR>), dim3(grid), dim3(256), 0, (const T*)stack, k, n, flag != 0,
                 aligned, (T*)out);
    return OISAT_OK;
}

}  // namespace

extern "C" int oisat_nanmean_stack(oisat_ctx* h, int dtype, const void* stack, int k, int64_t n, int inf_to_nan, void* out) {
    ARG_CHECK(h && stack && out && k > 0 && n > 0);
    ARG_CHECK(dtype == OISAT_F32 || dtype == OISAT_F64);
    if (dtype == OISAT_F32) return launch<float, false>(h, "nanmean_stack", stack, k, n, inf_to_nan, out);
    return launch<double, false>(h, "nanmean_stack", stack, k, n, inf_to_nan, out);
}

extern "C" int oisat_error_average(oisat_ctx* h, int dtype, const void* stack, int k, int64_t n, int square_input, void* out) {
    ARG_CHECK(h && stack && out && k > 0 && n > 0);
    ARG_CHECK(dtype == OISAT_F32 || dtype == OISAT_F64);
    if (dtype == OISAT_F32) return launch<float, true>(h, "error_average", stack, k, n, square_input, out);
    return launch<double, true>(h, "error_average", stack, k, n, square_input, out);
}

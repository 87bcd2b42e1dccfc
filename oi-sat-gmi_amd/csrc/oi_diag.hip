// Element-wise optimal interpolation: optimal_interpolation.py:6-52 of the reference.
//
//   sweep   (:26-33)  for each of the 99 scalings s: t = Sa*s, K = t*(t+So)^-1, Sb = (1-K)*t,
//                     AK = 1 - Sb/t, mean_s = nanmean(AK)      (the means: AK = K where t != 0, see ak_of)
//   apply   (:14,:46-52) Y[Y<0]=0; inc = K*(Y-Xa); Xb = Xa+inc; err = sqrt(Sb) for the chosen s
//
// HBM layout: four (ny*nx) fields of T in, four out, all contiguous; nothing else.
// The sweep never materialises the 99x3 temporaries the reference keeps alive: a wave loads 64
// cells coalesced, then walks them two at a time with lanes == scalings (three per lane and half-wave,
// see oi_curve_kernel), broadcasting the cells with v_readlane.  Every lane accumulates its own
// (sum, count) pairs in double; a second tiny kernel adds the per-block partials in block order.  The launch shape is fixed, so the 99 means are bitwise
// reproducible run to run -- the knee pick that follows is sensitive to 1-ulp changes.
//
// Built with -ffp-contract=off: the operation order below is the reference's, one rounding each.
#include "oisat_common.h"

namespace {

constexpr int kCurveBlocks = 1024;      // fixed: part of the reproducibility contract
constexpr int kCurveThreads = 256;
constexpr int kCurveWaves = kCurveBlocks * kCurveThreads / kWave;

// AK of the sweep (optimal_interpolation.py:27-31): K = t (t + So)^-1, Sb = (1 - K) t, AK = 1 - Sb / t.  For the 99 MEANS the
// last two steps are taken algebraically: Sb / t = 1 - K exactly in real arithmetic, so AK = K wherever t != 0, and 0/0 = NaN
// (dropped by nanmean) where t = 0 -- within one rounding of the reference's value (1.1e-16 absolute in float64, nine orders
// inside the curve's 1e-12 parity bar and four inside the smallest knee margin seen, 1.9e-6), for one division per (cell,
// scaling) instead of two: the sweep is bound by exactly those divisions.  The analysis kernel (oi_apply_kernel) keeps the
// reference's order: its AK FIELD is a returned quantity.
template <typename T>
__device__ __forceinline__ T ak_of(T sa, T so, T s) {
    T t = sa * s;
    T k = t * (T(1) / (t + so));
    return t == T(0) ? nan_of<T>() : k;
}

template <typename T>
__device__ __forceinline__ T bcast(T v, int src_lane);
template <>
__device__ __forceinline__ float bcast<float>(float v, int src_lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src_lane));
}
template <>
__device__ __forceinline__ double bcast<double>(double v, int src_lane) {
    long long b = __builtin_bit_cast(long long, v);
    int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffLL), src_lane);
    int hi = __builtin_amdgcn_readlane((int)(b >> 32), src_lane);
    long long r = ((long long)(unsigned)hi << 32) | (unsigned)lo;
    return __builtin_bit_cast(double, r);
}

// Lane layout of the sweep.  99 scalings on 64 lanes: with lane == scaling (two slots per lane, round 1) 29 of every 128
// lane-slots idle.  Here a wave walks its 64 cells TWO at a time -- lanes 0-31 take cell 2j, lanes 32-63 cell 2j+1 --
// and every lane of a half owns three scalings (l, l+32, l+64: 96 of them, all slots busy); the remaining scalings
// (96, 97, 98 of the reference's 99) are then evaluated with lane == cell and summed over the wave by a fixed xor
// tree, lane k keeping the running total of scaling 96+k.  Same arithmetic per (cell, scaling), 99 x 64 evaluations per
// 64 cells instead of 128 x 64; sums still in double, fixed launch shape and fixed combination order => reproducible.
constexpr int kMainScales = 96;

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, kWave);
    return v;
}
__device__ __forceinline__ unsigned wave_sum(unsigned v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, kWave);
    return v;
}

template <typename T>
__global__ __launch_bounds__(kCurveThreads) void oi_curve_kernel(const T* __restrict__ Sa, const T* __restrict__ So,
                                                                  int64_t n, const double* __restrict__ scales,
                                                                  int nscales, double* __restrict__ part_sum,
                                                                  unsigned* __restrict__ part_cnt) {
    const int lane = threadIdx.x & (kWave - 1);
    const int half = lane >> 5, l = lane & 31;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nmain = nscales < kMainScales ? nscales : kMainScales;
    const int nleft = nscales - nmain;                                  // scalings 96.. : lane == cell
    bool has[3];
    T sc[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        has[q] = l + 32 * q < nmain;
        sc[q] = has[q] ? (T)scales[l + 32 * q] : T(1);
    }
    double acc[3] = {0.0, 0.0, 0.0};
    unsigned cn[3] = {0u, 0u, 0u};
    double lacc = 0.0;                                                  // lane k: running sum of scaling 96 + k
    unsigned lcn = 0u;
    for (int64_t base = wave * kWave; base < n; base += (int64_t)kCurveWaves * kWave) {
        const int64_t i = base + lane;
        const T a = i < n ? Sa[i] : nan_of<T>();
        const T o = i < n ? So[i] : nan_of<T>();
        const int cnt = (n - base) < kWave ? (int)(n - base) : kWave;
        for (int j = 0; 2 * j < cnt; ++j) {
            const T a0 = bcast<T>(a, 2 * j), o0 = bcast<T>(o, 2 * j);
            const T a1 = bcast<T>(a, 2 * j + 1), o1 = bcast<T>(o, 2 * j + 1);
            const T aj = half ? a1 : a0, oj = half ? o1 : o0;
            const bool live = 2 * j + half < cnt;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const T ak = ak_of<T>(aj, oj, sc[q]);
                if (live && ak == ak) { acc[q] += (double)ak; ++cn[q]; }
            }
        }
        for (int k = 0; k < nleft; ++k) {                               // wave-uniform trip count
            const T ak = ak_of<T>(a, o, (T)scales[nmain + k]);
            const bool ok = i < n && ak == ak;
            const double s = wave_sum(ok ? (double)ak : 0.0);
            const unsigned c = wave_sum(ok ? 1u : 0u);
            if (lane == k) { lacc += s; lcn += c; }
        }
    }
    // the block's four waves (x two halves) are combined here in a fixed order (=> reproducible): one partial per block
    __shared__ double bs[2 * kCurveThreads / kWave][OISAT_MAX_SCALES];
    __shared__ unsigned bc[2 * kCurveThreads / kWave][OISAT_MAX_SCALES];
    const int wv = threadIdx.x >> 6;
    for (int k = threadIdx.x; k < 2 * (kCurveThreads / kWave) * OISAT_MAX_SCALES; k += kCurveThreads) {
        (&bs[0][0])[k] = 0.0;
        (&bc[0][0])[k] = 0u;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 3; ++q)
        if (has[q]) {
            bs[2 * wv + half][l + 32 * q] = acc[q];
            bc[2 * wv + half][l + 32 * q] = cn[q];
        }
    if (lane < nleft) {
        bs[2 * wv][nmain + lane] = lacc;
        bc[2 * wv][nmain + lane] = lcn;
    }
    __syncthreads();
    if (threadIdx.x < OISAT_MAX_SCALES) {
        double S = bs[0][threadIdx.x];
        unsigned Cn = bc[0][threadIdx.x];
        for (int k = 1; k < 2 * (kCurveThreads / kWave); ++k) { S += bs[k][threadIdx.x]; Cn += bc[k][threadIdx.x]; }
        part_sum[(int64_t)blockIdx.x * OISAT_MAX_SCALES + threadIdx.x] = S;
        part_cnt[(int64_t)blockIdx.x * OISAT_MAX_SCALES + threadIdx.x] = Cn;
    }
}

// NumPy's pairwise summation for n < 128 (8 running sums, then the remainder), so that the device
// knee pick uses bit-for-bit the threshold the host pick computes with np.mean(np.diff(xn)).
__device__ double numpy_sum_small(const double* a, int n) {
    if (n < 8) {
        double r = 0.0;
        for (int i = 0; i < n; ++i) r += a[i];
        return r;
    }
    double r[8];
    for (int j = 0; j < 8; ++j) r[j] = a[j];
    int i = 8;
    for (; i < n - (n % 8); i += 8)
        for (int j = 0; j < 8; ++j) r[j] += a[i + j];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += a[i];
    return res;
}

// Kneedle for an increasing concave curve -- the device twin of oisatgmi/_kneedle.py (itself a
// restatement of kneed.KneeLocator(x, y, direction='increasing').knee, optimal_interpolation.py:37-39).
// One workgroup of OISAT_MAX_SCALES threads: the element-wise steps (99 points) run one point per
// thread, the min/max, the NumPy-ordered mean and the threshold walk on thread 0.
// Returns -1 when no knee is found (the caller falls back to index 0, :40-41).
__device__ int kneedle_index_block(const double* x, const double* y, int n, double* w /* 4*n doubles of LDS */, int tid) {
    double* ds = w;            // interp1d(x, y)(x): left segment evaluated at the node
    double* xn = w + n;
    double* df = w + 2 * n;    // difference curve
    double* dx = w + 3 * n;
    __shared__ double mm[4];
    __shared__ int s_flag;
    if (tid < n) {
        if (tid == 0) ds[0] = y[0];
        else {
            const double slope = (y[tid] - y[tid - 1]) / (x[tid] - x[tid - 1]);
            ds[tid] = slope * (x[tid] - x[tid - 1]) + y[tid - 1];
        }
    }
    __syncthreads();
    if (tid == 0) {
        double xmin = x[0], xmax = x[0], ymin = ds[0], ymax = ds[0];
        bool ynan = ds[0] != ds[0];
        for (int i = 1; i < n; ++i) {
            xmin = x[i] < xmin ? x[i] : xmin;
            xmax = x[i] > xmax ? x[i] : xmax;
            if (ds[i] != ds[i]) ynan = true;
            ymin = ds[i] < ymin ? ds[i] : ymin;
            ymax = ds[i] > ymax ? ds[i] : ymax;
        }
        mm[0] = xmin; mm[1] = xmax; mm[2] = ymin; mm[3] = ymax;
        s_flag = (ynan || n < 3) ? 1 : 0;      // np.min/np.max propagate NaN -> every comparison below is False
    }
    __syncthreads();
    if (s_flag) return -1;
    if (tid < n) {
        xn[tid] = (x[tid] - mm[0]) / (mm[1] - mm[0]);
        df[tid] = (ds[tid] - mm[2]) / (mm[3] - mm[2]) - xn[tid];
    }
    __syncthreads();
    if (tid + 1 < n) dx[tid] = xn[tid + 1] - xn[tid];
    __syncthreads();
    if (tid != 0) return -1;                   // (only thread 0's return value is used)
    const double step = fabs(numpy_sum_small(dx, n - 1) / (double)(n - 1));
    // walk: thresholds reset at every local maximum (>= both neighbours, ends clipped), 0 at every local minimum
    int first = -1;
    for (int i = 0; i < n && first < 0; ++i) {
        const double l = df[i > 0 ? i - 1 : 0], r = df[i + 1 < n ? i + 1 : n - 1];
        if (df[i] >= l && df[i] >= r) first = i;
    }
    if (first < 0) return -1;
    double thr = __builtin_nan("");
    int at = -1;
    for (int i = first; i < n; ++i) {
        if (xn[i] == 1.0) return -1;
        const double l = df[i > 0 ? i - 1 : 0], r = df[i + 1 < n ? i + 1 : n - 1];
        if (df[i] >= l && df[i] >= r) { thr = df[i] - step; at = i; }
        if (df[i] <= l && df[i] <= r) thr = 0.0;
        if (df[i + 1] < thr) return at;
    }
    return -1;
}

// one block per scaling: 256 threads add the kCurveBlocks block partials (thread j takes j, j+256, ...),
// then a fixed binary tree in LDS -- same order every run => bitwise reproducible means
__global__ __launch_bounds__(256) void oi_curve_finish_kernel(const double* __restrict__ part_sum,
                                                               const unsigned* __restrict__ part_cnt,
                                                               double* __restrict__ mean_out, long long* __restrict__ cnt_out) {
    __shared__ double ss[256];
    __shared__ long long sc[256];
    const int t = blockIdx.x, j = threadIdx.x;
    double s = 0.0;
    long long c = 0;
#pragma unroll
    for (int k = 0; k < kCurveBlocks / 256; ++k) {
        s += part_sum[(int64_t)(j + 256 * k) * OISAT_MAX_SCALES + t];
        c += part_cnt[(int64_t)(j + 256 * k) * OISAT_MAX_SCALES + t];
    }
    ss[j] = s;
    sc[j] = c;
    __syncthreads();
    for (int h = 128; h > 0; h >>= 1) {
        if (j < h) { ss[j] += ss[j + h]; sc[j] += sc[j + h]; }
        __syncthreads();
    }
    if (j == 0) {
        mean_out[t] = ss[0] / (double)sc[0];      // 0/0 -> NaN like np.nanmean of an all-NaN slice
        cnt_out[t] = sc[0];
    }
}

// the knee pick on the device (one thread; 99 numbers)
__global__ void oi_knee_kernel(const double* __restrict__ scales, const double* __restrict__ mean, int nscales, int pick_knee,
                               int forced_index, int* __restrict__ index_out) {
    __shared__ double scratch[4 * OISAT_MAX_SCALES];
    __shared__ double smean[OISAT_MAX_SCALES], sx[OISAT_MAX_SCALES];
    if (threadIdx.x < nscales) { smean[threadIdx.x] = mean[threadIdx.x]; sx[threadIdx.x] = scales[threadIdx.x]; }
    __syncthreads();
    int idx = 0;
    if (forced_index >= 0) idx = forced_index;
    else if (pick_knee) {                       // block-uniform branch: every thread takes part
        const int k = kneedle_index_block(sx, smean, nscales, scratch, threadIdx.x);
        idx = k < 0 ? 0 : k;
    }
    if (threadIdx.x == 0) *index_out = idx;
}

template <typename T>
__global__ __launch_bounds__(256) void oi_apply_kernel(const T* __restrict__ Xa, T* __restrict__ Y,
                                                        const T* __restrict__ Sa, const T* __restrict__ So,
                                                        int64_t n, T s_host, const double* __restrict__ scales_dev,
                                                        const int* __restrict__ idx_dev, T* __restrict__ Xb,
                                                        T* __restrict__ AK, T* __restrict__ inc, T* __restrict__ err) {
    const T s = idx_dev ? (T)scales_dev[*idx_dev] : s_host;      // fused path: the index was picked on the device
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        T y = Y[i];
        if (y < T(0)) {                       // optimal_interpolation.py:14, in place
            y = T(0);
            Y[i] = y;
        }
        const T xa = Xa[i];
        const T t = Sa[i] * s;
        const T k = t * (T(1) / (t + So[i]));
        const T sb = (T(1) - k) * t;
        const T d = k * (y - xa);
        if (AK) AK[i] = T(1) - sb / t;
        if (inc) inc[i] = d;
        if (Xb) Xb[i] = xa + d;
        if (err) err[i] = sqrt(sb);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void affine_kernel(const T* __restrict__ x, int64_t n, T off, T slope, T* __restrict__ out) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = (x[i] - off) / slope;
}

template <typename T>
__global__ __launch_bounds__(256) void scaling_factor_kernel(const T* __restrict__ post, const T* __restrict__ prior, int64_t n,
                                                              T* __restrict__ out) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const T r = post[i] / prior[i];
        const bool bad = (r != r) || r == __builtin_inf() || r == -__builtin_inf() || r == T(0);
        out[i] = bad ? T(1) : r;                 // driver.py:204-206
    }
}

template <typename T>
__global__ __launch_bounds__(256) void variances_kernel(const T* __restrict__ xa, const T* __restrict__ e, int64_t n, T pct,
                                                         T* __restrict__ Sa, T* __restrict__ So) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        if (Sa) {
            const T v = xa[i] * pct / T(100);          // (Xa*error_ctm/100.0)**2, driver.py:111
            Sa[i] = v * v;
        }
        if (So) So[i] = e[i] * e[i];
    }
}

struct CurveWs {
    char* ws;
    static constexpr size_t off_scales = 0;
    static constexpr size_t off_psum = 1024;
    static constexpr size_t off_pcnt = off_psum + sizeof(double) * kCurveBlocks * OISAT_MAX_SCALES;
    static constexpr size_t off_mean = off_pcnt + sizeof(unsigned) * kCurveBlocks * OISAT_MAX_SCALES;
    static constexpr size_t off_cnt = off_mean + sizeof(double) * OISAT_MAX_SCALES;
    static constexpr size_t off_idx = off_cnt + sizeof(long long) * OISAT_MAX_SCALES;
    static constexpr size_t total = off_idx + 64;
};

// enqueue sweep + finish (+ optional device knee pick); results stay in workspace slot 0
template <typename T>
int curve_enqueue(oisat_ctx* h, const T* Sa, const T* So, int64_t n, const double* scales, int nscales, int pick_knee,
                  int forced_index, CurveWs& cw) {
    cw.ws = (char*)oisat_ws(h, 0, CurveWs::total);
    if (!cw.ws) return OISAT_ENOMEM;
    // the scaling sweep is the same 99 numbers call after call: upload it only when it changes, so the
    // steady-state fused path has no host synchronisation at all
    if (h->scales_dev != cw.ws || h->scales_n != nscales || memcmp(h->scales_host, scales, sizeof(double) * nscales) != 0) {
        char* pin = (char*)oisat_pinned(h, 4096);
        if (!pin) return OISAT_ENOMEM;
        HIP_TRY(hipStreamSynchronize(h->stream));          // earlier async copies may still read the staging block
        memcpy(pin, scales, sizeof(double) * nscales);
        HIP_TRY(hipMemcpyAsync(cw.ws + CurveWs::off_scales, pin, sizeof(double) * nscales, hipMemcpyHostToDevice, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
        memcpy(h->scales_host, scales, sizeof(double) * nscales);
        h->scales_n = nscales;
        h->scales_dev = cw.ws;
    }
    OISAT_LAUNCH(h, "oi_curve", (oi_curve_kernel<T>), dim3(kCurveBlocks), dim3(kCurveThreads), 0, Sa, So, n,
                 (const double*)(cw.ws + CurveWs::off_scales), nscales, (double*)(cw.ws + CurveWs::off_psum),
                 (unsigned*)(cw.ws + CurveWs::off_pcnt));
    OISAT_LAUNCH(h, "oi_curve_finish", oi_curve_finish_kernel, dim3(nscales), dim3(256), 0, (const double*)(cw.ws + CurveWs::off_psum),
                 (const unsigned*)(cw.ws + CurveWs::off_pcnt), (double*)(cw.ws + CurveWs::off_mean),
                 (long long*)(cw.ws + CurveWs::off_cnt));
    if (pick_knee || forced_index >= 0) {
        OISAT_LAUNCH(h, "oi_knee", oi_knee_kernel, dim3(1), dim3(OISAT_MAX_SCALES), 0, (const double*)(cw.ws + CurveWs::off_scales),
                     (const double*)(cw.ws + CurveWs::off_mean), nscales, pick_knee, forced_index, (int*)(cw.ws + CurveWs::off_idx));
    }
    return OISAT_OK;
}

template <typename T>
int curve_impl(oisat_ctx* h, const T* Sa, const T* So, int64_t n, const double* scales, int nscales, double* mean_out,
               int64_t* count_out) {
    CurveWs cw;
    const int rc = curve_enqueue<T>(h, Sa, So, n, scales, nscales, 0, -1, cw);
    if (rc) return rc;
    char* pin = (char*)oisat_pinned(h, 4096);
    HIP_TRY(hipMemcpyAsync(pin + 1024, cw.ws + CurveWs::off_mean,
                           sizeof(double) * OISAT_MAX_SCALES + sizeof(long long) * OISAT_MAX_SCALES, hipMemcpyDeviceToHost,
                           h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    memcpy(mean_out, pin + 1024, sizeof(double) * nscales);
    if (count_out) {
        const long long* c = (const long long*)(pin + 1024 + sizeof(double) * OISAT_MAX_SCALES);
        for (int i = 0; i < nscales; ++i) count_out[i] = (int64_t)c[i];
    }
    return OISAT_OK;
}

template <typename T>
int fused_impl(oisat_ctx* h, const T* Xa, T* Y, const T* Sa, const T* So, int64_t n, const double* scales, int nscales,
               int forced_index, T* Xb, T* AK, T* inc, T* err, int32_t* index_dev, double* curve_dev) {
    CurveWs cw;
    const int rc = curve_enqueue<T>(h, Sa, So, n, scales, nscales, 1, forced_index, cw);
    if (rc) return rc;
    const int grid = stream_grid(n, 256);
    OISAT_LAUNCH(h, "oi_apply", (oi_apply_kernel<T>), dim3(grid), dim3(256), 0, Xa, Y, Sa, So, n, T(1),
                 (const double*)(cw.ws + CurveWs::off_scales), (const int*)(cw.ws + CurveWs::off_idx), Xb, AK, inc, err);
    if (index_dev)
        HIP_TRY(hipMemcpyAsync(index_dev, cw.ws + CurveWs::off_idx, sizeof(int32_t), hipMemcpyDeviceToDevice, h->stream));
    if (curve_dev)
        HIP_TRY(hipMemcpyAsync(curve_dev, cw.ws + CurveWs::off_mean, sizeof(double) * nscales, hipMemcpyDeviceToDevice, h->stream));
    return OISAT_OK;
}

}  // namespace

extern "C" int oisat_oi_curve(oisat_ctx* h, int dtype, const void* Sa, const void* So, int64_t n, const double* scales,
                              int nscales, double* mean_out, int64_t* count_out) {
    ARG_CHECK(h && Sa && So && scales && mean_out);
    ARG_CHECK(n > 0 && nscales > 0 && nscales <= OISAT_MAX_SCALES);
    ARG_CHECK(dtype == OISAT_F32 || dtype == OISAT_F64);
    if (dtype == OISAT_F32) return curve_impl<float>(h, (const float*)Sa, (const float*)So, n, scales, nscales, mean_out, count_out);
    return curve_impl<double>(h, (const double*)Sa, (const double*)So, n, scales, nscales, mean_out, count_out);
}

extern "C" int oisat_oi_fused(oisat_ctx* h, int dtype, const void* Xa, void* Y, const void* Sa, const void* So, int64_t n,
                              const double* scales, int nscales, int forced_index, void* Xb, void* AK, void* inc, void* err,
                              int32_t* index_dev, double* curve_dev) {
    ARG_CHECK(h && Xa && Y && Sa && So && scales && n > 0);
    ARG_CHECK(nscales > 0 && nscales <= OISAT_MAX_SCALES && forced_index < nscales);
    ARG_CHECK(dtype == OISAT_F32 || dtype == OISAT_F64);
    if (dtype == OISAT_F32)
        return fused_impl<float>(h, (const float*)Xa, (float*)Y, (const float*)Sa, (const float*)So, n, scales, nscales,
                                 forced_index, (float*)Xb, (float*)AK, (float*)inc, (float*)err, index_dev, curve_dev);
    return fused_impl<double>(h, (const double*)Xa, (double*)Y, (const double*)Sa, (const double*)So, n, scales, nscales,
                              forced_index, (double*)Xb, (double*)AK, (double*)inc, (double*)err, index_dev, curve_dev);
}

extern "C" int oisat_oi_apply(oisat_ctx* h, int dtype, const void* Xa, void* Y, const void* Sa, const void* So, int64_t n,
                              double scale, void* Xb, void* AK, void* inc, void* err) {
    ARG_CHECK(h && Xa && Y && Sa && So && n > 0);
    ARG_CHECK(dtype == OISAT_F32 || dtype == OISAT_F64);
    const int grid = stream_grid(n, 256);
    if (dtype == OISAT_F32) {
        OISAT_LAUNCH(h, "oi_apply", (oi_apply_kernel<float>), dim3(grid), dim3(256), 0, (const float*)Xa, (float*)Y,
                     (const float*)Sa, (const float*)So, n, (float)scale, (const double*)nullptr, (const int*)nullptr, (float*)Xb,
                     (float*)AK, (float*)inc, (float*)err);
    } else {
        OISAT_LAUNCH(h, "oi_apply", (oi_apply_kernel<double>), dim3(grid), dim3(256), 0, (const double*)Xa, (double*)Y,
                     (const double*)Sa, (const double*)So, n, scale, (const double*)nullptr, (const int*)nullptr, (double*)Xb, (double*)AK,
                     (double*)inc, (double*)err);
    }
    return OISAT_OK;
}

extern "C" int oisat_affine(oisat_ctx* h, int dtype, const void* x, int64_t n, double offset, double slope, void* out) {
    ARG_CHECK(h && x && out && n > 0);
    ARG_CHECK(dtype == OISAT_F32 || dtype == OISAT_F64);
    const int grid = stream_grid(n, 256);
    if (dtype == OISAT_F32) {
        OISAT_LAUNCH(h, "affine", (affine_kernel<float>), dim3(grid), dim3(256), 0, (const float*)x, n, (float)offset,
                     (float)slope, (float*)out);
    } else {
        OISAT_LAUNCH(h, "affine", (affine_kernel<double>), dim3(grid), dim3(256), 0, (const double*)x, n, offset, slope,
                     (double*)out);
    }
    return OISAT_OK;
}

extern "C" int oisat_oi_variances(oisat_ctx* h, int dtype, const void* Xa, const void* sat_err, int64_t n, double error_ctm,
                                  void* Sa_out, void* So_out) {
    ARG_CHECK(h && n > 0 && (Sa_out || So_out));
    ARG_CHECK((!Sa_out || Xa) && (!So_out || sat_err));
    ARG_CHECK(dtype == OISAT_F32 || dtype == OISAT_F64);
    const int grid = stream_grid(n, 256);
    if (dtype == OISAT_F32) {
        OISAT_LAUNCH(h, "oi_variances", (variances_kernel<float>), dim3(grid), dim3(256), 0, (const float*)Xa,
                     (const float*)sat_err, n, (float)error_ctm, (float*)Sa_out, (float*)So_out);
    } else {
        OISAT_LAUNCH(h, "oi_variances", (variances_kernel<double>), dim3(grid), dim3(256), 0, (const double*)Xa,
                     (const double*)sat_err, n, error_ctm, (double*)Sa_out, (double*)So_out);
    }
    return OISAT_OK;
}

extern "C" int oisat_scaling_factor(oisat_ctx* h, int dtype, const void* posterior, const void* prior, int64_t n, void* out) {
    ARG_CHECK(h && posterior && prior && out && n > 0);
    ARG_CHECK(dtype == OISAT_F32 || dtype == OISAT_F64);
    const int grid = stream_grid(n, 256);
    if (dtype == OISAT_F32) {
        OISAT_LAUNCH(h, "scaling_factor", (scaling_factor_kernel<float>), dim3(grid), dim3(256), 0, (const float*)posterior,
                     (const float*)prior, n, (float*)out);
    } else {
        OISAT_LAUNCH(h, "scaling_factor", (scaling_factor_kernel<double>), dim3(grid), dim3(256), 0, (const double*)posterior,
                     (const double*)prior, n, (double*)out);
    }
    return OISAT_OK;
}

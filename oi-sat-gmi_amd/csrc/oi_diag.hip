// Element-wise optimal interpolation: optimal_interpolation.py:6-52 of the reference.
//
//   sweep   (:26-33)  for each of the 99 scalings s: t = Sa*s, K = t*(t+So)^-1, Sb = (1-K)*t,
//                     AK = 1 - Sb/t, mean_s = nanmean(AK)
//   apply   (:14,:46-52) Y[Y<0]=0; inc = K*(Y-Xa); Xb = Xa+inc; err = sqrt(Sb) for the chosen s
//
// HBM layout: four (ny*nx) fields of T in, four out, all contiguous; nothing else.
// The sweep never materialises the 99x3 temporaries the reference keeps alive: a wave loads 64
// cells coalesced, then walks them with lane == scaling (lanes 0..63 own scale `lane`, and scale
// `lane+64`), broadcasting one cell at a time with v_readlane.  Every lane therefore accumulates
// its own two (sum, count) pairs in double with NO cross-lane reduction; a second tiny kernel adds
// the per-wave partials in wave order.  The launch shape is fixed, so the 99 means are bitwise
// reproducible run to run -- the knee pick that follows is sensitive to 1-ulp changes.
//
// Built with -ffp-contract=off: the operation order below is the reference's, one rounding each.
#include "oisat_common.h"

namespace {

constexpr int kCurveBlocks = 1024;      // fixed: part of the reproducibility contract
constexpr int kCurveThreads = 256;
constexpr int kCurveWaves = kCurveBlocks * kCurveThreads / kWave;

template <typename T>
__device__ __forceinline__ T ak_of(T sa, T so, T s) {
    T t = sa * s;
    T k = t * (T(1) / (t + so));
    T sb = (T(1) - k) * t;
    return T(1) - sb / t;
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

template <typename T>
__global__ __launch_bounds__(kCurveThreads) void oi_curve_kernel(const T* __restrict__ Sa, const T* __restrict__ So,
                                                                  int64_t n, const double* __restrict__ scales,
                                                                  int nscales, double* __restrict__ part_sum,
                                                                  unsigned* __restrict__ part_cnt) {
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const bool has0 = lane < nscales;
    const bool has1 = lane + kWave < nscales;
    const T s0 = has0 ? (T)scales[lane] : T(1);
    const T s1 = has1 ? (T)scales[lane + kWave] : T(1);
    double acc0 = 0.0, acc1 = 0.0;
    unsigned c0 = 0, c1 = 0;
    for (int64_t base = wave * kWave; base < n; base += (int64_t)kCurveWaves * kWave) {
        const int64_t i = base + lane;
        const T a = i < n ? Sa[i] : nan_of<T>();
        const T o = i < n ? So[i] : nan_of<T>();
        const int cnt = (n - base) < kWave ? (int)(n - base) : kWave;
        for (int j = 0; j < cnt; ++j) {
            const T aj = bcast<T>(a, j);
            const T oj = bcast<T>(o, j);
            const T ak0 = ak_of<T>(aj, oj, s0);
            if (ak0 == ak0) { acc0 += (double)ak0; ++c0; }
            if (nscales > kWave) {
                const T ak1 = ak_of<T>(aj, oj, s1);
                if (ak1 == ak1) { acc1 += (double)ak1; ++c1; }
            }
        }
    }
    part_sum[wave * OISAT_MAX_SCALES + lane] = has0 ? acc0 : 0.0;
    part_sum[wave * OISAT_MAX_SCALES + kWave + lane] = has1 ? acc1 : 0.0;
    part_cnt[wave * OISAT_MAX_SCALES + lane] = has0 ? c0 : 0u;
    part_cnt[wave * OISAT_MAX_SCALES + kWave + lane] = has1 ? c1 : 0u;
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
// Single thread, 99 points.  Returns -1 when no knee is found (the caller falls back to index 0, :40-41).
__device__ int kneedle_index(const double* x, const double* y, int n, double* w /* 4*n doubles of scratch */) {
    if (n < 3) return -1;
    double* ds = w;            // interp1d(x, y)(x): left segment evaluated at the node
    double* xn = w + n;
    double* df = w + 2 * n;    // difference curve
    double* dx = w + 3 * n;
    ds[0] = y[0];
    for (int i = 1; i < n; ++i) {
        const double slope = (y[i] - y[i - 1]) / (x[i] - x[i - 1]);
        ds[i] = slope * (x[i] - x[i - 1]) + y[i - 1];
    }
    double xmin = x[0], xmax = x[0], ymin = ds[0], ymax = ds[0];
    bool ynan = ds[0] != ds[0];
    for (int i = 1; i < n; ++i) {
        xmin = x[i] < xmin ? x[i] : xmin;
        xmax = x[i] > xmax ? x[i] : xmax;
        if (ds[i] != ds[i]) ynan = true;
        ymin = ds[i] < ymin ? ds[i] : ymin;
        ymax = ds[i] > ymax ? ds[i] : ymax;
    }
    if (ynan) return -1;       // np.min/np.max propagate NaN -> every comparison below is False
    for (int i = 0; i < n; ++i) {
        xn[i] = (x[i] - xmin) / (xmax - xmin);
        df[i] = (ds[i] - ymin) / (ymax - ymin) - xn[i];
    }
    for (int i = 0; i + 1 < n; ++i) dx[i] = xn[i + 1] - xn[i];
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

// one block of 1024 threads = 128 scales x 8 slices of the per-wave partials, combined in slice order
// (fixed order => bitwise reproducible); thread 0 then optionally picks the knee on the device.
__global__ __launch_bounds__(1024) void oi_curve_finish_kernel(const double* __restrict__ part_sum,
                                                                const unsigned* __restrict__ part_cnt, int nscales,
                                                                const double* __restrict__ scales, double* __restrict__ mean_out,
                                                                long long* __restrict__ cnt_out, int pick_knee, int forced_index,
                                                                int* __restrict__ index_out) {
    __shared__ double ssum[8][OISAT_MAX_SCALES];
    __shared__ long long scnt[8][OISAT_MAX_SCALES];
    __shared__ double scratch[4 * OISAT_MAX_SCALES];
    __shared__ double smean[OISAT_MAX_SCALES], sx[OISAT_MAX_SCALES];
    const int t = threadIdx.x & (OISAT_MAX_SCALES - 1), sl = threadIdx.x >> 7;
    constexpr int per = kCurveWaves / 8;
    double s = 0.0;
    long long c = 0;
    for (int w = sl * per; w < (sl + 1) * per; ++w) {
        s += part_sum[(int64_t)w * OISAT_MAX_SCALES + t];
        c += part_cnt[(int64_t)w * OISAT_MAX_SCALES + t];
    }
    ssum[sl][t] = s;
    scnt[sl][t] = c;
    __syncthreads();
    if (sl == 0 && t < nscales) {
        double S = ssum[0][t];
        long long Cn = scnt[0][t];
        for (int k = 1; k < 8; ++k) { S += ssum[k][t]; Cn += scnt[k][t]; }
        const double mean = S / (double)Cn;      // 0/0 -> NaN like np.nanmean of an all-NaN slice
        mean_out[t] = mean;
        cnt_out[t] = Cn;
        smean[t] = mean;
        sx[t] = scales[t];
    }
    __syncthreads();
    if (threadIdx.x == 0 && index_out) {
        int idx = 0;
        if (forced_index >= 0) idx = forced_index;
        else if (pick_knee) {
            const int k = kneedle_index(sx, smean, nscales, scratch);
            idx = k < 0 ? 0 : k;
        }
        *index_out = idx;
    }
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
    static constexpr size_t off_pcnt = off_psum + sizeof(double) * kCurveWaves * OISAT_MAX_SCALES;
    static constexpr size_t off_mean = off_pcnt + sizeof(unsigned) * kCurveWaves * OISAT_MAX_SCALES;
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
    OISAT_LAUNCH(h, "oi_curve_finish", oi_curve_finish_kernel, dim3(1), dim3(1024), 0, (const double*)(cw.ws + CurveWs::off_psum),
                 (const unsigned*)(cw.ws + CurveWs::off_pcnt), nscales, (const double*)(cw.ws + CurveWs::off_scales),
                 (double*)(cw.ws + CurveWs::off_mean), (long long*)(cw.ws + CurveWs::off_cnt), pick_knee, forced_index,
                 (int*)(cw.ws + CurveWs::off_idx));
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

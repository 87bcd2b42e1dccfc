// Vertical operators immediately upstream of the monthly averaging: AMF recalculation
// (amf_recal.py:51-56, :85-119 and :160-183 of the reference) and, for
// optimal-estimation products, the averaging-kernel convolution (ak_conv_mopitt.py:60-146,
// ak_conv_gosat.py:60-143) -- same shape of work: a model column interpolated in log-pressure per pixel.
//
// The reference walks every pixel in a Python double loop and builds a scipy interp1d object per
// pixel (:97-119).  Here a thread owns a pixel: its scattering-weight profile is sorted by
// log-pressure in thread-private memory (interp1d sorts with a stable argsort), every model level is
// located by binary search and interpolated / extrapolated with the operation order of scipy's
// _call_linear, and the two column sums are taken in NumPy's pairwise order, so float64 results
// agree with the reference to the last bits.
//
// HBM layout: level-major cubes [nz][ny*nx] (what the readers and interpolator() produce), so for
// each level a wave reads 64 consecutive pixels: coalesced.  Everything is double.
#include "oisat_common.h"

namespace {

constexpr int kMaxSat = 64;      // satellite levels (OMI NO2: 35)
constexpr int kMaxCtm = 128;     // model levels (GMI: 72)

// (deltap * profile / g / Mair * N_A * 1e-4 * 1e-15 * 100.0 * 1e-9), evaluated left to right in T;
// profile == nullptr: the air column deltap / g / Mair * N_A * 1e-4 * 1e-15 * 100.0 (ak_conv_mopitt.py:66)
template <typename T>
__global__ __launch_bounds__(256) void partial_column_kernel(const T* __restrict__ deltap, const T* __restrict__ profile, int64_t n,
                                                              T* __restrict__ out) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        T v = profile ? deltap[i] * profile[i] : deltap[i];
        v = v / T(9.80665);
        v = v / T(28.97e-3);
        v = v * T(6.02214076e23);
        v = v * T(1e-4);
        v = v * T(1e-15);
        v = v * T(100.0);
        if (profile) v = v * T(1e-9);
        out[i] = v;
    }
}

// np.sum of n (<= 128) values in NumPy's pairwise order: 8 running sums, then the remainder
template <typename T>
__device__ T numpy_sum_le128(const T* a, int n) {
    if (n < 8) {
        T r = T(0);
        for (int i = 0; i < n; ++i) r += a[i];
        return r;
    }
    T r0 = a[0], r1 = a[1], r2 = a[2], r3 = a[3], r4 = a[4], r5 = a[5], r6 = a[6], r7 = a[7];
    int i = 8;
    for (; i < n - (n % 8); i += 8) {
        r0 += a[i]; r1 += a[i + 1]; r2 += a[i + 2]; r3 += a[i + 3];
        r4 += a[i + 4]; r5 += a[i + 5]; r6 += a[i + 6]; r7 += a[i + 7];
    }
    T res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
    for (; i < n; ++i) res += a[i];
    return res;
}

__device__ __forceinline__ bool before(double a, double b) {        // NaN sorts last, like np.argsort
    if (a != a) return false;
    if (b != b) return true;
    return a < b;
}

// T = dtype of the model cubes.  NumPy keeps it where the reference does: np.log(ctm_p) and
// np.nansum(partial column) are evaluated in T, everything multiplied by a float64 array is float64.
template <typename T>
__global__ __launch_bounds__(128) void amf_recal_kernel(const double* __restrict__ sat_p, const double* __restrict__ sat_sw, int nzs,
                                                         const T* __restrict__ ctm_p, const T* __restrict__ ctm_pc, int nzc,
                                                         const double* __restrict__ trop, const double* __restrict__ vcd,
                                                         const double* __restrict__ amf, int64_t n, double* __restrict__ new_amf,
                                                         double* __restrict__ vcd_out, double* __restrict__ ctm_vcd) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const double nan = __builtin_nan("");
    const double v0 = vcd[p];
    if (v0 != v0) {                       // amf_recal.py:99-100 skips the pixel; :176,:180 leave NaN everywhere
        new_amf[p] = nan;
        vcd_out[p] = (amf[p] * v0) / nan;
        ctm_vcd[p] = nan;
        return;
    }
    double xs[kMaxSat], ys[kMaxSat];
    for (int k = 0; k < nzs; ++k) {       // stable insertion sort by log-pressure
        const double x = log(sat_p[(int64_t)k * n + p]), y = sat_sw[(int64_t)k * n + p];
        int j = k;
        while (j > 0 && before(x, xs[j - 1])) { xs[j] = xs[j - 1]; ys[j] = ys[j - 1]; --j; }
        xs[j] = x;
        ys[j] = y;
    }
    double prod[kMaxCtm];
    T part[kMaxCtm];
    const bool has_trop = trop != nullptr;
    const double tp = has_trop ? trop[p] : 0.0;
    for (int c = 0; c < nzc; ++c) {
        const T pcT = ctm_p[(int64_t)c * n + p];
        const double pc = (double)pcT;
        double col = (double)ctm_pc[(int64_t)c * n + p];
        const double xq = (double)(T)log(pc);              // correctly rounded log in the model dtype (np.log(float32 array) is float32)
        int lo = 0, hi = nzs;              // np.searchsorted(xs, xq, 'left'); NaN query -> nzs
        if (xq != xq) lo = nzs;
        else
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (before(xs[mid], xq)) lo = mid + 1; else hi = mid;
            }
        int idx = lo < 1 ? 1 : (lo > nzs - 1 ? nzs - 1 : lo);
        const double x_lo = xs[idx - 1], x_hi = xs[idx], y_lo = ys[idx - 1], y_hi = ys[idx];
        const double slope = (y_hi - y_lo) / (x_hi - x_lo);
        double sw = slope * (xq - x_lo) + y_lo;
        if (sw == __builtin_inf() || sw == -__builtin_inf()) sw = 0.0;       // :109
        if (has_trop && pc < tp) { sw = nan; col = nan; }                     // :111-114
        const double pr = sw * col;
        prod[c] = (pr != pr) ? 0.0 : pr;   // np.nansum: NaN -> 0, then np.sum
        part[c] = (col != col) ? T(0) : (T)col;
    }
    const double scd = numpy_sum_le128<double>(prod, nzc);
    const double mv = (double)numpy_sum_le128<T>(part, nzc);
    const double a_new = (mv != 0.0) ? scd / mv : nan;                        // :117
    new_amf[p] = a_new;
    const double vc = (amf[p] * v0) / a_new;                                  // :179
    vcd_out[p] = vc;
    ctm_vcd[p] = (vc != vc || vc == __builtin_inf() || vc == -__builtin_inf()) ? nan : mv;      // :180-181
}

// no scattering weights: model VCD = nansum over levels of the (tropopause-masked) partial columns, :162-166
template <typename T>
__global__ __launch_bounds__(256) void column_sum_kernel(const T* __restrict__ ctm_p, const T* __restrict__ ctm_pc, int nzc,
                                                          const double* __restrict__ trop, const double* __restrict__ vcd, int64_t n,
                                                          T* __restrict__ ctm_vcd) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += stride) {
        T s = T(0);                         // np.nansum(axis=0) in the cube's dtype: level after level
        for (int c = 0; c < nzc; ++c) {
            T col = ctm_pc[(int64_t)c * n + p];
            if (trop && (double)ctm_p[(int64_t)c * n + p] < trop[p]) col = nan_of<T>();
            if (col == col) s += col;
        }
        const double v0 = vcd[p];
        ctm_vcd[p] = (v0 != v0) ? nan_of<T>() : s;
    }
}

// ---- model precipitable water (pwv_cal.py:60-99) -----------------------------------------------------
// deltap*profile/g/10000.0, left to right in T (:63,:70)
template <typename T>
__global__ __launch_bounds__(256) void water_column_kernel(const T* __restrict__ deltap, const T* __restrict__ profile, int64_t n,
                                                            T* __restrict__ out) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        T v = deltap[i] * profile[i];
        v = v / T(9.80665);
        v = v / T(10000.0);
        out[i] = v;
    }
}

// np.nansum(partial/1000.0, axis=0) level after level in T, then NaN where the observation is NaN or +/-inf (:96-98)
template <typename T>
__global__ __launch_bounds__(256) void pwv_sum_kernel(const T* __restrict__ partial, int nz, const double* __restrict__ vcd, int64_t n,
                                                       T* __restrict__ out) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += stride) {
        T s = T(0);
        for (int k = 0; k < nz; ++k) {
            const T v = partial[(int64_t)k * n + p] / T(1000.0);
            if (v == v) s += v;
        }
        const double v0 = vcd[p];
        out[p] = (v0 != v0 || v0 == __builtin_inf() || v0 == -__builtin_inf()) ? nan_of<T>() : s;
    }
}

// ---- averaging-kernel convolution ------------------------------------------------------------------
// interp1d(np.log(model pressure), model profile)(np.log(satellite pressure)) for one pixel: the model column is
// sorted by log-pressure in the cubes' dtype T (stable, NaN last), a satellite level is located by binary search and
// evaluated with scipy's _call_linear order -- the slope in T, the rest in double, exactly as NumPy promotes
// float32 model arrays against the float64 satellite levels.
template <typename T>
struct ModelColumn {
    T xs[kMaxCtm], ys[kMaxCtm];
    int n;
    __device__ void load(const T* __restrict__ pmid, const T* __restrict__ prof, int nzc, int64_t stride, int64_t p) {
        n = nzc;
        for (int k = 0; k < nzc; ++k) {
            const T x = (T)log(pmid[(int64_t)k * stride + p]), y = prof[(int64_t)k * stride + p];
            int j = k;
            while (j > 0 && before((double)x, (double)xs[j - 1])) { xs[j] = xs[j - 1]; ys[j] = ys[j - 1]; --j; }
            xs[j] = x;
            ys[j] = y;
        }
    }
    __device__ double at(double xq, bool extrapolate) const {
        int lo = 0, hi = n;                 // np.searchsorted(xs, xq, 'left'); NaN query -> n
        if (xq != xq) lo = n;
        else
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (before((double)xs[mid], xq)) lo = mid + 1; else hi = mid;
            }
        const int idx = lo < 1 ? 1 : (lo > n - 1 ? n - 1 : lo);
        const T x_lo = xs[idx - 1], x_hi = xs[idx], y_lo = ys[idx - 1], y_hi = ys[idx];
        const T slope = (y_hi - y_lo) / (x_hi - x_lo);
        double y = (double)slope * (xq - (double)x_lo) + (double)y_lo;
        if (!extrapolate && (xq < (double)xs[0] || xq > (double)xs[n - 1])) y = __builtin_nan("");   // fill_value=nan, bounds_error=False
        return y;
    }
};

// MOPITT (ak_conv_mopitt.py:118-138): log10-space averaging kernels, one surface row + nzs profile rows
template <typename T>
__global__ __launch_bounds__(128) void ak_conv_mopitt_kernel(const T* __restrict__ ctm_p, const T* __restrict__ ctm_prof,
                                                              const T* __restrict__ ctm_air, int nzc, const double* __restrict__ sat_p,
                                                              const double* __restrict__ ak, const double* __restrict__ ap_prof, int nzs,
                                                              const double* __restrict__ ap_col, const double* __restrict__ ap_surf,
                                                              const double* __restrict__ vcd, int64_t n, double* __restrict__ model_vcd,
                                                              double* __restrict__ model_xcol) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const double nan = __builtin_nan("");
    const double v0 = vcd[p];
    if (v0 != v0) {                        // :120-121 skips the pixel: both outputs stay NaN
        model_vcd[p] = nan;
        model_xcol[p] = nan;
        return;
    }
    ModelColumn<T> col;
    col.load(ctm_p, ctm_prof, nzc, n, p);
    double term[kMaxSat];
    for (int k = 0; k < nzs; ++k) {
        const double xi = col.at(log(sat_p[(int64_t)k * n + p]), false);
        const double t = ak[(int64_t)(k + 1) * n + p] * (log10(xi) - log10(ap_prof[(int64_t)k * n + p]));
        term[k] = (t != t) ? 0.0 : t;      // np.nansum
    }
    const double prof_part = ap_col[p] + numpy_sum_le128<double>(term, nzs);
    const T surf = ctm_prof[p];            // model level 0 (before sorting), log10 in the cubes' dtype
    const double surf_part = ak[p] * ((double)(T)log10(surf) - log10(ap_surf[p]));
    const double v = prof_part + surf_part;
    T air[kMaxCtm];
    for (int c = 0; c < nzc; ++c) {
        const T a = ctm_air[(int64_t)c * n + p];
        air[c] = (a != a) ? T(0) : a;
    }
    const double air_sum = (double)numpy_sum_le128<T>(air, nzc);
    model_xcol[p] = 1e6 * v / air_sum;     // ppmv, :138
    model_vcd[p] = (v0 == __builtin_inf() || v0 == -__builtin_inf()) ? nan : v;     // :141-142
}

// GOSAT (ak_conv_gosat.py:118-135): XCH4 = nansum over levels of pressure-weighted a-priori + AK*(model - a-priori)
template <typename T>
__global__ __launch_bounds__(128) void ak_conv_gosat_kernel(const T* __restrict__ ctm_p, const T* __restrict__ ctm_prof, int nzc,
                                                             const double* __restrict__ sat_p, const double* __restrict__ ak,
                                                             const double* __restrict__ ap_prof, const double* __restrict__ pw, int nzs,
                                                             const double* __restrict__ x_col, int64_t n,
                                                             double* __restrict__ model_xcol) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const double x0 = x_col[p];
    if (x0 != x0 || x0 == __builtin_inf() || x0 == -__builtin_inf()) {      // :119-120 and :139-140
        model_xcol[p] = __builtin_nan("");
        return;
    }
    ModelColumn<T> col;
    col.load(ctm_p, ctm_prof, nzc, n, p);
    double term[kMaxSat];
    for (int k = 0; k < nzs; ++k) {
        const double xi = col.at(log(sat_p[(int64_t)k * n + p]), true);
        const double a = ap_prof[(int64_t)k * n + p];
        double t = a + (xi - a) * ak[(int64_t)k * n + p];
        t = t * pw[(int64_t)k * n + p];
        term[k] = (t != t || t <= 0.0) ? 0.0 : t;                            // <= 0 -> NaN, then np.nansum
    }
    model_xcol[p] = numpy_sum_le128<double>(term, nzs);
}

}  // namespace

extern "C" int oisat_partial_column(oisat_ctx* h, int dtype, const void* deltap, const void* profile, int64_t n, void* out) {
    ARG_CHECK(h && deltap && out && n > 0);
    ARG_CHECK(dtype == OISAT_F32 || dtype == OISAT_F64);
    const int grid = stream_grid(n, 256);
    if (dtype == OISAT_F32) {
        OISAT_LAUNCH(h, "partial_column", (partial_column_kernel<float>), dim3(grid), dim3(256), 0, (const float*)deltap,
                     (const float*)profile, n, (float*)out);
    } else {
        OISAT_LAUNCH(h, "partial_column", (partial_column_kernel<double>), dim3(grid), dim3(256), 0, (const double*)deltap,
                     (const double*)profile, n, (double*)out);
    }
    return OISAT_OK;
}

extern "C" int oisat_amf_recal(oisat_ctx* h, const double* sat_pmid, const double* sat_sw, int nzs, int ctm_dtype,
                               const void* ctm_pmid, const void* ctm_partial, int nzc, const double* tropopause, const double* vcd,
                               const double* amf, int64_t n, double* new_amf, double* vcd_out, double* ctm_vcd) {
    ARG_CHECK(h && sat_pmid && sat_sw && ctm_pmid && ctm_partial && vcd && amf && new_amf && vcd_out && ctm_vcd && n > 0);
    ARG_CHECK(nzs >= 2 && nzs <= kMaxSat && nzc >= 1 && nzc <= kMaxCtm);
    ARG_CHECK(ctm_dtype == OISAT_F32 || ctm_dtype == OISAT_F64);
    if (ctm_dtype == OISAT_F32) {
        OISAT_LAUNCH(h, "amf_recal", (amf_recal_kernel<float>), dim3((unsigned)cdiv(n, 128)), dim3(128), 0, sat_pmid, sat_sw, nzs,
                     (const float*)ctm_pmid, (const float*)ctm_partial, nzc, tropopause, vcd, amf, n, new_amf, vcd_out, ctm_vcd);
    } else {
        OISAT_LAUNCH(h, "amf_recal", (amf_recal_kernel<double>), dim3((unsigned)cdiv(n, 128)), dim3(128), 0, sat_pmid, sat_sw, nzs,
                     (const double*)ctm_pmid, (const double*)ctm_partial, nzc, tropopause, vcd, amf, n, new_amf, vcd_out, ctm_vcd);
    }
    return OISAT_OK;
}

extern "C" int oisat_column_sum(oisat_ctx* h, int dtype, const void* ctm_pmid, const void* ctm_partial, int nzc,
                                const double* tropopause, const double* vcd, int64_t n, void* ctm_vcd) {
    ARG_CHECK(h && ctm_pmid && ctm_partial && vcd && ctm_vcd && n > 0 && nzc >= 1);
    ARG_CHECK(dtype == OISAT_F32 || dtype == OISAT_F64);
    if (dtype == OISAT_F32) {
        OISAT_LAUNCH(h, "column_sum", (column_sum_kernel<float>), dim3(stream_grid(n, 256)), dim3(256), 0, (const float*)ctm_pmid,
                     (const float*)ctm_partial, nzc, tropopause, vcd, n, (float*)ctm_vcd);
    } else {
        OISAT_LAUNCH(h, "column_sum", (column_sum_kernel<double>), dim3(stream_grid(n, 256)), dim3(256), 0, (const double*)ctm_pmid,
                     (const double*)ctm_partial, nzc, tropopause, vcd, n, (double*)ctm_vcd);
    }
    return OISAT_OK;
}

extern "C" int oisat_ak_conv_mopitt(oisat_ctx* h, int ctm_dtype, const void* ctm_pmid, const void* ctm_profile,
                                    const void* ctm_air_partial, int nzc, const double* sat_pmid, const double* averaging_kernels,
                                    const double* apriori_profile, int nzs, const double* aprior_column,
                                    const double* apriori_surface, const double* vcd, int64_t n, double* model_vcd,
                                    double* model_xcol) {
    ARG_CHECK(h && ctm_pmid && ctm_profile && ctm_air_partial && sat_pmid && averaging_kernels && apriori_profile);
    ARG_CHECK(aprior_column && apriori_surface && vcd && model_vcd && model_xcol && n > 0);
    ARG_CHECK(nzs >= 1 && nzs <= kMaxSat && nzc >= 2 && nzc <= kMaxCtm);
    ARG_CHECK(ctm_dtype == OISAT_F32 || ctm_dtype == OISAT_F64);
    const dim3 grid((unsigned)cdiv(n, 128)), block(128);
    if (ctm_dtype == OISAT_F32) {
        OISAT_LAUNCH(h, "ak_conv_mopitt", (ak_conv_mopitt_kernel<float>), grid, block, 0, (const float*)ctm_pmid,
                     (const float*)ctm_profile, (const float*)ctm_air_partial, nzc, sat_pmid, averaging_kernels, apriori_profile, nzs,
                     aprior_column, apriori_surface, vcd, n, model_vcd, model_xcol);
    } else {
        OISAT_LAUNCH(h, "ak_conv_mopitt", (ak_conv_mopitt_kernel<double>), grid, block, 0, (const double*)ctm_pmid,
                     (const double*)ctm_profile, (const double*)ctm_air_partial, nzc, sat_pmid, averaging_kernels, apriori_profile,
                     nzs, aprior_column, apriori_surface, vcd, n, model_vcd, model_xcol);
    }
    return OISAT_OK;
}

extern "C" int oisat_ak_conv_gosat(oisat_ctx* h, int ctm_dtype, const void* ctm_pmid, const void* ctm_profile, int nzc,
                                   const double* sat_pmid, const double* averaging_kernels, const double* apriori_profile,
                                   const double* pressure_weight, int nzs, const double* x_col, int64_t n, double* model_xcol) {
    ARG_CHECK(h && ctm_pmid && ctm_profile && sat_pmid && averaging_kernels && apriori_profile && pressure_weight && x_col);
    ARG_CHECK(model_xcol && n > 0 && nzs >= 1 && nzs <= kMaxSat && nzc >= 2 && nzc <= kMaxCtm);
    ARG_CHECK(ctm_dtype == OISAT_F32 || ctm_dtype == OISAT_F64);
    const dim3 grid((unsigned)cdiv(n, 128)), block(128);
    if (ctm_dtype == OISAT_F32) {
        OISAT_LAUNCH(h, "ak_conv_gosat", (ak_conv_gosat_kernel<float>), grid, block, 0, (const float*)ctm_pmid,
                     (const float*)ctm_profile, nzc, sat_pmid, averaging_kernels, apriori_profile, pressure_weight, nzs, x_col, n,
                     model_xcol);
    } else {
        OISAT_LAUNCH(h, "ak_conv_gosat", (ak_conv_gosat_kernel<double>), grid, block, 0, (const double*)ctm_pmid,
                     (const double*)ctm_profile, nzc, sat_pmid, averaging_kernels, apriori_profile, pressure_weight, nzs, x_col, n,
                     model_xcol);
    }
    return OISAT_OK;
}

extern "C" int oisat_water_column(oisat_ctx* h, int dtype, const void* deltap, const void* profile, int64_t n, void* out) {
    ARG_CHECK(h && deltap && profile && out && n > 0);
    ARG_CHECK(dtype == OISAT_F32 || dtype == OISAT_F64);
    const int grid = stream_grid(n, 256);
    if (dtype == OISAT_F32) {
        OISAT_LAUNCH(h, "water_column", (water_column_kernel<float>), dim3(grid), dim3(256), 0, (const float*)deltap,
                     (const float*)profile, n, (float*)out);
    } else {
        OISAT_LAUNCH(h, "water_column", (water_column_kernel<double>), dim3(grid), dim3(256), 0, (const double*)deltap,
                     (const double*)profile, n, (double*)out);
    }
    return OISAT_OK;
}

extern "C" int oisat_pwv_sum(oisat_ctx* h, int dtype, const void* partial, int nz, const double* vcd, int64_t n, void* out) {
    ARG_CHECK(h && partial && vcd && out && n > 0 && nz >= 1);
    ARG_CHECK(dtype == OISAT_F32 || dtype == OISAT_F64);
    const int grid = stream_grid(n, 256);
    if (dtype == OISAT_F32) {
        OISAT_LAUNCH(h, "pwv_sum", (pwv_sum_kernel<float>), dim3(grid), dim3(256), 0, (const float*)partial, nz, vcd, n, (float*)out);
    } else {
        OISAT_LAUNCH(h, "pwv_sum", (pwv_sum_kernel<double>), dim3(grid), dim3(256), 0, (const double*)partial, nz, vcd, n, (double*)out);
    }
    return OISAT_OK;
}

// Dense Gaussian background-error covariance pieces of the north-star analysis
//     x_a = x_b + B H^T (H B H^T + R)^-1 (y - H x_b),   B = D^1/2 C D^1/2,
//     C(p,q) = exp(-g |p-q|^2),  g = R_earth^2 / (2 L^2),  p,q unit vectors on the sphere.
// The reference has no counterpart (its OI is the L->0, H=I limit: optimal_interpolation.py:27,
// :49-50).  Chord distance keeps C positive definite on the sphere; |p-q|^2 is formed from
// coordinate DIFFERENCES (not 2-2p.q) so that fp32 keeps ~1e-7 relative accuracy at small range.
//
// Nothing n x m is ever materialised: B H^T (415 GB at n=1,036,800, m=1e5) is generated tile by
// tile in registers from 3 floats per point, so these kernels are VALU/transcendental-bound
// (v_exp_f32 is quarter rate), not HBM-bound -- except cov_build, which writes S once (4 m^2 B).
#include "oisat_common.h"

#include <algorithm>

namespace {

// ---- S = sig sig^T .* C + diag(var), lower 64x64 tiles (diagonal tiles complete) ----------------
__global__ __launch_bounds__(256) void cov_build_kernel(const double* __restrict__ oxyz, const double* __restrict__ osig,
                                                         const double* __restrict__ ovar, int64_t m, int64_t mp, float g2,
                                                         float* __restrict__ S, int64_t ld, int ntile) {
    // triangular tile index -> (ti >= tj)
    const int64_t b = blockIdx.x;
    int ti = (int)((sqrt(8.0 * (double)b + 1.0) - 1.0) * 0.5);
    while ((int64_t)ti * (ti + 1) / 2 > b) --ti;
    while ((int64_t)(ti + 1) * (ti + 2) / 2 <= b) ++ti;
    const int tj = (int)(b - (int64_t)ti * (ti + 1) / 2);
    __shared__ float4 pa[64], pb[64];            // x, y, z, sig
    const int t = threadIdx.x;
    if (t < 128) {
        const int64_t row = (t < 64 ? (int64_t)ti * 64 + t : (int64_t)tj * 64 + (t - 64));
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row < m) v = make_float4((float)oxyz[row], (float)oxyz[m + row], (float)oxyz[2 * m + row], (float)osig[row]);
        if (t < 64) pa[t] = v; else pb[t - 64] = v;
    }
    __syncthreads();
    const int cx = (t & 15) * 4;                 // 4 consecutive columns -> one 16-byte store
    const int ry = t >> 4;                       // rows ry, ry+16, ry+32, ry+48
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const int r = ry + rr * 16;
        const int64_t grow = (int64_t)ti * 64 + r;
        if (grow >= mp) continue;
        const float4 a = pa[r];
        float o[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float4 q = pb[cx + c];
            const int64_t gcol = (int64_t)tj * 64 + cx + c;
            const float dx = a.x - q.x, dy = a.y - q.y, dz = a.z - q.z;
            const float d2 = dx * dx + dy * dy + dz * dz;
            float v = a.w * q.w * __builtin_amdgcn_exp2f(-g2 * d2);
            if (grow == gcol) v = grow < m ? (float)(osig[grow] * osig[grow] + ovar[grow]) : 1.0f;     // padding: identity
            else if (grow >= m || gcol >= m) v = 0.0f;
            o[c] = v;
        }
        *reinterpret_cast<float4*>(&S[grow * ld + (int64_t)tj * 64 + cx]) = make_float4(o[0], o[1], o[2], o[3]);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void innovation_kernel(const T* __restrict__ xb, const int64_t* __restrict__ cell,
                                                          const double* __restrict__ y, int64_t m, double* __restrict__ d) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t a = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; a < m; a += stride) d[a] = y[a] - (double)xb[cell[a]];
}

#include "dense_solve_dev.inc"

// LDS of the solve-phase kernels below: one carve-out per workgroup (dense_solve_dev.inc)
#define SOLVE_LDS(W)                                                                    \
    __shared__ __attribute__((aligned(16))) char solve_lds_raw[kSolveLdsBytes];         \
    const SolveLds W = solve_lds_carve(solve_lds_raw)

// ---- r = d - S z in double (dense_solve_dev.inc: resid_rows_block); blockIdx.x = block of 64 rows ---------------------
template <bool BATCH>
__global__ __launch_bounds__(256) void cov_residual_kernel(const double* __restrict__ oxyz, const double* __restrict__ osig,
                                                            const double* __restrict__ ovar, int64_t m, double g,
                                                            const double* __restrict__ d, const double* __restrict__ z,
                                                            double* __restrict__ r, const double* __restrict__ olat, double win_deg,
                                                            const int* __restrict__ converged, const SolveMember* __restrict__ mem,
                                                            double* __restrict__ partial, int nsplit) {
    // partial != nullptr (single system with fewer than two blocks of rows per CU): blockIdx.y = one of nsplit slices of the
    // column range; the slice's sums go to partial[y][row] and resid_combine_kernel adds them in slice order
    if (BATCH) {                                                // batched: blockIdx.y = member, r = its padded right-hand side
        const SolveMember* mb = mem + blockIdx.y;
        m = mb->m;
        if ((int64_t)blockIdx.x * 64 >= m) return;
        oxyz = mb->oxyz;
        osig = mb->osig;
        ovar = mb->ovar;
        d = mb->d;
        z = mb->z;
        r = mb->rhs;
        if (olat != nullptr) olat = mb->olat;                   // (olat non-null = "use the latitude window")
        converged = &mb->st->conv;
    }
    SOLVE_LDS(W);
    if (converged != nullptr && *converged != 0) return;       // the refinement has met its tolerance: nothing left to evaluate
    resid_rows_block<false>(oxyz, osig, ovar, m, g, d, z, r, olat, win_deg, BATCH ? (double*)nullptr : partial, nsplit, (int)blockIdx.y,
                            (int64_t)blockIdx.x, W);
}

// r = d - (sig * sum of the column slices' sums, in slice order, + var * z): second half of the sliced residual
__global__ __launch_bounds__(256) void resid_combine_kernel(const double* __restrict__ partial, int nsplit, int64_t m,
                                                             const double* __restrict__ osig, const double* __restrict__ ovar,
                                                             const double* __restrict__ d, const double* __restrict__ z,
                                                             double* __restrict__ r, const int* __restrict__ converged) {
    if (converged != nullptr && *converged != 0) return;
    const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (row >= m) return;
    double s = 0.0;
    for (int y = 0; y < nsplit; ++y) s += partial[(int64_t)y * m + row];
    r[row] = d[row] - (osig[row] * s + ovar[row] * z[row]);
}

// ---- the same residual on COMPACT blocks of rows (dense_solve_dev.inc: resid_compact_block) ------------------------------
template <bool BATCH>
__global__ __launch_bounds__(256) void cov_residual_blocks_kernel(const double* __restrict__ oxyz, const double* __restrict__ osig,
                                                                   const double* __restrict__ ovar, int64_t m, double g,
                                                                   const double* __restrict__ d, const double* __restrict__ z,
                                                                   double* __restrict__ r, const double* __restrict__ olat, double win_deg,
                                                                   const int* __restrict__ converged, const SolveMember* __restrict__ mem,
                                                                   const int* __restrict__ perm, double cut_chord) {
    if (BATCH) {                                                // batched: blockIdx.y = member, r = its padded right-hand side
        const SolveMember* mb = mem + blockIdx.y;
        m = mb->m;
        if ((int64_t)blockIdx.x * 64 >= m) return;
        oxyz = mb->oxyz;
        osig = mb->osig;
        ovar = mb->ovar;
        d = mb->d;
        z = mb->z;
        r = mb->rhs;
        olat = mb->olat;
        perm = mb->perm;
        converged = &mb->st->conv;
    }
    SOLVE_LDS(W);
    if (converged != nullptr && *converged != 0) return;       // the refinement has met its tolerance: nothing left to evaluate
    resid_compact_block<false>(oxyz, osig, ovar, m, g, d, z, r, olat, win_deg, perm, cut_chord, (int64_t)blockIdx.x, W);
}

// ---- inc_i = sig_i * sum_a C(i,a) w_a ; xa = xb + inc (dense_solve_dev.inc: increment_patch); blockIdx.x = patch / run ----
template <typename T, int CELLS, bool BATCH>
__global__ __launch_bounds__(256) void apply_increment_kernel(const double* __restrict__ gxyz, const double* __restrict__ gsig,
                                                               int64_t n, const double* __restrict__ oxyz,
                                                               const double* __restrict__ osig, const double* __restrict__ z,
                                                               int64_t m, double g2,
                                                               const T* __restrict__ xb, T* __restrict__ xa, T* __restrict__ inc,
                                                               const double* __restrict__ glat, const double* __restrict__ olat,
                                                               double win_deg, const SolveMember* __restrict__ mem, int nx,
                                                               double cut_chord) {
    if (BATCH) {                                                // batched: blockIdx.y = member
        const SolveMember* mb = mem + blockIdx.y;
        n = mb->n;
        if (nx >= 0) nx = mb->nx;                                // (nx < 0: patches switched off)
        gxyz = mb->gxyz;
        gsig = mb->gsig;
        oxyz = mb->oxyz;
        osig = mb->osig;
        z = mb->z;
        m = mb->m;
        xb = (const T*)mb->xb;
        xa = (T*)mb->xa;
        inc = (T*)mb->inc;
        if (glat != nullptr) { glat = mb->glat; olat = mb->olat; }      // (non-null = "use the latitude window")
    }
    SOLVE_LDS(W);
    increment_patch<T, CELLS>(gxyz, gsig, n, oxyz, osig, z, m, g2, xb, xa, inc, glat, olat, win_deg, nx, cut_chord, (int64_t)blockIdx.x, W);
}

}  // namespace

extern "C" int oisat_cov_build(oisat_ctx* h, const double* oxyz, const double* osig, const double* ovar, int64_t m, double g,
                               float* S, int64_t ld) {
    ARG_CHECK(h && oxyz && osig && ovar && S && m > 0 && g >= 0.0);
    const int64_t mp = cdiv(m, 128) * 128;
    ARG_CHECK(ld >= mp && (ld % 4) == 0 && ((uintptr_t)S % 16) == 0);
    const int ntile = (int)(mp / 64);
    const int64_t nblk = (int64_t)ntile * (ntile + 1) / 2;
    ARG_CHECK(nblk < (int64_t)INT32_MAX);
    OISAT_LAUNCH(h, "cov_build", cov_build_kernel, dim3((unsigned)nblk), dim3(256), 0, oxyz, osig, ovar, m, mp,
                 (float)(g * (double)kLog2e), S, ld, ntile);
    return OISAT_OK;
}

extern "C" int oisat_innovation(oisat_ctx* h, int dtype, const void* xb, const int64_t* cell, const double* y, int64_t m,
                                double* d_out) {
    ARG_CHECK(h && xb && cell && y && d_out && m > 0);
    ARG_CHECK(dtype == OISAT_F32 || dtype == OISAT_F64);
    const int grid = stream_grid(m, 256);
    if (dtype == OISAT_F32) {
        OISAT_LAUNCH(h, "innovation", (innovation_kernel<float>), dim3(grid), dim3(256), 0, (const float*)xb, cell, y, m, d_out);
    } else {
        OISAT_LAUNCH(h, "innovation", (innovation_kernel<double>), dim3(grid), dim3(256), 0, (const double*)xb, cell, y, m, d_out);
    }
    return OISAT_OK;
}

// The permutation handed over by oisat_set_obs_blocks is ONE-SHOT: the next gain solve / residual on the handle takes it
// and the handle forgets it, so a buffer the caller frees or re-targets afterwards is never read again (a stale
// pointer matched by m alone was an out-of-bounds gather waiting to happen).
const int* oisat_take_obs_perm(oisat_ctx* h, int64_t m) {
    const int* perm = h->obs_perm_m == m ? h->obs_perm : nullptr;
    h->obs_perm = nullptr;
    h->obs_perm_m = 0;
    return perm;
}

// perm (optional, device): the observations along a space-filling curve (oisat_take_obs_perm) -> compact blocks of rows
int oisat_cov_residual_if(oisat_ctx* h, const double* oxyz, const double* osig, const double* ovar, int64_t m, double g,
                          const double* d, const double* z, double* r_out, const double* olat_sorted, const int* converged_dev,
                          const int* perm) {
    ARG_CHECK(h && oxyz && osig && ovar && d && z && r_out && m > 0);
    const double g2 = g * (double)kLog2e;
    const double win = lat_window_deg(g2);
    if (perm && olat_sorted && residual_blocks_pay(g2)) {
        OISAT_LAUNCH(h, "cov_residual", cov_residual_blocks_kernel<false>, dim3((unsigned)cdiv(m, 64)), dim3(256), 0, oxyz, osig, ovar, m, g,
                     d, z, r_out, olat_sorted, win, converged_dev, (const SolveMember*)nullptr, perm, cut_chord_of(g2));
        return OISAT_OK;
    }
    // fewer than two blocks of 64 rows per CU (m < 32,768 on 256 CUs): the column range is cut into slices, a workgroup per
    // (block of rows, slice), and a second launch adds the slices' sums in order -- 157 workgroups of a 10,000-observation
    // system left 40 % of the CUs empty and the others with one workgroup each: 0.21 -> 0.09 ms per evaluation
    const int64_t blocks = cdiv(m, 64), cus = h->cu_count > 0 ? h->cu_count : 256;
    int nsplit = blocks < 2 * cus ? (int)std::min<int64_t>(8, cdiv(2 * cus, blocks)) : 1;
    double* partial = nullptr;
    if (nsplit > 1) {                                       // behind the solve's two work vectors (oisat_dense_reserve sizes the slot)
        const int64_t mp = cdiv(m, 128) * 128;
        double* w = (double*)oisat_ws(h, 5, sizeof(double) * (2 + 8) * mp);
        if (!w) return OISAT_ENOMEM;
        partial = w + 2 * mp;
    }
    OISAT_LAUNCH(h, "cov_residual", cov_residual_kernel<false>, dim3((unsigned)blocks, (unsigned)nsplit), dim3(256), 0, oxyz, osig, ovar, m,
                 g, d, z, r_out, win < 180.0 ? olat_sorted : (const double*)nullptr, win, converged_dev, (const SolveMember*)nullptr,
                 partial, nsplit);
    if (nsplit > 1) {
        OISAT_LAUNCH(h, "cov_residual", resid_combine_kernel, dim3((unsigned)cdiv(m, 256)), dim3(256), 0, (const double*)partial, nsplit, m,
                     osig, ovar, d, z, r_out, converged_dev);
    }
    return OISAT_OK;
}

extern "C" int oisat_set_obs_blocks(oisat_ctx* h, const int32_t* perm, int64_t m) {
    ARG_CHECK(h != nullptr && m >= 0 && (perm != nullptr || m == 0));
    h->obs_perm = perm;
    h->obs_perm_m = perm ? m : 0;
    return OISAT_OK;
}

extern "C" int oisat_cov_residual(oisat_ctx* h, const double* oxyz, const double* osig, const double* ovar, int64_t m, double g,
                                  const double* d, const double* z, double* r_out, const double* olat_sorted) {
    ARG_CHECK(h != nullptr);
    return oisat_cov_residual_if(h, oxyz, osig, ovar, m, g, d, z, r_out, olat_sorted, nullptr, oisat_take_obs_perm(h, m));
}

template <typename T, int CELLS, bool BATCH>
static int increment_launch(oisat_ctx* h, unsigned gx, unsigned gy, const double* gxyz, const double* gsig, int64_t n, int nx, const double* oxyz,
                            const double* osig, const double* z, int64_t m, double g2, const void* xb, void* xa, void* inc,
                            const double* glat, const double* olat, double win, const SolveMember* mem) {
    OISAT_LAUNCH(h, "apply_increment", (apply_increment_kernel<T, CELLS, BATCH>), dim3(gx, gy), dim3(256), 0, gxyz, gsig, n, oxyz, osig, z,
                 m, g2, (const T*)xb, (T*)xa, (T*)inc, glat, olat, win, mem, nx, cut_chord_of(g2));
    return OISAT_OK;
}

// nx > 0: the n cells are a (n / nx) x nx grid, row-major (oisat_apply_increment_grid); 0: no known shape
static int apply_increment_impl(oisat_ctx* h, int dtype, const double* gxyz, const double* gsig, int64_t n, int64_t nx, const double* oxyz,
                                const double* osig, const double* z, int64_t m, double g, const void* xb, void* xa, void* inc,
                                const double* glat, const double* olat_sorted) {
    ARG_CHECK(h && gxyz && gsig && oxyz && osig && z && n > 0 && m > 0 && (xa || inc));
    ARG_CHECK(!xa || xb);
    ARG_CHECK(dtype == OISAT_F32 || dtype == OISAT_F64);
    ARG_CHECK(nx >= 0 && nx < (int64_t)INT32_MAX && (nx == 0 || n % nx == 0));
    const double g2 = g * (double)kLog2e;
    const double win = lat_window_deg(g2);
    if (!(win < 180.0) || !glat || !olat_sorted) { glat = nullptr; olat_sorted = nullptr; }
    const int cells = increment_cells(h->cu_count, n, 1);
    const int64_t gx = increment_blocks(n, nx, cells);
    ARG_CHECK(gx < (int64_t)INT32_MAX);
    const unsigned ux = (unsigned)gx;
    if (dtype == OISAT_F32)
        return cells == 1 ? increment_launch<float, 1, false>(h, ux, 1, gxyz, gsig, n, (int)nx, oxyz, osig, z, m, g2, xb, xa, inc, glat, olat_sorted, win, nullptr)
                          : increment_launch<float, 2, false>(h, ux, 1, gxyz, gsig, n, (int)nx, oxyz, osig, z, m, g2, xb, xa, inc, glat, olat_sorted, win, nullptr);
    return cells == 1 ? increment_launch<double, 1, false>(h, ux, 1, gxyz, gsig, n, (int)nx, oxyz, osig, z, m, g2, xb, xa, inc, glat, olat_sorted, win, nullptr)
                      : increment_launch<double, 2, false>(h, ux, 1, gxyz, gsig, n, (int)nx, oxyz, osig, z, m, g2, xb, xa, inc, glat, olat_sorted, win, nullptr);
}

extern "C" int oisat_apply_increment(oisat_ctx* h, int dtype, const double* gxyz, const double* gsig, int64_t n,
                                     const double* oxyz, const double* osig, const double* z, int64_t m, double g,
                                     const void* xb, void* xa, void* inc, const double* glat, const double* olat_sorted) {
    return apply_increment_impl(h, dtype, gxyz, gsig, n, 0, oxyz, osig, z, m, g, xb, xa, inc, glat, olat_sorted);
}

extern "C" int oisat_apply_increment_grid(oisat_ctx* h, int dtype, const double* gxyz, const double* gsig, int64_t ny, int64_t nx,
                                          const double* oxyz, const double* osig, const double* z, int64_t m, double g,
                                          const void* xb, void* xa, void* inc, const double* glat, const double* olat_sorted) {
    ARG_CHECK(ny > 0 && nx > 0);
    return apply_increment_impl(h, dtype, gxyz, gsig, ny * nx, nx, oxyz, osig, z, m, g, xb, xa, inc, glat, olat_sorted);
}

// ---- batched forms (oisat_batch_solve, dense_chol.hip): blockIdx.y = member of the device table --------------------------
int oisat_cov_residual_batched(oisat_ctx* h, const SolveMember* mem_dev, const std::vector<SolveMember>& mem_host, int64_t max_m, double g) {
    const double g2 = g * (double)kLog2e;
    const double win = lat_window_deg(g2);
    const int nmem = (int)mem_host.size();
    bool blocks = residual_blocks_pay(g2);
    for (const SolveMember& sm : mem_host) blocks = blocks && sm.perm != nullptr;
    if (blocks) {
        OISAT_LAUNCH(h, "cov_residual", cov_residual_blocks_kernel<true>, dim3((unsigned)cdiv(max_m, 64), (unsigned)nmem), dim3(256), 0,
                     (const double*)nullptr, (const double*)nullptr, (const double*)nullptr, (int64_t)0, g, (const double*)nullptr,
                     (const double*)nullptr, (double*)nullptr, (const double*)nullptr, win, (const int*)nullptr, mem_dev, (const int*)nullptr,
                     cut_chord_of(g2));
        return OISAT_OK;
    }
    static const double dummy = 0.0;                        // non-null marker: "use each member's latitude window"
    OISAT_LAUNCH(h, "cov_residual", cov_residual_kernel<true>, dim3((unsigned)cdiv(max_m, 64), (unsigned)nmem), dim3(256), 0,
                 (const double*)nullptr, (const double*)nullptr, (const double*)nullptr, (int64_t)0, g, (const double*)nullptr,
                 (const double*)nullptr, (double*)nullptr, win < 180.0 ? &dummy : (const double*)nullptr, win, (const int*)nullptr,
                 mem_dev, (double*)nullptr, 1);
    return OISAT_OK;
}

int oisat_apply_increment_batched(oisat_ctx* h, int dtype, const SolveMember* mem_dev, const std::vector<SolveMember>& mem_host, int64_t max_n,
                                  double g) {
    const double g2 = g * (double)kLog2e;
    const double win = lat_window_deg(g2);
    static const double dummy = 0.0;
    const double* use = win < 180.0 ? &dummy : (const double*)nullptr;
    const int nmem = (int)mem_host.size();
    const int cells = increment_cells(h->cu_count, max_n, nmem);
    int64_t gx = 0;                                         // workgroups of the member that needs the most
    for (const SolveMember& sm : mem_host) gx = std::max(gx, increment_blocks(sm.n, sm.nx, cells));
    ARG_CHECK(gx > 0 && gx < (int64_t)INT32_MAX);
    const unsigned ux = (unsigned)gx, gy = (unsigned)nmem;
    if (dtype == OISAT_F32)
        return cells == 1 ? increment_launch<float, 1, true>(h, ux, gy, nullptr, nullptr, max_n, 0, nullptr, nullptr, nullptr, 0, g2, nullptr, nullptr, nullptr, use, use, win, mem_dev)
                          : increment_launch<float, 2, true>(h, ux, gy, nullptr, nullptr, max_n, 0, nullptr, nullptr, nullptr, 0, g2, nullptr, nullptr, nullptr, use, use, win, mem_dev);
    return cells == 1 ? increment_launch<double, 1, true>(h, ux, gy, nullptr, nullptr, max_n, 0, nullptr, nullptr, nullptr, 0, g2, nullptr, nullptr, nullptr, use, use, win, mem_dev)
                      : increment_launch<double, 2, true>(h, ux, gy, nullptr, nullptr, max_n, 0, nullptr, nullptr, nullptr, 0, g2, nullptr, nullptr, nullptr, use, use, win, mem_dev);
}

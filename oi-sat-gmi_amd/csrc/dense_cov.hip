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

constexpr float kLog2e = 1.4426950408889634f;

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

// ---- latitude window ----------------------------------------------------------------------------------
// exp2(-g2 d^2) < 2^-64 once the chord d exceeds sqrt(64/g2): such a pair changes a double sum by less than its
// last bit times the cancellation factor, and two points whose latitudes differ by more than the matching angle are
// at least that far apart.  With the observations sorted by latitude the pairs worth evaluating for a block of rows /
// cells are therefore one contiguous index range, found by two binary searches.  (3.6x fewer pairs at L = 300 km.)
__host__ __device__ inline double lat_window_deg(double g2) {
    const double chord = sqrt(64.0 / g2);
    return chord >= 2.0 ? 1e9 : 2.0 * asin(0.5 * chord) * 57.29577951308232;
}

__device__ __forceinline__ int64_t lower_bound_lat(const double* __restrict__ a, int64_t n, double v) {   // first i with a[i] >= v
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (a[mid] < v) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// ---- culling by distance (round 3) -------------------------------------------------------------------------------------
// The latitude window keeps one contiguous range of the (latitude-ordered) observations per block of cells, but in
// longitude it keeps everything: at L = 300 km and 0.25 deg most of the pairs it leaves are still further apart than the
// chord at which 2^x < 2^-64 (a polar cap's cells see every observation of the cap's band, all around the pole).  The
// increment kernel therefore gives its block of cells -- a compact PATCH of the grid -- a bounding sphere: centre c (normalised
// mean of its points), radius rho (largest chord to c), and, while it stages the candidates of the window into LDS, keeps only
// the observations with |q - c| <= cut + rho (|p - q| >= |q - c| - |p - c|: nothing closer than the cut-off is dropped).  A
// test per (block, observation) instead of an exponential per (cell, observation); no latitudes, longitudes, poles or date
// line in it.  The survivors are compacted in candidate order (wave ballots), so the sums keep a fixed order.
// (The residual kernel's blocks are 64 consecutive observations in LATITUDE order -- all around the globe in longitude -- so a
// sphere around them culls nothing; it keeps the plain window.)
struct BlockSphere { double cx, cy, cz, r2cull; };

// centre and cull radius of the block's live points (px, py, pz per thread and slot; every thread calls this)
template <int SLOTS>
__device__ __forceinline__ BlockSphere block_sphere(const double (&px)[SLOTS], const double (&py)[SLOTS], const double (&pz)[SLOTS],
                                                    const bool (&live)[SLOTS], double cut_chord, double* __restrict__ red /*[16] shared*/) {
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    double sx = 0.0, sy = 0.0, sz = 0.0;
#pragma unroll
    for (int q = 0; q < SLOTS; ++q)
        if (live[q]) { sx += px[q]; sy += py[q]; sz += pz[q]; }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        sx += __shfl_xor(sx, o, kWave);
        sy += __shfl_xor(sy, o, kWave);
        sz += __shfl_xor(sz, o, kWave);
    }
    __syncthreads();
    if (lane == 0) { red[w] = sx; red[4 + w] = sy; red[8 + w] = sz; }
    __syncthreads();
    sx = (red[0] + red[1]) + (red[2] + red[3]);
    sy = (red[4] + red[5]) + (red[6] + red[7]);
    sz = (red[8] + red[9]) + (red[10] + red[11]);
    const double nrm = sqrt(sx * sx + sy * sy + sz * sz);
    BlockSphere b;
    if (!(nrm > 1e-9)) {                                        // points all around the sphere (or none): keep everything
        b.cx = b.cy = b.cz = 0.0;
        b.r2cull = 1e30;
        return b;
    }
    b.cx = sx / nrm; b.cy = sy / nrm; b.cz = sz / nrm;
    double r2 = 0.0;
#pragma unroll
    for (int q = 0; q < SLOTS; ++q)
        if (live[q]) {
            const double dx = px[q] - b.cx, dy = py[q] - b.cy, dz = pz[q] - b.cz;
            r2 = fmax(r2, dx * dx + dy * dy + dz * dz);
        }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) r2 = fmax(r2, __shfl_xor(r2, o, kWave));
    __syncthreads();
    if (lane == 0) red[12 + w] = r2;
    __syncthreads();
    r2 = fmax(fmax(red[12], red[13]), fmax(red[14], red[15]));
    const double rr = cut_chord + sqrt(r2) * (1.0 + 1e-12) + 1e-12;
    b.r2cull = cut_chord < 1e9 ? rr * rr : 1e30;
    return b;
}

constexpr int kStage = 768;                                     // staged observations: processed once 512 have gathered

// one pass of 256 candidates [c0, c0 + 256) /\ [.., j1): the near ones go to buf[fill ..) in candidate order; returns the new fill
__device__ __forceinline__ int stage_near(const double* __restrict__ oxyz, const double* __restrict__ osig, const double* __restrict__ z,
                                          int64_t m, int64_t c0, int64_t j1, const BlockSphere& bs, double2* __restrict__ bxy,
                                          double2* __restrict__ bzw, int fill, int* __restrict__ wcnt /*[4] shared*/) {
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int64_t c = c0 + t;
    double x = 0.0, y = 0.0, zz = 0.0;
    bool near = false;
    if (c < j1) {
        x = oxyz[c]; y = oxyz[m + c]; zz = oxyz[2 * m + c];
        const double dx = x - bs.cx, dy = y - bs.cy, dz = zz - bs.cz;
        near = dx * dx + dy * dy + dz * dz <= bs.r2cull;
    }
    const unsigned long long mask = __ballot(near);
    if (lane == 0) wcnt[w] = __popcll(mask);
    __syncthreads();
    const int c0w = wcnt[0], c1w = wcnt[1], c2w = wcnt[2], c3w = wcnt[3];
    const int base = fill + (w > 0 ? c0w : 0) + (w > 1 ? c1w : 0) + (w > 2 ? c2w : 0);
    if (near) {
        const int pos = base + __popcll(mask & ((1ull << lane) - 1ull));
        bxy[pos] = make_double2(x, y);
        bzw[pos] = make_double2(zz, osig[c] * z[c]);
    }
    __syncthreads();
    return fill + c0w + c1w + c2w + c3w;
}

// ---- r = d - S z in double, S regenerated on the fly (iterative refinement) ----------------------
// Block = 64 rows; 256 threads = 64 rows x 4 column phases; fixed-order combine (deterministic).
// olat (optional): latitudes of the observations in degrees, ASCENDING -- the block's rows then span
// [olat[row0], olat[row_last]] and only columns inside that span +/- the window are visited.
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
    __shared__ double sx[256], sy[256], sz[256], sw[256];      // chunk of 256 columns: coords and sig*z
    __shared__ double part[4][64];
    if (converged != nullptr && *converged != 0) return;       // the refinement has met its tolerance: nothing left to evaluate
    const int t = threadIdx.x;
    const int lr = t & 63, ph = t >> 6;
    const int64_t row = (int64_t)blockIdx.x * 64 + lr;
    const bool live = row < m;
    const double ax = live ? oxyz[row] : 0.0, ay = live ? oxyz[m + row] : 0.0, az = live ? oxyz[2 * m + row] : 0.0;
    int64_t j0 = 0, j1 = m;
    if (olat) {                                                 // block-uniform: every thread runs the same two searches
        const int64_t r0 = (int64_t)blockIdx.x * 64, r1 = (r0 + 63 < m - 1) ? r0 + 63 : m - 1;
        j0 = lower_bound_lat(olat, m, olat[r0] - win_deg);
        j1 = lower_bound_lat(olat, m, olat[r1] + win_deg + 1e-9);
        j0 &= ~(int64_t)255;                                    // keep the chunking aligned: same summation order per row
    }
    if (!BATCH && partial != nullptr) {                         // this slice of [j0, j1), in whole chunks of 256 columns
        const int64_t chunks = (j1 - j0 + 255) >> 8, per = (chunks + nsplit - 1) / nsplit;
        const int64_t a = j0 + (int64_t)blockIdx.y * per * 256, b = a + per * 256;
        j0 = a < j1 ? a : j1;
        j1 = b < j1 ? b : j1;
    }
    double acc = 0.0;
    for (int64_t c0 = j0; c0 < j1; c0 += 256) {
        const int64_t c = c0 + t;
        __syncthreads();
        if (c < m) {
            sx[t] = oxyz[c]; sy[t] = oxyz[m + c]; sz[t] = oxyz[2 * m + c];
            sw[t] = osig[c] * z[c];
        } else {
            sx[t] = 0.0; sy[t] = 0.0; sz[t] = 0.0; sw[t] = 0.0;
        }
        __syncthreads();
        for (int j = ph * 64; j < ph * 64 + 64; ++j) {
            const double dx = ax - sx[j], dy = ay - sy[j], dz = az - sz[j];
            acc += exp(-g * (dx * dx + dy * dy + dz * dz)) * sw[j];
        }
    }
    part[ph][lr] = acc;
    __syncthreads();
    if (ph == 0 && live) {
        const double s = ((part[0][lr] + part[1][lr]) + part[2][lr]) + part[3][lr];
        if (!BATCH && partial != nullptr) partial[(int64_t)blockIdx.y * m + row] = s;
        else r[row] = d[row] - (osig[row] * s + ovar[row] * z[row]);
    }
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

// ---- the same residual on COMPACT blocks of rows (round 3) ------------------------------------------------------------
// The kernel above takes 64 consecutive observations of the latitude order per block: one latitude, every longitude -- its
// window keeps all observations of a latitude band around the globe.  Here `perm` lists the observations along a space-
// filling curve (the host's Morton order of latitude x longitude), so the 64 rows of a block are neighbours in space: the
// block reduces its latitude span (the candidates are still one contiguous range of the latitude order), gets a bounding
// sphere and keeps, while staging the candidates into LDS, only the observations within the covariance's reach of it -- the
// increment's cull (block_sphere / stage_near).  Per row: the same terms in the same (latitude) order, four column phases
// combined in a fixed order; terms below 2^-64 of a term left out.
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
    __shared__ double2 bxy[kStage], bzw[kStage];                // staged near observations: (x, y), (z, sig * z_solve)
    __shared__ double part[4][64];
    __shared__ double red[16];
    __shared__ double s_lo, s_hi;
    __shared__ int wcnt[4];
    if (converged != nullptr && *converged != 0) return;       // the refinement has met its tolerance: nothing left to evaluate
    const int t = threadIdx.x;
    const int lr = t & 63, ph = t >> 6;
    const int64_t pos = (int64_t)blockIdx.x * 64 + lr;
    const bool live = pos < m;
    const int64_t row = live ? (perm ? (int64_t)perm[pos] : pos) : 0;
    const double ax = live ? oxyz[row] : 0.0, ay = live ? oxyz[m + row] : 0.0, az = live ? oxyz[2 * m + row] : 0.0;
    if (ph == 0) {                                              // latitude span of the block's rows (wave 0 holds each row once)
        double lo = live ? olat[row] : 1e9, hi = live ? olat[row] : -1e9;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) { lo = fmin(lo, __shfl_xor(lo, o, kWave)); hi = fmax(hi, __shfl_xor(hi, o, kWave)); }
        if (lr == 0) { s_lo = lo; s_hi = hi; }
    }
    __syncthreads();
    const int64_t j0 = lower_bound_lat(olat, m, s_lo - win_deg);
    const int64_t j1 = lower_bound_lat(olat, m, s_hi + win_deg + 1e-9);
    const double sx1[1] = {ax}, sy1[1] = {ay}, sz1[1] = {az};
    const bool lv1[1] = {live && ph == 0};
    const BlockSphere bs = block_sphere<1>(sx1, sy1, sz1, lv1, cut_chord, red);
    double acc = 0.0;
    int fill = 0;
    for (int64_t c0 = j0; c0 < j1 || fill > 0; c0 += 256) {
        if (c0 < j1) fill = stage_near(oxyz, osig, z, m, c0, j1, bs, bxy, bzw, fill, wcnt);
        if (fill >= 512 || c0 + 256 >= j1) {                    // block-uniform
            for (int j = ph; j < fill; j += 4) {                // column phase ph takes every fourth staged observation
                const double2 oxy = bxy[j];
                const double2 ozw = bzw[j];
                const double dx = ax - oxy.x, dy = ay - oxy.y, dz = az - ozw.x;
                acc += exp(-g * (dx * dx + dy * dy + dz * dz)) * ozw.y;
            }
            fill = 0;
            __syncthreads();
        }
    }
    part[ph][lr] = acc;
    __syncthreads();
    if (ph == 0 && live) {
        const double s = ((part[0][lr] + part[1][lr]) + part[2][lr]) + part[3][lr];
        r[row] = d[row] - (osig[row] * s + ovar[row] * z[row]);
    }
}

// 2^x for x <= 0 in double to 2e-10 relative: x = n + f, |f| <= 1/2, 2^f = e^(f ln 2) by a degree-8 polynomial, 2^n by ldexp
__device__ __forceinline__ double exp2_neg(double x) {
    x = fmax(x, -1100.0);                                                   // (far pairs outside any window: 2^x = 0)
    const double n = __builtin_rint(x);
    const double t = (x - n) * 0.6931471805599453;                          // |t| <= 0.3466
    double p = 2.48015873015873e-05;                                        // 1/8!
    p = __builtin_fma(p, t, 1.984126984126984e-04);
    p = __builtin_fma(p, t, 1.388888888888889e-03);
    p = __builtin_fma(p, t, 8.333333333333333e-03);
    p = __builtin_fma(p, t, 4.1666666666666664e-02);
    p = __builtin_fma(p, t, 1.6666666666666666e-01);
    p = __builtin_fma(p, t, 0.5);
    p = __builtin_fma(p, t, 1.0);
    p = __builtin_fma(p, t, 1.0);
    return __builtin_ldexp(p, (int)n);
}

// ---- inc_i = sig_i * sum_a C(i,a) w_a, w = osig.*z ; xa = xb + inc --------------------------------------------
// Thread = CELLS grid cells, block = 256 threads; observations stream through LDS in chunks and
// are read as wave-uniform (broadcast) 16-byte words.  |p-q|^2 is formed in double: with float
// coordinates the rounding of the inputs alone (3e-8 each) moves the exponent by g*2|p-q|*3e-8 ~ 1e-6
// at the ranges that matter, and the increment -- a sum of ~1e3 such terms of both signs -- was off by
// 1.3e-5 of the field scale at 720x1440 / 1e5 obs.  v_fma_f64 issues at the unpacked fp32 rate on
// CDNA4, so this costs a few extra issue slots per pair, not a factor.  exp2 stays fp32 (v_exp_f32); the
// product with sig*z and the running sum are double, because the terms cancel: sum|term| reaches several
// hundred times the field scale at swath densities, so fp32 partial sums alone cost ~1e-5.
// Cells per block: a 32-wide patch of the (ny x nx) grid (nx > 0: CELLS * 8 rows x 32 columns -- a compact patch has a small
// bounding sphere, 512 consecutive cells of a 1440-wide row span 128 degrees), or CELLS * 256 consecutive cells (nx = 0).
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
    constexpr int PW = 32, PH = CELLS * 8;
    const int t = threadIdx.x;
    int64_t cell[CELLS];
    bool live[CELLS];
    if (nx > 0) {                                               // patch (blockIdx.x) of the ny x nx grid
        const int64_t ny = n / nx;
        const int64_t ppr = (nx + PW - 1) / PW;                 // patches per row of patches
        if ((int64_t)blockIdx.x >= ppr * ((ny + PH - 1) / PH)) return;
        const int64_t py0 = ((int64_t)blockIdx.x / ppr) * PH, px0 = ((int64_t)blockIdx.x % ppr) * PW;
#pragma unroll
        for (int q = 0; q < CELLS; ++q) {
            const int64_t yy = py0 + q * 8 + (t >> 5), xx = px0 + (t & 31);
            live[q] = yy < ny && xx < nx;
            cell[q] = yy * nx + xx;
        }
    } else {
        if ((int64_t)blockIdx.x * CELLS * 256 >= n) return;
#pragma unroll
        for (int q = 0; q < CELLS; ++q) {
            cell[q] = ((int64_t)blockIdx.x * CELLS + q) * 256 + t;
            live[q] = cell[q] < n;
        }
    }
    __shared__ double2 bxy[kStage], bzw[kStage];                // staged near observations: (x, y), (z, sig * z_solve)
    __shared__ double red[16];
    __shared__ double s_lo[4], s_hi[4];
    __shared__ int64_t s_j[2];
    __shared__ int wcnt[4];
    double px[CELLS], py[CELLS], pz[CELLS];
    double acc[CELLS];
#pragma unroll
    for (int q = 0; q < CELLS; ++q) {
        px[q] = live[q] ? gxyz[cell[q]] : 0.0;
        py[q] = live[q] ? gxyz[n + cell[q]] : 0.0;
        pz[q] = live[q] ? gxyz[2 * n + cell[q]] : 0.0;
        acc[q] = 0.0;
    }
    int64_t j0 = 0, j1 = m;
    const bool windowed = glat && olat;
    if (windowed) {                                             // latitude span of this block's cells -> observation index range
        double lo = 1e9, hi = -1e9;
#pragma unroll
        for (int q = 0; q < CELLS; ++q)
            if (live[q]) { const double la = glat[cell[q]]; lo = fmin(lo, la); hi = fmax(hi, la); }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) { lo = fmin(lo, __shfl_xor(lo, o, kWave)); hi = fmax(hi, __shfl_xor(hi, o, kWave)); }
        if ((t & 63) == 0) { s_lo[t >> 6] = lo; s_hi[t >> 6] = hi; }
        __syncthreads();
        if (t == 0) {
            lo = fmin(fmin(s_lo[0], s_lo[1]), fmin(s_lo[2], s_lo[3]));
            hi = fmax(fmax(s_hi[0], s_hi[1]), fmax(s_hi[2], s_hi[3]));
            s_j[0] = lower_bound_lat(olat, m, lo - win_deg);
            s_j[1] = lower_bound_lat(olat, m, hi + win_deg + 1e-9);
        }
        __syncthreads();
        j0 = s_j[0];
        j1 = s_j[1];
    }
    const BlockSphere bs = block_sphere<CELLS>(px, py, pz, live, windowed ? cut_chord : 1e30, red);
    int fill = 0;
    for (int64_t c0 = j0; c0 < j1 || fill > 0; c0 += 256) {
        if (c0 < j1) fill = stage_near(oxyz, osig, z, m, c0, j1, bs, bxy, bzw, fill, wcnt);
        if (fill >= 512 || c0 + 256 >= j1) {                    // block-uniform
#pragma unroll 4
            for (int j = 0; j < fill; ++j) {
                const double2 oxy = bxy[j];
                const double2 ozw = bzw[j];
#pragma unroll
                for (int q = 0; q < CELLS; ++q) {
                    const double dx = px[q] - oxy.x, dy = py[q] - oxy.y, dz = pz[q] - ozw.x;
                    // C = 2^x, x = -g2 |p - q|^2, in DOUBLE (exp2_neg, 2e-10).  Rounds 1-2 used v_exp_f32: good to 1 ulp = 1.2e-7 of
                    // the term, but the terms cancel (sum|term| is several hundred times the field at swath densities, more at
                    // larger L) and that alone put the fields 2.5e-6 of their scale off at 720x1440 / 1e5 obs, L = 300 km, and
                    // 1.3e-5 -- outside the 1e-5 bar -- at 360x720 / 1e4 gridded obs, L = 500 km; with this: 4e-8.  Twice the
                    // instructions per pair (taking only the pairs with C > 2^-8 in double diverges inside the waves and is
                    // slower still at L = 500 km); the bounding-sphere cull above pays for it.
                    acc[q] += exp2_neg(-g2 * (dx * dx + dy * dy + dz * dz)) * ozw.y;
                }
            }
            fill = 0;
            __syncthreads();
        }
    }
#pragma unroll
    for (int q = 0; q < CELLS; ++q) {
        if (live[q]) {
            const double v = gsig[cell[q]] * acc[q];
            if (inc) inc[cell[q]] = (T)v;
            if (xa) xa[cell[q]] = (T)((double)xb[cell[q]] + v);
        }
    }
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

static inline double cut_chord_of(double g2);
// The compact-block form pays where the covariance's reach is small against the domain: measured at 720x1440 / 1e5
// observations, L = 300 km (reach 2 800 km): 6.6 -> 4.6 ms; a month's 50 tiles: 1.69 -> 1.59 ms; at 360x720 / 1e4 observations,
// L = 500 km (reach 4 700 km) the staging costs more than the cull saves: 0.21 -> 0.26 ms.  OISAT_RESID_BLOCKS=0 | 1 forces.
static inline bool residual_blocks_pay(double g2) {
    static const int forced = getenv("OISAT_RESID_BLOCKS") ? atoi(getenv("OISAT_RESID_BLOCKS")) : -1;
    if (forced >= 0) return forced != 0 && lat_window_deg(g2) < 180.0;
    return sqrt(64.0 / g2) <= 0.5;                              // chord on the unit sphere: 3 200 km
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
    static const int forced = getenv("OISAT_RESID_SPLIT") ? atoi(getenv("OISAT_RESID_SPLIT")) : 0;
    if (forced >= 1 && forced <= 8) nsplit = forced;
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

// two cells per thread share every observation fetched from LDS; with too few cells to fill the GPU that way (a polar cap:
// 338 workgroups) one cell per thread doubles the waves in flight instead (OISAT_INC_CELLS = 1 | 2 forces either)
static inline int increment_cells(const oisat_ctx* h, int64_t n, int nmem) {
    static const int forced = getenv("OISAT_INC_CELLS") ? atoi(getenv("OISAT_INC_CELLS")) : 0;
    if (forced == 1 || forced == 2) return forced;
    const int64_t wgs2 = cdiv(n, 512) * nmem;
    return wgs2 < 4 * (int64_t)(h->cu_count > 0 ? h->cu_count : 256) ? 1 : 2;
}

// workgroups of one system: patches of 32 x (8 CELLS) cells of its ny x nx grid (nx > 0) or runs of 256 CELLS cells
static inline bool increment_patches() {                   // OISAT_INC_PATCH=0: runs of consecutive cells whatever the grid (experiments)
    static const bool on = !getenv("OISAT_INC_PATCH") || atoi(getenv("OISAT_INC_PATCH")) != 0;
    return on;
}
static inline int64_t increment_blocks(int64_t n, int64_t nx, int cells) {
    if (nx > 0 && increment_patches()) return cdiv(nx, 32) * cdiv(n / nx, 8 * cells);
    return cdiv(n, 256 * cells);
}

// chord beyond which 2^(-g2 chord^2) < 2^-64 (the same cut-off as the latitude window's)
static inline double cut_chord_of(double g2) { return sqrt(64.0 / g2); }      // (declared above the residual's launcher)

template <typename T, int CELLS, bool BATCH>
static int increment_launch(oisat_ctx* h, unsigned gx, unsigned gy, const double* gxyz, const double* gsig, int64_t n, int nx, const double* oxyz,
                            const double* osig, const double* z, int64_t m, double g2, const void* xb, void* xa, void* inc,
                            const double* glat, const double* olat, double win, const SolveMember* mem) {
    OISAT_LAUNCH(h, "apply_increment", (apply_increment_kernel<T, CELLS, BATCH>), dim3(gx, gy), dim3(256), 0, gxyz, gsig, n, oxyz, osig, z,
                 m, g2, (const T*)xb, (T*)xa, (T*)inc, glat, olat, win, mem, increment_patches() ? nx : -1, cut_chord_of(g2));
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
    const int cells = increment_cells(h, n, 1);
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
    const int cells = increment_cells(h, max_n, nmem);
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

// Kalman-gain solve: in-place blocked Cholesky of S = H B H^T + R (fp32, lower) on the MI355X
// matrix cores, plus the triangular solves and the double-residual refinement around it.
// (North-star extension; the reference's `(Sa*reg+So)**(-1)` is element-wise,
//  optimal_interpolation.py:27.)
//
// Data layout in HBM: S is row-major float, leading dimension ld >= mp = roundup(m,128); rows
// m..mp are identity padding so no kernel needs a bounds check.  L overwrites the lower triangle.
//
// Algorithm: recursive (cache-oblivious) blocked Cholesky.  All O(m^3) work is in ONE kernel,
//     gemm_nt:  C[MxN] (-)= A[MxK] * B[NxK]^T      (both operands K-contiguous, "NT")
// built on v_mfma_f32_32x32x2_f32 (exact fp32, 256 flop/clk/CU = 157 TFLOP/s peak):
//   * 128x128 block tile, 4 waves as 2x2, each wave 64x64 = 2x2 MFMA tiles (64 accumulator VGPRs)
//   * K in steps of 32 through a double-buffered LDS image [128][36] floats per operand: the
//     4-float pad makes every ds_read_b128 fragment read bank-conflict-free (rows 36*r mod 64
//     hit 16 distinct 4-bank slots per 16-lane group)
//   * a lane reads k = 8s+4h..+3 as one ds_read_b128 and feeds 4 consecutive MFMAs; A and B use
//     the same k permutation, so the product is unchanged
//   * global->register prefetch of tile t+1 is issued before the MFMAs of tile t and written to
//     the other LDS buffer after them (one barrier per K-tile)
//   * blockIdx -> tile map keeps each XCD (private 4 MiB L2) on a contiguous strip of tiles that
//     share B rows
// The 128x128 diagonal blocks are factored and inverted by one workgroup in LDS; the panel below
// a diagonal block is then a GEMM with the inverse (TRSM as GEMM, the MAGMA trick), so it also
// runs on MFMA.  The inverses are kept: the triangular solves reuse them.
#include "oisat_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int NB = 128;          // diagonal block / tile edge
constexpr int BK = 32;           // K step
constexpr int LDSW = 36;         // padded LDS row stride in floats (144 B, 16-B aligned)

struct ChFactor {                // what potrs needs besides L: the inverted diagonal blocks
    const float* S = nullptr;
    int64_t m = 0, mp = 0, ld = 0;
    float* tinv = nullptr;       // [mp/NB][2][NB][NB]: Tinv row-major, then Tinv^T row-major
};
static thread_local ChFactor g_factor;   // one factor per host thread (one handle per thread/GPU)

// ---- gemm_nt ---------------------------------------------------------------------------------
// mode 0: C -= A*B^T      mode 1: C = A*B^T (C may alias A when N == K == 128: TRSM-as-GEMM)
// lower != 0: the C region is anchored on the diagonal; tiles strictly above it are skipped.
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(float* C, int64_t ldc, const float* A,
                                                          int64_t lda, const float* __restrict__ B, int64_t ldb, int ntm,
                                                          int ntn, int K, int mode, int lower, int ntiles_total) {
    __shared__ __attribute__((aligned(16))) float lds[2][2][NB * LDSW];      // [buf][A|B][row*36+k]  = 73,728 B
    // XCD-aware, bijective remap: blocks b, b+8, b+16.. share an XCD -> give each XCD a contiguous strip
    const int nwg = gridDim.x;
    const int orig = blockIdx.x;
    const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
    const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    int ti, tj;
    if (lower) {
        // wg enumerates lower tiles column by column: column tj holds rows tj..ntm-1
        int rem = wg;
        tj = 0;
        // closed form start then fix-up (ntm - tj tiles in column tj)
        {
            const double nn = (double)ntm;
            double t = nn + 0.5 - sqrt((nn + 0.5) * (nn + 0.5) - 2.0 * (double)wg);
            tj = (int)t;
            if (tj < 0) tj = 0;
            if (tj >= ntn) tj = ntn - 1;
            auto col_start = [&](int c) { return (int64_t)c * ntm - (int64_t)c * (c - 1) / 2; };
            while (tj > 0 && col_start(tj) > wg) --tj;
            while (tj + 1 < ntn && col_start(tj + 1) <= wg) ++tj;
            rem = wg - (int)col_start(tj);
        }
        ti = tj + rem;
    } else {
        tj = wg / ntm;
        ti = wg - tj * ntm;
    }
    (void)ntiles_total;
    const int t = threadIdx.x;
    const int lane = t & 63, wid = t >> 6;
    const int wr = wid >> 1, wc = wid & 1;
    const float* Ag = A + (int64_t)ti * NB * lda;
    const float* Bg = B + (int64_t)tj * NB * ldb;
    // staging: pass p covers rows p*32 + (t>>3), 16 bytes at k = (t&7)*4
    const int srow = t >> 3, sk = (t & 7) * 4;
    float4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;     // named registers (arrays behind lambdas went to scratch)
    const float* Ap = Ag + (int64_t)srow * lda + sk;
    const float* Bp = Bg + (int64_t)srow * ldb + sk;
#define OISAT_GLOAD(k0)                                                              \
    do {                                                                             \
        ra0 = *reinterpret_cast<const float4*>(Ap + (k0));                           \
        ra1 = *reinterpret_cast<const float4*>(Ap + 32 * lda + (k0));                \
        ra2 = *reinterpret_cast<const float4*>(Ap + 64 * lda + (k0));                \
        ra3 = *reinterpret_cast<const float4*>(Ap + 96 * lda + (k0));                \
        rb0 = *reinterpret_cast<const float4*>(Bp + (k0));                           \
        rb1 = *reinterpret_cast<const float4*>(Bp + 32 * ldb + (k0));                \
        rb2 = *reinterpret_cast<const float4*>(Bp + 64 * ldb + (k0));                \
        rb3 = *reinterpret_cast<const float4*>(Bp + 96 * ldb + (k0));                \
    } while (0)
#define OISAT_LSTORE(buf)                                                            \
    do {                                                                             \
        float* wa = &lds[buf][0][srow * LDSW + sk];                                  \
        float* wb = &lds[buf][1][srow * LDSW + sk];                                  \
        *reinterpret_cast<float4*>(wa) = ra0;                                        \
        *reinterpret_cast<float4*>(wa + 32 * LDSW) = ra1;                            \
        *reinterpret_cast<float4*>(wa + 64 * LDSW) = ra2;                            \
        *reinterpret_cast<float4*>(wa + 96 * LDSW) = ra3;                            \
        *reinterpret_cast<float4*>(wb) = rb0;                                        \
        *reinterpret_cast<float4*>(wb + 32 * LDSW) = rb1;                            \
        *reinterpret_cast<float4*>(wb + 64 * LDSW) = rb2;                            \
        *reinterpret_cast<float4*>(wb + 96 * LDSW) = rb3;                            \
    } while (0)
    f32x16 acc00 = {0}, acc01 = {0}, acc10 = {0}, acc11 = {0};

    const int nkt = K / BK;
    OISAT_GLOAD(0);
    OISAT_LSTORE(0);
    __syncthreads();
    const int frow = lane & 31, fh = lane >> 5;
    for (int kt = 0; kt < nkt; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nkt) OISAT_GLOAD((kt + 1) * BK);
        const float* la = &lds[cur][0][(wr * 64 + frow) * LDSW + 4 * fh];
        const float* lb = &lds[cur][1][(wc * 64 + frow) * LDSW + 4 * fh];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const float4 a0 = *reinterpret_cast<const float4*>(la + 8 * s);
            const float4 a1 = *reinterpret_cast<const float4*>(la + 32 * LDSW + 8 * s);
            const float4 b0 = *reinterpret_cast<const float4*>(lb + 8 * s);
            const float4 b1 = *reinterpret_cast<const float4*>(lb + 32 * LDSW + 8 * s);
#define OISAT_MFMA4(c)                                                                          \
    acc00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.c, b0.c, acc00, 0, 0, 0);                   \
    acc01 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.c, b1.c, acc01, 0, 0, 0);                   \
    acc10 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.c, b0.c, acc10, 0, 0, 0);                   \
    acc11 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.c, b1.c, acc11, 0, 0, 0);
            OISAT_MFMA4(x) OISAT_MFMA4(y) OISAT_MFMA4(z) OISAT_MFMA4(w)
        }
        if (kt + 1 < nkt) OISAT_LSTORE(cur ^ 1);
        __syncthreads();
    }
    // epilogue: C/D layout col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5)
    float* Cg = C + ((int64_t)ti * NB + wr * 64) * ldc + (int64_t)tj * NB + wc * 64;
#define OISAT_EPI(ACC, i, j)                                                         \
    _Pragma("unroll") for (int e = 0; e < 16; ++e) {                                 \
        const int row = (i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;                  \
        float* p = Cg + (int64_t)row * ldc + (j) * 32 + frow;                        \
        if (mode == 0) *p = *p - ACC[e];                                             \
        else *p = ACC[e];                                                            \
    }
    OISAT_EPI(acc00, 0, 0)
    OISAT_EPI(acc01, 0, 1)
    OISAT_EPI(acc10, 1, 0)
    OISAT_EPI(acc11, 1, 1)
}

// ---- diagonal block: Cholesky + inverse of one 128x128 block, one workgroup, all in LDS ----------
// Left-looking, thread = row: s_ij = a_ij - sum_{c<j} l_ic l_jc.  l_jc is wave-uniform (LDS
// broadcast), l_ic is the thread's own row with an odd row stride (conflict-free).
__global__ __launch_bounds__(128) void potrf_diag_kernel(float* __restrict__ S, int64_t ld, int64_t k0, float* __restrict__ tinv,
                                                          int* __restrict__ info, int block_index) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    constexpr int LD = NB + 1;
    float* a = sm;                     // [128][129]
    float* x = sm + NB * LD;           // [128][129]  inverse, stored transposed: x[c][r] = Tinv[r][c]
    __shared__ float diag_s;
    const int i = threadIdx.x;
    float* Sb = S + k0 * ld + k0;
    for (int c = 0; c < NB; ++c) {     // coalesced: thread = column here
        a[c * LD + i] = (i <= c) ? Sb[(int64_t)c * ld + i] : 0.f;
    }
    __syncthreads();
    for (int j = 0; j < NB; ++j) {
        float s = 0.f;
        if (i >= j) {
            s = a[i * LD + j];
            for (int c = 0; c < j; ++c) s -= a[i * LD + c] * a[j * LD + c];
        }
        if (i == j) {
            if (!(s > 0.f)) {          // non-positive or NaN pivot
                atomicCAS(info, 0, (int)(k0 + j + 1));
                s = 1.f;
            }
            diag_s = sqrtf(s);
        }
        __syncthreads();
        if (i == j) a[i * LD + j] = diag_s;
        else if (i > j) a[i * LD + j] = s / diag_s;
        __syncthreads();
    }
    // inverse: thread = column c of Tinv; x_r = (delta_rc - sum_{k<r} l_rk x_k) / l_rr, r >= c
    {
        const int c = i;
        for (int r = 0; r < NB; ++r) {
            float v = 0.f;
            if (r >= c) {
                v = (r == c) ? 1.f : 0.f;
                for (int k = c; k < r; ++k) v -= a[r * LD + k] * x[c * LD + k];
                v /= a[r * LD + r];
            }
            x[c * LD + r] = v;
        }
    }
    __syncthreads();
    for (int c = 0; c < NB; ++c) {
        const float l = a[c * LD + i];                         // row c, column i
        if (i <= c) Sb[(int64_t)c * ld + i] = l;
        tinv[(((int64_t)block_index * 2 + 0) * NB + c) * NB + i] = x[i * LD + c];     // Tinv[c][i]
        tinv[(((int64_t)block_index * 2 + 1) * NB + c) * NB + i] = x[c * LD + i];     // Tinv^T[c][i] = Tinv[i][c]
    }
}

// identity padding of rows m..mp (columns 0..mp)
__global__ __launch_bounds__(256) void pad_identity_kernel(float* __restrict__ S, int64_t ld, int64_t m, int64_t mp) {
    const int64_t total = (mp - m) * mp;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += stride) {
        const int64_t r = m + p / mp, c = p % mp;
        S[r * ld + c] = r == c ? 1.f : 0.f;
    }
}

// ---- triangular solves (vector right-hand side, double accumulation) --------------------------
// forward step j:  y_j = Tinv_j r_j ; r_b -= L[b,j] y_j for every block row b > j.
// Every workgroup recomputes y_j (64 KB of Tinv from L2) instead of waiting for another launch.
__global__ __launch_bounds__(128) void trsv_fwd_step_kernel(const float* __restrict__ L, int64_t ld, const float* __restrict__ tinv,
                                                             int j, double* __restrict__ r, double* __restrict__ y) {
    __shared__ double rj[NB], yj[NB];
    const int i = threadIdx.x;
    const int b = j + 1 + blockIdx.x;                         // block row to update (blockIdx 0 also stores y_j)
    rj[i] = r[(int64_t)j * NB + i];
    __syncthreads();
    const float* Tt = tinv + ((int64_t)j * 2 + 1) * NB * NB;  // Tinv^T: (Tinv r)_i = sum_c Tt[c][i] r_c, coalesced in i
    double s = 0.0;
    for (int c = 0; c <= i; ++c) s += (double)Tt[c * NB + i] * rj[c];
    yj[i] = s;
    __syncthreads();
    if (blockIdx.x == 0) y[(int64_t)j * NB + i] = s;
    // r_b -= L[b, j] y_j : half-wave per row, lane = 4 consecutive columns (one 512-B row segment per half-wave)
    const int lane = i & 63, wv = i >> 6;
    const int sub = lane & 31, hf = lane >> 5;
    const double y0 = yj[sub * 4], y1 = yj[sub * 4 + 1], y2 = yj[sub * 4 + 2], y3 = yj[sub * 4 + 3];
    for (int it = 0; it < 32; ++it) {
        const int row = wv * 64 + it * 2 + hf;
        const float4 l4 = *reinterpret_cast<const float4*>(L + ((int64_t)b * NB + row) * ld + (int64_t)j * NB + sub * 4);
        double u = ((double)l4.x * y0 + (double)l4.y * y1) + ((double)l4.z * y2 + (double)l4.w * y3);
#pragma unroll
        for (int msk = 16; msk >= 1; msk >>= 1) u += __shfl_xor(u, msk, kWave);
        if (sub == 0) r[(int64_t)b * NB + row] -= u;
    }
}

__global__ __launch_bounds__(128) void trsv_diag_kernel(const float* __restrict__ tinv, int j, const double* __restrict__ r,
                                                         double* __restrict__ y, int transpose) {
    __shared__ double rj[NB];
    const int i = threadIdx.x;
    rj[i] = r[(int64_t)j * NB + i];
    __syncthreads();
    const float* T = tinv + ((int64_t)j * 2 + 0) * NB * NB;
    const float* Tt = tinv + ((int64_t)j * 2 + 1) * NB * NB;
    double s = 0.0;
    if (!transpose) for (int c = 0; c <= i; ++c) s += (double)Tt[c * NB + i] * rj[c];     // Tinv r
    else for (int c = i; c < NB; ++c) s += (double)T[c * NB + i] * rj[c];                 // Tinv^T r
    y[(int64_t)j * NB + i] = s;
}

// backward step j:  z_j = Tinv_j^T y_j ; y_c -= L[j,c]^T z_j for every block column c < j.
__global__ __launch_bounds__(128) void trsv_bwd_step_kernel(const float* __restrict__ L, int64_t ld, const float* __restrict__ tinv,
                                                             int j, double* __restrict__ y, double* __restrict__ z) {
    __shared__ double yj[NB], zj[NB];
    const int i = threadIdx.x;
    const int c = blockIdx.x;                                 // block column to update, c < j
    yj[i] = y[(int64_t)j * NB + i];
    __syncthreads();
    const float* T = tinv + ((int64_t)j * 2 + 0) * NB * NB;
    double s = 0.0;
    for (int k = i; k < NB; ++k) s += (double)T[k * NB + i] * yj[k];      // (Tinv^T y)_i, coalesced across i
    zj[i] = s;
    __syncthreads();
    if (blockIdx.x == 0) z[(int64_t)j * NB + i] = s;
    // y_c[i] -= sum_k L[j*NB+k][c*NB+i] z_j[k]   (coalesced across i)
    const float* Lj = L + (int64_t)j * NB * ld + (int64_t)c * NB + i;
    double u = 0.0;
    for (int k = 0; k < NB; ++k) u += (double)Lj[(int64_t)k * ld] * zj[k];
    y[(int64_t)c * NB + i] -= u;
}

__global__ __launch_bounds__(256) void axpy_kernel(double* __restrict__ z, const double* __restrict__ dz, int64_t m) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += stride) z[i] += dz[i];
}

__global__ __launch_bounds__(256) void copy_pad_kernel(const double* __restrict__ src, int64_t m, int64_t mp, double* __restrict__ dst) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < mp; i += stride) dst[i] = i < m ? src[i] : 0.0;
}

// sum of squares, one block, fixed order (norms for the refinement log)
__global__ __launch_bounds__(1024) void sumsq_kernel(const double* __restrict__ v, int64_t m, double* __restrict__ out) {
    __shared__ double sm[1024];
    double s = 0.0;
    for (int64_t i = threadIdx.x; i < m; i += 1024) s += v[i] * v[i];
    sm[threadIdx.x] = s;
    __syncthreads();
    for (int k = 512; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) sm[threadIdx.x] += sm[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) *out = sm[0];
}

int launch_gemm(oisat_ctx* h, const char* name, float* C, int64_t ldc, const float* A, int64_t lda, const float* B, int64_t ldb,
                int64_t M, int64_t N, int K, int mode, int lower) {
    const int ntm = (int)(M / NB), ntn = (int)(N / NB);
    int64_t ntiles = lower ? (int64_t)ntn * ntm - (int64_t)ntn * (ntn - 1) / 2 : (int64_t)ntm * ntn;
    if (ntiles <= 0) return OISAT_OK;
    OISAT_LAUNCH(h, name, gemm_nt_kernel, dim3((unsigned)ntiles), dim3(256), 0, C, ldc, A, lda, B, ldb, ntm, ntn, K, mode, lower,
                 (int)ntiles);
    return OISAT_OK;
}

// factor block columns [b0, b1) (units of NB) given that everything to their left is applied
int potrf_rec(oisat_ctx* h, float* S, int64_t ld, int64_t mpb, int64_t b0, int64_t b1, float* tinv, int* info_dev) {
    if (b1 - b0 == 1) {
        const int64_t k0 = b0 * NB;
        const size_t shm = sizeof(float) * 2 * NB * (NB + 1);
        OISAT_LAUNCH(h, "potrf_diag", potrf_diag_kernel, dim3(1), dim3(NB), shm, S, ld, k0, tinv, info_dev, (int)b0);
        const int64_t rows = (mpb - b0 - 1) * NB;
        if (rows > 0) {
            float* P = S + (k0 + NB) * ld + k0;                  // panel below the diagonal block
            const int rc = launch_gemm(h, "trsm_gemm", P, ld, P, ld, tinv + b0 * 2 * NB * NB, NB, rows, NB, NB, 1, 0);
            if (rc) return rc;
        }
        return OISAT_OK;
    }
    const int64_t mid = b0 + (b1 - b0 + 1) / 2;
    int rc = potrf_rec(h, S, ld, mpb, b0, mid, tinv, info_dev);
    if (rc) return rc;
    // S[mid:, mid:b1] -= L[mid:, b0:mid] * L[mid:b1, b0:mid]^T   (region anchored on the diagonal)
    {
        float* Cc = S + mid * NB * ld + mid * NB;
        const float* Aa = S + mid * NB * ld + b0 * NB;
        rc = launch_gemm(h, "syrk_gemm", Cc, ld, Aa, ld, Aa, ld, (mpb - mid) * NB, (b1 - mid) * NB, (int)((mid - b0) * NB), 0, 1);
        if (rc) return rc;
    }
    return potrf_rec(h, S, ld, mpb, mid, b1, tinv, info_dev);
}

int trsv_solve(oisat_ctx* h, const ChFactor& f, double* rhs_pad /* mp, overwritten with the solution */, double* tmp) {
    const int nb = (int)(f.mp / NB);
    // forward: L y = rhs   (y -> tmp)
    for (int j = 0; j < nb; ++j) {
        if (j + 1 < nb) {
            OISAT_LAUNCH(h, "trsv_fwd", trsv_fwd_step_kernel, dim3(nb - 1 - j), dim3(NB), 0, f.S, f.ld, (const float*)f.tinv, j,
                         rhs_pad, tmp);
        } else {
            OISAT_LAUNCH(h, "trsv_diag", trsv_diag_kernel, dim3(1), dim3(NB), 0, (const float*)f.tinv, j, (const double*)rhs_pad,
                         tmp, 0);
        }
    }
    // backward: L^T z = y  (z -> rhs_pad)
    for (int j = nb - 1; j >= 0; --j) {
        if (j > 0) {
            OISAT_LAUNCH(h, "trsv_bwd", trsv_bwd_step_kernel, dim3(j), dim3(NB), 0, f.S, f.ld, (const float*)f.tinv, j, tmp,
                         rhs_pad);
        } else {
            OISAT_LAUNCH(h, "trsv_diag", trsv_diag_kernel, dim3(1), dim3(NB), 0, (const float*)f.tinv, j, (const double*)tmp,
                         rhs_pad, 1);
        }
    }
    return OISAT_OK;
}

}  // namespace

extern "C" int oisat_potrf(oisat_ctx* h, float* S, int64_t m, int64_t ld, int* info_host) {
    ARG_CHECK(h && S && m > 0);
    const int64_t mp = cdiv(m, NB) * NB;
    ARG_CHECK(ld >= mp && (ld % 4) == 0 && ((uintptr_t)S % 16) == 0);
    const int64_t mpb = mp / NB;
    float* tinv = (float*)oisat_ws(h, 3, sizeof(float) * mpb * 2 * NB * NB);
    int* info_dev = (int*)oisat_ws(h, 4, 256);
    if (!tinv || !info_dev) return OISAT_ENOMEM;
    static bool attr_set = false;
    if (!attr_set) {
        HIP_TRY(hipFuncSetAttribute((const void*)potrf_diag_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)(sizeof(float) * 2 * NB * (NB + 1))));
        attr_set = true;
    }
    HIP_TRY(hipMemsetAsync(info_dev, 0, sizeof(int), h->stream));
    if (mp > m) {
        OISAT_LAUNCH(h, "pad_identity", pad_identity_kernel, dim3(stream_grid((mp - m) * mp, 256)), dim3(256), 0, S, ld, m, mp);
    }
    const int rc = potrf_rec(h, S, ld, mpb, 0, mpb, tinv, info_dev);
    if (rc) return rc;
    g_factor.S = S;
    g_factor.m = m;
    g_factor.mp = mp;
    g_factor.ld = ld;
    g_factor.tinv = tinv;
    if (info_host) {
        int* pin = (int*)oisat_pinned(h, 64);
        if (!pin) return OISAT_ENOMEM;
        HIP_TRY(hipMemcpyAsync(pin, info_dev, sizeof(int), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
        *info_host = *pin;
        if (*pin != 0) {
            oisat_set_error("potrf: matrix not positive definite at column %d", *pin);
            return OISAT_ENOTPD;
        }
    }
    return OISAT_OK;
}

extern "C" int oisat_potrs(oisat_ctx* h, const float* L, int64_t m, int64_t ld, double* z_inout) {
    ARG_CHECK(h && L && z_inout && m > 0);
    ARG_CHECK(g_factor.S == L && g_factor.m == m && g_factor.ld == ld);     // must follow oisat_potrf of this matrix
    double* w = (double*)oisat_ws(h, 5, sizeof(double) * 2 * g_factor.mp);
    if (!w) return OISAT_ENOMEM;
    double* rhs = w;
    double* tmp = w + g_factor.mp;
    OISAT_LAUNCH(h, "copy_pad", copy_pad_kernel, dim3(stream_grid(g_factor.mp, 256)), dim3(256), 0, (const double*)z_inout, m,
                 g_factor.mp, rhs);
    const int rc = trsv_solve(h, g_factor, rhs, tmp);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(z_inout, rhs, sizeof(double) * m, hipMemcpyDeviceToDevice, h->stream));
    return OISAT_OK;
}

extern "C" int oisat_cov_residual(oisat_ctx* h, const double* oxyz, const double* osig, const double* ovar, int64_t m, double g,
                                  const double* d, const double* z, double* r_out);

extern "C" int oisat_gain_solve(oisat_ctx* h, const float* L, const double* oxyz, const double* osig, const double* ovar, int64_t m,
                                int64_t ld, double g, const double* d, int refine, double* z_out, double* resid_host) {
    ARG_CHECK(h && L && oxyz && osig && ovar && d && z_out && m > 0 && refine >= 0 && refine <= 8);
    ARG_CHECK(g_factor.S == L && g_factor.m == m && g_factor.ld == ld);
    double* r = (double*)oisat_ws(h, 6, sizeof(double) * (m + 16));
    if (!r) return OISAT_ENOMEM;
    double* nrm_dev = r + m;
    double* pin = resid_host ? (double*)oisat_pinned(h, 256) : nullptr;
    if (resid_host && !pin) return OISAT_ENOMEM;
    double dnorm = 1.0;
    if (resid_host) {
        OISAT_LAUNCH(h, "sumsq", sumsq_kernel, dim3(1), dim3(1024), 0, d, m, nrm_dev);
        HIP_TRY(hipMemcpyAsync(pin, nrm_dev, sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
        dnorm = sqrt(pin[0]);
        if (!(dnorm > 0.0)) dnorm = 1.0;
    }
    HIP_TRY(hipMemcpyAsync(z_out, d, sizeof(double) * m, hipMemcpyDeviceToDevice, h->stream));
    int rc = oisat_potrs(h, L, m, ld, z_out);
    if (rc) return rc;
    for (int it = 0; it <= refine; ++it) {
        if (it == refine && !resid_host) break;
        rc = oisat_cov_residual(h, oxyz, osig, ovar, m, g, d, z_out, r);
        if (rc) return rc;
        if (resid_host) {
            OISAT_LAUNCH(h, "sumsq", sumsq_kernel, dim3(1), dim3(1024), 0, (const double*)r, m, nrm_dev);
            HIP_TRY(hipMemcpyAsync(pin, nrm_dev, sizeof(double), hipMemcpyDeviceToHost, h->stream));
            HIP_TRY(hipStreamSynchronize(h->stream));
            resid_host[it] = sqrt(pin[0]) / dnorm;
        }
        if (it == refine) break;
        rc = oisat_potrs(h, L, m, ld, r);
        if (rc) return rc;
        OISAT_LAUNCH(h, "axpy", axpy_kernel, dim3(stream_grid(m, 256)), dim3(256), 0, z_out, (const double*)r, m);
    }
    return OISAT_OK;
}

#include <hip/hip_runtime.h>
#include <cstdint>
#include <cmath>
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int NB = 128, BK = 32, LDSW = 36;
template <int V>
__global__ __launch_bounds__(256, 2) void gemm_var_kernel(float* C, int64_t ldc, const float* A,
                                                          int64_t lda, const float* __restrict__ B, int64_t ldb, int ntm,
                                                          int ntn, int K, int mode, int lower, int ntiles_total) {
    __shared__ __attribute__((aligned(16))) float lds[2][2][NB * LDSW];      // [buf][A|B][row*36+k]  = 73,728 B
    // XCD-aware, bijective remap: blocks b, b+8, b+16.. share an XCD -> give each XCD a contiguous strip
    const int nwg = gridDim.x;
    const int orig = blockIdx.x;
    const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
    const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    // wg -> tile in ROW-BAND order: bands of 8 tile rows, inside a band column by column.  64 consecutive
    // workgroups (what one XCD runs at once: 32 CUs x 2) are then an 8x8 patch of tiles: per K-step they
    // pull 8 A + 8 B tiles through the XCD's L2 instead of 64 + 1 for a column strip.  In `lower` mode only
    // tiles with ti >= tj are enumerated (the last columns of a band are partial).  The band search is a
    // short wave-uniform loop (<= ntm/8 iterations of scalar arithmetic).
    int ti = 0, tj = 0;
    {
        int rem = wg;
        for (int R0 = 0; R0 < ntm; R0 += 8) {
            const int R1 = (R0 + 7 < ntm ? R0 + 7 : ntm - 1), nr = R1 - R0 + 1;
            const int cmax = lower ? (R1 < ntn - 1 ? R1 : ntn - 1) : ntn - 1;       // last column of this band
            const int cfull = lower ? (R0 < cmax ? R0 : cmax) : cmax;               // columns 0..cfull hold all nr rows
            const int tri = cmax - cfull;                                           // partial columns cfull+1..cmax
            const int count = nr * (cfull + 1) + tri * (R1 - cfull + 1) - tri * (tri + 1) / 2;   // + sum_{c} (R1 - c + 1)
            if (rem < count) {
                if (rem < nr * (cfull + 1)) {
                    tj = rem / nr;
                    ti = R0 + rem - tj * nr;
                } else {
                    rem -= nr * (cfull + 1);
                    int c = cfull + 1;
                    while (rem >= R1 - c + 1) { rem -= R1 - c + 1; ++c; }
                    tj = c;
                    ti = c + rem;
                }
                break;
            }
            rem -= count;
        }
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
    const float4 fa0 = *reinterpret_cast<const float4*>(&lds[0][0][(wr * 64 + frow) * LDSW + 4 * fh]);
    const float4 fa1 = *reinterpret_cast<const float4*>(&lds[0][0][(wr * 64 + 32 + frow) * LDSW + 4 * fh]);
    const float4 fb0 = *reinterpret_cast<const float4*>(&lds[0][1][(wc * 64 + frow) * LDSW + 4 * fh]);
    const float4 fb1 = *reinterpret_cast<const float4*>(&lds[0][1][(wc * 64 + 32 + frow) * LDSW + 4 * fh]);
    for (int kt = 0; kt < nkt; ++kt) {
        const int cur = (V >= 2) ? 0 : (kt & 1);
        if (V == 0 && kt + 1 < nkt) OISAT_GLOAD((kt + 1) * BK);
        const float* la = &lds[cur][0][(wr * 64 + frow) * LDSW + 4 * fh];
        const float* lb = &lds[cur][1][(wc * 64 + frow) * LDSW + 4 * fh];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            float4 a0, a1, b0, b1;
            if (V == 3) { a0 = fa0; a1 = fa1; b0 = fb0; b1 = fb1; asm volatile("" : "+v"(a0.x), "+v"(a1.x), "+v"(b0.x), "+v"(b1.x)); }
            else {
            a0 = *reinterpret_cast<const float4*>(la + 8 * s);
            a1 = *reinterpret_cast<const float4*>(la + 32 * LDSW + 8 * s);
            b0 = *reinterpret_cast<const float4*>(lb + 8 * s);
            b1 = *reinterpret_cast<const float4*>(lb + 32 * LDSW + 8 * s); }
#define OISAT_MFMA4(c)                                                                          \
    acc00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.c, b0.c, acc00, 0, 0, 0);                   \
    acc01 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.c, b1.c, acc01, 0, 0, 0);                   \
    acc10 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.c, b0.c, acc10, 0, 0, 0);                   \
    acc11 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.c, b1.c, acc11, 0, 0, 0);
            OISAT_MFMA4(x) OISAT_MFMA4(y) OISAT_MFMA4(z) OISAT_MFMA4(w)
        }
        if (V <= 1) { if (kt + 1 < nkt) OISAT_LSTORE(cur ^ 1);
        __syncthreads(); }
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


#undef OISAT_GLOAD
#undef OISAT_LSTORE
#undef OISAT_MFMA4
#undef OISAT_EPI
__global__ __launch_bounds__(256, 2) void gemm_v4_kernel(float* C, int64_t ldc, const float* A,
                                                          int64_t lda, const float* __restrict__ B, int64_t ldb, int ntm,
                                                          int ntn, int K, int mode, int lower, int ntiles_total) {
    __shared__ __attribute__((aligned(16))) float lds[2][2][NB * LDSW];      // [buf][A|B][row*36+k]  = 73,728 B
    // XCD-aware, bijective remap: blocks b, b+8, b+16.. share an XCD -> give each XCD a contiguous strip
    const int nwg = gridDim.x;
    const int orig = blockIdx.x;
    const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
    const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    // wg -> tile in ROW-BAND order: bands of 8 tile rows, inside a band column by column.  64 consecutive
    // workgroups (what one XCD runs at once: 32 CUs x 2) are then an 8x8 patch of tiles: per K-step they
    // pull 8 A + 8 B tiles through the XCD's L2 instead of 64 + 1 for a column strip.  In `lower` mode only
    // tiles with ti >= tj are enumerated (the last columns of a band are partial).  The band search is a
    // short wave-uniform loop (<= ntm/8 iterations of scalar arithmetic).
    int ti = 0, tj = 0;
    {
        int rem = wg;
        for (int R0 = 0; R0 < ntm; R0 += 8) {
            const int R1 = (R0 + 7 < ntm ? R0 + 7 : ntm - 1), nr = R1 - R0 + 1;
            const int cmax = lower ? (R1 < ntn - 1 ? R1 : ntn - 1) : ntn - 1;       // last column of this band
            const int cfull = lower ? (R0 < cmax ? R0 : cmax) : cmax;               // columns 0..cfull hold all nr rows
            const int tri = cmax - cfull;                                           // partial columns cfull+1..cmax
            const int count = nr * (cfull + 1) + tri * (R1 - cfull + 1) - tri * (tri + 1) / 2;   // + sum_{c} (R1 - c + 1)
            if (rem < count) {
                if (rem < nr * (cfull + 1)) {
                    tj = rem / nr;
                    ti = R0 + rem - tj * nr;
                } else {
                    rem -= nr * (cfull + 1);
                    int c = cfull + 1;
                    while (rem >= R1 - c + 1) { rem -= R1 - c + 1; ++c; }
                    tj = c;
                    ti = c + rem;
                }
                break;
            }
            rem -= count;
        }
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
    const int aoff = (wr * 64 + frow) * LDSW + 4 * fh, boff = (wc * 64 + frow) * LDSW + 4 * fh;
    // fragment registers, two sets: the operands of MFMA group s+1 are read from LDS while group s runs
    float4 fa0, fa1, fb0, fb1, ga0, ga1, gb0, gb1;
#define OISAT_FRAG(A0, A1, B0, B1, buf, s)                                                      \
    do {                                                                                        \
        A0 = *reinterpret_cast<const float4*>(&lds[buf][0][aoff + 8 * (s)]);                    \
        A1 = *reinterpret_cast<const float4*>(&lds[buf][0][aoff + 32 * LDSW + 8 * (s)]);        \
        B0 = *reinterpret_cast<const float4*>(&lds[buf][1][boff + 8 * (s)]);                    \
        B1 = *reinterpret_cast<const float4*>(&lds[buf][1][boff + 32 * LDSW + 8 * (s)]);        \
    } while (0)
#define OISAT_MFMA4(A0, A1, B0, B1, c)                                                          \
    acc00 = __builtin_amdgcn_mfma_f32_32x32x2f32(A0.c, B0.c, acc00, 0, 0, 0);                   \
    acc01 = __builtin_amdgcn_mfma_f32_32x32x2f32(A0.c, B1.c, acc01, 0, 0, 0);                   \
    acc10 = __builtin_amdgcn_mfma_f32_32x32x2f32(A1.c, B0.c, acc10, 0, 0, 0);                   \
    acc11 = __builtin_amdgcn_mfma_f32_32x32x2f32(A1.c, B1.c, acc11, 0, 0, 0);
#define OISAT_MFMA16(A0, A1, B0, B1)                                                            \
    OISAT_MFMA4(A0, A1, B0, B1, x) OISAT_MFMA4(A0, A1, B0, B1, y) OISAT_MFMA4(A0, A1, B0, B1, z) OISAT_MFMA4(A0, A1, B0, B1, w)
    OISAT_FRAG(fa0, fa1, fb0, fb1, 0, 0);
    for (int kt = 0; kt < nkt; ++kt) {
        const int cur = kt & 1;
        const bool more = kt + 1 < nkt;
        if (more) OISAT_GLOAD((kt + 1) * BK);
        OISAT_FRAG(ga0, ga1, gb0, gb1, cur, 1);
        OISAT_MFMA16(fa0, fa1, fb0, fb1)                       // s = 0
        OISAT_FRAG(fa0, fa1, fb0, fb1, cur, 2);
        OISAT_MFMA16(ga0, ga1, gb0, gb1)                       // s = 1
        if (more) OISAT_LSTORE(cur ^ 1);                        // other buffer is free since the last barrier
        OISAT_FRAG(ga0, ga1, gb0, gb1, cur, 3);
        OISAT_MFMA16(fa0, fa1, fb0, fb1)                       // s = 2
        __syncthreads();                                        // tile kt+1 visible; every read of tile kt has been issued
        if (more) OISAT_FRAG(fa0, fa1, fb0, fb1, cur ^ 1, 0);   // first operands of the next tile, behind the last MFMA group
        OISAT_MFMA16(ga0, ga1, gb0, gb1)                       // s = 3
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


#undef OISAT_GLOAD
#undef OISAT_LSTORE
#undef OISAT_MFMA4
#undef OISAT_EPI
#undef OISAT_FRAG
#undef OISAT_MFMA16
__global__ __launch_bounds__(256, 2) void gemm_v5_kernel(float* C, int64_t ldc, const float* A,
                                                          int64_t lda, const float* __restrict__ B, int64_t ldb, int ntm,
                                                          int ntn, int K, int mode, int lower, int ntiles_total) {
    __shared__ __attribute__((aligned(16))) float lds[2][2][NB * LDSW];      // [buf][A|B][row*36+k]  = 73,728 B
    // XCD-aware, bijective remap: blocks b, b+8, b+16.. share an XCD -> give each XCD a contiguous strip
    const int nwg = gridDim.x;
    const int orig = blockIdx.x;
    const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
    const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    // wg -> tile in ROW-BAND order: bands of 8 tile rows, inside a band column by column.  64 consecutive
    // workgroups (what one XCD runs at once: 32 CUs x 2) are then an 8x8 patch of tiles: per K-step they
    // pull 8 A + 8 B tiles through the XCD's L2 instead of 64 + 1 for a column strip.  In `lower` mode only
    // tiles with ti >= tj are enumerated (the last columns of a band are partial).  The band search is a
    // short wave-uniform loop (<= ntm/8 iterations of scalar arithmetic).
    int ti = 0, tj = 0;
    {
        int rem = wg;
        for (int R0 = 0; R0 < ntm; R0 += 8) {
            const int R1 = (R0 + 7 < ntm ? R0 + 7 : ntm - 1), nr = R1 - R0 + 1;
            const int cmax = lower ? (R1 < ntn - 1 ? R1 : ntn - 1) : ntn - 1;       // last column of this band
            const int cfull = lower ? (R0 < cmax ? R0 : cmax) : cmax;               // columns 0..cfull hold all nr rows
            const int tri = cmax - cfull;                                           // partial columns cfull+1..cmax
            const int count = nr * (cfull + 1) + tri * (R1 - cfull + 1) - tri * (tri + 1) / 2;   // + sum_{c} (R1 - c + 1)
            if (rem < count) {
                if (rem < nr * (cfull + 1)) {
                    tj = rem / nr;
                    ti = R0 + rem - tj * nr;
                } else {
                    rem -= nr * (cfull + 1);
                    int c = cfull + 1;
                    while (rem >= R1 - c + 1) { rem -= R1 - c + 1; ++c; }
                    tj = c;
                    ti = c + rem;
                }
                break;
            }
            rem -= count;
        }
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
    const int aoff = (wr * 64 + frow) * LDSW + 4 * fh, boff = (wc * 64 + frow) * LDSW + 4 * fh;
    // fragment registers, two sets: the operands of MFMA group s+1 are read from LDS while group s runs
    float4 fa0, fa1, fb0, fb1, ga0, ga1, gb0, gb1;
#define OISAT_FRAG(A0, A1, B0, B1, buf, s)                                                      \
    do {                                                                                        \
        A0 = *reinterpret_cast<const float4*>(&lds[buf][0][aoff + 8 * (s)]);                    \
        A1 = *reinterpret_cast<const float4*>(&lds[buf][0][aoff + 32 * LDSW + 8 * (s)]);        \
        B0 = *reinterpret_cast<const float4*>(&lds[buf][1][boff + 8 * (s)]);                    \
        B1 = *reinterpret_cast<const float4*>(&lds[buf][1][boff + 32 * LDSW + 8 * (s)]);        \
    } while (0)
#define OISAT_MFMA4(A0, A1, B0, B1, c)                                                          \
    acc00 = __builtin_amdgcn_mfma_f32_32x32x2f32(A0.c, B0.c, acc00, 0, 0, 0);                   \
    acc01 = __builtin_amdgcn_mfma_f32_32x32x2f32(A0.c, B1.c, acc01, 0, 0, 0);                   \
    acc10 = __builtin_amdgcn_mfma_f32_32x32x2f32(A1.c, B0.c, acc10, 0, 0, 0);                   \
    acc11 = __builtin_amdgcn_mfma_f32_32x32x2f32(A1.c, B1.c, acc11, 0, 0, 0);
#define OISAT_MFMA16(A0, A1, B0, B1)                                                            \
    OISAT_MFMA4(A0, A1, B0, B1, x) OISAT_MFMA4(A0, A1, B0, B1, y) OISAT_MFMA4(A0, A1, B0, B1, z) OISAT_MFMA4(A0, A1, B0, B1, w)
    OISAT_FRAG(fa0, fa1, fb0, fb1, 0, 0);
    for (int kt = 0; kt < nkt; ++kt) {
        const int cur = kt & 1;
        const bool more = kt + 1 < nkt;
        if (more) OISAT_GLOAD((kt + 1) * BK);
        OISAT_FRAG(ga0, ga1, gb0, gb1, cur, 1);
        __builtin_amdgcn_sched_barrier(0);
        OISAT_MFMA16(fa0, fa1, fb0, fb1)                       // s = 0
        __builtin_amdgcn_sched_barrier(0);
        OISAT_FRAG(fa0, fa1, fb0, fb1, cur, 2);
        __builtin_amdgcn_sched_barrier(0);
        OISAT_MFMA16(ga0, ga1, gb0, gb1)                       // s = 1
        __builtin_amdgcn_sched_barrier(0);
        if (more) OISAT_LSTORE(cur ^ 1);                        // other buffer is free since the last barrier
        OISAT_FRAG(ga0, ga1, gb0, gb1, cur, 3);
        __builtin_amdgcn_sched_barrier(0);
        OISAT_MFMA16(fa0, fa1, fb0, fb1)                       // s = 2
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();                                        // tile kt+1 visible; every read of tile kt has been issued
        if (more) OISAT_FRAG(fa0, fa1, fb0, fb1, cur ^ 1, 0);   // first operands of the next tile, behind the last MFMA group
        __builtin_amdgcn_sched_barrier(0);
        OISAT_MFMA16(ga0, ga1, gb0, gb1)                       // s = 3
        __builtin_amdgcn_sched_barrier(0);
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


extern "C" int gemm_var(int variant, float* C, int64_t ldc, const float* A, int64_t lda, const float* B, int64_t ldb, int64_t M, int64_t N, int K) {
    const int ntm = (int)(M / NB), ntn = (int)(N / NB);
    const int ntiles = ntm * ntn;
    if (variant == 0) hipLaunchKernelGGL(gemm_var_kernel<0>, dim3(ntiles), dim3(256), 0, 0, C, ldc, A, lda, B, ldb, ntm, ntn, K, 0, 0, ntiles);
    if (variant == 1) hipLaunchKernelGGL(gemm_var_kernel<1>, dim3(ntiles), dim3(256), 0, 0, C, ldc, A, lda, B, ldb, ntm, ntn, K, 0, 0, ntiles);
    if (variant == 5) hipLaunchKernelGGL(gemm_v5_kernel, dim3(ntiles), dim3(256), 0, 0, C, ldc, A, lda, B, ldb, ntm, ntn, K, 0, 0, ntiles);
    if (variant == 4) hipLaunchKernelGGL(gemm_v4_kernel, dim3(ntiles), dim3(256), 0, 0, C, ldc, A, lda, B, ldb, ntm, ntn, K, 0, 0, ntiles);
    if (variant == 3) hipLaunchKernelGGL(gemm_var_kernel<3>, dim3(ntiles), dim3(256), 0, 0, C, ldc, A, lda, B, ldb, ntm, ntn, K, 0, 0, ntiles);
    if (variant == 2) hipLaunchKernelGGL(gemm_var_kernel<2>, dim3(ntiles), dim3(256), 0, 0, C, ldc, A, lda, B, ldb, ntm, ntn, K, 0, 0, ntiles);
    return (int)hipGetLastError();
}
extern "C" int gemm_sync() { return (int)hipDeviceSynchronize(); }

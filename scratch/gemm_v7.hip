// experiment: 128x256 block tile, wave tile 64x128 (2x4 MFMA 32x32), BK=16, skewed pipeline
#include <hip/hip_runtime.h>
#include <cstdint>
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int BM = 128, BN = 128, BK = 16, LW = 20;

__global__ __launch_bounds__(256, 3) void gemm_v6_kernel(float* C, int64_t ldc, const float* A, int64_t lda, const float* __restrict__ B,
                                                          int64_t ldb, int ntm, int ntn, int K) {
    __shared__ __attribute__((aligned(16))) float lds[2][(BM + BN) * LW];     // [buf][A rows | B rows][k]
    const int nwg = gridDim.x, orig = blockIdx.x;
    const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
    const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    // row-band order (8 tile rows per band)
    int ti, tj;
    {
        const int band = wg / (8 * ntn), rem = wg - band * 8 * ntn;
        const int R0 = band * 8, nr = (R0 + 8 <= ntm) ? 8 : ntm - R0;
        tj = rem / nr;
        ti = R0 + rem - tj * nr;
    }
    const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
    const int wr = wid >> 1, wc = wid & 1;   // wave tile 64x64
    const float* Ag = A + (int64_t)ti * BM * lda;
    const float* Bg = B + (int64_t)tj * BN * ldb;
    const int srow = t >> 2, sk = (t & 3) * 4;                 // 64 rows per pass, 4 lanes per row
    const float* Ap = Ag + (int64_t)srow * lda + sk;
    const float* Bp = Bg + (int64_t)srow * ldb + sk;
    float4 ra0, ra1, rb0, rb1;
#define GLOAD(k0)                                                            \
    do {                                                                     \
        ra0 = *reinterpret_cast<const float4*>(Ap + (k0));                   \
        ra1 = *reinterpret_cast<const float4*>(Ap + 64 * lda + (k0));        \
        rb0 = *reinterpret_cast<const float4*>(Bp + (k0));                   \
        rb1 = *reinterpret_cast<const float4*>(Bp + 64 * ldb + (k0));        \
    } while (0)
#define LSTORE(buf)                                                          \
    do {                                                                     \
        float* wa = &lds[buf][srow * LW + sk];                               \
        float* wb = &lds[buf][(BM + srow) * LW + sk];                        \
        *reinterpret_cast<float4*>(wa) = ra0;                                \
        *reinterpret_cast<float4*>(wa + 64 * LW) = ra1;                      \
        *reinterpret_cast<float4*>(wb) = rb0;                                \
        *reinterpret_cast<float4*>(wb + 64 * LW) = rb1;                      \
    } while (0)
    f32x16 c00 = {0}, c01 = {0}, c10 = {0}, c11 = {0};
    const int nkt = K / BK;
    const int frow = lane & 31, fh = lane >> 5;
    const int aoff = (wr * 64 + frow) * LW + 4 * fh, boff = (BM + wc * 64 + frow) * LW + 4 * fh;
    float4 fa0, fa1, fb0, fb1, ga0, ga1, gb0, gb1;
#define FRAG(A0, A1, B0, B1, buf, s)                                                    \
    do {                                                                                        \
        A0 = *reinterpret_cast<const float4*>(&lds[buf][aoff + 8 * (s)]);                       \
        A1 = *reinterpret_cast<const float4*>(&lds[buf][aoff + 32 * LW + 8 * (s)]);             \
        B0 = *reinterpret_cast<const float4*>(&lds[buf][boff + 8 * (s)]);                       \
        B1 = *reinterpret_cast<const float4*>(&lds[buf][boff + 32 * LW + 8 * (s)]);             \
    } while (0)
#define MF(ACC, a, b) ACC = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, ACC, 0, 0, 0);
#define MFMA8(A0, A1, B0, B1, c)                                                        \
    MF(c00, A0.c, B0.c) MF(c01, A0.c, B1.c) MF(c10, A1.c, B0.c) MF(c11, A1.c, B1.c)
#define MFMA32(A0, A1, B0, B1) MFMA8(A0, A1, B0, B1, x) MFMA8(A0, A1, B0, B1, y) MFMA8(A0, A1, B0, B1, z) MFMA8(A0, A1, B0, B1, w)
    GLOAD(0);
    LSTORE(0);
    __syncthreads();
    FRAG(fa0, fa1, fb0, fb1, 0, 0);
    for (int kt = 0; kt < nkt; ++kt) {
        const int cur = kt & 1;
        const bool more = kt + 1 < nkt;
        if (more) GLOAD((kt + 1) * BK);
        FRAG(ga0, ga1, gb0, gb1, cur, 1);
        MFMA32(fa0, fa1, fb0, fb1)
        if (more) LSTORE(cur ^ 1);
        __syncthreads();
        if (more) FRAG(fa0, fa1, fb0, fb1, cur ^ 1, 0);
        MFMA32(ga0, ga1, gb0, gb1)
    }
    float* Cg = C + ((int64_t)ti * BM + wr * 64) * ldc + (int64_t)tj * BN + wc * 64;
#define EPI(ACC, i, j)                                                               \
    _Pragma("unroll") for (int e = 0; e < 16; ++e) {                                 \
        const int row = (i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;                  \
        float* p = Cg + (int64_t)row * ldc + (j) * 32 + frow;                        \
        *p = *p - ACC[e];                                                            \
    }
    EPI(c00, 0, 0) EPI(c01, 0, 1) EPI(c10, 1, 0) EPI(c11, 1, 1)
}

extern "C" int gemm_v6(float* C, int64_t ldc, const float* A, int64_t lda, const float* B, int64_t ldb, int64_t M, int64_t N, int K) {
    const int ntm = (int)(M / BM), ntn = (int)(N / BN);
    hipLaunchKernelGGL(gemm_v6_kernel, dim3(ntm * ntn), dim3(256), 0, 0, C, ldc, A, lda, B, ldb, ntm, ntn, K);
    return (int)hipGetLastError();
}
extern "C" int gemm_sync6() { return (int)hipDeviceSynchronize(); }

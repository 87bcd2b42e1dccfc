// gemm_nt "wide": 128 x 256 block tile, 4 waves as 2x2, each wave 64 x 128 = 2x4 MFMA 32x32 tiles (128 accumulator
// VGPRs), BK = 16, LDS-DMA into a double-buffered swizzled image (48 KB -> 2 workgroups per CU).  Experiment.
#include <hip/hip_runtime.h>
#include <cstdint>
namespace {
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int BM = 128, BN = 256, BK = 16;
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__global__ __launch_bounds__(256, 2) void gemm_v11_kernel(float* C, int64_t ldc, const float* A, int64_t lda,
                                                           const float* __restrict__ B, int64_t ldb, int ntm, int ntn, int K) {
    __shared__ __attribute__((aligned(16))) float ldsA[2][BM * BK];      // 16 KB
    __shared__ __attribute__((aligned(16))) float ldsB[2][BN * BK];      // 32 KB
    const int nwg = gridDim.x, orig = blockIdx.x;
    const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
    const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    // row-band order: bands of 8 tile rows, inside a band column by column
    int ti, tj;
    {
        const int band = wg / (8 * ntn), rem = wg - band * 8 * ntn;
        const int nr = (band * 8 + 8 <= ntm) ? 8 : ntm - band * 8;
        tj = rem / nr;
        ti = band * 8 + rem - tj * nr;
    }
    const int t = threadIdx.x, lane = t & 63, wid = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wr = wid >> 1, wc = wid & 1;
    const float* Ag = A + (int64_t)ti * BM * lda;
    const float* Bg = B + (int64_t)tj * BN * ldb;
    // DMA: one instruction = 16 rows x 64 B.  lane l -> row l>>2, physical chunk l&3 = logical chunk (l&3) ^ ((row>>2)&3)
    const int rl = lane >> 2;
#define V11_DMA(buf, k0)                                                                                            \
    do {                                                                                                            \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) {      /* A: 128 rows = 8 pieces, 2 per wave */               \
            const int row = wid * 32 + 16 * i + rl;                                                                 \
            __builtin_amdgcn_global_load_lds((gptr_t)(Ag + (int64_t)row * lda + 4 * ((lane & 3) ^ ((row >> 2) & 3)) + (k0)), \
                                             (lptr_t)&ldsA[buf][(wid * 32 + 16 * i) * BK], 16, 0, 0);               \
        }                                                                                                           \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {      /* B: 256 rows = 16 pieces, 4 per wave */              \
            const int row = wid * 64 + 16 * i + rl;                                                                 \
            __builtin_amdgcn_global_load_lds((gptr_t)(Bg + (int64_t)row * ldb + 4 * ((lane & 3) ^ ((row >> 2) & 3)) + (k0)), \
                                             (lptr_t)&ldsB[buf][(wid * 64 + 16 * i) * BK], 16, 0, 0);               \
        }                                                                                                           \
    } while (0)
    f32x16 c00 = {0}, c01 = {0}, c02 = {0}, c03 = {0}, c10 = {0}, c11 = {0}, c12 = {0}, c13 = {0};
    const int nkt = K / BK;
    V11_DMA(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int frow = lane & 31, fh = lane >> 5;
    const int key = (frow >> 2) & 3;                       // (row>>2)&3: tile-row offsets are multiples of 32
    const int arow = (wr * 64 + frow) * BK, brow = (wc * 128 + frow) * BK;
    float4 fa0, fa1, fb0, fb1, fb2, fb3, ga0, ga1, gb0, gb1, gb2, gb3;
#define V11_FRAG(A0, A1, B0, B1, B2, B3, buf, s)                                                    \
    do {                                                                                            \
        const int ch = 4 * ((2 * (s) + fh) ^ key);                                                  \
        A0 = *reinterpret_cast<const float4*>(&ldsA[buf][arow + ch]);                               \
        A1 = *reinterpret_cast<const float4*>(&ldsA[buf][arow + 32 * BK + ch]);                     \
        B0 = *reinterpret_cast<const float4*>(&ldsB[buf][brow + ch]);                               \
        B1 = *reinterpret_cast<const float4*>(&ldsB[buf][brow + 32 * BK + ch]);                     \
        B2 = *reinterpret_cast<const float4*>(&ldsB[buf][brow + 64 * BK + ch]);                     \
        B3 = *reinterpret_cast<const float4*>(&ldsB[buf][brow + 96 * BK + ch]);                     \
    } while (0)
#define V11_MFMA8(A0, A1, B0, B1, B2, B3, c)                                                        \
    c00 = __builtin_amdgcn_mfma_f32_32x32x2f32(A0.c, B0.c, c00, 0, 0, 0);                           \
    c01 = __builtin_amdgcn_mfma_f32_32x32x2f32(A0.c, B1.c, c01, 0, 0, 0);                           \
    c02 = __builtin_amdgcn_mfma_f32_32x32x2f32(A0.c, B2.c, c02, 0, 0, 0);                           \
    c03 = __builtin_amdgcn_mfma_f32_32x32x2f32(A0.c, B3.c, c03, 0, 0, 0);                           \
    c10 = __builtin_amdgcn_mfma_f32_32x32x2f32(A1.c, B0.c, c10, 0, 0, 0);                           \
    c11 = __builtin_amdgcn_mfma_f32_32x32x2f32(A1.c, B1.c, c11, 0, 0, 0);                           \
    c12 = __builtin_amdgcn_mfma_f32_32x32x2f32(A1.c, B2.c, c12, 0, 0, 0);                           \
    c13 = __builtin_amdgcn_mfma_f32_32x32x2f32(A1.c, B3.c, c13, 0, 0, 0);
#define V11_MFMA32(A0, A1, B0, B1, B2, B3)                                                          \
    V11_MFMA8(A0, A1, B0, B1, B2, B3, x) V11_MFMA8(A0, A1, B0, B1, B2, B3, y)                       \
    V11_MFMA8(A0, A1, B0, B1, B2, B3, z) V11_MFMA8(A0, A1, B0, B1, B2, B3, w)
    V11_FRAG(fa0, fa1, fb0, fb1, fb2, fb3, 0, 0);
    for (int kt = 0; kt < nkt; ++kt) {
        const int cur = kt & 1;
        const bool more = kt + 1 < nkt;
        if (more) V11_DMA(cur ^ 1, (kt + 1) * BK);
        V11_FRAG(ga0, ga1, gb0, gb1, gb2, gb3, cur, 1);
        V11_MFMA32(fa0, fa1, fb0, fb1, fb2, fb3)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (more) V11_FRAG(fa0, fa1, fb0, fb1, fb2, fb3, cur ^ 1, 0);
        V11_MFMA32(ga0, ga1, gb0, gb1, gb2, gb3)
    }
    float* Cg = C + ((int64_t)ti * BM + wr * 64) * ldc + (int64_t)tj * BN + wc * 128;
#define V11_EPI(ACC, i, j)                                                           \
    _Pragma("unroll") for (int e = 0; e < 16; ++e) {                                 \
        const int row = (i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;                  \
        float* p = Cg + (int64_t)row * ldc + (j) * 32 + frow;                        \
        *p = *p - ACC[e];                                                            \
    }
    V11_EPI(c00, 0, 0) V11_EPI(c01, 0, 1) V11_EPI(c02, 0, 2) V11_EPI(c03, 0, 3)
    V11_EPI(c10, 1, 0) V11_EPI(c11, 1, 1) V11_EPI(c12, 1, 2) V11_EPI(c13, 1, 3)
}
}  // namespace
extern "C" int gemm_v10(float* C, int64_t ldc, const float* A, int64_t lda, const float* B, int64_t ldb, int64_t M, int64_t N, int K, int lower) {
    const int ntm = (int)(M / BM), ntn = (int)(N / BN);
    hipLaunchKernelGGL(gemm_v11_kernel, dim3((unsigned)(ntm * ntn)), dim3(256), 0, 0, C, ldc, A, lda, B, ldb, ntm, ntn, K);
    return (int)hipGetLastError();
}
extern "C" int gemm_sync10() { return (int)hipDeviceSynchronize(); }

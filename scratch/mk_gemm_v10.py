# builds scratch/gemm_v10.hip: the production gemm_nt kernel (LDS-DMA, XOR-swizzled image) with the swizzle key
# f(row) selectable at compile time (-DSWZ=n) to study SQ_LDS_BANK_CONFLICT.
src = open('/root/repo/oi-sat-gmi_amd/csrc/dense_chol.hip').read()
a = src.index('typedef float f32x16')
b = src.index('// ---- gemm_nt for launches that cannot fill the chip')
k = src[a:b]
def rep(old, new):
    global k
    assert old in k, old[:80]
    k = k.replace(old, new, 1)
rep('__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(', '__global__ __launch_bounds__(256, 2) void gemm_v10_kernel(')
rep('    const int rl = lane >> 3, lc = (lane & 7) ^ rl;', '    const int rl = lane >> 3, lc = 0;')
rep('(gptr_t)(Ad + (int64_t)(8 * i) * lda + (k0))', '(gptr_t)(Ad + (int64_t)(8 * i) * lda + 4 * ((lane & 7) ^ SWZKEY(wid * 32 + 8 * i + rl)) + (k0))')
rep('(gptr_t)(Bd + (int64_t)(8 * i) * ldb + (k0))', '(gptr_t)(Bd + (int64_t)(8 * i) * ldb + 4 * ((lane & 7) ^ SWZKEY(wid * 32 + 8 * i + rl)) + (k0))')
rep('    const int sw = frow & 7;', '    const int sw = SWZKEY(frow);')
pre = """
#ifndef SWZ
#define SWZ 0
#endif
#if SWZ == 0
#define SWZKEY(r) ((r) & 7)
#elif SWZ == 1
#define SWZKEY(r) (((r) >> 1) & 7)
#elif SWZ == 2
#define SWZKEY(r) ((((r) & 7) + ((r) >> 3)) & 7)
#elif SWZ == 3
#define SWZKEY(r) (((((r) & 1) << 2) | ((r) & 2) | (((r) >> 2) & 1)))
#elif SWZ == 4
#define SWZKEY(r) (((r) & 3) << 1)
#endif
"""
main = '''
extern "C" int gemm_v10(float* C, int64_t ldc, const float* A, int64_t lda, const float* B, int64_t ldb, int64_t M, int64_t N, int K, int lower) {
    const int ntm = (int)(M / NB), ntn = (int)(N / NB);
    const int64_t ntiles = lower ? (int64_t)ntn * ntm - (int64_t)ntn * (ntn - 1) / 2 : (int64_t)ntm * ntn;
    hipLaunchKernelGGL(gemm_v10_kernel, dim3((unsigned)ntiles), dim3(256), 0, 0, C, ldc, A, lda, B, ldb, ntm, ntn, K, 0, lower, (int)ntiles);
    return (int)hipGetLastError();
}
extern "C" int gemm_sync10() { return (int)hipDeviceSynchronize(); }
'''
open('/root/repo/scratch/gemm_v10.hip', 'w').write('#include <hip/hip_runtime.h>\n#include <cstdint>\n' + pre + 'namespace {\n' + k + '}\n' + main)

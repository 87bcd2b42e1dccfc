# builds scratch/gemm_v9.hip: two of the production gemm_nt pipelines fused into one 512-thread workgroup with a shared
# barrier; half 1 (waves 4-7, the SIMD partners of waves 0-3) places its non-MFMA phases at complementary positions.
# MODE 0: both halves run the production order.  MODE 1: half 1 = [GLOAD][G3'][G0][LSTORE][G1][G2][barrier].
import sys
src = open('/root/repo/oi-sat-gmi_amd/csrc/dense_chol.hip').read()
a = src.index('typedef float f32x16')
b = src.index('// ---- diagonal block: Cholesky + inverse')
k = src[a:b]
def rep(old, new):
    global k
    assert old in k, old[:80]
    k = k.replace(old, new, 1)
rep('__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(', 'template <int MODE>\n__global__ __launch_bounds__(512, 2) void gemm_v9_kernel(')
rep('    __shared__ __attribute__((aligned(16))) float lds[2][2][NB * LDSW];      // [buf][A|B][row*36+k]  = 73,728 B\n',
    '    extern __shared__ __attribute__((aligned(16))) float lds_all[];\n    const int half = threadIdx.x >> 8;\n    float (*lds)[2][NB * LDSW] = reinterpret_cast<float (*)[2][NB * LDSW]>(lds_all + half * (2 * 2 * NB * LDSW));\n')
rep('    const int nwg = gridDim.x;\n    const int orig = blockIdx.x;', '    const int nwg = gridDim.x;\n    const int orig = blockIdx.x;\n    const int npairs_total = ntiles_total;')
rep('    const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);',
    '    const int wgp = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);\n    const int wg = 2 * wgp + half;\n    const bool have_tile = wg < npairs_total;')
rep('    (void)ntiles_total;\n    const int t = threadIdx.x;', '    if (!have_tile) return;            // odd tile count: the last workgroup has one half (s_barrier counts live waves only)\n    const int t = threadIdx.x & 255;')
# main loop: MODE 1 / half 1 variant
old_loop = k[k.index('    for (int kt = 0; kt < nkt; ++kt) {'):k.index('    // epilogue: C/D layout')]
new_loop = '''    if (MODE == 0 || half == 0) {
''' + old_loop.replace('\n    ', '\n        ').rstrip() + '''
    } else {
        // half 1: global prefetch right after the barrier (while half 0 runs its G3), LDS park between G0 and G1
        if (nkt > 1) OISAT_GLOAD(BK);
        for (int kt = 0; kt < nkt; ++kt) {
            const int cur = kt & 1;
            const bool more = kt + 1 < nkt;
            OISAT_FRAG(ga0, ga1, gb0, gb1, cur, 1);
            OISAT_MFMA16(fa0, fa1, fb0, fb1)                       // G0
            if (more) OISAT_LSTORE(cur ^ 1);
            OISAT_FRAG(fa0, fa1, fb0, fb1, cur, 2);
            OISAT_MFMA16(ga0, ga1, gb0, gb1)                       // G1
            OISAT_FRAG(ga0, ga1, gb0, gb1, cur, 3);
            OISAT_MFMA16(fa0, fa1, fb0, fb1)                       // G2
            __syncthreads();
            if (kt + 2 < nkt) OISAT_GLOAD((kt + 2) * BK);
            if (more) OISAT_FRAG(fa0, fa1, fb0, fb1, cur ^ 1, 0);
            OISAT_MFMA16(ga0, ga1, gb0, gb1)                       // G3
        }
    }
'''
k = k.replace(old_loop, new_loop)
main = '''
extern "C" int gemm_v9(int mode, float* C, int64_t ldc, const float* A, int64_t lda, const float* B, int64_t ldb, int64_t M, int64_t N, int K, int lower) {
    const int ntm = (int)(M / NB), ntn = (int)(N / NB);
    const int64_t ntiles = lower ? (int64_t)ntn * ntm - (int64_t)ntn * (ntn - 1) / 2 : (int64_t)ntm * ntn;
    const int nwg = (int)((ntiles + 1) / 2);
    const size_t shm = sizeof(float) * 2 * 2 * 2 * NB * LDSW;
    static bool attr = false;
    if (!attr) {
        hipFuncSetAttribute((const void*)gemm_v9_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
        hipFuncSetAttribute((const void*)gemm_v9_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
        attr = true;
    }
    if (mode == 0) hipLaunchKernelGGL(gemm_v9_kernel<0>, dim3(nwg), dim3(512), shm, 0, C, ldc, A, lda, B, ldb, ntm, ntn, K, 0, lower, (int)ntiles);
    else hipLaunchKernelGGL(gemm_v9_kernel<1>, dim3(nwg), dim3(512), shm, 0, C, ldc, A, lda, B, ldb, ntm, ntn, K, 0, lower, (int)ntiles);
    return (int)hipGetLastError();
}
extern "C" int gemm_sync9() { return (int)hipDeviceSynchronize(); }
'''
open('/root/repo/scratch/gemm_v9.hip', 'w').write('#include <hip/hip_runtime.h>\n#include <cstdint>\nnamespace {\n' + k + '}\n' + main)

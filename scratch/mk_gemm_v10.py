# builds scratch/gemm_v10.hip: gemm_nt with LDS-DMA (global_load_lds_dwordx4) into an XOR-swizzled, unpadded LDS image
# instead of global_load -> VGPR -> ds_write into the padded one.
src = open('/root/repo/oi-sat-gmi_amd/csrc/dense_chol.hip').read()
a = src.index('typedef float f32x16')
b = src.index('// ---- gemm_nt for launches that cannot fill the chip')
k = src[a:b]
def rep(old, new):
    global k
    assert old in k, old[:80]
    k = k.replace(old, new, 1)
rep('__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(', '__global__ __launch_bounds__(256, 2) void gemm_v10_kernel(')
rep('    __shared__ __attribute__((aligned(16))) float lds[2][2][NB * LDSW];      // [buf][A|B][row*36+k]  = 73,728 B\n',
    '    __shared__ __attribute__((aligned(16))) float lds[2][2][NB * 32];        // [buf][A|B][row*32 + 4*(chunk ^ (row&7))] = 65,536 B\n')
# staging -> DMA
s0 = k.index('    // staging: pass p covers rows p*32 + (t>>3), 16 bytes at k = (t&7)*4')
s1 = k.index('    f32x16 acc00 = {0}, acc01 = {0}, acc10 = {0}, acc11 = {0};')
dma = '''    // LDS-DMA: wave `wid` brings rows wid*32 + 8i .. +7 (i = 0..3) of each operand, 1 KiB per instruction; lane l lands
    // at physical chunk (l&7) of row (l>>3) and therefore fetches logical chunk (l&7) ^ (row&7)
    typedef __attribute__((address_space(1))) const void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    const int rl = lane >> 3, lc = (lane & 7) ^ rl;
    const float* Ad = Ag + (int64_t)(wid * 32 + rl) * lda + 4 * lc;
    const float* Bd = Bg + (int64_t)(wid * 32 + rl) * ldb + 4 * lc;
#define OISAT_DMA(buf, k0)                                                                                         \\
    do {                                                                                                           \\
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                            \\
            __builtin_amdgcn_global_load_lds((gptr_t)(Ad + (int64_t)(8 * i) * lda + (k0)), (lptr_t)&lds[buf][0][(wid * 32 + 8 * i) * 32], 16, 0, 0); \\
            __builtin_amdgcn_global_load_lds((gptr_t)(Bd + (int64_t)(8 * i) * ldb + (k0)), (lptr_t)&lds[buf][1][(wid * 32 + 8 * i) * 32], 16, 0, 0); \\
        }                                                                                                          \\
    } while (0)
'''
k = k[:s0] + dma + k[s1:]
rep('    const int t = threadIdx.x;\n    const int lane = t & 63, wid = t >> 6;', '    const int t = threadIdx.x;\n    const int lane = t & 63, wid = __builtin_amdgcn_readfirstlane(t >> 6);')
rep('''    OISAT_GLOAD(0);
    OISAT_LSTORE(0);
    __syncthreads();''', '''    OISAT_DMA(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();''')
rep('    const int aoff = (wr * 64 + frow) * LDSW + 4 * fh, boff = (wc * 64 + frow) * LDSW + 4 * fh;',
    '    const int sw = frow & 7;\n    const int arow = (wr * 64 + frow) * 32, brow = (wc * 64 + frow) * 32;')
rep('''        A0 = *reinterpret_cast<const float4*>(&lds[buf][0][aoff + 8 * (s)]);                    \\
        A1 = *reinterpret_cast<const float4*>(&lds[buf][0][aoff + 32 * LDSW + 8 * (s)]);        \\
        B0 = *reinterpret_cast<const float4*>(&lds[buf][1][boff + 8 * (s)]);                    \\
        B1 = *reinterpret_cast<const float4*>(&lds[buf][1][boff + 32 * LDSW + 8 * (s)]);        \\''',
    '''        A0 = *reinterpret_cast<const float4*>(&lds[buf][0][arow + 4 * ((2 * (s) + fh) ^ sw)]);            \\
        A1 = *reinterpret_cast<const float4*>(&lds[buf][0][arow + 32 * 32 + 4 * ((2 * (s) + fh) ^ sw)]);  \\
        B0 = *reinterpret_cast<const float4*>(&lds[buf][1][brow + 4 * ((2 * (s) + fh) ^ sw)]);            \\
        B1 = *reinterpret_cast<const float4*>(&lds[buf][1][brow + 32 * 32 + 4 * ((2 * (s) + fh) ^ sw)]);  \\''')
rep('        if (more) OISAT_GLOAD((kt + 1) * BK);\n', '        if (more) OISAT_DMA(cur ^ 1, (kt + 1) * BK);            // the other buffer is free since the last barrier\n')
rep('        if (more) OISAT_LSTORE(cur ^ 1);                        // other buffer is free since the last barrier\n', '')
rep('        __syncthreads();                                        // tile kt+1 visible; every read of tile kt has been issued\n',
    '        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave\'s DMA pieces of tile kt+1 have landed\n        __syncthreads();\n')
main = '''
extern "C" int gemm_v10(float* C, int64_t ldc, const float* A, int64_t lda, const float* B, int64_t ldb, int64_t M, int64_t N, int K, int lower) {
    const int ntm = (int)(M / NB), ntn = (int)(N / NB);
    const int64_t ntiles = lower ? (int64_t)ntn * ntm - (int64_t)ntn * (ntn - 1) / 2 : (int64_t)ntm * ntn;
    hipLaunchKernelGGL(gemm_v10_kernel, dim3((unsigned)ntiles), dim3(256), 0, 0, C, ldc, A, lda, B, ldb, ntm, ntn, K, 0, lower, (int)ntiles);
    return (int)hipGetLastError();
}
extern "C" int gemm_sync10() { return (int)hipDeviceSynchronize(); }
'''
open('/root/repo/scratch/gemm_v10.hip', 'w').write('#include <hip/hip_runtime.h>\n#include <cstdint>\nnamespace {\n' + k + '}\n' + main)

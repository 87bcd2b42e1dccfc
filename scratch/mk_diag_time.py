# builds scratch/diag_time.hip: the production potrf_diag kernel text + wall_clock64 stamps per phase + a tiny main()
src = open('/root/repo/oi-sat-gmi_amd/csrc/dense_chol.hip').read()
a = src.index('typedef float f32x4 __attribute__')
b = src.index('// identity padding of rows m..mp')
k = src[a:b]
def rep(old, new):
    global k
    assert old in k, old[:60]
    k = k.replace(old, new)
rep('int* __restrict__ info, int block_index) {', 'int* __restrict__ info, int block_index, long long* stamps) {\n    int sidx = 0;\n#define STAMP() do { if (threadIdx.x == 0) stamps[sidx] = wall_clock64(); ++sidx; } while (0)\n    STAMP();')
rep('    __syncthreads();\n    // Look-ahead: while waves', '    __syncthreads();\n    STAMP();\n    // Look-ahead: while waves')
rep('    if (w == 0) diag16_factor_invert(a, 0, dinv, info, (int)k0, lane);\n    __syncthreads();\n', '    if (w == 0) diag16_factor_invert(a, 0, dinv, info, (int)k0, lane);\n    __syncthreads();\n    STAMP();\n')
rep('        __syncthreads();\n        if (J == 7) break;', '        __syncthreads();\n        STAMP();\n        if (J == 7) break;')
rep('        if (w == 0) diag16_factor_invert(a, j0 + 16, dinv + (J + 1) * DINV_SZ, info, (int)(k0 + j0 + 16), lane);\n        __syncthreads();\n',
    '        long long t0w = wall_clock64();\n        if (w == 0) diag16_factor_invert(a, j0 + 16, dinv + (J + 1) * DINV_SZ, info, (int)(k0 + j0 + 16), lane);\n        if (tid == 0) stamps[40 + J] = wall_clock64() - t0w;\n        __syncthreads();\n        STAMP();\n')
rep('    float* Tg = tinv + (int64_t)block_index * NB * NB;', '    STAMP();\n    float* Tg = tinv + (int64_t)block_index * NB * NB;')
k = k.rstrip()
k = k[:-1] + '    __syncthreads();\n    STAMP();\n    if (threadIdx.x == 0) stamps[39] = sidx;\n}\n'
names = ["", "load", "diag16(0)"] + sum([["panel %d" % J, "trail %d + diag16(%d)" % (J, J + 1)] for J in range(7)], []) + ["panel 7", "T = L^-1", "store"]
main = '''
int main() {
    const int n = 128;
    std::vector<float> A(n * n);
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) { float d = (i - j) * 0.05f; A[i * n + j] = std::exp(-d * d) + (i == j ? 0.3f : 0.f); }
    float *dS, *dT; int* dinfo; long long* dst;
    hipMalloc(&dS, n * n * 4); hipMalloc(&dT, n * n * 4); hipMalloc(&dinfo, 4); hipMalloc(&dst, 64 * 8);
    hipMemset(dinfo, 0, 4);
    const size_t lds = (2 * NB * LDA + 8 * DINV_SZ) * sizeof(float);
    hipFuncSetAttribute((const void*)potrf_diag_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int rep = 0; rep < 3; ++rep) {
        hipMemcpy(dS, A.data(), n * n * 4, hipMemcpyHostToDevice);
        hipMemset(dst, 0, 64 * 8);
        hipLaunchKernelGGL(potrf_diag_kernel, dim3(1), dim3(256), lds, 0, dS, (int64_t)n, (int64_t)0, dT, dinfo, 0, dst);
        hipDeviceSynchronize();
    }
    long long st[64]; hipMemcpy(st, dst, sizeof(st), hipMemcpyDeviceToHost);
    std::vector<float> L(n * n), T(n * n); hipMemcpy(L.data(), dS, n * n * 4, hipMemcpyDeviceToHost); hipMemcpy(T.data(), dT, n * n * 4, hipMemcpyDeviceToHost);
    double e1 = 0, e2 = 0;
    for (int i = 0; i < n; ++i) for (int j = 0; j <= i; ++j) {
        double s = 0, u = 0;
        for (int q = 0; q <= j; ++q) s += (double)L[i * n + q] * L[j * n + q];
        for (int q = j; q <= i; ++q) u += (double)L[i * n + q] * T[q * n + j];
        e1 = std::fmax(e1, std::fabs(s - A[i * n + j])); e2 = std::fmax(e2, std::fabs(u - (i == j)));
    }
    const char* names[] = {NAMES};
    int ns = (int)st[39];
    for (int i = 1; i < ns; ++i) printf("  %-26s %6.2f us (cum %6.2f)\\n", names[i], (st[i] - st[i - 1]) * 0.01, (st[i] - st[0]) * 0.01);
    printf("  diag16 alone on wave 0: %.2f us;  |LL^T - A| %.2e  |L T - I| %.2e\\n", st[41] * 0.01, e1, e2);
    return 0;
}
'''.replace("NAMES", ", ".join('"%s"' % n for n in names))
open('/root/repo/scratch/diag_time.hip', 'w').write('#include <hip/hip_runtime.h>\n#include <cstdio>\n#include <cstdint>\n#include <cmath>\n#include <vector>\nconstexpr int NB = 128;\n' + k + main)

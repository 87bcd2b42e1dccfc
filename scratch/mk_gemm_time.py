# builds scratch/gemm_time.hip: the production gemm_nt kernel + s_memtime stamps around the phases of a K-tile
# (wave 0 of every workgroup accumulates per-phase cycles; the host prints the mean over workgroups)
src = open('/root/repo/oi-sat-gmi_amd/csrc/dense_chol.hip').read()
a = src.index('typedef float f32x16')
b = src.index('// ---- diagonal block: Cholesky + inverse')
k = src[a:b]
def rep(old, new, count=1):
    global k
    assert old in k, old[:70]
    k = k.replace(old, new, count)
rep('int ntn, int K, int mode, int lower, int ntiles_total) {', 'int ntn, int K, int mode, int lower, int ntiles_total, long long* stamps) {\n    long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}; long long tprev;\n#define STAMP(i) do { __builtin_amdgcn_sched_barrier(0); long long tn = clock64(); ph[i] += tn - tprev; tprev = tn; __builtin_amdgcn_sched_barrier(0); } while (0)\n')
rep('    for (int kt = 0; kt < nkt; ++kt) {\n        const int cur = kt & 1;', '    tprev = clock64(); const long long tstart = tprev;\n    for (int kt = 0; kt < nkt; ++kt) {\n        const int cur = kt & 1;')
rep('        OISAT_MFMA16(fa0, fa1, fb0, fb1)                       // s = 0\n', '        STAMP(0);\n        OISAT_MFMA16(fa0, fa1, fb0, fb1)                       // s = 0\n        STAMP(1);\n')
rep('        OISAT_MFMA16(ga0, ga1, gb0, gb1)                       // s = 1\n', '        OISAT_MFMA16(ga0, ga1, gb0, gb1)                       // s = 1\n        STAMP(2);\n')
rep('        if (more) OISAT_LSTORE(cur ^ 1);                        // other buffer is free since the last barrier\n', '        if (more) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); STAMP(3); OISAT_LSTORE(cur ^ 1); }\n')
rep('        OISAT_MFMA16(fa0, fa1, fb0, fb1)                       // s = 2\n', '        OISAT_MFMA16(fa0, fa1, fb0, fb1)                       // s = 2\n        STAMP(4);\n')
rep('        __syncthreads();                                        // tile kt+1 visible; every read of tile kt has been issued\n', '        __syncthreads();\n        STAMP(5);\n')
rep('        OISAT_MFMA16(ga0, ga1, gb0, gb1)                       // s = 3\n    }\n', '        OISAT_MFMA16(ga0, ga1, gb0, gb1)                       // s = 3\n        STAMP(6);\n    }\n    if (t == 0) { for (int i = 0; i < 7; ++i) atomicAdd((unsigned long long*)&stamps[i], (unsigned long long)ph[i]); atomicAdd((unsigned long long*)&stamps[7], (unsigned long long)(clock64() - tstart)); }\n')
names = ["gload issue + frag(1) issue", "MFMA group 0", "frag(2) + MFMA group 1", "wait vmcnt(0) (global tile arrived)", "lstore + frag(3) + MFMA group 2", "barrier", "frag(next,0) + MFMA group 3"]
main = '''
int main(int argc, char** argv) {
    const int64_t M = 8192, N = 8192; const int K = 8192;
    float *A, *B, *C; long long* st;
    hipMalloc(&A, M * K * 4); hipMalloc(&B, N * K * 4); hipMalloc(&C, M * N * 4); hipMalloc(&st, 64);
    std::vector<float> h(M * K); for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f - 0.5f;
    hipMemcpy(A, h.data(), M * K * 4, hipMemcpyHostToDevice); hipMemcpy(B, h.data(), N * K * 4, hipMemcpyHostToDevice); hipMemset(C, 0, M * N * 4);
    const int ntm = M / NB, ntn = N / NB, nt = ntm * ntn;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipMemset(st, 0, 64);
        hipEventRecord(e0);
        hipLaunchKernelGGL(gemm_nt_kernel, dim3(nt), dim3(256), 0, 0, C, N, A, (int64_t)K, B, (int64_t)K, ntm, ntn, K, 0, 0, nt, st);
        hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long s[8]; hipMemcpy(s, st, 64, hipMemcpyDeviceToHost);
        const double per = 1.0 / ((double)nt * (K / BK));
        printf("rep %d: %.3f ms = %.1f TFLOP/s (with stamps); cycles per K-tile per wave (mean over %d workgroups):\\n", rep, ms, 2.0 * M * N * K / ms / 1e9, nt);
        const char* names[] = {NAMES};
        double tot = 0; for (int i = 0; i < 7; ++i) tot += s[i] * per;
        for (int i = 0; i < 7; ++i) printf("   %-40s %8.1f  (%4.1f %%)\\n", names[i], s[i] * per, 100.0 * s[i] * per / tot);
        printf("   total %.1f cycles per K-tile; loop total/wg %.0f\\n", tot, (double)s[7] / nt);
    }
    return 0;
}
'''.replace("NAMES", ", ".join('"%s"' % n for n in names))
open('/root/repo/scratch/gemm_time.hip', 'w').write('#include <hip/hip_runtime.h>\n#include <cstdio>\n#include <cstdint>\n#include <vector>\n' + k + main)

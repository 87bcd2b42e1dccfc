#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include <vector>
constexpr int NB = 128;
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int LDA = 130;
constexpr int DINV_LD = 17;
constexpr int DINV_SZ = 16 * DINV_LD;

__device__ __forceinline__ float rdlane(float v, int l) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}

// Factor and invert in ONE sweep: column k of L is final after pivot step k, and that is exactly when the forward
// substitution for X = L^-1 (lane = column r of X) needs it, so each broadcast L[c][k] = readlane(d[k], c) feeds both
// the trailing update of the factor and the running sums of the inverse.  Half the serial broadcasts of doing the
// two one after the other.
__device__ __forceinline__ void diag16_factor_invert(float* a, int j0, float* dinvJ, int* info, int col0, int lane) {
    const int r = lane & 15;
    float d[16], x[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        d[c] = a[(j0 + r) * LDA + j0 + c];
        x[c] = (c == r) ? 1.f : 0.f;                       // running delta_{c,r} - sum_{k<c} L[c][k] X[k][r]
    }
    int bad = 0;                                           // first non-positive / NaN pivot (wave-uniform), reported once below
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        float piv = rdlane(d[k], k);
        const bool neg = !(piv > 0.f);
        bad = (neg && bad == 0) ? k + 1 : bad;
        piv = neg ? 1.f : piv;
        float ri = __builtin_amdgcn_rsqf(piv);
        ri = ri * (1.5f - 0.5f * piv * ri * ri);          // one Newton step: ~0.5 ulp
        d[k] = (r == k) ? piv * ri : d[k] * ri;
        x[k] = (k >= r) ? x[k] * ri : 0.f;
#pragma unroll
        for (int c = k + 1; c < 16; ++c) {
            const float l = rdlane(d[k], c);              // L[c][k]
            d[c] -= d[k] * l;
            x[c] -= l * x[k];
        }
    }
    if (bad && lane == 0) atomicCAS(info, 0, col0 + bad);
#pragma unroll
    for (int c = 0; c < 16; ++c)
        if (lane < 16) a[(j0 + r) * LDA + j0 + c] = (c <= r) ? d[c] : 0.f;
#pragma unroll
    for (int rr = 0; rr < 16; ++rr)
        if (lane < 16) dinvJ[rr * DINV_LD + r] = x[rr];
}

__global__ __launch_bounds__(256) void potrf_diag_kernel(float* __restrict__ S, int64_t ld, int64_t k0, float* __restrict__ tinv,
                                                          int* __restrict__ info, int block_index, long long* stamps) {
    int sidx = 0;
#define STAMP() do { if (threadIdx.x == 0) stamps[sidx] = wall_clock64(); ++sidx; } while (0)
    STAMP();
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* a = sm;                         // [128][LDA]
    float* t = sm + NB * LDA;              // [128][LDA]
    float* dinv = t + NB * LDA;            // [8][16][17]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int lr = lane & 15, lg = lane >> 4;
    float* Sb = S + k0 * ld + k0;
    {   // 16 independent 16-byte loads per thread, issued together (row = (tid>>5)+8p, 4 columns at (tid&31)*4).
        // Row stride 130 floats keeps (r, c) with c % 4 == 0 8-byte aligned: two ds_write_b64 per quad.
        // (t needs no clearing: every element of it that is read later has been written by then.)
        float4 v[16];
#pragma unroll
        for (int p = 0; p < 16; ++p) v[p] = *reinterpret_cast<const float4*>(Sb + (int64_t)((tid >> 5) + 8 * p) * ld + (tid & 31) * 4);
#pragma unroll
        for (int p = 0; p < 16; ++p) {
            const int r = (tid >> 5) + 8 * p, c = (tid & 31) * 4;
            float2* q = reinterpret_cast<float2*>(a + r * LDA + c);
            q[0] = make_float2((c + 0 <= r) ? v[p].x : 0.f, (c + 1 <= r) ? v[p].y : 0.f);
            q[1] = make_float2((c + 2 <= r) ? v[p].z : 0.f, (c + 3 <= r) ? v[p].w : 0.f);
        }
    }
    __syncthreads();
    STAMP();
    // Look-ahead: while waves 1-3 apply the trailing update of block column J, wave 0 updates only the next
    // diagonal block and immediately factors/inverts it, so the serial 16x16 factorizations (the longest
    // single-wave stretch) hide behind the MFMA updates instead of adding to them.
    if (w == 0) diag16_factor_invert(a, 0, dinv, info, (int)k0, lane);
    __syncthreads();
    STAMP();
    for (int J = 0; J < 8; ++J) {
        const int j0 = 16 * J;
        const float* dJ = dinv + J * DINV_SZ;
        for (int I = J + 1 + w; I < 8; I += 4) {          // panel: P_I = A[I,J] * Dinv^T
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const float av = a[(16 * I + lr) * LDA + j0 + 4 * s + lg];
                const float bv = dJ[lr * DINV_LD + 4 * s + lg];               // B[k][j] = Dinv[j][k]
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc, 0, 0, 0);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) a[(16 * I + 4 * lg + e) * LDA + j0 + lr] = acc[e];
        }
        __syncthreads();
        STAMP();
        if (J == 7) break;
        const int n = 7 - J, np = n * (n + 1) / 2;        // trailing pairs (I >= K > J); pair 0 = (J+1, J+1)
        const int pfirst = (w == 0) ? 0 : w, pstep = (w == 0) ? np : 3;          // wave 0: pair 0 only
        for (int p = pfirst; p < np; p += pstep) {
            int kk = 0, rem = p;
            while (rem >= n - kk) { rem -= n - kk; ++kk; }
            const int K = J + 1 + kk, I = K + rem;
            f32x4 acc;
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] = a[(16 * I + 4 * lg + e) * LDA + 16 * K + lr];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const float av = -a[(16 * I + lr) * LDA + j0 + 4 * s + lg];
                const float bv = a[(16 * K + lr) * LDA + j0 + 4 * s + lg];
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc, 0, 0, 0);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) a[(16 * I + 4 * lg + e) * LDA + 16 * K + lr] = acc[e];
        }
        long long t0w = wall_clock64();
        if (w == 0) diag16_factor_invert(a, j0 + 16, dinv + (J + 1) * DINV_SZ, info, (int)(k0 + j0 + 16), lane);
        if (tid == 0) stamps[40 + J] = wall_clock64() - t0w;
        __syncthreads();
        STAMP();
    }
    // ---- T = L^-1 -----------------------------------------------------------------------------
    for (int idx = tid; idx < 8 * 256; idx += 256) {       // diagonal 16-blocks of T
        const int J = idx >> 8, rr = (idx >> 4) & 15, cc = idx & 15;
        t[(16 * J + rr) * LDA + 16 * J + cc] = dinv[J * DINV_SZ + rr * DINV_LD + cc];
    }
    __syncthreads();
    for (int hb = 1; hb <= 4; hb *= 2) {                   // half size in 16-blocks
        const int h = 16 * hb, npairs = 8 / (2 * hb), nout = npairs * hb * hb;
        // phase A: X = L21 * T11  -> upper mirror of a
        for (int o = w; o < nout; o += 4) {
            const int pr = o / (hb * hb), bi = (o / hb) % hb, bj = o % hb;
            const int c0 = pr * 2 * h;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            for (int kb = bj; kb < hb; ++kb) {
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const float av = a[(c0 + h + 16 * bi + lr) * LDA + c0 + 16 * kb + 4 * s + lg];
                    const float bv = t[(c0 + 16 * kb + 4 * s + lg) * LDA + c0 + 16 * bj + lr];
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc, 0, 0, 0);
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) a[(c0 + 16 * bi + 4 * lg + e) * LDA + c0 + h + 16 * bj + lr] = acc[e];
        }
        __syncthreads();
        // phase B: T21 = -T22 * X
        for (int o = w; o < nout; o += 4) {
            const int pr = o / (hb * hb), bi = (o / hb) % hb, bj = o % hb;
            const int c0 = pr * 2 * h;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            for (int kb = 0; kb <= bi; ++kb) {
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const float av = -t[(c0 + h + 16 * bi + lr) * LDA + c0 + h + 16 * kb + 4 * s + lg];
                    const float bv = a[(c0 + 16 * kb + 4 * s + lg) * LDA + c0 + h + 16 * bj + lr];
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc, 0, 0, 0);
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) t[(c0 + h + 16 * bi + 4 * lg + e) * LDA + c0 + 16 * bj + lr] = acc[e];
        }
        __syncthreads();
    }
    STAMP();
    float* Tg = tinv + (int64_t)block_index * NB * NB;
#pragma unroll
    for (int hp = 0; hp < 2; ++hp) {       // LDS reads of 8 rows first (ds_read_b64), then their stores
        float2 ql[8][2], zl[8][2];
#pragma unroll
        for (int pp = 0; pp < 8; ++pp) {
            const int r = (tid >> 5) + 8 * (hp * 8 + pp), c = (tid & 31) * 4;
            const float2* q = reinterpret_cast<const float2*>(a + r * LDA + c);
            const float2* z = reinterpret_cast<const float2*>(t + r * LDA + c);
            ql[pp][0] = q[0]; ql[pp][1] = q[1];
            zl[pp][0] = z[0]; zl[pp][1] = z[1];
        }
#pragma unroll
        for (int pp = 0; pp < 8; ++pp) {
            const int r = (tid >> 5) + 8 * (hp * 8 + pp), c = (tid & 31) * 4;
            float* g = Sb + (int64_t)r * ld + c;
            if (c + 3 <= r) *reinterpret_cast<float4*>(g) = make_float4(ql[pp][0].x, ql[pp][0].y, ql[pp][1].x, ql[pp][1].y);
            else {
                if (c + 0 <= r) g[0] = ql[pp][0].x;
                if (c + 1 <= r) g[1] = ql[pp][0].y;
                if (c + 2 <= r) g[2] = ql[pp][1].x;
            }
            // T above the diagonal was never written in LDS: select, do not multiply
            *reinterpret_cast<float4*>(Tg + r * NB + c) = make_float4(c + 0 <= r ? zl[pp][0].x : 0.f, c + 1 <= r ? zl[pp][0].y : 0.f,
                                                                       c + 2 <= r ? zl[pp][1].x : 0.f, c + 3 <= r ? zl[pp][1].y : 0.f);
        }
    }
    __syncthreads();
    STAMP();
    if (threadIdx.x == 0) stamps[39] = sidx;
}

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
    const char* names[] = {"", "load", "diag16(0)", "panel 0", "trail 0 + diag16(1)", "panel 1", "trail 1 + diag16(2)", "panel 2", "trail 2 + diag16(3)", "panel 3", "trail 3 + diag16(4)", "panel 4", "trail 4 + diag16(5)", "panel 5", "trail 5 + diag16(6)", "panel 6", "trail 6 + diag16(7)", "panel 7", "T = L^-1", "store"};
    int ns = (int)st[39];
    for (int i = 1; i < ns; ++i) printf("  %-26s %6.2f us (cum %6.2f)\n", names[i], (st[i] - st[i - 1]) * 0.01, (st[i] - st[0]) * 0.01);
    printf("  diag16 alone on wave 0: %.2f us;  |LL^T - A| %.2e  |L T - I| %.2e\n", st[41] * 0.01, e1, e2);
    return 0;
}

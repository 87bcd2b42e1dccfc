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
//   * K in steps of 32 through a double-buffered LDS image [128][32] floats per operand, filled by LDS-DMA
//     (global_load_lds_dwordx4: no staging registers, no ds_write) and XOR-swizzled at 16-byte granularity --
//     chunk c of row r lives at chunk c ^ (r & 7) -- so that every ds_read_b128 fragment read is conflict-free
//     without padding (a DMA instruction fills LDS linearly, padding is not expressible)
//   * a lane reads k = 8s+4h..+3 as one ds_read_b128 and feeds 4 consecutive MFMAs; A and B use
//     the same k permutation, so the product is unchanged
//   * software pipeline, skewed across the barrier: the fragments of MFMA group s+1 are read from
//     LDS while group s runs (two fragment register sets); the one barrier per K-tile sits before
//     the LAST group, so the first fragments of tile t+1 are fetched behind 16 MFMAs instead of in
//     front of an idle pipe (measured +7 %: 129 -> 138 TFLOP/s at 8192^3)
//   * the DMA of tile t+1 is issued at the top of tile t into the other buffer; each wave waits for its own
//     pieces (vmcnt) only right before the barrier: three MFMA groups of cover
//   * blockIdx -> tile map keeps each XCD (private 4 MiB L2) on a contiguous strip of tiles that
//     share B rows
// The 128x128 diagonal blocks are factored and inverted by one workgroup in LDS; the panel below
// a diagonal block is then a GEMM with the inverse (TRSM as GEMM, the MAGMA trick), so it also
// runs on MFMA.  The inverses are kept: the triangular solves reuse them.
#include "oisat_common.h"

#include <algorithm>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int NB = 128;          // diagonal block / tile edge
constexpr int BK = 32;           // K step
constexpr int LDSW = 36;         // padded LDS row stride in floats (144 B, 16-B aligned)

// ---- batched launches -------------------------------------------------------------------------
// Many independent factorizations (the tiles of a localised analysis, the months of a batch) advance in LOCK-STEP:
// one launch applies the same node of the recursion to every matrix -- blockIdx.y picks the matrix from a device
// table, blockIdx.x its tile -- so that a step which is a 1-workgroup kernel or a 40-tile GEMM for one 6,000-
// observation tile becomes one launch that fills the chip, and the chain of dependent launches is paid once per batch
// instead of once per tile.  The table is sorted by block count (largest first): the matrices a node applies to are a
// prefix of it.  Operands are derived in the kernel from (table entry, node), so nothing is uploaded per launch.
struct BatchArgs {
    const BatchMat* mats;        // nullptr: not a batched launch
    int kind;                    // 0: trailing update of node (b0, mid, b1);  1: TRSM of the panel below diagonal block b0
    int b0, mid, b1;
    const int* cum = nullptr;    // persistent kernel: prefix sums of the members' tile counts at this node (cum[cnt] = total):
    int cnt = 0, total = 0;      // virtual ids 0 .. total-1 are exactly the real tiles, member = the w with cum[w] <= id < cum[w+1]
    int prio = 0;                // wave priority of this group's kernels (oisat_set_share): where the waves of two groups share a
};                               // SIMD the arbiter serves the higher one first -- the group with the longest dependent chain

__device__ __forceinline__ void group_prio(int p) {        // p is wave-uniform (a kernel argument)
    if (p == 3) __builtin_amdgcn_s_setprio(3);
    else if (p == 2) __builtin_amdgcn_s_setprio(2);
    else if (p == 1) __builtin_amdgcn_s_setprio(1);
}

// operands of matrix `which` for this node, in units of `unit` rows/columns (128 or 64); false: the node does not apply
__device__ __forceinline__ bool batch_operands(const BatchArgs& ba, int which, int unit, float*& C, int64_t& ldc, const float*& A,
                                               int64_t& lda, const float*& B, int64_t& ldb, int& ntm, int& ntn) {
    const BatchMat m = ba.mats[which];
    const int per = NB / unit;
    if (ba.kind == 0) {                                     // S[mid:, mid:b1] -= L[mid:, b0:mid] * L[mid:b1, b0:mid]^T
        const int rows = m.mpb - ba.mid, cols = (ba.b1 < m.mpb ? ba.b1 : m.mpb) - ba.mid;
        if (rows <= 0 || cols <= 0) return false;
        ntm = rows * per;
        ntn = cols * per;
        C = m.S + (int64_t)ba.mid * NB * m.ld + (int64_t)ba.mid * NB;
        A = m.S + (int64_t)ba.mid * NB * m.ld + (int64_t)ba.b0 * NB;
        B = A;
        ldc = lda = ldb = m.ld;
    } else {                                                // P <- P * T_b0^T, P = S[(b0+1)*128:, b0*128 : (b0+1)*128], in place
        const int rows = m.mpb - ba.b0 - 1;
        if (rows <= 0) return false;
        ntm = rows * per;
        ntn = per;
        float* P = m.S + (int64_t)(ba.b0 + 1) * NB * m.ld + (int64_t)ba.b0 * NB;
        C = P;
        A = P;
        ldc = lda = m.ld;
        B = m.tinv + (int64_t)ba.b0 * NB * NB;
        ldb = NB;
    }
    return true;
}

// Virtual id -> work item for launches whose ids are NOT all equally loaded (batched launches: the ids past a smaller
// matrix's tile count are empty).  Workgroups go to the 8 XCDs round-robin in dispatch order; giving each XCD one
// contiguous eighth of the whole range (as gemm_nt_big does for a single matrix) would hand some XCDs mostly empty ids.
// Instead every run of 512 ids is dealt out in 64-id chunks: id 512g + 8s + x (XCD x, its s-th workgroup of the run)
// -> 512g + 64x + s, so an XCD still works on 64 consecutive items (an 8x8 patch of tiles in row-band order) and all
// XCDs sweep the range together.  The ragged tail of the range keeps its ids.
__device__ __forceinline__ int64_t xcd_chunk_remap(int64_t v, int64_t total) {
    const int64_t g = v >> 9;
    if (((g + 1) << 9) > total) return v;
    const int64_t x = v & 7, sidx = (v >> 3) & 63;
    return (g << 9) + (x << 6) + sidx;
}

// ---- gemm_nt_big: one tile per workgroup, for K >= 2048 --------------------------------------------------------------
// The kernel the headline factorization spends 95 % of its time in (round 1's gemm_nt, unchanged): with 64+ K-steps per
// tile the 22 us of per-tile overhead are < 10 %, and this instruction schedule of the K-loop sustains 140 TFLOP/s at
// 8192^3 where the pipelined-across-tiles kernel below, with 64 more live registers for the prefetched C tile, reaches
// 135 (measured: K = 1024: 913 vs 870 us for 3240 tiles, K = 2048: 1683 vs 1684, K = 5120: 4054 vs 4250).
template <bool BATCH>
__global__ __launch_bounds__(256, 2) void gemm_nt_big_kernel(float* C, int64_t ldc, const float* A,
                                                          int64_t lda, const float* __restrict__ B, int64_t ldb, int ntm,
                                                          int ntn, int K, int mode, int lower, int ntiles_total, BatchArgs ba) {
    if (BATCH) group_prio(ba.prio);
    __shared__ __attribute__((aligned(16))) float lds[2][2][NB * BK];        // [buf][A|B][row*32 + 4*(chunk ^ (row&7))] = 65,536 B
    int wg;
    if (BATCH) {
        const float* Bb = nullptr;
        // Workgroups go to the 8 XCDs round-robin in dispatch order (x fastest, then y): chunked remap over the 2-D grid
        // (xcd_chunk_remap), so that neighbouring tiles of the same matrix share an L2 and no XCD is left with the
        // empty ids of the smaller matrices.
        const int64_t total = (int64_t)gridDim.x * gridDim.y, orig = (int64_t)blockIdx.y * gridDim.x + blockIdx.x;
        const int64_t lin = xcd_chunk_remap(orig, total);
        const int which = (int)(lin / gridDim.x);
        if (!batch_operands(ba, which, NB, C, ldc, A, lda, Bb, ldb, ntm, ntn)) return;
        B = Bb;
        wg = (int)(lin - (int64_t)which * gridDim.x);
    } else {
        // XCD-aware, bijective remap: blocks b, b+8, b+16.. share an XCD -> give each XCD a contiguous strip
        const int nwg = gridDim.x;
        const int orig = blockIdx.x;
        const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    }
    // wg -> tile in ROW-BAND order: bands of 8 tile rows, inside a band column by column.  64 consecutive
    // workgroups (what one XCD runs at once: 32 CUs x 2) are then an 8x8 patch of tiles: per K-step they
    // pull 8 A + 8 B tiles through the XCD's L2 instead of 64 + 1 for a column strip.  In `lower` mode only
    // tiles with ti >= tj are enumerated (the last columns of a band are partial).  The band search is a
    // short wave-uniform loop (<= ntm/8 iterations of scalar arithmetic).
    int ti = 0, tj = 0;
    {
        int rem = wg;
        bool found = false;
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
                found = true;
                break;
            }
            rem -= count;
        }
        if (BATCH && !found) return;                        // this matrix has fewer tiles than the largest of the batch
    }
    (void)ntiles_total;
    const int t = threadIdx.x;
    const int lane = t & 63, wid = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wr = wid >> 1, wc = wid & 1;
    const float* Ag = A + (int64_t)ti * NB * lda;
    const float* Bg = B + (int64_t)tj * NB * ldb;
    // LDS-DMA (global_load_lds_dwordx4): wave `wid` brings rows wid*32 + 8i .. +7 (i = 0..3) of each operand, one KiB
    // per instruction, straight into LDS -- no staging registers, no ds_write, and the wave only waits for its pieces
    // right before the barrier.  A DMA instruction fills LDS linearly (lane l -> 16 bytes at 16 l), so the image cannot
    // be padded; it is XOR-swizzled instead: lane l lands at physical chunk (l&7) of row (l>>3) and therefore fetches
    // the LOGICAL chunk (l&7) ^ (row&7) of that row from global memory (still one 128-byte segment per row).
    typedef __attribute__((address_space(1))) const void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    const int rl = lane >> 3, lc = (lane & 7) ^ rl;
    const float* Ad = Ag + (int64_t)(wid * 32 + rl) * lda + 4 * lc;
    const float* Bd = Bg + (int64_t)(wid * 32 + rl) * ldb + 4 * lc;
#define OISAT_DMA(buf, k0)                                                                                         \
    do {                                                                                                           \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                            \
            __builtin_amdgcn_global_load_lds((gptr_t)(Ad + (int64_t)(8 * i) * lda + (k0)),                         \
                                             (lptr_t)&lds[buf][0][(wid * 32 + 8 * i) * BK], 16, 0, 0);             \
            __builtin_amdgcn_global_load_lds((gptr_t)(Bd + (int64_t)(8 * i) * ldb + (k0)),                         \
                                             (lptr_t)&lds[buf][1][(wid * 32 + 8 * i) * BK], 16, 0, 0);             \
        }                                                                                                          \
    } while (0)
    f32x16 acc00 = {0}, acc01 = {0}, acc10 = {0}, acc11 = {0};

    const int nkt = K / BK;
    OISAT_DMA(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int frow = lane & 31, fh = lane >> 5;
    // a lane's fragment of K-group s is logical chunk 2s+fh of its row: physical chunk (2s+fh) ^ (row&7).  Any 8 consecutive
    // rows hold one logical chunk in 8 different physical chunks = all 32 banks once: conflict-free ds_read_b128.
    const int sw = frow & 7;
    const int arow = (wr * 64 + frow) * BK, brow = (wc * 64 + frow) * BK;
    // fragment registers, two sets: the operands of MFMA group s+1 are read from LDS while group s runs
    float4 fa0, fa1, fb0, fb1, ga0, ga1, gb0, gb1;
#define OISAT_FRAG(A0, A1, B0, B1, buf, s)                                                      \
    do {                                                                                        \
        A0 = *reinterpret_cast<const float4*>(&lds[buf][0][arow + 4 * ((2 * (s) + fh) ^ sw)]);            \
        A1 = *reinterpret_cast<const float4*>(&lds[buf][0][arow + 32 * BK + 4 * ((2 * (s) + fh) ^ sw)]);  \
        B0 = *reinterpret_cast<const float4*>(&lds[buf][1][brow + 4 * ((2 * (s) + fh) ^ sw)]);            \
        B1 = *reinterpret_cast<const float4*>(&lds[buf][1][brow + 32 * BK + 4 * ((2 * (s) + fh) ^ sw)]);  \
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
        if (more) OISAT_DMA(cur ^ 1, (kt + 1) * BK);            // the other buffer is free since the last barrier
        OISAT_FRAG(ga0, ga1, gb0, gb1, cur, 1);
        OISAT_MFMA16(fa0, fa1, fb0, fb1)                       // s = 0
        OISAT_FRAG(fa0, fa1, fb0, fb1, cur, 2);
        OISAT_MFMA16(ga0, ga1, gb0, gb1)                       // s = 1
        OISAT_FRAG(ga0, ga1, gb0, gb1, cur, 3);
        OISAT_MFMA16(fa0, fa1, fb0, fb1)                       // s = 2
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's DMA pieces of tile kt+1 have landed ...
        __syncthreads();                                        // ... and so have everyone else's; every read of tile kt has been issued
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
#undef OISAT_DMA
#undef OISAT_FRAG
#undef OISAT_MFMA4
#undef OISAT_MFMA16
#undef OISAT_EPI

// ---- gemm_nt ---------------------------------------------------------------------------------
// mode 0: C -= A*B^T      mode 1: C = A*B^T (C may alias A when N == K == 128: TRSM-as-GEMM)
// lower != 0: the C region is anchored on the diagonal; tiles strictly above it are skipped.
//
// PERSISTENT workgroups: the grid is min(tiles, 2 per CU) and workgroup b walks the tiles b, b + G, b + 2G, ...  A
// tile's fixed costs -- the first operand load in front of an idle MFMA pipe, the read-modify-write of C behind it --
// measured 22 us per tile (the time of six K-steps; 37 us per tile at K = 128, 81 us at K = 512 against 15 / 62 us of
// MFMA work), which is what the tall-and-thin updates deep in the recursion and every update of a 4,000-18,000-
// observation tile are made of.  Here the pipeline runs ACROSS tiles: the first K-step of the next tile is fetched (LDS-
// DMA into the free buffer) during the last K-step of the current one, and the C tile is prefetched into registers at
// the start of that last K-step, so the epilogue is 64 stores and the next tile's MFMAs start right behind it.
// Bit-identical to the one-tile-per-workgroup form (same accumulation order, C - sum rounded once).
template <bool BATCH>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(float* C0, int64_t ldc0, const float* A0, int64_t lda0,
                                                          const float* __restrict__ B0, int64_t ldb0, int ntm0, int ntn0, int K,
                                                          int mode, int lower, int ntiles_total, BatchArgs ba, int* dyn) {
    __shared__ __attribute__((aligned(16))) float lds[2][2][NB * BK];        // [buf][A|B][row*32 + 4*(chunk ^ (row&7))] = 65,536 B
    // DYNAMIC tile walk (dyn != nullptr): a workgroup's first tile is its blockIdx, every further one a ticket drawn
    // from its XCD's counter (ids x, x + 8, x + 16, ... keep the XCD and the 64-id chunks of xcd_chunk_remap), other XCDs'
    // counters once its own is exhausted.  With the static walk b, b + G, ... a workgroup that starts late -- another
    // stream's kernel held its slot -- still has its whole share of tiles in front of it and the launch ends that much
    // later; with tickets the workgroups that did get a slot take the work.  dyn[0..7]: tickets per XCD, dyn[8]: workgroups
    // that have left; the last one to leave zeroes the block again (the next launch on the stream finds it clean).
    __shared__ int s_tkt[2];
    if (BATCH) group_prio(ba.prio);
    const int t = threadIdx.x;
    const int lane = t & 63, wid = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wr = wid >> 1, wc = wid & 1;
    // virtual tile ids: plain launch: the tiles of C;  batched launch: (ntm0 = tiles of the largest matrix) x (ntn0 = matrices)
    const bool compact = BATCH && ba.cum != nullptr;
    const int64_t VT = compact ? (int64_t)ba.total : (BATCH ? (int64_t)ntm0 * ntn0 : (int64_t)ntiles_total);
    int cw = 0;                                             // compact enumeration: member of the last id located (ids only grow)
    const int64_t G = gridDim.x;                            // a multiple of 8 whenever a workgroup gets more than one tile
    // LDS-DMA (global_load_lds_dwordx4): wave `wid` brings rows wid*32 + 8i .. +7 (i = 0..3) of each operand, one KiB
    // per instruction, straight into LDS -- no staging registers, no ds_write, and the wave only waits for its pieces
    // right before the barrier.  A DMA instruction fills LDS linearly (lane l -> 16 bytes at 16 l), so the image cannot
    // be padded; it is XOR-swizzled instead: lane l lands at physical chunk (l&7) of row (l>>3) and therefore fetches
    // the LOGICAL chunk (l&7) ^ (row&7) of that row from global memory (still one 128-byte segment per row).
    typedef __attribute__((address_space(1))) const void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    const int rl = lane >> 3, lc = (lane & 7) ^ rl;
    const int frow = lane & 31, fh = lane >> 5;

    // virtual id -> tile.  Workgroups go to the 8 XCDs round-robin in dispatch order, and v = b + iG keeps b's XCD (G is
    // a multiple of 512 / 8): xcd_chunk_remap hands each XCD (private 4 MiB L2) 64 consecutive ids at a time, which the
    // row-band order below turns into 8x8 patches of tiles that share operand panels (in a batched launch: neighbouring
    // tiles of the same matrix).
    // Row-band order: bands of 8 tile rows, inside a band column by column; in `lower` mode only tiles with ti >= tj
    // are enumerated (the last columns of a band are partial).  The band search is a short wave-uniform loop.
    const bool dynamic = dyn != nullptr && G < VT;         // (host: G is a multiple of 8 then, and every id is a real tile)
    const int xcd = (int)(blockIdx.x & 7);
    const int64_t G8 = G >> 3;
    auto leave = [&]() {                                    // thread 0, once per workgroup
        if (dyn != nullptr && t == 0) {
            const int gone = __hip_atomic_fetch_add(&dyn[8], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (gone == (int)G - 1) {
#pragma unroll
                for (int x = 0; x < 9; ++x) __hip_atomic_store(&dyn[x], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    };
    auto ticket_to_id = [&](int x, int tk) -> int64_t { return (int64_t)x + 8 * (G8 + (int64_t)tk); };
    auto steal = [&]() -> int {                             // thread 0: own counter exhausted, try the other XCDs' (end of launch only)
        for (int k = 1; k < 8; ++k) {
            const int x = (xcd + k) & 7;
            const int tk = __hip_atomic_fetch_add(&dyn[x], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int64_t vv = ticket_to_id(x, tk);
            if (vv < VT) return (int)vv;
        }
        return (int)VT;
    };
    auto locate = [&](int64_t v, const float*& oAd, const float*& oBd, float*& oCg, int64_t& olda, int64_t& oldb,
                      int64_t& oldc) -> int64_t {
        for (; v < VT; v += (dynamic ? VT : G)) {
            const int64_t lin = xcd_chunk_remap(v, VT);
            float* C = C0;
            const float* A = A0;
            const float* B = B0;
            int64_t lda = lda0, ldb = ldb0, ldc = ldc0;
            int ntm = ntm0, ntn = ntn0, wg = (int)lin;
            if (BATCH) {
                int which;
                if (compact) {                              // ids of a workgroup (almost) only grow: gallop forward from the last
                    int lo = cw, step = 1;                  // member, then bisect (wave-uniform: scalar loads); a ticket stolen
                    if ((int)lin < ba.cum[lo]) lo = 0;      // from another XCD's counter may lie behind it
                    while (lo + step < ba.cnt && ba.cum[lo + step] <= (int)lin) { lo += step; step <<= 1; }
                    int hi = lo + step < ba.cnt ? lo + step : ba.cnt;
                    while (hi - lo > 1) {
                        const int md = (lo + hi) >> 1;
                        if (ba.cum[md] <= (int)lin) lo = md; else hi = md;
                    }
                    cw = which = lo;
                    wg = (int)lin - ba.cum[lo];
                } else {
                    which = (int)(lin / ntm0);
                    wg = (int)(lin - (int64_t)which * ntm0);
                }
                if (!batch_operands(ba, which, NB, C, ldc, A, lda, B, ldb, ntm, ntn)) continue;
            }
            int ti = 0, tj = 0, rem = wg;
            bool found = false;
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
                    found = true;
                    break;
                }
                rem -= count;
            }
            if (!found) continue;                           // batched: this matrix has fewer tiles than the largest one
            oAd = A + ((int64_t)ti * NB + wid * 32 + rl) * lda + 4 * lc;
            oBd = B + ((int64_t)tj * NB + wid * 32 + rl) * ldb + 4 * lc;
            oCg = C + ((int64_t)ti * NB + wr * 64) * ldc + (int64_t)tj * NB + wc * 64;
            olda = lda;
            oldb = ldb;
            oldc = ldc;
            return v;
        }
        return -1;
    };
#define OISAT_DMA(buf, PA, PB, LA, LB, k0)                                                                         \
    do {                                                                                                           \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                            \
            __builtin_amdgcn_global_load_lds((gptr_t)((PA) + (int64_t)(8 * i) * (LA) + (k0)),                      \
                                             (lptr_t)&lds[buf][0][(wid * 32 + 8 * i) * BK], 16, 0, 0);             \
            __builtin_amdgcn_global_load_lds((gptr_t)((PB) + (int64_t)(8 * i) * (LB) + (k0)),                      \
                                             (lptr_t)&lds[buf][1][(wid * 32 + 8 * i) * BK], 16, 0, 0);             \
        }                                                                                                          \
    } while (0)
    const float *Ad = nullptr, *Bd = nullptr, *nAd = nullptr, *nBd = nullptr;
    float *Cg = nullptr, *nCg = nullptr;
    int64_t lda = 0, ldb = 0, ldc = 0, nlda = 0, nldb = 0, nldc = 0;
    int64_t v = locate(blockIdx.x, Ad, Bd, Cg, lda, ldb, ldc);
    if (v < 0) {
        leave();
        return;
    }
    if (dynamic && t == 0) {                                // ticket of this workgroup's SECOND tile
        const int tk = __hip_atomic_fetch_add(&dyn[xcd], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int64_t vv = ticket_to_id(xcd, tk);
        s_tkt[0] = vv < VT ? (int)vv : steal();
    }

    const int nkt = K / BK;
    OISAT_DMA(0, Ad, Bd, lda, ldb, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // a lane's fragment of K-group s is logical chunk 2s+fh of its row: physical chunk (2s+fh) ^ (row&7).  Any 8 consecutive
    // rows hold one logical chunk in 8 different physical chunks = all 32 banks once: conflict-free ds_read_b128.
    const int sw = frow & 7;
    const int arow = (wr * 64 + frow) * BK, brow = (wc * 64 + frow) * BK;
    // fragment registers, two sets: the operands of MFMA group s+1 are read from LDS while group s runs
    float4 fa0, fa1, fb0, fb1, ga0, ga1, gb0, gb1;
#define OISAT_FRAG(A0, A1, B0, B1, buf, s)                                                      \
    do {                                                                                        \
        A0 = *reinterpret_cast<const float4*>(&lds[buf][0][arow + 4 * ((2 * (s) + fh) ^ sw)]);            \
        A1 = *reinterpret_cast<const float4*>(&lds[buf][0][arow + 32 * BK + 4 * ((2 * (s) + fh) ^ sw)]);  \
        B0 = *reinterpret_cast<const float4*>(&lds[buf][1][brow + 4 * ((2 * (s) + fh) ^ sw)]);            \
        B1 = *reinterpret_cast<const float4*>(&lds[buf][1][brow + 32 * BK + 4 * ((2 * (s) + fh) ^ sw)]);  \
    } while (0)
#define OISAT_MFMA4(A0, A1, B0, B1, c)                                                          \
    acc00 = __builtin_amdgcn_mfma_f32_32x32x2f32(A0.c, B0.c, acc00, 0, 0, 0);                   \
    acc01 = __builtin_amdgcn_mfma_f32_32x32x2f32(A0.c, B1.c, acc01, 0, 0, 0);                   \
    acc10 = __builtin_amdgcn_mfma_f32_32x32x2f32(A1.c, B0.c, acc10, 0, 0, 0);                   \
    acc11 = __builtin_amdgcn_mfma_f32_32x32x2f32(A1.c, B1.c, acc11, 0, 0, 0);
#define OISAT_MFMA16(A0, A1, B0, B1)                                                            \
    OISAT_MFMA4(A0, A1, B0, B1, x) OISAT_MFMA4(A0, A1, B0, B1, y) OISAT_MFMA4(A0, A1, B0, B1, z) OISAT_MFMA4(A0, A1, B0, B1, w)
    // C/D layout: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5).  The 64 elements a lane owns are addressed through
    // a buffer descriptor on the wave's 64x64 quadrant: ONE per-lane offset register (row 4*fh, column frow) and a
    // wave-uniform scalar offset per element -- 64 separate 64-bit addresses would not fit next to 64 prefetched values.
#define OISAT_CRSRC()                                                                                   \
    __builtin_amdgcn_make_buffer_rsrc((void*)Cg, (short)0, (int)(64 * ldc * 4), 0x00020000)
#define OISAT_CSOFF(i, j, e) (int)((((i) * 32 + ((e) & 3) + 8 * ((e) >> 2)) * ldc + (j) * 32) * 4)
#define OISAT_CPRE(DST, i, j)                                                                           \
    _Pragma("unroll") for (int e = 0; e < 16; ++e)                                                      \
        DST[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(crs, cvoff, OISAT_CSOFF(i, j, e), 0));
#define OISAT_EPI(ACC, PRE, i, j)                                                                       \
    _Pragma("unroll") for (int e = 0; e < 16; ++e) {                                                    \
        const float o = (mode == 0) ? PRE[e] - ACC[e] : ACC[e];                                         \
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, o), crs, cvoff, OISAT_CSOFF(i, j, e), 0); \
    }
    int base = 0;                                           // LDS buffer that holds K-step 0 of the current tile
    int nth = 0;                                            // tiles this workgroup has started
    OISAT_FRAG(fa0, fa1, fb0, fb1, 0, 0);
    while (true) {
        // next tile: static v + G, or the ticket parked in LDS one tile ago (barriers in between); thread 0 then draws the
        // ticket of the tile after that -- the atomic's round trip hides behind this tile's K-loop, its value is parked
        // right before the tile's last barrier
        const int64_t nvid = dynamic ? (int64_t)s_tkt[nth & 1] : v + G;
        int pend = 0;
        if (dynamic && t == 0 && nvid < VT)
            pend = __hip_atomic_fetch_add(&dyn[xcd], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int64_t nv = nvid < VT ? locate(nvid, nAd, nBd, nCg, nlda, nldb, nldc) : -1;
        const bool has_next = nv >= 0;
        f32x16 acc00 = {0}, acc01 = {0}, acc10 = {0}, acc11 = {0};
        f32x16 c00 = {0}, c01 = {0}, c10 = {0}, c11 = {0};
        const __amdgpu_buffer_rsrc_t crs = OISAT_CRSRC();
        const int cvoff = (int)(((4 * fh) * ldc + frow) * 4);
        // steady state: the loop of the one-tile kernel, branch-free (one scheduling region)
        for (int kt = 0; kt + 1 < nkt; ++kt) {
            const int cur = (kt + base) & 1;
            OISAT_DMA(cur ^ 1, Ad, Bd, lda, ldb, (kt + 1) * BK);   // the other buffer is free since the last barrier
            OISAT_FRAG(ga0, ga1, gb0, gb1, cur, 1);
            OISAT_MFMA16(fa0, fa1, fb0, fb1)                       // s = 0
            OISAT_FRAG(fa0, fa1, fb0, fb1, cur, 2);
            OISAT_MFMA16(ga0, ga1, gb0, gb1)                       // s = 1
            OISAT_FRAG(ga0, ga1, gb0, gb1, cur, 3);
            OISAT_MFMA16(fa0, fa1, fb0, fb1)                       // s = 2
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's DMA pieces of K-step kt+1 have landed ...
            __syncthreads();                                        // ... and so have everyone else's; every read of K-step kt has been issued
            OISAT_FRAG(fa0, fa1, fb0, fb1, cur ^ 1, 0);             // first operands of the next K-step, behind the last MFMA group
            OISAT_MFMA16(ga0, ga1, gb0, gb1)                       // s = 3
        }
        {   // last K-step of this tile: K-step 0 of the NEXT tile goes to the free buffer, the C tile to registers
            const int cur = (nkt - 1 + base) & 1;
            if (has_next) OISAT_DMA(cur ^ 1, nAd, nBd, nlda, nldb, 0);
            if (mode == 0) {
                OISAT_CPRE(c00, 0, 0)
                OISAT_CPRE(c01, 0, 1)
                OISAT_CPRE(c10, 1, 0)
                OISAT_CPRE(c11, 1, 1)
            }
            OISAT_FRAG(ga0, ga1, gb0, gb1, cur, 1);
            OISAT_MFMA16(fa0, fa1, fb0, fb1)                       // s = 0
            OISAT_FRAG(fa0, fa1, fb0, fb1, cur, 2);
            OISAT_MFMA16(ga0, ga1, gb0, gb1)                       // s = 1
            OISAT_FRAG(ga0, ga1, gb0, gb1, cur, 3);
            OISAT_MFMA16(fa0, fa1, fb0, fb1)                       // s = 2
            if (dynamic && t == 0) {
                int64_t vv = VT;
                if (nvid < VT) {
                    vv = ticket_to_id(xcd, pend);
                    if (vv >= VT) vv = steal();
                }
                s_tkt[(nth + 1) & 1] = (int)vv;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (has_next) OISAT_FRAG(fa0, fa1, fb0, fb1, cur ^ 1, 0);
            OISAT_MFMA16(ga0, ga1, gb0, gb1)                       // s = 3
        }
        OISAT_EPI(acc00, c00, 0, 0)
        OISAT_EPI(acc01, c01, 0, 1)
        OISAT_EPI(acc10, c10, 1, 0)
        OISAT_EPI(acc11, c11, 1, 1)
        if (!has_next) break;
        base = (base + nkt) & 1;
        ++nth;
        v = nv;
        Ad = nAd; Bd = nBd; Cg = nCg;
        lda = nlda; ldb = nldb; ldc = nldc;
    }
    leave();
}

// ---- gemm_nt for launches that cannot fill the chip with 128x128 tiles ------------------------------
// Deep in the recursion the updates are tall and thin (e.g. 71 x 2 blocks, K = 3 blocks): 141 tiles of 128x128 on
// 512 workgroup slots, every workgroup alone on its CU for 12 K-tiles of ~2.3 us plus load/epilogue latency.  The
// same launch cut into 64x64 tiles is 564 workgroups of a quarter of the work each, four per CU: ~2x faster.
// 4 waves as 2x2, one 32x32 MFMA tile per wave, BK = 32, same LDS image rows (36-float stride), same k permutation
// and per-element accumulation order as gemm_nt_kernel (bit-identical results).
// NBUF: LDS images per operand.  2 = double-buffered (36,864 B).  1 = single image, one more barrier per K-step (18,432 B):
// the batched instantiation -- its launches are the dependent chain of one group of systems while ANOTHER group's
// 128x128 GEMMs hold two 64-KB workgroups on every CU, which leaves 32 KB of the 160: a 36-KB workgroup then waits for
// one of them to retire (hundreds of microseconds with the persistent kernel), an 18-KB one moves in next to them.
constexpr int SB = 64;
template <bool BATCH, int NBUF>
__global__ __launch_bounds__(256) void gemm_nt_small_kernel(float* C, int64_t ldc, const float* A, int64_t lda,
                                                             const float* __restrict__ B, int64_t ldb, int ntm, int ntn, int K,
                                                             int mode, int lower, BatchArgs ba) {
    __shared__ __attribute__((aligned(16))) float lds[NBUF][2][SB * LDSW];    // 36,864 B / 18,432 B
    if (BATCH) group_prio(ba.prio);
    if (BATCH) {
        const float* Bb = nullptr;
        if (!batch_operands(ba, blockIdx.y, SB, C, ldc, A, lda, Bb, ldb, ntm, ntn)) return;
        B = Bb;
    }
    // tile decode: column-major; in `lower` mode column c holds rows c .. ntm-1 (units of 64)
    int ti, tj;
    {
        int rem = blockIdx.x;
        if (!lower) {
            tj = rem / ntm;
            ti = rem - tj * ntm;
            if (BATCH && tj >= ntn) return;
        } else {
            int c = 0;
            while (c < ntn && rem >= ntm - c) { rem -= ntm - c; ++c; }
            if (BATCH && c >= ntn) return;
            tj = c;
            ti = c + rem;
        }
    }
    const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
    const int wr = wid >> 1, wc = wid & 1;
    const float* Ag = A + (int64_t)ti * SB * lda;
    const float* Bg = B + (int64_t)tj * SB * ldb;
    const int srow = t >> 3, sk = (t & 7) * 4;           // rows srow and srow+32, 16 bytes at k = sk
    const float* Ap = Ag + (int64_t)srow * lda + sk;
    const float* Bp = Bg + (int64_t)srow * ldb + sk;
    float4 ra0, ra1, rb0, rb1;
#define OISAT_SGLOAD(k0)                                                             \
    do {                                                                             \
        ra0 = *reinterpret_cast<const float4*>(Ap + (k0));                           \
        ra1 = *reinterpret_cast<const float4*>(Ap + 32 * lda + (k0));                \
        rb0 = *reinterpret_cast<const float4*>(Bp + (k0));                           \
        rb1 = *reinterpret_cast<const float4*>(Bp + 32 * ldb + (k0));                \
    } while (0)
#define OISAT_SLSTORE(buf)                                                           \
    do {                                                                             \
        *reinterpret_cast<float4*>(&lds[buf][0][srow * LDSW + sk]) = ra0;            \
        *reinterpret_cast<float4*>(&lds[buf][0][(srow + 32) * LDSW + sk]) = ra1;     \
        *reinterpret_cast<float4*>(&lds[buf][1][srow * LDSW + sk]) = rb0;            \
        *reinterpret_cast<float4*>(&lds[buf][1][(srow + 32) * LDSW + sk]) = rb1;     \
    } while (0)
    f32x16 acc = {0};
    const int nkt = K / BK;
    OISAT_SGLOAD(0);
    OISAT_SLSTORE(0);
    __syncthreads();
    const int frow = lane & 31, fh = lane >> 5;
    const int aoff = (wr * 32 + frow) * LDSW + 4 * fh, boff = (wc * 32 + frow) * LDSW + 4 * fh;
    for (int kt = 0; kt < nkt; ++kt) {
        const int cur = NBUF == 2 ? (kt & 1) : 0;
        const bool more = kt + 1 < nkt;
        if (more) OISAT_SGLOAD((kt + 1) * BK);
        float4 fa[4], fb[4];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            fa[s4] = *reinterpret_cast<const float4*>(&lds[cur][0][aoff + 8 * s4]);
            fb[s4] = *reinterpret_cast<const float4*>(&lds[cur][1][boff + 8 * s4]);
        }
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s4].x, fb[s4].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s4].y, fb[s4].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s4].z, fb[s4].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s4].w, fb[s4].w, acc, 0, 0, 0);
        }
        if (NBUF == 1 && more) __syncthreads();                  // every wave has read this image
        if (more) OISAT_SLSTORE(NBUF == 2 ? (cur ^ 1) : 0);
        __syncthreads();
    }
    float* Cg = C + ((int64_t)ti * SB + wr * 32) * ldc + (int64_t)tj * SB + wc * 32;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int row = (e & 3) + 8 * (e >> 2) + 4 * fh;
        float* p = Cg + (int64_t)row * ldc + frow;
        if (mode == 0) *p = *p - acc[e];
        else *p = acc[e];
    }
}

// The in-place TRSM-as-GEMM (C aliases A, N == K == 128) cannot use 64-wide tiles: a tile would overwrite panel columns
// its row neighbour still reads.  64 rows x all 128 columns per workgroup instead (each wave 32 x 64 = two MFMA tiles):
// a workgroup owns whole rows and has read every K-tile of them before its epilogue writes.
template <bool BATCH, int NBUF>
__global__ __launch_bounds__(256) void gemm_nt_rows64_kernel(float* C, int64_t ldc, const float* A, int64_t lda,
                                                              const float* __restrict__ B, int64_t ldb, int K, int mode, BatchArgs ba) {
    __shared__ __attribute__((aligned(16))) float ldsA[NBUF][SB * LDSW];     // 18,432 B / 9,216 B
    __shared__ __attribute__((aligned(16))) float ldsB[NBUF][NB * LDSW];     // 36,864 B / 18,432 B  (NBUF = 1: 27,648 B in all)
    if (BATCH) group_prio(ba.prio);
    if (BATCH) {
        int ntm, ntn;
        const float* Bb = nullptr;
        if (!batch_operands(ba, blockIdx.y, SB, C, ldc, A, lda, Bb, ldb, ntm, ntn) || (int)blockIdx.x >= ntm) return;
        B = Bb;
    }
    const int ti = blockIdx.x;
    const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
    const int wr = wid >> 1, wc = wid & 1;
    const int srow = t >> 3, sk = (t & 7) * 4;
    const float* Ap = A + ((int64_t)ti * SB + srow) * lda + sk;
    const float* Bp = B + (int64_t)srow * ldb + sk;
    float4 ra0, ra1, rb0, rb1, rb2, rb3;
#define OISAT_RGLOAD(k0)                                                             \
    do {                                                                             \
        ra0 = *reinterpret_cast<const float4*>(Ap + (k0));                           \
        ra1 = *reinterpret_cast<const float4*>(Ap + 32 * lda + (k0));                \
        rb0 = *reinterpret_cast<const float4*>(Bp + (k0));                           \
        rb1 = *reinterpret_cast<const float4*>(Bp + 32 * ldb + (k0));                \
        rb2 = *reinterpret_cast<const float4*>(Bp + 64 * ldb + (k0));                \
        rb3 = *reinterpret_cast<const float4*>(Bp + 96 * ldb + (k0));                \
    } while (0)
#define OISAT_RLSTORE(buf)                                                           \
    do {                                                                             \
        *reinterpret_cast<float4*>(&ldsA[buf][srow * LDSW + sk]) = ra0;              \
        *reinterpret_cast<float4*>(&ldsA[buf][(srow + 32) * LDSW + sk]) = ra1;       \
        *reinterpret_cast<float4*>(&ldsB[buf][srow * LDSW + sk]) = rb0;              \
        *reinterpret_cast<float4*>(&ldsB[buf][(srow + 32) * LDSW + sk]) = rb1;       \
        *reinterpret_cast<float4*>(&ldsB[buf][(srow + 64) * LDSW + sk]) = rb2;       \
        *reinterpret_cast<float4*>(&ldsB[buf][(srow + 96) * LDSW + sk]) = rb3;       \
    } while (0)
    f32x16 acc0 = {0}, acc1 = {0};
    const int nkt = K / BK;
    OISAT_RGLOAD(0);
    OISAT_RLSTORE(0);
    __syncthreads();
    const int frow = lane & 31, fh = lane >> 5;
    const int aoff = (wr * 32 + frow) * LDSW + 4 * fh, boff = (wc * 64 + frow) * LDSW + 4 * fh;
    for (int kt = 0; kt < nkt; ++kt) {
        const int cur = NBUF == 2 ? (kt & 1) : 0;
        const bool more = kt + 1 < nkt;
        if (more) OISAT_RGLOAD((kt + 1) * BK);
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            const float4 a = *reinterpret_cast<const float4*>(&ldsA[cur][aoff + 8 * s4]);
            const float4 b0 = *reinterpret_cast<const float4*>(&ldsB[cur][boff + 8 * s4]);
            const float4 b1 = *reinterpret_cast<const float4*>(&ldsB[cur][boff + 32 * LDSW + 8 * s4]);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b0.x, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b1.x, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b0.y, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b1.y, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b0.z, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b1.z, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b0.w, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b1.w, acc1, 0, 0, 0);
        }
        if (NBUF == 1 && more) __syncthreads();
        if (more) OISAT_RLSTORE(NBUF == 2 ? (cur ^ 1) : 0);
        __syncthreads();
    }
    float* Cg = C + ((int64_t)ti * SB + wr * 32) * ldc + wc * 64;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int row = (e & 3) + 8 * (e >> 2) + 4 * fh;
        float* p = Cg + (int64_t)row * ldc + frow;
        if (mode == 0) { p[0] = p[0] - acc0[e]; p[32] = p[32] - acc1[e]; }
        else { p[0] = acc0[e]; p[32] = acc1[e]; }
    }
}

// ---- leaf PAIRS: two block columns (b, b+1) per leaf of the recursion ---------------------------------------------------
// With one block per leaf the rows below a pair of diagonal blocks are swept three times at K = 128 -- TRSM with T_b, the
// rank-128 update of column b+1, TRSM with T_b+1 -- each pass reading and writing a 64-KB tile per 4.2 MFLOP (16 flop/B:
// HBM-bound; in a month of 48 tiles these launches are 40 % of the factorization time for 10 % of its flops).  The pair
// form makes ONE pass:  P1 = X1 T_b^T;  X2 -= P1 L21^T;  P2 = X2 T_b+1^T  per 64 rows, with P1 and X2 handed from one
// product to the next through an LDS image (pair_panel_kernel), after a one-workgroup step has produced
// L21 = S[b+1,b] T_b^T and S[b+1,b+1] -= L21 L21^T between the two diagonal factorizations (pair_mid_kernel).
// Same products, same k order and the same single rounding of C - sum as the three launches they replace.
typedef float v4f __attribute__((ext_vector_type(4)));
constexpr int LDP = NB + 4;                // row stride of a full-K LDS operand image (floats)

// whole B operand [128 x 128] of one product into registers: 16 float4 per thread (chunk kt = rb[4 kt .. 4 kt + 3]), all
// loads in flight at once -- these kernels are latency-bound when few workgroups run (a polar cap, a single system)
__device__ __forceinline__ void b_load_all(v4f (&rb)[16], const float* __restrict__ g, int64_t ldg, int t) {
    const float* p = g + (int64_t)(t >> 3) * ldg + (t & 7) * 4;
#define OISAT_BL(kt, q) rb[4 * (kt) + (q)] = *reinterpret_cast<const v4f*>(p + (int64_t)(32 * (q)) * ldg + (kt) * BK);
    OISAT_BL(0, 0) OISAT_BL(0, 1) OISAT_BL(0, 2) OISAT_BL(0, 3) OISAT_BL(1, 0) OISAT_BL(1, 1) OISAT_BL(1, 2) OISAT_BL(1, 3)
    OISAT_BL(2, 0) OISAT_BL(2, 1) OISAT_BL(2, 2) OISAT_BL(2, 3) OISAT_BL(3, 0) OISAT_BL(3, 1) OISAT_BL(3, 2) OISAT_BL(3, 3)
#undef OISAT_BL
}

// acc0 | acc1 (a wave's 32 x 64 piece: rows wr*32.., columns wc*64.. and +32) += A[64 x 128] B[128 x 128]^T, A from a full-K
// LDS image (stride LDP), B from registers (b_load_all) through the one-chunk LDS image `ldsB`.  next != nullptr: as soon
// as chunk kt has gone to LDS its registers are refilled with chunk kt of the NEXT product's B operand, so that product
// finds its operand in registers -- one register set, and only the first product of a kernel waits for memory.
template <int KT, bool NEXT>
__device__ __forceinline__ void rows64_chunk(f32x16& acc0, f32x16& acc1, const float* __restrict__ imgA, v4f (&rb)[16],
                                             float* __restrict__ ldsB, int t, const float* __restrict__ next, int64_t ldnext) {
    const int lane = t & 63, wid = t >> 6;
    const int wr = wid >> 1, wc = wid & 1;
    const int frow = lane & 31, fh = lane >> 5;
    const int aoff = (wr * 32 + frow) * LDP + 4 * fh, boff = (wc * 64 + frow) * LDSW + 4 * fh;
    const int srow = t >> 3, sk = (t & 7) * 4;
    __syncthreads();                                                    // the previous chunk (or the image's writers) are done
    *reinterpret_cast<v4f*>(&ldsB[(srow + 0) * LDSW + sk]) = rb[4 * KT + 0];
    *reinterpret_cast<v4f*>(&ldsB[(srow + 32) * LDSW + sk]) = rb[4 * KT + 1];
    *reinterpret_cast<v4f*>(&ldsB[(srow + 64) * LDSW + sk]) = rb[4 * KT + 2];
    *reinterpret_cast<v4f*>(&ldsB[(srow + 96) * LDSW + sk]) = rb[4 * KT + 3];
    if (NEXT) {
        const float* g = next + (int64_t)srow * ldnext + KT * BK + sk;
        rb[4 * KT + 0] = *reinterpret_cast<const v4f*>(g);
        rb[4 * KT + 1] = *reinterpret_cast<const v4f*>(g + 32 * ldnext);
        rb[4 * KT + 2] = *reinterpret_cast<const v4f*>(g + 64 * ldnext);
        rb[4 * KT + 3] = *reinterpret_cast<const v4f*>(g + 96 * ldnext);
    }
    __syncthreads();
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
        const float4 a = *reinterpret_cast<const float4*>(&imgA[aoff + KT * BK + 8 * s4]);
        const float4 b0 = *reinterpret_cast<const float4*>(&ldsB[boff + 8 * s4]);
        const float4 b1 = *reinterpret_cast<const float4*>(&ldsB[boff + 32 * LDSW + 8 * s4]);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b0.x, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b1.x, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b0.y, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b1.y, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b0.z, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b1.z, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b0.w, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b1.w, acc1, 0, 0, 0);
    }
}

template <bool NEXT>
__device__ __forceinline__ void rows64_product(f32x16& acc0, f32x16& acc1, const float* __restrict__ imgA, v4f (&rb)[16],
                                               float* __restrict__ ldsB, int t, const float* __restrict__ next, int64_t ldnext) {
    rows64_chunk<0, NEXT>(acc0, acc1, imgA, rb, ldsB, t, next, ldnext);
    rows64_chunk<1, NEXT>(acc0, acc1, imgA, rb, ldsB, t, next, ldnext);
    rows64_chunk<2, NEXT>(acc0, acc1, imgA, rb, ldsB, t, next, ldnext);
    rows64_chunk<3, NEXT>(acc0, acc1, imgA, rb, ldsB, t, next, ldnext);
}

// element (e, half) of a wave's accumulator pair -> (row, column) inside the 64 x 128 piece of the workgroup
#define OISAT_PIECE_ROW(e) (wr * 32 + ((e) & 3) + 8 * ((e) >> 2) + 4 * fh)
#define OISAT_PIECE_COL(half) (wc * 64 + 32 * (half) + frow)

// rows of 64 below a leaf pair (b, b+1): one pass over S[rows, b*128 : (b+2)*128].  Every global load is issued as early
// as registers allow: X1 and T_b at once, L21 while the first product runs, T_b+1 and X2 while the second one does.
template <bool BATCH>
__global__ __launch_bounds__(256) void pair_panel_kernel(float* __restrict__ S, int64_t ld, int mpb, const float* __restrict__ tinv,
                                                          int b, const BatchMat* __restrict__ mats, int prio) {
    __shared__ __attribute__((aligned(16))) float img[SB * LDP];         // 33,792 B: X1 -> P1 -> X2'
    __shared__ __attribute__((aligned(16))) float ldsB[NB * LDSW];       // 18,432 B
    group_prio(prio);
    if (BATCH) {
        const BatchMat* bm = mats + blockIdx.y;
        S = bm->S;
        ld = bm->ld;
        mpb = bm->mpb;
        tinv = bm->tinv;
    }
    if ((int)blockIdx.x >= (mpb - b - 2) * 2) return;                   // this matrix has fewer rows below the pair (or none)
    const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
    const int wr = wid >> 1, wc = wid & 1, frow = lane & 31, fh = lane >> 5;
    float* X1 = S + ((int64_t)(b + 2) * NB + (int64_t)blockIdx.x * SB) * ld + (int64_t)b * NB;      // my 64 rows, column block b
    float* X2 = X1 + NB;                                                                            // ... column block b+1
    const float* Tb = tinv + (int64_t)b * NB * NB;
    const float* Tb1 = Tb + NB * NB;
    const float* L21 = S + (int64_t)(b + 1) * NB * ld + (int64_t)b * NB;
    v4f rb[16];
    {
        v4f xr[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) xr[q] = *reinterpret_cast<const v4f*>(X1 + (int64_t)((t >> 5) + 8 * q) * ld + (t & 31) * 4);
        b_load_all(rb, Tb, NB, t);
#pragma unroll
        for (int q = 0; q < 8; ++q) *reinterpret_cast<v4f*>(&img[((t >> 5) + 8 * q) * LDP + (t & 31) * 4]) = xr[q];
    }
    f32x16 acc0 = {0}, acc1 = {0};
    rows64_product<true>(acc0, acc1, img, rb, ldsB, t, L21, ld);              // P1 = X1 T_b^T            (rb <- L21 on the way)
    __syncthreads();                                                    // every wave has read X1 from the image
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int r = OISAT_PIECE_ROW(e);
        X1[(int64_t)r * ld + OISAT_PIECE_COL(0)] = acc0[e];
        X1[(int64_t)r * ld + OISAT_PIECE_COL(1)] = acc1[e];
        img[r * LDP + OISAT_PIECE_COL(0)] = acc0[e];
        img[r * LDP + OISAT_PIECE_COL(1)] = acc1[e];
    }
    f32x16 x2a, x2b;                                                    // my elements of X2, in accumulator layout
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int r = OISAT_PIECE_ROW(e);
        x2a[e] = X2[(int64_t)r * ld + OISAT_PIECE_COL(0)];
        x2b[e] = X2[(int64_t)r * ld + OISAT_PIECE_COL(1)];
    }
    acc0 = f32x16{0};
    acc1 = f32x16{0};
    rows64_product<true>(acc0, acc1, img, rb, ldsB, t, Tb1, NB);              // Q = P1 L21^T             (rb <- T_b+1 on the way)
    __syncthreads();                                                    // every wave has read P1 from the image
#pragma unroll
    for (int e = 0; e < 16; ++e) {                                      // X2' = X2 - Q, rounded once, as the rank-128 update does
        const int r = OISAT_PIECE_ROW(e);
        img[r * LDP + OISAT_PIECE_COL(0)] = x2a[e] - acc0[e];
        img[r * LDP + OISAT_PIECE_COL(1)] = x2b[e] - acc1[e];
    }
    acc0 = f32x16{0};
    acc1 = f32x16{0};
    rows64_product<false>(acc0, acc1, img, rb, ldsB, t, (const float*)nullptr, 0);      // P2 = X2' T_b+1^T
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int r = OISAT_PIECE_ROW(e);
        X2[(int64_t)r * ld + OISAT_PIECE_COL(0)] = acc0[e];
        X2[(int64_t)r * ld + OISAT_PIECE_COL(1)] = acc1[e];
    }
}

// one 64-row half of pair_mid's second product: D_half -= L21_half L21^T (B = L21 from registers)
__device__ __forceinline__ void pair_mid_update(float* __restrict__ Dh, int64_t ld, const float* __restrict__ imgA, v4f (&rb)[16],
                                                float* __restrict__ ldsB, int t, bool keep_b) {
    const int lane = t & 63, wid = t >> 6;
    const int wr = wid >> 1, wc = wid & 1, frow = lane & 31, fh = lane >> 5;
    f32x16 d0, d1;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int r = OISAT_PIECE_ROW(e);
        d0[e] = Dh[(int64_t)r * ld + OISAT_PIECE_COL(0)];
        d1[e] = Dh[(int64_t)r * ld + OISAT_PIECE_COL(1)];
    }
    f32x16 a0 = {0}, a1 = {0};
    (void)keep_b;
    rows64_product<false>(a0, a1, imgA, rb, ldsB, t, (const float*)nullptr, 0);
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int r = OISAT_PIECE_ROW(e);
        Dh[(int64_t)r * ld + OISAT_PIECE_COL(0)] = d0[e] - a0[e];
        Dh[(int64_t)r * ld + OISAT_PIECE_COL(1)] = d1[e] - a1[e];
    }
}

// between the two diagonal factorizations of a pair: L21 = S[b+1,b] T_b^T (in place), S[b+1,b+1] -= L21 L21^T.  One
// workgroup per matrix, on the dependent chain: X and T_b are requested at once, L21 is handed from the first product to
// the second through LDS (as A image and, chunk by chunk, as B), 64 rows at a time.
template <bool BATCH>
__global__ __launch_bounds__(256) void pair_mid_kernel(float* __restrict__ S, int64_t ld, int mpb, const float* __restrict__ tinv, int b,
                                                        const BatchMat* __restrict__ mats, int prio) {
    __shared__ __attribute__((aligned(16))) float img0[SB * LDP], img1[SB * LDP];      // the two 64-row halves of X, then of L21
    __shared__ __attribute__((aligned(16))) float ldsB[NB * LDSW];                     // 67,584 + 18,432 B
    group_prio(prio);
    if (BATCH) {
        const BatchMat* bm = mats + blockIdx.x;
        S = bm->S;
        ld = bm->ld;
        mpb = bm->mpb;
        tinv = bm->tinv;
    }
    if (b + 1 >= mpb) return;                                           // the pair's second block does not exist in this matrix
    const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
    const int wr = wid >> 1, wc = wid & 1, frow = lane & 31, fh = lane >> 5;
    float* X = S + (int64_t)(b + 1) * NB * ld + (int64_t)b * NB;          // S[b+1, b]
    float* D = X + NB;                                                    // S[b+1, b+1]
    v4f rb[16];
    {
        v4f xr[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) xr[q] = *reinterpret_cast<const v4f*>(X + (int64_t)((t >> 5) + 8 * q) * ld + (t & 31) * 4);
        b_load_all(rb, tinv + (int64_t)b * NB * NB, NB, t);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            *reinterpret_cast<v4f*>(&img0[((t >> 5) + 8 * q) * LDP + (t & 31) * 4]) = xr[q];
            *reinterpret_cast<v4f*>(&img1[((t >> 5) + 8 * q) * LDP + (t & 31) * 4]) = xr[8 + q];
        }
    }
    // L21 = X T_b^T: both halves need the same B, so its registers are refilled with T_b itself for the second half
    f32x16 la0 = {0}, la1 = {0}, lb0 = {0}, lb1 = {0};
    rows64_product<true>(la0, la1, img0, rb, ldsB, t, tinv + (int64_t)b * NB * NB, NB);
    rows64_product<false>(lb0, lb1, img1, rb, ldsB, t, (const float*)nullptr, 0);
    __syncthreads();                                                    // every wave has read X from the images
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int r = OISAT_PIECE_ROW(e);
        X[(int64_t)r * ld + OISAT_PIECE_COL(0)] = la0[e];
        X[(int64_t)r * ld + OISAT_PIECE_COL(1)] = la1[e];
        X[(int64_t)(SB + r) * ld + OISAT_PIECE_COL(0)] = lb0[e];
        X[(int64_t)(SB + r) * ld + OISAT_PIECE_COL(1)] = lb1[e];
        img0[r * LDP + OISAT_PIECE_COL(0)] = la0[e];
        img0[r * LDP + OISAT_PIECE_COL(1)] = la1[e];
        img1[r * LDP + OISAT_PIECE_COL(0)] = lb0[e];
        img1[r * LDP + OISAT_PIECE_COL(1)] = lb1[e];
    }
    __syncthreads();
    // B operand of the second product = L21 itself: its K-chunks come from the images (row r of L21, the chunk's columns)
    const int srow = t >> 3, sk = (t & 7) * 4;
#define OISAT_LOAD_L21()                                                                                        \
    do {                                                                                                        \
        rb[0] = *reinterpret_cast<const v4f*>(&img0[srow * LDP + 0 * BK + sk]);                              \
        rb[1] = *reinterpret_cast<const v4f*>(&img0[(srow + 32) * LDP + 0 * BK + sk]);                       \
        rb[2] = *reinterpret_cast<const v4f*>(&img1[srow * LDP + 0 * BK + sk]);                              \
        rb[3] = *reinterpret_cast<const v4f*>(&img1[(srow + 32) * LDP + 0 * BK + sk]);                       \
        rb[4] = *reinterpret_cast<const v4f*>(&img0[srow * LDP + 1 * BK + sk]);                              \
        rb[5] = *reinterpret_cast<const v4f*>(&img0[(srow + 32) * LDP + 1 * BK + sk]);                       \
        rb[6] = *reinterpret_cast<const v4f*>(&img1[srow * LDP + 1 * BK + sk]);                              \
        rb[7] = *reinterpret_cast<const v4f*>(&img1[(srow + 32) * LDP + 1 * BK + sk]);                       \
        rb[8] = *reinterpret_cast<const v4f*>(&img0[srow * LDP + 2 * BK + sk]);                              \
        rb[9] = *reinterpret_cast<const v4f*>(&img0[(srow + 32) * LDP + 2 * BK + sk]);                       \
        rb[10] = *reinterpret_cast<const v4f*>(&img1[srow * LDP + 2 * BK + sk]);                             \
        rb[11] = *reinterpret_cast<const v4f*>(&img1[(srow + 32) * LDP + 2 * BK + sk]);                      \
        rb[12] = *reinterpret_cast<const v4f*>(&img0[srow * LDP + 3 * BK + sk]);                             \
        rb[13] = *reinterpret_cast<const v4f*>(&img0[(srow + 32) * LDP + 3 * BK + sk]);                      \
        rb[14] = *reinterpret_cast<const v4f*>(&img1[srow * LDP + 3 * BK + sk]);                             \
        rb[15] = *reinterpret_cast<const v4f*>(&img1[(srow + 32) * LDP + 3 * BK + sk]);                      \
    } while (0)
    OISAT_LOAD_L21();
    pair_mid_update(D, ld, img0, rb, ldsB, t, true);
    OISAT_LOAD_L21();
    pair_mid_update(D + (int64_t)SB * ld, ld, img1, rb, ldsB, t, false);
#undef OISAT_LOAD_L21
}
#undef OISAT_PIECE_ROW
#undef OISAT_PIECE_COL

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Factor and invert a 16x16 diagonal block in ONE sweep, in registers.  Lane r of every 16-lane row holds row r of the
// block (d[c] = A[r][c]) and column r of the running inverse (x[c] = X[c][r], X = L^-1, starts as I): after pivot step k
// column k of L is final, which is exactly what the forward substitution for X needs next, so each broadcast
// L[c][k] = d[k] of lane c feeds both the factor's trailing update and the inverse's running sums.
// The broadcast is the DPP operand of the FMA itself (row_newbcast:c = lane c of each 16-lane row, gfx90a+):
//     d[c] += (-d[k] @ lane c) * d[k]        x[c] += (-d[k] @ lane c) * x[k]        (v_fmac_f32_dpp)
// two vector instructions per (k, c) and no scalar registers.  (Rounds 1-2 broadcast through v_readlane: 240 scalar
// values live across the unrolled sweep, which the compiler spilled to VGPR lanes with ~350 v_writelane / ~640
// v_readlane -- 310 cycles per pivot; the builtin DPP move is not folded into the FMA by the compiler either, hence
// the inline assembly.)  Hazard: a VGPR written by a VALU instruction may be read through DPP two wait states later at
// the earliest and the compiler does not see into the asm blocks -- diag16_settle() puts the s_nop between the
// instruction that finishes column k and its DPP readers, and every asm block here is volatile, i.e. stays in order.
template <int C>
__device__ __forceinline__ float row16_bcast_settled(float v) {          // v was last written by one of the asm blocks
    float o;
    asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(o) : "v"(v), "n"(C));
    return o;
}

__device__ __forceinline__ void diag16_settle(float& dk, float& xk) {
    asm volatile("s_nop 1" : "+v"(dk), "+v"(xk));
}

template <int K, int C, int CEND>            // columns C .. CEND-1 receive pivot K's update
__device__ __forceinline__ void diag16_columns(float (&d)[16], float (&x)[16]) {
    if constexpr (C < CEND) {
        asm volatile("v_fmac_f32_dpp %0, -%2, %2 row_newbcast:%4 row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f32_dpp %1, -%2, %3 row_newbcast:%4 row_mask:0xf bank_mask:0xf"
                     : "+v"(d[C]), "+v"(x[C]) : "v"(d[K]), "v"(x[K]), "n"(C));
        diag16_columns<K, C + 1, CEND>(d, x);
    }
}

// pivot K: broadcast, reciprocal square root (+ one Newton step), scale column K of L and row K of X.  Plain C between
// the asm blocks: the compiler is free to spread these dependent instructions among the (independent) FMAs of the
// previous pivot that follow in program order.
template <int K>
__device__ __forceinline__ void diag16_pivot(float (&d)[16], float (&x)[16], int r, int& bad) {
    float piv = row16_bcast_settled<K>(d[K]);
    const bool neg = !(piv > 0.f);
    bad = (neg && bad == 0) ? K + 1 : bad;
    piv = neg ? 1.f : piv;
    float ri = __builtin_amdgcn_rsqf(piv);
    ri = ri * (1.5f - 0.5f * piv * ri * ri);              // one Newton step: ~0.5 ulp
    d[K] = (r == K) ? piv * ri : d[K] * ri;               // L[k][k] = sqrt(piv); column k of L
    x[K] = (K >= r) ? x[K] * ri : 0.f;                    // X[k][:] /= L[k][k]
}

// Software-pipelined sweep: as soon as column K+1 has received pivot K's update, pivot K+1's dependent chain
// (broadcast -> rsq -> Newton -> scale) is started, and the remaining updates of pivot K (columns K+2..15, independent
// of it) fill its latency.  Same arithmetic in the same order as the plain sweep.
template <int K>
__device__ __forceinline__ void diag16_pivots(float (&d)[16], float (&x)[16], int r, int& bad) {
    if constexpr (K == 0) diag16_pivot<0>(d, x, r, bad);
    if constexpr (K < 15) {
        diag16_settle(d[K], x[K]);
        diag16_columns<K, K + 1, K + 2>(d, x);             // column K+1 is final
        diag16_pivot<K + 1>(d, x, r, bad);
        diag16_columns<K, K + 2, 16>(d, x);
        diag16_pivots<K + 1>(d, x, r, bad);
    }
}

__device__ __forceinline__ void diag16_report(int bad, int* info, int col0, int lane, int which) {
    if (bad && lane == 0) {               // info[0]: this factorization; info[1], info[2]: sticky (first column, count)
        atomicCAS(info, 0, col0 + bad);  //          until oisat_solve_status clears them
        atomicCAS(info + 1, 0, col0 + bad);
        atomicAdd(info + 2, 1);
        atomicCAS(info + 3, 0, which + 1);             // batched factorization: which matrix of the table (1-based)
    }
}

// ---- diagonal block, REGISTER-RESIDENT: Cholesky + inverse of one 128x128 block, one workgroup of 4 waves, 24 KB of LDS ------
// The 36 lower 16x16 tiles of the block live in MFMA accumulator registers for the whole kernel: the 8 diagonal tiles in
// wave 0 (which also runs the serial 16x16 factorizations), off-diagonal tile (I, K), idx = I(I-1)/2 + K, in wave
// 1 + idx % 3, slot idx / 3 (10 / 9 / 9 tiles).  A slot's accumulator holds A[I,K] until its panel step K makes it the
// final L[I,K]; from then on the SAME register accumulates the running inverse X[I,K] until step I finishes it.  The
// A phase holds the tile TRANSPOSED (lane: A[lr][4 lg + e], one float4 of a row): the trailing update just swaps its two
// operands, and the panel product P_I^T = Dinv_J * A[I,J]^T then takes the held tile straight from the registers as the
// MFMA's B operand (d3_mma_regb) -- as does X[J,K] = Dinv_J * accX[J,K] -- so the workers never stage a tile.  Every
// tile coordinate is a compile-time constant of the wave's code path (one instantiation per wave), so each wave
// executes straight-line code with exactly its own tile operations.  LDS only
// carries what other waves need as MFMA operands: the current panel column P (= final L[:, J]), row J of X = L^-1, the
// inverse of the current diagonal 16x16 block, and one staging tile for the 16x16 factorization (accumulator layout ->
// lane = row).  Right-looking at 16-column granularity, per block column J:
//   (1) the wave that owns tile (J, J) factors and inverts it in registers (diag16_pivots: lane = row);
//   (2) panel:      P_I = A[I,J] * Dinv_J^T  (I > J)           row J of X:  X[J,K] = Dinv_J * accX[J,K]  (K < J)
//   (3) trailing:   A[I,K] -= P_I * P_K^T  (I >= K > J)        inverse:     accX[I,K] -= P_I * X[J,K]    (I > J >= K)
// i.e. the forward substitution for X = L^-1 rides along with the factorization (row J of X is final as soon as Dinv_J
// is) instead of a separate doubling phase, the trailing updates never move an accumulator through LDS, and the owner
// of (J+1, J+1) updates that tile first and factors it while the other waves finish step J (look-ahead).
typedef __attribute__((address_space(1))) float gfloat;          // global address space: global_load / global_store, which the
typedef __attribute__((address_space(1))) f32x4 gf32x4;         // LDS-only barrier does not wait for (flat_* count on lgkmcnt too)
constexpr int D3_WAVES = 4, D3_THREADS = 64 * D3_WAVES, D3_LD = 17, D3_TILE = 16 * D3_LD, D3_PBUF = 7 * D3_TILE;

constexpr int d3_I(int idx) {              // idx = I(I-1)/2 + K, 0 <= K < I < 8
    int I = 1;
    while (I * (I + 1) / 2 <= idx) ++I;
    return I;
}
constexpr int d3_K(int idx) { return idx - d3_I(idx) * (d3_I(idx) - 1) / 2; }

// acc (rows 4*lg + e, column lr) += sum_k (+-a[lr][k]) * B[k][lr-th column]: 4 MFMAs over k = 4s + lg
// A operand: element [lr][4s + lg] of a row-major 16 x D3_LD tile (negated if NEG); B operand either
//   BT = true:  B[k][col] = bt[col][k]  -> element [lr][4s + lg] of bt           (C -= A * B^T form)
//   BT = false: B[k][col] = b[k][col]   -> element [4s + lg][lr] of b
template <bool NEG, bool BT>
__device__ __forceinline__ f32x4 d3_mma(f32x4 acc, const float* a, const float* b, int lr, int lg) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const float av = a[lr * D3_LD + 4 * s + lg];
        const float bv = BT ? b[lr * D3_LD + 4 * s + lg] : b[(4 * s + lg) * D3_LD + lr];
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(NEG ? -av : av, bv, acc, 0, 0, 0);
    }
    return acc;
}

// D = a * R with R held in accumulator layout by the calling wave (lane: R[4 lg + e][lr]): the MFMA's k index is only a
// summation index, so step s takes k = 4 lg + s -- B operand = the lane's own breg[s], A operand = a[lr][4 lg + s].
// No staging through LDS.
__device__ __forceinline__ f32x4 d3_mma_regb(const float* a, f32x4 breg, int lr, int lg) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[lr * D3_LD + 4 * lg + s], breg[s], acc, 0, 0, 0);
    return acc;
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for the wave's outstanding GLOBAL stores
// (vmcnt(0)) -- here the final L and T tiles stream out to HBM in every step and nobody in the workgroup reads them back,
// so waiting for their acknowledgement (1-2 us each time) is pure loss.
__device__ __forceinline__ void d3_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Global stores of the diagonal-block kernel: (wave-uniform base, per-lane element offset).  WT (the task-graph factorization,
// where OTHER workgroups of the same launch read L and T): write-through `sc1` stores -- the bytes leave the XCD's L2 at once,
// so publishing them takes a drained vmcnt and a flag instead of an L2 write-back (buffer_wbl2) per diagonal block.  Buffer
// stores, not inline assembly: the stored values come straight out of MFMA instructions, and the wait states between an MFMA
// and a memory instruction that reads its result are the compiler's to insert -- it does not look into asm blocks (a
// global_store in inline assembly stored the register's OLD content: one wrong element per tile, found the hard way).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t d3_rsrc(const gfloat* base) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, (short)0, 0x7ffffff0, 0x00020000);
}
template <bool WT>
__device__ __forceinline__ void d3_gst(gfloat* base, int64_t off, float v) {
    if (WT) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), d3_rsrc(base), (int)(off * 4), 0, 16);
    else base[off] = v;
}
template <bool WT>
__device__ __forceinline__ void d3_gst4(gfloat* base, int64_t off, f32x4 v) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    if (WT) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), d3_rsrc(base), (int)(off * 4), 0, 16);
    else *reinterpret_cast<gf32x4*>(base + off) = v;
}

// Loads of the block's tiles.  WT: `sc1` buffer loads (served by L2, never by this CU's L1, which may hold the tile as it was
// before another workgroup -- or this one's write-through stores -- updated it).
template <bool WT>
__device__ __forceinline__ float d3_gld(const gfloat* Sb, int64_t ld, int row, int col) {
    if (WT) {
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(d3_rsrc(Sb), (int)((row * ld + col) * 4), 0, 16));
    }
    return Sb[(int64_t)row * ld + col];
}
template <bool WT>
__device__ __forceinline__ f32x4 d3_gld4(const gfloat* Sb, int64_t ld, int row, int col) {
    if (WT) {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(d3_rsrc(Sb), (int)((row * ld + col) * 4), 0, 16));
    }
    return *reinterpret_cast<const gf32x4*>(Sb + (int64_t)row * ld + col);
}

__device__ __forceinline__ void d3_put(float* tile, f32x4 acc, int lr, int lg) {   // accumulator layout -> row-major tile
#pragma unroll
    for (int e = 0; e < 4; ++e) tile[(4 * lg + e) * D3_LD + lr] = acc[e];
}

// factor + invert the diagonal tile held in `acc` (by the calling wave): L16 (full rows, the upper part is garbage) to
// `stage`, X16 = L16^-1 to `dinv`; a worker wave copies both to global memory in the next panel phase
constexpr int D3_SLD = 20;                 // row stride of the staging tile: 16-byte aligned rows (ds_read_b128 / ds_write_b128)
__device__ __forceinline__ void d3_diag16(f32x4 acc, float* stage, float* dinv, int* info, int col0, int lane, int lr, int lg, int which) {
#pragma unroll
    for (int e = 0; e < 4; ++e) stage[(4 * lg + e) * D3_SLD + lr] = acc[e];
    __builtin_amdgcn_wave_barrier();
    float d[16], x[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(stage + lr * D3_SLD + 4 * q);
#pragma unroll
        for (int e = 0; e < 4; ++e) d[4 * q + e] = v[e];
    }
#pragma unroll
    for (int c = 0; c < 16; ++c) x[c] = (c == lr) ? 1.f : 0.f;
    int bad = 0;
    diag16_pivots<0>(d, x, lr, bad);
    diag16_report(bad, info, col0, lane, which);
    __builtin_amdgcn_wave_barrier();
    if (lane < 16) {                       // (the other three 16-lane rows hold copies)
#pragma unroll
        for (int q = 0; q < 4; ++q)
            *reinterpret_cast<f32x4*>(stage + lr * D3_SLD + 4 * q) = f32x4{d[4 * q], d[4 * q + 1], d[4 * q + 2], d[4 * q + 3]};   // L16[lr][:]
#pragma unroll
        for (int c = 0; c < 16; ++c) dinv[c * D3_LD + lr] = x[c];                                                             // X16[c][lr]
    }
}

// a worker copies the freshly factored diagonal tile J to global memory: L16 (lower part) and T's diagonal tile
template <bool WT>
__device__ __forceinline__ void d3_store_diag(gfloat* Sjj, int64_t ld, gfloat* Tjj, const float* stage, const float* dinv, int lr, int lg) {
    const f32x4 l = *reinterpret_cast<const f32x4*>(stage + lr * D3_SLD + 4 * lg);
    f32x4 t;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        t[e] = dinv[lr * D3_LD + 4 * lg + e];
        if (4 * lg + e <= lr) d3_gst<WT>(Sjj, (int64_t)lr * ld + 4 * lg + e, l[e]);
    }
    d3_gst4<WT>(Tjj, lr * NB + 4 * lg, t);
}

// wave 0: the diagonal tiles
template <bool WT>
__device__ __forceinline__ void d3_diagonal_wave(gfloat* Sb, int64_t ld, const float* Pp, float* dinvb, float* stage, int* info,
                                                 int col0, int lane, int lr, int lg, int which) {
    f32x4 accD[8];
#pragma unroll
    for (int J = 0; J < 8; ++J)
#pragma unroll
        for (int e = 0; e < 4; ++e) accD[J][e] = d3_gld<WT>(Sb, ld, 16 * J + 4 * lg + e, 16 * J + lr);
    d3_diag16(accD[0], stage, dinvb, info, col0, lane, lr, lg, which);
#pragma unroll
    for (int J = 0; J < 8; ++J) {
        d3_barrier();                      // Dinv_J is in LDS; every wave is done with step J-1
        if (J >= 1) {                      // while the workers build panel J: the later diagonal tiles catch up with panel J-1
            const float* pp = Pp + ((J - 1) & 1) * D3_PBUF;
#pragma unroll
            for (int K = J + 1; K < 8; ++K) accD[K] = d3_mma<true, true>(accD[K], pp + (K - 1) * D3_TILE, pp + (K - 1) * D3_TILE, lr, lg);
        }
        d3_barrier();                      // the panel and row J of X are in LDS
        if (J == 7) break;
        // look-ahead: tile (J+1, J+1) takes panel J's update and is factored at once (its older updates are in already)
        const float* pj = Pp + (J & 1) * D3_PBUF + J * D3_TILE;
        accD[J + 1] = d3_mma<true, true>(accD[J + 1], pj, pj, lr, lg);
        d3_diag16(accD[J + 1], stage, dinvb + ((J + 1) & 1) * D3_TILE, info, col0 + 16 * (J + 1), lane, lr, lg, which);
    }
}

// waves 1..3: the off-diagonal tiles idx = 3 t + W - 1
template <int W, bool WT>
__device__ __forceinline__ void d3_worker_wave(gfloat* Sb, int64_t ld, gfloat* Tg, float* Pp, float* Xr, const float* dinvb, const float* stage,
                                               int lr, int lg) {
    constexpr int NS = (28 - (W - 1) + 2) / 3;
    f32x4 acc[NS];
#pragma unroll
    for (int t = 0; t < NS; ++t) {
        const int I = d3_I(3 * t + W - 1), K = d3_K(3 * t + W - 1);
        acc[t] = d3_gld4<WT>(Sb, ld, 16 * I + lr, 16 * K + 4 * lg);                                             // A[I,K]^T
        d3_gst4<WT>(Tg, (16 * K + lr) * NB + 16 * I + 4 * lg, f32x4{0.f, 0.f, 0.f, 0.f});                         // T is lower triangular
    }
#pragma unroll
    for (int J = 0; J < 8; ++J) {
        const float* dinv = dinvb + (J & 1) * D3_TILE;
        float* P = Pp + (J & 1) * D3_PBUF;                 // panel J: tile I (>= 1) at P + (I - 1) * D3_TILE; two buffers, by parity of J
        d3_barrier();                      // Dinv_J is in LDS; every wave is done with step J-1 (P, Xr may be rewritten)
        if (J % 3 == W - 1)                // the diagonal tile wave 0 has just factored: to global memory
            d3_store_diag<WT>(Sb + (int64_t)16 * J * ld + 16 * J, ld, Tg + 16 * J * NB + 16 * J, stage, dinv, lr, lg);
#pragma unroll
        for (int t = 0; t < NS; ++t) {
            const int I = d3_I(3 * t + W - 1), K = d3_K(3 * t + W - 1);
            if (K == J) {                  // panel tile: P_I^T = Dinv_J * A[I,J]^T, P_I = final L[I,J]; the slot turns to X[I,J]
                const f32x4 pt = d3_mma_regb(dinv, acc[t], lr, lg);             // lane: P_I[lr][4 lg + e]
#pragma unroll
                for (int e = 0; e < 4; ++e) P[(I - 1) * D3_TILE + lr * D3_LD + 4 * lg + e] = pt[e];
                d3_gst4<WT>(Sb, (int64_t)(16 * I + lr) * ld + 16 * J + 4 * lg, pt);
                acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
            } else if (I == J) {           // row J of the inverse (K < J): X[J,K] = Dinv_J * acc
                const f32x4 xf = d3_mma_regb(dinv, acc[t], lr, lg);
                d3_put(Xr + K * D3_TILE, xf, lr, lg);
#pragma unroll
                for (int e = 0; e < 4; ++e) d3_gst<WT>(Tg, (16 * J + 4 * lg + e) * NB + 16 * K + lr, xf[e]);
            }
        }
        d3_barrier();                      // the panel and row J of X are in LDS
        if (J == 7) break;
#pragma unroll
        for (int t = 0; t < NS; ++t) {
            const int I = d3_I(3 * t + W - 1), K = d3_K(3 * t + W - 1);
            if (I > J) {
                if (K > J) acc[t] = d3_mma<true, true>(acc[t], P + (K - 1) * D3_TILE, P + (I - 1) * D3_TILE, lr, lg);          // A[I,K]^T -= P_K P_I^T
                else acc[t] = d3_mma<true, false>(acc[t], P + (I - 1) * D3_TILE, K == J ? dinv : Xr + K * D3_TILE, lr, lg);   // X[I,K] -= L[I,J] X[J,K]
            }
        }
    }
}

__global__ __launch_bounds__(D3_THREADS, 4) void potrf_diag3_kernel(float* __restrict__ S, int64_t ld, int64_t k0, float* __restrict__ tinv,
                                                                    int* __restrict__ info, int block_index, const BatchMat* __restrict__ mats,
                                                                    int prio) {
    __shared__ __attribute__((aligned(16))) float Pp[2 * D3_PBUF], Xr[7 * D3_TILE], dinvb[2 * D3_TILE], stage[16 * D3_SLD];
    group_prio(prio);
    if (mats) {                            // batched: workgroup = matrix blockIdx.x of the table (largest first)
        const BatchMat bm = mats[blockIdx.x];
        if (block_index >= bm.mpb) return;
        S = bm.S;
        ld = bm.ld;
        tinv = bm.tinv;
    }
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, lg = lane >> 4;
    gfloat* Sb = (gfloat*)(S + k0 * ld + k0);
    gfloat* Tg = (gfloat*)(tinv + (int64_t)block_index * NB * NB);
    if (w == 0) d3_diagonal_wave<false>(Sb, ld, Pp, dinvb, stage, info, (int)k0, lane, lr, lg, (int)blockIdx.x);
    else if (w == 1) d3_worker_wave<1, false>(Sb, ld, Tg, Pp, Xr, dinvb, stage, lr, lg);
    else if (w == 2) d3_worker_wave<2, false>(Sb, ld, Tg, Pp, Xr, dinvb, stage, lr, lg);
    else d3_worker_wave<3, false>(Sb, ld, Tg, Pp, Xr, dinvb, stage, lr, lg);
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
// ONE launch per sweep.  Workgroup with ticket b owns block row b (forward) / block column b
// (backward).  It streams its 128x128 blocks of L through LDS (prefetching the next block into
// registers while it waits), applies each update as soon as the producing workgroup has published
// that piece of the solution, then multiplies by the inverted diagonal block and publishes its own
// piece.  Tickets are drawn from an atomic counter, so a workgroup only ever waits on workgroups
// that started before it: no assumption on dispatch order or residency.
// Hand-off: the PAYLOAD IS THE FLAG.  The solution vector is pre-filled with a NaN bit pattern no
// computation produces; the producer publishes its 128 doubles with agent-scope relaxed atomic
// stores (each 8-byte store is self-contained, so no fence, no vmcnt drain, no separate flag), the
// consumer's lanes each poll their own element with agent-scope atomic loads until it differs from
// the pattern.  Agent-scope traffic bypasses the per-XCD L2s, ~2 us a round trip: this form costs
// two of them per block step on the critical path, the flag-after-payload form it replaces cost
// five (store, drain, flag store | flag poll, payload load) -- 11.5 -> 6 us per step.  Spins are bounded.
constexpr int TLD = NB + 1;              // LDS tile row stride (odd: row- and column-walks are conflict-free)
constexpr unsigned long long kTrsvEmpty = 0x7ff80bad7ff80badull;      // a quiet NaN with that payload, both halves equal

constexpr int kInfoUnconverged = 4, kInfoUnconvergedMember = 5, kInfoDagTimeouts = 6;     // status words (status_ws below)

struct TrsvCtl {                         // zero when a sweep starts: zeroed at allocation, then by the last workgroup of every sweep
    unsigned ticket;
    unsigned error;
    unsigned done;
    unsigned pad;
};


// 256 threads: thread t moves 16 B at row (t>>5)+8p, column (t&31)*4 -> every row is one 512-B segment.
// Macros, not functions: the 16 x float4 staging registers must stay in VGPRs (arrays passed by
// reference ended up in scratch).
#define TILE_PREFETCH(src, ldsrc)                                                                              \
    _Pragma("unroll") for (int p = 0; p < 16; ++p)                                                             \
        reg[p] = *reinterpret_cast<const float4*>((src) + (int64_t)((tid >> 5) + 8 * p) * (ldsrc) + (tid & 31) * 4);
#define TILE_STORE()                                                                                           \
    _Pragma("unroll") for (int p = 0; p < 16; ++p) {                                                           \
        float* q = tile + ((tid >> 5) + 8 * p) * TLD + (tid & 31) * 4;                                         \
        q[0] = reg[p].x; q[1] = reg[p].y; q[2] = reg[p].z; q[3] = reg[p].w;                                    \
    }

// threads 0..127 fetch one published double each into vec[]; returns false if a producer never showed up
__device__ __forceinline__ bool wait_payload(const double* src, double* vec, TrsvCtl* ctl, unsigned* err_total, int tid,
                                             unsigned* lds_ok) {
    if (tid < NB) {
        const unsigned long long* p = reinterpret_cast<const unsigned long long*>(src) + tid;
        unsigned long long bits;
        unsigned spins = 0;
        while ((bits = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == kTrsvEmpty) {
            __builtin_amdgcn_s_sleep(1);
            ++spins;
            if ((spins & 1023u) == 0u &&
                (spins > (1u << 22) || __hip_atomic_load(&ctl->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
                if (__hip_atomic_exchange(&ctl->error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u)
                    atomicAdd(err_total, 1u);             // survives the per-sweep memset: read by checked solves
                *lds_ok = 0u;
                break;
            }
        }
        vec[tid] = __builtin_bit_cast(double, bits);
    }
    __syncthreads();
    return *lds_ok != 0u;
}

// forward: L y = r.  transpose == 0.   backward: L^T z = y.  transpose == 1 (block index runs downwards).
// sol must arrive filled with kTrsvEmpty.
// A workgroup claims block rows by ticket until none is left (round 3).  A row waits only for rows with lower tickets, and
// every claimed row is in the hands of a running workgroup, so the sweep drains with ANY number of resident workgroups.
// The grid is one workgroup per OISAT_TRSV_ROWS_PER_WG rows; the default stays 1 (a workgroup per row, as in rounds 1-2):
// fewer, persistent workgroups would hold fewer of the slots the other group's GEMMs run on, but a row's producers are
// then fetched one agent-scope round trip after the other instead of while waiting for its turn -- measured at 4 rows per
// workgroup: 0.54 vs 0.25 ms per sweep at 10,000 observations, a localised month 73.3 vs 69.6 ms.
// one block row of a sweep: claim order tk (0 .. nb-1) of ITS system; false = a producer never showed up (bounded spin)
// DAG: the row is a task of the task-graph launch (dense_dag.inc) -- the vectors are then shared with workgroups of the SAME
// launch across sweeps, so every store to them is an agent-scope (write-through) store and every load of a word another
// workgroup wrote an agent-scope load (cdna_hip_programming.md Guideline 16); between launches plain accesses do.
template <int TRANSPOSE, bool DAG>
__device__ __forceinline__ bool trsv_row(const float* __restrict__ L, int64_t ld, const float* __restrict__ tinv, int nb, int tk,
                                         double* __restrict__ rhs, double* __restrict__ sol, TrsvCtl* __restrict__ ctl,
                                         unsigned* __restrict__ err_total, int two_tiles, double* __restrict__ zout, int64_t m,
                                         int accumulate, float* __restrict__ tile, double* __restrict__ vec, double* __restrict__ part,
                                         unsigned* __restrict__ s_ok) {
    const int tid = threadIdx.x;
    const int row = tid & (NB - 1), hf = tid >> 7;       // two threads per row: columns [64*hf, 64*hf+64)
    float* const tileT = tile + NB * TLD;
    const int b = TRANSPOSE ? nb - 1 - tk : tk;             // my block row (fwd) / block column (bwd)
    double acc = 0.0;
    if (hf == 0) {
        // my block of the right-hand side is read by nobody else: take it and leave the "not yet published" pattern
        // behind, so that the NEXT sweep (which publishes its solution into this vector) finds it prepared
        if (DAG) {
            acc = __hip_atomic_load(&rhs[(int64_t)b * NB + row], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(reinterpret_cast<unsigned long long*>(rhs) + (int64_t)b * NB + row, kTrsvEmpty, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        } else {
            acc = rhs[(int64_t)b * NB + row];
            reinterpret_cast<unsigned long long*>(rhs)[(int64_t)b * NB + row] = kTrsvEmpty;
        }
    }
    float4 reg[16];
    const int nsteps = tk;                                  // producers: claim orders 0 .. tk-1
    // step-th producer j = step (fwd) / nb-1-step (bwd); its block is L[b, j] (fwd, j < b) or L[j, b] (bwd, j > b)
    const float* Tb = tinv + (int64_t)b * NB * NB;
    const float* base = TRANSPOSE ? L + (int64_t)(nb - 1) * NB * ld + (int64_t)b * NB : L + (int64_t)b * NB * ld;
    const int64_t hop = TRANSPOSE ? -(int64_t)NB * ld : (int64_t)NB;      // pointer step from one producer's block to the next
    // two_tiles: T_b goes to a second LDS tile right away -- it depends on nobody -- so that the last pass finds it there
    // instead of loading and staging it behind the last producer's hand-over
    if (two_tiles) {
        TILE_PREFETCH(Tb, NB)
        float* tile = tileT;                                // TILE_STORE writes to the `tile` in scope
        TILE_STORE()
    }
    if (nsteps > 0) { TILE_PREFETCH(base, ld) } else if (!two_tiles) { TILE_PREFETCH(Tb, NB) }
    for (int step = 0; step <= nsteps; ++step) {
        const bool last = step == nsteps;                   // last pass: multiply by the inverted diagonal block
        if (!(last && two_tiles)) TILE_STORE()              // this step's block: in LDS before the wait, off the critical path
        if (!last) {
            const int j = TRANSPOSE ? nb - 1 - step : step;
            if (!wait_payload(sol + (int64_t)j * NB, vec, ctl, err_total, tid, s_ok)) return false;
            if (step + 1 < nsteps) { const float* nx = base + (int64_t)(step + 1) * hop; TILE_PREFETCH(nx, ld) }
            else if (!two_tiles) { TILE_PREFETCH(Tb, NB) }
        } else {
            if (hf == 0) vec[row] = acc;
            __syncthreads();
        }
        double u = 0.0;
        const int c0 = hf * 64;
        const float* blk = (last && two_tiles) ? tileT : tile;
        if (!TRANSPOSE) {
#pragma unroll 8
            for (int c = c0; c < c0 + 64; ++c) u += (double)blk[row * TLD + c] * vec[c];       // row of the block
        } else {
#pragma unroll 8
            for (int c = c0; c < c0 + 64; ++c) u += (double)blk[c * TLD + row] * vec[c];       // column of the block
        }
        if (hf == 1) part[row] = u;
        __syncthreads();
        if (hf == 0) {
            if (!last) acc -= u + part[row];
            else acc = u + part[row];
        }
    }
    if (hf == 0) {
        const int64_t i = (int64_t)b * NB + row;
        __hip_atomic_store(&sol[i], acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (zout != nullptr && i < m) {                     // the solve's result where the caller wants it
            if (DAG) {
                const double zi = accumulate ? __hip_atomic_load(&zout[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + acc : acc;
                __hip_atomic_store(&zout[i], zi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                zout[i] = accumulate ? zout[i] + acc : acc;
            }
        }
    }
    return true;
}

__device__ __forceinline__ void trsv_leave(TrsvCtl* __restrict__ ctl) {        // thread 0: the last workgroup to leave hands
    const unsigned gone = __hip_atomic_fetch_add(&ctl->done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);    // the block back clean
    if (gone == gridDim.x - 1u) {
        __hip_atomic_store(&ctl->ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&ctl->error, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&ctl->done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

template <int TRANSPOSE>
__global__ __launch_bounds__(256) void trsv_pipe_kernel(const float* __restrict__ L, int64_t ld, const float* __restrict__ tinv, int nb,
                                                         double* __restrict__ rhs, double* __restrict__ sol,
                                                         TrsvCtl* __restrict__ ctl, unsigned* __restrict__ err_total, int two_tiles,
                                                         const SolveState* __restrict__ st, double* __restrict__ zout, int64_t m,
                                                         int accumulate) {
    extern __shared__ __attribute__((aligned(16))) float tile[];        // [128][TLD]  (+ a second one for T_b if two_tiles)
    __shared__ double vec[NB], part[NB];
    __shared__ unsigned s_ticket, s_ok;
    const int tid = threadIdx.x;
    if (st != nullptr && st->conv != 0) return;          // refinement already converged: this sweep is not needed (block-uniform)
    while (true) {
        __syncthreads();                                 // the previous row's LDS (vec, part, tiles, ticket) is no longer read
        if (tid == 0) {
            s_ticket = atomicAdd(&ctl->ticket, 1u);
            s_ok = 1u;
        }
        __syncthreads();
        const int tk = (int)s_ticket;                    // 0 .. nb-1 in claim order
        if (tk >= nb) break;
        if (!trsv_row<TRANSPOSE, false>(L, ld, tinv, nb, tk, rhs, sol, ctl, err_total, two_tiles, zout, m, accumulate, tile, vec, part, &s_ok))
            break;
    }
    if (tid == 0) trsv_leave(ctl);
}

// The sweeps of MANY systems in one launch (oisat_batch_solve): tickets run over the list `ord` of (member, step) pairs,
// steps ascending -- step k of every system before step k+1 of any -- so a row still waits for lower tickets only, and the
// launch streams the factors of all systems level by level instead of one latency-bound chain per system and lane.
// A member whose refinement has converged is skipped row by row (nobody waits for its rows).
template <int TRANSPOSE>
__global__ __launch_bounds__(256) void trsv_batched_kernel(const SolveMember* __restrict__ mem, const int* __restrict__ ord, int total,
                                                            TrsvCtl* __restrict__ ctl, unsigned* __restrict__ err_total,
                                                            int first_solve, int accumulate) {
    extern __shared__ __attribute__((aligned(16))) float tile[];        // [128][TLD]
    __shared__ double vec[NB], part[NB];
    __shared__ unsigned s_ticket, s_ok;
    const int tid = threadIdx.x;
    while (true) {
        __syncthreads();
        if (tid == 0) {
            s_ticket = atomicAdd(&ctl->ticket, 1u);
            s_ok = 1u;
        }
        __syncthreads();
        const unsigned t = s_ticket;
        if (t >= (unsigned)total) break;
        const int e = ord[t];
        const SolveMember* mb = mem + (e >> 12);
        const int tk = e & 4095;
        if (!first_solve && mb->st->conv != 0) continue;             // this system needs no further correction
        double* in = TRANSPOSE ? mb->fwd : mb->rhs;
        double* out = TRANSPOSE ? mb->rhs : mb->fwd;
        if (!trsv_row<TRANSPOSE, false>(mb->S, mb->ld, mb->tinv, mb->mpb, tk, in, out, ctl, err_total, 0, TRANSPOSE ? mb->z : (double*)nullptr,
                                 mb->m, accumulate, tile, vec, part, &s_ok))
            break;
    }
    if (tid == 0) trsv_leave(ctl);
}

// rhs <- src padded with zeros to mp, fwd <- the "not yet published" pattern; resets the solve's convergence state
__global__ __launch_bounds__(256) void solve_prep_kernel(const double* __restrict__ src, int64_t m, int64_t mp, double* __restrict__ rhs,
                                                          double* __restrict__ fwd, SolveState* __restrict__ st) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t g0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int64_t i = g0; i < mp; i += stride) {
        rhs[i] = i < m ? src[i] : 0.0;
        reinterpret_cast<unsigned long long*>(fwd)[i] = kTrsvEmpty;
    }
    if (st != nullptr && g0 == 0) {
        st->conv = 0;
        st->computed = 0;
    }
}

// |r_k|^2 (and |d|^2 at k = 0) in one block, fixed order; sets the convergence flag when |r_k| <= tol |d|
// final: this is the check behind the LAST allowed correction -- a solve that is still above its tolerance here is counted in
// the handle's status words (info[4], info[5]); tol2 = 0 ("run every round") never counts
__global__ __launch_bounds__(1024) void resid_check_kernel(const double* __restrict__ r, const double* __restrict__ d, int64_t m, int k,
                                                            double tol2, SolveState* __restrict__ st, int final, int* __restrict__ info) {
    __shared__ double sr[1024], sd[1024];
    if (st->conv != 0) return;
    double a = 0.0, b = 0.0;
    for (int64_t i = threadIdx.x; i < m; i += 1024) {
        a += r[i] * r[i];
        if (k == 0) b += d[i] * d[i];
    }
    sr[threadIdx.x] = a;
    sd[threadIdx.x] = b;
    __syncthreads();
    for (int q = 512; q > 0; q >>= 1) {
        if ((int)threadIdx.x < q) {
            sr[threadIdx.x] += sr[threadIdx.x + q];
            sd[threadIdx.x] += sd[threadIdx.x + q];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (k == 0) st->dd = sd[0];
        st->norm[k] = sr[0];
        st->computed = k + 1;
        if (sr[0] <= tol2 * st->dd) st->conv = 1;
        else if (final && tol2 > 0.0) atomicAdd(&info[kInfoUnconverged], 1);
    }
}

// ---- posterior diagnostics: rows of X <- X L^-T, then row norms -----------------------------------
// rows [i0, i0+nrows) of (H B)^T: X[r][a] = sig_{i0+r} * osig_a * C(i0+r, a), zero in the padding columns
__global__ __launch_bounds__(256) void cross_cov_rows_kernel(const double* __restrict__ gxyz, const double* __restrict__ gsig,
                                                              int64_t n, int64_t i0, int64_t nrows, const double* __restrict__ oxyz,
                                                              const double* __restrict__ osig, int64_t m, int64_t mp, float g2,
                                                              float* __restrict__ X, int64_t ldx) {
    // block = 64 rows x 64 columns tile (like cov_build): blockIdx.x = column tile, blockIdx.y = row tile
    __shared__ float4 pr[64], pc[64];
    const int t = threadIdx.x;
    const int64_t r0 = (int64_t)blockIdx.y * 64, c0 = (int64_t)blockIdx.x * 64;
    if (t < 64) {
        const int64_t cell = i0 + r0 + t;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r0 + t < nrows && cell < n) v = make_float4((float)gxyz[cell], (float)gxyz[n + cell], (float)gxyz[2 * n + cell], (float)gsig[cell]);
        pr[t] = v;
    } else if (t < 128) {
        const int64_t a = c0 + (t - 64);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (a < m) v = make_float4((float)oxyz[a], (float)oxyz[m + a], (float)oxyz[2 * m + a], (float)osig[a]);
        pc[t - 64] = v;
    }
    __syncthreads();
    const int cx = (t & 15) * 4, ry = t >> 4;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const int r = ry + rr * 16;
        if (r0 + r >= nrows) continue;
        const float4 a = pr[r];
        float o[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float4 q = pc[cx + c];
            const float dx = a.x - q.x, dy = a.y - q.y, dz = a.z - q.z;
            o[c] = (c0 + cx + c < m) ? a.w * q.w * __builtin_amdgcn_exp2f(-g2 * (dx * dx + dy * dy + dz * dz)) : 0.f;
        }
        *reinterpret_cast<float4*>(&X[(r0 + r) * ldx + c0 + cx]) = make_float4(o[0], o[1], o[2], o[3]);
    }
}

// X = rows [a0, a0+nrows) of the identity (padded to mp columns)
__global__ __launch_bounds__(256) void identity_rows_kernel(float* __restrict__ X, int64_t ldx, int64_t a0, int64_t nrows, int64_t mp) {
    const int64_t total = nrows * mp;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += stride) {
        const int64_t r = p / mp, c = p % mp;
        X[r * ldx + c] = (c == a0 + r) ? 1.f : 0.f;
    }
}

// out[r] = sum_c X[r][c]^2 in double; one wave per row, fixed shuffle tree
__global__ __launch_bounds__(256) void row_sumsq_kernel(const float* __restrict__ X, int64_t nrows, int64_t ncols, int64_t ldx,
                                                         double* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (row >= nrows) return;
    const float* x = X + row * ldx;
    double s = 0.0;
    for (int64_t c = lane * 4; c < ncols; c += 256) {
        const float4 v = *reinterpret_cast<const float4*>(x + c);
        s += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
    }
#pragma unroll
    for (int k = 32; k >= 1; k >>= 1) s += __shfl_xor(s, k, kWave);
    if (lane == 0) out[row] = s;
}

__global__ __launch_bounds__(256) void post_err_kernel(const double* __restrict__ gsig, int64_t i0, int64_t nrows, int64_t n,
                                                        const double* __restrict__ ss, float* __restrict__ err) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nrows || i0 + r >= n) return;
    const double v = gsig[i0 + r] * gsig[i0 + r] - ss[r];
    err[r] = (float)sqrt(v > 0.0 ? v : 0.0);
}

__global__ __launch_bounds__(256) void gain_diag_kernel(const double* __restrict__ ovar, int64_t a0, int64_t nrows, int64_t m,
                                                         const double* __restrict__ ss, double* __restrict__ ak) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nrows || a0 + r >= m) return;
    ak[a0 + r] = 1.0 - ovar[a0 + r] * ss[r];         // diag(K H) at the observation: 1 - R_aa (S^-1)_aa
}

static const int kBigK = 2048;                           // K from which gemm_nt_big_kernel is used (below it the persistent kernel with the C prefetch wins)

// grid of the persistent gemm_nt_kernel: every tile its own workgroup while they all fit (2 per CU), else 2 per CU
// (a multiple of 8, so that workgroup b keeps its XCD for all of its tiles)
static inline unsigned persistent_grid(const oisat_ctx* h, int64_t virtual_tiles) {
    const int per_cu = h->gemm_wg_per_cu > 0 ? h->gemm_wg_per_cu : 2;
    const int64_t slots = ((int64_t)(h->cu_count > 0 ? h->cu_count : 256) * per_cu) / 8 * 8;
    return (unsigned)(virtual_tiles <= slots || per_cu <= 0 ? virtual_tiles : slots);
}

// ticket block of the dynamic tile walk (gemm_nt_kernel): 9 ints per stream of the handle (main | look-ahead aux), zero
// when a launch starts -- zeroed here when first allocated, by the last workgroup of every launch from then on
static int* dyn_tickets(oisat_ctx* h) {
    const bool fresh = h->ws[8] == nullptr;
    char* base = (char*)oisat_ws(h, 8, 256);
    if (!base) return nullptr;
    if (fresh && hipMemsetAsync(base, 0, 256, h->stream) != hipSuccess) return nullptr;
    return (int*)(base + (h->aux_stream != nullptr && h->stream == h->aux_stream ? 128 : 0));
}

int launch_gemm(oisat_ctx* h, const char* name, float* C, int64_t ldc, const float* A, int64_t lda, const float* B, int64_t ldb,
                int64_t M, int64_t N, int K, int mode, int lower) {
    const int ntm = (int)(M / NB), ntn = (int)(N / NB);
    const int64_t ntiles = lower ? (int64_t)ntn * ntm - (int64_t)ntn * (ntn - 1) / 2 : (int64_t)ntm * ntn;
    if (ntiles <= 0) return OISAT_OK;
    static const bool detail = getenv("OISAT_PROF_DETAIL") && atoi(getenv("OISAT_PROF_DETAIL")) != 0;
    char dname[64];
    if (detail && h->prof) {                                // profiling aid: one record per launch shape
        snprintf(dname, sizeof(dname), "%s K%d t%lld n1", name, K, (long long)ntiles);
        name = dname;
    }
    // too few 128x128 tiles for the 512 workgroup slots: 64x64 tiles, four workgroups per CU; the in-place TRSM form
    // (C aliases A, N == K == 128) takes 64 x 128 tiles (gemm_nt_rows64_kernel).
    constexpr int small_max = 700;
    if (ntiles <= small_max && h->small_tiles && C != A) {
        const int sm = (int)(M / SB), sn = (int)(N / SB);
        const int64_t st = lower ? (int64_t)sn * sm - (int64_t)sn * (sn - 1) / 2 : (int64_t)sm * sn;
        OISAT_LAUNCH(h, name, (gemm_nt_small_kernel<false, 2>), dim3((unsigned)st), dim3(256), 0, C, ldc, A, lda, B, ldb, sm, sn, K, mode, lower,
                     BatchArgs{});
        return OISAT_OK;
    }
    if (ntiles <= small_max && h->small_tiles && C == A && N == NB && !lower) {     // in-place TRSM-as-GEMM
        OISAT_LAUNCH(h, name, (gemm_nt_rows64_kernel<false, 2>), dim3((unsigned)(M / SB)), dim3(256), 0, C, ldc, A, lda, B, ldb, K, mode,
                     BatchArgs{});
        return OISAT_OK;
    }
    if (ntiles >= (int64_t)INT32_MAX) {
        oisat_set_error("gemm grid too large");
        return OISAT_EINVAL;
    }
    if (K >= kBigK) {
        OISAT_LAUNCH(h, name, gemm_nt_big_kernel<false>, dim3((unsigned)ntiles), dim3(256), 0, C, ldc, A, lda, B, ldb, ntm, ntn, K, mode,
                     lower, (int)ntiles, BatchArgs{});
        return OISAT_OK;
    }
    OISAT_LAUNCH(h, name, gemm_nt_kernel<false>, dim3(persistent_grid(h, ntiles)), dim3(256), 0, C, ldc, A, lda, B, ldb, ntm, ntn, K, mode,
                 lower, (int)ntiles, BatchArgs{}, dyn_tickets(h));
    return OISAT_OK;
}

// ---- batched recursion: the same tree as potrf_rec over the block range of the LARGEST matrix; a node applies to the
// matrices that reach it (a prefix of the table), every launch covers all of them ------------------------------------------
static inline int64_t tiles_lower(int64_t ntm, int64_t ntn) { return ntn * ntm - ntn * (ntn - 1) / 2; }

// Leaves of the recursion are PAIRS of block columns (pair_mid_kernel / pair_panel_kernel) where the leaf-level launches
// are bound by HBM traffic, i.e. in lock-step batches of many systems (>= OISAT_LEAF_PAIRS_MIN members, default 8; 0 = never):
// a month's 48 tiles factor in 23.6 ms instead of 25.6.  For one system or a couple of polar caps the same launches are
// latency-bound and the extra one-workgroup step between the two diagonal blocks costs more than the saved pass
// (10,000 observations: 8.6 ms per analysis against 8.0), so those keep single-block leaves.
static const int kLeafPairsMin = 8;                     // lock-step batches of at least this many systems take leaf PAIRS (fewer: 8.6 vs 8.0 ms at 1e4 observations)
static const bool kSinglePairs = false;                  // (single systems keep single-block leaves)
// split point of the node [b0, b1): the left part gets ceil(half), rounded up to an even number of blocks when leaves are
// pairs (so that the tree ends in pairs wherever it can); the one rule of potrf_rec, potrf_rec_batched and the tile tables
static inline int64_t split_mid(int64_t b0, int64_t b1, bool pairs) {
    const int64_t n = b1 - b0;
    int64_t left = (n + 1) / 2;
    if (pairs && n > 2 && (left & 1)) ++left;
    if (left >= n) left = n - 1;
    return b0 + left;
}

// node key of the compact-enumeration tables: kind 0 (b0, mid, b1) / kind 1 (b0)
static inline long long cum_key_of(int kind, int b0, int mid, int b1) {
    return ((long long)kind << 60) | ((long long)b0 << 40) | ((long long)mid << 20) | (long long)b1;
}

// prefix sums of the members' 128x128 tile counts for every node of the recursion tree (the tree of potrf_rec_batched)
static void build_cum_tables(ChBatch* bt, std::vector<int>& host) {
    std::vector<std::pair<long long, int>> keys;            // (key, slot)
    auto add = [&](int kind, int b0, int mid, int b1) {
        const int off = (int)host.size();
        int acc = 0, cnt = 0;
        host.push_back(0);
        for (const BatchMat& m : bt->table) {
            int64_t rows, cols;
            if (kind == 0) { rows = m.mpb - mid; cols = (b1 < m.mpb ? b1 : m.mpb) - mid; }
            else { rows = m.mpb - b0 - 1; cols = 1; }
            if (rows <= 0 || cols <= 0) break;
            acc += (int)(kind == 0 ? cols * rows - cols * (cols - 1) / 2 : rows);
            host.push_back(acc);
            ++cnt;
        }
        if (cnt == 0) { host.resize(off); return; }
        keys.emplace_back(cum_key_of(kind, b0, mid, b1), (int)bt->cum_off.size());
        bt->cum_off.push_back(off);
        bt->cum_cnt.push_back(cnt);
        bt->cum_total.push_back(acc);
    };
    struct Rec {
        static void go(int b0, int b1, const decltype(add)& add, bool pairs) {
            if (b1 - b0 == 1) { add(1, b0, 0, 0); return; }
            if (b1 - b0 == 2 && pairs) return;                  // a leaf pair: its kernels enumerate their own workgroups
            const int mid = (int)split_mid(b0, b1, pairs);
            go(b0, mid, add, pairs);
            add(0, b0, mid, b1);
            go(mid, b1, add, pairs);
        }
    };
    Rec::go(0, bt->max_mpb, add, bt->pairs);
    std::sort(keys.begin(), keys.end());
    std::vector<int> off, cnt, tot;
    for (auto& kv : keys) {
        bt->cum_key.push_back(kv.first);
        off.push_back(bt->cum_off[kv.second]); cnt.push_back(bt->cum_cnt[kv.second]); tot.push_back(bt->cum_total[kv.second]);
    }
    bt->cum_off = off; bt->cum_cnt = cnt; bt->cum_total = tot;
}

static inline void attach_cum(const ChBatch& bt, BatchArgs& ba) {
    if (!bt.cum_dev) return;
    const long long key = cum_key_of(ba.kind, ba.b0, ba.kind == 0 ? ba.mid : 0, ba.kind == 0 ? ba.b1 : 0);
    auto it = std::lower_bound(bt.cum_key.begin(), bt.cum_key.end(), key);
    if (it == bt.cum_key.end() || *it != key) return;
    const size_t s = it - bt.cum_key.begin();
    ba.cum = bt.cum_dev + bt.cum_off[s];
    ba.cnt = bt.cum_cnt[s];
    ba.total = bt.cum_total[s];
}

int launch_gemm_batched(oisat_ctx* h, const char* name, const ChBatch& bt, BatchArgs ba, int K, int mode, int lower) {
    ba.prio = h->wave_prio;
    // participants and tile counts (units of 128) from the host copy of the table
    int cnt = 0;
    int64_t sum = 0, mx = 0;
    for (const BatchMat& m : bt.table) {
        int64_t rows, cols;
        if (ba.kind == 0) { rows = m.mpb - ba.mid; cols = (ba.b1 < m.mpb ? ba.b1 : m.mpb) - ba.mid; }
        else { rows = m.mpb - ba.b0 - 1; cols = 1; }
        if (rows <= 0 || cols <= 0) break;                  // sorted: nobody further down reaches this node either
        const int64_t t = lower ? tiles_lower(rows, cols) : rows * cols;
        sum += t;
        if (t > mx) mx = t;
        ++cnt;
    }
    if (cnt == 0) return OISAT_OK;
    constexpr int small_max = 700;
    // OISAT_PROF_DETAIL=1 (profiling aid): one profile record per launch shape -- "name K tiles members" -- instead of per name
    static const bool detail = getenv("OISAT_PROF_DETAIL") && atoi(getenv("OISAT_PROF_DETAIL")) != 0;
    char dname[64];
    if (detail && h->prof) {
        snprintf(dname, sizeof(dname), "%s K%d t%lld n%d", name, K, (long long)sum, cnt);
        name = dname;
    }
    const BatchMat& big = bt.table[0];
    if (ba.kind == 1) {                                     // in-place TRSM: whole rows per workgroup
        const int64_t rows = big.mpb - ba.b0 - 1;
        if (sum <= small_max) {
            OISAT_LAUNCH(h, name, (gemm_nt_rows64_kernel<true, 1>), dim3((unsigned)(rows * 2), (unsigned)cnt), dim3(256), 0, (float*)nullptr,
                         (int64_t)0, (const float*)nullptr, (int64_t)0, (const float*)nullptr, (int64_t)0, K, mode, ba);
        } else {
            attach_cum(bt, ba);
            OISAT_LAUNCH(h, name, gemm_nt_kernel<true>, dim3(persistent_grid(h, ba.cum ? ba.total : rows * cnt)), dim3(256), 0,
                         (float*)nullptr, (int64_t)0, (const float*)nullptr, (int64_t)0, (const float*)nullptr, (int64_t)0, (int)rows, cnt, K,
                         mode, 0, 0, ba, ba.cum ? dyn_tickets(h) : nullptr);       // tickets only over the compact ids (all real tiles)
        }
        return OISAT_OK;
    }
    const int64_t rows = big.mpb - ba.mid, cols = (ba.b1 < big.mpb ? ba.b1 : big.mpb) - ba.mid;
    if (sum <= small_max) {
        const int64_t st = lower ? tiles_lower(rows * 2, cols * 2) : rows * cols * 4;
        OISAT_LAUNCH(h, name, (gemm_nt_small_kernel<true, 1>), dim3((unsigned)st, (unsigned)cnt), dim3(256), 0, (float*)nullptr, (int64_t)0,
                     (const float*)nullptr, (int64_t)0, (const float*)nullptr, (int64_t)0, 0, 0, K, mode, lower, ba);
    } else if (K >= kBigK) {
        OISAT_LAUNCH(h, name, gemm_nt_big_kernel<true>, dim3((unsigned)mx, (unsigned)cnt), dim3(256), 0, (float*)nullptr, (int64_t)0,
                     (const float*)nullptr, (int64_t)0, (const float*)nullptr, (int64_t)0, 0, 0, K, mode, lower, 0, ba);
    } else {
        if (lower) attach_cum(bt, ba);                          // (the tables hold the lower-triangle tile counts)
        OISAT_LAUNCH(h, name, gemm_nt_kernel<true>, dim3(persistent_grid(h, ba.cum ? ba.total : mx * cnt)), dim3(256), 0, (float*)nullptr,
                     (int64_t)0, (const float*)nullptr, (int64_t)0, (const float*)nullptr, (int64_t)0, (int)mx, cnt, K, mode, lower, 0, ba,
                     ba.cum ? dyn_tickets(h) : nullptr);
    }
    return OISAT_OK;
}

int potrf_rec_batched(oisat_ctx* h, const ChBatch& bt, int b0, int b1, int* info_dev) {
    if (b1 - b0 == 1) {
        int cnt = 0;
        for (const BatchMat& m : bt.table) {
            if (m.mpb <= b0) break;
            ++cnt;
        }
        if (cnt == 0) return OISAT_OK;
        OISAT_LAUNCH(h, "potrf_diag", potrf_diag3_kernel, dim3((unsigned)cnt), dim3(D3_THREADS), 0, (float*)nullptr, (int64_t)0,
                     (int64_t)b0 * NB, (float*)nullptr, info_dev, b0, (const BatchMat*)bt.table_dev, h->wave_prio);
        return launch_gemm_batched(h, "trsm_gemm", bt, BatchArgs{bt.table_dev, 1, b0, 0, 0}, NB, 1, 0);
    }
    if (b1 - b0 == 2 && bt.pairs) {                         // leaf pair (b0, b0 + 1)
        int cnt = 0, maxrows = 0;
        for (const BatchMat& m : bt.table) {
            if (m.mpb <= b0) break;
            ++cnt;
            if (m.mpb - b0 - 2 > maxrows) maxrows = m.mpb - b0 - 2;
        }
        if (cnt == 0) return OISAT_OK;
        const BatchMat* tb = (const BatchMat*)bt.table_dev;
        OISAT_LAUNCH(h, "potrf_diag", potrf_diag3_kernel, dim3((unsigned)cnt), dim3(D3_THREADS), 0, (float*)nullptr, (int64_t)0,
                     (int64_t)b0 * NB, (float*)nullptr, info_dev, b0, tb, h->wave_prio);
        OISAT_LAUNCH(h, "pair_mid", pair_mid_kernel<true>, dim3((unsigned)cnt), dim3(256), 0, (float*)nullptr, (int64_t)0, 0,
                     (const float*)nullptr, b0, tb, h->wave_prio);
        OISAT_LAUNCH(h, "potrf_diag", potrf_diag3_kernel, dim3((unsigned)cnt), dim3(D3_THREADS), 0, (float*)nullptr, (int64_t)0,
                     (int64_t)(b0 + 1) * NB, (float*)nullptr, info_dev, b0 + 1, tb, h->wave_prio);
        if (maxrows > 0) {
            OISAT_LAUNCH(h, "pair_panel", pair_panel_kernel<true>, dim3((unsigned)(maxrows * 2), (unsigned)cnt), dim3(256), 0,
                         (float*)nullptr, (int64_t)0, 0, (const float*)nullptr, b0, tb, h->wave_prio);
        }
        return OISAT_OK;
    }
    const int mid = (int)split_mid(b0, b1, bt.pairs);
    int rc = potrf_rec_batched(h, bt, b0, mid, info_dev);
    if (rc) return rc;
    rc = launch_gemm_batched(h, "syrk_gemm", bt, BatchArgs{bt.table_dev, 0, b0, mid, b1}, (mid - b0) * NB, 0, 1);
    if (rc) return rc;
    return potrf_rec_batched(h, bt, mid, b1, info_dev);
}

__global__ __launch_bounds__(256) void pad_identity_batched_kernel(const BatchMat* __restrict__ mats) {
    const BatchMat m = mats[blockIdx.y];
    const int64_t mp = (int64_t)m.mpb * NB;
    const int64_t total = (mp - m.m) * mp;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += stride) {
        const int64_t r = m.m + p / mp, c = p % mp;
        m.S[r * m.ld + c] = r == c ? 1.f : 0.f;
    }
}

// factor block columns [b0, b1) (units of NB) given that everything to their left is applied
int potrf_rec(oisat_ctx* h, float* S, int64_t ld, int64_t mpb, int64_t b0, int64_t b1, float* tinv, int* info_dev) {
    if (b1 - b0 == 1) {
        const int64_t k0 = b0 * NB;
        OISAT_LAUNCH(h, "potrf_diag", potrf_diag3_kernel, dim3(1), dim3(D3_THREADS), 0, S, ld, k0, tinv, info_dev, (int)b0,
                     (const BatchMat*)nullptr, 0);
        const int64_t rows = (mpb - b0 - 1) * NB;
        if (rows > 0) {
            float* P = S + (k0 + NB) * ld + k0;                  // panel below the diagonal block
            const int rc = launch_gemm(h, "trsm_gemm", P, ld, P, ld, tinv + b0 * NB * NB, NB, rows, NB, NB, 1, 0);
            if (rc) return rc;
        }
        return OISAT_OK;
    }
    if (b1 - b0 == 2 && kSinglePairs) {                     // leaf pair (b0, b0 + 1): OISAT_LEAF_PAIRS_SINGLE=1 (experiments)
        OISAT_LAUNCH(h, "potrf_diag", potrf_diag3_kernel, dim3(1), dim3(D3_THREADS), 0, S, ld, b0 * NB, tinv, info_dev, (int)b0,
                     (const BatchMat*)nullptr, 0);
        OISAT_LAUNCH(h, "pair_mid", pair_mid_kernel<false>, dim3(1), dim3(256), 0, S, ld, (int)mpb, (const float*)tinv, (int)b0,
                     (const BatchMat*)nullptr, 0);
        OISAT_LAUNCH(h, "potrf_diag", potrf_diag3_kernel, dim3(1), dim3(D3_THREADS), 0, S, ld, (b0 + 1) * NB, tinv, info_dev,
                     (int)(b0 + 1), (const BatchMat*)nullptr, 0);
        const int64_t rows = mpb - b0 - 2;
        if (rows > 0) {
            OISAT_LAUNCH(h, "pair_panel", pair_panel_kernel<false>, dim3((unsigned)(rows * 2)), dim3(256), 0, S, ld, (int)mpb,
                         (const float*)tinv, (int)b0, (const BatchMat*)nullptr, 0);
        }
        return OISAT_OK;
    }
    const int64_t mid = split_mid(b0, b1, kSinglePairs);
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

// ---- right-looking Cholesky with look-ahead on two streams (small and mid-size matrices) --------------
// The recursive schedule above is one dependent chain: while the panel work of a column runs (one CU for the
// diagonal block, <= mp/128 tiles for the TRSM) the other 250 CUs idle, and for m ~ 1e4 that chain IS the run
// time.  Here panel j's trailing update is split three ways:
//     A(j):  the columns of panel j+1   -> main stream, right behind panel j (the next panel needs them at once)
//     B1(j): the columns of panel j+2   -> aux stream
//     B2(j): everything further right   -> aux stream (the bulk of the trailing update)
// so the chain  panel(j) -> A(j) -> panel(j+1) ...  only ever waits for B1 of two panels back, and the bulk
// updates fill the idle CUs.  Tiles are never touched by two in-flight launches: the columns of panel c receive
// panel c-1 through A (main), panel c-2 through B1 and panels <= c-3 through B2 (aux, in order), and the main
// stream waits for B1(c-2) before it touches them.  A panel is `pw` diagonal blocks wide: rank-128 updates
// (pw = 1) stream the whole trailing matrix through HBM for 32 flop/B and lose to the chain they were meant to
// hide; pw = 2..4 gives 64..128 flop/B.  Large matrices use the recursive schedule (K = half the matrix).
int potrf_lookahead(oisat_ctx* h, float* S, int64_t ld, int64_t nb, float* tinv, int* info_dev, int64_t pw) {
    // panels of `pw` diagonal blocks: panel q covers block columns [q*pw, min((q+1)*pw, nb)); a panel is factored
    // (all rows below included) by the recursive routine on the main stream, its trailing update has K = pw*128
    const int64_t np = cdiv(nb, pw);
    if (!h->aux_stream) {
        // lowest priority: the bulk updates must not sit in front of the panel chain when a CU slot frees up
        int prio_low = 0, prio_high = 0;
        HIP_TRY(hipDeviceGetStreamPriorityRange(&prio_low, &prio_high));
        const int prio = prio_low;
        // (round 1 also tried keeping CUs off limits for the bulk updates -- a CU-masked stream -- so that the panel chain's
        // one-workgroup kernels always find a home: every masked GEMM loses >= 12 %, profiles/EXPERIMENTS.md)
        HIP_TRY(hipStreamCreateWithPriority(&h->aux_stream, hipStreamNonBlocking, prio));
    }
    while ((int64_t)h->sync_events.size() < 2 * np + 2) {
        hipEvent_t ev;
        HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        h->sync_events.push_back(ev);
    }
    hipStream_t s1 = h->stream, s2 = h->aux_stream;
    auto E1 = [&](int64_t q) { return h->sync_events[2 * q]; };          // panel q ready (main)
    auto E2 = [&](int64_t q) { return h->sync_events[2 * q + 1]; };      // B1(q) done (aux)
    hipEvent_t ev_edge = h->sync_events[2 * np];
    HIP_TRY(hipEventRecord(ev_edge, s1));
    HIP_TRY(hipStreamWaitEvent(s2, ev_edge, 0));                          // S is built before the aux stream touches it
    auto c0 = [&](int64_t q) { return (q * pw < nb ? q * pw : nb); };     // first block column of panel q
    int rc = OISAT_OK;
    // C[r0.., cb0..cb1) -= L[r0.., kb0..kb1) * L[cb0..cb1, kb0..kb1)^T  with r0 == cb0 (anchored on the diagonal)
    auto update = [&](int64_t cb0, int64_t cb1, int64_t kb0, int64_t kb1) {
        if (cb1 <= cb0) return (int)OISAT_OK;
        float* Cc = S + cb0 * NB * ld + cb0 * NB;
        const float* Aa = S + cb0 * NB * ld + kb0 * NB;
        return launch_gemm(h, "syrk_gemm", Cc, ld, Aa, ld, Aa, ld, (nb - cb0) * NB, (cb1 - cb0) * NB, (int)((kb1 - kb0) * NB), 0, 1);
    };
    for (int64_t q = 0; q < np && rc == OISAT_OK; ++q) {
        if (q >= 2) HIP_TRY(hipStreamWaitEvent(s1, E2(q - 2), 0));       // panel q's columns hold panel q-2 (and all earlier)
        rc = potrf_rec(h, S, ld, nb, c0(q), c0(q + 1), tinv, info_dev);  // diag blocks, TRSMs, in-panel updates
        if (rc || q + 1 == np) break;
        HIP_TRY(hipEventRecord(E1(q), s1));
        if (q >= 1) HIP_TRY(hipStreamWaitEvent(s1, E2(q - 1), 0));       // B1(q-1) wrote the same columns
        rc = update(c0(q + 1), c0(q + 2), c0(q), c0(q + 1));              // A(q): the next panel's columns, main stream
        if (rc) break;
        if (q + 2 < np) {
            HIP_TRY(hipStreamWaitEvent(s2, E1(q), 0));
            h->stream = s2;                                               // launches (and their profiling events) go to aux
            rc = update(c0(q + 2), c0(q + 3), c0(q), c0(q + 1));          // B1(q): panel q+2's columns
            if (rc == OISAT_OK && hipEventRecord(E2(q), s2) != hipSuccess) rc = OISAT_EHIP;
            if (rc == OISAT_OK) rc = update(c0(q + 3), nb, c0(q), c0(q + 1));   // B2(q): everything to the right
            h->stream = s1;
        }
    }
    h->stream = s1;
    if (rc) return rc;
    HIP_TRY(hipEventRecord(ev_edge, s2));
    HIP_TRY(hipStreamWaitEvent(s1, ev_edge, 0));                          // join: nothing of the factorization is left in flight
    return OISAT_OK;
}

// Status words of the dense solve, zeroed when first allocated and from then on only by oisat_solve_status(clear):
//   slot 4: int info[8]   = { first non-positive pivot column of the CURRENT factorization (reset by oisat_potrf),
//                             first such column since the last clear, number of failing diagonal blocks since then,
//                             1 + table index of the batch member that met the current one,
//                             [4] gain solves that used every refinement round and still ended above the tolerance,
//                             [5] 1 + member index of the first of them (0: a single-system solve, or none),
//                             [6] task-graph launches that ended on a time-out (their factor is incomplete), - }
//   slot 7: [unsigned err_total (16 B): triangular-solve workgroups that gave up waiting, since the last clear
//            | control block of sweep 0 | control block of sweep 1]
constexpr size_t kCtlBytes = ((sizeof(TrsvCtl) + 15) / 16) * 16;
int status_ws(oisat_ctx* h, int** info_dev, char** trsv_base) {
    const bool fresh4 = h->ws[4] == nullptr, fresh7 = h->ws[7] == nullptr;
    int* info = (int*)oisat_ws(h, 4, 256);
    char* base = (char*)oisat_ws(h, 7, 16 + 2 * kCtlBytes);
    if (!info || !base) return OISAT_ENOMEM;
    if (fresh4) HIP_TRY(hipMemsetAsync(info, 0, 256, h->stream));
    if (fresh7) HIP_TRY(hipMemsetAsync(base, 0, 16 + 2 * kCtlBytes, h->stream));
    if (info_dev) *info_dev = info;
    if (trsv_base) *trsv_base = base;
    return OISAT_OK;
}

__global__ __launch_bounds__(256) void solve_prep_batched_kernel(const SolveMember* __restrict__ mem) {
    const SolveMember* mb = mem + blockIdx.y;
    const int64_t m = mb->m, mp = (int64_t)mb->mpb * NB;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t g0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const double* src = mb->d;
    double* rhs = mb->rhs;
    unsigned long long* fwd = reinterpret_cast<unsigned long long*>(mb->fwd);
    for (int64_t i = g0; i < mp; i += stride) {
        rhs[i] = i < m ? src[i] : 0.0;
        fwd[i] = kTrsvEmpty;
    }
    if (g0 == 0) {
        mb->st->conv = 0;
        mb->st->computed = 0;
    }
}

__global__ __launch_bounds__(1024) void resid_check_batched_kernel(const SolveMember* __restrict__ mem, int k, double tol2, int final,
                                                                    int* __restrict__ info) {
    __shared__ double sr[1024], sd[1024];
    const SolveMember* mb = mem + blockIdx.x;
    SolveState* st = mb->st;
    if (st->conv != 0) return;
    const double* r = mb->rhs;
    const double* d = mb->d;
    const int64_t m = mb->m;
    double a = 0.0, b = 0.0;
    for (int64_t i = threadIdx.x; i < m; i += 1024) {
        a += r[i] * r[i];
        if (k == 0) b += d[i] * d[i];
    }
    sr[threadIdx.x] = a;
    sd[threadIdx.x] = b;
    __syncthreads();
    for (int q = 512; q > 0; q >>= 1) {
        if ((int)threadIdx.x < q) {
            sr[threadIdx.x] += sr[threadIdx.x + q];
            sd[threadIdx.x] += sd[threadIdx.x + q];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (k == 0) st->dd = sd[0];
        st->norm[k] = sr[0];
        st->computed = k + 1;
        if (sr[0] <= tol2 * st->dd) st->conv = 1;
        else if (final && tol2 > 0.0) {
            atomicAdd(&info[kInfoUnconverged], 1);
            atomicCAS(&info[kInfoUnconvergedMember], 0, mb->which + 1);
        }
    }
}

// L L^T x = rhs: forward sweep rhs -> fwd, backward sweep fwd -> rhs (the solution, padded).  On entry `fwd` must hold the
// "not yet published" pattern (solve_prep_kernel, or the previous solve's backward sweep, which leaves it behind); each
// sweep re-arms the vector it has consumed for the sweep that follows, and the last workgroup of a sweep zeroes its
// control block, so a solve is exactly two launches.  st: skip both when the refinement has converged.  zout (m entries):
// where the solution is to be written (accumulate = 0) or added (1) besides rhs.
int trsv_solve(oisat_ctx* h, const ChFactor& f, double* rhs_pad, double* fwd, const SolveState* st, double* zout, int accumulate) {
    const int nb = (int)(f.mp / NB);
    const size_t ctl_bytes = kCtlBytes;
    char* base = nullptr;
    if (int rc = status_ws(h, nullptr, &base)) return rc;
    unsigned* err_total = (unsigned*)base;
    char* ctl = base + 16;
    // a second LDS tile per workgroup (132 KB: one workgroup per CU) when every block row still gets its own CU at once;
    // larger systems keep two workgroups per CU in flight (they are bound by streaming L, not by the hop latency)
    constexpr bool allow_two = true;
    constexpr int rows_per_wg = 1;                       // (four rows per workgroup: 0.54 vs 0.25 ms per sweep at 1e4 observations)
    const int grid = (int)cdiv(nb, rows_per_wg);
    const int two = allow_two && grid <= h->cu_count ? 1 : 0;
    const size_t shm = sizeof(float) * NB * TLD * (two ? 2 : 1);
    OISAT_LAUNCH(h, "trsv_fwd", (trsv_pipe_kernel<0>), dim3(grid), dim3(256), shm, f.S, f.ld, (const float*)f.tinv, nb, rhs_pad, fwd,
                 (TrsvCtl*)ctl, err_total, two, st, (double*)nullptr, (int64_t)0, 0);
    OISAT_LAUNCH(h, "trsv_bwd", (trsv_pipe_kernel<1>), dim3(grid), dim3(256), shm, f.S, f.ld, (const float*)f.tinv, nb, fwd, rhs_pad,
                 (TrsvCtl*)(ctl + ctl_bytes), err_total, two, st, zout, f.m, accumulate);
    return OISAT_OK;
}

SolveState* solve_state(oisat_ctx* h) {
    const bool fresh = h->ws[9] == nullptr;
    SolveState* st = (SolveState*)oisat_ws(h, 9, 256);
    if (st && fresh && hipMemsetAsync(st, 0, 256, h->stream) != hipSuccess) return nullptr;
    return st;
}

// X[nrows x mp] <- X L^-T  by block forward substitution over column blocks [b0, b1)
int trsm_rows_rec(oisat_ctx* h, const ChFactor& f, float* X, int64_t nrows, int64_t ldx, int64_t b0, int64_t b1) {
    if (b1 - b0 == 1) {
        float* Xb = X + b0 * NB;
        return launch_gemm(h, "trsm_rows_diag", Xb, ldx, Xb, ldx, f.tinv + b0 * NB * NB, NB, nrows, NB, NB, 1, 0);
    }
    const int64_t mid = b0 + (b1 - b0 + 1) / 2;
    int rc = trsm_rows_rec(h, f, X, nrows, ldx, b0, mid);
    if (rc) return rc;
    // X[:, mid:b1] -= X[:, b0:mid] * L[mid:b1, b0:mid]^T
    rc = launch_gemm(h, "trsm_rows_gemm", X + mid * NB, ldx, X + b0 * NB, ldx, f.S + mid * NB * f.ld + b0 * NB, f.ld, nrows,
                     (b1 - mid) * NB, (int)((mid - b0) * NB), 0, 0);
    if (rc) return rc;
    return trsm_rows_rec(h, f, X, nrows, ldx, mid, b1);
}

// per-function attributes, set once per process (handles may be driven from different host threads)
hipError_t dense_kernel_attributes() {
    static const hipError_t attr_rc = []() {
        hipError_t e = hipFuncSetAttribute((const void*)trsv_pipe_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)(sizeof(float) * NB * TLD * 2));
        if (e == hipSuccess)
            e = hipFuncSetAttribute((const void*)trsv_pipe_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)(sizeof(float) * NB * TLD * 2));
        if (e == hipSuccess)
            e = hipFuncSetAttribute((const void*)trsv_batched_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)(sizeof(float) * NB * TLD));
        if (e == hipSuccess)
            e = hipFuncSetAttribute((const void*)trsv_batched_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)(sizeof(float) * NB * TLD));
        return e;
    }();
    return attr_rc;
}

#include "dense_solve_dev.inc"
#include "dense_dag.inc"

// Which factorizations run as a task graph: every system / batch of at least three block rows (below that there is nothing to
// overlap), unless the handle says otherwise (oisat_set_task_graph) or the environment does (OISAT_DAG=0 | 1: the parity tests
// compare the two schedules inside one process, so it is read at every call).  Measured against the recursion
// (factorization alone, TFLOP/s): 10,000 observations 91.9 vs 53.9, 20,000: 126 vs 94, 30,000: 135 vs 115, 50,000: 138 vs
// 128, 100,000: 140.8 vs 137.4 -- the task graph wins at every size.  A batch of more than 1024 systems keeps the lock-step
// recursion, and so does any launch whose chains would not leave room for the tasks they wait for (dag_fits).
static inline bool dag_wanted(const oisat_ctx* h, int64_t max_blocks, int nsys) {
    const int mode = h->dag_mode >= 0 ? h->dag_mode : (getenv("OISAT_DAG") ? atoi(getenv("OISAT_DAG")) : -1);
    if (mode == 0 || nsys > 1024) return false;
    return max_blocks >= (mode == 1 ? 2 : 3);
}

}  // namespace

void oisat_dag_plan_release(void* plan) { dag_plan_free((DagPlan*)plan); }

extern "C" int oisat_dag_task_order(int nsys, const int32_t* block_rows, int wave, int32_t* tasks_out, int64_t capacity,
                                    int64_t* ntasks_out, int32_t* reserve_out, int32_t* max_wave_chains_out) {
    ARG_CHECK(nsys > 0 && block_rows && ntasks_out && (tasks_out || capacity == 0));
    std::vector<int> nb_of(block_rows, block_rows + nsys);
    for (int s = 0; s < nsys; ++s) ARG_CHECK(nb_of[s] >= 1 && (s == 0 || nb_of[s] <= nb_of[s - 1]));
    DagOrder order;
    dag_task_order(nb_of, wave, order);
    *ntasks_out = (int64_t)order.tasks.size();
    if (reserve_out) *reserve_out = order.reserve_chains;
    if (max_wave_chains_out) *max_wave_chains_out = order.max_wave_chains;
    if ((int64_t)order.tasks.size() > capacity) return capacity == 0 ? OISAT_OK : OISAT_EINVAL;
    memcpy(tasks_out, order.tasks.data(), sizeof(int4) * order.tasks.size());
    return OISAT_OK;
}

extern "C" int oisat_set_refine_tol(oisat_ctx* h, double tol) {
    ARG_CHECK(h != nullptr && tol >= 0.0 && tol < 1.0);
    h->refine_tol = tol;
    return OISAT_OK;
}

extern "C" int oisat_set_task_graph(oisat_ctx* h, int mode) {
    ARG_CHECK(h != nullptr && mode >= -1 && mode <= 1);
    h->dag_mode = mode;
    return OISAT_OK;
}

extern "C" int oisat_set_share(oisat_ctx* h, int wave_prio, int gemm_wg_per_cu) {
    ARG_CHECK(h != nullptr && wave_prio >= 0 && wave_prio <= 3 && gemm_wg_per_cu >= 0 && gemm_wg_per_cu <= 2);
    h->wave_prio = wave_prio;
    h->gemm_wg_per_cu = gemm_wg_per_cu;
    return OISAT_OK;
}

extern "C" int oisat_gemm_nt(oisat_ctx* h, float* C, int64_t ldc, const float* A, int64_t lda, const float* B, int64_t ldb,
                             int64_t M, int64_t N, int64_t K, int mode, int lower) {
    ARG_CHECK(h && C && A && B && M > 0 && N > 0 && K > 0);
    ARG_CHECK(M % NB == 0 && N % NB == 0 && K % BK == 0 && K < (int64_t)INT32_MAX);
    ARG_CHECK(ldc % 4 == 0 && lda % 4 == 0 && ldb % 4 == 0 && ldc >= N && lda >= K && ldb >= K);
    ARG_CHECK(((uintptr_t)C % 16) == 0 && ((uintptr_t)A % 16) == 0 && ((uintptr_t)B % 16) == 0);
    ARG_CHECK((mode == 0 || mode == 1) && (!lower || M >= N));
    return launch_gemm(h, "gemm_nt", C, ldc, A, lda, B, ldb, M, N, (int)K, mode, lower);
}

extern "C" int oisat_potrf(oisat_ctx* h, float* S, int64_t m, int64_t ld, int* info_host) {
    ARG_CHECK(h && S && m > 0);
    const int64_t mp = cdiv(m, NB) * NB;
    ARG_CHECK(ld >= mp && (ld % 4) == 0 && ((uintptr_t)S % 16) == 0);
    const int64_t mpb = mp / NB;
    float* tinv = (float*)oisat_ws(h, 3, sizeof(float) * mpb * NB * NB);
    int* info_dev = nullptr;
    if (!tinv) return OISAT_ENOMEM;
    if (int rc = status_ws(h, &info_dev, nullptr)) return rc;
    HIP_TRY(dense_kernel_attributes());
    HIP_TRY(hipMemsetAsync(info_dev, 0, sizeof(int), h->stream));      // info[0] only: the sticky words survive
    if (mp > m) {
        OISAT_LAUNCH(h, "pad_identity", pad_identity_kernel, dim3(stream_grid((mp - m) * mp, 256)), dim3(256), 0, S, ld, m, mp);
    }
    // schedule: recursive by default.  The two-stream look-ahead schedule (OISAT_POTRF=lookahead[:panel blocks]) is
    // kept selectable: at m = 1e4 it ties the recursive one in every round (round 1: 12.7 ms per analysis either way,
    // the 137 KB diagonal kernel waiting 300-350 us for an empty CU under the bulk updates; round 2 with the 26 KB
    // register-resident kernel: 7.75-7.80 ms at panel widths 6..20 against 7.84 ms) -- its rank-256..1024 bulk updates
    // cost 5.4-6.3 ms where the recursion's GEMMs cost 4.1.  See DESIGN.md section 4.
    bool lookahead = false;
    int64_t pw = 4;
    if (const char* env = getenv("OISAT_POTRF")) {
        if (strcmp(env, "recursive") == 0) lookahead = false;
        else if (strncmp(env, "lookahead", 9) == 0) {
            lookahead = mpb >= 2;
            if (env[9] == ':' && atoi(env + 10) > 0) pw = atoi(env + 10);      // OISAT_POTRF=lookahead:4
        }
    }
    int rc;
    if (!lookahead && !getenv("OISAT_POTRF") && dag_wanted(h, mpb, 1) && dag_fits(1, dag_slots(h))) {
        // the plan of this (S, tinv, ld, block rows) -- a handle keeps the last few (a lane that factors its tiles one after
        // the other in ONE shared buffer meets the same few sizes month after month)
        DagSingle* hit = nullptr;
        for (DagSingle& c : h->dag_cache)
            if (c.plan && c.S == S && c.tinv == tinv && c.ld == ld && c.mpb == mpb) hit = &c;
        if (!hit) {
            DagSingle* slot = nullptr;
            for (DagSingle& c : h->dag_cache)
                if (!c.plan) { slot = &c; break; }
            if (!slot) {                                         // evict the least recently used
                slot = &h->dag_cache[0];
                for (DagSingle& c : h->dag_cache)
                    if (c.stamp < slot->stamp) slot = &c;
                HIP_TRY(hipStreamSynchronize(h->stream));        // (a plan is never freed under a running launch)
                oisat_dag_plan_release(slot->plan);
                slot->plan = nullptr;
            }
            slot->plan = dag_plan_create(std::vector<BatchMat>{BatchMat{S, tinv, ld, m, (int)mpb, 0}}, h->stream);
            if (!slot->plan) return OISAT_ENOMEM;
            slot->S = S; slot->tinv = tinv; slot->ld = ld; slot->mpb = mpb;
            hit = slot;
        }
        hit->stamp = ++h->dag_clock;
        rc = dag_launch(h, *(DagPlan*)hit->plan, info_dev, (unsigned*)(info_dev + kInfoDagTimeouts));
    } else {
        rc = lookahead ? potrf_lookahead(h, S, ld, mpb, tinv, info_dev, pw) : potrf_rec(h, S, ld, mpb, 0, mpb, tinv, info_dev);
    }
    if (rc) return rc;
    h->factor.S = S;
    h->factor.m = m;
    h->factor.mp = mp;
    h->factor.ld = ld;
    h->factor.tinv = tinv;
    if (info_host) {
        int* pin = (int*)oisat_pinned(h, 64);
        if (!pin) return OISAT_ENOMEM;
        HIP_TRY(hipMemcpyAsync(pin, info_dev, 8 * sizeof(int), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
        *info_host = *pin;
        if (pin[kInfoDagTimeouts] != 0) {                       // reported here, not again by oisat_solve_status
            HIP_TRY(hipMemsetAsync(info_dev + kInfoDagTimeouts, 0, sizeof(int), h->stream));
            oisat_set_error("potrf: the task-graph factorization timed out (a workgroup gave up waiting): the factor is incomplete");
            return OISAT_EHIP;
        }
        if (*pin != 0) {
            // reported to the caller right here: do not report it a second time through oisat_solve_status
            HIP_TRY(hipMemsetAsync(info_dev + 1, 0, 2 * sizeof(int), h->stream));
            oisat_set_error("potrf: matrix not positive definite at column %d", *pin);
            return OISAT_ENOTPD;
        }
    }
    return OISAT_OK;
}

extern "C" int oisat_potrs(oisat_ctx* h, const float* L, int64_t m, int64_t ld, double* z_inout) {
    ARG_CHECK(h && L && z_inout && m > 0);
    ARG_CHECK(h->factor.S == L && h->factor.m == m && h->factor.ld == ld);     // must follow oisat_potrf of this matrix
    double* w = (double*)oisat_ws(h, 5, sizeof(double) * (2 + 8) * h->factor.mp);
    if (!w) return OISAT_ENOMEM;
    double* rhs = w;
    double* fwd = w + h->factor.mp;
    OISAT_LAUNCH(h, "copy_pad", solve_prep_kernel, dim3(stream_grid(h->factor.mp, 256)), dim3(256), 0, (const double*)z_inout, m,
                 h->factor.mp, rhs, fwd, (SolveState*)nullptr);
    return trsv_solve(h, h->factor, rhs, fwd, nullptr, z_inout, 0);
}

int oisat_cov_residual_if(oisat_ctx* h, const double* oxyz, const double* osig, const double* ovar, int64_t m, double g,
                          const double* d, const double* z, double* r_out, const double* olat_sorted, const int* converged_dev,
                          const int* perm);
const int* oisat_take_obs_perm(oisat_ctx* h, int64_t m);

// z = S^-1 d through the fp32 factor as a preconditioner:  z <- M^-1 d;  repeat { r = d - S z (float64, S regenerated from
// coordinates);  stop if |r| <= tol |d|;  z <- z + M^-1 r }  at most `refine` times.  tol = 1e-6 (oisat_set_refine_tol; env OISAT_REFINE_TOL at init): the
// analysis increment is K r away from the exact one and |K| <= 1, so the fields are then 1e-6 |d| from the float64 answer,
// ten times inside the 1e-5 bar; a factor that is a good preconditioner (|r_0| / |d| = 2e-6 .. 2e-5 on the BASELINE
// workloads, 1e-9 .. 1e-10 after one correction) stops after one correction, a poor one gets all `refine` of them.  The
// test runs on the device: resid_check_kernel sets a flag and the launches of the rounds that follow return at once, so
// nothing waits for the host.  Launches: prep, 2 sweeps, then per round residual (written straight into the padded right-
// hand side), check, 2 sweeps (the backward one adds its solution to z): no fills, no copies.
extern "C" int oisat_gain_solve(oisat_ctx* h, const float* L, const double* oxyz, const double* osig, const double* ovar, int64_t m,
                                int64_t ld, double g, const double* d, int refine, double* z_out, double* resid_host,
                                const double* olat_sorted) {
    ARG_CHECK(h != nullptr);
    const int* perm = oisat_take_obs_perm(h, m);               // one-shot, whatever this call's outcome
    ARG_CHECK(L && oxyz && osig && ovar && d && z_out && m > 0 && refine >= 0 && refine <= 8);
    ARG_CHECK(h->factor.S == L && h->factor.m == m && h->factor.ld == ld);
    const double tol = h->refine_tol;
    const int64_t mp = h->factor.mp;
    double* w = (double*)oisat_ws(h, 5, sizeof(double) * (2 + 8) * mp);
    SolveState* st = solve_state(h);
    if (!w || !st) return OISAT_ENOMEM;
    double* rhs = w;
    double* fwd = w + mp;
    OISAT_LAUNCH(h, "copy_pad", solve_prep_kernel, dim3(stream_grid(mp, 256)), dim3(256), 0, d, m, mp, rhs, fwd, st);
    int rc = trsv_solve(h, h->factor, rhs, fwd, nullptr, z_out, 0);
    if (rc) return rc;
    // the residual is evaluated once more behind the last allowed correction: a converged solve skips it (no-op launches), one
    // that is still above the tolerance there is recorded in the handle's status (oisat_solve_status_ex)
    int* info_dev = nullptr;
    if (int rs = status_ws(h, &info_dev, nullptr)) return rs;
    // (refine = 0 asks for the plain solve: no residual is formed -- unless the caller wants it reported -- and nothing is flagged)
    for (int it = 0; it <= refine && (refine > 0 || resid_host); ++it) {
        rc = oisat_cov_residual_if(h, oxyz, osig, ovar, m, g, d, z_out, rhs, olat_sorted, &st->conv, perm);
        if (rc) return rc;
        OISAT_LAUNCH(h, "resid_check", resid_check_kernel, dim3(1), dim3(1024), 0, (const double*)rhs, d, m, it, tol * tol, st,
                     it == refine && refine > 0 ? 1 : 0, info_dev);
        if (it == refine) break;
        rc = trsv_solve(h, h->factor, rhs, fwd, st, z_out, 1);
        if (rc) return rc;
    }
    if (resid_host) {
        char* pin = (char*)oisat_pinned(h, 512);
        if (!pin) return OISAT_ENOMEM;
        HIP_TRY(hipMemcpyAsync(pin, st, sizeof(SolveState), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipMemcpyAsync(pin + 256, h->ws[7], sizeof(unsigned), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
        const SolveState* hs = (const SolveState*)pin;
        const double dn = hs->dd > 0.0 ? sqrt(hs->dd) : 1.0;
        for (int it = 0; it <= refine; ++it) {                  // rounds that were skipped after convergence repeat the last residual
            const int k = it < hs->computed ? it : hs->computed - 1;
            resid_host[it] = k >= 0 ? sqrt(hs->norm[k]) / dn : 0.0;
        }
        const unsigned gave_up = *(const unsigned*)(pin + 256);
        if (gave_up != 0) {
            HIP_TRY(hipMemsetAsync(h->ws[7], 0, 16, h->stream));          // reported here, not again by oisat_solve_status
            oisat_set_error("triangular solve: %u workgroup(s) gave up waiting for a predecessor (bounded spin)", gave_up);
            return OISAT_EHIP;
        }
    }
    return OISAT_OK;
}

extern "C" int oisat_trsm_rows(oisat_ctx* h, const float* L, int64_t m, int64_t ld, float* X, int64_t nrows, int64_t ldx) {
    ARG_CHECK(h && L && X && m > 0 && nrows > 0);
    ARG_CHECK(h->factor.S == L && h->factor.m == m && h->factor.ld == ld);     // must follow oisat_potrf of this matrix
    ARG_CHECK(nrows % NB == 0 && ldx >= h->factor.mp && ldx % 4 == 0 && ((uintptr_t)X % 16) == 0);
    return trsm_rows_rec(h, h->factor, X, nrows, ldx, 0, h->factor.mp / NB);
}

extern "C" int oisat_posterior_error(oisat_ctx* h, const float* L, int64_t m, int64_t ld, const double* gxyz, const double* gsig,
                                     int64_t n, int64_t i0, int64_t i1, const double* oxyz, const double* osig, double g,
                                     int64_t chunk_rows, float* err) {
    ARG_CHECK(h && L && gxyz && gsig && oxyz && osig && err && m > 0 && n > 0 && 0 <= i0 && i0 < i1 && i1 <= n);
    ARG_CHECK(h->factor.S == L && h->factor.m == m && h->factor.ld == ld);
    const int64_t mp = h->factor.mp;
    if (chunk_rows <= 0) chunk_rows = 4096;
    chunk_rows = cdiv(chunk_rows, NB) * NB;
    float* X = (float*)oisat_ws(h, 6, sizeof(float) * chunk_rows * mp + sizeof(double) * chunk_rows);
    if (!X) return OISAT_ENOMEM;
    double* ss = (double*)(X + chunk_rows * mp);
    const float g2 = (float)(g * 1.4426950408889634);
    for (int64_t c0 = i0; c0 < i1; c0 += chunk_rows) {
        const int64_t live = (i1 - c0 < chunk_rows) ? i1 - c0 : chunk_rows;
        const int64_t nrows = cdiv(live, NB) * NB;                 // rows live..nrows are zero rows
        OISAT_LAUNCH(h, "cross_cov_rows", cross_cov_rows_kernel, dim3((unsigned)(mp / 64), (unsigned)(nrows / 64)), dim3(256), 0,
                     gxyz, gsig, n, c0, live, oxyz, osig, m, mp, g2, X, mp);
        if (nrows > live) HIP_TRY(hipMemsetAsync(X + live * mp, 0, sizeof(float) * (nrows - live) * mp, h->stream));
        const int rc = trsm_rows_rec(h, h->factor, X, nrows, mp, 0, mp / NB);
        if (rc) return rc;
        OISAT_LAUNCH(h, "row_sumsq", row_sumsq_kernel, dim3((unsigned)cdiv(nrows * 64, 256)), dim3(256), 0, (const float*)X, nrows, mp,
                     mp, ss);
        OISAT_LAUNCH(h, "post_err", post_err_kernel, dim3((unsigned)cdiv(live, 256)), dim3(256), 0, gsig, c0, live, n,
                     (const double*)ss, err + (c0 - i0));
    }
    return OISAT_OK;
}

extern "C" int oisat_gain_diag(oisat_ctx* h, const float* L, int64_t m, int64_t ld, const double* ovar, int64_t chunk_rows,
                               double* ak_out) {
    ARG_CHECK(h && L && ovar && ak_out && m > 0);
    ARG_CHECK(h->factor.S == L && h->factor.m == m && h->factor.ld == ld);
    const int64_t mp = h->factor.mp;
    if (chunk_rows <= 0) chunk_rows = 4096;
    chunk_rows = cdiv(chunk_rows, NB) * NB;
    float* X = (float*)oisat_ws(h, 6, sizeof(float) * chunk_rows * mp + sizeof(double) * chunk_rows);
    if (!X) return OISAT_ENOMEM;
    double* ss = (double*)(X + chunk_rows * mp);
    for (int64_t a0 = 0; a0 < m; a0 += chunk_rows) {
        const int64_t live = (m - a0 < chunk_rows) ? m - a0 : chunk_rows;
        const int64_t nrows = cdiv(live, NB) * NB;
        OISAT_LAUNCH(h, "identity_rows", identity_rows_kernel, dim3(stream_grid(nrows * mp, 256)), dim3(256), 0, X, mp, a0, nrows, mp);
        const int rc = trsm_rows_rec(h, h->factor, X, nrows, mp, 0, mp / NB);
        if (rc) return rc;
        OISAT_LAUNCH(h, "row_sumsq", row_sumsq_kernel, dim3((unsigned)cdiv(nrows * 64, 256)), dim3(256), 0, (const float*)X, nrows, mp,
                     mp, ss);
        OISAT_LAUNCH(h, "gain_diag", gain_diag_kernel, dim3((unsigned)cdiv(live, 256)), dim3(256), 0, ovar, a0, live, m,
                     (const double*)ss, ak_out);
    }
    return OISAT_OK;
}

// words of oisat_solve_status_ex (include/oisat.h: OISAT_STATUS_*)
extern "C" int oisat_solve_status_ex(oisat_ctx* h, int32_t* out, int nwords, int clear) {
    ARG_CHECK(h != nullptr && (out != nullptr || nwords == 0) && nwords >= 0 && nwords <= OISAT_STATUS_WORDS);
    int* info = nullptr;
    char* base = nullptr;
    if (int rc = status_ws(h, &info, &base)) return rc;
    int* pin = (int*)oisat_pinned(h, 256);
    if (!pin) return OISAT_ENOMEM;
    HIP_TRY(hipMemcpyAsync(pin, info, 8 * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipMemcpyAsync(pin + 8, base, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    if (clear) {
        HIP_TRY(hipMemsetAsync(info + 1, 0, 2 * sizeof(int), h->stream));
        HIP_TRY(hipMemsetAsync(info + kInfoUnconverged, 0, 3 * sizeof(int), h->stream));
        HIP_TRY(hipMemsetAsync(base, 0, 16, h->stream));
    }
    HIP_TRY(hipStreamSynchronize(h->stream));
    const int32_t words[OISAT_STATUS_WORDS] = {pin[1], pin[2], pin[8], pin[kInfoUnconverged], pin[kInfoUnconvergedMember] - 1,
                                               pin[kInfoDagTimeouts]};
    for (int i = 0; i < nwords; ++i) out[i] = words[i];
    return OISAT_OK;
}

extern "C" int oisat_solve_status(oisat_ctx* h, int* first_notpd_col, int* n_notpd_blocks, int* trsv_timeouts, int clear) {
    int32_t w[OISAT_STATUS_WORDS];
    // (kept for callers of the round-2 ABI: the three words it knows; a clear resets all six, so such a caller folds the
    // others into the time-out count rather than lose them)
    if (int rc = oisat_solve_status_ex(h, w, OISAT_STATUS_WORDS, clear)) return rc;
    if (first_notpd_col) *first_notpd_col = w[OISAT_STATUS_NOTPD_COL];
    if (n_notpd_blocks) *n_notpd_blocks = w[OISAT_STATUS_NOTPD_BLOCKS];
    if (trsv_timeouts) *trsv_timeouts = w[OISAT_STATUS_TRSV_TIMEOUTS] + w[OISAT_STATUS_DAG_TIMEOUTS] + w[OISAT_STATUS_UNCONVERGED];
    return OISAT_OK;
}

extern "C" int oisat_dense_reserve(oisat_ctx* h, int64_t max_obs, int64_t diag_chunk_rows) {
    ARG_CHECK(h != nullptr && max_obs > 0 && diag_chunk_rows >= 0);
    const int64_t mp = cdiv(max_obs, NB) * NB;
    if (!oisat_ws(h, 3, sizeof(float) * mp * NB)) return OISAT_ENOMEM;                  // inverted diagonal blocks
    if (int rc = status_ws(h, nullptr, nullptr)) return rc;                              // slots 4 and 7
    if (!oisat_ws(h, 5, sizeof(double) * (2 + 8) * mp)) return OISAT_ENOMEM;                   // padded rhs + forward solution
    if (!solve_state(h)) return OISAT_ENOMEM;                                            // slot 9: convergence state of the gain solve
    size_t s6 = sizeof(double) * (max_obs + 16);                                         // refinement residual
    if (diag_chunk_rows > 0) {
        const int64_t ch = cdiv(diag_chunk_rows, NB) * NB;
        const size_t x = sizeof(float) * ch * mp + sizeof(double) * ch;                 // posterior-error / gain-diag rows
        if (x > s6) s6 = x;
    }
    if (!oisat_ws(h, 6, s6)) return OISAT_ENOMEM;
    if (!oisat_pinned(h, 4096)) return OISAT_ENOMEM;
    return OISAT_OK;
}

// ---- batched factorization ---------------------------------------------------------------------------------------
extern "C" int oisat_batch_create(oisat_ctx* h, int nmat, float* const* S, const int64_t* m, const int64_t* ld, float* const* tinv,
                                  int* batch_id_out) {
    ARG_CHECK(h && S && m && ld && tinv && batch_id_out && nmat > 0 && nmat <= 65535);
    ChBatch* bt = new ChBatch();
    bt->order.resize(nmat);
    for (int i = 0; i < nmat; ++i) bt->order[i] = i;
    for (int i = 0; i < nmat; ++i) {
        const int64_t mp = cdiv(m[i], NB) * NB;
        if (!(S[i] && tinv[i] && m[i] > 0 && ld[i] >= mp && (ld[i] % 4) == 0 && ((uintptr_t)S[i] % 16) == 0 && ((uintptr_t)tinv[i] % 16) == 0 &&
              mp / NB < (int64_t)INT32_MAX)) {
            delete bt;
            oisat_set_error("oisat_batch_create: matrix %d: bad pointer, size or leading dimension", i);
            return OISAT_EINVAL;
        }
    }
    // largest first (stable): the matrices a node of the recursion applies to are then a prefix of the table
    std::stable_sort(bt->order.begin(), bt->order.end(), [&](int a, int b) { return cdiv(m[a], NB) > cdiv(m[b], NB); });
    bt->table.resize(nmat);
    for (int i = 0; i < nmat; ++i) {
        const int k = bt->order[i];
        bt->table[i] = BatchMat{S[k], tinv[k], ld[k], m[k], (int)cdiv(m[k], NB), 0};
    }
    bt->max_mpb = bt->table[0].mpb;
    bt->pairs = kLeafPairsMin > 0 && nmat >= kLeafPairsMin;
    if (hipMalloc(&bt->table_dev, sizeof(BatchMat) * nmat) != hipSuccess) {
        delete bt;
        oisat_set_error("oisat_batch_create: hipMalloc of the table failed");
        return OISAT_ENOMEM;
    }
    if (hipMemcpy(bt->table_dev, bt->table.data(), sizeof(BatchMat) * nmat, hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipFree(bt->table_dev);
        delete bt;
        oisat_set_error("oisat_batch_create: upload of the table failed");
        return OISAT_EHIP;
    }
    {
        std::vector<int> host;
        build_cum_tables(bt, host);
        if (hipMalloc(&bt->cum_dev, sizeof(int) * host.size()) != hipSuccess ||
            hipMemcpy(bt->cum_dev, host.data(), sizeof(int) * host.size(), hipMemcpyHostToDevice) != hipSuccess) {
            if (bt->cum_dev) (void)hipFree(bt->cum_dev);
            (void)hipFree(bt->table_dev);
            delete bt;
            oisat_set_error("oisat_batch_create: tile-enumeration tables failed");
            return OISAT_ENOMEM;
        }
    }
    bool want_dag = dag_wanted(h, bt->max_mpb, nmat);
    if (want_dag) {                                          // would its chains crowd out the tasks they wait for?  (host-only check)
        std::vector<int> nb_of(nmat);
        for (int i = 0; i < nmat; ++i) nb_of[i] = bt->table[i].mpb;
        DagOrder probe;
        dag_task_order(nb_of, 0, probe);
        want_dag = dag_fits(probe.max_wave_chains, dag_slots(h));        // no: the batch keeps the lock-step recursion
    }
    if (want_dag) {
        bt->dag = dag_plan_create(bt->table, h->stream);
        if (bt->dag && hipStreamSynchronize(h->stream) != hipSuccess) {       // (the plan's words are zero before any stream can launch it)
            oisat_dag_plan_release(bt->dag);
            bt->dag = nullptr;
        }
        if (!bt->dag) {
            (void)hipFree(bt->cum_dev);
            (void)hipFree(bt->table_dev);
            delete bt;
            return OISAT_ENOMEM;
        }
    }
    int id = -1;
    for (size_t i = 0; i < h->batches.size(); ++i)
        if (!h->batches[i]) { id = (int)i; break; }
    if (id < 0) { h->batches.push_back(nullptr); id = (int)h->batches.size() - 1; }
    h->batches[id] = bt;
    *batch_id_out = id;
    return OISAT_OK;
}

extern "C" int oisat_batch_destroy(oisat_ctx* h, int batch_id) {
    ARG_CHECK(h && batch_id >= 0 && batch_id < (int)h->batches.size() && h->batches[batch_id]);
    HIP_TRY(hipStreamSynchronize(h->stream));
    ChBatch* bt = h->batches[batch_id];
    if (bt->table_dev) HIP_TRY(hipFree(bt->table_dev));
    if (bt->cum_dev) HIP_TRY(hipFree(bt->cum_dev));
    if (bt->solve_dev) HIP_TRY(hipFree(bt->solve_dev));
    if (bt->ord_dev) HIP_TRY(hipFree(bt->ord_dev));
    if (bt->ctl_dev) HIP_TRY(hipFree(bt->ctl_dev));
    oisat_dag_plan_release(bt->dag);
    oisat_dag_plan_release(bt->dag_solve);
    delete bt;
    h->batches[batch_id] = nullptr;
    return OISAT_OK;
}

int oisat_cov_residual_batched(oisat_ctx* h, const SolveMember* mem_dev, const std::vector<SolveMember>& mem_host, int64_t max_m, double g);
int oisat_apply_increment_batched(oisat_ctx* h, int dtype, const SolveMember* mem_dev, const std::vector<SolveMember>& mem_host, int64_t max_n,
                                  double g);

extern "C" int oisat_batch_set_solve(oisat_ctx* h, int batch_id, int nmat, const double* const* oxyz, const double* const* osig,
                                     const double* const* ovar, const double* const* d, const double* const* olat, double* const* z,
                                     double* const* work, void* const* state, const double* const* gxyz, const double* const* gsig,
                                     const double* const* glat, const int64_t* n, const void* const* xb, void* const* xa,
                                     void* const* inc) {
    ARG_CHECK(h && batch_id >= 0 && batch_id < (int)h->batches.size() && h->batches[batch_id]);
    ChBatch* bt = h->batches[batch_id];
    ARG_CHECK(nmat == (int)bt->table.size());
    ARG_CHECK(oxyz && osig && ovar && d && olat && z && work && state && gxyz && gsig && glat && n && xb && xa && inc);
    std::vector<SolveMember> mem(nmat);
    std::vector<int> ord;
    bt->max_m = bt->max_n = bt->max_mp = 0;
    for (int i = 0; i < nmat; ++i) {                         // table order (largest first); order[i] = the caller's index
        const int c = bt->order[i];
        const BatchMat& bm = bt->table[i];
        ARG_CHECK(oxyz[c] && osig[c] && ovar[c] && d[c] && olat[c] && z[c] && work[c] && state[c] && gxyz[c] && gsig[c] && glat[c]);
        ARG_CHECK(n[c] > 0 && xb[c] && (xa[c] || inc[c]) && bm.mpb < 4096);
        SolveMember& sm = mem[i];
        sm.S = bm.S; sm.tinv = bm.tinv; sm.ld = bm.ld; sm.m = bm.m; sm.mpb = bm.mpb; sm.nx = 0;
        sm.oxyz = oxyz[c]; sm.osig = osig[c]; sm.ovar = ovar[c]; sm.d = d[c]; sm.olat = olat[c];
        sm.z = z[c]; sm.rhs = work[c]; sm.fwd = work[c] + (int64_t)bm.mpb * NB; sm.st = (SolveState*)state[c];
        sm.gxyz = gxyz[c]; sm.gsig = gsig[c]; sm.glat = glat[c]; sm.n = n[c]; sm.xb = xb[c]; sm.xa = xa[c]; sm.inc = inc[c];
        sm.perm = nullptr;
        sm.which = c;
        if (bm.m > bt->max_m) bt->max_m = bm.m;
        if (n[c] > bt->max_n) bt->max_n = n[c];
        if ((int64_t)bm.mpb * NB > bt->max_mp) bt->max_mp = (int64_t)bm.mpb * NB;
    }
    // ticket -> (member, step): steps ascending, members in table order inside a step (a prefix: sorted by block count)
    for (int k = 0; k < bt->max_mpb; ++k)
        for (int i = 0; i < nmat && bt->table[i].mpb > k; ++i) ord.push_back((i << 12) | k);
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (bt->solve_dev) HIP_TRY(hipFree(bt->solve_dev));
    if (bt->ord_dev) HIP_TRY(hipFree(bt->ord_dev));
    if (bt->ctl_dev) HIP_TRY(hipFree(bt->ctl_dev));
    bt->solve_dev = nullptr; bt->ord_dev = nullptr; bt->ctl_dev = nullptr;
    HIP_TRY(hipMalloc((void**)&bt->solve_dev, sizeof(SolveMember) * nmat));
    HIP_TRY(hipMalloc((void**)&bt->ord_dev, sizeof(int) * ord.size()));
    HIP_TRY(hipMalloc(&bt->ctl_dev, 2 * kCtlBytes));
    bt->solve_host = mem;
    bt->max_patches = 0;
    HIP_TRY(hipMemcpy(bt->solve_dev, mem.data(), sizeof(SolveMember) * nmat, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(bt->ord_dev, ord.data(), sizeof(int) * ord.size(), hipMemcpyHostToDevice));
    HIP_TRY(hipMemset(bt->ctl_dev, 0, 2 * kCtlBytes));
    bt->ord_total = (int)ord.size();
    oisat_dag_plan_release(bt->dag_solve);
    bt->dag_solve = nullptr;
    return OISAT_OK;
}

// Width of every member's cell grid (its n cells are ny x nx, row-major), in the caller's member order; 0 = unknown.  The
// increment then works on compact patches of cells and skips the observations beyond the covariance's reach of a patch.
// perm (optional; entries may be NULL): member i's observations along a space-filling curve (dev int32[m_i], a permutation
// of its latitude order): the float64 residual then works on compact blocks of rows with the same cull.
extern "C" int oisat_batch_set_grid(oisat_ctx* h, int batch_id, int nmat, const int64_t* nx, const int32_t* const* perm) {
    ARG_CHECK(h && nx && batch_id >= 0 && batch_id < (int)h->batches.size() && h->batches[batch_id]);
    ChBatch* bt = h->batches[batch_id];
    ARG_CHECK(nmat == (int)bt->table.size() && bt->solve_dev != nullptr && (int)bt->solve_host.size() == nmat);
    for (int i = 0; i < nmat; ++i) {
        const int64_t w = nx[bt->order[i]];
        SolveMember& sm = bt->solve_host[i];
        ARG_CHECK(w >= 0 && w < (int64_t)INT32_MAX && (w == 0 || sm.n % w == 0));
        sm.nx = (int)w;
        sm.perm = perm ? perm[bt->order[i]] : nullptr;
    }
    HIP_TRY(hipStreamSynchronize(h->stream));
    HIP_TRY(hipMemcpy(bt->solve_dev, bt->solve_host.data(), sizeof(SolveMember) * nmat, hipMemcpyHostToDevice));
    oisat_dag_plan_release(bt->dag_solve);                  // (its increment patches depend on the grids' shapes)
    bt->dag_solve = nullptr;
    return OISAT_OK;
}

// the gain solve + increment of every member, each launch covering all of them (same arithmetic per member as
// oisat_gain_solve + oisat_apply_increment; same stopping rule, per member)
extern "C" int oisat_batch_solve(oisat_ctx* h, int batch_id, int dtype, double g, int refine) {
    ARG_CHECK(h && batch_id >= 0 && batch_id < (int)h->batches.size() && h->batches[batch_id]);
    ChBatch& bt = *h->batches[batch_id];
    ARG_CHECK(bt.solve_dev != nullptr && bt.ord_total > 0 && refine >= 0 && refine <= 8 && g >= 0.0);
    ARG_CHECK(dtype == OISAT_F32 || dtype == OISAT_F64);
    char* base = nullptr;
    if (int rc = status_ws(h, nullptr, &base)) return rc;
    HIP_TRY(dense_kernel_attributes());
    unsigned* err_total = (unsigned*)base;
    const int nmem = (int)bt.table.size();
    const SolveMember* mem = bt.solve_dev;
    TrsvCtl* ctl_f = (TrsvCtl*)bt.ctl_dev;
    TrsvCtl* ctl_b = (TrsvCtl*)((char*)bt.ctl_dev + kCtlBytes);
    const size_t shm = sizeof(float) * NB * TLD;
    // persistent workgroups: two per CU (66 KB of LDS each) or one per row if there are fewer rows
    const int slots = 2 * (h->cu_count > 0 ? h->cu_count : 256);
    const int grid = bt.ord_total < slots ? bt.ord_total : slots;
    const double tol = h->refine_tol;
    OISAT_LAUNCH(h, "copy_pad", solve_prep_batched_kernel, dim3((unsigned)stream_grid(bt.max_mp, 256) > 64u ? 64u : (unsigned)stream_grid(bt.max_mp, 256), (unsigned)nmem),
                 dim3(256), 0, mem);
    auto sweeps = [&](int first, int accumulate) -> int {
        OISAT_LAUNCH(h, "trsv_fwd", (trsv_batched_kernel<0>), dim3(grid), dim3(256), shm, mem, (const int*)bt.ord_dev, bt.ord_total, ctl_f,
                     err_total, first, 0);
        OISAT_LAUNCH(h, "trsv_bwd", (trsv_batched_kernel<1>), dim3(grid), dim3(256), shm, mem, (const int*)bt.ord_dev, bt.ord_total, ctl_b,
                     err_total, first, accumulate);
        return OISAT_OK;
    };
    int rc = sweeps(1, 0);
    if (rc) return rc;
    int* info_dev = nullptr;
    if (int rs = status_ws(h, &info_dev, nullptr)) return rs;
    for (int it = 0; it <= refine && refine > 0; ++it) {    // (the last evaluation only runs for members that are still above the tolerance)
        rc = oisat_cov_residual_batched(h, mem, bt.solve_host, bt.max_m, g);
        if (rc) return rc;
        OISAT_LAUNCH(h, "resid_check", resid_check_batched_kernel, dim3((unsigned)nmem), dim3(1024), 0, mem, it, tol * tol,
                     it == refine ? 1 : 0, info_dev);
        if (it == refine) break;
        rc = sweeps(0, 1);
        if (rc) return rc;
    }
    return oisat_apply_increment_batched(h, dtype, mem, bt.solve_host, bt.max_n, g);
}

extern "C" int oisat_batch_is_task_graph(oisat_ctx* h, int batch_id, int* yes_out) {
    ARG_CHECK(h && yes_out && batch_id >= 0 && batch_id < (int)h->batches.size() && h->batches[batch_id]);
    *yes_out = h->batches[batch_id]->dag != nullptr ? 1 : 0;
    return OISAT_OK;
}

// Factorization AND solve phase of every member as ONE task-graph launch (dense_dag.inc "The solve phase as tasks"): what
// oisat_batch_potrf + oisat_batch_solve do in 2 + 3 (refine + 2) launches with the solve phase exposed behind the
// factorization.  Same arithmetic per member.  Needs a batch whose factorization runs as a task graph (OISAT_ENOTSUP_DAG
// otherwise: the caller keeps the two calls).
extern "C" int oisat_batch_analyse(oisat_ctx* h, int batch_id, int dtype, double g, int refine, int* info_host) {
    ARG_CHECK(h && batch_id >= 0 && batch_id < (int)h->batches.size() && h->batches[batch_id]);
    ChBatch& bt = *h->batches[batch_id];
    ARG_CHECK(bt.solve_dev != nullptr && bt.ord_total > 0 && refine >= 0 && refine <= DAG_MAX_REFINE && g >= 0.0);
    ARG_CHECK(dtype == OISAT_F32 || dtype == OISAT_F64);
    if (!bt.dag) {
        oisat_set_error("oisat_batch_analyse: this batch's factorization does not run as a task graph (oisat_set_task_graph / size)");
        return OISAT_EINVAL;
    }
    int* info_dev = nullptr;
    char* base = nullptr;
    if (int rc = status_ws(h, &info_dev, &base)) return rc;
    HIP_TRY(dense_kernel_attributes());
    const int nmem = (int)bt.table.size();
    const double g2 = g * (double)kLog2e;
    DagSolve sv;
    sv.mem = bt.solve_dev;
    sv.g = g;
    sv.g2 = g2;
    sv.win_deg = lat_window_deg(g2);
    sv.cut_chord = cut_chord_of(g2);
    sv.tol2 = h->refine_tol * h->refine_tol;
    sv.refine = refine;
    sv.dtype = dtype;
    sv.cells = increment_cells(h->cu_count, bt.max_n, nmem);
    sv.blocks = residual_blocks_pay(g2) ? 1 : 0;
    for (const SolveMember& sm : bt.solve_host) sv.blocks = sv.blocks && sm.perm != nullptr;
    sv.trsv_timeouts = (unsigned*)base;
    sv.nsys = nmem;
    if (!bt.dag_solve || bt.dag_solve_refine != refine || bt.dag_solve_cells != sv.cells) {
        HIP_TRY(hipStreamSynchronize(h->stream));            // (a plan is never freed under a running launch)
        oisat_dag_plan_release(bt.dag_solve);
        bt.dag_solve = nullptr;
        DagSolveShape shape;
        shape.refine = refine;
        for (const SolveMember& sm : bt.solve_host) {
            shape.nres.push_back((int)cdiv(sm.m, 64));
            shape.ninc.push_back((int)increment_blocks(sm.n, sm.nx, sv.cells));
        }
        bt.dag_solve = dag_plan_create(bt.table, h->stream, shape);
        if (!bt.dag_solve) return OISAT_ENOMEM;
        HIP_TRY(hipStreamSynchronize(h->stream));            // (the plan's words are zero before any stream can launch it)
        bt.dag_solve_refine = refine;
        bt.dag_solve_cells = sv.cells;
    }
    sv.queue = ((DagPlan*)bt.dag_solve)->queue_dev;
    sv.qcap = (int)((DagPlan*)bt.dag_solve)->qcap;
    sv.qcap0 = (int)((DagPlan*)bt.dag_solve)->qcap0;
    HIP_TRY(hipMemsetAsync(info_dev, 0, sizeof(int), h->stream));
    HIP_TRY(hipMemsetAsync(info_dev + 3, 0, sizeof(int), h->stream));
    OISAT_LAUNCH(h, "pad_identity", pad_identity_batched_kernel, dim3(32, (unsigned)nmem), dim3(256), 0, (const BatchMat*)bt.table_dev);
    OISAT_LAUNCH(h, "copy_pad", solve_prep_batched_kernel, dim3((unsigned)stream_grid(bt.max_mp, 256) > 64u ? 64u : (unsigned)stream_grid(bt.max_mp, 256), (unsigned)nmem),
                 dim3(256), 0, (const SolveMember*)bt.solve_dev);
    if (int rc = dag_launch(h, *(DagPlan*)bt.dag_solve, info_dev, (unsigned*)(info_dev + kInfoDagTimeouts), &sv)) return rc;
    if (info_host) {
        int* pin = (int*)oisat_pinned(h, 64);
        if (!pin) return OISAT_ENOMEM;
        HIP_TRY(hipMemcpyAsync(pin, info_dev, 8 * sizeof(int), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
        info_host[0] = pin[0];
        info_host[1] = pin[0] ? bt.order[pin[3] - 1] : -1;          // the caller's matrix index
        if (pin[kInfoDagTimeouts] != 0) {
            HIP_TRY(hipMemsetAsync(info_dev + kInfoDagTimeouts, 0, sizeof(int), h->stream));
            oisat_set_error("batched analysis: the task-graph launch timed out (a workgroup gave up waiting): factors and fields are incomplete");
            return OISAT_EHIP;
        }
        if (pin[0] != 0) {
            HIP_TRY(hipMemsetAsync(info_dev + 1, 0, 2 * sizeof(int), h->stream));
            oisat_set_error("batched potrf: matrix %d not positive definite at column %d", info_host[1], pin[0]);
            return OISAT_ENOTPD;
        }
    }
    return OISAT_OK;
}

extern "C" int oisat_batch_potrf(oisat_ctx* h, int batch_id, int* info_host) {
    ARG_CHECK(h && batch_id >= 0 && batch_id < (int)h->batches.size() && h->batches[batch_id]);
    const ChBatch& bt = *h->batches[batch_id];
    int* info_dev = nullptr;
    if (int rc = status_ws(h, &info_dev, nullptr)) return rc;
    HIP_TRY(dense_kernel_attributes());
    HIP_TRY(hipMemsetAsync(info_dev, 0, sizeof(int), h->stream));
    HIP_TRY(hipMemsetAsync(info_dev + 3, 0, sizeof(int), h->stream));
    OISAT_LAUNCH(h, "pad_identity", pad_identity_batched_kernel, dim3(32, (unsigned)bt.table.size()), dim3(256), 0,
                 (const BatchMat*)bt.table_dev);
    int rc;
    if (bt.dag) {
        rc = dag_launch(h, *(DagPlan*)bt.dag, info_dev, (unsigned*)(info_dev + kInfoDagTimeouts));
    } else {
        rc = potrf_rec_batched(h, bt, 0, bt.max_mpb, info_dev);
    }
    if (rc) return rc;
    if (info_host) {
        int* pin = (int*)oisat_pinned(h, 64);
        if (!pin) return OISAT_ENOMEM;
        HIP_TRY(hipMemcpyAsync(pin, info_dev, 8 * sizeof(int), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
        info_host[0] = pin[0];
        info_host[1] = pin[0] ? bt.order[pin[3] - 1] : -1;          // the caller's matrix index
        if (pin[kInfoDagTimeouts] != 0) {
            HIP_TRY(hipMemsetAsync(info_dev + kInfoDagTimeouts, 0, sizeof(int), h->stream));
            oisat_set_error("batched potrf: the task-graph factorization timed out (a workgroup gave up waiting): the factors are incomplete");
            return OISAT_EHIP;
        }
        if (pin[0] != 0) {
            HIP_TRY(hipMemsetAsync(info_dev + 1, 0, 2 * sizeof(int), h->stream));
            oisat_set_error("batched potrf: matrix %d not positive definite at column %d", info_host[1], pin[0]);
            return OISAT_ENOTPD;
        }
    }
    return OISAT_OK;
}

extern "C" int oisat_factor_adopt(oisat_ctx* h, const float* L, int64_t m, int64_t ld, float* tinv) {
    ARG_CHECK(h && L && tinv && m > 0);
    const int64_t mp = cdiv(m, NB) * NB;
    ARG_CHECK(ld >= mp && (ld % 4) == 0 && ((uintptr_t)L % 16) == 0 && ((uintptr_t)tinv % 16) == 0);
    if (int rc = status_ws(h, nullptr, nullptr)) return rc;
    HIP_TRY(dense_kernel_attributes());
    h->factor.S = L;
    h->factor.m = m;
    h->factor.mp = mp;
    h->factor.ld = ld;
    h->factor.tinv = tinv;
    return OISAT_OK;
}


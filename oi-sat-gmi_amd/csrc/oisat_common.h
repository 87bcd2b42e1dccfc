// Internal to liboisat_hip.so -- not installed.  gfx950 only; 64-wide wavefronts are assumed.
#pragma once

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/oisat.h"

constexpr int kWave = 64;

void oisat_set_error(const char* fmt, ...);

struct ProfRec {
    char name[64];
    double total_ms = 0.0;
    int64_t launches = 0;
};

struct ProfPending {
    int rec;
    hipEvent_t a, b;
};

struct ChFactor {                // what the triangular solves need besides L: the inverted diagonal blocks
    const float* S = nullptr;
    int64_t m = 0, mp = 0, ld = 0;
    float* tinv = nullptr;       // [mp/128][128][128]: inverted diagonal blocks, row-major
};

struct BatchMat {                // one matrix of a batched factorization (device table entry)
    float* S;
    float* tinv;                 // [mpb][128][128]
    int64_t ld;
    int64_t m;
    int mpb;                     // roundup(m, 128) / 128
    int pad;
};

// state of one gain solve on the device: squared norms for the convergence test and the flag that turns the remaining
// refinement launches into no-ops (workspace slot 9 of a handle, or one 256-byte block per member of a batched solve)
struct SolveState {
    double dd;                   // |d|^2
    double norm[12];             // |r_k|^2, k = 0 .. refine
    int conv;                    // set when |r_k| <= tol |d|: every later residual / sweep launch of this solve returns at once
    int computed;                // number of residual norms stored
};

struct SolveMember {             // one system of a batched solve phase (device table, oisat_batch_set_solve)
    const float* S;              // its factor (after oisat_batch_potrf) and inverted diagonal blocks
    const float* tinv;
    int64_t ld, m;
    int mpb, nx;                 // block rows; width of the system's (ny x nx, row-major) cell grid, 0 = unknown (oisat_batch_set_grid)
    const double *oxyz, *osig, *ovar, *d, *olat;       // observations (ascending latitude), innovation
    double *z, *rhs, *fwd;       // solution; padded right-hand side and forward solution [mpb * 128] each
    SolveState* st;
    const double *gxyz, *gsig, *glat;                   // the system's grid cells
    int64_t n;
    const void* xb;              // background, analysis, increment (dtype of the batch)
    void* xa;
    void* inc;
    const int* perm;             // its observations along a space-filling curve (compact residual blocks) or nullptr
    int which;                   // the caller's index of this member (oisat_batch_set_solve order), reported by the status words
};

struct ChBatch {                 // matrices factored in lock-step by oisat_batch_potrf (sorted by block count, largest first)
    BatchMat* table_dev = nullptr;
    // compact tile enumeration of the persistent GEMM launches: for every node of the recursion tree the prefix sums of
    // the members' tile counts (cum[0] = 0 .. cum[cnt] = total), built once by oisat_batch_create
    int* cum_dev = nullptr;
    std::vector<int> cum_off, cum_cnt, cum_total;      // per node slot (see ChBatch::slot)
    std::vector<long long> cum_key;                     // node key -> slot (sorted, binary search)
    std::vector<BatchMat> table;
    std::vector<int> order;      // table[i] is the caller's matrix order[i]
    int max_mpb = 0;
    // batched solve phase (oisat_batch_set_solve): member table in the order of `table`, the sweep's ticket -> (member, step)
    // list (steps ascending, so that a row only waits for lower tickets) and two control blocks (forward | backward)
    SolveMember* solve_dev = nullptr;
    std::vector<SolveMember> solve_host;
    int64_t max_patches = 0;
    int* ord_dev = nullptr;
    void* ctl_dev = nullptr;
    int ord_total = 0;
    int64_t max_m = 0, max_n = 0, max_mp = 0;
    bool pairs = false;          // leaves of its recursion are pairs of block columns (many members: the leaf launches are HBM-bound)
    void* dag = nullptr;         // task-graph plan of the batch (dense_dag.inc: DagPlan), owned
    void* dag_solve = nullptr;   // ... of the batch's factorization + solve phase in one launch (oisat_batch_analyse), made on first use
    int dag_solve_refine = -1, dag_solve_cells = 0;             // what that plan was made for
};

struct DagSingle {               // a cached single-system task-graph plan (oisat_potrf)
    void* plan = nullptr;
    const float* S = nullptr;
    const float* tinv = nullptr;
    int64_t ld = 0, mpb = 0;
    uint64_t stamp = 0;
};

struct oisat_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    int cu_count = 0;
    size_t hbm_bytes = 0;
    char name[128] = {0};
    // profiling
    bool prof = false;
    std::vector<ProfRec> recs;
    std::vector<ProfPending> pending;
    std::vector<hipEvent_t> free_events;
    // grow-only device workspaces (never freed/reallocated inside a timed region once warm)
    void* ws[10] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    size_t ws_bytes[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    // last scaling sweep uploaded into workspace slot 0 (oi_diag.hip)
    double scales_host[OISAT_MAX_SCALES] = {0};
    int scales_n = 0;
    const void* scales_dev = nullptr;
    // the factor left by the last oisat_potrf on this handle (dense_chol.hip)
    ChFactor factor;
    bool small_tiles = true;            // gemm_nt: 64x64 tiles for launches of <= 700 128x128 tiles (latency-bound ones)
    double refine_tol = 1e-6;           // oisat_set_refine_tol: relative residual at which the gain solve stops refining
    int wave_prio = 0;                  // oisat_set_share: s_setprio of this handle's batched factorization kernels (0..3)
    int gemm_wg_per_cu = 0;             // oisat_set_share: workgroups per CU of its persistent GEMM launches (0 = default, 2)
    hipStream_t own_stream = nullptr;   // created by oisat_stream_create, destroyed at shutdown
    hipStream_t aux_stream = nullptr;   // look-ahead Cholesky: trailing updates run here, the panel chain on `stream`
    std::vector<hipEvent_t> sync_events;
    void* comm = nullptr;               // RCCL communicator (oisat_comm_init), opaque here
    int comm_rank = 0, comm_size = 1;
    hipEvent_t signal_event = nullptr;  // oisat_wait_for: recorded on this handle's stream, waited on by another handle's
    std::vector<ChBatch*> batches;      // oisat_batch_create
    // task-graph plans of the last single-system factorizations on this handle (dense_dag.inc), keyed by what a plan depends on
    const int* obs_perm = nullptr;      // oisat_set_obs_blocks: space-filling order of the observations of the next gain solves
    int64_t obs_perm_m = 0;
    int dag_mode = -1;                  // oisat_set_task_graph: -1 = by size (and OISAT_DAG), 0 = recursion only, 1 = task graph wherever it applies
    DagSingle dag_cache[8];
    uint64_t dag_clock = 0;
    // pinned host scratch for small synchronous read-backs
    void* pinned = nullptr;
    size_t pinned_bytes = 0;
    hipStream_t xfer_streams[4] = {nullptr, nullptr, nullptr, nullptr};   // oisat_d2h: slices of a large read-back, one host thread each
};

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess) {                                                                \
            oisat_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return OISAT_EHIP;                                                                 \
        }                                                                                      \
    } while (0)

#define ARG_CHECK(cond)                                                                        \
    do {                                                                                       \
        if (!(cond)) {                                                                         \
            oisat_set_error("invalid argument: %s (%s:%d)", #cond, __FILE__, __LINE__);        \
            return OISAT_EINVAL;                                                               \
        }                                                                                      \
    } while (0)

// workspace slot `slot` of at least `bytes` (grow-only); returns nullptr + error on failure
void* oisat_ws(oisat_ctx* h, int slot, size_t bytes);
void* oisat_pinned(oisat_ctx* h, size_t bytes);
void oisat_dag_plan_release(void* plan);                // dense_chol.hip: frees a task-graph plan (nullptr allowed)

int oisat_prof_begin(oisat_ctx* h, const char* name);   // returns pending index or -1
void oisat_prof_end(oisat_ctx* h, int pending);

// Launch with optional per-kernel event timing on the handle's stream, then check the launch.
#define OISAT_LAUNCH(h, NAME, kernel, grid, block, shmem, ...)                                 \
    do {                                                                                       \
        int _p = (h)->prof ? oisat_prof_begin((h), NAME) : -1;                                 \
        hipLaunchKernelGGL(kernel, grid, block, shmem, (h)->stream, __VA_ARGS__);              \
        if (_p >= 0) oisat_prof_end((h), _p);                                                  \
        hipError_t _le = hipGetLastError();                                                    \
        if (_le != hipSuccess) {                                                               \
            oisat_set_error("launch of %s failed: %s", NAME, hipGetErrorString(_le));          \
            return OISAT_EHIP;                                                                 \
        }                                                                                      \
    } while (0)

static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// memory-bound grid sizing: enough blocks to fill 256 CUs x 8, grid-stride the rest
static inline int stream_grid(int64_t work_items, int per_block) {
    int64_t g = cdiv(work_items, per_block);
    if (g < 1) g = 1;
    if (g > 2048) g = 2048;
    return (int)g;
}

template <typename T>
__device__ __forceinline__ T nan_of();
template <>
__device__ __forceinline__ float nan_of<float>() { return __builtin_nanf(""); }
template <>
__device__ __forceinline__ double nan_of<double>() { return __builtin_nan(""); }

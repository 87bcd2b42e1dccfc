// RCCL collectives of the path, as C-ABI entry points (SURVEY.md section 8(b) proposal, section 8(e)):
// one broadcast of what is genuinely shared (the month-invariant model grid) and a gather of finished
// analysis fields to rank 0.  The reference has no counterpart -- it runs one scheduler job per month and
// exchanges nothing (run/job_submitter_sbatch.py:45-68); these two calls are all the communication the
// MI355X sharding of (month x tile) units needs.  There is NO collective on the data path.
//
// librccl is resolved at the first oisat_comm_* call (dlopen; a copy already in the process -- PyTorch
// bundles one -- is reused), so the library itself loads on machines without RCCL.  The Python package
// reaches RCCL through torch.distributed (backend "nccl"); these entry points are for hosts that bind the
// C-ABI directly.
#include "oisat_common.h"

#include <dlfcn.h>

#include <mutex>

namespace {

typedef struct { char internal[128]; } rcclUniqueId;       // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128)
typedef void* rcclComm_t;
constexpr int kNcclChar = 0;                                // ncclInt8 / ncclChar

struct RcclApi {
    void* lib = nullptr;
    int (*GetUniqueId)(rcclUniqueId*) = nullptr;
    int (*CommInitRank)(rcclComm_t*, int, rcclUniqueId, int) = nullptr;
    int (*CommDestroy)(rcclComm_t) = nullptr;
    int (*Broadcast)(const void*, void*, size_t, int, int, rcclComm_t, hipStream_t) = nullptr;
    int (*Send)(const void*, size_t, int, int, rcclComm_t, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, rcclComm_t, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};

bool rccl_load(RcclApi& api) {
    const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {                           // a copy already mapped into the process first
        api.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
        if (api.lib) break;
    }
    for (int i = 0; !api.lib && i < 3; ++i) api.lib = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
    if (!api.lib) return false;
#define OISAT_SYM(field, name)                                          \
    *(void**)(&api.field) = dlsym(api.lib, name);                       \
    if (!api.field) { api.lib = nullptr; return false; }
    OISAT_SYM(GetUniqueId, "ncclGetUniqueId")
    OISAT_SYM(CommInitRank, "ncclCommInitRank")
    OISAT_SYM(CommDestroy, "ncclCommDestroy")
    OISAT_SYM(Broadcast, "ncclBroadcast")
    OISAT_SYM(Send, "ncclSend")
    OISAT_SYM(Recv, "ncclRecv")
    OISAT_SYM(GroupStart, "ncclGroupStart")
    OISAT_SYM(GroupEnd, "ncclGroupEnd")
    OISAT_SYM(GetErrorString, "ncclGetErrorString")
#undef OISAT_SYM
    return true;
}

// resolved once per process, whichever thread gets here first (LanePool drives handles from several threads)
RcclApi* rccl() {
    static RcclApi api;
    static std::once_flag once;
    static bool ok = false;
    std::call_once(once, [] { ok = rccl_load(api); });
    return ok ? &api : nullptr;
}

int rccl_missing() {
    const char* why = dlerror();
    oisat_set_error("librccl not found (dlopen librccl.so / librccl.so.1): %s", why ? why : "(no dlerror)");
    return OISAT_ENODEV;
}

#define RCCL_TRY(api, expr)                                                                    \
    do {                                                                                       \
        const int _r = (expr);                                                                 \
        if (_r != 0) {                                                                         \
            oisat_set_error("%s failed: %s", #expr, (api)->GetErrorString(_r));                \
            return OISAT_EHIP;                                                                 \
        }                                                                                      \
    } while (0)

}  // namespace

extern "C" int oisat_comm_unique_id(char* id_out, int cap) {
    ARG_CHECK(id_out != nullptr && cap >= 128);
    RcclApi* a = rccl();
    if (!a) return rccl_missing();
    rcclUniqueId id;
    RCCL_TRY(a, a->GetUniqueId(&id));
    memcpy(id_out, id.internal, 128);
    return OISAT_OK;
}

extern "C" int oisat_comm_init(oisat_ctx* h, int rank, int nranks, const char* unique_id) {
    ARG_CHECK(h != nullptr && unique_id != nullptr && nranks >= 1 && rank >= 0 && rank < nranks);
    ARG_CHECK(h->comm == nullptr);
    RcclApi* a = rccl();
    if (!a) return rccl_missing();
    HIP_TRY(hipSetDevice(h->device));
    rcclUniqueId id;
    memcpy(id.internal, unique_id, 128);
    rcclComm_t c = nullptr;
    RCCL_TRY(a, a->CommInitRank(&c, nranks, id, rank));
    h->comm = c;
    h->comm_rank = rank;
    h->comm_size = nranks;
    return OISAT_OK;
}

extern "C" int oisat_comm_destroy(oisat_ctx* h) {
    ARG_CHECK(h != nullptr);
    if (!h->comm) return OISAT_OK;
    RcclApi* a = rccl();
    ARG_CHECK(a != nullptr);
    HIP_TRY(hipStreamSynchronize(h->stream));
    RCCL_TRY(a, a->CommDestroy((rcclComm_t)h->comm));
    h->comm = nullptr;
    return OISAT_OK;
}

extern "C" int oisat_comm_bcast(oisat_ctx* h, void* dev_buf, size_t bytes, int root) {
    ARG_CHECK(h != nullptr && h->comm != nullptr && (bytes == 0 || dev_buf) && root >= 0 && root < h->comm_size);
    if (bytes == 0) return OISAT_OK;
    RcclApi* a = rccl();
    ARG_CHECK(a != nullptr);
    HIP_TRY(hipSetDevice(h->device));                       // HIP's current device is per thread
    RCCL_TRY(a, a->Broadcast(dev_buf, dev_buf, bytes, kNcclChar, root, (rcclComm_t)h->comm, h->stream));
    return OISAT_OK;
}

extern "C" int oisat_comm_gather(oisat_ctx* h, const void* send_dev, size_t bytes, void* recv_dev, int root) {
    ARG_CHECK(h != nullptr && h->comm != nullptr && send_dev != nullptr && bytes > 0 && root >= 0 && root < h->comm_size);
    ARG_CHECK(h->comm_rank != root || recv_dev != nullptr);
    RcclApi* a = rccl();
    ARG_CHECK(a != nullptr);
    HIP_TRY(hipSetDevice(h->device));
    rcclComm_t c = (rcclComm_t)h->comm;
    // a gather, not an all-gather: every rank sends its slab to the root, the root posts one receive per rank.
    // A failing Send / Recv must not leave the group open (every later RCCL call of this thread would be swallowed by
    // it): the first error is remembered, the group is always closed, then the error is reported.
    RCCL_TRY(a, a->GroupStart());
    int first = a->Send(send_dev, bytes, kNcclChar, root, c, h->stream);
    const char* what = "ncclSend";
    if (first == 0 && h->comm_rank == root)
        for (int r = 0; r < h->comm_size && first == 0; ++r) {
            first = a->Recv((char*)recv_dev + (size_t)r * bytes, bytes, kNcclChar, r, c, h->stream);
            what = "ncclRecv";
        }
    const int end = a->GroupEnd();
    if (first != 0) {
        oisat_set_error("%s failed inside the gather group: %s", what, a->GetErrorString(first));
        return OISAT_EHIP;
    }
    if (end != 0) {
        oisat_set_error("ncclGroupEnd failed: %s", a->GetErrorString(end));
        return OISAT_EHIP;
    }
    return OISAT_OK;
}

// Lifetime, errors, memory, stream and per-kernel event timing of liboisat_hip.so.
#include "oisat_common.h"
#include <thread>

static thread_local char g_err[512] = "";

void oisat_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* oisat_last_error(void) { return g_err; }
extern "C" const char* oisat_version(void) { return "oisat-hip 0.1 (gfx950)"; }

extern "C" int oisat_init(int device_id, oisat_ctx** out) {
    ARG_CHECK(out != nullptr);
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        oisat_set_error("no HIP device visible (%s)", e == hipSuccess ? "count = 0" : hipGetErrorString(e));
        return OISAT_ENODEV;
    }
    ARG_CHECK(device_id >= 0 && device_id < ndev);
    HIP_TRY(hipSetDevice(device_id));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device_id));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        oisat_set_error("device %d is %s; this library is built for gfx950 (MI355X) only", device_id, prop.gcnArchName);
        return OISAT_ENODEV;
    }
    oisat_ctx* h = new oisat_ctx();
    h->device = device_id;
    h->cu_count = prop.multiProcessorCount;
    h->hbm_bytes = prop.totalGlobalMem;
    snprintf(h->name, sizeof(h->name), "%s (%s)", prop.name, prop.gcnArchName);
    *out = h;
    return OISAT_OK;
}

extern "C" int oisat_comm_destroy(oisat_ctx* h);

extern "C" void oisat_shutdown(oisat_ctx* h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->comm) (void)oisat_comm_destroy(h);
    (void)hipDeviceSynchronize();
    for (auto& p : h->pending) {
        (void)hipEventDestroy(p.a);
        (void)hipEventDestroy(p.b);
    }
    for (auto ev : h->free_events) (void)hipEventDestroy(ev);
    for (int i = 0; i < 10; ++i)
        if (h->ws[i]) (void)hipFree(h->ws[i]);
    if (h->pinned) (void)hipHostFree(h->pinned);
    if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
    if (h->aux_stream) (void)hipStreamDestroy(h->aux_stream);
    for (hipStream_t& xs : h->xfer_streams)
        if (xs) (void)hipStreamDestroy(xs);
    for (auto ev : h->sync_events) (void)hipEventDestroy(ev);
    if (h->signal_event) (void)hipEventDestroy(h->signal_event);
    for (auto* b : h->batches)
        if (b) {
            if (b->table_dev) (void)hipFree(b->table_dev);
            if (b->cum_dev) (void)hipFree(b->cum_dev);
            if (b->solve_dev) (void)hipFree(b->solve_dev);
            if (b->ord_dev) (void)hipFree(b->ord_dev);
            if (b->ctl_dev) (void)hipFree(b->ctl_dev);
            oisat_dag_plan_release(b->dag);
            delete b;
        }
    for (DagSingle& c : h->dag_cache) oisat_dag_plan_release(c.plan);
    delete h;
}

extern "C" int oisat_device_info(oisat_ctx* h, char* name_out, int name_cap, int* cu_count, int64_t* hbm_bytes) {
    ARG_CHECK(h != nullptr);
    if (name_out && name_cap > 0) snprintf(name_out, name_cap, "%s", h->name);
    if (cu_count) *cu_count = h->cu_count;
    if (hbm_bytes) *hbm_bytes = (int64_t)h->hbm_bytes;
    return OISAT_OK;
}

extern "C" int oisat_set_stream(oisat_ctx* h, void* hip_stream) {
    ARG_CHECK(h != nullptr);
    h->stream = (hipStream_t)hip_stream;
    return OISAT_OK;
}

extern "C" int oisat_stream_create(oisat_ctx* h) {
    ARG_CHECK(h != nullptr);
    if (!h->own_stream) HIP_TRY(hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking));
    h->stream = h->own_stream;
    return OISAT_OK;
}

extern "C" int oisat_stream_create_masked(oisat_ctx* h, int reserve_per_xcd) {
    ARG_CHECK(h != nullptr && reserve_per_xcd >= 0 && h->own_stream == nullptr);
    const int ncu = h->cu_count > 0 ? h->cu_count : 256;
    ARG_CHECK(8 * reserve_per_xcd < ncu);
    if (reserve_per_xcd == 0) return oisat_stream_create(h);
    // bit i of a queue's CU mask is a CU of XCD (i mod 8) (probed with HW_REG_XCC_ID): the top
    // 8 * reserve bits take `reserve` CUs off every XCD
    std::vector<uint32_t> mask((ncu + 31) / 32, 0xffffffffu);
    for (int b = ncu - 8 * reserve_per_xcd; b < (int)mask.size() * 32; ++b) mask[b / 32] &= ~(1u << (b % 32));
    HIP_TRY(hipExtStreamCreateWithCUMask(&h->own_stream, (uint32_t)mask.size(), mask.data()));
    h->stream = h->own_stream;
    h->cu_count = ncu - 8 * reserve_per_xcd;            // grids sized per CU (persistent GEMMs, two-tile sweeps) follow the mask
    return OISAT_OK;
}

extern "C" int oisat_bind_thread(oisat_ctx* h) {
    ARG_CHECK(h != nullptr);
    HIP_TRY(hipSetDevice(h->device));       // the current device is per host thread in HIP
    return OISAT_OK;
}

extern "C" int oisat_wait_for(oisat_ctx* waiter, oisat_ctx* signaler) {
    ARG_CHECK(waiter != nullptr && signaler != nullptr);
    if (waiter->stream == signaler->stream) return OISAT_OK;          // same stream: already ordered
    if (!signaler->signal_event) HIP_TRY(hipEventCreateWithFlags(&signaler->signal_event, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(signaler->signal_event, signaler->stream));
    HIP_TRY(hipStreamWaitEvent(waiter->stream, signaler->signal_event, 0));
    return OISAT_OK;
}

extern "C" int oisat_sync(oisat_ctx* h) {
    ARG_CHECK(h != nullptr);
    HIP_TRY(hipStreamSynchronize(h->stream));
    return OISAT_OK;
}

extern "C" int oisat_query(oisat_ctx* h, int* busy) {
    ARG_CHECK(h != nullptr && busy != nullptr);
    const hipError_t e = hipStreamQuery(h->stream);
    if (e != hipSuccess && e != hipErrorNotReady) {
        oisat_set_error("hipStreamQuery failed: %s", hipGetErrorString(e));
        return OISAT_EHIP;
    }
    *busy = e == hipErrorNotReady ? 1 : 0;
    return OISAT_OK;
}

extern "C" int oisat_dmalloc(oisat_ctx* h, size_t bytes, void** dev_out) {
    ARG_CHECK(h != nullptr && dev_out != nullptr);
    *dev_out = nullptr;
    if (bytes == 0) bytes = 16;
    hipError_t e = hipMalloc(dev_out, bytes);
    if (e != hipSuccess) {
        oisat_set_error("hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
        return OISAT_ENOMEM;
    }
    return OISAT_OK;
}

extern "C" int oisat_dfree(oisat_ctx* h, void* dev) {
    ARG_CHECK(h != nullptr);
    if (dev) HIP_TRY(hipFree(dev));
    return OISAT_OK;
}

extern "C" int oisat_h2d(oisat_ctx* h, void* dev_dst, const void* host_src, size_t bytes) {
    ARG_CHECK(h != nullptr && (bytes == 0 || (dev_dst && host_src)));
    if (bytes) HIP_TRY(hipMemcpyAsync(dev_dst, host_src, bytes, hipMemcpyHostToDevice, h->stream));
    return OISAT_OK;
}

// A read-back into pageable memory (a fresh NumPy array: the drop-in returns host arrays) is bound by ONE host thread moving the
// runtime's staging buffer into pages it faults in as it goes (16 GB/s: 36 of the 47 ms of a 73-field type-4 granule, 580 MB).
// A large one is therefore cut into four slices, each copied by a host thread of its own on a stream of its own.
constexpr size_t kSlicedReadback = (size_t)48 << 20;

extern "C" int oisat_d2h(oisat_ctx* h, void* host_dst, const void* dev_src, size_t bytes) {
    ARG_CHECK(h != nullptr && (bytes == 0 || (host_dst && dev_src)));
    if (bytes < kSlicedReadback) {
        if (bytes) HIP_TRY(hipMemcpyAsync(host_dst, dev_src, bytes, hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
        return OISAT_OK;
    }
    HIP_TRY(hipStreamSynchronize(h->stream));                 // what is read back has been produced on the handle's stream
    constexpr int T = 4;
    for (int t = 0; t < T; ++t)
        if (!h->xfer_streams[t]) HIP_TRY(hipStreamCreateWithFlags(&h->xfer_streams[t], hipStreamNonBlocking));
    const size_t slice = ((bytes / T) + 4095) & ~(size_t)4095;
    hipError_t rc[T];
    std::thread th[T];
    for (int t = 0; t < T; ++t) {
        rc[t] = hipSuccess;
        const size_t off = (size_t)t * slice;
        if (off >= bytes) continue;
        const size_t len = bytes - off < slice ? bytes - off : slice;
        auto copy_slice = [=, &rc]() {
            hipError_t e = hipSetDevice(h->device);
            if (e == hipSuccess) e = hipMemcpyAsync((char*)host_dst + off, (const char*)dev_src + off, len, hipMemcpyDeviceToHost, h->xfer_streams[t]);
            if (e == hipSuccess) e = hipStreamSynchronize(h->xfer_streams[t]);
            rc[t] = e;
        };
        try {
            th[t] = std::thread(copy_slice);
        } catch (...) {                                       // no thread to be had: this slice on the calling thread
            copy_slice();
        }
    }
    for (int t = 0; t < T; ++t)
        if (th[t].joinable()) th[t].join();
    for (int t = 0; t < T; ++t) HIP_TRY(rc[t]);
    return OISAT_OK;
}

extern "C" int oisat_memset(oisat_ctx* h, void* dev, int byte_value, size_t bytes) {
    ARG_CHECK(h != nullptr && (bytes == 0 || dev));
    if (bytes) HIP_TRY(hipMemsetAsync(dev, byte_value, bytes, h->stream));
    return OISAT_OK;
}

void* oisat_ws(oisat_ctx* h, int slot, size_t bytes) {
    if (bytes == 0) bytes = 16;
    if (h->ws_bytes[slot] >= bytes) return h->ws[slot];
    // growing: the old block may still be in use by enqueued work on our stream
    if (h->ws[slot]) {
        (void)hipStreamSynchronize(h->stream);
        (void)hipFree(h->ws[slot]);
        h->ws[slot] = nullptr;
        h->ws_bytes[slot] = 0;
    }
    size_t want = bytes + bytes / 8;
    hipError_t e = hipMalloc(&h->ws[slot], want);
    if (e != hipSuccess) {
        want = bytes;
        e = hipMalloc(&h->ws[slot], want);
    }
    if (e != hipSuccess) {
        oisat_set_error("workspace hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
        h->ws[slot] = nullptr;
        return nullptr;
    }
    h->ws_bytes[slot] = want;
    return h->ws[slot];
}

void* oisat_pinned(oisat_ctx* h, size_t bytes) {
    if (h->pinned_bytes >= bytes) return h->pinned;
    if (h->pinned) (void)hipHostFree(h->pinned);
    h->pinned = nullptr;
    h->pinned_bytes = 0;
    size_t want = bytes < 65536 ? 65536 : bytes;
    if (hipHostMalloc(&h->pinned, want, hipHostMallocDefault) != hipSuccess) {
        oisat_set_error("hipHostMalloc(%zu) failed", want);
        h->pinned = nullptr;
        return nullptr;
    }
    h->pinned_bytes = want;
    return h->pinned;
}

// ---- profiling ---------------------------------------------------------------------------------
static hipEvent_t take_event(oisat_ctx* h) {
    if (!h->free_events.empty()) {
        hipEvent_t e = h->free_events.back();
        h->free_events.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

int oisat_prof_begin(oisat_ctx* h, const char* name) {
    int rec = -1;
    for (size_t i = 0; i < h->recs.size(); ++i)
        if (strncmp(h->recs[i].name, name, 63) == 0) { rec = (int)i; break; }
    if (rec < 0) {
        ProfRec r;
        snprintf(r.name, sizeof(r.name), "%s", name);
        h->recs.push_back(r);
        rec = (int)h->recs.size() - 1;
    }
    ProfPending p;
    p.rec = rec;
    p.a = take_event(h);
    p.b = take_event(h);
    if (!p.a || !p.b) return -1;
    (void)hipEventRecord(p.a, h->stream);
    h->pending.push_back(p);
    return (int)h->pending.size() - 1;
}

void oisat_prof_end(oisat_ctx* h, int pending) { (void)hipEventRecord(h->pending[pending].b, h->stream); }

static void prof_drain(oisat_ctx* h) {
    (void)hipStreamSynchronize(h->stream);
    for (auto& p : h->pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            h->recs[p.rec].total_ms += ms;
            h->recs[p.rec].launches += 1;
        }
        h->free_events.push_back(p.a);
        h->free_events.push_back(p.b);
    }
    h->pending.clear();
}

extern "C" int oisat_prof_enable(oisat_ctx* h, int on) {
    ARG_CHECK(h != nullptr);
    if (!on && h->prof) prof_drain(h);
    h->prof = on != 0;
    return OISAT_OK;
}

extern "C" int oisat_prof_reset(oisat_ctx* h) {
    ARG_CHECK(h != nullptr);
    prof_drain(h);
    h->recs.clear();
    return OISAT_OK;
}

extern "C" int oisat_prof_collect(oisat_ctx* h, int cap, char (*names)[64], double* total_ms, int64_t* launches) {
    ARG_CHECK(h != nullptr);
    prof_drain(h);
    int n = (int)h->recs.size();
    for (int i = 0; i < n && i < cap; ++i) {
        if (names) memcpy(names[i], h->recs[i].name, 64);
        if (total_ms) total_ms[i] = h->recs[i].total_ms;
        if (launches) launches[i] = h->recs[i].launches;
    }
    return n;
}

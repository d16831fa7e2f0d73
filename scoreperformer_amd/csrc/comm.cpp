// Data-parallel collective of the train step: spn_comm_{unique_id,init,allreduce,wait,destroy} -- thin wrappers over
// RCCL's ncclAllReduce on a dedicated communication stream, fenced against the producer / consumer streams with HIP events.
//
// The reference is single-device; what this replaces is the gradient all-reduce that `torch.distributed` would run for a
// data-parallel `trainer.py` (one process per GPU, gradients in ONE contiguous fp32 arena, bucket by bucket while the backward is
// still running: scoreperformer_amd/parallel.py).
//
// RCCL is bound at RUN time with dlopen / dlsym: PyTorch ships its own librccl.so, and a second copy linked into libspn.so would put
// two RCCL runtimes (two sets of proxy threads, two IPC handle caches) into one process.  spn_comm_init takes the path of the RCCL
// library to use (null: "librccl.so" through the loader's search path); RTLD_NOLOAD is tried first, so a copy that the process has
// already loaded (torch's) is the one that gets used.
//
// Stream contract:  spn_comm_allreduce(comm, buf, count, dtype, producer_stream)  records an event on `producer_stream` (everything
// that wrote `buf` so far), makes the communication stream wait for it, and enqueues an in-place sum all-reduce there: the call
// returns at once and the producer stream is free to run the rest of the backward.  spn_comm_wait(comm, consumer_stream) makes
// `consumer_stream` wait for every all-reduce enqueued so far (no host synchronisation anywhere).  Streams and events are created in
// spn_comm_init and released in spn_comm_destroy; the compute entry points never allocate.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <mutex>

#define SPN_OK 0
#define SPN_ERR_ARG -1
#define SPN_ERR_HIP -2
#define SPN_ERR_COMM -3

extern "C" void spn_set_error(const char* msg);

namespace {

// the part of rccl.h that is used (ABI-stable since NCCL 2.10)
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
enum { kNcclSuccess = 0, kNcclSum = 0, kNcclFloat32 = 7, kNcclBfloat16 = 9 };
typedef int (*ncclGetUniqueId_t)(ncclUniqueId*);
typedef int (*ncclCommInitRank_t)(ncclComm_t*, int, ncclUniqueId, int);
typedef int (*ncclAllReduce_t)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t);
typedef int (*ncclCommDestroy_t)(ncclComm_t);
typedef const char* (*ncclGetErrorString_t)(int);

struct Rccl {
    void* handle = nullptr;
    ncclGetUniqueId_t get_unique_id = nullptr;
    ncclCommInitRank_t comm_init_rank = nullptr;
    ncclAllReduce_t all_reduce = nullptr;
    ncclCommDestroy_t comm_destroy = nullptr;
    ncclGetErrorString_t error_string = nullptr;
};
Rccl g_rccl;
std::mutex g_rccl_mutex;

int bind_rccl(const char* path) {
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (g_rccl.handle) return SPN_OK;
    const char* name = (path && path[0]) ? path : "librccl.so";
    void* h = dlopen(name, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);   // the copy this process already uses, if any
    if (!h) h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
    if (!h) { spn_set_error("spn_comm: cannot load the RCCL library (pass its path to spn_comm_init / spn_comm_unique_id)"); return SPN_ERR_COMM; }
    Rccl r;
    r.handle = h;
    r.get_unique_id = (ncclGetUniqueId_t)dlsym(h, "ncclGetUniqueId");
    r.comm_init_rank = (ncclCommInitRank_t)dlsym(h, "ncclCommInitRank");
    r.all_reduce = (ncclAllReduce_t)dlsym(h, "ncclAllReduce");
    r.comm_destroy = (ncclCommDestroy_t)dlsym(h, "ncclCommDestroy");
    r.error_string = (ncclGetErrorString_t)dlsym(h, "ncclGetErrorString");
    if (!r.get_unique_id || !r.comm_init_rank || !r.all_reduce || !r.comm_destroy) {
        spn_set_error("spn_comm: the RCCL library lacks ncclGetUniqueId / ncclCommInitRank / ncclAllReduce / ncclCommDestroy");
        return SPN_ERR_COMM;
    }
    g_rccl = r;
    return SPN_OK;
}

int nccl_fail(const char* what, int rc) {
    char msg[256];
    const char* s = g_rccl.error_string ? g_rccl.error_string(rc) : "";
    snprintf(msg, sizeof(msg), "%s failed: RCCL error %d (%s)", what, rc, s ? s : "");
    spn_set_error(msg);
    return SPN_ERR_COMM;
}

struct SpnComm {
    ncclComm_t comm;
    hipStream_t stream;    // dedicated communication stream
    hipEvent_t produced;   // producer stream -> communication stream
    hipEvent_t reduced;    // communication stream -> consumer stream
    int nranks, rank;
    int device;            // the device that was current in spn_comm_init: stream, events and the RCCL communicator live on it
};

// Every entry point runs with the communicator's device current (streams / events of another device are invalid handles): a caller
// on a different device is switched over for the duration of the call and switched back.
struct DeviceGuard {
    int prev = -1; bool switched = false, ok = true;
    explicit DeviceGuard(int want) {
        if (hipGetDevice(&prev) != hipSuccess) { ok = false; return; }
        if (prev != want) { ok = hipSetDevice(want) == hipSuccess; switched = ok; }
    }
    ~DeviceGuard() { if (switched) (void)hipSetDevice(prev); }
};

}  // namespace

// No collective, no bootstrap thread, no device call: can this process reach RCCL through the library at all (the shared object loads and
// carries the four entry points)?  A data-parallel launcher asks every rank BEFORE any rank enters spn_comm_init's collective.
extern "C" int spn_comm_available(const char* rccl_path) { return bind_rccl(rccl_path); }

extern "C" int spn_comm_unique_id(void* id128, const char* rccl_path) {
    if (!id128) { spn_set_error("spn_comm_unique_id: null output"); return SPN_ERR_ARG; }
    if (int rc = bind_rccl(rccl_path)) return rc;
    ncclUniqueId id;
    const int rc = g_rccl.get_unique_id(&id);
    if (rc != kNcclSuccess) return nccl_fail("ncclGetUniqueId", rc);
    memcpy(id128, &id, sizeof(id));
    return SPN_OK;
}

// Collective: every rank of the job calls it with the SAME 128-byte id (made by spn_comm_unique_id on one rank and shared through
// any out-of-band channel), after selecting its device with hipSetDevice.
extern "C" int spn_comm_init(void** comm_out, int nranks, int rank, const void* id128, const char* rccl_path) {
    if (!comm_out || !id128 || nranks < 1 || rank < 0 || rank >= nranks) { spn_set_error("spn_comm_init: bad arguments"); return SPN_ERR_ARG; }
    if (int rc = bind_rccl(rccl_path)) return rc;
    SpnComm* c = new SpnComm();
    c->nranks = nranks; c->rank = rank;
    if (hipGetDevice(&c->device) != hipSuccess) { delete c; spn_set_error("spn_comm_init: hipGetDevice failed"); return SPN_ERR_HIP; }
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    int rc = g_rccl.comm_init_rank(&c->comm, nranks, id, rank);
    if (rc != kNcclSuccess) { delete c; return nccl_fail("ncclCommInitRank", rc); }
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&c->produced, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->reduced, hipEventDisableTiming) != hipSuccess) {
        spn_set_error("spn_comm_init: cannot create the communication stream / events");
        g_rccl.comm_destroy(c->comm);
        delete c;
        return SPN_ERR_HIP;
    }
    *comm_out = c;
    return SPN_OK;
}

// In-place sum over the ranks of buf[0 .. count), dtype 0 = fp32, 1 = bf16; asynchronous (see the stream contract above).
extern "C" int spn_comm_allreduce(void* comm, void* buf, size_t count, int dtype, hipStream_t producer_stream) {
    SpnComm* c = (SpnComm*)comm;
    if (!c || !buf || count == 0 || (dtype != 0 && dtype != 1)) { spn_set_error("spn_comm_allreduce: bad arguments"); return SPN_ERR_ARG; }
    DeviceGuard guard(c->device);
    if (!guard.ok) { spn_set_error("spn_comm_allreduce: cannot select the communicator's device"); return SPN_ERR_HIP; }
    int buf_dev = c->device;
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, buf) == hipSuccess) buf_dev = attr.device;
    if (buf_dev != c->device) { spn_set_error("spn_comm_allreduce: the buffer lives on another device than the communicator"); return SPN_ERR_ARG; }
    if (hipEventRecord(c->produced, producer_stream) != hipSuccess || hipStreamWaitEvent(c->stream, c->produced, 0) != hipSuccess) {
        spn_set_error("spn_comm_allreduce: cannot fence the communication stream behind the producer stream");
        return SPN_ERR_HIP;
    }
    const int rc = g_rccl.all_reduce(buf, buf, count, dtype == 0 ? kNcclFloat32 : kNcclBfloat16, kNcclSum, c->comm, c->stream);
    if (rc != kNcclSuccess) return nccl_fail("ncclAllReduce", rc);
    if (hipEventRecord(c->reduced, c->stream) != hipSuccess) { spn_set_error("spn_comm_allreduce: hipEventRecord failed"); return SPN_ERR_HIP; }
    return SPN_OK;
}

// `consumer_stream` waits (on the device) for every all-reduce enqueued so far.
extern "C" int spn_comm_wait(void* comm, hipStream_t consumer_stream) {
    SpnComm* c = (SpnComm*)comm;
    if (!c) { spn_set_error("spn_comm_wait: null communicator"); return SPN_ERR_ARG; }
    DeviceGuard guard(c->device);
    if (!guard.ok) { spn_set_error("spn_comm_wait: cannot select the communicator's device"); return SPN_ERR_HIP; }
    if (hipStreamWaitEvent(consumer_stream, c->reduced, 0) != hipSuccess) { spn_set_error("spn_comm_wait: hipStreamWaitEvent failed"); return SPN_ERR_HIP; }
    return SPN_OK;
}

extern "C" int spn_comm_destroy(void* comm) {
    SpnComm* c = (SpnComm*)comm;
    if (!c) return SPN_OK;
    DeviceGuard guard(c->device);
    (void)hipStreamSynchronize(c->stream);   // teardown only: nothing of ours is left in flight when stream and events go, whatever RCCL returns
    const int rc = g_rccl.comm_destroy ? g_rccl.comm_destroy(c->comm) : kNcclSuccess;
    (void)hipEventDestroy(c->produced);
    (void)hipEventDestroy(c->reduced);
    (void)hipStreamDestroy(c->stream);
    delete c;
    return rc == kNcclSuccess ? SPN_OK : nccl_fail("ncclCommDestroy", rc);
}

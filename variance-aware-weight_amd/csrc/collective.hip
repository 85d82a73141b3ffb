// Gradient-bucket collectives straight on RCCL (vaw_allreduce_bucket_start / wait; §8(e)): the C-ABI form of what
// parallel.py otherwise asks torch.distributed for -- one communicator per process (= per GPU), a side HIP stream of its own, every
// collective ordered behind the caller's stream by an event and joined back by another.  Replaces torch DDP's bucket all-reduce
// (reference main.py:347, tools/dist_util.py:55 setup).  RCCL is opened with dlopen at the first call: libvaw_hip.so loads and every
// other entry point works on a box without it; ring / direct algorithm selection over the point-to-point xGMI links is RCCL's.
// Not capturable into a hipGraph (events are recorded on the caller's stream at call time): call it from eager spans.
#include <dlfcn.h>
#include <string.h>
#include <rccl/rccl.h>

#include "common.h"

namespace {
struct Rccl {
    void* so = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*ReduceScatter)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;
constexpr int N_EVENTS = 64;
struct Comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 0;
    hipStream_t side = nullptr;
    hipEvent_t ready[N_EVENTS] = {};      // caller's stream -> side stream, one per start (round robin)
    hipEvent_t done = nullptr;            // side stream -> whoever waits
    unsigned issued = 0;
};
Comm g_comm;

bool rccl_open() {
    if (g_rccl.so) return true;
    void* so = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!so) so = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!so) {
        vaw_set_error("collective: RCCL not found (%s)", dlerror());
        return false;
    }
#define SYM(field, name)                                                   \
    *(void**)(&g_rccl.field) = dlsym(so, name);                            \
    if (!g_rccl.field) {                                                   \
        vaw_set_error("collective: librccl has no %s", name);              \
        dlclose(so);                                                       \
        return false;                                                      \
    }
    SYM(GetUniqueId, "ncclGetUniqueId")
    SYM(CommInitRank, "ncclCommInitRank")
    SYM(CommDestroy, "ncclCommDestroy")
    SYM(AllReduce, "ncclAllReduce")
    SYM(ReduceScatter, "ncclReduceScatter")
    SYM(AllGather, "ncclAllGather")
    SYM(GetErrorString, "ncclGetErrorString")
#undef SYM
    g_rccl.so = so;
    return true;
}
#define RCCL_CHECK(call, what)                                                                  \
    do {                                                                                        \
        const ncclResult_t r_ = (call);                                                         \
        if (r_ != ncclSuccess) {                                                                \
            vaw_set_error("collective: %s failed: %s", what, g_rccl.GetErrorString(r_));        \
            return VAW_ERR_LAUNCH;                                                              \
        }                                                                                       \
    } while (0)
#define HIP_CHECK(call, what)                                                                   \
    do {                                                                                        \
        const hipError_t e_ = (call);                                                           \
        if (e_ != hipSuccess) {                                                                 \
            vaw_set_error("collective: %s failed: %s", what, hipGetErrorString(e_));            \
            return VAW_ERR_LAUNCH;                                                              \
        }                                                                                       \
    } while (0)

ncclDataType_t wire_type(vaw_dtype dt) { return dt == VAW_BF16 ? ncclBfloat16 : ncclFloat32; }

// order the side stream behind everything enqueued on `stream` so far
int side_after(hipStream_t stream) {
    hipEvent_t ev = g_comm.ready[g_comm.issued++ % N_EVENTS];
    HIP_CHECK(hipEventRecord(ev, stream), "event record");
    HIP_CHECK(hipStreamWaitEvent(g_comm.side, ev, 0), "stream wait");
    return VAW_OK;
}
int mark_done() {
    HIP_CHECK(hipEventRecord(g_comm.done, g_comm.side), "event record");
    return VAW_OK;
}
}  // namespace

extern "C" int vaw_comm_unique_id(void* out128) {
    VAW_CHECK_ARG(out128 != nullptr, "comm_unique_id: null output");
    if (!rccl_open()) return VAW_ERR_UNSUPPORTED;
    ncclUniqueId id;
    RCCL_CHECK(g_rccl.GetUniqueId(&id), "ncclGetUniqueId");
    memcpy(out128, id.internal, NCCL_UNIQUE_ID_BYTES);
    return VAW_OK;
}

extern "C" int vaw_comm_init(const void* id128, int rank, int world) {
    VAW_CHECK_ARG(id128 && world >= 1 && rank >= 0 && rank < world, "comm_init: bad rank %d of %d", rank, world);
    VAW_CHECK_ARG(g_comm.comm == nullptr, "comm_init: a communicator already exists (vaw_comm_destroy first)");
    if (!rccl_open()) return VAW_ERR_UNSUPPORTED;
    ncclUniqueId id;
    memcpy(id.internal, id128, NCCL_UNIQUE_ID_BYTES);
    RCCL_CHECK(g_rccl.CommInitRank(&g_comm.comm, world, id, rank), "ncclCommInitRank");
    g_comm.rank = rank;
    g_comm.world = world;
    g_comm.issued = 0;
    HIP_CHECK(hipStreamCreateWithFlags(&g_comm.side, hipStreamNonBlocking), "stream create");
    for (int i = 0; i < N_EVENTS; ++i) HIP_CHECK(hipEventCreateWithFlags(&g_comm.ready[i], hipEventDisableTiming), "event create");
    HIP_CHECK(hipEventCreateWithFlags(&g_comm.done, hipEventDisableTiming), "event create");
    HIP_CHECK(hipEventRecord(g_comm.done, g_comm.side), "event record");
    return VAW_OK;
}

extern "C" int vaw_comm_world(void) { return g_comm.comm ? g_comm.world : 0; }

extern "C" int vaw_comm_destroy(void) {
    if (!g_comm.comm) return VAW_OK;
    (void)hipStreamSynchronize(g_comm.side);
    RCCL_CHECK(g_rccl.CommDestroy(g_comm.comm), "ncclCommDestroy");
    for (int i = 0; i < N_EVENTS; ++i) (void)hipEventDestroy(g_comm.ready[i]);
    (void)hipEventDestroy(g_comm.done);
    (void)hipStreamDestroy(g_comm.side);
    g_comm = Comm();
    return VAW_OK;
}

// buf[0..count) <- mean over ranks, in place, on the communicator's stream, after everything enqueued on `stream` so far
extern "C" int vaw_allreduce_bucket_start(void* buf, int64_t count, vaw_dtype dt, vaw_stream stream) {
    VAW_CHECK_ARG(g_comm.comm != nullptr, "allreduce_bucket_start: no communicator (vaw_comm_init)");
    VAW_CHECK_ARG(buf && count > 0 && (dt == VAW_F32 || dt == VAW_BF16), "allreduce_bucket_start: bad bucket");
    if (const int rc = side_after((hipStream_t)stream)) return rc;
    RCCL_CHECK(g_rccl.AllReduce(buf, buf, (size_t)count, wire_type(dt), ncclAvg, g_comm.comm, g_comm.side), "ncclAllReduce");
    return mark_done();
}

// ZeRO-1 form: rank r's chunk (count / world elements at buf + r * chunk) <- mean over ranks of that chunk; the rest of buf is scratch
extern "C" int vaw_reduce_scatter_bucket_start(void* buf, int64_t count, vaw_dtype dt, vaw_stream stream) {
    VAW_CHECK_ARG(g_comm.comm != nullptr, "reduce_scatter_bucket_start: no communicator (vaw_comm_init)");
    VAW_CHECK_ARG(buf && count > 0 && count % g_comm.world == 0 && (dt == VAW_F32 || dt == VAW_BF16),
                  "reduce_scatter_bucket_start: the bucket must split into %d equal chunks", g_comm.world);
    if (const int rc = side_after((hipStream_t)stream)) return rc;
    const int64_t chunk = count / g_comm.world;
    char* mine = (char*)buf + (size_t)g_comm.rank * chunk * (dt == VAW_BF16 ? 2 : 4);
    RCCL_CHECK(g_rccl.ReduceScatter(buf, mine, (size_t)chunk, wire_type(dt), ncclAvg, g_comm.comm, g_comm.side), "ncclReduceScatter");
    return mark_done();
}

// the way back: every rank's chunk of buf to all ranks, in place
extern "C" int vaw_allgather_bucket_start(void* buf, int64_t count, vaw_dtype dt, vaw_stream stream) {
    VAW_CHECK_ARG(g_comm.comm != nullptr, "allgather_bucket_start: no communicator (vaw_comm_init)");
    VAW_CHECK_ARG(buf && count > 0 && count % g_comm.world == 0 && (dt == VAW_F32 || dt == VAW_BF16),
                  "allgather_bucket_start: the bucket must split into %d equal chunks", g_comm.world);
    if (const int rc = side_after((hipStream_t)stream)) return rc;
    const int64_t chunk = count / g_comm.world;
    const char* mine = (const char*)buf + (size_t)g_comm.rank * chunk * (dt == VAW_BF16 ? 2 : 4);
    RCCL_CHECK(g_rccl.AllGather(mine, buf, (size_t)chunk, wire_type(dt), g_comm.comm, g_comm.side), "ncclAllGather");
    return mark_done();
}

// `stream` waits for every collective started so far (the host does not)
extern "C" int vaw_allreduce_bucket_wait(vaw_stream stream) {
    VAW_CHECK_ARG(g_comm.comm != nullptr, "allreduce_bucket_wait: no communicator (vaw_comm_init)");
    HIP_CHECK(hipStreamWaitEvent((hipStream_t)stream, g_comm.done, 0), "stream wait");
    return VAW_OK;
}

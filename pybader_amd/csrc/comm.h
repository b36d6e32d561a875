// comm.h -- the multi-GPU transport of libbader_hip.so: RCCL over xGMI through the C ABI (no PyTorch).
// Included by bader_hip.hip (one translation unit), after xb_ctx is defined.
//
// One process per GPU.  librccl is dlopen'ed on first use, so a single-GPU process never needs it.  Everything is
// enqueued on the context's own stream: plane exchanges are ncclSend/ncclRecv pairs inside one ncclGroup (xGMI is
// point to point: a slab talks to its two ring neighbours only), the small collectives (counters, maxima tables,
// brick masks) are one ncclAllReduce / ncclAllGather / ncclBroadcast each on device staging buffers.
// Replaces the plane traffic the reference never had (its threads share one address space,
// thread_handlers.py:28-58, 154-205).
#pragma once
#include <dlfcn.h>

namespace xbcomm {
// the slice of the NCCL API used here (rccl.h), bound by dlsym
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;
enum { ncclSuccess = 0 };
enum { ncclInt8 = 0, ncclInt32 = 2, ncclInt64 = 4 };
enum { ncclSum = 0, ncclMax = 2, ncclMin = 3 };
struct Api {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    // what the communicator itself says about the run (xb_comm_info); optional: an older librccl may lack one
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*CommCuDevice)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*GetVersion)(int *) = nullptr;
    std::string err;
};
static Api g_api;
static bool load_api() {
    Api &a = g_api;
    if (a.lib) return true;
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
        a.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (a.lib) break;
    }
    if (!a.lib) { a.err = std::string("dlopen(librccl) failed: ") + dlerror(); return false; }
    bool ok = true;
    auto bind = [&](const char *sym, void **slot) {
        *slot = dlsym(a.lib, sym);
        if (!*slot) { ok = false; a.err = std::string("librccl lacks ") + sym; }
    };
    bind("ncclGetUniqueId", (void **)&a.GetUniqueId);
    bind("ncclCommInitRank", (void **)&a.CommInitRank);
    bind("ncclCommDestroy", (void **)&a.CommDestroy);
    bind("ncclGetErrorString", (void **)&a.GetErrorString);
    bind("ncclSend", (void **)&a.Send);
    bind("ncclRecv", (void **)&a.Recv);
    bind("ncclGroupStart", (void **)&a.GroupStart);
    bind("ncclGroupEnd", (void **)&a.GroupEnd);
    bind("ncclAllReduce", (void **)&a.AllReduce);
    bind("ncclAllGather", (void **)&a.AllGather);
    bind("ncclBroadcast", (void **)&a.Broadcast);
    a.CommCount = (decltype(a.CommCount))dlsym(a.lib, "ncclCommCount");
    a.CommUserRank = (decltype(a.CommUserRank))dlsym(a.lib, "ncclCommUserRank");
    a.CommCuDevice = (decltype(a.CommCuDevice))dlsym(a.lib, "ncclCommCuDevice");
    a.GetVersion = (decltype(a.GetVersion))dlsym(a.lib, "ncclGetVersion");
    if (!ok) { dlclose(a.lib); a.lib = nullptr; }
    return ok;
}
struct State {
    ncclComm_t comm = nullptr;
    int rank = 0, size = 1;
    int64_t *dbuf = nullptr;   // device staging for the small collectives
    int64_t *hbuf = nullptr;   // pinned
    size_t cap = 0;            // int64 entries of each
    unsigned long long bytes_sent = 0;   // plane bytes this rank has sent (xb_comm_stats)
};
}  // namespace xbcomm

#define NCCLCHK(x)                                                                                                   \
    do {                                                                                                             \
        xbcomm::ncclResult_t r_ = (x);                                                                               \
        if (r_ != xbcomm::ncclSuccess)                                                                               \
            return fail(XB_E_COMM, "%s:%d %s: %s", __FILE__, __LINE__, #x, xbcomm::g_api.GetErrorString ? xbcomm::g_api.GetErrorString(r_) : "?"); \
    } while (0)

// Inside an ncclGroup an early return would leave the group OPEN, and the next collective of this rank would block for
// ever: calls between GroupStart and GroupEnd record the first error instead, GroupEnd always runs, and the
// error is reported after it.
struct GroupErr {
    xbcomm::ncclResult_t first = xbcomm::ncclSuccess;
    const char *what = "";
    void see(xbcomm::ncclResult_t r, const char *w) { if (r != xbcomm::ncclSuccess && first == xbcomm::ncclSuccess) { first = r; what = w; } }
};
#define NCCL_GROUP_END(ge, who)                                                                                       \
    do {                                                                                                              \
        (ge).see(xbcomm::g_api.GroupEnd(), "ncclGroupEnd");                                                           \
        if ((ge).first != xbcomm::ncclSuccess)                                                                        \
            return fail(XB_E_COMM, "%s: %s: %s", who, (ge).what, xbcomm::g_api.GetErrorString ? xbcomm::g_api.GetErrorString((ge).first) : "?"); \
    } while (0)

static int comm_need(xb_ctx *c, const char *who) {
    if (!c || !c->comm || !c->comm->comm) return fail(XB_E_STATE, "%s: call xb_comm_init first", who);
    HIPCHK(hipSetDevice(c->device));
    return XB_OK;
}
static int comm_stage(xb_ctx *c, size_t n) {   // staging for n int64 in + n * size out
    xbcomm::State &s = *c->comm;
    const size_t need = n * (size_t)(s.size + 1) + 16;
    if (need <= s.cap) return XB_OK;
    hipFree(s.dbuf); hipHostFree(s.hbuf); s.dbuf = nullptr; s.hbuf = nullptr; s.cap = 0;
    const size_t cap = std::max<size_t>(need, 1 << 16);
    HIPCHK(hipMalloc(&s.dbuf, cap * sizeof(int64_t)));
    HIPCHK(hipHostMalloc(&s.hbuf, cap * sizeof(int64_t)));
    s.cap = cap;
    return XB_OK;
}

extern "C" {

// rank 0 makes the id; the host side hands its 128 bytes to every rank (any byte transport will do)
int xb_comm_unique_id(uint8_t id_out[128]) {
    if (!xbcomm::load_api()) return fail(XB_E_COMM, "xb_comm_unique_id: %s", xbcomm::g_api.err.c_str());
    xbcomm::ncclUniqueId id;
    NCCLCHK(xbcomm::g_api.GetUniqueId(&id));
    memcpy(id_out, id.internal, 128);
    return XB_OK;
}
int xb_comm_init(xb_ctx *c, int rank, int nranks, const uint8_t id_in[128]) {
    if (!c || !id_in || nranks < 1 || rank < 0 || rank >= nranks) return fail(XB_E_ARG, "xb_comm_init: bad argument");
    if (!xbcomm::load_api()) return fail(XB_E_COMM, "xb_comm_init: %s", xbcomm::g_api.err.c_str());
    HIPCHK(hipSetDevice(c->device));
    if (!c->comm) c->comm = new xbcomm::State();
    if (c->comm->comm) return fail(XB_E_STATE, "xb_comm_init: already initialised");
    xbcomm::ncclUniqueId id;
    memcpy(id.internal, id_in, 128);
    NCCLCHK(xbcomm::g_api.CommInitRank(&c->comm->comm, nranks, id, rank));
    c->comm->rank = rank;
    c->comm->size = nranks;
    return XB_OK;
}
int xb_comm_destroy(xb_ctx *c) {
    if (!c || !c->comm) return XB_OK;
    hipSetDevice(c->device);
    hipStreamSynchronize(c->stream);
    if (c->comm->comm && xbcomm::g_api.CommDestroy) xbcomm::g_api.CommDestroy(c->comm->comm);
    hipFree(c->comm->dbuf);
    hipHostFree(c->comm->hbuf);
    delete c->comm;
    c->comm = nullptr;
    return XB_OK;
}

// Whole planes [xa, xb) of the label (which = 0, int32) or known (which = 1, int8) array between ranks: every send
// has a matching recv posted by the peer with the same plane range (slab.halo_plan builds both lists from the same
// table).  One group: RCCL runs the transfers concurrently, each over its own xGMI link.
int xb_comm_exchange_planes(xb_ctx *c, int which, int n_send, const int32_t *send_peer, const int64_t *send_xa,
                            const int64_t *send_xb, int n_recv, const int32_t *recv_peer, const int64_t *recv_xa,
                            const int64_t *recv_xb) {
    if (int rc = comm_need(c, "xb_comm_exchange_planes")) return rc;
    if (!c->has_grid) return fail(XB_E_STATE, "xb_comm_exchange_planes: no grid");
    if (which != 0 && which != 1) return fail(XB_E_ARG, "xb_comm_exchange_planes: which must be 0 (labels) or 1 (known)");
    if (which == 0) if (int rc = settle_labels(c)) return rc;
    const Grid &g = c->g;
    const size_t es = which == 0 ? 4 : 1;
    char *base = which == 0 ? (char *)c->labels : (char *)c->known;
    const int dt = which == 0 ? xbcomm::ncclInt32 : xbcomm::ncclInt8;
    // (a send to oneself is refused: no slab is its own neighbour -- except for the one-GPU test of this very path, option 19)
    auto bad = [&](int peer, int64_t xa, int64_t xb) {
        return peer < 0 || peer >= c->comm->size || (peer == c->comm->rank && !c->opt_self_exchange) || xa < 0 || xb > g.nx || xa >= xb;
    };
    for (int i = 0; i < n_send; i++) if (bad(send_peer[i], send_xa[i], send_xb[i])) return fail(XB_E_ARG, "xb_comm_exchange_planes: bad send %d", i);
    for (int i = 0; i < n_recv; i++) if (bad(recv_peer[i], recv_xa[i], recv_xb[i])) return fail(XB_E_ARG, "xb_comm_exchange_planes: bad recv %d", i);
    if (n_recv) { c->list_valid = false; c->chg_n = -1; c->buni_valid = c->buni_valid && c->buni_halo_safe; }   // halo planes of peers that ran the same assignment
    // Label planes travel in the narrowest signed type that holds every label (the reference's own dtype_calc(-n_maxima),
    // thread_handlers.py:70-74: int8 for every BASELINE configuration -- a quarter of the int32 bytes per halo): packed into
    // the staging buffer before the group, widened back behind it, all on the context's stream.
    // (the width follows whoever wrote the resident labels last -- xb_ctx::label_wire -- not the last assignment's basin count:
    // an uploaded map with more basins than the assignment before it must not be truncated)
    const int wire = (which == 0 && c->opt_narrow_halo) ? c->label_wire : (int)es;
    size_t total = 0;
    for (int i = 0; i < n_send; i++) total += (size_t)(send_xb[i] - send_xa[i]) * g.nyz;
    for (int i = 0; i < n_recv; i++) total += (size_t)(recv_xb[i] - recv_xa[i]) * g.nyz;
    const bool packed = which == 0 && wire < 4 && total * wire <= c->stage_bytes;
    char *pk = (char *)c->stage;
    std::vector<size_t> off_s(n_send), off_r(n_recv);
    if (packed) {
        size_t o = 0;
        for (int i = 0; i < n_send; i++) {
            off_s[i] = o;
            const long long n = (long long)(send_xb[i] - send_xa[i]) * g.nyz;
            const int *src = c->labels + (size_t)send_xa[i] * g.nyz;
            if (wire == 1) k_narrow<int8_t><<<nblocks(n), TPB, 0, c->stream>>>(src, (int8_t *)(pk + o), n);
            else k_narrow<int16_t><<<nblocks(n), TPB, 0, c->stream>>>(src, (int16_t *)(pk + o), n);
            o += ((size_t)n * wire + 15) & ~(size_t)15;
        }
        for (int i = 0; i < n_recv; i++) {
            off_r[i] = o;
            o += ((size_t)(recv_xb[i] - recv_xa[i]) * g.nyz * wire + 15) & ~(size_t)15;
        }
        HIPCHK(hipGetLastError());
    }
    const int wdt = packed ? xbcomm::ncclInt8 : dt;           // packed planes travel as bytes (send / recv only move them)
    const size_t wmul = packed ? (size_t)wire : 1;            // ... so their element count is a byte count
    NCCLCHK(xbcomm::g_api.GroupStart());
    GroupErr ge;
    for (int i = 0; i < n_recv; i++)
        ge.see(xbcomm::g_api.Recv(packed ? pk + off_r[i] : base + (size_t)recv_xa[i] * g.nyz * es, (size_t)(recv_xb[i] - recv_xa[i]) * g.nyz * wmul, wdt,
                                  recv_peer[i], c->comm->comm, c->stream), "ncclRecv");
    for (int i = 0; i < n_send; i++)
        ge.see(xbcomm::g_api.Send(packed ? pk + off_s[i] : base + (size_t)send_xa[i] * g.nyz * es, (size_t)(send_xb[i] - send_xa[i]) * g.nyz * wmul, wdt,
                                  send_peer[i], c->comm->comm, c->stream), "ncclSend");
    NCCL_GROUP_END(ge, "xb_comm_exchange_planes");
    if (packed) {
        for (int i = 0; i < n_recv; i++) {
            const long long n = (long long)(recv_xb[i] - recv_xa[i]) * g.nyz;
            int *dst = c->labels + (size_t)recv_xa[i] * g.nyz;
            if (wire == 1) k_widen<int8_t><<<nblocks(n), TPB, 0, c->stream>>>((const int8_t *)(pk + off_r[i]), dst, n);
            else k_widen<int16_t><<<nblocks(n), TPB, 0, c->stream>>>((const int16_t *)(pk + off_r[i]), dst, n);
        }
        HIPCHK(hipGetLastError());
    }
    {
        size_t t = 0;
        for (int i = 0; i < n_send; i++) t += (size_t)(send_xb[i] - send_xa[i]) * g.nyz;
        c->comm->bytes_sent += (unsigned long long)t * (packed ? (size_t)wire : es);
    }
    if (!c->opt_async_comm) HIPCHK(hipStreamSynchronize(c->stream));   // (option 24: whatever reads the planes next is ordered behind them)
    return XB_OK;
}

// plane bytes this rank has sent since xb_comm_init (the narrowed halos show here)
// out[0..3]: ncclCommCount, ncclCommUserRank, ncclCommCuDevice of this rank's communicator and ncclGetVersion -- the
// communicator's own word that the run saw N ranks on N devices (-1: that entry point is missing in this librccl)
int xb_comm_info(xb_ctx *c, int64_t out[4]) {
    if (int rc = comm_need(c, "xb_comm_info")) return rc;
    int v[4] = {-1, -1, -1, -1};
    const xbcomm::Api &a = xbcomm::g_api;
    if (a.CommCount) NCCLCHK(a.CommCount(c->comm->comm, &v[0]));
    if (a.CommUserRank) NCCLCHK(a.CommUserRank(c->comm->comm, &v[1]));
    if (a.CommCuDevice) NCCLCHK(a.CommCuDevice(c->comm->comm, &v[2]));
    if (a.GetVersion) NCCLCHK(a.GetVersion(&v[3]));
    for (int i = 0; i < 4; i++) out[i] = v[i];
    return XB_OK;
}
int xb_comm_stats(xb_ctx *c, int64_t *bytes_sent) {
    if (!c || !c->comm) return fail(XB_E_STATE, "xb_comm_stats: call xb_comm_init first");
    if (bytes_sent) *bytes_sent = (int64_t)c->comm->bytes_sent;
    return XB_OK;
}

// n int64 values reduced over all ranks in place (op: 0 sum, 1 min, 2 max) -- counters of a refinement iteration
int xb_comm_allreduce_i64(xb_ctx *c, int64_t *inout, int64_t n, int op) {
    if (int rc = comm_need(c, "xb_comm_allreduce_i64")) return rc;
    if (n <= 0) return XB_OK;
    if (op < 0 || op > 2) return fail(XB_E_ARG, "xb_comm_allreduce_i64: bad op");
    if (int rc = comm_stage(c, (size_t)n)) return rc;
    xbcomm::State &s = *c->comm;
    memcpy(s.hbuf, inout, n * sizeof(int64_t));
    HIPCHK(hipMemcpyAsync(s.dbuf, s.hbuf, n * sizeof(int64_t), hipMemcpyHostToDevice, c->stream));
    const int nop = op == 0 ? xbcomm::ncclSum : (op == 1 ? xbcomm::ncclMin : xbcomm::ncclMax);
    NCCLCHK(xbcomm::g_api.AllReduce(s.dbuf, s.dbuf, (size_t)n, xbcomm::ncclInt64, nop, s.comm, c->stream));
    HIPCHK(hipMemcpyAsync(s.hbuf, s.dbuf, n * sizeof(int64_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    memcpy(inout, s.hbuf, n * sizeof(int64_t));
    return XB_OK;
}
// every rank contributes n int64 values; out receives size * n, in rank order -- maxima tables, seeds, path queries
int xb_comm_allgather_i64(xb_ctx *c, const int64_t *in, int64_t n, int64_t *out) {
    if (int rc = comm_need(c, "xb_comm_allgather_i64")) return rc;
    if (n <= 0) return XB_OK;
    if (int rc = comm_stage(c, (size_t)n)) return rc;
    xbcomm::State &s = *c->comm;
    memcpy(s.hbuf, in, n * sizeof(int64_t));
    HIPCHK(hipMemcpyAsync(s.dbuf, s.hbuf, n * sizeof(int64_t), hipMemcpyHostToDevice, c->stream));
    NCCLCHK(xbcomm::g_api.AllGather(s.dbuf, s.dbuf + n, (size_t)n, xbcomm::ncclInt64, s.comm, c->stream));
    HIPCHK(hipMemcpyAsync(s.hbuf + n, s.dbuf + n, (size_t)n * s.size * sizeof(int64_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    memcpy(out, s.hbuf + n, (size_t)n * s.size * sizeof(int64_t));
    return XB_OK;
}
// the per-brick move masks (xb_brick_masks): rank r computed the bricks [first[r], first[r] + count[r]); afterwards
// every rank holds every chunk.  One broadcast per rank inside one group (chunks differ in size).
int xb_comm_share_brick_masks(xb_ctx *c, const int64_t *first, const int64_t *count) {
    if (int rc = comm_need(c, "xb_comm_share_brick_masks")) return rc;
    if (!c->has_grid) return fail(XB_E_STATE, "xb_comm_share_brick_masks: no grid");
    const int64_t nbr = c->N / 512;
    int *masks = c->list + nbr, *maxvox = c->list + 4 * nbr;   // move masks; the brick's single maximum (k_brick_masks)
    for (int r = 0; r < c->comm->size; r++)
        if (first[r] < 0 || count[r] < 0 || first[r] + count[r] > nbr) return fail(XB_E_ARG, "xb_comm_share_brick_masks: bad chunk of rank %d", r);
    NCCLCHK(xbcomm::g_api.GroupStart());
    GroupErr ge;
    for (int r = 0; r < c->comm->size; r++)
        if (count[r]) {
            ge.see(xbcomm::g_api.Broadcast(masks + first[r], masks + first[r], (size_t)count[r], xbcomm::ncclInt32, r, c->comm->comm, c->stream), "ncclBroadcast");
            ge.see(xbcomm::g_api.Broadcast(maxvox + first[r], maxvox + first[r], (size_t)count[r], xbcomm::ncclInt32, r, c->comm->comm, c->stream), "ncclBroadcast");
        }
    NCCL_GROUP_END(ge, "xb_comm_share_brick_masks");
    HIPCHK(hipStreamSynchronize(c->stream));
    return XB_OK;
}

}  // extern "C"

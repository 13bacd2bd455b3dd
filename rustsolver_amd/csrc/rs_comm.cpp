// rs_comm.cpp -- the one collective of the path: an RCCL (xGMI) all-reduce over the REPLICATED rounds.
// Boards shard across GPUs (one process per GPU, one rs_table per process); rounds whose boards are
// not sharded are replicated and every rank accumulates its own deltas into them.  After a local
// iteration:   x = snapshot + allreduce_sum(x - snapshot)   for regrets and strategy_sum.
// Integer deltas are summed as ncclInt32 (wrapping, order-independent => identical on every rank
// and for every rank count); f32 tables as ncclFloat (order-dependent in the last bits).
//
// RCCL is loaded lazily with dlopen so that the core library has no hard dependency on it and
// single-GPU users never pay for it.
#include <dlfcn.h>

#include <cstring>
#include <new>

#include "rs_internal.hpp"

using namespace rs;

namespace {

// the handful of RCCL entry points used here (signatures from <rccl/rccl.h>)
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
enum { ncclSuccess_ = 0 };
enum { ncclInt32_ = 2, ncclFloat32_ = 7 };  // ncclDataType_t
enum { ncclSum_ = 0 };                      // ncclRedOp_t

struct Rccl {
    void *handle = nullptr;
    int (*GetUniqueId)(ncclUniqueId *) = nullptr;
    int (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};

Rccl *rccl() {
    static Rccl r;
    static bool tried = false;
    if (tried) return r.handle ? &r : nullptr;
    tried = true;
    if (const char *over = rccl_library_override()) r.handle = dlopen(over, RTLD_NOW | RTLD_LOCAL);   // tests only; no fallback to the real library: a wrong path must fail loudly
    else
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (r.handle) break;
        }
    if (!r.handle) return nullptr;
    r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.handle, "ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.handle, "ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.handle, "ncclCommDestroy");
    r.AllReduce = (decltype(r.AllReduce))dlsym(r.handle, "ncclAllReduce");
    r.AllGather = (decltype(r.AllGather))dlsym(r.handle, "ncclAllGather");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.handle, "ncclGetErrorString");
    if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllReduce || !r.AllGather || !r.GetErrorString) {
        dlclose(r.handle);
        r.handle = nullptr;
        return nullptr;
    }
    return &r;
}

int comm_fail(int code, const char *what) {
    Rccl *r = rccl();
    return fail(RS_ERR_COMM, std::string(what) + ": " + (r ? r->GetErrorString(code) : "RCCL not loaded"));
}

}  // namespace

struct rs_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, n_ranks = 1;
    int device = 0;
};

extern "C" {

int rs_comm_unique_id(void *id_out) {
    if (!id_out) return fail(RS_ERR_INVALID, "rs_comm_unique_id: id_out is NULL");
    Rccl *r = rccl();
    if (!r) return fail(RS_ERR_COMM, "rs_comm_unique_id: librccl.so could not be loaded");
    static_assert(sizeof(ncclUniqueId) == RS_COMM_ID_BYTES, "ncclUniqueId size");
    ncclUniqueId id;
    int rc = r->GetUniqueId(&id);
    if (rc != ncclSuccess_) return comm_fail(rc, "ncclGetUniqueId");
    std::memcpy(id_out, &id, sizeof(id));
    return RS_OK;
}

int rs_comm_create(rs_table *table, const void *id, int rank, int n_ranks, rs_comm **out) {
    if (!table || !id || !out || n_ranks < 1 || rank < 0 || rank >= n_ranks)
        return fail(RS_ERR_INVALID, "rs_comm_create: bad argument");
    Rccl *r = rccl();
    if (!r) return fail(RS_ERR_COMM, "rs_comm_create: librccl.so could not be loaded");
    hipError_t e = hipSetDevice(table->device);
    if (e != hipSuccess) return hip_fail(e, "hipSetDevice");
    rs_comm *c = new (std::nothrow) rs_comm();
    if (!c) return fail(RS_ERR_OOM, "rs_comm_create: out of host memory");
    ncclUniqueId uid;
    std::memcpy(&uid, id, sizeof(uid));
    int rc = r->CommInitRank(&c->comm, n_ranks, uid, rank);
    if (rc != ncclSuccess_) {
        delete c;
        return comm_fail(rc, "ncclCommInitRank");
    }
    c->rank = rank;
    c->n_ranks = n_ranks;
    c->device = table->device;
    *out = c;
    return RS_OK;
}

void rs_comm_destroy(rs_comm *c) {
    if (!c) return;
    Rccl *r = rccl();
    if (r && c->comm) {
        (void)hipSetDevice(c->device);
        (void)r->CommDestroy(c->comm);
    }
    delete c;
}

// in-place all-gather of the exchange buffer [n_ranks][bytes_per_rank]: rank r contributes its r-th slot (xGMI: every rank
// sends its slot to 7 peers over dedicated links; the slots are ~MBs, so the call is latency-bound)
int rs_comm_allgather(rs_comm *c, rs_table *t, void *d_buf, size_t bytes_per_rank) {
    if (!c || !t || !d_buf) return fail(RS_ERR_INVALID, "rs_comm_allgather: NULL argument");
    if (bytes_per_rank % 4 != 0) return fail(RS_ERR_INVALID, "rs_comm_allgather: slot size must be a multiple of 4 bytes");
    Rccl *r = rccl();
    if (!r) return fail(RS_ERR_COMM, "rs_comm_allgather: librccl.so could not be loaded");
    hipError_t e = hipSetDevice(t->device);
    if (e != hipSuccess) return hip_fail(e, "hipSetDevice");
    if (bytes_per_rank == 0) return RS_OK;
    const int rc = r->AllGather((const char *)d_buf + size_t(c->rank) * bytes_per_rank, d_buf, bytes_per_rank / 4, ncclFloat32_, c->comm, t->stream);
    if (rc != ncclSuccess_) return comm_fail(rc, "ncclAllGather");
    return RS_OK;
}

// data-parallel deal batches: every rank swept its own deals into its delta tables; one in-place wrapping sum (ncclInt32) per
// array makes them the deltas of the UNION batch on every rank (integer adds commute: any rank count, any order, same bits)
int rs_comm_allreduce_deltas(rs_comm *c, rs_table *t) {
    if (!c || !t) return fail(RS_ERR_INVALID, "rs_comm_allreduce_deltas: NULL argument");
    if (!t->d_dregrets || !t->d_dssum) return fail(RS_ERR_INVALID, "rs_comm_allreduce_deltas: the table has no delta tables (rs_solver_create_deals makes them)");
    Rccl *r = rccl();
    if (!r) return fail(RS_ERR_COMM, "rs_comm_allreduce_deltas: librccl.so could not be loaded");
    hipError_t e = hipSetDevice(t->device);
    if (e != hipSuccess) return hip_fail(e, "hipSetDevice");
    int rc = r->AllReduce(t->d_dregrets, t->d_dregrets, t->n_cells, ncclInt32_, ncclSum_, c->comm, t->stream);
    if (rc == ncclSuccess_) rc = r->AllReduce(t->d_dssum, t->d_dssum, t->n_cells, ncclInt32_, ncclSum_, c->comm, t->stream);
    if (rc != ncclSuccess_) return comm_fail(rc, "ncclAllReduce(deal deltas)");
    return RS_OK;
}

}  // extern "C" (interrupted: the solver's own collectives)
namespace rs {
int comm_world(const rs_comm *c) { return c ? c->n_ranks : 1; }
int comm_rank(const rs_comm *c) { return c ? c->rank : 0; }
// in-place wrapping sum of n ints over the ranks
int comm_allreduce_i32(rs_comm *c, rs_table *t, void *d_buf, size_t n) {
    Rccl *r = rccl();
    if (!c || !t || !r) return fail(RS_ERR_COMM, "comm_allreduce_i32: no communicator / librccl.so could not be loaded");
    if (n == 0) return RS_OK;
    const int rc = r->AllReduce(d_buf, d_buf, n, ncclInt32_, ncclSum_, c->comm, t->stream);
    return rc == ncclSuccess_ ? RS_OK : comm_fail(rc, "ncclAllReduce(packed deal deltas)");
}
// every rank's n words, gathered in rank order: d_recv [n_ranks][n]
int comm_allgather_u32(rs_comm *c, rs_table *t, const void *d_send, void *d_recv, size_t n) {
    Rccl *r = rccl();
    if (!c || !t || !r) return fail(RS_ERR_COMM, "comm_allgather_u32: no communicator / librccl.so could not be loaded");
    if (n == 0) return RS_OK;
    const int rc = r->AllGather(d_send, d_recv, n, ncclInt32_, c->comm, t->stream);
    return rc == ncclSuccess_ ? RS_OK : comm_fail(rc, "ncclAllGather(deal delta items)");
}
}  // namespace rs
extern "C" {

int rs_replicated_begin(rs_table *t, uint32_t round_mask) {
    if (!t) return fail(RS_ERR_INVALID, "rs_replicated_begin: table is NULL");
    if (t->dtype == RS_F16) return fail(RS_ERR_UNSUPPORTED, "rs_replicated_begin: RS_F16 tables are not reduced (use RS_F32 accumulators)");
    hipError_t e = hipSetDevice(t->device);
    if (e != hipSuccess) return hip_fail(e, "hipSetDevice");
    if (t->rep_mask != round_mask || t->rep_nodes.empty()) {
        t->rep_mask = 0;   // set again only once both snapshots exist: a failed allocation must not leave a state a retry would accept
        t->rep_nodes.clear();
        t->rep_off.clear();
        t->rep_cells = 0;
        for (int n = 0; n < int(t->nodes.size()); ++n)
            if (round_mask & (1u << t->nodes[n].round_idx)) {
                t->rep_nodes.push_back(n);
                t->rep_off.push_back(t->rep_cells);
                t->rep_cells += t->pitch[n] * t->nodes[n].n_actions;
            }
        if (t->d_snap_regrets) (void)hipFree(t->d_snap_regrets);
        if (t->d_snap_ssum) (void)hipFree(t->d_snap_ssum);
        t->d_snap_regrets = t->d_snap_ssum = nullptr;
        if (t->rep_cells) {
            if ((e = hipMalloc(&t->d_snap_regrets, t->rep_cells * 4)) != hipSuccess ||
                (e = hipMalloc(&t->d_snap_ssum, t->rep_cells * 4)) != hipSuccess) {
                if (t->d_snap_regrets) (void)hipFree(t->d_snap_regrets);
                if (t->d_snap_ssum) (void)hipFree(t->d_snap_ssum);
                t->d_snap_regrets = t->d_snap_ssum = nullptr;
                t->rep_nodes.clear();
                t->rep_off.clear();
                t->rep_cells = 0;
                return hip_fail(e, "rs_replicated_begin: snapshot hipMalloc");
            }
        }
        t->rep_mask = round_mask;
    }
    for (size_t i = 0; i < t->rep_nodes.size(); ++i) {
        const int n = t->rep_nodes[i];
        const size_t bytes = t->pitch[n] * t->nodes[n].n_actions * 4;
        if ((e = hipMemcpyAsync((char *)t->d_snap_regrets + t->rep_off[i] * 4, t->regrets_ptr(n), bytes,
                                hipMemcpyDeviceToDevice, t->stream)) != hipSuccess ||
            (e = hipMemcpyAsync((char *)t->d_snap_ssum + t->rep_off[i] * 4, t->ssum_ptr(n), bytes, hipMemcpyDeviceToDevice,
                                t->stream)) != hipSuccess)
            return hip_fail(e, "rs_replicated_begin: snapshot copy");
    }
    return RS_OK;
}

int rs_allreduce_replicated(rs_table *t, rs_comm *c, uint32_t round_mask) {
    if (!t || !c) return fail(RS_ERR_INVALID, "rs_allreduce_replicated: NULL argument");
    if (t->rep_mask != round_mask || t->rep_nodes.empty())
        return fail(RS_ERR_INVALID, "rs_allreduce_replicated: call rs_replicated_begin with the same round_mask first");
    ++t->epoch;
    Rccl *r = rccl();
    if (!r) return fail(RS_ERR_COMM, "rs_allreduce_replicated: librccl.so could not be loaded");
    hipError_t e = hipSetDevice(t->device);
    if (e != hipSuccess) return hip_fail(e, "hipSetDevice");
    if (t->rep_cells == 0) return RS_OK;
    // 1. one fused pass per array: snap := snap - x (= minus the rank's own delta, contiguous) and x := the snapshot itself, bit for bit -- in f32
    //    x + (snap - x) is not snap and its error depends on the rank's own x, which would let the replicated rounds drift apart across ranks
    for (size_t i = 0; i < t->rep_nodes.size(); ++i) {
        const int n = t->rep_nodes[i];
        const size_t cells = t->pitch[n] * t->nodes[n].n_actions;
        void *sr = (char *)t->d_snap_regrets + t->rep_off[i] * 4, *ss = (char *)t->d_snap_ssum + t->rep_off[i] * 4;
        if ((e = launch_delta_swap(t->regrets_ptr(n), sr, cells, t->dtype, t->stream)) != hipSuccess ||
            (e = launch_delta_swap(t->ssum_ptr(n), ss, cells, t->dtype, t->stream)) != hipSuccess)
            return hip_fail(e, "rs_allreduce_replicated: delta kernels");
    }
    // 2. one all-reduce per array over the contiguous (negated) deltas
    const int dt = t->dtype == RS_I32 ? ncclInt32_ : ncclFloat32_;
    int rc = r->AllReduce(t->d_snap_regrets, t->d_snap_regrets, t->rep_cells, dt, ncclSum_, c->comm, t->stream);
    if (rc != ncclSuccess_) return comm_fail(rc, "ncclAllReduce(regrets)");
    rc = r->AllReduce(t->d_snap_ssum, t->d_snap_ssum, t->rep_cells, dt, ncclSum_, c->comm, t->stream);
    if (rc != ncclSuccess_) return comm_fail(rc, "ncclAllReduce(strategy_sum)");
    // 3. x := snapshot - sum(-delta)
    for (size_t i = 0; i < t->rep_nodes.size(); ++i) {
        const int n = t->rep_nodes[i];
        const size_t cells = t->pitch[n] * t->nodes[n].n_actions;
        if ((e = launch_delta_sub(t->regrets_ptr(n), (char *)t->d_snap_regrets + t->rep_off[i] * 4, cells, t->dtype, t->stream)) != hipSuccess ||
            (e = launch_delta_sub(t->ssum_ptr(n), (char *)t->d_snap_ssum + t->rep_off[i] * 4, cells, t->dtype, t->stream)) != hipSuccess)
            return hip_fail(e, "rs_allreduce_replicated: apply kernels");
    }
    return RS_OK;
}

}  // extern "C"

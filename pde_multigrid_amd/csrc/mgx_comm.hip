// mgx_comm.hip -- z-slab halo exchange and coarse-level collectives.
//
// New work relative to the reference (it is single-process, single-device; its thesis
// lists multi-GPU sub-grids as future work).  Storage is x fastest / z slowest, so a
// z-plane is one contiguous block and ghost planes need no packing: the halo exchange
// is a grouped ncclSend/ncclRecv pair with each of the (at most two) chain neighbours,
// each over its own point-to-point xGMI link, issued on the context's comm stream so that
// interior smoothing keeps running on the compute stream.
//
// Two transports behind the same entry points:
//   RCCL   one process per GPU (mgx_comm_init): the production path
//   local  several host threads of ONE process, one context each, all on the same device
//          (mgx_comm_init_local): device-to-device copies and a pthread barrier.  It exists so
//          that the slab-decomposed cycle can be checked bit-for-bit on a single-GPU box.
#include <pthread.h>
#include <rccl/rccl.h>

#include <vector>

#include "mgx_internal.hpp"

#define MGX_NCCL(expr)                                                                          \
    do {                                                                                        \
        ncclResult_t r_ = (expr);                                                               \
        if (r_ != ncclSuccess)                                                                  \
            return mgx::fail(MGX_ERR_RCCL, "%s failed: %s (%s:%d)", #expr, ncclGetErrorString(r_), \
                             __FILE__, __LINE__);                                               \
    } while (0)

static_assert(sizeof(ncclUniqueId) == MGX_UNIQUE_ID_BYTES, "MGX_UNIQUE_ID_BYTES must match ncclUniqueId");

// shared state of a local (in-process) group
struct mgx_local_group {
    int nranks = 0;
    pthread_barrier_t barrier;
    struct Post {
        const void* to_lower = nullptr;
        const void* to_upper = nullptr;
        const void* gather = nullptr;
    };
    std::vector<Post> post;
};

namespace {

// comm stream waits for everything enqueued so far on the compute stream
int order_after_compute(mgx_ctx* ctx) {
    MGX_HIP(hipEventRecord(ctx->ev_compute, ctx->compute));
    MGX_HIP(hipStreamWaitEvent(ctx->comm, ctx->ev_compute, 0));
    return MGX_OK;
}

ncclDataType_t dtype_of(int elem_bytes) { return elem_bytes == 4 ? ncclFloat32 : ncclFloat64; }

int local_halo(mgx_ctx* ctx, const void* send_lo, void* recv_lo, size_t n_from_lo, const void* send_up, void* recv_up,
               size_t n_from_up, int eb) {
    mgx_local_group* g = (mgx_local_group*)ctx->local_group;
    MGX_HIP(hipStreamSynchronize(ctx->compute));  // my planes are final before a neighbour copies them
    g->post[ctx->rank].to_lower = send_lo;
    g->post[ctx->rank].to_upper = send_up;
    pthread_barrier_wait(&g->barrier);
    if (ctx->rank > 0 && n_from_lo)
        MGX_HIP(hipMemcpyAsync(recv_lo, g->post[ctx->rank - 1].to_upper, n_from_lo * eb, hipMemcpyDeviceToDevice, ctx->comm));
    if (ctx->rank < ctx->nranks - 1 && n_from_up)
        MGX_HIP(hipMemcpyAsync(recv_up, g->post[ctx->rank + 1].to_lower, n_from_up * eb, hipMemcpyDeviceToDevice, ctx->comm));
    MGX_HIP(hipStreamSynchronize(ctx->comm));
    pthread_barrier_wait(&g->barrier);  // nobody overwrites a plane a neighbour is still reading
    return MGX_OK;
}

}  // namespace

extern "C" {

int mgx_comm_unique_id(void* host_id_bytes) {
    MGX_REQUIRE(host_id_bytes, MGX_ERR_INVALID, "id buffer is NULL");
    ncclUniqueId id;
    MGX_NCCL(ncclGetUniqueId(&id));
    memcpy(host_id_bytes, &id, sizeof id);
    return MGX_OK;
}

int mgx_comm_init(mgx_ctx* ctx, const void* host_id_bytes, int rank, int nranks) {
    MGX_REQUIRE(ctx && host_id_bytes, MGX_ERR_INVALID, "NULL argument");
    MGX_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, MGX_ERR_INVALID, "bad rank %d / %d", rank, nranks);
    MGX_REQUIRE(!ctx->rccl_comm && !ctx->local_group, MGX_ERR_INVALID, "communicator already initialised");
    MGX_HIP(hipSetDevice(ctx->device));
    ncclUniqueId id;
    memcpy(&id, host_id_bytes, sizeof id);
    ncclComm_t comm;
    MGX_NCCL(ncclCommInitRank(&comm, nranks, id, rank));
    ctx->rccl_comm = (void*)comm;
    ctx->rank = rank;
    ctx->nranks = nranks;
    return MGX_OK;
}

int mgx_local_group_create(int nranks, mgx_local_group** out) {
    MGX_REQUIRE(out && nranks >= 1, MGX_ERR_INVALID, "bad arguments");
    mgx_local_group* g = new mgx_local_group();
    g->nranks = nranks;
    g->post.resize(nranks);
    if (pthread_barrier_init(&g->barrier, nullptr, (unsigned)nranks) != 0) {
        delete g;
        return mgx::fail(MGX_ERR_INVALID, "pthread_barrier_init failed");
    }
    *out = g;
    return MGX_OK;
}

int mgx_local_group_destroy(mgx_local_group* g) {
    if (!g) return MGX_OK;
    pthread_barrier_destroy(&g->barrier);
    delete g;
    return MGX_OK;
}

int mgx_comm_init_local(mgx_ctx* ctx, mgx_local_group* group, int rank) {
    MGX_REQUIRE(ctx && group, MGX_ERR_INVALID, "NULL argument");
    MGX_REQUIRE(rank >= 0 && rank < group->nranks, MGX_ERR_INVALID, "bad rank %d / %d", rank, group->nranks);
    MGX_REQUIRE(!ctx->rccl_comm && !ctx->local_group, MGX_ERR_INVALID, "communicator already initialised");
    ctx->local_group = group;
    ctx->rank = rank;
    ctx->nranks = group->nranks;
    return MGX_OK;
}

int mgx_comm_destroy(mgx_ctx* ctx) {
    MGX_REQUIRE(ctx, MGX_ERR_INVALID, "ctx is NULL");
    if (ctx->rccl_comm) {
        (void)hipStreamSynchronize(ctx->comm);
        ncclCommDestroy((ncclComm_t)ctx->rccl_comm);
        ctx->rccl_comm = nullptr;
    }
    ctx->local_group = nullptr;
    ctx->rank = 0;
    ctx->nranks = 1;
    return MGX_OK;
}

int mgx_comm_rank(const mgx_ctx* ctx, int* rank, int* nranks) {
    MGX_REQUIRE(ctx && rank && nranks, MGX_ERR_INVALID, "NULL argument");
    *rank = ctx->rank;
    *nranks = ctx->nranks;
    return MGX_OK;
}

int mgx_comm_halo_exchange(mgx_ctx* ctx, const void* send_to_lower, size_t count_to_lower, void* recv_from_lower,
                           size_t count_from_lower, const void* send_to_upper, size_t count_to_upper, void* recv_from_upper,
                           size_t count_from_upper, int elem_bytes) {
    MGX_REQUIRE(ctx, MGX_ERR_INVALID, "ctx is NULL");
    MGX_REQUIRE(elem_bytes == 4 || elem_bytes == 8, MGX_ERR_INVALID, "elem_bytes = %d", elem_bytes);
    if (ctx->nranks == 1) return MGX_OK;
    const bool has_lo = ctx->rank > 0, has_up = ctx->rank < ctx->nranks - 1;
    MGX_REQUIRE(!has_lo || ((send_to_lower || !count_to_lower) && (recv_from_lower || !count_from_lower)), MGX_ERR_INVALID,
                "lower-neighbour buffers are NULL");
    MGX_REQUIRE(!has_up || ((send_to_upper || !count_to_upper) && (recv_from_upper || !count_from_upper)), MGX_ERR_INVALID,
                "upper-neighbour buffers are NULL");
    if (ctx->local_group)
        return local_halo(ctx, send_to_lower, recv_from_lower, count_from_lower, send_to_upper, recv_from_upper,
                          count_from_upper, elem_bytes);
    MGX_REQUIRE(ctx->rccl_comm, MGX_ERR_RCCL, "communicator not initialised");
    int st = order_after_compute(ctx);
    if (st) return st;
    ncclComm_t comm = (ncclComm_t)ctx->rccl_comm;
    const ncclDataType_t dt = dtype_of(elem_bytes);
    MGX_NCCL(ncclGroupStart());
    if (has_lo) {
        if (count_to_lower) MGX_NCCL(ncclSend(send_to_lower, count_to_lower, dt, ctx->rank - 1, comm, ctx->comm));
        if (count_from_lower) MGX_NCCL(ncclRecv(recv_from_lower, count_from_lower, dt, ctx->rank - 1, comm, ctx->comm));
    }
    if (has_up) {
        if (count_to_upper) MGX_NCCL(ncclSend(send_to_upper, count_to_upper, dt, ctx->rank + 1, comm, ctx->comm));
        if (count_from_upper) MGX_NCCL(ncclRecv(recv_from_upper, count_from_upper, dt, ctx->rank + 1, comm, ctx->comm));
    }
    MGX_NCCL(ncclGroupEnd());
    return MGX_OK;
}

int mgx_comm_wait(mgx_ctx* ctx) {
    MGX_REQUIRE(ctx, MGX_ERR_INVALID, "ctx is NULL");
    if (ctx->nranks == 1 || ctx->local_group) return MGX_OK;  // the local transport completes inside the call
    MGX_HIP(hipEventRecord(ctx->ev_comm, ctx->comm));
    MGX_HIP(hipStreamWaitEvent(ctx->compute, ctx->ev_comm, 0));
    return MGX_OK;
}

int mgx_comm_allgather(mgx_ctx* ctx, const void* send, void* recv, size_t count, int elem_bytes) {
    MGX_REQUIRE(ctx && send && recv, MGX_ERR_INVALID, "NULL argument");
    MGX_REQUIRE(elem_bytes == 4 || elem_bytes == 8, MGX_ERR_INVALID, "elem_bytes = %d", elem_bytes);
    if (ctx->nranks == 1) {
        if (send != recv) MGX_HIP(hipMemcpyAsync(recv, send, count * elem_bytes, hipMemcpyDeviceToDevice, ctx->compute));
        return MGX_OK;
    }
    if (ctx->local_group) {
        mgx_local_group* g = (mgx_local_group*)ctx->local_group;
        MGX_HIP(hipStreamSynchronize(ctx->compute));
        g->post[ctx->rank].gather = send;
        pthread_barrier_wait(&g->barrier);
        for (int r = 0; r < ctx->nranks; r++)
            MGX_HIP(hipMemcpyAsync((char*)recv + (size_t)r * count * elem_bytes, g->post[r].gather, count * elem_bytes,
                                   hipMemcpyDeviceToDevice, ctx->comm));
        MGX_HIP(hipStreamSynchronize(ctx->comm));
        pthread_barrier_wait(&g->barrier);
        return MGX_OK;
    }
    MGX_REQUIRE(ctx->rccl_comm, MGX_ERR_RCCL, "communicator not initialised");
    int st = order_after_compute(ctx);
    if (st) return st;
    MGX_NCCL(ncclAllGather(send, recv, count, dtype_of(elem_bytes), (ncclComm_t)ctx->rccl_comm, ctx->comm));
    return MGX_OK;
}

// Grouped ncclSend/ncclRecv of `count` doubles from this rank to itself on the comm stream, then an
// all-gather and an all-reduce: exercises RCCL linkage, communicator and stream/event plumbing on a box
// with a single GPU (nranks may be 1).  dev_src and dev_dst must not overlap.
int mgx_comm_selftest(mgx_ctx* ctx, const double* dev_src, double* dev_dst, size_t count) {
    MGX_REQUIRE(ctx && dev_src && dev_dst && count, MGX_ERR_INVALID, "NULL argument");
    MGX_REQUIRE(ctx->rccl_comm, MGX_ERR_RCCL, "RCCL communicator not initialised");
    int st = order_after_compute(ctx);
    if (st) return st;
    ncclComm_t comm = (ncclComm_t)ctx->rccl_comm;
    MGX_NCCL(ncclGroupStart());
    MGX_NCCL(ncclSend(dev_src, count, ncclFloat64, ctx->rank, comm, ctx->comm));
    MGX_NCCL(ncclRecv(dev_dst, count, ncclFloat64, ctx->rank, comm, ctx->comm));
    MGX_NCCL(ncclGroupEnd());
    MGX_HIP(hipEventRecord(ctx->ev_comm, ctx->comm));
    MGX_HIP(hipStreamWaitEvent(ctx->compute, ctx->ev_comm, 0));
    MGX_HIP(hipStreamSynchronize(ctx->comm));
    return MGX_OK;
}

int mgx_comm_allreduce_sum_f64(mgx_ctx* ctx, double* dev_inout, size_t count) {
    MGX_REQUIRE(ctx && dev_inout, MGX_ERR_INVALID, "NULL argument");
    if (ctx->nranks == 1) return MGX_OK;
    MGX_REQUIRE(!ctx->local_group, MGX_ERR_INVALID, "allreduce is not implemented by the local test transport");
    MGX_REQUIRE(ctx->rccl_comm, MGX_ERR_RCCL, "communicator not initialised");
    int st = order_after_compute(ctx);
    if (st) return st;
    MGX_NCCL(ncclAllReduce(dev_inout, dev_inout, count, ncclFloat64, ncclSum, (ncclComm_t)ctx->rccl_comm, ctx->comm));
    return MGX_OK;
}

}  // extern "C"

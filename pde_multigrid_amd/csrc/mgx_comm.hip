// mgx_comm.hip -- z-slab halo exchange and coarse-level collectives over RCCL (xGMI).
//
// New work relative to the reference (it is single-process, single-device; its thesis
// lists multi-GPU sub-grids as future work).  Storage is x fastest / z slowest, so a
// z-plane is one contiguous block and ghost planes need no packing: the halo exchange
// is a grouped ncclSend/ncclRecv pair with each of the (at most two) chain neighbours,
// each over its own point-to-point xGMI link, issued on the context's comm stream so that
// interior smoothing keeps running on the compute stream.
#include <rccl/rccl.h>

#include "mgx_internal.hpp"

#define MGX_NCCL(expr)                                                                          \
    do {                                                                                        \
        ncclResult_t r_ = (expr);                                                               \
        if (r_ != ncclSuccess)                                                                  \
            return mgx::fail(MGX_ERR_RCCL, "%s failed: %s (%s:%d)", #expr, ncclGetErrorString(r_), \
                             __FILE__, __LINE__);                                               \
    } while (0)

static_assert(sizeof(ncclUniqueId) == MGX_UNIQUE_ID_BYTES, "MGX_UNIQUE_ID_BYTES must match ncclUniqueId");

namespace {

// comm stream waits for everything enqueued so far on the compute stream
int order_after_compute(mgx_ctx* ctx) {
    MGX_HIP(hipEventRecord(ctx->ev_compute, ctx->compute));
    MGX_HIP(hipStreamWaitEvent(ctx->comm, ctx->ev_compute, 0));
    return MGX_OK;
}

ncclDataType_t dtype_of(int elem_bytes) { return elem_bytes == 4 ? ncclFloat32 : ncclFloat64; }

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
    MGX_REQUIRE(!ctx->rccl_comm, MGX_ERR_INVALID, "communicator already initialised");
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

int mgx_comm_destroy(mgx_ctx* ctx) {
    MGX_REQUIRE(ctx, MGX_ERR_INVALID, "ctx is NULL");
    if (ctx->rccl_comm) {
        (void)hipStreamSynchronize(ctx->comm);
        ncclCommDestroy((ncclComm_t)ctx->rccl_comm);
        ctx->rccl_comm = nullptr;
    }
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

int mgx_comm_halo_exchange(mgx_ctx* ctx, const void* send_down, void* recv_down, const void* send_up, void* recv_up,
                           size_t count, int elem_bytes) {
    MGX_REQUIRE(ctx, MGX_ERR_INVALID, "ctx is NULL");
    MGX_REQUIRE(elem_bytes == 4 || elem_bytes == 8, MGX_ERR_INVALID, "elem_bytes = %d", elem_bytes);
    if (ctx->nranks == 1 || count == 0) return MGX_OK;
    MGX_REQUIRE(ctx->rccl_comm, MGX_ERR_RCCL, "communicator not initialised");
    const bool has_down = ctx->rank > 0, has_up = ctx->rank < ctx->nranks - 1;
    MGX_REQUIRE(!has_down || (send_down && recv_down), MGX_ERR_INVALID, "down buffers are NULL");
    MGX_REQUIRE(!has_up || (send_up && recv_up), MGX_ERR_INVALID, "up buffers are NULL");
    int st = order_after_compute(ctx);
    if (st) return st;
    ncclComm_t comm = (ncclComm_t)ctx->rccl_comm;
    const ncclDataType_t dt = dtype_of(elem_bytes);
    MGX_NCCL(ncclGroupStart());
    if (has_down) {
        MGX_NCCL(ncclSend(send_down, count, dt, ctx->rank - 1, comm, ctx->comm));
        MGX_NCCL(ncclRecv(recv_down, count, dt, ctx->rank - 1, comm, ctx->comm));
    }
    if (has_up) {
        MGX_NCCL(ncclSend(send_up, count, dt, ctx->rank + 1, comm, ctx->comm));
        MGX_NCCL(ncclRecv(recv_up, count, dt, ctx->rank + 1, comm, ctx->comm));
    }
    MGX_NCCL(ncclGroupEnd());
    return MGX_OK;
}

int mgx_comm_wait(mgx_ctx* ctx) {
    MGX_REQUIRE(ctx, MGX_ERR_INVALID, "ctx is NULL");
    if (ctx->nranks == 1) return MGX_OK;
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
    MGX_REQUIRE(ctx->rccl_comm, MGX_ERR_RCCL, "communicator not initialised");
    int st = order_after_compute(ctx);
    if (st) return st;
    MGX_NCCL(ncclAllGather(send, recv, count, dtype_of(elem_bytes), (ncclComm_t)ctx->rccl_comm, ctx->comm));
    return MGX_OK;
}

int mgx_comm_allreduce_sum_f64(mgx_ctx* ctx, double* dev_inout, size_t count) {
    MGX_REQUIRE(ctx && dev_inout, MGX_ERR_INVALID, "NULL argument");
    if (ctx->nranks == 1) return MGX_OK;
    MGX_REQUIRE(ctx->rccl_comm, MGX_ERR_RCCL, "communicator not initialised");
    int st = order_after_compute(ctx);
    if (st) return st;
    MGX_NCCL(ncclAllReduce(dev_inout, dev_inout, count, ncclFloat64, ncclSum, (ncclComm_t)ctx->rccl_comm, ctx->comm));
    return MGX_OK;
}

}  // extern "C"

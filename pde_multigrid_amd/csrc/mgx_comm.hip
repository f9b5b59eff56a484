// mgx_comm.hip -- z-slab halo exchange and coarse-level collectives.
//
// New work relative to the reference (it is single-process, single-device; its thesis
// lists multi-GPU sub-grids as future work).  Storage is x fastest / z slowest, so a
// z-plane is one contiguous block and ghost planes need no packing: the halo exchange
// is a grouped ncclSend/ncclRecv pair with each of the (at most two) chain neighbours,
// each over its own point-to-point xGMI link, issued on the context's comm stream so that
// interior smoothing keeps running on the compute stream.
//
// Two transports behind the same entry points, with the SAME stream semantics:
//   RCCL   one process per GPU (mgx_comm_init): the production path
//   local  several host threads of ONE process, one context each, all on the same device
//          (mgx_comm_init_local).  It exists so that the slab-decomposed cycle -- including the event ordering
//          of its overlap schedule -- can be checked bit-for-bit on a single-GPU box.  Like RCCL it is
//          asynchronous: an exchange only ENQUEUES work on the comm stream (behind order_after_compute), peers
//          are ordered by cross-context events (hipStreamWaitEvent on the neighbour's "planes posted" and
//          "ghosts consumed" events), no stream is ever synchronised by the host, and the compute stream sees the
//          result only through mgx_comm_wait.  The host threads rendezvous only to learn that a neighbour's
//          event has been RECORDED (an event that was never recorded cannot be waited for); that is the job
//          RCCL's proxy does.  Test hooks (mgx_local_group_set_test_hooks): a delay kernel in front of every
//          transfer, so that a missing wait reads stale ghosts for certain, and a switch that turns
//          mgx_comm_wait into a no-op, so that the tests can prove they would notice.
#include <rccl/rccl.h>

#include <chrono>
#include <condition_variable>
#include <mutex>
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
    static constexpr int SLOTS = 3;  // event ring: exchange k uses slot k % SLOTS (see the reuse argument in local_begin)
    int nranks = 0;
    std::mutex mu;
    std::condition_variable cv;
    struct Rank {
        bool attached = false;
        hipEvent_t ready[SLOTS] = {}, done[SLOTS] = {};
        unsigned long long seq = 0;           // collectives this rank has started (only its own thread touches it)
        unsigned long long posted_ready = 0;  // ... whose "posted" event is recorded           (guarded by mu)
        unsigned long long posted_done = 0;   // ... whose "consumed" event is recorded          (guarded by mu)
        const void* to_lower[SLOTS] = {};
        const void* to_upper[SLOTS] = {};
        const void* gather[SLOTS] = {};
        double* red = nullptr;  // all-reduce staging: nranks x count doubles
        size_t red_cap = 0;
    };
    std::vector<Rank> rank;
    int delay_us = 0;    // test hook: every transfer starts this late on the receiving comm stream
    int drop_waits = 0;  // test hook: mgx_comm_wait does nothing (fault injection for the negative test)
    bool failed = false;
};

namespace {

// the stream collectives are enqueued on: the comm stream, or -- inline mode -- the compute stream itself
inline hipStream_t cstream(const mgx_ctx* ctx) { return ctx->comm_inline ? ctx->compute : ctx->comm; }

// comm stream waits for everything enqueued so far on the compute stream (inline mode: stream order does it)
int order_after_compute(mgx_ctx* ctx) {
    if (ctx->comm_inline) return MGX_OK;
    MGX_HIP(hipEventRecord(ctx->ev_compute, ctx->compute));
    MGX_HIP(hipStreamWaitEvent(ctx->comm, ctx->ev_compute, 0));
    return MGX_OK;
}

ncclDataType_t dtype_of(int elem_bytes) { return elem_bytes == 4 ? ncclFloat32 : ncclFloat64; }

__global__ void delay_kernel(unsigned us) {  // wall_clock64 ticks at 100 MHz on gfx950
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < (unsigned long long)us * 100ull) __builtin_amdgcn_s_sleep(64);
}

__global__ void __launch_bounds__(256) sum_ranks_kernel(const double* __restrict__ parts, double* __restrict__ out, size_t count,
                                                        int nranks) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    double s = parts[i];  // fixed order: rank 0, 1, 2, ... -> every rank forms the same sum
    for (int r = 1; r < nranks; r++) s += parts[(size_t)r * count + i];
    out[i] = s;
}

// ---- the local transport ---------------------------------------------------------------------------------
// Every collective of a rank has a sequence number k (all ranks issue the same collectives in the same order,
// as with RCCL).  Rank r, collective k, slot s = k % SLOTS:
//   1. comm_r waits for compute_r (order_after_compute); event ready[r][s] is recorded on comm_r ("my send
//      planes are final"), the send pointers are posted, posted_ready[r] = k;
//   2. the host thread waits until every peer p of this collective has posted_ready[p] >= k, then
//      comm_r waits for ready[p][s] and the copies FROM the peers' send buffers are enqueued on comm_r;
//   3. event done[r][s] is recorded on comm_r ("I have read my peers' planes"), posted_done[r] = k;
//   4. the host thread waits until posted_done[p] >= k for every peer and makes comm_r wait for done[p][s]:
//      like ncclSend, the exchange is complete on my comm stream only when my planes have been received.
// Event reuse: ready[r][s] is re-recorded in collective k + SLOTS; every wait on its k-th recording was issued
// in step 2 of a peer's collective k, which precedes that peer's posted_done = k, which rank r has seen in step 4
// of its own collective k.  The same argument covers the posted pointers, and done[r][s] as far as the peers of
// collective k + SLOTS are concerned.  Consecutive collectives may have different peer sets, though (all-gather:
// everybody; halo exchange: the chain neighbours): a rank p that was a peer of r in collective k but is none in the
// following halo exchanges may issue its step-4 wait on done[r][s] after r has re-recorded the event for collective
// k + SLOTS.  p then waits for the newer recording, which lies later in r's stream: it over-waits, it never sees an
// earlier state, and it cannot form a cycle (r's collective k + SLOTS does not involve p, so it does not wait for p).
// Three slots make that case rarer; they are not what makes it correct.
struct Peers {
    int p[64];
    int n = 0;
};

int wait_posted(mgx_local_group* g, const Peers& peers, unsigned long long k, bool done) {
    std::unique_lock<std::mutex> lk(g->mu);
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::seconds(120);
    for (;;) {
        if (g->failed) return mgx::fail(MGX_ERR_RCCL, "local transport: another rank failed");
        bool all = true;
        for (int i = 0; i < peers.n; i++) {
            const auto& q = g->rank[peers.p[i]];
            if ((done ? q.posted_done : q.posted_ready) < k) all = false;
        }
        if (all) return MGX_OK;
        if (g->cv.wait_until(lk, deadline) == std::cv_status::timeout) {
            g->failed = true;
            g->cv.notify_all();
            return mgx::fail(MGX_ERR_RCCL, "local transport: a peer did not reach collective %llu (unbalanced schedule?)", k);
        }
    }
}

int local_begin(mgx_ctx* ctx, const Peers& peers, const void* to_lower, const void* to_upper, const void* gather,
                unsigned long long* k_out) {
    mgx_local_group* g = (mgx_local_group*)ctx->local_group;
    auto& me = g->rank[ctx->rank];
    const unsigned long long k = ++me.seq;
    const int s = (int)(k % mgx_local_group::SLOTS);
    MGX_TRY_RET(order_after_compute(ctx));
    MGX_HIP(hipEventRecord(me.ready[s], cstream(ctx)));
    {
        std::lock_guard<std::mutex> lk(g->mu);
        me.to_lower[s] = to_lower;
        me.to_upper[s] = to_upper;
        me.gather[s] = gather;
        me.posted_ready = k;
    }
    g->cv.notify_all();
    MGX_TRY_RET(wait_posted(g, peers, k, false));
    for (int i = 0; i < peers.n; i++) MGX_HIP(hipStreamWaitEvent(cstream(ctx), g->rank[peers.p[i]].ready[s], 0));
    if (g->delay_us > 0) MGX_LAUNCH(delay_kernel, dim3(1), dim3(1), 0, cstream(ctx), (unsigned)g->delay_us);
    *k_out = k;
    return MGX_OK;
}

int local_end(mgx_ctx* ctx, const Peers& peers, unsigned long long k) {
    mgx_local_group* g = (mgx_local_group*)ctx->local_group;
    auto& me = g->rank[ctx->rank];
    const int s = (int)(k % mgx_local_group::SLOTS);
    MGX_HIP(hipEventRecord(me.done[s], cstream(ctx)));
    {
        std::lock_guard<std::mutex> lk(g->mu);
        me.posted_done = k;
    }
    g->cv.notify_all();
    MGX_TRY_RET(wait_posted(g, peers, k, true));
    for (int i = 0; i < peers.n; i++) MGX_HIP(hipStreamWaitEvent(cstream(ctx), g->rank[peers.p[i]].done[s], 0));
    return MGX_OK;
}

Peers chain_peers(const mgx_ctx* ctx) {
    Peers p;
    if (ctx->rank > 0) p.p[p.n++] = ctx->rank - 1;
    if (ctx->rank < ctx->nranks - 1) p.p[p.n++] = ctx->rank + 1;
    return p;
}

Peers all_peers(const mgx_ctx* ctx) {
    Peers p;
    for (int r = 0; r < ctx->nranks; r++)
        if (r != ctx->rank) p.p[p.n++] = r;
    return p;
}

// a rank that bails out between local_begin and local_end tells the others, so that they fail at once instead of
// waiting for the rendezvous timeout
void local_fail(mgx_local_group* g) {
    {
        std::lock_guard<std::mutex> lk(g->mu);
        g->failed = true;
    }
    g->cv.notify_all();
}

int local_halo(mgx_ctx* ctx, const void* send_lo, void* recv_lo, size_t n_from_lo, const void* send_up, void* recv_up,
               size_t n_from_up, int eb) {
    mgx_local_group* g = (mgx_local_group*)ctx->local_group;
    const Peers peers = chain_peers(ctx);
    unsigned long long k = 0;
    MGX_TRY_RET(local_begin(ctx, peers, send_lo, send_up, nullptr, &k));
    const int s = (int)(k % mgx_local_group::SLOTS);
    if (ctx->rank > 0 && n_from_lo) {
        const void* src = g->rank[ctx->rank - 1].to_upper[s];
        if (!src) local_fail(g);
        MGX_REQUIRE(src, MGX_ERR_INVALID, "halo exchange: the lower neighbour sends nothing up, but %zu elements are expected", n_from_lo);
        MGX_HIP(hipMemcpyAsync(recv_lo, src, n_from_lo * eb, hipMemcpyDeviceToDevice, cstream(ctx)));
    }
    if (ctx->rank < ctx->nranks - 1 && n_from_up) {
        const void* src = g->rank[ctx->rank + 1].to_lower[s];
        if (!src) local_fail(g);
        MGX_REQUIRE(src, MGX_ERR_INVALID, "halo exchange: the upper neighbour sends nothing down, but %zu elements are expected", n_from_up);
        MGX_HIP(hipMemcpyAsync(recv_up, src, n_from_up * eb, hipMemcpyDeviceToDevice, cstream(ctx)));
    }
    return local_end(ctx, peers, k);
}

int local_allgather(mgx_ctx* ctx, const void* send, void* recv, size_t count, int eb) {
    mgx_local_group* g = (mgx_local_group*)ctx->local_group;
    const Peers peers = all_peers(ctx);
    unsigned long long k = 0;
    MGX_TRY_RET(local_begin(ctx, peers, nullptr, nullptr, send, &k));
    const int s = (int)(k % mgx_local_group::SLOTS);
    for (int r = 0; r < ctx->nranks; r++) {
        const void* src = r == ctx->rank ? send : g->rank[r].gather[s];
        char* dst = (char*)recv + (size_t)r * count * eb;
        if (src != dst) MGX_HIP(hipMemcpyAsync(dst, src, count * eb, hipMemcpyDeviceToDevice, cstream(ctx)));
    }
    return local_end(ctx, peers, k);
}

int local_allreduce(mgx_ctx* ctx, double* inout, size_t count) {
    mgx_local_group* g = (mgx_local_group*)ctx->local_group;
    auto& me = g->rank[ctx->rank];
    const size_t need = (size_t)ctx->nranks * count;
    if (me.red_cap < need) {
        if (me.red) {
            MGX_HIP(hipStreamSynchronize(cstream(ctx)));
            MGX_HIP(hipFree(me.red));
            me.red = nullptr;
            me.red_cap = 0;
        }
        MGX_HIP(hipMalloc((void**)&me.red, need * sizeof(double)));
        me.red_cap = need;
    }
    const Peers peers = all_peers(ctx);
    unsigned long long k = 0;
    MGX_TRY_RET(local_begin(ctx, peers, nullptr, nullptr, inout, &k));
    const int s = (int)(k % mgx_local_group::SLOTS);
    for (int r = 0; r < ctx->nranks; r++) {
        const void* src = r == ctx->rank ? (const void*)inout : g->rank[r].gather[s];
        MGX_HIP(hipMemcpyAsync(me.red + (size_t)r * count, src, count * sizeof(double), hipMemcpyDeviceToDevice, cstream(ctx)));
    }
    MGX_TRY_RET(local_end(ctx, peers, k));  // nobody still reads my input: it may be overwritten in place now
    MGX_LAUNCH(sum_ranks_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, cstream(ctx), (const double*)me.red,
                       inout, count, ctx->nranks);
    MGX_LAUNCH_CHECK();
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

// REHEARSAL of one rank of a larger job on a single GPU: a one-rank RCCL communicator, but the context reports
// (virtual_rank, virtual_nranks), so that the slab plan, the overlap schedule, the stream / event pattern and the sizes of
// every message are those of that rank; every send and receive goes to THIS rank (RCCL self send / recv, as
// mgx_comm_selftest does), an all-gather receives its own share nranks times.  The planes received are wrong, so results
// are meaningless: for timing one rank's share of a multi-GPU cycle where only one GPU is available (tools/rehearse_rank.py).
int mgx_comm_init_rehearsal(mgx_ctx* ctx, const void* host_id_bytes, int virtual_rank, int virtual_nranks) {
    MGX_REQUIRE(ctx && host_id_bytes, MGX_ERR_INVALID, "NULL argument");
    MGX_REQUIRE(virtual_nranks >= 2 && virtual_rank >= 0 && virtual_rank < virtual_nranks, MGX_ERR_INVALID, "bad rank %d / %d", virtual_rank,
                virtual_nranks);
    MGX_REQUIRE(!ctx->rccl_comm && !ctx->local_group, MGX_ERR_INVALID, "communicator already initialised");
    MGX_HIP(hipSetDevice(ctx->device));
    ncclUniqueId id;
    memcpy(&id, host_id_bytes, sizeof id);
    ncclComm_t comm;
    MGX_NCCL(ncclCommInitRank(&comm, 1, id, 0));
    ctx->rccl_comm = (void*)comm;
    ctx->rank = virtual_rank;
    ctx->nranks = virtual_nranks;
    ctx->comm_rehearse = 1;
    return MGX_OK;
}

int mgx_local_group_create(int nranks, mgx_local_group** out) {
    MGX_REQUIRE(out && nranks >= 1 && nranks <= 64, MGX_ERR_INVALID, "bad arguments (1 <= nranks <= 64)");
    mgx_local_group* g = new mgx_local_group();
    g->nranks = nranks;
    g->rank.resize(nranks);
    *out = g;
    return MGX_OK;
}

int mgx_local_group_destroy(mgx_local_group* g) {
    if (!g) return MGX_OK;
    for (auto& r : g->rank) {
        for (int s = 0; s < mgx_local_group::SLOTS; s++) {
            if (r.ready[s]) (void)hipEventDestroy(r.ready[s]);
            if (r.done[s]) (void)hipEventDestroy(r.done[s]);
        }
        if (r.red) (void)hipFree(r.red);
    }
    delete g;
    return MGX_OK;
}

int mgx_local_group_set_test_hooks(mgx_local_group* g, int delay_us, int drop_waits) {
    MGX_REQUIRE(g && delay_us >= 0 && delay_us <= 100000, MGX_ERR_INVALID, "bad arguments (0 <= delay_us <= 100000)");
    std::lock_guard<std::mutex> lk(g->mu);
    g->delay_us = delay_us;
    g->drop_waits = drop_waits ? 1 : 0;
    return MGX_OK;
}

int mgx_comm_init_local(mgx_ctx* ctx, mgx_local_group* group, int rank) {
    MGX_REQUIRE(ctx && group, MGX_ERR_INVALID, "NULL argument");
    MGX_REQUIRE(rank >= 0 && rank < group->nranks, MGX_ERR_INVALID, "bad rank %d / %d", rank, group->nranks);
    MGX_REQUIRE(!ctx->rccl_comm && !ctx->local_group, MGX_ERR_INVALID, "communicator already initialised");
    MGX_REQUIRE(!group->rank[rank].attached, MGX_ERR_INVALID, "rank %d is already attached", rank);
    MGX_USE(ctx);
    auto& me = group->rank[rank];
    for (int s = 0; s < mgx_local_group::SLOTS; s++) {
        MGX_HIP(hipEventCreateWithFlags(&me.ready[s], hipEventDisableTiming));
        MGX_HIP(hipEventCreateWithFlags(&me.done[s], hipEventDisableTiming));
    }
    me.attached = true;
    ctx->local_group = group;
    ctx->rank = rank;
    ctx->nranks = group->nranks;
    return MGX_OK;
}

int mgx_comm_destroy(mgx_ctx* ctx) {
    MGX_REQUIRE(ctx, MGX_ERR_INVALID, "ctx is NULL");
    MGX_USE(ctx);
    // inline mode enqueues collectives on the compute stream: both streams may hold operations of the communicator
    (void)hipStreamSynchronize(ctx->compute);
    (void)hipStreamSynchronize(ctx->comm);
    if (ctx->rccl_comm) {
        ncclCommDestroy((ncclComm_t)ctx->rccl_comm);
        ctx->rccl_comm = nullptr;
    }
    ctx->local_group = nullptr;
    ctx->comm_inline = 0;  // a communicator initialised later starts in overlapped mode
    ctx->comm_rehearse = 0;
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

// what the communicator itself reports: the number of ranks RCCL sees (ncclCommCount; the local test transport: its
// group size; no communicator: 1) and the RCCL version the library is running on (ncclGetVersion, e.g. 22105)
int mgx_comm_capturable(const mgx_ctx* ctx) { return ctx && !ctx->local_group && (ctx->rccl_comm || ctx->nranks == 1); }

int mgx_comm_info(const mgx_ctx* ctx, int* ranks_seen, int* rccl_version) {
    MGX_REQUIRE(ctx && ranks_seen && rccl_version, MGX_ERR_INVALID, "NULL argument");
    *ranks_seen = ctx->nranks;
    *rccl_version = 0;
    int v = 0;
    if (ncclGetVersion(&v) == ncclSuccess) *rccl_version = v;
    if (ctx->rccl_comm) {
        int n = 0;
        MGX_NCCL(ncclCommCount((ncclComm_t)ctx->rccl_comm, &n));
        *ranks_seen = n;
    }
    return MGX_OK;
}

int mgx_comm_halo_exchange(mgx_ctx* ctx, const void* send_to_lower, size_t count_to_lower, void* recv_from_lower,
                           size_t count_from_lower, const void* send_to_upper, size_t count_to_upper, void* recv_from_upper,
                           size_t count_from_upper, int elem_bytes) {
    MGX_REQUIRE(ctx, MGX_ERR_INVALID, "ctx is NULL");
    MGX_USE(ctx);
    MGX_REQUIRE(elem_bytes == 4 || elem_bytes == 8, MGX_ERR_INVALID, "elem_bytes = %d", elem_bytes);
    if (ctx->nranks == 1) return MGX_OK;
    const bool has_lo = ctx->rank > 0, has_up = ctx->rank < ctx->nranks - 1;
    MGX_REQUIRE(!has_lo || ((send_to_lower || !count_to_lower) && (recv_from_lower || !count_from_lower)), MGX_ERR_INVALID,
                "lower-neighbour buffers are NULL");
    MGX_REQUIRE(!has_up || ((send_to_upper || !count_to_upper) && (recv_from_upper || !count_from_upper)), MGX_ERR_INVALID,
                "upper-neighbour buffers are NULL");
    if (ctx->local_group)
        return local_halo(ctx, count_to_lower ? send_to_lower : nullptr, recv_from_lower, recv_from_lower ? count_from_lower : 0,
                          count_to_upper ? send_to_upper : nullptr, recv_from_upper, recv_from_upper ? count_from_upper : 0,
                          elem_bytes);
    MGX_REQUIRE(ctx->rccl_comm, MGX_ERR_RCCL, "communicator not initialised");
    int st = order_after_compute(ctx);
    if (st) return st;
    ncclComm_t comm = (ncclComm_t)ctx->rccl_comm;
    const ncclDataType_t dt = dtype_of(elem_bytes);
    if (ctx->comm_rehearse) {
        // both neighbours are this rank itself: sends and receives pair up in the order they are posted, and a message whose
        // partner lives on another rank in the real job (the bottom / top rank of the chain) gets a scratch buffer as partner
        const void* sp[2];
        void* rp[2];
        size_t sc[2], rc[2];
        int ns = 0, nr = 0;
        if (has_lo && count_to_lower) { sp[ns] = send_to_lower; sc[ns++] = count_to_lower; }
        if (has_up && count_to_upper) { sp[ns] = send_to_upper; sc[ns++] = count_to_upper; }
        if (has_lo && count_from_lower) { rp[nr] = recv_from_lower; rc[nr++] = count_from_lower; }
        if (has_up && count_from_upper) { rp[nr] = recv_from_upper; rc[nr++] = count_from_upper; }
        const int np = ns > nr ? ns : nr;
        size_t need = 0;
        for (int i = 0; i < np; i++) need += (i < ns ? sc[i] : rc[i]) * (size_t)elem_bytes;
        if (ns != nr && ctx->rehearse_bytes < need) {
            MGX_HIP(hipStreamSynchronize(ctx->compute));
            MGX_HIP(hipStreamSynchronize(ctx->comm));
            if (ctx->rehearse_buf) MGX_HIP(hipFree(ctx->rehearse_buf));
            ctx->rehearse_buf = nullptr;
            MGX_HIP(hipMalloc(&ctx->rehearse_buf, need));
            MGX_HIP(hipMemset(ctx->rehearse_buf, 0, need));
            ctx->rehearse_bytes = need;
        }
        char* scratch = (char*)ctx->rehearse_buf;
        MGX_NCCL(ncclGroupStart());
        for (int i = 0; i < np; i++) {
            const size_t cnt = i < ns ? sc[i] : rc[i];
            const void* src = i < ns ? sp[i] : (const void*)scratch;
            void* dst = i < nr ? rp[i] : (void*)scratch;
            MGX_REQUIRE(i >= ns || i >= nr || sc[i] == rc[i], MGX_ERR_INVALID, "rehearsal: a send of %zu meets a receive of %zu elements", sc[i], rc[i]);
            MGX_NCCL(ncclSend(src, cnt, dt, 0, comm, cstream(ctx)));
            MGX_NCCL(ncclRecv(dst, cnt, dt, 0, comm, cstream(ctx)));
            if (i >= ns || i >= nr) scratch += cnt * (size_t)elem_bytes;
        }
        MGX_NCCL(ncclGroupEnd());
        return MGX_OK;
    }
    const int lo = ctx->rank - 1, up = ctx->rank + 1;
    MGX_NCCL(ncclGroupStart());
    if (has_lo) {
        if (count_to_lower) MGX_NCCL(ncclSend(send_to_lower, count_to_lower, dt, lo, comm, cstream(ctx)));
        if (count_from_lower) MGX_NCCL(ncclRecv(recv_from_lower, count_from_lower, dt, lo, comm, cstream(ctx)));
    }
    if (has_up) {
        if (count_to_upper) MGX_NCCL(ncclSend(send_to_upper, count_to_upper, dt, up, comm, cstream(ctx)));
        if (count_from_upper) MGX_NCCL(ncclRecv(recv_from_upper, count_from_upper, dt, up, comm, cstream(ctx)));
    }
    MGX_NCCL(ncclGroupEnd());
    return MGX_OK;
}

// the compute stream waits for everything enqueued so far on the comm stream (both transports)
int mgx_comm_wait(mgx_ctx* ctx) {
    MGX_REQUIRE(ctx, MGX_ERR_INVALID, "ctx is NULL");
    MGX_USE(ctx);
    if (ctx->nranks == 1) return MGX_OK;
    if (ctx->local_group && ((mgx_local_group*)ctx->local_group)->drop_waits) return MGX_OK;  // fault injection (tests)
    if (ctx->comm_inline) return MGX_OK;  // the collectives are on the compute stream: ordered by the stream
    MGX_HIP(hipEventRecord(ctx->ev_comm, ctx->comm));
    MGX_HIP(hipStreamWaitEvent(ctx->compute, ctx->ev_comm, 0));
    return MGX_OK;
}

}  // extern "C" (a template follows)

// ---- half planes: a colour pass changes ONE colour of a plane, i.e. one half-row of every x-split row (the even-x half where
// (colour + y + z) is even, the odd-x half elsewhere), so a ghost exchange behind a pass needs to carry only those: the changed
// half-rows of a plane are packed into a staging array (row pitch = the longer half), sent, and unpacked into the ghost plane
// on the stream the receive is enqueued on (so the unpacking overlaps the interior launch like the transfer itself).
template <class real>
__global__ void __launch_bounds__(256) halo_halfrows_kernel(const real* __restrict__ srcA, real* __restrict__ dstA, int zA,
                                                            const real* __restrict__ srcB, real* __restrict__ dstB, int zB, int sx, int sy,
                                                            int colour, int pack) {
    // blockIdx.y: plane A / B; blockIdx.x: row.  pack: plane -> staging; else staging -> plane
    const bool second = blockIdx.y != 0;
    const real* src = second ? srcB : srcA;
    real* dst = second ? dstB : dstA;
    if (!src || !dst) return;
    const int z = second ? zB : zA, y = blockIdx.x;
    const int al = 128 / (int)sizeof(real);
    const int H = (((sx + 1) >> 1) + al - 1) / al * al, H2 = ((sx >> 1) + al - 1) / al * al, P = H + H2;
    const int q = (colour + y + z) & 1;          // the half of row y that holds the colour
    const int n = q ? H2 : H;
    const size_t prow = (size_t)y * P + (q ? H : 0), srow = (size_t)y * H;
    for (int i = threadIdx.x; i < n; i += 256) {
        if (pack) dst[srow + i] = src[prow + i];
        else dst[prow + i] = src[srow + i];
    }
}

#define MGX_HALFROWS(SFX, real)                                                                                              \
    size_t mgx3dxs_halfplane_elems_##SFX(int sx, int sy) {                                                                   \
        const int al = 128 / (int)sizeof(real);                                                                              \
        return (size_t)sy * (size_t)((((sx + 1) >> 1) + al - 1) / al * al);                                                  \
    }                                                                                                                        \
    int mgx3dxs_halo_pack_##SFX(mgx_ctx* ctx, const real* plane_a, int z_a, real* stage_a, const real* plane_b, int z_b,    \
                                real* stage_b, int sx, int sy, int colour) {                                                 \
        MGX_REQUIRE(ctx && sx >= 3 && sy >= 3 && (colour == 0 || colour == 1), MGX_ERR_INVALID, "halo_pack: bad argument");  \
        MGX_USE(ctx);                                                                                                        \
        if (ctx->nranks == 1 || ((!plane_a || !stage_a) && (!plane_b || !stage_b))) return MGX_OK;                           \
        MGX_LAUNCH((halo_halfrows_kernel<real>), dim3(sy, 2), dim3(256), 0, ctx->compute, plane_a, stage_a, z_a,     \
                           plane_b, stage_b, z_b, sx, sy, colour, 1);                                                        \
        MGX_LAUNCH_CHECK();                                                                                                  \
        return MGX_OK;                                                                                                       \
    }                                                                                                                        \
    int mgx3dxs_halo_unpack_##SFX(mgx_ctx* ctx, const real* stage_a, real* plane_a, int z_a, const real* stage_b,           \
                                  real* plane_b, int z_b, int sx, int sy, int colour) {                                      \
        MGX_REQUIRE(ctx && sx >= 3 && sy >= 3 && (colour == 0 || colour == 1), MGX_ERR_INVALID, "halo_unpack: bad argument"); \
        MGX_USE(ctx);                                                                                                        \
        if (ctx->nranks == 1 || ((!plane_a || !stage_a) && (!plane_b || !stage_b))) return MGX_OK;                           \
        /* behind the receive, on the stream it was enqueued on: mgx_comm_wait covers it */                                  \
        MGX_LAUNCH((halo_halfrows_kernel<real>), dim3(sy, 2), dim3(256), 0, cstream(ctx), stage_a, plane_a, z_a,     \
                           stage_b, plane_b, z_b, sx, sy, colour, 0);                                                        \
        MGX_LAUNCH_CHECK();                                                                                                  \
        return MGX_OK;                                                                                                       \
    }
extern "C" {
MGX_HALFROWS(f32, float)
MGX_HALFROWS(f64, double)
}
#undef MGX_HALFROWS

extern "C" {
// Inline mode: collectives are enqueued on the compute stream, in order with the kernels, instead of on the comm stream
// behind an event -- no overlap with computation, but also none of the two cross-stream dependencies an overlapped
// exchange costs (measured 12.5 us each on MI355X / ROCm 7.2 against 2.5 us between kernels of one stream,
// tools/probes/hop_latency.hip): the right mode for levels whose interior pass is shorter than that.  Switching it on first
// orders the compute stream behind whatever the comm stream still holds (one communicator: its operations stay in one
// order on every rank); every rank must switch at the same points of its schedule.
int mgx_comm_set_inline(mgx_ctx* ctx, int on) {
    MGX_REQUIRE(ctx, MGX_ERR_INVALID, "ctx is NULL");
    MGX_USE(ctx);
    on = on != 0;
    if (ctx->nranks > 1 && on && !ctx->comm_inline) {
        MGX_HIP(hipEventRecord(ctx->ev_comm, ctx->comm));
        MGX_HIP(hipStreamWaitEvent(ctx->compute, ctx->ev_comm, 0));
    }
    ctx->comm_inline = on;
    return MGX_OK;
}

int mgx_comm_allgather(mgx_ctx* ctx, const void* send, void* recv, size_t count, int elem_bytes) {
    MGX_REQUIRE(ctx && send && recv, MGX_ERR_INVALID, "NULL argument");
    MGX_USE(ctx);
    MGX_REQUIRE(elem_bytes == 4 || elem_bytes == 8, MGX_ERR_INVALID, "elem_bytes = %d", elem_bytes);
    if (ctx->nranks == 1) {
        if (send != recv) MGX_HIP(hipMemcpyAsync(recv, send, count * elem_bytes, hipMemcpyDeviceToDevice, ctx->compute));
        return MGX_OK;
    }
    if (ctx->local_group) return local_allgather(ctx, send, recv, count, elem_bytes);
    MGX_REQUIRE(ctx->rccl_comm, MGX_ERR_RCCL, "communicator not initialised");
    int st = order_after_compute(ctx);
    if (st) return st;
    if (ctx->comm_rehearse) {  // every share arrives from this rank itself
        ncclComm_t comm = (ncclComm_t)ctx->rccl_comm;
        MGX_NCCL(ncclGroupStart());
        for (int r = 0; r < ctx->nranks; r++) {
            MGX_NCCL(ncclSend(send, count, dtype_of(elem_bytes), 0, comm, cstream(ctx)));
            MGX_NCCL(ncclRecv((char*)recv + (size_t)r * count * elem_bytes, count, dtype_of(elem_bytes), 0, comm, cstream(ctx)));
        }
        MGX_NCCL(ncclGroupEnd());
        return MGX_OK;
    }
    MGX_NCCL(ncclAllGather(send, recv, count, dtype_of(elem_bytes), (ncclComm_t)ctx->rccl_comm, cstream(ctx)));
    return MGX_OK;
}

// Grouped ncclSend/ncclRecv of `count` doubles from this rank to itself on the comm stream (inline mode: on the compute
// stream): exercises RCCL
// linkage, communicator and stream/event plumbing on a box with a single GPU (nranks may be 1).  dev_src and
// dev_dst must not overlap.
int mgx_comm_selftest(mgx_ctx* ctx, const double* dev_src, double* dev_dst, size_t count) {
    MGX_REQUIRE(ctx && dev_src && dev_dst && count, MGX_ERR_INVALID, "NULL argument");
    MGX_USE(ctx);
    MGX_REQUIRE(ctx->rccl_comm, MGX_ERR_RCCL, "RCCL communicator not initialised");
    int st = order_after_compute(ctx);
    if (st) return st;
    ncclComm_t comm = (ncclComm_t)ctx->rccl_comm;
    MGX_NCCL(ncclGroupStart());
    MGX_NCCL(ncclSend(dev_src, count, ncclFloat64, ctx->rank, comm, cstream(ctx)));
    MGX_NCCL(ncclRecv(dev_dst, count, ncclFloat64, ctx->rank, comm, cstream(ctx)));
    MGX_NCCL(ncclGroupEnd());
    if (!ctx->comm_inline) {
        MGX_HIP(hipEventRecord(ctx->ev_comm, ctx->comm));
        MGX_HIP(hipStreamWaitEvent(ctx->compute, ctx->ev_comm, 0));
    }
    MGX_HIP(hipStreamSynchronize(cstream(ctx)));
    return MGX_OK;
}

// sum over the ranks, in place, of `count` doubles; every rank receives the same bits
int mgx_comm_allreduce_sum_f64(mgx_ctx* ctx, double* dev_inout, size_t count) {
    MGX_REQUIRE(ctx && dev_inout, MGX_ERR_INVALID, "NULL argument");
    MGX_USE(ctx);
    if (ctx->nranks == 1 || count == 0) return MGX_OK;
    if (ctx->local_group) return local_allreduce(ctx, dev_inout, count);
    MGX_REQUIRE(ctx->rccl_comm, MGX_ERR_RCCL, "communicator not initialised");
    int st = order_after_compute(ctx);
    if (st) return st;
    MGX_NCCL(ncclAllReduce(dev_inout, dev_inout, count, ncclFloat64, ncclSum, (ncclComm_t)ctx->rccl_comm, cstream(ctx)));
    return MGX_OK;
}

}  // extern "C"

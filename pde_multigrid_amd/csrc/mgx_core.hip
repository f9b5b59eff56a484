// mgx_core.hip -- context, memory, events, errors behind include/mgx.h.
#include "mgx_internal.hpp"

namespace mgx {

static thread_local char g_err[1024] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

// zero fill at streaming-store speed: hipMemsetAsync reaches about 2.8 TB/s on large arrays, 16-byte non-temporal
// stores about twice that (the coarse v of every level is zeroed once per cycle)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) fill_zero_kernel(u32x4* __restrict__ p, size_t n16) {
    const u32x4 z = {0u, 0u, 0u, 0u};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) __builtin_nontemporal_store(z, &p[i]);
}

int fill_zero(mgx_ctx* ctx, void* dst, size_t bytes) {
    if (!bytes) return MGX_OK;
    if (bytes < ((size_t)4 << 20) || ((uintptr_t)dst & 15) || (bytes & 15)) {
        MGX_HIP(hipMemsetAsync(dst, 0, bytes, ctx->compute));
        return MGX_OK;
    }
    const size_t n16 = bytes >> 4;
    size_t blocks = (n16 + 256 * 8 - 1) / (256 * 8);  // 8 stores per thread
    const size_t cap = (size_t)ctx->num_cus * 32;
    if (blocks > cap) blocks = cap;
    MGX_LAUNCH(fill_zero_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->compute, (u32x4*)dst, n16);
    MGX_HIP(hipGetLastError());
    return MGX_OK;
}

// ---- test hook: LDS poisoning (mgx_test_set_lds_poison) ----
// One workgroup claims all of a CU's LDS (160 KB), so no two of them share a CU; each stays ~10 us, far longer than the
// dispatcher needs to hand out the grid, so a grid of 2 x CUs workgroups on an idle GPU visits every CU.  The pattern is a
// NaN as fp64 and as two fp32 words.
int g_poison_lds = 0;
static constexpr int POISON_LDS_BYTES = 160 * 1024;
__global__ void __launch_bounds__(1024) poison_lds_kernel(unsigned long long pattern) {
    extern __shared__ unsigned long long poison_lds_[];
    volatile unsigned long long* p = poison_lds_;
    for (int i = threadIdx.x; i < POISON_LDS_BYTES / 8; i += 1024) p[i] = pattern;
    __syncthreads();
    const long long t0 = wall_clock64();  // 100 MHz
    for (int k = 0; k < 4096 && wall_clock64() - t0 < 1000; k++) __builtin_amdgcn_s_sleep(16);
}

void poison_lds_launch(hipStream_t stream) {
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        cus = 256;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
        (void)hipFuncSetAttribute((const void*)poison_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, POISON_LDS_BYTES);
    }
    hipLaunchKernelGGL(poison_lds_kernel, dim3(2 * cus), dim3(1024), POISON_LDS_BYTES, stream, 0x7FF4DEAD7FA0DEADull);
}

// counts, over every workgroup of the grid, the LDS words that carry `pattern` WITHOUT having written any: what the previous
// launch left behind (mgx_test_lds_probe: the self-test of the poisoning)
__global__ void __launch_bounds__(1024) probe_lds_kernel(unsigned long long pattern, unsigned long long* hits) {
    extern __shared__ unsigned long long poison_lds_[];
    volatile unsigned long long* p = poison_lds_;
    unsigned n = 0;
    for (int i = threadIdx.x; i < POISON_LDS_BYTES / 8; i += 1024) n += p[i] == pattern;
    for (int o = 32; o > 0; o >>= 1) n += __shfl_down(n, o);
    if ((threadIdx.x & 63) == 0 && n) atomicAdd(hits, (unsigned long long)n);
    const long long t0 = wall_clock64();
    for (int k = 0; k < 4096 && wall_clock64() - t0 < 1000; k++) __builtin_amdgcn_s_sleep(16);
}

int workspace(mgx_ctx* ctx, size_t bytes, void** out) {
    if (ctx->scratch_bytes < bytes) {
        if (ctx->scratch) {
            MGX_HIP(hipStreamSynchronize(ctx->compute));
            MGX_HIP(hipFree(ctx->scratch));
            ctx->scratch = nullptr;
            ctx->scratch_bytes = 0;
        }
        size_t want = bytes < (1u << 20) ? (1u << 20) : bytes;
        MGX_HIP(hipMalloc(&ctx->scratch, want));
        ctx->scratch_bytes = want;
    }
    *out = ctx->scratch;
    return MGX_OK;
}

}  // namespace mgx

extern "C" {

const char* mgx_status_string(int s) {
    switch (s) {
        case MGX_OK: return "MGX_OK";
        case MGX_ERR_INVALID: return "MGX_ERR_INVALID";
        case MGX_ERR_SIZE: return "MGX_ERR_SIZE";
        case MGX_ERR_HIP: return "MGX_ERR_HIP";
        case MGX_ERR_NOMEM: return "MGX_ERR_NOMEM";
        case MGX_ERR_RCCL: return "MGX_ERR_RCCL";
        case MGX_ERR_NOGPU: return "MGX_ERR_NOGPU";
    }
    return "MGX_ERR_UNKNOWN";
}

const char* mgx_last_error(void) { return mgx::g_err; }
void mgx_set_last_error(const char* msg) { mgx::set_error("%s", msg ? msg : ""); }
const char* mgx_version(void) { return "mgx 0.1 (gfx950)"; }

int mgx_device_count(int* count) {
    MGX_REQUIRE(count, MGX_ERR_INVALID, "count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        (void)hipGetLastError();
        return mgx::fail(MGX_ERR_NOGPU, "hipGetDeviceCount: %s", hipGetErrorString(e));
    }
    *count = n;
    return MGX_OK;
}

int mgx_ctx_create(int device, mgx_ctx** out) {
    MGX_REQUIRE(out, MGX_ERR_INVALID, "out is NULL");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        return mgx::fail(MGX_ERR_NOGPU, "no HIP device visible: the HIP path cannot run (there is no CPU fallback)");
    }
    MGX_REQUIRE(device >= 0 && device < n, MGX_ERR_INVALID, "device %d out of range [0,%d)", device, n);
    MGX_HIP(hipSetDevice(device));
    mgx_ctx* c = new mgx_ctx();
    c->device = device;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) c->num_cus = prop.multiProcessorCount;
    hipError_t e;
    if ((e = hipStreamCreateWithFlags(&c->compute, hipStreamNonBlocking)) != hipSuccess ||
        (e = hipStreamCreateWithFlags(&c->comm, hipStreamNonBlocking)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&c->ev_compute, hipEventDisableTiming)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&c->ev_comm, hipEventDisableTiming)) != hipSuccess) {
        delete c;
        return mgx::fail(MGX_ERR_HIP, "stream/event creation failed: %s", hipGetErrorString(e));
    }
    *out = c;
    return MGX_OK;
}

int mgx_ctx_destroy(mgx_ctx* ctx) {
    if (!ctx) return MGX_OK;
    (void)hipSetDevice(ctx->device);
    if (ctx->rccl_comm) mgx_comm_destroy(ctx);
    (void)hipStreamSynchronize(ctx->compute);
    (void)hipStreamSynchronize(ctx->comm);
    if (ctx->scratch) (void)hipFree(ctx->scratch);
    if (ctx->rehearse_buf) (void)hipFree(ctx->rehearse_buf);
    if (ctx->sweep_dev) (void)hipFree(ctx->sweep_dev);
    if (ctx->resident_buf) (void)hipFree(ctx->resident_buf);
    if (ctx->sweep_abort) (void)hipHostFree(ctx->sweep_abort);
    (void)hipEventDestroy(ctx->ev_compute);
    (void)hipEventDestroy(ctx->ev_comm);
    (void)hipStreamDestroy(ctx->compute);
    (void)hipStreamDestroy(ctx->comm);
    delete ctx;
    return MGX_OK;
}

int mgx_ctx_sync(mgx_ctx* ctx) {
    MGX_REQUIRE(ctx, MGX_ERR_INVALID, "ctx is NULL");
    MGX_USE(ctx);
    MGX_HIP(hipStreamSynchronize(ctx->compute));
    MGX_HIP(hipStreamSynchronize(ctx->comm));
    return mgx_ctx_check(ctx);
}

int mgx_ctx_device(const mgx_ctx* ctx, int* device) {
    MGX_REQUIRE(ctx && device, MGX_ERR_INVALID, "NULL argument");
    *device = ctx->device;
    return MGX_OK;
}

int mgx_ctx_stream(const mgx_ctx* ctx, void** hip_stream) {
    MGX_REQUIRE(ctx && hip_stream, MGX_ERR_INVALID, "NULL argument");
    *hip_stream = (void*)ctx->compute;
    return MGX_OK;
}

int mgx_malloc(mgx_ctx* ctx, size_t bytes, void** dptr) {
    MGX_REQUIRE(ctx && dptr, MGX_ERR_INVALID, "NULL argument");
    MGX_USE(ctx);
    *dptr = nullptr;
    if (bytes == 0) return MGX_OK;
    hipError_t e = hipMalloc(dptr, bytes);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return mgx::fail(MGX_ERR_NOMEM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    }
    return MGX_OK;
}

int mgx_free(mgx_ctx* ctx, void* dptr) {
    MGX_REQUIRE(ctx, MGX_ERR_INVALID, "ctx is NULL");
    MGX_USE(ctx);
    if (!dptr) return MGX_OK;
    MGX_HIP(hipStreamSynchronize(ctx->compute));
    MGX_HIP(hipStreamSynchronize(ctx->comm));
    MGX_HIP(hipFree(dptr));
    return MGX_OK;
}

int mgx_memcpy_h2d(mgx_ctx* ctx, void* dst, const void* host_src, size_t bytes) {
    MGX_REQUIRE(ctx && (bytes == 0 || (dst && host_src)), MGX_ERR_INVALID, "NULL argument");
    MGX_USE(ctx);
    if (!bytes) return MGX_OK;
    MGX_HIP(hipMemcpyAsync(dst, host_src, bytes, hipMemcpyHostToDevice, ctx->compute));
    MGX_HIP(hipStreamSynchronize(ctx->compute));
    return MGX_OK;
}

int mgx_memcpy_d2h(mgx_ctx* ctx, void* host_dst, const void* src, size_t bytes) {
    MGX_REQUIRE(ctx && (bytes == 0 || (host_dst && src)), MGX_ERR_INVALID, "NULL argument");
    MGX_USE(ctx);
    if (!bytes) return MGX_OK;
    MGX_HIP(hipMemcpyAsync(host_dst, src, bytes, hipMemcpyDeviceToHost, ctx->compute));
    MGX_HIP(hipStreamSynchronize(ctx->compute));
    return MGX_OK;
}

int mgx_memcpy_d2d(mgx_ctx* ctx, void* dst, const void* src, size_t bytes) {
    MGX_REQUIRE(ctx && (bytes == 0 || (dst && src)), MGX_ERR_INVALID, "NULL argument");
    MGX_USE(ctx);
    if (!bytes) return MGX_OK;
    MGX_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, ctx->compute));
    return MGX_OK;
}

int mgx_memset_zero(mgx_ctx* ctx, void* dst, size_t bytes) {
    MGX_REQUIRE(ctx && (bytes == 0 || dst), MGX_ERR_INVALID, "NULL argument");
    MGX_USE(ctx);
    return mgx::fill_zero(ctx, dst, bytes);
}

int mgx_test_set_lds_poison(int on) {
    mgx::g_poison_lds = on != 0;
    return MGX_OK;
}

int mgx_test_lds_probe(mgx_ctx* ctx, double* fraction) {
    MGX_REQUIRE(ctx && fraction, MGX_ERR_INVALID, "NULL argument");
    MGX_USE(ctx);
    void* ws = nullptr;
    MGX_TRY_RET(mgx::workspace(ctx, 8, &ws));
    MGX_HIP(hipMemsetAsync(ws, 0, 8, ctx->compute));
    MGX_HIP(hipFuncSetAttribute((const void*)mgx::probe_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, mgx::POISON_LDS_BYTES));
    const int blocks = ctx->num_cus;
    MGX_LAUNCH(mgx::probe_lds_kernel, dim3(blocks), dim3(1024), mgx::POISON_LDS_BYTES, ctx->compute, 0x7FF4DEAD7FA0DEADull,
               (unsigned long long*)ws);
    unsigned long long hits = 0;
    MGX_HIP(hipMemcpyAsync(&hits, ws, 8, hipMemcpyDeviceToHost, ctx->compute));
    MGX_HIP(hipStreamSynchronize(ctx->compute));
    *fraction = (double)hits / ((double)blocks * (mgx::POISON_LDS_BYTES / 8));
    return MGX_OK;
}

int mgx_graph_begin(mgx_ctx* ctx) {
    MGX_REQUIRE(ctx, MGX_ERR_INVALID, "ctx is NULL");
    MGX_USE(ctx);
    MGX_HIP(hipStreamBeginCapture(ctx->compute, hipStreamCaptureModeThreadLocal));
    return MGX_OK;
}

int mgx_graph_end(mgx_ctx* ctx, void** graph_exec) {
    MGX_REQUIRE(ctx && graph_exec, MGX_ERR_INVALID, "NULL argument");
    MGX_USE(ctx);
    *graph_exec = nullptr;
    hipGraph_t g = nullptr;
    MGX_HIP(hipStreamEndCapture(ctx->compute, &g));
    hipGraphExec_t e = nullptr;
    hipError_t r = hipGraphInstantiate(&e, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (r != hipSuccess) return mgx::fail(MGX_ERR_HIP, "hipGraphInstantiate: %s", hipGetErrorString(r));
    *graph_exec = (void*)e;
    return MGX_OK;
}

int mgx_graph_launch(mgx_ctx* ctx, void* graph_exec) {
    MGX_REQUIRE(ctx && graph_exec, MGX_ERR_INVALID, "NULL argument");
    MGX_USE(ctx);
    MGX_HIP(hipGraphLaunch((hipGraphExec_t)graph_exec, ctx->compute));
    return MGX_OK;
}

int mgx_graph_destroy(mgx_ctx* ctx, void* graph_exec) {
    MGX_REQUIRE(ctx, MGX_ERR_INVALID, "ctx is NULL");
    MGX_USE(ctx);
    if (graph_exec) MGX_HIP(hipGraphExecDestroy((hipGraphExec_t)graph_exec));
    return MGX_OK;
}

int mgx_event_create(mgx_ctx* ctx, mgx_event** out) {
    MGX_REQUIRE(ctx && out, MGX_ERR_INVALID, "NULL argument");
    MGX_USE(ctx);
    mgx_event* e = new mgx_event();
    hipError_t r = hipEventCreate(&e->ev);
    if (r != hipSuccess) {
        delete e;
        return mgx::fail(MGX_ERR_HIP, "hipEventCreate: %s", hipGetErrorString(r));
    }
    *out = e;
    return MGX_OK;
}

int mgx_event_destroy(mgx_ctx* ctx, mgx_event* ev) {
    (void)ctx;
    if (!ev) return MGX_OK;
    (void)hipEventDestroy(ev->ev);
    delete ev;
    return MGX_OK;
}

int mgx_event_record(mgx_ctx* ctx, mgx_event* ev) {
    MGX_REQUIRE(ctx && ev, MGX_ERR_INVALID, "NULL argument");
    MGX_USE(ctx);
    MGX_HIP(hipEventRecord(ev->ev, ctx->compute));
    return MGX_OK;
}

int mgx_event_elapsed_ms(mgx_ctx* ctx, mgx_event* start, mgx_event* stop, float* ms) {
    MGX_REQUIRE(ctx && start && stop && ms, MGX_ERR_INVALID, "NULL argument");
    MGX_USE(ctx);
    MGX_HIP(hipEventSynchronize(stop->ev));
    MGX_HIP(hipEventElapsedTime(ms, start->ev, stop->ev));
    return MGX_OK;
}

}  // extern "C"

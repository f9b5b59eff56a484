// mgx_internal.hpp -- shared by the HIP translation units behind include/mgx.h.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>

#include "mgx.h"

struct mgx_ctx {
    int device = 0;
    hipStream_t compute = nullptr;  // every kernel
    hipStream_t comm = nullptr;     // RCCL halo exchange / collectives
    hipEvent_t ev_compute = nullptr;  // compute -> comm ordering
    hipEvent_t ev_comm = nullptr;     // comm -> compute ordering
    int comm_inline = 0;              // collectives are enqueued on the compute stream itself (mgx_comm_set_inline)
    int relax_ty = 4;      // waves (row groups) per block of relax3d_xs_kernel (tuning)
    int relax_zchunk = 0;  // planes per z-chunk, 0 = automatic
    int relax_xcd = 1;     // XCD-aware block -> tile mapping
    int rr_rows = 0;       // fine rows per wave of the pipelined residual+restrict kernel (0: by level size; 2: sixteen waves)
    int rr_rcp = 1;        // residual: multiply by exact reciprocals when the squared spacings are powers of two
    int rr_xcd = 1;        // the same for residual+restrict: 1 = the pipelined kernel only (measured: -2 % at 513^3, +14 % with
                           // the streaming kernel at 257^3), 2 = both kernels, 0 = plain order
    int relax_rows = 4;    // consecutive rows per lane (register blocking in y) of relax3d_xs_kernel
    int relax_wave_planes = 0;  // time-skewed slab height of relax3d_xsplit: 0 off (measured slower: the L2-miss path, not HBM, is the limit), <0 automatic
    int relax_small = 1;   // levels <= 17^3: all sweeps of a Relax call in one workgroup (LDS resident)
    int relax_ablate = 0;  // diagnostic kernel variants (tools only)
    int rr_stream = 3;     // x-split residual+restrict: 0 LDS window kernel, 1 streaming shuffle kernel, 2 pipelined kernel,
                           // 3 = pipelined on large levels, streaming otherwise
    int relax_lds = -1;  // smoother kernel choice: -1 automatic, 0 relax3d_xs_kernel, shape codes see relax3d_xs_pass_lds
    int rr_cr = 0, rr_tyw = 4;  // streaming residual+restrict: coarse rows per lane, waves per block
    int rr_pzchunk = 0;    // coarse planes per block of residual_restrict3d_kernel, 0 = automatic
    int rr_black = 1;      // the last black pass of the pre-smoothing inside the residual+restrict launch (mgx_relax_rr3d.hip):
                           // 0 off, 1 on the HBM-bound levels, 2 wherever the geometry allows (tests)
    int rr_black_abl = 0;  // diagnostic builds: ablation bits of that kernel (timing only, WRONG results)
    int rr_black_waves = 0;  // its waves per workgroup: 0 = by precision, 12, 16
    int relax_zero_first = 1;  // relax_from_zero: the first red pass on a zeroed level does not read v (and nothing is filled)
    int relax_zero_sweep = 1;  // relax_from_zero on the pipelined levels: the first red AND black pass in one launch (f in, both colours out)
    int corr_v2 = 1;       // fp32 on wide levels: the correcting red pass with two pairs per lane too (relax3d_xs_pipe_v2_kernel, VAR = 2)
    int relax_v2 = 1;      // fp32 smoother on wide levels: two x-pairs per lane (8-byte loads)
    int corr_fuse = 1;     // interpolate_correct_relax3d: the first red pass applies the coarse-grid correction on the fly
    int cyc2_tile = 0;     // tile edge of the cache-resident 2D cycle kernels: 0 = by level size, 16 / 32 / 64
    int cyc2_tail_points = 33 * 33;  // largest top level (points) handed to the one-workgroup tail kernel of the 2D cycle (measured:
                                     // 65^2 is faster on the tiled kernels, 16 workgroups, than inside the single tail workgroup)
    int sweep_fused = 0;   // 1: one launch per red+black sweep where the level takes it (mgx_sweep3d.hip); 0 = one per colour (default
                           // while the one-launch kernel measures slower: DESIGN.md section 5)
    int sweep_mid = 1;     // cache-resident levels (33 ... 129 points per row): one launch per red+black sweep (sweep3d_xs_mid_kernel)
    int relax_resident = 1;      // cache-resident levels: all passes of a Relax call in one launch (mgx_resident3d.hip) ...
    int relax_resident_min = 3;  // ... from this many sweeps per call on
    void* resident_buf = nullptr;  // its exchange buffer (tagged face lines)
    size_t resident_bytes = 0;
    int sweep_ilv = 0;     // that kernel with its memory instructions interleaved with the arithmetic instead of issued first
    int sweep_lead = 0;    // planes the red stage of that kernel runs ahead of the black stage (0 = default)
    int sweep_dbg = 0;     // diagnostic builds: 1 = cycle stamps, + 2 * ablation bits (sweep3d_xs_kernel)
    void* sweep_dev = nullptr;        // its device state: launch epoch, finished-workgroup counter, progress words
    unsigned* sweep_abort = nullptr;  // host-mapped word: != 0 once an inter-workgroup wait has given up
    unsigned sync_spin_limit = 1u << 21;  // "sync.spin_limit": polls (each ~1 us) before such a wait gives up
    unsigned handoff_fault = 0;           // "test.handoff_fault": test hook, see SweepSync::fault
    int handoff_broken = 0;   // set when mgx_ctx_check has seen the abort word: the kernels whose workgroups wait for each other are not
                              // used any more on this context (colour passes instead) until mgx_ctx_clear_abort(ctx, 1)
    int gpu_exclusive = 1;    // "gpu.exclusive": 1 = this context has the GPU to itself (the assumption behind those kernels); 0 = the GPU
                              // is shared with other contexts / processes: they are never launched
    mutable int resident_occ = -1;  // workgroups of the resident Relax kernels per CU (occupancy API), -1 = not asked yet
    void* scratch = nullptr;  // small device workspace (reductions, tables)
    size_t scratch_bytes = 0;
    void* rccl_comm = nullptr;  // ncclComm_t
    void* local_group = nullptr;  // mgx_local_group* (in-process test transport)
    int rank = 0, nranks = 1;
    int comm_rehearse = 0;  // mgx_comm_init_rehearsal: rank / nranks are pretended, every peer is this rank itself (timing only)
    void* rehearse_buf = nullptr;  // its scratch: the partner buffer of a send / receive that has none on this rank
    size_t rehearse_bytes = 0;
    int num_cus = 256;
    int pipe_unroll = 7;              // "relax3d.unroll": bit mask, see mgx_ctx_set_param
    int corr_low = 0;                 // "relax3d.corr_low": the fp64 correcting red pass in 8-wave workgroups, two to a CU
    int slab_edges_merged = 1;        // "slab.edges_merged": mgx3dxs_relax_colour_slab2_* makes one launch of its two plane ranges
    int resident_tile = 0;            // "relax3d.resident_tile": 0 = by level, 8 = tiles of 8 x 8 lines always (tests, timing)
    char last_corr_kernel[96] = "";   // the correcting red pass of the most recent interpolate_correct_relax ("" = none: small level or corr_fuse = 0)
    char last_rr_kernel[96] = "";     // the fused black pass + residual + restrict kernel of the most recent call ("" = not fused)
    char last_relax_kernel[96] = "";  // name of the smoother kernel of the most recent colour pass (bench.py: roofline.kernel)
};

struct mgx_event {
    hipEvent_t ev = nullptr;
};

namespace mgx {

void set_error(const char* fmt, ...);

inline int fail(int status, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    set_error("%s", buf);
    return status;
}

// workspace of at least `bytes` (grown on demand; contents undefined)
int workspace(mgx_ctx* ctx, size_t bytes, void** out);
int fill_zero(mgx_ctx* ctx, void* dst, size_t bytes);  // dst[0, bytes) := 0 on the compute stream

inline int ceil_div(long long a, long long b) { return (int)((a + b - 1) / b); }

template <class real>
inline void note_relax_kernel(mgx_ctx* ctx, const char* name, int a, int b, int c, bool fnt = false) {
    snprintf(ctx->last_relax_kernel, sizeof ctx->last_relax_kernel, "%s<%s,%d,%d,%d%s>", name, sizeof(real) == 8 ? "double" : "float",
             a, b, c, fnt ? ",true" : "");
}

inline bool valid_size(int n) { return n >= 3 && ((n - 1) % 2 == 0); }

// t / d for 0 <= t < 2^23, d >= 1 in ~8 instructions instead of the ~40 of a 32-bit integer division: the float product is
// within one of the quotient, one correction step makes it exact.  The one-workgroup tail kernels split linear LDS indices
// into (x, y, z) hundreds of times per launch; with `/` that was most of their run time.
struct SmallDiv {
    int d;
    float r;
    __host__ __device__ explicit SmallDiv(int d_) : d(d_), r(1.0f / (float)d_) {}
    __device__ __forceinline__ int operator()(int t) const {
        int q = (int)((float)t * r);
        const int rem = t - q * d;
        q += (rem >= d) - (rem < 0);
        return q;
    }
};

}  // namespace mgx

#define MGX_HIP(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return mgx::fail(MGX_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                             __FILE__, __LINE__);                                             \
    } while (0)

#define MGX_REQUIRE(cond, status, ...) \
    do {                               \
        if (!(cond)) return mgx::fail(status, __VA_ARGS__); \
    } while (0)

// every C-ABI entry that allocates, launches or synchronises makes the context's device current first: a context may be
// used from a thread other than its creator (ctypes, thread-ranks), whose current device is otherwise 0
#define MGX_USE(ctx) MGX_HIP(hipSetDevice((ctx)->device))
#define MGX_LAUNCH_CHECK() MGX_HIP(hipGetLastError())
// every kernel launch of the library goes through here.  Test hook (mgx_test_set_lds_poison): when armed, each launch is preceded
// by a launch that fills the LDS of every CU with signalling-NaN patterns, so that a kernel reading an LDS word before writing
// it fails its parity test deterministically instead of depending on what the previous kernel happened to leave there.
namespace mgx {
extern int g_poison_lds;
void poison_lds_launch(hipStream_t stream);
}  // namespace mgx
#define MGX_LAUNCH(kernel, grid, block, lds, stream, ...)                  \
    do {                                                                   \
        if (mgx::g_poison_lds) mgx::poison_lds_launch(stream);             \
        hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__); \
    } while (0)
#define MGX_TRY_RET(expr)            \
    do {                             \
        const int st_ = (expr);      \
        if (st_) return st_;         \
    } while (0)

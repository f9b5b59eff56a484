/* mgx.h -- thin C-ABI over the hand-written HIP (gfx950 / MI355X) multigrid kernels.
 *
 * This is the drop-in boundary of the hot path.  The reference
 * (MisterPup/PDE-MultiGrid, NOCUDA_TESI) has no FFI layer: its boundary is the public
 * C++ surface of MultiGrid{2,3}D.  Each entry point below replaces one of those member
 * functions and takes the same arguments as plain pointers and sizes (device pointers
 * instead of host `float*`; `real` = float or double chosen by the _f32/_f64 suffix;
 * the grid spacing / range members the reference reads from Grid3D* are passed
 * explicitly).  The C host layer in include/mg_multigrid.h is the only caller in the
 * product; tests and bench.py reach it through ctypes.
 *
 *   reference member function (file:line)                      replacement
 *   -------------------------------------------------------    -----------------------------
 *   MultiGrid3D::Relax            N3/MultiGrid3D.cpp:489-567   mgx3d_relax_{f32,f64}
 *   MultiGrid3D::CalculateResidual N3/MultiGrid3D.cpp:678-730  mgx3d_residual_*
 *   MultiGrid3D::Restrict         N3/MultiGrid3D.cpp:50-184    mgx3d_restrict_*
 *   MultiGrid3D::Interpolate      N3/MultiGrid3D.cpp:186-335   mgx3d_interpolate_*
 *   MultiGrid3D::ApplyCorrection  N3/MultiGrid3D.cpp:649-676   mgx3d_apply_correction_*
 *   MultiGrid3D::setToValue       N3/MultiGrid3D.cpp:587-621   mgx3d_set_*
 *   Grid3D::InitF                 N3/Grid3D.cpp:78-96          mgx3d_init_f_* (host sin tables)
 *   MultiGrid2D::Relax            N2/MultiGrid2D.cpp:199-273   mgx2d_relax_*
 *   MultiGrid2D::CalculateResidual N2/MultiGrid2D.cpp:367-408  mgx2d_residual_*
 *   MultiGrid2D::Restrict         N2/MultiGrid2D.cpp:63-126    mgx2d_restrict_*
 *   MultiGrid2D::Interpolate      N2/MultiGrid2D.cpp:128-196   mgx2d_interpolate_*
 *   MultiGrid2D::ApplyCorrection  N2/MultiGrid2D.cpp:343-366   mgx2d_apply_correction_*
 *   MultiGrid2D::setToValue       N2/MultiGrid2D.cpp:275-292   mgx2d_set_*
 *   (N3 = NOCUDA_TESI/POISSON_3D(TESI)/, N2 = NOCUDA_TESI/PDE Lyapunov 2D/)
 *
 * Fused forms (bit-identical to the two calls they replace):
 *   CalculateResidual + Restrict   (VCycle, N3/MultiGrid3D.cpp:629-632)  mgx3d_residual_restrict_*
 *   Interpolate + ApplyCorrection  (VCycle, N3/MultiGrid3D.cpp:638-642)  mgx3d_interpolate_correct_*
 *
 * Conventions
 *   - every function returns an int status (MGX_OK = 0); nothing aborts
 *     (the reference asserts, N3/MultiGrid3D.cpp:60-62).  mgx_last_error() gives text.
 *   - arrays are dense, unpadded, x fastest: idx = x + y*sx + z*sx*sy
 *     (N3/MultiGrid3D.cpp:531); sizes are int[dim] = {sx, sy(, sz)}, each 2^k+1.
 *   - all `real*` arguments are DEVICE pointers obtained from mgx_malloc unless the
 *     name starts with host_.
 *   - kernels run on the context's compute stream; calls are asynchronous with respect
 *     to the host unless stated; mgx_ctx_sync() waits.
 *   - one context per host thread; no global state.
 */
#ifndef MGX_H
#define MGX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    MGX_OK = 0,
    MGX_ERR_INVALID = 1, /* NULL pointer, bad enum, negative count ...            */
    MGX_ERR_SIZE = 2,    /* size relation violated (the reference would assert)   */
    MGX_ERR_HIP = 3,     /* a HIP runtime call failed                             */
    MGX_ERR_NOMEM = 4,
    MGX_ERR_RCCL = 5,
    MGX_ERR_NOGPU = 6    /* no HIP device visible                                 */
} mgx_status;

/* Residual variants: REF_COMPAT reproduces the reference's
 * sign quirk ((N-2v-S)/hy2, (D-2v-U)/hz2, N3/MultiGrid3D.cpp:723); CORRECT uses +S, +U. */
typedef enum { MGX_RESIDUAL_REF_COMPAT = 0, MGX_RESIDUAL_CORRECT = 1 } mgx_residual_mode;

typedef struct mgx_ctx mgx_ctx;
typedef struct mgx_event mgx_event;

const char* mgx_status_string(int status);
const char* mgx_last_error(void); /* thread-local, valid until the next failing call */
void mgx_set_last_error(const char* msg); /* for layers above (include/mg_multigrid.h) */
const char* mgx_version(void);

/* ---- context / device ------------------------------------------------------------ */
int mgx_device_count(int* count);
int mgx_ctx_create(int device, mgx_ctx** out);
int mgx_ctx_destroy(mgx_ctx* ctx);
int mgx_ctx_sync(mgx_ctx* ctx);                 /* waits for the compute and comm streams, then mgx_ctx_check */
/* MGX_OK unless an inter-workgroup wait of a kernel whose workgroups hand data to each other (the one-launch red+black
 * sweep of mgx3dxs_relax_pp; the resident Relax kernel that runs all passes of a long Relax call on a cache-resident
 * level) has given up on this context -- its workgroups were not resident together; results of that launch are invalid
 * ("relax3d.fused" = 0 / "relax3d.resident" = 0 avoid the kernels).  Meaningful after a synchronisation. */
int mgx_ctx_check(mgx_ctx* ctx);
/* After mgx_ctx_check / mgx_ctx_sync reported a given-up wait: the context has stopped launching those kernels (it runs the
 * colour-pass kernels instead, same results) and keeps reporting the condition until this call clears it.  Synchronises,
 * resets the hand-off state on the device and the abort word; reenable != 0 also lets the context use the kernels again
 * (the caller vouches that whatever kept their workgroups from being resident together is gone).  Data written by the
 * aborted launch stays invalid: upload or recompute it.
 * These kernels ASSUME the context has the GPU to itself while they run (their grids are sized to be resident at once:
 * tiles <= CUs x occupancy).  A process that shares the GPU with other contexts or processes sets "gpu.exclusive" to 0
 * (mgx_ctx_set_param): they are then never launched.  "sync.spin_limit" (default 2^21 polls, ~2 s) bounds every wait. */
int mgx_ctx_clear_abort(mgx_ctx* ctx, int reenable);
/* allocates now what some kernels would allocate on first use (the progress words and exchange buffer of the kernels whose
 * workgroups hand data to each other, ~17 MB): the hierarchies call it when they are created, so that their first cycle can
 * be captured into a HIP graph */
int mgx_ctx_prepare(mgx_ctx* ctx);
int mgx_ctx_device(const mgx_ctx* ctx, int* device);
/* tuning knobs of the x-split smoother kernel (speed only, never results): "relax3d.ty" waves
 * per block and "relax3d.rows" consecutive rows per lane, each in {1,2,4,8}; "relax3d.zchunk"
 * planes per block (0 = automatic); "relax3d.xcd" 0/1/2 block-to-tile mapping (2 = every XCD owns a y-slab and walks z);
 * "relax3d.wave_planes" slab height of the time-skewed pass order (< 0 automatic, 0 = whole-grid passes);
 * "cycle2d.tile" tile edge of the cache-resident 2D kernels (0 = by level size, 16, 32, 64), "cycle2d.tail_points" largest
 * top level (in points, <= 5120) the one-workgroup tail kernel of the 2D cycle takes;
 * "relax3d.lds" kernel / shape code of the pipelined smoother (-1 automatic), "relax3d.corr_fuse", "relax3d.v2",
 * "relax3d.zero_first", "relax3d.small" 0/1 switches of the fused forms; "residual_restrict3d.stream" 0..3 kernel choice,
 * ".pzchunk" coarse planes per run (0 automatic), ".tyw" waves per workgroup, ".cr" coarse rows per lane, ".rows" fine rows
 * per wave of the pipelined kernel (0 = by level size, 2, 4), ".xcd" 0/1/2 XCD-aware block order, ".rcp" 0/1: with
 * power-of-two squared spacings the residual multiplies by the exact reciprocals instead of dividing (same bits);
 * "relax3d.corr_v2" 0/1: fp32 on wide levels, the correcting red pass with two pairs per lane like the plain passes;
 * "relax3d.zero_sweep" 0/1: relax_from_zero on the pipelined levels runs its first red and black pass as one launch;
 * "relax3d.resident" 0 / 1 / 2 and "relax3d.resident_min" (sweeps per call, default 3): all colour passes of a Relax call on a level
 * of 33 ... 129 points per row in one launch (single-rank contexts only) -- off / the tiles exchange once per sweep (default) / once
 * per pass; "relax3d.resident_tile" 0 (by level) / 8: tiles of 8 x 8 lines always;
 * "rr3d.black" 0 / 1 / 2: the last black pass of the pre-smoothing inside the residual+restrict launch -- off / on the
 * HBM-bound levels / wherever the geometry allows (tests), "rr3d.black_waves" 0 (by precision), 8 (two workgroups per CU), 12, 16 waves per workgroup.
 * "gpu.exclusive" 1 (default) / 0: 0 = the GPU is shared with other contexts or processes, so the kernels whose workgroups
 * wait for each other (resident Relax, one-launch sweep of 513-point rows) are never launched; "sync.spin_limit" polls (~1 us
 * each, default 2^21) before such a wait gives up (see mgx_ctx_check); "test.handoff_fault" != 0 is a TEST HOOK that makes
 * workgroup 0 of those kernels wait for tags nobody writes (the give-up path under test).
 * Unknown names and out-of-range values are rejected (MGX_ERR_INVALID). */
int mgx_ctx_set_param(mgx_ctx* ctx, const char* name, int value);
/* name (kernel<template arguments>) of the smoother kernel the most recent 3D x-split colour pass launched; "" if none.
 * bench.py reports it as roofline.kernel so that the PMC traffic figure is attached only to the kernel it was taken from */
const char* mgx_ctx_last_relax_kernel(const mgx_ctx* ctx);
/* name of the kernel that ran the last black pass together with residual + restrict in the most recent
 * mgx3dxs_smooth_residual_restrict_* call; "" when that call ran the operators one after the other */
const char* mgx_ctx_last_rr_kernel(const mgx_ctx* ctx);
/* name of the kernel that ran the correcting red pass (the correction read on the fly) in the most recent
 * mgx3dxs_interpolate_correct_relax_* call; "" when that call corrected in a pass of its own */
const char* mgx_ctx_last_corr_kernel(const mgx_ctx* ctx);
/* TEST HOOK, process-wide: on != 0 -> every kernel launch of the library is preceded by a launch that fills the LDS of every
 * CU with signalling-NaN patterns (a kernel that reads an LDS word before writing it then fails its parity test for certain
 * instead of depending on the previous launch's leftovers).  Costs ~15 us per launch; results are unchanged by contract. */
int mgx_test_set_lds_poison(int on);
/* self-test of the above: one workgroup per CU reads all of its LDS without writing any; *fraction = share of the words
 * that carry the poison pattern (1.0 with poisoning on: the launch is preceded by the poisoning launch like any other) */
int mgx_test_lds_probe(mgx_ctx* ctx, double* fraction);
/* raw hipStream_t of the compute stream (for callers that bring their own HIP code) */
int mgx_ctx_stream(const mgx_ctx* ctx, void** hip_stream);

/* ---- device memory ------------------------------------------------------------------ */
int mgx_malloc(mgx_ctx* ctx, size_t bytes, void** dptr);
int mgx_free(mgx_ctx* ctx, void* dptr);
int mgx_memcpy_h2d(mgx_ctx* ctx, void* dst, const void* host_src, size_t bytes); /* blocking */
int mgx_memcpy_d2h(mgx_ctx* ctx, void* host_dst, const void* src, size_t bytes); /* blocking */
int mgx_memcpy_d2d(mgx_ctx* ctx, void* dst, const void* src, size_t bytes);      /* async    */
int mgx_memset_zero(mgx_ctx* ctx, void* dst, size_t bytes);                      /* async    */

/* ---- timing on the compute stream (HIP events) ------------------------------------ */
/* HIP graphs for the launch-bound parts (2D cycles, coarse 3D levels): everything the calls between _begin and _end
 * enqueue on the context's compute stream is captured instead of executed and instantiated as one executable graph;
 * _launch replays it with a single launch.  The host layer's VCycle does this itself when `use_graph` is set. */
/* A captured graph freezes the launch sequence as it was at capture time, context parameters (mgx_ctx_set_param) included:
 * change a parameter -> capture again.  The host layer re-captures when the cycle's own arguments change, not on parameters. */
int mgx_graph_begin(mgx_ctx* ctx);
int mgx_graph_end(mgx_ctx* ctx, void** graph_exec);
int mgx_graph_launch(mgx_ctx* ctx, void* graph_exec);
int mgx_graph_destroy(mgx_ctx* ctx, void* graph_exec);
int mgx_event_create(mgx_ctx* ctx, mgx_event** out);
int mgx_event_destroy(mgx_ctx* ctx, mgx_event* ev);
int mgx_event_record(mgx_ctx* ctx, mgx_event* ev);
int mgx_event_elapsed_ms(mgx_ctx* ctx, mgx_event* start, mgx_event* stop, float* ms); /* syncs on stop */

/* ---- operators ------------------------------------------------------------------------
 * Declared once per real type through MGX_DECLARE_OPS(suffix, real).
 *
 * mgx3d_relax: `ncycles` red-black Gauss-Seidel sweeps, in place (one launch per colour).
 * mgx3d_residual: r = f - A v on the interior (mode selects the sign variant), 0 on the boundary.
 * mgx3d_restrict: 27-point full weighting, boundary = injection; cn must equal (fn-1)/2+1.
 * mgx3d_interpolate: trilinear, interior of `fine` only (boundary untouched).
 * mgx3d_apply_correction: fine += err on the interior.
 * mgx3d_set: fill; modify_boundaries = 0 leaves the boundary untouched.
 * mgx3d_residual_restrict: coarse_f = Restrict(CalculateResidual(v, f)) without storing
 *   the fine residual.
 * mgx3d_interpolate_correct: v += Interpolate(coarse_v) on the interior.
 * mgx3d_init_f: f[x,y,z] = (real)(c * tx[x] * ty[y] * tz[z]) evaluated in double, left to
 *   right (Grid3D::InitF with host-computed sin tables; tables are host pointers).
 * mgx3d_jacobi: ADDITION (north_star names weighted Jacobi; the reference has only red-black Gauss-Seidel):
 *   `ncycles` sweeps of v <- v + omega*(u - v), u = the Gauss-Seidel value from the OLD iterate; `tmp` is a
 *   second n-point array (ping-pong), the result always ends in v.  Parity unpinned by the reference.
 * mgx3d_diff_stats: Grid3D::PrintDiff (N3/Grid3D.cpp:136-159) as a device reduction: diff = realSol - v with
 *   realSol = (real)(tx[x]*ty[y]*tz[z]) (host sin tables); host_out = {sum|diff|, max|diff|, sum diff^2,
 *   sum realSol^2} over all points.
 * mgx2d_mean_abs_error: mean over the interior of |v - (2x^2-4xy+2y^2)| -- the accuracy metric of the thesis
 *   (PrintMeanAbsoluteError, CUDA_TESI/CUDA Lyapunov 2D/Grid2D.cu:123-154), reduced on the device.
 * mgx3dxs_*: the same operators on the device-internal "x-split" layout
 *   idx = (x>>1) + (x&1)*H + y*P + z*P*sy,  H = roundup((sx+1)/2, A), P = H + roundup(sx/2, A),
 *   A = 128/sizeof(real) elements
 *   (every x-row stored as its even-x half followed by its odd-x half, both starting on a 128-byte
 *   boundary; rows and planes in the reference order).  In row (y,z) the points of one colour are one
 *   contiguous, line-aligned half-row, so a red+black sweep moves the algorithmic minimum of 3 reals per
 *   point through HBM in full cache lines.  Results are bit-identical to the natural-layout operators;
 *   mgx3dxs_pack / _unpack convert (out of place); array sizes: mgx3dxs_elems.
 * mgx_norm2: sum of squares of `count` reals in double (wave-wide shuffle reduction +
 *   one atomic per block); host result, blocking.  An addition: the reference has no norm.
 */
#define MGX_DECLARE_OPS(SFX, real)                                                                      \
    int mgx3d_relax_##SFX(mgx_ctx* ctx, real* v, const real* f, const int n[3], const real h[3],        \
                          int ncycles);                                                                 \
    /* relax_from_zero: v := 0 everywhere (setToValue(v, 0, true), N3/MultiGrid3D.cpp:634), then ncycles */ \
    /* sweeps (:626).  rim_is_zero != 0: the boundary (and pad) entries of v are zero already -- then      */ \
    /* nothing is filled and the first red pass does not read v; same bits either way.                   */ \
    int mgx3d_relax_from_zero_##SFX(mgx_ctx* ctx, real* v, const real* f, const int n[3], const real h[3], \
                                    int ncycles, int rim_is_zero);                                      \
    int mgx3d_residual_##SFX(mgx_ctx* ctx, const real* v, const real* f, real* r, const int n[3],       \
                             const real h[3], int mode);                                                \
    int mgx3d_restrict_##SFX(mgx_ctx* ctx, const real* fine, const int fn[3], real* coarse,             \
                             const int cn[3]);                                                          \
    int mgx3d_interpolate_##SFX(mgx_ctx* ctx, real* fine, const int fn[3], const real* coarse,          \
                                const int cn[3]);                                                       \
    int mgx3d_apply_correction_##SFX(mgx_ctx* ctx, real* fine, const int fn[3], const real* err,        \
                                     const int en[3]);                                                  \
    int mgx3d_set_##SFX(mgx_ctx* ctx, real* grid, const int n[3], real value, int modify_boundaries);   \
    int mgx3d_residual_restrict_##SFX(mgx_ctx* ctx, const real* v, const real* f, const int n[3],       \
                                      const real h[3], int mode, real* coarse_f, const int cn[3]);      \
    /* _keep_rim: the boundary entries of coarse_f are NOT rewritten (the caller knows they are 0,      */ \
    /* e.g. from the previous cycle); same interior results, one fill of the coarse array less.       */ \
    int mgx3d_residual_restrict_keep_rim_##SFX(mgx_ctx* ctx, const real* v, const real* f,              \
                                               const int n[3], const real h[3], int mode,               \
                                               real* coarse_f, const int cn[3]);                        \
    int mgx3d_interpolate_correct_##SFX(mgx_ctx* ctx, real* v, const int n[3], const real* coarse_v,    \
                                        const int cn[3]);                                               \
    int mgx3d_init_f_##SFX(mgx_ctx* ctx, real* f, const int n[3], double c, const double* host_tx,      \
                           const double* host_ty, const double* host_tz);                               \
    int mgx3d_jacobi_##SFX(mgx_ctx* ctx, real* v, real* tmp, const real* f, const int n[3],             \
                           const real h[3], real omega, int ncycles);                                   \
    /* vcycle_tail: the whole VCycle(v1, v2) (N3/MultiGrid3D.cpp:623-647) over levels[0 .. nlev), each  */ \
    /* at most 17 points per axis (n and h flattened {x0, y0, z0, x1, ...}; v / f HOST arrays of device  */ \
    /* pointers), in ONE workgroup with every level in LDS; top_zero != 0: v of levels[0] is taken as   */ \
    /* zeros without being read.  Leaves v of all levels and f of levels 1.. as the launch-per-operator */ \
    /* cycle does.  _fits: 1 when these levels can run that way ("relax3d.small" = 0 turns it off)      */ \
    int mgx3d_vcycle_tail_##SFX(mgx_ctx* ctx, int nlev, real* const* v, real* const* f, const int* n,   \
                                const real* h, int v1, int v2, int mode, int top_zero);                 \
    int mgx3d_vcycle_tail_fits_##SFX(const mgx_ctx* ctx, int nlev, const int* n);                       \
    int mgx3d_diff_stats_##SFX(mgx_ctx* ctx, const real* v, const int n[3], const double* host_tx,      \
                               const double* host_ty, const double* host_tz, double host_out[4]);       \
    /* x-split twins: same operators on arrays whose x-rows are de-interleaved (see below).  An        */ \
    /* x-split array of an (sx, sy, sz) grid has mgx3dxs_elems elements (rows are padded to cache      */ \
    /* lines), one z-plane has mgx3dxs_plane_elems; pad entries must be zero (mgx_memset_zero once).   */ \
    size_t mgx3dxs_elems_##SFX(const int n[3]);                                                         \
    size_t mgx3dxs_plane_elems_##SFX(int sx, int sy);                                                   \
    int mgx3dxs_pack_##SFX(mgx_ctx* ctx, const real* natural, real* xsplit, const int n[3]);            \
    int mgx3dxs_unpack_##SFX(mgx_ctx* ctx, const real* xsplit, real* natural, const int n[3]);          \
    int mgx3dxs_relax_##SFX(mgx_ctx* ctx, real* v, const real* f, const int n[3], const real h[3],      \
                            int ncycles);                                                               \
    int mgx3dxs_relax_from_zero_##SFX(mgx_ctx* ctx, real* v, const real* f, const int n[3],             \
                                      const real h[3], int ncycles, int rim_is_zero);                   \
    /* relax_pp: the same `ncycles` sweeps of v with a second array w (mgx3dxs_elems reals) as ping-pong */ \
    /* partner: on levels of 513-point rows every red+black sweep is ONE launch, out of place, that      */ \
    /* moves 2.5 instead of 3 reals per point (the black stage takes the new red values of its own tile  */ \
    /* from LDS and of the neighbouring tiles from memory, ordered by progress words; "relax3d.fused" = 0 */ \
    /* turns it off).  Result in v, bit-identical to mgx3dxs_relax; w is scratch.  w_rim_valid != 0: the  */ \
    /* caller vouches that the boundary entries of w equal those of v already.  _takes: 1 when a call    */ \
    /* with these sizes would run the one-launch sweeps (else it is mgx3dxs_relax).                      */ \
    int mgx3dxs_relax_pp_##SFX(mgx_ctx* ctx, real* v, real* w, const real* f, const int n[3],           \
                               const real h[3], int ncycles, int w_rim_valid);                          \
    int mgx3dxs_relax_pp_takes_##SFX(const mgx_ctx* ctx, const int n[3], int ncycles);                  \
    /* On cache-resident levels (33 ... 129 points per row) relax_pp also runs one launch per sweep      */ \
    /* (tile + halo in LDS, the neighbours' red values recomputed; "relax3d.fused_mid" = 0 turns it off). */ \
    /* relax_from_zero_pp / interpolate_correct_relax_pp: mgx3dxs_relax_from_zero /                      */ \
    /* mgx3dxs_interpolate_correct_relax with the same partner array for their sweeps; the first sweep of */ \
    /* relax_from_zero_pp does not read v on those levels (even ncycles, rim_is_zero).  Same bits.        */ \
    int mgx3dxs_relax_from_zero_pp_##SFX(mgx_ctx* ctx, real* v, real* w, const real* f, const int n[3], \
                                         const real h[3], int ncycles, int rim_is_zero, int w_rim_valid); \
    /* _takes: 1 when that call runs its sweeps on the partner array (and has brought w's boundary up to */ \
    /* date), 0 when it is plain mgx3dxs_relax_from_zero                                                 */ \
    /* sweep_once: ONE red+black sweep vin -> vout by the level's one-launch kernel (MGX_ERR_SIZE if it  */ \
    /* has none): interior points of vout are written, nothing else; zero != 0: vin counts as zeros and   */ \
    /* is not read (cache-resident levels).  The unit the _pp drivers are made of.                        */ \
    int mgx3dxs_sweep_once_##SFX(mgx_ctx* ctx, const real* vin, real* vout, const real* f,              \
                                 const int n[3], const real h[3], int zero);                            \
    int mgx3dxs_relax_from_zero_pp_takes_##SFX(const mgx_ctx* ctx, const int n[3], int ncycles,         \
                                               int rim_is_zero);                                        \
    int mgx3dxs_interpolate_correct_relax_pp_##SFX(mgx_ctx* ctx, real* v, real* w, const real* f,       \
                                                   const int n[3], const real h[3], const real* coarse_v, \
                                                   const int cn[3], int ncycles, int w_rim_valid);      \
    int mgx3dxs_residual_##SFX(mgx_ctx* ctx, const real* v, const real* f, real* r, const int n[3],     \
                               const real h[3], int mode);                                              \
    int mgx3dxs_restrict_##SFX(mgx_ctx* ctx, const real* fine, const int fn[3], real* coarse,           \
                               const int cn[3]);                                                        \
    int mgx3dxs_interpolate_##SFX(mgx_ctx* ctx, real* fine, const int fn[3], const real* coarse,        \
                                  const int cn[3]);                                                     \
    int mgx3dxs_apply_correction_##SFX(mgx_ctx* ctx, real* fine, const int fn[3], const real* err,      \
                                       const int en[3]);                                                \
    int mgx3dxs_set_##SFX(mgx_ctx* ctx, real* grid, const int n[3], real value, int modify_boundaries); \
    int mgx3dxs_residual_restrict_##SFX(mgx_ctx* ctx, const real* v, const real* f, const int n[3],     \
                                        const real h[3], int mode, real* coarse_f, const int cn[3]);    \
    int mgx3dxs_residual_restrict_keep_rim_##SFX(mgx_ctx* ctx, const real* v, const real* f,            \
                                                 const int n[3], const real h[3], int mode,             \
                                                 real* coarse_f, const int cn[3]);                      \
    int mgx3dxs_interpolate_correct_##SFX(mgx_ctx* ctx, real* v, const int n[3], const real* coarse_v,  \
                                          const int cn[3]);                                             \
    int mgx3dxs_init_f_##SFX(mgx_ctx* ctx, real* f, const int n[3], double c, const double* host_tx,    \
                             const double* host_ty, const double* host_tz);                             \
    int mgx3dxs_jacobi_##SFX(mgx_ctx* ctx, real* v, real* tmp, const real* f, const int n[3],           \
                             const real h[3], real omega, int ncycles);                                 \
    int mgx3dxs_vcycle_tail_##SFX(mgx_ctx* ctx, int nlev, real* const* v, real* const* f, const int* n, \
                                  const real* h, int v1, int v2, int mode, int top_zero);               \
    int mgx3dxs_vcycle_tail_fits_##SFX(const mgx_ctx* ctx, int nlev, const int* n);                     \
    int mgx3dxs_diff_stats_##SFX(mgx_ctx* ctx, const real* v, const int n[3], const double* host_tx,    \
                                 const double* host_ty, const double* host_tz, double host_out[4]);     \
    /* z-slab forms for the multi-GPU decomposition.  A slab is a local x-split array of consecutive */ \
    /* z-planes of a level whose GLOBAL sizes are n[] (cn[] for the coarse level); it starts at       */ \
    /* global plane zoff.  relax_colour_slab: ONE colour pass (colour 0 = red: (x+y+z_global) even)   */ \
    /* over the local planes [zbeg, zend); the planes next to that range are read as ghosts.          */ \
    /* residual_restrict_slab / interpolate_correct_slab: global coarse planes [pzbeg, pzend);        */ \
    /* interpolate_correct writes the fine planes 2pz and 2pz+1 of every listed pz (z = 0 skipped).   */ \
    int mgx3dxs_relax_colour_slab_##SFX(mgx_ctx* ctx, real* v, const real* f, int sx, int sy,           \
                                        const real h[3], int colour, int zbeg, int zend, int zoff);     \
    /* relax_colour_slab2: the same pass over TWO runs of local planes [zb1, ze1) and [zb2, ze2), ze1   */ \
    /* <= zb2, in one launch where both are short (the bottom and the top edge of a slab)               */ \
    int mgx3dxs_relax_colour_slab2_##SFX(mgx_ctx* ctx, real* v, const real* f, int sx, int sy,          \
                                         const real h[3], int colour, int zb1, int ze1, int zb2,        \
                                         int ze2, int zoff);                                             \
    /* relax_zero_colour_slab: the same pass on a slab whose v counts as all zeros (the coarse error    */ \
    /* at the start of a cycle): v is not read, no ghost plane is needed; boundary entries must be 0   */ \
    int mgx3dxs_relax_zero_colour_slab_##SFX(mgx_ctx* ctx, real* v, const real* f, int sx, int sy,      \
                                             const real h[3], int colour, int zbeg, int zend, int zoff); \
    int mgx3dxs_residual_restrict_slab_##SFX(mgx_ctx* ctx, const real* v, const real* f,                \
                                             const int n[3], int fzoff, const real h[3], int mode,      \
                                             real* coarse_f, const int cn[3], int czoff, int pzbeg,     \
                                             int pzend);                                                \
    /* residual_sumsq_slab (ADDITION, the reference has no norm): sum of the squared residual over the */ \
    /* (x, y)-interior points of the local planes [zbeg, zend) -> *dev_out (device double), async on   */ \
    /* the compute stream, reduced in a fixed order (same bits on every run)                           */ \
    int mgx3dxs_residual_sumsq_slab_##SFX(mgx_ctx* ctx, const real* v, const real* f, int sx, int sy,   \
                                          const real h[3], int mode, int zbeg, int zend,                \
                                          double* dev_out);                                             \
    /* FMG on slabs: Restrict (coarse planes [pzbeg, pzend); boundary points by injection) and plain  */ \
    /* Interpolate (fine planes 2pz, 2pz+1 of every listed pz, interior points, z = 0 skipped).       */ \
    /* setToValue(.., false) on the LOCAL planes [zbeg, zend) of a slab: their (x, y)-interior points  */ \
    int mgx3dxs_set_interior_slab_##SFX(mgx_ctx* ctx, real* grid, int sx, int sy, int zbeg, int zend,   \
                                        real value);                                                    \
    int mgx3dxs_restrict_slab_##SFX(mgx_ctx* ctx, const real* fine, const int fn[3], int fzoff,         \
                                    real* coarse, const int cn[3], int czoff, int pzbeg, int pzend);    \
    int mgx3dxs_interpolate_slab_##SFX(mgx_ctx* ctx, real* fine, const int fn[3], int fzoff,            \
                                       const real* coarse, const int cn[3], int czoff, int pzbeg,       \
                                       int pzend);                                                      \
    int mgx3dxs_interpolate_correct_slab_##SFX(mgx_ctx* ctx, real* v, const int n[3], int fzoff,        \
                                               const real* coarse_v, const int cn[3], int czoff,        \
                                               int pzbeg, int pzend);                                   \
    /* interpolate_correct_relax: v += Interpolate(coarse_v) on the interior, then ncycles >= 1 sweeps   */ \
    /* (N3/MultiGrid3D.cpp:638-645 in one call).  On large levels the first red pass reads the black     */ \
    /* points THROUGH the correction instead of the correction being stored first ("relax3d.corr_fuse"  */ \
    /* = 0 turns that off); the result is bit-identical to interpolate_correct followed by relax.       */ \
    int mgx3dxs_interpolate_correct_relax_##SFX(mgx_ctx* ctx, real* v, const real* f, const int n[3],   \
                                                const real h[3], const real* coarse_v, const int cn[3], \
                                                int ncycles);                                           \
    /* smooth_residual_restrict: the way down on one level in one call -- Relax(ncycles) (from_zero != 0: */ \
    /* on v = 0 as relax_from_zero, v_rim_is_zero as there), CalculateResidual, Restrict                  */ \
    /* (N3/MultiGrid3D.cpp:626-632; coarse_rim_is_zero as residual_restrict_keep_rim).  On the HBM-bound */ \
    /* levels the LAST BLACK PASS runs inside the residual+restrict launch (the black values are relaxed  */ \
    /* one plane ahead of the residual and stored on the way; "rr3d.black" = 0 turns that off); v and     */ \
    /* coarse_f are bit-identical to relax followed by residual_restrict.                                 */ \
    int mgx3dxs_smooth_residual_restrict_##SFX(mgx_ctx* ctx, real* v, const real* f, const int n[3],    \
                                               const real h[3], int ncycles, int from_zero,             \
                                               int v_rim_is_zero, int mode, real* coarse_f,             \
                                               const int cn[3], int coarse_rim_is_zero);                \
    /* relax_rr_slab: that fused launch alone, on a z-slab (or the whole grid): the BLACK pass of the      */ \
    /* GLOBAL fine planes [2 pzbeg - 1, 2 pzend - 1] + residual + restrict into the GLOBAL coarse planes   */ \
    /* [pzbeg, pzend).  n / cn are global sizes, v / f start at global plane fzoff (even), coarse_f at     */ \
    /* global coarse plane czoff.  It reads only red values of v (fine planes 2 pzbeg - 3 ... 2 pzend + 1, */ \
    /* clipped to the grid: they must be present and current) and f.  relax_rr_takes: 1 when a level of   */ \
    /* these global sizes has the kernel; otherwise relax_rr_slab fails with MGX_ERR_INVALID.              */ \
    int mgx3dxs_relax_rr_takes_##SFX(const mgx_ctx* ctx, const int n[3], const int cn[3]);              \
    int mgx3dxs_relax_rr_slab_##SFX(mgx_ctx* ctx, real* v, const real* f, const int n[3], int fzoff,    \
                                    const real h[3], int mode, real* coarse_f, const int cn[3],         \
                                    int czoff, int pzbeg, int pzend);                                   \
    /* The same on a z-slab (the post-smoothing of the slab-decomposed cycle), in its two pieces.  n / cn */ \
    /* are GLOBAL sizes, v / f start at global plane fzoff (even), coarse_v at global plane czoff and     */ \
    /* holds cplanes planes.  corr_fused_takes: 1 when a level with these rows and `nplanes` planes to    */ \
    /* update runs the on-the-fly form.  correct_pset_slab: the tile-edge cells ("set P") of the GLOBAL   */ \
    /* fine planes [zmin, zmax) get v += Interpolate(coarse_v) in place (black points) -- list the ghost  */ \
    /* planes next to the updated range too, the red pass reads them.  (Since the end of round 3 the set   */ \
    /* is EMPTY for the one-pair-per-lane kernel -- fp64, and fp32 rows of < 513 points -- which corrects  */ \
    /* everything it reads itself: the call is then a no-op; the fp32 two-pair kernel keeps the cell rows.) */ \
    /* relax_corr_colour_slab: the RED                                                                    */ \
    /* pass over the LOCAL planes [zbeg, zend), every other black value read through the correction; the  */ \
    /* black pass that must follow rewrites every black interior point.                                  */ \
    int mgx3dxs_corr_fused_takes_##SFX(const mgx_ctx* ctx, const int n[3], int nplanes);                \
    int mgx3dxs_correct_pset_slab_##SFX(mgx_ctx* ctx, real* v, const int n[3], int fzoff,               \
                                        const real* coarse_v, const int cn[3], int czoff, int zmin,     \
                                        int zmax);                                                      \
    int mgx3dxs_relax_corr_colour_slab_##SFX(mgx_ctx* ctx, real* v, const real* f, const int n[3],      \
                                             int fzoff, const real h[3], const real* coarse_v,          \
                                             const int cn[3], int czoff, int cplanes, int zbeg,         \
                                             int zend);                                                 \
    /* _colour forms: only the points with (x + y + z_global) % 2 == colour are corrected (-1 = all).  */ \
    /* The cycle passes colour 1 (black) when red-black sweeps follow: the red pass rewrites every red */ \
    /* interior point from black neighbours alone, so a corrected red value is never read.            */ \
    int mgx3dxs_interpolate_correct_colour_##SFX(mgx_ctx* ctx, real* v, const int n[3],                 \
                                                 const real* coarse_v, const int cn[3], int colour);    \
    int mgx3dxs_interpolate_correct_colour_slab_##SFX(mgx_ctx* ctx, real* v, const int n[3], int fzoff, \
                                                      const real* coarse_v, const int cn[3], int czoff, \
                                                      int pzbeg, int pzend, int colour);                \
    int mgx2d_relax_##SFX(mgx_ctx* ctx, real* v, const real* f, const int n[2], const real h[2],        \
                          const real a[2], const real A[4], int alfa, int ncycles);                     \
    int mgx2d_residual_##SFX(mgx_ctx* ctx, const real* v, const real* f, real* r, const int n[2],       \
                             const real h[2], const real a[2], const real A[4], int alfa);              \
    int mgx2d_restrict_##SFX(mgx_ctx* ctx, const real* fine, const int fn[2], real* coarse,             \
                             const int cn[2]);                                                          \
    /* fused forms of the cycle (N2/MultiGrid2D.cpp:320-323, 333-335): no residual / error array */    \
    int mgx2d_residual_restrict_##SFX(mgx_ctx* ctx, const real* v, const real* f, const int n[2],       \
                                      const real h[2], const real a[2], const real A[4], int alfa,      \
                                      real* coarse_f, const int cn[2]);                                 \
    int mgx2d_interpolate_correct_##SFX(mgx_ctx* ctx, real* v, const int n[2], const real* coarse_v,    \
                                        const int cn[2]);                                               \
    /* Cache-resident cycle kernels of the 2D path (the 1025^2 hierarchy never leaves L2 / Infinity     */ \
    /* Cache: the cycle is bound by launches, not HBM).  The upwind stencil {C, x+1, y+1} is one-sided, */ \
    /* so a workgroup that holds a tile plus a halo on the +x / +y side in LDS runs all colour passes   */ \
    /* of a Relax call on it, together with the operator next to it in VCycle -- bit-identical to the   */ \
    /* separate calls, OUT OF PLACE (tiles overlap: v_in != v_out), 0 <= ncycles <= 4:                  */ \
    /*   relax_residual_restrict: v_out = Relax(v_in, ncycles); coarse_f = Restrict(CalculateResidual(  */ \
    /*     v_out)) unless coarse_f is NULL (N2/MultiGrid2D.cpp:317-323); v_zero != 0: v_in is taken as  */ \
    /*     all zeros without being read (the coarse error after setToValue(v, 0, true), :326)           */ \
    /*   interpolate_correct_relax: v_out = Relax(v_in + Interpolate(coarse_v), ncycles)   (:333-338)   */ \
    /*   vcycle_tail: the whole VCycle(v1, v2) over levels[0 .. nlev) -- each at most 65^2, n and h     */ \
    /*     flattened {x0, y0, x1, y1, ...}, v / f HOST arrays of device pointers -- in ONE workgroup     */ \
    /*     with every level in LDS (:314-340); leaves v of all levels and f of levels 1.. as the        */ \
    /*     launch-per-operator cycle does.  _fits: may these levels run as one tail launch (they fit    */ \
    /*     into LDS and the top level has at most "cycle2d.tail_points" points: 1) or not (0)           */ \
    int mgx2d_relax_residual_restrict_##SFX(mgx_ctx* ctx, const real* v_in, real* v_out, const real* f, \
                                            const int n[2], const real h[2], const real a[2],           \
                                            const real A[4], int alfa, int ncycles, int v_zero,         \
                                            real* coarse_f, const int cn[2]);                           \
    int mgx2d_interpolate_correct_relax_##SFX(mgx_ctx* ctx, const real* v_in, real* v_out,              \
                                              const real* f, const int n[2], const real h[2],           \
                                              const real a[2], const real A[4], int alfa,               \
                                              const real* coarse_v, const int cn[2], int ncycles);      \
    int mgx2d_vcycle_tail_##SFX(mgx_ctx* ctx, int nlev, real* const* v, real* const* f, const int* n,   \
                                const real* h, const real a[2], const real A[4], int alfa, int v1,      \
                                int v2, int top_zero);                                                  \
    int mgx2d_vcycle_tail_fits_##SFX(const mgx_ctx* ctx, int nlev, const int* n);                       \
    int mgx2d_interpolate_##SFX(mgx_ctx* ctx, real* fine, const int fn[2], const real* coarse,          \
                                const int cn[2]);                                                       \
    int mgx2d_apply_correction_##SFX(mgx_ctx* ctx, real* fine, const int fn[2], const real* err,        \
                                     const int en[2]);                                                  \
    int mgx2d_set_##SFX(mgx_ctx* ctx, real* grid, const int n[2], real value, int modify_boundaries);   \
    int mgx2d_jacobi_##SFX(mgx_ctx* ctx, real* v, real* tmp, const real* f, const int n[2],             \
                           const real h[2], const real a[2], const real A[4], int alfa, real omega,     \
                           int ncycles);                                                                \
    /* init_v: Grid2D::InitV (N2/Grid2D.cpp:50-68) on the device: boundary 2*xj*xj-4*xj*yi+2*yi*yi with   */ \
    /* xj = a[0] + x*h[0], yi = a[1] + y*h[1] in `real`, interior 0; bit-identical to the host loop      */ \
    int mgx2d_init_v_##SFX(mgx_ctx* ctx, real* v, const int n[2], const real h[2], const real a[2]);    \
    int mgx2d_mean_abs_error_##SFX(mgx_ctx* ctx, const real* v, const int n[2], const real h[2],        \
                                   const real a[2], double* host_mean);                                 \
    int mgx_norm2_##SFX(mgx_ctx* ctx, const real* x, size_t count, double* host_sumsq);

MGX_DECLARE_OPS(f32, float)
MGX_DECLARE_OPS(f64, double)

/* ---- multi-GPU: z-slab halo exchange over RCCL (xGMI) ------------------------------
 * One process per GPU.  The unique id is created on rank 0 (mgx_comm_unique_id) and
 * handed to the other ranks by the launcher (bench.py broadcasts it with
 * torch.distributed); every rank then calls mgx_comm_init.  All communication runs on the
 * context's comm stream; mgx_comm_* functions that move planes take care of the
 * event ordering against the compute stream.
 */
#define MGX_UNIQUE_ID_BYTES 128
int mgx_comm_unique_id(void* host_id_bytes);
int mgx_comm_init(mgx_ctx* ctx, const void* host_id_bytes, int rank, int nranks);
/* rehearsal of ONE rank of a larger job on a single GPU (timing only, results are meaningless): a one-rank RCCL
 * communicator whose context reports (virtual_rank, virtual_nranks); every message keeps its size and its place in the
 * schedule but travels from this rank to itself */
int mgx_comm_init_rehearsal(mgx_ctx* ctx, const void* host_id_bytes, int virtual_rank, int virtual_nranks);
int mgx_comm_destroy(mgx_ctx* ctx);
int mgx_comm_rank(const mgx_ctx* ctx, int* rank, int* nranks);
/* 1 when the context's collectives can be captured into a HIP graph (an RCCL communicator, or a single rank without
 * one); 0 with the in-process test transport, whose exchanges are coordinated on the host */
int mgx_comm_capturable(const mgx_ctx* ctx);
/* the number of ranks the communicator itself reports (ncclCommCount) and the RCCL version in use (ncclGetVersion) */
int mgx_comm_info(const mgx_ctx* ctx, int* ranks_seen, int* rccl_version);
/* Test transport: `nranks` host threads of one process, one context each, all on the same device, so that the
 * slab-decomposed cycle AND the event ordering of its overlap schedule can be verified on a single-GPU box.  Same
 * stream semantics as RCCL: an exchange only enqueues device-to-device copies on the comm stream, ordered against
 * the peers by cross-context events; nothing is synchronised by the host; the compute stream sees the result only
 * through mgx_comm_wait.  Every rank's thread must take part in every exchange (as with RCCL).
 * _set_test_hooks: delay_us > 0 starts every transfer that late on the receiving comm stream (a missing wait then
 * reads stale ghosts for certain); drop_waits != 0 turns mgx_comm_wait into a no-op for the group's contexts (fault
 * injection: the tests must notice).  The hooks exist on this test transport only. */
typedef struct mgx_local_group mgx_local_group;
int mgx_local_group_create(int nranks, mgx_local_group** out);
int mgx_local_group_destroy(mgx_local_group* group);
int mgx_local_group_set_test_hooks(mgx_local_group* group, int delay_us, int drop_waits);
int mgx_comm_init_local(mgx_ctx* ctx, mgx_local_group* group, int rank);
/* Exchange ghost planes with the z-neighbours on a non-periodic chain of ranks (rank-1 = "lower",
 * rank+1 = "upper").  Counts are in elements of elem_bytes (4 or 8) and must match what the
 * neighbour passes for the opposite direction.  Buffers towards a missing neighbour are ignored.
 * Asynchronous on the comm stream: it first waits for everything enqueued so far on the compute
 * stream; mgx_comm_wait makes the compute stream wait for the exchange. */
int mgx_comm_halo_exchange(mgx_ctx* ctx, const void* send_to_lower, size_t count_to_lower, void* recv_from_lower,
                           size_t count_from_lower, const void* send_to_upper, size_t count_to_upper,
                           void* recv_from_upper, size_t count_from_upper, int elem_bytes);
int mgx_comm_wait(mgx_ctx* ctx);
/* Half planes for the ghost exchange behind a colour pass: only the half-rows that hold `colour` (even-x half where
 * colour + y + z is even, z = the plane's GLOBAL index) are packed into a staging array of mgx3dxs_halfplane_elems
 * elements (compute stream), exchanged with mgx_comm_halo_exchange, and unpacked into the ghost plane on the stream the
 * receive was enqueued on (mgx_comm_wait covers it).  Two planes per call (either may be NULL). */
size_t mgx3dxs_halfplane_elems_f32(int sx, int sy);
size_t mgx3dxs_halfplane_elems_f64(int sx, int sy);
int mgx3dxs_halo_pack_f32(mgx_ctx* ctx, const float* plane_a, int z_a, float* stage_a, const float* plane_b, int z_b, float* stage_b, int sx,
                          int sy, int colour);
int mgx3dxs_halo_pack_f64(mgx_ctx* ctx, const double* plane_a, int z_a, double* stage_a, const double* plane_b, int z_b, double* stage_b,
                          int sx, int sy, int colour);
int mgx3dxs_halo_unpack_f32(mgx_ctx* ctx, const float* stage_a, float* plane_a, int z_a, const float* stage_b, float* plane_b, int z_b, int sx,
                            int sy, int colour);
int mgx3dxs_halo_unpack_f64(mgx_ctx* ctx, const double* stage_a, double* plane_a, int z_a, const double* stage_b, double* plane_b, int z_b,
                            int sx, int sy, int colour);
/* on != 0: from now on collectives are enqueued on the COMPUTE stream, in order with the kernels (no overlap, no
 * cross-stream events; mgx_comm_wait is then a no-op); 0: back to the comm stream.  Every rank switches at the same
 * points of its schedule.  The slab driver switches per level (mgDistMultiGrid3D_*: inline_bytes). */
int mgx_comm_set_inline(mgx_ctx* ctx, int on);
/* all-gather `count` reals per rank (agglomeration of a coarse level) and all-reduce (sum, in place) of `count`
 * doubles (residual norm; every rank receives the same bits).  Both enqueue on the comm stream with the same
 * ordering rules, on both transports. */
int mgx_comm_allgather(mgx_ctx* ctx, const void* send, void* recv, size_t count, int elem_bytes);
int mgx_comm_allreduce_sum_f64(mgx_ctx* ctx, double* dev_inout, size_t count);
/* RCCL plumbing check usable on a single-GPU box: this rank sends `count` doubles to itself with a grouped
 * ncclSend/ncclRecv on the comm stream (the exact call pattern of the halo exchange). */
int mgx_comm_selftest(mgx_ctx* ctx, const double* dev_src, double* dev_dst, size_t count);

#ifdef __cplusplus
}
#endif
#endif /* MGX_H */

/* mg_multigrid.h -- C host layer: the reference's Grid{1,2,3}D / MultiGrid{1,2,3}D classes as C
 * structs + functions, driving the HIP kernels through the thin C-ABI of mgx.h.
 *
 * Mirrors, member for member, the public surface of the NOCUDA_TESI reference:
 *   class Grid3D        N3/Grid3D.h:4-38        -> struct mgGrid3D_<r>
 *   class MultiGrid3D   N3/MultiGrid3D.h:6-33   -> struct mgMultiGrid3D_<r> + mgMultiGrid3D_<r>_<Method>
 *   class Grid2D        N2/Grid2D.h:4-33        -> struct mgGrid2D_<r>
 *   class MultiGrid2D   N2/MultiGrid2D.h:6-37   -> struct mgMultiGrid2D_<r> + ...
 *   class Grid1D        N1/Grid1D.h:4-26        -> struct mgGrid1D_<r>      (CPU only: "plumbing")
 *   class MultiGrid1D   N1/MultiGrid1D.h:6-31   -> struct mgMultiGrid1D_<r> (CPU only)
 * with <r> = f32 (the reference's own type) or f64 (BASELINE.json's GPU configs).
 *
 * Same names, same argument meaning, same level-count rule (numGrids = (int)log2(minSize-1),
 * N3/MultiGrid3D.cpp:33-34; `numGrids` stays a public mutable field that may be lowered
 * after construction), same cycle control flow (VCycle N3/MultiGrid3D.cpp:623-647,
 * FullMultiGridVCycle :569-585).  Differences, all at the boundary:
 *   - every function returns an int status (mgx_status) instead of asserting/aborting;
 *   - d_v / d_f are device arrays; h_v / h_f are host mirrors (always the reference layout:
 *     dense, x fastest) filled by *_download_* and pushed by *_upload_*.  The 3D hierarchy keeps
 *     its device arrays in the x-split layout of mgx.h by default (`layout` = 1; rows
 *     de-interleaved by x parity so that each colour pass streams contiguous half-rows); raw
 *     device pointers handed to Restrict/Interpolate/... must use the hierarchy's layout;
 *   - residual / error scratch is owned by the level and preallocated (the reference mallocs
 *     both inside every VCycle call and never frees them, N3/MultiGrid3D.cpp:629,638);
 *   - `residual_mode` selects REF_COMPAT (default; reproduces the 3D residual sign quirk,
 *     N3/MultiGrid3D.cpp:723) or CORRECT;
 *   - `fuse` (default 1) runs CalculateResidual+Restrict and Interpolate+ApplyCorrection as
 *     one kernel each; results are bit-identical to the unfused sequence.
 * The 1D classes run entirely on the host in C (BASELINE.json configs[0]: CPU path, no GPU).
 */
#ifndef MG_MULTIGRID_H
#define MG_MULTIGRID_H

#include "mgx.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Which planes of a level one rank of a z-slab decomposition holds (global plane indices).  Pure host
 * arithmetic; also used by the CPU emulation test of the decomposition. */
#define MG_DEEP_GHOSTS 6      /* ghost planes on either side of a slab of >= MG_DEEP_MIN_PLANES planes */
#define MG_DEEP_MIN_PLANES 8
typedef struct mgSlabPlan {
    int zlo, zhi;   /* owned planes [zlo, zhi); the last rank also owns the boundary plane sizeZ-1 */
    int glo, ghi;   /* ghost planes below (ranks > 0) and above (ranks < P-1): MG_DEEP_GHOSTS each on slabs of at least */
                    /* MG_DEEP_MIN_PLANES planes, otherwise 2 below and 1 above */
    int zoff, nzl;  /* local array = global planes [zoff, zoff + nzl), zoff = zlo - glo */
    int ubeg, uend; /* planes the smoother updates: owned and interior, [max(zlo,1), min(zhi, sizeZ-1)) */
} mgSlabPlan;
/* number of leading levels that stay distributed: every rank owns an even number >= min_planes of planes */
int mg_dist_num_levels(int sizeZ_finest, int nranks, int numGrids, int min_planes);
int mg_dist_num_levels_single(int sizeZ_finest, int numGrids, int min_planes);
int mg_slab_plan(int sizeZ_level, int rank, int nranks, mgSlabPlan* out);

#define MG_MAX_LEVELS 32 /* (int)log2(size-1) of any int size */
#define MG_NORM_HISTORY 255 /* residual-norm history entries a distributed hierarchy keeps on the device */
#define MG_DECLARE(R, real)                                                                              \
    /* ------------------------------------------------------------------ 3D ------ */                  \
    typedef struct mgGrid3D_##R {                                                                        \
        real* h_v; /* host mirror of the approximate solution (NULL until first download/upload) */     \
        real* h_f; /* host mirror of the right-hand side */                                              \
        real* d_v; /* device: approximate solution */                                                    \
        real* d_f; /* device: right-hand side */                                                         \
        real* d_r; /* device scratch: residual (unfused path) */                                         \
        real* d_e; /* device scratch: interpolated error (unfused path) */                               \
        int sizeX, sizeY, sizeZ;                                                                         \
        int sizeXYZ[3];                                                                                  \
        real h_x, h_y, h_z;                                                                              \
        real x_a, x_b, y_a, y_b, z_a, z_b;                                                               \
    } mgGrid3D_##R;                                                                                      \
    typedef struct mgMultiGrid3D_##R {                                                                   \
        mgGrid3D_##R** grids3D;                                                                          \
        int numGrids;    /* public and mutable like the reference's; 1 <= numGrids <= maxGrids */        \
        int maxGrids;    /* levels allocated = (int)log2(minSize-1) */                                   \
        mgx_ctx* ctx;                                                                                    \
        int residual_mode; /* mgx_residual_mode */                                                       \
        int fuse;                                                                                        \
        int layout; /* device layout of d_v/d_f/d_r/d_e: 0 = reference layout, 1 = x-split (mgx.h) */    \
        int smoother; /* 0 = red-black Gauss-Seidel (the reference), 1 = weighted Jacobi (addition) */   \
        real omega;   /* Jacobi weight, default 2/3 */                                                   \
        int use_graph; /* 1: VCycle(gridID, v1, v2) is captured into a HIP graph on first use and     */ \
                       /* replayed afterwards (re-captured when its arguments or the fields above     */ \
                       /* change; context parameters are frozen at capture time).  Default 0.         */ \
        int capturing;                                                                                   \
        void* graph_exec[MG_MAX_LEVELS];                                                                 \
        long long graph_key[MG_MAX_LEVELS];                                                              \
        /* internal: 1 when the boundary entries of level l's d_f are known to be 0 (left so by the     */ \
        /* previous cycle's residual+restrict); cleared by InitF / upload_f / Restrict / setToValue.    */ \
        /* Those entries are never read by any operator; the flag only saves re-zeroing them.           */ \
        unsigned char f_rim_zero[MG_MAX_LEVELS];                                                         \
        /* internal: 1 when the boundary (and pad) entries of level l's d_v are known to be 0: the coarse */ \
        /* error then starts a cycle without a zero fill (relax_from_zero).  Cleared by upload_v and by  */ \
        /* setToValue(d_v, value != 0, true).                                                            */ \
        unsigned char v_rim_zero[MG_MAX_LEVELS];                                                         \
        /* internal: 1 when the boundary entries of level l's d_e equal those of its d_v: d_e is the      */ \
        /* ping-pong partner of the one-launch red+black sweeps (mgx3dxs_relax_pp), which never write     */ \
        /* boundary points.  Cleared by everything that may write either array's boundary.  External      */ \
        /* writers of d_v / d_e through the raw device pointers must clear it (and v_rim_zero) themselves. */ \
        unsigned char e_rim_valid[MG_MAX_LEVELS];                                                        \
    } mgMultiGrid3D_##R;                                                                                 \
    int mgMultiGrid3D_##R##_create(mgx_ctx* ctx, const int finestGridSizeXYZ[3], const real range[6],     \
                                   mgMultiGrid3D_##R** out);                                             \
    int mgMultiGrid3D_##R##_create_layout(mgx_ctx* ctx, const int finestGridSizeXYZ[3],                  \
                                          const real range[6], int layout, mgMultiGrid3D_##R** out);     \
    void mgMultiGrid3D_##R##_destroy(mgMultiGrid3D_##R* mg);                                             \
    int mgMultiGrid3D_##R##_InitV(mgMultiGrid3D_##R* mg, int gridID);                                    \
    int mgMultiGrid3D_##R##_InitF(mgMultiGrid3D_##R* mg, int gridID);                                    \
    int mgMultiGrid3D_##R##_Restrict(mgMultiGrid3D_##R* mg, const real* fine, const int fsizeXYZ[3],     \
                                     real* coarse, const int csizeXYZ[3]);                               \
    int mgMultiGrid3D_##R##_Interpolate(mgMultiGrid3D_##R* mg, real* fine, const int fsizeXYZ[3],        \
                                        const real* coarse, const int csizeXYZ[3]);                      \
    int mgMultiGrid3D_##R##_Relax(mgMultiGrid3D_##R* mg, mgGrid3D_##R* curGrid, int ncycles);            \
    int mgMultiGrid3D_##R##_setToValue(mgMultiGrid3D_##R* mg, real* grid, const int sizeXYZ[3],          \
                                       real value, int modifyBoundaries);                                \
    int mgMultiGrid3D_##R##_CalculateResidual(mgMultiGrid3D_##R* mg, mgGrid3D_##R* fine,                 \
                                              real** residual);                                          \
    int mgMultiGrid3D_##R##_ApplyCorrection(mgMultiGrid3D_##R* mg, real* fine, const int fsizeXYZ[3],    \
                                            const real* error, const int esizeXYZ[3]);                   \
    int mgMultiGrid3D_##R##_VCycle(mgMultiGrid3D_##R* mg, int gridID, int v1, int v2);                   \
    int mgMultiGrid3D_##R##_FullMultiGridVCycle(mgMultiGrid3D_##R* mg, int gridID, int v0, int v1,       \
                                                int v2);                                                 \
    int mgMultiGrid3D_##R##_upload_v(mgMultiGrid3D_##R* mg, int gridID, const real* host);               \
    int mgMultiGrid3D_##R##_upload_f(mgMultiGrid3D_##R* mg, int gridID, const real* host);               \
    int mgMultiGrid3D_##R##_download_v(mgMultiGrid3D_##R* mg, int gridID, real* host);                   \
    int mgMultiGrid3D_##R##_download_f(mgMultiGrid3D_##R* mg, int gridID, real* host);                   \
    int mgMultiGrid3D_##R##_download_residual(mgMultiGrid3D_##R* mg, int gridID, real* host);            \
    int mgMultiGrid3D_##R##_ResidualNorm(mgMultiGrid3D_##R* mg, int gridID, double* l2);                 \
    /* PrintDiff (N3/MultiGrid3D.cpp:760-764 -> Grid3D::PrintDiff) as numbers: mean |diff|, max |diff| and  */ \
    /* relative L2 of diff = sin(pi x) sin(pi y) sin(pi z) - v over all points of level gridID              */ \
    int mgMultiGrid3D_##R##_DiffStats(mgMultiGrid3D_##R* mg, int gridID, double* mean_abs,               \
                                      double* max_abs, double* rel_l2);                                  \
    /* solve(grid, rhs, nlevels): host arrays in the reference layout; grid = initial guess incl.     */ \
    /* boundary values on input, solution on output; nlevels = 0 -> reference rule; ncycles V(v1,v2)  */ \
    /* cycles from the given guess, or one FullMultiGridVCycle(v0,v1,v2) when fmg != 0.               */ \
    /* rhs == NULL: the reference's own right-hand side (Grid3D::InitF) is built on the device, nothing is  */ \
    /* uploaded for it.  grid_is_zero != 0 (mg3d_solve_from_zero): the guess is the reference's InitV state */ \
    /* (all zeros); `grid` is output only and no guess is uploaded either -- with both, the call is the     */ \
    /* reference's whole driver (N3/Poisson3DSolver.cpp:6-51) with ONE transfer, the result.               */ \
    int mg3d_solve_##R(mgx_ctx* ctx, real* grid, const real* rhs, const int sizeXYZ[3],                  \
                       const real range[6], int nlevels, int fmg, int v0, int v1, int v2, int ncycles,   \
                       int residual_mode);                                                               \
    int mg3d_solve_from_zero_##R(mgx_ctx* ctx, real* grid_out, const real* rhs, const int sizeXYZ[3],    \
                                 const real range[6], int nlevels, int fmg, int v0, int v1, int v2,      \
                                 int ncycles, int residual_mode);                                        \
    /* ---- z-slab decomposed 3D V-cycle (one process per GPU; csrc/host/mg_dist3d.inc) ---- */         \
    typedef struct mgSlab3D_##R {                                                                        \
        real* d_v; /* local planes [zoff, zoff+nzl) of the level, x-split layout, ghosts included */     \
        real* d_f;                                                                                       \
        int sizeXYZ[3]; /* GLOBAL sizes of the level */                                                  \
        mgSlabPlan plan;                                                                                 \
        real h_x, h_y, h_z;                                                                              \
        real x_a, y_a, z_a;                                                                              \
        int level;      /* its index in mgDistMultiGrid3D::slabs */                                      \
    } mgSlab3D_##R;                                                                                      \
    typedef struct mgDistMultiGrid3D_##R {                                                               \
        mgSlab3D_##R** slabs;      /* levels 0 .. numDist-1 */                                           \
        int numDist;               /* number of distributed levels (mg_dist_num_levels) */               \
        int numGrids;              /* total levels of the cycle, public and mutable like the reference */\
        int maxGrids;                                                                                    \
        mgMultiGrid3D_##R* tail;   /* replicated hierarchy for levels >= numDist */                      \
        mgx_ctx* ctx;                                                                                    \
        int rank, nranks;                                                                                \
        int residual_mode;                                                                               \
        real* d_share;             /* staging for the agglomeration all-gather */                        \
        real* d_bplane;            /* FMG: staging for the top boundary plane of the replicated f */     \
        double* d_norm;            /* device: [0] scratch of ResidualNorm, [1 ...] squared-norm history */ \
        int norm_count;            /* entries recorded by ResidualNormRecord (<= MG_NORM_HISTORY) */     \
        long long inline_bytes;    /* levels whose slab of v is at most this large exchange inline on the */ \
                                   /* compute stream, one launch per pass (0: always overlapped); public  */ \
        /* internal: 1 while the boundary entries (and ghost-plane rims) of distributed level l's v are  */ \
        /* known to be 0: the coarse error then starts a cycle without a zero fill.  Set by create and  */ \
        /* zero_v, cleared by upload_v.                                                                  */ \
        unsigned char v_rim_zero[MG_MAX_LEVELS];                                                         \
        /* use_graph != 0: VCycle(0, v1, v2) is captured into a HIP graph (both streams, the RCCL calls */ \
        /* included) and replayed, as mgMultiGrid3D does -- opt-in: it needs collectives that can be    */ \
        /* captured (RCCL or a single rank; mgx_comm_capturable), and RCCL under capture between        */ \
        /* different GPUs has not been run anywhere yet.  A failed capture is an error, not a fallback. */ \
        int use_graph;                                                                                   \
        void* graph_exec;                                                                                \
        long long graph_key;                                                                             \
        int graph_warm;            /* the first cycle runs eagerly: lazy allocations cannot be captured */ \
        /* pack_halos != 0: the ghost exchange behind a colour pass carries only the half-rows of the    */ \
        /* colour the pass changed (d_stage: 2 send + 2 receive arrays of stage_half elements); 0        */ \
        /* (default): whole planes -- at 1025^3 on 8 ranks a plane's transfer hides behind the interior  */ \
        /* launch and the pack / unpack launches only add latency (rehearsal: 4.52 against 4.36 ms)      */ \
        int pack_halos;                                                                                  \
        real* d_stage;                                                                                   \
        size_t stage_half;                                                                               \
        /* Communication-avoiding schedule (public knob): levels whose slabs have at least ca_min_planes */ \
        /* planes (default 16; 0 = never) exchange ghost planes of v ONCE PER Relax CALL (4 deep for two */ \
        /* sweeps) instead of once per colour pass: the first ghost planes are relaxed redundantly, the */ \
        /* owner's expression on the owner's inputs (same bits), the valid region shrinking by one plane */ \
        /* per pass.  Thinner levels keep one exchange per colour pass.                                  */ \
        int ca_min_planes;                                                                               \
        /* internal: how many ghost planes of v / f on either side of a level's slab hold current values */ \
        /* (the same number on every rank; >= MG_DEEP_GHOSTS: all of them); consumers ask for what they */ \
        /* read and an exchange happens only when that is not there                                     */ \
        signed char gv[MG_MAX_LEVELS], gf[MG_MAX_LEVELS];                                                \
        int comm_pending;          /* an exchange is in flight on the comm stream; nothing may touch ghost */ \
                                   /* planes before mgx_comm_wait                                       */ \
        /* halo exchanges + collectives enqueued since creation (bench.py: exchanges per cycle)          */ \
        long long n_exchanges;                                                                           \
    } mgDistMultiGrid3D_##R;                                                                             \
    int mgDistMultiGrid3D_##R##_create(mgx_ctx* ctx, const int finestGridSizeXYZ[3], const real range[6], \
                                       int min_planes, mgDistMultiGrid3D_##R** out);                     \
    void mgDistMultiGrid3D_##R##_destroy(mgDistMultiGrid3D_##R* mg);                                     \
    int mgDistMultiGrid3D_##R##_InitF(mgDistMultiGrid3D_##R* mg, int gridID);                            \
    int mgDistMultiGrid3D_##R##_Relax(mgDistMultiGrid3D_##R* mg, int gridID, int ncycles);               \
    int mgDistMultiGrid3D_##R##_VCycle(mgDistMultiGrid3D_##R* mg, int gridID, int v1, int v2);           \
    /* FullMultiGridVCycle (N3/MultiGrid3D.cpp:569-585) on slabs                                        */ \
    int mgDistMultiGrid3D_##R##_FullMultiGridVCycle(mgDistMultiGrid3D_##R* mg, int gridID, int v0,       \
                                                    int v1, int v2);                                     \
    /* l2 norm of the residual of a distributed level over the WHOLE grid (ADDITION: the reference has  */ \
    /* no norm; parity unpinned): every rank reduces the squared residual of the planes  */ \
    /* it owns on the device (wavefront-wide shuffles, fixed order), the partial sums are all-reduced   */ \
    /* over RCCL (one double) and every rank returns the same value.  Blocking.  Uses residual_mode.    */ \
    int mgDistMultiGrid3D_##R##_ResidualNorm(mgDistMultiGrid3D_##R* mg, int gridID, double* l2);         \
    /* the same without a host round trip: the squared norm is appended to a history kept on the device */ \
    /* (at most MG_NORM_HISTORY entries, e.g. one per cycle); _History downloads sqrt of the entries.    */ \
    int mgDistMultiGrid3D_##R##_ResidualNormRecord(mgDistMultiGrid3D_##R* mg, int gridID);               \
    int mgDistMultiGrid3D_##R##_ResidualNormHistory(mgDistMultiGrid3D_##R* mg, double* host_l2,          \
                                                    int capacity, int* count);                          \
    int mgDistMultiGrid3D_##R##_zero_v(mgDistMultiGrid3D_##R* mg, int gridID);                           \
    int mgDistMultiGrid3D_##R##_upload_v(mgDistMultiGrid3D_##R* mg, int gridID, const real* host_full);  \
    int mgDistMultiGrid3D_##R##_upload_f(mgDistMultiGrid3D_##R* mg, int gridID, const real* host_full);  \
    int mgDistMultiGrid3D_##R##_download_v(mgDistMultiGrid3D_##R* mg, int gridID, real* host_full);      \
    /* ------------------------------------------------------------------ 2D ------ */                  \
    typedef struct mgGrid2D_##R {                                                                        \
        real* h_v;                                                                                       \
        real* h_f;                                                                                       \
        real* d_v;                                                                                       \
        real* d_f;                                                                                       \
        real* d_r;                                                                                       \
        real* d_e;                                                                                       \
        int sizeX, sizeY;                                                                                \
        int sizeXY[2];                                                                                   \
        real h_x, h_y;                                                                                   \
        real x_a, x_b, y_a, y_b;                                                                         \
    } mgGrid2D_##R;                                                                                      \
    typedef struct mgMultiGrid2D_##R {                                                                   \
        mgGrid2D_##R** grids2D;                                                                          \
        int numGrids;                                                                                    \
        int maxGrids;                                                                                    \
        real matrixA[4];                                                                                 \
        int sizeA;                                                                                       \
        int alfa;                                                                                        \
        mgx_ctx* ctx;                                                                                    \
        int fuse; /* 2 (default): VCycle on the cache-resident kernels (one launch per level and direction, */ \
                  /* one for all levels <= 65^2); 1: CalculateResidual+Restrict and Interpolate+             */ \
                  /* ApplyCorrection fused, one launch per colour pass; 0: one launch per reference call.    */ \
                  /* Results are bit-identical.                                                              */ \
        int smoother; /* 0 = red-black Gauss-Seidel (the reference), 1 = weighted Jacobi (addition) */   \
        real omega;                                                                                      \
        int use_graph; /* as in mgMultiGrid3D: the 1025^2 cycle is launch-bound */                       \
        int capturing;                                                                                   \
        void* graph_exec[MG_MAX_LEVELS];                                                                 \
        long long graph_key[MG_MAX_LEVELS];                                                              \
    } mgMultiGrid2D_##R;                                                                                 \
    int mgMultiGrid2D_##R##_create(mgx_ctx* ctx, const int finestGridSizeXY[2], const real range[4],     \
                                   const real* A, int A_size, int alfa, mgMultiGrid2D_##R** out);        \
    void mgMultiGrid2D_##R##_destroy(mgMultiGrid2D_##R* mg);                                             \
    int mgMultiGrid2D_##R##_InitV(mgMultiGrid2D_##R* mg, int gridID);                                    \
    int mgMultiGrid2D_##R##_InitF(mgMultiGrid2D_##R* mg, int gridID);                                    \
    int mgMultiGrid2D_##R##_Restrict(mgMultiGrid2D_##R* mg, const real* fine, const int fsizeXY[2],      \
                                     real* coarse, const int csizeXY[2]);                                \
    int mgMultiGrid2D_##R##_Interpolate(mgMultiGrid2D_##R* mg, real* fine, const int fsizeXY[2],         \
                                        const real* coarse, const int csizeXY[2]);                       \
    int mgMultiGrid2D_##R##_Relax(mgMultiGrid2D_##R* mg, mgGrid2D_##R* curGrid, int ncycles);            \
    int mgMultiGrid2D_##R##_setToValue(mgMultiGrid2D_##R* mg, real* grid, const int sizeXY[2],           \
                                       real value, int modifyBoundaries);                                \
    int mgMultiGrid2D_##R##_CalculateResidual(mgMultiGrid2D_##R* mg, mgGrid2D_##R* fine,                 \
                                              real** residual);                                          \
    int mgMultiGrid2D_##R##_ApplyCorrection(mgMultiGrid2D_##R* mg, real* fine, const int fsizeXY[2],     \
                                            const real* error, const int esizeXY[2]);                    \
    int mgMultiGrid2D_##R##_VCycle(mgMultiGrid2D_##R* mg, int gridID, int v1, int v2);                   \
    int mgMultiGrid2D_##R##_FullMultiGridVCycle(mgMultiGrid2D_##R* mg, int gridID, int v0, int v1,       \
                                                int v2);                                                 \
    int mgMultiGrid2D_##R##_upload_v(mgMultiGrid2D_##R* mg, int gridID, const real* host);               \
    int mgMultiGrid2D_##R##_upload_f(mgMultiGrid2D_##R* mg, int gridID, const real* host);               \
    int mgMultiGrid2D_##R##_download_v(mgMultiGrid2D_##R* mg, int gridID, real* host);                   \
    int mgMultiGrid2D_##R##_download_f(mgMultiGrid2D_##R* mg, int gridID, real* host);                   \
    /* mean over the interior of |v - (2x^2-4xy+2y^2)| on the finest level (thesis Fig. 4.3 metric,     */ \
    /* CUDA_TESI/CUDA Lyapunov 2D/Grid2D.cu:123-154)                                                    */ \
    int mgMultiGrid2D_##R##_MeanAbsoluteError(mgMultiGrid2D_##R* mg, int gridID, double* mean);          \
    int mg2d_solve_##R(mgx_ctx* ctx, real* grid, const real* rhs, const int sizeXY[2],                   \
                       const real range[4], const real A[4], int alfa, int nlevels, int fmg, int v0,     \
                       int v1, int v2, int ncycles);                                                     \
    /* ------------------------------------------------------------------ 1D (host only) */             \
    typedef struct mgGrid1D_##R {                                                                        \
        real* h_v;                                                                                       \
        real* h_f;                                                                                       \
        int sizeX;                                                                                       \
        real h_x;                                                                                        \
        real x_a, x_b;                                                                                   \
    } mgGrid1D_##R;                                                                                      \
    typedef struct mgMultiGrid1D_##R {                                                                   \
        mgGrid1D_##R** grids1D;                                                                          \
        int numGrids;                                                                                    \
        int maxGrids;                                                                                    \
    } mgMultiGrid1D_##R;                                                                                 \
    int mgMultiGrid1D_##R##_create(int finestGridSize, const real range[2], mgMultiGrid1D_##R** out);    \
    void mgMultiGrid1D_##R##_destroy(mgMultiGrid1D_##R* mg);                                             \
    int mgMultiGrid1D_##R##_Restrict(mgMultiGrid1D_##R* mg, const real* fine, int fsize, real* coarse,   \
                                     int csize);                                                         \
    int mgMultiGrid1D_##R##_Interpolate(mgMultiGrid1D_##R* mg, real* fine, int fsize,                    \
                                        const real* coarse, int csize);                                  \
    int mgMultiGrid1D_##R##_Relax(mgMultiGrid1D_##R* mg, mgGrid1D_##R* curGrid, int ncycles);            \
    int mgMultiGrid1D_##R##_setToValue(mgMultiGrid1D_##R* mg, real* grid, int sizeX, real value,         \
                                       int modifyBoundaries);                                            \
    int mgMultiGrid1D_##R##_CalculateResidual(mgMultiGrid1D_##R* mg, mgGrid1D_##R* fine,                 \
                                              real* residual);                                           \
    int mgMultiGrid1D_##R##_ApplyCorrection(mgMultiGrid1D_##R* mg, real* fine, int fineSize,             \
                                            const real* error, int errorSize);                           \
    int mgMultiGrid1D_##R##_VCycle(mgMultiGrid1D_##R* mg, int gridID, int v1, int v2);                   \
    int mgMultiGrid1D_##R##_FullMultiGridVCycle(mgMultiGrid1D_##R* mg, int gridID, int v0, int v1,       \
                                                int v2);                                                 \
    int mg1d_solve_##R(real* grid, const real* rhs, int sizeX, const real range[2], int nlevels,         \
                       int fmg, int v0, int v1, int v2, int ncycles);

MG_DECLARE(f32, float)
MG_DECLARE(f64, double)

/* numGrids = (int)log2(minSize - 1)            N3/MultiGrid3D.cpp:33-34 */
int mg_num_grids(int minSize);
/* coarse size = ((size-1)/2)+1                 N3/MultiGrid3D.cpp:40-42 */
int mg_coarse_size(int size);

#ifdef __cplusplus
}
#endif
#endif /* MG_MULTIGRID_H */

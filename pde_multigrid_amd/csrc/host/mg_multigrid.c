/* mg_multigrid.c -- C host layer of include/mg_multigrid.h: stamps the 1D/2D/3D hierarchy and
 * cycle code for float and double.  Compiled with gcc -std=c11 -ffp-contract=off. */
#include "mg_common.h"

/* numGrids = log2(minSize - 1) truncated to int                   N3/MultiGrid3D.cpp:33-34 */
int mg_num_grids(int minSize) { return (int)log2((double)(minSize - 1)); }
int mg_coarse_size(int size) { return ((size - 1) / 2) + 1; } /* N3/MultiGrid3D.cpp:40-42 */

/* ---- z-slab plan: pure host arithmetic shared by the product (mg_dist3d.inc) and by the CPU/gloo
 * emulation test (tests/test_dist_gloo.py) ------------------------------------------------------------ */
int mg_dist_num_levels(int sizeZ_finest, int nranks, int numGrids, int min_planes) {
    if (nranks <= 1 || numGrids <= 0) return nranks <= 1 && numGrids > 0 ? mg_dist_num_levels_single(sizeZ_finest, numGrids, min_planes) : 0;
    if (min_planes < 2) min_planes = 2;
    int levels = 0, N = sizeZ_finest - 1;
    /* level l is distributed while every rank owns N/P >= min_planes planes, N/P even (so that the owner
     * of coarse plane k owns the fine planes 2k and 2k+1) */
    while (levels < numGrids && N % nranks == 0 && (N / nranks) >= min_planes && (N / nranks) % 2 == 0) {
        levels++;
        N /= 2;
    }
    return levels;
}

int mg_dist_num_levels_single(int sizeZ_finest, int numGrids, int min_planes) {
    /* one rank: the "distributed" levels are whole grids handled by the slab code path (no communication);
     * used by tests to exercise that path on one GPU.  Same rule as above with P = 1. */
    if (min_planes < 2) min_planes = 2;
    int levels = 0, N = sizeZ_finest - 1;
    while (levels < numGrids && N >= min_planes && N % 2 == 0) {
        levels++;
        N /= 2;
    }
    return levels;
}

int mg_slab_plan(int sizeZ_level, int rank, int nranks, mgSlabPlan* out) {
    if (!out || nranks < 1 || rank < 0 || rank >= nranks || sizeZ_level < 3) return mg_fail(MGX_ERR_INVALID, "mg_slab_plan: bad arguments");
    const int N = sizeZ_level - 1;
    if (N % nranks != 0) return mg_fail(MGX_ERR_SIZE, "mg_slab_plan: %d cells do not divide over %d ranks", N, nranks);
    out->zlo = rank * (N / nranks);
    out->zhi = (rank + 1) * (N / nranks) + (rank == nranks - 1 ? 1 : 0); /* the last rank also owns boundary plane N */
    /* ghost planes: slabs of at least MG_DEEP_MIN_PLANES planes carry MG_DEEP_GHOSTS on either side (the communication-avoiding
     * schedule of mg_dist3d.inc relaxes the first ghost planes redundantly and exchanges once per Relax call); thinner slabs
     * the 2 below / 1 above of the exchange-per-colour-pass schedule.  The same on every rank of a level. */
    const int deep = (N / nranks) >= MG_DEEP_MIN_PLANES;
    out->glo = rank > 0 ? (deep ? MG_DEEP_GHOSTS : 2) : 0;
    out->ghi = rank < nranks - 1 ? (deep ? MG_DEEP_GHOSTS : 1) : 0;
    out->zoff = out->zlo - out->glo;
    out->nzl = (out->zhi - out->zlo) + out->glo + out->ghi;
    out->ubeg = out->zlo > 1 ? out->zlo : 1;     /* global planes the smoother updates: owned and interior */
    out->uend = out->zhi < N ? out->zhi : N;
    return MGX_OK;
}

#define REAL float
#define R f32
#define MG_EXP(x) expf(x) /* exp(float) resolves to the float overload in the reference */
#include "mg_multigrid3d.inc"
#include "mg_multigrid2d.inc"
#include "mg_multigrid1d.inc"
#include "mg_dist3d.inc"
#undef REAL
#undef R
#undef MG_EXP

#define REAL double
#define R f64
#define MG_EXP(x) exp(x)
#include "mg_multigrid3d.inc"
#include "mg_multigrid2d.inc"
#include "mg_multigrid1d.inc"
#include "mg_dist3d.inc"
#undef REAL
#undef R
#undef MG_EXP

/* mg_multigrid.c -- C host layer of include/mg_multigrid.h: stamps the 1D/2D/3D hierarchy and
 * cycle code for float and double.  Compiled with gcc -std=c11 -ffp-contract=off. */
#include "mg_common.h"

/* numGrids = log2(minSize - 1) truncated to int                   N3/MultiGrid3D.cpp:33-34 */
int mg_num_grids(int minSize) { return (int)log2((double)(minSize - 1)); }
int mg_coarse_size(int size) { return ((size - 1) / 2) + 1; } /* N3/MultiGrid3D.cpp:40-42 */

#define REAL float
#define R f32
#define MG_EXP(x) expf(x) /* exp(float) resolves to the float overload in the reference */
#include "mg_multigrid3d.inc"
#include "mg_multigrid2d.inc"
#include "mg_multigrid1d.inc"
#undef REAL
#undef R
#undef MG_EXP

#define REAL double
#define R f64
#define MG_EXP(x) exp(x)
#include "mg_multigrid3d.inc"
#include "mg_multigrid2d.inc"
#include "mg_multigrid1d.inc"
#undef REAL
#undef R
#undef MG_EXP

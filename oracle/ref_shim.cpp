// oracle/ref_shim.cpp -- TEST INFRASTRUCTURE, container-only.
//
// Thin extern "C" driver around the *unmodified* NOCUDA_TESI classes of the
// reference (MultiGrid{1,2,3}D / Grid{1,2,3}D).  It contains no reference
// code: it #includes the reference's class headers (found through -I by
// oracle/Makefile) and is linked against objects that the Makefile compiles
// from the sources where they lie under /root/reference.  The result is
// oracle/_ref/libmgref.so (git-ignored), used to
//   * pin oracle/mg_oracle.cpp (the CPU restatement) bit-for-bit, and
//   * generate the committed fixtures in tests/golden/ (oracle/gen_golden.py).
// It never travels as source of the reference and is never part of the
// product path.
//
// Every entry point builds a fresh reference hierarchy, overwrites the public
// h_v / h_f arrays of the levels it needs with caller data, calls the public
// member function, and copies the result out.  `nlevels > 0` overwrites the
// public `numGrids` field.
#include <chrono>
#include <cstring>
#include <cstdlib>
#include <cstdint>

#include "Grid1D.h"
#include "MultiGrid1D.h"
#include "Grid2D.h"
#include "MultiGrid2D.h"
#include "Grid3D.h"
#include "MultiGrid3D.h"

namespace {

inline size_t vol3(const int n[3]) { return (size_t)n[0] * n[1] * n[2]; }
inline size_t vol2(const int n[2]) { return (size_t)n[0] * n[1]; }

// The reference's InitA under-allocates matrixA (A_size floats, writes
// A_size^2); hand it a roomy heap block afterwards so later reads are sane.
struct MG2 {
    MultiGrid2D* mg;
    MG2(const int n[2], const float range[4], const float A[4], int alfa) {
        int nn[2] = {n[0], n[1]};
        float rr[4] = {range[0], range[1], range[2], range[3]};
        float* a = (float*)malloc(16 * sizeof(float));
        for (int i = 0; i < 4; i++) a[i] = A[i];
        mg = new MultiGrid2D(nn, rr, a, 2, alfa);
        // replace the 8-byte block the reference allocated by a 4-float one
        float* fixedA = (float*)malloc(16 * sizeof(float));
        for (int i = 0; i < 4; i++) fixedA[i] = A[i];
        mg->matrixA = fixedA;
        free(a);
    }
};

}  // namespace

extern "C" {

// ---------------------------------------------------------------- 3D ------
int ref3d_num_grids(const int n[3]) {
    int nn[3] = {n[0], n[1], n[2]};
    float rr[6] = {0, 1, 0, 1, 0, 1};
    // numGrids only depends on the sizes; use a tiny clone when large
    MultiGrid3D mg(nn, rr);
    return mg.numGrids;
}

// analytic InitV/InitF of level `level` (interior of v is uninitialised in the
// reference: it is reported here as 0).
void ref3d_init(const int n[3], const float range[6], int level, float* v, float* f) {
    int nn[3] = {n[0], n[1], n[2]};
    float rr[6];
    memcpy(rr, range, sizeof rr);
    MultiGrid3D mg(nn, rr);
    Grid3D* g = mg.grids3D[level];
    mg.setToValue(g->h_v, g->sizeXYZ, 0.0f, false);
    size_t N = vol3(g->sizeXYZ);
    memcpy(v, g->h_v, N * sizeof(float));
    memcpy(f, g->h_f, N * sizeof(float));
}

void ref3d_relax(const int n[3], const float range[6], float* v, const float* f, int ncycles) {
    int nn[3] = {n[0], n[1], n[2]};
    float rr[6];
    memcpy(rr, range, sizeof rr);
    MultiGrid3D mg(nn, rr);
    Grid3D* g = mg.grids3D[0];
    size_t N = vol3(n);
    memcpy(g->h_v, v, N * sizeof(float));
    memcpy(g->h_f, f, N * sizeof(float));
    mg.Relax(g, ncycles);
    memcpy(v, g->h_v, N * sizeof(float));
}

void ref3d_residual(const int n[3], const float range[6], const float* v, const float* f, float* r) {
    int nn[3] = {n[0], n[1], n[2]};
    float rr[6];
    memcpy(rr, range, sizeof rr);
    MultiGrid3D mg(nn, rr);
    Grid3D* g = mg.grids3D[0];
    size_t N = vol3(n);
    memcpy(g->h_v, v, N * sizeof(float));
    memcpy(g->h_f, f, N * sizeof(float));
    float* res = mg.CalculateResidual(g);
    memcpy(r, res, N * sizeof(float));
    free(res);
}

static void mini3(MultiGrid3D*& mg) {
    int nn[3] = {5, 5, 5};
    float rr[6] = {0, 1, 0, 1, 0, 1};
    mg = new MultiGrid3D(nn, rr);
}

void ref3d_restrict(const int fn[3], const float* fine, float* coarse) {
    MultiGrid3D* mg; mini3(mg);
    int f3[3] = {fn[0], fn[1], fn[2]};
    int c3[3] = {(fn[0] - 1) / 2 + 1, (fn[1] - 1) / 2 + 1, (fn[2] - 1) / 2 + 1};
    mg->Restrict(const_cast<float*>(fine), f3, coarse, c3);
}

void ref3d_interpolate(const int fn[3], float* fine, const float* coarse) {
    MultiGrid3D* mg; mini3(mg);
    int f3[3] = {fn[0], fn[1], fn[2]};
    int c3[3] = {(fn[0] - 1) / 2 + 1, (fn[1] - 1) / 2 + 1, (fn[2] - 1) / 2 + 1};
    mg->Interpolate(fine, f3, const_cast<float*>(coarse), c3);
}

void ref3d_apply_correction(const int n[3], float* fine, const float* err) {
    MultiGrid3D* mg; mini3(mg);
    int f3[3] = {n[0], n[1], n[2]};
    mg->ApplyCorrection(fine, f3, const_cast<float*>(err), f3);
}

void ref3d_set(const int n[3], float* grid, float value, int modify_boundaries) {
    MultiGrid3D* mg; mini3(mg);
    int f3[3] = {n[0], n[1], n[2]};
    mg->setToValue(grid, f3, value, modify_boundaries != 0);
}

// One cycle on a fresh hierarchy.  v_in / f_in (finest level) may be NULL:
// then the reference's analytic InitF is kept and v interior is set to 0
// (setToValue(v,0,false)), exactly the recipe of SURVEY.md section 8c.
// mode 0: `reps` x VCycle(0,v1,v2);  mode 1: FullMultiGridVCycle(0,v0,v1,v2).
void ref3d_cycle(const int n[3], const float range[6], int nlevels, int mode, int v0, int v1,
                 int v2, int reps, const float* v_in, const float* f_in, float* v_out) {
    int nn[3] = {n[0], n[1], n[2]};
    float rr[6];
    memcpy(rr, range, sizeof rr);
    MultiGrid3D mg(nn, rr);
    if (nlevels > 0) mg.numGrids = nlevels;
    Grid3D* g = mg.grids3D[0];
    size_t N = vol3(n);
    if (f_in) memcpy(g->h_f, f_in, N * sizeof(float));
    if (v_in) memcpy(g->h_v, v_in, N * sizeof(float));
    else mg.setToValue(g->h_v, g->sizeXYZ, 0.0f, false);
    if (mode == 0) {
        for (int i = 0; i < reps; i++) mg.VCycle(0, v1, v2);
    } else {
        // interior of coarse h_v is uninitialised in the reference and is
        // overwritten by Interpolate / setToValue before any read.
        mg.FullMultiGridVCycle(0, v0, v1, v2);
    }
    memcpy(v_out, g->h_v, N * sizeof(float));
}

// CPU-baseline timers (bench.py cpu_baseline, kind "reference"): the reference's own loops on its own analytic
// problem ([0,1]^3), construction and InitF outside the timed region, one thread.
double ref3d_time_vcycle(int n, int nlevels, int v1, int v2, int reps) {
    int nn[3] = {n, n, n};
    float rr[6] = {0, 1, 0, 1, 0, 1};
    MultiGrid3D mg(nn, rr);
    if (nlevels > 0) mg.numGrids = nlevels;
    Grid3D* g = mg.grids3D[0];
    mg.setToValue(g->h_v, g->sizeXYZ, 0.0f, false);
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < reps; i++) mg.VCycle(0, v1, v2);
    auto t1 = std::chrono::steady_clock::now();
    return std::chrono::duration<double>(t1 - t0).count();
}

double ref3d_time_relax(int n, int sweeps) {
    int nn[3] = {n, n, n};
    float rr[6] = {0, 1, 0, 1, 0, 1};
    MultiGrid3D mg(nn, rr);
    Grid3D* g = mg.grids3D[0];
    mg.setToValue(g->h_v, g->sizeXYZ, 0.0f, false);
    auto t0 = std::chrono::steady_clock::now();
    mg.Relax(g, sweeps);
    auto t1 = std::chrono::steady_clock::now();
    return std::chrono::duration<double>(t1 - t0).count();
}

// ---------------------------------------------------------------- 2D ------
void ref2d_init(const int n[2], const float range[4], const float A[4], int alfa, int level,
                float* v, float* f) {
    MG2 m(n, range, A, alfa);
    Grid2D* g = m.mg->grids2D[level];
    size_t N = vol2(g->sizeXY);
    memcpy(v, g->h_v, N * sizeof(float));
    memcpy(f, g->h_f, N * sizeof(float));
}

void ref2d_relax(const int n[2], const float range[4], const float A[4], int alfa, float* v,
                 const float* f, int ncycles) {
    MG2 m(n, range, A, alfa);
    Grid2D* g = m.mg->grids2D[0];
    size_t N = vol2(n);
    memcpy(g->h_v, v, N * sizeof(float));
    memcpy(g->h_f, f, N * sizeof(float));
    m.mg->Relax(g, ncycles);
    memcpy(v, g->h_v, N * sizeof(float));
}

void ref2d_residual(const int n[2], const float range[4], const float A[4], int alfa,
                    const float* v, const float* f, float* r) {
    MG2 m(n, range, A, alfa);
    Grid2D* g = m.mg->grids2D[0];
    size_t N = vol2(n);
    memcpy(g->h_v, v, N * sizeof(float));
    memcpy(g->h_f, f, N * sizeof(float));
    float* res = m.mg->CalculateResidual(g);
    memcpy(r, res, N * sizeof(float));
    free(res);
}

static MultiGrid2D* mini2() {
    int nn[2] = {5, 5};
    float rr[4] = {0, 1, 0, 1};
    float A[4] = {-1, -2, 0, -3};
    MG2 m(nn, rr, A, 2);
    return m.mg;
}

void ref2d_restrict(const int fn[2], const float* fine, float* coarse) {
    MultiGrid2D* mg = mini2();
    int f2[2] = {fn[0], fn[1]};
    int c2[2] = {(fn[0] - 1) / 2 + 1, (fn[1] - 1) / 2 + 1};
    mg->Restrict(const_cast<float*>(fine), f2, coarse, c2);
}

void ref2d_interpolate(const int fn[2], float* fine, const float* coarse) {
    MultiGrid2D* mg = mini2();
    int f2[2] = {fn[0], fn[1]};
    int c2[2] = {(fn[0] - 1) / 2 + 1, (fn[1] - 1) / 2 + 1};
    mg->Interpolate(fine, f2, const_cast<float*>(coarse), c2);
}

void ref2d_apply_correction(const int n[2], float* fine, const float* err) {
    MultiGrid2D* mg = mini2();
    int f2[2] = {n[0], n[1]};
    mg->ApplyCorrection(fine, f2, const_cast<float*>(err), f2);
}

void ref2d_set(const int n[2], float* grid, float value, int modify_boundaries) {
    MultiGrid2D* mg = mini2();
    int f2[2] = {n[0], n[1]};
    mg->setToValue(grid, f2, value, modify_boundaries != 0);
}

void ref2d_cycle(const int n[2], const float range[4], const float A[4], int alfa, int nlevels,
                 int mode, int v0, int v1, int v2, int reps, const float* v_in,
                 const float* f_in, float* v_out) {
    MG2 m(n, range, A, alfa);
    MultiGrid2D* mg = m.mg;
    if (nlevels > 0) mg->numGrids = nlevels;
    Grid2D* g = mg->grids2D[0];
    size_t N = vol2(n);
    if (f_in) memcpy(g->h_f, f_in, N * sizeof(float));
    if (v_in) memcpy(g->h_v, v_in, N * sizeof(float));
    if (mode == 0) {
        for (int i = 0; i < reps; i++) mg->VCycle(0, v1, v2);
    } else {
        mg->FullMultiGridVCycle(0, v0, v1, v2);
    }
    memcpy(v_out, g->h_v, N * sizeof(float));
}

// ---------------------------------------------------------------- 1D ------
// InitV only sets the two end values; interior reported as 0.
void ref1d_init(int n, const float range[2], int level, float* v, float* f) {
    float rr[2] = {range[0], range[1]};
    MultiGrid1D mg(n, rr);
    Grid1D* g = mg.grids1D[level];
    mg.setToValue(g->h_v, g->sizeX, 0.0f, false);
    memcpy(v, g->h_v, (size_t)g->sizeX * sizeof(float));
    memcpy(f, g->h_f, (size_t)g->sizeX * sizeof(float));
}

void ref1d_relax(int n, const float range[2], float* v, const float* f, int ncycles) {
    float rr[2] = {range[0], range[1]};
    MultiGrid1D mg(n, rr);
    Grid1D* g = mg.grids1D[0];
    memcpy(g->h_v, v, (size_t)n * sizeof(float));
    memcpy(g->h_f, f, (size_t)n * sizeof(float));
    mg.Relax(g, ncycles);
    memcpy(v, g->h_v, (size_t)n * sizeof(float));
}

void ref1d_residual(int n, const float range[2], const float* v, const float* f, float* r) {
    float rr[2] = {range[0], range[1]};
    MultiGrid1D mg(n, rr);
    Grid1D* g = mg.grids1D[0];
    memcpy(g->h_v, v, (size_t)n * sizeof(float));
    memcpy(g->h_f, f, (size_t)n * sizeof(float));
    float* res = mg.CalculateResidual(g);
    memcpy(r, res, (size_t)n * sizeof(float));
    free(res);
}

static MultiGrid1D* mini1() {
    float rr[2] = {0, 1};
    return new MultiGrid1D(5, rr);
}

void ref1d_restrict(int fn, const float* fine, float* coarse) {
    mini1()->Restrict(const_cast<float*>(fine), fn, coarse, (fn - 1) / 2 + 1);
}

void ref1d_interpolate(int fn, float* fine, const float* coarse) {
    mini1()->Interpolate(fine, fn, const_cast<float*>(coarse), (fn - 1) / 2 + 1);
}

void ref1d_apply_correction(int n, float* fine, const float* err) {
    mini1()->ApplyCorrection(fine, n, const_cast<float*>(err), n);
}

void ref1d_set(int n, float* grid, float value, int modify_boundaries) {
    mini1()->setToValue(grid, n, value, modify_boundaries != 0);
}

void ref1d_cycle(int n, const float range[2], int nlevels, int mode, int v0, int v1, int v2,
                 int reps, const float* v_in, const float* f_in, float* v_out) {
    float rr[2] = {range[0], range[1]};
    MultiGrid1D mg(n, rr);
    if (nlevels > 0) mg.numGrids = nlevels;
    Grid1D* g = mg.grids1D[0];
    if (f_in) memcpy(g->h_f, f_in, (size_t)n * sizeof(float));
    if (v_in) memcpy(g->h_v, v_in, (size_t)n * sizeof(float));
    else mg.setToValue(g->h_v, g->sizeX, 0.0f, false);
    if (mode == 0) {
        for (int i = 0; i < reps; i++) mg.VCycle(0, v1, v2);
    } else {
        mg.FullMultiGridVCycle(0, v0, v1, v2);
    }
    memcpy(v_out, g->h_v, (size_t)n * sizeof(float));
}

}  // extern "C"

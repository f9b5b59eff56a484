// oracle/mg_oracle.cpp -- TEST INFRASTRUCTURE.  extern "C" surface of the CPU
// restatement in mg_oracle.hpp (see the header comment there for scope, pinning
// and who may call it).  Builds to oracle/libmgoracle.so via oracle/Makefile.
//
// Naming: mgo{1,2,3}d_<op>_{f32,f64}.  Arrays are dense, x fastest
// (idx = x + y*sx + z*sx*sy), exactly the reference layout.
#include "mg_oracle.hpp"

#include <chrono>

using namespace mgo;

namespace {

template <class real>
void geom3(const int n[3], const real range[6], real h[3], real a[3]) {
    Grid3<real> g;
    grid3_geometry(g, n, range);
    for (int d = 0; d < 3; d++) { h[d] = g.h[d]; a[d] = g.a[d]; }
}
template <class real>
void geom2(const int n[2], const real range[4], real h[2], real a[2]) {
    Grid2<real> g;
    grid2_geometry(g, n, range);
    for (int d = 0; d < 2; d++) { h[d] = g.h[d]; a[d] = g.a[d]; }
}

template <class real>
void cycle3(const int n[3], const real range[6], int nlevels, int mode, int v0, int v1, int v2, int reps,
            const real* v_in, const real* f_in, real* v_out, int residual_mode) {
    MultiGrid3<real> mg(n, range, nlevels);
    mg.residual_mode = residual_mode;
    Grid3<real>& g = mg.g[0];
    if (f_in) memcpy(g.f.data(), f_in, g.vol() * sizeof(real));
    else mg.init_f_all();
    if (v_in) memcpy(g.v.data(), v_in, g.vol() * sizeof(real));
    else set3(g.v.data(), g.n, (real)0, false);
    if (mode == 0) for (int i = 0; i < reps; i++) mg.VCycle(0, v1, v2);
    else mg.FullMultiGridVCycle(0, v0, v1, v2);
    memcpy(v_out, g.v.data(), g.vol() * sizeof(real));
}

template <class real>
void cycle2(const int n[2], const real range[4], const real A[4], int alfa, int nlevels, int mode, int v0,
            int v1, int v2, int reps, const real* v_in, const real* f_in, real* v_out) {
    MultiGrid2<real> mg(n, range, A, alfa, nlevels);
    Grid2<real>& g = mg.g[0];
    if (f_in) memcpy(g.f.data(), f_in, g.vol() * sizeof(real));
    if (v_in) memcpy(g.v.data(), v_in, g.vol() * sizeof(real));
    if (mode == 0) for (int i = 0; i < reps; i++) mg.VCycle(0, v1, v2);
    else mg.FullMultiGridVCycle(0, v0, v1, v2);
    memcpy(v_out, g.v.data(), g.vol() * sizeof(real));
}

template <class real>
void cycle1(int n, const real range[2], int nlevels, int mode, int v0, int v1, int v2, int reps,
            const real* v_in, const real* f_in, real* v_out) {
    MultiGrid1<real> mg(n, range, nlevels);
    Grid1<real>& g = mg.g[0];
    if (f_in) memcpy(g.f.data(), f_in, (size_t)n * sizeof(real));
    if (v_in) memcpy(g.v.data(), v_in, (size_t)n * sizeof(real));
    else set1(g.v.data(), g.n, (real)0, false);
    if (mode == 0) for (int i = 0; i < reps; i++) mg.VCycle(0, v1, v2);
    else mg.FullMultiGridVCycle(0, v0, v1, v2);
    memcpy(v_out, g.v.data(), (size_t)n * sizeof(real));
}

}  // namespace

#define MGO_STAMP(SFX, real)                                                                            \
    /* ------------------------------ 3D ------------------------------ */                              \
    void mgo3d_init_##SFX(const int n[3], const real range[6], int level, real* v, real* f) {           \
        MultiGrid3<real> mg(n, range, 0);                                                               \
        Grid3<real>& g = mg.g[level];                                                                   \
        init_f3(g);                                                                                     \
        memcpy(v, g.v.data(), g.vol() * sizeof(real));                                                  \
        memcpy(f, g.f.data(), g.vol() * sizeof(real));                                                  \
    }                                                                                                   \
    void mgo3d_relax_##SFX(const int n[3], const real range[6], real* v, const real* f, int ncycles) {  \
        real h[3], a[3];                                                                                \
        geom3<real>(n, range, h, a);                                                                    \
        relax3<real>(v, f, n, h, ncycles);                                                              \
    }                                                                                                   \
    void mgo3d_relax_colour_##SFX(const int n[3], const real range[6], real* v, const real* f, int colour) { \
        real h[3], a[3];                                                                                \
        geom3<real>(n, range, h, a);                                                                    \
        relax3_colour<real>(v, f, n, h, colour);                                                        \
    }                                                                                                   \
    void mgo3d_jacobi_##SFX(const int n[3], const real range[6], real* v, const real* f, real omega, int ncycles) { \
        real h[3], a[3];                                                                                \
        geom3<real>(n, range, h, a);                                                                    \
        jacobi3<real>(v, f, n, h, omega, ncycles);                                                      \
    }                                                                                                   \
    void mgo2d_jacobi_##SFX(const int n[2], const real range[4], const real A[4], int alfa, real* v,    \
                            const real* f, real omega, int ncycles) {                                   \
        real h[2], a[2];                                                                                \
        geom2<real>(n, range, h, a);                                                                    \
        jacobi2<real>(v, f, n, h, a, A, alfa, omega, ncycles);                                          \
    }                                                                                                   \
    void mgo3d_residual_##SFX(const int n[3], const real range[6], const real* v, const real* f,        \
                              real* r, int mode) {                                                      \
        real h[3], a[3];                                                                                \
        geom3<real>(n, range, h, a);                                                                    \
        residual3<real>(v, f, r, n, h, mode);                                                           \
    }                                                                                                   \
    void mgo3d_restrict_##SFX(const int fn[3], const real* fine, real* coarse) {                        \
        int cn[3] = {coarse_size(fn[0]), coarse_size(fn[1]), coarse_size(fn[2])};                       \
        restrict3<real>(fine, fn, coarse, cn);                                                          \
    }                                                                                                   \
    void mgo3d_interpolate_##SFX(const int fn[3], real* fine, const real* coarse) {                     \
        int cn[3] = {coarse_size(fn[0]), coarse_size(fn[1]), coarse_size(fn[2])};                       \
        interpolate3<real>(fine, fn, coarse, cn);                                                       \
    }                                                                                                   \
    void mgo3d_apply_correction_##SFX(const int n[3], real* fine, const real* err) {                    \
        correct3<real>(fine, err, n);                                                                   \
    }                                                                                                   \
    void mgo3d_set_##SFX(const int n[3], real* g, real value, int modify_boundaries) {                  \
        set3<real>(g, n, value, modify_boundaries != 0);                                                \
    }                                                                                                   \
    void mgo3d_cycle_##SFX(const int n[3], const real range[6], int nlevels, int mode, int v0, int v1,  \
                           int v2, int reps, const real* v_in, const real* f_in, real* v_out,           \
                           int residual_mode) {                                                         \
        cycle3<real>(n, range, nlevels, mode, v0, v1, v2, reps, v_in, f_in, v_out, residual_mode);      \
    }                                                                                                   \
    /* ------------------------------ 2D ------------------------------ */                              \
    void mgo2d_init_##SFX(const int n[2], const real range[4], int level, real* v, real* f) {           \
        real A[4] = {0, 0, 0, 0};                                                                       \
        MultiGrid2<real> mg(n, range, A, 0, 0);                                                         \
        Grid2<real>& g = mg.g[level];                                                                   \
        memcpy(v, g.v.data(), g.vol() * sizeof(real));                                                  \
        memcpy(f, g.f.data(), g.vol() * sizeof(real));                                                  \
    }                                                                                                   \
    void mgo2d_relax_##SFX(const int n[2], const real range[4], const real A[4], int alfa, real* v,     \
                           const real* f, int ncycles) {                                                \
        real h[2], a[2];                                                                                \
        geom2<real>(n, range, h, a);                                                                    \
        relax2<real>(v, f, n, h, a, A, alfa, ncycles);                                                  \
    }                                                                                                   \
    void mgo2d_residual_##SFX(const int n[2], const real range[4], const real A[4], int alfa,           \
                              const real* v, const real* f, real* r) {                                  \
        real h[2], a[2];                                                                                \
        geom2<real>(n, range, h, a);                                                                    \
        residual2<real>(v, f, r, n, h, a, A, alfa);                                                     \
    }                                                                                                   \
    void mgo2d_restrict_##SFX(const int fn[2], const real* fine, real* coarse) {                        \
        int cn[2] = {coarse_size(fn[0]), coarse_size(fn[1])};                                           \
        restrict2<real>(fine, fn, coarse, cn);                                                          \
    }                                                                                                   \
    void mgo2d_interpolate_##SFX(const int fn[2], real* fine, const real* coarse) {                     \
        int cn[2] = {coarse_size(fn[0]), coarse_size(fn[1])};                                           \
        interpolate2<real>(fine, fn, coarse, cn);                                                       \
    }                                                                                                   \
    void mgo2d_apply_correction_##SFX(const int n[2], real* fine, const real* err) {                    \
        correct2<real>(fine, err, n);                                                                   \
    }                                                                                                   \
    void mgo2d_set_##SFX(const int n[2], real* g, real value, int modify_boundaries) {                  \
        set2<real>(g, n, value, modify_boundaries != 0);                                                \
    }                                                                                                   \
    void mgo2d_cycle_##SFX(const int n[2], const real range[4], const real A[4], int alfa, int nlevels, \
                           int mode, int v0, int v1, int v2, int reps, const real* v_in,                \
                           const real* f_in, real* v_out) {                                             \
        cycle2<real>(n, range, A, alfa, nlevels, mode, v0, v1, v2, reps, v_in, f_in, v_out);            \
    }                                                                                                   \
    /* ------------------------------ 1D ------------------------------ */                              \
    void mgo1d_init_##SFX(int n, const real range[2], int level, real* v, real* f) {                    \
        MultiGrid1<real> mg(n, range, 0);                                                               \
        Grid1<real>& g = mg.g[level];                                                                   \
        memcpy(v, g.v.data(), (size_t)g.n * sizeof(real));                                              \
        memcpy(f, g.f.data(), (size_t)g.n * sizeof(real));                                              \
    }                                                                                                   \
    void mgo1d_relax_##SFX(int n, const real range[2], real* v, const real* f, int ncycles) {           \
        Grid1<real> g;                                                                                  \
        grid1_init(g, n, range);                                                                        \
        relax1<real>(v, f, n, g.h, g.a, ncycles);                                                       \
    }                                                                                                   \
    void mgo1d_residual_##SFX(int n, const real range[2], const real* v, const real* f, real* r) {      \
        Grid1<real> g;                                                                                  \
        grid1_init(g, n, range);                                                                        \
        residual1<real>(v, f, r, n, g.h, g.a);                                                          \
    }                                                                                                   \
    void mgo1d_restrict_##SFX(int fn, const real* fine, real* coarse) {                                 \
        restrict1<real>(fine, fn, coarse, coarse_size(fn));                                             \
    }                                                                                                   \
    void mgo1d_interpolate_##SFX(int fn, real* fine, const real* coarse) {                              \
        interpolate1<real>(fine, fn, coarse, coarse_size(fn));                                          \
    }                                                                                                   \
    void mgo1d_apply_correction_##SFX(int n, real* fine, const real* err) { correct1<real>(fine, err, n); } \
    void mgo1d_set_##SFX(int n, real* g, real value, int modify_boundaries) {                           \
        set1<real>(g, n, value, modify_boundaries != 0);                                                \
    }                                                                                                   \
    void mgo1d_cycle_##SFX(int n, const real range[2], int nlevels, int mode, int v0, int v1, int v2,   \
                           int reps, const real* v_in, const real* f_in, real* v_out) {                 \
        cycle1<real>(n, range, nlevels, mode, v0, v1, v2, reps, v_in, f_in, v_out);                     \
    }

extern "C" {

MGO_STAMP(f32, float)
MGO_STAMP(f64, double)

int mgo_num_grids(int min_size) { return num_grids(min_size); }

// 64-bit FNV-1a-style hash over 32-bit words in memory order (bit patterns, NaN-safe):
// h = 0xcbf29ce484222325; per word h = (h ^ w) * 0x100000001b3.
uint64_t mgo_fnv_words32(const uint32_t* w, size_t nwords) {
    uint64_t h = 0xcbf29ce484222325ULL;
    for (size_t i = 0; i < nwords; i++) h = (h ^ (uint64_t)w[i]) * 0x100000001b3ULL;
    return h;
}

// CPU-baseline timers (bench.py cpu_baseline leg): seconds for `sweeps` red-black
// sweeps of the 3D smoother / `reps` V-cycles, single thread, reference loop nest.
double mgo3d_time_relax_f32(int n, int sweeps) {
    int nn[3] = {n, n, n};
    float range[6] = {0, 1, 0, 1, 0, 1};
    MultiGrid3<float> mg(nn, range, 1);
    mg.init_f_all();
    auto t0 = std::chrono::steady_clock::now();
    relax3<float>(mg.g[0].v.data(), mg.g[0].f.data(), mg.g[0].n, mg.g[0].h, sweeps);
    auto t1 = std::chrono::steady_clock::now();
    return std::chrono::duration<double>(t1 - t0).count();
}
double mgo3d_time_relax_f64(int n, int sweeps) {
    int nn[3] = {n, n, n};
    double range[6] = {0, 1, 0, 1, 0, 1};
    MultiGrid3<double> mg(nn, range, 1);
    mg.init_f_all();
    auto t0 = std::chrono::steady_clock::now();
    relax3<double>(mg.g[0].v.data(), mg.g[0].f.data(), mg.g[0].n, mg.g[0].h, sweeps);
    auto t1 = std::chrono::steady_clock::now();
    return std::chrono::duration<double>(t1 - t0).count();
}
double mgo3d_time_vcycle_f64(int n, int nlevels, int v1, int v2, int reps) {
    int nn[3] = {n, n, n};
    double range[6] = {0, 1, 0, 1, 0, 1};
    MultiGrid3<double> mg(nn, range, nlevels);
    mg.init_f_all();
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < reps; i++) mg.VCycle(0, v1, v2);
    auto t1 = std::chrono::steady_clock::now();
    return std::chrono::duration<double>(t1 - t0).count();
}
double mgo3d_time_vcycle_f32(int n, int nlevels, int v1, int v2, int reps) {
    int nn[3] = {n, n, n};
    float range[6] = {0, 1, 0, 1, 0, 1};
    MultiGrid3<float> mg(nn, range, nlevels);
    mg.init_f_all();
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < reps; i++) mg.VCycle(0, v1, v2);
    auto t1 = std::chrono::steady_clock::now();
    return std::chrono::duration<double>(t1 - t0).count();
}

}  // extern "C"

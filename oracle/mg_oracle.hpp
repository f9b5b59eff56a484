// oracle/mg_oracle.hpp -- TEST INFRASTRUCTURE.  NOT part of the product path.
//
// CPU restatement of the NOCUDA_TESI multigrid operators of MisterPup/PDE-MultiGrid,
// templated on the real type.  The float instantiation is pinned BIT-EXACT against
// the compiled, unmodified reference (oracle/_ref/libmgref.so, built by `make ref`)
// by tests/test_oracle_vs_ref.py and against the committed fixtures in tests/golden/
// (generated from that reference by oracle/gen_golden.py).  The double instantiation
// is the fp64 oracle for the HIP kernels (the reference itself is fp32-only).
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this.
//
// Path abbreviations in the citations below:
//   N1 = NOCUDA_TESI/EQUAZIONE 1D/        N2 = NOCUDA_TESI/PDE Lyapunov 2D/
//   N3 = NOCUDA_TESI/POISSON_3D(TESI)/
//
// Bit-exactness rules followed everywhere: expressions are evaluated in `real`
// in the reference's association order, no FMA contraction (-ffp-contract=off),
// true IEEE division, int operands converted to real before use.  Loop nests keep
// the reference's Y -> X -> Z order so that this file also serves as the
// single-threaded CPU baseline ("port") that bench.py times.
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace mgo {

static const double PI_D = 3.141592653589793;  // N3/inclusion.h:9

enum ResidualMode { REF_COMPAT = 0, CORRECT = 1 };

// numGrids = (int)log2(minSize-1)            N3/MultiGrid3D.cpp:33-34
inline int num_grids(int min_size) { return (int)std::log2((double)(min_size - 1)); }
inline int coarse_size(int n) { return ((n - 1) / 2) + 1; }  // N3/MultiGrid3D.cpp:40-42

// ===========================================================================
//                                   3D
// ===========================================================================
template <class real>
struct Grid3 {  // N3/Grid3D.h:4-38
    int n[3];
    real h[3];      // h_x,h_y,h_z = range/(real)(size-1)      N3/Grid3D.cpp:43-45
    real a[3];      // x_a,y_a,z_a
    std::vector<real> v, f;
    size_t vol() const { return (size_t)n[0] * n[1] * n[2]; }
};

template <class real>
void grid3_geometry(Grid3<real>& g, const int n[3], const real range[6]) {
    for (int d = 0; d < 3; d++) {
        g.n[d] = n[d];
        real span = range[2 * d + 1] - range[2 * d];  // N3/Grid3D.cpp:31-33
        g.a[d] = range[2 * d];
        g.h[d] = span / (real)(n[d] - 1);
    }
}

// InitV: boundary v = 0, interior untouched.            N3/Grid3D.cpp:61-76
template <class real>
void init_v3(Grid3<real>& g) {
    const int sx = g.n[0], sy = g.n[1], sz = g.n[2];
    for (int y = 0; y < sy; y++)
        for (int x = 0; x < sx; x++)
            for (int z = 0; z < sz; z++)
                if (x == 0 || x == sx - 1 || y == 0 || y == sy - 1 || z == 0 || z == sz - 1)
                    g.v[x + (size_t)y * sx + (size_t)z * sx * sy] = (real)0;
}

// InitF: f = -3*PI*PI*sin(PI*x)*sin(PI*y)*sin(PI*z), double math left to right,
// x = x_a + posX*h_x in `real`.                          N3/Grid3D.cpp:78-96 (:88-92)
template <class real>
void init_f3(Grid3<real>& g) {
    const int sx = g.n[0], sy = g.n[1], sz = g.n[2];
    std::vector<double> sX(sx), sY(sy), sZ(sz);
    for (int i = 0; i < sx; i++) { real x = g.a[0] + i * g.h[0]; sX[i] = std::sin(PI_D * x); }
    for (int i = 0; i < sy; i++) { real y = g.a[1] + i * g.h[1]; sY[i] = std::sin(PI_D * y); }
    for (int i = 0; i < sz; i++) { real z = g.a[2] + i * g.h[2]; sZ[i] = std::sin(PI_D * z); }
    const double c = -3 * PI_D * PI_D;
    for (int z = 0; z < sz; z++)
        for (int y = 0; y < sy; y++)
            for (int x = 0; x < sx; x++)
                g.f[x + (size_t)y * sx + (size_t)z * sx * sy] = (real)(c * sX[x] * sY[y] * sZ[z]);
}

// Relax: ncycles x (red pass, black pass).               N3/MultiGrid3D.cpp:489-567
template <class real>
void relax3(real* v, const real* f, const int n[3], const real h[3], int ncycles) {
    const int sx = n[0], sy = n[1], sz = n[2];
    const real hx2 = h[0] * h[0], hy2 = h[1] * h[1], hz2 = h[2] * h[2];  // :498-500
    const size_t sxy = (size_t)sx * sy;
    for (int k = 0; k < ncycles; k++) {
        for (int colour = 0; colour < 2; colour++) {  // :515 red = even sum, :544 black
            for (int y = 1; y < sy - 1; y++)
                for (int x = 1; x < sx - 1; x++)
                    for (int z = 1; z < sz - 1; z++) {
                        if (((y + x + z) % 2 == 0) != (colour == 0)) continue;
                        const size_t i = x + (size_t)y * sx + (size_t)z * sxy;
                        const real O = v[i - 1], E = v[i + 1];
                        const real N = v[i - sx], S = v[i + sx];
                        const real D = v[i - sxy], U = v[i + sxy];
                        // :532 / :561
                        v[i] = (O * (hy2 * hz2) + E * (hy2 * hz2) + N * (hx2 * hz2) + S * (hx2 * hz2) +
                                D * (hx2 * hy2) + U * (hx2 * hy2) - f[i] * hx2 * hy2 * hz2) /
                               (2 * (hy2 * hz2 + hx2 * hz2 + hx2 * hy2));
                    }
        }
    }
}

// One colour pass of Relax (colour 0 = red: (x+y+z) even, :515; 1 = black, :544) -- the unit between
// two ghost-plane exchanges of the z-slab decomposition (test emulation of the multi-GPU schedule).
template <class real>
void relax3_colour(real* v, const real* f, const int n[3], const real h[3], int colour) {
    const int sx = n[0], sy = n[1], sz = n[2];
    const real hx2 = h[0] * h[0], hy2 = h[1] * h[1], hz2 = h[2] * h[2];
    const size_t sxy = (size_t)sx * sy;
    for (int y = 1; y < sy - 1; y++)
        for (int x = 1; x < sx - 1; x++)
            for (int z = 1; z < sz - 1; z++) {
                if (((y + x + z) % 2 == 0) != (colour == 0)) continue;
                const size_t i = x + (size_t)y * sx + (size_t)z * sxy;
                const real O = v[i - 1], E = v[i + 1];
                const real N = v[i - sx], S = v[i + sx];
                const real D = v[i - sxy], U = v[i + sxy];
                v[i] = (O * (hy2 * hz2) + E * (hy2 * hz2) + N * (hx2 * hz2) + S * (hx2 * hz2) + D * (hx2 * hy2) + U * (hx2 * hy2) -
                        f[i] * hx2 * hy2 * hz2) /
                       (2 * (hy2 * hz2 + hx2 * hz2 + hx2 * hy2));
            }
}

// Weighted Jacobi sweeps (ADDITION: named by north_star, absent from the reference -- parity unpinned; this is the
// restatement the HIP kernel is compared with): v <- v + omega*(u - v), u = the Gauss-Seidel value of :532
// evaluated on the old iterate.
template <class real>
void jacobi3(real* v, const real* f, const int n[3], const real h[3], real omega, int ncycles) {
    const int sx = n[0], sy = n[1], sz = n[2];
    const real hx2 = h[0] * h[0], hy2 = h[1] * h[1], hz2 = h[2] * h[2];
    const size_t sxy = (size_t)sx * sy;
    std::vector<real> old((size_t)sx * sy * sz);
    for (int k = 0; k < ncycles; k++) {
        memcpy(old.data(), v, old.size() * sizeof(real));
        for (int z = 1; z < sz - 1; z++)
            for (int y = 1; y < sy - 1; y++)
                for (int x = 1; x < sx - 1; x++) {
                    const size_t i = x + (size_t)y * sx + (size_t)z * sxy;
                    const real O = old[i - 1], E = old[i + 1], N = old[i - sx], S = old[i + sx], D = old[i - sxy], U = old[i + sxy];
                    const real u = (O * (hy2 * hz2) + E * (hy2 * hz2) + N * (hx2 * hz2) + S * (hx2 * hz2) + D * (hx2 * hy2) +
                                    U * (hx2 * hy2) - f[i] * hx2 * hy2 * hz2) /
                                   (2 * (hy2 * hz2 + hx2 * hz2 + hx2 * hy2));
                    v[i] = old[i] + omega * (u - old[i]);
                }
    }
}

// CalculateResidual.  REF_COMPAT keeps the reference's sign quirk (-S, -U);
// CORRECT uses +S, +U.                                   N3/MultiGrid3D.cpp:678-730 (:723)
template <class real>
void residual3(const real* v, const real* f, real* r, const int n[3], const real h[3], int mode) {
    const int sx = n[0], sy = n[1], sz = n[2];
    const real hx2 = h[0] * h[0], hy2 = h[1] * h[1], hz2 = h[2] * h[2];
    const size_t sxy = (size_t)sx * sy;
    for (int y = 0; y < sy; y++)
        for (int x = 0; x < sx; x++)
            for (int z = 0; z < sz; z++) {
                const size_t i = x + (size_t)y * sx + (size_t)z * sxy;
                if (x == 0 || x == sx - 1 || y == 0 || y == sy - 1 || z == 0 || z == sz - 1) {
                    r[i] = (real)0;  // :704-705
                    continue;
                }
                const real O = v[i - 1], E = v[i + 1];
                const real N = v[i - sx], S = v[i + sx];
                const real D = v[i - sxy], U = v[i + sxy];
                if (mode == REF_COMPAT)
                    r[i] = f[i] - ((O - 2 * v[i] + E) / hx2) - ((N - 2 * v[i] - S) / hy2) -
                           ((D - 2 * v[i] - U) / hz2);
                else
                    r[i] = f[i] - ((O - 2 * v[i] + E) / hx2) - ((N - 2 * v[i] + S) / hy2) -
                           ((D - 2 * v[i] + U) / hz2);
            }
}

// Restrict: 27-point full weighting, boundary = injection.  N3/MultiGrid3D.cpp:50-184
// Name key of the reference (:122-177): suffix _C/_N/_S = y, y-1, y+1;
// prefix N/S = z+1/z-1, E/O = x+1/x-1.
template <class real>
void restrict3(const real* fine, const int fn[3], real* coarse, const int cn[3]) {
    const int fx = fn[0], fy = fn[1];
    const int cx = cn[0], cy = cn[1], cz = cn[2];
    const size_t fxy = (size_t)fx * fy;
    for (int py = 0; py < cy; py++)
        for (int px = 0; px < cx; px++)
            for (int pz = 0; pz < cz; pz++) {
                const size_t ci = px + (size_t)py * cx + (size_t)pz * cx * cy;
                const size_t fi = 2 * px + (size_t)(2 * py) * fx + (size_t)(2 * pz) * fxy;
                if (px == 0 || px == cx - 1 || py == 0 || py == cy - 1 || pz == 0 || pz == cz - 1) {
                    coarse[ci] = fine[fi];  // :113-119
                    continue;
                }
#define F_(dx, dy, dz) fine[fi + (dx) + (ptrdiff_t)(dy) * fx + (ptrdiff_t)(dz) * (ptrdiff_t)fxy]
                const real C_C = F_(0, 0, 0), N_C = F_(0, 0, 1), S_C = F_(0, 0, -1);
                const real E_C = F_(1, 0, 0), O_C = F_(-1, 0, 0);
                const real NE_C = F_(1, 0, 1), NO_C = F_(-1, 0, 1), SE_C = F_(1, 0, -1), SO_C = F_(-1, 0, -1);
                const real C_N = F_(0, -1, 0), N_N = F_(0, -1, 1), S_N = F_(0, -1, -1);
                const real E_N = F_(1, -1, 0), O_N = F_(-1, -1, 0);
                const real NE_N = F_(1, -1, 1), NO_N = F_(-1, -1, 1), SE_N = F_(1, -1, -1), SO_N = F_(-1, -1, -1);
                const real C_S = F_(0, 1, 0), N_S = F_(0, 1, 1), S_S = F_(0, 1, -1);
                const real E_S = F_(1, 1, 0), O_S = F_(-1, 1, 0);
                const real NE_S = F_(1, 1, 1), NO_S = F_(-1, 1, 1), SE_S = F_(1, 1, -1), SO_S = F_(-1, 1, -1);
#undef F_
                // :180
                coarse[ci] = (1 / 8.0f) * (C_C) + (1 / 16.0f) * ((N_C + E_C + S_C + O_C) + (C_N + C_S)) +
                             (1 / 32.0f) * ((NE_C + SE_C + SO_C + NO_C) + (N_N + E_N + S_N + O_N) +
                                            (N_S + E_S + S_S + O_S)) +
                             (1 / 64.0f) * ((NE_N + SE_N + SO_N + NO_N) + (NE_S + SE_S + SO_S + NO_S));
            }
}

// Interpolate: trilinear by parity class, interior only.  N3/MultiGrid3D.cpp:186-335
template <class real>
void interpolate3(real* fine, const int fn[3], const real* coarse, const int cn[3]) {
    const int fx = fn[0], fy = fn[1], fz = fn[2];
    const int cx = cn[0], cy = cn[1];
    const size_t cxy = (size_t)cx * cy;
    for (int y = 1; y < fy - 1; y++)
        for (int x = 1; x < fx - 1; x++)
            for (int z = 1; z < fz - 1; z++) {
                const size_t fi = x + (size_t)y * fx + (size_t)z * fx * fy;
                const size_t ci = (x / 2) + (size_t)(y / 2) * cx + (size_t)(z / 2) * cxy;  // :208-213
#define C_(dx, dy, dz) coarse[ci + (dx) + (size_t)(dy) * cx + (size_t)(dz) * cxy]
                const bool ox = x % 2 != 0, oy = y % 2 != 0, oz = z % 2 != 0;
                if (!oy && !ox && !oz) fine[fi] = C_(0, 0, 0);                                   // :216
                else if (!oy && ox && !oz) fine[fi] = (1 / 2.0f) * (C_(0, 0, 0) + C_(1, 0, 0));   // :222-229
                else if (oy && !ox && !oz) fine[fi] = (1 / 2.0f) * (C_(0, 0, 0) + C_(0, 1, 0));   // :233-240
                else if (oy && ox && !oz)                                                         // :244-255
                    fine[fi] = (1 / 4.0f) * (C_(0, 0, 0) + C_(1, 0, 0) + C_(0, 1, 0) + C_(1, 1, 0));
                else if (!oy && !ox && oz) fine[fi] = (1 / 2.0f) * (C_(0, 0, 0) + C_(0, 0, 1));   // :261-268
                else if (!oy && ox && oz)                                                         // :272-283
                    fine[fi] = (1 / 4.0f) * (C_(0, 0, 1) + C_(1, 0, 1) + C_(0, 0, 0) + C_(1, 0, 0));
                else if (oy && !ox && oz)                                                         // :287-298
                    fine[fi] = (1 / 4.0f) * (C_(0, 0, 0) + C_(0, 0, 1) + C_(0, 1, 0) + C_(0, 1, 1));
                else                                                                              // :302-329
                    fine[fi] = (1 / 8.0f) * (C_(0, 0, 0) + C_(0, 0, 1) + C_(1, 0, 1) + C_(1, 0, 0) +
                                             C_(0, 1, 0) + C_(0, 1, 1) + C_(1, 1, 1) + C_(1, 1, 0));
#undef C_
            }
}

// ApplyCorrection: fine += err on the interior.           N3/MultiGrid3D.cpp:649-676
template <class real>
void correct3(real* fine, const real* err, const int n[3]) {
    for (int y = 1; y < n[1] - 1; y++)
        for (int x = 1; x < n[0] - 1; x++)
            for (int z = 1; z < n[2] - 1; z++) {
                const size_t i = x + (size_t)y * n[0] + (size_t)z * n[0] * n[1];
                fine[i] = fine[i] + err[i];
            }
}

// setToValue.                                             N3/MultiGrid3D.cpp:587-621
template <class real>
void set3(real* g, const int n[3], real value, bool modify_boundaries) {
    const int lo = modify_boundaries ? 0 : 1;
    for (int y = lo; y < n[1] - lo; y++)
        for (int x = lo; x < n[0] - lo; x++)
            for (int z = lo; z < n[2] - lo; z++) g[x + (size_t)y * n[0] + (size_t)z * n[0] * n[1]] = value;
}

template <class real>
struct MultiGrid3 {  // N3/MultiGrid3D.h:6-33
    std::vector<Grid3<real>> g;
    int numGrids;
    int residual_mode = REF_COMPAT;

    // InitGrids.  nlevels = 0 -> reference rule; nlevels > 0 == overwriting the public
    // numGrids after construction.          N3/MultiGrid3D.cpp:19-47
    MultiGrid3(const int n[3], const real range[6], int nlevels) {
        int m = n[0];
        if (n[1] < m) m = n[1];
        if (n[2] < m) m = n[2];
        const int native = num_grids(m);
        numGrids = nlevels > 0 ? nlevels : native;
        g.resize(numGrids > native ? numGrids : native);
        int cur[3] = {n[0], n[1], n[2]};
        for (size_t l = 0; l < g.size(); l++) {
            grid3_geometry(g[l], cur, range);
            // the reference mallocs (uninitialised); zero-fill here, every read of an
            // interior v is preceded by a write in both cycles.
            g[l].v.assign(g[l].vol(), (real)0);
            g[l].f.assign(g[l].vol(), (real)0);
            init_v3(g[l]);
            for (int d = 0; d < 3; d++) cur[d] = coarse_size(cur[d]);
        }
    }
    void init_f_all() { for (auto& l : g) init_f3(l); }

    void VCycle(int id, int v1, int v2) {  // N3/MultiGrid3D.cpp:623-647
        Grid3<real>& fine = g[id];
        relax3(fine.v.data(), fine.f.data(), fine.n, fine.h, v1);
        if (id != numGrids - 1) {
            Grid3<real>& coarse = g[id + 1];
            std::vector<real> res(fine.vol());
            residual3(fine.v.data(), fine.f.data(), res.data(), fine.n, fine.h, residual_mode);
            restrict3(res.data(), fine.n, coarse.f.data(), coarse.n);
            set3(coarse.v.data(), coarse.n, (real)0, true);
            VCycle(id + 1, v1, v2);
            std::vector<real> err(fine.vol());
            interpolate3(err.data(), fine.n, coarse.v.data(), coarse.n);
            correct3(fine.v.data(), err.data(), fine.n);
        }
        relax3(fine.v.data(), fine.f.data(), fine.n, fine.h, v2);
    }

    void FullMultiGridVCycle(int id, int v0, int v1, int v2) {  // N3/MultiGrid3D.cpp:569-585
        Grid3<real>& fine = g[id];
        if (id != numGrids - 1) {
            Grid3<real>& coarse = g[id + 1];
            restrict3(fine.f.data(), fine.n, coarse.f.data(), coarse.n);
            FullMultiGridVCycle(id + 1, v0, v1, v2);
            interpolate3(fine.v.data(), fine.n, coarse.v.data(), coarse.n);
        } else {
            set3(fine.v.data(), fine.n, (real)0, false);
        }
        for (int i = 0; i < v0; i++) VCycle(id, v1, v2);
    }
};

// ===========================================================================
//                                   2D
// ===========================================================================
template <class real>
struct Grid2 {  // N2/Grid2D.h:4-33
    int n[2];
    real h[2];
    real a[2];
    std::vector<real> v, f;
    size_t vol() const { return (size_t)n[0] * n[1]; }
};

template <class real>
void grid2_geometry(Grid2<real>& g, const int n[2], const real range[4]) {
    for (int d = 0; d < 2; d++) {
        g.n[d] = n[d];
        real span = range[2 * d + 1] - range[2 * d];  // N2/Grid2D.cpp:24-35
        g.a[d] = range[2 * d];
        g.h[d] = span / (real)(n[d] - 1);
    }
}

// InitV: boundary = 2*xj*xj-4*xj*yi+2*yi*yi in `real`, interior 0.  N2/Grid2D.cpp:50-68
template <class real>
void init_v2(Grid2<real>& g) {
    const int sx = g.n[0], sy = g.n[1];
    for (int y = 0; y < sy; y++)
        for (int x = 0; x < sx; x++) {
            const size_t i = x + (size_t)y * sx;
            if (x == 0 || x == sx - 1 || y == 0 || y == sy - 1) {
                real yi = g.a[1] + y * g.h[1];
                real xj = g.a[0] + x * g.h[0];
                g.v[i] = 2 * xj * xj - 4 * xj * yi + 2 * yi * yi;
            } else
                g.v[i] = (real)0;
        }
}

// Relax: 3-point upwind RBGS.                             N2/MultiGrid2D.cpp:199-273
template <class real>
void relax2(real* v, const real* f, const int n[2], const real h[2], const real a[2], const real A[4],
            int alfa, int ncycles) {
    const int sx = n[0], sy = n[1];
    const real hx = h[0], hy = h[1];
    for (int k = 0; k < ncycles; k++)
        for (int colour = 0; colour < 2; colour++)
            for (int y = 0; y < sy; y++)
                for (int x = 0; x < sx; x++) {
                    if (((y + x) % 2 == 0) != (colour == 0)) continue;               // :223 / :250
                    if (x == 0 || x == sx - 1 || y == 0 || y == sy - 1) continue;     // :227-228
                    const size_t i = x + (size_t)y * sx;
                    real xj = a[0] + x * hx;                                          // :230-231
                    real yi = a[1] + y * hy;
                    real K1 = A[0] * xj + A[1] * yi;                                  // :233-234
                    real K2 = A[2] * xj + A[3] * yi;
                    real den = K1 * hy + K2 * hx - alfa * hx * hy;                    // :236
                    v[i] = (hy * K1 * v[i + 1] + hx * K2 * v[i + sx] - f[i] * hx * hy) / (den);  // :241
                }
}

// Weighted Jacobi, 2D (addition, see jacobi3)
template <class real>
void jacobi2(real* v, const real* f, const int n[2], const real h[2], const real a[2], const real A[4], int alfa, real omega,
             int ncycles) {
    const int sx = n[0], sy = n[1];
    const real hx = h[0], hy = h[1];
    std::vector<real> old((size_t)sx * sy);
    for (int k = 0; k < ncycles; k++) {
        memcpy(old.data(), v, old.size() * sizeof(real));
        for (int y = 1; y < sy - 1; y++)
            for (int x = 1; x < sx - 1; x++) {
                const size_t i = x + (size_t)y * sx;
                real xj = a[0] + x * hx;
                real yi = a[1] + y * hy;
                real K1 = A[0] * xj + A[1] * yi;
                real K2 = A[2] * xj + A[3] * yi;
                real den = K1 * hy + K2 * hx - alfa * hx * hy;
                const real u = (hy * K1 * old[i + 1] + hx * K2 * old[i + sx] - f[i] * hx * hy) / (den);
                v[i] = old[i] + omega * (u - old[i]);
            }
    }
}

// CalculateResidual (consistent with Relax).              N2/MultiGrid2D.cpp:367-408 (:403)
template <class real>
void residual2(const real* v, const real* f, real* r, const int n[2], const real h[2], const real a[2],
               const real A[4], int alfa) {
    const int sx = n[0], sy = n[1];
    const real hx = h[0], hy = h[1];
    for (int y = 0; y < sy; y++)
        for (int x = 0; x < sx; x++) {
            const size_t i = x + (size_t)y * sx;
            if (x == 0 || x == sx - 1 || y == 0 || y == sy - 1) { r[i] = (real)0; continue; }
            real xj = a[0] + x * hx;
            real yi = a[1] + y * hy;
            real K1 = A[0] * xj + A[1] * yi;
            real K2 = A[2] * xj + A[3] * yi;
            r[i] = f[i] - (hy * K1 * v[i + 1] + hx * K2 * v[i + sx] -
                           v[i] * (hy * K1 + hx * K2 - alfa * hx * hy)) / (hx * hy);
        }
}

// Restrict: 9-point full weighting, boundary injection.   N2/MultiGrid2D.cpp:63-126 (:123)
template <class real>
void restrict2(const real* fine, const int fn[2], real* coarse, const int cn[2]) {
    const int fx = fn[0];
    const int cx = cn[0], cy = cn[1];
    for (int py = 0; py < cy; py++)
        for (int px = 0; px < cx; px++) {
            const size_t ci = px + (size_t)py * cx;
            const size_t fi = 2 * px + (size_t)(2 * py) * fx;
            if (px == 0 || px == cx - 1 || py == 0 || py == cy - 1) { coarse[ci] = fine[fi]; continue; }
            const real C = fine[fi], N = fine[fi - fx], S = fine[fi + fx], E = fine[fi + 1], O = fine[fi - 1];
            const real NE = fine[fi + 1 - fx], NO = fine[fi - 1 - fx], SE = fine[fi + 1 + fx], SO = fine[fi - 1 + fx];
            coarse[ci] = (1 / 16.0f) * (NO + NE + SO + SE + 2 * (O + E + N + S) + 4 * C);
        }
}

// Interpolate: bilinear, interior only.                   N2/MultiGrid2D.cpp:128-196
template <class real>
void interpolate2(real* fine, const int fn[2], const real* coarse, const int cn[2]) {
    const int fx = fn[0], fy = fn[1];
    const int cx = cn[0];
    for (int y = 1; y < fy - 1; y++)
        for (int x = 1; x < fx - 1; x++) {
            const size_t fi = x + (size_t)y * fx;
            const size_t ci = (x / 2) + (size_t)(y / 2) * cx;
            const bool ox = x % 2 != 0, oy = y % 2 != 0;
            if (!oy && !ox) fine[fi] = coarse[ci];                                                // :153-156
            else if (oy && !ox) fine[fi] = (1 / 2.0f) * (coarse[ci] + coarse[ci + cx]);           // :158-166
            else if (!oy && ox) fine[fi] = (1 / 2.0f) * (coarse[ci] + coarse[ci + 1]);            // :169-177
            else fine[fi] = (1 / 4.0f) * (coarse[ci] + coarse[ci + 1] + coarse[ci + cx] + coarse[ci + cx + 1]);  // :180-192
        }
}

template <class real>
void correct2(real* fine, const real* err, const int n[2]) {  // N2/MultiGrid2D.cpp:343-366
    for (int y = 1; y < n[1] - 1; y++)
        for (int x = 1; x < n[0] - 1; x++) {
            const size_t i = x + (size_t)y * n[0];
            fine[i] = fine[i] + err[i];
        }
}

template <class real>
void set2(real* g, const int n[2], real value, bool modify_boundaries) {  // N2/MultiGrid2D.cpp:275-292
    const int lo = modify_boundaries ? 0 : 1;
    for (int y = lo; y < n[1] - lo; y++)
        for (int x = lo; x < n[0] - lo; x++) g[x + (size_t)y * n[0]] = value;
}

template <class real>
struct MultiGrid2 {  // N2/MultiGrid2D.h:6-37
    std::vector<Grid2<real>> g;
    int numGrids;
    real A[4];
    int alfa;

    MultiGrid2(const int n[2], const real range[4], const real A_[4], int alfa_, int nlevels) {
        for (int i = 0; i < 4; i++) A[i] = A_[i];
        alfa = alfa_;
        int m = n[0] < n[1] ? n[0] : n[1];
        const int native = num_grids(m);  // N2/MultiGrid2D.cpp:29-30
        numGrids = nlevels > 0 ? nlevels : native;
        g.resize(numGrids > native ? numGrids : native);
        int cur[2] = {n[0], n[1]};
        for (size_t l = 0; l < g.size(); l++) {
            grid2_geometry(g[l], cur, range);
            g[l].v.assign(g[l].vol(), (real)0);
            g[l].f.assign(g[l].vol(), (real)0);  // InitF: f = 0     N2/Grid2D.cpp:70-80
            init_v2(g[l]);
            for (int d = 0; d < 2; d++) cur[d] = coarse_size(cur[d]);
        }
    }

    void VCycle(int id, int v1, int v2) {  // N2/MultiGrid2D.cpp:314-340
        Grid2<real>& fine = g[id];
        relax2(fine.v.data(), fine.f.data(), fine.n, fine.h, fine.a, A, alfa, v1);
        if (id != numGrids - 1) {
            Grid2<real>& coarse = g[id + 1];
            std::vector<real> res(fine.vol());
            residual2(fine.v.data(), fine.f.data(), res.data(), fine.n, fine.h, fine.a, A, alfa);
            restrict2(res.data(), fine.n, coarse.f.data(), coarse.n);
            set2(coarse.v.data(), coarse.n, (real)0, true);
            VCycle(id + 1, v1, v2);
            std::vector<real> err(fine.vol());
            interpolate2(err.data(), fine.n, coarse.v.data(), coarse.n);
            correct2(fine.v.data(), err.data(), fine.n);
        }
        relax2(fine.v.data(), fine.f.data(), fine.n, fine.h, fine.a, A, alfa, v2);
    }

    void FullMultiGridVCycle(int id, int v0, int v1, int v2) {  // N2/MultiGrid2D.cpp:296-312
        Grid2<real>& fine = g[id];
        if (id != numGrids - 1) {
            Grid2<real>& coarse = g[id + 1];
            restrict2(fine.f.data(), fine.n, coarse.f.data(), coarse.n);
            FullMultiGridVCycle(id + 1, v0, v1, v2);
            interpolate2(fine.v.data(), fine.n, coarse.v.data(), coarse.n);
        } else {
            set2(fine.v.data(), fine.n, (real)0, false);
        }
        for (int i = 0; i < v0; i++) VCycle(id, v1, v2);
    }
};

// ===========================================================================
//                                   1D
// ===========================================================================
template <class real>
struct Grid1 {  // N1/Grid1D.h:4-26
    int n;
    real h, a, b;
    std::vector<real> v, f;
};

// exp() on a `real` argument resolves to the overload of that type (expf for
// float under g++'s <math.h>).
template <class real>
void grid1_init(Grid1<real>& g, int n, const real range[2]) {
    g.n = n;
    real span = range[1] - range[0];  // N1/Grid1D.cpp:10-15
    g.a = range[0];
    g.b = range[1];
    g.h = span / (real)(n - 1);
    g.v.assign(n, (real)0);
    g.f.assign(n, (real)0);
    g.v[0] = (std::exp(g.a) + g.a - 3) / (1 + std::exp(-g.a));      // N1/Grid1D.cpp:30-34
    g.v[n - 1] = (std::exp(g.b) + g.b - 3) / (1 + std::exp(-g.b));
    for (int x = 0; x < n; x++) {                                    // N1/Grid1D.cpp:36-43
        real xj = g.a + x * g.h;
        g.f[x] = std::exp(xj);
    }
}

template <class real>
void relax1(real* v, const real* f, int n, real h, real a, int ncycles) {  // N1/MultiGrid1D.cpp:79-118
    for (int k = 0; k < ncycles; k++)
        for (int colour = 0; colour < 2; colour++)
            for (int x = 0; x < n; x++) {
                if ((x % 2 == 0) != (colour == 0)) continue;  // :94 red = even, :108 black = odd
                if (x == 0 || x == n - 1) continue;
                real xj = a + x * h;
                v[x] = (v[x + 1] * (std::exp(xj) + 1) - f[x] * h * (std::exp(xj) + 1)) / (std::exp(xj) + 1 + h);  // :101/:114
            }
}

template <class real>
void residual1(const real* v, const real* f, real* r, int n, real h, real a) {  // N1/MultiGrid1D.cpp:190-214
    for (int x = 0; x < n; x++) {
        if (x == 0 || x == n - 1) { r[x] = (real)0; continue; }
        real xj = a + x * h;
        r[x] = f[x] - (v[x + 1] - v[x]) / h - v[x] / (std::exp(xj) + 1);  // :210 (sign quirk)
    }
}

template <class real>
void restrict1(const real* fine, int fn, real* coarse, int cn) {  // N1/MultiGrid1D.cpp:34-58
    (void)fn;
    for (int c = 0; c < cn; c++) {
        if (c == 0 || c == cn - 1) { coarse[c] = fine[2 * c]; continue; }
        const real C = fine[2 * c], E = fine[2 * c + 1], O = fine[2 * c - 1];
        coarse[c] = (1 / 4.0f) * (O + 2 * C + E);  // :56
    }
}

template <class real>
void interpolate1(real* fine, int fn, const real* coarse, int cn) {  // N1/MultiGrid1D.cpp:60-77
    (void)cn;
    for (int x = 1; x < fn - 1; x++) {
        const int c = x / 2;
        if (x % 2 == 0) fine[x] = coarse[c];
        else fine[x] = (1 / 2.0f) * (coarse[c] + coarse[c + 1]);
    }
}

template <class real>
void correct1(real* fine, const real* err, int n) {  // N1/MultiGrid1D.cpp:177-188
    for (int x = 1; x < n - 1; x++) fine[x] = fine[x] + err[x];
}

template <class real>
void set1(real* g, int n, real value, bool modify_boundaries) {  // N1/MultiGrid1D.cpp:120-130
    const int lo = modify_boundaries ? 0 : 1;
    for (int x = lo; x < n - lo; x++) g[x] = value;
}

template <class real>
struct MultiGrid1 {  // N1/MultiGrid1D.h:6-31
    std::vector<Grid1<real>> g;
    int numGrids;

    MultiGrid1(int n, const real range[2], int nlevels) {
        const int native = num_grids(n);  // N1/MultiGrid1D.cpp:21-22
        numGrids = nlevels > 0 ? nlevels : native;
        g.resize(numGrids > native ? numGrids : native);
        int cur = n;
        for (size_t l = 0; l < g.size(); l++) {
            grid1_init(g[l], cur, range);
            cur = coarse_size(cur);
        }
    }

    void VCycle(int id, int v1, int v2) {  // N1/MultiGrid1D.cpp:150-175
        Grid1<real>& fine = g[id];
        relax1(fine.v.data(), fine.f.data(), fine.n, fine.h, fine.a, v1);
        if (id != numGrids - 1) {
            Grid1<real>& coarse = g[id + 1];
            std::vector<real> res(fine.n);
            residual1(fine.v.data(), fine.f.data(), res.data(), fine.n, fine.h, fine.a);
            restrict1(res.data(), fine.n, coarse.f.data(), coarse.n);
            set1(coarse.v.data(), coarse.n, (real)0, true);
            VCycle(id + 1, v1, v2);
            std::vector<real> err(fine.n);
            interpolate1(err.data(), fine.n, coarse.v.data(), coarse.n);
            correct1(fine.v.data(), err.data(), fine.n);
        }
        relax1(fine.v.data(), fine.f.data(), fine.n, fine.h, fine.a, v2);
    }

    void FullMultiGridVCycle(int id, int v0, int v1, int v2) {  // N1/MultiGrid1D.cpp:132-148
        Grid1<real>& fine = g[id];
        if (id != numGrids - 1) {
            Grid1<real>& coarse = g[id + 1];
            restrict1(fine.f.data(), fine.n, coarse.f.data(), coarse.n);
            FullMultiGridVCycle(id + 1, v0, v1, v2);
            interpolate1(fine.v.data(), fine.n, coarse.v.data(), coarse.n);
        } else {
            set1(fine.v.data(), fine.n, (real)0, false);
        }
        for (int i = 0; i < v0; i++) VCycle(id, v1, v2);
    }
};

}  // namespace mgo

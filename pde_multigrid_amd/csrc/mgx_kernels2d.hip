// mgx_kernels2d.hip -- 2D Lyapunov multigrid operators for gfx950 (MI355X), fp32 + fp64.
//
// The PDE is K1*V_x + K2*V_y + alfa*V = f with K1 = A00*x + A01*y, K2 = A10*x + A11*y,
// discretised by a 3-point upwind stencil {C, E = x+1, S = y+1}.
// Every red point's E and S neighbours are black and vice versa, so the red-black sweep
// is order-independent within a colour and the parallel result is bit-identical to the
// serial loops.  Expressions keep the reference's association order; this file is
// compiled with -ffp-contract=off.  Layout: idx = x + y*sx.
//
// At BASELINE's 1025^2 the whole hierarchy (about 22 MB fp64) sits in L2 / Infinity Cache:
// this path is launch/latency-bound, not HBM-bound, so the kernels are kept simple
// (one point per thread, coalesced rows) and the sweep loop is launch-minimal.
//
//   relax2d_colour_kernel   MultiGrid2D::Relax               N2/MultiGrid2D.cpp:199-273
//   residual2d_kernel       MultiGrid2D::CalculateResidual   N2/MultiGrid2D.cpp:367-408
//   restrict2d_kernel       MultiGrid2D::Restrict            N2/MultiGrid2D.cpp:63-126
//   interpolate2d_kernel    MultiGrid2D::Interpolate         N2/MultiGrid2D.cpp:128-196
//   correct2d_kernel        MultiGrid2D::ApplyCorrection     N2/MultiGrid2D.cpp:343-366
//   set2d_kernel            MultiGrid2D::setToValue          N2/MultiGrid2D.cpp:275-292
#include "mgx_internal.hpp"

namespace mgx {

template <class real>
struct Lyap2 {  // per-level constants of the 2D operator
    real hx, hy, ax, ay, A0, A1, A2, A3;
    int alfa;
};

template <class real>
__global__ void __launch_bounds__(256) relax2d_colour_kernel(real* __restrict__ v, const real* __restrict__ f, int sx,
                                                             int sy, Lyap2<real> k, int colour) {
    const int y = 1 + blockIdx.y * blockDim.y + threadIdx.y;
    if (y >= sy - 1) return;
    const int p = (colour + y) & 1;  // (x + y) % 2 == colour   N2/MultiGrid2D.cpp:223 / :250
    const int x = 2 * (blockIdx.x * blockDim.x + threadIdx.x) + p;
    if (x < 1 || x >= sx - 1) return;
    const size_t i = x + (size_t)y * sx;
    const real xj = k.ax + x * k.hx;  // :230-231
    const real yi = k.ay + y * k.hy;
    const real K1 = k.A0 * xj + k.A1 * yi;  // :233-234
    const real K2 = k.A2 * xj + k.A3 * yi;
    const real den = K1 * k.hy + K2 * k.hx - k.alfa * k.hx * k.hy;  // :236
    v[i] = (k.hy * K1 * v[i + 1] + k.hx * K2 * v[i + sx] - f[i] * k.hx * k.hy) / (den);  // :241
}

template <class real>
__global__ void __launch_bounds__(256) residual2d_kernel(const real* __restrict__ v, const real* __restrict__ f,
                                                         real* __restrict__ r, int sx, int sy, Lyap2<real> k) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= sx || y >= sy) return;
    const size_t i = x + (size_t)y * sx;
    if (x == 0 || x == sx - 1 || y == 0 || y == sy - 1) {
        r[i] = (real)0;  // :389-392
        return;
    }
    const real xj = k.ax + x * k.hx;
    const real yi = k.ay + y * k.hy;
    const real K1 = k.A0 * xj + k.A1 * yi;
    const real K2 = k.A2 * xj + k.A3 * yi;
    // :403
    r[i] = f[i] - (k.hy * K1 * v[i + 1] + k.hx * K2 * v[i + sx] - v[i] * (k.hy * K1 + k.hx * K2 - k.alfa * k.hx * k.hy)) /
                      (k.hx * k.hy);
}

template <class real>
__global__ void __launch_bounds__(256) restrict2d_kernel(const real* __restrict__ fine, int fx, real* __restrict__ coarse,
                                                         int cx, int cy) {
    const int px = blockIdx.x * blockDim.x + threadIdx.x;
    const int py = blockIdx.y * blockDim.y + threadIdx.y;
    if (px >= cx || py >= cy) return;
    const size_t ci = px + (size_t)py * cx;
    const size_t fi = 2 * px + (size_t)(2 * py) * fx;
    if (px == 0 || px == cx - 1 || py == 0 || py == cy - 1) {
        coarse[ci] = fine[fi];  // :95-101
        return;
    }
    const real C = fine[fi], N = fine[fi - fx], S = fine[fi + fx], E = fine[fi + 1], O = fine[fi - 1];
    const real NE = fine[fi + 1 - fx], NO = fine[fi - 1 - fx], SE = fine[fi + 1 + fx], SO = fine[fi - 1 + fx];
    coarse[ci] = (1 / 16.0f) * (NO + NE + SO + SE + 2 * (O + E + N + S) + 4 * C);  // :123
}

// residual + restrict in one launch (VCycle, N2/MultiGrid2D.cpp:320-323): one thread per coarse point evaluates the
// nine fine residuals it needs (the level is cache resident: recomputing costs less than a second launch and the
// residual array) with the expression of :403 and combines them with the association of :123.
template <class real>
__global__ void __launch_bounds__(256) residual_restrict2d_kernel(const real* __restrict__ v, const real* __restrict__ f, int fx,
                                                                  int fy, Lyap2<real> k, real* __restrict__ coarse, int cx,
                                                                  int cy) {
    const int px = blockIdx.x * blockDim.x + threadIdx.x;
    const int py = blockIdx.y * blockDim.y + threadIdx.y;
    if (px >= cx || py >= cy) return;
    const size_t ci = px + (size_t)py * cx;
    if (px == 0 || px == cx - 1 || py == 0 || py == cy - 1) {
        coarse[ci] = (real)0;  // injection of a boundary residual, which is 0 (:389-392 then :95-101)
        return;
    }
    auto res = [&](int x, int y) -> real {
        if (x == 0 || x == fx - 1 || y == 0 || y == fy - 1) return (real)0;
        const size_t i = x + (size_t)y * fx;
        const real xj = k.ax + x * k.hx;
        const real yi = k.ay + y * k.hy;
        const real K1 = k.A0 * xj + k.A1 * yi;
        const real K2 = k.A2 * xj + k.A3 * yi;
        return f[i] - (k.hy * K1 * v[i + 1] + k.hx * K2 * v[i + fx] - v[i] * (k.hy * K1 + k.hx * K2 - k.alfa * k.hx * k.hy)) /
                          (k.hx * k.hy);
    };
    const int x = 2 * px, y = 2 * py;
    const real C = res(x, y), N = res(x, y - 1), S = res(x, y + 1), E = res(x + 1, y), O = res(x - 1, y);
    const real NE = res(x + 1, y - 1), NO = res(x - 1, y - 1), SE = res(x + 1, y + 1), SO = res(x - 1, y + 1);
    coarse[ci] = (1 / 16.0f) * (NO + NE + SO + SE + 2 * (O + E + N + S) + 4 * C);  // :123
}

template <class real, bool ADD>
__global__ void __launch_bounds__(256) interpolate2d_kernel(real* __restrict__ fine, int fx, int fy,
                                                            const real* __restrict__ coarse, int cx) {
    const int x = 1 + blockIdx.x * blockDim.x + threadIdx.x;
    const int y = 1 + blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= fx - 1 || y >= fy - 1) return;
    const size_t fi = x + (size_t)y * fx;
    const size_t ci = (x >> 1) + (size_t)(y >> 1) * cx;
    const bool ox = x & 1, oy = y & 1;
    real e;
    if (!oy && !ox) e = coarse[ci];                                                                  // :153-156
    else if (oy && !ox) e = (1 / 2.0f) * (coarse[ci] + coarse[ci + cx]);                            // :158-166
    else if (!oy && ox) e = (1 / 2.0f) * (coarse[ci] + coarse[ci + 1]);                             // :169-177
    else e = (1 / 4.0f) * (coarse[ci] + coarse[ci + 1] + coarse[ci + cx] + coarse[ci + cx + 1]);   // :180-192
    if (ADD) fine[fi] = fine[fi] + e;  // :363
    else fine[fi] = e;
}

template <class real>
__global__ void __launch_bounds__(256) correct2d_kernel(real* __restrict__ fine, const real* __restrict__ err, int sx,
                                                        int sy) {
    const int x = 1 + blockIdx.x * blockDim.x + threadIdx.x;
    const int y = 1 + blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= sx - 1 || y >= sy - 1) return;
    const size_t i = x + (size_t)y * sx;
    fine[i] = fine[i] + err[i];
}

template <class real>
__global__ void __launch_bounds__(256) set2d_kernel(real* __restrict__ g, int sx, int sy, real value, int lo) {
    const int x = lo + blockIdx.x * blockDim.x + threadIdx.x;
    const int y = lo + blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= sx - lo || y >= sy - lo) return;
    g[x + (size_t)y * sx] = value;
}

// Levels up to 65^2 (<= 4225 points): all `ncycles` red-black sweeps of a Relax call in ONE workgroup with v and f
// in LDS and a barrier between colour passes (see relax3d_small_kernel); bit-identical to the multi-launch path.
constexpr int SMALL2_MAX = 65;
template <class real>
__global__ void __launch_bounds__(1024) relax2d_small_kernel(real* __restrict__ v, const real* __restrict__ f, int sx, int sy,
                                                             Lyap2<real> k, int ncycles) {
    __shared__ real sv[SMALL2_MAX * SMALL2_MAX];
    __shared__ real sf[SMALL2_MAX * SMALL2_MAX];
    constexpr int PT = (SMALL2_MAX * SMALL2_MAX + 1023) / 1024;
    const int n = sx * sy;
    int kind[PT];
    real hyK1[PT], hxK2[PT], den[PT];  // per-point coefficients (:230-236), constant over the sweeps
#pragma unroll
    for (int p = 0; p < PT; p++) {
        const int t = threadIdx.x + p * 1024;
        kind[p] = -1;
        hyK1[p] = hxK2[p] = den[p] = (real)0;
        if (t < n) {
            const int y = SmallDiv(sx)(t), x = t - y * sx;
            sv[t] = v[t];
            sf[t] = f[t];
            if (x > 0 && x < sx - 1 && y > 0 && y < sy - 1) {
                kind[p] = (x + y) & 1;
                const real xj = k.ax + x * k.hx;
                const real yi = k.ay + y * k.hy;
                const real K1 = k.A0 * xj + k.A1 * yi;
                const real K2 = k.A2 * xj + k.A3 * yi;
                den[p] = K1 * k.hy + K2 * k.hx - k.alfa * k.hx * k.hy;
                hyK1[p] = k.hy * K1;
                hxK2[p] = k.hx * K2;
            }
        }
    }
    __syncthreads();
    for (int c = 0; c < 2 * ncycles; c++) {
        const int colour = c & 1;
#pragma unroll
        for (int p = 0; p < PT; p++)
            if (kind[p] == colour) {
                const int t = threadIdx.x + p * 1024;
                sv[t] = (hyK1[p] * sv[t + 1] + hxK2[p] * sv[t + sx] - sf[t] * k.hx * k.hy) / (den[p]);  // :241
            }
        __syncthreads();
    }
#pragma unroll
    for (int p = 0; p < PT; p++)
        if (kind[p] >= 0) v[threadIdx.x + p * 1024] = sv[threadIdx.x + p * 1024];
}

// weighted Jacobi (addition, see mgx_kernels3d.hip): vout = v + omega*(u - v), u = the Gauss-Seidel value (:241)
template <class real>
__global__ void __launch_bounds__(256) jacobi2d_kernel(const real* __restrict__ v, real* __restrict__ vout,
                                                       const real* __restrict__ f, int sx, int sy, Lyap2<real> k, real omega) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= sx || y >= sy) return;
    const size_t i = x + (size_t)y * sx;
    const real c = v[i];
    if (x == 0 || x == sx - 1 || y == 0 || y == sy - 1) {
        vout[i] = c;
        return;
    }
    const real xj = k.ax + x * k.hx;
    const real yi = k.ay + y * k.hy;
    const real K1 = k.A0 * xj + k.A1 * yi;
    const real K2 = k.A2 * xj + k.A3 * yi;
    const real den = K1 * k.hy + K2 * k.hx - k.alfa * k.hx * k.hy;
    const real u = (k.hy * K1 * v[i + 1] + k.hx * K2 * v[i + sx] - f[i] * k.hx * k.hy) / (den);
    vout[i] = c + omega * (u - c);
}

// sum over the interior of |v - realsol|, realsol = 2*xj*xj-4*xj*yi+2*yi*yi in `real`
// (PrintMeanAbsoluteError, CUDA_TESI/CUDA Lyapunov 2D/Grid2D.cu:123-154; the host divides by the point count)
template <class real>
__global__ void __launch_bounds__(256) abs_error2d_kernel(const real* __restrict__ v, int sx, int sy, real hx, real hy, real ax,
                                                          real ay, double* __restrict__ out) {
    const int y = 1 + blockIdx.y;
    double s = 0;
    for (int x = 1 + threadIdx.x; x < sx - 1; x += blockDim.x) {
        const real xj = ax + x * hx;
        const real yi = ay + y * hy;
        const real realsol = 2 * xj * xj - 4 * xj * yi + 2 * yi * yi;
        const real diff = v[x + (size_t)y * sx] - realsol;
        s += fabs((double)diff);
    }
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    __shared__ double part[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) part[wave] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0;
        for (int w = 0; w < (int)((blockDim.x + 63) >> 6); w++) a += part[w];
        atomicAdd(out, a);
    }
}

// Grid2D::InitV on the device: boundary = 2*xj*xj-4*xj*yi+2*yi*yi, interior 0.           N2/Grid2D.cpp:50-68
// (kernel inventory of the CUDA twin: CUDASetBoundaries, C2/Grid2D.cu:159-182).  Only * and + / - in `real`, evaluated
// left to right as the reference's expression is ((2*xj)*xj - (4*xj)*yi) + (2*yi)*yi, no contraction: bit-identical to the
// host loop without any libm policy.
template <class real>
__global__ void __launch_bounds__(256) init_v2d_kernel(real* __restrict__ v, int sx, int sy, real hx, real hy, real ax, real ay) {
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= sx || y >= sy) return;
    real out = 0;
    if (x == 0 || x == sx - 1 || y == 0 || y == sy - 1) {
        const real yi = ay + y * hy;
        const real xj = ax + x * hx;
        out = 2 * xj * xj - 4 * xj * yi + 2 * yi * yi;  // :61
    }
    v[x + (size_t)y * sx] = out;
}

// =========================================================================== cache-resident cycle kernels
// At BASELINE's 1025^2 the hierarchy (22 MB in fp64) never leaves L2 / Infinity Cache, so the cycle is bound by the number
// of launches and by cache latency, not by HBM: one launch per colour pass per level costs ~60 launches per V(2,2).
// The upwind stencil {C, E = x+1, S = y+1} is ONE-SIDED, which makes temporal blocking cheap: a workgroup that loads a
// T x T tile plus a halo of `npass` points on the +x / +y side only can run all 2*ncycles colour passes of a Relax call on
// the tile in LDS (after pass c the entries closer than c+1 to the tile's high edge are stale, the rest are exactly the
// values whole-grid passes produce: same per-point expression, same colour order -- bit-identical), and the operator
// that follows or precedes the smoother in the cycle works on the same tile:
//   cycle2d_down_kernel  Relax(v1) + CalculateResidual + Restrict      N2/MultiGrid2D.cpp:317-323   (one launch)
//   cycle2d_up_kernel    Interpolate + ApplyCorrection + Relax(v2)     N2/MultiGrid2D.cpp:333-338   (one launch)
//   cycle2d_tail_kernel  the whole V-cycle below 65^2 in ONE workgroup, every level's v and f in LDS   :314-340
// Tiles overlap, so the kernels are out of place (vin -> vout; the host layer swaps the level's two arrays).
// A 7-level V(2,2) at 1025^2 is 9 launches (4 levels x 2 + the tail) instead of ~60.
constexpr int CYC2_MAXPASS = 8;  // colour passes per launch the tile halo is sized for (4 sweeps)

template <class real>
struct Pt2 {  // a point a thread owns: LDS offset (-1 = nothing to do) and its coefficients (N2/MultiGrid2D.cpp:230-236)
    int off;
    real hyK1, hxK2, den;
};

template <class real>
__device__ __forceinline__ Pt2<real> make_pt2(const Lyap2<real>& k, int x, int y, int off) {
    Pt2<real> p;
    const real xj = k.ax + x * k.hx;  // :230-231
    const real yi = k.ay + y * k.hy;
    const real K1 = k.A0 * xj + k.A1 * yi;  // :233-234
    const real K2 = k.A2 * xj + k.A3 * yi;
    p.den = K1 * k.hy + K2 * k.hx - k.alfa * k.hx * k.hy;  // :236
    p.hyK1 = k.hy * K1;
    p.hxK2 = k.hx * K2;
    p.off = off;
    return p;
}

// the points of colour `c` this thread owns in a W x Wy tile whose origin is the global point (gx0, gy0): slot s is the
// s-th of the tile's colour-c points taken NT apart; a point takes part if it is an interior point of the grid and its
// E and S neighbours are inside the tile
template <class real, int NT, int NP>
__device__ __forceinline__ void own_points2(Pt2<real> (&pts)[NP], const Lyap2<real>& k, int c, int W, int Wy, int gx0, int gy0, int sx,
                                            int sy) {
    const int Wh = (W + 1) >> 1;
#pragma unroll
    for (int s = 0; s < NP; s++) {
        const int idx = threadIdx.x + s * NT;
        pts[s].off = -1;
        pts[s].hyK1 = pts[s].hxK2 = pts[s].den = (real)0;
        if (idx < Wh * Wy) {
            const int ty = SmallDiv(Wh)(idx), i = idx - ty * Wh;
            const int tx = 2 * i + ((c + gx0 + gy0 + ty) & 1);
            const int x = gx0 + tx, y = gy0 + ty;
            if (tx + 1 < W && ty + 1 < Wy && x >= 1 && x <= sx - 2 && y >= 1 && y <= sy - 2) pts[s] = make_pt2<real>(k, x, y, ty * W + tx);
        }
    }
}

template <class real, int NP>
__device__ __forceinline__ void relax_owned2(real* sv, const real* sf, const Pt2<real> (&pts)[NP], int W, real hx, real hy) {
#pragma unroll
    for (int s = 0; s < NP; s++)
        if (pts[s].off >= 0) {
            const int t = pts[s].off;
            sv[t] = (pts[s].hyK1 * sv[t + 1] + pts[s].hxK2 * sv[t + W] - sf[t] * hx * hy) / (pts[s].den);  // :241
        }
}

template <class real, int T, int NT, int NP>
__global__ void __launch_bounds__(NT) cycle2d_down_kernel(const real* __restrict__ vin, real* __restrict__ vout,
                                                          const real* __restrict__ f, int sx, int sy, Lyap2<real> k, int npass,
                                                          int v_zero, real* __restrict__ coarse, int cx, int cy) {
    extern __shared__ __align__(16) unsigned char smem2[];
    // tile = core [X0, X0+T) + 1 point below (the restriction reads residuals at 2p-1) + 2 + npass above (the residual at
    // X0+T reads v at X0+T+1, and npass passes eat npass points of the high side)
    const int W = T + 3 + npass;
    real* sv = (real*)smem2;
    real* sf = sv + W * W;
    const int X0 = blockIdx.x * T, Y0 = blockIdx.y * T, gx0 = X0 - 1, gy0 = Y0 - 1;
    {   // all loads of the thread are issued before the first one is waited for (the tile comes from L2 / Infinity Cache)
        constexpr int WM = T + 3 + CYC2_MAXPASS, NL = (WM * WM + NT - 1) / NT;
        real a[NL], b[NL];
#pragma unroll
        for (int s = 0; s < NL; s++) {
            const int t = threadIdx.x + s * NT;
            const int ty = SmallDiv(W)(t), tx = t - ty * W, x = gx0 + tx, y = gy0 + ty;
            a[s] = b[s] = (real)0;
            if (t < W * W && x >= 0 && x < sx && y >= 0 && y < sy) {
                const size_t i = x + (size_t)y * sx;
                if (!v_zero) a[s] = vin[i];
                if (x > 0 && x < sx - 1 && y > 0 && y < sy - 1) b[s] = f[i];  // f of a boundary point is never read; 0 = its residual
            }
        }
#pragma unroll
        for (int s = 0; s < NL; s++) {
            const int t = threadIdx.x + s * NT;
            if (t < W * W) {
                sv[t] = a[s];
                sf[t] = b[s];
            }
        }
    }
    Pt2<real> red[NP], black[NP];
    own_points2<real, NT, NP>(red, k, 0, W, W, gx0, gy0, sx, sy);
    own_points2<real, NT, NP>(black, k, 1, W, W, gx0, gy0, sx, sy);
    __syncthreads();
    for (int sweep = 0; sweep < npass / 2; sweep++) {  // red = (x + y) % 2 == 0 first (:223), then black (:250)
        relax_owned2<real, NP>(sv, sf, red, W, k.hx, k.hy);
        __syncthreads();
        relax_owned2<real, NP>(sv, sf, black, W, k.hx, k.hy);
        __syncthreads();
    }
    // the smoothed core; the last tile of a row / column also owns the boundary point sx-1 / sy-1
    const int xe = blockIdx.x == gridDim.x - 1 ? sx : X0 + T, ye = blockIdx.y == gridDim.y - 1 ? sy : Y0 + T;
    const int cw = xe - X0, ch = ye - Y0;
    for (int t = threadIdx.x; t < cw * ch; t += NT) {
        const int ly = SmallDiv(cw)(t), lx = t - ly * cw;
        vout[(X0 + lx) + (size_t)(Y0 + ly) * sx] = sv[(ly + 1) * W + lx + 1];
    }
    if (!coarse) return;
    // residual of the points [X0-1, X0+T]^2 in place of their f (:403); boundary points keep their 0 (:389-392)
    const real hxhy_alfa = k.alfa * k.hx * k.hy;
#pragma unroll
    for (int s = 0; s < NP; s++) {
        if (red[s].off >= 0) {
            const int t = red[s].off;
            sf[t] = sf[t] - (red[s].hyK1 * sv[t + 1] + red[s].hxK2 * sv[t + W] - sv[t] * (red[s].hyK1 + red[s].hxK2 - hxhy_alfa)) / (k.hx * k.hy);
        }
        if (black[s].off >= 0) {
            const int t = black[s].off;
            sf[t] = sf[t] - (black[s].hyK1 * sv[t + 1] + black[s].hxK2 * sv[t + W] - sv[t] * (black[s].hyK1 + black[s].hxK2 - hxhy_alfa)) / (k.hx * k.hy);
        }
    }
    __syncthreads();
    const int px0 = X0 >> 1, py0 = Y0 >> 1;
    const int pw = blockIdx.x == gridDim.x - 1 ? cx - px0 : T / 2, ph = blockIdx.y == gridDim.y - 1 ? cy - py0 : T / 2;
    for (int t = threadIdx.x; t < pw * ph; t += NT) {
        const int ly = SmallDiv(pw)(t), lx = t - ly * pw;
        const int px = px0 + lx, py = py0 + ly;
        real out = (real)0;  // boundary coarse point: injection of a boundary residual, which is 0 (:389-392 then :95-101)
        if (px > 0 && px < cx - 1 && py > 0 && py < cy - 1) {
            const real* c = sf + (2 * ly + 1) * W + 2 * lx + 1;
            const real C = c[0], N = c[-W], S = c[W], E = c[1], O = c[-1];
            const real NE = c[1 - W], NO = c[-1 - W], SE = c[1 + W], SO = c[-1 + W];
            out = (1 / 16.0f) * (NO + NE + SO + SE + 2 * (O + E + N + S) + 4 * C);  // :123
        }
        coarse[px + (size_t)py * cx] = out;
    }
}

template <class real, int T, int NT, int NP>
__global__ void __launch_bounds__(NT) cycle2d_up_kernel(const real* __restrict__ vin, real* __restrict__ vout,
                                                        const real* __restrict__ f, int sx, int sy, Lyap2<real> k, int npass,
                                                        const real* __restrict__ coarse, int cx, int cy) {
    extern __shared__ __align__(16) unsigned char smem2[];
    const int W = T + 1 + npass;        // core [X0, X0+T] + npass points above
    const int Wc = (W - 1) / 2 + 2;     // coarse points under the tile
    real* sv = (real*)smem2;
    real* sf = sv + W * W;
    real* sc = sf + W * W;
    const int X0 = blockIdx.x * T, Y0 = blockIdx.y * T, px0 = X0 >> 1, py0 = Y0 >> 1;
    {
        constexpr int WM = T + 1 + CYC2_MAXPASS, NL = (WM * WM + NT - 1) / NT, WCM = (WM - 1) / 2 + 2, NLC = (WCM * WCM + NT - 1) / NT;
        real a[NL], b[NL], cc[NLC];
#pragma unroll
        for (int s = 0; s < NLC; s++) {
            const int t = threadIdx.x + s * NT;
            const int ty = SmallDiv(Wc)(t), tx = t - ty * Wc, px = px0 + tx, py = py0 + ty;
            cc[s] = (t < Wc * Wc && px < cx && py < cy) ? coarse[px + (size_t)py * cx] : (real)0;
        }
#pragma unroll
        for (int s = 0; s < NL; s++) {
            const int t = threadIdx.x + s * NT;
            const int ty = SmallDiv(W)(t), tx = t - ty * W, x = X0 + tx, y = Y0 + ty;
            a[s] = b[s] = (real)0;
            if (t < W * W && x < sx && y < sy) {
                const size_t i = x + (size_t)y * sx;
                a[s] = vin[i];
                if (x > 0 && x < sx - 1 && y > 0 && y < sy - 1) b[s] = f[i];
            }
        }
#pragma unroll
        for (int s = 0; s < NLC; s++) {
            const int t = threadIdx.x + s * NT;
            if (t < Wc * Wc) sc[t] = cc[s];
        }
#pragma unroll
        for (int s = 0; s < NL; s++) {
            const int t = threadIdx.x + s * NT;
            if (t < W * W) {
                sv[t] = a[s];
                sf[t] = b[s];
            }
        }
    }
    Pt2<real> red[NP], black[NP];
    own_points2<real, NT, NP>(red, k, 0, W, W, X0, Y0, sx, sy);
    own_points2<real, NT, NP>(black, k, 1, W, W, X0, Y0, sx, sy);
    __syncthreads();
    // v += Interpolate(coarse) on the interior points of the tile (N2/MultiGrid2D.cpp:153-192 then :363)
    for (int t = threadIdx.x; t < W * W; t += NT) {
        const int ty = SmallDiv(W)(t), tx = t - ty * W, x = X0 + tx, y = Y0 + ty;
        if (x >= 1 && x <= sx - 2 && y >= 1 && y <= sy - 2) {
            const real* c = sc + ((y >> 1) - py0) * Wc + ((x >> 1) - px0);
            const bool ox = x & 1, oy = y & 1;
            real e;
            if (!oy && !ox) e = c[0];                                              // :153-156
            else if (oy && !ox) e = (1 / 2.0f) * (c[0] + c[Wc]);                   // :158-166
            else if (!oy && ox) e = (1 / 2.0f) * (c[0] + c[1]);                    // :169-177
            else e = (1 / 4.0f) * (c[0] + c[1] + c[Wc] + c[Wc + 1]);               // :180-192
            sv[t] = sv[t] + e;
        }
    }
    __syncthreads();
    for (int sweep = 0; sweep < npass / 2; sweep++) {
        relax_owned2<real, NP>(sv, sf, red, W, k.hx, k.hy);
        __syncthreads();
        relax_owned2<real, NP>(sv, sf, black, W, k.hx, k.hy);
        __syncthreads();
    }
    const int xe = blockIdx.x == gridDim.x - 1 ? sx : X0 + T, ye = blockIdx.y == gridDim.y - 1 ? sy : Y0 + T;
    const int cw = xe - X0, ch = ye - Y0;
    for (int t = threadIdx.x; t < cw * ch; t += NT) {
        const int ly = SmallDiv(cw)(t), lx = t - ly * cw;
        vout[(X0 + lx) + (size_t)(Y0 + ly) * sx] = sv[ly * W + lx];
    }
}

// ---- the whole cycle below 65^2 in one workgroup ---------------------------------------------------------------------
constexpr int TAIL2_MAXLEV = 8;
template <class real>
struct Tail2 {
    int nlev;
    int sx[TAIL2_MAXLEV], sy[TAIL2_MAXLEV];
    real* v[TAIL2_MAXLEV];
    real* f[TAIL2_MAXLEV];
    real hx[TAIL2_MAXLEV], hy[TAIL2_MAXLEV];
};

constexpr int TAIL2_PT = 5;  // points per thread of the largest tail level (65^2 = 4225 <= 5 x 1024)

// the points thread t owns on an sx x sy level held in LDS (t, t + 1024, ...): colour of an interior point or -1
template <class real>
__device__ __forceinline__ void tail_points2(Pt2<real> (&pts)[TAIL2_PT], int (&kind)[TAIL2_PT], int sx, int sy, const Lyap2<real>& k) {
    const SmallDiv dsx(sx);
#pragma unroll
    for (int s = 0; s < TAIL2_PT; s++) {
        const int t = threadIdx.x + s * 1024;
        kind[s] = -1;
        pts[s].off = t;
        pts[s].hyK1 = pts[s].hxK2 = pts[s].den = (real)0;
        if (t < sx * sy) {
            const int y = dsx(t), x = t - y * sx;
            if (x > 0 && x < sx - 1 && y > 0 && y < sy - 1) {
                kind[s] = (x + y) & 1;
                pts[s] = make_pt2<real>(k, x, y, t);
            }
        }
    }
}

template <class real>
__device__ __forceinline__ void tail_relax2(real* sv, const real* sf, int sx, const Pt2<real> (&pts)[TAIL2_PT],
                                            const int (&kind)[TAIL2_PT], const Lyap2<real>& k, int ncycles) {
    for (int c = 0; c < 2 * ncycles; c++) {
        const int colour = c & 1;  // red = (x + y) % 2 == 0 first (:223), then black (:250)
#pragma unroll
        for (int s = 0; s < TAIL2_PT; s++)
            if (kind[s] == colour) {
                const int t = pts[s].off;
                sv[t] = (pts[s].hyK1 * sv[t + 1] + pts[s].hxK2 * sv[t + sx] - sf[t] * k.hx * k.hy) / (pts[s].den);  // :241
            }
        __syncthreads();
    }
}

template <class real>
__global__ void __launch_bounds__(1024) cycle2d_tail_kernel(Tail2<real> L, Lyap2<real> k0, int v1, int v2, int top_zero) {
    extern __shared__ __align__(16) unsigned char smem2[];
    real* base = (real*)smem2;
    int offv[TAIL2_MAXLEV], offf[TAIL2_MAXLEV];
    int o = 0;
#pragma unroll
    for (int l = 0; l < TAIL2_MAXLEV; l++) {
        offv[l] = o;
        if (l < L.nlev) o += L.sx[l] * L.sy[l];
        offf[l] = o;
        if (l < L.nlev) o += L.sx[l] * L.sy[l];
    }
    real* sr = base + o;  // residual scratch, as large as the top level
    {   // level 0 of the tail: v as it stands (or 0 when the caller knows it is the zeroed error of a coarse level), f
        const int n = L.sx[0] * L.sy[0];
        for (int t = threadIdx.x; t < n; t += 1024) {
            base[offv[0] + t] = top_zero ? (real)0 : L.v[0][t];
            base[offf[0] + t] = L.f[0][t];
        }
    }
    __syncthreads();
    const int last = L.nlev - 1;
    Pt2<real> pts[TAIL2_PT];
    int kind[TAIL2_PT];
    for (int l = 0; l <= last; l++) {  // MultiGrid2D::VCycle, way down                      N2/MultiGrid2D.cpp:317-328
        Lyap2<real> k = k0;
        k.hx = L.hx[l];
        k.hy = L.hy[l];
        real* sv = base + offv[l];
        real* sf = base + offf[l];
        const int sx = L.sx[l], sy = L.sy[l];
        tail_points2<real>(pts, kind, sx, sy, k);
        tail_relax2<real>(sv, sf, sx, pts, kind, k, v1);  // :317
        if (l == last) {
            tail_relax2<real>(sv, sf, sx, pts, kind, k, v2);  // :338 on the coarsest level
            break;
        }
        // residual (:403; 0 on the boundary, :389-392) into the scratch, then full weighting (:123)
        const real hxhy_alfa = k.alfa * k.hx * k.hy;
#pragma unroll
        for (int s = 0; s < TAIL2_PT; s++) {
            const int t = pts[s].off;
            if (t < sx * sy) {
                real r = (real)0;
                if (kind[s] >= 0)
                    r = sf[t] - (pts[s].hyK1 * sv[t + 1] + pts[s].hxK2 * sv[t + sx] - sv[t] * (pts[s].hyK1 + pts[s].hxK2 - hxhy_alfa)) / (k.hx * k.hy);
                sr[t] = r;
            }
        }
        __syncthreads();
        const int cx = L.sx[l + 1], cy = L.sy[l + 1];
        real* cv = base + offv[l + 1];
        real* cf = base + offf[l + 1];
        const SmallDiv dcx(cx);
        for (int t = threadIdx.x; t < cx * cy; t += 1024) {
            const int py = dcx(t), px = t - py * cx;
            real out = (real)0;  // boundary: injection of a boundary residual, which is 0 (:95-101)
            if (px > 0 && px < cx - 1 && py > 0 && py < cy - 1) {
                const real* c = sr + 2 * px + 2 * py * sx;
                const real C = c[0], N = c[-sx], S = c[sx], E = c[1], O = c[-1];
                const real NE = c[1 - sx], NO = c[-1 - sx], SE = c[1 + sx], SO = c[-1 + sx];
                out = (1 / 16.0f) * (NO + NE + SO + SE + 2 * (O + E + N + S) + 4 * C);  // :123
            }
            cf[t] = out;       // :320-323
            cv[t] = (real)0;   // :326
        }
        __syncthreads();
    }
    for (int l = last - 1; l >= 0; l--) {  // way up                                          N2/MultiGrid2D.cpp:333-338
        Lyap2<real> k = k0;
        k.hx = L.hx[l];
        k.hy = L.hy[l];
        real* sv = base + offv[l];
        real* sf = base + offf[l];
        const real* c = base + offv[l + 1];
        const int sx = L.sx[l], sy = L.sy[l], cx = L.sx[l + 1];
        tail_points2<real>(pts, kind, sx, sy, k);
        const SmallDiv dsx(sx);
#pragma unroll
        for (int s = 0; s < TAIL2_PT; s++)
            if (kind[s] >= 0) {
                const int t = pts[s].off;
                const int y = dsx(t), x = t - y * sx;
                const int ci = (x >> 1) + (y >> 1) * cx;
                const bool ox = x & 1, oy = y & 1;
                real e;
                if (!oy && !ox) e = c[ci];                                                   // :153-156
                else if (oy && !ox) e = (1 / 2.0f) * (c[ci] + c[ci + cx]);                   // :158-166
                else if (!oy && ox) e = (1 / 2.0f) * (c[ci] + c[ci + 1]);                    // :169-177
                else e = (1 / 4.0f) * (c[ci] + c[ci + 1] + c[ci + cx] + c[ci + cx + 1]);     // :180-192
                sv[t] = sv[t] + e;  // :363
            }
        __syncthreads();
        tail_relax2<real>(sv, sf, sx, pts, kind, k, v2);  // :338
    }
    // what the launch-per-operator path leaves in the level arrays: v of every level, the restricted residual in f below the top
    for (int l = 0; l <= last; l++) {
        const int n = L.sx[l] * L.sy[l];
        for (int t = threadIdx.x; t < n; t += 1024) {
            L.v[l][t] = base[offv[l] + t];
            if (l > 0) L.f[l][t] = base[offf[l] + t];
        }
    }
}

// =========================================================================== host side
static inline dim3 blk2() { return dim3(64, 4, 1); }
static inline dim3 grd2(int nx, int ny) { return dim3(ceil_div(nx, 64), ceil_div(ny, 4), 1); }

static int check_n2(const int n[2], const char* what) {
    MGX_REQUIRE(n, MGX_ERR_INVALID, "%s: size array is NULL", what);
    for (int d = 0; d < 2; d++)
        MGX_REQUIRE(valid_size(n[d]), MGX_ERR_SIZE, "%s: size[%d] = %d is not 2^k+1 >= 3", what, d, n[d]);
    return MGX_OK;
}

static int check_coarse2(const int fn[2], const int cn[2], const char* what) {
    MGX_REQUIRE(fn && cn, MGX_ERR_INVALID, "%s: size array is NULL", what);
    for (int d = 0; d < 2; d++)  // N2/MultiGrid2D.cpp:71-72
        MGX_REQUIRE(cn[d] == (fn[d] - 1) / 2 + 1, MGX_ERR_SIZE, "%s: coarse size[%d] = %d != (%d-1)/2+1", what, d, cn[d],
                    fn[d]);
    return MGX_OK;
}

template <class real>
static Lyap2<real> lyap(const real h[2], const real a[2], const real A[4], int alfa) {
    Lyap2<real> k;
    k.hx = h[0]; k.hy = h[1]; k.ax = a[0]; k.ay = a[1];
    k.A0 = A[0]; k.A1 = A[1]; k.A2 = A[2]; k.A3 = A[3];
    k.alfa = alfa;
    return k;
}

template <class real>
int relax2d(mgx_ctx* ctx, real* v, const real* f, const int n[2], const real h[2], const real a[2], const real A[4],
            int alfa, int ncycles) {
    MGX_REQUIRE(ctx && v && f && h && a && A, MGX_ERR_INVALID, "relax2d: NULL argument");
    MGX_USE(ctx);
    int st = check_n2(n, "relax2d");
    if (st) return st;
    MGX_REQUIRE(ncycles >= 0, MGX_ERR_INVALID, "relax2d: ncycles = %d < 0", ncycles);
    const Lyap2<real> k = lyap<real>(h, a, A, alfa);
    if (ncycles > 0 && n[0] <= SMALL2_MAX && n[1] <= SMALL2_MAX) {
        MGX_LAUNCH((relax2d_small_kernel<real>), dim3(1), dim3(1024), 0, ctx->compute, v, f, n[0], n[1], k, ncycles);
        MGX_LAUNCH_CHECK();
        return MGX_OK;
    }
    dim3 g(ceil_div((n[0] + 1) / 2, 64), ceil_div(n[1] - 2, 4), 1);
    for (int c = 0; c < ncycles; c++)
        for (int colour = 0; colour < 2; colour++)
            MGX_LAUNCH((relax2d_colour_kernel<real>), g, blk2(), 0, ctx->compute, v, f, n[0], n[1], k, colour);
    MGX_LAUNCH_CHECK();
    return MGX_OK;
}

template <class real>
int residual2d(mgx_ctx* ctx, const real* v, const real* f, real* r, const int n[2], const real h[2], const real a[2],
               const real A[4], int alfa) {
    MGX_REQUIRE(ctx && v && f && r && h && a && A, MGX_ERR_INVALID, "residual2d: NULL argument");
    MGX_USE(ctx);
    int st = check_n2(n, "residual2d");
    if (st) return st;
    MGX_LAUNCH((residual2d_kernel<real>), grd2(n[0], n[1]), blk2(), 0, ctx->compute, v, f, r, n[0], n[1],
                       lyap<real>(h, a, A, alfa));
    MGX_LAUNCH_CHECK();
    return MGX_OK;
}

template <class real>
int restrict2d(mgx_ctx* ctx, const real* fine, const int fn[2], real* coarse, const int cn[2]) {
    MGX_REQUIRE(ctx && fine && coarse, MGX_ERR_INVALID, "restrict2d: NULL argument");
    MGX_USE(ctx);
    int st = check_n2(fn, "restrict2d");
    if (st) return st;
    st = check_coarse2(fn, cn, "restrict2d");
    if (st) return st;
    MGX_LAUNCH((restrict2d_kernel<real>), grd2(cn[0], cn[1]), blk2(), 0, ctx->compute, fine, fn[0], coarse, cn[0],
                       cn[1]);
    MGX_LAUNCH_CHECK();
    return MGX_OK;
}

template <class real, bool ADD>
int interpolate2d(mgx_ctx* ctx, real* fine, const int fn[2], const real* coarse, const int cn[2]) {
    MGX_REQUIRE(ctx && fine && coarse, MGX_ERR_INVALID, "interpolate2d: NULL argument");
    MGX_USE(ctx);
    int st = check_n2(fn, "interpolate2d");
    if (st) return st;
    st = check_coarse2(fn, cn, "interpolate2d");
    if (st) return st;
    MGX_LAUNCH((interpolate2d_kernel<real, ADD>), grd2(fn[0] - 2, fn[1] - 2), blk2(), 0, ctx->compute, fine, fn[0],
                       fn[1], coarse, cn[0]);
    MGX_LAUNCH_CHECK();
    return MGX_OK;
}

template <class real>
int residual_restrict2d(mgx_ctx* ctx, const real* v, const real* f, const int n[2], const real h[2], const real a[2],
                        const real A[4], int alfa, real* coarse_f, const int cn[2]) {
    MGX_REQUIRE(ctx && v && f && h && a && A && coarse_f, MGX_ERR_INVALID, "residual_restrict2d: NULL argument");
    MGX_USE(ctx);
    int st = check_n2(n, "residual_restrict2d");
    if (st) return st;
    st = check_coarse2(n, cn, "residual_restrict2d");
    if (st) return st;
    MGX_LAUNCH((residual_restrict2d_kernel<real>), grd2(cn[0], cn[1]), blk2(), 0, ctx->compute, v, f, n[0], n[1],
                       lyap<real>(h, a, A, alfa), coarse_f, cn[0], cn[1]);
    MGX_LAUNCH_CHECK();
    return MGX_OK;
}

template <class real>
int correct2d(mgx_ctx* ctx, real* fine, const int fn[2], const real* err, const int en[2]) {
    MGX_REQUIRE(ctx && fine && err && en, MGX_ERR_INVALID, "apply_correction2d: NULL argument");
    MGX_USE(ctx);
    int st = check_n2(fn, "apply_correction2d");
    if (st) return st;
    for (int d = 0; d < 2; d++)  // N2/MultiGrid2D.cpp:351-352
        MGX_REQUIRE(fn[d] == en[d], MGX_ERR_SIZE, "apply_correction2d: size[%d] %d != %d", d, fn[d], en[d]);
    MGX_LAUNCH((correct2d_kernel<real>), grd2(fn[0] - 2, fn[1] - 2), blk2(), 0, ctx->compute, fine, err, fn[0],
                       fn[1]);
    MGX_LAUNCH_CHECK();
    return MGX_OK;
}

template <class real>
int set2d(mgx_ctx* ctx, real* g, const int n[2], real value, int modify_boundaries) {
    MGX_REQUIRE(ctx && g, MGX_ERR_INVALID, "set2d: NULL argument");
    MGX_USE(ctx);
    int st = check_n2(n, "set2d");
    if (st) return st;
    const int lo = modify_boundaries ? 0 : 1;
    if (modify_boundaries && value == (real)0 && !std::signbit(value)) {  // the cycle's "coarse v := 0" (:326)
        return fill_zero(ctx, g, (size_t)n[0] * n[1] * sizeof(real));
    }
    MGX_LAUNCH((set2d_kernel<real>), grd2(n[0] - 2 * lo, n[1] - 2 * lo), blk2(), 0, ctx->compute, g, n[0], n[1],
                       value, lo);
    MGX_LAUNCH_CHECK();
    return MGX_OK;
}

template <class real>
int jacobi2d(mgx_ctx* ctx, real* v, real* tmp, const real* f, const int n[2], const real h[2], const real a[2], const real A[4],
             int alfa, real omega, int ncycles) {
    MGX_REQUIRE(ctx && v && tmp && f && h && a && A && v != tmp, MGX_ERR_INVALID, "jacobi2d: NULL or aliased argument");
    MGX_USE(ctx);
    int st = check_n2(n, "jacobi2d");
    if (st) return st;
    MGX_REQUIRE(ncycles >= 0, MGX_ERR_INVALID, "jacobi2d: ncycles = %d < 0", ncycles);
    const Lyap2<real> k = lyap<real>(h, a, A, alfa);
    real *src = v, *dst = tmp;
    for (int c = 0; c < ncycles; c++) {
        MGX_LAUNCH((jacobi2d_kernel<real>), grd2(n[0], n[1]), blk2(), 0, ctx->compute, (const real*)src, dst, f, n[0], n[1],
                           k, omega);
        real* t = src; src = dst; dst = t;
    }
    MGX_LAUNCH_CHECK();
    if (src != v) MGX_HIP(hipMemcpyAsync(v, src, sizeof(real) * (size_t)n[0] * n[1], hipMemcpyDeviceToDevice, ctx->compute));
    return MGX_OK;
}

template <class real>
int mean_abs_error2d(mgx_ctx* ctx, const real* v, const int n[2], const real h[2], const real a[2], double* host_mean) {
    MGX_REQUIRE(ctx && v && h && a && host_mean, MGX_ERR_INVALID, "mean_abs_error2d: NULL argument");
    MGX_USE(ctx);
    int st = check_n2(n, "mean_abs_error2d");
    if (st) return st;
    void* ws = nullptr;
    st = workspace(ctx, sizeof(double), &ws);
    if (st) return st;
    MGX_HIP(hipMemsetAsync(ws, 0, sizeof(double), ctx->compute));
    if (n[0] > 2 && n[1] > 2) {
        MGX_LAUNCH((abs_error2d_kernel<real>), dim3(1, n[1] - 2), dim3(n[0] >= 256 ? 256 : 64), 0, ctx->compute, v, n[0],
                           n[1], h[0], h[1], a[0], a[1], (double*)ws);
        MGX_LAUNCH_CHECK();
    }
    double total = 0;
    MGX_HIP(hipMemcpyAsync(&total, ws, sizeof(double), hipMemcpyDeviceToHost, ctx->compute));
    MGX_HIP(hipStreamSynchronize(ctx->compute));
    const double cnt = (double)(n[0] - 2) * (double)(n[1] - 2);
    *host_mean = cnt > 0 ? total / cnt : 0.0;
    return MGX_OK;
}

template <class real>
int init_v2d(mgx_ctx* ctx, real* v, const int n[2], const real h[2], const real a[2]) {
    MGX_REQUIRE(ctx && v && h && a, MGX_ERR_INVALID, "init_v2d: NULL argument");
    MGX_USE(ctx);
    int st = check_n2(n, "init_v2d");
    if (st) return st;
    MGX_LAUNCH((init_v2d_kernel<real>), dim3(ceil_div(n[0], 64), ceil_div(n[1], 4)), dim3(64, 4, 1), 0, ctx->compute, v, n[0], n[1],
                       h[0], h[1], a[0], a[1]);
    MGX_LAUNCH_CHECK();
    return MGX_OK;
}

// ---- host side of the cache-resident cycle kernels ------------------------------------------------------------------
template <class K>
static int allow_lds(K kernel, size_t bytes) {  // dynamic LDS beyond the 64 KB default needs the attribute (160 KB per CU)
    if (bytes > 64 * 1024) MGX_HIP(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return MGX_OK;
}

// tile edge for an sx x sy level.  Measured on MI355X (tools/time_2d.py, profiles/r02_2d_*.txt): 64-point tiles (one
// 1024-thread workgroup per CU: 256 tiles) at 1025^2, 32-point tiles (512 threads, several workgroups per CU) above,
// 16-point tiles (256 threads) below
static int cyc2_tile(const mgx_ctx* ctx, const int n[2], size_t elem) {
    if (ctx->cyc2_tile == 16 || ctx->cyc2_tile == 32 || ctx->cyc2_tile == 64) return ctx->cyc2_tile;  // "cycle2d.tile"
    const int m = n[0] > n[1] ? n[0] : n[1];
    if (elem == 8 && m > 1025) return 32;  // fp64 above 1025^2: 32-point tiles of 512 threads; fp32 keeps 64
    return m > 513 ? 64 : 16;
}

template <class real>
int cycle2d_down(mgx_ctx* ctx, const real* vin, real* vout, const real* f, const int n[2], const real h[2], const real a[2],
                 const real A[4], int alfa, int ncycles, int v_zero, real* coarse_f, const int cn[2]) {
    MGX_REQUIRE(ctx && vout && f && h && a && A && (vin || v_zero), MGX_ERR_INVALID, "relax_residual_restrict2d: NULL argument");
    MGX_USE(ctx);
    MGX_REQUIRE(vin != vout, MGX_ERR_INVALID, "relax_residual_restrict2d: v_in and v_out must differ (tiles overlap)");
    int st = check_n2(n, "relax_residual_restrict2d");
    if (st) return st;
    if (coarse_f) {
        st = check_coarse2(n, cn, "relax_residual_restrict2d");
        if (st) return st;
    }
    MGX_REQUIRE(ncycles >= 0 && 2 * ncycles <= CYC2_MAXPASS, MGX_ERR_INVALID, "relax_residual_restrict2d: 0 <= ncycles <= %d", CYC2_MAXPASS / 2);
    const Lyap2<real> k = lyap<real>(h, a, A, alfa);
    const int npass = 2 * ncycles, T = cyc2_tile(ctx, n, sizeof(real)), W = T + 3 + npass;
    const size_t lds = (size_t)2 * W * W * sizeof(real);
    const dim3 g(max(1, ceil_div(n[0] - 1, T)), max(1, ceil_div(n[1] - 1, T)));
    const int cx = coarse_f ? cn[0] : 0, cy = coarse_f ? cn[1] : 0;
#define MGX_CYC_DOWN(TT, NT, NP)                                                                                              \
    do {                                                                                                                      \
        MGX_TRY_RET(allow_lds(cycle2d_down_kernel<real, TT, NT, NP>, lds));                                                   \
        MGX_LAUNCH((cycle2d_down_kernel<real, TT, NT, NP>), g, dim3(NT), lds, ctx->compute, vin, vout, f, n[0], n[1], k, \
                           npass, v_zero, coarse_f, cx, cy);                                                                  \
    } while (0)
    if (T == 64) MGX_CYC_DOWN(64, 1024, 3);        // 38 x 75 = 2850 points of a colour / 1024 threads
    else if (T == 32) MGX_CYC_DOWN(32, 512, 2);    // 22 x 43 = 946 / 512
    else MGX_CYC_DOWN(16, 256, 2);                 // 14 x 27 = 378 / 256
#undef MGX_CYC_DOWN
    MGX_LAUNCH_CHECK();
    return MGX_OK;
}

template <class real>
int cycle2d_up(mgx_ctx* ctx, const real* vin, real* vout, const real* f, const int n[2], const real h[2], const real a[2],
               const real A[4], int alfa, const real* coarse_v, const int cn[2], int ncycles) {
    MGX_REQUIRE(ctx && vin && vout && f && h && a && A && coarse_v, MGX_ERR_INVALID, "interpolate_correct_relax2d: NULL argument");
    MGX_USE(ctx);
    MGX_REQUIRE(vin != vout, MGX_ERR_INVALID, "interpolate_correct_relax2d: v_in and v_out must differ (tiles overlap)");
    int st = check_n2(n, "interpolate_correct_relax2d");
    if (st) return st;
    st = check_coarse2(n, cn, "interpolate_correct_relax2d");
    if (st) return st;
    MGX_REQUIRE(ncycles >= 0 && 2 * ncycles <= CYC2_MAXPASS, MGX_ERR_INVALID, "interpolate_correct_relax2d: 0 <= ncycles <= %d", CYC2_MAXPASS / 2);
    const Lyap2<real> k = lyap<real>(h, a, A, alfa);
    const int npass = 2 * ncycles, T = cyc2_tile(ctx, n, sizeof(real)), W = T + 1 + npass, Wc = (W - 1) / 2 + 2;
    const size_t lds = ((size_t)2 * W * W + (size_t)Wc * Wc) * sizeof(real);
    const dim3 g(max(1, ceil_div(n[0] - 1, T)), max(1, ceil_div(n[1] - 1, T)));
#define MGX_CYC_UP(TT, NT, NP)                                                                                              \
    do {                                                                                                                    \
        MGX_TRY_RET(allow_lds(cycle2d_up_kernel<real, TT, NT, NP>, lds));                                                   \
        MGX_LAUNCH((cycle2d_up_kernel<real, TT, NT, NP>), g, dim3(NT), lds, ctx->compute, vin, vout, f, n[0], n[1], k, \
                           npass, coarse_v, cn[0], cn[1]);                                                                  \
    } while (0)
    if (T == 64) MGX_CYC_UP(64, 1024, 3);
    else if (T == 32) MGX_CYC_UP(32, 512, 2);
    else MGX_CYC_UP(16, 256, 2);
#undef MGX_CYC_UP
    MGX_LAUNCH_CHECK();
    return MGX_OK;
}

// levels[0 .. nlev) of a hierarchy, finest of the tail first; n = {sx0, sy0, sx1, sy1, ...}, h likewise
template <class real>
int cycle2d_tail(mgx_ctx* ctx, int nlev, real* const* v, real* const* f, const int* n, const real* h, const real a[2],
                 const real A[4], int alfa, int v1, int v2, int top_zero) {
    MGX_REQUIRE(ctx && v && f && n && h && a && A, MGX_ERR_INVALID, "vcycle_tail2d: NULL argument");
    MGX_USE(ctx);
    MGX_REQUIRE(nlev >= 1 && nlev <= TAIL2_MAXLEV, MGX_ERR_INVALID, "vcycle_tail2d: 1 <= nlev <= %d", TAIL2_MAXLEV);
    MGX_REQUIRE(v1 >= 0 && v2 >= 0, MGX_ERR_INVALID, "vcycle_tail2d: negative sweep count");
    Tail2<real> L;
    memset(&L, 0, sizeof L);
    L.nlev = nlev;
    size_t elems = 0;
    for (int l = 0; l < nlev; l++) {
        const int nl[2] = {n[2 * l], n[2 * l + 1]};
        int st = check_n2(nl, "vcycle_tail2d");
        if (st) return st;
        if (l > 0) {
            const int fl[2] = {n[2 * l - 2], n[2 * l - 1]};
            st = check_coarse2(fl, nl, "vcycle_tail2d");
            if (st) return st;
        }
        MGX_REQUIRE(v[l] && f[l], MGX_ERR_INVALID, "vcycle_tail2d: NULL level array");
        L.sx[l] = nl[0];
        L.sy[l] = nl[1];
        L.v[l] = v[l];
        L.f[l] = f[l];
        L.hx[l] = h[2 * l];
        L.hy[l] = h[2 * l + 1];
        elems += (size_t)2 * nl[0] * nl[1];
    }
    elems += (size_t)n[0] * n[1];  // the residual scratch
    MGX_REQUIRE((size_t)n[0] * n[1] <= (size_t)TAIL2_PT * 1024, MGX_ERR_SIZE, "vcycle_tail2d: the top level has more than %d points", TAIL2_PT * 1024);
    const size_t lds = elems * sizeof(real);
    MGX_REQUIRE(lds <= 150 * 1024, MGX_ERR_SIZE, "vcycle_tail2d: the levels need %zu bytes of LDS (> 150 KB)", lds);
    const real h0[2] = {h[0], h[1]};
    const Lyap2<real> k = lyap<real>(h0, a, A, alfa);
    MGX_TRY_RET(allow_lds(cycle2d_tail_kernel<real>, lds));
    MGX_LAUNCH((cycle2d_tail_kernel<real>), dim3(1), dim3(1024), lds, ctx->compute, L, k, v1, v2, top_zero);
    MGX_LAUNCH_CHECK();
    return MGX_OK;
}

}  // namespace mgx

#define MGX_DEFINE_OPS2D(SFX, real)                                                                              \
    int mgx2d_relax_##SFX(mgx_ctx* ctx, real* v, const real* f, const int n[2], const real h[2], const real a[2], \
                          const real A[4], int alfa, int ncycles) {                                              \
        return mgx::relax2d<real>(ctx, v, f, n, h, a, A, alfa, ncycles);                                         \
    }                                                                                                            \
    int mgx2d_residual_##SFX(mgx_ctx* ctx, const real* v, const real* f, real* r, const int n[2], const real h[2], \
                             const real a[2], const real A[4], int alfa) {                                       \
        return mgx::residual2d<real>(ctx, v, f, r, n, h, a, A, alfa);                                            \
    }                                                                                                            \
    int mgx2d_restrict_##SFX(mgx_ctx* ctx, const real* fine, const int fn[2], real* coarse, const int cn[2]) {    \
        return mgx::restrict2d<real>(ctx, fine, fn, coarse, cn);                                                 \
    }                                                                                                            \
    int mgx2d_residual_restrict_##SFX(mgx_ctx* ctx, const real* v, const real* f, const int n[2], const real h[2], \
                                      const real a[2], const real A[4], int alfa, real* coarse_f, const int cn[2]) { \
        return mgx::residual_restrict2d<real>(ctx, v, f, n, h, a, A, alfa, coarse_f, cn);                         \
    }                                                                                                             \
    int mgx2d_interpolate_correct_##SFX(mgx_ctx* ctx, real* v, const int n[2], const real* coarse_v,              \
                                        const int cn[2]) {                                                        \
        return mgx::interpolate2d<real, true>(ctx, v, n, coarse_v, cn);                                           \
    }                                                                                                             \
    int mgx2d_interpolate_##SFX(mgx_ctx* ctx, real* fine, const int fn[2], const real* coarse, const int cn[2]) { \
        return mgx::interpolate2d<real, false>(ctx, fine, fn, coarse, cn);                                       \
    }                                                                                                            \
    int mgx2d_apply_correction_##SFX(mgx_ctx* ctx, real* fine, const int fn[2], const real* err,                 \
                                     const int en[2]) {                                                          \
        return mgx::correct2d<real>(ctx, fine, fn, err, en);                                                     \
    }                                                                                                            \
    int mgx2d_set_##SFX(mgx_ctx* ctx, real* grid, const int n[2], real value, int modify_boundaries) {           \
        return mgx::set2d<real>(ctx, grid, n, value, modify_boundaries);                                         \
    }                                                                                                            \
    int mgx2d_jacobi_##SFX(mgx_ctx* ctx, real* v, real* tmp, const real* f, const int n[2], const real h[2],     \
                           const real a[2], const real A[4], int alfa, real omega, int ncycles) {                \
        return mgx::jacobi2d<real>(ctx, v, tmp, f, n, h, a, A, alfa, omega, ncycles);                            \
    }                                                                                                            \
    int mgx2d_relax_residual_restrict_##SFX(mgx_ctx* ctx, const real* v_in, real* v_out, const real* f,          \
                                            const int n[2], const real h[2], const real a[2], const real A[4],   \
                                            int alfa, int ncycles, int v_zero, real* coarse_f, const int cn[2]) { \
        return mgx::cycle2d_down<real>(ctx, v_in, v_out, f, n, h, a, A, alfa, ncycles, v_zero, coarse_f, cn);     \
    }                                                                                                             \
    int mgx2d_interpolate_correct_relax_##SFX(mgx_ctx* ctx, const real* v_in, real* v_out, const real* f,         \
                                              const int n[2], const real h[2], const real a[2], const real A[4],  \
                                              int alfa, const real* coarse_v, const int cn[2], int ncycles) {     \
        return mgx::cycle2d_up<real>(ctx, v_in, v_out, f, n, h, a, A, alfa, coarse_v, cn, ncycles);               \
    }                                                                                                             \
    int mgx2d_vcycle_tail_##SFX(mgx_ctx* ctx, int nlev, real* const* v, real* const* f, const int* n,             \
                                const real* h, const real a[2], const real A[4], int alfa, int v1, int v2,        \
                                int top_zero) {                                                                   \
        return mgx::cycle2d_tail<real>(ctx, nlev, v, f, n, h, a, A, alfa, v1, v2, top_zero);                      \
    }                                                                                                             \
    int mgx2d_vcycle_tail_fits_##SFX(const mgx_ctx* ctx, int nlev, const int* n) {                                \
        if (!ctx || !n || nlev < 1 || nlev > mgx::TAIL2_MAXLEV) return 0;                                         \
        size_t e = (size_t)n[0] * n[1];                                                                           \
        if (e > (size_t)mgx::TAIL2_PT * 1024 || e > (size_t)ctx->cyc2_tail_points) return 0;                      \
        for (int l = 0; l < nlev; l++) e += (size_t)2 * n[2 * l] * n[2 * l + 1];                                  \
        return e * sizeof(real) <= 150 * 1024;                                                                    \
    }                                                                                                             \
    int mgx2d_init_v_##SFX(mgx_ctx* ctx, real* v, const int n[2], const real h[2], const real a[2]) {            \
        return mgx::init_v2d<real>(ctx, v, n, h, a);                                                             \
    }                                                                                                            \
    int mgx2d_mean_abs_error_##SFX(mgx_ctx* ctx, const real* v, const int n[2], const real h[2], const real a[2], \
                                   double* host_mean) {                                                          \
        return mgx::mean_abs_error2d<real>(ctx, v, n, h, a, host_mean);                                          \
    }

extern "C" {
MGX_DEFINE_OPS2D(f32, float)
MGX_DEFINE_OPS2D(f64, double)
}

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
            const int y = t / sx, x = t - y * sx;
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
        hipLaunchKernelGGL((relax2d_small_kernel<real>), dim3(1), dim3(1024), 0, ctx->compute, v, f, n[0], n[1], k, ncycles);
        MGX_LAUNCH_CHECK();
        return MGX_OK;
    }
    dim3 g(ceil_div((n[0] + 1) / 2, 64), ceil_div(n[1] - 2, 4), 1);
    for (int c = 0; c < ncycles; c++)
        for (int colour = 0; colour < 2; colour++)
            hipLaunchKernelGGL((relax2d_colour_kernel<real>), g, blk2(), 0, ctx->compute, v, f, n[0], n[1], k, colour);
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
    hipLaunchKernelGGL((residual2d_kernel<real>), grd2(n[0], n[1]), blk2(), 0, ctx->compute, v, f, r, n[0], n[1],
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
    hipLaunchKernelGGL((restrict2d_kernel<real>), grd2(cn[0], cn[1]), blk2(), 0, ctx->compute, fine, fn[0], coarse, cn[0],
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
    hipLaunchKernelGGL((interpolate2d_kernel<real, ADD>), grd2(fn[0] - 2, fn[1] - 2), blk2(), 0, ctx->compute, fine, fn[0],
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
    hipLaunchKernelGGL((residual_restrict2d_kernel<real>), grd2(cn[0], cn[1]), blk2(), 0, ctx->compute, v, f, n[0], n[1],
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
    hipLaunchKernelGGL((correct2d_kernel<real>), grd2(fn[0] - 2, fn[1] - 2), blk2(), 0, ctx->compute, fine, err, fn[0],
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
    hipLaunchKernelGGL((set2d_kernel<real>), grd2(n[0] - 2 * lo, n[1] - 2 * lo), blk2(), 0, ctx->compute, g, n[0], n[1],
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
        hipLaunchKernelGGL((jacobi2d_kernel<real>), grd2(n[0], n[1]), blk2(), 0, ctx->compute, (const real*)src, dst, f, n[0], n[1],
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
        hipLaunchKernelGGL((abs_error2d_kernel<real>), dim3(1, n[1] - 2), dim3(n[0] >= 256 ? 256 : 64), 0, ctx->compute, v, n[0],
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
    int mgx2d_mean_abs_error_##SFX(mgx_ctx* ctx, const real* v, const int n[2], const real h[2], const real a[2], \
                                   double* host_mean) {                                                          \
        return mgx::mean_abs_error2d<real>(ctx, v, n, h, a, host_mean);                                          \
    }

extern "C" {
MGX_DEFINE_OPS2D(f32, float)
MGX_DEFINE_OPS2D(f64, double)
}

// mgx_kernels3d.hip -- 3D Poisson multigrid operators for gfx950 (MI355X), fp32 + fp64.
//
// Every kernel evaluates the per-point expression of the reference in the reference's
// association order, in `real`, with true IEEE division and without FMA contraction
// (this file is compiled with -ffp-contract=off), so results are bit-identical to the
// serial CPU loops: red-black Gauss-Seidel is order-independent within a colour
// (SURVEY.md section 0, fact 7).  Layout: dense, x fastest, idx = x + y*sx + z*sx*sy.
//
// Kernels (reference function each one replaces):
//   relax3d_colour_kernel     one colour of MultiGrid3D::Relax        N3/MultiGrid3D.cpp:489-567
//   residual3d_kernel         MultiGrid3D::CalculateResidual          N3/MultiGrid3D.cpp:678-730
//   restrict3d_kernel         MultiGrid3D::Restrict                   N3/MultiGrid3D.cpp:50-184
//   interpolate3d_kernel      MultiGrid3D::Interpolate (+ApplyCorrection when ADD)
//                                                                     N3/MultiGrid3D.cpp:186-335, 649-676
//   correct3d_kernel          MultiGrid3D::ApplyCorrection            N3/MultiGrid3D.cpp:649-676
//   set3d_kernel              MultiGrid3D::setToValue                 N3/MultiGrid3D.cpp:587-621
//   init_f3d_kernel           Grid3D::InitF                           N3/Grid3D.cpp:78-96
//   residual_restrict3d_kernel  CalculateResidual + Restrict fused through LDS
#include "mgx_internal.hpp"
#include "mgx_kernels3d.hpp"

namespace mgx {

// ------------------------------------------------------------------ relax, one colour
// One thread per point of the colour.  x = 2*ix + p with p = (colour + y + z) & 1 so that
// (x + y + z) % 2 == colour  (red = 0: N3/MultiGrid3D.cpp:515, black = 1: :544).
template <class real>
__global__ void __launch_bounds__(256) relax3d_colour_kernel(real* __restrict__ v, const real* __restrict__ f, int sx,
                                                             int sy, int sz, real hx2, real hy2, real hz2, int colour) {
    const int y = 1 + blockIdx.y * blockDim.y + threadIdx.y;
    const int z = 1 + blockIdx.z;
    if (y >= sy - 1) return;
    const int p = (colour + y + z) & 1;
    const int x = 2 * (blockIdx.x * blockDim.x + threadIdx.x) + p;
    if (x < 1 || x >= sx - 1) return;
    const size_t sxy = (size_t)sx * sy;
    const size_t i = x + (size_t)y * sx + (size_t)z * sxy;
    const real O = v[i - 1], E = v[i + 1];
    const real N = v[i - sx], S = v[i + sx];
    const real D = v[i - sxy], U = v[i + sxy];
    v[i] = relax3d_point<real>(O, E, N, S, D, U, f[i], hx2, hy2, hz2);
}

// ------------------------------------------------------------------ residual
template <class real, int MODE>
__global__ void __launch_bounds__(256) residual3d_kernel(const real* __restrict__ v, const real* __restrict__ f,
                                                         real* __restrict__ r, int sx, int sy, int sz, real hx2,
                                                         real hy2, real hz2) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    const int z = blockIdx.z;
    if (x >= sx || y >= sy) return;
    const size_t sxy = (size_t)sx * sy;
    const size_t i = x + (size_t)y * sx + (size_t)z * sxy;
    if (x == 0 || x == sx - 1 || y == 0 || y == sy - 1 || z == 0 || z == sz - 1) {
        r[i] = (real)0;  // N3/MultiGrid3D.cpp:704-705
        return;
    }
    r[i] = residual3d_point<real, MODE>(v[i - 1], v[i + 1], v[i - sx], v[i + sx], v[i - sxy], v[i + sxy], v[i], f[i],
                                        hx2, hy2, hz2);
}

// ------------------------------------------------------------------ restrict
template <class real>
__global__ void __launch_bounds__(256) restrict3d_kernel(const real* __restrict__ fine, int fx, int fy,
                                                         real* __restrict__ coarse, int cx, int cy, int cz) {
    const int px = blockIdx.x * blockDim.x + threadIdx.x;
    const int py = blockIdx.y * blockDim.y + threadIdx.y;
    const int pz = blockIdx.z;
    if (px >= cx || py >= cy) return;
    const size_t fxy = (size_t)fx * fy;
    const size_t ci = px + (size_t)py * cx + (size_t)pz * cx * cy;
    const real* c = fine + (2 * px + (size_t)(2 * py) * fx + (size_t)(2 * pz) * fxy);
    if (px == 0 || px == cx - 1 || py == 0 || py == cy - 1 || pz == 0 || pz == cz - 1) {
        coarse[ci] = c[0];  // injection, N3/MultiGrid3D.cpp:113-119
        return;
    }
    const ptrdiff_t sy_ = fx, sz_ = (ptrdiff_t)fxy;
    coarse[ci] = restrict3d_point<real>([&](int dx, int dy, int dz) { return c[dx + dy * sy_ + dz * sz_]; });
}

// ------------------------------------------------------------------ interpolate (+ correct)
// ADD = false: fine = I(coarse) on the interior        (Interpolate)
// ADD = true : fine = fine + I(coarse) on the interior (Interpolate into a scratch error
//              array followed by ApplyCorrection, N3/MultiGrid3D.cpp:638-642, fused)
template <class real, bool ADD>
__global__ void __launch_bounds__(256) interpolate3d_kernel(real* __restrict__ fine, int fx, int fy, int fz,
                                                            const real* __restrict__ coarse, int cx, int cy) {
    const int x = 1 + blockIdx.x * blockDim.x + threadIdx.x;
    const int y = 1 + blockIdx.y * blockDim.y + threadIdx.y;
    const int z = 1 + blockIdx.z;
    if (x >= fx - 1 || y >= fy - 1 || z >= fz - 1) return;
    const size_t cxy = (size_t)cx * cy;
    const size_t fi = x + (size_t)y * fx + (size_t)z * fx * fy;
    const real* c = coarse + ((x >> 1) + (size_t)(y >> 1) * cx + (size_t)(z >> 1) * cxy);
    const real e = interpolate3d_point<real>(x & 1, y & 1, z & 1,
                                             [&](int dx, int dy, int dz) { return c[dx + (size_t)dy * cx + (size_t)dz * cxy]; });
    if (ADD) fine[fi] = fine[fi] + e;  // N3/MultiGrid3D.cpp:672
    else fine[fi] = e;
}

template <class real>
__global__ void __launch_bounds__(256) correct3d_kernel(real* __restrict__ fine, const real* __restrict__ err, int sx,
                                                        int sy, int sz) {
    const int x = 1 + blockIdx.x * blockDim.x + threadIdx.x;
    const int y = 1 + blockIdx.y * blockDim.y + threadIdx.y;
    const int z = 1 + blockIdx.z;
    if (x >= sx - 1 || y >= sy - 1 || z >= sz - 1) return;
    const size_t i = x + (size_t)y * sx + (size_t)z * sx * sy;
    fine[i] = fine[i] + err[i];
}

template <class real>
__global__ void __launch_bounds__(256) set3d_kernel(real* __restrict__ g, int sx, int sy, int sz, real value, int lo) {
    const int x = lo + blockIdx.x * blockDim.x + threadIdx.x;
    const int y = lo + blockIdx.y * blockDim.y + threadIdx.y;
    const int z = lo + blockIdx.z;
    if (x >= sx - lo || y >= sy - lo || z >= sz - lo) return;
    g[x + (size_t)y * sx + (size_t)z * sx * sy] = value;
}

// f = (real)(((c * tx[x]) * ty[y]) * tz[z]) in double: Grid3D::InitF's left-to-right product
// -3*PI*PI*sin(PI*x)*sin(PI*y)*sin(PI*z) with the three sines tabulated on the host.
template <class real>
__global__ void __launch_bounds__(256) init_f3d_kernel(real* __restrict__ f, int sx, int sy, int sz, double c,
                                                       const double* __restrict__ tx, const double* __restrict__ ty,
                                                       const double* __restrict__ tz) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    const int z = blockIdx.z;
    if (x >= sx || y >= sy) return;
    f[x + (size_t)y * sx + (size_t)z * sx * sy] = (real)(c * tx[x] * ty[y] * tz[z]);
}

// ------------------------------------------------------------------ residual + restrict fused
// One block produces a CTX x CTY tile of one coarse plane.  It evaluates the fine
// residual on the (2*CTX+1) x (2*CTY+1) x 3 fine points the tile's 27-point stencils touch,
// plane by plane into LDS (boundary points -> 0 exactly like CalculateResidual), then
// applies the full-weighting formula.  The fine residual never goes to HBM.
template <class real, int MODE, int CTX, int CTY>
__global__ void __launch_bounds__(256) residual_restrict3d_kernel(const real* __restrict__ v, const real* __restrict__ f,
                                                                  int sx, int sy, int sz, real hx2, real hy2, real hz2,
                                                                  real* __restrict__ coarse, int cx, int cy, int cz) {
    constexpr int FX = 2 * CTX + 1, FY = 2 * CTY + 1;
    __shared__ real res[3][FY][FX + 1];
    const int pz = blockIdx.z;
    const int px0 = blockIdx.x * CTX, py0 = blockIdx.y * CTY;
    const int tid = threadIdx.y * blockDim.x + threadIdx.x;
    const int nthreads = blockDim.x * blockDim.y;
    const size_t sxy = (size_t)sx * sy;
    const bool zinterior = pz > 0 && pz < cz - 1;
    // fine window origin (may be -1 at the low edge: those entries are never read)
    const int gx0 = 2 * px0 - 1, gy0 = 2 * py0 - 1;
    if (zinterior) {
        for (int k = 0; k < 3; k++) {
            const int gz = 2 * pz - 1 + k;  // 1 .. sz-2 for interior coarse planes
            for (int t = tid; t < FX * FY; t += nthreads) {
                const int ly = t / FX, lx = t - ly * FX;
                const int gx = gx0 + lx, gy = gy0 + ly;
                real rv = (real)0;
                if (gx >= 1 && gx < sx - 1 && gy >= 1 && gy < sy - 1) {
                    const size_t i = gx + (size_t)gy * sx + (size_t)gz * sxy;
                    rv = residual3d_point<real, MODE>(v[i - 1], v[i + 1], v[i - sx], v[i + sx], v[i - sxy], v[i + sxy],
                                                      v[i], f[i], hx2, hy2, hz2);
                }
                res[k][ly][lx] = rv;
            }
        }
    }
    __syncthreads();
    for (int t = tid; t < CTX * CTY; t += nthreads) {
        const int ty = t / CTX, tx = t - ty * CTX;
        const int px = px0 + tx, py = py0 + ty;
        if (px >= cx || py >= cy) continue;
        const size_t ci = px + (size_t)py * cx + (size_t)pz * cx * cy;
        if (px == 0 || px == cx - 1 || py == 0 || py == cy - 1 || !zinterior) {
            coarse[ci] = (real)0;  // injection of a boundary residual, which is 0 (:704-705 then :113-119)
            continue;
        }
        const int lx = 2 * tx + 1, ly = 2 * ty + 1;
        coarse[ci] = restrict3d_point<real>([&](int dx, int dy, int dz) { return res[1 + dz][ly + dy][lx + dx]; });
    }
}

// ------------------------------------------------------------------ sum of squares
template <class real>
__global__ void __launch_bounds__(256) sumsq_kernel(const real* __restrict__ x, size_t count, double* __restrict__ out) {
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
        const double t = (double)x[i];
        acc += t * t;
    }
    // wavefront-wide (64 lanes) shuffle reduction, then one LDS hop across the 4 waves
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    __shared__ double part[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) part[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, (part[0] + part[1]) + (part[2] + part[3]));
}

// =========================================================================== host side
static inline dim3 blk() { return dim3(64, 4, 1); }
static inline dim3 grd(int nx, int ny, int nz) { return dim3(ceil_div(nx, 64), ceil_div(ny, 4), nz); }

template <class real>
static int check_n3(const int n[3], const char* what) {
    MGX_REQUIRE(n, MGX_ERR_INVALID, "%s: size array is NULL", what);
    for (int d = 0; d < 3; d++)
        MGX_REQUIRE(valid_size(n[d]), MGX_ERR_SIZE, "%s: size[%d] = %d is not 2^k+1 >= 3", what, d, n[d]);
    MGX_REQUIRE((double)n[0] * n[1] * n[2] < 2147483647.0 * 4, MGX_ERR_SIZE, "%s: grid too large", what);
    return MGX_OK;
}

static int check_coarse3(const int fn[3], const int cn[3], const char* what) {
    MGX_REQUIRE(fn && cn, MGX_ERR_INVALID, "%s: size array is NULL", what);
    for (int d = 0; d < 3; d++)  // the reference asserts this (N3/MultiGrid3D.cpp:60-62)
        MGX_REQUIRE(cn[d] == (fn[d] - 1) / 2 + 1, MGX_ERR_SIZE, "%s: coarse size[%d] = %d != (%d-1)/2+1", what, d, cn[d],
                    fn[d]);
    return MGX_OK;
}

template <class real>
int relax3d_two_pass(mgx_ctx* ctx, real* v, const real* f, const int n[3], const real h[3], int ncycles) {
    const real hx2 = h[0] * h[0], hy2 = h[1] * h[1], hz2 = h[2] * h[2];  // N3/MultiGrid3D.cpp:498-500
    if (n[0] < 3 || n[1] < 3 || n[2] < 3) return MGX_OK;
    dim3 g(ceil_div((n[0] + 1) / 2, 64), ceil_div(n[1] - 2, 4), n[2] - 2);
    for (int k = 0; k < ncycles; k++)
        for (int colour = 0; colour < 2; colour++) {
            hipLaunchKernelGGL((relax3d_colour_kernel<real>), g, blk(), 0, ctx->compute, v, f, n[0], n[1], n[2], hx2, hy2,
                               hz2, colour);
        }
    MGX_LAUNCH_CHECK();
    return MGX_OK;
}

template <class real>
int relax3d(mgx_ctx* ctx, real* v, const real* f, const int n[3], const real h[3], int ncycles) {
    MGX_REQUIRE(ctx && v && f && h, MGX_ERR_INVALID, "relax3d: NULL argument");
    int st = check_n3<real>(n, "relax3d");
    if (st) return st;
    MGX_REQUIRE(ncycles >= 0, MGX_ERR_INVALID, "relax3d: ncycles = %d < 0", ncycles);
    return relax3d_two_pass<real>(ctx, v, f, n, h, ncycles);
}

template <class real>
int residual3d(mgx_ctx* ctx, const real* v, const real* f, real* r, const int n[3], const real h[3], int mode) {
    MGX_REQUIRE(ctx && v && f && r && h, MGX_ERR_INVALID, "residual3d: NULL argument");
    int st = check_n3<real>(n, "residual3d");
    if (st) return st;
    MGX_REQUIRE(mode == MGX_RESIDUAL_REF_COMPAT || mode == MGX_RESIDUAL_CORRECT, MGX_ERR_INVALID, "residual3d: bad mode %d", mode);
    const real hx2 = h[0] * h[0], hy2 = h[1] * h[1], hz2 = h[2] * h[2];  // N3/MultiGrid3D.cpp:687-689
    if (mode == MGX_RESIDUAL_REF_COMPAT)
        hipLaunchKernelGGL((residual3d_kernel<real, 0>), grd(n[0], n[1], n[2]), blk(), 0, ctx->compute, v, f, r, n[0], n[1],
                           n[2], hx2, hy2, hz2);
    else
        hipLaunchKernelGGL((residual3d_kernel<real, 1>), grd(n[0], n[1], n[2]), blk(), 0, ctx->compute, v, f, r, n[0], n[1],
                           n[2], hx2, hy2, hz2);
    MGX_LAUNCH_CHECK();
    return MGX_OK;
}

template <class real>
int restrict3d(mgx_ctx* ctx, const real* fine, const int fn[3], real* coarse, const int cn[3]) {
    MGX_REQUIRE(ctx && fine && coarse, MGX_ERR_INVALID, "restrict3d: NULL argument");
    int st = check_n3<real>(fn, "restrict3d");
    if (st) return st;
    st = check_coarse3(fn, cn, "restrict3d");
    if (st) return st;
    hipLaunchKernelGGL((restrict3d_kernel<real>), grd(cn[0], cn[1], cn[2]), blk(), 0, ctx->compute, fine, fn[0], fn[1],
                       coarse, cn[0], cn[1], cn[2]);
    MGX_LAUNCH_CHECK();
    return MGX_OK;
}

template <class real, bool ADD>
int interpolate3d(mgx_ctx* ctx, real* fine, const int fn[3], const real* coarse, const int cn[3]) {
    MGX_REQUIRE(ctx && fine && coarse, MGX_ERR_INVALID, "interpolate3d: NULL argument");
    int st = check_n3<real>(fn, "interpolate3d");
    if (st) return st;
    st = check_coarse3(fn, cn, "interpolate3d");
    if (st) return st;
    hipLaunchKernelGGL((interpolate3d_kernel<real, ADD>), grd(fn[0] - 2, fn[1] - 2, fn[2] - 2), blk(), 0, ctx->compute, fine,
                       fn[0], fn[1], fn[2], coarse, cn[0], cn[1]);
    MGX_LAUNCH_CHECK();
    return MGX_OK;
}

template <class real>
int correct3d(mgx_ctx* ctx, real* fine, const int fn[3], const real* err, const int en[3]) {
    MGX_REQUIRE(ctx && fine && err && en, MGX_ERR_INVALID, "apply_correction3d: NULL argument");
    int st = check_n3<real>(fn, "apply_correction3d");
    if (st) return st;
    for (int d = 0; d < 3; d++)  // N3/MultiGrid3D.cpp:660-662
        MGX_REQUIRE(fn[d] == en[d], MGX_ERR_SIZE, "apply_correction3d: size[%d] %d != %d", d, fn[d], en[d]);
    hipLaunchKernelGGL((correct3d_kernel<real>), grd(fn[0] - 2, fn[1] - 2, fn[2] - 2), blk(), 0, ctx->compute, fine, err,
                       fn[0], fn[1], fn[2]);
    MGX_LAUNCH_CHECK();
    return MGX_OK;
}

template <class real>
int set3d(mgx_ctx* ctx, real* g, const int n[3], real value, int modify_boundaries) {
    MGX_REQUIRE(ctx && g, MGX_ERR_INVALID, "set3d: NULL argument");
    int st = check_n3<real>(n, "set3d");
    if (st) return st;
    const int lo = modify_boundaries ? 0 : 1;
    hipLaunchKernelGGL((set3d_kernel<real>), grd(n[0] - 2 * lo, n[1] - 2 * lo, n[2] - 2 * lo), blk(), 0, ctx->compute, g,
                       n[0], n[1], n[2], value, lo);
    MGX_LAUNCH_CHECK();
    return MGX_OK;
}

template <class real>
int residual_restrict3d(mgx_ctx* ctx, const real* v, const real* f, const int n[3], const real h[3], int mode,
                        real* coarse_f, const int cn[3]) {
    MGX_REQUIRE(ctx && v && f && h && coarse_f, MGX_ERR_INVALID, "residual_restrict3d: NULL argument");
    int st = check_n3<real>(n, "residual_restrict3d");
    if (st) return st;
    st = check_coarse3(n, cn, "residual_restrict3d");
    if (st) return st;
    MGX_REQUIRE(mode == MGX_RESIDUAL_REF_COMPAT || mode == MGX_RESIDUAL_CORRECT, MGX_ERR_INVALID,
                "residual_restrict3d: bad mode %d", mode);
    const real hx2 = h[0] * h[0], hy2 = h[1] * h[1], hz2 = h[2] * h[2];
    constexpr int CTX = 32, CTY = 8;
    dim3 g(ceil_div(cn[0], CTX), ceil_div(cn[1], CTY), cn[2]);
    if (mode == MGX_RESIDUAL_REF_COMPAT)
        hipLaunchKernelGGL((residual_restrict3d_kernel<real, 0, CTX, CTY>), g, blk(), 0, ctx->compute, v, f, n[0], n[1], n[2],
                           hx2, hy2, hz2, coarse_f, cn[0], cn[1], cn[2]);
    else
        hipLaunchKernelGGL((residual_restrict3d_kernel<real, 1, CTX, CTY>), g, blk(), 0, ctx->compute, v, f, n[0], n[1], n[2],
                           hx2, hy2, hz2, coarse_f, cn[0], cn[1], cn[2]);
    MGX_LAUNCH_CHECK();
    return MGX_OK;
}

template <class real>
int init_f3d(mgx_ctx* ctx, real* f, const int n[3], double c, const double* tx, const double* ty, const double* tz) {
    MGX_REQUIRE(ctx && f && tx && ty && tz, MGX_ERR_INVALID, "init_f3d: NULL argument");
    int st = check_n3<real>(n, "init_f3d");
    if (st) return st;
    const size_t cnt = (size_t)n[0] + n[1] + n[2];
    void* ws = nullptr;
    st = workspace(ctx, cnt * sizeof(double), &ws);
    if (st) return st;
    double* d = (double*)ws;
    MGX_HIP(hipMemcpyAsync(d, tx, n[0] * sizeof(double), hipMemcpyHostToDevice, ctx->compute));
    MGX_HIP(hipMemcpyAsync(d + n[0], ty, n[1] * sizeof(double), hipMemcpyHostToDevice, ctx->compute));
    MGX_HIP(hipMemcpyAsync(d + n[0] + n[1], tz, n[2] * sizeof(double), hipMemcpyHostToDevice, ctx->compute));
    hipLaunchKernelGGL((init_f3d_kernel<real>), grd(n[0], n[1], n[2]), blk(), 0, ctx->compute, f, n[0], n[1], n[2], c, d,
                       d + n[0], d + n[0] + n[1]);
    MGX_LAUNCH_CHECK();
    MGX_HIP(hipStreamSynchronize(ctx->compute));  // host tables may be freed by the caller
    return MGX_OK;
}

template <class real>
int norm2(mgx_ctx* ctx, const real* x, size_t count, double* host_sumsq) {
    MGX_REQUIRE(ctx && (x || !count) && host_sumsq, MGX_ERR_INVALID, "norm2: NULL argument");
    void* ws = nullptr;
    int st = workspace(ctx, sizeof(double), &ws);
    if (st) return st;
    MGX_HIP(hipMemsetAsync(ws, 0, sizeof(double), ctx->compute));
    if (count) {
        size_t blocks = (count + 255) / 256;
        const size_t cap = (size_t)ctx->num_cus * 8;
        if (blocks > cap) blocks = cap;
        hipLaunchKernelGGL((sumsq_kernel<real>), dim3((unsigned)blocks), dim3(256), 0, ctx->compute, x, count, (double*)ws);
        MGX_LAUNCH_CHECK();
    }
    MGX_HIP(hipMemcpyAsync(host_sumsq, ws, sizeof(double), hipMemcpyDeviceToHost, ctx->compute));
    MGX_HIP(hipStreamSynchronize(ctx->compute));
    return MGX_OK;
}

}  // namespace mgx

#define MGX_DEFINE_OPS3D(SFX, real)                                                                              \
    int mgx3d_relax_##SFX(mgx_ctx* ctx, real* v, const real* f, const int n[3], const real h[3],              \
                          int ncycles) {                                                                         \
        return mgx::relax3d<real>(ctx, v, f, n, h, ncycles);                                                     \
    }                                                                                                            \
    int mgx3d_residual_##SFX(mgx_ctx* ctx, const real* v, const real* f, real* r, const int n[3], const real h[3], \
                             int mode) {                                                                         \
        return mgx::residual3d<real>(ctx, v, f, r, n, h, mode);                                                  \
    }                                                                                                            \
    int mgx3d_restrict_##SFX(mgx_ctx* ctx, const real* fine, const int fn[3], real* coarse, const int cn[3]) {    \
        return mgx::restrict3d<real>(ctx, fine, fn, coarse, cn);                                                 \
    }                                                                                                            \
    int mgx3d_interpolate_##SFX(mgx_ctx* ctx, real* fine, const int fn[3], const real* coarse, const int cn[3]) { \
        return mgx::interpolate3d<real, false>(ctx, fine, fn, coarse, cn);                                       \
    }                                                                                                            \
    int mgx3d_apply_correction_##SFX(mgx_ctx* ctx, real* fine, const int fn[3], const real* err,                 \
                                     const int en[3]) {                                                          \
        return mgx::correct3d<real>(ctx, fine, fn, err, en);                                                     \
    }                                                                                                            \
    int mgx3d_set_##SFX(mgx_ctx* ctx, real* grid, const int n[3], real value, int modify_boundaries) {           \
        return mgx::set3d<real>(ctx, grid, n, value, modify_boundaries);                                         \
    }                                                                                                            \
    int mgx3d_residual_restrict_##SFX(mgx_ctx* ctx, const real* v, const real* f, const int n[3],                \
                                      const real h[3], int mode, real* coarse_f, const int cn[3]) {              \
        return mgx::residual_restrict3d<real>(ctx, v, f, n, h, mode, coarse_f, cn);                              \
    }                                                                                                            \
    int mgx3d_interpolate_correct_##SFX(mgx_ctx* ctx, real* v, const int n[3], const real* coarse_v,             \
                                        const int cn[3]) {                                                       \
        return mgx::interpolate3d<real, true>(ctx, v, n, coarse_v, cn);                                          \
    }                                                                                                            \
    int mgx3d_init_f_##SFX(mgx_ctx* ctx, real* f, const int n[3], double c, const double* host_tx,               \
                           const double* host_ty, const double* host_tz) {                                       \
        return mgx::init_f3d<real>(ctx, f, n, c, host_tx, host_ty, host_tz);                                     \
    }                                                                                                            \
    int mgx_norm2_##SFX(mgx_ctx* ctx, const real* x, size_t count, double* host_sumsq) {                         \
        return mgx::norm2<real>(ctx, x, count, host_sumsq);                                                      \
    }

extern "C" {
MGX_DEFINE_OPS3D(f32, float)
MGX_DEFINE_OPS3D(f64, double)
}

// mgx_kernels3d.hpp -- per-point expressions of the 3D operators (device inline), shared by
// the kernels of both array layouts (natural and x-split) in mgx_kernels3d.hip.
//
// These are the ONLY places where the arithmetic of the reference is restated on the device;
// each keeps the reference's association order so that results are bit-identical
// (-ffp-contract=off; `/` is IEEE-correct for float and double on gfx950).
#pragma once
#include <cmath>
#include <hip/hip_runtime.h>

#include "mgx.h"

namespace mgx {

// Array layouts.  pos(x, H) = position of column x inside its x-row; pitch / half give the row geometry.
struct Natural {  // the reference layout: idx = x + y*sx + z*sx*sy   (N3/MultiGrid3D.cpp:531)
    static constexpr bool xsplit = false;
    static __host__ __device__ __forceinline__ int pos(int x, int) { return x; }
    template <class real> static __host__ __device__ __forceinline__ int half(int sx) { return (sx + 1) >> 1; }
    template <class real> static __host__ __device__ __forceinline__ int pitch(int sx) { return sx; }
};
// x-split: every row stores its even-x half at [0, (sx+1)/2) and its odd-x half at [H, H + sx/2); both halves
// start on a 128-byte boundary (H and the row pitch are multiples of 128/sizeof(real) elements), so every
// half-row -- the unit a colour pass streams -- begins on a cache line and full lines are written.
// Rows and planes keep the reference order; pad entries are never read as data and stay zero.
struct XSplit {
    static constexpr bool xsplit = true;
    static __host__ __device__ __forceinline__ int pos(int x, int H) { return (x >> 1) + (x & 1) * H; }
    template <class real> static __host__ __device__ __forceinline__ int al() { return 128 / (int)sizeof(real); }
    template <class real> static __host__ __device__ __forceinline__ int half(int sx) {
        return (((sx + 1) >> 1) + al<real>() - 1) / al<real>() * al<real>();
    }
    template <class real> static __host__ __device__ __forceinline__ int pitch(int sx) {
        return half<real>(sx) + ((sx >> 1) + al<real>() - 1) / al<real>() * al<real>();
    }
};

// Row geometry of an (sx, sy, *) array in layout L: H = offset of the odd-x half, P = row pitch, PL = plane pitch.
template <class L, class real>
struct Geo {
    int H, P;
    size_t PL;
    __host__ __device__ Geo(int sx, int sy) : H(L::template half<real>(sx)), P(L::template pitch<real>(sx)), PL((size_t)L::template pitch<real>(sx) * sy) {}
    __host__ __device__ __forceinline__ size_t row(int y, int z) const { return (size_t)y * P + (size_t)z * PL; }
    __host__ __device__ __forceinline__ int pos(int x) const { return L::pos(x, H); }
};

// two consecutive elements of `real` as one vector value
template <class real>
struct Vec2T {
    typedef real type __attribute__((ext_vector_type(2)));
};

// MultiGrid3D::Relax per-point update.                      N3/MultiGrid3D.cpp:532 (=:561)
//   v = (O*(hy2*hz2)+E*(hy2*hz2) + N*(hx2*hz2)+S*(hx2*hz2) + D*(hx2*hy2)+U*(hx2*hy2)
//        - f*hx2*hy2*hz2) / (2*(hy2*hz2 + hx2*hz2 + hx2*hy2))
// O/E = x-1/x+1, N/S = y-1/y+1, D/U = z-1/z+1 (:518-529).
template <class real>
__device__ __forceinline__ real relax3d_point(real O, real E, real N, real S, real D, real U, real f, real hx2,
                                              real hy2, real hz2) {
    return (O * (hy2 * hz2) + E * (hy2 * hz2) + N * (hx2 * hz2) + S * (hx2 * hz2) + D * (hx2 * hy2) + U * (hx2 * hy2) -
            f * hx2 * hy2 * hz2) /
           (2 * (hy2 * hz2 + hx2 * hz2 + hx2 * hy2));
}

// relax3d_point for a kernel that updates many points per thread: `rd` = relax3d_rd(hx2, hy2, hz2), formed once.
// fp64: the plain expression.  fp32: the quotient num / den is (float)((double)num * RN53(1 / den)) -- the same bits as the
// IEEE fp32 division, for three instructions instead of eleven: the product is within 2^-52 (relative) of num / den, while
// the quotient of two 24-bit significands lies at least 2^-49 (relative) from every midpoint of two neighbouring fp32
// numbers (num 2^24 - (2M + 1) den is a non-zero integer), so rounding the product to fp32 rounds the way rounding the
// exact quotient does.  That argument needs a normal result: anything below FLT_MIN (zero included) is divided for real.
template <class real>
__device__ __forceinline__ double relax3d_rd(real hx2, real hy2, real hz2) {
    if constexpr (sizeof(real) == 4) return 1.0 / (double)(2 * (hy2 * hz2 + hx2 * hz2 + hx2 * hy2));
    return 0.0;
}
template <class real>
__device__ __forceinline__ real relax3d_point_rd(real O, real E, real N, real S, real D, real U, real f, real hx2, real hy2,
                                                 real hz2, double rd) {
    if constexpr (sizeof(real) == 4) {
        const real num = O * (hy2 * hz2) + E * (hy2 * hz2) + N * (hx2 * hz2) + S * (hx2 * hz2) + D * (hx2 * hy2) + U * (hx2 * hy2) -
                         f * hx2 * hy2 * hz2;
        real q = (real)((double)num * rd);
        if (__builtin_expect(!(__builtin_fabsf(q) >= 1.17549435e-38f), 0)) q = num / (2 * (hy2 * hz2 + hx2 * hz2 + hx2 * hy2));
        return q;
    } else {
        return relax3d_point<real>(O, E, N, S, D, U, f, hx2, hy2, hz2);
    }
}

// relax3d_point_rd for the two points a lane of relax3d_xs_pipe_v2_kernel updates in a row, as ONE element-wise vector expression
// (fp32: v_pk_mul_f32 / v_pk_add_f32 -- every element goes through the operations of the scalar form in the same order: same bits).
template <class real, class vec2>
__device__ __forceinline__ vec2 relax3d_point_rd2(vec2 O, vec2 E, vec2 N, vec2 S, vec2 D, vec2 U, vec2 f, real hx2, real hy2, real hz2, double rd) {
    const vec2 num = O * (hy2 * hz2) + E * (hy2 * hz2) + N * (hx2 * hz2) + S * (hx2 * hz2) + D * (hx2 * hy2) + U * (hx2 * hy2) - f * hx2 * hy2 * hz2;
    const real den = 2 * (hy2 * hz2 + hx2 * hz2 + hx2 * hy2);
    if constexpr (sizeof(real) == 4) {
        vec2 q = {(real)((double)num.x * rd), (real)((double)num.y * rd)};
        // a result below FLT_MIN (zero included) in either element: both are divided for real (one branch per row instead of one per point)
        if (__builtin_expect(!(__builtin_fabsf(q.x) >= 1.17549435e-38f) || !(__builtin_fabsf(q.y) >= 1.17549435e-38f), 0)) q = vec2{num.x / den, num.y / den};
        return q;
    } else {
        return vec2{num.x / den, num.y / den};
    }
}

// MultiGrid3D::CalculateResidual interior expression.       N3/MultiGrid3D.cpp:723
// MODE 0 = REF_COMPAT (the reference's -S / -U), MODE 1 = CORRECT (+S / +U).
// MODE | 2: hx2, hy2, hz2 hold the RECIPROCALS of the squared spacings and the three divisions become multiplications.
// The host asks for this only when all three squared spacings are powers of two (the unit cube on 2^k + 1 points, the
// reference's own set-up): then 1 / h2 is exact, t / h2 and t * (1 / h2) are the correctly rounded value of the same real
// number, i.e. the same bits (overflow and gradual underflow included), and an IEEE fp64 division -- about fifteen
// instructions on CDNA -- costs one.
template <class real, int MODE>
__device__ __forceinline__ real residual3d_point(real O, real E, real N, real S, real D, real U, real c, real f,
                                                 real hx2, real hy2, real hz2) {
    if (MODE == 2) return f - ((O - 2 * c + E) * hx2) - ((N - 2 * c - S) * hy2) - ((D - 2 * c - U) * hz2);
    if (MODE == 3) return f - ((O - 2 * c + E) * hx2) - ((N - 2 * c + S) * hy2) - ((D - 2 * c + U) * hz2);
    if (MODE == 0) return f - ((O - 2 * c + E) / hx2) - ((N - 2 * c - S) / hy2) - ((D - 2 * c - U) / hz2);
    return f - ((O - 2 * c + E) / hx2) - ((N - 2 * c + S) / hy2) - ((D - 2 * c + U) / hz2);
}

// is 1 / h exact (h a normal power of two with a normal reciprocal)?
template <class real>
static inline bool exact_reciprocal(real h) {
    int e;
    return h > 0 && std::isnormal(h) && std::frexp(h, &e) == (real)0.5 && std::isnormal((real)1 / h);
}

// MultiGrid3D::Restrict interior formula.                   N3/MultiGrid3D.cpp:122-180
// get(dx,dy,dz) returns the fine value at offset (dx,dy,dz) from the fine point 2*(cx,cy,cz).
// Reference names: suffix _C/_N/_S = y, y-1, y+1; prefix N/S = z+1/z-1, E/O = x+1/x-1.
template <class real, class Get>
__device__ __forceinline__ real restrict3d_point(Get get) {
    const real C_C = get(0, 0, 0), N_C = get(0, 0, 1), S_C = get(0, 0, -1), E_C = get(1, 0, 0), O_C = get(-1, 0, 0);
    const real NE_C = get(1, 0, 1), NO_C = get(-1, 0, 1), SE_C = get(1, 0, -1), SO_C = get(-1, 0, -1);
    const real C_N = get(0, -1, 0), N_N = get(0, -1, 1), S_N = get(0, -1, -1), E_N = get(1, -1, 0), O_N = get(-1, -1, 0);
    const real NE_N = get(1, -1, 1), NO_N = get(-1, -1, 1), SE_N = get(1, -1, -1), SO_N = get(-1, -1, -1);
    const real C_S = get(0, 1, 0), N_S = get(0, 1, 1), S_S = get(0, 1, -1), E_S = get(1, 1, 0), O_S = get(-1, 1, 0);
    const real NE_S = get(1, 1, 1), NO_S = get(-1, 1, 1), SE_S = get(1, 1, -1), SO_S = get(-1, 1, -1);
    return (1 / 8.0f) * (C_C) + (1 / 16.0f) * ((N_C + E_C + S_C + O_C) + (C_N + C_S)) +
           (1 / 32.0f) * ((NE_C + SE_C + SO_C + NO_C) + (N_N + E_N + S_N + O_N) + (N_S + E_S + S_S + O_S)) +
           (1 / 64.0f) * ((NE_N + SE_N + SO_N + NO_N) + (NE_S + SE_S + SO_S + NO_S));
}

// MultiGrid3D::Interpolate value of a fine interior point by parity class.
// ox/oy/oz = fine index odd?; get(dx,dy,dz) = coarse value at (x/2+dx, y/2+dy, z/2+dz).
//                                                           N3/MultiGrid3D.cpp:216-329
template <class real, class Get>
__device__ __forceinline__ real interpolate3d_point(int ox, int oy, int oz, Get get) {
    if (!oz) {
        if (!oy) {
            if (!ox) return get(0, 0, 0);                                             // PPP :216
            return (1 / 2.0f) * (get(0, 0, 0) + get(1, 0, 0));                        // PDP :222-229
        }
        if (!ox) return (1 / 2.0f) * (get(0, 0, 0) + get(0, 1, 0));                   // DPP :233-240
        return (1 / 4.0f) * (get(0, 0, 0) + get(1, 0, 0) + get(0, 1, 0) + get(1, 1, 0));  // DDP :244-255
    }
    if (!oy) {
        if (!ox) return (1 / 2.0f) * (get(0, 0, 0) + get(0, 0, 1));                   // PPD :261-268
        return (1 / 4.0f) * (get(0, 0, 1) + get(1, 0, 1) + get(0, 0, 0) + get(1, 0, 0));  // PDD :272-283
    }
    if (!ox) return (1 / 4.0f) * (get(0, 0, 0) + get(0, 0, 1) + get(0, 1, 0) + get(0, 1, 1));  // DPD :287-298
    return (1 / 8.0f) * (get(0, 0, 0) + get(0, 0, 1) + get(1, 0, 1) + get(1, 0, 0) + get(0, 1, 0) + get(0, 1, 1) +
                         get(1, 1, 1) + get(1, 1, 0));                                // DDD :302-329
}

// ------------------------------------------------------------------ device helpers shared by the kernel files
// Loads and stores through buffer descriptors: the address is descriptor base (the plane, SGPRs) + a uniform offset (row and
// plane, one SGPR) + the lane's 32-bit offset inside the row -- no 64-bit address arithmetic in vector registers, which this
// kernel has none to spare.  The descriptor covers `planes` planes from its base; stride 0 (raw), the gfx950 format word.
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
template <class real>
__device__ __forceinline__ __amdgpu_buffer_rsrc_t plane_rsrc(const real* base, int plane_elems, int planes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)((unsigned)plane_elems * (unsigned)planes * (unsigned)sizeof(real)), 0x00020000);
}
template <class real>
__device__ __forceinline__ real buf_load(__amdgpu_buffer_rsrc_t r, unsigned voff, int soff_elems) {
    if constexpr (sizeof(real) == 8) {
        const u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(r, voff, (unsigned)soff_elems * 8u, 0);
        return __builtin_bit_cast(double, t);
    } else {
        return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff, (unsigned)soff_elems * 4u, 0));
    }
}
template <class real>
__device__ __forceinline__ void buf_store_nt(real x, __amdgpu_buffer_rsrc_t r, unsigned voff, int soff_elems) {
    if constexpr (sizeof(real) == 8) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, x), r, voff, (unsigned)soff_elems * 8u, 2);
    else __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(x), r, voff, (unsigned)soff_elems * 4u, 2);
}
// two consecutive elements (the two x-pairs of a lane of relax3d_xs_pipe_v2_kernel) as one 8- / 16-byte access
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <class real, int AUX>
__device__ __forceinline__ typename Vec2T<real>::type buf_load2_aux(__amdgpu_buffer_rsrc_t r, unsigned voff, int soff_elems) {
    typedef typename Vec2T<real>::type vec2;
    if constexpr (sizeof(real) == 8) return __builtin_bit_cast(vec2, __builtin_amdgcn_raw_buffer_load_b128(r, voff, (unsigned)soff_elems * 8u, AUX));
    else return __builtin_bit_cast(vec2, __builtin_amdgcn_raw_buffer_load_b64(r, voff, (unsigned)soff_elems * 4u, AUX));
}
template <class real>
__device__ __forceinline__ typename Vec2T<real>::type buf_load2(__amdgpu_buffer_rsrc_t r, unsigned voff, int soff_elems) {
    return buf_load2_aux<real, 0>(r, voff, soff_elems);
}
template <class real>
__device__ __forceinline__ typename Vec2T<real>::type buf_load2_nt(__amdgpu_buffer_rsrc_t r, unsigned voff, int soff_elems) {
    return buf_load2_aux<real, 2>(r, voff, soff_elems);
}
template <class real>
__device__ __forceinline__ void buf_store2_nt(typename Vec2T<real>::type x, __amdgpu_buffer_rsrc_t r, unsigned voff, int soff_elems) {
    if constexpr (sizeof(real) == 8) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, x), r, voff, (unsigned)soff_elems * 8u, 2);
    else __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, x), r, voff, (unsigned)soff_elems * 4u, 2);
}
template <class real>
__device__ __forceinline__ real buf_load_nt(__amdgpu_buffer_rsrc_t r, unsigned voff, int soff_elems) {  // read once: non-temporal
    if constexpr (sizeof(real) == 8) {
        const u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(r, voff, (unsigned)soff_elems * 8u, 2);
        return __builtin_bit_cast(double, t);
    } else {
        return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff, (unsigned)soff_elems * 4u, 2));
    }
}

// block -> tile of a 1-D launch over gx x gy x gz tiles (x fastest).  xcd_mode 1: consecutive blocks go to the eight XCDs
// in turn, so every XCD is given one contiguous run of the tile order -- workgroups whose tiles share rows or cache lines
// then share an L2 (see relax3d_xs_kernel); xcd_mode 0: plain order.
__device__ __forceinline__ void tile_of_block(int xcd_mode, int gx, int gy, int& bx, int& by, int& bz) {
    unsigned b = blockIdx.x;
    if (xcd_mode == 1) {
        const unsigned nb = gridDim.x, k = b & 7u, per = nb >> 3, rem = nb & 7u;
        b = k * per + (k < rem ? k : rem) + (b >> 3);
    }
    bx = b % gx;
    by = (b / gx) % gy;
    bz = b / (gx * gy);
}

template <class real>
__device__ __forceinline__ real wave_from_prev_lane(real x) {  // lane i gets lane i-1 (lane 0 keeps its own)
    if constexpr (sizeof(real) == 8) {
        int lo = __double2loint(x), hi = __double2hiint(x);
        lo = __builtin_amdgcn_update_dpp(lo, lo, 0x138, 0xf, 0xf, false);  // wave_shr:1
        hi = __builtin_amdgcn_update_dpp(hi, hi, 0x138, 0xf, 0xf, false);
        return __hiloint2double(hi, lo);
    } else {
        return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(x), __float_as_int(x), 0x138, 0xf, 0xf, false));
    }
}
template <class real>
__device__ __forceinline__ real wave_from_next_lane(real x) {  // lane i gets lane i+1 (lane 63 keeps its own)
    if constexpr (sizeof(real) == 8) {
        int lo = __double2loint(x), hi = __double2hiint(x);
        lo = __builtin_amdgcn_update_dpp(lo, lo, 0x130, 0xf, 0xf, false);  // wave_shl:1
        hi = __builtin_amdgcn_update_dpp(hi, hi, 0x130, 0xf, 0xf, false);
        return __hiloint2double(hi, lo);
    } else {
        return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(x), __float_as_int(x), 0x130, 0xf, 0xf, false));
    }
}

}  // namespace mgx

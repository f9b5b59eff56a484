// mgx_resident3d.hip -- MultiGrid3D::Relax (N3/MultiGrid3D.cpp:489-567) on the cache-resident levels (33 ... 129 points per
// row, x-split layout): ALL colour passes of a Relax call in ONE launch.
//
// On these levels a colour pass is a 5-8 us launch that moves almost nothing; a Relax call of the reference's own workloads
// (FMG(2, 3000, 3000): 6000 passes per level visit) is nothing but launch latency.  Here the level is cut into tiles of
// 8 rows x 8 planes x all of x, one workgroup per tile, all resident at once (at most one per CU: 129^3 = 16 x 16 tiles).  A
// workgroup loads its tile once, keeps v in LDS (with a ring of halo lines around it) and f in registers, runs every pass on
// it and writes the tile back at the end.  Between passes the tiles exchange the lines on their four faces through a small
// buffer in memory.  Every value travels WITH ITS TAG -- (launch epoch, pass) next to the value in one naturally aligned
// 8-byte (fp32) / 16-byte (fp64) element, stored and loaded by one lane in one write-through (sc1) access -- so there is no
// progress word, no draining of stores and no second round trip: after pass p a wave stores the half-lines it has just
// updated on the tile's faces; before pass p + 1 the wave in charge of a halo line loads the neighbour's element and repeats
// until every lane sees the tag of pass p (bounded; a wait that does not end sets the context's abort word, mgx_sync.hpp).
// A pass reads only the other colour, which the neighbours updated in the pass before, so one exchange per pass is all
// there is; the buffer is double-buffered by pass parity (a neighbour may be one pass ahead, never two).  Every point is
// computed from exactly the values the serial loops would use, with relax3d_point: bit-identical results.  One hand-off
// costs ~2 us (the per-workgroup-flag form of the same kernel: ~4 us; a launch of a colour pass: 3.7-7 us).
//
// A wave owns 2 rows x 2 planes of the tile (16 waves); lane i the x-pair {2i, 2i+1}.  With 129 points per row the last even
// entry (x = 128, a boundary value) has no lane: it rides in a register per line.
#include "mgx_internal.hpp"
#include "mgx_kernels3d.hpp"
#include "mgx_sync.hpp"

namespace mgx {

constexpr int RT = 8;  // tile edge in rows and in planes

// one exchanged element = {value, tag}: 16 bytes in both precisions (the tag = launch epoch << 20 | pass + 1 is 64 bits wide:
// it never wraps and is never 0, the content of a fresh buffer), stored and loaded as ONE aligned 16-byte unit
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
template <class real>
__device__ __forceinline__ void put_tagged(__amdgpu_buffer_rsrc_t r, unsigned elem, real x, u64 tag) {
    u32x4_t t;
    if constexpr (sizeof(real) == 4) {
        t.x = __float_as_uint(x);
        t.y = 0;
    } else {
        const u64 b = (u64)__double_as_longlong(x);
        t.x = (unsigned)b;
        t.y = (unsigned)(b >> 32);
    }
    t.z = (unsigned)tag;
    t.w = (unsigned)(tag >> 32);
    __builtin_amdgcn_raw_buffer_store_b128(t, r, elem * 16u, 0, 16);  // aux 16 = sc1: write-through, visible to the other XCDs
}
template <class real>
__device__ __forceinline__ bool get_tagged(__amdgpu_buffer_rsrc_t r, unsigned elem, u64 tag, real* x) {
    const u32x4_t t = __builtin_amdgcn_raw_buffer_load_b128(r, elem * 16u, 0, 16);
    if constexpr (sizeof(real) == 4) *x = __uint_as_float(t.x);
    else *x = __longlong_as_double((long long)(((u64)t.y << 32) | (u64)t.x));
    return (((u64)t.w << 32) | (u64)t.z) == tag;
}
template <class real>
__device__ __forceinline__ u64 pass_tag(u64 ep, int p) {
    return (ep << 20) | (u64)(p + 1);
}

// The per-sweep kernel's lines: only the lanes that hold entries (lane < M) take part, and fp32 packs the values of an even lane and
// its odd neighbour into ONE element {v[2j], v[2j + 1], tag} -- at 65^3 and 129^3 the exchange is bound by the bytes it moves
// (one line per relaxed line and sweep, 16 bytes per 4- or 8-byte value), not by its latency.  Element j of a line: fp64 lane j's
// value, fp32 the values of lanes 2j and 2j + 1.  put_line is called by whole waves (it shuffles).
template <class real>
__device__ __forceinline__ void put_line(__amdgpu_buffer_rsrc_t r, unsigned line_elem, int lane, int M, real x, u64 tag) {
    if constexpr (sizeof(real) == 4) {
        const real x1 = __shfl_down(x, 1, 64);
        if ((lane & 1) == 0 && lane < M) {
            u32x4_t t;
            t.x = __float_as_uint(x);
            t.y = __float_as_uint(x1);
            t.z = (unsigned)tag;
            t.w = (unsigned)(tag >> 32);
            __builtin_amdgcn_raw_buffer_store_b128(t, r, (line_elem + (unsigned)(lane >> 1)) * 16u, 0, 16);
        }
    } else {
        if (lane < M) put_tagged<real>(r, line_elem + (unsigned)lane, x, tag);
    }
}
template <class real>
__device__ __forceinline__ bool get_line(__amdgpu_buffer_rsrc_t r, unsigned line_elem, int lane, int M, u64 tag, real* x) {
    if (lane >= M) return true;
    if constexpr (sizeof(real) == 4) {
        const u32x4_t t = __builtin_amdgcn_raw_buffer_load_b128(r, (line_elem + (unsigned)(lane >> 1)) * 16u, 0, 16);
        *x = __uint_as_float((lane & 1) ? t.y : t.x);
        return (((u64)t.w << 32) | (u64)t.z) == tag;
    } else {
        return get_tagged<real>(r, line_elem + (unsigned)lane, tag, x);
    }
}

template <class real>
__global__ void __launch_bounds__(1024)
    relax3d_xs_resident_kernel(real* __restrict__ v, const real* __restrict__ f, int sx, int sy, int sz, real hx2, real hy2, real hz2,
                               int npasses, int zero_start, int gy, int gz, void* __restrict__ xbuf, unsigned xbytes, SweepSync sync) {
    __shared__ real L[RT + 2][RT + 2][2][64];  // [row slot][plane slot][half][pair]: slot 0 / RT + 1 = the halo ring
    const Geo<XSplit, real> g(sx, sy);
    const int lane = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.y);
    const int M = (sx + 1) >> 1;  // entries of the even half; the odd half has M - 1
    const int ty = blockIdx.x % gy, tz = blockIdx.x / gy;
    const int y0 = 1 + ty * RT, z0 = 1 + tz * RT;
    const bool has_ym = ty > 0, has_yp = ty < gy - 1, has_zm = tz > 0, has_zp = tz < gz - 1;
    const u64 ep = *sync.epoch;  // written by the previous launch's last workgroup: a kernel boundary lies in between
    const u64 ep_wait = ep + (blockIdx.x == 0 ? sync.fault : 0u);  // the epoch of the tags this workgroup waits for (test hook: see SweepSync)
    const double rd = relax3d_rd<real>(hx2, hy2, hz2);
    const unsigned facesz = RT * 64, tilesz = 4 * facesz, bufsz = gridDim.x * tilesz;  // xbuf[2][tile][face][RT][64] elements
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(xbuf, 0, (int)xbytes, 0x00020000);

    // ---- the tile and its halo ring from memory (corners are never read): all loads of a wave first, then LDS (one memory
    // round trip for the wave's seven lines instead of seven)
    {
        constexpr int NL = ((RT + 2) * (RT + 2) + 15) / 16;
        real ta[NL], tb[NL];
#pragma unroll
        for (int j = 0; j < NL; j++) {
            const int idx = w + 16 * j;
            const int ly = idx / (RT + 2), lz = idx % (RT + 2);
            const int y = y0 - 1 + ly, z = z0 - 1 + lz;
            ta[j] = tb[j] = 0;
            if (idx < (RT + 2) * (RT + 2) && y <= sy - 1 && z <= sz - 1) {
                const bool inner = y >= 1 && y <= sy - 2 && z >= 1 && z <= sz - 2;
                if (!(zero_start && inner)) {
                    const real* row = v + g.row(y, z);
                    if (lane < M) ta[j] = row[lane];
                    if (lane < M - 1) tb[j] = row[g.H + lane];
                }
            }
        }
#pragma unroll
        for (int j = 0; j < NL; j++) {
            const int idx = w + 16 * j;
            if (idx < (RT + 2) * (RT + 2)) {
                L[idx / (RT + 2)][idx % (RT + 2)][0][lane] = ta[j];
                L[idx / (RT + 2)][idx % (RT + 2)][1][lane] = tb[j];
            }
        }
    }
    // ---- own lines: f, the boundary entry past the last lane
    int ly[4], lz[4], par[4];
    bool valid[4];
    real fA[4], fB[4], xR[4];
#pragma unroll
    for (int l = 0; l < 4; l++) {
        const int ry = 2 * (w & 3) + (l & 1), rz = 2 * (w >> 2) + (l >> 1);
        const int y = y0 + ry, z = z0 + rz;
        ly[l] = ry + 1;
        lz[l] = rz + 1;
        par[l] = (y + z) & 1;
        valid[l] = y <= sy - 2 && z <= sz - 2;
        fA[l] = fB[l] = xR[l] = 0;
        if (valid[l]) {
            const size_t row = g.row(y, z);
            if (lane < M) fA[l] = f[row + lane];
            if (lane < M - 1) fB[l] = f[row + g.H + lane];
            if (M - 1 == 64 && !zero_start) xR[l] = v[row + 64];
        }
    }
    __syncthreads();

    bool gave_up = false;
    for (int p = 0; p < npasses; p++) {
        const int c = p & 1;  // red = 0 first (N3/MultiGrid3D.cpp:515, :544)
        if (p > 0) {
            // ---- the neighbours' face lines of pass p - 1 into the halo ring: wave w takes face w >> 2, lines 2 (w & 3), + 1
            const int face = w >> 2;
            const bool have = face == 0 ? has_ym : (face == 1 ? has_yp : (face == 2 ? has_zm : has_zp));
            if (have) {
                const int nb = face == 0 ? (int)blockIdx.x - 1 : (face == 1 ? (int)blockIdx.x + 1 : (face == 2 ? (int)blockIdx.x - gy : (int)blockIdx.x + gy));
                const unsigned src = (unsigned)((p - 1) & 1) * bufsz + (unsigned)nb * tilesz + (unsigned)(face ^ 1) * facesz;
                const u64 tag = pass_tag<real>(ep_wait, p - 1);
#pragma unroll
                for (int k2 = 0; k2 < 2; k2++) {
                    const int k = 2 * (w & 3) + k2;
                    const int sy_ = face == 0 ? 0 : (face == 1 ? RT + 1 : k + 1), sz_ = face == 2 ? 0 : (face == 3 ? RT + 1 : k + 1);
                    const int y = y0 - 1 + sy_, z = z0 - 1 + sz_;
                    if (y > sy - 2 || z > sz - 2) continue;  // past the grid: the neighbour has no such line (and nobody reads it)
                    real x = 0;
                    unsigned spins = 0;
                    while (!gave_up) {
                        asm volatile("" ::: "memory");  // a fresh load every time round (the builtin is an ordinary read to the compiler)
                        const bool ok = get_tagged<real>(xr, src + (unsigned)k * 64u + (unsigned)lane, tag, &x);
                        if (__builtin_amdgcn_readfirstlane((int)__all(ok))) break;
                        if (++spins > sync.spin_limit ||
                            ((spins & 1023u) == 0 && __hip_atomic_load(sync.abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0)) {
                            if (lane == 0) __hip_atomic_store(sync.abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                            gave_up = true;
                        }
                    }
                    const int hq = ((p - 1) + y + z) & 1;  // the half of that line pass p - 1 updated
                    L[sy_][sz_][hq][lane] = x;
                }
            }
            __syncthreads();
        }
        // ---- pass p on the wave's four lines
        const unsigned dst = (unsigned)(p & 1) * bufsz + blockIdx.x * tilesz;
        const u64 mytag = pass_tag<real>(ep, p);
#pragma unroll
        for (int l = 0; l < 4; l++) {
            if (!valid[l]) continue;
            const int q = c ^ par[l];  // the half of this line that holds colour c
            const int a = ly[l], b = lz[l];
            const real oth = L[a][b][1 - q][lane];
            real O, E;
            if (q == 0) {
                O = L[a][b][1][lane > 0 ? lane - 1 : 0];
                E = oth;
            } else {
                O = oth;
                const real t = L[a][b][0][lane < 63 ? lane + 1 : 63];
                E = lane < 63 ? t : xR[l];
            }
            const real N = L[a - 1][b][q][lane], S = L[a + 1][b][q][lane], D = L[a][b - 1][q][lane], U = L[a][b + 1][q][lane];
            const real nv = relax3d_point_rd<real>(O, E, N, S, D, U, q ? fB[l] : fA[l], hx2, hy2, hz2, rd);
            const bool upd = q == 0 ? (lane >= 1 && lane <= M - 2) : lane <= M - 2;
            if (upd) L[a][b][q][lane] = nv;
            if (p + 1 < npasses) {  // the faces of the tile for the neighbours' next pass (lanes that are not updated: never read)
                const int ry = a - 1, rz = b - 1;
                if (ry == 0 && has_ym) put_tagged<real>(xr, dst + 0 * facesz + (unsigned)rz * 64u + (unsigned)lane, nv, mytag);
                if (ry == RT - 1 && has_yp) put_tagged<real>(xr, dst + 1 * facesz + (unsigned)rz * 64u + (unsigned)lane, nv, mytag);
                if (rz == 0 && has_zm) put_tagged<real>(xr, dst + 2 * facesz + (unsigned)ry * 64u + (unsigned)lane, nv, mytag);
                if (rz == RT - 1 && has_zp) put_tagged<real>(xr, dst + 3 * facesz + (unsigned)ry * 64u + (unsigned)lane, nv, mytag);
            }
        }
        // (the barrier at the top of the next pass orders this pass's LDS writes before their readers)
    }
    __syncthreads();
    // ---- the tile back to memory (interior entries)
#pragma unroll
    for (int l = 0; l < 4; l++) {
        if (!valid[l]) continue;
        real* row = v + g.row(y0 + ly[l] - 1, z0 + lz[l] - 1);
        if (lane >= 1 && lane <= M - 2) row[lane] = L[ly[l]][lz[l]][0][lane];
        if (lane <= M - 2) row[g.H + lane] = L[ly[l]][lz[l]][1][lane];
    }
    // launch epoch: the last workgroup to finish advances it (every workgroup has read it by then)
    __syncthreads();
    if (threadIdx.x == 0 && threadIdx.y == 0) {
        const unsigned old = __hip_atomic_fetch_add(sync.done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == gridDim.x - 1) {
            __hip_atomic_store(sync.done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(sync.epoch, ep + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// ------------------------------------------------------------------ the same with ONE exchange per sweep
// The hand-off, not the arithmetic, is what a pass costs (~2-3 us against ~0.3 us).  Here the tiles exchange once per red+black
// SWEEP instead of once per pass: after its black pass a tile publishes the black values of the lines within TWO of its faces;
// before the next red pass it fetches, from its four face neighbours, their two outermost lines and, from its four diagonal
// neighbours, the one corner line -- the black values around its first halo ring.  With those it relaxes the red points of the
// first ring's face lines ITSELF (the same expression on the same inputs as the owner: the same bits), then the black points of
// its own lines.  LDS holds the tile with a halo two lines deep.  Buffer: [sweep parity][tile][face][depth 2][RT lines][64]
// elements; a neighbour can be one sweep ahead, never two, for the reason given above.
template <class real, int RT>
__global__ void __launch_bounds__(1024)
    relax3d_xs_resident2_kernel(real* __restrict__ v, const real* __restrict__ f, int sx, int sy, int sz, real hx2, real hy2, real hz2,
                                int nsweeps, int zero_start, int gy, int gz, void* __restrict__ xbuf, unsigned xbytes, SweepSync sync) {
    constexpr int LS = RT + 4;                 // line slots per direction: own line r at slot r + 2
    __shared__ real L[LS][LS][2][64];          // [row slot][plane slot][half][pair]
    const Geo<XSplit, real> g(sx, sy);
    const int lane = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.y);
    const int M = (sx + 1) >> 1;
    const int ty = blockIdx.x % gy, tz = blockIdx.x / gy;
    const int y0 = 1 + ty * RT, z0 = 1 + tz * RT;
    const bool has_ym = ty > 0, has_yp = ty < gy - 1, has_zm = tz > 0, has_zp = tz < gz - 1;
    const u64 ep = *sync.epoch;
    const u64 ep_wait = ep + (blockIdx.x == 0 ? sync.fault : 0u);
    const double rd = relax3d_rd<real>(hx2, hy2, hz2);
    const unsigned depthsz = RT * 64, facesz = 2 * depthsz, tilesz = 4 * facesz, bufsz = gridDim.x * tilesz;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(xbuf, 0, (int)xbytes, 0x00020000);
    auto interior = [&](int y, int z) { return y >= 1 && y <= sy - 2 && z >= 1 && z <= sz - 2; };

    // ---- the tile and two lines around it from memory (all loads of a wave first, then LDS)
    {
        constexpr int NL = (LS * LS + 15) / 16;
        real ta[NL], tb[NL];
#pragma unroll
        for (int j = 0; j < NL; j++) {
            const int idx = w + 16 * j;
            const int y = y0 - 2 + idx / LS, z = z0 - 2 + idx % LS;
            ta[j] = tb[j] = 0;
            if (idx < LS * LS && y >= 0 && z >= 0 && y <= sy - 1 && z <= sz - 1) {
                if (!(zero_start && interior(y, z))) {
                    const real* row = v + g.row(y, z);
                    if (lane < M) ta[j] = row[lane];
                    if (lane < M - 1) tb[j] = row[g.H + lane];
                }
            }
        }
#pragma unroll
        for (int j = 0; j < NL; j++) {
            const int idx = w + 16 * j;
            if (idx < LS * LS) {
                L[idx / LS][idx % LS][0][lane] = ta[j];
                L[idx / LS][idx % LS][1][lane] = tb[j];
            }
        }
    }
    // ---- the wave's lines: four of its own (both colours), two of the first halo ring (red only)
    constexpr int BE = RT / 4;            // a wave owns BE x BE lines of the tile ...
    constexpr int NO = BE * BE, NR = BE;  // ... and relaxes the red points of BE lines of the first halo ring
    int la[NO + NR], lb[NO + NR], par[NO + NR];  // row slot, plane slot, parity (y + z) & 1
    bool valid[NO + NR];
    real fred[NO + NR], fblk[NO], xR[NO + NR];  // f at the line's red / black points (halo lines: red only), the boundary entry past lane 63
    const int face = w >> 2;  // the face whose halo lines this wave fetches and relaxes: 0 = y-, 1 = y+, 2 = z-, 3 = z+
    const bool have = face == 0 ? has_ym : (face == 1 ? has_yp : (face == 2 ? has_zm : has_zp));
#pragma unroll
    for (int l = 0; l < NO + NR; l++) {
        int ry, rz;  // line coordinates relative to the tile
        if (l < NO) {
            ry = BE * (w & 3) + (l % BE);
            rz = BE * (w >> 2) + (l / BE);
        } else {
            const int k = BE * (w & 3) + (l - NO);
            ry = face == 0 ? -1 : (face == 1 ? RT : k);
            rz = face == 2 ? -1 : (face == 3 ? RT : k);
        }
        const int y = y0 + ry, z = z0 + rz;
        la[l] = ry + 2;
        lb[l] = rz + 2;
        par[l] = (y + z) & 1;
        valid[l] = interior(y, z) && (l < NO || have);
        xR[l] = 0;
        fred[l] = 0;
        if (l < NO) fblk[l] = 0;
        if (valid[l]) {
            const size_t row = g.row(y, z);
            if (l < NO) {
                real a = 0, b = 0;
                if (lane < M) a = f[row + lane];
                if (lane < M - 1) b = f[row + g.H + lane];
                fred[l] = par[l] ? b : a;  // red (colour 0) sits in half par, black in the other
                fblk[l] = par[l] ? a : b;
            } else {
                const int q = par[l];
                if (q == 0 ? lane < M : lane < M - 1) fred[l] = f[row + q * g.H + lane];
            }
            if (M - 1 == 64 && !zero_start) xR[l] = v[row + 64];
        }
    }
    // ---- what the wave fetches before a sweep: depth 0 and 1 of BE lines of its face, and (waves 1, 5, 9, 13) one corner line
    constexpr int NF = 2 * BE + 1;
    unsigned fsrc[NF];  // element offset inside one parity's buffer
    int fa[NF], fb[NF], fh[NF];
    unsigned fneed = 0;
#pragma unroll
    for (int i = 0; i < NF; i++) {
        int ry, rz, nb;
        unsigned off;
        if (i < 2 * BE) {
            const int k = BE * (w & 3) + (i % BE), d = i / BE;
            ry = face == 0 ? -1 - d : (face == 1 ? RT + d : k);
            rz = face == 2 ? -1 - d : (face == 3 ? RT + d : k);
            nb = face == 0 ? (int)blockIdx.x - 1 : (face == 1 ? (int)blockIdx.x + 1 : (face == 2 ? (int)blockIdx.x - gy : (int)blockIdx.x + gy));
            off = (unsigned)(face ^ 1) * facesz + (unsigned)d * depthsz + (unsigned)k * 64u;
            if (!have) nb = -1;
        } else {
            // corner c = face: 0 = (-1, -1), 1 = (-1, RT), 2 = (RT, -1), 3 = (RT, RT); the owner holds it in a y-face buffer, depth 0
            const int c = face;
            ry = c < 2 ? -1 : RT;
            rz = (c & 1) ? RT : -1;
            const bool ex = (c < 2 ? has_ym : has_yp) && ((c & 1) ? has_zp : has_zm) && (w & 3) == 1;
            nb = ex ? (int)blockIdx.x + (c < 2 ? -1 : 1) + ((c & 1) ? gy : -gy) : -1;
            off = (unsigned)(c < 2 ? 1 : 0) * facesz + (unsigned)((c & 1) ? 0 : RT - 1) * 64u;
        }
        const int y = y0 + ry, z = z0 + rz;
        fa[i] = ry + 2;
        fb[i] = rz + 2;
        fh[i] = 1 ^ ((y + z) & 1);  // black (colour 1) sits in half 1 ^ parity
        fsrc[i] = nb >= 0 ? (unsigned)nb * tilesz + off : 0u;
        if (nb >= 0 && interior(y, z)) fneed |= 1u << i;
    }
    fneed = (unsigned)__builtin_amdgcn_readfirstlane((int)fneed);
    __syncthreads();

    // LDS indices of the lane's entries of each line, formed ONCE and kept in vector registers (the empty asm makes them opaque:
    // rebuilt from the wave-uniform slot numbers at every use they cost three scalar and one vector instruction per LDS access):
    // ired = the entry in the half that holds the line's red point, iblk = the other half
    real* const Lf = &L[0][0][0][0];
    int ired[NO + NR], iblk[NO + NR];
#pragma unroll
    for (int l = 0; l < NO + NR; l++) {
        const int i0 = (la[l] * LS + lb[l]) * 128 + lane;
        ired[l] = i0 + par[l] * 64;
        iblk[l] = i0 + (1 - par[l]) * 64;
        asm volatile("" : "+v"(ired[l]), "+v"(iblk[l]));
    }
    // the new value of the entry at index iq (half q of its line; io = the same lane in the other half); any lane: callers mask
    auto relax_at = [&](int iq, int io, int q, real ff, real xr_) -> real {
        const real oth = Lf[io];
        real O, E;
        if (q == 0) {
            O = Lf[io - 1];  // lane 0: the entry in front of the half, in range, and x = 0 is never updated
            E = oth;
        } else {
            O = oth;
            const real t = Lf[io + 1];  // lane 63: the entry behind the half, in range, replaced by the line's boundary value
            E = lane < 63 ? t : xr_;
        }
        const real N = Lf[iq - LS * 128], S = Lf[iq + LS * 128], D = Lf[iq - 128], U = Lf[iq + 128];
        return relax3d_point_rd<real>(O, E, N, S, D, U, ff, hx2, hy2, hz2, rd);
    };
    auto updated = [&](int q) { return q == 0 ? (lane >= 1 && lane <= M - 2) : lane <= M - 2; };

    bool gave_up = false;
    for (int s = 0; s < nsweeps; s++) {
        if (s > 0) {
            // ---- the neighbours' black values of sweep s - 1: all loads of a round first, then the tags
            const unsigned base = (unsigned)((s - 1) & 1) * bufsz;
            const u64 tag = pass_tag<real>(ep_wait, s - 1);
            real x[NF];
            unsigned pending = fneed, spins = 0;
            while (pending && !gave_up) {
                asm volatile("" ::: "memory");  // fresh loads every time round
                bool ok[NF];
#pragma unroll
                for (int i = 0; i < NF; i++) {
                    ok[i] = true;
                    if (pending & (1u << i)) ok[i] = get_line<real>(xr, base + fsrc[i], lane, M, tag, &x[i]);
                }
#pragma unroll
                for (int i = 0; i < NF; i++)
                    if ((pending & (1u << i)) && __builtin_amdgcn_readfirstlane((int)__all(ok[i]))) pending &= ~(1u << i);
                if (pending && (++spins > sync.spin_limit ||
                                ((spins & 1023u) == 0 && __hip_atomic_load(sync.abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0))) {
                    if (lane == 0) __hip_atomic_store(sync.abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    gave_up = true;
                }
            }
#pragma unroll
            for (int i = 0; i < NF; i++)
                if ((fneed & (1u << i)) && updated(fh[i])) L[fa[i]][fb[i]][fh[i]][lane] = x[i];  // boundary entries keep their value
            __syncthreads();
        }
        // ---- red (colour 0, N3/MultiGrid3D.cpp:515) on the wave's own lines and its lines of the first halo ring
#pragma unroll
        for (int l = 0; l < NO + NR; l++) {
            if (!valid[l]) continue;
            const int q = par[l];
            const real nv = relax_at(ired[l], iblk[l], q, fred[l], xR[l]);
            if (updated(q)) Lf[ired[l]] = nv;
        }
        __syncthreads();
        // ---- black (:544) on the own lines; their values to the neighbours
        const unsigned dst = (unsigned)(s & 1) * bufsz + blockIdx.x * tilesz;
        const u64 mytag = pass_tag<real>(ep, s);
#pragma unroll
        for (int l = 0; l < NO; l++) {
            if (!valid[l]) continue;
            const int q = 1 ^ par[l];
            const real nv = relax_at(iblk[l], ired[l], q, fblk[l], xR[l]);
            if (updated(q)) Lf[iblk[l]] = nv;
            if (s + 1 < nsweeps) {
                const int ry = la[l] - 2, rz = lb[l] - 2;
                if (ry <= 1 && has_ym) put_line<real>(xr, dst + 0 * facesz + (unsigned)ry * depthsz + (unsigned)rz * 64u, lane, M, nv, mytag);
                if (ry >= RT - 2 && has_yp) put_line<real>(xr, dst + 1 * facesz + (unsigned)(RT - 1 - ry) * depthsz + (unsigned)rz * 64u, lane, M, nv, mytag);
                if (rz <= 1 && has_zm) put_line<real>(xr, dst + 2 * facesz + (unsigned)rz * depthsz + (unsigned)ry * 64u, lane, M, nv, mytag);
                if (rz >= RT - 2 && has_zp) put_line<real>(xr, dst + 3 * facesz + (unsigned)(RT - 1 - rz) * depthsz + (unsigned)ry * 64u, lane, M, nv, mytag);
            }
        }
        // (the barrier behind the next sweep's fetch orders these LDS writes before their readers; the fetch itself writes only
        // black entries of halo lines, which nobody reads during a black pass)
    }
    __syncthreads();
    // ---- the tile back to memory (interior entries)
#pragma unroll
    for (int l = 0; l < NO; l++) {
        if (!valid[l]) continue;
        real* row = v + g.row(y0 + la[l] - 2, z0 + lb[l] - 2);
        if (lane >= 1 && lane <= M - 2) row[lane] = L[la[l]][lb[l]][0][lane];
        if (lane <= M - 2) row[g.H + lane] = L[la[l]][lb[l]][1][lane];
    }
    __syncthreads();
    if (threadIdx.x == 0 && threadIdx.y == 0) {
        const unsigned old = __hip_atomic_fetch_add(sync.done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == gridDim.x - 1) {
            __hip_atomic_store(sync.done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(sync.epoch, ep + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// Does a Relax call of `ncycles` sweeps on this level run in the resident kernel?  Rows of 33 ... 129 points (below: the
// one-workgroup kernels; above: the level does not fit), all tiles resident at once, a context that has the GPU to itself
// (thread-ranks and ranks of a communicator launch side by side: co-residency is not given), enough passes to pay for the load
// and the write-back of the tile.
// workgroups of a resident kernel that fit a CU at once, asked of the runtime for every variant (the smallest answer counts; 0 = a
// variant cannot be launched at all on this device: the kernels are not used)
template <class K>
static int occupancy_of(K kernel) {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)kernel, 1024, 0) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return nb;
}
static int resident_occupancy(const mgx_ctx* ctx) {
    if (ctx->resident_occ < 0) {
        int o = occupancy_of(relax3d_xs_resident_kernel<float>);
        o = min(o, occupancy_of(relax3d_xs_resident_kernel<double>));
        o = min(o, occupancy_of(relax3d_xs_resident2_kernel<float, 4>));
        o = min(o, occupancy_of(relax3d_xs_resident2_kernel<double, 4>));
        o = min(o, occupancy_of(relax3d_xs_resident2_kernel<float, 8>));
        o = min(o, occupancy_of(relax3d_xs_resident2_kernel<double, 8>));
        ctx->resident_occ = o > 0 ? 1 : 0;  // the tile choice below counts on ONE workgroup per CU; more would only relax the test
    }
    return ctx->resident_occ;
}

bool relax3d_resident_takes(const mgx_ctx* ctx, const int n[3], int ncycles) {
    if (!ctx->relax_resident || ctx->nranks > 1 || ctx->local_group) return false;
    if (ctx->handoff_broken || !ctx->gpu_exclusive) return false;  // a wait has given up before / the GPU is shared: colour passes
    if (n[0] < 33 || n[0] > 129 || n[1] < 9 || n[2] < 9) return false;
    if (ncycles < ctx->relax_resident_min) return false;
    const int tiles = ceil_div(n[1] - 2, RT) * ceil_div(n[2] - 2, RT);
    if (tiles > SWEEP_MAX_WG) return false;
    return tiles <= ctx->num_cus * resident_occupancy(ctx);
}

// the exchange buffer at the size of the largest level the kernel takes (as many tiles as CUs): [2][tile][4 faces][RT lines][64]
// elements of 16 bytes, zeroed (tag 0 is never waited for)
static int resident_buffer(mgx_ctx* ctx, size_t bytes) {
    if (ctx->resident_bytes >= bytes) return MGX_OK;
    if (ctx->resident_buf) MGX_HIP(hipFree(ctx->resident_buf));
    ctx->resident_buf = nullptr;
    ctx->resident_bytes = 0;
    MGX_HIP(hipMalloc(&ctx->resident_buf, bytes));
    ctx->resident_bytes = bytes;
    return fill_zero(ctx, ctx->resident_buf, bytes);
}

// What the kernels with inter-workgroup hand-offs allocate on first use, allocated now: a hierarchy calls this when it is
// created, so that its first cycle can be captured into a HIP graph (an allocation inside a capture is an error).
int prepare_handoffs(mgx_ctx* ctx) {
    SweepSync sync;
    MGX_TRY_RET(sweep_state(ctx, &sync));
    const int tiles = ctx->num_cus < SWEEP_MAX_WG ? ctx->num_cus : SWEEP_MAX_WG;
    return resident_buffer(ctx, (size_t)2 * tiles * 4 * 2 * RT * 64 * 16);
}

template <class real>
int relax3d_resident(mgx_ctx* ctx, real* v, const real* f, const int n[3], real hx2, real hy2, real hz2, int ncycles, int zero_start) {
    // "relax3d.resident": 1 = one exchange per sweep, 2 = one per pass.  A call of ONE sweep always takes the per-pass form: its
    // single exchange is also what keeps a tile from writing its lines back before the neighbours have loaded them as their halo
    // (with an exchange per sweep a one-sweep launch has none)
    const bool per_sweep = ctx->relax_resident == 1 && ncycles >= 2;
    // tiles of 4 x 4 lines where the level then still has at most one workgroup per CU (33^3: 64, 65^3: 256 workgroups): the passes
    // are bound by instruction issue, a quarter of the lines per workgroup is worth more than the larger share of halo lines
    int rt = RT;
    if (per_sweep && ctx->resident_tile != 8 && ceil_div(n[1] - 2, 4) * ceil_div(n[2] - 2, 4) <= (ctx->num_cus < SWEEP_MAX_WG ? ctx->num_cus : SWEEP_MAX_WG)) rt = 4;
    const int gy = ceil_div(n[1] - 2, rt), gz = ceil_div(n[2] - 2, rt);
    SweepSync sync;
    MGX_TRY_RET(sweep_state(ctx, &sync));
    const size_t bytes = (size_t)2 * gy * gz * 4 * 2 * rt * 64 * 16;  // the per-sweep form's lay-out; the per-pass form uses half of it
    MGX_TRY_RET(resident_buffer(ctx, bytes));
    if (per_sweep)
        snprintf(ctx->last_relax_kernel, sizeof ctx->last_relax_kernel, "relax3d_xs_resident2_kernel<%s,%d>", sizeof(real) == 8 ? "double" : "float", rt);
    else
        snprintf(ctx->last_relax_kernel, sizeof ctx->last_relax_kernel, "relax3d_xs_resident_kernel<%s>", sizeof(real) == 8 ? "double" : "float");
    for (int left = ncycles, first = 1; left > 0; first = 0) {  // a tag has 20 bits for the pass / sweep
        int k = left < (1 << 18) ? left : (1 << 18);
        if (per_sweep && left - k == 1) k--;  // never leave a launch of one sweep behind
        if (per_sweep && rt == 4)
            MGX_LAUNCH((relax3d_xs_resident2_kernel<real, 4>), dim3(gy * gz), dim3(64, 16, 1), 0, ctx->compute, v, f, n[0], n[1], n[2], hx2,
                               hy2, hz2, k, first ? zero_start : 0, gy, gz, ctx->resident_buf, (unsigned)ctx->resident_bytes, sync);
        else if (per_sweep)
            MGX_LAUNCH((relax3d_xs_resident2_kernel<real, 8>), dim3(gy * gz), dim3(64, 16, 1), 0, ctx->compute, v, f, n[0], n[1], n[2], hx2,
                               hy2, hz2, k, first ? zero_start : 0, gy, gz, ctx->resident_buf, (unsigned)ctx->resident_bytes, sync);
        else
            MGX_LAUNCH((relax3d_xs_resident_kernel<real>), dim3(gy * gz), dim3(64, 16, 1), 0, ctx->compute, v, f, n[0], n[1], n[2], hx2,
                               hy2, hz2, 2 * k, first ? zero_start : 0, gy, gz, ctx->resident_buf, (unsigned)ctx->resident_bytes, sync);
        left -= k;
    }
    return MGX_OK;
}
template int relax3d_resident<float>(mgx_ctx*, float*, const float*, const int[3], float, float, float, int, int);
template int relax3d_resident<double>(mgx_ctx*, double*, const double*, const int[3], double, double, double, int, int);

}  // namespace mgx

extern "C" int mgx_ctx_prepare(mgx_ctx* ctx) {
    MGX_REQUIRE(ctx, MGX_ERR_INVALID, "ctx is NULL");
    MGX_USE(ctx);
    return mgx::prepare_handoffs(ctx);
}

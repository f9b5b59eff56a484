// mgx_sweep3d.hip -- ONE launch per red+black sweep of MultiGrid3D::Relax on the HBM-bound levels (x-split layout).
//
// The reference's sweep is one red nest and one black nest over the same arrays (N3/MultiGrid3D.cpp:509-565).  The
// two-launch form (relax3d_xs_pipe_kernel, one colour per launch) streams 1.5 words per point per colour: the black pass
// re-reads from HBM the red values the red launch has just written.  Here a workgroup marches its (x, y) tile through a
// run of planes doing BOTH colours, the red stage D planes ahead of the black stage:
//
//   step s:   red   plane s + D   from the OLD black values of `vin`  (planes s+D-1, s+D, s+D+1: the pipelined kernel's
//                                  registers, edge rows / lanes of neighbouring waves through LDS) -> vout, LDS ring
//             black plane s       from the NEW red values: own tile from the LDS ring (planes s-1, s, s+1), the rows just
//                                  outside the tile from `vout`, where the NEIGHBOURING WORKGROUP has stored them -> vout
//
// Out of place (vin -> vout, ping-pong): the red stage reads values that no workgroup writes in this launch, so the only
// inter-workgroup dependence is read-after-write (black needs the neighbours' red rim rows), and that one is ordered by
// per-workgroup progress flags -- no halo is recomputed.  Per sweep HBM sees: old black read (0.5 word per point), f read
// (1.0), red and black written (1.0) = 2.5 words instead of 3.  Every point is still computed from exactly the values the
// serial loops would use, with the reference's expression (relax3d_point): bit-identical results.
//
// Hand-off protocol (cdna_hip_programming.md, Guideline 16, form R1 / MI355X_MICROARCH.md visibility table, first row):
//   producer: the tile's first and last row of every red plane are stored WRITE-THROUGH (sc1); every wave drains its stores
//             (s_waitcnt vmcnt(0)) at the end of each step; behind the NEXT step's workgroup barrier one lane stores the
//             workgroup's progress word (sc1): "red planes <= p are in memory".
//   consumer: the wave that needs a rim row polls the neighbour's word (sc1 load, prefetched one step ahead, bounded spin),
//             and only then issues its own sc1 loads of that row.  Words carry a launch epoch (kept in device memory and
//             advanced by the last workgroup of a launch), so nothing has to be cleared between launches or graph replays.
//   All workgroups of a launch must be resident together (grid <= CUs, one workgroup per CU: the host checks); a wait that
//   does not end sets the context's abort word (host-visible) and gives up, so every wave terminates.
//
// z-runs of one tile are independent: a run recomputes the red plane below its first and above its last plane (from vin,
// never stored), so nothing is exchanged in z.
#include "mgx_internal.hpp"
#include "mgx_kernels3d.hpp"
#include "mgx_sync.hpp"

namespace mgx {

// Tile = 64 WX pairs x WY R rows, the whole x-extent of the level (gx = 1: the host only picks shapes with 64 WX >= M - 1),
// so the only neighbours are the tiles above and below in y.  D = lead of the red stage in planes; the ring holds the red
// planes s .. s + D of the tile.
// DBG (diagnostic builds only): per-wave cycle stamps of the phases of a step go to `dbg`, `abl` switches parts off
// (WRONG results): 1 no waits on progress words, 2 no write-through / sc1 accesses, 4 no black arithmetic, 8 no red
// arithmetic, 16 no global stores.
template <class real, int WX, int WY, int R, int D, bool FNT, int DBG = 0, bool ILV = false>
__global__ void __launch_bounds__(64 * WX * WY)
    sweep3d_xs_kernel(const real* __restrict__ vin, real* vout, const real* __restrict__ f, int sx, int sy, int sz, int zbeg,
                      int zend, real hx2, real hy2, real hz2, int c0, int zchunk, int gy, int xcd_mode, SweepSync sync,
                      long long* dbg = nullptr, int abl = 0) {
    long long tlast = 0, tph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define MGX_STAMP(k)                                         \
    do {                                                     \
        if (DBG) {                                           \
            const long long n_ = __builtin_readcyclecounter(); \
            tph[k] += n_ - tlast;                            \
            tlast = n_;                                      \
        }                                                    \
    } while (0)
    const bool a_nowait = DBG && (abl & 1), a_nosc = DBG && (abl & 2), a_noblk = DBG && (abl & 4), a_nored = DBG && (abl & 8),
               a_nost = DBG && (abl & 16);
    constexpr int TX = 64 * WX, TY = WY * R, NR = D + 1;
    static_assert(D >= 3 && WY >= 2 && R >= 1, "shape");
    __shared__ real ey[2][WY][WX][2][64];
    __shared__ real ex[2][WY][WX][2][R];
    __shared__ real ring[NR][TY][TX];
    const double rd = relax3d_rd<real>(hx2, hy2, hz2);
    const Geo<XSplit, real> g(sx, sy);
    const int H = g.H;
    const int M = (sx + 1) >> 1;
    unsigned b = blockIdx.x;
    if (xcd_mode == 1) {
        const unsigned nb = gridDim.x, k = b & 7u, per = nb >> 3, rem = nb & 7u;
        b = k * per + (k < rem ? k : rem) + (b >> 3);
    }
    const int by = b % gy, bz = b / gy;
    const int lane = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.y);
    const int wx = w % WX, wy = w / WX;
    const int jn = wx * 64 + lane;
    const bool lane_on = jn < M - 1;
    const int j = lane_on ? jn : M - 2;
    // j again, opaque to the compiler: inside `if (j == M - 2)` it knows the address is wave-uniform and turns a vector load
    // into a scalar one, which it then waits for on the spot -- a round trip to memory in the middle of the issue phase
    int jv = j;
    asm volatile("" : "+v"(jv));
    const int y0 = 1 + (by * WY + wy) * R;
    const int nrows = max(0, min(R, sy - 1 - y0));
    const int z0 = zbeg + bz * zchunk;
    const int z1 = min(z0 + zchunk, zend);  // the host sizes the grid so that no run is empty
    const int sxy = (int)g.PL;
    const bool rimR = j == M - 2;  // E neighbour of x = sx - 2 is the boundary column
    const bool rimL = j == 0;      // pair 0 holds the boundary column x = 0
    const int wyN = wy > 0 ? wy - 1 : 0, wyS = wy < WY - 1 ? wy + 1 : WY - 1;
    const int wxL = wx > 0 ? wx - 1 : 0, wxR = wx < WX - 1 ? wx + 1 : WX - 1;
    int roff[R];
#pragma unroll
    for (int r = 0; r < R; r++) roff[r] = min(y0 + r, sy - 1) * g.P;
    const int roffN = (y0 - 1) * g.P, roffS = min(y0 + R, sy - 1) * g.P;
    // black stage: the row above the wave's first row / below its last VALID row, where it is not a row of this tile
    const bool Nmem = wy == 0 && nrows > 0;
    const int yS = y0 + nrows;
    const bool Smem = nrows > 0 && (wy == WY - 1 || yS == sy - 1);
    const int rl = nrows - 1;  // the wave's last valid row
    const bool NfromNb = Nmem && by > 0;              // ... written by the tile above in this launch
    const bool SfromNb = Smem && yS < sy - 1;         // ... written by the tile below in this launch
    const int roffSb = min(yS, sy - 1) * g.P;
    const u64 ep = *sync.epoch;  // written by the previous launch's last workgroup: a kernel boundary lies in between
    const u64 epbase = ep << 20;
    const u64 epwait = (ep + (blockIdx.x == 0 ? sync.fault : 0u)) << 20;  // test hook: see SweepSync
    const u64* dep = nullptr;
    if (NfromNb) dep = sync.flags + (bz * gy + by - 1);
    if (SfromNb) dep = sync.flags + (bz * gy + by + 1);
    u64* myflag = sync.flags + (bz * gy + by);
    bool gave_up = false;

    // ---- red stage state (as relax3d_xs_pipe_kernel): planes zr-1, zr, zr+1 of the column, next plane on its way
    const int zrf = max(z0 - 1, 1), zrl = min(z1, sz - 2);  // first / last red plane this run computes
    const real* pv = vin + (size_t)zrf * g.PL;
    const real* pf = f + (size_t)zrf * g.PL;
    int q = (c0 + y0 + zrf) & 1;
    real cp[R], cc[R], cu[R], cn[R], fc[R], fn[R], xc[R], xn[R], oc[R], op[R];
    real Nc = 0, Sc = 0, Nn = 0, Sn = 0;
    // ---- black stage state
    real fb[R], fbn[R], ob[R], obp[R], Pw[R], Dw[R], xb[R], xbn[R], zb[R], zbn[R];
    real Nb = 0, Sb = 0, Nbn = 0, Sbn = 0;
    u64 pollv = 0;
#pragma unroll
    for (int r = 0; r < R; r++) fb[r] = fbn[r] = ob[r] = obp[r] = Pw[r] = Dw[r] = xb[r] = xbn[r] = zb[r] = zbn[r] = 0;

    // rim of the red stage: everything that comes from vin besides the column itself, plane at offset dz from pv, row parity
    // qq.  Lane j = M - 2 needs x = sx - 1 (index j + 1 of half 0) in q_r = 1 rows; lane j = 0 has x = 0 as its red point in
    // q_r = 0 rows: the boundary value itself is loaded (it goes into the ring in place of a result).
#define MGX_LOAD_RIM(dz, qq, X, Nv, Sv)                                                        \
    do {                                                                                       \
        const real* p_ = pv + (dz) * sxy;                                                      \
        if (wy == 0) Nv = p_[roffN + (qq) * H + j];                                            \
        if (wy == WY - 1) Sv = p_[roffS + ((qq) ^ ((R - 1) & 1)) * H + j];                     \
        if (rimL || rimR) {                                                                    \
            _Pragma("unroll") for (int r = 0; r < R; r++) {                                    \
                const int qr_ = (qq) ^ (r & 1);                                                \
                X[r] = p_[roff[r] + (qr_ ? (rimR ? j + 1 : j) : (rimL ? 0 : H + j))];          \
            }                                                                                  \
        }                                                                                      \
    } while (0)
    auto publish = [&](int slot, const real (&c)[R]) __attribute__((always_inline)) {
        ey[slot][wy][wx][0][lane] = c[0];
        ey[slot][wy][wx][1][lane] = c[R - 1];
        if (lane == 0 || lane == 63) {
#pragma unroll
            for (int r = 0; r < R; r++) ex[slot][wy][wx][lane == 63][r] = c[r];
        }
    };

    // prologue of the red stage: planes zrf-1, zrf, zrf+1 of the column, f and rim of plane zrf
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int qr = q ^ (r & 1);
        cp[r] = pv[roff[r] - sxy + qr * H + j];
        cc[r] = pv[roff[r] + (1 - qr) * H + j];
        cu[r] = pv[roff[r] + sxy + qr * H + j];
        fc[r] = (FNT ? __builtin_nontemporal_load(&pf[roff[r] + qr * H + j]) : pf[roff[r] + qr * H + j]);
        xc[r] = xn[r] = cn[r] = fn[r] = 0;
        op[r] = oc[r] = 0;
    }
    MGX_LOAD_RIM(0, q, xc, Nc, Sc);
    int es_c = zrf & 1, es_n = (zrf + 1) & 1;  // exchange slots of the red planes zr and zr + 1
    publish(es_c, cc);
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    // (Tried: the lower half of the waves half a step behind the upper half -- "issue | barrier | arithmetic | barrier" with
    // one half issuing while the other computes, a third exchange slot.  Bit-exact, but 0.90 instead of 0.69 ms per sweep at
    // 513^3: with four instead of eight waves issuing, a CU has too few bytes in flight and each half's issue phase takes
    // as long as the whole workgroup's did.  profiles/r03_sweep_stamps.txt)

    // ring slots of the planes zr = s + D (written), s (read across lanes) and s + 1 (own entry read)
    int s = z0 - 1 - D;
    int sl_r = (z0 - 1) % NR, sl_b = z0 % NR, sl_u = (z0 + 1) % NR;  // NR = D + 1: s = zr - D = zr + 1 (mod NR)
    real* const ringp = &ring[0][0][0];
    const int mycell = (wy * R) * TX + wx * 64 + lane;  // row 0 of this lane inside a ring slot

    if (DBG) tlast = __builtin_readcyclecounter();
    for (; s < z1; s++) {
        const int zr = s + D;
        const bool red_c = zr >= zrf && zr <= zrl;      // the red stage computes plane zr in this step
        const bool red_more = red_c && zr + 1 <= zrl;   // ... and plane zr + 1 in the next one
        const bool red_st = zr - 1 >= max(z0, zrf) && zr - 1 <= min(z1 - 1, zrl);  // plane zr - 1 (last step's) is stored
        const bool blk_c = s >= z0;                     // the black stage computes plane s
        const bool blk_st = s - 1 >= z0;                // plane s - 1 is stored
        const bool blk_next = s + 1 >= z0 && s + 1 < z1;
        const int qb0 = (c0 + 1 + y0 + s) & 1;          // x-parity of the black point of row 0 in plane s
        // The pieces of a step.  PHASED order (ILV = 0): all memory instructions first, then the arithmetic of both stages.
        // INTERLEAVED order (ILV = 1): the same pieces, memory instructions in groups between the rows of the arithmetic, so that
        // the CU's memory pipe has requests queued while waves compute (in the phased order the pipe
        // idles during the arithmetic, and the arithmetic waits during the issue phase, for every wave at once).
        const int qn0 = qb0 ^ 1;  // row 0's black parity in plane s + 1
#define MGX_P_STORE_RED                                                                                                       \
    if (red_st && !a_nost) {                                                                                                  \
        real* p = vout + (size_t)(zr - 1) * g.PL;                                                                             \
        _Pragma("unroll") for (int r = 0; r < R; r++) {                                                                       \
            const int qr = q ^ 1 ^ (r & 1); /* q already belongs to plane zr */                                               \
            if (lane_on && (qr | j) && r < nrows) {                                                                           \
                /* the rows a neighbouring workgroup reads in this launch go to memory write-through */                      \
                const bool shared = (wy == 0 && r == 0 && by > 0) || (wy == WY - 1 && r == R - 1 && by < gy - 1);             \
                if (shared && !a_nosc) st_sc1(&p[roff[r] + qr * H + j], op[r]);                                               \
                else __builtin_nontemporal_store(op[r], &p[roff[r] + qr * H + j]);                                            \
            }                                                                                                                 \
        }                                                                                                                     \
    }
#define MGX_P_STORE_BLK                                                                                                       \
    if (blk_st && !a_nost) {                                                                                                  \
        real* p = vout + (size_t)(s - 1) * g.PL;                                                                              \
        _Pragma("unroll") for (int r = 0; r < R; r++) {                                                                       \
            const int qb = qb0 ^ 1 ^ (r & 1);                                                                                 \
            if (lane_on && (qb | j) && r < nrows) __builtin_nontemporal_store(obp[r], &p[roff[r] + qb * H + j]);              \
        }                                                                                                                     \
    }
#define MGX_P_LOAD_CN                                                                                                         \
    if (red_more) {                                                                                                           \
        _Pragma("unroll") for (int r = 0; r < R; r++) cn[r] = pv[roff[r] + 2 * sxy + (q ^ 1 ^ (r & 1)) * H + j];              \
        MGX_LOAD_RIM(1, q ^ 1, xn, Nn, Sn);                                                                                   \
    }
#define MGX_P_LOAD_FN                                                                                                         \
    if (red_more) {                                                                                                           \
        _Pragma("unroll") for (int r = 0; r < R; r++) {                                                                       \
            const int qn = q ^ 1 ^ (r & 1);                                                                                   \
            fn[r] = (FNT ? __builtin_nontemporal_load(&pf[roff[r] + sxy + qn * H + j]) : pf[roff[r] + sxy + qn * H + j]);     \
        }                                                                                                                     \
    }
        // loads of the black stage's next step (plane s + 1): f, the boundary column / planes, and -- behind the neighbour's
        // progress word, polled a step ago -- its red rim row
#define MGX_P_LOAD_BLK                                                                                                        \
    if (blk_next) {                                                                                                           \
        const real* pfb = f + (size_t)(s + 1) * g.PL;                                                                         \
        _Pragma("unroll") for (int r = 0; r < R; r++) {                                                                       \
            const int qb = qn0 ^ (r & 1);                                                                                     \
            fbn[r] = (FNT ? __builtin_nontemporal_load(&pfb[roff[r] + qb * H + j]) : pfb[roff[r] + qb * H + j]);              \
        }                                                                                                                     \
        const real* pi = vin + (size_t)(s + 1) * g.PL;                                                                        \
        if (rimR) { /* E neighbour of x = sx - 2: the boundary column (index M - 1 of half 0), never written */              \
            _Pragma("unroll") for (int r = 0; r < R; r++) xbn[r] = pi[roff[r] + jv + 1]; /* through jv: see there */          \
        }                                                                                                                     \
        if (s + 1 == 1 || s + 1 == sz - 2) { /* the plane below / above is a boundary plane: its entries as they are */      \
            const real* pz = vin + (s + 1 == 1 ? (size_t)0 : (size_t)(sz - 1) * g.PL);                                        \
            _Pragma("unroll") for (int r = 0; r < R; r++) zbn[r] = pz[roff[r] + (qn0 ^ (r & 1)) * H + j];                     \
        }                                                                                                                     \
        MGX_STAMP(0);                                                                                                         \
        if (dep && !a_nowait) {                                                                                               \
            const u64 need = epwait + (u64)(s + 2);                                                                           \
            u64 pv_ = __builtin_amdgcn_readfirstlane((unsigned)(pollv >> 32));                                                \
            pv_ = (pv_ << 32) | (u64)__builtin_amdgcn_readfirstlane((unsigned)pollv);                                         \
            unsigned spins = 0;                                                                                               \
            while (pv_ < need && !gave_up) {                                                                                  \
                __builtin_amdgcn_s_sleep(8);                                                                                  \
                const u64 t_ = ld_sc1(dep);                                                                                   \
                pv_ = __builtin_amdgcn_readfirstlane((unsigned)(t_ >> 32));                                                   \
                pv_ = (pv_ << 32) | (u64)__builtin_amdgcn_readfirstlane((unsigned)t_);                                        \
                if (++spins > sync.spin_limit ||                                                                             \
                    ((spins & 1023u) == 0 && __hip_atomic_load(sync.abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0)) { \
                    if (lane == 0) __hip_atomic_store(sync.abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);           \
                    gave_up = true;                                                                                           \
                }                                                                                                             \
            }                                                                                                                 \
            asm volatile("" ::: "memory"); /* the rim loads stay behind the poll */                                           \
        }                                                                                                                     \
        MGX_STAMP(1);                                                                                                         \
        real* po_ = vout + (size_t)(s + 1) * g.PL;                                                                            \
        if (Nmem) {                                                                                                           \
            if (NfromNb && !a_nosc) Nbn = ld_sc1(&po_[roffN + qn0 * H + j]);                                                  \
            else Nbn = pi[roffN + qn0 * H + j];                                                                               \
        }                                                                                                                     \
        if (Smem) {                                                                                                           \
            const int qs = qn0 ^ (rl & 1);                                                                                    \
            if (SfromNb && !a_nosc) Sbn = ld_sc1(&po_[roffSb + qs * H + j]);                                                  \
            else Sbn = pi[roffSb + qs * H + j];                                                                               \
        }                                                                                                                     \
    }                                                                                                                         \
    if (dep && s + 2 >= z0 && s + 2 < z1) pollv = ld_sc1(dep); /* for the next step's check */
        // red stage, plane zr: row r of this lane
        const real Nl_ = red_c ? ey[es_c][wyN][wx][1][lane] : (real)0, Sl_ = red_c ? ey[es_c][wyS][wx][0][lane] : (real)0;
        const real Nedge = wy > 0 ? Nl_ : Nc;
        const real Sedge = wy < WY - 1 ? Sl_ : Sc;
        real* const rs = ringp + sl_r * (TY * TX) + mycell;
#define MGX_P_RED_ROW(r)                                                                                                      \
    if (red_c) {                                                                                                              \
        const int qr = q ^ ((r)&1);                                                                                           \
        const real fromR = ex[es_c][wy][wxR][0][r], fromL = ex[es_c][wy][wxL][1][r];                                          \
        real nb = qr ? __shfl_down(cc[r], 1, 64) : __shfl_up(cc[r], 1, 64);                                                   \
        if (qr) {                                                                                                             \
            if (lane == 63) nb = fromR;                                                                                       \
            if (rimR) nb = xc[r];                                                                                             \
        } else {                                                                                                              \
            if (lane == 0) nb = fromL;                                                                                        \
            if (rimL) nb = xc[r]; /* x = 0: not a neighbour but the point itself; the result is not used */                  \
        }                                                                                                                     \
        const real W = qr ? cc[r] : nb;                                                                                       \
        const real E = qr ? nb : cc[r];                                                                                       \
        const real N = (r) == 0 ? Nedge : cc[(r) > 0 ? (r)-1 : 0];                                                            \
        const real S = (r) == R - 1 ? Sedge : cc[(r) < R - 1 ? (r) + 1 : 0];                                                  \
        oc[r] = a_nored ? W + E + N + S + cp[r] + cu[r] + fc[r]                                                               \
                        : relax3d_point_rd<real>(W, E, N, S, cp[r], cu[r], fc[r], hx2, hy2, hz2, rd);                         \
        rs[(r)*TX] = (qr | j) ? oc[r] : xc[r]; /* x = 0 keeps its boundary value */                                          \
    }
        // black stage, plane s
        const real* const rb = ringp + sl_b * (TY * TX) + mycell;
        const real* const ru = ringp + sl_u * (TY * TX) + mycell;
        real U[R];
        real Nring = 0, Sring = 0;
#define MGX_P_BLK_PRE                                                                                                         \
    _Pragma("unroll") for (int r = 0; r < R; r++) U[r] = ru[r * TX]; /* red plane s + 1, own entry (written D - 1 steps ago) */ \
    if (blk_c) {                                                                                                              \
        if (wy > 0) Nring = rb[-TX];                                                                                          \
        if (wy < WY - 1) Sring = rb[R * TX];                                                                                  \
    }
#define MGX_P_BLK_ROW(r)                                                                                                      \
    if (blk_c) {                                                                                                              \
        const int qb = qb0 ^ ((r)&1);                                                                                         \
        real side = qb ? rb[(r)*TX + (jn < TX - 1 ? 1 : 0)] : rb[(r)*TX - (jn > 0 ? 1 : 0)];                                  \
        if (qb && rimR) side = xb[r];                                                                                         \
        const real Wv = qb ? Pw[r] : side;                                                                                    \
        const real Ev = qb ? side : Pw[r];                                                                                    \
        real Nv = (r) == 0 ? (Nmem ? Nb : Nring) : Pw[(r) > 0 ? (r)-1 : 0];                                                   \
        real Sv = (r) == R - 1 ? Sring : Pw[(r) < R - 1 ? (r) + 1 : 0];                                                       \
        if (Smem && (r) == rl) Sv = Sb;                                                                                       \
        const real Dv = s == 1 ? zb[r] : Dw[r];                                                                               \
        const real Uv = s == sz - 2 ? zb[r] : U[r];                                                                           \
        ob[r] = a_noblk ? Wv + Ev + Nv + Sv + Dv + Uv + fb[r]                                                                 \
                        : relax3d_point_rd<real>(Wv, Ev, Nv, Sv, Dv, Uv, fb[r], hx2, hy2, hz2, rd);                           \
    }
#define MGX_P_BLK_POST                                 \
    _Pragma("unroll") for (int r = 0; r < R; r++) {   \
        Dw[r] = Pw[r];                                 \
        Pw[r] = U[r];                                  \
    }
#define MGX_SCHED __builtin_amdgcn_sched_barrier(0)
        if constexpr (!ILV) {
            __builtin_amdgcn_s_setprio(3);
            MGX_P_STORE_RED
            MGX_P_STORE_BLK
            MGX_P_LOAD_CN
            MGX_P_LOAD_FN
            MGX_P_LOAD_BLK
            if (red_more) publish(es_n, cu);
            __builtin_amdgcn_s_setprio(0);
            MGX_STAMP(0);
#pragma unroll
            for (int r = 0; r < R; r++) { MGX_P_RED_ROW(r) }
            MGX_STAMP(2);
            MGX_P_BLK_PRE
#pragma unroll
            for (int r = 0; r < R; r++) { MGX_P_BLK_ROW(r) }
            MGX_P_BLK_POST
        } else {
            static_assert(!ILV || R == 4 || R == 2, "interleaved order is written for 2 or 4 rows per lane");
            MGX_P_STORE_RED
            MGX_P_LOAD_CN
            MGX_SCHED;
            MGX_P_RED_ROW(0)
            MGX_SCHED;
            MGX_P_LOAD_FN
            MGX_SCHED;
            MGX_P_RED_ROW(1)
            MGX_SCHED;
            MGX_P_LOAD_BLK
            MGX_SCHED;
            if constexpr (R == 4) {
                MGX_P_RED_ROW(2)
                MGX_SCHED;
                MGX_P_STORE_BLK
                MGX_SCHED;
                MGX_P_RED_ROW(3)
            } else {
                MGX_P_STORE_BLK
            }
            if (red_more) publish(es_n, cu);
            MGX_SCHED;
            MGX_STAMP(2);
            MGX_P_BLK_PRE
#pragma unroll
            for (int r = 0; r < R; r++) { MGX_P_BLK_ROW(r) }
            MGX_P_BLK_POST
        }
#undef MGX_P_STORE_RED
#undef MGX_P_STORE_BLK
#undef MGX_P_LOAD_CN
#undef MGX_P_LOAD_FN
#undef MGX_P_LOAD_BLK
#undef MGX_P_RED_ROW
#undef MGX_P_BLK_PRE
#undef MGX_P_BLK_ROW
#undef MGX_P_BLK_POST
#undef MGX_SCHED
        MGX_STAMP(3);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        MGX_STAMP(4);
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): this step's loads have arrived, its stores are in memory
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        MGX_STAMP(5);
        // Behind this step's barrier every wave has drained the stores of the step before (red plane zr - 2): publish it.
        if (w == 0 && lane == 0 && zr - 2 >= max(z0, zrf) && zr - 2 <= min(z1 - 1, zrl)) st_sc1(myflag, epbase + (u64)(zr - 2 + 1));
        if (red_c) {
#pragma unroll
            for (int r = 0; r < R; r++) {
                cp[r] = cc[r];
                cc[r] = cu[r];
                cu[r] = cn[r];
                fc[r] = fn[r];
                xc[r] = xn[r];
                op[r] = oc[r];
            }
            Nc = Nn;
            Sc = Sn;
            pv += sxy;
            pf += sxy;
            q ^= 1;
            es_c = es_n;
            es_n ^= 1;
        }
#pragma unroll
        for (int r = 0; r < R; r++) {
            fb[r] = fbn[r];
            xb[r] = xbn[r];
            zb[r] = zbn[r];
            obp[r] = ob[r];
        }
        Nb = Nbn;
        Sb = Sbn;
        sl_r = sl_r + 1 == NR ? 0 : sl_r + 1;
        sl_b = sl_b + 1 == NR ? 0 : sl_b + 1;
        sl_u = sl_u + 1 == NR ? 0 : sl_u + 1;
    }
    // the last black plane (z1 - 1); the last red plane a run stores (z1 - 1 at most) went out D steps ago
    {
        const int qb0 = (c0 + 1 + y0 + (z1 - 1)) & 1;
        real* p = vout + (size_t)(z1 - 1) * g.PL;
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int qb = qb0 ^ (r & 1);
            if (lane_on && (qb | j) && r < nrows) __builtin_nontemporal_store(obp[r], &p[roff[r] + qb * H + j]);
        }
    }
#undef MGX_LOAD_RIM
    if (DBG && dbg && lane == 0) {
#pragma unroll
        for (int k = 0; k < 8; k++) dbg[((size_t)blockIdx.x * (WX * WY) + w) * 8 + k] = tph[k];
    }
#undef MGX_STAMP
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

// ------------------------------------------------------------------ one launch per sweep on the cache-resident levels
// Levels of 33 ... 129 points per row hold a few MB: a colour pass is a 3-5 us launch that moves almost nothing, and a
// V(2,2) cycle spends ten such launches per level.  Here ONE launch does a whole red+black sweep, out of place: a workgroup
// loads the old black values of its tile (all of x, TY rows, TZ planes) plus TWO rows / planes around it into LDS, computes
// red on the tile plus ONE row / plane around it (the red values its black points need from its neighbours' tiles are
// recomputed, from the same inputs with the same expression: same bits), then black on the tile.  LDS holds, per row, one
// entry per x-pair: B = the old black value of the pair, Rd = its new red value.  In row (y, z) the red point of pair i is
// x = 2 i + q, its x-neighbours are the pair's own black value and that of pair i + 1 (q = 1) or i - 1 (q = 0); its y / z
// neighbours are the black values of the SAME pair index in the adjacent rows (the parity flips with y and z).  Pair M - 1
// is the single boundary point x = sx - 1.  ZERO: the input counts as all zeros and is not read (the first sweep on a
// coarse level, N3/MultiGrid3D.cpp:634 + :626): nothing is staged, B is never read, the first barrier is gone.
// LDS words read, per phase, against the words written before them (t < n covers every index of a phase: the host admits a
// tile only if its (TY + 2)(TZ + 2) M red items fit 4 per thread):
//   red, interior point of row (y, z) in [ra, rb) x [sa, sb): B at pair i, i +- 1 (x >= 1 resp. x <= sx - 2 keeps i +- 1 in
//     [0, M)), rows y +- 1 in [ya, yb), planes z +- 1 in [za, zb) -- interior means 1 <= y <= sy - 2, so y - 1 >= max(y0 - 2, 0)
//     = ya and y + 1 <= min(y1, sy - 1) < yb, likewise z; staging writes EVERY (i, y, z) of [0, M) x [ya, yb) x [za, zb),
//     including the pair without a point (2 i + q = sx: 0);
//   black, point of the tile: Rd at pair i, i +- 1, rows y +- 1 in [y0 - 1, y1] = [ra, rb), planes likewise; the red phase
//     writes Rd for EVERY item (kinds 0 ... 3: no point, boundary value, halo, own).
// Round 3 shipped ZERO as a run-time flag because a template-specialised form had produced wrong values next to boundary
// faces that depended on what the previous launch had left in LDS.  That form was never committed and cannot be diffed; this
// one is specialised again and runs the whole GPU suite with the LDS of every CU filled with NaN patterns before EVERY
// launch (mgx_test_set_lds_poison, tests/conftest.py; tests/test_gpu_sweep.py: test_lds_poisoning_reaches_every_cu proves
// the poison arrives, test_mid_from_zero_specialisation_at_129_rows is the case that had failed): no read of an unwritten
// word exists in it.  What the failing build did differently is not recoverable (the run-time-flag form of the same day
// passed the same case, so it was specific to how that specialisation was written, not to the tile geometry); the class of
// bug is now caught deterministically instead.  sweep3d_mid_tile also bounds the red items of a tile itself (4 per thread).
template <class real, bool ZERO>
__global__ void __launch_bounds__(1024) sweep3d_xs_mid_kernel(const real* __restrict__ vin, real* __restrict__ vout,
                                                              const real* __restrict__ f, int sx, int sy, int sz, real hx2, real hy2,
                                                              real hz2, int c0, int TY, int TZ, int gy) {
    extern __shared__ __attribute__((aligned(16))) char smem_[];
    const double rd = relax3d_rd<real>(hx2, hy2, hz2);
    const Geo<XSplit, real> g(sx, sy);
    const int H = g.H, M = (sx + 1) >> 1;
    const int by = blockIdx.x % gy, bz = blockIdx.x / gy;
    const int y0 = 1 + by * TY, y1 = min(y0 + TY, sy - 1), z0 = 1 + bz * TZ, z1 = min(z0 + TZ, sz - 1);
    const int ya = max(y0 - 2, 0), yb = min(y1 + 2, sy), za = max(z0 - 2, 0), zb = min(z1 + 2, sz);  // old black
    const int ra = max(y0 - 1, 0), rb = min(y1 + 1, sy), sa = max(z0 - 1, 0), sb = min(z1 + 1, sz);  // new red
    const int BW = M, BP = M * (TY + 4), RW = M, RP = M * (TY + 2);  // LDS strides (pairs, rows, planes)
    real* B = (real*)smem_;
    real* Rd = B + (size_t)BP * (TZ + 4);
    const int nt = blockDim.x, tid = threadIdx.x;
    const SmallDiv dM(M);
    // The red and the black phase each have at most UC items per thread (the host checks it).  Their global loads (f at the
    // points they relax, the boundary values the red phase passes on) do not depend on anything computed here, so ALL of them
    // are requested before the old black values are staged: one memory round trip for the whole launch instead of three in a
    // row (measured: 65^3 8.4 -> see DESIGN section 5, 33^3 likewise; the launch is latency, not bytes).
    constexpr int UC = 4;
    real fr[UC], fb[UC];
    int rli[UC], rbi[UC], rkind[UC], rg[UC];  // red items; rkind: -1 none, 0 no point (x = sx), 1 boundary (fr = its value), 2 interior halo, 3 interior own
    int bpi[UC], bg[UC];                      // black items; bpi < 0: none
    {
        const int NY = rb - ra, n = (sb - sa) * NY * M;
        const SmallDiv dNY(NY);
#pragma unroll
        for (int u = 0; u < UC; u++) {
            const int t = tid + u * nt;
            rkind[u] = -1;
            fr[u] = 0;
            rli[u] = rbi[u] = rg[u] = 0;
            if (t < n) {
                const int r = dM(t), i = t - r * M, zz = dNY(r), yy = r - zz * NY, y = ra + yy, z = sa + zz;
                const int q = (c0 + y + z) & 1, x = 2 * i + q;
                rli[u] = i + yy * RW + zz * RP;
                rbi[u] = (i + (y - ya) * BW + (z - za) * BP) * 2 + q;
                rkind[u] = 0;
                if (x < sx) {
                    rg[u] = (int)(g.row(y, z) + q * H + i);
                    if (x == 0 || x == sx - 1 || y == 0 || y == sy - 1 || z == 0 || z == sz - 1) {
                        rkind[u] = 1;
                        if (!ZERO) fr[u] = vin[rg[u]];
                    } else {
                        rkind[u] = (y >= y0 && y < y1 && z >= z0 && z < z1) ? 3 : 2;
                        fr[u] = f[rg[u]];
                    }
                }
            }
        }
    }
    {
        const int NY = y1 - y0, n = (z1 - z0) * NY * M;
        const SmallDiv dNY(NY);
#pragma unroll
        for (int u = 0; u < UC; u++) {
            const int t = tid + u * nt;
            bpi[u] = -1;
            fb[u] = 0;
            bg[u] = 0;
            if (t < n) {
                const int r = dM(t), i = t - r * M, zz = dNY(r), y = y0 + r - zz * NY, z = z0 + zz;
                const int q = (c0 + 1 + y + z) & 1, x = 2 * i + q;
                if (x >= 1 && x < sx - 1) {
                    bg[u] = (int)(g.row(y, z) + q * H + i);
                    bpi[u] = (i + (y - ra) * RW + (z - sa) * RP) * 2 + q;
                    fb[u] = f[bg[u]];
                }
            }
        }
    }
    // 1. the old black values of the rows [ya, yb) x [za, zb), in chunks of U per thread: first all loads of a chunk, then its
    // LDS stores.  ZERO: nothing is staged and B is never read -- every old value is the constant 0.
    constexpr int U = 6;
    if (!ZERO) {
        const int NY = yb - ya, n = (zb - za) * NY * M;
        const SmallDiv dNY(NY);
        for (int base = tid; base < n; base += nt * U) {
            real tmp[U];
            int li[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int t = base + u * nt;
                li[u] = -1;
                tmp[u] = 0;
                if (t < n) {
                    const int r = dM(t), i = t - r * M, zz = dNY(r), yy = r - zz * NY, y = ya + yy, z = za + zz;
                    const int q = (c0 + 1 + y + z) & 1;
                    li[u] = i + yy * BW + zz * BP;
                    if (2 * i + q < sx) tmp[u] = vin[g.row(y, z) + q * H + i];
                }
            }
#pragma unroll
            for (int u = 0; u < U; u++)
                if (li[u] >= 0) B[li[u]] = tmp[u];
        }
        __syncthreads();
    }
    // 2. red on the tile and one row / plane around it (boundary points keep their value)
#pragma unroll
    for (int u = 0; u < UC; u++) {
        if (rkind[u] < 0) continue;
        real val = (real)0;
        if (!ZERO && rkind[u] == 1) val = fr[u];
        if (rkind[u] >= 2) {
            if (ZERO) {
                const real z = (real)0;
                val = relax3d_point_rd<real>(z, z, z, z, z, z, fr[u], hx2, hy2, hz2, rd);
            } else {
                const int q = rbi[u] & 1;
                const real* b = B + (rbi[u] >> 1);  // the pair's own black value (x + 1 - 2 q)
                const real side = q ? b[1] : b[-1];
                val = relax3d_point_rd<real>(q ? b[0] : side, q ? side : b[0], b[-BW], b[BW], b[-BP], b[BP], fr[u], hx2, hy2, hz2, rd);
            }
            if (rkind[u] == 3) vout[rg[u]] = val;
        }
        Rd[rli[u]] = val;
    }
    __syncthreads();
    // 3. black on the tile
#pragma unroll
    for (int u = 0; u < UC; u++) {
        if (bpi[u] < 0) continue;
        const int q = bpi[u] & 1;
        const real* p = Rd + (bpi[u] >> 1);  // the pair's own red value
        const real side = q ? p[1] : p[-1];
        vout[bg[u]] = relax3d_point_rd<real>(q ? p[0] : side, q ? side : p[0], p[-RW], p[RW], p[-RP], p[RP], fb[u], hx2, hy2, hz2, rd);
    }
}

// ------------------------------------------------------------------ boundary faces of v -> w
// The sweep kernel never writes boundary points; a ping-pong partner must carry v's boundary values before it becomes
// the input of the next sweep (or the result).  One workgroup per (plane, part): the planes z = 0 and sz - 1 whole, of the
// others the rows y = 0 and sy - 1 and the entries x = 0 and sx - 1 of every row.
template <class real>
__global__ void __launch_bounds__(256) copy_rim3d_xs_kernel(const real* __restrict__ v, real* __restrict__ w, int sx, int sy, int sz) {
    const Geo<XSplit, real> g(sx, sy);
    const int z = blockIdx.x, part = blockIdx.y, nparts = gridDim.y, t = threadIdx.x;
    const size_t base = (size_t)z * g.PL;
    const int M = (sx + 1) >> 1;
    if (z == 0 || z == sz - 1) {
        for (size_t i = (size_t)part * 256 + t; i < g.PL; i += (size_t)nparts * 256) w[base + i] = v[base + i];
        return;
    }
    for (int i = part * 256 + t; i < g.P; i += nparts * 256) {
        w[base + i] = v[base + i];
        w[base + (size_t)(sy - 1) * g.P + i] = v[base + (size_t)(sy - 1) * g.P + i];
    }
    for (int y = part * 256 + t; y < sy; y += nparts * 256) {
        const size_t r = base + (size_t)y * g.P;
        w[r] = v[r];                  // x = 0
        w[r + M - 1] = v[r + M - 1];  // x = sx - 1 (even): index M - 1 of the even half
    }
}

// ------------------------------------------------------------------ host side

int sweep_state(mgx_ctx* ctx, SweepSync* out) {
    if (!ctx->sweep_dev) {
        const size_t bytes = 64 + sizeof(u64) * SWEEP_MAX_WG;
        void* d = nullptr;
        MGX_HIP(hipMalloc(&d, bytes));
        hipError_t e = hipMemsetAsync(d, 0, bytes, ctx->compute);
        const u64 one = 1;
        if (e == hipSuccess) e = hipMemcpyAsync(d, &one, sizeof one, hipMemcpyHostToDevice, ctx->compute);  // epoch starts at 1
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->compute);
        unsigned* a = nullptr;
        if (e == hipSuccess) e = hipHostMalloc((void**)&a, 64, hipHostMallocMapped);
        if (e != hipSuccess) {
            (void)hipFree(d);
            return fail(MGX_ERR_HIP, "sweep state: %s", hipGetErrorString(e));
        }
        *a = 0;
        ctx->sweep_dev = d;
        ctx->sweep_abort = a;
    }
    char* d = (char*)ctx->sweep_dev;
    out->epoch = (u64*)d;
    out->done = (unsigned*)(d + 8);
    out->flags = (u64*)(d + 64);
    void* adev = nullptr;
    MGX_HIP(hipHostGetDevicePointer(&adev, ctx->sweep_abort, 0));
    out->abort = (unsigned*)adev;
    out->spin_limit = ctx->sync_spin_limit;
    out->fault = ctx->handoff_fault;
    return MGX_OK;
}

constexpr int SWEEP_MID = 1;  // shape code of sweep3d_xs_mid_kernel

// tile (TY rows x TZ planes) of the mid-level kernel: at least ~128 workgroups where the level has them, LDS within 150 KB
template <class real>
static bool sweep3d_mid_tile(int sx, int sy, int sz, int* TY, int* TZ, int max_row = 65) {
    // measured (tools/level_timing.py, 513^3 cycle, MI355X): 33^3 level 44 -> 37 us per visit, 65^3 equal, 129^3 78 -> 100 us (256
    // workgroups of 1024 threads doing 2.6 x the arithmetic lose against two 5 us passes): rows of at most 65 points only
    // ("relax3d.fused_mid" = 2 lets rows of 129 points in: the size at which a specialised from-zero form once failed, kept testable)
    if (sx > max_row || sx < 33 || sy < 9 || sz < 9) return false;
    int ty = 8, tz = 8;
    auto wgs = [&]() { return ceil_div(sy - 2, ty) * ceil_div(sz - 2, tz); };
    auto bytes = [&]() { return (size_t)((sx + 1) / 2) * ((ty + 4) * (tz + 4) + (ty + 2) * (tz + 2)) * sizeof(real); };
    auto items = [&]() { return (ty + 2) * (tz + 2) * ((sx + 1) / 2); };  // red items of a tile: at most 4 per thread of 1024
    if (wgs() < 128 || bytes() > 150 * 1024 || items() > 4096) tz = 4;
    if (wgs() < 128 && sy - 2 > 4) ty = 4;
    if (bytes() > 150 * 1024 || items() > 4096) return false;
    *TY = ty;
    *TZ = tz;
    return true;
}

// shape (WX, WY, R) of the fused sweep for a level, 0 if the level does not take it
template <class real>
static int sweep3d_shape(const mgx_ctx* ctx, int sx, int sy, int sz) {
    int ty, tz;
    if (ctx->sweep_mid && sweep3d_mid_tile<real>(sx, sy, sz, &ty, &tz, ctx->sweep_mid >= 2 ? 129 : 65)) return SWEEP_MID;
    if (!ctx->sweep_fused || ctx->nranks > 1 || ctx->local_group) return 0;  // thread-ranks share one GPU: co-residency is not given
    if (ctx->handoff_broken || !ctx->gpu_exclusive) return 0;
    const int M = (sx + 1) / 2;
    if (M - 1 != 256) return 0;          // tiles span the x-extent: 513-point rows (4 waves of 64 pairs)
    if (sy - 2 < 64 || sz - 2 < 64) return 0;
    // one workgroup per CU (132 KB of LDS), all resident together: a level with more 8-row tiles than CUs (513 x 4097 x N on 256 CUs)
    // runs colour passes -- the same test as sweep3d_launch_shape's, so that the takes-predicates and the launch agree
    if (ceil_div(sy - 2, 8) > (ctx->num_cus < SWEEP_MAX_WG ? ctx->num_cus : SWEEP_MAX_WG)) return 0;
    return 424;
}

template <class real, int WX, int WY, int R, int D>
static int sweep3d_launch_shape(mgx_ctx* ctx, const real* vin, real* vout, const real* f, int sx, int sy, int sz, real hx2, real hy2,
                                real hz2, int c0) {
    const int zb = 1, ze = sz - 1;
    const int gy = ceil_div(sy - 2, WY * R);
    // one resident round of workgroups: as many z-runs per tile as there are CUs for, but no run shorter than 32 planes
    // (a run spends D + 1 steps filling its pipeline and computes two red planes it does not store)
    int nz = max(1, min(ctx->num_cus / gy, (ze - zb) / 32));
    int zchunk = ceil_div(ze - zb, nz);
    if (ctx->relax_zchunk > 0) zchunk = max(2, ctx->relax_zchunk);  // "relax3d.zchunk" (tests: runs shorter than the lead)
    const int gz = ceil_div(ze - zb, zchunk);
    MGX_REQUIRE(gy * gz <= ctx->num_cus && gy * gz <= SWEEP_MAX_WG, MGX_ERR_SIZE, "sweep3d: %d workgroups cannot be resident together on %d CUs",
                gy * gz, ctx->num_cus);
    SweepSync sync;
    MGX_TRY_RET(sweep_state(ctx, &sync));
    const bool fnt = (size_t)sx * sy * (size_t)sz * sizeof(real) > ((size_t)256 << 20);
    const dim3 grid((unsigned)gy * gz), block(64, WX * WY, 1);
    const int xcd = ctx->relax_xcd == 1 ? 1 : 0;
    snprintf(ctx->last_relax_kernel, sizeof ctx->last_relax_kernel, "sweep3d_xs_kernel<%s,%d,%d,%d,%d,%s%s>", sizeof(real) == 8 ? "double" : "float",
             WX, WY, R, D, fnt ? "true" : "false", ctx->sweep_ilv ? ",0,true" : "");
#ifdef MGX_DIAGNOSTICS
    if (ctx->sweep_dbg) {  // cycle stamps / ablations (tools/sweep_stamps.py)
        void* ws = nullptr;
        MGX_TRY_RET(workspace(ctx, (size_t)gy * gz * WX * WY * 8 * sizeof(long long), &ws));
        if (ctx->sweep_ilv)
            MGX_LAUNCH((sweep3d_xs_kernel<real, WX, WY, R, D, true, 1, true>), grid, block, 0, ctx->compute, vin, vout, f, sx, sy, sz, zb,
                               ze, hx2, hy2, hz2, c0, zchunk, gy, xcd, sync, (long long*)ws, ctx->sweep_dbg >> 1);
        else
            MGX_LAUNCH((sweep3d_xs_kernel<real, WX, WY, R, D, true, 1>), grid, block, 0, ctx->compute, vin, vout, f, sx, sy, sz, zb, ze,
                               hx2, hy2, hz2, c0, zchunk, gy, xcd, sync, (long long*)ws, ctx->sweep_dbg >> 1);
        return MGX_OK;
    }
#endif
    long long* const nodbg = nullptr;
    if (ctx->sweep_ilv) {  // memory instructions interleaved with the arithmetic ("relax3d.fused_ilv")
        if (fnt)
            MGX_LAUNCH((sweep3d_xs_kernel<real, WX, WY, R, D, true, 0, true>), grid, block, 0, ctx->compute, vin, vout, f, sx, sy, sz, zb,
                               ze, hx2, hy2, hz2, c0, zchunk, gy, xcd, sync, nodbg, 0);
        else
            MGX_LAUNCH((sweep3d_xs_kernel<real, WX, WY, R, D, false, 0, true>), grid, block, 0, ctx->compute, vin, vout, f, sx, sy, sz, zb,
                               ze, hx2, hy2, hz2, c0, zchunk, gy, xcd, sync, nodbg, 0);
        return MGX_OK;
    }
    if (fnt)
        MGX_LAUNCH((sweep3d_xs_kernel<real, WX, WY, R, D, true>), grid, block, 0, ctx->compute, vin, vout, f, sx, sy, sz, zb, ze, hx2,
                           hy2, hz2, c0, zchunk, gy, xcd, sync);
    else
        MGX_LAUNCH((sweep3d_xs_kernel<real, WX, WY, R, D, false>), grid, block, 0, ctx->compute, vin, vout, f, sx, sy, sz, zb, ze, hx2,
                           hy2, hz2, c0, zchunk, gy, xcd, sync);
    return MGX_OK;
}

template <class real>
static int sweep3d_mid_launch(mgx_ctx* ctx, const real* vin, real* vout, const real* f, int sx, int sy, int sz, real hx2, real hy2, real hz2,
                              bool zero) {
    int TY = 8, TZ = 8;
    MGX_REQUIRE(sweep3d_mid_tile<real>(sx, sy, sz, &TY, &TZ, 129), MGX_ERR_SIZE, "sweep3d_mid: level %d x %d x %d does not take the kernel", sx, sy, sz);
    const size_t lds = (size_t)((sx + 1) / 2) * ((TY + 4) * (TZ + 4) + (TY + 2) * (TZ + 2)) * sizeof(real);
    const int gy = ceil_div(sy - 2, TY), gz = ceil_div(sz - 2, TZ);
    const int threads = (size_t)TY * TZ * ((sx + 1) / 2) >= 2048 ? 1024 : 512;
    MGX_REQUIRE((TY + 2) * (TZ + 2) * ((sx + 1) / 2) <= 4 * threads, MGX_ERR_SIZE, "sweep3d_mid: tile %d x %d of %d-point rows has more than 4 items per thread",
                TY, TZ, sx);
    if (lds > 64 * 1024) {
        MGX_HIP(hipFuncSetAttribute((const void*)sweep3d_xs_mid_kernel<real, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        MGX_HIP(hipFuncSetAttribute((const void*)sweep3d_xs_mid_kernel<real, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    snprintf(ctx->last_relax_kernel, sizeof ctx->last_relax_kernel, "sweep3d_xs_mid_kernel<%s>%s", sizeof(real) == 8 ? "double" : "float",
             zero ? " (from zero)" : "");
    if (zero)
        MGX_LAUNCH((sweep3d_xs_mid_kernel<real, true>), dim3(gy * gz), dim3(threads), lds, ctx->compute, vin, vout, f, sx, sy, sz, hx2, hy2,
                   hz2, 0, TY, TZ, gy);
    else
        MGX_LAUNCH((sweep3d_xs_mid_kernel<real, false>), dim3(gy * gz), dim3(threads), lds, ctx->compute, vin, vout, f, sx, sy, sz, hx2, hy2,
                   hz2, 0, TY, TZ, gy);
    return MGX_OK;
}

// one red+black sweep vin -> vout (boundary entries of vout are not written)
template <class real>
static int sweep3d_launch(mgx_ctx* ctx, int shape, const real* vin, real* vout, const real* f, int sx, int sy, int sz, real hx2, real hy2,
                          real hz2, bool zero = false) {
    const int lead = ctx->sweep_lead;
    if (shape == SWEEP_MID) return sweep3d_mid_launch<real>(ctx, vin, vout, f, sx, sy, sz, hx2, hy2, hz2, zero);
    MGX_REQUIRE(!zero, MGX_ERR_INVALID, "sweep3d: no from-zero form of shape %d", shape);
    switch (shape) {
        case 424:
            if (lead == 5) return sweep3d_launch_shape<real, 4, 2, 4, 5>(ctx, vin, vout, f, sx, sy, sz, hx2, hy2, hz2, 0);
            if (lead == 7) return sweep3d_launch_shape<real, 4, 2, 4, 7>(ctx, vin, vout, f, sx, sy, sz, hx2, hy2, hz2, 0);
            return sweep3d_launch_shape<real, 4, 2, 4, 6>(ctx, vin, vout, f, sx, sy, sz, hx2, hy2, hz2, 0);
        default: return fail(MGX_ERR_INVALID, "sweep3d: unknown shape %d", shape);
    }
}

template <class real>
int relax3d_xs_colour_passes(mgx_ctx* ctx, real* v, const real* f, const int n[3], const real h[3], int ncycles);  // mgx_kernels3d.hip

// `ncycles` red-black sweeps of v (N3/MultiGrid3D.cpp:489-567), x-split layout, with a second array w of the same size
// as ping-pong partner: result in v, w is scratch.  w_rim_valid != 0: the caller vouches that the boundary entries of w
// already equal those of v (a hierarchy that has run this before and has not touched either boundary since).
template <class real>
int relax3d_xs_pp(mgx_ctx* ctx, real* v, real* w, const real* f, const int n[3], const real h[3], int ncycles, int w_rim_valid) {
    MGX_REQUIRE(ctx && v && w && f && h && n, MGX_ERR_INVALID, "relax_pp3d: NULL argument");
    MGX_REQUIRE(v != w, MGX_ERR_INVALID, "relax_pp3d: v and w must differ");
    MGX_USE(ctx);
    MGX_REQUIRE(ncycles >= 0, MGX_ERR_INVALID, "relax_pp3d: ncycles = %d < 0", ncycles);
    for (int d = 0; d < 3; d++) MGX_REQUIRE(valid_size(n[d]), MGX_ERR_SIZE, "relax_pp3d: size[%d] = %d is not 2^k+1 >= 3", d, n[d]);
    // a call the resident kernel takes (all passes in one launch, in place: mgx_resident3d.hip) needs no partner array
    const int shape = ncycles >= 2 && !relax3d_resident_takes(ctx, n, ncycles) ? sweep3d_shape<real>(ctx, n[0], n[1], n[2]) : 0;
    if (!shape) return relax3d_xs_colour_passes<real>(ctx, v, f, n, h, ncycles);
    const real hx2 = h[0] * h[0], hy2 = h[1] * h[1], hz2 = h[2] * h[2];  // N3/MultiGrid3D.cpp:498-500
    int k = ncycles;
    if (k & 1) {  // an odd sweep runs as two colour passes in place, the rest in ping-pong pairs
        MGX_TRY_RET(relax3d_xs_colour_passes<real>(ctx, v, f, n, h, 1));
        k--;
    }
    if (!w_rim_valid) {
        MGX_LAUNCH((copy_rim3d_xs_kernel<real>), dim3(n[2], 4), dim3(256), 0, ctx->compute, (const real*)v, w, n[0], n[1], n[2]);
        MGX_LAUNCH_CHECK();
    }
    for (; k > 0; k -= 2) {
        MGX_TRY_RET(sweep3d_launch<real>(ctx, shape, v, w, f, n[0], n[1], n[2], hx2, hy2, hz2));
        MGX_TRY_RET(sweep3d_launch<real>(ctx, shape, w, v, f, n[0], n[1], n[2], hx2, hy2, hz2));
    }
    MGX_LAUNCH_CHECK();
    return MGX_OK;
}

template int relax3d_xs_pp<float>(mgx_ctx*, float*, float*, const float*, const int[3], const float[3], int, int);
template int relax3d_xs_pp<double>(mgx_ctx*, double*, double*, const double*, const int[3], const double[3], int, int);

template <class real>
int relax3d_xs_from_zero(mgx_ctx* ctx, real* v, const real* f, const int n[3], const real h[3], int ncycles, int rim_is_zero);  // mgx_kernels3d.hip

template <class real>
static bool from_zero_pp_takes(const mgx_ctx* ctx, const int n[3], int ncycles, int rim_is_zero) {
    return ncycles >= 2 && (ncycles & 1) == 0 && rim_is_zero && ctx->relax_zero_first && !relax3d_resident_takes(ctx, n, ncycles) &&
           sweep3d_shape<real>(ctx, n[0], n[1], n[2]) == SWEEP_MID;
}

// relax_from_zero with a ping-pong partner: v := 0 (N3/MultiGrid3D.cpp:634), then `ncycles` sweeps (:626).  On the
// cache-resident levels, with the boundary of v known to be zero and an even sweep count, the first sweep does not read v
// at all (sweep3d_xs_mid_kernel with zero = 1) and every sweep is one launch; otherwise mgx3dxs_relax_from_zero.
template <class real>
int relax3d_xs_from_zero_pp(mgx_ctx* ctx, real* v, real* w, const real* f, const int n[3], const real h[3], int ncycles, int rim_is_zero,
                            int w_rim_valid) {
    MGX_REQUIRE(ctx && v && w && f && h && n, MGX_ERR_INVALID, "relax_from_zero_pp3d: NULL argument");
    MGX_REQUIRE(v != w, MGX_ERR_INVALID, "relax_from_zero_pp3d: v and w must differ");
    MGX_USE(ctx);
    for (int d = 0; d < 3; d++) MGX_REQUIRE(valid_size(n[d]), MGX_ERR_SIZE, "relax_from_zero_pp3d: size[%d] = %d is not 2^k+1 >= 3", d, n[d]);
    const int shape = from_zero_pp_takes<real>(ctx, n, ncycles, rim_is_zero) ? SWEEP_MID : 0;
    if (shape != SWEEP_MID) return relax3d_xs_from_zero<real>(ctx, v, f, n, h, ncycles, rim_is_zero);
    const real hx2 = h[0] * h[0], hy2 = h[1] * h[1], hz2 = h[2] * h[2];
    if (!w_rim_valid) {  // the boundary of v is zero: so must be w's
        MGX_LAUNCH((copy_rim3d_xs_kernel<real>), dim3(n[2], 4), dim3(256), 0, ctx->compute, (const real*)v, w, n[0], n[1], n[2]);
        MGX_LAUNCH_CHECK();
    }
    for (int k = 0; k < ncycles; k += 2) {
        MGX_TRY_RET(sweep3d_launch<real>(ctx, shape, v, w, f, n[0], n[1], n[2], hx2, hy2, hz2, k == 0));
        MGX_TRY_RET(sweep3d_launch<real>(ctx, shape, w, v, f, n[0], n[1], n[2], hx2, hy2, hz2));
    }
    MGX_LAUNCH_CHECK();
    return MGX_OK;
}

}  // namespace mgx

extern "C" {
int mgx3dxs_relax_from_zero_pp_f32(mgx_ctx* ctx, float* v, float* w, const float* f, const int n[3], const float h[3], int ncycles,
                                   int rim_is_zero, int w_rim_valid) {
    return mgx::relax3d_xs_from_zero_pp<float>(ctx, v, w, f, n, h, ncycles, rim_is_zero, w_rim_valid);
}
int mgx3dxs_relax_from_zero_pp_f64(mgx_ctx* ctx, double* v, double* w, const double* f, const int n[3], const double h[3], int ncycles,
                                   int rim_is_zero, int w_rim_valid) {
    return mgx::relax3d_xs_from_zero_pp<double>(ctx, v, w, f, n, h, ncycles, rim_is_zero, w_rim_valid);
}
// ONE out-of-place sweep vin -> vout (interior points of vout only) by the one-launch kernel of the level, if it has one
// (MGX_ERR_SIZE otherwise); zero != 0: vin counts as all zeros and is not read (cache-resident levels only).  The unit the
// ping-pong drivers above are made of; exposed for tests.
#define MGX_SWEEP_ONCE(SFX, real)                                                                                              \
    int mgx3dxs_sweep_once_##SFX(mgx_ctx* ctx, const real* vin, real* vout, const real* f, const int n[3], const real h[3],    \
                                 int zero) {                                                                                   \
        MGX_REQUIRE(ctx && vin && vout && f && n && h && vin != vout, MGX_ERR_INVALID, "sweep_once: bad argument");            \
        MGX_USE(ctx);                                                                                                          \
        for (int d = 0; d < 3; d++) MGX_REQUIRE(mgx::valid_size(n[d]), MGX_ERR_SIZE, "sweep_once: size[%d] = %d", d, n[d]);    \
        const int shape = mgx::sweep3d_shape<real>(ctx, n[0], n[1], n[2]);                                                     \
        MGX_REQUIRE(shape != 0, MGX_ERR_SIZE, "sweep_once: this level has no one-launch sweep");                               \
        MGX_TRY_RET(mgx::sweep3d_launch<real>(ctx, shape, vin, vout, f, n[0], n[1], n[2], h[0] * h[0], h[1] * h[1], h[2] * h[2], \
                                              zero != 0));                                                                     \
        MGX_LAUNCH_CHECK();                                                                                                    \
        return MGX_OK;                                                                                                         \
    }
MGX_SWEEP_ONCE(f32, float)
MGX_SWEEP_ONCE(f64, double)
#undef MGX_SWEEP_ONCE

int mgx3dxs_relax_from_zero_pp_takes_f32(const mgx_ctx* ctx, const int n[3], int ncycles, int rim_is_zero) {
    return ctx && n && mgx::from_zero_pp_takes<float>(ctx, n, ncycles, rim_is_zero);
}
int mgx3dxs_relax_from_zero_pp_takes_f64(const mgx_ctx* ctx, const int n[3], int ncycles, int rim_is_zero) {
    return ctx && n && mgx::from_zero_pp_takes<double>(ctx, n, ncycles, rim_is_zero);
}
int mgx3dxs_relax_pp_f32(mgx_ctx* ctx, float* v, float* w, const float* f, const int n[3], const float h[3], int ncycles, int w_rim_valid) {
    return mgx::relax3d_xs_pp<float>(ctx, v, w, f, n, h, ncycles, w_rim_valid);
}
int mgx3dxs_relax_pp_f64(mgx_ctx* ctx, double* v, double* w, const double* f, const int n[3], const double h[3], int ncycles,
                         int w_rim_valid) {
    return mgx::relax3d_xs_pp<double>(ctx, v, w, f, n, h, ncycles, w_rim_valid);
}
int mgx3dxs_relax_pp_takes_f32(const mgx_ctx* ctx, const int n[3], int ncycles) {
    return ctx && n && ncycles >= 2 && !mgx::relax3d_resident_takes(ctx, n, ncycles) && mgx::sweep3d_shape<float>(ctx, n[0], n[1], n[2]) != 0;
}
int mgx3dxs_relax_pp_takes_f64(const mgx_ctx* ctx, const int n[3], int ncycles) {
    return ctx && n && ncycles >= 2 && !mgx::relax3d_resident_takes(ctx, n, ncycles) && mgx::sweep3d_shape<double>(ctx, n[0], n[1], n[2]) != 0;
}

#ifdef MGX_DIAGNOSTICS
// diagnostic builds: the per-wave phase cycles the last stamped sweep left in the context's workspace
int mgx_sweep_debug_read(mgx_ctx* ctx, long long* host, size_t count) {
    MGX_REQUIRE(ctx && host, MGX_ERR_INVALID, "NULL argument");
    MGX_USE(ctx);
    MGX_REQUIRE(ctx->scratch && ctx->scratch_bytes >= count * sizeof(long long), MGX_ERR_INVALID, "no stamps recorded");
    MGX_HIP(hipMemcpyAsync(host, ctx->scratch, count * sizeof(long long), hipMemcpyDeviceToHost, ctx->compute));
    MGX_HIP(hipStreamSynchronize(ctx->compute));
    return MGX_OK;
}
#endif

// 0 while no inter-workgroup wait (one-launch sweep, resident Relax) has given up on this context; MGX_ERR_HIP afterwards (results of that
// launch are garbage).  Meaningful after a synchronisation (mgx_ctx_sync calls it).
int mgx_ctx_check(mgx_ctx* ctx) {
    MGX_REQUIRE(ctx, MGX_ERR_INVALID, "ctx is NULL");
    if (ctx->sweep_abort && *(volatile unsigned*)ctx->sweep_abort) {
        ctx->handoff_broken = 1;  // from now on: colour passes
        return mgx::fail(MGX_ERR_HIP, "a kernel whose workgroups wait for each other (one-launch sweep, resident Relax) gave up waiting for a "
                                      "neighbouring workgroup (is the GPU shared with another context or process? then set \"gpu.exclusive\" "
                                      "to 0); the results of that launch are invalid.  This context now runs colour passes instead; "
                                      "mgx_ctx_clear_abort clears the condition");
    }
    return MGX_OK;
}

int mgx_ctx_clear_abort(mgx_ctx* ctx, int reenable) {
    MGX_REQUIRE(ctx, MGX_ERR_INVALID, "ctx is NULL");
    MGX_USE(ctx);
    MGX_HIP(hipStreamSynchronize(ctx->compute));
    MGX_HIP(hipStreamSynchronize(ctx->comm));
    if (ctx->sweep_abort && *(volatile unsigned*)ctx->sweep_abort) ctx->handoff_broken = 1;
    if (ctx->sweep_dev) {
        // every workgroup of an aborted launch still ran to its end (the finished-counter wrapped, the epoch advanced), but nothing is
        // taken on trust here: progress words and counter cleared, the epoch moved past anything a stale tag could carry
        mgx::u64 ep = 0;
        MGX_HIP(hipMemcpy(&ep, ctx->sweep_dev, sizeof ep, hipMemcpyDeviceToHost));
        MGX_HIP(hipMemset(ctx->sweep_dev, 0, 64 + sizeof(mgx::u64) * mgx::SWEEP_MAX_WG));
        ep += 2;
        MGX_HIP(hipMemcpy(ctx->sweep_dev, &ep, sizeof ep, hipMemcpyHostToDevice));
    }
    if (ctx->sweep_abort) *(volatile unsigned*)ctx->sweep_abort = 0;
    if (reenable) ctx->handoff_broken = 0;
    return MGX_OK;
}
}

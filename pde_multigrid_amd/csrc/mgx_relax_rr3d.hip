// mgx_relax_rr3d.hip -- the LAST BLACK PASS of the pre-smoothing, CalculateResidual and Restrict in ONE launch (x-split
// layout, HBM-bound levels).
//
// On the way down a level runs Relax(v1), CalculateResidual, Restrict (N3/MultiGrid3D.cpp:626-632).  As separate launches
// the last black pass streams 1.5 words per point (red in, f of the black points in, black out) and residual + restrict
// streams 2.125 (v, f in; 1/8 out): 3.625.  Here a workgroup marches its tile through a run of planes and, one plane ahead
// of the residual, relaxes the black points itself:
//
//   iteration g:   black  plane g + 1   from the red values of planes g, g + 1, g + 2 (loaded) and f -> v (stored), registers
//                  residual plane g     from the new v of planes g - 1, g, g + 1 (registers) and f
//                  every second iteration: the three (x, z) sub-sums of a coarse plane per fine row, the coarse row one
//                  iteration later (with the sub-sums of the wave below)
//
// so HBM sees the red half of v once, f once, the new black half and the coarse array written: 2.125 words per point.
// Only red values are read from memory and only black values are written, so the update in place is race-free whatever the
// order of the workgroups; the black values just outside a tile (one row, one x-pair, one plane at either end of a run)
// are recomputed by the tile from the reds two deep -- there is nothing to hand over between workgroups.  Every point is
// computed from exactly the values the serial loops would use, with the reference's expressions (relax3d_point,
// residual3d_point, the row sub-sums of Restrict as in residual_restrict3d_xs_kernel): bit-identical results.
//
// Geometry.  A wave owns the fine rows 2c - 1 and 2c of one coarse row c and 64 x-pairs = coarse columns; lanes 0, 1 and 63
// are halo lanes (61 coarse columns per wave).  Wave 0 of a workgroup is a halo wave above the tile (its second row's new
// values are the row above the first producing wave), the last wave one below (its first row's residual sub-sums complete
// the last producing wave's coarse row; it loads the red half of one more row).  Rows of neighbouring waves go through LDS:
// the red half of plane g + 2 and the new black half of plane g + 1 of a wave's first and last row are published in
// iteration g (three / two slots by plane), read in iterations g + 1 and g + 2; ONE barrier per iteration, loads requested
// at the top of an iteration are waited for behind its barrier.  Boundary values (black-half entries on a face of the
// grid, which no pass writes) are loaded by the lanes / rows / planes that hold them.
#include <type_traits>

#include "mgx_internal.hpp"
#include "mgx_kernels3d.hpp"

namespace mgx {

// (plane_rsrc / buf_load / buf_store_nt: loads and stores through buffer descriptors, mgx_kernels3d.hpp)

// DBG (diagnostic builds only): `abl` switches parts of an iteration off for timing (WRONG results): 1 no loads, 2 no stores of
// v, 4 no relax arithmetic, 8 no residual arithmetic, 16 no barrier, 32 no sub-sums / coarse rows.
template <class real, int MODE, int TYW, int DBG = 0>
__global__ void __launch_bounds__(64 * TYW, 4)  // four waves per SIMD (128 VGPRs) whatever the shape: 8-wave workgroups run two to a CU
    relax_rr3d_xs_kernel(const real* __restrict__ vin, real* __restrict__ vout, const real* __restrict__ f, int sx, int sy, int sz,
                         real hx2, real hy2, real hz2, real qx, real qy, real qz, real* __restrict__ coarse, int cx, int cy, int cz,
                         int pzchunk, int gx, int gy, int xcd_mode, int pzbeg, int pzend, int fzoff, int czoff, int abl = 0) {
    const int A = DBG ? abl : 0;
    constexpr int NPW = TYW - 2;  // producing waves = coarse rows of a tile
    __shared__ real eR[3][TYW][2][64];  // [plane % 3][wave][first / last row][lane]: red half
    __shared__ real eK[2][TYW][2][64];  // [plane & 1][wave][first / last row][lane]: new black half
    __shared__ real pS[TYW][3][64];     // [wave][a / b / c][lane]: the (x, z) sub-sums of the wave's first row
    const Geo<XSplit, real> gf(sx, sy), gc(cx, cy);
    const int lane = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.y);
    int bx, by, bz;
    tile_of_block(xcd_mode, gx, gy, bx, by, bz);
    // the GLOBAL coarse planes [pzbeg, pzend) in runs of pzchunk; v / f start at global fine plane fzoff, coarse at global
    // coarse plane czoff (a z-slab; the whole grid: 1, cz - 1, 0, 0).  sz, cz and every plane index below are global.
    const int pz0 = pzbeg + bz * pzchunk, pz1 = min(pz0 + pzchunk, pzend);
    if (pz0 >= pz1) return;  // uniform over the workgroup
    vin -= (ptrdiff_t)fzoff * (ptrdiff_t)gf.PL;  // index by global plane from here on (never dereferenced outside the slab)
    vout -= (ptrdiff_t)fzoff * (ptrdiff_t)gf.PL;
    f -= (ptrdiff_t)fzoff * (ptrdiff_t)gf.PL;
    const int g0 = 2 * pz0 - 1, glast = 2 * pz1 - 1;
    const int rmax = min(glast + 2, sz - 1), fmax = min(glast + 1, sz - 2), kmax = min(glast + 1, sz - 1);
    const int pstore1 = pz1 < pzend ? glast - 1 : glast;  // black planes [g0, pstore1] are stored by this run (glast by the next, if any)

    const bool top = w == 0, bot = w == TYW - 1, prod = !top && !bot;
    const int cyb = by * NPW + w;  // coarse row of the wave (the halo waves: the neighbouring tiles' rows)
    const bool rlx0 = !top, res0 = !top, res1 = prod;
    const bool strow = prod || (bot && by == gy - 1);
    int roff[2];
    bool yint[2];
#pragma unroll
    for (int o = 0; o < 2; o++) {
        const int y = 2 * cyb - 1 + o;
        roff[o] = min(max(y, 0), sy - 1) * gf.P;
        yint[o] = y >= 1 && y <= sy - 2;
    }
    const int roffD = min(2 * cyb + 1, sy - 1) * gf.P;  // the row below the wave's two (bottom halo wave only)

    const int in = bx * 61 + lane - 1;  // nominal coarse column
    const int i = min(max(in, 0), cx - 1);
    const bool hasB = i <= cx - 2;
    const unsigned off[2] = {(unsigned)i * (unsigned)sizeof(real), (unsigned)(gf.H + (hasB ? i : 0)) * (unsigned)sizeof(real)};  // bytes
    const bool xbl = i == 0 || i == cx - 1;  // the even entry of the pair lies on an x-face
    const bool relaxOK[2] = {in >= 1 && in <= cx - 2, in >= 0 && in <= cx - 2};
    const bool inner = lane >= 2 && lane <= 62;
    const bool storeOK[2] = {relaxOK[0] && inner, relaxOK[1] && (inner || (lane == 1 && bx == 0))};
    const bool resOK[2] = {relaxOK[0] && lane >= 1, relaxOK[1] && lane <= 62};
    const bool produces = prod && inner && in >= 1 && in <= cx - 2 && cyb <= cy - 2;
    const int wU = w > 0 ? w - 1 : 0, wD = w < TYW - 1 ? w + 1 : TYW - 1;
    const int PL = (int)gf.PL;
    const double rd = relax3d_rd<real>(hx2, hy2, hz2);

    // planes relative to the iteration g: P = g - 1, C = g, N = g + 1 (its black half is computed in the iteration),
    // R2 = red half of g + 2, X = red half of g + 3 (in flight); f: C = g, N = g + 1, X = g + 2 (in flight).
    // Registers are what limits this kernel (sixteen waves: 128 each), so two kinds of values travel in f's slots: where an
    // entry of the black half is one no pass writes (a face of the grid), its slot holds v itself instead of f (f is not
    // needed there: no relaxation, residual 0); and in the bottom halo wave, whose second row needs f only at the black
    // points, the other slot of that row holds the red value of the row below.
    real vP[2][2], vC[2][2], vN[2][2], vR2[2], vX[2], fC[2][2], fN[2][2], fX[2][2];
    real rm[2][2], r0[2][2];  // residual planes g - 2, g - 1: [row][A / B]
    real k1 = 0, k2 = 0, k3 = 0, k4 = 0, k5 = 0;  // the coarse row without the wave below's share
#pragma unroll
    for (int o = 0; o < 2; o++) {
#pragma unroll
        for (int h = 0; h < 2; h++) vP[o][h] = vC[o][h] = vN[o][h] = fC[o][h] = fN[o][h] = fX[o][h] = rm[o][h] = r0[o][h] = 0;
        vR2[o] = vX[o] = 0;
    }

    // ---- set-up for the first iteration gs = g0 - 2 (odd; its black half of row o is o)
    const int gs = g0 - 2;
    {
        const auto q0 = plane_rsrc<real>(vin + (ptrdiff_t)max(gs, 0) * (ptrdiff_t)gf.PL, PL, 1);
        const auto q1 = plane_rsrc<real>(vin + (ptrdiff_t)(gs + 1) * (ptrdiff_t)gf.PL, PL, 2);
        const auto qf = plane_rsrc<real>(f + (ptrdiff_t)(gs + 1) * (ptrdiff_t)gf.PL, PL, 1);
#pragma unroll
        for (int o = 0; o < 2; o++) {
            vC[o][o] = buf_load<real>(q0, off[o], roff[o]);
            vN[o][0] = buf_load<real>(q1, off[0], roff[o]);
            vN[o][1] = buf_load<real>(q1, off[1], roff[o]);
            vR2[o] = buf_load<real>(q1, off[o], roff[o] + PL);
            fN[o][o] = buf_load<real>(qf, off[o], roff[o]);
        }
        if (bot) fN[1][0] = buf_load<real>(q1, off[1], roffD);
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
        if (!yint[0] || gs + 1 == 0 || xbl) fN[0][0] = vN[0][0];
        if (!yint[1] || gs + 1 == 0) fN[1][1] = vN[1][1];
        const int s = (gs + 1) % 3;
        eR[s][w][0][lane] = vN[0][1];
        eR[s][w][1][lane] = vN[1][0];
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    const real* pv = vin + (ptrdiff_t)gs * (ptrdiff_t)gf.PL;  // plane g (g = -1 is never dereferenced)
    const real* pf = f + (ptrdiff_t)gs * (ptrdiff_t)gf.PL;
    real* po = vout + (ptrdiff_t)gs * (ptrdiff_t)gf.PL;
    int s1 = (gs + 1) % 3;  // slot of plane g + 1 in eR

    // the coarse row of plane pz: own share (k1 .. k5, formed one iteration earlier) + the sub-sums of the wave below's first row
    auto complete = [&](int pz) __attribute__((always_inline)) {
        const real ea = pS[wD][0][lane], eb = pS[wD][1][lane], ec = pS[wD][2][lane];
        if (produces)
            coarse[gc.row(cyb, pz - czoff) + gc.pos(i)] =
                k1 + (1 / 16.0f) * (k2 + (k3 + ea)) + (1 / 32.0f) * (k4 + eb) + (1 / 64.0f) * (k5 + ec);
    };

    auto iteration = [&](auto parity, int g) __attribute__((always_inline)) {
        constexpr int PAR = decltype(parity)::value;  // g & 1
        // ---- requests: red half of plane g + 3, f of plane g + 2, the unwritten black entries of plane g + 2
        const bool moreR = g + 3 <= rmax && !(A & 1), moreF = g + 2 <= fmax && !(A & 1), moreK = g + 2 <= kmax && !(A & 1);
        const auto rv = plane_rsrc<real>(pv, PL, 4), rf = plane_rsrc<real>(pf, PL, 3), ro = plane_rsrc<real>(po, PL, 1);
        // ---- the black values of plane g (relaxed in the previous iteration) go out first: like the loads they have the whole
        // iteration before the wait at its end (stored right where they are computed, that wait stood for their round trip)
        __builtin_amdgcn_s_setprio(3);  // requests leave the CU before the other waves' arithmetic (as in relax3d_xs_pipe_kernel)
        if (g >= max(g0, 1) && g <= pstore1 && strow && !(A & 2)) {
#pragma unroll
            for (int o = 0; o < 2; o++) {
                const int hk = (o + PAR) & 1;  // black half of row o at plane g
                if (yint[o] && storeOK[hk]) buf_store_nt<real>(vC[o][hk], ro, off[hk], roff[o]);
            }
        }
#pragma unroll
        for (int o = 0; o < 2; o++) {
            const int hr = (o + PAR) & 1;  // red half of plane g + 3 = black half of plane g + 2
            if (moreR) vX[o] = buf_load<real>(rv, off[hr], roff[o] + 3 * PL);
            const bool full = o == 0 ? res0 : res1;
            if (moreF && full) fX[o][1 - hr] = buf_load<real>(rf, off[1 - hr], roff[o] + 2 * PL);
            // an entry no pass writes: v itself.  (The x-face entry is a matter of ONE lane, and the choice of the descriptor by lane makes
            // hipcc loop over the descriptors (~30 instructions per iteration).  Round 4 tried the lane's load under its own exec mask
            // behind / beside the load every lane makes: the compiler then loads into a temporary and copies it into place behind an
            // s_waitcnt vmcnt(0) in mid-iteration -- 614 against 570 us.  Both loads as inline assembly that updates the register in
            // place removes loop, copy and wait (221 instead of 248 vector instructions per iteration) and is WRONG: the compiler moves
            // a register whose pending load it does not know about (the 9^3 case of tests/test_gpu_relax_rr.py fails).  The loop stays.)
            if (moreK && (!yint[o] || g + 2 == sz - 1 || (hr == 0 && xbl)))
                fX[o][hr] = buf_load<real>(rv, off[hr], roff[o] + 2 * PL);
            else if (moreF && (full || o == 1 || rlx0))
                fX[o][hr] = buf_load<real>(rf, off[hr], roff[o] + 2 * PL);
            if (o == 1 && bot && moreK) fX[1][1 - hr] = buf_load<real>(rv, off[hr], roffD + 2 * PL);  // the row below, red there
        }
        __builtin_amdgcn_s_setprio(0);
        // ---- the coarse plane whose own share was formed in the previous iteration
        if (PAR == 0 && g >= g0 + 3 && !(A & 32)) complete((g - 2) >> 1);
        // ---- black points of plane g + 1
        const bool dorelax = g + 1 >= 1 && g + 1 <= sz - 2;
#pragma unroll
        for (int o = 0; o < 2; o++) {
            const int h = (o + 1 + PAR) & 1;  // black half of row o at plane g + 1
            if (o == 0 && !rlx0) continue;
            real N, S, O, E;
            if (o == 0) {
                N = eR[s1][wU][1][lane];
                S = vN[1][h];
            } else {
                N = vN[0][h];
                const real t = eR[s1][wD][0][lane];
                S = bot ? fN[1][1 - h] : t;
            }
            if (h == 0) {
                O = wave_from_prev_lane<real>(vN[o][1]);
                E = vN[o][1];
            } else {
                O = vN[o][0];
                E = wave_from_next_lane<real>(vN[o][0]);
            }
            const real nv = (A & 4) ? N : relax3d_point_rd<real>(O, E, N, S, vC[o][h], vR2[o], fN[o][h], hx2, hy2, hz2, rd);
            const bool ok = dorelax && yint[o] && relaxOK[h];
            vN[o][h] = ok ? nv : fN[o][h];
        }
        // ---- publish: new black half of plane g + 1, red half of plane g + 2 (first and last row)
        const int s2 = s1 == 2 ? 0 : s1 + 1, s0 = s1 == 0 ? 2 : s1 - 1;
        eK[(PAR + 1) & 1][w][0][lane] = vN[0][(1 + PAR) & 1];
        eK[(PAR + 1) & 1][w][1][lane] = vN[1][PAR & 1];
        eR[s2][w][0][lane] = vR2[0];
        eR[s2][w][1][lane] = vR2[1];
        // ---- residual of plane g
        real rn[2][2] = {{0, 0}, {0, 0}};
        if (g >= g0 && !(A & 8)) {
            const real upK = eK[PAR][wU][1][lane], upR = eR[s0][wU][1][lane];  // the last row of the wave above
            const real dnK = eK[PAR][wD][0][lane], dnR = eR[s0][wD][0][lane];  // the first row of the wave below
            const real up[2] = {PAR == 1 ? upK : upR, PAR == 1 ? upR : upK};
            const real dn[2] = {PAR == 0 ? dnK : dnR, PAR == 0 ? dnR : dnK};
#pragma unroll
            for (int o = 0; o < 2; o++) {
                if (o == 0 ? !res0 : !res1) continue;
                const real Bl = wave_from_prev_lane<real>(vC[o][1]);  // v(2i-1): odd entry of lane i-1
                const real Ar = wave_from_next_lane<real>(vC[o][0]);  // v(2i+2): even entry of lane i+1
                const real An = o == 0 ? up[0] : vC[0][0], As = o == 1 ? dn[0] : vC[1][0];
                const real Bn = o == 0 ? up[1] : vC[0][1], Bs = o == 1 ? dn[1] : vC[1][1];
                const real a = residual3d_point<real, MODE>(Bl, vC[o][1], An, As, vP[o][0], vN[o][0], vC[o][0], fC[o][0], qx, qy, qz);
                const real b = residual3d_point<real, MODE>(vC[o][0], Ar, Bn, Bs, vP[o][1], vN[o][1], vC[o][1], fC[o][1], qx, qy, qz);
                rn[o][0] = (yint[o] && resOK[0]) ? a : (real)0;
                rn[o][1] = (yint[o] && resOK[1]) ? b : (real)0;
            }
        }
        // ---- g = 2 pz + 1: the sub-sums of coarse plane pz per fine row; the first row's go to the wave above
        if (PAR == 1 && g >= g0 + 2 && !(A & 32)) {
            real sa[2], sb[2], sc[2];
#pragma unroll
            for (int o = 0; o < 2; o++) {
                const real lm = wave_from_prev_lane<real>(rm[o][1]), l0 = wave_from_prev_lane<real>(r0[o][1]),
                           lp = wave_from_prev_lane<real>(rn[o][1]);
                sa[o] = r0[o][0];
                sb[o] = ((rn[o][0] + r0[o][1]) + rm[o][0]) + l0;  // (N + E + S + O): (x,z+1), (x+1,z), (x,z-1), (x-1,z)
                sc[o] = ((rn[o][1] + rm[o][1]) + lm) + lp;        // (NE + SE + SO + NO)
            }
            pS[w][0][lane] = sa[0];
            pS[w][1][lane] = sb[0];
            pS[w][2][lane] = sc[0];
            k1 = (1 / 8.0f) * sa[1];
            k2 = sb[1];
            k3 = sa[0];
            k4 = sc[1] + sb[0];
            k5 = sc[0];
        }
#pragma unroll
        for (int o = 0; o < 2; o++) {
            rm[o][0] = r0[o][0]; rm[o][1] = r0[o][1];
            r0[o][0] = rn[o][0]; r0[o][1] = rn[o][1];
        }
        if (A & 16) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): this iteration's requests have had the whole iteration
#pragma unroll
        for (int o = 0; o < 2; o++) {
            const int h = (o + 1 + PAR) & 1;
            vP[o][0] = vC[o][0]; vP[o][1] = vC[o][1];
            vC[o][0] = vN[o][0]; vC[o][1] = vN[o][1];
            vN[o][h] = vR2[o];  // the red half of plane g + 2; its black half is the next iteration's work
            vR2[o] = vX[o];
            fC[o][0] = fN[o][0]; fC[o][1] = fN[o][1];
            fN[o][0] = fX[o][0]; fN[o][1] = fX[o][1];
        }
        pv += PL;
        pf += PL;
        po += PL;
        s1 = s2;
    };

    for (int g = gs;; g += 2) {
        iteration(std::integral_constant<int, 1>{}, g);
        if (g >= glast) break;
        iteration(std::integral_constant<int, 0>{}, g + 1);
    }
    if (!(A & 32)) complete(pz1 - 1);
}

// One launch: black pass + residual + restrict over the whole level.  hx2 .. : squared spacings; rcp: the residual multiplies by
// their exact reciprocals.  Returns false when the level is not taken (too small for the tile shape).
// Automatic choice (measured, tools/rr_black_time.py): fp64 from 513-point rows on (513^3: 0.55 against 0.71 ms for the black
// pass + residual + restrict, 1025^3: 11.5 against 13.6 ms with the two red+black sweeps before; 257^3: equal, left alone);
// fp32 moves half the bytes with the same instructions and is no faster than its separate launches (513^3: 0.91 / 0.87 ms).
bool relax_rr3d_xs_takes(const mgx_ctx* ctx, const int n[3], const int cn[3], size_t elem) {
    if (!ctx->rr_black || cn[0] < 3 || cn[1] < 3 || cn[2] < 3) return false;
    // the kernel addresses up to four fine planes through one buffer descriptor (a 32-bit range)
    if ((unsigned long long)(elem == 8 ? Geo<XSplit, double>(n[0], n[1]).PL : Geo<XSplit, float>(n[0], n[1]).PL) * elem * 4ull >= (1ull << 32)) return false;
    return ctx->rr_black == 2 || (elem == 8 && n[0] >= 385 && n[1] >= 129 && n[2] >= 65);
}

template <class real>
bool relax_rr3d_xs_launch(mgx_ctx* ctx, real* v, const real* f, const int n[3], real hx2, real hy2, real hz2, int mode, bool rcp,
                          real* coarse_f, const int cn[3], int fzoff, int czoff, int pzbeg, int pzend) {
    // n, cn: GLOBAL sizes; v / f start at global fine plane fzoff, coarse_f at global coarse plane czoff; the launch relaxes
    // the black points of the fine planes [2 pzbeg - 1, 2 pzend - 1] and forms the coarse planes [pzbeg, pzend)
    if (!relax_rr3d_xs_takes(ctx, n, cn, sizeof(real)) || pzbeg < 1 || pzend > cn[2] - 1 || pzend <= pzbeg) return false;
    const int T = ctx->rr_black_waves == 12 ? 12 : ctx->rr_black_waves == 8 ? 8 : 16;
    const int wg_per_cu = T == 8 ? 2 : 1;
    const int gx = ceil_div(cn[0] - 2, 61), gy = ceil_div(cn[1] - 2, T - 2);
    const int tiles = gx * gy, planes = pzend - pzbeg;
    int pzc = ctx->rr_pzchunk;
    if (pzc <= 0) {
        // all workgroups take the same time and one fits a CU: the fewest runs that fill whole rounds to 90 %, runs of at least
        // 8 coarse planes (the two planes a run relaxes before its first residual)
        int nchunks = 1;
        double best = 0;
        for (int c = 1; c <= 16 && planes / c >= 8; c++) {
            const long long wgs = (long long)tiles * c, cap = (long long)ctx->num_cus * wg_per_cu;
            const double eff = (double)wgs / (double)(((wgs + cap - 1) / cap) * cap) * (double)planes / (double)(planes + 2 * c);
            if (eff > best + 1e-9) { best = eff; nchunks = c; }
            if (eff >= 0.9) break;
        }
        pzc = ceil_div(planes, nchunks);
    }
    dim3 g(tiles * ceil_div(planes, pzc), 1, 1);
    real qx = hx2, qy = hy2, qz = hz2;
    if (rcp) {
        qx = (real)1 / hx2;
        qy = (real)1 / hy2;
        qz = (real)1 / hz2;
    }
#define MGX_BRR_D(M, W, D)                                                                                                     \
    MGX_LAUNCH((relax_rr3d_xs_kernel<real, M, W, D>), g, dim3(64, W, 1), 0, ctx->compute, (const real*)v, v, f, n[0], n[1], \
                       n[2], hx2, hy2, hz2, qx, qy, qz, coarse_f, cn[0], cn[1], cn[2], pzc, gx, gy, ctx->rr_xcd >= 1, pzbeg, pzend, fzoff,   \
                       czoff, ctx->rr_black_abl)
#ifdef MGX_DIAGNOSTICS
#define MGX_BRR(M, W) do { if (ctx->rr_black_abl) MGX_BRR_D(M, W, 1); else MGX_BRR_D(M, W, 0); } while (0)
#else
#define MGX_BRR(M, W) MGX_BRR_D(M, W, 0)
#endif
#define MGX_BRR_W(M)                              \
    do {                                          \
        if (T == 16) MGX_BRR(M, 16); else if (T == 12) MGX_BRR(M, 12); else MGX_BRR(M, 8); \
    } while (0)
    if (mode == MGX_RESIDUAL_REF_COMPAT) {
        if (rcp) MGX_BRR_W(2); else MGX_BRR_W(0);
    } else {
        if (rcp) MGX_BRR_W(3); else MGX_BRR_W(1);
    }
#undef MGX_BRR_W
#undef MGX_BRR
#undef MGX_BRR_D
    snprintf(ctx->last_rr_kernel, sizeof ctx->last_rr_kernel, "relax_rr3d_xs_kernel<%s,%d,%d>", sizeof(real) == 8 ? "double" : "float",
             (mode == MGX_RESIDUAL_REF_COMPAT ? 0 : 1) + (rcp ? 2 : 0), T);
    return true;
}
template bool relax_rr3d_xs_launch<float>(mgx_ctx*, float*, const float*, const int[3], float, float, float, int, bool, float*, const int[3],
                                          int, int, int, int);
template bool relax_rr3d_xs_launch<double>(mgx_ctx*, double*, const double*, const int[3], double, double, double, int, bool, double*,
                                           const int[3], int, int, int, int);

}  // namespace mgx

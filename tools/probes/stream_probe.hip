// What does the memory pipe of a CU sustain for the one-launch sweep's traffic (per step and workgroup: old black + two
// planes of f in, red + black out; 513^3 doubles, x-split geometry), as a function of how the SAME bytes are cut into
// wave-instructions?  No arithmetic, no LDS: every wave issues its loads and stores of a step, then all waves meet at a
// barrier and wait for their loads (as the kernel does).
//   NW waves per workgroup (one workgroup per CU), ROWS rows per wave, VEC doubles per lane and instruction
//   hipcc --offload-arch=gfx950 -O3 tools/probes/stream_probe.hip -o tools/probes/stream_probe
#include <hip/hip_runtime.h>

#include <cstdio>

template <int NW, int ROWS, int VEC, bool BARRIER>
__global__ void __launch_bounds__(64 * NW) stream(const double* __restrict__ vin, double* __restrict__ vout, const double* __restrict__ f,
                                                 int planes_per_wg, int D, double* sink) {
    typedef double vec __attribute__((ext_vector_type(VEC)));
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const size_t P = 528, PL = P * 513;
    const int tile = blockIdx.x % 64, run = blockIdx.x / 64;         // 64 tiles of 8 rows x 4 runs of 128 planes
    constexpr int WPR = 256 / (64 * VEC);                              // waves side by side in a 256-pair half-row
    static_assert(NW % WPR == 0 && (NW / WPR) * ROWS == 8, "a workgroup covers 8 rows of 256 pairs");
    const int wx = w % WPR, wy = w / WPR;
    const size_t col = (size_t)(wx * 64 + lane) * VEC;
    const int y0 = 1 + tile * 8 + wy * ROWS;
    const int z0 = 1 + run * planes_per_wg;
    vec acc = 0;
    for (int s = 0; s < planes_per_wg; s++) {
        const int zr = min(z0 + s + D, 511), zb = z0 + s;
        vec a[ROWS], b[ROWS], c[ROWS];
#pragma unroll
        for (int r = 0; r < ROWS; r++) {
            const size_t row = (size_t)min(y0 + r, 512) * P;
            const int q = (y0 + r + s) & 1;
            a[r] = *(const vec*)&vin[(size_t)zr * PL + row + q * 264 + col];                       // old black of the red stage's plane
            b[r] = __builtin_nontemporal_load((const vec*)&f[(size_t)zr * PL + row + (q ^ 1) * 264 + col]);  // f red
            c[r] = __builtin_nontemporal_load((const vec*)&f[(size_t)zb * PL + row + q * 264 + col]);        // f black
            __builtin_nontemporal_store(acc + (double)r, (vec*)&vout[(size_t)zr * PL + row + (q ^ 1) * 264 + col]);  // red
            __builtin_nontemporal_store(acc - (double)r, (vec*)&vout[(size_t)zb * PL + row + q * 264 + col]);        // black
        }
        if (BARRIER) asm volatile("s_barrier" ::: "memory");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int r = 0; r < ROWS; r++) acc += a[r] + b[r] + c[r];
    }
    if (acc[0] == 1.2345) *sink = acc[0];
}

template <int NW, int ROWS, int VEC, bool BARRIER>
static void run(const char* name, double* vin, double* vout, double* f, double* sink) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    float best = 1e9f;
    for (int it = 0; it < 6; it++) {
        (void)hipEventRecord(e0);
        stream<NW, ROWS, VEC, BARRIER><<<256, 64 * NW>>>(vin, vout, f, 128, 6, sink);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (it > 0 && ms < best) best = ms;
    }
    const double bytes = 256.0 * 128 * 8 * 256 * 8 * 5;  // 5 half-rows of 256 doubles per row, plane and workgroup
    printf("%-44s %2d waves x %d rows x %2d B/lane, barrier %d: %.4f ms, %.0f GB/s\n", name, NW, ROWS, VEC * 8, (int)BARRIER, best, bytes / best / 1e6);
}

int main() {
    const size_t elems = (size_t)528 * 513 * 513;
    double *vin, *vout, *f, *sink;
    if (hipMalloc(&vin, elems * 8) != hipSuccess || hipMalloc(&vout, elems * 8) != hipSuccess || hipMalloc(&f, elems * 8) != hipSuccess) return 1;
    (void)hipMalloc(&sink, 8);
    (void)hipMemset(vin, 0, elems * 8);
    (void)hipMemset(vout, 0, elems * 8);
    (void)hipMemset(f, 0, elems * 8);
    run<8, 4, 1, true>("the sweep kernel's cut", vin, vout, f, sink);
    run<8, 4, 1, false>("  without the barrier", vin, vout, f, sink);
    run<16, 2, 1, true>("16 waves of 2 rows", vin, vout, f, sink);
    run<16, 2, 1, false>("  without the barrier", vin, vout, f, sink);
    run<4, 4, 2, true>("4 waves, 16 B per lane", vin, vout, f, sink);
    run<8, 2, 2, true>("8 waves of 2 rows, 16 B per lane", vin, vout, f, sink);
    run<8, 2, 2, false>("  without the barrier", vin, vout, f, sink);
    run<16, 1, 2, true>("16 waves of 1 row, 16 B per lane", vin, vout, f, sink);
    run<16, 1, 2, false>("  without the barrier", vin, vout, f, sink);
    return 0;
}

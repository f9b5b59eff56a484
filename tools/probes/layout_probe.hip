// layout_probe.hip -- does the x-split layout's access pattern (a colour pass touches one 2112-byte half of every 4224-byte row of
// three arrays) cost bandwidth against three linear streams of the same bytes?  A triad c = a + k b over 513^3-sized arrays:
//   linear : every byte of the first half of each array (what a colour-contiguous "plane-split" layout would stream)
//   halfrow: 264 doubles used, 264 skipped, alternating which half by row parity (the x-split layout's colour pass)
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/bin/layout_probe tools/probes/layout_probe.hip
#include <hip/hip_runtime.h>

#include <cstdio>

typedef double d2 __attribute__((ext_vector_type(2)));

// one block of 132 threads x 16 B = one half-row (264 doubles); grid = rows
template <int MODE>
__global__ void __launch_bounds__(256) pass(const double* __restrict__ a, const double* __restrict__ b, double* __restrict__ c, int rows_per_plane, size_t nrows) {
    const size_t row = (size_t)blockIdx.x * 2 + (threadIdx.x >= 128 ? 1 : 0);
    const int t = threadIdx.x & 127;
    if (row >= nrows) return;
    size_t base;
    if (MODE == 0) base = row * 264;  // linear: half-rows back to back
    else {
        const size_t z = row / rows_per_plane, y = row % rows_per_plane;
        base = row * 528 + ((y + z) & 1) * 264;  // the half by row parity
    }
    for (int i = t; i < 132; i += 128) {
        const d2 x = *(const d2*)&a[base + 2 * i], y = __builtin_nontemporal_load((const d2*)&b[base + 2 * i]);
        d2 r = {x.x + 1.5 * y.x, x.y + 1.5 * y.y};
        __builtin_nontemporal_store(r, (d2*)&c[base + 2 * i]);
    }
}

int main() {
    const int sy = 513, sz = 513;
    const size_t nrows = (size_t)sy * sz, elems = nrows * 528;
    double *a, *b, *c;
    hipMalloc(&a, elems * 8); hipMalloc(&b, elems * 8); hipMalloc(&c, elems * 8);
    hipMemset(a, 0, elems * 8); hipMemset(b, 0, elems * 8); hipMemset(c, 0, elems * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = (int)((nrows + 1) / 2);
    const double bytes = 3.0 * nrows * 264 * 8;
    for (int rep = 0; rep < 3; rep++)
        for (int mode = 0; mode < 2; mode++) {
            for (int w = 0; w < 3; w++) { if (mode) pass<1><<<grid, 256>>>(a, b, c, sy, nrows); else pass<0><<<grid, 256>>>(a, b, c, sy, nrows); }
            hipEventRecord(e0);
            for (int r = 0; r < 20; r++) { if (mode) pass<1><<<grid, 256>>>(a, b, c, sy, nrows); else pass<0><<<grid, 256>>>(a, b, c, sy, nrows); }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("%s: %.1f us per pass, %.2f TB/s\n", mode ? "half-rows (x-split colour pass)" : "linear (colour-contiguous)     ", ms * 1e3 / 20, bytes * 20 / (ms * 1e-3) / 1e12);
        }
    return 0;
}

// mall_probe.hip -- streaming rates as a function of the working set: does data that fits the 256 MiB Infinity Cache (MALL) stream
// faster than HBM on this box, for reads, for read-after-write in reverse order, and with non-temporal stores?
//   hipcc --offload-arch=gfx950 -O3 -o mall_probe tools/probes/mall_probe.hip && ./mall_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef double d2 __attribute__((ext_vector_type(2)));

__global__ void triad(const d2* __restrict__ a, const d2* __restrict__ b, d2* __restrict__ c, size_t n, int nt) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        d2 x = a[i], y = __builtin_nontemporal_load(&b[i]);
        d2 r = {x.x + 1.5 * y.x, x.y + 1.5 * y.y};
        if (nt) __builtin_nontemporal_store(r, &c[i]);
        else c[i] = r;
    }
}
__global__ void rd(const d2* __restrict__ a, double* out, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        d2 x = a[i];
        if (x.x == 123.456) out[0] = x.y;
    }
}

int main() {
    const size_t maxb = (size_t)1 << 30;  // 1 GiB per array
    d2 *a, *b, *c;
    double* out;
    hipMalloc(&a, maxb); hipMalloc(&b, maxb); hipMalloc(&c, maxb); hipMalloc(&out, 64);
    hipMemset(a, 0, maxb); hipMemset(b, 0, maxb); hipMemset(c, 0, maxb);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int B = 256;
    for (size_t mb : {16, 32, 48, 64, 96, 128, 192, 256, 512, 1024}) {
        const size_t n = mb * ((size_t)1 << 20) / 16;  // elements (d2) per array
        const int grid = (int)((n + B - 1) / B);
        const int reps = (int)(8192 / mb) + 4;
        for (int nt = 0; nt < 2; nt++) {
            // (1) triad a, b -> c repeated on the same arrays: working set 3 x mb
            for (int w = 0; w < 3; w++) triad<<<grid, B>>>(a, b, c, n, nt);
            hipEventRecord(e0);
            for (int r = 0; r < reps; r++) triad<<<grid, B>>>(a, b, c, n, nt);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double t1 = 3.0 * mb * 1.048576e6 * reps / (ms * 1e-3) / 1e12;
            // (2) ping-pong: a -> c then c -> a (what one pass writes the next one reads): working set 3 x mb, read-after-write
            hipEventRecord(e0);
            for (int r = 0; r < reps; r++) { triad<<<grid, B>>>(a, b, c, n, nt); triad<<<grid, B>>>(c, b, a, n, nt); }
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
            const double t2 = 3.0 * mb * 1.048576e6 * 2 * reps / (ms * 1e-3) / 1e12;
            printf("array %5zu MiB (triad working set %5zu MiB) %s stores: repeated %6.2f TB/s   ping-pong %6.2f TB/s\n", mb, 3 * mb, nt ? "nt    " : "normal", t1, t2);
        }
        for (int w = 0; w < 3; w++) rd<<<grid, B>>>(a, out, n);
        hipEventRecord(e0);
        for (int r = 0; r < reps; r++) rd<<<grid, B>>>(a, out, n);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("array %5zu MiB read only, repeated: %6.2f TB/s\n", mb, mb * 1.048576e6 * reps / (ms * 1e-3) / 1e12);
        fflush(stdout);
    }
    return 0;
}

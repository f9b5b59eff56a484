// How long does ONE wave take to ISSUE a batch of independent 512-byte vector loads (no waiting for the data), as a function
// of where the addresses lie and of how many waves of the CU do the same at the same time?  Background: the one-launch sweep
// kernel spends ~200 cycles per vector-memory instruction in its issue phase, per wave.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/vmem_issue_probe.hip -o /tmp/vmem_issue_probe && /tmp/vmem_issue_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

constexpr int K = 28;  // loads per batch

// MODE 0: K consecutive 512-B pieces (one 14 KB block per wave and step)
// MODE 1: 4 rows (4224 B apart) x 7 planes (16 planes = 34 MB apart), the sweep kernel's pattern
// MODE 2: K pieces 64 KB apart
// MODE 3: K pieces 2 MB + 4 KB apart
template <int MODE>
__global__ void __launch_bounds__(512) probe(const double* __restrict__ a, size_t mask, int steps, int nwaves_active, long long* out,
                                             double* sink) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (w >= nwaves_active) return;
    const size_t row = 528, plane = (size_t)528 * 513;
    const size_t base = ((size_t)blockIdx.x * 8 + w) * 4 * row + lane;  // each wave its own rows
    long long t_issue = 0, t_all = 0;
    double acc = 0;
    for (int s = 0; s < steps; s++) {
        const double* p[K];
#pragma unroll
        for (int k = 0; k < K; k++) {
            size_t off;
            if (MODE == 0) off = base + (size_t)k * 64;
            else if (MODE == 1) off = base + (size_t)(k & 3) * row + (size_t)(k >> 2) * 16 * plane;
            else if (MODE == 2) off = base + (size_t)k * 8192;
            else off = base + (size_t)k * (262144 + 512);
            p[k] = a + ((off + (size_t)s * plane) & mask);
            asm volatile("" : "+v"(p[k]));  // addresses ready before the clock is read
        }
        double v[K];
        const long long t0 = __builtin_readcyclecounter();
#pragma unroll
        for (int k = 0; k < K; k++) v[k] = __builtin_nontemporal_load(p[k]);
        asm volatile("" ::: "memory");
        const long long t1 = __builtin_readcyclecounter();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const long long t2 = __builtin_readcyclecounter();
#pragma unroll
        for (int k = 0; k < K; k++) acc += v[k];
        t_issue += t1 - t0;
        t_all += t2 - t0;
    }
    if (lane == 0) {
        out[(blockIdx.x * 8 + w) * 2] = t_issue;
        out[(blockIdx.x * 8 + w) * 2 + 1] = t_all;
    }
    if (acc == 1.2345) *sink = acc;
}

int main() {
    const size_t elems = (size_t)1 << 28;  // 2 GiB of doubles (a power of two: offsets wrap with a mask)
    double *a, *sink;
    long long* out;
    if (hipMalloc(&a, elems * 8) != hipSuccess) return 1;
    (void)hipMemset(a, 0, elems * 8);
    (void)hipMalloc(&sink, 8);
    (void)hipMalloc(&out, 256 * 8 * 2 * 8);
    const int steps = 128;
    const char* names[] = {"consecutive 512 B", "4 rows x 7 planes (sweep pattern)", "64 KB apart", "2 MB + 4 KB apart"};
    for (int nw = 1; nw <= 8; nw *= 2)
        for (int mode = 0; mode < 4; mode++) {
            (void)hipMemset(out, 0, 256 * 8 * 2 * 8);
            if (mode == 0) probe<0><<<256, 512>>>(a, elems - 1, steps, nw, out, sink);
            if (mode == 1) probe<1><<<256, 512>>>(a, elems - 1, steps, nw, out, sink);
            if (mode == 2) probe<2><<<256, 512>>>(a, elems - 1, steps, nw, out, sink);
            if (mode == 3) probe<3><<<256, 512>>>(a, elems - 1, steps, nw, out, sink);
            (void)hipDeviceSynchronize();
            std::vector<long long> h(256 * 8 * 2);
            (void)hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
            double ti = 0, ta = 0;
            for (int b = 0; b < 256; b++)
                for (int w = 0; w < nw; w++) {
                    ti += h[(b * 8 + w) * 2];
                    ta += h[(b * 8 + w) * 2 + 1];
                }
            const double n = 256.0 * nw * steps * K;
            printf("%d wave(s)/CU, %-36s issue %7.1f cycles per load, issue + arrival %7.1f cycles per load (%.0f GB/s chip-wide)\n", nw,
                   names[mode], ti / n, ta / n, 512.0 * 256 * nw / (ta / n / 2.4e9) / 1e9);
        }
    return 0;
}

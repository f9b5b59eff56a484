// cross-stream dependency latency: kernel on s1 -> event -> s2 waits -> kernel on s2 -> event -> s1 waits ...
// build: hipcc --offload-arch=gfx950 -O2 -o tools/probes/hop_latency tools/probes/hop_latency.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void tiny(int* p) { if (threadIdx.x == 0) p[0] += 1; }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    int* d; hipMalloc(&d, 64); hipMemset(d, 0, 64);
    hipStream_t s1, s2; hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    const int N = 4000;
    std::vector<hipEvent_t> e(2 * N);
    for (auto& x : e) hipEventCreateWithFlags(&x, hipEventDisableTiming);
    for (int rep = 0; rep < 3; rep++) {
        hipDeviceSynchronize();
        double t0 = now();
        for (int i = 0; i < 2 * N; i++) hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s1, d);
        double th = now() - t0;
        hipDeviceSynchronize();
        double a = now() - t0;
        t0 = now();
        for (int i = 0; i < N; i++) {
            hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s1, d);
            hipEventRecord(e[2 * i], s1); hipStreamWaitEvent(s2, e[2 * i], 0);
            hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s2, d);
            hipEventRecord(e[2 * i + 1], s2); hipStreamWaitEvent(s1, e[2 * i + 1], 0);
        }
        double th2 = now() - t0;
        hipDeviceSynchronize();
        double b = now() - t0;
        printf("same stream %.2f us/kernel (host enqueue %.2f);  ping-pong %.2f us/kernel (host enqueue %.2f)  -> hop ~ %.2f us\n",
               a / (2 * N) * 1e6, th / (2 * N) * 1e6, b / (2 * N) * 1e6, th2 / (2 * N) * 1e6, (b - a) / (2 * N) * 1e6);
    }
    return 0;
}

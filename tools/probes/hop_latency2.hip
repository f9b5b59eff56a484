// cross-stream dependency through stream memory operations (hipStreamWriteValue32 / hipStreamWaitValue32) instead of events
// build: hipcc --offload-arch=gfx950 -O2 -o tools/probes/hop_latency2 tools/probes/hop_latency2.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
__global__ void tiny(int* p) { if (threadIdx.x == 0) p[0] += 1; }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main() {
    int* d; CK(hipMalloc(&d, 64)); CK(hipMemset(d, 0, 64));
    uint32_t *f1, *f2;
    CK(hipExtMallocWithFlags((void**)&f1, 8, hipMallocSignalMemory));
    CK(hipExtMallocWithFlags((void**)&f2, 8, hipMallocSignalMemory));
    CK(hipMemset(f1, 0, 8)); CK(hipMemset(f2, 0, 8));
    hipStream_t s1, s2; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    const int N = 4000;
    for (int rep = 0; rep < 3; rep++) {
        CK(hipDeviceSynchronize());
        CK(hipMemset(f1, 0, 8)); CK(hipMemset(f2, 0, 8)); CK(hipDeviceSynchronize());
        double t0 = now();
        for (int i = 1; i <= N; i++) {
            hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s1, d);
            CK(hipStreamWriteValue32(s1, f1, (uint32_t)i, 0));
            CK(hipStreamWaitValue32(s2, f1, (uint32_t)i, hipStreamWaitValueGte, 0xffffffffu));
            hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s2, d);
            CK(hipStreamWriteValue32(s2, f2, (uint32_t)i, 0));
            CK(hipStreamWaitValue32(s1, f2, (uint32_t)i, hipStreamWaitValueGte, 0xffffffffu));
        }
        double th = now() - t0;
        CK(hipDeviceSynchronize());
        double b = now() - t0;
        printf("ping-pong with stream memory operations: %.2f us/kernel (host enqueue %.2f)\n", b / (2 * N) * 1e6, th / (2 * N) * 1e6);
    }
    return 0;
}

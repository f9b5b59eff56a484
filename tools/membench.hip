// tools/membench.hip -- HBM streaming ceilings on this MI355X for the access shapes the smoother uses.
// Build: hipcc --offload-arch=gfx950 -O3 tools/membench.hip -o tools/membench   Run: tools/membench [MiB per array]
// Prints GB/s (bytes moved / time) for copy / triad / read-only kernels at 8 and 16 bytes per lane.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <class T> __global__ void copy_k(const T* __restrict__ a, T* __restrict__ b, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) b[i] = a[i];
}
template <class T> __global__ void copy_gs(const T* __restrict__ a, T* __restrict__ b, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
__global__ void triad8(const double* __restrict__ a, const double* __restrict__ b, double* __restrict__ c, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) c[i] = a[i] + 3.0 * b[i];
}
__global__ void triad16(const double2* __restrict__ a, const double2* __restrict__ b, double2* __restrict__ c, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { double2 x = a[i], y = b[i]; c[i] = make_double2(x.x + 3.0 * y.x, x.y + 3.0 * y.y); }
}
// triad with a long dependent fp64 chain (one IEEE division) per element, like the smoother's update
__global__ void triad8_div(const double* __restrict__ a, const double* __restrict__ b, double* __restrict__ c, size_t n, double d) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) c[i] = (a[i] * d + b[i] * d + a[i] * 0.5) / (d + 2.0);
}
__global__ void read8(const double* __restrict__ a, double* __restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    double v = i < n ? a[i] : 0.0;
    if (v == 123.456) out[0] = v;
}
__global__ void read16(const double2* __restrict__ a, double* __restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    double2 v = i < n ? a[i] : make_double2(0, 0);
    if (v.x == 123.456 && v.y == 1.0) out[0] = v.x;
}
__global__ void write8(double* __restrict__ a, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] = 1.0;
}

// the x-split smoother's pattern: array a is made of rows of ROW doubles; even rows are read, odd rows are
// written (same array, reads and writes interleaved at ROW*8 bytes), b is a second read stream
__global__ void interleaved_rw(double* __restrict__ a, const double* __restrict__ b, size_t nrows, int ROW) {
    size_t r = blockIdx.x;
    for (int j = threadIdx.x; j < ROW; j += blockDim.x) {
        double x = a[(2 * r) * (size_t)ROW + j], y = b[r * (size_t)ROW + j];
        a[(2 * r + 1) * (size_t)ROW + j] = x + 3.0 * y;
    }
}
// same traffic with the written rows in a separate array c
__global__ void separate_rw(const double* __restrict__ a, const double* __restrict__ b, double* __restrict__ c, size_t nrows, int ROW) {
    size_t r = blockIdx.x;
    for (int j = threadIdx.x; j < ROW; j += blockDim.x) {
        double x = a[r * (size_t)ROW + j], y = b[r * (size_t)ROW + j];
        c[r * (size_t)ROW + j] = x + 3.0 * y;
    }
}

template <class F> double timeit(F f, int reps = 10) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    f(); CK(hipDeviceSynchronize());
    std::vector<float> t;
    for (int r = 0; r < reps; r++) { CK(hipEventRecord(e0)); f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); t.push_back(ms); }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2] * 1e-3;
}

int main(int argc, char** argv) {
    size_t mib = argc > 1 ? atol(argv[1]) : 1024;
    size_t n = mib * 1024 * 1024 / 8;  // doubles per array
    double *a, *b, *c;
    CK(hipMalloc(&a, n * 8)); CK(hipMalloc(&b, n * 8)); CK(hipMalloc(&c, n * 8));
    CK(hipMemset(a, 0, n * 8)); CK(hipMemset(b, 0, n * 8)); CK(hipMemset(c, 0, n * 8));
    const int B = 256;
    auto g = [&](size_t cnt) { return dim3((unsigned)((cnt + B - 1) / B)); };
    double s;
    s = timeit([&] { copy_k<double><<<g(n), B>>>(a, b, n); });            printf("copy   8B/lane            : %8.1f GB/s\n", 2.0 * n * 8 / s / 1e9);
    s = timeit([&] { copy_k<double2><<<g(n / 2), B>>>((double2*)a, (double2*)b, n / 2); }); printf("copy  16B/lane            : %8.1f GB/s\n", 2.0 * n * 8 / s / 1e9);
    s = timeit([&] { copy_gs<double2><<<2048, B>>>((double2*)a, (double2*)b, n / 2); });     printf("copy  16B/lane grid-stride: %8.1f GB/s\n", 2.0 * n * 8 / s / 1e9);
    s = timeit([&] { triad8<<<g(n), B>>>(a, b, c, n); });                  printf("triad  8B/lane (2R+1W)    : %8.1f GB/s\n", 3.0 * n * 8 / s / 1e9);
    s = timeit([&] { triad16<<<g(n / 2), B>>>((double2*)a, (double2*)b, (double2*)c, n / 2); }); printf("triad 16B/lane (2R+1W)    : %8.1f GB/s\n", 3.0 * n * 8 / s / 1e9);
    s = timeit([&] { triad8_div<<<g(n), B>>>(a, b, c, n, 1.25); });        printf("triad  8B/lane + fp64 div : %8.1f GB/s\n", 3.0 * n * 8 / s / 1e9);
    s = timeit([&] { read8<<<g(n), B>>>(a, c, n); });                      printf("read   8B/lane            : %8.1f GB/s\n", 1.0 * n * 8 / s / 1e9);
    s = timeit([&] { read16<<<g(n / 2), B>>>((double2*)a, c, n / 2); });   printf("read  16B/lane            : %8.1f GB/s\n", 1.0 * n * 8 / s / 1e9);
    s = timeit([&] { write8<<<g(n), B>>>(a, n); });                        printf("write  8B/lane            : %8.1f GB/s\n", 1.0 * n * 8 / s / 1e9);
    // misaligned (offset by one double) 16B/lane copy: what unpadded odd-length rows cost
    s = timeit([&] { copy_k<double2><<<g(n / 2 - 1), B>>>((double2*)(a + 1), (double2*)(b + 1), n / 2 - 1); }); printf("copy  16B/lane misaligned : %8.1f GB/s\n", 2.0 * n * 8 / s / 1e9);
    for (int ROW : {256, 257, 1024, 4096}) {
        size_t nrows = n / 2 / ROW;  // a holds 2*nrows rows
        s = timeit([&] { interleaved_rw<<<(unsigned)nrows, 256>>>(a, b, nrows, ROW); });
        printf("rows of %4d: read/write interleaved in one array: %8.1f GB/s", ROW, 3.0 * nrows * ROW * 8 / s / 1e9);
        s = timeit([&] { separate_rw<<<(unsigned)nrows, 256>>>(a, b, c, nrows, ROW); });
        printf("   separate arrays: %8.1f GB/s\n", 3.0 * nrows * ROW * 8 / s / 1e9);
    }
    return 0;
}

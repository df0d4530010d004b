// Practical HBM ceilings of the box for the three traffic mixes the transform kernels have: read-only, write-only, copy, and a
// 1-read : 2.7-write mix (the split input transform's) / 1.8-read : 1-write mix (the output transform's).  16 bytes per lane,
// grid-stride, nontemporal like the transforms' own streams.  Build: hipcc --offload-arch=gfx950 -O3 hbm_rw.hip -o hbm_rw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k_read(const f4 *a, size_t n, f4 *sink)
{
    f4 s = {0, 0, 0, 0};
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += __builtin_nontemporal_load(a + i);
    if (s.x == 12345.678f) sink[0] = s;
}
__global__ void k_write(f4 *a, size_t n)
{
    const f4 v = {1.0f, 2.0f, 3.0f, (float)threadIdx.x};
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) __builtin_nontemporal_store(v, a + i);
}
// R reads and W writes of 16 bytes per iteration (distinct streams)
template <int R, int W> __global__ void k_mix(const f4 *a, f4 *b, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        f4 s = {0, 0, 0, 0};
#pragma unroll
        for (int r = 0; r < R; ++r) s += __builtin_nontemporal_load(a + i + (size_t)r * n);
#pragma unroll
        for (int w = 0; w < W; ++w) __builtin_nontemporal_store(s, b + i + (size_t)w * n);
    }
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
template <class F> static float timeit(F f, int reps)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    return ms / reps;
}
int main()
{
    const size_t n = (size_t)1 << 26;                 // 64 Mi x 16 B = 1 GiB per stream
    f4 *a, *b;
    CK(hipMalloc(&a, n * 16 * 8)); CK(hipMalloc(&b, n * 16 * 8));
    CK(hipMemset(a, 1, n * 16 * 8)); CK(hipMemset(b, 0, n * 16 * 8));
    const dim3 g(256 * 16), t(256);
    float ms;
    ms = timeit([&] { hipLaunchKernelGGL(k_read, g, t, 0, 0, a, n * 8, b); }, 5);            printf("read  8 GiB          %.3f ms  %.2f TB/s\n", ms, 8 * 1.0737 / ms);
    ms = timeit([&] { hipLaunchKernelGGL(k_write, g, t, 0, 0, b, n * 8); }, 5);              printf("write 8 GiB          %.3f ms  %.2f TB/s\n", ms, 8 * 1.0737 / ms);
    ms = timeit([&] { hipLaunchKernelGGL((k_mix<1, 1>), g, t, 0, 0, a, b, n * 4); }, 5);     printf("copy  4 + 4 GiB      %.3f ms  %.2f TB/s\n", ms, 8 * 1.0737 / ms);
    ms = timeit([&] { hipLaunchKernelGGL((k_mix<3, 8>), g, t, 0, 0, a, b, n); }, 5);         printf("3 R : 8 W  (1 : 2.7) %.3f ms  %.2f TB/s\n", ms, 11 * 1.0737 / ms);
    ms = timeit([&] { hipLaunchKernelGGL((k_mix<7, 4>), g, t, 0, 0, a, b, n); }, 5);         printf("7 R : 4 W  (1.8 : 1) %.3f ms  %.2f TB/s\n", ms, 11 * 1.0737 / ms);
    ms = timeit([&] { hipLaunchKernelGGL((k_mix<1, 8>), g, t, 0, 0, a, b, n); }, 5);         printf("1 R : 8 W            %.3f ms  %.2f TB/s\n", ms, 9 * 1.0737 / ms);
    return 0;
}

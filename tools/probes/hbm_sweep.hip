// Probe: does ANY plain streaming form beat tools/micro/hbm_rw's ceilings on this box?  Sweeps grid size, block size, unroll,
// address pattern (grid-stride / contiguous chunk per workgroup) and cache policy (plain / nt / sc1 / sc0 sc1) for write-only,
// read-only, copy and the transforms' mixes.  Build: hipcc --offload-arch=gfx950 -O3 hbm_sweep.hip -o hbm_sweep
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));
enum { POL_PLAIN = 0, POL_NT = 1, POL_SC1 = 2, POL_SC01 = 3, POL_SC01NT = 4 };
template <int POL> __device__ __forceinline__ void st(f4 *p, f4 v)
{
    if constexpr (POL == POL_PLAIN) *p = v;
    else if constexpr (POL == POL_NT) __builtin_nontemporal_store(v, p);
    else if constexpr (POL == POL_SC1) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    else if constexpr (POL == POL_SC01) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" ::"v"(p), "v"(v) : "memory");
}
template <int POL> __device__ __forceinline__ f4 ld(const f4 *p)
{
    if constexpr (POL == POL_PLAIN) return *p;
    else if constexpr (POL == POL_NT) return __builtin_nontemporal_load(p);
    else {
        f4 v;
        if constexpr (POL == POL_SC1) asm volatile("global_load_dwordx4 %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
        else if constexpr (POL == POL_SC01) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
        else asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1 nt\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
        return v;
    }
}
// R read streams, W write streams of n elements each; U elements per thread and trip; CH: contiguous chunk per workgroup
template <int R, int W, int U, int POL, bool CH> __global__ void k(const f4 *a, f4 *b, size_t n)
{
    size_t beg, end, step;
    if constexpr (CH) {
        const size_t per = (n + gridDim.x - 1) / gridDim.x;
        beg = blockIdx.x * per + threadIdx.x; end = min(n, (blockIdx.x + 1) * per); step = (size_t)blockDim.x * U;
    } else {
        beg = (size_t)blockIdx.x * blockDim.x * U + threadIdx.x; end = n; step = (size_t)gridDim.x * blockDim.x * U;
    }
    for (size_t i = beg; i < end; i += step) {
        f4 s[U];
#pragma unroll
        for (int u = 0; u < U; ++u) s[u] = f4{1.0f, 2.0f, 3.0f, (float)threadIdx.x};
        if constexpr (R > 0) {
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int u = 0; u < U; ++u) { const size_t j = i + (size_t)u * blockDim.x; if (j < end) s[u] += ld<POL>(a + j + (size_t)r * n); }
        }
        if constexpr (W > 0) {
#pragma unroll
            for (int w = 0; w < W; ++w)
#pragma unroll
                for (int u = 0; u < U; ++u) { const size_t j = i + (size_t)u * blockDim.x; if (j < end) st<POL>(b + j + (size_t)w * n, s[u]); }
        } else {
            f4 t = s[0];
#pragma unroll
            for (int u = 1; u < U; ++u) t += s[u];
            if (t.x == 12345.678f) b[0] = t;
        }
    }
}
template <class F> static float timeit(F f, int reps)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    hipEventDestroy(a); hipEventDestroy(b);
    return ms / reps;
}
static f4 *A, *B;
template <int R, int W, int U, int POL, bool CH> static void run(const char *name, size_t n_total /* elements over all streams */)
{
    const size_t n = n_total / (R + W);
    const char *pol[] = {"plain", "nt", "sc1", "sc0sc1", "sc0sc1nt"};
    for (int wg : {256, 512, 1024, 2048, 4096, 16384})
        for (int bs : {256, 1024}) {
            const float ms = timeit([&] { hipLaunchKernelGGL((k<R, W, U, POL, CH>), dim3(wg), dim3(bs), 0, 0, A, B, n); }, 3);
            printf("%-6s U=%d %-8s %-5s grid %5d x %4d   %.3f ms  %.2f TB/s\n", name, U, pol[POL], CH ? "chunk" : "strid", wg, bs, ms, (double)n * (R + W) * 16 / ms * 1e-9);
        }
}
template <int R, int W> static void mix(const char *name)
{
    const size_t tot = (size_t)1 << 29;      // 8 GiB over all streams
    run<R, W, 1, POL_NT, false>(name, tot);
    run<R, W, 4, POL_NT, false>(name, tot);
    run<R, W, 1, POL_PLAIN, false>(name, tot);
    run<R, W, 4, POL_PLAIN, false>(name, tot);
    run<R, W, 4, POL_NT, true>(name, tot);
    run<R, W, 4, POL_SC1, false>(name, tot);
    run<R, W, 4, POL_SC01, false>(name, tot);
    run<R, W, 4, POL_SC01NT, false>(name, tot);
}
int main(int argc, char **argv)
{
    const size_t bytes = (size_t)9 << 30;
    if (hipMalloc(&A, bytes) != hipSuccess || hipMalloc(&B, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(A, 1, bytes); hipMemset(B, 0, bytes);
    const int which = argc > 1 ? atoi(argv[1]) : -1;
    if (which < 0 || which == 0) mix<0, 1>("write");
    if (which < 0 || which == 1) mix<1, 0>("read");
    if (which < 0 || which == 2) mix<1, 1>("copy");
    if (which < 0 || which == 3) mix<7, 4>("7R4W");
    if (which < 0 || which == 4) mix<3, 8>("3R8W");
    return 0;
}

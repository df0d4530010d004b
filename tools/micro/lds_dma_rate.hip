// What one CU can pull out of L2 per clock: every workgroup (one per CU, 8 waves) streams a 48 KiB L2-resident block into LDS over
// and over, (a) by LDS-DMA (global_load_lds_dwordx4, the path of wino4s_fused.hip and wino_gemm_s3.hip), (b) through registers
// (global_load_dwordx4 + ds_write_b128).  SHARED = 1: all workgroups read the SAME block (the fused kernel's U stage), 0: each its own.
// Build: hipcc --offload-arch=gfx950 -O3 lds_dma_rate.hip -o lds_dma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(1))) const void gptr_t;
typedef __attribute__((address_space(3))) void lptr_t;
typedef float f4 __attribute__((ext_vector_type(4)));
#define PIECES 48          // 1 KiB pieces per pass = 48 KiB
template <bool DMA> __global__ __launch_bounds__(512) void k(const float *src, long long stride, int passes, unsigned long long *cyc, float *sink)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float *base = src + (long long)blockIdx.x * stride;
    f4 acc = {0, 0, 0, 0};
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < passes; ++it) {
#pragma unroll
        for (int i = 0; i < PIECES / 8; ++i) {
            const int piece = wave + 8 * i;
            if (DMA) __builtin_amdgcn_global_load_lds((gptr_t *)(base + piece * 256 + lane * 4), (lptr_t *)(lds + piece * 256), 16, 0, 0);
            else *reinterpret_cast<f4 *>(lds + piece * 256 + lane * 4) = *reinterpret_cast<const f4 *>(base + piece * 256 + lane * 4);
        }
        __builtin_amdgcn_s_waitcnt(0x0f70);
        __builtin_amdgcn_s_barrier();
        acc += *reinterpret_cast<f4 *>(lds + ((it * 64 + lane) & 4095) * 4);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    if (acc.x == 12345.678f) sink[0] = acc.y;
}
int main()
{
    const int nwg = 256, passes = 2000;
    float *src, *sink; unsigned long long *cyc;
    hipMalloc(&src, (size_t)nwg * 49152 + 4096); hipMemset(src, 0, (size_t)nwg * 49152 + 4096);
    hipMalloc(&sink, 64); hipMalloc(&cyc, nwg * 8);
    unsigned long long h[256];
    for (int dma = 1; dma >= 0; --dma)
        for (int shared = 1; shared >= 0; --shared) {
            const long long stride = shared ? 0 : 49152 / 4;
            for (int rep = 0; rep < 2; ++rep) {
                hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
                hipEventRecord(a);
                if (dma) hipLaunchKernelGGL(k<true>, dim3(nwg), dim3(512), 49152, 0, src, stride, passes, cyc, sink);
                else hipLaunchKernelGGL(k<false>, dim3(nwg), dim3(512), 49152, 0, src, stride, passes, cyc, sink);
                hipEventRecord(b); hipEventSynchronize(b);
                float ms = 0; hipEventElapsedTime(&ms, a, b);
                hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
                double c = 0; for (int i = 0; i < nwg; ++i) c += (double)h[i];
                c /= nwg;
                if (rep) printf("%-9s %-6s: %.0f cycles per 48 KiB pass (readcyclecounter, 100 MHz ticks x clock ratio unknown) | wall %.3f ms -> %.1f B per ns per CU, %.2f TB/s chip\n",
                                dma ? "LDS-DMA" : "registers", shared ? "shared" : "own", c / passes, ms, 49152.0 * passes / (ms * 1e6), 49152.0 * passes * nwg / (ms * 1e-3) / 1e12);
            }
        }
    return 0;
}

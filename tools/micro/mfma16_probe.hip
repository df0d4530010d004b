// micro-probe: issue rate of v_mfma_f32_16x16x4_f32 with 36 independent accumulators, 8 waves per CU (2 per SIMD),
// with operand traffic interleaved: mode 1 = one global_load_dword per MFMA (the B stream of a fused Winograd kernel),
// mode 8 = one ds_read2 + one global_load_dword per TWO MFMAs (wino4_fused.hip's position-split layout), for 1..32
// private copies of the streamed buffer (same rate: the limit is per CU, not an L2 hot spot).  Measured on MI355X:
// no operand traffic 149 TFLOP/s, mode 1 85, mode 8 110.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma16_probe.hip -o tools/micro/mfma16_probe && tools/micro/mfma16_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(512) void probe(const float *u, float *out, int iters, int ncopy)
{
    __shared__ float lds[36 * 256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 36 * 256; i += 512) lds[i] = 1.0f + i * 1e-6f;
    __syncthreads();
    f32x4 acc[36];
#pragma unroll
    for (int q = 0; q < 36; ++q) acc[q] = f32x4{0, 0, 0, 0};
    float b[36], a[36];
    const float *ub = u + (long long)(blockIdx.x % ncopy) * (16ll * 8 * 36 * 64) + wave * 36 * 64 + lane;
    const float *ub4 = u + (long long)(blockIdx.x % ncopy) * (16ll * 8 * 36 * 64) + wave * 36 * 64 + lane * 4;
#pragma unroll
    for (int q = 0; q < 36; ++q) { b[q] = ub[q * 64]; a[q] = lds[q * 256 + lane]; }
#pragma unroll 1
    for (int it = 0; it < (MODE >= 8 ? 0 : iters); ++it) {
        const float *un = ub + (long long)((it + 1) & 15) * 8 * 36 * 64;
#pragma unroll
        for (int q = 0; q < 36; ++q) {
            acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q], b[q], acc[q], 0, 0, 0);
            if (MODE & 1) b[q] = un[q * 64];
            if ((MODE & 4) && (q & 3) == 3) {      // one dwordx4 per four MFMAs: lane's 4 positions contiguous
                const f32x4 t = *reinterpret_cast<const f32x4 *>(ub4 + (long long)((it + 1) & 15) * 8 * 36 * 64 - lane + (q >> 2) * 256);
                b[q - 3] = t[0]; b[q - 2] = t[1]; b[q - 1] = t[2]; b[q] = t[3];
            }
            if (MODE & 2) a[q] = lds[q * 256 + ((lane + it) & 63)];
        }
    }
    if (MODE == 8) {      // the position-split layout: per iteration 18 x (one ds_read2 -> 2 MFMAs sharing one B dword, B reloaded)
#pragma unroll 1
        for (int it = 0; it < iters; ++it) {
            const float *un = ub + (long long)((it + 1) & 15) * 8 * 36 * 64;
            const float *va = lds + ((it & 1) * 18) * 256 + lane;
#pragma unroll
            for (int i = 0; i < 18; ++i) {
                const float x0 = va[i * 256], x1 = va[i * 256 + 128];
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(x0, b[i], acc[i], 0, 0, 0);
                acc[18 + i] = __builtin_amdgcn_mfma_f32_16x16x4f32(x1, b[i], acc[18 + i], 0, 0, 0);
                b[i] = un[i * 64];
            }
        }
    }
    if (MODE == 16) {     // position split, B as dwordx4: one load per 8 MFMAs (18 positions = 4 quads + 1 pair)
#pragma unroll 1
        for (int it = 0; it < iters; ++it) {
            const float *un = ub4 + (long long)((it + 1) & 15) * 8 * 36 * 64;
            const float *va = lds + ((it & 1) * 18) * 256 + lane;
#pragma unroll
            for (int i = 0; i < 18; ++i) {
                const float x0 = va[i * 256], x1 = va[i * 256 + 128];
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(x0, b[i], acc[i], 0, 0, 0);
                acc[18 + i] = __builtin_amdgcn_mfma_f32_16x16x4f32(x1, b[i], acc[18 + i], 0, 0, 0);
                if ((i & 3) == 3) {
                    const f32x4 t = *reinterpret_cast<const f32x4 *>(un + (i >> 2) * 256);
                    b[i - 3] = t[0]; b[i - 2] = t[1]; b[i - 1] = t[2]; b[i] = t[3];
                }
                if (i == 17) {
                    const float2 t = *reinterpret_cast<const float2 *>(un + 4 * 256 - 2 * lane);
                    b[16] = t.x; b[17] = t.y;
                }
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int q = 0; q < 36; ++q) s += acc[q][0] + acc[q][1] + acc[q][2] + acc[q][3];
    out[blockIdx.x * 512 + tid] = s;
}

int main()
{
    float *u, *out;
    hipMalloc(&u, 32 * 16ll * 8 * 36 * 64 * 4 + 4096);
    hipMemset(u, 0, 32 * 16ll * 8 * 36 * 64 * 4 + 4096);
    hipMalloc(&out, 256 * 512 * 4 * 4);
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int modes[] = {0, 1, 4, 8, 16};
    for (int ncopy = 1; ncopy <= 32; ncopy *= 32)
    for (int mode : modes) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(256), dim3(512), 0, 0, u, out, iters, ncopy);
            if (mode == 1) hipLaunchKernelGGL(probe<1>, dim3(256), dim3(512), 0, 0, u, out, iters, ncopy);
            if (mode == 2) hipLaunchKernelGGL(probe<2>, dim3(256), dim3(512), 0, 0, u, out, iters, ncopy);
            if (mode == 3) hipLaunchKernelGGL(probe<3>, dim3(256), dim3(512), 0, 0, u, out, iters, ncopy);
            if (mode == 4) hipLaunchKernelGGL(probe<4>, dim3(256), dim3(512), 0, 0, u, out, iters, ncopy);
            if (mode == 5 || mode == 7) continue;
            if (mode == 8) hipLaunchKernelGGL(probe<8>, dim3(256), dim3(512), 0, 0, u, out, iters, ncopy);
            if (mode == 16) hipLaunchKernelGGL(probe<16>, dim3(256), dim3(512), 0, 0, u, out, iters, ncopy);
            if (mode == 6) hipLaunchKernelGGL(probe<6>, dim3(256), dim3(512), 0, 0, u, out, iters, ncopy);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double fl = 256.0 * 8 * iters * 36 * 16 * 16 * 4 * 2;
        printf("copies %d mode %d (1: +global_load_dword per MFMA, 2: +ds_read per MFMA, 4: +global_load_dwordx4 per 4 MFMAs): %.3f ms  %.1f TFLOP/s\n", ncopy, mode, ms, fl / ms / 1e9);
    }
    return 0;
}

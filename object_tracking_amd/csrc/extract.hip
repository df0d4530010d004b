// extract.hip -- small elementwise kernels behind dt_detector_extract (KerasYOLO.extract for ANY layer name,
// models_detection/KerasYOLO.py:509-520).  Not on the hot path: extract() is a one-image debugging call.
#include "dt_internal.h"

// frames [n_pix][3] uint8 (through the x/255 table) or float32 -> [n_pix][32] float32, channels 3..31 zero:
// conv_1 as a Cin = 32 layer of the MFMA kernel, so that its un-pooled, un-activated outputs can be read out
__global__ void expand_rgb32_kernel(const void *frames, int dtype, long long n_pix, const float *lut, float *out)
{
    const long long total = n_pix * 8;   // one float4 per thread
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const long long p = e >> 3;
        const int q = (int)(e & 7);
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (q == 0) {
            if (dtype == DT_FRAMES_U8) {
                const unsigned char *s = reinterpret_cast<const unsigned char *>(frames) + p * 3;
                v[0] = lut[s[0]]; v[1] = lut[s[1]]; v[2] = lut[s[2]];
            } else {
                const float *s = reinterpret_cast<const float *>(frames) + p * 3;
                v[0] = s[0]; v[1] = s[1]; v[2] = s[2];
            }
        }
        *reinterpret_cast<f32x4 *>(out + p * 32 + q * 4) = v;
    }
}

int launch_expand_rgb32(hipStream_t st, const void *frames, int dtype, long long n_pix, const float *lut, float *out)
{
    long long blocks = (n_pix * 8 + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(expand_rgb32_kernel, dim3((unsigned)blocks), dim3(256), 0, st, frames, dtype, n_pix, lut, out);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

// x[r][c] = (x[r][c] - shift[c]) / scale[c]: the Conv2D output under a folded BatchNorm (y = scale * conv + shift)
__global__ void unfold_bn_kernel(float *x, long long rows, int C, const float *scale, const float *shift)
{
    const long long total = rows * C;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(e % C);
        x[e] = (x[e] - shift[c]) / scale[c];
    }
}

int launch_unfold_bn(hipStream_t st, float *x, long long rows, int C, const float *scale, const float *shift)
{
    long long blocks = (rows * C + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(unfold_bn_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, rows, C, scale, shift);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

// conv1.hip -- first block of the detector, fused:
//   normalize (utility/utils.py:150-153: x/255.)  ->  Conv2D(32,(3,3),'same')
//   -> folded BatchNorm -> LeakyReLU(0.1) -> MaxPooling2D(2,2)
// (models_detection/KerasYOLO.py:278-282).
//
// Cin = 3, K = 27: not a dense contraction worth an MFMA tile, and the frame is
// read exactly once -- this is a direct convolution on the packed-FMA VALU path.  A
// workgroup owns an 8x8 tile of POOLED output pixels (16x16 conv pixels, 18x18x3 input
// patch staged in LDS as RGBx float4, uint8 -> float through a 256-entry table so that
// the result equals the reference's float64 x/255. rounded to float32).  Wave = one group
// of 8 output channels, lane = pooled pixel: the 27 x 8 weights of a wave are wave-uniform,
// so they arrive through scalar loads and feed v_pk_fma_f32 from SGPRs -- no LDS reads
// for weights in the inner loop (the first version read them from LDS per thread and
// spent half its cycles there).  4 conv positions x 8 channels accumulate in registers,
// the 2x2 max is taken in registers; the four waves of a workgroup complete each pixel's
// 128-byte NHWC line.
#include "dt_internal.h"

struct Conv1Args {
    const void *frames;
    int dtype;
    int B, H, W;
    const float *w;      // [27][32]  (ky,kx,ci) x cout, BN scale folded
    const float *bias;   // [32]
    const float *lut;    // [256]
    float slope;
    float *out;          // [B][H/2][W/2][32]
};

__global__ __launch_bounds__(256) void conv1_direct_kernel(Conv1Args p)
{
    __shared__ __attribute__((aligned(16))) float s_patch[18 * 18 * 4];
    __shared__ float s_lut[256];

    const int tid = threadIdx.x;
    const int H2 = p.H >> 1, W2 = p.W >> 1;
    const int bx = blockIdx.x, by = blockIdx.y, b = blockIdx.z;
    const int cy0 = by * 16 - 1, cx0 = bx * 16 - 1;   // patch origin in input pixels

    s_lut[tid] = p.lut[tid];          // 256 threads, 256 entries: the x/255 table moves to LDS once per workgroup
    __syncthreads();
    for (int i = tid; i < 18 * 18; i += 256) {
        const int r = i / 18, c = i - r * 18;
        const int y = cy0 + r, x = cx0 + c;
        float v0 = 0.f, v1 = 0.f, v2 = 0.f;
        if (y >= 0 && y < p.H && x >= 0 && x < p.W) {
            const long long off = (((long long)b * p.H + y) * p.W + x) * 3;
            if (p.dtype == DT_FRAMES_U8) {
                const unsigned char *s = reinterpret_cast<const unsigned char *>(p.frames) + off;
                v0 = s_lut[s[0]]; v1 = s_lut[s[1]]; v2 = s_lut[s[2]];
            } else {
                const float *s = reinterpret_cast<const float *>(p.frames) + off;
                v0 = s[0]; v1 = s[1]; v2 = s[2];
            }
        }
        f32x4 v = {v0, v1, v2, 0.f};
        *reinterpret_cast<f32x4 *>(&s_patch[i * 4]) = v;
    }
    __syncthreads();

    const int g = __builtin_amdgcn_readfirstlane(tid >> 6);   // channel group (8 channels): wave-uniform
    const int pp = tid & 63;                                  // pooled pixel in tile
    const int py = pp >> 3, px = pp & 7;
    const float *wg = p.w + g * 8;                            // uniform address -> scalar loads

    float in[4][4][3];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(&s_patch[((2 * py + r) * 18 + 2 * px + c) * 4]);
            in[r][c][0] = v[0]; in[r][c][1] = v[1]; in[r][c][2] = v[2];
        }

    float acc[4][8];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[q][c] = 0.0f;

#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int ci = 0; ci < 3; ++ci) {
                const int t = (ky * 3 + kx) * 3 + ci;
                const f32x4 w0 = *reinterpret_cast<const f32x4 *>(wg + t * 32);
                const f32x4 w1 = *reinterpret_cast<const f32x4 *>(wg + t * 32 + 4);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float x = in[(q >> 1) + ky][(q & 1) + kx][ci];
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        acc[q][c] = __fmaf_rn(x, w0[c], acc[q][c]);
                        acc[q][4 + c] = __fmaf_rn(x, w1[c], acc[q][4 + c]);
                    }
                }
            }

    const int oy = by * 8 + py, ox = bx * 8 + px;
    if (oy < H2 && ox < W2) {
        f32x4 o0, o1;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float bv = p.bias[g * 8 + c];
            float m = -INFINITY;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float v = acc[q][c] + bv;
                v = v > 0.0f ? v : v * p.slope;
                m = fmaxf(m, v);
            }
            if (c < 4) o0[c] = m; else o1[c - 4] = m;
        }
        float *o = p.out + ((((long long)b * H2 + oy) * W2 + ox) * 32 + g * 8);
        *reinterpret_cast<f32x4 *>(o) = o0;
        *reinterpret_cast<f32x4 *>(o + 4) = o1;
    }
}

int launch_conv1_direct(hipStream_t st, const void *frames, int dtype, int B, int H, int W, const float *w_packed,
                        const float *bias, const float *lut, float slope, float *out)
{
    if ((H & 1) || (W & 1) || B <= 0) return 2;
    Conv1Args a;
    a.frames = frames; a.dtype = dtype; a.B = B; a.H = H; a.W = W;
    a.w = w_packed; a.bias = bias; a.lut = lut; a.slope = slope; a.out = out;
    const int H2 = H / 2, W2 = W / 2;
    dim3 grid((unsigned)((W2 + 7) / 8), (unsigned)((H2 + 7) / 8), (unsigned)B);
    hipLaunchKernelGGL(conv1_direct_kernel, grid, dim3(256), 0, st, a);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

// wino_fused.hip -- conv_2 (Conv2D(64,(3,3)) on 32 channels + BN + LeakyReLU + MaxPooling2D(2,2),
// models_detection/KerasYOLO.py:285-288) as ONE fused Winograd F(2x2,3x3) kernel on the matrix cores.
//
// The unfused Winograd form (winograd.hip) does not pay here: with K = Cin = 32 its batched GEMMs would move
// more bytes through HBM than the MFMA work they save.  Fused, V and M' never leave the CU:
//   * a workgroup owns 8x8 output tiles of 2x2 pixels (= 8x8 POOLED pixels: a tile is exactly one pooling
//     window) of one frame; the 18x18x32 input patch is staged once in LDS (41 KiB, 16-byte slots XOR-swizzled
//     by the pixel column so that the fragment reads below are bank-conflict free);
//   * for each of the 16 Winograd positions p = (xi, nu): every lane forms V_p for ITS tile and 16 of the 32
//     channels straight from the patch (4 pixels x 4 ds_read_b128, three add/sub per value) -- that IS the MFMA
//     A operand (row = tile, k = channel) -- and multiplies it with U_p = (G g Gt)[xi][nu] (32 x 32 block of the
//     wave's output channels, streamed from L2) in 16 v_mfma_f32_32x32x2_f32 steps;
//   * the output transform At M A has coefficients 0, +-1 only, so each M'_p is added to / subtracted from
//     the four 2x2-output accumulators as it is produced; bias + LeakyReLU + the 2x2 max finish in registers.
// MFMA work is 16/36 of the direct form's (2.25x less); the rounding error is the direct form's (F(2x2,3x3)
// only adds, subtracts and halves).  Wave = 32 tiles x 32 output channels, four waves per workgroup.
#include "dt_internal.h"

#ifndef WF_ABLATE
#define WF_ABLATE 0                  // timing-only ablation builds (tools/ablate_wf.sh): 1 no B reloads (-6.5 %), 2 no A formation (-2 %); both at once lets the compiler fold the positions
#endif
#define WF_T 8                       // tiles per workgroup side
#define WF_P (2 * WF_T + 2)          // patch side in pixels (18)
#define WF_C 32                      // input channels
#define WF_N 64                      // output channels

__device__ __forceinline__ f32x4 wf_ld4(const float *p) { return *reinterpret_cast<const f32x4 *>(p); }

__global__ __launch_bounds__(256, 2) void wino2_fused_pool_kernel(WinoFusedArgs p)
{
    extern __shared__ __attribute__((aligned(16))) float patch[];   // [WF_P][WF_P][32], slot-swizzled
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int H2 = p.H >> 1, W2 = p.W >> 1;
    const int nbx = (W2 + WF_T - 1) / WF_T, nby = (H2 + WF_T - 1) / WF_T;
    int bid = blockIdx.x;
    const int bx = bid % nbx;
    bid /= nbx;
    const int by = bid % nby, b = bid / nby;

    // ---- stage the input patch: pixel (2*WF_T*by - 1 + pr, 2*WF_T*bx - 1 + pc), zero outside the image ----
    const float *img = p.in + (long long)b * p.H * p.W * WF_C;
    const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int idx = tid; idx < WF_P * WF_P * 8; idx += 256) {
        const int slot = idx & 7, pix = idx >> 3;
        const int pr = pix / WF_P, pc = pix - pr * WF_P;
        const int h = 2 * WF_T * by - 1 + pr, w = 2 * WF_T * bx - 1 + pc;
        const bool ok = h >= 0 && h < p.H && w >= 0 && w < p.W;
        const f32x4 v = ok ? wf_ld4(img + ((long long)h * p.W + w) * WF_C + slot * 4) : zero;
        *reinterpret_cast<f32x4 *>(&patch[pix * WF_C + ((slot ^ ((pc >> 1) & 7)) << 2)]) = v;
    }
    __syncthreads();

    // ---- lane geometry: MFMA row = tile r of the wave's 32 (4 rows x 8 cols), k half hi -> channels hi*16.. ----
    const int r = lane & 31, hi = lane >> 5;
    const int ty = wm * 4 + (r >> 3), tx = r & 7;
    const float *pbase = patch + ((2 * ty) * WF_P + 2 * tx) * WF_C;   // patch pixel (2ty, 2tx) of this lane's tile
    // B operand: U laid out [pos][wn][n 32][hi 2][16 k] -> 64 contiguous bytes per lane and position
    const float *ub = p.u + (((long long)wn * 32 + r) * 2 + hi) * 16;
    const long long ustride = 2 * 32 * 2 * 16;

    f32x16 Y[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int e = 0; e < 16; ++e) Y[a][c][e] = 0.0f;

    // V_p for this lane's tile and 16 channels, straight from the patch
    auto form_v = [&](int pos, f32x4 *v) {
        const int xi = pos >> 2, nu = pos & 3;
        // Bt rows: 0: d0 - d2, 1: d1 + d2, 2: d2 - d1, 3: d1 - d3  ->  (first, second, sign of second)
        const int i1 = xi == 0 ? 0 : (xi == 2 ? 2 : 1), i2 = xi == 0 ? 2 : (xi == 1 ? 2 : (xi == 2 ? 1 : 3));
        const int j1 = nu == 0 ? 0 : (nu == 2 ? 2 : 1), j2 = nu == 0 ? 2 : (nu == 1 ? 2 : (nu == 2 ? 1 : 3));
        const bool si_plus = xi == 1, sj_plus = nu == 1;
        const int sw1 = (tx + (j1 >> 1)) & 7, sw2 = (tx + (j2 >> 1)) & 7;   // ((2tx + j) >> 1) & 7
        const float *p11 = pbase + (i1 * WF_P + j1) * WF_C, *p12 = pbase + (i1 * WF_P + j2) * WF_C;
        const float *p21 = pbase + (i2 * WF_P + j1) * WF_C, *p22 = pbase + (i2 * WF_P + j2) * WF_C;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int s1 = ((hi * 4 + q) ^ sw1) << 2, s2 = ((hi * 4 + q) ^ sw2) << 2;
            const f32x4 a = wf_ld4(p11 + s1), bb = wf_ld4(p12 + s2), c = wf_ld4(p21 + s1), d = wf_ld4(p22 + s2);
            const f32x4 top = sj_plus ? a + bb : a - bb, bot = sj_plus ? c + d : c - d;
            v[q] = si_plus ? top + bot : top - bot;
        }
    };

    // Software pipeline over the 16 positions: while the 16 dependent MFMAs of position p run, the lane forms
    // V_{p+1} (LDS reads + adds) and fetches U_{p+1}; the +-1 output transform of M'_p follows.
    f32x4 bq[4], vq[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) bq[q] = wf_ld4(ub + q * 4);
    form_v(0, vq);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int pos = 0; pos < 16; ++pos) {
        const int xi = pos >> 2, nu = pos & 3;
        f32x16 m;
#pragma unroll
        for (int e = 0; e < 16; ++e) m[e] = 0.0f;
#pragma unroll
        for (int s = 0; s < 16; ++s) m = __builtin_amdgcn_mfma_f32_32x32x2f32(vq[s >> 2][s & 3], bq[s >> 2][s & 3], m, 0, 0, 0);
        f32x4 bn[4], vn[4];
        if (pos < 15) {
#pragma unroll
            for (int q = 0; q < 4; ++q) bn[q] = (WF_ABLATE & 1) ? bq[q] : wf_ld4(ub + (pos + 1) * ustride + q * 4);
            if (WF_ABLATE & 2) { for (int q = 0; q < 4; ++q) vn[q] = vq[q]; } else
            form_v(pos + 1, vn);
        }
        // At = [[1,1,1,0],[0,1,-1,-1]]: Y[a][c] += At[a][xi] * At[c][nu] * M'
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int ca = a == 0 ? (xi < 3 ? 1 : 0) : (xi == 0 ? 0 : (xi == 1 ? 1 : -1));
                const int cc = c == 0 ? (nu < 3 ? 1 : 0) : (nu == 0 ? 0 : (nu == 1 ? 1 : -1));
                if (ca * cc == 1) Y[a][c] += m;
                if (ca * cc == -1) Y[a][c] -= m;
            }
        if (pos < 15) {
#pragma unroll
            for (int q = 0; q < 4; ++q) { bq[q] = bn[q]; vq[q] = vn[q]; }
        }
        __builtin_amdgcn_sched_barrier(0);   // keep the positions apart: hoisting further ahead costs registers
    }

    // ---- bias + LeakyReLU + 2x2 max; C/D layout: col = lane & 31 (output channel), row = tile ----
    const int n = wn * 32 + r;
    const float bv = p.bias[n];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int rt = (e & 3) + 8 * (e >> 2) + 4 * hi;            // tile index within the wave's 32
        const int py = by * WF_T + wm * 4 + (rt >> 3), px = bx * WF_T + (rt & 7);
        float mx = -INFINITY;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                float y = Y[a][c][e] + bv;
                y = y > 0.0f ? y : y * p.slope;
                mx = fmaxf(mx, y);
            }
        if (py < H2 && px < W2) p.out[(((long long)b * H2 + py) * W2 + px) * WF_N + n] = mx;
    }
}

int launch_wino2_fused_pool(hipStream_t st, const WinoFusedArgs &a)
{
    if (a.B <= 0 || (a.H & 1) || (a.W & 1)) return 2;
    const int H2 = a.H / 2, W2 = a.W / 2;
    const long long blocks = (long long)a.B * ((H2 + WF_T - 1) / WF_T) * ((W2 + WF_T - 1) / WF_T);
    if (blocks >= (1ll << 31)) return 2;
    const size_t lds = (size_t)WF_P * WF_P * WF_C * sizeof(float);
    hipLaunchKernelGGL(wino2_fused_pool_kernel, dim3((unsigned)blocks), dim3(256), lds, st, a);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

// Host: U_p = (G g Gt)[xi][nu] (F(2x2,3x3), BN scale folded) in the kernel's B-operand order
//   dst[pos 16][wn 2][n 32][hi 2][16]:  element = U_pos[cin = hi*16 + s][cout = wn*32 + n]
void wino2_fused_pack(const float *hwio /*[3][3][32][64]*/, const float *scale /*[64] or null*/, float *dst)
{
    static const double G[4][3] = {{1, 0, 0}, {.5, .5, .5}, {.5, -.5, .5}, {0, 0, 1}};
    for (int co = 0; co < WF_N; ++co)
        for (int ci = 0; ci < WF_C; ++ci) {
            double g[3][3], t[4][3];
            const double sc = scale ? (double)scale[co] : 1.0;
            for (int k = 0; k < 9; ++k) g[k / 3][k % 3] = (double)hwio[((size_t)k * WF_C + ci) * WF_N + co] * sc;
            for (int xi = 0; xi < 4; ++xi)
                for (int kx = 0; kx < 3; ++kx) t[xi][kx] = G[xi][0] * g[0][kx] + G[xi][1] * g[1][kx] + G[xi][2] * g[2][kx];
            const int wn = co >> 5, n = co & 31, hi = ci >> 4, s = ci & 15;
            for (int xi = 0; xi < 4; ++xi)
                for (int nu = 0; nu < 4; ++nu)
                    dst[((((size_t)(4 * xi + nu) * 2 + wn) * 32 + n) * 2 + hi) * 16 + s] =
                        (float)(t[xi][0] * G[nu][0] + t[xi][1] * G[nu][1] + t[xi][2] * G[nu][2]);
        }
}

// winograd.hip -- Winograd F(2x2, 3x3) form of the reference's 3x3 'same' stride-1 convolutions
// (models_detection/KerasYOLO.py:279-396 conv blocks, models_tracking/MultiObjDetTracker.py:176
// ConvLSTM2D input and recurrent convolutions) for the wide layers, where it pays:
//
//     Y = At [ (G g Gt) .* (Bt d B) ] A          per 2x2 output tile, 4x4 input patch d, 3x3 filter g
//
// 16 multiplies per 4 outputs instead of 36: the MFMA work of a layer drops 2.25x (1.94x at 13x13,
// whose 7x7 tiles cover 14x14).  The contraction over input channels becomes 16 independent GEMMs
//     M'[p][tile][cout] = sum_c V[p][tile][c] * U[p][cout][c],      p = 4*xi + nu
// which run as ONE launch of the fp32 MFMA kernel of conv_igemm.hip (1x1 path, grid.z = 16), so the
// matrix-core code is shared with the direct form.  This file holds what is around it:
//   wino_input_kernel    activation NHWC -> V [16][tiles][Cin]        (Bt d B; HBM-bound, 1 read : 4 writes)
//   wino_output_kernel   M' [16][tiles][Cout] -> NHWC output           (At m A + bias + LeakyReLU
//                        [+ MaxPooling2D(2,2): an output tile IS a pooling window]; or the ConvLSTM
//                        gate update of MultiObjDetTracker.py:176 in registers)
//   wino_pack_weights    host: U = G g Gt per (cin, cout), packed per position for the MFMA kernel
// fp32 throughout; the transforms only add/subtract (and halve, in G), so the result differs from
// the direct form by rounding only (measured: same error against float64 as the direct kernel).
#include "dt_internal.h"

#define WINO_THREADS 256

__device__ __forceinline__ f32x4 ld4(const float *p) { return *reinterpret_cast<const f32x4 *>(p); }
__device__ __forceinline__ void st4(float *p, f32x4 v) { *reinterpret_cast<f32x4 *>(p) = v; }

// ---- input transform ----------------------------------------------------------------------------
// One work item = (tile, 4 channels).  Tile (b, ty, tx) covers input rows 2ty-1 .. 2ty+2, cols
// 2tx-1 .. 2tx+2 ('same' padding and the odd last row/column of 13x13 read as zero).
__global__ __launch_bounds__(WINO_THREADS) void wino_input_kernel(WinoArgs p)
{
    const int cq_n = p.C >> 2;
    const long long items = (long long)p.Mt * cq_n;
    const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
    const long long plane = (long long)p.Mt * p.C;
    for (long long it = (long long)blockIdx.x * WINO_THREADS + threadIdx.x; it < items;
         it += (long long)gridDim.x * WINO_THREADS) {
        const int tile = (int)(it / cq_n);
        const int c = (int)(it - (long long)tile * cq_n) * 4;
        const int tpf = p.th * p.tw;
        const int b = tile / tpf;
        const int r = tile - b * tpf;
        const int ty = r / p.tw, tx = r - ty * p.tw;
        const float *src = p.in + (long long)b * p.in_bs + c;
        f32x4 d[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int h = 2 * ty - 1 + i;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int w = 2 * tx - 1 + j;
                const bool ok = h >= 0 && h < p.H && w >= 0 && w < p.W;
                d[i][j] = ok ? ld4(src + (long long)(h * p.W + w) * p.in_ld) : zero;
            }
        }
        // Bt d : rows
        f32x4 t[4][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            t[0][j] = d[0][j] - d[2][j];
            t[1][j] = d[1][j] + d[2][j];
            t[2][j] = d[2][j] - d[1][j];
            t[3][j] = d[1][j] - d[3][j];
        }
        float *dst = p.v + (long long)tile * p.C + c;
        // (Bt d) B : columns, stored plane by plane
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            st4(dst + (4 * i + 0) * plane, t[i][0] - t[i][2]);
            st4(dst + (4 * i + 1) * plane, t[i][1] + t[i][2]);
            st4(dst + (4 * i + 2) * plane, t[i][2] - t[i][1]);
            st4(dst + (4 * i + 3) * plane, t[i][1] - t[i][3]);
        }
    }
}

__device__ __forceinline__ float wino_leaky(float v, float slope) { return v > 0.0f ? v : v * slope; }
__device__ __forceinline__ float wino_hard_sigmoid(float x)
{
    const float y = __fmaf_rn(0.2f, x, 0.5f);   // Keras 2.x hard_sigmoid
    return fminf(fmaxf(y, 0.0f), 1.0f);
}

// At m A for four channels: m[16] planes -> y[2][2]
__device__ __forceinline__ void wino_at_m_a(const f32x4 *m, f32x4 y[2][2])
{
    f32x4 s0[4], s1[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        s0[j] = m[0 + j] + m[4 + j] + m[8 + j];
        s1[j] = m[4 + j] - m[8 + j] - m[12 + j];
    }
    y[0][0] = s0[0] + s0[1] + s0[2];
    y[0][1] = s0[1] - s0[2] - s0[3];
    y[1][0] = s1[0] + s1[1] + s1[2];
    y[1][1] = s1[1] - s1[2] - s1[3];
}

// ---- output transform, conv block epilogue -------------------------------------------------------
// One work item = (tile, 4 output channels): bias + LeakyReLU, optional full-resolution output and
// optional 2x2-pooled output (tiles start on even coordinates, so a tile is one pooling window).
__global__ __launch_bounds__(WINO_THREADS) void wino_output_kernel(WinoArgs p)
{
    const int nq = (p.N + 3) >> 2;
    const long long items = (long long)p.Mt * nq;
    const long long plane = (long long)p.Mt * p.m_ld;
    for (long long it = (long long)blockIdx.x * WINO_THREADS + threadIdx.x; it < items;
         it += (long long)gridDim.x * WINO_THREADS) {
        const int tile = (int)(it / nq);
        const int c = (int)(it - (long long)tile * nq) * 4;
        const int tpf = p.th * p.tw;
        const int b = tile / tpf;
        const int r = tile - b * tpf;
        const int ty = r / p.tw, tx = r - ty * p.tw;
        const float *src = p.m + (long long)tile * p.m_ld + c;
        f32x4 m[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) m[q] = ld4(src + q * plane);
        f32x4 y[2][2];
        wino_at_m_a(m, y);
        const f32x4 bv = p.bias ? ld4(p.bias + c) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        f32x4 mx;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                f32x4 v = y[i][j] + bv;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = wino_leaky(v[e], p.slope);
                y[i][j] = v;
                if (i == 0 && j == 0) mx = v;
                else
#pragma unroll
                    for (int e = 0; e < 4; ++e) mx[e] = fmaxf(mx[e], v[e]);
            }
        const int nvalid = min(4, p.N - c);
        if (p.out) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int h = 2 * ty + i, w = 2 * tx + j;
                    if (h < p.H && w < p.W) {
                        float *o = p.out + (long long)b * p.out_bs + (long long)(h * p.W + w) * p.out_ld + c;
                        if (nvalid == 4) st4(o, y[i][j]);
                        else
                            for (int e = 0; e < nvalid; ++e) o[e] = y[i][j][e];
                    }
                }
        }
        if (p.out2) {   // MaxPooling2D(2,2): H and W are even whenever the reference pools
            float *o = p.out2 + ((long long)(b * (p.H >> 1) + ty) * (p.W >> 1) + tx) * p.out2_ld + c;
            if (nvalid == 4) st4(o, mx);
            else
                for (int e = 0; e < nvalid; ++e) o[e] = mx[e];
        }
    }
}

// ---- output transform, ConvLSTM2D gate update ------------------------------------------------------
// N axis packed [j/32][gate][j%32] like EPI_GATES of conv_igemm.hip.  One work item = (tile, 4 hidden
// channels): the i,f,c,o pre-activations of the recurrent convolution come out of the transform in
// registers, the input projection (bias included) is added, c is updated in place and h written.
__global__ __launch_bounds__(WINO_THREADS) void wino_output_gates_kernel(WinoArgs p)
{
    const int U = p.N >> 2;
    const int uq = U >> 2;
    const long long items = (long long)p.Mt * uq;
    const long long plane = (long long)p.Mt * p.m_ld;
    for (long long it = (long long)blockIdx.x * WINO_THREADS + threadIdx.x; it < items;
         it += (long long)gridDim.x * WINO_THREADS) {
        const int tile = (int)(it / uq);
        const int jc = (int)(it - (long long)tile * uq) * 4;          // hidden channel
        const int col = (jc >> 5) * 128 + (jc & 31);                  // column of gate i; f,c,o at +32,+64,+96
        const int tpf = p.th * p.tw;
        const int b = tile / tpf;
        const int r = tile - b * tpf;
        const int ty = r / p.tw, tx = r - ty * p.tw;
        const float *src = p.m + (long long)tile * p.m_ld + col;
        f32x4 y[4][2][2];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 m[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) m[q] = ld4(src + q * plane + g * 32);
            wino_at_m_a(m, y[g]);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int h = 2 * ty + i, w = 2 * tx + j;
                if (h >= p.H || w >= p.W) continue;
                const long long pix = h * p.W + w;
                const float *xp = p.xproj + (long long)b * p.xp_bs + pix * p.xp_ld + col;
                const f32x4 zi = y[0][i][j] + ld4(xp), zf = y[1][i][j] + ld4(xp + 32);
                const f32x4 zc = y[2][i][j] + ld4(xp + 64), zo = y[3][i][j] + ld4(xp + 96);
                float *cp = p.cstate + (long long)b * p.c_bs + pix * p.c_ld + jc;
                const f32x4 cprev = ld4(cp);
                f32x4 cn, hn;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float gi = wino_hard_sigmoid(zi[e]), gf = wino_hard_sigmoid(zf[e]), go = wino_hard_sigmoid(zo[e]);
                    cn[e] = gf * cprev[e] + gi * tanhf(zc[e]);
                    hn[e] = go * tanhf(cn[e]);
                }
                st4(cp, cn);
                st4(p.out + (long long)b * p.out_bs + pix * p.out_ld + jc, hn);
            }
    }
}

static unsigned wino_blocks(long long items)
{
    long long nb = (items + WINO_THREADS - 1) / WINO_THREADS;
    const long long cap = 256 * 32;   // 32 workgroups of 256 threads per CU's worth of grid; grid-stride beyond
    if (nb > cap) nb = cap;
    return (unsigned)(nb < 1 ? 1 : nb);
}

int launch_wino_input(hipStream_t st, const WinoArgs &a)
{
    if (a.C % 4 || a.in_ld % 4 || a.Mt <= 0) return 2;
    hipLaunchKernelGGL(wino_input_kernel, dim3(wino_blocks((long long)a.Mt * (a.C / 4))), dim3(WINO_THREADS), 0, st, a);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

int launch_wino_output(hipStream_t st, const WinoArgs &a, int gates)
{
    if (a.m_ld % 4 || a.Mt <= 0) return 2;
    if (gates) {
        if (a.N % 128 || a.out_ld % 4 || a.c_ld % 4 || a.xp_ld % 4) return 2;
        hipLaunchKernelGGL(wino_output_gates_kernel, dim3(wino_blocks((long long)a.Mt * (a.N / 16))), dim3(WINO_THREADS),
                           0, st, a);
    } else {
        // vector stores need 16-byte aligned rows; ragged N (conv_23-like heads) never takes this path
        if ((a.out && a.out_ld % 4) || (a.out2 && a.out2_ld % 4)) return 2;
        hipLaunchKernelGGL(wino_output_kernel, dim3(wino_blocks((long long)a.Mt * ((a.N + 3) / 4))), dim3(WINO_THREADS),
                           0, st, a);
    }
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

// Host: U[p] = (G g Gt)[xi][nu] for every (cin, cout), p = 4*xi + nu, as 16 HWIO-shaped [1,1,Cin,Cout]
// kernels, each packed like a 1x1 layer of the MFMA kernel: dst [16][npad][cin_dst].
//   G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]
void wino_pack_weights(const float *hwio, int cin_src, int cout_src, const int *cin_map, int cin_dst, const int *n_map,
                       int npad, const float *scale, float *dst)
{
    static const double G[4][3] = {{1, 0, 0}, {.5, .5, .5}, {.5, -.5, .5}, {0, 0, 1}};
    const size_t plane = (size_t)cin_src * cout_src;
    std::vector<float> u(16 * plane);
    for (int ci = 0; ci < cin_src; ++ci)
        for (int co = 0; co < cout_src; ++co) {
            double g[3][3], t[4][3];
            const double sc = scale ? (double)scale[co] : 1.0;
            for (int k = 0; k < 9; ++k) g[k / 3][k % 3] = (double)hwio[((size_t)k * cin_src + ci) * cout_src + co] * sc;
            for (int xi = 0; xi < 4; ++xi)
                for (int kx = 0; kx < 3; ++kx)
                    t[xi][kx] = G[xi][0] * g[0][kx] + G[xi][1] * g[1][kx] + G[xi][2] * g[2][kx];
            for (int xi = 0; xi < 4; ++xi)
                for (int nu = 0; nu < 4; ++nu)
                    u[(size_t)(4 * xi + nu) * plane + (size_t)ci * cout_src + co] =
                        (float)(t[xi][0] * G[nu][0] + t[xi][1] * G[nu][1] + t[xi][2] * G[nu][2]);
        }
    for (int q = 0; q < 16; ++q)
        pack_conv_weights(u.data() + q * plane, 1, cin_src, cout_src, cin_map, cin_dst, n_map, npad, nullptr,
                          dst + (size_t)q * npad * cin_dst);
}

// wino4_fused.hip -- the early wide 3x3 layers (conv_3 / conv_5: 64 -> 128 channels at 104x104, conv_6 / conv_8:
// 128 -> 256 at 52x52; models_detection/KerasYOLO.py:291-320) as ONE fused Winograd F(4x4,3x3) kernel.
//
// Why: in the unfused form (winograd.hip + the batched GEMMs of conv_igemm.hip) these layers are HBM-bound -- with
// K = Cin = 64 / 128 the GEMMs stream V and M' (36/16 = 2.25x the activations each) at ~4 TB/s, and the two transform
// kernels move the same bytes again: conv_3 alone costs 5.4 ms (GEMMs) + 7.1 ms (transforms) per 1440 frames for
// 0.57 TFLOP of MFMA work.  Fused, V and M' never leave the CU and HBM sees the input once (+ halo) and the output once.
//
//   * a workgroup (8 waves) owns a block of 4x4 output tiles of 4x4 pixels (16x16 pixels of one frame) and 128
//     output channels; wave w computes the 16 tiles x 16 channels [16w, 16w+16) with v_mfma_f32_16x16x4_f32
//     (MFMA row = tile, column = output channel, k = input channel);
//   * the 18x18 input patch is staged through LDS 32 input channels at a time, CHANNEL-major ([32][385] floats, row
//     pitch 20): lane (tile r, k-slot kq) reads its 6x6 window of channel 4s + kq with 36 conflict-free ds_read_b32
//     (bank = 16 ty + 4 tx + kq + const), forms Bt d B in registers -- these 36 values ARE the A operands of the 36
//     Winograd positions for this K-step -- and issues 36 independent MFMAs, one per position, into 36 accumulators
//     (144 registers; the kernel runs at two waves per SIMD on the unified 512-entry VGPR/AGPR file);
//   * U_p = G g Gt streams from L2 as the B operand (one coalesced 256-byte load per position and K-step, layout
//     [n-half][channel group][k-step][wave][position][64]), fetched one K-step ahead;
//   * after the last channel group every lane holds M'[36] for its 4 (tile, channel) pairs: At M' A, bias, LeakyReLU
//     and (conv_5 / conv_8) the 2x2 max-pool finish in registers; 16 lanes write 64 contiguous bytes.
// MFMA work: 36/144 of the direct form.  fp32 throughout; F(4x4,3x3) rounds like winograd.hip's TS = 4 (1.5e-5 at
// activation scale 4, below the F(6x6) form these layers ran in before).
#include "dt_internal.h"

#define W4_PW 20                 // patch row pitch in pixels (18 used): 16*ty + 4*tx (+kq) covers all 64 LDS banks
#define W4_PH 18
#define W4_PLANE 385             // floats per channel plane (20*18 = 360, padded to 1 mod 64)
#define W4_GROUP 32              // input channels per LDS stage
#define W4_THREADS 512

__device__ __forceinline__ void w4_bt(float *d, int st)      // Bt (6x6) on d[0], d[st], ... d[5 st]
{
    const float d0 = d[0], d1 = d[st], d2 = d[2 * st], d3 = d[3 * st], d4 = d[4 * st], d5 = d[5 * st];
    d[0] = 4.0f * d0 - 5.0f * d2 + d4;
    d[st] = -4.0f * (d1 + d2) + d3 + d4;
    d[2 * st] = 4.0f * (d1 - d2) - d3 + d4;
    d[3 * st] = 2.0f * (d3 - d1) - d2 + d4;
    d[4 * st] = 2.0f * (d1 - d3) - d2 + d4;
    d[5 * st] = 4.0f * d1 - 5.0f * d3 + d5;
}

__device__ __forceinline__ void w4_at(float *m, int st)      // At (4x6): 6 inputs -> 4 outputs in the first 4 slots
{
    const float a = m[st] + m[2 * st], b = m[st] - m[2 * st], c = m[3 * st] + m[4 * st], e = m[3 * st] - m[4 * st];
    const float y0 = m[0] + a + c, y1 = b + 2.0f * e, y2 = a + 4.0f * c, y3 = b + 8.0f * e + m[5 * st];
    m[0] = y0; m[st] = y1; m[2 * st] = y2; m[3 * st] = y3;
}

template <bool POOL>
__global__ __launch_bounds__(W4_THREADS) void wino4_fused_kernel(Wino4FusedArgs p)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];     // [2][W4_GROUP][W4_PLANE]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, kq = lane >> 4;
    const int ty = r >> 2, tx = r & 3;
    int bid = blockIdx.x;
    const int bx = bid % p.nbx;
    bid /= p.nbx;
    const int by = bid % p.nby, b = bid / p.nby;
    const int nh = blockIdx.y;
    const int ngroups = p.Cin / W4_GROUP;
    const int h0 = by * 16 - 1, w0 = bx * 16 - 1;
    const float *img = p.in + (long long)b * p.in_bs;

    // ---- staging: thread -> (pixel, 4-channel quad) of the 18x18x32 patch; quad k of a thread is element
    // tid + 512 k of the 2592.  The NEXT group's quads are fetched one per K-step inside the compute loop (4 live
    // registers instead of 24) and written to the other LDS buffer a K-step later. ----
    constexpr int NQ = (W4_PH * 18 * 8 + W4_THREADS - 1) / W4_THREADS;      // 6
    auto quad_load = [&](int g, int k) -> f32x4 {
        const int idx = tid + k * W4_THREADS;
        const int q = idx & 7, pix = idx >> 3;
        const int pr = pix / 18, pc = pix - pr * 18;
        const int h = h0 + pr, w = w0 + pc;
        const bool ok = idx < W4_PH * 18 * 8 && h >= 0 && h < p.H && w >= 0 && w < p.W;
        f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
        if (ok) v = *reinterpret_cast<const f32x4 *>(img + ((long long)h * p.W + w) * p.in_ld + g * W4_GROUP + q * 4);
        return v;
    };
    auto quad_store = [&](int buf, int k, const f32x4 &v) {
        const int idx = tid + k * W4_THREADS;
        if (idx < W4_PH * 18 * 8) {
            const int q = idx & 7, pix = idx >> 3;
            const int pr = pix / 18, pc = pix - pr * 18;
            float *d = lds + buf * (W4_GROUP * W4_PLANE) + (q * 4) * W4_PLANE + pr * W4_PW + pc;
            d[0] = v[0]; d[W4_PLANE] = v[1]; d[2 * W4_PLANE] = v[2]; d[3 * W4_PLANE] = v[3];
        }
    };

    f32x4 acc[36];
#pragma unroll
    for (int q = 0; q < 36; ++q) acc[q] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

    // B operand stream of this wave: [nh][g][s][wave][pos][64]
    const float *ub = p.u + ((long long)nh * ngroups * 8 * 8 + wave) * (36 * 64) + lane;
    const long long u_s = 8ll * 36 * 64;        // k-step stride

#pragma unroll 1
    for (int k = 0; k < NQ; ++k) quad_store(0, k, quad_load(0, k));
    __syncthreads();
    float bq[36];
#pragma unroll
    for (int q = 0; q < 36; ++q) bq[q] = ub[q * 64];
    const int nsteps = ngroups * 8;
#pragma unroll 1
    for (int gs = 0; gs < nsteps; ++gs) {
        const int g = gs >> 3, s = gs & 7;
        const bool more = g + 1 < ngroups;
        f32x4 sq = {0.0f, 0.0f, 0.0f, 0.0f};
        if (more && s < NQ) sq = quad_load(g + 1, s);          // in flight under this K-step's MFMAs
        const float *pl = lds + (g & 1) * (W4_GROUP * W4_PLANE) + (4 * s + kq) * W4_PLANE + (4 * ty) * W4_PW + 4 * tx;
        float v[36];
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = 0; j < 6; ++j) v[i * 6 + j] = pl[i * W4_PW + j];
#pragma unroll
        for (int j = 0; j < 6; ++j) w4_bt(v + j, 6);          // columns: over the row index i  (xi)
        const float *un = ub + (long long)(gs + 1 < nsteps ? gs + 1 : gs) * u_s;
#pragma unroll
        for (int xi = 0; xi < 6; ++xi) {
            w4_bt(v + 6 * xi, 1);                             // row xi: over the column index j  (nu)
#pragma unroll
            for (int nu = 0; nu < 6; ++nu) {
                const int q = 6 * xi + nu;
                acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[q], bq[q], acc[q], 0, 0, 0);
                bq[q] = un[q * 64];                           // the next K-step's B operand, a whole K-step ahead
            }
        }
        if (more && s < NQ) quad_store((g + 1) & 1, s, sq);
        if (s == 7) __syncthreads();
    }

    // ---- output transform + bias + LeakyReLU (+ 2x2 max): C/D row = 4*kq + e (tile), col = lane & 15 (channel) ----
    const int n = nh * 128 + wave * 16 + r;
    const float bv = p.bias[n];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float m[36];
#pragma unroll
        for (int q = 0; q < 36; ++q) m[q] = acc[q][e];
#pragma unroll
        for (int j = 0; j < 6; ++j) w4_at(m + j, 6);          // over xi -> rows a = 0..3
#pragma unroll
        for (int a = 0; a < 4; ++a) w4_at(m + 6 * a, 1);      // over nu -> cols c = 0..3
        const int rt = 4 * kq + e;
        const int oy = by * 16 + 4 * (rt >> 2), ox = bx * 16 + 4 * (rt & 3);
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float y = m[6 * a + c] + bv;
                m[6 * a + c] = y > 0.0f ? y : y * p.slope;
            }
        if (!POOL) {
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (oy + a < p.H && ox + c < p.W)
                        p.out[(long long)b * p.out_bs + ((long long)(oy + a) * p.W + ox + c) * p.out_ld + n] = m[6 * a + c];
        } else {
            const int H2 = p.H >> 1, W2 = p.W >> 1;
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const float mx = fmaxf(fmaxf(m[6 * (2 * a) + 2 * c], m[6 * (2 * a) + 2 * c + 1]),
                                           fmaxf(m[6 * (2 * a + 1) + 2 * c], m[6 * (2 * a + 1) + 2 * c + 1]));
                    const int py = (oy >> 1) + a, px = (ox >> 1) + c;
                    if (py < H2 && px < W2)
                        p.out2[(((long long)b * H2 + py) * W2 + px) * p.out2_ld + n] = mx;
                }
        }
    }
}

int launch_wino4_fused(hipStream_t st, const Wino4FusedArgs &a_in)
{
    Wino4FusedArgs a = a_in;
    if (a.B <= 0 || a.Cin % W4_GROUP || a.N % 128 || a.in_ld % 4) return 2;
    const bool pool = a.out2 != nullptr;
    if (pool && ((a.H | a.W) & 1)) return 2;
    if (pool == (a.out != nullptr)) return 2;       // exactly one of the two outputs
    a.nby = (a.H + 15) / 16;
    a.nbx = (a.W + 15) / 16;
    const long long blocks = (long long)a.B * a.nby * a.nbx;
    if (blocks >= (1ll << 31)) return 2;
    const size_t lds = (size_t)2 * W4_GROUP * W4_PLANE * sizeof(float);
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(wino4_fused_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void *>(wino4_fused_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return 1;
        attr_done = true;
    }
    const dim3 grid((unsigned)blocks, (unsigned)(a.N / 128));
    if (pool) hipLaunchKernelGGL(wino4_fused_kernel<true>, grid, dim3(W4_THREADS), lds, st, a);
    else hipLaunchKernelGGL(wino4_fused_kernel<false>, grid, dim3(W4_THREADS), lds, st, a);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

// Host: u36 = wino_pack_weights(4, ...) output [36][npad][cin] (U_p[n][c]) -> the kernel's B-operand stream
//   dst[nh][g][s][wave 8][pos 36][kq 4][16]:  element = U_pos[c = 32 g + 4 s + kq][n = 128 nh + 16 wave + j]
void wino4_fused_pack(const float *u36, int npad, int cin, int cout, float *dst)
{
    const int ngroups = cin / W4_GROUP, nhalf = cout / 128;
    const size_t plane = (size_t)npad * cin;
    for (int nh = 0; nh < nhalf; ++nh)
        for (int g = 0; g < ngroups; ++g)
            for (int s = 0; s < 8; ++s)
                for (int w = 0; w < 8; ++w)
                    for (int pos = 0; pos < 36; ++pos)
                        for (int kq = 0; kq < 4; ++kq)
                            for (int j = 0; j < 16; ++j) {
                                const int c = g * W4_GROUP + 4 * s + kq, n = nh * 128 + w * 16 + j;
                                dst[((((((size_t)nh * ngroups + g) * 8 + s) * 8 + w) * 36 + pos) * 4 + kq) * 16 + j] =
                                    u36[(size_t)pos * plane + (size_t)n * cin + c];
                            }
}

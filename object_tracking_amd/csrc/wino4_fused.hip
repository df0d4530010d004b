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
//     (MFMA row = tile, column = output channel, k = input channel), 36 accumulators (one per Winograd position);
//   * input channels stream through LDS 16 at a time ("half-group"), double-buffered twice over:
//       patch  [2][16 ch][385]      the 18x18 input window, CHANNEL-major, row pitch 20 (bank = ch + 16 ty + 4 tx: the
//                                   6x6 window reads of 64 (tile, channel) lanes are conflict-free)
//       V      [2][36 pos][16 ch][16 tiles]   Bt d B, computed ONCE per workgroup (waves 0-3: one (tile, channel) pair
//                                   per lane) -- the eight waves would otherwise each redo the same transform;
//     the A operand of position q, K-step s is then one ds_read_b32 (64 consecutive floats per wave);
//   * U_p = G g Gt streams from L2 as the B operand (one coalesced 256-byte load per position and K-step, layout
//     [n-half][k-step][wave][position][64]); all 36 loads of a K-step are issued at the top of the previous one;
//   * per half-group: fetch patch h+2 (registers) -> transform patch h+1 into V -> 4 K-steps x 36 MFMAs on V(h) ->
//     store patch h+2 -> one barrier;
//   * after the last half-group every lane holds M'[36] for its 4 (tile, channel) pairs: At M' A, bias, LeakyReLU
//     and (conv_5 / conv_8) the 2x2 max-pool finish in registers; 16 lanes write 64 contiguous bytes.
// MFMA work: 36/144 of the direct form.  fp32 throughout; F(4x4,3x3) rounds like winograd.hip's TS = 4 (1.5e-5 at
// activation scale 4, below the F(6x6) form these layers ran in before).
#include "dt_internal.h"

#ifndef DT_W4_ABLATE
#define DT_W4_ABLATE 0      // timing-only ablation builds (tools/ablate_w4.sh): 1 no B loads, 2 no A reads, 4 no transform, 8 no staging, 16 no stores
#endif

#ifdef DT_W4_TIMING
// debug build only (tools/w4_timing.py): per-workgroup timestamps of the persistent step loop, waves 0 and 4
#define W4_TT_WG 256
#define W4_TT_STEPS 48
__device__ unsigned long long g_w4_times[W4_TT_WG * 2 * W4_TT_STEPS * 6];
extern "C" __attribute__((visibility("default"))) int dt_debug_w4_times(unsigned long long *dst, int clear)
{
    if (clear) {
        void *p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_w4_times)) != hipSuccess) return 1;
        return hipMemset(p, 0, sizeof(g_w4_times)) == hipSuccess ? 0 : 1;
    }
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_w4_times), sizeof(g_w4_times)) == hipSuccess ? 0 : 1;
}
#define W4_STAMP(k)                                                                                                        \
    do {                                                                                                                   \
        if (lane == 0 && (wave == 0 || wave == 4) && blockIdx.y == 0 && blockIdx.x < W4_TT_WG && tt_i < W4_TT_STEPS)        \
            g_w4_times[((blockIdx.x * 2 + (wave >> 2)) * W4_TT_STEPS + tt_i) * 6 + (k)] = __builtin_readcyclecounter();    \
    } while (0)
#else
#define W4_STAMP(k) do { } while (0)
#endif

#define W4_PW 20                 // patch row pitch in pixels (18 used): 16*ty + 4*tx (+ch) covers all 64 LDS banks
#define W4_PH 18
#define W4_PLANE 385             // floats per channel plane (20*18 = 360, padded to 1 mod 64)
#define W4_HG 16                 // input channels per LDS stage (half-group): 4 K-steps
#define W4_PBUF (W4_HG * W4_PLANE)
#define W4_VBUF (36 * W4_HG * 16)
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
    extern __shared__ __attribute__((aligned(16))) float lds[];     // patch [2][W4_PBUF] | V [2][W4_VBUF]
    float *const Pb = lds;
    float *const Vb = lds + 2 * W4_PBUF;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, kq = lane >> 4;
    const int nh = blockIdx.y;
    const int nhg = p.Cin / W4_HG;
    const int nblk = p.B * p.nby * p.nbx;

    // The workgroup is PERSISTENT: it walks blocks j = blockIdx.x, + gridDim.x, ... and treats their half-groups as one
    // stream of steps (j, h).  While step t's MFMAs run, step t+1's patch is transformed and step t+2's patch is fetched
    // -- across block boundaries too, so a block's prologue hides under its predecessor and its output stores drain
    // under its successor.
    struct Step { int j, h; };
    auto advance = [&](Step &t) { if (++t.h == nhg) { t.h = 0; t.j += gridDim.x; } };

    // ---- patch staging: item = (pixel, 4-channel quad) of the 18x18x16 window; thread takes items tid + 512 k ----
    constexpr int NI = W4_PH * 18 * 4;                                     // 1296
    constexpr int NQ = (NI + W4_THREADS - 1) / W4_THREADS;                 // 3
    auto quad_load = [&](const Step &t, int k) -> f32x4 {
        const int bx = t.j % p.nbx, by = (t.j / p.nbx) % p.nby, b = t.j / (p.nbx * p.nby);
        const int idx = tid + k * W4_THREADS;
        const int q = idx & 3, pix = idx >> 2;
        const int pr = pix / 18, pc = pix - pr * 18;
        const int hh = by * 16 - 1 + pr, ww = bx * 16 - 1 + pc;
        const bool ok = idx < NI && hh >= 0 && hh < p.H && ww >= 0 && ww < p.W;
        f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
        if (ok) v = *reinterpret_cast<const f32x4 *>(p.in + (long long)b * p.in_bs + ((long long)hh * p.W + ww) * p.in_ld + t.h * W4_HG + q * 4);
        return v;
    };
    auto quad_store = [&](int buf, int k, const f32x4 &v) {
        const int idx = tid + k * W4_THREADS;
        if (idx < NI) {
            const int q = idx & 3, pix = idx >> 2;
            const int pr = pix / 18, pc = pix - pr * 18;
            float *d = Pb + buf * W4_PBUF + (q * 4) * W4_PLANE + pr * W4_PW + pc;
            d[0] = v[0]; d[W4_PLANE] = v[1]; d[2 * W4_PLANE] = v[2]; d[3 * W4_PLANE] = v[3];
        }
    };
    // ---- input transform of one half-group: waves 0-3, lane -> (tile = lane & 15, channel = 4 wave + kq) ----
    auto transform = [&](int pbuf, int vbuf) {
        if (wave < 4) {
            const int ty = r >> 2, tx = r & 3, ch = 4 * wave + kq;
            const float *pl = Pb + pbuf * W4_PBUF + ch * W4_PLANE + (4 * ty) * W4_PW + 4 * tx;
            float v[36];
#pragma unroll
            for (int i = 0; i < 6; ++i)
#pragma unroll
                for (int j = 0; j < 6; ++j) v[i * 6 + j] = pl[i * W4_PW + j];
#pragma unroll
            for (int j = 0; j < 6; ++j) w4_bt(v + j, 6);          // over the row index i  -> xi
#pragma unroll
            for (int i = 0; i < 6; ++i) w4_bt(v + 6 * i, 1);      // over the column index j -> nu
            float *o = Vb + vbuf * W4_VBUF + ch * 16 + r;
#pragma unroll
            for (int q = 0; q < 36; ++q) o[q * (W4_HG * 16)] = v[q];
        }
    };

    f32x4 acc[36];

    // B operand stream of this wave: [nh][k-step][wave][pos][64]; it wraps around at a block boundary
    const int nsteps = nhg * 4;
    const float *ub = p.u + ((long long)nh * nsteps * 8 + wave) * (36 * 64) + lane;
    const long long u_s = 8ll * 36 * 64;        // k-step stride
    const int nl = wave * 16 + r;               // output channel within the workgroup's 128
    const float bv = p.bias[nh * 128 + nl];

    Step t0{(int)blockIdx.x, 0};
    Step t1 = t0; advance(t1);
    Step t2 = t1; advance(t2);
    if (t0.j >= nblk) return;
#pragma unroll 1
    for (int k = 0; k < NQ; ++k) quad_store(0, k, quad_load(t0, k));
    if (t1.j < nblk) {
#pragma unroll 1
        for (int k = 0; k < NQ; ++k) quad_store(1, k, quad_load(t1, k));
    }
    float bcur[36];
#pragma unroll
    for (int q = 0; q < 36; ++q) bcur[q] = ub[q * 64];
    __syncthreads();
    transform(0, 0);
    __syncthreads();

    // one K-step = 36 MFMAs, one per Winograd position.  Operand traffic is software-pipelined by hand (the compiler,
    // short of registers, would funnel every A operand through one register pair and expose the LDS latency 18 times
    // per K-step):
    //   B: ONE register per position; b[q] is reloaded for the next K-step right after MFMA q issues -- a full
    //      K-step (36 MFMA slots ~ 1150 cycles) ahead of its use;
    //   A: two sets of 9 registers; batch j+1 (positions 9j+9 .. 9j+17) is fetched from LDS under the MFMAs of
    //      batch j, the next K-step's first batch under the last one.
    float a0[9], a1[9];
    auto load_a = [&](float (&dst)[9], const float *vs, int q0) {
#pragma unroll
        for (int i = 0; i < 9; ++i) dst[i] = (DT_W4_ABLATE & 2) ? 1.0f : vs[(q0 + i) * (W4_HG * 16)];
    };
    auto kstep = [&](int gs, int s, const float *va, bool stage, int pbuf) {
        f32x4 sq = {0.0f, 0.0f, 0.0f, 0.0f};
        if (stage) sq = quad_load(t2, s);                          // one quad of the patch two steps ahead
        const float *un = ub + (long long)(gs + 1 < nsteps ? gs + 1 : 0) * u_s;
        const float *vs = va + (4 * s) * 16;
        auto batch = [&](int q0, float (&au)[9]) {
#pragma unroll
            for (int i = 0; i < 9; ++i) {
                acc[q0 + i] = __builtin_amdgcn_mfma_f32_16x16x4f32(au[i], bcur[q0 + i], acc[q0 + i], 0, 0, 0);
                if (!(DT_W4_ABLATE & 1)) bcur[q0 + i] = un[(q0 + i) * 64];
            }
        };
        load_a(a1, vs, 9);
        batch(0, a0);
        load_a(a0, vs, 18);
        batch(9, a1);
        load_a(a1, vs, 27);
        batch(18, a0);
        if (s < 3) load_a(a0, vs + 4 * 16, 0);
        batch(27, a1);
        if (stage) quad_store(pbuf, s, sq);
    };

    int cur = 0;                                                       // LDS buffer (patch and V) of the current step
#ifdef DT_W4_TIMING
    int tt_i = 0;
#endif
#pragma unroll 1
    while (t0.j < nblk) {
        W4_STAMP(0);
        if (t0.h == 0) {
#pragma unroll
            for (int q = 0; q < 36; ++q) acc[q] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        }
        if (t1.j < nblk && !(DT_W4_ABLATE & 4)) transform(cur ^ 1, cur ^ 1);
        W4_STAMP(1);
        const bool stage = t2.j < nblk && !(DT_W4_ABLATE & 8);
        const float *va = Vb + cur * W4_VBUF + kq * 16 + r;
        load_a(a0, va, 0);
        kstep(t0.h * 4 + 0, 0, va, stage, cur);
        kstep(t0.h * 4 + 1, 1, va, stage, cur);
        kstep(t0.h * 4 + 2, 2, va, stage, cur);
        kstep(t0.h * 4 + 3, 3, va, false, cur);
        W4_STAMP(2);

        if (t0.h == nhg - 1) {
            // ---- output transform + bias + LeakyReLU (+ 2x2 max): C/D row = 4*kq + e (tile: row kq, column e of the
            // block), col = lane & 15 (channel); the stores drain under the next block's MFMAs ----
            const int bx = t0.j % p.nbx, by = (t0.j / p.nbx) % p.nby, b = t0.j / (p.nbx * p.nby);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float m[36];
#pragma unroll
                for (int q = 0; q < 36; ++q) m[q] = acc[q][e];
#pragma unroll
                for (int j = 0; j < 6; ++j) w4_at(m + j, 6);          // over xi -> rows a = 0..3
#pragma unroll
                for (int a = 0; a < 4; ++a) w4_at(m + 6 * a, 1);      // over nu -> cols c = 0..3
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const float v = m[6 * a + c] + bv;
                        m[6 * a + c] = v > 0.0f ? v : v * p.slope;
                    }
                const int oy = by * 16 + 4 * kq, ox = bx * 16 + 4 * e;
                if (DT_W4_ABLATE & 16) {
                    if (m[0] == 123.456f) p.out[0] = m[7];
                } else if (!POOL) {
#pragma unroll
                    for (int a = 0; a < 4; ++a)
#pragma unroll
                        for (int c = 0; c < 4; ++c)
                            if (oy + a < p.H && ox + c < p.W)
                                p.out[(long long)b * p.out_bs + ((long long)(oy + a) * p.W + ox + c) * p.out_ld + nh * 128 + nl] = m[6 * a + c];
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
                                p.out2[(((long long)b * H2 + py) * W2 + px) * p.out2_ld + nh * 128 + nl] = mx;
                        }
                }
            }
        }
        W4_STAMP(3);
        __syncthreads();
        W4_STAMP(4);
#ifdef DT_W4_TIMING
        ++tt_i;
#endif
        t0 = t1; t1 = t2; advance(t2);
        cur ^= 1;
    }
}

int launch_wino4_fused(hipStream_t st, const Wino4FusedArgs &a_in)
{
    Wino4FusedArgs a = a_in;
    if (a.B <= 0 || a.Cin % W4_HG || a.N % 128 || a.in_ld % 4) return 2;
    if ((a.out && a.out_ld % 4) || (a.out2 && a.out2_ld % 4)) return 2;       // float4 pixel lines
    const bool pool = a.out2 != nullptr;
    if (pool && ((a.H | a.W) & 1)) return 2;
    if (pool == (a.out != nullptr)) return 2;       // exactly one of the two outputs
    a.nby = (a.H + 15) / 16;
    a.nbx = (a.W + 15) / 16;
    const long long blocks = (long long)a.B * a.nby * a.nbx;
    if (blocks >= (1ll << 31)) return 2;
    const size_t lds = (size_t)2 * (W4_PBUF + W4_VBUF) * sizeof(float);      // 123,008 B
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(wino4_fused_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void *>(wino4_fused_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return 1;
        attr_done = true;
    }
    // persistent: one workgroup per CU and output-channel half (8 waves x 256 registers fill a CU's register file)
    static const int cus = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        return n;
    }();
    const int halves = a.N / 128;
    long long gx = cus / halves;
    if (gx < 1) gx = 1;
    if (gx > blocks) gx = blocks;
    const dim3 grid((unsigned)gx, (unsigned)halves);
    if (pool) hipLaunchKernelGGL(wino4_fused_kernel<true>, grid, dim3(W4_THREADS), lds, st, a);
    else hipLaunchKernelGGL(wino4_fused_kernel<false>, grid, dim3(W4_THREADS), lds, st, a);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

// Host: u36 = wino_pack_weights(4, ...) output [36][npad][cin] (U_p[n][c]) -> the kernel's B-operand stream
//   dst[nh][k-step][wave 8][pos 36][kq 4][16]:  element = U_pos[c = 4 kstep + kq][n = 128 nh + 16 wave + j]
void wino4_fused_pack(const float *u36, int npad, int cin, int cout, float *dst)
{
    const int nsteps = cin / 4, nhalf = cout / 128;
    const size_t plane = (size_t)npad * cin;
    for (int nh = 0; nh < nhalf; ++nh)
        for (int gs = 0; gs < nsteps; ++gs)
            for (int w = 0; w < 8; ++w)
                for (int pos = 0; pos < 36; ++pos)
                    for (int kq = 0; kq < 4; ++kq)
                        for (int j = 0; j < 16; ++j) {
                            const int c = 4 * gs + kq, n = nh * 128 + w * 16 + j;
                            dst[(((((size_t)nh * nsteps + gs) * 8 + w) * 36 + pos) * 4 + kq) * 16 + j] =
                                u36[(size_t)pos * plane + (size_t)n * cin + c];
                        }
}

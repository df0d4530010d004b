// wino4_fused.hip -- the early wide 3x3 layers (conv_3 / conv_5: 64 -> 128 channels at 104x104, conv_6 / conv_8:
// 128 -> 256 at 52x52; models_detection/KerasYOLO.py:291-320) as ONE fused Winograd F(4x4,3x3) kernel.
//
// Why: in the unfused form (winograd.hip + the batched GEMMs of conv_igemm.hip) these layers are HBM-bound -- with
// K = Cin = 64 / 128 the GEMMs stream V and M' (36/16 = 2.25x the activations each) at ~4 TB/s, and the two transform
// kernels move the same bytes again: conv_3 alone costs 5.4 ms (GEMMs) + 7.1 ms (transforms) per 1440 frames for
// 0.57 TFLOP of MFMA work.  Fused, V and M' never leave the CU and HBM sees the input once (+ halo) and the output once.
//
// What bounds a fused form is the B operand: every element of U = G g Gt (36 x Cin x Cout floats: 1.2 MB for 64 -> 128)
// is needed once per block of tiles, and a CU's vector-memory path delivers only ~17 B/clk of streamed dword loads
// (tools/micro/mfma16_probe.hip: one B load per v_mfma_f32_16x16x4_f32 caps the matrix pipe at 55 %, one per two at 70 %).
// So a workgroup must spread each B fetch over as many tiles as its LDS can hold V for -- 32:
//
//   * a workgroup (8 waves) owns TWO blocks of 4x4 output tiles of 4x4 pixels (2 x 16x16 pixels) and 64 output channels;
//     the 36 Winograd positions are split between two wave sets: wave (ph, wn) computes positions xi in {3 ph .. 3 ph + 2}
//     (18 of 36) for BOTH blocks and the 16 channels [16 wn, 16 wn + 16) with v_mfma_f32_16x16x4_f32 (row = tile,
//     column = channel, k = input channel): 36 accumulators, each B register feeds two MFMAs;
//   * input channels stream through LDS 8 at a time (a stage = 2 K-steps), double-buffered:
//       patch  [2][2 blocks][8 ch][385]   the 18x18 input windows, CHANNEL-major, row pitch 20 (bank = ch + 16 ty + 4 tx:
//                                         the 6x6 window reads of 64 (tile, channel) lanes are conflict-free)
//       V      [2][36 pos][2 blocks][8 ch][16 tiles]   Bt d B, computed ONCE per workgroup (waves 0-3: one (tile,
//                                         channel) pair per lane); an A operand pair is one ds_read2_b32;
//   * U streams from L2 as the B operand (one coalesced 256-byte load per position and K-step, layout
//     [n-quarter][k-step][ph][wn][18 positions][64]); b[i] is reloaded right after its two MFMAs, a K-step ahead;
//   * per stage: fetch patch s+2 (registers) -> transform patch s+1 into V -> 2 K-steps x 36 MFMAs on V(s) ->
//     store patch s+2 -> one barrier;
//   * epilogue: the output transform At M' A is linear in the positions, so each wave set forms its PARTIAL 4x4 output
//     block (its three xi rows through the column pass, then the full row pass); the two partial results meet in LDS
//     (ph 1 -> ph 0 for block 0, ph 0 -> ph 1 for block 1), then bias, LeakyReLU, (conv_5 / conv_8) 2x2 max-pool, store.
// MFMA work: 36/144 of the direct form.  fp32 throughout; F(4x4,3x3) rounds like winograd.hip's TS = 4 (1.5e-5 at
// activation scale 4, below the F(6x6) form these layers ran in before).
#include <cstdlib>

#include "dt_internal.h"

#ifndef DT_W4_ABLATE
#define DT_W4_ABLATE 0      // timing-only ablation builds (tools/ablate_w4.sh): 1 no B loads, 2 no A reads, 4 no transform, 8 no staging, 16 no stores
#endif

#ifdef DT_W4_TIMING
// debug build only (tools/w4_timing.py): per-workgroup timestamps of the persistent step loop, waves 0 and 4
#define W4_TT_WG 256
#define W4_TT_STEPS 24
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
        if (lane == 0 && (wave == 0 || wave == 4) && blockIdx.y == 0 && blockIdx.x >= 2048 && blockIdx.x < 2048 + W4_TT_WG && tt_i < W4_TT_STEPS) \
            g_w4_times[(((blockIdx.x - 2048) * 2 + (wave >> 2)) * W4_TT_STEPS + tt_i) * 6 + (k)] = __builtin_readcyclecounter();    \
    } while (0)
#else
#define W4_STAMP(k) do { } while (0)
#endif

#define W4_PW 20                 // patch row pitch in pixels (18 used): 16*ty + 4*tx (+ch) covers all 64 LDS banks
#define W4_PH 18
#define W4_PLANE 385             // floats per channel plane (20*18 = 360, padded to 1 mod 64)
#define W4_SC 8                  // input channels per LDS stage: 2 K-steps
#define W4_PBUF (2 * W4_SC * W4_PLANE)      // two blocks
#define W4_VBUF (36 * 2 * W4_SC * 16)
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

// partial column pass of At over the three xi rows a wave set owns: m[3][6] (xi local, nu) -> t[4][6] (a, nu)
__device__ __forceinline__ void w4_at_partial(const float *m, int ph, float *t)
{
#pragma unroll
    for (int nu = 0; nu < 6; ++nu) {
        const float m0 = m[nu], m1 = m[6 + nu], m2 = m[12 + nu];
        if (ph == 0) {           // xi = 0, 1, 2:  At columns (1,0,0,0), (1,1,1,1), (1,-1,1,-1)
            t[nu] = m0 + m1 + m2; t[6 + nu] = m1 - m2; t[12 + nu] = m1 + m2; t[18 + nu] = m1 - m2;
        } else {                 // xi = 3, 4, 5:  At columns (1,2,4,8), (1,-2,4,-8), (0,0,0,1)
            const float a = m0 + m1, b = m0 - m1;
            t[nu] = a; t[6 + nu] = 2.0f * b; t[12 + nu] = 4.0f * a; t[18 + nu] = 8.0f * b + m2;
        }
    }
}

template <bool POOL>
__global__ __launch_bounds__(W4_THREADS) void wino4_fused_kernel(Wino4FusedArgs p)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];     // patch [2][W4_PBUF] | V [2][W4_VBUF]
    float *const Pb = lds;
    float *const Vb = lds + 2 * W4_PBUF;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, kq = lane >> 4;
    const int ph = wave >> 2, wn = wave & 3;
    const int nq = blockIdx.y;
    const int nst = p.Cin / W4_SC;                 // stages of 8 input channels
    const int nblk = p.B * p.nby * p.nbx;
    const int j0 = 2 * blockIdx.x;                 // this workgroup's two blocks: j0, j0 + 1 (the second may not exist)

#ifdef DT_W4_TIMING
    int tt_i = 0;
#endif
    W4_STAMP(0);
    // ---- patch staging: item = (block, pixel, 4-channel quad) of the 2 x 18x18x8 windows; thread takes items tid + 512 k ----
    constexpr int NI = 2 * W4_PH * 18 * 2;                                 // 1296
    constexpr int NQ = (NI + W4_THREADS - 1) / W4_THREADS;                 // 3
    auto quad_load = [&](int st, int k) -> f32x4 {
        const int idx = tid + k * W4_THREADS;
        const int blk = idx >= NI / 2, rem = idx - blk * (NI / 2);
        const int q = rem & 1, pix = rem >> 1;
        const int pr = pix / 18, pc = pix - pr * 18;
        const int j = j0 + blk;
        const int bx = j % p.nbx, by = (j / p.nbx) % p.nby, b = j / (p.nbx * p.nby);
        const int hh = by * 16 - 1 + pr, ww = bx * 16 - 1 + pc;
        const bool ok = idx < NI && j < nblk && hh >= 0 && hh < p.H && ww >= 0 && ww < p.W;
        f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
        if (ok) v = *reinterpret_cast<const f32x4 *>(p.in + (long long)b * p.in_bs + ((long long)hh * p.W + ww) * p.in_ld + st * W4_SC + q * 4);
        return v;
    };
    auto quad_store = [&](int buf, int k, const f32x4 &v) {
        const int idx = tid + k * W4_THREADS;
        if (idx < NI) {
            const int blk = idx >= NI / 2, rem = idx - blk * (NI / 2);
            const int q = rem & 1, pix = rem >> 1;
            const int pr = pix / 18, pc = pix - pr * 18;
            float *d = Pb + buf * W4_PBUF + (blk * W4_SC + q * 4) * W4_PLANE + pr * W4_PW + pc;
            d[0] = v[0]; d[W4_PLANE] = v[1]; d[2 * W4_PLANE] = v[2]; d[3 * W4_PLANE] = v[3];
        }
    };
    // ---- input transform of one stage: waves 0-3, wave w -> block w & 1, channels 4 (w >> 1) + kq, tile = lane & 15 ----
    auto transform = [&](int pbuf, int vbuf) {
        if (wave < 4) {
            const int ty = r >> 2, tx = r & 3, blk = wave & 1, ch = 4 * (wave >> 1) + kq;
            const float *pl = Pb + pbuf * W4_PBUF + (blk * W4_SC + ch) * W4_PLANE + (4 * ty) * W4_PW + 4 * tx;
            float v[36];
#pragma unroll
            for (int i = 0; i < 6; ++i)
#pragma unroll
                for (int j = 0; j < 6; ++j) v[i * 6 + j] = pl[i * W4_PW + j];
#pragma unroll
            for (int j = 0; j < 6; ++j) w4_bt(v + j, 6);          // over the row index i  -> xi
#pragma unroll
            for (int i = 0; i < 6; ++i) w4_bt(v + 6 * i, 1);      // over the column index j -> nu
            float *o = Vb + vbuf * W4_VBUF + (blk * W4_SC + ch) * 16 + r;
#pragma unroll
            for (int q = 0; q < 36; ++q) o[q * 256] = v[q];
        }
    };

    f32x4 acc[2][18];
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int i = 0; i < 18; ++i) acc[g][i] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

    // B operand stream of this wave: [nq][k-step][ph][wn][1152]; the 1152 floats of one K-step are the lane's operands of
    // positions 0..15 as four float4 ([quad][lane][4]) followed by positions 16, 17 as one float2 ([lane][2]): 16-byte
    // loads deliver the stream at a higher rate than dword loads do (tools/micro/mfma16_probe.hip: 127 vs 111 TFLOP/s)
    const int nsteps = nst * 2;
    const float *ub = p.u + (((long long)nq * nsteps * 2 + ph) * 4 + wn) * (18 * 64);
    auto bload = [&](const float *src, int quad, float (&b)[18]) {     // quad 0..3: positions 4 quad ..+3; quad 4: positions 16, 17
        if (quad < 4) {
            const f32x4 t = *reinterpret_cast<const f32x4 *>(src + quad * 256 + lane * 4);
            b[4 * quad] = t[0]; b[4 * quad + 1] = t[1]; b[4 * quad + 2] = t[2]; b[4 * quad + 3] = t[3];
        } else {
            const float2 t = *reinterpret_cast<const float2 *>(src + 1024 + lane * 2);
            b[16] = t.x; b[17] = t.y;
        }
    };
    const long long u_s = 2ll * 4 * 18 * 64;    // k-step stride

    float bq[18];
    {   // prologue: all loads of the first two patch stages and of the first K-step's B operands in flight together
        f32x4 pq[2][NQ];
#pragma unroll
        for (int k = 0; k < NQ; ++k) pq[0][k] = quad_load(0, k);
#pragma unroll
        for (int k = 0; k < NQ; ++k) pq[1][k] = nst > 1 ? quad_load(1, k) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int q = 0; q < 5; ++q) bload(ub, q, bq);
#pragma unroll
        for (int k = 0; k < NQ; ++k) quad_store(0, k, pq[0][k]);
#pragma unroll
        for (int k = 0; k < NQ; ++k) quad_store(1, k, pq[1][k]);
    }
    W4_STAMP(1);
    __syncthreads();
    W4_STAMP(2);
    transform(0, 0);
    __syncthreads();
    W4_STAMP(3);
#ifdef DT_W4_TIMING
    ++tt_i;
#endif

    // one K-step = 36 MFMAs: this wave's 18 positions x the two blocks, one B register per position (reloaded for the
    // next K-step, four positions per 16-byte load, right after the fourth's MFMAs), A operand pairs fetched from LDS six positions ahead
    auto kstep = [&](int st, int s, const float *va) {
        const int gs = st * 2 + s;
        const bool stage = st + 2 < nst && !(DT_W4_ABLATE & 8);
        f32x4 sq[2];
        if (stage) {                                               // quads 2s, 2s+1 (< NQ) of the patch two stages ahead
#pragma unroll
            for (int k = 0; k < 2; ++k) sq[k] = (2 * s + k < NQ) ? quad_load(st + 2, 2 * s + k) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        }
        const float *un = ub + (long long)(gs + 1 < nsteps ? gs + 1 : gs) * u_s;
        const float *vs = va + (4 * s) * 16;                       // + position * 256; block 1 at + 128
        float a0[6], a1[6];
        auto fetch = [&](int i, int slot) {
            a0[slot] = (DT_W4_ABLATE & 2) ? 1.0f : vs[(18 * ph + i) * 256];
            a1[slot] = (DT_W4_ABLATE & 2) ? 1.0f : vs[(18 * ph + i) * 256 + 128];
        };
#pragma unroll
        for (int i = 0; i < 6; ++i) fetch(i, i);
#pragma unroll
        for (int i = 0; i < 18; ++i) {
            acc[0][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[i % 6], bq[i], acc[0][i], 0, 0, 0);
            acc[1][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[i % 6], bq[i], acc[1][i], 0, 0, 0);
            if (!(DT_W4_ABLATE & 1) && ((i & 3) == 3 || i == 17)) bload(un, i >> 2, bq);
            if (i + 6 < 18) fetch(i + 6, i % 6);
        }
        if (stage) {
#pragma unroll
            for (int k = 0; k < 2; ++k)
                if (2 * s + k < NQ) quad_store(st & 1, 2 * s + k, sq[k]);
        }
    };
#pragma unroll 1
    for (int st = 0; st < nst; ++st) {
        W4_STAMP(0);
        if (st + 1 < nst && !(DT_W4_ABLATE & 4)) transform((st + 1) & 1, (st + 1) & 1);
        W4_STAMP(1);
        const float *va = Vb + (st & 1) * W4_VBUF + kq * 16 + r;
        kstep(st, 0, va);
        kstep(st, 1, va);
        W4_STAMP(2);
        __syncthreads();
        W4_STAMP(3);
#ifdef DT_W4_TIMING
        ++tt_i;
#endif
    }
    W4_STAMP(0);

    // ---- epilogue.  The output transform At M' A is linear in the positions: every wave forms the PARTIAL 4x4 output
    // blocks of its 18 positions for both blocks (balanced VALU work), the two wave sets' partials are summed and
    // transposed in LDS -- half a block (8 pixel rows) of each block at a time: 2 x 35 KB -- and leave as 256-byte
    // pixel lines with bias + LeakyReLU (+ 2x2 max) applied on the way out.
    // C/D row = 4*kq + e (tile: row kq, column e of the block), col = lane & 15 (channel).
    // LDS image of half a block: [8 rows][16 px][68] floats, row pitch 1096 (the four tile rows a wave-instruction
    // touches -- rows 2 kq + a' of the half -- land 16 banks apart). ----
    constexpr int PIX = 68, RP = 16 * PIX + 8, AREA = 8 * RP;
    const int nl = wn * 16 + r;                       // channel within the workgroup's 64
    float y[2][4][16];
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float m[18], t[24];
#pragma unroll
            for (int i = 0; i < 18; ++i) m[i] = acc[g][i][e];
            w4_at_partial(m, ph, t);
#pragma unroll
            for (int a = 0; a < 4; ++a) w4_at(t + 6 * a, 1);      // over nu -> cols c = 0..3
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int c = 0; c < 4; ++c) y[g][e][4 * a + c] = t[6 * a + c];
        }
    const int fq = tid & 15;
    const f32x4 bv = *reinterpret_cast<const f32x4 *>(p.bias + nq * 64 + 4 * fq);
    auto act = [&](f32x4 v) {
#pragma unroll
        for (int c = 0; c < 4; ++c) { const float z = v[c] + bv[c]; v[c] = z > 0.0f ? z : z * p.slope; }
        return v;
    };
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {                  // pixel rows 4 kq + 2 hf + {0, 1} of both blocks
        // area g holds block g.  Wave set ph stores its partial of block ph, then adds its partial of block 1 - ph
        // (wave-uniform branches keep the register indices of y static)
        auto put = [&](float *area, const float (&yy)[4][16], bool add) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int al = 0; al < 2; ++al)
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        float *d = area + (2 * kq + al) * RP + (4 * e + c) * PIX + nl;
                        const float v = yy[e][4 * (2 * hf + al) + c];
                        *d = add ? *d + v : v;
                    }
        };
        if (ph == 0) put(lds, y[0], false); else put(lds + AREA, y[1], false);
        __syncthreads();
        if (ph == 0) put(lds + AREA, y[1], true); else put(lds, y[0], true);
        __syncthreads();
        if (!(DT_W4_ABLATE & 16)) {
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const int j = j0 + g;
                if (j >= nblk) continue;
                const int bx = j % p.nbx, by = (j / p.nbx) % p.nby, b = j / (p.nbx * p.nby);
                const float *area = lds + g * AREA;
                if (!POOL) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int pix = k * 32 + (tid >> 4);             // 128 pixels of the half block
                        const int lr = pix >> 4, px = pix & 15;
                        const int oy = by * 16 + 4 * (lr >> 1) + 2 * hf + (lr & 1), ox = bx * 16 + px;
                        if (oy < p.H && ox < p.W)
                            *reinterpret_cast<f32x4 *>(p.out + (long long)b * p.out_bs + ((long long)oy * p.W + ox) * p.out_ld + nq * 64 + 4 * fq) =
                                act(*reinterpret_cast<const f32x4 *>(area + lr * RP + px * PIX + 4 * fq));
                    }
                } else {
                    const int H2 = p.H >> 1, W2 = p.W >> 1;
                    const int pp = tid >> 4;                             // 32 pooled pixels of the half block: 4 rows x 8
                    const int prow = pp >> 3, pcol = pp & 7;             // pooled row prow <- LDS rows 2 prow, 2 prow + 1
                    const float *src = area + (2 * prow) * RP + (2 * pcol) * PIX + 4 * fq;
                    const f32x4 v00 = act(*reinterpret_cast<const f32x4 *>(src)), v01 = act(*reinterpret_cast<const f32x4 *>(src + PIX));
                    const f32x4 v10 = act(*reinterpret_cast<const f32x4 *>(src + RP)), v11 = act(*reinterpret_cast<const f32x4 *>(src + RP + PIX));
                    f32x4 mx;
#pragma unroll
                    for (int c = 0; c < 4; ++c) mx[c] = fmaxf(fmaxf(v00[c], v01[c]), fmaxf(v10[c], v11[c]));
                    const int py = by * 8 + 2 * prow + hf, qx = bx * 8 + pcol;
                    if (py < H2 && qx < W2)
                        *reinterpret_cast<f32x4 *>(p.out2 + (((long long)b * H2 + py) * W2 + qx) * p.out2_ld + nq * 64 + 4 * fq) = mx;
                }
            }
        }
        __syncthreads();
        W4_STAMP(1 + hf);
    }
}

int launch_wino4_fused(hipStream_t st, const Wino4FusedArgs &a_in)
{
    Wino4FusedArgs a = a_in;
    if (a.B <= 0 || a.Cin % W4_SC || a.N % 64 || a.in_ld % 4) return 2;
    if ((a.out && a.out_ld % 4) || (a.out2 && a.out2_ld % 4)) return 2;       // float4 pixel lines
    const bool pool = a.out2 != nullptr;
    if (pool && ((a.H | a.W) & 1)) return 2;
    if (pool == (a.out != nullptr)) return 2;       // exactly one of the two outputs
    a.nby = (a.H + 15) / 16;
    a.nbx = (a.W + 15) / 16;
    const long long blocks = (long long)a.B * a.nby * a.nbx;
    if (blocks >= (1ll << 31)) return 2;
    const size_t lds = (size_t)2 * (W4_PBUF + W4_VBUF) * sizeof(float);      // 123,008 B
    static PerDeviceOnce attr;
    if (attr.first()) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(wino4_fused_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void *>(wino4_fused_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return 1;
        attr.done();
    }
    const dim3 grid((unsigned)((blocks + 1) / 2), (unsigned)(a.N / 64));
    if (pool) hipLaunchKernelGGL(wino4_fused_kernel<true>, grid, dim3(W4_THREADS), lds, st, a);
    else hipLaunchKernelGGL(wino4_fused_kernel<false>, grid, dim3(W4_THREADS), lds, st, a);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

// Host: u36 = wino_pack_weights(4, ...) output [36][npad][cin] (U_p[n][c]) -> the kernel's B-operand stream
//   dst[nq][k-step][ph 2][wn 4][1152], 1152 = [quad 4][lane = kq*16 + j][4 positions] ++ [lane][2 positions (16, 17)]:  element = U_{18 ph + i}[c = 4 kstep + kq][n = 64 nq + 16 wn + j]
void wino4_fused_pack(const float *u36, int npad, int cin, int cout, float *dst)
{
    const int nsteps = cin / 4, nquart = cout / 64;
    const size_t plane = (size_t)npad * cin;
    for (int nq = 0; nq < nquart; ++nq)
        for (int gs = 0; gs < nsteps; ++gs)
            for (int ph = 0; ph < 2; ++ph)
                for (int wn = 0; wn < 4; ++wn)
                    for (int i = 0; i < 18; ++i)
                        for (int kq = 0; kq < 4; ++kq)
                            for (int j = 0; j < 16; ++j) {
                                const int c = 4 * gs + kq, n = nq * 64 + wn * 16 + j, pos = 18 * ph + i;
                                const int ln = kq * 16 + j;
                                const size_t off = i < 16 ? (size_t)(i >> 2) * 256 + ln * 4 + (i & 3) : (size_t)1024 + ln * 2 + (i - 16);
                                dst[((((size_t)nq * nsteps + gs) * 2 + ph) * 4 + wn) * 1152 + off] =
                                    u36[(size_t)pos * plane + (size_t)n * cin + c];
                            }
}

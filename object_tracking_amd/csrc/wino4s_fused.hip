// wino4s_fused.hip -- second-generation fused Winograd F(4x4,3x3) kernel for the early wide 3x3 layers
// (conv_2: 32 -> 64 at 208x208 (+pool), conv_3 / conv_5: 64 -> 128 at 104x104, conv_6 / conv_8: 128 -> 256 at 52x52;
// models_detection/KerasYOLO.py:285-320).  V and M' never leave the CU.  How the operands reach the matrix cores (the first
// fused F(4x4) kernel of round 2, deleted in round 4, streamed U through each wave's registers: profiles/HISTORY.md):
//
//   * the B operand U = G g Gt does not stream through each wave's registers from L2 (a CU sustains only ~17 B/clk of such
//     loads: 38 % of the MFMA peak in the round-2 kernel): a stage's U slice is brought into LDS ONCE per
//     workgroup with asynchronous global->LDS DMA (global_load_lds_dwordx4, 36 x 1 KiB pieces per 4-channel stage,
//     double-buffered) and all eight waves read their fragments from there with conflict-free ds_read_b128;
//   * every wave owns ALL 36 Winograd positions of one block of 16 tiles x 16 output channels (144 accumulators), so
//     the output transform At M' A happens in the wave's own registers -- no cross-wave reduction through LDS, and
//     the epilogue is a short in-register pass followed by 64-byte-segment stores;
//   * the input patches also arrive by DMA (16 bytes = 4 channels of one pixel per lane); the LDS image is padded by
//     one 16-byte slot per 4 pixels and 41 slots per row, which makes the 6x6 window reads of the input transform
//     conflict-free without any per-read address arithmetic (slot = 164 ty + 9 tx + 41 i + 2 j + (j >> 2) + half);
//   * the workgroup is PERSISTENT and the stage pipeline runs ACROSS items: during the last stages of an item the DMA
//     already fetches the next item's first patch stages and U stage, and the last stage's transform slot produces the
//     next item's V(0) -- only the first item of a workgroup pays a prologue;
//   * two wave sets swap roles every stage: one issues the stage's data movement and then its MFMAs, the other its MFMAs
//     and then the input transform of the next stage -- on every SIMD one wave feeds the matrix pipe while its partner does
//     the side work.  Measured slower, each of them: the side work issued BETWEEN the MFMA quads of every wave; 2-6 of a wave's 9 U
//     pieces per stage moved to the transform set (at the start of its stage, between its MFMA block and its transform, or in its
//     ~1700 cycles of barrier slack after the transform); the patch pieces ahead of the U pieces; the transform's window reads ahead
//     of the MFMA block (profiles/HISTORY.md, profiles/r04_experiments.txt, profiles/r04_s4_timing.txt).
//
// Workgroup = 8 waves = 2 blocks (4x4 tiles of 4x4 pixels each) x 4 column groups of 16 channels; v_mfma_f32_16x16x4_f32
// (row = tile, column = channel, k = input channel).  K runs in stages of 4 input channels (one MFMA k-step), one
// barrier per stage:
//      stage s :  set s&1:      DMA U(s+1) -> Ubuf[(s+1)&1], half a patch stage -> Pbuf;  36 MFMAs on V(s), U(s)
//                 the other set: 36 MFMAs on V(s), U(s);  V(s+1) = Bt d B for (block, xi-half) from the patch -> Vbuf[(s+1)&1]
//                 vmcnt(0); barrier.
// LDS (157.7 KB of 160): U 2 x 36.9 KB | V 2 x 18.4 KB | patch 2 x 24.6 KB.
// fp32 throughout; rounding identical in kind to winograd.hip TS = 4 (products summed in another order).
#include <type_traits>

#include "dt_internal.h"

typedef __attribute__((address_space(1))) const void s4_gptr_t;
typedef __attribute__((address_space(3))) void s4_lptr_t;

#ifdef DT_S4_TIMING
// debug build only (tools/s4_timing.py): per workgroup / wave / item cycle stamps and per-phase cycle sums
#define S4_TT_WG 64
#define S4_TT_ITEMS 8
#define S4_TT_SLOTS 10     // 0 start, 1 prologue end, 2 loop end, 3 epilogue end, 4 sum(transform), 5 sum(mfma block), 6 sum(vmcnt wait, data-movement
                           // stages), 7 sum(dma issue), 8 sum(barrier wait, data-movement stages), 9 sum(vmcnt + barrier wait, transform stages)
__device__ unsigned long long g_s4_times[S4_TT_WG * 8 * S4_TT_ITEMS * S4_TT_SLOTS];
extern "C" __attribute__((visibility("default"))) int dt_debug_s4_times(unsigned long long *dst, int clear)
{
    if (clear) {
        void *p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_s4_times)) != hipSuccess) return 1;
        return hipMemset(p, 0, sizeof(g_s4_times)) == hipSuccess ? 0 : 1;
    }
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_s4_times), sizeof(g_s4_times)) == hipSuccess ? 0 : 1;
}
#define S4_NOW() __builtin_readcyclecounter()
#define S4_PUT(k, v)                                                                                              \
    do {                                                                                                          \
        if (lane == 0 && blockIdx.x < S4_TT_WG && tt_i < S4_TT_ITEMS)                                             \
            g_s4_times[((blockIdx.x * 8 + wave) * S4_TT_ITEMS + tt_i) * S4_TT_SLOTS + (k)] = (v);                 \
    } while (0)
#else
#define S4_NOW() 0ull
#define S4_PUT(k, v) do { } while (0)
#endif

#define S4_THREADS 512
#define S4_UBUF (36 * 4 * 64)          // floats per U stage: [pg 9][wn 4][lane 64][4 positions]
#define S4_VBUF (36 * 2 * 64)          // floats per V stage: [pg2 18][blk 2][lane 64][2 positions]
#define S4_PROW 41                     // 16-byte slots per patch row: 2 px + (px >> 2) + half, 40 used
#define S4_PBLK 768                    // slots per block (18 rows x 41 = 738 used): 12 DMA pieces
#define S4_PBUF (2 * S4_PBLK * 4)      // floats per patch buffer (two blocks, 8 channels)
#define S4_LDS_FLOATS (2 * S4_UBUF + 2 * S4_VBUF + 2 * S4_PBUF)

typedef float f32x2 __attribute__((ext_vector_type(2)));   // NOT HIP's float2: LDS accesses through the struct type carry TBAA
                                                           // info that makes hipcc wait vmcnt(0) for every LDS-DMA in flight

template <typename T>
__device__ __forceinline__ void s4_at(T *m, int st)      // At (4x6): 6 inputs -> 4 outputs in the first 4 slots (T: float or f32x2)
{
    const T a = m[st] + m[2 * st], b = m[st] - m[2 * st], c = m[3 * st] + m[4 * st], e = m[3 * st] - m[4 * st];
    const T y0 = m[0] + a + c, y1 = b + 2.0f * e, y2 = a + 4.0f * c, y3 = b + 8.0f * e + m[5 * st];
    m[0] = y0; m[st] = y1; m[2 * st] = y2; m[3 * st] = y3;
}

// Bt d B restricted to three xi rows (HALF 0: rows 0-2, HALF 1: rows 3-5) of one 6x6 window, in two steps so that the LDS round trip
// of the window reads can be taken BEFORE the wave's MFMA block and the arithmetic after it (S4_TLOAD_EARLY):
//   s4_window_load: the five window rows the half needs (HALF 0: rows 0-4, HALF 1: rows 1-5; 30 values); pl = window pixel (0,0) of
//                   this lane's channel in the padded patch image
//   s4_transform_half: column pass, row pass, stores; o = this lane's slot of pair 0 of the half in the V stage
__device__ __forceinline__ void s4_window_load(const float *pl, int half, float (&w5)[5][6])
{
    const float *p0 = pl + half * (S4_PROW * 4);
#pragma unroll
    for (int j = 0; j < 6; ++j)
#pragma unroll
        for (int i = 0; i < 5; ++i) w5[i][j] = p0[(S4_PROW * i + 2 * j + (j >> 2)) * 4];
}
template <int HALF>
__device__ __forceinline__ void s4_transform_half(const float (&w5)[5][6], float *o)
{
    float t[3][6];
#define W(i, j) w5[(i) - HALF][j]           /* window row i: HALF 0 never reads row 5, HALF 1 never row 0 */
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        if (HALF == 0) {           // Bt rows 0, 1, 2
            t[0][j] = 4.0f * W(0, j) - 5.0f * W(2, j) + W(4, j);
            t[1][j] = -4.0f * (W(1, j) + W(2, j)) + W(3, j) + W(4, j);
            t[2][j] = 4.0f * (W(1, j) - W(2, j)) - W(3, j) + W(4, j);
        } else {                   // Bt rows 3, 4, 5
            t[0][j] = 2.0f * (W(3, j) - W(1, j)) - W(2, j) + W(4, j);
            t[1][j] = 2.0f * (W(1, j) - W(3, j)) - W(2, j) + W(4, j);
            t[2][j] = 4.0f * W(1, j) - 5.0f * W(3, j) + W(5, j);
        }
    }
#undef W
#pragma unroll
    for (int x = 0; x < 3; ++x) {
        const float d0 = t[x][0], d1 = t[x][1], d2 = t[x][2], d3 = t[x][3], d4 = t[x][4], d5 = t[x][5];
        f32x2 v01, v23, v45;
        v01.x = 4.0f * d0 - 5.0f * d2 + d4;
        v01.y = -4.0f * (d1 + d2) + d3 + d4;
        v23.x = 4.0f * (d1 - d2) - d3 + d4;
        v23.y = 2.0f * (d3 - d1) - d2 + d4;
        v45.x = 2.0f * (d1 - d3) - d2 + d4;
        v45.y = 4.0f * d1 - 5.0f * d3 + d5;
        // positions 6 (3 HALF + x) + nu: pair index pg2 = 9 HALF + 3 x + nu / 2
        *reinterpret_cast<f32x2 *>(o + (3 * x + 0) * 256) = v01;
        *reinterpret_cast<f32x2 *>(o + (3 * x + 1) * 256) = v23;
        *reinterpret_cast<f32x2 *>(o + (3 * x + 2) * 256) = v45;
    }
}

#ifndef S4_NTS
#define S4_NTS 1            // 1: the epilogue's activation stores as nontemporal stores (A/B builds)
#endif
#if S4_NTS
#define S4_STORE(ptr, val) __builtin_nontemporal_store((val), (ptr))
#else
#define S4_STORE(ptr, val) (*(ptr) = (val))
#endif
#ifndef S4_ABLATE
#define S4_ABLATE 0         // timing-only ablation builds (results WRONG): 1 no U DMA, 2 no patch DMA, 4 no input transform, 8 no operand reads,
                            // 16 patch addresses as for a channel-blocked input, 32 patch pieces from a contiguous source
#endif
#ifndef S4_SRD
#define S4_SRD 0            // 1: patch pieces through a per-frame buffer descriptor (buffer_load ... lds): an out-of-range offset reads
#endif                      //    zeros, so padding needs no zero-source select and an interior block costs ONE add per piece; 0: global_load_lds
                            //    with a zero-block source.  Both pass the parity tests; measured 1.5 % SLOWER with the descriptor
                            //    (conv_2 / 3 / 5: 8.30 / 9.94 / 9.33 against 8.17 / 9.79 / 9.17 ms): the pieces' cost is not their address work
#ifndef S4_PRIO
#define S4_PRIO 1           // raise the wave's issue priority while it does side work (DMA issue, input transform) beside its
#endif                      // partner's MFMA block
#ifndef S4_PATCH_FIRST
#define S4_PATCH_FIRST 0    // 1: the data-movement set issues its three patch pieces (HBM: the longest latency) BEFORE its nine U pieces (L2 hits).
#endif                      // Measured in round 4: 30.0 against 26.3 ms per step for conv_2 + 3 + 5 (and the vmcnt wait it was meant to shorten
                            // is 50-70 cycles per stage: the pieces have long landed when a wave reaches its end-of-stage wait)
#ifndef S4_TLOAD_EARLY
#define S4_TLOAD_EARLY 0    // 1: the transforming set requests its 30 window values BEFORE its MFMA block and computes after it (the LDS round
#endif                      // trip under the MFMAs).  Measured in round 4: 26.9-27.0 against 25.7-26.0 ms per step for conv_2 + 3 + 5 -- slower
                            // (30 more live registers across the MFMA block, and the block's first operand reads queue behind the 30 requests)
#ifndef S4_PF
#define S4_PF 2             // MFMA operand prefetch distance in quads (1 or 2)
#endif

template <bool POOL>
__global__ __launch_bounds__(S4_THREADS) void wino4s_fused_kernel(Wino4FusedArgs p)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *const Ub = lds;                                  // [2][S4_UBUF]
    float *const Vb = lds + 2 * S4_UBUF;                    // [2][S4_VBUF]
    float *const Pb = lds + 2 * S4_UBUF + 2 * S4_VBUF;      // [2][S4_PBUF]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int blk = wave >> 2, wn = wave & 3;               // MFMA role: block, 16-channel column group
    const int kq = lane >> 4, r16 = lane & 15;
    const int nst = p.Cin >> 2;                             // MFMA stages of 4 input channels (a multiple of 4: Cin % 16 == 0)
    const int npatch = nst >> 1;                            // patch stages of 8 channels (even)
    const int nblk = p.B * p.nby * p.nbx;
    const int npair = (nblk + 1) >> 1;
    const int nitems = npair * (p.N >> 6);
    if ((int)blockIdx.x >= nitems) return;

    // ---- wave sets.  Set 0 = waves 0-3 (block 0), set 1 = waves 4-7 (block 1); the two waves that share a SIMD belong to
    // different sets.  In stage s, set (s & 1) is the DATA-MOVEMENT set: it issues all of the stage's DMA, then its MFMAs;
    // set (s & 1) ^ 1 issues its MFMAs first and then runs the input transform of stage s + 1.  So on every SIMD one wave
    // feeds the matrix pipe while its partner does the side work, and they swap half-way through the stage. ----
    const int vset = wave >> 2, w4 = wave & 3;
    // V production role within a set: (block, xi half); lanes 0-31 take tile rows 0-1, lanes 32-63 tile rows 2-3
    const int vblk = wave & 1, vhalf = (wave >> 1) & 1;
    const int vtile = (lane & 7) | ((lane >> 5) << 3), vk = (lane >> 3) & 3;
    const int vty = vtile >> 2, vtx = vtile & 3;
    const int vsrc = (vblk * S4_PBLK + 164 * vty + 9 * vtx) * 4 + vk;       // float index of window pixel (0,0), half 0
    const int vdst = (vblk * 64 + vk * 16 + vtile) * 2 + vhalf * (9 * 256); // float index in a V stage: pg2 = 9 half + 3 x + nu / 2

    // ---- patch DMA geometry of this lane for the block-image pieces w4, w4 + 4, w4 + 8 (the same for both blocks):
    // packed (py << 8 | px << 2 | half << 1 | exists); the source offset is formed at issue time ----
    int pgeo[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int P = (w4 + 4 * i) * 64 + lane;             // slot within the block image
        const int py = P / S4_PROW, rr = P - py * S4_PROW;
        const int g = rr / 9, r9 = rr - 9 * g;
        const int px = 4 * g + (r9 >> 1), hf = r9 & 1;
        pgeo[i] = (py << 8) | (px << 2) | (hf << 1) | ((py < 18 && r9 < 8 && px < 18) ? 1 : 0);
    }
#if S4_SRD
    // byte offset of the lane's 16 bytes from the block's patch origin (channel 0), or far out of range for the slots no
    // pixel maps to: with the frame as a raw buffer (num_records = its byte size) such lanes read zeros
    constexpr int S4_OOB = 0x40000000;
    int pstat[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int py = pgeo[i] >> 8, px = (pgeo[i] >> 2) & 63, hf = (pgeo[i] >> 1) & 1;
        pstat[i] = (pgeo[i] & 1) ? ((py * p.W + px) * p.in_ld + hf * 4) * 4 : S4_OOB;
    }
    const unsigned frame_bytes = (unsigned)(((long long)(p.H - 1) * p.W + p.W - 1) * p.in_ld + p.Cin) * 4u;   // last pixel's channels end here
#endif
    // an item = (64-channel slice nq, block pair j0, j0 + 1).  The geometry of its two block images is decoded ONCE here (the
    // integer divisions cost hundreds of cycles): origin pixel of the patch, its address for channel 0, existence.
    struct Blk { const float *base; const float *frame; int boff; int y0, x0; bool ok, interior; };   // named members only: arrays indexed at run time land in scratch
    struct Item { int nq, j0; const float *u; Blk b0, b1; };
    auto blk_of = [&](int j, bool exists) {
        Blk B;
        const int bxy = p.nbx * p.nby;
        const int b = j / bxy, r = j - b * bxy;
        const int by = r / p.nbx, bx = r - by * p.nbx;
        B.y0 = by * 16 - 1; B.x0 = bx * 16 - 1;
        B.frame = p.in + (long long)b * p.in_bs;
        B.boff = (B.y0 * p.W + B.x0) * p.in_ld * 4;                   // bytes from the frame to the patch origin (negative on the top / left border)
        B.base = B.frame + ((long long)B.y0 * p.W + B.x0) * p.in_ld;   // wave-uniform
        B.ok = exists && j < nblk;
        B.interior = by > 0 && bx > 0 && B.y0 + 18 <= p.H && B.x0 + 18 <= p.W;
        return B;
    };
    auto item_of = [&](int it, bool exists) {
        Item I;
        I.nq = it / npair;
        I.j0 = 2 * (it - I.nq * npair);
        I.u = p.u + (long long)I.nq * nst * S4_UBUF;
        I.b0 = blk_of(I.j0, exists);
        I.b1 = blk_of(I.j0 + 1, exists);
        return I;
    };
    // pieces w4 + 4 i (i = 0..2) of block image b01 of patch stage c (channels 8c .. 8c+7) of item I -> patch buffer buf
    auto patch_half = [&](const Blk &B, int b01, int c, int buf, bool exists) {
        const int y0 = B.y0, x0 = B.x0;
        const float *base = B.base + 8 * c;
        const bool blk_ok = exists && B.ok;
#if S4_SRD
        if (!(S4_ABLATE & 48)) {
            // raw buffer over the block's frame (word 3: 32-bit data format, no swizzle); the stage's channel offset rides in soffset
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(B.frame), 0, blk_ok ? frame_bytes : 0u, 0x00020000);
            const bool fast = B.interior;        // wave-uniform: every pixel of the 18x18 window is inside the image
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                int voff = pstat[i] + B.boff;
                if (!fast) {
                    const int py = pgeo[i] >> 8, px = (pgeo[i] >> 2) & 63;
                    const bool in = y0 + py >= 0 && y0 + py < p.H && x0 + px >= 0 && x0 + px < p.W;
                    voff = in ? voff : S4_OOB;
                }
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (s4_lptr_t *)(Pb + buf * S4_PBUF + (b01 * 12 + w4 + 4 * i) * 256), 16, voff, 32 * c, 0, 0);
            }
            return;
        }
#endif
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int py = pgeo[i] >> 8, px = (pgeo[i] >> 2) & 63, hf = (pgeo[i] >> 1) & 1;
            const bool ok = blk_ok && (pgeo[i] & 1) && y0 + py >= 0 && y0 + py < p.H && x0 + px >= 0 && x0 + px < p.W;
            const float *src = ok ? base + (py * p.W + px) * p.in_ld + hf * 4 : p.zeros;
            if (S4_ABLATE & 16)     // timing probe: the access pattern of a channel-blocked [C/8][H][W][8] input (values wrong)
                src = ok ? base + (long long)c * (p.H * p.W * 8 - 8) - (long long)(y0 * p.W + x0) * (p.in_ld - 8) + (py * p.W + px) * 8 + hf * 4 : p.zeros;
            if (S4_ABLATE & 32) src = p.u + (w4 + 4 * i) * 256 + lane * 4;                          // timing probe: a contiguous, always-valid source
            __builtin_amdgcn_global_load_lds((s4_gptr_t *)src, (s4_lptr_t *)(Pb + buf * S4_PBUF + (b01 * 12 + w4 + 4 * i) * 256), 16, 0, 0);
        }
    };
    auto u_pieces = [&](const float *ustage, int buf, int i0, int i1) {      // pieces w4 + 4 i, i in [i0, i1), of a stage's 36
#pragma unroll
        for (int i = 0; i < 9; ++i)
            if (i >= i0 && i < i1) {
                const int piece = w4 + 4 * i;
                __builtin_amdgcn_global_load_lds((s4_gptr_t *)(ustage + piece * 256 + lane * 4),
                                                 (s4_lptr_t *)(Ub + buf * S4_UBUF + piece * 256), 16, 0, 0);
            }
    };
    // input transform of one stage: (patch buffer pbuf, channel half hf) -> V buffer vbuf; this lane's (tile, channel), xi rows
    // 3 vhalf .. + 2.  All 30 window values are requested at once (one LDS round trip, not ten).
    float w5[5][6];
    auto transform_load = [&](int pbuf, int hf) {
        s4_window_load(Pb + pbuf * S4_PBUF + vsrc + hf * 4, vhalf, w5);
        __builtin_amdgcn_sched_barrier(0);      // all 30 requests go out before anything else is scheduled
    };
    auto transform_compute = [&](int vbuf) {
        float *o = Vb + vbuf * S4_VBUF + vdst;
        if (vhalf == 0) s4_transform_half<0>(w5, o);
        else s4_transform_half<1>(w5, o);
    };
    auto transform = [&](int pbuf, int hf, int vbuf) { transform_load(pbuf, hf); transform_compute(vbuf); };

    const float *const a_base = Vb + (blk * 64 + lane) * 2;            // + stage buffer + pg2 * 256
    const float *const b_base = Ub + (wn * 64 + lane) * 4;             // + stage buffer + pg * 1024

    f32x4 acc[36];
    // the 36 MFMAs of a stage: positions in quads; the operands of quad g+1 are requested BEFORE the MFMAs of quad g are
    // issued (pinned with sched_barrier: left alone hipcc sinks the reads behind the MFMAs and waits for them at once)
    auto mfma_block = [&](int sbuf) {
        const float *va = a_base + sbuf * S4_VBUF;
        const float *ua = b_base + sbuf * S4_UBUF;
        constexpr int NB = S4_PF + 1;                      // operand register sets in rotation
        f32x2 a0[NB], a1[NB];
        f32x4 bq[NB];
        auto request = [&](int g, int slot) {
            if (S4_ABLATE & 8) { a0[slot] = f32x2{1.0f, 2.0f}; a1[slot] = f32x2{3.0f, 4.0f}; bq[slot] = f32x4{1.0f, 1.0f, 1.0f, 1.0f}; return; }
            a0[slot] = *reinterpret_cast<const f32x2 *>(va + (2 * g) * 256);
            a1[slot] = *reinterpret_cast<const f32x2 *>(va + (2 * g + 1) * 256);
            bq[slot] = *reinterpret_cast<const f32x4 *>(ua + g * 1024);
        };
#pragma unroll
        for (int g = 0; g < S4_PF; ++g) request(g, g);
#pragma unroll
        for (int g = 0; g < 9; ++g) {
            const int c = g % NB;
            if (g + S4_PF < 9) request(g + S4_PF, (g + S4_PF) % NB);
            __builtin_amdgcn_sched_barrier(0);
            acc[4 * g + 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[c].x, bq[c][0], acc[4 * g + 0], 0, 0, 0);
            acc[4 * g + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[c].y, bq[c][1], acc[4 * g + 1], 0, 0, 0);
            acc[4 * g + 2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[c].x, bq[c][2], acc[4 * g + 2], 0, 0, 0);
            acc[4 * g + 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[c].y, bq[c][3], acc[4 * g + 3], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // ---- first item: prologue.  Patch stage 0 (set b fetches block image b), U stage 0, the block-0 half of patch stage 1
    // (its block-1 half is fetched by the data-movement set of stage 0); then V(0) by set 0. ----
    int item = blockIdx.x;
    float out_am = 0.0f;      // the largest |value| this lane stored (Wino4FusedArgs::amax_out)
    Item cur = item_of(item, true);
#ifdef DT_S4_TIMING
    int tt_i = 0;
    S4_PUT(0, S4_NOW());
#endif
    if (vset == 0) { patch_half(cur.b0, 0, 0, 0, true); u_pieces(cur.u, 0, 0, 5); }
    else { patch_half(cur.b1, 1, 0, 0, true); u_pieces(cur.u, 0, 5, 9); patch_half(cur.b0, 0, 1, 1, npatch > 1); }
    __builtin_amdgcn_s_waitcnt(0x0f70);     // vmcnt(0)
    __syncthreads();
    if (vset == 0) transform(0, 0, 0);
    __syncthreads();

#pragma unroll 1
    for (;;) {
        const int nxt = item + (int)gridDim.x;
        const bool has_next = nxt < nitems;
        const Item nx = item_of(has_next ? nxt : item, has_next);
#ifdef DT_S4_TIMING
        unsigned long long tt_tr = 0, tt_mm = 0, tt_bw = 0, tt_dm = 0, tt_bd = 0, tt_wt = 0;
#endif
        S4_PUT(1, S4_NOW());
#pragma unroll
        for (int i = 0; i < 36; ++i) acc[i] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

#pragma unroll 1
        for (int s = 0; s < nst; ++s) {
            [[maybe_unused]] const unsigned long long c0 = S4_NOW();
            const bool last = s + 1 == nst;
            const bool dset = vset == (s & 1);
            if (dset) {
                if (S4_PRIO) __builtin_amdgcn_s_setprio(2);
                // ---- data-movement set: U of the next stage (of this item, or stage 0 of the next item), then half a patch
                // stage: odd s -> block image 0 of patch stage (s+3)/2, even s -> block image 1 of patch stage (s+2)/2 (the
                // stage whose first half went out one stage earlier).  Past this item's patches the numbering continues
                // into the next item's (npatch is even, so the buffers line up). ----
                if (!S4_PATCH_FIRST && (!last || has_next) && !(S4_ABLATE & 1)) u_pieces(last ? nx.u : cur.u + (long long)(s + 1) * S4_UBUF, (s + 1) & 1, 0, 9);
                const int pc_all = (s + 2 + (s & 1)) >> 1;
                const bool pnx = pc_all >= npatch;
                if (!(S4_ABLATE & 2)) {
                    // block image (s & 1) ^ 1 of this item or the next: four wave-uniform candidates, selected on scalars
                    Blk B;
#define S4_SEL(m) B.m = (s & 1) ? (pnx ? nx.b0.m : cur.b0.m) : (pnx ? nx.b1.m : cur.b1.m)
                    S4_SEL(base); S4_SEL(frame); S4_SEL(boff); S4_SEL(y0); S4_SEL(x0); S4_SEL(ok); S4_SEL(interior);
#undef S4_SEL
                    patch_half(B, (s & 1) ^ 1, pnx ? pc_all - npatch : pc_all, pc_all & 1, !pnx || has_next);
                }
                if (S4_PATCH_FIRST && (!last || has_next) && !(S4_ABLATE & 1)) u_pieces(last ? nx.u : cur.u + (long long)(s + 1) * S4_UBUF, (s + 1) & 1, 0, 9);
                if (S4_PRIO) __builtin_amdgcn_s_setprio(0);
            }
            [[maybe_unused]] const unsigned long long c1 = S4_NOW();
            // ---- the other set: the input transform of stage s + 1 (stage 0 of the next item after the last stage) around its MFMAs:
            // patch buffer ((s+1)/2) & 1, channel half (s+1) & 1 (landed before the last barrier) -> V buffer (s+1) & 1 ----
            const bool do_tr = !dset && (!last || has_next) && !(S4_ABLATE & 4);
            const int s1 = s + 1;
            if (S4_TLOAD_EARLY && do_tr) transform_load((s1 >> 1) & 1, s1 & 1);
            mfma_block(s & 1);
            [[maybe_unused]] const unsigned long long c2 = S4_NOW();
            if (do_tr) {
                if (S4_PRIO) __builtin_amdgcn_s_setprio(2);
                if (!S4_TLOAD_EARLY) transform_load((s1 >> 1) & 1, s1 & 1);
                transform_compute(s1 & 1);
                if (S4_PRIO) __builtin_amdgcn_s_setprio(0);
            }
#ifdef DT_S4_TIMING
            tt_dm += c1 - c0; tt_mm += c2 - c1; tt_tr += S4_NOW() - c2;
#endif
            [[maybe_unused]] const unsigned long long c3 = S4_NOW();
            __builtin_amdgcn_s_waitcnt(0x0f70);     // this wave's DMA pieces have landed
            [[maybe_unused]] const unsigned long long c4 = S4_NOW();
            __syncthreads();
#ifdef DT_S4_TIMING
            if (dset) { tt_bw += c4 - c3; tt_bd += S4_NOW() - c4; }
            else tt_wt += S4_NOW() - c3;
#endif
        }
        S4_PUT(2, S4_NOW());

        // ---- epilogue: At M' A per (tile, channel) in registers; C/D row = 4 kq + e -> tile (ty = kq, tx = e), col = channel.
        // The DMAs of the next item's first stages are already in flight / landed; nothing here touches LDS. ----
        const int j = cur.j0 + blk;
        if (j < nblk) {
            const int bx = j % p.nbx, by = (j / p.nbx) % p.nby, b = j / (p.nbx * p.nby);
            const int ch = cur.nq * 64 + wn * 16 + r16;
            const float bias = p.bias[ch];
            // Retire the load HERE, on the straight path: left to be consumed inside the lane-divergent store branches
            // below, hipcc's waitcnt bookkeeping never sees it complete on every path and puts a vmcnt(0) -- which also
            // drains every store issued so far -- in front of each later use, and in front of every reuse of its register
            // in the next item's stage loop.
            asm volatile("" ::"v"(bias));
            const int y0 = by * 16 + 4 * kq, x0 = bx * 16;
            const bool full = by * 16 + 16 <= p.H && bx * 16 + 16 <= p.W;     // wave-uniform: no per-pixel bounds checks inside
            // two tiles (e, e + 1) at a time as float2 lanes: their accumulators are adjacent registers of acc[i], so every
            // operation of the output transform is one packed instruction (no MFMA runs beside the epilogue -- the one
            // place where v_pk_* f32 pays)
#pragma unroll
            for (int e2 = 0; e2 < 4; e2 += 2) {
                f32x2 m[36];
#pragma unroll
                for (int i = 0; i < 36; ++i) { m[i].x = acc[i][e2]; m[i].y = acc[i][e2 + 1]; }
#pragma unroll
                for (int nu = 0; nu < 6; ++nu) s4_at(m + nu, 6);        // over xi -> rows a = 0..3 (slots 6 a + nu)
#pragma unroll
                for (int a = 0; a < 4; ++a) s4_at(m + 6 * a, 1);        // over nu -> cols c = 0..3
                const f32x2 bias2 = {bias, bias}, slope2 = {p.slope, p.slope};
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const f32x2 z = m[6 * a + c] + bias2;
                        m[6 * a + c] = __builtin_elementwise_max(z, z * slope2);    // LeakyReLU, 0 <= slope <= 1 (1: linear)
                    }
                // two code paths, chosen wave-uniformly: interior blocks store unconditionally (no exec masking, no per-store
                // scalar work); border blocks check every pixel
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int e = e2 + h;
                    if (!POOL) {
                        float *ob = p.out + (long long)b * p.out_bs + ((long long)y0 * p.W + x0 + 4 * e) * p.out_ld + ch;
                        const int rs = p.W * p.out_ld;
                        if (full) {
#pragma unroll
                            for (int a = 0; a < 4; ++a)
#pragma unroll
                                for (int c = 0; c < 4; ++c) {
                                    if (p.amax_out) out_am = fmaxf(out_am, fabsf(m[6 * a + c][h]));
                                    S4_STORE(&ob[a * rs + c * p.out_ld], m[6 * a + c][h]);
                                }
                        } else {
#pragma unroll
                            for (int a = 0; a < 4; ++a)
#pragma unroll
                                for (int c = 0; c < 4; ++c)
                                    if (y0 + a < p.H && x0 + 4 * e + c < p.W) {
                                        if (p.amax_out) out_am = fmaxf(out_am, fabsf(m[6 * a + c][h]));
                                        S4_STORE(&ob[a * rs + c * p.out_ld], m[6 * a + c][h]);
                                    }
                        }
                    } else {
                        const int H2 = p.H >> 1, W2 = p.W >> 1;
                        float *ob = p.out2 + (((long long)b * H2 + (y0 >> 1)) * W2 + (x0 >> 1) + 2 * e) * p.out2_ld + ch;
                        const int rs = W2 * p.out2_ld;
                        float mx[2][2];
#pragma unroll
                        for (int a2 = 0; a2 < 2; ++a2)
#pragma unroll
                            for (int c2 = 0; c2 < 2; ++c2)
                                mx[a2][c2] = fmaxf(fmaxf(m[6 * (2 * a2) + 2 * c2][h], m[6 * (2 * a2) + 2 * c2 + 1][h]),
                                                   fmaxf(m[6 * (2 * a2 + 1) + 2 * c2][h], m[6 * (2 * a2 + 1) + 2 * c2 + 1][h]));
                        if (full) {
#pragma unroll
                            for (int a2 = 0; a2 < 2; ++a2)
#pragma unroll
                                for (int c2 = 0; c2 < 2; ++c2) {
                                    if (p.amax_out) out_am = fmaxf(out_am, fabsf(mx[a2][c2]));
                                    S4_STORE(&ob[a2 * rs + c2 * p.out2_ld], mx[a2][c2]);
                                }
                        } else {
#pragma unroll
                            for (int a2 = 0; a2 < 2; ++a2)
#pragma unroll
                                for (int c2 = 0; c2 < 2; ++c2)
                                    if ((y0 >> 1) + a2 < H2 && (x0 >> 1) + 2 * e + c2 < W2) {
                                        if (p.amax_out) out_am = fmaxf(out_am, fabsf(mx[a2][c2]));
                                        S4_STORE(&ob[a2 * rs + c2 * p.out2_ld], mx[a2][c2]);
                                    }
                        }
                    }
                }
            }
        }
#ifdef DT_S4_TIMING
        S4_PUT(3, S4_NOW()); S4_PUT(4, tt_tr); S4_PUT(5, tt_mm); S4_PUT(6, tt_bw); S4_PUT(7, tt_dm); S4_PUT(8, tt_bd); S4_PUT(9, tt_wt);
        ++tt_i;
        S4_PUT(0, S4_NOW());
#endif
        if (!has_next) break;
        item = nxt;
        cur = nx;
    }
    if (p.amax_out) dt_amax_publish(p.amax_out, out_am);      // (max |x| of the stored outputs: the fp16 form of the next layer's GEMM scales by it)
}

int launch_wino4s_fused(hipStream_t st, const Wino4FusedArgs &a_in, const float *zeros)
{
    Wino4FusedArgs a = a_in;
    if (a.B <= 0 || a.Cin % 16 || a.N % 64 || a.in_ld % 4) return 2;   // Cin % 16: an even number of 8-channel patch stages
    const bool pool = a.out2 != nullptr;
    if (pool && ((a.H | a.W) & 1)) return 2;
    if (pool == (a.out != nullptr)) return 2;       // exactly one of the two outputs
    a.nby = (a.H + 15) / 16;
    a.nbx = (a.W + 15) / 16;
    const long long blocks = (long long)a.B * a.nby * a.nbx;
    if (blocks >= (1ll << 30)) return 2;
    a.zeros = zeros;
    const size_t lds = (size_t)S4_LDS_FLOATS * sizeof(float);      // 157,696 B
    static PerDeviceOnce attr;
    static int cus[64];
    int dev = 0;
    if (attr.ensure(&dev, [&](int d) {
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(wino4s_fused_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
                hipFuncSetAttribute(reinterpret_cast<const void *>(wino4s_fused_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
                return 1;
            int n = 0;
            if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, d) != hipSuccess || n <= 0) n = 256;
            cus[d] = n;
            return 0;
        }))
        return 1;
    const long long items = ((blocks + 1) / 2) * (a.N / 64);
    long long grid = cus[dev];                 // one 8-wave workgroup per CU (157 KB of LDS), persistent over the items
    if (grid > items) grid = items;
    if (pool) hipLaunchKernelGGL(wino4s_fused_kernel<true>, dim3((unsigned)grid), dim3(S4_THREADS), lds, st, a);
    else hipLaunchKernelGGL(wino4s_fused_kernel<false>, dim3((unsigned)grid), dim3(S4_THREADS), lds, st, a);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

// Host: u36 = wino_pack_weights(4, ...) output [36][npad][cin] (U_p[n][c]) -> the kernel's LDS image, stage by stage:
//   dst[nq][stage s][pg 9][wn 4][lane = kq*16 + j][4]:  element = U_{4 pg + i}[c = 4 s + kq][n = 64 nq + 16 wn + j]
void wino4s_fused_pack(const float *u36, int npad, int cin, int cout, float *dst)
{
    const int nst = cin / 4, nquart = cout / 64;
    const size_t plane = (size_t)npad * cin;
    for (int nq = 0; nq < nquart; ++nq)
        for (int s = 0; s < nst; ++s)
            for (int pg = 0; pg < 9; ++pg)
                for (int wn = 0; wn < 4; ++wn)
                    for (int kq = 0; kq < 4; ++kq)
                        for (int j = 0; j < 16; ++j)
                            for (int i = 0; i < 4; ++i) {
                                const int c = 4 * s + kq, n = nq * 64 + wn * 16 + j, pos = 4 * pg + i;
                                dst[((((size_t)nq * nst + s) * 9 + pg) * 4 + wn) * 256 + (kq * 16 + j) * 4 + i] =
                                    u36[(size_t)pos * plane + (size_t)n * cin + c];
                            }
}

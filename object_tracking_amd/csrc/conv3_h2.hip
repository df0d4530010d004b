// conv3_h2.hip -- DIRECT 3x3 'same' stride-1 convolution + BatchNorm (folded) + LeakyReLU [+ MaxPooling2D(2,2)] of the reference's narrow
// early conv blocks (models_detection/KerasYOLO.py:285-320: conv_2, conv_3, conv_5) on the 16-bit matrix pipe at fp32 accuracy: the
// two-term fp16 form of wino_gemm_s3.hip (x 2^s = hi + lo, three products lo*hi + hi*lo + hi*hi on v_mfma_f32_32x32x16_f16, fp32
// accumulate; the power of two from the MEASURED max |x| of the input tensor, dt_internal.h: dt_h2_base) applied to the convolution itself.
//
// Why direct: with Cin = 32 / 64 the Winograd form is bound by its transforms (the fused F(4x4) kernel of wino4s_fused.hip runs the fp32
// MFMA instruction at 0.42 of its peak; the bf16 attempt of round 5 drowned in VALU work), while 9 * Cin * 3 fp16 MFMA products per output are
// 6.9 TFLOP per layer and 1440 frames: ~6 ms at the rate the split GEMMs reach, and nothing but MFMAs and LDS reads in the loop --
//   * the input PATCH of a tile ((TH + 2) x 18 pixels x 32 channels) is fetched ONCE per tile and 32-channel chunk, scaled and split
//     ONCE, and kept in LDS as two fp16 planes; a tap's A operand is the same image read at another pixel offset;
//   * the weights stream through a 3-stage LDS ring by LDS-DMA, one stage = one tap of the chunk (32 k: two MFMA k blocks);
//   * the next unit's patch is requested into registers at the start of a unit and written to LDS at its end (the loads have nine stages
//     of cover); two 256-thread workgroups per CU cover each other's patch turn-over and epilogue.
// Row order of a 32-row MFMA block: two image rows x 16 pixels, so that a 2x2 pooling window is registers {r, r + 1, r + 8, r + 9} of one
// lane and a store instruction writes whole 128-byte lines (32 channels of one pixel per half-wave).
// Weights: the layer's packed [Npad][9 * Cin] matrix (k = ((ci / 32) * 9 + tap) * 32 + ci % 32, conv_igemm.hip's order) as two fp16 terms
// [2][K / 16][Npad][16] of w * 2^sw (launch_wino_h2_pack, P = 1), epilogue factor pscale[0] = 2^-sw.
#include "dt_internal.h"
#include <type_traits>

typedef _Float16 c3_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 c3_h4 __attribute__((ext_vector_type(4)));
typedef float c3_f16 __attribute__((ext_vector_type(16)));
typedef float c3_f4 __attribute__((ext_vector_type(4)));
typedef unsigned c3_u4 __attribute__((ext_vector_type(4)));
typedef unsigned c3_u2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void c3_lptr_t;
typedef const __attribute__((address_space(1))) void c3_gptr_t;

#ifndef C3_PW
#define C3_PW 18          // pixels per patch row in LDS (16 + 2 halo; a build switch for bank-conflict experiments)
#endif
#define C3_THREADS 256
#define C3_NS 3           // weight stages in the LDS ring

template <int N>
__device__ __forceinline__ void c3_wait_vm() { __builtin_amdgcn_s_waitcnt(0x0f70 | (N & 15) | ((N >> 4) << 14)); }

__device__ __forceinline__ void c3_mfma(c3_f16 &c, const c3_u4 &w, const c3_u4 &a)      // D[i = pixel][j = channel]
{
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(c3_h8, a), __builtin_bit_cast(c3_h8, w), c, 0, 0, 0);
}

// WN = 1: 64 output channels per workgroup, waves 4 (rows) x 1, tile 16 x 16 pixels.  WN = 2: 128 channels, waves 2 x 2, tile 8 rows x 16.
// FUSE (WN = 2, no pooling: conv_3): the 1x1 layer that follows (conv_4: 128 -> 64, the only reader of this layer's output) is applied to
// the tile's activations before anything leaves the CU -- the 128-channel tensor (8 GB written, 8 GB read back per 1440 frames) never
// exists.  After the last tap: y = LeakyReLU(acc * inv + bias) in registers; the tile's own max |y| (a workgroup-local reduction) gives the
// power of two for y's two fp16 terms -- the second GEMM only ever sees this tile, so its scale need not wait for the tensor's maximum;
// four more pipeline stages (K = 128 in quarters of 32): the waves that hold a quarter's channels write them to LDS in the A operand's
// image (the patch region is free: the next patch waits in registers), every wave multiplies its 32 pixels x 64 output channels against
// the quarter's weights, which came through the same DMA ring as two more "taps" ... four more.
template <int WN, bool POOL, bool FUSE = false>
__global__ __launch_bounds__(C3_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv3_h2_kernel(Conv3H2Args p)
{
    static_assert(!FUSE || (WN == 2 && !POOL), "the fused 1x1 follows the 128-channel, un-pooled instance");
    constexpr int WM = 4 / WN;                       // waves along the pixel rows
    constexpr int TH = 4 * WM, TW = 16;              // tile: each wave owns 4 image rows = two MFMA blocks of 2 rows x 16 pixels
    constexpr int BN = 64 * WN;
    constexpr int PH = TH + 2, PW = C3_PW, NPIX = PH * PW;
    constexpr int PLANE = NPIX * 32;                 // bytes of one (half, term) plane of the patch
    constexpr int PATCH = 4 * PLANE;                 // [half 2][term 2]
    constexpr int BPL = BN * 32;                     // bytes of one (half, term) plane of a weight stage
    constexpr int BSTAGE = 4 * BPL;
    constexpr int PB = (4 * BN / 32) / 4;            // 1 KiB DMA pieces per wave and stage (32 rows x 32 B each): 2 (BN = 64) or 4
    constexpr int NJ = (PH * 18 * 8 + C3_THREADS - 1) / C3_THREADS;      // float4 loads per thread and patch
    extern __shared__ __attribute__((aligned(16))) unsigned char c3_lds[];
    unsigned char *patch = c3_lds, *bring = c3_lds + PATCH;
    float *btab = reinterpret_cast<float *>(c3_lds + PATCH + C3_NS * BSTAGE);      // [BN] bias of this workgroup's channel tile ... per item
    [[maybe_unused]] float *btab1 = btab + BN;        // FUSE: [64] bias of the fused 1x1 | [4] the waves' max |y|
    constexpr int SPL = 128 * 32;                     // FUSE: bytes of one (half, term) plane of the y scratch (128 pixels x 16 channels)
    constexpr int BPL1 = 64 * 32;                     // FUSE: ... of a stage of the 1x1's weights (64 rows)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave % WM, wn = wave / WM;
    const int rl = lane & 31, gl = lane >> 5;

    const unsigned am = dt_amax_read(p.amax);
    const float fwd = dt_h2_base(am);
    const float inv = p.pscale[0] * dt_h2_base_inv(am);
    const int nchunk = p.Cin >> 5;
    const int ntn = (p.N + BN - 1) / BN;
    const int tiles = p.tiles_x * p.tiles_y;
    const long long items = (long long)p.B * tiles * ntn;       // item = (frame, tile, channel tile), channel tile fastest
    const long long wterm = (long long)(9 * p.Cin / 16) * p.Np * 16;      // f16 elements of one term plane of the weights
    float out_am = 0.0f;

    long long item = blockIdx.x;
    if (item >= items) return;

    struct Unit { int b, ty0, tx0, n0, chunk; };
    auto unit_of = [&](long long it, int chunk) {
        Unit u;
        u.n0 = (int)(it % ntn) * BN;
        const long long r = it / ntn;
        const int t = (int)(r % tiles);
        u.b = (int)(r / tiles);
        u.ty0 = (t / p.tiles_x) * TH;
        u.tx0 = (t % p.tiles_x) * TW;
        u.chunk = chunk;
        return u;
    };
    // ---- patch: global -> registers (request), registers -> LDS (scale, split, store) ----
    c3_f4 pf[NJ];
    auto patch_request = [&](const Unit &u) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int e = tid + C3_THREADS * j;
            const int pix = e >> 3, c4 = e & 7;
            const int py = pix / 18, px = pix - py * 18;
            const int y = u.ty0 + py - 1, x = u.tx0 + px - 1;
            const bool ok = pix < PH * 18 && y >= 0 && y < p.H && x >= 0 && x < p.W;
            // the ADDRESS is selected (out-of-image pixels read the zero block), not the load: no lane-divergent branch around a load
            const float *src = ok ? p.in + (long long)u.b * p.in_bs + ((long long)y * p.W + x) * p.in_ld + u.chunk * 32 + c4 * 4 : p.zeros;
            pf[j] = *reinterpret_cast<const c3_f4 *>(src);
        }
    };
    auto patch_store = [&]() {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int e = tid + C3_THREADS * j;
            const int pix18 = e >> 3, c4 = e & 7;
            if (pix18 < PH * 18) {
                const int py = pix18 / 18, px = pix18 - py * 18;
                const int pix = py * PW + px;
                const c3_f4 x = pf[j] * fwd;
                const c3_h4 hi = __builtin_convertvector(x, c3_h4);
                const c3_f4 r = x - __builtin_convertvector(hi, c3_f4);
                const c3_h4 lo = __builtin_convertvector(r, c3_h4);
                // 16 channels of a pixel = two 16-byte granules, granule index XOR bit 3 of the pixel index (wino_gemm_s3.hip's image)
                const int half = c4 >> 2, g = ((c4 >> 1) & 1) ^ ((pix >> 3) & 1);
                unsigned char *dst = patch + (half * 2) * PLANE + pix * 32 + g * 16 + (c4 & 1) * 8;
                *reinterpret_cast<c3_u2 *>(dst) = __builtin_bit_cast(c3_u2, hi);
                *reinterpret_cast<c3_u2 *>(dst + PLANE) = __builtin_bit_cast(c3_u2, lo);
            }
        }
    };
    // ---- weight stages: a unit has nine (one per tap), stage `tap` lives in ring buffer tap % 3 (9 % 3 == 0: the same for every unit) ----
    // a piece = 32 rows x 32 B of one (half, term) plane: wave w moves pieces w * PB .. + PB - 1 of the 4 * BN / 32 of a stage.  Everything
    // of a piece's address that does not depend on the stage is computed once.
    const int lrow = lane >> 1;
    const int dgran = (lane & 1) ^ ((lrow >> 3) & 1);
    const unsigned short *wsrc[PB];
    int wdst[PB];
#pragma unroll
    for (int q = 0; q < PB; ++q) {
        const int pc = wave * PB + q;                 // piece index: (half, term, 32-row group)
        const int grp = pc % (BN / 32), ht = pc / (BN / 32);
        const int half = ht >> 1, term = ht & 1;
        wsrc[q] = p.w + term * wterm + ((long long)half * p.Np + 32 * grp + lrow) * 16 + dgran * 8;
        wdst[q] = ht * BPL + grp * 1024;
    }
    const long long wstage = (long long)2 * p.Np * 16;      // f16 elements from one stage's k blocks to the next's
    auto stage_off = [&](int n0, int chunk, int tap) { return (long long)(chunk * 9 + tap) * wstage + (long long)n0 * 16; };
    auto piece_issue = [&](int buf, long long off, int q) {
        __builtin_amdgcn_global_load_lds((c3_gptr_t *)(wsrc[q] + off), (c3_lptr_t *)(bring + buf * BSTAGE + wdst[q]), 16, 0, 0);
    };
    auto stage_issue = [&](int buf, int n0, int chunk, int tap) {
        const long long off = stage_off(n0, chunk, tap);
#pragma unroll
        for (int q = 0; q < PB; ++q) piece_issue(buf, off, q);
    };
    // operand read offsets
    int offB[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = wn * 64 + 32 * j + rl;
        offB[j] = (2 * row + (gl ^ ((row >> 3) & 1))) * 16;
    }
    // A: pixel of this lane's row in block i (before the tap offset): image rows 4 wm + 2 i + (rl >> 4), column rl & 15 of the tile
    int pixA[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) pixA[i] = (4 * wm + 2 * i + (rl >> 4)) * PW + (rl & 15);
    auto offA_of = [&](int i, int tap) {
        const int pix = pixA[i] + (tap / 3) * PW + tap % 3;
        return pix * 32 + ((gl ^ ((pix >> 3) & 1)) * 16);
    };

    // FUSE: pieces of a stage of the 1x1's weights (a quarter of K: k blocks 2 q, 2 q + 1): 8 of 1 KiB, two per wave
    [[maybe_unused]] const unsigned short *w1src[2] = {nullptr, nullptr};
    [[maybe_unused]] int w1dst[2] = {0, 0};
    if constexpr (FUSE) {
        const long long w1term = (long long)8 * p.Np1 * 16;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int pc = wave * 2 + q, grp = pc & 1, ht = pc >> 1, half = ht >> 1, term = ht & 1;
            w1src[q] = p.w1 + term * w1term + ((long long)half * p.Np1 + 32 * grp + lrow) * 16 + dgran * 8;
            w1dst[q] = ht * BPL1 + grp * 1024;
        }
        for (int i = tid; i < 64; i += C3_THREADS) btab1[i] = i < p.N1 ? p.bias1[i] : 0.0f;
    }
    auto fstage_issue = [&](int buf, int q, int k) {      // piece k (0, 1) of this wave of the 1x1's stage q
        if constexpr (FUSE)
            __builtin_amdgcn_global_load_lds((c3_gptr_t *)(w1src[k] + (long long)(2 * q) * p.Np1 * 16), (c3_lptr_t *)(bring + buf * BSTAGE + w1dst[k]), 16, 0, 0);
    };
    int sbase = 0;      // FUSE: ring buffer of a unit's tap 0 (an item ends with four more stages: 13 % 3 = 1); otherwise always 0
    auto bufof = [&](int k) { return FUSE ? (sbase + k) % C3_NS : k % C3_NS; };

    // ---- the sequence of units of this workgroup ----
    Unit cur = unit_of(item, 0);
    // prologue: first patch straight in, weight stages 0 and 1 in flight
    patch_request(cur);
    stage_issue(0, cur.n0, 0, 0);
    stage_issue(1, cur.n0, 0, 1);
    for (int i = tid; i < BN; i += C3_THREADS) btab[i] = (cur.n0 + i < p.N) ? p.bias[cur.n0 + i] : 0.0f;
    patch_store();

    c3_f16 acc[2][2];
    c3_u4 fa[2][2][2], fb[2][2][2];      // fragments [half][block][term]: half h of the running stage, the other half being filled for what comes next
    for (;;) {
        if (cur.chunk == 0) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[j][i][e] = 0.0f;
        }
        // the unit after this one
        const bool last_chunk = cur.chunk == nchunk - 1;
        const long long nitem = last_chunk ? item + gridDim.x : item;
        const bool has_next = nitem < items;
        Unit nxt = cur;
        if (has_next) nxt = unit_of(nitem, last_chunk ? 0 : cur.chunk + 1);

        // one stage = one tap, compile-time: 24 MFMAs (2 halves x 4 blocks x 3 products), 16 fragment reads pinned between them
        auto stage = [&](auto tapc) {
            constexpr int tap = decltype(tapc)::value;
            // EVERY weight stage issued so far has landed (this one and the next: the next one's first fragments are read ahead, below) --
            // in stage 1 all but the patch loads requested in stage 0
            if (tap == 1 && has_next) c3_wait_vm<NJ>();
            else c3_wait_vm<0>();
            __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0): this wave's LDS writes (patch, bias) and reads are done
            __builtin_amdgcn_s_barrier();
            // the buffer stage tap - 1 occupied is refilled with stage tap + 2 (of the next unit from tap 7 on) -- its PB pieces one per MFMA slot
            // of the first half's tail (a DMA instruction holds its issuer for 60 - 180 cycles: four at the head of a stage are felt)
            const bool fnext = FUSE && last_chunk && tap >= 7;      // what follows the ninth tap is the fused 1x1's first stages
            const bool refill = tap < 7 || (has_next && !fnext);
            const long long roff = tap < 7 ? stage_off(cur.n0, cur.chunk, tap + 2) : stage_off(nxt.n0, nxt.chunk, tap - 7);
            const unsigned char *sb = bring + bufof(tap) * BSTAGE, *sn = bring + bufof(tap + 1) * BSTAGE;
            if (tap == 0) {      // a unit's first stage reads its own first fragments (the patch was just turned over)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int t = 0; t < 2; ++t) fa[0][i][t] = *reinterpret_cast<const c3_u4 *>(patch + t * PLANE + offA_of(i, 0));
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int t = 0; t < 2; ++t) fb[0][j][t] = *reinterpret_cast<const c3_u4 *>(sb + t * BPL + offB[j]);
            }
            int oa[2], on[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) { oa[i] = offA_of(i, tap); on[i] = offA_of(i, tap < 8 ? tap + 1 : 0); }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                // products of half hf (three per multiply, smallest first, consecutive MFMAs to different accumulators); behind each of the
                // first eight: one fragment read of what runs next -- half 1 of this stage, or half 0 of the next stage (same patch, the
                // next ring buffer: landed, see above)
                const bool rd = hf == 0 || tap < 8;
#pragma unroll
                for (int m = 0; m < 12; ++m) {
                    const int pr = m >> 2, j = (m >> 1) & 1, i = m & 1;
                    c3_mfma(acc[j][i], fb[hf][j][pr == 0 ? 1 : 0], fa[hf][i][pr == 1 ? 1 : 0]);
                    if (rd && m < 8) {
                        const int k = m & 3, blk = k >> 1, t = k & 1;
                        if (m < 4) {
                            if (hf == 0) fa[1][blk][t] = *reinterpret_cast<const c3_u4 *>(patch + (2 + t) * PLANE + oa[blk]);
                            else fa[0][blk][t] = *reinterpret_cast<const c3_u4 *>(patch + t * PLANE + on[blk]);
                        } else {
                            if (hf == 0) fb[1][blk][t] = *reinterpret_cast<const c3_u4 *>(sb + (2 + t) * BPL + offB[blk]);
                            else fb[0][blk][t] = *reinterpret_cast<const c3_u4 *>(sn + t * BPL + offB[blk]);
                        }
                    }
                    if (hf == 0 && m >= 8 && m - 8 < PB && refill) piece_issue(bufof(tap + 2), roff, m - 8);
                    if (FUSE && hf == 0 && m >= 8 && m - 8 < 2 && fnext) fstage_issue(bufof(tap + 2), tap - 7, m - 8);
                    __builtin_amdgcn_sched_barrier(0);
                }
                // the next unit's patch: requested behind the weight pieces of stage 0 (the newest operations in flight at stage 1's wait)
                if (tap == 0 && hf == 0 && has_next) { patch_request(nxt); __builtin_amdgcn_sched_barrier(0); }
            }
        };
        stage(std::integral_constant<int, 0>{}); stage(std::integral_constant<int, 1>{}); stage(std::integral_constant<int, 2>{});
        stage(std::integral_constant<int, 3>{}); stage(std::integral_constant<int, 4>{}); stage(std::integral_constant<int, 5>{});
        stage(std::integral_constant<int, 6>{}); stage(std::integral_constant<int, 7>{}); stage(std::integral_constant<int, 8>{});

        if (FUSE && last_chunk) {
            // ---- the fused 1x1 (see the kernel's header) ----
            float am = 0.0f;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float bl = btab[wn * 64 + 32 * j + rl];
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const float z = __builtin_fmaf(acc[j][i][e], inv, bl);
                        const float y = fmaxf(z, z * p.slope);
                        acc[j][i][e] = y;
                        am = fmaxf(am, fabsf(y));
                    }
            }
#pragma unroll
            for (int o = 32; o; o >>= 1) am = fmaxf(am, __shfl_xor(am, o));
            if (lane == 0) btab1[64 + wave] = am;
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_s_barrier();      // (also: every wave has finished reading the patch -- its region becomes the y scratch)
            const float tmax = fmaxf(fmaxf(btab1[64], btab1[65]), fmaxf(btab1[66], btab1[67]));
            const unsigned tb = __float_as_uint(tmax);
            const float f1 = dt_h2_base(tb), inv1 = p.pscale1[0] * dt_h2_base_inv(tb);
            c3_f16 acc2[2];
#pragma unroll
            for (int jn = 0; jn < 2; ++jn)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc2[jn][e] = 0.0f;
            const int prow = 32 * wave + rl;                                        // this lane's pixel row of the scratch as an A operand
            const int offS = prow * 32 + ((gl ^ ((prow >> 3) & 1)) * 16);
            int offB1[2];
#pragma unroll
            for (int jn = 0; jn < 2; ++jn) { const int n = 32 * jn + rl; offB1[jn] = (2 * n + (gl ^ ((n >> 3) & 1))) * 16; }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (wn == (q >> 1)) {      // this wave holds the quarter's 32 channels (its block j = q & 1): lane = channel, 16 rows per block
                    const int j = q & 1, half = rl >> 4, g = (rl >> 3) & 1;
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int pp = 64 * wm + 32 * i + 8 * (r >> 2) + 4 * gl + (r & 3);
                            const float x = acc[j][i][r] * f1;
                            const _Float16 hi = (_Float16)x;
                            const _Float16 lo = (_Float16)(x - (float)hi);
                            unsigned char *dst = patch + (half * 2) * SPL + pp * 32 + ((g ^ ((pp >> 3) & 1)) * 16) + (rl & 7) * 2;
                            *reinterpret_cast<_Float16 *>(dst) = hi;
                            *reinterpret_cast<_Float16 *>(dst + SPL) = lo;
                        }
                }
                c3_wait_vm<0>();                             // the quarter's weights (and everything else in flight) have landed
                __builtin_amdgcn_s_waitcnt(0xc07f);
                __builtin_amdgcn_s_barrier();
                // two stages ahead: the 1x1's quarters 2, 3, then the next unit's taps 0, 1
                if (q < 2) { fstage_issue(bufof(11 + q), q + 2, 0); fstage_issue(bufof(11 + q), q + 2, 1); }
                else if (has_next) stage_issue(bufof(11 + q), nxt.n0, nxt.chunk, q - 2);
                const unsigned char *sq = bring + bufof(9 + q) * BSTAGE;
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    c3_u4 a2[2], b2[2][2];
#pragma unroll
                    for (int t = 0; t < 2; ++t) a2[t] = *reinterpret_cast<const c3_u4 *>(patch + (hf * 2 + t) * SPL + offS);
#pragma unroll
                    for (int jn = 0; jn < 2; ++jn)
#pragma unroll
                        for (int t = 0; t < 2; ++t) b2[jn][t] = *reinterpret_cast<const c3_u4 *>(sq + (hf * 2 + t) * BPL1 + offB1[jn]);
#pragma unroll
                    for (int jn = 0; jn < 2; ++jn) c3_mfma(acc2[jn], b2[jn][1], a2[0]);
#pragma unroll
                    for (int jn = 0; jn < 2; ++jn) c3_mfma(acc2[jn], b2[jn][0], a2[1]);
#pragma unroll
                    for (int jn = 0; jn < 2; ++jn) c3_mfma(acc2[jn], b2[jn][0], a2[0]);
                }
                if (q < 3) {                                 // the scratch is rewritten for the next quarter
                    __builtin_amdgcn_s_waitcnt(0xc07f);
                    __builtin_amdgcn_s_barrier();
                }
            }
            sbase = (sbase + 13) % C3_NS;
            // the 1x1's epilogue: lane holds channel 32 jn + rl of pixel rows m of block `wave` = (wm' = wave >> 1, i' = wave & 1)
            const int yb = cur.ty0 + 4 * (wave >> 1) + 2 * (wave & 1);
#pragma unroll
            for (int jn = 0; jn < 2; ++jn) {
                const int n = 32 * jn + rl;
                const float b1 = btab1[n];
                const bool nok = n < p.N1;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = 8 * (r >> 2) + 4 * gl + (r & 3);
                    const int y = yb + (m >> 4), x = cur.tx0 + (m & 15);
                    const float z = __builtin_fmaf(acc2[jn][r], inv1, b1);
                    const float v = fmaxf(z, z * p.slope1);
                    if (nok && y < p.H && x < p.W) {
                        if (p.amax_out) out_am = fmaxf(out_am, fabsf(v));
                        __builtin_nontemporal_store(v, p.out + (long long)cur.b * p.out_bs + ((long long)y * p.W + x) * p.out_ld + n);
                    }
                }
            }
        } else if (last_chunk) {
            // ---- epilogue: lane holds channel n0 + 64 wn + 32 j + rl of rows m = 8 (r >> 2) + 4 gl + (r & 3) of block i:
            //      image row 4 wm + 2 i + (m >> 4), column m & 15 ----
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int nl = wn * 64 + 32 * j + rl;
                const float bl = btab[nl];
                const bool nok = cur.n0 + nl < p.N;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const float z = __builtin_fmaf(acc[j][i][e], inv, bl);
                        acc[j][i][e] = fmaxf(z, z * p.slope);      // LeakyReLU, 0 <= slope <= 1 (1: linear)
                    }
                    const int yb = cur.ty0 + 4 * wm + 2 * i;
                    if (!POOL) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int m = 8 * (r >> 2) + 4 * gl + (r & 3);
                            const int y = yb + (m >> 4), x = cur.tx0 + (m & 15);
                            if (nok && y < p.H && x < p.W) {
                                if (p.amax_out) out_am = fmaxf(out_am, fabsf(acc[j][i][r]));
                                __builtin_nontemporal_store(acc[j][i][r], p.out + (long long)cur.b * p.out_bs + ((long long)y * p.W + x) * p.out_ld + cur.n0 + nl);
                            }
                        }
                    } else {
                        const int H2 = p.H >> 1, W2 = p.W >> 1;
                        const int y2 = yb >> 1;
#pragma unroll
                        for (int q = 0; q < 4; ++q) {      // windows {r, r + 1, r + 8, r + 9}, r = 2 q (q < 2: columns 4 gl + 2 q; q >= 2: 8 + 4 gl + 2 (q - 2))
                            const int r = 2 * q;
                            const float mx = fmaxf(fmaxf(acc[j][i][r], acc[j][i][r + 1]), fmaxf(acc[j][i][r + 8], acc[j][i][r + 9]));
                            const int x2 = (cur.tx0 >> 1) + 4 * (q >> 1) + 2 * gl + (q & 1);
                            if (nok && y2 < H2 && x2 < W2) {
                                if (p.amax_out) out_am = fmaxf(out_am, fabsf(mx));
                                __builtin_nontemporal_store(mx, p.out2 + (((long long)cur.b * H2 + y2) * W2 + x2) * p.out2_ld + cur.n0 + nl);
                            }
                        }
                    }
                }
            }
        }
        if (!has_next) break;
        // ---- turn-over: every wave has finished reading the patch (and, at an item boundary, the bias table) ----
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_s_barrier();
        patch_store();
        if (last_chunk && nxt.n0 != cur.n0)
            for (int i = tid; i < BN; i += C3_THREADS) btab[i] = (nxt.n0 + i < p.N) ? p.bias[nxt.n0 + i] : 0.0f;
        item = nitem;
        cur = nxt;
    }
    if (p.amax_out) dt_amax_publish(p.amax_out, out_am);
}

bool conv3_h2_usable(const Conv3H2Args &a)
{
    return a.Cin % 32 == 0 && a.Cin >= 32 && a.N >= 64 && a.N % 64 == 0 && a.Np % 64 == 0 && a.in_ld % 4 == 0 && (reinterpret_cast<uintptr_t>(a.in) & 15) == 0 &&
           a.w && a.pscale && a.amax && a.bias && a.zeros && (a.out != nullptr) != (a.out2 != nullptr) && (!a.out2 || !((a.H | a.W) & 1)) &&
           (!a.w1 || (a.N == 128 && a.out && a.pscale1 && a.bias1 && a.N1 > 0 && a.N1 <= 64 && a.Np1 >= 64 && a.Np1 % 32 == 0));
}
// executed fp16 MFMA FLOPs of one launch (three products per multiply, whole tiles)
double conv3_h2_flops(const Conv3H2Args &a)
{
    const bool wide = a.N % 128 == 0;
    const int th = wide ? 8 : 16;
    const double tiles = (double)a.B * ((a.H + th - 1) / th) * ((a.W + 15) / 16);
    return 6.0 * tiles * th * 16 * (9.0 * a.Cin * a.N + (a.w1 ? (double)a.N * 64 : 0.0));      // (+ the fused 1x1: N x 64 per pixel)
}

template <int WN, bool POOL, bool FUSE = false>
static int c3_launch(hipStream_t st, const Conv3H2Args &a, int cus)
{
    constexpr int TH = 16 / WN, BN = 64 * WN;
    const size_t lds = (size_t)4 * (TH + 2) * C3_PW * 32 + (size_t)C3_NS * 4 * BN * 32 + BN * 4 + (FUSE ? 68 * 4 : 0);
    static PerDeviceOnce attr;
    if (attr.ensure(nullptr, [&](int) {
            return hipFuncSetAttribute(reinterpret_cast<const void *>(conv3_h2_kernel<WN, POOL, FUSE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess;
        }))
        return 1;
    Conv3H2Args p = a;
    p.tiles_y = (a.H + TH - 1) / TH;
    p.tiles_x = (a.W + 15) / 16;
    const long long items = (long long)a.B * p.tiles_x * p.tiles_y * ((a.N + BN - 1) / BN);
    long long grid = 2ll * cus;      // two workgroups per CU, persistent over the items
    if (grid > items) grid = items;
    hipLaunchKernelGGL((conv3_h2_kernel<WN, POOL, FUSE>), dim3((unsigned)grid), dim3(C3_THREADS), lds, st, p);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

int launch_conv3_h2(hipStream_t st, const Conv3H2Args &a)
{
    if (!conv3_h2_usable(a) || a.B <= 0 || a.H <= 0 || a.W <= 0) return 2;
    static int cu_of[64];
    static PerDeviceOnce once;
    int dev = 0;
    if (once.ensure(&dev, [&](int d) {
            int n = 0;
            if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, d) != hipSuccess || n <= 0) return 1;
            cu_of[d] = n;
            return 0;
        }))
        return 1;
    const int cus = cu_of[dev];
    const bool wide = a.N % 128 == 0, pool = a.out2 != nullptr;
    if (a.w1) return c3_launch<2, false, true>(st, a, cus);
    if (wide) return pool ? c3_launch<2, true>(st, a, cus) : c3_launch<2, false>(st, a, cus);
    return pool ? c3_launch<1, true>(st, a, cus) : c3_launch<1, false>(st, a, cus);
}

// conv_igemm.hip -- fp32 implicit-GEMM convolution on the CDNA4 matrix cores.
//
// Computes the reference's  Conv2D(k,'same',stride 1) [+ folded BatchNorm bias
// + LeakyReLU(0.1)] [+ MaxPooling2D(2,2)] block (models_detection/KerasYOLO.py:
// 279-396) and the ConvLSTM2D recurrent convolution + gate update
// (models_tracking/MultiObjDetTracker.py:176) as ONE kernel family:
//
//     Out[M = B*H*W, N = Cout] = A[M, K = k*k*Cin] * Wt[N, K]^T
//
// A is never materialised: a row m is an output pixel, column k is read straight
// from the NHWC activation tensor (zero outside the image).
//
// MI355X design notes
//   * v_mfma_f32_32x32x2_f32 (exact fp32, 157 TFLOP/s peak): 64-lane wavefront
//     tiles of 32x32; each wave owns TM x TN such tiles, 4 waves per workgroup.
//   * K is consumed in chunks of 32 floats of ONE tap, so a chunk of A is 32
//     contiguous floats per pixel (128 B, coalesced); chunk order is channel-slice
//     outer / tap inner (k = ((ci/32)*taps + tap)*32 + ci%32) so the nine shifted
//     re-reads of a slice hit L1/L2.
//   * Staging (default, DT_GLDS=1): asynchronous global->LDS DMA, 1 KiB per
//     wave-instruction, no staging registers and no ds_write.  LDS tiles are
//     unpadded [rows][32] floats; the DMA writes LDS lane-linearly, so the bank
//     conflict fix is an XOR swizzle of the 16-byte slot applied on the SOURCE
//     address and again on the fragment reads.  'same' padding is branch-free:
//     out-of-image taps read a block of zeros.
//   * One ds_read_b128 feeds four MFMA k-steps: lanes 0-31 take k-slots {0..3},
//     lanes 32-63 {4..7} of every 8 -- A and B use the same permutation of K, so
//     the contraction is unchanged.
//   * Rotated software pipeline, one barrier per chunk; every block of reads /
//     DMA issues sits after the first k-step of an MFMA group (see the main loop).
//   * XCD-aware tile order: contiguous tile ranges per XCD, column-grouped, so the
//     workgroups resident on an XCD share weight and activation panels in its L2.
//   * ORD_QUAD row order m = ((b*H/2+h2)*W/2+w2)*4 + dy*2+dx puts the four
//     pixels of a 2x2 pooling window in four consecutive accumulator registers
//     of one lane (C/D layout row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)), so
//     MaxPooling2D and tf.space_to_depth are free in the epilogue.
//   * EPI_GATES: the N axis is packed [j/32][gate][j%32]; a wave's four 32-wide
//     column tiles are the i,f,c,o pre-activations of the same 32 hidden
//     channels, so the LSTM cell update happens in registers.
//   * EPI_PARTIAL: split-K partial sums for small-M layers, combined in split
//     order by splitk_reduce_kernel (deterministic, no atomics).
// Build options kept for A/B (profiles/README.md): -DDT_GLDS=0 register staging
// (global_load -> VGPR -> ds_write, rows padded to 36 floats), -DDT_BK=16 16-deep
// chunks with three LDS stages and three workgroups per CU (same speed), and the
// timing-only -DDT_ABLATE=mask builds of tools/ablate.sh.
#include <cstdlib>

#include "dt_internal.h"

#ifndef DT_GLDS
#define DT_GLDS 1   // 1: async global->LDS DMA staging (global_load_lds), XOR-swizzled unpadded tiles
#endif              // 0: register staging (global_load -> VGPR -> ds_write), rows padded to 36 floats
#ifndef DT_BK
#define DT_BK 32    // floats of K per chunk (one barrier per chunk): 32 -> 2 LDS stages, 2 workgroups/CU;
#endif              // 16 -> 3 LDS stages of 16 KiB (48 KiB), <= 168 registers, 3 workgroups/CU
#define KCH DT_BK
#define SLOTS (KCH / 4)     // 16-byte slots per LDS row
#define RPP (64 / SLOTS)    // rows covered by one 1 KiB DMA piece (64 lanes x 16 B)
#define KKC (KCH / 8)       // fragment reads (8 k each) per chunk
#if DT_GLDS
#define LDK KCH
#define NSTAGE (DT_BK == 16 ? 3 : 2)
#define SWZ(r) (SLOTS == 8 ? (((r) >> 1) & 7) : (((r) >> 2) & 3))   // rows sharing a 256 B bank row get distinct slots
#else
#if DT_BK != 32
#error "register staging is only built for DT_BK=32"
#endif
#define LDK 36  // LDS row stride in floats (32 + 4 pad)
#define NSTAGE 2
#endif
#define WAVES_PER_SIMD (DT_BK == 16 ? 3 : 2)

#ifndef DT_DMA_AUX
#define DT_DMA_AUX 0   // cache-policy bits of the global_load_lds instructions (0 = default)
#endif
#ifdef DT_TILE_TIMING
// debug build only (tools/tile_timing.py): per-workgroup, per-tile timestamps of the persistent loop
#define DT_TT_BLOCKS 512
#define DT_TT_TILES 48
__device__ unsigned long long g_tile_times[DT_TT_BLOCKS * DT_TT_TILES * 4];
extern "C" __attribute__((visibility("default"))) int dt_debug_tile_times(unsigned long long *dst, int clear)
{
    if (clear) {
        void *p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_tile_times)) != hipSuccess) return 1;
        return hipMemset(p, 0, sizeof(unsigned long long) * DT_TT_BLOCKS * DT_TT_TILES * 4) == hipSuccess ? 0 : 1;
    }
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_tile_times), sizeof(unsigned long long) * DT_TT_BLOCKS * DT_TT_TILES * 4) == hipSuccess ? 0 : 1;
}
#define TT_STAMP(slot)                                                                                   \
    do {                                                                                                  \
        if (tid == 0 && blockIdx.x < DT_TT_BLOCKS && tt_i < DT_TT_TILES)                                  \
            g_tile_times[((size_t)blockIdx.x * DT_TT_TILES + tt_i) * 4 + (slot)] = __builtin_readcyclecounter(); \
    } while (0)
#else
#define TT_STAMP(slot) do { } while (0)
#endif
typedef __attribute__((address_space(1))) const void gptr_t;
typedef __attribute__((address_space(3))) void lptr_t;

__device__ __forceinline__ float leaky_act(float v, float slope) { return v > 0.0f ? v : v * slope; }

__device__ __forceinline__ float hard_sigmoid_f(float x)
{
    // Keras 2.x hard_sigmoid: clip(0.2*x + 0.5, 0, 1)
    float y = __fmaf_rn(0.2f, x, 0.5f);
    return fminf(fmaxf(y, 0.0f), 1.0f);
}

template <int ORDER>
__device__ __forceinline__ void decode_row(int m, int H, int W, int &b, int &h, int &w)
{
    if (ORDER == ORD_LINEAR) {
        const int hw = H * W;
        b = m / hw;
        const int r = m - b * hw;
        h = r / W;
        w = r - h * W;
    } else {
        const int q = m >> 2, d = m & 3;
        const int W2 = W >> 1, hw2 = (H >> 1) * W2;
        b = q / hw2;
        const int r = q - b * hw2;
        const int h2 = r / W2;
        const int w2 = r - h2 * W2;
        h = 2 * h2 + (d >> 1);
        w = 2 * w2 + (d & 1);
    }
}

template <int KS, int BM, int BN, int WGM, int WGN, int ORDER, int EPI, bool PERSIST = false>
__global__ __launch_bounds__(64 * WGM * WGN, (WGM * WGN > 4 ? (WGM * WGN) / 4 : WAVES_PER_SIMD)) void conv_igemm_f32(ConvArgs p)
{
    constexpr int NW = WGM * WGN;                  // wavefronts per workgroup (4, or 8 for the 256x128 tile)
    constexpr int WTM = BM / WGM, WTN = BN / WGN;  // wave tile
    constexpr int TM = WTM / 32, TN = WTN / 32;    // MFMA tiles per wave
#if DT_GLDS
    constexpr int RPASS = NW * RPP;                // rows filled per pass of the NW waves
#else
    constexpr int RPASS = NW * 8;
#endif
    constexpr int PA = BM / RPASS, PB = BN / RPASS;   // loader passes
    constexpr int TAPS = KS * KS;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *sA = smem;                       // [NSTAGE][BM][LDK]
    float *sB = smem + NSTAGE * BM * LDK;   // [NSTAGE][BN][LDK]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;

    // Tile schedule.  The launch is PERSISTENT when there are more tiles than resident workgroup
    // slots: workgroup b walks the linear indices L = b, b + gridDim.x, ... over
    // total = (row tiles x column tiles) x (batched GEMMs, the 16 Winograd positions of
    // winograd.hip), and the chunk-0 DMA of its next tile is issued inside the last chunk of the
    // current one, so neither a workgroup relaunch nor a cold prologue sits between two tiles.
    //
    // XCD-aware order.  Workgroup b is dispatched to XCD b % 8 (observed; used for speed only) and
    // gridDim.x is a multiple of 8 when persistent, so L % 8 is the XCD.  Each XCD is given one
    // CONTIGUOUS range of tile numbers (n fastest), so the column tiles that re-read the same
    // activation rows (and neighbouring row panels that share the 3x3 halo) hit the same 4 MiB L2
    // instead of being fetched by all eight.
    const int ntn = (p.N + BN - 1) / BN;
    const int ntm_all = (p.M + BM - 1) / BM;
    const int ntiles = ntm_all * ntn;
    const int total = ntiles * (p.zbatch > 1 ? p.zbatch : 1);
    struct Tile {
        int z, m0, n0;
    };
    auto tile_of = [&](int L) {
        int t = L;
        if (p.xcd_remap) {
            const int q = total >> 3, r = total & 7;
            const int xcd = L & 7, j = L >> 3;
            t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;   // bijective for any total
        }
        Tile T;
        T.z = t / ntiles;
        const int bid = t - T.z * ntiles;
        // Within a problem, tiles go GN column tiles at a time over ALL row tiles (then the next GN
        // columns): the ~64 workgroups resident on an XCD then cover (64/GN) row tiles x GN column
        // tiles, which minimises weight-panel + activation-panel traffic per MFMA.
        const int gn = p.tile_gn > 0 && p.tile_gn < ntn ? p.tile_gn : ntn;
        const int per_group = gn * ntm_all;                 // tiles in a full column group
        const int grp = bid / per_group;
        const int rem = bid - grp * per_group;
        const int gw = min(gn, ntn - grp * gn);             // width of this (possibly last, narrower) group
        const int tile_m = rem / gw;
        T.m0 = tile_m * BM;
        T.n0 = (grp * gn + (rem - tile_m * gw)) * BN;
        return T;
    };

    // ---- loader set-up ----------------------------------------------------
    // Register staging: thread -> row lr + 32*i, 16-byte slot tid&7 of the 128-byte chunk row.
    // DMA staging (global_load_lds writes LDS at wave-uniform base + lane*16): wave w, pass i
    // fills the 8 consecutive rows of group g = 4*i + w; lane -> row g*8 + lane/8, PHYSICAL
    // slot lane&7, and loads the LOGICAL slot (lane&7) ^ ((row>>1)&7) from global memory, i.e.
    // the XOR swizzle is applied on the source side and again on the fragment reads, which makes
    // the unpadded 128-byte rows conflict-free for ds_read_b128.
#if DT_GLDS
    const int lr = wave * RPP + lane / SLOTS;              // + RPASS*i
    const int lc = ((lane % SLOTS) ^ SWZ(lr)) * 4;         // SWZ(row) does not depend on i (RPASS*i)
#else
    const int lr = tid >> 3;        // 0..31 row within pass
    const int lc = (tid & 7) * 4;   // float offset within the 32-float chunk
#endif
    const float *a_ptr[PA];
    unsigned a_mask[PA];
    const float *b_ptr[PB];
    auto setup = [&](const Tile &T) {
        const float *in_z = p.in + (long long)T.z * p.z_in;
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            const int m = T.m0 + lr + RPASS * i;
            unsigned mask = 0;
            const float *ptr = in_z;
            if (m < p.M) {
                if (KS == 1 && ORDER == ORD_LINEAR) {
                    // dense NHWC input (in_bs == H*W*in_ld, checked by the launcher): row m is pixel m
                    ptr = in_z + (long long)m * p.in_ld + lc;
                    mask = 1u;
                } else {
                int b, h, w;
                decode_row<ORDER>(m, p.H, p.W, b, h, w);
                ptr = in_z + (long long)b * p.in_bs + (long long)(h * p.W + w) * p.in_ld + lc;
                if (KS == 1) {
                    mask = 1u;
                } else {
#pragma unroll
                    for (int t = 0; t < 9; ++t) {
                        const int ih = h + t / 3 - 1, iw = w + t % 3 - 1;
                        if (ih >= 0 && ih < p.H && iw >= 0 && iw < p.W) mask |= 1u << t;
                    }
                }
                }
            }
            a_ptr[i] = ptr;
            a_mask[i] = mask;
        }
        const float *wt_z = p.wt + (long long)T.z * p.z_wt;
#pragma unroll
        for (int i = 0; i < PB; ++i) b_ptr[i] = wt_z + (long long)(T.n0 + lr + RPASS * i) * p.K + lc;
    };
    int Lcur = blockIdx.x;
    Tile cur_t = tile_of(Lcur);
    setup(cur_t);

    const int cpt = p.Cin / KCH;    // K chunks per tap
    // split-K: blockIdx.y owns chunks [k0, k0+nk) of the TAPS*cpt chunks of K
    const int nk_all = TAPS * cpt;
    const int k0 = (int)(((long long)nk_all * blockIdx.y) / gridDim.y);
    const int nk = (int)(((long long)nk_all * (blockIdx.y + 1)) / gridDim.y) - k0;

#if DT_GLDS
    // one 1 KiB DMA piece per (wave, pass): 8 rows x 128 B, LDS destination wave-uniform
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    auto dma = [&](int buf, int tap, int cc, int kc) {
        int aoff;
        if (KS == 1)
            aoff = cc * KCH;
        else
            aoff = ((tap / 3 - 1) * p.W + (tap % 3 - 1)) * p.in_ld + cc * KCH;
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            const bool ok = (a_mask[i] >> tap) & 1u;
            const float *src = ok ? a_ptr[i] + aoff : p.zeros;   // branch-free 'same' padding
            float *dst = sA + (buf * BM + (NW * i + wave_u) * RPP) * LDK;
            __builtin_amdgcn_global_load_lds((gptr_t *)src, (lptr_t *)dst, 16, 0, DT_DMA_AUX);
        }
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            float *dst = sB + (buf * BN + (NW * i + wave_u) * RPP) * LDK;
            __builtin_amdgcn_global_load_lds((gptr_t *)(b_ptr[i] + kc * KCH), (lptr_t *)dst, 16, 0, DT_DMA_AUX);
        }
    };
#endif
    f32x4 ra[PA], rb[PB];
    auto gload = [&](int tap, int cc, int kc) {
        int aoff;
        if (KS == 1)
            aoff = cc * 32;
        else
            aoff = ((tap / 3 - 1) * p.W + (tap % 3 - 1)) * p.in_ld + cc * 32;
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            // branch-free 'same' padding: out-of-image taps read a 16-byte block of zeros
            const bool ok = (a_mask[i] >> tap) & 1u;
            const float *src = ok ? a_ptr[i] + aoff : p.zeros;
            ra[i] = *reinterpret_cast<const f32x4 *>(src);
        }
#pragma unroll
        for (int i = 0; i < PB; ++i) rb[i] = *reinterpret_cast<const f32x4 *>(b_ptr[i] + kc * 32);
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < PA; ++i)
            *reinterpret_cast<f32x4 *>(&sA[(buf * BM + lr + 32 * i) * LDK + lc]) = ra[i];
#pragma unroll
        for (int i = 0; i < PB; ++i)
            *reinterpret_cast<f32x4 *>(&sB[(buf * BN + lr + 32 * i) * LDK + lc]) = rb[i];
    };

    f32x16 acc[TM][TN];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    };
    zero_acc();

    // ---- main loop ----------------------------------------------------------
    // Rotated software pipeline, one barrier per 32-deep K chunk:
    //   MFMA(kk=0) || read frags kk=1, write chunk t+1 to the other LDS buffer
    //   MFMA(kk=1) || read frags kk=2, issue global loads of chunk t+2
    //   MFMA(kk=2) || read frags kk=3
    //   barrier
    //   MFMA(kk=3) || read frags kk=0 of chunk t+1
    // so LDS/global instructions always issue under a queue of 16 independent
    // 64-cycle MFMAs and the matrix pipe only idles for barrier skew.
    const int fr = lane & 31;          // fragment row within a 32-row tile
    const int fk = (lane >> 5) * 4;    // k sub-slot: lanes 0-31 -> 0..3, 32-63 -> 4..7
    (void)fk;
    struct Frag {
        f32x4 a[TM], b[TN];
    };
    auto lfrag = [&](Frag &f, int buf, int kk) {
#if DT_GLDS
        const int ko = (((kk * 2 + (lane >> 5)) ^ SWZ(fr)) * 4);   // swizzled 16-byte slot
        const float *cA = sA + (buf * BM + wm * WTM + fr) * LDK + ko;
        const float *cB = sB + (buf * BN + wn * WTN + fr) * LDK + ko;
#else
        const float *cA = sA + (buf * BM + wm * WTM + fr) * LDK + fk + kk * 8;
        const float *cB = sB + (buf * BN + wn * WTN + fr) * LDK + fk + kk * 8;
#endif
#pragma unroll
        for (int i = 0; i < TM; ++i) f.a[i] = *reinterpret_cast<const f32x4 *>(cA + i * 32 * LDK);
#pragma unroll
        for (int j = 0; j < TN; ++j) f.b[j] = *reinterpret_cast<const f32x4 *>(cB + j * 32 * LDK);
    };
    // one fragment = four MFMA k-steps per tile; `mma_part<lo,hi>` issues steps [lo,hi)
    auto mma_part = [&](const Frag &f, int lo, int hi) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
            if (s >= lo && s < hi)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[i][s], f.b[j][s], acc[i][j], 0, 0, 0);
    };
    auto mma = [&](const Frag &f) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[i][s], f.b[j][s], acc[i][j], 0, 0, 0);
    };

    // K order is channel-chunk OUTER, tap INNER (k = (cc*TAPS + tap)*32 + c): the nine taps of one
    // 32-channel slice are consumed back to back, so their shifted re-reads of the same activation
    // rows hit L1/L2 instead of being re-fetched from the Infinity Cache nine chunk-rows apart.
    int gtap = k0 % TAPS, gcc = k0 / TAPS;   // (tap, chunk) of the NEXT tile to fetch from global memory
    int gk = k0;                           // its absolute chunk index (B column offset)
    const int k_end = k0 + nk;
    auto gadvance = [&]() {
        ++gk;
        if (++gtap == TAPS) { gtap = 0; ++gcc; }
    };
    const int last_tap = (k_end - 1) % TAPS, last_cc = (k_end - 1) / TAPS;
#ifndef DT_ABLATE
#define DT_ABLATE 0   // timing-only ablation builds (tools/ablate.sh): results are WRONG when != 0
#endif
    constexpr bool AB_BARRIER = (DT_ABLATE & 1) != 0, AB_GLOAD = (DT_ABLATE & 2) != 0;
    constexpr bool AB_LSTORE = (DT_ABLATE & 4) != 0, AB_LFRAG = (DT_ABLATE & 8) != 0;
    // ---- epilogue (per tile) --------------------------------------------------
    const int hi = lane >> 5;
    float out_am = 0.0f;      // the largest |value| this lane stored (ConvArgs::amax_out), over all its tiles
    auto epilogue = [&](const Tile &T) {
        const int m0 = T.m0, n0 = T.n0;
        float *outz = p.out + (long long)T.z * p.z_out;
        if (EPI == EPI_GATES) {
            // TN == 4: tile j = gate (i,f,c,o) of hidden channel jc
            static_assert(EPI != EPI_GATES || TN == 4, "gates need 4 column tiles per wave");
            const int colbase = n0 + wn * WTN;            // multiple of 128
            const int jc = (colbase >> 2) + fr;           // hidden channel
            const int hw = p.H * p.W;
    #pragma unroll
            for (int i = 0; i < TM; ++i)
    #pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    if (row < p.M) {
                        const int b = row / hw;
                        const int pix = row - b * hw;
                        const float *xp = p.xproj + (long long)b * p.xp_bs + (long long)pix * p.xp_ld + colbase + fr;
                        const float zi = acc[i][0][r] + xp[0];
                        const float zf = acc[i][1][r] + xp[32];
                        const float zc = acc[i][2][r] + xp[64];
                        const float zo = acc[i][3][r] + xp[96];
                        float *cp = p.cstate + (long long)b * p.c_bs + (long long)pix * p.c_ld + jc;
                        const float gi = hard_sigmoid_f(zi), gf = hard_sigmoid_f(zf), go = hard_sigmoid_f(zo);
                        const float cn = gf * (*cp) + gi * tanhf(zc);
                        *cp = cn;
                        outz[(long long)b * p.out_bs + (long long)pix * p.out_ld + jc] = go * tanhf(cn);
                    }
                }
            return;
        }
        // Raw GEMM output (the batched Winograd GEMMs: no bias, linear): the epilogue of a 256x256 tile is VALU-issue
        // bound (tools/tile_timing.py: 6 us), so skip the activation arithmetic and walk the rows with one pointer
        if (EPI == EPI_PLAIN && PERSIST && p.bias == nullptr && p.slope == 1.0f && p.act == 0) {
    #pragma unroll
            for (int i = 0; i < TM; ++i)
    #pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int col = n0 + wn * WTN + j * 32 + fr;
                    if (col >= p.N) continue;
                    const int r0 = m0 + wm * WTM + i * 32 + 4 * hi;
                    float *o = outz + (long long)r0 * p.out_ld + col;
    #pragma unroll
                    for (int g = 0; g < 4; ++g) {
    #pragma unroll
                        for (int q = 0; q < 4; ++q)
                            if (r0 + 8 * g + q < p.M) o[(long long)(8 * g + q) * p.out_ld] = acc[i][j][4 * g + q];
                    }
                }
            return;
        }
    #pragma unroll
        for (int i = 0; i < TM; ++i)
    #pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = n0 + wn * WTN + j * 32 + fr;
                const bool cok = col < p.N;
                const float bv = (p.bias != nullptr) ? p.bias[col] : 0.0f;  // bias is padded to Npad
    #pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int rbase = m0 + wm * WTM + i * 32 + 8 * g + 4 * hi;
                    float v[4];
    #pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float z = acc[i][j][4 * g + q] + bv;
                        v[q] = p.act == 1 ? 1.0f / (1.0f + expf(-z)) : leaky_act(z, p.slope);   // Dense(sigmoid) / LeakyReLU
                    }
                    if (!cok || rbase >= p.M) continue;
                    if (EPI == EPI_PARTIAL) {
                        // split-K partial sums: raw accumulators to slab [split][M][N]
                        float *slab = outz + (long long)blockIdx.y * p.M * p.out_ld;
    #pragma unroll
                        for (int q = 0; q < 4; ++q)
                            if (rbase + q < p.M) slab[(long long)(rbase + q) * p.out_ld + col] = acc[i][j][4 * g + q];
                    } else if (EPI == EPI_PLAIN) {
    #pragma unroll
                        for (int q = 0; q < 4; ++q)
                            if (rbase + q < p.M) { out_am = fmaxf(out_am, fabsf(v[q])); outz[(long long)(rbase + q) * p.out_ld + col] = v[q]; }
                    } else if (EPI == EPI_S2D) {
                        // tf.space_to_depth(2): pixel m>>2, channel (m&3)*N + col
    #pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            out_am = fmaxf(out_am, fabsf(v[q]));
                            outz[(long long)(rbase >> 2) * p.out_ld + q * p.N + col] = v[q];
                        }
                    } else {
                        const float mx = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
                        out_am = fmaxf(out_am, fabsf(mx));      // (EPI_POOL_BOTH: the pooled tensor's maximum -- the one the next layer reads)
                        if (EPI == EPI_POOL) {
                            outz[(long long)(rbase >> 2) * p.out_ld + col] = mx;
                        } else {  // EPI_POOL_BOTH: out = unpooled (standard NHWC), out2 = pooled
                            p.out2[(long long)(rbase >> 2) * p.out2_ld + col] = mx;
                            int b, h, w;
                            decode_row<ORD_QUAD>(rbase, p.H, p.W, b, h, w);
                            float *o = outz + ((long long)(b * p.H + h) * p.W + w) * p.out_ld + col;
                            o[0] = v[0];
                            o[p.out_ld] = v[1];
                            o[(long long)p.W * p.out_ld] = v[2];
                            o[(long long)(p.W + 1) * p.out_ld] = v[3];
                        }
                    }
                }
            }
    };
    Frag f0, f1;
#if DT_GLDS && DT_BK == 16
    // 16-deep chunks, THREE 16 KiB LDS stages (48 KiB -> three workgroups per CU, three waves per
    // SIMD within 168 registers).  The DMA of chunk t+2 is issued while chunk t is multiplied and
    // stays in flight across the barrier (raw s_barrier + counted vmcnt: only chunk t+1's pieces
    // are waited for), so every DMA has a whole chunk of MFMAs of cover.
    (void)ra; (void)rb; (void)gload; (void)lstore; (void)AB_LSTORE;
    constexpr int NPIECE = PA + PB;                       // DMA pieces per wave per chunk
    static_assert(NPIECE >= 1 && NPIECE <= 15, "vmcnt(NPIECE) must fit the 4 low bits of the immediate");
    constexpr int WAIT_PREV = 0x0f70 | NPIECE;            // s_waitcnt vmcnt(NPIECE): all but the newest chunk landed
    dma(0, gtap, gcc, gk);
    gadvance();
    {
        const bool in = gk < k_end;
        dma(1, in ? gtap : last_tap, in ? gcc : last_cc, in ? gk : k_end - 1);
        gadvance();
    }
    __builtin_amdgcn_s_waitcnt(WAIT_PREV);   // chunk 0 landed, chunk 1 may still fly
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    lfrag(f0, 0, 0);
    int b_cur = 0, b_nxt = 1, b_nn = 2;
#define SB() __builtin_amdgcn_sched_barrier(0)
    for (int kc = 0; kc < nk; ++kc) {
        mma_part(f0, 0, 1);
        SB();
        if (!AB_LFRAG) lfrag(f1, b_cur, 1);
        if (!AB_GLOAD) {
            const bool in = gk < k_end;          // past the end: re-fetch the last chunk into a dead buffer
            dma(b_nn, in ? gtap : last_tap, in ? gcc : last_cc, in ? gk : k_end - 1);
            gadvance();
        }
        __builtin_amdgcn_sched_barrier(0x16);
        mma_part(f0, 1, 4);
        SB();
        __builtin_amdgcn_s_waitcnt(WAIT_PREV);  // everything but the newest chunk has landed -> chunk t+1 ready
        __builtin_amdgcn_s_waitcnt(0xc07f);    // lgkmcnt(0): my reads of buffer b_cur are done (it is refilled next chunk)
        if (!AB_BARRIER) __builtin_amdgcn_s_barrier();
        SB();
        mma_part(f1, 0, 1);
        SB();
        if (!AB_LFRAG) lfrag(f0, b_nxt, 0);
        SB();
        mma_part(f1, 1, 4);
        const int t = b_cur; b_cur = b_nxt; b_nxt = b_nn; b_nn = t;
    }
#undef SB
    __builtin_amdgcn_s_waitcnt(0x0f70);    // drain the dead trailing DMAs before the LDS is released
    epilogue(cur_t);   // this staging variant is launched one tile per workgroup
#elif DT_GLDS
    // DMA variant: chunk t+1 streams straight into the other LDS buffer while chunk t is
    // multiplied; no staging registers, no ds_write.  The drain (vmcnt(0)) sits right before
    // the one barrier of the iteration, a full chunk of MFMAs after the DMA was issued.
    //
    // Every block of LDS reads / DMA issues sits AFTER the first k-step of an MFMA group and
    // BEFORE its other three (pinned with sched_barrier): hipcc waits lgkmcnt(0) ahead of the
    // first MFMA that consumes a fragment once LDS-DMA is in flight, so a read block placed in
    // front of a group would be waited for immediately; placed here it has 12 x TM x TN MFMAs of
    // cover and the wait in front of the next group finds nothing outstanding.
    //
    // Persistent tiles: the body of a tile's LAST chunk computes the loader addresses of the
    // workgroup's next tile and issues ITS chunk-0 DMA (when there is none, this tile's chunk 0
    // is re-fetched into the dead buffer, which keeps the body branch-free); the fragment read
    // after the barrier is then already the next tile's first one, and the epilogue stores of this
    // tile drain underneath the next tile's MFMAs.
    (void)ra; (void)rb; (void)gload; (void)lstore; (void)AB_LSTORE; 
    dma(0, gtap, gcc, gk);
    gadvance();
    __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0)
    __syncthreads();
    lfrag(f0, 0, 0);
#define SB() __builtin_amdgcn_sched_barrier(0)
    if constexpr (!PERSIST) {
        // one tile per workgroup
        for (int kc = 0; kc < nk; ++kc) {
            const int cur = kc & 1;
            mma_part(f0, 0, 1);
            SB();
            if (!AB_LFRAG) lfrag(f1, cur, 1);
            if (!AB_GLOAD) {
                const bool in = gk < k_end;          // past the end: re-fetch the last chunk into the dead buffer
                dma(cur ^ 1, in ? gtap : last_tap, in ? gcc : last_cc, in ? gk : k_end - 1);
                gadvance();
            }
            __builtin_amdgcn_sched_barrier(0x16);   // DS reads and MFMAs stay put; the DMA issues and their address
                                                    // VALU/SALU may sink in between the twelve MFMAs that follow
            mma_part(f0, 1, 4);
            mma_part(f1, 0, 1);
            SB();
            if (!AB_LFRAG) lfrag(f0, cur, 2);
            SB();
            mma_part(f1, 1, 4);
            mma_part(f0, 0, 1);
            SB();
            if (!AB_LFRAG) lfrag(f1, cur, 3);
            SB();
            mma_part(f0, 1, 4);
            SB();
            __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0): this wave's DMA pieces of chunk t+1 have landed
            if (!AB_BARRIER) __syncthreads();
            SB();
            mma_part(f1, 0, 1);
            SB();
            if (!AB_LFRAG) lfrag(f0, cur ^ 1, 0);
            SB();
            mma_part(f1, 1, 4);
        }
        if (AB_LFRAG) {   // keep the fragments formally live
            lfrag(f1, 0, 1);
            mma(f1);
        }
        epilogue(cur_t);
    } else {
        int cur = 0;
#ifdef DT_TILE_TIMING
        int tt_i = 0;
#endif
        for (;;) {
            TT_STAMP(0);
            for (int kc = 0; kc < nk - 1; ++kc) {
                mma_part(f0, 0, 1);
                SB();
                if (!AB_LFRAG) lfrag(f1, cur, 1);
                if (!AB_GLOAD) {
                    dma(cur ^ 1, gtap, gcc, gk);
                    gadvance();
                }
                __builtin_amdgcn_sched_barrier(0x16);   // DS reads and MFMAs stay put; the DMA issues and their address
                                                        // VALU/SALU may sink in between the twelve MFMAs that follow
                mma_part(f0, 1, 4);
                mma_part(f1, 0, 1);
                SB();
                if (!AB_LFRAG) lfrag(f0, cur, 2);
                SB();
                mma_part(f1, 1, 4);
                mma_part(f0, 0, 1);
                SB();
                if (!AB_LFRAG) lfrag(f1, cur, 3);
                SB();
                mma_part(f0, 1, 4);
                SB();
                __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0): this wave's DMA pieces of chunk t+1 have landed
                if (!AB_BARRIER) __syncthreads();
                SB();
                mma_part(f1, 0, 1);
                SB();
                if (!AB_LFRAG) lfrag(f0, cur ^ 1, 0);
                SB();
                mma_part(f1, 1, 4);
                cur ^= 1;
            }
            TT_STAMP(1);
            // last chunk of this tile: same body, the DMA belongs to the next tile
            const int Lnext = Lcur + (int)gridDim.x;
            const bool more = Lnext < total;
            const Tile next_t = tile_of(more ? Lnext : Lcur);
            mma_part(f0, 0, 1);
            SB();
            if (!AB_LFRAG) lfrag(f1, cur, 1);
            setup(next_t);
            gtap = k0 % TAPS; gcc = k0 / TAPS; gk = k0;
            if (!AB_GLOAD) {
                dma(cur ^ 1, gtap, gcc, gk);
                gadvance();
            }
            __builtin_amdgcn_sched_barrier(0x16);
            mma_part(f0, 1, 4);
            mma_part(f1, 0, 1);
            SB();
            if (!AB_LFRAG) lfrag(f0, cur, 2);
            SB();
            mma_part(f1, 1, 4);
            mma_part(f0, 0, 1);
            SB();
            if (!AB_LFRAG) lfrag(f1, cur, 3);
            SB();
            mma_part(f0, 1, 4);
            SB();
            __builtin_amdgcn_s_waitcnt(0x0f70);
            if (!AB_BARRIER) __syncthreads();
            SB();
            mma_part(f1, 0, 1);
            SB();
            if (!AB_LFRAG) lfrag(f0, cur ^ 1, 0);
            SB();
            mma_part(f1, 1, 4);
            cur ^= 1;
            if (AB_LFRAG) {   // keep the fragments formally live
                lfrag(f1, 0, 1);
                mma(f1);
            }
            TT_STAMP(2);
            epilogue(cur_t);
            TT_STAMP(3);
#ifdef DT_TILE_TIMING
            ++tt_i;
#endif
            if (!more) break;
            Lcur = Lnext;
            cur_t = next_t;
            zero_acc();
        }
    }
#undef SB
#else
    gload(gtap, gcc, gk);
    gadvance();
    lstore(0);
    if (nk > 1) { gload(gtap, gcc, gk); gadvance(); }
    __syncthreads();
    lfrag(f0, 0, 0);
    // The body is branch-free so that the scheduler can interleave it with the MFMAs:
    // past the end of K the loads re-read the last chunk and the LDS traffic goes to
    // the buffer nobody reads again.
    for (int kc = 0; kc < nk; ++kc) {
        const int cur = kc & 1;
        if (!AB_LFRAG) lfrag(f1, cur, 1);
        if (!AB_LSTORE) lstore(cur ^ 1);                // chunk kc+1 (in registers since last iteration)
        mma(f0);
        if (!AB_LFRAG) lfrag(f0, cur, 2);
        if (!AB_GLOAD) {
            const bool in = gk < k_end;
            gload(in ? gtap : last_tap, in ? gcc : last_cc, in ? gk : k_end - 1);
            gadvance();
        }
        mma(f1);
        if (!AB_LFRAG) lfrag(f1, cur, 3);
        mma(f0);
        if (!AB_BARRIER) __syncthreads();
        __builtin_amdgcn_sched_barrier(0);   // keep the kk=3 MFMAs BELOW the barrier: they cover the reads that follow it
        if (!AB_LFRAG) lfrag(f0, cur ^ 1, 0);
        __builtin_amdgcn_sched_barrier(0);   // ...and keep those reads ABOVE them
        mma(f1);
    }
    epilogue(cur_t);   // this staging variant is launched one tile per workgroup
#endif
    if ((EPI == EPI_PLAIN || EPI == EPI_POOL || EPI == EPI_POOL_BOTH || EPI == EPI_S2D) && p.amax_out) dt_amax_publish(p.amax_out, out_am);
}

template <int KS, int BM, int BN, int WGM, int WGN, int ORDER, int EPI, bool PERSIST = false>
static int launch_one(hipStream_t st, const ConvArgs &a, int ksplit = 1)
{
    const int ntm = (a.M + BM - 1) / BM, ntn = (a.N + BN - 1) / BN;
    const size_t lds = (size_t)NSTAGE * (BM + BN) * LDK * sizeof(float);
    auto kern = conv_igemm_f32<KS, BM, BN, WGM, WGN, ORDER, EPI, PERSIST>;
    static PerDeviceOnce attr;
    if (attr.ensure(nullptr, [&](int) {
            return hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess;
        }))
        return 1;
    // one workgroup per tile, or -- with more tiles than resident slots -- a persistent grid of one
    // workgroup per slot (256 CUs x 2 four-wave workgroups, or x 1 of the 8/16-wave ones) that walks the tiles
    int grid = ntm * ntn * (a.zbatch > 1 ? a.zbatch : 1);
#if DT_GLDS && DT_BK == 32
    static const int cus = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
            n = 256;
        return n;
    }();
    const int slots = ((cus * (WGM * WGN > 4 ? 1 : 2)) / 8) * 8;   // multiple of 8: L % 8 stays the XCD
    if (PERSIST && !a.no_persist && ksplit == 1 && slots > 0 && grid > slots) grid = slots;
#endif
    hipLaunchKernelGGL(kern, dim3((unsigned)grid, (unsigned)ksplit), dim3(64 * WGM * WGN), lds, st, a);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

template <int KS, int ORDER, int EPI>
static int launch_cfg(hipStream_t st, const ConvArgs &a, int cfg)
{
    // The GEMM-shaped launches (1x1 layers and the batched Winograd GEMMs: short K, many tiles) use the
    // persistent form of the kernel; the 3x3 layers keep one tile per workgroup (long K per tile, and the
    // extra live state of the tile loop costs them registers).
    constexpr bool P = (KS == 1 && ORDER == ORD_LINEAR && EPI == EPI_PLAIN) && DT_GLDS && DT_BK == 32;
    if (cfg == CFG_128x64) return launch_one<KS, 128, 64, 4, 1, ORDER, EPI, P>(st, a);
    if (cfg == CFG_64x128) return launch_one<KS, 64, 128, 2, 2, ORDER, EPI, P>(st, a);       // few rows (a handful of Winograd tiles)
    if (cfg == CFG_256x128) return launch_one<KS, 256, 128, 4, 2, ORDER, EPI, P>(st, a);   // 8 waves, 1 workgroup per CU
    if (cfg == CFG_256x256) return launch_one<KS, 256, 256, 4, 4, ORDER, EPI, P>(st, a);   // 16 waves, 1 workgroup per CU
    return launch_one<KS, 128, 128, 2, 2, ORDER, EPI, P>(st, a);
}

int launch_conv_igemm(hipStream_t st, const ConvArgs &a_in, int ks, int order, int epi, int cfg)
{
    ConvArgs a = a_in;
    // a.xcd_remap / a.no_persist / a.gn_default come from the caller's Policy (network.hip:launch_igemm)
    // column tiles per group of the tile order: measured FETCH_SIZE optimum is 1 for the 3x3
    // layers (all ~64 workgroups resident on an XCD share ONE weight panel) and 2 for 1x1 / gates
    if (a_in.tile_gn < 0) a.tile_gn = -a_in.tile_gn - 1;   // caller-chosen width, encoded as -(gn+1)
    else a.tile_gn = a_in.gn_default > 0 ? a_in.gn_default - 1 : ((ks == 3 && epi != EPI_GATES) ? 1 : 2);
    // a caller-forced tile configuration (Policy::conv_cfg: A/B runs and the tests that exercise every
    // configuration at small shapes) applies to the 128-wide-or-wider layers
    if (a.force_cfg > 0 && cfg != CFG_128x64 && cfg != CFG_64x128 && epi != EPI_PARTIAL) cfg = a.force_cfg - 1;
    if (cfg == CFG_256x256) {   // the column tile may only read weight rows that exist
        const int have = a.npad ? a.npad : (a.N + 127) / 128 * 128;
        if (have < (a.N + 255) / 256 * 256) cfg = CFG_256x128;
    }
    static float *zeros_dev[64];   // per device: 256 B of zeros for the padding taps
    static PerDeviceOnce zeros_once;
    int dev = 0;
    if (zeros_once.ensure(&dev, [&](int d) {
            float *z = nullptr;
            if (hipMalloc(reinterpret_cast<void **>(&z), 256) != hipSuccess) return 1;
            if (hipMemset(z, 0, 256) != hipSuccess) { (void)hipFree(z); return 1; }
            zeros_dev[d] = z;
            return 0;
        }))
        return 1;
    a.zeros = zeros_dev[dev];
    if (a.Cin % KCH != 0 || a.K != ks * ks * a.Cin) return 2;
    if (ks == 1 && order == ORD_LINEAR && a.in_bs != (long long)a.H * a.W * a.in_ld) return 2;   // flat row addressing
    if (epi == EPI_GATES) {
        if (order != ORD_LINEAR) return 2;
        if (ks == 3 && cfg == CFG_256x256) return launch_one<3, 256, 256, 8, 2, ORD_LINEAR, EPI_GATES>(st, a);   // 16 waves
        if (ks == 3) return launch_one<3, 128, 128, 4, 1, ORD_LINEAR, EPI_GATES>(st, a);
        return launch_one<1, 128, 128, 4, 1, ORD_LINEAR, EPI_GATES>(st, a);
    }
    if (epi == EPI_PARTIAL) {
        if (order != ORD_LINEAR || a.ksplit < 1) return 2;
        return ks == 3 ? launch_one<3, 128, 128, 2, 2, ORD_LINEAR, EPI_PARTIAL>(st, a, a.ksplit)
                       : launch_one<1, 128, 128, 2, 2, ORD_LINEAR, EPI_PARTIAL>(st, a, a.ksplit);
    }
    if (order == ORD_LINEAR) {
        if (epi != EPI_PLAIN) return 2;
        return ks == 3 ? launch_cfg<3, ORD_LINEAR, EPI_PLAIN>(st, a, cfg) : launch_cfg<1, ORD_LINEAR, EPI_PLAIN>(st, a, cfg);
    }
    // ORD_QUAD
    switch (epi) {
    case EPI_POOL:
        return ks == 3 ? launch_cfg<3, ORD_QUAD, EPI_POOL>(st, a, cfg) : launch_cfg<1, ORD_QUAD, EPI_POOL>(st, a, cfg);
    case EPI_POOL_BOTH:
        return ks == 3 ? launch_cfg<3, ORD_QUAD, EPI_POOL_BOTH>(st, a, cfg) : 2;
    case EPI_S2D:
        return ks == 1 ? launch_cfg<1, ORD_QUAD, EPI_S2D>(st, a, cfg) : 2;
    default:
        return 2;
    }
}

// split-K combine: out[row][col] = act(sum_s slab[s][row][col] + bias[col]); the
// sum runs in split order, so the result is deterministic.
__global__ void splitk_reduce_kernel(const float *slab, int S, long long M, int N, const float *bias, float slope,
                                     float *out, int out_ld)
{
    const long long total = M * N;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (long long)gridDim.x * blockDim.x) {
        const long long row = e / N;
        const int col = (int)(e - row * N);
        float v = slab[e];
        for (int s = 1; s < S; ++s) v += slab[(long long)s * total + e];
        v += bias ? bias[col] : 0.0f;
        out[row * out_ld + col] = leaky_act(v, slope);
    }
}

int launch_splitk_reduce(hipStream_t st, const float *slab, int S, long long M, int N, const float *bias, float slope,
                         float *out, int out_ld)
{
    const long long total = M * N;
    long long blocks = (total + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, st, slab, S, M, N, bias, slope, out,
                       out_ld);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

void pack_conv_weights(const float *hwio, int ks, int cin_src, int cout_src, const int *cin_map, int cin_dst,
                       const int *n_map, int npad, const float *scale, float *dst)
{
    const int taps = ks * ks;
    const size_t K = (size_t)taps * cin_dst;
    for (int n = 0; n < npad; ++n) {
        float *row = dst + (size_t)n * K;
        const int ns = n_map ? n_map[n] : (n < cout_src ? n : -1);
        if (ns < 0) {
            for (size_t k = 0; k < K; ++k) row[k] = 0.0f;
            continue;
        }
        const float sc = scale ? scale[ns] : 1.0f;
        // k = ((ci/KCH)*taps + t)*KCH + ci%KCH  (channel-chunk outer, tap inner; cin_dst % KCH == 0)
        for (int t = 0; t < taps; ++t)
            for (int ci = 0; ci < cin_dst; ++ci) {
                const int cs = cin_map ? cin_map[ci] : (ci < cin_src ? ci : -1);
                row[((size_t)(ci / KCH) * taps + t) * KCH + (ci % KCH)] =
                    cs < 0 ? 0.0f : hwio[((size_t)t * cin_src + cs) * cout_src + ns] * sc;
            }
    }
}

// wino4b_fused.hip -- fused Winograd F(4x4,3x3) for the early wide 3x3 layers ON THE BF16 MATRIX PIPE at fp32 accuracy
// (conv_2: 32 -> 64 at 208x208 (+pool), conv_3 / conv_5: 64 -> 128 at 104x104; models_detection/KerasYOLO.py:285-320).
// V and M' never leave the CU; the multiplies run as three-term bf16 splits (wino_gemm_s3.hip's arithmetic).  DESIGN.md 4.5.
//
// Why not the round-2..4 kernel (wino4s_fused.hip, v_mfma_f32_16x16x4_f32, 36 accumulators per (tile, channel) pair): its item is
// 32 tiles x 64 channels because 36 accumulators x 2048 pairs are all the registers a CU has, so a stage re-streams 36.9 KB of U
// for 32 tiles, and on the bf16 pipe (K >= 16 per instruction) one K step of all 36 positions is 221 KB of U -- no LDS stage.
// This kernel turns both around:
//
//   * the output transform is LINEAR, so it is accumulated:  Y += At[:, r] (x) (M'[r][0..5] A)  after every position ROW r of every
//     16-channel slice.  A (tile, channel) pair then owns 16 fp32 (its 4x4 outputs) instead of 36, positions stream through in
//     half-rows of three, and a CU holds 64 tiles x 64 channels: HALF the U traffic per tile, U and V stages of 18 KB;
//   * one v_mfma_f32_16x16x32_bf16 carries TWO of the six partial products of the split: its K = 32 is 16 input channels x 2 term
//     slots,  A = (v_a | v_b), B = (u_x | u_y)  ->  sum_c v_a[c] u_x[c] + v_b[c] u_y[c].  Three instructions per (position, block):
//         (v1|v3).(u3|u1)   (v2|v1).(u1|u2)   (v1|v2).(u1|u2)         = u3v1 + u1v3,  u1v2 + u2v1,  u1v1 + u2v2      (smallest first)
//     -- the matrix pipe at full rate on 16-channel slices, so the input patch of a slice (34 x 34 pixels x 16 channels, 78 KB)
//     fits LDS next to the U / V rings;
//   * workgroup = 8 waves = 4 (16 tiles each) x 2 (32 channels each: two 16x16 blocks sharing the A fragments); per wave 128 VGPRs
//     of Y, 48 of M' for the row in flight.  Two waves per SIMD in different phases: one feeds the matrix pipe while the other does
//     the input transform (set 1 at the head of a stage, set 0 at its tail), the accumulated output transform follows the MFMAs of
//     every second stage;
//   * every wave produces V for ITS tile row of the block (lane = tile x channel pair): window reads as 8-byte LDS reads from a
//     patch image whose pixel index is XOR-swizzled in groups of four (tiles 256 B apart would share banks), the six positions of
//     a row in two halves, split into bf16 terms in registers, 4-byte stores into the A-operand image;
//   * U arrives pre-split and pre-arranged (wino4b_fused_pack) by LDS-DMA, the patch by LDS-DMA too: even patch rows (all that
//     position row 0 reads, dead after row 4) are re-filled for the next slice during stages 9-10, odd rows (dead after row 5)
//     during stages 11 and 0 -- one patch buffer, no bubble;
//   * persistent over items = (8x8-tile block, 64-channel slice); the stage pipeline runs across slices and items.
//
// LDS: patch 2 x 39 KiB | U 2 x 18 KiB | V 2 x 18 KiB = 153,600 B.  fp32 in, fp32 out; products as in wino_gemm_s3.hip
// (|error| at the level of an fp32 product's own rounding), sums in fp32 in another order than wino4s_fused.hip's.
#include <type_traits>
#include <utility>

#include "dt_internal.h"

typedef __attribute__((address_space(1))) const void b4_gptr_t;
typedef __attribute__((address_space(3))) void b4_lptr_t;
typedef float b4_f2 __attribute__((ext_vector_type(2)));
typedef __bf16 b4_bf8 __attribute__((ext_vector_type(8)));
typedef __bf16 b4_bf2 __attribute__((ext_vector_type(2)));
typedef unsigned b4_u4 __attribute__((ext_vector_type(4)));

#ifndef B4_NB
#define B4_NB 2                          // 16-channel blocks per wave: 2 = eight waves per workgroup (default: 10.5 / 14.5 / 13.5 ms), 1 = sixteen (11.0 / 15.0 / 14.0)
#endif
#define B4_ROWPITCH 2304                 // bytes per patch row: 36 pixels x 4 slots x 16 B (34 pixels used; the XOR stays inside groups of 4)
#define B4_CLS_PIECES 39                 // 17 rows x 144 slots = 2448 slots -> 39 DMA pieces of 64 slots
#define B4_CLS_BYTES (B4_CLS_PIECES * 1024)
#define B4_PATCH_BYTES (2 * B4_CLS_BYTES)
#define B4_POS_BYTES 6144                // one position of one operand: 3 terms x 64 rows x 16 channels x 2 B
#define B4_STAGE_BYTES (3 * B4_POS_BYTES)
#define B4_LDS_BYTES (B4_PATCH_BYTES + 4 * B4_STAGE_BYTES)

#ifndef B4_ABLATE
#define B4_ABLATE 0      // timing-only probes (results WRONG): 1 no U DMA, 2 no patch DMA, 4 no input transform, 8 no MFMA operand reads, 16 no Y accumulation
#endif

__device__ __forceinline__ unsigned b4_cvt2(const b4_f2 x) { return __builtin_bit_cast(unsigned, __builtin_convertvector(x, b4_bf2)); }
__device__ __forceinline__ b4_f2 b4_up2(unsigned u) { return b4_f2{__uint_as_float(u << 16), __uint_as_float(u & 0xffff0000u)}; }

// Bt row R (F(4x4,3x3)) applied to six values; only the inputs with a non-zero coefficient are read
template <int R>
__device__ __forceinline__ b4_f2 b4_bt(const b4_f2 (&x)[6])
{
    if (R == 0) return 4.0f * x[0] - 5.0f * x[2] + x[4];
    if (R == 1) return -4.0f * (x[1] + x[2]) + x[3] + x[4];
    if (R == 2) return 4.0f * (x[1] - x[2]) - x[3] + x[4];
    if (R == 3) return 2.0f * (x[3] - x[1]) - x[2] + x[4];
    if (R == 4) return 2.0f * (x[1] - x[3]) - x[2] + x[4];
    return 4.0f * x[1] - 5.0f * x[3] + x[5];
}
template <int R> __device__ __forceinline__ constexpr bool b4_needs(int a)
{
    return R == 0 ? (a == 0 || a == 2 || a == 4) : (R == 5 ? (a == 1 || a == 3 || a == 5) : (a >= 1 && a <= 4));
}

#ifdef DT_B4_TIMING
// debug build only (tools/b4_timing.py): per workgroup / wave / item cycle sums of the stage phases
#define B4_TT_WG 64
#define B4_TT_ITEMS 8
#define B4_TT_SLOTS 8      // 0 item start, 1 stage loop end, 2 epilogue end, 3 sum(dma issue), 4 sum(input transform), 5 sum(mfma phase), 6 sum(Y accumulation), 7 sum(vmcnt + barrier wait)
__device__ unsigned long long g_b4_times[B4_TT_WG * 16 * B4_TT_ITEMS * B4_TT_SLOTS];
extern "C" __attribute__((visibility("default"))) int dt_debug_b4_times(unsigned long long *dst, int clear)
{
    if (clear) {
        void *p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_b4_times)) != hipSuccess) return 1;
        return hipMemset(p, 0, sizeof(g_b4_times)) == hipSuccess ? 0 : 1;
    }
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_b4_times), sizeof(g_b4_times)) == hipSuccess ? 0 : 1;
}
#define B4_NOW() __builtin_readcyclecounter()
#define B4_PUT(k, v)                                                                                              \
    do {                                                                                                          \
        if (lane == 0 && blockIdx.x < B4_TT_WG && tt_i < B4_TT_ITEMS)                                             \
            g_b4_times[((blockIdx.x * 16 + wave) * B4_TT_ITEMS + tt_i) * B4_TT_SLOTS + (k)] = (v);                 \
    } while (0)
#else
#define B4_NOW() 0ull
#define B4_PUT(k, v) do { } while (0)
#endif

#define B4_INL __attribute__((always_inline))
// an opaque copy of a lane constant: addresses formed from it are recomputed where they are used (3-5 VALU instructions) instead of being
// hoisted out of the stage loop into registers the loop does not have -- hipcc spilled them to scratch and reloaded them in every stage
// behind s_waitcnt vmcnt(0), i.e. behind the DMA in flight
__device__ __forceinline__ int b4_opaque(int x) { asm volatile("" : "+v"(x)); return x; }
// the lane id, recomputed where it is needed (two VALU instructions, volatile: never hoisted, never spilled)
__device__ __forceinline__ int b4_lane()
{
    int x;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(x));
    return x;
}
template <int... Is, class F>
__device__ __forceinline__ void b4_for(std::integer_sequence<int, Is...>, F &&f) { (f(std::integral_constant<int, Is>()), ...); }

// NB: 16-channel blocks per wave.  2: eight waves of 16 tiles x 32 channels (two per SIMD, 256 registers each, every wave transforms in every
// stage); 1: SIXTEEN waves of 16 tiles x 16 channels (four per SIMD, 128 registers each): the per-wave chain transform -> MFMAs -> Y update
// is half as long and four waves per SIMD cover each other's LDS / DMA latencies; the two wave sets alternate between transforming the
// next stage's V and issuing the next stage's DMA
template <bool POOL, int NB>
__global__ __launch_bounds__(1024 / NB) void wino4b_fused_kernel(Wino4FusedArgs p)
{
    constexpr bool YH = NB == 1;                        // Y is updated after every half-row (12 M' registers per block) instead of every row (24)
    extern __shared__ __attribute__((aligned(16))) unsigned char b4_lds[];
    unsigned char *const Pb = b4_lds;                                       // [2 classes][39 KiB]: even patch rows | odd patch rows
    unsigned char *const Ub = b4_lds + B4_PATCH_BYTES;                      // [2][stage]
    unsigned char *const Vb = b4_lds + B4_PATCH_BYTES + 2 * B4_STAGE_BYTES; // [2][stage]
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    [[maybe_unused]] const int lane = threadIdx.x & 63;      // (timing build only: every phase below derives its lane constants from b4_lane())
    const int wm = wave & 3, wn = wave >> 2;            // MFMA role: 16-tile row block, block of 16 NB channels.  Waves w, w + 4, .. share a SIMD
    const int set = NB == 2 ? wn : (wn & 1);            // NB 2: set 1 transforms at the head of a stage, set 0 at its tail.  NB 1: set (k + 1) & 1 transforms V(k + 1)
    const int i8 = NB == 2 ? wave : (wave & 3) + 4 * (wave >> 3);      // index within the set (NB 1): tile row it transforms, DMA pieces it issues
    const int NS = p.Cin >> 4;                          // 16-channel slices
    const int NQ = p.N >> 6;                            // 64-channel output slices
    const int nblk = p.B * p.nby * p.nbx;
    const int nitems = nblk * NQ;
    if ((int)blockIdx.x >= nitems) return;
    const unsigned char *const ug = reinterpret_cast<const unsigned char *>(p.u);

    struct Item { int nq, y0, x0, b; const float *frame; };
    auto item_of = [&](int it) B4_INL {
        Item I;
        const int j = it / NQ;
        I.nq = it - j * NQ;
        const int bxy = p.nbx * p.nby;
        I.b = j / bxy;
        const int r = j - I.b * bxy;
        const int by = r / p.nbx, bx = r - by * p.nbx;
        I.y0 = by * 32; I.x0 = bx * 32;
        I.frame = p.in + (long long)I.b * p.in_bs;
        return I;
    };

    // ---- DMA: raw buffer loads into LDS -- the descriptor and the stage's offset are scalars, the lane contributes ONE 32-bit offset
    // (with flat pointers every piece's 64-bit lane address was a register pair hipcc hoisted, spilled and reloaded behind vmcnt(0));
    // an offset past num_records reads zeros: out-of-image pixels and unused slots need no second source ----
    constexpr unsigned B4_OOB = 0x7fff0000u;
    const __amdgpu_buffer_rsrc_t urs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char *>(ug), 0, (unsigned)(36 * 6) * (unsigned)(p.Cin * p.N), 0x00020000);
    // U of one stage: 18 KiB contiguous in the packed image.  Waves 0-5 take three contiguous pieces each (slot < 0: all; else the wave's piece `slot`)
    auto u_issue = [&](int nq, int s, int k, int buf, int slot) B4_INL {
        if (B4_ABLATE & 1) return;
        if (i8 >= 6) return;
        const int soff = ((nq * NS + s) * 12 + k) * B4_STAGE_BYTES + i8 * 3072;
        const int voff = b4_lane() * 16;
        unsigned char *dst = Ub + buf * B4_STAGE_BYTES + i8 * 3072;
#pragma unroll
        for (int i = 0; i < 3; ++i)
            if (slot < 0 || slot == i) __builtin_amdgcn_raw_ptr_buffer_load_lds(urs, (b4_lptr_t *)(dst + i * 1024), 16, voff, soff + i * 1024, 0, 0);
    };
    const unsigned frame_bytes = (unsigned)(((long long)(p.H - 1) * p.W + p.W - 1) * p.in_ld + p.Cin) * 4u;   // the last pixel's channels end here
    // pieces [lo, hi) of one row class (0: even patch rows, 1: odd) of the 16-channel slice s of item I: slot = (row * 36 + px') * 4 + g with
    // px' = px ^ ((px >> 2) & 3)
    auto patch_issue = [&](const Item &I, int s, int cls, int lo, int hi, int slot) B4_INL {
        if (B4_ABLATE & 2) return;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(I.frame), 0, frame_bytes, 0x00020000);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int pc = lo + i8 + 8 * i;
            if ((slot < 0 || slot == i) && pc < hi) {
                const unsigned sl = (unsigned)pc * 64u + (unsigned)b4_lane();
                const unsigned row = sl / 144u, rem = sl - row * 144u;
                const unsigned pxs = rem >> 2, g = rem & 3u;
                const unsigned px = pxs ^ ((pxs >> 2) & 3u);
                const int y = I.y0 - 1 + (int)(2u * row) + cls, x = I.x0 - 1 + (int)px;
                const bool ok = row < 17u && px < 34u && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
                const unsigned voff = ok ? (unsigned)((y * p.W + x) * p.in_ld + 4 * (int)g) * 4u : B4_OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (b4_lptr_t *)(Pb + cls * B4_CLS_BYTES + pc * 1024), 16, (int)voff, 64 * s, 0, 0);
            }
        }
    };

    // ---- input transform: this wave's tile row (ty = wave), lane = (tx = lane >> 3, channel pair q = lane & 7) ----
    // half-row (R, HF) = positions (R, 3 HF + j), j = 0..2, into V buffer vbuf.  The column sums t[1..4] of the first half are kept in
    // registers for the second (eight VGPRs across one stage): it reads window column 5 only
#ifndef B4_TKEEP
#define B4_TKEEP 0       // 1: keep t[1..4] from the first half of a row for the second (saves 14 window reads and 13 packed operations per row; the
#endif                   //    eight registers it holds across the MFMA phase spill there -- scratch traffic behind vmcnt waits: not kept)
    [[maybe_unused]] b4_f2 tk[4];
    auto produce_rh = [&](auto Rtag, auto HFtag, int vbuf) B4_INL {
        constexpr int R = decltype(Rtag)::value, HF = decltype(HFtag)::value;
        if (B4_ABLATE & 4) return;
        const int ln = b4_lane();
        const int tx = ln >> 3, q = ln & 7;
        const unsigned char *const pl = Pb + i8 * (2 * B4_ROWPITCH) + q * 8;      // patch row 4 ty + a: class a & 1, row index 2 ty + (a >> 1); ty = i8
        auto colb = [&](int b) B4_INL {                     // byte offset of window column b (swizzled pixel)
            const int px = 4 * tx + b;
            return (px ^ ((px >> 2) & 3)) * 64;
        };
        // the window reads of a batch of columns first (one LDS round trip), then their column sums.  NB 2: all five columns in one batch
        // (40 registers in flight); NB 1 (128 registers per wave): three, then two
        b4_f2 t[6];
        constexpr int PB = NB == 2 ? 6 : 3;
#pragma unroll
        for (int b0 = 0; b0 < 6; b0 += PB) {
            b4_f2 d[6][6];
#pragma unroll
            for (int b = b0; b < b0 + PB && b < 6; ++b) {
                if (B4_TKEEP ? (HF == 0 ? b > 4 : b != 5) : (b < HF || b > HF + 4)) continue;         // HF 0: columns 0..4; HF 1: columns 1..5 (B4_TKEEP: 5 only)
                const int cb = colb(b);
#pragma unroll
                for (int a = 0; a < 6; ++a)
                    if (b4_needs<R>(a)) d[b][a] = *reinterpret_cast<const b4_f2 *>(pl + (a & 1) * B4_CLS_BYTES + (a >> 1) * B4_ROWPITCH + cb);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int b = b0; b < b0 + PB && b < 6; ++b) {
                if (B4_TKEEP ? (HF == 0 ? b > 4 : b != 5) : (b < HF || b > HF + 4)) continue;
                t[b] = b4_bt<R>(d[b]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (B4_TKEEP && HF == 0) {
#pragma unroll
            for (int b = 1; b < 5; ++b) tk[b - 1] = t[b];
        } else if (B4_TKEEP) {
#pragma unroll
            for (int b = 1; b < 5; ++b) t[b] = tk[b - 1];
        }
        unsigned char *o = Vb + (8 * i8 + tx) * 32 + q * 4 + vbuf * B4_STAGE_BYTES;         // + (j * 3 + term) * 2048
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            b4_f2 v;
            if (HF == 0) v = j == 0 ? b4_bt<0>(t) : (j == 1 ? b4_bt<1>(t) : b4_bt<2>(t));
            else v = j == 0 ? b4_bt<3>(t) : (j == 1 ? b4_bt<4>(t) : b4_bt<5>(t));
            const unsigned h = b4_cvt2(v);
            const b4_f2 r1 = v - b4_up2(h);
            const unsigned m = b4_cvt2(r1);
            const b4_f2 r2 = r1 - b4_up2(m);
            const unsigned l = b4_cvt2(r2);
            *reinterpret_cast<unsigned *>(o + (j * 3 + 0) * 2048) = h;
            *reinterpret_cast<unsigned *>(o + (j * 3 + 1) * 2048) = m;
            *reinterpret_cast<unsigned *>(o + (j * 3 + 2) * 2048) = l;
        }
    };
    auto produce = [&](auto Ktag, int vbuf) B4_INL {      // stage K of a slice: position row K >> 1, half K & 1
        constexpr int K = decltype(Ktag)::value;
        produce_rh(std::integral_constant<int, (K >> 1)>(), std::integral_constant<int, (K & 1)>(), vbuf);
    };

    // ---- MFMA operands: lane (row = lane & 15, kg = lane >> 4): kg 0, 1 = channels 0-7 / 8-15 of the FIRST term slot, kg 2, 3 of the second ----
    f32x4 tmp[6][NB];                   // M' of the position row (YH: half-row, columns 0..2 used) in flight: [column c][16-channel block]
    b4_f2 Y[NB][2][4][4];               // accumulated outputs: [block][tile pair (accumulator registers 2 ip, 2 ip + 1)][row a][column j]
    auto ld16 = [&](const unsigned char *q) B4_INL {
        if (B4_ABLATE & 8) return __builtin_bit_cast(b4_bf8, b4_u4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u});
        return __builtin_bit_cast(b4_bf8, *reinterpret_cast<const b4_u4 *>(q));
    };
    auto mfma_stage = [&](auto Ktag, auto &&dma) B4_INL {      // positions (k >> 1, 3 (k & 1) + j); dma(j): the data movement issued behind position j's MFMAs
        constexpr int k = decltype(Ktag)::value;
        const int so = (k & 1) * B4_STAGE_BYTES;
        const int ln = b4_lane();
        const int kg = ln >> 4, r16 = ln & 15;
        const int a_row = (16 * wm + r16) * 32 + (kg & 1) * 16, b_row = (16 * NB * wn + r16) * 32 + (kg & 1) * 16;
        const int t0 = kg < 2 ? 0 : 2048;                                  // term plane of the second slot, relative to the first's
        const unsigned char *const a12 = Vb + a_row + t0;                   // (v1 | v2)
        const unsigned char *const a21 = Vb + a_row + 2048 - t0;            // (v2 | v1)
        const unsigned char *const a13 = Vb + a_row + 2 * t0;               // (v1 | v3)
        const unsigned char *const b12 = Ub + b_row + t0;                   // (u1 | u2)
        const unsigned char *const b31 = Ub + b_row + 4096 - 2 * t0;        // (u3 | u1)
        // the A fragments of position j + 1 are requested BEFORE the MFMAs of position j are issued, its B fragments right behind them (into
        // the registers those MFMAs have just read: 40 operand registers instead of 56 -- the phase also holds 128 of Y and 48 of M');
        // pinned with sched_barrier: left alone hipcc requests a fragment right in front of its first use and waits with lgkmcnt(0)
#ifndef B4_BDBL
#define B4_BDBL 0        // 1: the B fragments double-buffered too (56 operand registers)
#endif
        b4_bf8 A[2][3], B[2][NB][2];
        auto request_a = [&](int j, int buf) B4_INL {
            const int o = so + j * B4_POS_BYTES;
            A[buf][0] = ld16(a13 + o); A[buf][1] = ld16(a21 + o); A[buf][2] = ld16(a12 + o);
        };
        auto request_b = [&](int j, int buf) B4_INL {
            const int o = so + j * B4_POS_BYTES;
#pragma unroll
            for (int blk = 0; blk < NB; ++blk) { B[buf][blk][0] = ld16(b31 + o + blk * 512); B[buf][blk][1] = ld16(b12 + o + blk * 512); }
        };
        request_a(0, 0); request_b(0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int cb = j & 1, bb = B4_BDBL ? cb : 0;
            if (j < 2) { request_a(j + 1, cb ^ 1); if (B4_BDBL) request_b(j + 1, cb ^ 1); }
            __builtin_amdgcn_sched_barrier(0);
            const int c = YH ? j : 3 * (k & 1) + j;
#pragma unroll
            for (int blk = 0; blk < NB; ++blk) {
                f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[cb][0], B[bb][blk][0], acc, 0, 0, 0);      // (v1|v3).(u3|u1): u3 v1 + u1 v3
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[cb][1], B[bb][blk][1], acc, 0, 0, 0);      // (v2|v1).(u1|u2): u1 v2 + u2 v1
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[cb][2], B[bb][blk][1], acc, 0, 0, 0);      // (v1|v2).(u1|u2): u1 v1 + u2 v2
                tmp[c][blk] = acc;
            }
            __builtin_amdgcn_sched_barrier(0);
            if (j < 2 && !B4_BDBL) request_b(j + 1, 0);
            dma(j);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // Y += At[:, R] (x) (M'[R][0..5] A) for the finished position row R; HALF < 0: the whole row from tmp[0..5]; HALF 0 / 1: the contribution of
    // columns 0..2 / 3..5 alone from tmp[0..2] (the transform is linear in the columns too)
    auto yacc = [&](auto Rtag, auto Htag) B4_INL {
        constexpr int R = decltype(Rtag)::value, HALF = decltype(Htag)::value;
        if (B4_ABLATE & 16) return;
#pragma unroll
        for (int blk = 0; blk < NB; ++blk)
#pragma unroll
            for (int ip = 0; ip < 2; ++ip) {
                b4_f2 m[6];
#pragma unroll
                for (int c = 0; c < (HALF < 0 ? 6 : 3); ++c) m[c] = b4_f2{tmp[c][blk][2 * ip], tmp[c][blk][2 * ip + 1]};
                b4_f2 T[4];
                if (HALF < 0) {
                    const b4_f2 a = m[1] + m[2], b = m[1] - m[2], cc = m[3] + m[4], e = m[3] - m[4];
                    T[0] = m[0] + a + cc; T[1] = b + 2.0f * e; T[2] = a + 4.0f * cc; T[3] = b + 8.0f * e + m[5];
                } else if (HALF == 0) {
                    const b4_f2 a = m[1] + m[2], b = m[1] - m[2];
                    T[0] = m[0] + a; T[1] = b; T[2] = a; T[3] = b;
                } else {
                    const b4_f2 cc = m[0] + m[1], e = m[0] - m[1];       // m[0..2] = columns 3, 4, 5
                    T[0] = cc; T[1] = 2.0f * e; T[2] = 4.0f * cc; T[3] = 8.0f * e + m[2];
                }
                auto &y = Y[blk][ip];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (R == 0) y[0][j] += T[j];
                    else if (R == 5) y[3][j] += T[j];
                    else if (R == 1) { y[0][j] += T[j]; y[1][j] += T[j]; y[2][j] += T[j]; y[3][j] += T[j]; }
                    else if (R == 2) { y[0][j] += T[j]; y[1][j] -= T[j]; y[2][j] += T[j]; y[3][j] -= T[j]; }
                    else if (R == 3) { y[0][j] += T[j]; y[1][j] += 2.0f * T[j]; y[2][j] += 4.0f * T[j]; y[3][j] += 8.0f * T[j]; }
                    else { y[0][j] += T[j]; y[1][j] -= 2.0f * T[j]; y[2][j] += 4.0f * T[j]; y[3][j] -= 8.0f * T[j]; }
                }
                // pin the updated sums HERE: nothing reads Y before the epilogue, and left alone LLVM sinks every update of an item (and
                // the M' registers of each row with it, through scratch) into the last stage
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (R == 0 ? a == 0 : (R == 5 ? a == 3 : true)) asm volatile("" : "+v"(y[a][j]));
            }
    };
    // ---- prologue of the workgroup's first item: whole patch of slice 0, U of stage 0, V of stage 0 ----
    int item = blockIdx.x;
    Item cur = item_of(item);
    if (NB == 2 || set == 0) {
        patch_issue(cur, 0, 0, 0, 24, -1); patch_issue(cur, 0, 0, 24, B4_CLS_PIECES, -1);
        patch_issue(cur, 0, 1, 0, 24, -1); patch_issue(cur, 0, 1, 24, B4_CLS_PIECES, -1);
        u_issue(cur.nq, 0, 0, 0, -1);
    }
    __builtin_amdgcn_s_waitcnt(0x0f70);     // vmcnt(0)
    __syncthreads();
    if (NB == 2 || set == 0) produce(std::integral_constant<int, 0>(), 0);
    __syncthreads();
    bool first = true;
#ifdef DT_B4_TIMING
    int tt_i = 0;
#endif

#pragma unroll 1
    for (;;) {
        const int nxt_it = item + (int)gridDim.x;
        const bool has_next = nxt_it < nitems;
        const Item nx = item_of(has_next ? nxt_it : item);
#pragma unroll
        for (int blk = 0; blk < NB; ++blk)
#pragma unroll
            for (int ip = 0; ip < 2; ++ip)
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int j = 0; j < 4; ++j) Y[blk][ip][a][j] = b4_f2{0.0f, 0.0f};

#ifdef DT_B4_TIMING
        unsigned long long tt_dm = 0, tt_tr = 0, tt_mm = 0, tt_ya = 0, tt_wt = 0;
        B4_PUT(0, B4_NOW());
#endif
#pragma unroll 1
        for (int s = 0; s < NS; ++s) {
            const bool last_slice = s + 1 == NS;
            // the slice after this one: (cur, s + 1), or slice 0 of the next item
            const bool up_exists = !last_slice || has_next;
            const Item &up = last_slice ? nx : cur;
            const int up_s = last_slice ? 0 : s + 1;
            b4_for(std::make_integer_sequence<int, 12>(), [&](auto Ktag) B4_INL {
                constexpr int k = decltype(Ktag)::value;
                constexpr int k1 = (k + 1) % 12;
                const bool more = k < 11 || up_exists;              // a stage follows this one
                [[maybe_unused]] const unsigned long long c0 = B4_NOW();
                // ---- data movement for what follows: one piece slot behind each position of the MFMA phase ----
                // NB 2: every wave issues its share; NB 1: the set that does not transform in this stage (set k & 1) issues all of it
                const bool dma_role = NB == 2 || set == (k & 1);
                auto dma = [&](int slot) B4_INL {
                    if (!dma_role) return;
                    if (more) {
                        if (k < 11) u_issue(cur.nq, s, k + 1, (k + 1) & 1, slot);
                        else u_issue(up.nq, up_s, 0, 0, slot);
                    }
                    if (k == 9 && up_exists) patch_issue(up, up_s, 0, 0, 24, slot);
                    if (k == 10 && up_exists) patch_issue(up, up_s, 0, 24, B4_CLS_PIECES, slot);
                    if (k == 11 && up_exists) patch_issue(up, up_s, 1, 0, 24, slot);
                    if (k == 0 && !(first && s == 0)) patch_issue(cur, s, 1, 24, B4_CLS_PIECES, slot);
                };
                __builtin_amdgcn_sched_barrier(0);
                [[maybe_unused]] const unsigned long long c1 = B4_NOW();
                if ((NB == 2 ? set == 1 : !dma_role) && more) produce(std::integral_constant<int, k1>(), k1 & 1);
                __builtin_amdgcn_sched_barrier(0);
                [[maybe_unused]] const unsigned long long c2 = B4_NOW();
                mfma_stage(Ktag, dma);
                __builtin_amdgcn_sched_barrier(0);
                [[maybe_unused]] const unsigned long long c3 = B4_NOW();
                if (YH) yacc(std::integral_constant<int, (k >> 1)>(), std::integral_constant<int, (k & 1)>());
                else if (k & 1) yacc(std::integral_constant<int, (k >> 1)>(), std::integral_constant<int, -1>());
                __builtin_amdgcn_sched_barrier(0);
                [[maybe_unused]] const unsigned long long c4 = B4_NOW();
                if (NB == 2 && set == 0 && more) produce(std::integral_constant<int, k1>(), k1 & 1);
                [[maybe_unused]] const unsigned long long c5 = B4_NOW();
                __builtin_amdgcn_s_waitcnt(0x0f70);     // this wave's DMA pieces have landed
                __syncthreads();
#ifdef DT_B4_TIMING
                tt_dm += c1 - c0; tt_tr += (c2 - c1) + (c5 - c4); tt_mm += c3 - c2; tt_ya += c4 - c3; tt_wt += B4_NOW() - c5;
#endif
            });
        }
        first = false;
        B4_PUT(1, B4_NOW());

        // ---- epilogue: bias + LeakyReLU [+ 2x2 max] on the accumulated outputs.  D row 4 (lane >> 4) + i = tile of this wave's 16
        // (two tile rows of the block), D column = channel ----
        {
            const int ln = b4_lane();
            const int kg = ln >> 4, r16 = ln & 15;
#pragma unroll
            for (int blk = 0; blk < NB; ++blk) {
                const int ch = cur.nq * 64 + wn * 16 * NB + blk * 16 + r16;
                const float bias = p.bias[ch];
                asm volatile("" ::"v"(bias));          // retire the load on the straight path (wino4s_fused.hip's note)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int tl = 4 * kg + i;                                  // tile within the wave's 16
                    const int ty = 2 * wm + (tl >> 3), tx = tl & 7;
                    const int y0 = cur.y0 + 4 * ty, x0 = cur.x0 + 4 * tx;
                    float v[4][4];
#pragma unroll
                    for (int a = 0; a < 4; ++a)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float z = Y[blk][i >> 1][a][j][i & 1] + bias;
                            v[a][j] = fmaxf(z, z * p.slope);                    // LeakyReLU, 0 <= slope <= 1 (1: linear)
                        }
                    if (!POOL) {
                        float *ob = p.out + (long long)cur.b * p.out_bs + ((long long)y0 * p.W + x0) * p.out_ld + ch;
                        const int rs = p.W * p.out_ld;
                        const bool full = y0 + 4 <= p.H && x0 + 4 <= p.W;
                        if (full) {
#pragma unroll
                            for (int a = 0; a < 4; ++a)
#pragma unroll
                                for (int j = 0; j < 4; ++j) ob[a * rs + j * p.out_ld] = v[a][j];
                        } else {
#pragma unroll
                            for (int a = 0; a < 4; ++a)
#pragma unroll
                                for (int j = 0; j < 4; ++j)
                                    if (y0 + a < p.H && x0 + j < p.W) ob[a * rs + j * p.out_ld] = v[a][j];
                        }
                    } else {
                        const int H2 = p.H >> 1, W2 = p.W >> 1;
                        float *ob = p.out2 + (((long long)cur.b * H2 + (y0 >> 1)) * W2 + (x0 >> 1)) * p.out2_ld + ch;
                        const int rs = W2 * p.out2_ld;
#pragma unroll
                        for (int a2 = 0; a2 < 2; ++a2)
#pragma unroll
                            for (int j2 = 0; j2 < 2; ++j2) {
                                const float mx = fmaxf(fmaxf(v[2 * a2][2 * j2], v[2 * a2][2 * j2 + 1]), fmaxf(v[2 * a2 + 1][2 * j2], v[2 * a2 + 1][2 * j2 + 1]));
                                if ((y0 >> 1) + a2 < H2 && (x0 >> 1) + j2 < W2) ob[a2 * rs + j2 * p.out2_ld] = mx;
                            }
                    }
                }
            }
        }
#ifdef DT_B4_TIMING
        B4_PUT(2, B4_NOW()); B4_PUT(3, tt_dm); B4_PUT(4, tt_tr); B4_PUT(5, tt_mm); B4_PUT(6, tt_ya); B4_PUT(7, tt_wt);
        ++tt_i;
#endif
        if (!has_next) break;
        item = nxt_it;
        cur = nx;
    }
}

int launch_wino4b_fused(hipStream_t st, const Wino4FusedArgs &a_in, const float *zeros)
{
    Wino4FusedArgs a = a_in;
    if (a.B <= 0 || a.Cin % 16 || a.N % 64 || a.in_ld % 4) return 2;
    const bool pool = a.out2 != nullptr;
    if (pool && ((a.H | a.W) & 1)) return 2;
    if (pool == (a.out != nullptr)) return 2;       // exactly one of the two outputs
    a.nby = (a.H + 31) / 32;
    a.nbx = (a.W + 31) / 32;
    const long long blocks = (long long)a.B * a.nby * a.nbx;
    const long long items = blocks * (a.N / 64);
    if (items >= (1ll << 30)) return 2;
    a.zeros = zeros;
    const size_t lds = B4_LDS_BYTES;                // 153,600 B
    static PerDeviceOnce attr;
    static int cus[64];
    int dev = 0;
    if (attr.ensure(&dev, [&](int d) {
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(wino4b_fused_kernel<false, B4_NB>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
                hipFuncSetAttribute(reinterpret_cast<const void *>(wino4b_fused_kernel<true, B4_NB>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
                return 1;
            int n = 0;
            if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, d) != hipSuccess || n <= 0) n = 256;
            cus[d] = n;
            return 0;
        }))
        return 1;
    long long grid = cus[dev];                      // one 8-wave workgroup per CU, persistent over the items
    if (grid > items) grid = items;
    if (pool) hipLaunchKernelGGL((wino4b_fused_kernel<true, B4_NB>), dim3((unsigned)grid), dim3(1024 / B4_NB), lds, st, a);
    else hipLaunchKernelGGL((wino4b_fused_kernel<false, B4_NB>), dim3((unsigned)grid), dim3(1024 / B4_NB), lds, st, a);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

// Host: u36 = wino_pack_weights(4, ...) output [36][npad][cin] (U_p[n][c], p = 6 xi + nu) -> the kernel's stage images, bf16 terms:
//   dst[nq][slice s][stage k 12][j 3][term 3][n 64][c 16]  =  term of U_{6 (k >> 1) + 3 (k & 1) + j}[64 nq + n][16 s + c]
void wino4b_fused_pack(const float *u36, int npad, int cin, int cout, unsigned short *dst)
{
    const int ns = cin / 16, nquart = cout / 64;
    const size_t plane = (size_t)npad * cin;
    for (int nq = 0; nq < nquart; ++nq)
        for (int s = 0; s < ns; ++s)
            for (int k = 0; k < 12; ++k)
                for (int j = 0; j < 3; ++j) {
                    const int pos = 6 * (k >> 1) + 3 * (k & 1) + j;
                    for (int n = 0; n < 64; ++n)
                        for (int c = 0; c < 16; ++c) {
                            unsigned short t[3];
                            wino_s3_split_host(u36[(size_t)pos * plane + (size_t)(nq * 64 + n) * cin + 16 * s + c], t);
                            const size_t base = ((((size_t)(nq * ns + s) * 12 + k) * 3 + j) * 3) * 64 * 16;
                            for (int t3 = 0; t3 < 3; ++t3) dst[base + ((size_t)t3 * 64 + n) * 16 + c] = t[t3];
                        }
                }
}

// wino_gemm_s3.hip -- the batched GEMMs  M'[p] = V[p] (Mt x K)  *  U[p]^T (N x K)  of the Winograd-form layers on the BF16
// matrix pipe at fp32 accuracy: the arithmetic of the reference's Conv2D layers conv_9 .. conv_22
// (models_detection/KerasYOLO.py:323-396), of ConvLSTM2D's input and recurrent convolutions
// (models_tracking/MultiObjDetTracker.py:160-189) once they are in F(6x6,3x3) / F(4x4,3x3) form, and of five 1x1 Conv2D layers
// (conv_7 / 10 / 12 / 15 / 17) as plain GEMMs (P = 1) straight on their fp32 input.  DESIGN.md section 4.4.
//
// gfx950 has no fp32-rate shortcut (v_mfma_f32_32x32x2_f32 runs at 1/16 of the bf16 rate, no xf32), so each fp32
// operand is carried as THREE bf16 terms   x = x1 + x2 + x3   (x1 = bf16(x), x2 = bf16(x - x1), x3 = bf16(x - x1 - x2):
// 3 x 8 significand bits cover fp32's 24, the split is exact up to the last term's rounding, |x - x1 - x2 - x3| <= 2^-25 |x|)
// and a product is the six partial products of weight 2^0, 2^-8, 2^-16
//      u*v ~= u1 v1 + (u1 v2 + u2 v1) + (u1 v3 + u2 v2 + u3 v1)            (dropped: u2 v3, u3 v2, u3 v3 <= 2^-24 |u v|)
// accumulated in fp32 by v_mfma_f32_32x32x16_bf16, smallest terms first.  Per 32x32 block and 16 k: six bf16 MFMAs of 32
// cycles instead of eight fp32 MFMAs of 64 (2.67x fewer matrix-pipe cycles), with errors at the level of fp32's own product
// rounding (tests/test_gpu_parity.py::test_split_bf16_gemm_error_against_float64: not above the fp32 MFMA path's).
//
// The Winograd operands arrive ALREADY split, from their producers: V from the input transforms (winograd.hip:
// wino_input_s3_kernel, wino_input_kernel<4,4,S3>), U split on the device at load time (wino_s3_pack_kernel); both K-blocked so
// that the 16-deep stage of a 256-row tile is ONE contiguous 8 KiB run per term:     [p][term 3][K/16][rows][16] bf16.
// A 1x1 layer's activation is NOT pre-split: the kernel reads the fp32 NHWC tensor and splits its fragments in registers (VF below).
//
// Kernel: persistent; tile = BM rows of V x BN (256 | 128) rows of U; k in stages of 16 through an LDS ring filled by
// global_load_lds_dwordx4 in 1 KiB pieces, the next tile's first stages in flight during a tile's epilogue.  Two forms:
// BM = 256, eight waves, three stages (144 KiB, one workgroup per CU) -- or BM = 128, four waves, two stages (72 KiB, two
// workgroups per CU) where that saves a round of tiles.  Wave (wm, wn) owns 64 rows of V x BN/2 rows of U; the MFMA takes U
// as its A operand, so a lane holds 4 consecutive n of one m and the epilogue stores 16 B.
// LDS image of one (operand, term, stage): [row][2 granules of 8 bf16], granule index XOR (row >> 3) & 1 -- with the
// ds_read_b128 lane groups of gfx950 ({0-3,12-15,20-27}, ...) every group then covers all 64 banks once
// (SQ_LDS_BANK_CONFLICT = 0 measured).
//
// Round 6: the same kernel with TWO fp16 terms (NT = 2, "h2"):   x * 2^s = hi + lo,  hi = f16(x 2^s), lo = f16(x 2^s - hi)
// (round to nearest even; x 2^s - hi is exact in fp32 and has at most 12 significant bits, so lo drops at most its last one:
// |x 2^s - hi - lo| <= 2^-23 |x 2^s|, zero for three elements in four) and THREE partial products per multiply
//      u*v ~= lo_u hi_v + hi_u lo_v + hi_u hi_v                                  (dropped: lo_u lo_v <= 2^-22 |u v|)
// on v_mfma_f32_32x32x16_f16 -- half the matrix-pipe work of the bf16 form, 4 bytes per operand element instead of 6.
// fp16's 5-bit exponent needs the operands SCALED into its range: V by a power of two from the measured max |activation| of the
// tensor the transform reads (dt_internal.h: dt_h2_base; the 16-sub-slot `amax` word its producer -- or absmax_kernel -- filled)
// times a static per-position factor (the transform's gain), U per position at load (its own max); the epilogue multiplies the
// fp32 accumulator by the inverse (GemmS3Args::pscale[p] * dt_h2_base_inv(amax)): powers of two, exact.  Elements more than
// ~2^13 below the tensor's maximum have a subnormal lo term (absolute error 2^-25 in scaled units = 2^-40 of the maximum).
#include "dt_internal.h"
#include <cstring>

typedef __bf16 s3_bf8 __attribute__((ext_vector_type(8)));
typedef _Float16 s3_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 s3_h2 __attribute__((ext_vector_type(2)));
typedef float s3_f16 __attribute__((ext_vector_type(16)));
typedef float s3_f4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void s3_lptr_t;
typedef const __attribute__((address_space(1))) void s3_gptr_t;

#ifndef S3_DEFAULT_WAVES
#define S3_DEFAULT_WAVES 8
#endif
#ifndef S3_EPI_ROWS
#define S3_EPI_ROWS 1    // 1 (default since round 4) = V is the MFMA's A operand: a lane holds ONE column n and 16 rows of a block, and a store
#endif                   //   instruction writes 4 bytes per lane = two whole 128-byte lines (32 consecutive n of rows m, m + 4).  0 = U is the A
                         //   operand: 16-byte stores, but each instruction touches 32 lines with 32 bytes -- the address path of 64 scattered pieces
                         //   held the accumulators (a register an in-flight store reads cannot be reset) longer than four times as many
                         //   line-sized stores do.  Measured (tools/micro/gemm_s3_bench, profiles/r04_gemm_s3_epilogue.txt): K = 1024 5.06 -> 5.00 ms,
                         //   K = 512 2.89 -> 2.78, K = 256 3.19 -> 2.96, K = 128 3.93 -> 3.43, the recurrent step 0.315 -> 0.295
#ifndef S3_ABLATE
#define S3_ABLATE 0      // probes (tools/micro/gemm_s3_bench.hip): 1 no DMA, 2 no operand reads, 4 no barrier, 8 DMA from one tile's panels only
#endif

// ---- the 1x1 layers' A operand arrives as plain fp32 rows (the producing layer's NHWC activation) and is split here ------------------
typedef float s3_f2 __attribute__((ext_vector_type(2)));
typedef __bf16 s3_bf2 __attribute__((ext_vector_type(2)));
typedef unsigned s3_u4 __attribute__((ext_vector_type(4)));
// The producers' split (winograd.hip:s3_split4: three roundings to nearest even), pair by pair: one v_cvt_pk_bf16_f32 per pair and
// term, the pair widened again with a shift and a mask, the residual as one packed subtract -- 9 instructions per pair (left to
// __builtin_convertvector on 8 lanes hipcc converts every element twice: 17 per pair).
__device__ __forceinline__ unsigned s3_cvt2(const s3_f2 x) { return __builtin_bit_cast(unsigned, __builtin_convertvector(x, s3_bf2)); }
__device__ __forceinline__ s3_f2 s3_up2(unsigned u) { return s3_f2{__uint_as_float(u << 16), __uint_as_float(u & 0xffff0000u)}; }
__device__ __forceinline__ void s3_split_pair(const s3_f2 x, unsigned &h, unsigned &m, unsigned &l)
{
    h = s3_cvt2(x);
    const s3_f2 r1 = x - s3_up2(h);
    m = s3_cvt2(r1);
    const s3_f2 r2 = r1 - s3_up2(m);
    l = s3_cvt2(r2);
}

// the two-term fp16 split of a pair (already scaled): one packed convert per term, the residual exact in fp32
__device__ __forceinline__ void h2_split_pair(const s3_f2 x, unsigned &h, unsigned &l)
{
    const s3_h2 hh = __builtin_convertvector(x, s3_h2);
    const s3_f2 r = x - __builtin_convertvector(hh, s3_f2);
    h = __builtin_bit_cast(unsigned, hh);
    l = __builtin_bit_cast(unsigned, __builtin_convertvector(r, s3_h2));
}

// vmcnt(N) alone (expcnt / lgkmcnt at their "don't wait" values); N < 64
template <int N>
__device__ __forceinline__ void s3_wait_vm() { __builtin_amdgcn_s_waitcnt(0x0f70 | (N & 15) | ((N >> 4) << 14)); }

template <int NT>
__device__ __forceinline__ void s3_mfma(s3_f16 &c, const s3_bf8 &a, const s3_bf8 &b)      // a: U fragment (rows n), b: V fragment (rows m)
{
    if constexpr (NT == 2) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(s3_h8, b), __builtin_bit_cast(s3_h8, a), c, 0, 0, 0);      // D[i = m][j = n]
    } else {
#if S3_EPI_ROWS
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, c, 0, 0, 0);      // D[i = m][j = n]
#else
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);      // D[i = n][j = m]
#endif
    }
}
// the partial products of one multiply, smallest first: term of U, term of V
template <int NT> struct S3Prod;
template <> struct S3Prod<3> { static constexpr int N = 6; static constexpr int UT[6] = {2, 1, 0, 1, 0, 0}; static constexpr int VT[6] = {0, 1, 2, 0, 1, 0}; };
template <> struct S3Prod<2> { static constexpr int N = 3; static constexpr int UT[3] = {1, 0, 0}; static constexpr int VT[3] = {0, 1, 0}; };

// NW waves per workgroup: 8 = 4 (m) x 2 (n) waves of 64 x BN/2 at two waves per SIMD (<= 256 registers each);
//                         4 = 2 x 2 waves of 128 x BN/2, one wave per SIMD with the whole 512-entry register file (256 accumulators
//                             in AGPRs): a third fewer LDS fragment reads per MFMA (0.25 instead of 0.375 ds_read_b128).  Built
//                             with -DS3_WITH_4WAVES only (tools/micro/gemm_s3_bench.hip, S3_WAVES=4): with one read or DMA piece
//                             placed between every two MFMAs it runs exactly as fast as the 8-wave form -- 5.10 vs 5.03 ms, 70 %
//                             MFMA-busy at 1.63 GHz either way.  Two very different issue streams ending at the same busy share
//                             and clock says the matrix pipe is being throttled (power), not starved
//   ACT: LeakyReLU(p.slope) on the accumulators before the epilogue's stores (the 1x1 layers); a template parameter, so that
//   the Winograd instances carry none of it
//   BM rows of V per tile, NS LDS stages: 256 x 3 (one workgroup per CU, 144 KiB) or 128 x 2 with four waves (72 KiB: TWO workgroups
//   per CU, each with one wave per SIMD -- the same two waves per SIMD in all, but one workgroup's epilogue and barriers overlap
//   the other's main loop, and 128-row tiles fit short GEMMs better: the recurrent step's 588 rows are 5 x 128 instead of 3 x 256)
//   NT: terms per operand -- 3 bf16 terms / six products (round 3), or 2 fp16 terms of SCALED operands / three products (round 6)
template <int BN, int NW, bool ACT, int BM = 256, int NS = 3, int NT = 3>
__device__ __forceinline__ void s3_body(const GemmS3Args &p)
{
    static_assert(NT == 3 || (NT == 2 && S3_EPI_ROWS), "the fp16 form has the row epilogue only");
    constexpr int NPROD = S3Prod<NT>::N;
    constexpr int WM = NW / 2;                        // waves along m (two along n)
    constexpr int MB = BM / (WM * 32);             // 32-high m blocks per wave
    constexpr int NBW = BN / 64;                      // 32-wide n blocks per wave (BN/2 columns)
    constexpr int PV = (BM / 32) / NW;                // 32-row DMA pieces of a BM-row V region per wave
    constexpr int PU = (BN / 32) / NW > 0 ? (BN / 32) / NW : 1;   // ... of a BN-row U region (BN = 128, NW = 8: waves 0..3 only)
    constexpr bool U_HALF = (BN / 32) < NW;           // only the first BN / 32 waves move U rows
    // VF (the 1x1 instances, ACT): the A operand is the producing layer's fp32 activation [M][a_ld], read as it lies -- a stage is
    // BM rows x 16 k x 4 bytes, 16 rows per 1 KiB DMA piece -- and split into its three bf16 terms when a wave reads its fragment
    // (two ds_read_b128 + 36 VALU instructions per 32-row block and stage instead of three reads): 4 bytes per element from HBM
    // instead of the 6 of a pre-split operand, and no producer kernel of its own (rounds 3-4 had the output transform in front of a
    // 1x1 layer write split rows).  Same terms, same products, same order: bit-identical to the pre-split form.
    constexpr bool VF = ACT;
    constexpr int PVF = (BM / 16) / NW;               // 16-row pieces of the fp32 A region per wave
    constexpr int PT = VF ? NT * PU + PVF : NT * (PU + PV);   // DMA instructions per wave per stage
    constexpr int OP_A = NT * BN * 32;                // bytes of U terms per stage
    constexpr int STAGE = OP_A + (VF ? BM * 64 : NT * BM * 32);     // + V (fp32 | terms)
    extern __shared__ __attribute__((aligned(16))) unsigned char s3_lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave % WM, wn = wave / WM;
    const int KB = p.K >> 4;
    // fp16 form: the power of two this launch's V was scaled by (1x1 form: applied HERE, to the fp32 fragments) and its inverse for the
    // epilogue; behind the stage ring: [64] per-position epilogue factors | [Np] bias of a 1x1 layer (read per lane in the epilogue --
    // LDS reads wait on lgkmcnt, a global load there would wait for the next tile's DMA in flight)
    [[maybe_unused]] float out_am = 0.0f;      // 1x1 form: the largest |value| this lane stored (GemmS3Args::amax_out)
    [[maybe_unused]] float h2_fwd = 1.0f, h2_inv = 1.0f;
    [[maybe_unused]] float *h2_tab = reinterpret_cast<float *>(s3_lds + NS * STAGE);
    if constexpr (NT == 2) {
        const unsigned am = dt_amax_read(p.amax);
        h2_fwd = dt_h2_base(am);
        h2_inv = dt_h2_base_inv(am);
        for (int i = tid; i < 64; i += NW * 64) h2_tab[i] = i < p.P ? p.pscale[i] : 0.0f;
        if (ACT)
            for (int i = tid; i < p.Np; i += NW * 64) h2_tab[64 + i] = (p.bias && i < p.N) ? p.bias[i] : 0.0f;
        __syncthreads();      // (nothing is in flight yet)
    }
    const int MT = (p.Mt + BM - 1) / BM, NTL = (p.N + BN - 1) / BN;      // (a ragged N only in the 1x1 form: its columns past N meet zero rows of U and are not stored)
    const int ntiles = p.P * MT * NTL;             // < 2^31 (launcher)
    // XCD-aware order: workgroup w runs on XCD w % 8; each XCD walks a contiguous range of the n-fastest tile order, so
    // the 32 tiles resident on an XCD share their V / U panels through that XCD's L2
    const int G = gridDim.x;
    const int first = (G % 8 == 0) ? (int)(blockIdx.x & 7) * (G >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;

    // ---- DMA geometry: wave w moves PV (PU) pieces of 32 rows of every V (U) term region; lane = LDS slot inside a piece ----
    const int lrow = lane >> 1;
    const int dgran = (lane & 1) ^ ((lrow >> 3) & 1);           // source granule for LDS slot `lane` (32-row pieces: the row's bit 3 is the lane's)
    const long long a_term = (long long)KB * p.Mp * 16, b_term = (long long)KB * p.Np * 16;   // elements per term
    struct Tile { long long a0, b0; int m0, n0, pz; };
    auto tile_of = [&](int L) {
        Tile t;
        const int nt = L % NTL;
        const int r = L / NTL;
        const int mt = r % MT;
        t.pz = r / MT;
        t.m0 = mt * BM; t.n0 = nt * BN;
        t.a0 = (long long)t.pz * NT * a_term + (long long)t.m0 * 16;
        t.b0 = (long long)t.pz * NT * b_term + (long long)t.n0 * 16;
        return t;
    };
    // running DMA sources of this lane (term 0; the other terms sit a_term / b_term elements further): advanced by one
    // K block per stage, recomputed when the issue cursor enters a new tile
    const unsigned short *src_a = nullptr, *src_b = nullptr;
    const float *src_af[PVF > 0 ? PVF : 1] = {};      // VF: this lane's 16 bytes of each of the wave's pieces: row lane >> 2 of the piece, granule (lane & 3) ^ ((row >> 2) & 3)
    long long ta = a_term, tb = b_term;      // term strides of the stage being issued
    // A 1x1 layer's bias rides in the GEMM as ONE extra K stage: A rows = [1, 0, ..., 0] (p.ones, the same 256 rows for every
    // tile), B rows = [bias[n], 0, ..., 0] as split terms (p.bias_s3) -- the three products b_t x 1.0 are among the six formed,
    // so the accumulator receives the bias to 2^-25 and the epilogue needs no loads (a vector load there has to wait on
    // vmcnt, i.e. on the stores before it and on the next tile's DMA in flight: measured 2.4x slower at K = 128)
    const int KBX = KB + ((NT == 3 && p.bias_s3) ? 1 : 0);      // (the fp16 form adds the bias in its epilogue: h2_bias below)
    auto issue_src = [&](const Tile &t) {
        ta = a_term; tb = b_term;
        src_b = p.b + ((S3_ABLATE & 8) ? 0 : t.b0) + (long long)(32 * PU * wave + lrow) * 16 + dgran * 8;   // probe 8: every tile streams the same panels (L2 hits only)
        src_a = p.a + ((S3_ABLATE & 8) ? 0 : t.a0) + (long long)(32 * PV * wave + lrow) * 16 + dgran * 8;
        if (VF) {
#pragma unroll
            for (int sp = 0; sp < PVF; ++sp) {
                int row = t.m0 + 16 * (PVF * wave + sp) + (lane >> 2);
                row = row < p.Mt ? row : p.Mt - 1;      // rows past the end re-read the last one (their results are never stored)
                src_af[sp] = p.a_f32 + (long long)row * p.a_ld + ((lane & 3) ^ ((lane >> 4) & 3)) * 4;
            }
        }
    };
    // one stage = PT pieces per wave (term 0..2 x {U x PU, V x PV}); pieces [lo, hi) of the stage going to buffer `buf`
    auto issue_pieces = [&](int buf, int lo, int hi) {
        if (S3_ABLATE & 1) return;
        unsigned char *dst = s3_lds + buf * STAGE;
        int k = 0;
#pragma unroll
        for (int t3 = 0; t3 < NT; ++t3) {
#pragma unroll
            for (int sp = 0; sp < PU; ++sp, ++k)
                if (k >= lo && k < hi && (!U_HALF || wave < BN / 32))
                    __builtin_amdgcn_global_load_lds((s3_gptr_t *)(src_b + t3 * tb + sp * 512),
                                                     (s3_lptr_t *)(dst + t3 * BN * 32 + (wave * PU + sp) * 1024), 16, 0, 0);
            if (!VF) {
#pragma unroll
                for (int sp = 0; sp < PV; ++sp, ++k)
                    if (k >= lo && k < hi)
                        __builtin_amdgcn_global_load_lds((s3_gptr_t *)(src_a + t3 * ta + sp * 512),
                                                         (s3_lptr_t *)(dst + OP_A + t3 * BM * 32 + (wave * PV + sp) * 1024), 16, 0, 0);
            }
        }
        if (VF) {
#pragma unroll
            for (int sp = 0; sp < PVF; ++sp, ++k)
                if (k >= lo && k < hi)
                    __builtin_amdgcn_global_load_lds((s3_gptr_t *)src_af[sp], (s3_lptr_t *)(dst + OP_A + (wave * PVF + sp) * 1024), 16, 0, 0);
        }
    };
    auto issue_done = [&]() {
        src_b += (long long)p.Np * 16;
        src_a += (long long)p.Mp * 16;
        if (VF) {
#pragma unroll
            for (int sp = 0; sp < PVF; ++sp) src_af[sp] += 16;      // the next 16 channels of the same rows
        }
    };
    // ---- operand read offsets (bytes inside a stage) ----
    const int rl = lane & 31, gl = lane >> 5;
    int offU[NBW], offV[MB];
#pragma unroll
    for (int j = 0; j < NBW; ++j) {
        const int row = wn * (BN / 2) + 32 * j + rl;
        offU[j] = (2 * row + (gl ^ ((row >> 3) & 1))) * 16;
    }
#pragma unroll
    for (int i = 0; i < MB; ++i) {
        const int row = wm * (MB * 32) + 32 * i + rl;
        offV[i] = OP_A + (2 * row + (gl ^ ((row >> 3) & 1))) * 16;
        if (VF) offV[i] = OP_A + row * 64 + (((2 * gl) ^ ((row >> 2) & 3)) * 16);      // the first of the lane's two fp32 granules (k = 8 gl .. + 3); the second: offset ^ 16
    }

    if (first >= ntiles) return;
    // Two cursors walk this workgroup's sequence of (tile, k stage): `iss` three stages ahead (DMA), the compute cursor behind it.
    struct Cursor { Tile t; int kb; int L; bool valid; };
    auto advance = [&](Cursor &c) {
        ++c.kb;
        if (c.kb == KB && KBX > KB) {          // next: the bias stage of this tile
            src_a = p.ones + (long long)(32 * PV * wave + lrow) * 16 + dgran * 8; ta = 256 * 16;      // p.ones is [3][256][16] whatever the tile height
            if (VF) {      // ... and [256][16] floats, rows (1, 0, .., 0), for the fp32 form
#pragma unroll
                for (int sp = 0; sp < PVF; ++sp)
                    src_af[sp] = reinterpret_cast<const float *>(p.ones) + (long long)(16 * (PVF * wave + sp) + (lane >> 2)) * 16 + ((lane & 3) ^ ((lane >> 4) & 3)) * 4;
            }
            src_b = p.bias_s3 + (long long)(c.t.n0 + 32 * PU * wave + lrow) * 16 + dgran * 8; tb = (long long)p.Np * 16;
        } else if (c.kb == KBX) {
            c.kb = 0; c.L += G;
            c.valid = c.L < ntiles;
            if (c.valid) { c.t = tile_of(c.L); issue_src(c.t); }
        }
    };
    Cursor iss;
    iss.t = tile_of(first); iss.kb = 0; iss.L = first; iss.valid = true;
    issue_src(iss.t);
    Tile cur = iss.t;
    int Lcur = first;
    int n_ahead = 0;          // stages issued and not yet consumed
    int buf_issue = 0, buf_use = 0;
    auto issue_next = [&]() {            // a whole stage at once (prologue)
        if (!iss.valid) return;
        issue_pieces(buf_issue, 0, PT);
        issue_done();
        buf_issue = buf_issue == NS - 1 ? 0 : buf_issue + 1;
        ++n_ahead;
        advance(iss);
    };
    // In the steady state a stage's six pieces are spread over the NBW MFMA groups that follow the barrier which frees
    // its buffer: 48 DMA instructions issued by all eight waves at once queue up in the texture addresser and hold the
    // issuing waves (and with them the MFMA pipe) for ~400 cycles per stage.
    bool iss_go = false;       // a stage is being issued piecewise
    int iss_buf = 0;
    auto frag = [&](const unsigned char *sb, int off) {
        if (S3_ABLATE & 2) {
            typedef int s3_i4 __attribute__((ext_vector_type(4)));
            const s3_i4 x = {0x3f803f80 + (off & 1), 0x3f803f80 + (lane & 1), 0x3f003f80, 0x3f803f00 + (int)(sb - s3_lds)};
            return __builtin_bit_cast(s3_bf8, x);
        }
        return *reinterpret_cast<const s3_bf8 *>(sb + off);
    };
    auto fragf = [&](const unsigned char *sb, int off) { return *reinterpret_cast<const s3_f4 *>(sb + off); };
    // wait until at most `keep` of this wave's DMA stages are still in flight (PT pieces per stage; 3 PV for the waves
    // that move no U rows when BN = 128 at eight waves)
    auto wait_dma = [&](int keep) {
        if (keep <= 0) s3_wait_vm<0>();
        else if (keep == 1) {
            if (!U_HALF || wave < BN / 32) s3_wait_vm<PT>();
            else s3_wait_vm<(VF ? PVF : NT * PV)>();
        } else {
            if (!U_HALF || wave < BN / 32) s3_wait_vm<2 * PT>();
            else s3_wait_vm<(VF ? 2 * PVF : 2 * NT * PV)>();
        }
    };

    // prologue: three stages in flight, the first one's leading fragments in registers
#pragma unroll
    for (int q = 0; q < NS; ++q) issue_next();
    wait_dma(n_ahead - 1);
    if (!(S3_ABLATE & 4)) __builtin_amdgcn_s_barrier();
    s3_bf8 v[MB][NT], vn[MB][NT], ua[NT], ub[NT];
    // VF: one pair of a fragment -> its NT terms (the fp16 form scales by the tensor's power of two first)
    auto split_to = [&](const s3_f2 x, unsigned (*t)[4], int e) {
        if constexpr (NT == 2) h2_split_pair(x * h2_fwd, t[0][e], t[1][e]);
        else s3_split_pair(x, t[0][e], t[1][e], t[2][e]);
    };
    {
        const unsigned char *sb = s3_lds;
#pragma unroll
        for (int i = 0; i < MB; ++i) {
            if (VF) {
                const s3_f4 lo = fragf(sb, offV[i]), hi = fragf(sb, offV[i] ^ 16);
                unsigned t[NT][4];
                split_to(s3_f2{lo[0], lo[1]}, t, 0);
                split_to(s3_f2{lo[2], lo[3]}, t, 1);
                split_to(s3_f2{hi[0], hi[1]}, t, 2);
                split_to(s3_f2{hi[2], hi[3]}, t, 3);
#pragma unroll
                for (int t3 = 0; t3 < NT; ++t3) v[i][t3] = __builtin_bit_cast(s3_bf8, s3_u4{t[t3][0], t[t3][1], t[t3][2], t[t3][3]});
            } else {
#pragma unroll
                for (int t3 = 0; t3 < NT; ++t3) v[i][t3] = frag(sb, offV[i] + t3 * BM * 32);
            }
        }
#pragma unroll
        for (int t3 = 0; t3 < NT; ++t3) ua[t3] = frag(sb, offU[0] + t3 * BN * 32);
    }
    [[maybe_unused]] s3_f4 vraw[MB][2];             // VF: the next stage's fp32 fragments ...
    [[maybe_unused]] unsigned vnu[MB][NT][4];       // ... and their terms, built pair by pair
    bool drain = false;       // global stores were issued since the last full wait
#ifdef S3_TIMING
    unsigned long long tm_lgkm = 0, tm_vm = 0, tm_bar = 0, tm_n = 0;
    const unsigned long long tm_start = __builtin_readcyclecounter();
#endif

    for (;;) {
        s3_f16 acc[NBW][MB];
#pragma unroll
        for (int j = 0; j < NBW; ++j)
#pragma unroll
            for (int i = 0; i < MB; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[j][i][e] = 0.0f;

#pragma unroll 1
        for (int kb = 0; kb < KBX; ++kb) {
            const unsigned char *sb = s3_lds + buf_use * STAGE;
            buf_use = buf_use == NS - 1 ? 0 : buf_use + 1;
            const unsigned char *sn = s3_lds + buf_use * STAGE;          // the stage after this one
#pragma unroll
            for (int j = 0; j < NBW; ++j) {
                const s3_bf8 *u = (j & 1) ? ub : ua;
                if (j == NBW - 1) {
                    // every read of this stage has been issued: once they are back the buffer can be refilled.  The
                    // stage after this one must have landed before its first fragments are read below.
#ifdef S3_TIMING
                    const unsigned long long t0 = __builtin_readcyclecounter();
#endif
                    __builtin_amdgcn_s_waitcnt(0xc07f);                  // lgkmcnt(0)
#ifdef S3_TIMING
                    const unsigned long long t1 = __builtin_readcyclecounter();
#endif
                    --n_ahead;                                           // this stage is consumed
                    // (after an epilogue: vmcnt(0).  vmcnt(63) -- "all but the newest 63 operations", i.e. the pieces issued before the
                    // epilogue's stores -- was tried in round 4: 0.5 % at best, and WRONG in the two-stage ring of the 128-row form,
                    // where a needed piece can be younger than the stores)
                    wait_dma(drain ? 0 : n_ahead - 1);
                    drain = false;
#ifdef S3_TIMING
                    const unsigned long long t2 = __builtin_readcyclecounter();
#endif
                    if (!(S3_ABLATE & 4)) __builtin_amdgcn_s_barrier();  // (no fence: a release fence would wait for the DMA in flight)
#ifdef S3_TIMING
                    const unsigned long long t3 = __builtin_readcyclecounter();
                    tm_lgkm += t1 - t0; tm_vm += t2 - t1; tm_bar += t3 - t2; ++tm_n;
#endif
                }
                // the group's first MFMA goes ahead of the loads for the NEXT group: the wait the compiler puts in front of
                // it (for this group's fragments, read one group ago) then does not cover those fresh loads
                s3_mfma<NT>(acc[j][0], u[S3Prod<NT>::UT[0]], v[0][S3Prod<NT>::VT[0]]);
                __builtin_amdgcn_sched_barrier(0);
                if (j == NBW - 1) {
                    iss_go = iss.valid;                                  // refill the buffer this stage occupied, piecewise from here on
                    iss_buf = buf_issue;
                    if (iss_go) { buf_issue = buf_issue == NS - 1 ? 0 : buf_issue + 1; ++n_ahead; }
                }
                // The group's other MFMAs -- six partial products per block, smallest first, consecutive MFMAs to DIFFERENT
                // accumulators (block by block in runs that share the V fragment measured the same: the kernel's clock does not care
                // which operand repeats, profiles/r04_experiments.txt) -- with the side work placed between them in source order and frozen there (sched_barrier): one
                // LDS fragment read or one DMA piece per MFMA slot, so that the issuing wave never leaves the pipe idle for
                // longer than one instruction (it matters when the wave has the SIMD to itself, NW = 4).
                //   side work: next group's U fragments (3 reads; last group: the next stage's V and first U fragments,
                //   3 MB + 3 reads), then this group's share of the DMA pieces
                constexpr int per = (PT + NBW - 1) / NBW, ng = (PT + per - 1) / per;   // pieces per group, groups that carry pieces
                const int g = (j + 1) % NBW;                             // groups since the barrier: 0 = the barrier's own group
                constexpr int VR = VF ? 2 * MB : NT * MB;                 // V fragment reads of the next stage (last group)
                const int n_reads = (j == NBW - 1) ? VR + NT : NT;
                const int n_side = n_reads + ((g < ng) ? per : 0);
                auto side = [&](int k) {
                    if (k < n_reads) {
                        if (j == NBW - 1) {
                            if (k < VR) {
                                if (VF) vraw[k / 2][k % 2] = fragf(sn, offV[k / 2] ^ ((k % 2) * 16));
                                else vn[k / NT][k % NT] = frag(sn, offV[k / NT] + (k % NT) * BM * 32);
                            } else (j & 1 ? ua : ub)[k - VR] = frag(sn, offU[0] + (k - VR) * BN * 32);
                        } else
                            (j & 1 ? ua : ub)[k] = frag(sb, offU[j + 1] + k * BN * 32);
                    } else if (iss_go) {
                        const int pc = g * per + (k - n_reads);
                        issue_pieces(iss_buf, pc, pc + 1);
                        if (g == ng - 1 && k == n_side - 1) { issue_done(); advance(iss); iss_go = false; }
                    }
                };
                {
                    // VF, last group: the next stage's fp32 fragments were requested in slots 0 .. 2 MB - 1; their split -- 4 MB pairs of
                    // 9 VALU instructions, about one MFMA's shadow each -- goes pair by pair into the slots behind the reads, two pairs
                    // per slot towards the end, the rest behind the group's last MFMA
                    auto split_pair = [&](int q) {      // pair q % 4 of block q / 4
                        const int i = q >> 2, e = q & 3;
                        const s3_f4 &src = vraw[i][e >> 1];
                        split_to(s3_f2{src[2 * (e & 1)], src[2 * (e & 1) + 1]}, vnu[i], e);
                    };
                    // (the fp16 form's last group has fewer MFMA slots than reads: all its pairs follow the group's last MFMA, in the
                    //  shadow of the SIMD's other wave)
                    constexpr int NPAIR = 4 * MB, SLOT0 = 2 * MB + NT, NSLOT = NPROD * MB - 1 - SLOT0;
                    auto split_slot = [&](int sidx) {
                        if (!(VF && j == NBW - 1) || NSLOT < 2 || sidx < SLOT0) return;
                        const int r = sidx - SLOT0;
                        const int lo = r < NSLOT - 2 ? r : (NSLOT - 2) + 2 * (r - (NSLOT - 2));
                        const int hi = r < NSLOT - 2 ? lo + 1 : lo + 2;
#pragma unroll
                        for (int q = 0; q < NPAIR; ++q)
                            if (q >= lo && q < hi) split_pair(q);
                    };
#pragma unroll
                    for (int m = 1; m < NPROD * MB; ++m) {
                        if (m - 1 < n_side) side(m - 1);                 // (indices are compile-time constants after unrolling)
                        split_slot(m - 1);
                        const int pr = m / MB, i = m % MB;
                        s3_mfma<NT>(acc[j][i], u[S3Prod<NT>::UT[pr]], v[i][S3Prod<NT>::VT[pr]]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
#pragma unroll
                    for (int kk = NPROD * MB - 1; kk < NT * MB + NT + per; ++kk)    // (more side work than MFMA slots: MB = 2, last group)
                        if (kk < n_side) side(kk);
                    if (VF && j == NBW - 1) {
#pragma unroll
                        for (int q = 0; q < NPAIR; ++q)
                            if (NSLOT < 2 || q >= (NSLOT - 2) + 4) split_pair(q);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int i = 0; i < MB; ++i)
#pragma unroll
                for (int t3 = 0; t3 < NT; ++t3)
                    v[i][t3] = VF ? __builtin_bit_cast(s3_bf8, s3_u4{vnu[i][t3][0], vnu[i][t3][1], vnu[i][t3][2], vnu[i][t3][3]}) : vn[i][t3];
        }

        if constexpr (NT == 2) {      // back to the operands' own scale (powers of two: exact), a 1x1 layer's bias in the same fma
            const float f = h2_tab[cur.pz] * h2_inv;
#pragma unroll
            for (int j = 0; j < NBW; ++j) {
                const float bl = ACT ? h2_tab[64 + cur.n0 + wn * (BN / 2) + rl + 32 * j] : 0.0f;
#pragma unroll
                for (int i = 0; i < MB; ++i)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[j][i][e] = ACT ? __builtin_fmaf(acc[j][i][e], f, bl) : acc[j][i][e] * f;
            }
        }
        if (ACT) {      // a 1x1 layer: LeakyReLU (its bias came in through the extra K stage).  ALL of it before the first store:
                        // a VALU write to a register an in-flight store still reads waits for that store (32 serialised stores, 2.2x)
#pragma unroll
            for (int j = 0; j < NBW; ++j)
#pragma unroll
                for (int i = 0; i < MB; ++i)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[j][i][e] = acc[j][i][e] > 0.0f ? acc[j][i][e] : acc[j][i][e] * p.slope;
            __builtin_amdgcn_sched_barrier(0);
        }
#if S3_EPI_ROWS
        // ---- epilogue: lane holds, per block, column n = n0 + (lane & 31) of rows m = m0 + 8 (r / 4) + 4 (lane >> 5) + r % 4 ----
        float *cz = p.c + (long long)cur.pz * p.c_ps + cur.n0 + wn * (BN / 2) + rl;
        const int ncol = ACT ? p.N - (cur.n0 + wn * (BN / 2) + rl) : BN;      // columns of this lane's 32-column blocks that exist: 32 j < ncol
#pragma unroll
        for (int i = 0; i < MB; ++i) {
            const int mb = cur.m0 + wm * (MB * 32) + 32 * i + 4 * gl;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mb + 8 * (r >> 2) + (r & 3);
                if (m < p.Mt) {
                    float *row = cz + (long long)m * p.ldc;
#pragma unroll
                    for (int j = 0; j < NBW; ++j)
                        if (!ACT || 32 * j < ncol) {
                            if (ACT && p.amax_out) out_am = fmaxf(out_am, fabsf(acc[j][i][r]));
                            __builtin_nontemporal_store(acc[j][i][r], row + 32 * j);
                        }
                }
            }
        }
#else
        // ---- epilogue: lane holds, per block, n = n0 + 8q + 4*(lane>>5) + (0..3) of row m = m0 + (lane & 31) ----
        float *cz = p.c + (long long)cur.pz * p.c_ps;
#pragma unroll
        for (int i = 0; i < MB; ++i) {
            const int m = cur.m0 + wm * (MB * 32) + 32 * i + rl;
            if (m < p.Mt) {
                float *row = cz + (long long)m * p.ldc + cur.n0 + wn * (BN / 2) + 4 * gl;
#pragma unroll
                for (int j = 0; j < NBW; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        s3_f4 o;
                        o[0] = acc[j][i][4 * q]; o[1] = acc[j][i][4 * q + 1]; o[2] = acc[j][i][4 * q + 2]; o[3] = acc[j][i][4 * q + 3];
#if S3_ABLATE & 16
                        *reinterpret_cast<s3_f4 *>(row + 32 * j + 8 * q) = o;      // probe 16: plain stores
#else
                        __builtin_nontemporal_store(o, reinterpret_cast<s3_f4 *>(row + 32 * j + 8 * q));
#endif
                    }
            }
        }
#endif
        Lcur += G;
#ifdef S3_TIMING
        if (Lcur >= ntiles && lane == 0 && p.dbg) {
            unsigned long long *d = p.dbg + ((long long)blockIdx.x * NW + wave) * 5;
            d[0] = tm_lgkm; d[1] = tm_vm; d[2] = tm_bar; d[3] = tm_n; d[4] = __builtin_readcyclecounter() - tm_start;
        }
#endif
        if (Lcur >= ntiles) break;
        drain = true;
        cur = tile_of(Lcur);
    }
    if (ACT && p.amax_out) dt_amax_publish(p.amax_out, out_am);
}

#ifndef S3_H2_NS
#define S3_H2_NS 4           // LDS stages of the fp16 form (32 KiB each at 256-row tiles: 128 KiB)
#endif
#ifndef S3_H2_HALF_NS
#define S3_H2_HALF_NS 3      // ... of its 128-row form (24 KiB each, two workgroups per CU: 144 KiB)
#endif
template <int NT> constexpr int s3_ns() { return NT == 2 ? S3_H2_NS : 3; }
template <int NT> constexpr int s3_ns_half() { return NT == 2 ? S3_H2_HALF_NS : 2; }
template <int BN, int NW, bool ACT, int NT>
__global__ __launch_bounds__(NW * 64) void wino_gemm_s3_kernel(GemmS3Args p)
{
    s3_body<BN, NW, ACT, 256, s3_ns<NT>(), NT>(p);
}
// the two-workgroups-per-CU form: 128-row tiles, four waves, two LDS stages
template <int BN, bool ACT, int NT>      // (two waves per SIMD: without the attribute the allocator spreads over all 512 registers and only one workgroup fits)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void wino_gemm_s3_half_kernel(GemmS3Args p)
{
    s3_body<BN, 4, ACT, 128, s3_ns_half<NT>(), NT>(p);
}
#ifdef S3_WITH_4WAVES
// the one-wave-per-SIMD form: told so, or the register allocator budgets for two waves and spills the accumulators
template <int BN, bool ACT, int NT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void wino_gemm_s3_kernel_w4(GemmS3Args p)
{
    s3_body<BN, 4, ACT, 256, s3_ns<NT>(), NT>(p);
}
#endif

// executed 16-bit MFMA FLOPs of one launch (six partial products per multiply in the bf16 form, three in the fp16 form)
double wino_gemm_s3_flops(const GemmS3Args &a) { return (a.nt == 2 ? 6.0 : 12.0) * a.P * (double)a.Mt * a.K * a.N; }

bool wino_gemm_s3_usable(int Mt, int K, int N)
{
    return Mt > 0 && K >= 32 && K % 16 == 0 && N >= 128 && N % 128 == 0;
}

#define S3_H2_MAXNP 2048      // widest 1x1 layer whose bias the fp16 form keeps in LDS
template <int BN, int NW, bool ACT, int NT>
static int s3_launch(hipStream_t st, const GemmS3Args &a, long long grid)
{
    static PerDeviceOnce attr;
    // the largest request of this instance (ACT: the fp32 A stage; fp16 form: + the epilogue tables)
    const size_t lds_max = (size_t)s3_ns<NT>() * (NT * BN * 32 + (ACT ? 256 * 64 : NT * 256 * 32)) + (NT == 2 ? 256 + (ACT ? S3_H2_MAXNP * 4 : 0) : 0);
    const size_t lds = lds_max - ((NT == 2 && ACT) ? (size_t)(S3_H2_MAXNP - a.Np) * 4 : 0);
    if (attr.ensure(nullptr, [&](int) {
            return hipFuncSetAttribute(reinterpret_cast<const void *>(wino_gemm_s3_kernel<BN, NW, ACT, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max) != hipSuccess;
        }))
        return 1;
    hipLaunchKernelGGL((wino_gemm_s3_kernel<BN, NW, ACT, NT>), dim3((unsigned)grid), dim3(NW * 64), lds, st, a);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
template <int BN, bool ACT, int NT>
static int s3_launch_half(hipStream_t st, const GemmS3Args &a, long long grid)
{
    static PerDeviceOnce attr;
    const size_t lds_max = (size_t)s3_ns_half<NT>() * (NT * BN * 32 + (ACT ? 128 * 64 : NT * 128 * 32)) + (NT == 2 ? 256 + (ACT ? S3_H2_MAXNP * 4 : 0) : 0);
    const size_t lds = lds_max - ((NT == 2 && ACT) ? (size_t)(S3_H2_MAXNP - a.Np) * 4 : 0);
    if (attr.ensure(nullptr, [&](int) {
            return hipFuncSetAttribute(reinterpret_cast<const void *>(wino_gemm_s3_half_kernel<BN, ACT, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max) != hipSuccess;
        }))
        return 1;
    hipLaunchKernelGGL((wino_gemm_s3_half_kernel<BN, ACT, NT>), dim3((unsigned)grid), dim3(256), lds, st, a);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

static int s3_cus(int cus)
{
    if (cus > 0) return cus;
    static int cu_of[64];
    static PerDeviceOnce once;
    int dev = 0;
    if (once.ensure(&dev, [&](int d) {
            int n = 0;
            if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, d) != hipSuccess || n <= 0) return 1;
            cu_of[d] = n;
            return 0;
        }))
        return 0;
    return cu_of[dev];
}

// which tile form launch_wino_gemm_s3 takes for these arguments (the profile records it; tests assert it)
bool wino_gemm_s3_half_chosen(const GemmS3Args &a, int cus)
{
    cus = s3_cus(cus);
    const bool wide = a.N % 256 == 0 && a.Np % 256 == 0;
    if (!wide || cus <= 0) return false;
    const long long tiles = (long long)a.P * ((a.Mt + 255) / 256) * (a.N / 256), tiles_h = (long long)a.P * ((a.Mt + 127) / 128) * (a.N / 256);
    const long long rounds = (tiles + cus - 1) / cus, rounds_h = (tiles_h + 2 * cus - 1) / (2 * cus);
    return a.half > 0 || (a.half == 0 && rounds <= 8 && rounds_h < rounds);
}

int launch_wino_gemm_s3(hipStream_t st, const GemmS3Args &a, int cus)
{
    // the Winograd form needs whole 128-column tiles; the 1x1 form (a_f32) any N >= 64 up to its padded weight rows Np
    if (a.a_f32 ? (a.Mt <= 0 || a.K < 32 || a.K % 16 || a.N < 64 || a.N > a.Np) : !wino_gemm_s3_usable(a.Mt, a.K, a.N)) return 2;
    if (a.Mp % 256 || a.Mp < a.Mt || (!S3_EPI_ROWS && a.ldc % 4) || a.P <= 0) return 2;      // (the row-form epilogue stores 4 bytes per lane: any ldc)
    if ((a.bias_s3 != nullptr) != (a.ones != nullptr)) return 2;
    if (a.nt != 0 && a.nt != 2 && a.nt != 3) return 2;
    const bool h2 = a.nt == 2;      // the fp16 form: scaled operands, its bias (1x1 form) as plain floats for the epilogue
    if (h2 && (!a.pscale || !a.amax || a.bias_s3 || a.P > 64 || (a.a_f32 && a.Np > S3_H2_MAXNP))) return 2;
    // the 1x1 form: A = fp32 rows [Mt][a_ld] (a_f32), P = 1, LeakyReLU(slope) in the epilogue (slope 1 = none); the Winograd form: A = split terms (a)
    if (a.a_f32 ? (a.P != 1 || a.a_ld % 4 || a.a_ld < a.K || (reinterpret_cast<uintptr_t>(a.a_f32) & 15)) : (a.a == nullptr || a.act)) return 2;
    const bool act = a.a_f32 != nullptr;
    const bool wide = a.N % 256 == 0 && a.Np % 256 == 0;
    if (!wide && a.Np % 128) return 2;
    const int BN = wide ? 256 : 128;
    const long long tiles = (long long)a.P * ((a.Mt + 255) / 256) * ((a.N + BN - 1) / BN);
    const long long tiles_h = (long long)a.P * ((a.Mt + 127) / 128) * ((a.N + BN - 1) / BN);
    if (tiles_h >= (1ll << 31) - 65536) return 2;
    cus = s3_cus(cus);
    if (cus <= 0) return 1;
    // 256-row tiles, one workgroup per CU -- or 128-row tiles, two per CU (a round of either takes about the same time): the half
    // form when it needs fewer rounds, i.e. for short GEMMs whose rows fill 128-row tiles better (the recurrent step at 48 clips:
    // 588 rows = 5 x 128 or 3 x 256 -> 3 rounds instead of 4: 0.352 -> 0.312 ms).  With many tiles it loses: a 128-row tile stages
    // 1.5x the operand bytes per MFMA (K = 1024: 4.88 -> 5.20 ms; K = 256, where overlapping one workgroup's epilogue with the other's
    // main loop was the hope: 3.09 -> 3.25 ms).  a.half: 1 / -1 force it on / off (Policy::s3_half).
    const bool half = wino_gemm_s3_half_chosen(a, cus);
    if (half && wide) {
        long long grid = 2ll * cus;
        if (grid > tiles_h) grid = tiles_h;
        if (h2) return act ? s3_launch_half<256, true, 2>(st, a, grid) : s3_launch_half<256, false, 2>(st, a, grid);
        return act ? s3_launch_half<256, true, 3>(st, a, grid) : s3_launch_half<256, false, 3>(st, a, grid);
    }
    long long grid = cus;      // one workgroup per CU (108 / 144 KiB of LDS), persistent over the tiles
    if (grid > tiles) grid = tiles;
    const int nw = a.waves == 8 ? 8 : (a.waves == 4 ? 4 : S3_DEFAULT_WAVES);
#ifdef S3_WITH_4WAVES      // the one-wave-per-SIMD form (micro-benchmark builds)
    if (nw == 4 && !act) {
        static PerDeviceOnce attr4[4];
        const size_t lds = h2 ? (size_t)s3_ns<2>() * (2 * BN * 32 + 2 * 256 * 32) + 256 : (size_t)3 * (3 * BN * 32 + 3 * 256 * 32);
        const void *fn = h2 ? (wide ? reinterpret_cast<const void *>(wino_gemm_s3_kernel_w4<256, false, 2>) : reinterpret_cast<const void *>(wino_gemm_s3_kernel_w4<128, false, 2>))
                            : (wide ? reinterpret_cast<const void *>(wino_gemm_s3_kernel_w4<256, false, 3>) : reinterpret_cast<const void *>(wino_gemm_s3_kernel_w4<128, false, 3>));
        if (attr4[wide + 2 * h2].ensure(nullptr, [&](int) { return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess; }))
            return 1;
        if (h2) {
            if (wide) hipLaunchKernelGGL((wino_gemm_s3_kernel_w4<256, false, 2>), dim3((unsigned)grid), dim3(256), lds, st, a);
            else hipLaunchKernelGGL((wino_gemm_s3_kernel_w4<128, false, 2>), dim3((unsigned)grid), dim3(256), lds, st, a);
        } else {
            if (wide) hipLaunchKernelGGL((wino_gemm_s3_kernel_w4<256, false, 3>), dim3((unsigned)grid), dim3(256), lds, st, a);
            else hipLaunchKernelGGL((wino_gemm_s3_kernel_w4<128, false, 3>), dim3((unsigned)grid), dim3(256), lds, st, a);
        }
        return hipGetLastError() == hipSuccess ? 0 : 1;
    }
#endif
    if (h2) {
        if (act) return wide ? s3_launch<256, 8, true, 2>(st, a, grid) : s3_launch<128, 8, true, 2>(st, a, grid);
        return wide ? s3_launch<256, 8, false, 2>(st, a, grid) : s3_launch<128, 8, false, 2>(st, a, grid);
    }
    if (act) return wide ? s3_launch<256, 8, true, 3>(st, a, grid) : s3_launch<128, 8, true, 3>(st, a, grid);
    (void)nw;
    return wide ? s3_launch<256, 8, false, 3>(st, a, grid) : s3_launch<128, 8, false, 3>(st, a, grid);
}

// fp32 -> three bf16 terms, round-to-nearest-even at every step (the same arithmetic as the device split, winograd.hip:s3_split)
static inline unsigned short s3_bf16_rne(float x)
{
    unsigned int u;
    memcpy(&u, &x, 4);
    if ((u & 0x7f800000u) == 0x7f800000u) return (unsigned short)(u >> 16);      // inf / nan: truncate
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
static inline float s3_bf16_f32(unsigned short h)
{
    const unsigned int u = (unsigned int)h << 16;
    float x;
    memcpy(&x, &u, 4);
    return x;
}
void wino_s3_split_host(float x, unsigned short t[3])
{
    t[0] = s3_bf16_rne(x);
    const float r1 = x - s3_bf16_f32(t[0]);
    t[1] = s3_bf16_rne(r1);
    const float r2 = r1 - s3_bf16_f32(t[1]);
    t[2] = s3_bf16_rne(r2);
}

// U in the fp32 batched-GEMM layout [P][npad][K] (wino_pack_weights) -> [P][3][K/16][npad][16] bf16 terms
void wino_s3_pack_weights(const float *u, int P, int npad, int K, unsigned short *dst)
{
    const int KB = K / 16;
    for (int p = 0; p < P; ++p)
        for (int n = 0; n < npad; ++n)
            for (int k = 0; k < K; ++k) {
                unsigned short t[3];
                wino_s3_split_host(u[((size_t)p * npad + n) * K + k], t);
                for (int t3 = 0; t3 < 3; ++t3)
                    dst[((((size_t)p * 3 + t3) * KB + (k >> 4)) * npad + n) * 16 + (k & 15)] = t[t3];
            }
}

// ---- the fp16 form's host twins -------------------------------------------------------------------------------------------------------
unsigned short h2_f16_rne(float x)
{
    unsigned int u;
    memcpy(&u, &x, 4);
    const unsigned sign = (u >> 16) & 0x8000u;
    const unsigned a = u & 0x7fffffffu;
    if (a >= 0x7f800000u) return (unsigned short)(sign | 0x7c00u | (a > 0x7f800000u ? 0x200u : 0u));      // inf / nan
    if (a >= 0x477ff000u) return (unsigned short)(sign | 0x7c00u);      // >= 65520: rounds to infinity
    const int e = (int)(a >> 23) - 127;
    if (e >= -14) {      // normal in fp16: drop 13 mantissa bits, nearest even (a carry out of the mantissa bumps the exponent: still the right bits)
        const unsigned m = a & 0x7fffffu;
        unsigned h = ((unsigned)(e + 15) << 10) | (m >> 13);
        const unsigned rem = m & 0x1fffu;
        if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) ++h;
        return (unsigned short)(sign | h);
    }
    if (e < -25) return (unsigned short)sign;      // below half the smallest subnormal
    // subnormal: value = m24 * 2^(e - 23), unit 2^-24
    const unsigned m24 = (a & 0x7fffffu) | 0x800000u;
    const int sh = -e - 1;      // 14 .. 24
    unsigned h = m24 >> sh;
    const unsigned rem = m24 & ((1u << sh) - 1u), halfway = 1u << (sh - 1);
    if (rem > halfway || (rem == halfway && (h & 1u))) ++h;
    return (unsigned short)(sign | h);
}
float h2_f16_f32(unsigned short h)
{
    const unsigned sign = ((unsigned)h & 0x8000u) << 16;
    const unsigned e = (h >> 10) & 31u, m = h & 0x3ffu;
    float x;
    if (e == 31u) { const unsigned u = sign | 0x7f800000u | (m << 13); memcpy(&x, &u, 4); return x; }
    if (e == 0u) { x = (float)m * 5.9604644775390625e-8f; return sign ? -x : x; }      // m * 2^-24
    const unsigned u = sign | ((e + 112u) << 23) | (m << 13);
    memcpy(&x, &u, 4);
    return x;
}
void wino_h2_split_host(float x, unsigned short t[2])
{
    t[0] = h2_f16_rne(x);
    t[1] = h2_f16_rne(x - h2_f16_f32(t[0]));
}
void wino_h2_pack_weights(const float *u, int P, int npad, int K, const float *uscale, unsigned short *dst)
{
    const int KB = K / 16;
    for (int p = 0; p < P; ++p)
        for (int n = 0; n < npad; ++n)
            for (int k = 0; k < K; ++k) {
                unsigned short t[2];
                wino_h2_split_host(u[((size_t)p * npad + n) * K + k] * uscale[p], t);
                for (int t2 = 0; t2 < 2; ++t2)
                    dst[((((size_t)p * 2 + t2) * KB + (k >> 4)) * npad + n) * 16 + (k & 15)] = t[t2];
            }
}

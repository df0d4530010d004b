// conv1.hip -- first block of the detector, fused:
//   normalize (utility/utils.py:150-153: x/255.)  ->  Conv2D(32,(3,3),'same')
//   -> folded BatchNorm -> LeakyReLU(0.1) -> MaxPooling2D(2,2)
// (models_detection/KerasYOLO.py:278-282).
//
// Cin = 3, K = 27 and the frame is read exactly once.  A workgroup walks 8x8-POOLED-pixel tiles (16x16 conv pixels,
// 18x18x3 input patch) along a row of the frame; uint8 -> float goes through a 256-entry table so that the result
// equals the reference's float64 x/255. rounded to float32.
#include "dt_internal.h"

struct Conv1Args {
    const void *frames;
    int dtype;
    int B, H, W;
    const float *w;      // [27][32]  (ky,kx,ci) x cout, BN scale folded
    const float *bias;   // [32]
    const float *lut;    // [256]
    float slope;
    float *out;          // [B][H/2][W/2][32]
    int tpw;             // tiles a workgroup walks along x
    const unsigned *w3;     // [2 k-blocks][3 terms][64 lanes][4]: the weights as the B operand of v_mfma_f32_32x32x16_bf16, split into three bf16 terms
    const unsigned *w3u8;   // the same for weights / 255 (uint8 frames: conv1_s3_kernel<true>)
    unsigned *amax_out;     // conv1_s3_kernel: non-null = max |x| of the stored outputs into this slot (dt_amax_publish); conv_2's fp16 form scales by it
};

// K = 27 is short, but the fp32 MFMA peak equals the packed-FMA peak and an MFMA is ONE issue slot per 64 cycles: the
// staging, table and epilogue instructions of the other waves run beside it instead of competing for the VALU (the
// round-1 kernel -- v_pk_fma_f32 with the weights as SGPR operands, lane = pooled pixel -- took 6.0 ms per 1440
// frames = 45 % of the VALU peak; this one 4.8 ms).  v_mfma_f32_32x32x2_f32: row = conv pixel,
// column = output channel, k = (ky,kx,ci) padded to 28 -> 14 MFMAs per group of 32 pixels.  A group is one pooled row
// (two conv rows x 16 pixels) ordered so that rows 4w..4w+3 of the MFMA are the 2x2 pooling window w: a lane's four
// accumulators (4h + 8j .. +3) are then exactly one window of its channel -- the max is taken in registers and a
// wave-instruction stores two pooled pixels x 32 channels (2 x 128 B).  The patch lies in LDS as three planes
// (lanes walk x: conflict-free b32 reads), one read per MFMA; the 14 B operands (weights) live in registers.
#define C1_PL 328            // plane stride (18 x 18 = 324, padded)
#define C1_TPW 26            // most tiles a workgroup walks along x: the next tile's pixels are in flight during this tile's MFMAs

__global__ __launch_bounds__(256) void conv1_mfma_kernel(Conv1Args p)
{
    __shared__ float s_plane[3 * C1_PL];
    __shared__ float s_lut[256];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int H2 = p.H >> 1, W2 = p.W >> 1;
    const int by = blockIdx.y, b = blockIdx.z;
    const int ntx = (W2 + 7) / 8;
    const int bx_first = blockIdx.x * p.tpw;
    const int bx_last = min(bx_first + p.tpw, ntx);
    const int cy0 = by * 16 - 1;                      // patch origin row in input pixels

    s_lut[tid] = p.lut[tid];

    // patch pixels of this thread: i0 = tid, i1 = tid + 256 (< 324 for tid < 68)
    const int r0 = tid / 18, c0 = tid - r0 * 18;
    const int i1 = tid + 256, r1 = i1 / 18, c1 = i1 - r1 * 18;
    const bool has1 = i1 < 18 * 18;
    const int y0 = cy0 + r0, y1 = cy0 + r1;
    const bool yok0 = y0 >= 0 && y0 < p.H, yok1 = has1 && y1 >= 0 && y1 < p.H;
    const long long row0 = ((long long)b * p.H + y0) * p.W, row1 = ((long long)b * p.H + y1) * p.W;
    float raw[2][3];      // u8 frames: the bytes as floats' bit patterns (table index); f32 frames: the values
    auto fetch = [&](int bx) {
        const int x0 = bx * 16 - 1 + c0, x1 = bx * 16 - 1 + c1;
        const bool ok0 = yok0 && x0 >= 0 && x0 < p.W, ok1 = yok1 && x1 >= 0 && x1 < p.W;
        if (p.dtype == DT_FRAMES_U8) {
            const unsigned char *f = reinterpret_cast<const unsigned char *>(p.frames);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                raw[0][c] = __int_as_float(ok0 ? (int)f[(row0 + x0) * 3 + c] : -1);
                raw[1][c] = __int_as_float(ok1 ? (int)f[(row1 + x1) * 3 + c] : -1);
            }
        } else {
            const float *f = reinterpret_cast<const float *>(p.frames);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                raw[0][c] = ok0 ? f[(row0 + x0) * 3 + c] : 0.0f;
                raw[1][c] = ok1 ? f[(row1 + x1) * 3 + c] : 0.0f;
            }
        }
    };
    auto stage = [&]() {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float v0 = raw[0][c], v1 = raw[1][c];
            if (p.dtype == DT_FRAMES_U8) {
                const int q0 = __float_as_int(v0), q1 = __float_as_int(v1);
                v0 = q0 >= 0 ? s_lut[q0] : 0.0f;
                v1 = q1 >= 0 ? s_lut[q1] : 0.0f;
            }
            s_plane[c * C1_PL + tid] = v0;
            if (has1) s_plane[c * C1_PL + i1] = v1;
        }
    };

    // lane geometry: MFMA row m = lane & 31 -> window w = m >> 2, (dy, dx) = (m >> 1) & 1, m & 1; k half h = lane >> 5
    const int m = lane & 31, h = lane >> 5, n = lane & 31;
    const int px0 = 2 * (m >> 2) + (m & 1), dy = (m >> 1) & 1;
    float bw[14];
    int aoff[14];
#pragma unroll
    for (int s = 0; s < 14; ++s) {
        const int k0 = 2 * s, k1 = 2 * s + 1;                   // compile-time per step; the lane half picks one
        const int o0 = (k0 % 3) * C1_PL + (k0 / 9) * 18 + (k0 / 3) % 3;
        const int o1 = k1 < 27 ? (k1 % 3) * C1_PL + (k1 / 9) * 18 + (k1 / 3) % 3 : 0;
        aoff[s] = (h ? o1 : o0) + dy * 18 + px0;
        const int k = h ? k1 : k0;
        bw[s] = k < 27 ? p.w[k * 32 + n] : 0.0f;               // k = 27: the zero row of the padded K
    }
    const float bias = p.bias[n];

    fetch(bx_first);
    __syncthreads();                                            // the x/255 table is in LDS
    for (int bx = bx_first; bx < bx_last; ++bx) {
        stage();
        __syncthreads();
        if (bx + 1 < bx_last) fetch(bx + 1);                    // in flight under the MFMAs below
        {
            const float *pa = s_plane + (4 * wave) * 18;        // pooled rows g = 2 wave, 2 wave + 1 of the tile: two independent MFMA chains
            f32x16 acc0, acc1;
#pragma unroll
            for (int i = 0; i < 16; ++i) { acc0[i] = 0.0f; acc1[i] = 0.0f; }
#pragma unroll
            for (int s = 0; s < 14; ++s) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[aoff[s]], bw[s], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[aoff[s] + 36], bw[s], acc1, 0, 0, 0);
            }
#pragma unroll
            for (int gi = 0; gi < 2; ++gi) {
                const int oy = by * 8 + wave * 2 + gi;
#pragma unroll
                for (int j = 0; j < 4; ++j) {                   // window w = h + 2 j
                    float mx = -INFINITY;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float v = (gi ? acc1[4 * j + q] : acc0[4 * j + q]) + bias;
                        v = v > 0.0f ? v : v * p.slope;
                        mx = fmaxf(mx, v);
                    }
                    const int ox = bx * 8 + h + 2 * j;
                    if (oy < H2 && ox < W2) p.out[(((long long)b * H2 + oy) * W2 + ox) * 32 + n] = mx;
                }
            }
        }
        __syncthreads();                                        // every wave is done reading the planes
    }
}


// ---- the same layer on the bf16 matrix pipe at fp32 accuracy (default; Policy::s3_conv1) ----------------------------------
// v_mfma_f32_32x32x2_f32 runs at 1/16 of the bf16 MFMA rate and this layer spent 3.1 of its 5.0 ms per 1440 frames on the
// matrix pipe, the rest on LDS reads, staging and the epilogue's VALU -- all of them about equally (measured: halving the MFMA
// time alone bought 6 %).  conv1_s3_kernel carries fp32 operands as bf16 terms (the arithmetic of wino_gemm_s3.hip: x = x1 + x2 +
// x3 exact to 2^-25 |x|, products formed from the partial products of weight >= 2^-16, smallest first, fp32 accumulate), K = 27
// padded to 32 = two k-blocks of v_mfma_f32_32x32x16_bf16:
//   * the patch lies in LDS as bf16 planes [term][ci][18 x 18]; a lane's A fragment (8 consecutive k of its pixel) is eight
//     16-bit reads per term at per-lane offsets (k -> (ky, kx, ci) is the weight file's order, k = 9 ky + 3 kx + ci);
//     k = 27..31 re-read k = 0 (finite) against zero weights;
//   * the weights' three terms are packed on the host in the B operand's lane order and live in 24 registers;
//   * the 2x2 max is taken BEFORE bias + LeakyReLU (both non-decreasing: same bits, a quarter of the epilogue's VALU).
typedef __bf16 c1_bf8 __attribute__((ext_vector_type(8)));
typedef unsigned short c1_us8 __attribute__((ext_vector_type(8)));
typedef unsigned int c1_u4 __attribute__((ext_vector_type(4)));
typedef unsigned int c1_u2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned short c1_bf16_rne(float x)
{
    return __builtin_bit_cast(unsigned short, static_cast<__bf16>(x));
}
__device__ __forceinline__ float c1_bf16_f32(unsigned short h) { return __uint_as_float((unsigned)h << 16); }

// U8 = true (uint8 frames, the production input): the byte values 0..255 are EXACT bf16 numbers, so the patch needs ONE term and
// no table at all; the 1/255 of normalize() is folded into the weights instead -- w' = float32(float64(w) / 255), carried as
// three bf16 terms -- and a product is three MFMAs (x w'3, x w'2, x w'1).  Against the reference's float32(x / 255) * w this
// moves one float32 rounding from the activation to the weight (each product still carries exactly one rounding of 2^-24):
// fp32 accuracy, not the same bits as the float32-frame path (tests: both within 5e-6 of the oracle, 2e-6 of each other).
// U8 = false (float32 frames, already normalised): arbitrary values, the general three-term form with six products.
#ifndef C1_PF
#define C1_PF 2              // tiles of frame loads in flight ahead of the tile being computed (1 or 2)
#endif
#ifndef C1_LDS_BARRIER
#define C1_LDS_BARRIER 1     // 0: __syncthreads() (round 3's form) for A/B
#endif
#if C1_LDS_BARRIER
#define C1_SYNC() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#else
#define C1_SYNC() __syncthreads()
#endif
#ifndef C1_NTS
#define C1_NTS 1             // 1: the output rows as nontemporal stores (A/B builds)
#endif
#if C1_NTS
#define C1_STORE(ptr, val) __builtin_nontemporal_store((val), (ptr))
#else
#define C1_STORE(ptr, val) (*(ptr) = (val))
#endif
#ifndef C1_ABLATE
#define C1_ABLATE 0          // timing-only builds (results WRONG): 1 no output stores, 2 no frame loads, 4 no fragment reads + MFMAs, 8 no staging
#endif
template <bool U8, bool FULL = false>      // FULL: H and W are multiples of 16 -- every tile is whole, no bounds checks around the stores
__global__ __launch_bounds__(256) void conv1_s3_kernel(Conv1Args p)
{
    float out_am = 0.0f;      // the largest |value| this lane stored (Conv1Args::amax_out)
    constexpr int NT = U8 ? 1 : 3;                    // bf16 terms of a patch value
    __shared__ unsigned short s_pl[3 * NT * C1_PL];   // [term][ci 3][18 x 18 (+4)]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int H2 = p.H >> 1, W2 = p.W >> 1;
    const int by = blockIdx.y, b = blockIdx.z;
    const int ntx = (W2 + 7) / 8;
    const int bx_first = blockIdx.x * p.tpw;
    const int bx_last = min(bx_first + p.tpw, ntx);
    const int cy0 = by * 16 - 1;                      // patch origin row in input pixels

    // ---- staging.  Every load is UNCONDITIONAL (the address is selected, not the load): a load inside a lane-divergent branch is
    // followed by vmcnt(0) at the join, which serialises the tile's loads and drains the stores in flight (round 3's form: 2.6 of
    // the layer's 4.4 ms).
    // uint8 frames (W % 4 == 0: every frame row starts on a dword): the 18 x 54-byte patch rows are read as 14 aligned dwords per row,
    // bytes 48 bx - 4 .. 48 bx + 51 of the row -- ONE dword per thread (252 of 256) instead of six byte loads; its four bytes are patch
    // elements e = 4 d - 1 + i (pixel e / 3, channel e % 3; e = -1 and 54 fall outside and go to a pad slot).  A dword lies
    // entirely inside or outside its row, and outside means zero padding.
    // float32 frames: pixel tid and tid + 256 of the patch, three values each.
    struct Raw { unsigned dw; float v[2][3]; };
    const int r0 = U8 ? tid / 14 : tid / 18, c0 = U8 ? tid - r0 * 14 : tid - r0 * 18;
    const int i1 = tid + 256, r1 = i1 / 18, c1 = i1 - r1 * 18;
    const bool has1 = !U8 && i1 < 18 * 18;
    const int y0 = cy0 + r0, y1 = cy0 + r1;
    const bool yok0 = (!U8 || tid < 252) && y0 >= 0 && y0 < p.H, yok1 = has1 && y1 >= 0 && y1 < p.H;
    const long long row0 = ((long long)b * p.H + y0) * p.W, row1 = ((long long)b * p.H + y1) * p.W;
    int poff[4];          // uint8: plane slot of byte i of this thread's dword
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int e = 4 * c0 - 1 + i;
        poff[i] = (U8 && tid < 252 && e >= 0 && e < 54) ? (e % 3) * C1_PL + r0 * 18 + e / 3 : C1_PL - 1;
    }
    auto fetch = [&](int bx, Raw &raw) {
        if (C1_ABLATE & 2) {
            raw.dw = 0x01020304u * (unsigned)(bx + 1);
#pragma unroll
            for (int c = 0; c < 3; ++c) { raw.v[0][c] = (float)(bx + c); raw.v[1][c] = (float)(bx - c); }
            return;
        }
        if (U8) {
            // an out-of-image dword is read from the zero words behind the weight table: no select after the load (it would wait for it)
            const int xb = 48 * bx - 4 + 4 * c0;
            const bool ok = yok0 && xb >= 0 && xb < 3 * p.W;
            const unsigned *src = ok ? reinterpret_cast<const unsigned *>(p.frames) + ((row0 * 3 + xb) >> 2) : p.w3u8 + 1536;
            raw.dw = *src;
        } else {
            const int x0 = bx * 16 - 1 + c0, x1 = bx * 16 - 1 + c1;
            const bool ok0 = yok0 && x0 >= 0 && x0 < p.W, ok1 = yok1 && x1 >= 0 && x1 < p.W;
            const float *zero = reinterpret_cast<const float *>(p.w3 + 1536);
            const float *f0 = ok0 ? reinterpret_cast<const float *>(p.frames) + (row0 + x0) * 3 : zero;
            const float *f1 = ok1 ? reinterpret_cast<const float *>(p.frames) + (row1 + x1) * 3 : zero;
#pragma unroll
            for (int c = 0; c < 3; ++c) { raw.v[0][c] = f0[c]; raw.v[1][c] = f1[c]; }
        }
    };
    auto stage = [&](const Raw &raw) {
        if (C1_ABLATE & 8) { if (raw.dw == 0x12345u && raw.v[0][0] == -1.0f) s_pl[tid] = 1; return; }
        if (U8) {
#pragma unroll
            for (int i = 0; i < 4; ++i)      // 0..255 has at most 8 significant bits: the float's high half is the exact bf16
                s_pl[poff[i]] = (unsigned short)(__float_as_uint((float)((raw.dw >> (8 * i)) & 255u)) >> 16);
        } else {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                unsigned short t0[3], t1[3];       // three roundings to nearest even (host twin: wino_s3_split_host)
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    unsigned short *t = q ? t1 : t0;
                    const float v = raw.v[q][c];
                    t[0] = c1_bf16_rne(v);
                    const float d1 = v - c1_bf16_f32(t[0]);
                    t[1] = c1_bf16_rne(d1);
                    t[2] = c1_bf16_rne(d1 - c1_bf16_f32(t[1]));
                }
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    s_pl[(3 * t + c) * C1_PL + tid] = t0[t];
                    if (has1) s_pl[(3 * t + c) * C1_PL + i1] = t1[t];
                }
            }
        }
    };

    // lane geometry: MFMA row m = lane & 31 -> window w = m >> 2, (dy, dx) = (m >> 1) & 1, m & 1; k half h = lane >> 5
    const int m = lane & 31, h = lane >> 5, n = lane & 31;
    const int px0 = 2 * (m >> 2) + (m & 1), dy = (m >> 1) & 1;
    int aoff[2][8];       // element offset (within a term's planes) of k = 16 kb + 8 h + t for this lane's pixel, group 0
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int k0 = 16 * kb + t, k1 = 16 * kb + 8 + t;           // compile-time; the lane half picks one
            const int o0 = k0 < 27 ? (k0 % 3) * C1_PL + (k0 / 9) * 18 + (k0 / 3) % 3 : 0;
            const int o1 = k1 < 27 ? (k1 % 3) * C1_PL + (k1 / 9) * 18 + (k1 / 3) % 3 : 0;   // k >= 27: k = 0's (finite) value x a zero weight
            aoff[kb][t] = (h ? o1 : o0) + dy * 18 + px0 + (4 * wave) * 18;
        }
    c1_bf8 bw[2][3];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int t = 0; t < 3; ++t)
            bw[kb][t] = __builtin_bit_cast(c1_bf8, reinterpret_cast<const c1_u4 *>(U8 ? p.w3u8 : p.w3)[(kb * 3 + t) * 64 + lane]);
    const float bias = p.bias[n];
    // Retire these loads HERE, on the straight path: consumed first inside the lane-divergent store branches of the loop, the bias
    // gets a vmcnt(0) in front of every use -- which waits for every store issued so far, one store at a time.
    asm volatile("" ::"v"(bias));
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int t = 0; t < 3; ++t) asm volatile("" ::"v"(bw[kb][t]));

    // One tile: planes -> LDS, barrier, the tile C1_PF ahead requested into the register set just freed, MFMAs + epilogue, barrier.
    // The barriers order LDS traffic only (C1_SYNC: lgkmcnt(0) + s_barrier): __syncthreads() also drains vmcnt, i.e. waits at
    // EVERY tile for the frame loads just issued and for the write acknowledgements of the tile's stores -- measured (round 4,
    // tools/c1_time.py ablations): 4.47 ms with, against 1.9 ms without the loads and 1.5 ms of compute alone.
    auto tile = [&](int bx, Raw &raw) {
        stage(raw);
        C1_SYNC();
        // in flight across the next C1_PF tiles.  FULL: requested unconditionally (past the end: the last tile again), so that
        // every trip of the loop issues the same loads and stores and hipcc can COUNT its way to the right vmcnt
        if (FULL) fetch(min(bx + C1_PF, bx_last - 1), raw);
        else if (bx + C1_PF < bx_last) fetch(bx + C1_PF, raw);
        {
            // pooled rows g = 2 wave, 2 wave + 1 of the tile (gi = 0, 1: +36 elements), one after the other: the second group's
            // fragment reads are in flight under the first group's MFMAs, and only one group's fragments are live
            f32x16 acc[2];
#pragma unroll
            for (int gi = 0; gi < 2; ++gi) {
                c1_bf8 a[2][NT];
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        c1_us8 v;
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = (C1_ABLATE & 4) ? (unsigned short)(0x3f80 + e + gi) : s_pl[3 * t * C1_PL + aoff[kb][e] + 36 * gi];
                        a[kb][t] = __builtin_bit_cast(c1_bf8, v);
                    }
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[gi][i] = 0.0f;
                // partial products smallest first; the two k-blocks alternate.  U8: x w3, x w2, x w1.  General form: the six of
                // weight >= 2^-16: (x3 w1, x2 w2, x1 w3), (x2 w1, x1 w2), x1 w1
#define C1_MM(ta, tb)                                                                                                  \
                acc[gi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][ta], bw[0][tb], acc[gi], 0, 0, 0);               \
                acc[gi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][ta], bw[1][tb], acc[gi], 0, 0, 0);
                if (C1_ABLATE & 4) { acc[gi][0] = bias * (float)(bx + gi); }
                else if (U8) { C1_MM(0, 2) C1_MM(0, 1) C1_MM(0, 0) }
                else { C1_MM(2 % NT, 0) C1_MM(1 % NT, 1) C1_MM(0, 2) C1_MM(1 % NT, 0) C1_MM(0, 1) C1_MM(0, 0) }
#undef C1_MM
            }
            // bias + LeakyReLU + 2x2 max: LeakyReLU (slope >= 0) and the bias add are non-decreasing, so the max of the window's
            // four accumulators goes first and the activation is applied once -- the same bits as activating all four
#pragma unroll
            for (int gi = 0; gi < 2; ++gi) {
                const int oy = by * 8 + wave * 2 + gi;
#pragma unroll
                for (int j = 0; j < 4; ++j) {                   // window w = h + 2 j
                    const float mx = fmaxf(fmaxf(acc[gi][4 * j], acc[gi][4 * j + 1]), fmaxf(acc[gi][4 * j + 2], acc[gi][4 * j + 3])) + bias;
                    const float v = mx > 0.0f ? mx : mx * p.slope;
                    const int ox = bx * 8 + h + 2 * j;
                    if ((C1_ABLATE & 1) ? v == 12345.678f : (FULL || (oy < H2 && ox < W2))) {
                        out_am = fmaxf(out_am, fabsf(v));
                        C1_STORE(&p.out[(((long long)b * H2 + oy) * W2 + ox) * 32 + n], v);
                    }
                }
            }
        }
        C1_SYNC();                                              // every wave is done reading the planes
    };
    Raw rawA, rawB;
    fetch(bx_first, rawA);
    if (FULL && C1_PF == 2) {
        // Whole tiles: a loop over PAIRS of tiles in which nothing is conditional (one load and eight stores per tile), entered with
        // both prefetched tiles LANDED (one wait per workgroup walk).  The wait in front of a tile's staging then counts the stores
        // and the load issued since its own load -- vmcnt(17) -- instead of draining them: the loop head's vmcnt(0) of the guarded
        // form waited for the previous tile's eight stores to be acknowledged (it must assume the path on which none was issued).
        fetch(min(bx_first + 1, bx_last - 1), rawB);
        asm volatile("" ::"v"(rawA.dw), "v"(rawB.dw), "v"(rawA.v[0][0]), "v"(rawB.v[0][0]));
        int bx = bx_first;
#pragma unroll 1
        for (; bx + 1 < bx_last; bx += 2) {
            tile(bx, rawA);
            tile(bx + 1, rawB);
        }
        if (bx < bx_last) tile(bx, rawA);
        if (p.amax_out) dt_amax_publish(p.amax_out, out_am);
        return;
    }
    if (C1_PF == 2 && bx_first + 1 < bx_last) fetch(bx_first + 1, rawB);
#pragma unroll 1
    for (int bx = bx_first; bx < bx_last; bx += C1_PF) {
        tile(bx, rawA);
        if (C1_PF == 2 && bx + 1 < bx_last) tile(bx + 1, rawB);
    }
    if (p.amax_out) dt_amax_publish(p.amax_out, out_am);
}

// Host: the weight tables of conv1_s3_kernel.
//   w3 [2][3][64][4]: B operand of v_mfma_f32_32x32x16_bf16 per (k-block kb, term): lane = 32 h + n holds k = 16 kb + 8 h .. + 7
//   of column n as four dwords (k even | k odd << 16); w = [27][32] (k = 9 ky + 3 kx + ci, BatchNorm scale folded), k >= 27: 0
//   scale255: the uint8 form -- every weight divided by 255 in float64 first (normalize(), utils.py:150-153, folded into the weights)
void conv1_split_tables(const float *w_in /*[27][32]*/, bool scale255, unsigned *w3 /*[1536]*/)
{
    float w[27 * 32];
    for (int i = 0; i < 27 * 32; ++i) w[i] = scale255 ? (float)((double)w_in[i] / 255.0) : w_in[i];
    for (int kb = 0; kb < 2; ++kb)
        for (int ln = 0; ln < 64; ++ln)
            for (int j = 0; j < 4; ++j) {
                unsigned short lo[3], hi[3];
                const int k0 = 16 * kb + 8 * (ln >> 5) + 2 * j, n = ln & 31;
                wino_s3_split_host(k0 < 27 ? w[k0 * 32 + n] : 0.0f, lo);
                wino_s3_split_host(k0 + 1 < 27 ? w[(k0 + 1) * 32 + n] : 0.0f, hi);
                for (int t = 0; t < 3; ++t) w3[((kb * 3 + t) * 64 + ln) * 4 + j] = (unsigned)lo[t] | ((unsigned)hi[t] << 16);
            }
}

// does launch_conv1_direct take the kernel that fills `amax_out` for these arguments?
bool conv1_direct_fills_amax(const void *frames, int dtype, int W, const unsigned *w3, const unsigned *w3u8)
{
    return w3 && w3u8 && (dtype != DT_FRAMES_U8 || (W % 4 == 0 && (reinterpret_cast<uintptr_t>(frames) & 3) == 0));
}
int launch_conv1_direct(hipStream_t st, const void *frames, int dtype, int B, int H, int W, const float *w_packed,
                        const float *bias, const float *lut, float slope, float *out, const unsigned *w3, const unsigned *w3u8, unsigned *amax_out)
{
    if ((H & 1) || (W & 1) || B <= 0) return 2;
    Conv1Args a;
    a.amax_out = amax_out;
    a.frames = frames; a.dtype = dtype; a.B = B; a.H = H; a.W = W;
    a.w = w_packed; a.bias = bias; a.lut = lut; a.slope = slope; a.out = out;
    a.w3 = w3; a.w3u8 = w3u8;
    const int H2 = H / 2, W2 = W / 2;
    // whole rows per workgroup when there are thousands of rows (the walk hides each tile's load latency); with a few
    // frames per call shorter walks keep every CU busy
    const long long rows = (long long)B * ((H2 + 7) / 8);
    const int ntx = (W2 + 7) / 8;
    int tpw = C1_TPW;
    while (tpw > 1 && rows * ((ntx + tpw - 1) / tpw) < 2048) tpw = (tpw + 1) / 2;
    a.tpw = tpw;
    const dim3 grid((unsigned)((ntx + tpw - 1) / tpw), (unsigned)((H2 + 7) / 8), (unsigned)B);
    if (w3 && w3u8 && (dtype != DT_FRAMES_U8 || (W % 4 == 0 && (reinterpret_cast<uintptr_t>(frames) & 3) == 0))) {      // split-bf16 form (Policy::s3_conv1); uint8: the kernel reads aligned dwords of the rows -- a view at an odd byte offset takes the fp32 MFMA kernel
        const bool full = H % 16 == 0 && W % 16 == 0;
        if (dtype == DT_FRAMES_U8 && full) hipLaunchKernelGGL((conv1_s3_kernel<true, true>), grid, dim3(256), 0, st, a);
        else if (dtype == DT_FRAMES_U8) hipLaunchKernelGGL((conv1_s3_kernel<true, false>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((conv1_s3_kernel<false, false>), grid, dim3(256), 0, st, a);
    } else
        hipLaunchKernelGGL(conv1_mfma_kernel, grid, dim3(256), 0, st, a);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

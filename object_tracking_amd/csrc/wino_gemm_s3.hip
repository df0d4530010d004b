// wino_gemm_s3.hip -- the P batched GEMMs  M'[p] = V[p] (Mt x K)  *  U[p]^T (N x K)  of the F(6x6,3x3) layers on the
// BF16 matrix pipe at fp32 accuracy (the arithmetic of nn/yolo.py:conv_6..conv_22 and of nn/tracker.py's input
// convolution once they are in Winograd form).
//
// gfx950 has no fp32-rate shortcut (v_mfma_f32_32x32x2_f32 runs at 1/16 of the bf16 rate, no xf32), so each fp32
// operand is carried as THREE bf16 terms   x = x1 + x2 + x3   (x1 = bf16(x), x2 = bf16(x - x1), x3 = bf16(x - x1 - x2):
// 3 x 8 significand bits cover fp32's 24, the split is exact up to the last term's rounding, |x - x1 - x2 - x3| <= 2^-25 |x|)
// and a product is the six partial products of weight 2^0, 2^-8, 2^-16
//      u*v ~= u1 v1 + (u1 v2 + u2 v1) + (u1 v3 + u2 v2 + u3 v1)            (dropped: u2 v3, u3 v2, u3 v3 <= 2^-24 |u v|)
// accumulated in fp32 by v_mfma_f32_32x32x16_bf16, smallest terms first.  Six bf16 MFMAs do the work of sixteen fp32
// MFMAs' cycles (2.67x), with errors at the level of fp32's own product rounding (tests/test_gpu_parity.py:
// the error against float64 is no larger than the fp32 MFMA path's).
//
// Operands arrive ALREADY split, from the producers: V from the input transform (winograd.hip, split form), U packed at
// load time (wino_s3_pack_weights).  Both are K-blocked so that the 16-deep stage of a 256-row tile is ONE contiguous
// 8 KiB run per term:     [p][term 3][K/16][rows][16] bf16.
//
// Kernel: persistent, one 512-thread workgroup per CU; tile = 256 rows of V x BN (256 | 128) rows of U; k in stages of 16
// through a 3-deep LDS ring filled by global_load_lds_dwordx4 (6 x 1 KiB pieces per wave per stage) -- the next tile's
// first two stages are in flight during a tile's epilogue.  Wave (wm, wn) owns 64 rows of V x BN/2 rows of U; the MFMA
// takes U as its A operand, so a lane holds 4 consecutive n of one m and the epilogue stores 16 B.
// LDS image of one (operand, term, stage): [row][2 granules of 8 bf16], granule index XOR (row >> 3) & 1 -- with the
// ds_read_b128 lane groups of gfx950 ({0-3,12-15,20-27}, ...) every group then covers all 64 banks once.
#include "dt_internal.h"
#include <cstring>

typedef __bf16 s3_bf8 __attribute__((ext_vector_type(8)));
typedef float s3_f16 __attribute__((ext_vector_type(16)));
typedef float s3_f4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void s3_lptr_t;
typedef const __attribute__((address_space(1))) void s3_gptr_t;

#define S3_THREADS 512
#define S3_BM 256
#define S3_STAGES 3
#ifndef S3_ABLATE
#define S3_ABLATE 0      // probes (tools/micro/gemm_s3_bench.hip): 1 no DMA, 2 no operand reads, 4 no barrier
#endif

template <int BN>
__global__ __launch_bounds__(S3_THREADS) void wino_gemm_s3_kernel(GemmS3Args p)
{
    constexpr int NBW = BN / 64;                      // 32-wide n blocks per wave (BN/2 columns)
    constexpr int OP_A = 3 * BN * 32;                 // bytes of U terms per stage
    constexpr int STAGE = OP_A + 3 * S3_BM * 32;      // + V terms
    extern __shared__ __attribute__((aligned(16))) unsigned char s3_lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 3, wn = wave >> 2;
    const int KB = p.K >> 4;
    const int MT = (p.Mt + S3_BM - 1) / S3_BM, NT = p.N / BN;
    const long long ntiles = (long long)p.P * MT * NT;
    // XCD-aware order: workgroup w runs on XCD w % 8; each XCD walks a contiguous range of the n-fastest tile order, so
    // the 32 tiles resident on an XCD share their V / U panels through that XCD's L2
    const int G = gridDim.x;
    const long long first = (G % 8 == 0) ? (long long)(blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3) : blockIdx.x;

    // ---- DMA geometry: wave w moves rows [32w, 32w+32) of every (operand, term) piece ----
    const int drow = 32 * wave + (lane >> 1);
    const int dgran = (lane & 1) ^ ((drow >> 3) & 1);           // source granule for LDS slot `lane`
    const long long a_term = (long long)KB * p.Mp * 16, b_term = (long long)KB * p.Np * 16;   // elements per term
    struct Tile { long long a0, b0; int m0, n0, pz; };
    auto tile_of = [&](long long L) {
        Tile t;
        const int nt = (int)(L % NT);
        const long long r = L / NT;
        const int mt = (int)(r % MT);
        t.pz = (int)(r / MT);
        t.m0 = mt * S3_BM; t.n0 = nt * BN;
        t.a0 = (long long)t.pz * 3 * a_term + (long long)t.m0 * 16;
        t.b0 = (long long)t.pz * 3 * b_term + (long long)t.n0 * 16;
        return t;
    };
    auto issue = [&](const Tile &t, int kb, int buf) {
        if (S3_ABLATE & 1) return;
        unsigned char *dst = s3_lds + buf * STAGE + wave * 1024;
        const unsigned short *bs = p.b + t.b0 + ((long long)kb * p.Np + drow) * 16 + dgran * 8;
        const unsigned short *as = p.a + t.a0 + ((long long)kb * p.Mp + drow) * 16 + dgran * 8;
#pragma unroll
        for (int t3 = 0; t3 < 3; ++t3) {
            if (BN == 256 || wave < 4)
                __builtin_amdgcn_global_load_lds((s3_gptr_t *)(bs + t3 * b_term), (s3_lptr_t *)(dst + t3 * BN * 32), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((s3_gptr_t *)(as + t3 * a_term), (s3_lptr_t *)(dst + OP_A + t3 * S3_BM * 32), 16, 0, 0);
        }
    };
    // ---- operand read offsets (bytes inside a stage) ----
    const int rl = lane & 31, gl = lane >> 5;
    int offU[NBW], offV[2];
#pragma unroll
    for (int j = 0; j < NBW; ++j) {
        const int row = wn * (BN / 2) + 32 * j + rl;
        offU[j] = (2 * row + (gl ^ ((row >> 3) & 1))) * 16;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = wm * 64 + 32 * i + rl;
        offV[i] = OP_A + (2 * row + (gl ^ ((row >> 3) & 1))) * 16;
    }

    if (first >= ntiles) return;
    Tile cur = tile_of(first);
    long long Lnext = first + G;
    int buf_issue = 0, buf_use = 0;
    // prologue: stages 0 and 1 of the first tile
    issue(cur, 0, 0);
    if (KB > 1) issue(cur, 1, 1);
    buf_issue = KB > 1 ? 2 : 1;
    bool drain = false;       // the previous iteration issued global stores: wait for everything

    for (;;) {
        const bool has_next = Lnext < ntiles;
        Tile nxt = cur;
        if (has_next) nxt = tile_of(Lnext);
        s3_f16 acc[NBW][2];
#pragma unroll
        for (int j = 0; j < NBW; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[j][i][e] = 0.0f;

#pragma unroll 1
        for (int kb = 0; kb < KB; ++kb) {
            // stage kb of this tile has landed (the one after it may still be in flight)
            const bool more_in_flight = (kb + 1 < KB) || has_next;
            if (drain || !more_in_flight) __builtin_amdgcn_s_waitcnt(0x0f70);                      // vmcnt(0)
            else if (BN == 256 || wave < 4) __builtin_amdgcn_s_waitcnt(0x0f76);                    // vmcnt(6)
            else __builtin_amdgcn_s_waitcnt(0x0f73);                                               // vmcnt(3)
            drain = false;
            if (!(S3_ABLATE & 4)) __builtin_amdgcn_s_barrier();   // (no fence: a release fence would wait for the DMA pieces in flight) everyone's pieces of stage kb are in LDS; everyone is done with the buffer two stages back
            // refill the buffer that was read in the previous iteration
            {
                const int k2 = kb + 2;
                if (k2 < KB) issue(cur, k2, buf_issue);
                else if (has_next && k2 - KB < KB) issue(nxt, k2 - KB, buf_issue);
                if (k2 < KB || (has_next && k2 - KB < KB)) buf_issue = buf_issue == S3_STAGES - 1 ? 0 : buf_issue + 1;
            }
            const unsigned char *sb = s3_lds + ((S3_ABLATE & 2) ? 0 : buf_use * STAGE);
            buf_use = buf_use == S3_STAGES - 1 ? 0 : buf_use + 1;
            auto frag = [&](int off) {
                if (S3_ABLATE & 2) {
                    typedef int s3_i4 __attribute__((ext_vector_type(4)));
                    const s3_i4 x = {0x3f803f80 + (off & 1), 0x3f803f80 + (lane & 1), 0x3f003f80, 0x3f803f00 + kb};
                    return __builtin_bit_cast(s3_bf8, x);
                }
                return *reinterpret_cast<const s3_bf8 *>(sb + off);
            };
            s3_bf8 v[2][3];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int t3 = 0; t3 < 3; ++t3) v[i][t3] = frag(offV[i] + t3 * S3_BM * 32);
#pragma unroll
            for (int j = 0; j < NBW; ++j) {
                s3_bf8 u[3];
#pragma unroll
                for (int t3 = 0; t3 < 3; ++t3) u[t3] = frag(offU[j] + t3 * BN * 32);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    s3_f16 c = acc[j][i];
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(u[2], v[i][0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(u[1], v[i][1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(u[0], v[i][2], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(u[1], v[i][0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(u[0], v[i][1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(u[0], v[i][0], c, 0, 0, 0);
                    acc[j][i] = c;
                }
            }
        }
        // a tile with KB == 1 leaves the ring one stage ahead only: handled by the generic wait logic above

        // ---- epilogue: lane holds, per block, n = n0 + 8q + 4*(lane>>5) + (0..3) of row m = m0 + (lane & 31) ----
        float *cz = p.c + (long long)cur.pz * p.c_ps;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = cur.m0 + wm * 64 + 32 * i + rl;
            if (m < p.Mt) {
                float *row = cz + (long long)m * p.ldc + cur.n0 + wn * (BN / 2) + 4 * gl;
#pragma unroll
                for (int j = 0; j < NBW; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        s3_f4 o;
                        o[0] = acc[j][i][4 * q]; o[1] = acc[j][i][4 * q + 1]; o[2] = acc[j][i][4 * q + 2]; o[3] = acc[j][i][4 * q + 3];
                        __builtin_nontemporal_store(o, reinterpret_cast<s3_f4 *>(row + 32 * j + 8 * q));
                    }
            }
        }
        if (!has_next) break;
        drain = true;
        cur = nxt;
        Lnext += G;
    }
}

// executed bf16 MFMA FLOPs of one launch (six partial products per multiply, whole tiles)
double wino_gemm_s3_flops(const GemmS3Args &a) { return 12.0 * a.P * (double)a.Mt * a.K * a.N; }

bool wino_gemm_s3_usable(int Mt, int K, int N)
{
    return Mt > 0 && K >= 32 && K % 16 == 0 && N >= 128 && N % 128 == 0;
}

int launch_wino_gemm_s3(hipStream_t st, const GemmS3Args &a, int cus)
{
    if (!wino_gemm_s3_usable(a.Mt, a.K, a.N) || a.Mp % S3_BM || a.Mp < a.Mt || a.ldc % 4 || a.P <= 0) return 2;
    static PerDeviceOnce attr256, attr128;
    const bool wide = a.N % 256 == 0 && a.Np % 256 == 0;
    if (!wide && a.Np % 128) return 2;
    const int BN = wide ? 256 : 128;
    const size_t lds = (size_t)S3_STAGES * (3 * BN * 32 + 3 * S3_BM * 32);
    if (wide) {
        if (attr256.first()) {
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(wino_gemm_s3_kernel<256>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return 1;
            attr256.done();
        }
    } else if (attr128.first()) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(wino_gemm_s3_kernel<128>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return 1;
        attr128.done();
    }
    const long long tiles = (long long)a.P * ((a.Mt + S3_BM - 1) / S3_BM) * (a.N / BN);
    if (cus <= 0) {
        static int cu_of[64];
        static PerDeviceOnce once;
        if (once.first()) {
            hipDeviceProp_t prop;
            if (hipGetDeviceProperties(&prop, once.dev) != hipSuccess) return 1;
            cu_of[once.dev] = prop.multiProcessorCount;
            once.done();
        }
        cus = cu_of[once.dev];
    }
    long long grid = cus;      // one 8-wave workgroup per CU (108 / 144 KiB of LDS), persistent over the tiles
    if (grid > tiles) grid = tiles;
    if (wide) hipLaunchKernelGGL(wino_gemm_s3_kernel<256>, dim3((unsigned)grid), dim3(S3_THREADS), lds, st, a);
    else hipLaunchKernelGGL(wino_gemm_s3_kernel<128>, dim3((unsigned)grid), dim3(S3_THREADS), lds, st, a);
    return hipGetLastError() == hipSuccess ? 0 : 1;
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

// wino4s_fused.hip -- second-generation fused Winograd F(4x4,3x3) kernel for the early wide 3x3 layers
// (conv_2: 32 -> 64 at 208x208 (+pool), conv_3 / conv_5: 64 -> 128 at 104x104, conv_6 / conv_8: 128 -> 256 at 52x52;
// models_detection/KerasYOLO.py:285-320).  Same mathematics as wino4_fused.hip (V and M' never leave the CU); what
// changed is how the operands reach the matrix cores:
//
//   * the B operand U = G g Gt no longer streams through each wave's registers from L2 (the ~17 B/clk/CU limit that
//     capped wino4_fused at 38 % of the MFMA peak): a stage's U slice is brought into LDS ONCE per
//     workgroup with asynchronous global->LDS DMA (global_load_lds_dwordx4, 36 x 1 KiB pieces per 4-channel stage,
//     double-buffered) and all eight waves read their fragments from there with conflict-free ds_read_b128;
//   * every wave owns ALL 36 Winograd positions of one block of 16 tiles x 16 output channels (144 accumulators), so
//     the output transform At M' A happens in the wave's own registers -- no cross-wave reduction through LDS, and
//     the epilogue is a short in-register pass followed by 64-byte-segment stores;
//   * the input patches also arrive by DMA (16 bytes = 4 channels of one pixel per lane); the LDS image is padded by
//     one 16-byte slot per 4 pixels and 41 slots per row, which makes the 6x6 window reads of the input transform
//     conflict-free without any per-read address arithmetic (slot = 164 ty + 9 tx + 41 i + 2 j + (j >> 2) + half);
//   * the workgroup is PERSISTENT: it walks (block pair, 64-channel slice) items; launch and tail effects of ~70 k
//     one-shot workgroups are gone.
//
// Workgroup = 8 waves = 2 blocks (4x4 tiles of 4x4 pixels each) x 4 column groups of 16 channels; v_mfma_f32_16x16x4_f32
// (row = tile, column = channel, k = input channel).  K runs in stages of 4 input channels (one MFMA k-step), one
// barrier per stage:
//      stage s :  DMA U(s+1) -> Ubuf[(s+1)&1];   DMA patch(c+1) -> Pbuf (8 channels, every second stage);
//                 4 of the 8 waves: V(s+1) = Bt d B for (block, xi-half) from the patch, -> Vbuf[(s+1)&1];
//                 all waves: 36 MFMAs on V(s), U(s);   vmcnt(0); barrier.
// LDS (157.7 KB of 160): U 2 x 36.9 KB | V 2 x 18.4 KB | patch 2 x 24.6 KB.
// fp32 throughout; rounding identical in kind to wino4_fused / winograd.hip TS = 4 (products summed in another order).
#include "dt_internal.h"

typedef __attribute__((address_space(1))) const void s4_gptr_t;
typedef __attribute__((address_space(3))) void s4_lptr_t;

#ifdef DT_S4_TIMING
// debug build only (tools/s4_timing.py): per workgroup / wave / item cycle stamps and per-phase cycle sums
#define S4_TT_WG 64
#define S4_TT_ITEMS 8
#define S4_TT_SLOTS 8      // 0 start, 1 prologue end, 2 loop end, 3 epilogue end, 4 sum(transform), 5 sum(mfma block), 6 sum(wait+barrier), 7 sum(dma issue)
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

__device__ __forceinline__ void s4_at(float *m, int st)      // At (4x6): 6 inputs -> 4 outputs in the first 4 slots
{
    const float a = m[st] + m[2 * st], b = m[st] - m[2 * st], c = m[3 * st] + m[4 * st], e = m[3 * st] - m[4 * st];
    const float y0 = m[0] + a + c, y1 = b + 2.0f * e, y2 = a + 4.0f * c, y3 = b + 8.0f * e + m[5 * st];
    m[0] = y0; m[st] = y1; m[2 * st] = y2; m[3 * st] = y3;
}

typedef float f32x2 __attribute__((ext_vector_type(2)));   // NOT HIP's float2: LDS accesses through the struct type carry TBAA
                                                           // info that makes hipcc wait vmcnt(0) for every LDS-DMA in flight

// Bt d B restricted to three xi rows (HALF 0: rows 0-2, HALF 1: rows 3-5) of one 6x6 window; pl = window pixel (0,0) of
// this lane's channel in the padded patch image, o = this lane's slot of pair 0 of the half in the V stage
template <int HALF>
__device__ __forceinline__ void s4_transform_half(const float *pl, float *o)
{
    float t[3][6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        float d[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) d[i] = pl[(S4_PROW * i + 2 * j + (j >> 2)) * 4];
        if (HALF == 0) {           // Bt rows 0, 1, 2
            t[0][j] = 4.0f * d[0] - 5.0f * d[2] + d[4];
            t[1][j] = -4.0f * (d[1] + d[2]) + d[3] + d[4];
            t[2][j] = 4.0f * (d[1] - d[2]) - d[3] + d[4];
        } else {                   // Bt rows 3, 4, 5
            t[0][j] = 2.0f * (d[3] - d[1]) - d[2] + d[4];
            t[1][j] = 2.0f * (d[1] - d[3]) - d[2] + d[4];
            t[2][j] = 4.0f * d[1] - 5.0f * d[3] + d[5];
        }
    }
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
    const int nst = p.Cin >> 2;                             // MFMA stages of 4 input channels
    const int nblk = p.B * p.nby * p.nbx;
    const int npair = (nblk + 1) >> 1;
    const int nitems = npair * (p.N >> 6);

    // ---- V production role (even stages: waves 0-3, odd stages: waves 4-7): (block, xi half) ----
    const int vblk = wave & 1, vhalf = (wave >> 1) & 1;
    const int vtile = (lane & 7) | ((lane >> 5) << 3), vk = (lane >> 3) & 3;
    const int vty = vtile >> 2, vtx = vtile & 3;
    const int vsrc = (vblk * S4_PBLK + 164 * vty + 9 * vtx) * 4 + vk;       // float index of window pixel (0,0), half 0
    const int vdst = (vblk * 64 + vk * 16 + vtile) * 2 + vhalf * (9 * 256); // float index in a V stage: pg2 = 9 half + ...

    // ---- patch DMA role: pieces wave, wave + 8, wave + 16 of the 24 (12 per block) ----
    int poff[3];            // float offset of this lane's 16 bytes relative to the block's pixel (0,0), channel 0; < 0: zeros
    int pblk[3];
    auto patch_setup = [&](int j0) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int piece = wave + 8 * i;
            const int b01 = piece / 12;
            const int P = (piece - 12 * b01) * 64 + lane;   // slot within the block image
            const int py = P / S4_PROW, rr = P - py * S4_PROW;
            const int g = rr / 9, r9 = rr - 9 * g;
            const int px = 4 * g + (r9 >> 1), hf = r9 & 1;
            const int j = j0 + b01;
            const int bx = j % p.nbx, by = (j / p.nbx) % p.nby, b = j / (p.nbx * p.nby);
            const int hh = by * 16 - 1 + py, ww = bx * 16 - 1 + px;
            const bool ok = py < 18 && r9 < 8 && px < 18 && j < nblk && hh >= 0 && hh < p.H && ww >= 0 && ww < p.W;
            // offsets fit 31 bits: the launcher checks B * in_bs < 2^31 floats
            poff[i] = ok ? (int)((long long)b * p.in_bs + ((long long)hh * p.W + ww) * p.in_ld + hf * 4) : -1;
            pblk[i] = b01;
        }
    };
    auto patch_dma = [&](int c, int buf) {              // patch stage c = channels 8c .. 8c+7
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const float *src = poff[i] >= 0 ? p.in + poff[i] + 8 * c : p.zeros;
            float *dst = Pb + buf * S4_PBUF + (wave + 8 * i) * 256;
            __builtin_amdgcn_global_load_lds((s4_gptr_t *)src, (s4_lptr_t *)dst, 16, 0, 0);
        }
    };
    // U DMA: pieces wave + 8 i of the stage's 36
    auto u_dma = [&](const float *ustage, int buf) {
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int piece = wave + 8 * i;
            if (piece < 36)
                __builtin_amdgcn_global_load_lds((s4_gptr_t *)(ustage + piece * 256 + lane * 4),
                                                 (s4_lptr_t *)(Ub + buf * S4_UBUF + piece * 256), 16, 0, 0);
        }
    };
    // input transform of MFMA stage s (channels 4 s .. 4 s + 3 = half s & 1 of patch stage s >> 1) into Vbuf[s & 1]:
    // this lane's (tile, channel), xi rows 3 vhalf .. 3 vhalf + 2, all six nu (wave-uniform branch on the half)
    auto transform = [&](int s) {
        const float *pl = Pb + ((s >> 1) & 1) * S4_PBUF + vsrc + (s & 1) * 4;
        float *o = Vb + (s & 1) * S4_VBUF + vdst;
        if (vhalf == 0) s4_transform_half<0>(pl, o);
        else s4_transform_half<1>(pl, o);
    };

    const float *const a_base = Vb + (blk * 64 + lane) * 2;            // + stage buffer + pg2 * 256
    const float *const b_base = Ub + (wn * 64 + lane) * 4;             // + stage buffer + pg * 1024

#pragma unroll 1
    for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
        const int nq = item / npair, pair = item - nq * npair;
        const int j0 = 2 * pair;
        const float *const u_item = p.u + (long long)nq * nst * S4_UBUF;
#ifdef DT_S4_TIMING
        const int tt_i = (item - (int)blockIdx.x) / (int)gridDim.x;
        unsigned long long tt_tr = 0, tt_mm = 0, tt_bw = 0, tt_dm = 0;
#endif
        S4_PUT(0, S4_NOW());
        patch_setup(j0);

        f32x4 acc[36];
#pragma unroll
        for (int i = 0; i < 36; ++i) acc[i] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

        // ---- prologue: patch stage 0, U stage 0; V(0) ----
        patch_dma(0, 0);
        u_dma(u_item, 0);
        __builtin_amdgcn_s_waitcnt(0x0f70);     // vmcnt(0)
        __syncthreads();
        if (wave < 4) transform(0);
        if (nst > 2) patch_dma(1, 1);
        __builtin_amdgcn_s_waitcnt(0x0f70);
        __syncthreads();
        S4_PUT(1, S4_NOW());

#pragma unroll 1
        for (int s = 0; s < nst; ++s) {
            // data movement of the stages ahead
            [[maybe_unused]] const unsigned long long c0 = S4_NOW();
            if (s + 1 < nst) u_dma(u_item + (long long)(s + 1) * S4_UBUF, (s + 1) & 1);
            // patch stage c is read by the transforms of MFMA stages 2c, 2c+1, which run during stages 2c-1 and 2c; its
            // buffer is free again after stage 2c, so patch c+2 is fetched during stage 2c+1
            if ((s & 1) && ((s + 3) >> 1) < (nst >> 1)) patch_dma((s + 3) >> 1, ((s + 3) >> 1) & 1);
            [[maybe_unused]] const unsigned long long c1 = S4_NOW();
            if (s + 1 < nst && ((wave >> 2) == ((s + 1) & 1))) transform(s + 1);
            [[maybe_unused]] const unsigned long long c2 = S4_NOW();
            // 36 MFMAs: positions in quads; the operands of quad g+1 are requested BEFORE the MFMAs of quad g are issued
            // (pinned with sched_barrier: left alone hipcc sinks the reads behind the MFMAs and waits for them at once)
            const float *va = a_base + (s & 1) * S4_VBUF;
            const float *ua = b_base + (s & 1) * S4_UBUF;
            f32x2 a0[2], a1[2];
            f32x4 bq[2];
            a0[0] = *reinterpret_cast<const f32x2 *>(va);
            a1[0] = *reinterpret_cast<const f32x2 *>(va + 256);
            bq[0] = *reinterpret_cast<const f32x4 *>(ua);
#pragma unroll
            for (int g = 0; g < 9; ++g) {
                const int c = g & 1, n = c ^ 1;
                if (g + 1 < 9) {
                    a0[n] = *reinterpret_cast<const f32x2 *>(va + (2 * g + 2) * 256);
                    a1[n] = *reinterpret_cast<const f32x2 *>(va + (2 * g + 3) * 256);
                    bq[n] = *reinterpret_cast<const f32x4 *>(ua + (g + 1) * 1024);
                }
                __builtin_amdgcn_sched_barrier(0);
                acc[4 * g + 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[c].x, bq[c][0], acc[4 * g + 0], 0, 0, 0);
                acc[4 * g + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[c].y, bq[c][1], acc[4 * g + 1], 0, 0, 0);
                acc[4 * g + 2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[c].x, bq[c][2], acc[4 * g + 2], 0, 0, 0);
                acc[4 * g + 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[c].y, bq[c][3], acc[4 * g + 3], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            [[maybe_unused]] const unsigned long long c3 = S4_NOW();
            __builtin_amdgcn_s_waitcnt(0x0f70);     // this wave's DMA pieces have landed
            __syncthreads();
#ifdef DT_S4_TIMING
            tt_dm += c1 - c0; tt_tr += c2 - c1; tt_mm += c3 - c2; tt_bw += S4_NOW() - c3;
#endif
        }
        S4_PUT(2, S4_NOW());

        // ---- epilogue: At M' A per (tile, channel) in registers; C/D row = 4 kq + e -> tile (ty = kq, tx = e), col = channel ----
        const int j = j0 + blk;
        if (j < nblk) {
            const int bx = j % p.nbx, by = (j / p.nbx) % p.nby, b = j / (p.nbx * p.nby);
            const int ch = nq * 64 + wn * 16 + r16;
            const float bias = p.bias[ch];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float m[36];
#pragma unroll
                for (int i = 0; i < 36; ++i) m[i] = acc[i][e];
#pragma unroll
                for (int nu = 0; nu < 6; ++nu) s4_at(m + nu, 6);        // over xi -> rows a = 0..3 (slots 6 a + nu)
#pragma unroll
                for (int a = 0; a < 4; ++a) s4_at(m + 6 * a, 1);        // over nu -> cols c = 0..3
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const float z = m[6 * a + c] + bias;
                        m[6 * a + c] = z > 0.0f ? z : z * p.slope;
                    }
                if (!POOL) {
#pragma unroll
                    for (int a = 0; a < 4; ++a) {
                        const int oy = by * 16 + 4 * kq + a;
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const int ox = bx * 16 + 4 * e + c;
                            if (oy < p.H && ox < p.W)
                                p.out[(long long)b * p.out_bs + ((long long)oy * p.W + ox) * p.out_ld + ch] = m[6 * a + c];
                        }
                    }
                } else {
                    const int H2 = p.H >> 1, W2 = p.W >> 1;
#pragma unroll
                    for (int a2 = 0; a2 < 2; ++a2)
#pragma unroll
                        for (int c2 = 0; c2 < 2; ++c2) {
                            const float mx = fmaxf(fmaxf(m[6 * (2 * a2) + 2 * c2], m[6 * (2 * a2) + 2 * c2 + 1]),
                                                   fmaxf(m[6 * (2 * a2 + 1) + 2 * c2], m[6 * (2 * a2 + 1) + 2 * c2 + 1]));
                            const int py = by * 8 + 2 * kq + a2, qx = bx * 8 + 2 * e + c2;
                            if (py < H2 && qx < W2)
                                p.out2[(((long long)b * H2 + py) * W2 + qx) * p.out2_ld + ch] = mx;
                        }
                }
            }
        }
#ifdef DT_S4_TIMING
        S4_PUT(3, S4_NOW()); S4_PUT(4, tt_tr); S4_PUT(5, tt_mm); S4_PUT(6, tt_bw); S4_PUT(7, tt_dm);
#endif
        // no barrier here: every LDS read of this item finished before the last stage's barrier, so a wave that is done
        // with its stores may start the next item's prologue DMA while the others are still in their epilogue
    }
}

int launch_wino4s_fused(hipStream_t st, const Wino4FusedArgs &a_in, const float *zeros)
{
    Wino4FusedArgs a = a_in;
    if (a.B <= 0 || a.Cin % 8 || a.N % 64 || a.in_ld % 4) return 2;
    const bool pool = a.out2 != nullptr;
    if (pool && ((a.H | a.W) & 1)) return 2;
    if (pool == (a.out != nullptr)) return 2;       // exactly one of the two outputs
    if ((long long)a.B * a.in_bs >= (1ll << 31)) return 2;   // 32-bit patch offsets
    a.nby = (a.H + 15) / 16;
    a.nbx = (a.W + 15) / 16;
    const long long blocks = (long long)a.B * a.nby * a.nbx;
    if (blocks >= (1ll << 30)) return 2;
    a.zeros = zeros;
    const size_t lds = (size_t)S4_LDS_FLOATS * sizeof(float);      // 157,696 B
    static PerDeviceOnce attr;
    static int cus[64];
    if (attr.first()) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(wino4s_fused_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void *>(wino4s_fused_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return 1;
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, attr.dev) != hipSuccess || n <= 0) n = 256;
        cus[attr.dev] = n;
        attr.done();
    }
    const long long items = ((blocks + 1) / 2) * (a.N / 64);
    long long grid = cus[attr.dev];                 // one 8-wave workgroup per CU (157 KB of LDS), persistent over the items
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

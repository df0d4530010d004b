// Internal declarations shared by the HIP translation units of libmi355_dt.so.
// gfx950 (MI355X, CDNA4) only -- no other target is supported or compiled.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <map>
#include <mutex>
#include <functional>
#include <string>
#include <vector>

#include "../../include/mi355_dt.h"

// One-time, PER-DEVICE set-up of a launch site (hipFuncSetAttribute for large dynamic LDS, small constant buffers):
// kernel attributes and allocations belong to a device, so a process that drives several GPUs must repeat them on
// each.  `ensure(&dev, setup)` runs `setup(dev)` once per device, serialised by a mutex, and hands the calling thread's
// device id back to the CALLER (nothing device-specific is kept in shared state, so two host threads on different
// devices cannot see each other's id).  It returns non-zero -- the caller fails its launch -- when the set-up failed,
// the device cannot be determined, or it lies beyond the 64 devices tracked (never aliased onto device 0).
struct PerDeviceOnce {
    std::atomic<unsigned long long> mask{0};
    std::mutex mu;
    template <class F> int ensure(int *dev_out, F &&setup)
    {
        int dev = -1;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) return 1;
        if (dev_out) *dev_out = dev;
        if ((mask.load(std::memory_order_acquire) >> dev) & 1ull) return 0;
        std::lock_guard<std::mutex> lk(mu);
        if ((mask.load(std::memory_order_acquire) >> dev) & 1ull) return 0;
        const int rc = setup(dev);
        if (rc == 0) mask.fetch_or(1ull << dev, std::memory_order_release);
        return rc;
    }
};

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---------------------------------------------------------------------------
// implicit-GEMM convolution (conv_igemm.hip)
// ---------------------------------------------------------------------------
enum { ORD_LINEAR = 0, ORD_QUAD = 1 };
enum { EPI_PLAIN = 0, EPI_POOL = 1, EPI_POOL_BOTH = 2, EPI_S2D = 3, EPI_GATES = 4, EPI_PARTIAL = 5 };

struct ConvArgs {
    // input activation: pixel (b,h,w) at in + b*in_bs + (h*W+w)*in_ld, Cin floats read
    const float *in;
    long long in_bs;
    int in_ld;
    // packed weights [Npad][K], k = ((ci/32)*taps + tap)*32 + ci%32 (k contiguous); bias [Npad] or null
    const float *wt;
    const float *bias;
    // primary output (dense unless EPI_GATES): row r at out + r*out_ld (+ col)
    float *out;
    long long out_bs;
    int out_ld;
    // secondary output: pooled tensor for EPI_POOL_BOTH
    float *out2;
    int out2_ld;
    // EPI_GATES: z = acc + xproj; c updated in place; h written to out
    const float *xproj;
    long long xp_bs;
    int xp_ld;
    float *cstate;
    long long c_bs;
    int c_ld;
    int B, H, W, Cin, N, M, K;
    int npad;     // rows of wt that exist (0 = N rounded up to 128); 256-wide column tiles need N rounded up to 256
    float slope;  // LeakyReLU slope; 1.0f = linear
    int act;      // 0: LeakyReLU(slope), 1: sigmoid (Dense heads)
    const float *zeros; // >= 16 B of device zeros: source of out-of-image taps (set by launch_conv_igemm)
    int ksplit;    // EPI_PARTIAL: number of K splits (grid.y); out = slab [ksplit][M][out_ld]
    int tile_gn;   // column tiles per group in the tile order (0 = all)
    int xcd_remap; // 1: give each XCD a contiguous range of tiles (set by launch_conv_igemm)
    // batched GEMMs (Winograd positions): grid.z problems, blockIdx.z advances in / wt / out by these (floats)
    int zbatch;
    long long z_in, z_wt, z_out;
    int force_cfg; // > 0: tile configuration + 1 forced by the caller's policy (Policy::conv_cfg); 0: none
    int no_persist; // 1: one tile per workgroup even where the persistent tile loop applies (Policy::persist = 0, A/B runs)
    unsigned *amax_out; // EPI_PLAIN (no split-K) / EPI_POOL / EPI_S2D: non-null = max |x| of the stored outputs into this slot (dt_amax_publish)
    int gn_default; // > 0: column-group width + 1 used when tile_gn is 0 (Policy::tile_gn, A/B runs); 0: the per-layer default
};

// Tile configurations of the MFMA kernel
enum { CFG_128x128 = 0, CFG_128x64 = 1, CFG_256x128 = 2, CFG_256x256 = 3, CFG_64x128 = 4 };

int launch_conv_igemm(hipStream_t st, const ConvArgs &a, int ks, int order, int epi, int cfg);
int launch_splitk_reduce(hipStream_t st, const float *slab, int S, long long M, int N, const float *bias, float slope,
                         float *out, int out_ld);

// host-side packing: Keras HWIO kernel -> [npad][ks*ks*cin_dst], k contiguous in the kernel's chunk order.
//   cin_map[cin_dst]: source input channel or -1 (zero);  n_map[npad]: source
//   output channel or -1 (zero row);  scale[cout_src] multiplies each output
//   channel (folded BatchNorm) or null.
void pack_conv_weights(const float *hwio, int ks, int cin_src, int cout_src, const int *cin_map,
                       int cin_dst, const int *n_map, int npad, const float *scale, float *dst);

// ---------------------------------------------------------------------------
// Winograd F(2x2,3x3) transforms around the batched MFMA GEMM (winograd.hip)
// ---------------------------------------------------------------------------
struct WinoArgs {
    // geometry: B images of H x W; th x tw tiles of ts x ts outputs per image (ts = 2 or 4); Mt = B*th*tw;
    // P = (ts+2)^2 Winograd positions
    int B, H, W, th, tw, Mt, ts;
    int g;   // frames per side of the virtual mosaic the tiles live on (1: one frame; winograd.hip:vpixel)
    int coop;  // F(6x6) lane-cooperative transform kernels: -1 small launches only (default), 0 never, 1 always (Policy::wino_coop)
    int grid_in, grid_out, thr_out;   // A/B knobs (Policy::wino_grid_in / wino_grid_out / wino_thr_out): workgroup cap of the big input / output transform launches
                                      // and the output kernel's workgroup size; 0 = the built-in choice
    // input transform: in (NHWC, pixel stride in_ld, image stride in_bs), C channels -> v [P][Mt][C]
    const float *in;
    long long in_bs;
    int in_ld, C;
    float *v;
    unsigned short *v_s3;   // non-null: V as three bf16 terms [P][3][C/16][Mp][16] for wino_gemm_s3.hip instead of v (ts == 6, C % 32 == 0)
    int Mp;                 // rows of that layout (Mt rounded up to the GEMM's row tile)
    int nt;                 // 2: v_s3 takes TWO fp16 terms [P][2][C/16][Mp][16] of the scaled V (wino_gemm_s3.hip's fp16 form); 0 / 3: three bf16 terms
    const unsigned *amax;   // nt = 2: the max-|x| slot (DT_AMAX_SUB words) of the input tensor
    unsigned *amax_out;     // output transform (plain epilogue, thread-per-item kernels): non-null = take max |x| over the values STORED to out2 if it is set,
                            // else to out, into this slot (dt_amax_publish); the caller zeroes the slot
    // output transform: m [P][Mt][m_ld], N columns -> out (full resolution, may be null) / out2 (2x2 pooled, may be null)
    const float *m;
    int m_ld, N;
    const float *bias;
    const float *bias16;    // non-null (plain epilogue, slope 1): a CORRECTION to `bias` for pixels on the image border, [16][N] indexed by
                            // (h == 0) | (h == H - 1) << 1 | (w == 0) << 2 | (w == W - 1) << 3 (entry 0 unused: interior pixels take `bias` alone) -- the
                            // merged ConvLSTM input projection: conv_23's bias reaches the projection through the taps that lie inside the image only
    float slope;
    float *out;
    long long out_bs;
    int out_ld;
    float *out2;
    int out2_ld;
    // gate variant (ConvLSTM2D): xproj / cstate as in ConvArgs
    const float *xproj;
    long long xp_bs;
    int xp_ld;
    float *cstate;
    long long c_bs;
    int c_ld;
};
// fused Winograd F(4x4,3x3) for the early layers (Cin 32 / 64 / 128 -> Cout 64 / 128 / 256: conv_2, conv_3, conv_5; conv_6 / conv_8 on request), wino4s_fused.hip
struct Wino4FusedArgs {
    const float *in;     // NHWC, pixel stride in_ld, frame stride in_bs
    long long in_bs;
    int in_ld;
    int B, H, W, Cin, N;
    const float *u;      // wino4s_fused_pack layout
    const float *bias;   // [N]
    float slope;
    float *out;          // full-resolution output (pixel stride out_ld, frame stride out_bs) or null
    long long out_bs;
    int out_ld;
    float *out2;         // 2x2 max-pooled output [B][H/2][W/2][out2_ld] or null (exactly one of out / out2)
    int out2_ld;
    int nby, nbx;        // set by the launcher
    const float *zeros;  // >= 16 B of device zeros: DMA source of out-of-image patch pixels (wino4s_fused.hip)
    unsigned *amax_out;  // wino4s_fused.hip: non-null = max |x| over the stored outputs into this slot (dt_amax_publish)
};
// U staged through LDS by DMA, in-register output transform, persistent over (block pair, 64-channel slice) items
int launch_wino4s_fused(hipStream_t st, const Wino4FusedArgs &a, const float *zeros);
void wino4s_fused_pack(const float *u36, int npad, int cin, int cout, float *dst);
// direct 3x3 convolution in the two-term fp16 form (conv3_h2.hip): conv_2 / conv_3 / conv_5's shapes
struct Conv3H2Args {
    const float *in;           // NHWC fp32, pixel stride in_ld, frame stride in_bs
    long long in_bs;
    int in_ld;
    int B, H, W, Cin, N, Np;   // Cin % 32 == 0; N % 64 == 0 output channels, Np rows in w / bias
    const unsigned short *w;   // two fp16 terms of the scaled packed weights [2][9 Cin / 16][Np][16], k = ((ci / 32) * 9 + tap) * 32 + ci % 32
    const float *pscale;       // [1]: 1 / the weights' power of two
    const unsigned *amax;      // max-|x| slot of the input tensor
    const float *bias;         // [Np]
    float slope;
    float *out;                // full-resolution output (pixel stride out_ld, frame stride out_bs) or null
    long long out_bs;
    int out_ld;
    float *out2;               // 2x2 max-pooled output [B][H/2][W/2][out2_ld] or null (exactly one of out / out2)
    int out2_ld;
    const float *zeros;        // >= 16 B of device zeros: where the loads of out-of-image patch pixels point
    unsigned *amax_out;        // non-null: max |x| of the stored outputs into this slot
    int tiles_x, tiles_y;      // set by the launcher
    // the 1x1 layer fused behind (N = 128, no pooling: conv_3 -> conv_4): non-null w1 = apply it to the tile before anything is stored; `out` then is ITS
    // output [..][N1] (out_ld / out_bs describe it), amax_out its maximum
    const unsigned short *w1;  // two fp16 terms of the scaled 1x1 weights [2][N / 16][Np1][16]
    const float *pscale1;      // [1]
    const float *bias1;        // [N1]
    int N1, Np1;               // N1 <= 64 output channels, Np1 rows in w1
    float slope1;
};
int launch_conv3_h2(hipStream_t st, const Conv3H2Args &a);
bool conv3_h2_usable(const Conv3H2Args &a);
double conv3_h2_flops(const Conv3H2Args &a);
// split-bf16 batched GEMM of the F(6x6,3x3) layers (wino_gemm_s3.hip): fp32 operands as three bf16 terms, six MFMAs per product
struct GemmS3Args {
    const unsigned short *a;   // V terms  [P][3][K/16][Mp][16]   (winograd.hip split input transform) -- the Winograd GEMMs; null for a 1x1 layer:
    const float *a_f32;        // ... whose A operand is the producing layer's fp32 activation as it lies, [Mt][a_ld] (P = 1), split into its terms by the
    int a_ld;                  //     kernel when a wave reads its fragment
    const unsigned short *b;   // U terms  [P][3][K/16][Np][16]   (wino_s3_pack_weights)
    float *c;                  // M'       [P] planes of [Mt][ldc], plane stride c_ps floats
    long long c_ps;
    int P, Mt, Mp, N, Np, K, ldc;
    // a 1x1 layer through this kernel: bias as one extra K stage -- ones = [256][16] FLOATS, rows (1, 0, .., 0),
    // bias_s3 = [3][Np][16] terms of (bias[n], 0, .., 0); both null for the Winograd GEMMs.  act: LeakyReLU(slope) in the epilogue
    const unsigned short *ones, *bias_s3;
    int act;
    float slope;
    int half;                  // 0: the launcher chooses between 256-row tiles (one workgroup per CU) and 128-row tiles (two per CU); 1 / -1: force
    int waves;                 // 8 / 4 waves per workgroup (64 x BN/2 or 128 x BN/2 per wave); 0 = the default (Policy::s3_waves)
    unsigned long long *dbg;   // -DS3_TIMING builds of the micro-benchmark only: per-wave wait cycles; otherwise null
    // the fp16 form (round 6): nt = 2 -- a / b hold TWO fp16 terms (hi, lo) of SCALED operands, [P][2][K/16][rows][16], three products per multiply
    int nt;                    // 0 / 3: three bf16 terms; 2: two fp16 terms
    const float *pscale;       // nt = 2: [P] epilogue factor of position p = 1 / (U's scale[p] * the static part of V's scale[p])
    const unsigned *amax;      // nt = 2: the DT_AMAX_SUB-word max-|x| slot of the tensor V was made from (bits of a non-negative float): V carries
                               //         dt_h2_base(amax) (x the static per-position factor), the epilogue multiplies by dt_h2_base_inv(amax)
    const float *bias;         // nt = 2, 1x1 form: fp32 bias [N] or null, added in the epilogue (ones / bias_s3 stay null)
    unsigned *amax_out;        // 1x1 form: non-null = max |x| over the stored outputs into this slot (dt_amax_publish)
};
// ---- scaling of the fp16 form's operands (wino_gemm_s3.hip header) ----------------------------------------------------------------
// max |x| of a tensor is kept as the integer bits of a non-negative float in a slot of DT_AMAX_SUB words (producers spread their atomicMax
// over the sub-slots; a reader takes the maximum).  base = 2^(14 - floor(log2 amax)): amax * base is in [2^14, 2^15), a factor 2 below
// fp16's largest finite value 65504.  The exponent field is clamped to [27, 240] (amax = 0 -> base 2^114 on zeros; nothing overflows).
#define DT_AMAX_SUB 16
// The sub-words of a slot lie on DISTINCT 128-byte lines (DT_AMAX_LINE words apart): the first resident round of a producer finds the slot at zero and every one
// of its waves performs the atomicMax -- on one line those serialise at one L2 channel (~12 ns each: 25 us for the ~2,000 waves of a small launch; round 6 measured
// it as +0.36 ms per batch-8 forward over 13 output transforms), on sixteen lines they spread over sixteen channels
#define DT_AMAX_LINE 32
#define DT_AMAX_WORDS (DT_AMAX_SUB * DT_AMAX_LINE)      // words per slot
__host__ __device__ inline unsigned dt_h2_expfield(unsigned amax_bits)
{
    const unsigned e = (amax_bits >> 23) & 0xffu;
    return e < 27u ? 27u : (e > 240u ? 240u : e);
}
__host__ __device__ inline float dt_h2_from_field(unsigned f)
{
    const unsigned u = f << 23;
    float x;
    __builtin_memcpy(&x, &u, 4);
    return x;
}
__host__ __device__ inline float dt_h2_base(unsigned amax_bits) { return dt_h2_from_field(268u - dt_h2_expfield(amax_bits)); }       // 2^(14 - e)
__host__ __device__ inline float dt_h2_base_inv(unsigned amax_bits) { return dt_h2_from_field(dt_h2_expfield(amax_bits) - 14u); }   // 2^(e - 14)
// static part of V's scale: the 1-D input transform's rows have absolute coefficient sums 12.5 / 7.5 / 15 (F(6,3)) and 10 / 6 (F(4,3));
// row i of Bt d B is scaled by the power of two that brings its sum below 1, so |V[p] * f[xi] * f[nu]| < max |d|
__host__ __device__ inline float dt_h2_rowfac(int ts, int i)
{
    if (ts == 6) return (i == 3 || i == 4) ? 0.125f : 0.0625f;
    if (ts == 4) return (i == 3 || i == 4) ? 0.125f : 0.0625f;
    return 1.0f;
}
#ifdef __HIPCC__
__device__ __forceinline__ unsigned dt_amax_read(const unsigned *slot)      // wave-uniform
{
    unsigned v = slot[(threadIdx.x & (DT_AMAX_SUB - 1)) * DT_AMAX_LINE];
#pragma unroll
    for (int o = DT_AMAX_SUB / 2; o; o >>= 1) {
        const unsigned w = (unsigned)__shfl_xor((int)v, o);
        v = w > v ? w : v;
    }
    return (unsigned)__builtin_amdgcn_readfirstlane((int)v);
}
// A producer's contribution to the slot of the tensor it writes: every lane brings the largest |value| it STORED (fmaxf: NaNs are skipped, like
// in absmax_kernel -- the same set of numbers gives the same word whichever kernel took the maximum); the wave's maximum goes to one of the
// slot's sub-words, and only if it is larger than what is there already (a relaxed read first: most waves find nothing to add)
__device__ __forceinline__ void dt_amax_publish(unsigned *slot, float am)
{
#pragma unroll
    for (int o = 32; o; o >>= 1) am = fmaxf(am, __shfl_xor(am, o));
    if ((threadIdx.x & 63) == 0) {
        const unsigned b = __float_as_uint(am);
        unsigned *s = slot + ((blockIdx.x + (threadIdx.x >> 6)) & (DT_AMAX_SUB - 1)) * DT_AMAX_LINE;
        if (b > __hip_atomic_load(s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(s, b);
    }
}
#endif
// fp32 <-> fp16 bits on the host, round to nearest even, subnormals kept (what v_cvt_f16_f32 does under the default mode)
unsigned short h2_f16_rne(float x);
float h2_f16_f32(unsigned short h);
void wino_h2_split_host(float x_scaled, unsigned short t[2]);
// U [P][npad][K] fp32 (host) -> two fp16 terms [P][2][K/16][npad][16] of U[p] * uscale[p] (host)
void wino_h2_pack_weights(const float *u, int P, int npad, int K, const float *uscale, unsigned short *dst);
int launch_wino_gemm_s3(hipStream_t st, const GemmS3Args &a, int cus);
bool wino_gemm_s3_usable(int Mt, int K, int N);
bool wino_gemm_s3_half_chosen(const GemmS3Args &a, int cus);
double wino_gemm_s3_flops(const GemmS3Args &a);
void wino_s3_split_host(float x, unsigned short t[3]);
void wino_s3_pack_weights(const float *u, int P, int npad, int K, unsigned short *dst);
int launch_wino_input(hipStream_t st, const WinoArgs &a);
// U [P][npad][K] fp32 (device) -> split-bf16 [P][3][K/16][npad][16] (device)
int launch_wino_s3_pack(hipStream_t st, const float *u, int P, int npad, int K, unsigned short *dst);
int launch_wino_output(hipStream_t st, const WinoArgs &a, int gates);
bool wino_output_fills_amax(const WinoArgs &a, int gates);
// max |x| of `planes` tensors of rows x cols floats (row stride ld, plane stride plane_stride) -> slots[planes][DT_AMAX_SUB] (zeroed here)
int launch_absmax(hipStream_t st, const float *x, long long rows, int cols, long long ld, int planes, long long plane_stride, unsigned *slots, bool zero = true);
// U [P][npad][K] fp32 (device) -> two fp16 terms [P][2][K/16][npad][16] + epilogue factors pscale[P] (device); ts: Winograd tile (0: plain GEMM operand)
int launch_wino_h2_pack(hipStream_t st, const float *u, int P, int npad, int K, int ts, unsigned *slots, unsigned short *dst, float *pscale);
void wino_pack_weights(int ts, const float *hwio, int cin_src, int cout_src, const int *cin_map, int cin_dst, const int *n_map,
                       int npad, const float *scale, float *dst);

// ---------------------------------------------------------------------------
// other kernels
// ---------------------------------------------------------------------------
int launch_conv1_direct(hipStream_t st, const void *frames, int dtype, int B, int H, int W,
                        const float *w_packed /*[27][32]*/, const float *bias /*[32]*/,
                        const float *lut /*[256] or null*/, float slope, float *out /*[B,H/2,W/2,32]*/,
                        const unsigned *w3 = nullptr /*[2][3][64][4]*/, const unsigned *w3u8 = nullptr /*weights / 255: both set -> conv1_s3_kernel*/,
                        unsigned *amax_out = nullptr /*conv1_s3_kernel: max |x| of the outputs into this slot*/);
bool conv1_direct_fills_amax(const void *frames, int dtype, int W, const unsigned *w3, const unsigned *w3u8);
#define C1_W3_WORDS 1540      // 1536 table words + 16 bytes of zeros: where conv1_s3_kernel points the loads of out-of-image pixels
void conv1_split_tables(const float *w /*[27][32]*/, bool scale255, unsigned *w3 /*[1536 of C1_W3_WORDS]*/);

int launch_decode(hipStream_t st, const float *netout, long long frame_stride, int batch, int GH, int GW, int NB,
                  int NC, float obj_thr, float nms_thr, const float *anchors_dev, int cap, float *boxes,
                  int *counts, float *classes, float *post, const float *frame_thr /*[batch][2] (obj, nms) or null*/,
                  float *scratch /*batch x decode_scratch_floats() floats for grids above 1920 cells, else null*/);
size_t decode_scratch_floats(int GH, int GW, int NB);

int launch_bbox_iou(hipStream_t st, const float *pairs, int n, float *iou);

int launch_associate(hipStream_t st, const float *boxes, const int *counts, int n_clips, int T, int cap,
                     float thr, int *ids, int *nids);

int launch_heatmap_from_boxes(hipStream_t st, const float *box4, const double *xywh64, int n, int hs, float *out);
int launch_rect_from_heatmap(hipStream_t st, const float *heat, int n, int hs, float thresh, int *rect);
void ingest_tables(int src, int dst, int *tab);
int launch_ingest_resize(hipStream_t st, const unsigned char *src, int n, int Hs, int Ws, unsigned char *dst, int Hd,
                         int Wd, const int *xt, const int *yt);
#define DT_MAX_ANCHOR_BOXES 16
int launch_encode_targets(hipStream_t st, const int *objs, const int *counts, const int *dims, const double *aug,
                          int n, int cap, int GH, int GW, int NB, int C, int IH, int IW, int TBB,
                          const double *anchors_host, double *y, double *b);
int launch_top_box(hipStream_t st, const float *boxes, const int *counts, int n_frames, int cap, float *out4);

int launch_convlstm_gates_only(hipStream_t st, const float *xproj, long long xp_bs, int xp_ld, float *cstate,
                               long long c_bs, int c_ld, float *hout, long long h_bs, int h_ld, int B, int HW,
                               int U);

int launch_expand_rgb32(hipStream_t st, const void *frames, int dtype, long long n_pix, const float *lut, float *out);
int launch_unfold_bn(hipStream_t st, float *x, long long rows, int C, const float *scale, const float *shift);

int launch_global_maxpool(hipStream_t st, const float *in, int n, int HW, int C, float *out, int out_ld);
int launch_maxpool4_flatten(hipStream_t st, const float *in, int n, int H, int W, int C, float *out, int out_ld);
int launch_copy_cols(hipStream_t st, const float *src, int src_ld, float *dst, int dst_ld, long long rows,
                     int cols);
int launch_lstm_step(hipStream_t st, const float *xproj /*[B][4U] gate-interleaved*/, long long xp_bs,
                     const float *h_prev, long long h_bs, float *cstate, const float *Ur_packed /*[U][4U]*/,
                     float *h_out, long long ho_bs, int B, int U);
int launch_lstm_step0(hipStream_t st, const float *xproj, long long xp_bs, float *cstate, float *h_out,
                      long long ho_bs, int B, int U);
int launch_dense_sigmoid(hipStream_t st, const float *h, long long h_bs, const float *Wd /*[U][O]*/,
                         const float *bd, int B, int U, int O, float *out, long long out_bs);

// ---------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------
struct ProfEntry {
    int64_t launches = 0;
    double ms = 0, flops = 0, bytes = 0;
};

struct PendingEvent {
    hipEvent_t a, b;
    std::string name, tag;
};

struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
};

struct ConvLayer {
    int idx, ks, cin, cout, npad, pool;  // pool: reference MaxPooling2D after this layer
    float *wt = nullptr;                 // device, packed
    unsigned short *wt_s3 = nullptr;     // device, 1x1 layers: wt as split-bf16 terms [3][cin/16][npad][16] (wino_gemm_s3.hip) or null
    unsigned short *bias_s3 = nullptr;   // device, with wt_s3: the bias as the B rows of one extra K stage, [3][npad][16]
    unsigned short *wt_h2 = nullptr;     // device, 1x1 layers: wt as two fp16 terms of the scaled weights [2][cin/16][npad][16] (wino_gemm_s3.hip's fp16 form) or null
    float *pscale_h2 = nullptr;          // device, with wt_h2: [1] the epilogue factor 1 / (the weights' power of two)
    unsigned short *w3_h2 = nullptr;     // device, narrow 3x3 layers (conv_2 / 3 / 5's shapes): wt as two fp16 terms [2][9 cin / 16][npad][16] for conv3_h2.hip, or null
    float *pscale_w3 = nullptr;          // device, with w3_h2: [1]
    float *wino = nullptr;               // device, [P][npad][cin] Winograd-domain weights (wide 3x3 layers) or null
    int wino_ts = 0;                     // their output tile size (2, 4 or 6)
    float *wino_alt = nullptr;           // device, F(4x4) weights kept next to F(6x6) ones for small-batch launches, or null
    float *fused4s = nullptr;            // device, fused F(4x4,3x3) weights (wino4s_fused.hip: conv_2 / 3 / 5 / 6 / 8's shapes) or null
    float *bias = nullptr;               // device, [npad]
    float *scale = nullptr;              // device, [cout]: folded BatchNorm scale (dt_detector_extract un-folds with it) or null
    bool scale_has_zero = false;
};

// Tuning / test knobs.  Read from the environment ONCE, in dt_create; the two layer-level
// test entry points (dt_conv2d, dt_convlstm_step) re-read it so that the parity tests can force a policy on a
// live context.  No launch path calls getenv.
struct Policy {
    int wino = 1;            // DT_WINO: 0 never / 1 default policy / 2 everywhere the transforms are defined
    int wino_tile = 0;       // DT_WINO_TILE: 2/4/6 for every layer; 0 = default (6, recurrent convolution 4)
    int wino_minc = 64, wino_minn = 128, wino_mint = 0;   // DT_WINO_MINC / MINN / MINT (A/B runs)
    double wino_ws_gb = 96.0;   // DT_WINO_WS_GB: V + M' workspace above this -> direct form
    int mosaic = -1;         // DT_WINO_MOSAIC: 1 never, 2/3/4 force, -1 = fewest tiles
    int fused4 = 1;          // DT_WINO_FUSED4: the fused F(4x4) kernel (wino4s_fused.hip): 0 never / 1 conv_2 / 3 / 5 (Cin <= 64) from 1024
                             //                 blocks / 3 also conv_6 / 8 (Cin 128) / 2 any eligible layer at any size.  Read at weight load (0) and per launch
    int trk_merge = 1;       // DT_TRK_MERGE: the ConvLSTM2D input projection reads conv_feat only -- conv_23 (1x1, linear: x_bbox = W23 feat + b23) is folded into the
                             //               projection's weights at load (W' = Wx[feat] + W23 Wx[bbox] in float64, b23 through a border-aware bias): K 1120 -> 1024 and no
                             //               conv_23 launch when the caller does not ask for the detector's grid; 0 = the two-step form.  Winograd path only (large batches)
    int pin = 0;             // DT_PIN: 1 = kernel selection independent of the batch a call happens to carry (the frame-sharded tracker runs the same
                             //         frame in batches of other sizes for other world sizes): Winograd wherever defined, no frame mosaics, the fused kernel
                             //         and the split GEMM at any size, one tile form, no split-K, no F(4x4)/F(6x6) choice by tile count.  Slower at small
                             //         batch; results -- and track ids -- then do not depend on the number of ranks (parallel.py: deterministic=True)
    int c3h2 = 1;            // DT_C3H2: conv_2 / conv_3 / conv_5 as DIRECT 3x3 convolutions in the two-term fp16 form (conv3_h2.hip) from 1024 16x16-pixel blocks
                             //          (where the fused F(4x4) fp32 kernel ran until round 5); 0 = the fused kernel; 2 = at any size (parity tests).  Needs the fp16
                             //          form (DT_S3_H2, not DT_PIN).  Read at weight load (0: no fp16 copy of the weights) and per launch
    int c3h2_blocks2 = 1024; // DT_C3H2_BLOCKS2: 16 x 16-pixel blocks from which conv_2 (Cin = 32) takes the direct kernel under DT_C3H2=1 (conv_3 / conv_5: 1024)
    int wino_cfg = -1, wino_gn = -1;   // DT_WINO_CFG / DT_WINO_GN (A/B runs)
    int wino_grid_in = 0, wino_grid_out = 0, wino_thr_out = 0;   // DT_WINO_GRID_IN / DT_WINO_GRID_OUT / DT_WINO_THR_OUT (A/B runs: WinoArgs::grid_in / grid_out / thr_out)
    int ksplit = 0;          // DT_KSPLIT
    int conv_cfg = -1;       // DT_CONV_CFG
    int s3_conv1 = 1;        // DT_S3_CONV1: conv_1 on the bf16 pipe with split operands (conv1_s3_kernel); 0 = conv1_mfma_kernel (fp32 MFMA)
    int s3 = 1;              // DT_S3: the F(6x6) layers' batched GEMMs on the bf16 matrix pipe with 3-term split operands (wino_gemm_s3.hip):
                             //        0 never (fp32 MFMA) / 1 where it wins (K >= s3_mink, GEMM rows >= s3_minrows) / 2 wherever the shape allows
    int s3_mink = 128, s3_minrows = 2048;   // DT_S3_MINK / DT_S3_MINROWS (K >= 128 since round 4: with the line-sized epilogue stores the K = 128 GEMMs of
                                            // conv_6 / conv_8 take 3.4 instead of 4.3 ms on the fp32 kernel, their split input transform costs 0.7 back)
    int s3_minrows_h2 = 64;  // ... the row threshold where the launch would take the fp16 form (three products per multiply: the split GEMM beats the fp32 MFMA
                             //     kernel from far fewer rows than the bf16 form does -- detector forward at 32 / 64 / 128 / 192 frames: 3.50 -> 2.85, 6.60 -> 4.45,
                             //     9.91 -> 7.17, 14.5 -> 10.2 ms; 64 rather than 128 for the 75 / 98 rows of the 13 x 13 layers at 12 / 16 frames; profiles/r06_experiments.txt section 10).
                             //     DT_S3_MINROWS, when set, is the threshold of BOTH forms
    int s3_h2 = 1;           // DT_S3_H2: the split GEMMs in the fp16 form -- two terms of SCALED operands, three products on v_mfma_f32_32x32x16_f16 (half the
                             //           matrix-pipe work and 4 instead of 6 bytes per operand element; wino_gemm_s3.hip) -- wherever the bf16 form would run; 0 = three
                             //           bf16 terms / six products (round 3).  Read at weight load (the fp16 terms are built then) and per launch.  DT_PIN takes the
                             //           bf16 form: the fp16 form's scale is the batch's max |x|, so its rounding of small elements depends on the batch
    int c3fuse = 1;          // DT_C3FUSE: conv_4 (1x1, 128 -> 64) applied inside conv_3's direct kernel (conv3_h2.hip, FUSE): one launch, no 128-channel tensor; 0 = two launches
    int h2_minframes = 12;   // DT_H2_MINFRAMES: a forward of fewer frames takes round 5's forms (bf16 terms, fused fp32 kernel) and its producers publish no max |x|:
                             //                  at batch 8 the publications (a dependent load + atomic at the tail of 40-us kernels) and the one stand-alone absmax pass
                             //                  cost 0.16 ms of a 1.26 ms forward and the fp16 form has nothing to win there (weights-bound GEMMs on the fp32 kernel).
                             //                  Detector forward with the fp16 form from 8 frames against from 32: batch 8 1.41 vs 1.26 ms, 12: 1.98 vs 1.81, 16: 2.13 vs 2.10,
                             //                  24: 2.29 vs 2.81 -- the crossover lay between 16 and 24 frames while every small producer paid ~25 us for its publication.
                             //                  With the sub-words of a slot on distinct cache lines (DT_AMAX_LINE) and the cooperative output transform publishing too
                             //                  (no stand-alone absmax passes): batch 8 1.25-1.31 vs 1.26, 12: 1.50 vs 1.81, 16: 1.74 vs 2.09, 20: 2.40 vs 2.53, 24: 1.91 vs 2.80
                             //                  (row threshold 64-96) -- from 12 frames
    int s3_half = 0;          // DT_S3_HALF: 128-row tiles / two workgroups per CU in the split GEMM: 0 where it needs fewer rounds (default) / 1 always (N % 256 == 0) / -1 never
    int s3_rec_minrows = 512; // DT_S3_REC_MINROWS: the ConvLSTM recurrent step's F(4x4) GEMM (gate update in its output transform) takes the split
                              //                    kernel from this many GEMM rows (48 clips at 13x13: 588); 0 = never
    int s3_rec_minrows_h2 = 16; // ... where the step would take the fp16 form (one clip at 13x13 = 16 rows: 5.30 -> 4.71 ms per 30-frame clip; 8 / 12 / 24 / 36 clips =
                                //     98 / 147 / 294 / 441 rows: 6-9 % of the whole tracking step); DT_S3_REC_MINROWS, when set, is the threshold of both forms
    int s3_1x1_mink = 256;   // DT_S3_1X1_MINK: ... only for 1x1 layers with at least this many input channels (conv_7 / 10 / 12 / 15 / 17 / 23; 512 while the
                             //                  producer had to write split rows -- the kernel reads the fp32 activation itself since round 4; conv_4 at
                             //                  K = 128, N = 64 measured 3.55 ms there against 2.85 on the fp32 kernel)
    int s3_1x1_minrows = 16384;   // DT_S3_1X1_MINROWS: ... and at least this many pixels: one split GEMM of a 1x1 layer has M / 256 row tiles and no split-K, so at
                                  // batch 8 (conv_10 / 12: 5408 rows = 22 workgroups) the fp32 kernel with split-K is faster (0.035 vs 0.061 ms)
    int s3_1x1 = 1;          // DT_S3_1X1: the 1x1 layers with N % 128 == 0 run on wino_gemm_s3.hip straight from the fp32 activation (the kernel splits
                             //            its A fragments itself); 0 = fp32 MFMA
    int wino_coop = -1;      // DT_WINO_COOP: lane-cooperative F(6x6) transform kernels: -1 for small launches (default) / 0 never / 1 always
    int persist = 1;         // DT_PERSIST: 0 = one tile per workgroup for the GEMM-shaped launches (A/B runs)
    int xcd_remap = 1;       // DT_XCD_REMAP: 0 = plain tile numbering (L2 traffic experiments)
    int tile_gn = -1;        // DT_TILE_GN: column tiles per group of the tile order; -1 = per-layer default
};
void policy_from_env(Policy &p, int pin_override = -1 /* >= 0: this value instead of DT_PIN */);

struct dt_ctx {
    std::string err;
    Policy pol;
    hipStream_t stream = nullptr;
    int device_ok = 0;
    // detector
    int image_h = 0, image_w = 0, nb_box = 0, nb_class = 0, cb = 0;
    float anchors[64];
    float *anchors_dev = nullptr;
    float dec_anchors_host[64];   // last anchors handed to dt_decode (persistent staging of the caller's host array)
    int dec_anchors_n = 0;
    bool det_loaded = false;
    ConvLayer layers[24];   // 1..23
    unsigned short *s3_ones = nullptr;   // device, [256][16] FLOATS: the A rows (1, 0, .., 0) that carry a 1x1 layer's bias through wino_gemm_s3.hip
    std::map<const void *, unsigned short *> wino_s3;   // F(6x6) Winograd weights (device pointer) -> their split-bf16 form (wino_gemm_s3.hip), when built
    struct H2Weights { unsigned short *terms = nullptr; float *pscale = nullptr; };
    std::map<const void *, H2Weights> wino_h2;          // ... -> their fp16 form: two terms of U[p] * 2^su[p], and the epilogue factors [P]
    // max-|x| slots of the fp16 form (DT_AMAX_SUB words each; dt_internal.h: dt_h2_base).  Slot 0 holds 1.0 (|h_t| < 1: the ConvLSTM recurrent step);
    // slot i in 1..23: the OUTPUT of conv_i, filled by the epilogue of the kernel that writes it (network.hip: amax_begin zeroes them per forward);
    // 32 + i: the INPUT of conv_i where no producer measured it (absmax_kernel); 56: the tracker's z / conv_feat; 57, 58: test entry points;
    // 64..127: scratch of the weight packs
    unsigned *amax = nullptr;
    struct AmaxTag { const float *lo, *hi; int cols, slot; };   // the tensor that occupies [lo, hi), `cols` channels per pixel -> the slot that holds its max |x|
    bool h2_small = false;                   // the running call carries fewer than Policy::h2_minframes frames: no fp16 form, no max-|x| publication
    std::vector<AmaxTag> amax_tag;           // valid inside one API call only (amax_reset), and until a layer writes into [lo, hi) (amax_forget)
    float *conv1_w = nullptr, *conv1_b = nullptr, *lut255 = nullptr;
    unsigned *conv1_w3 = nullptr, *conv1_w3u8 = nullptr;   // device: split-bf16 weight tables of conv1_s3_kernel: w and w / 255 (conv1.hip:conv1_split_tables)
    std::vector<float> conv1_hwio32, conv1_scale, conv1_shift;   // host copy of conv_1 as a Cin = 32 layer (dt_detector_extract)
    // tracker head
    bool trk_loaded = false;
    int trk_units = 0, trk_cx = 0 /*padded z channels*/;
    float *trk_wx = nullptr, *trk_bx = nullptr;   // input conv, N gate-interleaved
    float *trk_wh = nullptr;                      // recurrent conv
    float *trk_wx_wino = nullptr, *trk_wh_wino = nullptr;   // their Winograd-domain forms
    float *trk_wxm_wino = nullptr;                // F(6x6) weights of the MERGED input projection (conv_23 folded in; 1024 input channels) or null
    float *trk_bx16 = nullptr;                    // its bias: [17][4U] gate-interleaved -- row 0 the interior bias, rows 1 + k the correction of border case k
    std::vector<float> conv23_hwio, conv23_bias;  // host copies of conv_23 ([1024][cb], [cb]) and of the ConvLSTM input kernel / bias (Keras HWIO, x_bbox first):
    std::vector<float> trk_hkernel, trk_hbias;    // what the merge is computed from, whichever of the two loaders runs second
    int trk_wino_ts = 0, trk_wh_ts = 0;   // tile of the input / recurrent convolution's Winograd weights
    float *trk_wo = nullptr, *trk_bo = nullptr;   // tconv_2 1x1
    int trk_wo_npad = 0;
    // tiny tracker
    bool tiny_loaded = false;
    int tiny_D = 0, tiny_Dpad = 0, tiny_U = 0, tiny_O = 0, tiny_Opad = 0;
    float *tiny_wx = nullptr, *tiny_bx = nullptr, *tiny_ur = nullptr, *tiny_wd = nullptr, *tiny_bd = nullptr;
    // workspaces (grown on demand)
    std::map<std::string, DevBuf> ws;
    int last_batch = 0;
    bool tap_feat = false, tap_netout = false;   // last forward wrote the library-owned 'feat' / 'netout' workspaces
    int ing_key[4] = {0, 0, 0, 0};   // (Hs, Ws, Hd, Wd) of the cached ingest tables
    // hipGraph replay of the launch-bound inner sequences (dt_graph_enable; network.hip:graphed)
    bool graph_on = false, capturing = false;
    hipStream_t gstream = nullptr;
    hipEvent_t gev_in = nullptr, gev_out = nullptr;
    std::map<std::string, hipGraphExec_t> graphs;
    std::map<std::string, int> graph_seen;
    std::map<std::string, std::vector<AmaxTag>> graph_tags;   // amax_tag as a graphed sequence left it (re-applied on replay)
    int64_t graph_replays = 0, graph_captures = 0;
    // profiling
    bool prof = false;
    std::map<std::string, ProfEntry> prof_tab;
    std::vector<PendingEvent> pending;
};

int dt_fail(dt_ctx *ctx, int code, const char *fmt, ...);
float *ws_get(dt_ctx *ctx, const char *name, size_t bytes, bool zero_on_grow = false);

#define HIP_TRY(ctx, expr)                                                                       \
    do {                                                                                         \
        hipError_t _e = (expr);                                                                  \
        if (_e != hipSuccess)                                                                    \
            return dt_fail(ctx, DT_ERR_DEVICE, "%s: %s (%s:%d)", #expr, hipGetErrorString(_e),   \
                           __FILE__, __LINE__);                                                  \
    } while (0)

// profiling scope helper
struct ProfScope {
    dt_ctx *ctx;
    PendingEvent ev;
    bool on;
    ProfScope(dt_ctx *c, const char *name, double flops, double bytes, const char *tag = nullptr);
    ~ProfScope();
};

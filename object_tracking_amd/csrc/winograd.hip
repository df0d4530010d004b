// winograd.hip -- Winograd form of the reference's 3x3 'same' stride-1 convolutions
// (models_detection/KerasYOLO.py:279-396 conv blocks, models_tracking/MultiObjDetTracker.py:176
// ConvLSTM2D input and recurrent convolutions) for the wide layers, where it pays:
//
//     Y = At [ (G g Gt) .* (Bt d B) ] A      per TSxTS output tile, (TS+2)x(TS+2) input patch d, 3x3 filter g
//
// F(6x6,3x3) (TS=6): 64 multiplies per 36 outputs instead of 324 (5.06x on whole tiles).
// F(4x4,3x3) (TS=4): 36 multiplies per 16 outputs instead of 144 -- the MFMA work of a layer
// drops 4x (2.64x at 13x13, whose 4x4 tiles cover 16x16; 3.45x at 26x26).  F(2x2,3x3) (TS=2): 16 per 4
// outputs, 2.25x (1.94x at 13x13); kept selectable (DT_WINO_TILE=2) -- its fp32 rounding error equals the
// direct form's, F(4x4,3x3)'s is ~15x that (1.5e-5 absolute at activation scale 4; SURVEY.md's bar is 1e-3).
// The contraction over input channels becomes P = (TS+2)^2 independent GEMMs
//     M'[p][tile][cout] = sum_c V[p][tile][c] * U[p][cout][c],      p = (TS+2)*xi + nu
// which run as ONE persistent launch of the fp32 MFMA kernel of conv_igemm.hip (1x1 path, P problems), so
// the matrix-core code is shared with the direct form.  This file holds what is around it:
//   wino_input_kernel    activation NHWC -> V [P][tiles][Cin]          (Bt d B; HBM-bound)
//   wino_output_kernel   M' [P][tiles][Cout] -> NHWC output             (At m A + bias + LeakyReLU
//                        [+ MaxPooling2D(2,2): tiles start on even coordinates, so pooling windows never
//                        straddle tiles])
//   wino_output_gates_kernel   the same transform followed by the ConvLSTM2D gate update of
//                        MultiObjDetTracker.py:176 in registers
//   wino_pack_weights    host: U = G g Gt per (cin, cout), laid out per position for the MFMA kernel
// fp32 throughout.
#include <thread>

#include "dt_internal.h"

#define WINO_THREADS 256
#ifndef DT_WINO_NT
#define DT_WINO_NT 7   // streaming (nontemporal) accesses, A/B per bit: 1 M' loads, 2 activation stores of the output transform (13.95 -> 13.25 ms
                       // per step), 4 V stores of the input transform (-0.2 ms); 8 = activation loads of the input transform: slower (halo re-reads)
#endif

template <int V> struct VecOf;
template <> struct VecOf<1> { typedef float T; };
template <> struct VecOf<2> { typedef float T __attribute__((ext_vector_type(2))); };
template <> struct VecOf<4> { typedef float T __attribute__((ext_vector_type(4))); };

template <int V> __device__ __forceinline__ typename VecOf<V>::T vload(const float *p)
{
    return *reinterpret_cast<const typename VecOf<V>::T *>(p);
}
template <int V> __device__ __forceinline__ void vstore(float *p, typename VecOf<V>::T v)
{
    *reinterpret_cast<typename VecOf<V>::T *>(p) = v;
}
// streaming variants (DT_WINO_NT build switch, A/B): data that is read exactly once / not re-read by this kernel
template <int V> __device__ __forceinline__ typename VecOf<V>::T vload_nt(const float *p)
{
#if DT_WINO_NT & 1
    return __builtin_nontemporal_load(reinterpret_cast<const typename VecOf<V>::T *>(p));
#else
    return vload<V>(p);
#endif
}
template <int V> __device__ __forceinline__ void vstore_nt(float *p, typename VecOf<V>::T v)
{
#if DT_WINO_NT & 2
    __builtin_nontemporal_store(v, reinterpret_cast<typename VecOf<V>::T *>(p));
#else
    vstore<V>(p, v);
#endif
}
template <int V> __device__ __forceinline__ void vstore_v(float *p, typename VecOf<V>::T v)   // V planes (A/B bit 4)
{
#if DT_WINO_NT & 4
    __builtin_nontemporal_store(v, reinterpret_cast<typename VecOf<V>::T *>(p));
#else
    vstore<V>(p, v);
#endif
}
template <int V> __device__ __forceinline__ typename VecOf<V>::T vload_in(const float *p)     // activations (A/B bit 8)
{
#if DT_WINO_NT & 8
    return __builtin_nontemporal_load(reinterpret_cast<const typename VecOf<V>::T *>(p));
#else
    return vload<V>(p);
#endif
}
template <int V> __device__ __forceinline__ float lane_of(const typename VecOf<V>::T &v, int e) { return v[e]; }
template <> __device__ __forceinline__ float lane_of<1>(const float &v, int) { return v; }
template <int V> __device__ __forceinline__ void set_lane(typename VecOf<V>::T &v, int e, float x) { v[e] = x; }
template <> __device__ __forceinline__ void set_lane<1>(float &v, int, float x) { v = x; }
template <int V> __device__ __forceinline__ typename VecOf<V>::T vzero()
{
    typename VecOf<V>::T z;
#pragma unroll
    for (int e = 0; e < V; ++e) set_lane<V>(z, e, 0.0f);
    return z;
}

// 1-D transforms (Lavin & Gray, "Fast Algorithms for Convolutional Neural Networks", the standard
// interpolation points 0, +-1 for F(2,3), 0, +-1, +-2 for F(4,3) and 0, +-1, +-2, +-1/2 for F(6,3)), applied to
// rows then columns.
//   bt: (TS+2) inputs -> (TS+2) outputs in place;  at: (TS+2) inputs -> TS outputs (first TS slots)
template <int TS, typename T> __device__ __forceinline__ void bt_1d(T *d)
{
    if (TS == 2) {
        const T t0 = d[0] - d[2], t1 = d[1] + d[2], t2 = d[2] - d[1], t3 = d[1] - d[3];
        d[0] = t0; d[1] = t1; d[2] = t2; d[3] = t3;
    } else if (TS == 6) {
        // F(6,3), points 0, +-1, +-2, +-1/2:  rows of Bt paired as (even part) +- (odd part)
        const T t0 = d[0] - d[6] + 5.25f * (d[4] - d[2]);
        const T t7 = d[7] - d[1] + 5.25f * (d[3] - d[5]);
        const T a = d[2] - 4.25f * d[4] + d[6], b = d[1] - 4.25f * d[3] + d[5];
        const T c = 0.25f * d[2] - 1.25f * d[4] + d[6], e = 0.5f * d[1] - 2.5f * d[3] + 2.0f * d[5];
        const T f = 4.0f * d[2] - 5.0f * d[4] + d[6], g = 2.0f * d[1] - 2.5f * d[3] + 0.5f * d[5];
        d[0] = t0; d[1] = a + b; d[2] = a - b; d[3] = c + e; d[4] = c - e; d[5] = f + g; d[6] = f - g; d[7] = t7;
    } else {
        const T t0 = 4.0f * d[0] - 5.0f * d[2] + d[4];
        const T t1 = -4.0f * (d[1] + d[2]) + d[3] + d[4];
        const T t2 = 4.0f * (d[1] - d[2]) - d[3] + d[4];
        const T t3 = 2.0f * (d[3] - d[1]) - d[2] + d[4];
        const T t4 = 2.0f * (d[1] - d[3]) - d[2] + d[4];
        const T t5 = 4.0f * d[1] - 5.0f * d[3] + d[5];
        d[0] = t0; d[1] = t1; d[2] = t2; d[3] = t3; d[4] = t4; d[5] = t5;
    }
}
template <int TS, typename T> __device__ __forceinline__ void at_1d(T *m)
{
    if (TS == 2) {
        const T y0 = m[0] + m[1] + m[2], y1 = m[1] - m[2] - m[3];
        m[0] = y0; m[1] = y1;
    } else if (TS == 6) {
        const T p1 = m[1] + m[2], q1 = m[1] - m[2], p2 = m[3] + m[4], q2 = m[3] - m[4], p3 = m[5] + m[6], q3 = m[5] - m[6];
        const T y0 = m[0] + p1 + p2 + p3;
        const T y1 = q1 + 2.0f * q2 + 0.5f * q3;
        const T y2 = p1 + 4.0f * p2 + 0.25f * p3;
        const T y3 = q1 + 8.0f * q2 + 0.125f * q3;
        const T y4 = p1 + 16.0f * p2 + 0.0625f * p3;
        const T y5 = q1 + 32.0f * q2 + 0.03125f * q3 + m[7];
        m[0] = y0; m[1] = y1; m[2] = y2; m[3] = y3; m[4] = y4; m[5] = y5;
    } else {
        const T a = m[1] + m[2], b = m[1] - m[2], c = m[3] + m[4], e = m[3] - m[4];
        const T y0 = m[0] + a + c;
        const T y1 = b + 2.0f * e;
        const T y2 = a + 4.0f * c;
        const T y3 = b + 8.0f * e + m[5];
        m[0] = y0; m[1] = y1; m[2] = y2; m[3] = y3;
    }
}

// Tiles live on a VIRTUAL image: one frame (g = 1), or a g x g mosaic of frames laid out with a pitch of
// H+1 / W+1, i.e. with one row / column of zeros between neighbours.  The zero separator is exactly the
// 'same' padding both neighbours need, so tiles may straddle frames: a 13x13 grid costs (2*14/4)^2 / 4 =
// 12.25 F(4x4,3x3) tiles per frame instead of the 16 that cover 16x16 (-23 % GEMM rows and transform traffic).
struct TileId {
    int grp, ty, tx;   // frame group (g*g frames), tile row / column on its virtual image
};
__device__ __forceinline__ TileId tile_id(const WinoArgs &p, int tile)
{
    const int tpf = p.th * p.tw;
    TileId t;
    t.grp = tile / tpf;
    const int r = tile - t.grp * tpf;
    t.ty = r / p.tw;
    t.tx = r - t.ty * p.tw;
    return t;
}
// virtual (row, col) of group grp -> frame b and pixel (h, w); false = padding / separator / no such frame
__device__ __forceinline__ bool vpixel(const WinoArgs &p, int grp, int vh, int vw, int &b, int &h, int &w)
{
    if (vh < 0 || vw < 0) return false;
    if (p.g == 1) {
        b = grp; h = vh; w = vw;
        return vh < p.H && vw < p.W;
    }
    const int fy = vh / (p.H + 1), fx = vw / (p.W + 1);
    h = vh - fy * (p.H + 1);
    w = vw - fx * (p.W + 1);
    b = (grp * p.g + fy) * p.g + fx;
    return fy < p.g && fx < p.g && h < p.H && w < p.W && b < p.B;
}

// ---- input transform ----------------------------------------------------------------------------
// fp32 -> three bf16 terms (round to nearest even at every step; host twin: wino_gemm_s3.hip:wino_s3_split_host)
typedef __bf16 wino_bf4 __attribute__((ext_vector_type(4)));
typedef unsigned int wino_u2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void s3_split4(const VecOf<4>::T x, wino_u2 t[3])
{
    const wino_bf4 h = __builtin_convertvector(x, wino_bf4);
    const VecOf<4>::T r1 = x - __builtin_convertvector(h, VecOf<4>::T);
    const wino_bf4 m = __builtin_convertvector(r1, wino_bf4);
    const VecOf<4>::T r2 = r1 - __builtin_convertvector(m, VecOf<4>::T);
    const wino_bf4 l = __builtin_convertvector(r2, wino_bf4);
    t[0] = __builtin_bit_cast(wino_u2, h); t[1] = __builtin_bit_cast(wino_u2, m); t[2] = __builtin_bit_cast(wino_u2, l);
}

#ifndef WINO_S3_NT
#define WINO_S3_NT 1      // bit 0: the V terms of the input transforms as nontemporal stores (measured, profiles/r04_experiments.txt: 11.6 -> 11.1 ms per step)
#endif
template <int BIT, typename T> __device__ __forceinline__ void s3_store(unsigned short *p, const T &v)
{
    if (WINO_S3_NT & BIT) __builtin_nontemporal_store(v, reinterpret_cast<T *>(p));
    else *reinterpret_cast<T *>(p) = v;
}
typedef __bf16 wino_bf2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void s3_split2(const VecOf<2>::T x, unsigned t[3])
{
    const wino_bf2 h = __builtin_convertvector(x, wino_bf2);
    const VecOf<2>::T r1 = x - __builtin_convertvector(h, VecOf<2>::T);
    const wino_bf2 m = __builtin_convertvector(r1, wino_bf2);
    const VecOf<2>::T r2 = r1 - __builtin_convertvector(m, VecOf<2>::T);
    const wino_bf2 l = __builtin_convertvector(r2, wino_bf2);
    t[0] = __builtin_bit_cast(unsigned, h); t[1] = __builtin_bit_cast(unsigned, m); t[2] = __builtin_bit_cast(unsigned, l);
}

// ---- the fp16 form (wino_gemm_s3.hip, NT = 2): x scaled by a power of two, then  hi = f16(x), lo = f16(x - hi)  (nearest even) ----
typedef _Float16 wino_h4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void h2_split4(const VecOf<4>::T x, wino_u2 t[2])
{
    const wino_h4 h = __builtin_convertvector(x, wino_h4);
    const VecOf<4>::T r = x - __builtin_convertvector(h, VecOf<4>::T);
    t[0] = __builtin_bit_cast(wino_u2, h);
    t[1] = __builtin_bit_cast(wino_u2, __builtin_convertvector(r, wino_h4));
}

// One work item = (tile, V channels).  Tile (grp, ty, tx) covers virtual rows TS*ty-1 .. TS*ty+TS, cols
// TS*tx-1 .. TS*tx+TS ('same' padding, separators and the rows/columns past the image read as zero).
// S3: V leaves as split-bf16 terms [P][3][C/16][Mp][16] (wino_gemm_s3.hip).  Items then run (16-channel block, tile, channel group)
// with the channel group fastest and the tile next: the 16 / V lanes of a block's channel groups and the consecutive tiles behind
// them are CONTIGUOUS in every (position, term) plane -- a store instruction of a wavefront writes one 256-byte (V = 2) run of ONE
// plane, where the lane-cooperative producers write eight 128-byte lines of eight planes.  Built and measured in round 4 as a
// candidate for the big F(6x6) launches (V = 2): 12.2 ms per step against 11.5 for the cooperative producer -- kept as the
// DT_WINO_COOP=0 form (A/B, parity tests); V = 4: the F(4x4) recurrent step with DT_WINO_COOP=0.
template <int TS, int V, bool S3 = false> __global__ __launch_bounds__(WINO_THREADS) void wino_input_kernel(WinoArgs p)
{
    typedef typename VecOf<V>::T T;
    constexpr int NI = TS + 2;
    const int cq_n = p.C / V;
    const long long items = (long long)p.Mt * cq_n;
    const long long plane = (long long)p.Mt * p.C;
    for (long long it = (long long)blockIdx.x * blockDim.x + threadIdx.x; it < items;
         it += (long long)gridDim.x * blockDim.x) {
        int tile, c;
        if constexpr (S3) {
            constexpr int GPB = 16 / V;                 // channel groups per 16-channel block
            const long long tb = it / GPB;
            c = (int)(tb / p.Mt) * 16 + (int)(it - tb * GPB) * V;
            tile = (int)(tb % p.Mt);
        } else {
            tile = (int)(it / cq_n);
            c = (int)(it - (long long)tile * cq_n) * V;
        }
        const TileId t = tile_id(p, tile);
        T d[NI][NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                int b, h, w;
                const bool ok = vpixel(p, t.grp, TS * t.ty - 1 + i, TS * t.tx - 1 + j, b, h, w);
                d[i][j] = ok ? vload_in<V>(p.in + (long long)b * p.in_bs + (long long)(h * p.W + w) * p.in_ld + c) : vzero<V>();
            }
        }
        // Bt d : down the columns
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            T col[NI];
#pragma unroll
            for (int i = 0; i < NI; ++i) col[i] = d[i][j];
            bt_1d<TS>(col);
#pragma unroll
            for (int i = 0; i < NI; ++i) d[i][j] = col[i];
        }
        // (Bt d) B : along the rows, stored plane by plane
        if constexpr (S3 && V == 4) {
            const long long term = (long long)(p.C >> 4) * p.Mp * 16;
            unsigned short *dst = p.v_s3 + ((long long)(c >> 4) * p.Mp + tile) * 16 + (c & 15);
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                bt_1d<TS>(d[i]);
                wino_u2 tr[NI][3];       // a row's splits before its stores
#pragma unroll
                for (int j = 0; j < NI; ++j) s3_split4(d[i][j], tr[j]);
#pragma unroll
                for (int j = 0; j < NI; ++j)
#pragma unroll
                    for (int k = 0; k < 3; ++k) s3_store<1>(dst + ((long long)(NI * i + j) * 3 + k) * term, tr[j][k]);
            }
        } else if constexpr (S3 && V == 2) {
            const long long term = (long long)(p.C >> 4) * p.Mp * 16;
            unsigned short *dst = p.v_s3 + ((long long)(c >> 4) * p.Mp + tile) * 16 + (c & 15);
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                bt_1d<TS>(d[i]);
                unsigned tr[NI][3];      // a row's splits before its stores (a VALU write to a register an in-flight store reads waits for it)
#pragma unroll
                for (int j = 0; j < NI; ++j) s3_split2(d[i][j], tr[j]);
#pragma unroll
                for (int j = 0; j < NI; ++j)
#pragma unroll
                    for (int k = 0; k < 3; ++k)
                        s3_store<1>(dst + ((long long)(NI * i + j) * 3 + k) * term, tr[j][k]);
            }
        } else {
            float *dst = p.v + (long long)tile * p.C + c;
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                bt_1d<TS>(d[i]);
#pragma unroll
                for (int j = 0; j < NI; ++j) vstore_v<V>(dst + (long long)(NI * i + j) * plane, d[i][j]);
            }
        }
    }
}

__device__ __forceinline__ float wino_leaky(float v, float slope) { return v > 0.0f ? v : v * slope; }
__device__ __forceinline__ float wino_hard_sigmoid(float x)
{
    const float y = __fmaf_rn(0.2f, x, 0.5f);   // Keras 2.x hard_sigmoid
    return fminf(fmaxf(y, 0.0f), 1.0f);
}

// At m A: loads the (TS+2)^2 planes of one (tile, V columns) item and leaves y[TS][TS] in m[i][j], i,j < TS
template <int TS, int V>
__device__ __forceinline__ void wino_at_m_a(const float *src, long long plane, typename VecOf<V>::T (*m)[TS + 2])
{
    typedef typename VecOf<V>::T T;
    constexpr int NI = TS + 2;
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) m[i][j] = vload_nt<V>(src + (long long)(NI * i + j) * plane);
#pragma unroll
    for (int j = 0; j < NI; ++j) {   // At m : down the columns
        T col[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) col[i] = m[i][j];
        at_1d<TS>(col);
#pragma unroll
        for (int i = 0; i < TS; ++i) m[i][j] = col[i];
    }
#pragma unroll
    for (int i = 0; i < TS; ++i) at_1d<TS>(m[i]);   // (At m) A : along the rows
}

// which of the 16 border cases pixel (h, w) of an H x W frame is (WinoArgs::bias16)
__device__ __forceinline__ int wino_border_case(const WinoArgs &p, int h, int w)
{
    return (h == 0 ? 1 : 0) | (h == p.H - 1 ? 2 : 0) | (w == 0 ? 4 : 0) | (w == p.W - 1 ? 8 : 0);
}

// ---- output transform, conv block epilogue -------------------------------------------------------
// One work item = (tile, V output channels): bias + LeakyReLU, optional full-resolution output and
// optional 2x2-pooled output.
template <int TS, int V> __global__ __launch_bounds__(WINO_THREADS) void wino_output_kernel(WinoArgs p)
{
    typedef typename VecOf<V>::T T;
    constexpr int NI = TS + 2;
    const int nq = p.N / V;
    const long long items = (long long)p.Mt * nq;
    const long long plane = (long long)p.Mt * p.m_ld;
    float am = 0.0f;      // largest |value| this thread stored (WinoArgs::amax_out)
    const bool am_full = p.amax_out && !p.out2, am_pool = p.amax_out && p.out2;
    for (long long it = (long long)blockIdx.x * blockDim.x + threadIdx.x; it < items;
         it += (long long)gridDim.x * blockDim.x) {
        const int tile = (int)(it / nq);
        const int c = (int)(it - (long long)tile * nq) * V;
        const TileId t = tile_id(p, tile);
        T m[NI][NI];
        wino_at_m_a<TS, V>(p.m + (long long)tile * p.m_ld + c, plane, m);
        const T bv = p.bias ? vload<V>(p.bias + c) : vzero<V>();
#pragma unroll
        for (int i = 0; i < TS; ++i)
#pragma unroll
            for (int j = 0; j < TS; ++j) {
                T v = m[i][j] + bv;
#pragma unroll
                for (int e = 0; e < V; ++e) set_lane<V>(v, e, wino_leaky(lane_of<V>(v, e), p.slope));
                m[i][j] = v;
            }
        if (p.out) {
#pragma unroll
            for (int i = 0; i < TS; ++i)
#pragma unroll
                for (int j = 0; j < TS; ++j) {
                    int b, h, w;
                    if (vpixel(p, t.grp, TS * t.ty + i, TS * t.tx + j, b, h, w)) {
                        T v = m[i][j];
                        if (p.bias16) {    // (slope 1: launcher) border pixels add their case's correction to the interior bias already in v
                            const int k = wino_border_case(p, h, w);
                            if (k) v = v + vload<V>(p.bias16 + (long long)k * p.N + c);
                        }
                        if (am_full) {
#pragma unroll
                            for (int e = 0; e < V; ++e) am = fmaxf(am, fabsf(lane_of<V>(v, e)));
                        }
                        vstore_nt<V>(p.out + (long long)b * p.out_bs + (long long)(h * p.W + w) * p.out_ld + c, v);
                    }
                }
        }
        if (p.out2) {   // MaxPooling2D(2,2): H and W are even whenever the reference pools; g == 1 (launcher)
            const int H2 = p.H >> 1, W2 = p.W >> 1;
#pragma unroll
            for (int i = 0; i < TS / 2; ++i)
#pragma unroll
                for (int j = 0; j < TS / 2; ++j) {
                    const int h2 = (TS / 2) * t.ty + i, w2 = (TS / 2) * t.tx + j;
                    if (h2 < H2 && w2 < W2) {
                        T mx;
#pragma unroll
                        for (int e = 0; e < V; ++e)
                            set_lane<V>(mx, e, fmaxf(fmaxf(lane_of<V>(m[2 * i][2 * j], e), lane_of<V>(m[2 * i][2 * j + 1], e)),
                                                     fmaxf(lane_of<V>(m[2 * i + 1][2 * j], e), lane_of<V>(m[2 * i + 1][2 * j + 1], e))));
                        if (am_pool) {
#pragma unroll
                            for (int e = 0; e < V; ++e) am = fmaxf(am, fabsf(lane_of<V>(mx, e)));
                        }
                        vstore<V>(p.out2 + ((long long)(t.grp * H2 + h2) * W2 + w2) * p.out2_ld + c, mx);
                    }
                }
        }
    }
    if (p.amax_out) dt_amax_publish(p.amax_out, am);
}

// ---- output transform, ConvLSTM2D gate update ------------------------------------------------------
// N axis packed [j/32][gate][j%32] like EPI_GATES of conv_igemm.hip.  One work item = (tile, V hidden
// channels): the i,f,c,o pre-activations of the recurrent convolution come out of the transform in
// registers, the input projection (bias included) is added, c is updated in place and h written.
#ifndef WINO_GATES_PIN
#define WINO_GATES_PIN 1
#endif
template <int TS, int V> __global__ __launch_bounds__(WINO_THREADS) void wino_output_gates_kernel(WinoArgs p)
{
    typedef typename VecOf<V>::T T;
    constexpr int NI = TS + 2;
    const int U = p.N >> 2;
    const int uq = U / V;
    const long long items = (long long)p.Mt * uq;
    const long long plane = (long long)p.Mt * p.m_ld;
    for (long long it = (long long)blockIdx.x * blockDim.x + threadIdx.x; it < items;
         it += (long long)gridDim.x * blockDim.x) {
        const int tile = (int)(it / uq);
        const int jc = (int)(it - (long long)tile * uq) * V;          // hidden channel
        const int col = (jc >> 5) * 128 + (jc & 31);                  // column of gate i; f,c,o at +32,+64,+96
        const TileId t = tile_id(p, tile);
        T y[4][TS][TS];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            T m[NI][NI];
            wino_at_m_a<TS, V>(p.m + (long long)tile * p.m_ld + col + g * 32, plane, m);
#pragma unroll
            for (int i = 0; i < TS; ++i)
#pragma unroll
                for (int j = 0; j < TS; ++j) y[g][i][j] = m[i][j];
        }
        // The x-projection and cell-state reads of a whole group of pixel rows go out together, with the address SELECTED for the
        // pixels outside the frame (they re-read pixel 0; nothing is stored for them) -- not pixel by pixel behind the previous
        // pixel's stores, each a round trip of its own (round 3's form: 17 dependent round trips per item).
        constexpr int RG = TS >= 4 ? TS / 2 : TS;          // pixel rows per group (register budget: 5 V (RG TS) values in flight)
#if WINO_GATES_PIN      // y is materialised HERE (hipcc otherwise sinks the whole transform below the first group's loads: 4 NI NI + 5 RG TS values live)
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int i = 0; i < TS; ++i)
#pragma unroll
                for (int j = 0; j < TS; ++j) asm volatile("" ::"v"(y[g][i][j]));
#endif
#pragma unroll
        for (int i0 = 0; i0 < TS; i0 += RG) {
            T xz[RG][TS][4], cpv[RG][TS];
            long long cofs[RG][TS], oofs[RG][TS];
            bool ok[RG][TS];
#pragma unroll
            for (int i = 0; i < RG; ++i)
#pragma unroll
                for (int j = 0; j < TS; ++j) {
                    int b = 0, h = 0, w = 0;
                    ok[i][j] = vpixel(p, t.grp, TS * t.ty + i0 + i, TS * t.tx + j, b, h, w);
                    if (!ok[i][j]) { b = 0; h = 0; w = 0; }
                    const long long pix = h * p.W + w;
                    const float *xp = p.xproj + (long long)b * p.xp_bs + pix * p.xp_ld + col;
                    cofs[i][j] = (long long)b * p.c_bs + pix * p.c_ld + jc;
                    oofs[i][j] = (long long)b * p.out_bs + pix * p.out_ld + jc;
#pragma unroll
                    for (int g = 0; g < 4; ++g) xz[i][j][g] = vload<V>(xp + 32 * g);
                    cpv[i][j] = vload<V>(p.cstate + cofs[i][j]);
                }
#pragma unroll
            for (int i = 0; i < RG; ++i)
#pragma unroll
                for (int j = 0; j < TS; ++j) {
                    const T zi = y[0][i0 + i][j] + xz[i][j][0], zf = y[1][i0 + i][j] + xz[i][j][1];
                    const T zc = y[2][i0 + i][j] + xz[i][j][2], zo = y[3][i0 + i][j] + xz[i][j][3];
                    T cn, hn;
#pragma unroll
                    for (int e = 0; e < V; ++e) {
                        const float gi = wino_hard_sigmoid(lane_of<V>(zi, e)), gf = wino_hard_sigmoid(lane_of<V>(zf, e));
                        const float go = wino_hard_sigmoid(lane_of<V>(zo, e));
                        const float cv = gf * lane_of<V>(cpv[i][j], e) + gi * tanhf(lane_of<V>(zc, e));
                        set_lane<V>(cn, e, cv);
                        set_lane<V>(hn, e, go * tanhf(cv));
                    }
                    if (ok[i][j]) {
                        vstore<V>(p.cstate + cofs[i][j], cn);
                        vstore<V>(p.out + oofs[i][j], hn);
                    }
                }
        }
    }
}

// Workgroup size: 256 threads, or one wavefront per workgroup when the whole launch has fewer than 512 x 256 work items
// (a few frames per call): the transforms are bound by what ONE CU can stream, so the same threads spread over four
// times as many CUs finish sooner (batch-8 forward: 26 transform launches).
static unsigned wino_threads(long long items) { return items < 512ll * WINO_THREADS ? 64u : (unsigned)WINO_THREADS; }
static unsigned wino_blocks(long long items)
{
    const long long th = wino_threads(items);
    long long nb = (items + th - 1) / th;
    const long long cap = 256 * 32;   // 32 workgroups of 256 threads per CU's worth of grid; grid-stride beyond
    if (nb > cap) nb = cap;
    return (unsigned)(nb < 1 ? 1 : nb);
}

// vector width per work item: 4 channels for the input transforms, 4 / 2 / 1 for the F(2,3) / F(4,3) / F(4,3)-gates
// output transforms (register budget of the 36-plane patch)

// ---- lane-cooperative F(6x6) transforms for SMALL launches (a few frames per call: BASELINE configs[1]) ------------------
// With ~25 k work items the kernels above are ~100 workgroups of one serial chain per thread (64 loads, ~600 VALU, 64
// stores: 10-28 us per launch, 26 launches per batch-8 forward).  Here EIGHT lanes share one (tile, 4 channels) item:
// each lane does one column in the first pass and one row in the second, with an 8x8 transpose through LDS in between --
// 8 loads and 8 stores per lane, 8x the workgroups.  Operation order per output value is exactly that of
// wino_input_kernel / wino_output_kernel (column pass, then row pass), so results are bit-identical to them.
// LDS image per item: [8][9] float4 (row stride 9: the column-major writes and the row-major reads are conflict-free;
// 288 floats per item = 32 banks past a multiple of 64, so the two items of a 16-lane access group do not collide either).
#define WINO_COOP_ITEM 288
// The 8x8 transpose of an item happens among EIGHT LANES OF ONE WAVEFRONT: LDS operations of a wavefront execute in issue order,
// so all that is needed between its writes and its reads is that the compiler keeps that order (wavefront-scope fence) -- no
// s_barrier, the wavefronts of a workgroup never wait for each other.  (Round 3 used __syncthreads() here and two LDS images per
// item in the 8-channel kernels: 36.8 KB per 128-thread workgroup = 8 wavefronts per CU; with one image reused for both channel
// halves and no barrier: 16 wavefronts per CU.)  WINO_WAVE_SYNC=0 restores the barriers (A/B).
#ifndef WINO_WAVE_SYNC
#define WINO_WAVE_SYNC 1
#endif
#ifndef WINO_ONE_IMAGE
#define WINO_ONE_IMAGE WINO_WAVE_SYNC      // 8-channel kernels: one LDS image reused for both channel halves (needs the wave-local sync)
#endif
__device__ __forceinline__ void wino_item_sync()
{
#if WINO_WAVE_SYNC
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#else
    __syncthreads();
#endif
}
#define WINO_COOP_MAX_ITEMS (768 * WINO_THREADS)     // (tile, channel-pair) items below which a launch takes the cooperative kernels
// S3: V leaves as the split-bf16 operand of wino_gemm_s3.hip, [P][3][C/16][Mp][16]; the item order then puts 4 channel
// quads x 2 tiles in a wavefront and 2 k-blocks x 2 tile pairs in a workgroup (64-byte store runs per wave-instruction,
// neighbours of the same 128-byte lines in the same workgroup, for the loads as well)
template <bool S3>
__global__ __launch_bounds__(WINO_THREADS) void wino_input_coop6_kernel(WinoArgs p)
{
    typedef VecOf<4>::T T;
    __shared__ __attribute__((aligned(16))) float s_t[(WINO_THREADS / 8) * WINO_COOP_ITEM];
    const int cq_n = p.C / 4;
    const int mt4 = (p.Mt + 3) & ~3;
    const long long items = S3 ? (long long)mt4 * cq_n : (long long)p.Mt * cq_n;
    const long long plane = (long long)p.Mt * p.C;
    const int sub = threadIdx.x & 7, slot = threadIdx.x >> 3;
    float *st = s_t + slot * WINO_COOP_ITEM;
    for (long long base = (long long)blockIdx.x * (WINO_THREADS / 8); base < items; base += (long long)gridDim.x * (WINO_THREADS / 8)) {
        const long long it = base + slot;
        bool live = it < items;
        int tile, c;
        if (S3) {
            const long long hi = it >> 5;                    // (k-block pair, tile group of 4), tile group fastest
            const int tg = (int)(hi % (mt4 >> 2)), kp = (int)(hi / (mt4 >> 2));
            tile = tg * 4 + (int)((it >> 4) & 1) * 2 + (int)((it >> 2) & 1);
            c = (kp * 2 + (int)((it >> 3) & 1)) * 16 + (int)(it & 3) * 4;
            live = live && tile < p.Mt;
            if (!live) { tile = 0; c = 0; }
        } else {
            tile = live ? (int)(it / cq_n) : 0;
            c = live ? (int)(it - (long long)tile * cq_n) * 4 : 0;
        }
        const TileId t = tile_id(p, tile);
        T col[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {           // this lane's column `sub` of the 8x8 window
            int b, h, w;
            const bool ok = live && vpixel(p, t.grp, 6 * t.ty - 1 + i, 6 * t.tx - 1 + sub, b, h, w);
            col[i] = ok ? vload_in<4>(p.in + (long long)b * p.in_bs + (long long)(h * p.W + w) * p.in_ld + c) : vzero<4>();
        }
        bt_1d<6>(col);                           // Bt d : down the column
#pragma unroll
        for (int i = 0; i < 8; ++i) vstore<4>(st + (sub * 9 + i) * 4, col[i]);      // [column][xi]
        wino_item_sync();
        T row[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) row[j] = vload<4>(st + (j * 9 + sub) * 4);     // row xi = sub: all eight columns
        bt_1d<6>(row);                           // (Bt d) B : along the row
        if (live && S3) {
            const long long term = (long long)(p.C >> 4) * p.Mp * 16;        // elements of one (plane, term)
            unsigned short *dst = p.v_s3 + ((long long)(c >> 4) * p.Mp + tile) * 16 + (c & 15);
            wino_u2 tr[8][3];       // all splits before the first store (see wino_input_s3_kernel)
#pragma unroll
            for (int j = 0; j < 8; ++j) s3_split4(row[j], tr[j]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int k = 0; k < 3; ++k)
                    s3_store<1>(dst + ((long long)(8 * sub + j) * 3 + k) * term, tr[j][k]);
        } else if (live) {
            float *dst = p.v + (long long)tile * p.C + c;
#pragma unroll
            for (int j = 0; j < 8; ++j) vstore_v<4>(dst + (long long)(8 * sub + j) * plane, row[j]);
        }
        wino_item_sync();
    }
}

// The split-operand producer proper: EIGHT channels per lane (two float4 halves through the same column / row passes), so
// that every (plane, term) store is 16 bytes and a wavefront -- 2 channel octets x 4 consecutive tiles -- writes whole
// 128-byte lines of the [P][3][C/16][Mp][16] layout; the two wavefronts of a workgroup take neighbouring K blocks (the
// other half of the same input lines).  wino_input_coop6_kernel<true> is the 4-channel form of the same thing (DT_S3_IN=0).
#define WINO_S3IN_THREADS 128
typedef unsigned int wino_u4 __attribute__((ext_vector_type(4)));
// NT = 3: three bf16 terms; NT = 2: two fp16 terms of V[p] * dt_h2_base(amax of the input) * rowfac[xi] * rowfac[nu] (dt_internal.h)
template <int TS, int NT = 3>      // TS = 6: the 8x8 window of F(6x6); TS = 4: the 6x6 window of F(4x4) (lanes 6, 7 of each group idle) -- the recurrent step
__global__ __launch_bounds__(WINO_S3IN_THREADS) void wino_input_s3_kernel(WinoArgs p)
{
    constexpr int NI = TS + 2;
    [[maybe_unused]] float h2_row = 1.0f;      // fp16 form: this lane's row factor times the tensor's power of two
    if constexpr (NT == 2) h2_row = dt_h2_base(dt_amax_read(p.amax)) * dt_h2_rowfac(TS, (int)(threadIdx.x & 7));
    typedef VecOf<4>::T T;
    constexpr int IPW = WINO_S3IN_THREADS / 8;      // items per workgroup
    // measured (profiles/r04_transform_ab.txt): the big F(6x6) launches are fastest with round 3's form -- two LDS images and a
    // workgroup barrier (11.5 vs 12.4 ms per step: the serialised halves of the one-image form cost more than its occupancy
    // returns) --, the recurrent step's small F(4x4) launch with the wave-local sync (25 vs 28 us)
    constexpr bool ONE = WINO_ONE_IMAGE && TS == 4;
    __shared__ __attribute__((aligned(16))) float s_t[ONE ? 1 : 2][IPW * WINO_COOP_ITEM];
    const int mt4 = (p.Mt + 3) & ~3;
    const long long items = (long long)mt4 * (p.C / 8);
    const int sub = threadIdx.x & 7, slot = threadIdx.x >> 3;
    float *st0 = s_t[0] + slot * WINO_COOP_ITEM;
    [[maybe_unused]] float *st1 = s_t[ONE ? 0 : 1] + slot * WINO_COOP_ITEM;
    const long long term = (long long)(p.C >> 4) * p.Mp * 16;        // elements of one (plane, term)
    for (long long base = (long long)blockIdx.x * IPW; base < items; base += (long long)gridDim.x * IPW) {
        const long long it = base + slot;
        const long long hi = it >> 4;                    // (k-block pair, tile group of 4), tile group fastest
        const int tg = (int)(hi % (mt4 >> 2)), kp = (int)(hi / (mt4 >> 2));
        int tile = tg * 4 + (int)((it >> 1) & 3);
        int c = (kp * 2 + (int)((it >> 3) & 1)) * 16 + (int)(it & 1) * 8;
        const bool live = it < items && tile < p.Mt;
        if (!live) { tile = 0; c = 0; }
        const TileId t = tile_id(p, tile);
        T ca[8], cb[8];
#pragma unroll
        for (int i = NI; i < 8; ++i) { ca[i] = vzero<4>(); cb[i] = vzero<4>(); }
#pragma unroll
        for (int i = 0; i < NI; ++i) {          // this lane's column `sub` of the window, channels c .. c+7
            int b = 0, h = 0, w = 0;
            const bool ok = live && sub < NI && vpixel(p, t.grp, TS * t.ty - 1 + i, TS * t.tx - 1 + sub, b, h, w);
            const float *src = p.in + (long long)b * p.in_bs + (long long)(h * p.W + w) * p.in_ld + c;
            ca[i] = ok ? vload_in<4>(src) : vzero<4>();
            cb[i] = ok ? vload_in<4>(src + 4) : vzero<4>();
        }
        bt_1d<TS>(ca);                           // Bt d : down the column
        bt_1d<TS>(cb);
if constexpr (ONE) {      // one LDS image, the two channel halves one after the other
#pragma unroll
        for (int i = 0; i < NI; ++i) vstore<4>(st0 + (sub * 9 + i) * 4, ca[i]);
        wino_item_sync();
#pragma unroll
        for (int j = 0; j < NI; ++j) ca[j] = vload<4>(st0 + (j * 9 + sub) * 4);
        wino_item_sync();
#pragma unroll
        for (int i = 0; i < NI; ++i) vstore<4>(st0 + (sub * 9 + i) * 4, cb[i]);
        wino_item_sync();
#pragma unroll
        for (int j = 0; j < NI; ++j) cb[j] = vload<4>(st0 + (j * 9 + sub) * 4);
        } else {
#pragma unroll
        for (int i = 0; i < NI; ++i) { vstore<4>(st0 + (sub * 9 + i) * 4, ca[i]); vstore<4>(st1 + (sub * 9 + i) * 4, cb[i]); }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NI; ++j) { ca[j] = vload<4>(st0 + (j * 9 + sub) * 4); cb[j] = vload<4>(st1 + (j * 9 + sub) * 4); }
        }
        bt_1d<TS>(ca);                           // (Bt d) B : along the row
        bt_1d<TS>(cb);
        if (live && sub < NI) {
            unsigned short *dst = p.v_s3 + ((long long)(c >> 4) * p.Mp + tile) * 16 + (c & 15);
            // every split BEFORE the first store, each result in its own registers: a VALU write to a register that a store
            // in flight still reads waits for that store (the stores then run one after the other)
            wino_u4 o[NI][NT];
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                wino_u2 ta[NT], tb[NT];
                if constexpr (NT == 2) {
                    const float f = h2_row * dt_h2_rowfac(TS, j);
                    h2_split4(ca[j] * f, ta);
                    h2_split4(cb[j] * f, tb);
                } else {
                    s3_split4(ca[j], ta);
                    s3_split4(cb[j], tb);
                }
#pragma unroll
                for (int k = 0; k < NT; ++k) o[j][k] = wino_u4{ta[k][0], ta[k][1], tb[k][0], tb[k][1]};
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int k = 0; k < NT; ++k) s3_store<1>(dst + ((long long)(NI * sub + j) * NT + k) * term, o[j][k]);
        }
        if constexpr (ONE) wino_item_sync(); else __syncthreads();
    }
}

__global__ __launch_bounds__(WINO_THREADS) void wino_output_coop6_kernel(WinoArgs p)
{
    typedef VecOf<4>::T T;
    __shared__ __attribute__((aligned(16))) float s_t[(WINO_THREADS / 8) * WINO_COOP_ITEM];
    const int nq = p.N / 4;
    const long long items = (long long)p.Mt * nq;
    const long long plane = (long long)p.Mt * p.m_ld;
    const int sub = threadIdx.x & 7, slot = threadIdx.x >> 3;
    float *st = s_t + slot * WINO_COOP_ITEM;
    float am = 0.0f;      // largest |value| this lane stored (WinoArgs::amax_out: of out2 where there is one, else of out -- like wino_output_kernel)
    const bool am_full = p.amax_out && !p.out2, am_pool = p.amax_out && p.out2;
    for (long long base = (long long)blockIdx.x * (WINO_THREADS / 8); base < items; base += (long long)gridDim.x * (WINO_THREADS / 8)) {
        const long long it = base + slot;
        const bool live = it < items;
        const int tile = live ? (int)(it / nq) : 0;
        const int c = live ? (int)(it - (long long)tile * nq) * 4 : 0;
        const TileId t = tile_id(p, tile);
        T col[8];
        const float *src = p.m + (long long)tile * p.m_ld + c;
        T bv = vzero<4>();                       // requested with the M' loads (not inside the divergent part: a second round trip per item)
        if (p.bias) bv = vload<4>(p.bias + c);
#pragma unroll
        for (int i = 0; i < 8; ++i)
            col[i] = live ? vload_nt<4>(src + (long long)(8 * i + sub) * plane) : vzero<4>();   // column nu = sub
        at_1d<6>(col);                           // At m : down the column -> rows 0..5
        asm volatile("" ::"v"(bv));
#pragma unroll
        for (int i = 0; i < 6; ++i) vstore<4>(st + (sub * 9 + i) * 4, col[i]);
        wino_item_sync();
        T row[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) row[j] = vzero<4>();
        if (sub < 6) {
#pragma unroll
            for (int j = 0; j < 8; ++j) row[j] = vload<4>(st + (j * 9 + sub) * 4);
            at_1d<6>(row);                       // (At m) A : along the row -> 6 pixels of output row `sub`
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                T v = row[j] + bv;
#pragma unroll
                for (int e = 0; e < 4; ++e) set_lane<4>(v, e, wino_leaky(lane_of<4>(v, e), p.slope));
                row[j] = v;
                int b, h, w;
                if (live && p.out && vpixel(p, t.grp, 6 * t.ty + sub, 6 * t.tx + j, b, h, w)) {
                    if (p.bias16) {
                        const int k = wino_border_case(p, h, w);
                        if (k) v = v + vload<4>(p.bias16 + (long long)k * p.N + c);
                    }
                    if (am_full) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) am = fmaxf(am, fabsf(lane_of<4>(v, e)));
                    }
                    vstore_nt<4>(p.out + (long long)b * p.out_bs + (long long)(h * p.W + w) * p.out_ld + c, v);
                }
            }
        }
        if (p.out2) {   // MaxPooling2D(2,2) (g == 1: launcher): output rows 2k, 2k+1 sit in neighbouring lanes
            const int H2 = p.H >> 1, W2 = p.W >> 1;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                T mx;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float a = fmaxf(lane_of<4>(row[2 * k], e), lane_of<4>(row[2 * k + 1], e));
                    set_lane<4>(mx, e, fmaxf(a, __shfl_xor(a, 1)));
                }
                const int h2 = 3 * t.ty + (sub >> 1), w2 = 3 * t.tx + k;
                if (live && sub < 6 && !(sub & 1) && h2 < H2 && w2 < W2) {
                    if (am_pool) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) am = fmaxf(am, fabsf(lane_of<4>(mx, e)));
                    }
                    vstore<4>(p.out2 + ((long long)(t.grp * H2 + h2) * W2 + w2) * p.out2_ld + c, mx);
                }
            }
        }
        wino_item_sync();
    }
    if (p.amax_out) dt_amax_publish(p.amax_out, am);
}

// weights of the split-bf16 GEMM: one thread per (position, output channel, 4 input channels)
__global__ __launch_bounds__(256) void wino_s3_pack_kernel(const float *u, int P, int npad, int K, unsigned short *dst)
{
    const int kq = K / 4;
    const long long n_items = (long long)P * npad * kq;
    const long long term = (long long)(K >> 4) * npad * 16;
    for (long long it = (long long)blockIdx.x * blockDim.x + threadIdx.x; it < n_items; it += (long long)gridDim.x * blockDim.x) {
        const int k = (int)(it % kq) * 4;
        const long long r = it / kq;
        const int n = (int)(r % npad), pz = (int)(r / npad);
        wino_u2 tr[3];
        s3_split4(vload<4>(u + r * K + k), tr);
#pragma unroll
        for (int t3 = 0; t3 < 3; ++t3)
            *reinterpret_cast<wino_u2 *>(dst + ((long long)pz * 3 + t3) * term + ((long long)(k >> 4) * npad + n) * 16 + (k & 15)) = tr[t3];
    }
}
int launch_wino_s3_pack(hipStream_t st, const float *u, int P, int npad, int K, unsigned short *dst)
{
    if (!u || !dst || P <= 0 || npad <= 0 || K % 16) return 2;
    const long long n_items = (long long)P * npad * (K / 4);
    long long nb = (n_items + 255) / 256;
    if (nb > 65536) nb = 65536;
    hipLaunchKernelGGL(wino_s3_pack_kernel, dim3((unsigned)nb), dim3(256), 0, st, u, P, npad, K, dst);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

// ---- fp16 form: max |x| of a tensor, weights / caller operands as two fp16 terms --------------------------------------------------------
// max |x| over rows x cols floats (row stride ld) per plane z -> slot z (DT_AMAX_WORDS words, zeroed by the caller): the integer order of
// the bits of non-negative floats is their order as numbers, so the reduction is an unsigned max and its result does not depend on
// the order of the atomics
__global__ __launch_bounds__(256) void absmax_kernel(const float *x, long long rows, int cols, long long ld, long long plane, unsigned *slots)
{
    const float *xz = x + (long long)blockIdx.y * plane;
    const int cq = cols >> 2;
    const long long items = rows * cq;
    float m = 0.0f;      // (fmaxf skips NaNs: the producers' epilogues take their maxima the same way, dt_amax_publish)
    for (long long it = (long long)blockIdx.x * blockDim.x + threadIdx.x; it < items; it += (long long)gridDim.x * blockDim.x) {
        const long long r = it / cq;
        const int c = (int)(it - r * cq) * 4;
        const VecOf<4>::T v = __builtin_nontemporal_load(reinterpret_cast<const VecOf<4>::T *>(xz + r * ld + c));
#pragma unroll
        for (int e = 0; e < 4; ++e) m = fmaxf(m, fabsf(v[e]));
    }
    if ((cols & 3) && blockIdx.x == 0)      // ragged tail columns (caller tensors)
        for (long long r = threadIdx.x; r < rows; r += blockDim.x)
            for (int c = cq * 4; c < cols; ++c) m = fmaxf(m, fabsf(xz[r * ld + c]));
    dt_amax_publish(slots + (long long)blockIdx.y * DT_AMAX_WORDS, m);
}
int launch_absmax(hipStream_t st, const float *x, long long rows, int cols, long long ld, int planes, long long plane_stride, unsigned *slots, bool zero)
{
    if (!x || !slots || rows <= 0 || cols <= 0 || planes <= 0 || planes > 65535) return 2;
    if (cols >= 4 && ((ld & 3) || (plane_stride & 3) || (reinterpret_cast<uintptr_t>(x) & 15))) return 2;
    if (zero && hipMemsetAsync(slots, 0, (size_t)planes * DT_AMAX_WORDS * sizeof(unsigned), st) != hipSuccess) return 1;
    const long long items = rows * (cols >= 4 ? cols >> 2 : 1);
    long long nb = (items + 255) / 256 / 8;      // ~8 items (128 bytes) per thread
    nb = nb < 1 ? 1 : (nb > 4096 ? 4096 : nb);
    hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)nb, (unsigned)planes), dim3(256), 0, st, x, rows, cols, ld, plane_stride, slots);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
// U [P][npad][K] fp32 -> two fp16 terms [P][2][K/16][npad][16] of U[p] * dt_h2_base(max |U[p]|); the plane maxima in `slots` (launch_absmax)
__global__ __launch_bounds__(256) void wino_h2_pack_kernel(const float *u, int P, int npad, int K, const unsigned *slots, int slot_stride, unsigned short *dst)
{
    const int kq = K / 4;
    const long long n_items = (long long)P * npad * kq;
    const long long term = (long long)(K >> 4) * npad * 16;
    for (long long it = (long long)blockIdx.x * blockDim.x + threadIdx.x; it < n_items; it += (long long)gridDim.x * blockDim.x) {
        const int k = (int)(it % kq) * 4;
        const long long r = it / kq;
        const int n = (int)(r % npad), pz = (int)(r / npad);
        unsigned am = 0;
#pragma unroll
        for (int q = 0; q < DT_AMAX_SUB; ++q) { const unsigned w = slots[pz * slot_stride + q * DT_AMAX_LINE]; am = w > am ? w : am; }
        wino_u2 tr[2];
        h2_split4(vload<4>(u + r * K + k) * dt_h2_base(am), tr);
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2)
            *reinterpret_cast<wino_u2 *>(dst + ((long long)pz * 2 + t2) * term + ((long long)(k >> 4) * npad + n) * 16 + (k & 15)) = tr[t2];
    }
}
// epilogue factor of position p: 1 / (U's power of two  x  the static row factors of V's scale); ts = 0: a plain GEMM (no row factors)
__global__ void wino_h2_pscale_kernel(const unsigned *slots, int P, int ts, float *pscale)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    unsigned am = 0;
    for (int q = 0; q < DT_AMAX_SUB; ++q) { const unsigned w = slots[p * DT_AMAX_WORDS + q * DT_AMAX_LINE]; am = w > am ? w : am; }
    const int ni = ts + 2;
    const float rf = ts ? dt_h2_rowfac(ts, p / ni) * dt_h2_rowfac(ts, p % ni) : 1.0f;
    pscale[p] = dt_h2_base_inv(am) / rf;
}
// weights (or a caller's operand, ts = 0) in the fp16 form: plane maxima -> terms + per-position epilogue factors.  `slots`: P x DT_AMAX_WORDS words of scratch
// pscale == null: ONE scale for all planes (an activation-like operand: `slots` is one slot, the maximum over the whole tensor)
int launch_wino_h2_pack(hipStream_t st, const float *u, int P, int npad, int K, int ts, unsigned *slots, unsigned short *dst, float *pscale)
{
    if (!u || !dst || !slots || P <= 0 || npad <= 0 || K % 16) return 2;
    if (int rc = pscale ? launch_absmax(st, u, npad, K, K, P, (long long)npad * K, slots) : launch_absmax(st, u, (long long)P * npad, K, K, 1, 0, slots)) return rc;
    const long long n_items = (long long)P * npad * (K / 4);
    long long nb = (n_items + 255) / 256;
    if (nb > 65536) nb = 65536;
    hipLaunchKernelGGL(wino_h2_pack_kernel, dim3((unsigned)nb), dim3(256), 0, st, u, P, npad, K, slots, pscale ? DT_AMAX_WORDS : 0, dst);
    if (pscale) hipLaunchKernelGGL(wino_h2_pscale_kernel, dim3((P + 63) / 64), dim3(64), 0, st, slots, P, ts, pscale);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

// a launch this small is latency-bound on the one-thread-per-item kernels: take the cooperative form
static inline bool wino_coop_wanted(const WinoArgs &a, long long items_pairs)
{
    return a.coop > 0 || (a.coop < 0 && items_pairs < (long long)WINO_COOP_MAX_ITEMS);
}

int launch_wino_input(hipStream_t st, const WinoArgs &a)
{
    if (a.C % 4 || a.in_ld % 4 || a.Mt <= 0 || (a.ts != 2 && a.ts != 4 && a.ts != 6) || a.g < 1) return 2;
    if (a.v_s3 && a.nt == 2) {      // the fp16 form: the cooperative 8-channel producer only
        if ((a.ts != 6 && a.ts != 4) || a.C % 32 || a.Mp < a.Mt || !a.amax) return 2;
        const long long wgs = ((long long)((a.Mt + 3) & ~3) * (a.C / 8) + WINO_S3IN_THREADS / 8 - 1) / (WINO_S3IN_THREADS / 8);
        // workgroup cap: 24576 (grid-stride beyond) instead of one workgroup per 16 items -- measured in the step on two boxes
        // (profiles/r06_experiments.txt section 9): 9.84-10.09 -> 9.34-9.45 ms per step for caps of 8192 .. 32768, back to 9.8 at 65536
        const long long cap = a.grid_in > 0 ? a.grid_in : 24576;
        const unsigned grid = (unsigned)(wgs < cap ? wgs : cap);
        if (a.ts == 6) hipLaunchKernelGGL((wino_input_s3_kernel<6, 2>), dim3(grid), dim3(WINO_S3IN_THREADS), 0, st, a);
        else hipLaunchKernelGGL((wino_input_s3_kernel<4, 2>), dim3(grid), dim3(WINO_S3IN_THREADS), 0, st, a);
    } else if (a.v_s3 && a.ts == 4) {
        if (a.C % 16 || a.Mp < a.Mt) return 2;
        if (a.C % 32 == 0 && a.coop != 0) {      // the cooperative 8-channel form (20 instead of 44 us per recurrent step at 48 clips)
            const long long wgs = ((long long)((a.Mt + 3) & ~3) * (a.C / 8) + WINO_S3IN_THREADS / 8 - 1) / (WINO_S3IN_THREADS / 8);
            hipLaunchKernelGGL(wino_input_s3_kernel<4>, dim3((unsigned)(wgs < 262144 ? wgs : 262144)), dim3(WINO_S3IN_THREADS), 0, st, a);
        } else
            hipLaunchKernelGGL((wino_input_kernel<4, 4, true>), dim3(wino_blocks((long long)a.Mt * (a.C / 4))), dim3(wino_threads((long long)a.Mt * (a.C / 4))), 0, st, a);
    } else if (a.v_s3) {
        if (a.ts != 6 || a.C % 32 || a.Mp < a.Mt) return 2;
        if (a.coop == 0) {
            // A/B (DT_WINO_COOP=0): one thread per (tile, channel pair), 256-byte runs of ONE plane per store instruction -- measured
            // SLOWER than the cooperative producer's eight 128-byte lines of eight planes (12.2 vs 11.5 ms per step)
            const long long items = (long long)a.Mt * (a.C / 2);
            hipLaunchKernelGGL((wino_input_kernel<6, 2, true>), dim3(wino_blocks(items)), dim3(wino_threads(items)), 0, st, a);
        } else if (a.coop == 2) {      // A/B: the 4-channel cooperative form
            const long long wgs = ((long long)((a.Mt + 3) & ~3) * (a.C / 4) + WINO_THREADS / 8 - 1) / (WINO_THREADS / 8);
            hipLaunchKernelGGL(wino_input_coop6_kernel<true>, dim3((unsigned)(wgs < 65536 ? wgs : 65536)), dim3(WINO_THREADS), 0, st, a);
        } else {
            const long long wgs = ((long long)((a.Mt + 3) & ~3) * (a.C / 8) + WINO_S3IN_THREADS / 8 - 1) / (WINO_S3IN_THREADS / 8);
            hipLaunchKernelGGL(wino_input_s3_kernel<6>, dim3((unsigned)(wgs < 262144 ? wgs : 262144)), dim3(WINO_S3IN_THREADS), 0, st, a);
        }
    } else if (a.ts == 6 && wino_coop_wanted(a, (long long)a.Mt * (a.C / 2))) {
        const long long wgs = ((long long)a.Mt * (a.C / 4) + WINO_THREADS / 8 - 1) / (WINO_THREADS / 8);
        hipLaunchKernelGGL(wino_input_coop6_kernel<false>, dim3((unsigned)(wgs < 65536 ? wgs : 65536)), dim3(WINO_THREADS), 0, st, a);
    } else if (a.ts == 6)
        hipLaunchKernelGGL((wino_input_kernel<6, 2>), dim3(wino_blocks((long long)a.Mt * (a.C / 2))), dim3(wino_threads((long long)a.Mt * (a.C / 2))), 0,
                           st, a);
    else if (a.ts == 2)
        hipLaunchKernelGGL((wino_input_kernel<2, 4>), dim3(wino_blocks((long long)a.Mt * (a.C / 4))), dim3(wino_threads((long long)a.Mt * (a.C / 4))), 0,
                           st, a);
    else   // 188 VGPRs, two waves per SIMD: still 7 % faster than <4,2> (1 KiB per wave per plane store)
        hipLaunchKernelGGL((wino_input_kernel<4, 4>), dim3(wino_blocks((long long)a.Mt * (a.C / 4))), dim3(wino_threads((long long)a.Mt * (a.C / 4))), 0,
                           st, a);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

// does launch_wino_output's kernel for these arguments fill WinoArgs::amax_out?  (every plain epilogue does: the thread-per-item kernels and, since round 6,
// the lane-cooperative F(6x6) kernel of the small launches, whose consumers otherwise run a stand-alone absmax pass each)
bool wino_output_fills_amax(const WinoArgs &a, int gates)
{
    return !gates && a.amax_out != nullptr;
}
int launch_wino_output(hipStream_t st, const WinoArgs &a, int gates)
{
    if (a.m_ld % 4 || a.Mt <= 0 || (a.ts != 2 && a.ts != 4 && a.ts != 6) || a.g < 1 || (a.out2 && a.g != 1)) return 2;
    if (gates) {
        if (a.N % 128 || a.out_ld % 4 || a.c_ld % 4 || a.xp_ld % 4) return 2;
        if (a.ts == 6)
            hipLaunchKernelGGL((wino_output_gates_kernel<6, 1>), dim3(wino_blocks((long long)a.Mt * (a.N / 4))), dim3(wino_threads((long long)a.Mt * (a.N / 4))), 0, st, a);
        else if (a.ts == 2)
            hipLaunchKernelGGL((wino_output_gates_kernel<2, 4>), dim3(wino_blocks((long long)a.Mt * (a.N / 16))), dim3(wino_threads((long long)a.Mt * (a.N / 16))), 0, st, a);
        else
            hipLaunchKernelGGL((wino_output_gates_kernel<4, 1>), dim3(wino_blocks((long long)a.Mt * (a.N / 4))), dim3(wino_threads((long long)a.Mt * (a.N / 4))), 0, st, a);
    } else {
        // vector stores need aligned rows; ragged N (conv_23-like heads) never takes this path
        if (a.N % 4 || (a.out && a.out_ld % 4) || (a.out2 && a.out2_ld % 4)) return 2;
        if (a.bias16 && (a.slope != 1.0f || a.out2 || !a.out)) return 2;      // the border corrections: linear, full-resolution epilogue only
        if (a.ts == 6 && wino_coop_wanted(a, (long long)a.Mt * (a.N / 2))) {
            const long long wgs = ((long long)a.Mt * (a.N / 4) + WINO_THREADS / 8 - 1) / (WINO_THREADS / 8);
            hipLaunchKernelGGL(wino_output_coop6_kernel, dim3((unsigned)(wgs < 65536 ? wgs : 65536)), dim3(WINO_THREADS), 0, st, a);
        } else if (a.ts == 6) {
            const long long items = (long long)a.Mt * (a.N / 2);
            unsigned th = wino_threads(items), nb = wino_blocks(items);
            if (th == WINO_THREADS && (a.thr_out == 64 || a.thr_out == 128)) { th = (unsigned)a.thr_out; nb = (unsigned)((items + th - 1) / th < 256 * 32 * (WINO_THREADS / th) ? (items + th - 1) / th : 256 * 32 * (WINO_THREADS / th)); }
            if (a.grid_out > 0 && nb > (unsigned)a.grid_out) nb = (unsigned)a.grid_out;
            hipLaunchKernelGGL((wino_output_kernel<6, 2>), dim3(nb), dim3(th), 0, st, a);
        }
        else if (a.ts == 2)
            hipLaunchKernelGGL((wino_output_kernel<2, 4>), dim3(wino_blocks((long long)a.Mt * (a.N / 4))), dim3(wino_threads((long long)a.Mt * (a.N / 4))),
                               0, st, a);
        else   // <4,4> measured equal
            hipLaunchKernelGGL((wino_output_kernel<4, 2>), dim3(wino_blocks((long long)a.Mt * (a.N / 2))), dim3(wino_threads((long long)a.Mt * (a.N / 2))),
                               0, st, a);
    }
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

// Host: U[p] = (G g Gt)[xi][nu] for every (cin, cout), p = (ts+2)*xi + nu, written directly in the MFMA
// kernel's 1x1 weight layout dst[p][npad][cin_dst] (for a 1x1 kernel the packed K order is the channel order).
//   cin_map[cin_dst] / n_map[npad]: source channel or -1 (zero), like pack_conv_weights; scale[cout_src]: folded BN
void wino_pack_weights(int ts, const float *hwio, int cin_src, int cout_src, const int *cin_map, int cin_dst,
                       const int *n_map, int npad, const float *scale, float *dst)
{
    static const double G2[4][3] = {{1, 0, 0}, {.5, .5, .5}, {.5, -.5, .5}, {0, 0, 1}};
    static const double G4[6][3] = {{1. / 4, 0, 0},          {-1. / 6, -1. / 6, -1. / 6}, {-1. / 6, 1. / 6, -1. / 6},
                                    {1. / 24, 1. / 12, 1. / 6}, {1. / 24, -1. / 12, 1. / 6}, {0, 0, 1}};
    static const double G6[8][3] = {{1, 0, 0},
                                    {-2. / 9, -2. / 9, -2. / 9},   {-2. / 9, 2. / 9, -2. / 9},
                                    {1. / 90, 1. / 45, 2. / 45},   {1. / 90, -1. / 45, 2. / 45},
                                    {32. / 45, 16. / 45, 8. / 45}, {32. / 45, -16. / 45, 8. / 45},
                                    {0, 0, 1}};
    const int ni = ts + 2, P = ni * ni;
    const double(*G)[3] = ts == 2 ? G2 : (ts == 4 ? G4 : G6);
    const size_t plane = (size_t)npad * cin_dst;
    auto work = [&](int n_lo, int n_hi) {
        for (int n = n_lo; n < n_hi; ++n) {
            const int ns = n_map ? n_map[n] : (n < cout_src ? n : -1);
            for (int ci = 0; ci < cin_dst; ++ci) {
                const int cs = cin_map ? cin_map[ci] : (ci < cin_src ? ci : -1);
                float *o = dst + (size_t)n * cin_dst + ci;
                if (ns < 0 || cs < 0) {
                    for (int q = 0; q < P; ++q) o[q * plane] = 0.0f;
                    continue;
                }
                const double sc = scale ? (double)scale[ns] : 1.0;
                double g[3][3], t[8][3];
                for (int k = 0; k < 9; ++k) g[k / 3][k % 3] = (double)hwio[((size_t)k * cin_src + cs) * cout_src + ns] * sc;
                for (int xi = 0; xi < ni; ++xi)
                    for (int kx = 0; kx < 3; ++kx) t[xi][kx] = G[xi][0] * g[0][kx] + G[xi][1] * g[1][kx] + G[xi][2] * g[2][kx];
                for (int xi = 0; xi < ni; ++xi)
                    for (int nu = 0; nu < ni; ++nu)
                        o[(size_t)(ni * xi + nu) * plane] = (float)(t[xi][0] * G[nu][0] + t[xi][1] * G[nu][1] + t[xi][2] * G[nu][2]);
            }
        }
    };
    const int nthreads = npad >= 64 ? 8 : 1;
    std::vector<std::thread> pool;
    for (int i = 0; i < nthreads; ++i)
        pool.emplace_back(work, (int)((long long)npad * i / nthreads), (int)((long long)npad * (i + 1) / nthreads));
    for (auto &th : pool) th.join();
}

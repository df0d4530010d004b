// recurrent.hip -- the bandwidth/latency-bound pieces of the track-state update:
//   * ConvLSTM2D cell update for t = 0 (h_{-1} = 0, so z = W*x + b only)
//     models_tracking/MultiObjDetTracker.py:176
//   * TinyTracker: GlobalMaxPooling2D / MaxPooling2D(4,4)+Flatten feature
//     reduction, the per-track LSTM(512, implementation=2) recurrent GEMV with
//     the cell update, and Dense(4, sigmoid)   models_tracking/TinyTracker.py:29-37
// Keras 2.x defaults throughout: tanh / hard_sigmoid, gate order i,f,c,o.
#include "dt_internal.h"

__device__ __forceinline__ float hard_sigmoid_r(float x)
{
    float y = __fmaf_rn(0.2f, x, 0.5f);
    return fminf(fmaxf(y, 0.0f), 1.0f);
}

// ---------------------------------------------------------------------------
// ConvLSTM t = 0: xproj is gate-interleaved [.., (j/32)*128 + g*32 + j%32]
// ---------------------------------------------------------------------------
__global__ void convlstm_gates0_kernel(const float *xproj, long long xp_bs, int xp_ld, float *cstate, long long c_bs,
                                       int c_ld, float *hout, long long h_bs, int h_ld, int B, int HW, int U)
{
    const long long total = (long long)B * HW * U;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (long long)gridDim.x * blockDim.x) {
        const int j = (int)(e % U);
        const long long r = e / U;
        const int pix = (int)(r % HW);
        const int b = (int)(r / HW);
        const float *xp = xproj + (long long)b * xp_bs + (long long)pix * xp_ld + (j >> 5) * 128 + (j & 31);
        const float gi = hard_sigmoid_r(xp[0]);
        const float gc = tanhf(xp[64]);
        const float go = hard_sigmoid_r(xp[96]);
        // c_{-1} = 0: the forget term vanishes
        const float cn = gi * gc;
        cstate[(long long)b * c_bs + (long long)pix * c_ld + j] = cn;
        hout[(long long)b * h_bs + (long long)pix * h_ld + j] = go * tanhf(cn);
    }
}

int launch_convlstm_gates_only(hipStream_t st, const float *xproj, long long xp_bs, int xp_ld, float *cstate,
                               long long c_bs, int c_ld, float *hout, long long h_bs, int h_ld, int B, int HW, int U)
{
    const long long total = (long long)B * HW * U;
    if (total <= 0) return 0;
    long long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(convlstm_gates0_kernel, dim3((unsigned)blocks), dim3(256), 0, st, xproj, xp_bs, xp_ld, cstate,
                       c_bs, c_ld, hout, h_bs, h_ld, B, HW, U);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

// ---------------------------------------------------------------------------
// GlobalMaxPooling2D: in [n][HW][C] -> out[n*out_ld + c].  HBM-bound: every
// wavefront reads 1 KiB contiguous (64 lanes x float4) per pixel; the four
// wavefronts of a block split the pixels and combine through LDS.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void global_maxpool_kernel(const float *in, int HW, int C, float *out, int out_ld)
{
    __shared__ f32x4 s_part[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c4 = (blockIdx.y * 64 + lane) * 4;
    const long long n = blockIdx.x;
    f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    if (c4 < C) {
        const float *p = in + n * HW * C + c4;
        for (int i = wave; i < HW; i += 4) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(p + (long long)i * C);
            m[0] = fmaxf(m[0], v[0]); m[1] = fmaxf(m[1], v[1]);
            m[2] = fmaxf(m[2], v[2]); m[3] = fmaxf(m[3], v[3]);
        }
    }
    s_part[wave][lane] = m;
    __syncthreads();
    if (wave == 0 && c4 < C) {
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            const f32x4 o = s_part[w][lane];
            m[0] = fmaxf(m[0], o[0]); m[1] = fmaxf(m[1], o[1]);
            m[2] = fmaxf(m[2], o[2]); m[3] = fmaxf(m[3], o[3]);
        }
        float *o = out + n * out_ld + c4;
        o[0] = m[0]; o[1] = m[1]; o[2] = m[2]; o[3] = m[3];
    }
}

int launch_global_maxpool(hipStream_t st, const float *in, int n, int HW, int C, float *out, int out_ld)
{
    if (n <= 0) return 0;
    if (C % 4 != 0) return 2;
    dim3 grid((unsigned)n, (unsigned)((C / 4 + 63) / 64));
    hipLaunchKernelGGL(global_maxpool_kernel, grid, dim3(256), 0, st, in, HW, C, out, out_ld);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

// MaxPooling2D((4,4),strides=(4,4)) + Flatten: out[n*out_ld + ((h4*W4)+w4)*C + c]
__global__ void maxpool4_flatten_kernel(const float *in, int n, int H, int W, int C, float *out, int out_ld)
{
    const int H4 = H / 4, W4 = W / 4;
    const long long total = (long long)n * H4 * W4 * C;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(e % C);
        long long r = e / C;
        const int w4 = (int)(r % W4); r /= W4;
        const int h4 = (int)(r % H4);
        const long long b = r / H4;
        float m = -INFINITY;
        for (int dy = 0; dy < 4; ++dy)
            for (int dx = 0; dx < 4; ++dx)
                m = fmaxf(m, in[((b * H + 4 * h4 + dy) * W + 4 * w4 + dx) * C + c]);
        out[b * out_ld + ((long long)h4 * W4 + w4) * C + c] = m;
    }
}

int launch_maxpool4_flatten(hipStream_t st, const float *in, int n, int H, int W, int C, float *out, int out_ld)
{
    const long long total = (long long)n * (H / 4) * (W / 4) * C;
    if (total <= 0) return 0;
    long long blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(maxpool4_flatten_kernel, dim3((unsigned)blocks), dim3(256), 0, st, in, n, H, W, C, out, out_ld);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

// strided column-block copy: dst[r*dst_ld + c] = src[r*src_ld + c], c < cols
__global__ void copy_cols_kernel(const float *src, int src_ld, float *dst, int dst_ld, long long rows, int cols)
{
    const long long total = rows * cols;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (long long)gridDim.x * blockDim.x) {
        const long long r = e / cols;
        const int c = (int)(e - r * cols);
        dst[r * dst_ld + c] = src[r * src_ld + c];
    }
}

int launch_copy_cols(hipStream_t st, const float *src, int src_ld, float *dst, int dst_ld, long long rows, int cols)
{
    const long long total = rows * cols;
    if (total <= 0) return 0;
    long long blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(copy_cols_kernel, dim3((unsigned)blocks), dim3(256), 0, st, src, src_ld, dst, dst_ld, rows, cols);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

// ---------------------------------------------------------------------------
// Per-track LSTM step (TinyTracker.py:36, implementation=2):
//   z[b] = xproj[b] (= x.W + bias, precomputed for all t by the MFMA GEMM)
//          + h_prev[b] . Ur          <- this kernel: recurrent gate GEMV
//   i,f,o = hard_sigmoid, g = tanh;  c' = f*c + i*g;  h' = o*tanh(c')
// One wavefront per hidden unit j.  Its four gate columns of Ur (packed
// [j][gate][k], k contiguous, 8 KiB) are read ONCE, coalesced, into registers
// (lane l holds k = 8l..8l+7 of each gate); then for every track the wavefront
// reads the 2 KiB h_prev row coalesced, forms four partial dot products per lane
// and reduces them with wavefront shuffles.  Lane (b mod 64) keeps track b's
// four sums, so the cell update and the stores run on all lanes at the end.
// 512 wavefronts in 256 workgroups of 2: every CU streams a 16 KiB weight slice.
// ---------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

__global__ __launch_bounds__(128) void lstm_step_kernel(const float *xproj, long long xp_bs, const float *h_prev,
                                                        long long h_bs, float *cstate, const float *ur, float *h_out,
                                                        long long ho_bs, int B, int U)
{
    const int lane = threadIdx.x & 63;
    const int j = blockIdx.x * 2 + (threadIdx.x >> 6);
    if (j >= U) return;
    const int KP = U / 64;                  // k per lane (8 for U = 512)
    // weights: [j][gate][k]
    float w[4][8];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float *wp = ur + ((long long)j * 4 + g) * U + lane * 8;
        const f32x4 a = *reinterpret_cast<const f32x4 *>(wp);
        const f32x4 b = *reinterpret_cast<const f32x4 *>(wp + 4);
        w[g][0] = a[0]; w[g][1] = a[1]; w[g][2] = a[2]; w[g][3] = a[3];
        w[g][4] = b[0]; w[g][5] = b[1]; w[g][6] = b[2]; w[g][7] = b[3];
    }
    (void)KP;
    for (int b0 = 0; b0 < B; b0 += 64) {
        float keep[4] = {0.f, 0.f, 0.f, 0.f};
        const int bn = min(64, B - b0);
        // tracks in groups of 8: the sixteen 16-byte loads of a group are issued back to back
        // (one L2 round trip per group instead of one per track), then reduced
        for (int g0 = 0; g0 < bn; g0 += 8) {
            f32x4 x0[8], x1[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int bb = min(g0 + u, bn - 1);     // clamp: tail slots re-read the last row
                const float *hp = h_prev + (long long)(b0 + bb) * h_bs + lane * 8;
                x0[u] = *reinterpret_cast<const f32x4 *>(hp);
                x1[u] = *reinterpret_cast<const f32x4 *>(hp + 4);
            }
            float v[8][4];                               // per-lane partial dot products: 8 tracks x 4 gates
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float a = x0[u][0] * w[g][0];
                    a = __fmaf_rn(x0[u][1], w[g][1], a);
                    a = __fmaf_rn(x0[u][2], w[g][2], a);
                    a = __fmaf_rn(x0[u][3], w[g][3], a);
                    a = __fmaf_rn(x1[u][0], w[g][4], a);
                    a = __fmaf_rn(x1[u][1], w[g][5], a);
                    a = __fmaf_rn(x1[u][2], w[g][6], a);
                    a = __fmaf_rn(x1[u][3], w[g][7], a);
                    v[u][g] = a;
                }
            // Wavefront reduce-scatter: three halving exchanges (lane^32, ^16, ^8) leave each lane with
            // ONE track's four gate sums over 8 lanes (16+8+4 shuffles instead of 32 butterflies of 6),
            // three butterfly steps (^4, ^2, ^1) finish them.  Track held by a lane: bits 5,4,3 of its id.
#define HALVE(OFF, N)                                                        \
    {                                                                        \
        const bool up = (lane & (OFF)) != 0;                                 \
        _Pragma("unroll") for (int u = 0; u < (N); ++u)                      \
            _Pragma("unroll") for (int g = 0; g < 4; ++g) {                  \
                const float send = up ? v[u][g] : v[u + (N)][g];             \
                const float keep = up ? v[u + (N)][g] : v[u][g];             \
                v[u][g] = keep + __shfl_xor(send, (OFF));                    \
            }                                                                \
    }
            HALVE(32, 4)
            HALVE(16, 2)
            HALVE(8, 1)
#undef HALVE
#pragma unroll
            for (int o = 4; o > 0; o >>= 1)
#pragma unroll
                for (int g = 0; g < 4; ++g) v[0][g] += __shfl_xor(v[0][g], o);
            const int tr = g0 + (((lane >> 5) & 1) << 2 | ((lane >> 4) & 1) << 1 | ((lane >> 3) & 1));
            // hand the sums to lane `tr` (the lane that owns this track in the update below)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                // source lane for track t of this group: bits 5,4,3 = t, low bits 0
                const int t = (lane - g0) & 7;
                const int src = ((t >> 2) & 1) << 5 | ((t >> 1) & 1) << 4 | (t & 1) << 3;
                const float r = __shfl(v[0][g], src);
                if (lane >= g0 && lane < g0 + 8) keep[g] = r;
            }
            (void)tr;
        }
        if (lane < bn) {
            const int b = b0 + lane;
            const float *xp = xproj + (long long)b * xp_bs;
            const float zi = keep[0] + xp[j];
            const float zf = keep[1] + xp[U + j];
            const float zc = keep[2] + xp[2 * U + j];
            const float zo = keep[3] + xp[3 * U + j];
            float *cp = cstate + (long long)b * U + j;
            const float cn = hard_sigmoid_r(zf) * (*cp) + hard_sigmoid_r(zi) * tanhf(zc);
            *cp = cn;
            h_out[(long long)b * ho_bs + j] = hard_sigmoid_r(zo) * tanhf(cn);
        }
    }
}

int launch_lstm_step(hipStream_t st, const float *xproj, long long xp_bs, const float *h_prev, long long h_bs,
                     float *cstate, const float *Ur_packed, float *h_out, long long ho_bs, int B, int U)
{
    if (U != 512) return 2;   // lane k-slice of 8 is compiled in (LSTM_UNITS = 512, config.json:19)
    if (B <= 0) return 0;
    hipLaunchKernelGGL(lstm_step_kernel, dim3((unsigned)((U + 1) / 2)), dim3(128), 0, st, xproj, xp_bs, h_prev, h_bs,
                       cstate, Ur_packed, h_out, ho_bs, B, U);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

// LSTM step for t = 0 (h_{-1} = c_{-1} = 0): elementwise on xproj
__global__ void lstm_step0_kernel(const float *xproj, long long xp_bs, float *cstate, float *h_out, long long ho_bs,
                                  int B, int U)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= B * U) return;
    const int b = e / U, j = e - b * U;
    const float *xp = xproj + (long long)b * xp_bs;
    const float cn = hard_sigmoid_r(xp[j]) * tanhf(xp[2 * U + j]);
    cstate[(long long)b * U + j] = cn;
    h_out[(long long)b * ho_bs + j] = hard_sigmoid_r(xp[3 * U + j]) * tanhf(cn);
}

int launch_lstm_step0(hipStream_t st, const float *xproj, long long xp_bs, float *cstate, float *h_out,
                      long long ho_bs, int B, int U)
{
    if (B <= 0) return 0;
    hipLaunchKernelGGL(lstm_step0_kernel, dim3((unsigned)((B * U + 255) / 256)), dim3(256), 0, st, xproj, xp_bs,
                       cstate, h_out, ho_bs, B, U);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

// ---------------------------------------------------------------------------
// Dense(O, sigmoid) over rows of h (TinyTracker.py:37): one wavefront per row,
// lanes split U, wavefront-shuffle reduction per output.  O <= 8.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dense_sigmoid_kernel(const float *h, long long h_bs, const float *Wd,
                                                            const float *bd, int B, int U, int O, float *out,
                                                            long long out_bs)
{
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= B) return;
    float acc[8];
#pragma unroll
    for (int o = 0; o < 8; ++o) acc[o] = 0.0f;
    const float *hp = h + (long long)row * h_bs;
    for (int k = lane; k < U; k += 64) {
        const float x = hp[k];
#pragma unroll
        for (int o = 0; o < 8; ++o)
            if (o < O) acc[o] = __fmaf_rn(x, Wd[(long long)k * O + o], acc[o]);
    }
#pragma unroll
    for (int o = 0; o < 8; ++o)
        if (o < O) {
            const float s = wave_sum(acc[o]) + bd[o];
            if (lane == 0) out[(long long)row * out_bs + o] = 1.0f / (1.0f + expf(-s));
        }
}

int launch_dense_sigmoid(hipStream_t st, const float *h, long long h_bs, const float *Wd, const float *bd, int B,
                         int U, int O, float *out, long long out_bs)
{
    if (B <= 0) return 0;
    if (O > 8) return 2;
    hipLaunchKernelGGL(dense_sigmoid_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, st, h, h_bs, Wd, bd, B, U, O,
                       out, out_bs);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

// ---------------------------------------------------------------------------
// TinyHeatmapTracker helpers.
// generate_heatmap_feat (utility/utils.py:53-58), called with the top-left corner of a
// centre-format box (preprocessing.py:455: x - w/2.0, y - h/2.0, w, h): float64 arithmetic,
// int() truncation toward zero, then numpy slice assignment
//   heatmap[sy:sy+sh+1, sx:sx+sw+1] = 1.0   -- INCLUDING numpy's treatment of negative slice
// bounds (a negative start counts from the end), which the reference inherits for boxes that
// stick out of the image on the top/left.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void py_slice(int start, int stop, int n, int &lo, int &hi)
{
    if (start < 0) { start += n; if (start < 0) start = 0; } else if (start > n) start = n;
    if (stop < 0) { stop += n; if (stop < 0) stop = 0; } else if (stop > n) stop = n;
    lo = start; hi = stop;
}

// box4 != null: float32 centre-format boxes (cx,cy,w,h), corner formed in float64 as the data
// generator does; xywh64 != null: float64 (det_x, det_y, det_w, det_h) exactly as the reference
// function receives them.
__global__ void heatmap_from_boxes_kernel(const float *box4, const double *xywh64, int n, int hs, float *out)
{
    const long long total = (long long)n * hs * hs;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(e % hs);
        const int y = (int)((e / hs) % hs);
        const long long b = e / ((long long)hs * hs);
        double tx, ty, w, h;
        if (box4) {
            const double cx = box4[b * 4 + 0], cy = box4[b * 4 + 1];
            w = box4[b * 4 + 2]; h = box4[b * 4 + 3];
            tx = cx - w / 2.0; ty = cy - h / 2.0;
        } else {
            tx = xywh64[b * 4 + 0]; ty = xywh64[b * 4 + 1]; w = xywh64[b * 4 + 2]; h = xywh64[b * 4 + 3];
        }
        const int sx = (int)(tx * hs), sy = (int)(ty * hs);
        const int sh = (int)(h * hs), sw = (int)(w * hs);
        int y0, y1, x0, x1;
        py_slice(sy, sy + sh + 1, hs, y0, y1);
        py_slice(sx, sx + sw + 1, hs, x0, x1);
        out[e] = (y >= y0 && y < y1 && x >= x0 && x < x1) ? 1.0f : 0.0f;
    }
}

int launch_heatmap_from_boxes(hipStream_t st, const float *box4, const double *xywh64, int n, int hs, float *out)
{
    const long long total = (long long)n * hs * hs;
    if (total <= 0) return 0;
    long long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(heatmap_from_boxes_kernel, dim3((unsigned)blocks), dim3(256), 0, st, box4, xywh64, n, hs, out);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

// generate_rectangle_from_heatmap (utility/utils.py:61-79): bounding rectangle
// (x1,y1,x2,y2) of the cells >= thresh; (hs, hs, -1, -1) when none.  One wavefront
// per heatmap, min/max reduced with wavefront shuffles.
__global__ __launch_bounds__(64) void rect_from_heatmap_kernel(const float *heat, int hs, float thresh, int *rect)
{
    const long long b = blockIdx.x;
    const int lane = threadIdx.x;
    const float *hm = heat + b * hs * hs;
    int x1 = hs, y1 = hs, x2 = -1, y2 = -1;
    for (int e = lane; e < hs * hs; e += 64)
        if (hm[e] >= thresh) {
            const int i = e / hs, j = e - i * hs;
            y1 = min(y1, i); y2 = max(y2, i); x1 = min(x1, j); x2 = max(x2, j);
        }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        x1 = min(x1, __shfl_xor(x1, o)); y1 = min(y1, __shfl_xor(y1, o));
        x2 = max(x2, __shfl_xor(x2, o)); y2 = max(y2, __shfl_xor(y2, o));
    }
    if (lane == 0) { rect[b * 4 + 0] = x1; rect[b * 4 + 1] = y1; rect[b * 4 + 2] = x2; rect[b * 4 + 3] = y2; }
}

int launch_rect_from_heatmap(hipStream_t st, const float *heat, int n, int hs, float thresh, int *rect)
{
    if (n <= 0) return 0;
    hipLaunchKernelGGL(rect_from_heatmap_kernel, dim3((unsigned)n), dim3(64), 0, st, heat, hs, thresh, rect);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

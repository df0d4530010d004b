// Training-target encoding for the YOLO loss -- the step before the
// detector when the reference fine-tunes: the object-coordinate fix at the end of
// BaseBatchGenerator.aug_image (utility/preprocessing.py:171-188) and the y / b
// construction of BatchGenerator.output_from_instance (:214-293).
//
// Python floats are float64 and every operation here is a single IEEE add / mul / div /
// compare, so the kernel is bit-exact against the reference as long as the compiler does
// not contract a*b+c (this file is built with -ffp-contract=off, like decode.hip).
//
// One workgroup per frame.  HBM-bound: y (GH*GW*NB*(5+C) float64, 115 KB at G=13, C=12) and
// b (TBB*4 float64) are produced in full -- the zero fill IS the traffic (in-kernel for small
// batches, two stream-ordered memsets ahead of the kernel for large ones); besides that the
// kernel scatters the object rows and reads 20 B per object.  Objects are evaluated in parallel (one lane each:
// rescale, clamp, flip, cell, best anchor by bbox_iou) into LDS, then lane 0 replays the
// reference's sequential side effects in object order: a later object overwrites x,y,w,h,
// conf of a shared (cell, anchor) slot while the class bits accumulate, and
// true_box_index wraps modulo TRUE_BOX_BUFFER.
#include <hip/hip_runtime.h>

#include "dt_internal.h"

#define ENC_THREADS 256

struct EncodeArgs {
    const int *objs;      // [n, cap, 5] xmin, ymin, xmax, ymax, label (-1: not in LABELS)
    const int *counts;    // [n]
    const int *dims;      // [n, 2] original (w, h)
    const double *aug;    // [n, 4] scale, offx, offy, flip   or nullptr
    int cap, GH, GW, NB, C, IH, IW, TBB;
    int fill;             // 1: the kernel zero-fills its frame itself (small batches: one launch instead of three)
    double anchors[2 * DT_MAX_ANCHOR_BOXES];
    double *y, *b;
};

// utility/utils.py:175-188 on float64
static __device__ __forceinline__ double overlap_d(double x1, double x2, double x3, double x4)
{
    if (x3 < x1) {
        if (x4 < x1) return 0.0;
        return (x2 < x4 ? x2 : x4) - x1;
    }
    if (x2 < x3) return 0.0;
    return (x2 < x4 ? x2 : x4) - x3;
}

// utility/utils.py:155-173 for BoundBox(0,0,w1,h1) against BoundBox(0,0,w2,h2)
static __device__ __forceinline__ double iou_centred_d(double w1, double h1, double w2, double h2)
{
    const double iw = overlap_d(0.0 - w1 / 2, 0.0 + w1 / 2, 0.0 - w2 / 2, 0.0 + w2 / 2);
    const double ih = overlap_d(0.0 - h1 / 2, 0.0 + h1 / 2, 0.0 - h2 / 2, 0.0 + h2 / 2);
    const double inter = iw * ih;
    return inter / (w1 * h1 + w2 * h2 - inter);
}

static __device__ __forceinline__ int fix_coord(int v, bool use_aug, double scale, int off, int image, int orig)
{
    if (use_aug) v = (int)(v * scale - off);          // preprocessing.py:174,180  int() truncates
    v = (int)(v * (double)image / orig);              // :176,182
    v = v < image ? v : image;                        // :177,183
    return v > 0 ? v : 0;
}

__global__ __launch_bounds__(ENC_THREADS) void encode_targets_kernel(EncodeArgs p)
{
    __shared__ double s_box[ENC_THREADS][4];
    __shared__ int s_slot[ENC_THREADS];    // (cell*NB + anchor), -1 when the object is skipped
    __shared__ int s_lab[ENC_THREADS];
    const int f = blockIdx.x, tid = threadIdx.x;
    const int S = 5 + p.C;
    const long long ysz = (long long)p.GH * p.GW * p.NB * S;
    double *yf = p.y + f * ysz;
    double *bf = p.b + (long long)f * p.TBB * 4;
    if (p.fill) {
        for (long long i = tid; i < ysz; i += ENC_THREADS) yf[i] = 0.0;
        for (int i = tid; i < p.TBB * 4; i += ENC_THREADS) bf[i] = 0.0;
    }

    const int w = p.dims[f * 2], h = p.dims[f * 2 + 1];
    const bool use_aug = p.aug != nullptr;
    const double scale = use_aug ? p.aug[f * 4] : 1.0;
    const int offx = use_aug ? (int)p.aug[f * 4 + 1] : 0, offy = use_aug ? (int)p.aug[f * 4 + 2] : 0;
    const bool flip = use_aug && p.aug[f * 4 + 3] > 0.5;
    const int n = min(p.counts[f], p.cap);
    const double cellw = (double)p.IW / p.GW, cellh = (double)p.IH / p.GH;
    int tbi = 0;   // lane 0 only
    for (int base = 0; base < n; base += ENC_THREADS) {
        __syncthreads();   // zero fill / previous replay done before LDS is reused
        const int k = base + tid;
        int slot = -1;
        if (k < n) {
            const int *o = p.objs + ((long long)f * p.cap + k) * 5;
            int xmin = fix_coord(o[0], use_aug, scale, offx, p.IW, w), xmax = fix_coord(o[2], use_aug, scale, offx, p.IW, w);
            const int ymin = fix_coord(o[1], use_aug, scale, offy, p.IH, h), ymax = fix_coord(o[3], use_aug, scale, offy, p.IH, h);
            if (flip) { const int t = xmin; xmin = p.IW - xmax; xmax = p.IW - t; }        // :185-188
            if (xmax > xmin && ymax > ymin && o[4] >= 0 && o[4] < p.C) {                  // :224
                const double cx = (.5 * (xmin + xmax)) / cellw, cy = (.5 * (ymin + ymax)) / cellh;
                const int gx = (int)floor(cx), gy = (int)floor(cy);
                if (gx < p.GW && gy < p.GH) {                                              // :233
                    const double cw = (xmax - xmin) / cellw, ch = (ymax - ymin) / cellh;
                    int best = -1;
                    double best_iou = -1.0;
                    for (int a = 0; a < p.NB; ++a) {                                       // :246-252
                        const double iou = iou_centred_d(cw, ch, p.anchors[2 * a], p.anchors[2 * a + 1]);
                        if (best_iou < iou) { best = a; best_iou = iou; }
                    }
                    if (best < 0) best += p.NB;                                            // python index -1
                    slot = (gy * p.GW + gx) * p.NB + best;
                    s_box[tid][0] = cx; s_box[tid][1] = cy; s_box[tid][2] = cw; s_box[tid][3] = ch;
                    s_lab[tid] = o[4];
                }
            }
        }
        s_slot[tid] = slot;
        __syncthreads();
        if (tid == 0) {
            const int m = min(ENC_THREADS, n - base);
            for (int j = 0; j < m; ++j) {
                const int sl = s_slot[j];
                if (sl < 0) continue;
                double *cell = yf + (long long)sl * S;
                cell[0] = s_box[j][0]; cell[1] = s_box[j][1]; cell[2] = s_box[j][2]; cell[3] = s_box[j][3];
                cell[4] = 1.0;                                                             // :255-256
                cell[5 + s_lab[j]] = 1.0;                                                  // :257
                double *tb = bf + tbi * 4;                                                 // :260
                tb[0] = s_box[j][0]; tb[1] = s_box[j][1]; tb[2] = s_box[j][2]; tb[3] = s_box[j][3];
                tbi = (tbi + 1) % p.TBB;                                                   // :262-263
            }
        }
    }
}

int launch_encode_targets(hipStream_t st, const int *objs, const int *counts, const int *dims, const double *aug,
                          int n, int cap, int GH, int GW, int NB, int C, int IH, int IW, int TBB,
                          const double *anchors_host, double *y, double *b)
{
    if (n <= 0) return 0;
    if (NB <= 0 || NB > DT_MAX_ANCHOR_BOXES || GH <= 0 || GW <= 0 || C <= 0 || TBB <= 0 || cap <= 0 || IH <= 0 || IW <= 0)
        return 2;
    EncodeArgs a;
    a.objs = objs; a.counts = counts; a.dims = dims; a.aug = aug;
    a.cap = cap; a.GH = GH; a.GW = GW; a.NB = NB; a.C = C; a.IH = IH; a.IW = IW; a.TBB = TBB;
    for (int i = 0; i < 2 * DT_MAX_ANCHOR_BOXES; ++i) a.anchors[i] = i < 2 * NB ? anchors_host[i] : 0.0;
    a.y = y; a.b = b;
    // the zero fill is the traffic.  Large batches: two stream-ordered memsets (4.97 TB/s measured at
    // 590 MB vs 3.7 in-kernel), the kernel then only scatters the object rows.  Small batches are
    // launch-bound, so the kernel fills its own frame and the call stays a single launch.
    const size_t ybytes = sizeof(double) * (size_t)n * GH * GW * NB * (5 + C);
    a.fill = ybytes < ((size_t)256 << 20);
    if (!a.fill) {
        if (hipMemsetAsync(y, 0, ybytes, st) != hipSuccess) return 1;
        if (hipMemsetAsync(b, 0, sizeof(double) * (size_t)n * TBB * 4, st) != hipSuccess) return 1;
    }
    hipLaunchKernelGGL(encode_targets_kernel, dim3((unsigned)n), dim3(ENC_THREADS), 0, st, a);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

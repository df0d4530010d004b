// decode.hip -- YOLO output-grid decode + per-class greedy NMS + final filter,
// and the build-defined track-ID association.  One workgroup per frame.
//
// Follows utility/utils.py:208-257 (decode_netout) with sigmoid :259, the
// GLOBAL-max/min softmax :262-270, bbox_iou :155-173, interval_overlap :175-188
// and BoundBox.get_label/get_score :128-136 (post-NMS argmax).
//
// This translation unit is compiled with -ffp-contract=off: the reference's
// numpy float32 arithmetic rounds after every multiply and add, so no fused
// multiply-add may be formed in the IoU / box math.
//
// HBM/latency-bound integer+float bookkeeping: the netout frame is staged
// through LDS in coalesced chunks ([cells][5+C] rows; the odd row length keeps
// the per-cell reads bank-conflict-free), the candidate list is compacted with
// wavefront ballots, NMS runs one class per wavefront with the suppression
// sweep spread over the 64 lanes.
#include "dt_internal.h"

#define DEC_THREADS 1024           // 16 waves: the frame is read with few bytes in flight per thread, so more threads = shorter passes
#define DEC_MAX_CELLS 1920        // 19*19*5 = 1805 fits; (5+16)*mc*4 B of LDS must stay under 160 KiB
#define DEC_CHUNK_BYTES (96 * 1024)

__device__ __forceinline__ float sigmoid_ref(float x) { return 1.0f / (1.0f + expf(-x)); }

// utils.py:175-188
__device__ __forceinline__ float interval_overlap_ref(float x1, float x2, float x3, float x4)
{
    if (x3 < x1) {
        if (x4 < x1) return 0.0f;
        return fminf(x2, x4) - x1;
    } else {
        if (x2 < x3) return 0.0f;
        return fminf(x2, x4) - x3;
    }
}

// utils.py:155-173 (centre-format boxes)
__device__ __forceinline__ float bbox_iou_ref(float ax, float ay, float aw, float ah, float bx, float by, float bw,
                                              float bh)
{
    const float x1_min = ax - aw / 2, x1_max = ax + aw / 2;
    const float y1_min = ay - ah / 2, y1_max = ay + ah / 2;
    const float x2_min = bx - bw / 2, x2_max = bx + bw / 2;
    const float y2_min = by - bh / 2, y2_max = by + bh / 2;
    const float iw = interval_overlap_ref(x1_min, x1_max, x2_min, x2_max);
    const float ih = interval_overlap_ref(y1_min, y1_max, y2_min, y2_max);
    const float inter = iw * ih;
    const float a1 = aw * ah;
    const float a2 = bw * bh;
    const float uni = (a1 + a2) - inter;
    return inter / uni;
}

__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wave_min(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o));
    return v;
}

// block-wide exclusive prefix of a 0/1 flag, in thread order; returns prefix and total
__device__ __forceinline__ int block_flag_scan(bool flag, int *s_wave_tot /*[DEC_THREADS / 64]*/, int &total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long bal = __ballot(flag);
    const int within = __popcll(bal & ((1ull << lane) - 1ull));
    __syncthreads();
    if (lane == 0) s_wave_tot[wave] = __popcll(bal);
    __syncthreads();
    int base = 0;
    total = 0;
#pragma unroll
    for (int w = 0; w < DEC_THREADS / 64; ++w) {
        const int t = s_wave_tot[w];
        if (w < wave) base += t;
        total += t;
    }
    return base + within;
}

struct DecodeArgs {
    const float *netout;
    long long frame_stride;
    int GH, GW, NB, NC;
    float obj_thr, nms_thr;
    const float *frame_thr;   // optional [batch][2] (obj, nms) per frame: one threshold pair per camera / stream
    const float *anchors;
    int cap;
    float *boxes;
    int *counts;
    float *classes;
    float *post;        // [batch][ncell][S] (user buffer or internal scratch)
    int chunk_cells;
    int nms_waves;      // wavefronts that run the per-class NMS (each needs 4 x mc floats of LDS)
    int mc;             // ncell rounded up to 64: stride of the LDS candidate / sort arrays
    int ncp;            // NC rounded up to 4: per-class candidate counters in LDS
};

__global__ __launch_bounds__(DEC_THREADS) void decode_nms_kernel(DecodeArgs p)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int S = 5 + p.NC;
    const int ncell = p.GH * p.GW * p.NB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int frame = blockIdx.x;
    const float obj_thr = p.frame_thr ? p.frame_thr[2 * frame] : p.obj_thr;
    const float nms_thr = p.frame_thr ? p.frame_thr[2 * frame + 1] : p.nms_thr;
    const float *net = p.netout + (long long)frame * p.frame_stride;
    float *post = p.post + (long long)frame * ncell * S;

    // LDS carve: candidate arrays first (persistent), then a region that is the
    // staging chunk in phase 2 and the per-wave sort lists in phase 3.
    const int MC = p.mc;
    int *s_cell = reinterpret_cast<int *>(smem);              // [MC]
    float *s_bx = smem + MC;                                  // [4][MC]
    float *s_red = s_bx + 4 * MC;                             // [32]: per-wave max, per-wave min
    int *s_tot = reinterpret_cast<int *>(s_red + 32);         // [16]
    int *s_ccnt = reinterpret_cast<int *>(s_red + 48);        // [ncp] candidates with a non-zero score per class
    float *s_dyn = s_red + 48 + p.ncp;                        // chunk / sort lists

    // ---- phase 1: global max / min of the class logits (utils.py:263-264) ----
    float vmax = -INFINITY, vmin = INFINITY;
    const int nelem = ncell * S;
    for (int c = tid; c < p.ncp; c += DEC_THREADS) s_ccnt[c] = 0;
    {
        // channel of element e is e % S; advance it incrementally (DEC_THREADS % S per step) and keep
        // eight independent loads in flight instead of one dependent load + integer modulo per element
        const int step = DEC_THREADS % S;
        int ch = tid % S;
        int e = tid;
        for (; e + 7 * DEC_THREADS < nelem; e += 8 * DEC_THREADS) {
            float v[8];
            int c4[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                v[u] = net[e + u * DEC_THREADS];
                c4[u] = ch;
                ch += step;
                if (ch >= S) ch -= S;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (c4[u] >= 5) { vmax = fmaxf(vmax, v[u]); vmin = fminf(vmin, v[u]); }
        }
        for (; e < nelem; e += DEC_THREADS) {
            if (ch >= 5) { const float v = net[e]; vmax = fmaxf(vmax, v); vmin = fminf(vmin, v); }
            ch += step;
            if (ch >= S) ch -= S;
        }
    }
    vmax = wave_max(vmax);
    vmin = wave_min(vmin);
    if (lane == 0) { s_red[wave] = vmax; s_red[16 + wave] = vmin; }
    __syncthreads();
    float gmax = s_red[0], gmn = s_red[16];
#pragma unroll
    for (int w = 1; w < DEC_THREADS / 64; ++w) { gmax = fmaxf(gmax, s_red[w]); gmn = fminf(gmn, s_red[16 + w]); }
    const float gmin = gmn - gmax;  // min(x - max)
    const bool rescale = gmin < -100.0f;   // utils.py:265-266

    // ---- phase 2: conf / class scores / threshold, candidate boxes -----------
    int ncand = 0;
    for (int c0 = 0; c0 < ncell; c0 += p.chunk_cells) {
        const int cn = min(p.chunk_cells, ncell - c0);
        __syncthreads();
        for (int e = tid; e < cn * S; e += DEC_THREADS) s_dyn[e] = net[(long long)c0 * S + e];
        __syncthreads();
        for (int lc0 = 0; lc0 < cn; lc0 += DEC_THREADS) {
            const int lc = lc0 + tid;
            bool any = false;
            float bx = 0, by = 0, bw = 0, bh = 0;
            if (lc < cn) {
                float *r = s_dyn + lc * S;
                const float conf = sigmoid_ref(r[4]);   // utils.py:214
                r[4] = conf;
                float sum = 0.0f;
                for (int c = 0; c < p.NC; ++c) {
                    float v = r[5 + c] - gmax;
                    if (rescale) v = v / gmin * -100.0f;
                    v = expf(v);
                    r[5 + c] = v;
                    sum += v;
                }
                for (int c = 0; c < p.NC; ++c) {
                    const float pr = conf * (r[5 + c] / sum);   // :215
                    const float keep = pr > obj_thr ? pr : 0.0f;   // :216
                    r[5 + c] = keep;
                    if (keep != 0.0f) { any = true; atomicAdd(&s_ccnt[c], 1); }
                }
                if (any) {   // :227-231
                    const int cell = c0 + lc;
                    const int b = cell % p.NB;
                    const int col = (cell / p.NB) % p.GW;
                    const int row = cell / (p.NB * p.GW);
                    bx = ((float)col + sigmoid_ref(r[0])) / (float)p.GW;
                    by = ((float)row + sigmoid_ref(r[1])) / (float)p.GH;
                    bw = p.anchors[2 * b + 0] * expf(r[2]) / (float)p.GW;
                    bh = p.anchors[2 * b + 1] * expf(r[3]) / (float)p.GH;
                }
            }
            int tot;
            const int slot = ncand + block_flag_scan(any, s_tot, tot);
            if (any) {
                s_cell[slot] = c0 + lc;
                s_bx[0 * MC + slot] = bx;
                s_bx[1 * MC + slot] = by;
                s_bx[2 * MC + slot] = bw;
                s_bx[3 * MC + slot] = bh;
            }
            ncand += tot;
        }
        __syncthreads();
        for (int e = tid; e < cn * S; e += DEC_THREADS) post[(long long)c0 * S + e] = s_dyn[e];
    }
    __syncthreads();   // post[] and candidate arrays visible to the whole workgroup

    // ---- phase 3: greedy NMS, one class per wavefront (utils.py:239-252) -----
    {
        // per wavefront: sorted (score,id) list + unsorted staging list, MC entries each
        volatile float *l_sc = s_dyn + wave * (4 * MC);
        volatile int *l_id = reinterpret_cast<volatile int *>(s_dyn + wave * (4 * MC) + MC);
        volatile float *u_sc = s_dyn + wave * (4 * MC) + 2 * MC;
        volatile int *u_id = reinterpret_cast<volatile int *>(s_dyn + wave * (4 * MC) + 3 * MC);
        for (int c = wave; c < p.NC && wave < p.nms_waves; c += p.nms_waves) {
            if (s_ccnt[c] < 2) continue;   // fewer than two boxes carry this class: nothing to suppress (wave-uniform)
            // gather candidates with a non-zero score for class c
            int n = 0;
            for (int k0 = 0; k0 < ncand; k0 += 64) {
                const int k = k0 + lane;
                float sc = 0.0f;
                if (k < ncand) sc = post[(long long)s_cell[k] * S + 5 + c];
                const bool nz = sc != 0.0f;
                const unsigned long long bal = __ballot(nz);
                if (nz) {
                    const int pos = n + __popcll(bal & ((1ull << lane) - 1ull));
                    u_sc[pos] = sc;
                    u_id[pos] = k;
                }
                n += __popcll(bal);
            }
            if (n < 2) continue;
            // rank sort: descending score, ties -> higher candidate index first
            for (int i0 = 0; i0 < n; i0 += 64) {
                const int i = i0 + lane;
                if (i < n) {
                    const float si = u_sc[i];
                    const int ki = u_id[i];
                    int rank = 0;
                    for (int j = 0; j < n; ++j) {
                        const float sj = u_sc[j];
                        const int kj = u_id[j];
                        rank += (sj > si || (sj == si && kj > ki)) ? 1 : 0;
                    }
                    l_sc[rank] = si;
                    l_id[rank] = ki;
                }
            }
            // greedy sweep: i sequential, j over lanes; l_sc[j] = 0 marks suppressed
            for (int i = 0; i < n - 1; ++i) {
                if (l_sc[i] == 0.0f) continue;   // wave-uniform
                const int ki = l_id[i];
                const float ax = s_bx[ki], ay = s_bx[MC + ki];
                const float aw = s_bx[2 * MC + ki], ah = s_bx[3 * MC + ki];
                for (int j = i + 1 + lane; j < n; j += 64) {
                    const int kj = l_id[j];
                    const float iou = bbox_iou_ref(ax, ay, aw, ah, s_bx[kj], s_bx[MC + kj],
                                                   s_bx[2 * MC + kj], s_bx[3 * MC + kj]);
                    if (iou >= nms_thr) {
                        l_sc[j] = 0.0f;
                        post[(long long)s_cell[kj] * S + 5 + c] = 0.0f;   // :252
                    }
                }
            }
        }
    }
    __syncthreads();

    // ---- phase 4: final filter, output in creation order (utils.py:255) ------
    int nout = 0;
    for (int k0 = 0; k0 < ncand; k0 += DEC_THREADS) {
        const int k = k0 + tid;
        bool keep = false;
        int lab = 0;
        float best = 0.0f;
        if (k < ncand) {
            const float *r = post + (long long)s_cell[k] * S;
            best = r[5];
            for (int c = 1; c < p.NC; ++c) {
                const float v = r[5 + c];
                if (v > best) { best = v; lab = c; }   // np.argmax: first maximum
            }
            keep = best > obj_thr;
        }
        int tot;
        const int slot = nout + block_flag_scan(keep, s_tot, tot);
        if (keep && slot < p.cap) {
            const int cell = s_cell[k];
            float *o = p.boxes + ((long long)frame * p.cap + slot) * DT_BOX_FLOATS;
            o[0] = s_bx[k];
            o[1] = s_bx[MC + k];
            o[2] = s_bx[2 * MC + k];
            o[3] = s_bx[3 * MC + k];
            o[4] = post[(long long)cell * S + 4];
            o[5] = (float)lab;
            o[6] = best;
            o[7] = (float)cell;
            if (p.classes != nullptr) {
                float *cl = p.classes + ((long long)frame * p.cap + slot) * p.NC;
                for (int c = 0; c < p.NC; ++c) cl[c] = post[(long long)cell * S + 5 + c];
            }
        }
        nout += tot;
    }
    if (tid == 0) p.counts[frame] = nout;
}

int launch_decode(hipStream_t st, const float *netout, long long frame_stride, int batch, int GH, int GW, int NB,
                  int NC, float obj_thr, float nms_thr, const float *anchors_dev, int cap, float *boxes,
                  int *counts, float *classes, float *post, const float *frame_thr)
{
    const int S = 5 + NC;
    const int ncell = GH * GW * NB;
    if (ncell > DEC_MAX_CELLS || batch <= 0) return 2;
    int chunk = DEC_CHUNK_BYTES / (S * (int)sizeof(float));
    chunk = (chunk / 64) * 64;
    if (chunk > DEC_THREADS) chunk = DEC_THREADS;
    if (chunk < 64) return 2;   // class count too large for the LDS staging chunk
    DecodeArgs a;
    a.netout = netout; a.frame_stride = frame_stride;
    a.GH = GH; a.GW = GW; a.NB = NB; a.NC = NC;
    a.obj_thr = obj_thr; a.nms_thr = nms_thr; a.frame_thr = frame_thr; a.anchors = anchors_dev; a.cap = cap;
    a.boxes = boxes; a.counts = counts; a.classes = classes; a.post = post; a.chunk_cells = chunk;
    const int mc = ((ncell + 63) / 64) * 64;
    a.mc = mc;
    int nms_waves = (int)((96 * 1024) / ((size_t)4 * mc * sizeof(float)));
    nms_waves = nms_waves < 1 ? 1 : (nms_waves > DEC_THREADS / 64 ? DEC_THREADS / 64 : nms_waves);
    a.nms_waves = nms_waves;
    const size_t sort_bytes = (size_t)nms_waves * 4 * mc * sizeof(float);
    const size_t chunk_bytes = (size_t)chunk * S * sizeof(float);
    a.ncp = (NC + 3) / 4 * 4;
    const size_t lds = (size_t)(5 * mc + 48 + a.ncp) * sizeof(float) + (sort_bytes > chunk_bytes ? sort_bytes : chunk_bytes);
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(decode_nms_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return 1;
        attr_done = true;
    }
    if (lds > 160 * 1024) return 2;
    hipLaunchKernelGGL(decode_nms_kernel, dim3((unsigned)batch), dim3(DEC_THREADS), lds, st, a);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

// ---------------------------------------------------------------------------
__global__ void bbox_iou_kernel(const float *pairs, int n, float *iou)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const float *p = pairs + (long long)i * 8;
        iou[i] = bbox_iou_ref(p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7]);
    }
}

int launch_bbox_iou(hipStream_t st, const float *pairs, int n, float *iou)
{
    if (n <= 0) return 0;
    hipLaunchKernelGGL(bbox_iou_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, pairs, n, iou);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

// ---------------------------------------------------------------------------
// Track identity (BUILD-DEFINED -- the reference has no association step and
// never reads `trackid`).  Specification, DESIGN.md
// "Track identity": frames in order; boxes of frame t in decode order; box i
// takes the id of the not-yet-claimed frame-(t-1) box j of the SAME label with
// the largest bbox_iou(i,j) >= thr (ties -> lowest j), else opens a new id.
// One wavefront per clip: t and i are sequential, j is spread over the lanes
// and reduced with wavefront shuffles.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(64) void associate_kernel(const float *boxes, const int *counts, int T, int cap,
                                                       float thr, int *ids, int *nids)
{
    // LDS: the previous and the current frame's boxes (x,y,w,h,label) and ids, so the
    // strictly sequential greedy loop runs on LDS latency, not on L2 round trips
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *pb = smem;                                   // [5][cap] previous frame (SoA)
    float *cb = pb + 5 * cap;                           // [5][cap] current frame
    volatile int *pid = reinterpret_cast<volatile int *>(cb + 5 * cap);   // [cap] previous ids (-1 once claimed)
    volatile int *cid = pid + cap;                      // [cap] current ids
    const int clip = blockIdx.x, lane = threadIdx.x;
    const float *bx = boxes + (long long)clip * T * cap * DT_BOX_FLOATS;
    const int *cnt = counts + (long long)clip * T;
    int *id = ids + (long long)clip * T * cap;
    int next_id = 0;
    int np = 0;
    for (int t = 0; t < T; ++t) {
        const int n = min(cnt[t], cap);
        const float *cur = bx + (long long)t * cap * DT_BOX_FLOATS;
        for (int j = lane; j < n; j += 64) {
            const float *q = cur + j * 8;
            cb[j] = q[0]; cb[cap + j] = q[1]; cb[2 * cap + j] = q[2]; cb[3 * cap + j] = q[3]; cb[4 * cap + j] = q[5];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        for (int i = 0; i < n; ++i) {
            const float ax = cb[i], ay = cb[cap + i], aw = cb[2 * cap + i], ah = cb[3 * cap + i], al = cb[4 * cap + i];
            float best = -1.0f;
            int bj = 0x7fffffff;
            for (int j = lane; j < np; j += 64) {
                if (pid[j] < 0 || pb[4 * cap + j] != al) continue;      // claimed, or another label
                const float iou = bbox_iou_ref(ax, ay, aw, ah, pb[j], pb[cap + j], pb[2 * cap + j], pb[3 * cap + j]);
                if (iou >= thr && iou > best) { best = iou; bj = j; }   // ascending j per lane: keeps lowest j on ties
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const float ob = __shfl_xor(best, o);
                const int oj = __shfl_xor(bj, o);
                if (ob > best || (ob == best && oj < bj)) { best = ob; bj = oj; }
            }
            int my_id;
            if (best >= 0.0f) {
                my_id = pid[bj];
                __builtin_amdgcn_wave_barrier();
                if (lane == 0) pid[bj] = -1;            // claimed
            } else {
                my_id = next_id++;
            }
            if (lane == 0) cid[i] = my_id;
            __builtin_amdgcn_wave_barrier();
        }
        // publish this frame's ids, then it becomes the previous frame
        for (int j = lane; j < cap; j += 64) id[t * cap + j] = j < n ? cid[j] : -1;
        for (int j = lane; j < n; j += 64) {
            pb[j] = cb[j]; pb[cap + j] = cb[cap + j]; pb[2 * cap + j] = cb[2 * cap + j];
            pb[3 * cap + j] = cb[3 * cap + j]; pb[4 * cap + j] = cb[4 * cap + j];
            pid[j] = cid[j];
        }
        np = n;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
    }
    if (lane == 0) nids[clip] = next_id;
}

int launch_associate(hipStream_t st, const float *boxes, const int *counts, int n_clips, int T, int cap, float thr,
                     int *ids, int *nids)
{
    if (n_clips <= 0) return 0;
    const size_t lds = (size_t)cap * 12 * sizeof(float);   // 2 x 5 box fields + 2 id arrays
    if (lds > 160 * 1024) return 2;
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(associate_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return 1;
        attr_done = true;
    }
    hipLaunchKernelGGL(associate_kernel, dim3((unsigned)n_clips), dim3(64), lds, st, boxes, counts, T, cap, thr, ids,
                       nids);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

// ---------------------------------------------------------------------------
// highest-score box per frame (ties -> lowest index), one wavefront per frame
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(64) void top_box_kernel(const float *boxes, const int *counts, int cap, float *out4)
{
    const int f = blockIdx.x, lane = threadIdx.x;
    const int n = min(counts[f], cap);
    const float *b = boxes + (long long)f * cap * DT_BOX_FLOATS;
    float best = -1.0f;
    int bi = 0x7fffffff;
    for (int i = lane; i < n; i += 64) {
        const float s = b[i * 8 + 6];
        if (s > best) { best = s; bi = i; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o);
        const int oi = __shfl_xor(bi, o);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if (lane < 4) out4[(long long)f * 4 + lane] = (n > 0) ? b[bi * 8 + lane] : 0.0f;
}

int launch_top_box(hipStream_t st, const float *boxes, const int *counts, int n_frames, int cap, float *out4)
{
    if (n_frames <= 0) return 0;
    hipLaunchKernelGGL(top_box_kernel, dim3((unsigned)n_frames), dim3(64), 0, st, boxes, counts, cap, out4);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

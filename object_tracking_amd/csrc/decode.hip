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
#define DEC_MAX_CELLS 1920        // 19*19*5 = 1805 fits; (5+16)*mc*4 B of LDS must stay under 160 KiB.  Larger grids (up to
#define DEC_BIG_MAX_CELLS 8192    // 8192 cells: the 13-bit cell field of the kept-score keys) run the BIG instance of the kernel, whose
                                  // per-candidate arrays and overflow sort lists live in a global scratch instead of LDS
#define DEC_CELL_BITS 13
#define DEC_CHUNK_BYTES (96 * 1024)
#define DEC_NZ_CAP 4096           // kept (cell, class) scores held in LDS for the NMS (8 B each + 16 B of sort lists)

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

// maximum of an unsigned value over the 64 lanes (wave-uniform result): four DPP row shifts leave each row's maximum in
// its last lane (lanes shifted in from outside a row read 0), four lane reads and scalar max finish
__device__ __forceinline__ unsigned wave_umax_dpp(unsigned v)
{
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false));   // row_shr:1
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false));   // row_shr:2
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false));   // row_shr:4
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false));   // row_shr:8
    const unsigned a = (unsigned)__builtin_amdgcn_readlane((int)v, 15), b = (unsigned)__builtin_amdgcn_readlane((int)v, 31);
    const unsigned c = (unsigned)__builtin_amdgcn_readlane((int)v, 47), d = (unsigned)__builtin_amdgcn_readlane((int)v, 63);
    return max(max(a, b), max(c, d));
}

__device__ __forceinline__ float readlane_f(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }

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

#ifdef DT_DEC_TIMING
// debug build only (tools/dec_timing.py): phase timestamps of frame 0's workgroup
__device__ unsigned long long g_dec_times[16];
extern "C" __attribute__((visibility("default"))) int dt_debug_dec_times(unsigned long long *dst)
{
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_dec_times), sizeof(g_dec_times)) == hipSuccess ? 0 : 1;
}
#define DEC_STAMP(k) do { if (threadIdx.x == 0 && blockIdx.x == 0) g_dec_times[k] = __builtin_readcyclecounter(); } while (0)
#else
#define DEC_STAMP(k) do { } while (0)
#endif

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
    int nzcap;          // capacity of the LDS list of non-zero (cell, class) scores; more than that: NMS gathers from global memory
    float *scratch;     // BIG instance: per frame [7][mc] candidate arrays | [2 mc] keys (u64) | [nms_waves][4][mc] overflow lists
    long long scratch_stride;   // floats per frame
};

// BIG = false: every per-candidate array in LDS (grids up to DEC_MAX_CELLS cells; the production sizes).  BIG = true: the
// same algorithm with the [mc]-sized arrays in a global scratch region of the frame (read and written by this workgroup
// only, ordered by its barriers) -- slower, but any grid up to DEC_BIG_MAX_CELLS cells decodes (utils.py:208-257 has no size limit).
template <bool BIG>
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

    // LDS carve: candidate arrays first (persistent), then a region that is the staging chunk in phase 2 and the
    // per-class sort lists in phase 3.
    const int MC = p.mc;
    float *const cand = BIG ? p.scratch + (long long)frame * p.scratch_stride : smem;   // [7][MC] candidate arrays
    int *s_cell = reinterpret_cast<int *>(cand);              // [MC] cell of candidate k
    float *s_bx = cand + MC;                                  // [4][MC] box of candidate k
    int *s_kof = reinterpret_cast<int *>(cand + 5 * MC);      // [MC] per cell: 'any class kept' flag, then its candidate index
    float *s_conf = cand + 6 * MC;                            // [MC] objectness of candidate k
    float *s_red = BIG ? smem : smem + 7 * MC;                // [32]: per-wave max, per-wave min
    int *s_tot = reinterpret_cast<int *>(s_red + 32);         // [16]
    int *s_nzn = reinterpret_cast<int *>(s_red + 48);         // [0]: number of non-zero (cell, class) scores
    int *s_ccnt = reinterpret_cast<int *>(s_red + 64);        // [ncp] candidates with a non-zero score per class
    int *s_coff = s_ccnt + p.ncp;                             // [ncp] start of the class's segment in the sort lists
    int *s_cfill = s_coff + p.ncp;                            // [ncp]
    float *s_sum = s_red + 64 + 3 * p.ncp;                    // [DEC_THREADS] softmax denominators of the chunk's cells
    unsigned *s_nzk = reinterpret_cast<unsigned *>(s_sum + DEC_THREADS);   // [nzcap] cell | class << DEC_CELL_BITS
    float *s_nzs = reinterpret_cast<float *>(s_nzk + p.nzcap);             // [nzcap] score
    float *s_dyn = s_nzs + p.nzcap;                           // chunk / sort lists
    // [MC] per candidate: max over its classes of (score bits << 32 | ~class) after the NMS; takes the place of the
    // (cell, class, score) list once that has been bucketed (8 nzcap >= 8 MC bytes: launch_decode)
    unsigned long long *s_key = BIG ? reinterpret_cast<unsigned long long *>(cand + 7 * MC) : reinterpret_cast<unsigned long long *>(s_nzk);

    DEC_STAMP(0);
    // ---- phase 1: global max / min of the class logits (utils.py:263-264) ----
    float vmax = -INFINITY, vmin = INFINITY;
    const int nelem = ncell * S;
    for (int c = tid; c < p.ncp; c += DEC_THREADS) { s_ccnt[c] = 0; s_cfill[c] = 0; }
    if (tid == 0) s_nzn[0] = 0;
    // 16-byte loads (a CU streams dword loads at ~12 B/clk only): frames are 4-byte aligned, so up to three leading
    // elements (channels 0..2 of cell 0: never class logits) and up to three trailing ones are taken apart
    const int head = (4 - (int)((reinterpret_cast<uintptr_t>(net) >> 2) & 3)) & 3;
    {
        // channel of element e is e % S; advance it incrementally and keep eight independent loads in flight
        typedef float f4 __attribute__((ext_vector_type(4)));
        const f4 *vp = reinterpret_cast<const f4 *>(net + head);
        const int nvec = (nelem - head) >> 2;
        const int step = (4 * DEC_THREADS) % S;
        int ch = (head + 4 * tid) % S;
        auto upd4 = [&](const f4 &v, int c0) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                int c = c0 + q;
                if (c >= S) c -= S;
                if (c >= 5) { vmax = fmaxf(vmax, v[q]); vmin = fminf(vmin, v[q]); }
            }
        };
        int i = tid;
        for (; i + 7 * DEC_THREADS < nvec; i += 8 * DEC_THREADS) {
            f4 v[8];
            int c8[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                v[u] = vp[i + u * DEC_THREADS];
                c8[u] = ch;
                ch += step;
                if (ch >= S) ch -= S;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) upd4(v[u], c8[u]);
        }
        for (; i < nvec; i += DEC_THREADS) {
            upd4(vp[i], ch);
            ch += step;
            if (ch >= S) ch -= S;
        }
        const int e = head + 4 * nvec + tid;      // up to three trailing elements
        if (e < nelem && e % S >= 5) { const float v = net[e]; vmax = fmaxf(vmax, v); vmin = fminf(vmin, v); }
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

    DEC_STAMP(1);
    // ---- phase 2: conf / class scores / threshold, candidate boxes -----------
    // Per staged chunk: (a) exp / sigmoid element-parallel, (b) the softmax denominator per cell, summed in class
    // order like the reference's row sum, (c) conf * (e / sum) and the threshold element-parallel -- kept scores also
    // go to the LDS list the NMS reads -- (d) one thread per cell: box of a cell with any kept class, compaction.
    int ncand = 0;
    float *const cbuf = s_dyn + ((4 - head) & 3);   // element `head` of a chunk -- 16-byte aligned in global memory -- is 16-byte aligned in LDS too
    const unsigned c_magic = 0xFFFFFFFFu / (unsigned)p.NC + 1u;   // e2 / NC == umulhi(e2, magic) for e2 * NC < 2^32 (chunk elements < 2^15); NC == 1: identity
    for (int c0 = 0; c0 < ncell; c0 += p.chunk_cells) {
        const int cn = min(p.chunk_cells, ncell - c0);
        const int ne = cn * S;
        const float *src = net + (long long)c0 * S;
        __syncthreads();
        {   // chunk_cells is a multiple of 64, so the chunk starts at the frame's alignment
            typedef float f4 __attribute__((ext_vector_type(4)));
            const f4 *vp = reinterpret_cast<const f4 *>(src + head);
            const int nvec = (ne - head) >> 2;
            int i = tid;
            for (; i + 3 * DEC_THREADS < nvec; i += 4 * DEC_THREADS) {
                f4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = vp[i + u * DEC_THREADS];
#pragma unroll
                for (int u = 0; u < 4; ++u) *reinterpret_cast<f4 *>(cbuf + head + 4 * (i + u * DEC_THREADS)) = v[u];
            }
            for (; i < nvec; i += DEC_THREADS) {
                *reinterpret_cast<f4 *>(cbuf + head + 4 * i) = vp[i];
            }
            if (tid < head) cbuf[tid] = src[tid];
            const int e = head + 4 * nvec + tid;
            if (e < ne) cbuf[e] = src[e];
        }
        __syncthreads();
        if (c0 == 0) DEC_STAMP(9);
        const int nce = cn * p.NC;                               // class elements of the chunk: (cell lc, class c) <- e2 = lc * NC + c
        // (p) objectness first: conf * softmax <= conf, so a cell with conf <= threshold keeps no class whatever its
        // logits are -- its scores are written as 0 without evaluating a single exp (most cells of a frame)
        if (tid < cn) {
            float *r = cbuf + tid * S;
            const float conf = sigmoid_ref(r[4]);               // utils.py:214
            r[4] = conf;
            s_sum[tid] = conf > obj_thr ? 0.0f : -1.0f;         // < 0: no class of this cell can pass
            s_kof[c0 + tid] = 0;
        }
        __syncthreads();
        for (int e2 = tid; e2 < nce; e2 += DEC_THREADS) {       // (a)
            const int lc = p.NC == 1 ? e2 : (int)__umulhi((unsigned)e2, c_magic);
            const int e = e2 + 5 * (lc + 1);                    // lc * S + 5 + c
            float v = 0.0f;
            if (s_sum[lc] >= 0.0f) {                            // (nearly) wave-uniform: 64 lanes span one or two cells at NC = 80
                v = cbuf[e] - gmax;
                if (rescale) v = v / gmin * -100.0f;
                v = expf(v);
            }
            cbuf[e] = v;
        }
        __syncthreads();
        if (c0 == 0) DEC_STAMP(10);
        if (tid < cn && s_sum[tid] >= 0.0f) {                   // (b)
            const float *r = cbuf + tid * S + 5;
            float sum = 0.0f;
            int c = 0;
            for (; c + 8 <= p.NC; c += 8) {                     // reads batched, adds in class order
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = r[c + u];
#pragma unroll
                for (int u = 0; u < 8; ++u) sum += v[u];
            }
            for (; c < p.NC; ++c) sum += r[c];
            s_sum[tid] = sum;                                   // >= 0 (or NaN, which is not < 0: the cell is still scored, like the reference)
        }
        __syncthreads();
        if (c0 == 0) DEC_STAMP(11);
        for (int e2 = tid; e2 < nce; e2 += DEC_THREADS) {       // (c)
            const int lc = p.NC == 1 ? e2 : (int)__umulhi((unsigned)e2, c_magic);
            const float sum = s_sum[lc];
            if (sum < 0.0f) continue;                           // its class entries are already 0
            const int c = e2 - lc * p.NC;
            const int e = e2 + 5 * (lc + 1);
            const float pr = cbuf[lc * S + 4] * (cbuf[e] / sum);           // :215
            const float keep = pr > obj_thr ? pr : 0.0f;                   // :216
            cbuf[e] = keep;
            if (keep != 0.0f) {
                s_kof[c0 + lc] = 1;
                atomicAdd(&s_ccnt[c], 1);
                const int at = atomicAdd(&s_nzn[0], 1);
                if (at < p.nzcap) { s_nzk[at] = (unsigned)(c0 + lc) | ((unsigned)c << DEC_CELL_BITS); s_nzs[at] = keep; }
            }
        }
        __syncthreads();
        if (c0 == 0) DEC_STAMP(12);
        {                                                       // (d): chunk_cells <= DEC_THREADS
            const int lc = tid;
            const bool any = lc < cn && s_kof[c0 + lc] != 0;
            float bx = 0, by = 0, bw = 0, bh = 0;
            if (any) {   // :227-231
                const float *r = cbuf + lc * S;
                const int cell = c0 + lc;
                const int b = cell % p.NB;
                const int col = (cell / p.NB) % p.GW;
                const int row = cell / (p.NB * p.GW);
                bx = ((float)col + sigmoid_ref(r[0])) / (float)p.GW;
                by = ((float)row + sigmoid_ref(r[1])) / (float)p.GH;
                bw = p.anchors[2 * b + 0] * expf(r[2]) / (float)p.GW;
                bh = p.anchors[2 * b + 1] * expf(r[3]) / (float)p.GH;
            }
            int tot;
            const int slot = ncand + block_flag_scan(any, s_tot, tot);
            if (any) {
                s_cell[slot] = c0 + lc;
                s_kof[c0 + lc] = slot;
                s_conf[slot] = cbuf[lc * S + 4];
                s_bx[0 * MC + slot] = bx;
                s_bx[1 * MC + slot] = by;
                s_bx[2 * MC + slot] = bw;
                s_bx[3 * MC + slot] = bh;
            }
            ncand += tot;
        }
        if (c0 == 0) DEC_STAMP(13);
        {
            float *dst = post + (long long)c0 * S;
            for (int e = tid; e < ne; e += DEC_THREADS) dst[e] = cbuf[e];
        }
        if (c0 == 0) DEC_STAMP(14);
    }
    __syncthreads();   // post[] and candidate arrays visible to the whole workgroup

    DEC_STAMP(2);
    // ---- phase 3: greedy NMS, one class per wavefront (utils.py:239-252) -----
    const int nzn = s_nzn[0];
    if (nzn <= p.nzcap) {
        // the kept (cell, class, score) triples are in LDS: bucket them by class -- each class's segment of U is its
        // unsorted list, the same segment of L its sorted list -- and let all 16 wavefronts take classes
        float *u_sc0 = s_dyn;
        int *u_id0 = reinterpret_cast<int *>(s_dyn + p.nzcap);
        float *l_sc0 = s_dyn + 2 * p.nzcap;
        int *l_id0 = reinterpret_cast<int *>(s_dyn + 3 * p.nzcap);
        if (wave == 0) {   // exclusive prefix of the per-class counts
            int running = 0;
            for (int cb = 0; cb < p.NC; cb += 64) {
                const int v = cb + lane < p.NC ? s_ccnt[cb + lane] : 0;
                int inc = v;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const int t = __shfl_up(inc, o);
                    if (lane >= o) inc += t;
                }
                if (cb + lane < p.NC) s_coff[cb + lane] = running + inc - v;
                running += __builtin_amdgcn_readlane(inc, 63);
            }
        }
        __syncthreads();
        for (int i = tid; i < nzn; i += DEC_THREADS) {
            const unsigned key = s_nzk[i];
            const int c = (int)(key >> DEC_CELL_BITS);
            const int at = s_coff[c] + atomicAdd(&s_cfill[c], 1);
            u_sc0[at] = s_nzs[i];
            u_id0[at] = s_kof[key & ((1u << DEC_CELL_BITS) - 1u)];
        }
        __syncthreads();
        for (int k = tid; k < ncand; k += DEC_THREADS) s_key[k] = 0ull;
        __syncthreads();
        for (int c = wave; c < p.NC; c += DEC_THREADS / 64) {
            const int n = s_ccnt[c];
            if (n == 0) continue;   // wave-uniform
            volatile float *u_sc = u_sc0 + s_coff[c];
            volatile int *u_id = u_id0 + s_coff[c];
            volatile float *l_sc = l_sc0 + s_coff[c];
            volatile int *l_id = l_id0 + s_coff[c];
            const unsigned long long ctag = 0xFFFFFFFFull - (unsigned long long)c;   // equal scores: the lower class wins (np.argmax)
            if (n == 1) {   // nothing to suppress
                if (lane == 0) atomicMax(&s_key[u_id[0]], ((unsigned long long)__float_as_uint(u_sc[0]) << 32) | ctag);
                continue;
            }
            if (n <= 64) {
                // the whole class in one wavefront's registers: lane = list entry; rank by lane broadcasts, one LDS
                // exchange into sorted order, then the greedy sweep with the suppressed set as a wave-uniform bit mask
                const bool in = lane < n;
                const float si = in ? u_sc[lane] : 0.0f;
                const int ki = in ? u_id[lane] : -1;
                int rank = 0;
                for (int j = 0; j < n; ++j) {
                    const float sj = readlane_f(si, j);
                    const int kj = __builtin_amdgcn_readlane(ki, j);
                    rank += (sj > si || (sj == si && kj > ki)) ? 1 : 0;
                }
                if (in) { l_sc[rank] = si; l_id[rank] = ki; }
                const float sc = in ? l_sc[lane] : 0.0f;
                const int kk = in ? l_id[lane] : 0;
                const float bx = s_bx[kk], by = s_bx[MC + kk], bw = s_bx[2 * MC + kk], bh = s_bx[3 * MC + kk];
                unsigned long long dead = 0ull;
                for (int i = 0; i < n - 1; ++i) {
                    if ((dead >> i) & 1ull) continue;   // wave-uniform
                    const float ax = readlane_f(bx, i), ay = readlane_f(by, i), aw = readlane_f(bw, i), ah = readlane_f(bh, i);
                    const bool hit = in && lane > i && bbox_iou_ref(ax, ay, aw, ah, bx, by, bw, bh) >= nms_thr;
                    dead |= __ballot(hit);
                    if (hit) post[(long long)s_cell[kk] * S + 5 + c] = 0.0f;   // :252
                }
                // survivors of this class compete for their box's label (BoundBox.get_label / get_score, utils.py:128-136)
                if (in && !((dead >> lane) & 1ull)) atomicMax(&s_key[kk], ((unsigned long long)__float_as_uint(sc) << 32) | ctag);
                continue;
            }
            // rank sort: descending score, ties -> higher candidate index first (the order within U does not matter)
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
            for (int j = lane; j < n; j += 64) {   // survivors -> label keys, as above
                const float sc = l_sc[j];
                if (sc != 0.0f) atomicMax(&s_key[l_id[j]], ((unsigned long long)__float_as_uint(sc) << 32) | ctag);
            }
        }
    } else {
        // more kept scores than the LDS list holds: gather each class's scores from post[] (global memory)
        // per wavefront: sorted (score,id) list + unsorted staging list, MC entries each
        float *const ovf = BIG ? cand + 9 * MC : s_dyn;     // BIG: the lists follow the candidate arrays and keys in the scratch
        volatile float *l_sc = ovf + wave * (4 * MC);
        volatile int *l_id = reinterpret_cast<volatile int *>(ovf + wave * (4 * MC) + MC);
        volatile float *u_sc = ovf + wave * (4 * MC) + 2 * MC;
        volatile int *u_id = reinterpret_cast<volatile int *>(ovf + wave * (4 * MC) + 3 * MC);
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

    DEC_STAMP(3);
    // ---- phase 4: final filter, output in creation order (utils.py:255) ------
    int nout = 0;
    for (int k0 = 0; k0 < ncand; k0 += DEC_THREADS) {
        const int k = k0 + tid;
        bool keep = false;
        int lab = 0;
        float best = 0.0f;
        if (k < ncand && nzn <= p.nzcap) {
            const unsigned long long key = s_key[k];   // 0: every class of this box was suppressed -> score 0, label 0
            best = __uint_as_float((unsigned)(key >> 32));
            lab = key ? (int)(0xFFFFFFFFu - (unsigned)key) : 0;
            keep = best > obj_thr;
        } else if (k < ncand) {
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
            o[4] = s_conf[k];
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
    {   // rows past the last box read as zero (callers may hand over uninitialised memory)
        const int first = min(nout, p.cap) * DT_BOX_FLOATS, total = p.cap * DT_BOX_FLOATS;
        float *o = p.boxes + (long long)frame * total;
        for (int e = first + tid; e < total; e += DEC_THREADS) o[e] = 0.0f;
    }
    DEC_STAMP(4);
#ifdef DT_DEC_TIMING
    if (tid == 0 && frame == 0) g_dec_times[8] = (unsigned long long)ncand;
#endif
}

// floats of global scratch per frame the BIG instance needs (0: the grid fits the LDS-resident instance)
size_t decode_scratch_floats(int GH, int GW, int NB)
{
    const int ncell = GH * GW * NB;
    if (ncell <= DEC_MAX_CELLS) return 0;
    const size_t mc = (size_t)((ncell + 63) / 64) * 64;
    return (9 + 4 * (DEC_THREADS / 64)) * mc;      // [7][mc] candidates | [mc] u64 keys | [16 waves][4][mc] overflow lists
}

int launch_decode(hipStream_t st, const float *netout, long long frame_stride, int batch, int GH, int GW, int NB,
                  int NC, float obj_thr, float nms_thr, const float *anchors_dev, int cap, float *boxes,
                  int *counts, float *classes, float *post, const float *frame_thr, float *scratch)
{
    const int S = 5 + NC;
    const int ncell = GH * GW * NB;
    if (ncell > DEC_BIG_MAX_CELLS || batch <= 0) return 2;
    const bool big = ncell > DEC_MAX_CELLS;
    if (big && !scratch) return 2;
    const int mc = ((ncell + 63) / 64) * 64;
    const int ncp = (NC + 3) / 4 * 4;
    // LDS budget: candidate arrays [7][mc] (LDS-resident instance only), counters, softmax denominators, the list of kept
    // (cell, class, score) triples, and one region that is the staging chunk in phase 2 and the sort lists in phase 3
    const size_t fixed = (size_t)((big ? 0 : 7 * mc) + 64 + 3 * ncp + DEC_THREADS) * sizeof(float);
    int nzcap = DEC_NZ_CAP;
    while (nzcap > 256 && fixed + (size_t)nzcap * 8 + (size_t)nzcap * 16 > 160 * 1024) nzcap /= 2;
    size_t dyn = 160 * 1024 - fixed - (size_t)nzcap * 8;
    if (dyn > DEC_CHUNK_BYTES) dyn = DEC_CHUNK_BYTES;
    // LDS-resident instance: the per-candidate keys take the place of the kept-score list (8 nzcap >= 8 mc bytes)
    if (dyn < (size_t)nzcap * 16 || (!big && nzcap < mc)) return 2;
    int chunk = (int)((dyn - 16) / (S * sizeof(float)));   // up to three floats of alignment padding in front of the chunk
    chunk = (chunk / 64) * 64;
    if (chunk > DEC_THREADS) chunk = DEC_THREADS;
    if (chunk < 64) return 2;   // class count too large for the LDS staging chunk
    if ((long long)chunk * S * S >= (1ll << 32)) return 2;   // range of the kernel's multiply-high division by S
    DecodeArgs a;
    a.netout = netout; a.frame_stride = frame_stride;
    a.GH = GH; a.GW = GW; a.NB = NB; a.NC = NC;
    a.obj_thr = obj_thr; a.nms_thr = nms_thr; a.frame_thr = frame_thr; a.anchors = anchors_dev; a.cap = cap;
    a.boxes = boxes; a.counts = counts; a.classes = classes; a.post = post; a.chunk_cells = chunk;
    a.mc = mc;
    a.ncp = ncp;
    a.nzcap = nzcap;
    a.scratch = scratch;
    a.scratch_stride = (long long)decode_scratch_floats(GH, GW, NB);
    // overflow path of the NMS (more kept scores than nzcap): per-wave lists of 4 x mc floats -- in the dyn region of the
    // LDS-resident instance (as many waves as fit), in the scratch of the BIG one (all 16)
    int nms_waves = big ? DEC_THREADS / 64 : (int)(dyn / ((size_t)4 * mc * sizeof(float)));
    if (nms_waves < 1) return 2;
    a.nms_waves = nms_waves > DEC_THREADS / 64 ? DEC_THREADS / 64 : nms_waves;
    const size_t lds = fixed + (size_t)nzcap * 8 + dyn;
    static PerDeviceOnce attr;
    if (attr.ensure(nullptr, [](int) {
            return hipFuncSetAttribute(reinterpret_cast<const void *>(decode_nms_kernel<false>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
                   hipFuncSetAttribute(reinterpret_cast<const void *>(decode_nms_kernel<true>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess;
        }))
        return 1;
    if (lds > 160 * 1024) return 2;
    if (big) hipLaunchKernelGGL(decode_nms_kernel<true>, dim3((unsigned)batch), dim3(DEC_THREADS), lds, st, a);
    else hipLaunchKernelGGL(decode_nms_kernel<false>, dim3((unsigned)batch), dim3(DEC_THREADS), lds, st, a);
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
    // ---- register form (every frame of the clip has at most 64 boxes, T <= 64: the tracking workloads) ----
    // Lane j holds box j of the previous and of the current frame and its id in registers; box i reaches the other lanes
    // through v_readlane, the claim is a lane-select.  No LDS, no fence: the frame loop is the wavefront's own program order.
    // The next frame's boxes are requested a whole frame ahead (address selected, not the load: lanes >= n re-read box 0), and
    // a frame's ids are stored one frame late, after the wait for the boxes -- so the loop never waits for a store.
    const int cnt_l = lane < T ? min(cnt[lane], cap) : 0;
    if (T <= 64 && !__ballot(cnt_l > 64)) {
        auto request = [&](int t, float4 &q, float &ql) {
            const int n = __builtin_amdgcn_readlane(cnt_l, t);
            const float *src = bx + ((long long)t * cap + (lane < n ? lane : 0)) * DT_BOX_FLOATS;
            q = *reinterpret_cast<const float4 *>(src);
            ql = src[5];
        };
        float4 nq; float nl;
        request(0, nq, nl);
        float pbx = 0.0f, pby = 0.0f, pbw = 0.0f, pbh = 0.0f, pbl = 0.0f;
        int pidv = -1, outv = -1;
        for (int t = 0; t < T; ++t) {
            const int n = __builtin_amdgcn_readlane(cnt_l, t);
            const float cx = nq.x, cy = nq.y, cw = nq.z, ch = nq.w, cl = nl;        // waits for frame t's boxes
            if (t + 1 < T) request(t + 1, nq, nl);
            if (t > 0)
                for (int j = lane; j < cap; j += 64) id[(t - 1) * cap + j] = j < 64 ? outv : -1;
            int cid = -1;
            for (int i = 0; i < n; ++i) {
                const float ax = readlane_f(cx, i), ay = readlane_f(cy, i), aw = readlane_f(cw, i), ah = readlane_f(ch, i);
                const float al = readlane_f(cl, i);
                // largest IoU among the unclaimed same-label boxes of the previous frame, ties -> lowest j (as below)
                const float iou = bbox_iou_ref(ax, ay, aw, ah, pbx, pby, pbw, pbh);
                const unsigned key = (lane < np && pidv >= 0 && pbl == al && iou >= thr) ? __float_as_uint(iou) + 1u : 0u;
                const unsigned m = wave_umax_dpp(key);
                int my_id;
                if (m != 0u) {
                    const int bj = __ffsll((long long)__ballot(key == m)) - 1;
                    my_id = __builtin_amdgcn_readlane(pidv, bj);
                    pidv = lane == bj ? -1 : pidv;      // claimed
                } else {
                    my_id = next_id++;
                }
                cid = lane == i ? my_id : cid;
            }
            outv = cid;                                 // lanes >= n keep -1
            pbx = cx; pby = cy; pbw = cw; pbh = ch; pbl = cl; pidv = cid; np = n;
        }
        for (int j = lane; j < cap; j += 64) id[(T - 1) * cap + j] = j < 64 ? outv : -1;
        if (lane == 0) nids[clip] = next_id;
        return;
    }
    // ---- general form (any count up to cap): previous / current frame in LDS ----
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
            // best = largest IoU among the unclaimed same-label boxes of the previous frame, ties -> lowest j.  Per chunk
            // of 64 candidates: key = IoU bits + 1 (0 = not eligible; IoU >= 0, so the bit pattern orders like the value),
            // wave maximum by DPP row shifts + four lane reads, lowest lane holding it by ballot; a later chunk must be
            // strictly better to replace an earlier one.
            unsigned bestk = 0u;
            int bj = 0;
            for (int j0 = 0; j0 < np; j0 += 64) {
                const int j = j0 + lane;
                unsigned key = 0u;
                if (j < np && pid[j] >= 0 && pb[4 * cap + j] == al) {
                    const float iou = bbox_iou_ref(ax, ay, aw, ah, pb[j], pb[cap + j], pb[2 * cap + j], pb[3 * cap + j]);
                    if (iou >= thr) key = __float_as_uint(iou) + 1u;
                }
                const unsigned m = wave_umax_dpp(key);
                if (m > bestk) {
                    bestk = m;
                    bj = j0 + __ffsll((long long)__ballot(key == m)) - 1;
                }
            }
            int my_id;
            if (bestk != 0u) {
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
    static PerDeviceOnce attr;
    if (attr.ensure(nullptr, [](int) {
            return hipFuncSetAttribute(reinterpret_cast<const void *>(associate_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       160 * 1024) != hipSuccess;
        }))
        return 1;
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

// exchange.hip -- the cross-stream detection exchange on the C ABI (north_star: "RCCL all-gather of detections over
// xGMI only for cross-stream association"; the reference has no multi-GPU code).
//
// A rank's detection table (boxes / counts / track ids / ids opened per clip, as dt_decode + dt_associate leave them)
// is packed into ONE int32 row per clip so that the exchange is a single collective per step; after the all-gather
// the rows of all ranks are compacted back into tables in global clip order and the per-clip track ids become
// globally unique (id + exclusive prefix sum of the ids opened by the clips before it).
//
//   row = [ boxes T*cap*8 (float bits) | ids T*cap | counts T | nids | valid ]      (dt_packed_row_ints)
//
// Pure data movement: HBM-bound copies, one workgroup per (row, slice).
#include "dt_internal.h"

static inline size_t row_ints(int T, int cap) { return (size_t)T * cap * 8 + (size_t)T * cap + (size_t)T + 2; }

__global__ __launch_bounds__(256) void pack_rows_kernel(const int *boxes, const int *counts, const int *ids, const int *nids,
                                                        int n_clips, int T, int cap, int *rows, long long row)
{
    const int r = blockIdx.y;
    int *dst = rows + (long long)r * row;
    const long long nb = (long long)T * cap * 8, ni = (long long)T * cap;
    const long long i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x, step = (long long)gridDim.x * blockDim.x;
    if (r >= n_clips) {                     // padding row: all zero (valid = 0)
        for (long long i = i0; i < row; i += step) dst[i] = 0;
        return;
    }
    const int *sb = boxes + (long long)r * nb, *si = ids + (long long)r * ni, *sc = counts + (long long)r * T;
    for (long long i = i0; i < nb; i += step) dst[i] = sb[i];
    for (long long i = i0; i < ni; i += step) dst[nb + i] = si[i];
    for (long long i = i0; i < T; i += step) dst[nb + ni + i] = sc[i];
    if (i0 == 0) { dst[nb + ni + T] = nids[r]; dst[nb + ni + T + 1] = 1; }
}

// one workgroup: destination slot and id offset of every row (rows with valid != 1 get slot -1), number of valid rows
__global__ __launch_bounds__(256) void unpack_plan_kernel(const int *rows, int n_rows, long long row, int *slot, long long *idoff,
                                                          int *n_valid)
{
    __shared__ int s_cnt[256];
    __shared__ long long s_ids[256];
    const int tid = threadIdx.x;
    // each thread owns a contiguous run of rows; two-level exclusive scan (n_rows is at most a few thousand)
    const int per = (n_rows + 255) / 256, lo = tid * per, hi = min(n_rows, lo + per);
    int c = 0;
    long long s = 0;
    for (int r = lo; r < hi; ++r) {
        const int *p = rows + (long long)r * row + row - 2;
        if (p[1] == 1) { ++c; s += p[0]; }
    }
    s_cnt[tid] = c; s_ids[tid] = s;
    __syncthreads();
    if (tid == 0) {
        int ac = 0;
        long long as = 0;
        for (int i = 0; i < 256; ++i) {
            const int tc = s_cnt[i];
            const long long ts = s_ids[i];
            s_cnt[i] = ac; s_ids[i] = as;
            ac += tc; as += ts;
        }
        *n_valid = ac;
    }
    __syncthreads();
    c = s_cnt[tid]; s = s_ids[tid];
    for (int r = lo; r < hi; ++r) {
        const int *p = rows + (long long)r * row + row - 2;
        if (p[1] == 1) { slot[r] = c++; idoff[r] = s; s += p[0]; }
        else { slot[r] = -1; idoff[r] = 0; }
    }
}

__global__ __launch_bounds__(256) void unpack_rows_kernel(const int *rows, long long row, const int *slot, const long long *idoff,
                                                          int T, int cap, int *boxes, int *counts, int *ids, int *nids,
                                                          long long *gids)
{
    const int r = blockIdx.y, d = slot[r];
    if (d < 0) return;
    const int *src = rows + (long long)r * row;
    const long long nb = (long long)T * cap * 8, ni = (long long)T * cap, off = idoff[r];
    const long long i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x, step = (long long)gridDim.x * blockDim.x;
    for (long long i = i0; i < nb; i += step) boxes[(long long)d * nb + i] = src[i];
    for (long long i = i0; i < ni; i += step) {
        const int v = src[nb + i];
        ids[(long long)d * ni + i] = v;
        if (gids) gids[(long long)d * ni + i] = v >= 0 ? (long long)v + off : -1ll;
    }
    for (long long i = i0; i < T; i += step) counts[(long long)d * T + i] = src[nb + ni + i];
    if (i0 == 0) nids[d] = src[nb + ni + T];
}

extern "C" size_t dt_packed_row_ints(int T, int cap)
{
    return (T > 0 && cap > 0) ? row_ints(T, cap) : 0;
}

extern "C" int dt_pack_detections(dt_ctx *ctx, const float *d_boxes, const int *d_counts, const int *d_ids, const int *d_nids,
                                  int n_clips, int T, int cap, int n_rows, int32_t *d_rows)
{
    if (!ctx || !d_rows || (n_clips > 0 && (!d_boxes || !d_counts || !d_ids || !d_nids))) return dt_fail(ctx, DT_ERR_ARG, "null argument");
    if (n_clips < 0 || T <= 0 || cap <= 0) return dt_fail(ctx, DT_ERR_ARG, "bad detection table shape");
    if (n_rows < n_clips) return dt_fail(ctx, DT_ERR_ARG, "dt_pack_detections: %d rows cannot hold %d clips (n_rows is the per-rank maximum)", n_rows, n_clips);
    if (n_rows == 0) return DT_OK;
    if (n_rows > 65535) return dt_fail(ctx, DT_ERR_ARG, "dt_pack_detections: %d rows exceed the 65535 rows one launch addresses (grid.y); exchange in slices", n_rows);
    const long long row = (long long)row_ints(T, cap);
    int gx = (int)((row + 256 * 8 - 1) / (256 * 8));
    if (gx > 64) gx = 64;
    if (gx < 1) gx = 1;
    ProfScope ps(ctx, "exchange_pack", 0.0, 8.0 * (double)n_rows * row);
    hipLaunchKernelGGL(pack_rows_kernel, dim3(gx, n_rows), dim3(256), 0, ctx->stream, reinterpret_cast<const int *>(d_boxes),
                       d_counts, d_ids, d_nids, n_clips, T, cap, d_rows, row);
    return hipGetLastError() == hipSuccess ? DT_OK : dt_fail(ctx, DT_ERR_DEVICE, "pack launch failed");
}

extern "C" int dt_unpack_detections(dt_ctx *ctx, const int32_t *d_rows, int n_rows, int T, int cap, float *d_boxes, int *d_counts,
                                    int *d_ids, int *d_nids, int64_t *d_gids, int *d_n_valid)
{
    if (!ctx || !d_rows || !d_boxes || !d_counts || !d_ids || !d_nids || !d_n_valid) return dt_fail(ctx, DT_ERR_ARG, "null argument");
    if (n_rows <= 0 || T <= 0 || cap <= 0) return dt_fail(ctx, DT_ERR_ARG, "bad detection table shape");
    if (n_rows > 65535) return dt_fail(ctx, DT_ERR_ARG, "dt_unpack_detections: %d rows exceed the 65535 rows one launch addresses (grid.y); exchange in slices", n_rows);
    const long long row = (long long)row_ints(T, cap);
    int *slot = reinterpret_cast<int *>(ws_get(ctx, "xchg_slot", (size_t)n_rows * sizeof(int)));
    long long *idoff = reinterpret_cast<long long *>(ws_get(ctx, "xchg_idoff", (size_t)n_rows * sizeof(long long)));
    if (!slot || !idoff) return DT_ERR_DEVICE;
    int gx = (int)((row + 256 * 8 - 1) / (256 * 8));
    if (gx > 64) gx = 64;
    if (gx < 1) gx = 1;
    ProfScope ps(ctx, "exchange_unpack", 0.0, 8.0 * (double)n_rows * row);
    hipLaunchKernelGGL(unpack_plan_kernel, dim3(1), dim3(256), 0, ctx->stream, d_rows, n_rows, row, slot, idoff, d_n_valid);
    if (hipGetLastError() != hipSuccess) return dt_fail(ctx, DT_ERR_DEVICE, "unpack plan launch failed");
    hipLaunchKernelGGL(unpack_rows_kernel, dim3(gx, n_rows), dim3(256), 0, ctx->stream, d_rows, row, slot, idoff, T, cap,
                       reinterpret_cast<int *>(d_boxes), d_counts, d_ids, d_nids, reinterpret_cast<long long *>(d_gids));
    return hipGetLastError() == hipSuccess ? DT_OK : dt_fail(ctx, DT_ERR_DEVICE, "unpack rows launch failed");
}

// ingest.hip -- frame ingest: the  cv2.resize(image, (IMAGE_H, IMAGE_W))  step in front of
// the detector (models_detection/KerasYOLO.py:525-528, MultiObjDetTracker.py:301-304),
// on decoded uint8 HWC frames already in HBM.  The /255. that follows it in the reference is
// fused into conv_1, so the kernel's output is the uint8 network input.
//
// Definition (parity vs OpenCV itself is unpinned: cv2 is absent): OpenCV's
// 8-bit INTER_LINEAR scheme -- half-pixel centres, no anti-aliasing, 11-bit coefficients,
// ((b0*(S0>>4))>>16 + (b1*(S1>>4))>>16 + 2) >> 2 -- restated in oracle.c:orc_resize_bilinear_u8.
// Integer arithmetic only; the coefficient tables are built on the host with the oracle's
// formula, so the kernel is bit-exact against it.
//
// HBM-bound: one thread per output pixel (3 channels), 4 source pixels read, 3 bytes written;
// a wavefront writes 192 contiguous bytes.  Algorithmic bytes per frame: Hs*Ws*3 read (each
// source pixel is touched ~(Hd*Wd*4)/(Hs*Ws) times through L1/L2) + Hd*Wd*3 written.
#include <cmath>

#include "dt_internal.h"

struct IngestArgs {
    const unsigned char *src;
    unsigned char *dst;
    int n, Hs, Ws, Hd, Wd;
    const int *xt;   // [4][Wd]: x0, x1, a0, a1
    const int *yt;   // [4][Hd]: y0, y1, b0, b1
};

__global__ __launch_bounds__(256) void ingest_resize_kernel(IngestArgs p)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    const long long f = blockIdx.z;
    if (x >= p.Wd) return;
    const int x0 = p.xt[x] * 3, x1 = p.xt[p.Wd + x] * 3, a0 = p.xt[2 * p.Wd + x], a1 = p.xt[3 * p.Wd + x];
    const int y0 = p.yt[y], y1 = p.yt[p.Hd + y], b0 = p.yt[2 * p.Hd + y], b1 = p.yt[3 * p.Hd + y];
    const unsigned char *s = p.src + f * p.Hs * p.Ws * 3;
    const unsigned char *r0 = s + (long long)y0 * p.Ws * 3, *r1 = s + (long long)y1 * p.Ws * 3;
    unsigned char *o = p.dst + ((f * p.Hd + y) * p.Wd + x) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int S0 = r0[x0 + c] * a0 + r0[x1 + c] * a1;
        const int S1 = r1[x0 + c] * a0 + r1[x1 + c] * a1;
        const int v = (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2;
        o[c] = (unsigned char)min(max(v, 0), 255);
    }
}

// same formula as oracle.c:orc_resize_tables
void ingest_tables(int src, int dst, int *tab /*[4][dst]*/)
{
    const double scale = (double)src / (double)dst;
    for (int d = 0; d < dst; ++d) {
        float f = (float)(((double)d + 0.5) * scale - 0.5);
        int s = (int)floorf(f);
        f -= (float)s;
        if (s < 0) { f = 0.0f; s = 0; }
        if (s >= src - 1) { f = 0.0f; s = src - 1; }
        tab[d] = s;
        tab[dst + d] = s + 1 < src ? s + 1 : src - 1;
        tab[2 * dst + d] = (int)lrintf((1.0f - f) * 2048.0f);
        tab[3 * dst + d] = (int)lrintf(f * 2048.0f);
    }
}

int launch_ingest_resize(hipStream_t st, const unsigned char *src, int n, int Hs, int Ws, unsigned char *dst, int Hd,
                         int Wd, const int *xt, const int *yt)
{
    if (n <= 0) return 0;
    IngestArgs a;
    a.src = src; a.dst = dst; a.n = n; a.Hs = Hs; a.Ws = Ws; a.Hd = Hd; a.Wd = Wd; a.xt = xt; a.yt = yt;
    dim3 grid((unsigned)((Wd + 255) / 256), (unsigned)Hd, (unsigned)n);
    hipLaunchKernelGGL(ingest_resize_kernel, grid, dim3(256), 0, st, a);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

/*
 * oracle.c -- CPU restatement of the reference's detect-and-track hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (object_tracking_amd/)
 * may import, link or call this file; only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg do, and only as the checker / reported baseline.
 *
 * Each function cites the reference file:line it follows (paths relative to
 * the upstream ktzsh/object-tracking tree).
 *
 * Parity status
 *   - PINNED by golden vectors produced by EXECUTING the reference's own numpy
 *     code (tools/make_goldens.py -> tests/golden/): decode_netout / NMS /
 *     bbox_iou (utility/utils.py:113-188,208-270), WeightReader (:138-148),
 *     normalize (:150-153), heat maps (:53-79), target encoding and sequence
 *     windows (utility/preprocessing.py:12-89,171-188,195-371).
 *   - GRAPH TOPOLOGY PINNED by executing the reference's own load_model bodies
 *     (KerasYOLO.py:239-407 incl. init_weights on a darknet file,
 *     MultiObjDetTracker.py:160-189, TinyTracker.py:25-41) against a float64
 *     stand-in for the Keras names they use (tools/make_graph_goldens.py +
 *     tools/kshim.py -> tests/golden/graph_*.npz): layer order, the skip tap,
 *     space_to_depth channel order, both concat orders, darknet read order and
 *     OIHW->HWIO.  This file agrees with those fixtures to 8e-5.
 *   - STILL UNPINNED: the per-layer ARITHMETIC of Keras/TensorFlow itself
 *     (inference BatchNorm with eps 1e-3, LeakyReLU, 'same' padding,
 *     hard_sigmoid, gate order i,f,c,o, NHWC space_to_depth) and cv2.resize:
 *     un-vendored, unversioned dependencies that cannot run in this image; the
 *     reference holds no tests or fixtures for them.  Both this file and kshim
 *     restate the public Keras 2.x / TF1 / OpenCV definitions, independently,
 *     and are cross-checked against torch-CPU in tests/.
 *
 * All tensors are float32, NHWC, dense.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

/* thread count of the OpenMP loops below (bench.py's cpu_baseline leg picks the count that is fastest on the host;
 * 0 = OpenMP's default).  Results do not depend on it: every output element is computed by one thread. */
#ifdef _OPENMP
#include <omp.h>
ORC_API void orc_set_threads(int n) { if (n > 0) omp_set_num_threads(n); }
ORC_API int orc_max_threads(void) { return omp_get_max_threads(); }
#else
ORC_API void orc_set_threads(int n) { (void)n; }
ORC_API int orc_max_threads(void) { return 1; }
#endif

/* utility/utils.py:150-153  normalize(): image / 255.  (numpy float64 divide,
 * cast to float32 at the Keras input boundary, KerasYOLO.py:527-531). */
ORC_API void orc_normalize_u8(const uint8_t *img, int64_t n, float *out)
{
    for (int64_t i = 0; i < n; ++i) out[i] = (float)((double)img[i] / 255.0);
}

/* Conv2D(strides=(1,1), padding='same'), kernel HWIO, optional bias.
 * KerasYOLO.py:279 (and every conv_N ctor down to :399);
 * MultiObjDetTracker.py:182 (tconv_2). */
/* One (row, x-block, cout-block) tile: XB pixels x CB output channels held in
 * accumulators while (ky, kx, ci) run in the reference order -- every output
 * element is the same fmaf chain as the naive loop nest (out-of-image taps
 * contribute fmaf(0, w, acc) = acc), the weights are just reused XB times. */
#define ORC_XB 3
#define ORC_CB 32
static void conv_tile(const float *in, int H, int W, int Cin, const float *w, int KS, int Cout,
                      const float *bias, float *out, int b, int h, int x0, int co0, int xb, int cb, const float *zrow)
{
    const int pad = KS / 2;
    float acc[ORC_XB][ORC_CB];
    for (int x = 0; x < ORC_XB; ++x)
        for (int c = 0; c < ORC_CB; ++c) acc[x][c] = (bias && c < cb) ? bias[co0 + c] : 0.0f;
    for (int ky = 0; ky < KS; ++ky) {
        const int ih = h + ky - pad;
        if (ih < 0 || ih >= H) continue;
        for (int kx = 0; kx < KS; ++kx) {
            const float *ip[ORC_XB];
            int any = 0;
            for (int x = 0; x < ORC_XB; ++x) {
                const int iw = x0 + x + kx - pad;
                const int ok = x < xb && iw >= 0 && iw < W;
                ip[x] = ok ? in + (((size_t)b * H + ih) * W + iw) * Cin : zrow;
                any |= ok;
            }
            if (!any) continue;
            const float *wp = w + ((size_t)(ky * KS + kx) * Cin) * Cout + co0;
            if (cb == ORC_CB) {
                for (int ci = 0; ci < Cin; ++ci) {
                    const float *wr = wp + (size_t)ci * Cout;
                    for (int x = 0; x < ORC_XB; ++x) {
                        const float v = ip[x][ci];
                        for (int c = 0; c < ORC_CB; ++c) acc[x][c] = fmaf(v, wr[c], acc[x][c]);
                    }
                }
            } else {
                for (int ci = 0; ci < Cin; ++ci) {
                    const float *wr = wp + (size_t)ci * Cout;
                    for (int x = 0; x < ORC_XB; ++x) {
                        const float v = ip[x][ci];
                        for (int c = 0; c < cb; ++c) acc[x][c] = fmaf(v, wr[c], acc[x][c]);
                    }
                }
            }
        }
    }
    for (int x = 0; x < xb; ++x)
        memcpy(out + (((size_t)b * H + h) * W + x0 + x) * Cout + co0, acc[x], sizeof(float) * (size_t)cb);
}

ORC_API void orc_conv2d(const float *in, int B, int H, int W, int Cin,
                        const float *w, int KS, int Cout, const float *bias,
                        float *out)
{
    const int nxb = (W + ORC_XB - 1) / ORC_XB, ncb = (Cout + ORC_CB - 1) / ORC_CB;
    const long long ntile = (long long)B * H * nxb * ncb;
    float *zrow = (float *)calloc((size_t)Cin, sizeof(float));   /* source of out-of-image taps */
#pragma omp parallel for schedule(dynamic, 8)
    for (long long t = 0; t < ntile; ++t) {
        const int cbi = (int)(t % ncb);
        long long r = t / ncb;
        const int xbi = (int)(r % nxb); r /= nxb;
        const int h = (int)(r % H);
        const int b = (int)(r / H);
        const int x0 = xbi * ORC_XB, co0 = cbi * ORC_CB;
        conv_tile(in, H, W, Cin, w, KS, Cout, bias, out, b, h, x0, co0,
                  W - x0 < ORC_XB ? W - x0 : ORC_XB, Cout - co0 < ORC_CB ? Cout - co0 : ORC_CB, zrow);
    }
    free(zrow);
}

/* BatchNormalization() in inference mode (moving statistics, Keras default
 * epsilon = 1e-3 because no epsilon= is passed) followed by LeakyReLU(alpha).
 * KerasYOLO.py:280-281.  Formulation: tf.nn.batch_normalization,
 * inv = gamma * rsqrt(var + eps); y = x * inv + (beta - mean * inv). */
ORC_API void orc_bn_leaky(float *x, int64_t npix, int C, const float *gamma,
                          const float *beta, const float *mean,
                          const float *var, float eps, float alpha)
{
    float *inv = (float *)malloc(sizeof(float) * (size_t)C);
    float *sh = (float *)malloc(sizeof(float) * (size_t)C);
    for (int c = 0; c < C; ++c) {
        inv[c] = gamma[c] * (1.0f / sqrtf(var[c] + eps));
        sh[c] = beta[c] - mean[c] * inv[c];
    }
#pragma omp parallel for schedule(static)
    for (int64_t p = 0; p < npix; ++p) {
        float *r = x + p * C;
        for (int c = 0; c < C; ++c) {
            const float y = r[c] * inv[c] + sh[c];
            r[c] = y > 0.0f ? y : alpha * y;
        }
    }
    free(inv);
    free(sh);
}

/* LeakyReLU alone (identity BN) -- helper for tests. */
ORC_API void orc_leaky(float *x, int64_t n, float alpha)
{
    for (int64_t i = 0; i < n; ++i) x[i] = x[i] > 0.0f ? x[i] : alpha * x[i];
}

/* MaxPooling2D(pool_size=(2,2)) 'valid', stride 2.  KerasYOLO.py:282. */
ORC_API void orc_maxpool2(const float *in, int B, int H, int W, int C, float *out)
{
    const int H2 = H / 2, W2 = W / 2;
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int h = 0; h < H2; ++h)
            for (int x = 0; x < W2; ++x) {
                const float *p00 = in + (((size_t)b * H + 2 * h) * W + 2 * x) * C;
                const float *p01 = p00 + C;
                const float *p10 = p00 + (size_t)W * C;
                const float *p11 = p10 + C;
                float *o = out + (((size_t)b * H2 + h) * W2 + x) * C;
                for (int c = 0; c < C; ++c) {
                    float m = p00[c];
                    if (p01[c] > m) m = p01[c];
                    if (p10[c] > m) m = p10[c];
                    if (p11[c] > m) m = p11[c];
                    o[c] = m;
                }
            }
}

/* tf.space_to_depth(x, block_size=2), NHWC:
 * out[b,h,w,(dy*2+dx)*C + c] = in[b,2h+dy,2w+dx,c].  KerasYOLO.py:241-242. */
ORC_API void orc_space_to_depth2(const float *in, int B, int H, int W, int C,
                                 float *out)
{
    const int H2 = H / 2, W2 = W / 2;
    for (int b = 0; b < B; ++b)
        for (int h = 0; h < H2; ++h)
            for (int x = 0; x < W2; ++x)
                for (int dy = 0; dy < 2; ++dy)
                    for (int dx = 0; dx < 2; ++dx)
                        memcpy(out + ((((size_t)b * H2 + h) * W2 + x) * 4 + (dy * 2 + dx)) * C,
                               in + (((size_t)b * H + 2 * h + dy) * W + 2 * x + dx) * C,
                               sizeof(float) * (size_t)C);
}

/* concatenate([a, b]) on the channel axis.  KerasYOLO.py:391 (skip first),
 * MultiObjDetTracker.py:175 (x_bbox first). */
ORC_API void orc_concat_c(const float *a, int Ca, const float *b, int Cb,
                          int64_t npix, float *out)
{
    for (int64_t p = 0; p < npix; ++p) {
        memcpy(out + p * (Ca + Cb), a + p * Ca, sizeof(float) * (size_t)Ca);
        memcpy(out + p * (Ca + Cb) + Ca, b + p * Cb, sizeof(float) * (size_t)Cb);
    }
}

static inline float hard_sigmoid(float x)
{
    /* Keras 2.x K.hard_sigmoid: clip(0.2*x + 0.5, 0, 1) */
    float y = 0.2f * x + 0.5f;
    return y < 0.0f ? 0.0f : (y > 1.0f ? 1.0f : y);
}

/* One time step of ConvLSTM2D(U,(3,3),padding='same'), Keras 2.x defaults:
 * activation tanh, recurrent_activation hard_sigmoid, bias on the input conv
 * only, gate order i,f,c,o on the last kernel axis.
 * MultiObjDetTracker.py:176.
 *   z = conv(x, Wk) + bias + conv(h, Uk);  i,f,o = hs(z_i,z_f,z_o)
 *   c' = f*c + i*tanh(z_c);  h' = o*tanh(c')
 * x [B,H,W,Cx]; h,c [B,H,W,U]; Wk [3,3,Cx,4U]; Uk [3,3,U,4U]; bias [4U]. */
ORC_API void orc_convlstm_step(const float *x, int B, int H, int W, int Cx,
                               const float *h, const float *c, int U,
                               const float *Wk, const float *Uk,
                               const float *bias, float *h_out, float *c_out)
{
    const size_t npix = (size_t)B * H * W;
    float *zx = (float *)malloc(sizeof(float) * npix * 4 * U);
    float *zh = (float *)malloc(sizeof(float) * npix * 4 * U);
    orc_conv2d(x, B, H, W, Cx, Wk, 3, 4 * U, bias, zx);
    orc_conv2d(h, B, H, W, U, Uk, 3, 4 * U, NULL, zh);
#pragma omp parallel for schedule(static)
    for (int64_t p = 0; p < (int64_t)npix; ++p) {
        const float *a = zx + p * 4 * U, *r = zh + p * 4 * U;
        for (int j = 0; j < U; ++j) {
            const float gi = hard_sigmoid(a[j] + r[j]);
            const float gf = hard_sigmoid(a[U + j] + r[U + j]);
            const float gc = tanhf(a[2 * U + j] + r[2 * U + j]);
            const float go = hard_sigmoid(a[3 * U + j] + r[3 * U + j]);
            const float cn = gf * c[p * U + j] + gi * gc;
            c_out[p * U + j] = cn;
            h_out[p * U + j] = go * tanhf(cn);
        }
    }
    free(zx);
    free(zh);
}

/* One time step of LSTM(U, implementation=2), Keras 2.x defaults (tanh /
 * hard_sigmoid, gate order i,f,c,o): z = x.W + h.Ur + b in one fused matmul.
 * models_tracking/TinyTracker.py:36.
 * x [B,D]; h,c [B,U]; Wk [D,4U]; Ur [U,4U]; bias [4U]. */
ORC_API void orc_lstm_step(const float *x, int B, int D, const float *h,
                           const float *c, int U, const float *Wk,
                           const float *Ur, const float *bias, float *h_out,
                           float *c_out)
{
    float *z = (float *)malloc(sizeof(float) * (size_t)4 * U);
    for (int b = 0; b < B; ++b) {
        for (int n = 0; n < 4 * U; ++n) z[n] = bias[n];
        for (int d = 0; d < D; ++d) {
            const float v = x[(size_t)b * D + d];
            const float *wr = Wk + (size_t)d * 4 * U;
            for (int n = 0; n < 4 * U; ++n) z[n] += v * wr[n];
        }
        for (int d = 0; d < U; ++d) {
            const float v = h[(size_t)b * U + d];
            const float *wr = Ur + (size_t)d * 4 * U;
            for (int n = 0; n < 4 * U; ++n) z[n] += v * wr[n];
        }
        for (int j = 0; j < U; ++j) {
            const float gi = hard_sigmoid(z[j]);
            const float gf = hard_sigmoid(z[U + j]);
            const float gc = tanhf(z[2 * U + j]);
            const float go = hard_sigmoid(z[3 * U + j]);
            const float cn = gf * c[(size_t)b * U + j] + gi * gc;
            c_out[(size_t)b * U + j] = cn;
            h_out[(size_t)b * U + j] = go * tanhf(cn);
        }
    }
    free(z);
}

/* Dense(O, activation='sigmoid').  TinyTracker.py:37. */
ORC_API void orc_dense_sigmoid(const float *x, int B, int U, const float *Wd,
                               const float *bd, int O, float *out)
{
    for (int b = 0; b < B; ++b)
        for (int o = 0; o < O; ++o) {
            float s = bd[o];
            for (int j = 0; j < U; ++j) s += x[(size_t)b * U + j] * Wd[(size_t)j * O + o];
            out[(size_t)b * O + o] = 1.0f / (1.0f + expf(-s));
        }
}

/* GlobalMaxPooling2D over (w,h).  TinyTracker.py:33 (pool == 'Global'). */
ORC_API void orc_global_maxpool(const float *in, int B, int HW, int C, float *out)
{
    for (int b = 0; b < B; ++b) {
        float *o = out + (size_t)b * C;
        const float *p = in + (size_t)b * HW * C;
        for (int c = 0; c < C; ++c) o[c] = p[c];
        for (int i = 1; i < HW; ++i)
            for (int c = 0; c < C; ++c)
                if (p[(size_t)i * C + c] > o[c]) o[c] = p[(size_t)i * C + c];
    }
}

/* MaxPooling2D((4,4), strides=(4,4)) + Flatten.  TinyTracker.py:29-31
 * (pool == 'Max').  Output order is Keras Flatten of (H/4, W/4, C). */
ORC_API void orc_maxpool4_flatten(const float *in, int B, int H, int W, int C,
                                  float *out)
{
    const int H4 = H / 4, W4 = W / 4;
    for (int b = 0; b < B; ++b)
        for (int h = 0; h < H4; ++h)
            for (int x = 0; x < W4; ++x)
                for (int c = 0; c < C; ++c) {
                    float m = -INFINITY;
                    for (int dy = 0; dy < 4; ++dy)
                        for (int dx = 0; dx < 4; ++dx) {
                            const float v = in[(((size_t)b * H + 4 * h + dy) * W + 4 * x + dx) * C + c];
                            if (v > m) m = v;
                        }
                    out[(((size_t)b * H4 + h) * W4 + x) * C + c] = m;
                }
}

/* ---------------------------------------------------------------------------
 * decode_netout + NMS.  utility/utils.py:208-257 with sigmoid :259,
 * softmax :262-270, BoundBox :113-136, bbox_iou :155-173,
 * interval_overlap :175-188.  float32 arithmetic throughout (numpy >= 2
 * scalar promotion keeps np.float32 op python-scalar in float32).
 * ------------------------------------------------------------------------- */

/* utils.py:175-188 */
static float interval_overlap(float x1, float x2, float x3, float x4)
{
    if (x3 < x1) {
        if (x4 < x1) return 0.0f;
        return (x2 < x4 ? x2 : x4) - x1;
    } else {
        if (x2 < x3) return 0.0f;
        return (x2 < x4 ? x2 : x4) - x3;
    }
}

/* utils.py:155-173.  Boxes are (x,y,w,h) centre format. */
ORC_API float orc_bbox_iou(const float *b1, const float *b2)
{
    const float x1_min = b1[0] - b1[2] / 2, x1_max = b1[0] + b1[2] / 2;
    const float y1_min = b1[1] - b1[3] / 2, y1_max = b1[1] + b1[3] / 2;
    const float x2_min = b2[0] - b2[2] / 2, x2_max = b2[0] + b2[2] / 2;
    const float y2_min = b2[1] - b2[3] / 2, y2_max = b2[1] + b2[3] / 2;
    const float iw = interval_overlap(x1_min, x1_max, x2_min, x2_max);
    const float ih = interval_overlap(y1_min, y1_max, y2_min, y2_max);
    const float inter = iw * ih;
    const float a1 = b1[2] * b1[3];
    const float a2 = b2[2] * b2[3];
    const float uni = (a1 + a2) - inter;
    return inter / uni;
}

static float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); } /* utils.py:259 */

typedef struct {
    float score;
    int idx;
} sc_t;

/* descending score; ties -> higher candidate index first (the reference's
 * reversed(np.argsort) leaves tie order undefined, utils.py:240; this is the
 * build's documented choice). */
static int sc_cmp(const void *a, const void *b)
{
    const sc_t *p = (const sc_t *)a, *q = (const sc_t *)b;
    if (p->score > q->score) return -1;
    if (p->score < q->score) return 1;
    return q->idx - p->idx;
}

/* netout [GH,GW,NB,5+NC] is transformed IN PLACE exactly like the reference
 * (utils.py:214-216 and the NMS zeroing :252 through the `classes` views).
 * Output rows, in creation (row,col,b) order, for boxes surviving the final
 * filter (utils.py:255): out_box[i*8 + {0..7}] = x,y,w,h,conf,label,score,cell
 * where cell = (row*GW+col)*NB+b.  Returns the number of surviving boxes (may
 * exceed cap; only the first cap are written). */
ORC_API int orc_decode_netout(float *netout, int GH, int GW, int NB, int NC,
                              float obj_thr, float nms_thr,
                              const float *anchors, float *out_box, int cap)
{
    const int S = 5 + NC;
    const int ncell = GH * GW * NB;
    /* softmax(): GLOBAL max / min over the whole class block (utils.py:263-266) */
    float gmax = -INFINITY;
    for (int i = 0; i < ncell; ++i)
        for (int c = 0; c < NC; ++c)
            if (netout[i * S + 5 + c] > gmax) gmax = netout[i * S + 5 + c];
    float gmin = INFINITY;
    for (int i = 0; i < ncell; ++i)
        for (int c = 0; c < NC; ++c) {
            const float v = netout[i * S + 5 + c] - gmax;
            if (v < gmin) gmin = v;
        }
    const int rescale = gmin < -100.0f;
    for (int i = 0; i < ncell; ++i) {
        float *r = netout + (size_t)i * S;
        r[4] = sigmoidf_(r[4]); /* :214 */
        float sum = 0.0f;
        for (int c = 0; c < NC; ++c) {
            float v = r[5 + c] - gmax;
            if (rescale) v = v / gmin * -100.0f;
            v = expf(v);
            r[5 + c] = v;
            sum += v;
        }
        for (int c = 0; c < NC; ++c) {
            float p = r[4] * (r[5 + c] / sum); /* :215 */
            r[5 + c] = p > obj_thr ? p : 0.0f;   /* :216 */
        }
    }
    /* candidate boxes (:218-236) */
    int *cand = (int *)malloc(sizeof(int) * (size_t)ncell);
    float *bx = (float *)malloc(sizeof(float) * 4 * (size_t)ncell);
    int n = 0;
    for (int row = 0; row < GH; ++row)
        for (int col = 0; col < GW; ++col)
            for (int b = 0; b < NB; ++b) {
                const int i = (row * GW + col) * NB + b;
                const float *r = netout + (size_t)i * S;
                int any = 0;
                for (int c = 0; c < NC; ++c)
                    if (r[5 + c] != 0.0f) { any = 1; break; }
                if (!any) continue;
                bx[n * 4 + 0] = ((float)col + sigmoidf_(r[0])) / (float)GW;
                bx[n * 4 + 1] = ((float)row + sigmoidf_(r[1])) / (float)GH;
                bx[n * 4 + 2] = anchors[2 * b + 0] * expf(r[2]) / (float)GW;
                bx[n * 4 + 3] = anchors[2 * b + 1] * expf(r[3]) / (float)GH;
                cand[n++] = i;
            }
    /* per-class greedy NMS (:239-252) */
    sc_t *ord = (sc_t *)malloc(sizeof(sc_t) * (size_t)(n > 0 ? n : 1));
    /* an IoU never exceeds 1: with nms_thr > 1 the sweep cannot suppress anything (tests use that to list candidates) */
    for (int c = 0; c < NC && nms_thr <= 1.0f; ++c) {
        for (int k = 0; k < n; ++k) {
            ord[k].score = netout[(size_t)cand[k] * S + 5 + c];
            ord[k].idx = k;
        }
        qsort(ord, (size_t)n, sizeof(sc_t), sc_cmp);
        for (int i = 0; i < n; ++i) {
            const int ii = ord[i].idx;
            if (netout[(size_t)cand[ii] * S + 5 + c] == 0.0f) continue;
            for (int j = i + 1; j < n; ++j) {
                const int jj = ord[j].idx;
                if (orc_bbox_iou(bx + ii * 4, bx + jj * 4) >= nms_thr)
                    netout[(size_t)cand[jj] * S + 5 + c] = 0.0f;
            }
        }
    }
    /* final filter (:255): post-NMS argmax label, score > obj_thr */
    int m = 0;
    for (int k = 0; k < n; ++k) {
        const float *r = netout + (size_t)cand[k] * S;
        int lab = 0;
        for (int c = 1; c < NC; ++c)
            if (r[5 + c] > r[5 + lab]) lab = c;
        const float sc = r[5 + lab];
        if (sc > obj_thr) {
            if (m < cap) {
                float *o = out_box + (size_t)m * 8;
                o[0] = bx[k * 4 + 0];
                o[1] = bx[k * 4 + 1];
                o[2] = bx[k * 4 + 2];
                o[3] = bx[k * 4 + 3];
                o[4] = r[4];
                o[5] = (float)lab;
                o[6] = sc;
                o[7] = (float)cand[k];
            }
            ++m;
        }
    }
    free(ord);
    free(bx);
    free(cand);
    return m;
}

/* ---------------------------------------------------------------------------
 * Track-ID assignment.  BUILD-DEFINED: the reference has no association step
 * and never reads `trackid`; MultiObjDetTracker.predict
 * (MultiObjDetTracker.py:295-315) only decodes each frame.  Specification
 * (DESIGN.md "Track identity"): frames of one clip are visited in order; the
 * boxes of frame t are visited in decode order; box i takes the id of the
 * not-yet-claimed frame-(t-1) box j of the SAME label with the largest
 * bbox_iou(i,j) provided that IoU >= assoc_thr (ties -> lowest j); otherwise
 * it opens a new id (ids count up from 0 per clip).
 *   boxes  [T, cap, 8]  rows as written by orc_decode_netout
 *   counts [T]
 *   ids    [T, cap]     output; -1 in unused slots
 * Returns the number of ids opened. */
ORC_API int orc_associate_clip(const float *boxes, const int *counts, int T,
                               int cap, float assoc_thr, int *ids)
{
    int next_id = 0;
    char *claimed = (char *)malloc((size_t)(cap > 0 ? cap : 1));
    for (int t = 0; t < T; ++t) {
        const int n = counts[t] < cap ? counts[t] : cap;
        for (int i = 0; i < cap; ++i) ids[t * cap + i] = -1;
        const int np_ = t > 0 ? (counts[t - 1] < cap ? counts[t - 1] : cap) : 0;
        memset(claimed, 0, (size_t)(cap > 0 ? cap : 1));
        for (int i = 0; i < n; ++i) {
            const float *bi = boxes + ((size_t)t * cap + i) * 8;
            int best = -1;
            float best_iou = -1.0f;
            for (int j = 0; j < np_; ++j) {
                if (claimed[j]) continue;
                const float *bj = boxes + ((size_t)(t - 1) * cap + j) * 8;
                if (bj[5] != bi[5]) continue;
                const float iou = orc_bbox_iou(bi, bj);
                if (iou >= assoc_thr && iou > best_iou) {
                    best_iou = iou;
                    best = j;
                }
            }
            if (best >= 0) {
                claimed[best] = 1;
                ids[t * cap + i] = ids[(t - 1) * cap + best];
            } else {
                ids[t * cap + i] = next_id++;
            }
        }
    }
    free(claimed);
    return next_id;
}

/* ---------------------------------------------------------------------------
 * TinyHeatmapTracker helpers (utility/utils.py:53-79).
 * ------------------------------------------------------------------------- */
static void py_slice(int start, int stop, int n, int *lo, int *hi)
{
    /* numpy basic-slice normalisation of a[start:stop] on an axis of length n */
    if (start < 0) { start += n; if (start < 0) start = 0; } else if (start > n) start = n;
    if (stop < 0) { stop += n; if (stop < 0) stop = 0; } else if (stop > n) stop = n;
    *lo = start; *hi = stop;
}

/* generate_heatmap_feat(det_x, det_y, det_w, det_h, hmap_size) (utils.py:53-58) called as
 * the data generator does, preprocessing.py:455: (cx - w/2.0, cy - h/2.0, w, h).  float64
 * arithmetic, int() truncation, numpy slice assignment.  box4 [n,4] -> out [n, hs*hs]. */
ORC_API void orc_heatmap_from_boxes(const float *box4, int n, int hs, float *out)
{
    for (int b = 0; b < n; ++b) {
        const double cx = box4[b * 4 + 0], cy = box4[b * 4 + 1], w = box4[b * 4 + 2], h = box4[b * 4 + 3];
        const int sx = (int)((cx - w / 2.0) * hs), sy = (int)((cy - h / 2.0) * hs);
        const int sh = (int)(h * hs), sw = (int)(w * hs);
        int y0, y1, x0, x1;
        py_slice(sy, sy + sh + 1, hs, &y0, &y1);
        py_slice(sx, sx + sw + 1, hs, &x0, &x1);
        float *o = out + (size_t)b * hs * hs;
        for (int y = 0; y < hs; ++y)
            for (int x = 0; x < hs; ++x) o[y * hs + x] = (y >= y0 && y < y1 && x >= x0 && x < x1) ? 1.0f : 0.0f;
    }
}

/* generate_rectangle_from_heatmap (utils.py:61-79): heat [n, hs*hs] -> rect [n,4] = x1,y1,x2,y2 */
ORC_API void orc_rect_from_heatmap(const float *heat, int n, int hs, float thresh, int *rect)
{
    for (int b = 0; b < n; ++b) {
        int x1 = hs, y1 = hs, y2 = -1, x2 = -1;
        const float *hm = heat + (size_t)b * hs * hs;
        for (int i = 0; i < hs; ++i)
            for (int j = 0; j < hs; ++j)
                if (hm[i * hs + j] >= thresh) {
                    if (i < y1) y1 = i;
                    if (i > y2) y2 = i;
                    if (j < x1) x1 = j;
                    if (j > x2) x2 = j;
                }
        rect[b * 4 + 0] = x1; rect[b * 4 + 1] = y1; rect[b * 4 + 2] = x2; rect[b * 4 + 3] = y2;
    }
}

/* ---------------------------------------------------------------------------
 * Training-target encoding: the coordinate fix at the end of
 * BaseBatchGenerator.aug_image (utility/preprocessing.py:171-188) followed by
 * BatchGenerator.output_from_instance's y / b construction (:214-293).  Python
 * float (= float64) arithmetic, int() truncation toward zero, bbox_iou (utils.py:155-173)
 * in float64 on BoundBox(0,0,w,h) vs the anchors, strict `max_iou < iou` (first best wins),
 * later objects overwrite x,y,w,h,conf in a shared (cell, anchor) slot while class bits
 * accumulate, true_box_index wraps modulo TRUE_BOX_BUFFER.  PINNED by
 * tests/golden/targets.npz (the reference's own statements exec'd by tools/make_goldens.py).
 *   objs   [n, cap, 5] int32  xmin, ymin, xmax, ymax, label index (-1: name not in LABELS)
 *   counts [n]; dims [n,2] = original image (w, h); aug [n,4] = scale, offx, offy, flip or NULL
 *   y [n, GH, GW, NB, 5+C] float64, b [n, TBB, 4] float64 (both fully written)
 * ------------------------------------------------------------------------- */
static double interval_overlap_d(double x1, double x2, double x3, double x4)
{
    if (x3 < x1) {
        if (x4 < x1) return 0.0;
        return (x2 < x4 ? x2 : x4) - x1;
    } else {
        if (x2 < x3) return 0.0;
        return (x2 < x4 ? x2 : x4) - x3;
    }
}

static double bbox_iou_d(double x1, double y1, double w1, double h1, double x2, double y2, double w2, double h2)
{
    const double iw = interval_overlap_d(x1 - w1 / 2, x1 + w1 / 2, x2 - w2 / 2, x2 + w2 / 2);
    const double ih = interval_overlap_d(y1 - h1 / 2, y1 + h1 / 2, y2 - h2 / 2, y2 + h2 / 2);
    const double inter = iw * ih;
    return inter / (w1 * h1 + w2 * h2 - inter);
}

static int fix_coord(int v, int use_aug, double scale, int off, int image, int orig)
{
    if (use_aug) v = (int)(v * scale - off);              /* preprocessing.py:174,180 */
    v = (int)(v * (double)image / orig);                  /* :176,182 */
    v = v < image ? v : image;                            /* :177,183  max(min(v, IMAGE), 0) */
    return v > 0 ? v : 0;
}

ORC_API void orc_encode_targets(const int *objs, const int *counts, const int *dims, const double *aug, int n,
                                int cap, int GH, int GW, int NB, int C, int IH, int IW, int TBB,
                                const double *anchors, double *y, double *b)
{
    const int S = 5 + C;
    for (int f = 0; f < n; ++f) {
        double *yf = y + (size_t)f * GH * GW * NB * S;
        double *bf = b + (size_t)f * TBB * 4;
        memset(yf, 0, sizeof(double) * (size_t)GH * GW * NB * S);
        memset(bf, 0, sizeof(double) * (size_t)TBB * 4);
        const int w = dims[f * 2], h = dims[f * 2 + 1];
        const int use_aug = aug != NULL;
        const double scale = use_aug ? aug[f * 4] : 1.0;
        const int offx = use_aug ? (int)aug[f * 4 + 1] : 0, offy = use_aug ? (int)aug[f * 4 + 2] : 0;
        const int flip = use_aug && aug[f * 4 + 3] > 0.5;
        int tbi = 0;
        for (int k = 0; k < counts[f]; ++k) {
            const int *o = objs + ((size_t)f * cap + k) * 5;
            int xmin = fix_coord(o[0], use_aug, scale, offx, IW, w), xmax = fix_coord(o[2], use_aug, scale, offx, IW, w);
            const int ymin = fix_coord(o[1], use_aug, scale, offy, IH, h), ymax = fix_coord(o[3], use_aug, scale, offy, IH, h);
            if (flip) { const int t = xmin; xmin = IW - xmax; xmax = IW - t; }       /* :185-188 */
            if (!(xmax > xmin && ymax > ymin && o[4] >= 0)) continue;                  /* :224 */
            const double cx = (.5 * (xmin + xmax)) / ((double)IW / GW);
            const double cy = (.5 * (ymin + ymax)) / ((double)IH / GH);
            const int gx = (int)floor(cx), gy = (int)floor(cy);
            if (!(gx < GW && gy < GH)) continue;                                        /* :233 */
            const double cw = (xmax - xmin) / ((double)IW / GW), ch = (ymax - ymin) / ((double)IH / GH);
            int best = -1;
            double best_iou = -1;
            for (int a = 0; a < NB; ++a) {                                              /* :246-252 */
                const double iou = bbox_iou_d(0, 0, cw, ch, 0, 0, anchors[2 * a], anchors[2 * a + 1]);
                if (best_iou < iou) { best = a; best_iou = iou; }
            }
            if (best < 0) best += NB;                                                   /* python index -1 */
            double *cell = yf + (((size_t)gy * GW + gx) * NB + best) * S;
            cell[0] = cx; cell[1] = cy; cell[2] = cw; cell[3] = ch; cell[4] = 1.0;      /* :255-256 */
            cell[5 + o[4]] = 1.0;                                                       /* :257 */
            double *tb = bf + (size_t)tbi * 4;                                          /* :260 */
            tb[0] = cx; tb[1] = cy; tb[2] = cw; tb[3] = ch;
            tbi = (tbi + 1) % TBB;                                                      /* :262-263 */
        }
    }
}

/* ---------------------------------------------------------------------------
 * Frame ingest: cv2.resize(image, (IMAGE_H, IMAGE_W)) on uint8 frames
 * (models_detection/KerasYOLO.py:526; interpolation defaults to INTER_LINEAR).
 * PARITY UNPINNED versus OpenCV (cv2 is not part of this image): this is
 * the build's definition -- OpenCV's published 8-bit bilinear scheme: half-pixel centres,
 * no anti-aliasing, coefficients quantised to 11 bits (round-half-even), horizontal pass
 * into 32-bit integers, vertical pass ((b0*(S0>>4))>>16 + (b1*(S1>>4))>>16 + 2) >> 2.
 * Integer arithmetic only, so the HIP kernel can be (and is) bit-exact against it.
 * ------------------------------------------------------------------------- */
ORC_API void orc_resize_tables(int src, int dst, int *i0, int *i1, int *c0, int *c1)
{
    const double scale = (double)src / (double)dst;
    for (int d = 0; d < dst; ++d) {
        float f = (float)(((double)d + 0.5) * scale - 0.5);
        int s = (int)floorf(f);
        f -= (float)s;
        if (s < 0) { f = 0.0f; s = 0; }
        if (s >= src - 1) { f = 0.0f; s = src - 1; }
        i0[d] = s;
        i1[d] = s + 1 < src ? s + 1 : src - 1;
        c0[d] = (int)lrintf((1.0f - f) * 2048.0f);
        c1[d] = (int)lrintf(f * 2048.0f);
    }
}

ORC_API void orc_resize_bilinear_u8(const uint8_t *src, int n, int Hs, int Ws, uint8_t *dst, int Hd, int Wd)
{
    int *x0 = (int *)malloc(sizeof(int) * 4 * (size_t)Wd), *x1 = x0 + Wd, *a0 = x1 + Wd, *a1 = a0 + Wd;
    int *y0 = (int *)malloc(sizeof(int) * 4 * (size_t)Hd), *y1 = y0 + Hd, *b0 = y1 + Hd, *b1 = b0 + Hd;
    orc_resize_tables(Ws, Wd, x0, x1, a0, a1);
    orc_resize_tables(Hs, Hd, y0, y1, b0, b1);
    for (int f = 0; f < n; ++f) {
        const uint8_t *s = src + (size_t)f * Hs * Ws * 3;
        uint8_t *o = dst + (size_t)f * Hd * Wd * 3;
        for (int y = 0; y < Hd; ++y) {
            const uint8_t *r0 = s + (size_t)y0[y] * Ws * 3, *r1 = s + (size_t)y1[y] * Ws * 3;
            for (int x = 0; x < Wd; ++x)
                for (int c = 0; c < 3; ++c) {
                    const int S0 = r0[x0[x] * 3 + c] * a0[x] + r0[x1[x] * 3 + c] * a1[x];
                    const int S1 = r1[x0[x] * 3 + c] * a0[x] + r1[x1[x] * 3 + c] * a1[x];
                    const int v = (((b0[y] * (S0 >> 4)) >> 16) + ((b1[y] * (S1 >> 4)) >> 16) + 2) >> 2;
                    o[((size_t)y * Wd + x) * 3 + c] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
                }
        }
    }
    free(x0);
    free(y0);
}

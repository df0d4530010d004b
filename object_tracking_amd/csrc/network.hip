// network.hip -- context, weight loading and layer sequencing behind the C ABI
// (include/mi355_dt.h).  Graph topology follows the reference:
//   detector  models_detection/KerasYOLO.py:277-405 (weight order :244-274)
//   tracker   models_tracking/MultiObjDetTracker.py:160-189
//   tiny      models_tracking/TinyTracker.py:25-41
#include <cmath>
#include <cstdarg>
#include <cstdlib>
#include <cstring>
#include <thread>

#include "dt_internal.h"

static const float BN_EPS = 1e-3f;   // Keras BatchNormalization default (no epsilon= passed)
static const float LEAKY = 0.1f;     // LeakyReLU(alpha=0.1)
static char g_static_err[512] = "";

// (idx, k, cin, cout, pool_after) -- main trunk, KerasYOLO.py:279-384
static const int TRUNK[20][5] = {
    {1, 3, 3, 32, 1},      {2, 3, 32, 64, 1},     {3, 3, 64, 128, 0},    {4, 1, 128, 64, 0},
    {5, 3, 64, 128, 1},    {6, 3, 128, 256, 0},   {7, 1, 256, 128, 0},   {8, 3, 128, 256, 1},
    {9, 3, 256, 512, 0},   {10, 1, 512, 256, 0},  {11, 3, 256, 512, 0},  {12, 1, 512, 256, 0},
    {13, 3, 256, 512, 1},  {14, 3, 512, 1024, 0}, {15, 1, 1024, 512, 0}, {16, 3, 512, 1024, 0},
    {17, 1, 1024, 512, 0}, {18, 3, 512, 1024, 0}, {19, 3, 1024, 1024, 0}, {20, 3, 1024, 1024, 0}};

static inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

int dt_fail(dt_ctx *ctx, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf;
    else snprintf(g_static_err, sizeof(g_static_err), "%s", buf);
    return code;
}

static void graphs_clear(dt_ctx *ctx)
{
    if (ctx->graphs.empty() && ctx->graph_seen.empty()) return;
    (void)hipStreamSynchronize(ctx->stream);      // replays run on the caller's stream
    for (auto &kv : ctx->graphs) (void)hipGraphExecDestroy(kv.second);
    ctx->graphs.clear();
    ctx->graph_seen.clear();
    ctx->graph_tags.clear();
}

// Runs `body` (a sequence of launches on ctx->stream that touches library-owned buffers only), as a replayed
// hipGraph when graphs are on: first sighting of `key` runs plainly (allocates workspaces, one-time kernel
// attribute calls), the second captures on the internal stream and instantiates, later ones replay.
//   in_tensor: the tensor the sequence reads from outside (or null).  A replay runs no host code, so what the host knows about a tensor's
//   max |x| when the graph is captured must hold at every replay: the tag of `in_tensor` -- "its producer, which ran just before in this
//   call, published into slot s" -- is kept and becomes part of the key (another precondition, another graph); every other tag is dropped,
//   i.e. the sequence measures whatever else it reads itself.  What the sequence leaves behind is re-applied on replay.
static int graphed(dt_ctx *ctx, const std::string &key_in, const std::function<int()> &body, const float *in_tensor = nullptr)
{
    if (!ctx->graph_on || ctx->prof || ctx->capturing) return body();
    hipStream_t user = ctx->stream;
    std::string key = key_in;
    {
        std::vector<dt_ctx::AmaxTag> keep;
        for (const auto &t : ctx->amax_tag)
            if (in_tensor && t.lo == in_tensor) { keep.push_back(t); key += ":am" + std::to_string(t.slot) + "c" + std::to_string(t.cols); }
        ctx->amax_tag.swap(keep);
    }
    auto it = ctx->graphs.find(key);
    if (it == ctx->graphs.end()) {
        int &seen = ctx->graph_seen[key];
        if (seen < 0 || seen++ == 0) { const int rc0 = body(); ctx->graph_tags[key] = ctx->amax_tag; return rc0; }
        hipGraph_t g = nullptr;
        if (hipStreamBeginCapture(ctx->gstream, hipStreamCaptureModeThreadLocal) != hipSuccess) { seen = -1; return body(); }
        ctx->capturing = true; ctx->stream = ctx->gstream;
        const int rc = body();
        ctx->graph_tags[key] = ctx->amax_tag;
        ctx->stream = user; ctx->capturing = false;
        const hipError_t e = hipStreamEndCapture(ctx->gstream, &g);
        hipGraphExec_t ex = nullptr;
        if (rc != DT_OK || e != hipSuccess || !g || hipGraphInstantiate(&ex, g, nullptr, nullptr, 0) != hipSuccess) {
            if (g) (void)hipGraphDestroy(g);
            (void)hipGetLastError();
            ctx->graph_seen[key] = -1;      // do not try again for this shape
            return rc != DT_OK ? rc : body();
        }
        (void)hipGraphDestroy(g);
        ++ctx->graph_captures;
        it = ctx->graphs.emplace(key, ex).first;
    }
    // the instantiated graph is launched on the CALLER's stream (only the capture needs the internal one): stream order is the
    // synchronisation -- no event pair per replay (round 3 replayed on the internal stream between two events and was 0.06 ms
    // SLOWER than plain launches at batch 8)
    HIP_TRY(ctx, hipGraphLaunch(it->second, user));
    ctx->amax_tag = ctx->graph_tags[key];
    ++ctx->graph_replays;
    return DT_OK;
}

extern "C" int dt_graph_enable(dt_ctx *ctx, int on)
{
    if (!ctx) return DT_ERR_ARG;
    if (on && !ctx->gstream) {
        HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->gstream, hipStreamNonBlocking));
        HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->gev_in, hipEventDisableTiming));
        HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->gev_out, hipEventDisableTiming));
    }
    if (!on) graphs_clear(ctx);
    ctx->graph_on = on != 0;
    return DT_OK;
}

float *ws_get(dt_ctx *ctx, const char *name, size_t bytes, bool zero_on_grow)
{
    DevBuf &b = ctx->ws[name];
    if (b.bytes < bytes) {
        if (ctx->capturing) {   // cannot allocate inside a stream capture (graphed() runs every shape once uncaptured first)
            dt_fail(ctx, DT_ERR_STATE, "workspace %s would grow during graph capture", name);
            return nullptr;
        }
        graphs_clear(ctx);      // captured kernels hold the old pointer
        if (b.p) {
            (void)hipStreamSynchronize(ctx->stream);
            (void)hipFree(b.p);
            b.p = nullptr;
            b.bytes = 0;
        }
        if (hipMalloc(&b.p, bytes) != hipSuccess) {
            b.p = nullptr;
            dt_fail(ctx, DT_ERR_DEVICE, "hipMalloc(%zu) failed for workspace %s", bytes, name);
            return nullptr;
        }
        b.bytes = bytes;
        if (zero_on_grow) (void)hipMemsetAsync(b.p, 0, bytes, ctx->stream);
    }
    return static_cast<float *>(b.p);
}

ProfScope::ProfScope(dt_ctx *c, const char *name, double flops, double bytes, const char *tag)
    : ctx(c), on(c->prof && !c->capturing)
{
    if (!on) return;
    ev.name = name;
    if (tag) ev.tag = std::string(name) + ":" + tag;
    (void)hipEventCreate(&ev.a);
    (void)hipEventCreate(&ev.b);
    ProfEntry &e = ctx->prof_tab[name];
    e.launches += 1;
    e.flops += flops;
    e.bytes += bytes;
    if (tag) {
        ProfEntry &t = ctx->prof_tab[ev.tag];
        t.launches += 1;
        t.flops += flops;
        t.bytes += bytes;
    }
    (void)hipEventRecord(ev.a, ctx->stream);
}
ProfScope::~ProfScope()
{
    if (!on) return;
    (void)hipEventRecord(ev.b, ctx->stream);
    ctx->pending.push_back(ev);
}

// forget (and free) the split-bf16 twin of a Winograd weight buffer that is about to be freed
static void s3_drop(dt_ctx *ctx, const void *wino)
{
    auto it = ctx->wino_s3.find(wino);
    if (wino && it != ctx->wino_s3.end()) { (void)hipFree(it->second); ctx->wino_s3.erase(it); }
    auto ih = ctx->wino_h2.find(wino);
    if (wino && ih != ctx->wino_h2.end()) { (void)hipFree(ih->second.terms); (void)hipFree(ih->second.pscale); ctx->wino_h2.erase(ih); }
}

// ---- max-|x| slots of the fp16 form (dt_internal.h: dt_ctx::amax) ------------------------------------------------------------------
#define DT_AMAX_SLOTS 128
enum { AMAX_ONE = 0, AMAX_IN = 32, AMAX_TRK = 56, AMAX_TEST = 57, AMAX_PACK = 64 };
static unsigned *amax_slot(dt_ctx *ctx, int slot) { return ctx->amax ? ctx->amax + (size_t)slot * DT_AMAX_WORDS : nullptr; }
// every API entry that runs layers starts here: what a previous call knew about a tensor's maximum says nothing about the bytes behind the pointer now
static void amax_reset(dt_ctx *ctx) { ctx->amax_tag.clear(); ctx->h2_small = false; }
// the slot that holds max |x| of the rows x cols tensor at x: the one its producer filled (tagged), else measured here into `slot`
// a layer is about to write `floats` floats from `lo` on: what was known about tensors in that range is void
static void amax_forget(dt_ctx *ctx, const float *lo, long long floats)
{
    if (!lo || ctx->amax_tag.empty()) return;
    const float *hi = lo + floats;
    auto &v = ctx->amax_tag;
    for (size_t i = v.size(); i-- > 0;)
        if (v[i].lo < hi && lo < v[i].hi) v.erase(v.begin() + (long)i);
}
static void amax_note(dt_ctx *ctx, const float *lo, long long floats, int cols, int slot)
{
    amax_forget(ctx, lo, floats);
    ctx->amax_tag.push_back(dt_ctx::AmaxTag{lo, lo + floats, cols, slot});
}
// the producers' slots of one detector forward start from zero (the epilogues only ever raise them)
static int amax_begin(dt_ctx *ctx)
{
    if (!ctx->amax) return DT_OK;
    HIP_TRY(ctx, hipMemsetAsync(amax_slot(ctx, 1), 0, (size_t)(AMAX_IN - 1) * DT_AMAX_WORDS * sizeof(unsigned), ctx->stream));
    return DT_OK;
}
// slot a layer's epilogue fills with the max |x| of what it writes: conv_1 .. conv_23 only (the test entry points run "layer 0")
// blocks of 16 x 16 pixels from which a narrow 3x3 layer takes the direct fp16-form kernel (conv3_h2.hip) instead of the fused fp32 one (Policy::c3h2_blocks2 for
// conv_2, Cin = 32; 1024 for conv_3 / conv_5)
static long long c3h2_min_blocks(const dt_ctx *ctx, int cin) { return cin <= 32 ? ctx->pol.c3h2_blocks2 : 1024; }
static int amax_out_slot(const ConvLayer &L) { return L.idx >= 1 && L.idx <= 23 ? L.idx : 0; }
static const unsigned *ensure_amax(dt_ctx *ctx, const float *x, long long rows, int cols, long long ld, int slot)
{
    for (const auto &t : ctx->amax_tag)
        if (t.lo == x && t.hi == x + rows * ld && t.cols == cols) return amax_slot(ctx, t.slot);
    unsigned *s = amax_slot(ctx, slot);
    if (!s) { dt_fail(ctx, DT_ERR_STATE, "max-|x| slots not allocated"); return nullptr; }
    ProfScope ps(ctx, "absmax", 0.0, 4.0 * (double)rows * cols);
    if (launch_absmax(ctx->stream, x, rows, cols, ld, 1, 0, s)) { dt_fail(ctx, DT_ERR_DEVICE, "absmax launch failed"); return nullptr; }
    for (const auto &t : ctx->amax_tag)      // (a slot names ONE tensor)
        if (t.slot == slot) { amax_forget(ctx, t.lo, t.hi - t.lo); break; }
    amax_note(ctx, x, rows * ld, cols, slot);
    return s;
}
// does this launch take the fp16 form of the split GEMM?  (DT_PIN keeps the bf16 form: see Policy::s3_h2)
static bool h2_wanted(const dt_ctx *ctx) { return ctx->pol.s3 != 0 && ctx->pol.s3_h2 != 0 && !ctx->pol.pin && !ctx->h2_small; }

static int upload(dt_ctx *ctx, float **dst, const std::vector<float> &h)
{
    if (*dst) { (void)hipFree(*dst); *dst = nullptr; }
    HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(dst), h.size() * sizeof(float)));
    HIP_TRY(ctx, hipMemcpy(*dst, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
    return DT_OK;
}

// ---------------------------------------------------------------------------
extern "C" int dt_abi_version(void) { return 107; }   // 1.07: + dt_gemm_split (test entry point of wino_gemm_s3.hip: bf16 x 3 or fp16 x 2 terms), dt_policy_set

extern "C" int dt_create(dt_ctx **out)
{
    if (!out) return dt_fail(nullptr, DT_ERR_ARG, "dt_create: null out");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return dt_fail(nullptr, DT_ERR_DEVICE,
                       "dt_create: no HIP device visible -- libmi355_dt has no CPU fallback");
    hipDeviceProp_t prop;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess)
        return dt_fail(nullptr, DT_ERR_DEVICE, "dt_create: hipGetDeviceProperties failed");
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return dt_fail(nullptr, DT_ERR_DEVICE, "dt_create: device is %s; this library is built for gfx950 only",
                       prop.gcnArchName);
    dt_ctx *c = new dt_ctx();
    c->device_ok = 1;
    policy_from_env(c->pol);
    std::vector<float> lut(256);
    for (int i = 0; i < 256; ++i) lut[i] = (float)((double)i / 255.0);   // utils.py:150-153
    if (upload(c, &c->lut255, lut) != DT_OK) {
        snprintf(g_static_err, sizeof(g_static_err), "%s", c->err.c_str());
        delete c;
        return DT_ERR_DEVICE;
    }
    {   // max-|x| slots; slot 0 = 1.0
        std::vector<unsigned> am((size_t)DT_AMAX_SLOTS * DT_AMAX_WORDS, 0u);
        for (int q = 0; q < DT_AMAX_SUB; ++q) am[(size_t)q * DT_AMAX_LINE] = 0x3f800000u;
        if (hipMalloc(reinterpret_cast<void **>(&c->amax), am.size() * sizeof(unsigned)) != hipSuccess ||
            hipMemcpy(c->amax, am.data(), am.size() * sizeof(unsigned), hipMemcpyHostToDevice) != hipSuccess) {
            snprintf(g_static_err, sizeof(g_static_err), "dt_create: max-|x| slot allocation failed");
            delete c;
            return DT_ERR_DEVICE;
        }
    }
    *out = c;
    return DT_OK;
}

extern "C" void dt_destroy(dt_ctx *ctx)
{
    if (!ctx) return;
    (void)hipStreamSynchronize(ctx->stream);
    for (auto &kv : ctx->ws)
        if (kv.second.p) (void)hipFree(kv.second.p);
    for (int i = 0; i <= 23; ++i) {
        if (ctx->layers[i].wt) (void)hipFree(ctx->layers[i].wt);
        if (ctx->layers[i].bias) (void)hipFree(ctx->layers[i].bias);
        s3_drop(ctx, ctx->layers[i].wino);
        if (ctx->layers[i].wt_s3) (void)hipFree(ctx->layers[i].wt_s3);
        if (ctx->layers[i].bias_s3) (void)hipFree(ctx->layers[i].bias_s3);
        if (ctx->layers[i].wt_h2) (void)hipFree(ctx->layers[i].wt_h2);
        if (ctx->layers[i].pscale_h2) (void)hipFree(ctx->layers[i].pscale_h2);
        if (ctx->layers[i].w3_h2) (void)hipFree(ctx->layers[i].w3_h2);
        if (ctx->layers[i].pscale_w3) (void)hipFree(ctx->layers[i].pscale_w3);
        if (ctx->layers[i].wino) (void)hipFree(ctx->layers[i].wino);
        if (ctx->layers[i].wino_alt) (void)hipFree(ctx->layers[i].wino_alt);
        if (ctx->layers[i].fused4s) (void)hipFree(ctx->layers[i].fused4s);
        if (ctx->layers[i].scale) (void)hipFree(ctx->layers[i].scale);
    }
    if (ctx->trk_wxm_wino) { s3_drop(ctx, ctx->trk_wxm_wino); (void)hipFree(ctx->trk_wxm_wino); }
    if (ctx->trk_bx16) (void)hipFree(ctx->trk_bx16);
    float *singles[] = {ctx->conv1_w, ctx->conv1_b, ctx->lut255, ctx->anchors_dev, ctx->trk_wx, ctx->trk_bx,
                        ctx->trk_wh,  ctx->trk_wo,  ctx->trk_bo, ctx->trk_wx_wino, ctx->trk_wh_wino, ctx->tiny_wx,     ctx->tiny_bx, ctx->tiny_ur,
                        ctx->tiny_wd, ctx->tiny_bd};
    s3_drop(ctx, ctx->trk_wx_wino);
    s3_drop(ctx, ctx->trk_wh_wino);
    if (ctx->s3_ones) (void)hipFree(ctx->s3_ones);
    if (ctx->amax) (void)hipFree(ctx->amax);
    if (ctx->conv1_w3) (void)hipFree(ctx->conv1_w3);
    if (ctx->conv1_w3u8) (void)hipFree(ctx->conv1_w3u8);
    for (float *p : singles)
        if (p) (void)hipFree(p);
    for (auto &e : ctx->pending) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    if (ctx->gstream) {
        graphs_clear(ctx);
        (void)hipEventDestroy(ctx->gev_in); (void)hipEventDestroy(ctx->gev_out);
        (void)hipStreamDestroy(ctx->gstream);
    }
    delete ctx;
}

extern "C" const char *dt_last_error(dt_ctx *ctx) { return ctx ? ctx->err.c_str() : g_static_err; }

extern "C" int dt_set_stream(dt_ctx *ctx, void *hip_stream)
{
    if (!ctx) return DT_ERR_ARG;
    hipStream_t next = static_cast<hipStream_t>(hip_stream);
    if (next != ctx->stream) {
        // work already queued on the old stream may still use the library's workspaces, and a replayed graph may be in flight there:
        // order the new stream behind it (an event, no host wait), and drop the captured graphs -- graphs_clear() and the workspace
        // grow path synchronise ctx->stream only, which from now on is the new one
        graphs_clear(ctx);                              // (synchronises the OLD stream if any graph exists)
        hipEvent_t ev = nullptr;
        if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) == hipSuccess) {
            if (hipEventRecord(ev, ctx->stream) != hipSuccess || hipStreamWaitEvent(next, ev, 0) != hipSuccess) {
                (void)hipGetLastError();
                (void)hipStreamSynchronize(ctx->stream);      // (e.g. the caller already destroyed the old stream: nothing of it can be in flight then)
            }
            (void)hipEventDestroy(ev);
        } else {
            (void)hipStreamSynchronize(ctx->stream);
        }
        ctx->stream = next;
    }
    return DT_OK;
}

// ---------------------------------------------------------------------------
// detector
// ---------------------------------------------------------------------------
extern "C" int dt_detector_config(dt_ctx *ctx, int image_h, int image_w, int nb_box, int nb_class,
                                  const float *h_anchors)
{
    if (!ctx) return DT_ERR_ARG;
    if (image_h <= 0 || image_w <= 0 || image_h % 32 || image_w % 32)
        return dt_fail(ctx, DT_ERR_ARG, "image size %dx%d must be a positive multiple of 32", image_h, image_w);
    if (nb_box <= 0 || nb_box > 32 || nb_class <= 0 || !h_anchors)
        return dt_fail(ctx, DT_ERR_ARG, "bad nb_box/nb_class/anchors");
    ctx->image_h = image_h; ctx->image_w = image_w;
    ctx->nb_box = nb_box; ctx->nb_class = nb_class;
    ctx->cb = nb_box * (5 + nb_class);
    std::vector<float> a(h_anchors, h_anchors + 2 * nb_box);
    memcpy(ctx->anchors, h_anchors, sizeof(float) * 2 * nb_box);
    ctx->det_loaded = false;
    return upload(ctx, &ctx->anchors_dev, a);
}

// kernel in the darknet file is (O,I,H,W) (KerasYOLO.py:267-268 reshapes the
// reversed Keras shape and transposes [2,3,1,0]) -> HWIO
static void oihw_to_hwio(const float *src, int O, int I, int k, std::vector<float> &dst)
{
    dst.resize((size_t)O * I * k * k);
    for (int o = 0; o < O; ++o)
        for (int i = 0; i < I; ++i)
            for (int y = 0; y < k; ++y)
                for (int x = 0; x < k; ++x)
                    dst[(((size_t)y * k + x) * I + i) * O + o] = src[(((size_t)o * I + i) * k + y) * k + x];
}

static bool wino_wanted(const dt_ctx *ctx, int ks, int cin, int cout);
static int wino_tile(const dt_ctx *ctx, bool recurrent);
static int upload_wino(dt_ctx *ctx, float **dst, int ts, const float *hwio, int cin_src, int cout_src, const int *cin_map,
                       int cin_dst, const int *n_map, int npad, const float *scale, bool want_s3);

static int load_conv_layer(dt_ctx *ctx, int idx, int ks, int cin, int cout, const float *hwio, const float *scale,
                           const float *bias_src)
{
    ConvLayer &L = ctx->layers[idx];
    L.idx = idx; L.ks = ks; L.cin = cin; L.cout = cout;
    L.npad = round_up(cout, 256);   // weight rows padded to the widest column tile (256)
    std::vector<float> packed((size_t)L.npad * ks * ks * cin);
    pack_conv_weights(hwio, ks, cin, cout, nullptr, cin, nullptr, L.npad, scale, packed.data());
    std::vector<float> bias(L.npad, 0.0f);
    for (int c = 0; c < cout; ++c) bias[c] = bias_src[c];
    int rc = upload(ctx, &L.wt, packed);
    if (rc) return rc;
    if (L.wt_s3) { (void)hipStreamSynchronize(ctx->stream); (void)hipFree(L.wt_s3); L.wt_s3 = nullptr; }
    if (L.bias_s3) { (void)hipStreamSynchronize(ctx->stream); (void)hipFree(L.bias_s3); L.bias_s3 = nullptr; }
    if (ks == 1 && ctx->pol.s3 != 0 && ctx->pol.s3_1x1 != 0 && cin % 32 == 0 && L.npad % 128 == 0 && cout >= 64) {
        // a 1x1 layer's weights are a plain [npad][cin] matrix: also as the split-bf16 B operand of wino_gemm_s3.hip
        HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&L.wt_s3), packed.size() * 3 * sizeof(unsigned short)));
        if (launch_wino_s3_pack(ctx->stream, L.wt, 1, L.npad, cin, L.wt_s3)) return dt_fail(ctx, DT_ERR_DEVICE, "split-bf16 weight pack launch failed");
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        // the bias as one extra K stage (wino_gemm_s3.hip): B rows (bias[n], 0, .., 0), A rows (1, 0, .., 0)
        std::vector<unsigned short> bs((size_t)3 * L.npad * 16, 0);
        for (int n = 0; n < cout; ++n) {
            unsigned short t[3];
            wino_s3_split_host(bias_src[n], t);
            for (int t3 = 0; t3 < 3; ++t3) bs[((size_t)t3 * L.npad + n) * 16] = t[t3];
        }
        HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&L.bias_s3), bs.size() * sizeof(unsigned short)));
        HIP_TRY(ctx, hipMemcpy(L.bias_s3, bs.data(), bs.size() * sizeof(unsigned short), hipMemcpyHostToDevice));
        if (!ctx->s3_ones) {
            std::vector<float> ones((size_t)256 * 16, 0.0f);      // the bias stage's A rows, fp32 like the layer's activation: (1, 0, .., 0)
            for (int r = 0; r < 256; ++r) ones[(size_t)r * 16] = 1.0f;
            HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->s3_ones), ones.size() * sizeof(float)));
            HIP_TRY(ctx, hipMemcpy(ctx->s3_ones, ones.data(), ones.size() * sizeof(float), hipMemcpyHostToDevice));
        }
    }
    if (L.wt_h2) { (void)hipStreamSynchronize(ctx->stream); (void)hipFree(L.wt_h2); L.wt_h2 = nullptr; }
    if (L.pscale_h2) { (void)hipStreamSynchronize(ctx->stream); (void)hipFree(L.pscale_h2); L.pscale_h2 = nullptr; }
    if (L.wt_s3 && ctx->pol.s3_h2 != 0) {
        // ... and in the fp16 form: two terms of wt * 2^s (s from max |wt|), the epilogue factor 2^-s; the bias stays fp32 (L.bias)
        HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&L.wt_h2), packed.size() * 2 * sizeof(unsigned short)));
        HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&L.pscale_h2), sizeof(float)));
        if (launch_wino_h2_pack(ctx->stream, L.wt, 1, L.npad, cin, 0, amax_slot(ctx, AMAX_PACK), L.wt_h2, L.pscale_h2))
            return dt_fail(ctx, DT_ERR_DEVICE, "fp16-form weight pack launch failed");
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    if (L.w3_h2) { (void)hipStreamSynchronize(ctx->stream); (void)hipFree(L.w3_h2); L.w3_h2 = nullptr; }
    if (L.pscale_w3) { (void)hipStreamSynchronize(ctx->stream); (void)hipFree(L.pscale_w3); L.pscale_w3 = nullptr; }
    if (ks == 3 && (cin == 32 || cin == 64) && cout % 64 == 0 && cout <= 128 && ctx->pol.s3 != 0 && ctx->pol.s3_h2 != 0 && ctx->pol.c3h2 != 0) {
        // conv_2 / conv_3 / conv_5's shapes: the packed direct-form weights as two fp16 terms for conv3_h2.hip (the same k order)
        HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&L.w3_h2), packed.size() * 2 * sizeof(unsigned short)));
        HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&L.pscale_w3), sizeof(float)));
        if (launch_wino_h2_pack(ctx->stream, L.wt, 1, L.npad, 9 * cin, 0, amax_slot(ctx, AMAX_PACK), L.w3_h2, L.pscale_w3))
            return dt_fail(ctx, DT_ERR_DEVICE, "fp16-form weight pack launch failed");
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    if (L.scale) { (void)hipStreamSynchronize(ctx->stream); (void)hipFree(L.scale); L.scale = nullptr; }
    L.scale_has_zero = false;
    if (scale) {
        std::vector<float> sc(scale, scale + cout);
        for (float v : sc) L.scale_has_zero |= (v == 0.0f);
        if ((rc = upload(ctx, &L.scale, sc))) return rc;
    }
    if (L.wino) { (void)hipStreamSynchronize(ctx->stream); s3_drop(ctx, L.wino); (void)hipFree(L.wino); L.wino = nullptr; }
    if (L.wino_alt) { (void)hipStreamSynchronize(ctx->stream); (void)hipFree(L.wino_alt); L.wino_alt = nullptr; }
    if (L.fused4s) { (void)hipStreamSynchronize(ctx->stream); (void)hipFree(L.fused4s); L.fused4s = nullptr; }
    const bool f4_shape = ((cin == 64 || cin == 128) && cout % 128 == 0 && cout <= 256) || (cin == 32 && cout == 64);
    if (ks == 3 && f4_shape && ctx->pol.wino != 0 && ctx->pol.fused4 != 0) {
        // conv_2 / conv_3 / conv_5 / conv_6 / conv_8's shapes: the fused F(4x4,3x3) kernel (wino4s_fused.hip)
        std::vector<float> u36((size_t)36 * L.npad * cin), uf((size_t)36 * cin * cout);
        wino_pack_weights(4, hwio, cin, cout, nullptr, cin, nullptr, L.npad, scale, u36.data());
        wino4s_fused_pack(u36.data(), L.npad, cin, cout, uf.data());
        if ((rc = upload(ctx, &L.fused4s, uf))) return rc;
    }
    if (wino_wanted(ctx, ks, cin, cout)) {
        L.wino_ts = wino_tile(ctx, false);
        rc = upload_wino(ctx, &L.wino, L.wino_ts, hwio, cin, cout, nullptr, cin, nullptr, L.npad, scale, L.wino_ts == 6);
        if (rc) return rc;
        // small batches of the 13x13 / 26x26 layers: F(4x4) needs 36 GEMMs of ONE (partly filled) row tile where F(6x6)
        // needs 64 -- keep both weight sets and choose per launch (run_conv); only with the default tile policy
        if (L.wino_ts == 6 && ctx->pol.wino_tile == 0 && cin >= 256) {
            rc = upload_wino(ctx, &L.wino_alt, 4, hwio, cin, cout, nullptr, cin, nullptr, L.npad, scale, false);
            if (rc) return rc;
        }
    }
    return upload(ctx, &L.bias, bias);
}

// The ConvLSTM2D input projection with conv_23 folded in (Policy::trk_merge).  z = [x_bbox | conv_feat] with x_bbox = W23 feat + b23 inside
// the image (conv_23 is a 1x1 convolution with a LINEAR activation, KerasYOLO.py:396-400) and zero in the 'same' padding, so
//     Wx * z + b  =  (Wx[feat] + W23 Wx[bbox]) * feat  +  b  +  sum over the taps INSIDE the image of  b23 . Wx[bbox][tap]
// -- one 3x3 convolution of conv_feat alone (K = 1024 instead of 1109 -> 1120) with a bias that depends on which of its nine taps lie
// inside the image (16 border cases).  Merged in float64 on the host whenever both weight sets are present; used by the Winograd path
// of the projection (convlstm_sequence), the two-step form stays for the small-batch direct path and for DT_TRK_MERGE=0.
static void gate_interleave_map(int U, std::vector<int> &n_map);
static int build_merged_xproj(dt_ctx *ctx)
{
    if (ctx->trk_wxm_wino) { (void)hipStreamSynchronize(ctx->stream); s3_drop(ctx, ctx->trk_wxm_wino); (void)hipFree(ctx->trk_wxm_wino); ctx->trk_wxm_wino = nullptr; }
    if (ctx->trk_bx16) { (void)hipStreamSynchronize(ctx->stream); (void)hipFree(ctx->trk_bx16); ctx->trk_bx16 = nullptr; }
    const int Cb = ctx->cb, U = ctx->trk_units, N4 = 4 * U, Csrc = Cb + 1024;
    if (!ctx->pol.trk_merge || !U || ctx->conv23_hwio.size() != (size_t)1024 * Cb || ctx->trk_hkernel.size() != (size_t)9 * Csrc * N4) return DT_OK;
    if (!wino_wanted(ctx, 3, 1024, N4) || wino_tile(ctx, false) != 6 || ctx->trk_wino_ts != 6) return DT_OK;
    const float *wk = ctx->trk_hkernel.data(), *w23 = ctx->conv23_hwio.data();
    std::vector<float> merged((size_t)9 * 1024 * N4);
    {   // 9 x 1024 independent rows of Cb x 4U float64 multiply-adds (1.6 G at C = 12): over the host's threads
        auto work = [&](int r0, int r1) {
            std::vector<double> row(N4);
            for (int r = r0; r < r1; ++r) {
                const int tap = r / 1024, c = r % 1024;
                const float *src = wk + ((size_t)tap * Csrc + Cb + c) * N4;           // Keras input order: x_bbox first, then conv_feat
                for (int n = 0; n < N4; ++n) row[n] = src[n];
                for (int j = 0; j < Cb; ++j) {
                    const double a = w23[(size_t)c * Cb + j];
                    const float *wb = wk + ((size_t)tap * Csrc + j) * N4;
                    for (int n = 0; n < N4; ++n) row[n] += a * wb[n];
                }
                float *dst = merged.data() + ((size_t)tap * 1024 + c) * N4;
                for (int n = 0; n < N4; ++n) dst[n] = (float)row[n];
            }
        };
        unsigned nth = std::thread::hardware_concurrency();
        nth = nth < 1 ? 1 : (nth > 16 ? 16 : nth);
        std::vector<std::thread> pool;
        for (unsigned i = 0; i < nth; ++i) pool.emplace_back(work, (int)(9216ull * i / nth), (int)(9216ull * (i + 1) / nth));
        for (auto &th : pool) th.join();
    }
    std::vector<int> n_map;
    gate_interleave_map(U, n_map);
    // bias of border case k: b + sum over the taps (dy, dx) whose pixel exists: dy = 0 needs h > 0, dy = 2 needs h < H - 1, likewise dx
    std::vector<double> tapb((size_t)9 * N4, 0.0);
    for (int tap = 0; tap < 9; ++tap)
        for (int j = 0; j < Cb; ++j) {
            const double bj = ctx->conv23_bias[j];
            const float *wb = wk + ((size_t)tap * Csrc + j) * N4;
            for (int n = 0; n < N4; ++n) tapb[(size_t)tap * N4 + n] += bj * wb[n];
        }
    // row 0: the interior bias (all nine taps); row 1 + k: what border case k takes away from it (the taps outside the image), as a correction
    std::vector<float> b16((size_t)17 * N4);
    for (int k = 0; k < 16; ++k)
        for (int np = 0; np < N4; ++np) {
            const int n = n_map[np];
            double all = ctx->trk_hbias[n], out = 0.0;
            for (int dy = 0; dy < 3; ++dy)
                for (int dx = 0; dx < 3; ++dx) {
                    const bool in = !(dy == 0 && (k & 1)) && !(dy == 2 && (k & 2)) && !(dx == 0 && (k & 4)) && !(dx == 2 && (k & 8));
                    all += tapb[(size_t)(dy * 3 + dx) * N4 + n];
                    if (!in) out += tapb[(size_t)(dy * 3 + dx) * N4 + n];
                }
            if (k == 0) b16[np] = (float)all;
            b16[(size_t)(1 + k) * N4 + np] = (float)(-out);
        }
    int rc = upload_wino(ctx, &ctx->trk_wxm_wino, 6, merged.data(), 1024, N4, nullptr, 1024, n_map.data(), N4, nullptr, true);
    if (rc) return rc;
    return upload(ctx, &ctx->trk_bx16, b16);
}

extern "C" int dt_load_darknet_weights(dt_ctx *ctx, const float *h_blob, size_t n_floats, size_t *consumed)
{
    if (!ctx || !h_blob) return DT_ERR_ARG;
    if (!ctx->cb) return dt_fail(ctx, DT_ERR_STATE, "dt_detector_config must be called first");
    size_t off = 4;   // WeightReader.offset = 4 (utils.py:140)
    auto take = [&](size_t n) -> const float * {
        if (off + n > n_floats) return nullptr;
        const float *p = h_blob + off;
        off += n;
        return p;
    };
    struct Spec { int idx, k, cin, cout; };
    std::vector<Spec> specs;
    for (auto &t : TRUNK) specs.push_back({t[0], t[1], t[2], t[3]});
    specs.push_back({21, 1, 512, 64});
    specs.push_back({22, 3, 1280, 1024});
    std::vector<float> hwio, scale, shift;
    for (const Spec &s : specs) {   // file order conv_1 .. conv_22, each: beta, gamma, mean, var, kernel
        const float *beta = take(s.cout), *gamma = take(s.cout), *mean = take(s.cout), *var = take(s.cout);
        const float *kern = take((size_t)s.cout * s.cin * s.k * s.k);
        if (!kern) return dt_fail(ctx, DT_ERR_ARG, "weights blob too short at conv_%d", s.idx);
        scale.resize(s.cout); shift.resize(s.cout);
        for (int c = 0; c < s.cout; ++c) {
            scale[c] = gamma[c] * (1.0f / sqrtf(var[c] + BN_EPS));
            shift[c] = beta[c] - mean[c] * scale[c];
        }
        oihw_to_hwio(kern, s.cout, s.cin, s.k, hwio);
        if (s.idx == 1) {   // direct kernel layout [27][32]
            std::vector<float> w(27 * 32);
            for (int t = 0; t < 27; ++t)
                for (int c = 0; c < 32; ++c) w[t * 32 + c] = hwio[(size_t)t * 32 + c] * scale[c];
            int rc = upload(ctx, &ctx->conv1_w, w);
            if (rc) return rc;
            rc = upload(ctx, &ctx->conv1_b, shift);
            if (rc) return rc;
            {   // the same weights as three bf16 terms, plain (float32 frames) and with normalize()'s 1/255 folded in (uint8 frames): conv1_s3_kernel
                std::vector<unsigned> w3(C1_W3_WORDS, 0u), w3u8(C1_W3_WORDS, 0u);   // 1536 table words + 4 zero words (the kernel's source of padding pixels)
                conv1_split_tables(w.data(), false, w3.data());
                conv1_split_tables(w.data(), true, w3u8.data());
                for (auto pr : {std::make_pair(&ctx->conv1_w3, &w3), std::make_pair(&ctx->conv1_w3u8, &w3u8)}) {
                    if (*pr.first) { (void)hipStreamSynchronize(ctx->stream); (void)hipFree(*pr.first); *pr.first = nullptr; }
                    HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(pr.first), pr.second->size() * sizeof(unsigned)));
                    HIP_TRY(ctx, hipMemcpy(*pr.first, pr.second->data(), pr.second->size() * sizeof(unsigned), hipMemcpyHostToDevice));
                }
            }
            ctx->conv1_hwio32.assign((size_t)9 * 32 * 32, 0.0f);   // [3][3][32 (3 used)][32]
            for (int t = 0; t < 9; ++t)
                for (int ci = 0; ci < 3; ++ci)
                    for (int c = 0; c < 32; ++c)
                        ctx->conv1_hwio32[((size_t)t * 32 + ci) * 32 + c] = hwio[((size_t)t * 3 + ci) * 32 + c];
            ctx->conv1_scale = scale; ctx->conv1_shift = shift;
        } else {
            int rc = load_conv_layer(ctx, s.idx, s.k, s.cin, s.cout, hwio.data(), scale.data(), shift.data());
            if (rc) return rc;
        }
    }
    {   // conv_23: bias then kernel, no BN (KerasYOLO.py:264-269)
        const int co = ctx->cb;
        const float *bias = take(co);
        const float *kern = take((size_t)co * 1024);
        if (!kern) return dt_fail(ctx, DT_ERR_ARG, "weights blob too short at conv_23");
        oihw_to_hwio(kern, co, 1024, 1, hwio);
        int rc = load_conv_layer(ctx, 23, 1, 1024, co, hwio.data(), nullptr, bias);
        if (rc) return rc;
        ctx->conv23_hwio.assign(hwio.begin(), hwio.begin() + (size_t)1024 * co);      // [1024][co]
        ctx->conv23_bias.assign(bias, bias + co);
    }
    if (consumed) *consumed = off;
    graphs_clear(ctx);
    ctx->det_loaded = true;
    return build_merged_xproj(ctx);
}

// Direct-form FLOPs (2*M*K*N of the reference's convolution) and bytes (in + weights + out, fp32) of every layer, booked
// under the kernel FAMILY whose launch computed it -- "conv_direct_form" (conv_igemm_f32 launches), "conv_direct_form_s3"
// (wino_gemm_s3 launches), "conv_direct_form_fused" (the fused Winograd kernel) -- so that bench.py divides each family's
// algorithmic work by that family's own time and no family is credited with work another kernel did.
enum { DF_IGEMM = 0, DF_FUSED = 1, DF_S3 = 2, DF_CONV1 = 3, DF_C3H2 = 4 };
static void prof_direct_form(dt_ctx *ctx, double flops, double bytes, int family = DF_IGEMM)
{
    if (!ctx->prof) return;
    ProfEntry &e = ctx->prof_tab[family == DF_FUSED ? "conv_direct_form_fused" : (family == DF_S3 ? "conv_direct_form_s3" : (family == DF_CONV1 ? "conv_direct_form_conv1" :
                                 (family == DF_C3H2 ? "conv_direct_form_c3h2" : "conv_direct_form")))];
    e.flops += flops;
    e.bytes += bytes;   // in + weights + out of the reference's layer, float32
}

// ---------------------------------------------------------------------------
// Winograd path for the 3x3 layers from conv_3 up and the ConvLSTM convolutions (winograd.hip)
// ---------------------------------------------------------------------------
// Policy::wino: 1 (default) = layers with Cin >= 64 and Cout >= 128 (conv_3 and up: below that the batched
//          GEMMs have K <= 32 and the transforms' traffic costs more than the MFMA work saved) when a
//          launch has enough tiles (wino_runs);
//          0 = never (direct MFMA form everywhere); 2 = every 3x3 layer the transforms support,
//          at any size (parity tests of the path at small shapes).  Applied when weights are loaded.
void policy_from_env(Policy &p, int pin_override)
{
    auto geti = [](const char *name, int dflt) { const char *e = getenv(name); return e ? atoi(e) : dflt; };
    Policy d;
    p.wino = geti("DT_WINO", d.wino);
    p.wino_tile = geti("DT_WINO_TILE", d.wino_tile);
    p.wino_minc = geti("DT_WINO_MINC", d.wino_minc);
    p.wino_minn = geti("DT_WINO_MINN", d.wino_minn);
    p.wino_mint = geti("DT_WINO_MINT", d.wino_mint);
    { const char *e = getenv("DT_WINO_WS_GB"); p.wino_ws_gb = e ? atof(e) : d.wino_ws_gb; }
    p.mosaic = geti("DT_WINO_MOSAIC", d.mosaic);
    p.fused4 = geti("DT_WINO_FUSED4", d.fused4);
    p.c3h2 = geti("DT_C3H2", d.c3h2);
    p.c3fuse = geti("DT_C3FUSE", d.c3fuse);
    p.c3h2_blocks2 = geti("DT_C3H2_BLOCKS2", d.c3h2_blocks2);
    p.wino_cfg = geti("DT_WINO_CFG", d.wino_cfg);
    p.wino_gn = geti("DT_WINO_GN", d.wino_gn);
    p.wino_grid_in = geti("DT_WINO_GRID_IN", d.wino_grid_in);
    p.wino_grid_out = geti("DT_WINO_GRID_OUT", d.wino_grid_out);
    p.wino_thr_out = geti("DT_WINO_THR_OUT", d.wino_thr_out);
    p.ksplit = geti("DT_KSPLIT", d.ksplit);
    p.conv_cfg = geti("DT_CONV_CFG", d.conv_cfg);
    p.wino_coop = geti("DT_WINO_COOP", d.wino_coop);
    p.s3 = geti("DT_S3", d.s3);
    p.s3_conv1 = geti("DT_S3_CONV1", d.s3_conv1);
    p.s3_mink = geti("DT_S3_MINK", d.s3_mink);
    p.s3_minrows = geti("DT_S3_MINROWS", d.s3_minrows);
    p.s3_minrows_h2 = getenv("DT_S3_MINROWS") ? p.s3_minrows : d.s3_minrows_h2;
    p.s3_1x1 = geti("DT_S3_1X1", d.s3_1x1);
    p.s3_1x1_mink = geti("DT_S3_1X1_MINK", d.s3_1x1_mink);
    p.s3_1x1_minrows = geti("DT_S3_1X1_MINROWS", d.s3_1x1_minrows);
    p.s3_rec_minrows = geti("DT_S3_REC_MINROWS", d.s3_rec_minrows);
    p.s3_rec_minrows_h2 = getenv("DT_S3_REC_MINROWS") ? p.s3_rec_minrows : d.s3_rec_minrows_h2;
    p.s3_half = geti("DT_S3_HALF", d.s3_half);
    p.s3_h2 = geti("DT_S3_H2", d.s3_h2);
    p.h2_minframes = geti("DT_H2_MINFRAMES", d.h2_minframes);
    p.persist = geti("DT_PERSIST", d.persist);
    p.xcd_remap = geti("DT_XCD_REMAP", d.xcd_remap);
    p.tile_gn = geti("DT_TILE_GN", d.tile_gn);
    p.trk_merge = geti("DT_TRK_MERGE", d.trk_merge);
    p.pin = pin_override >= 0 ? pin_override : geti("DT_PIN", d.pin);
    if (p.pin) {      // every choice below otherwise looks at the number of frames / rows / tiles of the launch
        p.wino = 2; p.mosaic = 1; p.fused4 = 2; p.s3_half = -1; p.ksplit = 1;
        if (p.s3) p.s3 = 2;
        p.trk_merge = 0;      // ONE form of the input projection, whichever entry point carries the frame (the merged weights are a different rounding of the
                              // same network, and dt_track_recurrent on a caller's z rows cannot take them)
        // (the fp16 form of the split GEMM is off under DT_PIN as well: h2_wanted)
    }
}

static bool wino_wanted(const dt_ctx *ctx, int ks, int cin, int cout)
{
    const Policy &p = ctx->pol;
    if (ks != 3 || p.wino == 0 || cin % 32 || cout % 4) return false;
    return p.wino == 2 || (cin >= p.wino_minc && cout >= p.wino_minn);
}

// Output tile of the Winograd form: 6 = F(6x6,3x3) (default), 4 = F(4x4,3x3), 2 = F(2x2,3x3); Policy::wino_tile
// forces one size for every layer.  Applied when weights are loaded.
static int wino_tile(const dt_ctx *ctx, bool recurrent)
{
    // The ConvLSTM recurrent convolution keeps F(4x4): its per-step GEMM has only clips/4 * 49 rows, which F(6x6)
    // (clips/9 * 49) fills no better, and the F(6x6) gate-update transform needs all 256 VGPRs (2.0 vs 5.3 TB/s).
    const int t = ctx->pol.wino_tile ? ctx->pol.wino_tile : (recurrent ? 4 : 6);
    return (t == 2 || t == 4) ? t : 6;
}

static bool wino_runs(const dt_ctx *ctx, const float *wino_wt, int ts, int B, int H, int W, int cin, int N)
{
    if (!wino_wt) return false;
    const long long mt = (long long)B * ((H + ts - 1) / ts) * ((W + ts - 1) / ts);
    if (mt >= (1ll << 31) / 64) return false;
    // V and M' workspaces are (ts+2)^2 * tiles * (Cin + N) floats (27 GB for conv_3 at 1440 frames); a launch
    // that would need more than Policy::wino_ws_gb (default 96) takes the direct form instead of failing to allocate
    if (4.0 * (ts + 2) * (ts + 2) * (double)mt * ((double)cin + N) > ctx->pol.wino_ws_gb * 1e9) return false;
    // below this many tiles the batched GEMMs' row tiles are mostly empty and the direct (split-K) form wins
    // (detector-only sweep, batch 1/4/8/16: threshold 512 -> 679/1822/2672/3487 frames/s, 64 -> 695/1960/3341/4402,
    // 16 -> 556/1947/3337/4427)
    const int mint = ctx->pol.wino_mint;
    // (F(4x4) -- the ConvLSTM recurrent step -- from 16 tiles = ONE clip at 13x13 since round 6: with the step's GEMM on the split kernel a 30-frame clip takes 4.7 instead
    //  of 12.3 ms, two clips 6.5 instead of 14.3, three 7.7 instead of 15.7 (profiles/r06_experiments.txt section 10); on the fp32 kernel too the Winograd form wins there)
    return ctx->pol.wino == 2 || mt >= (mint > 0 ? mint : (ts == 2 ? 256 : (ts == 4 ? 16 : 32)));
}

static int upload_wino(dt_ctx *ctx, float **dst, int ts, const float *hwio, int cin_src, int cout_src, const int *cin_map,
                       int cin_dst, const int *n_map, int npad, const float *scale, bool want_s3)
{
    std::vector<float> u((size_t)(ts + 2) * (ts + 2) * npad * cin_dst);
    wino_pack_weights(ts, hwio, cin_src, cout_src, cin_map, cin_dst, n_map, npad, scale, u.data());
    s3_drop(ctx, *dst);
    int rc = upload(ctx, dst, u);
    if (rc) return rc;
    // also in the split-bf16 form of wino_gemm_s3.hip (the split runs on the device, from the fp32 copy) -- only where run_wino
    // can use it: F(6x6) weights of a plain layer, F(4x4) weights of the ConvLSTM recurrent convolution (the caller says which)
    if (want_s3 && ctx->pol.s3 != 0 && cin_dst % 32 == 0 && npad % 128 == 0) {
        unsigned short *s3 = nullptr;
        HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&s3), u.size() * 3 * sizeof(unsigned short)));
        if (launch_wino_s3_pack(ctx->stream, *dst, (ts + 2) * (ts + 2), npad, cin_dst, s3) || hipStreamSynchronize(ctx->stream) != hipSuccess) {
            (void)hipFree(s3);
            return dt_fail(ctx, DT_ERR_DEVICE, "split-bf16 weight pack failed");
        }
        ctx->wino_s3[*dst] = s3;
        if (ctx->pol.s3_h2 != 0 && (ts + 2) * (ts + 2) <= 64) {      // ... and the fp16 form next to it (DT_PIN runs take the bf16 one)
            dt_ctx::H2Weights h;
            const int P = (ts + 2) * (ts + 2);
            HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&h.terms), u.size() * 2 * sizeof(unsigned short)));
            if (hipMalloc(reinterpret_cast<void **>(&h.pscale), P * sizeof(float)) != hipSuccess) { (void)hipFree(h.terms); return dt_fail(ctx, DT_ERR_DEVICE, "hipMalloc failed"); }
            if (launch_wino_h2_pack(ctx->stream, *dst, P, npad, cin_dst, ts, amax_slot(ctx, AMAX_PACK), h.terms, h.pscale) || hipStreamSynchronize(ctx->stream) != hipSuccess) {
                (void)hipFree(h.terms); (void)hipFree(h.pscale);
                return dt_fail(ctx, DT_ERR_DEVICE, "fp16-form weight pack failed");
            }
            ctx->wino_h2[*dst] = h;
        }
    }
    return DT_OK;
}

// Tile configuration of the P batched GEMMs [Mt x Cin] x [Cin x N] (persistent launch, conv_igemm.hip).
// Measured (tools/wino_ab.sh): a GEMM tile is only Cin/32 = 4..40 chunks long, so what matters is what
// happens BETWEEN tiles.  One tile per workgroup: 128x128 (two workgroups per CU overlap each other's
// epilogue/prologue) 128.6 vs 256x256 121.4 TFLOP/s at K=1024.  Persistent with the next tile's first DMA
// issued inside the last chunk: 256x256 132.4 (K=1024), 128.3 (K=512), 122.3 (K=256) -- ahead of 128x128
// everywhere, so the wide tile is taken whenever wave quantisation does not eat the gain.
// Small batched GEMMs (a few frames per call): what counts is the busiest CU.  cost = rounds over the 256 CUs x tile
// area / relative efficiency of the tile shape, in units of one 128x128 tile of this K (measured at batch 8,
// tools/batch8_layers.py: the 13x13 layers' F(4x4) GEMMs as 36 x 8 tiles of 128x128 leave 224 CUs with one tile and 32
// with two -- 139 us; as 36 x 16 tiles of 128x64: 112 us; as F(6x6) with 64 x 8 tiles of 64x128 (50 Winograd tiles fit
// one 64-row tile): 81 us).  Returns the cheapest shape and its cost.
static double small_gemm_cost(int Mt, int N, int P, int *cfg_out)
{
    const long long t128 = (long long)P * ((Mt + 127) / 128) * ((N + 127) / 128);
    const long long t64n = (long long)P * ((Mt + 127) / 128) * ((N + 63) / 64);
    const long long t64m = (long long)P * ((Mt + 63) / 64) * ((N + 127) / 128);
    const long long t256 = (long long)P * ((Mt + 255) / 256) * ((N + 255) / 256);
    double best = (double)((t128 + 255) / 256);
    int cfg = CFG_128x128;
    const double c64n = N % 64 == 0 ? (double)((t64n + 255) / 256) * 0.5 / 0.92 : 1e30;
    const double c64m = (double)((t64m + 255) / 256) * 0.5 / 0.92;
    const double c256 = N % 256 == 0 ? (double)((t256 + 255) / 256) * 4.0 / 1.03 : 1e30;
    if (c64n < best) { best = c64n; cfg = CFG_128x64; }
    if (c64m < best) { best = c64m; cfg = CFG_64x128; }
    if (c256 < best) { best = c256; cfg = CFG_256x256; }
    if (cfg_out) *cfg_out = cfg;
    return best;
}

static int pick_cfg_gemm(int Mt, int N, int P)
{
    const long long t128 = (long long)P * ((Mt + 127) / 128) * ((N + 127) / 128);
    if (t128 <= 4096) {
        int cfg;
        (void)small_gemm_cost(Mt, N, P, &cfg);
        return cfg;
    }
    if (N % 256 == 0) {
        const long long t256 = (long long)P * ((Mt + 255) / 256) * (N / 256);
        const double e256 = (double)Mt / (((Mt + 255) / 256) * 256.0) * (double)t256 / (double)(((t256 + 255) / 256) * 256);
        const double e128 = (double)Mt / (((Mt + 127) / 128) * 128.0) * (double)t128 / (double)(((t128 + 511) / 512) * 512);
        if (e256 * 1.03 > e128) return CFG_256x256;
    }
    return CFG_128x128;
}

// every MFMA-kernel launch of this file goes through here: the context's forced tile configuration (tests, A/B) rides along
static int launch_igemm(dt_ctx *ctx, ConvArgs &a, int ks, int order, int epi, int cfg)
{
    a.force_cfg = ctx->pol.conv_cfg >= 0 ? ctx->pol.conv_cfg + 1 : 0;
    a.no_persist = ctx->pol.persist ? 0 : 1;
    a.xcd_remap = ctx->pol.xcd_remap;
    a.gn_default = ctx->pol.tile_gn >= 0 ? ctx->pol.tile_gn + 1 : 0;
    return launch_conv_igemm(ctx->stream, a, ks, order, epi, cfg);
}

struct WinoIO {
    const float *in; long long in_bs; int in_ld;      // NHWC input
    float *out; long long out_bs; int out_ld;         // full-resolution output (null: pooled only)
    float *out2; int out2_ld;                         // 2x2 pooled output or null
    const float *xproj; long long xp_bs; int xp_ld;   // gates variant (cstate != null)
    float *cstate; long long c_bs; int c_ld;
    const float *bias16;                              // border-aware bias [16][N] instead of `bias` (WinoArgs::bias16) or null
    int amax_out_slot;                                // > 0: the output transform takes max |x| of what it stores into this slot (and the tensor is tagged with it)
};

// Tile geometry of a Winograd launch.  Mosaic factor g: g x g frames with zero separators share one virtual image
// (winograd.hip:vpixel) when that needs fewer tiles per frame (13x13 F(4x4): 12.25 instead of 16); pooled outputs need
// frame-aligned tiles.
struct WinoGeom { int g, th, tw, Mt; };
static WinoGeom wino_geometry(const dt_ctx *ctx, int ts, int B, int H, int W, bool pooled)
{
    WinoGeom q;
    q.g = 1;
    const int g_env = ctx->pol.mosaic;   // 1: never (tests, A/B), 2 / 3 / 4: force
    double best = (double)((H + ts - 1) / ts) * ((W + ts - 1) / ts);
    for (int g = 2; g <= 4 && !pooled && g_env != 1; ++g) {
        const double t = (double)((g * (H + 1) + ts - 1) / ts) * ((g * (W + 1) + ts - 1) / ts) / (g * g);
        if ((t < best * 0.97 && B >= g * g) || g_env == g) { best = t; q.g = g; }
    }
    if (q.g == 1) { q.th = (H + ts - 1) / ts; q.tw = (W + ts - 1) / ts; q.Mt = B * q.th * q.tw; }
    else {
        q.th = (q.g * (H + 1) + ts - 1) / ts; q.tw = (q.g * (W + 1) + ts - 1) / ts;
        q.Mt = ((B + q.g * q.g - 1) / (q.g * q.g)) * q.th * q.tw;
    }
    return q;
}

static int run_wino(dt_ctx *ctx, const float *wino_wt, int ts, const float *bias, int cin, int N, int npad, int B, int H,
                    int W, const WinoIO &io, float slope, const char *tag, int cin_alg = 0 /* channels of the reference's layer when cin is padded */,
                    int in_slot = AMAX_TEST /* max-|x| slot of the input tensor (fp16 form); AMAX_ONE: bounded by 1, nothing to measure */)
{
    const double cin_df = cin_alg > 0 ? cin_alg : cin;
    if (io.out) amax_forget(ctx, io.out, (long long)B * H * W * io.out_ld);
    if (io.out2) amax_forget(ctx, io.out2, (long long)B * H * W / 4 * io.out2_ld);
    WinoArgs w;
    memset(&w, 0, sizeof(w));
    w.B = B; w.H = H; w.W = W; w.ts = ts;
    w.coop = ctx->pol.wino_coop;
    w.grid_in = ctx->pol.wino_grid_in; w.grid_out = ctx->pol.wino_grid_out; w.thr_out = ctx->pol.wino_thr_out;
    {
        const WinoGeom q = wino_geometry(ctx, ts, B, H, W, io.out2 != nullptr);
        w.g = q.g; w.th = q.th; w.tw = q.tw; w.Mt = q.Mt;
    }
    if (ctx->prof && !ctx->capturing) {   // which mosaic / tile size a launch took (asserted by the configuration parity tests)
        char mtag[40];
        snprintf(mtag, sizeof(mtag), "wino_mosaic:g%d_ts%d", w.g, ts);
        ctx->prof_tab[mtag].launches += 1;
    }
    const int P = (ts + 2) * (ts + 2);
    const size_t mt = (size_t)w.Mt;
    // the GEMMs on the bf16 pipe with split operands (wino_gemm_s3.hip) where that form exists and wins: long K, enough rows
    const unsigned short *u_s3 = nullptr;
    const bool rec = ts == 4 && io.cstate;      // the recurrent step: F(4x4), gate update in the output transform
    // (row thresholds: the fp16 form wins from far fewer rows than the bf16 form)
    const bool h2_avail = h2_wanted(ctx) && ctx->wino_h2.find(wino_wt) != ctx->wino_h2.end();
    const int minrows = h2_avail ? ctx->pol.s3_minrows_h2 : ctx->pol.s3_minrows, rec_minrows = h2_avail ? ctx->pol.s3_rec_minrows_h2 : ctx->pol.s3_rec_minrows;
    if (((ts == 6 && !io.cstate) || rec) && ctx->pol.s3 != 0 && cin % 32 == 0 && N % 128 == 0 && npad % 128 == 0 && wino_gemm_s3_usable(w.Mt, cin, N) &&
        (ctx->pol.s3 == 2 || (cin >= ctx->pol.s3_mink && (rec ? (rec_minrows > 0 && w.Mt >= rec_minrows) : w.Mt >= minrows)))) {
        auto it = ctx->wino_s3.find(wino_wt);
        if (it != ctx->wino_s3.end()) u_s3 = it->second;
    }
    // ... in the fp16 form (two terms of scaled operands, three products) where its weights exist
    const dt_ctx::H2Weights *h2 = nullptr;
    if (u_s3 && h2_wanted(ctx)) {
        auto ih = ctx->wino_h2.find(wino_wt);
        if (ih != ctx->wino_h2.end()) { h2 = &ih->second; u_s3 = h2->terms; }
    }
    const int NT = h2 ? 2 : 3;
    const size_t mp = (mt + 255) / 256 * 256;
    float *V = ws_get(ctx, "wino_v", u_s3 ? (size_t)P * NT * mp * cin * sizeof(unsigned short) : P * mt * cin * sizeof(float));
    float *Mp = ws_get(ctx, "wino_m", P * mt * N * sizeof(float));
    if (!V || !Mp) return DT_ERR_DEVICE;
    if (u_s3) { w.v_s3 = reinterpret_cast<unsigned short *>(V); w.Mp = (int)mp; }
    const unsigned *amax = nullptr;
    if (h2) {      // V's power of two comes from the input's max |x|
        amax = in_slot == AMAX_ONE ? amax_slot(ctx, AMAX_ONE) : ensure_amax(ctx, io.in, (long long)B * H * W, cin, io.in_ld, in_slot);
        if (!amax) return DT_ERR_DEVICE;
        w.nt = 2; w.amax = amax;
    }
    w.in = io.in; w.in_bs = io.in_bs; w.in_ld = io.in_ld; w.C = cin; w.v = V;
    w.m = Mp; w.m_ld = N; w.N = N; w.bias = bias; w.bias16 = io.bias16; w.slope = slope;
    w.out = io.out; w.out_bs = io.out_bs; w.out_ld = io.out_ld; w.out2 = io.out2; w.out2_ld = io.out2_ld;
    w.xproj = io.xproj; w.xp_bs = io.xp_bs; w.xp_ld = io.xp_ld;
    w.cstate = io.cstate; w.c_bs = io.c_bs; w.c_ld = io.c_ld;
    {
        ProfScope ps(ctx, "wino_input", 0.0, 4.0 * (double)B * H * W * cin + (u_s3 ? 2.0 * NT : 4.0) * (double)P * mt * cin, tag);
        const int rc = launch_wino_input(ctx->stream, w);
        if (rc) return dt_fail(ctx, rc == 2 ? DT_ERR_ARG : DT_ERR_DEVICE, "%s: Winograd input transform launch failed", tag);
    }
    if (u_s3) {
        GemmS3Args g;
        memset(&g, 0, sizeof(g));
        g.a = w.v_s3; g.b = u_s3; g.c = Mp; g.c_ps = (long long)mt * N; g.P = P; g.Mt = w.Mt; g.Mp = (int)mp; g.N = N; g.Np = npad;
        g.K = cin; g.ldc = N; g.half = ctx->pol.s3_half;
        if (h2) { g.nt = 2; g.pscale = h2->pscale; g.amax = amax; }
        // flops = EXECUTED 16-bit MFMA work (six partial products per multiply, three in the fp16 form); bytes = V + U (NT 16-bit terms each) + M'
        ProfScope ps(ctx, "conv_gemm_s3", wino_gemm_s3_flops(g), (double)P * (2.0 * NT * mt * cin + 2.0 * NT * (double)cin * N + 4.0 * (double)mt * N), tag);
        if (ctx->prof && !ctx->capturing) ctx->prof_tab[h2 ? "s3_form:f16x2" : "s3_form:bf16x3"].launches += 1;
        prof_direct_form(ctx, 2.0 * B * H * W * 9.0 * cin_df * N,
                         4.0 * ((double)B * H * W * cin_df + 9.0 * cin_df * N + (io.out ? (double)B * H * W * N : 0.0) +
                                (io.out2 ? (double)B * H * W * N / 4.0 : 0.0) + (io.cstate ? 4.0 * B * H * W * N / 4.0 : 0.0)), DF_S3);
        if (ctx->prof && !ctx->capturing) ctx->prof_tab[wino_gemm_s3_half_chosen(g, 0) ? "s3_tile:128x2" : "s3_tile:256"].launches += 1;
        const int rc = launch_wino_gemm_s3(ctx->stream, g, 0);
        if (rc) return dt_fail(ctx, rc == 2 ? DT_ERR_ARG : DT_ERR_DEVICE, "%s: split-bf16 Winograd GEMM launch failed (rc=%d)", tag, rc);
    } else {
        ConvArgs a;
        memset(&a, 0, sizeof(a));
        a.in = V; a.in_ld = cin; a.in_bs = (long long)mt * cin;
        a.wt = wino_wt; a.bias = nullptr;
        a.out = Mp; a.out_ld = N; a.out_bs = (long long)mt * N;
        a.B = 1; a.H = 1; a.W = w.Mt; a.Cin = cin; a.N = N; a.M = w.Mt; a.K = cin;
        a.npad = npad; a.slope = 1.0f;
        a.zbatch = P; a.z_in = (long long)mt * cin; a.z_wt = (long long)npad * cin; a.z_out = (long long)mt * N;
        // flops = executed MFMA work of the P GEMMs (the direct form of the same layer is 9*ts*ts/P times
        // that on whole tiles: 4x for F(4x4,3x3), 2.25x for F(2x2,3x3)); bytes = V + U + M'
        ProfScope ps(ctx, "conv_igemm", 2.0 * P * mt * (double)cin * N,
                     4.0 * P * ((double)mt * cin + (double)cin * N + (double)mt * N), tag);
        prof_direct_form(ctx, 2.0 * B * H * W * 9.0 * cin_df * N,
                         4.0 * ((double)B * H * W * cin_df + 9.0 * cin_df * N + (io.out ? (double)B * H * W * N : 0.0) +
                                (io.out2 ? (double)B * H * W * N / 4.0 : 0.0) + (io.cstate ? 4.0 * B * H * W * N / 4.0 : 0.0)));
        int cfg = pick_cfg_gemm(w.Mt, N, P);
        if (ctx->pol.wino_cfg >= 0) cfg = ctx->pol.wino_cfg;               // A/B runs
        if (ctx->pol.wino_gn >= 0) a.tile_gn = -ctx->pol.wino_gn - 1;      // A/B runs: column-group width (see launch_conv_igemm)
        const int rc = launch_igemm(ctx, a, 1, ORD_LINEAR, EPI_PLAIN, cfg);
        if (rc) return dt_fail(ctx, rc == 2 ? DT_ERR_ARG : DT_ERR_DEVICE, "%s: Winograd GEMM launch failed (rc=%d)", tag, rc);
    }
    {
        const double outb = (io.out ? (double)B * H * W * N : 0.0) + (io.out2 ? (double)B * H * W * N / 4.0 : 0.0) +
                            (io.cstate ? 3.0 * B * H * W * N / 4.0 + (double)B * H * W * N : 0.0);
        ProfScope ps(ctx, "wino_output", 0.0, 4.0 * ((double)P * mt * N + outb), tag);
        w.amax_out = io.amax_out_slot > 0 && h2_wanted(ctx) ? amax_slot(ctx, io.amax_out_slot) : nullptr;
        if (!wino_output_fills_amax(w, io.cstate != nullptr)) w.amax_out = nullptr;
        const int rc = launch_wino_output(ctx->stream, w, io.cstate != nullptr);
        if (rc) return dt_fail(ctx, rc == 2 ? DT_ERR_ARG : DT_ERR_DEVICE, "%s: Winograd output transform launch failed", tag);
        if (w.amax_out) {      // (the pooled tensor when there is one: its consumer is the next layer; the unpooled twin of conv_13 feeds conv_21's fp32 kernel)
            if (io.out2) amax_note(ctx, io.out2, (long long)B * H * W / 4 * io.out2_ld, N, io.amax_out_slot);
            else amax_note(ctx, io.out, (long long)B * H * W * io.out_ld, N, io.amax_out_slot);
        }
    }
    return DT_OK;
}

struct Dest { float *p; int ld; };

// Tile configuration for a plain / pooled 3x3 or 1x1 layer.  The loss of the MFMA kernel scales
// with the bytes staged per MFMA, so the 16-wave 256x256 tile (half the staging of 128x128) is
// ~5 % faster on the 3x3 layers -- when Cout is a multiple of 256 and there are enough tiles that
// its single resident workgroup per CU does not lose more to wave quantisation than it gains.
static int pick_cfg(int M, int cout, int ks)
{
    if (cout <= 64) return CFG_128x64;
    if (ks == 3 && cout % 256 == 0) {
        const long long t256 = (long long)((M + 255) / 256) * (cout / 256);
        const long long t128 = (long long)((M + 127) / 128) * (cout / 128);
        const double e256 = (double)t256 / (double)(((t256 + 255) / 256) * 256);
        const double e128 = (double)t128 / (double)(((t128 + 511) / 512) * 512);
        if (e256 * 1.05 > e128) return CFG_256x256;
    }
    // 1x1 layers over a whole batch (short K, thousands of row tiles): the 256-row tile halves the weight staging per
    // MFMA -- conv_7 2.43 -> 2.33, conv_10/12 2.19/2.14 -> 2.10/2.05, conv_15/17 2.03/2.01 -> 1.97/1.93 ms per 1440
    // frames; 256x256 is worse for N = 512 and unusable for N = 128
    if (ks == 1 && cout >= 128 && (long long)((M + 255) / 256) * ((cout + 127) / 128) >= 4096) return CFG_256x128;
    return CFG_128x128;
}

// A 1x1 layer with enough channels and pixels (conv_7 / 10 / 12 / 15 / 17 at the benched size) runs as ONE split GEMM straight on the
// producing layer's fp32 activation: the kernel splits its A fragments itself (wino_gemm_s3.hip, VF), bias as an extra K stage,
// LeakyReLU in the epilogue.  (Rounds 3-4 had the producer's output transform write pre-split rows for it.)
static bool s3_1x1_eligible(const dt_ctx *ctx, const ConvLayer &L, long long M)
{
    return L.ks == 1 && L.wt_s3 && ctx->pol.s3 != 0 && ctx->pol.s3_1x1 != 0 && L.cin % 32 == 0 && L.cout >= 64 && M < (1ll << 31) - 256 &&
           (ctx->pol.s3 == 2 || (L.cin >= ctx->pol.s3_mink && L.cin >= ctx->pol.s3_1x1_mink && M >= ctx->pol.s3_minrows && M >= ctx->pol.s3_1x1_minrows));
}

static int run_conv(dt_ctx *ctx, const ConvLayer &L, const float *in, int in_ld, int B, int H, int W, float *out,
                    int out_ld, int order, int epi, float slope, float *out2 = nullptr, int out2_ld = 0)
{
    {   // what this layer overwrites (the pooled epilogues write a quarter of the pixels; the s2d epilogue a quarter with 4x the row)
        const long long pix = (long long)B * H * W;
        amax_forget(ctx, out, (epi == EPI_POOL || epi == EPI_S2D ? pix / 4 : pix) * out_ld);
        if (out2) amax_forget(ctx, out2, pix / 4 * out2_ld);
    }
    // (the split 1x1 form reads its A rows with 16-byte DMA pieces: a caller tensor at a 4 / 8 / 12-byte offset takes the fp32 kernel instead of failing)
    if (L.bias_s3 && ctx->s3_ones && epi == EPI_PLAIN && order == ORD_LINEAR && !out2 && in_ld % 4 == 0 && (reinterpret_cast<uintptr_t>(in) & 15) == 0 &&
        (reinterpret_cast<uintptr_t>(out) & 3) == 0 && s3_1x1_eligible(ctx, L, (long long)B * H * W)) {
        const long long M = (long long)B * H * W;
        const bool h2 = L.wt_h2 && L.pscale_h2 && L.npad <= 2048 && h2_wanted(ctx);
        GemmS3Args g;
        memset(&g, 0, sizeof(g));
        g.a_f32 = in; g.a_ld = in_ld; g.b = L.wt_s3; g.c = out; g.c_ps = 0; g.P = 1; g.Mt = (int)M; g.Mp = (int)((M + 255) / 256 * 256); g.N = L.cout; g.Np = L.npad;
        g.half = ctx->pol.s3_half;
        g.K = L.cin; g.ldc = out_ld; g.ones = ctx->s3_ones; g.bias_s3 = L.bias_s3; g.act = 1; g.slope = slope;
        if (h2) {      // the fp16 form: scaled operands (the activation's power of two from its max |x|), the bias added in the epilogue
            g.nt = 2; g.b = L.wt_h2; g.pscale = L.pscale_h2; g.bias = L.bias; g.ones = nullptr; g.bias_s3 = nullptr;
            g.amax = ensure_amax(ctx, in, M, L.cin, in_ld, L.idx >= 1 && L.idx <= 23 ? AMAX_IN + L.idx : AMAX_TEST);
            if (!g.amax) return DT_ERR_DEVICE;
        }
        char tag[32];
        snprintf(tag, sizeof(tag), "conv_%d", L.idx);
        if (h2_wanted(ctx) && amax_out_slot(L)) g.amax_out = amax_slot(ctx, amax_out_slot(L));
        if (ctx->prof && !ctx->capturing) ctx->prof_tab[h2 ? "s3_form:f16x2" : "s3_form:bf16x3"].launches += 1;
        // bytes = A (fp32) + U (NT 16-bit terms) + out
        ProfScope ps(ctx, "conv_gemm_s3", wino_gemm_s3_flops(g), 4.0 * M * L.cin + (h2 ? 4.0 : 6.0) * (double)L.cin * L.cout + 4.0 * (double)M * L.cout, tag);
        prof_direct_form(ctx, 2.0 * M * (double)L.cin * L.cout, 4.0 * ((double)M * L.cin + (double)L.cin * L.cout + (double)M * L.cout), DF_S3);
        const int rc = launch_wino_gemm_s3(ctx->stream, g, 0);
        if (rc) return dt_fail(ctx, rc == 2 ? DT_ERR_ARG : DT_ERR_DEVICE, "%s: split-bf16 1x1 GEMM launch failed (rc=%d)", tag, rc);
        if (g.amax_out) amax_note(ctx, out, M * out_ld, L.cout, amax_out_slot(L));
        return DT_OK;
    }
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.in = in; a.in_ld = in_ld; a.in_bs = (long long)H * W * in_ld;
    a.wt = L.wt; a.bias = L.bias;
    a.out = out; a.out_ld = out_ld; a.out_bs = (long long)H * W * out_ld;
    a.out2 = out2; a.out2_ld = out2_ld;
    a.B = B; a.H = H; a.W = W; a.Cin = L.cin; a.N = L.cout; a.M = B * H * W; a.K = L.ks * L.ks * L.cin;
    a.npad = L.npad;
    a.slope = slope;
    int cfg = pick_cfg(a.M, L.cout, L.ks);
    const double flops = 2.0 * a.M * (double)a.K * L.cout;
    const double bytes = 4.0 * ((double)a.M * L.cin + (double)a.K * L.cout +
                                (double)a.M * L.cout / (epi == EPI_POOL ? 4.0 : 1.0));
    char tag[32];
    snprintf(tag, sizeof(tag), L.idx == 102 ? "tconv_2" : "conv_%d", L.idx);
    // conv_2 / 3 / 5's shapes: the DIRECT convolution in the two-term fp16 form (conv3_h2.hip) once there are enough tiles to fill the chip
    if (L.w3_h2 && L.pscale_w3 && ctx->pol.c3h2 != 0 && h2_wanted(ctx) && in_ld % 4 == 0 && (reinterpret_cast<uintptr_t>(in) & 15) == 0 &&
        ((epi == EPI_PLAIN && order == ORD_LINEAR) || (epi == EPI_POOL && !((H | W) & 1)))) {
        const long long blocks = (long long)B * ((H + 15) / 16) * ((W + 15) / 16) * ((L.cout + 127) / 128);
        if (ctx->pol.c3h2 == 2 || blocks >= c3h2_min_blocks(ctx, L.cin)) {
            Conv3H2Args c;
            memset(&c, 0, sizeof(c));
            c.in = in; c.in_bs = a.in_bs; c.in_ld = in_ld; c.B = B; c.H = H; c.W = W; c.Cin = L.cin; c.N = L.cout; c.Np = L.npad;
            c.w = L.w3_h2; c.pscale = L.pscale_w3; c.bias = L.bias; c.slope = slope;
            if (epi == EPI_POOL) { c.out2 = out; c.out2_ld = out_ld; }
            else { c.out = out; c.out_ld = out_ld; c.out_bs = a.out_bs; }
            c.zeros = ws_get(ctx, "zeros256", 256, /*zero_on_grow=*/true);
            if (!c.zeros) return DT_ERR_DEVICE;
            c.amax = ensure_amax(ctx, in, (long long)B * H * W, L.cin, in_ld, L.idx >= 1 && L.idx <= 23 ? AMAX_IN + L.idx : AMAX_TEST);
            if (!c.amax) return DT_ERR_DEVICE;
            if (amax_out_slot(L)) c.amax_out = amax_slot(ctx, amax_out_slot(L));
            // executed fp16 MFMA FLOPs (three products per multiply, whole tiles); bytes: the input once per channel tile (+ halo 27 / 41 %) and the output
            ProfScope ps(ctx, "conv_direct_h2", conv3_h2_flops(c),
                         4.0 * ((double)B * H * W * L.cin * (L.cout % 128 ? 1.27 : 1.41) * ((L.cout + 127) / 128) + (double)a.M * L.cout / (epi == EPI_POOL ? 4.0 : 1.0)), tag);
            prof_direct_form(ctx, flops, bytes, DF_C3H2);
            const int rc = launch_conv3_h2(ctx->stream, c);
            if (rc) return dt_fail(ctx, rc == 2 ? DT_ERR_ARG : DT_ERR_DEVICE, "%s: direct fp16-form 3x3 launch failed", tag);
            if (c.amax_out) amax_note(ctx, out, (long long)a.M / (epi == EPI_POOL ? 4 : 1) * out_ld, L.cout, amax_out_slot(L));
            return DT_OK;
        }
    }
    // conv_2 / 3 / 5 (6 / 8)'s shapes: the fused F(4x4,3x3) kernel (V and M' stay on the CU) once there are enough 16x16-pixel
    // blocks to fill the chip several times over; below that the unfused forms win (few, half-empty workgroups)
    if (L.fused4s && ctx->pol.fused4 != 0 && in_ld % 4 == 0 && ((epi == EPI_PLAIN && order == ORD_LINEAR) || (epi == EPI_POOL && !((H | W) & 1)))) {
        const long long blocks = (long long)B * ((H + 15) / 16) * ((W + 15) / 16) * ((L.cout + 127) / 128);
        if (ctx->pol.fused4 == 2 || (((ctx->pol.fused4 == 1 && L.cin <= 64) || ctx->pol.fused4 == 3) && blocks >= 1024)) {
            Wino4FusedArgs f;
            memset(&f, 0, sizeof(f));
            f.in = in; f.in_bs = a.in_bs; f.in_ld = in_ld; f.B = B; f.H = H; f.W = W; f.Cin = L.cin; f.N = L.cout;
            f.u = L.fused4s; f.bias = L.bias; f.slope = slope;
            if (epi == EPI_POOL) { f.out2 = out; f.out2_ld = out_ld; }
            else { f.out = out; f.out_ld = out_ld; f.out_bs = a.out_bs; }
            // executed MFMA FLOPs: 36 positions x (whole 4x4 tiles) x Cin x N x 2; bytes: input once per 128 output channels (+ halo 27 %) and the output
            const double tiles = (double)B * ((H + 3) / 4) * ((W + 3) / 4);
            ProfScope ps(ctx, "conv_fused", 2.0 * 36.0 * tiles * L.cin * L.cout,
                         4.0 * ((double)B * H * W * L.cin * 1.27 * ((L.cout + 127) / 128) + (double)a.M * L.cout / (epi == EPI_POOL ? 4.0 : 1.0)), tag);
            prof_direct_form(ctx, flops, bytes, DF_FUSED);
            float *zeros = ws_get(ctx, "zeros256", 256, /*zero_on_grow=*/true);
            if (!zeros) return DT_ERR_DEVICE;
            if (h2_wanted(ctx) && amax_out_slot(L)) f.amax_out = amax_slot(ctx, amax_out_slot(L));
            const int rc = launch_wino4s_fused(ctx->stream, f, zeros);
            if (rc) return dt_fail(ctx, rc == 2 ? DT_ERR_ARG : DT_ERR_DEVICE, "%s: fused F(4x4) launch failed", tag);
            if (f.amax_out) amax_note(ctx, out, (long long)a.M / (epi == EPI_POOL ? 4 : 1) * out_ld, L.cout, amax_out_slot(L));
            return DT_OK;
        }
    }
    if (wino_runs(ctx, L.wino, L.wino_ts, B, H, W, L.cin, L.cout) && ((epi == EPI_PLAIN && order == ORD_LINEAR) || epi == EPI_POOL || epi == EPI_POOL_BOTH)) {
        WinoIO io;
        memset(&io, 0, sizeof(io));
        io.in = in; io.in_ld = in_ld; io.in_bs = a.in_bs;
        if (epi == EPI_POOL) { io.out2 = out; io.out2_ld = out_ld; }
        else { io.out = out; io.out_ld = out_ld; io.out_bs = a.out_bs; io.out2 = out2; io.out2_ld = out2_ld; }
        io.amax_out_slot = amax_out_slot(L);
        // F(6x6) or F(4x4) for this launch: with many tiles F(6x6)'s fewer multiplies win; with a few frames the choice
        // is about how the positions x row tiles x column tiles spread over the CUs (small_gemm_cost)
        const float *wt = L.wino;
        int ts = L.wino_ts;
        if (L.wino_alt && !ctx->pol.pin) {
            const bool pooled = io.out2 != nullptr;
            const WinoGeom q6 = wino_geometry(ctx, 6, B, H, W, pooled), q4 = wino_geometry(ctx, 4, B, H, W, pooled);
            const long long t6 = 64ll * ((q6.Mt + 127) / 128) * ((L.cout + 127) / 128);
            // (the cost model prices the fp32 MFMA kernel's tiles: where the F(6x6) launch takes the split GEMM in the fp16 form -- run_wino's own test -- F(4x4) on the
            //  fp32 kernel is no alternative: at 20 frames it cost the 13x13 layers 0.10-0.24 ms each against 0.05-0.11 on the split kernel)
            const bool s3_h2_takes_it = ctx->pol.s3 != 0 && h2_wanted(ctx) && ctx->wino_h2.find(L.wino) != ctx->wino_h2.end() && L.cin % 32 == 0 && L.cout % 128 == 0 &&
                                        L.npad % 128 == 0 && (ctx->pol.s3 == 2 || (L.cin >= ctx->pol.s3_mink && q6.Mt >= ctx->pol.s3_minrows_h2));
            if (!s3_h2_takes_it && t6 <= 4096 && small_gemm_cost(q4.Mt, L.cout, 36, nullptr) < small_gemm_cost(q6.Mt, L.cout, 64, nullptr)) { wt = L.wino_alt; ts = 4; }
        }
        return run_wino(ctx, wt, ts, L.bias, L.cin, L.cout, L.npad, B, H, W, io, slope, tag, 0, L.idx >= 1 && L.idx <= 23 ? AMAX_IN + L.idx : AMAX_TEST);
    }
    // Wave quantisation for small batches (few frames at 13x13 / 26x26): with 512 resident
    // workgroup slots (256 CUs x 2) a layer of a few hundred output tiles leaves the chip
    // partly idle or spills a nearly empty last round.  Split K over grid.y into a slab and
    // combine deterministically; the split count minimises a simple round model
    //   time(s) ~ rounds(tiles*s) / s + 0.003*s,  rounds(n) = full rounds + cost of the partial one
    // (a half-empty round still costs ~0.6 of a full one: single workgroups per CU run faster).
    prof_direct_form(ctx, flops, bytes);
    int ksplit = 1;
    if (epi == EPI_PLAIN && order == ORD_LINEAR && cfg != CFG_128x64) {
        const int tiles = ((a.M + 127) / 128) * ((L.cout + 127) / 128);
        const int nk = a.K / 32;
        if (tiles < 2 * 512) {
            int smax = nk / 6;                           // keep >= 6 chunks (192 of K) per split
            if (smax > 32) smax = 32;
            double best = 1e30;
            for (int sp = 1; sp <= (smax < 1 ? 1 : smax); ++sp) {
                const int n = tiles * sp, full = n / 512, part = n % 512;
                const double rounds = full + (part == 0 ? 0.0 : (part <= 256 ? 0.6 : 0.6 + 0.4 * (part - 256) / 256.0));
                const double t = rounds / sp + 0.003 * sp;   // + slab write/read and combine launch per split
                if (t < best - 1e-9) { best = t; ksplit = sp; }
            }
        }
        if (ctx->pol.ksplit > 0) ksplit = ctx->pol.ksplit < nk ? ctx->pol.ksplit : nk;
        if (ksplit > 1) cfg = CFG_128x128;   // the split-K path is built for the 128x128 tile
    }
    if (ksplit > 1) {
        float *slab = ws_get(ctx, "splitk", (size_t)ksplit * a.M * L.cout * sizeof(float));
        if (!slab) return DT_ERR_DEVICE;
        ConvArgs b = a;
        b.out = slab; b.out_ld = L.cout; b.ksplit = ksplit; b.bias = nullptr;
        {
            ProfScope ps(ctx, "conv_igemm", flops, bytes + 4.0 * ksplit * a.M * (double)L.cout, tag);
            const int rc = launch_igemm(ctx, b, L.ks, ORD_LINEAR, EPI_PARTIAL, CFG_128x128);
            if (rc) return dt_fail(ctx, rc == 2 ? DT_ERR_ARG : DT_ERR_DEVICE, "conv_%d split-K launch failed (rc=%d)", L.idx, rc);
        }
        ProfScope ps2(ctx, "splitk_reduce", 0.0, 4.0 * (ksplit + 1.0) * a.M * (double)L.cout, tag);
        if (launch_splitk_reduce(ctx->stream, slab, ksplit, a.M, L.cout, L.bias, slope, out, out_ld))
            return dt_fail(ctx, DT_ERR_DEVICE, "conv_%d split-K reduce launch failed", L.idx);
        return DT_OK;
    }
    ProfScope ps(ctx, "conv_igemm", flops, bytes, tag);
    const bool am_epi = epi == EPI_PLAIN || epi == EPI_POOL || epi == EPI_POOL_BOTH || epi == EPI_S2D;
    if (am_epi && h2_wanted(ctx) && amax_out_slot(L)) a.amax_out = amax_slot(ctx, amax_out_slot(L));
    const int rc = launch_igemm(ctx, a, L.ks, order, epi, cfg);
    if (rc) return dt_fail(ctx, rc == 2 ? DT_ERR_ARG : DT_ERR_DEVICE, "conv_%d launch failed (rc=%d)", L.idx, rc);
    if (a.amax_out) {      // the tensor the next layer reads: pooled where the epilogue pools, 4 N channels per quarter-resolution pixel after space_to_depth
        if (epi == EPI_POOL_BOTH) amax_note(ctx, out2, (long long)a.M / 4 * out2_ld, L.cout, amax_out_slot(L));
        else if (epi == EPI_POOL) amax_note(ctx, out, (long long)a.M / 4 * out_ld, L.cout, amax_out_slot(L));
        else if (epi == EPI_S2D) amax_note(ctx, out, (long long)a.M / 4 * out_ld, 4 * L.cout, amax_out_slot(L));
        else amax_note(ctx, out, (long long)a.M * out_ld, L.cout, amax_out_slot(L));
    }
    return DT_OK;
}

// What dt_detector_extract wants out of the graph: the output of layer `idx` in one of the forms the reference's
// layer names denote (KerasYOLO.py:279-400): conv_N = the Conv2D output, norm_N = BatchNormalization output,
// leaky_re_lu_N / conv_feat = after LeakyReLU, max_pooling2d_k = after the pool; lambda_1 = space_to_depth(conv_21
// block), concatenate_1 = [skip, main].
enum { EX_CONV = 0, EX_NORM = 1, EX_ACT = 2, EX_POOL = 3, EX_S2D = 4, EX_CAT = 5 };
struct Extract {
    int idx, kind;
    float *out;      // dense [B][h][w][C]
    bool done = false;
};

// layer `L` on its own, un-fused, into ex.out: pre-activation (slope 1) for conv_N / norm_N, then the BatchNorm
// un-folded for conv_N
static int extract_layer(dt_ctx *ctx, const ConvLayer &L, const float *in, int in_ld, int B, int h, int w, Extract &ex)
{
    if (ex.kind == EX_CONV && L.scale && L.scale_has_zero)
        return dt_fail(ctx, DT_ERR_ARG, "conv_%d: a BatchNorm gamma of 0 was folded into the kernel; the raw Conv2D output cannot be recovered", L.idx);
    int rc = run_conv(ctx, L, in, in_ld, B, h, w, ex.out, L.cout, ORD_LINEAR, EPI_PLAIN, ex.kind == EX_ACT ? LEAKY : 1.0f);
    if (rc) return rc;
    if (ex.kind == EX_CONV && L.scale && launch_unfold_bn(ctx->stream, ex.out, (long long)B * h * w, L.cout, L.scale, L.bias))
        return dt_fail(ctx, DT_ERR_DEVICE, "BatchNorm un-fold launch failed");
    ex.done = true;
    return DT_OK;
}

// conv_3 and the 1x1 conv_4 behind it as ONE launch of conv3_h2.hip (its FUSE instance: the 128-channel tensor between them never exists): where
// conv_3 would take the direct fp16-form kernel anyway, conv_4 is a 128 -> <= 64 channel 1x1 layer with fp16-form weights, and DT_C3FUSE is on
static bool conv34_fusable(const dt_ctx *ctx, const ConvLayer &L3, const ConvLayer &L4, const float *in, int in_ld, int B, int H, int W)
{
    const long long blocks = (long long)B * ((H + 15) / 16) * ((W + 15) / 16);
    return ctx->pol.c3fuse != 0 && L3.ks == 3 && L3.w3_h2 && L3.pscale_w3 && L3.cout == 128 && L4.ks == 1 && L4.cin == 128 && L4.cout <= 64 && L4.wt_h2 && L4.pscale_h2 &&
           ctx->pol.c3h2 != 0 && h2_wanted(ctx) && in_ld % 4 == 0 && (reinterpret_cast<uintptr_t>(in) & 15) == 0 && (ctx->pol.c3h2 == 2 || blocks >= 1024);
}
static int run_conv34_fused(dt_ctx *ctx, const ConvLayer &L3, const ConvLayer &L4, const float *in, int in_ld, int B, int H, int W, float *out, int out_ld, float slope)
{
    const long long M = (long long)B * H * W;
    amax_forget(ctx, out, M * out_ld);
    Conv3H2Args c;
    memset(&c, 0, sizeof(c));
    c.in = in; c.in_bs = (long long)H * W * in_ld; c.in_ld = in_ld; c.B = B; c.H = H; c.W = W; c.Cin = L3.cin; c.N = L3.cout; c.Np = L3.npad;
    c.w = L3.w3_h2; c.pscale = L3.pscale_w3; c.bias = L3.bias; c.slope = slope;
    c.w1 = L4.wt_h2; c.pscale1 = L4.pscale_h2; c.bias1 = L4.bias; c.N1 = L4.cout; c.Np1 = L4.npad; c.slope1 = slope;
    c.out = out; c.out_ld = out_ld; c.out_bs = (long long)H * W * out_ld;
    c.zeros = ws_get(ctx, "zeros256", 256, /*zero_on_grow=*/true);
    if (!c.zeros) return DT_ERR_DEVICE;
    c.amax = ensure_amax(ctx, in, M, L3.cin, in_ld, AMAX_IN + L3.idx);
    if (!c.amax) return DT_ERR_DEVICE;
    if (amax_out_slot(L4)) c.amax_out = amax_slot(ctx, amax_out_slot(L4));
    char tag[32];
    snprintf(tag, sizeof(tag), "conv_%d", L3.idx);
    if (ctx->prof && !ctx->capturing) ctx->prof_tab["conv_direct_h2:fused_1x1"].launches += 1;
    // bytes: conv_3's input once (+ 41 % halo), conv_4's output; direct form: both layers
    ProfScope ps(ctx, "conv_direct_h2", conv3_h2_flops(c), 4.0 * ((double)M * L3.cin * 1.41 + (double)M * L4.cout), tag);
    prof_direct_form(ctx, 2.0 * M * (9.0 * L3.cin * L3.cout + (double)L4.cin * L4.cout),
                     4.0 * ((double)M * L3.cin + 9.0 * L3.cin * L3.cout + 2.0 * (double)M * L3.cout + (double)L4.cin * L4.cout + (double)M * L4.cout), DF_C3H2);
    const int rc = launch_conv3_h2(ctx->stream, c);
    if (rc) return dt_fail(ctx, rc == 2 ? DT_ERR_ARG : DT_ERR_DEVICE, "%s: fused direct 3x3 + 1x1 launch failed", tag);
    if (c.amax_out) amax_note(ctx, out, M * out_ld, L4.cout, amax_out_slot(L4));
    return DT_OK;
}

// conv_2 .. conv_21 on the library-owned buffers (bufA holds conv_1's pooled output).  With `ex` the walk stops at
// the requested layer and writes it to ex->out instead.
static int run_trunk(dt_ctx *ctx, int B, float *bufA, float *bufB, float *skip, float *cat, Extract *ex)
{
    const int H = ctx->image_h, W = ctx->image_w;
    float *cur = bufA, *nxt = bufB;
    int h = H / 2, w = W / 2;
    int rc;
    bool cat20 = false;
    for (int li = 1; li < 20; ++li) {   // conv_2 .. conv_20
        const int idx = TRUNK[li][0], pool = TRUNK[li][4];
        const ConvLayer &L = ctx->layers[idx];
        if (ex && ex->idx == idx && ex->kind <= EX_ACT) return extract_layer(ctx, L, cur, L.cin, B, h, w, *ex);
        if (idx == 3 && !ex && TRUNK[li + 1][0] == 4 && conv34_fusable(ctx, L, ctx->layers[4], cur, L.cin, B, h, w)) {
            // conv_3 + conv_4 in one launch: conv_4's output goes to `nxt`, one swap, conv_4's turn of the walk is skipped
            rc = run_conv34_fused(ctx, L, ctx->layers[4], cur, L.cin, B, h, w, nxt, ctx->layers[4].cout, LEAKY);
            if (rc) return rc;
            float *t = cur; cur = nxt; nxt = t;
            ++li;                      // conv_4 is done
            continue;
        }
        if (idx == 13) {   // skip tapped before the pool (KerasYOLO.py:347)
            rc = run_conv(ctx, L, cur, L.cin, B, h, w, skip, 512, ORD_QUAD, EPI_POOL_BOTH, LEAKY, nxt, 512);
        } else if (idx == 20) {   // writes channels [256,1280) of the concat buffer (KerasYOLO.py:391)
            rc = run_conv(ctx, L, cur, L.cin, B, h, w, cat + 256, 1280, ORD_LINEAR, EPI_PLAIN, LEAKY);
            for (const auto &t : ctx->amax_tag) cat20 |= t.lo == cat + 256 && t.slot == 20;      // conv_20's epilogue measured its 1024 channels into slot 20
        } else if (pool) {
            rc = run_conv(ctx, L, cur, L.cin, B, h, w, nxt, L.cout, ORD_QUAD, EPI_POOL, LEAKY);
        } else {
            rc = run_conv(ctx, L, cur, L.cin, B, h, w, nxt, L.cout, ORD_LINEAR, EPI_PLAIN, LEAKY);
        }
        if (rc) return rc;
        if (pool) { h /= 2; w /= 2; }
        if (ex && ex->idx == idx && ex->kind == EX_POOL) {
            HIP_TRY(ctx, hipMemcpyAsync(ex->out, nxt, (size_t)B * h * w * L.cout * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
            ex->done = true;
            return DT_OK;
        }
        if (ex && ex->idx == 21 && idx == 13 && ex->kind != EX_CAT) break;   // conv_21's block only needs the skip tensor
        float *t = cur; cur = nxt; nxt = t;
    }
    if (ex && ex->idx == 21 && ex->kind <= EX_ACT)
        return extract_layer(ctx, ctx->layers[21], skip, 512, B, H / 16, W / 16, *ex);
    if (ex && ex->idx == 21 && ex->kind == EX_S2D) {
        ex->done = true;
        return run_conv(ctx, ctx->layers[21], skip, 512, B, H / 16, W / 16, ex->out, 256, ORD_QUAD, EPI_S2D, LEAKY);
    }
    // conv_21 on the skip tensor + tf.space_to_depth(2) -> channels [0,256) (KerasYOLO.py:386-391)
    rc = run_conv(ctx, ctx->layers[21], skip, 512, B, H / 16, W / 16, cat, 1280, ORD_QUAD, EPI_S2D, LEAKY);
    if (rc) return rc;
    if (cat20 && h2_wanted(ctx)) {      // ... and the 256 channels conv_21's fp32 kernel wrote are added to the same slot: the concat tensor's max |x| (conv_22 reads it)
        const long long rows = (long long)B * (H / 32) * (W / 32);
        ProfScope ps(ctx, "absmax", 0.0, 4.0 * (double)rows * 256, "cat_skip");
        if (launch_absmax(ctx->stream, cat, rows, 256, 1280, 1, 0, amax_slot(ctx, 20), /*zero=*/false)) return dt_fail(ctx, DT_ERR_DEVICE, "absmax launch failed");
        amax_note(ctx, cat, rows * 1280, 1280, 20);
    }
    if (ex && ex->kind == EX_CAT) {
        HIP_TRY(ctx, hipMemcpyAsync(ex->out, cat, (size_t)B * (H / 32) * (W / 32) * 1280 * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
        ex->done = true;
    }
    return DT_OK;
}

// Runs conv_1 .. conv_23.  feat/netout destinations may alias caller buffers.
static int detect_internal(dt_ctx *ctx, const void *frames, int dtype, int B, Dest feat, Dest netout, bool skip23 = false)
{
    if (!ctx->det_loaded) return dt_fail(ctx, DT_ERR_STATE, "detector weights not loaded");
    if (B <= 0) return dt_fail(ctx, DT_ERR_ARG, "batch must be positive");
    if (dtype != DT_FRAMES_U8 && dtype != DT_FRAMES_F32) return dt_fail(ctx, DT_ERR_ARG, "bad frames dtype");
    const int H = ctx->image_h, W = ctx->image_w;
    if ((long long)B * (H / 2) * (W / 2) >= (1ll << 31)) return dt_fail(ctx, DT_ERR_ARG, "batch too large");
    const size_t per_frame = (size_t)(H / 2) * (W / 2) * 32;   // largest activation (floats)
    float *bufA = ws_get(ctx, "actA", per_frame * B * sizeof(float));
    float *bufB = ws_get(ctx, "actB", per_frame * B * sizeof(float));
    float *skip = ws_get(ctx, "skip", (size_t)B * (H / 16) * (W / 16) * 512 * sizeof(float));
    float *cat = ws_get(ctx, "cat", (size_t)B * (H / 32) * (W / 32) * 1280 * sizeof(float));
    if (!bufA || !bufB || !skip || !cat) return DT_ERR_DEVICE;
    ctx->last_batch = B;
    ctx->h2_small = B < ctx->pol.h2_minframes;
    {   // dt_detector_tap may only hand out 'feat' / 'netout' if THIS forward wrote the library-owned workspaces
        auto owned = [&](const char *name, const float *p) {
            auto it = ctx->ws.find(name);
            return it != ctx->ws.end() && it->second.p == static_cast<const void *>(p);
        };
        ctx->tap_feat = owned("feat", feat.p) && feat.ld == 1024;
        ctx->tap_netout = owned("netout", netout.p) && netout.ld == ctx->cb;
    }

    {   // conv_1 + norm_1 + leaky + pool, with x/255 fused
        const bool c1s3 = ctx->pol.s3 != 0 && ctx->pol.s3_conv1 != 0;
        const double c1_bytes = (double)B * H * W * 3.0 * (dtype == DT_FRAMES_U8 ? 1 : 4) + 4.0 * B * (H / 2) * (W / 2) * 32.0;
        // EXECUTED MFMA FLOPs: K = 27 padded to 32 (bf16: 3 partial products per multiply for uint8 frames, 6 for float32 frames) or to 28 (fp32 MFMA)
        const double c1_exec = c1s3 ? (dtype == DT_FRAMES_U8 ? 3.0 : 6.0) * 2.0 * B * H * W * 32.0 * 32.0 : 2.0 * B * H * W * 28.0 * 32.0;
        ProfScope ps(ctx, "conv1_direct", c1_exec, c1_bytes, c1s3 ? "bf16" : "f32");
        prof_direct_form(ctx, 2.0 * B * H * W * 27.0 * 32.0, c1_bytes, DF_CONV1);
        amax_forget(ctx, bufA, (long long)per_frame * B);
        if (int rcz = amax_begin(ctx)) return rcz;
        // (its epilogue takes max |x| of what it writes: conv_2's direct fp16-form kernel scales its input by it)
        // (conv_1 publishes only where conv_2 will read it: the direct fp16-form kernel)
        const bool c2_direct = ctx->pol.c3h2 == 2 || (ctx->pol.c3h2 != 0 && (long long)B * ((H / 2 + 15) / 16) * ((W / 2 + 15) / 16) >= c3h2_min_blocks(ctx, 32));
        unsigned *am1 = c1s3 && h2_wanted(ctx) && c2_direct && conv1_direct_fills_amax(frames, dtype, W, ctx->conv1_w3, ctx->conv1_w3u8) ? amax_slot(ctx, 1) : nullptr;
        if (launch_conv1_direct(ctx->stream, frames, dtype, B, H, W, ctx->conv1_w, ctx->conv1_b, ctx->lut255, LEAKY,
                                bufA, c1s3 ? ctx->conv1_w3 : nullptr, c1s3 ? ctx->conv1_w3u8 : nullptr, am1))
            return dt_fail(ctx, DT_ERR_DEVICE, "conv_1 launch failed");
        if (am1) amax_note(ctx, bufA, (long long)B * (H / 2) * (W / 2) * 32, 32, 1);
    }
    const int h = H / 32, w = W / 32;
    int rc = graphed(ctx, "trunk:" + std::to_string(B), [&]() -> int {   // conv_2 .. conv_21: library-owned buffers only
        return run_trunk(ctx, B, bufA, bufB, skip, cat, nullptr);
    }, bufA);
    if (rc) return rc;
    // conv_22 -> 'conv_feat'
    rc = run_conv(ctx, ctx->layers[22], cat, 1280, B, h, w, feat.p, feat.ld, ORD_LINEAR, EPI_PLAIN, LEAKY);
    if (rc) return rc;
    // conv_23 (bias, linear) -- unless nobody reads it: the tracker's merged input projection (build_merged_xproj) carries it in its weights
    if (skip23) { ctx->tap_netout = false; return DT_OK; }
    rc = run_conv(ctx, ctx->layers[23], feat.p, feat.ld, B, h, w, netout.p, netout.ld, ORD_LINEAR, EPI_PLAIN, 1.0f);
    return rc;
}

extern "C" int dt_detect_forward(dt_ctx *ctx, const void *d_frames, int frames_dtype, int batch, float *d_netout,
                                 float *d_feat)
{
    if (!ctx || !d_frames) return dt_fail(ctx, DT_ERR_ARG, "null argument");
    amax_reset(ctx);
    const int G2 = (ctx->image_h / 32) * (ctx->image_w / 32);
    Dest feat{d_feat, 1024}, net{d_netout, ctx->cb};
    if (!d_feat) {
        feat.p = ws_get(ctx, "feat", (size_t)batch * G2 * 1024 * sizeof(float));
        if (!feat.p) return DT_ERR_DEVICE;
    }
    if (!d_netout) {
        net.p = ws_get(ctx, "netout", (size_t)batch * G2 * ctx->cb * sizeof(float));
        if (!net.p) return DT_ERR_DEVICE;
    }
    return detect_internal(ctx, d_frames, frames_dtype, batch, feat, net);
}

extern "C" int dt_detector_tap(dt_ctx *ctx, const char *name, int batch, float *d_out)
{
    if (!ctx || !name || !d_out) return dt_fail(ctx, DT_ERR_ARG, "null argument");
    if (batch != ctx->last_batch) return dt_fail(ctx, DT_ERR_STATE, "tap batch %d != last forward batch %d", batch, ctx->last_batch);
    const int H = ctx->image_h, W = ctx->image_w;
    const char *wsname = nullptr;
    size_t n = 0;
    if (!strcmp(name, "act_13")) { wsname = "skip"; n = (size_t)batch * (H / 16) * (W / 16) * 512; }
    else if (!strcmp(name, "conv_feat")) { wsname = "feat"; n = (size_t)batch * (H / 32) * (W / 32) * 1024; }
    else if (!strcmp(name, "conv_23")) { wsname = "netout"; n = (size_t)batch * (H / 32) * (W / 32) * ctx->cb; }
    else return dt_fail(ctx, DT_ERR_ARG, "unknown tap '%s'", name);
    auto it = ctx->ws.find(wsname);
    const bool stale = (!strcmp(wsname, "feat") && !ctx->tap_feat) || (!strcmp(wsname, "netout") && !ctx->tap_netout);
    if (it == ctx->ws.end() || it->second.bytes < n * sizeof(float) || stale)
        return dt_fail(ctx, DT_ERR_STATE, "tap '%s' not materialised by the last forward (it wrote caller-owned outputs; "
                       "call dt_detect_forward with d_netout = d_feat = NULL first)", name);
    HIP_TRY(ctx, hipMemcpyAsync(d_out, it->second.p, n * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
    return DT_OK;
}

// layer name -> (layer index, form, output geometry).  Names are the reference's (KerasYOLO.py:279-400) plus the
// names Keras gives its unnamed layers in a fresh session (leaky_re_lu_N, max_pooling2d_k, lambda_1, concatenate_1).
static bool parse_layer_name(const dt_ctx *ctx, const char *name, int *idx, int *kind, int *oh, int *ow, int *oc)
{
    const int H = ctx->image_h, W = ctx->image_w;
    int n = 0;
    char tail = 0;
    auto geom = [&](int layer, bool pooled) {   // output geometry of conv block `layer` (1..23), before / after its pool
        int div = 1, c = 0;
        for (auto &t : TRUNK) {
            if (t[0] == layer) { c = t[3]; if (pooled) div *= 2; break; }
            if (t[4]) div *= 2;
            if (t[0] == 20) break;
        }
        if (layer == 21) { div = 16; c = 64; }
        if (layer == 22) { div = 32; c = 1024; }
        if (layer == 23) { div = 32; c = ctx->cb; }
        *oh = H / div; *ow = W / div; *oc = c;
    };
    if (sscanf(name, "conv_%d%c", &n, &tail) == 1 && n >= 1 && n <= 23) { *idx = n; *kind = EX_CONV; geom(n, false); return true; }
    if (sscanf(name, "norm_%d%c", &n, &tail) == 1 && n >= 1 && n <= 22) { *idx = n; *kind = EX_NORM; geom(n, false); return true; }
    if ((sscanf(name, "leaky_re_lu_%d%c", &n, &tail) == 1 || sscanf(name, "act_%d%c", &n, &tail) == 1) && n >= 1 && n <= 22) {
        *idx = n; *kind = EX_ACT; geom(n, false); return true;
    }
    if (!strcmp(name, "conv_feat")) { *idx = 22; *kind = EX_ACT; geom(22, false); return true; }
    if (sscanf(name, "max_pooling2d_%d%c", &n, &tail) == 1 && n >= 1 && n <= 5) {
        static const int pooled_after[5] = {1, 2, 5, 8, 13};
        *idx = pooled_after[n - 1]; *kind = EX_POOL; geom(*idx, true); return true;
    }
    if (!strcmp(name, "lambda_1")) { *idx = 21; *kind = EX_S2D; *oh = H / 32; *ow = W / 32; *oc = 256; return true; }
    if (!strcmp(name, "concatenate_1")) { *idx = 21; *kind = EX_CAT; *oh = H / 32; *ow = W / 32; *oc = 1280; return true; }
    if (!strcmp(name, "reshape_1") || !strcmp(name, "lambda_2")) { *idx = 23; *kind = EX_CONV; geom(23, false); return true; }
    return false;
}

extern "C" int dt_detector_extract(dt_ctx *ctx, const void *d_frames, int frames_dtype, int batch, const char *layer,
                                   float *d_out, size_t out_floats, int *shape4)
{
    if (ctx) amax_reset(ctx);
    if (!ctx || !layer) return dt_fail(ctx, DT_ERR_ARG, "null argument");
    if (!ctx->cb) return dt_fail(ctx, DT_ERR_STATE, "dt_detector_config must be called first");
    int idx = 0, kind = 0, oh = 0, ow = 0, oc = 0;
    if (!parse_layer_name(ctx, layer, &idx, &kind, &oh, &ow, &oc)) return dt_fail(ctx, DT_ERR_ARG, "No such layer: %s", layer);
    if (shape4) { shape4[0] = batch; shape4[1] = oh; shape4[2] = ow; shape4[3] = oc; }
    if (!d_out) return DT_OK;   // shape query
    if (!d_frames || batch <= 0) return dt_fail(ctx, DT_ERR_ARG, "bad frames / batch");
    if (!ctx->det_loaded) return dt_fail(ctx, DT_ERR_STATE, "detector weights not loaded");
    if (frames_dtype != DT_FRAMES_U8 && frames_dtype != DT_FRAMES_F32) return dt_fail(ctx, DT_ERR_ARG, "bad frames dtype");
    const size_t need = (size_t)batch * oh * ow * oc;
    if (out_floats < need) return dt_fail(ctx, DT_ERR_ARG, "output buffer holds %zu floats, layer %s needs %zu", out_floats, layer, need);
    const int B = batch, H = ctx->image_h, W = ctx->image_w;
    Extract ex{idx, kind, d_out};
    if (idx == 1 && kind <= EX_ACT) {   // conv_1 un-pooled: as a Cin = 32 layer of the MFMA kernel on zero-padded channels
        float *x32 = ws_get(ctx, "extract_x32", (size_t)B * H * W * 32 * sizeof(float));
        if (!x32) return DT_ERR_DEVICE;
        if (launch_expand_rgb32(ctx->stream, d_frames, frames_dtype, (long long)B * H * W, ctx->lut255, x32))
            return dt_fail(ctx, DT_ERR_DEVICE, "channel expansion launch failed");
        int rc = load_conv_layer(ctx, 0, 3, 32, 32, ctx->conv1_hwio32.data(), ctx->conv1_scale.data(), ctx->conv1_shift.data());
        if (rc) return rc;
        ctx->layers[0].idx = 1;
        return extract_layer(ctx, ctx->layers[0], x32, 32, B, H, W, ex);
    }
    if (idx <= 21 || kind == EX_CAT) {
        const size_t per_frame = (size_t)(H / 2) * (W / 2) * 32;
        float *bufA = ws_get(ctx, "actA", per_frame * B * sizeof(float));
        float *bufB = ws_get(ctx, "actB", per_frame * B * sizeof(float));
        float *skip = ws_get(ctx, "skip", (size_t)B * (H / 16) * (W / 16) * 512 * sizeof(float));
        float *cat = ws_get(ctx, "cat", (size_t)B * (H / 32) * (W / 32) * 1280 * sizeof(float));
        if (!bufA || !bufB || !skip || !cat) return DT_ERR_DEVICE;
        ctx->last_batch = 0;            // the tap workspaces no longer hold a complete forward
        ctx->tap_feat = ctx->tap_netout = false;
        const bool c1s3 = ctx->pol.s3 != 0 && ctx->pol.s3_conv1 != 0;
        if (int rcz = amax_begin(ctx)) return rcz;
        if (launch_conv1_direct(ctx->stream, d_frames, frames_dtype, B, H, W, ctx->conv1_w, ctx->conv1_b, ctx->lut255, LEAKY, bufA,
                                c1s3 ? ctx->conv1_w3 : nullptr, c1s3 ? ctx->conv1_w3u8 : nullptr))
            return dt_fail(ctx, DT_ERR_DEVICE, "conv_1 launch failed");
        if (idx == 1) {   // max_pooling2d_1
            HIP_TRY(ctx, hipMemcpyAsync(d_out, bufA, need * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
            return DT_OK;
        }
        int rc = run_trunk(ctx, B, bufA, bufB, skip, cat, &ex);
        if (rc) return rc;
        return ex.done ? DT_OK : dt_fail(ctx, DT_ERR_ARG, "layer %s was not reached", layer);
    }
    // conv_22 / conv_23 blocks: a whole forward into the library-owned buffers, then the last layer(s) un-fused
    const int G2 = (H / 32) * (W / 32);
    float *feat = ws_get(ctx, "feat", (size_t)B * G2 * 1024 * sizeof(float));
    float *net = ws_get(ctx, "netout", (size_t)B * G2 * ctx->cb * sizeof(float));
    if (!feat || !net) return DT_ERR_DEVICE;
    int rc = detect_internal(ctx, d_frames, frames_dtype, B, Dest{feat, 1024}, Dest{net, ctx->cb});
    if (rc) return rc;
    if (idx == 23) {
        HIP_TRY(ctx, hipMemcpyAsync(d_out, net, need * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
        return DT_OK;
    }
    float *cat = static_cast<float *>(ctx->ws["cat"].p);
    return extract_layer(ctx, ctx->layers[22], cat, 1280, B, H / 32, W / 32, ex);
}

// ---------------------------------------------------------------------------
// frame ingest
// ---------------------------------------------------------------------------
extern "C" int dt_ingest_resize(dt_ctx *ctx, const uint8_t *d_src, int n, int src_h, int src_w, uint8_t *d_dst,
                                int dst_h, int dst_w)
{
    if (!ctx || !d_src || !d_dst) return dt_fail(ctx, DT_ERR_ARG, "null argument");
    if (n <= 0 || src_h <= 0 || src_w <= 0 || dst_h <= 0 || dst_w <= 0 || n > 65535 || dst_h > 65535)
        return dt_fail(ctx, DT_ERR_ARG, "bad ingest shape");
    int *tabs = reinterpret_cast<int *>(ws_get(ctx, "ingest_tabs", sizeof(int) * 4 * ((size_t)dst_w + dst_h)));
    if (!tabs) return DT_ERR_DEVICE;
    const int key[4] = {src_h, src_w, dst_h, dst_w};
    if (memcmp(key, ctx->ing_key, sizeof(key)) != 0) {
        std::vector<int> h(4 * ((size_t)dst_w + dst_h));
        ingest_tables(src_w, dst_w, h.data());
        ingest_tables(src_h, dst_h, h.data() + 4 * (size_t)dst_w);
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        HIP_TRY(ctx, hipMemcpy(tabs, h.data(), h.size() * sizeof(int), hipMemcpyHostToDevice));
        memcpy(ctx->ing_key, key, sizeof(key));
    }
    ProfScope ps(ctx, "ingest", 0.0, 3.0 * n * ((double)src_h * src_w + (double)dst_h * dst_w));
    if (launch_ingest_resize(ctx->stream, d_src, n, src_h, src_w, d_dst, dst_h, dst_w, tabs, tabs + 4 * (size_t)dst_w))
        return dt_fail(ctx, DT_ERR_DEVICE, "ingest launch failed");
    return DT_OK;
}

// ---------------------------------------------------------------------------
// decode / iou / associate
// ---------------------------------------------------------------------------
static int decode_common(dt_ctx *ctx, const float *d_netout, int batch, int GH, int GW, int NB, int NC,
                         float obj_threshold, float nms_threshold, const float *d_frame_thr, const float *h_anchors,
                         int cap, float *d_boxes, int *d_counts, float *d_classes, float *d_post)
{
    if (!ctx || !d_netout || !d_boxes || !d_counts || !h_anchors) return dt_fail(ctx, DT_ERR_ARG, "null argument");
    if (batch <= 0 || cap <= 0 || NB <= 0 || NB > 32 || NC <= 0) return dt_fail(ctx, DT_ERR_ARG, "bad decode shape");
    float *anch = ws_get(ctx, "dec_anchors", 64 * sizeof(float));
    if (!anch) return DT_ERR_DEVICE;
    // the caller's host buffer may be a temporary: stage it in the context before the (stream-ordered) upload
    if (memcmp(ctx->dec_anchors_host, h_anchors, sizeof(float) * 2 * NB) != 0 || ctx->dec_anchors_n != 2 * NB) {
        memcpy(ctx->dec_anchors_host, h_anchors, sizeof(float) * 2 * NB);
        ctx->dec_anchors_n = 2 * NB;
        HIP_TRY(ctx, hipMemcpyAsync(anch, ctx->dec_anchors_host, sizeof(float) * 2 * NB, hipMemcpyHostToDevice, ctx->stream));
    }
    const size_t fsz = (size_t)GH * GW * NB * (5 + NC);
    float *post = d_post;
    if (!post) {
        post = ws_get(ctx, "dec_post", (size_t)batch * fsz * sizeof(float));
        if (!post) return DT_ERR_DEVICE;
    }
    float *scratch = nullptr;      // grids above 1920 cells: the kernel's per-candidate arrays live in global memory
    if (const size_t sf = decode_scratch_floats(GH, GW, NB)) {
        scratch = ws_get(ctx, "dec_scratch", (size_t)batch * sf * sizeof(float));
        if (!scratch) return DT_ERR_DEVICE;
    }
    ProfScope ps(ctx, "decode_nms", 0.0, 4.0 * 3.0 * batch * (double)fsz);
    const int rc = launch_decode(ctx->stream, d_netout, (long long)fsz, batch, GH, GW, NB, NC, obj_threshold,
                                 nms_threshold, anch, cap, d_boxes, d_counts, d_classes, post, d_frame_thr, scratch);
    if (rc == 2)
        return dt_fail(ctx, DT_ERR_ARG, "decode: grid %dx%dx%d (limit 8192 cells) / %d classes exceeds the kernel's limits", GH, GW,
                       NB, NC);
    if (rc) return dt_fail(ctx, DT_ERR_DEVICE, "decode launch failed");
    return DT_OK;
}

extern "C" int dt_decode(dt_ctx *ctx, const float *d_netout, int batch, int GH, int GW, int NB, int NC,
                         float obj_threshold, float nms_threshold, const float *h_anchors, int cap, float *d_boxes,
                         int *d_counts, float *d_classes, float *d_post)
{
    return decode_common(ctx, d_netout, batch, GH, GW, NB, NC, obj_threshold, nms_threshold, nullptr, h_anchors, cap,
                         d_boxes, d_counts, d_classes, d_post);
}

extern "C" int dt_decode_per_frame(dt_ctx *ctx, const float *d_netout, int batch, int GH, int GW, int NB, int NC,
                                   const float *d_thresholds, const float *h_anchors, int cap, float *d_boxes,
                                   int *d_counts, float *d_classes, float *d_post)
{
    if (!d_thresholds) return dt_fail(ctx, DT_ERR_ARG, "null thresholds");
    return decode_common(ctx, d_netout, batch, GH, GW, NB, NC, 0.0f, 0.0f, d_thresholds, h_anchors, cap, d_boxes,
                         d_counts, d_classes, d_post);
}

extern "C" int dt_bbox_iou(dt_ctx *ctx, const float *d_pairs, int n, float *d_iou)
{
    if (!ctx || !d_pairs || !d_iou) return dt_fail(ctx, DT_ERR_ARG, "null argument");
    if (launch_bbox_iou(ctx->stream, d_pairs, n, d_iou)) return dt_fail(ctx, DT_ERR_DEVICE, "bbox_iou launch failed");
    return DT_OK;
}

extern "C" int dt_associate(dt_ctx *ctx, const float *d_boxes, const int *d_counts, int n_clips, int T, int cap,
                            float assoc_threshold, int *d_ids, int *d_nids)
{
    if (!ctx || !d_boxes || !d_counts || !d_ids || !d_nids) return dt_fail(ctx, DT_ERR_ARG, "null argument");
    ProfScope ps(ctx, "associate", 0.0, 4.0 * n_clips * (double)T * cap * 9.0);
    const int rc = launch_associate(ctx->stream, d_boxes, d_counts, n_clips, T, cap, assoc_threshold, d_ids, d_nids);
    if (rc) return dt_fail(ctx, rc == 2 ? DT_ERR_ARG : DT_ERR_DEVICE, "associate launch failed");
    return DT_OK;
}

// ---------------------------------------------------------------------------
// tracker head (ConvLSTM2D + 1x1)
// ---------------------------------------------------------------------------
static void gate_interleave_map(int U, std::vector<int> &n_map)
{
    // packed column n' = (j/32)*128 + g*32 + j%32  <-  Keras column g*U + j
    n_map.resize((size_t)4 * U);
    for (int np = 0; np < 4 * U; ++np) {
        const int jb = np / 128, g = (np % 128) / 32, jj = np % 32;
        n_map[np] = g * U + jb * 32 + jj;
    }
}

extern "C" int dt_tracker_load(dt_ctx *ctx, int units, const float *h_kernel, const float *h_recurrent,
                               const float *h_bias, const float *h_out_kernel, const float *h_out_bias)
{
    if (!ctx || !h_kernel || !h_recurrent || !h_bias || !h_out_kernel || !h_out_bias)
        return dt_fail(ctx, DT_ERR_ARG, "null argument");
    if (!ctx->cb) return dt_fail(ctx, DT_ERR_STATE, "dt_detector_config must be called first");
    if (units <= 0 || units % 32) return dt_fail(ctx, DT_ERR_ARG, "units must be a positive multiple of 32");
    const int U = units, Cb = ctx->cb, Csrc = Cb + 1024;
    const int Cx = round_up(1024 + Cb, 32);   // device z layout: [conv_feat 1024 | x_bbox Cb | zero pad]
    std::vector<int> cin_map(Cx, -1), n_map;
    for (int c = 0; c < 1024; ++c) cin_map[c] = Cb + c;          // Keras order: x_bbox first, then x_vis
    for (int c = 0; c < Cb; ++c) cin_map[1024 + c] = c;
    gate_interleave_map(U, n_map);
    std::vector<float> wx((size_t)4 * U * 9 * Cx), wh((size_t)4 * U * 9 * U), bx((size_t)4 * U);
    pack_conv_weights(h_kernel, 3, Csrc, 4 * U, cin_map.data(), Cx, n_map.data(), 4 * U, nullptr, wx.data());
    pack_conv_weights(h_recurrent, 3, U, 4 * U, nullptr, U, n_map.data(), 4 * U, nullptr, wh.data());
    for (int np = 0; np < 4 * U; ++np) bx[np] = h_bias[n_map[np]];
    const int npad = round_up(Cb, 256);
    std::vector<float> wo((size_t)npad * U), bo(npad, 0.0f);
    pack_conv_weights(h_out_kernel, 1, U, Cb, nullptr, U, nullptr, npad, nullptr, wo.data());
    for (int c = 0; c < Cb; ++c) bo[c] = h_out_bias[c];
    int rc;
    if ((rc = upload(ctx, &ctx->trk_wx, wx))) return rc;
    if ((rc = upload(ctx, &ctx->trk_wh, wh))) return rc;
    if ((rc = upload(ctx, &ctx->trk_bx, bx))) return rc;
    if ((rc = upload(ctx, &ctx->trk_wo, wo))) return rc;
    if ((rc = upload(ctx, &ctx->trk_bo, bo))) return rc;
    for (float **w : {&ctx->trk_wx_wino, &ctx->trk_wh_wino})
        if (*w) { (void)hipStreamSynchronize(ctx->stream); s3_drop(ctx, *w); (void)hipFree(*w); *w = nullptr; }
    ctx->trk_wino_ts = wino_tile(ctx, false);
    ctx->trk_wh_ts = wino_tile(ctx, true);
    if (wino_wanted(ctx, 3, Cx, 4 * U) &&
        (rc = upload_wino(ctx, &ctx->trk_wx_wino, ctx->trk_wino_ts, h_kernel, Csrc, 4 * U, cin_map.data(), Cx, n_map.data(),
                          4 * U, nullptr, ctx->trk_wino_ts == 6)))
        return rc;
    if (wino_wanted(ctx, 3, U, 4 * U) &&
        (rc = upload_wino(ctx, &ctx->trk_wh_wino, ctx->trk_wh_ts, h_recurrent, U, 4 * U, nullptr, U, n_map.data(), 4 * U,
                          nullptr, ctx->trk_wh_ts == 4)))
        return rc;
    ctx->trk_units = U; ctx->trk_cx = Cx; ctx->trk_wo_npad = npad;
    ctx->trk_hkernel.assign(h_kernel, h_kernel + (size_t)9 * Csrc * 4 * U);
    ctx->trk_hbias.assign(h_bias, h_bias + (size_t)4 * U);
    graphs_clear(ctx);
    ctx->trk_loaded = true;
    return build_merged_xproj(ctx);
}

// does the input projection of F frames take the merged form (conv_23 folded in)?  The same predicate decides whether dt_track_forward
// may skip conv_23 when the caller does not ask for the detector's grid.
static bool xproj_merged(const dt_ctx *ctx, int F, int gh, int gw)
{
    return ctx->pol.trk_merge && ctx->trk_wxm_wino && ctx->trk_bx16 && ctx->trk_wino_ts == 6 &&
           wino_runs(ctx, ctx->trk_wxm_wino, 6, F, gh, gw, 1024, 4 * ctx->trk_units);
}

// xproj = conv3x3(z, Wx) + b for all frames; then the sequential recurrence.
//   z != null, xproj_ext == null : both halves, xproj in the library's workspace (dt_track_forward / dt_track_recurrent)
//   z != null, xproj_ext != null, hseq == null : the input projection ONLY, into the caller's buffer (dt_track_detect_xproj:
//       the projection does not depend on the recurrence, so a frame-sharded deployment runs it where the frame is)
//   z == null, xproj_ext != null : the recurrence ONLY, on the caller's stitched projection rows (dt_track_recurrent_xproj)
static int convlstm_sequence(dt_ctx *ctx, const float *z, int Cx, int n_clips, int T, int gh, int gw, int U,
                             const float *wx, const float *bx, const float *wh, float *hseq /*[n_clips][T][GG][U]*/,
                             const float *wx_wino = nullptr, const float *wh_wino = nullptr, float *xproj_ext = nullptr)
{
    const int GG = gh * gw, F = n_clips * T, N4 = 4 * U;
    ctx->h2_small = F < ctx->pol.h2_minframes;
    float *xproj = xproj_ext ? xproj_ext : ws_get(ctx, "trk_xproj", (size_t)F * GG * N4 * sizeof(float));
    float *cst = hseq ? ws_get(ctx, "trk_c", (size_t)n_clips * GG * U * sizeof(float)) : nullptr;
    if (!xproj || (hseq && !cst)) return DT_ERR_DEVICE;
    // hseq, xproj and the cell state are library-owned, z only when it is the 'trk_z' workspace (dt_track_forward).  A
    // captured graph keeps the pointers it was captured with, so the input projection -- the one part that reads z --
    // is inside the replayed graph only for the library-owned z; for a caller's rows (dt_track_recurrent) it runs
    // plainly and the recurrence alone (3 launches per step, library-owned buffers only) replays, under its own key.
    bool z_owned = false;
    {
        auto it = ctx->ws.find("trk_z");
        z_owned = it != ctx->ws.end() && it->second.p == static_cast<const void *>(z);
    }
    const std::string shape = std::to_string(n_clips) + "x" + std::to_string(T);
    auto input_projection = [&]() -> int {
    if (z_owned && xproj_merged(ctx, F, gh, gw) && wx_wino == ctx->trk_wx_wino) {
        // conv_23 folded into the projection (build_merged_xproj): the 1024 conv_feat channels of z only, border-aware bias -- for the library's OWN z
        // rows only (dt_track_forward, dt_track_detect_xproj): a caller's rows (dt_track_recurrent) may carry any x_bbox and get the two-step form
        WinoIO io;
        memset(&io, 0, sizeof(io));
        io.in = z; io.in_ld = Cx; io.in_bs = (long long)GG * Cx;
        io.out = xproj; io.out_ld = N4; io.out_bs = (long long)GG * N4;
        io.bias16 = ctx->trk_bx16 + N4;         // corrections of the 16 border cases; row 0 of the table is the interior bias
        if (ctx->prof && !ctx->capturing) ctx->prof_tab["convlstm_xproj:merged_conv23"].launches += 1;
        const int rc = run_wino(ctx, ctx->trk_wxm_wino, 6, ctx->trk_bx16, 1024, N4, N4, F, gh, gw, io, 1.0f, "convlstm_xproj", ctx->cb + 1024, AMAX_TRK);
        if (rc) return rc;
    } else if (wino_runs(ctx, wx_wino, ctx->trk_wino_ts, F, gh, gw, Cx, N4)) {
        WinoIO io;
        memset(&io, 0, sizeof(io));
        io.in = z; io.in_ld = Cx; io.in_bs = (long long)GG * Cx;
        io.out = xproj; io.out_ld = N4; io.out_bs = (long long)GG * N4;
        const int rc = run_wino(ctx, wx_wino, ctx->trk_wino_ts, bx, Cx, N4, N4, F, gh, gw, io, 1.0f, "convlstm_xproj", ctx->cb + 1024, AMAX_TRK);
        if (rc) return rc;
    } else {
        ConvArgs a;
        memset(&a, 0, sizeof(a));
        a.in = z; a.in_ld = Cx; a.in_bs = (long long)GG * Cx;
        a.wt = wx; a.bias = bx;
        a.out = xproj; a.out_ld = N4; a.out_bs = (long long)GG * N4;
        a.B = F; a.H = gh; a.W = gw; a.Cin = Cx; a.N = N4; a.M = F * GG; a.K = 9 * Cx;
        a.slope = 1.0f;
        ProfScope ps(ctx, "conv_igemm", 2.0 * a.M * 9.0 * (ctx->cb + 1024) * N4,
                     4.0 * ((double)a.M * Cx + (double)a.K * N4 + (double)a.M * N4), "convlstm_xproj");
        a.npad = N4;
        prof_direct_form(ctx, 2.0 * a.M * 9.0 * (ctx->cb + 1024) * N4, 4.0 * ((double)a.M * Cx + (double)a.K * N4 + (double)a.M * N4));
        if (launch_igemm(ctx, a, 3, ORD_LINEAR, EPI_PLAIN, pick_cfg(a.M, N4, 3)))
            return dt_fail(ctx, DT_ERR_DEVICE, "ConvLSTM input projection launch failed");
    }
    return DT_OK;
    };
    auto recurrence = [&]() -> int {
    const long long xp_bs = (long long)T * GG * N4, h_bs = (long long)T * GG * U, c_bs = (long long)GG * U;
    {   // t = 0: h_{-1} = c_{-1} = 0
        ProfScope ps(ctx, "convlstm_gates", 0.0, 4.0 * n_clips * GG * (3.0 * U + 2.0 * U));
        if (launch_convlstm_gates_only(ctx->stream, xproj, xp_bs, N4, cst, c_bs, U, hseq, h_bs, U, n_clips, GG, U))
            return dt_fail(ctx, DT_ERR_DEVICE, "ConvLSTM t=0 launch failed");
    }
    for (int t = 1; t < T; ++t) {
        if (wino_runs(ctx, wh_wino, ctx->trk_wh_ts, n_clips, gh, gw, U, N4)) {
            WinoIO io;
            memset(&io, 0, sizeof(io));
            io.in = hseq + (long long)(t - 1) * GG * U; io.in_ld = U; io.in_bs = h_bs;
            io.out = hseq + (long long)t * GG * U; io.out_ld = U; io.out_bs = h_bs;
            io.xproj = xproj + (long long)t * GG * N4; io.xp_ld = N4; io.xp_bs = xp_bs;
            io.cstate = cst; io.c_ld = U; io.c_bs = c_bs;
            // (h_{t-1} = o * tanh(c) lies in (-1, 1): the fp16 form's scale is static, nothing is measured)
            const int rc = run_wino(ctx, wh_wino, ctx->trk_wh_ts, nullptr, U, N4, N4, n_clips, gh, gw, io, 1.0f, "convlstm_step", 0, AMAX_ONE);
            if (rc) return rc;
            continue;
        }
        ConvArgs a;
        memset(&a, 0, sizeof(a));
        a.in = hseq + (long long)(t - 1) * GG * U; a.in_ld = U; a.in_bs = h_bs;
        a.wt = wh; a.bias = nullptr;
        a.out = hseq + (long long)t * GG * U; a.out_ld = U; a.out_bs = h_bs;
        a.xproj = xproj + (long long)t * GG * N4; a.xp_ld = N4; a.xp_bs = xp_bs;
        a.cstate = cst; a.c_ld = U; a.c_bs = c_bs;
        a.B = n_clips; a.H = gh; a.W = gw; a.Cin = U; a.N = N4; a.M = n_clips * GG; a.K = 9 * U;
        a.slope = 1.0f;
        ProfScope ps(ctx, "conv_igemm", 2.0 * a.M * (double)a.K * N4,
                     4.0 * ((double)a.M * U + (double)a.K * N4 + (double)a.M * N4 + 3.0 * a.M * U), "convlstm_step");
        a.npad = N4;
        prof_direct_form(ctx, 2.0 * a.M * (double)a.K * N4, 4.0 * ((double)a.M * U + (double)a.K * N4 + (double)a.M * N4 + 3.0 * a.M * U));
        if (launch_igemm(ctx, a, 3, ORD_LINEAR, EPI_GATES, pick_cfg(a.M, N4, 3)))
            return dt_fail(ctx, DT_ERR_DEVICE, "ConvLSTM step launch failed");
    }
    return DT_OK;
    };
    if (xproj_ext) {       // caller-owned projection rows: nothing here may be baked into a replayed graph
        if (z) { const int rc = input_projection(); if (rc || !hseq) return rc; }
        return hseq ? recurrence() : DT_OK;
    }
    if (z_owned)
        return graphed(ctx, "clstm:" + shape, [&]() -> int {
            const int rc = input_projection();
            return rc ? rc : recurrence();
        }, z);
    const int rc = input_projection();
    return rc ? rc : graphed(ctx, "clstm_steps:" + shape, recurrence);
}

// the recurrent head on z [n_clips][T][G*G][Cx] (library- or caller-owned): ConvLSTM2D over T, then tconv_2
static int track_recurrent_internal(dt_ctx *ctx, const float *z, int n_clips, int T, float *d_trk)
{
    const int gh = ctx->image_h / 32, gw = ctx->image_w / 32, GG = gh * gw;
    const int F = n_clips * T, U = ctx->trk_units, Cx = ctx->trk_cx, Cb = ctx->cb;
    float *hseq = ws_get(ctx, "trk_h", (size_t)F * GG * U * sizeof(float));
    if (!hseq) return DT_ERR_DEVICE;
    int rc = convlstm_sequence(ctx, z, Cx, n_clips, T, gh, gw, U, ctx->trk_wx, ctx->trk_bx, ctx->trk_wh, hseq, ctx->trk_wx_wino,
                               ctx->trk_wh_wino);
    if (rc) return rc;
    float *trk = d_trk;
    if (!trk) {
        trk = ws_get(ctx, "trk_out", (size_t)F * GG * Cb * sizeof(float));
        if (!trk) return DT_ERR_DEVICE;
    }
    // TimeDistributed(Conv2D(Cb,(1,1)))  'tconv_2'  (MultiObjDetTracker.py:182)
    ConvLayer L;
    L.idx = 102; L.ks = 1; L.cin = U; L.cout = Cb; L.npad = ctx->trk_wo_npad; L.wt = ctx->trk_wo; L.bias = ctx->trk_bo;
    return run_conv(ctx, L, hseq, U, F, gh, gw, trk, Cb, ORD_LINEAR, EPI_PLAIN, 1.0f);
}

extern "C" int dt_track_forward(dt_ctx *ctx, const void *d_frames, int frames_dtype, int n_clips, int T,
                                float *d_trk, float *d_det)
{
    if (ctx) amax_reset(ctx);
    if (!ctx || !d_frames) return dt_fail(ctx, DT_ERR_ARG, "null argument");
    if (!ctx->trk_loaded) return dt_fail(ctx, DT_ERR_STATE, "tracker weights not loaded");
    if (n_clips <= 0 || T <= 0) return dt_fail(ctx, DT_ERR_ARG, "n_clips and T must be positive");
    const int gh = ctx->image_h / 32, gw = ctx->image_w / 32, GG = gh * gw;
    const int F = n_clips * T, Cx = ctx->trk_cx, Cb = ctx->cb;
    float *z = ws_get(ctx, "trk_z", (size_t)F * GG * Cx * sizeof(float), /*zero_on_grow=*/true);
    if (!z) return DT_ERR_DEVICE;
    int rc = detect_internal(ctx, d_frames, frames_dtype, F, Dest{z, Cx}, Dest{z + 1024, Cx}, /*skip23=*/!d_det && xproj_merged(ctx, F, gh, gw));
    if (rc) return rc;
    rc = track_recurrent_internal(ctx, z, n_clips, T, d_trk);
    if (rc) return rc;
    if (d_det) {
        ProfScope ps(ctx, "misc", 0.0, 8.0 * F * GG * (double)Cb);
        if (launch_copy_cols(ctx->stream, z + 1024, Cx, d_det, Cb, (long long)F * GG, Cb))
            return dt_fail(ctx, DT_ERR_DEVICE, "detection copy launch failed");
    }
    return DT_OK;
}

// The two halves of dt_track_forward for a FRAME-sharded deployment (BASELINE.json configs[4]):
// every rank runs the detector on its share of the frames and emits their z rows; the rows are exchanged (RCCL
// all-gather, 13*13*1120*4 = 757 KB per frame at 416); the owner of a clip runs the recurrence on the stitched rows.
extern "C" int dt_track_row_width(dt_ctx *ctx)
{
    return (ctx && ctx->trk_loaded) ? ctx->trk_cx : 0;
}

extern "C" int dt_track_detect(dt_ctx *ctx, const void *d_frames, int frames_dtype, int n_frames, float *d_z)
{
    if (ctx) amax_reset(ctx);
    if (!ctx || !d_frames || !d_z) return dt_fail(ctx, DT_ERR_ARG, "null argument");
    if (!ctx->trk_loaded) return dt_fail(ctx, DT_ERR_STATE, "tracker weights not loaded");
    if (n_frames <= 0) return dt_fail(ctx, DT_ERR_ARG, "n_frames must be positive");
    const int GG = (ctx->image_h / 32) * (ctx->image_w / 32), Cx = ctx->trk_cx, Cb = ctx->cb;
    if (Cx > 1024 + Cb)   // zero the pad columns [1024+Cb, Cx) of the caller's rows (the ConvLSTM kernel reads them)
        HIP_TRY(ctx, hipMemset2DAsync(d_z + 1024 + Cb, (size_t)Cx * sizeof(float), 0, (size_t)(Cx - 1024 - Cb) * sizeof(float),
                                      (size_t)n_frames * GG, ctx->stream));
    return detect_internal(ctx, d_frames, frames_dtype, n_frames, Dest{d_z, Cx}, Dest{d_z + 1024, Cx});
}

extern "C" int dt_track_recurrent(dt_ctx *ctx, const float *d_z, int n_clips, int T, float *d_trk, float *d_det)
{
    if (ctx) amax_reset(ctx);
    if (!ctx || !d_z) return dt_fail(ctx, DT_ERR_ARG, "null argument");
    if (!ctx->trk_loaded) return dt_fail(ctx, DT_ERR_STATE, "tracker weights not loaded");
    if (n_clips <= 0 || T <= 0) return dt_fail(ctx, DT_ERR_ARG, "n_clips and T must be positive");
    int rc = track_recurrent_internal(ctx, d_z, n_clips, T, d_trk);
    if (rc) return rc;
    if (d_det) {
        const int GG = (ctx->image_h / 32) * (ctx->image_w / 32);
        if (launch_copy_cols(ctx->stream, d_z + 1024, ctx->trk_cx, d_det, ctx->cb, (long long)n_clips * T * GG, ctx->cb))
            return dt_fail(ctx, DT_ERR_DEVICE, "detection copy launch failed");
    }
    return DT_OK;
}

// The same split one step later in the graph: the ConvLSTM2D INPUT projection W * x_t + b (MultiObjDetTracker.py:176; 55 % of
// the recurrent head's FLOPs) does not depend on the recurrence, so the rank that ran the detector on a frame runs it too and
// the rows that travel are xproj rows [G, G, 4U] (1.38 MB per frame at 416x416); the clip's owner is left with the sequential part only.
extern "C" int dt_track_xproj_width(dt_ctx *ctx)
{
    return (ctx && ctx->trk_loaded) ? 4 * ctx->trk_units : 0;
}

extern "C" int dt_track_detect_xproj(dt_ctx *ctx, const void *d_frames, int frames_dtype, int n_frames, float *d_xp, float *d_det)
{
    if (ctx) amax_reset(ctx);
    if (!ctx || !d_frames || !d_xp) return dt_fail(ctx, DT_ERR_ARG, "null argument");
    if (!ctx->trk_loaded) return dt_fail(ctx, DT_ERR_STATE, "tracker weights not loaded");
    if (n_frames <= 0) return dt_fail(ctx, DT_ERR_ARG, "n_frames must be positive");
    const int gh = ctx->image_h / 32, gw = ctx->image_w / 32, GG = gh * gw;
    const int Cx = ctx->trk_cx, Cb = ctx->cb;
    float *z = ws_get(ctx, "trk_z", (size_t)n_frames * GG * Cx * sizeof(float), /*zero_on_grow=*/true);
    if (!z) return DT_ERR_DEVICE;
    int rc = detect_internal(ctx, d_frames, frames_dtype, n_frames, Dest{z, Cx}, Dest{z + 1024, Cx}, /*skip23=*/!d_det && xproj_merged(ctx, n_frames, gh, gw));
    if (rc) return rc;
    rc = convlstm_sequence(ctx, z, Cx, n_frames, 1, gh, gw, ctx->trk_units, ctx->trk_wx, ctx->trk_bx, ctx->trk_wh, nullptr,
                           ctx->trk_wx_wino, ctx->trk_wh_wino, d_xp);
    if (rc) return rc;
    if (d_det && launch_copy_cols(ctx->stream, z + 1024, Cx, d_det, Cb, (long long)n_frames * GG, Cb))
        return dt_fail(ctx, DT_ERR_DEVICE, "detection copy launch failed");
    return DT_OK;
}

extern "C" int dt_track_recurrent_xproj(dt_ctx *ctx, const float *d_xp, int n_clips, int T, float *d_trk)
{
    if (ctx) amax_reset(ctx);
    if (!ctx || !d_xp) return dt_fail(ctx, DT_ERR_ARG, "null argument");
    if (!ctx->trk_loaded) return dt_fail(ctx, DT_ERR_STATE, "tracker weights not loaded");
    if (n_clips <= 0 || T <= 0) return dt_fail(ctx, DT_ERR_ARG, "n_clips and T must be positive");
    const int gh = ctx->image_h / 32, gw = ctx->image_w / 32, GG = gh * gw;
    const int F = n_clips * T, U = ctx->trk_units, Cb = ctx->cb;
    float *hseq = ws_get(ctx, "trk_h", (size_t)F * GG * U * sizeof(float));
    if (!hseq) return DT_ERR_DEVICE;
    int rc = convlstm_sequence(ctx, nullptr, ctx->trk_cx, n_clips, T, gh, gw, U, ctx->trk_wx, ctx->trk_bx, ctx->trk_wh, hseq,
                               ctx->trk_wx_wino, ctx->trk_wh_wino, const_cast<float *>(d_xp));
    if (rc) return rc;
    float *trk = d_trk;
    if (!trk) {
        trk = ws_get(ctx, "trk_out", (size_t)F * GG * Cb * sizeof(float));
        if (!trk) return DT_ERR_DEVICE;
    }
    ConvLayer L;       // TimeDistributed(Conv2D(Cb,(1,1)))  'tconv_2'  (MultiObjDetTracker.py:182)
    L.idx = 102; L.ks = 1; L.cin = U; L.cout = Cb; L.npad = ctx->trk_wo_npad; L.wt = ctx->trk_wo; L.bias = ctx->trk_bo;
    return run_conv(ctx, L, hseq, U, F, gh, gw, trk, Cb, ORD_LINEAR, EPI_PLAIN, 1.0f);
}

// ---------------------------------------------------------------------------
// TinyTracker
// ---------------------------------------------------------------------------
extern "C" int dt_tiny_load(dt_ctx *ctx, int D, int units, int out_dim, const float *h_kernel,
                            const float *h_recurrent, const float *h_bias, const float *h_dense_kernel,
                            const float *h_dense_bias)
{
    if (!ctx || !h_kernel || !h_recurrent || !h_bias || !h_dense_kernel || !h_dense_bias)
        return dt_fail(ctx, DT_ERR_ARG, "null argument");
    if (units != 512) return dt_fail(ctx, DT_ERR_ARG, "LSTM units must be 512 (config.json:19)");
    if (D < 8 || out_dim < 1) return dt_fail(ctx, DT_ERR_ARG, "bad D / out_dim");
    const int U = units, Dp = round_up(D, 32), N4 = 4 * U;
    // x.W as a 1x1 "convolution": kernel [D,4U] is HWIO with k=1
    std::vector<float> wx((size_t)N4 * Dp), bx(h_bias, h_bias + N4);
    pack_conv_weights(h_kernel, 1, D, N4, nullptr, Dp, nullptr, N4, nullptr, wx.data());
    // recurrent [U,4U] (k, g*U+j) -> [j][g][k]
    std::vector<float> ur((size_t)U * N4);
    for (int j = 0; j < U; ++j)
        for (int g = 0; g < 4; ++g)
            for (int k = 0; k < U; ++k) ur[((size_t)j * 4 + g) * U + k] = h_recurrent[(size_t)k * N4 + g * U + j];
    // Dense head: O <= 8 (TinyTracker, 4) keeps the [U][O] matrix for the wavefront-reduction
    // kernel; wider heads (TinyHeatmapTracker, 32*32) run as a 1x1 MFMA GEMM with a sigmoid epilogue
    const int O = out_dim, Opad = round_up(O, 256);
    std::vector<float> wd, bd;
    if (O <= 8) {
        wd.assign(h_dense_kernel, h_dense_kernel + (size_t)U * O);
        bd.assign(h_dense_bias, h_dense_bias + O);
    } else {
        wd.resize((size_t)Opad * U);
        pack_conv_weights(h_dense_kernel, 1, U, O, nullptr, U, nullptr, Opad, nullptr, wd.data());
        bd.assign(Opad, 0.0f);
        for (int o = 0; o < O; ++o) bd[o] = h_dense_bias[o];
    }
    int rc;
    if ((rc = upload(ctx, &ctx->tiny_wx, wx))) return rc;
    if ((rc = upload(ctx, &ctx->tiny_bx, bx))) return rc;
    if ((rc = upload(ctx, &ctx->tiny_ur, ur))) return rc;
    if ((rc = upload(ctx, &ctx->tiny_wd, wd))) return rc;
    if ((rc = upload(ctx, &ctx->tiny_bd, bd))) return rc;
    ctx->tiny_D = D; ctx->tiny_Dpad = Dp; ctx->tiny_U = U; ctx->tiny_O = O; ctx->tiny_Opad = Opad;
    graphs_clear(ctx);
    ctx->tiny_loaded = true;
    return DT_OK;
}

extern "C" int dt_tiny_features(dt_ctx *ctx, const float *d_feat, const float *d_det, int n_rows, int fh, int fw,
                                int fc, int pool, float *d_x)
{
    if (!ctx || !d_feat || !d_det || !d_x) return dt_fail(ctx, DT_ERR_ARG, "null argument");
    if (!ctx->tiny_loaded) return dt_fail(ctx, DT_ERR_STATE, "TinyTracker weights not loaded");
    const int D = ctx->tiny_D;
    const int fdim = pool == 0 ? fc : (fh / 4) * (fw / 4) * fc;
    const int ddim = D - fdim;   // 4 (TinyTracker box) or heatmap_size^2 (TinyHeatmapTracker.py:31)
    if (ddim < 1) return dt_fail(ctx, DT_ERR_ARG, "pooled feature width %d leaves no room for the detection input (D %d)", fdim, D);
    // GlobalMaxPooling2D / MaxPooling2D(4,4)+Flatten, then concatenate([x, det]) (TinyTracker.py:29-34)
    ProfScope ps(ctx, "pool", 0.0, 4.0 * n_rows * ((double)fh * fw * fc + fdim));
    int rc = pool == 0 ? launch_global_maxpool(ctx->stream, d_feat, n_rows, fh * fw, fc, d_x, D)
                       : launch_maxpool4_flatten(ctx->stream, d_feat, n_rows, fh, fw, fc, d_x, D);
    if (rc) return dt_fail(ctx, rc == 2 ? DT_ERR_ARG : DT_ERR_DEVICE, "pool launch failed");
    if (launch_copy_cols(ctx->stream, d_det, ddim, d_x + fdim, D, n_rows, ddim))
        return dt_fail(ctx, DT_ERR_DEVICE, "det concat launch failed");
    return DT_OK;
}

extern "C" int dt_tiny_sequence(dt_ctx *ctx, const float *d_x, int n_seq, int T, float *d_out)
{
    if (!ctx || !d_x || !d_out) return dt_fail(ctx, DT_ERR_ARG, "null argument");
    if (!ctx->tiny_loaded) return dt_fail(ctx, DT_ERR_STATE, "TinyTracker weights not loaded");
    if (n_seq <= 0 || T <= 0) return dt_fail(ctx, DT_ERR_ARG, "n_seq and T must be positive");
    const int U = ctx->tiny_U, D = ctx->tiny_D, Dp = ctx->tiny_Dpad, N4 = 4 * U;
    const int R = n_seq * T;
    float *x = ws_get(ctx, "tiny_x", (size_t)R * Dp * sizeof(float), /*zero_on_grow=*/true);
    float *xproj = ws_get(ctx, "tiny_xproj", (size_t)R * N4 * sizeof(float));
    float *hseq = ws_get(ctx, "tiny_h", (size_t)R * U * sizeof(float));
    float *cst = ws_get(ctx, "tiny_c", (size_t)n_seq * U * sizeof(float));
    if (!x || !xproj || !hseq || !cst) return DT_ERR_DEVICE;
    if (launch_copy_cols(ctx->stream, d_x, D, x, Dp, R, D))   // K padded to a multiple of 32 (pad columns stay 0)
        return dt_fail(ctx, DT_ERR_DEVICE, "x staging launch failed");
    // staged x, xproj, h and c are library-owned: the projection and the T launch-bound steps replay as a graph
    int grc = graphed(ctx, "lstm:" + std::to_string(n_seq) + "x" + std::to_string(T), [&]() -> int {
    {   // x.W + b for every (sequence, t) at once on the matrix cores
        ConvArgs a;
        memset(&a, 0, sizeof(a));
        a.in = x; a.in_ld = Dp; a.in_bs = Dp;
        a.wt = ctx->tiny_wx; a.bias = ctx->tiny_bx;
        a.out = xproj; a.out_ld = N4; a.out_bs = N4;
        a.B = R; a.H = 1; a.W = 1; a.Cin = Dp; a.N = N4; a.M = R; a.K = Dp;
        a.slope = 1.0f;
        ProfScope ps(ctx, "conv_igemm", 2.0 * R * (double)D * N4, 4.0 * ((double)R * Dp + (double)Dp * N4 + (double)R * N4),
                     "lstm_xproj");
        if (launch_igemm(ctx, a, 1, ORD_LINEAR, EPI_PLAIN, CFG_128x128))
            return dt_fail(ctx, DT_ERR_DEVICE, "LSTM input projection launch failed");
    }
    const long long xp_bs = (long long)T * N4, h_bs = (long long)T * U;
    for (int t = 0; t < T; ++t) {
        ProfScope ps(ctx, "lstm_step", 2.0 * n_seq * (double)U * N4, 4.0 * ((double)U * N4 + n_seq * (6.0 * U + N4)));
        int rc;
        if (t == 0)
            rc = launch_lstm_step0(ctx->stream, xproj, xp_bs, cst, hseq, h_bs, n_seq, U);
        else
            rc = launch_lstm_step(ctx->stream, xproj + (long long)t * N4, xp_bs, hseq + (long long)(t - 1) * U, h_bs,
                                  cst, ctx->tiny_ur, hseq + (long long)t * U, h_bs, n_seq, U);
        if (rc) return dt_fail(ctx, DT_ERR_DEVICE, "LSTM step launch failed");
    }
    return DT_OK;
    });
    if (grc) return grc;
    const int O = ctx->tiny_O;
    if (O <= 8) {
        ProfScope ps(ctx, "misc", 2.0 * R * U * (double)O, 4.0 * R * (U + (double)O));
        if (launch_dense_sigmoid(ctx->stream, hseq, U, ctx->tiny_wd, ctx->tiny_bd, R, U, O, d_out, O))
            return dt_fail(ctx, DT_ERR_DEVICE, "Dense launch failed");
    } else {   // TimeDistributed(Dense(heatmap_size^2, sigmoid))  (TinyHeatmapTracker.py:43)
        ConvArgs a;
        memset(&a, 0, sizeof(a));
        a.in = hseq; a.in_ld = U; a.in_bs = U;
        a.wt = ctx->tiny_wd; a.bias = ctx->tiny_bd;
        a.out = d_out; a.out_ld = O; a.out_bs = O;
        a.B = R; a.H = 1; a.W = 1; a.Cin = U; a.N = O; a.M = R; a.K = U;
        a.slope = 1.0f; a.act = 1;
        ProfScope ps(ctx, "conv_igemm", 2.0 * R * (double)U * O, 4.0 * ((double)R * U + (double)U * O + (double)R * O), "dense_head");
        if (launch_igemm(ctx, a, 1, ORD_LINEAR, EPI_PLAIN, CFG_128x128))
            return dt_fail(ctx, DT_ERR_DEVICE, "Dense head launch failed");
    }
    return DT_OK;
}

extern "C" int dt_tiny_forward(dt_ctx *ctx, const float *d_feat, const float *d_det, int n_seq, int T, int fh, int fw,
                               int fc, int pool, float *d_out)
{
    if (!ctx) return DT_ERR_ARG;
    if (!ctx->tiny_loaded) return dt_fail(ctx, DT_ERR_STATE, "TinyTracker weights not loaded");
    float *rows = ws_get(ctx, "tiny_rows", (size_t)n_seq * T * ctx->tiny_D * sizeof(float));
    if (!rows) return DT_ERR_DEVICE;
    int rc = dt_tiny_features(ctx, d_feat, d_det, n_seq * T, fh, fw, fc, pool, rows);
    if (rc) return rc;
    return dt_tiny_sequence(ctx, rows, n_seq, T, d_out);
}

// generate_heatmap_feat (utility/utils.py:53-58) for n centre-format boxes and
// generate_rectangle_from_heatmap (utility/utils.py:61-79)
extern "C" int dt_heatmap_from_boxes(dt_ctx *ctx, const float *d_box4, int n, int hmap_size, float *d_heat)
{
    if (!ctx || !d_box4 || !d_heat || hmap_size < 1) return dt_fail(ctx, DT_ERR_ARG, "bad argument");
    if (launch_heatmap_from_boxes(ctx->stream, d_box4, nullptr, n, hmap_size, d_heat))
        return dt_fail(ctx, DT_ERR_DEVICE, "heatmap launch failed");
    return DT_OK;
}

extern "C" int dt_heatmap_from_xywh64(dt_ctx *ctx, const double *d_xywh, int n, int hmap_size, float *d_heat)
{
    if (!ctx || !d_xywh || !d_heat || hmap_size < 1) return dt_fail(ctx, DT_ERR_ARG, "bad argument");
    if (launch_heatmap_from_boxes(ctx->stream, nullptr, d_xywh, n, hmap_size, d_heat))
        return dt_fail(ctx, DT_ERR_DEVICE, "heatmap launch failed");
    return DT_OK;
}

extern "C" int dt_encode_targets(dt_ctx *ctx, const int *d_objs, const int *d_counts, const int *d_dims,
                                 const double *d_aug, int n_frames, int cap, int grid_h, int grid_w, int nb_box,
                                 int nb_class, int image_h, int image_w, int true_box_buffer,
                                 const double *h_anchors, double *d_y, double *d_b)
{
    if (!ctx || !d_objs || !d_counts || !d_dims || !h_anchors || !d_y || !d_b || n_frames < 0)
        return dt_fail(ctx, DT_ERR_ARG, "bad argument");
    ProfScope ps(ctx, "encode_targets", 0.0,
                 8.0 * n_frames * ((double)grid_h * grid_w * nb_box * (5 + nb_class) + 4.0 * true_box_buffer) +
                     20.0 * n_frames * cap,
                 nullptr);
    const int rc = launch_encode_targets(ctx->stream, d_objs, d_counts, d_dims, d_aug, n_frames, cap, grid_h, grid_w,
                                         nb_box, nb_class, image_h, image_w, true_box_buffer, h_anchors, d_y, d_b);
    if (rc == 2) return dt_fail(ctx, DT_ERR_ARG, "encode_targets: unsupported shape (nb_box <= %d)", DT_MAX_ANCHOR_BOXES);
    if (rc) return dt_fail(ctx, DT_ERR_DEVICE, "encode_targets launch failed");
    return DT_OK;
}

extern "C" int dt_rect_from_heatmap(dt_ctx *ctx, const float *d_heat, int n, int hmap_size, float thresh, int *d_rect)
{
    if (!ctx || !d_heat || !d_rect || hmap_size < 1) return dt_fail(ctx, DT_ERR_ARG, "bad argument");
    if (launch_rect_from_heatmap(ctx->stream, d_heat, n, hmap_size, thresh, d_rect))
        return dt_fail(ctx, DT_ERR_DEVICE, "rect_from_heatmap launch failed");
    return DT_OK;
}

// detection box fed to the single-object tracker: the highest-score survivor of a frame
// as (cx, cy, w, h) in image-relative units (preprocessing.py:441-444), zeros if none
extern "C" int dt_top_box(dt_ctx *ctx, const float *d_boxes, const int *d_counts, int n_frames, int cap, float *d_out4)
{
    if (!ctx || !d_boxes || !d_counts || !d_out4) return dt_fail(ctx, DT_ERR_ARG, "null argument");
    if (launch_top_box(ctx->stream, d_boxes, d_counts, n_frames, cap, d_out4))
        return dt_fail(ctx, DT_ERR_DEVICE, "top_box launch failed");
    return DT_OK;
}

// ---------------------------------------------------------------------------
// layer-level entry points for the parity tests
// ---------------------------------------------------------------------------
extern "C" int dt_conv2d(dt_ctx *ctx, const float *d_in, int B, int H, int W, int Cin, const float *h_kernel, int k,
                         int Cout, const float *h_bias, float leaky_slope, int pool, float *d_out, float *d_out2)
{
    if (ctx) amax_reset(ctx);
    if (!ctx || !d_in || !h_kernel || !d_out) return dt_fail(ctx, DT_ERR_ARG, "null argument");
    if (Cin % 32) return dt_fail(ctx, DT_ERR_ARG, "Cin must be a multiple of 32");
    if (k != 1 && k != 3) return dt_fail(ctx, DT_ERR_ARG, "kernel size must be 1 or 3");
    if (pool && ((H | W) & 1)) return dt_fail(ctx, DT_ERR_ARG, "pooling needs even H and W");
    policy_from_env(ctx->pol);   // test entry point: the parity tests force policies on a live context through the environment
    std::vector<float> zero(Cout, 0.0f);
    int rc = load_conv_layer(ctx, 0, k, Cin, Cout, h_kernel, nullptr, h_bias ? h_bias : zero.data());
    if (rc) return rc;
    const ConvLayer &L = ctx->layers[0];
    switch (pool) {
    case 0: return run_conv(ctx, L, d_in, Cin, B, H, W, d_out, Cout, ORD_LINEAR, EPI_PLAIN, leaky_slope);
    case 1: return run_conv(ctx, L, d_in, Cin, B, H, W, d_out, Cout, ORD_QUAD, EPI_POOL, leaky_slope);
    case 2:
        if (!d_out2 || k != 3) return dt_fail(ctx, DT_ERR_ARG, "pool=2 needs d_out2 and k=3");
        return run_conv(ctx, L, d_in, Cin, B, H, W, d_out, Cout, ORD_QUAD, EPI_POOL_BOTH, leaky_slope, d_out2, Cout);
    case 3:
        if (k != 1) return dt_fail(ctx, DT_ERR_ARG, "space_to_depth epilogue needs k=1");
        return run_conv(ctx, L, d_in, Cin, B, H, W, d_out, 4 * Cout, ORD_QUAD, EPI_S2D, leaky_slope);
    default: return dt_fail(ctx, DT_ERR_ARG, "bad pool mode");
    }
}

namespace {
struct DevTemps {   // device temporaries of a test entry point: freed on every return path
    std::vector<float *> p;
    hipStream_t st;
    explicit DevTemps(hipStream_t s) : st(s) {}
    ~DevTemps()
    {
        (void)hipStreamSynchronize(st);
        for (float *q : p)
            if (q) (void)hipFree(q);
    }
    float **add() { p.push_back(nullptr); return &p.back(); }
};
}   // namespace

extern "C" int dt_convlstm_step(dt_ctx *ctx, const float *d_x, int B, int H, int W, int Cx, const float *d_h,
                                const float *d_c, int U, const float *h_kernel, const float *h_recurrent,
                                const float *h_bias, float *d_h_out, float *d_c_out)
{
    if (ctx) amax_reset(ctx);
    if (!ctx || !d_x || !d_h || !d_c || !h_kernel || !h_recurrent || !h_bias || !d_h_out || !d_c_out)
        return dt_fail(ctx, DT_ERR_ARG, "null argument");
    if (Cx % 32 || U % 32) return dt_fail(ctx, DT_ERR_ARG, "Cx and U must be multiples of 32");
    policy_from_env(ctx->pol);   // test entry point (see dt_conv2d)
    const int N4 = 4 * U, GG = H * W;
    std::vector<int> n_map;
    gate_interleave_map(U, n_map);
    std::vector<float> wx((size_t)N4 * 9 * Cx), wh((size_t)N4 * 9 * U), bx(N4);
    pack_conv_weights(h_kernel, 3, Cx, N4, nullptr, Cx, n_map.data(), N4, nullptr, wx.data());
    pack_conv_weights(h_recurrent, 3, U, N4, nullptr, U, n_map.data(), N4, nullptr, wh.data());
    for (int np = 0; np < N4; ++np) bx[np] = h_bias[n_map[np]];
    DevTemps tmp(ctx->stream);
    tmp.p.reserve(8);
    float **dwx = tmp.add(), **dwh = tmp.add(), **dbx = tmp.add();
    int rc;
    if ((rc = upload(ctx, dwx, wx)) || (rc = upload(ctx, dwh, wh)) || (rc = upload(ctx, dbx, bx))) return rc;
    float *xproj = ws_get(ctx, "cl_xproj", (size_t)B * GG * N4 * sizeof(float));
    if (!xproj) return DT_ERR_DEVICE;
    if (ctx->pol.wino == 2 && wino_wanted(ctx, 3, Cx, N4) && wino_wanted(ctx, 3, U, N4)) {   // the same step through the Winograd path
        float **uwx = tmp.add(), **uwh = tmp.add();
        const int ts = wino_tile(ctx, false);
        if ((rc = upload_wino(ctx, uwx, ts, h_kernel, Cx, N4, nullptr, Cx, n_map.data(), N4, nullptr, ts == 6)) ||
            (rc = upload_wino(ctx, uwh, ts, h_recurrent, U, N4, nullptr, U, n_map.data(), N4, nullptr, ts == 4)))
            return rc;
        WinoIO io;
        memset(&io, 0, sizeof(io));
        io.in = d_x; io.in_ld = Cx; io.in_bs = (long long)GG * Cx;
        io.out = xproj; io.out_ld = N4; io.out_bs = (long long)GG * N4;
        struct Twins {      // the split twins of the two temporaries go with them
            dt_ctx *c; float **a, **b;
            ~Twins() { (void)hipStreamSynchronize(c->stream); s3_drop(c, *a); s3_drop(c, *b); }
        } twins{ctx, uwx, uwh};
        if ((rc = run_wino(ctx, *uwx, ts, *dbx, Cx, N4, N4, B, H, W, io, 1.0f, "convlstm_xproj", 0, AMAX_TEST))) return rc;
        HIP_TRY(ctx, hipMemcpyAsync(d_c_out, d_c, (size_t)B * GG * U * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
        memset(&io, 0, sizeof(io));
        io.in = d_h; io.in_ld = U; io.in_bs = (long long)GG * U;
        io.out = d_h_out; io.out_ld = U; io.out_bs = (long long)GG * U;
        io.xproj = xproj; io.xp_ld = N4; io.xp_bs = (long long)GG * N4;
        io.cstate = d_c_out; io.c_ld = U; io.c_bs = (long long)GG * U;
        return run_wino(ctx, *uwh, ts, nullptr, U, N4, N4, B, H, W, io, 1.0f, "convlstm_step", 0, AMAX_TEST + 1);      // (a caller's h: measured)
    }
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.in = d_x; a.in_ld = Cx; a.in_bs = (long long)GG * Cx;
    a.wt = *dwx; a.bias = *dbx;
    a.out = xproj; a.out_ld = N4; a.out_bs = (long long)GG * N4;
    a.B = B; a.H = H; a.W = W; a.Cin = Cx; a.N = N4; a.M = B * GG; a.K = 9 * Cx; a.slope = 1.0f;
    if (launch_igemm(ctx, a, 3, ORD_LINEAR, EPI_PLAIN, CFG_128x128))
        return dt_fail(ctx, DT_ERR_DEVICE, "xproj launch failed");
    HIP_TRY(ctx, hipMemcpyAsync(d_c_out, d_c, (size_t)B * GG * U * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
    memset(&a, 0, sizeof(a));
    a.in = d_h; a.in_ld = U; a.in_bs = (long long)GG * U;
    a.wt = *dwh;
    a.out = d_h_out; a.out_ld = U; a.out_bs = (long long)GG * U;
    a.xproj = xproj; a.xp_ld = N4; a.xp_bs = (long long)GG * N4;
    a.cstate = d_c_out; a.c_ld = U; a.c_bs = (long long)GG * U;
    a.B = B; a.H = H; a.W = W; a.Cin = U; a.N = N4; a.M = B * GG; a.K = 9 * U; a.slope = 1.0f;
    if (launch_igemm(ctx, a, 3, ORD_LINEAR, EPI_GATES, CFG_128x128))
        return dt_fail(ctx, DT_ERR_DEVICE, "gates launch failed");
    return DT_OK;
}

// wino_gemm_s3.hip on caller data (parity tests at the benched shapes): both operands padded to the kernel's row tiles, split by the
// production pack kernels -- nt = 3: three bf16 terms; nt = 2: two fp16 terms of the scaled operands (V by ONE power of two from its
// max |x| like an activation, U per plane like the weights) -- then the production launcher.
extern "C" int dt_gemm_split(dt_ctx *ctx, const float *d_v, const float *d_u, int P, int Mt, int K, int N, int half, int nt, float *d_m)
{
    if (!ctx || !d_v || !d_u || !d_m) return dt_fail(ctx, DT_ERR_ARG, "null argument");
    amax_reset(ctx);
    if (P <= 0 || Mt <= 0 || K <= 0 || N <= 0 || K % 32 || N % 128 || !wino_gemm_s3_usable(Mt, K, N) || (nt != 2 && nt != 3) || (nt == 2 && P > 64))
        return dt_fail(ctx, DT_ERR_ARG, "dt_gemm_split: unsupported shape P=%d Mt=%d K=%d N=%d nt=%d", P, Mt, K, N, nt);
    const bool rows_form = half == 2;      // the 1x1 layers' form: the kernel reads d_v as fp32 rows and splits its fragments itself
    if (rows_form && P != 1) return dt_fail(ctx, DT_ERR_ARG, "dt_gemm_split: the fp32-rows form is one GEMM (P = 1)");
    if (rows_form) half = 0;
    const size_t Mp = ((size_t)Mt + 255) / 256 * 256, Np = ((size_t)N + 255) / 256 * 256;
    DevTemps tmp(ctx->stream);
    tmp.p.reserve(5);
    float **vpad = tmp.add(), **upad = tmp.add(), **vs = tmp.add(), **us = tmp.add(), **ps = tmp.add();
    HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(vpad), (size_t)P * Mp * K * sizeof(float)));
    HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(upad), (size_t)P * Np * K * sizeof(float)));
    HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(vs), (size_t)P * nt * Mp * K * sizeof(unsigned short)));
    HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(us), (size_t)P * nt * Np * K * sizeof(unsigned short)));
    HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(ps), 64 * sizeof(float)));
    HIP_TRY(ctx, hipMemsetAsync(*vpad, 0, (size_t)P * Mp * K * sizeof(float), ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(*upad, 0, (size_t)P * Np * K * sizeof(float), ctx->stream));
    HIP_TRY(ctx, hipMemcpy2DAsync(*vpad, Mp * K * sizeof(float), d_v, (size_t)Mt * K * sizeof(float), (size_t)Mt * K * sizeof(float), P,
                                  hipMemcpyDeviceToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpy2DAsync(*upad, Np * K * sizeof(float), d_u, (size_t)N * K * sizeof(float), (size_t)N * K * sizeof(float), P,
                                  hipMemcpyDeviceToDevice, ctx->stream));
    GemmS3Args g;
    memset(&g, 0, sizeof(g));
    if (nt == 2) {
        unsigned *vslot = amax_slot(ctx, AMAX_TEST);
        if (launch_wino_h2_pack(ctx->stream, *vpad, P, (int)Mp, K, 0, vslot, reinterpret_cast<unsigned short *>(*vs), nullptr) ||
            launch_wino_h2_pack(ctx->stream, *upad, P, (int)Np, K, 0, amax_slot(ctx, AMAX_PACK), reinterpret_cast<unsigned short *>(*us), *ps))
            return dt_fail(ctx, DT_ERR_DEVICE, "fp16-form pack launch failed");
        g.nt = 2; g.pscale = *ps; g.amax = vslot;      // (the fp32-rows form measures the same tensor: the padding rows are zeros)
    } else if (launch_wino_s3_pack(ctx->stream, *vpad, P, (int)Mp, K, reinterpret_cast<unsigned short *>(*vs)) ||
               launch_wino_s3_pack(ctx->stream, *upad, P, (int)Np, K, reinterpret_cast<unsigned short *>(*us)))
        return dt_fail(ctx, DT_ERR_DEVICE, "split-bf16 pack launch failed");
    g.a = reinterpret_cast<unsigned short *>(*vs); g.b = reinterpret_cast<unsigned short *>(*us); g.c = d_m;
    if (rows_form) { g.a = nullptr; g.a_f32 = d_v; g.a_ld = K; g.act = 1; g.slope = 1.0f; }      // (no bias, no activation)
    g.c_ps = (long long)Mt * N; g.P = P; g.Mt = Mt; g.Mp = (int)Mp; g.N = N; g.Np = (int)Np; g.K = K; g.ldc = N; g.half = half;
    ProfScope pscope(ctx, "conv_gemm_s3", wino_gemm_s3_flops(g), (double)P * (2.0 * nt * Mt * K + 2.0 * nt * (double)K * N + 4.0 * (double)Mt * N), "test_gemm");
    if (ctx->prof && !ctx->capturing) ctx->prof_tab[wino_gemm_s3_half_chosen(g, 0) ? "s3_tile:128x2" : "s3_tile:256"].launches += 1;
    const int rc = launch_wino_gemm_s3(ctx->stream, g, 0);
    if (rc) return dt_fail(ctx, rc == 2 ? DT_ERR_ARG : DT_ERR_DEVICE, "dt_gemm_split: launch failed (rc=%d)", rc);
    return DT_OK;
}
extern "C" int dt_gemm_split_bf16(dt_ctx *ctx, const float *d_v, const float *d_u, int P, int Mt, int K, int N, int half, float *d_m)
{
    return dt_gemm_split(ctx, d_v, d_u, P, Mt, K, N, half, 3, d_m);
}

// Re-reads the tuning / test knobs from the environment into the context (they are otherwise read once, in
// dt_create).  Knobs that shape the uploaded weights (DT_WINO, DT_WINO_TILE) take effect at the next weight load.
extern "C" int dt_policy_reload(dt_ctx *ctx)
{
    if (!ctx) return DT_ERR_ARG;
    graphs_clear(ctx);        // a captured graph holds the launches of the OLD kernel selection
    policy_from_env(ctx->pol);
    return DT_OK;
}

// One knob of ONE context, without going through the process environment (which other contexts and threads share):
//   "pin" 1 / 0 : kernel selection independent of the batch (DT_PIN; parallel.py: deterministic=True) -- the other knobs are re-read
//                 from the environment as dt_policy_reload does, then pinned or not.  Captured graphs are dropped only when the value changes.
extern "C" int dt_policy_set(dt_ctx *ctx, const char *name, int value)
{
    if (!ctx || !name) return DT_ERR_ARG;
    if (strcmp(name, "pin") != 0) return dt_fail(ctx, DT_ERR_ARG, "dt_policy_set: unknown knob '%s'", name);
    const int v = value ? 1 : 0;
    if (v != (ctx->pol.pin ? 1 : 0)) graphs_clear(ctx);
    policy_from_env(ctx->pol, v);
    return DT_OK;
}

// ---------------------------------------------------------------------------
// profiling
// ---------------------------------------------------------------------------
static void prof_drain(dt_ctx *ctx)
{
    (void)hipStreamSynchronize(ctx->stream);
    for (auto &e : ctx->pending) {
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, e.a, e.b) == hipSuccess) {
            ctx->prof_tab[e.name].ms += ms;
            if (!e.tag.empty()) ctx->prof_tab[e.tag].ms += ms;
        }
        (void)hipEventDestroy(e.a);
        (void)hipEventDestroy(e.b);
    }
    ctx->pending.clear();
}

extern "C" int dt_profile_enable(dt_ctx *ctx, int on)
{
    if (!ctx) return DT_ERR_ARG;
    prof_drain(ctx);
    ctx->prof = on != 0;
    return DT_OK;
}

extern "C" int dt_profile_reset(dt_ctx *ctx)
{
    if (!ctx) return DT_ERR_ARG;
    prof_drain(ctx);
    ctx->prof_tab.clear();
    return DT_OK;
}

extern "C" int dt_profile_names(dt_ctx *ctx, char *buf, size_t buflen)
{
    if (!ctx || !buf || !buflen) return DT_ERR_ARG;
    prof_drain(ctx);
    std::string all;
    for (auto &kv : ctx->prof_tab) { all += kv.first; all += "\n"; }
    if (all.size() + 1 > buflen) return dt_fail(ctx, DT_ERR_ARG, "dt_profile_names: buffer too small (%zu needed)", all.size() + 1);
    memcpy(buf, all.c_str(), all.size() + 1);
    return DT_OK;
}

extern "C" int dt_profile_read(dt_ctx *ctx, const char *name, int64_t *launches, double *total_ms, double *flops,
                               double *bytes)
{
    if (!ctx || !name) return DT_ERR_ARG;
    prof_drain(ctx);
    auto it = ctx->prof_tab.find(name);
    ProfEntry e;
    if (it != ctx->prof_tab.end()) e = it->second;
    if (!strcmp(name, "graph_replay")) e.launches = ctx->graph_replays;        // counters of the hipGraph path
    if (!strcmp(name, "graph_capture")) e.launches = ctx->graph_captures;
    if (launches) *launches = e.launches;
    if (total_ms) *total_ms = e.ms;
    if (flops) *flops = e.flops;
    if (bytes) *bytes = e.bytes;
    return DT_OK;
}

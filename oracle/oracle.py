"""ctypes binding + graph composition for the CPU oracle (oracle.c).

TEST INFRASTRUCTURE ONLY -- see the header of oracle.c.  Imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the product
package.

Composition follows the reference graph definitions:
  yolov2_forward       models_detection/KerasYOLO.py:277-405 (+ weight order :244-274)
  tracker_forward      models_tracking/MultiObjDetTracker.py:160-189
  tinytracker_forward  models_tracking/TinyTracker.py:25-41
  decode_netout        utility/utils.py:208-257
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

BN_EPS = 1e-3      # Keras BatchNormalization default epsilon (none passed, KerasYOLO.py:280)
LEAKY = 0.1        # LeakyReLU(alpha=0.1), KerasYOLO.py:281

# (name, kernel, cin, cout, maxpool_after) -- KerasYOLO.py:279-396, main trunk
TRUNK = [
    (1, 3, 3, 32, True), (2, 3, 32, 64, True), (3, 3, 64, 128, False), (4, 1, 128, 64, False),
    (5, 3, 64, 128, True), (6, 3, 128, 256, False), (7, 1, 256, 128, False), (8, 3, 128, 256, True),
    (9, 3, 256, 512, False), (10, 1, 512, 256, False), (11, 3, 256, 512, False),
    (12, 1, 512, 256, False), (13, 3, 256, 512, True),   # skip tapped BEFORE the pool (:347)
    (14, 3, 512, 1024, False), (15, 1, 1024, 512, False), (16, 3, 512, 1024, False),
    (17, 1, 1024, 512, False), (18, 3, 512, 1024, False), (19, 3, 1024, 1024, False),
    (20, 3, 1024, 1024, False),
]


def build():
    """Compile liboracle.so with the committed Makefile (gcc)."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        _LIB = ctypes.CDLL(path)
        _LIB.orc_bbox_iou.restype = ctypes.c_float
        _LIB.orc_decode_netout.restype = ctypes.c_int
        _LIB.orc_associate_clip.restype = ctypes.c_int
        _LIB.orc_max_threads.restype = ctypes.c_int
    return _LIB


def set_threads(n):
    """OpenMP thread count of the C loops (0 keeps the default); returns the count in effect."""
    lib().orc_set_threads(int(n))
    return int(lib().orc_max_threads())


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def normalize_u8(img):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    out = np.empty(img.shape, dtype=np.float32)
    lib().orc_normalize_u8(_p(img), ctypes.c_int64(img.size), _p(out))
    return out


def resize_bilinear_u8(frames, out_h, out_w):
    """frames uint8 [n,Hs,Ws,3] -> [n,out_h,out_w,3] (cv2.resize INTER_LINEAR restatement, see oracle.c)."""
    f = np.ascontiguousarray(frames, dtype=np.uint8)
    n, Hs, Ws, _ = f.shape
    out = np.empty((n, out_h, out_w, 3), dtype=np.uint8)
    lib().orc_resize_bilinear_u8(_p(f), n, Hs, Ws, _p(out), out_h, out_w)
    return out


def conv2d(x, w_hwio, bias=None):
    x = _f(x); w = _f(w_hwio)
    B, H, W, Cin = x.shape
    KS, _, ci, Cout = w.shape
    assert ci == Cin
    out = np.empty((B, H, W, Cout), dtype=np.float32)
    b = _f(bias) if bias is not None else None
    lib().orc_conv2d(_p(x), B, H, W, Cin, _p(w), KS, Cout, _p(b) if b is not None else None, _p(out))
    return out


def bn_leaky(x, gamma, beta, mean, var, eps=BN_EPS, alpha=LEAKY):
    x = _f(x).copy()
    C = x.shape[-1]
    lib().orc_bn_leaky(_p(x), ctypes.c_int64(x.size // C), C, _p(_f(gamma)), _p(_f(beta)),
                       _p(_f(mean)), _p(_f(var)), ctypes.c_float(eps), ctypes.c_float(alpha))
    return x


def maxpool2(x):
    x = _f(x)
    B, H, W, C = x.shape
    out = np.empty((B, H // 2, W // 2, C), dtype=np.float32)
    lib().orc_maxpool2(_p(x), B, H, W, C, _p(out))
    return out


def space_to_depth2(x):
    x = _f(x)
    B, H, W, C = x.shape
    out = np.empty((B, H // 2, W // 2, 4 * C), dtype=np.float32)
    lib().orc_space_to_depth2(_p(x), B, H, W, C, _p(out))
    return out


def concat_c(a, b):
    a = _f(a); b = _f(b)
    out = np.empty(a.shape[:-1] + (a.shape[-1] + b.shape[-1],), dtype=np.float32)
    lib().orc_concat_c(_p(a), a.shape[-1], _p(b), b.shape[-1], ctypes.c_int64(a.size // a.shape[-1]), _p(out))
    return out


def convlstm_step(x, h, c, Wk, Uk, bias):
    x = _f(x); h = _f(h); c = _f(c)
    B, H, W, Cx = x.shape
    U = h.shape[-1]
    ho = np.empty_like(h); co = np.empty_like(c)
    lib().orc_convlstm_step(_p(x), B, H, W, Cx, _p(h), _p(c), U, _p(_f(Wk)), _p(_f(Uk)), _p(_f(bias)), _p(ho), _p(co))
    return ho, co


def lstm_step(x, h, c, Wk, Ur, bias):
    x = _f(x); h = _f(h); c = _f(c)
    B, D = x.shape
    U = h.shape[-1]
    ho = np.empty_like(h); co = np.empty_like(c)
    lib().orc_lstm_step(_p(x), B, D, _p(h), _p(c), U, _p(_f(Wk)), _p(_f(Ur)), _p(_f(bias)), _p(ho), _p(co))
    return ho, co


def dense_sigmoid(x, Wd, bd):
    x = _f(x)
    B, U = x.shape
    O = Wd.shape[1]
    out = np.empty((B, O), dtype=np.float32)
    lib().orc_dense_sigmoid(_p(x), B, U, _p(_f(Wd)), _p(_f(bd)), O, _p(out))
    return out


def global_maxpool(x):
    x = _f(x)
    B, H, W, C = x.shape
    out = np.empty((B, C), dtype=np.float32)
    lib().orc_global_maxpool(_p(x), B, H * W, C, _p(out))
    return out


def maxpool4_flatten(x):
    x = _f(x)
    B, H, W, C = x.shape
    out = np.empty((B, (H // 4) * (W // 4) * C), dtype=np.float32)
    lib().orc_maxpool4_flatten(_p(x), B, H, W, C, _p(out))
    return out


def bbox_iou(b1, b2):
    return float(lib().orc_bbox_iou(_p(_f(b1)), _p(_f(b2))))


def decode_netout(netout, obj_threshold, nms_threshold, anchors, nb_class, cap=None):
    """Returns (rows[n,8] = x,y,w,h,conf,label,score,cell ; netout_post).  The
    caller's array is NOT modified (a copy is transformed)."""
    work = _f(netout).copy()
    GH, GW, NB, S = work.shape
    assert S == 5 + nb_class
    if cap is None:
        cap = GH * GW * NB
    out = np.zeros((cap, 8), dtype=np.float32)
    n = lib().orc_decode_netout(_p(work), GH, GW, NB, nb_class, ctypes.c_float(obj_threshold),
                                ctypes.c_float(nms_threshold), _p(_f(anchors)), _p(out), cap)
    return out[:min(n, cap)].copy(), work


def associate_clip(boxes, counts, assoc_thr):
    """boxes [T,cap,8], counts [T] -> ids [T,cap] int32, n_ids."""
    boxes = _f(boxes)
    counts = np.ascontiguousarray(counts, dtype=np.int32)
    T, cap, _ = boxes.shape
    ids = np.empty((T, cap), dtype=np.int32)
    n = lib().orc_associate_clip(_p(boxes), _p(counts), T, cap, ctypes.c_float(assoc_thr), _p(ids))
    return ids, n


# ----------------------------------------------------------------------------
# darknet .weights parsing -- utility/utils.py:138-148 + KerasYOLO.py:244-274
# ----------------------------------------------------------------------------
def parse_darknet_blob(blob, nb_class, nb_box=5):
    """`blob` = float32 contents of the .weights file INCLUDING the 4-float
    header (WeightReader.offset starts at 4).  Returns per-layer dicts with
    Keras-layout tensors (kernel HWIO)."""
    blob = np.asarray(blob, dtype=np.float32)
    off = 4
    specs = [(i, k, ci, co) for (i, k, ci, co, _) in TRUNK] + [(21, 1, 512, 64), (22, 3, 1280, 1024)]
    layers = {}

    def take(n):
        nonlocal off
        v = blob[off:off + n]
        assert v.size == n, "weights blob too short"
        off += n
        return v

    for (i, k, ci, co) in specs:   # file order conv_1..conv_22
        beta = take(co); gamma = take(co); mean = take(co); var = take(co)
        kern = take(co * ci * k * k).reshape(co, ci, k, k).transpose(2, 3, 1, 0)
        layers[i] = dict(k=k, cin=ci, cout=co, gamma=gamma, beta=beta, mean=mean, var=var, kernel=np.ascontiguousarray(kern))
    co = nb_box * (5 + nb_class)
    bias = take(co)
    kern = take(co * 1024).reshape(co, 1024, 1, 1).transpose(2, 3, 1, 0)
    layers[23] = dict(k=1, cin=1024, cout=co, bias=bias, kernel=np.ascontiguousarray(kern))
    return layers, off


def yolov2_forward(frames, layers, taps=()):
    """frames float32 [B,H,W,3] already normalised.  Returns (netout
    [B,G,G,5,5+C], conv_feat [B,G,G,1024], {tap_name: tensor})."""
    x = _f(frames)
    got = {}
    skip = None
    for (i, k, ci, co, pool) in TRUNK:
        L = layers[i]
        x = bn_leaky(conv2d(x, L["kernel"]), L["gamma"], L["beta"], L["mean"], L["var"])
        if ("act_%d" % i) in taps:
            got["act_%d" % i] = x
        if i == 13:
            skip = x
        if pool:
            x = maxpool2(x)
    L = layers[21]
    s = bn_leaky(conv2d(skip, L["kernel"]), L["gamma"], L["beta"], L["mean"], L["var"])
    s = space_to_depth2(s)
    x = concat_c(s, x)                              # skip first (KerasYOLO.py:391)
    L = layers[22]
    feat = bn_leaky(conv2d(x, L["kernel"]), L["gamma"], L["beta"], L["mean"], L["var"])
    L = layers[23]
    raw = conv2d(feat, L["kernel"], L["bias"])
    B, G1, G2, ch = raw.shape
    return raw.reshape(B, G1, G2, 5, ch // 5), feat, got


def tracker_forward(frames, layers, trk):
    """MultiObjDetTracker graph for ONE clip.  frames [T,H,W,3] normalised.
    trk: dict(kernel [3,3,Cb+1024,4U], recurrent [3,3,U,4U], bias [4U],
    out_kernel [1,1,U,Cb], out_bias [Cb]).  Returns (tracking [T,G,G,5,5+C],
    detection [T,G,G,5,5+C])."""
    det, feat, _ = yolov2_forward(frames, layers)
    T, G1, G2, NB, S = det.shape
    z = concat_c(det.reshape(T, G1, G2, NB * S), feat)   # x_bbox first (MultiObjDetTracker.py:175)
    U = trk["recurrent"].shape[2]
    h = np.zeros((1, G1, G2, U), dtype=np.float32)
    c = np.zeros_like(h)
    outs = []
    for t in range(T):
        h, c = convlstm_step(z[t:t + 1], h, c, trk["kernel"], trk["recurrent"], trk["bias"])
        outs.append(conv2d(h, trk["out_kernel"], trk["out_bias"]))
    trkout = np.concatenate(outs, 0).reshape(T, G1, G2, NB, S)
    return trkout, det


def heatmap_from_boxes(box4, hs):
    box4 = _f(box4)
    out = np.empty((box4.shape[0], hs * hs), dtype=np.float32)
    lib().orc_heatmap_from_boxes(_p(box4), box4.shape[0], hs, _p(out))
    return out


def rect_from_heatmap(heat, hs, thresh=0.75):
    heat = _f(heat)
    out = np.empty((heat.shape[0], 4), dtype=np.int32)
    lib().orc_rect_from_heatmap(_p(heat), heat.shape[0], hs, ctypes.c_float(thresh), _p(out))
    return out


def encode_targets(objs, counts, dims, aug, grid_h, grid_w, nb_box, nb_class, image_h, image_w, true_box_buffer,
                   anchors):
    """preprocessing.py:171-188 + :214-293 -> (y [n,GH,GW,NB,5+C], b [n,TBB,4]) float64."""
    objs = np.ascontiguousarray(objs, dtype=np.int32)
    counts = np.ascontiguousarray(counts, dtype=np.int32)
    dims = np.ascontiguousarray(dims, dtype=np.int32)
    n, cap, _ = objs.shape
    anchors = np.ascontiguousarray(anchors, dtype=np.float64)
    y = np.empty((n, grid_h, grid_w, nb_box, 5 + nb_class), dtype=np.float64)
    b = np.empty((n, true_box_buffer, 4), dtype=np.float64)
    a = None if aug is None else np.ascontiguousarray(aug, dtype=np.float64)
    lib().orc_encode_targets(_p(objs), _p(counts), _p(dims), _p(a) if a is not None else None, n, cap, grid_h,
                             grid_w, nb_box, nb_class, image_h, image_w, true_box_buffer, _p(anchors), _p(y), _p(b))
    return y, b


def sequence_windows(folders, seq_len):
    """create_sequences_from_parsed_annotations (preprocessing.py:79-89) on folder ids: returns the
    start index of every emitted window, in order -- including the reference's behaviour at folder
    boundaries (the skipped-forward start is re-emitted for each loop index that lands before it)
    and its IndexError when the forward skip runs off the end.  Pinned by tests/golden/windows.npz."""
    folders = list(folders)
    starts = []
    for i in range(len(folders) - seq_len + 1):
        while folders[i] != folders[i + seq_len - 1]:      # IndexError propagates, like the reference
            i += 1
        starts.append(i)
    return starts


def tinytracker_forward(feat, det, tt, pool="Global"):
    """TinyTracker / TinyHeatmapTracker graph (TinyTracker.py:25-41,
    TinyHeatmapTracker.py:26-48).  feat [B,T,w,h,c], det [B,T,4 | hs*hs]; tt:
    dict(kernel [D,4U], recurrent [U,4U], bias [4U], dense_kernel [U,O], dense_bias [O]).
    Returns [B,T,O]."""
    B, T = feat.shape[:2]
    U = tt["recurrent"].shape[0]
    h = np.zeros((B, U), dtype=np.float32)
    c = np.zeros_like(h)
    out = np.zeros((B, T, tt["dense_kernel"].shape[1]), dtype=np.float32)
    for t in range(T):
        f = feat[:, t]
        v = global_maxpool(f) if pool == "Global" else maxpool4_flatten(f)
        x = np.concatenate([v, _f(det[:, t])], axis=1)
        h, c = lstm_step(x, h, c, tt["kernel"], tt["recurrent"], tt["bias"])
        out[:, t] = dense_sigmoid(h, tt["dense_kernel"], tt["dense_bias"])
    return out

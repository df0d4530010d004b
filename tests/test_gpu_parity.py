"""GPU suite (-m gpu): the HIP path, called through the C ABI, against
 (a) the golden vectors generated from the reference's numpy code,
 (b) the CPU oracle on the same seeded inputs (sizes the oracle finishes in seconds),
 (c) size-independent properties at full size.
Bars: exact box set / order / labels / track ids; coordinates and scores within
2e-6 relative (float32 exp / summation-order ulps); conv stacks within 1e-3 of
the activation scale (north_star: box coords within 1e-3 fp32)."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import oracle as orc
from utility import synth
import mi355_dt

pytestmark = pytest.mark.gpu

ANCHORS = [0.57273, 0.677385, 1.87446, 2.06253, 3.33843, 5.47434, 7.88282, 3.52778, 9.77052, 9.16828]


def dev(a, ctx):
    return torch.from_numpy(np.ascontiguousarray(a)).to(ctx.device)


def relerr(a, b):
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-12))


def chan_err(got, ref):
    """whole-network bar: per channel (last axis), max|got-ref| / max(1, max|ref| of that channel) -- one bad channel
    cannot hide behind the largest value of the grid.  Bar 3e-4 (measured 1e-5..1e-4 through 23 float32 layers)."""
    g = got.reshape(-1, got.shape[-1]).astype(np.float64)
    r = ref.reshape(-1, ref.shape[-1]).astype(np.float64)
    return float((np.abs(g - r).max(0) / np.maximum(1.0, np.abs(r).max(0))).max())


NET_TOL = 3e-4


def flat_c(a):
    """[..., NB, 5+C] grids -> [..., NB*(5+C)] so that every (anchor, field) pair is its own channel"""
    return a.reshape(a.shape[:-2] + (-1,))


def gap_threshold(values, default, lo, hi):
    """midpoint of the widest gap between neighbouring decision values inside [lo, hi]: a threshold no value of the
    oracle's lies close to, so a float32 rounding difference cannot flip a decision (tests/test_gpu_configs.py)"""
    v = np.sort(np.asarray(values, dtype=np.float64))
    v = v[(v > lo) & (v < hi)]
    if v.size == 0:
        return float(default)
    edges = np.concatenate([[lo], v, [hi]])
    k = int(np.argmax(np.diff(edges)))
    return float(np.float32(0.5 * (edges[k] + edges[k + 1])))


def oracle_scores(grid, C, anchors=None):
    _, post = orc.decode_netout(grid, 0.0, 2.0, anchors if anchors is not None else ANCHORS, C)
    return post[..., 5:]


def iou_rows(a, b):
    a = a.astype(np.float64); b = b.astype(np.float64)
    ix = np.minimum(a[:, 0] + a[:, 2] / 2, b[:, 0] + b[:, 2] / 2) - np.maximum(a[:, 0] - a[:, 2] / 2, b[:, 0] - b[:, 2] / 2)
    iy = np.minimum(a[:, 1] + a[:, 3] / 2, b[:, 1] + b[:, 3] / 2) - np.maximum(a[:, 1] - a[:, 3] / 2, b[:, 1] - b[:, 3] / 2)
    inter = np.clip(ix, 0, None) * np.clip(iy, 0, None)
    return inter / (a[:, 2] * a[:, 3] + b[:, 2] * b[:, 3] - inter)


def box_err(got, ref):
    """north_star bar: box coords within 1e-3.  x,y live in [0,1] (absolute);
    w,h = anchor*exp(t)/G are unbounded, so they are compared relative to max(1,|ref|)."""
    if len(ref) == 0:
        return 0.0
    exy = np.abs(got[:, :2] - ref[:, :2]).max()
    ewh = (np.abs(got[:, 2:4] - ref[:, 2:4]) / np.maximum(1.0, np.abs(ref[:, 2:4]))).max()
    return float(max(exy, ewh))


# ------------------------------------------------------------------ decode / NMS
def _check_decode_rows(rows, g_rows):
    assert len(rows) == len(g_rows), "box count differs"
    if len(g_rows):
        assert np.array_equal(rows[:, 5], g_rows[:, 5]), "labels / order differ"
        np.testing.assert_allclose(rows[:, :5], g_rows[:, :5], rtol=2e-6, atol=1e-6)
        np.testing.assert_allclose(rows[:, 6], g_rows[:, 6], rtol=2e-6, atol=1e-6)


@pytest.mark.parametrize("name", [
    "g13_c80_n24", "g13_c12_n32", "g19_c12_n128", "g3_background", "g5_rescale", "g3_relabel",
    "g7_c5_lowthr", "g7_c12_nms09", "g7_c12_nms01", "g5_c1", "g4_dense"])
def test_decode_matches_reference_golden(ctx, golden_dir, name):
    d = np.load(os.path.join(golden_dir, "decode_%s.npz" % name))
    C = int(d["nb_class"])
    r = ctx.decode(dev(d["netout"][None], ctx), float(d["obj_threshold"]), float(d["nms_threshold"]), d["anchors"],
                   C, want_classes=True, want_post=True)
    n = int(r["counts"][0])
    rows = r["boxes"][0, :n].cpu().numpy()
    _check_decode_rows(rows, d["boxes"])
    if n:
        np.testing.assert_allclose(r["classes"][0, :n].cpu().numpy(), d["classes"], rtol=2e-6, atol=1e-7)
    if "netout_post" in d:
        np.testing.assert_allclose(r["post"][0].cpu().numpy(), d["netout_post"], rtol=2e-6, atol=1e-7)


def test_decode_drop_in_function_mutates_like_reference(golden_dir):
    from utility.utils import decode_netout
    d = np.load(os.path.join(golden_dir, "decode_g5_rescale.npz"))
    net = d["netout"].copy()
    boxes = decode_netout(net, float(d["obj_threshold"]), float(d["nms_threshold"]), list(d["anchors"]), int(d["nb_class"]))
    assert len(boxes) == len(d["boxes"])
    np.testing.assert_allclose(net, d["netout_post"], rtol=2e-6, atol=1e-7)      # in-place, like utils.py:214-216
    for b, g in zip(boxes, d["boxes"]):
        assert b.get_label() == int(g[5])
        assert abs(b.get_score() - g[6]) < 1e-6 and abs(b.x - g[0]) < 1e-6
        assert np.shares_memory(b.classes, net)                                   # views, like the reference


def _planted(seed, G, C, n_obj):
    rs = np.random.RandomState(seed)
    g = rs.randn(G, G, 5, 5 + C).astype(np.float32)
    g[..., 4] -= 4.0
    for k, cell in enumerate(rs.permutation(G * (G - 1))[:n_obj]):
        row, col = divmod(int(cell), G - 1)
        b = int(rs.randint(0, 5)); cls = int(rs.randint(0, C))
        g[row, col, b, 4] = 4.0 + rs.rand()
        g[row, col, b, 5 + cls] += 12.0 + rs.rand()
        if k % 3 == 0:
            g[row, col + 1, b, :] = g[row, col, b, :]
            g[row, col + 1, b, 0] -= 3.0
            g[row, col + 1, b, 4] -= 0.25 + 0.5 * rs.rand()
    return g


@pytest.mark.parametrize("G,C,n_obj,B", [(13, 80, 24, 6), (13, 12, 32, 16), (19, 12, 128, 5), (7, 3, 40, 9)])
def test_decode_batch_vs_oracle(ctx, G, C, n_obj, B):
    grids = np.stack([_planted(1000 + 17 * i + G, G, C, n_obj) for i in range(B)])
    r = ctx.decode(dev(grids, ctx), 0.5, 0.45, ANCHORS, C)
    counts = r["counts"].cpu().numpy()
    boxes = r["boxes"].cpu().numpy()
    for i in range(B):
        rows, _ = orc.decode_netout(grids[i], 0.5, 0.45, ANCHORS, C)
        assert counts[i] == len(rows)
        got = boxes[i, :counts[i]]
        assert np.array_equal(got[:, 5], rows[:, 5]) and np.array_equal(got[:, 7], rows[:, 7])
        np.testing.assert_allclose(got[:, :7], rows[:, :7], rtol=2e-6, atol=1e-6)
        assert np.all(np.diff(got[:, 7]) > 0), "creation (row,col,b) order"


@pytest.mark.parametrize("C,thr", [(3, 1e-4), (20, 1e-4), (80, 2e-3)])
def test_decode_dense_low_threshold(ctx, C, thr):
    """Thousands of kept (cell, class) scores per frame: C=3 fills the kernel's LDS score list almost to its capacity
    (845 x 3 of 4096: per-class lists of 845 boxes), C=20 overflows it (16.9 k: the NMS gathers from global memory
    instead), C=80 at 2e-3 is a mix across frames.  Same boxes, order and surviving scores as the oracle."""
    B, G = 3, 13
    rs = np.random.RandomState(77 + C)
    grids = rs.randn(B, G, G, 5, 5 + C).astype(np.float32)
    grids[..., 4] -= 2.0
    grids[..., 2:4] *= 0.5
    r = ctx.decode(dev(grids, ctx), thr, 0.45, ANCHORS, C, want_post=True)
    counts = r["counts"].cpu().numpy()
    boxes = r["boxes"].cpu().numpy()
    post = r["post"].cpu().numpy()
    kept = 0
    for i in range(B):
        rows, opost = orc.decode_netout(grids[i], thr, 0.45, ANCHORS, C)
        kept += int((opost[..., 5:] > 0).sum())
        assert counts[i] == len(rows) and len(rows) > 100
        got = boxes[i, :counts[i]]
        assert np.array_equal(got[:, 5], rows[:, 5]) and np.array_equal(got[:, 7], rows[:, 7])
        np.testing.assert_allclose(got[:, :7], rows[:, :7], rtol=2e-6, atol=1e-6)
        assert np.array_equal(post[i][..., 5:] > 0, opost[..., 5:] > 0), "suppressed set differs"
        np.testing.assert_allclose(post[i], opost, rtol=2e-6, atol=1e-7)
    print("decode dense: C=%d, %d scores survive NMS in %d frames" % (C, kept, B))


@pytest.mark.parametrize("G,C,thr,dense", [(26, 12, 0.5, False), (26, 80, 0.3, False), (32, 3, 1e-3, True), (40, 12, 0.02, True),
                                           (20, 12, 0.5, False)])
def test_decode_grids_above_1920_cells(ctx, G, C, thr, dense):
    """utils.py:208-257 has no size limit.  Grids above 19x19x5 cells (832x832 -> 26x26, 1024 -> 32x32, 1280 -> 40x40:
    3380 / 5120 / 8000 cells) run the kernel instance whose per-candidate arrays live in a global scratch; 20x20
    (2000 cells) is the first size past the LDS-resident limit.  Planted objects over a quiet background (a few boxes per
    frame, overlapping pairs for the NMS) and dense low-threshold frames (thousands of kept scores: the overflow path
    of the NMS): same boxes, order, labels, scores and post-NMS grid as the oracle."""
    rs = np.random.RandomState(G * 7 + C)
    B = 3
    if dense:
        grids = rs.randn(B, G, G, 5, 5 + C).astype(np.float32)
        grids[..., 4] -= 2.0
        grids[..., 2:4] *= 0.5
    else:
        grids = (rs.randn(B, G, G, 5, 5 + C) * 0.3).astype(np.float32)
        grids[..., 4] -= 6.0                                       # quiet background
        for b in range(B):
            for k in range(40):                                    # planted objects, in overlapping pairs
                r, c, a, cl = rs.randint(G), rs.randint(G - 1), rs.randint(5), rs.randint(C)
                for dc in (0, 1):
                    grids[b, r, c + dc, a, 4] = 4.0 + rs.rand()
                    grids[b, r, c + dc, a, 5 + cl] = 9.0 + rs.rand()
                    grids[b, r, c + dc, a, 2:4] = 1.2
    r = ctx.decode(dev(grids, ctx), thr, 0.45, ANCHORS, C, want_post=True, want_classes=True)
    counts = r["counts"].cpu().numpy()
    boxes = r["boxes"].cpu().numpy()
    post = r["post"].cpu().numpy()
    nbox = 0
    for i in range(B):
        rows, opost = orc.decode_netout(grids[i], thr, 0.45, ANCHORS, C)
        assert counts[i] == len(rows), (counts[i], len(rows))
        got = boxes[i, :counts[i]]
        assert np.array_equal(got[:, 5], rows[:, 5]) and np.array_equal(got[:, 7], rows[:, 7])
        np.testing.assert_allclose(got[:, :7], rows[:, :7], rtol=2e-6, atol=1e-6)
        assert np.array_equal(post[i][..., 5:] > 0, opost[..., 5:] > 0), "suppressed set differs"
        np.testing.assert_allclose(post[i], opost, rtol=2e-6, atol=1e-7)
        nbox += len(rows)
    assert nbox >= (300 if dense else 60), nbox
    # the same call twice: bitwise identical (the scratch is reused)
    r2 = ctx.decode(dev(grids, ctx), thr, 0.45, ANCHORS, C, want_post=True)
    assert torch.equal(r2["boxes"], r["boxes"]) and torch.equal(r2["counts"], r["counts"]) and torch.equal(r2["post"], r["post"])


def test_detector_832_decodes_end_to_end(ctx):
    """an 832x832 frame (26x26 grid, 3380 cells: above the LDS-resident decode limit) through KerasYOLO.detect"""
    det, layers, _ = _detector(ctx, 832, 832, 12)
    frames = synth.synth_clip(1, 832, 832, 3, seed=33)
    c = det.model.ctx
    net = c.detect_forward(dev(frames, c))
    assert net.shape == (1, 26, 26, 5, 17)
    netc = net.cpu().numpy()
    thr = float(np.sort((1 / (1 + np.exp(-netc[0, ..., 4]))).ravel())[-40])       # ~40 cells above the objectness threshold
    r = c.decode(net, thr * 0.5, 0.45, ANCHORS, 12)
    rows, _ = orc.decode_netout(netc[0], thr * 0.5, 0.45, ANCHORS, 12)
    assert int(r["counts"][0]) == len(rows)
    got = r["boxes"][0, :len(rows)].cpu().numpy()
    assert np.array_equal(got[:, 7], rows[:, 7]) and np.array_equal(got[:, 5], rows[:, 5])
    np.testing.assert_allclose(got[:, :7], rows[:, :7], rtol=2e-6, atol=1e-6)


def test_decode_is_deterministic_across_runs(ctx):
    """The kernel's LDS lists are filled through atomics (order differs from run to run); boxes, counts, order and the
    post-NMS grid must not: 64 dense frames, three thresholds, decoded five times each -- bitwise equal."""
    rs = np.random.RandomState(5)
    for C, thr in ((80, 0.02), (12, 0.05), (3, 1e-4)):
        grids = rs.randn(64, 13, 13, 5, 5 + C).astype(np.float32)
        grids[..., 4] -= 1.0
        grids[..., 2:4] *= 0.5
        d = dev(grids, ctx)
        first = None
        for _ in range(5):
            r = ctx.decode(d, thr, 0.45, ANCHORS, C, want_post=True)
            cur = (r["counts"].clone(), r["boxes"].clone(), r["post"].clone())
            if first is None:
                first = cur
                assert int(cur[0].min()) > 20
            else:
                assert all(torch.equal(a, b) for a, b in zip(first, cur)), "decode differs between identical launches"


def test_decode_properties_full_size(ctx):
    """Size-independent properties on 64 frames of the 19x19 / 128-object case:
    batch invariance, cap truncation keeps a prefix, survivors are above threshold,
    no two same-label survivors overlap at IoU >= nms threshold."""
    B, G, C = 64, 19, 12
    grids = np.stack([_planted(5000 + i, G, C, 128) for i in range(B)])
    d = dev(grids, ctx)
    full = ctx.decode(d, 0.5, 0.45, ANCHORS, C)
    one = ctx.decode(d[7:8].contiguous(), 0.5, 0.45, ANCHORS, C)
    n7 = int(full["counts"][7])
    assert int(one["counts"][0]) == n7
    assert torch.equal(one["boxes"][0, :n7], full["boxes"][7, :n7])
    capped = ctx.decode(d, 0.5, 0.45, ANCHORS, C, cap=50)
    assert torch.equal(capped["counts"], full["counts"])
    assert torch.equal(capped["boxes"][:, :50], full["boxes"][:, :50])
    bx = full["boxes"].cpu().numpy(); cn = full["counts"].cpu().numpy()
    for i in range(0, B, 9):
        r = bx[i, :cn[i]]
        assert np.all(r[:, 6] > 0.5)
        pairs = [(a, b) for a in range(len(r)) for b in range(a + 1, len(r)) if r[a, 5] == r[b, 5]]
        if pairs:
            pr = np.array([np.concatenate([r[a, :4], r[b, :4]]) for a, b in pairs], dtype=np.float32)
            iou = ctx.bbox_iou(dev(pr, ctx)).cpu().numpy()
            assert np.all(iou < 0.45)


def test_bbox_iou_bit_exact_vs_reference(ctx, golden_dir):
    d = np.load(os.path.join(golden_dir, "bbox_iou.npz"))
    got = ctx.bbox_iou(dev(d["pairs"], ctx)).cpu().numpy()
    assert np.array_equal(got, d["iou"].astype(np.float32))


# ------------------------------------------------------------------ conv kernel
@pytest.mark.parametrize("B,H,W,Cin,k,Cout,pool", [
    (2, 10, 12, 32, 3, 64, 0),     # N<=64 tile config, M edge
    (1, 13, 13, 64, 3, 128, 0),    # 128x128 config, M=169 (one partial tile)
    (3, 13, 13, 96, 1, 85, 0),     # 1x1, N edge (85), Cin = 3 chunks
    (2, 8, 8, 32, 3, 32, 1),       # fused 2x2 max-pool
    (1, 26, 26, 64, 3, 160, 1),    # pool, two N tiles with edge
    (2, 12, 8, 64, 3, 128, 2),     # pool + unpooled (skip tap of conv_13)
    (2, 6, 10, 64, 1, 64, 3),      # 1x1 + tf.space_to_depth(2) (conv_21)
    (1, 13, 13, 1280, 3, 256, 0),  # long K (conv_22's Cin)
])
def test_conv2d_vs_oracle(ctx, B, H, W, Cin, k, Cout, pool):
    rs = np.random.RandomState(B * 1000 + H + Cin + Cout)
    x = rs.randn(B, H, W, Cin).astype(np.float32)
    w = (rs.randn(k, k, Cin, Cout) * np.sqrt(2.0 / (k * k * Cin))).astype(np.float32)
    b = rs.randn(Cout).astype(np.float32)
    ref = orc.conv2d(x, w, b)
    orc.lib().orc_leaky(ref.ctypes.data_as(__import__("ctypes").c_void_p), __import__("ctypes").c_int64(ref.size),
                        __import__("ctypes").c_float(0.1))
    got = ctx.conv2d(dev(x, ctx), w, b, leaky_slope=0.1, pool=pool)
    tol = 2e-5
    if pool == 0:
        assert relerr(got.cpu().numpy(), ref) < tol
    elif pool == 1:
        assert relerr(got.cpu().numpy(), orc.maxpool2(ref)) < tol
    elif pool == 2:
        assert relerr(got[0].cpu().numpy(), ref) < tol
        assert relerr(got[1].cpu().numpy(), orc.maxpool2(ref)) < tol
    else:
        assert relerr(got.cpu().numpy(), orc.space_to_depth2(ref)) < tol


def test_conv2d_detects_transpose(ctx):
    """asymmetric one-hot kernel: output channel n copies input channel (n*7)%Cin
    shifted by the tap -- a swapped A/B fragment or C/D row/col map fails this."""
    B, H, W, Cin, Cout = 1, 9, 11, 32, 96
    x = np.arange(B * H * W * Cin, dtype=np.float32).reshape(B, H, W, Cin) % 251
    w = np.zeros((3, 3, Cin, Cout), dtype=np.float32)
    for n in range(Cout):
        w[n % 3, (n // 3) % 3, (n * 7) % Cin, n] = 1.0
    got = ctx.conv2d(dev(x, ctx), w, None, leaky_slope=1.0, pool=0).cpu().numpy()
    assert np.array_equal(got, orc.conv2d(x, w))


# ------------------------------------------------------------------ detector
def _detector(ctx_unused, H, W, C, seed=1234):
    from models_detection.KerasYOLO import KerasYOLO
    labels = [str(i) for i in range(C)]
    blob = synth.synth_darknet_blob(C, seed=seed)
    det = KerasYOLO({'LABELS': labels, 'BATCH_SIZE': 4, 'IMAGE_H': H, 'IMAGE_W': W, 'GRID_H': H // 32,
                     'GRID_W': W // 32}, weights=blob)
    layers, used = orc.parse_darknet_blob(blob, C)
    assert used == blob.size
    return det, layers, blob


@pytest.mark.parametrize("H,W,B", [(64, 96, 3), (416, 416, 2), (32, 32, 1)])
def test_conv1_split_bf16_vs_oracle_and_fp32_mfma(ctx, monkeypatch, H, W, B):
    """conv_1 + x/255 + BN + LeakyReLU + 2x2 max (KerasYOLO.py:278-282) as conv1_s3_kernel (bf16 MFMA on split operands; default)
    against the oracle and against the fp32 MFMA kernel (DT_S3_CONV1=0), for uint8 frames (bytes are exact bf16 numbers; 1/255
    folded into the three-term weights) and for the same frames handed over as float32 x/255 (general three-term form).  The fp32
    MFMA kernel gives the same bits for both inputs (x/255 table); the split kernel's two forms place one float32 rounding
    differently (activation vs weight) and agree to 2e-6."""
    det, layers, _ = _detector(ctx, H, W, 12)
    c = det.model.ctx
    frames = np.random.RandomState(H + W).randint(0, 256, size=(B, H, W, 3)).astype(np.uint8)
    frames[0, :2, :5] = 255; frames[0, -1, -3:] = 0                # extremes of the table on the image border
    x = orc.normalize_u8(frames)
    L = layers[1]
    want = orc.maxpool2(orc.bn_leaky(orc.conv2d(x, L["kernel"]), L["gamma"], L["beta"], L["mean"], L["var"]))
    got = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("DT_S3_CONV1", mode)
        c.reload_policy()
        got[mode] = c.detector_extract(dev(frames, c), "max_pooling2d_1").cpu().numpy()
        got[mode + "f"] = c.detector_extract(dev(x, c), "max_pooling2d_1").cpu().numpy()
    monkeypatch.delenv("DT_S3_CONV1")
    c.reload_policy()
    assert np.array_equal(got["0"], got["0f"])
    for k in ("1", "1f", "0"):
        assert got[k].shape == (B, H // 2, W // 2, 32)
        assert chan_err(got[k], want) < 5e-6, (k, chan_err(got[k], want))
    assert chan_err(got["1"], got["0"]) < 2e-6 and chan_err(got["1f"], got["0"]) < 2e-6 and chan_err(got["1"], got["1f"]) < 2e-6


@pytest.mark.parametrize("H,W,C,B", [(64, 64, 12, 3), (96, 64, 80, 2)])
def test_detector_forward_vs_oracle_small(ctx, H, W, C, B):
    det, layers, _ = _detector(ctx, H, W, C)
    rs = np.random.RandomState(42)
    frames = rs.randint(0, 256, size=(B, H, W, 3)).astype(np.uint8)
    ref_net, ref_feat, taps = orc.yolov2_forward(orc.normalize_u8(frames), layers, taps=("act_13",))
    net, feat = det.model.ctx.detect_forward(dev(frames, det.model.ctx), want_feat=True)
    assert chan_err(flat_c(net.cpu().numpy()), flat_c(ref_net)) < NET_TOL
    assert chan_err(feat.cpu().numpy(), ref_feat) < NET_TOL
    # float32 frames (already normalised): conv_1's general split form instead of the uint8 one (one float32 rounding sits on
    # the activation instead of on the weight): the same network to rounding
    net32 = det.model.ctx.detect_forward(dev(orc.normalize_u8(frames), det.model.ctx))
    assert chan_err(flat_c(net32.cpu().numpy()), flat_c(net.cpu().numpy())) < 1e-4      # (F(6x6) on every layer, DT_WINO=2: 4e-5)
    assert chan_err(flat_c(net32.cpu().numpy()), flat_c(ref_net)) < NET_TOL
    # named taps (KerasYOLO.extract)
    det.model.ctx.detect_forward_internal(dev(frames, det.model.ctx))
    a13 = det.model.ctx.detector_tap("act_13", B).cpu().numpy()
    assert chan_err(a13, taps["act_13"]) < NET_TOL
    assert torch.equal(det.model.ctx.detector_tap("conv_23", B).reshape(net.shape), net)


def test_detector_full_size_one_frame_vs_oracle(ctx):
    """configs[0]-shaped case: one 416x416 frame through the full YOLOv2 (C=80)."""
    det, layers, _ = _detector(ctx, 416, 416, 80)
    frame = synth.synth_clip(1, 416, 416, 3, seed=7)
    ref_net, _, _ = orc.yolov2_forward(orc.normalize_u8(frame), layers)
    net = det.model.ctx.detect_forward(dev(frame, det.model.ctx)).cpu().numpy()
    assert net.shape == (1, 13, 13, 5, 85)
    assert chan_err(flat_c(net), flat_c(ref_net)) < NET_TOL
    # box parity on the decoded output (boost objectness so boxes exist); the threshold sits in the widest gap of the
    # oracle's scores near 0.3, so the comparison is unconditional
    boost = net.copy(); boost[..., 4] += 2.0; boost[..., 5:] *= 4.0
    rboost = ref_net.copy(); rboost[..., 4] += 2.0; rboost[..., 5:] *= 4.0
    thr = gap_threshold(oracle_scores(rboost[0], 80).ravel(), 0.3, 0.25, 0.35)
    rows, _ = orc.decode_netout(rboost[0], thr, 0.45, ANCHORS, 80)
    r = det.model.ctx.decode(dev(boost, det.model.ctx), thr, 0.45, ANCHORS, 80)
    n = int(r["counts"][0])
    got = r["boxes"][0, :n].cpu().numpy()
    assert n > 0 and n == len(rows)
    assert np.array_equal(got[:, 7], rows[:, 7]) and np.array_equal(got[:, 5], rows[:, 5])
    assert box_err(got, rows) < 1e-3
    # IoU >= 0.999 for every box at least one pixel high and wide.  The random head also emits degenerate boxes (w = 1.6 image
    # widths x h = 0.0009 = 0.4 pixels): there a centre error of 7e-7 -- three orders inside the coordinate bar above -- is
    # 7.5e-4 of the box height and the IoU reads 0.9985 (tools/diag_iou.py); for those the coordinate bar is the statement
    real = np.minimum(rows[:, 2], rows[:, 3]) >= 1.0 / 416.0
    assert real.sum() > 100
    assert iou_rows(got[real, :4], rows[real, :4]).min() >= 0.999
    assert iou_rows(got[:, :4], rows[:, :4]).min() >= 0.995


def test_detector_full_size_default_policy_vs_oracle(ctx):
    """Four 416x416 frames with the DEFAULT policy: conv_3..8 in Winograd form, the 13x13 layers in Winograd form
    on a 2x2 frame mosaic (64 tiles) writing into the strided concat / feature buffers -- against the oracle."""
    det, layers, _ = _detector(ctx, 416, 416, 12)
    frames = np.concatenate([synth.synth_clip(2, 416, 416, 3, seed=s) for s in (11, 12)])
    ref_net, ref_feat, _ = orc.yolov2_forward(orc.normalize_u8(frames), layers)
    c = det.model.ctx
    c.profile_reset(); c.profile_enable(True)
    net, feat = c.detect_forward(dev(frames, c), want_feat=True)
    c.profile_enable(False)
    assert c.profile_read("wino_input")["launches"] >= 12       # conv_3,5,6,8,9,11,13,14,16,18,19,20,22
    assert chan_err(flat_c(net.cpu().numpy()), flat_c(ref_net)) < NET_TOL
    assert chan_err(feat.cpu().numpy(), ref_feat) < NET_TOL


def test_detector_batch_invariance_full_size(ctx, monkeypatch):
    """A frame's output must not depend on WHERE it sits in the batch or on its neighbours, and must be
    reproducible run to run (bit-exact).  Position independence is bit-exact with one frame per Winograd
    tile grid (DT_WINO_MOSAIC=1); with the default 2x2 frame mosaic of the 13x13 layers a frame's tile
    offsets depend on its slot in the group of four, so its values move at rounding level (<= 2e-5
    relative; north_star's bar is 1e-3).  Across different batch sizes the split-K factor / the
    Winograd-vs-direct choice of the small-M layers changes the fp32 summation order: closeness only."""
    det, _, _ = _detector(ctx, 416, 416, 12)
    frames = np.concatenate([synth.synth_clip(3, 416, 416, 2, seed=s) for s in (1, 2)])
    d = dev(frames, det.model.ctx)
    perm = [4, 0, 5, 2, 1, 3]
    full = det.model.ctx.detect_forward(d)
    shuffled = det.model.ctx.detect_forward(d[perm].contiguous())
    assert relerr(shuffled.cpu().numpy(), full[perm].cpu().numpy()) < 2e-5, "position / neighbour independence"
    again = det.model.ctx.detect_forward(d)
    assert torch.equal(again, full), "run-to-run determinism"
    single = det.model.ctx.detect_forward(d[4:5].contiguous())
    assert relerr(single[0].cpu().numpy(), full[4].cpu().numpy()) < 1e-4
    monkeypatch.setenv("DT_WINO_MOSAIC", "1")
    det.model.ctx.reload_policy()
    full1 = det.model.ctx.detect_forward(d)
    shuffled1 = det.model.ctx.detect_forward(d[perm].contiguous())
    assert torch.equal(shuffled1, full1[perm]), "position / neighbour independence (bit-exact without the mosaic)"
    assert relerr(full1.cpu().numpy(), full.cpu().numpy()) < 2e-5


@pytest.mark.parametrize("ks,Cin,Cout,M_hw,B", [(3, 512, 1024, 13, 1), (3, 1024, 1024, 13, 2), (1, 1024, 512, 13, 1),
                                                (1, 1024, 85, 13, 3)])
def test_conv2d_split_k_path(ctx, ks, Cin, Cout, M_hw, B):
    """deep 13x13 layers at tiny batch take the split-K + deterministic combine path"""
    rs = np.random.RandomState(ks * 100 + B)
    x = rs.randn(B, M_hw, M_hw, Cin).astype(np.float32)
    w = (rs.randn(ks, ks, Cin, Cout) * np.sqrt(2.0 / (ks * ks * Cin))).astype(np.float32)
    b = rs.randn(Cout).astype(np.float32)
    ref = orc.conv2d(x, w, b)
    ref = np.where(ref > 0, ref, 0.1 * ref).astype(np.float32)
    got = ctx.conv2d(dev(x, ctx), w, b, leaky_slope=0.1, pool=0)
    assert relerr(got.cpu().numpy(), ref) < 2e-5
    assert torch.equal(ctx.conv2d(dev(x, ctx), w, b, leaky_slope=0.1, pool=0), got)


# ------------------------------------------------------------------ ConvLSTM / tracker
def test_convlstm_step_vs_oracle(ctx, atol=2e-5):
    rs = np.random.RandomState(9)
    B, H, W, Cx, U = 3, 5, 7, 96, 64
    x = rs.randn(B, H, W, Cx).astype(np.float32)
    h = (rs.randn(B, H, W, U) * .5).astype(np.float32); c = rs.randn(B, H, W, U).astype(np.float32)
    Wk = (rs.randn(3, 3, Cx, 4 * U) * .05).astype(np.float32); Uk = (rs.randn(3, 3, U, 4 * U) * .05).astype(np.float32)
    b = rs.randn(4 * U).astype(np.float32) * .1
    rh, rc = orc.convlstm_step(x, h, c, Wk, Uk, b)
    gh, gc = ctx.convlstm_step(dev(x, ctx), dev(h, ctx), dev(c, ctx), Wk, Uk, b)
    np.testing.assert_allclose(gh.cpu().numpy(), rh, rtol=1e-4, atol=atol)
    np.testing.assert_allclose(gc.cpu().numpy(), rc, rtol=1e-4, atol=atol)


def _tracker(H, W, T, C=12, seed=1235):
    from models_tracking.MultiObjDetTracker import MultiObjDetTracker

    class Trk(MultiObjDetTracker):
        IMAGE_H, IMAGE_W = H, W
        GRID_H, GRID_W = H // 32, W // 32
        SEQUENCE_LENGTH = T
        LABELS = [str(i) for i in range(C)]
        LOAD_MODEL = False
    blob = synth.synth_darknet_blob(C)
    tw = synth.synth_tracker_weights(C, seed=seed)
    return Trk(detector_weights=blob, tracker_weights=tw), blob, tw


def test_track_forward_vs_oracle_small(ctx):
    H, W, T, n_clips, C = 64, 96, 4, 3, 12
    trk, blob, tw = _tracker(H, W, T, C)
    layers, _ = orc.parse_darknet_blob(blob, C)
    frames = np.stack([synth.synth_clip(T, H, W, 2, seed=20 + i) for i in range(n_clips)])
    got_trk, got_det = trk.model.predict([frames, None])
    for i in range(n_clips):
        ref_trk, ref_det = orc.tracker_forward(orc.normalize_u8(frames[i]), layers, tw)
        assert chan_err(flat_c(got_det[i]), flat_c(ref_det)) < NET_TOL
        assert chan_err(flat_c(got_trk[i]), flat_c(ref_trk)) < NET_TOL


def test_track_clips_boxes_and_ids_vs_oracle(ctx):
    """End to end on device (forward -> decode -> associate) vs the oracle chain;
    the tracker head is scaled so that boxes exist.  Track ids bit-exact."""
    H, W, T, n_clips, C = 64, 64, 6, 4, 12
    trk, blob, tw = _tracker(H, W, T, C)
    tw = dict(tw)
    tw["out_kernel"] = tw["out_kernel"] * 40.0
    ob = tw["out_bias"].copy(); ob[4::17] = 1.5; tw["out_bias"] = ob
    trk.model.set_weights(tw)
    trk.OBJ_THRESHOLD, trk.NMS_THRESHOLD, trk.ASSOC_THRESHOLD = 0.3, 0.45, 0.3
    layers, _ = orc.parse_darknet_blob(blob, C)
    frames = np.stack([synth.synth_clip(T, H, W, 2, seed=40 + i) for i in range(n_clips)])
    res = trk.track_clips(frames)
    cap = res["boxes"].shape[2]
    total = 0
    for i in range(n_clips):
        ref_trk, _ = orc.tracker_forward(orc.normalize_u8(frames[i]), layers, tw)
        rb = np.zeros((T, cap, 8), dtype=np.float32); rc = np.zeros(T, dtype=np.int32)
        for t in range(T):
            rows, _ = orc.decode_netout(ref_trk[t], 0.3, 0.45, ANCHORS, C)
            rb[t, :len(rows)] = rows; rc[t] = len(rows)
        assert np.array_equal(res["counts"][i].cpu().numpy(), rc)
        gb = res["boxes"][i].cpu().numpy()
        for t in range(T):
            assert np.array_equal(gb[t, :rc[t], 5], rb[t, :rc[t], 5])
            assert np.array_equal(gb[t, :rc[t], 7], rb[t, :rc[t], 7])
            assert box_err(gb[t, :rc[t]], rb[t, :rc[t]]) < 1e-3
        rid, rn = orc.associate_clip(rb, rc, 0.3)
        assert np.array_equal(res["ids"][i].cpu().numpy(), rid), "track ids must be bit-exact"
        assert int(res["nids"][i]) == rn
        total += int(rc.sum())
    assert total > 0, "test is vacuous without boxes"


@pytest.mark.parametrize("n_clips,T,cap,obj0,dobj", [(7, 12, 40, 5, 4), (3, 8, 120, 70, 20), (2, 70, 16, 5, 3)],
                         ids=["register_form", "general_form_over_64_boxes", "general_form_T_over_64"])
def test_associate_vs_oracle_synthetic(ctx, n_clips, T, cap, obj0, dobj):
    """Moving boxes with births, deaths, label changes and ties.  associate_kernel has two forms: boxes and ids in registers
    (every frame of the clip <= 64 boxes, T <= 64) and in LDS (anything else)."""
    rs = np.random.RandomState(3)
    boxes = np.zeros((n_clips, T, cap, 8), dtype=np.float32)
    counts = np.zeros((n_clips, T), dtype=np.int32)
    for c in range(n_clips):
        n_obj = obj0 + dobj * c
        pos = rs.rand(n_obj, 2); vel = (rs.rand(n_obj, 2) - .5) * .06; wh = rs.rand(n_obj, 2) * .2 + .05
        lab = rs.randint(0, 3, n_obj)
        for t in range(T):
            alive = [k for k in range(n_obj) if rs.rand() > 0.15]
            rs.shuffle(alive)
            for i, k in enumerate(alive[:cap]):
                p = pos[k] + vel[k] * t
                boxes[c, t, i] = [p[0], p[1], wh[k, 0], wh[k, 1], .9, lab[k] if rs.rand() > .05 else (lab[k] + 1) % 3, .8, i]
            counts[c, t] = min(len(alive), cap)
        boxes[c, 3, 1, :4] = boxes[c, 3, 0, :4]      # exact duplicate -> tie on IoU
    ids, nids = ctx.associate(dev(boxes, ctx), dev(counts, ctx), 0.3)
    for c in range(n_clips):
        rid, rn = orc.associate_clip(boxes[c], counts[c], 0.3)
        assert np.array_equal(ids[c].cpu().numpy(), rid)
        assert int(nids[c]) == rn


# ------------------------------------------------------------------ TinyTracker
@pytest.mark.parametrize("pool,fh,fw,fc,n_seq,T", [("Global", 26, 26, 512, 5, 6), ("Global", 13, 13, 512, 70, 3),
                                                    ("Max", 8, 8, 32, 3, 4)])
def test_tiny_forward_vs_oracle(ctx, pool, fh, fw, fc, n_seq, T):
    from models_tracking.TinyTracker import TinyTracker
    feat_dim = fc if pool == "Global" else (fh // 4) * (fw // 4) * fc
    tw = synth.synth_tiny_weights(feat_dim)
    cfg = {"model_tracker": {"name": "TinyTracker", "lstm_units": 512, "sequence_length": T},
           "train": {"pool": pool, "batch_size": 4}}
    tt = TinyTracker(cfg, feature_dims=(fh, fw, fc), weights=tw, ctx=ctx)
    rs = np.random.RandomState(11)
    feat = rs.randn(n_seq, T, fh, fw, fc).astype(np.float32)
    det = rs.rand(n_seq, T, 4).astype(np.float32)
    got = tt.model_tracker.predict([feat, det])
    ref = orc.tinytracker_forward(feat, det, tw, pool=pool)
    np.testing.assert_allclose(got, ref, rtol=1e-4, atol=2e-5)


def test_tinytracker_pipeline_vs_oracle(ctx):
    """detector -> act_13 tap + decode -> top box -> pool (+) det -> LSTM -> Dense,
    the whole single-object chain against the oracle."""
    from models_detection.KerasYOLO import KerasYOLO
    from models_tracking.TinyTracker import TinyTracker
    H = W = 64
    C, n_seq, T = 12, 3, 4
    blob = synth.synth_darknet_blob(C, head_std=0.3)
    det = KerasYOLO({'LABELS': [str(i) for i in range(C)], 'BATCH_SIZE': 4, 'IMAGE_H': H, 'IMAGE_W': W,
                     'GRID_H': 2, 'GRID_W': 2}, weights=blob)
    det.OBJ_THRESHOLD = 0.2
    tw = synth.synth_tiny_weights(512)
    cfg = {"model_tracker": {"name": "TinyTracker", "lstm_units": 512, "sequence_length": T},
           "train": {"pool": "Global", "batch_size": 4}}
    tt = TinyTracker(cfg, feature_dims=(4, 4, 512), weights=tw, ctx=det.model.ctx)
    frames = np.stack([synth.synth_clip(T, H, W, 2, seed=90 + i) for i in range(n_seq)])
    got = tt.track_sequences(frames, det).cpu().numpy()
    layers, _ = orc.parse_darknet_blob(blob, C)
    flat = frames.reshape(n_seq * T, H, W, 3)
    net, _, taps = orc.yolov2_forward(orc.normalize_u8(flat), layers, taps=("act_13",))
    det4 = np.zeros((n_seq * T, 4), dtype=np.float32)
    nbox = 0
    for f in range(n_seq * T):
        rows, _ = orc.decode_netout(net[f], 0.2, 0.45, ANCHORS, C)
        if len(rows):
            det4[f] = rows[int(np.argmax(rows[:, 6])), :4]
            nbox += 1
    assert nbox > 0, "vacuous without detections"
    ref = orc.tinytracker_forward(taps["act_13"].reshape(n_seq, T, 4, 4, 512), det4.reshape(n_seq, T, 4), tw)
    np.testing.assert_allclose(got, ref, rtol=1e-3, atol=1e-4)


def test_multi_process_path_on_one_gpu(tmp_path):
    """Functional test of the N>1 code path on the single GPU of this box: 2
    processes (gloo, both on cuda:0) each run detect+track on their shard of 4
    clips and gather; the table and the global ids must equal the 1-process result."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    worker = tmp_path / "w.py"
    worker.write_text(r'''
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
import object_tracking_amd
from parallel import gather_detections, shard_range, init_from_env
from utility import synth
from models_tracking.MultiObjDetTracker import MultiObjDetTracker
rank, world, _ = init_from_env()
H = W = 64; T = 3; N = 4; C = 12
class Trk(MultiObjDetTracker):
    IMAGE_H, IMAGE_W = H, W
    GRID_H, GRID_W = 2, 2
    SEQUENCE_LENGTH = T
    LOAD_MODEL = False
    OBJ_THRESHOLD = 0.3
tw = synth.synth_tracker_weights(C); tw["out_kernel"] = tw["out_kernel"] * 40.0; tw["out_bias"][4::17] = 1.5
trk = Trk(detector_weights=synth.synth_darknet_blob(C), tracker_weights=tw)
frames = np.stack([synth.synth_clip(T, H, W, 2, seed=300 + i) for i in range(N)])
a, b = shard_range(N, rank, world)
out = gather_detections(trk.track_clips(frames[a:b]))
full = gather_detections.__globals__["global_track_ids"]
ref = trk.track_clips(frames)
# a rank runs its shard at another batch size than the single-process reference, so layers may take another
# form (direct / Winograd, split-K factor): box values agree to rounding, everything discrete is exact
ok = (torch.allclose(out["boxes"], ref["boxes"], rtol=1e-4, atol=1e-5) and torch.equal(out["boxes"][..., 5], ref["boxes"][..., 5])
      and torch.equal(out["boxes"][..., 7], ref["boxes"][..., 7]) and torch.equal(out["counts"], ref["counts"])
      and torch.equal(out["ids"], ref["ids"]) and torch.equal(out["gids"], full(ref["ids"], ref["nids"]))
      and int(ref["counts"].sum()) > 0)
print("RANK", rank, "OK" if ok else "MISMATCH", int(ref["counts"].sum()), flush=True)
dist.barrier(); dist.destroy_process_group()
sys.exit(0 if ok else 1)
''')
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29741", WORLD_SIZE="2", DT_ONE_DEVICE="1",
               DT_DIST_BACKEND="gloo")
    procs = [subprocess.Popen([sys.executable, str(worker), root], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, o[-2000:]
        assert "RANK %d OK" % r in o, o[-2000:]


def test_heatmap_kernels_match_reference_golden(ctx, golden_dir):
    d = np.load(os.path.join(golden_dir, "heatmap.npz"))
    heat = ctx.heatmap_from_boxes(dev(d["box4"], ctx), 32).cpu().numpy()
    assert np.array_equal(heat, d["heat"])
    rects = ctx.rect_from_heatmap(dev(d["soft"].reshape(64, -1), ctx), 32, 0.75).cpu().numpy()
    assert np.array_equal(rects, d["rects"])
    from utility.utils import generate_heatmap_feat, generate_rectangle_from_heatmap
    cx, cy, w, h = [float(v) for v in d["box4"][0]]
    assert np.array_equal(generate_heatmap_feat(cx - w / 2.0, cy - h / 2.0, w, h, 32), d["heat"][0])
    assert generate_rectangle_from_heatmap(d["soft"][5], 0.75, 32) == tuple(d["rects"][5].tolist())


def test_tinyheatmap_forward_vs_oracle(ctx):
    """TinyHeatmapTracker graph: D = 512 + 1024, Dense(1024, sigmoid) through the MFMA head."""
    from models_tracking.TinyHeatmapTracker import TinyHeatmapTracker
    n_seq, T, hs = 5, 4, 32
    tw = synth.synth_heatmap_weights(512, hs)
    cfg = {"model_tracker": {"name": "TinyHeatmapTracker", "lstm_units": 512, "sequence_length": T, "heatmap_size": hs},
           "train": {"pool": "Global", "batch_size": 4}}
    tt = TinyHeatmapTracker(cfg, feature_dims=(13, 13, 512), weights=tw, ctx=ctx)
    rs = np.random.RandomState(21)
    feat = rs.randn(n_seq, T, 13, 13, 512).astype(np.float32)
    box = rs.rand(n_seq * T, 4).astype(np.float32) * [0.8, 0.8, 0.4, 0.4] + [0.1, 0.1, 0.05, 0.05]
    det = orc.heatmap_from_boxes(box.astype(np.float32), hs).reshape(n_seq, T, hs * hs)
    got = tt.model_tracker.predict([feat, det])
    ref = orc.tinytracker_forward(feat, det, tw, pool="Global")
    assert got.shape == (n_seq, T, hs * hs)
    np.testing.assert_allclose(got, ref, rtol=1e-4, atol=2e-5)
    # rectangles read back from the predicted maps agree with the oracle where no cell sits on the threshold
    r_got = ctx.rect_from_heatmap(dev(got.reshape(n_seq * T, -1), ctx), hs, 0.75).cpu().numpy()
    r_ref = orc.rect_from_heatmap(ref.reshape(n_seq * T, -1), hs, 0.75)
    safe = np.all(np.abs(ref.reshape(n_seq * T, -1) - 0.75) > 1e-4, axis=1)
    assert np.array_equal(r_got[safe], r_ref[safe]) and safe.any()


@pytest.mark.parametrize("n,Hs,Ws,Hd,Wd", [(3, 37, 53, 64, 64), (2, 480, 640, 416, 416), (1, 1080, 1920, 416, 416),
                                           (2, 100, 60, 416, 416), (1, 416, 416, 416, 416)])
def test_ingest_resize_bit_exact_vs_oracle(ctx, n, Hs, Ws, Hd, Wd):
    rs = np.random.RandomState(Hs + Wd)
    src = rs.randint(0, 256, size=(n, Hs, Ws, 3)).astype(np.uint8)
    got = ctx.ingest_resize(dev(src, ctx), Hd, Wd).cpu().numpy()
    assert np.array_equal(got, orc.resize_bilinear_u8(src, Hd, Wd))


def test_predict_drop_in_on_image_files(ctx, tmp_path):
    """KerasYOLO.predict(input_path, output_path): decode file -> device resize -> x/255 fused
    conv stack -> decode/NMS -> annotated file written; boxes equal to the oracle chain."""
    from PIL import Image
    from models_detection.KerasYOLO import KerasYOLO
    C = 12
    blob = synth.synth_darknet_blob(C, head_std=0.3)
    det = KerasYOLO({'LABELS': [str(i) for i in range(C)], 'BATCH_SIZE': 1, 'IMAGE_H': 96, 'IMAGE_W': 96,
                     'GRID_H': 3, 'GRID_W': 3}, weights=blob)
    det.OBJ_THRESHOLD = 0.2
    rgb = np.random.RandomState(4).randint(0, 256, size=(120, 200, 3)).astype(np.uint8)
    src, dst = str(tmp_path / "in.png"), str(tmp_path / "out.png")
    Image.fromarray(rgb).save(src)
    boxes = det.predict(src, dst)
    assert os.path.exists(dst) and Image.open(dst).size == (200, 120)
    bgr = np.ascontiguousarray(rgb[..., ::-1])
    frame = orc.resize_bilinear_u8(bgr[None], 96, 96)
    layers, _ = orc.parse_darknet_blob(blob, C)
    net, _, _ = orc.yolov2_forward(orc.normalize_u8(frame), layers)
    rows, _ = orc.decode_netout(net[0], 0.2, 0.45, ANCHORS, C)
    assert len(boxes) == len(rows) and len(rows) > 0
    got = np.array([[b.x, b.y, b.w, b.h] for b in boxes], dtype=np.float32)
    assert [b.get_label() for b in boxes] == [int(v) for v in rows[:, 5]]
    assert box_err(got, rows) < 1e-3


def test_track_608_vs_oracle(ctx):
    """BASELINE.json configs[4] shape, quick version (tests/test_gpu_configs.py runs 4 clips x 30 frames): 608x608 ->
    19x19 grid (1805 cells, the decode kernel's largest supported grid), C=12, one 2-frame clip through detector +
    ConvLSTM + 1x1 + decode, head calibrated like bench.py's (peaky class scores, ~128 candidates per frame)."""
    import bench
    H = W = 608
    T, C = 2, 12
    frames = torch.from_numpy(synth.synth_clip(T, H, W, 4, seed=608)[None]).to(ctx.device)
    trk, blob, tw = bench.build_tracker(H, W, T, 128, frames)
    layers, _ = orc.parse_darknet_blob(blob, C)
    res = trk.track_clips(frames)
    ref_trk, _ = orc.tracker_forward(orc.normalize_u8(frames[0].cpu().numpy()), layers, tw)
    got = res["netout"][0].cpu().numpy()
    assert got.shape == (T, 19, 19, 5, 17)
    assert chan_err(flat_c(got), flat_c(ref_trk)) < NET_TOL
    # decode with gap thresholds per frame (no oracle score / candidate-pair IoU within float noise of its threshold):
    # unconditional comparison of the box set, order, labels and coordinates
    thr = np.array([gap_threshold(oracle_scores(ref_trk[t], C).ravel(), 0.5, 0.45, 0.55) for t in range(T)], dtype=np.float32)
    nms = np.zeros(T, dtype=np.float32)
    for t in range(T):
        cand, _ = orc.decode_netout(ref_trk[t], thr[t], 2.0, ANCHORS, C)
        A = np.repeat(cand[:, None, :4], len(cand), 1).reshape(-1, 4); B = np.repeat(cand[None, :, :4], len(cand), 0).reshape(-1, 4)
        nms[t] = gap_threshold(iou_rows(A, B), 0.45, 0.40, 0.50)
    r = trk.model.ctx.decode(res["netout"][0], thr, nms, ANCHORS, C)
    cnt = r["counts"].cpu().numpy()
    for t in range(T):
        rows, _ = orc.decode_netout(ref_trk[t], thr[t], nms[t], ANCHORS, C)
        assert len(rows) == cnt[t]
        gb = r["boxes"][t, :cnt[t]].cpu().numpy()
        assert np.array_equal(gb[:, 7], rows[:, 7]) and np.array_equal(gb[:, 5], rows[:, 5])
        assert box_err(gb, rows) < 1e-3
        assert len(rows) == 0 or iou_rows(gb[:, :4], rows[:, :4]).min() >= 0.999
    assert cnt.sum() > 20


def test_decode_nonsquare_grid_and_three_anchors(ctx):
    """GH != GW and NB = 3: row/col/anchor decomposition of the cell index"""
    rs = np.random.RandomState(77)
    GH, GW, NB, C = 5, 9, 3, 7
    g = rs.randn(GH, GW, NB, 5 + C).astype(np.float32)
    g[..., 4] -= 3.0
    for k in range(14):
        r, c, b = rs.randint(GH), rs.randint(GW), rs.randint(NB)
        g[r, c, b, 4] = 4.0 + rs.rand()
        g[r, c, b, 5 + rs.randint(C)] += 10.0 + rs.rand()
    anchors = [1.0, 1.5, 2.5, 2.0, 4.0, 6.0]
    rows, _ = orc.decode_netout(g, 0.4, 0.45, anchors, C)
    r = ctx.decode(dev(g[None], ctx), 0.4, 0.45, anchors, C)
    n = int(r["counts"][0])
    assert n == len(rows) and n > 5
    got = r["boxes"][0, :n].cpu().numpy()
    assert np.array_equal(got[:, 7], rows[:, 7]) and np.array_equal(got[:, 5], rows[:, 5])
    np.testing.assert_allclose(got[:, :7], rows[:, :7], rtol=2e-6, atol=1e-6)


def test_error_paths_fail_loudly(ctx):
    import mi355_dt
    x = torch.zeros((1, 8, 8, 48), dtype=torch.float32, device=ctx.device)          # Cin not a multiple of 32
    with pytest.raises(mi355_dt.NativeError):
        ctx.conv2d(x, np.zeros((3, 3, 48, 32), dtype=np.float32))
    x = torch.zeros((1, 7, 8, 32), dtype=torch.float32, device=ctx.device)          # odd H with pooling
    with pytest.raises(mi355_dt.NativeError):
        ctx.conv2d(x, np.zeros((3, 3, 32, 32), dtype=np.float32), pool=1)
    big = torch.zeros((1, 48, 48, 5, 17), dtype=torch.float32, device=ctx.device)   # 11520 cells > the 13-bit cell field (8192)
    with pytest.raises(mi355_dt.NativeError):
        ctx.decode(big, 0.5, 0.45, ANCHORS, 12)
    c2 = mi355_dt.Context()
    with pytest.raises(mi355_dt.NativeError):                                       # image side not a multiple of 32
        c2.detector_config(100, 416, 5, 12, ANCHORS)
    c2.detector_config(64, 64, 5, 12, ANCHORS)
    frames = torch.zeros((1, 64, 64, 3), dtype=torch.uint8, device=ctx.device)
    with pytest.raises(mi355_dt.NativeError):                                       # weights not loaded
        c2.detect_forward(frames)
    with pytest.raises(mi355_dt.NativeError):                                       # short weight stream
        c2.load_darknet_weights(np.zeros(1000, dtype=np.float32))
    with pytest.raises(mi355_dt.NativeError):                                       # tracker head not loaded
        c2.track_forward(torch.zeros((1, 2, 64, 64, 3), dtype=torch.uint8, device=ctx.device))
    with pytest.raises(mi355_dt.NativeError):                                       # unsupported frame dtype
        c2.detect_forward(torch.zeros((1, 64, 64, 3), dtype=torch.float16, device=ctx.device))
    c2.close()


@pytest.mark.parametrize("cfg", ["2", "3"])
def test_conv_tile_configurations_forced(ctx, cfg, monkeypatch):
    """The 8-wave 256x128 and 16-wave 256x256 tiles are chosen by a batch-size heuristic; force
    them here (DT_CONV_CFG) so that plain / pooled / pooled+skip / ConvLSTM-gate epilogues, M and N
    edges are checked at small shapes."""
    monkeypatch.setenv("DT_CONV_CFG", cfg)
    rs = np.random.RandomState(int(cfg))
    for (B, H, W, Cin, k, Cout, pool) in [(2, 20, 18, 32, 3, 256, 0), (1, 26, 26, 64, 3, 512, 1), (2, 12, 8, 64, 3, 256, 2),
                                          (3, 13, 13, 96, 1, 300, 0)]:
        x = rs.randn(B, H, W, Cin).astype(np.float32)
        w = (rs.randn(k, k, Cin, Cout) * np.sqrt(2.0 / (k * k * Cin))).astype(np.float32)
        b = rs.randn(Cout).astype(np.float32)
        ref = orc.conv2d(x, w, b)
        ref = np.where(ref > 0, ref, 0.1 * ref).astype(np.float32)
        got = ctx.conv2d(dev(x, ctx), w, b, leaky_slope=0.1, pool=pool)
        if pool == 0:
            assert relerr(got.cpu().numpy(), ref) < 2e-5
        elif pool == 1:
            assert relerr(got.cpu().numpy(), orc.maxpool2(ref)) < 2e-5
        else:
            assert relerr(got[0].cpu().numpy(), ref) < 2e-5 and relerr(got[1].cpu().numpy(), orc.maxpool2(ref)) < 2e-5
    # ConvLSTM gates epilogue on the forced tile (U = 64 -> N = 256)
    B, H, W, Cx, U = 2, 9, 7, 32, 64
    x = rs.randn(B, H, W, Cx).astype(np.float32)
    h = (rs.randn(B, H, W, U) * .5).astype(np.float32); c = rs.randn(B, H, W, U).astype(np.float32)
    Wk = (rs.randn(3, 3, Cx, 4 * U) * .05).astype(np.float32); Uk = (rs.randn(3, 3, U, 4 * U) * .05).astype(np.float32)
    bb = (rs.randn(4 * U) * .1).astype(np.float32)
    rh, rc = orc.convlstm_step(x, h, c, Wk, Uk, bb)
    gh, gc = ctx.convlstm_step(dev(x, ctx), dev(h, ctx), dev(c, ctx), Wk, Uk, bb)
    np.testing.assert_allclose(gh.cpu().numpy(), rh, rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(gc.cpu().numpy(), rc, rtol=1e-4, atol=2e-5)


# ---- training-target encoding + sequence generators ----------------------
TARGET_CASES = ["g13_c12", "g13_c12_aug", "g19_c20_wrap", "g13_dense_cell"]


def _encode_on_device(ctx, objs, counts, dims, aug, G, C, IM, TBB, anchors):
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(ctx.device)
    y, b = ctx.encode_targets(t(objs), t(counts), t(dims), None if aug is None else t(aug), G, G, 5, C, IM, IM, TBB,
                              anchors)
    return y.cpu().numpy(), b.cpu().numpy()


@pytest.mark.gpu
@pytest.mark.parametrize("name", TARGET_CASES)
def test_encode_targets_bit_exact_vs_reference_golden(ctx, golden_dir, name):
    """dt_encode_targets against the reference's own statements (preprocessing.py:171-188,214-293)."""
    z = np.load(os.path.join(golden_dir, "targets.npz"))
    G, IM, C, TBB = [int(v) for v in z[name + "/cfg"]]
    aug = z[name + "/aug"] if (name + "/aug") in z.files else None
    y, b = _encode_on_device(ctx, z[name + "/objs"], z[name + "/counts"], z[name + "/dims"], aug, G, C, IM, TBB,
                             z["anchors"])
    assert np.array_equal(y, z[name + "/y"])
    assert np.array_equal(b, z[name + "/b"])


@pytest.mark.gpu
@pytest.mark.parametrize("n,cap,G,C,TBB,use_aug", [(64, 600, 13, 12, 50, True), (7, 3, 19, 80, 10, False),
                                                   (1, 1, 13, 1, 1, False), (480, 20, 13, 80, 50, True)])
def test_encode_targets_bit_exact_vs_oracle(ctx, n, cap, G, C, TBB, use_aug):
    """Larger and ragged batches (more objects than one pass of the kernel holds, empty frames; the
    480-frame case is above the size where the zero fill moves from the kernel to memsets)."""
    rs = np.random.RandomState(n * 7 + cap)
    IM = 32 * G
    dims = np.stack([rs.randint(200, 2000, n), rs.randint(200, 1200, n)], 1).astype(np.int32)
    counts = rs.randint(0, cap + 1, n).astype(np.int32)
    counts[0] = cap
    if n > 2:
        counts[1] = 0
    objs = np.zeros((n, cap, 5), dtype=np.int32)
    for i in range(n):
        w, h = dims[i]
        x0 = rs.randint(-20, w, cap); y0 = rs.randint(-20, h, cap)
        objs[i, :, 0] = x0; objs[i, :, 1] = y0
        objs[i, :, 2] = x0 + rs.randint(-5, w // 2, cap); objs[i, :, 3] = y0 + rs.randint(-5, h // 2, cap)
        objs[i, :, 4] = rs.randint(-1, C, cap)
    aug = None
    if use_aug:
        sc = rs.uniform(size=n) / 10. + 1.
        aug = np.stack([sc, np.floor(rs.uniform(size=n) * (sc - 1) * dims[:, 0]),
                        np.floor(rs.uniform(size=n) * (sc - 1) * dims[:, 1]), rs.binomial(1, .5, n)], 1).astype(np.float64)
    anchors = np.asarray(ANCHORS, dtype=np.float64)
    y, b = _encode_on_device(ctx, objs, counts, dims, aug, G, C, IM, TBB, anchors)
    yo, bo = orc.encode_targets(objs, counts, dims, aug, G, G, 5, C, IM, IM, TBB, anchors)
    assert np.array_equal(y, yo) and np.array_equal(b, bo)
    assert (y[..., 4] == 1).sum() > 0 or cap == 1


@pytest.mark.gpu
def test_batch_generators_on_image_files(ctx, tmp_path):
    """BatchGenerator / BatchSequenceGenerator1 (preprocessing.py:195-371, augment=False): frames through
    the device resize, targets through dt_encode_targets; checked against the oracle chain."""
    from PIL import Image
    from utility.preprocessing import BatchGenerator, BatchSequenceGenerator1
    from utility.utils import normalize
    labels = ["car", "person"]
    cfg = dict(IMAGE_H=96, IMAGE_W=96, GRID_H=3, GRID_W=3, BOX=5, CLASS=2, LABELS=labels, ANCHORS=ANCHORS,
               BATCH_SIZE=2, TRUE_BOX_BUFFER=6, SEQUENCE_LENGTH=2)
    rs = np.random.RandomState(12)
    recs, raw = [], []
    for folder, n in (("a/", 3), ("b/", 2)):
        for k in range(n):
            h, w = int(rs.randint(100, 180)), int(rs.randint(120, 260))
            rgb = rs.randint(0, 256, size=(h, w, 3)).astype(np.uint8)
            path = str(tmp_path / ("%s_%d.png" % (folder[0], k)))
            Image.fromarray(rgb).save(path)
            raw.append(np.ascontiguousarray(rgb[..., ::-1]))
            objs = [{"name": labels[int(rs.randint(0, 2))] if j else "tree", "xmin": int(rs.randint(0, w // 2)),
                     "ymin": int(rs.randint(0, h // 2)), "xmax": int(rs.randint(w // 2, w)),
                     "ymax": int(rs.randint(h // 2, h))} for j in range(int(rs.randint(1, 5)))]
            recs.append({"folder": folder, "filename": path, "width": w, "height": h, "object": objs})

    def expect(rec_list):
        x = np.stack([orc.resize_bilinear_u8(raw[recs.index(r)][None], 96, 96)[0][:, :, ::-1] for r in rec_list])
        from utility.preprocessing import pack_objects
        o, c, d = pack_objects(rec_list, labels)
        y, b = orc.encode_targets(o, c, d, None, 3, 3, 5, 2, 96, 96, 6, np.asarray(ANCHORS, dtype=np.float64))
        return normalize(x), b, y

    gen = BatchGenerator(list(recs), cfg, shuffle=False, augment=False, norm=normalize, ctx=ctx)
    assert len(gen) == 3
    for idx in range(len(gen)):
        (x, b), y = gen[idx]
        lo, hi = (idx * 2, idx * 2 + 2) if idx * 2 + 2 <= len(recs) else (len(recs) - 2, len(recs))
        ex, eb, ey = expect(recs[lo:hi])
        assert x.shape == (2, 96, 96, 3) and x.dtype == np.float64 and np.array_equal(x, ex)
        assert b.shape == (2, 1, 1, 1, 6, 4) and np.array_equal(b.reshape(2, 6, 4), eb)
        assert y.shape == (2, 3, 3, 5, 7) and np.array_equal(y, ey)
    seq = BatchSequenceGenerator1(list(recs), cfg, shuffle=False, augment=False, norm=normalize, ctx=ctx)
    # windows of 2 inside one folder: a0a1, a1a2, (a2b0 slides to) b0b1, b0b1
    assert [[recs.index(r) for r in win] for win in seq.images] == [[0, 1], [1, 2], [3, 4], [3, 4]]
    (x, b), (y1, y2) = seq[0]
    ex, eb, ey = expect([recs[0], recs[1], recs[1], recs[2]])
    assert x.shape == (2, 2, 96, 96, 3) and np.array_equal(x.reshape(4, 96, 96, 3), ex)
    assert b.shape == (2, 2, 1, 1, 1, 6, 4) and np.array_equal(b.reshape(4, 6, 4), eb)
    assert y1 is y2 and y1.shape == (2, 2, 3, 3, 5, 7) and np.array_equal(y1.reshape(4, 3, 3, 5, 7), ey)
    with pytest.raises(NotImplementedError):
        BatchGenerator(list(recs), cfg, augment=True, ctx=ctx)


# ---- Winograd F(2x2,3x3) form of the wide 3x3 layers (csrc/winograd.hip) ------------------
@pytest.fixture(params=[6, 4, 2], ids=["F6x6", "F4x4", "F2x2"])
def wino_all(monkeypatch, request):
    """DT_WINO=2: every 3x3 layer the transforms support goes through the Winograd path, at any size
    (the default policy only takes it for Cin >= 128, Cout >= 256 and >= 512 tiles per launch), once as
    F(6x6,3x3) (the default tile), F(4x4,3x3) and F(2x2,3x3).  Returns the output tile size."""
    monkeypatch.setenv("DT_WINO", "2")
    monkeypatch.setenv("DT_WINO_TILE", str(request.param))
    return request.param


@pytest.mark.parametrize("B,H,W,Cin,Cout,pool", [
    (2, 13, 13, 64, 128, 0),     # odd H, W: 7x7 tiles cover 14x14
    (3, 8, 12, 32, 64, 1),       # pooled output only: a tile is a pooling window
    (2, 6, 10, 64, 160, 2),      # pooled + full resolution (conv_13's skip tap)
    (1, 7, 5, 96, 36, 0),        # ragged everything, N a multiple of 4 only
    (5, 26, 26, 32, 32, 0),      # more tiles than one row tile of the GEMM
    (1, 13, 13, 1280, 256, 0),   # conv_22's Cin, 256-wide column tile
    (6, 13, 13, 64, 128, 0),     # 2x2 frame mosaic with zero separators (tiles straddle frames), ragged last group
    (18, 26, 26, 32, 64, 0),     # 4x4 mosaic for F(4x4), 2x2 for F(6x6) (F(2x2) tiles 26 exactly), ragged last group
    (11, 13, 13, 32, 64, 0),     # 3x3 mosaic for F(6x6): 3*14 = 7 tiles of 6, ragged last group
    (9, 5, 7, 32, 64, 0),        # non-square mosaic
])
def test_conv2d_winograd_vs_oracle(ctx, wino_all, B, H, W, Cin, Cout, pool):
    rs = np.random.RandomState(B * 1000 + H + Cin + Cout)
    x = rs.randn(B, H, W, Cin).astype(np.float32)
    w = (rs.randn(3, 3, Cin, Cout) * np.sqrt(2.0 / (9 * Cin))).astype(np.float32)
    b = rs.randn(Cout).astype(np.float32)
    ref = orc.conv2d(x, w, b)
    ref = np.where(ref > 0, ref, ref * np.float32(0.1)).astype(np.float32)
    ctx.profile_reset(); ctx.profile_enable(True)
    got = ctx.conv2d(dev(x, ctx), w, b, leaky_slope=0.1, pool=pool)
    ctx.profile_enable(False)
    assert ctx.profile_read("wino_input")["launches"] == 1 and ctx.profile_read("wino_output")["launches"] == 1
    tol = {2: 2e-5, 4: 1e-4, 6: 2e-4}[wino_all]    # F(4x4) / F(6x6): ~15x / ~20x the rounding error of the direct form
    if pool == 0:
        assert relerr(got.cpu().numpy(), ref) < tol
    elif pool == 1:
        assert relerr(got.cpu().numpy(), orc.maxpool2(ref)) < tol
    else:
        assert relerr(got[0].cpu().numpy(), ref) < tol
        assert relerr(got[1].cpu().numpy(), orc.maxpool2(ref)) < tol


def test_conv2d_winograd_detects_transpose(ctx, wino_all):
    """one-hot taps: every Winograd position / tile offset / channel map must line up exactly
    (values are small integers, the transforms are exact on them)."""
    B, H, W, Cin, Cout = 2, 9, 11, 32, 96
    x = (np.arange(B * H * W * Cin, dtype=np.float32).reshape(B, H, W, Cin) % 251)
    w = np.zeros((3, 3, Cin, Cout), dtype=np.float32)
    for n in range(Cout):
        w[n % 3, (n // 3) % 3, (n * 7) % Cin, n] = 1.0
    got = ctx.conv2d(dev(x, ctx), w, None, leaky_slope=1.0, pool=0).cpu().numpy()
    ref = orc.conv2d(x, w)
    if wino_all == 2:
        assert np.array_equal(got, ref)          # halves and small integers only: exact
    else:
        assert np.abs(got - ref).max() < 0.05    # values are integers up to 250: any misplaced tap is off by >= 1


def test_convlstm_step_winograd_vs_oracle(ctx, wino_all):
    atol = {2: 2e-5, 4: 5e-5, 6: 1e-4}[wino_all]      # state values are O(1): F(4x4) / F(6x6) round ~15x / ~20x coarser
    ctx.profile_reset(); ctx.profile_enable(True)
    test_convlstm_step_vs_oracle(ctx, atol=atol)
    ctx.profile_enable(False)
    assert ctx.profile_read("wino_output")["launches"] == 2      # input projection + gate step
    # 13x13 grids of 6 clips: the 2x2 frame mosaic (ragged last group) through the gate-update transform
    rs = np.random.RandomState(19)
    B, H, W, Cx, U = 6, 13, 13, 64, 32
    x = rs.randn(B, H, W, Cx).astype(np.float32)
    h = (rs.randn(B, H, W, U) * .5).astype(np.float32); c = rs.randn(B, H, W, U).astype(np.float32)
    Wk = (rs.randn(3, 3, Cx, 4 * U) * .05).astype(np.float32); Uk = (rs.randn(3, 3, U, 4 * U) * .05).astype(np.float32)
    b = rs.randn(4 * U).astype(np.float32) * .1
    rh, rc = orc.convlstm_step(x, h, c, Wk, Uk, b)
    gh, gc = ctx.convlstm_step(dev(x, ctx), dev(h, ctx), dev(c, ctx), Wk, Uk, b)
    np.testing.assert_allclose(gh.cpu().numpy(), rh, rtol=1e-4, atol=max(atol, 5e-5))
    np.testing.assert_allclose(gc.cpu().numpy(), rc, rtol=1e-4, atol=max(atol, 5e-5))


def test_detector_winograd_vs_oracle(ctx, wino_all):
    test_detector_forward_vs_oracle_small(ctx, 64, 64, 12, 3)
    test_detector_full_size_one_frame_vs_oracle(ctx)


def test_tracker_winograd_boxes_and_ids_vs_oracle(ctx, wino_all):
    test_track_forward_vs_oracle_small(ctx)
    test_track_clips_boxes_and_ids_vs_oracle(ctx)


def test_winograd_default_policy_engages_on_wide_layers(ctx):
    """Default policy: 40 frames of 13x13x512 -> 640 F(4x4,3x3) tiles (>= 64) takes the Winograd path and
    agrees with the direct MFMA form of the same layer (DT_WINO=0) to rounding; 2 frames stay direct."""
    rs = np.random.RandomState(77)
    x = rs.randn(40, 13, 13, 512).astype(np.float32)
    w = (rs.randn(3, 3, 512, 256) * np.sqrt(2.0 / (9 * 512))).astype(np.float32)
    b = rs.randn(256).astype(np.float32)
    ctx.profile_reset(); ctx.profile_enable(True)
    got = ctx.conv2d(dev(x, ctx), w, b, leaky_slope=0.1, pool=0)
    ctx.profile_enable(False)
    assert ctx.profile_read("wino_input")["launches"] == 1
    ctx.profile_reset(); ctx.profile_enable(True)
    small = ctx.conv2d(dev(x[:2], ctx), w, b, leaky_slope=0.1, pool=0)
    ctx.profile_enable(False)
    assert ctx.profile_read("wino_input")["launches"] == 0 and relerr(small.cpu().numpy(), got[:2].cpu().numpy()) < 1e-4
    os.environ["DT_WINO_WS_GB"] = "0.01"      # workspace cap below what this launch needs -> direct form, not an error
    try:
        ctx.profile_reset(); ctx.profile_enable(True)
        capped = ctx.conv2d(dev(x, ctx), w, b, leaky_slope=0.1, pool=0)
        ctx.profile_enable(False)
        assert ctx.profile_read("wino_input")["launches"] == 0 and relerr(capped.cpu().numpy(), got.cpu().numpy()) < 1e-4
    finally:
        del os.environ["DT_WINO_WS_GB"]
    os.environ["DT_WINO"] = "0"
    try:
        ctx.profile_reset(); ctx.profile_enable(True)
        direct = ctx.conv2d(dev(x, ctx), w, b, leaky_slope=0.1, pool=0)
        ctx.profile_enable(False)
        assert ctx.profile_read("wino_input")["launches"] == 0
    finally:
        del os.environ["DT_WINO"]
    assert relerr(got.cpu().numpy(), direct.cpu().numpy()) < 1e-4
    ref = orc.conv2d(x[:2], w, b)
    ref = np.where(ref > 0, ref, ref * np.float32(0.1)).astype(np.float32)
    assert relerr(got[:2].cpu().numpy(), ref) < 1e-4


# ---- conv_23 folded into the ConvLSTM input projection (DT_TRK_MERGE, network.hip:build_merged_xproj) ----------------
@pytest.mark.parametrize("H,W,n_clips,T", [(96, 128, 6, 5), (32, 32, 40, 3), (416, 416, 2, 4)])
def test_tracker_merged_input_projection(ctx, monkeypatch, H, W, n_clips, T):
    """x_bbox = conv_23(conv_feat) is linear, so Wx * [x_bbox | conv_feat] + b is ONE 3x3 convolution of conv_feat with merged weights and a bias
    that knows which taps lie inside the image.  The merged form (default where the projection's Winograd path runs; conv_23 is not launched when
    the detector's grid is not asked for) against the two-step form (DT_TRK_MERGE=0) and against the oracle, with and without want_det, on grids
    with every border case (3x4), with one cell (1x1: all four borders at once) and at 13x13."""
    monkeypatch.setenv("DT_WINO", "2")                     # the projection's Winograd path at any size
    trk, blob, tw = _tracker(H, W, T, 12)
    c = trk.model.ctx
    frames = np.stack([synth.synth_clip(T, H, W, 2, seed=40 + i) for i in range(n_clips)])
    d = dev(frames, c)
    outs = {}
    for merge in ("1", "0"):
        monkeypatch.setenv("DT_TRK_MERGE", merge)
        c.reload_policy()
        c.profile_reset(); c.profile_enable(True)
        t_only = c.track_forward(d, want_det=False)
        n23 = c.profile_read("conv_gemm_s3:conv_23")["launches"] + c.profile_read("conv_igemm:conv_23")["launches"]
        t_both, det = c.track_forward(d, want_det=True)
        c.profile_enable(False)
        assert c.profile_read("convlstm_xproj:merged_conv23")["launches"] == (2 if merge == "1" else 0)
        assert n23 == (0 if merge == "1" else 1)           # merged and nobody reads x_bbox: conv_23 is not launched
        assert torch.equal(t_only, t_both)
        outs[merge] = (t_both.cpu().numpy(), det.cpu().numpy())
    assert np.array_equal(outs["1"][1], outs["0"][1])      # the detector's grid itself is conv_23's output either way
    assert chan_err(flat_c(outs["1"][0]), flat_c(outs["0"][0])) < 1e-4         # two roundings of the same network through T recurrent steps (measured 5e-5)
    layers, used = orc.parse_darknet_blob(blob, 12)
    ref = np.stack([orc.tracker_forward(orc.normalize_u8(frames[i]), layers, tw)[0] for i in range(min(2, n_clips))])
    assert chan_err(flat_c(outs["1"][0][:ref.shape[0]]), flat_c(ref)) < 3e-4


# ---- DT_PIN: kernel selection independent of the batch a call carries -----------------------
def test_pinned_policy_is_batch_independent(ctx):
    """Under parallel.pinned_policy (DT_PIN=1) a frame / a clip gets the same bits whatever batch it travels in: the detector on 12 frames
    against the same frames in calls of 5 + 7 and one by one, the tracker on 4 clips against 3 + 1 -- torch.equal.  (Under the default
    policy the same comparison differs at rounding level: other kernels are selected for other batch sizes -- also checked, so that the
    test would notice if it stopped exercising anything.)"""
    from parallel import pinned_policy
    H, W, T, C = 96, 128, 3, 12
    trk, blob, tw = _tracker(H, W, T, C)
    c = trk.model.ctx
    frames = np.stack([synth.synth_clip(T, H, W, 2, seed=90 + i) for i in range(4)])
    d = dev(frames, c)                                    # [4, T, H, W, 3]
    flat = d.reshape(4 * T, H, W, 3).contiguous()
    with pinned_policy(c):
        whole = c.detect_forward(flat)
        parts = torch.cat([c.detect_forward(flat[:5].contiguous()), c.detect_forward(flat[5:].contiguous())])
        singles = torch.cat([c.detect_forward(flat[i:i + 1].contiguous()) for i in range(4 * T)])
        assert torch.equal(whole, parts) and torch.equal(whole, singles)
        t_whole, _ = c.track_forward(d)
        t_parts = torch.cat([c.track_forward(d[:3].contiguous())[0], c.track_forward(d[3:].contiguous())[0]])
        assert torch.equal(t_whole, t_parts)
    # the policy is restored on exit; under it the single-frame calls take other kernels than the 12-frame call
    plain_whole = c.detect_forward(flat)
    plain_singles = torch.cat([c.detect_forward(flat[i:i + 1].contiguous()) for i in range(4 * T)])
    assert chan_err(flat_c(plain_whole.cpu().numpy()), flat_c(plain_singles.cpu().numpy())) < 1e-4
    assert chan_err(flat_c(plain_whole.cpu().numpy()), flat_c(whole.cpu().numpy())) < 1e-4


# ---- hipGraph replay of the launch-bound inner sequences (dt_graph_enable) ------------------
def test_graph_replay_is_bit_identical(ctx):
    """Detector trunk, ConvLSTM recurrence and LSTM sequence captured on the second call with a shape and
    replayed afterwards: same bits as the plain launches, across weight reloads and shape changes."""
    H, W, T, n_clips, C = 64, 96, 4, 3, 12
    trk, blob, tw = _tracker(H, W, T, C)
    c = trk.model.ctx
    frames = np.stack([synth.synth_clip(T, H, W, 2, seed=60 + i) for i in range(n_clips)])
    d = dev(frames, c)
    plain_trk, plain_det = c.track_forward(d)
    plain_net = c.detect_forward(d[0].contiguous())
    plain2, _ = c.track_forward(d[:2].contiguous())      # another batch size rounds differently: its own reference
    c.graph_enable(True)
    try:
        for it in range(4):
            g_trk, g_det = c.track_forward(d)
            assert torch.equal(g_trk, plain_trk) and torch.equal(g_det, plain_det), "iteration %d" % it
            assert torch.equal(c.detect_forward(d[0].contiguous()), plain_net)
        assert c.profile_read("graph_capture")["launches"] >= 3      # trunk x 2 batch sizes + recurrence
        assert c.profile_read("graph_replay")["launches"] >= 6
        # another shape -> its own graphs; the first stays valid
        g2, _ = c.track_forward(d[:2].contiguous())
        assert torch.equal(g2, plain2)
        assert torch.equal(c.track_forward(d)[0], plain_trk)
        # reloading weights drops the graphs (they hold the old weight pointers)
        tw2 = dict(tw); tw2["out_bias"] = tw["out_bias"] + 0.5
        trk.model.set_weights(tw2)
        changed, _ = c.track_forward(d)
        c.graph_enable(False)
        ref_changed, _ = c.track_forward(d)
        assert torch.equal(changed, ref_changed) and not torch.equal(changed, plain_trk)
    finally:
        c.graph_enable(False)


def test_graph_replay_lstm_sequence(ctx):
    tw = synth.synth_tiny_weights(512)
    c = mi355_dt.Context()
    c.tiny_load(516, 512, tw["kernel"], tw["recurrent"], tw["bias"], tw["dense_kernel"], tw["dense_bias"])
    x = torch.randn(7, 9, 516, device=c.device)
    plain = c.tiny_sequence(x)
    c.graph_enable(True)
    for _ in range(3):
        assert torch.equal(c.tiny_sequence(x), plain)
    assert c.profile_read("graph_replay")["launches"] >= 1
    c.close()


# ---- conv_2's shape (32 -> 64 channels, pooled) through the fused F(4x4,3x3) kernel (csrc/wino4s_fused.hip) ---------
F4_KERNELS = [("0", "conv_fused")]      # wino4s_fused.hip (round 5's bf16 twin wino4b_fused.hip is gone: conv3_h2.hip took its place)


@pytest.mark.parametrize("f4b,ktag", F4_KERNELS, ids=["fp32_mfma"])
@pytest.mark.parametrize("B,H,W", [(2, 16, 16), (3, 32, 48), (1, 18, 34), (2, 2, 2), (5, 104, 104)])
def test_conv2_fused_winograd_vs_oracle(ctx, monkeypatch, B, H, W, f4b, ktag):
    """32 -> 64 channels with the 2x2 pooling epilogue through the fused kernels (forced at any size): whole and
    partial blocks, image borders, several frames; against the oracle and against the direct form."""
    monkeypatch.setenv("DT_WINO_FUSED4", "2")
    rs = np.random.RandomState(B * 100 + H + W)
    x = rs.randn(B, H, W, 32).astype(np.float32)
    w = (rs.randn(3, 3, 32, 64) * np.sqrt(2.0 / (9 * 32))).astype(np.float32)
    b = rs.randn(64).astype(np.float32)
    ref = orc.conv2d(x, w, b)
    ref = orc.maxpool2(np.where(ref > 0, ref, ref * np.float32(0.1)).astype(np.float32))
    ctx.profile_reset(); ctx.profile_enable(True)
    got = ctx.conv2d(dev(x, ctx), w, b, leaky_slope=0.1, pool=1)
    ctx.profile_enable(False)
    assert ctx.profile_read("conv_fused")["launches"] == 1 and ctx.profile_read(ktag)["launches"] == 1
    assert relerr(got.cpu().numpy(), ref) < 1e-4            # F(4x4,3x3): ~15x the direct form's rounding error
    monkeypatch.setenv("DT_WINO_FUSED4", "0")
    ctx.profile_reset(); ctx.profile_enable(True)
    direct = ctx.conv2d(dev(x, ctx), w, b, leaky_slope=0.1, pool=1)
    ctx.profile_enable(False)
    assert ctx.profile_read("conv_fused")["launches"] == 0
    assert relerr(got.cpu().numpy(), direct.cpu().numpy()) < 1e-4


@pytest.mark.parametrize("f4b,ktag", F4_KERNELS, ids=["fp32_mfma"])
def test_conv2_fused_winograd_one_hot(ctx, monkeypatch, f4b, ktag):
    """one-hot taps on small integers: any misplaced tile / channel / position is off by >= 1"""
    monkeypatch.setenv("DT_WINO_FUSED4", "2")
    B, H, W = 2, 20, 12
    x = (np.arange(B * H * W * 32, dtype=np.float32).reshape(B, H, W, 32) % 251)
    w = np.zeros((3, 3, 32, 64), dtype=np.float32)
    for n in range(64):
        w[n % 3, (n // 3) % 3, (n * 7) % 32, n] = 1.0
    got = ctx.conv2d(dev(x, ctx), w, None, leaky_slope=1.0, pool=1).cpu().numpy()
    assert np.abs(got - orc.maxpool2(orc.conv2d(x, w))).max() < 0.05


@pytest.mark.parametrize("mode", ["default", "all_winograd", "direct_h2"])
def test_detector_non_square_odd_grid_vs_oracle(ctx, monkeypatch, mode):
    """352x288 frames (grid 11x9: odd, non-square, not a multiple of any Winograd tile), 5 frames (ragged mosaic
    groups, partial 16x16-pixel blocks of the fused kernel at 176x144 / 88x72 pixels): whole detector vs oracle."""
    if mode != "default":
        monkeypatch.setenv("DT_WINO", "2")
        monkeypatch.setenv("DT_WINO_FUSED4", "2")
    if mode == "direct_h2":
        monkeypatch.setenv("DT_C3H2", "2")           # conv_2 / 3 / 5 on conv3_h2.hip at this small batch too, inside the whole network
        monkeypatch.setenv("DT_H2_MINFRAMES", "0")
    det, layers, _ = _detector(ctx, 352, 288, 12)
    frames = synth.synth_clip(5, 352, 288, 3, seed=21)
    ref_net, ref_feat, _ = orc.yolov2_forward(orc.normalize_u8(frames), layers)
    c = det.model.ctx
    c.profile_reset(); c.profile_enable(True)
    net, feat = c.detect_forward(dev(frames, c), want_feat=True)
    c.profile_enable(False)
    if mode == "direct_h2":
        # conv_2, conv_3 (+ conv_4 inside it: no conv_4 launch of any family), conv_5
        assert c.profile_read("conv_direct_h2")["launches"] == 3 and c.profile_read("conv_direct_h2:fused_1x1")["launches"] == 1
        assert not [n for n in c.profile_names() if n.endswith(":conv_4")]
        monkeypatch.setenv("DT_C3FUSE", "0")
        c.reload_policy()
        c.profile_reset(); c.profile_enable(True)
        net2, feat2 = c.detect_forward(dev(frames, c), want_feat=True)
        c.profile_enable(False)
        assert c.profile_read("conv_direct_h2:fused_1x1")["launches"] == 0 and c.profile_read("conv_igemm:conv_4")["launches"] == 1
        assert chan_err(flat_c(net.cpu().numpy()), flat_c(net2.cpu().numpy())) < 5e-5      # the same network, conv_4 in its own launch
    assert net.shape == (5, 11, 9, 5, 17)
    assert chan_err(flat_c(net.cpu().numpy()), flat_c(ref_net)) < NET_TOL
    assert chan_err(feat.cpu().numpy(), ref_feat) < NET_TOL


def test_winograd_randomized_shapes_vs_oracle(ctx, monkeypatch):
    """40 seeded random layers through the Winograd path (every tile size, forced and automatic mosaics, pooled /
    both / plain epilogues, ragged groups, grids smaller than a tile) against the oracle.  tools/wino_fuzz.py runs
    the same loop for any seed / count."""
    monkeypatch.setenv("DT_WINO", "2")
    rs = np.random.RandomState(1)
    for it in range(40):
        ts = int(rs.choice([2, 4, 6]))
        monkeypatch.setenv("DT_WINO_TILE", str(ts))
        B = int(rs.randint(1, 21))
        H = int(rs.randint(1, 21)) * (2 if rs.rand() < 0.5 else 1)
        W = int(rs.randint(1, 21)) * (2 if rs.rand() < 0.5 else 1)
        Cin = int(rs.choice([32, 64, 96])); Cout = int(rs.randint(1, 41)) * 4
        pool = int(rs.choice([0, 1, 2])) if (H % 2 == 0 and W % 2 == 0) else 0
        mos = rs.choice(["", "1", "2", "3", "4"])
        if mos:
            monkeypatch.setenv("DT_WINO_MOSAIC", mos)
        else:
            monkeypatch.delenv("DT_WINO_MOSAIC", raising=False)
        x = rs.randn(B, H, W, Cin).astype(np.float32)
        w = (rs.randn(3, 3, Cin, Cout) * np.sqrt(2.0 / (9 * Cin))).astype(np.float32)
        b = rs.randn(Cout).astype(np.float32)
        ref = orc.conv2d(x, w, b)
        ref = np.where(ref > 0, ref, ref * np.float32(0.1)).astype(np.float32)
        got = ctx.conv2d(dev(x, ctx), w, b, leaky_slope=0.1, pool=pool)
        if pool == 0:
            e = relerr(got.cpu().numpy(), ref)
        elif pool == 1:
            e = relerr(got.cpu().numpy(), orc.maxpool2(ref))
        else:
            e = max(relerr(got[0].cpu().numpy(), ref), relerr(got[1].cpu().numpy(), orc.maxpool2(ref)))
        assert e < {2: 2e-5, 4: 1e-4, 6: 3e-4}[ts], (it, ts, B, H, W, Cin, Cout, pool, mos, e)


# ---- conv_3/5/6/8 as one fused Winograd F(4x4,3x3) kernel (csrc/wino4s_fused.hip) --------------------------
@pytest.mark.parametrize("B,H,W,Cin,Cout,pool", [
    (2, 16, 16, 64, 128, 0),      # exactly one 16x16-pixel block per frame
    (3, 32, 48, 64, 128, 1),      # pooled output only (conv_5's epilogue), several blocks
    (1, 26, 22, 64, 128, 0),      # partial blocks and partial tiles on both edges
    (2, 18, 34, 128, 256, 1),     # conv_8's shape class: four channel groups, two 128-wide output halves, pooled
    (1, 13, 13, 128, 256, 0),     # odd size, smaller than a block
    (5, 104, 104, 64, 128, 0),    # conv_3's real geometry (6.5 blocks per side)
])
@pytest.mark.parametrize("f4b,ktag", F4_KERNELS, ids=["fp32_mfma"])
def test_conv_fused_f4x4_vs_oracle(ctx, monkeypatch, B, H, W, Cin, Cout, pool, f4b, ktag):
    monkeypatch.setenv("DT_WINO_FUSED4", "2")
    rs = np.random.RandomState(B * 100 + H + W + Cin)
    x = rs.randn(B, H, W, Cin).astype(np.float32)
    w = (rs.randn(3, 3, Cin, Cout) * np.sqrt(2.0 / (9 * Cin))).astype(np.float32)
    b = rs.randn(Cout).astype(np.float32)
    ref = orc.conv2d(x, w, b)
    ref = np.where(ref > 0, ref, ref * np.float32(0.1)).astype(np.float32)
    if pool:
        ref = orc.maxpool2(ref)
    ctx.profile_reset(); ctx.profile_enable(True)
    got = ctx.conv2d(dev(x, ctx), w, b, leaky_slope=0.1, pool=pool)
    ctx.profile_enable(False)
    assert ctx.profile_read("conv_fused")["launches"] == 1 and ctx.profile_read("wino_input")["launches"] == 0 and ctx.profile_read(ktag)["launches"] == 1
    assert relerr(got.cpu().numpy(), ref) < 1e-4            # F(4x4,3x3): ~15x the direct form's rounding error
    monkeypatch.setenv("DT_WINO_FUSED4", "0")
    other = ctx.conv2d(dev(x, ctx), w, b, leaky_slope=0.1, pool=pool)
    assert relerr(got.cpu().numpy(), other.cpu().numpy()) < 2e-4


# ---- conv_2 / conv_3 / conv_5 as DIRECT 3x3 convolutions in the two-term fp16 form (csrc/conv3_h2.hip) ----------------------
@pytest.mark.parametrize("B,H,W,Cin,Cout,pool", [
    (2, 16, 16, 64, 128, 0),      # exactly one 16-wide tile column, two 8-row tiles per frame; two 32-channel chunks
    (3, 32, 48, 64, 128, 1),      # pooled epilogue (conv_5), several tiles
    (1, 26, 22, 64, 128, 0),      # partial tiles on both edges
    (5, 104, 104, 64, 128, 0),    # conv_3's real geometry (6.5 tiles per row)
    (3, 32, 48, 32, 64, 1),       # conv_2's shape class: one chunk, 64 output channels (16x16 tiles, waves 4 x 1), pooled
    (2, 208, 208, 32, 64, 1),     # conv_2's real geometry
    (1, 26, 22, 32, 64, 0),
    (2, 24, 40, 64, 64, 1),       # 64 -> 64: two chunks on the 16x16-tile instance
    (2, 16, 32, 32, 128, 0),      # 32 -> 128: one chunk on the 8x16-tile instance
    (70, 13, 13, 64, 128, 0),     # frames smaller than a tile, more items than workgroups (the persistent loop turns over)
])
def test_conv3_direct_h2_vs_oracle(ctx, monkeypatch, B, H, W, Cin, Cout, pool):
    monkeypatch.setenv("DT_C3H2", "2")
    monkeypatch.setenv("DT_WINO_FUSED4", "2")
    rs = np.random.RandomState(B * 100 + H + W + Cin)
    x = rs.randn(B, H, W, Cin).astype(np.float32)
    w = (rs.randn(3, 3, Cin, Cout) * np.sqrt(2.0 / (9 * Cin))).astype(np.float32)
    b = rs.randn(Cout).astype(np.float32)
    ref = orc.conv2d(x, w, b)
    ref = np.where(ref > 0, ref, ref * np.float32(0.1)).astype(np.float32)
    if pool:
        ref = orc.maxpool2(ref)
    ctx.profile_reset(); ctx.profile_enable(True)
    got = ctx.conv2d(dev(x, ctx), w, b, leaky_slope=0.1, pool=pool)
    ctx.profile_enable(False)
    assert ctx.profile_read("conv_direct_h2")["launches"] == 1 and ctx.profile_read("conv_fused")["launches"] == 0 and ctx.profile_read("absmax")["launches"] == 1
    assert relerr(got.cpu().numpy(), ref) < 5e-6            # direct form, fp32-class products: the fp32 MFMA kernel's own level
    e = 12
    got2 = ctx.conv2d(dev(np.ldexp(x, e), ctx), w, None, leaky_slope=0.1, pool=pool).cpu().numpy()      # the scale follows the data: exact powers of two
    base = ctx.conv2d(dev(x, ctx), w, None, leaky_slope=0.1, pool=pool).cpu().numpy()
    assert np.array_equal(got2, np.ldexp(base, e))


def test_conv3_direct_h2_one_hot(ctx, monkeypatch):
    """one-hot taps on small integers: every tap / chunk / half / tile offset / channel slot of the direct kernel must line up"""
    monkeypatch.setenv("DT_C3H2", "2")
    for (B, H, W, Cin, Cout) in ((2, 20, 36, 64, 128), (2, 20, 36, 32, 64)):
        x = (np.arange(B * H * W * Cin, dtype=np.float32).reshape(B, H, W, Cin) % 251)
        w = np.zeros((3, 3, Cin, Cout), dtype=np.float32)
        for n in range(Cout):
            w[n % 3, (n // 3) % 3, (n * 7) % Cin, n] = 1.0
        ctx.profile_reset(); ctx.profile_enable(True)
        got = ctx.conv2d(dev(x, ctx), w, None, leaky_slope=1.0, pool=0).cpu().numpy()
        ctx.profile_enable(False)
        assert ctx.profile_read("conv_direct_h2")["launches"] == 1
        assert np.array_equal(got, orc.conv2d(x, w)), (Cin, Cout)      # integers below 2^11: exact in hi alone


@pytest.mark.parametrize("f4b,ktag", F4_KERNELS, ids=["fp32_mfma"])
@pytest.mark.parametrize("B,H,W,pool", [(3, 32, 48, 1), (2, 208, 208, 1), (1, 26, 22, 0)])
def test_conv2_shape_through_staged_f4x4_kernel(ctx, monkeypatch, B, H, W, pool, f4b, ktag):
    """conv_2's shape (32 -> 64 channels) through the fused F(4x4) kernels: pooled and plain epilogue, the real 208x208 geometry"""
    monkeypatch.setenv("DT_WINO_FUSED4", "2")
    rs = np.random.RandomState(B + H + W)
    x = rs.randn(B, H, W, 32).astype(np.float32)
    w = (rs.randn(3, 3, 32, 64) * np.sqrt(2.0 / (9 * 32))).astype(np.float32)
    b = rs.randn(64).astype(np.float32)
    ref = orc.conv2d(x, w, b)
    ref = np.where(ref > 0, ref, ref * np.float32(0.1)).astype(np.float32)
    if pool:
        ref = orc.maxpool2(ref)
    ctx.profile_reset(); ctx.profile_enable(True)
    got = ctx.conv2d(dev(x, ctx), w, b, leaky_slope=0.1, pool=pool)
    ctx.profile_enable(False)
    assert ctx.profile_read("conv_fused:conv_0")["launches"] == 1
    assert relerr(got.cpu().numpy(), ref) < 1e-4


@pytest.mark.parametrize("f4b,ktag", F4_KERNELS, ids=["fp32_mfma"])
def test_conv_fused_f4x4_one_hot(ctx, monkeypatch, f4b, ktag):
    """one-hot taps on small integers: every position / tile offset / channel slot of the fused kernels must line up"""
    monkeypatch.setenv("DT_WINO_FUSED4", "2")
    B, H, W, Cin, Cout = 2, 20, 36, 64, 128
    x = (np.arange(B * H * W * Cin, dtype=np.float32).reshape(B, H, W, Cin) % 251)
    w = np.zeros((3, 3, Cin, Cout), dtype=np.float32)
    for n in range(Cout):
        w[n % 3, (n // 3) % 3, (n * 7) % Cin, n] = 1.0
    got = ctx.conv2d(dev(x, ctx), w, None, leaky_slope=1.0, pool=0).cpu().numpy()
    assert np.abs(got - orc.conv2d(x, w)).max() < 0.05       # integers up to 250: a misplaced tap is off by >= 1


# ---- F(6x6) layers' GEMMs on the bf16 matrix pipe with 3-term split operands (csrc/wino_gemm_s3.hip) ------------
def _conv_case(rs, B, H, W, Cin, Cout, heavy_tail=False):
    x = rs.randn(B, H, W, Cin).astype(np.float32)
    if heavy_tail:
        x *= np.exp(rs.randn(B, H, W, Cin)).astype(np.float32)      # values over several binades: the split must not care
    w = (rs.randn(3, 3, Cin, Cout) * np.sqrt(2.0 / (9 * Cin))).astype(np.float32)
    b = rs.randn(Cout).astype(np.float32)
    return x, w, b


@pytest.mark.parametrize("B,H,W,Cin,Cout,pool", [
    (2, 13, 13, 64, 128, 0),     # one ragged row tile, 128-wide column tile, K = 4 stages
    (6, 13, 13, 256, 256, 0),    # mosaic g = 2, 256-wide column tile
    (3, 26, 26, 128, 512, 2),    # pooled + full resolution outputs, two column tiles
    (1, 13, 13, 1280, 256, 0),   # conv_22's K = 1280 (80 stages)
    (40, 26, 26, 32, 128, 1),    # several row tiles (Mt > 256), the shortest K the kernel takes (2 stages), pooled
    (7, 12, 18, 96, 384, 0),     # K = 6 stages, N = 3 x 128
])
@pytest.mark.parametrize("half", ["0", "1"], ids=["tile256", "tile128x2"])
def test_conv2d_split_bf16_gemm_vs_oracle(ctx, monkeypatch, B, H, W, Cin, Cout, pool, half):
    """The split-operand GEMM (DT_S3=2: wherever the shape allows) against the oracle AND against the fp32 MFMA path of
    the same layer: the two differ by no more than the fp32 path differs from the oracle."""
    monkeypatch.setenv("DT_WINO", "2")
    monkeypatch.setenv("DT_WINO_TILE", "6")
    monkeypatch.setenv("DT_S3_HALF", half)      # 1: the 128-row-tile / two-workgroups-per-CU form wherever N % 256 == 0
    x, w, b = _conv_case(np.random.RandomState(B * 7 + Cin + Cout), B, H, W, Cin, Cout)
    ref = orc.conv2d(x, w, b)
    ref = np.where(ref > 0, ref, ref * np.float32(0.1)).astype(np.float32)
    outs = {}
    for s3 in ("2", "0"):
        monkeypatch.setenv("DT_S3", s3)
        ctx.profile_reset(); ctx.profile_enable(True)
        got = ctx.conv2d(dev(x, ctx), w, b, leaky_slope=0.1, pool=pool)
        ctx.profile_enable(False)
        assert ctx.profile_read("conv_gemm_s3")["launches"] == (1 if s3 == "2" else 0)
        assert ctx.profile_read("conv_igemm")["launches"] == (0 if s3 == "2" else 1)
        if s3 == "2":
            assert ctx.profile_read("s3_tile:128x2")["launches"] == (1 if (half == "1" and Cout % 256 == 0) else 0)
        outs[s3] = [g.cpu().numpy() for g in (got if pool == 2 else (got,))]
    refs = {0: [ref], 1: [orc.maxpool2(ref)], 2: [ref, orc.maxpool2(ref)]}[pool]
    for a, f, r in zip(outs["2"], outs["0"], refs):
        e_s3, e_f32 = relerr(a, r), relerr(f, r)
        assert e_s3 < 2e-4 and e_f32 < 2e-4                 # the F(6x6) tolerance of test_conv2d_winograd_vs_oracle
        assert e_s3 < 1.25 * e_f32 + 1e-6, (e_s3, e_f32)    # and the split form is as close to the oracle as fp32 MFMA is
        assert relerr(a, f) < 5e-5                          # ... the two forms differ by GEMM rounding (amplified by the output transform) only


def test_split_bf16_gemm_error_against_float64(ctx, monkeypatch):
    """Accuracy claim of wino_gemm_s3.hip: against a float64 convolution of the same fp32 inputs the split form's error
    is not larger than the fp32 MFMA form's (both are dominated by the F(6x6) transforms), also on inputs spread over
    many binades."""
    monkeypatch.setenv("DT_WINO", "2")
    monkeypatch.setenv("DT_WINO_TILE", "6")
    rs = np.random.RandomState(5)
    x, w, b = _conv_case(rs, 4, 26, 26, 256, 256, heavy_tail=True)
    xp = np.pad(x.astype(np.float64), ((0, 0), (1, 1), (1, 1), (0, 0)))
    ref64 = np.zeros((4, 26, 26, 256)) + b.astype(np.float64)
    for dy in range(3):
        for dx in range(3):
            ref64 += np.tensordot(xp[:, dy:dy + 26, dx:dx + 26, :], w[dy, dx].astype(np.float64), axes=([3], [0]))
    err = {}
    for s3 in ("2", "0"):
        monkeypatch.setenv("DT_S3", s3)
        got = ctx.conv2d(dev(x, ctx), w, b, leaky_slope=1.0, pool=0).cpu().numpy().astype(np.float64)
        err[s3] = np.sqrt(np.mean((got - ref64) ** 2)) / np.sqrt(np.mean(ref64 ** 2))
    assert err["2"] < 1.1 * err["0"] + 1e-8, err
    assert err["2"] < 5e-5, err


@pytest.mark.parametrize("nt", [2, 3], ids=["f16x2", "bf16x3"])
@pytest.mark.parametrize("P,Mt,K,N,half,what", [
    (64, 7840, 1024, 1024, -1, "conv_19 / conv_20 at 48 clips (160 3x3 mosaics x 49 tiles), 256-row tiles"),
    (64, 7840, 1280, 1024, -1, "conv_22"),
    (64, 7840, 512, 1024, -1, "conv_14 / 16 / 18"),
    (64, 7840, 1120, 2048, -1, "convlstm_xproj (K = 1109 padded to 1120)"),
    (64, 29160, 256, 512, -1, "conv_9 / conv_11 (26x26 on 2x2 mosaics)"),
    (36, 588, 512, 2048, 1, "convlstm_step at 48 clips: F(4x4), 128-row tiles, two workgroups per CU"),
    (36, 588, 512, 2048, -1, "convlstm_step, 256-row tiles"),
    (1, 243360, 1024, 512, 2, "conv_15 / conv_17 as one split GEMM over 1440 x 169 pixels: A read as fp32 rows, split in the kernel"),
    (1, 973440, 512, 256, 2, "conv_10 / conv_12 over 1440 x 676 pixels, the same form"),
    (1, 500000, 256, 128, 2, "conv_7's shape (K = 256, 128-column tile), the same form"),
])
def test_split_bf16_gemm_benched_shapes_against_float64(ctx, P, Mt, K, N, half, what, nt):
    """wino_gemm_s3.hip AT THE SHAPES THE BENCH STEP LAUNCHES (48 clips x 30 frames x 416x416), through the production
    pack kernels and launcher (dt_gemm_split), in BOTH operand forms -- nt = 2: two fp16 terms of the scaled operands, three
    products (the default since round 6); nt = 3: three bf16 terms, six products (DT_S3_H2=0, DT_PIN) -- against float64 products of the same fp32 operands.  Operands spread
    over 13 binades (heavy-tailed sums: a few terms dominate, so the accumulator rounds at the sum's own scale).  Error relative
    to sum_k |v||u| (the scale fp32 rounding is relative to): the split form must not be above the error of the fp32 library
    GEMM (torch.bmm: fp32 MFMA) on the same data -- rms and max -- and stay at the level of fp32 rounding in absolute terms
    (rms < 1.5 x 2^-24).  (On Gaussian data tools/micro/gemm_s3_bench.hip measures 3.7e-8 rms against 4.4e-8 for an fp32 fmaf chain.)"""
    t = torch
    g = t.Generator(device=ctx.device)
    g.manual_seed(1000 + K + N)
    v = t.randn((P, Mt, K), generator=g, device=ctx.device) * t.exp2(t.randint(-6, 7, (P, Mt, K), generator=g, device=ctx.device).float())
    u = t.randn((P, N, K), generator=g, device=ctx.device) * t.exp2(t.randint(-3, 4, (P, N, K), generator=g, device=ctx.device).float()) / float(np.sqrt(K))
    ctx.profile_reset(); ctx.profile_enable(True)
    got = ctx.gemm_split(v, u, half=half, nt=nt)
    ctx.profile_enable(False)
    assert ctx.profile_read("s3_tile:128x2" if half == 1 else "s3_tile:256")["launches"] == 1
    se = sf = 0.0
    me = mf = 0.0
    n = 0
    step = max(1, int(2.5e8 // (Mt * max(K, N))))          # positions per float64 chunk (<= 2 GB per operand)
    for p0 in range(0, P, step):
        for m0 in range(0, Mt, 65536):
            vv = v[p0:p0 + step, m0:m0 + 65536]
            uu = u[p0:p0 + step]
            ref = t.bmm(vv.double(), uu.double().transpose(1, 2))
            scale = t.bmm(vv.double().abs(), uu.double().abs().transpose(1, 2))
            e = (got[p0:p0 + step, m0:m0 + 65536].double() - ref).abs() / scale
            f = (t.bmm(vv, uu.transpose(1, 2)).double() - ref).abs() / scale
            se += float((e * e).sum()); sf += float((f * f).sum()); n += e.numel()
            me = max(me, float(e.max())); mf = max(mf, float(f.max()))
    rms_e, rms_f = float(np.sqrt(se / n)), float(np.sqrt(sf / n))
    form = "f16x2" if nt == 2 else "bf16x3"
    print("gemm_s3 %s P=%d Mt=%d K=%d N=%d half=%d: rms %.3g max %.3g | fp32 library GEMM rms %.3g max %.3g  (%s)" % (form, P, Mt, K, N, half, rms_e, me, rms_f, mf, what))
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, "parity_r06_gemm_split_f64.txt"), "a") as fh:
        fh.write("%-6s P=%d Mt=%d K=%d N=%d half=%d  split: rms %.4g max %.4g   fp32 library GEMM: rms %.4g max %.4g   # %s\n" % (form, P, Mt, K, N, half, rms_e, me, rms_f, mf, what))
    # measured (profiles/parity_r04_gemm_s3_f64.txt): split 6.5-6.8e-8 rms / 0.94-1.1e-6 max, fp32 library GEMM 8.2-8.8e-8 / 1.3-1.7e-6
    assert rms_e <= 1.02 * rms_f and me <= 1.05 * mf, (rms_e, rms_f, me, mf)      # not above an fp32 GEMM's error on the same data
    assert rms_e < 1.5 * 2.0 ** -24 and me < 32 * 2.0 ** -24, (rms_e, me)        # and at the level of fp32 rounding in absolute terms


@pytest.mark.parametrize("h2", ["1", "0"], ids=["f16x2", "bf16x3"])
def test_split_bf16_gemm_one_hot_taps(ctx, monkeypatch, h2):
    """one-hot taps on small integers: every K block / term plane / row tile / column tile of the split layout must line up."""
    monkeypatch.setenv("DT_S3_H2", h2)
    monkeypatch.setenv("DT_WINO", "2")
    monkeypatch.setenv("DT_WINO_TILE", "6")
    monkeypatch.setenv("DT_S3", "2")
    B, H, W, Cin, Cout = 9, 13, 13, 96, 384
    x = (np.arange(B * H * W * Cin, dtype=np.float32).reshape(B, H, W, Cin) % 251)
    w = np.zeros((3, 3, Cin, Cout), dtype=np.float32)
    for n in range(Cout):
        w[n % 3, (n // 3) % 3, (n * 7) % Cin, n] = 1.0
    ctx.profile_reset(); ctx.profile_enable(True)
    got = ctx.conv2d(dev(x, ctx), w, None, leaky_slope=1.0, pool=0).cpu().numpy()
    ctx.profile_enable(False)
    assert ctx.profile_read("conv_gemm_s3")["launches"] == 1
    assert ctx.profile_read("s3_form:f16x2" if h2 == "1" else "s3_form:bf16x3")["launches"] == 1
    assert np.abs(got - orc.conv2d(x, w)).max() < 0.05


@pytest.mark.parametrize("k", [3, 1], ids=["winograd", "1x1"])
def test_h2_scale_follows_the_data(ctx, monkeypatch, k):
    """The fp16 form scales its operands by a power of two from the MEASURED max |x| of the input tensor: the same layer on
    x * 2^e (e = -70 ... +70: far outside fp16's exponent range) must give exactly 2^e times the same bits (bias-free layer,
    LeakyReLU is positively homogeneous), an all-zero input exactly the bias, and one huge element must not disturb the rest."""
    monkeypatch.setenv("DT_S3_H2", "1")
    monkeypatch.setenv("DT_WINO", "2")
    monkeypatch.setenv("DT_WINO_TILE", "6")
    monkeypatch.setenv("DT_S3", "2")
    rs = np.random.RandomState(5)
    B, H, W, Cin, Cout = (20, 13, 13, 128, 256) if k == 3 else (64, 13, 13, 256, 128)
    x = rs.randn(B, H, W, Cin).astype(np.float32)
    w = (rs.randn(k, k, Cin, Cout) / np.sqrt(k * k * Cin)).astype(np.float32)
    b = rs.randn(Cout).astype(np.float32)
    ctx.profile_reset(); ctx.profile_enable(True)
    base = ctx.conv2d(dev(x, ctx), w, None, leaky_slope=0.1, pool=0).cpu().numpy()
    ctx.profile_enable(False)
    assert ctx.profile_read("s3_form:f16x2")["launches"] == 1 and ctx.profile_read("absmax")["launches"] == 1
    ref = orc.conv2d(x[:2], w)
    ref = np.where(ref > 0, ref, ref * np.float32(0.1))
    assert relerr(base[:2], ref) < 1e-4
    for e in (-70, -20, 9, 70):
        got = ctx.conv2d(dev(np.ldexp(x, e), ctx), w, None, leaky_slope=0.1, pool=0).cpu().numpy()
        assert np.array_equal(got, np.ldexp(base, e)), e
    zero = ctx.conv2d(dev(np.zeros_like(x), ctx), w, b, leaky_slope=1.0, pool=0).cpu().numpy()
    assert np.array_equal(zero, np.broadcast_to(b, zero.shape))
    # One outlier: the tensor's power of two follows it, elements more than ~2^11 below it get a subnormal lo term (absolute precision
    # 2^-31 of the maximum instead of relative 2^-24): full precision with an outlier of 2^10, one bit lost per binade beyond --
    # at 2^20 the frame WITHOUT the spike is still good to 1e-3 (measured 2e-4), and the parity bars are 3e-4 / 1e-3.
    for e, bar in ((10, 3e-5), (20, 1e-3)):
        xs = x.copy(); xs[0, 6, 6, 3] = 2.0 ** e
        got = ctx.conv2d(dev(xs, ctx), w, None, leaky_slope=1.0, pool=0).cpu().numpy()
        ref = orc.conv2d(xs[:2], w)
        assert np.abs(got[1] - ref[1]).max() < bar * np.abs(ref[1]).max(), (e, np.abs(got[1] - ref[1]).max() / np.abs(ref[1]).max())
        assert np.abs(got[0] - ref[0]).max() < 1e-3 * np.abs(ref[0]).max()      # (F(6x6) itself cancels spike-sized numbers in that tile)


def test_split_bf16_default_policy_engages_on_deep_layers(ctx, monkeypatch):
    """Default policy (DT_S3=1): K >= 128 and enough GEMM rows take the split form -- 128 rows where the launch takes the fp16 form, 2048 in the bf16 form
    (profiles/r06_experiments.txt section 10) --, few rows stay on fp32 MFMA."""
    monkeypatch.delenv("DT_S3", raising=False)
    monkeypatch.delenv("DT_S3_MINROWS", raising=False)
    rs = np.random.RandomState(11)
    for (B, H, W, Cin, Cout), h2, want in (((384, 13, 13, 256, 256), "1", 1), ((8, 13, 13, 256, 256), "1", 0), ((96, 26, 26, 128, 256), "1", 1),      # 2107 / 49 / 1944 rows
                                           ((384, 13, 13, 256, 256), "0", 1), ((96, 26, 26, 128, 256), "0", 0)):
        monkeypatch.setenv("DT_S3_H2", h2)
        x, w, b = _conv_case(rs, B, H, W, Cin, Cout)
        ctx.profile_reset(); ctx.profile_enable(True)
        got = ctx.conv2d(dev(x, ctx), w, b, leaky_slope=0.1, pool=0)
        ctx.profile_enable(False)
        assert ctx.profile_read("conv_gemm_s3")["launches"] == want, (B, H, W, Cin, Cout)
        ref = orc.conv2d(x[:2], w, b)
        ref = np.where(ref > 0, ref, ref * np.float32(0.1)).astype(np.float32)
        assert relerr(got[:2].cpu().numpy(), ref) < 2e-4


def test_split_bf16_handover_to_1x1_layers(ctx, monkeypatch):
    """conv_7, 10, 12, 15, 17 (and conv_23: ragged N = 85, unaligned output rows) as split GEMMs straight on the producing layer's fp32 activation (DT_S3_1X1=1, the default: the kernel
    splits its A fragments itself, bias as an extra K stage, LeakyReLU in its epilogue) against the same network with those layers
    on the fp32 MFMA kernel, and against the oracle on the frames the oracle is run on."""
    B, H, W, C = 16, 416, 416, 12
    frames = np.random.RandomState(3).randint(0, 256, (B, H, W, 3)).astype(np.uint8)
    outs = {}
    layers = None
    for mode in ("1", "b", "0"):      # fp16 form (default) / bf16 form / fp32 MFMA
        monkeypatch.setenv("DT_S3_H2", "0" if mode == "b" else "1")
        monkeypatch.setenv("DT_H2_MINFRAMES", "0")     # (the default policy takes the fp16 form from 12 frames per forward)
        monkeypatch.setenv("DT_S3_MINROWS", "2048")    # (the Winograd-form layers of this 16-frame forward stay on the fp32 MFMA kernel in every mode: only the 1x1 layers differ)
        monkeypatch.setenv("DT_S3_1X1", "0" if mode == "0" else "1")          # read when the context is created (dt_create)
        monkeypatch.setenv("DT_S3_1X1_MINK", "256")    # (the default)
        monkeypatch.setenv("DT_S3_1X1_MINROWS", "0")   # (the default policy takes these layers from 16384 pixels per launch)
        det, layers, _ = _detector(ctx, H, W, C, seed=77)
        c = det.model.ctx
        c.profile_reset(); c.profile_enable(True)
        outs[mode] = c.detect_forward(dev(frames, c)).cpu().numpy()
        c.profile_enable(False)
        hand = sorted(int(n.split("_")[-1]) for n in c.profile_names() if n.startswith("conv_gemm_s3:conv_") and c.profile_read(n)["launches"]
                      and int(n.split("_")[-1]) in (4, 7, 10, 12, 15, 17, 21, 23))
        # (conv_23 joins since round 5: the row-form epilogue stores 4 bytes per lane, so the netout's 85-float rows need no alignment)
        assert hand == ([7, 10, 12, 15, 17, 23] if mode != "0" else []), hand
        assert c.profile_read("wino_output:conv_14")["launches"] == 1
        assert (c.profile_read("s3_form:f16x2")["launches"] > 0) == (mode == "1") and (c.profile_read("s3_form:bf16x3")["launches"] > 0) == (mode == "b")
    assert chan_err(flat_c(outs["1"]), flat_c(outs["0"])) < 1e-4          # two roundings of the same network (measured 5e-5)
    assert chan_err(flat_c(outs["b"]), flat_c(outs["0"])) < 1e-4
    ref_net, _, _ = orc.yolov2_forward(orc.normalize_u8(frames[:2]), layers, taps=())
    assert chan_err(flat_c(outs["1"][:2]), flat_c(ref_net)) < NET_TOL


@pytest.mark.parametrize("tile", ["", "4"], ids=["projection_F6x6", "recurrent_F4x4"])
@pytest.mark.parametrize("B,H,W,Cx,U", [(6, 13, 13, 64, 32), (50, 13, 13, 96, 64), (3, 19, 19, 128, 128)])
def test_convlstm_step_split_bf16_gemm_vs_oracle(ctx, monkeypatch, B, H, W, Cx, U, tile):
    """ConvLSTM2D step with one of its convolutions on the split-operand GEMM (DT_S3=2): with the default tile the input
    projection as F(6x6); with DT_WINO_TILE=4 the recurrent convolution as F(4x4) with the gate update in its output transform
    (wino_input_kernel<4,4,S3>, P = 36) -- the form the tracker's recurrence takes.  Against the oracle and against the fp32 MFMA
    form of the same step."""
    monkeypatch.setenv("DT_WINO", "2")
    monkeypatch.setenv("DT_S3_HALF", "1" if tile else "0")      # the recurrent form also through the 128-row tiles
    if tile:
        monkeypatch.setenv("DT_WINO_TILE", tile)
    rs = np.random.RandomState(B + U)
    x = rs.randn(B, H, W, Cx).astype(np.float32)
    h = (rs.randn(B, H, W, U) * .5).astype(np.float32); c = rs.randn(B, H, W, U).astype(np.float32)
    Wk = (rs.randn(3, 3, Cx, 4 * U) * .05).astype(np.float32); Uk = (rs.randn(3, 3, U, 4 * U) * .05).astype(np.float32)
    b = rs.randn(4 * U).astype(np.float32) * .1
    rh, rc = orc.convlstm_step(x, h, c, Wk, Uk, b)
    out = {}
    for s3 in ("2", "0"):
        monkeypatch.setenv("DT_S3", s3)
        ctx.profile_reset(); ctx.profile_enable(True)
        gh, gc = ctx.convlstm_step(dev(x, ctx), dev(h, ctx), dev(c, ctx), Wk, Uk, b)
        ctx.profile_enable(False)
        tag = "conv_gemm_s3:convlstm_step" if tile else "conv_gemm_s3:convlstm_xproj"
        assert ctx.profile_read(tag)["launches"] == (1 if s3 == "2" else 0)
        assert ctx.profile_read("conv_gemm_s3")["launches"] == (1 if s3 == "2" else 0)
        out[s3] = (gh.cpu().numpy(), gc.cpu().numpy())
        np.testing.assert_allclose(out[s3][0], rh, rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(out[s3][1], rc, rtol=1e-4, atol=1e-4)
    assert np.abs(out["2"][0] - out["0"][0]).max() < 1e-4 and np.abs(out["2"][1] - out["0"][1]).max() < 1e-4      # two roundings of one GEMM through F(6x6): measured 3e-5


def test_tracker_recurrence_on_split_bf16_gemm(ctx, monkeypatch):
    """A whole tracker forward (detector + ConvLSTM over T + 1x1 head) with every eligible GEMM forced onto the split kernel
    (DT_S3=2, thresholds off) against the oracle chain -- the path the bench takes at 48 clips, at a size the oracle finishes."""
    monkeypatch.setenv("DT_S3", "2")
    monkeypatch.setenv("DT_S3_HALF", "1")
    test_track_forward_vs_oracle_small(ctx)
    test_track_clips_boxes_and_ids_vs_oracle(ctx)

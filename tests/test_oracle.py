"""CPU suite, part 1: the oracle itself.
 - decode/NMS/bbox_iou/WeightReader/normalize restatements against the golden
   vectors produced by executing the reference's numpy code (tools/make_goldens.py);
 - conv / pool / s2d / ConvLSTM / LSTM restatements against torch-CPU, an
   independent implementation (parity vs Keras itself is unpinned, see oracle.c).
"""
import glob
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import oracle as orc


def _golden_cases(golden_dir):
    return sorted(glob.glob(os.path.join(golden_dir, "decode_*.npz")))


def test_goldens_present(golden_dir):
    assert len(_golden_cases(golden_dir)) >= 11


@pytest.mark.parametrize("name", [
    "g13_c80_n24", "g13_c12_n32", "g19_c12_n128", "g3_background", "g5_rescale", "g3_relabel",
    "g7_c5_lowthr", "g7_c12_nms09", "g7_c12_nms01", "g5_c1", "g4_dense"])
def test_oracle_decode_matches_reference_golden(golden_dir, name):
    d = np.load(os.path.join(golden_dir, "decode_%s.npz" % name))
    rows, post = orc.decode_netout(d["netout"], float(d["obj_threshold"]), float(d["nms_threshold"]),
                                   d["anchors"], int(d["nb_class"]))
    g = d["boxes"]
    assert len(rows) == len(g), "box count differs from the reference"
    if len(g):
        assert np.array_equal(rows[:, 5], g[:, 5]), "labels / order differ"
        # box coords, conf, score: float32 arithmetic, only exp/summation-order ulps differ
        np.testing.assert_allclose(rows[:, :5], g[:, :5], rtol=2e-6, atol=1e-6)
        np.testing.assert_allclose(rows[:, 6], g[:, 6], rtol=2e-6, atol=1e-6)
    if "netout_post" in d:
        np.testing.assert_allclose(post, d["netout_post"], rtol=2e-6, atol=1e-7)


def test_oracle_bbox_iou_bit_exact(golden_dir):
    d = np.load(os.path.join(golden_dir, "bbox_iou.npz"))
    got = np.array([orc.bbox_iou(p[:4], p[4:]) for p in d["pairs"]], dtype=np.float32)
    assert np.array_equal(got, d["iou"].astype(np.float32))


def test_oracle_normalize_golden(golden_dir):
    d = np.load(os.path.join(golden_dir, "normalize.npz"))
    assert np.array_equal(orc.normalize_u8(d["img"]), d["out"].astype(np.float32))


def test_oracle_conv_vs_torch():
    rs = np.random.RandomState(0)
    x = rs.randn(2, 10, 12, 16).astype(np.float32)
    w = (rs.randn(3, 3, 16, 24) * 0.1).astype(np.float32)
    b = rs.randn(24).astype(np.float32)
    y = orc.conv2d(x, w, b)
    yt = F.conv2d(torch.tensor(x).permute(0, 3, 1, 2), torch.tensor(w).permute(3, 2, 0, 1), torch.tensor(b),
                  padding=1).permute(0, 2, 3, 1).numpy()
    np.testing.assert_allclose(y, yt, rtol=1e-4, atol=1e-5)
    w1 = rs.randn(1, 1, 16, 8).astype(np.float32)
    y1 = orc.conv2d(x, w1)
    y1t = F.conv2d(torch.tensor(x).permute(0, 3, 1, 2), torch.tensor(w1).permute(3, 2, 0, 1)).permute(0, 2, 3, 1).numpy()
    np.testing.assert_allclose(y1, y1t, rtol=1e-4, atol=1e-5)


def test_oracle_bn_leaky_pool_s2d_vs_torch():
    rs = np.random.RandomState(1)
    x = rs.randn(2, 8, 6, 12).astype(np.float32)
    g, b, m, v = rs.rand(12) + .5, rs.randn(12), rs.randn(12), rs.rand(12) + .5
    y = orc.bn_leaky(x, g, b, m, v)
    xt = torch.tensor(x).permute(0, 3, 1, 2)
    yt = F.leaky_relu(F.batch_norm(xt, torch.tensor(m, dtype=torch.float32), torch.tensor(v, dtype=torch.float32),
                                   torch.tensor(g, dtype=torch.float32), torch.tensor(b, dtype=torch.float32),
                                   training=False, eps=1e-3), 0.1).permute(0, 2, 3, 1).numpy()
    np.testing.assert_allclose(y, yt, rtol=1e-5, atol=1e-5)
    np.testing.assert_array_equal(orc.maxpool2(x), F.max_pool2d(xt, 2).permute(0, 2, 3, 1).numpy())
    # tf.space_to_depth(2) NHWC: out[..., (dy*2+dx)*C + c] = in[2h+dy, 2w+dx, c]
    ref = x.reshape(2, 4, 2, 3, 2, 12).transpose(0, 1, 3, 2, 4, 5).reshape(2, 4, 3, 48)
    np.testing.assert_array_equal(orc.space_to_depth2(x), ref)


def _hs(v):
    return torch.clamp(0.2 * v + 0.5, 0, 1)


def test_oracle_lstm_vs_torch():
    rs = np.random.RandomState(2)
    B, D, U = 3, 20, 8
    x, h, c = (rs.randn(B, D).astype(np.float32), rs.randn(B, U).astype(np.float32), rs.randn(B, U).astype(np.float32))
    W, Ur, b = (rs.randn(D, 4 * U) * .3).astype(np.float32), (rs.randn(U, 4 * U) * .3).astype(np.float32), rs.randn(4 * U).astype(np.float32)
    ho, co = orc.lstm_step(x, h, c, W, Ur, b)
    z = torch.tensor(x) @ torch.tensor(W) + torch.tensor(h) @ torch.tensor(Ur) + torch.tensor(b)
    i, f, g, o = z[:, :U], z[:, U:2 * U], z[:, 2 * U:3 * U], z[:, 3 * U:]
    cn = _hs(f) * torch.tensor(c) + _hs(i) * torch.tanh(g)
    hn = _hs(o) * torch.tanh(cn)
    np.testing.assert_allclose(ho, hn.numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(co, cn.numpy(), rtol=1e-5, atol=1e-6)


def test_oracle_convlstm_vs_torch():
    rs = np.random.RandomState(3)
    B, H, W, Cx, U = 2, 5, 4, 6, 8
    x = rs.randn(B, H, W, Cx).astype(np.float32)
    h = rs.randn(B, H, W, U).astype(np.float32); c = rs.randn(B, H, W, U).astype(np.float32)
    Wk = (rs.randn(3, 3, Cx, 4 * U) * .2).astype(np.float32); Uk = (rs.randn(3, 3, U, 4 * U) * .2).astype(np.float32)
    b = rs.randn(4 * U).astype(np.float32)
    ho, co = orc.convlstm_step(x, h, c, Wk, Uk, b)
    cv = lambda a, k, bias=None: F.conv2d(torch.tensor(a).permute(0, 3, 1, 2), torch.tensor(k).permute(3, 2, 0, 1),
                                          bias, padding=1).permute(0, 2, 3, 1)
    z = cv(x, Wk, torch.tensor(b)) + cv(h, Uk)
    i, f, g, o = z[..., :U], z[..., U:2 * U], z[..., 2 * U:3 * U], z[..., 3 * U:]
    cn = _hs(f) * torch.tensor(c) + _hs(i) * torch.tanh(g)
    hn = _hs(o) * torch.tanh(cn)
    np.testing.assert_allclose(ho, hn.numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(co, cn.numpy(), rtol=1e-4, atol=1e-5)


def test_oracle_pool_and_dense():
    rs = np.random.RandomState(4)
    x = rs.randn(3, 8, 8, 12).astype(np.float32)
    np.testing.assert_array_equal(orc.global_maxpool(x), x.max(axis=(1, 2)))
    mp = F.max_pool2d(torch.tensor(x).permute(0, 3, 1, 2), 4).permute(0, 2, 3, 1).reshape(3, -1).numpy()
    np.testing.assert_array_equal(orc.maxpool4_flatten(x), mp)
    h = rs.randn(5, 16).astype(np.float32); Wd = rs.randn(16, 4).astype(np.float32); bd = rs.randn(4).astype(np.float32)
    np.testing.assert_allclose(orc.dense_sigmoid(h, Wd, bd), torch.sigmoid(torch.tensor(h) @ torch.tensor(Wd) + torch.tensor(bd)).numpy(),
                               rtol=1e-5, atol=1e-6)


def test_oracle_associate_spec():
    """Hand-checked case of the build-defined association (DESIGN.md "Track identity")."""
    cap, T = 4, 3
    boxes = np.zeros((T, cap, 8), dtype=np.float32)
    def put(t, i, x, y, w, h, lab):
        boxes[t, i] = [x, y, w, h, 0.9, lab, 0.8, 0]
    put(0, 0, .2, .2, .1, .1, 0); put(0, 1, .7, .7, .1, .1, 1)
    put(1, 0, .71, .7, .1, .1, 1); put(1, 1, .21, .2, .1, .1, 0); put(1, 2, .5, .5, .1, .1, 0)
    put(2, 0, .22, .2, .1, .1, 1)   # overlaps track 0 but label differs -> new id
    put(2, 1, .5, .51, .1, .1, 0)
    ids, n = orc.associate_clip(boxes, np.array([2, 3, 2]), 0.3)
    assert ids[0, :2].tolist() == [0, 1]
    assert ids[1, :3].tolist() == [1, 0, 2]
    assert ids[2, :2].tolist() == [3, 2]
    assert n == 4 and ids[0, 2] == -1


def test_oracle_heatmap_helpers_match_reference_golden(golden_dir):
    """utility/utils.py:53-79 incl. numpy's negative-slice semantics for boxes off the top/left"""
    d = np.load(os.path.join(golden_dir, "heatmap.npz"))
    assert np.array_equal(orc.heatmap_from_boxes(d["box4"], 32), d["heat"])
    assert np.array_equal(orc.rect_from_heatmap(d["soft"].reshape(64, -1), 32, 0.75), d["rects"])


TARGET_CASES = ["g13_c12", "g13_c12_aug", "g19_c20_wrap", "g13_dense_cell"]


def load_target_case(golden_dir, name):
    z = np.load(os.path.join(golden_dir, "targets.npz"))
    G, IM, C, TBB = [int(v) for v in z[name + "/cfg"]]
    aug = z[name + "/aug"] if (name + "/aug") in z.files else None
    return dict(objs=z[name + "/objs"], counts=z[name + "/counts"], dims=z[name + "/dims"], aug=aug, G=G, IM=IM,
                C=C, TBB=TBB, anchors=z["anchors"], y=z[name + "/y"], b=z[name + "/b"])


@pytest.mark.parametrize("name", TARGET_CASES)
def test_oracle_encode_targets_matches_reference_golden(golden_dir, name):
    """preprocessing.py:171-188,214-293 (exec'd by tools/make_goldens.py): float64, bit-exact."""
    c = load_target_case(golden_dir, name)
    y, b = orc.encode_targets(c["objs"], c["counts"], c["dims"], c["aug"], c["G"], c["G"], 5, c["C"], c["IM"],
                              c["IM"], c["TBB"], c["anchors"])
    assert np.array_equal(y, c["y"])
    assert np.array_equal(b, c["b"])
    assert (c["y"][..., 4] == 1).sum() > 0


def test_oracle_sequence_windows_match_reference_golden(golden_dir):
    """preprocessing.py:79-89, including its duplicated windows and IndexError at folder boundaries."""
    z = np.load(os.path.join(golden_dir, "windows.npz"))
    seen_err = False
    for i in range(int(z["n"])):
        folders, T = z["folders_%d" % i], int(z["T_%d" % i])
        if int(z["err_%d" % i]):
            seen_err = True
            with pytest.raises(IndexError):
                orc.sequence_windows(folders, T)
        else:
            assert orc.sequence_windows(folders, T) == z["starts_%d" % i].tolist()
    assert seen_err


def test_oracle_tracker_vs_torch_cpu_graph_mid_size():
    """the C port against the torch-CPU (oneDNN) statement of the same graph at 128x160, T=3 -- both float32, two
    independent convolution implementations (bench.py times the faster one as the CPU baseline)"""
    from oracle import torch_cpu
    from utility import synth
    C = 12
    layers, _ = orc.parse_darknet_blob(synth.synth_darknet_blob(C), C)
    tw = synth.synth_tracker_weights(C)
    x = orc.normalize_u8(synth.synth_clip(3, 128, 160, 2, seed=3))
    a_trk, a_det = orc.tracker_forward(x, layers, tw)
    b_trk, b_det = torch_cpu.tracker_forward(x, layers, tw)
    for a, b in ((a_trk, b_trk), (a_det, b_det)):
        assert a.shape == b.shape
        assert (np.abs(a - b) / np.maximum(1.0, np.abs(a))).max() < 2e-4

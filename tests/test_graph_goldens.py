"""CPU suite, part 3: the oracle's GRAPH against fixtures produced by executing the reference's own
graph-building code (tools/make_graph_goldens.py: KerasYOLO.load_model incl. init_weights + WeightReader,
MultiObjDetTracker.load_model, TinyTracker.load_tracker_model, run under a torch-float64 stand-in for
the Keras names).  What this pins: layer order and names, skip tap, space_to_depth channel order, both
concat orders (skip first; x_bbox first), the darknet read order and OIHW->HWIO transposes, gate order
i,f,c,o, BN epsilon, LeakyReLU slope -- by code that shares no line with oracle/.  What it does not pin:
Keras' own arithmetic (cannot run here).

Bar: |oracle_f32 - truth_f64| <= 2e-4 * max(1, |truth|) per element -- a wrong topology decision is an
O(1) error, float32 rounding through 23 layers is ~1e-5."""
import os

import numpy as np
import pytest

from oracle import oracle as orc
from utility import synth


def _close(got, ref, tol=2e-4):
    err = np.abs(got.astype(np.float64) - ref.astype(np.float64)) / np.maximum(1.0, np.abs(ref))
    return float(err.max())


def test_oracle_yolov2_small_vs_reference_graph(golden_dir):
    d = np.load(os.path.join(golden_dir, "graph_yolov2_64x96_c12.npz"))
    C = int(d["nb_class"])
    layers, used = orc.parse_darknet_blob(synth.synth_darknet_blob(C, seed=int(d["seed_blob"])), C)
    net, feat, _ = orc.yolov2_forward(orc.normalize_u8(d["frames"]), layers)
    assert net.shape == d["netout"].shape
    assert _close(net, d["netout"]) < 2e-4
    assert _close(feat, d["conv_feat"]) < 2e-4


def test_oracle_yolov2_full_size_vs_reference_graph(golden_dir):
    """one 416x416 frame, C=80, every named tap the fixture holds"""
    d = np.load(os.path.join(golden_dir, "graph_yolov2_416_c80.npz"))
    C = int(d["nb_class"])
    blob = synth.synth_darknet_blob(C, seed=int(d["seed_blob"]))
    layers, used = orc.parse_darknet_blob(blob, C)
    assert used == blob.size
    frame = synth.synth_clip(1, 416, 416, 3, seed=int(d["seed_frame"]))
    x = orc.normalize_u8(frame)
    net, feat, taps = orc.yolov2_forward(x, layers, taps=("act_13",))
    assert _close(net, d["netout"]) < 2e-4
    assert _close(feat, d["conv_feat"]) < 2e-4
    assert _close(net.reshape(d["conv_23"].shape), d["conv_23"]) < 2e-4
    # 'norm_13' is the BatchNormalization output BEFORE LeakyReLU: leaky is monotone, so apply it to the fixture
    n13 = d["norm_13_stride4"]
    a13 = np.where(n13 > 0, n13, 0.1 * n13)
    assert _close(taps["act_13"][:, ::4, ::4], a13) < 2e-4
    # conv_21's raw convolution output (no BN) on the skip tensor
    c21 = orc.conv2d(taps["act_13"], layers[21]["kernel"])
    assert _close(c21, d["conv_21"]) < 2e-4


@pytest.mark.parametrize("tag", ["64_T4", "416_T3"])
def test_oracle_tracker_vs_reference_graph(golden_dir, tag):
    d = np.load(os.path.join(golden_dir, "graph_tracker_%s.npz" % tag))
    C, H, W, T = int(d["nb_class"]), int(d["H"]), int(d["W"]), int(d["T"])
    layers, _ = orc.parse_darknet_blob(synth.synth_darknet_blob(C, seed=int(d["seed_blob"])), C)
    tw = synth.synth_tracker_weights(C, seed=int(d["seed_tracker"]))
    clip = synth.synth_clip(T, H, W, 2, seed=int(d["seed_clip"]))
    trk, det = orc.tracker_forward(orc.normalize_u8(clip), layers, tw)
    assert _close(det, d["detection"]) < 2e-4
    assert _close(trk, d["tracking"]) < 2e-4


@pytest.mark.parametrize("pool", ["global", "max"])
def test_oracle_tinytracker_vs_reference_graph(golden_dir, pool):
    d = np.load(os.path.join(golden_dir, "graph_tiny_%s.npz" % pool))
    n_seq, T, w, h, c = [int(v) for v in d["shape"]]
    rs = np.random.RandomState(int(d["seed"]))
    feat = rs.randn(n_seq, T, w, h, c).astype(np.float32)
    det4 = rs.rand(n_seq, T, 4).astype(np.float32)
    tw = synth.synth_tiny_weights(int(d["feat_dim"]))
    out = orc.tinytracker_forward(feat, det4, tw, pool=str(d["pool"]))
    assert _close(out, d["out"], 1e-5) < 1e-5

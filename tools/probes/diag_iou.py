#!/usr/bin/env python3
"""diagnostic: the boosted-decode box comparison of tests/test_gpu_parity.py::test_detector_full_size_one_frame_vs_oracle under
DT_WINO=2 / DT_WINO_TILE=6, for conv_1 on the split-bf16 kernel (DT_S3_CONV1=1) and on the fp32 MFMA kernel (0)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["DT_WINO"] = "2"; os.environ["DT_WINO_TILE"] = "6"
import numpy as np, torch
import object_tracking_amd  # noqa
from oracle import oracle as orc
from utility import synth
import test_gpu_parity as T
import mi355_dt
ctx = mi355_dt.Context()
frame = synth.synth_clip(1, 416, 416, 3, seed=7)
for mode in ("1", "0"):
    os.environ["DT_S3_CONV1"] = mode
    det, layers, _ = T._detector(ctx, 416, 416, 80)
    ref_net, _, _ = orc.yolov2_forward(orc.normalize_u8(frame), layers)
    net = det.model.ctx.detect_forward(T.dev(frame, det.model.ctx)).cpu().numpy()
    boost = net.copy(); boost[..., 4] += 2.0; boost[..., 5:] *= 4.0
    rboost = ref_net.copy(); rboost[..., 4] += 2.0; rboost[..., 5:] *= 4.0
    thr = T.gap_threshold(T.oracle_scores(rboost[0], 80).ravel(), 0.3, 0.25, 0.35)
    rows, _ = orc.decode_netout(rboost[0], thr, 0.45, T.ANCHORS, 80)
    r = det.model.ctx.decode(T.dev(boost, det.model.ctx), thr, 0.45, T.ANCHORS, 80)
    n = int(r["counts"][0]); got = r["boxes"][0, :n].cpu().numpy()
    ious = T.iou_rows(got[:, :4], rows[:, :4]) if n == len(rows) else np.zeros(1)
    k = int(np.argmin(ious))
    print("DT_S3_CONV1=%s: chan_err %.3g  n %d/%d  box_err %.3g  min IoU %.6f at box %d: got %s ref %s" % (
        mode, T.chan_err(T.flat_c(net), T.flat_c(ref_net)), n, len(rows), T.box_err(got, rows) if n == len(rows) else -1, ious.min(), k,
        got[k, :4] if n == len(rows) else None, rows[k, :4] if n == len(rows) else None))
    cell = int(rows[k, 7]) if n == len(rows) else 0
    g = net[0].reshape(-1, 85)[cell]; rr = ref_net[0].reshape(-1, 85)[cell]
    print("   cell %d raw t (x,y,w,h,o): got %s ref %s" % (cell, g[:5], rr[:5]))

#!/usr/bin/env python3
"""End-to-end error of the detector output against the float32 CPU oracle, per Winograd policy (4 frames 416x416)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import object_tracking_amd  # noqa: F401
from oracle import oracle as orc
from utility import synth
from models_detection.KerasYOLO import KerasYOLO

C = 12
blob = synth.synth_darknet_blob(C)
layers, _ = orc.parse_darknet_blob(blob, C)
frames = np.concatenate([synth.synth_clip(2, 416, 416, 3, seed=s) for s in (11, 12)])
ref_net, ref_feat, _ = orc.yolov2_forward(orc.normalize_u8(frames), layers)
rel = lambda a, b: float(np.abs(a - b).max() / np.abs(b).max())
for name, env in (("direct form everywhere", {"DT_WINO": "0"}), ("F(2x2,3x3) default policy", {"DT_WINO_TILE": "2"}),
                  ("F(4x4,3x3) default policy", {}), ("F(4x4,3x3) every 3x3 layer", {"DT_WINO": "2"})):
    for k in ("DT_WINO", "DT_WINO_TILE"):
        os.environ.pop(k, None)
    os.environ.update(env)
    det = KerasYOLO({'LABELS': [str(i) for i in range(C)], 'BATCH_SIZE': 4, 'IMAGE_H': 416, 'IMAGE_W': 416, 'GRID_H': 13,
                     'GRID_W': 13}, weights=blob)
    c = det.model.ctx
    net, feat = c.detect_forward(torch.from_numpy(frames).to(c.device), want_feat=True)
    print("%-28s netout rel. err %.2e   conv_feat rel. err %.2e" % (name, rel(net.cpu().numpy(), ref_net), rel(feat.cpu().numpy(), ref_feat)))

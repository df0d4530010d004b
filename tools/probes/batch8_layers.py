#!/usr/bin/env python3
"""Per-launch HIP-event times of BASELINE.json configs[1] (YOLOv2 C=80, batch 8, 416x416): where a 1.8 ms batch goes.
   python tools/batch8_layers.py [batch]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from bench import KerasYOLO, synth, make_frames

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
H = W = 416
blob = synth.synth_darknet_blob(80, seed=1234)
det = KerasYOLO({'LABELS': KerasYOLO.LABELS_COCO, 'BATCH_SIZE': B, 'IMAGE_H': H, 'IMAGE_W': W, 'GRID_H': 13, 'GRID_W': 13}, weights=blob)
ctx = det.model.ctx
frames = make_frames(1, B, H, W, ctx.device, seed0=7)[0].contiguous()
for _ in range(5):
    det.detect(frames)
torch.cuda.synchronize()
for graphs in (False, True):
    ctx.graph_enable(graphs)
    for _ in range(5):
        det.detect(frames)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(100):
        det.detect(frames)
    torch.cuda.synchronize()
    print("graphs=%d  %.3f ms per batch of %d" % (graphs, 10 * (time.perf_counter() - t0), B))
ctx.graph_enable(False)
ctx.profile_reset(); ctx.profile_enable(True)
N = 20
for _ in range(N):
    det.detect(frames)
torch.cuda.synchronize()
ctx.profile_enable(False)
tot = 0.0
rows = []
for name in ctx.profile_names():
    r = ctx.profile_read(name)
    rows.append((name, r["launches"] / N, r["ms"] / N))
for name, l, ms in rows:
    if ":" not in name:
        tot += ms
    print("%-32s %6.1f launches  %8.4f ms" % (name, l, ms))
print("sum of kernel times %.3f ms" % tot)

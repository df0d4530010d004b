#!/usr/bin/env python3
"""BASELINE configs[1] forward + decode, N plain iterations and nothing else: run under
   rocprofv3 --kernel-trace --stats to compare the kernels' own durations with the wall time per batch (launch gaps).
   python tools/batch8_trace.py [batch] [iters] [graphs 0|1]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from bench import KerasYOLO, synth, make_frames

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
N = int(sys.argv[2]) if len(sys.argv) > 2 else 100
G = int(sys.argv[3]) if len(sys.argv) > 3 else 0
blob = synth.synth_darknet_blob(80, seed=1234)
det = KerasYOLO({'LABELS': KerasYOLO.LABELS_COCO, 'BATCH_SIZE': B, 'IMAGE_H': 416, 'IMAGE_W': 416, 'GRID_H': 13, 'GRID_W': 13}, weights=blob)
frames = make_frames(1, B, 416, 416, det.model.ctx.device, seed0=7)[0].contiguous()
det.model.ctx.graph_enable(bool(G))
for _ in range(5):
    det.detect(frames)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(N):
    det.detect(frames)
torch.cuda.synchronize()
print("WALL %.4f ms per batch of %d over %d iterations (+5 warm-up), graphs=%d" % (1e3 * (time.perf_counter() - t0) / N, B, N, G))

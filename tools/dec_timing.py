#!/usr/bin/env python3
"""Phase timestamps inside decode_nms_kernel (debug build: tools/dec_timing.sh), BASELINE.json configs[1] netout (batch 8, C=80).
   MI355_DT_LIB=.../libmi355_dt_dectt.so python tools/dec_timing.py"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from bench import KerasYOLO, synth, make_frames

B, H, W = 8, 416, 416
blob = synth.synth_darknet_blob(80, seed=1234)
det = KerasYOLO({'LABELS': KerasYOLO.LABELS_COCO, 'BATCH_SIZE': B, 'IMAGE_H': H, 'IMAGE_W': W, 'GRID_H': 13, 'GRID_W': 13}, weights=blob)
ctx = det.model.ctx
frames = make_frames(1, B, H, W, ctx.device, seed0=7)[0].contiguous()
for _ in range(3):
    out = det.detect(frames)
torch.cuda.synchronize()
ctx.lib.dt_debug_dec_times.argtypes = [ctypes.c_void_p]
buf = np.zeros(16, dtype=np.uint64)
assert ctx.lib.dt_debug_dec_times(buf.ctypes.data_as(ctypes.c_void_p)) == 0
t = buf.astype(np.int64)
names = ["1 max/min of the class logits", "2 scores + candidates", "3 per-class NMS", "4 final filter"]
for i, n in enumerate(names):
    print("phase %-32s %8d cycles" % (n, t[i + 1] - t[i]))
steps = ["chunk load", "(a) exp / sigmoid", "(b) row sums", "(c) scores, threshold, lists", "(d) boxes + compaction", "post store"]
prev = t[1]
for i, n in enumerate(steps):
    print("   first chunk: %-30s %8d cycles" % (n, t[9 + i] - prev)); prev = t[9 + i]
print("whole workgroup %d cycles (100 MHz-independent shader clock); candidates in frame 0: %d" % (t[4] - t[0], t[8]))
ctx.profile_reset(); ctx.profile_enable(True)
for _ in range(10):
    det.detect(frames)
torch.cuda.synchronize()
ctx.profile_enable(False)
print("decode_nms launch %.4f ms" % (ctx.profile_read("decode_nms")["ms"] / 10))

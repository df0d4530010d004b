#!/usr/bin/env python3
"""Workload for the rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE in separate
runs, MI355X_MICROARCH.md "HBM"): a calibration copy of a KNOWN byte count
(float4-vectorised device copy of 1 GiB: 1 GiB read + 1 GiB written) followed by
exactly one detect+track step of bench.py's default workload, so that per-launch
counter values line up with bench.py's per-launch `roofline.achieved`."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench

clips = int(sys.argv[1]) if len(sys.argv) > 1 else 48
dev = torch.device("cuda", 0)
a = torch.empty(1 << 28, dtype=torch.float32, device=dev).normal_()
b = torch.empty_like(a)
b.copy_(a)                       # calibration kernel: 1 GiB in, 1 GiB out
torch.cuda.synchronize()
frames = bench.make_frames(clips, 30, 416, 416, dev, seed0=42)
trk, blob, tw = bench.build_tracker(416, 416, 30, 32, frames)   # includes one full forward (calibration of the head)
res = trk.track_clips(frames, cap=128)                           # the measured step
torch.cuda.synchronize()
print("pmc probe done: clips", clips, "boxes/frame", float(res["counts"].float().mean()))

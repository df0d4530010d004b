#!/usr/bin/env python3
"""Probe: do two half-batches on two HIP streams (two contexts) overlap the HBM-bound and the MFMA-bound phases?
   python tools/two_stream_probe.py [clips_total]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import object_tracking_amd  # noqa: F401
import mi355_dt
from utility import synth

ANCHORS = [0.57273, 0.677385, 1.87446, 2.06253, 3.33843, 5.47434, 7.88282, 3.52778, 9.77052, 9.16828]
C, T, H = 12, 30, 416
clips = int(sys.argv[1]) if len(sys.argv) > 1 else 48
blob = synth.synth_darknet_blob(C)
tw = synth.synth_tracker_weights(C, seed=1235)


def make():
    c = mi355_dt.Context()
    c.detector_config(H, H, 5, C, ANCHORS)
    c.load_darknet_weights(blob)
    c.tracker_load(512, tw["kernel"], tw["recurrent"], tw["bias"], tw["out_kernel"], tw["out_bias"])
    return c


dev = torch.device("cuda")
frames = torch.randint(0, 256, (clips, T, H, H, 3), dtype=torch.uint8, device=dev)


def timeit(fn, steps=4, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


c0 = make()
ms1 = timeit(lambda: c0.track_forward(frames, want_det=False))
print("one stream   : %d clips  %.2f ms  %.0f frames/s" % (clips, ms1, clips * T / ms1 * 1e3))
for nway in (2, 3):
    ctxs = [c0] + [make() for _ in range(nway - 1)]
    streams = [torch.cuda.Stream() for _ in range(nway)]
    parts = [p.contiguous() for p in torch.chunk(frames, nway, dim=0)]

    def run():
        cur = torch.cuda.current_stream()
        for s in streams:
            s.wait_stream(cur)
        for c, s, p in zip(ctxs, streams, parts):
            with torch.cuda.stream(s):
                c.track_forward(p, want_det=False)
        for s in streams:
            cur.wait_stream(s)
    ms = timeit(run)
    print("%d streams    : %d clips  %.2f ms  %.0f frames/s  (x%.3f)" % (nway, clips, ms, clips * T / ms * 1e3, ms1 / ms))
    del ctxs[1:]

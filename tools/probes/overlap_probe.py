#!/usr/bin/env python3
"""Probe: does an HBM-bound kernel on a second (unmasked) stream overlap the persistent MFMA GEMM?  The 16-wave GEMM
workgroup uses 64 VGPRs x 4 waves per SIMD and ~147 KB of LDS: half the register file and four wave slots per SIMD stay
free for a kernel without LDS.   python tools/overlap_probe.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import object_tracking_amd  # noqa: F401
import mi355_dt

os.environ["DT_CONV_CFG"] = "3"


def timeit(fn, iters=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


cg, cm = mi355_dt.Context(), mi355_dt.Context()
rs = np.random.RandomState(0)
x = torch.randn(2304, 13, 13, 1024, device=cg.device)
w = (rs.randn(1, 1, 1024, 1024) * 0.03).astype(np.float32)
feat = torch.randn(4096, 26, 26, 512, device=cg.device)      # 5.7 GB read by the pooling kernel
det = torch.rand(4096, 4, device=cg.device)
tw = __import__("utility.synth", fromlist=["x"]).synth_tiny_weights(512)
cm.tiny_load(516, 512, tw["kernel"], tw["recurrent"], tw["bias"], tw["dense_kernel"], tw["dense_bias"])
gflop = 2.0 * 2304 * 169 * 1024 * 1024 / 1e9
gbytes = feat.numel() * 4 / 1e9
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def gemm():
    with torch.cuda.stream(s1):       # the bindings launch on torch's current stream
        cg.conv2d(x, w, None, leaky_slope=1.0, pool=0)


def mem():
    with torch.cuda.stream(s2):
        cm.tiny_features(feat, det, 516)


tg = timeit(gemm); tm = timeit(mem)
print("alone: GEMM %.3f ms (%.1f TF)   pool %.3f ms (%.2f TB/s)" % (tg, gflop / tg, tm, gbytes / tm))
for k in (1, 2, 4):
    def both():
        gemm()
        for _ in range(k):
            mem()
    tb = timeit(both)
    print("GEMM || %d x pool on two streams: %.3f ms   (serial %.3f, perfect overlap %.3f)" % (k, tb, tg + k * tm, max(tg, k * tm)))

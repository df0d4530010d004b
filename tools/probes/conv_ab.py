#!/usr/bin/env python3
"""Single-layer timing through dt_conv2d (HIP events inside the library): python tools/conv_ab.py B H W Cin k Cout [pool]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import object_tracking_amd  # noqa: F401
import mi355_dt

B, H, W, Cin, k, Cout = [int(v) for v in sys.argv[1:7]]
pool = int(sys.argv[7]) if len(sys.argv) > 7 else 0
ctx = mi355_dt.Context()
rs = np.random.RandomState(0)
x = torch.randn(B, H, W, Cin, device=ctx.device)
w = (rs.randn(k, k, Cin, Cout) * np.sqrt(2.0 / (k * k * Cin))).astype(np.float32)
b = rs.randn(Cout).astype(np.float32)
for _ in range(2):
    ctx.conv2d(x, w, b, leaky_slope=0.1, pool=pool)
ctx.profile_reset(); ctx.profile_enable(True)
for _ in range(5):
    ctx.conv2d(x, w, b, leaky_slope=0.1, pool=pool)
ctx.profile_enable(False)
p = ctx.profile_read("conv_igemm")
pf = ctx.profile_read("conv_fused")
if pf["launches"]:
    p = pf
fl = 2.0 * B * H * W * k * k * Cin * Cout
ms = p["ms"] / 5
extra = ctx.profile_read("wino_input")["ms"] / 5 + ctx.profile_read("wino_output")["ms"] / 5
print("%s  conv %dx%dx%dx%d k%d -> %d pool %d: igemm %.3f ms  executed %.1f TF  direct-form %.1f TF (incl. transforms %.1f)" % (
    os.environ.get("TAG", ""), B, H, W, Cin, k, Cout, pool, ms, p["flops"] / 5 / ms / 1e9, fl / ms / 1e9, fl / (ms + extra) / 1e9))

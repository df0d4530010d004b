#!/usr/bin/env python3
"""one layer shape through dt_conv2d a few times (workload for rocprofv3 --pmc passes on a single kernel)
    python tools/fused4_probe.py conv_3 480"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import object_tracking_amd  # noqa: F401
import mi355_dt

SHAPES = {"conv_3": (104, 64, 128, 0), "conv_5": (104, 64, 128, 1), "conv_6": (52, 128, 256, 0), "conv_8": (52, 128, 256, 1),
          "conv_2": (208, 32, 64, 1)}
name = sys.argv[1] if len(sys.argv) > 1 else "conv_3"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 480
H, Cin, Cout, pool = SHAPES[name]
ctx = mi355_dt.Context()
rs = np.random.RandomState(0)
x = torch.randn((B, H, H, Cin), dtype=torch.float32, device=ctx.device)
w = (rs.randn(3, 3, Cin, Cout) * np.sqrt(2.0 / (9 * Cin))).astype(np.float32)
b = rs.randn(Cout).astype(np.float32)
for _ in range(3):
    ctx.conv2d(x, w, b, leaky_slope=0.1, pool=pool)
torch.cuda.synchronize()
print("probe done", name, B)

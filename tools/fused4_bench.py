#!/usr/bin/env python3
"""conv_3 / conv_5 / conv_6 / conv_8 at the bench batch (1440 frames): fused F(4x4) kernel vs the unfused form.
    python tools/fused4_bench.py [frames]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import object_tracking_amd  # noqa: F401
import mi355_dt

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1440
ctx = mi355_dt.Context()
rs = np.random.RandomState(0)
for (name, H, Cin, Cout, pool) in [("conv_2", 208, 32, 64, 1), ("conv_3", 104, 64, 128, 0), ("conv_5", 104, 64, 128, 1),
                                    ("conv_6", 52, 128, 256, 0), ("conv_8", 52, 128, 256, 1)]:
    x = torch.randn((B, H, H, Cin), dtype=torch.float32, device=ctx.device)
    w = (rs.randn(3, 3, Cin, Cout) * np.sqrt(2.0 / (9 * Cin))).astype(np.float32)
    b = rs.randn(Cout).astype(np.float32)
    for mode in ("0", "2"):     # unfused (conv_2: direct form) / wino4s_fused
        os.environ["DT_WINO_FUSED4"] = mode
        for _ in range(2):
            ctx.conv2d(x, w, b, leaky_slope=0.1, pool=pool)
        ctx.profile_reset(); ctx.profile_enable(True)
        for _ in range(3):
            ctx.conv2d(x, w, b, leaky_slope=0.1, pool=pool)
        ctx.profile_enable(False)
        parts = {n: ctx.profile_read(n) for n in ("conv_fused", "conv_igemm", "wino_input", "wino_output")}
        ms = sum(p["ms"] for p in parts.values()) / 3
        fl = (parts["conv_fused"]["flops"] + parts["conv_igemm"]["flops"]) / 3
        direct = 2.0 * B * H * H * 9 * Cin * Cout
        print("%-7s fused4=%s  %7.3f ms  executed %6.1f TFLOP/s  direct-form %6.1f TFLOP/s  (%s)" % (
            name, mode, ms, fl / ms / 1e9, direct / ms / 1e9,
            ", ".join("%s %.2f" % (n, p["ms"] / 3) for n, p in parts.items() if p["ms"] > 0)), flush=True)
    del x

#!/usr/bin/env python3
"""Timestamps inside the fused F(4x4) kernel (debug build: -DDT_W4_TIMING, tools/w4_timing.sh): workgroups 2048..2303.
   MI355_DT_LIB=.../libmi355_dt_w4tt.so python tools/w4_timing.py conv_3 480"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import object_tracking_amd  # noqa: F401
import mi355_dt

SHAPES = {"conv_3": (104, 64, 128, 0), "conv_5": (104, 64, 128, 1), "conv_6": (52, 128, 256, 0), "conv_8": (52, 128, 256, 1)}
name = sys.argv[1]; B = int(sys.argv[2])
H, Cin, Cout, pool = SHAPES[name]
os.environ["DT_WINO_FUSED4"] = "2"
ctx = mi355_dt.Context()
lib = ctx.lib
rs = np.random.RandomState(0)
x = torch.randn(B, H, H, Cin, device=ctx.device)
w = (rs.randn(3, 3, Cin, Cout) * np.sqrt(2.0 / (9 * Cin))).astype(np.float32)
b = rs.randn(Cout).astype(np.float32)
for _ in range(2):
    ctx.conv2d(x, w, b, leaky_slope=0.1, pool=pool)
torch.cuda.synchronize()
lib.dt_debug_w4_times.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert lib.dt_debug_w4_times(None, 1) == 0
ctx.profile_reset(); ctx.profile_enable(True)
ctx.conv2d(x, w, b, leaky_slope=0.1, pool=pool)
ctx.profile_enable(False)
ms = ctx.profile_read("conv_fused")["ms"]
NS = 24
buf = np.zeros(256 * 2 * NS * 6, dtype=np.uint64)
assert lib.dt_debug_w4_times(buf.ctypes.data_as(ctypes.c_void_p), 0) == 0
t = buf.reshape(256, 2, NS, 6).astype(np.int64)
nst = Cin // 8
print("%s %d frames: launch %.3f ms; %d stages per workgroup" % (name, B, ms, nst))
for wv, label in ((0, "wave 0 (transforms)"), (1, "wave 4")):
    tt = t[:, wv]
    ok = tt[:, 0, 0] > 0
    tt = tt[ok]
    pro = tt[:, 0]
    print("  %s: prologue: loads+stores %6.0f | barrier %6.0f | transform0+barrier %6.0f" % (
        label, (pro[:, 1] - pro[:, 0]).mean(), (pro[:, 2] - pro[:, 1]).mean(), (pro[:, 3] - pro[:, 2]).mean()))
    st = tt[:, 1:1 + nst]
    print("     per stage: transform %6.0f | 2 K-steps %6.0f (MFMA-only bound %d) | barrier wait %6.0f | stage total %6.0f" % (
        (st[:, :, 1] - st[:, :, 0]).mean(), (st[:, :, 2] - st[:, :, 1]).mean(), 2 * 2 * 36 * 32, (st[:, :, 3] - st[:, :, 2]).mean(),
        (st[:, :, 3] - st[:, :, 0]).mean()))
    ep = tt[:, 1 + nst]
    print("     epilogue: block 0 %6.0f | block 1 %6.0f | whole workgroup %7.0f cycles (MFMA-only bound %d)" % (
        (ep[:, 1] - ep[:, 0]).mean(), (ep[:, 2] - ep[:, 1]).mean(), (ep[:, 2] - pro[:, 0]).mean(), nst * 2 * 2 * 36 * 32))

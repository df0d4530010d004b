#!/usr/bin/env python3
"""Per-step timestamps of the fused F(4x4) kernel's persistent loop (debug build: -DDT_W4_TIMING, tools/w4_timing.sh).
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
buf = np.zeros(256 * 2 * 48 * 6, dtype=np.uint64)
assert lib.dt_debug_w4_times(buf.ctypes.data_as(ctypes.c_void_p), 0) == 0
t = buf.reshape(256, 2, 48, 6).astype(np.int64)
nhg = Cin // 16
print("%s %d frames: launch %.3f ms" % (name, B, ms))
for wv, label in ((0, "wave 0 (transforms)"), (1, "wave 4")):
    tt = t[:, wv]
    used = tt[:, :, 0] > 0
    step = (tt[:, 1:, 0] - tt[:, :-1, 0])[used[:, 1:]]
    tr = (tt[:, :, 1] - tt[:, :, 0])[used]
    mf = (tt[:, :, 2] - tt[:, :, 1])[used]
    ep = (tt[:, :, 3] - tt[:, :, 2])[used]
    ba = (tt[:, :, 4] - tt[:, :, 3])[used]
    # steps with an epilogue: h == nhg-1
    idx = np.arange(48) % nhg == nhg - 1
    epi_steps = (tt[:, :, 3] - tt[:, :, 2])[:, idx][used[:, idx]]
    print("  %s: cycles per step %8.0f | zero+transform %7.0f | 4 K-steps %7.0f (MFMA-only bound %d for 2 waves/SIMD) | epilogue (avg over steps) %7.0f, on block-end steps %7.0f | barrier wait %7.0f" % (
        label, step.mean(), tr.mean(), mf.mean(), 2 * 4 * 36 * 32, ep.mean(), epi_steps.mean(), ba.mean()))
span = (t[:, 0, :, 4].max(1) - t[:, 0, 0, 0])
print("  steps recorded per WG:", int((t[:, 0, :, 0] > 0).sum(1).mean()))

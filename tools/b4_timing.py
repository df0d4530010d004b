#!/usr/bin/env python3
"""Cycle sums inside wino4b_fused_kernel (debug build -DDT_B4_TIMING, tools/b4_timing.sh): workgroups 0..63, all 8 waves, the first 8
items of each.
   MI355_DT_LIB=tools/_probe_builds/libmi355_dt_b4tt.so python tools/b4_timing.py conv_3 1440"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import object_tracking_amd  # noqa: F401
import mi355_dt

SHAPES = {"conv_2": (208, 32, 64, 1), "conv_3": (104, 64, 128, 0), "conv_5": (104, 64, 128, 1)}
name = sys.argv[1]; B = int(sys.argv[2])
H, Cin, Cout, pool = SHAPES[name]
os.environ["DT_WINO_FUSED4"] = "2"; os.environ["DT_F4B"] = "1"
ctx = mi355_dt.Context()
lib = ctx.lib
rs = np.random.RandomState(0)
x = torch.randn(B, H, H, Cin, device=ctx.device)
w = (rs.randn(3, 3, Cin, Cout) * np.sqrt(2.0 / (9 * Cin))).astype(np.float32)
b = rs.randn(Cout).astype(np.float32)
for _ in range(2):
    ctx.conv2d(x, w, b, leaky_slope=0.1, pool=pool)
torch.cuda.synchronize()
lib.dt_debug_b4_times.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert lib.dt_debug_b4_times(None, 1) == 0
ctx.profile_reset(); ctx.profile_enable(True)
ctx.conv2d(x, w, b, leaky_slope=0.1, pool=pool)
ctx.profile_enable(False)
ms = ctx.profile_read("conv_fused")["ms"]
WG, NI, NSL, NWV = 64, 8, 8, 16
buf = np.zeros(WG * NWV * NI * NSL, dtype=np.uint64)
assert lib.dt_debug_b4_times(buf.ctypes.data_as(ctypes.c_void_p), 0) == 0
t = buf.reshape(WG, NWV, NI, NSL).astype(np.int64)
nst = 12 * (Cin // 16)
items = B * ((H + 31) // 32) ** 2 * (Cout // 64)
print("%s %d frames: launch %.3f ms = %.0f ns per item per CU; %d stages per item; MFMA-only bound per stage %d cycles" % (
    name, B, ms, ms * 1e6 / (items / 256.0), nst, 2 * 18 * 16))
tt = t[:, :, 1:, :]
ok = tt[..., 0] > 0
def m(a):
    return float(a[ok].mean())
print("  item total %7.0f cycles | loop %7.0f (per stage %5.0f) | epilogue %6.0f" % (
    m(tt[..., 2] - tt[..., 0]), m(tt[..., 1] - tt[..., 0]), m(tt[..., 1] - tt[..., 0]) / nst, m(tt[..., 2] - tt[..., 1])))
print("  per stage, mean over waves: dma issue %5.0f | input transform %5.0f | mfma phase %5.0f | Y accumulation %5.0f | vmcnt + barrier wait %5.0f" % (
    m(tt[..., 3]) / nst, m(tt[..., 4]) / nst, m(tt[..., 5]) / nst, m(tt[..., 6]) / nst, m(tt[..., 7]) / nst))
for wv in range(NWV):
    sel = tt[:, wv]
    o = sel[..., 0] > 0
    print("    wave %d: dma %5.0f transform %5.0f mfma %5.0f yacc %5.0f wait %5.0f | epilogue %6.0f" % (
        wv, sel[..., 3][o].mean() / nst, sel[..., 4][o].mean() / nst, sel[..., 5][o].mean() / nst, sel[..., 6][o].mean() / nst,
        sel[..., 7][o].mean() / nst, (sel[..., 2] - sel[..., 1])[o].mean()))

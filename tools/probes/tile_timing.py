#!/usr/bin/env python3
"""Per-tile timestamps of the persistent GEMM loop (debug build: hipcc -DDT_TILE_TIMING of conv_igemm.hip linked as
libmi355_dt_tt.so).  Runs one 1x1 GEMM-shaped launch and prints where a tile's time goes.
   MI355_DT_LIB=.../libmi355_dt_tt.so DT_CONV_CFG=3 python tools/tile_timing.py B H W Cin Cout"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import object_tracking_amd  # noqa: F401
import mi355_dt

B, H, W, Cin, Cout = [int(v) for v in sys.argv[1:6]]
ctx = mi355_dt.Context()
lib = ctx.lib
rs = np.random.RandomState(0)
x = torch.randn(B, H, W, Cin, device=ctx.device)
w = (rs.randn(1, 1, Cin, Cout) * np.sqrt(2.0 / Cin)).astype(np.float32)
for _ in range(2):
    ctx.conv2d(x, w, None, leaky_slope=1.0, pool=0)
torch.cuda.synchronize()
lib.dt_debug_tile_times.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert lib.dt_debug_tile_times(None, 1) == 0
ctx.conv2d(x, w, None, leaky_slope=1.0, pool=0)
torch.cuda.synchronize()
buf = np.zeros(512 * 48 * 4, dtype=np.uint64)
assert lib.dt_debug_tile_times(buf.ctypes.data_as(ctypes.c_void_p), 0) == 0
t = buf.reshape(512, 48, 4).astype(np.int64)
used = t[:, :, 0] > 0
nb = int(used[:, 0].sum())
print("workgroups that ran:", nb, " tiles per workgroup: min %d max %d" % (used.sum(1)[used[:, 0]].min(), used.sum(1).max()))
# __builtin_readcyclecounter() counts shader-clock cycles; the rate is calibrated from the launch's HIP-event time
ctx.profile_reset(); ctx.profile_enable(True)
ctx.conv2d(x, w, None, leaky_slope=1.0, pool=0)
ctx.profile_enable(False)
ms = ctx.profile_read("conv_igemm")["ms"]
loop = (t[:, :, 1] - t[:, :, 0])[used]
last = (t[:, :, 2] - t[:, :, 1])[used]
epi = (t[:, :, 3] - t[:, :, 2])[used]
start0 = t[:, 0, 0][used[:, 0]]
end0 = np.array([t[b, used[b].sum() - 1, 3] for b in range(512) if used[b, 0]])
clk = float(np.median(end0 - start0)) / (ms * 1e-3)     # counters of different XCDs are not synchronised: per-workgroup spans only
print("launch %.3f ms by HIP events -> counter runs at %.3f GHz" % (ms, clk / 1e9))
for name, v in (("chunks 0..n-2", loop), ("last chunk (+next-tile setup)", last), ("epilogue", epi)):
    print("  %-30s mean %8.2f us   p10 %8.2f  p90 %8.2f" % (name, v.mean() / clk * 1e6, np.percentile(v, 10) / clk * 1e6, np.percentile(v, 90) / clk * 1e6))
tile = (t[:, 1:, 0] - t[:, :-1, 0])[used[:, 1:]]
print("  %-30s mean %8.2f us" % ("tile start to next tile start", tile.mean() / clk * 1e6))
start = t[:, 0, 0][used[:, 0]]
end = np.array([t[b, used[b].sum() - 1, 3] for b in range(512) if used[b, 0]])
span = (end - start) / clk * 1e6
print("  per-workgroup busy span: min %.1f us  median %.1f  max %.1f  (launch %.1f us)" % (span.min(), np.median(span), span.max(), ms * 1e3))
nk = Cin // 32
print("  cycles per chunk (chunks 0..n-2): %.0f; MFMA-only bound 16384 (64 MFMAs x 4 waves x 64 cycles per SIMD) -> %.1f %%" % (
    loop.mean() / max(1, nk - 1), 100.0 * 16384.0 / (loop.mean() / max(1, nk - 1))))
tot = tile.mean()
print("  share of a tile: chunk loop %.1f %%, last chunk %.1f %%, epilogue %.1f %%; MFMA-only bound / tile = %.1f %%" % (
    100 * loop.mean() / tot, 100 * last.mean() / tot, 100 * epi.mean() / tot, 100.0 * 16384.0 * nk / tot))

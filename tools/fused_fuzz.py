#!/usr/bin/env python3
"""Randomized frame sizes through the fused conv_2 kernel vs the oracle:  python tools/fused_fuzz.py [seed] [count]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import object_tracking_amd, mi355_dt  # noqa: F401
from oracle import oracle as orc
ctx = mi355_dt.Context()
os.environ["DT_WINO_FUSED"] = "2"
rs = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
bad = 0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
    B = int(rs.randint(1, 7)); H = 2 * int(rs.randint(1, 41)); W = 2 * int(rs.randint(1, 41))
    x = rs.randn(B, H, W, 32).astype(np.float32)
    w = (rs.randn(3, 3, 32, 64) * np.sqrt(2.0 / 288)).astype(np.float32)
    b = rs.randn(64).astype(np.float32)
    ref = orc.conv2d(x, w, b); ref = orc.maxpool2(np.where(ref > 0, ref, ref * np.float32(0.1)).astype(np.float32))
    ctx.profile_reset(); ctx.profile_enable(True)
    got = ctx.conv2d(torch.from_numpy(x).to(ctx.device), w, b, leaky_slope=0.1, pool=1).cpu().numpy()
    ctx.profile_enable(False)
    used = ctx.profile_read("conv_fused")["launches"] == 1
    e = float(np.abs(got - ref).max() / np.abs(ref).max())
    ok = used and e < 2e-5
    bad += not ok
    print("%s B=%d H=%d W=%d fused=%s err=%.2e" % ("ok " if ok else "BAD", B, H, W, used, e), flush=True)
print("bad:", bad)

#!/usr/bin/env python3
"""Randomized shapes through the Winograd path vs the oracle:  python tools/wino_fuzz.py [seed] [count]"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import object_tracking_amd, mi355_dt
from oracle import oracle as orc
ctx = mi355_dt.Context()
os.environ["DT_WINO"] = "2"
rs = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
bad = 0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 60):
    ts = int(rs.choice([2, 4, 6])); os.environ["DT_WINO_TILE"] = str(ts)
    B = int(rs.randint(1, 21)); H = int(rs.randint(1, 21)) * (2 if rs.rand() < 0.5 else 1); W = int(rs.randint(1, 21)) * (2 if rs.rand() < 0.5 else 1)
    Cin = int(rs.choice([32, 64, 96])); Cout = int(rs.randint(1, 41)) * 4
    pool = int(rs.choice([0, 1, 2])) if (H % 2 == 0 and W % 2 == 0) else 0
    mos = rs.choice(["", "1", "2", "3", "4"])
    if mos: os.environ["DT_WINO_MOSAIC"] = mos
    else: os.environ.pop("DT_WINO_MOSAIC", None)
    x = rs.randn(B, H, W, Cin).astype(np.float32)
    w = (rs.randn(3, 3, Cin, Cout) * np.sqrt(2.0 / (9 * Cin))).astype(np.float32)
    b = rs.randn(Cout).astype(np.float32)
    ref = orc.conv2d(x, w, b); ref = np.where(ref > 0, ref, ref * np.float32(0.1)).astype(np.float32)
    got = ctx.conv2d(torch.from_numpy(x).to(ctx.device), w, b, leaky_slope=0.1, pool=pool)
    rel = lambda a, c: float(np.abs(a - c).max() / (np.abs(c).max() + 1e-12))
    if pool == 0: e = rel(got.cpu().numpy(), ref)
    elif pool == 1: e = rel(got.cpu().numpy(), orc.maxpool2(ref))
    else: e = max(rel(got[0].cpu().numpy(), ref), rel(got[1].cpu().numpy(), orc.maxpool2(ref)))
    ok = e < {2: 2e-5, 4: 1e-4, 6: 3e-4}[ts]
    if not ok:
        bad += 1
    print("%s ts=%d B=%d H=%d W=%d Cin=%d Cout=%d pool=%d mosaic=%s err=%.2e" % ("ok " if ok else "BAD", ts, B, H, W, Cin, Cout, pool, mos or "auto", e), flush=True)
print("bad:", bad)

"""The tracking step at a given number of clips: HIP-event time of every profiled scope (which kernel each layer took)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import bench
C = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device("cuda", 0)
frames = bench.make_frames(C, 30, 416, 416, dev, seed0=42)
trk, _, _ = bench.build_tracker(416, 416, 30, 32, frames)
ctx = trk.model.ctx
for _ in range(5): trk.track_clips(frames, cap=128)
torch.cuda.synchronize()
ctx.profile_enable(True); ctx.profile_reset()
N = 10
for _ in range(N): trk.track_clips(frames, cap=128)
torch.cuda.synchronize()
rows = [(ctx.profile_read(n)["ms"] / N, n, ctx.profile_read(n)["launches"] // N) for n in ctx.profile_names() if ctx.profile_read(n)["launches"]]
tot = sum(ms for ms, n, _ in rows if ":" not in n)
print("clips %d: sum of scopes %.3f ms" % (C, tot))
for ms, nm, l in sorted(rows, reverse=True):
    if ms > 0.01: print("  %-34s %3d  %.4f ms" % (nm, l, ms))

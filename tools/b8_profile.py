"""BASELINE configs[1] (YOLOv2 C=80, batch 8, 416x416 uint8): wall time per detect() and the HIP-event time of every profiled scope."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from object_tracking_amd.models_detection.KerasYOLO import KerasYOLO
from utility import synth

B, H, W = 8, 416, 416
det = KerasYOLO({'LABELS': KerasYOLO.LABELS_COCO, 'BATCH_SIZE': B, 'IMAGE_H': H, 'IMAGE_W': W, 'GRID_H': 13, 'GRID_W': 13},
                weights=synth.synth_darknet_blob(80, seed=1234))
ctx = det.model.ctx
frames = torch.from_numpy(synth.synth_clip(B, H, W, 32, seed=7)).cuda().contiguous()
for _ in range(10): det.detect(frames)
torch.cuda.synchronize()
t0 = time.perf_counter(); n = 100
for _ in range(n): det.detect(frames)
torch.cuda.synchronize()
print("wall %.4f ms per batch" % (1e3 * (time.perf_counter() - t0) / n))
ctx.profile_enable(True); ctx.profile_reset()
for _ in range(20): det.detect(frames)
torch.cuda.synchronize()
rows = []
for nm in ctx.profile_names():
    r = ctx.profile_read(nm)
    if r["launches"]: rows.append((r["ms"] / 20, nm, r["launches"] // 20))
tot = sum(ms for ms, nm, _ in rows if ":" not in nm)
for ms, nm, l in sorted(rows, reverse=True): print("%-34s %3d  %.4f ms" % (nm, l, ms))
print("sum of top-level scopes %.4f ms" % tot)

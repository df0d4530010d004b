"""YOLOv2 C=80 detector forward at a given batch: HIP-event time of every profiled scope (which kernel each layer took)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import object_tracking_amd  # noqa
from models_detection.KerasYOLO import KerasYOLO
from utility import synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 20
H = W = 416
det = KerasYOLO({'LABELS': KerasYOLO.LABELS_COCO, 'BATCH_SIZE': B, 'IMAGE_H': H, 'IMAGE_W': W, 'GRID_H': 13, 'GRID_W': 13}, weights=synth.synth_darknet_blob(80, seed=1234))
ctx = det.model.ctx
frames = torch.from_numpy(synth.synth_clip(B, H, W, 32, seed=7)).cuda().contiguous()
for _ in range(5): det.detect(frames)
torch.cuda.synchronize()
ctx.profile_enable(True); ctx.profile_reset()
for _ in range(10): det.detect(frames)
torch.cuda.synchronize()
rows = [(ctx.profile_read(n)["ms"] / 10, n, ctx.profile_read(n)["launches"] // 10) for n in ctx.profile_names() if ctx.profile_read(n)["launches"]]
for ms, nm, l in sorted(rows, reverse=True):
    if ":" in nm and ms > 0.004: print("B=%d %-34s %3d  %.4f ms" % (B, nm, l, ms))

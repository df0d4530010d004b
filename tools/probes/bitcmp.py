"""Bit-comparison of two builds of the library on the same inputs: run once per build (MI355_DT_LIB=...), `save <file>`, then `cmp a b`.
Covers the 48-clip tracker step (netout of every frame, boxes, ids) and a batch-8 C=80 detector forward."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np

if sys.argv[1] == "cmp":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    bad = 0
    for k in a.files:
        same = a[k].shape == b[k].shape and np.array_equal(a[k].view(np.uint8), b[k].view(np.uint8))
        d = float(np.abs(a[k].astype(np.float64) - b[k].astype(np.float64)).max()) if a[k].shape == b[k].shape and a[k].size else -1
        print("%-12s %-22s %s  max|diff| %.3g" % (k, a[k].shape, "bit-identical" if same else "DIFFERENT", d))
        bad += not same
    sys.exit(1 if bad else 0)

import torch
import bench
from models_detection.KerasYOLO import KerasYOLO
from utility import synth
dev = torch.device("cuda:0")
clips = int(os.environ.get("CLIPS", "12"))
frames = bench.make_frames(clips, 30, 416, 416, dev, seed0=100)
trk, _, _ = bench.build_tracker(416, 416, 30, 32, frames)
res = trk.track_clips(frames)
out = {"counts": res["counts"].cpu().numpy(), "ids": res["ids"].cpu().numpy(), "boxes": res["boxes"].cpu().numpy()}
z = trk.model.forward(frames, want_det=True)
out["trk_grid"] = z[0].float().cpu().numpy(); out["det_grid"] = z[1].float().cpu().numpy()
det = KerasYOLO({'LABELS': KerasYOLO.LABELS_COCO, 'BATCH_SIZE': 8, 'IMAGE_H': 416, 'IMAGE_W': 416, 'GRID_H': 13, 'GRID_W': 13}, weights=synth.synth_darknet_blob(80, seed=1234))
f8 = torch.from_numpy(synth.synth_clip(8, 416, 416, 32, seed=7)).cuda().contiguous()
out["net8"] = det.model.ctx.detect_forward(f8).cpu().numpy()
f64 = torch.from_numpy(synth.synth_clip(16, 416, 416, 32, seed=9)).cuda().contiguous().repeat(4, 1, 1, 1)
out["net64"] = det.model.ctx.detect_forward(f64).cpu().numpy()
np.savez(sys.argv[2], **out)
print("saved", sys.argv[2], {k: v.shape for k, v in out.items()})

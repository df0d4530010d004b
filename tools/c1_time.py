"""conv_1 alone inside a 1440-frame detector forward (HIP-event time of the `conv1_direct` scope): A/B of conv1.hip builds
selected with MI355_DT_LIB (tools/build_variant.sh c1x conv1.hip -D...)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from object_tracking_amd import mi355_dt
from object_tracking_amd.models_detection.KerasYOLO import KerasYOLO
from utility import synth

B, H, W = 1440, 416, 416
blob = synth.synth_darknet_blob(12, seed=1234)
det = KerasYOLO({'LABELS': ['c%d' % i for i in range(12)], 'BATCH_SIZE': B, 'IMAGE_H': H, 'IMAGE_W': W, 'GRID_H': 13, 'GRID_W': 13}, weights=blob)
ctx = det.model.ctx
g = torch.Generator(device="cuda").manual_seed(1)
frames = torch.randint(0, 256, (B, H, W, 3), dtype=torch.uint8, device="cuda", generator=g)
for _ in range(2):
    ctx.detect_forward_internal(frames)
ctx.profile_enable(True); ctx.profile_reset()
n = 4
for _ in range(n):
    ctx.detect_forward_internal(frames)
torch.cuda.synchronize()
r = ctx.profile_read("conv1_direct")
print("%s conv1_direct %.3f ms" % (os.environ.get("MI355_DT_LIB", "default").split("_")[-1], r["ms"] / max(1, r["launches"])))

import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import object_tracking_amd, mi355_dt
from utility import synth
ANCH=[0.57273, 0.677385, 1.87446, 2.06253, 3.33843, 5.47434, 7.88282, 3.52778, 9.77052, 9.16828]
for mode in ("random", "zeros"):
    ctx = mi355_dt.Context()
    ctx.detector_config(416, 416, 5, 12, ANCH)
    blob = synth.synth_darknet_blob(12)
    if mode == "zeros":
        blob = np.zeros_like(blob); 
        # var must be > 0 for the BN fold; zeros give scale = 0/sqrt(eps) = 0 anyway
    ctx.load_darknet_weights(blob)
    frames = (torch.randint(0, 256, (480, 416, 416, 3), dtype=torch.uint8, device="cuda") if mode == "random"
              else torch.zeros((480, 416, 416, 3), dtype=torch.uint8, device="cuda"))
    for _ in range(2): ctx.detect_forward(frames)
    ctx.profile_reset(); ctx.profile_enable(True)
    for _ in range(3): ctx.detect_forward(frames)
    ctx.profile_enable(False)
    p = ctx.profile_read("conv_igemm"); q = ctx.profile_read("conv_igemm:conv_19")
    print(mode, "conv family TF", round(p["flops"]/(p["ms"]*1e-3)/1e12, 2), "conv_19 TF", round(q["flops"]/(q["ms"]*1e-3)/1e12, 2))

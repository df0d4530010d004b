"""The narrow form of the split GEMM (128 x 128 tiles, three workgroups per CU) against the default selection at the short launches' shapes: time of the GEMM scope, bits."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import object_tracking_amd  # noqa
import mi355_dt
ctx = mi355_dt.Context()
g = torch.Generator(device="cuda"); g.manual_seed(5)
for (P, Mt, K, N, what) in ((36, 16, 512, 2048, "recurrent step, 1 clip"), (36, 49, 512, 2048, "recurrent step, 2-4 clips"), (36, 98, 512, 2048, "8 clips"),
                            (64, 75, 1024, 1024, "conv_19 at 12 frames"), (64, 98, 1024, 1024, "16 frames"), (64, 125, 1024, 1024, "20 frames"), (64, 49, 1024, 1024, "8 frames"),
                            (64, 98, 512, 1024, "conv_14 at 16 frames"), (64, 98, 1280, 1024, "conv_22 at 16 frames")):
    v = torch.randn((P, Mt, K), generator=g, device="cuda")
    u = torch.randn((P, N, K), generator=g, device="cuda") / K ** 0.5
    outs, ms = {}, {}
    for half in (0, 3):
        for _ in range(3): outs[half] = ctx.gemm_split(v, u, half=half, nt=2)
        torch.cuda.synchronize()
        ctx.profile_reset(); ctx.profile_enable(True)
        for _ in range(20): ctx.gemm_split(v, u, half=half, nt=2)
        torch.cuda.synchronize(); ctx.profile_enable(False)
        r = ctx.profile_read("conv_gemm_s3"); ms[half] = r["ms"] / r["launches"]
    ref = torch.bmm(v.double(), u.double().transpose(1, 2))
    print("P=%2d Mt=%3d K=%4d N=%4d  default %.1f us  narrow %.1f us  same bits %s  err %.2e   # %s" % (
        P, Mt, K, N, 1e3 * ms[0], 1e3 * ms[3], bool(torch.equal(outs[0], outs[3])), float((outs[3].double() - ref).abs().max()), what))

#!/usr/bin/env python3
"""Per-kernel measurements of the bandwidth/latency-bound kernels of the path (the MFMA
kernel is covered by bench.py): algorithmic bytes / HIP-event time -> GB/s against the
8 TB/s HBM3E roof (6.3 TB/s achievable, MI355X_MICROARCH.md).  Writes a small table.

    python tools/bench_kernels.py > profiles/rNN_kernels.txt
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import object_tracking_amd  # noqa: F401
import mi355_dt
from utility import synth

ANCHORS = [0.57273, 0.677385, 1.87446, 2.06253, 3.33843, 5.47434, 7.88282, 3.52778, 9.77052, 9.16828]
PEAK = 8000.0


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters   # ms   (kernels are launched on torch's current stream)


def row(name, shape, ms, nbytes, note=""):
    gbs = nbytes / (ms * 1e-3) / 1e9
    print("%-26s %-34s %9.4f ms %9.1f GB/s %6.1f%% of 8 TB/s  %s" % (name, shape, ms, gbs, 100 * gbs / PEAK, note))


def planted(seed, G, C, n_obj):
    rs = np.random.RandomState(seed)
    g = rs.randn(G, G, 5, 5 + C).astype(np.float32)
    g[..., 4] -= 4.0
    for cell in rs.permutation(G * G)[:n_obj]:
        r, c = divmod(int(cell), G)
        b = int(rs.randint(0, 5))
        g[r, c, b, 4] = 4.0 + rs.rand()
        g[r, c, b, 5 + int(rs.randint(0, C))] += 12.0 + rs.rand()
    return g


def main():
    ctx = mi355_dt.Context()
    dev = ctx.device
    print("# kernel                     shape                                   time        algorithmic rate")
    # ingest: 1080p -> 416
    src = torch.randint(0, 256, (64, 1080, 1920, 3), dtype=torch.uint8, device=dev)
    ms = timeit(lambda: ctx.ingest_resize(src, 416, 416))
    row("ingest_resize", "64 x 1080x1920 -> 416x416 u8", ms, 64 * 3.0 * (1080 * 1920 + 416 * 416),
        "(only ~4/25 of the source pixels are touched when downscaling 1080p)")
    src = torch.randint(0, 256, (256, 480, 640, 3), dtype=torch.uint8, device=dev)
    ms = timeit(lambda: ctx.ingest_resize(src, 416, 416))
    row("ingest_resize", "256 x 480x640 -> 416x416 u8", ms, 256 * 3.0 * (480 * 640 + 416 * 416))
    # decode
    for (G, C, n_obj, B) in ((13, 12, 32, 1440), (13, 80, 24, 256), (19, 12, 128, 256)):
        grids = torch.from_numpy(np.stack([planted(i, G, C, n_obj) for i in range(16)])).to(dev)
        grids = grids.repeat((B + 15) // 16, 1, 1, 1, 1)[:B].contiguous()
        ms = timeit(lambda: ctx.decode(grids, 0.5, 0.45, ANCHORS, C, cap=128))
        row("decode_nms", "%d x %dx%dx5x%d, %d obj" % (B, G, G, 5 + C, n_obj), ms, 3.0 * grids.numel() * 4,
            "(read + staged write + NMS re-read of the frame)")
    # associate
    r = ctx.decode(grids, 0.5, 0.45, ANCHORS, 12, cap=128)
    boxes = r["boxes"][:240].reshape(8, 30, 128, 8).contiguous()
    counts = r["counts"][:240].reshape(8, 30).contiguous()
    ms = timeit(lambda: ctx.associate(boxes, counts, 0.3))
    row("associate", "8 clips x 30 frames, ~%d boxes" % int(counts.float().mean()), ms, boxes.numel() * 4 * 2.0,
        "(sequential in t and box: latency-bound)")
    # training-target encoding (8f.3): the kernel writes y and b in full, float64
    for (n, cap, G, C, TBB) in ((1024, 50, 13, 12, 50), (1024, 50, 13, 80, 50), (256, 128, 19, 12, 50)):
        rs = np.random.RandomState(3)
        IM = 32 * G
        dims = np.stack([rs.randint(400, 2000, n), rs.randint(300, 1200, n)], 1).astype(np.int32)
        objs = np.zeros((n, cap, 5), dtype=np.int32)
        objs[:, :, 0] = rs.randint(0, 400, (n, cap)); objs[:, :, 1] = rs.randint(0, 300, (n, cap))
        objs[:, :, 2] = objs[:, :, 0] + rs.randint(1, 300, (n, cap)); objs[:, :, 3] = objs[:, :, 1] + rs.randint(1, 300, (n, cap))
        objs[:, :, 4] = rs.randint(0, C, (n, cap))
        counts = rs.randint(0, cap + 1, n).astype(np.int32)
        d_o, d_c, d_d = [torch.from_numpy(a).to(dev) for a in (objs, counts, dims)]
        ms = timeit(lambda: ctx.encode_targets(d_o, d_c, d_d, None, G, G, 5, C, IM, IM, TBB, ANCHORS))
        row("encode_targets", "%d x %dx%dx5x%d f64, <=%d obj" % (n, G, G, 5 + C, cap), ms,
            8.0 * n * (G * G * 5 * (5 + C) + 4 * TBB) + 20.0 * n * cap, "(write-only: y and b are produced in full)")
    # TinyTracker pieces
    tw = synth.synth_tiny_weights(512)
    ctx.tiny_load(516, 512, tw["kernel"], tw["recurrent"], tw["bias"], tw["dense_kernel"], tw["dense_bias"])
    feat = torch.randn(2048, 26, 26, 512, device=dev)
    det = torch.rand(2048, 4, device=dev)
    ms = timeit(lambda: ctx.tiny_features(feat, det, 516), iters=10)
    row("global_maxpool (+concat)", "2048 x 26x26x512 f32", ms, feat.numel() * 4.0)
    ctx.profile_reset(); ctx.profile_enable(True)
    x = torch.randn(32, 64, 516, device=dev)
    for _ in range(5):
        ctx.tiny_sequence(x)
    ctx.profile_enable(False)
    p = ctx.profile_read("lstm_step")
    ms = p["ms"] / p["launches"]
    row("lstm_step", "32 tracks, U=512", ms, 4.0 * (512 * 2048 + 32 * (6 * 512 + 2048)),
        "(4.2 MB of U per step, L2-resident; launch-latency-bound)")
    x = torch.randn(128, 64, 516, device=dev)
    ctx.profile_reset(); ctx.profile_enable(True)
    for _ in range(5):
        ctx.tiny_sequence(x)
    ctx.profile_enable(False)
    p = ctx.profile_read("lstm_step")
    row("lstm_step", "128 tracks, U=512", p["ms"] / p["launches"], 4.0 * (512 * 2048 + 128 * (6 * 512 + 2048)))
    # conv_1
    from models_detection.KerasYOLO import KerasYOLO  # noqa: F401  (weights come from the full detector)
    c2 = mi355_dt.Context()
    c2.detector_config(416, 416, 5, 12, ANCHORS)
    c2.load_darknet_weights(synth.synth_darknet_blob(12))
    frames = torch.randint(0, 256, (64, 416, 416, 3), dtype=torch.uint8, device=dev)
    c2.profile_enable(True)
    for _ in range(5):
        c2.detect_forward(frames)
    c2.profile_enable(False)
    p = c2.profile_read("conv1_direct")
    row("conv1_direct (+x/255+pool)", "64 x 416x416x3 u8 -> 208x208x32", p["ms"] / p["launches"], p["bytes"] / p["launches"],
        "(%.1f TFLOP/s VALU; 0.30 GFLOP/frame)" % (p["flops"] / (p["ms"] * 1e-3) / 1e12))


if __name__ == "__main__":
    main()

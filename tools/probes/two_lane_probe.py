"""GPU box: does the step gain from running as TWO half-batches on two plain (unmasked) streams, each in its own context?
The GEMMs are bound by the matrix pipe / power, the Winograd transforms by HBM: one lane's transforms can run under the other's GEMMs.
    python tools/probes/two_lane_probe.py [clips] [lanes]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import bench

clips = int(sys.argv[1]) if len(sys.argv) > 1 else 48
lanes = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = torch.device("cuda", 0)
T, size, boxes = 30, 416, 32
frames = bench.make_frames(clips, T, size, size, dev, seed0=42)
trks = [bench.build_tracker(size, size, T, boxes, frames[:max(1, clips // lanes)])[0] for _ in range(lanes)]
one = bench.build_tracker(size, size, T, boxes, frames)[0]
cap = 128
parts = [frames[i * clips // lanes:(i + 1) * clips // lanes].contiguous() for i in range(lanes)]
streams = [torch.cuda.Stream(device=dev) for _ in range(lanes)]


def step_one():
    return one.track_clips(frames, cap=cap)


def step_lanes():
    cur = torch.cuda.current_stream(dev)
    outs = []
    for trk, part, st in zip(trks, parts, streams):
        st.wait_stream(cur)
        with torch.cuda.stream(st):
            outs.append(trk.track_clips(part, cap=cap))
    for st in streams:
        cur.wait_stream(st)
    return outs


def timeit(fn, n=6, w=3):
    for _ in range(w):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for rep in range(2):
    a = timeit(step_one)
    b = timeit(step_lanes)
    print("round %d: one stream %d clips %.2f ms (%.0f frames/s) | %d lanes x %d clips %.2f ms (%.0f frames/s)" % (
        rep, clips, a, clips * T / a * 1e3, lanes, clips // lanes, b, clips * T / b * 1e3), flush=True)
# same per-lane batch as the one-stream case (twice the frames in flight)
if len(sys.argv) > 3:
    parts = [frames for _ in range(lanes)]
    b = timeit(step_lanes)
    print("%d lanes x %d clips each: %.2f ms per %d frames (%.0f frames/s)" % (lanes, clips, b, lanes * clips * T, lanes * clips * T / b * 1e3))

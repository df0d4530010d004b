"""Experiment: the 48-clip step as TWO half-batches on two streams with complementary CU masks (hipExtStreamCreateWithCUMask), one
context each, against the same 48 clips on one stream.  If the split-GEMM phases are power-capped while the HBM-bound phases are not,
two half-chip partitions in different phases should finish sooner than the whole chip doing one phase after the other."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench

hip = ctypes.CDLL("libamdhip64.so")
def masked_stream(first_cu, n_cu, total=256):
    words = (total + 31) // 32
    mask = (ctypes.c_uint32 * words)()
    for cu in range(first_cu, first_cu + n_cu):
        mask[cu // 32] |= 1 << (cu % 32)
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), words, mask)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value)

def interleaved_stream(phase, total=256):      # every other CU
    words = (total + 31) // 32
    mask = (ctypes.c_uint32 * words)()
    for cu in range(phase, total, 2):
        mask[cu // 32] |= 1 << (cu % 32)
    st = ctypes.c_void_p()
    assert hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), words, mask) == 0
    return torch.cuda.ExternalStream(st.value)

dev = torch.device("cuda:0")
clips, T = int(os.environ.get("CLIPS", "48")), 30
frames = bench.make_frames(clips, T, 416, 416, dev, seed0=100)
half = clips // 2
fa, fb = frames[:half].contiguous(), frames[half:].contiguous()
trk_full, _, _ = bench.build_tracker(416, 416, T, 32, frames)
trk_a, _, _ = bench.build_tracker(416, 416, T, 32, fa)
trk_b, _, _ = bench.build_tracker(416, 416, T, 32, fb)

def timeit(fn, n=5, w=2):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n

def one():
    trk_full.track_clips(frames, cap=128)
print("one stream, %d clips: %.2f ms" % (clips, timeit(one)))
def seq_halves():
    trk_a.track_clips(fa, cap=128); trk_b.track_clips(fb, cap=128)
print("one stream, two halves one after the other: %.2f ms" % timeit(seq_halves))
for name, sa, sb in (("plain streams", torch.cuda.Stream(), torch.cuda.Stream()),
                     ("CU masks 0-127 / 128-255", masked_stream(0, 128), masked_stream(128, 128)),
                     ("CU masks even / odd", interleaved_stream(0), interleaved_stream(1)),
                     ("CU masks 0-159 / 96-255 (overlapping)", masked_stream(0, 160), masked_stream(96, 160))):
    def two():
        with torch.cuda.stream(sa): trk_a.track_clips(fa, cap=128)
        with torch.cuda.stream(sb): trk_b.track_clips(fb, cap=128)
    try:
        print("two streams, %s: %.2f ms" % (name, timeit(two)))
    except Exception as e:
        print("two streams, %s: failed: %r" % (name, e))

# steady state with the two partitions half a step out of phase: stream B starts behind an extra half-size forward; per-stream
# step time between events recorded after steps 2 and 10
def steady(sa, sb, offset):
    torch.cuda.synchronize()
    ev = {k: torch.cuda.Event(enable_timing=True) for k in ("a0", "a1", "b0", "b1")}
    if offset:
        with torch.cuda.stream(sb): trk_b.model.forward(fb[: max(1, half // 2)].contiguous(), want_det=False)
    for i in range(12):
        with torch.cuda.stream(sa):
            trk_a.track_clips(fa, cap=128)
            if i == 1: ev["a0"].record(sa)
            if i == 9: ev["a1"].record(sa)
        with torch.cuda.stream(sb):
            trk_b.track_clips(fb, cap=128)
            if i == 1: ev["b0"].record(sb)
            if i == 9: ev["b1"].record(sb)
    torch.cuda.synchronize()
    return ev["a0"].elapsed_time(ev["a1"]) / 8, ev["b0"].elapsed_time(ev["b1"]) / 8
for name, mk in (("plain streams", lambda: (torch.cuda.Stream(), torch.cuda.Stream())),
                 ("CU masks 0-127 / 128-255", lambda: (masked_stream(0, 128), masked_stream(128, 128)))):
    for off in (False, True):
        sa, sb = mk()
        a, b = steady(sa, sb, off)
        print("steady state, %s, %s: %.2f / %.2f ms per step of %d + %d clips" % (name, "half a step out of phase" if off else "in phase", a, b, half, clips - half))

#!/usr/bin/env python3
"""Probe: CU-masked HIP streams (hipExtStreamCreateWithCUMask) -- can an HBM-bound kernel on a few CUs run next to
the persistent MFMA GEMM on the rest, and at what rates?   python tools/cu_mask_probe.py [n_mem_cus]"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import object_tracking_amd  # noqa: F401
import mi355_dt

hip = ctypes.CDLL("libamdhip64.so")
NCU = torch.cuda.get_device_properties(0).multi_processor_count
n_mem = int(sys.argv[1]) if len(sys.argv) > 1 else 32


def masked_stream(bits):
    words = (NCU + 31) // 32
    arr = (ctypes.c_uint32 * words)()
    for b in bits:
        arr[b // 32] |= (1 << (b % 32))
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), ctypes.c_uint32(words), arr)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)


def timeit(fn, iters=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


print("CUs:", NCU)
cg, cm = mi355_dt.Context(), mi355_dt.Context()
rs = np.random.RandomState(0)
x = torch.randn(2304, 13, 13, 1024, device=cg.device)
w = (rs.randn(1, 1, 1024, 1024) * 0.03).astype(np.float32)
feat = torch.randn(4096, 26, 26, 512, device=cg.device)      # 5.7 GB read by the pooling kernel
det = torch.rand(4096, 4, device=cg.device)
tw = __import__("utility.synth", fromlist=["x"]).synth_tiny_weights(512)
cm.tiny_load(516, 512, tw["kernel"], tw["recurrent"], tw["bias"], tw["dense_kernel"], tw["dense_bias"])
gflop = 2.0 * 2304 * 169 * 1024 * 1024 / 1e9
gbytes = feat.numel() * 4 / 1e9


def gemm():
    cg.conv2d(x, w, None, leaky_slope=1.0, pool=0)


def mem():
    cm.tiny_features(feat, det, 516)


os.environ["DT_CONV_CFG"] = "3"
tg = timeit(gemm); tm = timeit(mem)
print("full GPU      : GEMM %.3f ms (%.1f TF)   pool %.3f ms (%.2f TB/s)   serial sum %.3f ms" % (tg, gflop / tg, tm, gbytes / tm, tg + tm))
# interleaved bit assignment: every (NCU/n_mem)-th CU id goes to the memory stream
step = NCU // n_mem
mem_bits = list(range(0, NCU, step))[:n_mem]
gem_bits = [b for b in range(NCU) if b not in mem_bits]
for name, mb, gb in (("strided", mem_bits, gem_bits), ("block", list(range(n_mem)), list(range(n_mem, NCU)))):
    sg, sm = masked_stream(gb), masked_stream(mb)
    with torch.cuda.stream(sg):
        tg2 = timeit(gemm)
    with torch.cuda.stream(sm):
        tm2 = timeit(mem)

    def both():
        with torch.cuda.stream(sg):
            gemm()
        with torch.cuda.stream(sm):
            mem(); mem()
    tb = timeit(both)
    print("%-8s masks: GEMM on %d CUs %.3f ms (%.1f TF)   pool on %d CUs %.3f ms (%.2f TB/s)   GEMM || 2 x pool: %.3f ms (serial would be %.3f)" % (
        name, len(gb), tg2, gflop / tg2, len(mb), tm2, gbytes / tm2, tb, tg + 2 * tm))

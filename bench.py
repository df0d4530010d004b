#!/usr/bin/env python3
"""bench.py -- frames/s of the detect-and-track hot path on MI355X.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python bench.py --gpus N --steps K --warmup W        # N > 1 without a launcher: re-executes itself under
                                                         # torch.distributed.run, one rank per GPU (RCCL)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Workload (default `--workload track`, BASELINE.json configs[2], the 1-GPU
configuration the metric "frames/sec detect+track @416x416" is quoted on):
per GPU `--clips` MOT17-shaped synthetic clips of T=30 frames, 416x416x3 uint8,
already resident in HBM; one STEP = one pass of the whole path over that batch:
  x/255 + YOLOv2 (23 conv, C=12)  ->  ConvLSTM2D(512,3x3) recurrence over T
  -> 1x1 conv -> decode_netout + NMS for every frame -> track-id association
  (+ for N>1 the cross-stream all-gather of the detection records).
`--workload detect` runs BASELINE.json configs[1] instead (YOLOv2 C=80 forward
+ decode, batch 8) -- reported as an extra line item, never as `value`.

Synthetic data: darknet-format random weights (utility/synth.py, seeds
1234/1235); the tracker's 1x1 head is calibrated once, outside the timed region,
so that ~32 boxes/frame survive (the "32 tracks" of configs[2]); frames are
low-frequency backgrounds with moving rectangles.

Prints ONE JSON line on rank 0 (contract in the task statement) including
  roofline     -- the fp32 MFMA implicit-GEMM conv kernel family: algorithmic
                  FLOP / HIP-event time of its launches inside the timed region,
                  against the 157.3 TFLOP/s fp32 matrix peak of MI355X
  cpu_baseline -- the CPU oracle ("port") timed on a bounded sample on this host.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

import object_tracking_amd  # noqa: F401
from models_detection.KerasYOLO import KerasYOLO
from models_tracking.MultiObjDetTracker import MultiObjDetTracker
from parallel import gather_detections, init_from_env, track_clips_frame_sharded
from utility import synth

PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CU x 2.4 GHz
PEAK_BF16_MFMA_TFLOPS = 2500.0    # MI355X_MICROARCH.md: v_mfma_f32_32x32x16_bf16 dense (16x the fp32 MFMA rate)
PEAK_HBM_GBS = 8000.0
GFLOP_TRACK_416 = 39.460          # 29.346 detect (C=12) + 10.099 ConvLSTM + 0.015 1x1
GFLOP_DETECT_416_C80 = 29.464


def make_frames(n_clips, T, H, W, device, seed0):
    """uint8 [n_clips,T,H,W,3] on the device.  Up to 8 distinct clips are rendered
    on the host (utility/synth.synth_clip); further clips are spatial rolls of them."""
    base = [torch.from_numpy(synth.synth_clip(T, H, W, 32, seed=seed0 + i)) for i in range(min(n_clips, 8))]
    out = torch.empty((n_clips, T, H, W, 3), dtype=torch.uint8, device=device)
    for i in range(n_clips):
        src = base[i % len(base)].to(device)
        if i >= len(base):
            src = torch.roll(src, shifts=(17 * (i // len(base)), 29 * (i // len(base))), dims=(1, 2))
        out[i] = src
    return out


def build_tracker(H, W, T, target_boxes, frames_for_calib):
    class Trk(MultiObjDetTracker):
        IMAGE_H, IMAGE_W = H, W
        GRID_H, GRID_W = H // 32, W // 32
        SEQUENCE_LENGTH = T
        LOAD_MODEL = False

    C = len(Trk.LABELS)
    blob = synth.synth_darknet_blob(C, seed=1234)
    tw = synth.synth_tracker_weights(C, seed=1235)
    S = 5 + C
    cls_ch = [b * S + 5 + c for b in range(5) for c in range(C)]
    obj_ch = [b * S + 4 for b in range(5)]
    tw["out_kernel"][..., cls_ch] *= 200.0          # peaky class softmax: score ~ objectness
    tw["out_bias"][obj_ch] = 0.0
    trk = Trk(detector_weights=blob, tracker_weights=tw)
    # calibration (untimed): shift the objectness bias so that ~target_boxes cells/frame have t_o > 0
    netout = trk.model.forward(frames_for_calib, want_det=False)
    to = netout[..., 4].reshape(netout.shape[0] * netout.shape[1], -1).float()
    k = max(1, min(to.shape[1] - 1, to.shape[1] - target_boxes))
    q = torch.kthvalue(to, k, dim=1).values.mean().item()
    tw["out_bias"][obj_ch] = -q
    trk.model.set_weights(tw)
    return trk, blob, tw


def cpu_baseline_track(blob, tw, H, W, n_frames):
    """The CPU statements of the path ("port" of the Keras graph, not Keras itself -- SURVEY.md section 0.1) on a
    bounded sample: one clip of n_frames frames through detector, ConvLSTM, 1x1, decode and association.  Two
    variants are timed -- oracle/oracle.c (C + OpenMP loop nest) and oracle/torch_cpu.py (ATen/oneDNN convolutions) --
    each warmed up and with the thread count that is fastest on this host (a short sweep on the detector: the GPU
    boxes' 256 logical cores are slower with every thread busy than with 32), and the FASTER one is the reported
    baseline.  Returns {variant: (frames/s, seconds, threads)}."""
    from oracle import oracle as orc
    from oracle import torch_cpu
    C = 12
    layers, _ = orc.parse_darknet_blob(blob, C)
    frames = orc.normalize_u8(synth.synth_clip(n_frames, H, W, 32, seed=999))
    ncpu = os.cpu_count() or 1
    cand = sorted({max(1, ncpu // d) for d in (1, 2, 4, 8, 16, 32, 64)}, reverse=True)
    sample = frames[:min(4, n_frames)]

    def sweep(set_threads, fwd):
        best = (None, 1e30)
        for nt in cand:
            set_threads(nt)
            fwd(sample[:1], layers)                      # primitive creation / page-in outside the timed pass
            t0 = time.perf_counter()
            fwd(sample, layers)
            dt = time.perf_counter() - t0
            if dt < best[1]:
                best = (nt, dt)
            if dt > 2.0 * best[1]:
                break
        set_threads(best[0])
        return best[0]

    out = {}
    default_torch = torch.get_num_threads()
    for name, fwd, det_fwd, set_threads in (
            ("oracle_c_openmp", orc.tracker_forward, orc.yolov2_forward, orc.set_threads),
            ("torch_cpu_onednn", torch_cpu.tracker_forward, torch_cpu.yolov2_forward, torch.set_num_threads)):
        nt = sweep(set_threads, det_fwd)
        t0 = time.perf_counter()
        trk, _ = fwd(frames, layers, tw)
        cap = trk.shape[1] * trk.shape[2] * 5
        rb = np.zeros((n_frames, cap, 8), dtype=np.float32)
        rc = np.zeros(n_frames, dtype=np.int32)
        for t in range(n_frames):
            rows, _ = orc.decode_netout(trk[t], 0.5, 0.45, MultiObjDetTracker.ANCHORS, C)
            rb[t, :len(rows)] = rows
            rc[t] = len(rows)
        orc.associate_clip(rb, rc, 0.3)
        dt = time.perf_counter() - t0
        out[name] = (n_frames / dt, dt, nt)
    torch.set_num_threads(default_torch)
    orc.set_threads(ncpu)
    return out


def detect_batch8_extra(device, H, W, seed0):
    """BASELINE.json configs[1] under the same clock: YOLOv2 C=80 forward + decode/NMS on 8 frames, plain launches
    and hipGraph replay, in its own context, after the main timed region."""
    C, B = 80, 8
    blob = synth.synth_darknet_blob(C, seed=1234)
    det = KerasYOLO({'LABELS': KerasYOLO.LABELS_COCO, 'BATCH_SIZE': B, 'IMAGE_H': H, 'IMAGE_W': W,
                     'GRID_H': H // 32, 'GRID_W': W // 32}, weights=blob)
    frames = make_frames(1, B, H, W, device, seed0=seed0)[0].contiguous()
    out = {"workload": "BASELINE.json configs[1]: YOLOv2 C=80 forward + decode/NMS, batch 8, %dx%d uint8" % (H, W)}
    for label, graphs in (("plain", False), ("graphs", True)):
        det.model.ctx.graph_enable(graphs)
        for _ in range(5):
            det.detect(frames)
        torch.cuda.synchronize()
        n = 50
        t0 = time.perf_counter()
        for _ in range(n):
            det.detect(frames)
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / n
        out[label] = {"ms_per_batch": ms, "frames_per_s": B / (ms * 1e-3),
                      "direct_form_tflops": B * GFLOP_DETECT_416_C80 * (H * W) / (416.0 * 416.0) / ms}
    det.model.ctx.graph_enable(False)
    return out


def _time_steps(step, warmup, steps):
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def track_extra(device, size, clips, T, boxes, steps=3, env=None, what=None):
    """BASELINE.json configs[4]'s single-GPU shard under the same clock: MultiObjDetTracker at size x size with
    ~`boxes` candidate boxes per frame, `clips` clips x T frames per step, its own context.  `env`: policy knobs the
    context is created under (read once in dt_create), e.g. {"DT_S3": "0"} = every GEMM on the fp32 MFMA instruction."""
    frames = make_frames(clips, T, size, size, device, seed0=7000)
    saved = {k: os.environ.get(k) for k in (env or {})}
    os.environ.update(env or {})
    try:
        trk, _, _ = build_tracker(size, size, T, boxes, frames)
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    cap = max(128, 2 * boxes)
    res = {}

    def step():
        res["r"] = trk.track_clips(frames, cap=cap)
    sec = _time_steps(step, 2, steps)
    return {"workload": what or "BASELINE.json configs[4] on one GPU: MultiObjDetTracker %dx%d, %d clips x %d frames per step, "
                                "head calibrated to %d candidate boxes per frame" % (size, size, clips, T, boxes),
            "ms_per_step": 1e3 * sec, "frames_per_s": clips * T / sec,
            "boxes_per_frame": float(res["r"]["counts"].float().mean().item())}


def tiny_extra(device, H, W, seqs, steps=3):
    """BASELINE.json configs[3] on one GPU under the same clock: TinyTracker over `seqs` sequences x 64 frames
    (YOLOv2 C=80 + act_13 global max-pool + decode / top box per frame, LSTM(512) + Dense(4) over T)."""
    from models_tracking.TinyTracker import TinyTracker
    C, T = 80, 64
    blob = synth.synth_darknet_blob(C, seed=1234)
    det = KerasYOLO({'LABELS': KerasYOLO.LABELS_COCO, 'BATCH_SIZE': 4, 'IMAGE_H': H, 'IMAGE_W': W,
                     'GRID_H': H // 32, 'GRID_W': W // 32}, weights=blob)
    ctx = det.model.ctx
    cfg = {"model_tracker": {"name": "TinyTracker", "lstm_units": 512, "sequence_length": T},
           "train": {"pool": "Global", "batch_size": 4}}
    tt = TinyTracker(cfg, feature_dims=(H // 16, W // 16, 512), weights=synth.synth_tiny_weights(512), ctx=ctx)
    frames = make_frames(seqs, T, H, W, device, seed0=9000)

    def step():
        rows, _ = tt.frame_rows(frames.reshape(seqs * T, H, W, 3), det)
        return ctx.tiny_sequence(rows.reshape(seqs, T, -1).contiguous())
    sec = _time_steps(step, 2, steps)
    return {"workload": "BASELINE.json configs[3] on one GPU: TinyTracker, %d sequences x %d frames, %dx%d" % (seqs, T, H, W),
            "ms_per_step": 1e3 * sec, "frames_per_s": seqs * T / sec}


def load_traffic(clips, T, size):
    """PMC-derived bytes per launch of the dominant GEMM kernel for this exact workload, measured
    with rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over
    tools/pmc_probe.py and committed under profiles/ (tools/make_traffic_json.py).
    bench.py cannot collect hardware counters itself; returns None if no
    committed measurement matches the workload."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*traffic*.json"))):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        w = d.get("workload", {})
        if (w.get("clips"), w.get("T"), w.get("size")) == (clips, T, size):
            best = (f, d)
    return best


def main():
    # the classes print a model summary like Keras does; stdout must carry ONLY the JSON line
    import contextlib
    with contextlib.redirect_stdout(sys.stderr):
        out = _run()
    if out is not None:
        print(json.dumps(out), flush=True)


def _run():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", choices=["track", "detect", "tiny"], default="track")
    ap.add_argument("--seqs", type=int, default=32, help="sequences per step (tiny)")
    ap.add_argument("--clips", type=int, default=48, help="clips per GPU per step (track)")
    ap.add_argument("--T", type=int, default=30)
    ap.add_argument("--size", type=int, default=416)
    ap.add_argument("--graphs", action="store_true", help="dt_graph_enable: hipGraph replay of the detector trunk / recurrences")
    ap.add_argument("--boxes", type=int, default=32, help="boxes/frame the synthetic tracker head is calibrated to (track)")
    ap.add_argument("--batch", type=int, default=8, help="frames per step (detect)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--h2d", action="store_true", help="also copy the frames from pinned host memory every step (PCIe-inclusive rate; never the headline value)")
    ap.add_argument("--cpu-frames", type=int, default=30)
    ap.add_argument("--shard", choices=["clip", "frame"], default="clip",
                    help="N>1 (track): clip = each rank owns --clips whole clips (weak scaling, no data-path collective); "
                         "frame = --clips clips in TOTAL, every rank runs the detector on its time steps of all of them, "
                         "per-frame rows all-gathered, recurrence on each clip's owner (strong scaling; configs[4])")
    ap.add_argument("--no-extra", action="store_true", help="skip the configs[1] extra block")
    ap.add_argument("--layer-report", default=None, help="write a per-layer table (HIP-event times) to this file")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # typed as a plain command: become the launcher -- one rank per GPU of this node under torch.distributed.run
        # (backend nccl = RCCL unless DT_DIST_BACKEND says otherwise); rank 0 of the child job prints the JSON line to
        # the stdout this process inherited
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        rc = subprocess.call(cmd, env=env, stdout=sys.__stdout__, stderr=sys.__stderr__)
        if rc != 0:
            sys.exit(rc)
        return None
    rank, world, local = init_from_env()
    assert world == args.gpus, "--gpus %d but WORLD_SIZE=%d: launch with torchrun --nproc-per-node %d, or run " \
                               "`python bench.py --gpus %d` without a launcher" % (args.gpus, world, args.gpus, args.gpus)
    assert torch.cuda.is_available(), "bench.py needs an MI355X; there is no CPU fallback"
    device = torch.device("cuda", torch.cuda.current_device())
    H = W = args.size

    if args.workload == "track":
        frame_shard = args.shard == "frame" and world > 1
        # frame-shard: ONE set of --clips clips for the whole job; a rank keeps only its own time steps of them resident
        # (sharded ingest: t mod N = rank) once the head is calibrated
        frames = make_frames(args.clips, args.T, H, W, device, seed0=42 + (0 if frame_shard else 100 * rank))
        trk, blob, tw = build_tracker(H, W, args.T, args.boxes, frames)
        if frame_shard:
            from parallel import frame_shard_times
            frames = frames[:, frame_shard_times(args.T, rank, world)].contiguous()
        xstats = {}
        ctx = trk.model.ctx
        frames_per_step = args.clips * args.T / (world if frame_shard else 1)
        gflop_per_frame = GFLOP_TRACK_416 * (H * W) / (416.0 * 416.0)

        host_frames = frames.cpu().pin_memory() if args.h2d else None

        def step():
            src = frames
            if host_frames is not None:
                frames.copy_(host_frames, non_blocking=True)     # same stream: serialised in front of the step
            xstats.clear()
            if frame_shard:
                return track_clips_frame_sharded(trk, src, cap=max(128, 2 * args.boxes), T=args.T, stats=xstats)
            res = trk.track_clips(src, cap=max(128, 2 * args.boxes))
            if world > 1:
                res = gather_detections(res, n_clips_max=args.clips, ctx=ctx, stats=xstats)
            return res
    elif args.workload == "tiny":
        # BASELINE.json configs[3]: TinyTracker (ROLO-style) over 64-frame sequences, FRAME-sharded:
        # every rank runs detector + pooling on its slice of the time axis of all sequences, the
        # small per-frame rows are all-gathered, the LSTM over T runs replicated.  Strong scaling.
        from models_tracking.TinyTracker import TinyTracker
        from parallel import gather_frame_rows
        C, T = 80, 64
        assert T % world == 0
        blob = synth.synth_darknet_blob(C, seed=1234)
        det = KerasYOLO({'LABELS': KerasYOLO.LABELS_COCO, 'BATCH_SIZE': 4, 'IMAGE_H': H, 'IMAGE_W': W,
                         'GRID_H': H // 32, 'GRID_W': W // 32}, weights=blob)
        ctx = det.model.ctx
        cfg = {"model_tracker": {"name": "TinyTracker", "lstm_units": 512, "sequence_length": T},
               "train": {"pool": "Global", "batch_size": 4}}
        tt = TinyTracker(cfg, feature_dims=(H // 16, W // 16, 512), weights=synth.synth_tiny_weights(512), ctx=ctx)
        t_loc = T // world
        frames = make_frames(args.seqs, t_loc, H, W, device, seed0=42 + 100 * rank)
        frames_per_step = args.seqs * t_loc
        gflop_per_frame = GFLOP_DETECT_416_C80 * (H * W) / (416.0 * 416.0)
        tw = None
        xstats = {}

        def step():
            rows, _ = tt.frame_rows(frames.reshape(args.seqs * t_loc, H, W, 3), det)
            rows = gather_frame_rows(rows.reshape(args.seqs, t_loc, -1).contiguous())
            return ctx.tiny_sequence(rows)
    else:
        C = 80
        blob = synth.synth_darknet_blob(C, seed=1234)
        det = KerasYOLO({'LABELS': KerasYOLO.LABELS_COCO, 'BATCH_SIZE': args.batch, 'IMAGE_H': H, 'IMAGE_W': W,
                         'GRID_H': H // 32, 'GRID_W': W // 32}, weights=blob)
        ctx = det.model.ctx
        frames = make_frames(1, args.batch, H, W, device, seed0=42 + 100 * rank)[0].contiguous()
        frames_per_step = args.batch
        gflop_per_frame = GFLOP_DETECT_416_C80 * (H * W) / (416.0 * 416.0)
        tw = None
        xstats = {}

        def step():
            return det.detect(frames)

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if args.graphs:
        ctx.graph_enable(True)        # captured during the warm-up steps (second call per shape), replayed afterwards
    res = None
    for _ in range(max(args.warmup, 3) if args.graphs else args.warmup):
        res = step()
    sync_all()
    ctx.profile_reset()
    if not args.graphs:
        ctx.profile_enable(True)      # HIP events around every launch, on the launch stream (graphs mode: none,
                                      # profiling bypasses the replayed graphs)
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    sync_all()
    elapsed = time.perf_counter() - t0
    ctx.profile_enable(False)
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=device if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    kern = {}
    for name in ("conv_gemm_s3", "conv_igemm", "conv_fused", "wino_input", "wino_output", "conv1_direct", "convlstm_gates", "decode_nms", "associate", "splitk_reduce", "pool",
                 "lstm_step", "misc"):
        p = ctx.profile_read(name)
        if p["launches"]:
            kern[name] = dict(launches=p["launches"], ms_per_step=p["ms"] / args.steps,
                              tflops=(p["flops"] / (p["ms"] * 1e-3) / 1e12) if p["ms"] > 0 else None,
                              gbs=(p["bytes"] / (p["ms"] * 1e-3) / 1e9) if p["ms"] > 0 else None)
    ig = ctx.profile_read("conv_igemm")
    achieved_f32 = ig["flops"] / (ig["ms"] * 1e-3) / 1e12 if ig["ms"] > 0 else 0.0
    # the F(6x6) layers' batched GEMMs: bf16 MFMAs on 3-term split fp32 operands (six partial products per multiply);
    # flops = EXECUTED bf16 FLOPs, so /6 is the fp32 multiply-add rate the layer sees
    s3 = ctx.profile_read("conv_gemm_s3")
    achieved_s3 = s3["flops"] / (s3["ms"] * 1e-3) / 1e12 if s3["ms"] > 0 else 0.0
    dominant_s3 = s3["ms"] >= ig["ms"]
    # direct-form FLOPs (SURVEY.md 8d figures) of the layers those launches computed; > executed where the
    # wide 3x3 layers run in Winograd form.  Time base: the MFMA kernel alone / with its transform kernels.
    direct_form = ctx.profile_read("conv_direct_form")["flops"]
    direct_form_bytes = ctx.profile_read("conv_direct_form")["bytes"]
    wino_in, wino_out = ctx.profile_read("wino_input"), ctx.profile_read("wino_output")
    wino_ms = wino_in["ms"] + wino_out["ms"]
    fused = ctx.profile_read("conv_fused")
    conv1 = ctx.profile_read("conv1_direct")
    direct_form_fused = ctx.profile_read("conv_direct_form_fused")["flops"]
    # whole conv path: every MFMA FLOP the conv kernels execute (batched / direct GEMMs, the fused Winograd kernels,
    # conv_1) over ALL the time the conv path takes (those kernels + the Winograd transform kernels)
    conv_path_ms = s3["ms"] + ig["ms"] + fused["ms"] + conv1["ms"] + wino_ms
    conv_path_flops = ig["flops"] + fused["flops"] + conv1["flops"]        # executed on the fp32 MFMA instructions
    # time the matrix pipe would need at its peaks for everything the conv path executes (bf16 and fp32 instructions
    # have different peaks) / the time the conv path takes, transforms included
    conv_path_pipe_frac = ((s3["flops"] / (PEAK_BF16_MFMA_TFLOPS * 1e12) + conv_path_flops / (PEAK_F32_MFMA_TFLOPS * 1e12)) /
                           (conv_path_ms * 1e-3)) if conv_path_ms > 0 else None
    # split the family's launches by arithmetic intensity (executed FLOP per algorithmic byte): below the ridge
    # of the chip (157.3 TFLOP/s over ~6.3 TB/s achievable = 25 FLOP/B; 40 used as the class boundary) a launch is
    # HBM-bound whatever the kernel does, and is priced against the HBM roof instead
    regimes = {"mfma": [0.0, 0.0, 0.0, []], "hbm": [0.0, 0.0, 0.0, []]}
    for name in ctx.profile_names():
        if not name.startswith("conv_igemm:"):
            continue
        p = ctx.profile_read(name)
        if p["ms"] <= 0 or p["bytes"] <= 0:
            continue
        r = regimes["mfma" if p["flops"] / p["bytes"] >= 40.0 else "hbm"]
        r[0] += p["flops"]; r[1] += p["bytes"]; r[2] += p["ms"]; r[3].append(name.split(":", 1)[1])
    s3_layers = {}
    for name in ctx.profile_names():
        if name.startswith("conv_gemm_s3:"):
            p = ctx.profile_read(name)
            if p["ms"] > 0:
                s3_layers[name.split(":", 1)[1]] = {"ms_per_step": p["ms"] / args.steps,
                                                    "executed_bf16_tflops": p["flops"] / (p["ms"] * 1e-3) / 1e12,
                                                    "fp32_equivalent_tflops": p["flops"] / 6.0 / (p["ms"] * 1e-3) / 1e12,
                                                    "algorithmic_GBps": p["bytes"] / (p["ms"] * 1e-3) / 1e9}
    boxes_per_frame = None
    if args.workload == "track" and res is not None and isinstance(res, dict):
        boxes_per_frame = float(res["counts"].float().mean().item())

    if rank == 0 and args.layer_report:
        with open(args.layer_report, "w") as f:
            f.write("# per-layer HIP-event times inside the timed region (%d steps); EXECUTED MFMA FLOP / algorithmic bytes\n" % args.steps)
            f.write("%-32s %8s %12s %10s %10s\n" % ("name", "launches", "ms/step", "TFLOP/s", "GB/s(alg)"))
            for name in sorted(ctx.profile_names()):
                p = ctx.profile_read(name)
                if p["ms"] <= 0:
                    continue
                f.write("%-32s %8d %12.4f %10.2f %10.1f\n" % (name, p["launches"], p["ms"] / args.steps,
                                                            p["flops"] / (p["ms"] * 1e-3) / 1e12,
                                                            p["bytes"] / (p["ms"] * 1e-3) / 1e9))
    if rank == 0:
        total_frames = frames_per_step * world * args.steps
        fps = total_frames / elapsed
        out = {
            "metric": {"track": "frames/sec detect+track @416x416", "detect": "frames/sec detect @416x416",
                       "tiny": "frames/sec detect + single-object LSTM track @416x416"}[args.workload],
            "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "strong" if (args.workload == "tiny" or (args.workload == "track" and args.shard == "frame" and world > 1)) else "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "dtype_detail": ("fp32 storage, fp32 accumulation everywhere; the F(6x6,3x3) layers' GEMMs form each fp32 product from six bf16 MFMA "
                             "partial products of 3-term split operands (x = x1 + x2 + x3, exact to 2^-25 |x|): error against float64 no "
                             "larger than the fp32 MFMA path's (DT_S3=0), which the parity tests assert") if s3["ms"] > 0 else
                            "fp32 storage, fp32 MFMA arithmetic, fp32 accumulation",
            "config": {"workload": ("BASELINE.json configs[2]: MultiObjDetTracker (YOLOv2 C=12 + ConvLSTM2D(512) + 1x1 "
                                    "+ decode/NMS + track ids), %d clips x %d frames per GPU per step, %dx%d uint8"
                                    % (args.clips, args.T, H, W)) if args.workload == "track" else
                       ("BASELINE.json configs[1]: YOLOv2 C=80 forward + decode/NMS, batch %d, %dx%d uint8"
                        % (args.batch, H, W)) if args.workload == "detect" else
                       ("BASELINE.json configs[3]: TinyTracker, %d sequences x 64 frames, frame-sharded x%d: YOLOv2 C=80 "
                        "+ act_13 global max-pool + decode/top box per frame, LSTM(512)+Dense(4) over T" % (args.seqs, world)),
                       "frames_per_step_per_gpu": frames_per_step,
                       "parallelism": ("frame-shard x%d (detector on t mod N, rows all-gathered, recurrence on the clip owner)" % world)
                       if (args.workload == "track" and args.shard == "frame" and world > 1) else "clip-shard x%d" % world,
                       "gflop_per_frame": gflop_per_frame, "boxes_per_frame": boxes_per_frame},
            "whole_path_tflops": fps * gflop_per_frame / 1e3, "h2d_included": bool(args.h2d),
            "ranks_seen": dist.get_world_size() if (world > 1 and dist.is_initialized()) else 1,
            "exchange_bytes_received_per_step_rank0": int(xstats.get("bytes_received", 0)),
            "roofline": {"bound": "mfma",
                         "kernel": ("wino_gemm_s3_kernel (v_mfma_f32_32x32x16_bf16 on 3-term split fp32 operands: six bf16 products per "
                                    "fp32 multiply, fp32 accumulate -- the F(6x6,3x3) layers' batched GEMMs)") if dominant_s3 else
                                   "conv_igemm_f32 (v_mfma_f32_32x32x2_f32 implicit GEMM)",
                         "achieved": achieved_s3 if dominant_s3 else achieved_f32,
                         "peak": PEAK_BF16_MFMA_TFLOPS if dominant_s3 else PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": (achieved_s3 / PEAK_BF16_MFMA_TFLOPS) if dominant_s3 else (achieved_f32 / PEAK_F32_MFMA_TFLOPS),
                         "dominant_kernel_ms_per_step": (s3["ms"] if dominant_s3 else ig["ms"]) / max(1, args.steps),
                         "split_bf16_gemm": None if s3["ms"] <= 0 else {
                             "executed_bf16_tflops": achieved_s3, "peak_bf16_tflops": PEAK_BF16_MFMA_TFLOPS,
                             "frac": achieved_s3 / PEAK_BF16_MFMA_TFLOPS,
                             "fp32_equivalent_tflops": achieved_s3 / 6.0,
                             "fp32_equivalent_over_fp32_mfma_peak": achieved_s3 / 6.0 / PEAK_F32_MFMA_TFLOPS,
                             "ms_per_step": s3["ms"] / max(1, args.steps), "launches_per_step": s3["launches"] / max(1, args.steps),
                             "avg_launch_ms": s3["ms"] / max(1, s3["launches"]),
                             "executed_gflop_per_launch": s3["flops"] / max(1, s3["launches"]) / 1e9,
                             "algorithmic_bytes_per_launch": s3["bytes"] / max(1, s3["launches"]),
                             "layers": s3_layers},
                         "fp32_mfma_kernel": {"kernel": "conv_igemm_f32 (v_mfma_f32_32x32x2_f32)", "achieved": achieved_f32,
                                              "peak": PEAK_F32_MFMA_TFLOPS, "frac": achieved_f32 / PEAK_F32_MFMA_TFLOPS,
                                              "ms_per_step": ig["ms"] / max(1, args.steps)},
                         "frac_executed": achieved_f32 / PEAK_F32_MFMA_TFLOPS,
                         # the transform kernels exist only because of the Winograd form: charge them to the conv path
                         "frac_incl_transforms": (ig["flops"] / ((ig["ms"] + wino_ms) * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS)
                         if ig["ms"] > 0 else None,
                         "frac_whole_conv_path": conv_path_pipe_frac,
                         "whole_conv_path": {"executed_fp32_mfma_tflop_per_step": conv_path_flops / max(1, args.steps) / 1e12,
                                             "executed_bf16_mfma_tflop_per_step": s3["flops"] / max(1, args.steps) / 1e12,
                                             "fp32_equivalent_tflops": ((conv_path_flops + s3["flops"] / 6.0) / (conv_path_ms * 1e-3) / 1e12)
                                             if conv_path_ms > 0 else None,
                                             "ms_per_step": conv_path_ms / max(1, args.steps),
                                             "direct_form_tflop_per_step": (direct_form + direct_form_fused + conv1["flops"]) / max(1, args.steps) / 1e12},
                         "transform_ms_per_step": wino_ms / max(1, args.steps),
                         # conv family as a whole (MFMA GEMMs + transforms + fused conv_2): direct-form bytes the
                         # reference's layers need (in + W + out, fp32) vs the bytes this implementation's kernels
                         # move by their own algorithmic count (V + U + M' for the GEMMs, the transforms' reads+writes)
                         "direct_form_bytes_per_step": direct_form_bytes / max(1, args.steps),
                         "implementation_bytes_per_step": (s3["bytes"] + ig["bytes"] + wino_in["bytes"] + wino_out["bytes"] + fused["bytes"]) / max(1, args.steps),
                         "traffic": None, "traffic_per_step": None,
                         "launches_per_step": ig["launches"] / max(1, args.steps),
                         "avg_launch_ms": ig["ms"] / max(1, ig["launches"]),
                         "executed_gflop_per_launch": ig["flops"] / max(1, ig["launches"]) / 1e9,
                         "algorithmic_gflop_per_launch": direct_form / max(1, ig["launches"]) / 1e9,
                         "algorithmic_bytes_per_launch": ig["bytes"] / max(1, ig["launches"]),
                         "achieved_algorithmic": (direct_form / (ig["ms"] * 1e-3) / 1e12) if ig["ms"] > 0 else None,
                         "achieved_algorithmic_incl_transforms":
                             (direct_form / ((ig["ms"] + wino_ms) * 1e-3) / 1e12) if ig["ms"] > 0 else None,
                         "mfma_bound_launches": None if regimes["mfma"][2] <= 0 else {
                             "layers": sorted(regimes["mfma"][3]), "ms_per_step": regimes["mfma"][2] / args.steps,
                             "achieved": regimes["mfma"][0] / (regimes["mfma"][2] * 1e-3) / 1e12,
                             "frac": regimes["mfma"][0] / (regimes["mfma"][2] * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS},
                         "hbm_bound_launches": None if regimes["hbm"][2] <= 0 else {
                             "layers": sorted(regimes["hbm"][3]), "ms_per_step": regimes["hbm"][2] / args.steps,
                             "achieved_GBps": regimes["hbm"][1] / (regimes["hbm"][2] * 1e-3) / 1e9,
                             "frac_of_8TBps": regimes["hbm"][1] / (regimes["hbm"][2] * 1e-3) / 8e12,
                             "executed_tflops": regimes["hbm"][0] / (regimes["hbm"][2] * 1e-3) / 1e12},
                         "note": "achieved/frac = MFMA FLOPs the DOMINANT kernel executes / its HIP-event time, against the dense "
                                 "peak of the instruction it issues (pipe utilisation, <= 1). With the default policy the dominant "
                                 "kernel is wino_gemm_s3_kernel: the F(6x6,3x3) layers' GEMMs carry every fp32 operand as three bf16 "
                                 "terms and form each product from six bf16 MFMA partial products with fp32 accumulation -- fp32 "
                                 "accuracy (tests/test_gpu_parity.py::test_split_bf16_gemm_error_against_float64) at 16/6 of the "
                                 "fp32 MFMA rate; split_bf16_gemm.fp32_equivalent_tflops = executed / 6 is what the layer sees "
                                 "(DT_S3=0 runs the same GEMMs on v_mfma_f32_32x32x2_f32: fp32_mfma_kernel). "
                                 "launches_per_step .. achieved_algorithmic and the mfma_/hbm_bound split below describe the fp32 "
                                 "kernel family (conv_igemm_f32: 1x1 layers, short-K Winograd GEMMs, recurrent step). "
                                 "frac_whole_conv_path = time the matrix pipe needs at its peaks for everything the conv path "
                                 "executes (bf16 FLOPs / 2500 + fp32 FLOPs / 157.3) / the conv path's time incl. transforms. "
                                 "The 3x3 layers from conv_3 up and both ConvLSTM convolutions run in Winograd "
                                 "form (F(6x6,3x3): 64 batched GEMMs through the same kernel; F(4x4,3x3) for the recurrent "
                                 "step), which executes up to 5x fewer FLOPs than the direct form SURVEY.md 8d counts; "
                                 "achieved_algorithmic = direct-form FLOPs of the layers THESE launches computed (layers run "
                                 "by the fused Winograd kernels or conv_1 are not counted) / the same time, and exceeds "
                                 "the peak; ..._incl_transforms adds the HBM-bound transform kernels to the time; "
                                 "frac_whole_conv_path = every executed MFMA FLOP of the conv path (GEMMs + fused kernels + "
                                 "conv_1) / (their time + the transforms' time) / peak. "
                                 "mfma_bound_launches / hbm_bound_launches split the family by arithmetic intensity "
                                 "(>= / < 40 executed FLOP per algorithmic byte): the short-K launches (K = 64/128 Winograd "
                                 "GEMMs, early 1x1 layers) sit under the HBM roof, not the MFMA one."},
            "kernels": kern,
        }
        tr = load_traffic(args.clips, args.T, args.size) if args.workload == "track" else None
        if tr is not None:
            out["roofline"]["traffic"] = tr[1]["traffic_bytes_per_launch"]
            out["roofline"]["traffic_per_step"] = tr[1].get("traffic_bytes_per_step")
            out["roofline"]["traffic_unit"] = "bytes/launch (FETCH_SIZE x in-run calibration + WRITE_SIZE; beyond-L2, incl. Infinity Cache hits)"
            out["roofline"]["traffic_source"] = os.path.relpath(tr[0], ROOT)
        if world == 1 and not args.no_extra and args.workload == "track":
            import gc
            out["extra"] = {}
            for key, fn in (("detect_batch8", lambda: detect_batch8_extra(device, H, W, seed0=4242)),
                            ("track_608_128boxes", lambda: track_extra(device, 608, 24, args.T, 128)),
                            ("tiny_T64", lambda: tiny_extra(device, H, W, 32)),
                            # the headline workload with the split-bf16 GEMMs switched off (DT_S3=0: v_mfma_f32_32x32x2_f32
                            # everywhere), same run, same clock: what the bf16-pipe arithmetic buys
                            ("track_fp32_mfma_only", lambda: track_extra(
                                device, args.size, args.clips, args.T, args.boxes, env={"DT_S3": "0"},
                                what="the headline workload (BASELINE.json configs[2], %d clips x %d frames) with DT_S3=0: every GEMM on "
                                     "the fp32 MFMA instruction" % (args.clips, args.T)))):
                out["extra"][key] = fn()
                gc.collect()                      # each extra owns a context with its own workspaces: release them
                torch.cuda.empty_cache()
        if world == 1 and not args.no_cpu_baseline and args.workload == "track":
            from oracle import oracle as orc
            orc.lib()
            variants = cpu_baseline_track(blob, tw, H, W, args.cpu_frames)
            best = max(variants, key=lambda k: variants[k][0])
            out["cpu_baseline"] = {"value": variants[best][0], "unit": "frames/s", "cores": variants[best][2], "kind": "port",
                                   "variant": best, "host_logical_cores": os.cpu_count(),
                                   "variants": {k: {"frames_per_s": v[0], "seconds": v[1], "threads": v[2]} for k, v in variants.items()},
                                   "sample": "CPU restatement of the graph (NOT Keras/TF, which cannot run here), the faster of "
                                             "oracle/oracle.c (C + OpenMP) and oracle/torch_cpu.py (ATen/oneDNN): 1 clip x %d "
                                             "frames %dx%d through detector+ConvLSTM+1x1+decode+association after a warm-up and "
                                             "a thread-count sweep (cores = threads of the reported variant), %.1f s"
                                             % (args.cpu_frames, H, W, variants[best][1])}
    else:
        out = None
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return out


if __name__ == "__main__":
    main()

"""GPU suite, part 4 (-m gpu): the multi-process path on the ONE GPU a test box has.

  * RCCL itself: backend "nccl" initialises and runs the collectives parallel.py uses (all_gather_into_tensor,
    all_reduce) at world size 1; at world size 2 with both ranks on cuda:0 if RCCL accepts two ranks per device
    (it may refuse with "Duplicate GPU detected" -- then the test is skipped with that message: an 8-GPU node is the
    only place two RCCL ranks can really meet).
  * frame-shard of MultiObjDetTracker (SURVEY.md 8e row 3 / BASELINE.json configs[4]): two ranks (gloo, both on
    cuda:0) split the time axis of every clip for the detector, all-gather the rows, run the recurrence on the clip
    owner and gather detections -- the global table must carry the SAME track ids as the single-process run.
  * bench.py --shard frame under torch.distributed.run with two ranks.
"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_ranks(script, world, env_extra, port, timeout=600):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world),
               HSA_ENABLE_IPC_MODE_LEGACY="0", **env_extra)
    procs = [subprocess.Popen([sys.executable, str(script), ROOT], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=timeout)[0])
        except subprocess.TimeoutExpired:
            p.kill()
            outs.append(p.communicate()[0] + "\n<timeout>")
    return procs, outs


_NCCL = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
import object_tracking_amd
from parallel import _all_gather_cat, gather_detections, global_track_ids, shard_range
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=rank, world_size=world)
assert dist.get_backend() == "nccl"
dev = torch.device("cuda", 0)
x = torch.arange(6, dtype=torch.int32, device=dev).reshape(2, 3) + 100 * rank
g = _all_gather_cat(x)                                       # all_gather_into_tensor on the device
want = torch.cat([torch.arange(6, dtype=torch.int32).reshape(2, 3) + 100 * r for r in range(world)]).to(dev)
ok = torch.equal(g, want)
t = torch.tensor([float(rank + 1)], device=dev); dist.all_reduce(t, op=dist.ReduceOp.MAX); ok &= float(t) == world
# the packed detection gather, uneven shards
torch.manual_seed(0)
N, T, cap = 5, 3, 4
boxes = torch.rand(N, T, cap, 8, device=dev); counts = torch.randint(0, cap + 1, (N, T), dtype=torch.int32, device=dev)
ids = torch.randint(-1, 3, (N, T, cap), dtype=torch.int32, device=dev); nids = torch.randint(1, 4, (N,), dtype=torch.int32, device=dev)
a, b = shard_range(N, rank, world)
out = gather_detections(dict(boxes=boxes[a:b], counts=counts[a:b], ids=ids[a:b], nids=nids[a:b]), n_clips_max="max")
ok &= torch.equal(out["boxes"], boxes) and torch.equal(out["ids"], ids) and torch.equal(out["gids"], global_track_ids(ids, nids))
torch.cuda.synchronize()
print("RANK", rank, "OK" if ok else "MISMATCH", flush=True)
dist.barrier(); dist.destroy_process_group()
sys.exit(0 if ok else 1)
'''


def test_rccl_backend_world1(tmp_path):
    script = tmp_path / "nccl1.py"
    script.write_text(_NCCL)
    procs, outs = _run_ranks(script, 1, {}, 29751)
    assert procs[0].returncode == 0 and "RANK 0 OK" in outs[0], outs[0][-3000:]


def test_rccl_backend_world2_on_one_device(tmp_path):
    script = tmp_path / "nccl2.py"
    script.write_text(_NCCL)
    procs, outs = _run_ranks(script, 2, {}, 29753, timeout=300)
    if all(p.returncode == 0 for p in procs):
        assert "RANK 0 OK" in outs[0] and "RANK 1 OK" in outs[1]
        return
    text = "\n".join(outs)
    refused = any(k in text for k in ("Duplicate GPU", "duplicate GPU", "invalid usage", "ncclInvalidUsage", "<timeout>"))
    assert refused, text[-4000:]
    pytest.skip("RCCL does not run two ranks on one device here: " + [l for l in text.splitlines() if "uplicate" in l or "nvalid" in l or "timeout" in l][:1][0][:200])


_FRAMESHARD = r'''
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
import object_tracking_amd
from parallel import gather_detections, global_track_ids, track_clips_frame_sharded, init_from_env
from utility import synth
from models_tracking.MultiObjDetTracker import MultiObjDetTracker
rank, world, _ = init_from_env()
H = W = 96; T = 5; N = 3; C = 12                      # odd T: uneven time shards; 3 clips on 2 owners
class Trk(MultiObjDetTracker):
    IMAGE_H, IMAGE_W = H, W
    GRID_H, GRID_W = 3, 3
    SEQUENCE_LENGTH = T
    LOAD_MODEL = False
    OBJ_THRESHOLD = 0.3
tw = synth.synth_tracker_weights(C); tw["out_kernel"] = tw["out_kernel"] * 40.0; tw["out_bias"][4::17] = 1.5
trk = Trk(detector_weights=synth.synth_darknet_blob(C), tracker_weights=tw)
frames = np.stack([synth.synth_clip(T, H, W, 2, seed=500 + i) for i in range(N)])
out = track_clips_frame_sharded(trk, frames)
ref = trk.track_clips(frames)
# the two halves run the layers at other batch sizes than the one-process forward (split-K / Winograd choice): values
# agree to rounding, everything discrete -- counts, cells, labels, track ids -- is exact
flags = dict(counts=torch.equal(out["counts"], ref["counts"]), labels=torch.equal(out["boxes"][..., 5], ref["boxes"][..., 5]),
             cells=torch.equal(out["boxes"][..., 7], ref["boxes"][..., 7]),
             boxes=bool(torch.allclose(out["boxes"], ref["boxes"], rtol=1e-3, atol=1e-4)),      # 7-8-frame detector batches take other kernels than the 15-frame one
             ids=torch.equal(out["ids"], ref["ids"]), gids=torch.equal(out["gids"], global_track_ids(ref["ids"], ref["nids"])),
             nonempty=int(ref["counts"].sum()) > 0)
flags["box_err"] = float((out["boxes"] - ref["boxes"]).abs().max())
# and the halves compose to the whole: detect + recurrent == forward, bit for bit at equal batch
ctx = trk.model.ctx
d = trk.detector.model.to_device(frames)
z = ctx.track_detect(d.reshape(N * T, H, W, 3))
halves = ctx.track_recurrent(z.reshape(N, T, 3, 3, -1))
whole = ctx.track_forward(d, want_det=False)
# (caller-owned z rows take the two-step input projection, dt_track_forward the one with conv_23 folded into its weights when its Winograd form
#  runs: equal bit for bit at this size, where both take the direct form; the merged form is compared in test_tracker_merged_input_projection)
flags["halves_z"] = torch.equal(halves, whole)
flags["halves_z_err"] = float((halves - whole).abs().max())
flags["deterministic"] = torch.equal(whole, ctx.track_forward(d, want_det=False))
# ... and so does the split one step later (input projection on the detector's side), which is what the frame-shard ships by default
xp = ctx.track_detect_xproj(d.reshape(N * T, H, W, 3))
flags["xp_width"] = xp.shape[-1] == ctx.track_xproj_width() == 4 * 512
hx = ctx.track_recurrent_xproj(xp.reshape(N, T, 3, 3, -1))
flags["halves_xproj"] = torch.equal(hx, whole)
flags["halves_xproj_err"] = float((hx - whole).abs().max())
# both row kinds give the single-process ids
out_z = track_clips_frame_sharded(trk, frames, rows="z")
flags["rows_z"] = torch.equal(out_z["ids"], ref["ids"]) and torch.equal(out_z["counts"], ref["counts"])
ok = all(v for k, v in flags.items() if not k.endswith("err"))
print("RANK", rank, "OK" if ok else "MISMATCH", int(ref["counts"].sum()), flags, flush=True)
dist.barrier(); dist.destroy_process_group()
sys.exit(0 if ok else 1)
'''


def test_frame_sharded_tracker_two_ranks_one_gpu(tmp_path):
    script = tmp_path / "fs.py"
    script.write_text(_FRAMESHARD)
    procs, outs = _run_ranks(script, 2, {"DT_ONE_DEVICE": "1", "DT_DIST_BACKEND": "gloo"}, 29755)
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and ("RANK %d OK" % r) in o, o[-3000:]


@pytest.mark.parametrize("shard", ["clip", "frame"])
def test_bench_two_ranks_on_one_gpu(shard):
    """bench.py under torch.distributed.run with 2 ranks (gloo, one device): the N>1 code of both shard modes runs and
    rank 0 prints one JSON line with the contract's keys."""
    env = dict(os.environ, DT_ONE_DEVICE="1", DT_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29757" if shard == "clip" else "29759", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2",
           "--warmup", "1", "--clips", "2", "--T", "4", "--size", "96", "--boxes", "4", "--shard", shard, "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, (r.stdout + r.stderr)[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["value"] > 0 and out["scaling"] == ("strong" if shard == "frame" else "weak")
    frames_total = 2 * 4 * (1 if shard == "frame" else 2)
    assert abs(out["value"] * out["ms_per_step"] * 1e-3 - frames_total) < 1e-6 * frames_total


def test_bench_self_launches_its_ranks():
    """`python bench.py --gpus 2 ...` typed as a plain command (no torchrun on the command line, no WORLD_SIZE in the
    environment): bench.py re-executes itself under torch.distributed.run with one rank per GPU and rank 0 prints the
    one JSON line.  Here both ranks share the box's single GPU over gloo (DT_ONE_DEVICE / DT_DIST_BACKEND); on an
    8-GPU node the same command uses RCCL."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(DT_ONE_DEVICE="1", DT_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--clips", "2",
           "--T", "4", "--size", "96", "--boxes", "4", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, (r.stdout + r.stderr)[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks_seen"] == 2 and out["value"] > 0 and out["scaling"] == "weak"
    assert out["exchange_bytes_received_per_step_rank0"] > 0
    assert abs(out["value"] * out["ms_per_step"] * 1e-3 - 2 * 2 * 4) < 1e-5


@pytest.mark.parametrize("mode", ["clip", "frame", "tiny"])
def test_eight_ranks_dry_run_of_the_scale_command(mode):
    """The process shape the driver's SCALE run takes -- `python bench.py --gpus 8 ...` typed as a plain command -- has
    executed once: eight ranks under torch.distributed.run (here all on the box's one GPU over gloo, tiny sizes), both
    shard modes of the tracker and the frame-sharded TinyTracker (T = 64 = 8 x 8).  Frame-shard with 3 clips on 8 ranks:
    five ranks own no clip and idle in the owner stage -- the per-rank stage times in the line say so."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(DT_ONE_DEVICE="1", DT_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--size", "96", "--no-cpu-baseline"]
    if mode == "tiny":
        cmd += ["--workload", "tiny", "--seqs", "2"]
    else:
        cmd += ["--clips", "3" if mode == "frame" else "1", "--T", "8", "--boxes", "4", "--shard", mode]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert r.returncode == 0, (r.stdout + r.stderr)[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["ranks_seen"] == 8 and out["value"] > 0
    assert out["scaling"] == ("weak" if mode == "clip" else "strong")
    frames_total = {"clip": 8 * 1 * 8, "frame": 3 * 8, "tiny": 2 * 64}[mode]
    assert abs(out["value"] * out["ms_per_step"] * 1e-3 - frames_total) < 1e-5 * frames_total
    if mode == "frame":
        st = out["frame_shard_stage_ms_per_rank"]
        assert len(st) == 8 and sorted(s["rank"] for s in st) == list(range(8))
        assert sum(s["clips_owned"] for s in st) == 3 and sum(1 for s in st if s["clips_owned"] == 0) == 5
        assert all(s["sharded_stage_ms"] > 0 and s["owner_stage_ms"] >= 0 for s in st)
        assert out["exchange_bytes_received_per_step_rank0"] > 0


def test_native_pack_unpack_matches_host_logic(ctx):
    """dt_pack_detections / dt_unpack_detections (the exchange a C-ABI caller has) against parallel.py's torch
    statements of the same layout: bit-identical rows, tables, global ids; padding rows and empty ranks handled."""
    import torch
    from parallel import _pack_rows, _unpack_rows, global_track_ids
    dev = ctx.device
    torch.manual_seed(5)
    T, cap = 7, 9
    shards = [3, 0, 2, 4]                     # clips per "rank"; rows padded to 4
    n_pad = max(shards)
    parts, tables = [], []
    for n in shards:
        boxes = torch.rand(n, T, cap, 8, device=dev)
        counts = torch.randint(0, cap + 1, (n, T), dtype=torch.int32, device=dev)
        ids = torch.randint(-1, 6, (n, T, cap), dtype=torch.int32, device=dev)
        nids = torch.randint(0, 7, (n,), dtype=torch.int32, device=dev)
        rows = ctx.pack_detections(boxes, counts, ids, nids, n_pad)
        assert rows.shape == (n_pad, T * cap * 8 + T * cap + T + 2)
        assert torch.equal(rows, _pack_rows(boxes, counts, ids, nids, n_pad))
        parts.append(rows)
        tables.append((boxes, counts, ids, nids))
    allrows = torch.cat(parts).contiguous()
    b, c, i, n, g = ctx.unpack_detections(allrows, T, cap)
    wb, wc, wi, wn = _unpack_rows(allrows, T, cap)
    assert torch.equal(b, wb) and torch.equal(c, wc) and torch.equal(i, wi) and torch.equal(n, wn)
    assert torch.equal(b, torch.cat([t[0] for t in tables])) and b.shape[0] == sum(shards)
    assert torch.equal(g, global_track_ids(wi, wn))
    with pytest.raises(AssertionError):
        ctx.pack_detections(*tables[3], 2)           # 4 clips into 2 rows: refused with a clear message


_FRAMESHARD4 = r'''
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
import object_tracking_amd
from parallel import global_track_ids, track_clips_frame_sharded, init_from_env, frame_shard_times, gather_frame_rows
from models_tracking.MultiObjDetTracker import MultiObjDetTracker
from utility import synth
rank, world, _ = init_from_env()
H = W = 96; T = 6; N = 3; C = 12
class Trk(MultiObjDetTracker):
    IMAGE_H, IMAGE_W = H, W
    GRID_H, GRID_W = 3, 3
    SEQUENCE_LENGTH = T
    LOAD_MODEL = False
    OBJ_THRESHOLD = 0.3
tw = synth.synth_tracker_weights(C); tw["out_kernel"] = tw["out_kernel"] * 40.0; tw["out_bias"][4::17] = 1.5
trk = Trk(detector_weights=synth.synth_darknet_blob(C), tracker_weights=tw)
frames = np.stack([synth.synth_clip(T, H, W, 2, seed=500 + i) for i in range(N)])
ref = trk.track_clips(frames)
mine = frame_shard_times(T, rank, world)
st = {}
out = track_clips_frame_sharded(trk, np.ascontiguousarray(frames[:, mine]), T=T, chunks=2, stats=st)      # sharded ingest
flags = dict(counts=torch.equal(out["counts"], ref["counts"]), labels=torch.equal(out["boxes"][..., 5], ref["boxes"][..., 5]),
             cells=torch.equal(out["boxes"][..., 7], ref["boxes"][..., 7]),
             boxes=bool(torch.allclose(out["boxes"], ref["boxes"], rtol=1e-3, atol=1e-4)),   # 3-frame detector batches take other kernels than 18-frame ones
             ids=torch.equal(out["ids"], ref["ids"]), gids=torch.equal(out["gids"], global_track_ids(ref["ids"], ref["nids"])),
             nonempty=int(ref["counts"].sum()) > 0, bytes=st["bytes_received"] > 0)
ok = all(flags.values())
# TinyTracker frame-shard (BASELINE configs[3]): rows of this rank's slice of the time axis, all-gathered, LSTM replicated
from models_detection.KerasYOLO import KerasYOLO
from models_tracking.TinyTracker import TinyTracker
Tt, S = 64, 2
det = KerasYOLO({'LABELS': [str(i) for i in range(C)], 'BATCH_SIZE': 4, 'IMAGE_H': H, 'IMAGE_W': W, 'GRID_H': 3, 'GRID_W': 3},
                weights=synth.synth_darknet_blob(C, head_std=0.01))
ctx = det.model.ctx
cfg = {"model_tracker": {"name": "TinyTracker", "lstm_units": 512, "sequence_length": Tt}, "train": {"pool": "Global", "batch_size": 4}}
tt = TinyTracker(cfg, feature_dims=(H // 16, W // 16, 512), weights=synth.synth_tiny_weights(512), ctx=ctx)
fr = torch.from_numpy(np.stack([synth.synth_clip(Tt, H, W, 2, seed=900 + i) for i in range(S)])).to(ctx.device)
rows_all, _ = tt.frame_rows(fr.reshape(S * Tt, H, W, 3), det)
want = ctx.tiny_sequence(rows_all.reshape(S, Tt, -1).contiguous())
tl = Tt // world
loc = fr[:, rank * tl:(rank + 1) * tl].contiguous()
rows, _ = tt.frame_rows(loc.reshape(S * tl, H, W, 3), det)
got = ctx.tiny_sequence(gather_frame_rows(rows.reshape(S, tl, -1).contiguous()))
flags["tiny"] = bool(torch.allclose(got, want, rtol=0, atol=2e-5)) and got.shape == (S, Tt, 4)
flags["tiny_err"] = float((got - want).abs().max())
ok &= flags["tiny"]
print("RANK", rank, "OK" if ok else "MISMATCH", int(ref["counts"].sum()), flags, flush=True)
dist.barrier(); dist.destroy_process_group()
sys.exit(0 if ok else 1)
'''


def test_frame_shard_four_ranks_one_gpu(tmp_path):
    """4 ranks (gloo, one device): MultiObjDetTracker frame-shard with sharded ingest, rows to round-robin owners in two
    chunks -- ids identical to one process (3 clips on 4 ranks: one rank owns no clip); and the TinyTracker T=64
    frame-shard (gather_frame_rows) equal to the one-process sequence."""
    script = tmp_path / "fs4.py"
    script.write_text(_FRAMESHARD4)
    procs, outs = _run_ranks(script, 4, {"DT_ONE_DEVICE": "1", "DT_DIST_BACKEND": "gloo"}, 29771, timeout=900)
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and ("RANK %d OK" % r) in o, o[-3000:]


def test_graph_replay_with_caller_owned_rows():
    """dt_graph_enable + dt_track_recurrent with a DIFFERENT caller-owned z buffer on every call (what the
    frame-shard does): a captured graph must never replay a stale row pointer -- every call equals the ungraphed
    result for ITS rows, and dt_track_forward (library-owned z) still replays beside it."""
    import numpy as np
    import torch
    from models_tracking.MultiObjDetTracker import MultiObjDetTracker
    from utility import synth
    H = W = 96; T = 5; N = 2; C = 12

    class Trk(MultiObjDetTracker):
        IMAGE_H, IMAGE_W = H, W
        GRID_H, GRID_W = 3, 3
        SEQUENCE_LENGTH = T
        LOAD_MODEL = False
    trk = Trk(detector_weights=synth.synth_darknet_blob(C), tracker_weights=synth.synth_tracker_weights(C))
    ctx = trk.model.ctx
    dev = ctx.device
    clips = [torch.from_numpy(np.stack([synth.synth_clip(T, H, W, 2, seed=300 + 10 * k + i) for i in range(N)])).to(dev) for k in range(4)]
    zs = [ctx.track_detect(c.reshape(N * T, H, W, 3)).reshape(N, T, 3, 3, -1).clone() for c in clips]
    want = [ctx.track_recurrent(z) for z in zs]
    want_fwd = [ctx.track_forward(c, want_det=False) for c in clips]
    assert not torch.equal(want[0], want[1])
    ctx.graph_enable(True)
    try:
        for rep in range(2):
            for k, z in enumerate(zs):
                zz = z.clone()                      # a fresh buffer each call: a stale captured pointer would show
                got = ctx.track_recurrent(zz)
                assert torch.equal(got, want[k]), "graph replay used another call's rows (call %d, pass %d)" % (k, rep)
                del zz
            for k, c in enumerate(clips):
                assert torch.equal(ctx.track_forward(c, want_det=False), want_fwd[k])
        assert ctx.profile_read("graph_replay")["launches"] > 0
    finally:
        ctx.graph_enable(False)


_WORLDS = r'''
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
import object_tracking_amd
import bench
from parallel import track_clips_frame_sharded, init_from_env
rank, world, _ = init_from_env()
size, n_clips, T, cap = 416, 9, 30, 128
dev = torch.device("cuda", torch.cuda.current_device())
frames = bench.make_frames(n_clips, T, size, size, dev, seed0=42)
trk, blob, tw = bench.build_tracker(size, size, T, 32, frames)
trk.OBJ_THRESHOLD, trk.NMS_THRESHOLD, trk.ASSOC_THRESHOLD = 0.5, 0.45, 0.3      # the reference's defaults (KerasYOLO.py:43-44)
out = {}
for tag, det, rows in (("default", False, None), ("pinned", True, None), ("pinnedz", True, "z")):
    r = track_clips_frame_sharded(trk, frames, cap=cap, deterministic=det, rows=rows)
    for k in ("boxes", "counts", "ids", "gids"):
        out[tag + "_" + k] = r[k].cpu().numpy()
if rank == 0:
    np.savez(os.environ["DT_WORLDS_OUT"], **out)
print("RANK", rank, "OK", flush=True)
if world > 1:
    dist.barrier(); dist.destroy_process_group()
'''


def test_frame_shard_world_sizes_agree(tmp_path):
    """track_clips_frame_sharded at world 1 / 2 / 4 (gloo, all ranks on this box's GPU) on the SAME nine 30-frame 416x416 clips at the
    reference's default thresholds.  A rank's detector batch shrinks with the world size, and the library's kernel selection looks at
    the batch, so under the DEFAULT policy the same frame is another rounding of the network: how many frames / boxes / ids then differ
    between world sizes is REPORTED (gpurun_out/parity_r06_world_sizes.json).  With deterministic=True (DT_PIN=1: selection independent
    of the batch) boxes, counts and global ids must be BIT-IDENTICAL for every world size."""
    import numpy as np
    script = tmp_path / "worlds.py"
    script.write_text(_WORLDS)
    res = {}
    for world, port in ((1, 29771), (2, 29773), (4, 29775)):
        out = tmp_path / ("w%d.npz" % world)
        procs, outs = _run_ranks(script, world, {"DT_ONE_DEVICE": "1", "DT_DIST_BACKEND": "gloo", "DT_WORLDS_OUT": str(out)}, port, timeout=900)
        for r, (p, o) in enumerate(zip(procs, outs)):
            assert p.returncode == 0 and ("RANK %d OK" % r) in o, o[-3000:]
        res[world] = dict(np.load(out))
    report = {"config": "9 clips x T=30 x 416x416, C=12, reference-default thresholds, frame-shard at world 1 / 2 / 4 on one GPU (gloo)"}
    base = res[1]
    assert int(base["pinned_counts"].sum()) > 9 * 30 * 5            # the comparison is about real boxes
    for world in (2, 4):
        r = res[world]
        for tag in ("default", "pinned"):
            cnt_same = r[tag + "_counts"] == base[tag + "_counts"]
            frames_diff = int((~cnt_same).sum())
            # frames with equal counts: cells / labels / ids compared box by box
            same = cnt_same.copy()
            for i, t in zip(*np.nonzero(cnt_same)):
                n = int(base[tag + "_counts"][i, t])
                a, b = r[tag + "_boxes"][i, t, :n], base[tag + "_boxes"][i, t, :n]
                if not (np.array_equal(a[:, 5], b[:, 5]) and np.array_equal(a[:, 7], b[:, 7]) and
                        np.array_equal(r[tag + "_gids"][i, t, :n], base[tag + "_gids"][i, t, :n])):
                    same[i, t] = False
            report["world%d_%s" % (world, tag)] = {
                "frames_with_another_box_set_or_ids": int((~same).sum()), "frames_with_another_count": frames_diff, "frames": int(same.size),
                "max_box_value_difference": float(np.abs(r[tag + "_boxes"] - base[tag + "_boxes"]).max()),
                "bit_identical": bool(np.array_equal(r[tag + "_boxes"], base[tag + "_boxes"]) and np.array_equal(r[tag + "_gids"], base[tag + "_gids"]))}
    d = os.path.join(ROOT, "gpurun_out")
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, "parity_r06_world_sizes.json"), "w") as f:
        json.dump(report, f, indent=1)
    print(json.dumps(report))
    for world in (1, 2, 4):      # deterministic=True is ONE rounding of the network whichever rows travel (rows="z": the owner runs the projection)
        for k in ("boxes", "counts", "gids"):
            assert np.array_equal(res[world]["pinnedz_" + k], res[1]["pinned_" + k]), (world, k)
    for world in (2, 4):
        assert report["world%d_pinned" % world]["bit_identical"], report
        assert report["world%d_default" % world]["max_box_value_difference"] < 1.0      # default policy: rounding-level values, discrete flips reported above

"""CPU suite, part 2: host logic, the C-ABI surface, the N>1 path on gloo.
No compute entry point is called here (there is no GPU in this container)."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    import mi355_dt
    hdr = open(os.path.join(ROOT, "include", "mi355_dt.h")).read()
    declared = sorted(set(re.findall(r"DT_API[^;(]*?\b(dt_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 20
    assert sorted(mi355_dt.SYMBOLS) == declared, "binding and header disagree"
    assert os.path.exists(mi355_dt.LIB_PATH), "libmi355_dt.so must be built in-tree (python -m object_tracking_amd.build)"
    lib = ctypes.CDLL(mi355_dt.LIB_PATH)
    for s in declared:
        assert hasattr(lib, s), "missing export " + s
    assert lib.dt_abi_version() == 107


def test_no_cpu_fallback_without_gpu():
    import torch
    import mi355_dt
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(mi355_dt.NativeError):
        mi355_dt.Context()
    from utility.utils import decode_netout
    with pytest.raises(mi355_dt.NativeError):
        decode_netout(np.zeros((3, 3, 5, 9), dtype=np.float32), 0.5, 0.45, [1.0] * 10, 4)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "object_tracking_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, re.M), f
                assert "liboracle" not in src, f


def test_weight_reader_golden(golden_dir, tmp_path):
    from utility.utils import WeightReader
    d = np.load(os.path.join(golden_dir, "weight_reader.npz"))
    p = tmp_path / "w.weights"
    d["blob"].tofile(str(p))
    wr = WeightReader(str(p))
    assert np.array_equal(wr.read_bytes(5), d["first5"])
    assert np.array_equal(wr.read_bytes(3), d["next3"])
    wr.reset()
    assert wr.offset == 4


def test_normalize_golden(golden_dir):
    from utility.utils import normalize
    d = np.load(os.path.join(golden_dir, "normalize.npz"))
    assert np.array_equal(normalize(d["img"]), d["out"])


def test_darknet_stream_size_known_answer():
    """reading what the C=80 graph asks for consumes 50,983,561
    floats -> file length 16 + 4*N = 203,934,260 B, the public yolov2.weights."""
    from utility.synth import darknet_blob_size
    n = darknet_blob_size(80)
    assert n - 4 == 50983561
    assert 4 * n == 203934260
    assert darknet_blob_size(12) - 4 == 50635061


def test_layer_tables_agree_with_file_order():
    from oracle import oracle as orc
    from utility.synth import FILE_ORDER
    sizes = [4 * co + co * ci * k * k for (_, k, ci, co) in FILE_ORDER]
    assert sum(sizes) + 85 + 85 * 1024 == 50635061        # C=12 head
    assert [s[0] for s in FILE_ORDER] == list(range(1, 23))
    assert [t[0] for t in orc.TRUNK] == list(range(1, 21))


def test_class_surface_defaults():
    from models_detection.KerasYOLO import KerasYOLO
    from models_tracking.MultiObjDetTracker import MultiObjDetTracker
    import trainer
    assert (KerasYOLO.IMAGE_H, KerasYOLO.GRID_H, KerasYOLO.BOX, KerasYOLO.CLASS) == (416, 13, 5, 80)
    assert KerasYOLO.OBJ_THRESHOLD == 0.5 and KerasYOLO.NMS_THRESHOLD == 0.45
    assert KerasYOLO.ANCHORS[:2] == [0.57273, 0.677385] and KerasYOLO.weight_path == 'darknet/yolov2.weights'
    assert MultiObjDetTracker.SEQUENCE_LENGTH == 4 and MultiObjDetTracker.CLASS == 12
    assert MultiObjDetTracker.SAVED_MODEL_PATH.endswith('CHKPNT-03-0.55.hdf5')
    for m in ("load_model", "load_weights", "normalize_input", "extract", "predict", "train"):
        assert callable(getattr(KerasYOLO, m))
    for m in ("load_model", "load_weights", "predict", "train"):
        assert callable(getattr(MultiObjDetTracker, m))
    for f in ("single_object_tracking", "simult_multi_obj_detection_tracking", "keras_yolo_obj_detection"):
        assert callable(getattr(trainer, f))


def test_oracle_resize_properties():
    """the ingest restatement: identity at equal size, constants stay constant, and it stays
    within one grey level of exact (float64) half-pixel-centre bilinear interpolation"""
    from oracle import oracle as orc
    rs = np.random.RandomState(0)
    img = rs.randint(0, 256, size=(2, 20, 30, 3)).astype(np.uint8)
    assert np.array_equal(orc.resize_bilinear_u8(img, 20, 30), img)
    assert orc.resize_bilinear_u8(img, 40, 60).shape == (2, 40, 60, 3)
    const = np.full((1, 7, 9, 3), 77, dtype=np.uint8)
    assert np.all(orc.resize_bilinear_u8(const, 32, 32) == 77)
    H, W, oh, ow = 20, 30, 13, 17
    ys = (np.arange(oh) + .5) * (H / oh) - .5; xs = (np.arange(ow) + .5) * (W / ow) - .5
    y0 = np.floor(ys).astype(int); x0 = np.floor(xs).astype(int)
    fy = np.where(y0 < 0, 0, ys - y0)[:, None, None]; fx = np.where(x0 < 0, 0, xs - x0)[None, :, None]
    y0c, y1c = np.clip(y0, 0, H - 1), np.clip(y0 + 1, 0, H - 1)
    x0c, x1c = np.clip(x0, 0, W - 1), np.clip(x0 + 1, 0, W - 1)
    im = img[0].astype(np.float64)
    ref = (im[y0c][:, x0c] * (1 - fx) + im[y0c][:, x1c] * fx) * (1 - fy) + (im[y1c][:, x0c] * (1 - fx) + im[y1c][:, x1c] * fx) * fy
    assert np.abs(orc.resize_bilinear_u8(img[:1], oh, ow)[0].astype(np.float64) - ref).max() <= 1.0


def test_shard_range_partitions():
    from parallel import shard_range
    for n in (1, 7, 8, 30, 33):
        for world in (1, 2, 4, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for a, b in zip(spans, spans[1:]):
                assert a[1] == b[0]
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_global_track_ids():
    import torch
    from parallel import global_track_ids
    ids = torch.tensor([[[0, 1, -1]], [[0, -1, -1]], [[1, 0, 2]]], dtype=torch.int32)
    nids = torch.tensor([2, 1, 3], dtype=torch.int32)
    g = global_track_ids(ids, nids)
    assert g.tolist() == [[[0, 1, -1]], [[2, -1, -1]], [[4, 3, 5]]]


_WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
import object_tracking_amd
from parallel import gather_detections, shard_range, init_from_env
rank, world, _ = init_from_env("gloo")
torch.manual_seed(0)
N, T, cap = 5, 3, 4
boxes = torch.rand(N, T, cap, 8); counts = torch.randint(0, cap + 1, (N, T), dtype=torch.int32)
ids = torch.randint(-1, 3, (N, T, cap), dtype=torch.int32); nids = torch.randint(1, 4, (N,), dtype=torch.int32)
a, b = shard_range(N, rank, world)
res = dict(boxes=boxes[a:b], counts=counts[a:b], ids=ids[a:b], nids=nids[a:b])
n_max = max(shard_range(N, r, world)[1] - shard_range(N, r, world)[0] for r in range(world))
out = gather_detections(res, n_clips_max=n_max)
full = gather_detections(dict(boxes=boxes, counts=counts, ids=ids, nids=nids)) if False else None
from parallel import global_track_ids
ok = (torch.equal(out["boxes"], boxes) and torch.equal(out["counts"], counts) and torch.equal(out["ids"], ids)
      and torch.equal(out["gids"], global_track_ids(ids, nids)))
print("RANK", rank, "OK" if ok else "MISMATCH", flush=True)
dist.barrier(); dist.destroy_process_group()
sys.exit(0 if ok else 1)
'''


@pytest.mark.parametrize("world", [2, 8])
def test_gather_detections_gloo(tmp_path, world):
    """N>1 path on CPU: 2 / 8 processes, gloo, uneven shards (3+2 clips; 5 clips on 8 ranks: three ranks hold none) ->
    identical global table and globally unique ids on every rank, equal to the 1-process result."""
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29731 + 100 * (world == 8)), WORLD_SIZE=str(world))
    procs = []
    for r in range(world):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT], env=e, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=180)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, o
        assert "RANK %d OK" % r in o, o


_WORKER_ROWS = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
import object_tracking_amd
from parallel import gather_frame_rows, init_from_env
rank, world, _ = init_from_env("gloo")
n_seq, T, D = 3, 8, 5
full = torch.arange(n_seq * T * D, dtype=torch.float32).reshape(n_seq, T, D)
t_loc = T // world
out = gather_frame_rows(full[:, rank * t_loc:(rank + 1) * t_loc].contiguous())
ok = torch.equal(out, full)
print("RANK", rank, "OK" if ok else "MISMATCH", flush=True)
dist.barrier(); dist.destroy_process_group()
sys.exit(0 if ok else 1)
'''


def test_gather_frame_rows_gloo_world2(tmp_path):
    """configs[3] exchange: time-sharded per-frame rows are stitched back in order."""
    script = tmp_path / "worker_rows.py"
    script.write_text(_WORKER_ROWS)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29733", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=180)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, o
        assert "RANK %d OK" % r in o, o


# ---- data-side host logic (utility/preprocessing.py) ----------------------
def test_parse_annotation_matches_reference_golden(golden_dir):
    """preprocessing.py:12-77 exec'd on tests/golden/ann/ by tools/make_goldens.py.  os.walk's
    directory order is the file system's, so images are compared keyed by file name."""
    import json
    from utility.preprocessing import parse_annotation
    want = json.load(open(os.path.join(golden_dir, "parse_annotation.json")))
    cwd = os.getcwd()
    os.chdir(golden_dir)
    try:
        for key, labels in [("all", []), ("car_person", ["car", "person"]), ("none", ["zebra"])]:
            imgs, seen = parse_annotation("ann/", "frames/", labels)
            assert seen == want[key]["seen"]
            by_name = lambda lst: sorted(lst, key=lambda r: r["filename"])
            assert by_name(imgs) == by_name(want[key]["images"])
    finally:
        os.chdir(cwd)
    assert len(want["all"]["images"]) == 4 and want["none"]["images"] == []


def test_sequence_windows_match_reference_golden(golden_dir):
    from utility.preprocessing import create_sequences_from_parsed_annotations, sequence_window_starts
    z = np.load(os.path.join(golden_dir, "windows.npz"))
    for i in range(int(z["n"])):
        folders, T = z["folders_%d" % i].tolist(), int(z["T_%d" % i])
        data = [{"folder": "f%d/" % f, "i": k} for k, f in enumerate(folders)]
        if int(z["err_%d" % i]):
            with pytest.raises(IndexError):
                create_sequences_from_parsed_annotations(data, T)
            continue
        assert sequence_window_starts(folders, T) == z["starts_%d" % i].tolist()
        seqs = create_sequences_from_parsed_annotations(data, T)
        assert [s[0]["i"] for s in seqs] == z["starts_%d" % i].tolist()
        assert all(len(s) == T and len({d["folder"] for d in s}) == 1 for s in seqs)


def test_pack_objects_layout():
    from utility.preprocessing import pack_objects
    recs = [{"width": 640, "height": 480, "object": [{"name": "b", "xmin": 1, "ymin": 2, "xmax": 3, "ymax": 4},
                                                      {"name": "zz", "xmin": 5, "ymin": 6, "xmax": 7, "ymax": 8}]},
            {"width": 320, "height": 240, "object": []}]
    objs, counts, dims = pack_objects(recs, ["a", "b", "b"])
    assert objs.shape == (2, 2, 5) and objs.dtype == np.int32
    assert objs[0].tolist() == [[1, 2, 3, 4, 1], [5, 6, 7, 8, -1]]      # first index of a repeated label
    assert counts.tolist() == [2, 0] and dims.tolist() == [[640, 480], [320, 240]]
    with pytest.raises(ValueError):
        pack_objects(recs, ["a"], cap=1)


def test_pack_objects_without_size_block_and_short_batches():
    """records parsed from XML without a <size> block carry no width/height (the generators then use the decoded
    image's dims, preprocessing.py:144); fewer items than one batch give a short batch, not a negative slice"""
    from utility.preprocessing import BatchGenerator, pack_objects
    recs = [{"filename": "x.png", "folder": "a/", "object": [{"name": "car", "xmin": 1, "ymin": 2, "xmax": 30, "ymax": 40}]}]
    objs, counts, dims = pack_objects(recs, ["car"])
    assert counts.tolist() == [1] and dims.tolist() == [[0, 0]] and objs[0, 0].tolist() == [1, 2, 30, 40, 0]
    g = BatchGenerator.__new__(BatchGenerator)
    g.config = {"BATCH_SIZE": 4}
    g.images = list(range(3))
    assert g._bounds(0) == (0, 3)
    g.images = list(range(10))
    assert g._bounds(0) == (0, 4) and g._bounds(2) == (6, 10)


_WORKER_FRAMESHARD = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
import object_tracking_amd
from parallel import gather_detections, global_track_ids, track_clips_frame_sharded, init_from_env, frame_shard_times
rank, world, _ = init_from_env("gloo")

class Ctx(object):                         # torch-CPU stand-in for the two halves of dt_track_forward
    grid = (2, 3)
    device = torch.device("cpu")
    calls = []
    def track_row_width(self): return 8
    def track_detect(self, fr):            # per-frame rows: a function of that frame only
        self.calls.append(fr.shape[0])
        m = fr.float().mean(dim=(1, 2, 3))
        return m.view(-1, 1, 1, 1) * torch.ones(fr.shape[0], 2, 3, 8) + torch.arange(8.0)
    def track_recurrent(self, z):          # a recurrence over T: running sum
        return torch.cumsum(z, dim=1)
    # the split one step later: detector + input projection per frame (here: x2, rows twice as wide), recurrence on those rows
    def track_xproj_width(self): return 16
    def track_detect_xproj(self, fr):
        z = self.track_detect(fr)
        return torch.cat([z, 2.0 * z], dim=-1)
    def track_recurrent_xproj(self, xp):
        return torch.cumsum(xp[..., :8], dim=1)

class Det(object):
    class model(object):
        @staticmethod
        def to_device(x): return x

class Trk(object):
    detector = Det()
    class model(object):
        ctx = Ctx()
    def decode_and_associate(self, g, cap=None):
        n, T = g.shape[:2]
        boxes = g.reshape(n, T, -1)[:, :, :cap * 8].reshape(n, T, cap, 8).contiguous()
        counts = (g.reshape(n, T, -1).sum(-1) % 3).to(torch.int32)
        ids = (boxes[..., 0] % 5).to(torch.int32)
        nids = torch.full((n,), 5, dtype=torch.int32)
        return dict(boxes=boxes, counts=counts, ids=ids, nids=nids, netout=g)
    def empty_result(self, T, cap=None):
        return dict(boxes=torch.zeros(0, T, cap, 8), counts=torch.zeros(0, T, dtype=torch.int32),
                    ids=torch.zeros(0, T, cap, dtype=torch.int32), nids=torch.zeros(0, dtype=torch.int32), netout=None)
    def track_clips(self, frames, cap=None):
        n, T = frames.shape[:2]
        z = self.model.ctx.track_detect(frames.reshape((n * T,) + tuple(frames.shape[2:])))
        return self.decode_and_associate(self.model.ctx.track_recurrent(z.reshape(n, T, 2, 3, 8)), cap=cap)

torch.manual_seed(1)
ok = True
for (n_clips, T) in [(3, 5), (1, 4), (4, 2), (5, 9), (2, 1)]:   # uneven time shards, fewer clips than ranks, T == world, T < world
    frames = torch.randint(0, 255, (n_clips, T, 4, 4, 3), dtype=torch.uint8)
    trk = Trk()
    want = trk.track_clips(frames, cap=4)          # the single-process table, no collective
    want["gids"] = global_track_ids(want["ids"], want["nids"])
    mine = frame_shard_times(T, rank, world)
    for chunks in (1, 2, 3):
        for local, rows in ((False, "z"), (True, "z"), (True, "xproj"), (False, None)):   # whole batch on every rank / sharded ingest; which rows travel
            Ctx.calls = []
            st = {}
            rwid = 8 if rows == "z" else 16
            if local:
                got = track_clips_frame_sharded(trk, frames[:, mine].contiguous(), cap=4, T=T, chunks=chunks, stats=st, rows=rows)
            else:
                got = track_clips_frame_sharded(trk, frames, cap=4, chunks=chunks, stats=st, rows=rows)
            ok &= all(torch.equal(got[k], want[k]) for k in ("boxes", "counts", "ids", "nids", "gids"))
            ok &= sum(Ctx.calls) == n_clips * len(mine)          # the detector ran on this rank's frames only
            # rows arrive at the clip's owner only: (own clips) x (other ranks' time steps) rows of 2*3*8 floats
            own = len(range(rank, n_clips, world))
            want_rows = own * (T - len(mine)) * 2 * 3 * rwid * 4
            row_ints = T * 4 * 8 + T * 4 + T + 2
            want_det = (world - 1) * ((n_clips + world - 1) // world) * row_ints * 4
            ok &= st["bytes_received"] == want_rows + want_det
print("RANK", rank, "OK" if ok else "MISMATCH", flush=True)
dist.barrier(); dist.destroy_process_group()
sys.exit(0 if ok else 1)
'''


@pytest.mark.parametrize("world", [2, 4, 8])
def test_frame_sharded_tracker_gloo(tmp_path, world):
    """configs[4] split: detector frame-shard {t : t mod N = r} (whole batch on every rank, or
    sharded ingest: only the rank's own frames), rows sent to the clip's round-robin owner with chunked all_to_all,
    recurrence on the owner, detection gather in global clip order -- with a torch-CPU stand-in for the two library
    halves, 2 / 4 / 8 gloo ranks give exactly the single-process table incl. the global ids; the detector ran on each rank's
    own frames only; the bytes a rank received are the owner-only volume."""
    script = tmp_path / "worker_fs.py"
    script.write_text(_WORKER_FRAMESHARD)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29735 + 10 * world), WORLD_SIZE=str(world))
    procs = [subprocess.Popen([sys.executable, str(script), ROOT], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, o
        assert "RANK %d OK" % r in o, o


def test_gather_detections_finds_padding_itself(tmp_path):
    """uneven shards with n_clips_max="max": one scalar all-reduce finds the padding (the default, None, means equal shards and no extra collective)"""
    script = tmp_path / "worker_auto.py"
    script.write_text(_WORKER.replace("out = gather_detections(res, n_clips_max=n_max)", "out = gather_detections(res, n_clips_max=\"max\")"))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29737", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=180)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, o
        assert "RANK %d OK" % r in o, o

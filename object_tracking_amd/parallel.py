"""Multi-GPU sharding of the detect-and-track path: one process per GPU,
`torch.distributed` (backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU
tests).

The reference has no distributed code at all (zero
collective call sites).  Design: streams/clips are
independent, so each rank owns a contiguous block of clips and runs detector +
ConvLSTM recurrence + decode + association for them with NO data-path
collective.  The only exchange is the cross-stream step north_star names: an
all-gather of the fixed-size, padded per-frame detection records plus per-clip
track counts, after which every rank holds the complete detection table and
track ids are made globally unique by an exclusive prefix sum of the per-clip
id counts in GLOBAL clip order -- so 1-GPU and N-GPU runs produce bit-identical
global ids.  Messages are a few KB per frame (latency-bound, not xGMI-bandwidth
bound): one collective per step for all frames of all local clips, never one
per frame.
"""
import torch
import torch.distributed as dist


def shard_range(n_total, rank, world):
    """Contiguous block partition of n_total clips: returns (start, stop)."""
    base, rem = divmod(n_total, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def _all_gather_cat(t, group=None):
    """all-gather equal-shaped tensors along dim 0.  With the nccl (= RCCL)
    backend the device tensors go straight over xGMI; with gloo (CPU tests, or a
    functional test of the multi-process path on a single GPU) device tensors are
    staged through the host."""
    world = dist.get_world_size(group)
    backend = dist.get_backend(group)
    dev = t.device
    src = t.contiguous()
    if backend == "gloo" and src.is_cuda:
        src = src.cpu()
    out = torch.empty((world * src.shape[0],) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    if backend == "nccl" and hasattr(dist, "all_gather_into_tensor"):
        dist.all_gather_into_tensor(out, src, group=group)
    else:
        parts = list(out.chunk(world, dim=0))
        dist.all_gather(parts, src, group=group)
    return out.to(dev)


def global_track_ids(ids, nids):
    """ids [n_clips,T,cap] (per-clip ids from 0, -1 = unused), nids [n_clips] ->
    globally unique ids: id + exclusive_cumsum(nids)[clip]."""
    off = torch.cumsum(nids.to(torch.int64), 0) - nids.to(torch.int64)
    g = ids.to(torch.int64) + off.view(-1, 1, 1)
    return torch.where(ids >= 0, g, torch.full_like(g, -1))


def _pack_rows(boxes, counts, ids, nids, n_pad):
    """One int32 row per clip: [boxes T*cap*8 (float bits) | ids T*cap | counts T | nids | valid]; rows beyond the
    local clips are empty (valid = 0).  One buffer -> ONE collective per step (the exchange is latency-bound)."""
    n, T, cap = ids.shape
    row = T * cap * 8 + T * cap + T + 2
    buf = torch.zeros((n_pad, row), dtype=torch.int32, device=boxes.device)
    o = 0
    buf[:n, o:o + T * cap * 8] = boxes.contiguous().view(torch.int32).reshape(n, T * cap * 8); o += T * cap * 8
    buf[:n, o:o + T * cap] = ids.reshape(n, T * cap); o += T * cap
    buf[:n, o:o + T] = counts.reshape(n, T); o += T
    buf[:n, o] = nids
    buf[:n, o + 1] = 1
    return buf


def _unpack_rows(buf, T, cap):
    keep = buf[:, -1] == 1
    buf = buf[keep]
    n = buf.shape[0]
    o = 0
    boxes = buf[:, o:o + T * cap * 8].contiguous().view(torch.float32).reshape(n, T, cap, 8); o += T * cap * 8
    ids = buf[:, o:o + T * cap].reshape(n, T, cap).contiguous(); o += T * cap
    counts = buf[:, o:o + T].contiguous(); o += T
    nids = buf[:, o].contiguous()
    return boxes, counts, ids, nids


def gather_detections(res, n_clips_max=None, group=None):
    """Cross-stream exchange.  `res` is MultiObjDetTracker.track_clips output for
    this rank's clips (device tensors).  Returns the dict for ALL clips of all ranks in global clip
    order with an extra `gids` tensor of globally unique track ids.  Shards may be uneven: rows are padded to
    `n_clips_max` clips per rank (pass it when known, e.g. ceil(total / world) -- otherwise one extra scalar
    all-reduce finds it).  ONE all-gather of one packed buffer per step.  Without an initialised process group
    (single process) only the id globalisation is applied."""
    boxes, counts, ids, nids = res["boxes"], res["counts"], res["ids"], res["nids"]
    n_local = boxes.shape[0]
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        if n_clips_max is None:
            m = torch.tensor([n_local], dtype=torch.int64, device=boxes.device if dist.get_backend(group) == "nccl" else "cpu")
            dist.all_reduce(m, op=dist.ReduceOp.MAX, group=group)
            n_clips_max = int(m.item())
        T, cap = ids.shape[1], ids.shape[2]
        allrows = _all_gather_cat(_pack_rows(boxes, counts, ids, nids, n_clips_max), group)
        boxes, counts, ids, nids = _unpack_rows(allrows, T, cap)
    return dict(boxes=boxes, counts=counts, ids=ids, nids=nids, gids=global_track_ids(ids, nids))


# ---- frame-shard of the detect+track path (BASELINE.json configs[4]) --------------------
def frame_shard_times(T, rank, world):
    """time steps whose DETECTOR pass runs on `rank`: t = rank, rank + world, ... (round-robin keeps every rank's
    share within one frame of the others for any T)."""
    return list(range(rank, T, world))


def stitch_frame_rows(gathered, T, world):
    """gathered [world, n_clips, Tl, ...] (rank r holds times r, r+world, ...; slots past T are padding) ->
    [n_clips, T, ...] in time order."""
    w, n_clips, Tl = gathered.shape[:3]
    assert w == world
    out = gathered.permute(1, 2, 0, *range(3, gathered.dim()))          # [n_clips, Tl, world, ...]: t = j*world + r
    out = out.reshape((n_clips, Tl * world) + tuple(gathered.shape[3:]))
    return out[:, :T].contiguous()


def track_clips_frame_sharded(trk, frames, cap=None, group=None):
    """MultiObjDetTracker on clips whose frames are spread over the ranks of `group`: ONE stream can use all GPUs.
      1. detector (the 74 % of a frame's FLOPs) on this rank's time steps of EVERY clip (dt_track_detect),
      2. all-gather of the per-frame rows z = [conv_feat | x_bbox] (757 KB/frame at 416x416; over xGMI this is
         ~0.3 GB/s per GPU at 375 frames/s -- far below one link) and stitching into time order,
      3. ConvLSTM recurrence + 1x1 + decode + association on the OWNER of each clip (contiguous block partition of
         the clips, shard_range): the recurrence is sequential in T, so it cannot be split further,
      4. the cross-stream detection all-gather of gather_detections.
    `frames` [n_clips,T,H,W,3] must be the same on every rank (a rank only touches its own time steps).  Returns
    the global table like gather_detections.  Single process: identical to track_clips + gather_detections."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return gather_detections(trk.track_clips(frames, cap=cap))
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    ctx = trk.model.ctx
    frames = trk.detector.model.to_device(frames)
    n_clips, T = frames.shape[:2]
    mine = frame_shard_times(T, rank, world)
    Tl = (T + world - 1) // world
    gh, gw = ctx.grid
    z_local = torch.zeros((n_clips, Tl, gh, gw, ctx.track_row_width()), dtype=torch.float32, device=ctx.device)
    if mine:
        sub = frames[:, mine].contiguous()
        z = ctx.track_detect(sub.reshape((n_clips * len(mine),) + tuple(frames.shape[2:])))
        z_local[:, :len(mine)] = z.reshape((n_clips, len(mine)) + tuple(z.shape[1:]))
    z_all = stitch_frame_rows(_all_gather_cat(z_local.unsqueeze(0), group), T, world)
    lo, hi = shard_range(n_clips, rank, world)
    if hi > lo:
        res = trk.decode_and_associate(ctx.track_recurrent(z_all[lo:hi].contiguous()), cap=cap)
    else:
        res = trk.empty_result(T, cap)
    return gather_detections(res, n_clips_max=(n_clips + world - 1) // world, group=group)


def gather_frame_rows(rows_local, group=None):
    """Frame-sharded TinyTracker (BASELINE.json configs[3]): every rank ran the
    detector + pooling on its contiguous slice of the time axis and holds
    rows_local [n_seq, T_local, D]; all-gather the small rows (a few KB per
    frame) and stitch the time axis back in rank order -> [n_seq, world*T_local, D]
    on every rank, which then runs the (cheap, sequential) LSTM replicated."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return rows_local
    world = dist.get_world_size(group)
    n_seq, t_loc, D = rows_local.shape
    allr = _all_gather_cat(rows_local.reshape(1, n_seq, t_loc, D), group)      # [world, n_seq, T_local, D]
    return allr.permute(1, 0, 2, 3).reshape(n_seq, world * t_loc, D).contiguous()


def init_from_env(backend=None):
    """Initialise the default process group from torchrun's environment
    (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT).  Returns
    (rank, world, local_rank); no-op for a single process."""
    import os
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("DT_ONE_DEVICE") == "1":     # functional test: every rank on GPU 0 (use with gloo)
        local = 0
    if backend is None:
        backend = os.environ.get("DT_DIST_BACKEND")
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if torch.cuda.is_available():
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    elif torch.cuda.is_available():
        torch.cuda.set_device(local)
    return rank, world, local

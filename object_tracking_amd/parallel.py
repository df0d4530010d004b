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
import os

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
    assert n <= n_pad, ("this rank holds %d clips but the exchange was sized for at most %d per rank: pass "
                        "n_clips_max >= the largest clip count of any rank (or None to let one all-reduce find it)" % (n, n_pad))
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


def gather_detections(res, n_clips_max=None, group=None, ctx=None, clip_ids=None, stats=None):
    """Cross-stream exchange.  `res` is MultiObjDetTracker.track_clips output for
    this rank's clips (device tensors).  Returns the dict for ALL clips of all ranks in global clip
    order with an extra `gids` tensor of globally unique track ids.  Shards may be uneven: rows are padded to
    `n_clips_max` clips per rank.  n_clips_max=None / "max" (the default): one extra scalar all-reduce finds the largest shard -- safe for
    any partition.  A caller that KNOWS its partition passes the number (ceil(total / world)), or n_clips_max="equal" (every rank holds as
    many clips as this one: the clip-shard's normal case) -- then no rank issues anything but the ONE all-gather of one packed buffer per
    step.  ("equal" on shards that are not equal would hand the collective buffers of different sizes: it is an assertion by the caller,
    and the valid-row count of the result is checked against world x n_local.)  Without an initialised process group (single process)
    only the id globalisation is applied.
      ctx       a mi355_dt.Context: pack / unpack / id globalisation run as the library's kernels
                (dt_pack_detections / dt_unpack_detections) instead of torch indexing -- the path a C-ABI caller has;
      clip_ids  global clip index of every gathered row in rank-major order (block partition: omit; round-robin
                owners of the frame-shard: see track_clips_frame_sharded) -- rows are put in global clip order
                BEFORE the ids are globalised, so every partition gives the single-process ids;
      stats     dict that receives `bytes_received` (bytes of other ranks' rows this rank took in)."""
    boxes, counts, ids, nids = res["boxes"], res["counts"], res["ids"], res["nids"]
    n_local = boxes.shape[0]
    native = ctx is not None and boxes.is_cuda
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        world = dist.get_world_size(group)
        equal = isinstance(n_clips_max, str) and n_clips_max == "equal"
        if equal:
            n_clips_max = n_local          # the caller's assertion: no scalar collective
        elif n_clips_max is None or (isinstance(n_clips_max, str) and n_clips_max == "max"):
            m = torch.tensor([n_local], dtype=torch.int64, device=boxes.device if dist.get_backend(group) == "nccl" else "cpu")
            dist.all_reduce(m, op=dist.ReduceOp.MAX, group=group)
            n_clips_max = int(m.item())
        T, cap = ids.shape[1], ids.shape[2]
        mine = ctx.pack_detections(boxes.contiguous(), counts.contiguous(), ids.contiguous(), nids.contiguous(), n_clips_max) \
            if native else _pack_rows(boxes, counts, ids, nids, n_clips_max)
        allrows = _all_gather_cat(mine, group)
        if stats is not None:
            stats["bytes_received"] = stats.get("bytes_received", 0) + (world - 1) * mine.numel() * 4
        if native and clip_ids is None:
            boxes, counts, ids, nids, gids = ctx.unpack_detections(allrows.contiguous(), T, cap)
            if equal and boxes.shape[0] != world * n_local:
                raise RuntimeError("gather_detections(n_clips_max='equal'): %d clips arrived from %d ranks, this rank holds %d -- the shards are not equal" % (boxes.shape[0], world, n_local))
            return dict(boxes=boxes, counts=counts, ids=ids, nids=nids, gids=gids)
        boxes, counts, ids, nids = _unpack_rows(allrows, T, cap)
        if equal and boxes.shape[0] != world * n_local:
            raise RuntimeError("gather_detections(n_clips_max='equal'): %d clips arrived from %d ranks, this rank holds %d -- the shards are not equal" % (boxes.shape[0], world, n_local))
        if clip_ids is not None:
            order = torch.argsort(torch.as_tensor(list(clip_ids), dtype=torch.int64)).to(boxes.device)
            assert order.numel() == boxes.shape[0], "clip_ids names %d rows, the exchange delivered %d" % (order.numel(), boxes.shape[0])
            boxes, counts, ids, nids = boxes[order], counts[order], ids[order], nids[order]
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


def owned_clips(n_clips, rank, world):
    """clips whose RECURRENCE runs on `rank`: round-robin (clip c -> rank c mod world), so that any number of clips
    spreads over as many ranks as it can and every rank's count is within one of the others."""
    return list(range(rank, n_clips, world))


def _all_to_all_rows(send, group, async_op):
    """send [world, ...] (slice d goes to rank d) -> (recv [world, ...] (slice s came from rank s), work handle).
    nccl: device buffers straight over xGMI; gloo: staged through the host."""
    backend = dist.get_backend(group)
    src = send.contiguous()
    if backend == "gloo" and src.is_cuda:
        src = src.cpu()
    out = torch.empty_like(src)
    work = dist.all_to_all_single(out, src, group=group, async_op=async_op)
    return out, work


class pinned_policy(object):
    """DT_PIN=1 on ONE live context (dt_policy_set: no process-wide environment variable is touched, other contexts and threads keep
    their policy): the library's kernel selection no longer looks at the batch a call carries, so a frame is the same rounding of the
    network whatever batch it travels in -- for any world size / `chunks`.  The context is un-pinned on exit (captured hipGraphs are
    dropped when the value changes: a deployment that replays graphs pins its context once, not per call)."""

    def __init__(self, ctx, on=True):
        self.ctx, self.on = ctx, on

    def __enter__(self):
        if self.on:
            self.ctx.policy_set("pin", 1)
        return self

    def __exit__(self, *exc):
        if self.on:
            self.ctx.policy_set("pin", 0)


def track_clips_frame_sharded(trk, frames, cap=None, group=None, T=None, chunks=2, stats=None, rows=None, deterministic=False):
    """MultiObjDetTracker on clips whose frames are spread over the ranks of `group`: ONE stream can use all GPUs.
      1. detector (the 74 % of a frame's FLOPs) on this rank's time steps {t : t mod N = rank} of EVERY clip
         (dt_track_detect).  `frames` is either the whole [n_clips,T,H,W,3] batch (a rank then touches only its own
         time steps of it) or -- sharded ingest, pass T -- only this rank's frames [n_clips, len(frame_shard_times),
         H,W,3]: nothing but a rank's own frames has to reach its HBM;
         With rows="xproj" (the default where the context offers dt_track_detect_xproj) the same rank also runs the
         ConvLSTM2D INPUT projection of its frames -- it does not depend on the recurrence, and it is 55 % of the recurrent
         head's FLOPs -- so only the sequential part is left for the clip's owner; rows="z" exchanges the detector rows;
      2. the per-frame rows (xproj [G,G,4U]: 1.38 MB/frame at 416x416; z = [conv_feat | x_bbox]: 757 KB) go to the OWNER of their clip only
         (round-robin owners, `owned_clips`): one all_to_all_single per chunk of the local time axis, issued
         asynchronously so that the exchange of chunk k runs under the detector pass of chunk k+1.  A rank receives
         (its clips) x T rows -- 1/N of what an all-gather would deliver;
      3. ConvLSTM recurrence + 1x1 + decode + association on the owner (the recurrence is sequential in T, so it
         cannot be split further; with fewer clips than ranks the surplus ranks idle in this phase);
      4. the cross-stream detection all-gather of gather_detections, rows put back in global clip order.
    Returns the global table like gather_detections.  With ONE process it is track_clips + gather_detections, bit for bit.
    Across ranks the VALUES agree with the single-process result to float32 rounding only (boxes within 1e-3, the bar the
    tests pin): a rank's detector batch is n_clips * len(chunk) frames, and the library's kernel selection depends on the
    batch (split-bf16 vs fp32 MFMA GEMMs by row count, F(4x4) vs F(6x6), the frame-mosaic groups, split-K), so the
    same frame is a different rounding of the same network for another world size or `chunks`.  Everything discrete --
    counts, cells, labels, track ids -- is identical unless a score / IoU lies within that rounding (~1e-4) of its
    threshold.  A deployment that needs ids that do not vary with the number of ranks pins the selection with the
    selection.  deterministic=True runs the call under DT_PIN=1 (`pinned_policy`): the selection then ignores the batch, every frame is
    computed by the same kernels in the same order for any world size, and boxes and ids are BIT-IDENTICAL between 1, 2, 4, 8 ranks
    (tests/test_gpu_multi.py::test_frame_shard_world_sizes_agree) -- at the price of the small-batch optimisations.
    `stats` (dict) receives bytes_received (rows + detection records from other ranks) for this call."""
    if deterministic:
        with pinned_policy(trk.model.ctx):
            return track_clips_frame_sharded(trk, frames, cap=cap, group=group, T=T, chunks=chunks, stats=stats, rows=rows)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return gather_detections(trk.track_clips(frames, cap=cap))
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    ctx = trk.model.ctx
    frames = trk.detector.model.to_device(frames)
    n_clips = frames.shape[0]
    local_only = T is not None
    if T is None:
        T = frames.shape[1]
    mine_t = frame_shard_times(T, rank, world)
    if local_only:
        assert frames.shape[1] == len(mine_t), "rank %d owns %d time steps of T=%d, got %d frames per clip" % (rank, len(mine_t), T, frames.shape[1])
    Tl = (T + world - 1) // world                      # local slots per clip, padded
    n_own = (n_clips + world - 1) // world             # clips per owner, padded
    gh, gw = ctx.grid
    if rows is None:
        rows = "xproj" if hasattr(ctx, "track_detect_xproj") else "z"
    assert rows in ("xproj", "z")
    rw = ctx.track_xproj_width() if rows == "xproj" else ctx.track_row_width()
    detect = ctx.track_detect_xproj if rows == "xproj" else ctx.track_detect
    recurrent = ctx.track_recurrent_xproj if rows == "xproj" else ctx.track_recurrent
    # clip order of the send buffer: owner-major, each owner's clips in increasing index
    perm = [c for r in range(world) for c in owned_clips(n_clips, r, world)]
    slot_of = {}
    for r in range(world):
        for k, c in enumerate(owned_clips(n_clips, r, world)):
            slot_of[c] = (r, k)
    # the schedule is the same on every rank (a collective needs equal shapes everywhere): the PADDED local time axis
    # [0, Tl) is cut into `chunks` pieces; a rank whose last slot does not exist (T not a multiple of N) sends zeros
    chunks = max(1, min(int(chunks), Tl))
    bounds = [(Tl * i) // chunks for i in range(chunks + 1)]
    pending = []
    timed = stats is not None and torch.cuda.is_available() and ctx.device.type == "cuda"
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)] if timed else None
    if timed:
        ev[0].record()
    for ci in range(chunks):
        j0, j1 = bounds[ci], bounds[ci + 1]
        send = torch.zeros((world, n_own, j1 - j0, gh, gw, rw), dtype=torch.float32, device=ctx.device)
        jr = min(j1, len(mine_t))                      # slots of this chunk that exist on this rank
        if jr > j0:
            sub = (frames[:, j0:jr] if local_only else frames[:, mine_t[j0:jr]]).contiguous()
            z = detect(sub.reshape((n_clips * (jr - j0),) + tuple(frames.shape[2:])))
            z = z.reshape((n_clips, jr - j0) + tuple(z.shape[1:]))
            for c in range(n_clips):
                r, k = slot_of[c]
                send[r, k, :jr - j0] = z[c]
        # issued asynchronously: the exchange of this chunk overlaps the detector pass of the next one (nccl orders the
        # collective after the kernels already queued on the current stream)
        recv, work = _all_to_all_rows(send, group, async_op=True)
        pending.append((j0, j1, recv, work))
    if timed:
        ev[1].record()         # everything before: the part that is spread over all ranks (detector [+ input projection])
    my_owned = owned_clips(n_clips, rank, world)
    z_mine = torch.zeros((len(my_owned), T, gh, gw, rw), dtype=torch.float32, device=ctx.device)
    recv_bytes = 0
    for j0, j1, recv, work in pending:
        work.wait()
        recv = recv.to(ctx.device)
        for src in range(world):
            ts = frame_shard_times(T, src, world)[j0:j1]
            if ts and my_owned:
                z_mine[:, ts] = recv[src, :len(my_owned), :len(ts)]
                if src != rank:
                    recv_bytes += len(my_owned) * len(ts) * gh * gw * rw * 4
    if my_owned:
        res = trk.decode_and_associate(recurrent(z_mine), cap=cap)
    else:
        res = trk.empty_result(T, cap)
    if stats is not None:
        stats["bytes_received"] = stats.get("bytes_received", 0) + recv_bytes
        stats["clips_owned"] = len(my_owned)
        if timed:
            ev[2].record()     # ev[1] .. ev[2]: waiting for rows + the sequential part on the clips this rank owns (0 clips: idle)
            stats["stage_events"] = ev      # read with frame_shard_stage_ms() after a synchronize
    return gather_detections(res, n_clips_max=n_own, group=group, clip_ids=perm, stats=stats,
                             ctx=ctx if getattr(ctx, "pack_detections", None) else None)


def frame_shard_stage_ms(stats):
    """(sharded-stage ms, owner-stage ms) of the last track_clips_frame_sharded call that was handed `stats`, from the
    events it recorded on the current stream; call after torch.cuda.synchronize().  The owner stage of a rank that owns
    no clip (fewer clips than ranks) is the time it idles while the owners run the recurrence."""
    ev = stats.get("stage_events") if stats else None
    if not ev:
        return None
    return ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2])


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


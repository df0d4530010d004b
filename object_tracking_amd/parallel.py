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


def gather_detections(res, n_clips_max=None, group=None):
    """Cross-stream exchange.  `res` is MultiObjDetTracker.track_clips output for
    this rank's clips (device tensors).  Every rank must pass the same number of
    clips or give `n_clips_max` (rows are padded with empty clips).  Returns the
    dict for ALL clips of all ranks in global clip order with an extra `gids`
    tensor of globally unique track ids.  Without an initialised process group
    (single process) only the id globalisation is applied."""
    boxes, counts, ids, nids = res["boxes"], res["counts"], res["ids"], res["nids"]
    n_local = boxes.shape[0]
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        n_pad = n_clips_max if n_clips_max is not None else n_local
        valid = torch.zeros((n_pad,), dtype=torch.int32, device=boxes.device)
        valid[:n_local] = 1

        def pad(t, fill):
            if t.shape[0] == n_pad:
                return t
            p = torch.full((n_pad - t.shape[0],) + tuple(t.shape[1:]), fill, dtype=t.dtype, device=t.device)
            return torch.cat([t, p], 0)

        boxes = _all_gather_cat(pad(boxes, 0.0), group)
        counts = _all_gather_cat(pad(counts, 0), group)
        ids = _all_gather_cat(pad(ids, -1), group)
        nids = _all_gather_cat(pad(nids, 0), group)
        keep = _all_gather_cat(valid, group).bool()
        boxes, counts, ids, nids = boxes[keep], counts[keep], ids[keep], nids[keep]
    return dict(boxes=boxes, counts=counts, ids=ids, nids=nids, gids=global_track_ids(ids, nids))


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

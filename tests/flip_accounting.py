"""Accounting of the DISCRETE decisions of the detect-and-track path at fixed (reference-default) thresholds.

Two float32 implementations of the same graph differ by rounding (~1e-5 on the grid).  Everything downstream of the
grid is a cascade of comparisons against fixed thresholds (utils.py:216 `score > obj_threshold`, :249 `bbox_iou >=
nms_threshold`, :255 the final filter; DESIGN.md section 6 `IoU >= ASSOC_THRESHOLD`), so a decision value that
lies closer to its threshold than the value error can legitimately come out differently -- and one flipped box
renumbers every later track of its clip.  This module turns "ids are bit-exact" into a checkable statement at the
operating point a user of the reference gets (OBJ 0.5 / NMS 0.45, KerasYOLO.py:43-44; ASSOC 0.3):

  * eps_s / eps_iou = the MEASURED error of the decision values that can flip (class scores within 0.01 of the score
    threshold; IoUs of candidate pairs within 0.01 of a threshold), over the whole configuration -- and the error of
    ALL scores / IoUs is asserted to be below 0.01 (and below the 1e-3 parity bar), so nothing outside that window can;
  * a decision of the ORACLE is "in band" when its margin to the threshold (or to the competing value, for an order
    or arg-max decision) is <= eps;
  * every frame without an in-band decode decision must come out IDENTICAL (box set, order, labels);
  * in a frame that does differ, every differing box must trace to an in-band decision (directly, or through the NMS
    interaction with a box that does);
  * track ids must be bit-identical up to each clip's first frame with a decode flip or an in-band association
    decision.

Used by tests/test_gpu_configs.py (HIP path vs oracle) and, on the CPU, by tests/test_flip_accounting.py, which
feeds it an oracle-vs-perturbed-oracle pair to check the accounting itself (a real corruption must be caught).
Test infrastructure only.
"""
import numpy as np

from oracle import oracle as orc


def _sigmoid32(x):
    return (np.float32(1.0) / (np.float32(1.0) + np.exp(-x.astype(np.float32)))).astype(np.float32)


def boxes_of_grid(grid, anchors):
    """centre-format boxes of EVERY cell of one raw grid [GH,GW,NB,5+C], float32 arithmetic of utils.py:228-233
    -> [GH*GW*NB, 4]"""
    GH, GW, NB, _ = grid.shape
    g = grid.astype(np.float32)
    col = np.arange(GW, dtype=np.float32).reshape(1, GW, 1)
    row = np.arange(GH, dtype=np.float32).reshape(GH, 1, 1)
    an = np.asarray(anchors, dtype=np.float32).reshape(NB, 2)
    x = (col + _sigmoid32(g[..., 0])) / np.float32(GW)
    y = (row + _sigmoid32(g[..., 1])) / np.float32(GH)
    w = an[:, 0].reshape(1, 1, NB) * np.exp(g[..., 2]) / np.float32(GW)
    h = an[:, 1].reshape(1, 1, NB) * np.exp(g[..., 3]) / np.float32(GH)
    return np.stack([x, y, w, h], -1).reshape(-1, 4).astype(np.float32)


def iou_matrix(a, b):
    """utils.py:155-188 on centre-format boxes, float64: [n,4] x [m,4] -> [n,m]"""
    if len(a) == 0 or len(b) == 0:
        return np.zeros((len(a), len(b)))
    a = a[:, None, :4].astype(np.float64)
    b = b[None, :, :4].astype(np.float64)
    ix = np.minimum(a[..., 0] + a[..., 2] / 2, b[..., 0] + b[..., 2] / 2) - np.maximum(a[..., 0] - a[..., 2] / 2, b[..., 0] - b[..., 2] / 2)
    iy = np.minimum(a[..., 1] + a[..., 3] / 2, b[..., 1] + b[..., 3] / 2) - np.maximum(a[..., 1] - a[..., 3] / 2, b[..., 1] - b[..., 3] / 2)
    inter = np.clip(ix, 0, None) * np.clip(iy, 0, None)
    return inter / (a[..., 2] * a[..., 3] + b[..., 2] * b[..., 3] - inter)


def oracle_scores(grid, anchors, C):
    """all class scores conf*softmax of one frame exactly as the reference forms them (utils.py:214-215): decode with
    nothing thresholded and nothing suppressed -> [ncell, C]"""
    _, post = orc.decode_netout(grid, 0.0, 2.0, anchors, C)
    return post[..., 5:].reshape(-1, C)


def measure_eps(sc_ref, sc_got, box_ref, box_got, obj_thr, iou_thrs, wide=0.25, narrow=0.01):
    """measured error of the decision values of one frame -> (e_s, e_iou, e_s_wide, e_iou_wide):
      e_s / e_iou            over the values a flip can come from: scores within `narrow` of the score threshold, IoUs of
                             candidate pairs within `narrow` of one of `iou_thrs`;
      e_s_wide / e_iou_wide  over everything within `wide` of / above the score threshold -- account() asserts these stay
                             below `narrow`, which is what makes the narrow window sufficient (a value further than
                             `narrow` from its threshold cannot cross it).
    The error of a class score scales with how undecided its softmax is (p (1 - p) times the logit error), so it is
    measured where decisions are made, not as one number for the whole grid."""
    d = np.abs(sc_got.astype(np.float64) - sc_ref.astype(np.float64))
    dist = np.abs(sc_ref.astype(np.float64) - obj_thr)
    e_s_wide = float(d[dist < wide].max()) if (dist < wide).any() else 0.0
    e_s = float(d[dist < narrow].max()) if (dist < narrow).any() else 0.0
    cells = np.nonzero(sc_ref.max(1) > obj_thr - wide)[0]
    e_iou = e_iou_wide = 0.0
    if len(cells) >= 2:
        ir = iou_matrix(box_ref[cells], box_ref[cells])
        di = np.abs(ir - iou_matrix(box_got[cells], box_got[cells]))
        e_iou_wide = float(di.max())
        for thr in iou_thrs:
            near = np.abs(ir - thr) < narrow
            if near.any():
                e_iou = max(e_iou, float(di[near].max()))
    return e_s, e_iou, e_s_wide, e_iou_wide


def decode_decisions(sc_ref, box_ref, eps_s, eps_iou, obj_thr, nms_thr):
    """in-band decode decisions of one ORACLE frame.  Returns (list of dicts, set of cells they involve)."""
    out, cells = [], set()
    band = np.argwhere(np.abs(sc_ref.astype(np.float64) - obj_thr) <= eps_s)
    for cell, c in band:
        out.append(dict(kind="score", cell=int(cell), cls=int(c), value=float(sc_ref[cell, c]),
                        margin=float(abs(float(sc_ref[cell, c]) - obj_thr))))
        cells.add(int(cell))
    cand = sc_ref.astype(np.float64) >= obj_thr - eps_s
    for c in np.nonzero(cand.sum(0) >= 2)[0]:
        idx = np.nonzero(cand[:, c])[0]
        iou = iou_matrix(box_ref[idx], box_ref[idx])
        s = sc_ref[idx, c].astype(np.float64)
        for a in range(len(idx)):
            for b in range(a + 1, len(idx)):
                if abs(iou[a, b] - nms_thr) <= eps_iou:
                    out.append(dict(kind="nms_iou", cls=int(c), cells=[int(idx[a]), int(idx[b])], value=float(iou[a, b]),
                                    margin=float(abs(iou[a, b] - nms_thr))))
                    cells.update((int(idx[a]), int(idx[b])))
                if iou[a, b] >= nms_thr - eps_iou and abs(s[a] - s[b]) <= 2.0 * eps_s:
                    out.append(dict(kind="nms_order", cls=int(c), cells=[int(idx[a]), int(idx[b])],
                                    value=float(s[a] - s[b]), margin=float(abs(s[a] - s[b]))))
                    cells.update((int(idx[a]), int(idx[b])))
    return out, cells


def explain_diff(diff_cells, band_cells, sc_ref, box_ref, eps_s, eps_iou, obj_thr, nms_thr):
    """every differing box must be an in-band cell or reach one through NMS interactions (same class above the
    threshold band, IoU >= nms_thr - eps) -- returns the cells that are NOT explained"""
    if not diff_cells:
        return set()
    cand_cells = np.nonzero((sc_ref.astype(np.float64) >= obj_thr - eps_s).any(1))[0]
    pos = {int(c): k for k, c in enumerate(cand_cells)}
    iou = iou_matrix(box_ref[cand_cells], box_ref[cand_cells])
    cls = sc_ref[cand_cells].argmax(1)
    explained = set(c for c in diff_cells if c in band_cells)
    frontier = set(band_cells) & set(pos)
    seen = set(frontier)
    while frontier:       # closure over the NMS interaction graph, starting from the in-band cells
        nxt = set()
        for c in frontier:
            k = pos[c]
            for j in np.nonzero((iou[k] >= nms_thr - eps_iou) & (cls == cls[k]))[0]:
                cj = int(cand_cells[j])
                if cj not in seen:
                    seen.add(cj)
                    nxt.add(cj)
        frontier = nxt
    explained |= set(c for c in diff_cells if c in seen)
    return set(diff_cells) - explained


def assoc_decisions(rows_t, rows_p, eps_iou, assoc_thr):
    """in-band association decisions between the ORACLE's final boxes of frame t and frame t-1"""
    out = []
    if len(rows_t) == 0 or len(rows_p) == 0:
        return out
    iou = iou_matrix(rows_t, rows_p)
    same = rows_t[:, 5][:, None] == rows_p[:, 5][None, :]
    for i in range(len(rows_t)):
        v = np.where(same[i], iou[i], -1.0)
        for j in np.nonzero(np.abs(v - assoc_thr) <= eps_iou)[0]:
            out.append(dict(kind="assoc_iou", box=int(i), prev=int(j), value=float(v[j]), margin=float(abs(v[j] - assoc_thr))))
        top = np.sort(v[v >= assoc_thr - eps_iou])[::-1]
        if len(top) >= 2 and top[0] - top[1] <= 2.0 * eps_iou:
            out.append(dict(kind="assoc_argmax", box=int(i), value=float(top[0] - top[1]), margin=float(top[0] - top[1])))
    return out


def account(ref_grids, got_grids, got_scores, got_rows, got_counts, got_ids, got_nids, anchors, C,
            obj_thr=0.5, nms_thr=0.45, assoc_thr=0.3, eps_floor=(1e-7, 1e-6), max_eps=(1e-3, 1e-3), narrow=0.01):
    """ref_grids / got_grids [n_clips,T,GH,GW,NB,5+C] raw tracking grids (oracle / implementation under test);
    got_scores [n_clips,T,ncell,C] the implementation's own conf*softmax scores (its decode at threshold 0);
    got_rows [n_clips,T,cap,8], got_counts [n_clips,T], got_ids [n_clips,T,cap], got_nids [n_clips]: its outputs at
    the fixed thresholds.  Raises AssertionError when a disagreement does not trace to an in-band decision; returns
    the report dict."""
    n_clips, T = ref_grids.shape[:2]
    cap = got_rows.shape[2]
    ncell = int(np.prod(ref_grids.shape[2:5]))
    # ---- pass 1: measured value errors over the whole configuration
    sc_ref = np.empty((n_clips, T, ncell, C), dtype=np.float32)
    bx_ref = np.empty((n_clips, T, ncell, 4), dtype=np.float32)
    eps_s, eps_iou = eps_floor
    eps_s_wide = eps_iou_wide = 0.0
    for i in range(n_clips):
        for t in range(T):
            sc_ref[i, t] = oracle_scores(ref_grids[i, t], anchors, C)
            bx_ref[i, t] = boxes_of_grid(ref_grids[i, t], anchors)
            e_s, e_iou, e_sw, e_iw = measure_eps(sc_ref[i, t], got_scores[i, t], bx_ref[i, t], boxes_of_grid(got_grids[i, t], anchors),
                                                 obj_thr, (nms_thr, assoc_thr), narrow=narrow)
            eps_s, eps_iou = max(eps_s, e_s + eps_floor[0]), max(eps_iou, e_iou + eps_floor[1])
            eps_s_wide, eps_iou_wide = max(eps_s_wide, e_sw), max(eps_iou_wide, e_iw)
    # nothing further than `narrow` from its threshold can cross it; and the parity bars themselves
    assert eps_s_wide < min(narrow, max_eps[0]), "class-score error %g (scores within 0.25 of the threshold) exceeds %g" % (eps_s_wide, min(narrow, max_eps[0]))
    assert eps_iou_wide < min(narrow, max_eps[1]), "IoU error %g of candidate pairs exceeds %g" % (eps_iou_wide, min(narrow, max_eps[1]))

    # ---- pass 2: the oracle's outputs at the fixed thresholds, in-band decisions, frame-by-frame comparison
    ref_rows = np.zeros((n_clips, T, cap, 8), dtype=np.float32)
    ref_counts = np.zeros((n_clips, T), dtype=np.int32)
    flips, dirty_frames, clean_frames, n_band = [], 0, 0, 0
    worst_coord, worst_iou, nbox = 0.0, 1.0, 0
    frame_flip = np.zeros((n_clips, T), dtype=bool)
    for i in range(n_clips):
        for t in range(T):
            rows, _ = orc.decode_netout(ref_grids[i, t], obj_thr, nms_thr, anchors, C)
            assert len(rows) <= cap, "cap %d too small for %d boxes" % (cap, len(rows))
            ref_rows[i, t, :len(rows)] = rows
            ref_counts[i, t] = len(rows)
            dec, band_cells = decode_decisions(sc_ref[i, t], bx_ref[i, t], eps_s, eps_iou, obj_thr, nms_thr)
            n_band += len(dec)
            g = got_rows[i, t, :int(got_counts[i, t])]
            same = len(g) == len(rows) and np.array_equal(g[:, 7], rows[:, 7]) and np.array_equal(g[:, 5], rows[:, 5])
            if dec:
                dirty_frames += 1
            else:
                clean_frames += 1
                assert same, ("clip %d t %d: no oracle decision lies within eps (score %g, IoU %g) of a threshold, yet the "
                              "boxes differ: %d vs %d" % (i, t, eps_s, eps_iou, len(g), len(rows)))
            if same:
                if len(rows):
                    exy = np.abs(g[:, :2] - rows[:, :2]).max()
                    ewh = (np.abs(g[:, 2:4] - rows[:, 2:4]) / np.maximum(1.0, np.abs(rows[:, 2:4]))).max()
                    worst_coord = max(worst_coord, float(exy), float(ewh))
                    worst_iou = min(worst_iou, float(np.diag(iou_matrix(g, rows)).min()))
                    nbox += len(rows)
                continue
            frame_flip[i, t] = True
            key = lambda r: set((int(c), int(l)) for c, l in zip(r[:, 7], r[:, 5]))
            diff = key(g) ^ key(rows)
            unexplained = explain_diff(set(c for c, _ in diff), band_cells, sc_ref[i, t], bx_ref[i, t], eps_s, eps_iou, obj_thr, nms_thr)
            assert not unexplained, ("clip %d t %d: boxes at cells %s differ without an in-band decision behind them "
                                     "(in band: %s)" % (i, t, sorted(unexplained), dec))
            flips.append(dict(clip=i, t=t, boxes_got=int(len(g)), boxes_ref=int(len(rows)),
                              differing=sorted([int(c), int(l)] for c, l in diff), in_band=dec))

    # ---- track ids: bit-identical up to each clip's first flip / in-band association decision
    clips_identical, clips_broken, id_mismatch, id_compared, assoc_band = 0, [], 0, 0, 0
    for i in range(n_clips):
        rid, rn = orc.associate_clip(ref_rows[i], ref_counts[i], assoc_thr)
        t_break = T
        for t in range(T):
            a = assoc_decisions(ref_rows[i, t, :ref_counts[i, t]], ref_rows[i, t - 1, :ref_counts[i, t - 1]], eps_iou, assoc_thr) if t else []
            assoc_band += len(a)
            if frame_flip[i, t] or a:
                t_break = t
                break
        assert np.array_equal(got_ids[i, :t_break], rid[:t_break]), \
            "clip %d: track ids differ before the first in-band decision (t=%d)" % (i, t_break)
        if t_break == T:
            assert int(got_nids[i]) == rn
        if np.array_equal(got_ids[i], rid) and int(got_nids[i]) == rn:
            clips_identical += 1
        else:
            clips_broken.append(dict(clip=i, first_in_band_t=int(t_break)))
        for t in range(T):
            if not frame_flip[i, t]:
                n = int(ref_counts[i, t])
                id_compared += n
                id_mismatch += int((got_ids[i, t, :n] != rid[t, :n]).sum())
    n_dec = n_clips * T * ncell * C
    return dict(obj_threshold=obj_thr, nms_threshold=nms_thr, assoc_threshold=assoc_thr,
                eps_score=eps_s, eps_iou=eps_iou, eps_score_whole_window=eps_s_wide, eps_iou_all_pairs=eps_iou_wide,
                eps_window=narrow, frames=int(n_clips * T), score_decisions=int(n_dec),
                in_band_decode_decisions=int(n_band), in_band_assoc_decisions_before_break=int(assoc_band),
                frames_with_in_band_decision=int(dirty_frames), frames_clean_and_identical=int(clean_frames),
                frames_with_a_flip=int(frame_flip.sum()), flips_per_1e5_score_decisions=1e5 * float(frame_flip.sum()) / n_dec,
                clips=int(n_clips), clips_ids_bit_identical=int(clips_identical), clips_renumbered=clips_broken,
                box_ids_compared=int(id_compared), box_ids_renumbered=int(id_mismatch),
                boxes_in_identical_frames=int(nbox), box_coord_err=worst_coord, box_iou_min=worst_iou, flips=flips)

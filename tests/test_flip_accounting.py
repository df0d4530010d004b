"""CPU check of tests/flip_accounting.py itself: the oracle against a rounding-level perturbation of its own input
must pass the accounting (flips allowed, all inside the measured band); a real corruption must be caught."""
import numpy as np
import pytest

from oracle import oracle as orc

import flip_accounting as fa

ANCHORS = [0.57273, 0.677385, 1.87446, 2.06253, 3.33843, 5.47434, 7.88282, 3.52778, 9.77052, 9.16828]


def _grids(n_clips, T, G, C, seed, dense):
    """tracking grids with many objectness values near the 0.5 score threshold (dense=True: thousands of decisions
    inside +-1e-4, so that a 1e-5 perturbation flips some) and slowly moving boxes so that NMS and association act"""
    rng = np.random.default_rng(seed)
    g = np.zeros((n_clips, T, G, G, 5, 5 + C), dtype=np.float32)
    base = rng.normal(0, 0.3, (n_clips, 1, G, G, 5, 4)).astype(np.float32)
    g[..., :4] = base + rng.normal(0, 0.02, (n_clips, T, G, G, 5, 4)).astype(np.float32)
    spread = 2e-3 if dense else 1.0
    g[..., 4] = rng.uniform(-spread, spread, g.shape[:-1]).astype(np.float32) + np.where(rng.random(g.shape[:-1]) < 0.5, 0.0, -3.0)
    cls = rng.integers(0, C, g.shape[:-1])
    g[..., 5:] = -30.0
    np.put_along_axis(g[..., 5:], cls[..., None], 30.0, axis=-1)      # peaky softmax: score ~ objectness
    return g


def _run_impl(grids, C, cap):
    """the 'implementation under test' of the CPU check: the oracle itself on `grids`"""
    n_clips, T = grids.shape[:2]
    ncell = int(np.prod(grids.shape[2:5]))
    sc = np.zeros((n_clips, T, ncell, C), dtype=np.float32)
    rows = np.zeros((n_clips, T, cap, 8), dtype=np.float32)
    counts = np.zeros((n_clips, T), dtype=np.int32)
    ids = np.zeros((n_clips, T, cap), dtype=np.int32)
    nids = np.zeros(n_clips, dtype=np.int32)
    for i in range(n_clips):
        for t in range(T):
            sc[i, t] = fa.oracle_scores(grids[i, t], ANCHORS, C)
            r, _ = orc.decode_netout(grids[i, t], 0.5, 0.45, ANCHORS, C)
            rows[i, t, :len(r)] = r
            counts[i, t] = len(r)
        ids[i], nids[i] = orc.associate_clip(rows[i], counts[i], 0.3)
    return sc, rows, counts, ids, nids


def test_accounting_passes_on_rounding_level_noise_and_reports_flips():
    C, G = 3, 6
    ref = _grids(4, 8, G, C, seed=3, dense=True)
    rng = np.random.default_rng(4)
    got = ref + rng.normal(0, 1e-5, ref.shape).astype(np.float32)
    got[..., 5:] = ref[..., 5:]
    cap = G * G * 5
    rep = fa.account(ref, got, *_run_impl(got, C, cap), ANCHORS, C)
    assert rep["frames_with_a_flip"] > 0, "the dense configuration is meant to produce flips"
    assert rep["frames_with_a_flip"] <= rep["frames_with_in_band_decision"]
    assert rep["eps_score"] < 1e-4 and rep["eps_score_whole_window"] < 1e-3
    for f in rep["flips"]:
        assert f["in_band"], "a reported flip carries the in-band decisions behind it"
        assert all(d["margin"] <= max(2 * rep["eps_score"], rep["eps_iou"]) for d in f["in_band"])


def test_accounting_is_exact_without_noise():
    C, G = 3, 5
    ref = _grids(2, 5, G, C, seed=7, dense=False)
    rep = fa.account(ref, ref.copy(), *_run_impl(ref, C, G * G * 5), ANCHORS, C)
    assert rep["frames_with_a_flip"] == 0 and rep["clips_ids_bit_identical"] == 2 and rep["box_ids_renumbered"] == 0


def test_accounting_catches_a_real_error():
    C, G = 3, 5
    ref = _grids(2, 5, G, C, seed=9, dense=False)
    cap = G * G * 5
    sc, rows, counts, ids, nids = _run_impl(ref, C, cap)
    # (a) a box dropped although no decision is near a threshold
    t = int(np.argmax(counts[0] > 1))
    r2, c2 = rows.copy(), counts.copy()
    r2[0, t, :-1] = rows[0, t, 1:]
    c2[0, t] -= 1
    with pytest.raises(AssertionError):
        fa.account(ref, ref.copy(), sc, r2, c2, ids, nids, ANCHORS, C)
    # (b) a wrong track id in a clip with no in-band decision
    i2 = ids.copy()
    tt = int(np.argmax(counts[1] > 0))
    i2[1, tt, 0] += 5
    with pytest.raises(AssertionError):
        fa.account(ref, ref.copy(), sc, rows, counts, i2, nids, ANCHORS, C)
    # (c) a grid error far above rounding level
    bad = ref.copy()
    bad[0, 0, ..., 4] += 0.01
    with pytest.raises(AssertionError):
        fa.account(ref, bad, *_run_impl(bad, C, cap), ANCHORS, C)

#!/usr/bin/env python3
"""Generate golden vectors for decode_netout / NMS / bbox_iou by EXECUTING the
reference's own numpy code.

Runs only in the build container (needs /root/reference).  The reference module
utility/utils.py is Python 2 (print statement at line 10) and imports cv2, so it
cannot be imported; its hot-path block -- BoundBox, WeightReader, normalize,
bbox_iou, interval_overlap (lines 113-188) and decode_netout, sigmoid, softmax
(lines 208-270) -- is py3-clean and numpy-only.  This script reads those line
ranges at run time and exec()s them with `np` in scope; no reference source is
copied into this repository, only the resulting data (inputs + expected outputs)
is written to tests/golden/*.npz.

    python tools/make_goldens.py            # rewrites tests/golden/
"""
import os
import sys

import numpy as np

REF = os.environ.get("REFERENCE_ROOT", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")

ANCHORS = [0.57273, 0.677385, 1.87446, 2.06253, 3.33843, 5.47434, 7.88282, 3.52778, 9.77052, 9.16828]


def load_reference_slice():
    path = os.path.join(REF, "utility", "utils.py")
    with open(path) as f:
        lines = f.read().split("\n")
    src = "\n".join(lines[112:188] + lines[207:270])
    ns = {"np": np}
    exec(compile(src, path + "[113-188,208-270]", "exec"), ns)
    return ns


def planted_grid(seed, G, C, n_obj, dup_every=3):
    """SURVEY.md section 8c recipe: background logits with objectness pushed
    down, n_obj planted objects, every dup_every-th duplicated into the next
    cell (same class, shifted tx) to force NMS suppressions."""
    rs = np.random.RandomState(seed)
    g = rs.randn(G, G, 5, 5 + C).astype(np.float32)
    g[..., 4] -= 4.0
    cells = rs.permutation(G * (G - 1))[:n_obj]
    for k, cell in enumerate(cells):
        row, col = divmod(int(cell), G - 1)
        b = int(rs.randint(0, 5))
        cls = int(rs.randint(0, C))
        g[row, col, b, 4] = 4.0 + rs.rand()
        g[row, col, b, 5 + cls] += 12.0 + rs.rand()
        if dup_every and k % dup_every == 0:
            g[row, col + 1, b, :] = g[row, col, b, :]
            g[row, col + 1, b, 0] -= 3.0
            g[row, col + 1, b, 4] -= 0.25 + 0.5 * rs.rand()  # distinct score, no ties
    return g


def run_case(ns, netout, obj_thr, nms_thr, nb_class):
    work = netout.copy()
    boxes = ns["decode_netout"](work, obj_thr, nms_thr, ANCHORS, nb_class)
    rows = np.zeros((len(boxes), 7), dtype=np.float64)
    cls = np.zeros((len(boxes), nb_class), dtype=np.float32)
    for i, b in enumerate(boxes):
        rows[i] = [b.x, b.y, b.w, b.h, b.c, b.get_label(), b.get_score()]
        cls[i] = b.classes
    return rows, cls, work


def make_preprocessing_goldens(ns):
    """SURVEY.md 8f.3.  preprocessing.py cannot be imported (py2 print at l.334, cv2/imgaug/keras
    imports), so three of its statement ranges are exec'd from the file at run time:
      79-89   create_sequences_from_parsed_annotations (as is)
      172-188 the `for obj in all_objs` coordinate fix of aug_image (loop body run with the
              augmentation draw (scale, offx, offy, flip) given instead of sampled; image I/O skipped)
      214-293 BatchGenerator.output_from_instance (as a method of a stand-in `self` whose
              aug_image runs the 172-188 range)
    BoundBox / bbox_iou come from the utils.py slice above."""
    import copy
    import textwrap
    path = os.path.join(REF, "utility", "preprocessing.py")
    with open(path) as f:
        lines = f.read().split("\n")
    wns = {}
    exec(compile("\n".join(lines[78:89]), path + "[79-89]", "exec"), wns)
    make_windows = wns["create_sequences_from_parsed_annotations"]
    fix_src = compile(textwrap.dedent("\n".join(lines[171:188])), path + "[172-188]", "exec")
    mns = {"np": np, "os": os, "BoundBox": ns["BoundBox"], "bbox_iou": ns["bbox_iou"]}
    exec(compile(textwrap.dedent("\n".join(lines[213:293])), path + "[214-293]", "exec"), mns)
    encode = mns["output_from_instance"]

    # windows: folder id per frame -> list of start indices (or IndexError)
    rs = np.random.RandomState(21)
    wcases = []
    for k, (runs, T) in enumerate([([7], 4), ([5, 6, 4], 4), ([3, 9, 2, 8], 4), ([4, 4], 4), ([10, 1, 10], 3),
                                   ([2, 2, 2, 12], 5), ([6, 3], 4), ([4], 4), ([3], 4), ([9, 9, 9], 1)]):
        folders = np.concatenate([np.full(r, i, dtype=np.int32) for i, r in enumerate(runs)])
        data = [{"folder": "f%d/" % f, "i": i} for i, f in enumerate(folders)]
        try:
            seqs = make_windows(data, T)
            starts = np.array([s[0]["i"] for s in seqs], dtype=np.int32)
            assert all([d["i"] for d in s] == list(range(s[0]["i"], s[0]["i"] + T)) for s in seqs)
            err = 0
        except IndexError:
            starts, err = np.zeros(0, dtype=np.int32), 1
        wcases.append((folders, T, starts, err))
    np.savez_compressed(os.path.join(OUT, "windows.npz"), n=np.int32(len(wcases)),
                        **{"folders_%d" % i: c[0] for i, c in enumerate(wcases)},
                        **{"T_%d" % i: np.int32(c[1]) for i, c in enumerate(wcases)},
                        **{"starts_%d" % i: c[2] for i, c in enumerate(wcases)},
                        **{"err_%d" % i: np.int32(c[3]) for i, c in enumerate(wcases)})


    # parse_annotation (preprocessing.py:12-77) on the XML fixtures under tests/golden/ann/
    import json
    import xml.etree.ElementTree as ET
    pns = {"os": os, "ET": ET}
    exec(compile("\n".join(lines[11:77]), path + "[12-77]", "exec"), pns)
    cwd = os.getcwd()
    os.chdir(OUT)                      # keep the recorded paths relative
    try:
        parsed = {}
        for key, labels in [("all", []), ("car_person", ["car", "person"]), ("none", ["zebra"])]:
            imgs, seen = pns["parse_annotation"]("ann/", "frames/", labels)
            parsed[key] = {"images": imgs, "seen": seen}
    finally:
        os.chdir(cwd)
    with open(os.path.join(OUT, "parse_annotation.json"), "w") as f:
        json.dump(parsed, f, indent=1, sort_keys=True)
    print("parse_annotation: %s" % {k: len(v["images"]) for k, v in parsed.items()})

    class Stub(object):
        pass

    def run_encode(cfg, objs_list, dims, aug):
        n = len(objs_list)
        y = np.zeros((n, cfg["GRID_H"], cfg["GRID_W"], cfg["BOX"], 5 + cfg["CLASS"]))
        b = np.zeros((n, cfg["TRUE_BOX_BUFFER"], 4))
        for i in range(n):
            st = Stub()
            st.config, st.debug, st.norm = cfg, False, None
            st.augment = aug is not None
            st.anchors = [ns["BoundBox"](0, 0, cfg["ANCHORS"][2 * a], cfg["ANCHORS"][2 * a + 1])
                          for a in range(len(cfg["ANCHORS"]) // 2)]

            def aug_image(train_instance, augment, i=i, st=st):
                env = {"self": st, "all_objs": copy.deepcopy(train_instance["object"]), "augment": augment,
                       "resize": True, "w": int(dims[i][0]), "h": int(dims[i][1]), "int": int, "float": float,
                       "max": max, "min": min}
                if augment:
                    env.update(scale=float(aug[i][0]), offx=int(aug[i][1]), offy=int(aug[i][2]), flip=float(aug[i][3]))
                exec(fix_src, env)
                return np.zeros((2, 2, 3)), env["all_objs"]
            st.aug_image = aug_image
            (_, bi), yi = encode(st, {"object": objs_list[i], "filename": "x/y.jpg"}, 0)
            y[i] = yi
            b[i] = bi[0, 0, 0]
        return y, b

    labels12 = ["l%d" % i for i in range(12)]
    labels20 = ["l%d" % i for i in range(20)]
    out = {}
    for name, G, IM, labels, TBB, n, max_obj, use_aug, seed in [
            ("g13_c12", 13, 416, labels12, 50, 12, 40, False, 31),
            ("g13_c12_aug", 13, 416, labels12, 50, 12, 40, True, 32),
            ("g19_c20_wrap", 19, 608, labels20, 4, 8, 24, False, 33),
            ("g13_dense_cell", 13, 416, labels12, 50, 6, 30, False, 34)]:
        rs = np.random.RandomState(seed)
        cfg = dict(IMAGE_H=IM, IMAGE_W=IM, GRID_H=G, GRID_W=G, BOX=5, CLASS=len(labels), LABELS=labels,
                   ANCHORS=ANCHORS, TRUE_BOX_BUFFER=TBB)
        objs = np.full((n, max_obj, 5), -1, dtype=np.int32)
        counts = np.zeros(n, dtype=np.int32)
        dims = np.zeros((n, 2), dtype=np.int32)
        aug = np.zeros((n, 4)) if use_aug else None
        objs_list = []
        for i in range(n):
            w, h = int(rs.randint(320, 1920)), int(rs.randint(240, 1080))
            dims[i] = (w, h)
            if use_aug:
                scale = rs.uniform() / 10. + 1.
                aug[i] = (scale, int(rs.uniform() * (scale - 1.) * w), int(rs.uniform() * (scale - 1.) * h),
                          float(rs.binomial(1, .5)))
            k = int(rs.randint(0, max_obj + 1)) if i else max_obj
            counts[i] = k
            lst = []
            for j in range(k):
                if name == "g13_dense_cell":        # many objects in few cells: overwrite + class-bit residue
                    cx, cy = w * (0.3 + 0.1 * rs.rand()), h * (0.3 + 0.1 * rs.rand())
                else:
                    cx, cy = w * rs.rand(), h * rs.rand()
                bw, bh = w * (0.02 + 0.5 * rs.rand() ** 2), h * (0.02 + 0.5 * rs.rand() ** 2)
                xmin, xmax = int(round(cx - bw / 2)), int(round(cx + bw / 2))
                ymin, ymax = int(round(cy - bh / 2)), int(round(cy + bh / 2))
                kind = rs.randint(0, 12)
                if kind == 0:
                    xmax = xmin                        # degenerate -> skipped
                if kind == 1:
                    xmin, xmax = w - 3, w + 40         # clamps at the right edge
                lab = int(rs.randint(0, len(labels))) if kind != 2 else -1     # -1: name not in LABELS
                objs[i, j] = (xmin, ymin, xmax, ymax, lab)
                lst.append({"name": labels[lab] if lab >= 0 else "other", "xmin": xmin, "ymin": ymin,
                            "xmax": xmax, "ymax": ymax})
            objs_list.append(lst)
        y, b = run_encode(cfg, objs_list, dims, aug)
        out.update({name + "/objs": objs, name + "/counts": counts, name + "/dims": dims,
                    name + "/cfg": np.array([G, IM, len(labels), TBB], dtype=np.int32), name + "/y": y, name + "/b": b})
        if use_aug:
            out[name + "/aug"] = aug
        print("targets %-16s frames %d  objects set %d" % (name, n, int((y[..., 4] == 1).sum())))
    np.savez_compressed(os.path.join(OUT, "targets.npz"), anchors=np.asarray(ANCHORS, dtype=np.float64), **out)


def main():
    ns = load_reference_slice()
    os.makedirs(OUT, exist_ok=True)
    cases = {}

    # --- big planted cases ---
    cases["g13_c80_n24"] = (planted_grid(101, 13, 80, 24), 0.5, 0.45, 80)
    cases["g13_c12_n32"] = (planted_grid(102, 13, 12, 32), 0.5, 0.45, 12)
    cases["g19_c12_n128"] = (planted_grid(103, 19, 12, 128), 0.5, 0.45, 12)

    # --- edge cases on small grids ---
    rs = np.random.RandomState(7)
    bg = rs.randn(3, 3, 5, 5 + 12).astype(np.float32)
    bg[..., 4] -= 6.0
    cases["g3_background"] = (bg, 0.5, 0.45, 12)

    # softmax rescale branch: one logit far below the global max (utils.py:265-266)
    g = planted_grid(11, 5, 12, 6)
    g[0, 0, 0, 5 + 3] = -250.0
    cases["g5_rescale"] = (g, 0.5, 0.45, 12)

    # two classes above a LOW threshold in one box; the stronger one is
    # suppressed by a neighbour so the post-NMS argmax relabels (utils.py:255)
    g = np.full((3, 3, 5, 5 + 4), -8.0, dtype=np.float32)
    g[..., :4] = 0.0
    g[1, 1, 2, 4] = 6.0                           # box A: anchor 2, scores ~0.60 / ~0.40
    g[1, 1, 2, 5:9] = [2.0, 1.6, -6.0, -6.0]
    g[1, 1, 3, 4] = 8.0                           # box B: same centre, anchor 3, class 0 ~1.0
    g[1, 1, 3, 5:9] = [6.0, -6.0, -6.0, -6.0]     # IoU(A,B) ~ 0.34 >= 0.3 -> A.class0 zeroed
    cases["g3_relabel"] = (g, 0.3, 0.3, 4)

    # low thresholds, many survivors & suppressions, non-square class count
    cases["g7_c5_lowthr"] = (planted_grid(21, 7, 5, 30, dup_every=2), 0.3, 0.3, 5)
    # high nms threshold (nothing suppressed) / tiny nms threshold (aggressive)
    cases["g7_c12_nms09"] = (planted_grid(22, 7, 12, 20, dup_every=2), 0.5, 0.9, 12)
    cases["g7_c12_nms01"] = (planted_grid(23, 7, 12, 20, dup_every=2), 0.5, 0.1, 12)
    # single class
    cases["g5_c1"] = (planted_grid(24, 5, 1, 8), 0.5, 0.45, 1)
    # dense: every cell hot (maximum candidate count for the grid)
    g = np.random.RandomState(25).randn(4, 4, 5, 5 + 3).astype(np.float32)
    g[..., 4] = 5.0 + np.random.RandomState(26).rand(4, 4, 5).astype(np.float32)
    g[..., 5] += 9.0 + np.random.RandomState(27).rand(4, 4, 5).astype(np.float32)
    cases["g4_dense"] = (g, 0.5, 0.45, 3)

    summary = []
    for name, (netout, obj_thr, nms_thr, C) in cases.items():
        rows, cls, post = run_case(ns, netout, obj_thr, nms_thr, C)
        # reject exact score ties between surviving boxes of one class (tie order
        # is undefined in the reference, utils.py:240)
        for c in range(C):
            s = post[..., 5 + c].ravel()
            s = s[s != 0]
            assert len(np.unique(s)) == len(s), (name, "score tie in class", c)
        small = netout.size <= 5 * 5 * 5 * 17
        np.savez_compressed(
            os.path.join(OUT, "decode_%s.npz" % name),
            netout=netout, obj_threshold=np.float32(obj_thr), nms_threshold=np.float32(nms_thr),
            anchors=np.asarray(ANCHORS, dtype=np.float32), nb_class=np.int32(C),
            boxes=rows, classes=cls,
            **({"netout_post": post} if small else {}))
        summary.append((name, netout.shape, len(rows)))

    # --- bbox_iou known answers (utils.py:155-188) ---
    rs = np.random.RandomState(5)
    BB = ns["BoundBox"]
    pairs = rs.rand(256, 8).astype(np.float32)
    pairs[:, 2:4] = pairs[:, 2:4] * 0.5 + 0.01
    pairs[:, 6:8] = pairs[:, 6:8] * 0.5 + 0.01
    pairs[:32, 4:6] = pairs[:32, 0:2]            # concentric
    pairs[32:40, 4:] = pairs[32:40, :4]          # identical -> 1.0
    pairs[40:48, 4] = pairs[40:48, 0] + 5.0      # disjoint -> 0.0
    iou = np.zeros(256, dtype=np.float64)
    for i, p in enumerate(pairs):
        iou[i] = ns["bbox_iou"](BB(*p[:4]), BB(*p[4:]))
    np.savez_compressed(os.path.join(OUT, "bbox_iou.npz"), pairs=pairs, iou=iou)

    # --- heatmap helpers (utils.py:53-79), exec'd from their own line range ---
    with open(os.path.join(REF, "utility", "utils.py")) as f:
        lines = f.read().split("\n")
    hns = {"np": np}
    exec(compile("\n".join(lines[52:79]), "utils.py[53-79]", "exec"), hns)
    rs = np.random.RandomState(9)
    box4 = rs.rand(96, 4).astype(np.float32)
    box4[:, 2:] *= 0.6
    box4[:8] = [[0.5, 0.5, 0.2, 0.3], [0.05, 0.05, 0.3, 0.3], [0.98, 0.97, 0.2, 0.2], [0.5, 0.5, 0.0, 0.0],
                [0.5, 0.5, 1.5, 1.5], [0.0, 0.0, 0.0, 0.0], [0.25, 0.75, 0.5, 0.5], [1.0, 1.0, 0.1, 0.1]]
    heat = np.zeros((96, 32 * 32), dtype=np.float32)
    for i, (cx, cy, w, h) in enumerate(box4):
        cx, cy, w, h = float(cx), float(cy), float(w), float(h)
        heat[i] = hns["generate_heatmap_feat"](cx - w / 2.0, cy - h / 2.0, w, h, hmap_size=32)   # preprocessing.py:455
    soft = rs.rand(64, 32, 32).astype(np.float32)
    soft[0] = 0.0
    soft[1, 5, 7] = 0.9
    rects = np.array([hns["generate_rectangle_from_heatmap"](m, 0.75, 32) for m in soft], dtype=np.int32)
    np.savez_compressed(os.path.join(OUT, "heatmap.npz"), box4=box4, heat=heat, soft=soft, rects=rects)

    # --- WeightReader known answer (utils.py:138-148): offset starts at 4 ---
    blob = np.arange(64, dtype=np.float32)
    p = os.path.join(OUT, "_tmp.weights")
    blob.tofile(p)
    wr = ns["WeightReader"](p)
    a = wr.read_bytes(5).copy()
    b = wr.read_bytes(3).copy()
    os.remove(p)
    np.savez_compressed(os.path.join(OUT, "weight_reader.npz"), blob=blob, first5=a, next3=b)

    # --- normalize known answer (utils.py:150-153) ---
    img = np.arange(256, dtype=np.uint8)
    np.savez_compressed(os.path.join(OUT, "normalize.npz"), img=img, out=ns["normalize"](img))


    # --- sequence windows + YOLO target encoding (utility/preprocessing.py:79-89, 171-188, 214-293) ---
    make_preprocessing_goldens(ns)

    for s in summary:
        print("%-18s grid %-18s -> %d boxes" % (s[0], s[1], s[2]))


if __name__ == "__main__":
    sys.exit(main())

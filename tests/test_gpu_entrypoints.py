"""GPU suite, part 3 (-m gpu): the reference's own entry points, called the way a user of the reference calls them
(paths in, files out) on generated image / weight / checkpoint files:

  KerasYOLO.extract(input_path, layer)            models_detection/KerasYOLO.py:509-520   any layer name
  KerasYOLO.predict(input_path, output_path)      :522-537                               (tests/test_gpu_parity.py)
  MultiObjDetTracker.predict(input_paths, outs)   models_tracking/MultiObjDetTracker.py:295-315
  trainer.keras_yolo_obj_detection()              trainer.py:22-30   (weights at darknet/yolov2.weights, images under darknet/data/)
  trainer.simult_multi_obj_detection_tracking()   trainer.py:18-20   (checkpoint at models/MultiObjDetTracker-CHKPNT-03-0.55.hdf5)
"""
import os

import numpy as np
import pytest
import torch

from oracle import oracle as orc
from utility import synth

pytestmark = pytest.mark.gpu

ANCHORS = [0.57273, 0.677385, 1.87446, 2.06253, 3.33843, 5.47434, 7.88282, 3.52778, 9.77052, 9.16828]


def chan_err(got, ref):
    g = got.reshape(-1, got.shape[-1]).astype(np.float64)
    r = ref.reshape(-1, ref.shape[-1]).astype(np.float64)
    return float((np.abs(g - r).max(0) / np.maximum(1.0, np.abs(r).max(0))).max())


def _save_png(path, rgb):
    from PIL import Image
    Image.fromarray(rgb).save(path)


def _oracle_layers(x, layers):
    """every named tensor of the detector graph for normalised frames x: conv_N (Conv2D output), norm_N
    (BatchNormalization output), leaky_re_lu_N (after LeakyReLU), max_pooling2d_k, lambda_1, concatenate_1"""
    out = {}
    pool_k = 0
    skip = None
    for (i, k, ci, co, pool) in orc.TRUNK:
        L = layers[i]
        c = orc.conv2d(x, L["kernel"]); out["conv_%d" % i] = c
        n = orc.bn_leaky(c, L["gamma"], L["beta"], L["mean"], L["var"], alpha=1.0); out["norm_%d" % i] = n
        x = np.where(n > 0, n, np.float32(0.1) * n).astype(np.float32); out["leaky_re_lu_%d" % i] = x
        if i == 13:
            skip = x
        if pool:
            pool_k += 1
            x = orc.maxpool2(x); out["max_pooling2d_%d" % pool_k] = x
    L = layers[21]
    c = orc.conv2d(skip, L["kernel"]); out["conv_21"] = c
    n = orc.bn_leaky(c, L["gamma"], L["beta"], L["mean"], L["var"], alpha=1.0); out["norm_21"] = n
    a = np.where(n > 0, n, np.float32(0.1) * n).astype(np.float32); out["leaky_re_lu_21"] = a
    out["lambda_1"] = orc.space_to_depth2(a)
    x = orc.concat_c(out["lambda_1"], x); out["concatenate_1"] = x
    L = layers[22]
    c = orc.conv2d(x, L["kernel"]); out["conv_22"] = c
    n = orc.bn_leaky(c, L["gamma"], L["beta"], L["mean"], L["var"], alpha=1.0); out["norm_22"] = n
    out["conv_feat"] = np.where(n > 0, n, np.float32(0.1) * n).astype(np.float32)
    out["conv_23"] = orc.conv2d(out["conv_feat"], layers[23]["kernel"], layers[23]["bias"])
    return out


def test_extract_any_layer_vs_oracle(tmp_path):
    """KerasYOLO.extract for every kind of layer name, on an image file (non-square network input)."""
    from models_detection.KerasYOLO import KerasYOLO
    C = 12
    blob = synth.synth_darknet_blob(C)
    det = KerasYOLO({'LABELS': [str(i) for i in range(C)], 'BATCH_SIZE': 1, 'IMAGE_H': 96, 'IMAGE_W': 160,
                     'GRID_H': 3, 'GRID_W': 5}, weights=blob)
    rgb = np.random.RandomState(5).randint(0, 256, size=(130, 210, 3)).astype(np.uint8)
    src = str(tmp_path / "in.png")
    _save_png(src, rgb)
    frame = orc.resize_bilinear_u8(np.ascontiguousarray(rgb[..., ::-1])[None], 96, 160)
    layers, _ = orc.parse_darknet_blob(blob, C)
    ref = _oracle_layers(orc.normalize_u8(frame), layers)
    names = ["conv_1", "norm_1", "leaky_re_lu_1", "max_pooling2d_1", "conv_2", "norm_2", "max_pooling2d_2", "conv_4",
             "norm_5", "max_pooling2d_3", "leaky_re_lu_7", "conv_9", "norm_12", "conv_13", "norm_13", "leaky_re_lu_13",
             "max_pooling2d_5", "conv_14", "norm_17", "conv_20", "leaky_re_lu_20", "conv_21", "norm_21", "lambda_1",
             "concatenate_1", "conv_22", "norm_22", "conv_feat", "conv_23"]
    for name in names:
        got = det.extract(src, name)
        want = ref[name][0]
        assert got.shape == want.shape, name
        assert chan_err(got, want) < 3e-4, "%s: %g" % (name, chan_err(got, want))
    assert np.array_equal(det.extract(src, "act_13"), det.extract(src, "leaky_re_lu_13"))
    assert np.array_equal(det.extract(src, "reshape_1"), det.extract(src, "conv_23"))
    with pytest.raises(ValueError):
        det.extract(src, "conv_24")
    with pytest.raises(ValueError):
        det.extract(src, "dense_1")
    # the production forward still works after extracts (they share workspaces)
    net = det.model.ctx.detect_forward(torch.from_numpy(frame).to(det.model.ctx.device)).cpu().numpy()
    assert chan_err(net.reshape(1, 3, 5, -1), ref["conv_23"]) < 3e-4


def test_extract_full_size_vs_f64_reference_graph(golden_dir, tmp_path):
    """extract at 416x416, C=80 against the float64 taps of the reference's EXECUTED graph (norm_13 = BatchNorm output
    before LeakyReLU, conv_21 = raw Conv2D output before its BatchNorm, conv_feat, conv_23): the meaning of the layer
    names is checked against the reference's own get_layer(name).output, not against this repository's reading."""
    from models_detection.KerasYOLO import KerasYOLO
    d = np.load(os.path.join(golden_dir, "graph_yolov2_416_c80.npz"))
    C = int(d["nb_class"])
    det = KerasYOLO({'LABELS': KerasYOLO.LABELS_COCO, 'BATCH_SIZE': 1, 'IMAGE_H': 416, 'IMAGE_W': 416, 'GRID_H': 13,
                     'GRID_W': 13}, weights=synth.synth_darknet_blob(C, seed=int(d["seed_blob"])))
    frame = synth.synth_clip(1, 416, 416, 3, seed=int(d["seed_frame"]))[0]
    src = str(tmp_path / "frame.png")
    _save_png(src, np.ascontiguousarray(frame[..., ::-1]))       # file holds RGB; imread returns the BGR frame the fixture used
    assert chan_err(det.extract(src, "norm_13")[::4, ::4], d["norm_13_stride4"][0]) < 3e-4
    assert chan_err(det.extract(src, "conv_21"), d["conv_21"][0]) < 3e-4
    assert chan_err(det.extract(src, "conv_feat"), d["conv_feat"][0]) < 3e-4
    assert chan_err(det.extract(src, "conv_23"), d["conv_23"][0]) < 3e-4


def _tracker_class(H, W, T):
    from models_tracking.MultiObjDetTracker import MultiObjDetTracker

    class Trk(MultiObjDetTracker):
        IMAGE_H, IMAGE_W = H, W
        GRID_H, GRID_W = H // 32, W // 32
        SEQUENCE_LENGTH = T
        LOAD_MODEL = False
        OBJ_THRESHOLD = 0.3
    return Trk


def _oracle_predict(frames_u8, blob, tw, C, obj_thr, nms_thr, assoc_thr):
    layers, _ = orc.parse_darknet_blob(blob, C)
    trk, _ = orc.tracker_forward(orc.normalize_u8(frames_u8), layers, tw)
    T = trk.shape[0]
    cap = trk.shape[1] * trk.shape[2] * 5
    rb = np.zeros((T, cap, 8), dtype=np.float32); rc = np.zeros(T, dtype=np.int32)
    for t in range(T):
        rows, _ = orc.decode_netout(trk[t], obj_thr, nms_thr, ANCHORS, C)
        rb[t, :len(rows)] = rows; rc[t] = len(rows)
    ids, _ = orc.associate_clip(rb, rc, assoc_thr)
    return rb, rc, ids


def _check_frames(per_frame, rb, rc, ids):
    total = 0
    for t, boxes in enumerate(per_frame):
        assert len(boxes) == rc[t]
        for i, b in enumerate(boxes):
            r = rb[t, i]
            assert b.get_label() == int(r[5]) and b.track_id == int(ids[t, i])
            assert max(abs(b.x - r[0]), abs(b.y - r[1])) < 1e-3
            assert abs(b.w - r[2]) <= 1e-3 * max(1.0, abs(r[2])) and abs(b.h - r[3]) <= 1e-3 * max(1.0, abs(r[3]))
        total += len(boxes)
    return total


def test_tracker_predict_on_image_files(tmp_path):
    """MultiObjDetTracker.predict(input_paths, output_paths): SEQUENCE_LENGTH image files in, annotated files out,
    per-frame boxes with track ids returned -- equal to the oracle chain (resize, graph, decode, association)."""
    from PIL import Image
    C, T, H, W = 12, 4, 96, 96
    blob = synth.synth_darknet_blob(C)
    tw = synth.synth_tracker_weights(C)
    tw["out_kernel"] = tw["out_kernel"] * 40.0
    tw["out_bias"][4::5 + C] = 1.5
    trk = _tracker_class(H, W, T)(detector_weights=blob, tracker_weights=tw)
    clip = synth.synth_clip(T, 120, 180, 3, seed=77)                    # BGR frames as cv2.imread would return them
    ins, outs = [], []
    for t in range(T):
        ins.append(str(tmp_path / ("f%02d.png" % t))); outs.append(str(tmp_path / ("o%02d.png" % t)))
        _save_png(ins[-1], np.ascontiguousarray(clip[t][..., ::-1]))
    per_frame = trk.predict(ins, outs)
    assert len(per_frame) == T and all(os.path.exists(o) and Image.open(o).size == (180, 120) for o in outs)
    resized = orc.resize_bilinear_u8(clip, H, W)
    rb, rc, ids = _oracle_predict(resized, blob, tw, C, trk.OBJ_THRESHOLD, trk.NMS_THRESHOLD, trk.ASSOC_THRESHOLD)
    assert _check_frames(per_frame, rb, rc, ids) > 0
    with pytest.raises(AssertionError):
        trk.predict(ins[:2], outs[:2])                                   # len(input_paths) == SEQUENCE_LENGTH (:296)


def test_trainer_entry_points(tmp_path, monkeypatch):
    """trainer.keras_yolo_obj_detection / simult_multi_obj_detection_tracking with the files where the reference
    looks for them: darknet/yolov2.weights (a darknet-format file, C=80 head), darknet/data/<sample>.jpg,
    models/MultiObjDetTracker-CHKPNT-03-0.55.hdf5 (a Keras HDF5 checkpoint written by utility/keras_h5.py's writer)."""
    import trainer
    from utility import keras_h5
    from models_detection.KerasYOLO import KerasYOLO
    from models_tracking.MultiObjDetTracker import MultiObjDetTracker
    monkeypatch.chdir(tmp_path)
    os.makedirs("darknet/data"); os.makedirs("models")
    blob = synth.synth_darknet_blob(80, head_std=0.05)
    blob.tofile("darknet/yolov2.weights")
    rs = np.random.RandomState(8)
    imgs = {}
    for name in ("dog.jpg", "person.jpg"):
        rgb = rs.randint(0, 256, size=(int(rs.randint(200, 300)), int(rs.randint(240, 400)), 3)).astype(np.uint8)
        from PIL import Image
        Image.fromarray(rgb).save(os.path.join("darknet/data", name), quality=95)
        imgs[name] = rgb
    monkeypatch.setattr(KerasYOLO, "OBJ_THRESHOLD", 0.2)
    found = trainer.keras_yolo_obj_detection()
    assert sorted(found) == ["dog.jpg", "person.jpg"] and all(os.path.exists(n) for n in found)
    from utility.frames import imread_bgr
    layers, used = orc.parse_darknet_blob(blob, 80)
    assert used == blob.size
    nbox = 0
    for name, boxes in found.items():
        frame = orc.resize_bilinear_u8(imread_bgr(os.path.join("darknet/data", name))[None], 416, 416)
        net, _, _ = orc.yolov2_forward(orc.normalize_u8(frame), layers)
        rows, _ = orc.decode_netout(net[0], 0.2, 0.45, ANCHORS, 80)
        assert len(boxes) == len(rows)
        assert [b.get_label() for b in boxes] == [int(v) for v in rows[:, 5]]
        for b, r in zip(boxes, rows):
            assert max(abs(b.x - r[0]), abs(b.y - r[1])) < 1e-3
            assert abs(b.w - r[2]) <= 1e-3 * max(1.0, abs(r[2])) and abs(b.h - r[3]) <= 1e-3 * max(1.0, abs(r[3]))
        nbox += len(boxes)
    assert nbox > 0

    # detect + track: the C=12 head reads the same darknet file and leaves its tail unread (KerasYOLO.py:244-274)
    tw = synth.synth_tracker_weights(12)
    tw["out_kernel"] = tw["out_kernel"] * 40.0
    tw["out_bias"][4::17] = 1.5
    keras_h5.write_tracker_checkpoint(MultiObjDetTracker.SAVED_MODEL_PATH, tw)
    monkeypatch.setattr(MultiObjDetTracker, "IMAGE_H", 96); monkeypatch.setattr(MultiObjDetTracker, "IMAGE_W", 96)
    monkeypatch.setattr(MultiObjDetTracker, "GRID_H", 3); monkeypatch.setattr(MultiObjDetTracker, "GRID_W", 3)
    monkeypatch.setattr(MultiObjDetTracker, "OBJ_THRESHOLD", 0.3)
    model = trainer.simult_multi_obj_detection_tracking()
    assert model.INITIAL_EPOCH == 3                                        # parsed from the checkpoint name (:293)
    T = MultiObjDetTracker.SEQUENCE_LENGTH
    clip = synth.synth_clip(T, 96, 96, 3, seed=5)
    ins, outs = [], []
    for t in range(T):
        ins.append("t%d.png" % t); outs.append("out%d.png" % t)
        _save_png(ins[-1], np.ascontiguousarray(clip[t][..., ::-1]))
    per_frame = model.predict(ins, outs)
    file_blob = np.fromfile("darknet/yolov2.weights", dtype=np.float32)   # read with the C=12 head: the tail stays unread
    rb, rc, ids = _oracle_predict(clip, file_blob, tw, 12, 0.3, 0.45, 0.3)
    assert _check_frames(per_frame, rb, rc, ids) > 0
    for name in ("single",):
        assert callable(trainer.ENTRY_POINTS[name])

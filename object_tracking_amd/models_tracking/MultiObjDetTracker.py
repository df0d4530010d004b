"""MultiObjDetTracker -- the reference's simultaneous detect-and-track model
(models_tracking/MultiObjDetTracker.py) with the same public surface, running
on the MI355X through libmi355_dt.so.

Kept from the reference (file:line into the reference):
  class attributes / defaults             MultiObjDetTracker.py:70-120
  __init__(argv={})                       :122-133  (builds KerasYOLO with the 6-key argv)
  load_model()                            :160-189  TimeDistributed(YOLO) -> concat([x_bbox, x_vis])
                                                    -> ConvLSTM2D(512,(3,3)) -> 1x1 conv -> reshape
  load_weights()                          :291-293
  predict(input_paths, output_paths)      :295-315  (broken as written; the
                                                    INTENDED behaviour is implemented)
  train()                                 :221-288  (out of scope: raises)

Track identity is an ADDITION with no reference semantics (`trackid` is written
by the dataset converters and never read): ids come from
the deterministic greedy IoU association specified in DESIGN.md "Track identity".
"""
import os

import numpy as np

import mi355_dt
from models_detection.KerasYOLO import KerasYOLO
from utility.frames import imwrite_bgr, load_frame
from utility.utils import BoundBox, draw_boxes

mot17_class_map = {
    '1': 'Pedestrian', '2': 'Person on vehicle', '3': 'Car', '4': 'Bicycle', '5': 'Motorbike',
    '6': 'Non motorized vehicle', '7': 'Static person', '8': 'Distractor', '9': 'Occluder',
    '10': 'Occluder on the ground', '11': 'Occluder full', '12': 'Reflection'}


class NativeTrackerModel(object):
    """Replaces the Keras `Model([images, true_boxes], [tracking, detection])`
    (MultiObjDetTracker.py:185-188)."""

    def __init__(self, owner):
        self.owner = owner
        self.ctx = owner.detector.model.ctx      # one dt_ctx holds detector + recurrent head
        self.loaded = False

    def set_weights(self, w):
        """w: dict(kernel [3,3,Cb+1024,4U], recurrent [3,3,U,4U], bias [4U],
        out_kernel [1,1,U,Cb], out_bias [Cb]) in Keras layouts
        ('tconv_lstm' and 'timedist_tconv2', MultiObjDetTracker.py:176,182)."""
        units = int(w["recurrent"].shape[2])
        self.ctx.tracker_load(units, w["kernel"], w["recurrent"], w["bias"], w["out_kernel"], w["out_bias"])
        self.loaded = True

    def load_weights(self, path):
        """Keras-style load_weights (MultiObjDetTracker.py:291-293).  Accepts
          * a Keras HDF5 checkpoint (.hdf5 / .h5: whole-model file as ModelCheckpoint writes it, :253-259, or a
            weights-only file): layers 'tconv_lstm' and 'timedist_tconv2' (:176,182); if the file also carries the
            detector (the TimeDistributed copies 'timedist_bbox'), its conv_N / norm_N weights replace the
            detector's.  Read with utility/keras_h5.py (pure Python; h5py is used when importable);
          * an .npz with arrays kernel / recurrent / bias / out_kernel / out_bias (+ optional `darknet` stream)."""
        if path.endswith(".npz"):
            d = np.load(path)
            if "darknet" in d:
                self.owner.detector.model.set_darknet_blob(d["darknet"])
            self.set_weights({k: d[k] for k in ("kernel", "recurrent", "bias", "out_kernel", "out_bias")})
            return
        from utility import keras_h5
        layers = keras_h5.read_keras_weights(path)
        blob = keras_h5.darknet_blob_from_keras(layers)
        if blob is not None:
            self.owner.detector.model.set_darknet_blob(blob)
        self.set_weights(keras_h5.tracker_weights_from_keras(layers))

    def forward(self, frames, want_det=True):
        return self.ctx.track_forward(self.owner.detector.model.to_device(frames), want_det=want_det)

    def predict(self, inputs, batch_size=None):
        """Keras-style: [x (B,T,H,W,3), b] -> [tracking, detection] numpy grids
        (B,T,G,G,BOX,5+CLASS)  (MultiObjDetTracker.py:307)."""
        x = inputs[0] if isinstance(inputs, (list, tuple)) else inputs
        trk, det = self.forward(x, want_det=True)
        return [trk.cpu().numpy(), det.cpu().numpy()]

    def summary(self):
        o = self.owner
        print("NativeTrackerModel: TimeDistributed(YOLOv2) -> ConvLSTM2D(512,3x3) -> Conv2D(%d,1x1); T=%d on %s" % (
            o.BOX * (5 + o.CLASS), o.SEQUENCE_LENGTH, self.ctx.device))


class MultiObjDetTracker(object):
    LABELS_IMAGENET_VIDEO = [
        'n02691156', 'n02419796', 'n02131653', 'n02834778', 'n01503061', 'n02924116', 'n02958343', 'n02402425',
        'n02084071', 'n02121808', 'n02503517', 'n02118333', 'n02510455', 'n02342885', 'n02374451', 'n02129165',
        'n01674464', 'n02484322', 'n03790512', 'n02324045', 'n02509815', 'n02411705', 'n01726692', 'n02355227',
        'n02129604', 'n04468005', 'n01662784', 'n04530566', 'n02062744', 'n02391049']

    LABELS_MOT17 = ['1', '2', '3', '4', '5', '6', '7', '8', '9', '10', '11', '12']

    LABELS = LABELS_MOT17
    IMAGE_H, IMAGE_W = 416, 416
    GRID_H, GRID_W = 13, 13
    BOX = 5
    CLASS = len(LABELS)
    CLASS_WEIGHTS = np.ones(CLASS, dtype='float32')
    OBJ_THRESHOLD = 0.5
    NMS_THRESHOLD = 0.45
    ANCHORS = [0.57273, 0.677385, 1.87446, 2.06253, 3.33843, 5.47434, 7.88282, 3.52778, 9.77052, 9.16828]

    NO_OBJECT_SCALE = 1.0
    OBJECT_SCALE = 5.0
    COORD_SCALE = 1.0
    CLASS_SCALE = 1.0

    BATCH_SIZE = 1
    WARM_UP_BATCHES = 0
    TRUE_BOX_BUFFER = 50

    SEQUENCE_LENGTH = 4
    MAX_BOX_PER_IMAGE = 50

    LOAD_MODEL = True
    INITIAL_EPOCH = 0
    SAVED_MODEL_PATH = 'models/MultiObjDetTracker-CHKPNT-03-0.55.hdf5'

    # build-defined (no reference counterpart): IoU a frame-t box needs with a
    # frame-(t-1) box of the same label to inherit its track id
    ASSOC_THRESHOLD = 0.3

    train_image_folder = 'data/MOT17/MOT17Det/train/'
    train_annot_folder = 'data/MOT17Ann/train/'
    valid_image_folder = 'data/MOT17/MOT17Det/train/'
    valid_annot_folder = 'data/MOT17Ann/val/'

    model = None
    detector = None
    model_detector = None

    def __init__(self, argv={}, detector_weights=None, tracker_weights=None):
        """`argv` as in the reference (its 6 keys are overwritten from the class
        attributes, MultiObjDetTracker.py:123-128).  `detector_weights` /
        `tracker_weights` (additions) supply a darknet stream and a Keras-layout
        weight dict instead of files."""
        argv = dict(argv)
        argv['LABELS'] = self.LABELS
        argv['BATCH_SIZE'] = self.BATCH_SIZE * self.SEQUENCE_LENGTH
        argv['IMAGE_H'] = self.IMAGE_H
        argv['IMAGE_W'] = self.IMAGE_W
        argv['GRID_H'] = self.GRID_H
        argv['GRID_W'] = self.GRID_W
        self.CLASS = len(self.LABELS)

        self.detector = KerasYOLO(argv, weights=detector_weights)
        self._tracker_weights = tracker_weights
        self.load_model()
        if tracker_weights is not None:
            self.model.set_weights(tracker_weights)
        elif self.LOAD_MODEL:
            self.load_weights()

    def load_model(self):
        # the reference's two-output detector sub-model (:162-164) is the same
        # native detector, tapped at conv_23 and conv_feat inside dt_track_forward
        self.model_detector = self.detector.model
        self.model_detector.get_layer('conv_23')
        self.model_detector.get_layer('conv_feat')
        self.model = NativeTrackerModel(self)
        self.model.summary()

    def load_weights(self):
        path = self.SAVED_MODEL_PATH
        if not os.path.isfile(path):
            alt = os.path.splitext(path)[0] + ".npz"
            if os.path.isfile(alt):
                path = alt
            else:
                raise IOError("checkpoint %r not found (the reference loads it in __init__ when LOAD_MODEL, "
                              "MultiObjDetTracker.py:131-133)" % self.SAVED_MODEL_PATH)
        self.model.load_weights(path)
        try:
            self.INITIAL_EPOCH = int(self.SAVED_MODEL_PATH.split('-')[2])
        except (IndexError, ValueError):
            pass

    def train(self):
        raise NotImplementedError("training (custom_loss_*, fit_generator) is outside the MI355X hot path this "
                                  "build covers (SURVEY.md C7/C8); only inference entry points are implemented")

    # ------------------------------------------------------------------
    def track_clips(self, frames, cap=None):
        """Batched device path (addition).  frames [n_clips,T,H,W,3] uint8/float32
        (numpy or device tensor).  Returns device tensors:
          boxes  [n_clips,T,cap,8]  x,y,w,h,conf,label,score,cell (decode order)
          counts [n_clips,T]        boxes per frame
          ids    [n_clips,T,cap]    track ids (-1 in unused slots), per clip from 0
          nids   [n_clips]          ids opened per clip
        One dt_track_forward (YOLOv2 x T, ConvLSTM recurrence, 1x1), one dt_decode
        over all frames, one dt_associate."""
        return self.decode_and_associate(self.model.forward(frames, want_det=False), cap=cap)

    def decode_and_associate(self, trk, cap=None):
        """tracking grid [n_clips,T,G,G,BOX,5+C] (device) -> the track_clips result dict: one dt_decode over all frames
        (OBJ_THRESHOLD / NMS_THRESHOLD may be arrays of one value per frame, clip-major), one dt_associate."""
        ctx = self.model.ctx
        n_clips, T = trk.shape[:2]
        flat = trk.reshape((n_clips * T,) + tuple(trk.shape[2:]))
        if cap is None:
            cap = self.GRID_H * self.GRID_W * self.BOX
        r = ctx.decode(flat, self.OBJ_THRESHOLD, self.NMS_THRESHOLD, self.ANCHORS, len(self.LABELS), cap=cap)
        boxes = r["boxes"].reshape(n_clips, T, cap, mi355_dt.DT_BOX_FLOATS)
        counts = r["counts"].reshape(n_clips, T)
        ids, nids = ctx.associate(boxes, counts, self.ASSOC_THRESHOLD)
        return dict(boxes=boxes, counts=counts, ids=ids, nids=nids, netout=trk)

    def empty_result(self, T, cap=None):
        """result dict for zero clips (a rank that owns no clip in a frame-sharded run)"""
        import torch
        ctx = self.model.ctx
        if cap is None:
            cap = self.GRID_H * self.GRID_W * self.BOX
        z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=ctx.device)
        return dict(boxes=z((0, T, cap, mi355_dt.DT_BOX_FLOATS), torch.float32), counts=z((0, T), torch.int32),
                    ids=z((0, T, cap), torch.int32), nids=z((0,), torch.int32), netout=None)

    @staticmethod
    def boxes_from_result(res, clip=0):
        """Host view of track_clips output for one clip: list over t of BoundBox lists."""
        boxes = res["boxes"][clip].cpu().numpy()
        counts = res["counts"][clip].cpu().numpy()
        ids = res["ids"][clip].cpu().numpy()
        out = []
        for t in range(boxes.shape[0]):
            n = min(int(counts[t]), boxes.shape[1])
            lst = []
            for i in range(n):
                x, y, w, h, c, lab, sc, _ = boxes[t, i]
                bb = BoundBox(x, y, w, h, c, None)
                bb.label = int(lab)
                bb.score = sc
                bb.track_id = int(ids[t, i])
                lst.append(bb)
            out.append(lst)
        return out

    def predict(self, input_paths, output_paths):
        """Intended behaviour of MultiObjDetTracker.py:295-315: read SEQUENCE_LENGTH
        frames, one forward of the tracker, decode the TRACKING grid of every
        time step, draw and save each frame.  Returns the per-frame box lists
        (each box carries .track_id)."""
        assert len(input_paths) == self.SEQUENCE_LENGTH
        x = np.zeros((1, self.SEQUENCE_LENGTH, self.IMAGE_H, self.IMAGE_W, 3), dtype=np.uint8)
        images = []
        for i, input_path in enumerate(input_paths):
            image, resized = load_frame(input_path, self.IMAGE_H, self.IMAGE_W)
            images.append(image)
            x[0, i] = resized
        res = self.track_clips(x)
        per_frame = self.boxes_from_result(res, 0)
        for i, boxes in enumerate(per_frame):
            image = draw_boxes(images[i], boxes, self.LABELS)
            print(len(boxes), 'Bounding Boxes Found')
            print("File Saved to", output_paths[i])
            imwrite_bgr(output_paths[i], image)
        return per_frame

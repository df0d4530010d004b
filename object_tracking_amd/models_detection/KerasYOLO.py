"""KerasYOLO -- the reference's YOLOv2 detector class (models_detection/KerasYOLO.py)
with the same public surface, running on the MI355X through libmi355_dt.so.

Kept from the reference (file:line are into the reference):
  class attributes / defaults            KerasYOLO.py:20-65
  __init__(argv={}) 6-key override       KerasYOLO.py:67-79
  load_model / load_weights              KerasYOLO.py:239-410
  normalize_input                        KerasYOLO.py:412-413
  extract(input_path, layer)             KerasYOLO.py:509-520
  predict(input_path, output_path)       KerasYOLO.py:522-537
  train()                                KerasYOLO.py:447-507   (out of scope: raises)
There is no Keras/TensorFlow here: `self.model` is a thin handle whose
`predict([images, dummy])` runs conv_1..conv_23 as hand-written HIP kernels and
whose `get_layer(name)` exposes the two taps MultiObjDetTracker reaches for
('conv_23', 'conv_feat'; MultiObjDetTracker.py:162-164).
"""
import os

import numpy as np

import mi355_dt
from utility.frames import imwrite_bgr, load_frame
from utility.utils import WeightReader, decode_netout_batch, draw_boxes, normalize


class _Tap(object):
    """Stand-in for a Keras layer handle: carries the tap name only."""

    def __init__(self, name):
        self.name = name
        self.output = name


class NativeDetectorModel(object):
    """Replaces the Keras `Model([input_image, true_boxes], output_det)`
    (KerasYOLO.py:405).  Frames are NHWC uint8 (x/255. fused on device) or
    float32 already normalised."""

    TAPS = ("conv_23", "conv_feat", "act_13")

    def __init__(self, owner):
        self.owner = owner
        self.ctx = mi355_dt.Context()
        self.input = ["input_image", "true_boxes"]
        self.loaded = False

    def configure(self):
        o = self.owner
        self.ctx.detector_config(o.IMAGE_H, o.IMAGE_W, o.BOX, o.CLASS, o.ANCHORS)

    def set_darknet_blob(self, blob):
        used = self.ctx.load_darknet_weights(blob)
        self.loaded = True
        return used

    def load_weights(self, weight_path):
        """`model.load_weights(weight_path)` (KerasYOLO.py:409-410) for: a darknet .weights file (what load_model
        reads, :404), an .npy holding the same float32 stream, or a Keras HDF5 file such as the reference's
        'weights/WEIGHTS_KerasYOLO.h5' checkpoint (:480-486; layers conv_N / norm_N, read by utility/keras_h5.py)."""
        if weight_path.endswith(".npy"):
            blob = np.load(weight_path)
        elif weight_path.endswith((".h5", ".hdf5")):
            from utility import keras_h5
            blob = keras_h5.darknet_blob_from_keras(keras_h5.read_keras_weights(weight_path))
            if blob is None:
                raise IOError("%r holds no detector layers (conv_1 ... conv_23)" % weight_path)
        else:
            blob = WeightReader(weight_path).all_weights
        return self.set_darknet_blob(blob)

    def get_layer(self, name):
        """Any layer name of the reference graph (conv_N, norm_N, leaky_re_lu_N, conv_feat, max_pooling2d_k,
        lambda_1, concatenate_1, ...): validated by the library (dt_detector_extract shape query)."""
        if name not in self.TAPS:
            try:
                self.ctx.layer_shape(name, 1)
            except mi355_dt.NativeError:
                raise ValueError("No such layer: " + name)
        return _Tap(name)

    def to_device(self, frames):
        import torch
        if isinstance(frames, np.ndarray):
            if frames.dtype not in (np.uint8, np.float32):
                frames = frames.astype(np.float32)       # Keras casts float64 input to float32
            frames = torch.from_numpy(np.ascontiguousarray(frames))
        return frames.to(self.ctx.device).contiguous()

    def forward(self, frames, want_feat=False):
        """device-resident batched forward (no host copies)."""
        return self.ctx.detect_forward(self.to_device(frames), want_feat=want_feat)

    def predict(self, inputs, batch_size=None):
        """Keras-style: inputs = [images, dummy_true_boxes] -> netout numpy
        [B,G,G,BOX,5+CLASS]  (KerasYOLO.py:531)."""
        images = inputs[0] if isinstance(inputs, (list, tuple)) else inputs
        return self.forward(images).cpu().numpy()

    def summary(self):
        o = self.owner
        print("NativeDetectorModel: YOLOv2 %dx%dx3 -> %dx%dx%dx%d on %s (23 conv, fp32 results: Winograd + MFMA GEMMs)" % (
            o.IMAGE_H, o.IMAGE_W, o.GRID_H, o.GRID_W, o.BOX, 5 + o.CLASS, self.ctx.device))


class KerasYOLO(object):
    LABELS_COCO = [
        'person', 'bicycle', 'car', 'motorcycle', 'airplane', 'bus', 'train', 'truck', 'boat', 'traffic light',
        'fire hydrant', 'stop sign', 'parking meter', 'bench', 'bird', 'cat', 'dog', 'horse', 'sheep', 'cow',
        'elephant', 'bear', 'zebra', 'giraffe', 'backpack', 'umbrella', 'handbag', 'tie', 'suitcase', 'frisbee',
        'skis', 'snowboard', 'sports ball', 'kite', 'baseball bat', 'baseball glove', 'skateboard', 'surfboard',
        'tennis racket', 'bottle', 'wine glass', 'cup', 'fork', 'knife', 'spoon', 'bowl', 'banana', 'apple',
        'sandwich', 'orange', 'broccoli', 'carrot', 'hot dog', 'pizza', 'donut', 'cake', 'chair', 'couch',
        'potted plant', 'bed', 'dining table', 'toilet', 'tv', 'laptop', 'mouse', 'remote', 'keyboard',
        'cell phone', 'microwave', 'oven', 'toaster', 'sink', 'refrigerator', 'book', 'clock', 'vase', 'scissors',
        'teddy bear', 'hair drier', 'toothbrush']

    LABELS = LABELS_COCO
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

    BATCH_SIZE = 32
    WARM_UP_BATCHES = 0
    TRUE_BOX_BUFFER = 50

    MAX_BOX_PER_IMAGE = 50

    weight_path = 'darknet/yolov2.weights'
    train_image_folder = 'data/coco/train2014/'
    train_annot_folder = 'data/coco/train2014ann/'
    valid_image_folder = 'data/coco/val2014/'
    valid_annot_folder = 'data/coco/val2014ann/'

    model = None

    def __init__(self, argv={}, weights=None):
        """`argv`: the reference's 6-key override dict (KerasYOLO.py:69-77).
        `weights` (addition): a float32 darknet-format stream to use instead of
        reading `self.weight_path` -- needed because no weights ship anywhere."""
        if len(argv) == 6:
            self.LABELS = argv['LABELS']
            self.CLASS = len(self.LABELS)
            self.CLASS_WEIGHTS = np.ones(self.CLASS, dtype='float32')
            self.BATCH_SIZE = argv['BATCH_SIZE']
            self.IMAGE_H = argv['IMAGE_H']
            self.IMAGE_W = argv['IMAGE_W']
            self.GRID_H = argv['GRID_H']
            self.GRID_W = argv['GRID_W']
        if self.GRID_H * 32 != self.IMAGE_H or self.GRID_W * 32 != self.IMAGE_W:
            raise ValueError("GRID must be IMAGE/32 (five 2x2 max-pools, KerasYOLO.py:282-348)")
        self._weights = weights
        self.load_model()

    def load_model(self):
        """Builds the native detector and initialises it from the darknet
        stream exactly as init_weights does (KerasYOLO.py:244-274): skip 4
        floats, then per conv beta,gamma,mean,var,kernel(O,I,H,W)."""
        self.model = NativeDetectorModel(self)
        self.model.configure()
        if self._weights is not None:
            self.model.set_darknet_blob(self._weights)
        else:
            if not os.path.isfile(self.weight_path):
                raise IOError("weight file %r not found (the reference reads it in load_model, "
                              "KerasYOLO.py:404); pass weights=<float32 stream> to use synthetic weights"
                              % self.weight_path)
            self.model.load_weights(self.weight_path)
        self.model.summary()

    def load_weights(self, weight_path):
        self.model.load_weights(weight_path)

    def normalize_input(self, input_instance):
        return normalize(input_instance)

    def train(self):
        raise NotImplementedError("training (loss_fxn, fit_generator) is outside the MI355X hot path this build "
                                  "covers (SURVEY.md C7/C8); only inference entry points are implemented")

    def _frame(self, input_path):
        image, resized = load_frame(input_path, self.IMAGE_H, self.IMAGE_W)
        return image, resized.reshape((1, self.IMAGE_H, self.IMAGE_W, 3))

    def extract(self, input_path, layer):
        """Output of ANY named layer for one image (KerasYOLO.py:509-520): conv_N / norm_N / leaky_re_lu_N
        (alias act_N) / conv_feat / max_pooling2d_k / lambda_1 / concatenate_1 / reshape_1.  The production
        path fuses BatchNorm, LeakyReLU and the pools into the convolutions, so the library re-runs the graph up
        to the layer and executes that one un-fused (dt_detector_extract)."""
        _, frame = self._frame(input_path)
        self.model.get_layer(layer)
        return self.model.ctx.detector_extract(self.model.to_device(frame), layer)[0].cpu().numpy()

    def extract_frames_tap(self, layer, batch):
        ctx = self.model.ctx
        if layer not in NativeDetectorModel.TAPS:
            raise ValueError("No such layer: " + layer)
        return ctx.detector_tap(layer, batch).cpu().numpy()

    def detect(self, frames):
        """Batched device path (addition): frames [B,H,W,3] uint8/float32 (numpy or
        device tensor) -> dict(boxes [B,cap,8], counts [B]) as device tensors."""
        netout = self.model.forward(frames)
        return self.model.ctx.decode(netout, self.OBJ_THRESHOLD, self.NMS_THRESHOLD, self.ANCHORS,
                                     len(self.LABELS))

    def predict(self, input_path, output_path):
        """KerasYOLO.py:522-537; additionally returns the box list."""
        image, frame = self._frame(input_path)
        netout = self.model.predict([frame, None])
        boxes = decode_netout_batch(netout, self.OBJ_THRESHOLD, self.NMS_THRESHOLD, self.ANCHORS,
                                    len(self.LABELS), writeback=True)[0][0]
        image = draw_boxes(image, boxes, self.LABELS)
        print(len(boxes), 'Bounding Boxes Found')
        print("File Saved to", output_path)
        imwrite_bgr(output_path, image)
        return boxes

"""BaseTracker -- configuration holder for the single-object trackers
(models_tracking/BaseTracker.py:12-50 in the reference).

The reference's constructor reads config.json, selects GPUs through
CUDA_VISIBLE_DEVICES and dlopens libdarknet.so / imports caffe to obtain the
detector's feature-layer dimensions (`_w,_h,_c`, BaseTracker.py:53-60).  Those
external detector backends are out of scope; here the
feature layer is this build's own YOLOv2 tap: 'act_13', the 26x26x512 tensor a
stock yolov2.cfg exposes at `fv_layer` 25 (config.json:9).  Only the keys the
TinyTracker forward needs are read; everything else in config.json is ignored.
"""
import json
import os

DEFAULT_CONFIG = {
    "model_tracker": {"name": "TinyTracker", "lstm_units": 512, "sequence_length": 4, "heatmap_size": 32},
    "train": {"pool": "Global", "batch_size": 4},
}


class BaseTracker(object):
    def __init__(self, config=None, feature_dims=(26, 26, 512)):
        if config is None:
            if os.path.isfile("config.json"):
                with open("config.json") as config_buffer:
                    config = json.loads(config_buffer.read())
            else:
                config = DEFAULT_CONFIG
        self.config = config
        self.pool = config.get("train", {}).get("pool", "Global")
        self.batch_size = config.get("train", {}).get("batch_size", 4)
        self.sequence_length = config["model_tracker"]["sequence_length"]
        self._w, self._h, self._c = feature_dims
        self.model_tracker = None

    def train(self):
        raise NotImplementedError("training (fit_generator over BatchSequenceGenerator2) is outside the MI355X "
                                  "hot path this build covers (SURVEY.md C8/C9)")

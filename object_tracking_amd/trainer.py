"""Entry points with the reference's names (trainer.py:8,18,22).

The reference's three functions construct a model and call `.train()`; training
is outside the hot path this build covers, so each function here constructs the
model and runs the inference part of the same entry point.  Weight files are
looked up where the reference looks (darknet/yolov2.weights,
models/MultiObjDetTracker-CHKPNT-*.hdf5|.npz); when absent an IOError explains
what is missing.
"""
import importlib
import json
import os

from models_detection.KerasYOLO import KerasYOLO
from models_tracking.MultiObjDetTracker import MultiObjDetTracker


def single_object_tracking():
    """trainer.py:8-16: instantiate the tracker class named in config.json."""
    name = "TinyTracker"
    if os.path.isfile("config.json"):
        with open("config.json") as config_buffer:
            name = json.loads(config_buffer.read())["model_tracker"]["name"]
    tracker_class = getattr(importlib.import_module("models_tracking." + name), name)
    return tracker_class()


def simult_multi_obj_detection_tracking():
    """trainer.py:18-20."""
    return MultiObjDetTracker()


def keras_yolo_obj_detection():
    """trainer.py:22-30: detect on darknet's sample images."""
    prefix = 'darknet/data/'
    inputs = ['dog.jpg', 'eagle.jpg', 'giraffe.jpg', 'horses.jpg', 'person.jpg']
    model = KerasYOLO()
    results = {}
    for input_instance in inputs:
        if os.path.isfile(prefix + input_instance):
            results[input_instance] = model.predict(prefix + input_instance, input_instance)
    return results


if __name__ == '__main__':
    if not os.path.exists('logs'):
        os.mkdir('logs/')
    if not os.path.exists('models'):
        os.mkdir('models/')
    simult_multi_obj_detection_tracking()

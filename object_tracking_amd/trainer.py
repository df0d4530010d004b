"""Entry points under the reference's names (trainer.py:8, :18, :22 upstream).

Upstream each entry point builds a model and calls `.train()`.  Training is not
part of the path this build covers, so each one builds the corresponding
MI355X-native model and (for the detector) runs the inference half that the
upstream function also contains.  Weight files are looked up exactly where the
reference looks for them; a missing file raises IOError naming it.

    python trainer.py [multi|single|detect]
"""
import importlib
import json
import os
import sys

from models_detection.KerasYOLO import KerasYOLO
from models_tracking.MultiObjDetTracker import MultiObjDetTracker

SAMPLE_DIR = os.path.join('darknet', 'data')
SAMPLE_IMAGES = ('dog.jpg', 'eagle.jpg', 'giraffe.jpg', 'horses.jpg', 'person.jpg')


def _tracker_name(default="TinyTracker"):
    if not os.path.isfile("config.json"):
        return default
    with open("config.json") as fh:
        return json.load(fh)["model_tracker"]["name"]


def single_object_tracking():
    """Instantiate the single-object tracker class config.json names
    (TinyTracker or TinyHeatmapTracker)."""
    name = _tracker_name()
    module = importlib.import_module("models_tracking." + name)
    return getattr(module, name)()


def simult_multi_obj_detection_tracking():
    """Build the simultaneous detect-and-track model (loads its checkpoint)."""
    return MultiObjDetTracker()


def keras_yolo_obj_detection():
    """Detect on darknet's sample images that are present; returns {image: boxes}."""
    detector = KerasYOLO()
    found = {}
    for image_name in SAMPLE_IMAGES:
        source = os.path.join(SAMPLE_DIR, image_name)
        if os.path.isfile(source):
            found[image_name] = detector.predict(source, image_name)
    return found


ENTRY_POINTS = {
    "multi": simult_multi_obj_detection_tracking,
    "single": single_object_tracking,
    "detect": keras_yolo_obj_detection,
}

if __name__ == '__main__':
    for folder in ('logs', 'models'):
        if not os.path.isdir(folder):
            os.makedirs(folder)
    ENTRY_POINTS[sys.argv[1] if len(sys.argv) > 1 else "multi"]()

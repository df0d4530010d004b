"""object_tracking_amd -- MI355X-native detect-and-track hot path of
ktzsh/object-tracking behind the reference's own class surface.

The directory doubles as a drop-in replacement for the reference's repository
root: with it on sys.path, `from models_detection.KerasYOLO import KerasYOLO`,
`from models_tracking.MultiObjDetTracker import MultiObjDetTracker`,
`from utility.utils import decode_netout` and `import trainer` resolve to this
implementation (same names, defaults and call shapes).
Importing this package performs that sys.path insertion and aliases the
sub-packages so that both spellings share one module instance.
"""
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
if _HERE not in sys.path:
    sys.path.insert(0, _HERE)

import mi355_dt  # noqa: E402  (ctypes binding of libmi355_dt.so)
import utility  # noqa: E402
import models_detection  # noqa: E402
import models_tracking  # noqa: E402

for _name, _mod in (("mi355_dt", mi355_dt), ("utility", utility), ("models_detection", models_detection),
                    ("models_tracking", models_tracking)):
    sys.modules[__name__ + "." + _name] = _mod

__all__ = ["mi355_dt", "utility", "models_detection", "models_tracking"]

#!/usr/bin/env python3
"""Convert a Keras HDF5 checkpoint of the reference (models/MultiObjDetTracker-CHKPNT-*.hdf5,
weights/WEIGHTS_KerasYOLO.h5) to the .npz form MultiObjDetTracker.load_weights also accepts:
    kernel, recurrent, bias, out_kernel, out_bias  (+ `darknet`: the detector as a darknet-format float32 stream)
Uses object_tracking_amd/utility/keras_h5.py (h5py when importable, else its own reader).

    python tools/keras_h5_to_npz.py models/MultiObjDetTracker-CHKPNT-03-0.55.hdf5 [out.npz]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "object_tracking_amd"))
from utility import keras_h5  # noqa: E402


def main(src, dst=None):
    dst = dst or os.path.splitext(src)[0] + ".npz"
    layers = keras_h5.read_keras_weights(src)
    out = {}
    if "tconv_lstm" in layers:
        out.update(keras_h5.tracker_weights_from_keras(layers))
    blob = keras_h5.darknet_blob_from_keras(layers)
    if blob is not None:
        out["darknet"] = blob
    if not out:
        raise SystemExit("%s: neither tracker (tconv_lstm) nor detector (conv_N) layers found: %s" % (src, sorted(layers)))
    np.savez(dst, **out)
    print("wrote", dst, {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    if len(sys.argv) < 2:
        raise SystemExit(__doc__)
    main(*sys.argv[1:3])

#!/opt/conda/bin/python3.9
"""Small Keras-layout HDF5 fixtures written by the REAL library (h5py 3.3.0 / libhdf5 1.10.6).

    /opt/conda/bin/python3.9 tools/make_h5_fixtures.py        # build container only (the Anaconda tree has h5py)

They pin object_tracking_amd/utility/keras_h5.py's pure-Python reader against files it did not write:
  tests/golden/h5/keras_tracker_ckpt.hdf5   whole-model file as keras.models.save_model lays it out (root attrs
        keras_version / backend / model_config as variable-length strings; model_weights/<layer>/<layer>/<w>:0;
        a nested TimeDistributed(Model) group; optimizer_weights) -- h5py defaults (superblock v0, v1 headers)
  tests/golden/h5/keras_weights_latest.h5   weights-only file written with libver='latest' (superblock v3,
        v2 object headers, compact link messages), float64 and big-endian datasets
Values come from numpy RandomState(seed) in creation order; tests regenerate them from the seed.
"""
import json
import os

import h5py
import numpy as np

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "h5")
U, CB = 8, 17          # tiny ConvLSTM: 8 units, 17 head channels


def tracker_layers(rs):
    cin = CB + 24
    return [
        ("timedist_bbox", [("conv_1/kernel:0", rs.randn(3, 3, 3, 4)), ("norm_1/gamma:0", rs.rand(4)),
                           ("norm_1/beta:0", rs.randn(4)), ("norm_1/moving_mean:0", rs.randn(4)),
                           ("norm_1/moving_variance:0", rs.rand(4) + 0.5),
                           ("conv_23/kernel:0", rs.randn(1, 1, 4, CB)), ("conv_23/bias:0", rs.randn(CB))]),
        ("tconv_lstm", [("tconv_lstm/kernel:0", rs.randn(3, 3, cin, 4 * U)),
                        ("tconv_lstm/recurrent_kernel:0", rs.randn(3, 3, U, 4 * U)),
                        ("tconv_lstm/bias:0", rs.randn(4 * U))]),
        ("timedist_tconv2", [("timedist_tconv2/kernel:0", rs.randn(1, 1, U, CB)),
                             ("timedist_tconv2/bias:0", rs.randn(CB))]),
        ("detection", []),          # layers without weights have empty groups
    ]


def main():
    os.makedirs(OUT, exist_ok=True)
    rs = np.random.RandomState(77)
    with h5py.File(os.path.join(OUT, "keras_tracker_ckpt.hdf5"), "w") as f:
        f.attrs["keras_version"] = "2.1.5"
        f.attrs["backend"] = "tensorflow"
        f.attrs["model_config"] = json.dumps({"class_name": "Model", "config": {"name": "tracker", "layers": ["x" * 40] * 30}})
        g = f.create_group("model_weights")
        layers = tracker_layers(rs)
        g.attrs["layer_names"] = [n.encode("utf8") for n, _ in layers]
        g.attrs["backend"] = b"tensorflow"
        g.attrs["keras_version"] = b"2.1.5"
        for name, ws in layers:
            lg = g.create_group(name)
            lg.attrs["weight_names"] = [w.encode("utf8") for w, _ in ws]
            for wname, val in ws:
                d = lg.create_dataset(wname, val.shape, dtype="float32")
                d[...] = val
        og = f.create_group("optimizer_weights")
        og.attrs["weight_names"] = [b"Adam/iterations:0"]
        og.create_dataset("Adam/iterations:0", data=np.array(1234, dtype=np.int64))
    rs = np.random.RandomState(78)
    with h5py.File(os.path.join(OUT, "keras_weights_latest.h5"), "w", libver="latest") as f:
        f.attrs["layer_names"] = [b"conv_22", b"norm_22", b"conv_23"]
        f.attrs["backend"] = "tensorflow"
        a = f.create_group("conv_22")
        a.create_dataset("conv_22/kernel:0", data=rs.randn(3, 3, 5, 6).astype(np.float32))
        b = f.create_group("norm_22")
        for n in ("gamma", "beta", "moving_mean", "moving_variance"):
            b.create_dataset("norm_22/%s:0" % n, data=rs.randn(6).astype(np.float64))          # float64 on purpose
        c = f.create_group("conv_23")
        c.create_dataset("conv_23/kernel:0", data=rs.randn(1, 1, 6, 7).astype(">f4"))           # big-endian on purpose
        c.create_dataset("conv_23/bias:0", data=rs.randn(7).astype(np.float32))
    for n in os.listdir(OUT):
        print(n, os.path.getsize(os.path.join(OUT, n)))


if __name__ == "__main__":
    main()

"""CPU suite: the pure-Python HDF5 reader / writer behind load_weights (utility/keras_h5.py;
MultiObjDetTracker.py:291-293, KerasYOLO.py:409-410).  Pinned against files written by the REAL library
(tests/golden/h5/*, h5py 3.3.0 / libhdf5 1.10.6, tools/make_h5_fixtures.py) and, where the build container's
Anaconda interpreter exists, by having that h5py read what this module writes."""
import os
import subprocess

import numpy as np
import pytest

from oracle import oracle as orc
from utility import keras_h5, synth

H5PY_PYTHON = "/opt/conda/bin/python3.9"


def test_reader_on_real_h5py_checkpoint(golden_dir):
    path = os.path.join(golden_dir, "h5", "keras_tracker_ckpt.hdf5")
    f = keras_h5.H5File(path)
    assert f.attrs("/")["keras_version"] == "2.1.5" and f.attrs("/")["backend"] == "tensorflow"      # variable-length strings
    assert f.attrs("/")["model_config"].startswith('{"class_name": "Model"')
    assert list(f.attrs("/model_weights")["layer_names"]) == ["timedist_bbox", "tconv_lstm", "timedist_tconv2", "detection"]
    assert list(f.attrs("/model_weights/tconv_lstm")["weight_names"]) == [
        "tconv_lstm/kernel:0", "tconv_lstm/recurrent_kernel:0", "tconv_lstm/bias:0"]
    tree = f.tree()
    assert int(tree["optimizer_weights"]["Adam"]["iterations:0"]) == 1234
    layers = keras_h5.read_keras_weights(path)
    rs = np.random.RandomState(77)          # the generator's draw order (tools/make_h5_fixtures.py:tracker_layers)
    U, CB = 8, 17
    want = {"timedist_bbox": [("conv_1/kernel", rs.randn(3, 3, 3, 4)), ("norm_1/gamma", rs.rand(4)), ("norm_1/beta", rs.randn(4)),
                              ("norm_1/moving_mean", rs.randn(4)), ("norm_1/moving_variance", rs.rand(4) + 0.5),
                              ("conv_23/kernel", rs.randn(1, 1, 4, CB)), ("conv_23/bias", rs.randn(CB))]}
    want["tconv_lstm"] = [("kernel", rs.randn(3, 3, CB + 24, 4 * U)), ("recurrent_kernel", rs.randn(3, 3, U, 4 * U)),
                          ("bias", rs.randn(4 * U))]
    want["timedist_tconv2"] = [("kernel", rs.randn(1, 1, U, CB)), ("bias", rs.randn(CB))]
    assert sorted(layers) == sorted(want)
    for lname, ws in want.items():
        assert sorted(layers[lname]) == sorted(k for k, _ in ws)
        for k, v in ws:
            assert np.array_equal(layers[lname][k], v.astype(np.float32)), (lname, k)
    tw = keras_h5.tracker_weights_from_keras(layers)
    assert tw["recurrent"].shape == (3, 3, U, 4 * U) and tw["out_bias"].shape == (CB,)


def test_reader_on_latest_libver_file(golden_dir):
    """superblock v3, version-2 object headers, compact link messages, float64 and big-endian datasets"""
    path = os.path.join(golden_dir, "h5", "keras_weights_latest.h5")
    assert list(keras_h5.H5File(path).attrs("/")["layer_names"]) == ["conv_22", "norm_22", "conv_23"]
    layers = keras_h5.read_keras_weights(path)
    rs = np.random.RandomState(78)
    assert np.array_equal(layers["conv_22"]["kernel"], rs.randn(3, 3, 5, 6).astype(np.float32))
    for n in ("gamma", "beta", "moving_mean", "moving_variance"):
        assert np.array_equal(layers["norm_22"][n], rs.randn(6).astype(np.float64).astype(np.float32))
    assert np.array_equal(layers["conv_23"]["kernel"], rs.randn(1, 1, 6, 7).astype(np.float32))
    assert np.array_equal(layers["conv_23"]["bias"], rs.randn(7).astype(np.float32))


def test_writer_reader_round_trip_tracker_checkpoint(tmp_path):
    tw = synth.synth_tracker_weights(12, units=32)
    path = str(tmp_path / "MultiObjDetTracker-CHKPNT-03-0.55.hdf5")
    keras_h5.write_tracker_checkpoint(path, tw)
    got = keras_h5.tracker_weights_from_keras(keras_h5.read_keras_weights(path))
    assert all(np.array_equal(got[k], tw[k]) for k in tw)
    assert keras_h5.darknet_blob_from_keras(keras_h5.read_keras_weights(path)) is None
    with pytest.raises(keras_h5.H5Error):
        keras_h5.H5File(__file__)


def test_detector_layers_round_trip_to_darknet_stream(tmp_path):
    """Keras-layout detector layers -> file -> darknet stream == the stream they were parsed from (C=1 keeps it small
    where it can; the 22 trunk layers are what they are: 200 MB)."""
    C = 1
    blob = synth.synth_darknet_blob(C)
    layers, used = orc.parse_darknet_blob(blob, C)
    assert used == blob.size
    path = str(tmp_path / "ckpt.hdf5")
    keras_h5.write_tracker_checkpoint(path, synth.synth_tracker_weights(C, units=32), darknet_layers=layers)
    back = keras_h5.darknet_blob_from_keras(keras_h5.read_keras_weights(path))
    assert back.size == blob.size and np.array_equal(back[4:], blob[4:])


@pytest.mark.skipif(not os.path.exists(H5PY_PYTHON), reason="needs the build container's Anaconda interpreter (h5py)")
def test_real_h5py_reads_what_the_writer_writes(tmp_path):
    tw = synth.synth_tracker_weights(12, units=32)
    path = str(tmp_path / "w.hdf5")
    keras_h5.write_tracker_checkpoint(path, tw)
    np.savez(str(tmp_path / "want.npz"), **tw)
    code = r'''
import sys, warnings
warnings.filterwarnings("ignore")
import h5py, numpy as np
f = h5py.File(sys.argv[1], "r"); w = np.load(sys.argv[2])
g = f["model_weights"]
assert [n.decode() for n in g.attrs["layer_names"]] == ["tconv_lstm", "timedist_tconv2"]
assert [n.decode() for n in g["tconv_lstm"].attrs["weight_names"]] == ["tconv_lstm/kernel:0", "tconv_lstm/recurrent_kernel:0", "tconv_lstm/bias:0"]
pairs = {"kernel": "tconv_lstm/tconv_lstm/kernel:0", "recurrent": "tconv_lstm/tconv_lstm/recurrent_kernel:0",
         "bias": "tconv_lstm/tconv_lstm/bias:0", "out_kernel": "timedist_tconv2/timedist_tconv2/kernel:0",
         "out_bias": "timedist_tconv2/timedist_tconv2/bias:0"}
for k, p in pairs.items():
    assert g[p].dtype == np.float32 and np.array_equal(g[p][...], w[k]), k
print("H5PY_OK")
'''
    env = {k: v for k, v in os.environ.items() if not k.startswith("PYTHON")}
    r = subprocess.run([H5PY_PYTHON, "-c", code, path, str(tmp_path / "want.npz")], capture_output=True, text=True, env=env)
    assert "H5PY_OK" in r.stdout, r.stdout + r.stderr


def test_h5py_branch_of_the_reader_with_a_stand_in_module(golden_dir, monkeypatch):
    """read_keras_weights prefers h5py when it is importable.  h5py is not in this image, so the branch is driven with a
    minimal stand-in module (File / Group / Dataset with `in`, iteration, item access and visititems -- the calls
    the branch makes) backed by this repository's own reader: both branches must give the same layer dict."""
    import sys
    import types
    from utility import keras_h5

    class Dataset(object):
        def __init__(self, arr):
            self.arr = arr

        def __array__(self, dtype=None, copy=None):
            return np.asarray(self.arr, dtype=dtype)

    class Group(object):
        def __init__(self, tree):
            self.tree = tree

        def __contains__(self, k):
            return k in self.tree

        def __iter__(self):
            return iter(self.tree)

        def __getitem__(self, k):
            v = self.tree[k]
            return Group(v) if isinstance(v, dict) else Dataset(v)

        def visititems(self, fn, prefix=""):
            for k, v in self.tree.items():
                name = prefix + k
                if isinstance(v, dict):
                    fn(name, Group(v))
                    Group(v).visititems(fn, name + "/")
                else:
                    fn(name, Dataset(v))

    class File(Group):
        def __init__(self, path, mode="r"):
            assert mode == "r"
            Group.__init__(self, keras_h5.H5File(path).tree())

        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

    path = os.path.join(golden_dir, "h5", "keras_tracker_ckpt.hdf5")
    monkeypatch.setitem(sys.modules, "h5py", None)               # import h5py -> ImportError: the pure-Python branch
    want = keras_h5.read_keras_weights(path)
    fake = types.ModuleType("h5py")
    fake.File, fake.Dataset, fake.Group = File, Dataset, Group
    monkeypatch.setitem(sys.modules, "h5py", fake)
    got = keras_h5.read_keras_weights(path)
    assert sorted(got) == sorted(want) and len(got) >= 2
    for lname in want:
        assert sorted(got[lname]) == sorted(want[lname])
        for k in want[lname]:
            assert np.array_equal(got[lname][k], want[lname][k])

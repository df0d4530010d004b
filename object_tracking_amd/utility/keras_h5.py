"""Keras HDF5 checkpoints without h5py.

The reference saves / loads its models through Keras' HDF5 files:
  models_tracking/MultiObjDetTracker.py:253-259  ModelCheckpoint('models/MultiObjDetTracker-CHKPNT-{epoch:02d}-{val_loss:.2f}.hdf5')
                                                 (whole-model file: weights live under the group 'model_weights')
  models_tracking/MultiObjDetTracker.py:291-293  self.model.load_weights(self.SAVED_MODEL_PATH)
  models_detection/KerasYOLO.py:480-486,409-410  'weights/WEIGHTS_KerasYOLO.h5' / self.model.load_weights(weight_path)
h5py is not part of this image's Python, so this module carries a small pure-Python reader for the subset of the
HDF5 file format those files use -- superblock v0-v3, object headers v1/v2, old-style groups (symbol table +
v1 B-tree + local heap) and compact new-style groups (link messages), contiguous / compact datasets of
little- or big-endian IEEE floats and integers, fixed- and variable-length string attributes -- and a writer for
the classic layout (superblock v0, v1 object headers, symbol-table groups, contiguous float32 datasets) so that
checkpoints can be exported and the loader tested.  Not supported (clear errors): chunked / compressed datasets,
dense (fractal-heap) link storage.

Verified against the real library: tests/golden/keras_*.hdf5 were written by h5py 3.3.0 / libhdf5 1.10.6
(tools/make_h5_fixtures.py, run with /opt/conda/bin/python3.9 in the build container) and files written by this
module are read back by that h5py in tests/test_keras_h5.py when the interpreter is present.

Layer names (Keras): detector conv_1..conv_23 / norm_1..norm_22 (KerasYOLO.py:279-399); tracker 'tconv_lstm'
(kernel, recurrent_kernel, bias) and 'timedist_tconv2' (kernel, bias) (MultiObjDetTracker.py:176,182); the
TimeDistributed detector copies 'timedist_bbox' / 'timedist_vis' hold the detector's weights (:171).
"""
import struct

import numpy as np

SIG = b"\x89HDF\r\n\x1a\n"
UNDEF = 0xFFFFFFFFFFFFFFFF


class H5Error(IOError):
    pass


# ======================================================================================================
# reader
# ======================================================================================================
class _Dataset(object):
    def __init__(self, f, name, shape, dtype, layout):
        self.f, self.name, self.shape, self.dtype, self.layout = f, name, shape, dtype, layout

    def read(self):
        kind = self.layout[0]
        n = int(np.prod(self.shape)) if len(self.shape) else 1
        nbytes = n * self.dtype.itemsize
        if kind == "compact":
            raw = self.layout[1][:nbytes]
        elif kind == "contiguous":
            addr = self.layout[1]
            if addr == UNDEF:
                raw = b"\0" * nbytes          # never written: fill value 0
            else:
                raw = self.f.buf[self.f.base + addr:self.f.base + addr + nbytes]
        else:
            raise H5Error("dataset %s: %s layout is not supported (Keras weight files are contiguous)" % (self.name, kind))
        if len(raw) < nbytes:
            raise H5Error("dataset %s: file truncated" % self.name)
        return np.frombuffer(raw, dtype=self.dtype, count=n).reshape(self.shape).copy()


class H5File(object):
    """Read-only view of an HDF5 file: `tree()` returns nested dicts {name: dict | ndarray}; `attrs(path)`
    returns the decoded attributes of a group / dataset."""

    def __init__(self, path):
        with open(path, "rb") as fh:
            self.buf = fh.read()
        self.path = path
        pos = 0
        while True:                          # the superblock may sit at 0, 512, 1024, ...
            if self.buf[pos:pos + 8] == SIG:
                break
            pos = 512 if pos == 0 else pos * 2
            if pos + 8 > len(self.buf):
                raise H5Error("%s is not an HDF5 file" % path)
        b = self.buf
        ver = b[pos + 8]
        if ver in (0, 1):
            self.O, self.L = b[pos + 13], b[pos + 14]
            p = pos + 24 + (4 if ver == 1 else 0)
            self.base = self._u(p, self.O) if self._u(p, self.O) != self._undef() else 0
            p += 4 * self.O
            root_hdr = self._u(p + self.O, self.O)       # root symbol table entry: name offset, header address
        elif ver in (2, 3):
            self.O, self.L = b[pos + 9], b[pos + 10]
            p = pos + 12
            self.base = self._u(p, self.O)
            root_hdr = self._u(p + 3 * self.O, self.O)
        else:
            raise H5Error("superblock version %d is not supported" % ver)
        if self.O not in (4, 8) or self.L not in (4, 8):
            raise H5Error("unsupported offset/length sizes %d/%d" % (self.O, self.L))
        self.root = root_hdr
        self._gcol = {}

    # ---- primitives ------------------------------------------------------------------------------
    def _undef(self):
        return (1 << (8 * self.O)) - 1

    def _u(self, p, n):
        return int.from_bytes(self.buf[p:p + n], "little")

    def _abs(self, addr):
        return self.base + addr

    # ---- object headers --------------------------------------------------------------------------
    def _messages(self, addr):
        """[(type, flags, bytes)] of the object header at file address `addr` (v1 or v2)."""
        b = self.buf
        p = self._abs(addr)
        out = []
        if b[p:p + 4] == b"OHDR":
            if b[p + 4] != 2:
                raise H5Error("object header version %d" % b[p + 4])
            flags = b[p + 5]
            q = p + 6
            if flags & 0x20:
                q += 16
            if flags & 0x10:
                q += 4
            nsz = 1 << (flags & 3)
            chunk = self._u(q, nsz)
            q += nsz
            blocks = [(q, q + chunk)]
            track = bool(flags & 0x04)
            while blocks:
                s, e = blocks.pop(0)
                while s + 4 <= e:
                    mtype = b[s]
                    msize = self._u(s + 1, 2)
                    mflags = b[s + 3]
                    s += 4 + (2 if track else 0)
                    data = b[s:s + msize]
                    s += msize
                    if mtype == 0x10:
                        off, ln = self._u_from(data, 0, self.O), self._u_from(data, self.O, self.L)
                        a = self._abs(off)
                        if b[a:a + 4] != b"OCHK":
                            raise H5Error("bad object header continuation")
                        blocks.append((a + 4, a + ln - 4))
                    elif mtype != 0:
                        out.append((mtype, mflags, data))
            return out
        if b[p] != 1:
            raise H5Error("object header version %d at %d" % (b[p], addr))
        nmsg = self._u(p + 2, 2)
        hsize = self._u(p + 8, 4)
        blocks = [(p + 16, p + 16 + hsize)]
        while blocks and len(out) < nmsg + 64:
            s, e = blocks.pop(0)
            while s + 8 <= e:
                mtype = self._u(s, 2)
                msize = self._u(s + 2, 2)
                mflags = b[s + 4]
                data = b[s + 8:s + 8 + msize]
                s += 8 + msize
                if mtype == 0x10:
                    off, ln = self._u_from(data, 0, self.O), self._u_from(data, self.O, self.L)
                    blocks.append((self._abs(off), self._abs(off) + ln))
                elif mtype != 0:
                    out.append((mtype, mflags, data))
        return out

    @staticmethod
    def _u_from(data, p, n):
        return int.from_bytes(data[p:p + n], "little")

    # ---- groups ------------------------------------------------------------------------------------
    def _links(self, msgs):
        """{name: object header address} of a group."""
        links = {}
        for (t, _, d) in msgs:
            if t == 0x11:                                  # symbol table: v1 B-tree + local heap
                btree, heap = self._u_from(d, 0, self.O), self._u_from(d, self.O, self.O)
                hp = self._abs(heap)
                if self.buf[hp:hp + 4] != b"HEAP":
                    raise H5Error("bad local heap")
                seg = self._abs(self._u(hp + 8 + 2 * self.L, self.O))
                self._walk_btree(btree, seg, links)
            elif t == 0x06:                                # link message (compact new-style group)
                ver, fl = d[0], d[1]
                q = 2
                ltype = 0
                if fl & 0x08:
                    ltype = d[q]; q += 1
                if fl & 0x04:
                    q += 8
                if fl & 0x10:
                    q += 1
                nsz = 1 << (fl & 3)
                nlen = self._u_from(d, q, nsz); q += nsz
                name = bytes(d[q:q + nlen]).decode("utf-8"); q += nlen
                if ltype == 0:
                    links[name] = self._u_from(d, q, self.O)
            elif t == 0x02:                                # link info: dense storage?
                fl = d[1]
                q = 2 + (8 if fl & 1 else 0)
                fheap = self._u_from(d, q, self.O)
                if fheap != self._undef():
                    raise H5Error("dense (fractal heap) link storage is not supported; re-save the file with "
                                  "h5py's default libver or convert it with tools/keras_h5_to_npz.py")
        return links

    def _walk_btree(self, addr, heap_seg, links):
        b = self.buf
        p = self._abs(addr)
        if b[p:p + 4] != b"TREE" or b[p + 4] != 0:
            raise H5Error("bad group B-tree node")
        level = b[p + 5]
        n = self._u(p + 6, 2)
        q = p + 8 + 2 * self.O
        for i in range(n):
            child = self._u(q + self.L, self.O)
            q += self.L + self.O
            if level > 0:
                self._walk_btree(child, heap_seg, links)
                continue
            s = self._abs(child)
            if b[s:s + 4] != b"SNOD":
                raise H5Error("bad symbol table node")
            cnt = self._u(s + 6, 2)
            e = s + 8
            for _ in range(cnt):
                noff = self._u(e, self.O)
                ohdr = self._u(e + self.O, self.O)
                end = b.index(b"\0", heap_seg + noff)
                links[b[heap_seg + noff:end].decode("utf-8")] = ohdr
                e += 2 * self.O + 24

    # ---- datatypes / dataspaces / attributes ---------------------------------------------------------
    def _dtype(self, d):
        cls, ver = d[0] & 0x0F, d[0] >> 4
        bits0, size = d[1], self._u_from(d, 4, 4)
        order = ">" if bits0 & 1 else "<"
        if cls == 1:
            if size not in (2, 4, 8):
                raise H5Error("float size %d" % size)
            return np.dtype(order + "f%d" % size), None
        if cls == 0:
            return np.dtype(order + ("i" if bits0 & 0x08 else "u") + "%d" % size), None
        if cls == 3:
            return np.dtype("S%d" % size), None
        if cls == 9:
            return None, ("vlen_str" if (d[1] & 0x0F) == 1 else "vlen")
        return None, "class%d" % cls

    def _shape(self, d):
        ver, rank = d[0], d[1]
        if ver == 1:
            q = 8
        elif ver == 2:
            if d[3] == 2:
                return None                                # null dataspace
            q = 4
        else:
            raise H5Error("dataspace version %d" % ver)
        return tuple(self._u_from(d, q + i * self.L, self.L) for i in range(rank))

    def _gheap_obj(self, addr, index):
        if addr not in self._gcol:
            p = self._abs(addr)
            if self.buf[p:p + 4] != b"GCOL":
                raise H5Error("bad global heap")
            size = self._u(p + 8, self.L)
            objs = {}
            q = p + 8 + self.L
            while q + 8 + self.L <= p + size:
                idx = self._u(q, 2)
                osz = self._u(q + 8, self.L)
                if idx == 0:
                    break
                objs[idx] = self.buf[q + 8 + self.L:q + 8 + self.L + osz]
                q += 8 + self.L + ((osz + 7) // 8) * 8
            self._gcol[addr] = objs
        return self._gcol[addr].get(index, b"")

    def _attribute(self, d):
        ver = d[0]
        nsz, tsz, ssz = self._u_from(d, 2, 2), self._u_from(d, 4, 2), self._u_from(d, 6, 2)
        q = 8 + (1 if ver == 3 else 0)
        pad = (lambda n: (n + 7) // 8 * 8) if ver == 1 else (lambda n: n)
        name = bytes(d[q:q + nsz]).split(b"\0")[0].decode("utf-8"); q += pad(nsz)
        tmsg = d[q:q + tsz]; q += pad(tsz)
        smsg = d[q:q + ssz]; q += pad(ssz)
        dtype, special = self._dtype(tmsg)
        shape = self._shape(smsg)
        if shape is None:
            return name, None
        n = int(np.prod(shape)) if len(shape) else 1
        if special == "vlen_str":
            vals = []
            for i in range(n):
                e = q + i * (4 + self.O + 4)
                ln = self._u_from(d, e, 4)
                ga = self._u_from(d, e + 4, self.O)
                gi = self._u_from(d, e + 4 + self.O, 4)
                vals.append(bytes(self._gheap_obj(ga, gi))[:ln].decode("utf-8", "replace") if ln else "")
            return name, (vals[0] if not shape else np.array(vals, dtype=object).reshape(shape))
        if dtype is None:
            return name, None
        arr = np.frombuffer(bytes(d[q:q + n * dtype.itemsize]), dtype=dtype, count=n).reshape(shape)
        if dtype.kind == "S":
            arr = np.array([v.split(b"\0")[0].decode("utf-8", "replace") for v in arr.ravel()], dtype=object).reshape(shape)
        return name, (arr[()] if not shape else arr)

    # ---- public --------------------------------------------------------------------------------------
    def _node(self, addr, name):
        msgs = self._messages(addr)
        types = set(t for (t, _, _) in msgs)
        if 0x11 in types or 0x06 in types or 0x02 in types and 0x08 not in types:
            return {k: self._node(a, name + "/" + k) for k, a in sorted(self._links(msgs).items())}
        shape = dtype = layout = None
        for (t, _, d) in msgs:
            if t == 0x01:
                shape = self._shape(d)
            elif t == 0x03:
                dtype, special = self._dtype(d)
                if dtype is None:
                    raise H5Error("dataset %s: datatype %s is not supported" % (name, special))
            elif t == 0x08:
                ver = d[0]
                if ver in (1, 2):
                    rank, cls = d[1], d[2]
                    q = 8
                    if cls == 1:
                        layout = ("contiguous", self._u_from(d, q, self.O))
                    elif cls == 0:
                        q += 4 * rank
                        sz = self._u_from(d, q, 4)
                        layout = ("compact", bytes(d[q + 4:q + 4 + sz]))
                    else:
                        layout = ("chunked",)
                elif ver in (3, 4):
                    cls = d[1]
                    if cls == 0:
                        sz = self._u_from(d, 2, 2)
                        layout = ("compact", bytes(d[4:4 + sz]))
                    elif cls == 1:
                        layout = ("contiguous", self._u_from(d, 2, self.O))
                    else:
                        layout = ("chunked",)
                else:
                    raise H5Error("layout message version %d" % ver)
        if shape is None or dtype is None or layout is None:
            if not msgs or types <= {0x0C, 0x12, 0x0A}:
                return {}
            raise H5Error("object %s is neither a group nor a simple dataset" % name)
        return _Dataset(self, name, shape, dtype, layout).read()

    def tree(self):
        return self._node(self.root, "")

    def _find(self, path):
        addr = self.root
        for part in [p for p in path.split("/") if p]:
            links = self._links(self._messages(addr))
            if part not in links:
                raise KeyError(path)
            addr = links[part]
        return addr

    def attrs(self, path="/"):
        out = {}
        for (t, _, d) in self._messages(self._find(path)):
            if t == 0x0C:
                k, v = self._attribute(d)
                out[k] = v
        return out


def _flatten(tree, prefix=""):
    out = {}
    for k, v in tree.items():
        if isinstance(v, dict):
            out.update(_flatten(v, prefix + k + "/"))
        else:
            out[prefix + k] = v
    return out


def read_keras_weights(path):
    """{layer_name: {weight_name: float32 ndarray}} of a Keras weight file (save_weights) or whole-model file
    (model.save / ModelCheckpoint: weights under 'model_weights').  weight_name is the last path component without
    the ':0' suffix ('kernel', 'recurrent_kernel', 'bias', 'gamma', 'beta', 'moving_mean', 'moving_variance').
    Nested models (TimeDistributed(Model)) appear under their own layer name with 'inner_layer/weight' keys."""
    try:
        import h5py                      # the real thing where it exists
    except ImportError:
        h5py = None
    if h5py is not None:
        with h5py.File(path, "r") as f:
            g = f["model_weights"] if "model_weights" in f else f
            layers = {}
            for lname in g:
                flat = {}
                g[lname].visititems(lambda n, o: flat.__setitem__(n, np.asarray(o)) if isinstance(o, h5py.Dataset) else None)
                layers[lname] = flat
    else:
        tree = H5File(path).tree()
        g = tree.get("model_weights", tree)
        if not isinstance(g, dict):
            raise H5Error("%s: no layer groups found" % path)
        layers = {k: _flatten(v) for k, v in g.items() if isinstance(v, dict)}
    out = {}
    for lname, flat in layers.items():
        ws = {}
        for full, arr in flat.items():
            parts = full.split("/")
            leaf = parts[-1].split(":")[0]
            inner = parts[-2] if len(parts) >= 2 and parts[-2] != lname else None
            ws[(inner + "/" + leaf) if inner else leaf] = np.asarray(arr, dtype=np.float32)
        if ws:
            out[lname] = ws
    return out


def tracker_weights_from_keras(layers):
    """Keras layer dict (read_keras_weights) -> the tracker head in this build's dict form
    (MultiObjDetTracker.py:176 'tconv_lstm', :182 'timedist_tconv2')."""
    try:
        lstm, head = layers["tconv_lstm"], layers["timedist_tconv2"]
    except KeyError as e:
        raise H5Error("checkpoint has no layer %s (layers: %s)" % (e, sorted(layers)))

    def pick(d, *names):
        for n in names:
            for k, v in d.items():
                if k == n or k.endswith("/" + n):
                    return v
        raise H5Error("weight %s not found among %s" % (names[0], sorted(d)))
    return dict(kernel=pick(lstm, "kernel"), recurrent=pick(lstm, "recurrent_kernel"), bias=pick(lstm, "bias"),
                out_kernel=pick(head, "kernel"), out_bias=pick(head, "bias"))


def darknet_blob_from_keras(layers, nb_class_hint=None):
    """Keras detector layers conv_1..conv_23 / norm_1..norm_22 (directly, or inside the TimeDistributed copy
    'timedist_bbox' of a tracker checkpoint) -> the darknet-format float32 stream init_weights consumes
    (KerasYOLO.py:244-274: beta, gamma, mean, var, kernel (O,I,H,W); conv_23: bias, kernel), 4-float header included.
    Returns None when the file holds no detector."""
    def find(lname, wname):
        if lname in layers and wname in layers[lname]:
            return layers[lname][wname]
        for outer in ("timedist_bbox", "timedist_vis"):
            key = lname + "/" + wname
            if outer in layers and key in layers[outer]:
                return layers[outer][key]
        return None
    if find("conv_1", "kernel") is None:
        return None
    parts = [np.zeros(4, dtype=np.float32)]
    order = list(range(1, 21)) + [21, 22]
    for i in order:
        k = find("conv_%d" % i, "kernel")
        vals = [find("norm_%d" % i, n) for n in ("beta", "gamma", "moving_mean", "moving_variance")]
        if k is None or any(v is None for v in vals):
            raise H5Error("detector layer %d incomplete in checkpoint" % i)
        parts += [v.ravel() for v in vals] + [np.ascontiguousarray(k.transpose(3, 2, 0, 1)).ravel()]     # HWIO -> OIHW
    k, b = find("conv_23", "kernel"), find("conv_23", "bias")
    if k is None or b is None:
        raise H5Error("conv_23 incomplete in checkpoint")
    parts += [b.ravel(), np.ascontiguousarray(k.transpose(3, 2, 0, 1)).ravel()]
    return np.concatenate(parts).astype(np.float32)


# ======================================================================================================
# writer (classic layout: superblock v0, v1 object headers, symbol-table groups, contiguous datasets)
# ======================================================================================================
class _Writer(object):
    def __init__(self):
        self.buf = bytearray()

    def alloc(self, n, align=8):
        while len(self.buf) % align:
            self.buf.append(0)
        a = len(self.buf)
        self.buf.extend(b"\0" * n)
        return a

    def put(self, addr, data):
        self.buf[addr:addr + len(data)] = data


def _msg(mtype, data, flags=0):
    data = bytes(data)
    pad = (-len(data)) % 8
    return struct.pack("<HHB3x", mtype, len(data) + pad, flags) + data + b"\0" * pad


def _dtype_msg(dt):
    dt = np.dtype(dt)
    if dt == np.float32:
        return struct.pack("<B3BI", 0x11, 0x20, 0x1F, 0x00, 4) + struct.pack("<HHBBBBI", 0, 32, 23, 8, 0, 23, 127)
    if dt == np.float64:
        return struct.pack("<B3BI", 0x11, 0x20, 0x3F, 0x00, 8) + struct.pack("<HHBBBBI", 0, 64, 52, 11, 0, 52, 1023)
    if dt.kind == "S":
        return struct.pack("<B3BI", 0x13, 0x00, 0x00, 0x00, dt.itemsize)        # null-terminated ASCII
    if dt.kind in "iu":
        return struct.pack("<B3BI", 0x10, 0x08 if dt.kind == "i" else 0x00, 0, 0, dt.itemsize) + \
            struct.pack("<HH", 0, 8 * dt.itemsize)
    raise H5Error("cannot write dtype %s" % dt)


def _space_msg(shape):
    if len(shape) == 0:
        return struct.pack("<BBB5x", 1, 0, 0)
    return struct.pack("<BBB5x", 1, len(shape), 0) + b"".join(struct.pack("<Q", int(s)) for s in shape)


def _attr_msg(name, value):
    if isinstance(value, str):
        value = np.array(value.encode("utf-8") + b"\0")                     # scalar fixed-length string
    elif isinstance(value, (list, tuple)) and value and isinstance(value[0], (str, bytes)):
        enc = [v.encode("utf-8") if isinstance(v, str) else v for v in value]
        value = np.array(enc, dtype="S%d" % (max(len(v) for v in enc) + 1))
    value = np.ascontiguousarray(value)
    nm = name.encode("utf-8") + b"\0"
    t, s = _dtype_msg(value.dtype), _space_msg(value.shape)
    pad = lambda b: b + b"\0" * ((-len(b)) % 8)
    return _msg(0x0C, struct.pack("<BxHHH", 1, len(nm), len(t), len(s)) + pad(nm) + pad(t) + pad(s) + value.tobytes())


def write_h5(path, tree, attrs=None):
    """tree: nested dict {name: dict | ndarray}; attrs: {group_or_dataset_path: {attr_name: value}} with values
    str, list of str, or numeric arrays.  Classic HDF5 layout readable by libhdf5 / h5py."""
    attrs = attrs or {}
    w = _Writer()
    O = 8
    leaf_k = 4

    def count(t):
        n = len(t)
        for v in t.values():
            if isinstance(v, dict):
                n = max(n, count(v))
        return n
    leaf_k = max(4, (count(tree) + 1) // 2)
    int_k = 16
    w.alloc(96)                                   # superblock v0 with O = L = 8 is 96 bytes
    snod_size = 8 + 2 * leaf_k * 40
    tree_size = 8 + 2 * O + (2 * int_k + 1) * 8 + 2 * int_k * O

    def header(messages):
        body = b"".join(messages)
        a = w.alloc(16 + len(body))
        w.put(a, struct.pack("<BxHII4x", 1, len(messages), 1, len(body)) + body)
        return a

    def write_dataset(arr, path):
        arr = np.ascontiguousarray(arr)
        if arr.dtype not in (np.float32, np.float64) and arr.dtype.kind not in "iuS":
            arr = arr.astype(np.float32)
        da = w.alloc(max(arr.nbytes, 1))
        w.put(da, arr.tobytes())
        msgs = [_msg(0x01, _space_msg(arr.shape)), _msg(0x03, _dtype_msg(arr.dtype), flags=1),
                _msg(0x05, struct.pack("<BBBB", 2, 2, 2, 0)),                     # fill value v2: late alloc, never written, undefined
                _msg(0x08, struct.pack("<BBQQ", 3, 1, da, arr.nbytes))]
        msgs += [_attr_msg(k, v) for k, v in attrs.get(path, {}).items()]
        return header(msgs)

    def write_group(t, path):
        """returns (object header address, B-tree address, local heap address)"""
        names = sorted(t, key=lambda s: s.encode("utf-8"))
        children = {}
        for n in names:
            v = t[n]
            p = path.rstrip("/") + "/" + n
            children[n] = write_group(v, p)[0] if isinstance(v, dict) else write_dataset(v, p)
        # local heap: offset 0 holds the empty string (the B-tree's first key), names follow, 8-byte aligned
        heap = bytearray(b"\0" * 8)
        offs = {}
        for n in names:
            offs[n] = len(heap)
            e = n.encode("utf-8") + b"\0"
            heap += e + b"\0" * ((-len(e)) % 8)
        free_off = len(heap)
        heap += struct.pack("<QQ", 1, 16)           # one free block: next = 1 (none), size = 16
        seg = w.alloc(len(heap))
        w.put(seg, bytes(heap))
        hp = w.alloc(8 + 3 * 8)
        w.put(hp, b"HEAP" + struct.pack("<B3xQQQ", 0, len(heap), free_off, seg))
        snod = w.alloc(snod_size)
        ent = b"".join(struct.pack("<QQII16x", offs[n], children[n], 0, 0) for n in names)
        w.put(snod, b"SNOD" + struct.pack("<BxH", 1, len(names)) + ent)
        bt = w.alloc(tree_size)
        last = offs[names[-1]] if names else 0
        w.put(bt, b"TREE" + struct.pack("<BBHQQ", 0, 0, 1 if names else 0, UNDEF, UNDEF) + struct.pack("<QQQ", 0, snod, last))
        msgs = [_msg(0x11, struct.pack("<QQ", bt, hp))]
        msgs += [_attr_msg(k, v) for k, v in attrs.get(path if path else "/", {}).items()]
        return header(msgs), bt, hp

    root_hdr, root_bt, root_hp = write_group(tree, "")
    eof = len(w.buf)
    sb = SIG + struct.pack("<BBBxBBBxHHI", 0, 0, 0, 0, 8, 8, leaf_k, int_k, 0)
    sb += struct.pack("<QQQQ", 0, UNDEF, eof, UNDEF)
    sb += struct.pack("<QQII", 0, root_hdr, 1, 0) + struct.pack("<QQ", root_bt, root_hp)
    assert len(sb) == 96
    w.put(0, sb)
    with open(path, "wb") as fh:
        fh.write(bytes(w.buf))


def write_keras_weights(path, layers, order=None, whole_model=True):
    """layers: {layer_name: [(weight_name, ndarray), ...]} -> a file laid out the way Keras 2.x writes it:
    <root or model_weights>/<layer>/<layer>/<weight>:0 with the 'layer_names' / 'weight_names' / 'backend' /
    'keras_version' attributes Keras' load_weights reads (keras/engine/topology.py, save_weights_to_hdf5_group)."""
    order = list(order) if order is not None else list(layers)
    tree, attrs = {}, {}
    prefix = "/model_weights" if whole_model else ""
    for lname in order:
        g = {}
        wnames = []
        for (wname, arr) in layers[lname]:
            full = "%s/%s:0" % (lname, wname)
            wnames.append(full)
            node = g
            parts = full.split("/")
            for part in parts[:-1]:
                node = node.setdefault(part, {})
            node[parts[-1]] = np.asarray(arr, dtype=np.float32)
        tree[lname] = g
        attrs["%s/%s" % (prefix, lname)] = {"weight_names": wnames}
    top = {"layer_names": order, "backend": "tensorflow", "keras_version": "2.1.5"}
    if whole_model:
        attrs["/model_weights"] = top
        attrs["/"] = {"keras_version": "2.1.5", "backend": "tensorflow"}
        tree = {"model_weights": tree}
    else:
        attrs["/"] = top
    write_h5(path, tree, attrs)


def write_tracker_checkpoint(path, tw, darknet_layers=None):
    """The tracker head (and optionally the detector's Keras-layout layers) as a whole-model Keras file like
    'models/MultiObjDetTracker-CHKPNT-03-0.55.hdf5' (MultiObjDetTracker.py:253-259)."""
    layers = {
        "tconv_lstm": [("kernel", tw["kernel"]), ("recurrent_kernel", tw["recurrent"]), ("bias", tw["bias"])],
        "timedist_tconv2": [("kernel", tw["out_kernel"]), ("bias", tw["out_bias"])],
    }
    order = ["tconv_lstm", "timedist_tconv2"]
    if darknet_layers:
        det = []
        for i in sorted(darknet_layers):
            L = darknet_layers[i]
            det.append(("conv_%d/kernel" % i, L["kernel"]))
            if "bias" in L:
                det.append(("conv_%d/bias" % i, L["bias"]))
            else:
                det += [("norm_%d/gamma" % i, L["gamma"]), ("norm_%d/beta" % i, L["beta"]),
                        ("norm_%d/moving_mean" % i, L["mean"]), ("norm_%d/moving_variance" % i, L["var"])]
        layers["timedist_bbox"] = det
        order = ["timedist_bbox"] + order
    write_keras_weights(path, layers, order, whole_model=True)

"""Data-side callers of the hot path: annotation parsing, stride-1 sequence
windows and the YOLO target encoding of the reference's utility/preprocessing.py.

    parse_annotation                            preprocessing.py:12-77    host (XML walk)
    create_sequences_from_parsed_annotations    preprocessing.py:79-89    host (index logic)
    BatchGenerator / BatchSequenceGenerator1    preprocessing.py:195-371  frames: device resize
                                                                          targets: dt_encode_targets

Only the deterministic part is here.  The image augmentation pipeline (imgaug, :106-134) is
outside the hot path; the *coordinate* side of an augmentation draw (scale, offx, offy, flip)
is supported by the kernel so a caller that augments frames itself can still encode on device.
The generators therefore accept augment=False only.  They are plain indexable objects
(__len__/__getitem__/on_epoch_end), i.e. what keras.utils.Sequence asks of a subclass.

Where the reference is broken (BatchGenerator.__getitem__ clobbers y_batch and
reads an undefined name) the generators implement the evident intent: one (x, b, y) row per
instance.
"""
import os
import xml.etree.ElementTree as ET

import numpy as np

import mi355_dt
from utility.frames import imread_bgr

_BOX_TAGS = ("xmin", "ymin", "xmax", "ymax")


def _annotation_files(ann_dir):
    found = []
    for root, _dirs, files in os.walk(ann_dir):
        found.extend(root + "/" + f for f in sorted(files) if f.endswith(".xml"))
    return found


def _read_object(node, labels, record, seen):
    """One <object>/<part> element.  Mirrors the reference's ordering rules: the object joins the
    image as soon as its <name> is read; a name outside `labels` stops the walk over this
    element's children (so a following <bndbox> is never read)."""
    obj = {}
    for child in list(node):
        if "name" in child.tag:
            obj["name"] = child.text
            if labels and obj["name"] not in labels:
                return
            record["object"].append(obj)
            seen[obj["name"]] = seen.get(obj["name"], 0) + 1
        if "bndbox" in child.tag:
            for dim in list(child):
                for key in _BOX_TAGS:
                    if key in dim.tag:
                        obj[key] = int(round(float(dim.text)))


def parse_annotation(ann_dir, img_dir, labels=[]):
    """preprocessing.py:12-77: Pascal-VOC style XML tree -> (list of image records, label histogram).
    A record is {'object': [{'name','xmin','ymin','xmax','ymax'}...], 'folder', 'filename',
    'width', 'height'}; images without a kept object are dropped.  Substring tag matching, the
    '.JPEG' default extension and the walk order are the reference's."""
    images, seen = [], {}
    for path in _annotation_files(ann_dir):
        record = {"object": []}
        folder = ""
        for elem in ET.parse(path).iter():
            tag = elem.tag
            if "folder" in tag:
                folder = elem.text + "/"
                record["folder"] = folder
            if "filename" in tag:
                name = img_dir + folder + elem.text
                record["filename"] = name if "." in name else name + ".JPEG"
            if "width" in tag:
                record["width"] = int(elem.text)
            if "height" in tag:
                record["height"] = int(elem.text)
            if "object" in tag or "part" in tag:
                _read_object(elem, labels, record, seen)
        if record["object"]:
            images.append(record)
    return images, seen


def sequence_window_starts(folders, seq_len):
    """Start index of every window create_sequences_from_parsed_annotations emits, in order.
    Reference behaviour kept as is: when a window would straddle two folders its start slides
    forward to the next folder -- and the same slid window is emitted again for every loop index
    that lands before it; sliding off the end raises IndexError (preprocessing.py:83-87)."""
    starts = []
    last = len(folders) - seq_len
    for first in range(last + 1):
        at = first
        while folders[at] != folders[at + seq_len - 1]:
            at += 1
        starts.append(at)
    return starts


def create_sequences_from_parsed_annotations(parsed_data, SEQUENCE_LENGTH):
    """preprocessing.py:79-89."""
    folders = [rec["folder"] for rec in parsed_data]
    return [parsed_data[s:s + SEQUENCE_LENGTH] for s in sequence_window_starts(folders, SEQUENCE_LENGTH)]


def pack_objects(instances, labels, cap=None):
    """Image records -> the int32 arrays dt_encode_targets takes:
    objs [n,cap,5] (xmin, ymin, xmax, ymax, LABELS index or -1), counts [n], dims [n,2] (w,h)."""
    n = len(instances)
    cap = cap if cap is not None else max(1, max((len(r["object"]) for r in instances), default=1))
    objs = np.full((n, cap, 5), -1, dtype=np.int32)
    counts = np.zeros(n, dtype=np.int32)
    dims = np.zeros((n, 2), dtype=np.int32)
    index = {name: i for i, name in reversed(list(enumerate(labels)))}     # list.index: first match
    for i, rec in enumerate(instances):
        if len(rec["object"]) > cap:
            raise ValueError("record %d has %d objects, cap is %d" % (i, len(rec["object"]), cap))
        counts[i] = len(rec["object"])
        dims[i] = (rec.get("width", 0), rec.get("height", 0))     # absent in records without a <size> block: the
        # generators then pass the decoded image's dims (what the reference always uses, preprocessing.py:144)
        for k, o in enumerate(rec["object"]):
            objs[i, k] = (o["xmin"], o["ymin"], o["xmax"], o["ymax"], index.get(o["name"], -1))
    return objs, counts, dims


class BatchGenerator(object):
    """preprocessing.py:195-324 with augment=False: x = resized RGB frames (through `norm` if given),
    b = true boxes (B,1,1,1,TRUE_BOX_BUFFER,4), y = (B,GRID_H,GRID_W,BOX,4+1+CLASS); float64 numpy
    like the reference.  `config` keys: IMAGE_H IMAGE_W GRID_H GRID_W BOX CLASS LABELS ANCHORS
    BATCH_SIZE TRUE_BOX_BUFFER.  Record sizes come from the image file when 'width'/'height' are
    absent (the reference always reads them from the decoded image, :144)."""

    def __init__(self, images, config, shuffle=True, augment=False, norm=None, ctx=None):
        if augment:
            raise NotImplementedError("image augmentation (imgaug pipeline, preprocessing.py:106-166) is outside "
                                      "the MI355X hot path; pass augment=False")
        self.images = images
        self.config = config
        self.shuffle = shuffle
        self.augment = False
        self.norm = norm
        self.counter = 0
        self.ctx = ctx if ctx is not None else mi355_dt.default_context()
        if shuffle:
            np.random.shuffle(self.images)

    def __len__(self):
        return int(np.ceil(float(len(self.images)) / self.config["BATCH_SIZE"]))

    def on_epoch_end(self):
        if self.shuffle:
            np.random.shuffle(self.images)
        self.counter = 0

    def _bounds(self, idx):
        size = self.config["BATCH_SIZE"]
        hi = min((idx + 1) * size, len(self.images))
        lo = hi - size if (idx + 1) * size > len(self.images) else idx * size
        return max(lo, 0), hi      # fewer items than one batch: a short batch (the reference leaves the rest zero)

    def load_frames(self, instances):
        """cv2.imread -> cv2.resize -> [:,:,::-1] (preprocessing.py:143,168-169): uint8 RGB [n,H,W,3]
        (resize on the device, csrc/ingest.hip) plus the decoded (w,h) of every file."""
        import torch
        H, W = self.config["IMAGE_H"], self.config["IMAGE_W"]
        out = np.empty((len(instances), H, W, 3), dtype=np.uint8)
        dims = np.zeros((len(instances), 2), dtype=np.int32)
        for i, rec in enumerate(instances):
            img = imread_bgr(rec["filename"])
            dims[i] = (img.shape[1], img.shape[0])
            d = torch.from_numpy(img[None]).to(self.ctx.device)
            out[i] = self.ctx.ingest_resize(d, H, W)[0].cpu().numpy()[:, :, ::-1]
        return out, dims

    def encode(self, instances, dims=None):
        """Targets for a list of image records -> (y [n,GH,GW,BOX,5+C], b [n,TBB,4]) float64 numpy."""
        import torch
        c = self.config
        objs, counts, rec_dims = pack_objects(instances, c["LABELS"])
        dims = rec_dims if dims is None else dims
        dev = self.ctx.device
        y, b = self.ctx.encode_targets(torch.from_numpy(objs).to(dev), torch.from_numpy(counts).to(dev),
                                       torch.from_numpy(np.ascontiguousarray(dims, dtype=np.int32)).to(dev), None,
                                       c["GRID_H"], c["GRID_W"], c["BOX"], c["CLASS"], c["IMAGE_H"], c["IMAGE_W"],
                                       c["TRUE_BOX_BUFFER"], c["ANCHORS"])
        return y.cpu().numpy(), b.cpu().numpy()

    def output_from_instances(self, instances):
        frames, dims = self.load_frames(instances)
        y, b = self.encode(instances, dims)
        x = self.norm(frames) if self.norm is not None else frames.astype(np.float64)
        return x, b.reshape(len(instances), 1, 1, 1, self.config["TRUE_BOX_BUFFER"], 4), y

    def __getitem__(self, idx):
        lo, hi = self._bounds(idx)
        x, b, y = self.output_from_instances(self.images[lo:hi])
        self.counter += 1
        return [x, b], y


class BatchSequenceGenerator1(BatchGenerator):
    """preprocessing.py:326-371: items are stride-1 windows of SEQUENCE_LENGTH records from one folder;
    returns [x (B,T,H,W,3), b (B,T,1,1,1,TBB,4)], [y, y] with y (B,T,GH,GW,BOX,5+C)."""

    def __init__(self, images, config, shuffle=True, augment=False, norm=None, ctx=None):
        windows = create_sequences_from_parsed_annotations(images, config["SEQUENCE_LENGTH"])
        print("Samples/Sequences %d %d" % (len(images), len(windows)))
        super(BatchSequenceGenerator1, self).__init__(windows, config, shuffle=shuffle, augment=augment, norm=norm,
                                                      ctx=ctx)

    def __getitem__(self, idx):
        lo, hi = self._bounds(idx)
        T = self.config["SEQUENCE_LENGTH"]
        flat = [rec for window in self.images[lo:hi] for rec in window]
        x, b, y = self.output_from_instances(flat)
        n = len(self.images[lo:hi])
        x = x.reshape((n, T) + x.shape[1:])
        b = b.reshape((n, T) + b.shape[1:])
        y = y.reshape((n, T) + y.shape[1:])
        return [x, b], [y, y]

"""Synthetic weights and frames.  No weights ship with the
reference and none can be downloaded here, so tests and bench.py use a
darknet-FORMAT blob with random contents of the right shapes:
  kernels He-normal (sigma = sqrt(2/K)), gamma ~ U(0.5,1.5), beta ~ N(0,0.1),
  mean ~ N(0,0.1), var ~ U(0.5,1.5)  -- keeps activations O(1) through 23 layers.
File layout is exactly what WeightReader/init_weights consume
(utility/utils.py:138-148, models_detection/KerasYOLO.py:244-274)."""
import numpy as np

# (idx, k, cin, cout) in FILE order: conv_1..conv_20, conv_21 (skip 1x1), conv_22
FILE_ORDER = [
    (1, 3, 3, 32), (2, 3, 32, 64), (3, 3, 64, 128), (4, 1, 128, 64), (5, 3, 64, 128), (6, 3, 128, 256),
    (7, 1, 256, 128), (8, 3, 128, 256), (9, 3, 256, 512), (10, 1, 512, 256), (11, 3, 256, 512),
    (12, 1, 512, 256), (13, 3, 256, 512), (14, 3, 512, 1024), (15, 1, 1024, 512), (16, 3, 512, 1024),
    (17, 1, 1024, 512), (18, 3, 512, 1024), (19, 3, 1024, 1024), (20, 3, 1024, 1024), (21, 1, 512, 64),
    (22, 3, 1280, 1024),
]


def darknet_blob_size(nb_class, nb_box=5):
    n = 4
    for (_, k, ci, co) in FILE_ORDER:
        n += 4 * co + co * ci * k * k
    co = nb_box * (5 + nb_class)
    return n + co + co * 1024


def synth_darknet_blob(nb_class, nb_box=5, seed=1234, head_std=None):
    """float32 array = the contents of a yolov2-style .weights file (4-float
    header included).  `head_std`: std of the conv_23 kernel (default He)."""
    rs = np.random.RandomState(seed)
    parts = [np.zeros(4, dtype=np.float32)]
    for (_, k, ci, co) in FILE_ORDER:
        K = k * k * ci
        parts.append((rs.randn(co) * 0.1).astype(np.float32))              # beta
        parts.append(rs.uniform(0.5, 1.5, co).astype(np.float32))          # gamma
        parts.append((rs.randn(co) * 0.1).astype(np.float32))              # mean
        parts.append(rs.uniform(0.5, 1.5, co).astype(np.float32))          # var
        parts.append((rs.randn(co * ci * k * k) * np.sqrt(2.0 / K)).astype(np.float32))   # kernel (O,I,H,W)
    co = nb_box * (5 + nb_class)
    std = head_std if head_std is not None else np.sqrt(1.0 / 1024)
    parts.append((rs.randn(co) * 0.1).astype(np.float32))                  # conv_23 bias
    parts.append((rs.randn(co * 1024) * std).astype(np.float32))           # conv_23 kernel
    blob = np.concatenate(parts)
    assert blob.size == darknet_blob_size(nb_class, nb_box)
    return blob


def _glorot(rs, shape, fan_in, fan_out):
    lim = np.sqrt(6.0 / (fan_in + fan_out))
    return rs.uniform(-lim, lim, shape).astype(np.float32)


def synth_tracker_weights(nb_class, nb_box=5, units=512, seed=1235):
    """ConvLSTM2D + tconv_2 weights in Keras layouts (MultiObjDetTracker.py:176,182):
    Glorot-uniform kernels, forget-gate bias 1."""
    rs = np.random.RandomState(seed)
    cb = nb_box * (5 + nb_class)
    cin = cb + 1024
    w = dict(
        kernel=_glorot(rs, (3, 3, cin, 4 * units), 9 * cin, 9 * 4 * units),
        recurrent=_glorot(rs, (3, 3, units, 4 * units), 9 * units, 9 * 4 * units),
        bias=np.zeros(4 * units, dtype=np.float32),
        out_kernel=_glorot(rs, (1, 1, units, cb), units, cb),
        out_bias=(rs.randn(cb) * 0.1).astype(np.float32),
    )
    w["bias"][units:2 * units] = 1.0
    return w


def synth_tiny_weights(feat_dim=512, units=512, seed=1236):
    """LSTM(512) + Dense(4) weights in Keras layouts (TinyTracker.py:36-37)."""
    rs = np.random.RandomState(seed)
    D = feat_dim + 4
    w = dict(
        kernel=_glorot(rs, (D, 4 * units), D, 4 * units),
        recurrent=_glorot(rs, (units, 4 * units), units, 4 * units),
        bias=np.zeros(4 * units, dtype=np.float32),
        dense_kernel=_glorot(rs, (units, 4), units, 4),
        dense_bias=(rs.randn(4) * 0.1).astype(np.float32),
    )
    w["bias"][units:2 * units] = 1.0
    return w


def synth_heatmap_weights(feat_dim=512, hmap=32, units=512, seed=1237):
    """LSTM(512) + Dense(hmap*hmap) weights (TinyHeatmapTracker.py:42-43)."""
    rs = np.random.RandomState(seed)
    D, O = feat_dim + hmap * hmap, hmap * hmap
    w = dict(
        kernel=_glorot(rs, (D, 4 * units), D, 4 * units),
        recurrent=_glorot(rs, (units, 4 * units), units, 4 * units),
        bias=np.zeros(4 * units, dtype=np.float32),
        dense_kernel=_glorot(rs, (units, O), units, O) * 4.0,
        dense_bias=(rs.randn(O) * 0.5).astype(np.float32),
    )
    w["bias"][units:2 * units] = 1.0
    return w


def synth_clip(T, H, W, n_obj, seed):
    """uint8 [T,H,W,3] frames: low-frequency background + n_obj bright rectangles
    moving at constant velocity (ImageNet-VID / MOT17-shaped: consecutive frames
    are correlated)."""
    rs = np.random.RandomState(seed)
    coarse = rs.randint(40, 160, size=(H // 32 + 1, W // 32 + 1, 3)).astype(np.float32)
    bg = np.kron(coarse, np.ones((32, 32, 1), dtype=np.float32))[:H, :W]
    frames = np.empty((T, H, W, 3), dtype=np.uint8)
    pos = rs.rand(n_obj, 2) * [H * 0.8, W * 0.8]
    vel = (rs.rand(n_obj, 2) - 0.5) * 8.0
    size = (rs.rand(n_obj, 2) * 0.12 + 0.04) * [H, W]
    col = rs.randint(150, 256, size=(n_obj, 3))
    for t in range(T):
        f = bg + rs.randn(H, W, 1).astype(np.float32) * 4.0
        for k in range(n_obj):
            y0, x0 = pos[k] + vel[k] * t
            y0 = int(np.clip(y0, 0, H - 2)); x0 = int(np.clip(x0, 0, W - 2))
            y1 = int(min(H, y0 + size[k, 0])); x1 = int(min(W, x0 + size[k, 1]))
            f[y0:y1, x0:x1] = col[k]
        frames[t] = np.clip(f, 0, 255).astype(np.uint8)
    return frames

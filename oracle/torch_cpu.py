"""The same graph as oracle.py on torch-CPU float32 (ATen / oneDNN convolutions, all host cores).

TEST INFRASTRUCTURE ONLY, like the rest of oracle/ (see oracle.c's header): used by bench.py's cpu_baseline leg
as the FASTER of the two CPU statements of the path (the C port in oracle.c is a plain loop nest; a vendor-tuned
convolution is the fairer thing to time next to the GPU) and by tests/test_oracle.py to cross-check the C port at
sizes where both finish in seconds.  Graph composition follows the same reference lines as oracle.py:
  models_detection/KerasYOLO.py:277-405, models_tracking/MultiObjDetTracker.py:160-189.
It is NOT Keras/TensorFlow (neither can run in this image)."""
import numpy as np
import torch
import torch.nn.functional as F

from . import oracle as orc

CHANNELS_LAST = True      # bench.py's cpu_baseline times both settings and reports the faster


def _conv(x, kernel_hwio, bias=None):
    """x [N,C,H,W] float32, kernel HWIO numpy -> [N,O,H,W] ('same', stride 1)"""
    w = torch.from_numpy(np.ascontiguousarray(kernel_hwio.transpose(3, 2, 0, 1)))
    b = torch.from_numpy(np.ascontiguousarray(bias)) if bias is not None else None
    if CHANNELS_LAST and x.dim() == 4:      # oneDNN's preferred (NHWC) layout: no reorder in front of every primitive
        x = x.contiguous(memory_format=torch.channels_last)
        w = w.contiguous(memory_format=torch.channels_last)
    return F.conv2d(x, w, b, padding=kernel_hwio.shape[0] // 2)


def _bn_leaky(x, L):
    inv = torch.from_numpy(L["gamma"] * (1.0 / np.sqrt(L["var"] + np.float32(orc.BN_EPS)))).float()
    sh = torch.from_numpy(L["beta"]).float() - torch.from_numpy(L["mean"]).float() * inv
    y = x * inv.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)
    return F.leaky_relu(y, orc.LEAKY)


def yolov2_forward(frames, layers):
    """frames float32 [B,H,W,3] normalised -> (netout [B,G,G,5,5+C], conv_feat [B,G,G,1024]) numpy"""
    with torch.no_grad():
        x = torch.from_numpy(np.ascontiguousarray(frames, dtype=np.float32)).permute(0, 3, 1, 2)
        skip = None
        for (i, k, ci, co, pool) in orc.TRUNK:
            x = _bn_leaky(_conv(x, layers[i]["kernel"]), layers[i])
            if i == 13:
                skip = x
            if pool:
                x = F.max_pool2d(x, 2)
        s = _bn_leaky(_conv(skip, layers[21]["kernel"]), layers[21])
        B, C, H, W = s.shape                                       # tf.space_to_depth(2), NHWC channel order (dy, dx, c)
        s = s.view(B, C, H // 2, 2, W // 2, 2).permute(0, 3, 5, 1, 2, 4).reshape(B, 4 * C, H // 2, W // 2)
        x = torch.cat([s, x], dim=1)                               # skip first (KerasYOLO.py:391)
        feat = _bn_leaky(_conv(x, layers[22]["kernel"]), layers[22])
        raw = _conv(feat, layers[23]["kernel"], layers[23]["bias"])
        raw = raw.permute(0, 2, 3, 1).contiguous().numpy()
        B, G1, G2, ch = raw.shape
        return raw.reshape(B, G1, G2, 5, ch // 5), feat.permute(0, 2, 3, 1).contiguous().numpy()


def tracker_forward(frames, layers, trk):
    """one clip [T,H,W,3] normalised -> (tracking, detection) [T,G,G,5,5+C] numpy  (MultiObjDetTracker.py:160-189)"""
    det, feat = yolov2_forward(frames, layers)
    T, G1, G2, NB, S = det.shape
    with torch.no_grad():
        z = torch.from_numpy(np.concatenate([det.reshape(T, G1, G2, NB * S), feat], -1)).permute(0, 3, 1, 2)   # x_bbox first (:175)
        U = trk["recurrent"].shape[2]
        zx = _conv(z, trk["kernel"], trk["bias"])                  # all T input projections at once
        h = torch.zeros((1, U, G1, G2)); c = torch.zeros_like(h)
        hs = []
        hard = lambda v: torch.clamp(0.2 * v + 0.5, 0.0, 1.0)
        for t in range(T):
            g = zx[t:t + 1] + _conv(h, trk["recurrent"])
            i, f, cc, o = g[:, :U], g[:, U:2 * U], g[:, 2 * U:3 * U], g[:, 3 * U:]
            c = hard(f) * c + hard(i) * torch.tanh(cc)
            h = hard(o) * torch.tanh(c)
            hs.append(h)
        out = _conv(torch.cat(hs, 0), trk["out_kernel"], trk["out_bias"]).permute(0, 2, 3, 1).contiguous().numpy()
    return out.reshape(T, G1, G2, NB, S), det

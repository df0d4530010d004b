"""kshim -- a minimal functional-API stand-in for the Keras 2.x names the reference's
graph-building code uses, evaluated with torch-CPU in float64.

TEST INFRASTRUCTURE, build container only (tools/make_graph_goldens.py).  Purpose: execute the
reference's OWN `load_model` bodies (models_detection/KerasYOLO.py:239-405,
models_tracking/MultiObjDetTracker.py:160-189, models_tracking/TinyTracker.py:25-41) so that the graph
TOPOLOGY -- layer order, names, concat orders, where the skip is tapped, which tensor feeds the ConvLSTM,
the weight-file read order of init_weights -- comes from the reference's code and not from this
repository's reading of it.  What the shim supplies is the per-layer arithmetic, written from the public
Keras 2.x / TF1 layer definitions (it shares no code with oracle/ or the HIP kernels):

  Conv2D('same', stride 1)        zero padding k//2, cross-correlation, kernel HWIO, optional bias
  BatchNormalization()            inference form, epsilon 1e-3 (Keras default), weights [gamma, beta, mean, var]
  LeakyReLU(alpha)                x if x > 0 else alpha*x
  MaxPooling2D(p[, strides])      'valid'
  tf.space_to_depth(x, 2)         NHWC: out[h, w, (dy*2+dx)*C + c] = in[2h+dy, 2w+dx, c]
  concatenate                     last axis
  ConvLSTM2D(U, 3x3, 'same')      gates i,f,c,o along the last kernel axis, hard_sigmoid / tanh,
                                  bias on the input convolution, zero initial state
  LSTM(U, implementation=2)       z = x.W + h.U + b, same gate order and activations
  Dense(n, activation)            x.W + b
  GlobalMaxPooling2D / Flatten / Reshape / Lambda / TimeDistributed / Model

This is NOT Keras: arithmetic parity with Keras/TensorFlow stays unpinned (neither can run here).
"""
import numpy as np
import torch
import torch.nn.functional as F

DT = torch.float64


class KTensor(object):
    def __init__(self, layer, inputs, shape=None):
        self.layer = layer          # producing layer (None for Input)
        self.inputs = inputs        # list of KTensor
        self.shape = shape

    def __iter__(self):             # `for i, out in enumerate(model.output)` over a list handled by list itself
        raise TypeError("KTensor is not iterable")


def Input(batch_shape=None, shape=None, dtype=None, name=None):
    t = KTensor(None, [], batch_shape if batch_shape is not None else (None,) + tuple(shape))
    t.name = name
    return t


class Layer(object):
    def __init__(self, name=None, **kw):
        self.name = name
        self.output = None
        self.weights = []

    def __call__(self, x):
        ins = list(x) if isinstance(x, (list, tuple)) else [x]
        self.output = KTensor(self, ins)
        return self.output

    def get_weights(self):
        return [w.numpy() for w in self.weights]

    def set_weights(self, ws):
        assert len(ws) == len(self.weights), "%s: %d weights given, %d expected" % (self.name, len(ws), len(self.weights))
        for i, w in enumerate(ws):
            w = torch.as_tensor(np.asarray(w), dtype=DT)
            assert tuple(w.shape) == tuple(self.weights[i].shape), (self.name, w.shape, self.weights[i].shape)
            self.weights[i] = w.clone()

    def compute(self, vals):
        raise NotImplementedError


def _evaluate(t, feed, memo):
    if id(t) in memo:
        return memo[id(t)]
    if t.layer is None:
        v = feed[id(t)]
    else:
        v = t.layer.compute([_evaluate(i, feed, memo) for i in t.inputs])
    memo[id(t)] = v
    return v


def _conv_same(x, kernel, bias):
    """x [N,H,W,C], kernel HWIO -> [N,H,W,O]"""
    k = kernel.shape[0]
    y = F.conv2d(x.permute(0, 3, 1, 2), kernel.permute(3, 2, 0, 1), bias, stride=1, padding=k // 2)
    return y.permute(0, 2, 3, 1)


class Conv2D(Layer):
    def __init__(self, filters, kernel_size, strides=(1, 1), padding='valid', use_bias=True, name=None,
                 kernel_initializer=None, **kw):
        Layer.__init__(self, name)
        assert tuple(strides) == (1, 1) and padding == 'same' and kernel_size[0] == kernel_size[1] and kernel_size[0] % 2 == 1
        self.filters, self.k, self.use_bias = filters, kernel_size[0], use_bias
        self.built = False

    def build(self, cin):
        if not self.built:
            self.weights = [torch.zeros((self.k, self.k, cin, self.filters), dtype=DT)]
            if self.use_bias:
                self.weights.append(torch.zeros((self.filters,), dtype=DT))
            self.built = True

    def __call__(self, x):
        out = Layer.__call__(self, x)
        cin = _static_channels(out.inputs[0])
        if cin is not None:
            self.build(cin)
        return out

    def set_weights(self, ws):
        self.build(np.asarray(ws[0]).shape[2])
        Layer.set_weights(self, ws)

    def compute(self, vals):
        x = vals[0]
        self.build(x.shape[-1])
        return _conv_same(x, self.weights[0], self.weights[1] if self.use_bias else None)


class BatchNormalization(Layer):
    def __init__(self, name=None, epsilon=1e-3, **kw):
        Layer.__init__(self, name)
        self.epsilon = epsilon

    def __call__(self, x):
        out = Layer.__call__(self, x)
        c = _static_channels(out.inputs[0])
        self.weights = [torch.ones(c, dtype=DT), torch.zeros(c, dtype=DT), torch.zeros(c, dtype=DT), torch.ones(c, dtype=DT)]
        return out

    def compute(self, vals):
        gamma, beta, mean, var = self.weights
        return (vals[0] - mean) / torch.sqrt(var + self.epsilon) * gamma + beta


class LeakyReLU(Layer):
    def __init__(self, alpha=0.3, name=None):
        Layer.__init__(self, name)
        self.alpha = alpha

    def compute(self, vals):
        x = vals[0]
        return torch.where(x > 0, x, self.alpha * x)


class MaxPooling2D(Layer):
    def __init__(self, pool_size=(2, 2), strides=None, name=None, **kw):
        Layer.__init__(self, name)
        self.pool = tuple(pool_size)
        self.strides = tuple(strides) if strides is not None else self.pool

    def compute(self, vals):
        y = F.max_pool2d(vals[0].permute(0, 3, 1, 2), self.pool, self.strides)
        return y.permute(0, 2, 3, 1)


class GlobalMaxPooling2D(Layer):
    def compute(self, vals):
        return vals[0].amax(dim=(1, 2))


class Flatten(Layer):
    def compute(self, vals):
        return vals[0].reshape(vals[0].shape[0], -1)


class Reshape(Layer):
    def __init__(self, target_shape, name=None):
        Layer.__init__(self, name)
        self.target = tuple(target_shape)

    def compute(self, vals):
        return vals[0].reshape((vals[0].shape[0],) + self.target)


class Lambda(Layer):
    def __init__(self, fn, name=None):
        Layer.__init__(self, name)
        self.fn = fn

    def __call__(self, x):
        self.multi = isinstance(x, (list, tuple))
        out = Layer.__call__(self, x)
        try:        # static channel count of the result, by probing fn on a tiny tensor
            cin = _static_channels(out.inputs[0])
            probe = torch.zeros((1, 2, 2, cin), dtype=DT)
            self.out_channels = int(self.fn([probe] * len(out.inputs) if self.multi else probe).shape[-1])
        except Exception:
            self.out_channels = None
        return out

    def compute(self, vals):
        return self.fn(vals if self.multi else vals[0])


class _Concat(Layer):
    def compute(self, vals):
        return torch.cat(vals, dim=-1)


def concatenate(tensors, axis=-1):
    assert axis == -1
    return _Concat()(list(tensors))


class Dense(Layer):
    def __init__(self, units, activation=None, name=None, **kw):
        Layer.__init__(self, name)
        self.units, self.activation = units, activation

    def compute(self, vals):
        x = vals[0]
        if not self.weights:
            self.weights = [torch.zeros((x.shape[-1], self.units), dtype=DT), torch.zeros(self.units, dtype=DT)]
        y = x @ self.weights[0] + self.weights[1]
        if self.activation == 'sigmoid':
            y = torch.sigmoid(y)
        elif self.activation is not None:
            raise NotImplementedError(self.activation)
        return y

    def set_weights(self, ws):
        self.weights = [torch.as_tensor(np.asarray(w), dtype=DT).clone() for w in ws]


def hard_sigmoid(x):
    return torch.clamp(0.2 * x + 0.5, 0.0, 1.0)


class ConvLSTM2D(Layer):
    """keras.layers.ConvLSTM2D defaults: activation tanh, recurrent_activation hard_sigmoid, use_bias."""

    def __init__(self, filters, kernel_size, strides=(1, 1), padding='valid', return_sequences=False, name=None, **kw):
        Layer.__init__(self, name)
        assert tuple(strides) == (1, 1) and padding == 'same' and return_sequences
        self.U, self.k = filters, kernel_size[0]

    def set_weights(self, ws):      # [kernel (k,k,Cin,4U), recurrent_kernel (k,k,U,4U), bias (4U)]
        assert len(ws) == 3
        self.weights = [torch.as_tensor(np.asarray(w), dtype=DT).clone() for w in ws]
        assert self.weights[1].shape == (self.k, self.k, self.U, 4 * self.U)

    def compute(self, vals):
        x = vals[0]                               # [B,T,H,W,C]
        Wk, Uk, b = self.weights
        B, T, H, W, _ = x.shape
        U = self.U
        h = torch.zeros((B, H, W, U), dtype=DT)
        c = torch.zeros((B, H, W, U), dtype=DT)
        outs = []
        for t in range(T):
            xt = x[:, t]
            zx = [_conv_same(xt, Wk[..., g * U:(g + 1) * U], b[g * U:(g + 1) * U]) for g in range(4)]   # i, f, c, o
            zh = [_conv_same(h, Uk[..., g * U:(g + 1) * U], None) for g in range(4)]
            i = hard_sigmoid(zx[0] + zh[0])
            f = hard_sigmoid(zx[1] + zh[1])
            c = f * c + i * torch.tanh(zx[2] + zh[2])
            o = hard_sigmoid(zx[3] + zh[3])
            h = o * torch.tanh(c)
            outs.append(h)
        return torch.stack(outs, dim=1)


class LSTM(Layer):
    """keras.layers.LSTM defaults (tanh / hard_sigmoid); implementation=2 is one fused matmul."""

    def __init__(self, units, return_sequences=False, implementation=1, name=None, **kw):
        Layer.__init__(self, name)
        assert return_sequences
        self.U = units

    def set_weights(self, ws):      # [kernel (D,4U), recurrent_kernel (U,4U), bias (4U)]
        assert len(ws) == 3
        self.weights = [torch.as_tensor(np.asarray(w), dtype=DT).clone() for w in ws]

    def compute(self, vals):
        x = vals[0]                               # [B,T,D]
        Wk, Ur, b = self.weights
        B, T, _ = x.shape
        U = self.U
        h = torch.zeros((B, U), dtype=DT)
        c = torch.zeros((B, U), dtype=DT)
        outs = []
        for t in range(T):
            z = x[:, t] @ Wk + h @ Ur + b
            i = hard_sigmoid(z[:, :U]); f = hard_sigmoid(z[:, U:2 * U])
            c = f * c + i * torch.tanh(z[:, 2 * U:3 * U])
            o = hard_sigmoid(z[:, 3 * U:])
            h = o * torch.tanh(c)
            outs.append(h)
        return torch.stack(outs, dim=1)


class TimeDistributed(Layer):
    def __init__(self, layer, name=None):
        Layer.__init__(self, name)
        self.layer = layer

    def get_weights(self):
        return self.layer.get_weights()

    def set_weights(self, ws):
        self.layer.set_weights(ws)

    def compute(self, vals):
        x = vals[0]
        B, T = x.shape[:2]
        flat = x.reshape((B * T,) + tuple(x.shape[2:]))
        y = self.layer.compute([flat])
        return y.reshape((B, T) + tuple(y.shape[1:]))


class Model(Layer):
    def __init__(self, inputs=None, outputs=None, name=None):
        Layer.__init__(self, name)
        self.input = inputs
        self.outputs_spec = outputs
        self.output = outputs            # Keras: tensor or list of tensors
        self._ins = list(inputs) if isinstance(inputs, (list, tuple)) else [inputs]
        self._outs = list(outputs) if isinstance(outputs, (list, tuple)) else [outputs]

    def __call__(self, x):
        ins = list(x) if isinstance(x, (list, tuple)) else [x]
        return KTensor(self, ins)

    def compute(self, vals):
        feed = {id(t): v for t, v in zip(self._ins, vals)}
        memo = {}
        outs = [_evaluate(o, feed, memo) for o in self._outs]
        return outs if isinstance(self.outputs_spec, (list, tuple)) else outs[0]

    def predict(self, xs, batch_size=None):
        xs = list(xs) if isinstance(xs, (list, tuple)) else [xs]
        vals = [None if v is None else torch.as_tensor(np.asarray(v), dtype=DT) for v in xs]
        with torch.no_grad():
            out = self.compute(vals)
        return [o.numpy() for o in out] if isinstance(out, list) else out.numpy()

    def layers(self):
        seen, order = set(), []

        def walk(t):
            if id(t) in seen:
                return
            seen.add(id(t))
            for i in t.inputs:
                walk(i)
            if t.layer is not None:
                order.append(t.layer)
                if isinstance(t.layer, Model):
                    for o in t.layer._outs:
                        walk(o)
                if isinstance(t.layer, TimeDistributed) and isinstance(t.layer.layer, Model):
                    for o in t.layer.layer._outs:
                        walk(o)
        for o in self._outs:
            walk(o)
        return order

    def get_layer(self, name):
        for l in self.layers():
            if l.name == name:
                return l
        raise ValueError("No such layer: " + name)

    def summary(self):
        pass

    def compile(self, **kw):
        pass


def _static_channels(t):
    """channel count of a symbolic tensor, walking back through shape-preserving layers"""
    while True:
        if t.layer is None:
            return t.shape[-1]
        l = t.layer
        if isinstance(l, Conv2D):
            return l.filters
        if isinstance(l, _Concat):
            return sum(_static_channels(i) for i in t.inputs)
        if isinstance(l, Lambda):
            return l.out_channels
        if isinstance(l, (BatchNormalization, LeakyReLU, MaxPooling2D)):
            t = t.inputs[0]
            continue
        raise NotImplementedError(type(l))


class _TF(object):
    """the two `tf.` names the graph-building code touches"""

    @staticmethod
    def space_to_depth(x, block_size):
        assert block_size == 2
        B, H, W, C = x.shape
        y = x.reshape(B, H // 2, 2, W // 2, 2, C)          # [b, h, dy, w, dx, c]
        return y.permute(0, 1, 3, 2, 4, 5).reshape(B, H // 2, W // 2, 4 * C)   # channel = (dy*2+dx)*C + c


tf = _TF()


class Adam(object):
    def __init__(self, **kw):
        pass

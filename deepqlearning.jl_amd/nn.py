"""
Host-side description of solver.qnetwork: the Flux vocabulary the reference's users write
(README.md:26-46: Chain(Dense(2,32), Dense(32,n))) mirrored as plain Python descriptors, plus
create_dueling_network (src/dueling.jl:36-58).  No arithmetic happens here: descriptors are lowered to
dqn_layer_desc records and handed to the HIP engine.
"""
from __future__ import annotations

import numpy as np

from . import _abi

identity, relu, tanh, sigmoid = _abi.ACT_IDENTITY, _abi.ACT_RELU, _abi.ACT_TANH, _abi.ACT_SIGMOID


class flattenbatch:  # src/helpers.jl:6-8 -- a no-op marker: the engine flattens between Conv and Dense
    kind = "flatten"


class Dense:
    """Flux Dense(in, out, act)."""
    kind = "dense"

    def __init__(self, n_in, n_out, act=identity):
        self.n_in, self.n_out, self.act = int(n_in), int(n_out), act

    def shapes(self):  # host array shapes in Julia memory order: weight (out,in) == C (in,out); bias (out,)
        return [(self.n_in, self.n_out), (self.n_out,)]

    def fans(self):
        return self.n_in, self.n_out


class Conv:
    """Flux Conv((k,k), cin=>cout, act; stride) -- true convolution, no padding."""
    kind = "conv"

    def __init__(self, k, cin, cout, act=identity, stride=1):
        self.kh, self.kw = (k, k) if np.isscalar(k) else (int(k[0]), int(k[1]))
        self.sh, self.sw = (stride, stride) if np.isscalar(stride) else (int(stride[0]), int(stride[1]))
        self.cin, self.cout, self.act = int(cin), int(cout), act

    def shapes(self):  # weight (kw,kh,cin,cout) == C (cout,cin,kh,kw)
        return [(self.cout, self.cin, self.kh, self.kw), (self.cout,)]

    def fans(self):
        return self.kh * self.kw * self.cin, self.kh * self.kw * self.cout


class LSTM:
    """Flux LSTM(in, out) = Recur(LSTMCell): params Wi (4out,in), Wh (4out,out), b (4out, forget gate bias 1), state0 (h0, c0)."""
    kind = "lstm"

    def __init__(self, n_in, n_out):
        self.n_in, self.n_out, self.act = int(n_in), int(n_out), identity

    def shapes(self):   # Julia memory order: Wi (4out,in) == C (in,4out), Wh (4out,out) == C (out,4out), b, h0, c0
        h = self.n_out
        return [(self.n_in, 4 * h), (h, 4 * h), (4 * h,), (h,), (h,)]


class Chain:
    def __init__(self, *layers):
        self.layers = [l for l in layers if getattr(l, "kind", None) != "flatten" and l is not flattenbatch]

    def __iter__(self):
        return iter(self.layers)

    def __len__(self):
        return len(self.layers)


class DuelingNetwork:
    """src/dueling.jl:1-6: base, val, adv chains; Flux.params order = base, val, adv."""

    def __init__(self, base, val, adv):
        self.base, self.val, self.adv = base, val, adv


def create_dueling_network(m: Chain) -> DuelingNetwork:
    """src/dueling.jl:36-58: split the trailing run of Dense layers into value / advantage streams; the value stream
    gets a fresh Dense(in_of_last, 1).  Throws the reference's error string if there is no trailing Dense."""
    layers = m.layers
    n = len(layers)
    duel_layer = -1
    for i in range(1, n + 1):
        if getattr(layers[n - i], "kind", None) != "dense":
            duel_layer = n - i + 1
            break
        elif i == n:
            duel_layer = 0
    if duel_layer == -1:
        raise _abi.DQNError("DeepQLearningError: the qnetwork provided is incompatible with dueling")
    trailing = layers[duel_layer:]
    if not trailing:      # the chain does not end in a Dense layer: nothing to split into value / advantage streams
        raise _abi.DQNError("DeepQLearningError: the qnetwork provided is incompatible with dueling")
    last = trailing[-1]
    val = Chain(*[Dense(l.n_in, l.n_out, l.act) for l in trailing[:-1]], Dense(last.n_in, 1))
    adv = Chain(*[Dense(l.n_in, l.n_out, l.act) for l in trailing])
    return DuelingNetwork(Chain(*layers[:duel_layer]), val, adv)


def lower(net):
    """Chain | DuelingNetwork -> (list[LayerDesc], dueling flag)."""
    out = []

    def add(chain, stream):
        for l in chain:
            d = _abi.LayerDesc()
            if getattr(l, "kind", None) not in ("dense", "lstm", "conv"):
                raise _abi.DQNError(f"DeepQLearningError: unsupported layer {l!r} (Conv / Dense / LSTM / flattenbatch only)")
            d.act, d.stream = l.act, stream
            if l.kind == "dense":
                d.kind, d.n_in, d.n_out = _abi.LAYER_DENSE, l.n_in, l.n_out
            elif l.kind == "lstm":
                d.kind, d.n_in, d.n_out = _abi.LAYER_LSTM, l.n_in, l.n_out
            else:
                d.kind = _abi.LAYER_CONV
                d.cin, d.cout, d.kh, d.kw, d.sh, d.sw = l.cin, l.cout, l.kh, l.kw, l.sh, l.sw
            out.append(d)

    if isinstance(net, DuelingNetwork):
        add(net.base, _abi.STREAM_BASE)
        add(net.val, _abi.STREAM_VAL)
        add(net.adv, _abi.STREAM_ADV)
        return out, True
    add(net, _abi.STREAM_BASE)
    return out, False


def isrecurrent(m):
    """src/helpers.jl:25-32."""
    return any(getattr(l, "kind", None) == "lstm" for l in all_layers(m))


def all_layers(net):
    return list(net.base) + list(net.val) + list(net.adv) if isinstance(net, DuelingNetwork) else list(net)


def glorot_params(net, seed=1):
    """Flux default init: glorot_uniform weights ((rand - 0.5) * sqrt(24/(fan_in+fan_out))), zero biases, as one flat
    fp32 vector in Flux.params order.  (NumPy's RNG stream, not Julia's.)"""
    rng = np.random.default_rng(seed)
    parts = []
    for l in all_layers(net):
        if l.kind == "lstm":
            h = l.n_out
            for shp, fi, fo in (((l.n_in, 4 * h), l.n_in, 4 * h), ((h, 4 * h), h, 4 * h)):
                parts.append(((rng.random(shp, dtype=np.float32) - np.float32(0.5)) * np.sqrt(np.float32(24.0) / np.float32(fi + fo))).astype(np.float32).reshape(-1))
            b = np.zeros(4 * h, np.float32)
            b[h:2 * h] = 1.0        # Flux LSTMCell: forget-gate bias initialised to 1
            parts += [b, np.zeros(h, np.float32), np.zeros(h, np.float32)]
            continue
        wshape, bshape = l.shapes()
        fi, fo = l.fans()
        parts.append(((rng.random(wshape, dtype=np.float32) - np.float32(0.5)) * np.sqrt(np.float32(24.0) / np.float32(fi + fo))).astype(np.float32).reshape(-1))
        parts.append(np.zeros(bshape, np.float32))
    return np.concatenate(parts)


def nature_dqn(n_actions=4, in_channels=4):
    """BASELINE config 2: Chain(Conv((8,8),4=>32,relu;stride=4), Conv((4,4),32=>64,relu;stride=2),
    Conv((3,3),64=>64,relu), flattenbatch, Dense(3136,512,relu), Dense(512,nA)) for 84x84 inputs."""
    return Chain(Conv(8, in_channels, 32, relu, 4), Conv(4, 32, 64, relu, 2), Conv(3, 64, 64, relu, 1), flattenbatch,
                 Dense(3136, 512, relu), Dense(512, n_actions))

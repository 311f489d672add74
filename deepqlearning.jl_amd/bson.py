"""
qnetwork.bson -- the checkpoint file of the reference (src/solver.jl:290-300):

    bson(joinpath(solver.logdir, "qnetwork.bson"), qnetwork = [w for w in Flux.params(active_q)])

read back by restore_best_model (src/solver.jl:302-318) as BSON.load(path)[:qnetwork] -> Flux.loadparams!.

This module writes that file from the engine's flat Flux.params vector (dqn_get_params) and reads it back, with no
dependency beyond NumPy.  The container format is BSON 1.1 (bsonspec.org); the array lowering is BSON.jl's
(third-party, RECALLED -- src/extensions.jl of BSON.jl 0.3):

    Array{T,N} with isbits T  ->  {"tag": "array", "type": <datatype T>, "size": [Int64...], "data": binary(raw column-major bytes)}
    DataType T                ->  {"tag": "datatype", "name": ["Core", "Float32"], "params": []}
    Vector of non-isbits      ->  plain BSON array of the lowered elements        (the vector of parameter arrays)
    top level                 ->  document {"qnetwork": [...]}                    (bson(path; kw...) == bson(path, Dict(kw)))

UNVERIFIED AGAINST BSON.jl: no Julia exists in the build image or on the GPU box, so the lowering above cannot be
exercised against the real reader; tests/test_bson_cpu.py round-trips it through the independent reader below and
checks the byte-level structure against the BSON spec.  Arrays are stored in JULIA memory order and with Julia's
`size`: Dense weight (out, in), Conv weight (kw, kh, cin, cout), LSTM Wi (4h, in), Wh (4h, h) -- exactly the bytes the
C ABI's flat vector already holds.
"""
from __future__ import annotations

import struct

import numpy as np

# ---- BSON element type bytes (bsonspec.org)
_DOUBLE, _STRING, _DOC, _ARRAY, _BINARY, _BOOL, _NULL, _INT32, _INT64 = 0x01, 0x02, 0x03, 0x04, 0x05, 0x08, 0x0A, 0x10, 0x12


def julia_param_shapes(net):
    """[(julia_size_tuple, n_elements)] of Flux.params(net) in order; `net` is an nn.Chain or nn.DuelingNetwork."""
    from . import nn
    out = []
    for l in nn.all_layers(net):
        if l.kind == "dense":
            out += [((l.n_out, l.n_in), l.n_out * l.n_in), ((l.n_out,), l.n_out)]
        elif l.kind == "conv":
            out += [((l.kw, l.kh, l.cin, l.cout), l.kw * l.kh * l.cin * l.cout), ((l.cout,), l.cout)]
        elif l.kind == "lstm":   # Flux 0.14 Recur(LSTMCell): Wi, Wh, b, state0 = (h0, c0) as (out, 1) matrices
            h = l.n_out
            out += [((4 * h, l.n_in), 4 * h * l.n_in), ((4 * h, h), 4 * h * h), ((4 * h,), 4 * h), ((h, 1), h), ((h, 1), h)]
        else:
            raise ValueError(f"unsupported layer kind {l.kind}")
    return out


# ------------------------------------------------------------------ writer
def _cstring(s: str) -> bytes:
    b = s.encode("utf-8")
    assert b"\x00" not in b
    return b + b"\x00"


def _enc_value(v):
    """-> (type byte, payload)"""
    if isinstance(v, dict):
        return _DOC, _enc_doc(v.items())
    if isinstance(v, (list, tuple)):
        return _ARRAY, _enc_doc((str(i), x) for i, x in enumerate(v))       # BSON arrays: documents keyed "0", "1", ...
    if isinstance(v, (bytes, bytearray, memoryview)):
        b = bytes(v)
        return _BINARY, struct.pack("<i", len(b)) + b"\x00" + b             # generic binary subtype 0
    if isinstance(v, str):
        b = v.encode("utf-8") + b"\x00"
        return _STRING, struct.pack("<i", len(b)) + b
    if isinstance(v, bool):
        return _BOOL, b"\x01" if v else b"\x00"
    if isinstance(v, (int, np.integer)):
        return _INT64, struct.pack("<q", int(v))                            # Julia Int is Int64: size entries are written as int64
    if isinstance(v, float):
        return _DOUBLE, struct.pack("<d", v)
    if v is None:
        return _NULL, b""
    raise TypeError(type(v))


def _enc_doc(items) -> bytes:
    body = b""
    for k, v in items:
        t, payload = _enc_value(v)
        body += bytes([t]) + _cstring(k) + payload
    body += b"\x00"
    return struct.pack("<i", len(body) + 4) + body


_JULIA_TYPES = {np.dtype(np.float32): "Float32", np.dtype(np.float64): "Float64", np.dtype(np.int32): "Int32", np.dtype(np.int64): "Int64", np.dtype(np.uint8): "UInt8"}


def lower_array(a: np.ndarray, julia_size):
    """BSON.jl's lowering of an Array{T,N} of bits type; `a` holds the elements in Julia (column-major) memory order."""
    a = np.ascontiguousarray(a)
    assert int(np.prod(julia_size)) == a.size
    return {"tag": "array", "type": {"tag": "datatype", "name": ["Core", _JULIA_TYPES[a.dtype]], "params": []},
            "size": [int(d) for d in julia_size], "data": a.tobytes()}


def dumps_qnetwork(flat, shapes) -> bytes:
    """flat: the Flux.params vector (dqn_get_params); shapes: julia_param_shapes(net)."""
    flat = np.ascontiguousarray(flat, np.float32).reshape(-1)
    assert sum(n for _, n in shapes) == flat.size, (sum(n for _, n in shapes), flat.size)
    arrays, off = [], 0
    for size, n in shapes:
        arrays.append(lower_array(flat[off:off + n], size))
        off += n
    return _enc_doc([("qnetwork", arrays)])


def save_qnetwork(path, flat, shapes):
    with open(path, "wb") as f:
        f.write(dumps_qnetwork(flat, shapes))


# ------------------------------------------------------------------ reader (independent of the writer: a generic BSON decoder + BSON.jl "raising")
def _dec_doc(buf: memoryview, pos: int, as_array=False):
    (n,) = struct.unpack_from("<i", buf, pos)
    end = pos + n
    assert buf[end - 1] == 0, "document does not end in 0x00"
    pos += 4
    keys, vals = [], []
    while pos < end - 1:
        t = buf[pos]; pos += 1
        z = pos
        while buf[z] != 0:
            z += 1
        key = bytes(buf[pos:z]).decode("utf-8"); pos = z + 1
        if t == _DOUBLE:
            v = struct.unpack_from("<d", buf, pos)[0]; pos += 8
        elif t == _STRING:
            (m,) = struct.unpack_from("<i", buf, pos); v = bytes(buf[pos + 4:pos + 4 + m - 1]).decode("utf-8"); pos += 4 + m
        elif t == _DOC:
            v, pos = _dec_doc(buf, pos)
        elif t == _ARRAY:
            v, pos = _dec_doc(buf, pos, as_array=True)
        elif t == _BINARY:
            (m,) = struct.unpack_from("<i", buf, pos); v = bytes(buf[pos + 5:pos + 5 + m]); pos += 5 + m
        elif t == _BOOL:
            v = buf[pos] != 0; pos += 1
        elif t == _NULL:
            v = None
        elif t == _INT32:
            v = struct.unpack_from("<i", buf, pos)[0]; pos += 4
        elif t == _INT64:
            v = struct.unpack_from("<q", buf, pos)[0]; pos += 8
        else:
            raise ValueError(f"unsupported BSON element type 0x{t:02x}")
        keys.append(key); vals.append(v)
    assert pos == end - 1
    if as_array:
        assert keys == [str(i) for i in range(len(keys))], "BSON array keys must be 0..n-1"
        return vals, end
    return dict(zip(keys, vals)), end


_NP_TYPES = {"Float32": np.float32, "Float64": np.float64, "Int32": np.int32, "Int64": np.int64, "UInt8": np.uint8}


def _raise(v):
    """BSON.jl `raise`: tagged documents back to arrays (element order stays Julia's column-major order; the array is
    returned 1-D together with its Julia size)."""
    if isinstance(v, list):
        return [_raise(x) for x in v]
    if isinstance(v, dict) and v.get("tag") == "array":
        ty = v["type"]
        assert ty["tag"] == "datatype" and ty["name"][0] == "Core" and ty["params"] == []
        a = np.frombuffer(v["data"], dtype=_NP_TYPES[ty["name"][1]])
        assert a.size == int(np.prod(v["size"]))
        return a, tuple(v["size"])
    if isinstance(v, dict):
        return {k: _raise(x) for k, x in v.items()}
    return v


def loads(data: bytes):
    doc, end = _dec_doc(memoryview(data), 0)
    assert end == len(data)
    return _raise(doc)


def load_qnetwork(path):
    """-> (flat fp32 Flux.params vector, [julia sizes])   == BSON.load(path)[:qnetwork], flattened"""
    with open(path, "rb") as f:
        arrays = loads(f.read())["qnetwork"]
    return np.concatenate([a.astype(np.float32) for a, _ in arrays]), [s for _, s in arrays]

"""Dependency-free reader for the Keras HDF5 weight files the reference writes with
``vae.save('vae_model.h5')`` and reads back with ``load_model('vae_model.h5',
custom_objects={'Sampling': Sampling}, compile=False)`` (reference myasp-A1.py:544, 567).

Only the subset of HDF5 that Keras 2.x/h5py emit for such files is implemented (SURVEY.md
appendix B.2): superblock v0 with 8-byte offsets/lengths, old-style groups (object header v1 ->
symbol-table message 0x11 -> B-tree ``TREE`` -> ``SNOD`` -> symbol entries, names in the local
``HEAP``), header continuation blocks (0x10), contiguous (layout v3 class 1) and compact (class 0)
datasets of little-endian floats/ints, and attributes (message 0x0C, versions 1-3) holding fixed
or variable-length strings (the latter in the global heap ``GCOL``).  Anything else raises.
"""
from __future__ import annotations

import json
import struct
from typing import Dict, List, Optional, Tuple

import numpy as np

_SIG = b"\x89HDF\r\n\x1a\n"
_UNDEF = 0xFFFFFFFFFFFFFFFF


class H5Error(ValueError):
    pass


class _Datatype:
    def __init__(self, cls: int, size: int, np_dtype: Optional[np.dtype], vlen_string: bool = False):
        self.cls, self.size, self.np_dtype, self.vlen_string = cls, size, np_dtype, vlen_string


class H5File:
    def __init__(self, path: str):
        self.path = path
        with open(path, "rb") as f:
            self.buf = f.read()
        b = self.buf
        if b[:8] != _SIG:
            raise H5Error(f"{path}: not an HDF5 file")
        if b[8] != 0:
            raise H5Error(f"{path}: superblock version {b[8]} not supported (only v0)")
        if b[13] != 8 or b[14] != 8:
            raise H5Error(f"{path}: only 8-byte offsets/lengths supported")
        # v0 superblock: 8 sig, 8 version/size bytes, 2+2 group K, 4 flags, then 4 addresses, then
        # the root group's symbol-table entry.
        base = 24
        self.base_addr = struct.unpack_from("<Q", b, base)[0]
        root_entry = base + 32
        self.root_header = struct.unpack_from("<Q", b, root_entry + 8)[0]

    # ------------------------------------------------------------------ object headers
    def _messages(self, addr: int) -> List[Tuple[int, int, int]]:
        """(type, offset, size) of every message of a version-1 object header."""
        b = self.buf
        if b[addr] != 1:
            raise H5Error(f"object header version {b[addr]} at {addr} not supported")
        n_msgs = struct.unpack_from("<H", b, addr + 2)[0]
        hdr_size = struct.unpack_from("<I", b, addr + 8)[0]
        blocks = [(addr + 16, hdr_size)]
        out = []
        while blocks and len(out) < n_msgs:
            pos, size = blocks.pop(0)
            end = pos + size
            while pos + 8 <= end and len(out) < n_msgs:
                mtype, msize = struct.unpack_from("<HH", b, pos)
                body = pos + 8
                if mtype == 0x10:                              # continuation
                    coff, clen = struct.unpack_from("<QQ", b, body)
                    blocks.append((coff, clen))
                out.append((mtype, body, msize))
                pos = body + msize
        return out

    # ------------------------------------------------------------------ groups
    def _heap_data(self, heap_addr: int) -> int:
        if self.buf[heap_addr:heap_addr + 4] != b"HEAP":
            raise H5Error("bad local heap signature")
        return struct.unpack_from("<Q", self.buf, heap_addr + 24)[0]

    def _cstr(self, pos: int) -> str:
        end = self.buf.index(b"\0", pos)
        return self.buf[pos:end].decode("utf-8")

    def _walk_btree(self, addr: int, heap_data: int, out: Dict[str, int]):
        b = self.buf
        sig = b[addr:addr + 4]
        if sig == b"TREE":
            node_type, level, used = struct.unpack_from("<BBH", b, addr + 4)
            if node_type != 0:
                raise H5Error("unexpected chunk B-tree in a group")
            pos = addr + 24            # after sig(4) type(1) level(1) used(2) left(8) right(8)
            pos += 8                   # key 0
            for _ in range(used):
                child = struct.unpack_from("<Q", b, pos)[0]
                pos += 16              # child pointer + next key
                self._walk_btree(child, heap_data, out)
        elif sig == b"SNOD":
            n_sym = struct.unpack_from("<H", b, addr + 6)[0]
            pos = addr + 8
            for _ in range(n_sym):
                name_off, hdr = struct.unpack_from("<QQ", b, pos)
                out[self._cstr(heap_data + name_off)] = hdr
                pos += 40
        else:
            raise H5Error(f"bad group node signature {sig!r}")

    def children(self, header_addr: int) -> Dict[str, int]:
        for mtype, body, _ in self._messages(header_addr):
            if mtype == 0x11:
                btree, heap = struct.unpack_from("<QQ", self.buf, body)
                out: Dict[str, int] = {}
                self._walk_btree(btree, self._heap_data(heap), out)
                return out
        return {}

    def resolve(self, path: str) -> int:
        addr = self.root_header
        for part in [p for p in path.split("/") if p]:
            kids = self.children(addr)
            if part not in kids:
                raise KeyError(f"{path}: no member {part!r}")
            addr = kids[part]
        return addr

    def is_dataset(self, header_addr: int) -> bool:
        return any(m[0] == 0x08 for m in self._messages(header_addr))

    def walk_datasets(self, path: str = "/") -> Dict[str, int]:
        out: Dict[str, int] = {}

        def rec(prefix: str, addr: int):
            for name, child in self.children(addr).items():
                full = f"{prefix}/{name}" if prefix else name
                if self.is_dataset(child):
                    out[full] = child
                else:
                    rec(full, child)

        rec(path.strip("/"), self.resolve(path))
        return out

    # ------------------------------------------------------------------ datatypes / dataspaces
    def _datatype(self, pos: int) -> _Datatype:
        b = self.buf
        cls_ver = b[pos]
        cls = cls_ver & 0x0F
        bits0 = b[pos + 1]
        size = struct.unpack_from("<I", b, pos + 4)[0]
        if cls == 0:                                   # fixed-point
            if bits0 & 1:
                raise H5Error("big-endian integers not supported")
            signed = bool(bits0 & 0x08)
            return _Datatype(cls, size, np.dtype(f"<{'i' if signed else 'u'}{size}"))
        if cls == 1:                                   # floating point
            if bits0 & 1:
                raise H5Error("big-endian floats not supported")
            return _Datatype(cls, size, np.dtype(f"<f{size}"))
        if cls == 3:                                   # fixed-length string
            return _Datatype(cls, size, None)
        if cls == 9:                                   # variable length
            vtype = bits0 & 0x0F
            return _Datatype(cls, size, None, vlen_string=(vtype == 1))
        raise H5Error(f"datatype class {cls} not supported")

    def _dataspace(self, pos: int) -> Tuple[int, ...]:
        b = self.buf
        ver, rank, flags = b[pos], b[pos + 1], b[pos + 2]
        if ver == 1:
            p = pos + 8
        elif ver == 2:
            if b[pos + 3] == 2:                        # null dataspace
                return (0,)
            p = pos + 4
        else:
            raise H5Error(f"dataspace version {ver} not supported")
        return tuple(struct.unpack_from("<Q", b, p + 8 * i)[0] for i in range(rank))

    def _global_heap_object(self, coll_addr: int, index: int) -> bytes:
        b = self.buf
        if b[coll_addr:coll_addr + 4] != b"GCOL":
            raise H5Error("bad global heap signature")
        coll_size = struct.unpack_from("<Q", b, coll_addr + 8)[0]
        pos = coll_addr + 16
        end = coll_addr + coll_size
        while pos + 16 <= end:
            idx, _refs = struct.unpack_from("<HH", b, pos)
            osize = struct.unpack_from("<Q", b, pos + 8)[0]
            if idx == 0:
                break
            if idx == index:
                return b[pos + 16:pos + 16 + osize]
            pos += 16 + ((osize + 7) & ~7)
        raise H5Error(f"global heap object {index} not found")

    # ------------------------------------------------------------------ attributes
    def attrs(self, header_addr: int) -> Dict[str, object]:
        b = self.buf
        out: Dict[str, object] = {}
        for mtype, body, _msize in self._messages(header_addr):
            if mtype != 0x0C:
                continue
            ver = b[body]
            name_size, dt_size, ds_size = struct.unpack_from("<HHH", b, body + 2)
            if ver == 1:
                pos = body + 8
                pad = lambda n: (n + 7) & ~7
            elif ver in (2, 3):
                pos = body + 8 + (1 if ver == 3 else 0)
                pad = lambda n: n
            else:
                raise H5Error(f"attribute version {ver} not supported")
            name = b[pos:pos + name_size].split(b"\0")[0].decode()
            pos += pad(name_size)
            dt = self._datatype(pos)
            pos += pad(dt_size)
            shape = self._dataspace(pos)
            pos += pad(ds_size)
            count = int(np.prod(shape)) if shape else 1
            if dt.vlen_string:
                vals = []
                for i in range(count):
                    _ln, coll, idx = struct.unpack_from("<IQI", b, pos + 16 * i)
                    vals.append(self._global_heap_object(coll, idx).decode("utf-8"))
                out[name] = vals[0] if not shape else vals
            elif dt.cls == 3:
                vals = [b[pos + i * dt.size:pos + (i + 1) * dt.size].split(b"\0")[0].decode("utf-8")
                        for i in range(count)]
                out[name] = vals[0] if not shape else vals
            elif dt.np_dtype is not None:
                arr = np.frombuffer(b, dtype=dt.np_dtype, count=count, offset=pos)
                out[name] = arr.reshape(shape).copy() if shape else arr[0]
        return out

    # ------------------------------------------------------------------ datasets
    def read_dataset(self, header_addr: int) -> np.ndarray:
        b = self.buf
        dt = shape = layout = None
        for mtype, body, _ in self._messages(header_addr):
            if mtype == 0x03:
                dt = self._datatype(body)
            elif mtype == 0x01:
                shape = self._dataspace(body)
            elif mtype == 0x08:
                layout = body
        if dt is None or shape is None or layout is None or dt.np_dtype is None:
            raise H5Error("not a numeric dataset")
        if b[layout] != 3:
            raise H5Error(f"data layout version {b[layout]} not supported")
        lclass = b[layout + 1]
        count = int(np.prod(shape)) if shape else 1
        if lclass == 1:                                # contiguous
            addr, _size = struct.unpack_from("<QQ", b, layout + 2)
            if addr == _UNDEF:
                return np.zeros(shape, dtype=dt.np_dtype)
            arr = np.frombuffer(b, dtype=dt.np_dtype, count=count, offset=addr)
        elif lclass == 0:                              # compact
            arr = np.frombuffer(b, dtype=dt.np_dtype, count=count, offset=layout + 4)
        else:
            raise H5Error("chunked datasets are not supported")
        return arr.reshape(shape).copy()

    def get(self, path: str) -> np.ndarray:
        return self.read_dataset(self.resolve(path))


# ---------------------------------------------------------------------- Keras model view
_ACTS = {"linear": 0, "relu": 1, "sigmoid": 2}


def _layer_inputs(layer_cfg) -> List[str]:
    """Names of the layers feeding a functional-API layer (Keras 2.x ``inbound_nodes``)."""
    names: List[str] = []

    def rec(node):
        if isinstance(node, list):
            if len(node) >= 3 and isinstance(node[0], str) and isinstance(node[1], int):
                names.append(node[0])
                return
            for x in node:
                rec(x)
        elif isinstance(node, dict):
            for x in node.values():
                rec(x)

    rec(layer_cfg.get("inbound_nodes", []))
    return names


def vae_topology(model_config: dict, where: str = "model_config") -> dict:
    """The Dense chain of a reference VAE from its Keras ``model_config`` alone (no weights): which
    Dense layers form the encoder, which are z_mean / z_log_var, the decoder layers in call order, their
    activations and the log-variance clip.

    The sampling layer (class ``Sampling``, myasp-A1.py:78-84, or the WaDi ``Lambda``,
    myasp-wadi.py:106-108) names its inputs [z_mean, clipped z_log_var]; the mean path follows z_mean.
    Nested sub-models (the WaDi ``decoder``, myasp-wadi.py:111-117) are flattened in call order.

    Returns {"layer_names", "acts", "mu_layer", "logvar_layer", "logvar_clip"} with layers ordered
    encoder hidden..., mu, logvar, decoder hidden..., output.
    """
    layers_cfg = model_config["config"]["layers"]
    by_name = {l["name"]: l for l in layers_cfg}
    dense = {n: l for n, l in by_name.items() if l["class_name"] == "Dense"}

    sampler = None
    for l in layers_cfg:
        if l["class_name"] in ("Sampling", "Lambda") and len(_layer_inputs(l)) >= 2:
            sampler = l
            break
    if sampler is None:
        raise H5Error(f"{where}: no sampling layer with two inputs in model_config")
    s_in = _layer_inputs(sampler)

    def back_to_dense(name: str) -> str:
        # z_log_var reaches the sampler through clip_by_value (a TFOpLambda / TensorFlowOpLayer).
        seen = 0
        while name not in dense:
            ins = _layer_inputs(by_name[name])
            if not ins or seen > 8:
                raise H5Error(f"{where}: cannot trace {name!r} back to a Dense layer")
            name = ins[0]
            seen += 1
        return name

    mu_name, lv_name = back_to_dense(s_in[0]), back_to_dense(s_in[1])
    logvar_clip = float("inf") if s_in[1] == lv_name else 10.0
    clip_layer = by_name.get(s_in[1])
    if clip_layer is not None and s_in[1] != lv_name:
        # Keras records the clip bounds as call kwargs when it can; default to the scripts' +-10.
        txt = json.dumps(clip_layer)
        if '"clip_value_max"' in txt:
            try:
                node = clip_layer["inbound_nodes"][0]
                kw = node[-1] if isinstance(node[-1], dict) else node[0][-1]
                logvar_clip = float(kw.get("clip_value_max", 10.0))
            except Exception:
                logvar_clip = 10.0

    # encoder chain: walk back from mu to the input
    enc: List[str] = []
    cur = _layer_inputs(dense[mu_name])[0]
    while cur in dense:
        enc.append(cur)
        cur = _layer_inputs(dense[cur])[0]
    enc.reverse()

    # decoder chain: Dense layers downstream of the sampler, in file order; flatten nested models
    dec: List[Tuple[str, dict]] = []
    consumers = {sampler["name"]}
    for l in layers_cfg:
        ins = _layer_inputs(l)
        if not ins or ins[0] not in consumers:
            continue
        if l["class_name"] == "Dense":
            dec.append((l["name"], l))
            consumers.add(l["name"])
        elif l["class_name"] in ("Functional", "Model", "Sequential"):
            for sub in l["config"]["layers"]:
                if sub["class_name"] == "Dense":
                    dec.append((sub["name"], sub))
            consumers.add(l["name"])
        else:
            consumers.add(l["name"])

    names = enc + [mu_name, lv_name] + [n for n, _ in dec]
    cfgs = [dense[n] for n in enc] + [dense[mu_name], dense[lv_name]] + [c for _, c in dec]
    acts = []
    for n, c in zip(names, cfgs):
        act = c["config"].get("activation", "linear")
        if act not in _ACTS:
            raise H5Error(f"{where}: activation {act!r} of layer {n} not supported")
        acts.append(_ACTS[act])
    return {"layer_names": names, "acts": acts, "mu_layer": len(enc), "logvar_layer": len(enc) + 1,
            "logvar_clip": logvar_clip}


def load_keras_vae(path: str) -> dict:
    """Read a reference VAE ``.h5`` into the flat dense-chain description ``drl_set_vae`` takes.

    The topology comes from the file's ``model_config`` (vae_topology; not from the driver script:
    the committed ``vae_model.h5`` has two encoder hidden layers where myasp-A1.py:89-91 builds three,
    SURVEY.md §2 row 17), the weights from ``/model_weights/<layer>/.../{kernel,bias}:0``.

    Returns the vae_topology fields plus {"kernels", "biases", "keras_version", "backend"}.
    """
    f = H5File(path)
    root_attrs = f.attrs(f.root_header)
    if "model_config" not in root_attrs:
        raise H5Error(f"{path}: no model_config attribute (weights-only files are not supported)")
    topo = vae_topology(json.loads(root_attrs["model_config"]), path)
    weights = f.walk_datasets("model_weights")

    def dense_weights(name: str):
        k = [p for p in weights if p.split("/")[-1].startswith("kernel") and f"/{name}/" in f"/{p}"]
        bb = [p for p in weights if p.split("/")[-1].startswith("bias") and f"/{name}/" in f"/{p}"]
        if len(k) != 1 or len(bb) != 1:
            raise H5Error(f"{path}: cannot locate weights of layer {name!r}")
        return f.read_dataset(weights[k[0]]).astype(np.float32), f.read_dataset(weights[bb[0]]).astype(np.float32)

    kernels, biases = [], []
    for n in topo["layer_names"]:
        k, bvec = dense_weights(n)
        kernels.append(np.ascontiguousarray(k))
        biases.append(np.ascontiguousarray(bvec))
    return dict(topo, kernels=kernels, biases=biases, keras_version=root_attrs.get("keras_version"),
                backend=root_attrs.get("backend"))

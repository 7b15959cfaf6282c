"""Reader for TensorFlow-1 checkpoint bundles (``model.index`` + ``model.data-00000-of-00001``).

The reference saves its Q-networks (and, because the Keras VAE lives in the same session, the VAE
Dense layers) with ``tf.compat.v1.train.Saver()`` (reference myasp-A1.py:309-319, 380-382).  Those
bundles are the only source of trained LSTM-DQN weights, so the drop-in has to read them without
TensorFlow.

Format:
  * ``.index`` is an uncompressed LevelDB-style SSTable.  The last 48 bytes are the footer:
    two block handles (varint offset, varint size) for the metaindex and the index block, zero
    padding, and the magic ``57 fb 80 8b 24 75 47 db``.
  * A block is a run of prefix-compressed entries (varint shared, varint non_shared, varint
    value_len, key suffix, value) followed by a uint32 restart array and its uint32 length.  One
    compression byte and a 4-byte CRC follow the block and are not part of its size.
  * Index-block values are handles of data blocks; data-block values are ``BundleEntryProto``
    messages {1: dtype, 2: TensorShapeProto, 3: shard_id, 4: offset, 5: size, 6: crc32c}.
  * ``.data-*`` holds the raw little-endian row-major tensor bytes.
"""
from __future__ import annotations

import os
import struct
from dataclasses import dataclass
from typing import Dict, Iterator, List, Tuple

import numpy as np

_MAGIC = bytes([0x57, 0xFB, 0x80, 0x8B, 0x24, 0x75, 0x47, 0xDB])

# tensorflow/core/framework/types.proto DataType values that occur in these checkpoints.
_DTYPES = {1: np.dtype("<f4"), 2: np.dtype("<f8"), 3: np.dtype("<i4"), 9: np.dtype("<i8")}


def _varint(buf: bytes, pos: int) -> Tuple[int, int]:
    result = 0
    shift = 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7


def _block_entries(block: bytes) -> Iterator[Tuple[bytes, bytes]]:
    """Yield (key, value) pairs of one SSTable block."""
    (n_restarts,) = struct.unpack_from("<I", block, len(block) - 4)
    end = len(block) - 4 - 4 * n_restarts
    pos = 0
    key = b""
    while pos < end:
        shared, pos = _varint(block, pos)
        non_shared, pos = _varint(block, pos)
        value_len, pos = _varint(block, pos)
        key = key[:shared] + block[pos:pos + non_shared]
        pos += non_shared
        value = block[pos:pos + value_len]
        pos += value_len
        yield key, value


def _parse_proto(buf: bytes) -> Dict[int, list]:
    """Minimal protobuf wire-format parser: field number -> list of raw values."""
    out: Dict[int, list] = {}
    pos = 0
    while pos < len(buf):
        tag, pos = _varint(buf, pos)
        field, wire = tag >> 3, tag & 7
        if wire == 0:
            val, pos = _varint(buf, pos)
        elif wire == 1:
            val = buf[pos:pos + 8]
            pos += 8
        elif wire == 2:
            ln, pos = _varint(buf, pos)
            val = buf[pos:pos + ln]
            pos += ln
        elif wire == 5:
            val = buf[pos:pos + 4]
            pos += 4
        else:
            raise ValueError(f"unsupported protobuf wire type {wire}")
        out.setdefault(field, []).append(val)
    return out


# ---- CRC32C (Castagnoli), table driven; TF stores a *masked* CRC of the tensor bytes. ----
def _make_crc_table() -> np.ndarray:
    tbl = np.zeros(256, dtype=np.uint32)
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
        tbl[i] = c
    return tbl


_CRC_TABLE = _make_crc_table()


def crc32c(data: bytes) -> int:
    tbl = _CRC_TABLE
    c = 0xFFFFFFFF
    for b in data:
        c = int(tbl[(c ^ b) & 0xFF]) ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def masked_crc32c(data: bytes) -> int:
    c = crc32c(data)
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


@dataclass
class BundleEntry:
    name: str
    dtype: np.dtype
    shape: Tuple[int, ...]
    shard: int
    offset: int
    size: int
    crc: int


class TFBundle:
    """Random access to the tensors of one checkpoint prefix (e.g. ``.../checkpoints/model``)."""

    def __init__(self, prefix: str):
        self.prefix = prefix
        index_path = prefix + ".index"
        with open(index_path, "rb") as f:
            idx = f.read()
        if len(idx) < 48 or idx[-8:] != _MAGIC:
            raise ValueError(f"{index_path}: not a TF checkpoint index (bad SSTable magic)")
        footer = idx[-48:]
        _, p = _varint(footer, 0)      # metaindex offset
        _, p = _varint(footer, p)      # metaindex size
        ioff, p = _varint(footer, p)
        isize, p = _varint(footer, p)
        self.entries: Dict[str, BundleEntry] = {}
        self.num_shards = 1
        for _, handle in _block_entries(idx[ioff:ioff + isize]):
            boff, hp = _varint(handle, 0)
            bsize, hp = _varint(handle, hp)
            if idx[boff + bsize] != 0:
                raise ValueError(f"{index_path}: compressed SSTable blocks are not supported")
            for key, value in _block_entries(idx[boff:boff + bsize]):
                if key == b"":
                    hdr = _parse_proto(value)
                    self.num_shards = hdr.get(1, [1])[0]
                    if hdr.get(2, [0])[0] != 0:
                        raise ValueError(f"{index_path}: big-endian bundles are not supported")
                    continue
                msg = _parse_proto(value)
                dt = msg.get(1, [0])[0]
                if dt not in _DTYPES:
                    continue           # strings etc. never hold weights
                shape: List[int] = []
                if 2 in msg:
                    for dim in _parse_proto(msg[2][0]).get(2, []):
                        shape.append(_parse_proto(dim).get(1, [0])[0])
                crc = struct.unpack("<I", msg[6][0])[0] if 6 in msg else 0
                self.entries[key.decode()] = BundleEntry(
                    key.decode(), _DTYPES[dt], tuple(shape), msg.get(3, [0])[0],
                    msg.get(4, [0])[0], msg.get(5, [0])[0], crc)

    def names(self) -> List[str]:
        return sorted(self.entries)

    def shape(self, name: str) -> Tuple[int, ...]:
        return self.entries[name].shape

    def _data_path(self, shard: int) -> str:
        return f"{self.prefix}.data-{shard:05d}-of-{self.num_shards:05d}"

    def get(self, name: str, verify_crc: bool = False) -> np.ndarray:
        e = self.entries[name]
        with open(self._data_path(e.shard), "rb") as f:
            f.seek(e.offset)
            raw = f.read(e.size)
        if len(raw) != e.size:
            raise ValueError(f"{name}: truncated data file")
        if verify_crc and e.crc and masked_crc32c(raw) != e.crc:
            raise ValueError(f"{name}: CRC32C mismatch")
        return np.frombuffer(raw, dtype=e.dtype).reshape(e.shape).copy()


def load_q_estimator(prefix: str, scope: str = "qlearn") -> Dict[str, np.ndarray]:
    """LSTM-DQN weights of one ``Q_Estimator_Nonlinear`` scope (reference myasp-A1.py:165-179).

    Returns kernel [I+H,4H] (gate column blocks i|j|f|o), bias [4H], w_out [H,A], b_out [A].
    The head variables are created unnamed (``tf.Variable``), hence ``Variable``/``Variable_1``;
    the legacy AdaptiveTauRL runs named them ``W_out``/``b_out``.
    """
    b = TFBundle(prefix)
    kernel = b.get(f"{scope}/rnn/lstm_cell/kernel")
    bias = b.get(f"{scope}/rnn/lstm_cell/bias")
    if f"{scope}/Variable" in b.entries:
        w_out, b_out = b.get(f"{scope}/Variable"), b.get(f"{scope}/Variable_1")
    else:
        w_out, b_out = b.get(f"{scope}/W_out"), b.get(f"{scope}/b_out")
    return {"kernel": kernel, "bias": bias, "w_out": w_out, "b_out": b_out}


def load_vae_dense_chain(prefix: str):
    """The Keras VAE Dense layers that ride along in the Saver bundle.

    The layers are named ``dense_<n>`` with consecutive n (``dense`` = n 0); the chain is
    hidden..., mu, logvar, decoder hidden, decoder out (reference myasp-A1.py:87-99).  Returns the
    list of (kernel, bias) in creation order, or None when the bundle holds no Dense layers.
    """
    b = TFBundle(prefix)
    found = {}
    for name in b.entries:
        parts = name.split("/")
        if len(parts) == 2 and parts[1] == "kernel" and (parts[0] == "dense" or parts[0].startswith("dense_")):
            n = 0 if parts[0] == "dense" else int(parts[0].split("_")[1])
            found[n] = parts[0]
    if not found:
        return None
    return [(b.get(f"{found[n]}/kernel"), b.get(f"{found[n]}/bias")) for n in sorted(found)]

"""TFRecord files of preprocessed source / target examples (reference util/tfrecord.py:40-118)
without TensorFlow: the TFRecord framing, the ``tf.train.Example`` protobuf wire format and the reference's field
layout, in pure Python / NumPy.  Host-side I/O only; nothing here touches the GPU.

TFRecord framing (TensorFlow's ``record_writer.cc``): per record ``uint64 length | uint32 masked_crc32c(length) | data |
uint32 masked_crc32c(data)``, little endian, CRC-32C (Castagnoli), mask ``((crc >> 15 | crc << 17) + 0xa282ead8) mod 2^32``.

``tf.train.Example`` (``example.proto`` / ``feature.proto``): ``Example{features=1: Features}``,
``Features{feature=1: map<string, Feature>}`` (map entries ``{key=1, value=2}``), ``Feature{bytes_list=1 | float_list=2 |
int64_list=3}``, ``BytesList{value=1 repeated bytes}``, ``FloatList{value=1 packed float}``, ``Int64List{value=1 packed
varint}``.  The parser also accepts the unpacked encoding of the numeric lists.
"""
from __future__ import annotations

import struct
from typing import Dict, Iterator, List, Sequence, Union

import numpy as np

# ---- CRC-32C ------------------------------------------------------------------------------------------------------
_POLY = 0x82F63B78
_TABLE = []
for _i in range(256):
    _c = _i
    for _ in range(8):
        _c = (_c >> 1) ^ (_POLY if _c & 1 else 0)
    _TABLE.append(_c)


def crc32c(data: bytes) -> int:
    c = 0xFFFFFFFF
    for b in data:
        c = _TABLE[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def masked_crc32c(data: bytes) -> int:
    c = crc32c(data)
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


# ---- TFRecord framing ----------------------------------------------------------------------------------------------
def write_records(filename: str, records: Sequence[bytes]) -> None:
    with open(filename, "wb") as f:
        for data in records:
            head = struct.pack("<Q", len(data))
            f.write(head + struct.pack("<I", masked_crc32c(head)) + data + struct.pack("<I", masked_crc32c(data)))


def read_records(filename: str, check_crc: bool = True) -> Iterator[bytes]:
    with open(filename, "rb") as f:
        while True:
            head = f.read(8)
            if not head:
                return
            if len(head) != 8:
                raise ValueError(f"{filename}: truncated record header")
            (n,) = struct.unpack("<Q", head)
            (hc,) = struct.unpack("<I", f.read(4))
            data = f.read(n)
            tail = f.read(4)
            if len(data) != n or len(tail) != 4:
                raise ValueError(f"{filename}: truncated record")
            if check_crc and (hc != masked_crc32c(head) or struct.unpack("<I", tail)[0] != masked_crc32c(data)):
                raise ValueError(f"{filename}: corrupted record (crc mismatch)")
            yield data


# ---- protobuf wire format (the subset tf.train.Example uses) ----------------------------------------------------------
def _varint(n: int) -> bytes:
    n &= (1 << 64) - 1                       # int64 values are encoded as their 64-bit two's complement
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        out.append(b | (0x80 if n else 0))
        if not n:
            return bytes(out)


def _read_varint(buf: bytes, pos: int):
    shift = result = 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7
        if shift > 70:
            raise ValueError("malformed varint")


def _len_field(field: int, payload: bytes) -> bytes:
    return _varint((field << 3) | 2) + _varint(len(payload)) + payload


def _fields(buf: bytes):
    """Yield (field number, wire type, value) of one message; length-delimited values as bytes."""
    pos = 0
    while pos < len(buf):
        key, pos = _read_varint(buf, pos)
        field, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _read_varint(buf, pos)
        elif wt == 1:
            v, pos = buf[pos:pos + 8], pos + 8
        elif wt == 2:
            n, pos = _read_varint(buf, pos)
            v, pos = buf[pos:pos + n], pos + n
            if len(v) != n:
                raise ValueError("truncated length-delimited field")
        elif wt == 5:
            v, pos = buf[pos:pos + 4], pos + 4
        else:
            raise ValueError(f"unsupported wire type {wt}")
        yield field, wt, v


FeatureValue = Union[List[bytes], np.ndarray]


def encode_example(features: Dict[str, FeatureValue]) -> bytes:
    """``tf.train.Example(features=...).SerializeToString()``: a list of ``bytes`` -> BytesList, an integer array ->
    Int64List, a floating array -> FloatList.  Map entries are written in sorted key order (deterministic output)."""
    entries = b""
    for key in sorted(features):
        v = features[key]
        if isinstance(v, (list, tuple)) and all(isinstance(x, (bytes, bytearray)) for x in v):
            feat = _len_field(1, b"".join(_len_field(1, bytes(x)) for x in v))
        else:
            arr = np.asarray(v)
            if arr.dtype.kind in "iu" or arr.dtype.kind == "b":
                packed = b"".join(_varint(int(x)) for x in arr.reshape(-1))
                feat = _len_field(3, _len_field(1, packed) if arr.size else b"")
            elif arr.dtype.kind == "f":
                feat = _len_field(2, _len_field(1, arr.astype("<f4").tobytes()) if arr.size else b"")
            else:
                raise TypeError(f"feature {key}: unsupported value type {arr.dtype}")
        entries += _len_field(1, _len_field(1, key.encode("utf-8")) + _len_field(2, feat))
    return _len_field(1, entries)


def decode_example(data: bytes) -> Dict[str, FeatureValue]:
    out: Dict[str, FeatureValue] = {}
    for f, wt, features in _fields(data):
        if f != 1 or wt != 2:
            continue
        for f2, wt2, entry in _fields(features):
            if f2 != 1 or wt2 != 2:
                continue
            key, feat = None, b""
            for f3, _, v in _fields(entry):
                if f3 == 1:
                    key = v.decode("utf-8")
                elif f3 == 2:
                    feat = v
            if key is None:
                raise ValueError("map entry without a key")
            value: FeatureValue = []
            for kind, _, lst in _fields(feat):
                if kind == 1:
                    value = [bytes(v) for f4, _, v in _fields(lst) if f4 == 1]
                elif kind == 2:
                    vals: List[float] = []
                    for f4, wt4, v in _fields(lst):
                        if f4 == 1:
                            vals.extend(np.frombuffer(v, "<f4").tolist() if wt4 in (2, 5) else [])
                    value = np.asarray(vals, np.float32)
                elif kind == 3:
                    ints: List[int] = []
                    for f4, wt4, v in _fields(lst):
                        if f4 != 1:
                            continue
                        if wt4 == 0:
                            ints.append(v)
                        else:
                            p = 0
                            while p < len(v):
                                x, p = _read_varint(v, p)
                                ints.append(x)
                    value = np.asarray([x - (1 << 64) if x >= (1 << 63) else x for x in ints], np.int64)
            out[key] = value
    return out


# ---- the reference's examples (util/tfrecord.py) ---------------------------------------------------------------------
def _scalar(ex, key):
    v = ex[key]
    if not isinstance(v, np.ndarray) or v.shape != (1,):
        raise ValueError(f"feature {key}: expected one int64 (FixedLenFeature((), tf.int64))")
    return int(v[0])


def write_preprocessed_target_data(id: int, spec: np.ndarray, mel: np.ndarray, filename: str) -> None:
    """util/tfrecord.py:40-51: raw float32 bytes of spec / mel plus their widths and the frame count."""
    spec, mel = np.ascontiguousarray(spec, np.float32), np.ascontiguousarray(mel, np.float32)
    write_records(filename, [encode_example({
        "id": np.array([id], np.int64), "spec": [spec.tobytes()], "spec_width": np.array([spec.shape[1]], np.int64),
        "mel": [mel.tobytes()], "mel_width": np.array([mel.shape[1]], np.int64), "target_length": np.array([len(mel)], np.int64)})])


def write_preprocessed_source_data2(id: int, text1: str, source1: np.ndarray, text2: str, source2: np.ndarray, filename: str) -> None:
    """util/tfrecord.py:54-64: two (text, int64 id sequence) pairs per example."""
    s1, s2 = np.ascontiguousarray(source1, np.int64), np.ascontiguousarray(source2, np.int64)
    write_records(filename, [encode_example({
        "id": np.array([id], np.int64), "text": [text1.encode("utf-8"), text2.encode("utf-8")],
        "source": [s1.tobytes(), s2.tobytes()], "source_length": np.array([len(s1), len(s2)], np.int64)})])


def decode_preprocessed_target_data(record: bytes) -> Dict[str, object]:
    """parse_preprocessed_target_data + decode_preprocessed_target_data (util/tfrecord.py:67-94)."""
    ex = decode_example(record)
    n, sw, mw = _scalar(ex, "target_length"), _scalar(ex, "spec_width"), _scalar(ex, "mel_width")
    spec = np.frombuffer(ex["spec"][0], "<f4").reshape(n, sw)
    mel = np.frombuffer(ex["mel"][0], "<f4").reshape(n, mw)
    return {"id": _scalar(ex, "id"), "spec": spec, "spec_width": sw, "mel": mel, "mel_width": mw, "target_length": n}


def decode_preprocessed_source_data(record: bytes) -> Dict[str, object]:
    """parse_preprocessed_source_data + decode_preprocessed_source_data (util/tfrecord.py:97-118)."""
    ex = decode_example(record)
    text, source, length = ex["text"], ex["source"], ex["source_length"]
    if len(text) != 2 or len(source) != 2 or length.shape != (2,):
        raise ValueError("source example: text, source and source_length must hold two entries (FixedLenFeature((2), ...))")
    return {"id": _scalar(ex, "id"), "text": text[0].decode("utf-8"), "source": np.frombuffer(source[0], "<i8"),
            "source_length": int(length[0]), "text2": text[1].decode("utf-8"), "source2": np.frombuffer(source[1], "<i8"),
            "source_length2": int(length[1])}


def read_examples(source_files: Sequence[str], target_files: Sequence[str]):
    """Pairs (source dict, mel [T, M]) in file order, ids asserted equal (datasets/dataset.py:88-100): the input of
    ``datasets.group_by_batch``."""
    for sf, tf_ in zip(source_files, target_files):
        for s_rec, t_rec in zip(read_records(sf), read_records(tf_)):
            s, t = decode_preprocessed_source_data(s_rec), decode_preprocessed_target_data(t_rec)
            if s["id"] != t["id"]:
                raise ValueError(f"source id {s['id']} != target id {t['id']} ({sf}, {tf_})")
            yield s, t["mel"]

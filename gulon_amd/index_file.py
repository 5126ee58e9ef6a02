"""Gulon's on-disk index format (core/src/main/protobuf/index.proto, Index.toProtobuf /
fromProtobuf at Index.scala:151-207, EncodedMatrix.scala:38-51, ProductQuantizer.scala:88-105),
read and written with a hand-rolled proto2 wire codec: no protoc, no generated code.

Writing follows scalapb's defaults for this schema: fields in field-number order, `repeated float`
/ `repeated int32` UNPACKED (proto2), required fields always present.  Reading accepts packed and
unpacked repeated scalars and skips unknown fields.  The key strings stay on the host; the vector
side becomes a device index (`load_index`).
"""
import struct
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

from .coder import Coder
from .kmeans import KMeans
from .product_quantizer import EncodedMatrix, ProductQuantizer, Quantizer

VARINT, FIXED64, LEN, FIXED32 = 0, 1, 2, 5
LIMIT_GROUPS, LIMIT_VECTORS = 0, 2          # GroupedIndex.Strategy (index.proto:52-55)
METRICS = {0: "l2", 1: "cosine"}


# ------------------------------------------------------------------ wire primitives
def _varint(v):
    v &= (1 << 64) - 1                       # negative int32/enum are sign-extended to 64 bits
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        out.append(b | (0x80 if v else 0))
        if not v:
            return bytes(out)


def _tag(field_no, wire):
    return _varint((field_no << 3) | wire)


def _len_field(field_no, payload):
    return _tag(field_no, LEN) + _varint(len(payload)) + payload


def _int_field(field_no, v):
    return _tag(field_no, VARINT) + _varint(int(v))


def _read_varint(buf, pos):
    shift = v = 0
    while True:
        if pos >= len(buf):
            raise ValueError("truncated varint")
        b = buf[pos]
        pos += 1
        v |= (b & 0x7F) << shift
        if not b & 0x80:
            return v, pos
        shift += 7
        if shift > 70:
            raise ValueError("varint too long")


def _int32(v):
    v &= 0xFFFFFFFF
    return v - (1 << 32) if v & 0x80000000 else v


def _fields(buf):
    """Yields (field number, wire type, value): int for VARINT/FIXED*, memoryview for LEN."""
    buf = memoryview(buf)
    pos = 0
    while pos < len(buf):
        key, pos = _read_varint(buf, pos)
        no, wire = key >> 3, key & 7
        if wire == VARINT:
            v, pos = _read_varint(buf, pos)
        elif wire == FIXED64:
            v, pos = bytes(buf[pos:pos + 8]), pos + 8
        elif wire == FIXED32:
            v, pos = bytes(buf[pos:pos + 4]), pos + 4
        elif wire == LEN:
            n, pos = _read_varint(buf, pos)
            if pos + n > len(buf):
                raise ValueError("truncated length-delimited field")
            v, pos = buf[pos:pos + n], pos + n
        else:
            raise ValueError(f"unsupported wire type {wire}")
        if pos > len(buf):
            raise ValueError("truncated field")
        yield no, wire, v


# ------------------------------------------------------------------ messages
def _enc_float_vector(values):
    return b"".join(_tag(1, FIXED32) + struct.pack("<f", float(x)) for x in np.asarray(values, np.float32))


def _dec_float_vector(buf):
    out = []
    for no, wire, v in _fields(buf):
        if no == 1 and wire == FIXED32:
            out.append(struct.unpack("<f", v)[0])
        elif no == 1 and wire == LEN:                         # packed
            out.extend(np.frombuffer(bytes(v), "<f4").tolist())
    return np.asarray(out, np.float32)


def _enc_pq(pq: ProductQuantizer):
    out = _int_field(1, pq.num_clusters)
    for q in pq.quantizers:
        body = _int_field(1, q.frm) + _int_field(2, q.dimension)
        for c in q.clusters.centroids:
            body += _len_field(3, _enc_float_vector(c))
        out += _len_field(2, body)
    return out


def _dec_pq(buf):
    num_clusters, quantizers = None, []
    for no, wire, v in _fields(buf):
        if no == 1:
            num_clusters = _int32(v)
        elif no == 2:
            start = dim = None
            cents = []
            for n2, w2, v2 in _fields(v):
                if n2 == 1:
                    start = _int32(v2)
                elif n2 == 2:
                    dim = _int32(v2)
                elif n2 == 3:
                    cents.append(_dec_float_vector(v2))
            if start is None or dim is None:
                raise ValueError("ProductQuantizer.Quantizer: missing required field")
            c = np.stack(cents).astype(np.float32) if cents else np.zeros((0, dim), np.float32)
            quantizers.append(Quantizer(start, KMeans(dim, c)))
    if num_clusters is None:
        raise ValueError("ProductQuantizer: missing num_clusters")
    return ProductQuantizer(num_clusters, quantizers)


def _enc_encoded(em: EncodedMatrix):
    out = _int_field(1, em.coder.width) + _int_field(2, em.coder.length)
    for e in em.encodings:
        out += _len_field(3, bytes(np.ascontiguousarray(e, np.uint8)))
    return out


def _dec_encoded(buf):
    width = length = None
    enc = []
    for no, wire, v in _fields(buf):
        if no == 1:
            width = _int32(v)
        elif no == 2:
            length = _int32(v)
        elif no == 3:
            enc.append(np.frombuffer(bytes(v), np.uint8).copy())
    if width is None or length is None:
        raise ValueError("EncodedMatrix: missing required field")
    coder = Coder(width, length)                              # "unsupported width" -> ValueError, Coder.scala:57
    for e in enc:
        if len(e) != coder.bytes_per_code:
            raise ValueError("EncodedMatrix: encoding length does not match the coder")
    return EncodedMatrix(coder, enc)


def _enc_pq_index(pq, em):
    return _len_field(1, _enc_pq(pq)) + _len_field(2, _enc_encoded(em))


def _dec_pq_index(buf):
    pq = em = None
    for no, wire, v in _fields(buf):
        if no == 1:
            pq = _dec_pq(v)
        elif no == 2:
            em = _dec_encoded(v)
    if pq is None or em is None:
        raise ValueError("PQIndex: missing required field")
    return pq, em


@dataclass
class IndexFile:
    """The content of one protobuf.Index message."""
    kind: str                                   # "sorted" | "grouped"
    words: List[str]
    quantizer: ProductQuantizer
    data: EncodedMatrix
    metric: str = "l2"
    centroids: Optional[np.ndarray] = None      # grouped: [g][d]
    offsets: Optional[np.ndarray] = None        # grouped: [g-1]
    strategy: int = LIMIT_GROUPS                # grouped
    limit: int = 0
    extras: dict = field(default_factory=dict)


def dumps(f: IndexFile) -> bytes:
    """Index.toProtobuf(...).toByteArray (Index.scala:151-175)."""
    metric = {v: k for k, v in METRICS.items()}[f.metric]
    body = b"".join(_len_field(1, w.encode("utf-8")) for w in f.words)
    body += _len_field(2, _enc_pq_index(f.quantizer, f.data)) + _int_field(3, metric)
    if f.kind == "sorted":
        return _len_field(1, body)
    if f.kind != "grouped":
        raise ValueError("missing index implementation")
    for c in np.asarray(f.centroids, np.float32):
        body += _len_field(4, _enc_float_vector(c))
    for o in np.asarray(f.offsets, np.int32):
        body += _int_field(5, int(o))
    body += _int_field(6, f.strategy) + _int_field(7, f.limit)
    return _len_field(2, body)


def loads(buf: bytes) -> IndexFile:
    """Index.fromProtobuf (Index.scala:177-207); raises ValueError where the reference throws
    IllegalArgumentException."""
    impl = None
    for no, wire, v in _fields(buf):
        if no in (1, 2) and wire == LEN:
            impl = (no, v)                                    # oneof: the last one wins
    if impl is None:
        raise ValueError("missing index implementation")
    no, body = impl
    words, pqi, metric = [], None, None
    cents, offs, strategy, limit = [], [], None, None
    for n2, w2, v2 in _fields(body):
        if n2 == 1:
            words.append(bytes(v2).decode("utf-8"))
        elif n2 == 2:
            pqi = _dec_pq_index(v2)
        elif n2 == 3:
            metric = _int32(v2)
        elif no == 2 and n2 == 4:
            cents.append(_dec_float_vector(v2))
        elif no == 2 and n2 == 5:
            if w2 == LEN:                                     # packed int32
                p, mv = 0, v2
                while p < len(mv):
                    x, p = _read_varint(mv, p)
                    offs.append(_int32(x))
            else:
                offs.append(_int32(v2))
        elif no == 2 and n2 == 6:
            strategy = _int32(v2)
        elif no == 2 and n2 == 7:
            limit = _int32(v2)
    if pqi is None or metric is None:
        raise ValueError("index: missing required field")
    if metric not in METRICS:
        raise ValueError("unrecognized metric")
    if no == 1:
        return IndexFile("sorted", words, pqi[0], pqi[1], METRICS[metric])
    if strategy not in (LIMIT_GROUPS, LIMIT_VECTORS):
        raise ValueError("strategy must be one of LIMIT_GROUPS or LIMIT_VECTORS")
    if limit is None:
        raise ValueError("GroupedIndex: missing limit")
    d = pqi[0].dimension
    c = np.stack(cents).astype(np.float32) if cents else np.zeros((0, d), np.float32)
    return IndexFile("grouped", words, pqi[0], pqi[1], METRICS[metric], c, np.asarray(offs, np.int32), strategy, limit)


# ------------------------------------------------------------------ device indexes
def load_index(buf: bytes):
    """bytes of a protobuf.Index -> (words, device index): SortedIndex or GroupedIndex on the GPU."""
    from .grouped import GroupedIndex, LimitGroups, LimitVectors
    from .index import PQIndex, SortedIndex
    f = loads(buf)
    if f.kind == "sorted":
        return f.words, SortedIndex(PQIndex(f.quantizer, f.data), f.metric)
    strat = LimitGroups(f.limit) if f.strategy == LIMIT_GROUPS else LimitVectors(f.limit)
    return f.words, GroupedIndex(f.quantizer, f.data, f.centroids, f.offsets, strat, f.metric)


def dump_index(index, words) -> bytes:
    """Device index (+ its key strings) -> bytes of a protobuf.Index."""
    from .grouped import GroupedIndex, LimitGroups
    from .index import SortedIndex
    if isinstance(index, SortedIndex):
        vi = index.vector_index
        return dumps(IndexFile("sorted", list(words), vi.product_quantizer, vi.data, index.metric))
    if isinstance(index, GroupedIndex):
        strat = LIMIT_GROUPS if isinstance(index.strategy, LimitGroups) else LIMIT_VECTORS
        return dumps(IndexFile("grouped", list(words), index.quantizer, index.data, index.metric, index.centroids,
                               index.offsets, strat, index.strategy.count))
    raise ValueError("missing index implementation")

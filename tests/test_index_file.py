"""The on-disk index format (index.proto): hand-rolled proto2 codec vs hand-computed wire bytes,
vs google.protobuf on the same schema (test-only dependency, built from a descriptor at run time),
round trips, and the reference's error cases (Index.scala:177-207)."""
import struct

import numpy as np
import pytest

from gulon_amd import index_file as F
from gulon_amd.coder import Coder
from gulon_amd.kmeans import KMeans
from gulon_amd.product_quantizer import EncodedMatrix, ProductQuantizer, Quantizer


def _pq(rng, d=6, m=2, k=4):
    s = d // m
    return ProductQuantizer(k, [Quantizer(j * s, KMeans(s, rng.standard_normal((k, s)).astype(np.float32)))
                                for j in range(m)])


def _em(rng, n, m, k):
    import gulon_amd.native  # noqa: F401  (Coder needs the library for width -> bytes)
    from gulon_amd.coder import width_for_clusters
    coder = Coder(width_for_clusters(k), n)
    return EncodedMatrix(coder, [coder.build_code(rng.integers(0, k, n)) for _ in range(m)])


def _schema():
    """index.proto (minus the scalapb options) as google.protobuf message classes."""
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
    fd = descriptor_pb2.FileDescriptorProto(name="index.proto", package="gulon", syntax="proto2")
    T = descriptor_pb2.FieldDescriptorProto

    def msg(parent, name):
        m = parent.message_type.add() if hasattr(parent, "message_type") else parent.nested_type.add()
        m.name = name
        return m

    def fld(m, name, no, typ, label, type_name=None):
        f = m.field.add(name=name, number=no, type=typ, label=label)
        if type_name:
            f.type_name = type_name
        return f
    REQ, REP, OPT = T.LABEL_REQUIRED, T.LABEL_REPEATED, T.LABEL_OPTIONAL
    fv = msg(fd, "FloatVector"); fld(fv, "values", 1, T.TYPE_FLOAT, REP)
    pq = msg(fd, "ProductQuantizer")
    fld(pq, "num_clusters", 1, T.TYPE_INT32, REQ); fld(pq, "quantizers", 2, T.TYPE_MESSAGE, REP, ".gulon.ProductQuantizer.Quantizer")
    qz = msg(pq, "Quantizer")
    fld(qz, "start_index", 1, T.TYPE_INT32, REQ); fld(qz, "dimension", 2, T.TYPE_INT32, REQ)
    fld(qz, "centroids", 3, T.TYPE_MESSAGE, REP, ".gulon.FloatVector")
    em = msg(fd, "EncodedMatrix")
    fld(em, "code_width", 1, T.TYPE_INT32, REQ); fld(em, "length", 2, T.TYPE_INT32, REQ); fld(em, "encodings", 3, T.TYPE_BYTES, REP)
    en = fd.enum_type.add(name="Metric"); en.value.add(name="L2", number=0); en.value.add(name="COSINE", number=1)
    pi = msg(fd, "PQIndex")
    fld(pi, "product_quantizer", 1, T.TYPE_MESSAGE, REQ, ".gulon.ProductQuantizer"); fld(pi, "data", 2, T.TYPE_MESSAGE, REQ, ".gulon.EncodedMatrix")
    si = msg(fd, "SortedIndex")
    fld(si, "sorted_words", 1, T.TYPE_STRING, REP); fld(si, "vector_index", 2, T.TYPE_MESSAGE, REQ, ".gulon.PQIndex")
    fld(si, "metric", 3, T.TYPE_ENUM, REQ, ".gulon.Metric")
    gi = msg(fd, "GroupedIndex")
    fld(gi, "grouped_words", 1, T.TYPE_STRING, REP); fld(gi, "vector_index", 2, T.TYPE_MESSAGE, REQ, ".gulon.PQIndex")
    fld(gi, "metric", 3, T.TYPE_ENUM, REQ, ".gulon.Metric"); fld(gi, "centroids", 4, T.TYPE_MESSAGE, REP, ".gulon.FloatVector")
    fld(gi, "offsets", 5, T.TYPE_INT32, REP)
    st = gi.enum_type.add(name="Strategy"); st.value.add(name="LIMIT_GROUPS", number=0); st.value.add(name="LIMIT_VECTORS", number=2)
    fld(gi, "strategy", 6, T.TYPE_ENUM, REQ, ".gulon.GroupedIndex.Strategy"); fld(gi, "limit", 7, T.TYPE_INT32, REQ)
    ix = msg(fd, "Index")
    ix.oneof_decl.add(name="implementation")
    fld(ix, "sorted", 1, T.TYPE_MESSAGE, OPT, ".gulon.SortedIndex").oneof_index = 0
    fld(ix, "grouped", 2, T.TYPE_MESSAGE, OPT, ".gulon.GroupedIndex").oneof_index = 0
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    return message_factory.GetMessageClass(pool.FindMessageTypeByName("gulon.Index"))


def test_wire_bytes_known_answers():
    assert F._varint(0) == b"\x00" and F._varint(300) == b"\xac\x02"
    assert F._varint(-1) == b"\xff" * 9 + b"\x01"                        # int32 -1: 10-byte varint
    assert F._enc_float_vector([1.0, -2.5]) == b"\x0d" + struct.pack("<f", 1.0) + b"\x0d" + struct.pack("<f", -2.5)
    assert F._int_field(7, 5) == b"\x38\x05" and F._len_field(3, b"ab") == b"\x1a\x02ab"
    assert np.array_equal(F._dec_float_vector(b"\x0a\x08" + struct.pack("<2f", 3.0, 4.0)), [3.0, 4.0])   # packed form


@pytest.mark.parametrize("kind", ["sorted", "grouped"])
def test_round_trip_and_google_protobuf_agree(kind):
    rng = np.random.default_rng(3)
    n, d, m, k = 37, 6, 2, 4
    pq, em = _pq(rng, d, m, k), _em(rng, n, m, k)
    words = [f"wörd{i}" for i in range(n)]
    if kind == "sorted":
        f = F.IndexFile("sorted", words, pq, em, "cosine")
    else:
        f = F.IndexFile("grouped", words, pq, em, "l2", rng.standard_normal((3, d)).astype(np.float32),
                        np.array([10, 25], np.int32), F.LIMIT_VECTORS, 123)
    blob = F.dumps(f)
    g = F.loads(blob)
    assert g.kind == kind and g.words == words and g.metric == f.metric
    assert g.data == em and g.quantizer.num_clusters == k
    for a, b in zip(g.quantizer.quantizers, pq.quantizers):
        assert a.frm == b.frm and a.clusters == b.clusters
    if kind == "grouped":
        assert np.array_equal(g.centroids, f.centroids) and g.offsets.tolist() == [10, 25]
        assert (g.strategy, g.limit) == (F.LIMIT_VECTORS, 123)
    # the same message through google.protobuf: parses ours, and its own serialisation parses back here
    Index = _schema()
    msg = Index()
    msg.ParseFromString(blob)
    body = getattr(msg, kind)
    assert list(body.sorted_words if kind == "sorted" else body.grouped_words) == words
    assert body.vector_index.data.length == n and len(body.vector_index.product_quantizer.quantizers) == m
    assert msg.SerializeToString() == blob                     # same field order, unpacked repeated scalars
    h = F.loads(msg.SerializeToString())
    assert h.words == words and h.data == em


def test_reference_error_cases():
    with pytest.raises(ValueError, match="missing index implementation"):
        F.loads(b"")
    rng = np.random.default_rng(1)
    pq, em = _pq(rng), _em(rng, 5, 2, 4)
    good = F.dumps(F.IndexFile("grouped", ["a"] * 5, pq, em, "l2", np.zeros((1, 6), np.float32), np.zeros(0, np.int32),
                               F.LIMIT_GROUPS, 2))
    bad = good[:-4] + F._int_field(6, 1) + F._int_field(7, 2)    # strategy 1 is not in the enum
    with pytest.raises(ValueError, match="strategy must be one of"):
        F.loads(bad)
    with pytest.raises(ValueError):
        F.loads(good[:-1] + b"\x80")                              # truncated varint


@pytest.mark.gpu
def test_written_index_loads_onto_the_gpu_and_answers_the_same():
    import gulon_amd as g
    rng = np.random.default_rng(8)
    n, d, m, k, K = 6000, 16, 4, 32, 5
    X = (rng.standard_normal((n, d)) + 3 * rng.integers(0, 3, (n, 1))).astype(np.float32)
    dm = g.DeviceMatrix.from_host(X)
    words = [f"w{i:05d}" for i in range(n)]
    pq = g.ProductQuantizer.apply(dm, g.ProductQuantizerConfig(k, m, 3))
    sorted_index = g.Index.sorted(dm, pq, "cosine")
    w2, again = F.load_index(F.dump_index(sorted_index, words))
    Q = X[:9]
    a, b = sorted_index.batch_query(K, Q), again.batch_query(K, Q)
    assert w2 == words and all(x.rows.tolist() == y.rows.tolist() for x, y in zip(a, b))
    coarse = g.KMeans.compute_clusters(g.Vectors(dm), g.KMeansConfig(7, 3))
    gv = g.group(dm, coarse)
    rpq = g.ProductQuantizer.apply(gv.residuals, g.ProductQuantizerConfig(k, m, 3))
    gi = g.Index.grouped(gv, rpq, g.LimitVectors(900))
    w3, gagain = F.load_index(F.dump_index(gi, [words[i] for i in gv.perm]))
    a, b = gi.batch_query(K, Q), gagain.batch_query(K, Q)
    assert all(x.rows.tolist() == y.rows.tolist() and np.array_equal(x.distances, y.distances) for x, y in zip(a, b))
    assert w3[0] == words[gv.perm[0]]

"""KeyIndex and the word2vec text reader (gulon_amd/word_vectors.py; KeyIndex.scala, WordVectors.scala:141-252):
host logic, checked against independent restatements (exact rational rounding, UTF-16 order)."""
import random
import struct
from fractions import Fraction

import numpy as np
import pytest

from gulon_amd import word_vectors as W


def _nearest_f32(fr):
    """Round a Fraction to binary32, ties to even, by brute force on the neighbours of a first guess."""
    g = np.float32(float(fr))
    with np.errstate(over="ignore"):
        cands = {float(g), float(np.nextafter(g, np.float32(np.inf))), float(np.nextafter(g, np.float32(-np.inf)))}
    best = None
    for c in sorted(cands):
        if c in (float('inf'), float('-inf')):
            continue
        err = abs(Fraction(c) - fr)
        even = (struct.unpack("<I", struct.pack("<f", c))[0] & 1) == 0
        if best is None or err < best[0] or (err == best[0] and even and not best[2]):
            best = (err, c, even)
    return np.float32(best[1])


def test_parse_float_is_correctly_rounded():
    rnd = random.Random(5)
    toks = ["0.1", "1e-3", "-2.5", "3", "1.000000059604644775390625", "1.00000005960464477539062500001",
            "1.00000005960464477539062499999", "16777217", "16777219", "-16777217.0", "0.30000001192092896",
            "1.1754943508222875e-38", "7.0064923216240854e-46", "3.4028235e38"]
    for _ in range(3000):
        mant = rnd.randrange(1 << 23, 1 << 24)
        e = rnd.randrange(-30, 30)
        mid = Fraction(2 * mant + 1, 2) * Fraction(2) ** e          # exact midpoint between two floats
        delta = Fraction(rnd.choice([-1, 0, 1]), 10 ** 40)
        v = mid + delta
        toks.append(f"{v.numerator * 10 ** 45 // v.denominator}e-45")
        toks.append(repr(rnd.uniform(-10, 10)))
    for t in toks:
        want = _nearest_f32(Fraction(t))
        got = W.parse_float(t)
        assert got.view(np.uint32) == want.view(np.uint32), t


def test_read_word2vec_with_and_without_header():
    a = W.read_word2vec_text("3 2\nb 1.5 -2\na 0.1 3e-2\nc 4 5\n")
    b = W.read_word2vec_text("b 1.5 -2\na 0.1 3e-2\n\nc 4 5")          # no header, an empty line, no final newline
    for wv in (a, b):
        assert wv.words == ["b", "a", "c"] and wv.dimension == 2 and wv.size == 3
        assert np.array_equal(wv.data, np.array([[1.5, -2], [0.1, 0.03], [4, 5]], np.float32))
    assert W.read_word2vec_text("").size == 0


def test_read_word2vec_normalize_uses_the_reference_arithmetic(tmp_path):
    from gulon_amd.index import normalize
    p = tmp_path / "v.txt"
    p.write_text("x 3 4 12\ny 1e-3 2e-3 5\n", encoding="utf-8")
    wv = W.read_word2vec(str(p), normalize=True)
    for i, row in enumerate([[3, 4, 12], [1e-3, 2e-3, 5]]):
        assert np.array_equal(wv[i].view(np.uint32), normalize(np.array(row, np.float32)).view(np.uint32))


def test_key_order_is_utf16_code_units_like_string_compareTo():
    bmp_high, astral = "￿", "\U00010000"              # code points: bmp_high < astral; UTF-16 units: D800 DC00 < FFFF
    wv = W.WordVectors([bmp_high, "b", astral, "a"], np.arange(8, dtype=np.float32).reshape(4, 2)).sorted()
    assert wv.words == ["a", "b", astral, bmp_high]
    assert np.array_equal(wv.data[:, 0], np.array([6, 2, 4, 0], np.float32))
    for i, w in enumerate(wv.words):
        assert wv.key_index.lookup(w) == i
    assert wv.key_index.lookup("c") is None and wv.key_index.lookup("") is None


def test_grouped_key_index_searches_group_by_group():
    keys = ["b", "d", "a", "c", "e", "a2"]                   # groups [b d] [a c e] [a2]
    ki = W.KeyIndexGrouped(keys, [2, 5])
    assert [ki.lookup(k) for k in keys] == list(range(6))
    assert ki.lookup("zz") is None and len(ki) == 6 and ki[3] == "c"


@pytest.mark.gpu
def test_grouped_word_vectors_order_and_keyed_queries():
    import gulon_amd as g
    rng = np.random.default_rng(0)
    n, d = 3000, 8
    X = (rng.standard_normal((n, d)) + 4.0 * rng.integers(0, 3, (n, 1))).astype(np.float32)
    words = [f"w{int(i):05d}" for i in rng.permutation(n)]
    wv = W.WordVectors(words, X)
    clustering = g.KMeans.compute_clusters(g.Vectors(g.DeviceMatrix.from_host(wv.sorted().data)), g.KMeansConfig(5, 3))
    gw, gv = wv.grouped(clustering)
    # rows ordered by (cluster, word), WordVectors.scala:27-29
    assign = clustering.par_assign(g.Vectors(g.DeviceMatrix.from_host(gw.data)))
    assert (np.diff(assign) >= 0).all()
    bounds = [0] + gw.offsets.tolist() + [n]
    for lo, hi in zip(bounds[:-1], bounds[1:]):
        assert gw.words[lo:hi] == sorted(gw.words[lo:hi])
    by_word = dict(zip(words, X))
    for i in (0, 17, n - 1):
        assert np.array_equal(gw[i], by_word[gw.word(i)]) and gw.key_index.lookup(gw.word(i)) == i
        assert lo <= n and gw.cluster_of(i) == int(np.searchsorted(gw.offsets, i, side="right"))
    pq = g.ProductQuantizer.apply(gv.residuals, g.ProductQuantizerConfig(16, 4, 3))
    index = W.KeyedIndex(g.Index.grouped(gv, pq, g.LimitGroups(5)), gw)
    res = index.query_word(3, gw.word(10))
    assert len(res) == 3 and all(isinstance(w, str) for w, _ in res)
    assert index.query_word(3, "no such word") is None

"""Keys and word2vec text ingest: KeyIndex (KeyIndex.scala:9-62) and
WordVectors.readWord2Vec / sorted / grouped (WordVectors.scala:13-110,141-268).

Host-side conveniences: the native path works on row ids; these classes carry the words next to
them so that a caller of the reference's CLI finds `query(word)` / `(word, distance)` results.
Strings order as on the JVM: String.compareTo compares UTF-16 code units, which differs from
Python's code-point order for characters beyond the BMP."""
import bisect
import io
import re
from fractions import Fraction

import numpy as np

from . import native as N


def _jkey(s):
    """Sort key with java.lang.String.compareTo's order (UTF-16 code units)."""
    return s.encode("utf-16-be", "surrogatepass")


class KeyIndexSorted:
    """KeyIndex.Sorted (KeyIndex.scala:15-29): binary search over keys in String order."""

    def __init__(self, keys):
        self.keys = list(keys)
        self._jk = [_jkey(k) for k in self.keys]

    def __len__(self):
        return len(self.keys)

    def __getitem__(self, i):
        return self.keys[i]

    def lookup(self, key):
        i = bisect.bisect_left(self._jk, _jkey(key))
        return i if i < len(self._jk) and self._jk[i] == _jkey(key) else None


class KeyIndexGrouped:
    """KeyIndex.Grouped (KeyIndex.scala:31-61): keys sorted inside each group; groups tried in order."""

    def __init__(self, keys, group_offsets):
        self.keys = list(keys)
        self.group_offsets = [int(o) for o in group_offsets]
        self._jk = [_jkey(k) for k in self.keys]

    def __len__(self):
        return len(self.keys)

    def __getitem__(self, i):
        return self.keys[i]

    def lookup(self, key):
        jk, frm = _jkey(key), 0
        for to in self.group_offsets + [len(self.keys)]:
            i = bisect.bisect_left(self._jk, jk, frm, to)
            if i < to and self._jk[i] == jk:
                return i
            frm = to
        return None


def parse_float(token):
    """java.lang.Float.parseFloat for the decimal tokens of a word2vec file: the binary32 nearest to
    the decimal value.  float(token) is the nearest binary64; casting that down rounds a second time,
    which is wrong only when the binary64 sits exactly on a binary32 midpoint -- those tokens are
    rounded exactly."""
    x = float(token)
    f = np.float32(x)
    if x != x or x in (float("inf"), float("-inf")) or f in (np.float32("inf"), np.float32("-inf")):
        return f
    # bracket x by neighbouring binary32 values (differences of doubles this close are exact)
    with np.errstate(over="ignore"):
        lo = f if float(f) <= x else np.nextafter(f, np.float32("-inf"))
        hi = np.nextafter(lo, np.float32("inf"))
    if x - float(lo) != float(hi) - x:                 # not a binary32 midpoint: one rounding is enough
        return f
    try:
        exact = Fraction(token.strip())
    except (ValueError, ZeroDivisionError):
        return f
    dl, dh = exact - Fraction(float(lo)), Fraction(float(hi)) - exact
    if dl < dh:
        return np.float32(lo)
    if dh < dl:
        return np.float32(hi)
    return np.float32(lo) if (int(np.float32(lo).view(np.uint32)) & 1) == 0 else np.float32(hi)   # ties to even


_HEADER = re.compile(r"(\d+) (\d+)")


class WordVectors:
    """WordVectors.Unindexed / Sorted (WordVectors.scala:112-126): words[i] belongs to row i."""

    def __init__(self, words, data, key_index=None):
        self.words = list(words)
        self.data = N.f32(data).reshape(len(self.words), -1) if len(self.words) else np.zeros((0, 0), np.float32)
        self.key_index = key_index

    @property
    def size(self):
        return len(self.words)

    @property
    def dimension(self):
        return self.data.shape[1]

    def word(self, i):
        return self.words[i]

    def __getitem__(self, i):
        return self.data[i]

    def sorted(self):
        """WordVectors.sorted (WordVectors.scala:60-71): rows in String order of their words."""
        order = sorted(range(self.size), key=lambda i: _jkey(self.words[i]))
        words = [self.words[i] for i in order]
        return WordVectors(words, self.data[order] if self.size else self.data, KeyIndexSorted(words))

    def grouped(self, clustering):
        """WordVectors.grouped (WordVectors.scala:24-58): rows ordered by (cluster, word); returns
        (GroupedWordVectors, gulon_amd.grouped.GroupedVectors) -- the second is what Index.grouped takes."""
        from .grouped import group
        # parAssign over the rows as they are (:27), THEN the two stable sorts, by word and by cluster (:28-30)
        order = np.asarray(sorted(range(self.size), key=lambda i: _jkey(self.words[i])), np.int64)
        gv = group(self.data, clustering, word_order=order)
        words = [self.words[i] for i in gv.perm]
        return GroupedWordVectors(words, self.data[gv.perm], gv.centroids, gv.offsets), gv


class GroupedWordVectors(WordVectors):
    """WordVectors.Grouped (WordVectors.scala:96-139)."""

    def __init__(self, words, data, centroids, offsets):
        super().__init__(words, data, KeyIndexGrouped(words, offsets))
        self.centroids, self.offsets = centroids, np.asarray(offsets, np.int32)

    def cluster_of(self, i):                                 # WordVectors.scala:108-111
        return int(np.searchsorted(self.offsets, i, side="right"))


def read_word2vec(source, normalize=False):
    """WordVectors.readWord2Vec (WordVectors.scala:141-252): text format, one `word v0 v1 ...` per line,
    optionally preceded by a `count dimension` header line."""
    from .index import normalize as normalize_vec
    own = isinstance(source, (str, bytes))
    fh = open(source, "r", encoding="utf-8", newline="\n") if own else source
    try:
        first = fh.readline()
        if first == "":
            return WordVectors([], np.zeros((0, 0), np.float32))
        line = first[:-1] if first.endswith("\n") else first
        m = _HEADER.fullmatch(line)
        pending = []
        if m:
            dim = int(m.group(2))
        else:
            dim = len(line.split(" ")) - 1
            pending.append(line)
        words, rows = [], []

        def take(ln):
            if ln == "":                                     # an empty line still counts as read, adds nothing
                return
            parts = ln.split(" ")
            vec = np.array([parse_float(t) for t in parts[1:1 + dim]], np.float32)
            if len(vec) != dim:
                raise ValueError(f"expected {dim} components after {parts[0]!r}, found {len(vec)}")
            words.append(parts[0])
            rows.append(normalize_vec(vec) if normalize else vec)

        for ln in pending:
            take(ln)
        for raw in fh:
            take(raw[:-1] if raw.endswith("\n") else raw)
        data = np.stack(rows) if rows else np.zeros((0, dim), np.float32)
        return WordVectors(words, data)
    finally:
        if own:
            fh.close()


def read_word2vec_text(text, normalize=False):
    return read_word2vec(io.StringIO(text, newline="\n"), normalize)


class KeyedIndex:
    """An index whose results carry words: Index.query returns (word, distance) pairs
    (Index.scala:40-62 with keyIndex); `query_word` looks the vector up first (Index.lookup)."""

    def __init__(self, index, word_vectors):
        self.index, self.vectors = index, word_vectors

    def query(self, k, vector):
        r = self.index.query(k, vector)
        return [(self.vectors.word(int(i)), float(d)) for i, d in zip(r.rows, r.distances)]

    def query_word(self, k, word):
        i = self.vectors.key_index.lookup(word) if self.vectors.key_index is not None else None
        return None if i is None else self.query(k, self.vectors[i])

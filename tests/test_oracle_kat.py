"""Known answers that pin the oracle: JDK java.util.Random values, the
reference's CoderSpec packed-length KAT (CoderSpec.scala:31-40) and the
Vectors.subvectors rule (Vectors.scala:84-104)."""
import numpy as np

from oracle import py_oracle as po


def test_java_random_known_answers(oracle):
    # Widely published JDK values: new Random(0).nextInt(), new Random(42).nextInt()
    assert oracle.JavaRandom(0).next_int() == -1155484576
    assert oracle.JavaRandom(42).next_int() == -1170105035
    r = oracle.JavaRandom(0)
    assert [r.next_int(10_000_000) for _ in range(5)] == [9741360, 5505948, 6548029, 2116447, 8843515]
    # new Random(42): first nextInt(10) values 0, 3, 8, 4, 0 (JDK docs/tutorial staple)
    r = oracle.JavaRandom(42)
    assert [r.next_int(10) for _ in range(5)] == [0, 3, 8, 4, 0]


def test_java_random_c_vs_python(oracle):
    for seed in (0, 1, 42, -7, 2 ** 31 - 1, -2 ** 31):
        a, b = oracle.JavaRandom(seed), po.JRandom(seed)
        for bound in (1, 2, 3, 7, 10, 255, 256, 25000, 10_000_000, 2 ** 30, 2 ** 31 - 1):
            assert a.next_int(bound) == b.next_int(bound)
        for _ in range(64):
            assert a.next_boolean() == b.next_boolean()
        assert a.next_int() == b.next_int()


def test_coder_packed_length_kat(oracle):
    # CoderSpec "produces small code": [1,1,1,1,1] packs into ceil(5*w/8) bytes
    for w in (2, 4, 8, 10, 12, 16):
        code = oracle.coder_build(w, [1, 1, 1, 1, 1])
        assert code.size == (5 * w + 7) // 8
        assert [oracle.coder_get(w, code, 5, i) for i in range(5)] == [1] * 5


def test_coder_width_rule(oracle):
    # ProductQuantizer.coderFactory: 32 - nlz(k-1), rounded up by Coder.factoryFor
    expect = {1: 0, 2: 2, 3: 2, 4: 2, 5: 4, 16: 4, 17: 8, 256: 8, 257: 10, 1024: 10, 1025: 12,
              4096: 12, 4097: 16, 65536: 16, 65537: -1}
    for k, w in expect.items():
        assert oracle.coder_width_for_clusters(k) == w
    for w in range(1, 17):                      # CoderSpec: factoryFor defined for 1..16
        assert oracle.coder_round_width(w) >= w


def test_subvectors_rule(oracle):
    for d, m in [(50, 25), (128, 16), (300, 32), (1024, 64), (7, 3), (10, 10), (5, 1)]:
        fr, un = oracle.subvectors(d, m)
        assert list(zip(fr.tolist(), un.tolist())) == po.subvectors(d, m)
        sizes = un - fr
        assert fr[0] == 0 and un[-1] == d
        assert np.all(fr[1:] == un[:-1])
        assert sizes.max() - sizes.min() <= 1     # VectorsSpec.scala:36-64
    fr, un = oracle.subvectors(300, 32)
    assert (un - fr).tolist() == [10] * 12 + [9] * 20

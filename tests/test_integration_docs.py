"""The three descriptions of the JNI layer -- integration/jni/gulon_jni.c, Native.scala and the table of
INTEGRATION.md -- must name the same natives, and every C entry point the glue calls must be declared in
include/gulon_hip.h.  (No JDK in the image: the glue cannot be compiled here, so at least its text is kept right.)"""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _read(*p):
    return open(os.path.join(ROOT, *p), encoding="utf-8").read()


def test_jni_names_agree_everywhere():
    c_names = set(re.findall(r"^NAT\(\w+, (\w+)\)", _read("integration", "jni", "gulon_jni.c"), re.M))
    scala = set(re.findall(r"@native def (\w+)\(", _read("integration", "scala", "net", "tixxit", "gulon", "hip", "Native.scala")))
    doc = _read("INTEGRATION.md")
    table = doc[doc.index("| Scala method (reference file:line)"):doc.index("Every JNI name above exists")]
    md = set()
    for line in table.splitlines()[2:]:
        cells = [x.strip() for x in line.strip().strip("|").split("|")]
        if len(cells) == 3:
            md.update(re.findall(r"`(\w+)`", cells[2]))
    assert c_names == scala, (sorted(c_names - scala), sorted(scala - c_names))
    assert c_names == md, (sorted(c_names - md), sorted(md - c_names))
    assert len(c_names) >= 23


def test_glue_calls_only_declared_entry_points_and_includes_stdlib():
    glue = _read("integration", "jni", "gulon_jni.c")
    header = _read("include", "gulon_hip.h")
    declared = set(re.findall(r"\b(gulon_\w+)\s*\(", header))
    called = set(re.findall(r"\b(gulon_\w+)\s*\(", glue))
    assert called <= declared, sorted(called - declared)
    assert "#include <stdlib.h>" in glue            # malloc / free
    assert glue.count("malloc(") == glue.count("if (!reps)") == 2            # every allocation is checked


_SCALA_TO_JNI = {"Long": "jlong", "Int": "jint", "Array[Float]": "jfloatArray", "Array[Int]": "jintArray",
                 "Array[Byte]": "jbyteArray", "FloatBuffer": "jobject", "Unit": "void"}


def _scala_natives():
    src = _read("integration", "scala", "net", "tixxit", "gulon", "hip", "Native.scala")
    out = {}
    for name, params, ret in re.findall(r"@native def (\w+)\(([^)]*)\)\s*:\s*(\w+)", src, re.S):
        types = [p.split(":", 1)[1].strip() for p in params.split(",") if p.strip()]
        out[name] = ([_SCALA_TO_JNI[t] for t in types], _SCALA_TO_JNI[ret])
    return out


def _c_natives():
    src = _read("integration", "jni", "gulon_jni.c")
    out = {}
    for ret, name, params in re.findall(r"^NAT\((\w+), (\w+)\)\(([^)]*)\)", src, re.M | re.S):
        out[name] = ([" ".join(p.split()[:-1]) for p in params.split(",")], ret)
    return out


def test_jni_symbols_are_those_of_a_scala_object():
    """`@native def`s of `object Native` live on the module class Native$: the JVM looks up
    Java_..._Native_00024_<method>(JNIEnv *, jobject, ...)."""
    glue = _read("integration", "jni", "gulon_jni.c")
    scala = _read("integration", "scala", "net", "tixxit", "gulon", "hip", "Native.scala")
    assert re.search(r"^object Native\b", scala, re.M)
    assert "#define NAT(ret, name) JNIEXPORT ret JNICALL Java_net_tixxit_gulon_hip_Native_00024_##name" in glue
    for params, _ in _c_natives().values():
        assert params[:2] == ["JNIEnv", "jobject"] or params[:2] == ["JNIEnv *", "jobject"], params[:2]


def test_jni_parameter_types_match_the_scala_signatures():
    sc, cn = _scala_natives(), _c_natives()
    assert sc.keys() == cn.keys()
    for name in sc:
        s_params, s_ret = sc[name]
        c_params, c_ret = cn[name]
        assert c_ret == s_ret, (name, c_ret, s_ret)
        assert c_params[2:] == s_params, (name, c_params[2:], s_params)     # after (JNIEnv *, jobject self)


def test_scala_replacement_bodies_are_present_and_call_only_declared_natives():
    """The bodies a maintainer drops in behind the reference's signatures (KMeans.scala:57-68,134-157;
    ProductQuantizer.scala:25-35,121-153; Index.scala:107-114,310-337,385-440) exist and only call natives that
    Native.scala declares, with the declared number of arguments."""
    natives = _scala_natives()
    seen = set()
    for f, needed in (("HipKMeans.scala", ["def computeClusters", "def parAssign", "def assign", "def fromAssignment",
                                           "def init", "def iterate", "config.report"]),
                      ("HipProductQuantizer.scala", ["def apply", "def encode", "config.report", "Vectors.subvectors"]),
                      ("HipIndex.scala", ["def batchQuery", "def sortedBatchQuery", "def sorted", "def prepareQuery",
                                          "def exactNearestNeighbours", "require(from <= until"]),
                      ("DeviceMatrix.scala", ["def of"])):
        src = _read("integration", "scala", "net", "tixxit", "gulon", "hip", f)
        for n in needed:
            assert n in src, (f, n)
        for m in re.finditer(r"Native\.(\w+)\(", src):
            name = m.group(1)
            if name == "flatten":
                continue
            assert name in natives, (f, name)
            seen.add(name)
            depth, i, args, cur = 1, m.end(), 0, False
            while depth:                                    # count top-level commas of the call
                ch = src[i]
                if ch in "([{":
                    depth += 1
                elif ch in ")]}":
                    depth -= 1
                elif ch == "," and depth == 1:
                    args += 1
                if not ch.isspace() and depth:
                    cur = True
                i += 1
            assert args + (1 if cur else 0) == len(natives[name][0]), (f, name, args + 1, len(natives[name][0]))
    assert {"kmeansTrain", "kmeansAssign", "pqTrain", "pqEncode", "indexCreate", "indexBatchQuery"} <= seen

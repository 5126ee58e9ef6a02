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

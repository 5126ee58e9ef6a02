"""The LDS gather cost model behind conflict_order.hip, re-fitted from the committed measurements
(profiles/r02/lds_model/: scripts/micro/lds_pattern.hip run on an MI355X).  No GPU needed: this checks that the model
DESIGN.md states -- lane groups, bank columns, cycles = sum over the groups of the fullest column -- is what the data
says, and that the kernel's lane-group tables are those groups."""
import importlib.util
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DATA = os.path.join(ROOT, "profiles", "r02", "lds_model")


def _fit_module():
    spec = importlib.util.spec_from_file_location("lds_pattern_fit", os.path.join(ROOT, "scripts", "micro", "lds_pattern_fit.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _model_error(mod, patterns, groups, columns):
    rows = [line.split() for line in open(patterns)]
    t = np.array([float(r[1]) for r in rows])
    pats = [[int(x) for x in r[2:]] for r in rows]

    def fullest(idx):
        per = {}
        for a in set(idx):
            per[a % columns] = per.get(a % columns, 0) + 1
        return max(per.values())
    pred = np.array([sum(fullest([p[lane] for lane in g]) for g in groups) for p in pats], float)
    a, b = np.polyfit(pred, t, 1)
    return a, b, float((t - a * pred - b).std())


def test_b128_groups_and_model():
    mod = _fit_module()
    groups = mod.groups_from_swaps(os.path.join(DATA, "lds_swaps.txt"))
    assert groups == [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27],
                      [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31],
                      [32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59],
                      [36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63]]
    a, b, rms = _model_error(mod, os.path.join(DATA, "lds_patterns.txt"), groups, 16)
    assert 0.95 < a < 1.05 and abs(b) < 1.0 and rms < 0.4, (a, b, rms)
    # the kernel deals rows to exactly these groups
    src = open(os.path.join(ROOT, "gulon_amd", "csrc", "conflict_order.hip")).read()
    assert "0x96" in src and "0x7421" in src and "0x6530" in src
    for lane in range(64):
        g = 2 * (lane >> 5) + ((0x96 >> ((lane & 31) >> 2)) & 1)
        assert lane in groups[g]
    for g in range(4):
        quads = 0x7421 if g & 1 else 0x6530
        lanes = [32 * (g >> 1) + 4 * ((quads >> (4 * (r >> 2))) & 15) + (r & 3) for r in range(16)]
        assert sorted(lanes) == groups[g]


def test_b64_and_b32_models():
    mod = _fit_module()
    for width in (8, 4):
        groups = mod.groups_from_swaps(os.path.join(DATA, f"lds_swaps_b{width}.txt"))
        assert groups == [list(range(32)), list(range(32, 64))]
        a, b, rms = _model_error(mod, os.path.join(DATA, f"lds_patterns_b{width}.txt"), groups, 32)
        assert 0.95 < a < 1.05 and abs(b) < 1.0 and rms < 0.4, (width, a, b, rms)


def test_design_quotes_the_measured_groups():
    txt = open(os.path.join(ROOT, "DESIGN.md")).read()
    assert re.search(r"\{0–3, 12–15, 20–23, 24–27\}, \{4–7, 8–11, 16–19, 28–31\}", txt)

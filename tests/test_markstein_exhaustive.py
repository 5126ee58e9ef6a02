"""The corrected product the k-means update uses for (x - c) / n (gulon_amd/csrc/kmeans.hip, mean_quotient_fast;
KMeans.scala:218) is THE correctly rounded quotient for the divisors 2^j - 1, for every numerator significand: the
sweep the GPU self-test (gulon_selftest_mean_division) repeats with the device's own reciprocal."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_corrected_quotient_is_exact_for_every_numerator_of_the_all_ones_divisors(tmp_path):
    exe = str(tmp_path / "markstein")
    src = os.path.join(ROOT, "tests", "native", "markstein_exhaustive.c")
    # -ffp-contract=off: the three operations are the ones written, nothing fused behind the test's back
    subprocess.run(["gcc", "-O2", "-march=native", "-ffp-contract=off", "-o", exe, src, "-lm"], check=True)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    rows = [tuple(map(int, ln.split())) for ln in out.stdout.split("\n") if ln.strip()]
    assert [j for j, _ in rows] == list(range(1, 25))
    assert all(bad == 0 for _, bad in rows), rows
    assert out.returncode == 0

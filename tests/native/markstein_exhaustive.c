// Test infrastructure (tests/test_markstein_exhaustive.py).  The k-means update computes (x - c) / n as the corrected
// product  q0 = RN(a y), r = fma(-n, q0, a), q = fma(r, y, q0)  with y = RN(1 / n)  (gulon_amd/csrc/kmeans.hip,
// mean_quotient_fast; KMeans.scala:218 is a float division).  The proofs single out divisors whose significand is all
// ones; for the integer counts nearest to that shape, n = 2^j - 1 (j = 24 IS that shape), this sweeps EVERY numerator
// significand and sign of one binade -- the arithmetic is scale invariant away from under- and overflow, which the
// kernels keep out of -- and prints, per divisor, how many quotients differ from a / n.
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

int main(void) {
  long total = 0;
  for (int j = 1; j <= 24; j++) {
    const float nf = (float)((1u << j) - 1u);
    const float y = 1.0f / nf;
    long bad = 0;
    for (uint32_t m = 0; m < (1u << 24); m++) {
      const uint32_t bits = (m >> 23) << 31 | 127u << 23 | (m & 0x7FFFFFu);
      float a;
      memcpy(&a, &bits, 4);
      const float q0 = a * y;
      const float r = fmaf(-nf, q0, a);
      const float q = fmaf(r, y, q0);
      const float ref = a / nf;
      if (memcmp(&q, &ref, 4) != 0) bad++;
    }
    printf("%d %ld\n", j, bad);
    total += bad;
  }
  return total != 0;
}

/*
 * lpf.c -- host-side low-pass prototype designer behind include/lpf.h.
 *
 * Contract: reference src/lpf.c:12-99 (create_low_pass_filter): Hamming-windowed sinc, tap count from the
 * transition width (forced odd), unit DC gain.  One-time O(T) work per client; its float32 output is the
 * `taps` argument of create_frequency_xlating_filter(), so it has to agree with the reference bit for bit --
 * the double/float mix of every expression below is therefore deliberate (compile with -ffp-contract=off).
 */
#include "../../include/lpf.h"

#include <errno.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

static const double XL_PI = 3.14159265358979323846;

/* reference lpf.c:31-38 */
static int xl_lpf_tap_count(uint32_t fs, uint32_t tw) {
  /* 53/22 * fs/tw: numerator in double, denominator as a float product, quotient in double */
  float den = 22.0F * tw;
  double q = (53.0 * fs) / den;
  int count = (int)q;
  return count | 1; /* even -> next odd, odd unchanged */
}

int create_low_pass_filter(float gain, uint32_t sampling_freq, uint32_t cutoff_freq, uint32_t transition_width,
                           float **taps, size_t *len) {
  /* reference lpf.c:12-29: argument checks and their journald-style messages */
  if (sampling_freq == 0) {
    fprintf(stderr, "<3>sampling frequency should be positive\n");
    return -1;
  }
  if (cutoff_freq == 0 || cutoff_freq > (float)sampling_freq / 2) {
    fprintf(stderr, "<3>cutoff frequency should be positive and less than sampling freq / 2. got: %u\n", cutoff_freq);
    return -1;
  }
  if (transition_width == 0) {
    fprintf(stderr, "<3>transition width should be positive\n");
    return -1;
  }

  const int count = xl_lpf_tap_count(sampling_freq, transition_width);
  float *h = (float *)malloc(sizeof(float) * (size_t)count);
  if (h == NULL) return -ENOMEM;

  const int centre = (count - 1) / 2;
  const float omega = 2 * XL_PI * cutoff_freq / sampling_freq; /* lpf.c:72, narrowed to float */

  /* walk outwards from the centre; the sinc is evaluated per tap exactly as lpf.c:74-81 does (no symmetry
   * shortcut: sin(-x)/(-x) and sin(x)/x round identically but the window index differs) */
  for (int k = 0; k < count; k++) {
    const int n = k - centre;
    const float window = (float)(0.54 - 0.46 * cos((2 * XL_PI * k) / (count - 1))); /* lpf.c:45-48 */
    if (n != 0) {
      h[k] = (float)(sin((double)n * omega) / (n * XL_PI) * window);
    } else {
      h[k] = omega / XL_PI * window;
    }
  }

  /* lpf.c:85-94: DC gain summed in float over centre + 2 x upper half, then scale */
  float dc = h[centre];
  for (int k = centre + 1; k < count; k++) dc += 2 * h[k];
  const float scale = gain / dc;
  for (int k = 0; k < count; k++) h[k] *= scale;

  *taps = h;
  *len = (size_t)count;
  return 0;
}

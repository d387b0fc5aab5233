/*
 * xl_taps.c -- turn the real low-pass prototype into what the kernels consume.  Host, once per client.
 *
 * Reference: src/xlating.c:524-534 (band-pass shift + reversal) and :543-549 (NCO increment, Q15 copies).
 * libm's cexpf must be the one the reference would call on this host so that the taps are bit-identical
 * (SURVEY.md section 8(a) a2); compile with -ffp-contract=off.
 */
#include "xl_taps.h"

#include <complex.h>
#include <math.h>

void xl_prepare_taps(const float *taps, size_t T, int32_t center_freq, uint32_t sampling_freq, uint32_t decimation,
                     float *rt, int16_t *rt_q15, float *incr, int16_t *incr_q15) {
  /* :524 -- 2*pi*fc/fs in double, stored as float */
  const float w0 = 2 * 3.14159265358979323846 * center_freq / sampling_freq;

  /* :525-534.  Tap i of the prototype moves to slot T-1-i, except that the reference's reversal loop
   * (bound i <= T/2) swaps the two central taps of an EVEN-length filter twice, leaving that pair in
   * original order.  Write each shifted tap straight to its final slot. */
  for (size_t i = 0; i < T; i++) {
    size_t slot = T - 1 - i;
    if ((T & 1) == 0 && (i == T / 2 - 1 || i == T / 2)) slot = i;
    const float complex rot = cexpf(0.0f + ((float)i * w0) * I); /* angle is a float product */
    const float complex v = taps[i] * rot;
    rt[2 * slot] = crealf(v);
    rt[2 * slot + 1] = cimagf(v);
    rt_q15[2 * slot] = (int16_t)(crealf(v) * (1 << 15));
    rt_q15[2 * slot + 1] = (int16_t)(cimagf(v) * (1 << 15));
  }

  /* :544 -- the step per OUTPUT sample: -w0 * D as a float product */
  const float complex step = cexpf(0.0f + (-w0 * decimation) * I);
  incr[0] = crealf(step);
  incr[1] = cimagf(step);
  incr_q15[0] = (int16_t)(crealf(step) * INT16_MAX);
  incr_q15[1] = (int16_t)(cimagf(step) * INT16_MAX);
}

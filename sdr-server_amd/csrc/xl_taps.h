/* xl_taps.h -- host-side, one-time tap preparation (C, uses libm's cexpf like the reference). */
#ifndef XL_TAPS_H_
#define XL_TAPS_H_
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
/* Reference src/xlating.c:524-534 and :543-549.
 *  rt      [2*T] out: reversed, frequency-shifted taps, interleaved (re,im)
 *  rt_q15  [2*T] out: the same truncated to Q15 (xlating.c:486-487)
 *  incr    [2]   out: phase increment cexpf(-j*w0*D) (xlating.c:544)
 *  incr_q15[2]   out: (int16)(incr * 32767) (xlating.c:548-549) */
void xl_prepare_taps(const float *taps, size_t T, int32_t center_freq, uint32_t sampling_freq, uint32_t decimation,
                     float *rt, int16_t *rt_q15, float *incr, int16_t *incr_q15);
#ifdef __cplusplus
}
#endif
#endif

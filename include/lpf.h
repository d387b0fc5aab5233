/*
 * include/lpf.h -- low-pass tap designer, compatible with the reference's src/lpf.h:6
 * (implementation contract: src/lpf.c:12-99).  Host-side, one-time per client; its output is the
 * `taps` argument of create_frequency_xlating_filter().  The caller owns (free()s) *taps unless it
 * hands them to create_frequency_xlating_filter().
 * (The reference header forgets <stddef.h> -- SURVEY A.5(9); this one includes it.)
 */
#ifndef SDR_SERVER_AMD_LPF_H_
#define SDR_SERVER_AMD_LPF_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 0 ok; -1 invalid arguments (message "<3>..." on stderr, lpf.c:12-29); -ENOMEM. */
int create_low_pass_filter(float gain, uint32_t sampling_freq, uint32_t cutoff_freq, uint32_t transition_width,
                           float **taps, size_t *len);

#ifdef __cplusplus
}
#endif
#endif /* SDR_SERVER_AMD_LPF_H_ */

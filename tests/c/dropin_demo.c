/*
 * tests/c/dropin_demo.c -- a plain C11 caller of the drop-in library, written the way the reference's own callers
 * are (src/dsp_worker.c:90-124 for the set-up, test/test_xlating.c:39-61 for the call pattern): it includes the
 * compatible headers, links libxlating_hip.so, and never sees HIP.  tests/test_c_dropin.py builds and runs it on the
 * GPU box and compares the printed samples with the oracle.
 *
 *   dropin_demo <variant: native|optimized> <sampling_freq> <rate> <transition_width> <center_freq> <nbytes> <ncalls>
 * Input: cu8 ramp in[i] = (uint8_t)(offset + i) like test/utils.c:137-145, offset advancing per call.
 * Output: one line per call: "<output_len> <hex of every complex sample>".
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "lpf.h"
#include "xlating.h"

int main(int argc, char **argv) {
  if (argc < 8) return 2;
  const int optimized = strcmp(argv[1], "optimized") == 0;
  const uint32_t fs = (uint32_t)atol(argv[2]), rate = (uint32_t)atol(argv[3]), tw = (uint32_t)atol(argv[4]);
  const int32_t fc = (int32_t)atol(argv[5]);
  const size_t nbytes = (size_t)atol(argv[6]);
  const int ncalls = atoi(argv[7]);

  float *taps = NULL;
  size_t len = 0;
  if (create_low_pass_filter(1.0f, fs, rate / 2, tw, &taps, &len) != 0) return 1;
  xlating *filter = NULL;
  if (create_frequency_xlating_filter(fs / rate, taps, len, fc, fs, (uint32_t)nbytes, &filter) != 0) return 1;
  fprintf(stderr, "SIMD optimization: %s, %zu taps\n", SIMD_STATUS, len);

  uint8_t *input = malloc(nbytes);
  for (int call = 0; call < ncalls; call++) {
    for (size_t i = 0; i < nbytes; i++) input[i] = (uint8_t)(call * nbytes + i);
    float complex *out = NULL;
    size_t out_len = 0;
    if (optimized) process_optimized_cu8_cf32(input, nbytes, &out, &out_len, filter);
    else process_native_cu8_cf32(input, nbytes, &out, &out_len, filter);
    printf("%zu", out_len);
    for (size_t k = 0; k < out_len; k++) {
      float re = crealf(out[k]), im = cimagf(out[k]);
      uint32_t a, b;
      memcpy(&a, &re, 4);
      memcpy(&b, &im, 4);
      printf(" %08x%08x", a, b);
    }
    printf("\n");
  }
  free(input);
  destroy_xlating(filter); /* frees taps too (xlating.c:600-602) */
  return 0;
}

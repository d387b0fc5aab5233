/*
 * tests/c/dropin_latency.c -- latency distribution of the drop-in process_* call as dsp_worker.c:49-86 makes it: one
 * thread, one filter, one synchronous call per 262144-byte block, the result read right after.  Plain C11, no Python in
 * the process (the interpreter's allocator and collector are not part of what a C server pays).
 *
 *   dropin_latency <variant: native|optimized> <ncalls> [pace_us]
 * Server-default shape (config.conf:13,38,63): 2.016 Msps -> 48 kHz, lpf_cutoff_rate 5 -> 505 taps.  pace_us > 0 sleeps
 * between calls (a real-time stream delivers a block every 65 ms; 0 = back to back).
 * Prints one JSON object: median / p99 / p99.9 / max in microseconds, and the calls over 1 ms with their position.
 */
#define _POSIX_C_SOURCE 200809L
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "lpf.h"
#include "xlating.h"

static double now_us(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec * 1e6 + (double)ts.tv_nsec * 1e-3;
}

static int cmp(const void *a, const void *b) {
  const double x = *(const double *)a, y = *(const double *)b;
  return x < y ? -1 : (x > y ? 1 : 0);
}

int main(int argc, char **argv) {
  if (argc < 3) return 2;
  const int optimized = strcmp(argv[1], "optimized") == 0;
  const int ncalls = atoi(argv[2]);
  const long pace_us = argc > 3 ? atol(argv[3]) : 0;
  const uint32_t fs = 2016000, rate = 48000;
  const size_t nbytes = 262144;
  float *taps = NULL;
  size_t len = 0;
  if (create_low_pass_filter(1.0f, fs, rate / 2, rate / 5, &taps, &len) != 0) return 1;
  xlating *filter = NULL;
  if (create_frequency_xlating_filter(fs / rate, taps, len, -12000, fs, (uint32_t)nbytes, &filter) != 0) return 1;
  uint8_t *input = malloc(nbytes);
  uint64_t x = 88172645463325252ull;
  for (size_t i = 0; i < nbytes; i++) {
    x ^= x << 13, x ^= x >> 7, x ^= x << 17;
    input[i] = (uint8_t)(x >> 32);
  }
  double *t = malloc(sizeof(double) * (size_t)ncalls), *sorted = malloc(sizeof(double) * (size_t)ncalls);
  float acc = 0.0f;
  for (int call = -50; call < ncalls; call++) {
    float complex *out = NULL;
    size_t out_len = 0;
    const double t0 = now_us();
    if (optimized) process_optimized_cu8_cf32(input, nbytes, &out, &out_len, filter);
    else process_native_cu8_cf32(input, nbytes, &out, &out_len, filter);
    if (out_len) acc += crealf(out[out_len - 1]);  /* the caller reads the result (dsp_worker.c:74-77) */
    const double t1 = now_us();
    if (call >= 0) t[call] = t1 - t0;
    if (pace_us > 0) {
      struct timespec ts = {pace_us / 1000000, (pace_us % 1000000) * 1000};
      nanosleep(&ts, NULL);
    }
  }
  memcpy(sorted, t, sizeof(double) * (size_t)ncalls);
  qsort(sorted, (size_t)ncalls, sizeof(double), cmp);
  double sum = 0.0;
  int over = 0;
  for (int i = 0; i < ncalls; i++) sum += t[i], over += t[i] > 1000.0;
  printf("{\"variant\": \"%s\", \"calls\": %d, \"pace_us\": %ld, \"mean_us\": %.1f, \"median_us\": %.1f, \"p99_us\": %.1f, \"p999_us\": %.1f, "
         "\"max_us\": %.1f, \"calls_over_1ms\": %d, \"over_1ms\": [",
         argv[1], ncalls, pace_us, sum / ncalls, sorted[ncalls / 2], sorted[(int)(ncalls * 0.99)], sorted[(int)(ncalls * 0.999)],
         sorted[ncalls - 1], over);
  for (int i = 0, k = 0; i < ncalls && k < 40; i++)
    if (t[i] > 1000.0) printf("%s[%d, %.0f]", k++ ? ", " : "", i, t[i]);
  printf("], \"checksum\": %g}\n", (double)acc);
  free(input);
  free(t);
  free(sorted);
  destroy_xlating(filter);
  return 0;
}

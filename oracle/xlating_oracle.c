/*
 * oracle/xlating_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 * See xlating_oracle.h for the parity status ("pinned") and usage rules.
 *
 * A plain-C restatement of the arithmetic of the reference's hot path, written
 * from the behavioural spec in SURVEY.md Appendix A.  Every function cites the
 * reference file:line whose semantics it follows.  Build with
 *   gcc -std=c11 -O2 -fno-fast-math -ffp-contract=off
 * so that float expressions are evaluated exactly as written (one rounding per
 * operation, no FMA contraction).
 */
#define _GNU_SOURCE
#include "xlating_oracle.h"

#include <complex.h>
#include <errno.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ------------------------------------------------------------------ LPF -- */

/* lpf.c:31-38: ntaps = (int)(53.0 * fs / (22.0f * tw)), forced odd.  Note the
 * denominator is a FLOAT product (uint32 -> float), the quotient is double. */
int orc_lpf_ntaps(uint32_t sampling_freq, uint32_t transition_width) {
  double numer = 53.0 * sampling_freq;
  float denom = 22.0F * transition_width;
  int n = (int)(numer / denom);
  return (n % 2 == 0) ? n + 1 : n;
}

/* lpf.c:12-29 (argument checks), :40-51 (Hamming), :53-99 (windowed sinc,
 * DC-gain normalisation accumulated in float). */
int orc_lpf_design(float gain, uint32_t sampling_freq, uint32_t cutoff_freq,
                   uint32_t transition_width, float **taps_out, size_t *len_out) {
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
  const int ntaps = orc_lpf_ntaps(sampling_freq, transition_width);
  float *h = malloc(sizeof(float) * (size_t)ntaps);
  if (h == NULL) return -ENOMEM;

  const int half = (ntaps - 1) / 2;
  const int span = ntaps - 1;
  /* lpf.c:72: 2*pi*cutoff/fs evaluated in double, narrowed to float */
  const float wc = (float)(2 * M_PI * cutoff_freq / sampling_freq);
  for (int k = 0; k < ntaps; k++) {
    /* lpf.c:45-48: Hamming coefficient computed in double, stored as float */
    const float win = (float)(0.54 - 0.46 * cos((2 * M_PI * k) / span));
    const int n = k - half;
    if (n == 0) {
      h[k] = (float)(wc / M_PI * win); /* lpf.c:76 */
    } else {
      h[k] = (float)(sin((double)n * wc) / (n * M_PI) * win); /* lpf.c:79 */
    }
  }
  /* lpf.c:85-88: float accumulation, centre + 2 * upper half */
  float dc = h[half];
  for (int n = 1; n <= half; n++) dc += 2 * h[n + half];
  gain /= dc;
  for (int k = 0; k < ntaps; k++) h[k] *= gain;
  *taps_out = h;
  *len_out = (size_t)ntaps;
  return 0;
}

/* -------------------------------------------------------------- filter -- */

struct orc_xlating {
  uint32_t D;
  size_t T;
  int sum_mode;
  int fma_step;   /* 1: the phase step as gcc -ffast-math -mfma contracts it (see orc_xlating_set_fma_step) */
  int renorm;     /* 1: renormalise the phase once per call (xlating.c:73, scalar / NEON paths); 0: never (AVX path, :338-339) */
  float *rt;      /* reversed band-pass taps, interleaved re,im (T pairs) */
  int16_t *rt_q;  /* the same in Q15, interleaved */
  /* streaming state.  `hist` is SHARED by the cf32 and cs16 families exactly
   * like the reference's history_offset (xlating.c:29,76,133). */
  size_t hist;
  size_t cap; /* samples in each work buffer */
  float *work_f;
  int16_t *work_q;
  float *out_f;
  int16_t *out_q;
  size_t out_cap;
  float ph_re, ph_im, inc_re, inc_im;         /* float NCO (xlating.c:36-37) */
  int16_t qph_re, qph_im, qinc_re, qinc_im;   /* Q15 NCO (xlating.c:39-42) */
};

static int16_t sat16(int32_t v) { /* xlating.c:85-90 */
  if (v > INT16_MAX) return INT16_MAX;
  if (v < INT16_MIN) return INT16_MIN;
  return (int16_t)v;
}

int orc_xlating_create(uint32_t decimation, const float *taps, size_t taps_len,
                       int32_t center_freq, uint32_t sampling_freq,
                       uint32_t max_input_buffer_length, orc_xlating **out) {
  if (taps_len == 0) return -1; /* xlating.c:496-498 */
  orc_xlating *f = calloc(1, sizeof(*f));
  if (f == NULL) return -ENOMEM;
  f->D = decimation;
  f->T = taps_len;
  f->sum_mode = ORC_SUM_SEQ_F32;
  f->renorm = 1;
  f->rt = malloc(sizeof(float) * 2 * taps_len);
  f->rt_q = malloc(sizeof(int16_t) * 2 * taps_len);
  float complex *bp = malloc(sizeof(float complex) * taps_len);
  if (!f->rt || !f->rt_q || !bp) {
    free(bp);
    orc_xlating_destroy(f);
    return -ENOMEM;
  }
  /* xlating.c:524: angular step in double, narrowed */
  const float w0 = (float)(2 * M_PI * center_freq / sampling_freq);
  /* xlating.c:525-528: shift the low-pass prototype up to w0.  The angle is a
   * FLOAT product index*w0; libm cexpf supplies cos/sin. */
  for (size_t i = 0; i < taps_len; i++) {
    float ang = (float)i * w0;
    float complex e = cexpf(0.0f + ang * I);
    bp[i] = taps[i] * e;
  }
  /* xlating.c:530-534: in-place reversal whose loop bound is i <= T/2.  For
   * even T the central pair is exchanged twice, i.e. stays un-reversed
   * (SURVEY D6).  Restated: mirror everything, then undo the central pair when
   * T is even. */
  for (size_t lo = 0, hi = taps_len - 1; lo < hi; lo++, hi--) {
    float complex t = bp[lo];
    bp[lo] = bp[hi];
    bp[hi] = t;
  }
  if (taps_len % 2 == 0) {
    size_t m = taps_len / 2;
    float complex t = bp[m];
    bp[m] = bp[m - 1];
    bp[m - 1] = t;
  }
  for (size_t i = 0; i < taps_len; i++) {
    f->rt[2 * i] = crealf(bp[i]);
    f->rt[2 * i + 1] = cimagf(bp[i]);
    /* xlating.c:486-487: Q15 taps by truncating conversion */
    f->rt_q[2 * i] = (int16_t)(crealf(bp[i]) * (1 << 15));
    f->rt_q[2 * i + 1] = (int16_t)(cimagf(bp[i]) * (1 << 15));
  }
  free(bp);

  /* xlating.c:543-549 */
  f->ph_re = 1.0f;
  f->ph_im = 0.0f;
  {
    float step = -w0 * decimation; /* float * uint32 -> float */
    float complex inc = cexpf(0.0f + step * I);
    f->inc_re = crealf(inc);
    f->inc_im = cimagf(inc);
  }
  f->qph_re = INT16_MAX;
  f->qph_im = 0;
  f->qinc_re = (int16_t)(f->inc_re * INT16_MAX);
  f->qinc_im = (int16_t)(f->inc_im * INT16_MAX);

  /* xlating.c:552-578: history starts as T-1 zeros; capacities */
  f->hist = taps_len - 1;
  f->cap = max_input_buffer_length / 2 + f->hist;
  f->out_cap = max_input_buffer_length / 2 / decimation + 1;
  f->work_f = calloc(f->cap ? f->cap : 1, 2 * sizeof(float));
  f->work_q = calloc(f->cap ? f->cap : 1, 2 * sizeof(int16_t));
  f->out_f = malloc(2 * sizeof(float) * f->out_cap);
  f->out_q = malloc(2 * sizeof(int16_t) * f->out_cap);
  if (!f->work_f || !f->work_q || !f->out_f || !f->out_q) {
    orc_xlating_destroy(f);
    return -ENOMEM;
  }
  *out = f;
  return 0;
}

void orc_xlating_destroy(orc_xlating *f) {
  if (f == NULL) return;
  free(f->rt);
  free(f->rt_q);
  free(f->work_f);
  free(f->work_q);
  free(f->out_f);
  free(f->out_q);
  free(f);
}

void orc_xlating_set_sum_mode(orc_xlating *f, int mode) { f->sum_mode = mode; }
/* xlating.c:338-339: the x86 AVX path never renormalises (process_optimized_* of an x86 build) */
void orc_xlating_set_renorm(orc_xlating *f, int on) { f->renorm = on != 0; }
/* A reference build with FMA enabled (-O3 -ffast-math -mfma) contracts `phase * phase_incr` (xlating.c:338) to
 * re = fma(pr, ir, -(pi * ii)), im = fma(pr, ii, pi * ir): established by matching the phase sequence of the unmodified
 * reference built that way (oracle/_ref/libref_fast.so) bit for bit; the plain step matches the build without FMA
 * (libref_avx.so) the same way (tests/test_oracle.py). */
void orc_xlating_set_fma_step(orc_xlating *f, int on) { f->fma_step = on != 0; }
size_t orc_xlating_history(const orc_xlating *f) { return f->hist; }
size_t orc_xlating_taps_len(const orc_xlating *f) { return f->T; }
const float *orc_xlating_rtaps(const orc_xlating *f) { return f->rt; }
const int16_t *orc_xlating_rtaps_q15(const orc_xlating *f) { return f->rt_q; }
void orc_xlating_phase(const orc_xlating *f, float *re, float *im) { *re = f->ph_re; *im = f->ph_im; }
void orc_xlating_phase_incr(const orc_xlating *f, float *re, float *im) { *re = f->inc_re; *im = f->inc_im; }
void orc_xlating_phase_q15(const orc_xlating *f, int16_t *re, int16_t *im) { *re = f->qph_re; *im = f->qph_im; }

/* What glibc 2.35's hypotf evaluates: sqrt(x*x + y*y) in double, narrowed.
 * The HIP NCO kernel uses this form; tests compare it with libm's hypotf. */
float orc_hypotf_via_double(float x, float y) {
  double dx = x, dy = y;
  return (float)sqrt(dx * dx + dy * dy);
}

/* xlating.c:52-83 (process_native_cf32): `fresh` new samples have already been
 * appended at work_f[hist ...]. */
static void run_cf32(orc_xlating *f, size_t fresh, float **output, size_t *output_len) {
  const size_t T = f->T;
  const size_t avail = f->hist + fresh;
  size_t made = 0;
  size_t pos = 0; /* window start of the next output */
  if (avail > T - 1) {
    const size_t limit = avail - (T - 1);
    for (; pos < limit; pos += f->D, made++) {
      const float *w = f->work_f + 2 * pos;
      float yr, yi;
      if (f->sum_mode == ORC_SUM_F64) {
        double ar = 0.0, ai = 0.0;
        for (size_t i = 0; i < T; i++) {
          double xr = w[2 * i], xi = w[2 * i + 1], hr = f->rt[2 * i], hi = f->rt[2 * i + 1];
          ar += xr * hr - xi * hi;
          ai += xr * hi + xi * hr;
        }
        yr = (float)ar;
        yi = (float)ai;
      } else {
        /* xlating.c:66-69: acc += x*h, C complex product then complex add, all
         * in float32, taps in order 0..T-1.  (The reference's <=3 leading
         * alignment slots multiply by exact zeros and do not change acc.) */
        float ar = 0.0f, ai = 0.0f;
        for (size_t i = 0; i < T; i++) {
          float xr = w[2 * i], xi = w[2 * i + 1], hr = f->rt[2 * i], hi = f->rt[2 * i + 1];
          float pr = xr * hr - xi * hi;
          float pi = xr * hi + xi * hr;
          ar = ar + pr;
          ai = ai + pi;
        }
        yr = ar;
        yi = ai;
      }
      /* xlating.c:70: out = acc * phase */
      f->out_f[2 * made] = yr * f->ph_re - yi * f->ph_im;
      f->out_f[2 * made + 1] = yr * f->ph_im + yi * f->ph_re;
      /* xlating.c:71: phase *= phase_incr (float32 recurrence) */
      float nr, ni;
      if (f->fma_step) {
        nr = fmaf(f->ph_re, f->inc_re, -(f->ph_im * f->inc_im));
        ni = fmaf(f->ph_re, f->inc_im, f->ph_im * f->inc_re);
      } else {
        nr = f->ph_re * f->inc_re - f->ph_im * f->inc_im;
        ni = f->ph_re * f->inc_im + f->ph_im * f->inc_re;
      }
      f->ph_re = nr;
      f->ph_im = ni;
    }
    /* xlating.c:73: one renormalisation per call that could produce output (not on the AVX path, :338-339) */
    if (f->renorm) {
      float mag = hypotf(f->ph_re, f->ph_im);
      f->ph_re = f->ph_re / mag;
      f->ph_im = f->ph_im / mag;
    }
  }
  /* xlating.c:76-79: keep the unconsumed tail as history */
  f->hist = avail - pos;
  if (pos > 0) memmove(f->work_f, f->work_f + 2 * pos, 2 * sizeof(float) * f->hist);
  *output = f->out_f;
  *output_len = made;
}

/* Test-harness helper (bench.py's parity spot check after millions of blocks): what `ncalls` process_*_cf32 calls of
 * `fresh` samples each do to the stream state (xlating.c:52-83) WITHOUT the filtering: output counts, the float32 phase
 * recurrence + per-call renormalisation (:70-73), the history counter (:76).  The sample history itself is not
 * maintained: feed one real block afterwards before comparing outputs. */
void orc_xlating_skip_calls_cf32(orc_xlating *f, size_t fresh, size_t ncalls) {
  const size_t T = f->T;
  for (size_t c = 0; c < ncalls; c++) {
    const size_t avail = f->hist + fresh;
    size_t pos = 0;
    if (avail > T - 1) {
      const size_t limit = avail - (T - 1);
      for (; pos < limit; pos += f->D) {
        float nr, ni;
        if (f->fma_step) {
          nr = fmaf(f->ph_re, f->inc_re, -(f->ph_im * f->inc_im));
          ni = fmaf(f->ph_re, f->inc_im, f->ph_im * f->inc_re);
        } else {
          nr = f->ph_re * f->inc_re - f->ph_im * f->inc_im;
          ni = f->ph_re * f->inc_im + f->ph_im * f->inc_re;
        }
        f->ph_re = nr;
        f->ph_im = ni;
      }
      if (f->renorm) {
        float mag = hypotf(f->ph_re, f->ph_im);
        f->ph_re = f->ph_re / mag;
        f->ph_im = f->ph_im / mag;
      }
    }
    f->hist = avail - pos;
  }
}

/* xlating.c:92-140 (process_native_cs16) */
static void run_q15(orc_xlating *f, size_t fresh, int16_t **output, size_t *output_len) {
  const size_t T = f->T;
  const size_t avail = f->hist + fresh;
  size_t made = 0;
  size_t pos = 0;
  if (avail > T - 1) {
    const size_t limit = avail - (T - 1);
    for (; pos < limit; pos += f->D, made++) {
      const int16_t *w = f->work_q + 2 * pos;
      int64_t sr = 0, si = 0;
      for (size_t i = 0; i < T; i++) {
        int16_t xr = w[2 * i], xi = w[2 * i + 1], hr = f->rt_q[2 * i], hi = f->rt_q[2 * i + 1];
        sr += (int32_t)xr * hr - (int32_t)xi * hi; /* :114 */
        si += (int32_t)xr * hi + (int32_t)xi * hr; /* :115 */
      }
      int16_t ar = sat16((int32_t)(sr >> 15)); /* :118-119 */
      int16_t ai = sat16((int32_t)(si >> 15));
      int64_t tr = ar * f->qph_re - ai * f->qph_im; /* :121-122 */
      int64_t ti = ar * f->qph_im + ai * f->qph_re;
      f->out_q[2 * made] = sat16((int32_t)(tr >> 15));
      f->out_q[2 * made + 1] = sat16((int32_t)(ti >> 15));
      tr = f->qph_re * f->qinc_re - f->qph_im * f->qinc_im; /* :126-129 */
      ti = f->qph_re * f->qinc_im + f->qph_im * f->qinc_re;
      f->qph_re = sat16((int32_t)(tr >> 15));
      f->qph_im = sat16((int32_t)(ti >> 15));
    }
  }
  f->hist = avail - pos; /* :133 -- the shared history counter */
  if (pos > 0) memmove(f->work_q, f->work_q + 2 * pos, 2 * sizeof(int16_t) * f->hist);
  *output = f->out_q;
  *output_len = made;
}

/* converters, SURVEY A.4 */
void orc_process_cu8_cf32(const uint8_t *in, size_t input_len, float **output, size_t *output_len, orc_xlating *f) {
  size_t n = input_len / 2; /* xlating.c:387-392 */
  float *dst = f->work_f + 2 * f->hist;
  for (size_t i = 0; i < 2 * n; i++) dst[i] = ((float)in[i] - 127.5F) / 128.0F;
  run_cf32(f, n, output, output_len);
}

void orc_process_cs8_cf32(const int8_t *in, size_t input_len, float **output, size_t *output_len, orc_xlating *f) {
  size_t n = input_len / 2; /* xlating.c:397-402 */
  float *dst = f->work_f + 2 * f->hist;
  for (size_t i = 0; i < 2 * n; i++) dst[i] = in[i] / 128.0F;
  run_cf32(f, n, output, output_len);
}

void orc_process_cs16_cf32(const int16_t *in, size_t input_len, float **output, size_t *output_len, orc_xlating *f) {
  size_t n = input_len / 2; /* xlating.c:407-412 */
  float *dst = f->work_f + 2 * f->hist;
  for (size_t i = 0; i < 2 * n; i++) dst[i] = in[i] / 32768.0F;
  run_cf32(f, n, output, output_len);
}

void orc_process_cf32_cf32(const float *in, size_t input_len, float **output, size_t *output_len, orc_xlating *f) {
  size_t n = input_len / 2; /* extension: identity convert */
  memcpy(f->work_f + 2 * f->hist, in, 2 * n * sizeof(float));
  run_cf32(f, n, output, output_len);
}

void orc_process_cu8_cs16(const uint8_t *in, size_t input_len, int16_t **output, size_t *output_len, orc_xlating *f) {
  int16_t *dst = f->work_q + 2 * f->hist; /* xlating.c:417-419 */
  for (size_t i = 0; i < input_len; i++) dst[i] = (int16_t)((((int16_t)in[i]) - 128) * 256);
  run_q15(f, input_len / 2, output, output_len);
}

void orc_process_cs8_cs16(const int8_t *in, size_t input_len, int16_t **output, size_t *output_len, orc_xlating *f) {
  int16_t *dst = f->work_q + 2 * f->hist; /* xlating.c:424-426 */
  for (size_t i = 0; i < input_len; i++) dst[i] = (int16_t)(((int16_t)in[i]) * 256);
  run_q15(f, input_len / 2, output, output_len);
}

void orc_process_cs16_cs16(const int16_t *in, size_t input_len, int16_t **output, size_t *output_len, orc_xlating *f) {
  int16_t *dst = f->work_q + 2 * f->hist; /* xlating.c:431-433 */
  for (size_t i = 0; i < input_len; i++) dst[i] = in[i];
  run_q15(f, input_len / 2, output, output_len);
}

/*
 * oracle/population.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (part of liboracle.so).
 *
 * A whole population of reference filters on the host cores: N independent orc_xlating instances (the reference's
 * thread-per-client model, src/dsp_worker.c:41-88) over the same sequence of IQ blocks, spread over pthreads, their
 * outputs collected per client.  This is what lets the GPU tests and bench.py compare EVERY client of the headline
 * shape (1024 clients x one 8-block super-block = 1.07 G client-samples), not a sample of them.
 *
 * Per client: create (xlating.c:495-582) -> optional fast-forward over `skip_calls` earlier calls of `skip_fresh`
 * samples (orc_xlating_skip_calls_cf32: phase recurrence, renormalisation and history counter only) -> `nwarm` real
 * blocks whose outputs are dropped (they load the sample history) -> `nblocks` blocks whose outputs are appended to
 * the client's row out[c * out_cap ...] (interleaved re, im), the count going to out_len[c].
 * Fixture semantics: test/test_xlating.c:24-61 (consecutive calls on one filter), test/utils.c:176-196 (comparison).
 */
#define _GNU_SOURCE
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#include "xlating_oracle.h"

struct pop_shared {
  uint32_t D, fs, max_len;
  const float *taps;
  size_t T;
  const int32_t *fc;
  size_t n;
  int fmt; /* 0 cu8, 1 cs8, 2 cs16, 3 cf32 */
  size_t skip_fresh, skip_calls;
  const void *blocks;
  size_t block_len; /* scalar elements per block (the reference's input_len) */
  unsigned nwarm, nblocks;
  float *out;
  size_t out_cap;
  size_t *out_len;
  int sum_mode;
  volatile size_t next; /* work queue: next client index */
  int failed;
};

static size_t elem_bytes(int fmt) { return fmt == 2 ? 2u : (fmt == 3 ? 4u : 1u); }

static void run_block(struct pop_shared *s, orc_xlating *f, const char *p, float **o, size_t *k) {
  switch (s->fmt) {
    case 0: orc_process_cu8_cf32((const uint8_t *)p, s->block_len, o, k, f); break;
    case 1: orc_process_cs8_cf32((const int8_t *)p, s->block_len, o, k, f); break;
    case 2: orc_process_cs16_cf32((const int16_t *)p, s->block_len, o, k, f); break;
    default: orc_process_cf32_cf32((const float *)p, s->block_len, o, k, f); break;
  }
}

static void *pop_worker(void *arg) {
  struct pop_shared *s = (struct pop_shared *)arg;
  const size_t stride = s->block_len * elem_bytes(s->fmt);
  for (;;) {
    const size_t c = __atomic_fetch_add(&s->next, 1, __ATOMIC_RELAXED);
    if (c >= s->n) break;
    orc_xlating *f = NULL;
    if (orc_xlating_create(s->D, s->taps, s->T, s->fc[c], s->fs, s->max_len, &f) != 0) {
      s->failed = 1;
      continue;
    }
    orc_xlating_set_sum_mode(f, s->sum_mode);
    if (s->skip_calls) orc_xlating_skip_calls_cf32(f, s->skip_fresh, s->skip_calls);
    const char *p = (const char *)s->blocks;
    float *o = NULL;
    size_t k = 0, have = 0;
    for (unsigned b = 0; b < s->nwarm; b++, p += stride) run_block(s, f, p, &o, &k);
    for (unsigned b = 0; b < s->nblocks; b++, p += stride) {
      run_block(s, f, p, &o, &k);
      if (have + k > s->out_cap) {
        s->failed = 1;
        break;
      }
      memcpy(s->out + 2 * (c * s->out_cap + have), o, 2 * k * sizeof(float));
      have += k;
    }
    s->out_len[c] = have;
    orc_xlating_destroy(f);
  }
  return NULL;
}

/* Returns 0, or -1 if a filter could not be created / a row overflowed.  threads <= 0: one per online CPU. */
int orc_population_cf32(uint32_t decimation, const float *taps, size_t taps_len, const int32_t *center_freq, size_t nclients,
                        uint32_t sampling_freq, uint32_t max_input_buffer_length, int input_format, size_t skip_fresh,
                        size_t skip_calls, const void *blocks, size_t block_len, unsigned nwarm, unsigned nblocks, int sum_mode,
                        float *out, size_t out_cap, size_t *out_len, int threads) {
  struct pop_shared s;
  memset(&s, 0, sizeof(s));
  s.D = decimation, s.fs = sampling_freq, s.max_len = max_input_buffer_length;
  s.taps = taps, s.T = taps_len, s.fc = center_freq, s.n = nclients, s.fmt = input_format;
  s.skip_fresh = skip_fresh, s.skip_calls = skip_calls;
  s.blocks = blocks, s.block_len = block_len, s.nwarm = nwarm, s.nblocks = nblocks;
  s.out = out, s.out_cap = out_cap, s.out_len = out_len, s.sum_mode = sum_mode;
  if (threads <= 0) {
    cpu_set_t set;
    threads = sched_getaffinity(0, sizeof(set), &set) == 0 ? CPU_COUNT(&set) : 1;
  }
  if ((size_t)threads > nclients) threads = (int)nclients;
  if (threads < 1) threads = 1;
  pthread_t *tids = calloc((size_t)threads, sizeof(*tids));
  if (!tids) return -1;
  int started = 0;
  for (int i = 0; i < threads; i++)
    if (pthread_create(&tids[i], NULL, pop_worker, &s) == 0) tids[started++] = tids[i];
  if (started == 0) pop_worker(&s);
  for (int i = 0; i < started; i++) pthread_join(tids[i], NULL);
  free(tids);
  return s.failed ? -1 : 0;
}

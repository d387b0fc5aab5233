/*
 * oracle/cpu_bench.c -- TEST/BENCH INFRASTRUCTURE (cpu_baseline leg of bench.py), not product code.
 *
 * Times the CPU path the way the reference deploys it: one thread per client, each with its own filter, all
 * consuming the same IQ block (src/dsp_worker.c:41-88; harness shape from test/perf_xlating.c:14-80, but
 * wall-clock instead of clock()).  The arithmetic comes from a shared object given on the command line:
 *   api=ref  oracle/_ref/libref_*.so   the UNMODIFIED reference (xlating.h API)
 *   api=orc  oracle/liboracle.so       the repo's scalar restatement
 *
 *   cpu_bench <lib.so> <api> <variant native|optimized> <threads> <seconds> <fs> <rate> <tw> <block_bytes>
 * prints one JSON line.
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef int (*lpf_fn)(float, uint32_t, uint32_t, uint32_t, float **, size_t *);
typedef int (*ref_create_fn)(uint32_t, float *, size_t, int32_t, uint32_t, uint32_t, void **);
typedef void (*ref_process_fn)(const uint8_t *, size_t, void **, size_t *, void *);
typedef void (*ref_destroy_fn)(void *);
typedef int (*orc_create_fn)(uint32_t, const float *, size_t, int32_t, uint32_t, uint32_t, void **);

static double now(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

struct job {
  void *filter;
  ref_process_fn process;
  const uint8_t *block;
  size_t block_bytes;
  double deadline;
  long calls;
  size_t outputs;
};

static void *worker(void *arg) {
  struct job *j = (struct job *)arg;
  void *out;
  size_t n = 0;
  j->process(j->block, j->block_bytes, &out, &n, j->filter); /* warm */
  while (now() < j->deadline) {
    j->process(j->block, j->block_bytes, &out, &n, j->filter);
    j->calls++;
    j->outputs += n;
  }
  return NULL;
}

int main(int argc, char **argv) {
  if (argc < 10) {
    fprintf(stderr, "usage: %s lib api variant threads seconds fs rate tw block_bytes\n", argv[0]);
    return 2;
  }
  const char *lib = argv[1], *api = argv[2], *variant = argv[3];
  int threads = atoi(argv[4]);
  double seconds = atof(argv[5]);
  uint32_t fs = (uint32_t)atol(argv[6]), rate = (uint32_t)atol(argv[7]), tw = (uint32_t)atol(argv[8]);
  size_t block_bytes = (size_t)atol(argv[9]);
  void *h = dlopen(lib, RTLD_NOW);
  if (!h) {
    fprintf(stderr, "dlopen %s: %s\n", lib, dlerror());
    return 1;
  }
  int is_ref = strcmp(api, "ref") == 0;
  char name[128];
  lpf_fn lpf = (lpf_fn)dlsym(h, is_ref ? "create_low_pass_filter" : "orc_lpf_design");
  snprintf(name, sizeof(name), is_ref ? "process_%s_cu8_cf32" : "orc_process_cu8_cf32", variant);
  ref_process_fn process = (ref_process_fn)dlsym(h, name);
  ref_destroy_fn destroy = (ref_destroy_fn)dlsym(h, is_ref ? "destroy_xlating" : "orc_xlating_destroy");
  void *create = dlsym(h, is_ref ? "create_frequency_xlating_filter" : "orc_xlating_create");
  if (!lpf || !process || !destroy || !create) {
    fprintf(stderr, "missing symbol in %s\n", lib);
    return 1;
  }
  uint8_t *block = malloc(block_bytes);
  uint64_t x = 0x5DEECE66DULL;
  for (size_t i = 0; i < block_bytes; i++) {
    x ^= x >> 12;
    x ^= x << 25;
    x ^= x >> 27;
    block[i] = (uint8_t)((x * 0x2545F4914F6CDD1DULL) >> 56);
  }
  struct job *jobs = calloc((size_t)threads, sizeof(*jobs));
  pthread_t *tids = calloc((size_t)threads, sizeof(*tids));
  size_t ntaps = 0;
  for (int i = 0; i < threads; i++) {
    float *taps = NULL;
    if (lpf(1.0f, fs, rate / 2, tw, &taps, &ntaps) != 0) return 1;
    int32_t fc = -984000 + 1920 * (i % 1024);
    int code = is_ref ? ((ref_create_fn)create)(fs / rate, taps, ntaps, fc, fs, (uint32_t)block_bytes, &jobs[i].filter)
                      : ((orc_create_fn)create)(fs / rate, taps, ntaps, fc, fs, (uint32_t)block_bytes, &jobs[i].filter);
    if (!is_ref) free(taps); /* the reference owns taps, the restatement copies them */
    if (code != 0) return 1;
    jobs[i].process = process;
    jobs[i].block = block;
    jobs[i].block_bytes = block_bytes;
  }
  double t0 = now();
  for (int i = 0; i < threads; i++) {
    jobs[i].deadline = t0 + seconds;
    pthread_create(&tids[i], NULL, worker, &jobs[i]);
  }
  long calls = 0;
  size_t outs = 0;
  for (int i = 0; i < threads; i++) {
    pthread_join(tids[i], NULL);
    calls += jobs[i].calls;
    outs += jobs[i].outputs;
  }
  double dt = now() - t0;
  const char **simd = (const char **)dlsym(h, "SIMD_STATUS");
  printf("{\"threads\": %d, \"calls\": %ld, \"seconds\": %.4f, \"ntaps\": %zu, \"msps\": %.3f, \"outputs\": %zu, \"simd\": \"%s\"}\n",
         threads, calls, dt, ntaps, (double)calls * (block_bytes / 2) / dt / 1e6, outs, simd ? *simd : "n/a");
  for (int i = 0; i < threads; i++) destroy(jobs[i].filter);
  return 0;
}

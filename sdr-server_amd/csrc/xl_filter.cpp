// xl_filter.cpp -- the xlating.h drop-in: one `xlating` handle == one client filter, all arithmetic on the GPU.
//
// State machine = the reference's (src/xlating.c:14-43, 52-140, 495-616), held in device memory:
//   work_f / work_q   converted sample images [history | new samples] of the two output families
//   hist              ONE history counter shared by both families (xlating.c:29,76,133 -- quirk kept)
//   phase / qphase    float32 and Q15 NCO state (xlating.c:36-42)
// Per process_* call (xlating.c:352-447): H2D of the raw block -> convert kernel (appends at work[hist]) ->
// NCO phase-table kernel -> FIR kernel (the batch kernel with a one-client tile) -> D2H of the K outputs ->
// in-place history memmove kernel -> stream sync -> return the filter-owned host pointer.
#include <errno.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <new>
#include <utility>
#include <vector>

#include "../../include/xlating.h"
#include "xl_common.h"
#include "xl_device.h"
#include "xl_taps.h"

struct xlating_t {
  uint32_t D = 0;
  size_t T = 0;
  uint32_t Tpad = 0;
  float *original_taps = nullptr;  // owned, like xlating.c:508
  int device = -1;
  hipStream_t stream = nullptr;
  size_t hist = 0;         // shared history counter
  size_t cap = 0;          // samples per work image
  size_t max_samples = 0;  // max_input_buffer_length / 2
  size_t out_cap = 0;
  int16_t qinc[2] = {0, 0};
  bool warned = false;
  int x86 = 0;             // process_optimized_*: 1 = never renormalise the phase (the reference's x86 AVX build, xlating.c:338-339),
                           // 2 = and take the FMA-contracted phase step of an -mfma build (xl_grid.h: XL_POS_FMA_STEP)
  uint32_t spec_flags = 0; // XlPos flags the look-ahead table was tabulated with
  uint32_t ota = 64;       // outputs per wave (64 unless the window image would not fit the LDS)

  void *d_raw = nullptr;
  float2 *d_work_f = nullptr;
  short2 *d_work_q = nullptr;
  float2 *d_out_f = nullptr;
  short2 *d_out_q = nullptr;
  float2 *d_phtab = nullptr;      // phase table of the current call (every XL_PH_STRIDE-th phase)
  float2 *d_phtab_next = nullptr; // look-ahead: table of the NEXT call, tabulated on stream_nco while the host is busy
  float2 *d_phase_next = nullptr; // look-ahead: the phase after the next call
  hipStream_t stream_nco = nullptr;
  hipEvent_t ev_nco = nullptr;    // the look-ahead tabulation has finished
  bool spec_valid = false;        // d_phtab_next / d_phase_next hold a call of spec_K outputs
  bool lookahead = true;          // XL_EXP_NOLOOKAHEAD disables (tuning)
  size_t spec_K = 0;
  short2 *d_qphtab = nullptr;       // Q15 phase table of the current call (every XL_PH_STRIDE-th phase)
  short2 *d_qphtab_next = nullptr;  // look-ahead (as for the float family): the NEXT Q15 call's table and the phase after it
  short2 *d_qphase_next = nullptr;
  hipEvent_t ev_qnco = nullptr;
  bool qspec_valid = false;
  size_t qspec_K = 0;
  float2 *d_phase = nullptr;
  short2 *d_qphase = nullptr;
  float2 *d_taps = nullptr;
  short2 *d_qtaps = nullptr;
  XlGroup *d_group = nullptr;
  XlNcoClient *d_nco = nullptr;

  bool zero_copy = true;  // kernels read the pinned input / write the pinned output over PCIe (XL_EXP_DROPIN_COPY=1: staged copies)
  void *h_in = nullptr;
  float2 *h_out_f = nullptr;
  short2 *h_out_q = nullptr;
};

static void xl_filter_free(xlating *f) {
  if (f == nullptr) return;
  if (f->device >= 0) (void)hipSetDevice(f->device);
  if (f->stream) (void)hipStreamSynchronize(f->stream);
  if (f->stream_nco) (void)hipStreamSynchronize(f->stream_nco);
  void *dev[] = {f->d_phtab_next, f->d_phase_next, f->d_qphtab_next, f->d_qphase_next,
                 f->d_raw,   f->d_work_f, f->d_work_q, f->d_out_f, f->d_out_q, f->d_phtab, f->d_qphtab,
                 f->d_phase, f->d_qphase, f->d_taps,   f->d_qtaps, f->d_group, f->d_nco};
  for (void *p : dev)
    if (p) (void)hipFree(p);
  void *host[] = {f->h_in, f->h_out_f, f->h_out_q};
  for (void *p : host)
    if (p) (void)hipHostFree(p);
  if (f->ev_nco) (void)hipEventDestroy(f->ev_nco);
  if (f->ev_qnco) (void)hipEventDestroy(f->ev_qnco);
  if (f->stream_nco) (void)hipStreamDestroy(f->stream_nco);
  if (f->stream) (void)hipStreamDestroy(f->stream);
  if (f->original_taps) free(f->original_taps);  // xlating.c:600-602
  delete f;
}

extern "C" int create_frequency_xlating_filter(uint32_t decimation, float *taps, size_t taps_len, int32_t center_freq,
                                               uint32_t sampling_freq, uint32_t max_input_buffer_length,
                                               xlating **filter) {
  if (taps_len == 0) return -1;  // xlating.c:496-498 (taps NOT consumed)
  if (filter == nullptr || taps == nullptr) return -EINVAL;
  // From here on the taps are CONSUMED on every path, failures included: the caller (dsp_worker_start,
  // dsp_worker.c:98-107) treats them as handed over once taps_len != 0, like the reference's own error paths
  // (xlating.c:521,556).
  if (decimation == 0) {
    free(taps);
    return -EINVAL;
  }
  const int dev = xl_hip_select_device(-1);
  if (dev < 0) {
    XL_LOG_ERR("no usable HIP device (%s); this build has no CPU arithmetic path", xlating_hip_device_info());
    free(taps);
    return -ENODEV;
  }
  xlating *f = new (std::nothrow) xlating_t();
  if (f == nullptr) {
    free(taps);
    return -ENOMEM;
  }
  f->original_taps = taps;
  f->D = decimation;
  f->T = taps_len;
  f->Tpad = xl_roundup((uint32_t)taps_len, XL_TAP_UNROLL);
  f->device = dev;
  f->hist = taps_len - 1;  // xlating.c:552
  f->max_samples = max_input_buffer_length / 2;
  f->cap = f->max_samples + f->hist;                             // xlating.c:553
  f->out_cap = max_input_buffer_length / 2 / decimation + 1;     // xlating.c:568

  std::vector<float> rt(2 * (size_t)f->Tpad, 0.0f);  // zero-padded to the unroll width: pad taps are exact zeros
  std::vector<int16_t> rtq(2 * taps_len);
  float incr[2];
  xl_prepare_taps(taps, taps_len, center_freq, sampling_freq, decimation, rt.data(), rtq.data(), incr, f->qinc);

  XlGroup g;
  memset(&g, 0, sizeof(g));
  g.D = decimation;
  g.T = (uint32_t)taps_len;
  g.Tpad = f->Tpad;
  g.rem0 = 0;
  g.hv0 = 0;
  g.ntiles = 1;
  g.wide = (decimation % 2 == 0) ? 1u : 0u;
  g.tiles[0].tap_off = 0;
  g.tiles[0].nclients = 1;
  g.tiles[0].incr[0] = make_float2(incr[0], incr[1]);
  XlNcoClient nc;
  memset(&nc, 0, sizeof(nc));
  nc.incr = make_float2(incr[0], incr[1]);
  nc.D = decimation;
  const float2 one = make_float2(1.0f, 0.0f);          // xlating.c:543
  const short2 qone = make_short2(INT16_MAX, 0);        // xlating.c:546-547
  // work images hold a few samples past `cap`: the FIR kernel's window image is padded to the unroll width
  const size_t work_n = f->cap + XL_TAP_UNROLL;

  XL_TRY(hipSetDevice(dev));
  XL_TRY(hipStreamCreateWithFlags(&f->stream, hipStreamNonBlocking));
  XL_TRY(hipStreamCreateWithFlags(&f->stream_nco, hipStreamNonBlocking));
  XL_TRY(hipEventCreateWithFlags(&f->ev_nco, hipEventDisableTiming));
  XL_TRY(hipEventCreateWithFlags(&f->ev_qnco, hipEventDisableTiming));
  f->lookahead = xl_exp_getenv("XL_EXP_NOLOOKAHEAD") == nullptr;
  f->x86 = getenv("XLATING_OPTIMIZED_X86") != nullptr ? atoi(getenv("XLATING_OPTIMIZED_X86")) : 0;
  if (f->x86 < 0 || f->x86 > 2) f->x86 = 0;
  f->zero_copy = xl_exp_getenv("XL_EXP_DROPIN_COPY") == nullptr;
  XL_TRY(hipMalloc(&f->d_raw, f->max_samples * 8 + 16));
  XL_TRY(hipMalloc((void **)&f->d_work_f, work_n * sizeof(float2)));
  XL_TRY(hipMalloc((void **)&f->d_work_q, work_n * sizeof(short2)));
  XL_TRY(hipMalloc((void **)&f->d_out_f, f->out_cap * sizeof(float2)));
  XL_TRY(hipMalloc((void **)&f->d_out_q, f->out_cap * sizeof(short2)));
  XL_TRY(hipMalloc((void **)&f->d_phtab, (f->out_cap / XL_PH_STRIDE + 8) * sizeof(float2)));  // every XL_PH_STRIDE-th phase
  XL_TRY(hipMalloc((void **)&f->d_qphtab, (f->out_cap / XL_PH_STRIDE + 8) * sizeof(short2)));
  XL_TRY(hipMalloc((void **)&f->d_qphtab_next, (f->out_cap / XL_PH_STRIDE + 8) * sizeof(short2)));
  XL_TRY(hipMalloc((void **)&f->d_qphase_next, sizeof(short2)));
  XL_TRY(hipMalloc((void **)&f->d_phase, sizeof(float2)));
  XL_TRY(hipMalloc((void **)&f->d_phtab_next, (f->out_cap / XL_PH_STRIDE + 8) * sizeof(float2)));
  XL_TRY(hipMalloc((void **)&f->d_phase_next, sizeof(float2)));
  XL_TRY(hipMalloc((void **)&f->d_qphase, sizeof(short2)));
  XL_TRY(hipMalloc((void **)&f->d_taps, rt.size() * sizeof(float)));
  XL_TRY(hipMalloc((void **)&f->d_qtaps, rtq.size() * sizeof(int16_t)));
  XL_TRY(hipMalloc((void **)&f->d_group, sizeof(XlGroup)));
  XL_TRY(hipMalloc((void **)&f->d_nco, sizeof(XlNcoClient)));
  XL_TRY(hipHostMalloc(&f->h_in, f->max_samples * 8 + 16, hipHostMallocDefault));
  XL_TRY(hipHostMalloc((void **)&f->h_out_f, f->out_cap * sizeof(float2), hipHostMallocDefault));
  XL_TRY(hipHostMalloc((void **)&f->h_out_q, f->out_cap * sizeof(short2), hipHostMallocDefault));
  // xlating.c:559,565: both work images start as zeros (the stream is preceded by T-1 zero samples)
  XL_TRY(hipMemsetAsync(f->d_work_f, 0, work_n * sizeof(float2), f->stream));
  XL_TRY(hipMemsetAsync(f->d_work_q, 0, work_n * sizeof(short2), f->stream));
  XL_TRY(hipMemcpyAsync(f->d_taps, rt.data(), rt.size() * sizeof(float), hipMemcpyHostToDevice, f->stream));
  XL_TRY(hipMemcpyAsync(f->d_qtaps, rtq.data(), rtq.size() * sizeof(int16_t), hipMemcpyHostToDevice, f->stream));
  XL_TRY(hipMemcpyAsync(f->d_group, &g, sizeof(g), hipMemcpyHostToDevice, f->stream));
  XL_TRY(hipMemcpyAsync(f->d_nco, &nc, sizeof(nc), hipMemcpyHostToDevice, f->stream));
  XL_TRY(hipMemcpyAsync(f->d_phase, &one, sizeof(one), hipMemcpyHostToDevice, f->stream));
  XL_TRY(hipMemcpyAsync(f->d_qphase, &qone, sizeof(qone), hipMemcpyHostToDevice, f->stream));
  XL_TRY(hipStreamSynchronize(f->stream));
  f->ota = xl_fir_pick_ota(decimation, f->Tpad, 160 * 1024);
  if (f->ota == 0) {
    XL_LOG_ERR("decimation %u with %zu taps needs a %zu-byte window image even for 8 outputs per wave (> 160 KiB LDS)",
               decimation, taps_len, xl_fir_lds_bytes_ota(decimation, f->Tpad, 8));
    xl_filter_free(f);
    return -EINVAL;
  }
  *filter = f;
  return 0;
fail:
  xl_filter_free(f);  // frees taps too, like the reference's -ENOMEM paths (xlating.c:521,556,...)
  return xl_errno_of_last_hip_error();  // -ENOMEM (allocation), -ENODEV (no gfx950 code object / device), -EIO
}

extern "C" void destroy_xlating(xlating *filter) { xl_filter_free(filter); }

// include/xlating.h extension: which build of the reference process_optimized_* follows (see the header)
extern "C" int xlating_set_optimized_x86(xlating *filter, int on) {
  if (filter == nullptr || on < 0 || on > 2) return -EINVAL;
  filter->x86 = on;
  return 0;
}

// Output count and consumed samples for `fresh` new samples (xlating.c:53-60,76): W = hist + fresh;
// outputs at window starts 0, D, 2D, ... < W - (T-1).
static inline void xl_counts(const xlating *f, size_t fresh, size_t *W, size_t *K, size_t *pos) {
  *W = f->hist + fresh;
  *K = 0;
  if (*W > f->T - 1) {
    const size_t limit = *W - (f->T - 1);
    *K = (limit + f->D - 1) / f->D;
  }
  *pos = *K * (size_t)f->D;
}

static bool xl_check_len(xlating *f, size_t nsamples) {
  if (nsamples <= f->max_samples) return true;
  if (!f->warned) {
    XL_LOG_ERR("input of %zu samples exceeds max_input_buffer_length/2 = %zu; block dropped", nsamples, f->max_samples);
    f->warned = true;
  }
  return false;
}

// Latency trace of the drop-in call (XL_DROPIN_TRACE_US=<threshold>): a call that takes longer than the threshold prints
// where its time went -- copy into the pinned block | enqueue of the launches | wait for the stream | the look-ahead
// request -- as one "<6>" line.  Off by default; the stamps cost ~20 ns each.
static long xl_trace_threshold_us() {
  static const long v = getenv("XL_DROPIN_TRACE_US") ? atol(getenv("XL_DROPIN_TRACE_US")) : -1;
  return v;
}
static inline double xl_now_us() {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec * 1e6 + (double)ts.tv_nsec * 1e-3;
}

static void xl_run_cf32(xlating *f, const void *input, size_t input_len, int fmt, int mode, XL_CF32 **output,
                        size_t *output_len) {
  const long trace_us = xl_trace_threshold_us();
  double ts0 = 0.0, ts1 = 0.0, ts2 = 0.0, ts3 = 0.0;
  if (trace_us >= 0) ts0 = xl_now_us();
  *output = reinterpret_cast<XL_CF32 *>(f->h_out_f);
  *output_len = 0;
  const size_t n = input_len / 2;
  if (!xl_check_len(f, n)) return;
  const size_t bytes = n * xl_bytes_per_sample(fmt);
  size_t W, K, pos;
  bool ahead = false;  // the next call's table is requested during this call (else after the sync, if at all)
  const uint32_t pflags = (mode == 1 && f->x86) ? (XL_POS_NORENORM | (f->x86 == 2 ? XL_POS_FMA_STEP : 0u)) : 0u;
  xl_counts(f, n, &W, &K, &pos);
  XL_TRY(hipSetDevice(f->device));
  if (n > 0) {
    memcpy(f->h_in, input, bytes);
    if (trace_us >= 0) ts1 = xl_now_us();
    // the convert kernel reads the pinned block over PCIe itself: one stream operation less than copy + convert
    if (!f->zero_copy) XL_TRY(hipMemcpyAsync(f->d_raw, f->h_in, bytes, hipMemcpyHostToDevice, f->stream));
    XL_TRY(xl_launch_convert_cf32(f->zero_copy ? f->h_in : f->d_raw, fmt, (uint32_t)n, f->d_work_f + f->hist, f->stream));
  }
  if (K > 0) {
    XlPos pos;
    memset(&pos, 0, sizeof(pos));
    pos.S = (uint32_t)n;
    pos.G = 1;
    pos.pad = pflags;
    // The phases of this call: tabulated ahead on stream_nco after the previous call if that call guessed this one's
    // output count (the recurrence is data independent: xlating.c:70-73), else now.  The chain is ~30 us of pure
    // latency; ahead of time it overlaps the host's work between calls, the upload and the convert kernel.
    ahead = false;
    if (f->spec_valid && f->spec_K == K && f->spec_flags == pflags) {
      XL_TRY(hipStreamWaitEvent(f->stream, f->ev_nco, 0));
      std::swap(f->d_phtab, f->d_phtab_next);
      std::swap(f->d_phase, f->d_phase_next);
      // d_phase now holds the committed phase AFTER this call (tabulated ahead together with this call's table), and
      // the swapped-out buffers were last read by the previous call, which has been waited for: the NEXT call's table
      // can be requested during this call (below, behind this call's own launches) instead of after its final sync.
      ahead = f->lookahead;
    } else {
      if (f->spec_valid) XL_TRY(hipStreamWaitEvent(f->stream, f->ev_nco, 0));  // (its buffers are reused below)
      XL_TRY(xl_launch_nco_table(f->d_nco, 1, f->d_phase, f->d_phase, f->d_phtab, pos, (uint32_t)K, 0, f->stream));
    }
    f->spec_valid = false;
    XlFirArgs a;
    memset(&a, 0, sizeof(a));
    a.in0 = f->d_work_f;
    a.n0 = (uint32_t)W;
    a.in1 = nullptr;
    a.n1 = 0;
    a.fmt = XLF_CF32;
    a.pos = pos;
    a.explicit_dyn = 1;  // the work image carries its own history: window of output 0 starts at sample 0 (xlating.c:61)
    a.dyn1.base = 0;
    a.dyn1.K = (uint32_t)K;
    a.dyn1.zero_below = 0;
    a.dyn1.j0 = 0;
    a.groups = f->d_group;
    a.ngroups = 1;
    a.ota = f->ota;
    a.xtiles = (uint32_t)((K + f->ota - 1) / f->ota);
    a.flags = ((f->D % 2 == 0) ? 1u : 0u) | 4u;
    a.taps = f->d_taps;
    a.phtab = f->d_phtab;
    a.out = f->zero_copy ? f->h_out_f : f->d_out_f;  // (the K outputs go straight to the pinned result buffer)
    XL_TRY(xl_launch_fir(1, mode, XL_NW_DEFAULT, a, xl_fir_lds_bytes_ota(f->D, f->Tpad, f->ota), f->stream));
    if (!f->zero_copy) XL_TRY(hipMemcpyAsync(f->h_out_f, f->d_out_f, K * sizeof(float2), hipMemcpyDeviceToHost, f->stream));
  }
  {
    // xlating.c:76-79.  pos can only exceed W when D > T (the reference underflows there); clamp.
    const size_t keep = pos <= W ? W - pos : 0;
    if (pos > 0) XL_TRY(xl_launch_move_down(f->d_work_f, (uint32_t)pos, (uint32_t)keep, 8, f->stream));
    f->hist = keep;
  }
  if (ahead) {
    // the launch overlaps this call's convert / FIR / move kernels on the device and costs the caller no wait
    // (measured per 262144-byte block, 505 taps: 69 us with the request after the sync, 47 us here)
    size_t Wn, Kn, posn;
    xl_counts(f, n, &Wn, &Kn, &posn);
    if (Kn > 0) {
      XlPos pn;
      memset(&pn, 0, sizeof(pn));
      pn.S = (uint32_t)n;
      pn.G = 1;
      pn.pad = pflags;
      XL_TRY(xl_launch_nco_table(f->d_nco, 1, f->d_phase, f->d_phase_next, f->d_phtab_next, pn, (uint32_t)Kn, 0, f->stream_nco));
      XL_TRY(hipEventRecord(f->ev_nco, f->stream_nco));
      f->spec_valid = true;
      f->spec_K = Kn;
      f->spec_flags = pflags;
    }
  }
  if (trace_us >= 0) ts2 = xl_now_us();
  XL_TRY(hipStreamSynchronize(f->stream));
  if (trace_us >= 0) ts3 = xl_now_us();
  if (K > 0 && f->lookahead && !f->spec_valid) {
    // look-ahead: if the next call brings the same number of samples (and no cs16-family call moves the shared
    // history in between) it produces Knext outputs; tabulate them now, off the caller's critical path.
    // d_phase (the committed post-call phase) is only read; nothing in flight on `stream` after the sync above.
    size_t Wn, Kn, posn;
    xl_counts(f, n, &Wn, &Kn, &posn);
    if (Kn > 0) {
      XlPos pn;
      memset(&pn, 0, sizeof(pn));
      pn.S = (uint32_t)n;
      pn.G = 1;
      pn.pad = pflags;
      XL_TRY(xl_launch_nco_table(f->d_nco, 1, f->d_phase, f->d_phase_next, f->d_phtab_next, pn, (uint32_t)Kn, 0, f->stream_nco));
      XL_TRY(hipEventRecord(f->ev_nco, f->stream_nco));
      f->spec_valid = true;
      f->spec_K = Kn;
      f->spec_flags = pflags;
    }
  }
  if (trace_us >= 0) {
    const double ts4 = xl_now_us();
    if (ts4 - ts0 > (double)trace_us)
      fprintf(stderr, "<6>xlating-hip: slow call %.1f us = copy-in %.1f + enqueue %.1f + stream wait %.1f + look-ahead %.1f (K %zu, ahead %d)\n",
              ts4 - ts0, ts1 - ts0, ts2 - ts1, ts3 - ts2, ts4 - ts3, K, (int)ahead);
  }
  *output_len = K;
  return;
fail:
  *output_len = 0;
}

static void xl_run_q15(xlating *f, const void *input, size_t input_len, int fmt, int16_t **output, size_t *output_len) {
  *output = reinterpret_cast<int16_t *>(f->h_out_q);
  *output_len = 0;
  const size_t n = input_len / 2;
  if (!xl_check_len(f, (input_len + 1) / 2)) return;
  const size_t bytes = input_len * (fmt == XLF_CS16 ? 2 : 1);
  size_t W, K, pos;
  xl_counts(f, n, &W, &K, &pos);
  XL_TRY(hipSetDevice(f->device));
  if (input_len > 0) {
    memcpy(f->h_in, input, bytes);
    if (!f->zero_copy) XL_TRY(hipMemcpyAsync(f->d_raw, f->h_in, bytes, hipMemcpyHostToDevice, f->stream));
    // xlating.c:417-419: every scalar element is converted (also a trailing odd one)
    XL_TRY(xl_launch_convert_q15(f->zero_copy ? f->h_in : f->d_raw, fmt, (uint32_t)input_len, reinterpret_cast<int16_t *>(f->d_work_q + f->hist),
                                 f->stream));
  }
  if (K > 0) {
    // The Q15 phases of this call (truncating int16 recurrence, xlating.c:126-129: data independent): tabulated ahead on
    // stream_nco after the previous Q15 call if that call guessed this one's output count, else now (~30 us of pure latency)
    if (f->qspec_valid && f->qspec_K == K) {
      XL_TRY(hipStreamWaitEvent(f->stream, f->ev_qnco, 0));
      std::swap(f->d_qphtab, f->d_qphtab_next);
      std::swap(f->d_qphase, f->d_qphase_next);  // (d_qphase now holds the phase AFTER this call)
    } else {
      if (f->qspec_valid) XL_TRY(hipStreamWaitEvent(f->stream, f->ev_qnco, 0));  // (its buffers are reused below)
      XL_TRY(xl_launch_nco_table_q15(f->qinc[0], f->qinc[1], f->d_qphase, f->d_qphase, f->d_qphtab, (uint32_t)K, f->stream));
    }
    f->qspec_valid = false;
    XL_TRY(xl_launch_fir_q15(f->d_work_q, f->d_qtaps, (uint32_t)f->T, f->D, (uint32_t)K, f->qinc[0], f->qinc[1], f->d_qphtab,
                             f->zero_copy ? f->h_out_q : f->d_out_q, f->stream));
    if (!f->zero_copy) XL_TRY(hipMemcpyAsync(f->h_out_q, f->d_out_q, K * sizeof(short2), hipMemcpyDeviceToHost, f->stream));
  }
  {
    const size_t keep = pos <= W ? W - pos : 0;  // xlating.c:133-136
    if (pos > 0) XL_TRY(xl_launch_move_down(f->d_work_q, (uint32_t)pos, (uint32_t)keep, 4, f->stream));
    f->hist = keep;
  }
  XL_TRY(hipStreamSynchronize(f->stream));
  if (K > 0 && f->lookahead) {
    // look-ahead: the next Q15 call's table if it brings the same number of samples (the shared history counter decides its
    // output count: a cf32-family call in between changes it, and the guess is then simply redone)
    size_t Wn, Kn, posn;
    xl_counts(f, n, &Wn, &Kn, &posn);
    if (Kn > 0) {
      XL_TRY(xl_launch_nco_table_q15(f->qinc[0], f->qinc[1], f->d_qphase, f->d_qphase_next, f->d_qphtab_next, (uint32_t)Kn,
                                     f->stream_nco));
      XL_TRY(hipEventRecord(f->ev_qnco, f->stream_nco));
      f->qspec_valid = true;
      f->qspec_K = Kn;
    }
  }
  *output_len = K;
  return;
fail:
  *output_len = 0;
}

#define XL_CF32_ENTRY(name, ctype, fmt, mode)                                                                    \
  extern "C" void name(const ctype *input, size_t input_len, XL_CF32 **output, size_t *output_len, xlating *filter) { \
    xl_run_cf32(filter, input, input_len, fmt, mode, output, output_len);                                        \
  }
#define XL_Q15_ENTRY(name, ctype, fmt)                                                                           \
  extern "C" void name(const ctype *input, size_t input_len, int16_t **output, size_t *output_len, xlating *filter) { \
    xl_run_q15(filter, input, input_len, fmt, output, output_len);                                               \
  }

// reference src/xlating.c:384-414 (native) and :352-382 (optimized)
XL_CF32_ENTRY(process_native_cu8_cf32, uint8_t, XLF_CU8, 0)
XL_CF32_ENTRY(process_native_cs8_cf32, int8_t, XLF_CS8, 0)
XL_CF32_ENTRY(process_native_cs16_cf32, int16_t, XLF_CS16, 0)
XL_CF32_ENTRY(process_optimized_cu8_cf32, uint8_t, XLF_CU8, 1)
XL_CF32_ENTRY(process_optimized_cs8_cf32, int8_t, XLF_CS8, 1)
XL_CF32_ENTRY(process_optimized_cs16_cf32, int16_t, XLF_CS16, 1)
// extension: cf32 input
XL_CF32_ENTRY(process_native_cf32_cf32, float, XLF_CF32, 0)
XL_CF32_ENTRY(process_optimized_cf32_cf32, float, XLF_CF32, 1)
// reference src/xlating.c:416-447; optimized == native there (:437-447)
XL_Q15_ENTRY(process_native_cu8_cs16, uint8_t, XLF_CU8)
XL_Q15_ENTRY(process_native_cs8_cs16, int8_t, XLF_CS8)
XL_Q15_ENTRY(process_native_cs16_cs16, int16_t, XLF_CS16)
XL_Q15_ENTRY(process_optimized_cu8_cs16, uint8_t, XLF_CU8)
XL_Q15_ENTRY(process_optimized_cs8_cs16, int8_t, XLF_CS8)
XL_Q15_ENTRY(process_optimized_cs16_cs16, int16_t, XLF_CS16)

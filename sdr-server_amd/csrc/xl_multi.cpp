// xl_multi.cpp -- C host of the multi-GPU path (include/xlating_multi.h): G engines, one RCCL broadcast of the raw
// super-block per feed on a communication stream, c mod G sharding.  Reference fan-out replaced:
// src/tcp_server.c:257-271, src/queue.c:87-119.
//
// Per GPU: an engine (which owns the compute stream: XL_STREAM_ENGINE), a communicator, a communication stream, two
// receive buffers and per buffer an event pair: `ready` (recorded on the communication stream behind the broadcast; the
// compute stream waits for it before the engine's launches) and `free` (recorded behind the launches that read the buffer; the
// communication stream waits for it before the broadcast two feeds later overwrites the buffer).  Consecutive feeds
// alternate the buffers, so the broadcast of super-block k+1 runs while super-block k is filtered.  The `free` events live in a
// ring of XL_MRING (feed k: slot k mod XL_MRING) and carry timestamps, and with feed timing on (xlating_multi_feed_timing) every
// broadcast is bracketed by two more events on the communication stream; xlating_multi_feed_timing_read -- called after a sync, never
// on the data path -- reads the latest feeds' broadcast durations and how much of each ran while the PREVIOUS feed's filtering was
// still going on: the measured "is the broadcast hidden" number of a multi-GPU run.
// The driver of GPU 0 also owns `src_done`: recorded behind the root's broadcast (or, without a communicator, behind the
// launches that read d_src in place) -- what xlating_multi_feed_done / _feed_query / _feed_wait_on_stream observe.
//
// Loop-back communicator: world == 1 WITH an id creates a one-rank RCCL communicator and takes the broadcast path of
// world > 1 line for line (receive buffers, ready / free events, source event).  It exists so that a one-GPU box can
// execute that path (tests/test_batch_gpu.py); a deployment with one GPU passes no id and filters d_src in place.
#include <errno.h>
#include <rccl/rccl.h>
#include <stdlib.h>
#include <string.h>

#include <new>
#include <vector>

#include "../../include/xlating_multi.h"
#include "xl_common.h"

#define XL_MRING 16  // feeds whose events are kept (timing samples per read: XL_MRING - 2)

namespace {
struct Gpu {
  int index = 0;   // position in the job (0 .. world-1)
  int device = 0;  // HIP ordinal
  xlating_batch *engine = nullptr;
  ncclComm_t comm = nullptr;
  hipStream_t comm_stream = nullptr;
  void *recv[2] = {nullptr, nullptr};
  hipEvent_t ready[2] = {nullptr, nullptr};
  hipEvent_t free_[XL_MRING] = {};  // feed k: slot k mod XL_MRING (behind the launches that read buffer k & 1)
  bool free_valid[XL_MRING] = {};
  hipEvent_t tb0[XL_MRING] = {}, tb1[XL_MRING] = {};  // around feed k's broadcast
  bool tb_valid[XL_MRING] = {};
  hipEvent_t src_done = nullptr;  // GPU 0's driver only: the latest feed no longer reads d_src
  bool src_valid = false;
};
}  // namespace

struct xlating_multi_t {
  int world = 1;
  bool bcast = false;     // the feed goes through the communicators (world > 1, or the one-rank loop-back)
  std::vector<Gpu> gpus;  // the GPUs this process drives
  size_t recv_bytes = 0;
  uint32_t bps = 2;
  uint64_t feeds = 0;
  bool timing = false;  // bracket the broadcasts (first local GPU)
  double bcast_ms = 0.0, hidden_ms = 0.0;
  int timed = 0;
};

#define XL_NCCL(expr)                                                                          \
  do {                                                                                         \
    ncclResult_t xl_r_ = (expr);                                                               \
    if (xl_r_ != ncclSuccess) {                                                                \
      XL_LOG_ERR("%s failed: %s (%s:%d)", #expr, ncclGetErrorString(xl_r_), __FILE__, __LINE__); \
      goto fail;                                                                               \
    }                                                                                          \
  } while (0)

extern "C" int xlating_multi_unique_id(void *id) {
  if (id == nullptr) return -EINVAL;
  static_assert(sizeof(ncclUniqueId) == XLATING_MULTI_ID_BYTES, "id size");
  ncclUniqueId u;
  if (ncclGetUniqueId(&u) != ncclSuccess) return -EIO;
  memcpy(id, &u, sizeof(u));
  return 0;
}

static int xl_multi_open_gpu(xlating_multi *m, Gpu &g, uint32_t fs, int fmt, uint32_t max_len, unsigned gcap) {
  int rc = xlating_batch_create_grouped(fs, fmt, max_len, gcap, g.device, &g.engine);
  if (rc != 0) return rc;
  XL_TRY(hipSetDevice(g.device));
  XL_TRY(hipStreamCreateWithFlags(&g.comm_stream, hipStreamNonBlocking));
  if (g.index == 0) XL_TRY(hipEventCreateWithFlags(&g.src_done, hipEventDisableTiming));
  for (int i = 0; i < 2; ++i) {
    if (m->bcast) XL_TRY(hipMalloc(&g.recv[i], m->recv_bytes));
    XL_TRY(hipEventCreateWithFlags(&g.ready[i], hipEventDisableTiming));
  }
  for (int i = 0; i < XL_MRING; ++i) {
    XL_TRY(hipEventCreate(&g.free_[i]));
    XL_TRY(hipEventCreate(&g.tb0[i]));
    XL_TRY(hipEventCreate(&g.tb1[i]));
  }
  return 0;
fail:
  return xl_errno_of_last_hip_error();
}

static int xl_multi_check(uint32_t fs, int fmt, uint32_t max_len, unsigned gcap) {
  if (fs == 0 || fmt < XL_FMT_CU8 || fmt > XL_FMT_CF32 || max_len < 2 || gcap < 1 || gcap > 64) return -EINVAL;
  return 0;
}

extern "C" int xlating_multi_create_rank(int rank, int world, const void *id, uint32_t fs, int fmt, uint32_t max_len,
                                         unsigned gcap, int device, xlating_multi **out) {
  if (out == nullptr || world < 1 || rank < 0 || rank >= world || (world > 1 && id == nullptr) ||
      xl_multi_check(fs, fmt, max_len, gcap) != 0)
    return -EINVAL;
  const int dev = xl_hip_select_device(device);
  if (dev < 0) {
    XL_LOG_ERR("no usable HIP device (%s); this build has no CPU arithmetic path", xlating_hip_device_info());
    return -ENODEV;
  }
  xlating_multi *m = new (std::nothrow) xlating_multi_t();
  if (m == nullptr) return -ENOMEM;
  m->world = world;
  m->bcast = world > 1 || id != nullptr;
  m->bps = fmt <= XL_FMT_CS8 ? 2u : (fmt == XL_FMT_CS16 ? 4u : 8u);
  m->recv_bytes = (size_t)(max_len / 2) * gcap * m->bps + 16;
  m->gpus.resize(1);
  m->gpus[0].index = rank;
  m->gpus[0].device = dev;
  int rc = xl_multi_open_gpu(m, m->gpus[0], fs, fmt, max_len, gcap);
  if (rc != 0) {
    xlating_multi_destroy(m);
    return rc;
  }
  if (m->bcast) {
    ncclUniqueId u;
    memcpy(&u, id, sizeof(u));
    XL_NCCL(ncclCommInitRank(&m->gpus[0].comm, world, u, rank));
  }
  *out = m;
  return 0;
fail:
  xlating_multi_destroy(m);
  return -EIO;
}

extern "C" int xlating_multi_create_local(int ngpus, const int *devices, uint32_t fs, int fmt, uint32_t max_len,
                                          unsigned gcap, xlating_multi **out) {
  if (out == nullptr || ngpus < 1 || ngpus > 64 || xl_multi_check(fs, fmt, max_len, gcap) != 0) return -EINVAL;
  if (xl_hip_select_device(devices ? devices[0] : 0) < 0) {
    XL_LOG_ERR("no usable HIP device (%s); this build has no CPU arithmetic path", xlating_hip_device_info());
    return -ENODEV;
  }
  xlating_multi *m = new (std::nothrow) xlating_multi_t();
  if (m == nullptr) return -ENOMEM;
  m->world = ngpus;
  m->bcast = ngpus > 1;
  m->bps = fmt <= XL_FMT_CS8 ? 2u : (fmt == XL_FMT_CS16 ? 4u : 8u);
  m->recv_bytes = (size_t)(max_len / 2) * gcap * m->bps + 16;
  m->gpus.resize((size_t)ngpus);
  std::vector<int> devs((size_t)ngpus);
  std::vector<ncclComm_t> comms((size_t)ngpus, nullptr);
  for (int i = 0; i < ngpus; ++i) {
    devs[i] = devices ? devices[i] : i;
    if (xl_hip_select_device(devs[i]) < 0) {
      xlating_multi_destroy(m);
      return -ENODEV;
    }
    m->gpus[i].index = i;
    m->gpus[i].device = devs[i];
    int rc = xl_multi_open_gpu(m, m->gpus[i], fs, fmt, max_len, gcap);
    if (rc != 0) {
      xlating_multi_destroy(m);
      return rc;
    }
  }
  if (ngpus > 1) {
    XL_NCCL(ncclCommInitAll(comms.data(), ngpus, devs.data()));
    for (int i = 0; i < ngpus; ++i) m->gpus[i].comm = comms[i];
  }
  *out = m;
  return 0;
fail:
  xlating_multi_destroy(m);
  return -EIO;
}

extern "C" int xlating_multi_world(const xlating_multi *m) { return m ? m->world : 0; }
extern "C" int xlating_multi_local(const xlating_multi *m) { return m ? (int)m->gpus.size() : 0; }

static Gpu *xl_multi_find(xlating_multi *m, int gpu) {
  for (Gpu &g : m->gpus)
    if (g.index == gpu) return &g;
  return nullptr;
}

extern "C" xlating_batch *xlating_multi_engine(xlating_multi *m, int gpu) {
  if (m == nullptr) return nullptr;
  Gpu *g = xl_multi_find(m, gpu);
  return g ? g->engine : nullptr;
}

extern "C" int xlating_multi_add_client(xlating_multi *m, int global_client, uint32_t decimation, const float *taps,
                                        size_t taps_len, int32_t center_freq) {
  if (m == nullptr || global_client < 0) return -EINVAL;
  Gpu *g = xl_multi_find(m, global_client % m->world);  // client c -> GPU (c mod G)
  if (g == nullptr) return -ENOENT;
  return xlating_batch_add_client(g->engine, decimation, taps, taps_len, center_freq);
}

extern "C" int xlating_multi_feed(xlating_multi *m, const void *d_src, size_t input_len, unsigned nblocks, int mode) {
  if (m == nullptr || nblocks < 1) return -EINVAL;
  const size_t bytes = (input_len / 2) * nblocks * m->bps;
  if (m->bcast && bytes > m->recv_bytes) return -EINVAL;
  Gpu *root = xl_multi_find(m, 0);
  if (root != nullptr && d_src == nullptr && bytes > 0) return -EINVAL;  // the driver of GPU 0 holds the source
  const int i = (int)(m->feeds & 1);
  const int k4 = (int)(m->feeds % XL_MRING), k2 = (int)((m->feeds + XL_MRING - 2) % XL_MRING);  // this feed's slot, feed - 2's
  if (!m->bcast) {  // one GPU, no communicator: the engine reads d_src in place; the source is free behind its launches
    // (no per-feed event: a launch that carries one delays the next launch by ~8 us, a fifth of a one-block feed; the three
    // "is the source free" calls below ask the engine's stream instead, or record src_done when somebody wants to wait on it)
    Gpu &g = m->gpus[0];
    int rc = xlating_batch_process_device_group_ev(g.engine, d_src, input_len, nblocks, mode, XL_STREAM_ENGINE, nullptr, nullptr);
    if (rc != 0) return rc;
    g.src_valid = true;
    m->feeds++;
    return 0;
  }
  // ---- the path's only exchange step: GPU 0's raw blocks -> every GPU, on the communication streams
  for (Gpu &g : m->gpus) {
    XL_TRY(hipSetDevice(g.device));
    if (g.free_valid[k2]) XL_TRY(hipStreamWaitEvent(g.comm_stream, g.free_[k2], 0));  // (the feed that read this buffer last)
  }
  if (m->timing) {
    Gpu &g = m->gpus[0];
    XL_TRY(hipSetDevice(g.device));
    g.tb_valid[k4] = false;
    XL_TRY(hipEventRecord(g.tb0[k4], g.comm_stream));
  }
  if (m->gpus.size() > 1) XL_NCCL(ncclGroupStart());
  for (Gpu &g : m->gpus) {
    XL_TRY(hipSetDevice(g.device));
    XL_NCCL(ncclBroadcast(g.index == 0 ? d_src : g.recv[i], g.recv[i], bytes, ncclUint8, 0, g.comm, g.comm_stream));
  }
  if (m->gpus.size() > 1) XL_NCCL(ncclGroupEnd());
  if (m->timing) {
    Gpu &g = m->gpus[0];
    XL_TRY(hipSetDevice(g.device));
    XL_TRY(hipEventRecord(g.tb1[k4], g.comm_stream));
    g.tb_valid[k4] = true;
  }
  if (root != nullptr) {  // the broadcast is the only reader of d_src
    XL_TRY(hipSetDevice(root->device));
    XL_TRY(hipEventRecord(root->src_done, root->comm_stream));
    root->src_valid = true;
  }
  // ---- independent per-GPU work: every local engine filters its own clients from its receive buffer
  for (Gpu &g : m->gpus) {
    XL_TRY(hipSetDevice(g.device));
    XL_TRY(hipEventRecord(g.ready[i], g.comm_stream));
    // on the engine's own compute stream (CU-masked for these calls): wait for the broadcast, filter, mark the buffer free
    int rc = xlating_batch_process_device_group_ev(g.engine, g.recv[i], input_len, nblocks, mode, XL_STREAM_ENGINE, g.ready[i],
                                                   g.free_[k4]);
    if (rc != 0) return rc;
    g.free_valid[k4] = true;
  }
  m->feeds++;
  return 0;
fail:
  return -EIO;
}

// ---- "may d_src of the latest feed be overwritten?"  Only the driver of GPU 0 holds a source; elsewhere: yes, at once.
extern "C" int xlating_multi_feed_done(xlating_multi *m) {
  if (m == nullptr) return -EINVAL;
  Gpu *root = xl_multi_find(m, 0);
  if (root == nullptr || !root->src_valid) return 0;
  if (hipSetDevice(root->device) != hipSuccess) return -EIO;
  if (!m->bcast) return xlating_batch_sync(root->engine) == 0 ? 0 : -EIO;  // (in place: the readers are the engine's launches)
  return hipEventSynchronize(root->src_done) == hipSuccess ? 0 : -EIO;
}

extern "C" int xlating_multi_feed_query(xlating_multi *m) {
  if (m == nullptr) return -EINVAL;
  Gpu *root = xl_multi_find(m, 0);
  if (root == nullptr || !root->src_valid) return 1;
  if (hipSetDevice(root->device) != hipSuccess) return -EIO;
  if (!m->bcast) {
    const int q = xlating_batch_query(root->engine);
    return q < 0 ? -EIO : q;
  }
  const hipError_t e = hipEventQuery(root->src_done);
  if (e == hipSuccess) return 1;
  if (e == hipErrorNotReady) {
    (void)hipGetLastError();  // (not an error: clear the sticky code)
    return 0;
  }
  return -EIO;
}

extern "C" int xlating_multi_feed_wait_on_stream(xlating_multi *m, void *hip_stream) {
  if (m == nullptr) return -EINVAL;
  Gpu *root = xl_multi_find(m, 0);
  if (root == nullptr || !root->src_valid) return 0;
  if (hipSetDevice(root->device) != hipSuccess) return -EIO;
  if (!m->bcast && xlating_batch_record_event(root->engine, root->src_done) != 0) return -EIO;  // (behind the latest feed's launches, now)
  return hipStreamWaitEvent(reinterpret_cast<hipStream_t>(hip_stream), root->src_done, 0) == hipSuccess ? 0 : -EIO;
}

extern "C" int xlating_multi_feed_timing(xlating_multi *m, int enable) {
  if (m == nullptr) return -EINVAL;
  m->timing = enable != 0 && m->bcast;
  return 0;
}

// The latest feeds whose three timestamps are complete (call it after xlating_multi_sync): feed j's broadcast [tb0, tb1] against the
// end of feed j - 1's filtering (free_ of j - 1).  At most XL_MRING - 2 samples: older events have been re-recorded.
extern "C" int xlating_multi_feed_timing_read(xlating_multi *m, double *bcast_ms_total, double *hidden_ms_total, int reset) {
  if (m == nullptr) return -EINVAL;
  if (!m->gpus.empty() && m->timing && hipSetDevice(m->gpus[0].device) == hipSuccess) {
    Gpu &g = m->gpus[0];
    const uint64_t lo = m->feeds > (uint64_t)(XL_MRING - 2) ? m->feeds - (XL_MRING - 2) : 1;
    for (uint64_t j = lo; j < m->feeds; ++j) {
      const int sj = (int)(j % XL_MRING), sp = (int)((j - 1) % XL_MRING);
      if (!g.tb_valid[sj] || !g.free_valid[sp]) continue;
      if (hipEventQuery(g.tb1[sj]) != hipSuccess || hipEventQuery(g.free_[sp]) != hipSuccess) continue;
      float bms = 0.0f, hms = 0.0f;
      if (hipEventElapsedTime(&bms, g.tb0[sj], g.tb1[sj]) == hipSuccess && hipEventElapsedTime(&hms, g.tb0[sj], g.free_[sp]) == hipSuccess) {
        m->bcast_ms += bms;
        m->hidden_ms += hms <= 0.0f ? 0.0 : (hms >= bms ? bms : hms);
        m->timed++;
      }
      g.tb_valid[sj] = false;  // (counted once)
    }
    (void)hipGetLastError();  // (hipErrorNotReady of a query is not an error)
  }
  if (bcast_ms_total) *bcast_ms_total = m->bcast_ms;
  if (hidden_ms_total) *hidden_ms_total = m->hidden_ms;
  const int n = m->timed;
  if (reset) {
    m->bcast_ms = m->hidden_ms = 0.0;
    m->timed = 0;
  }
  return n;
}

extern "C" int xlating_multi_comm_count(const xlating_multi *m) {
  if (m == nullptr) return -EINVAL;
  for (const Gpu &g : m->gpus)
    if (g.comm != nullptr) {
      int n = 0;
      return ncclCommCount(g.comm, &n) == ncclSuccess ? n : -EIO;
    }
  return 0;
}

extern "C" int xlating_multi_sync(xlating_multi *m) {
  if (m == nullptr) return -EINVAL;
  for (Gpu &g : m->gpus) {
    if (hipSetDevice(g.device) != hipSuccess) return -EIO;
    if (g.comm_stream && hipStreamSynchronize(g.comm_stream) != hipSuccess) return -EIO;
    if (g.engine && xlating_batch_sync(g.engine) != 0) return -EIO;
  }
  return 0;
}

extern "C" void xlating_multi_destroy(xlating_multi *m) {
  if (m == nullptr) return;
  (void)xlating_multi_sync(m);
  for (Gpu &g : m->gpus) {
    (void)hipSetDevice(g.device);
    if (g.engine) xlating_batch_destroy(g.engine);
    if (g.comm) (void)ncclCommDestroy(g.comm);
    for (int i = 0; i < 2; ++i) {
      if (g.recv[i]) (void)hipFree(g.recv[i]);
      if (g.ready[i]) (void)hipEventDestroy(g.ready[i]);
    }
    for (int i = 0; i < XL_MRING; ++i) {
      if (g.free_[i]) (void)hipEventDestroy(g.free_[i]);
      if (g.tb0[i]) (void)hipEventDestroy(g.tb0[i]);
      if (g.tb1[i]) (void)hipEventDestroy(g.tb1[i]);
    }
    if (g.src_done) (void)hipEventDestroy(g.src_done);
    if (g.comm_stream) (void)hipStreamDestroy(g.comm_stream);
  }
  delete m;
}

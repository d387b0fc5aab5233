// xl_sinks.cpp -- per-client output sinks of the batched path (include/xlating_sinks.h).
//
// What the reference does per client thread (src/dsp_worker.c:10-39 write_to_file / write_to_socket, :74-86 "close the
// client on failure", :126-144 <base>/<id>.cf32[.gz]) is done here by a small pool of writer threads behind bounded
// per-client byte queues.  Host-only: no HIP in this file.
#include <errno.h>
#include <fcntl.h>
#include <signal.h>
#include <stdio.h>
#include <string.h>
#include <sys/eventfd.h>
#include <sys/socket.h>
#include <sys/types.h>
#include <unistd.h>
#include <zlib.h>

#include <poll.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <map>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/xlating_batch.h"
#include "../../include/xlating_sinks.h"
#include "xl_common.h"

// ThreadSanitizer (tools/sanitize.sh) learns a mutex's lifetime from pthread_mutex_init / _destroy; libstdc++'s std::mutex
// calls neither (constexpr constructor, trivial destructor), so a mutex living in heap memory that once held another
// object's mutex looks "already destroyed" to the tool, which then ignores its ordering and reports every pair of
// correctly locked accesses as a race.  These annotations tell it where a mutex starts and ends; they compile to nothing
// outside -fsanitize=thread builds.
#if defined(__SANITIZE_THREAD__)
extern "C" void __tsan_mutex_create(void *addr, unsigned flags);
extern "C" void __tsan_mutex_destroy(void *addr, unsigned flags);
#define XL_TSAN_MUTEX_CREATE(m) __tsan_mutex_create((m)->native_handle(), 0)
#define XL_TSAN_MUTEX_DESTROY(m) __tsan_mutex_destroy((m)->native_handle(), 0)
#else
#define XL_TSAN_MUTEX_CREATE(m) ((void)0)
#define XL_TSAN_MUTEX_DESTROY(m) ((void)0)
#endif

// Timed wait with a predicate.  condition_variable::wait_for measures on the steady clock, which glibc implements with
// pthread_cond_clockwait -- a call this image's libtsan (GCC 11) does not intercept: it misses the unlock / relock inside
// the wait, believes the waiter still holds the mutex and reports "double lock" plus a race for every access the other
// side makes meanwhile.  The instrumented build waits on the system clock instead (pthread_cond_timedwait, intercepted);
// production keeps the steady clock (immune to clock steps).
template <class Rep, class Period, class Pred>
static bool xl_wait_for(std::condition_variable &cv, std::unique_lock<std::mutex> &lk, std::chrono::duration<Rep, Period> d, Pred pred) {
#if defined(__SANITIZE_THREAD__)
  return cv.wait_until(lk, std::chrono::system_clock::now() + d, pred);
#else
  return cv.wait_for(lk, d, pred);
#endif
}

namespace {

enum Kind { K_FD = 0, K_FILE = 1, K_GZ = 2 };

struct Sink {
  int id = -1;
  Kind kind = K_FD;
  int fd = -1;
  bool close_fd = false;
  FILE *file = nullptr;
  gzFile gz = nullptr;
  std::vector<uint8_t> ring;  // byte queue
  size_t head = 0, used = 0;  // head = read position
  bool busy = false;          // the writer thread holds bytes popped from the ring (in flight, or pending below)
  bool failed = false, reported = false;
  std::atomic<bool> cancel{false};  // abandon an in-flight write (the sink failed or is being torn down)
  // descriptor sinks only, touched by the writer thread alone: bytes popped from the ring that the peer has not taken
  // yet.  A peer that stops reading parks its bytes here and the thread goes on serving its other sinks -- one stalled
  // client must never drop another (the reference has a thread per client, dsp_worker.c:41-88).
  std::vector<uint8_t> pend;
  size_t pend_off = 0;
  bool is_sock = true;
};

struct Worker {
  Worker() { XL_TSAN_MUTEX_CREATE(&m); }
  ~Worker() { XL_TSAN_MUTEX_DESTROY(&m); }
  std::mutex m;
  std::condition_variable cv_work, cv_done;
  std::map<int, std::unique_ptr<Sink>> sinks;  // the sinks this thread serves (id % nthreads)
  std::thread th;
  bool stop = false;
  int rr = 0;  // round-robin start for fairness
  int wake_fd = -1;  // eventfd: wakes the thread out of its poll() on the pending descriptors
};

}  // namespace

struct xlating_sinks_t {
  xlating_sinks_t() { XL_TSAN_MUTEX_CREATE(&stats_m); }
  ~xlating_sinks_t() { XL_TSAN_MUTEX_DESTROY(&stats_m); }
  std::vector<std::unique_ptr<Worker>> workers;
  size_t queue_bytes = 0;
  std::mutex stats_m;
  uint64_t bytes_written = 0, blocks_dropped = 0;
};

namespace {

const size_t kChunk = 1u << 20;  // most bytes a writer takes out of a queue at a time

// write_to_socket() semantics (dsp_worker.c:28-39): every byte, or failure -- but never stuck on a peer: the writes are
// non-blocking.  Pushes as much of the sink's pending bytes as the descriptor takes now.
// Returns 1 = all gone, 0 = the descriptor is full (try again when it is writable), -1 = failed.
int xl_push_pending(Sink *s) {
  while (s->pend_off < s->pend.size()) {
    const uint8_t *p = s->pend.data() + s->pend_off;
    const size_t n = s->pend.size() - s->pend_off;
    ssize_t w = s->is_sock ? send(s->fd, p, n, MSG_NOSIGNAL | MSG_DONTWAIT) : write(s->fd, p, n);
    if (w < 0) {
      if (s->is_sock && errno == ENOTSOCK) {
        s->is_sock = false;
        const int fl = fcntl(s->fd, F_GETFL);
        if (fl >= 0 && !(fl & O_NONBLOCK)) (void)fcntl(s->fd, F_SETFL, fl | O_NONBLOCK);  // pipes / files: same rule
        continue;
      }
      if (errno == EINTR) continue;
      if (errno == EAGAIN || errno == EWOULDBLOCK) return 0;
      return -1;
    }
    s->pend_off += (size_t)w;
  }
  s->pend.clear();
  s->pend_off = 0;
  return 1;
}

bool xl_sink_emit(Sink *s, const uint8_t *p, size_t n) {
  switch (s->kind) {
    case K_FD: return false;  // (descriptor sinks go through xl_push_pending)
    case K_FILE: return fwrite(p, 1, n, s->file) == n;  // short write (disk full) ends the client (dsp_worker.c:20-24)
    case K_GZ: {
      while (n > 0) {
        const unsigned part = (unsigned)std::min<size_t>(n, 1u << 30);
        if (gzwrite(s->gz, p, part) != (int)part) return false;
        p += part;
        n -= part;
      }
      return true;
    }
  }
  return false;
}

void xl_sink_close(Sink *s) {
  if (s->file) fclose(s->file);
  if (s->gz) gzclose(s->gz);
  if (s->kind == K_FD && s->close_fd && s->fd >= 0) close(s->fd);
  s->file = nullptr;
  s->gz = nullptr;
  s->fd = -1;
}

void xl_worker_main(xlating_sinks *S, Worker *w) {
  // a peer that went away must surface as EPIPE from write(), not as a signal (sockets use MSG_NOSIGNAL anyway)
  sigset_t set;
  sigemptyset(&set);
  sigaddset(&set, SIGPIPE);
  (void)pthread_sigmask(SIG_BLOCK, &set, nullptr);
  std::vector<uint8_t> buf;
  std::vector<Sink *> parked;  // descriptor sinks with pending bytes (busy == true)
  std::unique_lock<std::mutex> lk(w->m);
  auto finish = [&](Sink *s, bool ok, size_t n) {  // (lock held)
    s->busy = false;
    if (ok) {
      std::lock_guard<std::mutex> g(S->stats_m);
      S->bytes_written += n;
    } else {
      s->failed = true;
      s->used = 0;
      s->pend.clear();
      s->pend_off = 0;
    }
    w->cv_done.notify_all();
  };
  for (;;) {
    bool progressed = false;
    // ---- 1. parked descriptor sinks: push what their peers take now (never waits)
    for (size_t k = 0; k < parked.size();) {
      Sink *s = parked[k];
      const size_t total = s->pend.size();
      lk.unlock();
      const int rc = s->cancel.load() ? -1 : xl_push_pending(s);
      lk.lock();
      if (rc == 0) {
        ++k;
        continue;
      }
      finish(s, rc > 0, total);
      parked[k] = parked.back();
      parked.pop_back();
      progressed = true;
    }
    // ---- 2. one sink with queued bytes, round-robin
    Sink *pick = nullptr;
    if (!w->sinks.empty()) {
      auto it = w->sinks.lower_bound(w->rr);
      for (size_t k = 0; k < w->sinks.size(); ++k, ++it) {
        if (it == w->sinks.end()) it = w->sinks.begin();
        Sink *s = it->second.get();
        if (s->used > 0 && !s->busy && !s->failed) {
          pick = s;
          w->rr = s->id + 1;
          break;
        }
      }
    }
    if (pick != nullptr) {
      const size_t n = std::min(pick->used, kChunk);
      std::vector<uint8_t> &dst = pick->kind == K_FD ? pick->pend : buf;
      dst.resize(n);
      const size_t cap = pick->ring.size();
      const size_t first = std::min(n, cap - pick->head);
      memcpy(dst.data(), pick->ring.data() + pick->head, first);
      memcpy(dst.data() + first, pick->ring.data(), n - first);
      pick->head = (pick->head + n) % cap;
      pick->used -= n;
      pick->busy = true;
      pick->pend_off = 0;
      lk.unlock();
      int rc;
      if (pick->kind == K_FD) rc = xl_push_pending(pick);
      else rc = xl_sink_emit(pick, buf.data(), n) ? 1 : -1;
      lk.lock();
      if (rc == 0) parked.push_back(pick);  // the peer is not taking bytes right now: come back to it, serve the others
      else finish(pick, rc > 0, n);
      continue;
    }
    if (progressed) continue;
    if (parked.empty()) {
      if (w->stop) return;
      w->cv_work.wait(lk);
      continue;
    }
    // ---- 3. only parked sinks are left: sleep until one of their descriptors is writable, new bytes are queued
    // (wake_fd) or 50 ms pass (cancel flags are polled at that rate)
    if (w->stop) {
      bool all_cancelled = true;
      for (Sink *s : parked) all_cancelled = all_cancelled && s->cancel.load();
      if (!all_cancelled) {  // teardown: nobody will wait for these peers
        for (Sink *s : parked) s->cancel.store(true);
      }
    }
    std::vector<struct pollfd> pfds;
    for (Sink *s : parked) pfds.push_back({s->fd, POLLOUT, 0});
    if (w->wake_fd >= 0) pfds.push_back({w->wake_fd, POLLIN, 0});
    lk.unlock();
    (void)poll(pfds.data(), (nfds_t)pfds.size(), 50);
    if (w->wake_fd >= 0) {
      uint64_t v;
      while (read(w->wake_fd, &v, sizeof(v)) > 0) {
      }
    }
    lk.lock();
  }
}

Worker *xl_worker_of(xlating_sinks *S, int id) { return S->workers[(size_t)((unsigned)id % S->workers.size())].get(); }

// takes ownership of `s` only on success
int xl_attach(xlating_sinks *S, std::unique_ptr<Sink> &s) {
  Worker *w = xl_worker_of(S, s->id);
  try {
    s->ring.resize(S->queue_bytes);
    std::lock_guard<std::mutex> g(w->m);
    if (w->sinks.count(s->id)) return -EEXIST;
    const int id = s->id;
    w->sinks[id] = std::move(s);
  } catch (...) {
    return -ENOMEM;
  }
  return 0;
}

}  // namespace

extern "C" int xlating_sinks_create(unsigned writer_threads, size_t queue_bytes, xlating_sinks **out) {
  if (out == nullptr || writer_threads == 0 || writer_threads > 256 || queue_bytes < 8) return -EINVAL;
  xlating_sinks *S = new (std::nothrow) xlating_sinks_t();
  if (S == nullptr) return -ENOMEM;
  S->queue_bytes = queue_bytes;
  try {
    for (unsigned i = 0; i < writer_threads; ++i) {
      S->workers.emplace_back(new Worker());
      S->workers.back()->wake_fd = eventfd(0, EFD_NONBLOCK | EFD_CLOEXEC);
    }
    for (auto &w : S->workers) w->th = std::thread(xl_worker_main, S, w.get());
  } catch (...) {
    xlating_sinks_destroy(S);
    return -ENOMEM;
  }
  *out = S;
  return 0;
}

extern "C" int xlating_sinks_attach_fd(xlating_sinks *S, int client_id, int fd, int close_on_detach) {
  if (S == nullptr || client_id < 0 || fd < 0) return -EINVAL;
  std::unique_ptr<Sink> s(new (std::nothrow) Sink());
  if (!s) return -ENOMEM;
  s->id = client_id;
  s->kind = K_FD;
  s->fd = fd;
  s->close_fd = false;  // (stays the caller's if attaching fails)
  const int rc = xl_attach(S, s);
  if (rc == 0 && close_on_detach) {
    Worker *w = xl_worker_of(S, client_id);
    std::lock_guard<std::mutex> g(w->m);
    auto it = w->sinks.find(client_id);
    if (it != w->sinks.end()) it->second->close_fd = true;
  }
  return rc;
}

extern "C" int xlating_sinks_attach_file(xlating_sinks *S, int client_id, const char *base_path, int use_gzip) {
  if (S == nullptr || client_id < 0 || base_path == nullptr) return -EINVAL;
  {
    Worker *w = xl_worker_of(S, client_id);
    std::lock_guard<std::mutex> g(w->m);
    if (w->sinks.count(client_id)) return -EEXIST;
  }
  std::unique_ptr<Sink> s(new (std::nothrow) Sink());
  if (!s) return -ENOMEM;
  s->id = client_id;
  const std::string path = std::string(base_path) + "/" + std::to_string(client_id) + (use_gzip ? ".cf32.gz" : ".cf32");
  if (use_gzip) {
    s->kind = K_GZ;
    s->gz = gzopen(path.c_str(), "wb");
    if (s->gz == nullptr) {
      const int e = errno ? errno : EIO;
      XL_LOG_ERR("unable to open gz file for output: %s", path.c_str());
      return -e;
    }
  } else {
    s->kind = K_FILE;
    s->file = fopen(path.c_str(), "wb");
    if (s->file == nullptr) {
      const int e = errno ? errno : EIO;
      XL_LOG_ERR("unable to open file for output: %s", path.c_str());
      return -e;
    }
  }
  const int rc = xl_attach(S, s);
  if (rc != 0 && s) xl_sink_close(s.get());
  return rc;
}

extern "C" int xlating_sinks_write(xlating_sinks *S, int client_id, const float *samples, size_t n_complex) {
  if (S == nullptr || client_id < 0 || (samples == nullptr && n_complex > 0)) return -EINVAL;
  Worker *w = xl_worker_of(S, client_id);
  const size_t n = n_complex * 2 * sizeof(float);
  std::unique_lock<std::mutex> lk(w->m);
  auto it = w->sinks.find(client_id);
  if (it == w->sinks.end()) return -ENOENT;
  Sink *s = it->second.get();
  if (!s->failed && n > s->ring.size() - s->used) {
    // the peer does not keep up: the reference would block its dsp thread until the queue overruns and then drop the
    // client's blocks; here the client is failed at once
    s->failed = true;
    s->used = 0;
    s->cancel.store(true);
  }
  if (s->failed) {
    lk.unlock();
    std::lock_guard<std::mutex> g(S->stats_m);
    S->blocks_dropped++;
    return -EPIPE;
  }
  if (n == 0) return 0;
  const size_t cap = s->ring.size();
  const size_t tail = (s->head + s->used) % cap;
  const size_t first = std::min(n, cap - tail);
  const uint8_t *src = reinterpret_cast<const uint8_t *>(samples);
  memcpy(s->ring.data() + tail, src, first);
  memcpy(s->ring.data(), src + first, n - first);
  s->used += n;
  w->cv_work.notify_one();
  if (w->wake_fd >= 0) {
    const uint64_t one = 1;
    (void)!write(w->wake_fd, &one, sizeof(one));
  }
  return 0;
}

extern "C" int xlating_sinks_submit(xlating_sinks *S, struct xlating_batch_t *engine) {
  if (S == nullptr || engine == nullptr) return -EINVAL;
  int queued = 0;
  for (auto &w : S->workers) {
    std::vector<int> ids;
    {
      std::lock_guard<std::mutex> g(w->m);
      for (auto &kv : w->sinks) ids.push_back(kv.first);
    }
    for (int id : ids) {
      const float *out = nullptr;
      size_t n = 0;
      if (xlating_batch_output_host(engine, id, &out, &n) != 0) continue;  // not a live client / nothing fetched
      if (xlating_sinks_write(S, id, out, n) == 0) ++queued;
    }
  }
  return queued;
}

extern "C" size_t xlating_sinks_failed(xlating_sinks *S, int *ids, size_t cap) {
  if (S == nullptr || (ids == nullptr && cap > 0)) return 0;
  size_t n = 0;
  for (auto &w : S->workers) {
    std::lock_guard<std::mutex> g(w->m);
    for (auto &kv : w->sinks) {
      Sink *s = kv.second.get();
      if (s->failed && !s->reported && n < cap) {
        ids[n++] = s->id;
        s->reported = true;
      }
    }
  }
  return n;
}

extern "C" int xlating_sinks_flush(xlating_sinks *S) {
  if (S == nullptr) return -EINVAL;
  for (auto &w : S->workers) {
    std::unique_lock<std::mutex> lk(w->m);
    w->cv_done.wait(lk, [&] {
      for (auto &kv : w->sinks) {
        Sink *s = kv.second.get();
        if (s->busy || (s->used > 0 && !s->failed)) return false;
      }
      return true;
    });
    for (auto &kv : w->sinks)
      if (kv.second->file && !kv.second->failed) (void)fflush(kv.second->file);
  }
  return 0;
}

extern "C" int xlating_sinks_detach(xlating_sinks *S, int client_id) {
  if (S == nullptr || client_id < 0) return -EINVAL;
  Worker *w = xl_worker_of(S, client_id);
  std::unique_ptr<Sink> s;
  {
    std::unique_lock<std::mutex> lk(w->m);
    auto it = w->sinks.find(client_id);
    if (it == w->sinks.end()) return -ENOENT;
    Sink *p = it->second.get();
    if (p->failed) p->cancel.store(true);
    // a healthy peer gets two seconds to take what is queued; one that has stopped reading is cut off
    if (!xl_wait_for(w->cv_done, lk, std::chrono::seconds(2), [&] { return !p->busy && (p->used == 0 || p->failed); })) {
      p->cancel.store(true);
      p->failed = true;
      p->used = 0;
      if (w->wake_fd >= 0) {
        const uint64_t one = 1;
        (void)!write(w->wake_fd, &one, sizeof(one));
      }
      w->cv_done.wait(lk, [&] { return !p->busy; });
    }
    s = std::move(it->second);
    w->sinks.erase(it);
  }
  xl_sink_close(s.get());
  return 0;
}

extern "C" void xlating_sinks_stats(xlating_sinks *S, uint64_t *bytes_written, uint64_t *blocks_dropped) {
  if (S == nullptr) return;
  std::lock_guard<std::mutex> g(S->stats_m);
  if (bytes_written) *bytes_written = S->bytes_written;
  if (blocks_dropped) *blocks_dropped = S->blocks_dropped;
}

extern "C" void xlating_sinks_destroy(xlating_sinks *S) {
  if (S == nullptr) return;
  for (auto &w : S->workers) {
    {
      // give queued bytes two seconds to drain, then abandon whatever a stuck peer still holds up
      std::unique_lock<std::mutex> lk(w->m);
      (void)xl_wait_for(w->cv_done, lk, std::chrono::seconds(2), [&] {
        for (auto &kv : w->sinks) {
          Sink *s = kv.second.get();
          if (s->busy || (s->used > 0 && !s->failed)) return false;
        }
        return true;
      });
      for (auto &kv : w->sinks) {
        kv.second->cancel.store(true);
        kv.second->failed = true;
        kv.second->used = 0;
      }
      w->stop = true;
    }
    w->cv_work.notify_all();
    if (w->wake_fd >= 0) {
      const uint64_t one = 1;
      (void)!write(w->wake_fd, &one, sizeof(one));
    }
    if (w->th.joinable()) w->th.join();
    if (w->wake_fd >= 0) close(w->wake_fd);
    w->wake_fd = -1;
    for (auto &kv : w->sinks) xl_sink_close(kv.second.get());
    w->sinks.clear();
  }
  delete S;
}

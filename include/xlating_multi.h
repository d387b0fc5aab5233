/*
 * include/xlating_multi.h -- C host for the multi-GPU path (BASELINE.json configs[3]).
 *
 * The reference fans one IQ block out to its clients inside one process (src/tcp_server.c:257-271 -> src/queue.c:87-119).
 * Across the GPUs of one node the same fan-out is: client c lives on GPU (c mod G); the raw block is BROADCAST once over
 * RCCL/xGMI (the path's only exchange step) and every GPU then filters its own clients -- no other collective, outputs
 * stay per client.  This library is that host in C: it owns the engines (include/xlating_batch.h), the RCCL
 * communicators, two receive buffers per GPU and the streams/events that overlap the broadcast of super-block k+1 with
 * the filtering of super-block k.
 *
 * Two ways to run it:
 *   one process per GPU   (the torch.distributed.run launch shape of bench.py): rank 0 obtains an id with
 *                         xlating_multi_unique_id(), hands the 128 bytes to the other processes by any means, and every
 *                         rank calls xlating_multi_create_rank();
 *   one process, G GPUs   (how a C server would embed it): xlating_multi_create_local() opens G devices at once
 *                         (ncclCommInitAll) and xlating_multi_feed() drives them all.
 * Plain C ABI; links librccl and libxlating_hip.
 */
#ifndef SDR_SERVER_AMD_XLATING_MULTI_H_
#define SDR_SERVER_AMD_XLATING_MULTI_H_

#include <stddef.h>
#include <stdint.h>

#include "xlating_batch.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct xlating_multi_t xlating_multi;

#define XLATING_MULTI_ID_BYTES 128

/* rank 0: fill id[128] (ncclGetUniqueId).  0 or -EIO. */
int xlating_multi_unique_id(void *id);

/* One process per GPU: this process is `rank` of `world` and drives HIP device `device` (-1: the current one).
 * Engine parameters as xlating_batch_create_grouped().  Collective: returns when every rank has called it.
 * world == 1: `id` NULL -> no communicator, feeds are filtered in place from d_src; `id` given -> a one-rank "loop-back"
 * communicator, feeds take the broadcast path of world > 1 line for line (what the one-GPU tests run).
 * 0, -EINVAL, -ENODEV, -ENOMEM, -EIO (RCCL/HIP failure). */
int xlating_multi_create_rank(int rank, int world, const void *id, uint32_t sampling_freq, int input_format,
                              uint32_t max_input_buffer_length, unsigned max_group_blocks, int device,
                              xlating_multi **multi);

/* One process, `ngpus` GPUs (`devices` = HIP ordinals, or NULL for 0 .. ngpus-1). */
int xlating_multi_create_local(int ngpus, const int *devices, uint32_t sampling_freq, int input_format,
                               uint32_t max_input_buffer_length, unsigned max_group_blocks, xlating_multi **multi);

/* GPUs in the job, and how many of them this process drives (1, or ngpus). */
int xlating_multi_world(const xlating_multi *multi);
int xlating_multi_local(const xlating_multi *multi);

/* Client `global_client` (0, 1, 2, ...) lives on GPU (global_client mod world).  Every process makes the same calls;
 * the call adds the client where this process drives its GPU and is a no-op elsewhere.
 * Returns the client id inside its engine (>= 0) if the GPU is local, -ENOENT if another process owns it, or the
 * error of xlating_batch_add_client(). */
int xlating_multi_add_client(xlating_multi *multi, int global_client, uint32_t decimation, const float *taps,
                             size_t taps_len, int32_t center_freq);
/* The engine of GPU `gpu` (0 .. world-1) if this process drives it, else NULL: fetch outputs, describe, timing. */
xlating_batch *xlating_multi_engine(xlating_multi *multi, int gpu);

/* Feed one super-block: `nblocks` blocks of `input_len` scalar elements each, back to back.
 *   d_src  device memory on GPU 0 holding the blocks (the process that drives GPU 0 passes it; the others pass NULL).
 * The blocks are broadcast from GPU 0 into the next receive buffer of every GPU on the communication stream, and every
 * local engine then processes them there (xlating_batch_process_device_group) on its compute stream.  Asynchronous:
 * returns once the work is enqueued; d_src is still being read then -- by the root's broadcast on the communication
 * stream, or (world == 1 without a communicator: nothing is broadcast) by the engine's launches, in place -- and may be
 * overwritten once xlating_multi_feed_done() returns (or _feed_query() says 1, or on a stream that has passed
 * _feed_wait_on_stream(), or after xlating_multi_sync()).
 * 0, -EINVAL, -EIO. */
int xlating_multi_feed(xlating_multi *multi, const void *d_src, size_t input_len, unsigned nblocks, int mode);

/* The source buffer of the LATEST feed: block the calling thread until it is no longer read / poll (1 free, 0 still read)
 * / make `hip_stream` (a hipStream_t of GPU 0's device; NULL = the default stream) wait for that moment, so that an
 * asynchronous refill of d_src enqueued there afterwards is safe.  With a communicator the event sits behind the root's
 * broadcast only -- the filtering of the receive buffers goes on -- so a streaming host refills ONE buffer per feed and
 * the broadcast of super-block k+1 still overlaps the filtering of k.  WITHOUT a communicator (one GPU, the blocks are filtered
 * in place) "no longer read" means "the engine's launches of that feed have run": stricter than behind a broadcast, since the
 * launches themselves read d_src.  Processes that do not drive GPU 0 hold no source: 0 / 1 / 0 at once.
 * Threading: a host object is driven by ONE thread -- these three calls read the state xlating_multi_feed() writes (which stream
 * the latest feed used) without a lock, so they belong to the feeding thread, like every other call on the object.
 * 0 (query: 0 or 1), -EINVAL, -EIO. */
int xlating_multi_feed_done(xlating_multi *multi);
int xlating_multi_feed_query(xlating_multi *multi);
int xlating_multi_feed_wait_on_stream(xlating_multi *multi, void *hip_stream);

/* Feed timing (off by default; only with a communicator): every broadcast is bracketed by two HIP events on the
 * communication stream of the first local GPU.  _read -- call it after xlating_multi_sync: it touches nothing on the data path --
 * looks at the latest feeds (at most 14: the events live in a ring) and returns how many it measured and their totals: the
 * broadcasts' duration and the part of it that elapsed while the PREVIOUS feed's filtering was still running on the compute
 * stream (hidden / bcast = the fraction of the exchange step hidden behind the independent per-GPU work).  0 / n, -EINVAL. */
int xlating_multi_feed_timing(xlating_multi *multi, int enable);
int xlating_multi_feed_timing_read(xlating_multi *multi, double *bcast_ms_total, double *hidden_ms_total, int reset);
/* Ranks of the RCCL communicator this host broadcasts over (ncclCommCount): `world` when it exists, 0 without one. */
int xlating_multi_comm_count(const xlating_multi *multi);

/* Wait until everything fed so far has been filtered on the local GPUs. */
int xlating_multi_sync(xlating_multi *multi);

void xlating_multi_destroy(xlating_multi *multi);

#ifdef __cplusplus
}
#endif
#endif /* SDR_SERVER_AMD_XLATING_MULTI_H_ */

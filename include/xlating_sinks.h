/*
 * include/xlating_sinks.h -- per-client output sinks for the batched path.
 *
 * In the reference every client's dsp thread delivers its own filter output right after process_*():
 * write_to_socket() loops write() until the block is out, write_to_file() does one fwrite()/gzwrite() into
 * <base_path>/<id>.cf32[.gz]; a failed or short write ends the client (src/dsp_worker.c:10-39, 74-86, 126-144).
 * With the batched engine (xlating_batch.h) one call produces the outputs of ALL clients at once, so delivery
 * moves to a small pool of writer threads fed through bounded per-client queues:
 *
 *     xlating_batch_process_host(engine, ...); xlating_batch_fetch(engine);
 *     xlating_sinks_submit(sinks, engine);               // copy each attached client's block into its queue
 *     n = xlating_sinks_failed(sinks, ids, cap);         // clients to drop (the reference closes their socket)
 *
 * The bytes a sink emits are exactly the client's cf32 output stream (interleaved re, im float32, native
 * endianness) -- what fwrite(filter_output, sizeof(float complex), len, file) produces; the gzip sink's stream
 * decompresses to the same bytes.  Back-pressure rule: a client whose queue would overflow (its peer does not keep
 * up) or whose write fails is marked failed and dropped, like the reference's "if disk is full, then terminate the
 * client".  Host-only code: no GPU is needed to create or use sinks.
 */
#ifndef SDR_SERVER_AMD_XLATING_SINKS_H_
#define SDR_SERVER_AMD_XLATING_SINKS_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct xlating_sinks_t xlating_sinks;
struct xlating_batch_t;

/* writer_threads: threads that drain the queues (a client is served by thread id % writer_threads, so a client's
 * bytes stay in order); queue_bytes: capacity of each client's queue (e.g. 16 blocks x 25 KB).
 * Returns 0, -EINVAL, -ENOMEM. */
int xlating_sinks_create(unsigned writer_threads, size_t queue_bytes, xlating_sinks **sinks);

/* Deliver client_id's stream to an open descriptor (socket, pipe, file), write_to_socket() semantics: every byte is
 * written, EINTR retried, any other error fails the sink.  Sockets are written with MSG_NOSIGNAL.  The descriptor
 * stays owned by the caller unless close_on_detach is non-zero.  0, -EINVAL, -EEXIST (already attached), -ENOMEM. */
int xlating_sinks_attach_fd(xlating_sinks *sinks, int client_id, int fd, int close_on_detach);

/* Deliver client_id's stream to <base_path>/<client_id>.cf32 (use_gzip == 0) or <base_path>/<client_id>.cf32.gz
 * (dsp_worker.c:126-144).  0, -EINVAL, -EEXIST, -ENOMEM, or -errno of the failed open. */
int xlating_sinks_attach_file(xlating_sinks *sinks, int client_id, const char *base_path, int use_gzip);

/* Queue n_complex samples (interleaved re, im) for client_id.  The data is copied; the call never blocks on the
 * peer.  0; -ENOENT (no such sink); -EPIPE (the sink has failed, or this block does not fit its queue: the sink
 * is marked failed and will be reported by xlating_sinks_failed). */
int xlating_sinks_write(xlating_sinks *sinks, int client_id, const float *samples, size_t n_complex);

/* After xlating_batch_fetch(): queue the latest output block of every attached client of `engine` (clients that
 * are not alive in the engine are skipped).  Returns the number of blocks queued, or -EINVAL. */
int xlating_sinks_submit(xlating_sinks *sinks, struct xlating_batch_t *engine);

/* Ids of sinks that failed since the last call (each reported once), up to cap; returns how many were stored.
 * A failed sink discards further writes; detach it (and remove the client from the engine). */
size_t xlating_sinks_failed(xlating_sinks *sinks, int *client_ids, size_t cap);

/* Block until everything queued so far has been written (or its sink failed).  A peer that stops reading holds this
 * up until its queue overflows on a later write (descriptor writes are non-blocking and abandoned once the sink has
 * failed).  0 or -EINVAL. */
int xlating_sinks_flush(xlating_sinks *sinks);

/* Flush client_id's queue, close its file (and the descriptor if close_on_detach) and forget it.  0 or -ENOENT. */
int xlating_sinks_detach(xlating_sinks *sinks, int client_id);

/* Totals since creation: bytes handed to the OS / zlib, and blocks dropped because a sink had failed. */
void xlating_sinks_stats(xlating_sinks *sinks, uint64_t *bytes_written, uint64_t *blocks_dropped);

/* Detaches every sink and stops the threads; queued bytes get two seconds to drain, then stuck peers are abandoned. */
void xlating_sinks_destroy(xlating_sinks *sinks);

#ifdef __cplusplus
}
#endif
#endif /* SDR_SERVER_AMD_XLATING_SINKS_H_ */

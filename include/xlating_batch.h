/*
 * include/xlating_batch.h -- batched fan-out boundary (extension).
 *
 * The reference fans one IQ block out to N clients with N memcpy's and N per-thread process_* calls
 * (src/tcp_server.c:257-271 sdr_callback -> src/dsp_worker.c:202-204 -> src/queue.c:87-119, then
 * dsp_worker.c:57-65 per client).  On a GPU that is N redundant H2D copies and N tiny launches.
 * This API is what sdr_callback would call instead: ONE block in, ONE fused launch for all clients
 * of a device, per-client outputs out.  Per client the arithmetic is exactly that of the
 * single-filter API in xlating.h (same taps, same NCO recurrence, same streaming rules), so each
 * client's output stream equals what process_{native,optimized}_<fmt>_cf32 would have produced for a
 * filter created when the client was added.
 *
 * Everything here is a plain C ABI: pointers, sizes, ints.  Device pointers and the HIP stream are
 * passed as void* so that a torch/RCCL host (bench.py) or a C host can drive it.
 */
#ifndef SDR_SERVER_AMD_XLATING_BATCH_H_
#define SDR_SERVER_AMD_XLATING_BATCH_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct xlating_batch_t xlating_batch;

/* input sample formats == the reference's three device formats (dsp_worker.c:55-67) + the cf32 extension */
enum { XL_FMT_CU8 = 0, XL_FMT_CS8 = 1, XL_FMT_CS16 = 2, XL_FMT_CF32 = 3 };
/* arithmetic variants == the reference's cpu_optimization setting (config.h:20-23, dsp_worker.c:110-124) */
enum { XL_MODE_NATIVE = 0, XL_MODE_OPTIMIZED = 1,
       /* the reference's cs16 output family (process_*_cs16, xlating.c:92-140, 416-447: exact Q15 integer arithmetic, its
        * own never-renormalised int16 phase; optimized == native there).  Outputs are int16 (re, im) pairs:
        * xlating_batch_output_host_cs16().  cu8 / cs8 / cs16 engines only.  The reference server itself only ever calls the
        * cf32 family (dsp_worker.c:110-124); a client stream should stay in one family (the reference keeps separate sample
        * buffers per family behind one history counter, a quirk the single-filter API reproduces and this engine -- one raw
        * history -- does not). */
       XL_MODE_Q15 = 2,
       /* process_optimized_* as the reference's x86 AVX build runs it: the optimized arithmetic, but the NCO phase is
        * NEVER renormalised (xlating.c:338-339; the scalar and NEON paths renormalise once per call, :73, :255, and so do
        * XL_MODE_NATIVE and XL_MODE_OPTIMIZED).  The float32 phase recurrence itself is the same unfused multiply; only
        * the per-call division by |phase| is left out, so the amplitude drifts like the x86 server's (about 1e-3 after 400
        * server-default blocks).  For deployments that must reproduce an x86 OPTIMIZED_CF32 server's long streams; the
        * running phase is shared with the other cf32 modes. */
       XL_MODE_OPTIMIZED_X86 = 3,
       /* the same, for reference builds compiled with FMA enabled (-mfma, -march=native on an AVX2 host): gcc -ffast-math
        * contracts the phase step `phase * phase_incr` to re = fma(pr, ir, -(pi * ii)), im = fma(pr, ii, pi * ir), which
        * walks away from the plain step's rounding by ~1e-5 within a few blocks at some offsets; this mode takes that
        * step (and never renormalises).  Found by matching the phase sequence of the unmodified reference, built both
        * ways, bit for bit (oracle/_ref/libref_avx.so, libref_fast.so; tests/golden/x86_*.npz, fast_*.npz). */
       XL_MODE_OPTIMIZED_X86_FMA = 4 };

/* Create an engine for one input stream on one GPU.
 *   sampling_freq             band sampling rate (server_config->band_sampling_rate)
 *   input_format              XL_FMT_*
 *   max_input_buffer_length   like create_frequency_xlating_filter(): bytes of a cu8 stream, i.e. the
 *                             engine accepts up to max_input_buffer_length/2 complex samples per block
 *   device                    HIP device ordinal, or -1 for the calling thread's current device
 * Returns 0, -ENODEV (no usable HIP device), -EINVAL, -ENOMEM. */
int xlating_batch_create(uint32_t sampling_freq, int input_format, uint32_t max_input_buffer_length, int device,
                         xlating_batch **batch);
/* The same, for an engine whose process calls may cover up to `max_group_blocks` (1..64) consecutive blocks of up to
 * max_input_buffer_length each ("group"; see xlating_batch_process_device_group). */
int xlating_batch_create_grouped(uint32_t sampling_freq, int input_format, uint32_t max_input_buffer_length,
                                 unsigned max_group_blocks, int device, xlating_batch **batch);

/* Plan options (result-neutral: every setting passes the same parity tests).  Six of them; name / value:
 *   "polyphase"         -1 by the size rule (default: classes of >= 32 clients with >= 2 taps per polyphase branch), 0 never, 1
 *                       whenever the shape allows: which classes take the polyphase overlap-save path in XL_MODE_OPTIMIZED
 *   "polyphase_m"       0 by the size rule, 64, 128, 256: its transform length (the rule: 64 points for classes of more than 64 branches with up
 *                       to 8 taps per branch, else 128 for classes of >= 768 clients with <= 32 taps per branch, else 256; a forced
 *                       length a class's filter does not fit falls back to the next one that does)
 *   "mix_kernel"        polyphase classes: the mix launch (spectra x branch spectra, summed over the branches) runs on the matrix
 *                       cores.  1 (default): every float32 operand is carried as two halves (three v_mfma_f32_32x32x16_f16 per 8
 *                       branches, FP32 accumulation; as accurate as the float32 FMA chain) after a power-of-two scale -- per column for
 *                       the branch spectra; for the shared spectra a constant where the input format bounds them (cu8 / cs8 / cs16)
 *                       and, for cf32 streams, one per SEGMENT from the segment's largest spectrum value (a row scale of the per-bin
 *                       product, undone exactly) -- for classes of up to 112 branches (decimation <= 112); larger decimations
 *                       multiply float32 operands (v_mfma_f32_32x32x2_f32: exactly the float32 FMA chain).  3: float32 operands for
 *                       EVERY class -- the all-float32 arithmetic of the path, ~40 % slower in the mix launch.  Same 1e-5 bar
 *   "inverse_kernel"    128-point polyphase classes: the inverse launch's transform -- in the registers of EIGHT lanes per client
 *                       column as 16 x 8 points with one exchange through LDS (5: xl_inv8.hip), in the registers of FOUR lanes per
 *                       column as 32 x 4 points -- whole-line loads, 256-byte store runs (6: xl_inv32.hip) --, or staged in LDS on
 *                       dense XOR-swizzled rows (3).  0 (default): by the size of the launch -- the 8-lane kernel for launches of up
 *                       to 2048 tiles (one block per call: 5-9 % ahead), the 32 x 4 cut from 8193 (8 blocks per call at >= 2048
 *                       clients: 6-10 % ahead of the LDS transform, round 4's pick there), the LDS transform in between (up to
 *                       10 % ahead of both); measured alternating in one process (bench.py "inverse launch A/B")
 *   "nco_side_stream"   -1 by rule (default: calls of >= 2 blocks whose launches are polyphase or light, and one-block polyphase
 *                       calls of up to 2048 clients), 0 never, 1 always: the NCO phase recurrence of the following calls runs as a
 *                       kernel of its own on a side stream (on CUs reserved for it when the call uses XL_STREAM_ENGINE) instead of
 *                       riding inside the call's launches
 *   "expected_clients"  0 (default) .. 8192: the CUs of that side kernel are reserved for this many clients from the first plan on
 *                       (64 clients per CU up to 2048 clients; none from there to 3008 -- the side kernel then shares the chip --; beyond,
 *                       it runs in rounds on half, a third, ... as many CUs), not for the clients joined so far -- the reservation then never grows while clients
 *                       join up to that number (growing it re-creates two streams: ~25 ms, once per 512 clients); until then the
 *                       launches run on correspondingly fewer CUs
 * Returns 0, -ENOENT (unknown name), -EINVAL.  The plan is rebuilt at the next call.
 * (Launch-shaping knobs of the tuning sessions -- tile heights, riders, slices, passes per workgroup, calls per chain launch, the
 * size rule's client threshold -- are not options: they are read from XL_EXP_* environment variables when an engine is created,
 * csrc/xl_batch.cpp, and ONLY when the process also sets XL_TESTING=1 (tests and tools do) or in -DXL_TUNING builds: a plain process
 * ignores every XL_EXP_* variable and logs one "<4>" line if any is set; rounds 1-4's measured-and-lost kernel variants are gone from
 * the library: tools/experiments/retired/.) */
int xlating_batch_set_option(xlating_batch *batch, const char *name, long value);

/* Add a client whose stream starts with the NEXT block (like dsp_worker_start, dsp_worker.c:90-108).
 * `taps` is the real low-pass prototype (lpf.h); it is COPIED (unlike create_frequency_xlating_filter
 * the caller keeps ownership).  Returns a client id >= 0, or -1 (taps_len == 0), -EINVAL, -ENOMEM. */
int xlating_batch_add_client(xlating_batch *batch, uint32_t decimation, const float *taps, size_t taps_len,
                             int32_t center_freq);

/* Remove a client (dsp_worker_destroy).  0 or -EINVAL. */
int xlating_batch_remove_client(xlating_batch *batch, int client_id);

int xlating_batch_num_clients(const xlating_batch *batch);

/* Process one block for ALL clients.  `input_len` = scalar elements, as in xlating.h.
 * _host:   `input` is host memory; copied H2D on the engine's stream.
 * _device: `d_input` is device memory on the engine's GPU (e.g. the receive buffer of an RCCL
 *          broadcast); read in place, not copied.  `hip_stream` is a hipStream_t (NULL = HIP's legacy
 *          default stream, as everywhere in HIP): the block's work is ordered behind what that stream
 *          holds at the time of the call; the call does not synchronise it.  The caller must keep d_input unmodified until that work has run.
 * Results stay on the device until fetched.  Returns 0, -EINVAL (too long / bad mode), -EIO (HIP error). */
int xlating_batch_process_host(xlating_batch *batch, const void *input, size_t input_len, int mode);
int xlating_batch_process_device(xlating_batch *batch, const void *d_input, size_t input_len, int mode,
                                 void *hip_stream);
/* Consecutive calls may use different streams: a call on another stream than the previous one is ordered behind the
 * previous call's work (the engine's device state -- history, phases -- chains the calls). */

/* Process `nblocks` consecutive blocks of `input_len` scalar elements each, stored back to back, in ONE call: the
 * results are exactly those of nblocks successive process calls (every client's NCO phase is renormalised at each
 * block end like xlating.c:73 does per call; per-client outputs of the blocks are concatenated), from one set of
 * launches.  What sdr_callback would do with a super-block of several device buffers; it is
 * also what makes the launches big enough to fill the chip and streams the per-client filter images once per call
 * instead of once per block.  nblocks <= the engine's max_group_blocks; for nblocks > 1 every block must hold at least
 * one output of every client (input_len / 2 >= the largest decimation).  Same return codes as above. */
int xlating_batch_process_host_group(xlating_batch *batch, const void *input, size_t input_len, unsigned nblocks,
                                     int mode);
int xlating_batch_process_device_group(xlating_batch *batch, const void *d_input, size_t input_len, unsigned nblocks,
                                       int mode, void *hip_stream);
/* hip_stream may be XL_STREAM_ENGINE: the work then runs on a compute stream the engine owns (xlating_batch_sync waits for
 * it).  For calls of several blocks on the polyphase path that stream is CU-masked: it leaves a few CUs (one per 64
 * clients) to the kernel that tabulates the next call's NCO phases on the engine's side stream, which needs whole CUs
 * and otherwise has to wait for a kernel boundary of the caller's stream to find them.
 * _ev: `wait_event` (a hipEvent_t or NULL) is waited for on that stream before the call's work, `record_event` is recorded
 * behind it -- how a host orders its own streams (e.g. an RCCL broadcast into d_input) against the engine's. */
#define XL_STREAM_ENGINE ((void *)(intptr_t)-1)
int xlating_batch_process_device_group_ev(xlating_batch *batch, const void *d_input, size_t input_len, unsigned nblocks,
                                          int mode, void *hip_stream, void *wait_event, void *record_event);

/* Number of complex output samples client produced in the last processed call (host-side integer state), and in
 * block `block` of that call (the client's output row holds the blocks' outputs back to back). */
size_t xlating_batch_output_len(const xlating_batch *batch, int client_id);
size_t xlating_batch_output_len_block(const xlating_batch *batch, int client_id, unsigned block);

/* Copy every client's last-block output D2H into engine-owned pinned memory (one copy) and wait for it. */
int xlating_batch_fetch(xlating_batch *batch);
/* After xlating_batch_fetch(): pointer to client's samples as interleaved (re,im) float pairs (-EINVAL after an
 * XL_MODE_Q15 call: use the _cs16 accessor, interleaved (re,im) int16 pairs, and vice versa). */
int xlating_batch_output_host(xlating_batch *batch, int client_id, const float **output, size_t *output_len);
int xlating_batch_output_host_cs16(xlating_batch *batch, int client_id, const int16_t **output, size_t *output_len);
/* Device pointer to the client's last-block output (valid until the next process call). */
int xlating_batch_output_device(xlating_batch *batch, int client_id, const void **d_output, size_t *output_len);

/* Current NCO phase of a client (synchronises the engine stream). */
int xlating_batch_client_phase(xlating_batch *batch, int client_id, float *re, float *im);

/* Block until all enqueued work of the engine has finished. */
int xlating_batch_sync(xlating_batch *batch);
/* Without blocking: 1 if everything enqueued by the process calls so far has completed, 0 if not yet; -EINVAL, -EIO.
 * xlating_batch_record_event: record `hip_event` (a hipEvent_t of the engine's device) behind the latest call's launches, on the
 * stream(s) they were enqueued on -- what a caller needs to make another stream wait for the latest call without a per-call
 * event (a launch that carries a completion event delays the next launch by ~8 us: a fifth of a one-block call).  0, -EINVAL, -EIO. */
int xlating_batch_query(xlating_batch *batch);
int xlating_batch_record_event(xlating_batch *batch, void *hip_event);

/* Kernel timing with HIP events recorded on the launch stream around the FIR kernel of every block
 * (for bench.py's roofline figures).  enable: 0/1.  _read returns the number of timed launches since
 * the last reset and their summed duration in milliseconds; it synchronises the stream. */
int xlating_batch_timing(xlating_batch *batch, int enable);  /* 2: additionally time the three polyphase launches */
int xlating_batch_timing_read(xlating_batch *batch, double *fir_ms_total, double *nco_ms_total, int reset);

/* Bracket only every n-th block with events (default 1 = every block).  An event pair costs a few microseconds of stream
 * time, which is not negligible against a 60 us block. */
int xlating_batch_timing_stride(xlating_batch *batch, unsigned every_n);

/* enable == 2 only: summed durations (ms) of the forward / mix / inverse launches of the polyphase path since the last
 * reset; returns the number of timed blocks. */
int xlating_batch_timing_polyphase(xlating_batch *batch, double ms_total[3], int reset);

/* One-line description of the resident plan, e.g.
 * "clients 1024 classes 1 | direct: h10 x 104 groups | polyphase: cls0 D42 T505 cols1024 V244".  For logs and tests:
 * which arithmetic path the optimized mode takes for which class.  If the client set or an option changed and no call
 * has run since the last plan, the plan is built first; if calls HAVE run, their outputs must stay where they are until
 * the next process call, so the plan they ran with is described, followed by "| re-plan pending ...".
 * Returns the length written (excluding NUL). */
int xlating_batch_describe(xlating_batch *batch, char *buf, size_t buf_len);

void xlating_batch_destroy(xlating_batch *batch);

/* Build/selection information, e.g. "HIP gfx950 (AMD Instinct MI355X), 256 CUs". Never NULL. */
const char *xlating_hip_device_info(void);

#ifdef __cplusplus
}
#endif
#endif /* SDR_SERVER_AMD_XLATING_BATCH_H_ */

/*
 * include/xlating.h -- drop-in boundary for sdr-server's frequency-xlating FIR filter.
 *
 * This header is source- and ABI-compatible with the reference's src/xlating.h:8-38: the same
 * opaque handle, the same 1 create + 12 process + 1 destroy entry points with the same argument
 * meaning, plus the global SIMD_STATUS string (reference src/xlating.c:145,148,156,268, read by
 * src/main.c:10,23 and test/perf_xlating.c:12,15).  src/dsp_worker.c (the only caller, :57-65,
 * :104, :110-124, :195) links against libxlating_hip.so unchanged.
 *
 * Behind it every process_* call runs hand-written HIP kernels on an MI355X (gfx950); there is no
 * CPU arithmetic path.  If no HIP device is usable, create_frequency_xlating_filter() fails with
 * -ENODEV and logs a "<3>" line on stderr (the reference's logging convention).
 *
 * Semantics kept from the reference (file:line are into /root/reference/src/xlating.c):
 *  - create takes OWNERSHIP of `taps` on success and on EVERY failure (-ENOMEM like the reference, and this
 *    library's -ENODEV no usable device / -EINVAL bad shape / -EIO HIP error: the caller, dsp_worker.c:98-107,
 *    assumes the hand-over whenever taps_len != 0); returns -1 for taps_len == 0 WITHOUT consuming taps
 *    (:496-498, :508, :600-602).
 *  - `input_len` counts scalar elements of the input type: bytes for cu8/cs8, int16 values for cs16
 *    (:355, :365, :375; caller at dsp_worker.c:65), i.e. 2 x complex samples.
 *  - `*output` points to filter-owned host memory, valid until the next call on the same filter;
 *    `*output_len` is in complex samples and may be 0 (:81-82, :138-139).
 *  - streaming state (history, NCO phase) persists across calls; the cf32 and cs16 output families
 *    share one history counter but keep separate sample buffers and phases (:29, :76, :133).
 *  - process_native_*_cf32 reproduces the reference's scalar path BIT FOR BIT (sequential float32
 *    tap order, unfused multiply/add, float32 phase recurrence, one hypotf renormalisation per call).
 *    process_optimized_*_cf32 is this platform's fast variant (fused multiply-add inner loop), the
 *    analogue of the reference's AVX/NEON builds; it keeps the per-call phase renormalisation like
 *    the NEON build (:255) and agrees with native to ~1e-7 of full scale.
 *  - the *_cs16 family is exact Q15 integer arithmetic (:92-140); optimized == native (:437-447).
 */
#ifndef SDR_SERVER_AMD_XLATING_H_
#define SDR_SERVER_AMD_XLATING_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
#define XL_CF32 float _Complex
extern "C" {
#else
#include <complex.h>
#define XL_CF32 float complex
#endif

typedef struct xlating_t xlating;

/* "HIP gfx950" -- replaces the reference's "AVX" / "ARM NEON" / "Not detected" strings. */
extern const char *SIMD_STATUS;

/* reference src/xlating.h:10, src/xlating.c:495-582 */
int create_frequency_xlating_filter(uint32_t decimation, float *taps, size_t taps_len, int32_t center_freq,
                                    uint32_t sampling_freq, uint32_t max_input_buffer_length, xlating **filter);

/* reference src/xlating.h:12-22, src/xlating.c:352-414 (cf32 math and output) */
void process_native_cu8_cf32(const uint8_t *input, size_t input_len, XL_CF32 **output, size_t *output_len, xlating *filter);
void process_native_cs8_cf32(const int8_t *input, size_t input_len, XL_CF32 **output, size_t *output_len, xlating *filter);
void process_native_cs16_cf32(const int16_t *input, size_t input_len, XL_CF32 **output, size_t *output_len, xlating *filter);
void process_optimized_cu8_cf32(const uint8_t *input, size_t input_len, XL_CF32 **output, size_t *output_len, xlating *filter);
void process_optimized_cs8_cf32(const int8_t *input, size_t input_len, XL_CF32 **output, size_t *output_len, xlating *filter);
void process_optimized_cs16_cf32(const int16_t *input, size_t input_len, XL_CF32 **output, size_t *output_len, xlating *filter);

/* reference src/xlating.h:26-36, src/xlating.c:416-447 (cs16 / Q15 math and output) */
void process_native_cu8_cs16(const uint8_t *input, size_t input_len, int16_t **output, size_t *output_len, xlating *filter);
void process_native_cs8_cs16(const int8_t *input, size_t input_len, int16_t **output, size_t *output_len, xlating *filter);
void process_native_cs16_cs16(const int16_t *input, size_t input_len, int16_t **output, size_t *output_len, xlating *filter);
void process_optimized_cu8_cs16(const uint8_t *input, size_t input_len, int16_t **output, size_t *output_len, xlating *filter);
void process_optimized_cs8_cs16(const int8_t *input, size_t input_len, int16_t **output, size_t *output_len, xlating *filter);
void process_optimized_cs16_cs16(const int16_t *input, size_t input_len, int16_t **output, size_t *output_len, xlating *filter);

/* reference src/xlating.h:38, src/xlating.c:584-616.  NULL-safe; frees the caller-supplied taps. */
void destroy_xlating(xlating *filter);

/* ---- extension, not in the reference (BASELINE config 5 / SURVEY D4): cf32 input, identity convert.
 * `input_len` = number of floats (2 x complex samples).  Needs max_input_buffer_length/2 >= samples. */
void process_native_cf32_cf32(const float *input, size_t input_len, XL_CF32 **output, size_t *output_len, xlating *filter);
void process_optimized_cf32_cf32(const float *input, size_t input_len, XL_CF32 **output, size_t *output_len, xlating *filter);

/* ---- extension, not in the reference: which build of the reference process_optimized_* follows.
 * The reference's scalar and NEON paths renormalise the NCO phase once per call (src/xlating.c:73, :255); its x86 AVX path
 * never does (:338-339), so an x86 OPTIMIZED_CF32 server's long streams drift in amplitude (about 1e-3 after 400
 * server-default blocks).  on = 0 (default): renormalise, like native.  on = 1: never renormalise -- the x86 build's streams.
 * on = 2: the same for an x86 build compiled with FMA (-mfma / -march=native), whose phase step is contracted to
 * re = fma(pr, ir, -(pi ii)), im = fma(pr, ii, pi ir) and rounds differently.  The environment variable
 * XLATING_OPTIMIZED_X86=<0|1|2> sets the default of filters created afterwards.  0 or -EINVAL. */
int xlating_set_optimized_x86(xlating *filter, int on);

#ifdef __cplusplus
}
#endif
#endif /* SDR_SERVER_AMD_XLATING_H_ */

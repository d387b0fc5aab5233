/*
 * include/xlating_wire.h -- the client wire format and the admission rules in front of the batched path
 * (SURVEY.md section 8(f) rank 4).
 *
 * sdr-server's TCP protocol (src/api.h:4-38) is two packed, big-endian messages:
 *   client -> server   header {u8 protocol_version = 0, u8 type} [+ request {u32 center_freq, u32 sampling_rate,
 *                      u32 band_freq, u8 destination}]  for type REQUEST (0); SHUTDOWN (1) and PING (3) are header only
 *   server -> client   header {0, RESPONSE (2)} + response {u8 status, u32 details}   (details: client id on success,
 *                      failure reason otherwise)
 * and a request is admitted by the rules of src/tcp_server.c:83-141 (integer decimation, non-zero fields, known
 * destination, the client's band inside the server's band -- in the reference's uint32 arithmetic) and :358-367 (all
 * clients of a running device share one band_freq).  An admitted request maps onto the filter exactly as
 * dsp_worker_start does (src/dsp_worker.c:96-104): decimation = band_rate / rate, taps = LPF(1, band_rate, rate/2,
 * rate/lpf_cutoff_rate), centre offset = (int64)center_freq - (int64)band_freq.
 * Host-only code; it does no socket I/O itself (bytes in, bytes out), so any server loop can use it.
 */
#ifndef SDR_SERVER_AMD_XLATING_WIRE_H_
#define SDR_SERVER_AMD_XLATING_WIRE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

struct xlating_batch_t;

enum { XL_WIRE_PROTOCOL_VERSION = 0 };
enum { XL_WIRE_TYPE_REQUEST = 0, XL_WIRE_TYPE_SHUTDOWN = 1, XL_WIRE_TYPE_RESPONSE = 2, XL_WIRE_TYPE_PING = 3 };
enum { XL_WIRE_DESTINATION_FILE = 0, XL_WIRE_DESTINATION_SOCKET = 1 };
enum { XL_WIRE_STATUS_SUCCESS = 0, XL_WIRE_STATUS_FAILURE = 1 };
enum { XL_WIRE_DETAILS_INVALID_REQUEST = 1, XL_WIRE_DETAILS_OUT_OF_BAND_FREQ = 2, XL_WIRE_DETAILS_INTERNAL_ERROR = 3 };
enum { XL_WIRE_HEADER_BYTES = 2, XL_WIRE_REQUEST_BYTES = 13, XL_WIRE_RESPONSE_BYTES = 5 };

typedef struct {
  uint32_t center_freq;   /* Hz */
  uint32_t sampling_rate; /* requested output rate, Hz */
  uint32_t band_freq;     /* centre of the band the device is tuned to, Hz */
  uint8_t destination;    /* XL_WIRE_DESTINATION_* */
} xlating_wire_request;

/* What an admitted request asks of the DSP path. */
typedef struct {
  uint32_t decimation;      /* band_sampling_rate / sampling_rate */
  int32_t center_offset;    /* create_frequency_xlating_filter's center_freq argument */
  uint32_t lpf_cutoff;      /* create_low_pass_filter: cutoff_freq = sampling_rate / 2 */
  uint32_t lpf_transition;  /*                          transition_width = sampling_rate / lpf_cutoff_rate */
} xlating_wire_admission;

/* header: 0 and *type set; -EAGAIN (fewer than 2 bytes); -EPROTO (protocol_version != 0, tcp_server.c:414-417). */
int xlating_wire_parse_header(const uint8_t *buf, size_t len, uint8_t *type);
/* request body (the 13 bytes after the header): 0; -EAGAIN (short, "unable to read request fully"). */
int xlating_wire_parse_request(const uint8_t *buf, size_t len, xlating_wire_request *req);
/* Serialisers; each returns the number of bytes written (2 + 13, 2 + 5, 2). */
size_t xlating_wire_build_request(const xlating_wire_request *req, uint8_t out[15]);
size_t xlating_wire_build_response(uint8_t status, uint32_t details, uint8_t out[7]);
size_t xlating_wire_build_header(uint8_t type, uint8_t out[2]);
/* response message (header + body) from the server: 0; -EAGAIN; -EPROTO (version or type). */
int xlating_wire_parse_response(const uint8_t *buf, size_t len, uint8_t *status, uint32_t *details);

/* Admission (tcp_server.c:100-104, 111-141, 358-367).  current_band_freq: band of the clients already running, 0 if
 * none.  Returns 0 and fills *adm, or -EINVAL with *failure_details = XL_WIRE_DETAILS_INVALID_REQUEST /
 * XL_WIRE_DETAILS_OUT_OF_BAND_FREQ (what the server answers). */
int xlating_wire_admit(const xlating_wire_request *req, uint32_t band_sampling_rate, uint32_t current_band_freq,
                       uint32_t lpf_cutoff_rate, xlating_wire_admission *adm, uint32_t *failure_details);

/* dsp_worker_start for the batched path: designs the client's low-pass (lpf.h) and adds it to `engine`.
 * Returns the engine's client id (>= 0), or a negative errno (the server answers INTERNAL_ERROR). */
int xlating_wire_add_client(struct xlating_batch_t *engine, const xlating_wire_admission *adm, uint32_t band_sampling_rate);

#ifdef __cplusplus
}
#endif
#endif /* SDR_SERVER_AMD_XLATING_WIRE_H_ */

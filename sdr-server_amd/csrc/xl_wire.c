/* xl_wire.c -- client wire format + admission rules (include/xlating_wire.h).
 * Restates the protocol of /root/reference/src/api.h:4-38 and the checks of src/tcp_server.c:83-141, 358-367.
 * Host-only C; no socket I/O. */
#include <errno.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/lpf.h"
#include "../../include/xlating_batch.h"
#include "../../include/xlating_wire.h"

static uint32_t xl_be32(const uint8_t *p) {
  return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | (uint32_t)p[3];
}

static void xl_put_be32(uint8_t *p, uint32_t v) {
  p[0] = (uint8_t)(v >> 24);
  p[1] = (uint8_t)(v >> 16);
  p[2] = (uint8_t)(v >> 8);
  p[3] = (uint8_t)v;
}

int xlating_wire_parse_header(const uint8_t *buf, size_t len, uint8_t *type) {
  if (buf == NULL || type == NULL) return -EINVAL;
  if (len < XL_WIRE_HEADER_BYTES) return -EAGAIN;
  if (buf[0] != XL_WIRE_PROTOCOL_VERSION) return -EPROTO;
  *type = buf[1];
  return 0;
}

int xlating_wire_parse_request(const uint8_t *buf, size_t len, xlating_wire_request *req) {
  if (buf == NULL || req == NULL) return -EINVAL;
  if (len < XL_WIRE_REQUEST_BYTES) return -EAGAIN;
  req->center_freq = xl_be32(buf);       /* ntohl, tcp_server.c:96-98 */
  req->sampling_rate = xl_be32(buf + 4);
  req->band_freq = xl_be32(buf + 8);
  req->destination = buf[12];
  return 0;
}

size_t xlating_wire_build_header(uint8_t type, uint8_t out[2]) {
  out[0] = XL_WIRE_PROTOCOL_VERSION;
  out[1] = type;
  return XL_WIRE_HEADER_BYTES;
}

size_t xlating_wire_build_request(const xlating_wire_request *req, uint8_t out[15]) {
  (void)xlating_wire_build_header(XL_WIRE_TYPE_REQUEST, out);
  xl_put_be32(out + 2, req->center_freq);
  xl_put_be32(out + 6, req->sampling_rate);
  xl_put_be32(out + 10, req->band_freq);
  out[14] = req->destination;
  return XL_WIRE_HEADER_BYTES + XL_WIRE_REQUEST_BYTES;
}

size_t xlating_wire_build_response(uint8_t status, uint32_t details, uint8_t out[7]) {
  (void)xlating_wire_build_header(XL_WIRE_TYPE_RESPONSE, out);
  out[2] = status;
  xl_put_be32(out + 3, details); /* htonl, tcp_server.c:149 */
  return XL_WIRE_HEADER_BYTES + XL_WIRE_RESPONSE_BYTES;
}

int xlating_wire_parse_response(const uint8_t *buf, size_t len, uint8_t *status, uint32_t *details) {
  uint8_t type;
  int rc = xlating_wire_parse_header(buf, len, &type);
  if (rc != 0) return rc;
  if (type != XL_WIRE_TYPE_RESPONSE) return -EPROTO;
  if (len < XL_WIRE_HEADER_BYTES + XL_WIRE_RESPONSE_BYTES) return -EAGAIN;
  if (status) *status = buf[2];
  if (details) *details = xl_be32(buf + 3);
  return 0;
}

int xlating_wire_admit(const xlating_wire_request *req, uint32_t band_sampling_rate, uint32_t current_band_freq,
                       uint32_t lpf_cutoff_rate, xlating_wire_admission *adm, uint32_t *failure_details) {
  uint32_t why = XL_WIRE_DETAILS_INVALID_REQUEST;
  int ok = 0;
  if (req == NULL || adm == NULL || band_sampling_rate == 0 || lpf_cutoff_rate == 0) return -EINVAL;
  do {
    /* tcp_server.c:100-104: the rate must divide the band rate */
    if (req->sampling_rate > 0 && band_sampling_rate % req->sampling_rate != 0) break;
    /* :111-127 */
    if (req->center_freq == 0 || req->sampling_rate == 0 || req->band_freq == 0) break;
    if (req->destination != XL_WIRE_DESTINATION_FILE && req->destination != XL_WIRE_DESTINATION_SOCKET) break;
    /* :128-139, in the reference's unsigned 32-bit arithmetic */
    {
      const uint32_t req_min = req->center_freq - req->sampling_rate / 2;
      const uint32_t srv_min = req->band_freq - band_sampling_rate / 2;
      const uint32_t req_max = req->center_freq + req->sampling_rate / 2;
      const uint32_t srv_max = req->band_freq + band_sampling_rate / 2;
      if (req_min < srv_min || req_max > srv_max) break;
    }
    /* :358-367: one band per running device */
    if (current_band_freq != 0 && current_band_freq != req->band_freq) {
      why = XL_WIRE_DETAILS_OUT_OF_BAND_FREQ;
      break;
    }
    ok = 1;
  } while (0);
  if (!ok) {
    if (failure_details) *failure_details = why;
    return -EINVAL;
  }
  adm->decimation = band_sampling_rate / req->sampling_rate;
  adm->center_offset = (int32_t)((int64_t)req->center_freq - (int64_t)req->band_freq); /* dsp_worker.c:104 */
  adm->lpf_cutoff = req->sampling_rate / 2;                                             /* dsp_worker.c:98 */
  adm->lpf_transition = req->sampling_rate / lpf_cutoff_rate;
  if (failure_details) *failure_details = 0;
  return 0;
}

int xlating_wire_add_client(struct xlating_batch_t *engine, const xlating_wire_admission *adm, uint32_t band_sampling_rate) {
  float *taps = NULL;
  size_t len = 0;
  int id;
  if (engine == NULL || adm == NULL) return -EINVAL;
  if (create_low_pass_filter(1.0F, band_sampling_rate, adm->lpf_cutoff, adm->lpf_transition, &taps, &len) != 0) return -EINVAL;
  id = xlating_batch_add_client(engine, adm->decimation, taps, len, adm->center_offset);
  free(taps); /* the engine copies the prototype */
  return id;
}

/* tests/c/ref_test_helpers.c -- TEST INFRASTRUCTURE.  The four helpers of the reference's test/utils.c that
 * test/test_xlating.c calls, restated (semantics: utils.c:137-145 byte ramp, :157-165 int16 ramp centred on len/2,
 * :176-189 comparisons on (int32_t)(x * 10000) / exact int16), declared by the reference's own test/utils.h.  The
 * reference's utils.c is not compiled because it also pulls in libpng, zlib and the server configuration. */
#include <stdlib.h>
#include <unity.h>

#include "utils.h"

void setup_input_cu8(uint8_t **input, size_t input_offset, size_t len) {
  uint8_t *p = malloc(len ? len : 1);
  TEST_ASSERT_NOT_NULL(p);
  for (size_t k = 0; k < len; ++k) p[k] = (uint8_t)(input_offset + k);
  *input = p;
}

void setup_input_cs16(int16_t **input, size_t input_offset, size_t len) {
  int16_t *p = malloc(sizeof(int16_t) * (len ? len : 1));
  TEST_ASSERT_NOT_NULL(p);
  for (size_t k = 0; k < len; ++k) p[k] = (int16_t)(input_offset + k) - (int16_t)(len / 2);
  *input = p;
}

void assert_cf32(const float expected[], size_t expected_size, float complex *actual, size_t actual_size) {
  TEST_ASSERT_EQUAL_INT(expected_size, actual_size);
  for (size_t k = 0; k < expected_size; ++k) {
    TEST_ASSERT_EQUAL_INT((int32_t)(expected[2 * k] * 10000), (int32_t)(crealf(actual[k]) * 10000));
    TEST_ASSERT_EQUAL_INT((int32_t)(expected[2 * k + 1] * 10000), (int32_t)(cimagf(actual[k]) * 10000));
  }
}

void assert_cs16(const int16_t expected[], size_t expected_size, int16_t *actual, size_t actual_size) {
  TEST_ASSERT_EQUAL_INT(expected_size, actual_size);
  for (size_t k = 0; k < 2 * expected_size; ++k) TEST_ASSERT_EQUAL_INT(expected[k], actual[k]);
}

/*
 * njode_selftest.h -- test hooks of libnjode_hip.so (C ABI).  Not needed to use the library:
 * they expose device-side building blocks whose behaviour is pinned by tests
 * (tests/test_dropout_stream.py).
 */
#ifndef NJODE_SELFTEST_H
#define NJODE_SELFTEST_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/*
 * The dropout keep-bit stream of the matrix-core kernels (njode_amd/csrc/njode_device.h:
 * drop_state; njode_mfma.h: keep_bits): for lane group g = 0..3 the first `n_words`
 * xorshift32 words of the stream keyed by (seed, global path id, time key, network id, g),
 * written to out_words[g * n_words + i] (device pointer, uint32).  The reference has no such
 * function (torch's nn.Dropout draws the masks, models.py:148-160); the stream is restated in
 * oracle/dropout_oracle.py and compared word for word.
 */
int njode_selftest_dropout_words(uint64_t seed, uint64_t path_id, uint32_t time_key, uint32_t net,
                                 int32_t n_words, uint32_t* out_words, void* stream);

/*
 * Maintainer aid of the one-launch plan (njode_amd/csrc/njode_plan.h), active with NJODE_PLAN_STAMPS=1
 * in the environment: the 100 MHz wall clock every plan block wrote at its stage ends during the LAST
 * plan launch on the current device.  out[block * 8 + slot]: slot 6 = entry, slots 0..5 = end of stage
 * 0..5.  Synchronises the device.  Returns the number of blocks copied (0: none / switched off).
 */
int njode_debug_plan_stamps(unsigned long long* out, int cap_blocks);

#ifdef __cplusplus
}
#endif
#endif

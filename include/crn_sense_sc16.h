/*
 * crn_sense_sc16.h — OPTIONAL extension of the C ABI in crn_sense.h: wire-format input.
 *
 * NOT part of libcrnsense.so.  `make -C cognitive-radio-network_amd/csrc SC16=1` builds libcrnsense_sc16.so = everything libcrnsense.so
 * exports + the wire-format kernels (csrc/crn_kernels_sc16.hip) + the five entry points below; a program that uses them includes this
 * header and links that library instead.  (BASELINE.json's north star reads complex floats — "coalesced complex-float HBM loads" —
 * so the product does; this halves the bytes for callers that hold the radio's int16 samples.)
 */
#ifndef CRN_SENSE_SC16_H
#define CRN_SENSE_SC16_H

#include "crn_sense.h"

#ifdef __cplusplus
extern "C" {
#endif

/* crn_sense_run_device on samples kept in the radio's WIRE FORMAT: two int16 per complex sample (re, im; full scale 32768), 4 bytes
 * instead of 8 — what the reference's USRPs put on the network (src/extensible_cognitive_radio.cpp:1263-1265: 363-364 samples per
 * 1500-byte packet) and UHD's recv converts to the complex floats of ce_usrp_rx_buffer (:1071-1072).  The kernel converts in its
 * first pass (int16 / 32768, exact in fp32), so every output is bit-identical to crn_sense_run_device on the converted floats,
 * while HBM holds and streams half the bytes.  Same arguments otherwise (strides in samples; d_iq 4-byte aligned); every configuration
 * the float entry point takes. */
CRN_API int crn_sense_run_device_sc16(crn_handle *h, const int16_t *d_iq, int64_t n_epochs, int32_t samples_per_frame,
                                      int64_t epoch_stride, const crn_out *d_out, void *stream);
/* The constant of the int16 -> float conversion that wire-format launches stand for: a sample is k / full_scale.  32768 (the
 * default) is a power of two, applied once per epoch and exact — that is what makes the outputs bit-identical to the float path.
 * A converter with another constant (UHD's sc16 -> fc32 scales by 1/32767) is matched by naming it here: magnitudes then carry
 * 1/full_scale and energies its square, and the outputs equal the float path's on floats converted with that constant to within
 * rounding (~1e-7 relative) instead of bit for bit. */
CRN_API int crn_sense_set_wire_full_scale(crn_handle *h, double full_scale);
/* Complex floats -> wire format on the device: round(x * full_scale) (the handle's: 32768 unless crn_sense_set_wire_full_scale
 * changed it), clipped to int16; n_samples complex samples. */
CRN_API int crn_pack_sc16_device(crn_handle *h, const float *d_iq, int64_t n_samples, int16_t *d_out, void *stream);

/* The same ring for packets in the radio's wire format (int16 pairs, 4 bytes per complex sample: crn_sense_run_device_sc16): half
 * the bytes through the pushing thread's copy, the bus and HBM.  Packets go in through crn_ingest_push_sc16 (CRN_ERR_STATE for the
 * other kind of push); everything else is shared. */
CRN_API int crn_ingest_create_sc16(crn_handle *h, int32_t n_streams, int32_t samples_per_packet, int32_t epochs_per_batch,
                                   crn_ingest **out);
CRN_API int crn_ingest_push_sc16(crn_ingest *g, int32_t stream, const int16_t *iq_packet);

#ifdef __cplusplus
}
#endif
#endif /* CRN_SENSE_SC16_H */

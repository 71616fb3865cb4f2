/*
 * crn_sense.h — C ABI of libcrnsense: the MI355X (gfx950) spectrum-sensing hot path.
 *
 * This is the drop-in boundary for the FFT + energy-detect + decision loop that the
 * reference engine runs on every USRP rx buffer
 *   (reference: cognitive_engines/CE_Predictive_Node/CE_Predictive_Node.cpp:146-289).
 * The reference has no FFI of its own: the path is C++ calling liquid-dsp's C API
 *   fft_create_plan / fft_execute        (CE_Predictive_Node.cpp:42-45, :150)
 * followed by inline loops.  Every entry point below names the reference lines it replaces.
 *
 * Conventions
 *   - plain C, no C++/torch types; all pointers are raw host or device addresses.
 *   - every function returns 0 on success or a negative crn_status; crn_last_error()
 *     returns a thread-local human readable message (the reference itself has no error
 *     convention: void returns, printf + exit — src/crts.cpp:111-115).
 *   - IQ samples are interleaved complex fp32 (re, im), 8 bytes per sample — the layout of
 *     ExtensibleCognitiveRadio::ce_usrp_rx_buffer (include/extensible_cognitive_radio.hpp:547)
 *     and of liquid_float_complex.
 *   - a "frame" is one FFT input (N samples, of which the first L <= N come from the
 *     caller and the rest are zero: CE_Predictive_Node.cpp:37,149);
 *     an "epoch" is K consecutive frames that yield one decision (fft_averaging, .hpp:32).
 *   - thread-compatible: one handle per host thread.  The one concurrency a handle is built for: an ingest ring's launcher thread
 *     launching through it while the thread that owns it changes thresholds, weights or the band plan (crn_sense_set_thresholds /
 *     _set_ann / _set_bands) — launches and those updates are serialised inside the handle, each launch sees one plan, whole.
 *   - every entry point that touches the GPU makes cfg.device current on the calling thread first
 *     (a new thread starts on device 0), so handles of several devices can be driven from any thread.
 */
#ifndef CRN_SENSE_H
#define CRN_SENSE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CRN_ABI_VERSION 4

#if defined(__GNUC__)
#define CRN_API __attribute__((visibility("default")))
#else
#define CRN_API
#endif

#define CRN_MAX_BANDS 80   /* features per epoch (reference: 4 = NF, CH1, CH2, CH3)     */
#define CRN_MAX_SEGS 160   /* contiguous bin ranges (reference: 5, CH1 wraps around DC) */
#define CRN_ANN_IN 4       /* INPUTS          CE_Predictive_Node.hpp:20 */
#define CRN_ANN_HID 5      /* HIDDEN_NEURONS  CE_Predictive_Node.hpp:21 */
#define CRN_ANN_OUT 3      /* OUTPUT_NEURONS  CE_Predictive_Node.hpp:22 */

typedef enum crn_status {
  CRN_OK = 0,
  CRN_ERR_ARG = -1,      /* bad argument / unsupported configuration   */
  CRN_ERR_DEVICE = -2,   /* HIP runtime error (message has the detail) */
  CRN_ERR_NOMEM = -3,
  CRN_ERR_STATE = -4,    /* call sequence error                        */
  CRN_ERR_BUSY = -5      /* ingest ring: both batch buffers are on the GPU; the packet was NOT taken */
} crn_status;

/* How per-bin values are accumulated over the K frames of an epoch and turned into features. */
typedef enum crn_mode {
  /* Reference-exact: a[k] += |X[k]| / K per frame (CE_Predictive_Node.cpp:152-154);
   * M_b = sum_{k in band b} a[k] (:173-191); feature_b = M_b * M_b (:194-197). */
  CRN_MODE_REF_MAG = 0,
  /* Energy detector (BASELINE.json configs[1..4]): P[k] = (sum_f |X_f[k]|^2) / K;
   * feature_b = sum_{k in band b} P[k]. */
  CRN_MODE_ENERGY = 1
} crn_mode;

/* How features become the occupied/idle decision. */
typedef enum crn_decide {
  /* 4-5-3 sigmoid net in double + first-output->=0.8 cascade (CE_Predictive_Node.cpp:214-261).
   * Requires n_bands == 4 with band order {NF, CH1, CH2, CH3} (.cpp:200). */
  CRN_DECIDE_ANN = 0,
  /* occupancy[b] = feature_b > thresh[b] * (ref_band >= 0 ? feature_ref : 1), fp32. */
  CRN_DECIDE_THRESHOLD = 1,
  CRN_DECIDE_NONE = 2
} crn_decide;

typedef enum crn_window {
  CRN_WINDOW_RECT = 0,   /* the reference applies no window (CE_Predictive_Node.cpp:149-150) */
  CRN_WINDOW_HANN = 1,   /* periodic Hann, w[n] = 0.5 - 0.5 cos(2 pi n / N) (Welch mode)     */
  /* 4-term Blackman-Harris over N-1, a = {0.35875, 0.48829, 0.14128, 0.01168}: the window of the
   * reference's GNU Radio monitor (spectrum_analyzer.py:262-275, firdes.WIN_BLACKMAN_hARRIS). */
  CRN_WINDOW_BLACKMAN_HARRIS = 2
} crn_window;

/* Bins [lo, hi) of the N-point spectrum contribute to feature `band`. */
typedef struct crn_band_seg {
  int32_t lo;
  int32_t hi;
  int32_t band;
} crn_band_seg;

/* Everything the reference hard-codes as `static constexpr` members
 * (CE_Predictive_Node.hpp:30-33,42-43,55-57) or literals (.cpp:78-120,173-191,245-261). */
typedef struct crn_cfg {
  int32_t abi_version;        /* must be CRN_ABI_VERSION                                   */
  int32_t fft_len;            /* N in {512, 1024, 2048, 4096}; reference: 512 (.hpp:31)    */
  int32_t frames_per_epoch;   /* K >= 1; reference: 10 (.hpp:32)                           */
  int32_t hop;                /* samples between frame starts. hop == fft_len: disjoint
                                 frames (reference). hop == fft_len/2: Welch, 50 % overlap */
  int32_t mode;               /* crn_mode                                                  */
  int32_t decide;             /* crn_decide                                                */
  int32_t window;             /* crn_window                                                */
  int32_t n_bands;            /* 1..CRN_MAX_BANDS                                          */
  int32_t n_segs;             /* 1..CRN_MAX_SEGS                                           */
  int32_t ref_band;           /* THRESHOLD: band whose feature scales the thresholds, or -1 */
  int32_t device;             /* HIP device ordinal                                        */
  int32_t reserved0;
  crn_band_seg segs[CRN_MAX_SEGS];
  float thresh[CRN_MAX_BANDS];
  /* ANN weights, reference indexing: w_ih[i][j], i = 0 bias, 1..4 inputs {NF,CH1,CH2,CH3},
   * j = 1..5 hidden (column 0 unused); w_ho[j][k], j = 0 bias, 1..5 hidden, k = 1..3 outputs
   * (CE_Predictive_Node.hpp:66,71; values .cpp:78-120). */
  double ann_w_ih[CRN_ANN_IN + 1][CRN_ANN_HID + 1];
  double ann_w_ho[CRN_ANN_HID + 1][CRN_ANN_OUT + 1];
  double ann_threshold;       /* 0.8 (.cpp:245,250,255)                                    */
  /* Transmit frequency the engine tunes to for decision d = 0 (none), 1, 2, 3
   * (.cpp:247,252,257: CH1 busy -> CHANNEL2, CH2 busy -> CHANNEL1, CH3 busy -> CHANNEL2;
   * entry 0 is unused: "ALL BUSY" makes no call, .cpp:260-261). */
  double tx_freq_for_decision[4];
} crn_cfg;

typedef struct crn_handle crn_handle;

/* Per-epoch results.  Any pointer may be NULL (that output is skipped).  For the *_device
 * entry point these are device addresses, for the *_host entry point host addresses. */
typedef struct crn_out {
  float *features;      /* [n_epochs][n_bands] fp32. REF_MAG order {NF,CH1,CH2,CH3}: the
                           Features_Buffer[1..4] of CE_Predictive_Node.cpp:200               */
  double *ann_out;      /* [n_epochs][3] Output[1..3] (.cpp:229-235); DECIDE_ANN only        */
  int32_t *decision;    /* [n_epochs] DECIDE_ANN: 0 = "ALL BUSY" (no output >= 0.8), 1/2/3 =
                           Channel_State[d] OCCUPIED (.cpp:245-261). DECIDE_THRESHOLD: number
                           of occupied bands.                                               */
  uint8_t *occupancy;   /* [n_epochs][n_bands] 1 = occupied. DECIDE_ANN: one-hot over
                           bands 1..3 from `decision`; band 0 (NF) always 0.                */
  float *spectrum;      /* [n_epochs][fft_len] the K-frame average per bin: fft_avg[] of
                           CE_Predictive_Node.hpp:51 (REF_MAG) or P[k] (ENERGY).             */
} crn_out;

/* -- configuration helpers ------------------------------------------------------------ */

/* The reference engine's own parameters: N=512, K=10, rectangular, REF_MAG, the five bin
 * ranges of CE_Predictive_Node.cpp:173-191 (bin 511 is NOT part of CH1), DECIDE_ANN with the
 * weights of .cpp:78-120, threshold 0.8, tx map 835e6/833e6/835e6 (.hpp:55-57, .cpp:245-258). */
CRN_API int crn_cfg_reference(crn_cfg *cfg);

/* The reference engine's estimator at another FFT size: crn_cfg_reference with fft_len N (multiple of 512) and the five bin
 * ranges scaled by N/512 (the same frequency spans at fs = 13 MHz); REF_MAG + DECIDE_ANN.  The weights stay the reference's,
 * which were fitted at N = 512 and one receiver gain (features grow with N^2): fit others with crn_ann_train_device and put
 * them in with crn_cfg_load_ann / crn_sense_set_ann. */
CRN_API int crn_cfg_reference_scaled(crn_cfg *cfg, int32_t fft_len);

/* The 4-5-3 network's weights as a text file — what crn_ann_train_device produces, for an engine's `-w <file>` (ce_args:
 * src/crts.cpp:43-81): '#' comments, then WeightIH[5][6] and WeightHO[6][4] in the reference's array order
 * (CE_Predictive_Node.hpp:66,71; .cpp:78-120), then optionally the decision threshold.  Doubles are written with 17 digits
 * (exact round trip).  load leaves everything else in cfg untouched. */
CRN_API int crn_cfg_save_ann(const crn_cfg *cfg, const char *path);
CRN_API int crn_cfg_load_ann(crn_cfg *cfg, const char *path);

/* Build-side generalisation used by BASELINE.json configs[1] and the headline metric:
 * fft_len N (multiple of 512), K=10, rectangular, ENERGY, the reference's bin ranges scaled by
 * N/512 (same frequency spans at fs = 13 MHz), DECIDE_THRESHOLD relative to the noise-floor band
 * with thresh[b] = lambda * bins_b / bins_NF. */
CRN_API int crn_cfg_energy_scaled(crn_cfg *cfg, int32_t fft_len, float lambda);

/* Welch PSD configuration of BASELINE.json configs[2]: N-point Hann, hop N/2, n_bands equal
 * contiguous bands covering all N bins, absolute thresholds (ref_band = -1) to be filled in by
 * the caller in cfg->thresh. */
CRN_API int crn_cfg_welch(crn_cfg *cfg, int32_t fft_len, int32_t frames_per_epoch, int32_t n_bands);

/* crn_cfg_energy_scaled's channel plan and relative thresholds on a Welch estimate: periodic Hann, hop N/2, K frames. */
CRN_API int crn_cfg_welch_scaled(crn_cfg *cfg, int32_t fft_len, int32_t frames_per_epoch, float lambda);

/* -- lifecycle --------------------------------------------------------------------------- */

/* Replaces the constructor's buffer zeroing + fft_create_plan (CE_Predictive_Node.cpp:36-45):
 * validates cfg, selects cfg->device, builds twiddle/window/band tables in HBM. */
CRN_API int crn_sense_create(const crn_cfg *cfg, crn_handle **out);

/* The reference never destroys its plan (empty destructor, CE_Predictive_Node.cpp:49; no
 * `delete CE` in the ECR); a GPU-backed engine needs an explicit release. */
/* CRN_ERR_STATE while an ingest ring is attached (its launcher thread launches through the handle): crn_ingest_destroy first. */
CRN_API int crn_sense_destroy(crn_handle *h);

/* -- the hot path --------------------------------------------------------------------------
 * Replaces, for n_epochs * K frames in one launch, CE_Predictive_Node.cpp:148-261:
 *   memcpy into the zero-padded FFT buffer (:149), fft_execute (:150), magnitude + /K running
 *   mean (:152-154), band sums + squares (:173-197), feature widening (:200), ANN (:214-235),
 *   decision cascade (:245-261) and the fft_avg reset (:287-288).
 *
 * d_iq: device pointer, interleaved complex fp32.  Epoch e starts at sample
 *       e * epoch_stride; its frame f covers samples [f*hop, f*hop + samples_per_frame) of the
 *       epoch, zero-padded to fft_len.  samples_per_frame (L) must be in 1..fft_len — the
 *       reference does not check (:149) and overruns its buffer when L > 512; this ABI rejects
 *       it.  With hop < fft_len, samples_per_frame must equal fft_len.
 *       epoch_stride <= 0 selects the dense default K * samples_per_frame (hop == fft_len)
 *       or K * hop (overlapped: consecutive epochs share fft_len - hop samples, so the
 *       buffer must hold n_epochs * K * hop + (fft_len - hop) samples).
 * stream: a hipStream_t (NULL = default stream).  The call only enqueues work.
 */
CRN_API int crn_sense_run_device(crn_handle *h, const float *d_iq, int64_t n_epochs,
                         int32_t samples_per_frame, int64_t epoch_stride,
                         const crn_out *d_out, void *stream);

/* Host-buffer convenience used by the engine wrapper: H2D, run, D2H, synchronise. */
CRN_API int crn_sense_run_host(crn_handle *h, const float *iq, int64_t n_epochs,
                       int32_t samples_per_frame, int64_t epoch_stride, const crn_out *out);

/* Allocate, now, the device scratch and pinned staging that crn_sense_run_host needs for up to
 * max_epochs dense epochs of full-length frames (and their per-bin spectra when want_spectrum != 0),
 * and load the kernels: a later crn_sense_run_host within that size allocates nothing.  An engine
 * calls this from its constructor so that no allocation ever runs inside execute(), where CE_mutex
 * is held (src/extensible_cognitive_radio.cpp:1792-1803). */
CRN_API int crn_sense_reserve_host(crn_handle *h, int64_t max_epochs, int32_t want_spectrum);

/* -- ingest ring: the rx-worker side of the boundary --------------------------------------------
 * The ECR's rx worker hands the engine one USRP packet at a time (memcpy into ce_usrp_rx_buffer +
 * condition signal, src/extensible_cognitive_radio.cpp:1310-1324) and the engine must return
 * quickly because CE_mutex is held (:1792-1803).  The ring coalesces packets of many streams into
 * K-packet epochs — each packet is copied once, straight into its place in a pinned batch buffer —
 * ships batches to the GPU asynchronously on a private stream (H2D -> kernel -> D2H, two buffers)
 * and hands decisions back through a non-blocking poll.  crn_ingest_push and crn_ingest_poll never
 * wait for the GPU and never allocate: `execute()` only copies one packet and checks an event.
 * The ring owns a launcher thread that makes every HIP call (it calls crn_sense_run_device on the
 * handle: do not use crn_sense_run_host / crn_sense_reserve_host on that handle from another
 * thread while batches are in flight; crn_sense_run_device is safe to call concurrently).
 * Environment, read at creation: $CRN_INGEST_ZEROCOPY_BYTES (524288: batches up to this size are read from and written to the
 * pinned buffers by the kernel itself, no copies), $CRN_INGEST_SPIN_US (150: how long after a small batch's hand-off the launcher
 * polls its event before sleeping), $CRN_INGEST_PREWAKE_US (600; 0 = off: ten packets before a small batch's hand-off the pushing
 * thread tells the launcher — one condition-variable signal, no wait — and the launcher polls for the work for at most this long
 * instead of sleeping: a sleeping thread comes back tens of microseconds late; when the ring's stream has sat idle for more than
 * 2 ms the launcher also queues one empty launch at that moment — the HIP calls of the first launch on an idle queue take several
 * times longer than back to back, and this way the empty one pays for it; $CRN_INGEST_WARM_GPU=0 leaves it out), $CRN_INGEST_TRACE (1: crn_ingest_destroy prints to
 * stderr where the hand-off-to-results time went: the launcher's wake-up, the HIP calls, device time + noticing the event). */
typedef struct crn_ingest crn_ingest;

typedef struct crn_epoch_result {
  int32_t stream;                    /* as given to crn_ingest_push                         */
  int32_t decision;                  /* crn_out.decision                                    */
  int64_t epoch_seq;                 /* 0, 1, 2, .. per stream                              */
  double ann_out[3];                 /* DECIDE_ANN only                                     */
  float features[CRN_MAX_BANDS];
  uint8_t occupancy[CRN_MAX_BANDS];
  int32_t flags;                     /* CRN_EPOCH_* bits                                    */
  float noise_floor;                 /* the estimate the epoch's thresholds came from (crn_ingest_calibrate); 0 without one */
} crn_epoch_result;
/* The epoch was launched while a calibration was collecting, or before the thresholds it produced were in place: its features are
 * measurements, its decision / occupancy were taken against the thresholds of before — not to be acted on. */
#define CRN_EPOCH_CALIBRATION 1

/* samples_per_packet: L of every pushed packet (1..fft_len) — also the largest
 * L crn_ingest_set_packet_len may select later (buffers are sized for it here, once).
 * Disjoint frames (hop == fft_len, the reference's shape): a packet is a frame, zero-padded to fft_len by the kernel; K packets
 * make an epoch.  Overlapped frames (hop < fft_len, the Welch plans): the packets of an epoch are consecutive pieces of ONE
 * contiguous run of samples, ceil(((K - 1) hop + fft_len) / L) of them (crn_ingest_packets_per_epoch), from which the kernel
 * cuts its K overlapped frames.
 * epochs_per_batch: completed epochs that trigger a launch.  All device, pinned and host memory the
 * ring will ever use is allocated by this call. */
CRN_API int crn_ingest_create(crn_handle *h, int32_t n_streams, int32_t samples_per_packet,
                              int32_t epochs_per_batch, crn_ingest **out);
/* Threshold plans whose thresholds are lambda x a measured noise floor (the scan engine's start-up): the next
 * n_epochs epochs that come back only feed the estimate — crn_noise_floor_device's median of medians over their features — and with
 * the last of them the ring's launcher thread uploads them, reduces them and puts lambda x the estimate in as every band's
 * threshold, ordered on the ring's stream.  Every epoch launched before that update comes out of crn_ingest_poll marked
 * CRN_EPOCH_CALIBRATION; the first unmarked one carries the estimate in `noise_floor`.  The call itself only posts the request
 * (no HIP call, no allocation: the buffers were made by crn_ingest_create), so an engine may issue it from execute();
 * CRN_ERR_STATE while an earlier request is still collecting, or on a handle that does not decide by thresholds. */
CRN_API int crn_ingest_calibrate(crn_ingest *g, int32_t n_epochs, float lambda);
/* The estimate in force (0 before the first calibration has finished) and whether one is collecting now.  Never blocks. */
CRN_API int crn_ingest_noise_floor(crn_ingest *g, float *nf_out, int32_t *calibrating);
/* Packets of a different length (<= the creation length) from now on: the rx worker learns the UHD
 * packet size only when it starts (src/extensible_cognitive_radio.cpp:1263-1265), after the engine
 * was constructed.  CRN_ERR_STATE while epochs are staged. */
CRN_API int crn_ingest_set_packet_len(crn_ingest *g, int32_t samples_per_packet);
/* Copy one packet of stream `stream` (L interleaved complex samples) into its epoch's pinned slot;
 * the K-th packet of the last missing epoch of a batch also enqueues the batch.  Never blocks: when
 * both batch buffers are on the GPU the packet is refused with CRN_ERR_BUSY (counted by
 * crn_ingest_dropped) — the caller may drop it, as the reference's hand-off does with frames the CE
 * thread is not ready for, or call crn_ingest_wait and push again. */
CRN_API int crn_ingest_push(crn_ingest *g, int32_t stream, const float *iq_packet);
/* Launch the completed epochs staged so far, even if fewer than epochs_per_batch (may wait for the
 * older batch when open epochs have to move to the other buffer). */
CRN_API int crn_ingest_flush(crn_ingest *g);
/* Block until the buffer being filled is free again (back-pressure for callers that must not drop). */
CRN_API int crn_ingest_wait(crn_ingest *g);
/* Non-blocking: results of finished batches, oldest first.  *n_out <= max_results. */
CRN_API int crn_ingest_poll(crn_ingest *g, crn_epoch_result *out, int32_t max_results, int32_t *n_out);
/* flush + wait for everything in flight (results stay queued for crn_ingest_poll). */
CRN_API int crn_ingest_drain(crn_ingest *g);
/* Packets that complete an epoch at the current packet length: K for disjoint frames, see crn_ingest_create for overlapped ones. */
CRN_API int crn_ingest_packets_per_epoch(crn_ingest *g, int32_t *n_packets);
/* Packets refused with CRN_ERR_BUSY so far (a caller that waits and pushes again has each such packet counted once per refusal). */
CRN_API int crn_ingest_dropped(crn_ingest *g, int64_t *n_packets);
/* Counters of a ring since its creation (the operational view the reference has only as printf lines: SURVEY.md §8b proposed
 * `crn_sense_stats`).  Never blocks; call from the thread that pushes. */
typedef struct crn_ingest_stats {
  int64_t packets;           /* packets accepted */
  int64_t dropped;           /* packets refused with CRN_ERR_BUSY */
  int64_t batches;           /* launches (H2D + kernel + D2H) handed to the GPU */
  int64_t batches_failed;    /* launches that failed (their epochs are lost; the error is reported once by the next call) */
  int64_t epochs_launched;   /* epochs those batches carried */
  int64_t epochs_ready;      /* epochs whose results came back */
  int64_t epochs_polled;     /* of which crn_ingest_poll has handed out */
  double latency_us_sum;     /* per batch: hand-off by the pushing thread -> results readable, summed over `batches` that came back */
  double latency_us_max;
} crn_ingest_stats;
CRN_API int crn_ingest_get_stats(crn_ingest *g, crn_ingest_stats *out);
CRN_API int crn_ingest_destroy(crn_ingest *g);

/* -- multi-GPU: the occupancy exchange ----------------------------------------------------------
 * The path shards by stream: every (stream, epoch) is independent — the only state that crosses
 * frames is the K-frame mean inside one epoch of one stream (fft_avg[], CE_Predictive_Node.hpp:51) —
 * so each GPU (one process per GPU, one crn_handle each) runs crn_sense_run_device on its own
 * streams' batch with no data-path collective.  The one exchange: every node's engine needs the whole
 * occupancy picture to pick a free channel, so each rank contributes its [epochs][n_bands] uint8
 * block to an all-gather over RCCL (xGMI inside a node).  Latency-bound (KiB per rank): one
 * collective per batch, queued on a side stream behind an event so that it overlaps the next launch;
 * `depth` slots so that step i + 1's kernel can write while step i's gather is in flight.
 *
 *   rank 0: crn_comm_unique_id(id), handed to the other ranks out of band (any launcher's store)
 *   all:    crn_comm_create(device, rank, world, id, epochs * n_bands, 2, &c)        (collective)
 *   step i: crn_comm_local(c, i, stream, &occ)   -> crn_out.occupancy = occ; crn_sense_run_device(.., stream)
 *           crn_comm_allgather(c, i, stream)     -> returns at once; the gather runs on the side stream
 *           ... crn_comm_gathered(c, i, &all)    valid once `stream` has passed crn_comm_finish / the
 *                                                crn_comm_local of step i + depth
 * RCCL is loaded at run time by crn_comm_unique_id / crn_comm_create — librccl.so.1, or exactly the library $CRN_RCCL_LIB names
 * (if that does not load the calls fail: no other RCCL is tried) — so single-GPU users of libcrnsense do not need it. */
typedef struct crn_comm crn_comm;
#define CRN_COMM_ID_BYTES 128   /* sizeof(ncclUniqueId) */

CRN_API int crn_comm_unique_id(uint8_t id[CRN_COMM_ID_BYTES]);
/* Collective over all `world` ranks (blocks until every rank has called it).  Allocates, once, the
 * `depth` local and gathered slots on `device` and a private side stream.  Like ncclCommInitRank itself it cannot tell the
 * other ranks about a local failure: if this call fails on one rank (bad argument, no memory, RCCL missing) the others stay
 * inside the collective, so the launcher must take the job down — torch.distributed.run does; a caller with a control
 * plane of its own should first agree that every rank can load RCCL (crn_comm_unique_id into a scratch id is a
 * non-collective probe: bench.py / sharding.py do exactly that). */
CRN_API int crn_comm_create(int32_t device, int32_t rank, int32_t world, const uint8_t id[CRN_COMM_ID_BYTES],
                            int64_t bytes_per_rank, int32_t depth, crn_comm **out);
/* What the communicator is, asked of RCCL itself (ncclCommCount / ncclCommUserRank / ncclCommCuDevice / ncclGetVersion on the
 * communicator crn_comm_create built — not an echo of its arguments), so that a caller can state, and a reader of its output can
 * check, that the collective really spans `nranks` ranks on `nranks` different GPUs: bench.py puts every rank's answer into its N > 1
 * JSON line and refuses to report a real-RCCL run whose ranks do not name N distinct PCI devices.  Fields RCCL cannot answer (a library
 * without the query) read -1.  Not collective; never blocks. */
typedef struct crn_comm_info_t {
  int32_t nranks;         /* ncclCommCount */
  int32_t rank;           /* ncclCommUserRank */
  int32_t rccl_device;    /* ncclCommCuDevice: the HIP device RCCL bound this rank to */
  int32_t rccl_version;   /* ncclGetVersion (e.g. 22203) */
  int32_t device;         /* the device crn_comm_create was given */
  int32_t depth;
  int64_t bytes_per_rank;
  int64_t gathers;        /* all-gathers queued through crn_comm_allgather so far */
  char library[128];      /* the name the RCCL library was loaded under (librccl.so.1, or $CRN_RCCL_LIB) */
  char pci_bus_id[32];    /* hipDeviceGetPCIBusId of rccl_device (of `device` when RCCL cannot say): "0000:05:00.0" — the physical GPU, whatever
                             ordinal a launcher's device isolation gave it in this process (ABI version 4) */
} crn_comm_info_t;
CRN_API int crn_comm_info(crn_comm *c, crn_comm_info_t *out);
/* Device address of the block step `step` writes (slot step % depth).  If that slot's previous gather
 * is still in flight, `stream` is made to wait for it (on the device; the host does not block). */
CRN_API int crn_comm_local(crn_comm *c, int64_t step, void *stream, uint8_t **d_local);
/* The same address without touching the slot's state or any stream (for reading the block back, say). */
CRN_API int crn_comm_local_addr(crn_comm *c, int64_t step, uint8_t **d_local);
/* Queue the all-gather of step `step`'s block behind everything enqueued on `stream` so far.  Only enqueues. */
CRN_API int crn_comm_allgather(crn_comm *c, int64_t step, void *stream);
/* Device address of step `step`'s gathered vector, [world][bytes_per_rank], rank order. */
CRN_API int crn_comm_gathered(crn_comm *c, int64_t step, const uint8_t **d_all);
/* Make `stream` wait for step `step`'s gather alone (work queued on `stream` after the call may read crn_comm_gathered(step)).
 * The slot stays marked in flight: whichever stream later takes it through crn_comm_local, or calls crn_comm_finish, still waits. */
CRN_API int crn_comm_wait(crn_comm *c, int64_t step, void *stream);
/* Make `stream` wait for every gather queued so far (call before the final synchronise). */
CRN_API int crn_comm_finish(crn_comm *c, void *stream);
CRN_API int crn_comm_destroy(crn_comm *c);

/* -- the transform on its own ------------------------------------------------------------------
 * Unnormalised forward DFT X[k] = sum_n x[n] exp(-j 2 pi k n / N) (the contract of liquid-dsp's
 * fft_execute with LIQUID_FFT_FORWARD, CE_Predictive_Node.cpp:42-45,150) of n_frames frames:
 * frame i takes samples_per_frame (<= fft_len, zero-padded) interleaved complex fp32 samples starting
 * frame_stride samples (0 = samples_per_frame) after frame i-1, and yields fft_len complex bins at
 * d_out + 2 * fft_len * i.  Same passes as the sensing kernel, complex output instead of the fused
 * accumulate.  include/crn_liquid_fft.h puts liquid's three entry points on top of this. */
CRN_API int crn_fft_forward_device(crn_handle *h, const float *d_in, int64_t n_frames, int32_t samples_per_frame,
                                   int64_t frame_stride, float *d_out, void *stream);

/* -- monitor rows -----------------------------------------------------------
 * The display side of the reference's GNU Radio monitor — qtgui.freq_sink_c(fft_size 1024,
 * firdes.WIN_BLACKMAN_hARRIS), set_fft_average(0.1), and the waterfall sink (spectrum_analyzer.py:262-275) —
 * over rows of the `spectrum` output of crn_sense_run_device (window CRN_WINDOW_BLACKMAN_HARRIS, mode
 * CRN_MODE_ENERGY; frames_per_epoch frames per row, 1 for the sink's one-FFT-per-update):
 *   d_waterfall_db[r][j]  10 log10(P_r[(j + N/2) mod N] / norm)         fftshifted, like the sinks draw it
 *   d_average_db[r][j]    the single-pole IIR over rows, a = alpha:  y <- (1 - a) y + a x
 *   d_state[j]            the IIR state, carried between calls (first != 0: seeded with row 0)
 * CRN_MONITOR_GNURADIO: norm = N^2 and the IIR runs on the dB values (gr-qtgui 3.7 freq_sink_c_impl.cc: volk
 *   power_spectral_density then d_magbuf = (1 - a) d_magbuf + a new — GNU Radio is third-party and absent from
 *   the reference tree; this is its published algorithm).
 * CRN_MONITOR_PSD: norm = N * sum(w^2) and the IIR runs on linear power (a Welch-style averaged PSD), dB after. */
typedef enum crn_monitor_kind { CRN_MONITOR_GNURADIO = 0, CRN_MONITOR_PSD = 1 } crn_monitor_kind;
CRN_API int crn_monitor_rows_device(crn_handle *h, const float *d_spectrum, int64_t n_rows, int32_t kind, float alpha,
                                    int32_t first, float *d_state, float *d_waterfall_db, float *d_average_db, void *stream);

/* -- training -------------------------------------------------------------
 * The reference ships weights for one FFT size and one receiver gain (CE_Predictive_Node.cpp:78-120)
 * and no way to make others.  crn_ann_train_device fits the same 4-5-3 sigmoid network
 * (CE_Predictive_Node.hpp:20-22,62-73) to labelled features that are already on the device — e.g.
 * the `features` output of crn_sense_run_device over crn_synth_fill_device traffic with its d_truth —
 * by full-batch backpropagation (squared error, momentum), `restarts` independent initialisations in
 * parallel (one workgroup each), and returns the best in the reference's array layout, ready for
 * crn_cfg.ann_w_ih / ann_w_ho.  With `normalise`, feature i is scaled by 1/mean_i during training and
 * the gains are folded back into W_IH, so inference stays exactly the reference's forward pass. */
typedef struct crn_train_cfg {
  uint64_t seed;
  int32_t iterations;  /* gradient steps */
  int32_t restarts;    /* >= 1 */
  float eta;           /* learning rate */
  float alpha;         /* momentum */
  int32_t normalise;   /* 0 / 1 */
  int32_t reserved;
} crn_train_cfg;

/* d_features: [n][4] float32 in ANN input order {NOISE_FLOOR, CH1, CH2, CH3}; d_labels: [n] int32,
 * 0 = no channel occupied, k = channel k.  w_ih / w_ho / final_loss are host pointers.  Blocking. */
CRN_API int crn_ann_train_device(crn_handle *h, const crn_train_cfg *tc, const float *d_features,
                                 const int32_t *d_labels, int64_t n, double w_ih[5][6], double w_ho[6][4],
                                 double *final_loss, void *stream);

/* -- measurement / test aids ------------------------------------------------------------- */

/* Fill d_iq with n_epochs * samples_per_epoch seeded synthetic samples on the device:
 * complex AWGN with E|x|^2 = noise_power, plus — for the band the per-epoch occupancy pattern
 * selects (a seeded uniform pick among {none, band 1, .., band n_active}, mirroring the PU of
 * cognitive_engines/CE_Random_Behaviour_PU/CE_Random_Behaviour_PU.cpp:41-53) — tones_per_band
 * on-grid tones of total RMS amplitude signal_rms.  d_truth (nullable) receives the picked band
 * per epoch (0 = none).  The FFT grid is the handle's fft_len; "active" bands are the handle's
 * bands 1..min(3, n_bands-1) (for Welch cfgs: all bands, pick in 0..n_bands). */
CRN_API int crn_synth_fill_device(crn_handle *h, float *d_iq, int64_t n_epochs,
                          int64_t samples_per_epoch, uint64_t seed, float noise_power,
                          float signal_rms, int32_t tones_per_band, int32_t *d_truth,
                          void *stream);

/* Traffic models of the driven primary user (which band an epoch's signal occupies):
 *  CRN_PU_UNIFORM           seeded uniform pick among {idle, band 1 .. band n_active}, independent per
 *                           epoch — cognitive_engines/CE_Random_Behaviour_PU/CE_Random_Behaviour_PU.cpp:41-53
 *                           (rand() % 3 over the channels) plus an idle state.
 *  CRN_PU_MARKOV_AS_WRITTEN the chain of CE_PU_MARKOV_Chain_Tx.cpp:88-128 exactly as its conditions
 *                           evaluate: outcome = rand() % 10; 0 -> CH1, anything else -> CH2 (the
 *                           `>= 1 || < 4` tests are always true), CH3 unreachable, never idle.
 *  CRN_PU_MARKOV_INTENDED   the same chain with the thresholds its comments intend (`&&`): from
 *                           CH1/CH3: 0 -> CH1, 1..3 -> CH2, 4..9 -> CH3; from CH2: 0 -> CH1,
 *                           1..5 -> CH2, 6..9 -> CH3.  Never idle.
 *  CRN_PU_SWEEP            the interferer's TX_FREQ_BEHAVIOR_SWEEP (src/interferer.cpp:339-345: step up by one
 *                           resolution per dwell, reverse at either end) with one band per epoch:
 *                           1, 2, .., n_active, n_active - 1, .., 1, 2, ..  Never idle.
 * Markov and sweep models run per stream: the batch is n_streams consecutive runs of n_epochs / n_streams
 * epochs, each started in CH1 (the Markov chains independent; d_truth is required for them). */
typedef enum crn_pu_model { CRN_PU_UNIFORM = 0, CRN_PU_MARKOV_AS_WRITTEN = 1, CRN_PU_MARKOV_INTENDED = 2, CRN_PU_SWEEP = 3 } crn_pu_model;

/* What the occupied band carries (src/interferer.cpp:128-140 CW / NOISE, :160-215 GMSK, :221-248 RRC, :254-282 OFDM):
 *  CRN_SIG_TONES      tones_per_band on-grid tones spread over the band, fixed random phases per epoch
 *  CRN_SIG_CW         one carrier at the band's centre bin
 *  CRN_SIG_BAND_NOISE every bin of the band, fresh random phases every fft_len samples (a
 *                     frame-synchronous multicarrier burst: flat in-band spectrum)
 * The three below are continuous modulated carriers at the band's centre frequency (midway between its lowest and
 * highest bin, bins >= fft_len / 2 counted as negative frequencies), NOT aligned to the FFT grid or to frames (they
 * leak like a real transmitter); "band width" is the band's number of bins:
 *  CRN_SIG_RRC_QPSK   random QPSK symbols through a root-raised-cosine pulse, roll-off 0.35 (RRC_BETA,
 *                     include/interferer.hpp:21), truncated at +-8 symbols; symbol rate such that the occupied
 *                     bandwidth (1 + beta) Rs is the band's width
 *  CRN_SIG_GMSK       constant-envelope GMSK, BT = 0.5, modulation index 1/2 (the gmskframegen the interferer
 *                     drives); symbol rate = band width / 1.5 (>= 99.9 % of the power inside the band)
 *  CRN_SIG_OFDM       QPSK subcarriers 15 kHz apart at the 13 MHz sample rate of CE_Predictive_Node.hpp:42-43
 *                     (the interferer's tx_rate / num_subcarriers, src/interferer.cpp:255), i.e.
 *                     fft_len * 15e3 / 13e6 bins, filling the band; cyclic prefix 1/4 of the useful symbol, no taper
 * All kinds have total power signal_rms^2. */
typedef enum crn_signal_kind {
  CRN_SIG_TONES = 0, CRN_SIG_CW = 1, CRN_SIG_BAND_NOISE = 2, CRN_SIG_RRC_QPSK = 3, CRN_SIG_GMSK = 4, CRN_SIG_OFDM = 5
} crn_signal_kind;

typedef struct crn_synth_cfg {
  uint64_t seed;
  float noise_power;      /* E|x|^2 of the complex AWGN */
  float signal_rms;       /* RMS amplitude of the PU signal */
  int32_t tones_per_band; /* CRN_SIG_TONES only */
  int32_t pu_model;       /* crn_pu_model */
  int32_t signal_kind;    /* crn_signal_kind */
  int32_t n_streams;      /* >= 1; must divide n_epochs for the Markov models */
  int32_t adc_bits;       /* 0: full fp32 samples.  2..24: every component rounded to a multiple of 2^-(adc_bits - 1) and clipped to
                           * [-1, 1): what a radio delivers — the reference's USRPs send 16-bit integers over the wire (4 bytes per
                           * complex sample: the 363-364 samples of a 1500-byte packet, src/extensible_cognitive_radio.cpp:1263-1265),
                           * which UHD converts to the complex floats of recv(.., COMPLEX_FLOAT32, ..) (:1071-1072) with a constant
                           * of its own (1/32767: the values then are not on this power-of-two grid) */
} crn_synth_cfg;

/* crn_synth_fill_device with a traffic model and a signal kind.  crn_synth_fill_device(.., seed,
 * noise_power, signal_rms, tones, ..) == this with {CRN_PU_UNIFORM, CRN_SIG_TONES, n_streams 1}. */
CRN_API int crn_synth_fill_device_ex(crn_handle *h, const crn_synth_cfg *sc, float *d_iq, int64_t n_epochs,
                                     int64_t samples_per_epoch, int32_t *d_truth, void *stream);

/* Noise-floor estimate for threshold plans (thr_b = lambda x NF_est, NF_est = the median band energy): the
 * median over epochs of every epoch's median band energy, from a features matrix [n_epochs][n_bands] on the device (what
 * crn_sense_run_device wrote; at most the first 4096 epochs are used).  "Median" = element (n - 1) / 2 of the sorted values.  With
 * fewer than half of the bands occupied the estimate does not see the signals.  Enqueues on `stream` and waits for the result. */
CRN_API int crn_noise_floor_device(crn_handle *h, const float *d_features, int64_t n_epochs, float *nf_out, void *stream);
/* Replace the per-band thresholds of a handle (cfg.thresh; CRN_DECIDE_THRESHOLD): ordered on `stream` — launches enqueued on it
 * after the call see the new values, launches before it the old ones.  `thresh` is read before the call returns (it is staged in
 * pinned memory of the handle's own, one slot per update: the asynchronous copy never reads memory a later call rewrites).
 * Updates (this call, crn_sense_set_ann, crn_sense_calibrate_thresholds) cannot be captured into a hipGraph — a replay would upload
 * whatever the reused staging slot holds by then: on a stream that is capturing they return CRN_ERR_STATE and enqueue nothing.
 * Launches capture fine.  With all eight staging slots still in flight an update waits for the oldest with the handle's lock released
 * (a launch on another thread never waits for an update's copy); if a crn_sense_set_bands on another thread changed the number of
 * bands (or the decision rule) in that window the update returns CRN_ERR_STATE and writes nothing. */
CRN_API int crn_sense_set_thresholds(crn_handle *h, const float *thresh, int32_t n_bands, void *stream);

/* Allocate, now, what crn_noise_floor_host and crn_sense_calibrate_thresholds need (pinned + device upload buffers for 4096 epochs
 * of CRN_MAX_BANDS features, the reduction scratch): after it neither allocates.  An engine calls it from its constructor. */
CRN_API int crn_sense_reserve_noise_floor(crn_handle *h);
/* Calibration in one call, for callers that hold the features on the host (the engine's synchronous form; the ingest ring's launcher
 * thread): upload features [n_epochs][n_bands] (n_epochs <= 4096), crn_noise_floor_device, then every band's threshold = lambda x
 * the estimate, set on `stream` like crn_sense_set_thresholds.  Blocks for the reduction; allocates nothing (CRN_ERR_STATE unless
 * crn_sense_reserve_noise_floor ran). */
CRN_API int crn_sense_calibrate_thresholds(crn_handle *h, const float *features, int64_t n_epochs, float lambda, float *nf_out, void *stream);
/* The same estimate from a features matrix in host memory (an engine that keeps no device buffers of its own): uploaded, then
 * crn_noise_floor_device.  Blocking; not for the packet path.  Allocates its upload buffers on the first call unless
 * crn_sense_reserve_noise_floor did. */
CRN_API int crn_noise_floor_host(crn_handle *h, const float *features, int64_t n_epochs, float *nf_out);
/* Replace the network of a DECIDE_ANN handle (cfg.ann_w_ih / ann_w_ho / ann_threshold; the reference has its weights as
 * literals, CE_Predictive_Node.cpp:78-120): ordered on `stream` like crn_sense_set_thresholds.  The threshold rides in the launch
 * parameters: launches made after the call use it. */
CRN_API int crn_sense_set_ann(crn_handle *h, const double w_ih[5][6], const double w_ho[6][4], double threshold, void *stream);
/* Replace the band plan of a live handle (cfg.segs / n_segs / n_bands; the reference's is the five loops of
 * CE_Predictive_Node.cpp:173-191).  thresh: n_bands new thresholds, or NULL to keep the current ones (then n_bands must not
 * change).  Same validation as crn_sense_create.  Every table derived from the plan is rebuilt into a fresh allocation and the
 * call waits for launches in flight on the device before it frees the old one: not for the packet path.  Safe against an attached
 * ingest ring's launcher thread: the swap and the release of the old tables happen under the handle's lock, which every launch
 * holds from its first read of the plan until its kernel is enqueued.  Ingest rings size their result buffers for the n_bands they
 * were created with: a call that changes n_bands is refused (CRN_ERR_STATE) while one is attached. */
CRN_API int crn_sense_set_bands(crn_handle *h, const crn_band_seg *segs, int32_t n_segs, int32_t n_bands, const float *thresh);

/* Wait until everything queued on `stream` (NULL = the default stream) of the handle's device has completed: for callers that are
 * pure host code against this ABI (the engine) and need an asynchronous update to have landed before they go on. */
CRN_API int crn_sense_synchronize(crn_handle *h, void *stream);

/* Counters of a sensing handle since its creation.  `samples` counts every input sample a launch covers once (8 bytes each: the
 * algorithmic read of the path), so samples * 8 / kernel seconds is the same figure bench.py's roofline reports.  Durations are
 * measured only while crn_sense_set_timing(h, 1) is in effect (two HIP events per launch on the launch stream, a ring of 16 pairs;
 * leave it off while capturing launches into a hipGraph); crn_sense_get_stats collects those that have finished and never blocks. */
typedef struct crn_sense_stats {
  int64_t launches;          /* crn_sense_run_device calls that launched (run_host and the ingest ring go through it) */
  int64_t epochs;
  int64_t samples;
  int64_t timed_launches;    /* launches whose duration is in kernel_ms */
  double kernel_ms;          /* sum of their durations */
  double kernel_ms_last;
  double kernel_ms_min, kernel_ms_max;
} crn_sense_stats;
CRN_API int crn_sense_set_timing(crn_handle *h, int32_t on);
CRN_API int crn_sense_get_stats(crn_handle *h, crn_sense_stats *out);

/* Launches of a few epochs — up to one per compute unit; the engine's launch is ONE — at fft_len 512 / 1024 (any window, disjoint or
 * overlapped frames) run the sensing kernel in its dealt-frame form: one epoch per workgroup, its frames_per_epoch frames spread over the workgroup's lane
 * groups (8 at 512 points, 4 at 1024) instead of run one after the other by one of them, the K-frame accumulate replayed in frame
 * order afterwards: the same operations in the same order, so every output is bit for bit what the streaming form gives, in about
 * ceil(K / groups) frame latencies instead of K.  Chosen per launch from n_epochs; *n = how many launches of this handle RAN that
 * form (a launch the device refused the dealt form's LDS for runs the streaming form and is not counted).  (crn_sense_set_variant 400 / 401 / 402: automatic / never / at any batch size — for measurements and the equality test.) */
CRN_API int crn_sense_dealt_launches(crn_handle *h, int64_t *n);

/* Name, registers and LDS of the sensing kernel selected for this handle. */
CRN_API int crn_sense_kernel_info(crn_handle *h, char *name, int32_t name_len, int32_t *threads_per_block,
                          int32_t *lds_bytes, int32_t *epochs_per_block);

/* Which form of the sensing kernel a handle launches (set before running).  The shipped library carries the forms that are sensing
 * results: 0 (= 13) the default; 2 without the pruning of pass 3 to the registers the reference channel plan reaches (what any band
 * table outside that plan runs anyway); 23 every twiddle in registers at three workgroups per CU.  100 + n / 200 + n / 300 + n
 * override the launch geometry (epochs per big workgroup; x 256 epochs in tail workgroups; epochs per tail workgroup) and change
 * no result.  The measurement forms (7, 17, 19-22, 26, 27: the same flags in other combinations, and a build with in-kernel time
 * stamps) are compiled only into libcrnsense_ab.so (make -C csrc ab) and refused here with CRN_ERR_ARG; every other number is refused
 * by both. */
CRN_API int crn_sense_set_variant(crn_handle *h, int32_t variant);

/* (Optional, not in this library: samples kept in the radio's wire format, int16 pairs — include/crn_sense_sc16.h, libcrnsense_sc16.so.) */

CRN_API const char *crn_last_error(void);
CRN_API int crn_abi_version(void);
/* Toolchain contract.  built_hip: HIP_VERSION of the headers the library was compiled with (major * 10^7 + minor * 10^5 + patch);
 * runtime_hip: what hipRuntimeGetVersion reports on this machine (0 when no runtime answers).  The library links libamdhip64.so.<major>
 * and carries a gfx950 code object only: it runs on any ROCm whose HIP runtime has the SAME MAJOR version as built_hip and is not
 * older than ROCm 7.0 (the first release with gfx950).  Returns CRN_OK when that holds, CRN_ERR_STATE (with the two versions in
 * crn_last_error) when it does not.  Either pointer may be NULL. */
CRN_API int crn_build_info(int32_t *built_hip, int32_t *runtime_hip);

#ifdef __cplusplus
}
#endif
#endif /* CRN_SENSE_H */

// crn_kernels.h — kernel parameter blocks shared by crn_kernels.hip and crn_api.cpp (internal).
#ifndef CRN_KERNELS_H
#define CRN_KERNELS_H

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace crn {

constexpr int kBandTabWords = 656;   // SenseParams::band_tab, copied to LDS by every workgroup
constexpr int kRowEntryWords = 32;   // row-entry slots of the register-resident band sums: 32 / R3 per row

enum { CRN_DECIDE_ANN_K = 0, CRN_DECIDE_THRESHOLD_K = 1, CRN_DECIDE_NONE_K = 2 };  // == crn_decide

struct SenseParams {
  // input stream
  const float2 *iq;        // device, interleaved complex fp32
  long long n_epochs;
  long long epoch_stride;  // samples between epoch starts
  long long total_samples; // samples the batch holds (loads beyond it return zero)
  int frame_stride;        // samples between frame starts inside an epoch
  int L;                   // samples taken per frame (zero-padded to N)
  int K;                   // frames per epoch
  int groups_per_wg;       // consecutive epoch groups one workgroup streams through (>= 1)
  long long n_big_wgs;     // workgroups [0, n_big_wgs) take groups_per_wg groups each; the ones after them
                           // tail_groups_per_wg each (they are dispatched last: the end of the kernel drains in small steps)
  int tail_groups_per_wg;  // >= 1 (1 for the plain kernels; the Welch stream, which re-reads one half-frame per workgroup span,
                           // keeps its tail workgroups longer)
  // tables (device, built at crn_sense_create)
  const float2 *tw1;       // [17][T]  W_N^{t a}, a = 0..16
  const float2 *tw2;       // [16][R3] W_T^{m c}
  const float *window;     // [N] or null
  const int *band_tab;        // [kBandTabWords] packed copy of the tables below, staged into LDS by every workgroup:
                              //   [0,96) band_seg_begin, [96,256) seg_lo, [256,416) seg_hi, [416,496) thresh (float bits),
                              //   [512,544) row entries band<<18 | lo<<9 | hi, 32 / R3 slots per 256-bin row, 0 = unused
                              //   (only when n_row_entries > 0), [544,604) ann_w_ih as 30 doubles, [604,652) ann_w_ho as 24 doubles
  const int *band_seg_begin;  // [n_bands + 1] into seg_lo/seg_hi (segments grouped by band)
  const int *seg_lo;
  const int *seg_hi;
  const float *thresh;     // [n_bands]
  const double *ann_w_ih;  // [5][6]
  const double *ann_w_ho;  // [6][4]
  double ann_threshold;
  int n_bands;
  int decide;
  int ref_band;
  int n_row_entries;       // > 0: band sums from registers (epoch_close); entries in band_tab
  int aligned_shift;       // N = 4096 and the plan is n_bands equal contiguous bands of 2^aligned_shift bins (6..8), else 0
  int hann_sym;            // the window table is a periodic Hann: w[n + N/2] = 1 - w[n] (may be folded into pass 1)
  unsigned acc_mask;       // bit j R3 + d set when some band holds a bin of the form a + 16 (g J + j) + 256 d, i.e. when accumulator
                           // register j R3 + d of some thread holds a band bin (N = 4096: bit d = the 256-bin row d)
  int deal_rounds;         // > 0: sense_kernel_dealt (one epoch per workgroup, its frames dealt to the lane groups, this many rounds of them);
                           // set by the host for launches of a few epochs at N <= 1024 (crn_api.cpp)
  float wire_unscale;      // wire-format launches: 1 / full scale (2^-15 by default) for a sum of magnitudes, its square for energies
  // outputs (device, nullable)
  float *features;
  double *ann_out;
  int32_t *decision;
  uint8_t *occupancy;
  float *spectrum;
};

struct SynthParams {
  float2 *iq;
  long long n_epochs;
  long long samples_per_epoch;
  unsigned long long seed;
  float noise_sigma;  // per component
  float tone_amp;     // per tone
  int tones;
  int fft_len;
  int n_active;       // pick uniformly in 0..n_active (0 = idle)
  int active_band0;   // first band index that can be picked
  const int *band_bins_begin;  // [n_bands + 1]
  const int *band_bins;        // flattened bin lists per band
  const int *band_c2;          // [n_bands] twice the band's signed centre bin (modulated carriers)
  int32_t *truth;              // [n_epochs] or null
  int pu_model;                // crn_pu_model; the Markov models read truth[] (filled by launch_pu_pattern)
  int signal_kind;             // crn_signal_kind
  float signal_rms;
  long long epochs_per_stream; // Markov models
  float adc_scale;             // 0: off; 2^(adc_bits - 1): components rounded to multiples of 1 / adc_scale, clipped to [-1, 1)
};

// *deal_rounds_run (when asked for) = the form that was really launched: p.deal_rounds when the dealt-frame kernel took it, 0 when the
// streaming kernel did — also when the device refused the dealt form's LDS and the launch fell back (crn_sense_dealt_launches counts from this)
hipError_t launch_sense(const SenseParams &p, int fft_len, bool mag, bool win, int variant,
                        hipStream_t stream, bool sc16 = false, int *deal_rounds_run = nullptr);
int sense_num_variants();
// Rounds of dealt frames (ceil(K / lane groups)) when sense_kernel_dealt can take this size / mode / window / K — N <= 1024, no window or
// the periodic Hann on whole frames in energy mode, and the frame slots fit in the device's LDS per workgroup (lds_budget_bytes:
// hipDeviceAttributeMaxSharedMemoryPerBlock, 160 KiB on gfx950) — else 0: the streaming kernel takes the launch.
int sense_deal_rounds(int fft_len, bool mag, bool win, bool hann_whole_frames, int frames_per_epoch, size_t lds_budget_bytes);
unsigned sense_ref_acc_mask(int fft_len);    // accumulator registers (bit j R3 + d) the reference channel plan reaches at this size
bool sense_variant_available(int variant);   // the shipped library carries 0 (= 13) and 2; libcrnsense_ab.so the measurement forms too
bool sense_variant_traces(int variant);      // a measurement form that writes time stamps over the ann_out buffer (never in the shipped library)
void sense_variant(int fft_len, int variant, int *nbuf, int *prefetch, int *nt, int *tw2lds, int *pk);
void sense_geometry(int fft_len, int variant, int *threads, int *lds_bytes, int *epochs_per_block);
struct FftParams {
  const float2 *in;        // [n_frames] frames of L samples, frame_stride samples apart
  float2 *out;             // [n_frames][N]
  long long n_frames;
  long long frame_stride;
  int L;                   // samples taken per frame (zero-padded to N)
  const float2 *tw1;       // [17][T]
  const float2 *tw2;       // [16][R3]
};
hipError_t launch_fft(const FftParams &p, int fft_len, hipStream_t stream);
struct MonitorParams {
  const float *spectrum;   // [n_rows][n] the sensing kernel's per-bin output (mean |X|^2 over the row's frames)
  long long n_rows;
  int n;                   // fft_len (power of two)
  float alpha;             // IIR weight of the new row
  float scale;             // 1 / normalisation of |X|^2
  int db_domain;           // 1: average the dB values (gr-qtgui); 0: average linear power
  int first;               // 1: the state is seeded with row 0
  float *state;            // [n] IIR state per displayed column, in / out
  float *waterfall_db;     // [n_rows][n] or null
  float *average_db;       // [n_rows][n] or null
};
hipError_t launch_monitor(const MonitorParams &p, hipStream_t stream);
// noise-floor estimate: scratch holds kNoiseFloorMaxEpochs per-epoch medians + the result
constexpr int kNoiseFloorMaxEpochs = 4096;
hipError_t launch_noise_floor(const float *feat, int n_epochs, int nb, float *scratch, hipStream_t stream);
hipError_t launch_synth(const SynthParams &p, hipStream_t stream);
hipError_t launch_pack_sc16(const float *iq, long long n_samples, short *out, float full_scale, hipStream_t stream);   // crn_kernels_sc16.hip (make SC16=1)
hipError_t launch_pu_pattern(const SynthParams &p, hipStream_t stream);
hipError_t launch_nop(hipStream_t stream);   // one empty workgroup (crn_sense_warm_stream)

}  // namespace crn
#endif

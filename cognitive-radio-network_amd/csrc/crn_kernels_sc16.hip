// crn_kernels_sc16.hip — OPTIONAL (make SC16=1 -> libcrnsense_sc16.so): samples kept in HBM in the radio's wire format (int16 pairs, 4 bytes
// per complex sample).  The wire-format (kSc16) instantiations of sense_kernel, their dispatch, and the float -> int16 pack kernel.  The
// default library is built without this file: crn_kernels.hip's launch_sense refers to launch_sense_sc16 weakly.
#include "crn_sense_impl.h"

namespace crn {

// Wire-format input (kSc16): the default kernels of every size, mode and window, the plain 4096-point kernel's three forms, and the
// Welch configuration's kernel (periodic Hann, whole frames, energy).
template <int R3>
static hipError_t launch_r_sc16(const SenseParams &p, bool mag, bool win, int variant, hipStream_t stream, int *deal_rounds_run) {
  constexpr int kBase = kSpread | kLdsBlk | kPrioValu | kMulti | kSc16;
  if constexpr (R3 <= 4) {   // a launch of a few epochs: frames dealt to the workgroup's lane groups (sense_kernel_dealt)
    if (p.deal_rounds > 0) {
      const hipError_t e = win ? launch_dealt_win<R3, kSc16>(p, stream) : launch_dealt<R3, kSc16>(p, mag, stream);
      if (e != hipErrorLaunchOutOfResources) {
        if (e == hipSuccess && deal_rounds_run) *deal_rounds_run = p.deal_rounds;
        return e;
      }
      SenseParams q = p;   // (the device refused the LDS the frame slots need: the streaming form)
      q.deal_rounds = 0;
      return launch_r_sc16<R3>(q, mag, win, variant, stream, deal_rounds_run);
    }
  }
  if (win) {
    // everything that is not the Welch configuration's kernel: the generic windowed kernels (window table in registers)
    if (mag || !p.hann_sym || p.L != Geo<R3>::N) return launch_default<R3, 1, true, true, true, 3, true, kBase, 2, false>(p, mag, win, stream);
    if constexpr (R3 == 16) {
      if (p.aligned_shift != 0)
        return launch_cfg<Cfg<R3, 1, true, true, false, true, true, 3, true, true, kBase | kHannSym | kTw2Early | kAlignedBands>>(p, stream);
    }
    return launch_cfg<Cfg<R3, 1, true, true, false, true, true, 3, true, true, kBase | kHannSym | kTw2Early>>(p, stream);
  }
  if constexpr (R3 == 16) {
    if (!mag && p.L == Geo<R3>::N) {
      if (reg_bands(p) && (p.acc_mask & ~kRefPlanRows) == 0)
        return launch_rn<R3, 1, true, true, true, 4, true, kBase | kTw1C | kRows | kRegBands>(p, mag, win, stream);
      if (reg_bands(p)) return launch_rn<R3, 1, true, true, true, 4, true, kBase | kTw1C | kRegBands>(p, mag, win, stream);
      return launch_rn<R3, 1, true, true, true, 4, true, kBase | kTw1C>(p, mag, win, stream);
    }
  }
  return launch_default<R3, 1, true, true, false, 3, true, kBase, 1, false, R3 != 16>(p, mag, win, stream);   // (no plan-specific pruning in wire format)
}

hipError_t launch_sense_sc16(const SenseParams &p, int fft_len, bool mag, bool win, int variant, hipStream_t stream, int *deal_rounds_run) {
  switch (fft_len) {
    case 512: return launch_r_sc16<2>(p, mag, win, variant, stream, deal_rounds_run);
    case 1024: return launch_r_sc16<4>(p, mag, win, variant, stream, deal_rounds_run);
    case 2048: return launch_r_sc16<8>(p, mag, win, variant, stream, deal_rounds_run);
    case 4096: return launch_r_sc16<16>(p, mag, win, variant, stream, deal_rounds_run);
    default: return hipErrorInvalidValue;
  }
}

// complex floats -> the radio's wire format (int16 pairs, full scale 32768): crn_pack_sc16_device
__global__ __launch_bounds__(256) void pack_sc16_kernel(const float2 *iq, long long n, short2 *out, float full_scale) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float2 v = iq[i];
    out[i] = make_short2((short)fminf(fmaxf(rintf(v.x * full_scale), -32768.f), 32767.f), (short)fminf(fmaxf(rintf(v.y * full_scale), -32768.f), 32767.f));
  }
}

hipError_t launch_pack_sc16(const float *iq, long long n_samples, short *out, float full_scale, hipStream_t stream) {
  if (n_samples <= 0) return hipSuccess;
  long long blocks = (n_samples + 255) / 256;
  if (blocks > 256 * 32) blocks = 256 * 32;
  hipLaunchKernelGGL(pack_sc16_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, reinterpret_cast<const float2 *>(iq), n_samples,
                     reinterpret_cast<short2 *>(out), full_scale);
  return hipGetLastError();
}

}  // namespace crn

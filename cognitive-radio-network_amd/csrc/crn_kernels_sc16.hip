// crn_kernels_sc16.hip — the wire-format (kSc16) instantiations of sense_kernel and their dispatch: a translation unit of its own
// so that they compile beside the complex-float ones (csrc/Makefile builds the objects in parallel).
#include "crn_sense_impl.h"

namespace crn {

// Wire-format input (kSc16): the default kernels of every size, mode and window, the plain 4096-point kernel's three forms, and the
// Welch configuration's kernel (periodic Hann, whole frames, energy).
template <int R3>
static hipError_t launch_r_sc16(const SenseParams &p, bool mag, bool win, int variant, hipStream_t stream) {
  constexpr int kBase = kSpread | kLdsBlk | kPrioValu | kMulti | kSc16;
  if constexpr (R3 <= 4) {   // a launch of a few epochs: frames dealt to the workgroup's lane groups (sense_kernel_dealt)
    if (p.deal_rounds > 0) return win ? launch_dealt_win<R3, kSc16>(p, mag, stream) : launch_dealt<R3, kSc16>(p, mag, stream);
  }
  if (win) {
    // everything that is not the Welch configuration's kernel: the generic windowed kernels (window table in registers)
    if (mag || !p.hann_sym || p.L != Geo<R3>::N) return launch_default<R3, 1, true, true, true, 3, true, kBase, 2, false>(p, mag, win, stream);
    if constexpr (R3 == 16) {
      if (p.aligned_shift != 0)
        return launch_cfg<Cfg<R3, 1, true, true, false, true, true, 3, 0, true, true, kBase | kHannSym | kTw2Early | kAlignedBands>>(p, stream);
    }
    return launch_cfg<Cfg<R3, 1, true, true, false, true, true, 3, 0, true, true, kBase | kHannSym | kTw2Early>>(p, stream);
  }
  if constexpr (R3 == 16) {
    if (!mag && p.L == Geo<R3>::N) {
      if (variant == 23) {  // A/B: every twiddle in registers, 3 workgroups per CU (the float path's variant 23)
        if (reg_bands(p) && (p.acc_mask & ~kRefPlanRows) == 0)
          return launch_rn<R3, 1, true, true, false, 3, 0, true, kBase | kRows | kRegBands>(p, mag, win, stream);
        return launch_rn<R3, 1, true, true, false, 3, 0, true, kBase>(p, mag, win, stream);
      }
      if (reg_bands(p) && (p.acc_mask & ~kRefPlanRows) == 0)
        return launch_rn<R3, 1, true, true, true, 4, 0, true, kBase | kTw1C | kRows | kRegBands>(p, mag, win, stream);
      if (reg_bands(p)) return launch_rn<R3, 1, true, true, true, 4, 0, true, kBase | kTw1C | kRegBands>(p, mag, win, stream);
      return launch_rn<R3, 1, true, true, true, 4, 0, true, kBase | kTw1C>(p, mag, win, stream);
    }
  }
  return launch_default<R3, 1, true, true, false, 3, true, kBase, 1, false>(p, mag, win, stream);   // (no plan-specific pruning in wire format)
}

hipError_t launch_sense_sc16(const SenseParams &p, int fft_len, bool mag, bool win, int variant, hipStream_t stream) {
  switch (fft_len) {
    case 512: return launch_r_sc16<2>(p, mag, win, variant, stream);
    case 1024: return launch_r_sc16<4>(p, mag, win, variant, stream);
    case 2048: return launch_r_sc16<8>(p, mag, win, variant, stream);
    case 4096: return launch_r_sc16<16>(p, mag, win, variant, stream);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace crn

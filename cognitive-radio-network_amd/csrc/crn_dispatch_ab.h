// crn_dispatch_ab.h — MEASUREMENT BUILD ONLY (libcrnsense_ab.so, -DCRN_AB_VARIANTS; included by crn_kernels.hip inside namespace crn).  The
// forms crn_sense_set_variant selects besides the product's two (13 = default, 2 = unpruned): combinations of the shipped OPT flags, and
// the build with in-kernel time stamps (kTrace: crn_frame_ab.h).  N = 4096 only, whole frames, energy mode.
//    7  the default without the wave-priority raise in passes 1 and 2
//   17  the default + s_memtime stamps of the epoch close in the ann_out buffer (windowed handles: the Welch kernel with stamps)
//   19  windowed kernels: Hann folded into pass 1's first butterflies          20  = 19 + early pass-2 twiddle reads (what ships)
//   21  windowed kernels: early pass-2 twiddle reads alone                     22  the plain windowed kernel (table window)
//   26  the Welch kernel with pass-2 twiddles in registers, 2 workgroups / CU  27  = 26 + two exchange buffers (one barrier per frame)
static bool measurement_variant(int v) { return v == 7 || v == 17 || (v >= 19 && v <= 22) || v == 26 || v == 27; }
static bool measurement_variant_traces(int v) { return v == 17; }
static void measurement_variant_desc(int v, int *nbuf, int *tw2lds) {   // what crn_sense_kernel_info prints for it
  if (v == 26 || v == 27) *tw2lds = 0;
  if (v == 27) *nbuf = 2;
}

template <int R3>
static bool launch_measurement_form(const SenseParams &p, bool mag, bool win, int variant, hipStream_t stream, hipError_t *e) {
  constexpr int kBase = kSpread | kLdsBlk | kPrioValu | kMulti;
  if (!measurement_variant(variant) || mag || p.L != Geo<R3>::N) return false;
  if constexpr (R3 == 16) {
    if (win && p.hann_sym && (variant == 26 || variant == 27)) {
      constexpr int kW = kBase | kHannSym;
      const bool al = p.aligned_shift != 0;
      if (variant == 26) *e = al ? launch_cfg<Cfg<R3, 1, true, true, false, true, false, 2, true, true, kW | kAlignedBands>>(p, stream)
                                 : launch_cfg<Cfg<R3, 1, true, true, false, true, false, 2, true, true, kW>>(p, stream);
      else *e = al ? launch_cfg<Cfg<R3, 2, true, true, false, true, false, 2, true, true, kW | kAlignedBands>>(p, stream)
                   : launch_cfg<Cfg<R3, 2, true, true, false, true, false, 2, true, true, kW>>(p, stream);
      return true;
    }
    if (win && variant == 17) {   // close stamps for the windowed / Welch kernel
      *e = launch_cfg<Cfg<R3, 1, true, true, false, true, true, 3, true, true, kBase | kTrace>>(p, stream);
      return true;
    }
    if (!win && (variant == 7 || variant == 17)) {
      constexpr int kPlain = kSpread | kLdsBlk | kTw1C | kMulti;
      const bool ref_rows = reg_bands(p) && (p.acc_mask & ~kRefPlanRows) == 0;
      if (variant == 17) *e = launch_rn<R3, 1, true, true, true, 4, true, kPlain | kRows | kPrioValu | kRegBands | kTrace>(p, mag, win, stream);
      else *e = ref_rows ? launch_rn<R3, 1, true, true, true, 4, true, kPlain | kRows | kRegBands>(p, mag, win, stream)
                         : launch_rn<R3, 1, true, true, true, 4, true, kPlain>(p, mag, win, stream);
      return true;
    }
  }
  if (win && variant >= 19 && variant <= 22) {
    if (variant == 19 && p.hann_sym) *e = launch_cfg<Cfg<R3, 1, true, true, false, true, true, 3, true, true, kBase | kHannSym>>(p, stream);
    else if (variant == 20 && p.hann_sym) *e = launch_cfg<Cfg<R3, 1, true, true, false, true, true, 3, true, true, kBase | kHannSym | kTw2Early>>(p, stream);
    else if (variant == 21) *e = launch_cfg<Cfg<R3, 1, true, true, false, true, true, 3, true, true, kBase | kTw2Early>>(p, stream);
    else *e = launch_cfg<Cfg<R3, 1, true, true, false, true, true, 3, true, true, kBase>>(p, stream);
    return true;
  }
  return false;
}

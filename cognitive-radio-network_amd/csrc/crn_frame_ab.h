// crn_frame_ab.h — MEASUREMENT BUILD ONLY (libcrnsense_ab.so, -DCRN_AB_VARIANTS; included by crn_epoch_close.h).  What the shipped
// library does not carry: the kTrace flag and the in-kernel time stamps of variant 17 (tools/gpu_wg_placement.py).  Every other
// measurement variant that is still compiled (7, 19-22, 26, 27: crn_dispatch_ab.h) is a combination of the shipped flags.  The schedules,
// ablations and layouts that were measured and not kept in rounds 1-4 (variants 1, 3-6, 8-12, 14-16, 18, 24, 25) were deleted in
// round 5: docs/history/removed_variants.md names the commit that last held them.
#ifndef CRN_FRAME_AB_H
#define CRN_FRAME_AB_H
#include "crn_frame.h"

namespace crn {
enum : int {
  kTrace = 4096,   // s_memtime / s_memrealtime stamps of the workgroup start and the epoch close, written over the ann_out buffer
};

template <class C>
struct CloseTrace {
  static constexpr bool ON = (C::OPT & kTrace) != 0;
  unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0;
  __device__ __forceinline__ unsigned long long now() { return __builtin_amdgcn_s_memtime(); }
  // workgroup start on the wall clock, behind the [epoch][3] close stamps (the caller's buffer holds 4 words per epoch)
  // ... and, half a buffer further, where it runs: HW_ID (wave / SIMD / CU / SE) in the low word, XCC_ID in the high one
  static __device__ __forceinline__ void workgroup_start(const SenseParams &p, int tid) {
    if constexpr (ON) {
      if (tid == 0 && p.ann_out != nullptr) {
        unsigned long long *tr = reinterpret_cast<unsigned long long *>(p.ann_out);
        tr[p.n_epochs * 3 + blockIdx.x] = __builtin_amdgcn_s_memrealtime();
        tr[p.n_epochs * 3 + p.n_epochs / 2 + blockIdx.x] =
            (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);
      }
    }
  }
  // [epoch][3] uint64: entry of the group's first wave (s_memrealtime: 100 MHz, the same clock on every XCD); its later stamps as
  // four 16-bit deltas in shader clocks (band sums done, barrier passed, features ready, exit); entry of the group's last wave
  __device__ __forceinline__ void enter(const SenseParams &p, long long epoch, bool active, int t, int T) {
    if constexpr (ON) {
      unsigned long long *tr = reinterpret_cast<unsigned long long *>(p.ann_out);
      const unsigned long long wall = __builtin_amdgcn_s_memrealtime();
      t0 = now();
      if (active && tr != nullptr) {
        if (t == 0) tr[epoch * 3 + 0] = wall;
        if (t == T - 1) tr[epoch * 3 + 2] = wall;
      }
    }
  }
  __device__ __forceinline__ void stamp1() { if constexpr (ON) t1 = now(); }
  __device__ __forceinline__ void stamp2() { if constexpr (ON) t2 = now(); }
  __device__ __forceinline__ void stamp3() { if constexpr (ON) t3 = now(); }
  __device__ __forceinline__ void leave(const SenseParams &p, long long epoch, bool active, int t) {
    if constexpr (ON) {
      unsigned long long *tr = reinterpret_cast<unsigned long long *>(p.ann_out);
      const unsigned long long t4 = now();
      auto d16 = [&](unsigned long long x) { return (x - t0) > 0xFFFFull ? 0xFFFFull : (x - t0); };
      if (active && tr != nullptr && t == 0) tr[epoch * 3 + 1] = d16(t1) | (d16(t2) << 16) | (d16(t3) << 32) | (d16(t4) << 48);
    }
  }
};
}  // namespace crn
#endif

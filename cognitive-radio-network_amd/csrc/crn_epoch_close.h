// crn_epoch_close.h — what runs once per K frames: band sums in their three forms, features, the 4-5-3 network and the cascade / the threshold
// decision (reference: CE_Predictive_Node.cpp:157-261, reset at :287-288).
#ifndef CRN_EPOCH_CLOSE_H
#define CRN_EPOCH_CLOSE_H
#include "crn_frame.h"

// In-kernel time stamps exist only in the measurement build (libcrnsense_ab.so, variant 17: crn_frame_ab.h).  The shipped library's
// hooks are empty: no stamp, no instruction.
#ifdef CRN_AB_VARIANTS
#include "crn_frame_ab.h"
#else
namespace crn {
template <class C>
struct CloseTrace {
  CRN_DEV void workgroup_start(const SenseParams &, int) {}
  CRN_DEV void enter(const SenseParams &, long long, bool, int, int) {}
  CRN_DEV void stamp1() {}
  CRN_DEV void stamp2() {}
  CRN_DEV void stamp3() {}
  CRN_DEV void leave(const SenseParams &, long long, bool, int) {}
};
}  // namespace crn
#endif

namespace crn {
// ---------------------------------------------------------------------------------------------
// Epoch close (reference .cpp:157-261 + the reset at :287-288): K-frame averages -> LDS in natural
// bin order -> band sums -> features -> decision.  Resets the accumulators for the next epoch.
// ---------------------------------------------------------------------------------------------
// LDS behind the exchange buffers and the tw2 table, used by the epoch close: the band table copy,
// then [8 teams][16] per-team band partials of the register path.
constexpr int kCloseLdsBytes = kBandTabWords * 4 + 8 * 16 * 4;

// LDS address-space views for the epoch close (see epoch_close): ds_* instructions, lgkmcnt only.
typedef __attribute__((address_space(3))) float lds_f32;
typedef __attribute__((address_space(3))) int lds_i32;
typedef __attribute__((address_space(3))) double lds_f64;
CRN_DEV unsigned lds_offset(const void *p) {
  return (unsigned)(unsigned long long)(__attribute__((address_space(3))) const void *)p;
}

// Sum over a team of TEAM consecutive lanes (32 or 64), every lane gets the total: butterflies on
// the DPP path (quad_perm xor 1, xor 2, row_half_mirror, row_mirror) up to 16-lane rows, then the
// four row sums come back through v_readlane.  No LDS-pipe shuffles.
template <int CTRL>
CRN_DEV float dpp_add(float v) {
  return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
template <int TEAM>
CRN_DEV float team_sum(float v, int tid) {
  static_assert(TEAM == 32 || TEAM == 64, "team is half a wave or a wave");
  v = dpp_add<0xB1>(v);   // quad_perm [1,0,3,2]
  v = dpp_add<0x4E>(v);   // quad_perm [2,3,0,1]
  v = dpp_add<0x141>(v);  // row_half_mirror
  v = dpp_add<0x140>(v);  // row_mirror
  const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
  const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
  const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
  const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
  if constexpr (TEAM == 64) return (r0 + r1) + (r2 + r3);
  else return (tid & 32) ? r2 + r3 : r0 + r1;
}

// W consecutive table words through the scalar cache into SGPRs (the epoch close must not touch
// vmcnt, and an LDS read queues behind the other waves' exchange traffic): issue with s_load_row,
// then s_wait_row before the first use.
template <int W> struct SWords;
template <> struct SWords<2> { typedef int T __attribute__((ext_vector_type(2))); };
template <> struct SWords<4> { typedef int T __attribute__((ext_vector_type(4))); };
template <> struct SWords<8> { typedef int T __attribute__((ext_vector_type(8))); };
template <> struct SWords<16> { typedef int T __attribute__((ext_vector_type(16))); };
template <int W, int BYTE_OFF>
CRN_DEV typename SWords<W>::T s_load_row(const int *base) {
  typename SWords<W>::T r;
  if constexpr (W == 2) asm volatile("s_load_dwordx2 %0, %1, %2" : "=s"(r) : "s"(base), "n"(BYTE_OFF));
  if constexpr (W == 4) asm volatile("s_load_dwordx4 %0, %1, %2" : "=s"(r) : "s"(base), "n"(BYTE_OFF));
  if constexpr (W == 8) asm volatile("s_load_dwordx8 %0, %1, %2" : "=s"(r) : "s"(base), "n"(BYTE_OFF));
  if constexpr (W == 16) asm volatile("s_load_dwordx16 %0, %1, %2" : "=s"(r) : "s"(base), "n"(BYTE_OFF));
  return r;
}
template <class V>
CRN_DEV void s_wait_row(V &r) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(r));
}

// The reference's 4-5-3 sigmoid network and cascade (CE_Predictive_Node.cpp:200-261), spread over the
// lanes of a team: hidden unit j on lane j, output k on lane k, values passed with v_readlane.  Each
// unit's sum is formed by one lane in the reference's order, so the results are those of the serial
// loop; what changes is the latency — two exp() in sequence instead of eight (one lane doing all of
// it cost the reference-mode kernel 3.8 %).  Weights come from the LDS copy of the table (a per-lane
// global load would wait on vmcnt behind the prefetch).  Every lane of the team must call this.
template <int TEAM>
CRN_DEV double lane_f64(double v, int src, int half) {
  const int lo = __builtin_amdgcn_readlane((int)(__double_as_longlong(v) & 0xffffffffll), src);
  const int hi = __builtin_amdgcn_readlane((int)(__double_as_longlong(v) >> 32), src);
  double r = __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
  if constexpr (TEAM == 32) {  // two groups share a wave: the upper one reads lanes 32 + src
    const int lo2 = __builtin_amdgcn_readlane((int)(__double_as_longlong(v) & 0xffffffffll), 32 + src);
    const int hi2 = __builtin_amdgcn_readlane((int)(__double_as_longlong(v) >> 32), 32 + src);
    const double r2 = __longlong_as_double(((long long)hi2 << 32) | (unsigned int)lo2);
    r = half ? r2 : r;
  }
  return r;
}

template <int TEAM>
CRN_DEV void ann_decide_team(const SenseParams &p, const lds_f64 *w_ih, const lds_f64 *w_ho, long long epoch,
                             bool store, int lane, int half, float nf, float ch1, float ch2, float ch3) {
  // .cpp:200: Features_Buffer = {0, NOISE_FLOOR, CH1, CH2, CH3} widened to double
  const double f1 = (double)nf, f2 = (double)ch1, f3 = (double)ch2, f4 = (double)ch3;
  const int j = (lane >= 1 && lane <= 5) ? lane : 1;  // .cpp:214-220, unit j
  double s = w_ih[0 * 6 + j];
  s += f1 * w_ih[1 * 6 + j];
  s += f2 * w_ih[2 * 6 + j];
  s += f3 * w_ih[3 * 6 + j];
  s += f4 * w_ih[4 * 6 + j];
  const double hj = 1.0 / (1.0 + exp(-s));
  double hid[6];
#pragma unroll
  for (int q = 1; q <= 5; q++) hid[q] = lane_f64<TEAM>(hj, q, half);
  const int k = (lane >= 1 && lane <= 3) ? lane : 1;  // .cpp:229-235, output k
  double so = w_ho[0 * 4 + k];
#pragma unroll
  for (int q = 1; q <= 5; q++) so += hid[q] * w_ho[q * 4 + k];
  const double ok = 1.0 / (1.0 + exp(-so));
  const double o1 = lane_f64<TEAM>(ok, 1, half), o2 = lane_f64<TEAM>(ok, 2, half), o3 = lane_f64<TEAM>(ok, 3, half);
  // .cpp:245-261 cascade
  int d = 0;
  if (o1 >= p.ann_threshold) d = 1;
  else if (o2 >= p.ann_threshold) d = 2;
  else if (o3 >= p.ann_threshold) d = 3;
  if (store) {
    if (lane >= 1 && lane <= 3 && p.ann_out != nullptr) p.ann_out[epoch * 3 + (lane - 1)] = ok;
    if (lane == 0 && p.decision != nullptr) p.decision[epoch] = d;
    if (lane < p.n_bands && p.occupancy != nullptr) p.occupancy[epoch * p.n_bands + lane] = (uint8_t)(lane >= 1 && lane == d);
  }
}

template <class C>
CRN_DEV void epoch_close(FrameCtx<C> &c, const SenseParams &p, long long epoch_base) {
  constexpr int R3 = C::R3;
  constexpr bool MAG = C::MAG;
  using G = Geo<R3>;
  constexpr int T = G::T, N = G::N, J = G::J;
  float (&acc)[16] = c.acc;
  const float Kf = c.Kf;
  // Everything this block needs is re-derived here from uniform values (SGPRs) and the hardware
  // lane id, so that nothing but the accumulators stays live in VGPRs across the frame loop for a
  // block that runs once per K frames: what the allocator kept for it, it spilled, and a scratch
  // reload waits on vmcnt behind the next frame's prefetch.
  // latency-bound stretch with nothing of this wave's in flight behind it, and the workgroup's other
  // waves waiting at its barriers: outrank the butterflies (+0.6 % at N = 4096, +1.3 % at 2048; the
  // barrier-free sizes lose 0.5 % with it)
  if constexpr ((C::OPT & kPrioValu) != 0 && G::XWAVE) __builtin_amdgcn_s_setprio(3);
  const int tid = c.wave * 64 + (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  const int t = tid % T, grp = tid / T;
  const long long epoch = epoch_base + (long long)grp * c.grp_epoch_stride;
  const bool active = epoch < p.n_epochs;
  const int a = t / R3, m_lo = t % R3;
  CloseTrace<C> trace;
  trace.enter(p, epoch, active, t, T);
  // Energy mode: the division by K is applied to the band sums (and to the per-bin values only
  // when a spectrum is requested) — sixteen IEEE divides per thread per epoch were half a frame's
  // worth of VALU work.  Sum-then-divide differs from the reference order (divide-then-sum) by
  // rounding only.
  // The opaque values are 32-bit LDS offsets, not generic pointers: through a generic pointer every
  // access below became a FLAT instruction followed by s_waitcnt vmcnt(0), which also drained the
  // next frame's prefetch at every epoch close.
  const unsigned tab_off = c.lds_base + (unsigned)((G::GROUPS * C::NBUF * G::GROUP_CPLX + 16 * R3) * sizeof(cx));
  const unsigned gb_off = c.lds_base + (unsigned)grp * (unsigned)(C::NBUF * G::GROUP_CPLX * sizeof(cx));
  const lds_i32 *tab = reinterpret_cast<const lds_i32 *>(tab_off);
  const lds_f32 *thr = reinterpret_cast<const lds_f32 *>(tab_off + 416 * 4);
  const lds_f64 *w_ih = reinterpret_cast<const lds_f64 *>(tab_off + 544 * 4);  // [5][6]
  const lds_f64 *w_ho = reinterpret_cast<const lds_f64 *>(tab_off + 604 * 4);  // [6][4]
  lds_f32 *spec = reinterpret_cast<lds_f32 *>(gb_off);    // N + N/16 floats
  lds_f32 *featl = spec + spec_phys(N);                   // CRN_MAX_BANDS floats (LDS path)
  constexpr int TEAM = G::TEAM;
  constexpr int TPG = T / TEAM;  // teams (waves) per group
  const int lane = t % TEAM;
  lds_f32 *part = reinterpret_cast<lds_f32 *>(tab_off + kBandTabWords * 4);  // [256 / TEAM][16]
  [[maybe_unused]] const lds_f32 *feat = nullptr;
  // Three forms of the close, chosen per launch (one kernel holding several spilled in the frame loop):
  if constexpr ((C::OPT & kAlignedBands) != 0) {
    // Equal contiguous bands of W = 2^sh bins, sh = 6..8 (the Welch scan's 64 channels of 64 bins), N = 4096:
    // thread (a, m_lo) holds bins 256 d + 16 m_lo + a in acc[d], so band (256 d + 16 m_lo) >> sh is the sum over
    // all 16 a and over a group of G = W / 16 consecutive m_lo — lanes of one DPP row.  Group sums by DPP
    // (no LDS), the four rows of a wave through the wave's OWN exchange rows (only x1 writes of other waves
    // ever touch them, and those sit between the frame's two barriers), one barrier, then one lane per band adds
    // the 16 values of a: 2-4 DPP adds per register, <= 16 narrow LDS writes and one barrier instead of a
    // spectrum image, three barriers and a table walk (5 % of the Welch stream at K = 8).
    static_assert(R3 == 16 && !MAG, "aligned-band close: N = 4096, energy mode");
    const int sh = p.aligned_shift;   // uniform
    const int G = 1 << (sh - 4), nb = p.n_bands, al = (tid >> 4) & 3, r = m_lo & (G - 1), grp_b = m_lo >> (sh - 4);
    constexpr int kStride = 72;       // floats per a-row of partials: 72 mod 32 = 8 keeps a wave's rows on distinct banks
    lds_f32 *mine = reinterpret_cast<lds_f32 *>(c.lds_base + (unsigned)(4 * c.wave * Geo<R3>::ROW * sizeof(cx))) + al * kStride;
    const float thr_lane = thr[tid & 63];
#pragma unroll
    for (int d = 0; d < 16; d++) {
      float v = acc[d];
      acc[d] = 0.f;  // .cpp:287
      v = dpp_add<0xB1>(v);
      v = dpp_add<0x4E>(v);
      if (sh >= 7) v = dpp_add<0x141>(v);
      if (sh >= 8) v = dpp_add<0x140>(v);
      if ((d & (G - 1)) == r) mine[(d << (8 - sh)) + grp_b] = v;   // one lane of the group stores the group's sum
    }
    trace.stamp1();
    __syncthreads();
    trace.stamp2();
    if (c.wave == 0) {
      const int b = tid;  // one lane per band (n_bands <= 64)
      float sum = 0.f;
      if (b < nb) {
#pragma unroll
        for (int w4 = 0; w4 < 4; w4++) {
          const lds_f32 *src = reinterpret_cast<const lds_f32 *>(c.lds_base + (unsigned)(4 * w4 * Geo<R3>::ROW * sizeof(cx)));
          const float a0 = src[b], a1 = src[kStride + b], a2 = src[2 * kStride + b], a3 = src[3 * kStride + b];
          sum += a0;
          sum += a1;
          sum += a2;
          sum += a3;
        }
      }
      trace.stamp3();
      const float f = sc_unscale<C>(__fdiv_rn(sum, Kf), p);
      const bool in = b < nb;
      if (p.decide == CRN_DECIDE_THRESHOLD_K) {
        const float ref = p.ref_band >= 0 ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(f), p.ref_band)) : 1.0f;
        const bool occ = active && in && f > thr_lane * ref;
        if (active && in && p.occupancy != nullptr) p.occupancy[epoch * nb + b] = (uint8_t)occ;
        const unsigned long long m = __ballot(occ);
        if (active && b == 0 && p.decision != nullptr) p.decision[epoch] = __popcll(m);
      } else if (active) {
        if (b == 0 && p.decision != nullptr) p.decision[epoch] = 0;
        if (in && p.occupancy != nullptr) p.occupancy[epoch * nb + b] = 0;
      }
      if (active && in && p.features != nullptr) p.features[epoch * nb + b] = f;
    }
  } else if constexpr ((C::OPT & kRegBands) != 0) {
    // Band sums straight from the accumulator registers: no LDS image of the spectrum, no barrier
    // before it (nothing aliases the exchange buffers) and none after the decision.  Thread bins are
    // base_j + 256 d; the host cut the band plan at the 256-bin rows (crn_api.cpp), so each entry is
    // (row d, band, [lo, hi) in the row): masked add over j, DPP team sum, lane `band` keeps it.
    // LDS round trips are what this block avoids (an LDS read here queues behind the exchange
    // traffic of the CU's other waves: measured ~500 ticks each): the entries come through the
    // scalar cache, the features stay in lanes, the thresholds are fetched first and used last.
    constexpr int CAP = kRowEntryWords / R3;  // entry slots per row
    constexpr unsigned AM = acc_mask<C>();   // registers j R3 + d the kernel keeps (all of them unless kRows)
    constexpr auto row_live = [](int d) {     // some register of row d is kept
      for (int j = 0; j < J; j++)
        if ((acc_mask<C>() >> (j * R3 + d)) & 1u) return true;
      return false;
    };
    float thr_lane = 0.f;
    if constexpr (TPG == 1) thr_lane = thr[lane & 15];  // fetched first, used last
    // one row's entries at a time, the next row's load in flight meanwhile: holding all of them
    // costs SGPRs the frame loop needs (the spill lanes' VGPR pushed a loop address to scratch)
    constexpr auto next_live = [](int d) {
      for (int x = d + 1; x < R3; x++)
        for (int j = 0; j < J; j++)
          if ((acc_mask<C>() >> (j * R3 + x)) & 1u) return x;
      return (int)R3;
    };
    constexpr int kFirst = next_live(-1);
    float fsum = 0.f;
    typename SWords<CAP>::T ent_next = s_load_row<CAP, (512 + kFirst * CAP) * 4>(p.band_tab);
    static_for<R3>([&](auto dc) {
      constexpr int d = decltype(dc)::value;
      if constexpr (row_live(d)) {
        typename SWords<CAP>::T ent = ent_next;
        s_wait_row(ent);
        constexpr int dn = next_live(d);
        if constexpr (dn < R3) ent_next = s_load_row<CAP, (512 + dn * CAP) * 4>(p.band_tab);
#pragma unroll
        for (int e = 0; e < CAP; e++) {
          const int w = ent[e];
          if (w != 0) {  // uniform; 0 = unused slot
            const int band = w >> 18, lo = (w >> 9) & 511, span = (w & 511) - lo;
            float v = 0.f;
#pragma unroll
            for (int j = 0; j < J; j++) {
              if (((AM >> (j * R3 + d)) & 1u) == 0) continue;   // never accumulated: no band of the plan reaches it
              const int base = a + 16 * (m_lo * J + j);
              v += (unsigned)(base - lo) < (unsigned)span ? acc[j * R3 + d] : 0.f;
            }
            v = team_sum<TEAM>(v, tid);
            fsum += lane == band ? v : 0.f;
          }
        }
      }
    });
#pragma unroll
    for (int i = 0; i < 16; i++) acc[i] = 0.f;  // .cpp:287
    trace.stamp1();
    if constexpr (TPG > 1) {
      if (lane < 16) part[(tid / TEAM) * 16 + lane] = fsum;
      __syncthreads();
      trace.stamp2();
      if (t < TEAM && lane < 16) {
        thr_lane = thr[lane];  // same LDS round trip as the partials
        fsum = 0.f;
#pragma unroll
        for (int w = 0; w < TPG; w++) fsum += part[(grp * TPG + w) * 16 + lane];
      }
    }
    trace.stamp3();
    // the first team of the group stores and decides; lane b holds band b (n_bands <= 16)
    if (t < TEAM) {
      const float fs1 = MAG ? sc_unscale<C>(fsum, p) : fsum;
      const float f = MAG ? fs1 * fs1 : sc_unscale<C>(__fdiv_rn(fsum, Kf), p);  // .cpp:194-197
      const int half = TEAM == 32 ? (tid & 32) : 0;             // two groups share a wave at T = 32
      auto from_lane = [&](int b) {
        const float lo_half = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(f), b));
        if constexpr (TEAM == 32) {
          const float hi_half = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(f), 32 + b));
          return half ? hi_half : lo_half;
        }
        return lo_half;
      };
      const bool in = lane < p.n_bands;
      if (p.decide == CRN_DECIDE_ANN_K) {
        const float nf = from_lane(0), ch1 = from_lane(1), ch2 = from_lane(2), ch3 = from_lane(3);
        ann_decide_team<TEAM>(p, w_ih, w_ho, epoch, active, lane, half, nf, ch1, ch2, ch3);
      } else if (p.decide == CRN_DECIDE_THRESHOLD_K) {
        const float ref = p.ref_band >= 0 ? from_lane(p.ref_band) : 1.0f;
        const bool occ = active && in && f > thr_lane * ref;
        if (active && in && p.occupancy != nullptr) p.occupancy[epoch * p.n_bands + lane] = (uint8_t)occ;
        unsigned long long m = __ballot(occ);
        if constexpr (TEAM == 32) m = (m >> half) & 0xffffffffull;
        if (active && t == 0 && p.decision != nullptr) p.decision[epoch] = __popcll(m);
      } else if (active) {
        if (t == 0 && p.decision != nullptr) p.decision[epoch] = 0;
        if (in && p.occupancy != nullptr) p.occupancy[epoch * p.n_bands + lane] = 0;
      }
      if (active && in && p.features != nullptr) p.features[epoch * p.n_bands + lane] = f;
    }
  } else {
    // descriptor of this lane's first band (one lane per band below): fetched now, used after the barriers
    int pre_s0 = 0, pre_s1 = 0, pre_lo = 0, pre_hi = 0;
    if (t < p.n_bands) {
      pre_s0 = tab[t];
      pre_s1 = tab[t + 1];
      if (pre_s1 > pre_s0) {
        pre_lo = tab[96 + pre_s0];
        pre_hi = tab[256 + pre_s0];
      }
    }
    if constexpr (G::XWAVE) __syncthreads();
    else wave_sync();
#pragma unroll
    for (int j = 0; j < J; j++)
#pragma unroll
      for (int d = 0; d < R3; d++) {
        // the pruned kernels never accumulate (or read back) the other registers
        if (((acc_mask<C>() >> (j * R3 + d)) & 1u) == 0) continue;
        const int k = a + 16 * (m_lo * J + j) + 256 * d;
        spec[spec_phys(k)] = acc[j * R3 + d];
      }
#pragma unroll
    for (int i = 0; i < 16; i++) acc[i] = 0.f;  // .cpp:287
    if constexpr (G::XWAVE) __syncthreads();
    else wave_sync();

    trace.stamp1();  // LDS form: spectrum image visible
    if (p.spectrum != nullptr && active) {
      float *dst = p.spectrum + epoch * N;
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const float x = spec[spec_phys(t + T * r)];
        dst[t + T * r] = sc_unscale<C>(MAG ? x : __fdiv_rn(x, Kf), p);
      }
    }

    // band sums (reference .cpp:173-191).  This stretch is pure latency (the wave has no loads in
    // flight beyond its prefetched frame), every LDS round trip queues behind the CU's exchange
    // traffic, and a dependent VALU chain issues one instruction per ~10 cycles — so the work is
    // spread over lanes instead of bands (one team per band, 16 bands in sequence per wave, cost
    // the 64-band Welch kernel 11 us per epoch, 40 % of the wave's time):
    //   every thread sums the 16 consecutive bins of "its" block (thread t: bins 16 t .. 16 t + 15,
    //   ascending like the reference) -> blk[t]; the 16 lanes of a DPP row add their block totals
    //   -> rows[t / 16] (256 bins);  then ONE LANE PER BAND walks its segments in steps of 256, 16
    //   and 1 bins (at most 15 + 15 + 16 + 15 + 15 reads), all bands at once.
    lds_f32 *blk = featl + 80;  // [T] block totals (behind the CRN_MAX_BANDS features)
    lds_f32 *rows = blk + T;                 // [R3] row totals
    {
      float bs = 0.f;
#pragma unroll
      for (int j = 0; j < 16; j++) bs += spec[17 * t + j];  // spec_phys(16 t + j)
      blk[t] = bs;
      float rs = dpp_add<0xB1>(bs);   // the 16 block totals of a 256-bin row sit in one DPP row
      rs = dpp_add<0x4E>(rs);
      rs = dpp_add<0x141>(rs);
      rs = dpp_add<0x140>(rs);
      if ((t & 15) == 0) rows[t >> 4] = rs;
    }
    if constexpr (G::XWAVE) __syncthreads();
    else wave_sync();
    // four block (or row) totals per LDS round trip, added in ascending order
    auto add_run = [&](float &sum, const lds_f32 *tot, int i, int n) {  // tot[i .. i + n)
      for (; n >= 4; n -= 4, i += 4) {
        const float a0 = tot[i], a1 = tot[i + 1], a2 = tot[i + 2], a3 = tot[i + 3];
        sum += a0;
        sum += a1;
        sum += a2;
        sum += a3;
      }
      for (; n > 0; n--, i++) sum += tot[i];
    };
    for (int b = t; b < p.n_bands; b += T) {
      float sum = 0.f;
      const bool first = b == t;  // this lane's first band: descriptor fetched before the barriers
      const int s0 = first ? pre_s0 : tab[b], s1 = first ? pre_s1 : tab[b + 1];
      for (int sg = s0; sg < s1; sg++) {
        int k = (first && sg == s0) ? pre_lo : tab[96 + sg];
        const int hi = (first && sg == s0) ? pre_hi : tab[256 + sg];
        while (k < hi && (k & 15) != 0) sum += spec[spec_phys(k++)];
        if (k + 16 <= hi) {
          // whole blocks up to the next row boundary, whole rows, whole blocks after them
          int n = ((hi - k) >> 4);                       // whole blocks available
          const int to_row = ((256 - (k & 255)) & 255) >> 4;  // blocks until k is row-aligned
          const int head = n < to_row ? n : to_row;
          add_run(sum, blk, k >> 4, head);
          k += head * 16;
          n -= head;
          const int nrows = n >> 4;
          add_run(sum, rows, k >> 8, nrows);
          k += nrows * 256;
          n -= nrows * 16;
          add_run(sum, blk, k >> 4, n);
          k += n * 16;
        }
        while (k < hi) sum += spec[spec_phys(k++)];
      }
      const float msum = MAG ? sc_unscale<C>(sum, p) : sum;
      featl[b] = MAG ? msum * msum : sc_unscale<C>(__fdiv_rn(sum, Kf), p);  // .cpp:194-197
    }
    trace.stamp2();  // LDS form: this wave's band sums done
    if constexpr (G::XWAVE) __syncthreads();
    else wave_sync();
    trace.stamp3();  // LDS form: every feature written

    feat = featl;

    // LDS path: the first team of the group stores and decides.
    if (active && t < TEAM) {
      if (p.features != nullptr)
        for (int b = t; b < p.n_bands; b += TEAM) p.features[epoch * p.n_bands + b] = feat[b];

      if (p.decide == CRN_DECIDE_ANN_K) {
        ann_decide_team<TEAM>(p, w_ih, w_ho, epoch, true, lane, TEAM == 32 ? (tid & 32) : 0, feat[0], feat[1], feat[2], feat[3]);
      } else if (p.decide == CRN_DECIDE_THRESHOLD_K) {
        // lane i takes bands i, i + TEAM, ...; the count of occupied bands is a ballot, not a serial walk
        {
          const float ref = p.ref_band >= 0 ? feat[p.ref_band] : 1.0f;
          int cnt = 0;
          for (int b0 = 0; b0 < p.n_bands; b0 += TEAM) {
            const int b = b0 + t;
            const bool in = b < p.n_bands;
            const bool occ = in && feat[in ? b : 0] > thr[in ? b : 0] * ref;
            if (in && p.occupancy != nullptr) p.occupancy[epoch * p.n_bands + b] = (uint8_t)occ;
            unsigned long long m = __ballot(occ);
            if constexpr (TEAM == 32) m = (m >> (tid & 32)) & 0xffffffffull;
            cnt += __popcll(m);
          }
          if (t == 0 && p.decision != nullptr) p.decision[epoch] = cnt;
        }
      } else {
        if (t == 0 && p.decision != nullptr) p.decision[epoch] = 0;
        if (p.occupancy != nullptr)
          for (int b = t; b < p.n_bands; b += TEAM) p.occupancy[epoch * p.n_bands + b] = 0;
      }
    }
  }
  // With one exchange buffer the next epoch's first frame syncs the workgroup before it writes
  // exchange 1 (frame_compute), which is after every wave has passed this point: no barrier here.
  if constexpr (G::XWAVE && C::NBUF == 2) __syncthreads();
  if constexpr (!G::XWAVE) wave_sync();
  trace.leave(p, epoch, active, t);
  if constexpr ((C::OPT & kPrioValu) != 0 && G::XWAVE) __builtin_amdgcn_s_setprio(0);
}

}  // namespace crn
#endif

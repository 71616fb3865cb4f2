// crn_sense_kernel.h — the sensing kernel itself (the frame loop in its streaming forms) and its launch helpers.
#ifndef CRN_SENSE_KERNEL_H
#define CRN_SENSE_KERNEL_H
#include <atomic>

#include "crn_epoch_close.h"

namespace crn {
// Buffer resource over the IQ window of epoch group `eg` (GROUPS consecutive epochs): anything
// past the window, or past the end of the batch, reads as zero.
template <int R3, int SB = 8>
CRN_DEV __amdgpu_buffer_rsrc_t group_rsrc(const SenseParams &p, long long eg, int span = 1) {
  using G = Geo<R3>;
  const long long first = eg * G::GROUPS * p.epoch_stride;
  long long left = (p.total_samples - first) * SB;
  const long long window = ((long long)span * G::GROUPS * p.epoch_stride + (long long)p.K * p.frame_stride + G::N) * SB;
  if (left > window) left = window;
  if (left < 0) left = 0;
  char *base = reinterpret_cast<char *>(const_cast<float2 *>(p.iq)) + first * SB;   // p.iq is int16 pairs when SB == 4
  return __builtin_amdgcn_make_buffer_rsrc(base, 0, (int)left, 0x00020000);
}

// Epoch groups [g0, g0 + n_local) of a streaming workgroup (graded launch: launch_cfg).
struct StreamSpan {
  long long g0;
  int epw, n_local;
};
template <int R3>
CRN_DEV StreamSpan stream_span(const SenseParams &p) {
  using G = Geo<R3>;
  const long long n_groups = (p.n_epochs + G::GROUPS - 1) / G::GROUPS;
  // two tiers in dispatch order: big spans, then the short tail
  const long long b = (long long)blockIdx.x;
  const bool big = b < p.n_big_wgs;
  StreamSpan s;
  s.epw = big ? p.groups_per_wg : p.tail_groups_per_wg;
  s.g0 = big ? b * p.groups_per_wg : p.n_big_wgs * p.groups_per_wg + (b - p.n_big_wgs) * p.tail_groups_per_wg;
  s.n_local = (int)((n_groups - s.g0) < s.epw ? (n_groups - s.g0) : s.epw);
  return s;
}

// ---------------------------------------------------------------------------------------------
// the sensing kernel
// ---------------------------------------------------------------------------------------------
template <class C>
__global__ __launch_bounds__(256, C::OCC) void sense_kernel(const SenseParams p) {
  constexpr int R3 = C::R3, NBUF = C::NBUF;
  constexpr bool NT = C::NT, SC = C::SC16;
  constexpr unsigned SB = C::SB;
  using G = Geo<R3>;
  constexpr int T = G::T;
  extern __shared__ __attribute__((aligned(16))) cx lds[];

  const int tid = threadIdx.x;
  const int grp = tid / T;
  const int t = tid % T;       // pass-1 column, n_lo
  const int a = t / R3;        // pass-2/3 sub-transform id (k mod 16)
  const int m_lo = t % R3;     // pass-2 column / pass-3 slot g

  FrameCtx<C> c;
  c.t = t;
  c.a = a;
  c.m_lo = m_lo;
  c.L = p.L;
  c.gbuf = lds + grp * (NBUF * G::GROUP_CPLX);
  c.wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  c.grp_epoch_stride = 1;
  c.lds_base = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_offset(lds));
  c.tw2_lds = lds + G::GROUPS * NBUF * G::GROUP_CPLX;  // [16][R3], TW2LDS only
  CloseTrace<C>::workgroup_start(p, tid);   // (measurement build only: crn_frame_ab.h)
  const int K = p.K;
  c.Kf = (float)K;
  c.invK = 1.0f / (float)K;

  constexpr unsigned kNowhere = 0x80000000u;  // scalar offset past every window: dropped by the range check
  const unsigned voff = (unsigned)(grp * (unsigned)p.epoch_stride + t) * SB;
  const unsigned fbytes = (unsigned)p.frame_stride * SB;
  cx ua[16], ub[16];
  // A plain streaming workgroup asks for its first frame before anything else: twiddles and tables (L2 hits, but queued behind the
  // CU's streaming loads: ~4 us a round trip on a full machine) then arrive with it instead of ahead of it.  Measured from inside
  // the kernel (tools/gpu_wg_placement.py): a workgroup's first epoch took 26 us longer than its later ones.
  static_assert((C::OPT & kMulti) != 0 && C::PREFETCH, "the kernels are streaming workgroups with the prefetch spread through the butterflies");
  constexpr bool kEarlyLoad = !C::WIN;
  if constexpr (kEarlyLoad) {
    const StreamSpan sp0 = stream_span<R3>(p);
    load_frame<R3, NT, SC>(ua, group_rsrc<R3, (int)SB>(p, sp0.g0, sp0.epw), voff, 0u, C::FULL ? G::N : c.L);
  }

  // frame-invariant twiddles, kept in registers across frames and epochs
#pragma unroll
  for (int i = 1; i < ((C::OPT & kTw1C) != 0 ? 9 : 16); i++) c.tw1[i] = reinterpret_cast<const cx *>(p.tw1)[i * T + t];
  if constexpr ((C::OPT & kTw1C) != 0) c.tw1[0] = reinterpret_cast<const cx *>(p.tw1)[16 * T + t];  // W_N^{16 t}
  // (register-resident tables are requested ahead of the barrier too: nothing after it starts another round trip)
  if constexpr (!C::TW2LDS) {
#pragma unroll
    for (int i = 1; i < 16; i++) c.tw2[i] = reinterpret_cast<const cx *>(p.tw2)[i * R3 + m_lo];
  }
  if constexpr (C::WIN && (C::OPT & kHannSym) != 0) {
#pragma unroll
    for (int q = 0; q < 4; q++) c.winp[q] = cx{p.window[t + T * (2 * q)], p.window[t + T * (2 * q + 1)]};
  } else if constexpr (C::WIN) {
#pragma unroll
    for (int r = 0; r < 16; r++) c.win[r] = p.window[t + T * r];
  }
  {
    // band table -> LDS (2 KiB behind the exchange buffers and the tw2 table): the epoch close walks
    // it, and from global memory every walk step was a dependent ~1 us vector load.  Every load of the prologue is issued before the
    // first LDS write (unconditional loads, clamped indices): one wait for all of them instead of three round trips in a row.
    int *tab = reinterpret_cast<int *>(lds + G::GROUPS * NBUF * G::GROUP_CPLX + 16 * R3);
    const int w0 = p.band_tab[tid], w1 = p.band_tab[tid + 256];
    const int w2 = p.band_tab[tid < kBandTabWords - 512 ? tid + 512 : kBandTabWords - 1];  // row entries
    [[maybe_unused]] cx tw2v = cx{0.f, 0.f};
    if constexpr (C::TW2LDS) tw2v = reinterpret_cast<const cx *>(p.tw2)[tid < 16 * R3 ? tid : 16 * R3 - 1];
    tab[tid] = w0;
    tab[tid + 256] = w1;
    if (tid < kBandTabWords - 512) tab[tid + 512] = w2;
    if constexpr (C::TW2LDS) {
      if (tid < 16 * R3) lds[G::GROUPS * NBUF * G::GROUP_CPLX + tid] = tw2v;
    }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 16; i++) c.acc[i] = 0.f;

  if constexpr (C::WIN) {
    if (p.frame_stride * 2 == G::N && p.epoch_stride == (long long)K * (G::N / 2)) {
      // Welch (hop = N/2) over dense epochs: a lane group's epochs are one uninterrupted stream of
      // half-frames H(g) = samples [g N/2, (g+1) N/2) — frame g = H(g) | H(g+1), and the half an epoch
      // ends with is the half the next one starts with.  Three half-frame register sets: two hold the
      // current frame's raw samples, the third receives H(g+2) while frame g is computed, so every
      // sample is fetched once per lane group and the prefetch runs across epoch boundaries; the
      // close fires after every K-th frame.  The workgroup's GROUPS x epw epochs are dealt to its
      // lane groups in runs of epw (group g: epochs E0 + g epw ...), so that each group's stream is
      // contiguous; every group runs epw x K frames (the ragged end closes inactive epochs: uniform
      // barriers, loads past the batch return zero).
      constexpr unsigned hbytes = (unsigned)(G::N / 2) * SB;
      const StreamSpan sp = stream_span<R3>(p);
      const __amdgpu_buffer_rsrc_t rs = group_rsrc<R3, (int)SB>(p, sp.g0, sp.epw);
      const unsigned voff = (unsigned)(grp * sp.epw * (unsigned)p.epoch_stride + t) * SB;  // shadows the per-epoch one
      c.grp_epoch_stride = sp.epw;
      load_frame<R3, NT, SC>(ua, rs, voff, 0u);
      // Three half-frame sets whose roles rotate (current low half, current high half, incoming):
      // the loop is unrolled by three so the rotation is a renaming, not 16 register moves a frame.
      cx ha[16], hb[16], hc[16];  // only [0, 8) of each is used (frame_compute's prefetch target is a cx[16])
#pragma unroll
      for (int r = 0; r < 8; r++) {
        ha[r] = ua[r];
        hb[r] = ua[8 + r];
      }
      const int F = (G::GROUPS == 1 ? sp.n_local : sp.epw) * K;
      int f = 0, j = 0, g = 0;
#define CRN_WELCH_STEP(LO, HI, IN)                                                                  \
      {                                                                                             \
        _Pragma("unroll") for (int r = 0; r < 8; r++) {                                             \
          ub[r] = LO[r];                                                                            \
          ub[8 + r] = HI[r];                                                                        \
        }                                                                                           \
        frame_compute<C, true>(ub, c, f, IN, rs, voff, g + 1 < F ? (unsigned)(g + 2) * hbytes : kNowhere); \
        if (++f == K) {                                                                             \
          f = 0;                                                                                    \
          epoch_close<C>(c, p, sp.g0 * G::GROUPS + j);                                              \
          j++;                                                                                      \
        }                                                                                           \
        g++;                                                                                        \
      }
      while (g < F) {
        CRN_WELCH_STEP(ha, hb, hc)
        if (g >= F) break;
        CRN_WELCH_STEP(hb, hc, ha)
        if (g >= F) break;
        CRN_WELCH_STEP(hc, ha, hb)
      }
#undef CRN_WELCH_STEP
      return;
    }
  }
  if constexpr (C::WIN) {
    // (windowed kernels ask for their first frame after the tables; Welch over epochs with gaps between them runs one epoch group
    // per workgroup)
    const long long epoch_base = (long long)blockIdx.x * G::GROUPS;
    const __amdgpu_buffer_rsrc_t rsrc = group_rsrc<R3, (int)SB>(p, blockIdx.x);
    load_frame<R3, NT, SC>(ua, rsrc, voff, 0u, C::FULL ? G::N : c.L);
    if (p.frame_stride * 2 == G::N) {
      // Welch, hop = N/2: frame f = halves H(f) | H(f+1) with H(j) = samples [j N/2, (j+1) N/2).
      // Three half-frame register sets: two hold the current frame's raw samples, the third
      // receives H(f+2) while frame f is computed, so every sample is fetched from HBM once per
      // epoch.  (ua was loaded as a whole frame above: its two halves are H(0) and H(1).)
      constexpr unsigned hbytes = (unsigned)(G::N / 2) * SB;
      cx h0[8], h1[8], hn[16];
#pragma unroll
      for (int r = 0; r < 8; r++) {
        h0[r] = ua[r];
        h1[r] = ua[8 + r];
      }
      for (int f = 0; f < K; f++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
          ub[r] = h0[r];
          ub[8 + r] = h1[r];
        }
        // H(f+2) is fetched from inside frame f's first pass, one load per radix-4 group
        frame_compute<C, true>(ub, c, f, hn, rsrc, voff, f + 1 < K ? (unsigned)(f + 2) * hbytes : kNowhere);
#pragma unroll
        for (int r = 0; r < 8; r++) {
          h0[r] = h1[r];
          h1[r] = hn[r];
        }
      }
      epoch_close<C>(c, p, epoch_base);
      return;
    }
  }
  // This workgroup owns p.groups_per_wg consecutive epoch groups and treats their frames as
  // one stream: twiddles are loaded once, and the first frame of the next epoch is already in
  // flight while the last frame of this one is computed and closed (the per-workgroup prologue
  // and the exposed first load cost ~6 % at one epoch per workgroup).  Two register sets in
  // ping-pong; frame f+1's loads are issued from inside frame f's butterflies.
  // Workgroups are dispatched in blockIdx order; the last ones take a single epoch group, so the
  // machine drains in steps of one epoch instead of one 4-epoch workgroup (measured with
  // s_memrealtime stamps: the last 1024 workgroups used to finish spread over 200 us of a
  // 1.4 ms kernel).
  const StreamSpan sp = stream_span<R3>(p);
  const int epw = sp.epw, n_local = sp.n_local;
  const long long g0 = sp.g0;
  const __amdgpu_buffer_rsrc_t rs = group_rsrc<R3, (int)SB>(p, g0, epw);
  const unsigned gbytes = (unsigned)(G::GROUPS * (unsigned)p.epoch_stride) * SB;
  if constexpr (!kEarlyLoad) load_frame<R3, NT, SC>(ua, rs, voff, 0u, C::FULL ? G::N : c.L);   // windowed kernels: after the tables
  int j = 0, f = 0;
#define CRN_STREAM_STEP(CUR, NXT)                                                                 \
  {                                                                                               \
    const bool last = f + 1 == K;                                                                 \
    const int j_n = last ? j + 1 : j;                                                             \
    const int f_n = last ? 0 : f + 1;                                                             \
    const unsigned soff_n = j_n < n_local ? (unsigned)j_n * gbytes + (unsigned)f_n * fbytes : kNowhere; \
    frame_compute<C>(CUR, c, f, NXT, rs, voff, soff_n);                                           \
    if (last) epoch_close<C>(c, p, (g0 + j) * G::GROUPS);                                         \
    j = j_n;                                                                                      \
    f = f_n;                                                                                      \
  }
  while (true) {
    CRN_STREAM_STEP(ua, ub)
    if (j >= n_local) break;
    CRN_STREAM_STEP(ub, ua)
    if (j >= n_local) break;
  }
#undef CRN_STREAM_STEP
}

// ---------------------------------------------------------------------------------------------
// the sensing kernel for launches of a few epochs (the engine's shape: ONE epoch of ten 512-point frames, reference
// CE_Predictive_Node.cpp:148-156 run once per sensing period)
// ---------------------------------------------------------------------------------------------
// sense_kernel gives an epoch to one lane group and runs its K frames one after the other — the right shape when thousands of epochs
// fill the machine, but a single epoch then occupies T of a workgroup's 256 threads for K dependent frame latencies while the other
// lane groups idle.  Here a workgroup takes ONE epoch and deals its frames to its GROUPS lane groups (group g: frames g, g + GROUPS,
// ...): ceil(K / GROUPS) frame latencies instead of K.  Every frame's per-bin values are parked in LDS (ph_pass3_park); after one
// barrier lane group 0 replays the accumulate from the slots in frame order — the same operations in the same order as the
// streaming kernel, so features, network outputs and decisions are bit for bit the same — and closes the epoch as usual.
// Sizes whose frames stay inside one wave (N <= 1024: no workgroup barrier inside a frame).  Overlapped frames (the Welch plans) are
// fetched whole by their lane group: a half-frame two groups share is read twice, from L2 the second time.
template <class C>
__global__ __launch_bounds__(256, 1) void sense_kernel_dealt(const SenseParams p) {
  constexpr int R3 = C::R3;
  constexpr bool NT = C::NT, SC = C::SC16;
  constexpr unsigned SB = C::SB;
  using G = Geo<R3>;
  constexpr int T = G::T;
  static_assert(!G::XWAVE && C::NBUF == 1 && !C::TW2LDS && (C::OPT & kDeal) != 0,
                "dealt frames: N <= 1024, twiddles in registers");
  extern __shared__ __attribute__((aligned(16))) cx lds[];

  const int tid = threadIdx.x;
  const int grp = tid / T;
  const int t = tid % T;

  FrameCtx<C> c;
  c.t = t;
  c.a = t / R3;
  c.m_lo = t % R3;
  c.L = p.L;
  c.gbuf = lds + grp * G::GROUP_CPLX;
  c.wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  c.lds_base = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_offset(lds));
  c.tw2_lds = lds + G::GROUPS * G::GROUP_CPLX;
  // only lane group 0 holds an epoch when the close runs: the others' epoch index lands past the batch (inactive)
  c.grp_epoch_stride = (int)p.n_epochs;
  const int K = p.K;
  c.Kf = (float)K;
  c.invK = 1.0f / (float)K;

  // this epoch's samples: anything past the epoch's last frame, or past the batch, reads as zero
  const long long first = (long long)blockIdx.x * p.epoch_stride;
  long long left = (p.total_samples - first) * SB;
  const long long window = ((long long)(K - 1) * p.frame_stride + G::N) * SB;
  if (left > window) left = window;
  if (left < 0) left = 0;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
      reinterpret_cast<char *>(const_cast<float2 *>(p.iq)) + first * SB, 0, (int)left, 0x00020000);
  const unsigned voff = (unsigned)t * SB;
  const unsigned fbytes = (unsigned)p.frame_stride * SB;
  constexpr unsigned kNowhere = 0x80000000u;
  const int rounds = p.deal_rounds;            // ceil(K / GROUPS): every lane group runs that many frames (uniform wave barriers;
                                               // a frame index >= K loads zeros into a slot the replay never reads)
  cx ua[16], ub[16];
  load_frame<R3, NT, SC>(ua, rs, voff, grp < K ? (unsigned)grp * fbytes : kNowhere, c.L);

#pragma unroll
  for (int i = 1; i < 16; i++) c.tw1[i] = reinterpret_cast<const cx *>(p.tw1)[i * T + t];
#pragma unroll
  for (int i = 1; i < 16; i++) c.tw2[i] = reinterpret_cast<const cx *>(p.tw2)[i * R3 + c.m_lo];
  if constexpr (C::WIN && (C::OPT & kHannSym) != 0) {
#pragma unroll
    for (int q = 0; q < 4; q++) c.winp[q] = cx{p.window[t + T * (2 * q)], p.window[t + T * (2 * q + 1)]};
  } else if constexpr (C::WIN) {
#pragma unroll
    for (int r = 0; r < 16; r++) c.win[r] = p.window[t + T * r];
  }
  {
    int *tab = reinterpret_cast<int *>(lds + G::GROUPS * G::GROUP_CPLX + 16 * R3);
    const int w0 = p.band_tab[tid], w1 = p.band_tab[tid + 256];
    const int w2 = p.band_tab[tid < kBandTabWords - 512 ? tid + 512 : kBandTabWords - 1];
    tab[tid] = w0;
    tab[tid + 256] = w1;
    if (tid < kBandTabWords - 512) tab[tid + 512] = w2;
  }
#pragma unroll
  for (int i = 0; i < 16; i++) c.acc[i] = 0.f;

  // frame slots behind everything the streaming kernel keeps in LDS: [rounds x GROUPS][16][T] values
  constexpr unsigned kSlotBytes = (unsigned)G::N * (C::MAG ? 4u : 8u);
  const unsigned park_base =
      c.lds_base + (unsigned)((G::GROUPS * G::GROUP_CPLX + 16 * R3) * sizeof(cx)) + (unsigned)kCloseLdsBytes;
  int r = 0;
#define CRN_DEAL_STEP(CUR, NXT)                                                                       \
  {                                                                                                   \
    const int f = r * G::GROUPS + grp, fn = f + G::GROUPS;                                            \
    c.park_off = park_base + (unsigned)f * kSlotBytes;                                                \
    frame_compute<C>(CUR, c, 0, NXT, rs, voff, (r + 1 < rounds && fn < K) ? (unsigned)fn * fbytes : kNowhere); \
    r++;                                                                                              \
  }
  while (true) {
    CRN_DEAL_STEP(ua, ub)
    if (r >= rounds) break;
    CRN_DEAL_STEP(ub, ua)
    if (r >= rounds) break;
  }
#undef CRN_DEAL_STEP
  __syncthreads();   // every frame of the epoch is parked (and the band table is in place)
  if (grp == 0) replay_parked<C>(c, park_base, K);
  epoch_close<C>(c, p, (long long)blockIdx.x);
}

template <class C>
static hipError_t launch_dealt_cfg(const SenseParams &p, hipStream_t stream) {
  using G = Geo<C::R3>;
  const size_t slot_bytes = (size_t)G::N * (C::MAG ? 4 : 8);
  const size_t lds = ((size_t)G::GROUPS * G::GROUP_CPLX + 16 * C::R3) * sizeof(cx) + kCloseLdsBytes +
                     (size_t)p.deal_rounds * G::GROUPS * slot_bytes;
  if (p.n_epochs <= 0) return hipSuccess;
  auto kfn = sense_kernel_dealt<C>;
  if (lds > 48 * 1024) {
    // once per kernel, device and size: this is the engine's launch, repeated every sensing period with the same K
    static std::atomic<size_t> allowed[64];
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::atomic<size_t> &a = allowed[dev & 63];
    if (a.load(std::memory_order_acquire) < lds) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) {   // a device (or partition) with less LDS than crn_sense_create was told: the caller launches the streaming form
        (void)hipGetLastError();
        return hipErrorLaunchOutOfResources;
      }
      a.store(lds, std::memory_order_release);
    }
  }
  hipLaunchKernelGGL(kfn, dim3((unsigned)p.n_epochs), dim3(256), lds, stream, p);
  return hipGetLastError();
}

// The windowed dealt-frame form: periodic Hann on whole frames in energy mode, riding in pass 1's first butterflies (kHannSym) exactly as
// the streaming dispatch picks it for the same launch, so that the arithmetic is the same bit for bit — what the engine's `-m welch` /
// `-m scan` launch.  (Other windows and |X| mode have no dealt form: sense_deal_rounds says 0 and the streaming kernel takes the launch.)
// Windowed kernels close through the LDS walk.
template <int R3, int OPT>
static hipError_t launch_dealt_win(const SenseParams &p, hipStream_t stream) {
  constexpr int kO = kSpread | kLdsBlk | kDeal | OPT;
  return launch_dealt_cfg<Cfg<R3, 1, true, false, false, true, false, 1, false, true, kO | kHannSym>>(p, stream);
}

// The register form of the epoch close applies to band plans the host could cut into row entries
// (crn_api.cpp) when no per-bin spectrum is stored.
static bool reg_bands(const SenseParams &p) { return p.n_row_entries > 0 && p.spectrum == nullptr; }
// ... and pass 3 / the accumulate keep only the reference channel plan's registers when every band bin sits in one of them.
template <int R3>
static bool ref_plan_rows(const SenseParams &p) { return reg_bands(p) && ref_acc_mask(R3) != 0xFFFFu && (p.acc_mask & ~ref_acc_mask(R3)) == 0; }
// Which close a launch gets is ONE rule for the streaming and the dealt-frame kernels (their outputs are bit-identical because they sum
// in the same order): the register close for |X| mode, and for energy mode on whole frames or on the reference plan; the LDS walk
// otherwise (energy mode, short packets, another small plan; every plan too big for row entries; every spectrum request).
template <int R3>
static bool register_close(const SenseParams &p, bool mag) { return reg_bands(p) && (mag || p.L == Geo<R3>::N || ref_plan_rows<R3>(p)); }

// Dealt-frame forms of a size: |X| or energy, band sums from registers or through the LDS walk; short frames are masked at run time.
template <int R3, int OPT>
static hipError_t launch_dealt(const SenseParams &p, bool mag, hipStream_t stream) {
  constexpr int kO = kSpread | kLdsBlk | kDeal | OPT;
  const bool regb = register_close<R3>(p, mag);
  if (mag) {
    if (regb) return launch_dealt_cfg<Cfg<R3, 1, true, false, true, false, false, 1, false, true, kO | kRegBands>>(p, stream);
    return launch_dealt_cfg<Cfg<R3, 1, true, false, true, false, false, 1, false, true, kO>>(p, stream);
  }
  if (regb) return launch_dealt_cfg<Cfg<R3, 1, true, false, false, false, false, 1, false, true, kO | kRegBands>>(p, stream);
  return launch_dealt_cfg<Cfg<R3, 1, true, false, false, false, false, 1, false, true, kO>>(p, stream);
}

// ---------------------------------------------------------------------------------------------
// launch dispatch
// ---------------------------------------------------------------------------------------------
template <class C>
static hipError_t launch_cfg(const SenseParams &p, hipStream_t stream) {
  using G = Geo<C::R3>;
  const long long n_groups = (p.n_epochs + G::GROUPS - 1) / G::GROUPS;
  // Welch (hop = N/2) streams when the epochs are dense (see sense_kernel)
  const bool welch = C::WIN && p.frame_stride * 2 == G::N;
  const bool welch_stream = welch && p.epoch_stride == (long long)p.K * (G::N / 2);
  const bool multi = (C::OPT & kMulti) != 0 && C::PREFETCH && (!welch || welch_stream);
  SenseParams q = p;
  unsigned grid;
  if (multi) {
    // n_big_wgs workgroups of groups_per_wg groups, then workgroups of tail_groups_per_wg over the remaining groups
    if (q.tail_groups_per_wg < 1) q.tail_groups_per_wg = 1;
    if (q.n_big_wgs * q.groups_per_wg > n_groups) q.n_big_wgs = n_groups / q.groups_per_wg;
    const long long rest = n_groups - q.n_big_wgs * q.groups_per_wg;
    grid = (unsigned)(q.n_big_wgs + (rest + q.tail_groups_per_wg - 1) / q.tail_groups_per_wg);
  } else {
    q.n_big_wgs = 0;
    q.tail_groups_per_wg = 1;
    grid = (unsigned)n_groups;
  }
  const size_t lds = ((size_t)G::GROUPS * C::NBUF * G::GROUP_CPLX + 16 * C::R3) * sizeof(cx) + kCloseLdsBytes;
  if (grid == 0) return hipSuccess;
  auto kfn = sense_kernel<C>;
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kfn),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(kfn, dim3(grid), dim3(256), lds, stream, q);
  return hipGetLastError();
}

// The forms every size has, by mode / window / packet length / close.  What is specialised is what BASELINE.json's configurations and
// the engine run: whole frames (FULL: no zero-padding mask) where the band plan is the reference's (kRows | kRegBands) or a small one in
// energy mode (kRegBands); everything else — table windows, |X| mode with another plan or a spectrum request, short packets with a
// small custom plan — runs ONE form that masks at run time (and closes through the LDS walk where the register close has no form).
//   WHICH        0 all, 1 unwindowed kernels only, 2 windowed only
//   PRUNE        the reference-plan forms exist (not in the wire-format unit)
//   ENERGY_FULL  the unwindowed energy-mode whole-frame forms belong to this call (false at N = 4096: launch_rn has them)
template <int R3, int NBUF, bool PREFETCH, bool NT, bool TW2LDS, int OCC, bool PK, int OPT = kSpread | kLdsBlk | kPrioValu | kMulti,
          int WHICH = 0, bool PRUNE = true, bool ENERGY_FULL = true>
static hipError_t launch_default(const SenseParams &p, bool mag, bool win, hipStream_t stream) {
  const bool full = p.L == Geo<R3>::N;
  const bool regb = register_close<R3>(p, mag);  // small band plan, no spectrum: band sums from registers (see register_close)
  // ... and when every band bin sits in a register the reference channel plan also uses (ref_acc_mask), pass 3 and the accumulate
  // keep only those registers: 7 of 16 at N = 512 (where the reference's |X| costs a square root per bin and frame), 12 / 11 / 7 at
  // 1024 / 2048 / 4096
  [[maybe_unused]] const bool prune = PRUNE && ref_plan_rows<R3>(p);
#define CRN_GO(MAGV, WINV, FULLV, EXTRA) return launch_cfg<Cfg<R3, NBUF, PREFETCH, NT, MAGV, WINV, TW2LDS, OCC, FULLV, PK, OPT | (EXTRA)>>(p, stream)
  if constexpr (WHICH != 1) {   // table windows: one form per mode
    if (win) { if (mag) CRN_GO(true, true, false, 0); else CRN_GO(false, true, false, 0); }
  }
  if constexpr (WHICH != 2) {
    if constexpr (PRUNE) {      // the reference channel plan: |X| and energy, whole frames and short packets
      if (prune) {
        if (mag) { if (full) CRN_GO(true, false, true, kRegBands | kRows); else CRN_GO(true, false, false, kRegBands | kRows); }
        if constexpr (ENERGY_FULL) { if (full) CRN_GO(false, false, true, kRegBands | kRows); }
        if (!full) CRN_GO(false, false, false, kRegBands | kRows);
      }
    }
    if (regb) {                 // another small plan: register close for |X| (any packet length) and for energy on whole frames
      if (mag) CRN_GO(true, false, false, kRegBands);
      if constexpr (ENERGY_FULL) { if (full) CRN_GO(false, false, true, kRegBands); }
      // (a unit without the reference-plan forms — the wire-format one — closes that plan's short packets from registers all the same:
      // the same sums in the same order as the float path's pruned form)
      if constexpr (!PRUNE) { if (!full) CRN_GO(false, false, false, kRegBands); }
    }
    if (mag) CRN_GO(true, false, false, 0);   // any plan, spectrum requests: the LDS walk
    if constexpr (ENERGY_FULL) { if (full) CRN_GO(false, false, true, 0); }
    if (!full) CRN_GO(false, false, false, 0);
  }
  return hipErrorInvalidValue;
#undef CRN_GO
}

// The plain 4096-point kernel's forms (energy mode, no window, L = N): the default and its unpruned form.
template <int R3, int NBUF, bool PREFETCH, bool NT, bool TW2LDS, int OCC, bool PK, int OPT = 0>
static hipError_t launch_rn(const SenseParams &p, bool, bool, hipStream_t stream) {
  return launch_cfg<Cfg<R3, NBUF, PREFETCH, NT, false, false, TW2LDS, OCC, true, PK, OPT>>(p, stream);
}


}  // namespace crn
#endif

// ecr_threads.cpp — TEST INFRASTRUCTURE.  The ECR's two workers as THREADS around the engine, with the reference's
// own locking (src/extensible_cognitive_radio.cpp):
//   rx worker  (:1299-1324)  per packet: if ce_sensing_flag -> lock CE_mutex, re-check, memcpy into
//                            ce_usrp_rx_buffer, CE_event = USRP_RX_SAMPS, cond_signal, unlock
//   CE worker  (:1775-1803)  loop: lock CE_mutex; timedwait(CE_execute_sig, now + ce_timeout_ms) -> on ETIMEDOUT
//                            CE_event = TIMEOUT; CE->execute(); unlock          (ce_timeout_ms = 0: it spins)
// so the engine runs exactly as in a CRTS node: execute() with CE_mutex held, the rx thread waiting on that
// mutex, signals lost whenever the CE thread is not inside timedwait (the engine sees SOME frames).
//
//   ecr_threads <iq.bin> <samples_per_packet> <packets_per_segment> <seconds> [ce args...]
// iq.bin holds 4 segments (idle, CH1, CH2, CH3 driven) of <packets_per_segment> packets; the "radio" replays the
// current segment's packets in a loop at the real packet rate (L / 13 Msps) and moves to the next segment every
// 0.25 s.  Printed: one line per decision with the segment(s) its 10 frames came from, then what the rx thread
// paid for the hand-off (time spent waiting for CE_mutex per packet) and how many packets got through; the
// distribution of execute()'s duration with its ten longest calls attributed (which event, which packet of the
// epoch, whether a decision came back in it), and beside it a CONTROL: two clock reads with nothing between them,
// taken in the same loop — what the operating system alone does to this thread.
#include <errno.h>
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/time.h>
#include <time.h>

#include <algorithm>
#include <deque>
#include <vector>

#include "CE_Predictive_Node_GPU.hpp"

#ifndef CRN_USE_REFERENCE_BASE
CognitiveEngine::CognitiveEngine() : ECR(NULL) {}
CognitiveEngine::~CognitiveEngine() {}
void CognitiveEngine::execute() {}
#endif

static double now_s() {
  struct timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

struct Shared {
  ExtensibleCognitiveRadio ecr;
  pthread_mutex_t CE_mutex;
  pthread_cond_t CE_execute_sig;
  volatile int running;
  float ce_timeout_ms;
  // "radio"
  std::vector<std::complex<float> > capture;
  int L, per_seg;
  double t_end;
  volatile int seg_of_buffer;            // segment of the packet now in ce_usrp_rx_buffer (written under CE_mutex)
  // measurements
  std::vector<float> rx_wait_us;
  long packets, forwarded;
};

static void *rx_worker(void *arg) {
  Shared *s = (Shared *)arg;
  const double dt = (double)s->L / 13e6;   // packet period at 13 Msps (CE_Predictive_Node.hpp:43)
  const double t0 = now_s();
  double next = t0;
  long k = 0;
  while (s->running && now_s() < s->t_end) {
    while (now_s() < next) {}             // recv() blocks until the radio has a packet
    next += dt;
    const int seg = (int)fmod((now_s() - t0) / 0.25, 4.0);
    const std::complex<float> *pkt = &s->capture[((size_t)seg * s->per_seg + (size_t)(k++ % s->per_seg)) * s->L];
    s->packets++;
    if (s->ecr.ce_sensing_flag) {                        // :1310
      const double a = now_s();
      pthread_mutex_lock(&s->CE_mutex);                  // :1311
      s->rx_wait_us.push_back((float)((now_s() - a) * 1e6));
      if (s->ecr.ce_sensing_flag) {                      // :1314
        memcpy(s->ecr.ce_usrp_rx_buffer, pkt, (size_t)s->L * sizeof(std::complex<float>));   // :1316
        s->seg_of_buffer = seg;
        s->ecr.CE_metrics.CE_event = ExtensibleCognitiveRadio::USRP_RX_SAMPS;               // :1320
        pthread_cond_signal(&s->CE_execute_sig);                                             // :1321
        s->forwarded++;
      }
      pthread_mutex_unlock(&s->CE_mutex);                // :1323
    }
  }
  s->running = 0;
  return NULL;
}

int main(int argc, char **argv) {
  if (argc < 5) {
    fprintf(stderr, "usage: %s iq.bin samples_per_packet packets_per_segment seconds [ce args]\n", argv[0]);
    return 2;
  }
  Shared s;
  s.L = atoi(argv[2]);
  s.per_seg = atoi(argv[3]);
  const double seconds = atof(argv[4]);
  s.capture.resize((size_t)4 * s.per_seg * s.L);
  FILE *f = fopen(argv[1], "rb");
  if (!f || fread(s.capture.data(), sizeof(std::complex<float>), s.capture.size(), f) != s.capture.size()) {
    fprintf(stderr, "cannot read 4 x %d packets of %d samples from %s\n", s.per_seg, s.L, argv[1]);
    return 2;
  }
  fclose(f);
  pthread_mutex_init(&s.CE_mutex, NULL);
  pthread_cond_init(&s.CE_execute_sig, NULL);
  s.ce_timeout_ms = 0.0f;                                // scenarios/predictive_model.cfg:61
  s.packets = s.forwarded = 0;
  s.seg_of_buffer = -1;
  std::vector<std::complex<float> > buf((size_t)s.L);
  s.ecr.ce_usrp_rx_buffer = buf.data();                  // :1268
  s.ecr.ce_usrp_rx_buffer_length = s.L;                  // :1265
  std::vector<char *> ce_argv;
  ce_argv.push_back(argv[0]);
  for (int i = 5; i < argc; i++) ce_argv.push_back(argv[i]);
  ce_argv.push_back(NULL);
  CE_Predictive_Node_GPU *engine = new CE_Predictive_Node_GPU((int)ce_argv.size() - 1, ce_argv.data(), &s.ecr);
  s.ecr.CE = engine;
  s.running = 1;
  s.t_end = now_s() + seconds;
  pthread_t rx;
  pthread_create(&rx, NULL, rx_worker, &s);

  // CE worker (this thread), :1775-1803
  std::vector<float> exec_us, null_us;
  struct Call { float us; double t; char rx; short k; char closed; };   // k: packets of the open epoch the engine held before the call
  std::vector<Call> calls;
  const double t_start = now_s();
  std::deque<int> open_epoch;                 // segments of the frames the engine has taken for the epoch being staged
  std::deque<std::vector<int> > awaiting;     // epochs handed to the GPU, decision not yet reported
  long seen = 0, taken = 0;
  while (s.running) {
    struct timeval tv;
    gettimeofday(&tv, NULL);
    double timeout_ns = (double)tv.tv_usec * 1e3 + (double)tv.tv_sec * 1e9 + s.ce_timeout_ms * 1e6, sp;
    const double nsp = modf(timeout_ns / 1e9, &sp);
    struct timespec timeout;
    timeout.tv_sec = (long)sp;
    timeout.tv_nsec = (long)(nsp * 1e9);
    pthread_mutex_lock(&s.CE_mutex);
    if (ETIMEDOUT == pthread_cond_timedwait(&s.CE_execute_sig, &s.CE_mutex, &timeout))
      s.ecr.CE_metrics.CE_event = ExtensibleCognitiveRadio::TIMEOUT;
    const bool rx_event = s.ecr.CE_metrics.CE_event == ExtensibleCognitiveRadio::USRP_RX_SAMPS;
    const int seg = s.seg_of_buffer;
    const long closed_before = engine->epochs_closed;
    const double a = now_s();
    engine->execute();                                   // :1802, CE_mutex held
    const double b = now_s();
    const double c = now_s();                            // control: nothing between b and c
    exec_us.push_back((float)((b - a) * 1e6));
    null_us.push_back((float)((c - b) * 1e6));
    calls.push_back(Call{(float)((b - a) * 1e6), a - t_start, (char)rx_event, (short)open_epoch.size(), (char)(engine->epochs_closed != closed_before)});
    if (rx_event && !engine->packets_dropped) {
      taken++;
      open_epoch.push_back(seg);
      if (open_epoch.size() == 10) {
        awaiting.push_back(std::vector<int>(open_epoch.begin(), open_epoch.end()));
        open_epoch.clear();
      }
    }
    engine->packets_dropped = 0;
    s.ecr.CE_metrics.CE_event = ExtensibleCognitiveRadio::TIMEOUT;   // the event is consumed
    // one execute() may report more than one decision (two batches that came back together, e.g. after the launcher thread was
    // descheduled): engine->decision is the last of them, the earlier ones are counted but cannot be attributed from here
    while (engine->epochs_closed != seen && !awaiting.empty()) {
      seen++;
      const std::vector<int> &segs = awaiting.front();
      const bool last = seen == engine->epochs_closed;
      const bool pure = std::count(segs.begin(), segs.end(), segs[0]) == (long)segs.size();
      printf("decision %d frames_from_segment %d pure %d t %.4f\n", last ? engine->decision : -1, segs[0], (int)(pure && last), s.ecr.now());
      awaiting.pop_front();
    }
    pthread_mutex_unlock(&s.CE_mutex);                   // :1803
  }
  pthread_join(rx, NULL);
  std::sort(s.rx_wait_us.begin(), s.rx_wait_us.end());
  std::sort(exec_us.begin(), exec_us.end());
  const size_t n = s.rx_wait_us.size(), m = exec_us.size();
  printf("packets %ld offered_while_sensing %zu forwarded %ld taken_by_engine %ld epochs %ld\n", s.packets, n, s.forwarded, taken,
         engine->epochs_closed);
  if (n) printf("rx_wait_for_CE_mutex_us n %zu median %.3f p99 %.3f max %.3f\n", n, s.rx_wait_us[n / 2], s.rx_wait_us[(size_t)(n * 0.99)], s.rx_wait_us[n - 1]);
  if (m) printf("execute_us n %zu median %.3f p99 %.3f p9999 %.3f max %.3f\n", m, exec_us[m / 2], exec_us[(size_t)(m * 0.99)], exec_us[(size_t)(m * 0.9999)], exec_us[m - 1]);
  std::sort(null_us.begin(), null_us.end());
  if (m) printf("control_two_clock_reads_us n %zu median %.3f p99 %.3f p9999 %.3f max %.3f\n", m, null_us[m / 2], null_us[(size_t)(m * 0.99)], null_us[(size_t)(m * 0.9999)], null_us[m - 1]);
  std::sort(calls.begin(), calls.end(), [](const Call &x, const Call &y) { return x.us > y.us; });
  for (size_t i = 0; i < calls.size() && i < 10; i++)
    printf("longest_execute %zu: %.3f us at t %.4f s, event %s, packet %d of the epoch, %s\n", i + 1, calls[i].us, calls[i].t,
           calls[i].rx ? "USRP_RX_SAMPS" : "TIMEOUT", calls[i].rx ? calls[i].k + 1 : 0, calls[i].closed ? "a decision was reported in it" : "no decision");
  engine->release();
  return 0;
}

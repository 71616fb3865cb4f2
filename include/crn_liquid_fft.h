/* crn_liquid_fft.h — the three liquid-dsp entry points the reference's sensing path binds, over the
 * MI355X transform (libcrnliquidfft.so, which links libcrnsense.so).
 *
 * The reference calls (cognitive_engines/CE_Predictive_Node/CE_Predictive_Node.cpp):
 *     fft = fft_create_plan(512, buffer, buffer_F, LIQUID_FFT_FORWARD, 0);     // :42-45
 *     fft_execute(fft);                                                        // :150, once per packet
 * and never destroys the plan (empty destructor, :49).  Signatures and types below are liquid-dsp's
 * (liquid.h of git a4d7c80d3: liquid_float_complex = float _Complex in C / std::complex<float> in
 * C++; fftplan is an opaque pointer; LIQUID_FFT_FORWARD = +1, LIQUID_FFT_BACKWARD = -1).
 *
 * Scope: forward transforms of N in {512, 1024, 2048, 4096} (the sizes of the sensing kernel) run on
 * the GPU.  Any other (n, dir) — liquid's own internal plans included: the ECR constructor's
 * ofdmflexframegen_create / ofdmflexframesync_create (src/extensible_cognitive_radio.cpp:113,123) build
 * backward plans of the subcarrier count through these same symbols once this library precedes
 * -lliquid — is handed to the next definition in the search order (dlsym(RTLD_NEXT, ..): liquid's), and
 * fft_execute / fft_destroy_plan dispatch on which library created the plan.  Only when there is no next
 * definition, or no GPU for a sensing plan, does the reference's error convention for set-up failures apply —
 * a message on stderr and exit(EXIT_FAILURE) (src/crts.cpp:111-115) — because the liquid API has no error
 * return.  There is no CPU fallback for the sensing sizes.
 *
 * One fft_execute is one host->device copy, one launch and one device->host copy: it is the drop-in
 * for an engine that must link unchanged, not the fast path (that is crn_sense_run_device, which
 * keeps K frames x many epochs resident).  To let these definitions win over liquid's
 * in a CRTS link line, name -lcrnliquidfft before -lliquid.
 */
#ifndef CRN_LIQUID_FFT_H
#define CRN_LIQUID_FFT_H

#ifdef __cplusplus
#include <complex>
typedef std::complex<float> liquid_float_complex;
extern "C" {
#else
#include <complex.h>
typedef float _Complex liquid_float_complex;
#endif

#define LIQUID_FFT_FORWARD (+1)
#define LIQUID_FFT_BACKWARD (-1)

typedef struct fftplan_s *fftplan;

/* x: input array [n], y: output array [n]; both stay owned by the caller and are read / written by
 * every fft_execute(plan), exactly as liquid binds them at plan creation. */
__attribute__((visibility("default"))) fftplan fft_create_plan(unsigned int n, liquid_float_complex *x,
                                                               liquid_float_complex *y, int dir, int flags);
__attribute__((visibility("default"))) void fft_execute(fftplan p);
__attribute__((visibility("default"))) void fft_destroy_plan(fftplan p);
/* diagnostic: plans handed on to the next definition (liquid's) so far */
__attribute__((visibility("default"))) long crn_liquid_fft_forwarded(void);

#ifdef __cplusplus
}
#endif
#endif

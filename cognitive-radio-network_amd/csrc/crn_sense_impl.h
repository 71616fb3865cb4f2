// crn_sense_impl.h — the sensing kernel template and its launch helpers, included by crn_kernels.hip (complex-float input) and
// crn_kernels_sc16.hip (wire-format input) so that the two sets of instantiations compile in parallel.  Four parts:
//   crn_butterflies.h   packed-f32 complex arithmetic, 4 / 8 / 16-point transforms
//   crn_frame.h         geometry, loads, configuration flags, the phases of one frame
//   crn_epoch_close.h   band sums, features, decision: once per K frames
//   crn_sense_kernel.h  the kernel (frame loop) and launch_cfg / launch_default / launch_rn
// Internal linkage throughout (static / constexpr / templates).
#ifndef CRN_SENSE_IMPL_H
#define CRN_SENSE_IMPL_H
#include "crn_sense_kernel.h"
#endif

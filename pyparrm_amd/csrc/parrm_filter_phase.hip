// filter_data phase-major kernel, f64 -> f64 instantiations + the host-side planner
// (implementation: parrm_filter_phase_impl.h)
#define PARRM_PHASE_TI double
#define PARRM_PHASE_TO double
#define PARRM_PHASE_WITH_PLAN 1
#include "parrm_filter_phase_impl.h"

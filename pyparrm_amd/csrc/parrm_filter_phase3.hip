// filter_data phase3 kernel (three residues per lane), f64 -> f64 instantiations + the host-side planner
// (implementation: parrm_filter_phase3_impl.h)
#define PARRM_PHASE_TI double
#define PARRM_PHASE_TO double
#define PARRM_PHASE3_WITH_PLAN 1
#include "parrm_filter_phase3_impl.h"

// filter_data phase3 kernel, f32 -> f32 instantiations (implementation: parrm_filter_phase3_impl.h)
#define PARRM_PHASE_TI float
#define PARRM_PHASE_TO float
#include "parrm_filter_phase3_impl.h"

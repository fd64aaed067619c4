// cg_fast6.hip — what the 6-colour (triangular-lattice) lane programs share across sites-per-lane counts (kernels: cg_fast_npl.hip, -DELPH_LP_MC=6); see cg_fast.hip.
#include "cg_fast_common.h"

#define ELPH_LP_MC 6
#define LPNS lp6
#include "cg_fast_shared.inc"
#undef ELPH_LP_MC
#undef LPNS

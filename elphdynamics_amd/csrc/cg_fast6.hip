// cg_fast6.hip — the 6-colour (triangular-lattice) instantiation of the lane-program kernels; see cg_fast.hip.
#include "cg_fast_common.h"

#define ELPH_LP_MC 6
#define LPNS lp6
#include "cg_fast_impl.inc"
#undef ELPH_LP_MC
#undef LPNS

// cg_fast_npl.hip — one sites-per-lane count of the lane-program kernels (cg_fast_impl.inc).  build.py compiles this file once per
// (ELPH_LP_MC, ELPH_LP_NPL) in {4, 6} x {1 ... 8} (-DELPH_LP_MC=... -DELPH_LP_NPL=...); without the flags it is the 4-colour, 4-sites-per-lane
// unit (the 16 x 16 square lattice of BASELINE config C).  See cg_fast.hip for what the family is.
#include "cg_fast_common.h"
#include "kpm_sq_dev.h"

#ifndef ELPH_LP_MC
#define ELPH_LP_MC 4
#endif
#ifndef ELPH_LP_NPL
#define ELPH_LP_NPL 4
#endif
#if ELPH_LP_MC == 4
#define LPNS lp4
#elif ELPH_LP_MC == 6
#define LPNS lp6
#else
#error "lane programs of 4 or 6 colours"
#endif
#if ELPH_LP_NPL < 1 || ELPH_LP_NPL > 8
#error "1 ... 8 sites per lane"
#endif
#include "cg_fast_impl.inc"

/* Plain-C oracle for the Pi-block step (TEST INFRASTRUCTURE ONLY -- see pi_oracle_impl.h).
 * Build: make -C oracle   (gcc -O2 -ffp-contract=off: every rounding is the one written down). */
#include <math.h>

#define REAL float
#define SUF f32
#define FMA(a, b, c) fmaf((a), (b), (c))
#include "pi_oracle_impl.h"
#undef REAL
#undef SUF
#undef FMA

#define REAL double
#define SUF f64
#define FMA(a, b, c) fma((a), (b), (c))
#include "pi_oracle_impl.h"
#undef REAL
#undef SUF
#undef FMA

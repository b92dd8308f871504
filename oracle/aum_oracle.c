/*
 * ORACLE (test infrastructure only; see aum_oracle_impl.h for the rules and citations).
 * Builds both precisions of the CPU restatement into one shared object:
 *   *_f32 : the reference's own fp32 internal arithmetic
 *   *_f64 : double-precision truth
 * Build: `make -C oracle` (gcc -O2 -fopenmp -shared).  Output: oracle/_build/libaum_oracle.so
 */
#include <math.h>
#include <stddef.h>
#include <stdlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define REAL float
#define SUFFIX _f32
#include "aum_oracle_impl.h"
#undef REAL
#undef SUFFIX

#define REAL double
#define SUFFIX _f64
#include "aum_oracle_impl.h"
#undef REAL
#undef SUFFIX

int aum_oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

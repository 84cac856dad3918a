// c2_loglik_t2.hip -- the one-lane-per-series kernels of c2_loglik_t.hip compiled for width J = 2 (tiles of 8 rows).
#define C2T_J 2
#include "c2_loglik_t.hip"

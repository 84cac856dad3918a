// c2_loglik_t4.hip -- the one-lane-per-series kernels of c2_loglik_t.hip compiled for width J = 4 (tiles of 4 rows).
#define C2T_J 4
#include "c2_loglik_t.hip"

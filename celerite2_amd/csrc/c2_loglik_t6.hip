// One lane per series for J = 6: rows of 6 in memory, computed as rows of 8 with two empty columns (see C2T_JS).
#define C2T_J 8
#define C2T_JS 6
#include "c2_loglik_t.hip"

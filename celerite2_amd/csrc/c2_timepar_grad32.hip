// c2_timepar_grad32.hip -- the time-parallel gradient / Newton factor of c2_timepar_grad.hip with chunks of 32 rows (a
// handful of series: twice as many lanes busy, half the walk per lane).
#define C2TG_ROWS 32
#include "c2_timepar_grad.hip"

// c2_timepar_grad16.hip -- the time-parallel gradient / Newton factor of c2_timepar_grad.hip with chunks of 16 rows (a
// single short series: four times as many lanes busy, a quarter of the walk per lane).
#define C2TG_ROWS 16
#include "c2_timepar_grad.hip"

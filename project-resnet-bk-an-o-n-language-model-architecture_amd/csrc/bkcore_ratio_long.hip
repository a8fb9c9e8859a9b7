// The ratio-form forward compiled with double-precision chains and product, for rows longer than
// BKCORE_INTERNAL_LONG_ROW elements (see the note on BK_RATIO_REAL in bkcore_ratio.hip).  Entry points carry the suffix
// _long.
#define BK_RATIO_REAL double
#define BK_RATIO_LONG 1
#include "bkcore_ratio.hip"

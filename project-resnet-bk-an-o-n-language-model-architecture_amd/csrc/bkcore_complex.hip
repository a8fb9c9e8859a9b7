// The scan kernel of bkcore_kernels.hip compiled once more for complex operands: complex he_diag, h0_super and h0_sub
// (interleaved complex64) for the raw resolvent diagonal, which the reference computes for any operand type by
// converting to complex128 (src/models/bk_core.py:45-47).  k = b c is complex here, the diagonal gets an imaginary
// plane in LDS, and a chunk is 256 x 8 elements so that the staging area stays within one workgroup's LDS.
// Exports bkcore_get_diag_complex and bkcore_get_diag_complex_workspace_bytes.
#define BK_COMPLEX_BUILD 1
#define BK_E 8
#include "bkcore_kernels.hip"

// Functions shared between the translation units of libbkcore.so (not part of the public ABI).
#ifndef BKCORE_INTERNAL_H
#define BKCORE_INTERNAL_H
#include <stddef.h>
#include <stdint.h>

// Float32 forward for the analytic backward: stores, per element, r_j = theta_{j+1}/theta_j and s_j = phi_j/phi_{j+1}
// (complex64 each), and writes to G the resolvent diagonal BEFORE the final log-magnitude clamp and the |G| <= 50
// clip (0 where |G_raw| is outside [1e-18, 1e18] or not a number).  Other arguments as bkcore_forward; no features.
extern "C" __attribute__((visibility("hidden"))) int bkcore_internal_forward_save(
    const float *a, int64_t lda, const float *b, int64_t ldb, const float *c, int64_t ldc, const float *z_dev,
    float z_re, float z_im, int64_t B, int64_t N, float *G, void *workspace, void *state, unsigned flags, void *stream,
    float *save_r, float *save_s);

// The same by one launch of the two-scan kernel, for the rows with row_filter[row] != 0 (every row when NULL) and only
// if *gate != 0 (always when NULL): the analytic backward's path for rows its sweeps flagged.
extern "C" __attribute__((visibility("hidden"))) int bkcore_internal_forward_save_rows(
    const float *a, int64_t lda, const float *b, int64_t ldb, const float *c, int64_t ldc, const float *z_dev,
    float z_re, float z_im, int64_t B, int64_t N, float *G, void *workspace, unsigned flags, void *stream,
    float *save_r, float *save_s, const int *row_filter, const int *gate);

// Launches of the ratio-form forward (bkcore_ratio.hip); errors are left for the caller's hipGetLastError().
// mode 3: rows cut into S units of seg_ch chunks that hand over inside ONE launch; `save` then carries the number of
// workgroups the GPU holds at once (the grid is at most that).
// mode 0: one workgroup per row (save == 1: the variant that also stores r, s and writes G_raw; save == 2: the variant
// that stores the boundary states of the analytic backward into save_r).  mode 4: one workgroup per chunk, S = chunks per
// row, handing the running product over inside the launch.  With mode 4 a non-null save_r selects the variant that
// stores the boundary states there (float32 operands, float32 chains only; not available with mode 3).
extern "C" __attribute__((visibility("hidden"))) void bkcore_internal_ratio_forward(
    int in_dtype, int phi_exact, int mode, int save, const void *a, int64_t lda, const void *b, int64_t ldb,
    const void *c, int64_t ldc, const float *z_dev, float z_re, float z_im, int64_t B, int64_t N, float *G,
    float *features, void *ws, void *state, unsigned flags, float fclamp, int S, int seg_ch, float *save_r,
    float *save_s, void *stream);
// The same launch with double-precision chains and product (bkcore_ratio_long.hip; no saving variant), taken for
// rows longer than BKCORE_INTERNAL_LONG_ROW elements.  With float32 chains the rounding of the N factors of the running
// product is a random walk; the worst row of 54 per setting measured (tools/dev_err_vs_n.py, against the C oracle, for
// z = 1j, 0.8j, 0.7j, -0.4+0.8j, 0.45+0.75j): N = 32768 4.1e-5, 49152 5.0e-5, 65536 6.7e-5, 98304 7.6e-5, 131072 1.03e-4.
// The operator's bound is 1e-4, so float32 stops at 49152 (twelve chunks).
#define BKCORE_INTERNAL_LONG_ROW 49152
extern "C" __attribute__((visibility("hidden"))) void bkcore_internal_ratio_forward_long(
    int in_dtype, int phi_exact, int mode, int save, const void *a, int64_t lda, const void *b, int64_t ldb,
    const void *c, int64_t ldc, const float *z_dev, float z_re, float z_im, int64_t B, int64_t N, float *G,
    float *features, void *ws, void *state, unsigned flags, float fclamp, int S, int seg_ch, float *save_r,
    float *save_s, void *stream);
// Workgroups of the mode 3 kernel one CU holds at once on the current device (0: unknown).
extern "C" __attribute__((visibility("hidden"))) int bkcore_internal_ratio_unit_occupancy(int in_dtype, int phi_exact);
// Elements per chunk of the ratio-form kernel (its workgroup size x 16).
extern "C" __attribute__((visibility("hidden"))) int bkcore_internal_ratio_chunk(void);
// Records the HIP error code that bkcore_last_hip_error() reports (per thread).
extern "C" __attribute__((visibility("hidden"))) void bkcore_internal_set_hip_error(int code);
#endif

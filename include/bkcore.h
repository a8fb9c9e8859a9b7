/*
 * bkcore.h -- C ABI of the MI355X (gfx950) BK-Core library, libbkcore.so.
 *
 * BK-Core: for a batch of tridiagonal matrices T (diagonal a[B,N], super-diagonal b[B,N-1],
 * sub-diagonal c[B,N-1]) and a complex shift z, the diagonal of the resolvent
 * G_ii = diag((T - zI)^-1), by the theta (prefix) / phi (suffix) three-term recurrences.
 *
 * Every entry point is plain C: raw device pointers, sizes, a HIP stream passed as void*.
 * The library never allocates or frees device memory and keeps no device pointer after a call
 * returns.  All functions return 0 on success or a negative BKCORE_E* code; nothing throws.
 * They are re-entrant; work is enqueued on `stream` of the CURRENT device (the caller sets the
 * device, e.g. torch.cuda.device) and the call returns without synchronising.
 *
 * Reference interfaces these replace are cited per function as file:line of
 * neko-jpg/Project-ResNet-BK-An-O-N-Language-Model-Architecture.
 */
#ifndef BKCORE_H
#define BKCORE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* element type of a, b, c */
#define BKCORE_F32 0
#define BKCORE_BF16 1
#define BKCORE_F16 2
#define BKCORE_F64 3

/* flags for bkcore_forward */
#define BKCORE_FLAG_RAW 1u        /* skip the operator's input clamp / NaN sanitising: semantics of the batched
                                     get-diag function instead of BKCoreFunction.forward */
#define BKCORE_FLAG_PHI_EXACT 2u  /* phi recurrence uses k_{i+1} (true trailing minors) instead of the
                                     reference's k_i (src/models/bk_core.py:165-171) */

#define BKCORE_FLAG_SCAN_PATH 4u  /* diagnostics: take the scan-based first launch instead of the ratio-form one
                                     (same results within the tolerance; see bkcore_forward) */
#define BKCORE_FLAG_ADAPTIVE 8u   /* opt-in: when the previous call on this state block had to recompute more than half of
                                   * its rows with the scan kernel (inputs on which the ratio maps do not contract, e.g.
                                   * the randn distribution of scripts/benchmark_bk_triton.py:66-70), the next calls skip
                                   * the ratio-form arithmetic (the inputs are still read once for the NaN verdicts) and
                                   * go to the scan kernel for every row; every 16th such call tries the fast path again.
                                   * Which kernel computes a well-conditioned row then depends on the calls before it: the
                                   * two kernels agree to the operator's tolerance, not bit for bit. */

#define BKCORE_FLAG_ROW_UNITS 16u  /* diagnostics: cut every row into two units (with BKCORE_FLAG_ROW_UNITS_MANY, 32u: into
                                    * as many units of two chunks as it has) that hand the running product on inside the
                                    * first launch.  The library does this by itself, two units a row, when there are more
                                    * rows than the GPU holds workgroups at once but fewer than three times as many; the
                                    * flags force it for any row count.  Bit for bit the results of whole rows. */
#define BKCORE_FLAG_ROW_UNITS_MANY 32u
#define BKCORE_FLAG_WHOLE_ROWS 128u /* never take the form in which a grid of one workgroup per place on the GPU hands
                                    * the running product on between the two units of a row (the library's choice for
                                    * float32 / float64 operands between 1.24 and 1.37 times as many rows as places):
                                    * that form needs its whole grid resident at once, which kernels of other streams --
                                    * the collectives of a data-parallel backward -- can prevent; a unit then waits up
                                    * to 50 ms before the call falls back.  Rows go as one workgroup per chunk instead,
                                    * whose hand-over only depends on workgroups that started earlier. */
#define BKCORE_FLAG_ROWS_UNCUT 256u /* diagnostics: one workgroup per row whatever the shape (the form the others are
                                    * checked against: units give its results bit for bit, chunks within 1e-5) */
#define BKCORE_FLAG_ROW_UNITS_MUTE 64u /* diagnostics, with one of the two above: the units do not publish their records, so
                                    * every unit that waits for one gives up after 50 ms and the call's last launch
                                    * recomputes every row (exercises that recovery; same results within the tolerance) */

/* error codes */
#define BKCORE_OK 0
#define BKCORE_EINVAL (-1)   /* bad argument (null pointer, negative size, unknown dtype) */
#define BKCORE_ETOOLONG (-2) /* N larger than bkcore_max_seq_len() */
#define BKCORE_ELAUNCH (-3)  /* HIP launch failed; bkcore_last_hip_error() has the HIP code */
#define BKCORE_EWORKSPACE (-4) /* workspace missing or smaller than bkcore_workspace_bytes(B, N) */
#define BKCORE_ETOOBIG (-5)  /* bkcore_forward_noalloc: bkcore_workspace_bytes(B, N) exceeds BKCORE_NOALLOC_MAX_WORKSPACE */

/* Size in bytes of the persistent per-(device,stream) state block bkcore_forward may be given.
 * Allocate once with device memory, zero it once, reuse it for every call on that stream. */
#define BKCORE_STATE_BYTES 64

/*
 * Forward: G and features.
 *
 * Replaces, on the reference side,
 *   - BKCoreFunction.forward                       src/models/bk_core.py:239-300  (flags = 0)
 *   - vmapped_get_diag / get_tridiagonal_inverse_diagonal  src/models/bk_core.py:27-216  (flags = BKCORE_FLAG_RAW)
 * i.e. the functional slot f(a, b, c, z) -> G that all of the reference's back-ends share
 * (src/models/bk_core.py:284,288,291).
 *
 * a: B rows of N elements, row r at a + r*lda elements (lda = 0 broadcasts one row, as the layer's
 *    expand()ed buffers do, src/models/resnet_bk.py:155-157); b, c likewise with N-1 elements per row.
 *    Inner stride is 1.  in_dtype is one of BKCORE_F32/BF16/F16/F64 and applies to all three.
 * z: either z_dev != NULL (device pointer to 2 floats: re, im -- no host sync needed when z is a
 *    device tensor, src/models/resnet_bk.py:159-161) or the host values z_re, z_im.
 * G: B*N interleaved (re, im) float pairs = complex64, row-major.  Required.
 * features: B*N*2 floats, the same numbers as G (src/models/bk_core.py:296-298), or NULL to skip.
 * state: BKCORE_STATE_BYTES of zero-initialised persistent device memory (one block per device and stream,
 *    reused by every call), or NULL.  It carries two verdicts from the first kernel launch of a call to its second
 *    one without a host sync: (1) the first launch takes a shortcut for the scan that only feeds the reference's
 *    in-recurrence soft clamp, checks it per row, and rows that fail the check are recomputed by the full two-scan
 *    path in the second launch; (2) NaNs in b or c are replaced by 1 only when he_diag holds a NaN somewhere in the
 *    batch (src/models/bk_core.py:260-264).  With state == NULL there is a single launch on the full two-scan path
 *    and NaNs in b/c are never replaced (such rows come out as zeros, as in the reference when he_diag is clean).
 * workspace: device scratch of at least bkcore_workspace_bytes(B, N) bytes (256-byte aligned), owned by the caller
 *    and free to reuse as soon as the stream has passed this call.
 */
int bkcore_forward(const void *a, int64_t lda, const void *b, int64_t ldb, const void *c, int64_t ldc, int in_dtype,
                   const float *z_dev, float z_re, float z_im, int64_t B, int64_t N, float *G, float *features,
                   void *workspace, size_t workspace_bytes, void *state, unsigned flags, void *stream);

/* Scratch bytes bkcore_forward needs for a (B, N) problem: 4 bytes per row plus 20 bytes per 16 elements (rounded up
 * to whole 4096-element chunks).  No counterpart in the reference, which allocates its theta/phi buffers inside the
 * call (src/kernels/bk_scan.py:416-417). */
size_t bkcore_workspace_bytes(int64_t B, int64_t N);

/*
 * The raw resolvent diagonal for complex operands.
 *
 * Replaces get_tridiagonal_inverse_diagonal / vmapped_get_diag called with complex he_diag, h0_super, h0_sub:
 * the reference converts whatever it is given to complex128 (src/models/bk_core.py:45-47) and runs the same
 * recurrences, regularisation, soft clamps and clip (src/models/bk_core.py:58-210).  BKCoreFunction itself cannot take
 * complex operands (its input clamp, src/models/bk_core.py:256-258, is not defined for them), so there is no clamped /
 * sanitised form, no features tensor and no backward here.
 *
 * a, b, c: interleaved (re, im) float pairs = complex64; B rows of N (a) and N-1 (b, c) elements, leading dimensions in
 *    complex elements (0 broadcasts one row).  z, G, stream as for bkcore_forward.
 * flags: 0 or BKCORE_FLAG_PHI_EXACT; anything else is BKCORE_EINVAL.
 * workspace: at least bkcore_get_diag_complex_workspace_bytes(B, N) bytes of device scratch.
 * One launch of the two-scan kernel (no assumption about the input: nothing to verify or recompute); arithmetic in
 * float32 mantissas with integer exponents like the real-operand kernels, results to 1e-4 of the reference's.
 */
int bkcore_get_diag_complex(const float *a, int64_t lda, const float *b, int64_t ldb, const float *c, int64_t ldc,
                            const float *z_dev, float z_re, float z_im, int64_t B, int64_t N, float *G, void *workspace,
                            size_t workspace_bytes, unsigned flags, void *stream);
size_t bkcore_get_diag_complex_workspace_bytes(int64_t B, int64_t N);

/*
 * Backward, reference mode: grad_he_diag = (1-alpha) * -Re(conj(g) G^2) + alpha * -Re(conj(g)/G^2) with the
 * reference's stabilisers, where g = grad_features[...,0] + i grad_features[...,1] (+ grad_G).
 * Replaces BKCoreFunction.backward, src/models/bk_core.py:303-385.  alpha is BKCoreFunction.GRAD_BLEND read by
 * the caller at backward time.
 *
 * G: n complex64 saved by forward; grad_features: n*2 floats or NULL; grad_G: n complex64 or NULL;
 * grad_a: n floats (float32, src/models/bk_core.py:382).  n = B*N.
 */
int bkcore_backward_ref(const float *G, const float *grad_features, const float *grad_G, float alpha, int64_t n,
                        float *grad_a, void *stream);
/*
 * The same, and *nonfinite_flag (one int in device memory, may be NULL) is set to 1 when grad_features holds a NaN or
 * an infinity -- the condition under which the reference warns (src/models/bk_core.py:321-325) before it replaces
 * them.  The flag is only ever raised: the caller clears it and reads it back when it wishes (no host
 * synchronisation inside the call).
 */
int bkcore_backward_ref_flagged(const float *G, const float *grad_features, const float *grad_G, float alpha, int64_t n,
                                float *grad_a, int *nonfinite_flag, void *stream);

/*
 * The reference's other backward heuristics on the saved G, same arguments as bkcore_backward_ref plus `variant`:
 *   BKCORE_BWD_REFERENCE   BKCoreFunction.backward, src/models/bk_core.py:303-385 (= bkcore_backward_ref)
 *   BKCORE_BWD_BATCHED     compute_single_gradient of BatchedAnalyticBKCoreFunction / MemoryOptimizedBKCoreFunction,
 *                          src/models/batched_gradient.py:12-54,65-130,174-217: the same blend with min |G^2| 1e-3, the
 *                          cotangent not conjugated, a non-finite result -> 0, clamp +-1000; the cotangents are taken
 *                          as they come (no nan_to_num)
 *   BKCORE_BWD_SCAN_STABLE BKScanStableFunction.backward, src/kernels/bk_scan_stable.py:597: grad_a = -Re(g) |G|^2
 *                          (alpha unused; grad_b, grad_c are zeros there, :605-609)
 * The cotangent g is grad_features read as (Re, Im) pairs plus grad_G, either may be NULL.
 */
#define BKCORE_BWD_REFERENCE 0
#define BKCORE_BWD_BATCHED 1
#define BKCORE_BWD_SCAN_STABLE 2
int bkcore_backward_variant(const float *G, const float *grad_features, const float *grad_G, float alpha, int64_t n,
                            float *grad_a, int variant, void *stream);

/*
 * Backward, analytic mode: the gradient of L with respect to he_diag, h0_super and h0_sub, where
 * dL = sum Re(conj(gbar) dG) and gbar = grad_features[...,0] + i grad_features[...,1] (+ grad_G).  It is the
 * derivative of what bkcore_forward computes -- regularisation, both soft clamps, the |G| <= 50 clip and the chosen
 * phi index convention included -- i.e. what torch.autograd gives through the reference's sequential forward
 * (src/models/bk_core.py:27-210) evaluated in float64.  No counterpart in the reference, whose backward is the
 * heuristic blend above and returns None for h0_super / h0_sub (src/models/bk_core.py:384-385).
 *
 * a, b, c: float32, ALREADY clamped / NaN-sanitised as BKCoreFunction.forward does (src/models/bk_core.py:256-264; the
 *    caller zeroes the gradient where the clamp was active); leading dimensions as in bkcore_forward.
 * grad_features: B*N*2 floats or NULL; grad_G: B*N complex64 or NULL.
 * grad_a: B*N floats.  grad_b, grad_c: B*(N-1) floats each, or both NULL.
 * flags: BKCORE_FLAG_PHI_EXACT or 0, as given to the forward.
 * G: the B*N complex64 that bkcore_forward wrote for these inputs and flags (contiguous), or NULL to have it computed
 *    here.  The theta / phi ratios are recomputed inside the two sweeps of this call from a, b, c; only G is read
 *    from the forward.
 * workspace: at least bkcore_backward_analytic_workspace_bytes(B, N) bytes; state: as in bkcore_forward (rows whose
 *    in-kernel checks fail are redone by a slower three-pass path in the same call; with state == NULL every row
 *    takes that path).
 */
int bkcore_backward_analytic(const float *a, int64_t lda, const float *b, int64_t ldb, const float *c, int64_t ldc,
                             const float *z_dev, float z_re, float z_im, int64_t B, int64_t N, const float *G,
                             const float *grad_features, const float *grad_G, float *grad_a, float *grad_b,
                             float *grad_c, void *workspace, size_t workspace_bytes, void *state, unsigned flags,
                             void *stream);
size_t bkcore_backward_analytic_workspace_bytes(int64_t B, int64_t N);

/*
 * The forward keeping what the analytic backward would otherwise recompute, and that backward starting from it.
 * The reference's forward saves nothing but G (src/models/bk_core.py:293) and its backward is elementwise; these two
 * have no counterpart there -- they are how "the analytic backward reuses the forward's scan states" is done here.
 *
 * bkcore_forward_states = bkcore_forward_clamped that also writes, for every block of 16 consecutive elements of every
 * row, the two values its ratio recurrences end on at the block's borders (theta: what enters the block from the left;
 * phi: the value at the block's second element, one step short of what enters the block to its left): 16 bytes per 16
 * elements, rows padded to whole chunks of 4096 plus one block -- bkcore_boundary_state_bytes(B, N) in all -- into
 * boundary_states (device memory, 16-byte aligned, boundary_state_bytes long: BKCORE_EWORKSPACE if that is less than
 * bkcore_boundary_state_bytes(B, N)).  *states_written (host int, may be NULL) says whether this call's
 * path wrote them: yes for float32 operands with a state block, rows of at most 49152 elements and none of
 * BKCORE_FLAG_SCAN_PATH / BKCORE_FLAG_ADAPTIVE; otherwise the call is bkcore_forward_clamped and the backward has to
 * be given NULL.  boundary_states == NULL: plain bkcore_forward_clamped.
 *
 * bkcore_backward_analytic_states = bkcore_backward_analytic with boundary_states from the forward call that produced
 * G for the same operands and flags (G is then required): each lane reaches the start of its ratio chains from the
 * kept values by eight exact steps of one of the two recurrences instead of a 16-step warm-up of both.  Rows whose
 * forward was recomputed on the two-scan path hold states that
 * do not belong to their G; the lanes' consistency checks catch that and the row takes the three-pass path, as any row
 * failing its checks does.  boundary_states == NULL: bkcore_backward_analytic.
 */
size_t bkcore_boundary_state_bytes(int64_t B, int64_t N);
int bkcore_forward_states(const void *a, int64_t lda, const void *b, int64_t ldb, const void *c, int64_t ldc,
                          int in_dtype, const float *z_dev, float z_re, float z_im, int64_t B, int64_t N, float *G,
                          float *features, float feature_clamp, float *boundary_states, size_t boundary_state_bytes,
                          int *states_written, void *workspace, size_t workspace_bytes, void *state, unsigned flags,
                          void *stream);
int bkcore_backward_analytic_states(const float *a, int64_t lda, const float *b, int64_t ldb, const float *c,
                                    int64_t ldc, const float *z_dev, float z_re, float z_im, int64_t B, int64_t N,
                                    const float *G, const float *boundary_states, size_t boundary_state_bytes,
                                    const float *grad_features, const float *grad_G, float *grad_a, float *grad_b,
                                    float *grad_c, void *workspace, size_t workspace_bytes, void *state, unsigned flags,
                                    void *stream);
/* Diagnostics: byte offset, inside the workspace of bkcore_backward_analytic(B, N), of B ints that are non-zero after
 * the call for the rows the three-pass path recomputed (valid when state was given).  No counterpart in the reference. */
size_t bkcore_backward_analytic_rowflag_offset(int64_t B, int64_t N);

/*
 * bkcore_forward with the caller's clamp of the features folded into the store: features = clamp((Re G, Im G),
 * -feature_clamp, feature_clamp) as MoEResNetBKLayer.forward applies after the operator
 * (src/models/resnet_bk.py:216-217); G itself is written unclamped.  feature_clamp <= 0: no clamp (identical to
 * bkcore_forward).  All other arguments as in bkcore_forward.
 */
int bkcore_forward_clamped(const void *a, int64_t lda, const void *b, int64_t ldb, const void *c, int64_t ldc,
                           int in_dtype, const float *z_dev, float z_re, float z_im, int64_t B, int64_t N, float *G,
                           float *features, float feature_clamp, void *workspace, size_t workspace_bytes, void *state,
                           unsigned flags, void *stream);

/*
 * bkcore_forward_clamped without a workspace and without a state block from the caller: both live in a fixed array of
 * device memory that is part of the library's code object (BKCORE_NOALLOC_MAX_WORKSPACE + 256 bytes on every GPU the
 * library runs on, held until the process ends).  The call allocates and frees nothing, on the host or the device.
 * No counterpart in the reference; it is for callers that cannot allocate: code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc.
 *
 * Limit: bkcore_workspace_bytes(B, N) <= BKCORE_NOALLOC_MAX_WORKSPACE, else BKCORE_ETOOBIG and nothing is launched
 * (4 bytes per row + 5120 bytes per row and started 4096-element chunk: e.g. B = 256 rows of N = 24576, or B = 8 of
 * N = 2^19).  Every call on a GPU shares that GPU's array: calls that overlap on one GPU (two streams, two threads)
 * race on it; calls on one stream are ordered and safe.  All other arguments and results as bkcore_forward_clamped.
 */
#define BKCORE_NOALLOC_MAX_WORKSPACE (8u << 20)
int bkcore_forward_noalloc(const void *a, int64_t lda, const void *b, int64_t ldb, const void *c, int64_t ldc,
                           int in_dtype, const float *z_dev, float z_re, float z_im, int64_t B, int64_t N, float *G,
                           float *features, float feature_clamp, unsigned flags, void *stream);

/*
 * The caller's elementwise steps in front of the operator, in one pass: replaces the tensor-library chain of
 * MoEResNetBKLayer.forward, src/models/resnet_bk.py:139-141 (nan_to_num(v, 0, +v_max, -v_max); v_max * tanh(v / v_max)),
 * :158 (he_diag = h0_diag + v) and :165 (cast to bfloat16).
 *
 * v_pre: B rows of N floats (the output of v_proj), row stride ldv; h0_diag: N floats (one row, broadcast over B);
 * he_diag: B*N contiguous elements of out_dtype (BKCORE_BF16: round to nearest even; or BKCORE_F32).
 */
int bkcore_layer_prologue(const float *v_pre, int64_t ldv, const float *h0_diag, float v_max, int64_t B, int64_t N,
                          void *he_diag, int out_dtype, void *stream);

/*
 * Backward of the same chain fused with bkcore_backward_ref: gradient with respect to v_pre given the cotangents of the
 * CLAMPED features (clamp(features, -feature_clamp, feature_clamp), src/models/resnet_bk.py:216-217) and of G.
 * Replaces, in one pass over n = B*N elements: the clamp's backward, BKCoreFunction.backward
 * (src/models/bk_core.py:303-385), the autograd engine's cast of that float32 gradient to the bfloat16 operand's dtype
 * (round_bf16 != 0), and the backward of the tanh / nan_to_num chain.
 *
 * G: n complex64 saved by forward; grad_features: n*2 floats or NULL; grad_G: n complex64 or NULL;
 * feature_clamp <= 0: no clamp; v_pre: n contiguous floats as given to bkcore_layer_prologue; grad_v: n floats.
 */
int bkcore_layer_backward_ref(const float *G, const float *grad_features, const float *grad_G, float alpha,
                              float feature_clamp, const float *v_pre, float v_max, int round_bf16, int64_t n,
                              float *grad_v, void *stream);

/*
 * The spectral branch's output stage: out = residual + bk_scale * (features @ weight^T + bias), replacing
 * `spec_out = self.output_proj(features); output = ffn_out + self.bk_scale * spec_out`
 * (src/models/resnet_bk.py:218-219; output_proj = nn.Linear(2, D), :86-87) in one pass.
 *
 * features: n*2 floats (already clamped); weight: D*2 floats (row d = weights of output d, the layout of
 * nn.Linear(2, D).weight); bias, scale: D floats; residual: n*D floats or NULL; out: n*D floats.  n = B*N positions.
 */
int bkcore_layer_epilogue(const float *features, const float *weight, const float *bias, const float *scale,
                          const float *residual, int64_t n, int64_t D, float *out, void *stream);

/*
 * Backward of bkcore_layer_epilogue in one pass over grad_out (n*D floats): grad_features (n*2 floats), and -- each
 * optional -- grad_weight (D*2), grad_bias (D), grad_scale (D).  The gradient of `residual` is grad_out itself.
 * Replaces the autograd chain of src/models/resnet_bk.py:218-219.  Column sums are formed per workgroup and combined
 * in a fixed order (deterministic).  D <= 1024 (BKCORE_EINVAL beyond).  workspace: at least
 * bkcore_layer_epilogue_backward_workspace_bytes(n, D) bytes.
 */
int bkcore_layer_epilogue_backward(const float *grad_out, const float *features, const float *weight, const float *bias,
                                   const float *scale, int64_t n, int64_t D, float *grad_features, float *grad_weight,
                                   float *grad_bias, float *grad_scale, void *workspace, size_t workspace_bytes,
                                   void *stream);
size_t bkcore_layer_epilogue_backward_workspace_bytes(int64_t n, int64_t D);

/*
 * The same two with the clamp of the features (`features = torch.clamp(features, -c, c)`,
 * src/models/resnet_bk.py:216-217) applied as the features are read: `features` may then be the operator's unclamped
 * output -- the float32 view of G_ii itself -- so that no clamped copy is ever written.  The backward's grad_features
 * is the gradient with respect to the UNCLAMPED values (zero where |feature| > feature_clamp).  feature_clamp <= 0:
 * no clamp (the plain entry points).
 */
int bkcore_layer_epilogue_clamped(const float *features, float feature_clamp, const float *weight, const float *bias,
                                  const float *scale, const float *residual, int64_t n, int64_t D, float *out,
                                  void *stream);
int bkcore_layer_epilogue_backward_clamped(const float *grad_out, const float *features, float feature_clamp,
                                           const float *weight, const float *bias, const float *scale, int64_t n,
                                           int64_t D, float *grad_features, float *grad_weight, float *grad_bias,
                                           float *grad_scale, void *workspace, size_t workspace_bytes, void *stream);

/*
 * Cumulative sum along the middle dimension of a contiguous float32 tensor viewed as (outer, L, inner):
 * y[o, l, i] = sum_{l' <= l} x[o, l', i]  (reverse != 0: sum over l' >= l).  Replaces
 * fused_associative_scan(x, dim, reverse), src/kernels/associative_scan.py:211-308, and the torch.cumsum it falls back
 * to in the AR-SSM layer (src/models/phase1/ar_ssm_layer.py:517-525: k_cumsum over dim 1 of (B, L, rank)).
 * x, y: outer*L*inner floats, distinct buffers.
 */
int bkcore_cumsum(const float *x, float *y, int64_t outer, int64_t L, int64_t inner, int reverse, void *stream);

/*
 * Woodbury low-rank correction of the resolvent diagonal for every row of a batch:
 *     out_b = G_b - G_b^2 * rowsum((U (I + V^T diag(G_b) U)^{-1}) o V)
 * Replaces the Python loop over the batch in BirmanSchwingerCore.compute_semiseparable_resolvent,
 * src/models/birman_schwinger_core.py:567-604 (complex64 arithmetic as there, :563).
 *
 * G, out: B*N complex64; U, V: N*r floats, row-major (N, r) as SemiseparableMatrix.U / .V; r <= 22.
 * A row whose r x r system is singular to working precision is returned uncorrected (the reference falls back to a
 * pseudo-inverse there, :584-586).  workspace: at least bkcore_woodbury_workspace_bytes(B, N, r) bytes.
 */
int bkcore_woodbury_correct(const float *G, const float *U, const float *V, int64_t B, int64_t N, int64_t r,
                            float *out, void *workspace, size_t workspace_bytes, void *stream);
size_t bkcore_woodbury_workspace_bytes(int64_t B, int64_t N, int64_t r);

/* Largest N bkcore_forward accepts. */
int64_t bkcore_max_seq_len(void);

/* Diagnostics: how bkcore_forward would run a (B, N) problem of operand type in_dtype with `flags` and a state block.
 * cus <= 0: on the current device, with the workgroups per compute unit its runtime reports for the kernel (whole rows
 * if there is no device or no answer: exactly what the dispatch does); cus > 0: what-if for a GPU with that many
 * compute units holding 3 workgroups each (2 for 16-bit operands).  Returns pieces per row: 1 = one workgroup per row
 * (rows that fill the GPU evenly: between 0.95 and 1 times as many rows as it holds workgroups, or more than 1.85
 * times; 16-bit operands whose rows fill rounds of 2 workgroups per compute unit to 90 % or exactly; rows of one or
 * two chunks of 4096); the number of chunks of the row = one workgroup per chunk, the running
 * product handed from chunk to chunk inside the launch (every other shape: fewer rows, or a ragged second round); 2 =
 * two units a row on a grid that must be resident at once (float32 / float64 operands between 1.24 and 1.37 times as
 * many rows as places, e.g. 1024 rows on 256 CUs; never with BKCORE_FLAG_WHOLE_ROWS).
 * *launches: 2 = that launch plus the (normally empty) launch for rows to recompute; 4 = few long rows in scan form
 * (BKCORE_FLAG_SCAN_PATH): piece aggregates / combination / piece outputs / rows to recompute.  No counterpart in the
 * reference. */
int bkcore_forward_plan(int64_t B, int64_t N, int in_dtype, unsigned flags, int cus, int *launches);
/* HIP error code of the last failed launch on this thread (0 if none). */
int bkcore_last_hip_error(void);
/* Static description of an error code returned by this library. */
const char *bkcore_error_string(int code);
/* ABI version, bumped on any signature change. */
int bkcore_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* BKCORE_H */

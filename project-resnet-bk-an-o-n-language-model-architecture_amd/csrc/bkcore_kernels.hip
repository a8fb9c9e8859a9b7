// BK-Core for MI355X (gfx950): diag((T - zI)^-1) of batched tridiagonal matrices.
//
// Semantics follow the reference's sequential log-scaled path
// (src/models/bk_core.py:27-210 of neko-jpg/Project-ResNet-BK-An-O-N-Language-Model-Architecture):
// theta (leading minors) and phi (trailing minors) three-term recurrences with diagonal
// regularisation, the soft clamp inside the recurrence, the clamp of the final log-magnitude and
// the |G| <= 50 clip.  The evaluation strategy is different: each recurrence step is the 2x2
// complex transfer matrix M_p = [[alpha_p, -kappa_p],[1, 0]] and the row is an associative scan.
//
// Number representation: every scan quantity is a float32 mantissa times 2^e with an int32
// exponent (exact power-of-two rescaling), so nothing overflows for any N and the difference of
// the O(N)-sized log-magnitudes in the epilogue is an exact integer subtraction.
//
// The soft clamp inside the recurrence (bk_core.py:127-136) multiplies kappa_p by
// w_p = exp(88 tanh(d/88) - d), d = ln|prev2| - ln|prev|, which makes the recurrence weakly
// non-linear.  Lane aggregates need w_p before the lane's incoming state is known, so they are
// built by a two-pass fixed point: pass 1 scans the linear recurrence and yields d_p (only the
// ratio of consecutive terms matters), pass 2 scans with kappa_p * w_p.  Once a lane's incoming
// state is known the lane can also run the non-linear recurrence itself, exactly as the reference
// does step by step.
//
// Kernel shape: one workgroup of 256 threads per row, chunks of 256 x 16 elements, each thread
// owning 16 consecutive elements of a chunk in registers.
//   sweep A, chunks left -> right : theta scan; stores each lane's incoming theta state
//                                   (20 bytes per 16 elements) to the workspace
//   sweep B, chunks right -> left : theta re-run inside the lane from the stored state, phi scan with
//                                   a running carry, epilogue fused into the last phi re-run,
//                                   G / features staged through LDS for coalesced stores.
// First launch: pass 1 by a verified per-lane warm-up (one scan per chunk and direction); second
// launch: only rows whose verification failed, with pass 1 as a full scan too.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "../../include/bkcore.h"
#include "bkcore_internal.h"

namespace {

#ifndef BK_T
#define BK_T 256
#endif
constexpr int T = BK_T;            // threads per workgroup
#ifndef BK_E
#define BK_E 16
#endif
constexpr int E = BK_E;            // elements per thread per chunk
constexpr int CHUNK = T * E;       // elements per chunk
constexpr int WAVES = T / 64;
#ifndef BK_WAVES_PER_SIMD
#define BK_WAVES_PER_SIMD 2
#endif
constexpr long MAX_N = 1L << 24;
constexpr float LN2_HALF = 0.34657359027997264f;  // 0.5 * ln 2
constexpr float LOG2E = 1.4426950408889634f;
constexpr float WALL_DMAX = 22.0f;                // |d| beyond this is clamped in the series below

struct Vec {  // (x0, x1) * 2^e : two consecutive terms of a recurrence
  float2 x0, x1;
  int e;
};
struct Mat {  // [[a b],[c d]] * 2^e
  float2 a, b, c, d;
  int e;
};

__device__ __forceinline__ float2 cmul(float2 p, float2 q) {
  return make_float2(fmaf(p.x, q.x, -p.y * q.y), fmaf(p.x, q.y, p.y * q.x));
}
// acc + p*q
__device__ __forceinline__ float2 cmad(float2 acc, float2 p, float2 q) {
  return make_float2(fmaf(p.x, q.x, fmaf(-p.y, q.y, acc.x)), fmaf(p.x, q.y, fmaf(p.y, q.x, acc.y)));
}
// al*x - ka*p   (one recurrence step; ka real)
__device__ __forceinline__ float2 rstep(float2 al, float ka, float2 x, float2 p) {
  return make_float2(fmaf(al.x, x.x, fmaf(-al.y, x.y, -ka * p.x)), fmaf(al.x, x.y, fmaf(al.y, x.x, -ka * p.y)));
}
// p / q
__device__ __forceinline__ float2 cdiv(float2 p, float2 q) {
  const float inv = __builtin_amdgcn_rcpf(fmaf(q.x, q.x, q.y * q.y));
  return make_float2(fmaf(p.x, q.x, p.y * q.y) * inv, fmaf(p.y, q.x, -p.x * q.y) * inv);
}
__device__ __forceinline__ float amax2(float2 p) { return fmaxf(fabsf(p.x), fabsf(p.y)); }
__device__ __forceinline__ float2 ld2(float2 p, int e) { return make_float2(ldexpf(p.x, e), ldexpf(p.y, e)); }
// 2^-de as a float, exact for -126 <= de <= 126.  de comes from v_frexp_exp of a mantissa that is renormalised
// every few steps, so it only leaves that range once a state has already overflowed to inf/NaN.
__device__ __forceinline__ float pow2_neg(int de) { return __int_as_float((127 - de) << 23); }
__device__ __forceinline__ float2 sc2(float2 p, float s) { return make_float2(p.x * s, p.y * s); }
// exponent e with m * 2^-e in [0.5,1); 0 for m == 0, inf, nan
__device__ __forceinline__ int expo(float m) { return __builtin_amdgcn_frexp_expf(m); }
__device__ __forceinline__ float log2_norm2(float2 p) { return __builtin_amdgcn_logf(fmaf(p.x, p.x, p.y * p.y)); }

__device__ __forceinline__ void vnorm(Vec &v) {
  const int de = expo(fmaxf(amax2(v.x0), amax2(v.x1)));
  const float s = pow2_neg(de);
  v.x0 = sc2(v.x0, s);
  v.x1 = sc2(v.x1, s);
  v.e += de;
}
__device__ __forceinline__ void mnorm(Mat &m) {
  const int de = expo(fmaxf(fmaxf(amax2(m.a), amax2(m.b)), fmaxf(amax2(m.c), amax2(m.d))));
  const float s = pow2_neg(de);
  m.a = sc2(m.a, s);
  m.b = sc2(m.b, s);
  m.c = sc2(m.c, s);
  m.d = sc2(m.d, s);
  m.e += de;
}
// p * q (p applied after q)
__device__ __forceinline__ Mat matmul(const Mat &p, const Mat &q) {
  Mat r;
  r.a = cmad(cmul(p.a, q.a), p.b, q.c);
  r.b = cmad(cmul(p.a, q.b), p.b, q.d);
  r.c = cmad(cmul(p.c, q.a), p.d, q.c);
  r.d = cmad(cmul(p.c, q.b), p.d, q.d);
  r.e = p.e + q.e;
  mnorm(r);
  return r;
}
__device__ __forceinline__ Vec matvec(const Mat &p, const Vec &v) {
  Vec r;
  r.x0 = cmad(cmul(p.a, v.x0), p.b, v.x1);
  r.x1 = cmad(cmul(p.c, v.x0), p.d, v.x1);
  r.e = p.e + v.e;
  vnorm(r);
  return r;
}
__device__ __forceinline__ Mat mat_identity() {
  Mat m;
  m.a = make_float2(1.f, 0.f);
  m.b = make_float2(0.f, 0.f);
  m.c = make_float2(0.f, 0.f);
  m.d = make_float2(1.f, 0.f);
  m.e = 0;
  return m;
}
__device__ __forceinline__ Vec vec_start() {  // (1, 0): theta_0 = 1, theta_{-1} = 0 (and phi_{n-1} = 1, phi_n = 0)
  Vec v;
  v.x0 = make_float2(1.f, 0.f);
  v.x1 = make_float2(0.f, 0.f);
  v.e = 0;
  return v;
}
// p * q without renormalisation (exponents add); callers bound the growth
__device__ __forceinline__ Mat matmul_raw(const Mat &p, const Mat &q) {
  Mat r;
  r.a = cmad(cmul(p.a, q.a), p.b, q.c);
  r.b = cmad(cmul(p.a, q.b), p.b, q.d);
  r.c = cmad(cmul(p.c, q.a), p.d, q.c);
  r.d = cmad(cmul(p.c, q.b), p.d, q.d);
  r.e = p.e + q.e;
  return r;
}
__device__ __forceinline__ Vec matvec_raw(const Mat &p, const Vec &v) {
  Vec r;
  r.x0 = cmad(cmul(p.a, v.x0), p.b, v.x1);
  r.x1 = cmad(cmul(p.c, v.x0), p.d, v.x1);
  r.e = p.e + v.e;
  return r;
}

// Cross-lane moves by DPP (v_mov_b32_dpp): lanes without a source keep `old`.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f(float old, float src) {
  return __int_as_float(
      __builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(src), CTRL, ROW_MASK, 0xF, false));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_i(int old, int src) {
  return __builtin_amdgcn_update_dpp(old, src, CTRL, ROW_MASK, 0xF, false);
}
// The matrix of the lane selected by CTRL; identity where the selection has no source lane.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ Mat mat_dpp(const Mat &m) {
  Mat r;
  r.a.x = dpp_f<CTRL, ROW_MASK>(1.0f, m.a.x);
  r.a.y = dpp_f<CTRL, ROW_MASK>(0.0f, m.a.y);
  r.b.x = dpp_f<CTRL, ROW_MASK>(0.0f, m.b.x);
  r.b.y = dpp_f<CTRL, ROW_MASK>(0.0f, m.b.y);
  r.c.x = dpp_f<CTRL, ROW_MASK>(0.0f, m.c.x);
  r.c.y = dpp_f<CTRL, ROW_MASK>(0.0f, m.c.y);
  r.d.x = dpp_f<CTRL, ROW_MASK>(1.0f, m.d.x);
  r.d.y = dpp_f<CTRL, ROW_MASK>(0.0f, m.d.y);
  r.e = dpp_i<CTRL, ROW_MASK>(0, m.e);
  return r;
}
constexpr int DPP_ROW_SHR1 = 0x111, DPP_ROW_SHR2 = 0x112, DPP_ROW_SHR4 = 0x114, DPP_ROW_SHR8 = 0x118;
constexpr int DPP_ROW_BCAST15 = 0x142, DPP_ROW_BCAST31 = 0x143, DPP_WAVE_SHR1 = 0x138;

// exp(88 tanh(d/88) - d): the reference's soft clamp exp(88 tanh(d/88)) (bk_core.py:60-75) relative to exp(d).
// Branch-free: tanh series to u^11 (relative error of the exponent < 2e-6 for |d| <= 22); |d| is clamped to 22
// (a ratio of consecutive terms beyond e^22 is below float32 resolution of the recurrence).  A non-finite d
// (an exactly zero term, e.g. the start vector's theta_{-1} = 0) gives 1.
__device__ __forceinline__ float wall_exponent(float d) {
  const float dc = __builtin_amdgcn_fmed3f(d, -WALL_DMAX, WALL_DMAX);  // a NaN d comes out as a bound, i.e. finite
  const float u = dc * (1.0f / 88.0f);
  const float s = u * u;
  float p = fmaf(s, 1382.0f / 155925.0f, -62.0f / 2835.0f);
  p = fmaf(s, p, 17.0f / 315.0f);
  p = fmaf(s, p, -2.0f / 15.0f);
  p = fmaf(s, p, 1.0f / 3.0f);
  return -88.0f * u * s * p;
}
// (a non-finite d only arises next to an exactly zero term, whose product with the factor is zero anyway;
//  the clamp keeps the factor finite)
__device__ __forceinline__ float wall_corr(float d) { return __builtin_amdgcn_exp2f(wall_exponent(d) * LOG2E); }

constexpr int A_PAD = CHUNK + CHUNK / 32 + 8;
constexpr int O_PAD = CHUNK + CHUNK / 16 + 8;

struct Shared {
  union {
    struct {
      float a_s[A_PAD];       // staged diagonal, index idx + idx/32
      float k_s[A_PAD + 4];   // staged k_{i-1} = b_{i-1} c_{i-1} for i = chunk_base .. chunk_base+CHUNK (CHUNK+1 values)
    } in;
    float2 out_s[O_PAD];      // G of one chunk, index idx + idx/16, for coalesced stores
  } u;
  Mat tot[2][WAVES];          // per-wave aggregates, double-buffered across scans
  Vec carry[2][2];            // running chunk carries (pass 1 / pass 2), double-buffered across chunks
  Vec warm[WAVES];            // last lane's warm-up state of each wave (fast path hand-off to the next wave)
};

// Which LDS slots the next scan uses: per-wave totals alternate every scan, each carry alternates every use.
struct ScanCtx {
  int nscan;
  int cpar[2];
};

__device__ __forceinline__ int pad32(int idx) { return idx + (idx >> 5); }
__device__ __forceinline__ int pad16(int idx) { return idx + (idx >> 4); }

// ---- lane-level serial pieces -------------------------------------------------------------------------------
// Element order inside a lane: ascending for theta, descending (REV) for phi.  kw holds k_{i-1} for the lane's
// elements plus one more (k of the lane's last element); KO selects which one a step uses.

// One more element folded into the running product A <- M(al, k) * A.
__device__ __forceinline__ void agg_step(Mat &A, float2 al, float k) {
  const float2 na = rstep(al, k, A.a, A.c);
  const float2 nb = rstep(al, k, A.b, A.d);
  A.c = A.a;
  A.d = A.b;
  A.a = na;
  A.b = nb;
}

// Pass 1: product of the lane's E transfer matrices (linear recurrence, plain k) in processing order.
// Renormalised every 4 elements: one step grows an entry by at most |alpha| + |k| (2^7.7 for clamped inputs).
template <bool REV, bool TAIL, int KO>
__device__ __forceinline__ Mat lane_aggregate(const float2 (&al)[E], const float (&kw)[E + 1], unsigned vmask) {
  Mat A = mat_identity();
#pragma unroll
  for (int j = 0; j < E; ++j) {
    const int ei = REV ? E - 1 - j : j;
    if (!TAIL || ((vmask >> ei) & 1u)) agg_step(A, al[ei], kw[ei + KO]);
    if ((j & 3) == 3) mnorm(A);
  }
  return A;
}

// Pass 2 set-up, three independent chains in one loop: the linear recurrence re-run from the lane's incoming
// pass-1 state gives d_p and so the wall factor; k2 = k * w is recorded and folded straight into the pass-2
// product of the lane.
template <bool REV, bool TAIL, int KO>
__device__ __forceinline__ Mat wall_and_aggregate(const float2 (&al)[E], const float (&kw)[E + 1], unsigned vmask,
                                                  Vec x, float (&k2)[E]) {
  Mat A = mat_identity();
  float l0 = log2_norm2(x.x0);
  float l1 = log2_norm2(x.x1);
#pragma unroll
  for (int j = 0; j < E; ++j) {
    const int ei = REV ? E - 1 - j : j;
    const float d = LN2_HALF * (l1 - l0);  // ln|prev2| - ln|prev|
    const float kk = kw[ei + KO] * wall_corr(d);
    k2[ei] = kk;
    if (!TAIL || ((vmask >> ei) & 1u)) {
      const float2 y = rstep(al[ei], kw[ei + KO], x.x0, x.x1);
      x.x1 = x.x0;
      x.x0 = y;
      l1 = l0;
      agg_step(A, al[ei], kk);
      if ((j & 1) == 1) {
        const int de = expo(fmaxf(amax2(x.x0), amax2(x.x1)));
        const float s = pow2_neg(de);
        x.x0 = sc2(x.x0, s);
        x.x1 = sc2(x.x1, s);
        l1 -= 2.0f * (float)de;
      }
      l0 = log2_norm2(x.x0);
    }
    if ((j & 3) == 3) mnorm(A);
  }
  return A;
}

// ---- fast path for pass 1 -----------------------------------------------------------------------------------
// The wall factor needs only the RATIO of consecutive terms, to about 1e-3.  Wherever the ratio map of the
// recurrence is contracting (every well-conditioned input), a lane's outgoing ratio forgets its incoming one
// within a few steps: each lane runs the linear recurrence over its own 16 elements from the start vector and
// hands the result to the next lane as that lane's pass-1 state; no pass-1 product and no pass-1 scan.
// The estimate is checked afterwards against the exact pass-2 state (ratio_differs); a row with any mismatch is
// recomputed by the two-scan path in the second launch.
template <bool REV, bool TAIL, int KO>
__device__ __forceinline__ Vec warm_run(const float2 (&al)[E], const float (&kw)[E + 1], unsigned vmask) {
  Vec x = vec_start();
#pragma unroll
  for (int j = 0; j < E; ++j) {
    const int ei = REV ? E - 1 - j : j;
    if (!TAIL || ((vmask >> ei) & 1u)) {
      const float2 y = rstep(al[ei], kw[ei + KO], x.x0, x.x1);
      x.x1 = x.x0;
      x.x0 = y;
    }
    if ((j & 3) == 3) vnorm(x);
  }
  return x;
}

template <bool REV, bool TAIL, int KO>
__device__ __forceinline__ Vec estimate_incoming(const float2 (&al)[E], const float (&kw)[E + 1], unsigned vmask,
                                                 Shared &sh, const ScanCtx &ctx) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const Vec x = warm_run<REV, TAIL, KO>(al, kw, vmask);
  Vec est;
  est.x0.x = dpp_f<DPP_WAVE_SHR1, 0xF>(1.0f, x.x0.x);
  est.x0.y = dpp_f<DPP_WAVE_SHR1, 0xF>(0.0f, x.x0.y);
  est.x1.x = dpp_f<DPP_WAVE_SHR1, 0xF>(0.0f, x.x1.x);
  est.x1.y = dpp_f<DPP_WAVE_SHR1, 0xF>(0.0f, x.x1.y);
  est.e = 0;
  if (lane == 63) sh.warm[wave] = x;
  __syncthreads();
  if (lane == 0) {
    // first lane of the chunk: the exact running (pass-2) carry; first lane of a later wave: previous wave's last lane
    est = (wave == 0) ? sh.carry[1][ctx.cpar[1]] : sh.warm[wave > 0 ? wave - 1 : 0];
  }
  return est;
}

// |est.x0/est.x1 - x.x0/x.x1| > tol |x.x0/x.x1| without divisions (false when both second terms are zero)
__device__ __forceinline__ bool ratio_differs(const Vec &est, const Vec &x) {
  const float2 c = cmul(est.x0, x.x1);
  const float2 d = cmul(x.x0, est.x1);
  const float dx = c.x - d.x, dy = c.y - d.y;
  const float lhs = fmaf(dx, dx, dy * dy);
  const float rhs = 4.0e-6f * fmaf(d.x, d.x, d.y * d.y);  // tol = 2e-3
  return !(lhs <= rhs);
}

// theta inside one lane from its stored incoming (pass-2) state, two chains in lockstep: a linear one that
// only supplies d_p for the wall factor (kept off the critical path) and the wall-corrected one whose terms
// are the outputs.  Both start from the same state (their ratio of consecutive terms agrees to ~1e-5, far
// tighter than the wall factor needs) and share the rescaling exponent.  The mantissas go to the lane's 16 slots
// of the LDS output tile (`om`, consecutive), where the epilogue later replaces each by G in place; exponents
// are shared by element pairs and stay in registers.
template <bool TAIL, bool SAVE>
__device__ __forceinline__ void theta_lane(const float2 (&al)[E], const float (&kw)[E + 1], unsigned vmask, Vec x,
                                           float2 *__restrict__ om, int (&oe)[E / 2], float2 *__restrict__ save_r) {
  float2 u0 = x.x0, u1 = x.x1;  // linear chain
  float l0 = log2_norm2(u0);
  float l1 = log2_norm2(u1);
#pragma unroll
  for (int j = 0; j < E; ++j) {
    om[j] = x.x0;
    if ((j & 1) == 0) oe[j >> 1] = x.e;
    const float d = LN2_HALF * (l1 - l0);
    const float kk = kw[j] * wall_corr(d);
    if (!TAIL || ((vmask >> j) & 1u)) {
      const float2 yu = rstep(al[j], kw[j], u0, u1);
      const float2 y = rstep(al[j], kk, x.x0, x.x1);
      if (SAVE) save_r[j] = cdiv(y, x.x0);  // r_j = theta_{j+1} / theta_j, kept for the analytic backward
      u1 = u0;
      u0 = yu;
      x.x1 = x.x0;
      x.x0 = y;
      l1 = l0;
      if ((j & 1) == 1) {
        const int de = expo(fmaxf(amax2(x.x0), amax2(x.x1)));
        const float s = pow2_neg(de);
        x.x0 = sc2(x.x0, s);
        x.x1 = sc2(x.x1, s);
        u0 = sc2(u0, s);
        u1 = sc2(u1, s);
        x.e += de;
        l1 -= 2.0f * (float)de;
      }
      l0 = log2_norm2(u0);
    }
  }
}

// ---- workgroup-level exclusive scan of lane aggregates ------------------------------------------------------
// Lanes are scanned in ascending thread order.  Returns this lane's incoming state (product of all earlier
// lanes' aggregates applied to the running carry `which`, kept in LDS) and advances that carry across the chunk.
// Scans alternate between carry 0 (pass 1) and carry 1 (pass 2).
// Wave level: Kogge-Stone over DPP moves (row_shr 1,2,4,8 inside 16-lane rows, then row_bcast 15 / 31 carry the
// row totals upward); lanes without a source combine with the identity.  Normalised inputs (largest component
// in [0.5,1)) let two products go by between renormalisations: |entry| <= 2^0.5 -> 2^2 -> 2^5.
__device__ __forceinline__ Vec block_exclusive(Mat A, int which, Shared &sh, ScanCtx &ctx) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  A = matmul_raw(A, mat_dpp<DPP_ROW_SHR1, 0xF>(A));
  A = matmul(A, mat_dpp<DPP_ROW_SHR2, 0xF>(A));
  A = matmul_raw(A, mat_dpp<DPP_ROW_SHR4, 0xF>(A));
  A = matmul(A, mat_dpp<DPP_ROW_SHR8, 0xF>(A));
  A = matmul_raw(A, mat_dpp<DPP_ROW_BCAST15, 0xA>(A));
  A = matmul(A, mat_dpp<DPP_ROW_BCAST31, 0xC>(A));
  // a carry's two slots alternate so that the new value never overwrites one that a slower wave may still
  // be reading (at least one other barrier lies between two uses of the same carry)
  const int buf = ctx.nscan & 1;
  const int par = ctx.cpar[which];
  ctx.nscan++;
  ctx.cpar[which] ^= 1;
  if (lane == 63) sh.tot[buf][wave] = A;
  const Mat ex = mat_dpp<DPP_WAVE_SHR1, 0xF>(A);  // exclusive prefix inside the wave (identity in lane 0)
  __syncthreads();
  Vec v = sh.carry[which][par];
  Vec xin = v;
#pragma unroll
  for (int w = 0; w < WAVES; ++w) {
    if (w == wave) xin = v;
    v = matvec_raw(sh.tot[buf][w], v);  // four un-normalised products grow a component by at most 2^8
  }
  if (threadIdx.x == 0) {
    vnorm(v);
    sh.carry[which][par ^ 1] = v;
  }
  return matvec(ex, xin);
}

// ---- input staging ------------------------------------------------------------------------------------------
template <typename TIn>
__device__ __forceinline__ float to_f32(TIn v);
template <>
__device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <>
__device__ __forceinline__ float to_f32<double>(double v) { return (float)v; }
template <>
__device__ __forceinline__ float to_f32<uint16_t>(uint16_t v) { return __uint_as_float(((unsigned)v) << 16); }  // bf16
template <>
__device__ __forceinline__ float to_f32<_Float16>(_Float16 v) { return (float)v; }

__device__ __forceinline__ float clamp_keep_nan(float v, float lim) { return v < -lim ? -lim : (v > lim ? lim : v); }

struct RowCfg {
  float2 z;
  long N;
  bool raw;     // no clamp / sanitise
  bool fix_bc;  // replace NaN in b/c by 1 (second launch when he_diag held a NaN)
  float fclamp; // the features output is clamp(Re G, Im G; +-fclamp) (+inf: no clamp); G itself is never clamped
};

// Stages chunk at `base` of one row into LDS: a_s[idx] = a[base+idx], k_s[idx] = b[g] c[g] with g = base-1+idx
// (k_s holds CHUNK+1 values; indices outside the row give a = 1, k = 0).
// FULL: every index idx < CHUNK of all three operands lies inside the row (interior chunks): no bounds logic.
// Otherwise the loads use clamped (always valid) indices and out-of-row values are replaced afterwards.
// Returns bit0: NaN seen in a, bit1: NaN seen in b or c.
template <typename TIn, bool FULL>
__device__ __forceinline__ unsigned stage_chunk(const TIn *__restrict__ a, const TIn *__restrict__ b,
                                                const TIn *__restrict__ c, long base, const RowCfg &cfg, Shared &sh) {
  unsigned seen = 0;
  const int t = threadIdx.x;
  const TIn *__restrict__ ap = a + base;
  const int a_hi = (int)min((long)CHUNK, cfg.N - base) - 1;  // last valid idx for a (>= 0)
  {
    float av[E];
#pragma unroll
    for (int j = 0; j < E; ++j) {
      const int idx = t + j * T;
      if (FULL) {
        av[j] = to_f32<TIn>(ap[idx]);
      } else {
        const float v = to_f32<TIn>(ap[min(idx, a_hi)]);
        av[j] = idx <= a_hi ? v : 1.0f;
      }
    }
    float probe = 0.0f;  // NaN iff some value is NaN (or +inf and -inf meet: handled exactly below as well)
#pragma unroll
    for (int j = 0; j < E; ++j) probe += av[j];
    if (cfg.raw) {
#pragma unroll
      for (int j = 0; j < E; ++j) sh.u.in.a_s[pad32(t + j * T)] = av[j];
    } else if (__builtin_expect(!__any(probe != probe), 1)) {
#pragma unroll
      for (int j = 0; j < E; ++j) sh.u.in.a_s[pad32(t + j * T)] = __builtin_amdgcn_fmed3f(av[j], -100.0f, 100.0f);
    } else {
#pragma unroll
      for (int j = 0; j < E; ++j) {
        float v = clamp_keep_nan(av[j], 100.0f);
        if (v != v) { v = 0.0f; seen |= 1u; }
        sh.u.in.a_s[pad32(t + j * T)] = v;
      }
    }
  }
  // b, c: k_s[idx] needs global index g = base - 1 + idx, valid for 0 <= g <= N-2
  const int k_lo = base == 0 ? 1 : 0;
  const int k_hi = (int)min((long)CHUNK, cfg.N - 1 - base);  // may be < k_lo (N == 1)
  if (FULL || k_hi >= k_lo) {
    const TIn *__restrict__ bp = b + (base - 1);
    const TIn *__restrict__ cp = c + (base - 1);
    float bv[E], cv[E];
#pragma unroll
    for (int j = 0; j < E; ++j) {
      const int idx = t + j * T;
      if (FULL) {
        bv[j] = to_f32<TIn>(bp[idx]);
        cv[j] = to_f32<TIn>(cp[idx]);
      } else {
        const int ii = min(max(idx, k_lo), k_hi);
        const float p = to_f32<TIn>(bp[ii]);
        const float q = to_f32<TIn>(cp[ii]);
        const bool ok = idx >= k_lo && idx <= k_hi;
        bv[j] = ok ? p : 0.0f;
        cv[j] = ok ? q : 0.0f;
      }
    }
    // the one extra value k_s[CHUNK] (k of the chunk's last element), loaded by thread 0
    float be = 0.0f, ce = 0.0f;
    if (t == 0 && k_hi >= CHUNK) {
      be = to_f32<TIn>(bp[CHUNK]);
      ce = to_f32<TIn>(cp[CHUNK]);
    }
    float probe = be + ce;
#pragma unroll
    for (int j = 0; j < E; ++j) probe += bv[j] + cv[j];
    if (cfg.raw) {
#pragma unroll
      for (int j = 0; j < E; ++j) sh.u.in.k_s[pad32(t + j * T)] = bv[j] * cv[j];
      if (t == 0) sh.u.in.k_s[pad32(CHUNK)] = be * ce;
    } else if (__builtin_expect(!__any(probe != probe), 1)) {
#pragma unroll
      for (int j = 0; j < E; ++j)
        sh.u.in.k_s[pad32(t + j * T)] =
            __builtin_amdgcn_fmed3f(bv[j], -10.0f, 10.0f) * __builtin_amdgcn_fmed3f(cv[j], -10.0f, 10.0f);
      if (t == 0)
        sh.u.in.k_s[pad32(CHUNK)] = __builtin_amdgcn_fmed3f(be, -10.0f, 10.0f) * __builtin_amdgcn_fmed3f(ce, -10.0f, 10.0f);
    } else {
#pragma unroll
      for (int j = 0; j <= E; ++j) {
        if (j < E || t == 0) {
          float p = clamp_keep_nan(j < E ? bv[j] : be, 10.0f), q = clamp_keep_nan(j < E ? cv[j] : ce, 10.0f);
          if (p != p) { seen |= 2u; if (cfg.fix_bc) p = 1.0f; }
          if (q != q) { seen |= 2u; if (cfg.fix_bc) q = 1.0f; }
          sh.u.in.k_s[pad32(t + j * T)] = p * q;
        }
      }
    }
  } else {
#pragma unroll
    for (int j = 0; j <= E; ++j)
      if (j < E || t == 0) sh.u.in.k_s[pad32(t + j * T)] = 0.0f;
  }
  return seen;
}

// Reads the thread's E elements back in blocked order and forms alpha = reg(a - z) (bk_core.py:78-94) and
// kw[e] = k_{i-1} for i = first element + e, e = 0..E.
// `seg` is the 16-element segment of the chunk this thread owns (thread t in sweep A, T-1-t in sweep B, so that
// the phi scan also runs over ascending threads).
__device__ __forceinline__ void fetch_elems(const Shared &sh, long base, int seg, const RowCfg &cfg, float2 (&al)[E],
                                            float (&kw)[E + 1], unsigned &vmask) {
#pragma unroll
  for (int e = 0; e <= E; ++e) kw[e] = sh.u.in.k_s[pad32(seg * E + e)];
  const long first = base + (long)seg * E;
  const long left = cfg.N - first;
  vmask = left >= E ? ((1u << E) - 1u) : (left <= 0 ? 0u : ((1u << (int)left) - 1u));
  const float xi = -cfg.z.y;
  if (xi * xi >= 1e-24f) {
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const float xr = sh.u.in.a_s[pad32(seg * E + e)] - cfg.z.x;
      const float f = fmaf(1e-3f, __builtin_amdgcn_rsqf(fmaf(xr, xr, xi * xi)), 1.0f);
      al[e] = make_float2(xr * f, xi * f);
    }
  } else {
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const float xr = sh.u.in.a_s[pad32(seg * E + e)] - cfg.z.x;
      const float n = fmaf(xr, xr, xi * xi);
      float f = fmaf(1e-3f, __builtin_amdgcn_rsqf(n), 1.0f);
      float add = 0.0f;
      if (!(n >= 1e-24f)) { f = 1.0f; add = 1e-3f; }  // |x| < 1e-12: direction taken as 1 (bk_core.py:89-90)
      al[e] = make_float2(fmaf(xr, f, add), xi * f);
    }
  }
}

// ---- the forward kernel -------------------------------------------------------------------------------------
// lane-state storage: 5 floats per (chunk, thread), laid out [chunk][field][thread]
__device__ __forceinline__ void put_state(float *st, int ch, int seg, const Vec &x) {
  float *p = st + (long)ch * 5 * T + seg;
  p[0] = x.x0.x;
  p[T] = x.x0.y;
  p[2 * T] = x.x1.x;
  p[3 * T] = x.x1.y;
  p[4 * T] = __int_as_float(x.e);
}
__device__ __forceinline__ Vec get_state(const float *st, int ch, int seg) {
  const float *p = st + (long)ch * 5 * T + seg;
  Vec x;
  x.x0.x = p[0];
  x.x0.y = p[T];
  x.x1.x = p[2 * T];
  x.x1.y = p[3 * T];
  x.e = __float_as_int(p[4 * T]);
  return x;
}

// Pass 1 + pass 2 of one chunk in one direction: fills k2 (wall-corrected k per element) and returns the lane's
// exact incoming pass-2 state.  FAST: pass 1 by the verified warm-up estimate instead of a product + scan.
template <bool FAST, bool REV, bool TAIL, int KO>
__device__ __forceinline__ Vec two_pass(const float2 (&al)[E], const float (&kw)[E + 1], unsigned vmask,
                                        float (&k2)[E], Shared &sh, ScanCtx &ctx, bool &bad) {
  Vec est;
  if (FAST) {
    est = estimate_incoming<REV, TAIL, KO>(al, kw, vmask, sh, ctx);
  } else {
    const Mat A1 = lane_aggregate<REV, TAIL, KO>(al, kw, vmask);
    est = block_exclusive(A1, 0, sh, ctx);
  }
  const Mat A2 = wall_and_aggregate<REV, TAIL, KO>(al, kw, vmask, est, k2);
  const Vec x = block_exclusive(A2, 1, sh, ctx);
  // (a lane past the end of the row has no element that uses its estimate)
  if (FAST) bad |= (!TAIL || vmask != 0u) && ratio_differs(est, x);
  return x;
}

template <bool FAST, bool TAIL, int KO_PHI, bool SAVE>
__device__ __forceinline__ void sweepB_chunk(const float2 (&al)[E], const float (&kw)[E + 1], unsigned vmask,
                                             const Vec &xt, float2 inv_det, int e_det, Shared &sh, ScanCtx &ctx,
                                             int seg, bool &bad, float2 *__restrict__ save_r,
                                             float2 *__restrict__ save_s) {
  float k2[E];
  Vec x = two_pass<FAST, true, TAIL, KO_PHI>(al, kw, vmask, k2, sh, ctx, bad);
  // theta of the lane's elements into the lane's slots of the output tile.  (All reads of the staged inputs
  // happened before the scans' barriers, so the tile may overwrite them; each slot is touched by this thread only.)
  float2 *slot = &sh.u.out_s[pad16(seg * E)];
  int the[E / 2];
  theta_lane<TAIL, SAVE>(al, kw, vmask, xt, slot, the, save_r);
  // last phi re-run fused with the epilogue (bk_core.py:190-208): theta_i in the slot is replaced by G_i
#pragma unroll
  for (int j = 0; j < E; ++j) {
    const int ei = E - 1 - j;
    const float2 P = cmul(cmul(slot[ei], x.x0), inv_det);
    const int Ee = the[ei >> 1] + x.e - e_det;
    const float n = fmaf(P.x, P.x, P.y * P.y);
    const float L = LN2_HALF * fmaf(2.0f, (float)Ee, __builtin_amdgcn_logf(n));  // ln |G_raw|
    float2 g;
    if (SAVE) {
      // for the analytic backward: G_raw itself (no log-magnitude clamp, no clip); 0 outside 1e-18 <= |G_raw| <= 1e18
      g = (fabsf(L) <= 41.4f) ? make_float2(ldexpf(P.x, Ee), ldexpf(P.y, Ee)) : make_float2(0.f, 0.f);
    } else {
      const float sc = ldexpf(__builtin_amdgcn_exp2f(wall_exponent(L) * LOG2E), Ee);
      g = make_float2(P.x * sc, P.y * sc);
    }
    const float m2 = fmaf(g.x, g.x, g.y * g.y);
    if (!SAVE && __builtin_expect(!(m2 <= 2500.0f), 0)) {
      // |G| > 50: clipped to 50 G/|G| (bk_core.py:205-208; also when 2^Ee overflowed although G_raw is finite:
      // exp(88 tanh(L/88)) > 50 for every L > 4); anything that is not a number: 0 (bk_core.py:204)
      const float u = 50.0f * __builtin_amdgcn_rsqf(n);
      g = (L > 3.9f && L <= 3.0e38f && u <= 3.0e38f) ? make_float2(P.x * u, P.y * u) : make_float2(0.f, 0.f);
    }
    slot[ei] = g;
    if (!TAIL || ((vmask >> ei) & 1u)) {
      const float2 y = rstep(al[ei], k2[ei], x.x0, x.x1);
      if (SAVE) save_s[ei] = cdiv(y, x.x0);  // s_i = phi_i / phi_{i+1} (phi_i = minor of rows i..N-1)
      x.x1 = x.x0;
      x.x0 = y;
      if ((j & 1) == 1) vnorm(x);
    }
  }
}

// |G| below exp(-22 + wall_exponent(-22)): the series in the epilogue clamped L at -22.  Redo the magnitude
// with the direct formula (only reachable for |G| < 4.4e-10).
__device__ __forceinline__ float2 fix_tiny(float2 g) {
  const float m2 = fmaf(g.x, g.x, g.y * g.y);
  if (__builtin_expect(m2 < 1.9e-19f && m2 > 0.0f, 0)) {
    const float g22 = wall_exponent(-WALL_DMAX);
    const float L = 0.5f * logf(m2) - g22;
    const float corr = expf(88.0f * tanhf(L * (1.0f / 88.0f)) - L - g22);
    g = make_float2(g.x * corr, g.y * corr);
  }
  return g;
}

// ---- the two sweeps over a range of chunks ---------------------------------------------------------------------
// Three kinds of chunk: the row's first (no k to its left), interior ones (every load in range: no bounds logic),
// and the row's last (ragged end).  Each kind is handled outside the loop over interior chunks, so that loop's body
// holds a single variant of the staging and scan code.
template <bool V>
using BoolC = std::integral_constant<bool, V>;

// Sweep A over chunks [ch0, ch1), left -> right: theta.  The running carries (sh.carry[*][ctx.cpar[*]]) must hold
// the state entering chunk ch0; on return they hold the state after chunk ch1 - 1 (after a barrier).
template <typename TIn, bool FAST>
__device__ __forceinline__ void sweepA_range(const TIn *__restrict__ a, const TIn *__restrict__ b,
                                             const TIn *__restrict__ c, const RowCfg &cfg, Shared &sh, ScanCtx &ctx,
                                             float *__restrict__ st, int ch0, int ch1, unsigned &seen, bool &bad) {
  const int t = threadIdx.x;
  const int nch = (int)((cfg.N + CHUNK - 1) / CHUNK);
  auto one = [&](auto full, auto tail, int ch) __attribute__((always_inline)) {
    constexpr bool FULL = decltype(full)::value, TAIL = decltype(tail)::value;
    const long base = (long)ch * CHUNK;
    float2 al[E];
    float kw[E + 1];
    unsigned vmask;
    __syncthreads();  // previous chunk's LDS reads done
    seen |= stage_chunk<TIn, FULL>(a, b, c, base, cfg, sh);
    __syncthreads();
    fetch_elems(sh, base, t, cfg, al, kw, vmask);
    float k2[E];
    const Vec x = two_pass<FAST, false, TAIL, 0>(al, kw, vmask, k2, sh, ctx, bad);
    put_state(st, ch, t, x);
  };
  int ch = ch0;
  if (ch < ch1 && ch == 0 && nch > 1) {
    one(BoolC<false>{}, BoolC<false>{}, ch);
    ++ch;
  }
  const int interior_end = min(ch1, nch - 1);
  for (; ch < interior_end; ++ch) one(BoolC<true>{}, BoolC<false>{}, ch);
  if (ch < ch1) one(BoolC<false>{}, BoolC<true>{}, ch);  // the row's last chunk
}

// Sweep B over chunks [ch0, ch1), right -> left: phi with running carry, theta from the stored lane states,
// epilogue with the factor inv_det * 2^-e_det, coalesced stores of G and features.
template <typename TIn, bool FAST, int KO_PHI, bool SAVE>
__device__ __forceinline__ void sweepB_range(const TIn *__restrict__ a, const TIn *__restrict__ b,
                                             const TIn *__restrict__ c, const RowCfg &cfg, Shared &sh, ScanCtx &ctx,
                                             const float *__restrict__ st, int ch0, int ch1, float2 inv_det, int e_det,
                                             float2 *__restrict__ G, float2 *__restrict__ F,
                                             float2 *__restrict__ save_r, float2 *__restrict__ save_s, bool &bad) {
  const int t = threadIdx.x;
  const long N = cfg.N;
  const int nch = (int)((N + CHUNK - 1) / CHUNK);
  auto one = [&](auto full, auto tail, int ch) __attribute__((always_inline)) {
    constexpr bool FULL = decltype(full)::value, TAIL = decltype(tail)::value;
    const long base = (long)ch * CHUNK;
    float2 al[E];
    float kw[E + 1];
    unsigned vmask;
    __syncthreads();  // previous chunk's out_s reads done (and lane states visible)
    stage_chunk<TIn, FULL>(a, b, c, base, cfg, sh);
    __syncthreads();
    const int seg = T - 1 - t;  // descending segments over ascending threads: phi's processing order
    fetch_elems(sh, base, seg, cfg, al, kw, vmask);
    const Vec xt = get_state(st, ch, seg);
    float2 *sr = SAVE ? save_r + base + (long)seg * E : nullptr;
    float2 *ss = SAVE ? save_s + base + (long)seg * E : nullptr;
    sweepB_chunk<FAST, TAIL, KO_PHI, SAVE>(al, kw, vmask, xt, inv_det, e_det, sh, ctx, seg, bad, sr, ss);
    __syncthreads();
    // coalesced stores of G and features (same numbers, bk_core.py:296-298)
    float2 *__restrict__ Gc = G + base;
    float2 *__restrict__ Fc = F ? F + base : nullptr;
    const int n_here = TAIL ? (int)min((long)CHUNK, N - base) : CHUNK;
#pragma unroll
    for (int j = 0; j < E; ++j) {
      const int idx = t + j * T;
      if (!TAIL || idx < n_here) {
        const float2 g = SAVE ? sh.u.out_s[pad16(idx)] : fix_tiny(sh.u.out_s[pad16(idx)]);
        Gc[idx] = g;
        if (Fc)
          Fc[idx] = make_float2(__builtin_amdgcn_fmed3f(g.x, -cfg.fclamp, cfg.fclamp),
                                __builtin_amdgcn_fmed3f(g.y, -cfg.fclamp, cfg.fclamp));
      }
    }
  };
  int ch = ch1 - 1;
  if (ch >= ch0 && ch == nch - 1) {  // the row's last chunk
    one(BoolC<false>{}, BoolC<true>{}, ch);
    --ch;
  }
  const int interior_begin = max(ch0, 1);
  for (; ch >= interior_begin; --ch) one(BoolC<true>{}, BoolC<false>{}, ch);
  if (ch >= ch0 && ch == 0) one(BoolC<false>{}, BoolC<false>{}, 0);
}

// Workspace layout: B ints (per-row verdict of the fast path), padded to 256 bytes, then the lane states.
__host__ __device__ inline size_t rowflag_bytes(long B) { return (((size_t)B * sizeof(int)) + 255) & ~(size_t)255; }

// FAST = first launch (fast pass 1, verified); !FAST = second launch: rows whose verification failed, or every row
// when he_diag held a NaN and b/c held a NaN (then with b/c NaNs replaced by 1, bk_core.py:260-264).
template <typename TIn, bool PHI_EXACT, bool FAST, bool SAVE>
__global__ __launch_bounds__(T, BK_WAVES_PER_SIMD) void bk_forward_kernel(
    const TIn *__restrict__ a_, long lda, const TIn *__restrict__ b_, long ldb, const TIn *__restrict__ c_, long ldc,
    const float *__restrict__ z_dev, float z_re, float z_im, long B, long N, float2 *__restrict__ G_,
    float2 *__restrict__ F_, char *__restrict__ ws, int *state, int call_id, int second, unsigned flags, float fclamp,
    float2 *__restrict__ save_r_, float2 *__restrict__ save_s_) {
  __shared__ Shared sh;
  const long row = blockIdx.x;
  int *rowflag = reinterpret_cast<int *>(ws);
  bool fix_bc = false;
  if (second) {
    const int sa = __hip_atomic_load(&state[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int sb = __hip_atomic_load(&state[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int sc = __hip_atomic_load(&state[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    fix_bc = (sa == call_id && sb == call_id) && !(flags & BKCORE_FLAG_RAW);
    const bool redo_row = (sc == call_id) && rowflag[row] != 0;
    if (!fix_bc && !redo_row) return;
  }
  const TIn *a = a_ + row * lda;
  const TIn *b = b_ + row * ldb;
  const TIn *c = c_ + row * ldc;
  float2 *G = G_ + row * N;
  float2 *F = F_ ? F_ + row * N : nullptr;
  const int nch = (int)((N + CHUNK - 1) / CHUNK);
  float *st = reinterpret_cast<float *>(ws + rowflag_bytes(B)) + row * (long)nch * 5 * T;

  RowCfg cfg;
  cfg.z = z_dev ? make_float2(z_dev[0], z_dev[1]) : make_float2(z_re, z_im);
  cfg.N = N;
  cfg.raw = (flags & BKCORE_FLAG_RAW) != 0;
  cfg.fix_bc = fix_bc;
  cfg.fclamp = fclamp;

  const int t = threadIdx.x;
  ScanCtx ctx;
  ctx.nscan = 0;
  ctx.cpar[0] = 0;
  ctx.cpar[1] = 0;
  unsigned seen = 0;
  bool bad = false;

  // ---- sweep A: theta; each lane's incoming state per chunk is stored
  if (t == 0) {
    sh.carry[0][0] = vec_start();
    sh.carry[1][0] = vec_start();
  }
  sweepA_range<TIn, FAST>(a, b, c, cfg, sh, ctx, st, 0, nch, seen, bad);
  // theta_N = first component of the pass-2 state after the last element
  __syncthreads();
  const Vec cdet = sh.carry[1][ctx.cpar[1]];
  const float2 dm = cdet.x0;
  const int e_det = cdet.e;
  const float dn = fmaf(dm.x, dm.x, dm.y * dm.y);
  const float2 inv_det = make_float2(dm.x / dn, -dm.y / dn);
  __syncthreads();

  // ---- sweep B: phi with running carry, theta from the stored lane states, epilogue, stores
  if (t == 0) {
    sh.carry[0][ctx.cpar[0]] = vec_start();
    sh.carry[1][ctx.cpar[1]] = vec_start();
  }
  constexpr int KO_PHI = PHI_EXACT ? 1 : 0;
  sweepB_range<TIn, FAST, KO_PHI, SAVE>(a, b, c, cfg, sh, ctx, st, 0, nch, inv_det, e_det, G, F,
                                        SAVE ? save_r_ + row * N : nullptr, SAVE ? save_s_ + row * N : nullptr, bad);
  if (FAST) {
    // verdict of the fast path for this row; the second launch recomputes the rows that failed
    const int any_bad = __syncthreads_or(bad ? 1 : 0);
    if (t == 0) {
      rowflag[row] = any_bad;
      if (any_bad) __hip_atomic_store(&state[2], call_id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (state && !second && seen) {
    if (seen & 1u) __hip_atomic_store(&state[0], call_id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (seen & 2u) __hip_atomic_store(&state[1], call_id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// ---- long rows, few of them: several workgroups per row --------------------------------------------------------
// With B below the number of workgroups the chip holds, a row is cut into S segments of `seg_ch` chunks and each
// (row, segment) gets its own workgroup.  A segment cannot know the state entering it, but it does not need to:
//   * the recurrences' wall factors and all ratios of consecutive terms depend on the entering state only through
//     the entering RATIO, and wherever the ratio map contracts that ratio is fixed by the few hundred elements
//     before the segment.  It is estimated by running the recurrence serially over the WARM elements on the far
//     side of the boundary, starting from the start vector;
//   * started from (ratio, 1), the segment's terms are the true ones up to one complex factor per segment.
// bk_split_aggregate_kernel runs sweep A over each segment (storing lane states as the one-workgroup kernel does)
// and a phi scan without outputs, and records each segment's final states.  bk_split_combine_kernel walks the
// segments of a row: it checks every estimate against the true ratio coming out of the neighbouring segment,
// accumulates the per-segment factors of theta and phi, and from them the factor theta_N^-1 * C_s * D_s that
// bk_split_output_kernel (sweep B over each segment) applies in the epilogue.  A row with any failed check --
// these or the per-lane checks of the fast path -- is flagged and recomputed by the one-workgroup two-scan kernel.
constexpr int WARM = 128;          // elements of serial warm-up at a segment boundary
constexpr int SEG_INFO = 32;       // floats of per-segment record
// record layout (floats): [0..4] theta out, [5..9] phi out, [10..14] theta start, [15..19] phi start,
//                         [20,21] factor mantissa, [22] factor exponent (as -e), [23] failed check in this segment,
//                         [24,25,26] running theta factor (combine kernel scratch)

__device__ __forceinline__ void put_vec(float *p, const Vec &x) {
  p[0] = x.x0.x; p[1] = x.x0.y; p[2] = x.x1.x; p[3] = x.x1.y; p[4] = __int_as_float(x.e);
}
__device__ __forceinline__ Vec get_vec(const float *p) {
  Vec x;
  x.x0 = make_float2(p[0], p[1]);
  x.x1 = make_float2(p[2], p[3]);
  x.e = __float_as_int(p[4]);
  return x;
}

__device__ __forceinline__ float2 reg_alpha(float a, const RowCfg &cfg) {
  const float xr = a - cfg.z.x, xi = -cfg.z.y;
  const float n = fmaf(xr, xr, xi * xi);
  float f = fmaf(1e-3f, __builtin_amdgcn_rsqf(n), 1.0f);
  float add = 0.0f;
  if (!(n >= 1e-24f)) { f = 1.0f; add = 1e-3f; }
  return make_float2(fmaf(xr, f, add), xi * f);
}

// Boundary warm-up: the recurrence run serially over `count` <= WARM elements i_first, i_first + dir, ... from the
// start vector, wall factor included; the step at element i uses k_{i-1+ko}.  warm_stage: one thread loads and
// regularises element number n into the LDS arrays at `wa`; warm_loop: one wave runs the serial loop (two chains
// as in theta_lane: a linear one that supplies the wall factor off the critical path, and the corrected one) and
// returns the final pair of terms with exponent 0.  A NaN in b or c stays NaN (such a row fails its checks and is
// recomputed by the kernel that implements the sanitising rules).
template <typename TIn>
__device__ __forceinline__ void warm_stage(const TIn *__restrict__ a, const TIn *__restrict__ b,
                                           const TIn *__restrict__ c, const RowCfg &cfg, long i_first, int dir, int ko,
                                           float *wa, int n) {
  const long i = i_first + (long)dir * n;
  float av = to_f32<TIn>(a[i]);
  const long j = i - 1 + ko;
  float kk = 0.0f;
  if (j >= 0 && j <= cfg.N - 2) {
    float bv = to_f32<TIn>(b[j]), cv = to_f32<TIn>(c[j]);
    if (!cfg.raw) {
      bv = clamp_keep_nan(bv, 10.0f);
      cv = clamp_keep_nan(cv, 10.0f);
    }
    kk = bv * cv;
  }
  if (!cfg.raw) {
    av = clamp_keep_nan(av, 100.0f);
    if (av != av) av = 0.0f;
  }
  const float2 al = reg_alpha(av, cfg);
  wa[n] = al.x;
  wa[WARM + n] = al.y;
  wa[2 * WARM + n] = kk;
}

__device__ __forceinline__ Vec warm_loop(const float *wa, int count) {
  Vec x = vec_start();
  float2 u0 = x.x0, u1 = x.x1;
  float l0 = log2_norm2(u0);
  float l1 = log2_norm2(u1);
#pragma unroll 4
  for (int n = 0; n < count; ++n) {
    const float2 al = make_float2(wa[n], wa[WARM + n]);
    const float k = wa[2 * WARM + n];
    const float d = LN2_HALF * (l1 - l0);
    const float kk = k * wall_corr(d);
    const float2 yu = rstep(al, k, u0, u1);
    const float2 y = rstep(al, kk, x.x0, x.x1);
    u1 = u0;
    u0 = yu;
    x.x1 = x.x0;
    x.x0 = y;
    l1 = l0;
    if ((n & 1) == 1) {
      const int de = expo(fmaxf(amax2(x.x0), amax2(x.x1)));
      const float sc = pow2_neg(de);
      x.x0 = sc2(x.x0, sc);
      x.x1 = sc2(x.x1, sc);
      u0 = sc2(u0, sc);
      u1 = sc2(u1, sc);
      l1 -= 2.0f * (float)de;
    }
    l0 = log2_norm2(u0);
  }
  vnorm(x);
  x.e = 0;
  return x;
}

// |est ratio - true ratio| > 2e-5 |true ratio| (also true when anything is not a number)
__device__ __forceinline__ bool ratio_differs_tight(const Vec &est, const Vec &x) {
  const float2 c = cmul(est.x0, x.x1);
  const float2 d = cmul(x.x0, est.x1);
  const float dx = c.x - d.x, dy = c.y - d.y;
  const float lhs = fmaf(dx, dx, dy * dy);
  const float rhs = 4.0e-10f * fmaf(d.x, d.x, d.y * d.y);
  return !(lhs <= rhs);
}

__host__ __device__ inline size_t lane_state_bytes(long B, long N) {
  return (size_t)B * (size_t)((N + CHUNK - 1) / CHUNK) * 5 * T * sizeof(float);
}

template <typename TIn, bool PHI_EXACT>
__global__ __launch_bounds__(T, BK_WAVES_PER_SIMD) void bk_split_aggregate_kernel(
    const TIn *__restrict__ a_, long lda, const TIn *__restrict__ b_, long ldb, const TIn *__restrict__ c_, long ldc,
    const float *__restrict__ z_dev, float z_re, float z_im, long B, long N, char *__restrict__ ws, int *state,
    int call_id, unsigned flags, int S, int seg_ch) {
  __shared__ Shared sh;
  const long row = blockIdx.x / S;
  const int sg = blockIdx.x % S;
  const TIn *a = a_ + row * lda;
  const TIn *b = b_ + row * ldb;
  const TIn *c = c_ + row * ldc;
  const int nch = (int)((N + CHUNK - 1) / CHUNK);
  const int ch0 = sg * seg_ch;
  const int ch1 = min(nch, ch0 + seg_ch);
  float *st = reinterpret_cast<float *>(ws + rowflag_bytes(B)) + row * (long)nch * 5 * T;
  float *info = reinterpret_cast<float *>(ws + rowflag_bytes(B) + lane_state_bytes(B, N)) + (row * S + sg) * SEG_INFO;
  RowCfg cfg;
  cfg.z = z_dev ? make_float2(z_dev[0], z_dev[1]) : make_float2(z_re, z_im);
  cfg.N = N;
  cfg.raw = (flags & BKCORE_FLAG_RAW) != 0;
  cfg.fix_bc = false;
  cfg.fclamp = __builtin_huge_valf();
  const int t = threadIdx.x;
  ScanCtx ctx;
  ctx.nscan = 0;
  ctx.cpar[0] = 0;
  ctx.cpar[1] = 0;
  unsigned seen = 0;
  bool bad = false;
  constexpr int KO_PHI = PHI_EXACT ? 1 : 0;

  // both boundary warm-ups at once: threads 0..WARM-1 stage theta's elements and wave 0 runs its loop, threads
  // WARM..2 WARM-1 and wave 2 do the same for phi
  static_assert(2 * WARM <= T && 6 * WARM <= A_PAD && WARM % 64 == 0, "warm-up staging");
  const long p0 = (long)ch0 * CHUNK;
  const long p1 = (long)ch1 * CHUNK;  // first element right of the segment
  const int cnt_th = sg != 0 ? WARM : 0;
  const int cnt_ph = sg != S - 1 ? (int)min((long)WARM, N - p1) : 0;
  {
    float *wa = sh.u.in.a_s;
    if (t < WARM) {
      if (t < cnt_th) warm_stage<TIn>(a, b, c, cfg, p0 - WARM, +1, 0, wa, t);
    } else if (t < 2 * WARM) {
      if (t - WARM < cnt_ph) warm_stage<TIn>(a, b, c, cfg, p1 + cnt_ph - 1, -1, KO_PHI, wa + 3 * WARM, t - WARM);
    }
    __syncthreads();
    Vec w0 = vec_start();
    if (t < 64) {
      if (cnt_th) w0 = warm_loop(wa, cnt_th);
    } else if (t >= WARM && t < WARM + 64) {
      if (cnt_ph) w0 = warm_loop(wa + 3 * WARM, cnt_ph);
    }
    if (t == 0) {
      sh.carry[0][0] = w0;
      sh.carry[1][0] = w0;
      put_vec(info + 10, w0);
    }
    if (t == WARM) put_vec(info + 15, w0);
  }
  // theta over the segment
  sweepA_range<TIn, true>(a, b, c, cfg, sh, ctx, st, ch0, ch1, seen, bad);
  __syncthreads();
  if (t == 0) put_vec(info + 0, sh.carry[1][ctx.cpar[1]]);
  __syncthreads();

  // phi over the segment (no outputs yet: only the state leaving it on the left); its start was recorded by
  // thread WARM above
  if (t == WARM) {
    const Vec ph0 = get_vec(info + 15);
    sh.carry[0][ctx.cpar[0]] = ph0;
    sh.carry[1][ctx.cpar[1]] = ph0;
  }
  {
    auto one = [&](auto full, auto tail, int ch) __attribute__((always_inline)) {
      constexpr bool FULL = decltype(full)::value, TAIL = decltype(tail)::value;
      const long base = (long)ch * CHUNK;
      float2 al[E];
      float kw[E + 1];
      unsigned vmask;
      __syncthreads();
      stage_chunk<TIn, FULL>(a, b, c, base, cfg, sh);
      __syncthreads();
      fetch_elems(sh, base, T - 1 - t, cfg, al, kw, vmask);
      float k2[E];
      two_pass<true, true, TAIL, KO_PHI>(al, kw, vmask, k2, sh, ctx, bad);
    };
    int ch = ch1 - 1;
    if (ch >= ch0 && ch == nch - 1) {
      one(BoolC<false>{}, BoolC<true>{}, ch);
      --ch;
    }
    const int interior_begin = max(ch0, 1);
    for (; ch >= interior_begin; --ch) one(BoolC<true>{}, BoolC<false>{}, ch);
    if (ch >= ch0 && ch == 0) one(BoolC<false>{}, BoolC<false>{}, 0);
  }
  const int any_bad = __syncthreads_or(bad ? 1 : 0);
  if (t == 0) {
    put_vec(info + 5, sh.carry[1][ctx.cpar[1]]);
    info[23] = any_bad ? 1.0f : 0.0f;
  }
  if (state && seen) {
    if (seen & 1u) __hip_atomic_store(&state[0], call_id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (seen & 2u) __hip_atomic_store(&state[1], call_id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

struct Scal {  // complex mantissa * 2^e
  float2 m;
  int e;
};
__device__ __forceinline__ void snorm(Scal &v) {
  const int de = expo(amax2(v.m));
  v.m = sc2(v.m, pow2_neg(de));
  v.e += de;
}

__device__ __forceinline__ Scal smul(const Scal &p, const Scal &q) {
  Scal r;
  r.m = cmul(p.m, q.m);
  r.e = p.e + q.e;
  snorm(r);
  return r;
}
__device__ __forceinline__ Scal sone() {
  Scal r;
  r.m = make_float2(1.f, 0.f);
  r.e = 0;
  return r;
}
template <bool UP>
__device__ __forceinline__ Scal sshfl(const Scal &v, int d) {
  Scal r;
  r.m.x = UP ? __shfl_up(v.m.x, d) : __shfl_down(v.m.x, d);
  r.m.y = UP ? __shfl_up(v.m.y, d) : __shfl_down(v.m.y, d);
  r.e = UP ? __shfl_up(v.e, d) : __shfl_down(v.e, d);
  return r;
}

constexpr int MAX_SEGMENTS = 1024;

// One wave per row, one lane per segment (64 at a time): checks every boundary estimate against the neighbouring
// segment's outgoing state, forms the running factors of theta (C_s, left to right) and phi (D_s, right to left)
// as products of complex scalars with exponents (wave scans over shuffles), and writes each segment's epilogue
// factor C_s D_s / theta_N.
__global__ __launch_bounds__(64) void bk_split_combine_kernel(long B, long N, char *__restrict__ ws, int *state,
                                                              int call_id, int S) {
  __shared__ float cm[3][MAX_SEGMENTS];
  const long row = blockIdx.x;
  const int lane = threadIdx.x;
  int *rowflag = reinterpret_cast<int *>(ws);
  float *info = reinterpret_cast<float *>(ws + rowflag_bytes(B) + lane_state_bytes(B, N)) + row * S * SEG_INFO;
  bool bad = false;
  Scal carry = sone();
  for (int g0 = 0; g0 < S; g0 += 64) {
    const int sg = g0 + lane;
    Scal q = sone();
    if (sg < S) {
      const float *me = info + sg * SEG_INFO;
      bad |= me[23] != 0.0f;
      if (sg > 0) {
        const Vec out = get_vec(me - SEG_INFO + 0);  // theta leaving the segment on the left
        const Vec est = get_vec(me + 10);
        bad |= ratio_differs_tight(est, out);
        q.m = cdiv(out.x0, est.x0);  // this segment's terms times the running product are the true ones
        q.e = out.e;
        snorm(q);
      }
    }
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const Scal o = sshfl<true>(q, d);
      if (lane >= d) q = smul(q, o);
    }
    const Scal C = smul(carry, q);
    if (sg < S) {
      cm[0][sg] = C.m.x;
      cm[1][sg] = C.m.y;
      cm[2][sg] = __int_as_float(C.e);
    }
    carry.m.x = __shfl(C.m.x, 63);
    carry.m.y = __shfl(C.m.y, 63);
    carry.e = __shfl(C.e, 63);
  }
  __syncthreads();
  // theta_N = C_{S-1} * (theta leaving the last segment)
  const Vec last = get_vec(info + (S - 1) * SEG_INFO);
  Scal det;
  det.m = cmul(carry.m, last.x0);
  det.e = carry.e + last.e;
  snorm(det);
  carry = sone();
  for (int g0 = ((S - 1) / 64) * 64; g0 >= 0; g0 -= 64) {
    const int sg = g0 + lane;
    Scal q = sone();
    if (sg < S - 1) {
      const float *me = info + sg * SEG_INFO;
      const Vec out = get_vec(me + SEG_INFO + 5);  // phi leaving the segment on the right
      const Vec est = get_vec(me + 15);
      bad |= ratio_differs_tight(est, out);
      q.m = cdiv(out.x0, est.x0);
      q.e = out.e;
      snorm(q);
    }
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const Scal o = sshfl<false>(q, d);
      if (lane + d < 64) q = smul(q, o);
    }
    const Scal D = smul(carry, q);
    if (sg < S) {
      Scal f;
      f.m = cdiv(cmul(make_float2(cm[0][sg], cm[1][sg]), D.m), det.m);
      f.e = __float_as_int(cm[2][sg]) + D.e - det.e;
      snorm(f);
      float *me = info + sg * SEG_INFO;
      me[20] = f.m.x;
      me[21] = f.m.y;
      me[22] = __int_as_float(-f.e);
    }
    carry.m.x = __shfl(D.m.x, 0);
    carry.m.y = __shfl(D.m.y, 0);
    carry.e = __shfl(D.e, 0);
  }
  const int any_bad = __syncthreads_or(bad ? 1 : 0);
  if (lane == 0) {
    rowflag[row] = any_bad ? 1 : 0;
    if (any_bad) __hip_atomic_store(&state[2], call_id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

template <typename TIn, bool PHI_EXACT>
__global__ __launch_bounds__(T, BK_WAVES_PER_SIMD) void bk_split_output_kernel(
    const TIn *__restrict__ a_, long lda, const TIn *__restrict__ b_, long ldb, const TIn *__restrict__ c_, long ldc,
    const float *__restrict__ z_dev, float z_re, float z_im, long B, long N, float2 *__restrict__ G_,
    float2 *__restrict__ F_, char *__restrict__ ws, int *state, int call_id, unsigned flags, float fclamp, int S,
    int seg_ch) {
  __shared__ Shared sh;
  const long row = blockIdx.x / S;
  const int sg = blockIdx.x % S;
  int *rowflag = reinterpret_cast<int *>(ws);
  if (rowflag[row] != 0) return;  // the whole row is recomputed by the one-workgroup kernel
  const TIn *a = a_ + row * lda;
  const TIn *b = b_ + row * ldb;
  const TIn *c = c_ + row * ldc;
  float2 *G = G_ + row * N;
  float2 *F = F_ ? F_ + row * N : nullptr;
  const int nch = (int)((N + CHUNK - 1) / CHUNK);
  const int ch0 = sg * seg_ch;
  const int ch1 = min(nch, ch0 + seg_ch);
  const float *st = reinterpret_cast<const float *>(ws + rowflag_bytes(B)) + row * (long)nch * 5 * T;
  const float *info =
      reinterpret_cast<const float *>(ws + rowflag_bytes(B) + lane_state_bytes(B, N)) + (row * S + sg) * SEG_INFO;
  RowCfg cfg;
  cfg.z = z_dev ? make_float2(z_dev[0], z_dev[1]) : make_float2(z_re, z_im);
  cfg.N = N;
  cfg.raw = (flags & BKCORE_FLAG_RAW) != 0;
  cfg.fix_bc = false;
  cfg.fclamp = fclamp;
  const int t = threadIdx.x;
  ScanCtx ctx;
  ctx.nscan = 0;
  ctx.cpar[0] = 0;
  ctx.cpar[1] = 0;
  bool bad = false;
  // uniform values, kept in scalar registers
  const float2 factor = make_float2(__int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(info[20]))),
                                    __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(info[21]))));
  const int e_factor = __builtin_amdgcn_readfirstlane(__float_as_int(info[22]));
  if (t == 0) {
    const Vec ph0 = get_vec(info + 15);
    sh.carry[0][0] = ph0;
    sh.carry[1][0] = ph0;
  }
  constexpr int KO_PHI = PHI_EXACT ? 1 : 0;
  sweepB_range<TIn, true, KO_PHI, false>(a, b, c, cfg, sh, ctx, st, ch0, ch1, factor, e_factor, G, F, nullptr, nullptr,
                                         bad);
  const int any_bad = __syncthreads_or(bad ? 1 : 0);
  if (t == 0 && any_bad) {
    atomicOr(&rowflag[row], 1);
    __hip_atomic_store(&state[2], call_id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// ---- single-sweep forward in ratio form ------------------------------------------------------------------------
// The same numbers from a different evaluation order.  With r_i = theta_{i+1}/theta_i and s_m = phi_m/phi_{m+1} the
// reference's recurrences (bk_core.py:120-184) read
//     r_i = a'_i - k_{i-1} om(r_{i-1}),   s_m = a'_m - kphi_m om(s_{m+1}),   om(r) = exp(g(-ln|r|)) / r
// (g(d) = 88 tanh(d/88) - d is the in-recurrence soft clamp relative to the exact recurrence), and
//     G_raw,j = theta_j phi_{j+1} / theta_N = (1/r_j) prod_{m>j} s_m / r_m .
// Wherever these first-order maps contract (|k| < |r|^2: every well-conditioned row) a ratio forgets where it started
// within a few steps, so a lane does not need a scan to know the ratio entering its 16 elements: it runs the map over
// the 16 elements before them (after them, for s) from the start value.  The estimate is verified: the lane that owns
// those 16 elements ran the same steps from its own, different, start; both ends must agree to 2e-6.  A row with any
// disagreement is flagged and recomputed by the scan kernel in the second launch, exactly like the scan fast path.
// What still needs a scan is the product over m > j -- of complex scalars, carried as mantissa x 2^e, one scan per
// chunk -- and the only state carried from chunk to chunk is that running product, so the row is read ONCE
// (chunks right -> left): 12 bytes in, 16 out per element, no lane states.
#ifndef BK_RATIO_WAVES
#define BK_RATIO_WAVES 2
#endif
constexpr int RH = 16;                          // warm-up length = halo staged on both sides of a chunk
#ifndef BK_RH_COARSE
#define BK_RH_COARSE 8
#endif
constexpr int RH_COARSE = BK_RH_COARSE;         // leading warm-up steps taken in the coarse form
constexpr int RN = CHUNK + 2 * RH;              // staged elements per chunk
constexpr int RAL_PAD = RN + RN / E + 8;        // float2 entries, index q + q/E
constexpr int RK_PAD = RN + 1 + (RN + 1) / E + 8;

struct Prod {  // complex mantissa * 2^e
  float2 m;
  int e;
};
__device__ __forceinline__ Prod prod_one() {
  Prod p;
  p.m = make_float2(1.f, 0.f);
  p.e = 0;
  return p;
}
__device__ __forceinline__ void pnorm(Prod &p) {
  const int de = expo(amax2(p.m));
  p.m = sc2(p.m, pow2_neg(de));
  p.e += de;
}
__device__ __forceinline__ Prod pmul(const Prod &a, const Prod &b) {
  Prod r;
  r.m = cmul(a.m, b.m);
  r.e = a.e + b.e;
  pnorm(r);
  return r;
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ Prod prod_dpp(const Prod &p) {  // identity where the selection has no source lane
  Prod r;
  r.m.x = dpp_f<CTRL, ROW_MASK>(1.0f, p.m.x);
  r.m.y = dpp_f<CTRL, ROW_MASK>(0.0f, p.m.y);
  r.e = dpp_i<CTRL, ROW_MASK>(0, p.e);
  return r;
}

struct SharedR {
  union {
    struct {
      float2 al_s[RAL_PAD];  // regularised diagonal a' of elements base - RH .. base + CHUNK + RH - 1
      float k_s[RK_PAD];     // k_{i-1} for the same elements, plus one more
    } in;
    float2 out_s[O_PAD];     // G of one chunk for coalesced stores
  } u;
  float4 wedge[2][WAVES];    // last lane of each wave: theta warm-up result (xy) and phi end value (zw), for the
                             // first lane of the next wave
  float4 edge[2];            // the same from the first lane of the chunk to the right
  Prod tot[2][WAVES];
  Prod carry[2];
};

__device__ __forceinline__ int rpad16(int q) { return q + q / E; }  // lane stride E + 1 entries
// staged slot q0 + d of a lane whose first slot q0 is a multiple of 16: base + roff(d), d a compile-time constant
__device__ __forceinline__ constexpr int roff(int d) { return d + (d >= 0 ? d / E : -((E - 1 - d) / E)); }

// A lane's 16 consecutive complex values as eight 16-byte stores (the struct is only 4-byte aligned: rows start
// anywhere); `count` < 16 only at the ragged end of a row.
struct __attribute__((aligned(4))) F4u {
  float v[4];
};
__device__ __forceinline__ void store_lane_block(float2 *__restrict__ dst, const float2 (&v)[E], int count) {
  float *p = reinterpret_cast<float *>(dst);
  if (count >= E) {
#pragma unroll
    for (int q = 0; q < E / 2; ++q) {
      F4u t;
      t.v[0] = v[2 * q].x;
      t.v[1] = v[2 * q].y;
      t.v[2] = v[2 * q + 1].x;
      t.v[3] = v[2 * q + 1].y;
      *reinterpret_cast<F4u *>(p + 4 * q) = t;
    }
  } else {
#pragma unroll
    for (int e = 0; e < E; ++e)
      if (e < count) dst[e] = v[e];
  }
}

// One step of either ratio map: r = al - k * om_prev; returns om(r) and hands back r and 1/r.  The reciprocal of
// |r|^2 is refined by one Newton step and the clamp factor exp(g) is an exp2 of a tiny argument: the hardware log
// and exp2 round with a bias of a fraction of an ulp, which is harmless inside the contracting recurrence and in g,
// but would add up linearly over the N factors of the product if 1/|r|^2 were formed as exp2(-log2 |r|^2).
__device__ __forceinline__ float2 ratio_step(float2 al, float k, float2 om, float2 &r, float2 &inv) {
  r = make_float2(fmaf(-k, om.x, al.x), fmaf(-k, om.y, al.y));
  const float n = fmaf(r.x, r.x, r.y * r.y);
  float y = __builtin_amdgcn_rcpf(n);
  y = fmaf(y, fmaf(-n, y, 1.0f), y);
  inv = make_float2(r.x * y, -r.y * y);
  // clamp factor exp(g(d)), d = -ln|r| = -(ln 2 / 2) log2 |r|^2, |d| limited to 22.  g = -88 (u^3/3 - 2u^5/15),
  // u = d/88 (|u| <= 1/4: relative error of g below 7e-4, and below 1e-6 for |r| between 1e-2 and 1e2); exp(g) by
  // its series to g^2 (|g| <= 0.02 for any |r| between 1e-3 and 1e9: 1.4e-6 there, 1e-14 at |r| ~ 2).  The clamp is
  // a 1e-5 effect on k, so 1e-3 of it is far below float32 resolution.
  const float u = __builtin_amdgcn_fmed3f(__builtin_amdgcn_logf(n) * (-LN2_HALF / 88.0f), -0.25f, 0.25f);
  const float s2 = u * u;
  const float p = fmaf(s2, 88.0f * 2.0f / 15.0f, -88.0f / 3.0f);
  const float g = (u * s2) * p;
  const float wc = fmaf(g, fmaf(g, 0.5f, 1.0f), 1.0f);
  return make_float2(inv.x * wc, inv.y * wc);
}

// The same step without the Newton refinement and without the clamp factor (a relative 1e-4 at most for |r| within
// e^+-2 of 1): for the first warm-up steps, whose errors the remaining ones contract away.
__device__ __forceinline__ float2 ratio_step_coarse(float2 al, float k, float2 om) {
  const float2 r = make_float2(fmaf(-k, om.x, al.x), fmaf(-k, om.y, al.y));
  const float y = __builtin_amdgcn_rcpf(fmaf(r.x, r.x, r.y * r.y));
  return make_float2(r.x * y, -r.y * y);
}

// |p - q| > 2e-6 |q| (also true when anything is not a number; false when both are zero)
__device__ __forceinline__ bool differs_tight(float2 p, float2 q) {
  const float dx = p.x - q.x, dy = p.y - q.y;
  return !(fmaf(dx, dx, dy * dy) <= 4.0e-12f * fmaf(q.x, q.x, q.y * q.y));
}

// Staging of a' and k for elements base - RH .. base + CHUNK + RH - 1 (k: one more), in two halves so that the
// global loads of the next chunk can be in flight while the current one is computed:
//   load_ratio   global -> registers (17 rounds of 256 slots cover the RN + 1 = 4129 slots).  Outside the row a = 1,
//                b = c = 0.  `full` (uniform): every index lies inside the row, no bounds logic.
//   commit_ratio registers -> LDS: the operator's clamp / NaN rules, regularisation, k = b c.  Returns the NaN bits as
//                stage_chunk does.
constexpr int RNJ = E + 1;
template <typename TIn>
__device__ __forceinline__ void load_ratio(const TIn *__restrict__ a, const TIn *__restrict__ b,
                                           const TIn *__restrict__ c, long base, long N, bool full, float (&av)[RNJ],
                                           float (&bv)[RNJ], float (&cv)[RNJ]) {
  const int t = threadIdx.x;
  // 32-bit offsets from uniform bases (scalar base + vector offset addressing); slot q holds element base - RH + q
  const TIn *__restrict__ a0 = a + (base - RH);
  const TIn *__restrict__ b0 = b + (base - RH - 1);
  const TIn *__restrict__ c0 = c + (base - RH - 1);
  if (full) {
#pragma unroll
    for (int j = 0; j < RNJ; ++j) {
      const int q = min(t + j * T, RN);
      av[j] = to_f32<TIn>(a0[q]);
      bv[j] = to_f32<TIn>(b0[q]);
      cv[j] = to_f32<TIn>(c0[q]);
    }
  } else {
    // branch-free: every load uses a clamped, always valid index (b and c must be readable at index 0 even when the
    // row has no off-diagonal: the caller passes `a` for them then); out-of-row values are replaced afterwards
    const int lo_a = (int)max(0L, RH - base);                  // first slot whose element index is >= 0
    const int hi_a = (int)min((long)RN, N - 1 - (base - RH));  // last slot whose element index is <= N-1 (may be < lo)
    const int lo_k = (int)max(0L, RH + 1 - base);              // k index = element index - 1 >= 0
    const int hi_k = (int)min((long)RN, N - 2 - (base - RH - 1));
    const int ca = max(hi_a, lo_a), ck = max(hi_k, lo_k);
#pragma unroll
    for (int j = 0; j < RNJ; ++j) {
      const int q = min(t + j * T, RN);
      const int qa = min(max(q, lo_a), ca);
      const int qk = min(max(q, lo_k), ck);
      const float va = to_f32<TIn>(a0[qa]);
      const float vb = to_f32<TIn>(b0[N >= 2 ? qk : qa + 1]);
      const float vc = to_f32<TIn>(c0[N >= 2 ? qk : qa + 1]);
      av[j] = (q >= lo_a && q <= hi_a) ? va : 1.0f;
      bv[j] = (q >= lo_k && q <= hi_k) ? vb : 0.0f;
      cv[j] = (q >= lo_k && q <= hi_k) ? vc : 0.0f;
    }
  }
}

__device__ __forceinline__ unsigned commit_ratio(const RowCfg &cfg, float (&av)[RNJ], float (&bv)[RNJ],
                                                 float (&cv)[RNJ], SharedR &sh) {
  unsigned seen = 0;
  const int t = threadIdx.x;
  if (!cfg.raw) {
    float probe = 0.0f;
#pragma unroll
    for (int j = 0; j < RNJ; ++j) probe += av[j] + (bv[j] + cv[j]);
    if (__builtin_expect(!__any(probe != probe), 1)) {
#pragma unroll
      for (int j = 0; j < RNJ; ++j) {
        av[j] = __builtin_amdgcn_fmed3f(av[j], -100.0f, 100.0f);
        bv[j] = __builtin_amdgcn_fmed3f(bv[j], -10.0f, 10.0f);
        cv[j] = __builtin_amdgcn_fmed3f(cv[j], -10.0f, 10.0f);
      }
    } else {
#pragma unroll
      for (int j = 0; j < RNJ; ++j) {
        float v = clamp_keep_nan(av[j], 100.0f);
        if (v != v) { v = 0.0f; seen |= 1u; }
        av[j] = v;
        float p = clamp_keep_nan(bv[j], 10.0f), qv = clamp_keep_nan(cv[j], 10.0f);
        if (p != p) { seen |= 2u; if (cfg.fix_bc) p = 1.0f; }
        if (qv != qv) { seen |= 2u; if (cfg.fix_bc) qv = 1.0f; }
        bv[j] = p;
        cv[j] = qv;
      }
    }
  }
#pragma unroll
  for (int j = 0; j < RNJ; ++j) {
    const int q = t + j * T;
    if (j < E || q <= RN) {
      sh.u.in.k_s[rpad16(q)] = bv[j] * cv[j];
      if (j < E || q < RN) sh.u.in.al_s[rpad16(q)] = reg_alpha(av[j], cfg);
    }
  }
  return seen;
}

// MODE 0: one workgroup per row.  Few long rows (split_plan) are cut into S segments of seg_ch chunks, one workgroup
// each, in two launches around bk_ratio_combine_kernel: the chunks of a row are independent except for the running
// product, so MODE 1 computes only each segment's product (and its checks; no outputs), the combine kernel turns
// those into the product entering each segment and checks the estimates across segment boundaries, and MODE 2 is
// the full computation of each segment started from that product.
// Per-segment record (floats): [0,1] product mantissa, [2] exponent, [3] a check failed, [4..7] right edge
// (theta end value, phi warm-up of the last lane), [8..11] left edge (theta warm-up, phi end value of the first lane),
// [12,13] mantissa and [14] exponent of the product entering the segment.
// SAVE (MODE 0 only, for the analytic backward): also stores r_j and s_j, and writes G_raw (before the final
// log-magnitude clamp and the clip; 0 outside 1e-18 <= |G_raw| <= 1e18) instead of G.
template <typename TIn, bool PHI_EXACT, int MODE, bool SAVE = false>
__global__ __launch_bounds__(T, BK_RATIO_WAVES) void bk_ratio_forward_kernel(
    const TIn *__restrict__ a_, long lda, const TIn *__restrict__ b_, long ldb, const TIn *__restrict__ c_, long ldc,
    const float *__restrict__ z_dev, float z_re, float z_im, long B, long N, float2 *__restrict__ G_,
    float2 *__restrict__ F_, char *__restrict__ ws, int *state, int call_id, unsigned flags, float fclamp, int S,
    int seg_ch, float2 *__restrict__ save_r_ = nullptr, float2 *__restrict__ save_s_ = nullptr) {
  __shared__ SharedR sh;
  const long row = MODE == 0 ? (long)blockIdx.x : (long)(blockIdx.x / S);
  const int sg = MODE == 0 ? 0 : (int)(blockIdx.x % S);
  int *rowflag = reinterpret_cast<int *>(ws);
  if (MODE == 2 && rowflag[row] != 0) return;  // the whole row is recomputed by the scan kernel
  const TIn *a = a_ + row * lda;
  const TIn *b = N >= 2 ? b_ + row * ldb : a;  // never used for data when N == 1, but must be readable
  const TIn *c = N >= 2 ? c_ + row * ldc : a;
  float2 *G = G_ + row * N;
  float2 *F = F_ ? F_ + row * N : nullptr;
  const int nch = (int)((N + CHUNK - 1) / CHUNK);
  const int ch0 = MODE == 0 ? 0 : sg * seg_ch;
  const int ch1 = MODE == 0 ? nch : min(nch, ch0 + seg_ch);
  float *info = reinterpret_cast<float *>(ws + rowflag_bytes(B) + lane_state_bytes(B, N)) + (row * S + sg) * SEG_INFO;
  RowCfg cfg;
  cfg.z = z_dev ? make_float2(z_dev[0], z_dev[1]) : make_float2(z_re, z_im);
  cfg.N = N;
  cfg.raw = (flags & BKCORE_FLAG_RAW) != 0;
  cfg.fix_bc = false;
  cfg.fclamp = fclamp;
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = t >> 6;
  const int seg = T - 1 - t;  // descending segments over ascending threads: the product runs right -> left
  constexpr int KO = PHI_EXACT ? 1 : 0;
  unsigned seen = 0;
  bool bad = false;
  int nscan = 0;
  if (t == 0) {
    Prod p0 = prod_one();
    if (MODE == 2) {
      p0.m = make_float2(info[12], info[13]);
      p0.e = __float_as_int(info[14]);
    }
    sh.carry[0] = p0;
  }

  // raw inputs of the chunk about to be processed, loaded one chunk ahead
  float av[RNJ], bv[RNJ], cv[RNJ];
  auto prefetch = [&](int ch) __attribute__((always_inline)) {
    // chunks 1 .. nch-3 lie at least a halo away from both ends of the row
    load_ratio<TIn>(a, b, c, (long)ch * CHUNK, N, ch >= 1 && ch <= nch - 3, av, bv, cv);
  };
  auto one = [&](auto tail, int ch) __attribute__((always_inline)) {
    constexpr bool TAIL = decltype(tail)::value;
    const long base = (long)ch * CHUNK;

    __syncthreads();  // previous chunk's output tile has been stored
    seen |= commit_ratio(cfg, av, bv, cv, sh);

    __syncthreads();
    if (ch > ch0) prefetch(ch - 1);  // in flight during this chunk's arithmetic

    const int q0 = seg * E + RH;      // staged slot of the lane's first element (a multiple of 16)
    const float2 *al_l = sh.u.in.al_s + rpad16(q0);
    const float *k_l = sh.u.in.k_s + rpad16(q0);
    const long i0 = base + (long)seg * E;
    float2 ir[E];                     // 1 / r_e
    float2 sv[E];                     // s_e; later the lane-local part of G_raw
    // theta runs left -> right, phi right -> left; the two chains are independent and advance in the same loop
    // iteration so that each wave always has two dependency chains in flight.  Warm-up: theta over the 16 elements
    // to the left of the lane's own, phi over the 16 to the right.  Past the end of the row om_phi = 0
    // (phi_{N+1} = 0: the recurrence starts with s_{N-1} = a'_{N-1}).
    float2 wt = make_float2(0.f, 0.f), wp = make_float2(0.f, 0.f), r, inv;
#pragma unroll
    for (int j = 0; j < RH; ++j) {
      const int d = E + RH - 1 - j;
      if (j < RH_COARSE) {
        wt = ratio_step_coarse(al_l[roff(j - RH)], k_l[roff(j - RH)], wt);
        wp = ratio_step_coarse(al_l[roff(d)], k_l[roff(d + KO)], wp);
      } else {
        wt = ratio_step(al_l[roff(j - RH)], k_l[roff(j - RH)], wt, r, inv);
        wp = ratio_step(al_l[roff(d)], k_l[roff(d + KO)], wp, r, inv);
      }
      if (TAIL && i0 + d >= N) wp = make_float2(0.f, 0.f);
    }
    const float2 th_west = wt, ph_west = wp;
#pragma unroll
    for (int j = 0; j < E; ++j) {
      wt = ratio_step(al_l[roff(j)], k_l[roff(j)], wt, r, inv);
      ir[j] = inv;
      const int e = E - 1 - j;
      wp = ratio_step(al_l[roff(e)], k_l[roff(e + KO)], wp, r, inv);
      if (TAIL && i0 + e >= N) wp = make_float2(0.f, 0.f);
      sv[e] = r;
    }
    const float2 th_fin = wt, ph_fin = wp;
    if (SAVE) {
      // r and s leave through the output tile (it overlays the staging area: wait until every wave's chains are done)
      const int n_sv = TAIL ? (int)max(0L, min((long)CHUNK, N - base)) : CHUNK;
      float2 *slot_sv = &sh.u.out_s[pad16(seg * E)];
#pragma unroll 1
      for (int pass = 0; pass < 2; ++pass) {
        __syncthreads();
        if (pass == 0) {
#pragma unroll
          for (int e = 0; e < E; ++e) {  // r = 1 / (1/r)
            const float y = 1.0f / fmaf(ir[e].x, ir[e].x, ir[e].y * ir[e].y);
            slot_sv[e] = make_float2(ir[e].x * y, -ir[e].y * y);
          }
        } else {
#pragma unroll
          for (int e = 0; e < E; ++e) slot_sv[e] = sv[e];
        }
        __syncthreads();
        float2 *__restrict__ dst = (pass == 0 ? save_r_ : save_s_) + row * N + base;
#pragma unroll
        for (int j = 0; j < E; ++j) {
          const int idx = t + j * T;
          if (!TAIL || idx < n_sv) dst[idx] = sh.u.out_s[pad16(idx)];
        }
      }
      __syncthreads();
    }

    // lane-local products, descending: loc_e = (1/r_e) * prod_{e' > e} s_e' / r_e', exponents shared by groups of 4
    Prod run = prod_one();
    int le[E / 4];
#pragma unroll
    for (int j = 0; j < E; ++j) {
      const int e = E - 1 - j;
      const bool valid = !TAIL || i0 + e < N;
      const float2 q1 = cmul(sv[e], ir[e]);  // s_e / r_e
      sv[e] = cmul(ir[e], run.m);
      if (valid) run.m = cmul(run.m, q1);
      if ((e & 3) == 0) {  // end of a group of 4: record the exponent it was processed under, then renormalise
        le[e >> 2] = run.e;
        pnorm(run);
      }
    }

    // workgroup exclusive product over ascending threads
    Prod A = run;
    A = pmul(A, prod_dpp<DPP_ROW_SHR1, 0xF>(A));
    A = pmul(A, prod_dpp<DPP_ROW_SHR2, 0xF>(A));
    A = pmul(A, prod_dpp<DPP_ROW_SHR4, 0xF>(A));
    A = pmul(A, prod_dpp<DPP_ROW_SHR8, 0xF>(A));
    A = pmul(A, prod_dpp<DPP_ROW_BCAST15, 0xA>(A));
    A = pmul(A, prod_dpp<DPP_ROW_BCAST31, 0xC>(A));
    const int buf = nscan & 1;
    nscan++;
    if (lane == 63) {
      sh.tot[buf][wave] = A;
      sh.wedge[buf][wave] = make_float4(th_west.x, th_west.y, ph_fin.x, ph_fin.y);
    }
    const Prod ex = prod_dpp<DPP_WAVE_SHR1, 0xF>(A);
    const float4 nbv = make_float4(dpp_f<DPP_WAVE_SHR1, 0xF>(0.f, th_west.x), dpp_f<DPP_WAVE_SHR1, 0xF>(0.f, th_west.y),
                                   dpp_f<DPP_WAVE_SHR1, 0xF>(0.f, ph_fin.x), dpp_f<DPP_WAVE_SHR1, 0xF>(0.f, ph_fin.y));
    __syncthreads();
    Prod v = sh.carry[buf];
    Prod fin_in = v;
#pragma unroll
    for (int wv = 0; wv < WAVES; ++wv) {
      if (wv == wave) fin_in = v;
      v = pmul(v, sh.tot[buf][wv]);
    }
    if (t == 0) sh.carry[buf ^ 1] = v;
    const Prod Fin = pmul(fin_in, ex);
    // checks of the warm-up estimates against the neighbour that owns those elements (thread t - 1 owns the
    // segment to the right; for thread 0 it is the first lane of the chunk processed before this one): the
    // neighbour's theta warm-up ran over this lane's elements, this lane's phi warm-up over the neighbour's
    {
      float4 nb = nbv;  // thread t-1's (theta warm-up, phi end) inside the wave
      if (lane == 0) nb = wave > 0 ? sh.wedge[buf][wave - 1] : sh.edge[buf ^ 1];
      if (t > 0 || ch < ch1 - 1) {
        bad |= differs_tight(make_float2(nb.x, nb.y), th_fin);
        bad |= differs_tight(ph_west, make_float2(nb.z, nb.w));
      }
      if (t == T - 1) sh.edge[buf] = make_float4(th_west.x, th_west.y, ph_fin.x, ph_fin.y);
      if (MODE == 1) {  // the two ends of the segment are checked against its neighbours by the combine kernel
        if (t == 0 && ch == ch1 - 1) {
          info[4] = th_fin.x; info[5] = th_fin.y; info[6] = ph_west.x; info[7] = ph_west.y;
        }
        if (t == T - 1 && ch == ch0) {
          info[8] = th_west.x; info[9] = th_west.y; info[10] = ph_fin.x; info[11] = ph_fin.y;
        }
      }
    }
    if (MODE == 1) return;  // only the product was wanted

    __syncthreads();  // staged inputs are dead: the tile may overwrite them

    // epilogue (bk_core.py:190-208) into the output tile
    float2 *slot = &sh.u.out_s[pad16(seg * E)];
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const float2 P = cmul(sv[e], Fin.m);
      const int Ee = Fin.e + le[e >> 2];
      const float n = fmaf(P.x, P.x, P.y * P.y);
      const float L = LN2_HALF * fmaf(2.0f, (float)Ee, __builtin_amdgcn_logf(n));  // ln |G_raw|
      float2 gv;
      if (SAVE) {
        gv = (fabsf(L) <= 41.4f) ? make_float2(ldexpf(P.x, Ee), ldexpf(P.y, Ee)) : make_float2(0.f, 0.f);
      } else {
        const float sc = ldexpf(__builtin_amdgcn_exp2f(wall_exponent(L) * LOG2E), Ee);
        gv = make_float2(P.x * sc, P.y * sc);
        const float m2 = fmaf(gv.x, gv.x, gv.y * gv.y);
        if (__builtin_expect(!(m2 <= 2500.0f), 0)) {
          const float u = 50.0f * __builtin_amdgcn_rsqf(n);
          gv = (L > 3.9f && L <= 3.0e38f && u <= 3.0e38f) ? make_float2(P.x * u, P.y * u) : make_float2(0.f, 0.f);
        }
      }
      slot[e] = gv;
    }
    __syncthreads();
    float2 *__restrict__ Gc = G + base;
    float2 *__restrict__ Fc = F ? F + base : nullptr;
    const int n_here = TAIL ? (int)max(0L, min((long)CHUNK, N - base)) : CHUNK;
#pragma unroll
    for (int j = 0; j < E; ++j) {
      const int idx = t + j * T;
      if (!TAIL || idx < n_here) {
        const float2 gv = SAVE ? sh.u.out_s[pad16(idx)] : fix_tiny(sh.u.out_s[pad16(idx)]);
        __builtin_nontemporal_store(gv.x, &Gc[idx].x);
        __builtin_nontemporal_store(gv.y, &Gc[idx].y);
        if (Fc) {
          __builtin_nontemporal_store(__builtin_amdgcn_fmed3f(gv.x, -cfg.fclamp, cfg.fclamp), &Fc[idx].x);
          __builtin_nontemporal_store(__builtin_amdgcn_fmed3f(gv.y, -cfg.fclamp, cfg.fclamp), &Fc[idx].y);
        }
      }
    }
  };
  // chunk kinds: the last two (anything at or past the end of the row within reach of a halo), then the others
  prefetch(ch1 - 1);
  int ch = ch1 - 1;
  for (; ch >= ch0 && ch >= nch - 2; --ch) one(BoolC<true>{}, ch);
  for (; ch >= ch0; --ch) one(BoolC<false>{}, ch);
  const int any_bad = __syncthreads_or(bad ? 1 : 0);
  if (t == 0) {
    if (MODE == 0) {
      rowflag[row] = any_bad;
      if (any_bad) __hip_atomic_store(&state[2], call_id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if (MODE == 1) {
      const Prod tot = sh.carry[nscan & 1];  // started from one: the segment's own product
      info[0] = tot.m.x;
      info[1] = tot.m.y;
      info[2] = __int_as_float(tot.e);
      info[3] = any_bad ? 1.0f : 0.0f;
    } else if (any_bad) {
      atomicOr(&rowflag[row], 1);
      __hip_atomic_store(&state[2], call_id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (seen) {
    if (seen & 1u) __hip_atomic_store(&state[0], call_id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (seen & 2u) __hip_atomic_store(&state[1], call_id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// One wave per row, one lane per segment (64 at a time, right to left): the product entering each segment, the
// checks across segment boundaries, the row's verdict.
__global__ __launch_bounds__(64) void bk_ratio_combine_kernel(long B, long N, char *__restrict__ ws, int *state,
                                                              int call_id, int S) {
  const long row = blockIdx.x;
  const int lane = threadIdx.x;
  int *rowflag = reinterpret_cast<int *>(ws);
  float *info = reinterpret_cast<float *>(ws + rowflag_bytes(B) + lane_state_bytes(B, N)) + row * S * SEG_INFO;
  bool bad = false;
  Prod carry = prod_one();
  for (int g0 = 0; g0 < S; g0 += 64) {
    const int idx = g0 + lane;
    const int sg = S - 1 - idx;  // ascending lanes = segments right to left
    Prod q = prod_one();
    if (idx < S) {
      const float *me = info + sg * SEG_INFO;
      q.m = make_float2(me[0], me[1]);
      q.e = __float_as_int(me[2]);
      bad |= me[3] != 0.0f;
      if (sg < S - 1) {
        const float *rt = me + SEG_INFO;  // the segment to the right
        bad |= differs_tight(make_float2(rt[8], rt[9]), make_float2(me[4], me[5]));
        bad |= differs_tight(make_float2(me[6], me[7]), make_float2(rt[10], rt[11]));
      }
    }
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      Prod o;
      o.m.x = __shfl_up(q.m.x, d);
      o.m.y = __shfl_up(q.m.y, d);
      o.e = __shfl_up(q.e, d);
      if (lane >= d) q = pmul(q, o);
    }
    Prod ex;
    ex.m.x = __shfl_up(q.m.x, 1);
    ex.m.y = __shfl_up(q.m.y, 1);
    ex.e = __shfl_up(q.e, 1);
    if (lane == 0) ex = prod_one();
    ex = pmul(carry, ex);
    if (idx < S) {
      float *me = info + sg * SEG_INFO;
      me[12] = ex.m.x;
      me[13] = ex.m.y;
      me[14] = __int_as_float(ex.e);
    }
    Prod last;
    last.m.x = __shfl(q.m.x, 63);
    last.m.y = __shfl(q.m.y, 63);
    last.e = __shfl(q.e, 63);
    carry = pmul(carry, last);
  }
  const int any_bad = __syncthreads_or(bad ? 1 : 0);
  if (lane == 0) {
    rowflag[row] = any_bad ? 1 : 0;
    if (any_bad) __hip_atomic_store(&state[2], call_id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// ---- reference-mode backward (elementwise), bk_core.py:303-385 ------------------------------------------------
__device__ __forceinline__ float nan_to_num_default(float v) {  // torch.nan_to_num defaults
  if (v != v) return 0.0f;
  if (v > 3.4028234663852886e38f) return 3.4028234663852886e38f;
  if (v < -3.4028234663852886e38f) return -3.4028234663852886e38f;
  return v;
}

__device__ __forceinline__ float bwd_ref_one(float2 G, float2 g, float alpha) {
  // g holds the cotangent of (Re G, Im G); conj(g) = (g.x, -g.y)
  const float2 g2 = cmul(G, G);
  const float mag = sqrtf(fmaf(g2.x, g2.x, g2.y * g2.y));
  float2 den = g2;
  if (mag < 1e-4f) {
    const float s = 1e-4f / (mag + 1e-9f);
    den = make_float2(g2.x * s, g2.y * s);
  }
  den.x += 1e-6f;
  // -Re(conj(g) * G^2)
  const float theo = -fmaf(g.x, g2.x, g.y * g2.y);
  // -Re(conj(g) / den) = -Re(conj(g) conj(den)) / |den|^2 = -(g.x den.x - g.y den.y) / |den|^2
  const float dn = fmaf(den.x, den.x, den.y * den.y);
  const float h7 = -fmaf(g.x, den.x, -g.y * den.y) / dn;
  float gv = fmaf(1.0f - alpha, theo, alpha * h7);
  if (gv != gv) gv = 0.0f;
  gv = fminf(fmaxf(gv, -100.0f), 100.0f);  // also maps +-inf to +-100
  return gv;
}

template <bool HAS_GF, bool HAS_GG>
__global__ __launch_bounds__(256) void bk_backward_ref_kernel(const float2 *__restrict__ G, const float2 *__restrict__ gf,
                                                              const float2 *__restrict__ gG, float alpha, long n,
                                                              float *__restrict__ out) {
  const long stride = (long)gridDim.x * blockDim.x;
  // two elements per thread per iteration: 16-byte loads of G / grads, 8-byte store
  const long npairs = n >> 1;
  for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < npairs; p += stride) {
    const float4 Gv = reinterpret_cast<const float4 *>(G)[p];
    float4 gv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (HAS_GF) {
      gv = reinterpret_cast<const float4 *>(gf)[p];
      gv.x = nan_to_num_default(gv.x);
      gv.y = nan_to_num_default(gv.y);
      gv.z = nan_to_num_default(gv.z);
      gv.w = nan_to_num_default(gv.w);
    }
    if (HAS_GG) {
      const float4 q = reinterpret_cast<const float4 *>(gG)[p];
      gv.x += q.x;
      gv.y += q.y;
      gv.z += q.z;
      gv.w += q.w;
    }
    float2 o;
    o.x = bwd_ref_one(make_float2(Gv.x, Gv.y), make_float2(gv.x, gv.y), alpha);
    o.y = bwd_ref_one(make_float2(Gv.z, Gv.w), make_float2(gv.z, gv.w), alpha);
    __builtin_nontemporal_store(o.x, &out[2 * p]);
    __builtin_nontemporal_store(o.y, &out[2 * p + 1]);
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
    const long i = n - 1;
    float2 g = make_float2(0.f, 0.f);
    if (HAS_GF) g = make_float2(nan_to_num_default(gf[i].x), nan_to_num_default(gf[i].y));
    if (HAS_GG) { g.x += gG[i].x; g.y += gG[i].y; }
    out[i] = bwd_ref_one(G[i], g, alpha);
  }
}

// ---- the caller's elementwise steps around the operator (resnet_bk.py:137-167, 216-217) ------------------------
// Prologue: v -> nan_to_num(v, 0, +-v_max) -> v_max tanh(v / v_max) -> h0_diag + v -> bfloat16 (round to nearest
// even) or float32.  The division is a multiplication by the rounded reciprocal, as the tensor library evaluates a
// division by a scalar on the device.
__device__ __forceinline__ float saturate_potential(float v, float v_max, float inv_v_max) {
  if (v != v) v = 0.0f;
  else if (v > 3.4028234663852886e38f) v = v_max;
  else if (v < -3.4028234663852886e38f) v = -v_max;
  return v_max * tanhf(v * inv_v_max);
}
__device__ __forceinline__ uint16_t to_bf16_rne(float x) {
  const unsigned u = __float_as_uint(x);
  if (x != x) return (uint16_t)0x7FC0u;
  return (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}
__device__ __forceinline__ float bf16_round(float x) { return __uint_as_float((unsigned)to_bf16_rne(x) << 16); }

template <bool OUT_BF16, bool VEC>  // VEC: N and ldv multiples of 4, 16-byte aligned rows: four elements per thread
__global__ __launch_bounds__(256) void bk_layer_prologue_kernel(const float *__restrict__ v_pre, long ldv,
                                                                const float *__restrict__ h0_diag, float v_max,
                                                                float inv_v_max, long B, long N,
                                                                void *__restrict__ he_diag) {
  for (long row = blockIdx.y; row < B; row += gridDim.y) {
    const float *__restrict__ v = v_pre + row * ldv;
    if (VEC) {
      for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < N / 4; q += (long)gridDim.x * blockDim.x) {
        const float4 vv = reinterpret_cast<const float4 *>(v)[q];
        const float4 hh = reinterpret_cast<const float4 *>(h0_diag)[q];
        float4 he;
        he.x = hh.x + saturate_potential(vv.x, v_max, inv_v_max);
        he.y = hh.y + saturate_potential(vv.y, v_max, inv_v_max);
        he.z = hh.z + saturate_potential(vv.z, v_max, inv_v_max);
        he.w = hh.w + saturate_potential(vv.w, v_max, inv_v_max);
        if (OUT_BF16) {
          uint2 pk;
          pk.x = (unsigned)to_bf16_rne(he.x) | ((unsigned)to_bf16_rne(he.y) << 16);
          pk.y = (unsigned)to_bf16_rne(he.z) | ((unsigned)to_bf16_rne(he.w) << 16);
          reinterpret_cast<uint2 *>(reinterpret_cast<uint16_t *>(he_diag) + row * N)[q] = pk;
        } else {
          reinterpret_cast<float4 *>(reinterpret_cast<float *>(he_diag) + row * N)[q] = he;
        }
      }
    } else {
      for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (long)gridDim.x * blockDim.x) {
        const float he = h0_diag[i] + saturate_potential(v[i], v_max, inv_v_max);
        if (OUT_BF16) reinterpret_cast<uint16_t *>(he_diag)[row * N + i] = to_bf16_rne(he);
        else reinterpret_cast<float *>(he_diag)[row * N + i] = he;
      }
    }
  }
}

// Backward of the same chain in one pass: clamp mask on the features (the features are Re G and Im G, so the mask is
// recomputed from G), the operator's reference-mode gradient, its rounding to bfloat16 (what the autograd engine does
// to a float32 gradient of a bfloat16 operand), the tanh saturation and the nan_to_num mask.
template <bool HAS_GF, bool HAS_GG>
__global__ __launch_bounds__(256) void bk_layer_backward_ref_kernel(const float2 *__restrict__ G,
                                                                    const float2 *__restrict__ gf,
                                                                    const float2 *__restrict__ gG, float alpha,
                                                                    float fclamp, const float *__restrict__ v_pre,
                                                                    float v_max, float inv_v_max, int round_bf16, long n,
                                                                    float *__restrict__ out) {
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float2 Gv = G[i];
    float2 g = make_float2(0.f, 0.f);
    if (HAS_GF) {
      g = gf[i];
      if (!(fabsf(Gv.x) <= fclamp)) g.x = 0.0f;   // clamp passes the gradient on [-fclamp, fclamp] only
      if (!(fabsf(Gv.y) <= fclamp)) g.y = 0.0f;
      g.x = nan_to_num_default(g.x);
      g.y = nan_to_num_default(g.y);
    }
    if (HAS_GG) {
      const float2 q = gG[i];
      g.x += q.x;
      g.y += q.y;
    }
    float ga = bwd_ref_one(Gv, g, alpha);
    if (round_bf16) ga = bf16_round(ga);
    const float v = v_pre[i];
    float gv = 0.0f;
    if (fabsf(v) <= 3.4028234663852886e38f) {       // nan_to_num passes the gradient of finite entries only
      const float th = tanhf(v * inv_v_max);
      gv = ((ga * v_max) * (1.0f - th * th)) * inv_v_max;
    }
    __builtin_nontemporal_store(gv, &out[i]);
  }
}

thread_local int g_last_hip_error = 0;
int g_call_counter = 0;

inline int chunks_of(int64_t N) { return (int)((N + CHUNK - 1) / CHUNK); }

// How a (B, N) problem is cut: segments per row (1 = one workgroup per row) and chunks per segment.  Rows are cut
// only when there are too few of them to occupy half of the 256 CUs and they are long enough (measured break-even:
// about 140 rows).
inline void split_plan(int64_t B, int64_t N, int &S, int &seg_ch) {
  const int nch = chunks_of(N);
  S = 1;
  seg_ch = nch;
  if (B > 128 || nch < 8) return;  // up to 256 rows run as one round of one workgroup per CU, which is faster
  int want = (int)((1024 + B - 1) / B);  // segments per row that give about 1024 workgroups
  if (want > nch / 2) want = nch / 2;     // at least two chunks per segment
  if (want > MAX_SEGMENTS) want = MAX_SEGMENTS;
  if (want < 2) return;
  seg_ch = (nch + want - 1) / want;
  S = (nch + seg_ch - 1) / seg_ch;
}

template <typename TIn, bool PHI_EXACT, bool SAVE>
int launch_forward2(const void *a, int64_t lda, const void *b, int64_t ldb, const void *c, int64_t ldc,
                    const float *z_dev, float z_re, float z_im, int64_t B, int64_t N, float *G, float *features,
                    void *ws, void *state, unsigned flags, hipStream_t stream, float *save_r, float *save_s,
                    float fclamp = __builtin_huge_valf()) {
  int call_id = 0;
  if (state) {
    call_id = __atomic_add_fetch(&g_call_counter, 1, __ATOMIC_RELAXED);
    if (call_id <= 0) {  // wrapped: restart above zero
      __atomic_store_n(&g_call_counter, 1, __ATOMIC_RELAXED);
      call_id = 1;
    }
  }
#define BK_LAUNCH(FASTV, SECOND)                                                                                     \
  hipLaunchKernelGGL((bk_forward_kernel<TIn, PHI_EXACT, FASTV, SAVE>), dim3((unsigned)B), dim3(T), 0, stream,        \
                     (const TIn *)a, (long)lda, (const TIn *)b, (long)ldb, (const TIn *)c, (long)ldc, z_dev, z_re,   \
                     z_im, (long)B, (long)N, (float2 *)G, (float2 *)features, (char *)ws, (int *)state, call_id,     \
                     SECOND, flags, fclamp, (float2 *)save_r, (float2 *)save_s)
  int S = 1, seg_ch = 0;
  if (state && !SAVE) split_plan(B, N, S, seg_ch);
  if (S > 1 && !(flags & BKCORE_FLAG_SCAN_PATH)) {
    // few long rows, ratio form: segment products, combine, segment outputs, then the (normally empty) launch of
    // the scan kernel for rows whose checks failed
    const unsigned grid = (unsigned)(B * S);
#define BK_RATIO_SEG(MODE)                                                                                           \
  hipLaunchKernelGGL((bk_ratio_forward_kernel<TIn, PHI_EXACT, MODE>), dim3(grid), dim3(T), 0, stream, (const TIn *)a, \
                     (long)lda, (const TIn *)b, (long)ldb, (const TIn *)c, (long)ldc, z_dev, z_re, z_im, (long)B,     \
                     (long)N, (float2 *)G, (float2 *)features, (char *)ws, (int *)state, call_id, flags, fclamp, S,   \
                     seg_ch)
    BK_RATIO_SEG(1);
    hipLaunchKernelGGL(bk_ratio_combine_kernel, dim3((unsigned)B), dim3(64), 0, stream, (long)B, (long)N, (char *)ws,
                       (int *)state, call_id, S);
    BK_RATIO_SEG(2);
#undef BK_RATIO_SEG
    hipError_t err = hipGetLastError();
    if (err == hipSuccess) {
      BK_LAUNCH(false, 1);
      err = hipGetLastError();
    }
    if (err != hipSuccess) {
      g_last_hip_error = (int)err;
      return BKCORE_ELAUNCH;
    }
  } else if (S > 1) {
    // few long rows, scan form: several workgroups per row, then the launch for rows whose checks failed
    const unsigned grid = (unsigned)(B * S);
    hipLaunchKernelGGL((bk_split_aggregate_kernel<TIn, PHI_EXACT>), dim3(grid), dim3(T), 0, stream, (const TIn *)a,
                       (long)lda, (const TIn *)b, (long)ldb, (const TIn *)c, (long)ldc, z_dev, z_re, z_im, (long)B,
                       (long)N, (char *)ws, (int *)state, call_id, flags, S, seg_ch);
    hipLaunchKernelGGL(bk_split_combine_kernel, dim3((unsigned)B), dim3(64), 0, stream, (long)B, (long)N,
                       (char *)ws, (int *)state, call_id, S);
    hipLaunchKernelGGL((bk_split_output_kernel<TIn, PHI_EXACT>), dim3(grid), dim3(T), 0, stream, (const TIn *)a,
                       (long)lda, (const TIn *)b, (long)ldb, (const TIn *)c, (long)ldc, z_dev, z_re, z_im, (long)B,
                       (long)N, (float2 *)G, (float2 *)features, (char *)ws, (int *)state, call_id, flags, fclamp, S,
                       seg_ch);
    hipError_t err = hipGetLastError();
    if (err == hipSuccess) {
      BK_LAUNCH(false, 1);
      err = hipGetLastError();
    }
    if (err != hipSuccess) {
      g_last_hip_error = (int)err;
      return BKCORE_ELAUNCH;
    }
  } else if (state && !(flags & BKCORE_FLAG_SCAN_PATH)) {
    // single-sweep ratio-form first launch, then the (normally empty) second launch for rows whose checks failed
    hipLaunchKernelGGL((bk_ratio_forward_kernel<TIn, PHI_EXACT, 0, SAVE>), dim3((unsigned)B), dim3(T), 0, stream,
                       (const TIn *)a, (long)lda, (const TIn *)b, (long)ldb, (const TIn *)c, (long)ldc, z_dev, z_re,
                       z_im, (long)B, (long)N, (float2 *)G, (float2 *)features, (char *)ws, (int *)state, call_id,
                       flags, fclamp, 1, 0, (float2 *)save_r, (float2 *)save_s);
    hipError_t err = hipGetLastError();
    if (err == hipSuccess) {
      BK_LAUNCH(false, 1);
      err = hipGetLastError();
    }
    if (err != hipSuccess) {
      g_last_hip_error = (int)err;
      return BKCORE_ELAUNCH;
    }
  } else if (state) {
    // scan-based fast first launch, then the (normally empty) second launch for rows that need the two-scan path
    BK_LAUNCH(true, 0);
    hipError_t err = hipGetLastError();
    if (err == hipSuccess) {
      BK_LAUNCH(false, 1);
      err = hipGetLastError();
    }
    if (err != hipSuccess) {
      g_last_hip_error = (int)err;
      return BKCORE_ELAUNCH;
    }
  } else {
    // no persistent state block: nothing can carry a verdict to a second launch, so take the two-scan path at once
    BK_LAUNCH(false, 0);
    const hipError_t err = hipGetLastError();
    if (err != hipSuccess) {
      g_last_hip_error = (int)err;
      return BKCORE_ELAUNCH;
    }
  }
#undef BK_LAUNCH
  return BKCORE_OK;
}

template <typename TIn>
int launch_forward(const void *a, int64_t lda, const void *b, int64_t ldb, const void *c, int64_t ldc,
                   const float *z_dev, float z_re, float z_im, int64_t B, int64_t N, float *G, float *features,
                   void *ws, void *state, unsigned flags, hipStream_t s, float fclamp) {
  if (flags & BKCORE_FLAG_PHI_EXACT)
    return launch_forward2<TIn, true, false>(a, lda, b, ldb, c, ldc, z_dev, z_re, z_im, B, N, G, features, ws, state,
                                             flags, s, nullptr, nullptr, fclamp);
  return launch_forward2<TIn, false, false>(a, lda, b, ldb, c, ldc, z_dev, z_re, z_im, B, N, G, features, ws, state,
                                            flags, s, nullptr, nullptr, fclamp);
}

}  // namespace

extern "C" {

size_t bkcore_workspace_bytes(int64_t B, int64_t N) {
  if (B <= 0 || N <= 0) return 0;
  int S = 1, seg_ch = 0;
  split_plan(B, N, S, seg_ch);
  const size_t seg_records = S > 1 ? (size_t)B * (size_t)S * SEG_INFO * sizeof(float) : 0;
  return rowflag_bytes(B) + lane_state_bytes(B, N) + seg_records;
}

int bkcore_forward_clamped(const void *a, int64_t lda, const void *b, int64_t ldb, const void *c, int64_t ldc,
                           int in_dtype, const float *z_dev, float z_re, float z_im, int64_t B, int64_t N, float *G,
                           float *features, float feature_clamp, void *workspace, size_t workspace_bytes, void *state,
                           unsigned flags, void *stream) {
  const float fclamp = feature_clamp > 0.0f ? feature_clamp : __builtin_huge_valf();
  if (feature_clamp != feature_clamp) return BKCORE_EINVAL;
  if (B < 0 || N < 0 || lda < 0 || ldb < 0 || ldc < 0) return BKCORE_EINVAL;
  if (B == 0 || N == 0) return BKCORE_OK;
  if (!a || !G || (N > 1 && (!b || !c))) return BKCORE_EINVAL;
  if (N > bkcore_max_seq_len()) return BKCORE_ETOOLONG;
  if (B > 2147483647LL) return BKCORE_EINVAL;
  const size_t need = bkcore_workspace_bytes(B, N);
  if (need > 0 && (!workspace || workspace_bytes < need)) return BKCORE_EWORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  switch (in_dtype) {
    case BKCORE_F32:
      return launch_forward<float>(a, lda, b, ldb, c, ldc, z_dev, z_re, z_im, B, N, G, features, workspace, state, flags, s,
                                   fclamp);
#ifndef BK_DEV_F32_ONLY
    case BKCORE_BF16:
      return launch_forward<uint16_t>(a, lda, b, ldb, c, ldc, z_dev, z_re, z_im, B, N, G, features, workspace, state, flags, s,
                                   fclamp);
    case BKCORE_F16:
      return launch_forward<_Float16>(a, lda, b, ldb, c, ldc, z_dev, z_re, z_im, B, N, G, features, workspace, state, flags, s,
                                   fclamp);
    case BKCORE_F64:
      return launch_forward<double>(a, lda, b, ldb, c, ldc, z_dev, z_re, z_im, B, N, G, features, workspace, state, flags, s,
                                   fclamp);
#endif
    default:
      return BKCORE_EINVAL;
  }
}

int bkcore_forward(const void *a, int64_t lda, const void *b, int64_t ldb, const void *c, int64_t ldc, int in_dtype,
                   const float *z_dev, float z_re, float z_im, int64_t B, int64_t N, float *G, float *features,
                   void *workspace, size_t workspace_bytes, void *state, unsigned flags, void *stream) {
  return bkcore_forward_clamped(a, lda, b, ldb, c, ldc, in_dtype, z_dev, z_re, z_im, B, N, G, features, 0.0f, workspace,
                                workspace_bytes, state, flags, stream);
}

int bkcore_internal_forward_save(const float *a, int64_t lda, const float *b, int64_t ldb, const float *c, int64_t ldc,
                                 const float *z_dev, float z_re, float z_im, int64_t B, int64_t N, float *G,
                                 void *workspace, void *state, unsigned flags, void *stream, float *save_r,
                                 float *save_s) {
  hipStream_t s = (hipStream_t)stream;
  if (flags & BKCORE_FLAG_PHI_EXACT)
    return launch_forward2<float, true, true>(a, lda, b, ldb, c, ldc, z_dev, z_re, z_im, B, N, G, nullptr, workspace,
                                              state, flags, s, save_r, save_s);
  return launch_forward2<float, false, true>(a, lda, b, ldb, c, ldc, z_dev, z_re, z_im, B, N, G, nullptr, workspace,
                                             state, flags, s, save_r, save_s);
}

int bkcore_backward_ref(const float *G, const float *grad_features, const float *grad_G, float alpha, int64_t n,
                        float *grad_a, void *stream) {
  if (n < 0) return BKCORE_EINVAL;
  if (n == 0) return BKCORE_OK;
  if (!G || !grad_a) return BKCORE_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const long pairs = (n + 1) / 2;
  long blocks = (pairs + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  const dim3 grid((unsigned)blocks), block(256);
  const float2 *Gp = (const float2 *)G, *gf = (const float2 *)grad_features, *gg = (const float2 *)grad_G;
  if (gf && gg)
    hipLaunchKernelGGL((bk_backward_ref_kernel<true, true>), grid, block, 0, s, Gp, gf, gg, alpha, (long)n, grad_a);
  else if (gf)
    hipLaunchKernelGGL((bk_backward_ref_kernel<true, false>), grid, block, 0, s, Gp, gf, gg, alpha, (long)n, grad_a);
  else if (gg)
    hipLaunchKernelGGL((bk_backward_ref_kernel<false, true>), grid, block, 0, s, Gp, gf, gg, alpha, (long)n, grad_a);
  else
    hipLaunchKernelGGL((bk_backward_ref_kernel<false, false>), grid, block, 0, s, Gp, gf, gg, alpha, (long)n, grad_a);
  const hipError_t err = hipGetLastError();
  if (err != hipSuccess) {
    g_last_hip_error = (int)err;
    return BKCORE_ELAUNCH;
  }
  return BKCORE_OK;
}

int bkcore_layer_prologue(const float *v_pre, int64_t ldv, const float *h0_diag, float v_max, int64_t B, int64_t N,
                          void *he_diag, int out_dtype, void *stream) {
  if (B < 0 || N < 0 || ldv < 0 || !(v_max > 0.0f)) return BKCORE_EINVAL;
  if (out_dtype != BKCORE_BF16 && out_dtype != BKCORE_F32) return BKCORE_EINVAL;
  if (B == 0 || N == 0) return BKCORE_OK;
  if (!v_pre || !h0_diag || !he_diag) return BKCORE_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  long gx = (N + 1023) / 1024;
  if (gx > 1024) gx = 1024;
  const dim3 grid((unsigned)gx, (unsigned)(B > 32768 ? 32768 : B)), block(256);
  const float inv = 1.0f / v_max;
  const bool vec = (N % 4) == 0 && (ldv % 4) == 0 && ((uintptr_t)v_pre % 16) == 0 && ((uintptr_t)h0_diag % 16) == 0 &&
                   ((uintptr_t)he_diag % 16) == 0;
#define BK_PRO(BF, VEC)                                                                                              \
  hipLaunchKernelGGL((bk_layer_prologue_kernel<BF, VEC>), grid, block, 0, s, v_pre, (long)ldv, h0_diag, v_max, inv,  \
                     (long)B, (long)N, he_diag)
  if (out_dtype == BKCORE_BF16) {
    if (vec) BK_PRO(true, true);
    else BK_PRO(true, false);
  } else {
    if (vec) BK_PRO(false, true);
    else BK_PRO(false, false);
  }
#undef BK_PRO
  const hipError_t err = hipGetLastError();
  if (err != hipSuccess) {
    g_last_hip_error = (int)err;
    return BKCORE_ELAUNCH;
  }
  return BKCORE_OK;
}

int bkcore_layer_backward_ref(const float *G, const float *grad_features, const float *grad_G, float alpha,
                              float feature_clamp, const float *v_pre, float v_max, int round_bf16, int64_t n,
                              float *grad_v, void *stream) {
  if (n < 0 || !(v_max > 0.0f)) return BKCORE_EINVAL;
  if (n == 0) return BKCORE_OK;
  if (!G || !v_pre || !grad_v) return BKCORE_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const float fc = feature_clamp > 0.0f ? feature_clamp : __builtin_huge_valf();
  long blocks = (n + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  const dim3 grid((unsigned)blocks), block(256);
  const float inv = 1.0f / v_max;
  const float2 *Gp = (const float2 *)G, *gf = (const float2 *)grad_features, *gg = (const float2 *)grad_G;
#define BK_LB(GF, GG)                                                                                                \
  hipLaunchKernelGGL((bk_layer_backward_ref_kernel<GF, GG>), grid, block, 0, s, Gp, gf, gg, alpha, fc, v_pre, v_max, \
                     inv, round_bf16, (long)n, grad_v)
  if (gf && gg) BK_LB(true, true);
  else if (gf) BK_LB(true, false);
  else if (gg) BK_LB(false, true);
  else BK_LB(false, false);
#undef BK_LB
  const hipError_t err = hipGetLastError();
  if (err != hipSuccess) {
    g_last_hip_error = (int)err;
    return BKCORE_ELAUNCH;
  }
  return BKCORE_OK;
}

int64_t bkcore_max_seq_len(void) { return (int64_t)MAX_N; }
int bkcore_last_hip_error(void) { return g_last_hip_error; }
int bkcore_abi_version(void) { return 8; }

const char *bkcore_error_string(int code) {
  switch (code) {
    case BKCORE_OK: return "ok";
    case BKCORE_EINVAL: return "invalid argument";
    case BKCORE_ETOOLONG: return "sequence length exceeds bkcore_max_seq_len()";
    case BKCORE_ELAUNCH: return "HIP kernel launch failed (see bkcore_last_hip_error())";
    case BKCORE_EWORKSPACE: return "workspace missing or smaller than bkcore_workspace_bytes()";
    default: return "unknown bkcore error";
  }
}

}  // extern "C"

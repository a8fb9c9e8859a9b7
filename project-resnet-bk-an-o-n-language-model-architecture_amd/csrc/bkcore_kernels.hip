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
#include "bkcore_device.h"

// Device memory of bkcore_forward_noalloc (one copy per GPU the library runs on, held until the process ends).
#ifndef BK_COMPLEX_BUILD
static __device__ __align__(256) unsigned char bk_static_scratch[256 + BKCORE_NOALLOC_MAX_WORKSPACE];
#endif

namespace {


// ---- lane-level serial pieces -------------------------------------------------------------------------------
// Element order inside a lane: ascending for theta, descending (REV) for phi.  kw holds k_{i-1} for the lane's
// elements plus one more (k of the lane's last element); KO selects which one a step uses.

// One more element folded into the running product A <- M(al, k) * A.
__device__ __forceinline__ void agg_step(Mat &A, float2 al, KT k) {
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
__device__ __forceinline__ Mat lane_aggregate(const float2 (&al)[E], const KT (&kw)[E + 1], unsigned vmask) {
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
__device__ __forceinline__ Mat wall_and_aggregate(const float2 (&al)[E], const KT (&kw)[E + 1], unsigned vmask,
                                                  Vec x, KT (&k2)[E]) {
  Mat A = mat_identity();
  float l0 = log2_norm2(x.x0);
  float l1 = log2_norm2(x.x1);
#pragma unroll
  for (int j = 0; j < E; ++j) {
    const int ei = REV ? E - 1 - j : j;
    const float d = LN2_HALF * (l1 - l0);  // ln|prev2| - ln|prev|
    const KT kk = kscale(kw[ei + KO], wall_corr(d));
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
__device__ __forceinline__ Vec warm_run(const float2 (&al)[E], const KT (&kw)[E + 1], unsigned vmask) {
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
__device__ __forceinline__ Vec estimate_incoming(const float2 (&al)[E], const KT (&kw)[E + 1], unsigned vmask,
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
__device__ __forceinline__ void theta_lane(const float2 (&al)[E], const KT (&kw)[E + 1], unsigned vmask, Vec x,
                                           float2 *__restrict__ om, int (&oe)[E / 2], float2 *__restrict__ save_r) {
  float2 u0 = x.x0, u1 = x.x1;  // linear chain
  float l0 = log2_norm2(u0);
  float l1 = log2_norm2(u1);
#pragma unroll
  for (int j = 0; j < E; ++j) {
    om[j] = x.x0;
    if ((j & 1) == 0) oe[j >> 1] = x.e;
    const float d = LN2_HALF * (l1 - l0);
    const KT kk = kscale(kw[j], wall_corr(d));
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

// Stages chunk at `base` of one row into LDS: a_s[idx] = a[base+idx], k_s[idx] = b[g] c[g] with g = base-1+idx
// (k_s holds CHUNK+1 values; indices outside the row give a = 1, k = 0).
// FULL: every index idx < CHUNK of all three operands lies inside the row (interior chunks): no bounds logic.
// Otherwise the loads use clamped (always valid) indices and out-of-row values are replaced afterwards.
// Returns bit0: NaN seen in a, bit1: NaN seen in b or c.
#ifdef BK_COMPLEX_BUILD
// Complex operands (interleaved complex64), raw semantics only: nothing is clamped or repaired (bk_core.py:42-56).
template <typename TIn, bool FULL>
__device__ __forceinline__ unsigned stage_chunk(const TIn *__restrict__ a, const TIn *__restrict__ b,
                                                const TIn *__restrict__ c, long base, const RowCfg &cfg, Shared &sh) {
  static_assert(std::is_same<TIn, float2>::value, "the complex build takes complex64 operands");
  const int t = threadIdx.x;
  const float2 *__restrict__ ap = a + base;
  const int a_hi = (int)min((long)CHUNK, cfg.N - base) - 1;
#pragma unroll
  for (int j = 0; j < E; ++j) {
    const int idx = t + j * T;
    float2 v = ap[FULL ? idx : min(idx, a_hi)];
    if (!FULL && idx > a_hi) v = make_float2(1.0f, 0.0f);
    sh.u.in.a_s[pad32(idx)] = v.x;
    sh.u.in.a_i[pad32(idx)] = v.y;
  }
  const int k_lo = base == 0 ? 1 : 0;
  const int k_hi = (int)min((long)CHUNK, cfg.N - 1 - base);
  if (FULL || k_hi >= k_lo) {
    const float2 *__restrict__ bp = b + (base - 1);
    const float2 *__restrict__ cp = c + (base - 1);
#pragma unroll
    for (int j = 0; j <= E; ++j) {
      if (j < E || t == 0) {
        const int idx = t + j * T;   // j == E: the one extra value k_s[CHUNK], by thread 0
        const int ii = min(max(idx, k_lo), k_hi);
        const bool ok = idx >= k_lo && idx <= k_hi;
        sh.u.in.k_s[pad32(idx)] = ok ? cmul(bp[ii], cp[ii]) : kzero();
      }
    }
  } else {
#pragma unroll
    for (int j = 0; j <= E; ++j)
      if (j < E || t == 0) sh.u.in.k_s[pad32(t + j * T)] = kzero();
  }
  return 0u;
}

__device__ __forceinline__ void fetch_elems(const Shared &sh, long base, int seg, const RowCfg &cfg, float2 (&al)[E],
                                            KT (&kw)[E + 1], unsigned &vmask) {
#pragma unroll
  for (int e = 0; e <= E; ++e) kw[e] = sh.u.in.k_s[pad32(seg * E + e)];
  const long first = base + (long)seg * E;
  const long left = cfg.N - first;
  vmask = left >= E ? ((1u << E) - 1u) : (left <= 0 ? 0u : ((1u << (int)left) - 1u));
#pragma unroll
  for (int e = 0; e < E; ++e) {
    // reg(a - z) = x + 1e-3 x / |x|, and x + 1e-3 where |x| < 1e-12 (bk_core.py:78-94)
    const float xr = sh.u.in.a_s[pad32(seg * E + e)] - cfg.z.x;
    const float xi = sh.u.in.a_i[pad32(seg * E + e)] - cfg.z.y;
    const float n = fmaf(xr, xr, xi * xi);
    float f = fmaf(1e-3f, __builtin_amdgcn_rsqf(n), 1.0f);
    float add = 0.0f;
    if (!(n >= 1e-24f)) { f = 1.0f; add = 1e-3f; }
    al[e] = make_float2(fmaf(xr, f, add), xi * f);
  }
}
#else
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
                                            KT (&kw)[E + 1], unsigned &vmask) {
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

#endif  // BK_COMPLEX_BUILD (staging)

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
__device__ __forceinline__ Vec two_pass(const float2 (&al)[E], const KT (&kw)[E + 1], unsigned vmask,
                                        KT (&k2)[E], Shared &sh, ScanCtx &ctx, bool &bad) {
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
__device__ __forceinline__ void sweepB_chunk(const float2 (&al)[E], const KT (&kw)[E + 1], unsigned vmask,
                                             const Vec &xt, float2 inv_det, int e_det, Shared &sh, ScanCtx &ctx,
                                             int seg, bool &bad, float2 *__restrict__ save_r,
                                             float2 *__restrict__ save_s) {
  KT k2[E];
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


// ---- the two sweeps over a range of chunks ---------------------------------------------------------------------
// Three kinds of chunk: the row's first (no k to its left), interior ones (every load in range: no bounds logic),
// and the row's last (ragged end).  Each kind is handled outside the loop over interior chunks, so that loop's body
// holds a single variant of the staging and scan code.

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
    KT kw[E + 1];
    unsigned vmask;
    __syncthreads();  // previous chunk's LDS reads done
    seen |= stage_chunk<TIn, FULL>(a, b, c, base, cfg, sh);
    __syncthreads();
    fetch_elems(sh, base, t, cfg, al, kw, vmask);
    KT k2[E];
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
    KT kw[E + 1];
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


// The state block: state[0] a NaN was seen in he_diag, state[1] a NaN was seen in b or c, state[2] some row failed its
// checks -- raised (to 1) by the first launch(es) of a call, read by every workgroup of the call's last launch, which
// is always the second-launch form of bk_forward_kernel.  Clean calls leave these all zero and cost nothing.
// state[9] counts the first launches that cut rows into units handing over inside the launch (the epoch their records
// are tagged with; moved on by the call's last launch), state[12] says that such a hand-over was given up and every
// row is to be recomputed.  When a flag is up, every workgroup of that last launch takes a ticket in state[4] once it has read the flags and done its
// work, and the last ticket puts the block back to zero for the next call on the stream.  All of it lives on the
// device, so a captured forward replays correctly in a HIP graph.
__device__ __forceinline__ void state_release(int *state, bool redone = false, bool adaptive = false) {
  __syncthreads();
  if (threadIdx.x == 0) {
    if (adaptive && redone) __hip_atomic_fetch_add(&state[6], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int ticket = __hip_atomic_fetch_add(&state[4], 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (ticket == (int)gridDim.x - 1) {
      if (adaptive) {
        // state[6] rows recomputed by this launch; state[7] "most rows were recomputed": the next calls with
        // BKCORE_FLAG_ADAPTIVE skip the fast path's arithmetic, except when state[8] (calls since) wraps to 0
        const int cnt = __hip_atomic_load(&state[6], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int since = __hip_atomic_load(&state[8], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool most = 2 * (long)cnt > (long)gridDim.x;
        __hip_atomic_store(&state[7], most ? 1 : 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&state[8], most ? ((since + 1) & 15) : 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&state[6], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      __hip_atomic_store(&state[0], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&state[1], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&state[2], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&state[4], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&state[12], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// FAST = first launch (fast pass 1, verified); !FAST = second launch: rows whose verification failed, or every row
// when he_diag held a NaN and b/c held a NaN (then with b/c NaNs replaced by 1, bk_core.py:260-264).
template <typename TIn, bool PHI_EXACT, bool FAST, bool SAVE>
__global__ __launch_bounds__(T, BK_WAVES_PER_SIMD) void bk_forward_kernel(
    const TIn *__restrict__ a_, long lda, const TIn *__restrict__ b_, long ldb, const TIn *__restrict__ c_, long ldc,
    const float *__restrict__ z_dev, float z_re, float z_im, long B, long N, float2 *__restrict__ G_,
    float2 *__restrict__ F_, char *__restrict__ ws, int *state, int second, unsigned flags, float fclamp,
    float2 *__restrict__ save_r_, float2 *__restrict__ save_s_, const int *__restrict__ row_filter = nullptr,
    const int *gate = nullptr) {
  __shared__ Shared sh;
  const long row = blockIdx.x;
  // the analytic backward's launches for the rows its sweeps flagged: nothing to do unless *gate is up, and then only
  // for rows with row_filter[row] != 0
  if (gate && __hip_atomic_load(gate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) return;
  if (row_filter && row_filter[row] == 0) return;
  int *rowflag = reinterpret_cast<int *>(ws);
  bool fix_bc = false;
  if (second) {
    const int sa = __hip_atomic_load(&state[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int sb = __hip_atomic_load(&state[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int sc = __hip_atomic_load(&state[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // state[12]: a hand-off inside the first launch was given up (rows cut into units of one launch): every row again
    const int sd = __hip_atomic_load(&state[12], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // second & 3 == 2 / 3: the first launch cut rows into units / chunks that hand over inside the launch (second >> 2
    // records per row) and tag their records with the epoch in state[9].  The next launch needs a new epoch, and every
    // tag of this one comes down here: a record whose reader gave up, or one whose epoch repeats because the state
    // block was zeroed again (the memset of a replayed graph), must not pass for a hand-over of a later call.
    if ((second & 3) >= 2) {
      if (blockIdx.x == 0 && threadIdx.x == 0) {
        __hip_atomic_fetch_add(&state[9], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&state[10], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // the chunk counter of mode 4
      }
      const int per_row = second >> 2;
      float *recs = reinterpret_cast<float *>(ws + rowflag_bytes(B));
      if ((second & 3) == 2) {
        for (int i = 1 + (int)threadIdx.x; i < per_row; i += T)
          __hip_atomic_store(reinterpret_cast<unsigned long long *>(recs + (row * per_row + i) * 16 + 8), 0ull,
                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        for (int i = (int)threadIdx.x; i < per_row; i += T) {
          float *rec = recs + (row * per_row + i) * 32;
          __hip_atomic_store(reinterpret_cast<unsigned long long *>(rec + 12), 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(reinterpret_cast<unsigned long long *>(rec + 28), 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }
    if (!(sa | sb | sc | sd)) return;  // nothing was flagged by the first launch: the usual case
    fix_bc = (sa != 0 && sb != 0) && !(flags & BKCORE_FLAG_RAW);
    const bool redo_row = sd != 0 || (sc != 0 && rowflag[row] != 0);
    if (!fix_bc && !redo_row) {
      state_release(state, false, (flags & BKCORE_FLAG_ADAPTIVE) != 0);
      return;
    }
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
      if (any_bad) __hip_atomic_store(&state[2], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (state && !second && seen) {
    if (seen & 1u) __hip_atomic_store(&state[0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (seen & 2u) __hip_atomic_store(&state[1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (second) state_release(state, true, (flags & BKCORE_FLAG_ADAPTIVE) != 0);
}

#ifndef BK_COMPLEX_BUILD
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


template <typename TIn, bool PHI_EXACT>
__global__ __launch_bounds__(T, BK_WAVES_PER_SIMD) void bk_split_aggregate_kernel(
    const TIn *__restrict__ a_, long lda, const TIn *__restrict__ b_, long ldb, const TIn *__restrict__ c_, long ldc,
    const float *__restrict__ z_dev, float z_re, float z_im, long B, long N, char *__restrict__ ws, int *state, unsigned flags, int S, int seg_ch) {
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
    if (seen & 1u) __hip_atomic_store(&state[0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (seen & 2u) __hip_atomic_store(&state[1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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


// One wave per row, one lane per segment (64 at a time): checks every boundary estimate against the neighbouring
// segment's outgoing state, forms the running factors of theta (C_s, left to right) and phi (D_s, right to left)
// as products of complex scalars with exponents (wave scans over shuffles), and writes each segment's epilogue
// factor C_s D_s / theta_N.
__global__ __launch_bounds__(64) void bk_split_combine_kernel(long B, long N, char *__restrict__ ws, int *state, int S) {
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
    if (any_bad) __hip_atomic_store(&state[2], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

template <typename TIn, bool PHI_EXACT>
__global__ __launch_bounds__(T, BK_WAVES_PER_SIMD) void bk_split_output_kernel(
    const TIn *__restrict__ a_, long lda, const TIn *__restrict__ b_, long ldb, const TIn *__restrict__ c_, long ldc,
    const float *__restrict__ z_dev, float z_re, float z_im, long B, long N, float2 *__restrict__ G_,
    float2 *__restrict__ F_, char *__restrict__ ws, int *state, unsigned flags, float fclamp, int S,
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
    __hip_atomic_store(&state[2], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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

// The reference's other backward heuristics on the same saved G, as variants of the same elementwise pass:
//   1: compute_single_gradient, src/models/batched_gradient.py:12-54 -- the blend with min |G^2| 1e-3, clamp +-1000 and
//      the cotangent taken as it comes (not conjugated); a result that is not finite counts as 0
//   2: BKScanStableFunction.backward, src/kernels/bk_scan_stable.py:597 -- grad_a = Re(-g G conj(G)) = -Re(g) |G|^2
template <int VARIANT>
__device__ __forceinline__ float bwd_one(float2 G, float2 g, float alpha) {
  if (VARIANT == 0) return bwd_ref_one(G, g, alpha);
  if (VARIANT == 2) return -g.x * fmaf(G.x, G.x, G.y * G.y);
  const float2 g2 = cmul(G, G);
  const float mag = sqrtf(fmaf(g2.x, g2.x, g2.y * g2.y));
  float2 den = g2;
  if (mag < 1e-3f) {
    const float s = 1e-3f / (mag + 1e-9f);
    den = make_float2(g2.x * s, g2.y * s);
  }
  den.x += 1e-6f;
  const float theo = -fmaf(g.x, g2.x, -g.y * g2.y);              // -Re(g G^2)
  const float dn = fmaf(den.x, den.x, den.y * den.y);
  const float h7 = -fmaf(g.x, den.x, g.y * den.y) / dn;           // -Re(g / den)
  float gv = fmaf(1.0f - alpha, theo, alpha * h7);
  if (!(fabsf(gv) <= 3.4028234663852886e38f)) gv = 0.0f;
  return fminf(fmaxf(gv, -1000.0f), 1000.0f);
}

// Four elements per thread per iteration: two 16-byte loads of G and of each cotangent (read once: nontemporal), one
// 16-byte store.  The loads need only the 8-byte alignment of a complex64 / float pair.
typedef float bk_v4f __attribute__((ext_vector_type(4)));
struct __attribute__((aligned(8))) V4a8 { bk_v4f v; };
struct __attribute__((aligned(4))) V4a4 { bk_v4f v; };
template <bool HAS_GF, bool HAS_GG, int VARIANT = 0>
__global__ __launch_bounds__(256) void bk_backward_ref_kernel(const float2 *__restrict__ G, const float2 *__restrict__ gf,
                                                              const float2 *__restrict__ gG, float alpha, long n,
                                                              float *__restrict__ out, int *__restrict__ nonfinite) {
  const long stride = (long)gridDim.x * blockDim.x;
  const long nquads = n >> 2;
  bool seen = false;  // a NaN or an infinity in grad_features (reported through *nonfinite when that is given)
  for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < nquads; p += stride) {
    const V4a8 *Gq = reinterpret_cast<const V4a8 *>(G + 4 * p);
    const bk_v4f G0 = __builtin_nontemporal_load(&Gq[0].v), G1 = __builtin_nontemporal_load(&Gq[1].v);
    bk_v4f g0 = {0.f, 0.f, 0.f, 0.f}, g1 = {0.f, 0.f, 0.f, 0.f};
    if (HAS_GF) {
      const V4a8 *q = reinterpret_cast<const V4a8 *>(gf + 4 * p);
      g0 = __builtin_nontemporal_load(&q[0].v);
      g1 = __builtin_nontemporal_load(&q[1].v);
      if (nonfinite) {
#pragma unroll
        for (int i = 0; i < 4; ++i) seen |= !(fabsf(g0[i]) <= 3.4028234663852886e38f) || !(fabsf(g1[i]) <= 3.4028234663852886e38f);
      }
      if (VARIANT == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          g0[i] = nan_to_num_default(g0[i]);
          g1[i] = nan_to_num_default(g1[i]);
        }
      }
    }
    if (HAS_GG) {
      const V4a8 *q = reinterpret_cast<const V4a8 *>(gG + 4 * p);
      g0 += __builtin_nontemporal_load(&q[0].v);
      g1 += __builtin_nontemporal_load(&q[1].v);
    }
    bk_v4f o;
    o[0] = bwd_one<VARIANT>(make_float2(G0[0], G0[1]), make_float2(g0[0], g0[1]), alpha);
    o[1] = bwd_one<VARIANT>(make_float2(G0[2], G0[3]), make_float2(g0[2], g0[3]), alpha);
    o[2] = bwd_one<VARIANT>(make_float2(G1[0], G1[1]), make_float2(g1[0], g1[1]), alpha);
    o[3] = bwd_one<VARIANT>(make_float2(G1[2], G1[3]), make_float2(g1[2], g1[3]), alpha);
    __builtin_nontemporal_store(o, &reinterpret_cast<V4a4 *>(out + 4 * p)->v);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const long i = (n & ~3L) + threadIdx.x;
    float2 g = make_float2(0.f, 0.f);
    if (HAS_GF) {
      seen |= !(fabsf(gf[i].x) <= 3.4028234663852886e38f) || !(fabsf(gf[i].y) <= 3.4028234663852886e38f);
      g = VARIANT == 0 ? make_float2(nan_to_num_default(gf[i].x), nan_to_num_default(gf[i].y)) : gf[i];
    }
    if (HAS_GG) { g.x += gG[i].x; g.y += gG[i].y; }
    out[i] = bwd_one<VARIANT>(G[i], g, alpha);
  }
  if (nonfinite && seen) __hip_atomic_store(nonfinite, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
      // PQ quads per thread and pass, a workgroup's threads side by side within each: all loads of a pass are issued
      // before the first tanh (one quad per thread left two loads in flight per thread and the kernel waiting on them)
      constexpr int PQ = 4;
      const long nq = N / 4;
      for (long q0 = (long)blockIdx.x * (blockDim.x * PQ) + threadIdx.x; q0 < nq; q0 += (long)gridDim.x * blockDim.x * PQ) {
        float4 vv[PQ], hh[PQ];
#pragma unroll
        for (int k = 0; k < PQ; ++k) {
          const long q = q0 + (long)k * blockDim.x;
          if (q < nq) {
            vv[k] = reinterpret_cast<const float4 *>(v)[q];
            hh[k] = reinterpret_cast<const float4 *>(h0_diag)[q];
          }
        }
#pragma unroll
        for (int k = 0; k < PQ; ++k) {
          const long q = q0 + (long)k * blockDim.x;
          if (q >= nq) break;
          float4 he;
          he.x = hh[k].x + saturate_potential(vv[k].x, v_max, inv_v_max);
          he.y = hh[k].y + saturate_potential(vv[k].y, v_max, inv_v_max);
          he.z = hh[k].z + saturate_potential(vv[k].z, v_max, inv_v_max);
          he.w = hh[k].w + saturate_potential(vv[k].w, v_max, inv_v_max);
          if (OUT_BF16) {
            uint2 pk;
            pk.x = (unsigned)to_bf16_rne(he.x) | ((unsigned)to_bf16_rne(he.y) << 16);
            pk.y = (unsigned)to_bf16_rne(he.z) | ((unsigned)to_bf16_rne(he.w) << 16);
            reinterpret_cast<uint2 *>(reinterpret_cast<uint16_t *>(he_diag) + row * N)[q] = pk;
          } else {
            reinterpret_cast<float4 *>(reinterpret_cast<float *>(he_diag) + row * N)[q] = he;
          }
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
  // one element: clamp mask, the operator's reference-mode gradient, its rounding, tanh and nan_to_num masks
  auto one = [&](float2 Gv, float2 gfv, float2 gGv, float v) -> float {
    float2 g = make_float2(0.f, 0.f);
    if (HAS_GF) {
      g = gfv;
      if (!(fabsf(Gv.x) <= fclamp)) g.x = 0.0f;   // clamp passes the gradient on [-fclamp, fclamp] only
      if (!(fabsf(Gv.y) <= fclamp)) g.y = 0.0f;
      g.x = nan_to_num_default(g.x);
      g.y = nan_to_num_default(g.y);
    }
    if (HAS_GG) {
      g.x += gGv.x;
      g.y += gGv.y;
    }
    float ga = bwd_ref_one(Gv, g, alpha);
    if (round_bf16) ga = bf16_round(ga);
    float gv = 0.0f;
    if (fabsf(v) <= 3.4028234663852886e38f) {       // nan_to_num passes the gradient of finite entries only
      const float th = tanhf(v * inv_v_max);
      gv = ((ga * v_max) * (1.0f - th * th)) * inv_v_max;
    }
    return gv;
  };
  // four elements per thread per iteration, 16-byte nontemporal accesses (everything is read once)
  const long stride = (long)gridDim.x * blockDim.x;
  const long nquads = n >> 2;
  for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < nquads; p += stride) {
    const V4a8 *Gq = reinterpret_cast<const V4a8 *>(G + 4 * p);
    const bk_v4f G0 = __builtin_nontemporal_load(&Gq[0].v), G1 = __builtin_nontemporal_load(&Gq[1].v);
    bk_v4f g0 = {0.f, 0.f, 0.f, 0.f}, g1 = g0, h0 = g0, h1 = g0;
    if (HAS_GF) {
      const V4a8 *q = reinterpret_cast<const V4a8 *>(gf + 4 * p);
      g0 = __builtin_nontemporal_load(&q[0].v);
      g1 = __builtin_nontemporal_load(&q[1].v);
    }
    if (HAS_GG) {
      const V4a8 *q = reinterpret_cast<const V4a8 *>(gG + 4 * p);
      h0 = __builtin_nontemporal_load(&q[0].v);
      h1 = __builtin_nontemporal_load(&q[1].v);
    }
    const bk_v4f vq = __builtin_nontemporal_load(&reinterpret_cast<const V4a4 *>(v_pre + 4 * p)->v);
    bk_v4f o;
    o[0] = one(make_float2(G0[0], G0[1]), make_float2(g0[0], g0[1]), make_float2(h0[0], h0[1]), vq[0]);
    o[1] = one(make_float2(G0[2], G0[3]), make_float2(g0[2], g0[3]), make_float2(h0[2], h0[3]), vq[1]);
    o[2] = one(make_float2(G1[0], G1[1]), make_float2(g1[0], g1[1]), make_float2(h1[0], h1[1]), vq[2]);
    o[3] = one(make_float2(G1[2], G1[3]), make_float2(g1[2], g1[3]), make_float2(h1[2], h1[3]), vq[3]);
    __builtin_nontemporal_store(o, &reinterpret_cast<V4a4 *>(out + 4 * p)->v);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const long i = (n & ~3L) + threadIdx.x;
    out[i] = one(G[i], HAS_GF ? gf[i] : make_float2(0.f, 0.f), HAS_GG ? gG[i] : make_float2(0.f, 0.f), v_pre[i]);
  }
}

thread_local int g_last_hip_error = 0;

// Scratch of bkcore_forward_noalloc: state block + workspace in device memory that belongs to the code object.
constexpr size_t STATIC_STATE_BYTES = 256;
constexpr size_t STATIC_WS_BYTES = (size_t)BKCORE_NOALLOC_MAX_WORKSPACE;
constexpr int MAX_DEVICES = 64;
unsigned char *g_static_scratch_ptr[MAX_DEVICES] = {};

// How a (B, N) problem is cut: segments per row (1 = one workgroup per row) and chunks per segment, in chunks of
// `chunk` elements (the tile of the kernel that will run: the ratio-form kernel's or the scan kernel's).  Rows are cut
// only when there are too few of them to occupy half of the 256 CUs and they are long enough (measured break-even:
// about 140 rows).
inline void split_plan(int64_t B, int64_t N, int chunk, int &S, int &seg_ch) {
  const int nch = (int)((N + chunk - 1) / chunk);
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

// Many rows, one launch (mode 3 of the ratio-form kernel): with more rows than the GPU holds workgroups at once, whole
// rows leave a last round in which most CUs hold a single workgroup (1024 rows on 3 x 256 places: the launch lasts
// about 1.55 row times where 1.33 would do).  Rows cut into S units of seg_ch chunks that hand the running product on
// inside the launch fill that round evenly.  S = 1: keep one workgroup per row.
inline int device_cu_count() {
  static int cus[MAX_DEVICES] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES) return 256;
  if (cus[dev] == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cus[dev] = n;
  }
  return cus[dev];
}
inline void relay_plan(int64_t B, int64_t N, int chunk, int wgs_per_cu, unsigned flags, int &S, int &seg_ch,
                       int cus = 0) {
  const int nch = (int)((N + chunk - 1) / chunk);
  S = 1;
  seg_ch = nch;
  // development: BKCORE_RELAY_UNITS=S in the environment asks for S units whatever the row count (0: never)
  static const int from_env = [] { const char *e = getenv("BKCORE_RELAY_UNITS"); return e ? atoi(e) : -1; }();
  int forced = from_env;
  if (flags & BKCORE_FLAG_ROW_UNITS_MANY) forced = nch / 2;
  else if (flags & BKCORE_FLAG_ROW_UNITS) forced = 2;
  int want = forced >= 0 ? forced : 2;
  // Worth it from the first row that does not fit the GPU at once up to two rounds of rows: beyond that the last
  // round is a small part of the launch and the hand-overs cost more than it loses (measured: 769 rows -6 %, 1024
  // rows -5 %, 2000 rows +5 %, 8192 rows +3 %).
  // Nor when the last round is full or nearly so (16-bit operands run two workgroups per CU: 1024 rows are exactly
  // two rounds of 512).
  const int64_t places = (int64_t)wgs_per_cu * (cus > 0 ? cus : device_cu_count());
  const int64_t last = B % places;
  if (want < 2 || nch < 4 || (forced < 0 && (B <= places || B >= 2 * places || last == 0 || 4 * last > 3 * places)))
    return;
  if (want > nch / 2) want = nch / 2;  // at least two chunks per unit
  seg_ch = (nch + want - 1) / want;
  S = (nch + seg_ch - 1) / seg_ch;
  if (B * S > 2147483647LL) { S = 1; seg_ch = nch; }
}

// Which form the ratio-form first launch takes (state block given, no scan-path flag, nothing saved for the three-pass
// backward).  Measured on MI355X (tools/ab_default_vs_chunks.sh; P = workgroups of the chunk kernel the GPU holds at
// once, 3 per CU, 2 with the double-precision chains of long rows):
//   whole rows   one workgroup per row.  Best when the rows fill the GPU evenly: B between 0.95 P and P, or B >= 1.85 P
//                (a row's chunks cost 4-5 % more as workgroups of their own), and for rows of one or two chunks.
//   chunks       one workgroup per chunk with the look-back hand-over (mode 4).  Best whenever whole rows would leave
//                CUs unevenly or thinly occupied: B <= 0.95 P (129 rows x 32768: 0.064 -> 0.033 ms; 600 rows: 0.119 ->
//                0.102 ms; 129 rows x 131072: 0.238 -> 0.090 ms) and P < B < 1.85 P (1300 rows: 0.205 -> 0.200 ms;
//                16-bit operands, whose whole rows run two per CU: 0.236 -> 0.200 ms) -- except 16-bit operands whose
//                whole rows fill their rounds of 2 x CUs (see below).
//   units        rows in two units on a grid of one workgroup per place (mode 3): 4 % ahead of chunks between 1.24 P
//                and 1.37 P for float32 / float64 operands (1024 rows: 0.155 against 0.162 ms) and nowhere else; needs
//                its grid resident at once, so not with BKCORE_FLAG_WHOLE_ROWS.
enum { FORM_WHOLE = 0, FORM_UNITS = 1, FORM_CHUNKS = 2 };
inline int ratio_form_plan(int64_t B, int64_t N, int in_dtype, bool phi_exact, unsigned flags, bool keeps_states,
                           int cus_given, int &S, int &seg_ch) {
  const int chunk = bkcore_internal_ratio_chunk();
  const int nch = (int)((N + chunk - 1) / chunk);
  S = 1;
  seg_ch = nch;
  const bool long_rows = N > BKCORE_INTERNAL_LONG_ROW;
  const bool sixteen = in_dtype == BKCORE_BF16 || in_dtype == BKCORE_F16;
  // development: BKCORE_FORCE_CHUNKS=1 takes the chunk form wherever a row has two chunks, -1 never takes it
  static const int force_chunks = [] { const char *e = getenv("BKCORE_FORCE_CHUNKS"); return e ? atoi(e) : 0; }();
  const bool chunks_possible = nch >= 2 && B * (int64_t)nch <= 2000000000LL && force_chunks >= 0;
  if (flags & BKCORE_FLAG_ROWS_UNCUT) return FORM_WHOLE;
  if (flags & BKCORE_FLAG_ADAPTIVE)  // (its shortcut lives in the whole-row kernel)
    return (chunks_possible && B <= 128 && nch >= 8) ? FORM_CHUNKS : FORM_WHOLE;
  int per_cu = 0;  // of the unit kernel; 0: units not available
  if (!long_rows && !(flags & BKCORE_FLAG_WHOLE_ROWS)) {
    if (cus_given > 0) per_cu = sixteen ? 2 : 3;
    else per_cu = bkcore_internal_ratio_unit_occupancy(in_dtype, phi_exact);
  }
  static const int units_env = [] { const char *e = getenv("BKCORE_RELAY_UNITS"); return e ? atoi(e) : -1; }();
  if (per_cu > 0 && ((flags & (BKCORE_FLAG_ROW_UNITS | BKCORE_FLAG_ROW_UNITS_MANY)) || units_env > 0)) {
    relay_plan(B, N, chunk, per_cu, flags, S, seg_ch, cus_given);  // asked for by flag (diagnostics)
    if (S > 1) return FORM_UNITS;
  }
  if (force_chunks > 0 && chunks_possible) return FORM_CHUNKS;
  if (!chunks_possible) return FORM_WHOLE;
  const int64_t cus = cus_given > 0 ? cus_given : device_cu_count();
  const int64_t P = (int64_t)(long_rows ? 2 : 3) * cus;
  if (sixteen) {
    // whole rows of 16-bit operands run two per CU: with their last round full, or nearly, they are ahead of the chunk
    // form (bf16, N = 32768: 512 rows 0.088 against 0.091 ms, 980 rows 0.159 / 0.163, 1024 rows 0.162 / 0.169, 1536 rows
    // 0.225 / 0.237; but 560 rows 0.124 / 0.099, 1080 rows 0.193 / 0.177)
    const int64_t Pw = 2 * cus, fill = B % Pw;
    if (10 * B >= 9 * Pw && (fill == 0 || 10 * fill >= 9 * Pw)) return FORM_WHOLE;
  }
  if (nch >= 3 && 20 * B <= 19 * P) return FORM_CHUNKS;
  if (nch >= 4 && B > P && 20 * B < 37 * P) {
    if (per_cu > 0 && !sixteen && 100 * B >= 124 * P && 100 * B <= 137 * P) {
      relay_plan(B, N, chunk, per_cu, flags, S, seg_ch, cus_given);
      if (S > 1) return FORM_UNITS;
    }
    return FORM_CHUNKS;
  }
  return FORM_WHOLE;
}

template <typename TIn>
constexpr int dtype_tag() {
  return std::is_same<TIn, float>::value ? BKCORE_F32
         : std::is_same<TIn, uint16_t>::value ? BKCORE_BF16
         : std::is_same<TIn, _Float16>::value ? BKCORE_F16 : BKCORE_F64;
}

template <typename TIn, bool PHI_EXACT, bool SAVE>
int launch_forward2(const void *a, int64_t lda, const void *b, int64_t ldb, const void *c, int64_t ldc,
                    const float *z_dev, float z_re, float z_im, int64_t B, int64_t N, float *G, float *features,
                    void *ws, void *state, unsigned flags, hipStream_t stream, float *save_r, float *save_s,
                    float fclamp = __builtin_huge_valf(), float *bstates = nullptr) {
#define BK_LAUNCH(FASTV, SECOND)                                                                                     \
  hipLaunchKernelGGL((bk_forward_kernel<TIn, PHI_EXACT, FASTV, SAVE>), dim3((unsigned)B), dim3(T), 0, stream,        \
                     (const TIn *)a, (long)lda, (const TIn *)b, (long)ldb, (const TIn *)c, (long)ldc, z_dev, z_re,   \
                     z_im, (long)B, (long)N, (float2 *)G, (float2 *)features, (char *)ws, (int *)state,     \
                     SECOND, flags, fclamp, (float2 *)save_r, (float2 *)save_s)
  int S = 1, seg_ch = 0;         // scan form: segments per row
  int form = FORM_WHOLE, RS = 1, rseg = 0;  // ratio form
  if (state && !SAVE) {
    if (flags & BKCORE_FLAG_SCAN_PATH) split_plan(B, N, CHUNK, S, seg_ch);
    else form = ratio_form_plan(B, N, dtype_tag<TIn>(), PHI_EXACT, flags, bstates != nullptr, 0, RS, rseg);
  }
  if (form == FORM_CHUNKS) {
    // ratio form, one workgroup per chunk, the running product handed from chunk to chunk inside the launch; then the
    // (normally empty) launch of the scan kernel for rows whose checks failed
    const auto fwd = N > BKCORE_INTERNAL_LONG_ROW ? bkcore_internal_ratio_forward_long : bkcore_internal_ratio_forward;
    const int nch = (int)((N + bkcore_internal_ratio_chunk() - 1) / bkcore_internal_ratio_chunk());
    fwd(dtype_tag<TIn>(), PHI_EXACT, 4, 0, a, lda, b, ldb, c, ldc, z_dev, z_re, z_im, B, N, G, features, ws, state,
        flags, fclamp, nch, 1, bstates, nullptr, stream);
    hipError_t err = hipGetLastError();
    if (err == hipSuccess) {
      BK_LAUNCH(false, 3 | (nch << 2));
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
                       (long)N, (char *)ws, (int *)state, flags, S, seg_ch);
    hipLaunchKernelGGL(bk_split_combine_kernel, dim3((unsigned)B), dim3(64), 0, stream, (long)B, (long)N,
                       (char *)ws, (int *)state, S);
    hipLaunchKernelGGL((bk_split_output_kernel<TIn, PHI_EXACT>), dim3(grid), dim3(T), 0, stream, (const TIn *)a,
                       (long)lda, (const TIn *)b, (long)ldb, (const TIn *)c, (long)ldc, z_dev, z_re, z_im, (long)B,
                       (long)N, (float2 *)G, (float2 *)features, (char *)ws, (int *)state, flags, fclamp, S,
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
    const auto fwd = (N > BKCORE_INTERNAL_LONG_ROW && !SAVE) ? bkcore_internal_ratio_forward_long
                                                             : bkcore_internal_ratio_forward;
    // units: the grid must fit the GPU at once (its workgroups wait for each other): workgroups per CU as the runtime
    // reports them for this kernel on this device
    const int per_cu = form == FORM_UNITS ? bkcore_internal_ratio_unit_occupancy(dtype_tag<TIn>(), PHI_EXACT) : 0;
    if (form != FORM_UNITS || per_cu <= 0) RS = 1;
    if (RS > 1)
      fwd(dtype_tag<TIn>(), PHI_EXACT, 3, per_cu * device_cu_count(), a, lda, b, ldb, c, ldc, z_dev, z_re, z_im, B, N, G, features, ws, state,
          flags, fclamp, RS, rseg, bstates, nullptr, stream);
    else if (bstates && !SAVE)
      fwd(dtype_tag<TIn>(), PHI_EXACT, 0, 2, a, lda, b, ldb, c, ldc, z_dev, z_re, z_im, B, N, G, features, ws, state,
          flags, fclamp, 1, 0, bstates, nullptr, stream);
    else
      fwd(dtype_tag<TIn>(), PHI_EXACT, 0, SAVE, a, lda, b, ldb, c, ldc, z_dev, z_re, z_im, B, N, G, features, ws, state,
          flags, fclamp, 1, 0, save_r, save_s, stream);
    hipError_t err = hipGetLastError();
    if (err == hipSuccess) {
      if (RS > 1) BK_LAUNCH(false, 2 | (RS << 2));
      else BK_LAUNCH(false, 1);
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
                   void *ws, void *state, unsigned flags, hipStream_t s, float fclamp, float *bstates = nullptr) {
  if (flags & BKCORE_FLAG_PHI_EXACT)
    return launch_forward2<TIn, true, false>(a, lda, b, ldb, c, ldc, z_dev, z_re, z_im, B, N, G, features, ws, state,
                                             flags, s, nullptr, nullptr, fclamp, bstates);
  return launch_forward2<TIn, false, false>(a, lda, b, ldb, c, ldc, z_dev, z_re, z_im, B, N, G, features, ws, state,
                                            flags, s, nullptr, nullptr, fclamp, bstates);
}

// Whether a forward call with these arguments takes a path that writes the boundary states: float32 operands, a state
// block, the ratio form with float32 chains.
bool forward_writes_states(int in_dtype, int64_t N, const void *state, unsigned flags) {
  return in_dtype == BKCORE_F32 && state != nullptr && !(flags & (BKCORE_FLAG_SCAN_PATH | BKCORE_FLAG_ADAPTIVE)) &&
         N <= BKCORE_INTERNAL_LONG_ROW;
}

}  // namespace

extern "C" {

size_t bkcore_workspace_bytes(int64_t B, int64_t N) {
  if (B <= 0 || N <= 0) return 0;
  int S = 1, S2 = 1, seg_ch = 0;  // whichever first launch runs: room for the larger number of segment records
  split_plan(B, N, bkcore_internal_ratio_chunk(), S, seg_ch);
  split_plan(B, N, CHUNK, S2, seg_ch);
  if (S2 > S) S = S2;
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

size_t bkcore_boundary_state_bytes(int64_t B, int64_t N) {
  if (B <= 0 || N <= 0) return 0;
  // per row two planes of one float2 per block of 8 elements of whole chunks of 4096, plus one block
  return (size_t)B * 2 * (size_t)(((N + 4095) / 4096) * 512 + 1) * 2 * sizeof(float);
}

int bkcore_forward_states(const void *a, int64_t lda, const void *b, int64_t ldb, const void *c, int64_t ldc, int in_dtype,
                          const float *z_dev, float z_re, float z_im, int64_t B, int64_t N, float *G, float *features,
                          float feature_clamp, float *boundary_states, size_t boundary_state_bytes,
                          int *states_written, void *workspace, size_t workspace_bytes, void *state, unsigned flags,
                          void *stream) {
  if (states_written) *states_written = 0;
  if (!boundary_states)
    return bkcore_forward_clamped(a, lda, b, ldb, c, ldc, in_dtype, z_dev, z_re, z_im, B, N, G, features, feature_clamp,
                                  workspace, workspace_bytes, state, flags, stream);
  const float fclamp = feature_clamp > 0.0f ? feature_clamp : __builtin_huge_valf();
  if (feature_clamp != feature_clamp) return BKCORE_EINVAL;
  if (B < 0 || N < 0 || lda < 0 || ldb < 0 || ldc < 0) return BKCORE_EINVAL;
  if (B == 0 || N == 0) return BKCORE_OK;
  if (!a || !G || (N > 1 && (!b || !c))) return BKCORE_EINVAL;
  if (N > bkcore_max_seq_len()) return BKCORE_ETOOLONG;
  if (B > 2147483647LL) return BKCORE_EINVAL;
  const size_t need = bkcore_workspace_bytes(B, N);
  if (need > 0 && (!workspace || workspace_bytes < need)) return BKCORE_EWORKSPACE;
  if (boundary_state_bytes < bkcore_boundary_state_bytes(B, N)) return BKCORE_EWORKSPACE;
  if (!forward_writes_states(in_dtype, N, state, flags))
    return bkcore_forward_clamped(a, lda, b, ldb, c, ldc, in_dtype, z_dev, z_re, z_im, B, N, G, features, feature_clamp,
                                  workspace, workspace_bytes, state, flags, stream);
  if (states_written) *states_written = 1;
  return launch_forward<float>(a, lda, b, ldb, c, ldc, z_dev, z_re, z_im, B, N, G, features, workspace, state, flags,
                               (hipStream_t)stream, fclamp, boundary_states);
}

int bkcore_forward_noalloc(const void *a, int64_t lda, const void *b, int64_t ldb, const void *c, int64_t ldc,
                           int in_dtype, const float *z_dev, float z_re, float z_im, int64_t B, int64_t N, float *G,
                           float *features, float feature_clamp, unsigned flags, void *stream) {
  if (B < 0 || N < 0) return BKCORE_EINVAL;
  if (B == 0 || N == 0) return BKCORE_OK;
  if (N > bkcore_max_seq_len()) return BKCORE_ETOOLONG;
  const size_t need = bkcore_workspace_bytes(B, N);
  if (need > STATIC_WS_BYTES) return BKCORE_ETOOBIG;
  int dev = -1;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess || dev < 0 || dev >= MAX_DEVICES) {
    g_last_hip_error = (int)e;
    return e != hipSuccess ? BKCORE_ELAUNCH : BKCORE_EINVAL;
  }
  // each device holds its own copy of the array at its own address: looked up once per device, with that device
  // current, and kept; the array's name itself is never passed to a kernel
  unsigned char *base = __atomic_load_n(&g_static_scratch_ptr[dev], __ATOMIC_ACQUIRE);
  if (!base) {
    void *p = nullptr;
    e = hipGetSymbolAddress(&p, HIP_SYMBOL(bk_static_scratch));
    if (e != hipSuccess || !p) {
      g_last_hip_error = (int)e;
      return BKCORE_ELAUNCH;
    }
    base = static_cast<unsigned char *>(p);
    __atomic_store_n(&g_static_scratch_ptr[dev], base, __ATOMIC_RELEASE);
  }
  // [0, STATIC_STATE_BYTES): the state block (zero when the library is loaded, self-clearing afterwards);
  // behind it the workspace, which every call writes before it reads
  return bkcore_forward_clamped(a, lda, b, ldb, c, ldc, in_dtype, z_dev, z_re, z_im, B, N, G, features, feature_clamp,
                                base + STATIC_STATE_BYTES, need, base, flags, stream);
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

int bkcore_internal_forward_save_rows(const float *a, int64_t lda, const float *b, int64_t ldb, const float *c,
                                      int64_t ldc, const float *z_dev, float z_re, float z_im, int64_t B, int64_t N,
                                      float *G, void *workspace, unsigned flags, void *stream, float *save_r,
                                      float *save_s, const int *row_filter, const int *gate) {
  // one launch of the two-scan kernel (it needs no verification and no state block), saving variant
  hipStream_t s = (hipStream_t)stream;
#define BK_SAVE_ROWS(EXACT)                                                                                          \
  hipLaunchKernelGGL((bk_forward_kernel<float, EXACT, false, true>), dim3((unsigned)B), dim3(T), 0, s, a, (long)lda, \
                     b, (long)ldb, c, (long)ldc, z_dev, z_re, z_im, (long)B, (long)N, (float2 *)G, (float2 *)nullptr, \
                     (char *)workspace, (int *)nullptr, 0, flags, __builtin_huge_valf(), (float2 *)save_r,           \
                     (float2 *)save_s, row_filter, gate)
  if (flags & BKCORE_FLAG_PHI_EXACT) BK_SAVE_ROWS(true);
  else BK_SAVE_ROWS(false);
#undef BK_SAVE_ROWS
  const hipError_t err = hipGetLastError();
  if (err != hipSuccess) {
    g_last_hip_error = (int)err;
    return BKCORE_ELAUNCH;
  }
  return BKCORE_OK;
}

int bkcore_backward_ref(const float *G, const float *grad_features, const float *grad_G, float alpha, int64_t n,
                        float *grad_a, void *stream) {
  return bkcore_backward_ref_flagged(G, grad_features, grad_G, alpha, n, grad_a, nullptr, stream);
}

int bkcore_backward_ref_flagged(const float *G, const float *grad_features, const float *grad_G, float alpha, int64_t n,
                                float *grad_a, int *nonfinite_flag, void *stream) {
  if (n < 0) return BKCORE_EINVAL;
  if (n == 0) return BKCORE_OK;
  if (!G || !grad_a) return BKCORE_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const long quads = (n + 3) / 4;
  long blocks = (quads + 255) / 256;
  if (blocks > (1L << 20)) blocks = 1L << 20;  // one quad per thread (measured faster than a grid-stride loop over fewer blocks)
  if (blocks < 1) blocks = 1;
  const dim3 grid((unsigned)blocks), block(256);
  const float2 *Gp = (const float2 *)G, *gf = (const float2 *)grad_features, *gg = (const float2 *)grad_G;
  if (gf && gg)
    hipLaunchKernelGGL((bk_backward_ref_kernel<true, true>), grid, block, 0, s, Gp, gf, gg, alpha, (long)n, grad_a,
                       nonfinite_flag);
  else if (gf)
    hipLaunchKernelGGL((bk_backward_ref_kernel<true, false>), grid, block, 0, s, Gp, gf, gg, alpha, (long)n, grad_a,
                       nonfinite_flag);
  else if (gg)
    hipLaunchKernelGGL((bk_backward_ref_kernel<false, true>), grid, block, 0, s, Gp, gf, gg, alpha, (long)n, grad_a,
                       nonfinite_flag);
  else
    hipLaunchKernelGGL((bk_backward_ref_kernel<false, false>), grid, block, 0, s, Gp, gf, gg, alpha, (long)n, grad_a,
                       nonfinite_flag);
  const hipError_t err = hipGetLastError();
  if (err != hipSuccess) {
    g_last_hip_error = (int)err;
    return BKCORE_ELAUNCH;
  }
  return BKCORE_OK;
}

int bkcore_backward_variant(const float *G, const float *grad_features, const float *grad_G, float alpha, int64_t n,
                            float *grad_a, int variant, void *stream) {
  if (variant == BKCORE_BWD_REFERENCE) return bkcore_backward_ref(G, grad_features, grad_G, alpha, n, grad_a, stream);
  if (variant != BKCORE_BWD_BATCHED && variant != BKCORE_BWD_SCAN_STABLE) return BKCORE_EINVAL;
  if (n < 0) return BKCORE_EINVAL;
  if (n == 0) return BKCORE_OK;
  if (!G || !grad_a) return BKCORE_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const long quads = (n + 3) / 4;
  long blocks = (quads + 255) / 256;
  if (blocks > (1L << 20)) blocks = 1L << 20;  // one quad per thread (measured faster than a grid-stride loop over fewer blocks)
  const dim3 grid((unsigned)blocks), block(256);
  const float2 *Gp = (const float2 *)G, *gf = (const float2 *)grad_features, *gg = (const float2 *)grad_G;
#define BK_BV(GF, GG, V)                                                                                             \
  hipLaunchKernelGGL((bk_backward_ref_kernel<GF, GG, V>), grid, block, 0, s, Gp, gf, gg, alpha, (long)n, grad_a,    \
                     (int *)nullptr)
#define BK_BV_ALL(V)                                                                                                 \
  do {                                                                                                               \
    if (gf && gg) BK_BV(true, true, V);                                                                              \
    else if (gf) BK_BV(true, false, V);                                                                              \
    else if (gg) BK_BV(false, true, V);                                                                              \
    else BK_BV(false, false, V);                                                                                     \
  } while (0)
  if (variant == BKCORE_BWD_BATCHED) BK_BV_ALL(1);
  else BK_BV_ALL(2);
#undef BK_BV_ALL
#undef BK_BV
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
  const bool vec = (N % 4) == 0 && (ldv % 4) == 0 && ((uintptr_t)v_pre % 16) == 0 && ((uintptr_t)h0_diag % 16) == 0 &&
                   ((uintptr_t)he_diag % 16) == 0;
  long gx = vec ? (N + 4095) / 4096 : (N + 1023) / 1024;  // vectorised: 4 quads per thread and pass
  if (gx > 1024) gx = 1024;
  const dim3 grid((unsigned)gx, (unsigned)(B > 32768 ? 32768 : B)), block(256);
  const float inv = 1.0f / v_max;
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
  long blocks = ((n + 3) / 4 + 255) / 256;     // four elements per thread, one pass
  if (blocks > (1L << 20)) blocks = 1L << 20;
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

int bkcore_forward_plan(int64_t B, int64_t N, int in_dtype, unsigned flags, int cus, int *launches) {
  if (launches) *launches = 0;
  if (B <= 0 || N <= 0 || N > (int64_t)MAX_N || in_dtype < BKCORE_F32 || in_dtype > BKCORE_F64) return BKCORE_EINVAL;
  int S = 1, seg_ch = 0;
  if (flags & BKCORE_FLAG_SCAN_PATH) {
    split_plan(B, N, CHUNK, S, seg_ch);
    if (launches) *launches = S > 1 ? 4 : 2;
    return S;
  }
  if (launches) *launches = 2;
  if (cus <= 0) {  // as the dispatch: what the runtime reports on the current device; whole rows without one
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
      (void)hipGetLastError();
      return 1;
    }
  }
  const int form = ratio_form_plan(B, N, in_dtype, (flags & BKCORE_FLAG_PHI_EXACT) != 0, flags, false, cus, S, seg_ch);
  if (form == FORM_CHUNKS) return (int)((N + bkcore_internal_ratio_chunk() - 1) / bkcore_internal_ratio_chunk());
  return form == FORM_UNITS ? S : 1;
}
int bkcore_last_hip_error(void) { return g_last_hip_error; }
void bkcore_internal_set_hip_error(int code) { g_last_hip_error = code; }
int bkcore_abi_version(void) { return 18; }

const char *bkcore_error_string(int code) {
  switch (code) {
    case BKCORE_OK: return "ok";
    case BKCORE_EINVAL: return "invalid argument";
    case BKCORE_ETOOLONG: return "sequence length exceeds bkcore_max_seq_len()";
    case BKCORE_ELAUNCH: return "HIP kernel launch failed (see bkcore_last_hip_error())";
    case BKCORE_EWORKSPACE: return "workspace missing or smaller than bkcore_workspace_bytes()";
    case BKCORE_ETOOBIG: return "problem needs more scratch than bkcore_forward_noalloc holds (BKCORE_NOALLOC_MAX_WORKSPACE)";
    default: return "unknown bkcore error";
  }
}

}  // extern "C"

#else  // BK_COMPLEX_BUILD: the scan kernel above with a complex k, and its one entry point

}  // namespace

extern "C" {

// Lane states of the complex kernel (its chunks are half as long as the main build's).
size_t bkcore_get_diag_complex_workspace_bytes(int64_t B, int64_t N) {
  if (B <= 0 || N <= 0) return 0;
  return rowflag_bytes(B) + lane_state_bytes(B, N);
}

int bkcore_get_diag_complex(const float *a, int64_t lda, const float *b, int64_t ldb, const float *c, int64_t ldc,
                            const float *z_dev, float z_re, float z_im, int64_t B, int64_t N, float *G, void *workspace,
                            size_t workspace_bytes, unsigned flags, void *stream) {
  if (B < 0 || N < 0 || lda < 0 || ldb < 0 || ldc < 0) return BKCORE_EINVAL;
  if (flags & ~(unsigned)BKCORE_FLAG_PHI_EXACT) return BKCORE_EINVAL;
  if (B == 0 || N == 0) return BKCORE_OK;
  if (!a || !G || (N > 1 && (!b || !c))) return BKCORE_EINVAL;
  if (N > MAX_N) return BKCORE_ETOOLONG;
  if (B > 2147483647LL) return BKCORE_EINVAL;
  const size_t need = bkcore_get_diag_complex_workspace_bytes(B, N);
  if (!workspace || workspace_bytes < need) return BKCORE_EWORKSPACE;
  const unsigned kflags = flags | BKCORE_FLAG_RAW;
  const hipStream_t s = (hipStream_t)stream;
  // one launch of the two-scan form (no assumption about the input, nothing to verify afterwards)
#define BK_LAUNCH_C(PHI)                                                                                              \
  hipLaunchKernelGGL((bk_forward_kernel<float2, PHI, false, false>), dim3((unsigned)B), dim3(T), 0, s,                \
                     (const float2 *)a, (long)lda, (const float2 *)b, (long)ldb, (const float2 *)c, (long)ldc, z_dev, \
                     z_re, z_im, (long)B, (long)N, (float2 *)G, (float2 *)nullptr, (char *)workspace, (int *)nullptr, \
                     0, kflags, __builtin_huge_valf(), (float2 *)nullptr, (float2 *)nullptr)
  if (flags & BKCORE_FLAG_PHI_EXACT) BK_LAUNCH_C(true);
  else BK_LAUNCH_C(false);
#undef BK_LAUNCH_C
  const hipError_t err = hipGetLastError();
  if (err != hipSuccess) {
    bkcore_internal_set_hip_error((int)err);
    return BKCORE_ELAUNCH;
  }
  return BKCORE_OK;
}

}  // extern "C"
#endif  // BK_COMPLEX_BUILD

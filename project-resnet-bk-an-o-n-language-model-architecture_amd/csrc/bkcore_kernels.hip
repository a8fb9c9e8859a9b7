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
// non-linear.  It is reproduced by a two-pass fixed point: pass 1 scans the linear recurrence and
// yields d_p (only the ratio of consecutive terms matters), pass 2 scans with kappa_p * w_p.
//
// Kernel shape (one workgroup of 256 threads per row, chunks of 256 x E elements):
//   sweep A, chunks left -> right : theta scan, keeps only the chunk-boundary carries (LDS)
//   sweep B, chunks right -> left : phi scan with a running carry, theta re-scanned inside the
//                                   chunk from the saved carry, epilogue, stores of G / features.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/bkcore.h"

namespace {

constexpr int T = 256;             // threads per workgroup
constexpr int E = 8;               // elements per thread per chunk
constexpr int CHUNK = T * E;       // elements per chunk
constexpr int WAVES = T / 64;
constexpr int MAX_CHUNKS = 512;    // chunk-boundary carries kept in LDS
constexpr float LN2_HALF = 0.34657359027997264f;  // 0.5 * ln 2
constexpr float LOG2E = 1.4426950408889634f;

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
__device__ __forceinline__ float amax2(float2 p) { return fmaxf(fabsf(p.x), fabsf(p.y)); }
__device__ __forceinline__ float2 ld2(float2 p, int e) { return make_float2(ldexpf(p.x, e), ldexpf(p.y, e)); }
// exponent e with m * 2^-e in [0.5,1); 0 for m == 0, inf, nan
__device__ __forceinline__ int expo(float m) { return __builtin_amdgcn_frexp_expf(m); }

__device__ __forceinline__ void vnorm(Vec &v) {
  const int de = expo(fmaxf(amax2(v.x0), amax2(v.x1)));
  v.x0 = ld2(v.x0, -de);
  v.x1 = ld2(v.x1, -de);
  v.e += de;
}
__device__ __forceinline__ void mnorm(Mat &m) {
  const int de = expo(fmaxf(fmaxf(amax2(m.a), amax2(m.b)), fmaxf(amax2(m.c), amax2(m.d))));
  m.a = ld2(m.a, -de);
  m.b = ld2(m.b, -de);
  m.c = ld2(m.c, -de);
  m.d = ld2(m.d, -de);
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
__device__ __forceinline__ Mat mat_shfl(const Mat &m, int src) {
  Mat r;
  r.a.x = __shfl(m.a.x, src);
  r.a.y = __shfl(m.a.y, src);
  r.b.x = __shfl(m.b.x, src);
  r.b.y = __shfl(m.b.y, src);
  r.c.x = __shfl(m.c.x, src);
  r.c.y = __shfl(m.c.y, src);
  r.d.x = __shfl(m.d.x, src);
  r.d.y = __shfl(m.d.y, src);
  r.e = __shfl(m.e, src);
  return r;
}

// exp(88 tanh(d/88) - d): the reference's soft clamp exp(88 tanh(d/88)) (bk_core.py:60-75) relative to exp(d).
// |d| <= 8: tanh series (next term < 1e-9); otherwise evaluated directly.  Non-finite d (a zero term) gives 1.
__device__ __forceinline__ float wall_corr(float d) {
  const float u = d * (1.0f / 88.0f);
  const float s = u * u;
  float g = -88.0f * u * s * fmaf(-s, fmaf(-s, 17.0f / 315.0f, 2.0f / 15.0f), 1.0f / 3.0f);
  if (__builtin_expect(fabsf(d) > 8.0f, 0)) g = fmaf(88.0f, tanhf(u), -d);
  float w = __builtin_amdgcn_exp2f(g * LOG2E);
  if (!(fabsf(d) <= 3.0e38f)) w = 1.0f;
  return w;
}

struct Shared {
  float a_s[CHUNK + CHUNK / 32 + 8];      // staged diagonal (padded: idx + idx/32)
  float k_s[CHUNK + 1 + CHUNK / 32 + 8];  // staged k_{i-1} = b_{i-1} c_{i-1} for i = chunk_base .. chunk_base+CHUNK
  Mat tot[2][WAVES];                      // per-wave aggregates, double-buffered across scans
  Vec carry1[MAX_CHUNKS];                 // theta carries entering each chunk, pass 1 (linear)
  Vec carry2[MAX_CHUNKS];                 // theta carries entering each chunk, pass 2 (wall-corrected)
};

__device__ __forceinline__ int pad(int idx) { return idx + (idx >> 5); }

// ---- lane-level serial pieces -------------------------------------------------------------------------------

// Product of the lane's E transfer matrices in processing order (REV: element E-1 first).
template <bool REV, bool TAIL>
__device__ __forceinline__ Mat lane_aggregate(const float2 (&al)[E], const float (&ka)[E], unsigned vmask) {
  Mat A = mat_identity();
#pragma unroll
  for (int j = 0; j < E; ++j) {
    const int ei = REV ? E - 1 - j : j;
    if (!TAIL || ((vmask >> ei) & 1u)) {
      const float2 na = rstep(al[ei], ka[ei], A.a, A.c);
      const float2 nb = rstep(al[ei], ka[ei], A.b, A.d);
      A.c = A.a;
      A.d = A.b;
      A.a = na;
      A.b = nb;
    }
    if ((j & 1) == 1) mnorm(A);
  }
  return A;
}

// Pass 1 re-run from the lane's incoming state: wall factor per element -> k2 = ka * w.
template <bool REV, bool TAIL>
__device__ __forceinline__ void rerun_wall(const float2 (&al)[E], const float (&ka)[E], unsigned vmask, Vec x,
                                           float (&k2)[E]) {
  float l0 = __builtin_amdgcn_logf(fmaf(x.x0.x, x.x0.x, x.x0.y * x.x0.y));  // log2 |x0|^2
  float l1 = __builtin_amdgcn_logf(fmaf(x.x1.x, x.x1.x, x.x1.y * x.x1.y));
#pragma unroll
  for (int j = 0; j < E; ++j) {
    const int ei = REV ? E - 1 - j : j;
    const float d = LN2_HALF * (l1 - l0);  // ln|prev2| - ln|prev|
    k2[ei] = ka[ei] * wall_corr(d);
    if (!TAIL || ((vmask >> ei) & 1u)) {
      const float2 y = rstep(al[ei], ka[ei], x.x0, x.x1);
      x.x1 = x.x0;
      x.x0 = y;
      l1 = l0;
      if ((j & 1) == 1) {
        const int de = expo(fmaxf(amax2(x.x0), amax2(x.x1)));
        x.x0 = ld2(x.x0, -de);
        x.x1 = ld2(x.x1, -de);
        l1 -= 2.0f * (float)de;
      }
      l0 = __builtin_amdgcn_logf(fmaf(x.x0.x, x.x0.x, x.x0.y * x.x0.y));
    }
  }
}

// Pass 2 re-run: incoming first component (mantissa, exponent) per element.
template <bool REV, bool TAIL>
__device__ __forceinline__ void rerun_out(const float2 (&al)[E], const float (&k2)[E], unsigned vmask, Vec x,
                                          float2 (&om)[E], int (&oe)[E]) {
#pragma unroll
  for (int j = 0; j < E; ++j) {
    const int ei = REV ? E - 1 - j : j;
    om[ei] = x.x0;
    oe[ei] = x.e;
    if (!TAIL || ((vmask >> ei) & 1u)) {
      const float2 y = rstep(al[ei], k2[ei], x.x0, x.x1);
      x.x1 = x.x0;
      x.x0 = y;
      if ((j & 1) == 1) vnorm(x);
    }
  }
}

// ---- workgroup-level exclusive scan of lane aggregates ------------------------------------------------------
// Returns this lane's incoming state (product of all earlier lanes' aggregates applied to `carry`) and advances
// `carry` across the whole chunk.  "Earlier" follows the processing order: ascending threads, or descending if REV.
template <bool REV>
__device__ __forceinline__ Vec block_exclusive(Mat A, Vec &carry, Shared &sh, int &scan_count) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  // inclusive Kogge-Stone scan inside the wave
#pragma unroll
  for (int s = 1; s < 64; s <<= 1) {
    const int src = REV ? lane + s : lane - s;
    const Mat o = mat_shfl(A, src & 63);
    const Mat cmb = matmul(A, o);
    const bool take = REV ? (lane + s < 64) : (lane >= s);
    if (take) A = cmb;
  }
  const int buf = scan_count & 1;
  scan_count++;
  if (lane == (REV ? 0 : 63)) sh.tot[buf][wave] = A;
  // exclusive prefix inside the wave
  Mat ex = mat_shfl(A, (REV ? lane + 1 : lane - 1) & 63);
  if (lane == (REV ? 63 : 0)) ex = mat_identity();
  __syncthreads();
  Vec v = carry;
  Vec xin = carry;
#pragma unroll
  for (int w = 0; w < WAVES; ++w) {
    const int wi = REV ? WAVES - 1 - w : w;
    if (wi == wave) xin = v;
    v = matvec(sh.tot[buf][wi], v);
  }
  carry = v;
  return matvec(ex, xin);
}

// Two-pass scan of one chunk in one direction.
template <bool REV, bool TAIL, bool WANT_OUT>
__device__ __forceinline__ void scan_dir(const float2 (&al)[E], const float (&ka)[E], unsigned vmask, Vec &c1, Vec &c2,
                                         float2 (&om)[E], int (&oe)[E], Shared &sh, int &scan_count) {
  Mat A = lane_aggregate<REV, TAIL>(al, ka, vmask);
  Vec x = block_exclusive<REV>(A, c1, sh, scan_count);
  float k2[E];
  rerun_wall<REV, TAIL>(al, ka, vmask, x, k2);
  A = lane_aggregate<REV, TAIL>(al, k2, vmask);
  x = block_exclusive<REV>(A, c2, sh, scan_count);
  if (WANT_OUT) rerun_out<REV, TAIL>(al, k2, vmask, x, om, oe);
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
  bool raw;        // no clamp / sanitise
  bool phi_exact;
  bool fix_bc;     // replace NaN in b/c by 1 (second launch when he_diag held a NaN)
};

// Stages chunk `ch` of one row into LDS: a_s[idx] = a[base+idx], k_s[idx] = b[g] c[g] with g = base-1+idx.
// Returns bit0: NaN seen in a, bit1: NaN seen in b or c.
template <typename TIn>
__device__ __forceinline__ unsigned stage_chunk(const TIn *__restrict__ a, const TIn *__restrict__ b,
                                                const TIn *__restrict__ c, long base, const RowCfg &cfg, Shared &sh) {
  unsigned seen = 0;
  const int t = threadIdx.x;
#pragma unroll
  for (int j = 0; j < E; ++j) {
    const int idx = t + j * T;
    const long gi = base + idx;
    float v = 1.0f;
    if (gi < cfg.N) {
      v = to_f32<TIn>(a[gi]);
      if (!cfg.raw) {
        v = clamp_keep_nan(v, 100.0f);
        if (v != v) { v = 0.0f; seen |= 1u; }
      }
    }
    sh.a_s[pad(idx)] = v;
  }
#pragma unroll
  for (int j = 0; j <= E; ++j) {
    const int idx = t + j * T;
    if (j < E || t == 0) {
      const long g = base - 1 + idx;
      float k = 0.0f;
      if (g >= 0 && g < cfg.N - 1) {
        float bv = to_f32<TIn>(b[g]);
        float cv = to_f32<TIn>(c[g]);
        if (!cfg.raw) {
          bv = clamp_keep_nan(bv, 10.0f);
          cv = clamp_keep_nan(cv, 10.0f);
          if (bv != bv) { seen |= 2u; if (cfg.fix_bc) bv = 1.0f; }
          if (cv != cv) { seen |= 2u; if (cfg.fix_bc) cv = 1.0f; }
        }
        k = bv * cv;
      }
      sh.k_s[pad(idx)] = k;
    }
  }
  return seen;
}

// Reads the thread's E elements back in blocked order and forms alpha = reg(a - z) (bk_core.py:78-94),
// kth[e] = k_{i-1} (theta, and phi under the reference convention), kph[e] (phi).
__device__ __forceinline__ void fetch_elems(const Shared &sh, long base, const RowCfg &cfg, float2 (&al)[E],
                                            float (&kth)[E], float (&kph)[E], unsigned &vmask) {
  const int t = threadIdx.x;
  vmask = 0;
  float kw[E + 1];
#pragma unroll
  for (int e = 0; e <= E; ++e) kw[e] = sh.k_s[pad(t * E + e)];
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const float av = sh.a_s[pad(t * E + e)];
    const long gi = base + t * E + e;
    if (gi < cfg.N) vmask |= (1u << e);
    const float xr = av - cfg.z.x;
    const float xi = -cfg.z.y;
    const float n = fmaf(xr, xr, xi * xi);
    float2 alv;
    if (n >= 1e-24f) {
      const float f = fmaf(1e-3f, __builtin_amdgcn_rsqf(n), 1.0f);
      alv = make_float2(xr * f, xi * f);
    } else {
      alv = make_float2(xr + 1e-3f, xi);
    }
    al[e] = alv;
    kth[e] = kw[e];
    // phi step that produces phi_{i-1} from (phi_i, phi_{i+1}) uses matrix index i: k_{i-1} in the reference's
    // convention (bk_core.py:171), k_i in the exact one.  At i = N-1 the second term is phi_N = 0.
    kph[e] = cfg.phi_exact ? ((gi < cfg.N - 1) ? kw[e + 1] : 0.0f) : kw[e];
  }
}

// ---- the forward kernel -------------------------------------------------------------------------------------
template <typename TIn>
__global__ __launch_bounds__(T) void bk_forward_kernel(const TIn *__restrict__ a_, long lda, const TIn *__restrict__ b_,
                                                       long ldb, const TIn *__restrict__ c_, long ldc,
                                                       const float *__restrict__ z_dev, float z_re, float z_im, long N,
                                                       float2 *__restrict__ G_, float2 *__restrict__ F_, int *state,
                                                       int call_id, int fixup, unsigned flags) {
  __shared__ Shared sh;
  if (fixup) {
    // second launch: only when this call saw a NaN in he_diag AND a NaN in b/c
    const int sa = __hip_atomic_load(&state[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int sb = __hip_atomic_load(&state[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (sa != call_id || sb != call_id) return;
  }
  const long row = blockIdx.x;
  const TIn *a = a_ + row * lda;
  const TIn *b = b_ + row * ldb;
  const TIn *c = c_ + row * ldc;
  float2 *G = G_ + row * N;
  float2 *F = F_ ? F_ + row * N : nullptr;

  RowCfg cfg;
  cfg.z = z_dev ? make_float2(z_dev[0], z_dev[1]) : make_float2(z_re, z_im);
  cfg.N = N;
  cfg.raw = (flags & BKCORE_FLAG_RAW) != 0;
  cfg.phi_exact = (flags & BKCORE_FLAG_PHI_EXACT) != 0;
  cfg.fix_bc = fixup != 0;

  const int nch = (int)((N + CHUNK - 1) / CHUNK);
  const int t = threadIdx.x;
  int scan_count = 0;
  unsigned seen = 0;

  float2 al[E];
  float kth[E], kph[E];
  float2 thm[E], phm[E];
  int the[E], phe[E];
  unsigned vmask;

  // ---- sweep A: theta carries at chunk boundaries
  Vec c1 = vec_start(), c2 = vec_start();
  for (int ch = 0; ch < nch; ++ch) {
    const long base = (long)ch * CHUNK;
    if (t == 0) {
      sh.carry1[ch] = c1;
      sh.carry2[ch] = c2;
    }
    __syncthreads();  // previous chunk's LDS reads done
    seen |= stage_chunk<TIn>(a, b, c, base, cfg, sh);
    __syncthreads();
    fetch_elems(sh, base, cfg, al, kth, kph, vmask);
    if (base + CHUNK <= N)
      scan_dir<false, false, false>(al, kth, vmask, c1, c2, thm, the, sh, scan_count);
    else
      scan_dir<false, true, false>(al, kth, vmask, c1, c2, thm, the, sh, scan_count);
  }
  // theta_N = first component of the state after the last element
  const float2 dm = c2.x0;
  const int e_det = c2.e;
  const float dn = fmaf(dm.x, dm.x, dm.y * dm.y);
  const float2 inv_det = make_float2(dm.x / dn, -dm.y / dn);

  // ---- sweep B: phi with running carry, theta from saved carries, epilogue
  Vec p1 = vec_start(), p2 = vec_start();
  for (int ch = nch - 1; ch >= 0; --ch) {
    const long base = (long)ch * CHUNK;
    __syncthreads();
    stage_chunk<TIn>(a, b, c, base, cfg, sh);
    __syncthreads();
    fetch_elems(sh, base, cfg, al, kth, kph, vmask);
    Vec t1 = sh.carry1[ch];
    Vec t2 = sh.carry2[ch];
    if (base + CHUNK <= N) {
      scan_dir<false, false, true>(al, kth, vmask, t1, t2, thm, the, sh, scan_count);
      scan_dir<true, false, true>(al, kph, vmask, p1, p2, phm, phe, sh, scan_count);
    } else {
      scan_dir<false, true, true>(al, kth, vmask, t1, t2, thm, the, sh, scan_count);
      scan_dir<true, true, true>(al, kph, vmask, p1, p2, phm, phe, sh, scan_count);
    }
    // epilogue (bk_core.py:190-208)
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const long gi = base + (long)t * E + e;
      const float2 P = cmul(cmul(thm[e], phm[e]), inv_det);
      const int Ee = the[e] + phe[e] - e_det;
      const float n = fmaf(P.x, P.x, P.y * P.y);
      const float L = LN2_HALF * fmaf(2.0f, (float)Ee, __builtin_amdgcn_logf(n));  // ln |G_raw|
      float2 g;
      if (__builtin_expect(fabsf(L) < 8.0f, 1)) {
        const float sc = ldexpf(wall_corr(L), Ee);
        g = make_float2(P.x * sc, P.y * sc);
      } else {
        const float mag = expf(88.0f * tanhf(L * (1.0f / 88.0f))) * __builtin_amdgcn_rsqf(n);
        g = make_float2(P.x * mag, P.y * mag);
      }
      if (!(fabsf(g.x) <= 3.0e38f) || !(fabsf(g.y) <= 3.0e38f)) g = make_float2(0.f, 0.f);
      const float m2 = fmaf(g.x, g.x, g.y * g.y);
      if (__builtin_expect(m2 > 2500.0f, 0)) {
        const float f = 50.0f / (sqrtf(m2) + 1e-9f);
        g = make_float2(g.x * f, g.y * f);
      }
      if (gi < N) {
        G[gi] = g;
        if (F) F[gi] = g;
      }
    }
  }
  if (state && !fixup && seen) {
    if (seen & 1u) __hip_atomic_store(&state[0], call_id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (seen & 2u) __hip_atomic_store(&state[1], call_id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
  // g already holds grad wrt (Re G, Im G); conj(g) = (g.x, -g.y)
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
    reinterpret_cast<float2 *>(out)[p] = o;
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
    const long i = n - 1;
    float2 g = make_float2(0.f, 0.f);
    if (HAS_GF) g = make_float2(nan_to_num_default(gf[i].x), nan_to_num_default(gf[i].y));
    if (HAS_GG) { g.x += gG[i].x; g.y += gG[i].y; }
    out[i] = bwd_ref_one(G[i], g, alpha);
  }
}

thread_local int g_last_hip_error = 0;
int g_call_counter = 0;

template <typename TIn>
int launch_forward(const void *a, int64_t lda, const void *b, int64_t ldb, const void *c, int64_t ldc,
                   const float *z_dev, float z_re, float z_im, int64_t B, int64_t N, float *G, float *features,
                   void *state, unsigned flags, hipStream_t stream) {
  int call_id = 0;
  if (state) {
    call_id = __atomic_add_fetch(&g_call_counter, 1, __ATOMIC_RELAXED);
    if (call_id <= 0) {  // wrapped: restart above zero
      __atomic_store_n(&g_call_counter, 1, __ATOMIC_RELAXED);
      call_id = 1;
    }
  }
  const int launches = (state && !(flags & BKCORE_FLAG_RAW)) ? 2 : 1;
  for (int fixup = 0; fixup < launches; ++fixup) {
    hipLaunchKernelGGL(bk_forward_kernel<TIn>, dim3((unsigned)B), dim3(T), 0, stream, (const TIn *)a, (long)lda,
                       (const TIn *)b, (long)ldb, (const TIn *)c, (long)ldc, z_dev, z_re, z_im, (long)N, (float2 *)G,
                       (float2 *)features, (int *)state, call_id, fixup, flags);
    const hipError_t err = hipGetLastError();
    if (err != hipSuccess) {
      g_last_hip_error = (int)err;
      return BKCORE_ELAUNCH;
    }
  }
  return BKCORE_OK;
}

}  // namespace

extern "C" {

int bkcore_forward(const void *a, int64_t lda, const void *b, int64_t ldb, const void *c, int64_t ldc, int in_dtype,
                   const float *z_dev, float z_re, float z_im, int64_t B, int64_t N, float *G, float *features,
                   void *state, unsigned flags, void *stream) {
  if (B < 0 || N < 0 || lda < 0 || ldb < 0 || ldc < 0) return BKCORE_EINVAL;
  if (B == 0 || N == 0) return BKCORE_OK;
  if (!a || !G || (N > 1 && (!b || !c))) return BKCORE_EINVAL;
  if (N > bkcore_max_seq_len()) return BKCORE_ETOOLONG;
  if (B > 2147483647LL) return BKCORE_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  switch (in_dtype) {
    case BKCORE_F32:
      return launch_forward<float>(a, lda, b, ldb, c, ldc, z_dev, z_re, z_im, B, N, G, features, state, flags, s);
    case BKCORE_BF16:
      return launch_forward<uint16_t>(a, lda, b, ldb, c, ldc, z_dev, z_re, z_im, B, N, G, features, state, flags, s);
    case BKCORE_F16:
      return launch_forward<_Float16>(a, lda, b, ldb, c, ldc, z_dev, z_re, z_im, B, N, G, features, state, flags, s);
    case BKCORE_F64:
      return launch_forward<double>(a, lda, b, ldb, c, ldc, z_dev, z_re, z_im, B, N, G, features, state, flags, s);
    default:
      return BKCORE_EINVAL;
  }
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

int64_t bkcore_max_seq_len(void) { return (int64_t)MAX_CHUNKS * CHUNK; }
int bkcore_last_hip_error(void) { return g_last_hip_error; }
int bkcore_abi_version(void) { return 1; }

const char *bkcore_error_string(int code) {
  switch (code) {
    case BKCORE_OK: return "ok";
    case BKCORE_EINVAL: return "invalid argument";
    case BKCORE_ETOOLONG: return "sequence length exceeds bkcore_max_seq_len()";
    case BKCORE_ELAUNCH: return "HIP kernel launch failed (see bkcore_last_hip_error())";
    default: return "unknown bkcore error";
  }
}

}  // extern "C"

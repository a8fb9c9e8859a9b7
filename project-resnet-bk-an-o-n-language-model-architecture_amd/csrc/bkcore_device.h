// Device helpers shared by the translation units of libbkcore.so: scaled complex numbers, DPP moves, the soft
// clamp series, input conversion and regularisation, workspace layout.  Everything lives in an anonymous
// namespace: each translation unit gets its own copy.
#ifndef BKCORE_DEVICE_H
#define BKCORE_DEVICE_H
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
// The type of k = b c.  The operator's operands are real; the second compilation of the scan kernel
// (bkcore_complex.hip: complex he_diag / h0_super / h0_sub for the raw resolvent diagonal, which the reference
// accepts by converting every operand to complex128, bk_core.py:45-47) carries a complex k through the same code.
#ifdef BK_COMPLEX_BUILD
typedef float2 KT;
__device__ __forceinline__ float2 rstep(float2 al, float2 ka, float2 x, float2 p) {
  const float2 kp = cmul(ka, p);
  return make_float2(fmaf(al.x, x.x, fmaf(-al.y, x.y, -kp.x)), fmaf(al.x, x.y, fmaf(al.y, x.x, -kp.y)));
}
__device__ __forceinline__ float2 kscale(float2 k, float w) { return make_float2(k.x * w, k.y * w); }
__device__ __forceinline__ float2 kzero() { return make_float2(0.0f, 0.0f); }
#else
typedef float KT;
__device__ __forceinline__ float kscale(float k, float w) { return k * w; }
__device__ __forceinline__ float kzero() { return 0.0f; }
#endif
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
#ifdef BK_COMPLEX_BUILD
      float a_i[A_PAD];       // its imaginary part
#endif
      KT k_s[A_PAD + 4];      // staged k_{i-1} = b_{i-1} c_{i-1} for i = chunk_base .. chunk_base+CHUNK (CHUNK+1 values)
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

__device__ __forceinline__ float2 reg_alpha(float a, const RowCfg &cfg) {
  const float xr = a - cfg.z.x, xi = -cfg.z.y;
  const float n = fmaf(xr, xr, xi * xi);
  float f = fmaf(1e-3f, __builtin_amdgcn_rsqf(n), 1.0f);
  float add = 0.0f;
  if (!(n >= 1e-24f)) { f = 1.0f; add = 1e-3f; }
  return make_float2(fmaf(xr, f, add), xi * f);
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

template <bool V>
using BoolC = std::integral_constant<bool, V>;

// Workspace layout: B ints (per-row verdict of the fast path), padded to 256 bytes, then the lane states.
__host__ __device__ inline size_t rowflag_bytes(long B) { return (((size_t)B * sizeof(int)) + 255) & ~(size_t)255; }
constexpr int SEG_INFO = 32;       // floats of per-segment record
constexpr int MAX_SEGMENTS = 1024;
__host__ __device__ inline size_t lane_state_bytes(long B, long N) {
  return (size_t)B * (size_t)((N + CHUNK - 1) / CHUNK) * 5 * T * sizeof(float);
}

}  // namespace
#endif

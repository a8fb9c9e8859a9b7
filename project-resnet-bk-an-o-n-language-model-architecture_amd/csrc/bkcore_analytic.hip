// Analytic backward of BK-Core for MI355X (gfx950): the gradient of the operator's own forward with respect to
// he_diag, h0_super and h0_sub.
//
// The reference's backward is a heuristic blend and returns no gradient for the off-diagonals
// (src/models/bk_core.py:303-385); this file differentiates what the forward (bk_core.py:96-210) computes, written in
// ratios of consecutive minors:
//     r_0 = a'_0,          r_i = a'_i - k_{i-1} w(r_{i-1})          (theta_{i+1}/theta_i,  bk_core.py:120-140)
//     s_{N-1} = a'_{N-1},  s_m = a'_m - kphi_m  w(s_{m+1})          (phi_m/phi_{m+1},      bk_core.py:165-184)
//     w(r) = exp(g(-ln|r|)) / r,  g(d) = 88 tanh(d/88) - d          (in-recurrence soft clamp, bk_core.py:60-75)
//     G_raw,j = (1/r_j) prod_{m>j} s_m / r_m ,   G = clip50(G_raw exp(g(ln|G_raw|)))      (bk_core.py:190-208)
// kphi_m = k_{m-1} in the reference's index convention (bk_core.py:171), k_m with BKCORE_FLAG_PHI_EXACT.
// With q_j = conj(h_j) G_raw,j (h the cotangent of G_raw) and P_m = sum_{j<=m} q_j,
//     dL = sum_m Re(-P_m dr_m/r_m) + sum_{m>=1} Re(P_{m-1} ds_m/s_m)
// and the adjoints of the two ratio recurrences obey first-order recurrences that are linear over the reals
// (the clamp depends on |r| only, so they are not complex-linear):
//     lam_m = -P_m    + y - tau(r_m) Re y,   y = k_m    w(r_m) lam_{m+1}/r_{m+1}          right -> left
//     mu_m  = P_{m-1} + y - tau(s_m) Re y,   y = kphi_{m-1} w(s_m) mu_{m-1}/s_{m-1}      left -> right
//     tau(r) = tanh^2(ln|r| / 88)
//     grad_a_m = Re[(lam_m/r_m + mu_m/s_m) D_m],   D_m = d a'_m / d a_m   (regularisation, bk_core.py:78-94)
//     grad_k_m = -Re[w(r_m) lam_{m+1}/r_{m+1}]  -  Re[w(s_{m'+1}) mu_{m'}/s_{m'}] at kphi_{m'} = k_m
//     grad_b = c grad_k,  grad_c = b grad_k
// r, s and G_raw come from a forward launch inside bkcore_backward_analytic (nothing is kept from the caller's
// forward).  Two kernels, one 256-thread workgroup per row, chunks of 256 x 8 elements; the scans run over affine
// maps of R^2 (2x2 matrix + vector) with DPP inside a wave and LDS across waves:
//   bk_adjoint_phi_kernel   : left -> right; P (prefix sum) and mu; writes P, the phi part of grad_a and of grad_k
//   bk_adjoint_theta_kernel : right -> left; lam; adds the theta parts and writes grad_a, grad_b, grad_c
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/bkcore.h"
#include "bkcore_internal.h"

namespace {

constexpr int T = 256;
constexpr int EB = 8;
constexpr int CHUNK = T * EB;
constexpr int WAVES = T / 64;

__device__ __forceinline__ float2 cmul(float2 p, float2 q) {
  return make_float2(fmaf(p.x, q.x, -p.y * q.y), fmaf(p.x, q.y, p.y * q.x));
}
__device__ __forceinline__ float2 cadd(float2 p, float2 q) { return make_float2(p.x + q.x, p.y + q.y); }

// x -> A x + v on (Re, Im)
struct Aff {
  float a00, a01, a10, a11, v0, v1;
};
__device__ __forceinline__ Aff aff_identity() { return Aff{1.f, 0.f, 0.f, 1.f, 0.f, 0.f}; }
__device__ __forceinline__ float2 apply(const Aff &m, float2 x) {
  return make_float2(fmaf(m.a00, x.x, fmaf(m.a01, x.y, m.v0)), fmaf(m.a10, x.x, fmaf(m.a11, x.y, m.v1)));
}
// later after earlier
__device__ __forceinline__ Aff compose(const Aff &l, const Aff &e) {
  Aff r;
  r.a00 = fmaf(l.a00, e.a00, l.a01 * e.a10);
  r.a01 = fmaf(l.a00, e.a01, l.a01 * e.a11);
  r.a10 = fmaf(l.a10, e.a00, l.a11 * e.a10);
  r.a11 = fmaf(l.a10, e.a01, l.a11 * e.a11);
  r.v0 = fmaf(l.a00, e.v0, fmaf(l.a01, e.v1, l.v0));
  r.v1 = fmaf(l.a10, e.v0, fmaf(l.a11, e.v1, l.v1));
  return r;
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f(float old, float src) {
  return __int_as_float(
      __builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(src), CTRL, ROW_MASK, 0xF, false));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ Aff aff_dpp(const Aff &m) {  // identity where there is no source lane
  Aff r;
  r.a00 = dpp_f<CTRL, ROW_MASK>(1.0f, m.a00);
  r.a01 = dpp_f<CTRL, ROW_MASK>(0.0f, m.a01);
  r.a10 = dpp_f<CTRL, ROW_MASK>(0.0f, m.a10);
  r.a11 = dpp_f<CTRL, ROW_MASK>(1.0f, m.a11);
  r.v0 = dpp_f<CTRL, ROW_MASK>(0.0f, m.v0);
  r.v1 = dpp_f<CTRL, ROW_MASK>(0.0f, m.v1);
  return r;
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float2 sum_dpp(float2 v) {
  return make_float2(dpp_f<CTRL, ROW_MASK>(0.0f, v.x), dpp_f<CTRL, ROW_MASK>(0.0f, v.y));
}

struct SharedAff {
  Aff tot[2][WAVES];
  float2 carry[2];
};
struct SharedSum {
  float2 tot[2][WAVES];
  float2 carry[2];
};

// Exclusive scan of the lanes' maps in ascending thread order, applied to the running value kept in LDS (two
// alternating slots, so one barrier per scan is enough); returns the value entering this lane, advances the carry.
__device__ __forceinline__ float2 block_exclusive_aff(Aff A, SharedAff &sh, int &nscan) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  A = compose(A, aff_dpp<0x111, 0xF>(A));  // row_shr:1
  A = compose(A, aff_dpp<0x112, 0xF>(A));  // row_shr:2
  A = compose(A, aff_dpp<0x114, 0xF>(A));  // row_shr:4
  A = compose(A, aff_dpp<0x118, 0xF>(A));  // row_shr:8
  A = compose(A, aff_dpp<0x142, 0xA>(A));  // row_bcast:15 into rows 1, 3
  A = compose(A, aff_dpp<0x143, 0xC>(A));  // row_bcast:31 into rows 2, 3
  const int buf = nscan & 1;
  nscan++;
  if (lane == 63) sh.tot[buf][wave] = A;
  const Aff ex = aff_dpp<0x138, 0xF>(A);  // wave_shr:1
  __syncthreads();
  float2 x = sh.carry[buf];
  float2 xin = x;
#pragma unroll
  for (int w = 0; w < WAVES; ++w) {
    if (w == wave) xin = x;
    x = apply(sh.tot[buf][w], x);
  }
  if (threadIdx.x == 0) sh.carry[buf ^ 1] = x;
  return apply(ex, xin);
}

// Exclusive running sum over lanes (ascending thread order) continued across calls.
__device__ __forceinline__ float2 block_exclusive_sum(float2 v, SharedSum &sh, int &nscan) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  v = cadd(v, sum_dpp<0x111, 0xF>(v));
  v = cadd(v, sum_dpp<0x112, 0xF>(v));
  v = cadd(v, sum_dpp<0x114, 0xF>(v));
  v = cadd(v, sum_dpp<0x118, 0xF>(v));
  v = cadd(v, sum_dpp<0x142, 0xA>(v));
  v = cadd(v, sum_dpp<0x143, 0xC>(v));
  const int buf = nscan & 1;
  nscan++;
  if (lane == 63) sh.tot[buf][wave] = v;
  const float2 ex = sum_dpp<0x138, 0xF>(v);
  __syncthreads();
  float2 x = sh.carry[buf];
  float2 xin = x;
#pragma unroll
  for (int w = 0; w < WAVES; ++w) {
    if (w == wave) xin = x;
    x = cadd(x, sh.tot[buf][w]);
  }
  if (threadIdx.x == 0) sh.carry[buf ^ 1] = x;
  return cadd(xin, ex);
}

// g(d) = 88 tanh(d/88) - d
__device__ __forceinline__ float soft_g(float d) {
  if (fabsf(d) > 22.0f) return 88.0f * tanhf(d * (1.0f / 88.0f)) - d;
  const float u = d * (1.0f / 88.0f);
  const float s = u * u;
  float p = fmaf(s, 1382.0f / 155925.0f, -62.0f / 2835.0f);
  p = fmaf(s, p, 17.0f / 315.0f);
  p = fmaf(s, p, -2.0f / 15.0f);
  p = fmaf(s, p, 1.0f / 3.0f);
  return -88.0f * u * s * p;
}

// For a ratio r: inv = 1/r, w = exp(g(-ln|r|))/r, tau = tanh^2(ln|r|/88)
__device__ __forceinline__ void ratio_terms(float2 r, float2 &inv, float2 &w, float &tau) {
  const float n = fmaf(r.x, r.x, r.y * r.y);
  const float lnr = 0.5f * __logf(n);
  const float gd = soft_g(-lnr);
  const float ex = __expf(gd);
  const float rn = 1.0f / n;
  inv = make_float2(r.x * rn, -r.y * rn);
  w = make_float2(inv.x * ex, inv.y * ex);
  const float th = (gd - lnr) * (1.0f / 88.0f);  // tanh(-ln|r| / 88) = (g(d) + d) / 88
  tau = th * th;
}

// The map  x -> (c0 + y - tau Re y) * inv,  y = kk w x
__device__ __forceinline__ Aff step_map(float kk, float2 w, float tau, float2 inv, float2 c0) {
  const float2 cw = make_float2(kk * w.x, kk * w.y);
  const float t1 = 1.0f - tau;
  const float y00 = t1 * cw.x, y01 = -t1 * cw.y, y10 = cw.y, y11 = cw.x;
  Aff m;
  m.a00 = fmaf(inv.x, y00, -inv.y * y10);
  m.a01 = fmaf(inv.x, y01, -inv.y * y11);
  m.a10 = fmaf(inv.y, y00, inv.x * y10);
  m.a11 = fmaf(inv.y, y01, inv.x * y11);
  const float2 v = cmul(inv, c0);
  m.v0 = v.x;
  m.v1 = v.y;
  return m;
}

// q = conj(h) G_raw for the output G = clip50(G_raw exp(g(ln|G_raw|))) and its cotangent gbar.
__device__ __forceinline__ float2 cot_q(float2 Graw, float2 gbar) {
  const float m2 = fmaf(Graw.x, Graw.x, Graw.y * Graw.y);
  if (!(m2 > 0.0f)) return make_float2(0.f, 0.f);  // zeroed by the forward (not finite, or out of range)
  const float L = 0.5f * __logf(m2);
  const float gL = soft_g(L);
  const float F = __expf(gL);
  const float2 Gc = make_float2(Graw.x * F, Graw.y * F);
  const float2 X = make_float2(fmaf(gbar.x, Gc.x, gbar.y * Gc.y), fmaf(gbar.x, Gc.y, -gbar.y * Gc.x));  // conj(gbar) Gc
  const float mc2 = m2 * F * F;
  if (mc2 > 2500.0f) return make_float2(0.0f, X.y * 50.0f * rsqrtf(mc2));  // clipped: only the phase moves
  const float th = (gL + L) * (1.0f / 88.0f);  // tanh(L / 88)
  return make_float2(X.x - th * th * X.x, X.y);
}

// Re(x D),  D = d reg(a - z)/d a,  reg(x) = x (1 + 1e-3/|x|)
__device__ __forceinline__ float times_dreg(float2 x, float a, float2 z) {
  const float xr = a - z.x, xi = -z.y;
  const float n2 = fmaf(xr, xr, xi * xi);
  if (!(n2 > 1e-24f)) return x.x;
  const float inv = rsqrtf(n2);
  const float f1 = fmaf(1e-3f, inv, 1.0f);
  const float cf = 1e-3f * xr * inv * inv * inv;
  return x.x * (f1 - cf * xr) + x.y * (cf * xi);
}

__device__ __forceinline__ float finite_or_zero(float v) { return fabsf(v) <= 3.0e38f ? v : 0.0f; }

// ---- per-lane block I/O ----------------------------------------------------------------------------------------
// A lane owns EB consecutive elements.  When the whole block lies inside the row it is moved with 16-byte accesses
// (the struct is only 4-byte aligned: rows of N or N-1 elements start anywhere); otherwise element by element with
// `fill` for the positions outside [lo, hi).
struct __attribute__((aligned(4))) F4u {
  float v[4];
};
template <int NF>
__device__ __forceinline__ void load_block(const float *__restrict__ p, long first, long lo, long hi, float fill,
                                           float (&out)[NF]) {
  static_assert(NF % 4 == 0, "whole 16-byte pieces");
  if (first >= lo && first + NF <= hi) {
#pragma unroll
    for (int q = 0; q < NF / 4; ++q) {
      const F4u t = *reinterpret_cast<const F4u *>(p + first + 4 * q);
#pragma unroll
      for (int r = 0; r < 4; ++r) out[4 * q + r] = t.v[r];
    }
  } else {
#pragma unroll
    for (int e = 0; e < NF; ++e) {
      const long j = first + e;
      out[e] = (j >= lo && j < hi) ? p[j] : fill;
    }
  }
}
template <int NF>
__device__ __forceinline__ void store_block(float *__restrict__ p, long first, long lo, long hi,
                                            const float (&in)[NF]) {
  if (first >= lo && first + NF <= hi) {
#pragma unroll
    for (int q = 0; q < NF / 4; ++q) {
      F4u t;
#pragma unroll
      for (int r = 0; r < 4; ++r) t.v[r] = in[4 * q + r];
      *reinterpret_cast<F4u *>(p + first + 4 * q) = t;
    }
  } else {
#pragma unroll
    for (int e = 0; e < NF; ++e) {
      const long j = first + e;
      if (j >= lo && j < hi) p[j] = in[e];
    }
  }
}

// ---- left -> right ------------------------------------------------------------------------------------------------
template <int OFF>  // kphi_m = k_{m-OFF}
__global__ __launch_bounds__(T) void bk_adjoint_phi_kernel(
    const float *__restrict__ a_, long lda, const float *__restrict__ b_, long ldb, const float *__restrict__ c_,
    long ldc, const float *__restrict__ z_dev, float z_re, float z_im, long N, const float *__restrict__ Graw_,
    const float *__restrict__ gf_, const float *__restrict__ gG_, const float *__restrict__ s_,
    float *__restrict__ P_, float *__restrict__ ga_, float *__restrict__ gk_, const int *__restrict__ row_filter,
    const int *gate) {
  __shared__ SharedAff sha;
  __shared__ SharedSum shs;
  const long row = blockIdx.x;
  if (gate && __hip_atomic_load(gate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) return;
  if (row_filter && row_filter[row] == 0) return;
  const float *a = a_ + row * lda, *b = b_ + row * ldb, *c = c_ + row * ldc;
  const float *Graw = Graw_ + row * N * 2;
  const float *gf = gf_ ? gf_ + row * N * 2 : nullptr, *gG = gG_ ? gG_ + row * N * 2 : nullptr;
  const float *sv = s_ + row * N * 2;
  float *P = P_ + row * N * 2;
  float *ga = ga_ + row * N;
  float *gk = gk_ ? gk_ + row * (N - 1) : nullptr;
  const float2 z = z_dev ? make_float2(z_dev[0], z_dev[1]) : make_float2(z_re, z_im);
  const int t = threadIdx.x;
  const int nch = (int)((N + CHUNK - 1) / CHUNK);
  int nsum = 0, naff = 0;
  if (t == 0) {
    sha.carry[0] = make_float2(0.f, 0.f);
    shs.carry[0] = make_float2(0.f, 0.f);
  }
  __syncthreads();
  for (int ch = 0; ch < nch; ++ch) {
    const long first = (long)ch * CHUNK + (long)t * EB;
    float q[2 * EB];
    {
      float Gv[2 * EB], gv[2 * EB];
      load_block<2 * EB>(Graw, 2 * first, 0, 2 * N, 0.0f, Gv);
      if (gf) {
        load_block<2 * EB>(gf, 2 * first, 0, 2 * N, 0.0f, gv);
      } else {
#pragma unroll
        for (int e = 0; e < 2 * EB; ++e) gv[e] = 0.0f;
      }
      if (gG) {
        float hv[2 * EB];
        load_block<2 * EB>(gG, 2 * first, 0, 2 * N, 0.0f, hv);
#pragma unroll
        for (int e = 0; e < 2 * EB; ++e) gv[e] += hv[e];
      }
#pragma unroll
      for (int e = 0; e < EB; ++e) {
        // a cotangent that is not a number counts as 0, as in the reference's backward (bk_core.py:321-325)
        const float gx = fabsf(gv[2 * e]) <= 3.0e38f ? gv[2 * e] : 0.0f;
        const float gy = fabsf(gv[2 * e + 1]) <= 3.0e38f ? gv[2 * e + 1] : 0.0f;
        const float2 qq = cot_q(make_float2(Gv[2 * e], Gv[2 * e + 1]), make_float2(gx, gy));  // 0 for G_raw = 0
        q[2 * e] = qq.x;
        q[2 * e + 1] = qq.y;
      }
    }
    float2 sum = make_float2(0.f, 0.f);
#pragma unroll
    for (int e = 0; e < EB; ++e) sum = cadd(sum, make_float2(q[2 * e], q[2 * e + 1]));
    float2 Prun = block_exclusive_sum(sum, shs, nsum);  // P_{first-1}
    float sbuf[2 * EB], bk[EB], ck[EB];
    load_block<2 * EB>(sv, 2 * first, 0, 2 * N, 1.0f, sbuf);
    load_block<EB>(b, first - 1 - OFF, 0, N - 1, 0.0f, bk);  // k index of element j is j - 1 - OFF
    load_block<EB>(c, first - 1 - OFF, 0, N - 1, 0.0f, ck);
    Aff st[EB];
    float2 wv[EB];
    Aff A = aff_identity();
#pragma unroll
    for (int e = 0; e < EB; ++e) {
      const long j = first + e;
      st[e] = aff_identity();
      wv[e] = make_float2(0.f, 0.f);
      if (j >= 1 && j < N) {
        float2 inv;
        float tau;
        ratio_terms(make_float2(sbuf[2 * e], sbuf[2 * e + 1]), inv, wv[e], tau);
        st[e] = step_map(bk[e] * ck[e], wv[e], tau, inv, Prun);  // c0 = P_{j-1}
      }
      Prun = cadd(Prun, make_float2(q[2 * e], q[2 * e + 1]));
      q[2 * e] = Prun.x;  // q now holds P_j
      q[2 * e + 1] = Prun.y;
      A = compose(st[e], A);
    }
    store_block<2 * EB>(P, 2 * first, 0, 2 * N, q);
    float2 x = block_exclusive_aff(A, sha, naff);  // mu_{first-1} / s_{first-1}
    float av[EB], gav[EB], gkv[EB];
    load_block<EB>(a, first, 0, N, 0.0f, av);
#pragma unroll
    for (int e = 0; e < EB; ++e) {
      const long j = first + e;
      gkv[e] = -fmaf(x.x, wv[e].x, -x.y * wv[e].y);
      gav[e] = 0.0f;
      if (j < N) {
        x = apply(st[e], x);
        if (j >= 1) gav[e] = times_dreg(x, av[e], z);
      }
    }
    store_block<EB>(ga, first, 0, N, gav);
    // element j >= 1 wrote the phi part of grad_k at index j - 1 - OFF (0 from j = 1: mu_0 = 0)
    if (gk) store_block<EB>(gk, first - 1 - OFF, 0, N - 1 - OFF, gkv);
  }
}

// ---- right -> left ------------------------------------------------------------------------------------------------
template <int OFF>
__global__ __launch_bounds__(T) void bk_adjoint_theta_kernel(
    const float *__restrict__ a_, long lda, const float *__restrict__ b_, long ldb, const float *__restrict__ c_,
    long ldc, const float *__restrict__ z_dev, float z_re, float z_im, long N, const float *__restrict__ r_,
    const float *__restrict__ P_, float *__restrict__ ga_, float *__restrict__ gb_, float *__restrict__ gc_,
    const int *__restrict__ row_filter, const int *gate) {
  __shared__ SharedAff sha;
  const long row = blockIdx.x;
  if (gate && __hip_atomic_load(gate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) return;
  if (row_filter && row_filter[row] == 0) return;
  const float *a = a_ + row * lda, *b = b_ + row * ldb, *c = c_ + row * ldc;
  const float *rv = r_ + row * N * 2, *P = P_ + row * N * 2;
  float *ga = ga_ + row * N;
  float *gb = gb_ ? gb_ + row * (N - 1) : nullptr;
  float *gc = gc_ ? gc_ + row * (N - 1) : nullptr;
  const float2 z = z_dev ? make_float2(z_dev[0], z_dev[1]) : make_float2(z_re, z_im);
  const int t = threadIdx.x;
  const int nch = (int)((N + CHUNK - 1) / CHUNK);
  int naff = 0;
  if (t == 0) sha.carry[0] = make_float2(0.f, 0.f);
  __syncthreads();
  for (int ch = nch - 1; ch >= 0; --ch) {
    const long first = (long)ch * CHUNK + (long)(T - 1 - t) * EB;  // descending segments over ascending threads
    float rbuf[2 * EB], pbuf[2 * EB], bk[EB], ck[EB];
    load_block<2 * EB>(rv, 2 * first, 0, 2 * N, 1.0f, rbuf);
    load_block<2 * EB>(P, 2 * first, 0, 2 * N, 0.0f, pbuf);
    load_block<EB>(b, first, 0, N - 1, 0.0f, bk);
    load_block<EB>(c, first, 0, N - 1, 0.0f, ck);
    Aff st[EB];
    float2 wv[EB];
    Aff A = aff_identity();
#pragma unroll
    for (int qi = 0; qi < EB; ++qi) {
      const int e = EB - 1 - qi;
      const long j = first + e;
      st[e] = aff_identity();
      wv[e] = make_float2(0.f, 0.f);
      if (j < N) {
        float2 inv;
        float tau;
        ratio_terms(make_float2(rbuf[2 * e], rbuf[2 * e + 1]), inv, wv[e], tau);
        st[e] = step_map(bk[e] * ck[e], wv[e], tau, inv, make_float2(-pbuf[2 * e], -pbuf[2 * e + 1]));
      }
      A = compose(st[e], A);
    }
    float2 x = block_exclusive_aff(A, sha, naff);  // lam_{j+1} / r_{j+1} for the lane's last element j
    float av[EB], gav[EB], gkp[EB], gbv[EB], gcv[EB];
    load_block<EB>(a, first, 0, N, 0.0f, av);
    load_block<EB>(ga, first, 0, N, 0.0f, gav);
    if (gb) {
      load_block<EB>(gb, first, 0, N - 1 - OFF, 0.0f, gkp);  // the phi part of grad_k, left there by the phi kernel
    } else {
#pragma unroll
      for (int e = 0; e < EB; ++e) gkp[e] = 0.0f;
    }
#pragma unroll
    for (int qi = 0; qi < EB; ++qi) {
      const int e = EB - 1 - qi;
      const long j = first + e;
      const float gk = gkp[e] - fmaf(x.x, wv[e].x, -x.y * wv[e].y);
      if (j < N) x = apply(st[e], x);
      gav[e] = finite_or_zero(gav[e] + times_dreg(x, av[e], z));
      gbv[e] = finite_or_zero(ck[e] * gk);
      gcv[e] = finite_or_zero(bk[e] * gk);
    }
    store_block<EB>(ga, first, 0, N, gav);
    if (gb) {
      store_block<EB>(gb, first, 0, N - 1, gbv);
      store_block<EB>(gc, first, 0, N - 1, gcv);
    }
  }
}


// ==== both sweeps in one kernel, the ratios recomputed in it =======================================================
// Nothing but G is kept from the forward (as in the reference, bk_core.py:293).  The cotangent of G_raw follows from G
// itself: with X = conj(gbar) G,  q = X - tanh^2(L_raw/88) Re X  and  tanh(L_raw/88) = ln|G| / 88  (the final clamp is
// ln|G| = 88 tanh(L_raw/88)); a clipped element (|G| = 50) passes only Im X; G = 0 passes nothing.  The ratios r and s
// are recomputed from a, b, c exactly as the forward does: wherever the ratio maps contract, a lane gets the ratio
// entering its 8 elements by running the map over the 16 elements before them (after them, for s), verified against
// the neighbouring lane's own value.  The adjoint recurrences contract at the same rate (their linear part is
// k w(r) / r), so the value entering a lane is taken from the two lanes before it,
//     x_in(l) = v(l-1) + m(l-1) v(l-2),     v = the lane's 8 maps applied to 0,  m = the product of their linear parts,
// checked by |m(l-1) m(l-2)| <= 2e-5.  One workgroup per row runs
//   phase phi   : chunks left -> right : s, P (prefix sums of q), mu; writes the phi part of grad_a and of grad_k
//                 into the output buffers
//   phase theta : chunks right -> left : r, P = total - suffix sums, lam; adds the theta parts, writes grad_a, grad_b,
//                 grad_c  (it starts where the first phase ended, on data that is still in the caches)
// with a, b, c, G and the cotangent staged through LDS by coalesced loads.  A row with a failed check is flagged
// (rowbad[row], state[5]) and recomputed by the three-pass path above, whose launches follow and skip every other row.
// development: BK_SWEEP_STAMPS=1 writes s_memtime stamps of thread 0 at the stage boundaries of every chunk of both
// phases into dbg (8 slots of 8 bytes per (row, phase, chunk))
#ifndef BK_SWEEP_STAMPS
#define BK_SWEEP_STAMPS 0
#endif
#if BK_SWEEP_STAMPS
#define BK_SSTAMP(k)                                                                                                 \
  do {                                                                                                               \
    __builtin_amdgcn_sched_barrier(0);                                                                               \
    const unsigned long long tm_ = __builtin_amdgcn_s_memtime();                                                     \
    if (threadIdx.x == 0) dbg[(((long)blockIdx.x * 2 + (THETA ? 1 : 0)) * nch + cc) * 8 + (k)] = tm_;                \
    __builtin_amdgcn_sched_barrier(0);                                                                               \
  } while (0)
#else
#define BK_SSTAMP(k) do { } while (0)
#endif
#ifndef BK_SWEEP_PREFETCH
#define BK_SWEEP_PREFETCH 1   // raw inputs of the next chunk held in registers during the current chunk's arithmetic
#endif
#ifndef BK_SWEEP_WG
#define BK_SWEEP_WG 2   // workgroups per CU the register budget is set for
#endif
constexpr int WH = 16;                      // warm-up steps = halo staged on both sides of a chunk
constexpr int SN = CHUNK + 2 * WH;          // staged elements per chunk
constexpr int SPAD = SN + SN / EB + 8;
constexpr int QPAD = CHUNK + CHUNK / EB + 8;
static_assert(SN % EB == 0 && WH % EB == 0, "padded slot arithmetic");
// offset of slot d (may be negative) from a slot whose index is a multiple of EB, in the layout q -> q + q / EB
__device__ __forceinline__ constexpr int soff(int d) { return d + (d >= 0 ? d / EB : -((EB - 1 - d) / EB)); }

struct SharedSweep {
  float2 al_s[SPAD];            // regularised diagonal of elements base - WH .. base + CHUNK + WH - 1
  float k_s[SPAD];              // theta: k_{i-1};  phi: kphi_i (0 from the last element on)
  float2 q_s[QPAD];             // q of elements base .. base + CHUNK - 1
  float a_s[QPAD];              // he_diag of the same elements
  float2 bc_s[QPAD];            // theta: (b_j, c_j) of the same elements
  float2 sumtot[2][WAVES];
  float2 sumcarry[2];
  float2 wedge[2][WAVES];       // last lane of each wave: its warm-up estimate, for the first lane of the next wave
  float2 edge[2];               // the same from the last lane of the chunk processed before
  float4 xedge[2][WAVES][2];    // (v, m) of lanes 62 and 63 of each wave
};

constexpr float WALL_C = 0.69314718055994531f / 176.0f;
constexpr float WALL_A = 88.0f * WALL_C * WALL_C * WALL_C / 3.0f;
constexpr float WALL_B = 176.0f * WALL_C * WALL_C * WALL_C * WALL_C * WALL_C / 15.0f;

__device__ __forceinline__ float2 reg_diag(float a, float2 z) {  // reg(a - z), bk_core.py:78-94
  const float xr = a - z.x, xi = -z.y;
  const float n = fmaf(xr, xr, xi * xi);
  float f = fmaf(1e-3f, __builtin_amdgcn_rsqf(n), 1.0f);
  float add = 0.0f;
  if (!(n >= 1e-24f)) { f = 1.0f; add = 1e-3f; }
  return make_float2(fmaf(xr, f, add), xi * f);
}

// Steps of the ratio map r = al - k om, returning om(r) = exp(g(-ln|r|)) / r.  Coarse: no clamp factor; short: the
// leading term of the clamp factor, unrefined reciprocal.
__device__ __forceinline__ float2 om_coarse(float2 al, float k, float2 om) {
  const float2 r = make_float2(fmaf(-k, om.x, al.x), fmaf(-k, om.y, al.y));
  const float y = __builtin_amdgcn_rcpf(fmaf(r.x, r.x, r.y * r.y));
  return make_float2(r.x * y, -r.y * y);
}
__device__ __forceinline__ float2 om_short(float2 al, float k, float2 om) {
  const float2 r = make_float2(fmaf(-k, om.x, al.x), fmaf(-k, om.y, al.y));
  const float n = fmaf(r.x, r.x, r.y * r.y);
  const float Ln = __builtin_amdgcn_logf(n);
  const float y = __builtin_amdgcn_rcpf(n) * fmaf(WALL_A * Ln, Ln * Ln, 1.0f);
  return make_float2(r.x * y, -r.y * y);
}
__device__ __forceinline__ float2 om_full(float2 al, float k, float2 om, float2 &inv, float &wc, float &tau,
                                          float &lmin) {
  const float2 r = make_float2(fmaf(-k, om.x, al.x), fmaf(-k, om.y, al.y));
  const float n = fmaf(r.x, r.x, r.y * r.y);
  float y = __builtin_amdgcn_rcpf(n);
  y = fmaf(y, fmaf(-n, y, 1.0f), y);
  inv = make_float2(r.x * y, -r.y * y);
  const float Ln = __builtin_amdgcn_logf(n);  // log2 |r|^2: ln|r| / 88 = WALL_C Ln
  lmin = fminf(lmin, Ln);
  const float t2 = Ln * Ln, q3 = t2 * Ln;
  wc = fmaf(q3, fmaf(q3, 0.5f * WALL_A * WALL_A, fmaf(t2, -WALL_B, WALL_A)), 1.0f);
  const float u2 = WALL_C * WALL_C * t2;       // tanh^2(u) = u^2 (1 - 2u^2/3 ...)
  tau = u2 * fmaf(u2, -2.0f / 3.0f, 1.0f);
  return make_float2(inv.x * wc, inv.y * wc);
}
__device__ __forceinline__ bool om_differs(float2 p, float2 q) {  // |p - q| > 2e-5 |q|, true for anything not a number
  const float dx = p.x - q.x, dy = p.y - q.y;
  return !(fmaf(dx, dx, dy * dy) <= 4.0e-10f * fmaf(q.x, q.x, q.y * q.y));
}

// q = X - tanh^2(L_raw / 88) Re X for the saved output G and its cotangent gbar (see above)
__device__ __forceinline__ float2 cot_q_saved(float2 G, float2 gbar) {
  const float m2 = fmaf(G.x, G.x, G.y * G.y);
  const float2 X = make_float2(fmaf(gbar.x, G.x, gbar.y * G.y), fmaf(gbar.x, G.y, -gbar.y * G.x));  // conj(gbar) G
  const float th = __builtin_amdgcn_logf(m2) * (0.5f * 0.69314718055994531f / 88.0f);  // ln|G| / 88
  float qx = fmaf(-th * th, X.x, X.x);
  if (m2 > 2499.99f) qx = 0.0f;
  if (!(m2 > 0.0f)) return make_float2(0.f, 0.f);
  return make_float2(qx, X.y);
}

struct __attribute__((aligned(4))) F2u {
  float v[2];
};
__device__ __forceinline__ float2 ld2(const float *p) {
  const F2u t = *reinterpret_cast<const F2u *>(p);
  return make_float2(t.v[0], t.v[1]);
}

// The raw values one thread loads for one chunk (coalesced: element base - WH + t + 256 j), held in registers while
// the chunk before it is processed.
constexpr int ROUNDS = (SN + T - 1) / T;
struct Raw {
  float av[ROUNDS], bv[ROUNDS], cv[ROUNDS];
  float2 Gv[EB], gv[EB];
};

// CHECK = false when every staged element and its off-diagonal entries lie inside the row.
template <int OFF, bool THETA, bool CHECK>
__device__ __forceinline__ void issue_loads(const float *__restrict__ a, const float *__restrict__ b,
                                            const float *__restrict__ c, const float *__restrict__ G,
                                            const float *__restrict__ gf, long base, long N, Raw &R) {
  const int t = threadIdx.x;
  constexpr int KSH = THETA ? 1 : OFF;  // slot i holds k_{i - KSH}
#pragma unroll
  for (int j = 0; j < ROUNDS; ++j) {
    const int q = t + j * T;
    const long i = base - WH + q, ki = i - KSH;
    R.av[j] = 1.0f;
    R.bv[j] = 0.0f;
    R.cv[j] = 0.0f;
    if (CHECK) {
      const bool in = (j < ROUNDS - 1 || q < SN);
      if (in && i >= 0 && i < N) R.av[j] = __builtin_nontemporal_load(a + i);
      // phi: kphi is 0 from the last element on (the recurrence starts there)
      if (in && ki >= 0 && ki <= N - 2 && (THETA || i <= N - 2)) {
        R.bv[j] = __builtin_nontemporal_load(b + ki);
        R.cv[j] = __builtin_nontemporal_load(c + ki);
      }
    } else if (j < ROUNDS - 1 || q < SN) {
      R.av[j] = __builtin_nontemporal_load(a + i);
      R.bv[j] = __builtin_nontemporal_load(b + ki);
      R.cv[j] = __builtin_nontemporal_load(c + ki);
    }
  }
#pragma unroll
  for (int r = 0; r < EB; ++r) {
    const long i = base + t + r * T;
    R.Gv[r] = make_float2(0.f, 0.f);
    R.gv[r] = make_float2(0.f, 0.f);
    if (!CHECK || i < N) {
      R.Gv[r] = ld2(G + 2 * i);
      if (gf) R.gv[r] = ld2(gf + 2 * i);
    }
  }
}
template <int OFF, bool THETA>
__device__ __forceinline__ void issue_chunk(const float *__restrict__ a, const float *__restrict__ b,
                                            const float *__restrict__ c, const float *__restrict__ G,
                                            const float *__restrict__ gf, long base, long N, Raw &R) {
  if (base - WH - 1 >= 0 && base + CHUNK + WH - 1 <= N - 2) issue_loads<OFF, THETA, false>(a, b, c, G, gf, base, N, R);
  else issue_loads<OFF, THETA, true>(a, b, c, G, gf, base, N, R);
}

// Writes a chunk's staged form: regularised diagonal and off-diagonal products with their halo, q, and for the
// lanes' output pass the raw diagonal (and, in the theta phase, b and c) of the chunk's own elements.
template <bool THETA>
__device__ __forceinline__ void commit_chunk(const Raw &R, const float *__restrict__ gG, long base, long N, float2 z,
                                             SharedSweep &sh) {
  const int t = threadIdx.x;
#pragma unroll
  for (int j = 0; j < ROUNDS; ++j) {
    const int q = t + j * T;
    if (j < ROUNDS - 1 || q < SN) {
      sh.k_s[q + q / EB] = R.bv[j] * R.cv[j];
      sh.al_s[q + q / EB] = reg_diag(R.av[j], z);
      const int e = q - WH;
      if (e >= 0 && e < CHUNK) sh.a_s[e + e / EB] = R.av[j];
      if (THETA && e >= 1 && e <= CHUNK) sh.bc_s[(e - 1) + (e - 1) / EB] = make_float2(R.bv[j], R.cv[j]);
    }
  }
#pragma unroll
  for (int r = 0; r < EB; ++r) {
    const int e = t + r * T;
    float2 g = R.gv[r];
    if (gG && base + e < N) g = cadd(g, ld2(gG + 2 * (base + e)));
    // a cotangent that is not a number counts as 0, as in the reference's backward (bk_core.py:321-325)
    const float gx = fabsf(g.x) <= 3.0e38f ? g.x : 0.0f;
    const float gy = fabsf(g.y) <= 3.0e38f ? g.y : 0.0f;
    sh.q_s[e + e / EB] = cot_q_saved(R.Gv[r], make_float2(gx, gy));
  }
}

// One step of an adjoint recurrence:  x -> (c0 + y - tau Re y) inv,  y = kk w x,  w = inv wc
__device__ __forceinline__ float2 adj_step(float2 x, float2 cw, float2 inv, float tau, float2 c0) {
  const float2 y = cmul(cw, x);
  const float2 u = make_float2(fmaf(-tau, y.x, y.x) + c0.x, y.y + c0.y);
  return cmul(inv, u);
}

template <int OFF, bool THETA>
__device__ __forceinline__ void sweep_phase(const float *__restrict__ a, const float *__restrict__ b,
                                            const float *__restrict__ c, const float *__restrict__ G,
                                            const float *__restrict__ gf, const float *__restrict__ gG,
                                            float *__restrict__ ga, float *__restrict__ gb, float *__restrict__ gc,
                                            long N, float2 z, float2 Q, SharedSweep &sh, bool &bad, int &par,
                                            unsigned long long *dbg) {
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int nch = (int)((N + CHUNK - 1) / CHUNK);
  const int seg = THETA ? T - 1 - t : t;  // theta: descending segments over ascending threads
  if (t == 0) {
    sh.sumcarry[par & 1] = make_float2(0.f, 0.f);
    sh.xedge[(par & 1) ^ 1][WAVES - 1][0] = make_float4(0.f, 0.f, 0.f, 0.f);
    sh.xedge[(par & 1) ^ 1][WAVES - 1][1] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  Raw R;
  if (BK_SWEEP_PREFETCH) issue_chunk<OFF, THETA>(a, b, c, G, gf, (long)(THETA ? nch - 1 : 0) * CHUNK, N, R);
  for (int cc = 0; cc < nch; ++cc, ++par) {
    const int ch = THETA ? nch - 1 - cc : cc;
    const int buf = par & 1;
    const long base = (long)ch * CHUNK;
    const long first = base + (long)seg * EB;
    BK_SSTAMP(0);
    if (!BK_SWEEP_PREFETCH) issue_chunk<OFF, THETA>(a, b, c, G, gf, base, N, R);
    __syncthreads();  // every lane finished reading the staging area
    commit_chunk<THETA>(R, gG, base, N, z, sh);
    __syncthreads();
    BK_SSTAMP(1);
    const float2 *al_l = sh.al_s + (WH + seg * EB) + (WH + seg * EB) / EB;
    const float *k_l = sh.k_s + (WH + seg * EB) + (WH + seg * EB) / EB;
    const float2 *q_l = sh.q_s + seg * (EB + 1);
    // What the lane's chains read from the staging area is loaded in batches, each one issued before the arithmetic on
    // the batch before it.  Ratio chain: warm-up over the WH elements before the lane's own in the direction of the
    // recurrence (past the end of the row the staged values give om * 0), then the lane's own.
    float2 alw[WH], alo[EB], q[EB], inv[EB];
    float kw[WH], ko[EB], kx, wc[EB], tau[EB];
    float2 om = make_float2(0.f, 0.f);
    float lmin = 0.0f;
#pragma unroll
    for (int j = 0; j < WH / 2; ++j) {
      const int d = THETA ? j - WH : EB + WH - 1 - j;
      alw[j] = al_l[soff(d)];
      kw[j] = k_l[soff(d)];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = WH / 2; j < WH; ++j) {
      const int d = THETA ? j - WH : EB + WH - 1 - j;
      alw[j] = al_l[soff(d)];
      kw[j] = k_l[soff(d)];
    }
#pragma unroll
    for (int j = 0; j < WH / 2; ++j) om = om_coarse(alw[j], kw[j], om);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int e = 0; e < EB; ++e) {
      alo[e] = al_l[soff(e)];
      ko[e] = k_l[soff(e)];
      q[e] = q_l[e];
    }
    kx = k_l[soff(THETA ? EB : -1)];
#pragma unroll
    for (int j = WH / 2; j < WH; ++j) {
      if (j < 3 * WH / 4) om = om_short(alw[j], kw[j], om);
      else {
        float2 iv;
        float wv, tv, lv = 0.0f;
        om = om_full(alw[j], kw[j], om, iv, wv, tv, lv);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    const float2 west = om;
#pragma unroll
    for (int j = 0; j < EB; ++j) {
      const int e = THETA ? j : EB - 1 - j;
      om = om_full(alo[e], ko[e], om, inv[e], wc[e], tau[e], lmin);
    }
    float2 sum = make_float2(0.f, 0.f);
#pragma unroll
    for (int e = 0; e < EB; ++e) sum = cadd(sum, q[e]);
    // the off-diagonal product of the adjoint step at element e: theta k_j = slot e + 1, phi kphi_{j-1} = slot e - 1
    float kk[EB];
#pragma unroll
    for (int e = 0; e < EB; ++e) kk[e] = THETA ? (e + 1 < EB ? ko[e + 1 < EB ? e + 1 : 0] : kx) : (e >= 1 ? ko[e >= 1 ? e - 1 : 0] : kx);
    if (THETA && base + CHUNK > N) {  // lam = 0 enters the last element: elements past it map everything to 0
#pragma unroll
      for (int e = 0; e < EB; ++e)
        if (first + e >= N) inv[e] = make_float2(0.f, 0.f);
    }
    // lane sums of q -> exclusive prefix in thread order
    float2 incl = sum;
    incl = cadd(incl, sum_dpp<0x111, 0xF>(incl));
    incl = cadd(incl, sum_dpp<0x112, 0xF>(incl));
    incl = cadd(incl, sum_dpp<0x114, 0xF>(incl));
    incl = cadd(incl, sum_dpp<0x118, 0xF>(incl));
    incl = cadd(incl, sum_dpp<0x142, 0xA>(incl));
    incl = cadd(incl, sum_dpp<0x143, 0xC>(incl));
    if (lane == 63) {
      sh.sumtot[buf][wave] = incl;
      sh.wedge[buf][wave] = west;
    }
    if (t == T - 1) sh.edge[buf] = west;
    float2 run = sum_dpp<0x138, 0xF>(incl);  // wave_shr:1
    float2 nb = sum_dpp<0x138, 0xF>(west);
    BK_SSTAMP(2);
    __syncthreads();
    BK_SSTAMP(3);
    {
      float2 x = sh.sumcarry[buf];
#pragma unroll
      for (int w = 0; w < WAVES; ++w) {
        if (w == wave) run = cadd(run, x);
        x = cadd(x, sh.sumtot[buf][w]);
      }
      if (t == 0) sh.sumcarry[buf ^ 1] = x;
      // the previous lane's warm-up ran over this lane's elements: both must agree on om after the lane's last step
      if (lane == 0) nb = wave > 0 ? sh.wedge[buf][wave - 1] : sh.edge[buf ^ 1];
      const bool have_nb = (t > 0 || cc > 0) && (THETA ? first + EB < N : first < N);
      if (have_nb) bad |= om_differs(nb, om);
      if (first < N) bad |= !(lmin >= -13.3f);
    }
    // the next chunk's raw values travel while the adjoint chains of this one run
    if (BK_SWEEP_PREFETCH && cc + 1 < nch) issue_chunk<OFF, THETA>(a, b, c, G, gf, base + (THETA ? -CHUNK : CHUNK), N, R);
    float gav[EB], gkp[EB];
    if (THETA) {  // what the first phase left in the output buffers
      load_block<EB>(ga, first, 0, N, 0.0f, gav);
      if (gb) load_block<EB>(gb, first, 0, N - 1 - OFF, 0.0f, gkp);
    }
    // the lane's maps applied to 0, and the product of their linear parts
    float2 v = make_float2(0.f, 0.f), m = make_float2(1.f, 0.f);
    {
      float2 r2 = run;
#pragma unroll
      for (int j = 0; j < EB; ++j) {
        const int e = THETA ? EB - 1 - j : j;
        const float g = kk[e] * wc[e];
        const float2 cw = make_float2(inv[e].x * g, inv[e].y * g);
        // phi: c0 = P_{j-1};  theta: c0 = -P_j = (sum of q to the right of j) - total
        const float2 c0 = THETA ? make_float2(r2.x - Q.x, r2.y - Q.y) : r2;
        v = adj_step(v, cw, inv[e], tau[e], c0);
        m = cmul(m, cmul(inv[e], cw));
        r2 = cadd(r2, q[e]);
      }
    }
    if (lane >= 62) sh.xedge[buf][wave][lane - 62] = make_float4(v.x, v.y, m.x, m.y);
    float2 v1 = sum_dpp<0x138, 0xF>(v), m1 = sum_dpp<0x138, 0xF>(m);
    BK_SSTAMP(4);
    __syncthreads();
    BK_SSTAMP(5);
    if (lane == 0) {
      const float4 s1 = wave > 0 ? sh.xedge[buf][wave - 1][1] : sh.xedge[buf ^ 1][WAVES - 1][1];
      v1 = make_float2(s1.x, s1.y);
      m1 = make_float2(s1.z, s1.w);
    }
    float2 v2 = sum_dpp<0x138, 0xF>(v1), m2 = sum_dpp<0x138, 0xF>(m1);
    if (lane == 0) {
      const float4 s2 = wave > 0 ? sh.xedge[buf][wave - 1][0] : sh.xedge[buf ^ 1][WAVES - 1][0];
      v2 = make_float2(s2.x, s2.y);
      m2 = make_float2(s2.z, s2.w);
    }
    float2 x = cadd(v1, cmul(m1, v2));
    bad |= !(fmaf(m1.x, m1.x, m1.y * m1.y) * fmaf(m2.x, m2.x, m2.y * m2.y) <= 4.0e-10f);
    const float *a_l = sh.a_s + seg * (EB + 1);
    const float2 *bc_l = sh.bc_s + seg * (EB + 1);
    float ao[EB];
    float2 bco[EB];
#pragma unroll
    for (int e = 0; e < EB; ++e) {
      ao[e] = a_l[e];
      if (THETA) bco[e] = bc_l[e];
    }
    __builtin_amdgcn_sched_barrier(0);
    float gkv[EB];
    {
      float2 r2 = run;
#pragma unroll
      for (int j = 0; j < EB; ++j) {
        const int e = THETA ? EB - 1 - j : j;
        const float g = kk[e] * wc[e];
        const float2 w = make_float2(inv[e].x * wc[e], inv[e].y * wc[e]);
        const float2 cw = make_float2(inv[e].x * g, inv[e].y * g);
        const float2 c0 = THETA ? make_float2(r2.x - Q.x, r2.y - Q.y) : r2;
        const float gk = -fmaf(x.x, w.x, -x.y * w.y);
        x = adj_step(x, cw, inv[e], tau[e], c0);
        r2 = cadd(r2, q[e]);
        const float da = times_dreg(x, ao[e], z);
        if (THETA) {
          gav[e] = finite_or_zero(gav[e] + da);
          if (gb) {
            const float gt = gkp[e] + gk;
            const float2 bc = bco[e];
            gkv[e] = finite_or_zero(bc.y * gt);
            gkp[e] = finite_or_zero(bc.x * gt);
          }
        } else {
          gav[e] = da;
          gkv[e] = gk;
        }
      }
    }
    BK_SSTAMP(6);
    store_block<EB>(ga, first, 0, N, gav);
    if (gb) {
      if (THETA) {
        store_block<EB>(gb, first, 0, N - 1, gkv);
        store_block<EB>(gc, first, 0, N - 1, gkp);
      } else {
        // element j >= 1 wrote the phi part of grad_k at index j - 1 - OFF (0 from j = 1: mu_0 = 0)
        store_block<EB>(gb, first - 1 - OFF, 0, N - 1 - OFF, gkv);
      }
    }
    BK_SSTAMP(7);
  }
}

template <int OFF>  // kphi_m = k_{m-OFF}
__global__ __launch_bounds__(T, BK_SWEEP_WG) void bk_adjoint_sweeps(
    const float *__restrict__ a_, long lda, const float *__restrict__ b_, long ldb, const float *__restrict__ c_,
    long ldc, const float *__restrict__ z_dev, float z_re, float z_im, long N, const float *__restrict__ G_,
    const float *__restrict__ gf_, const float *__restrict__ gG_, float *__restrict__ ga_, float *__restrict__ gb_,
    float *__restrict__ gc_, int *__restrict__ rowbad, int *state, unsigned long long *dbg) {
  __shared__ SharedSweep sh;
  const long row = blockIdx.x;
  const float *a = a_ + row * lda;
  const float *b = N >= 2 ? b_ + row * ldb : a, *c = N >= 2 ? c_ + row * ldc : a;
  const float *G = G_ + row * N * 2;
  const float *gf = gf_ ? gf_ + row * N * 2 : nullptr, *gG = gG_ ? gG_ + row * N * 2 : nullptr;
  float *ga = ga_ + row * N;
  float *gb = gb_ ? gb_ + row * (N - 1) : nullptr;
  float *gc = gc_ ? gc_ + row * (N - 1) : nullptr;
  const float2 z = z_dev ? make_float2(z_dev[0], z_dev[1]) : make_float2(z_re, z_im);
  bool bad = false;
  int par = 0;
  sweep_phase<OFF, false>(a, b, c, G, gf, gG, ga, gb, gc, N, z, make_float2(0.f, 0.f), sh, bad, par, dbg);
  __syncthreads();
  const float2 Q = sh.sumcarry[par & 1];  // sum of q over the row
  __syncthreads();
  sweep_phase<OFF, true>(a, b, c, G, gf, gG, ga, gb, gc, N, z, Q, sh, bad, par, dbg);
  const int any_bad = __syncthreads_or(bad ? 1 : 0);
  if (threadIdx.x == 0) {
    rowbad[row] = any_bad;
    if (any_bad) __hip_atomic_store(&state[5], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

__global__ void bk_adjoint_finish(int *state) {
  if (threadIdx.x == 0) __hip_atomic_store(&state[5], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

}  // namespace

extern "C" {

size_t bkcore_backward_analytic_workspace_bytes(int64_t B, int64_t N) {
  if (B <= 0 || N <= 0) return 0;
  // forward scratch + four complex64 planes (G_raw, r, s, P of the three-pass path; the P plane also takes G when the
  // caller did not keep it) + one flag per row
  return ((bkcore_workspace_bytes(B, N) + 255) & ~(size_t)255) + 4 * (size_t)B * (size_t)N * 2 * sizeof(float) +
         (((size_t)B * sizeof(int) + 255) & ~(size_t)255);
}

size_t bkcore_backward_analytic_rowflag_offset(int64_t B, int64_t N) {
  if (B <= 0 || N <= 0) return 0;
  return ((bkcore_workspace_bytes(B, N) + 255) & ~(size_t)255) + 4 * (size_t)B * (size_t)N * 2 * sizeof(float);
}

int bkcore_backward_analytic(const float *a, int64_t lda, const float *b, int64_t ldb, const float *c, int64_t ldc,
                             const float *z_dev, float z_re, float z_im, int64_t B, int64_t N, const float *G_saved,
                             const float *grad_features, const float *grad_G, float *grad_a, float *grad_b,
                             float *grad_c, void *workspace, size_t workspace_bytes, void *state, unsigned flags,
                             void *stream) {
  if (B < 0 || N < 0 || lda < 0 || ldb < 0 || ldc < 0) return BKCORE_EINVAL;
  if (B == 0 || N == 0) return BKCORE_OK;
  if (!a || !grad_a || (N > 1 && (!b || !c))) return BKCORE_EINVAL;
  if ((grad_b == nullptr) != (grad_c == nullptr)) return BKCORE_EINVAL;
  if (flags & ~BKCORE_FLAG_PHI_EXACT) return BKCORE_EINVAL;
  if (N > bkcore_max_seq_len()) return BKCORE_ETOOLONG;
  if (B > 2147483647LL) return BKCORE_EINVAL;
  const size_t need = bkcore_backward_analytic_workspace_bytes(B, N);
  if (!workspace || workspace_bytes < need) return BKCORE_EWORKSPACE;
  char *ws = (char *)workspace;
  const size_t fwd_ws = (bkcore_workspace_bytes(B, N) + 255) & ~(size_t)255;
  const size_t plane = (size_t)B * (size_t)N * 2 * sizeof(float);
  float *Graw = (float *)(ws + fwd_ws);
  float *sr = (float *)(ws + fwd_ws + plane);
  float *ss = (float *)(ws + fwd_ws + 2 * plane);
  float *Pp = (float *)(ws + fwd_ws + 3 * plane);
  int *rowbad = (int *)(ws + fwd_ws + 4 * plane);
  hipStream_t s = (hipStream_t)stream;
  float *gk = N > 1 ? grad_b : nullptr;
  float *gc = N > 1 ? grad_c : nullptr;
  const unsigned fflags = BKCORE_FLAG_RAW | (flags & BKCORE_FLAG_PHI_EXACT);  // inputs are already sanitised
  const int *filter = nullptr, *gate = nullptr;
  if (state) {
    // 1. both sweeps in one launch, on the forward's own output
    const float *G = G_saved;
    if (!G) {
      const int rc = bkcore_forward(a, lda, b, ldb, c, ldc, BKCORE_F32, z_dev, z_re, z_im, B, N, Pp, nullptr, ws, fwd_ws,
                                    state, fflags, stream);
      if (rc != BKCORE_OK) return rc;
      G = Pp;
    }
    int *st = (int *)state;
#define BK_LAUNCH_SWEEPS(OFF)                                                                                         \
  hipLaunchKernelGGL((bk_adjoint_sweeps<OFF>), dim3((unsigned)B), dim3(T), 0, s, a, (long)lda, b, (long)ldb, c,      \
                     (long)ldc, z_dev, z_re, z_im, (long)N, G, grad_features, grad_G, grad_a, gk, gc, rowbad, st,      \
                     (unsigned long long *)sr)
    if (flags & BKCORE_FLAG_PHI_EXACT) {
      BK_LAUNCH_SWEEPS(0);
    } else {
      BK_LAUNCH_SWEEPS(1);
    }
#undef BK_LAUNCH_SWEEPS
    filter = rowbad;
    gate = st + 5;
  }
  // 2. the three-pass path (forward keeping the theta / phi ratios and G before its final clamps, then the two adjoint
  //    kernels on them): every row when there is no state block, otherwise only rows whose checks failed in a sweep --
  //    each launch returns at once unless state[5] is up
  const int rc = bkcore_internal_forward_save_rows(a, lda, b, ldb, c, ldc, z_dev, z_re, z_im, B, N, Graw, ws, fflags,
                                                   stream, sr, ss, filter, gate);
  if (rc != BKCORE_OK) return rc;
#define BK_LAUNCH_ADJ(OFF)                                                                                            \
  hipLaunchKernelGGL((bk_adjoint_phi_kernel<OFF>), dim3((unsigned)B), dim3(T), 0, s, a, (long)lda, b, (long)ldb, c,  \
                     (long)ldc, z_dev, z_re, z_im, (long)N, (const float *)Graw, grad_features, grad_G,               \
                     (const float *)ss, Pp, grad_a, gk, filter, gate);                                                \
  hipLaunchKernelGGL((bk_adjoint_theta_kernel<OFF>), dim3((unsigned)B), dim3(T), 0, s, a, (long)lda, b, (long)ldb,   \
                     c, (long)ldc, z_dev, z_re, z_im, (long)N, (const float *)sr, (const float *)Pp, grad_a, gk, gc,  \
                     filter, gate)
  if (flags & BKCORE_FLAG_PHI_EXACT) {
    BK_LAUNCH_ADJ(0);
  } else {
    BK_LAUNCH_ADJ(1);
  }
#undef BK_LAUNCH_ADJ
  if (state) hipLaunchKernelGGL(bk_adjoint_finish, dim3(1), dim3(64), 0, s, (int *)state);
  const hipError_t err = hipGetLastError();
  if (err != hipSuccess) {
    bkcore_internal_set_hip_error((int)err);
    return BKCORE_ELAUNCH;
  }
  return BKCORE_OK;
}

}  // extern "C"

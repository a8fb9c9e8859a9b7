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
// Two implementations in this file:
//   bk_adjoint_sweeps (second half of the file): one pass over the row that needs only a, b, c, the forward's output G
//     and the cotangent; the ratios are recomputed in the kernel; rows whose in-kernel checks fail are flagged.
//   the three-pass path (first half): a forward launch that keeps r, s and G_raw, then bk_adjoint_phi_kernel
//     (left -> right; P (prefix sum) and mu; writes P, the phi part of grad_a and of grad_k) and
//     bk_adjoint_theta_kernel (right -> left; lam; adds the theta parts and writes grad_a, grad_b, grad_c), one
//     256-thread workgroup per row, chunks of 256 x 8 elements, scans over affine maps of R^2 (2x2 matrix + vector)
//     with DPP inside a wave and LDS across waves.  It makes no assumption about contraction and runs for the flagged
//     rows, or for every row when the caller gives no state block.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/bkcore.h"
#include "bkcore_internal.h"

#ifndef BK_SWEEP_PRIO
// wave priority from the start of a chunk to the barrier behind its staging (what the workgroup's other waves wait
// for); measured at B = 1024, N = 32768: gradient of the diagonal alone 0.379 -> 0.366 ms, all three 0.455 -> 0.45 ms
#define BK_SWEEP_PRIO 3
#endif

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


// ==== one pass over the row, the ratios recomputed in it ===========================================================
// Nothing but G is kept from the forward (as in the reference, bk_core.py:293).  The cotangent of G_raw follows from G
// itself: with X = conj(gbar) G,  q = X - tanh^2(L_raw/88) Re X  and  tanh(L_raw/88) = ln|G| / 88  (the final clamp is
// ln|G| = 88 tanh(L_raw/88)); a clipped element (|G| = 50) passes only Im X; G = 0 passes nothing.  The ratios r and s
// are recomputed from a, b, c exactly as the forward does: wherever the ratio maps contract, a lane gets the ratio
// entering its 8 elements by running the map over the 16 elements before them (after them, for s), verified against
// the neighbouring lane's own value.  The adjoint recurrences contract at the same rate (their linear part is
// k w(r) / r), so the value entering a lane is taken from the two lanes before it (after it, for lam),
//     x_in(l) = v(l-1) + m(l-1) v(l-2),     v = the lane's 8 maps applied to 0,  m = the product of their linear parts,
// checked by |m(l-1) m(l-2)| <= 2e-5.  One workgroup per row walks left -> right over chunks of 256 lanes x 8 elements
// that overlap by two lanes: mu enters a chunk exactly (carried from the chunk before), lam enters its last lane as 0,
// which is exact at the end of the row and otherwise only reaches the two overlap lanes, whose results the next chunk
// produces instead.  So a, b, c, G and the cotangent are read once (coalesced, through LDS) and grad_a, grad_b,
// grad_c written once.  A row with a failed check is flagged (rowbad[row], state[5]) and recomputed by the three-pass
// path above, whose launches follow and skip every other row.
// development: BK_SWEEP_STAMPS=1 writes s_memtime stamps of thread 0 at the stage boundaries of every chunk into dbg
// (8 slots of 8 bytes per (row, chunk))
#ifndef BK_SWEEP_STAMPS
#define BK_SWEEP_STAMPS 0
#endif
#if BK_SWEEP_STAMPS
#define BK_SSTAMP(k)                                                                                                 \
  do {                                                                                                               \
    __builtin_amdgcn_sched_barrier(0);                                                                               \
    const unsigned long long tm_ = __builtin_amdgcn_s_memtime();                                                     \
    if (threadIdx.x == 0) dbg[((long)blockIdx.x * nch + cc) * 8 + (k)] = tm_;                                        \
    __builtin_amdgcn_sched_barrier(0);                                                                               \
  } while (0)
#else
#define BK_SSTAMP(k) do { } while (0)
#endif
#ifndef BK_SWEEP_PREFETCH
#define BK_SWEEP_PREFETCH 0   // 1: raw inputs of the next chunk held in registers during the current chunk's arithmetic
#endif
#ifndef BK_SWEEP_LDS_ABC
#define BK_SWEEP_LDS_ABC 1   // raw a, b, c of the chunk kept in LDS for the output pass (0: read again from memory)
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

constexpr int SWEEP_MAX_SEGMENTS = 64;              // workgroups per row at most (few long rows)
constexpr int HALO_LANES = 2;                       // lanes at the end of a chunk that the next chunk computes again
constexpr int STRIDE = CHUNK - HALO_LANES * EB;     // elements a chunk advances by
struct SharedSweep {
  float2 al_s[SPAD];            // regularised diagonal of elements base - WH .. base + CHUNK + WH - 1
  float k_s[SPAD];              // k_{i-1} = b_{i-1} c_{i-1} for the same elements (0 outside the row)
  float2 q_s[QPAD];             // q of elements base .. base + CHUNK - 1
#if BK_SWEEP_LDS_ABC
  float a_s[QPAD];              // he_diag of the same elements
  float2 bc_s[QPAD];            // (b_j, c_j) of the same elements
#endif
  float2 sumtot[2][WAVES];
  double2 sumcarry[2];          // sum of q over everything before the chunk: the one sum that runs over the whole row
                                // (float32 would lose 1e-7 sqrt(N) of a typical term: 1e-4 at N = 2^20)
  float2 wedge_t[2][WAVES];     // theta: lane 63 of each wave, om after its last own step
  float2 wedge_p[2][WAVES];     // phi: lane 63 of each wave, its warm-up estimate
  float2 edge_t[2], edge_p[2];  // the same of lane T - 3, the left neighbour of the next chunk's first lane
  float4 xe_p[2][WAVES][2];     // mu: (v, m) of lanes 62 and 63 of each wave
  float4 xe_t[2][WAVES][2];     // lam: (v, m) of lanes 0 and 1 of each wave
  float4 xc_p[2][2];            // mu: (v, m) of lanes T - 4 and T - 3
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
  float4 st;  // the lane's boundary states kept by the forward: om(r) entering its block, om(s) two elements past it
};

// CHECK = false when every staged element and its off-diagonal entries lie inside the row.  Addresses are a uniform
// pointer plus a small per-thread offset, so that no 64-bit address per load has to be kept in vector registers.
template <bool CHECK>
__device__ __forceinline__ void issue_loads(const float *__restrict__ a, const float *__restrict__ b,
                                            const float *__restrict__ c, const float *__restrict__ G,
                                            const float *__restrict__ gf, const float2 *__restrict__ bst, long base,
                                            long N, Raw &R, int t) {
  // boundary states: per row the forward keeps two planes of one float2 per block of 8 elements -- plane 0: om(r)
  // entering block m, plane 1: om(s) at the block's second element, i.e. two elements past block m - 1; the planes cover
  // whole chunks of 4096 elements plus one block
  R.st = make_float4(0.f, 0.f, 0.f, 0.f);
  if (bst && base + (long)t * EB < N) {
    const long m = base / EB + t;
    const long plane = ((N + 4095) / 4096) * 512 + 1;
    const float2 th = bst[m];
    const float2 ph = bst[plane + m + 1];
    R.st = make_float4(th.x, th.y, ph.x, ph.y);
  }
  const float *__restrict__ pa = a + (base - WH);       // slot q: a[base - WH + q]
  const float *__restrict__ pb = b + (base - WH - 1);   // slot q holds k_{i-1}, i = base - WH + q
  const float *__restrict__ pc = c + (base - WH - 1);
  const float *__restrict__ pG = G + 2 * base;
  const float *__restrict__ pg = gf ? gf + 2 * base : nullptr;
  // slots with an element inside the row: q in [qa0, qa1); with an off-diagonal entry: q in [qk0, qk1)
  const long qa0 = WH - base, qa1 = N - base + WH, qk0 = WH + 1 - base, qk1 = N - base + WH;
  const int ia0 = (int)(qa0 < 0 ? 0 : (qa0 > SN ? SN : qa0)), ia1 = (int)(qa1 > SN ? SN : (qa1 < 0 ? 0 : qa1));
  const int ik0 = (int)(qk0 < 0 ? 0 : (qk0 > SN ? SN : qk0)), ik1 = (int)(qk1 > SN ? SN : (qk1 < 0 ? 0 : qk1));
  const long ng = N - base;
  const int ig1 = (int)(ng > CHUNK ? CHUNK : (ng < 0 ? 0 : ng));
#pragma unroll
  for (int j = 0; j < ROUNDS; ++j) {
    const int q = t + j * T;
    R.av[j] = 1.0f;
    R.bv[j] = 0.0f;
    R.cv[j] = 0.0f;
    if (CHECK) {
      if (q >= ia0 && q < ia1) R.av[j] = __builtin_nontemporal_load(pa + q);
      if (q >= ik0 && q < ik1) {
        R.bv[j] = __builtin_nontemporal_load(pb + q);
        R.cv[j] = __builtin_nontemporal_load(pc + q);
      }
    } else if (j < ROUNDS - 1 || q < SN) {
      R.av[j] = __builtin_nontemporal_load(pa + q);
      R.bv[j] = __builtin_nontemporal_load(pb + q);
      R.cv[j] = __builtin_nontemporal_load(pc + q);
    }
  }
#pragma unroll
  for (int r = 0; r < EB; ++r) {
    const int e = t + r * T;
    R.Gv[r] = make_float2(0.f, 0.f);
    R.gv[r] = make_float2(0.f, 0.f);
    if (!CHECK || e < ig1) {
      R.Gv[r] = ld2(pG + 2 * e);
      if (pg) R.gv[r] = ld2(pg + 2 * e);
    }
  }
}
__device__ __forceinline__ void issue_chunk(const float *__restrict__ a, const float *__restrict__ b,
                                            const float *__restrict__ c, const float *__restrict__ G,
                                            const float *__restrict__ gf, const float2 *__restrict__ bst, long base,
                                            long N, Raw &R, int t) {
  if (base - WH - 1 >= 0 && base + CHUNK + WH - 1 <= N - 1) issue_loads<false>(a, b, c, G, gf, bst, base, N, R, t);
  else issue_loads<true>(a, b, c, G, gf, bst, base, N, R, t);
}

// Writes a chunk's staged form: regularised diagonal and off-diagonal products with their halo, q, and for the
// lanes' output pass the raw diagonal, b and c of the chunk's own elements.
__device__ __forceinline__ void commit_chunk(const Raw &R, const float *__restrict__ gG, long base, long N, float2 z,
                                             SharedSweep &sh, int t) {
  // Padded slot of q = t + j T is q + q / EB = (t + t / EB) + j (T + T / EB): one per-thread index plus compile-time
  // offsets (the compiler does not split the division by itself), and the slots of the chunk's own elements e = q - WH
  // and off-diagonal entries e - 1 are fixed distances from it.
  static_assert(T % EB == 0 && WH % EB == 0 && WH + 1 < T && CHUNK == (ROUNDS - 1) * T, "slot arithmetic");
  constexpr int PT = T + T / EB;
  const int st = t + t / EB;                           // slot of q (j = 0)
  const int sa = st - (WH + WH / EB);                  // slot of e = q - WH
  const int sb = t + (t + EB - 1) / EB - (WH + 1) - (WH / EB + 1);  // slot of e - 1 = q - WH - 1
#pragma unroll
  for (int j = 0; j < ROUNDS; ++j) {
    const int q = t + j * T;
    if (j < ROUNDS - 1 || q < SN) {
      sh.k_s[st + j * PT] = R.bv[j] * R.cv[j];
      sh.al_s[st + j * PT] = reg_diag(R.av[j], z);
#if BK_SWEEP_LDS_ABC
      // e = q - WH in [0, CHUNK): only the first round has slots before it, only the last one slots after it
      if ((j > 0 || t >= WH) && (j < ROUNDS - 1 || t < WH)) sh.a_s[sa + j * PT] = R.av[j];
      if ((j > 0 || t >= WH + 1) && (j < ROUNDS - 1 || t <= WH))  // e in [1, CHUNK]
        sh.bc_s[sb + j * PT] = make_float2(R.bv[j], R.cv[j]);
#endif
    }
  }
#pragma unroll
  for (int r = 0; r < EB; ++r) {
    const int e = t + r * T;
    float2 g = R.gv[r];
    if (gG && e < N - base) g = cadd(g, ld2(gG + 2 * base + 2 * e));
    // a cotangent that is not a number counts as 0, as in the reference's backward (bk_core.py:321-325)
    const float gx = fabsf(g.x) <= 3.0e38f ? g.x : 0.0f;
    const float gy = fabsf(g.y) <= 3.0e38f ? g.y : 0.0f;
    sh.q_s[st + r * PT] = cot_q_saved(R.Gv[r], make_float2(gx, gy));
  }
}

// One step of an adjoint recurrence:  x -> (c0 + y - tau Re y) inv,  y = kk w x,  w = inv wc
__device__ __forceinline__ float2 adj_step(float2 x, float2 cw, float2 inv, float tau, float2 c0) {
  const float2 y = cmul(cw, x);
  const float2 u = make_float2(fmaf(-tau, y.x, y.x) + c0.x, y.y + c0.y);
  return cmul(inv, u);
}

template <int CTRL>
__device__ __forceinline__ float2 shift2(float2 v) {  // lane shift by DPP, 0 where there is no source lane
  return make_float2(dpp_f<CTRL, 0xF>(0.0f, v.x), dpp_f<CTRL, 0xF>(0.0f, v.y));
}

// SAVED: the ratio chains start from the boundary states the forward kept (bst_: per block of EB elements the om(r)
// entering it and the om(s) two elements past it) instead of running a 16-step warm-up on either side; the checks
// between neighbouring lanes stay, now comparing the forward's value with the one recomputed from the block before.
template <int OFF, bool SAVED>  // kphi_m = k_{m-OFF}
__global__ __launch_bounds__(T, BK_SWEEP_WG) void bk_adjoint_sweeps(
    const float *__restrict__ a_, long lda, const float *__restrict__ b_, long ldb, const float *__restrict__ c_,
    long ldc, const float *__restrict__ z_dev, float z_re, float z_im, long N, const float *__restrict__ G_,
    const float *__restrict__ gf_, const float *__restrict__ gG_, float *__restrict__ ga_, float *__restrict__ gb_,
    float *__restrict__ gc_, int *__restrict__ rowbad, int *state, unsigned long long *dbg, int S, long seg_len,
    const double2 *__restrict__ segsum, const float4 *__restrict__ bst_) {
  __shared__ SharedSweep sh;
  static_assert(EB == 8, "the forward keeps its boundary states per block of 8 elements");
  // Few long rows: a row is cut into S segments of seg_len elements, one workgroup each.  A
  // segment after the first starts two lanes early: mu enters those lanes as 0, which reaches only them (their results
  // belong to the segment before), exactly as lam does at the end of every chunk; the sum of q over everything before
  // the segment comes from bk_adjoint_segsum.
  const long row = S == 1 ? (long)blockIdx.x : (long)(blockIdx.x / S);
  const int sg = S == 1 ? 0 : (int)(blockIdx.x % S);
  const long out_lo = (long)sg * seg_len, out_hi = sg == S - 1 ? N : min(N, out_lo + seg_len);
  const long start = out_lo > 0 ? out_lo - HALO_LANES * EB : 0;
  const float *a = a_ + row * lda;
  const float *b = N >= 2 ? b_ + row * ldb : a, *c = N >= 2 ? c_ + row * ldc : a;
  const float *G = G_ + row * N * 2;
  const float *gf = gf_ ? gf_ + row * N * 2 : nullptr, *gG = gG_ ? gG_ + row * N * 2 : nullptr;
  float *ga = ga_ + row * N;
  float *gb = gb_ ? gb_ + row * (N - 1) : nullptr;
  float *gc = gc_ ? gc_ + row * (N - 1) : nullptr;
  const float2 z = z_dev ? make_float2(z_dev[0], z_dev[1]) : make_float2(z_re, z_im);
  const float2 *bst = SAVED ? reinterpret_cast<const float2 *>(bst_) + row * (2 * (((N + 4095) / 4096) * 512 + 1)) : nullptr;
  const int t = threadIdx.x;
  const int nch = N <= CHUNK ? 1 : 1 + (int)((N - CHUNK + STRIDE - 1) / STRIDE);  // (index of the debug stamps)
  (void)nch;
  bool bad = false;
  if (t == 0) {
    double2 before = make_double2(0.0, 0.0);
    for (int q = 0; q < sg; ++q) {
      const double2 v = segsum[row * S + q];
      before.x += v.x;
      before.y += v.y;
    }
    sh.sumcarry[0] = before;
    sh.xc_p[1][0] = make_float4(0.f, 0.f, 0.f, 0.f);
    sh.xc_p[1][1] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  Raw R;
  if (BK_SWEEP_PREFETCH) issue_chunk(a, b, c, G, gf, bst, start, N, R, t);
  bool more = true;
  for (int cc = 0; more; ++cc) {
    const int buf = cc & 1;
    const long base = start + (long)cc * STRIDE;
    const bool reaches_end = base + CHUNK >= N;                   // the chunk holds the end of the row
    more = !reaches_end && base + STRIDE < out_hi;
    const bool tail = base + CHUNK + WH >= N - 1;  // the end of the row lies in what this chunk stages
    BK_SSTAMP(0);
    // (the thread index goes through an opaque copy at each stage: offsets derived from it are then recomputed where
    // they are used instead of being kept in registers across the whole loop)
    int ts = t;
    asm volatile("" : "+v"(ts));
#if BK_SWEEP_PRIO
    __builtin_amdgcn_s_setprio(BK_SWEEP_PRIO);
#endif
    if (!BK_SWEEP_PREFETCH) issue_chunk(a, b, c, G, gf, bst, base, N, R, ts);
    __syncthreads();  // every lane finished reading the staging area
    commit_chunk(R, gG, base, N, z, sh, ts);
    __syncthreads();
#if BK_SWEEP_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
    BK_SSTAMP(1);
    __builtin_amdgcn_sched_barrier(0);
    int tl = t;
    asm volatile("" : "+v"(tl));
    const long first = base + (long)tl * EB;
    const float2 *al_l = sh.al_s + (WH + tl * EB) + (WH + tl * EB) / EB;
    const float *k_l = sh.k_s + (WH + tl * EB) + (WH + tl * EB) / EB;
    const float2 *q_l = sh.q_s + tl * (EB + 1);
    // What the lane's chains read from the staging area is loaded in batches, each one issued before the arithmetic on
    // the batch before it.  Ratio chains: warm-up over the WH elements before the lane's own in the direction of the
    // recurrence (theta: to the left; phi: to the right, where past the end of the row the staged values give om * 0),
    // then the lane's own.  kphi_d = k_{d-OFF} sits in slot d + 1 - OFF and is 0 from the last element on.
    float2 alo[EB];
    float kx[EB + 2];  // kx[x] = slot x - 1
    float2 omt = make_float2(0.f, 0.f), omp = make_float2(0.f, 0.f);
    float2 inv8 = make_float2(0.f, 0.f), om9 = make_float2(0.f, 0.f);  // phi at elements 8 and 9 of the lane
    float wc8 = 0.0f, tau8 = 0.0f;
    if constexpr (SAVED) {
      const float4 sv = R.st;
#pragma unroll
      for (int e = 0; e < EB; ++e) alo[e] = al_l[soff(e)];
#pragma unroll
      for (int x = 0; x < EB + 2; ++x) kx[x] = k_l[soff(x - 1)];
      const float2 al8 = al_l[soff(EB)];
      float k8 = k_l[soff(EB + 1 - OFF)];
      if (tail && first + EB >= N - 1) k8 = 0.0f;
      if (first > 0) omt = make_float2(sv.x, sv.y);                    // theta enters a row with om = 0
      if (first + EB + 1 <= N - 1) om9 = make_float2(sv.z, sv.w);      // phi likewise, from the other end
      float lv = 0.0f;
      omp = om_full(al8, k8, om9, inv8, wc8, tau8, lv);
    } else {
    float2 alt[WH], alp[WH];
    float kt[WH], kp[WH];
#pragma unroll
    for (int j = 0; j < WH / 2; ++j) {
      alt[j] = al_l[soff(j - WH)];
      kt[j] = k_l[soff(j - WH)];
      alp[j] = al_l[soff(EB + WH - 1 - j)];
      kp[j] = k_l[soff(EB + WH - j - OFF)];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = WH / 2; j < WH; ++j) {
      alt[j] = al_l[soff(j - WH)];
      kt[j] = k_l[soff(j - WH)];
      alp[j] = al_l[soff(EB + WH - 1 - j)];
      kp[j] = k_l[soff(EB + WH - j - OFF)];
    }
    if (tail) {
#pragma unroll
      for (int j = 0; j < WH / 2; ++j)
        if (first + (EB + WH - 1 - j) >= N - 1) kp[j] = 0.0f;
    }
#pragma unroll
    for (int j = 0; j < WH / 2; ++j) {
      omt = om_coarse(alt[j], kt[j], omt);
      omp = om_coarse(alp[j], kp[j], omp);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int e = 0; e < EB; ++e) alo[e] = al_l[soff(e)];
#pragma unroll
    for (int x = 0; x < EB + 2; ++x) kx[x] = k_l[soff(x - 1)];
    if (tail) {
#pragma unroll
      for (int j = WH / 2; j < WH; ++j)
        if (first + (EB + WH - 1 - j) >= N - 1) kp[j] = 0.0f;
    }
#pragma unroll
    for (int j = WH / 2; j < WH; ++j) {
      if (j < 3 * WH / 4) {
        omt = om_short(alt[j], kt[j], omt);
        omp = om_short(alp[j], kp[j], omp);
      } else {
        float2 iv;
        float wv, tv, lv = 0.0f;
        omt = om_full(alt[j], kt[j], omt, iv, wv, tv, lv);
        if (j == WH - 1) om9 = omp;
        omp = om_full(alp[j], kp[j], omp, inv8, wc8, tau8, lv);
      }
    }
    }
    __builtin_amdgcn_sched_barrier(0);
    float2 q[EB];
#pragma unroll
    for (int e = 0; e < EB; ++e) q[e] = q_l[e];
    const float2 west_t = omt, west_p = omp;
    float2 invt[EB], invp[EB];
    float wct[EB], taut[EB], wcp[EB], taup[EB];
    float lmin = 0.0f;
#pragma unroll
    for (int j = 0; j < EB; ++j) {
      const int et = j, ep = EB - 1 - j;
      float kpe = kx[ep + 2 - OFF];
      if (tail && first + ep >= N - 1) kpe = 0.0f;
      omt = om_full(alo[et], kx[et + 1], omt, invt[et], wct[et], taut[et], lmin);
      omp = om_full(alo[ep], kpe, omp, invp[ep], wcp[ep], taup[ep], lmin);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (tail) {  // elements past the end of the row map everything to 0 (lam = 0 enters the last element)
#pragma unroll
      for (int e = 0; e < EB; ++e)
        if (first + e >= N) {
          invt[e] = make_float2(0.f, 0.f);
          invp[e] = make_float2(0.f, 0.f);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    // exclusive prefix sums of q inside the lane, then over the lanes
    float2 pre[EB + 1];
    pre[0] = make_float2(0.f, 0.f);
#pragma unroll
    for (int e = 0; e < EB; ++e) pre[e + 1] = cadd(pre[e], q[e]);
    float2 incl = pre[EB];
    incl = cadd(incl, sum_dpp<0x111, 0xF>(incl));
    incl = cadd(incl, sum_dpp<0x112, 0xF>(incl));
    incl = cadd(incl, sum_dpp<0x114, 0xF>(incl));
    incl = cadd(incl, sum_dpp<0x118, 0xF>(incl));
    incl = cadd(incl, sum_dpp<0x142, 0xA>(incl));
    incl = cadd(incl, sum_dpp<0x143, 0xC>(incl));
    __builtin_amdgcn_sched_barrier(0);
    const int lane = tl & 63, wave = tl >> 6;
    if (lane == 63) {
      sh.sumtot[buf][wave] = incl;
      sh.wedge_t[buf][wave] = omt;
      sh.wedge_p[buf][wave] = west_p;
    }
    if (tl == T - 1 - HALO_LANES) {
      sh.edge_t[buf] = omt;
      sh.edge_p[buf] = west_p;
    }
    float2 run = shift2<0x138>(incl);  // wave_shr:1
    float2 nb_t = shift2<0x138>(omt), nb_p = shift2<0x138>(west_p);
    BK_SSTAMP(2);
    __syncthreads();
    BK_SSTAMP(3);
    {
      const double2 carry = sh.sumcarry[buf];
      float2 x = make_float2(0.f, 0.f);  // sum over the chunk's lanes before this one
#pragma unroll
      for (int w = 0; w < WAVES; ++w) {
        if (w == wave) run = cadd(run, x);
        x = cadd(x, sh.sumtot[buf][w]);
      }
      if (tl == T - 1 - HALO_LANES) {  // sum up to the next chunk's start
        const float2 upto = cadd(run, pre[EB]);
        sh.sumcarry[buf ^ 1] = make_double2(carry.x + (double)upto.x, carry.y + (double)upto.y);
      }
      run = make_float2((float)(carry.x + (double)run.x), (float)(carry.y + (double)run.y));
      // the left neighbour's last own step (theta) / warm-up (phi) and this lane's warm-up (theta) / first own step
      // (phi) are two evaluations of the same om
      if (lane == 0) {
        nb_t = wave > 0 ? sh.wedge_t[buf][wave - 1] : sh.edge_t[buf ^ 1];
        nb_p = wave > 0 ? sh.wedge_p[buf][wave - 1] : sh.edge_p[buf ^ 1];
      }
      if ((tl > 0 || cc > 0) && first < N) bad |= om_differs(west_t, nb_t) || om_differs(nb_p, omp);
      if (first < N) bad |= !(lmin >= -13.3f);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int e = 0; e <= EB; ++e) pre[e] = cadd(pre[e], run);  // pre[e] = P_{first + e - 1}
    if (BK_SWEEP_PREFETCH && more) issue_chunk(a, b, c, G, gf, bst, base + STRIDE, N, R, tl);
    __builtin_amdgcn_sched_barrier(0);
    // the lane's maps applied to 0 and the product of their linear parts, for both adjoints:
    //   mu  (left -> right): c0 = P_{j-1},  off-diagonal product kphi_{j-1} = slot e - OFF
    //   lam (right -> left): c0 = -P_j,     off-diagonal product k_j        = slot e + 1
    float2 vp = make_float2(0.f, 0.f), mp = make_float2(1.f, 0.f);
    float2 vt = make_float2(0.f, 0.f), mt = make_float2(1.f, 0.f);
#pragma unroll
    for (int j = 0; j < EB; ++j) {
      const int ep = j, et = EB - 1 - j;
      const float gp = kx[ep + 1 - OFF] * wcp[ep], gt = kx[et + 2] * wct[et];
      const float2 cwp = make_float2(invp[ep].x * gp, invp[ep].y * gp);
      const float2 cwt = make_float2(invt[et].x * gt, invt[et].y * gt);
      vp = adj_step(vp, cwp, invp[ep], taup[ep], pre[ep]);
      vt = adj_step(vt, cwt, invt[et], taut[et], make_float2(-pre[et + 1].x, -pre[et + 1].y));
      mp = cmul(mp, cmul(invp[ep], cwp));
      mt = cmul(mt, cmul(invt[et], cwt));
    }
    __builtin_amdgcn_sched_barrier(0);
    {
      int tv = t;
      asm volatile("" : "+v"(tv));
      const int lv = tv & 63, wv = tv >> 6;
      if (lv >= 62) sh.xe_p[buf][wv][lv - 62] = make_float4(vp.x, vp.y, mp.x, mp.y);
      if (lv <= 1) sh.xe_t[buf][wv][lv] = make_float4(vt.x, vt.y, mt.x, mt.y);
      if (tv >= T - 2 - HALO_LANES && tv < T - HALO_LANES)
        sh.xc_p[buf][tv - (T - 2 - HALO_LANES)] = make_float4(vp.x, vp.y, mp.x, mp.y);
    }
    float2 vp1 = shift2<0x138>(vp), mp1 = shift2<0x138>(mp);  // wave_shr:1
    float2 vt1 = shift2<0x130>(vt), mt1 = shift2<0x130>(mt);  // wave_shl:1
    BK_SSTAMP(4);
    __syncthreads();
    BK_SSTAMP(5);
    int to = t;
    asm volatile("" : "+v"(to));
    const int lo = to & 63, wo = to >> 6;
    if (lo == 0) {
      const float4 s1 = wo > 0 ? sh.xe_p[buf][wo - 1][1] : sh.xc_p[buf ^ 1][1];
      vp1 = make_float2(s1.x, s1.y);
      mp1 = make_float2(s1.z, s1.w);
    }
    if (lo == 63) {
      const float4 s1 = wo < WAVES - 1 ? sh.xe_t[buf][wo < WAVES - 1 ? wo + 1 : 0][0]
                                         : make_float4(0.f, 0.f, 0.f, 0.f);
      vt1 = make_float2(s1.x, s1.y);
      mt1 = make_float2(s1.z, s1.w);
    }
    float2 vp2 = shift2<0x138>(vp1), mp2 = shift2<0x138>(mp1);
    float2 vt2 = shift2<0x130>(vt1), mt2 = shift2<0x130>(mt1);
    if (lo == 0) {
      const float4 s2 = wo > 0 ? sh.xe_p[buf][wo - 1][0] : sh.xc_p[buf ^ 1][0];
      vp2 = make_float2(s2.x, s2.y);
      mp2 = make_float2(s2.z, s2.w);
    }
    if (lo == 63) {
      const float4 s2 = wo < WAVES - 1 ? sh.xe_t[buf][wo < WAVES - 1 ? wo + 1 : 0][1]
                                         : make_float4(0.f, 0.f, 0.f, 0.f);
      vt2 = make_float2(s2.x, s2.y);
      mt2 = make_float2(s2.z, s2.w);
    }
    __builtin_amdgcn_sched_barrier(0);
    float2 xp = cadd(vp1, cmul(mp1, vp2));  // mu_{first-1} / s_{first-1}
    float2 xt = cadd(vt1, cmul(mt1, vt2));  // lam_{first+8} / r_{first+8}
    bad |= !(fmaf(mp1.x, mp1.x, mp1.y * mp1.y) * fmaf(mp2.x, mp2.x, mp2.y * mp2.y) <= 4.0e-10f);
    bad |= !(fmaf(mt1.x, mt1.x, mt1.y * mt1.y) * fmaf(mt2.x, mt2.x, mt2.y * mt2.y) <= 4.0e-10f);
    __builtin_amdgcn_sched_barrier(0);
    // output pass: xs = lam_j / r_j + mu_j / s_j;  gk[e] = gradient with respect to k_{first+e}: the theta part comes
    // from element e, the phi part from element e + 1 + OFF (elements 8 and 9 lie in the next lane's block: one or
    // two more steps of mu through the values the phi warm-up ended on)
    float2 xs[EB];
    float gk[EB];
#pragma unroll
    for (int e = 0; e < EB; ++e) {
      xs[e] = make_float2(0.f, 0.f);
      gk[e] = 0.0f;
    }
#pragma unroll
    for (int j = 0; j < EB; ++j) {
      const int ep = j, et = EB - 1 - j;
      const float gp = kx[ep + 1 - OFF] * wcp[ep], gt = kx[et + 2] * wct[et];
      const float2 wp = make_float2(invp[ep].x * wcp[ep], invp[ep].y * wcp[ep]);
      const float2 wt = make_float2(invt[et].x * wct[et], invt[et].y * wct[et]);
      const float2 cwp = make_float2(invp[ep].x * gp, invp[ep].y * gp);
      const float2 cwt = make_float2(invt[et].x * gt, invt[et].y * gt);
      if (ep - 1 - OFF >= 0) gk[ep - 1 - OFF >= 0 ? ep - 1 - OFF : 0] -= fmaf(xp.x, wp.x, -xp.y * wp.y);
      gk[et] -= fmaf(xt.x, wt.x, -xt.y * wt.y);
      xp = adj_step(xp, cwp, invp[ep], taup[ep], pre[ep]);
      xt = adj_step(xt, cwt, invt[et], taut[et], make_float2(-pre[et + 1].x, -pre[et + 1].y));
      xs[ep] = cadd(xs[ep], xp);
      xs[et] = cadd(xs[et], xt);
    }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" : "+v"(to));
    float ao[EB];
    float2 bco[EB];
#if BK_SWEEP_LDS_ABC
    const float *a_l = sh.a_s + to * (EB + 1);
    const float2 *bc_l = sh.bc_s + to * (EB + 1);
#pragma unroll
    for (int e = 0; e < EB; ++e) {
      ao[e] = a_l[e];
      bco[e] = bc_l[e];
    }
#else
    {  // read again from memory (the chunk was staged from there a moment ago)
      float bo[EB], co[EB];
      load_block<EB>(a + base, to * EB, 0, N - base, 0.0f, ao);
      load_block<EB>(b + base, to * EB, 0, N - 1 - base, 0.0f, bo);
      load_block<EB>(c + base, to * EB, 0, N - 1 - base, 0.0f, co);
#pragma unroll
      for (int e = 0; e < EB; ++e) bco[e] = make_float2(bo[e], co[e]);
    }
#endif
    {
      // element 8: w(s_8) is what the phi warm-up ended on; element 9 (OFF = 1): one more step of mu through 8
      const float2 w8 = make_float2(inv8.x * wc8, inv8.y * wc8);
      float g8 = -fmaf(xp.x, w8.x, -xp.y * w8.y);
      if (!(first + EB - 1 - OFF < N - 1 - OFF)) g8 = 0.0f;
      gk[EB - 1 - OFF] += g8;
      if (OFF == 1) {
        const float g = kx[EB + 1 - OFF] * wc8;
        const float2 cw = make_float2(inv8.x * g, inv8.y * g);
        xp = adj_step(xp, cw, inv8, tau8, pre[EB]);
        float g9 = -fmaf(xp.x, om9.x, -xp.y * om9.y);
        if (!(first + EB - 1 < N - 1 - OFF)) g9 = 0.0f;
        gk[EB - 1] += g9;
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    float gav[EB], gbv[EB], gcv[EB];
#pragma unroll
    for (int e = 0; e < EB; ++e) {
      gav[e] = finite_or_zero(times_dreg(xs[e], ao[e], z));
      gbv[e] = finite_or_zero(bco[e].y * gk[e]);
      gcv[e] = finite_or_zero(bco[e].x * gk[e]);
    }
    BK_SSTAMP(6);
    const long first_o = base + (long)to * EB;
    if ((reaches_end || to < T - HALO_LANES) && first_o >= out_lo && first_o < out_hi) {
      store_block<EB>(ga + base, to * EB, 0, N - base, gav);
      if (gb) {
        store_block<EB>(gb + base, to * EB, 0, N - 1 - base, gbv);
        store_block<EB>(gc + base, to * EB, 0, N - 1 - base, gcv);
      }
    }
    BK_SSTAMP(7);
  }
  const int any_bad = __syncthreads_or(bad ? 1 : 0);
  if (threadIdx.x == 0) {
    if (S == 1) rowbad[row] = any_bad;                            // (with segments bk_adjoint_segsum zeroed it)
    else if (any_bad) __hip_atomic_store(&rowbad[row], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (any_bad) __hip_atomic_store(&state[5], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// Sum of q over each segment's own stretch [start_s, start_{s+1}) of the row (start_s = where the segment's first chunk
// begins), for the prefix sums of bk_adjoint_sweeps run with several workgroups per row.  Also clears the row's flag.
__global__ __launch_bounds__(T) void bk_adjoint_segsum(const float *__restrict__ G_, const float *__restrict__ gf_,
                                                       const float *__restrict__ gG_, long N, int S, long seg_len,
                                                       double2 *__restrict__ segsum, int *__restrict__ rowbad) {
  __shared__ double2 part[T];
  const long row = blockIdx.x / S;
  const int sg = (int)(blockIdx.x % S);
  const float *G = G_ + row * N * 2;
  const float *gf = gf_ ? gf_ + row * N * 2 : nullptr, *gG = gG_ ? gG_ + row * N * 2 : nullptr;
  const long lo = sg == 0 ? 0 : (long)sg * seg_len - HALO_LANES * EB;
  const long hi = sg == S - 1 ? N : min(N, (long)(sg + 1) * seg_len - HALO_LANES * EB);
  float2 acc = make_float2(0.f, 0.f);
  double2 tot = make_double2(0.0, 0.0);
  int n = 0;
  for (long i = lo + threadIdx.x; i < hi; i += T) {
    float2 g = gf ? ld2(gf + 2 * i) : make_float2(0.f, 0.f);
    if (gG) g = cadd(g, ld2(gG + 2 * i));
    const float gx = fabsf(g.x) <= 3.0e38f ? g.x : 0.0f;
    const float gy = fabsf(g.y) <= 3.0e38f ? g.y : 0.0f;
    acc = cadd(acc, cot_q_saved(ld2(G + 2 * i), make_float2(gx, gy)));
    if (++n == 64) {  // short float32 runs, collected in double
      tot.x += acc.x;
      tot.y += acc.y;
      acc = make_float2(0.f, 0.f);
      n = 0;
    }
  }
  tot.x += acc.x;
  tot.y += acc.y;
  part[threadIdx.x] = tot;
  __syncthreads();
  for (int w = T / 2; w >= 1; w >>= 1) {
    if ((int)threadIdx.x < w) {
      part[threadIdx.x].x += part[threadIdx.x + w].x;
      part[threadIdx.x].y += part[threadIdx.x + w].y;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    segsum[row * S + sg] = part[0];
    if (sg == 0) rowbad[row] = 0;
  }
}

__global__ void bk_adjoint_finish(int *state) {
  if (threadIdx.x == 0) __hip_atomic_store(&state[5], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

}  // namespace

extern "C" {

// Segments per row for bk_adjoint_sweeps: rows are cut when there are too few of them to fill the GPU (two workgroups
// per CU).  Measured on MI355X at N = 32768 (one workgroup per row -> segments): 8 rows 0.090 -> 0.039 ms, 64 rows
// 0.114 -> 0.056 ms, 128 rows 0.119 -> 0.082 ms, 256 rows 0.162 -> 0.143 ms; from 384 rows on whole rows are ahead (the
// segments' pre-pass reads G and the cotangent once more).
static void sweep_plan(int64_t B, int64_t N, int &S, long &seg_len) {
  S = 1;
  seg_len = 0;
  const long chunks = (long)((N + STRIDE - 1) / STRIDE);
  constexpr long ROWS_MAX = 256, MIN_CHUNKS = 2;   // chunks per segment at least
  if (B > ROWS_MAX || chunks < 2 * MIN_CHUNKS) return;
  long want = (1024 + B - 1) / B;
  if (want > chunks / MIN_CHUNKS) want = chunks / MIN_CHUNKS;
  if (want > SWEEP_MAX_SEGMENTS) want = SWEEP_MAX_SEGMENTS;
  if (want < 2) return;
  // m chunks per segment: the first one of a later segment starts two lanes early, so m chunks cover m * STRIDE
  // minus those two lanes
  seg_len = (chunks + want - 1) / want * STRIDE - HALO_LANES * EB;
  S = (int)((N + seg_len - 1) / seg_len);
  if (S < 2) { S = 1; seg_len = 0; }
}

size_t bkcore_backward_analytic_workspace_bytes(int64_t B, int64_t N) {
  if (B <= 0 || N <= 0) return 0;
  // forward scratch + four complex64 planes (G_raw, r, s, P of the three-pass path; the P plane also takes G when the
  // caller did not keep it) + one flag per row
  return ((bkcore_workspace_bytes(B, N) + 255) & ~(size_t)255) + 4 * (size_t)B * (size_t)N * 2 * sizeof(float) +
         (((size_t)B * sizeof(int) + 255) & ~(size_t)255) + (size_t)B * SWEEP_MAX_SEGMENTS * sizeof(double2);
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
  return bkcore_backward_analytic_states(a, lda, b, ldb, c, ldc, z_dev, z_re, z_im, B, N, G_saved, nullptr, 0,
                                         grad_features, grad_G, grad_a, grad_b, grad_c, workspace, workspace_bytes,
                                         state, flags, stream);
}

int bkcore_backward_analytic_states(const float *a, int64_t lda, const float *b, int64_t ldb, const float *c,
                                    int64_t ldc, const float *z_dev, float z_re, float z_im, int64_t B, int64_t N,
                                    const float *G_saved, const float *boundary_states, size_t boundary_state_bytes,
                                    const float *grad_features, const float *grad_G, float *grad_a, float *grad_b,
                                    float *grad_c, void *workspace, size_t workspace_bytes, void *state, unsigned flags,
                                    void *stream) {
  if (boundary_states && !G_saved) return BKCORE_EINVAL;  // the states belong to the forward call that produced G
  if (boundary_states && B > 0 && N > 0 && boundary_state_bytes < bkcore_boundary_state_bytes(B, N))
    return BKCORE_EWORKSPACE;
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
  double2 *segsum = (double2 *)(ws + fwd_ws + 4 * plane + (((size_t)B * sizeof(int) + 255) & ~(size_t)255));
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
    int S = 1;
    long seg_len = 0;
    sweep_plan(B, N, S, seg_len);
    if (S > 1)
      hipLaunchKernelGGL(bk_adjoint_segsum, dim3((unsigned)(B * S)), dim3(T), 0, s, G, grad_features, grad_G, (long)N,
                         S, seg_len, segsum, rowbad);
#define BK_LAUNCH_SWEEPS(OFF, SAVEDV)                                                                                 \
  hipLaunchKernelGGL((bk_adjoint_sweeps<OFF, SAVEDV>), dim3((unsigned)(B * S)), dim3(T), 0, s, a, (long)lda, b,      \
                     (long)ldb, c, (long)ldc, z_dev, z_re, z_im, (long)N, G, grad_features, grad_G, grad_a, gk, gc,   \
                     rowbad, st, (unsigned long long *)sr, S, seg_len, (const double2 *)segsum,                       \
                     (const float4 *)boundary_states)
    if (flags & BKCORE_FLAG_PHI_EXACT) {
      if (boundary_states) BK_LAUNCH_SWEEPS(0, true);
      else BK_LAUNCH_SWEEPS(0, false);
    } else {
      if (boundary_states) BK_LAUNCH_SWEEPS(1, true);
      else BK_LAUNCH_SWEEPS(1, false);
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

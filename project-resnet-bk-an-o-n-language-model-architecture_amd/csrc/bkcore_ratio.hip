// BK-Core forward in ratio form (first launch of bkcore_forward) for MI355X (gfx950).
#include "bkcore_device.h"
#include "bkcore_internal.h"

namespace {


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

}  // namespace

namespace {
template <typename TIn, bool PHI_EXACT>
void ratio_launch(int mode, int save, const void *a, long lda, const void *b, long ldb, const void *c, long ldc,
                  const float *z_dev, float z_re, float z_im, long B, long N, float *G, float *features, void *ws,
                  void *state, int call_id, unsigned flags, float fclamp, int S, int seg_ch, float *save_r,
                  float *save_s, hipStream_t stream) {
#define BK_RATIO(MODE, SAVE, GRID)                                                                                   \
  hipLaunchKernelGGL((bk_ratio_forward_kernel<TIn, PHI_EXACT, MODE, SAVE>), dim3((unsigned)(GRID)), dim3(T), 0,      \
                     stream, (const TIn *)a, lda, (const TIn *)b, ldb, (const TIn *)c, ldc, z_dev, z_re, z_im, B, N, \
                     (float2 *)G, (float2 *)features, (char *)ws, (int *)state, call_id, flags, fclamp, S, seg_ch,   \
                     (float2 *)save_r, (float2 *)save_s)
  if (mode == 1) BK_RATIO(1, false, B * S);
  else if (mode == 2) BK_RATIO(2, false, B * S);
  else if (!save) BK_RATIO(0, false, B);
  else if (std::is_same<TIn, float>::value) BK_RATIO(0, true, B);  // the saving variant exists for float32 operands
#undef BK_RATIO
}
}  // namespace

extern "C" void bkcore_internal_ratio_forward(int in_dtype, int phi_exact, int mode, int save, const void *a,
                                              int64_t lda, const void *b, int64_t ldb, const void *c, int64_t ldc,
                                              const float *z_dev, float z_re, float z_im, int64_t B, int64_t N, float *G,
                                              float *features, void *ws, void *state, int call_id, unsigned flags,
                                              float fclamp, int S, int seg_ch, float *save_r, float *save_s,
                                              void *stream) {
#define BK_DISPATCH(TIN)                                                                                             \
  do {                                                                                                               \
    if (phi_exact)                                                                                                   \
      ratio_launch<TIN, true>(mode, save, a, (long)lda, b, (long)ldb, c, (long)ldc, z_dev, z_re, z_im, (long)B,      \
                              (long)N, G, features, ws, state, call_id, flags, fclamp, S, seg_ch, save_r, save_s,    \
                              (hipStream_t)stream);                                                                  \
    else                                                                                                             \
      ratio_launch<TIN, false>(mode, save, a, (long)lda, b, (long)ldb, c, (long)ldc, z_dev, z_re, z_im, (long)B,     \
                               (long)N, G, features, ws, state, call_id, flags, fclamp, S, seg_ch, save_r, save_s,   \
                               (hipStream_t)stream);                                                                 \
  } while (0)
  switch (in_dtype) {
    case BKCORE_F32: BK_DISPATCH(float); break;
#ifndef BK_DEV_F32_ONLY
    case BKCORE_BF16: BK_DISPATCH(uint16_t); break;
    case BKCORE_F16: BK_DISPATCH(_Float16); break;
    case BKCORE_F64: BK_DISPATCH(double); break;
#endif
    default: break;
  }
#undef BK_DISPATCH
}

extern "C" void bkcore_internal_ratio_combine(int64_t B, int64_t N, void *ws, void *state, int call_id, int S,
                                              void *stream) {
  hipLaunchKernelGGL(bk_ratio_combine_kernel, dim3((unsigned)B), dim3(64), 0, (hipStream_t)stream, (long)B, (long)N,
                     (char *)ws, (int *)state, call_id, S);
}

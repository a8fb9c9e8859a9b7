// BK-Core forward in ratio form (first launch of bkcore_forward) for MI355X (gfx950).
#include "bkcore_device.h"
#include "bkcore_internal.h"

// The arithmetic type of the chains and of the product, fixed per translation unit: float here; bkcore_ratio_long.hip
// compiles this file again with double (BK_RATIO_REAL, BK_RATIO_LONG) for rows so long that the float32 rounding of
// N ratios, a random walk of about 8.5e-8 sqrt(N) in the product, would use up the 1e-4 tolerance (3e-5 at N = 2^17,
// 9e-5 at N = 2^20).
#ifndef BK_RATIO_REAL
#define BK_RATIO_REAL float
#define BK_RATIO_LONG 0
#endif

namespace {

typedef BK_RATIO_REAL Real;
constexpr bool LONG = BK_RATIO_LONG != 0;
typedef std::conditional<LONG, double2, float2>::type C2;
__device__ __forceinline__ C2 c2(Real x, Real y) { C2 v; v.x = x; v.y = y; return v; }
__device__ __forceinline__ C2 to_c2(float2 p) { return c2((Real)p.x, (Real)p.y); }
__device__ __forceinline__ float2 to_f2(C2 p) { return make_float2((float)p.x, (float)p.y); }
__device__ __forceinline__ float rfma(float a, float b, float c) { return fmaf(a, b, c); }
__device__ __forceinline__ double rfma(double a, double b, double c) { return fma(a, b, c); }
__device__ __forceinline__ C2 cmulr(C2 p, C2 q) {
  return c2(rfma(p.x, q.x, -p.y * q.y), rfma(p.x, q.y, p.y * q.x));
}

// Tile of the ratio-form kernel: RT threads per workgroup, E elements per thread per chunk.  The workspace layout
// (rowflag_bytes, lane_state_bytes) stays that of the scan kernel's tile.
#ifndef BK_RATIO_T
#define BK_RATIO_T 256
#endif
constexpr int RT = BK_RATIO_T;
constexpr int RCHUNK = RT * E;
constexpr int RWAVES = RT / 64;
constexpr int RO_PAD = RCHUNK + RCHUNK / 16 + 8;
static_assert(RT % 64 == 0 && RT >= 64 && RT <= 256, "whole waves, at most the scan kernel's workgroup");


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
#define BK_RATIO_WAVES 3
#endif
// Scheduling fence every BK_SCHED_GROUP chain steps (0: none): keeps the compiler from hoisting the LDS reads of all
// 64 steps of a chunk to the front, which costs more registers than 3 waves per SIMD leave.
#ifndef BK_SCHED_GROUP
#define BK_SCHED_GROUP 0
#endif
#if BK_SCHED_GROUP > 0
#define BK_SCHED_FENCE(j) do { if (((j) % BK_SCHED_GROUP) == BK_SCHED_GROUP - 1) __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define BK_SCHED_FENCE(j) do { } while (0)
#endif
// development: BK_RATIO_STAMPS=1 writes s_memtime stamps of wave 0 at the phase boundaries of every chunk into the
// (otherwise unused) lane-state region of the workspace: 16 slots of 8 bytes per (row, chunk)
#ifndef BK_RATIO_STAMPS
#define BK_RATIO_STAMPS 0
#endif
#if BK_RATIO_STAMPS
#define BK_STAMP(k)                                                                                                  \
  if ((BK_RATIO_STAMPS >> (k)) & 1) do {                                                                             \
    __builtin_amdgcn_sched_barrier(0);                                                                               \
    const unsigned long long tm_ = __builtin_amdgcn_s_memtime();                                                     \
    if (threadIdx.x == 0)                                                                                            \
      reinterpret_cast<unsigned long long *>(ws + rowflag_bytes(B))[((row * nch + ch) * 16) + (k)] = tm_;            \
    __builtin_amdgcn_sched_barrier(0);                                                                               \
  } while (0)
#else
#define BK_STAMP(k) do { } while (0)
#endif
#ifndef BK_RATIO_STAGGER
#define BK_RATIO_STAGGER 0
#endif
#ifndef BK_RATIO_EXPERIMENT
#define BK_RATIO_EXPERIMENT 0
#endif
#ifndef BK_RELAY_DEBUG
#define BK_RELAY_DEBUG 0      // development: poll statistics of the in-launch hand-off in state[13..15]
#endif
#ifndef BK_UNIT_STAMPS
#define BK_UNIT_STAMPS 0      // development: per workgroup start / end wall clock and hardware id, 32 bytes each, 8 MiB
#endif                        // into the lane-state region of the workspace (tools/dev_units.py)
#ifndef BK_RELAY_PERSIST
#define BK_RELAY_PERSIST 1    // MODE 3: one workgroup per place on the GPU, each taking every gridDim-th unit
#endif
#ifndef BK_RELAY_NOWAIT
#define BK_RELAY_NOWAIT 0     // development: no hand-off at all (wrong results; the cost of everything else)
#endif
#ifndef BK_LOOK_TICKET
#define BK_LOOK_TICKET 0      // MODE 4: a workgroup draws its chunk from a counter in the state block when it starts
#endif                        // (0: the chunk is the workgroup's index, relying on workgroups starting in index order)
#ifndef BK_LOOK_LONG3
#define BK_LOOK_LONG3 0       // MODE 4 with double chains: three workgroups per CU (register budget 168)
#endif
#ifndef BK_LOOK_NOWAIT
#define BK_LOOK_NOWAIT 0      // development: the chunks of MODE 4 do not wait for each other (wrong results; timing only)
#endif
#ifndef BK_PRIO
// Wave priority by phase: raised from the start of a chunk to the barrier behind its staging (the part of a chunk
// that three other waves wait for), normal in chains, epilogue and stores.  Measured at B = 1024, N = 32768: forward
// 0.1535 -> 0.1520 ms, forward + backward 0.2650 -> 0.2610 ms; with the double-precision chains of long rows 1-2 %
// slower, so not there.  (Development: 1 = only while the loads are issued, 3 = chains and epilogue instead,
// 4 = 2 and the stores, 5 = staging only: all equal to or slower than 2.)
#define BK_PRIO (BK_RATIO_LONG ? 0 : 2)
#endif
#ifndef BK_COMMIT_SUMS
#define BK_COMMIT_SUMS 1
#endif
#ifndef BK_RATIO_PREFETCH
#define BK_RATIO_PREFETCH 1   // raw inputs of the next chunk held in registers during the current chunk's arithmetic
#endif
constexpr int RH = 16;                          // warm-up length = halo staged on both sides of a chunk
#ifndef BK_RH_COARSE
#define BK_RH_COARSE 8
#endif
constexpr int RH_COARSE = BK_RH_COARSE;         // leading warm-up steps taken in the coarse form
#ifndef BK_RH_FULL
#define BK_RH_FULL 4
#endif
constexpr int RH_FULL = BK_RH_FULL;             // trailing warm-up steps taken as full steps
constexpr int RN = RCHUNK + 2 * RH;              // staged elements per chunk
constexpr int RAL_PAD = RN + RN / E + 8;        // float2 entries, index q + q/E
constexpr int RK_PAD = RN + 1 + (RN + 1) / E + 8;

struct Prod {  // complex mantissa * 2^e
  C2 m;
  int e;
};
__device__ __forceinline__ Prod prod_one() {
  Prod p;
  p.m = c2((Real)1, (Real)0);
  p.e = 0;
  return p;
}
__device__ __forceinline__ void pnorm(Prod &p) {
  const int de = expo(fmaxf(fabsf((float)p.m.x), fabsf((float)p.m.y)));
  const Real sc = (Real)pow2_neg(de);
  p.m = c2(p.m.x * sc, p.m.y * sc);
  p.e += de;
}
__device__ __forceinline__ Prod pmul(const Prod &a, const Prod &b) {
  Prod r;
  r.m = cmulr(a.m, b.m);
  r.e = a.e + b.e;
  pnorm(r);
  return r;
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_r(float old, float src) { return dpp_f<CTRL, ROW_MASK>(old, src); }
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_r(double old, double src) {  // the two halves of a double move separately
  const long long o = __double_as_longlong(old), v = __double_as_longlong(src);
  const int lo = dpp_i<CTRL, ROW_MASK>((int)(o & 0xffffffffLL), (int)(v & 0xffffffffLL));
  const int hi = dpp_i<CTRL, ROW_MASK>((int)(o >> 32), (int)(v >> 32));
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ Prod prod_dpp(const Prod &p) {  // identity where the selection has no source lane
  Prod r;
  r.m.x = dpp_r<CTRL, ROW_MASK>((Real)1, p.m.x);
  r.m.y = dpp_r<CTRL, ROW_MASK>((Real)0, p.m.y);
  r.e = dpp_i<CTRL, ROW_MASK>(0, p.e);
  return r;
}
struct SharedR {
  union {
    struct {
      float2 al_s[RAL_PAD];  // regularised diagonal a' of elements base - RH .. base + RCHUNK + RH - 1
      float k_s[RK_PAD];     // k_{i-1} for the same elements, plus one more
    } in;
    float2 out_s[RO_PAD];     // G of one chunk for coalesced stores
  } u;
  float4 wedge[2][RWAVES];    // last lane of each wave: theta warm-up result (xy) and phi end value (zw), for the
                             // first lane of the next wave
  float4 edge[2];            // the same from the first lane of the chunk to the right
  Prod tot[2][RWAVES];
  Prod carry[2];
};

// MODE 3 hand-off between the workgroups of one row, all in ONE launch.  A record of RELAY_REC floats per (row,
// segment) in the lane-state region of the workspace (the ratio form keeps no lane states; the scan kernel of the
// second launch rewrites that region before it reads it): [0,1] mantissa and [2] exponent of the running product
// after the segment, [4..7] the edge values of its leftmost lane, [8,9] a 64-bit tag that says "written during this
// launch".  The tag is derived from a per-launch epoch kept in the state block, so whatever another owner of the
// memory left in the record cannot pass for a hand-off (a foreign value would have to equal a 64-bit hash of the
// current epoch), and the reader of a record takes its tag down, so neither can what an earlier launch left.
constexpr int RELAY_REC = 16;
constexpr int ST_EPOCH = 9, ST_REDO_ALL = 12;
[[maybe_unused]] constexpr int ST_TICKET = 10;               // (BK_LOOK_TICKET builds only)  // ints of the state block used by MODE 3 and 4
constexpr long long RELAY_WAIT_TICKS = 5000000;  // of the 100 MHz wall clock: 50 ms before a unit gives the launch up
__device__ __forceinline__ unsigned long long relay_tag(int epoch, long unit) {
  unsigned long long x = ((unsigned long long)(unsigned)epoch << 32) ^ (unsigned long long)unit ^ 0x9E3779B97F4A7C15ull;
  x ^= x >> 31; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 29; x *= 0x94D049BB133111EBull; x ^= x >> 32;
  return x | 1ull;
}

// MODE 4 hand-over between the chunks of one row, all in ONE launch (few long rows): one workgroup per chunk, started
// in the order "rightmost chunks of all rows first", so that what a chunk waits for always has a lower index.  The
// chains of a chunk do not depend on the running product, only its epilogue does: a chunk publishes its own product
// (its "aggregate", with the edge values of its leftmost lane) as soon as its chains and the workgroup scan are done,
// and one wave then collects the product of everything to its right from its neighbours' records.  So that the
// result does not depend on who finished when, WHICH records are combined is fixed: every LB_W-th chunk (an anchor)
// also publishes the running product up to and including itself, and a chunk reads the anchor to its right plus
// every aggregate between that anchor and itself -- at most LB_W records, one per lane, multiplied in a fixed tree.
// Record of LB_REC floats per (row, chunk) in the lane-state region of the workspace: [0..3] aggregate mantissa (two
// floats, or two doubles in the long-row build), [4] its exponent, [8..11] edge values, [12,13] tag of the aggregate,
// [16..19] running product mantissa, [20] exponent, [28,29] its tag.  Tags as for MODE 3 (hash of the call's epoch and
// the record's index); the call's last launch takes every tag down again.
constexpr int LB_REC = 32;
constexpr int LB_W = 64;
constexpr int LB_A_EXP = 4, LB_EDGE = 8, LB_A_TAG = 12, LB_P = 16, LB_P_EXP = 20, LB_P_TAG = 28;
static_assert(LB_W == 64, "one record per lane of the collecting wave");
__device__ __forceinline__ void lb_put(float *rec, int at, int ate, const Prod &p) {
  if (LONG) {
    __hip_atomic_store(reinterpret_cast<double *>(rec + at), (double)p.m.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(reinterpret_cast<double *>(rec + at) + 1, (double)p.m.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else {
    __hip_atomic_store(rec + at, (float)p.m.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(rec + at + 1, (float)p.m.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __hip_atomic_store(reinterpret_cast<int *>(rec + ate), p.e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ Prod lb_get(const float *rec, int at, int ate) {
  Prod p;
  if (LONG) {
    p.m = c2((Real)__hip_atomic_load(reinterpret_cast<const double *>(rec + at), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
             (Real)__hip_atomic_load(reinterpret_cast<const double *>(rec + at) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  } else {
    p.m = c2((Real)__hip_atomic_load(rec + at, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
             (Real)__hip_atomic_load(rec + at + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  }
  p.e = __hip_atomic_load(reinterpret_cast<const int *>(rec + ate), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return p;
}

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
// |r|^2 is refined by one Newton step: the hardware reciprocal rounds with a bias of a fraction of an ulp, which would
// add up linearly over the N factors of the product.
// Clamp factor exp(g(d)), d = -ln|r|: g = 88 tanh(d/88) - d = -88 (u^3/3 - 2u^5/15 + ...), u = d/88.  With
// Ln = log2 |r|^2 (one hardware log) u = -c Ln, c = ln2/176, so g = WALL_A Ln^3 - WALL_B Ln^5 and
//     exp(g) = 1 + Ln^3 (WALL_A - WALL_B Ln^2 + (WALL_A^2 / 2) Ln^3)      (wall_factor)
// to g^2 and u^5.  g is a 1e-5 effect on k, but what it leaves out is one-signed, and a one-signed relative error
// eps of every ratio becomes N eps in the product: the g^2/2 and u^5 terms are 5e-10 each for |r| ~ 2, i.e. 1.6e-4 at
// N = 2^20, so the steps that feed the product keep them.  The next terms (u^7, g^3) are below 1e-13 there, below
// 1e-6 for |r| between 1e-2 and 1e2.  Larger |r| only shrinks the influence of k om on the next step; smaller |r| is
// recorded in lmin (the minimum of Ln over the lane's steps) and the row is handed to the scan kernel.
// wall_factor_short = 1 + WALL_A Ln^3, for warm-up steps whose error is contracted away before it reaches a product.
constexpr float WALL_C = 0.69314718055994531f / 176.0f;
constexpr float WALL_A = 88.0f * WALL_C * WALL_C * WALL_C / 3.0f;
constexpr float WALL_B = 176.0f * WALL_C * WALL_C * WALL_C * WALL_C * WALL_C / 15.0f;
constexpr float WALL_LN_MIN = -13.3f;
__device__ __forceinline__ float wall_factor(float Ln) {
  const float t = Ln * Ln;
  const float q = t * Ln;
  return fmaf(q, fmaf(q, 0.5f * WALL_A * WALL_A, fmaf(t, -WALL_B, WALL_A)), 1.0f);
}
__device__ __forceinline__ float wall_factor_short(float Ln) { return fmaf(WALL_A * Ln, Ln * Ln, 1.0f); }

// 1 / n: hardware reciprocal plus one Newton step.  In double the seed is the float32 reciprocal (2e-7), which one
// step takes to 4e-14.
__device__ __forceinline__ float rcp_refined(float n, float) {
  const float y = __builtin_amdgcn_rcpf(n);
  return fmaf(y, fmaf(-n, y, 1.0f), y);
}
__device__ __forceinline__ double rcp_refined(double n, float nf) {
  const double y = (double)__builtin_amdgcn_rcpf(nf);
  return fma(y, fma(-n, y, 1.0), y);
}
// exp(g) as 1 + (the rest), the sum formed in Real so that in double the factor carries g's full relative precision
__device__ __forceinline__ Real wall_factor_r(float Ln) {
  if (!LONG) return (Real)wall_factor(Ln);
  const float t = Ln * Ln;
  const float q = t * Ln;
  return (Real)1 + (Real)(q * fmaf(q, 0.5f * WALL_A * WALL_A, fmaf(t, -WALL_B, WALL_A)));
}

__device__ __forceinline__ C2 ratio_step(float2 al, float k, C2 om, C2 &r, C2 &inv, float &lmin) {
  r = c2(rfma(-(Real)k, om.x, (Real)al.x), rfma(-(Real)k, om.y, (Real)al.y));
  const Real n = rfma(r.x, r.x, r.y * r.y);
  const float nf = (float)n;
  const Real y = rcp_refined(n, nf);
  inv = c2(r.x * y, -r.y * y);
  const float Ln = __builtin_amdgcn_logf(nf);
  lmin = fminf(lmin, Ln);
  const Real wc = wall_factor_r(Ln);
  return c2(inv.x * wc, inv.y * wc);
}
// The same step where only om(r) and r are wanted (phi: the product takes s itself, not its reciprocal).
__device__ __forceinline__ C2 ratio_step_om(float2 al, float k, C2 om, C2 &r, float &lmin) {
  r = c2(rfma(-(Real)k, om.x, (Real)al.x), rfma(-(Real)k, om.y, (Real)al.y));
  const Real n = rfma(r.x, r.x, r.y * r.y);
  const float nf = (float)n;
  const Real y = rcp_refined(n, nf);
  const float Ln = __builtin_amdgcn_logf(nf);
  lmin = fminf(lmin, Ln);
  const Real yw = y * wall_factor_r(Ln);
  return c2(r.x * yw, -r.y * yw);
}
// Warm-up form: the short clamp factor and no Newton refinement.  Both leave one-signed errors of 1e-9 .. 1e-8, which
// the steps after them contract; the last warm-up steps are full steps (ratio_step_om) so that what reaches the lane's
// own elements, and through them the product, stays below 1e-11 per element.
__device__ __forceinline__ float2 ratio_step_warm(float2 al, float k, float2 om) {
  const float2 r = make_float2(fmaf(-k, om.x, al.x), fmaf(-k, om.y, al.y));
  const float n = fmaf(r.x, r.x, r.y * r.y);
  const float yw = __builtin_amdgcn_rcpf(n) * wall_factor_short(__builtin_amdgcn_logf(n));
  return make_float2(r.x * yw, -r.y * yw);
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

// Staging of a' and k for elements base - RH .. base + RCHUNK + RH - 1 (k: one more), in two halves so that the
// global loads of the next chunk can be in flight while the current one is computed:
//   load_ratio   global -> registers (17 rounds of 256 slots cover the RN + 1 = 4129 slots).  Outside the row a = 1,
//                b = c = 0.  `full` (uniform): every index lies inside the row, no bounds logic.
//   commit_ratio registers -> LDS: the operator's clamp / NaN rules, regularisation, k = b c.  Returns the NaN bits as
//                stage_chunk does.
constexpr int RNJ = E + 1;
template <typename TIn>
__device__ __forceinline__ void load_ratio(const TIn *__restrict__ a, const TIn *__restrict__ b,
                                           const TIn *__restrict__ c, long base, long N, bool full, bool left_open,
                                           float (&av)[RNJ], float (&bv)[RNJ], float (&cv)[RNJ]) {
  const int t = threadIdx.x;
  // 32-bit offsets from uniform bases (scalar base + vector offset addressing); slot q holds element base - RH + q
  const TIn *__restrict__ a0 = a + (base - RH);
  const TIn *__restrict__ b0 = b + (base - RH - 1);
  const TIn *__restrict__ c0 = c + (base - RH - 1);
  if (full) {
#pragma unroll
    for (int j = 0; j < RNJ; ++j) {
      const int q = min(t + j * RT, RN);
      av[j] = to_f32<TIn>(__builtin_nontemporal_load(a0 + q));  // each input is read once (plus a halo's worth)
      bv[j] = to_f32<TIn>(__builtin_nontemporal_load(b0 + q));
      cv[j] = to_f32<TIn>(__builtin_nontemporal_load(c0 + q));
    }
  } else {
    // branch-free: every load uses a clamped, always valid index (b and c must be readable at index 0 even when the
    // row has no off-diagonal: the caller passes `a` for them then); out-of-row values are replaced afterwards
    // (left_open: the row continues to the left of index 0, only its right end bounds the loads)
    const int lo_a = left_open ? 0 : (int)max(0L, RH - base);  // first slot whose element index is >= 0
    const int hi_a = (int)min((long)RN, N - 1 - (base - RH));  // last slot whose element index is <= N-1 (may be < lo)
    const int lo_k = left_open ? 0 : (int)max(0L, RH + 1 - base);  // k index = element index - 1 >= 0
    const int hi_k = (int)min((long)RN, N - 2 - (base - RH - 1));
    const int ca = max(hi_a, lo_a), ck = max(hi_k, lo_k);
#pragma unroll
    for (int j = 0; j < RNJ; ++j) {
      const int q = min(t + j * RT, RN);
      const int qa = min(max(q, lo_a), ca);
      const int qk = min(max(q, lo_k), ck);
      const float va = to_f32<TIn>(a0[qa]);
      const float vb = to_f32<TIn>(b0[(N >= 2 || left_open) ? qk : qa + 1]);
      const float vc = to_f32<TIn>(c0[(N >= 2 || left_open) ? qk : qa + 1]);
      av[j] = (q >= lo_a && q <= hi_a) ? va : 1.0f;
      bv[j] = (q >= lo_k && q <= hi_k) ? vb : 0.0f;
      cv[j] = (q >= lo_k && q <= hi_k) ? vc : 0.0f;
    }
  }
}

// Padded LDS slot of staged slot q = t + j * RT: q + q / E = (t + t / E) + j * (RT + RT / E), a per-thread base plus a
// compile-time offset (written out because the compiler does not split the division by itself).
static_assert(RT % E == 0, "padded slot arithmetic");
constexpr int RQS = RT + RT / E;

// KIND 1: the operator's NaN rules element by element (bk_core.py:256-264), for a chunk in which some value is a NaN;
// KIND 0: the clamps alone (v_med3); KIND 2: no clamps either (every value is known to lie inside them).  The raw input
// registers are left as loaded.
template <int KIND>
__device__ __forceinline__ unsigned commit_pass(const RowCfg &cfg, const float (&av)[RNJ], const float (&bv)[RNJ],
                                                const float (&cv)[RNJ], SharedR &sh) {
  unsigned seen = 0;
  const int t = threadIdx.x;
  float2 *__restrict__ al_t = sh.u.in.al_s + (t + t / E);
  float *__restrict__ k_t = sh.u.in.k_s + (t + t / E);
  const float xi = -cfg.z.y, xi2 = xi * xi;
  const bool zfar = xi2 >= 1e-24f;  // uniform: a - z is never near zero, the regularisation needs no special case
#pragma unroll
  for (int j = 0; j < RNJ; ++j) {
    float a = av[j], b = bv[j], c = cv[j];
    if (KIND != 2 && !cfg.raw) {
      if (KIND == 1) {
        a = clamp_keep_nan(a, 100.0f);
        if (a != a) { a = 0.0f; seen |= 1u; }
        b = clamp_keep_nan(b, 10.0f);
        c = clamp_keep_nan(c, 10.0f);
        if (b != b) { seen |= 2u; if (cfg.fix_bc) b = 1.0f; }
        if (c != c) { seen |= 2u; if (cfg.fix_bc) c = 1.0f; }
      } else {
        a = __builtin_amdgcn_fmed3f(a, -100.0f, 100.0f);
        b = __builtin_amdgcn_fmed3f(b, -10.0f, 10.0f);
        c = __builtin_amdgcn_fmed3f(c, -10.0f, 10.0f);
      }
    }
    float2 al;
    if (zfar) {
      const float xr = a - cfg.z.x;
      const float f = fmaf(1e-3f, __builtin_amdgcn_rsqf(fmaf(xr, xr, xi2)), 1.0f);
      al = make_float2(xr * f, xi * f);
    } else {
      al = reg_alpha(a, cfg);
    }
    const int q = t + j * RT;
    if (j < E || q <= RN) {
      k_t[j * RQS] = b * c;
      if (j < E || q < RN) al_t[j * RQS] = al;
    }
  }
  return seen;
}

__device__ __forceinline__ unsigned commit_ratio(const RowCfg &cfg, const float (&av)[RNJ], const float (&bv)[RNJ],
                                                 const float (&cv)[RNJ], SharedR &sh) {
#if BK_COMMIT_SUMS
  // Sums of squares (full-rate fma) instead of a clamp per value and a NaN test per pair (both half rate): a wave
  // whose sums stay below the squares of the clamps' bounds holds no value outside them and no NaN, and its values go
  // to LDS as they are.  Any other wave takes the passes below.
  if (!cfg.raw) {
    float sa = 0.0f, sb = 0.0f, sc = 0.0f;
#pragma unroll
    for (int j = 0; j < RNJ; ++j) {
      sa = fmaf(av[j], av[j], sa);
      sb = fmaf(bv[j], bv[j], sb);
      sc = fmaf(cv[j], cv[j], sc);
    }
    const bool plain = sa <= 1.0e4f && sb <= 1.0e2f && sc <= 1.0e2f;  // false for a NaN or an infinity
    if (__builtin_expect(__all(plain), 1)) return commit_pass<2>(cfg, av, bv, cv, sh);
  }
#endif
  bool nan = false;  // v_cmp_u tests two values per instruction
  if (!cfg.raw) {
#pragma unroll
    for (int j = 0; j < RNJ; ++j) {
      nan |= __builtin_isunordered(av[j], bv[j]);
      if (j & 1) nan |= __builtin_isunordered(cv[j - 1], cv[j]);
    }
    nan |= __builtin_isunordered(cv[RNJ - 1], cv[RNJ - 1]);
  }
  unsigned seen = commit_pass<0>(cfg, av, bv, cv, sh);
  if (__builtin_expect(__any(nan), 0)) seen = commit_pass<1>(cfg, av, bv, cv, sh);  // rewrites the wave's slots
  return seen;
}

// MODE 0: one workgroup per row.
// MODE 3: more rows than the GPU holds workgroups at once: rows cut into S units of seg_ch chunks that hand the running
// product and edge values on inside ONE launch (records and their tags: see RELAY_REC above); the same outputs as
// MODE 0, bit for bit.
// SAVE (MODE 0 only, for the analytic backward): also stores r_j and s_j, and writes G_raw (before the final
// log-magnitude clamp and the clip; 0 outside 1e-18 <= |G_raw| <= 1e18) instead of G.
// BST (float32 chains, for the analytic backward): also stores, for every block of 8 elements m, the two values the ratio
// chains of that block start from -- om(r_{8m-1}), what the theta chain carries into the block from the left, and
// om(s_{8m+9}), what the phi chain carries towards it from the right (one more step, at element 8m+8, gives the backward
// the terms it needs of that element) -- in save_r_, per row two planes of one complex number per block: plane 0 holds
// om(r_{8m-1}) at index m, plane 1 om(s_{8m+1}) at index m (a wave's store is then one strided run of 8-byte pieces).
// Each value is stored the moment it exists, at an address worked out on the spot, and the planes are padded to whole
// chunks plus one block so that no store needs a bounds test: anything that lives across the chains, or any condition
// around the stores, costs this kernel its budget of three workgroups per CU (a variant that kept the four values
// until the end of the chunk spilled 100-156 registers there).
static_assert(E == 16, "one record per lane and chunk");
template <typename TIn, bool PHI_EXACT, int MODE, bool SAVE = false, bool BST = false>
__global__ __launch_bounds__(RT, (MODE == 4 && (!LONG || BK_LOOK_LONG3)) ? BK_RATIO_WAVES : ((MODE == 0 || MODE == 3) && !SAVE && !LONG && sizeof(TIn) != 2) ? BK_RATIO_WAVES : 2) void bk_ratio_forward_kernel(
    const TIn *__restrict__ a_, long lda, const TIn *__restrict__ b_, long ldb, const TIn *__restrict__ c_, long ldc,
    const float *__restrict__ z_dev, float z_re, float z_im, long B, long N_, float2 *__restrict__ G_,
    float2 *__restrict__ F_, char *__restrict__ ws, int *state, unsigned flags, float fclamp, int S,
    int seg_ch, float2 *__restrict__ save_r_ = nullptr, float2 *__restrict__ save_s_ = nullptr) {
  __shared__ SharedR sh;
  constexpr bool RELAY = MODE == 3;
  constexpr bool LOOK = MODE == 4;   // one workgroup per chunk; S = chunks per row, sg = the chunk counted from the right
  constexpr bool PERSIST = RELAY && BK_RELAY_PERSIST;
  const unsigned units = RELAY ? (unsigned)B * (unsigned)S : 0u;
  // PERSIST, two units a row, at most half as many rows again as workgroups ("paired"): workgroup w does its own row
  // w as right half, then left half -- a hand-over to itself -- and the E = B - gridDim extra rows go to the first 2 E
  // workgroups: the right half of row gridDim + w BEFORE workgroup w's own row (w < E), the left half of row
  // gridDim + w - E AFTER workgroup w's own row (E <= w < 2 E).  A left half then never waits for a unit that runs at
  // the same time as the one before it (in plain index order a tenth of the left halves did, up to 40 us).
  const bool paired = PERSIST && S == 2 && gridDim.x < (unsigned)B && 2u * ((unsigned)B - gridDim.x) <= gridDim.x;
  for (unsigned unit = (paired && blockIdx.x < (unsigned)B - gridDim.x) ? gridDim.x + blockIdx.x : blockIdx.x;;) {
  long row = MODE == 0 ? (long)blockIdx.x : (long)(blockIdx.x / S);
  int sg = MODE == 0 ? 0 : (int)(blockIdx.x % S);
  if (LOOK) {  // rightmost chunks of all rows first: what a chunk waits for has a lower index
    unsigned u = blockIdx.x;
#if BK_LOOK_TICKET
    // the index comes from a counter, drawn when the workgroup starts: every lower index then belongs to a workgroup
    // that has started already, whatever order the hardware starts workgroups in
    __shared__ unsigned ticket_s;
    if (threadIdx.x == 0)
      ticket_s = (unsigned)__hip_atomic_fetch_add(&state[ST_TICKET], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    u = ticket_s;
#endif
    row = (long)(u % (unsigned)B);
    sg = (int)(u / (unsigned)B);
  }
  if (RELAY) {
    // Units in index order, the rightmost segments of all rows first: the unit a workgroup depends on (same row, the
    // segment to the right) has an index lower by B, and rightmost segments wait for nothing.  The grid is at most one
    // workgroup per place on the GPU (PERSIST), workgroup w taking units w, w + gridDim, w + 2 gridDim, ... in that
    // order: no counter to draw from (an atomic with a return value at every unit boundary measured 0.22 ms against
    // 0.16), no workgroup turnover inside the launch.  The lowest unfinished unit belongs to a workgroup that has
    // finished everything before it, and what it waits for is lower still, hence done: it can always proceed once
    // its workgroup runs, and every workgroup of the grid gets to run because the grid fits the GPU (other work on
    // the GPU delays that, it does not prevent it).  Without PERSIST: one workgroup per unit, started by the hardware
    // in index order.  (Should a wait outlast RELAY_WAIT_TICKS all the same -- fewer places than the host assumed --
    // the unit gives up and the second launch recomputes every row.)
    row = (long)(unit % (unsigned)B);
    sg = S - 1 - (int)(unit / (unsigned)B);
  }
  int *rowflag = reinterpret_cast<int *>(ws);
  // MODE 3 works in coordinates that start at the unit's first chunk (element index, row length and chunk numbers
  // all count from there; what lies to the left is reachable through `left_open`): the chunk loop then is the loop
  // of MODE 0, which is what fits the register budget of three workgroups per CU.
  const long shift = RELAY ? (long)sg * seg_ch * RCHUNK : 0;
  const bool left_open = RELAY && sg > 0;  // the elements before index 0 belong to the row
  const long N = N_ - shift;
  const TIn *a = a_ + row * lda + shift;
  const TIn *b = N_ >= 2 ? b_ + row * ldb + shift : a;  // never used for data when N == 1, but must be readable
  const TIn *c = N_ >= 2 ? c_ + row * ldc + shift : a;
  float2 *G = G_ + row * N_ + shift;
  float2 *F = F_ ? F_ + row * N_ + shift : nullptr;
  // boundary states of the row (two complex numbers per block), in the unit's coordinates like everything else
  // boundary states of the row: two planes of one float2 per block of 8 elements (the blocks of whole chunks, plus
  // one) -- om(r) entering the block, then om(s) at the block's second element --, in the unit's coordinates like
  // everything else
  const long bst_plane = ((N_ + RCHUNK - 1) / RCHUNK) * (2 * RT) + 1;
  float2 *bst16 = BST ? reinterpret_cast<float2 *>(save_r_) + row * (2 * bst_plane) + shift / 8 : nullptr;
  const int nch = (int)((N + RCHUNK - 1) / RCHUNK);
  const int ch0 = LOOK ? nch - 1 - sg : 0;
  const int ch1 = LOOK ? ch0 + 1 : RELAY ? min(nch, seg_ch) : nch;  // (a unit: seg_ch chunks from its start)
  RowCfg cfg;
  cfg.z = z_dev ? make_float2(z_dev[0], z_dev[1]) : make_float2(z_re, z_im);
  cfg.N = N;
  cfg.raw = (flags & BKCORE_FLAG_RAW) != 0;
  cfg.fix_bc = false;
  cfg.fclamp = fclamp;
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = t >> 6;
  const int seg = RT - 1 - t;  // descending segments over ascending threads: the product runs right -> left
  constexpr int KO = PHI_EXACT ? 1 : 0;
  unsigned seen = 0;
  bool bad = false;
  int nscan = 0;
  if (t == 0) {
    sh.carry[0] = prod_one();
  }
  // BKCORE_FLAG_ADAPTIVE: the previous call recomputed most of its rows (state[7]) and this is not one of the calls
  // that try again (state[8] == 0): only read and stage the inputs (that is what finds the NaN verdicts the second
  // launch needs), skip the arithmetic, and leave every row to the scan kernel
  const bool skip_arith = MODE == 0 && !SAVE && (flags & BKCORE_FLAG_ADAPTIVE) && state &&
                          __hip_atomic_load(&state[7], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 &&
                          __hip_atomic_load(&state[8], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;

  // Raw inputs of the chunk about to be processed.  One workgroup per row without the saving stores fits 3 waves per
  // SIMD (168 registers) when the loads are issued at the start of the chunk and the CU's other workgroups cover their
  // latency; the other variants hold the next chunk's inputs in registers during the current chunk's arithmetic
  // (2 waves per SIMD).
  // 16-bit operands (the layer's bfloat16 diagonal) measured faster with two workgroups per CU and the prefetch:
  // twice the load instructions per byte
  constexpr bool PREFETCH = !((MODE == 0 || MODE == 3) && !SAVE && BK_RATIO_WAVES >= 3 && sizeof(TIn) != 2) && !LONG && !LOOK && BK_RATIO_PREFETCH;
  float av[RNJ], bv[RNJ], cv[RNJ];
  auto prefetch = [&](int ch) __attribute__((always_inline)) {
    // `full`: the chunk and its halos (and the one extra k) lie inside the row
    load_ratio<TIn>(a, b, c, (long)ch * RCHUNK, N, (ch >= 1 || left_open) && (long)(ch + 1) * RCHUNK + RH + 1 <= N,
                    left_open, av, bv, cv);
  };
  if (RELAY) {
    if (!BK_RELAY_NOWAIT && sg < S - 1 && t == 0) {
      // a segment that has a right neighbour starts from that neighbour's running product and edge values (carry[0]
      // and edge[1] are the slots the first chunk reads)
      const float *src = reinterpret_cast<const float *>(ws + rowflag_bytes(B)) + (row * S + sg + 1) * RELAY_REC;
      const int epoch = __hip_atomic_load(&state[ST_EPOCH], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned long long want = relay_tag(epoch, row * S + sg + 1);
      const unsigned long long *tagp = reinterpret_cast<const unsigned long long *>(src + 8);
      // Every access to a record is an agent-scope atomic (served at the level all CUs share), ordered by waiting for
      // the memory counter: no acquire / release fences, which on this GPU write back and invalidate the whole L2.
      bool got = __hip_atomic_load(tagp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == want;
      if (!got) {
        const long long t0 = wall_clock64();
#if BK_RELAY_DEBUG
        __hip_atomic_fetch_add(&state[15], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
#pragma unroll 1
        while (!got && wall_clock64() - t0 < RELAY_WAIT_TICKS) {
          __builtin_amdgcn_s_sleep(8);
          got = __hip_atomic_load(tagp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == want;
        }
#if BK_RELAY_DEBUG
        __hip_atomic_fetch_max(&state[14], (int)(wall_clock64() - t0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the record is read only after its tag has been seen
      if (!got) {
        // never expected; the second launch then recomputes every row, so nothing of this launch is used
        __hip_atomic_store(&state[ST_REDO_ALL], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        Prod p;
        p.m = c2((Real)__hip_atomic_load(src + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                 (Real)__hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        p.e = __float_as_int(__hip_atomic_load(src + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        sh.carry[0] = p;
        sh.edge[1] = make_float4(__hip_atomic_load(src + 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                                 __hip_atomic_load(src + 5, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                                 __hip_atomic_load(src + 6, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                                 __hip_atomic_load(src + 7, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        // A record has one reader, who takes the tag down again: the next launch finds no tag of this one even if
        // its epoch is the same (a state block that was zeroed again, e.g. by the memset of a replayed graph).
        __hip_atomic_store(const_cast<unsigned long long *>(tagp), 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
  auto one = [&](auto tail, int ch) __attribute__((always_inline)) {
    constexpr bool TAIL = decltype(tail)::value;
    const long base = (long)ch * RCHUNK;

    BK_STAMP(0);
#if BK_PRIO == 1 || BK_PRIO == 2 || BK_PRIO == 4
    __builtin_amdgcn_s_setprio(3);
#endif
#if BK_PRIO == 6
    __builtin_amdgcn_s_setprio(1);
#endif
    if (!PREFETCH) prefetch(ch);  // no software pipelining: the other workgroups of the CU cover the latency
#if BK_PRIO == 1
    __builtin_amdgcn_s_setprio(0);
#endif
    __syncthreads();  // previous chunk's output tile has been stored
#if BK_PRIO == 5
    __builtin_amdgcn_s_setprio(3);
#endif
    BK_STAMP(1);
    seen |= commit_ratio(cfg, av, bv, cv, sh);
    BK_STAMP(2);
    __syncthreads();
#if BK_PRIO == 2 || BK_PRIO == 4 || BK_PRIO == 5 || BK_PRIO == 6
    __builtin_amdgcn_s_setprio(0);
#endif
#if BK_PRIO == 3
    __builtin_amdgcn_s_setprio(3);
#endif
    BK_STAMP(3);
    if (PREFETCH && ch > ch0) prefetch(ch - 1);  // in flight during this chunk's arithmetic

    const int q0 = seg * E + RH;      // staged slot of the lane's first element (a multiple of 16)
    const float2 *al_l = sh.u.in.al_s + rpad16(q0);
    const float *k_l = sh.u.in.k_s + rpad16(q0);
    const long i0 = base + (long)seg * E;
    C2 ir[E];                         // 1 / r_e; then (unless SAVE) the lane-local part of G_raw
    C2 sv[SAVE ? E : 1];              // s_e and the lane-local part of G_raw as arrays of their own: saving variant only
    C2 lc[SAVE ? E : 1];
    // Warm-up: theta over the 16 elements to the left of the lane's own, phi over the 16 to the right, the two
    // independent chains side by side.  Past the end of the row om_phi = 0 (phi_{N+1} = 0: the recurrence starts with
    // s_{N-1} = a'_{N-1}).
#if BK_RATIO_EXPERIMENT == 1
    // development: no chain arithmetic at all (memory and staging floor of the kernel)
    C2 wp = c2((Real)0, (Real)0);
    float lmin = 0.0f;
    const float2 th_west = make_float2(0.f, 0.f), ph_west = th_west, th_fin = th_west;
    Prod run = prod_one();
    int le[E / 4];
#pragma unroll
    for (int j = 0; j < E; ++j) {
      ir[j] = to_c2(al_l[roff(j)]);
      ir[j].x += (Real)k_l[roff(j)];
      le[j >> 2] = 0;
    }
#else
    // The coarse and short-factor steps run in float32 whatever Real is (their error is contracted away); the last
    // RH_FULL steps and everything after them in Real.
    float2 wtf = make_float2(0.f, 0.f), wpf = make_float2(0.f, 0.f);
    C2 wt, wp, r, inv;
    float lmin = 0.0f;
#pragma unroll
    for (int j = 0; j < RH; ++j) {
      const int d = E + RH - 1 - j;
      if (j < RH_COARSE) {
        wtf = ratio_step_coarse(al_l[roff(j - RH)], k_l[roff(j - RH)], wtf);
        wpf = ratio_step_coarse(al_l[roff(d)], k_l[roff(d + KO)], wpf);
        if (TAIL && i0 + d >= N) wpf = make_float2(0.f, 0.f);
      } else if (j < RH - RH_FULL) {
        wtf = ratio_step_warm(al_l[roff(j - RH)], k_l[roff(j - RH)], wtf);
        wpf = ratio_step_warm(al_l[roff(d)], k_l[roff(d + KO)], wpf);
        if (TAIL && i0 + d >= N) wpf = make_float2(0.f, 0.f);
      } else {
        if (j == RH - RH_FULL) {
          wt = to_c2(wtf);
          wp = to_c2(wpf);
        }
        float lw = 0.0f;  // the warm-up's own small ratios are not the row's
        wt = ratio_step_om(al_l[roff(j - RH)], k_l[roff(j - RH)], wt, r, lw);
        wp = ratio_step_om(al_l[roff(d)], k_l[roff(d + KO)], wp, r, lw);
        if (TAIL && i0 + d >= N) wp = c2((Real)0, (Real)0);
      }
      BK_SCHED_FENCE(j);
    }
    const float2 th_west = to_f2(wt), ph_west = to_f2(wp);

    BK_STAMP(4);
    // theta over the lane's own elements, left -> right: keeps 1 / r_e
#pragma unroll
    for (int j = 0; j < E; ++j) {
      wt = ratio_step(al_l[roff(j)], k_l[roff(j)], wt, r, inv, lmin);
      ir[j] = inv;
      if constexpr (BST) {
        if (j == 7) {  // om(r) behind the lane's eighth element: what enters its second block of 8
          int tv = threadIdx.x;
          asm volatile("" : "+v"(tv));
          float *dst = reinterpret_cast<float *>(bst16 + (base >> 3) + 2 * (RT - 1 - tv) + 1);
          const float2 tm = to_f2(wt);
          *reinterpret_cast<float2 *>(dst) = tm;  // (plain stores: the other half of the line follows within the chunk)
        }
      }
      BK_SCHED_FENCE(j);
    }
    const float2 th_fin = to_f2(wt);
    if constexpr (BST) {  // om(r) behind the lane's last element: what enters the next block of 16
      int tv = threadIdx.x;
      asm volatile("" : "+v"(tv));
      float *dst = reinterpret_cast<float *>(bst16 + (base >> 3) + 2 * (RT - 1 - tv) + 2);
      *reinterpret_cast<float2 *>(dst) = th_fin;
    }

    BK_STAMP(5);
    // phi right -> left with the lane-local products folded in: loc_e = (1/r_e) * prod_{e' > e} s_e' / r_e' replaces
    // 1/r_e in its registers, and the running product picks up s_e / r_e as loc_e * s_e (a second dependency chain
    // beside the recurrence).  Exponents are shared by groups of 4.
    Prod run = prod_one();
    int le[E / 4];
#pragma unroll
    for (int j = 0; j < E; ++j) {
      const int e = E - 1 - j;
      wp = ratio_step_om(al_l[roff(e)], k_l[roff(e + KO)], wp, r, lmin);
      if (TAIL && i0 + e >= N) wp = c2((Real)0, (Real)0);
      if constexpr (BST) {  // om(s) at the lane's second element: one step short of what enters the block to the left
        if (e == 1 || e == 9) {  // (e == 9: the same for the lane's second block of 8)
          int tv = threadIdx.x;
          asm volatile("" : "+v"(tv));
          float *dst = reinterpret_cast<float *>(bst16 + bst_plane + (base >> 3) + 2 * (RT - 1 - tv) + (e == 9 ? 1 : 0));
          const float2 p1 = to_f2(wp);
          *reinterpret_cast<float2 *>(dst) = p1;
        }
      }
      const bool valid = !TAIL || i0 + e < N;
      const C2 loc = cmulr(ir[e], run.m);
      if (valid) run.m = cmulr(loc, r);
      if constexpr (SAVE) {
        sv[e] = r;
        lc[e] = loc;
      } else {
        ir[e] = loc;
      }
      if ((e & 3) == 0) {  // end of a group of 4: record the exponent it was processed under, then renormalise
        le[e >> 2] = run.e;
        pnorm(run);
      }
      BK_SCHED_FENCE(j);
    }
#endif
    const float2 ph_fin = to_f2(wp);
    BK_STAMP(6);
    bad |= !(lmin >= WALL_LN_MIN);  // a ratio below 1e-2 in magnitude: outside the clamp factor's series
    if constexpr (SAVE) {
      // r and s leave through the output tile (it overlays the staging area: wait until every wave's chains are done)
      const int n_sv = TAIL ? (int)max(0L, min((long)RCHUNK, N - base)) : RCHUNK;
      float2 *slot_sv = &sh.u.out_s[pad16(seg * E)];
#pragma unroll 1
      for (int pass = 0; pass < 2; ++pass) {
        __syncthreads();
        if (pass == 0) {
#pragma unroll
          for (int e = 0; e < E; ++e) {  // r = 1 / (1/r)
            const float2 q = to_f2(ir[e]);
            const float y = 1.0f / fmaf(q.x, q.x, q.y * q.y);
            slot_sv[e] = make_float2(q.x * y, -q.y * y);
          }
        } else {
#pragma unroll
          for (int e = 0; e < E; ++e) slot_sv[e] = to_f2(sv[e]);
        }
        __syncthreads();
        float2 *__restrict__ dst = (pass == 0 ? save_r_ : save_s_) + row * N + base;
#pragma unroll
        for (int j = 0; j < E; ++j) {
          const int idx = t + j * RT;
          if (!TAIL || idx < n_sv) dst[idx] = sh.u.out_s[pad16(idx)];
        }
      }
      __syncthreads();
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
    BK_STAMP(7);
    if constexpr (LOOK) {
      if (wave == 0) {
        // the chunk's own product and the edge values of its leftmost lane go out first: chunks to the left wait for them
        Prod total = sh.tot[buf][0];
#pragma unroll
        for (int wv = 1; wv < RWAVES; ++wv) total = pmul(total, sh.tot[buf][wv]);
        float *recs = reinterpret_cast<float *>(ws + rowflag_bytes(B)) + row * (long)nch * LB_REC;
        float *me = recs + (long)sg * LB_REC;
        const int epoch = __hip_atomic_load(&state[ST_EPOCH], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const long uid0 = (row * (long)nch) << 1;
        if (lane == 0) {
          // the row's verdict starts at zero with its rightmost chunk; the others add to it after they have seen a
          // record that was published after this store
          if (sg == 0) __hip_atomic_store(&rowflag[row], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (!(flags & BKCORE_FLAG_ROW_UNITS_MUTE)) {
            const float4 ed = sh.wedge[buf][RWAVES - 1];
            lb_put(me, 0, LB_A_EXP, total);
            __hip_atomic_store(me + LB_EDGE + 0, ed.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(me + LB_EDGE + 1, ed.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(me + LB_EDGE + 2, ed.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(me + LB_EDGE + 3, ed.w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the record (and the verdict's start) before its tag
            __hip_atomic_store(reinterpret_cast<unsigned long long *>(me + LB_A_TAG), relay_tag(epoch, uid0 + 2 * sg),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
        Prod entering = prod_one();
        float4 edge_in = make_float4(0.f, 0.f, 0.f, 0.f);
        bool failed = false;
        if (sg > 0) {
          const int q = (sg / LB_W) * LB_W - 1;           // the anchor to the right (-1: none, the row ends first)
          const int nA = sg - 1 - q;                       // aggregates between it and this chunk: 0 .. LB_W - 1
          const bool isA = lane < nA;
          const bool isP = lane == LB_W - 1 && q >= 0;
          const bool mine = isA || isP || lane == 0;       // lane 0 also fetches the right neighbour's edge values
          const int pos = isP ? q : sg - 1 - (mine ? lane : 0);
          const float *src = recs + (long)pos * LB_REC;
          const unsigned long long want = relay_tag(epoch, uid0 + 2 * pos + (isP ? 1 : 0));
          const unsigned long long *tagp = reinterpret_cast<const unsigned long long *>(src + (isP ? LB_P_TAG : LB_A_TAG));
          bool got = !mine || BK_LOOK_NOWAIT;
          const long long t0 = wall_clock64();
#pragma unroll 1
          for (;;) {
            if (!got) got = __hip_atomic_load(tagp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == want;
            if (__all(got)) break;
            // never expected (every record waited for belongs to a workgroup that started earlier); once one chunk
            // has given up nobody waits any more and the second launch recomputes every row
            if (__any(wall_clock64() - t0 > RELAY_WAIT_TICKS ||
                      __hip_atomic_load(&state[ST_REDO_ALL], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
              failed = true;
              break;
            }
            __builtin_amdgcn_s_sleep(4);
          }
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // records are read only after their tags have been seen
          if (failed) {
            if (lane == 0) __hip_atomic_store(&state[ST_REDO_ALL], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          } else {
            Prod w = prod_one();
            if (isA) w = lb_get(src, 0, LB_A_EXP);
            else if (isP) w = lb_get(src, LB_P, LB_P_EXP);
            if (lane == 0)
              edge_in = make_float4(__hip_atomic_load(src + LB_EDGE + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                                    __hip_atomic_load(src + LB_EDGE + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                                    __hip_atomic_load(src + LB_EDGE + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                                    __hip_atomic_load(src + LB_EDGE + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            // the product of the wave's 64 values, always in the same tree
            w = pmul(w, prod_dpp<DPP_ROW_SHR1, 0xF>(w));
            w = pmul(w, prod_dpp<DPP_ROW_SHR2, 0xF>(w));
            w = pmul(w, prod_dpp<DPP_ROW_SHR4, 0xF>(w));
            w = pmul(w, prod_dpp<DPP_ROW_SHR8, 0xF>(w));
            w = pmul(w, prod_dpp<DPP_ROW_BCAST15, 0xA>(w));
            w = pmul(w, prod_dpp<DPP_ROW_BCAST31, 0xC>(w));
            entering.m.x = __shfl(w.m.x, 63);
            entering.m.y = __shfl(w.m.y, 63);
            entering.e = __shfl(w.e, 63);
          }
        }
        if (lane == 0) {
          sh.carry[buf] = entering;
          sh.edge[buf ^ 1] = edge_in;
          if (!failed && (sg % LB_W) == LB_W - 1 && sg < nch - 1 && !(flags & BKCORE_FLAG_ROW_UNITS_MUTE)) {
            // an anchor: the running product up to and including this chunk
            lb_put(me, LB_P, LB_P_EXP, pmul(entering, total));
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(reinterpret_cast<unsigned long long *>(me + LB_P_TAG), relay_tag(epoch, uid0 + 2 * sg + 1),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
      }
      __syncthreads();
    }
    Prod v = sh.carry[buf];
    Prod fin_in = v;
#pragma unroll
    for (int wv = 0; wv < RWAVES; ++wv) {
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
      if (t > 0 || ch < ((RELAY || LOOK) ? nch : ch1) - 1) {
        bad |= differs_tight(make_float2(nb.x, nb.y), th_fin);
        bad |= differs_tight(ph_west, make_float2(nb.z, nb.w));
      }
      if (t == RT - 1) sh.edge[buf] = make_float4(th_west.x, th_west.y, ph_fin.x, ph_fin.y);
    }
    // (every wave finished reading the staged inputs before the barrier above: the tile may overwrite them)

    // epilogue (bk_core.py:190-208) into the output tile.  Lq = log2 |G_raw|^2; the final soft clamp multiplies by
    // exp(g(L)), L = ln|G_raw| = (ln2 / 2) Lq, g(L) = 88 tanh(L/88) - L = -88 (u^3/3 - 2u^5/15 + 17u^7/315 ...),
    // u = L/88 = WALL_C Lq.  In base 2: log2(e) g = Lq^3 (EPI_A + EPI_B Lq^2); the dropped u^7 term is below 2e-6 of
    // g for |L| <= 10.  Elements with L outside [-8, 3.9] (the clip |G| > 50 is L > 3.9146; tiny |G|; anything not a
    // number) take the exact path of the scan kernel's epilogue.
    constexpr float EPI_A = -LOG2E * 88.0f * WALL_C * WALL_C * WALL_C / 3.0f;
    constexpr float EPI_B = LOG2E * 88.0f * 2.0f * WALL_C * WALL_C * WALL_C * WALL_C * WALL_C / 15.0f;
    constexpr float LQ_LO = -8.0f / LN2_HALF, LQ_HI = 3.9f / LN2_HALF;
    BK_STAMP(8);
    float2 *slot = &sh.u.out_s[pad16(seg * E)];
    bool rare = false;
#pragma unroll
    for (int e = 0; e < E; ++e) {
      float2 P;
      if constexpr (SAVE) P = to_f2(cmulr(lc[e], Fin.m));
      else P = to_f2(cmulr(ir[e], Fin.m));
      const int Ee = Fin.e + le[e >> 2];
      const float Ef = (float)Ee;
      const float n = fmaf(P.x, P.x, P.y * P.y);
      const float Lq = fmaf(2.0f, Ef, __builtin_amdgcn_logf(n));
      float2 gv;
      if (SAVE) {
        gv = (fabsf(LN2_HALF * Lq) <= 41.4f) ? make_float2(ldexpf(P.x, Ee), ldexpf(P.y, Ee)) : make_float2(0.f, 0.f);
      } else {
        const float t2 = Lq * Lq;
        const float sc = __builtin_amdgcn_exp2f(fmaf(t2 * Lq, fmaf(t2, EPI_B, EPI_A), Ef));
        gv = make_float2(P.x * sc, P.y * sc);
        rare |= !(fabsf(Lq - 0.5f * (LQ_HI + LQ_LO)) <= 0.5f * (LQ_HI - LQ_LO));
      }
      slot[e] = gv;
    }
    if (!SAVE && __builtin_expect(__any(rare), 0)) {
      // some element of this wave is outside the short series' range: the lanes redo their elements exactly
#pragma unroll
      for (int e = 0; e < E; ++e) {
        float2 P;
        if constexpr (SAVE) P = to_f2(cmulr(lc[e], Fin.m));
        else P = to_f2(cmulr(ir[e], Fin.m));
        const int Ee = Fin.e + le[e >> 2];
        const float n = fmaf(P.x, P.x, P.y * P.y);
        const float L = LN2_HALF * fmaf(2.0f, (float)Ee, __builtin_amdgcn_logf(n));  // ln |G_raw|
        const float sx = ldexpf(__builtin_amdgcn_exp2f(wall_exponent(L) * LOG2E), Ee);
        float2 gv = make_float2(P.x * sx, P.y * sx);
        const float m2 = fmaf(gv.x, gv.x, gv.y * gv.y);
        if (!(m2 <= 2500.0f)) {
          const float u = 50.0f * __builtin_amdgcn_rsqf(n);
          gv = (L > 3.9f && L <= 3.0e38f && u <= 3.0e38f) ? make_float2(P.x * u, P.y * u) : make_float2(0.f, 0.f);
        }
        slot[e] = fix_tiny(gv);
      }
    }
    BK_STAMP(9);
#if BK_PRIO == 3
    __builtin_amdgcn_s_setprio(0);
#endif
    __syncthreads();
#if BK_PRIO == 4
    __builtin_amdgcn_s_setprio(3);
#endif
    BK_STAMP(10);
    float2 *__restrict__ Gc = G + base;
    float2 *__restrict__ Fc = F ? F + base : nullptr;
    const int n_here = TAIL ? (int)max(0L, min((long)RCHUNK, N - base)) : RCHUNK;
    const bool clampf = cfg.fclamp <= 3.0e38f;  // uniform
    // (the three cases as loops of their own: with the tests on Fc and clampf inside one loop the compiler kept a
    // scalar branch, selects and a second address computation at every store)
    auto store_tile = [&](auto with_f, auto clamped) __attribute__((always_inline)) {
#pragma unroll
      for (int j = 0; j < E; ++j) {
        const int idx = t + j * RT;
        if (!TAIL || idx < n_here) {
          const float2 gv = sh.u.out_s[(t + (t >> 4)) + j * (RT + RT / 16)];  // pad16(idx)
          __builtin_nontemporal_store(gv.x, &Gc[idx].x);
          __builtin_nontemporal_store(gv.y, &Gc[idx].y);
          if constexpr (decltype(with_f)::value) {
            if constexpr (decltype(clamped)::value) {
              __builtin_nontemporal_store(__builtin_amdgcn_fmed3f(gv.x, -cfg.fclamp, cfg.fclamp), &Fc[idx].x);
              __builtin_nontemporal_store(__builtin_amdgcn_fmed3f(gv.y, -cfg.fclamp, cfg.fclamp), &Fc[idx].y);
            } else {
              __builtin_nontemporal_store(gv.x, &Fc[idx].x);
              __builtin_nontemporal_store(gv.y, &Fc[idx].y);
            }
          }
        }
      }
    };
    if (!Fc) store_tile(BoolC<false>{}, BoolC<false>{});
    else if (clampf) store_tile(BoolC<true>{}, BoolC<true>{});
    else store_tile(BoolC<true>{}, BoolC<false>{});
    BK_STAMP(11);
  };
#if BK_RATIO_STAMPS
  if (threadIdx.x == 0)
    reinterpret_cast<unsigned long long *>(ws + rowflag_bytes(B))[(row * nch) * 16 + 14] = wall_clock64();
  if (threadIdx.x == 0)
    reinterpret_cast<unsigned long long *>(ws + rowflag_bytes(B))[(row * nch) * 16 + 13] =
        ((unsigned long long)__builtin_amdgcn_s_getreg(0xF814) << 32) | (unsigned)__builtin_amdgcn_s_getreg(0xF804);
#endif
#if BK_RATIO_STAGGER > 0
  // development: start the workgroups at different phases of the chunk period
  for (int i = (int)((blockIdx.x * 2654435761u) >> 28); i > 0; --i) __builtin_amdgcn_s_sleep(BK_RATIO_STAGGER);
#endif
#if BK_UNIT_STAMPS
  unsigned long long *ustamp = reinterpret_cast<unsigned long long *>(ws + rowflag_bytes(B) + (8u << 20)) + 4 * (size_t)(RELAY ? unit : blockIdx.x);
  if (threadIdx.x == 0) {
    ustamp[0] = wall_clock64();
    ustamp[2] = ((unsigned long long)__builtin_amdgcn_s_getreg(0xF814) << 32) | (unsigned)__builtin_amdgcn_s_getreg(0xF804);
  }
#endif
  // chunk kinds: those within reach of a halo of the end of the row (the last one or two), then the others
  // (whole rows at three workgroups per CU only: the same few lines in the kernel for rows in units made the compiler
  // spill 31 registers in its chunk loop)
  if constexpr (MODE == 0 && !SAVE && !BST && !PREFETCH) {
    int ch = ch1 - 1;
    bool stage_only = skip_arith;
    if (!skip_arith) {
      for (; ch >= ch0 && (long)(ch + 1) * RCHUNK + RH + 1 > N; --ch) one(BoolC<true>{}, ch);
      // A row whose chunks at the right end have failed their checks already is lost to the scan kernel whatever the
      // rest does: it is only staged from here on (the NaN rules of the operator concern the whole batch, so every
      // element still has to be looked at).  The decision depends on the row's own data alone.
      if (ch >= ch0) stage_only = __syncthreads_or(bad ? 1 : 0) != 0;
      if (!stage_only)
        for (; ch >= ch0; --ch) one(BoolC<false>{}, ch);
    }
    if (stage_only) {  // uniform over the workgroup: stage the chunks for their NaN verdicts, nothing else
#pragma unroll 1
      for (; ch >= ch0; --ch) {
        prefetch(ch);
        __syncthreads();
        seen |= commit_ratio(cfg, av, bv, cv, sh);
        __syncthreads();
      }
      bad = true;
    }
  } else if (skip_arith) {  // uniform over the grid: stage every chunk for its NaN verdicts, nothing else
#pragma unroll 1
    for (int ch = ch1 - 1; ch >= ch0; --ch) {
      prefetch(ch);
      __syncthreads();
      seen |= commit_ratio(cfg, av, bv, cv, sh);
      __syncthreads();
    }
    bad = true;
  } else {
    if (PREFETCH) prefetch(ch1 - 1);
    int ch = ch1 - 1;
    for (; ch >= ch0 && (long)(ch + 1) * RCHUNK + RH + 1 > N; --ch) one(BoolC<true>{}, ch);
    for (; ch >= ch0; --ch) one(BoolC<false>{}, ch);
  }
#if BK_RATIO_STAMPS
  if (threadIdx.x == 0)
    reinterpret_cast<unsigned long long *>(ws + rowflag_bytes(B))[(row * nch) * 16 + 15] = wall_clock64();
#endif
  const int any_bad = __syncthreads_or(bad ? 1 : 0);
#if BK_UNIT_STAMPS
  if (threadIdx.x == 0) ustamp[1] = wall_clock64();
#endif
  if (t == 0) {
    if (MODE == 0) {
      rowflag[row] = any_bad;
      if (any_bad) __hip_atomic_store(&state[2], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if (RELAY) {
      // (which unit this is, worked out again rather than kept in scalar registers across the chunk loop: the kernel
      // has none to spare)
      const long row = (long)(unit % (unsigned)B);
      const int sg = S - 1 - (int)(unit / (unsigned)B);
      const int epoch = __hip_atomic_load(&state[ST_EPOCH], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      float *relay = reinterpret_cast<float *>(ws + rowflag_bytes(B)) + (row * S + sg) * RELAY_REC;
      // The row's verdict: the rightmost segment stores it before it publishes its record, the others have acquired
      // that record (directly or through the segments between) before they add to it.
      if (sg == S - 1) __hip_atomic_store(&rowflag[row], any_bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else if (any_bad) __hip_atomic_fetch_or(&rowflag[row], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (any_bad) __hip_atomic_store(&state[2], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (sg > 0 && !(flags & BKCORE_FLAG_ROW_UNITS_MUTE)) {
        const Prod p = sh.carry[nscan & 1];
        const float4 ed = sh.edge[(nscan & 1) ^ 1];  // written in the segment's last (leftmost) chunk
        __hip_atomic_store(relay + 0, (float)p.m.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(relay + 1, (float)p.m.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(relay + 2, __int_as_float(p.e), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(relay + 4, ed.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(relay + 5, ed.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(relay + 6, ed.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(relay + 7, ed.w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the record (and the row's verdict) before its tag
        __hip_atomic_store(reinterpret_cast<unsigned long long *>(relay + 8), relay_tag(epoch, row * S + sg),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      // (the epoch moves on in the call's last launch, which every MODE 3 launch is followed by)
    } else if (LOOK) {
      if (any_bad) {
        __hip_atomic_fetch_or(&rowflag[row], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&state[2], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
  if (seen) {
    if (seen & 1u) __hip_atomic_store(&state[0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (seen & 2u) __hip_atomic_store(&state[1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (!PERSIST) break;
  if (paired) {
    const unsigned w = blockIdx.x, G = gridDim.x, E = (unsigned)B - G, Bu = (unsigned)B;
    if (unit == G + w) unit = w;                                   // (w < E) the extra right half is done: own row
    else if (unit == w) unit = Bu + w;                             // own right half -> own left half
    else if (unit == Bu + w && w >= E && w < 2u * E) unit = Bu + G + (w - E);  // then the extra left half
    else break;
  } else {
    unit += gridDim.x;
    if (unit >= units) break;
  }
  }  // units
}

}  // namespace

namespace {
template <typename TIn, bool PHI_EXACT>
void ratio_launch(int mode, int save, const void *a, long lda, const void *b, long ldb, const void *c, long ldc,
                  const float *z_dev, float z_re, float z_im, long B, long N, float *G, float *features, void *ws,
                  void *state, unsigned flags, float fclamp, int S, int seg_ch, float *save_r,
                  float *save_s, hipStream_t stream) {
#define BK_RATIO(MODE, SAVE, GRID) BK_RATIO2(MODE, SAVE, false, GRID)
#define BK_RATIO2(MODE, SAVE, BSTV, GRID)                                                                            \
  hipLaunchKernelGGL((bk_ratio_forward_kernel<TIn, PHI_EXACT, MODE, SAVE, BSTV>), dim3((unsigned)(GRID)), dim3(RT), 0, \
                     stream, (const TIn *)a, lda, (const TIn *)b, ldb, (const TIn *)c, ldc, z_dev, z_re, z_im, B, N, \
                     (float2 *)G, (float2 *)features, (char *)ws, (int *)state, flags, fclamp, S, seg_ch,   \
                     (float2 *)save_r, (float2 *)save_s)
  // save_r given with mode 3 / 4, or save == 2 with mode 0: the variant that also stores the boundary states there
  // (float32 operands, float32 chains)
#if !BK_RATIO_LONG
  if constexpr (std::is_same<TIn, float>::value) {
    if (save_r != nullptr && (mode == 4 || mode == 3 || (mode == 0 && save == 2))) {
      if (mode == 4) BK_RATIO2(4, false, true, B * S);
      else if (mode == 3) BK_RATIO2(3, false, true, (BK_RELAY_PERSIST && save > 0 && save < B * S) ? save : B * S);
      else BK_RATIO2(0, false, true, B);
      return;
    }
  }
#endif
  if (mode == 4) BK_RATIO(4, false, B * S);
#if !BK_RATIO_LONG
  else if (mode == 3) BK_RATIO(3, false, (BK_RELAY_PERSIST && save > 0 && save < B * S) ? save : B * S);  // save: places
#endif
  else if (!save) BK_RATIO(0, false, B);
#if !BK_RATIO_LONG
  else if (std::is_same<TIn, float>::value) BK_RATIO(0, true, B);  // the saving variant exists for float32 operands
#endif
#undef BK_RATIO
#undef BK_RATIO2
}
}  // namespace

#if BK_RATIO_LONG
#define BK_RATIO_ENTRY(name) name##_long
#else
#define BK_RATIO_ENTRY(name) name
#endif

extern "C" void BK_RATIO_ENTRY(bkcore_internal_ratio_forward)(int in_dtype, int phi_exact, int mode, int save, const void *a,
                                              int64_t lda, const void *b, int64_t ldb, const void *c, int64_t ldc,
                                              const float *z_dev, float z_re, float z_im, int64_t B, int64_t N, float *G,
                                              float *features, void *ws, void *state, unsigned flags,
                                              float fclamp, int S, int seg_ch, float *save_r, float *save_s,
                                              void *stream) {
#define BK_DISPATCH(TIN)                                                                                             \
  do {                                                                                                               \
    if (phi_exact)                                                                                                   \
      ratio_launch<TIN, true>(mode, save, a, (long)lda, b, (long)ldb, c, (long)ldc, z_dev, z_re, z_im, (long)B,      \
                              (long)N, G, features, ws, state, flags, fclamp, S, seg_ch, save_r, save_s,    \
                              (hipStream_t)stream);                                                                  \
    else                                                                                                             \
      ratio_launch<TIN, false>(mode, save, a, (long)lda, b, (long)ldb, c, (long)ldc, z_dev, z_re, z_im, (long)B,     \
                               (long)N, G, features, ws, state, flags, fclamp, S, seg_ch, save_r, save_s,   \
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

#if !BK_RATIO_LONG
extern "C" int bkcore_internal_ratio_chunk(void) { return RCHUNK; }

// Workgroups of the MODE 3 kernel that one CU holds at once on the current device, as the runtime computes it from the
// kernel's registers and LDS (3 for float32 and float64 operands, 2 for 16-bit ones on the GPUs this was written for);
// 0 if the query fails.  The grid of MODE 3 must not exceed CUs x this: its workgroups wait for each other.
extern "C" int bkcore_internal_ratio_unit_occupancy(int in_dtype, int phi_exact) {
  static int cache[64][4][2] = {};  // per device: GPUs of one process need not be alike
  int dev = 0;
  if (in_dtype < 0 || in_dtype > 3) return 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) {
    (void)hipGetLastError();
    return 0;
  }
  int &slot = cache[dev][in_dtype][phi_exact ? 1 : 0];
  if (slot == 0) {
    int n = 0;
    hipError_t e = hipErrorInvalidValue;
#define BK_OCC(TIN)                                                                                                   \
  e = phi_exact ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, bk_ratio_forward_kernel<TIN, true, 3, false>, RT, 0) \
                : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, bk_ratio_forward_kernel<TIN, false, 3, false>, RT, 0)
    switch (in_dtype) {
      case BKCORE_F32: BK_OCC(float); break;
#ifndef BK_DEV_F32_ONLY
      case BKCORE_BF16: BK_OCC(uint16_t); break;
      case BKCORE_F16: BK_OCC(_Float16); break;
      case BKCORE_F64: BK_OCC(double); break;
#endif
      default: break;
    }
#undef BK_OCC
    slot = (e == hipSuccess && n > 0) ? n : -1;
    if (e != hipSuccess) (void)hipGetLastError();  // not an error of the caller's launch
  }
  return slot > 0 ? slot : 0;
}
#endif

// Cumulative sum along one dimension of a contiguous float32 tensor viewed as (outer, L, inner): the causal
// accumulation of the reference's AR-SSM layer (src/models/phase1/ar_ssm_layer.py:517-525, k_cumsum over the sequence
// dimension of (B, L, rank)) and its drop-in `fused_associative_scan(x, dim, reverse)` (src/kernels/associative_scan.py:
// 211-308).  HBM-bound: 4 bytes read + 4 written per element, every element touched once.
//   inner > 1  (the AR-SSM case): bk_cumsum_strided_kernel, lanes along `inner` (coalesced), 8 segments of the
//              sequence per workgroup pass, running carry per column in a register.  When there are too few columns
//              to occupy the chip (B = 8, rank = 32: 8 workgroups) the sequence is cut into S parts, one workgroup
//              each: a part first SUMS everything before it (plain independent loads, no barriers; the tensor is
//              small enough to sit in L2 whenever this path is taken), then scans its own rows from that carry.
//   inner == 1 (scan along the contiguous dimension): bk_cumsum_row_kernel, one workgroup per row, 4 elements per
//              thread, wave scan by DPP moves, wave totals and the running carry through LDS.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/bkcore.h"
#include "bkcore_internal.h"

// BKCORE_OK, or BKCORE_ELAUNCH with the HIP code recorded for bkcore_last_hip_error()
static inline int bk_launch_status() {
  const hipError_t e = hipGetLastError();
  if (e == hipSuccess) return BKCORE_OK;
  bkcore_internal_set_hip_error((int)e);
  return BKCORE_ELAUNCH;
}

namespace {

constexpr int ST = 256;   // threads per workgroup
constexpr int TC = 32;    // columns per workgroup (strided kernel)
constexpr int NSEG = ST / TC;
constexpr int SEG = 16;   // consecutive sequence positions per thread

template <bool REVERSE>
__global__ __launch_bounds__(ST) void bk_cumsum_strided_kernel(const float *__restrict__ x, float *__restrict__ y,
                                                               long L, long inner, long col_tiles, int S,
                                                               long part_len) {
  __shared__ float tot[2][NSEG][TC];
  const int part = blockIdx.x % S;
  const long tile = blockIdx.x / S;
  const long o = tile / col_tiles;
  const long c = (tile % col_tiles) * TC + (threadIdx.x % TC);
  const int sg = threadIdx.x / TC;
  const bool live = c < inner;
  const float *__restrict__ xp = x + o * L * inner + (live ? c : 0);
  float *__restrict__ yp = y + o * L * inner + (live ? c : 0);
  const long l_begin = (long)part * part_len, l_end = min(L, l_begin + part_len);
  float carry = 0.0f;
  int buf = 0;
  if (part > 0) {  // sum of the column over all rows before this part (uniform branch)
    constexpr int PF = 16;  // independent loads in flight per thread
    float acc[PF];
#pragma unroll
    for (int q = 0; q < PF; ++q) acc[q] = 0.0f;
    long l = sg;
    for (; l + (PF - 1) * NSEG < l_begin; l += PF * NSEG) {
      if (live) {
        float ld[PF];
#pragma unroll
        for (int q = 0; q < PF; ++q) {
          const long lq = l + q * NSEG;
          ld[q] = xp[(REVERSE ? L - 1 - lq : lq) * inner];
        }
#pragma unroll
        for (int q = 0; q < PF; ++q) acc[q] += ld[q];
      }
    }
    for (; l < l_begin; l += NSEG)
      if (live) acc[0] += xp[(REVERSE ? L - 1 - l : l) * inner];
#pragma unroll
    for (int q = PF / 2; q >= 1; q >>= 1)
#pragma unroll
      for (int e = 0; e < q; ++e) acc[e] += acc[e + q];
    tot[0][sg][threadIdx.x % TC] = acc[0];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NSEG; ++q) carry += tot[0][q][threadIdx.x % TC];
    buf = 1;  // tot[0] is written again two passes from now, after two more barriers
  }
  for (long l0 = l_begin; l0 < l_end; l0 += NSEG * SEG) {
    float v[SEG];
    const long lb = l0 + (long)sg * SEG;
#pragma unroll
    for (int j = 0; j < SEG; ++j) {
      const long l = lb + j;
      const long lp = REVERSE ? L - 1 - l : l;
      v[j] = (live && l < l_end) ? xp[lp * inner] : 0.0f;
    }
#pragma unroll
    for (int j = 1; j < SEG; ++j) v[j] += v[j - 1];
    tot[buf][sg][threadIdx.x % TC] = v[SEG - 1];
    __syncthreads();
    float before = carry, all = 0.0f;
#pragma unroll
    for (int s = 0; s < NSEG; ++s) {
      const float ts = tot[buf][s][threadIdx.x % TC];
      if (s < sg) before += ts;
      all += ts;
    }
    carry += all;
    buf ^= 1;  // the other buffer is free: every thread passed the barrier after reading it in the previous pass
#pragma unroll
    for (int j = 0; j < SEG; ++j) {
      const long l = lb + j;
      const long lp = REVERSE ? L - 1 - l : l;
      if (live && l < l_end) yp[lp * inner] = v[j] + before;
    }
  }
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add0(float x) {
  return x + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, ROW_MASK, 0xF, false));
}
__device__ __forceinline__ float wave_inclusive_scan(float x) {
  x = dpp_add0<0x111, 0xF>(x);  // row_shr:1
  x = dpp_add0<0x112, 0xF>(x);  // row_shr:2
  x = dpp_add0<0x114, 0xF>(x);  // row_shr:4
  x = dpp_add0<0x118, 0xF>(x);  // row_shr:8
  x = dpp_add0<0x142, 0xA>(x);  // row_bcast:15
  x = dpp_add0<0x143, 0xC>(x);  // row_bcast:31
  return x;
}

typedef float bk_sv4 __attribute__((ext_vector_type(4)));

// A pass covers RQ blocks of ST quads (4 consecutive elements per thread and block, so that a wave's access is one
// contiguous kilobyte), scanned side by side with one barrier per pass.  VEC: rows start 16-byte aligned and L is a
// multiple of 4, so a quad is one 16-byte access (everything is read once and written once: nontemporal).  In reverse
// mode scan position l is memory position L - 1 - l: a quad's elements are taken in descending order.
template <bool REVERSE, bool VEC, int RQ>
__global__ __launch_bounds__(ST) void bk_cumsum_row_kernel(const float *__restrict__ x, float *__restrict__ y, long L) {
  __shared__ float wtot[2][RQ][ST / 64];
  const float *__restrict__ xp = x + (long)blockIdx.x * L;
  float *__restrict__ yp = y + (long)blockIdx.x * L;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  float carry = 0.0f;
  int buf = 0;
  for (long l0 = 0; l0 < L; l0 += (long)ST * 4 * RQ) {
    float v[RQ][4], incl[RQ];
#pragma unroll
    for (int q = 0; q < RQ; ++q) {
      const long l = l0 + (long)q * ST * 4 + (long)t * 4;  // first scan position of this thread's quad in block q
      if (VEC && l + 3 < L) {
        const bk_sv4 w = __builtin_nontemporal_load(reinterpret_cast<const bk_sv4 *>(xp + (REVERSE ? L - 4 - l : l)));
        v[q][0] = REVERSE ? w[3] : w[0];
        v[q][1] = REVERSE ? w[2] : w[1];
        v[q][2] = REVERSE ? w[1] : w[2];
        v[q][3] = REVERSE ? w[0] : w[3];
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[q][j] = l + j < L ? xp[REVERSE ? L - 1 - (l + j) : l + j] : 0.0f;
      }
    }
#pragma unroll
    for (int q = 0; q < RQ; ++q) {
      v[q][1] += v[q][0];
      v[q][2] += v[q][1];
      v[q][3] += v[q][2];
      incl[q] = wave_inclusive_scan(v[q][3]);
      if (lane == 63) wtot[buf][q][wave] = incl[q];
    }
    __syncthreads();
    float running = carry;
#pragma unroll
    for (int q = 0; q < RQ; ++q) {
      // exclusive prefix of the lane inside its wave: the inclusive scan moved one lane up (0 into lane 0), not
      // incl - v[3], which is inf - inf = NaN once the lane's own group holds an infinity and loses the prefix by
      // cancellation when the group is much larger than what came before it
      const float excl =
          __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(incl[q]), 0x138, 0xF, 0xF, false));
      float before = running + excl, all = 0.0f;
#pragma unroll
      for (int w = 0; w < ST / 64; ++w) {
        const float tw = wtot[buf][q][w];
        if (w < wave) before += tw;
        all += tw;
      }
      running += all;
      const long l = l0 + (long)q * ST * 4 + (long)t * 4;
      if (VEC && l + 3 < L) {
        bk_sv4 w;
        w[REVERSE ? 3 : 0] = v[q][0] + before;
        w[REVERSE ? 2 : 1] = v[q][1] + before;
        w[REVERSE ? 1 : 2] = v[q][2] + before;
        w[REVERSE ? 0 : 3] = v[q][3] + before;
        __builtin_nontemporal_store(w, reinterpret_cast<bk_sv4 *>(yp + (REVERSE ? L - 4 - l : l)));
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (l + j < L) yp[REVERSE ? L - 1 - (l + j) : l + j] = v[q][j] + before;
      }
    }
    carry = running;
    buf ^= 1;
  }
}

}  // namespace

extern "C" int bkcore_cumsum(const float *x, float *y, int64_t outer, int64_t L, int64_t inner, int reverse,
                             void *stream) {
  if (outer < 0 || L < 0 || inner < 0) return BKCORE_EINVAL;
  if (outer == 0 || L == 0 || inner == 0) return BKCORE_OK;
  if (!x || !y) return BKCORE_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (inner == 1) {
    if (outer > 2147483647LL) return BKCORE_EINVAL;
    // 16-byte accesses need aligned rows; long rows take four blocks per pass (a quarter of the barriers)
    const bool vec = (L % 4) == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)y % 16) == 0;
#define BK_ROW(REV, VEC, RQ)                                                                                         \
  hipLaunchKernelGGL((bk_cumsum_row_kernel<REV, VEC, RQ>), dim3((unsigned)outer), dim3(ST), 0, s, x, y, (long)L)
    if (L >= 4096) {
      if (reverse) { if (vec) BK_ROW(true, true, 4); else BK_ROW(true, false, 4); }
      else { if (vec) BK_ROW(false, true, 4); else BK_ROW(false, false, 4); }
    } else {
      if (reverse) { if (vec) BK_ROW(true, true, 1); else BK_ROW(true, false, 1); }
      else { if (vec) BK_ROW(false, true, 1); else BK_ROW(false, false, 1); }
    }
#undef BK_ROW
  } else {
    const int64_t col_tiles = (inner + TC - 1) / TC;
    if (outer * col_tiles > 2147483647LL / 64) return BKCORE_EINVAL;
    // few columns and a tensor that fits L2 (<= 16 MB): cut the sequence into S parts of whole passes
    int S = 1;
    const int64_t tiles = outer * col_tiles;
    if (tiles < 128 && outer * L * inner * 4 <= (16LL << 20)) {
      int64_t want = (256 + tiles - 1) / tiles;
      const int64_t max_parts = L / (4 * NSEG * SEG);  // at least four passes per part
      if (want > max_parts) want = max_parts;
      if (want > 64) want = 64;
      if (want >= 2) S = (int)want;
    }
    int64_t part_len = (L + S - 1) / S;
    part_len = (part_len + NSEG * SEG - 1) / (NSEG * SEG) * (NSEG * SEG);
    S = (int)((L + part_len - 1) / part_len);
    const dim3 grid((unsigned)(tiles * S));
    if (reverse)
      hipLaunchKernelGGL((bk_cumsum_strided_kernel<true>), grid, dim3(ST), 0, s, x, y, (long)L, (long)inner,
                         (long)col_tiles, S, (long)part_len);
    else
      hipLaunchKernelGGL((bk_cumsum_strided_kernel<false>), grid, dim3(ST), 0, s, x, y, (long)L, (long)inner,
                         (long)col_tiles, S, (long)part_len);
  }
  return bk_launch_status();
}

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

template <bool REVERSE>
__global__ __launch_bounds__(ST) void bk_cumsum_row_kernel(const float *__restrict__ x, float *__restrict__ y, long L) {
  __shared__ float wtot[2][ST / 64];
  const float *__restrict__ xp = x + (long)blockIdx.x * L;
  float *__restrict__ yp = y + (long)blockIdx.x * L;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  float carry = 0.0f;
  int buf = 0;
  for (long l0 = 0; l0 < L; l0 += ST * 4) {
    const long lb = l0 + (long)t * 4;
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const long l = lb + j;
      v[j] = l < L ? xp[REVERSE ? L - 1 - l : l] : 0.0f;
    }
    v[1] += v[0];
    v[2] += v[1];
    v[3] += v[2];
    const float incl = wave_inclusive_scan(v[3]);
    if (lane == 63) wtot[buf][wave] = incl;
    __syncthreads();
    // exclusive prefix of the lane inside its wave: the inclusive scan moved one lane up (0 into lane 0), not
    // incl - v[3], which is inf - inf = NaN once the lane's own group holds an infinity and loses the prefix by
    // cancellation when the group is much larger than what came before it
    const float excl = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(incl), 0x138, 0xF, 0xF, false));
    float before = carry + excl, all = 0.0f;
#pragma unroll
    for (int w = 0; w < ST / 64; ++w) {
      const float tw = wtot[buf][w];
      if (w < wave) before += tw;
      all += tw;
    }
    carry += all;
    buf ^= 1;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const long l = lb + j;
      if (l < L) yp[REVERSE ? L - 1 - l : l] = v[j] + before;
    }
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
    if (reverse)
      hipLaunchKernelGGL((bk_cumsum_row_kernel<true>), dim3((unsigned)outer), dim3(ST), 0, s, x, y, (long)L);
    else
      hipLaunchKernelGGL((bk_cumsum_row_kernel<false>), dim3((unsigned)outer), dim3(ST), 0, s, x, y, (long)L);
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

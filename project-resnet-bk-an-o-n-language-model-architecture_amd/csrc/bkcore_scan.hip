// Cumulative sum along one dimension of a contiguous float32 tensor viewed as (outer, L, inner): the causal
// accumulation of the reference's AR-SSM layer (src/models/phase1/ar_ssm_layer.py:517-525, k_cumsum over the sequence
// dimension of (B, L, rank)) and its drop-in `fused_associative_scan(x, dim, reverse)` (src/kernels/associative_scan.py:
// 211-308).  HBM-bound: 4 bytes read + 4 written per element, every element touched once.
//   inner > 1  (the AR-SSM case): bk_cumsum_strided_kernel, lanes along `inner` (coalesced), 8 segments of the
//              sequence per workgroup pass, running carry per column in a register.
//   inner == 1 (scan along the contiguous dimension): bk_cumsum_row_kernel, one workgroup per row, 4 elements per
//              thread, wave scan by DPP moves, wave totals and the running carry through LDS.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/bkcore.h"

namespace {

constexpr int ST = 256;   // threads per workgroup
constexpr int TC = 32;    // columns per workgroup (strided kernel)
constexpr int NSEG = ST / TC;
constexpr int SEG = 8;    // consecutive sequence positions per thread

template <bool REVERSE>
__global__ __launch_bounds__(ST) void bk_cumsum_strided_kernel(const float *__restrict__ x, float *__restrict__ y,
                                                               long L, long inner, long col_tiles) {
  __shared__ float tot[2][NSEG][TC];
  const long o = blockIdx.x / col_tiles;
  const long c = (blockIdx.x % col_tiles) * TC + (threadIdx.x % TC);
  const int sg = threadIdx.x / TC;
  const bool live = c < inner;
  const float *__restrict__ xp = x + o * L * inner + (live ? c : 0);
  float *__restrict__ yp = y + o * L * inner + (live ? c : 0);
  float carry = 0.0f;
  int buf = 0;
  for (long l0 = 0; l0 < L; l0 += NSEG * SEG) {
    float v[SEG];
    const long lb = l0 + (long)sg * SEG;
#pragma unroll
    for (int j = 0; j < SEG; ++j) {
      const long l = lb + j;
      const long lp = REVERSE ? L - 1 - l : l;
      v[j] = (live && l < L) ? xp[lp * inner] : 0.0f;
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
      if (live && l < L) yp[lp * inner] = v[j] + before;
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
    float before = carry + (incl - v[3]), all = 0.0f;
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
    if (outer * col_tiles > 2147483647LL) return BKCORE_EINVAL;
    const dim3 grid((unsigned)(outer * col_tiles));
    if (reverse)
      hipLaunchKernelGGL((bk_cumsum_strided_kernel<true>), grid, dim3(ST), 0, s, x, y, (long)L, (long)inner,
                         (long)col_tiles);
    else
      hipLaunchKernelGGL((bk_cumsum_strided_kernel<false>), grid, dim3(ST), 0, s, x, y, (long)L, (long)inner,
                         (long)col_tiles);
  }
  return hipGetLastError() == hipSuccess ? BKCORE_OK : BKCORE_ELAUNCH;
}

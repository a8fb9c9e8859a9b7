// The spectral branch's output stage of the reference layer (src/models/resnet_bk.py:218-219):
//     output = ffn_out + bk_scale * output_proj(features),   output_proj = Linear(2 -> D)
// as one pass forward (reads the 2 features and, if given, the D residual values per position; writes D values) and
// one pass backward (reads the D incoming gradients per position once: gradient of the features by a reduction over D,
// gradients of weight / bias / bk_scale by column sums over all positions, two-stage and deterministic).
// HBM-bound: 4 D bytes written (+ 4 D read) per position forward, 4 D read backward; the tensor-library chain
// (K = 2 GEMM, multiply, add, and their backward twins) moves about three times that.
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

constexpr int LT = 256;        // threads per workgroup
constexpr int MAX_CPT = 4;     // columns per thread in the backward: D <= LT * MAX_CPT
constexpr int ROWS_TILE = 16;  // positions whose feature gradients are reduced together

// ---- forward: float4 over D when D % 4 == 0, scalar otherwise
template <bool VEC, bool HAS_RES>
__global__ __launch_bounds__(LT) void bk_layer_epilogue_kernel(const float *__restrict__ feat,
                                                               const float *__restrict__ W,
                                                               const float *__restrict__ bias,
                                                               const float *__restrict__ scale,
                                                               const float *__restrict__ res, long n, int D,
                                                               float fclamp, float *__restrict__ out) {
  constexpr int V = VEC ? 4 : 1;
  const int dv = D / V;  // vector slots per position
  const long total = n * (long)dv;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += stride) {
    const long i = idx / dv;
    const int d0 = (int)(idx - i * dv) * V;
    // (fclamp = +inf: the features as they are)
    const float f0 = __builtin_amdgcn_fmed3f(feat[2 * i], -fclamp, fclamp);
    const float f1 = __builtin_amdgcn_fmed3f(feat[2 * i + 1], -fclamp, fclamp);
    float y[V];
#pragma unroll
    for (int q = 0; q < V; ++q) {
      const int d = d0 + q;
      const float t = fmaf(f1, W[2 * d + 1], f0 * W[2 * d]) + bias[d];
      y[q] = scale[d] * t;
    }
    if constexpr (VEC) {
      float4 o = make_float4(y[0], y[1], y[2], y[3]);
      if (HAS_RES) {
        const float4 r = *reinterpret_cast<const float4 *>(res + i * D + d0);
        o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
      }
      *reinterpret_cast<float4 *>(out + i * D + d0) = o;
    } else {
      out[i * D + d0] = HAS_RES ? res[i * D + d0] + y[0] : y[0];
    }
  }
}

// ---- backward, stage 1: each workgroup takes a contiguous range of positions.  Thread t owns columns t, t + 256, ...
// Per position: u = go * scale; the two feature gradients are sums over D of u W[d][j] (wave shuffles, then LDS across
// the 4 waves, ROWS_TILE positions per barrier); per column: running sums of go f0, go f1, go over the range.
__global__ __launch_bounds__(LT) void bk_layer_epilogue_bwd_kernel(const float *__restrict__ go,
                                                                   const float *__restrict__ feat,
                                                                   const float *__restrict__ W,
                                                                   const float *__restrict__ scale, long n, int D,
                                                                   long rows_per_block, float fclamp,
                                                                   float *__restrict__ gfeat,
                                                                   float *__restrict__ partials) {
  __shared__ float red[ROWS_TILE][LT / 64][2];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const long r0 = (long)blockIdx.x * rows_per_block;
  const long r1 = min(n, r0 + rows_per_block);
  float a0[MAX_CPT], a1[MAX_CPT], s0[MAX_CPT], s1[MAX_CPT], sb[MAX_CPT];
#pragma unroll
  for (int q = 0; q < MAX_CPT; ++q) {
    const int d = t + q * LT;
    const float sc = d < D ? scale[d] : 0.0f;
    a0[q] = d < D ? sc * W[2 * d] : 0.0f;
    a1[q] = d < D ? sc * W[2 * d + 1] : 0.0f;
    s0[q] = s1[q] = sb[q] = 0.0f;
  }
  for (long rt = r0; rt < r1; rt += ROWS_TILE) {
#pragma unroll 4
    for (int rr = 0; rr < ROWS_TILE; ++rr) {
      const long i = rt + rr;
      float p0 = 0.0f, p1 = 0.0f;
      if (i < r1) {  // uniform over the workgroup
        const float f0 = __builtin_amdgcn_fmed3f(feat[2 * i], -fclamp, fclamp);
        const float f1 = __builtin_amdgcn_fmed3f(feat[2 * i + 1], -fclamp, fclamp);
#pragma unroll
        for (int q = 0; q < MAX_CPT; ++q) {
          const int d = t + q * LT;
          if (d < D) {
            const float g = go[i * D + d];
            p0 = fmaf(g, a0[q], p0);
            p1 = fmaf(g, a1[q], p1);
            s0[q] = fmaf(g, f0, s0[q]);
            s1[q] = fmaf(g, f1, s1[q]);
            sb[q] += g;
          }
        }
      }
#pragma unroll
      for (int m = 32; m >= 1; m >>= 1) {
        p0 += __shfl_xor(p0, m);
        p1 += __shfl_xor(p1, m);
      }
      if (lane == 0) {
        red[rr][wave][0] = p0;
        red[rr][wave][1] = p1;
      }
    }
    __syncthreads();
    if (t < ROWS_TILE * 2) {
      const int rr = t >> 1, j = t & 1;
      const long i = rt + rr;
      // the clamp passes the gradient on [-fclamp, fclamp] only
      if (i < r1)
        gfeat[2 * i + j] = fabsf(feat[2 * i + j]) <= fclamp
                               ? (red[rr][0][j] + red[rr][1][j]) + (red[rr][2][j] + red[rr][3][j]) : 0.0f;
    }
    __syncthreads();
  }
  float *P = partials + (long)blockIdx.x * 3 * D;
#pragma unroll
  for (int q = 0; q < MAX_CPT; ++q) {
    const int d = t + q * LT;
    if (d < D) {
      P[d] = s0[q];
      P[D + d] = s1[q];
      P[2 * D + d] = sb[q];
    }
  }
}

// ---- backward, stage 1, D a multiple of 256: one WAVE per position.  Lane l owns columns 4 l .. 4 l + 3 of every
// block of 256 columns (16-byte loads, 1 KB per wave per block); the two feature gradients are reduced inside the wave
// with DPP moves (no LDS, no barrier); every wave keeps its own column sums.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float x) {  // x + (value of the selected lane, 0 where there is none)
  return x + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, ROW_MASK, 0xF, false));
}
__device__ __forceinline__ float wave_total_in_lane63(float x) {
  x = dpp_add<0x111, 0xF>(x);  // row_shr:1
  x = dpp_add<0x112, 0xF>(x);  // row_shr:2
  x = dpp_add<0x114, 0xF>(x);  // row_shr:4
  x = dpp_add<0x118, 0xF>(x);  // row_shr:8
  x = dpp_add<0x142, 0xA>(x);  // row_bcast:15
  x = dpp_add<0x143, 0xC>(x);  // row_bcast:31
  return x;
}

template <int Q>  // D = 256 Q
__global__ __launch_bounds__(LT) void bk_layer_epilogue_bwd_wave_kernel(const float *__restrict__ go,
                                                                        const float *__restrict__ feat,
                                                                        const float *__restrict__ W,
                                                                        const float *__restrict__ scale, long n,
                                                                        long rows_per_block, float fclamp,
                                                                        float *__restrict__ gfeat,
                                                                        float *__restrict__ partials) {
  constexpr int D = 256 * Q;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const long r0 = (long)blockIdx.x * rows_per_block;
  const long r1 = min(n, r0 + rows_per_block);
  float a0[Q][4], a1[Q][4], s0[Q][4], s1[Q][4], sb[Q][4];
#pragma unroll
  for (int q = 0; q < Q; ++q)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int d = 256 * q + 4 * lane + e;
      const float sc = scale[d];
      a0[q][e] = sc * W[2 * d];
      a1[q][e] = sc * W[2 * d + 1];
      s0[q][e] = s1[q][e] = sb[q][e] = 0.0f;
    }
  // two positions per pass, both rows' loads issued before either is used
  constexpr int WS = LT / 64;
  for (long i = r0 + wave; i < r1; i += 2 * WS) {
    const long i2 = i + WS;
    const bool two = i2 < r1;  // uniform over the wave
    const float r0f = feat[2 * i], r1f = feat[2 * i + 1];  // as stored; the sums take them clamped
    const float t0f = two ? feat[2 * i2] : 0.0f, t1f = two ? feat[2 * i2 + 1] : 0.0f;
    const float f0 = __builtin_amdgcn_fmed3f(r0f, -fclamp, fclamp), f1 = __builtin_amdgcn_fmed3f(r1f, -fclamp, fclamp);
    const float h0 = __builtin_amdgcn_fmed3f(t0f, -fclamp, fclamp), h1 = __builtin_amdgcn_fmed3f(t1f, -fclamp, fclamp);
    const float4 *__restrict__ row = reinterpret_cast<const float4 *>(go + i * D) + lane;
    const float4 *__restrict__ row2 = reinterpret_cast<const float4 *>(go + (two ? i2 : i) * D) + lane;
    float4 g[Q], g2[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) g[q] = row[64 * q];
#pragma unroll
    for (int q = 0; q < Q; ++q) g2[q] = row2[64 * q];
    float p0 = 0.0f, p1 = 0.0f, q0 = 0.0f, q1 = 0.0f;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const float ge[4] = {g[q].x, g[q].y, g[q].z, g[q].w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        p0 = fmaf(ge[e], a0[q][e], p0);
        p1 = fmaf(ge[e], a1[q][e], p1);
        s0[q][e] = fmaf(ge[e], f0, s0[q][e]);
        s1[q][e] = fmaf(ge[e], f1, s1[q][e]);
        sb[q][e] += ge[e];
      }
    }
    if (two) {
#pragma unroll
      for (int q = 0; q < Q; ++q) {
        const float ge[4] = {g2[q].x, g2[q].y, g2[q].z, g2[q].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          q0 = fmaf(ge[e], a0[q][e], q0);
          q1 = fmaf(ge[e], a1[q][e], q1);
          s0[q][e] = fmaf(ge[e], h0, s0[q][e]);
          s1[q][e] = fmaf(ge[e], h1, s1[q][e]);
          sb[q][e] += ge[e];
        }
      }
    }
    p0 = wave_total_in_lane63(p0);
    p1 = wave_total_in_lane63(p1);
    q0 = wave_total_in_lane63(q0);
    q1 = wave_total_in_lane63(q1);
    if (lane == 63) {  // the clamp passes the gradient on [-fclamp, fclamp] only
      *reinterpret_cast<float2 *>(gfeat + 2 * i) =
          make_float2(fabsf(r0f) <= fclamp ? p0 : 0.0f, fabsf(r1f) <= fclamp ? p1 : 0.0f);
      if (two)
        *reinterpret_cast<float2 *>(gfeat + 2 * i2) =
            make_float2(fabsf(t0f) <= fclamp ? q0 : 0.0f, fabsf(t1f) <= fclamp ? q1 : 0.0f);
    }
  }
  // the four waves' column sums are combined in LDS (fixed order) into one record per workgroup
  __shared__ float part[LT / 64][3 * D];
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    const int d = 256 * q + 4 * lane;
    *reinterpret_cast<float4 *>(&part[wave][d]) = make_float4(s0[q][0], s0[q][1], s0[q][2], s0[q][3]);
    *reinterpret_cast<float4 *>(&part[wave][D + d]) = make_float4(s1[q][0], s1[q][1], s1[q][2], s1[q][3]);
    *reinterpret_cast<float4 *>(&part[wave][2 * D + d]) = make_float4(sb[q][0], sb[q][1], sb[q][2], sb[q][3]);
  }
  __syncthreads();
  float *P = partials + (long)blockIdx.x * 3 * D;
#pragma unroll
  for (int q = 0; q < 3 * Q; ++q) {
    const int x = t + q * LT;
    P[x] = (part[0][x] + part[1][x]) + (part[2][x] + part[3][x]);
  }
}

// ---- backward, stage 2: the workgroups' partial column sums added in a fixed order
__global__ __launch_bounds__(LT) void bk_layer_epilogue_bwd_finish_kernel(const float *__restrict__ partials, int nblk,
                                                                          const float *__restrict__ W,
                                                                          const float *__restrict__ bias,
                                                                          const float *__restrict__ scale, int D,
                                                                          float *__restrict__ gW,
                                                                          float *__restrict__ gbias,
                                                                          float *__restrict__ gscale) {
  // 32 columns per workgroup; 8 groups of threads each sum every 8th record, then the groups are added in order
  __shared__ float red[LT / 32][3][32];
  const int c = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const int d = blockIdx.x * 32 + c;
  float S0 = 0.0f, S1 = 0.0f, Sb = 0.0f;
  if (d < D)
    for (int k = grp; k < nblk; k += LT / 32) {
      const float *P = partials + (long)k * 3 * D;
      S0 += P[d];
      S1 += P[D + d];
      Sb += P[2 * D + d];
    }
  red[grp][0][c] = S0;
  red[grp][1][c] = S1;
  red[grp][2][c] = Sb;
  __syncthreads();
  if (grp != 0 || d >= D) return;
  S0 = S1 = Sb = 0.0f;
#pragma unroll
  for (int k = 0; k < LT / 32; ++k) {
    S0 += red[k][0][c];
    S1 += red[k][1][c];
    Sb += red[k][2][c];
  }
  const float sc = scale[d];
  if (gW) {
    gW[2 * d] = sc * S0;
    gW[2 * d + 1] = sc * S1;
  }
  if (gbias) gbias[d] = sc * Sb;
  if (gscale) gscale[d] = fmaf(W[2 * d], S0, fmaf(W[2 * d + 1], S1, bias[d] * Sb));
}

inline int bwd_blocks(int64_t n) {
  int64_t nblk = (n + 4 * ROWS_TILE - 1) / (4 * ROWS_TILE);  // at least 64 positions per workgroup
  if (nblk > 1024) nblk = 1024;
  if (nblk < 1) nblk = 1;
  return (int)nblk;
}

}  // namespace

extern "C" {

int bkcore_layer_epilogue(const float *features, const float *weight, const float *bias, const float *scale,
                          const float *residual, int64_t n, int64_t D, float *out, void *stream) {
  return bkcore_layer_epilogue_clamped(features, 0.0f, weight, bias, scale, residual, n, D, out, stream);
}

int bkcore_layer_epilogue_clamped(const float *features, float feature_clamp, const float *weight, const float *bias,
                                  const float *scale, const float *residual, int64_t n, int64_t D, float *out,
                                  void *stream) {
  if (feature_clamp != feature_clamp) return BKCORE_EINVAL;
  const float fclamp = feature_clamp > 0.0f ? feature_clamp : __builtin_huge_valf();
  if (n < 0 || D < 0 || D > 2147483647LL / 4) return BKCORE_EINVAL;
  if (n == 0 || D == 0) return BKCORE_OK;
  if (!features || !weight || !bias || !scale || !out) return BKCORE_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const bool vec = (D % 4) == 0 && ((uintptr_t)out % 16) == 0 && (!residual || ((uintptr_t)residual % 16) == 0);
  const int64_t total = n * (vec ? D / 4 : D);
  int64_t blocks = (total + LT - 1) / LT;
  if (blocks > 16384) blocks = 16384;
  const dim3 grid((unsigned)blocks), block(LT);
#define BK_EPI(VEC, RES)                                                                                              \
  hipLaunchKernelGGL((bk_layer_epilogue_kernel<VEC, RES>), grid, block, 0, s, features, weight, bias, scale, residual, \
                     (long)n, (int)D, fclamp, out)
  if (vec && residual) BK_EPI(true, true);
  else if (vec) BK_EPI(true, false);
  else if (residual) BK_EPI(false, true);
  else BK_EPI(false, false);
#undef BK_EPI
  return bk_launch_status();
}

size_t bkcore_layer_epilogue_backward_workspace_bytes(int64_t n, int64_t D) {
  if (n <= 0 || D <= 0) return 0;
  return (size_t)bwd_blocks(n) * 3 * (size_t)D * sizeof(float);  // one record per workgroup
}

int bkcore_layer_epilogue_backward(const float *grad_out, const float *features, const float *weight, const float *bias,
                                   const float *scale, int64_t n, int64_t D, float *grad_features, float *grad_weight,
                                   float *grad_bias, float *grad_scale, void *workspace, size_t workspace_bytes,
                                   void *stream) {
  return bkcore_layer_epilogue_backward_clamped(grad_out, features, 0.0f, weight, bias, scale, n, D, grad_features,
                                                grad_weight, grad_bias, grad_scale, workspace, workspace_bytes, stream);
}

int bkcore_layer_epilogue_backward_clamped(const float *grad_out, const float *features, float feature_clamp,
                                           const float *weight, const float *bias, const float *scale, int64_t n,
                                           int64_t D, float *grad_features, float *grad_weight, float *grad_bias,
                                           float *grad_scale, void *workspace, size_t workspace_bytes, void *stream) {
  if (feature_clamp != feature_clamp) return BKCORE_EINVAL;
  const float fclamp = feature_clamp > 0.0f ? feature_clamp : __builtin_huge_valf();
  if (n < 0 || D < 0) return BKCORE_EINVAL;
  if (D > (int64_t)LT * MAX_CPT) return BKCORE_EINVAL;
  if (n == 0 || D == 0) return BKCORE_OK;
  if (!grad_out || !features || !weight || !bias || !scale || !grad_features) return BKCORE_EINVAL;
  const size_t need = bkcore_layer_epilogue_backward_workspace_bytes(n, D);
  if (!workspace || workspace_bytes < need) return BKCORE_EWORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  const int nblk = bwd_blocks(n);
  long rows = (long)((n + nblk - 1) / nblk);
  rows = (rows + ROWS_TILE - 1) / ROWS_TILE * ROWS_TILE;
  const bool wave_rows = (D % 256) == 0 && ((uintptr_t)grad_out % 16) == 0 && ((uintptr_t)workspace % 16) == 0 &&
                         ((uintptr_t)grad_features % 8) == 0;
  if (wave_rows) {
#define BK_BWD_WAVE(Q)                                                                                              \
  hipLaunchKernelGGL((bk_layer_epilogue_bwd_wave_kernel<Q>), dim3((unsigned)nblk), dim3(LT), 0, s, grad_out, features, \
                     weight, scale, (long)n, rows, fclamp, grad_features, (float *)workspace)
    switch ((int)(D / 256)) {
      case 1: BK_BWD_WAVE(1); break;
      case 2: BK_BWD_WAVE(2); break;
      case 3: BK_BWD_WAVE(3); break;
      default: BK_BWD_WAVE(4); break;
    }
#undef BK_BWD_WAVE
  } else {
    hipLaunchKernelGGL(bk_layer_epilogue_bwd_kernel, dim3((unsigned)nblk), dim3(LT), 0, s, grad_out, features, weight,
                       scale, (long)n, (int)D, rows, fclamp, grad_features, (float *)workspace);
  }
  if (grad_weight || grad_bias || grad_scale)
    hipLaunchKernelGGL(bk_layer_epilogue_bwd_finish_kernel, dim3((unsigned)((D + 31) / 32)), dim3(LT), 0, s,
                       (const float *)workspace, nblk, weight, bias, scale, (int)D, grad_weight, grad_bias,
                       grad_scale);
  return bk_launch_status();
}

}  // extern "C"

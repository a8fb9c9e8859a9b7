// Woodbury low-rank correction of the resolvent diagonal, batched over rows: the loop over the batch of the reference's
// Birman-Schwinger core (src/models/birman_schwinger_core.py:567-604) as three launches,
//     M_b   = I + V^T diag(G_b) U                     (r x r complex; bk_woodbury_gram_kernel, O(N r^2) per row)
//     W_b   = M_b^{-1}                                (bk_woodbury_solve_kernel, Gauss-Jordan with partial pivoting)
//     out_b = G_b - G_b^2 * rowsum((U W_b) o V)       (bk_woodbury_apply_kernel, O(N r^2) per row)
// in complex64 like the reference (its tridiagonal diagonal is complex64, :563).  r <= 22 (r = ceil(log2 N) in the
// reference: every N the operator accepts).
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

constexpr int WT = 256;       // threads per workgroup
constexpr int MAX_R = 22;     // r^2 <= 2 * WT
constexpr int TN = 64;        // positions staged per pass of the gram kernel
constexpr int GRAM_SPAN = 2048;  // positions per gram workgroup

__device__ __forceinline__ float2 cmulf(float2 a, float2 b) {
  return make_float2(fmaf(a.x, b.x, -a.y * b.y), fmaf(a.x, b.y, a.y * b.x));
}

// partial[b][s][k*r + l] = sum over the positions n of span s of V[n][k] U[n][l] G[b][n].
// A thread owns a strip (k; l0 .. l0+3): per position one 16-byte LDS read of U, one read of V, one of G for four
// (k, l) pairs.  The r * ceil(r/4) strips are replicated over as many groups of threads as fit the workgroup, each
// group taking every rep-th staged position; the groups' sums are added through LDS at the end.
constexpr int UROW = 24;  // floats per staged row of U (16-byte aligned quads, zero beyond r)
__global__ __launch_bounds__(WT) void bk_woodbury_gram_kernel(const float2 *__restrict__ G,
                                                              const float *__restrict__ U,
                                                              const float *__restrict__ V, long N, int r, int spans,
                                                              float2 *__restrict__ partial) {
  __shared__ __align__(16) float Us[TN][UROW];
  __shared__ float Vs[TN][MAX_R + 1];
  __shared__ float2 Gs[TN];
  __shared__ float2 red[WT][4];
  const long b = blockIdx.x / spans;
  const int sp = blockIdx.x % spans;
  const long n0 = (long)sp * GRAM_SPAN, n1 = min(N, n0 + GRAM_SPAN);
  const int t = threadIdx.x, rr = r * r;
  const int lq = (r + 3) / 4, nstr = r * lq;
  const int rep = WT / nstr;                 // >= 1 because r <= 22
  const int strip = t % nstr, sub = t / nstr;
  const bool live = sub < rep;
  const int k = strip / lq, l0 = (strip % lq) * 4;
  for (int i = t; i < TN * UROW; i += WT) (&Us[0][0])[i] = 0.0f;
  float2 acc[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) acc[e] = make_float2(0.f, 0.f);
  for (long nt = n0; nt < n1; nt += TN) {
    const int cnt = (int)min((long)TN, n1 - nt);
    __syncthreads();
    for (int i = t; i < cnt * r; i += WT) {
      Us[i / r][i % r] = U[nt * r + i];
      Vs[i / r][i % r] = V[nt * r + i];
    }
    if (t < cnt) Gs[t] = G[b * N + nt + t];
    __syncthreads();
    if (live)
      for (int j = sub; j < cnt; j += rep) {
        const float4 u4 = *reinterpret_cast<const float4 *>(&Us[j][l0]);
        const float vk = Vs[j][k];
        const float2 g = Gs[j];
        const float vgx = vk * g.x, vgy = vk * g.y;
        acc[0].x = fmaf(u4.x, vgx, acc[0].x); acc[0].y = fmaf(u4.x, vgy, acc[0].y);
        acc[1].x = fmaf(u4.y, vgx, acc[1].x); acc[1].y = fmaf(u4.y, vgy, acc[1].y);
        acc[2].x = fmaf(u4.z, vgx, acc[2].x); acc[2].y = fmaf(u4.z, vgy, acc[2].y);
        acc[3].x = fmaf(u4.w, vgx, acc[3].x); acc[3].y = fmaf(u4.w, vgy, acc[3].y);
      }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) red[t][e] = acc[e];
  __syncthreads();
  if (sub == 0) {
    float2 *P = partial + ((long)b * spans + sp) * rr;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float2 sum = acc[e];
      for (int q = 1; q < rep; ++q) {
        sum.x += red[q * nstr + strip][e].x;
        sum.y += red[q * nstr + strip][e].y;
      }
      if (l0 + e < r) P[k * r + l0 + e] = sum;
    }
  }
}

// One wave per row: M = I + sum of the partials, inverse by Gauss-Jordan on [M | I] with partial pivoting.
// A row whose M is singular to working precision gets a zero inverse (no correction) and flag[b] = 1.
__global__ __launch_bounds__(64) void bk_woodbury_solve_kernel(const float2 *__restrict__ partial, int r, int spans,
                                                               float2 *__restrict__ inv, int *__restrict__ flag) {
  __shared__ float2 A[MAX_R][2 * MAX_R];
  __shared__ float2 fac[MAX_R];
  __shared__ int piv;
  const long b = blockIdx.x;
  const int lane = threadIdx.x, rr = r * r, w = 2 * r;
  for (int i = lane; i < rr; i += 64) {
    float2 s = make_float2(0.f, 0.f);
    for (int sp = 0; sp < spans; ++sp) {
      const float2 v = partial[((long)b * spans + sp) * rr + i];
      s.x += v.x;
      s.y += v.y;
    }
    const int k = i / r, l = i % r;
    if (k == l) s.x += 1.0f;
    A[k][l] = s;
    A[k][r + l] = make_float2(k == l ? 1.0f : 0.0f, 0.0f);
  }
  __syncthreads();
  bool singular = false;
  for (int c = 0; c < r; ++c) {
    if (lane == 0) {
      int best = c;
      float bm = -1.0f;
      for (int i = c; i < r; ++i) {
        const float m = fmaf(A[i][c].x, A[i][c].x, A[i][c].y * A[i][c].y);
        if (m > bm) { bm = m; best = i; }
      }
      piv = (bm > 1e-30f) ? best : -1;
    }
    __syncthreads();
    const int p = piv;
    if (p < 0) { singular = true; break; }  // uniform
    if (p != c)
      for (int j = lane; j < w; j += 64) {
        const float2 tmp = A[c][j];
        A[c][j] = A[p][j];
        A[p][j] = tmp;
      }
    __syncthreads();
    const float2 d = A[c][c];
    const float dn = 1.0f / fmaf(d.x, d.x, d.y * d.y);
    const float2 dinv = make_float2(d.x * dn, -d.y * dn);
    if (lane < r) fac[lane] = A[lane][c];
    __syncthreads();
    for (int j = lane; j < w; j += 64) A[c][j] = cmulf(A[c][j], dinv);
    __syncthreads();
    for (int q = lane; q < r * w; q += 64) {
      const int i = q / w, j = q % w;
      if (i != c) {
        const float2 f = fac[i], pc = A[c][j];
        A[i][j].x -= fmaf(f.x, pc.x, -f.y * pc.y);
        A[i][j].y -= fmaf(f.x, pc.y, f.y * pc.x);
      }
    }
    __syncthreads();
  }
  for (int i = lane; i < rr; i += 64)
    inv[b * rr + i] = singular ? make_float2(0.f, 0.f) : A[i / r][r + i % r];
  if (lane == 0) flag[b] = singular ? 1 : 0;
}

// out[b][n] = G - G^2 * sum_{k,l} U[n][k] W[k][l] V[n][l]; NP positions per thread
constexpr int NP = 1;  // 2 measured slower (registers)
__global__ __launch_bounds__(WT) void bk_woodbury_apply_kernel(const float2 *__restrict__ G,
                                                               const float *__restrict__ U,
                                                               const float *__restrict__ V,
                                                               const float2 *__restrict__ inv, long N, int r,
                                                               long tiles, float2 *__restrict__ out) {
  __shared__ float2 Ws[MAX_R * MAX_R];
  const long b = blockIdx.x / tiles;
  const int rr = r * r;
  for (int i = threadIdx.x; i < rr; i += WT) Ws[i] = inv[b * rr + i];
  __syncthreads();
  const long nb = (long)(blockIdx.x % tiles) * (WT * NP) + threadIdx.x;
  float u[NP][MAX_R], v[NP][MAX_R];
#pragma unroll
  for (int e = 0; e < NP; ++e) {
    const long n = nb + (long)e * WT;
#pragma unroll
    for (int k = 0; k < MAX_R; ++k) {
      const bool ok = k < r && n < N;
      u[e][k] = ok ? U[n * r + k] : 0.0f;
      v[e][k] = ok ? V[n * r + k] : 0.0f;
    }
  }
  float2 q[NP];
#pragma unroll
  for (int e = 0; e < NP; ++e) q[e] = make_float2(0.f, 0.f);
#pragma unroll
  for (int k = 0; k < MAX_R; ++k) {
    if (k < r) {
      float2 tk[NP];
#pragma unroll
      for (int e = 0; e < NP; ++e) tk[e] = make_float2(0.f, 0.f);
#pragma unroll
      for (int l = 0; l < MAX_R; ++l)
        if (l < r) {
          const float2 w = Ws[k * r + l];
#pragma unroll
          for (int e = 0; e < NP; ++e) {
            tk[e].x = fmaf(w.x, v[e][l], tk[e].x);
            tk[e].y = fmaf(w.y, v[e][l], tk[e].y);
          }
        }
#pragma unroll
      for (int e = 0; e < NP; ++e) {
        q[e].x = fmaf(u[e][k], tk[e].x, q[e].x);
        q[e].y = fmaf(u[e][k], tk[e].y, q[e].y);
      }
    }
  }
#pragma unroll
  for (int e = 0; e < NP; ++e) {
    const long n = nb + (long)e * WT;
    if (n < N) {
      const float2 g = G[b * N + n];
      const float2 c = cmulf(cmulf(g, q[e]), g);
      out[b * N + n] = make_float2(g.x - c.x, g.y - c.y);
    }
  }
}

inline int gram_spans(int64_t N) { return (int)((N + GRAM_SPAN - 1) / GRAM_SPAN); }

}  // namespace

extern "C" {

size_t bkcore_woodbury_workspace_bytes(int64_t B, int64_t N, int64_t r) {
  if (B <= 0 || N <= 0 || r <= 0) return 0;
  const size_t rr = (size_t)(r * r);
  // gram partials, inverses, one flag per row (padded)
  return ((size_t)B * gram_spans(N) * rr + (size_t)B * rr) * sizeof(float2) + (((size_t)B * sizeof(int) + 255) & ~(size_t)255);
}

int bkcore_woodbury_correct(const float *G, const float *U, const float *V, int64_t B, int64_t N, int64_t r,
                            float *out, void *workspace, size_t workspace_bytes, void *stream) {
  if (B < 0 || N < 0 || r < 0 || r > MAX_R) return BKCORE_EINVAL;
  if (B == 0 || N == 0) return BKCORE_OK;
  if (!G || !out) return BKCORE_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (r == 0) {
    const hipError_t e = hipMemcpyAsync(out, G, (size_t)B * N * sizeof(float2), hipMemcpyDeviceToDevice, s);
    if (e != hipSuccess) bkcore_internal_set_hip_error((int)e);
    return e == hipSuccess ? BKCORE_OK : BKCORE_ELAUNCH;
  }
  if (!U || !V) return BKCORE_EINVAL;
  const size_t need = bkcore_woodbury_workspace_bytes(B, N, r);
  if (!workspace || workspace_bytes < need) return BKCORE_EWORKSPACE;
  const int spans = gram_spans(N);
  const size_t rr = (size_t)(r * r);
  float2 *partial = (float2 *)workspace;
  float2 *inv = partial + (size_t)B * spans * rr;
  int *flag = (int *)(inv + (size_t)B * rr);
  const int64_t tiles = (N + WT * NP - 1) / (WT * NP);
  if ((int64_t)B * spans > 2147483647LL || B * tiles > 2147483647LL) return BKCORE_EINVAL;
  hipLaunchKernelGGL(bk_woodbury_gram_kernel, dim3((unsigned)(B * spans)), dim3(WT), 0, s, (const float2 *)G, U, V,
                     (long)N, (int)r, spans, partial);
  hipLaunchKernelGGL(bk_woodbury_solve_kernel, dim3((unsigned)B), dim3(64), 0, s, (const float2 *)partial, (int)r,
                     spans, inv, flag);
  hipLaunchKernelGGL(bk_woodbury_apply_kernel, dim3((unsigned)(B * tiles)), dim3(WT), 0, s, (const float2 *)G, U, V,
                     (const float2 *)inv, (long)N, (int)r, (long)tiles, (float2 *)out);
  return bk_launch_status();
}

}  // extern "C"

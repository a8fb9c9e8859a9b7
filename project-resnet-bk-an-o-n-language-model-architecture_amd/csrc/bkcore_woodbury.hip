// Woodbury low-rank correction of the resolvent diagonal, batched over rows: the loop over the batch of the reference's
// Birman-Schwinger core (src/models/birman_schwinger_core.py:567-604) as three launches,
//     M_b   = I + V^T diag(G_b) U                     (r x r complex; bk_woodbury_gram_kernel, O(N r^2) per row)
//     W_b   = M_b^{-1}                                (bk_woodbury_solve_kernel, Gauss-Jordan with partial pivoting)
//     out_b = G_b - G_b^2 * rowsum((U W_b) o V)       (bk_woodbury_apply_kernel, O(N r^2) per row)
// in complex64 like the reference (its tridiagonal diagonal is complex64, :563).  r <= 22 (r = ceil(log2 N) in the
// reference: every N the operator accepts).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

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
constexpr int GRAM_SPAN = 2048;  // positions per gram workgroup (vector form)

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
inline int mf_spans(int64_t N);

// ---- the two O(N r^2) passes on the matrix cores (r <= 16, at least 16 rows) ------------------------------------
// Both are products of a complex matrix with a REAL one that is the same for every row of the batch,
//     gram:   M_b[k][l] - delta_kl = sum_n G_b[n] X[n][(k,l)],   X[n][(k,l)] = V[n][k] U[n][l]
//     apply:  q_b[n] = sum_(k,l) W_b[k][l] U[n][k] V[n][l],       out = G - G^2 q
// i.e. two real GEMMs each (real and imaginary parts of G_b, of W_b) against an operand formed on the fly from the
// staged rows of U and V.  v_mfma_f32_16x16x4_f32 (float32 in, float32 accumulate: the arithmetic of the fma chains
// it replaces) with 16 batch rows as the tile's rows; lane l holds A[row l & 15][k = l >> 4] and B[k = l >> 4][col
// l & 15], and the four results D[row 4 (l >> 4) + i][col l & 15].
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int MF_ROWS = 64;  // batch rows per workgroup: 16 per wave
constexpr int MF_TN = 64;    // positions staged per pass
constexpr int MF_SPAN = 1024; // positions per workgroup of the matrix-core gram
inline int mf_spans(int64_t N) { return (int)((N + MF_SPAN - 1) / MF_SPAN); }

// gram: the tile's columns are l (16 lanes), one accumulator pair per k, the sum runs over positions (4 per MFMA)
template <int RMAX>
__global__ __launch_bounds__(WT) void bk_woodbury_gram_mfma_kernel(const float2 *__restrict__ G,
                                                                   const float *__restrict__ U,
                                                                   const float *__restrict__ V, long B, long N, int r,
                                                                   int spans, float2 *__restrict__ partial) {
  __shared__ float2 Gs[MF_ROWS][MF_TN + 1];
  __shared__ __align__(16) float Us[MF_TN][16];
  __shared__ float4 Vs[MF_TN][4];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int col = lane & 15, kq = lane >> 4;
  const long b0 = (long)(blockIdx.x / spans) * MF_ROWS;
  const int sp = blockIdx.x % spans;
  const long n0 = (long)sp * MF_SPAN, n1 = min(N, n0 + MF_SPAN);
  f32x4 are[RMAX], aim[RMAX];
#pragma unroll
  for (int k = 0; k < RMAX; ++k) {
    are[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    aim[k] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  // the next pass's tiles travel in registers while the matrix cores work on the current one
  constexpr int NUV = MF_TN * 16 / WT, NG = MF_ROWS * MF_TN / WT;
  float pu[NUV], pv[NUV];
  float2 pg[NG];
  auto fetch = [&](long nt) __attribute__((always_inline)) {
    const int cnt = (int)min((long)MF_TN, n1 - nt);
#pragma unroll
    for (int j = 0; j < NUV; ++j) {
      const int i = t + j * WT, n = i >> 4, l = i & 15;
      const bool ok = n < cnt && l < r;  // zeros elsewhere: nothing beyond the span or the rank enters a sum
      pu[j] = ok ? U[(nt + n) * r + l] : 0.0f;
      pv[j] = ok ? V[(nt + n) * r + l] : 0.0f;
    }
#pragma unroll
    for (int j = 0; j < NG; ++j) {
      const int i = t + j * WT, row = i / MF_TN, c = i % MF_TN;
      pg[j] = (b0 + row < B && c < cnt) ? G[(b0 + row) * N + nt + c] : make_float2(0.f, 0.f);
    }
  };
  if (n0 < n1) fetch(n0);
  for (long nt = n0; nt < n1; nt += MF_TN) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NUV; ++j) {
      const int i = t + j * WT;
      Us[i >> 4][i & 15] = pu[j];
      (&Vs[0][0].x)[i] = pv[j];
    }
#pragma unroll
    for (int j = 0; j < NG; ++j) {
      const int i = t + j * WT;
      Gs[i / MF_TN][i % MF_TN] = pg[j];
    }
    __syncthreads();
    if (nt + MF_TN < n1) fetch(nt + MF_TN);
#pragma unroll 2
    for (int n = 0; n < MF_TN; n += 4) {
      const float2 g = Gs[16 * wave + col][n + kq];
      const float u = Us[n + kq][col];
      float vv[16];
#pragma unroll
      for (int q = 0; q < (RMAX + 3) / 4; ++q) {
        const float4 v4 = Vs[n + kq][q];
        vv[4 * q] = v4.x; vv[4 * q + 1] = v4.y; vv[4 * q + 2] = v4.z; vv[4 * q + 3] = v4.w;
      }
      // every k up to RMAX, no branches between the MFMAs: V is zero beyond the rank
#pragma unroll
      for (int k = 0; k < RMAX; ++k) {
        const float x = u * vv[k];
        are[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(g.x, x, are[k], 0, 0, 0);
        aim[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(g.y, x, aim[k], 0, 0, 0);
      }
    }
  }
  const int rr = r * r;
  if (col < r) {
#pragma unroll
    for (int k = 0; k < RMAX; ++k)
      if (k < r) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const long b = b0 + 16 * wave + 4 * kq + i;
          if (b < B) partial[(b * spans + sp) * rr + k * r + col] = make_float2(are[k][i], aim[k][i]);
        }
      }
  }
}

// apply: the tile's columns are 16 positions, the sum runs over the (k, l) pairs (4 per MFMA, rr rounded up), W_b of
// the wave's 16 rows stays in registers for the whole span, X is formed once per pass for all four waves.
constexpr int AP_SPAN = 512;     // positions per workgroup
template <int AP_KS>             // MFMA steps: 4 AP_KS >= r^2 pairs (W is zero beyond r^2: no branches between MFMAs)
__global__ __launch_bounds__(WT) void bk_woodbury_apply_mfma_kernel(const float2 *__restrict__ G,
                                                                    const float *__restrict__ U,
                                                                    const float *__restrict__ V,
                                                                    const float2 *__restrict__ inv, long B, long N,
                                                                    int r, int spans, float2 *__restrict__ out) {
  __shared__ float Xs[4 * AP_KS][MF_TN];
  __shared__ float Us[MF_TN][17];
  __shared__ float Vs[MF_TN][17];
  __shared__ unsigned char kof[4 * AP_KS], lof[4 * AP_KS];  // (k, l) of each pair index
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int col = lane & 15, kq = lane >> 4;
  const long b0 = (long)(blockIdx.x / spans) * MF_ROWS;
  const int sp = blockIdx.x % spans;
  const long n0 = (long)sp * AP_SPAN, n1 = min(N, n0 + AP_SPAN);
  const int rr = r * r;
  constexpr int ks = AP_KS;
  if (t < 4 * AP_KS) {
    kof[t] = (unsigned char)(t < rr ? t / r : 0);
    lof[t] = (unsigned char)(t < rr ? t % r : 0);
  }
  // A operands: W_b[(k,l) = 4 s + kq] of row b0 + 16 wave + col, real and imaginary parts
  float wre[AP_KS], wim[AP_KS];
  {
    const long b = b0 + 16 * wave + col;
#pragma unroll
    for (int s = 0; s < AP_KS; ++s) {
      const int kl = 4 * s + kq;
      float2 w = make_float2(0.f, 0.f);
      if (kl < rr && b < B) w = inv[b * rr + kl];
      wre[s] = w.x;
      wim[s] = w.y;
    }
  }
  for (long nt = n0; nt < n1; nt += MF_TN) {
    const int cnt = (int)min((long)MF_TN, n1 - nt);
    __syncthreads();
    for (int i = t; i < MF_TN * 16; i += WT) {
      const int n = i >> 4, l = i & 15;
      const bool ok = n < cnt && l < r;
      Us[n][l] = ok ? U[(nt + n) * r + l] : 0.0f;
      Vs[n][l] = ok ? V[(nt + n) * r + l] : 0.0f;
    }
    __syncthreads();
    // X[(k,l)][n] = U[n][k] V[n][l]; rows beyond r^2 up to the last step are zero
    for (int i = t; i < 4 * ks * MF_TN; i += WT) {
      const int kl = i / MF_TN, n = i % MF_TN;
      Xs[kl][n] = kl < rr ? Us[n][kof[kl]] * Vs[n][lof[kl]] : 0.0f;
    }
    __syncthreads();
#pragma unroll 1
    for (int nb = 0; nb < MF_TN; nb += 16) {
      f32x4 qre = f32x4{0.f, 0.f, 0.f, 0.f}, qim = f32x4{0.f, 0.f, 0.f, 0.f};
      // the tile's G: asked for before the MFMAs, needed after them
      float2 gt[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const long b = b0 + 16 * wave + 4 * kq + i;
        gt[i] = (b < B && nb + col < cnt) ? G[b * N + nt + nb + col] : make_float2(0.f, 0.f);
      }
#pragma unroll
      for (int s = 0; s < AP_KS; ++s) {
        const float x = Xs[4 * s + kq][nb + col];
        qre = __builtin_amdgcn_mfma_f32_16x16x4f32(wre[s], x, qre, 0, 0, 0);
        qim = __builtin_amdgcn_mfma_f32_16x16x4f32(wim[s], x, qim, 0, 0, 0);
      }
      const long n = nt + nb + col;
      if (nb + col < cnt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const long b = b0 + 16 * wave + 4 * kq + i;
          if (b < B) {
            const float2 g = gt[i];
            const float2 c = cmulf(cmulf(g, make_float2(qre[i], qim[i])), g);
            out[b * N + n] = make_float2(g.x - c.x, g.y - c.y);
          }
        }
      }
    }
  }
}

}  // namespace

extern "C" {

size_t bkcore_woodbury_workspace_bytes(int64_t B, int64_t N, int64_t r) {
  if (B <= 0 || N <= 0 || r <= 0) return 0;
  const size_t rr = (size_t)(r * r);
  // gram partials (either form's span count), inverses, one flag per row (padded)
  return ((size_t)B * mf_spans(N) * rr + (size_t)B * rr) * sizeof(float2) + (((size_t)B * sizeof(int) + 255) & ~(size_t)255);
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
  // The matrix-core forms work on tiles of 16 rows and 16 values of the rank index, 64 rows per workgroup: taken when
  // that gives the GPU enough workgroups (a few rows by a few thousand positions stay with the vector forms).
  const int64_t row_tiles = (B + MF_ROWS - 1) / MF_ROWS;
  // (development: BKCORE_WOODBURY_NO_MFMA in the environment keeps the vector forms)
  static const bool no_mfma = getenv("BKCORE_WOODBURY_NO_MFMA") != nullptr;
  const bool mfma = r <= 16 && B >= 16 && row_tiles * mf_spans(N) >= 64 && !no_mfma;
  const int spans = mfma ? mf_spans(N) : gram_spans(N);
  const size_t rr = (size_t)(r * r);
  float2 *partial = (float2 *)workspace;
  float2 *inv = partial + (size_t)B * spans * rr;
  int *flag = (int *)(inv + (size_t)B * rr);
  const int64_t tiles = (N + WT * NP - 1) / (WT * NP);
  if ((int64_t)B * spans > 2147483647LL || B * tiles > 2147483647LL) return BKCORE_EINVAL;
  if (mfma && row_tiles * spans <= 2147483647LL) {
#define BK_GRAM(RM)                                                                                                   \
  hipLaunchKernelGGL(bk_woodbury_gram_mfma_kernel<RM>, dim3((unsigned)(row_tiles * spans)), dim3(WT), 0, s,           \
                     (const float2 *)G, U, V, (long)B, (long)N, (int)r, spans, partial)
    if (r <= 4) BK_GRAM(4);
    else if (r <= 8) BK_GRAM(8);
    else if (r <= 10) BK_GRAM(10);
    else if (r <= 12) BK_GRAM(12);
    else if (r <= 14) BK_GRAM(14);
    else BK_GRAM(16);
#undef BK_GRAM
  } else
  hipLaunchKernelGGL(bk_woodbury_gram_kernel, dim3((unsigned)(B * spans)), dim3(WT), 0, s, (const float2 *)G, U, V,
                     (long)N, (int)r, spans, partial);
  hipLaunchKernelGGL(bk_woodbury_solve_kernel, dim3((unsigned)B), dim3(64), 0, s, (const float2 *)partial, (int)r,
                     spans, inv, flag);
  const int64_t ap_spans = (N + AP_SPAN - 1) / AP_SPAN;
  if (mfma && row_tiles * ap_spans <= 2147483647LL) {
#define BK_APPLY(KS)                                                                                                  \
  hipLaunchKernelGGL(bk_woodbury_apply_mfma_kernel<KS>, dim3((unsigned)(row_tiles * ap_spans)), dim3(WT), 0, s,       \
                     (const float2 *)G, U, V, (const float2 *)inv, (long)B, (long)N, (int)r, (int)ap_spans,           \
                     (float2 *)out)
    if (r <= 4) BK_APPLY(4);
    else if (r <= 8) BK_APPLY(16);
    else if (r <= 10) BK_APPLY(25);
    else if (r <= 12) BK_APPLY(36);
    else if (r <= 14) BK_APPLY(49);
    else if (r == 15) BK_APPLY(57);
    else BK_APPLY(64);
#undef BK_APPLY
  } else
  hipLaunchKernelGGL(bk_woodbury_apply_kernel, dim3((unsigned)(B * tiles)), dim3(WT), 0, s, (const float2 *)G, U, V,
                     (const float2 *)inv, (long)N, (int)r, (long)tiles, (float2 *)out);
  return bk_launch_status();
}

}  // extern "C"

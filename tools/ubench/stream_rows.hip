// Development micro-benchmark: the memory access pattern of the ratio-form forward without its arithmetic.
// One workgroup of 256 threads per row, chunks of 4096 elements right -> left; per chunk: read a, b, c (plus halo),
// write G and F (8 bytes per element each).  Variants of load width / alignment / store width / cache policy.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
constexpr int T = 256, E = 16, CH = T * E, RH = 16;
struct __attribute__((aligned(4))) F4u { float v[4]; };

template <int LOADV, int STOREV, bool NT>
__global__ __launch_bounds__(T, 2) void kern(const float *__restrict__ a_, long lda, const float *__restrict__ b_, long ldb,
                                             const float *__restrict__ c_, long ldc, long N, float2 *__restrict__ G_,
                                             float2 *__restrict__ F_) {
  __shared__ float sm[3 * (CH + 2 * RH + 8)];
  const long row = blockIdx.x;
  const float *a = a_ + row * lda, *b = b_ + row * ldb, *c = c_ + row * ldc;
  float2 *G = G_ + row * N, *F = F_ + row * N;
  const int t = threadIdx.x;
  const int nch = (int)(N / CH);
  for (int ch = nch - 1; ch >= 0; --ch) {
    const long base = (long)ch * CH;
    float acc = 0.f;
    __syncthreads();
    if (LOADV == 1) {
      const bool full = ch >= 1 && ch <= nch - 2;
#pragma unroll
      for (int j = 0; j < E + 1; ++j) {
        int q = t + j * T;
        if (q > CH + 2 * RH) q = CH + 2 * RH;
        long ia = base - RH + q, ik = base - RH - 1 + q;
        if (!full) { ia = ia < 0 ? 0 : (ia > N - 1 ? N - 1 : ia); ik = ik < 0 ? 0 : (ik > N - 2 ? N - 2 : ik); }
        const float va = NT ? __builtin_nontemporal_load(a + ia) : a[ia];
        const float vb = NT ? __builtin_nontemporal_load(b + ik) : b[ik];
        const float vc = NT ? __builtin_nontemporal_load(c + ik) : c[ik];
        sm[q] = va; sm[CH + 40 + q] = vb; sm[2 * (CH + 40) + q] = vc;
      }
    } else {
      // body by 16-byte loads (4-byte aligned), halo by dword loads of the first 33 threads
#pragma unroll
      for (int j = 0; j < E / 4; ++j) {
        const int e = 4 * (t + j * T);
        long ik = base + e - 1;
        if (ik < 0) ik = 0;
        if (ik > N - 5) ik = N - 5;
        const F4u va = *reinterpret_cast<const F4u *>(a + base + e);
        const F4u vb = *reinterpret_cast<const F4u *>(b + ik);
        const F4u vc = *reinterpret_cast<const F4u *>(c + ik);
#pragma unroll
        for (int i = 0; i < 4; ++i) { sm[RH + e + i] = va.v[i]; sm[CH + 40 + RH + e + i] = vb.v[i]; sm[2 * (CH + 40) + RH + e + i] = vc.v[i]; }
      }
      if (t < 2 * RH + 1) {
        long q = t < RH ? t : CH + t;       // slots 0..15 and CH+16 .. CH+32
        long ia = base - RH + q, ik = ia - 1;
        ia = ia < 0 ? 0 : (ia > N - 1 ? N - 1 : ia); ik = ik < 0 ? 0 : (ik > N - 2 ? N - 2 : ik);
        sm[q] = a[ia]; sm[CH + 40 + q] = b[ik]; sm[2 * (CH + 40) + q] = c[ik];
      }
    }
    __syncthreads();
    // each lane reads its 16 consecutive staged elements (as the chains do), forms something, writes the tile
#pragma unroll
    for (int e = 0; e < E; ++e) acc += sm[RH + t * E + e] + sm[CH + 40 + RH + t * E + e] * sm[2 * (CH + 40) + RH + t * E + e];
    __syncthreads();
    if (STOREV == 1) {
#pragma unroll
      for (int j = 0; j < E; ++j) {
        const int idx = t + j * T;
        const float2 gv = make_float2(acc, sm[idx]);
        __builtin_nontemporal_store(gv.x, &G[base + idx].x); __builtin_nontemporal_store(gv.y, &G[base + idx].y);
        __builtin_nontemporal_store(gv.x, &F[base + idx].x); __builtin_nontemporal_store(gv.y, &F[base + idx].y);
      }
    } else {
#pragma unroll
      for (int j = 0; j < E / 2; ++j) {
        const int idx = 2 * (t + j * T);
        float4 gv = make_float4(acc, sm[idx], acc, sm[idx + 1]);
        float4 *pg = reinterpret_cast<float4 *>(&G[base + idx]), *pf = reinterpret_cast<float4 *>(&F[base + idx]);
        if (STOREV == 2) { *pg = gv; *pf = gv; }
        else {
          typedef float v4f __attribute__((ext_vector_type(4)));
          v4f w = {gv.x, gv.y, gv.z, gv.w};
          __builtin_nontemporal_store(w, reinterpret_cast<v4f *>(pg)); __builtin_nontemporal_store(w, reinterpret_cast<v4f *>(pf));
        }
      }
    }
  }
}

template <int LOADV, int STOREV, bool NT>
void run(const char *name, float *a, float *b, float *c, long ldk, float2 *G, float2 *F, long B, long N) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) kern<LOADV, STOREV, NT><<<B, T>>>(a, N, b, ldk, c, ldk, N, G, F);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  const int reps = 20;
  for (int i = 0; i < reps; ++i) kern<LOADV, STOREV, NT><<<B, T>>>(a, N, b, ldk, c, ldk, N, G, F);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
  printf("%-44s ldk=%ld: %.4f ms  %.2f TB/s (28 B/elem)\n", name, ldk, ms, 28.0 * B * N / (ms * 1e-3) / 1e12);
}

int main() {
  const long B = 1024, N = 32768;
  float *a, *b, *c; float2 *G, *F;
  hipMalloc(&a, B * N * 4); hipMalloc(&b, B * N * 4 + 64); hipMalloc(&c, B * N * 4 + 64);
  hipMalloc(&G, B * N * 8); hipMalloc(&F, B * N * 8);
  hipMemset(a, 0, B * N * 4); hipMemset(b, 0, B * N * 4); hipMemset(c, 0, B * N * 4);
  for (long ldk : {N - 1, N}) {
    run<1, 1, false>("dword loads, 8B nt stores (kernel today)", a, b, c, ldk, G, F, B, N);
    run<1, 1, true>("dword nt loads, 8B nt stores", a, b, c, ldk, G, F, B, N);
    run<4, 1, false>("dwordx4 loads, 8B nt stores", a, b, c, ldk, G, F, B, N);
    run<1, 3, false>("dword loads, 16B nt stores", a, b, c, ldk, G, F, B, N);
    run<4, 3, false>("dwordx4 loads, 16B nt stores", a, b, c, ldk, G, F, B, N);
    run<4, 2, false>("dwordx4 loads, 16B plain stores", a, b, c, ldk, G, F, B, N);
  }
  // aliasing c = b (what the first A/B runs measured)
  {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) kern<1, 1, false><<<B, T>>>(a, N, b, N - 1, b, N - 1, N, G, F);
    hipDeviceSynchronize(); hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) kern<1, 1, false><<<B, T>>>(a, N, b, N - 1, b, N - 1, N, G, F);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 20;
    printf("dword loads, c aliased to b: %.4f ms\n", ms);
  }
  return 0;
}

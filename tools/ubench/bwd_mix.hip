// Development micro-benchmark: the reference-mode backward's memory pattern (per element: 8 B of G + 8 B of cotangent
// read, 4 B written) as a plain streaming kernel, over grid sizes and elements per thread.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
template <int Q, bool NTS>  // Q quads (4 elements) per thread per iteration
__global__ __launch_bounds__(256) void kern(const v4f *__restrict__ G, const v4f *__restrict__ g, v4f *__restrict__ out, long nquads) {
  const long stride = (long)gridDim.x * blockDim.x * Q;
  for (long p0 = ((long)blockIdx.x * blockDim.x) * Q + threadIdx.x; p0 < nquads; p0 += stride) {
    v4f a[Q][2], b[Q][2];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const long p = p0 + q * 256;
      if (p < nquads) {
        a[q][0] = __builtin_nontemporal_load(G + 2 * p); a[q][1] = __builtin_nontemporal_load(G + 2 * p + 1);
        b[q][0] = __builtin_nontemporal_load(g + 2 * p); b[q][1] = __builtin_nontemporal_load(g + 2 * p + 1);
      }
    }
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const long p = p0 + q * 256;
      if (p < nquads) {
        v4f o;
        o.x = a[q][0].x * b[q][0].x + a[q][0].y * b[q][0].y; o.y = a[q][0].z * b[q][0].z + a[q][0].w * b[q][0].w;
        o.z = a[q][1].x * b[q][1].x + a[q][1].y * b[q][1].y; o.w = a[q][1].z * b[q][1].z + a[q][1].w * b[q][1].w;
        if (NTS) __builtin_nontemporal_store(o, out + p); else out[p] = o;
      }
    }
  }
}
template <int Q, bool NTS>
void run(v4f *G, v4f *g, v4f *out, long nquads, int blocks) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) kern<Q, NTS><<<blocks, 256>>>(G, g, out, nquads);
  hipDeviceSynchronize(); hipEventRecord(e0);
  for (int i = 0; i < 30; ++i) kern<Q, NTS><<<blocks, 256>>>(G, g, out, nquads);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 30;
  printf("Q=%d nt_store=%d blocks %6d: %.4f ms  %.2f TB/s\n", Q, (int)NTS, blocks, ms, 20.0 * nquads * 4 / (ms * 1e-3) / 1e12);
}
int main() {
  const long n = 33554432L, nquads = n / 4;
  v4f *G, *g, *out;
  hipMalloc(&G, n * 8); hipMalloc(&g, n * 8); hipMalloc(&out, n * 4);
  hipMemset(G, 0, n * 8); hipMemset(g, 0, n * 8);
  for (int blocks : {1024, 2048, 4096, 8192, 16384, 32768}) {
    run<1, true>(G, g, out, nquads, blocks);
    run<2, true>(G, g, out, nquads, blocks / 2);
    run<4, true>(G, g, out, nquads, blocks / 4);
    run<1, false>(G, g, out, nquads, blocks);
  }
  return 0;
}

// Development micro-benchmark: what plain streaming kernels reach on this device -- read-only, write-only, copy, and the
// forward's mix (12 bytes read + 16 bytes written per element) -- with 16-byte accesses, grid-stride, 4 per thread.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
template <int MODE, bool NT>  // 0 read, 1 write, 2 copy, 3 mix (3 reads : 4 writes)
__global__ __launch_bounds__(256) void kern(const v4f *__restrict__ in, v4f *__restrict__ out, long n4, float *sink) {
  const long stride = (long)gridDim.x * blockDim.x;
  v4f acc = {0, 0, 0, 0};
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    if (MODE == 0) acc += NT ? __builtin_nontemporal_load(in + i) : in[i];
    if (MODE == 1) { v4f w = {(float)i, 1, 2, 3}; if (NT) __builtin_nontemporal_store(w, out + i); else out[i] = w; }
    if (MODE == 2) { v4f w = NT ? __builtin_nontemporal_load(in + i) : in[i]; if (NT) __builtin_nontemporal_store(w, out + i); else out[i] = w; }
    if (MODE == 3) {  // element group g: reads in[3g..3g+2], writes out[4g..4g+3]
      v4f x = in[3 * i], y = in[3 * i + 1], z = in[3 * i + 2];
      v4f w = x + y * z;
      if (NT) { __builtin_nontemporal_store(w, out + 4 * i); __builtin_nontemporal_store(x, out + 4 * i + 1); __builtin_nontemporal_store(y, out + 4 * i + 2); __builtin_nontemporal_store(z, out + 4 * i + 3); }
      else { out[4 * i] = w; out[4 * i + 1] = x; out[4 * i + 2] = y; out[4 * i + 3] = z; }
    }
  }
  if (MODE == 0 && acc.x == 1234.5f) *sink = acc.y;
}
template <int MODE, bool NT>
void run(const char *name, v4f *in, v4f *out, long n4, double bytes, float *sink, int blocks) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) kern<MODE, NT><<<blocks, 256>>>(in, out, n4, sink);
  hipDeviceSynchronize(); hipEventRecord(e0);
  for (int i = 0; i < 20; ++i) kern<MODE, NT><<<blocks, 256>>>(in, out, n4, sink);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 20;
  printf("%-28s blocks %5d: %.4f ms  %.2f TB/s\n", name, blocks, ms, bytes / (ms * 1e-3) / 1e12);
}
int main() {
  const long bytes_in = 402653184L, bytes_out = 536870912L;   // 12 and 16 bytes x 33554432 elements
  v4f *in, *out; float *sink;
  hipMalloc(&in, bytes_out); hipMalloc(&out, bytes_out); hipMalloc(&sink, 4);
  hipMemset(in, 0, bytes_out);
  for (int blocks : {2048, 8192}) {
    run<0, false>("read 384 MB", in, out, bytes_in / 16, bytes_in, sink, blocks);
    run<0, true>("read 384 MB nt", in, out, bytes_in / 16, bytes_in, sink, blocks);
    run<1, false>("write 512 MB", in, out, bytes_out / 16, bytes_out, sink, blocks);
    run<1, true>("write 512 MB nt", in, out, bytes_out / 16, bytes_out, sink, blocks);
    run<2, false>("copy 512 MB", in, out, bytes_out / 16, 2.0 * bytes_out, sink, blocks);
    run<2, true>("copy 512 MB nt", in, out, bytes_out / 16, 2.0 * bytes_out, sink, blocks);
    run<3, false>("mix 384 R + 512 W", in, out, bytes_in / 48, bytes_in + bytes_out, sink, blocks);
    run<3, true>("mix 384 R + 512 W nt", in, out, bytes_in / 48, bytes_in + bytes_out, sink, blocks);
  }
  return 0;
}

// Development micro-benchmark: VALU issue rate per SIMD on gfx950 as a function of waves per SIMD.
// Each wave runs ITER iterations of 16 independent instructions of one kind; grid = 256 CUs x 4 SIMDs x k waves.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
constexpr int ITER = 8192;
template <int KIND>
__global__ __launch_bounds__(64) void kern(float *out, float seed, unsigned long long mask) {
  float a[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) a[i] = seed + i + threadIdx.x * 1e-3f;
  const float m = 1.0f + seed * 1e-7f, c = seed * 1e-9f;
  const int one = (int)seed;
  const long long t0 = __builtin_amdgcn_s_memtime(), w0 = wall_clock64();
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
      if (KIND == 1) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
      if (KIND == 2) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
      if (KIND == 3) asm volatile("v_log_f32 %0, %0" : "+v"(a[i]));
      if (KIND == 5) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
      if (KIND == 6) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(m));
      if (KIND == 7) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "s"(mask));
      if (KIND == 8) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
      if (KIND == 9) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
      if (KIND == 10) asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(a[i]) : "v"(one));
      if (KIND == 11) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
      if (KIND == 12) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[i]), "v"(m) : "vcc");
      if (KIND == 13) asm volatile("v_frexp_exp_i32_f32 %0, %0" : "+v"(a[i]));
      if (KIND == 14) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
      if (KIND == 15) asm volatile("v_rsq_f32 %0, %0" : "+v"(a[i]));
      if (KIND == 16) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[i]) : "v"(m));
      if (KIND == 17) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
      if (KIND == 18) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(m) : "vcc");
      if (KIND == 19) asm volatile("v_cmp_lt_f32_e64 s[20:21], %0, %1\n v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(a[i]) : "v"(m) : "s20", "s21");
      if (KIND == 20) asm volatile("v_fmaak_f32 %0, %0, %1, 0x3f800001" : "+v"(a[i]) : "v"(m));
      if (KIND == 21) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(m));
      if (KIND == 22) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(one));
      if (KIND == 23) asm volatile("v_fma_f32 %0, -%0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
      if (KIND == 24) asm volatile("v_mul_f32_e64 %0, -%0, %1" : "+v"(a[i]) : "v"(m));
      if (KIND == 25) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
      if (KIND == 26) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(a[i]));
      if (KIND == 27) asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(a[i]));
      if (KIND == 28) asm volatile("v_fma_f32 %0, %0, %1, 1.0" : "+v"(a[i]) : "v"(m));
      if (KIND == 29) asm volatile("v_mul_f32 %0, 0.5, %0" : "+v"(a[i]));
    }
    if (KIND == 30) {
#pragma unroll
      for (int i = 0; i < 16; i += 2) {
        float2 v = make_float2(a[i], a[i + 1]);
        float2 mm = make_float2(m, m);
        asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(v) : "v"(mm));
        a[i] = v.x; a[i + 1] = v.y;
      }
    }
    if (KIND == 4) {  // packed: 8 instructions on register pairs = 16 lane-fmas
#pragma unroll
      for (int i = 0; i < 16; i += 2) {
        float2 v = make_float2(a[i], a[i + 1]);
        float2 mm = make_float2(m, m), cc = make_float2(c, c);
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v) : "v"(mm), "v"(cc));
        a[i] = v.x; a[i + 1] = v.y;
      }
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime(), w1 = wall_clock64();
  if (blockIdx.x == 0 && threadIdx.x == 0) { out[512] = (float)(t1 - t0); out[513] = (float)(w1 - w0); }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += a[i];
  if (s == 12345.678f) out[threadIdx.x] = s;
}
template <int KIND>
void run(const char *name, float *d, int ninst) {
  for (int k : {2, 4}) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256 * 4 * k;
    kern<KIND><<<grid, 64>>>(d, 1.0f, 0x5555555555555555ull);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    kern<KIND><<<grid, 64>>>(d, 1.0f, 0x5555555555555555ull);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double inst_per_simd = (double)ITER * ninst * k;
    float h[2]; hipMemcpy(h, d + 512, 8, hipMemcpyDeviceToHost);
    const double ghz = h[0] / (h[1] * 10.0);  // shader cycles per ns (wall clock ticks at 100 MHz)
    printf("%-10s waves/SIMD %d: %.3f ms, %.2f ns per wave-instruction per SIMD; clock %.2f GHz -> %.2f cycles\n", name, k, ms,
           ms * 1e6 / inst_per_simd, ghz, ms * 1e6 / inst_per_simd * ghz);
  }
}
int main() {
  float *d; hipMalloc(&d, 8192);
  run<0>("v_fma", d, 16); run<1>("v_mul", d, 16); run<2>("v_rcp", d, 16); run<3>("v_log", d, 16);
  run<4>("v_pk_fma", d, 8); run<5>("v_med3", d, 16); run<6>("v_cndmask", d, 16);
  run<7>("cndmask64", d, 16); run<8>("v_max", d, 16); run<9>("v_add", d, 16); run<10>("v_ldexp", d, 16);
  run<11>("mov_dpp", d, 16); run<12>("v_cmp", d, 16); run<13>("frexp_exp", d, 16); run<14>("v_exp", d, 16);
  run<15>("v_rsq", d, 16); run<16>("v_mul_b", d, 16); run<17>("v_fmac", d, 16);
  run<18>("cmp+cnd vcc (2)", d, 32); run<19>("cmp+cnd sgpr(2)", d, 32); run<20>("v_fmaak", d, 16); run<21>("v_mov", d, 16);
  run<22>("v_add_u32", d, 16); run<23>("fma neg", d, 16); run<24>("mul_e64 neg", d, 16); run<25>("v_min", d, 16);
  run<26>("v_lshlrev", d, 16); run<27>("v_cvt", d, 16); run<28>("fma inline1.0", d, 16); run<29>("mul inline", d, 16);
  run<30>("v_pk_mul", d, 8);
  return 0;
}

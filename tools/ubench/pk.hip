// Micro-benchmark: issue rate of v_fma_f32 vs v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 on gfx950.
// hipcc --offload-arch=gfx950 -O3 -o pk pk.hip && ./pk
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  float a[16], b = threadIdx.x * 1e-9f, c = 1.0f + threadIdx.x * 1e-9f;
  f2 p[8], pb = {b, b}, pc = {c, c};
#pragma unroll
  for (int i = 0; i < 16; ++i) a[i] = i + threadIdx.x;
#pragma unroll
  for (int i = 0; i < 8; ++i) p[i] = f2{(float)i, (float)(i + threadIdx.x)};
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(b));
    } else if (MODE == 1) {
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pc), "v"(pb));
    } else if (MODE == 2) {
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pc));
    } else if (MODE == 3) {
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pb));
    } else if (MODE == 4) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
    } else if (MODE == 5) {   // fma with abs modifier
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, |%0|, %1, %0" : "+v"(a[i]) : "v"(c));
    } else if (MODE == 6) {   // cvt_pk_bf16
#pragma unroll
      for (int i = 0; i < 16; i += 2) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(a[i]) : "v"(a[i + 1]));
    } else if (MODE == 7) {   // mul by inline constant via VOP2
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += a[i];
#pragma unroll
  for (int i = 0; i < 8; ++i) s += p[i].x + p[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, int per_iter_values, int insts, int waves_per_simd) {
  float* out;
  const int blocks = 256 * waves_per_simd;   // 4 waves per block -> one per SIMD per CU
  hipMalloc(&out, blocks * 256 * 4);
  const int iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  k<MODE><<<blocks, 256>>>(out, 100);
  hipEventRecord(e0);
  k<MODE><<<blocks, 256>>>(out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  // per SIMD: waves_per_simd waves x iters x insts instructions
  const double inst_per_simd = (double)waves_per_simd * iters * insts;
  printf("%-14s waves/SIMD %d: %.3f ms  -> %.2f ns per wave-instruction per SIMD (%.2f cycles @2.4GHz), %.2f values/ns/SIMD\n",
         name, waves_per_simd, ms, ms * 1e6 / inst_per_simd, ms * 1e6 / inst_per_simd * 2.4,
         (double)waves_per_simd * iters * per_iter_values * 64 / (ms * 1e6));
  hipFree(out);
}

int main() {
  for (int w : {1, 2, 4}) {
    run<0>("v_fma_f32", 16, 16, w);
    run<1>("v_pk_fma_f32", 16, 8, w);
    run<2>("v_pk_mul_f32", 16, 8, w);
    run<3>("v_pk_add_f32", 16, 8, w);
    run<4>("v_max_f32", 16, 16, w);
    run<5>("v_fma |abs|", 16, 16, w);
    run<6>("v_cvt_pk_bf16", 16, 8, w);
    run<7>("v_mul_f32", 16, 16, w);
  }
  return 0;
}

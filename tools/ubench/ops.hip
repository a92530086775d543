// Micro-benchmark: issue cost of the VALU / LDS opcodes of the chain kernels' inner loops on gfx950.
// 16 independent dependency chains per wave, 4 waves per SIMD.   hipcc --offload-arch=gfx950 -O3 -o ops ops.hip
#include <hip/hip_runtime.h>
#include <cstdio>

#define KERNEL(NAME, ASM)                                                                    \
  __global__ __launch_bounds__(256) void NAME(float* out, int iters) {                       \
    float a[16];                                                                             \
    float b = threadIdx.x * 1e-9f, c = 1.0f + threadIdx.x * 1e-9f;                           \
    unsigned m = 0xffff0000u | threadIdx.x;                                                  \
    __shared__ float lds[4096];                                                              \
    unsigned addr = (threadIdx.x * 16) & 0x3ff0;                                             \
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = i;                                \
    __syncthreads();                                                                         \
    _Pragma("unroll") for (int i = 0; i < 16; ++i) a[i] = i + threadIdx.x;                   \
    for (int it = 0; it < iters; ++it) {                                                     \
      _Pragma("unroll") for (int i = 0; i < 16; ++i)                                         \
          asm volatile(ASM : "+v"(a[i]) : "v"(c), "v"(b), "v"(m), "v"(addr) : "vcc", "s20", "s21", "memory"); \
    }                                                                                        \
    float s = 0;                                                                             \
    _Pragma("unroll") for (int i = 0; i < 16; ++i) s += a[i];                                \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + lds[threadIdx.x];                       \
  }

KERNEL(k_fma, "v_fma_f32 %0, %0, %1, %2")
KERNEL(k_fmac, "v_fmac_f32 %0, %1, %2")
KERNEL(k_add, "v_add_f32 %0, %0, %1")
KERNEL(k_mul, "v_mul_f32 %0, %0, %1")
KERNEL(k_max, "v_max_f32 %0, %0, %1")
KERNEL(k_max3, "v_max3_f32 %0, %0, %1, %2")
KERNEL(k_mov, "v_mov_b32 %0, %1")
KERNEL(k_and, "v_and_b32 %0, %0, %3")
KERNEL(k_or, "v_or_b32 %0, %0, %3")
KERNEL(k_addu, "v_add_u32 %0, %0, %3")
KERNEL(k_lshl, "v_lshlrev_b32 %0, 1, %0")
KERNEL(k_lshlor, "v_lshl_or_b32 %0, %0, 1, %3")
KERNEL(k_perm, "v_perm_b32 %0, %0, %1, %3")
KERNEL(k_cmp, "v_cmp_lt_f32 vcc, %0, %1")
KERNEL(k_cmp_cnd, "v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc")
KERNEL(k_cnd, "v_cndmask_b32 %0, %0, %2, vcc")
KERNEL(k_cvtpk, "v_cvt_pk_bf16_f32 %0, %0, %1")
KERNEL(k_dot2c, "v_dot2c_f32_bf16 %0, %1, %3")
KERNEL(k_exp, "v_exp_f32 %0, %0")
KERNEL(k_rcp, "v_rcp_f32 %0, %0")
KERNEL(k_dpp_mov, "v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf")
KERNEL(k_dpp_add, "v_add_f32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf")
KERNEL(k_dpp_max, "v_max_f32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf")
KERNEL(k_readlane, "v_readlane_b32 s20, %0, 3")
KERNEL(k_bperm, "ds_bpermute_b32 %0, %4, %0\n s_waitcnt lgkmcnt(0)")
KERNEL(k_bperm_nw, "ds_bpermute_b32 %0, %4, %1")
KERNEL(k_swap, "v_permlane32_swap_b32 %0, %0")
KERNEL(k_dsr128, "ds_read_b32 %0, %4")
KERNEL(k_dsw32, "ds_write_b32 %4, %0")
KERNEL(k_dsw16, "ds_write_b16 %4, %0")
KERNEL(k_dsadd, "ds_add_f32 %4, %1")
KERNEL(k_med3, "v_med3_f32 %0, %0, %1, %2")
KERNEL(k_mulabs, "v_mul_f32 %0, |%0|, %1")
KERNEL(k_sub, "v_sub_f32 %0, %0, %1")
KERNEL(k_bfe, "v_bfe_u32 %0, %0, 3, 5")
KERNEL(k_cvt_u, "v_cvt_f32_u32 %0, %0")
KERNEL(k_accw, "v_accvgpr_write_b32 a0, %0")
KERNEL(k_bfi, "v_bfi_b32 %0, %3, %1, %0")
KERNEL(k_xor, "v_xor_b32 %0, %0, %3")
KERNEL(k_andor, "v_and_or_b32 %0, %0, %3, %1")
KERNEL(k_cnd64, "v_cndmask_b32 %0, %0, %2, s[20:21]")
KERNEL(k_cmp64, "v_cmp_lt_f32 s[20:21], %0, %1")
KERNEL(k_cmpcnd64, "v_cmp_lt_f32 s[20:21], %0, %1\n v_cndmask_b32 %0, %0, %2, s[20:21]")
KERNEL(k_fmak, "v_fma_f32 %0, %0, 0.5, %2")

typedef void (*kern_t)(float*, int);
void run(const char* name, kern_t f, int per_it) {
  float* out;
  const int w = 4, blocks = 256 * w;
  (void)hipMalloc(&out, blocks * 256 * 4);
  const int iters = 10000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  f<<<blocks, 256>>>(out, 100);
  (void)hipEventRecord(e0);
  f<<<blocks, 256>>>(out, iters);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double n = (double)w * iters * 16 * per_it;
  printf("%-12s %.3f ms  %.2f ns per wave-instruction per SIMD\n", name, ms, ms * 1e6 / n);
  (void)hipFree(out);
}
#define RUN(K) run(#K, K, 1)
int main() {
  RUN(k_fma); RUN(k_fmac); RUN(k_add); RUN(k_sub); RUN(k_mul); RUN(k_mulabs); RUN(k_max); RUN(k_max3); RUN(k_med3);
  RUN(k_mov); RUN(k_and); RUN(k_or); RUN(k_addu); RUN(k_lshl); RUN(k_lshlor); RUN(k_perm); RUN(k_bfe); RUN(k_cvt_u);
  RUN(k_cmp); run("k_cmp_cnd", k_cmp_cnd, 2); RUN(k_cnd); RUN(k_cvtpk); RUN(k_dot2c); RUN(k_exp); RUN(k_rcp);
  RUN(k_swap); RUN(k_dsr128); RUN(k_dsw32); RUN(k_dsw16); RUN(k_dsadd); RUN(k_accw); RUN(k_bfi); RUN(k_xor); RUN(k_andor); RUN(k_cnd64); RUN(k_cmp64); run("k_cmpcnd64", k_cmpcnd64, 2); RUN(k_fmak);
  return 0;
}

// semantics probe of ds_read_b64_tr_b16 (gfx950) for a [view][32 channels] bf16 image
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(const unsigned short* src, unsigned long long* out, int rs_bytes) {
  __shared__ __attribute__((aligned(16))) unsigned short buf[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) buf[i] = src[i];
  __syncthreads();
  const int l = threadIdx.x, ip = l & 15, g = l >> 4;
  for (int t = 0; t < 2; ++t) {
    const int k0 = 8 * (g >> 1) + 4 * t;
    const unsigned addr = (unsigned)((k0 + (ip >> 2)) * rs_bytes + (16 * (g & 1) + 4 * (ip & 3)) * 2);
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s16x4*)((__attribute__((address_space(3))) char*)buf + addr));
    out[l * 2 + t] = __builtin_bit_cast(unsigned long long, v);
  }
}
int main() {
  const int RS = 80;   // bytes per view row (64 data + pad)
  unsigned short h[8192];
  for (int i = 0; i < 8192; ++i) h[i] = 0xffff;
  for (int v = 0; v < 32; ++v)
    for (int c = 0; c < 32; ++c) h[(v * RS) / 2 + c] = (unsigned short)(v * 100 + c);
  unsigned short* d; unsigned long long* o;
  hipMalloc(&d, sizeof(h)); hipMalloc(&o, 128 * 8);
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, RS);
  unsigned long long r[128];
  hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l)
    for (int t = 0; t < 2; ++t) {
      const int g = l >> 4, n = 16 * (g & 1) + (l & 15), k0 = 8 * (g >> 1) + 4 * t;
      for (int j = 0; j < 4; ++j) {
        const int got = (int)((r[l * 2 + t] >> (16 * j)) & 0xffff), want = (k0 + j) * 100 + n;
        if (got != want) { if (bad < 12) printf("lane %d t %d j %d: got %d want %d\n", l, t, j, got, want); ++bad; }
      }
    }
  printf("mismatches: %d\n", bad);
  return 0;
}

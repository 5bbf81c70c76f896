// probe: what does each lane receive from ds_read_b64_tr_b16 with the addressing used for the wgrad operands?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(int* out) {
  __shared__ __attribute__((aligned(16))) short tile[16 * 40];     // [pixel 0..15][channel 0..31 (+8 pad)]
  int l = threadIdx.x;
  for (int i = l; i < 16 * 40; i += 64) tile[i] = (short)((i / 40) * 100 + (i % 40));   // value = pix*100 + ch
  __syncthreads();
  int g = l >> 4, i = l & 15, q = i >> 2, p = i & 3;
  for (int half = 0; half < 2; ++half) {
    short* addr = tile + (8 * (g >> 1) + 4 * half + q) * 40 + 16 * (g & 1) + 4 * p;
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)addr);
    for (int j = 0; j < 4; ++j) out[(l * 2 + half) * 4 + j] = v[j];
  }
}
int main() {
  int* d; hipMalloc(&d, 64 * 8 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  int h[512]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) {
    int r = l & 31, hh = l >> 5;
    for (int j = 0; j < 8; ++j) { int want = (8 * hh + j) * 100 + r; if (h[l * 8 + j] != want) ++bad; }
  }
  printf("lane0: "); for (int j = 0; j < 8; ++j) printf("%d ", h[j]); printf("\nlane17: "); for (int j = 0; j < 8; ++j) printf("%d ", h[17 * 8 + j]);
  printf("\nlane40: "); for (int j = 0; j < 8; ++j) printf("%d ", h[40 * 8 + j]);
  printf("\nexpected lane(r,h) elem j = pixel(8h+j)*100 + channel r ; mismatches: %d\n", bad);
  return 0;
}

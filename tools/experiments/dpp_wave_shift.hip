// Probe: semantics of v_mov_b32_dpp wave_shr:1 / wave_shl:1 on gfx950 (direction, what the lane without a source keeps).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float* x, float* y, float* z) {
  float c = x[threadIdx.x];
  float e = -1.0f;
  int ci = __builtin_bit_cast(int, c), ei = __builtin_bit_cast(int, e);
  int l = __builtin_amdgcn_update_dpp(ei, ci, 0x138, 0xf, 0xf, false);   // wave_shr:1
  int r = __builtin_amdgcn_update_dpp(ei, ci, 0x130, 0xf, 0xf, false);   // wave_shl:1
  y[threadIdx.x] = __builtin_bit_cast(float, l);
  z[threadIdx.x] = __builtin_bit_cast(float, r);
}
int main() {
  float h[64], hy[64], hz[64], *x, *y, *z;
  for (int i = 0; i < 64; ++i) h[i] = (float)i;
  hipMalloc(&x, 256); hipMalloc(&y, 256); hipMalloc(&z, 256);
  hipMemcpy(x, h, 256, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, x, y, z);
  hipMemcpy(hy, y, 256, hipMemcpyDeviceToHost); hipMemcpy(hz, z, 256, hipMemcpyDeviceToHost);
  printf("shr:"); for (int i : {0, 1, 15, 16, 31, 32, 62, 63}) printf(" [%d]=%g", i, hy[i]); printf("\n");
  printf("shl:"); for (int i : {0, 1, 15, 16, 31, 32, 62, 63}) printf(" [%d]=%g", i, hz[i]); printf("\n");
  return 0;
}

// Developer tool: which XCD does workgroup g of a 2-D / 3-D grid run on?  Reads HW_REG_XCC_ID per workgroup.  hipcc --offload-arch=gfx950 -O2 xcc_probe.hip -o xcc_probe && ./xcc_probe
// Measured on MI355X (round 2): XCD = linear workgroup id % 8 (x fastest) for every grid shape tried - what xcd_tile_net / chain_decode (csrc) rely on.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(int* out) {
  if (threadIdx.x == 0) {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    out[blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)] = (int)(v & 0xf);
  }
}
int main() {
  int* d; const int W = 157, L = 32; hipMalloc(&d, W * L * 4);
  k<<<dim3(W, L), 256>>>(d); hipDeviceSynchronize();
  static int h[157 * 32]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int ok = 0; for (int g = 0; g < W * L; ++g) ok += (h[g] == g % 8);
  printf("2D grid %dx%d: %d of %d blocks on XCD (linear id %% 8)\n", W, L, ok, W * L);
  printf("first 24: "); for (int g = 0; g < 24; ++g) printf("%d ", h[g]); printf("\n");
  printf("block (0,y) y=0..15: "); for (int y = 0; y < 16; ++y) printf("%d ", h[y * W]); printf("\n");
  k<<<dim3(64, 4, 8), 1024>>>(d); hipDeviceSynchronize(); hipMemcpy(h, d, 64*4*8*4, hipMemcpyDeviceToHost);
  ok = 0; for (int g = 0; g < 64*4*8; ++g) ok += (h[g] == g % 8);
  printf("3D grid 64x4x8 (1024 thr): %d of %d\n", ok, 64*4*8);
  return 0;
}

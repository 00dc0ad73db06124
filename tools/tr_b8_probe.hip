// What ds_read_b64_tr_b8 (gfx950) delivers: every lane supplies the address of 8 contiguous bytes of a [row][col] byte image; the
// probe fills the image once with the row index and once with the column index and prints, per lane, the 8 bytes received.
// Hypothesis (the b16 form's rule, cdna_hip_programming.md T10, scaled to bytes): per group of 16 lanes a block of 8 rows x 16
// columns; lane 2q+p supplies row q, columns 8p..8p+7; lane i receives column i of the 8 rows (byte q = row q).
// build + run (GPU box): hipcc --offload-arch=gfx950 -O2 tools/tr_b8_probe.hip -o /tmp/tr8 && /tmp/tr8
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
__global__ void k(unsigned char* out, int mode) {
  __shared__ __attribute__((aligned(16))) unsigned char img[64][64];
  const int t = threadIdx.x;
  for (int i = t; i < 64 * 64; i += 64) img[i / 64][i % 64] = mode ? (i % 64) : (i / 64);
  __syncthreads();
  const int g = t >> 4, l = t & 15, q = l >> 1, p = l & 1;
  // group g reads the block with first row 8 * g, first column 16 * g (distinct per group, to see the grouping)
  const unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)&img[8 * g + q][16 * g + 8 * p];
  u32x2 v;
  asm volatile("ds_read_b64_tr_b8 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  for (int j = 0; j < 8; ++j) out[t * 8 + j] = (v[j >> 2] >> (8 * (j & 3))) & 0xff;
}
int main() {
  unsigned char* d; hipMalloc(&d, 512);
  unsigned char h[2][512];
  for (int mode = 0; mode < 2; ++mode) { hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, mode); hipMemcpy(h[mode], d, 512, hipMemcpyDeviceToHost); }
  for (int t = 0; t < 64; ++t) {
    printf("lane %2d (group %d, supplies row %2d cols %2d..): ", t, t >> 4, 8 * (t >> 4) + ((t & 15) >> 1), 16 * (t >> 4) + 8 * (t & 1));
    for (int j = 0; j < 8; ++j) printf("(%2d,%2d) ", h[0][t * 8 + j], h[1][t * 8 + j]);
    printf("\n");
  }
  return 0;
}

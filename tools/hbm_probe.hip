// What the HBM3E of this MI355X sustains for the access mixes of the step's HBM-bound kernels (MI355X_MICROARCH.md prices
// them at the 8 TB/s data-sheet figure; this is the figure a perfect streaming kernel reaches on the box, the way
// tools/clock_probe.hip measures what 100 % MFMA issue is worth).  Plain grid-stride kernels, 16 bytes per lane per access,
// buffers of 1.5 GiB each (>> the 256 MiB Infinity Cache), 20 timed launches after 3 warm-ups, HIP events:
//   read      sum of one buffer                         (1 R)
//   write     fill of one buffer                        (1 W)
//   copy      y = a                                     (1 R : 1 W)     avgpool / relu_bwd / layout kernels
//   add       y = a + b                                 (2 R : 1 W)     residual epilogues
//   expand    y[m][0:4c] = f(a[m][0:c]) + r[m][0:4c]    (1.25 R : 1 W by rows: c in, 4c residual in, 4c out)
//                                                       the N = 4K 1x1 layers (K = 512 -> N = 2048 with residual)
//   L2 -> LDS what buffer_load ... lds delivers from an L2-resident buffer with nothing else running (the conv kernels' operand path)
// Rates are ALGORITHMIC bytes / time.  build + run:
//   hipcc --offload-arch=gfx950 -O3 tools/hbm_probe.hip -o /tmp/hbm_probe && /tmp/hbm_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

__global__ __launch_bounds__(256) void k_read(const u32x4* a, unsigned* sink, long n) {
  unsigned s = 0;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) { u32x4 v = a[i]; s += v[0] ^ v[1] ^ v[2] ^ v[3]; }
  if (s == 0x12345u) sink[0] = s;
}
__global__ __launch_bounds__(256) void k_write(u32x4* y, long n, unsigned c) {
  u32x4 v = {c, c + 1, c + 2, c + 3};
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) y[i] = v;
}
__global__ __launch_bounds__(256) void k_copy(const u32x4* a, u32x4* y, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) y[i] = a[i];
}
__global__ __launch_bounds__(256) void k_add(const u32x4* a, const u32x4* b, u32x4* y, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) { u32x4 u = a[i], v = b[i]; y[i] = u + v; }
}
// the same copy the way ATen's vectorized elementwise kernel runs it (6.2 TB/s on 2 GiB tensors, measured with torch.mul(a, c, out=b)):
// ONE-SHOT grid, every thread U independent 16-byte loads (block-strided) first, then its U stores
template <int U>
__global__ __launch_bounds__(256) void k_copy_oneshot(const u32x4* a, u32x4* y, long n) {
  const long base = (long)blockIdx.x * (256 * U) + threadIdx.x;
  u32x4 v[U];
#pragma unroll
  for (int u = 0; u < U; ++u) if (base + u * 256 < n) v[u] = a[base + u * 256];
#pragma unroll
  for (int u = 0; u < U; ++u) if (base + u * 256 < n) y[base + u * 256] = v[u];
}
// ... and a grid-stride loop with the same U loads in flight per thread
template <int U>
__global__ __launch_bounds__(256) void k_copy_unrolled(const u32x4* a, u32x4* y, long n) {
  for (long base = (long)blockIdx.x * (256 * U) + threadIdx.x; base < n; base += (long)gridDim.x * (256 * U)) {
    u32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) if (base + u * 256 < n) v[u] = a[base + u * 256];
#pragma unroll
    for (int u = 0; u < U; ++u) if (base + u * 256 < n) y[base + u * 256] = v[u];
  }
}

// rows of c chunks in, 4c chunks residual in, 4c chunks out (c = 64 chunks of 16 B = 512 bf16 channels)
__global__ __launch_bounds__(256) void k_expand(const u32x4* a, const u32x4* r, u32x4* y, long rows, int c) {
  for (long row = blockIdx.x; row < rows; row += gridDim.x) {
    const u32x4* ar = a + row * c;
    for (int j = threadIdx.x; j < 4 * c; j += 256) { u32x4 u = ar[j & (c - 1)], v = r[row * 4 * c + j]; y[row * 4 * c + j] = u + v; }
  }
}

// L2 -> LDS: every workgroup (512 threads, one per CU) streams 64 KiB tiles of ONE 2 MiB buffer (L2-resident in every XCD after the
// first pass) into LDS with buffer_load ... lds, 16 bytes per lane, the conv kernels' operand path without their compute
__global__ __launch_bounds__(512) void k_l2_to_lds(const char* a, int iters, unsigned* sink) {
  __shared__ __attribute__((aligned(16))) u32x4 lds[2][4096];          // 2 x 64 KiB
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)a, 0, 0x7fffffff, 0x00020000);
  const int t = threadIdx.x, wv = __builtin_amdgcn_readfirstlane(t >> 6);
  unsigned tile = blockIdx.x & 31;
  for (int it = 0; it < iters; ++it) {
    char* dst = (char*)lds[it & 1] + wv * 1024;
#pragma unroll
    for (int i = 0; i < 8; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(dst + i * 8192), 16, (int)(t * 16 + i * 8192), (int)(tile * 65536), 0, 0);
    tile = (tile + 1) & 31;
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                 // the previous tile has landed; this one stays in flight
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (lds[0][t][0] == 0x12345u) sink[0] = 1;
}

template <typename F> static float timed(F launch) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) launch();
  hipEventRecord(e0, 0);
  for (int i = 0; i < 20; ++i) launch();
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / 20.f;
}

int main() {
  const long bytes = 1536L << 20, n = bytes / 16;
  u32x4 *a, *b, *y;
  unsigned* sink;
  if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess || hipMalloc(&y, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMalloc(&sink, 64);
  hipMemset(a, 1, bytes); hipMemset(b, 2, bytes); hipMemset(y, 0, bytes);
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  printf("# %s, %d CUs; buffers %ld MiB; algorithmic TB/s\n", prop.name, prop.multiProcessorCount, bytes >> 20);
  for (int per_cu : {4, 8, 16, 32}) {
    const int grid = prop.multiProcessorCount * per_cu;
    float t;
    t = timed([&] { hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, a, sink, n); });
    printf("blocks/CU %2d  read   %.2f", per_cu, bytes / t / 1e9);
    t = timed([&] { hipLaunchKernelGGL(k_write, dim3(grid), dim3(256), 0, 0, y, n, 7u); });
    printf("  write  %.2f", bytes / t / 1e9);
    t = timed([&] { hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, a, y, n); });
    printf("  copy   %.2f", 2.0 * bytes / t / 1e9);
    t = timed([&] { hipLaunchKernelGGL(k_add, dim3(grid), dim3(256), 0, 0, a, b, y, n); });
    printf("  add    %.2f", 3.0 * bytes / t / 1e9);
    const int c = 64;
    const long rows = n / (4 * c);
    t = timed([&] { hipLaunchKernelGGL(k_expand, dim3(grid), dim3(256), 0, 0, a, b, y, rows, c); });
    printf("  expand %.2f\n", (double)rows * c * 9 * 16 / t / 1e9);
  }
  {
    const int iters = 2000;
    float tl = timed([&] { hipLaunchKernelGGL(k_l2_to_lds, dim3(prop.multiProcessorCount), dim3(512), 0, 0, (const char*)a, iters, sink); });
    printf("L2 -> LDS (buffer_load ... lds, 64 KiB tiles of a 2 MiB buffer, one 512-thread workgroup per CU)  %.2f TB/s\n", (double)prop.multiProcessorCount * iters * 65536.0 / tl / 1e9);
  }
  {
    float t1 = timed([&] { hipLaunchKernelGGL(k_copy_oneshot<1>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, a, y, n); });
    float t4 = timed([&] { hipLaunchKernelGGL(k_copy_oneshot<4>, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, 0, a, y, n); });
    float t8 = timed([&] { hipLaunchKernelGGL(k_copy_oneshot<8>, dim3((unsigned)((n + 2047) / 2048)), dim3(256), 0, 0, a, y, n); });
    printf("copy, one-shot grid: 1 / 4 / 8 loads per thread  %.2f  %.2f  %.2f\n", 2.0 * bytes / t1 / 1e9, 2.0 * bytes / t4 / 1e9, 2.0 * bytes / t8 / 1e9);
    for (int per_cu : {8, 32}) {
      const int grid = prop.multiProcessorCount * per_cu;
      float u4 = timed([&] { hipLaunchKernelGGL(k_copy_unrolled<4>, dim3(grid), dim3(256), 0, 0, a, y, n); });
      float u8 = timed([&] { hipLaunchKernelGGL(k_copy_unrolled<8>, dim3(grid), dim3(256), 0, 0, a, y, n); });
      printf("copy, grid-stride %2d blocks/CU: 4 / 8 loads in flight per thread  %.2f  %.2f\n", per_cu, 2.0 * bytes / u4 / 1e9, 2.0 * bytes / u8 / 1e9);
    }
  }
  float t = timed([&] { hipMemcpyAsync(y, a, bytes, hipMemcpyDeviceToDevice, 0); });
  printf("hipMemcpyAsync D2D copy %.2f\n", 2.0 * bytes / t / 1e9);
  return 0;
}

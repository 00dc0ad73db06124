// In-kernel clock of an MFMA-dense loop on this MI355X (MI355X_MICROARCH.md "DVFS give-back", item 6): the chip lowers its
// shader clock under matrix load, so "100 % MFMA issue" is worth  clock x 256 CUs x 4 SIMDs x 32*32*16*2 FLOP / 32 cycles  --
// not the 2.5 PFLOP/s of the 2.4 GHz data sheet.  Two loops on random bf16 data, one wave per SIMD and two waves per SIMD
// (the occupancy of k_conv_fwd256): (a) bare v_mfma_f32_32x32x16_bf16 from registers, (b) the same with every operand
// re-read from LDS by ds_read_b128 (the main loop's instruction mix without its DMA).  Clock = d(s_memtime) / d(s_memrealtime)
// x 100 MHz, stamped around the loop after >= 2 s of back-to-back launches; median over workgroups.
//   build + run:  hipcc --offload-arch=gfx950 -O3 tools/clock_probe.hip -o /tmp/clock_probe && /tmp/clock_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int LDS>
__global__ __launch_bounds__(512) void k_probe(const u32x4* src, float* sink, unsigned long long* stamps, int iters) {
  __shared__ u32x4 lds[4096];                                  // 64 KiB
  const int t = threadIdx.x, lane = t & 63;
  for (int i = t; i < 4096; i += blockDim.x) lds[i] = src[(blockIdx.x * 4096 + i) & 0xfffff];
  __syncthreads();
  f32x16 acc[8];
  for (int a = 0; a < 8; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  u32x4 fa[4], fb[4];
  for (int k = 0; k < 4; ++k) { fa[k] = lds[(t * 4 + k) & 4095]; fb[k] = lds[(t * 4 + k + 2048) & 4095]; }
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
    if (LDS == 1) {
#pragma unroll
      for (int k = 0; k < 4; ++k) { fa[k] = lds[(lane + 64 * k + 256 * (it & 7)) & 4095]; fb[k] = lds[(lane + 64 * k + 2048 + 256 * (it & 7)) & 4095]; }
    }
    if (LDS == 2) {                               // the conv kernels' mix: 0.75 KB of fragment reads per MFMA (24 x ds_read_b128 per 32 MFMAs)
#pragma unroll
      for (int rep = 0; rep < 3; ++rep)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          fa[k] = lds[(lane + 64 * k + 256 * ((it + rep) & 7)) & 4095]; fb[k] = lds[(lane + 64 * k + 2048 + 256 * ((it + rep) & 7)) & 4095];
          asm volatile("" : "+v"(fa[k]), "+v"(fb[k]));
        }
    }
    if (LDS == 3) {                               // the weight-gradient kernel's mix: the same bytes as 48 x ds_read_b64_tr_b16
      const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
#pragma unroll
      for (int rep = 0; rep < 3; ++rep)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const unsigned a0 = base + (((lane + 64 * k + 256 * ((it + rep) & 7)) & 4095) << 4), b0 = base + (((lane + 64 * k + 2048 + 256 * ((it + rep) & 7)) & 4095) << 4);
          typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
          u32x2 p0, p1, q0, q1;
          asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(p0) : "v"(a0));
          asm volatile("ds_read_b64_tr_b16 %0, %1 offset:8" : "=v"(p1) : "v"(a0));
          asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(q0) : "v"(b0));
          asm volatile("ds_read_b64_tr_b16 %0, %1 offset:8" : "=v"(q1) : "v"(b0));
          if (rep == 2) {
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(p0), "+v"(p1), "+v"(q0), "+v"(q1));
            fa[k] = u32x4{p0[0], p0[1], p1[0], p1[1]}; fb[k] = u32x4{q0[0], q0[1], q1[0], q1[1]};
          }
        }
    }
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
      for (int k = 0; k < 4; ++k)
        acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[k]), __builtin_bit_cast(bf16x8, fb[(k + a) & 3]), acc[a], 0, 0, 0);
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int a = 0; a < 8; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
  if (s == 12345.678f) sink[0] = s;                            // keeps the loop alive
  if (t == 0) { stamps[2 * blockIdx.x] = c1 - c0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int LDS> static void run(const char* what, int threads, const u32x4* src, float* sink, unsigned long long* stamps, int nblk) {
  const int iters = 4000;
  auto t0 = std::chrono::steady_clock::now();
  int launches = 0;
  while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 2.0) {      // warm: the clock settles under load
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k_probe<LDS>, dim3(nblk), dim3(threads), 0, 0, src, sink, stamps, iters);
    hipDeviceSynchronize();
    launches += 20;
  }
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k_probe<LDS>, dim3(nblk), dim3(threads), 0, 0, src, sink, stamps, iters);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(2 * nblk);
  hipMemcpy(h.data(), stamps, sizeof(unsigned long long) * 2 * nblk, hipMemcpyDeviceToHost);
  std::vector<double> ghz;
  for (int b = 0; b < nblk; ++b) ghz.push_back((double)h[2 * b] / (double)h[2 * b + 1] * 0.1);
  std::sort(ghz.begin(), ghz.end());
  const double flops = (double)nblk * (threads / 64) * iters * 32.0 * 2.0 * 32 * 32 * 16;
  const double clk = ghz[ghz.size() / 2];
  printf("%-44s waves/SIMD %d  in-kernel clock median %.3f GHz (min %.3f max %.3f)  %.1f TFLOP/s  = %.3f of the clock's MFMA peak (%.0f TF)  [%d warm launches]\n",
         what, threads / 256, clk, ghz.front(), ghz.back(), flops / (ms * 1e-3) / 1e12, flops / (ms * 1e-3) / (clk * 1e9 * 1024 * 1024.0), clk * 1024 * 1024.0 / 1e3, launches);
}

int main() {
  const int nblk = 256;
  u32x4* src; float* sink; unsigned long long* stamps;
  hipMalloc(&src, sizeof(u32x4) << 20); hipMalloc(&sink, 64); hipMalloc(&stamps, sizeof(unsigned long long) * 2 * nblk);
  std::vector<unsigned short> h((size_t)8 << 20);
  srand(1);
  for (auto& v : h) { float f = (float)(rand() & 0xffff) / 32768.f - 1.f; unsigned u; memcpy(&u, &f, 4); v = (unsigned short)(u >> 16); }   // random bf16 in (-1, 1)
  hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  printf("device %s, %d CUs, clockRate %d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
  printf("MFMA peak at clock f: f x 256 CUs x 4 SIMDs x (2*32*32*16 FLOP / 32 cycles) = f[GHz] x 1048.6 TFLOP/s; 2.4 GHz -> 2517\n");
  run<0>("bare MFMA, operands in registers", 256, src, sink, stamps, nblk);
  run<0>("bare MFMA, operands in registers", 512, src, sink, stamps, nblk);
  run<1>("MFMA + ds_read_b128 operand re-reads", 256, src, sink, stamps, nblk);
  run<1>("MFMA + ds_read_b128 operand re-reads", 512, src, sink, stamps, nblk);
  run<2>("MFMA + 0.75 KB/MFMA of ds_read_b128", 512, src, sink, stamps, nblk);
  run<3>("MFMA + 0.75 KB/MFMA of ds_read_b64_tr_b16", 512, src, sink, stamps, nblk);
  return 0;
}

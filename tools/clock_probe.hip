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

// Round 3: would a 128 x 128-per-wave tiling (ONE wave per SIMD, 256 accumulator registers, 0.5 KB of fragment reads per MFMA
// instead of 0.75) lift the LDS-port ceiling of the 256 x 256 kernels?  Same loop with 16 accumulator tiles: per k-step 4 + 4
// fragment reads feed 16 MFMAs.  STAGE adds the operand staging's LDS writes (0.25 KB per MFMA: 8 x 1 KiB ds_write_b128 per 32
// MFMAs and wave in either tiling) -- the LDS-DMA of the real kernels writes the same bytes through the same port.
template <int NACC, int STAGE>
__global__ __launch_bounds__(NACC == 16 ? 256 : 512) void k_probe_tile(const u32x4* src, float* sink, unsigned long long* stamps, int iters) {
  __shared__ u32x4 lds[8192];                                  // 128 KiB: reads from the lower half, staging writes into the upper
  const int t = threadIdx.x, lane = t & 63;
  for (int i = t; i < 4096; i += blockDim.x) lds[i] = src[(blockIdx.x * 4096 + i) & 0xfffff];
  __syncthreads();
  constexpr int NA = NACC == 16 ? 4 : 4, NB = NACC == 16 ? 4 : 2;       // A row tiles x B column tiles per wave: 4 x 4 or 4 x 2
  f32x16 acc[NACC];
  for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  u32x4 fa[NA], fb[NB], wv = src[t & 4095];
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int ks = 0; ks < 32 / (NA * NB); ++ks) {              // k-steps per 32 MFMAs: 2 (4 x 4) or 4 (4 x 2)
#pragma unroll
      for (int a = 0; a < NA; ++a) { fa[a] = lds[(lane + 64 * a + 256 * ((it + ks) & 7)) & 4095]; asm volatile("" : "+v"(fa[a])); }
#pragma unroll
      for (int b = 0; b < NB; ++b) { fb[b] = lds[(lane + 64 * b + 2048 + 256 * ((it + ks) & 7)) & 4095]; asm volatile("" : "+v"(fb[b])); }
      if (STAGE) {
#pragma unroll
        for (int w = 0; w < 8 * (NA * NB) / 32; ++w) lds[4096 + ((t + 512 * w + 64 * (it & 3)) & 4095)] = wv;
      }
#pragma unroll
      for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b)
          acc[a * NB + b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[a]), __builtin_bit_cast(bf16x8, fb[b]), acc[a * NB + b], 0, 0, 0);
    }
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
  if (s == 12345.678f) sink[0] = s + lds[4096 + t][0];
  if (t == 0) { stamps[2 * blockIdx.x] = c1 - c0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

// Round 3 (the review's proposal): B fragments straight from L2 into registers -- weights re-laid in MFMA-fragment order so that a
// wave's fragment is one contiguous 1 KiB global_load_dwordx4 -- while A stays in LDS: 0.5 KB of LDS reads + 0.125 KB of staging
// writes per MFMA instead of 0.75 + 0.25.  8 waves of 128 x 64 as in k_conv_fwd256; the B fragments of the NEXT 32 MFMAs (8 loads,
// 32 registers) are requested before the current 32 MFMAs are issued (DEPTH = 1) or two groups ahead (DEPTH = 2, 64 registers).
template <int DEPTH>
__global__ __launch_bounds__(512) void k_probe_bl2(const u32x4* src, float* sink, unsigned long long* stamps, int iters) {
  __shared__ u32x4 lds[8192];
  const int t = threadIdx.x, lane = t & 63, wv_ = t >> 6;
  for (int i = t; i < 4096; i += blockDim.x) lds[i] = src[(blockIdx.x * 4096 + i) & 0xfffff];
  __syncthreads();
  f32x16 acc[8];
  for (int a = 0; a < 8; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  u32x4 fa[4], fb[DEPTH + 1][8], wv = src[t & 4095];
  // a 4 MiB window of "weights" per workgroup pair (L2-resident after the first pass): 64 lanes x 16 B per fragment
  const u32x4* wsrc = src + ((blockIdx.x & 3) * 65536);
  auto loadB = [&](int it, u32x4* dst) {
#pragma unroll
    for (int k = 0; k < 8; ++k) dst[k] = __builtin_nontemporal_load(wsrc + ((((it * 8 + k) * 4 + (wv_ & 3)) * 64 + lane) & 65535));
  };
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) loadB(d, fb[d]);
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it0 = 0; it0 < iters; it0 += DEPTH + 1) {
#pragma unroll
    for (int u = 0; u <= DEPTH; ++u) {                        // (the register buffers rotate by unrolling, not by indexing)
      const int it = it0 + u;
      loadB(it + DEPTH, fb[(u + DEPTH) % (DEPTH + 1)]);
      if (DEPTH == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
        for (int a = 0; a < 4; ++a) { fa[a] = lds[(lane + 64 * a + 256 * ((it + ks) & 7)) & 4095]; asm volatile("" : "+v"(fa[a])); }
        lds[4096 + ((t + 512 * ks + 64 * (it & 3)) & 4095)] = wv;        // A staging only: 4 x 1 KiB per 32 MFMAs and wave
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b)
            acc[a * 2 + b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[a]), __builtin_bit_cast(bf16x8, fb[u][ks * 2 + b]), acc[a * 2 + b], 0, 0, 0);
      }
    }
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int a = 0; a < 8; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
  if (s == 12345.678f) sink[0] = s + lds[4096 + t][0];
  if (t == 0) { stamps[2 * blockIdx.x] = c1 - c0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int DEPTH> static void run_bl2(const char* what, const u32x4* src, float* sink, unsigned long long* stamps, int nblk) {
  const int iters = 3996, threads = 512;                       // (a multiple of 2 and of 3)
  auto t0 = std::chrono::steady_clock::now();
  while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 2.0) {
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((k_probe_bl2<DEPTH>), dim3(nblk), dim3(threads), 0, 0, src, sink, stamps, iters);
    hipDeviceSynchronize();
  }
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k_probe_bl2<DEPTH>), dim3(nblk), dim3(threads), 0, 0, src, sink, stamps, iters);
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
  printf("%-64s waves/SIMD %d  clock %.3f GHz  %.1f TFLOP/s  = %.3f of the clock's MFMA peak (%.0f TF)\n",
         what, threads / 256, clk, flops / (ms * 1e-3) / 1e12, flops / (ms * 1e-3) / (clk * 1e9 * 1024 * 1024.0), clk * 1024 * 1024.0 / 1e3);
}

// The same 128 x 128-per-wave loop, SOFTWARE-PIPELINED by hand: the 8 fragment reads of k-step s+1 are issued between the 16 MFMAs of
// k-step s (one ds_read_b128 per two MFMAs, second register set), no lgkmcnt(0) in front of the MFMAs -- MI355X_MICROARCH.md (LDS
// section): up to two ds_read_b128 per MFMA gap cost one wave per SIMD at most 3 cycles per gap.
template <int STAGE>
__global__ __launch_bounds__(256) void k_probe_pipe(const u32x4* src, float* sink, unsigned long long* stamps, int iters) {
  __shared__ u32x4 lds[8192];
  const int t = threadIdx.x, lane = t & 63;
  for (int i = t; i < 4096; i += blockDim.x) lds[i] = src[(blockIdx.x * 4096 + i) & 0xfffff];
  __syncthreads();
  f32x16 acc[16];
  for (int a = 0; a < 16; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  u32x4 fa[2][4], fb[2][4], wv = src[t & 4095];
#pragma unroll
  for (int a = 0; a < 4; ++a) { fa[0][a] = lds[(lane + 64 * a) & 4095]; fb[0][a] = lds[(lane + 64 * a + 2048) & 4095]; }
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {                           // two k-steps = 32 MFMAs per iteration; register sets alternate
      const int cur = ks, nxt = ks ^ 1;
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        fa[nxt][a] = lds[(lane + 64 * a + 256 * ((it + ks + 1) & 7)) & 4095];
        fb[nxt][a] = lds[(lane + 64 * a + 2048 + 256 * ((it + ks + 1) & 7)) & 4095];
      }
      if (STAGE == 1) {
#pragma unroll
        for (int w = 0; w < 4; ++w) lds[4096 + ((t + 512 * w + 64 * (it & 3)) & 4095)] = wv;
      }
      if (STAGE == 2) {                                        // the same 4 KB per wave and k-step by LDS-DMA (global_load_lds_dwordx4)
#pragma unroll
        for (int w = 0; w < 4; ++w)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + ((blockIdx.x * 4096 + t + 256 * w + 1024 * ((it + ks) & 3)) & 0xfffff)),
                                           (__attribute__((address_space(3))) void*)&lds[4096 + (t & ~63) + 256 * w + 1024 * ((it + ks) & 3)], 16, 0, 0);
      }
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
          acc[a * 4 + b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[cur][a]), __builtin_bit_cast(bf16x8, fb[cur][b]), acc[a * 4 + b], 0, 0, 0);
      // issue order within the k-step: 2 MFMAs, 1 LDS read, ... (8 reads under 16 MFMAs), the staging writes behind them
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);     // 2 MFMA
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);     // 1 DS read
      }
      if (STAGE == 1) __builtin_amdgcn_sched_group_barrier(0x200, 4, 0);   // DS writes
    }
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int a = 0; a < 16; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
  if (s == 12345.678f) sink[0] = s + lds[4096 + t][0];
  if (t == 0) { stamps[2 * blockIdx.x] = c1 - c0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

// 8 waves of 128 x 64 (two per SIMD), each software-pipelined the same way: 6 fragment reads of k-step s+1 under the 8 MFMAs of k-step s,
// 2 LDS-DMA instructions per wave and k-step for the staging (STAGE == 2)
template <int STAGE>
__global__ __launch_bounds__(512) void k_probe_pipe8(const u32x4* src, float* sink, unsigned long long* stamps, int iters) {
  __shared__ u32x4 lds[8192];
  const int t = threadIdx.x, lane = t & 63;
  for (int i = t; i < 4096; i += blockDim.x) lds[i] = src[(blockIdx.x * 4096 + i) & 0xfffff];
  __syncthreads();
  f32x16 acc[8];
  for (int a = 0; a < 8; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  u32x4 fa[2][4], fb[2][2];
#pragma unroll
  for (int a = 0; a < 4; ++a) fa[0][a] = lds[(lane + 64 * a) & 4095];
#pragma unroll
  for (int b = 0; b < 2; ++b) fb[0][b] = lds[(lane + 64 * b + 2048) & 4095];
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {                           // four k-steps = 32 MFMAs per iteration
      const int cur = ks & 1, nxt = cur ^ 1;
#pragma unroll
      for (int a = 0; a < 4; ++a) fa[nxt][a] = lds[(lane + 64 * a + 256 * ((it + ks + 1) & 7)) & 4095];
#pragma unroll
      for (int b = 0; b < 2; ++b) fb[nxt][b] = lds[(lane + 64 * b + 2048 + 256 * ((it + ks + 1) & 7)) & 4095];
      if (STAGE == 2) {
#pragma unroll
        for (int w = 0; w < 2; ++w)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + ((blockIdx.x * 4096 + t + 512 * w + 1024 * ((it + ks) & 3)) & 0xfffff)),
                                           (__attribute__((address_space(3))) void*)&lds[4096 + (t & ~63) + 512 * w + 1024 * ((it + ks) & 3)], 16, 0, 0);
      }
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
          acc[a * 2 + b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[cur][a]), __builtin_bit_cast(bf16x8, fb[cur][b]), acc[a * 2 + b], 0, 0, 0);
#pragma unroll
      for (int g = 0; g < 6; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // 1 MFMA
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);     // 1 DS read
      }
    }
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int a = 0; a < 8; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
  if (s == 12345.678f) sink[0] = s + lds[4096 + t][0];
  if (t == 0) { stamps[2 * blockIdx.x] = c1 - c0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int STAGE> static void run_pipe8(const char* what, const u32x4* src, float* sink, unsigned long long* stamps, int nblk) {
  const int iters = 4000, threads = 512;
  auto t0 = std::chrono::steady_clock::now();
  while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 2.0) {
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((k_probe_pipe8<STAGE>), dim3(nblk), dim3(threads), 0, 0, src, sink, stamps, iters);
    hipDeviceSynchronize();
  }
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k_probe_pipe8<STAGE>), dim3(nblk), dim3(threads), 0, 0, src, sink, stamps, iters);
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
  printf("%-64s waves/SIMD %d  clock %.3f GHz  %.1f TFLOP/s  = %.3f of the clock's MFMA peak (%.0f TF)\n",
         what, threads / 256, clk, flops / (ms * 1e-3) / 1e12, flops / (ms * 1e-3) / (clk * 1e9 * 1024 * 1024.0), clk * 1024 * 1024.0 / 1e3);
}

template <int STAGE> static void run_pipe(const char* what, const u32x4* src, float* sink, unsigned long long* stamps, int nblk) {
  const int iters = 4000, threads = 256;
  auto t0 = std::chrono::steady_clock::now();
  while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 2.0) {
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((k_probe_pipe<STAGE>), dim3(nblk), dim3(threads), 0, 0, src, sink, stamps, iters);
    hipDeviceSynchronize();
  }
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k_probe_pipe<STAGE>), dim3(nblk), dim3(threads), 0, 0, src, sink, stamps, iters);
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
  printf("%-64s waves/SIMD %d  clock %.3f GHz  %.1f TFLOP/s  = %.3f of the clock's MFMA peak (%.0f TF)\n",
         what, threads / 256, clk, flops / (ms * 1e-3) / 1e12, flops / (ms * 1e-3) / (clk * 1e9 * 1024 * 1024.0), clk * 1024 * 1024.0 / 1e3);
}

template <int NACC, int STAGE> static void run_tile(const char* what, const u32x4* src, float* sink, unsigned long long* stamps, int nblk) {
  const int iters = 4000, threads = NACC == 16 ? 256 : 512;
  auto t0 = std::chrono::steady_clock::now();
  while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 2.0) {
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((k_probe_tile<NACC, STAGE>), dim3(nblk), dim3(threads), 0, 0, src, sink, stamps, iters);
    hipDeviceSynchronize();
  }
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k_probe_tile<NACC, STAGE>), dim3(nblk), dim3(threads), 0, 0, src, sink, stamps, iters);
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
  printf("%-64s waves/SIMD %d  clock %.3f GHz  %.1f TFLOP/s  = %.3f of the clock's MFMA peak (%.0f TF)\n",
         what, threads / 256, clk, flops / (ms * 1e-3) / 1e12, flops / (ms * 1e-3) / (clk * 1e9 * 1024 * 1024.0), clk * 1024 * 1024.0 / 1e3);
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
  printf("tilings of a 256 x 256 workgroup tile (round 3): fragment reads per MFMA, with and without the staging writes (0.25 KB per MFMA)\n");
  run_tile<8, 0>("8 waves of 128 x 64: 0.75 KB reads", src, sink, stamps, nblk);
  run_tile<8, 1>("8 waves of 128 x 64: 0.75 KB reads + 0.25 KB staging writes", src, sink, stamps, nblk);
  run_tile<16, 0>("4 waves of 128 x 128: 0.5 KB reads", src, sink, stamps, nblk);
  run_tile<16, 1>("4 waves of 128 x 128: 0.5 KB reads + 0.25 KB staging writes", src, sink, stamps, nblk);
  run_pipe<0>("4 waves of 128 x 128, reads of step s+1 under the MFMAs of step s", src, sink, stamps, nblk);
  run_pipe<1>("4 waves of 128 x 128, pipelined reads + 0.25 KB staging writes", src, sink, stamps, nblk);
  run_pipe<2>("4 waves of 128 x 128, pipelined reads + 0.25 KB staged by LDS-DMA", src, sink, stamps, nblk);
  run_pipe8<0>("8 waves of 128 x 64, reads of step s+1 under the MFMAs of step s", src, sink, stamps, nblk);
  run_pipe8<2>("8 waves of 128 x 64, pipelined reads + 0.25 KB staged by LDS-DMA", src, sink, stamps, nblk);
  run_bl2<1>("8 waves, A from LDS (0.5 + 0.125 KB), B from L2, 1 group ahead", src, sink, stamps, nblk);
  run_bl2<2>("8 waves, A from LDS (0.5 + 0.125 KB), B from L2, 2 groups ahead", src, sink, stamps, nblk);
  return 0;
}

// Toolchain/runtime probe: y = a*x + b on the caller's stream. Used by tests to
// check that the C-ABI library, torch's HIP runtime and stream handles interoperate.
#include "common.h"
__global__ void k_axpb(const float* x, float* y, float a, float b, long n) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = a * x[i] + b;
}
extern "C" int cddmsl_probe_axpb(const float* x, float* y, float a, float b, long n, void* stream) {
  if (n < 0) return CDDMSL_ERR_ARG;
  if (n == 0) return CDDMSL_OK;
  hipLaunchKernelGGL(k_axpb, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, y, a, b, n);
  return launch_status();
}
extern "C" int cddmsl_abi_version() { return 1; }

// probe.hip — measurement aid of bench.py: the rate at which this chip SUSTAINS bf16 MFMA work.
// The datasheet peak (2.5 PFLOP/s dense bf16, MI355X_MICROARCH.md) assumes 2.4 GHz; under a matrix-pipe-bound
// load with non-trivial operands the power management settles well below that (tools/micro/chain.hip:
// 96 % pipe occupancy at ~1.5 GHz).  nphm_probe_mfma_rate() runs an MFMA-only loop shaped like the GEMM
// body of the field kernels (8 wavefronts per workgroup = 2 per SIMD, one workgroup per CU, A fragments
// re-read from LDS every K-step, B operands in registers, v_mfma_f32_32x32x16_bf16, pseudo-random
// operands) and reports executed FLOP/s and the shader clock it ran at.  Not on any product path.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "capi_common.h"

namespace nphm {
namespace probe {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int WAVES = 8, KSTEPS = 13, SLOT_BYTES = KSTEPS * 2 * 1024, LDS_BYTES = 100 * 1024;

__device__ __forceinline__ unsigned hash32(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
// two bf16 in [-2, 2) with random mantissas
__device__ __forceinline__ unsigned rand_bf16x2(unsigned seed) {
  const unsigned r = hash32(seed);
  return (r & 0x807f807fu) | 0x3f803f80u;
}

__global__ __launch_bounds__(64 * WAVES) void mfma_only_kernel(float* out, long long* ticks, int chunks) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  for (unsigned i = threadIdx.x; i < 2 * SLOT_BYTES / 4; i += blockDim.x)
    reinterpret_cast<unsigned*>(lds)[i] = rand_bf16x2(i * 2654435761u + blockIdx.x);
  __syncthreads();
  const int lane = threadIdx.x & 63;
  u32x4 bh[2], bl[2];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      bh[s][q] = rand_bf16x2(threadIdx.x * 97u + s * 8u + q);
      bl[s][q] = rand_bf16x2(threadIdx.x * 131u + s * 8u + q + 77u) & 0xbfffbfffu;   // smaller magnitudes, like the lo parts
    }
  f32x16 acc = {};
  const long long t0 = clock64(), w0 = wall_clock64();
#pragma unroll 1
  for (int c = 0; c < chunks; ++c) {
    const bf16x8* A = reinterpret_cast<const bf16x8*>(lds + (c & 1) * SLOT_BYTES) + lane;
    bf16x8 wh = A[0], wl = A[64];
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      const bf16x8 nh = A[(2 * (ks + 1 < KSTEPS ? ks + 1 : ks)) * 64], nl = A[(2 * (ks + 1 < KSTEPS ? ks + 1 : ks) + 1) * 64];
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, __builtin_bit_cast(bf16x8, bh[ks & 1]), acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, __builtin_bit_cast(bf16x8, bl[ks & 1]), acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, __builtin_bit_cast(bf16x8, bh[ks & 1]), acc, 0, 0, 0);
      wh = nh; wl = nl;
    }
    // keep the accumulator bounded without leaving the matrix pipe idle for long
    if ((c & 63) == 63) {
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] *= 1e-6f;
    }
  }
  const long long t1 = clock64(), w1 = wall_clock64();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { ticks[0] = t1 - t0; ticks[1] = w1 - w0; }
}

}  // namespace probe
}  // namespace nphm

extern "C" int nphm_probe_mfma_rate(double* tflops, double* clock_ghz, void* stream) {
  using namespace nphm::probe;
  if (!tflops || !clock_ghz) return nphm_fail_msg("nphm_probe_mfma_rate: null pointer");
  hipStream_t st = static_cast<hipStream_t>(stream);
  int dev = 0, cus = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e == hipSuccess) e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  if (e != hipSuccess || cus <= 0) return nphm_fail("nphm_probe_mfma_rate: device query", e);
  float* out = nullptr;
  long long* ticks = nullptr;
  e = hipMalloc(&out, size_t(cus) * 64 * WAVES * sizeof(float) + 64);
  if (e != hipSuccess) return nphm_fail("nphm_probe_mfma_rate: hipMalloc", e);
  ticks = reinterpret_cast<long long*>(out + size_t(cus) * 64 * WAVES);
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(mfma_only_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (e == hipSuccess) e = hipEventCreate(&e0);
  if (e == hipSuccess) e = hipEventCreate(&e1);
  const int chunks = 6000;                       // ~5 ms: long enough for the clock to settle
  float ms = 0.f;
  long long h[2] = {0, 0};
  if (e == hipSuccess) {
    (void)hipEventRecord(e0, st);
    hipLaunchKernelGGL(mfma_only_kernel, dim3(cus), dim3(64 * WAVES), LDS_BYTES, st, out, ticks, chunks);
    (void)hipEventRecord(e1, st);
    e = hipEventSynchronize(e1);
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    if (e == hipSuccess) e = hipMemcpy(h, ticks, sizeof(h), hipMemcpyDeviceToHost);
  }
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  (void)hipFree(out);
  if (e != hipSuccess) return nphm_fail("nphm_probe_mfma_rate", e);
  const double mfmas = double(chunks) * KSTEPS * 3 * WAVES * cus;
  *tflops = mfmas * 2.0 * 32 * 32 * 16 / (double(ms) * 1e-3) / 1e12;
  *clock_ghz = h[1] > 0 ? double(h[0]) / (double(h[1]) * 10.0) : 0.0;     // wall_clock64 ticks at 100 MHz
  return 0;
}

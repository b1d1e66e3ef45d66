// Microbenchmark (development tool): can the VALU epilogue of one chunk hide in the shadow of the
// dependent MFMA chain of the next one, inside ONE wavefront's instruction stream?
//   hipcc --offload-arch=gfx950 -O3 tools/micro/overlap.hip -o gpurun_tmp/overlap && gpurun_tmp/overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

template <int MODE, int NV>   // MODE 0: MFMA only, 1: VALU only, 2: interleaved (1 MFMA : NV softplus values)
__global__ __launch_bounds__(512) void k(float* out, int iters) {
  __shared__ float pad[24 * 1024];   // 96 KiB: one workgroup per CU
  pad[threadIdx.x] = threadIdx.x;
  __syncthreads();
  f32x16 acc = {};
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(i * 0.5f); }
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.01f + i;
  const long long w0 = wall_clock64();
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (MODE != 1) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
      if (MODE != 0) {
#pragma unroll
        for (int r = 0; r < NV; ++r) {
          float d = v[(u + r) & 7];
          const float t = __builtin_amdgcn_exp2f(-fabsf(d));
          v[(u + r) & 7] = fmaxf(d, 0.f) + __builtin_amdgcn_logf(1.f + t) - 0.25f;
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const long long t1 = clock64();
  const long long w1 = wall_clock64();
  float s = pad[(threadIdx.x * 7) & 511];
  for (int i = 0; i < 16; ++i) s += acc[i];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) { out[2 * (threadIdx.x >> 6)] = float(t1 - t0) / (iters * 8); out[2 * (threadIdx.x >> 6) + 1] = float(w1 - w0) * 10.f / (iters * 8); }
}

template <int MODE, int NV>
void run(const char* name, int threads, float* d) {
  float h[16] = {0};
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<MODE, NV>), dim3(256), dim3(threads), 0, 0, d, 2000);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    hipEventElapsedTime(&ms, e0, e1);
  }
  hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
  printf("%-20s waves/SIMD %d : kernel %.1f ns per item; per-wave ticks:", name, threads / 256, ms * 1e6f / 16000.f);
  for (int w = 0; w < threads / 64; ++w) printf(" %.1f", h[2 * w]);
  printf("\n");
}

int main() {
  float* d;
  hipMalloc(&d, 256 * 512 * 4);
  for (int threads : {256, 512}) {
    if (threads == 256) {
      run<0, 0>("mfma only", 256, d); run<1, 1>("valu only nv=1", 256, d); run<2, 1>("interleaved nv=1", 256, d);
      run<1, 2>("valu only nv=2", 256, d); run<2, 2>("interleaved nv=2", 256, d);
    } else {
      run<0, 0>("mfma only", 512, d); run<1, 1>("valu only nv=1", 512, d); run<2, 1>("interleaved nv=1", 512, d);
      run<1, 2>("valu only nv=2", 512, d); run<2, 2>("interleaved nv=2", 512, d);
    }
  }
  return 0;
}

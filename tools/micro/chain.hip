// Microbenchmark (development tool): how well does the identity kernel's chunk pipeline keep the
// matrix pipe busy, as a function of its STRUCTURE?  One "chunk" = 13 K-steps x PASSES MFMAs
// (v_mfma_f32_32x32x16_bf16) per point tile, A fragments read from LDS one K-step ahead, the VALU
// epilogue of the previous chunk (16 values per tile: softplus + bf16 hi/lo split) threaded between the
// MFMAs.  Variants: wavefronts per workgroup (8 = two per SIMD, 4 = one per SIMD), point tiles per
// wavefront (independent accumulation chains sharing the A fragments), epilogue flavour, barrier per
// pair of chunks.  Output: ns per MFMA and SIMD (13.3 ns = matrix pipe saturated at 2.4 GHz).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-honor-nans tools/micro/chain.hip -o gpurun_tmp/chain
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <type_traits>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (N > 0) {
    static_for<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
}
template <int B, int E, class F>
__device__ __forceinline__ void static_range(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    static_range<B + 1, E>(f);
  }
}
__host__ __device__ constexpr int unit_begin(int s, int ns, int nu) { return (s * nu + ns - 1) / ns; }

struct ActB { bf16x8 hi[2], lo[2]; };

// EPI 1: max + exp + add + log + add (5 ops), EPI 2: exp + add + log + med3 (4 ops), EPI 3: polynomial (light members)
template <int EPI>
__device__ __forceinline__ float softplus(float d) {
  if constexpr (EPI == 1) {
    const float t = __builtin_amdgcn_exp2f(-fabsf(d));
    return fmaxf(d, 0.f) + __builtin_amdgcn_logf(1.f + t);
  } else if constexpr (EPI == 2) {
    const float r = __builtin_amdgcn_logf(1.f + __builtin_amdgcn_exp2f(d));
    return __builtin_amdgcn_fmed3f(d, r, 127.f);
  } else {
    float q = fmaxf(fmaf(fabsf(d), -0.06564446f, 1.00028698f), 0.f);
    q *= q;
    q *= q;
    return fmaf(q, q, fmaxf(d, 0.f));
  }
}

template <int R, bool LIGHT>
__device__ __forceinline__ void pack_pair(const f32x16& a, ActB& o) {
  constexpr int s = R >> 3, i = R & 7;
  const __bf16 h0 = (__bf16)a[R], h1 = (__bf16)a[R + 1];
  o.hi[s][i] = h0;
  o.hi[s][i + 1] = h1;
  if constexpr (!LIGHT) {
    o.lo[s][i] = (__bf16)(a[R] - (float)h0);
    o.lo[s][i + 1] = (__bf16)(a[R + 1] - (float)h1);
  }
  if constexpr (i == 6) {
    asm volatile("" : "+v"(o.hi[s]));
    if constexpr (!LIGHT) asm volatile("" : "+v"(o.lo[s]));
  }
}

// ORDER 0: per pass all tiles (t0 hh, t1 hh, t0 hl, t1 hl, ...), ORDER 1: per tile all passes (t0 hh hl lh, t1 hh hl lh)
template <int WAVES, int NT, int PASSES, int EPI, int PF, int BAR, int ORDER>
__global__ __launch_bounds__(64 * WAVES) void k(float* out, int chunks) {
  extern __shared__ __attribute__((aligned(16))) char lds[];   // 2 x 26 KiB of "weights"
  for (int i = threadIdx.x; i < 2 * 26 * 1024 / 4; i += blockDim.x) reinterpret_cast<float*>(lds)[i] = 0.001f * (i & 1023);
  __syncthreads();
  const int lane = threadIdx.x & 63;
  constexpr bool LIGHT = PASSES == 1;
  constexpr int NKS = 13;
  f32x16 acc[NT], prev[NT];
  ActB b[NT], ob[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc[t][i] = 0.f; prev[t][i] = 0.01f * (lane + i + t); }
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int i = 0; i < 8; ++i) { b[t].hi[s][i] = (__bf16)(0.01f * (lane + i + 3 * t)); b[t].lo[s][i] = (__bf16)(1e-4f * (lane - i - t)); }
    asm volatile("" : "+v"(b[t].hi[0]), "+v"(b[t].hi[1]), "+v"(b[t].lo[0]), "+v"(b[t].lo[1]));
    ob[t] = b[t];
  }
  const long long t0 = clock64();
  const long long w0 = wall_clock64();
#pragma unroll 1
  for (int c = 0; c < chunks; ++c) {
    if (BAR && (c & 1) == 0) __builtin_amdgcn_s_barrier();
    const bf16x8* A = reinterpret_cast<const bf16x8*>(lds + (c & 1) * 26 * 1024) + lane;
    bf16x8 wh[NKS], wl[NKS];
#pragma unroll
    for (int ks = 0; ks < PF && ks < NKS; ++ks) {
      wh[ks] = A[(2 * ks) * 64];
      if (!LIGHT) wl[ks] = A[(2 * ks + 1) * 64];
    }
    constexpr int NS = NKS * PASSES * NT, NU = EPI ? 16 * NT : 0;
    auto epi = [&](auto uu) __attribute__((always_inline)) {
      constexpr int u = decltype(uu)::value, t = u / 16, r = u % 16;
      if constexpr (EPI != 0) {
        prev[t][r] = softplus<EPI>(prev[t][r]);
        if constexpr (r & 1) pack_pair<r - 1, LIGHT>(prev[t], ob[t]);
      }
    };
    static_for<NKS>([&](auto kk) __attribute__((always_inline)) {
      constexpr int ks = decltype(kk)::value;
      if constexpr (ks + PF < NKS) {
        wh[ks + PF] = A[(2 * (ks + PF)) * 64];
        if (!LIGHT) wl[ks + PF] = A[(2 * (ks + PF) + 1) * 64];
      }
      __builtin_amdgcn_sched_barrier(0);
      static_for<PASSES * NT>([&](auto ss) __attribute__((always_inline)) {
        constexpr int s = decltype(ss)::value;
        constexpr int m = ORDER == 0 ? s / NT : s % PASSES, t = ORDER == 0 ? s % NT : s / PASSES;
        if constexpr (m == 0) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[ks], b[t].hi[ks & 1], acc[t], 0, 0, 0);
        else if constexpr (m == 1) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[ks], b[t].lo[ks & 1], acc[t], 0, 0, 0);
        else acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl[ks], b[t].hi[ks & 1], acc[t], 0, 0, 0);
        constexpr int slot = ks * PASSES * NT + s;
        static_range<unit_begin(slot, NS, NU), unit_begin(slot + 1, NS, NU)>(epi);
        __builtin_amdgcn_sched_barrier(0);
      });
    });
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      asm volatile("" : "+v"(acc[t]));
      prev[t] = acc[t];
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
      asm volatile("" :: "v"(ob[t].hi[0]), "v"(ob[t].hi[1]), "v"(ob[t].lo[0]), "v"(ob[t].lo[1]));
    }
  }
  const long long t1 = clock64();
  const long long w1 = wall_clock64();
  float s = 0.f;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) s += prev[t][i];
  out[blockIdx.x * blockDim.x + threadIdx.x + 64] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = float(t1 - t0); out[1] = float(w1 - w0); }
}

template <int WAVES, int NT, int PASSES, int EPI, int PF, int BAR, int ORDER>
void run(float* d) {
  const int chunks = 2000;
  float h[2] = {0, 0};
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1); float ms = 0, best = 1e9f;
  auto kern = k<WAVES, NT, PASSES, EPI, PF, BAR, ORDER>;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(kern, dim3(256), dim3(64 * WAVES), 100 * 1024, 0, d, chunks);
    (void)hipEventRecord(e1, 0);
    (void)hipDeviceSynchronize();
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  (void)hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
  const double mfma_per_simd = double(chunks) * 13 * PASSES * NT * (WAVES / 4);
  const double ns = best * 1e6 / mfma_per_simd;
  const double ghz = h[0] / (h[1] * 10.0);     // wall_clock64 ticks at 100 MHz
  printf("waves %d tiles %d passes %d epi %d pf %d bar %d order %d : %.2f ms, %6.2f ns/MFMA/SIMD, %.2f GHz, pipe busy %.1f %%  (ticks/chunk/wave %.0f)\n",
         WAVES, NT, PASSES, EPI, PF, BAR, ORDER, best, ns, ghz, 100.0 * 32.0 / (ns * ghz), h[0] / chunks);
}

int main() {
  float* d; (void)hipMalloc(&d, (256 * 512 + 64) * 4);
  // the shipped structure: 8 wavefronts, one tile each, heavy (3 passes)
  run<8, 1, 3, 0, 1, 0, 0>(d); run<8, 1, 3, 1, 1, 0, 0>(d); run<8, 1, 3, 2, 1, 0, 0>(d); run<8, 1, 3, 1, 1, 1, 0>(d); run<8, 1, 3, 1, 2, 0, 0>(d);
  // light (single pass): 13 MFMAs per chunk
  run<8, 1, 1, 0, 3, 0, 0>(d); run<8, 1, 1, 3, 3, 0, 0>(d); run<8, 1, 1, 2, 3, 0, 0>(d);
  // one wavefront per SIMD, two tiles (independent chains, shared A fragments)
  run<4, 2, 3, 0, 1, 0, 0>(d); run<4, 2, 3, 1, 1, 0, 0>(d); run<4, 2, 3, 2, 1, 0, 0>(d); run<4, 2, 3, 1, 1, 0, 1>(d); run<4, 2, 3, 1, 1, 1, 0>(d); run<4, 2, 3, 1, 2, 0, 0>(d);
  run<4, 2, 1, 0, 3, 0, 0>(d); run<4, 2, 1, 3, 3, 0, 0>(d); run<4, 2, 1, 2, 3, 0, 0>(d);
  // one wavefront per SIMD, one tile (reference), and two per SIMD with two tiles each
  run<4, 1, 3, 0, 1, 0, 0>(d); run<4, 1, 3, 1, 1, 0, 0>(d);
  run<8, 2, 3, 0, 1, 0, 0>(d); run<8, 2, 3, 1, 1, 0, 0>(d); run<8, 2, 3, 2, 1, 0, 0>(d); run<8, 2, 1, 3, 3, 0, 0>(d);
  // four tiles on one wavefront per SIMD
  run<4, 4, 3, 1, 1, 0, 0>(d); run<4, 4, 1, 3, 3, 0, 0>(d);
  return 0;
}

// Microbenchmark / numerics probe (development tool) for the split-f16 path of the identity kernel:
//   1. does v_mfma_f32_32x32x16_f16 keep f16 SUBNORMAL inputs (the lo halves of typical weights are subnormal)?
//   2. accuracy of the 3-term products xh wh + xl wh + xh wl with f16 halves against bf16 halves (K = 208, fp64 reference)
//   3. issue cost of the chunk body: NM dependent MFMAs with the epilogue of 16 values threaded through, for
//      the bf16 epilogue (softplus + cvt_pk_bf16 / shift / and / sub / cvt_pk_bf16) and the f16 one
//      (softplus + cvt_pk_f16 + v_fma_mixlo/hi_f16), and the single-pass ("light") bodies: fp32 polynomial +
//      cvt_pk_bf16 against cvt_pk_f16 + packed-f16 polynomial.
//   hipcc --offload-arch=gfx950 -O3 -fno-honor-nans tools/micro/f16split.hip -o gpurun_tmp/f16split && gpurun_tmp/f16split
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include <vector>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// ---- 1 + 2: numerics ---------------------------------------------------------------------------------
// one wavefront: D[32 x 32] = A[32 x K] B[K x 32], K = 16 * KS, operands given as fp32 [row][k] / [k][col]
template <int FMT>   // 0: bf16 three-term, 1: f16 three-term, 2: f16 two-term (weights rounded), 3: bf16 two-term
__global__ void gemm_probe(const float* A, const float* B, float* D, int KS) {
  const int lane = threadIdx.x & 63, h = lane >> 5, j = lane & 31;
  f32x16 acc = {};
  for (int ks = 0; ks < KS; ++ks) {
    float a[8], b[8];
    for (int i = 0; i < 8; ++i) {
      a[i] = A[j * (16 * KS) + ks * 16 + 8 * h + i];
      b[i] = B[(ks * 16 + 8 * h + i) * 32 + j];
    }
    if constexpr (FMT == 0 || FMT == 3) {
      bf16x8 ah, al, bh, bl;
      for (int i = 0; i < 8; ++i) {
        ah[i] = (__bf16)a[i]; al[i] = (__bf16)(a[i] - (float)ah[i]);
        bh[i] = (__bf16)b[i]; bl[i] = (__bf16)(b[i] - (float)bh[i]);
      }
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
      if (FMT == 0) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
    } else {
      f16x8 ah, al, bh, bl;
      for (int i = 0; i < 8; ++i) {
        ah[i] = (_Float16)a[i]; al[i] = (_Float16)(a[i] - (float)ah[i]);
        bh[i] = (_Float16)b[i]; bl[i] = (_Float16)(b[i] - (float)bh[i]);
      }
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc, 0, 0, 0);
      if (FMT == 1) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);
    }
  }
  for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + j] = acc[r];
}

// the split as the kernel would do it: cvt_pk_f16_f32 + v_fma_mixlo/hi_f16; checked against the C casts
__global__ void split_probe(const float* x, unsigned* hi, unsigned* lo, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (2 * i + 1 >= n) return;
  const float x0 = x[2 * i], x1 = x[2 * i + 1];
  unsigned ph, pl;
  asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(ph) : "v"(x0), "v"(x1));
  asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(pl) : "v"(ph), "v"(x0));
  asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(pl) : "v"(ph), "v"(x1));
  hi[i] = ph; lo[i] = pl;
}

// ---- 3: chunk body timing ----------------------------------------------------------------------------
__device__ __forceinline__ float softplus2(float d) {
  const float r = __builtin_amdgcn_logf(1.f + __builtin_amdgcn_exp2f(d));
  return __builtin_amdgcn_fmed3f(d, r, 127.f);
}
__device__ __forceinline__ float softplus2_light(float d) {
  const float q = fmaxf(fmaf(fabsf(d), -0.18805397f, 0.9767937f), 0.f);
  return fmaf(q, q, fmaxf(d, 0.f));
}
__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {
  const f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
struct Act { u32x4 hi[2], lo[2]; };

// VARIANT: 0 bf16 heavy, 1 f16 heavy, 2 bf16 light, 3 f16 light (packed epilogue), 4 MFMA only (NPASS), 5 f16 heavy with the C split
template <int VARIANT, int NPASS, int NKS>
__global__ __launch_bounds__(512, 2) void chunk_body(float* out, int iters) {
  __shared__ float pad[20 * 1024];   // 80 KiB: two workgroups per CU like the kernel
  pad[threadIdx.x] = threadIdx.x;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  f32x16 accs[2] = {};
  for (int r = 0; r < 16; ++r) accs[1][r] = 0.01f * (lane + r) - 0.3f;
  Act in[7];
  for (int b = 0; b < 7; ++b)
    for (int s = 0; s < 2; ++s)
      for (int q = 0; q < 4; ++q) { in[b].hi[s][q] = 0x3c003c00u + lane + 17 * b; in[b].lo[s][q] = 0x0c000c00u + lane; }
  u32x4 wh, wl;
  for (int q = 0; q < 4; ++q) { wh[q] = 0x2c002c00u + lane * 3; wl[q] = 0x10001000u + lane; }
  Act dst;
  const long long t0 = clock64();
  // two chunks per trip with STATIC register indices (a run-time index into accs / in would go through scratch)
  auto chunk = [&](auto PAR) __attribute__((always_inline)) {
    constexpr int par = decltype(PAR)::value;
    f32x16& acc = accs[par];
    f32x16& a = accs[par ^ 1];
    constexpr int NS = NKS * NPASS, NU = VARIANT == 4 ? 0 : 16;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      const int b = ks < 12 ? (ks >> 1) : 6, sb = ks < 12 ? (ks & 1) : 0;
#pragma unroll
      for (int m = 0; m < NPASS; ++m) {
        const u32x4& w = m == 2 ? wl : wh;
        const u32x4& x = m == 1 ? in[b].lo[sb] : in[b].hi[sb];
        if constexpr (VARIANT == 0 || VARIANT == 2 || VARIANT == 4)
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, x), acc, 0, 0, 0);
        else
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, w), __builtin_bit_cast(f16x8, x), acc, 0, 0, 0);
        const int slot = ks * NPASS + m;
        const int u0 = (slot * NU + NS - 1) / NS, u1 = ((slot + 1) * NU + NS - 1) / NS;
#pragma unroll
        for (int r = u0; r < u1; ++r) {
          if constexpr (VARIANT == 0) {
            a[r] = softplus2(a[r]);
            if (r & 1) {
              const unsigned ph = cvt_pk_bf16(a[r - 1], a[r]);
              dst.hi[r >> 3][(r & 7) >> 1] = ph;
              const float h0 = __builtin_bit_cast(float, ph << 16), h1 = __builtin_bit_cast(float, ph & 0xffff0000u);
              dst.lo[r >> 3][(r & 7) >> 1] = cvt_pk_bf16(a[r - 1] - h0, a[r] - h1);
            }
          } else if constexpr (VARIANT == 1) {
            a[r] = softplus2(a[r]);
            if (r & 1) {
              unsigned ph, pl;
              asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(ph) : "v"(a[r - 1]), "v"(a[r]));
              asm volatile("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(pl) : "v"(ph), "v"(a[r - 1]));
              asm volatile("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(pl) : "v"(ph), "v"(a[r]));
              dst.hi[r >> 3][(r & 7) >> 1] = ph;
              dst.lo[r >> 3][(r & 7) >> 1] = pl;
            }
          } else if constexpr (VARIANT == 5) {
            a[r] = softplus2(a[r]);
            if (r & 1) {
              const f32x2 v = {a[r - 1], a[r]};
              const f16x2 hh = __builtin_convertvector(v, f16x2);
              f16x2 ll;
              ll[0] = (_Float16)(a[r - 1] - (float)hh[0]);
              ll[1] = (_Float16)(a[r] - (float)hh[1]);
              dst.hi[r >> 3][(r & 7) >> 1] = __builtin_bit_cast(unsigned, hh);
              dst.lo[r >> 3][(r & 7) >> 1] = __builtin_bit_cast(unsigned, ll);
            }
          } else if constexpr (VARIANT == 2) {
            a[r] = softplus2_light(a[r]);
            if (r & 1) dst.hi[r >> 3][(r & 7) >> 1] = cvt_pk_bf16(a[r - 1], a[r]);
          } else if constexpr (VARIANT == 3) {
            if (r & 1) {
              const f32x2 v = {a[r - 1], a[r]};
              const f16x2 d = __builtin_convertvector(v, f16x2);
              const f16x2 ad = __builtin_elementwise_max(d, -d);
              const f16x2 c1 = {(_Float16)-0.18805397f, (_Float16)-0.18805397f}, c0 = {(_Float16)0.9767937f, (_Float16)0.9767937f};
              const f16x2 z = {(_Float16)0.f, (_Float16)0.f};
              const f16x2 q = __builtin_elementwise_max(ad * c1 + c0, z);
              const f16x2 o = q * q + __builtin_elementwise_max(d, z);
              dst.hi[r >> 3][(r & 7) >> 1] = __builtin_bit_cast(unsigned, o);
            }
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if constexpr (VARIANT != 4) {
      // the finished operands feed the next chunks' GEMMs (keeps them alive, as in the kernel)
      asm volatile("" : "+v"(dst.hi[0]), "+v"(dst.hi[1]));
      in[par].hi[0] = dst.hi[0]; in[par].hi[1] = dst.hi[1];
      if constexpr (VARIANT == 0 || VARIANT == 1 || VARIANT == 5) {
        asm volatile("" : "+v"(dst.lo[0]), "+v"(dst.lo[1]));
        in[par].lo[0] = dst.lo[0]; in[par].lo[1] = dst.lo[1];
      }
      // fresh pre-activations for the next epilogue: a cheap, exact-ish rescale that keeps exp / log busy with finite values
#pragma unroll
      for (int r = 0; r < 16; ++r) a[r] = __builtin_bit_cast(float, (__builtin_bit_cast(unsigned, acc[r]) & 0x807fffffu) | 0x3e800000u);
    }
  };
  for (int it = 0; it < iters; it += 2) {
    chunk(std::integral_constant<int, 0>{});
    chunk(std::integral_constant<int, 1>{});
  }
  const long long t1 = clock64();
  float s = pad[(threadIdx.x * 7) & 511];
  for (int i = 0; i < 16; ++i) s += accs[0][i] + accs[1][i];
  for (int b = 0; b < 7; ++b) s += __builtin_bit_cast(float, in[b].hi[0][0] ^ in[b].lo[1][3]);
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = float(t1 - t0) / iters;
}

template <int VARIANT, int NPASS, int NKS>
void time_body(const char* name, float* d) {
  float h = 0, ms = 0;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 4000;
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((chunk_body<VARIANT, NPASS, NKS>), dim3(512), dim3(512), 0, 0, d, iters);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    hipEventElapsedTime(&ms, e0, e1);
  }
  hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
  // 512 workgroups x 8 wavefronts on 1024 SIMDs: 4 wavefront-chunks per SIMD and iteration
  printf("%-44s %6.0f ticks per chunk and wavefront, %7.1f ns per chunk and SIMD (%d MFMAs: %.1f ns each)\n", name, h,
         ms * 1e6f / (iters * 4.f), NPASS * NKS, ms * 1e6f / (iters * 4.f) / (NPASS * NKS));
}

int main() {
  // ---- 1: subnormal inputs --------------------------------------------------------------------
  const int KS = 13, K = 16 * KS;
  std::vector<float> A(32 * K), B(K * 32), D(32 * 32);
  float *dA, *dB, *dD;
  hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dD, D.size() * 4);
  for (auto& v : A) v = 3.0e-6f;            // f16 subnormal (min normal 6.1e-5), exactly representable? 3e-6 / 5.96e-8 = 50.3 -> rounds to 50 quanta
  for (auto& v : B) v = 1024.f;
  hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL((gemm_probe<2>), dim3(1), dim3(64), 0, 0, dA, dB, dD, KS);
  hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
  const double q = 50 * 5.9604644775390625e-8;
  printf("subnormal A (3e-6 -> %.6e as f16) x 1024 x K=%d: D = %.6e, expected %.6e (flushed inputs would give 0)\n", q, K, D[0], q * 1024 * K);
  for (auto& v : A) v = 1024.f;
  for (auto& v : B) v = 3.0e-6f;
  hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL((gemm_probe<2>), dim3(1), dim3(64), 0, 0, dA, dB, dD, KS);
  hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
  printf("subnormal B: D = %.6e, expected %.6e\n", D[0], q * 1024 * K);

  // ---- 2: accuracy of the split products ---------------------------------------------------------
  srand(1);
  auto rnd = []() { return (rand() / (float)RAND_MAX) * 2.f - 1.f; };
  for (int scale_case = 0; scale_case < 3; ++scale_case) {
    const float wscale = scale_case == 0 ? 0.0707f : scale_case == 1 ? 0.25f : 1.5f;   // kaiming bound 1/sqrt(200), larger trained weights
    for (auto& v : A) v = rnd() * wscale;
    for (auto& v : B) { const float u = rnd(); v = u > 0 ? u * 300.f : u * 0.01f + 0.011f; }   // scaled softplus outputs: 0 .. 300, many tiny
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    std::vector<double> ref(32 * 32, 0.0);
    double mag = 0;
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
      double s = 0;
      for (int k = 0; k < K; ++k) s += (double)A[i * K + k] * (double)B[k * 32 + j];
      ref[i * 32 + j] = s; mag = fmax(mag, fabs(s));
    }
    const char* names[4] = {"bf16 3-term", "f16 3-term", "f16 2-term (w rounded)", "bf16 2-term (w rounded)"};
    for (int f = 0; f < 4; ++f) {
      if (f == 0) hipLaunchKernelGGL((gemm_probe<0>), dim3(1), dim3(64), 0, 0, dA, dB, dD, KS);
      if (f == 1) hipLaunchKernelGGL((gemm_probe<1>), dim3(1), dim3(64), 0, 0, dA, dB, dD, KS);
      if (f == 2) hipLaunchKernelGGL((gemm_probe<2>), dim3(1), dim3(64), 0, 0, dA, dB, dD, KS);
      if (f == 3) hipLaunchKernelGGL((gemm_probe<3>), dim3(1), dim3(64), 0, 0, dA, dB, dD, KS);
      hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
      double e = 0;
      for (int i = 0; i < 32 * 32; ++i) e = fmax(e, fabs(D[i] - ref[i]));
      printf("w scale %.4f  %-24s max |err| %.3e (max |d| %.1f, scaled domain; / 144.27 in activation units)\n", wscale, names[f], e, mag);
    }
  }

  // ---- split instruction sequence vs casts ----------------------------------------------------------
  {
    const int n = 1 << 16;
    std::vector<float> x(n);
    for (int i = 0; i < n; ++i) x[i] = (i & 1 ? 1.f : -1.f) * expf(rnd() * 12.f);     // 6e-6 .. 1.6e5
    x[0] = 0.f; x[1] = 70000.f; x[2] = 131000.f; x[3] = 1e-7f;
    float* dx; unsigned *dh, *dl;
    hipMalloc(&dx, n * 4); hipMalloc(&dh, n * 2); hipMalloc(&dl, n * 2);
    hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(split_probe, dim3(n / 2 / 256), dim3(256), 0, 0, dx, dh, dl, n);
    std::vector<unsigned> hh(n / 2), ll(n / 2);
    hipMemcpy(hh.data(), dh, n * 2, hipMemcpyDeviceToHost); hipMemcpy(ll.data(), dl, n * 2, hipMemcpyDeviceToHost);
    double worst = 0; int bad = 0;
    for (int i = 0; i < n; ++i) {
      const unsigned short hb = (hh[i / 2] >> (16 * (i & 1))) & 0xffff, lb = (ll[i / 2] >> (16 * (i & 1))) & 0xffff;
      _Float16 hv, lv; memcpy(&hv, &hb, 2); memcpy(&lv, &lb, 2);
      const double rec = (double)(float)hv + (double)(float)lv;
      if (fabs(x[i]) < 60000.f) {
        const double rel = fabs(rec - x[i]) / fmax(fabs((double)x[i]), 1e-30), ab = fabs(rec - x[i]);
        if (ab > 6e-8 && rel > worst) worst = rel;
        if (ab > 6e-8 && rel > 1e-6) ++bad;
      }
      if (i < 4) printf("split x = %.6e -> hi %.6e lo %.6e\n", x[i], (double)(float)hv, (double)(float)lv);
    }
    printf("cvt_pk_f16 + fma_mix split of %d values: worst relative error (where the absolute one exceeds 6e-8) %.3e, bad %d\n", n, worst, bad);
  }

  // ---- 3: chunk bodies ---------------------------------------------------------------------------------
  float* d;
  hipMalloc(&d, 512 * 512 * 4);
  time_body<4, 3, 13>("MFMA only, 39 per chunk (bf16)", d);
  time_body<0, 3, 13>("heavy bf16: 39 MFMA + 16-value epilogue", d);
  time_body<1, 3, 13>("heavy f16 (fma_mix): 39 MFMA + epilogue", d);
  time_body<5, 3, 13>("heavy f16 (C split): 39 MFMA + epilogue", d);
  time_body<0, 3, 7>("heavy bf16, lin2 chunk: 21 MFMA + epilogue", d);
  time_body<1, 3, 7>("heavy f16, lin2 chunk: 21 MFMA + epilogue", d);
  time_body<0, 2, 13>("two-pass bf16: 26 MFMA + epilogue", d);
  time_body<1, 2, 13>("two-pass f16: 26 MFMA + epilogue", d);
  time_body<0, 2, 7>("two-pass bf16, lin2 chunk: 14 MFMA + epilogue", d);
  time_body<1, 2, 7>("two-pass f16, lin2 chunk: 14 MFMA + epilogue", d);
  time_body<4, 1, 13>("MFMA only, 13 per chunk", d);
  time_body<2, 1, 13>("light bf16: 13 MFMA + fp32 poly epilogue", d);
  time_body<3, 1, 13>("light f16: 13 MFMA + packed-f16 epilogue", d);
  time_body<2, 1, 7>("light bf16, lin2 chunk: 7 MFMA + epilogue", d);
  time_body<3, 1, 7>("light f16, lin2 chunk: 7 MFMA + epilogue", d);
  return 0;
}

// Microbenchmark (development tool): wave-time cost of streaming weights L2 -> LDS with LDS-DMA, 8 wavefronts
// per workgroup, one workgroup per CU.  Every wavefront moves PIECES x 1 KiB per round into a ring in LDS:
//   V0  global_load_lds_dwordx4, 64-bit VGPR address, M0 saved / set / restored around every piece
//       (the identity kernel's dma16)
//   V1  the same, M0 set once per piece without save / restore
//   V2  buffer_load_dwordx4 ... offen lds: constant VGPR offset (lane * 16), SGPR soffset per piece
//   V3  V2 with ONE M0 per 4 pieces and inst_offset 0 / 1024 / 2048 / 3072 (4 consecutive groups)
// Reports s_memtime ticks per piece spent ISSUING (wave-time), and the end-to-end GB/s per CU.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef int v4i __attribute__((ext_vector_type(4)));

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int V, int NMFMA>
__global__ __launch_bounds__(512) void k(const char* src, float* out, int rounds, long long* ticks) {
  extern __shared__ char ring[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)ring;
  // buffer resource: base = src, num_records = 1 MiB, gfx950 raw-buffer flags
  v4i rsrc;
  rsrc[0] = (int)(uint32_t)(uintptr_t)src;
  rsrc[1] = (int)(uint32_t)((uintptr_t)src >> 32) & 0xffff;
  rsrc[2] = 1 << 20;
  rsrc[3] = 0x00020000;
  const unsigned voff = lane * 16;
  long long t_issue = 0;
  f32x16 acc = {};
  bf16x8 fa, fb;
  for (int i = 0; i < 8; ++i) { fa[i] = (__bf16)(lane * 0.01f + i); fb[i] = (__bf16)(i * 0.25f); }
  const long long t0 = clock64();
  for (int r = 0; r < rounds; ++r) {
    const unsigned chunk = (r * 26u) & 255u;              // 1 KiB groups, wraps inside 512 KiB
    const unsigned slot = lds0 + (r % 4) * 32768;
    const long long ta = clock64();
    if (V == 0 || V == 1) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const unsigned g = wave + 8 * i;
        const char* p = src + size_t(chunk + g) * 1024 + lane * 16;
        const unsigned dst = __builtin_amdgcn_readfirstlane(slot + g * 1024);
        if (V == 0) {
          unsigned keep;
          asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                       : "=&s"(keep) : "v"(p), "s"(dst) : "memory");
        } else {
          asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(p), "s"(dst) : "memory", "m0");
        }
      }
    } else if (V == 2) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const unsigned g = wave + 8 * i;
        const unsigned soff = __builtin_amdgcn_readfirstlane((chunk + g) * 1024);
        const unsigned dst = __builtin_amdgcn_readfirstlane(slot + g * 1024);
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %3 offen lds"
                     :: "v"(voff), "s"(rsrc), "s"(dst), "s"(soff) : "memory", "m0");
      }
    } else {
      const unsigned g = wave * 4;
      const unsigned soff = __builtin_amdgcn_readfirstlane((chunk + g) * 1024);
      const unsigned dst = __builtin_amdgcn_readfirstlane(slot + g * 1024);
      asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\t"
                   "buffer_load_dwordx4 %0, %1, %3 offen lds\n\t"
                   "buffer_load_dwordx4 %0, %1, %3 offen offset:1024 lds\n\t"
                   "buffer_load_dwordx4 %0, %1, %3 offen offset:2048 lds\n\t"
                   "buffer_load_dwordx4 %0, %1, %3 offen offset:3072 lds"
                   :: "v"(voff), "s"(rsrc), "s"(dst), "s"(soff) : "memory", "m0");
    }
    const long long tb = clock64();
    t_issue += tb - ta;
#pragma unroll
    for (int m = 0; m < NMFMA; ++m) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc, 0, 0, 0);   // the GEMM between two fetches
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");       // the previous round has landed
    __builtin_amdgcn_s_barrier();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const long long t1 = clock64();
  float s = acc[0] + acc[7];
  for (int i = 0; i < 8; ++i) s += reinterpret_cast<float*>(ring)[threadIdx.x + 512 * i];
  out[blockIdx.x * 512 + threadIdx.x] = s;
  if (blockIdx.x == 0 && lane == 0) { ticks[2 * wave] = t_issue; ticks[2 * wave + 1] = t1 - t0; }
}

template <int V, int NMFMA> void run(const char* name, const char* src, float* out, long long* ticks) {
  const int rounds = 2000;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<V, NMFMA>), dim3(256), dim3(512), 128 * 1024, 0, src, out, rounds, ticks);
    (void)hipEventRecord(e1, 0);
    (void)hipDeviceSynchronize();
    (void)hipEventElapsedTime(&ms, e0, e1);
  }
  long long h[16];
  (void)hipMemcpy(h, ticks, sizeof(h), hipMemcpyDeviceToHost);
  double issue = 0; for (int w = 0; w < 8; ++w) issue += double(h[2 * w]);
  printf("%-44s + %2d MFMA/round: issue %.0f ticks/piece (wave-time), round %.0f ticks, %.1f GB/s per CU\n", name, NMFMA,
         issue / 8 / rounds / 4, double(h[1]) / rounds, 32.0 * 1024 * rounds / (ms * 1e-3) / 1e9);
}

int main() {
  char* src; float* out; long long* ticks;
  (void)hipMalloc(&src, 1 << 20); (void)hipMemset(src, 1, 1 << 20);
  (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&ticks, 16 * 8);
#define RUN(V, N, name) (void)hipFuncSetAttribute((const void*)k<V, N>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024); run<V, N>(name, src, out, ticks)
  RUN(0, 0, "V0 global_load_lds, M0 save/set/restore");
  RUN(1, 0, "V1 global_load_lds, M0 set only");
  RUN(2, 0, "V2 buffer_load lds, SGPR offsets");
  RUN(3, 0, "V3 buffer_load lds, 1 M0 + inst_offset x4");
  RUN(0, 39, "V0 global_load_lds, M0 save/set/restore");
  RUN(1, 39, "V1 global_load_lds, M0 set only");
  RUN(2, 39, "V2 buffer_load lds, SGPR offsets");
  RUN(3, 39, "V3 buffer_load lds, 1 M0 + inst_offset x4");
  RUN(3, 0, "(no DMA reference: see round ticks of 39 MFMA = 2 x 39 x 32)");
  return 0;
}

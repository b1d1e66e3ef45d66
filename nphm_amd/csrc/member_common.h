// member_common.h — shared pieces of the member-centric kernels (ident_bwd_kernel.hip: first-order backward
// of the fitting loop; ident_train_kernel.hip: value + spatial gradient and their reverse sweep for training):
// tile geometry, the transposed weight pack, split-bf16 / softplus helpers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>
#include <utility>

#include "layout.h"

namespace nphm {
namespace bwd {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int WAVES = 8;
constexpr int M = 64;                    // points per workgroup (2 m-tiles)
constexpr int MT = 2;
constexpr int NCH = 28;                  // 7 blocks x 4 K chunks of 8 features
constexpr int PLANE_BYTES = NCH * M * 16;

// transposed pack, per weight set (uint16 units): stage A = lin3^T, B = lin2^T (104 rows: h1 | coords),
// C = lin1^T, D = lin0[:, :3]^T; fragments [ob][ks][hi|lo][lane][8] like the forward pack
constexpr int A_OB = 7, A_KS = 13, B_OB = 4, B_KS = 13, C_OB = 7, C_KS = 7, D_OB = 1, D_KS = 13;
constexpr int OFF_A = 0;
constexpr int OFF_B = OFF_A + A_OB * A_KS * 1024;
constexpr int OFF_C = OFF_B + B_OB * B_KS * 1024;
constexpr int OFF_D = OFF_C + C_OB * C_KS * 1024;
constexpr int BWD_SET_STRIDE = OFF_D + D_OB * D_KS * 1024;

__device__ inline uint16_t f32_to_bf16_rn(float x) {
  uint32_t u = __float_as_uint(x);
  uint32_t r = u + 0x7fffu + ((u >> 16) & 1u);
  return uint16_t(r >> 16);
}
__device__ inline float bf16_to_f32(uint16_t v) { return __uint_as_float(uint32_t(v) << 16); }

__device__ __forceinline__ float softplus2(float d) {
  const float t = __builtin_amdgcn_exp2f(-fabsf(d));
  return fmaxf(d, 0.f) + __builtin_amdgcn_logf(1.f + t);   // one v_max_f32 under -fno-honor-nans
}
__device__ __forceinline__ float sigmoid2(float d) {
  const float t = __builtin_amdgcn_exp2f(-fabsf(d));
  return (d >= 0.f ? 1.f : t) * __builtin_amdgcn_rcpf(1.f + t);
}

struct Split8 { bf16x8 hi, lo; };
__device__ __forceinline__ Split8 split8(const float* x) {
  Split8 o;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const __bf16 hb = (__bf16)x[i];
    o.hi[i] = hb;
    o.lo[i] = (__bf16)(x[i] - (float)hb);
  }
  return o;
}

// B operand of the coordinate K-step (prep_kernels.hip, L0 block): h=0: xh | xl | 1 1; h=1: xh | 1 | xll | 0
__device__ __forceinline__ bf16x8 coord_operand(float x, float y, float z, int h) {
  const float cs[3] = {x, y, z};
  __bf16 xh[3], xl[3], xll[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    xh[i] = (__bf16)cs[i];
    const float r1 = cs[i] - (float)xh[i];
    xl[i] = (__bf16)r1;
    xll[i] = (__bf16)(r1 - (float)xl[i]);
  }
  const __bf16 one = (__bf16)1.f, zero = (__bf16)0.f;
  bf16x8 bv;
  bv[0] = xh[0]; bv[1] = xh[1]; bv[2] = xh[2];
  bv[3] = h ? one : xl[0];
  bv[4] = h ? xll[0] : xl[1];
  bv[5] = h ? xll[1] : xl[2];
  bv[6] = h ? xll[2] : one;
  bv[7] = h ? zero : one;
  return bv;
}

__device__ __forceinline__ f32x16 load_frag16(const float* p) {
  const f32x4* q = reinterpret_cast<const f32x4*>(p);
  f32x4 a = q[0], b = q[1], c = q[2], d = q[3];
  f32x16 o;
  o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; o[3] = a[3];
  o[4] = b[0]; o[5] = b[1]; o[6] = b[2]; o[7] = b[3];
  o[8] = c[0]; o[9] = c[1]; o[10] = c[2]; o[11] = c[3];
  o[12] = d[0]; o[13] = d[1]; o[14] = d[2]; o[15] = d[3];
  return o;
}

// sum over the 32 lanes of a half-wave (all lanes of the half end up with the total).  Four DPP steps inside each
// row of 16 lanes (VALU operand modifiers: no LDS crossbar traffic) and ONE cross-row exchange, instead of five
// ds_bpermute round trips (__shfl_xor): the reductions of the bias gradients are 32 of these per wavefront and tile.
#ifndef NPHM_DPP_REDUCE
#define NPHM_DPP_REDUCE 1
#endif
__device__ __forceinline__ float half_wave_sum(float v) {
#if NPHM_DPP_REDUCE
  auto dpp = [](float x, auto ctrl) __attribute__((always_inline)) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value, 0xf, 0xf, true));
  };
  v += dpp(v, std::integral_constant<int, 0xB1>{});      // quad_perm [1,0,3,2]: lane ^ 1
  v += dpp(v, std::integral_constant<int, 0x4E>{});      // quad_perm [2,3,0,1]: lane ^ 2
  v += dpp(v, std::integral_constant<int, 0x141>{});     // row_half_mirror: quads 0 <-> 1, 2 <-> 3 (every lane then holds its 8-lane sum)
  v += dpp(v, std::integral_constant<int, 0x140>{});     // row_mirror: the two halves of the 16-lane row
  return v + __shfl_xor(v, 16);                          // the two rows of the half-wave
#else
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
#endif
}

// ---- per-wavefront LDS ring of weight K-steps, filled by LDS-DMA ------------------------------------------------------
// The member-centric kernels give every wavefront its own output tile, so a weight fragment is read by ONE wavefront, once
// per tile: through VGPRs (two K-steps in flight: all the registers left next to the sweep's sigma' state) every K-step
// waits for most of an L2 round trip - 0.6 us x 13 K-steps x 8 GEMM stages is the 58 us a tile took on a CU that needs 12 us
// of MFMA time for it.  MUBUF loads with the LDS destination (buffer_load ... lds, the fused kernel's weight path,
// eval_kernel.hip) cost no VGPRs and are not tracked by the compiler: a wavefront keeps RING K-steps (2 KiB each: hi | lo
// fragments, contiguous in both packs) in flight into its private slice of LDS and waits with explicit vmcnt counts.
typedef int v4i __attribute__((ext_vector_type(4)));
__device__ __forceinline__ v4i raw_rsrc(const void* base) {
  const uint64_t a = reinterpret_cast<uint64_t>(base);
  v4i r;
  r[0] = __builtin_amdgcn_readfirstlane((int)(uint32_t)a);
  r[1] = __builtin_amdgcn_readfirstlane((int)((uint32_t)(a >> 32) & 0xffffu));   // stride 0: raw buffer
  r[2] = 0x7fffffff;
  r[3] = 0x00020000;                                                             // gfx950 raw-buffer descriptor word
  return r;
}
__device__ __forceinline__ unsigned lds_addr_of(const void* q) {
  return (unsigned)(size_t)(__attribute__((address_space(3))) const char*)q;
}
// one K-step: 2 x 1 KiB (lane's 16 bytes at voff = 16 lane), memory offset soff (SGPR), LDS destination lds_dst (M0 is
// reserved for hipcc: saved and restored inside the statement)
__device__ __forceinline__ void dma_kstep(const v4i& rsrc, unsigned voff, unsigned soff, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
               "buffer_load_dwordx4 %1, %2, %4 offen lds\n\t"
               "buffer_load_dwordx4 %1, %2, %4 offen offset:1024 lds\n\t"
               "s_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(lds_dst), "s"(soff) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "i"(N) : "memory"); }

template <class F, int... I> __device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F> __device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(f, std::make_integer_sequence<int, N>{});
}

}  // namespace bwd
}  // namespace nphm

// mlp_bwd_kernel.hip — first-order backward of the dense skip-MLP (DeepSDF backbone of the forward-deformation
// network, src/NPHM/models/deepSDF.py:64-89 / :184-239) with respect to its CONDITIONING vector, for the latent
// fitting loop (src/NPHM/models/fitting.py:99-106: `decoder_expr(p_corresp, cond)` feeds the implicit
// differentiation of the correspondences; loss.backward() then needs d L / d [z_id-compressed | z_ex] through the
// 7 nn.Linear + 6 Softplus of the backbone - what autograd computes there with ~40 kernels per step).
//
// With the conditioning folded into the biases of lin0 and of the skip layer (mlp_layout.h), d L / d cond =
//   W0[:, lat]^T gb0 + W_skip[:, lat]^T gb_skip / sqrt2,   gb_l = sum over the row's points of d L / d d_l,
// so this kernel returns the two bias gradients gb0, gb_skip [n_rows, hidden]; the host applies the two small
// matrix products (and everything upstream: compressor, anchors) with ordinary autograd.  No weight gradients
// (the fitting loop freezes the decoders), no d/dxyz (the correspondences are detached roots).
//
// Structure = the forward kernel's (mlp_kernel.hip): one workgroup = 8 wavefronts = 64 points, the current
// layer's gradient tile in LDS as split-bf16 K chunks, wavefront w owns output tiles w, w+8 of every stage,
// A fragments (TRANSPOSED weights, own pack) stream L2 -> VGPR, split-bf16 x3 on v_mfma_f32_32x32x16_bf16.
// The forward (KIND 3 of mlp_eval_kernel) left sigma'(d_l) of every hidden layer as register dumps of the owning
// wavefront; stage s multiplies its accumulators by them:
//   G_{L-1} = (W_last^T g / k) * s_{L-1};   G_s = (W_{s+1}^T G_{s+1}) * s_s   (skip layer: its first K columns / sqrt2)
// all in the scaled domain of mlp_layout.h (d' = k d), so gb_l = k * sum_n G_l[n].
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "capi_common.h"
#include "mlp_layout.h"

namespace nphm {
namespace mlp {
namespace bwd {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int NTW = 2;                            // the hidden <= 512 variant of the forward kernel
constexpr int HMAX = 32 * WAVES * NTW;
constexpr int NCH = HMAX / 8;
// MT point tiles of 32 per workgroup: 2 (64 points, the forward's workgroup) or 1 - a workgroup's time is the stream of the
// transposed pack out of L2 (7.3 MB at the CU's 64 B/clk) whatever its columns, so a launch that leaves CUs idle with
// 64-point workgroups (the fitting loop: 5 x 16 = 80 of them) runs twice as many 32-point ones, each faster
constexpr int part_bytes(int mt) { return NCH * 32 * mt * 16; }

__device__ inline uint16_t f32_to_bf16_rn(float x) {
  uint32_t u = __float_as_uint(x);
  uint32_t r = u + 0x7fffu + ((u >> 16) & 1u);
  return uint16_t(r >> 16);
}
__device__ inline float bf16_to_f32(uint16_t v) { return __uint_as_float(uint32_t(v) << 16); }

// ---- transposed pack ------------------------------------------------------------------------------
// stage s = 0 .. n_linear-3 (produces G_s from G_{s+1} through W_{s+1}^T): fragments
//   [n_tile over the features of layer s][k_step over the features of layer s+1][hi|lo][lane][8]
// last stage (G_{L-1} from the 3 .. 4 output gradients): one coordinate-style K-step per tile, [n_tile][lane][8]
// (the layout of the forward's coordinate step, mlp_layout.h: the output gradient plays the coordinates).
struct BwdLayer {
  int n_tiles, k_steps;
  uint32_t w_off;
};
struct BwdPlan {
  int n_stage;                 // n_linear - 1 stages: index n_stage-1 is the output stage
  BwdLayer stage[MAX_LINEAR];
  size_t packed_bytes;
};
inline void make_bwd_plan(const Plan& p, BwdPlan& b) {
  b.n_stage = p.n_linear - 1;
  size_t w = 0;
  for (int s = 0; s < b.n_stage; ++s) {
    BwdLayer& S = b.stage[s];
    S.n_tiles = p.layer[s].n_tiles;                                   // features of layer s, in 32-row tiles
    S.k_steps = s + 1 < b.n_stage ? 2 * p.layer[s + 1].n_tiles : 1;     // features of layer s+1 / one output K-step
    S.w_off = uint32_t(w);
    w += s + 1 < b.n_stage ? size_t(S.n_tiles) * S.k_steps * 2 * 64 * 16 : size_t(S.n_tiles) * 64 * 16;
  }
  b.packed_bytes = w;
}

struct PackArgs {
  const float* w[MAX_LINEAR];
  Plan plan;
  BwdPlan bplan;
  uint16_t* out;
};

__global__ void mlp_pack_bwd_kernel(PackArgs a) {
  const int s = blockIdx.y;
  const BwdLayer& S = a.bplan.stage[s];
  const Layer& Lnext = a.plan.layer[s + 1];
  const int feat_s = a.plan.layer[s].out_dim;            // real features of layer s (rows of this stage)
  uint16_t* out = a.out + S.w_off / 2;
  if (s + 1 < a.bplan.n_stage) {
    const size_t total = size_t(S.n_tiles) * S.k_steps * 2 * 64 * 8;
    for (size_t e = size_t(blockIdx.x) * blockDim.x + threadIdx.x; e < total; e += size_t(gridDim.x) * blockDim.x) {
      const int i = e & 7, lane = (e >> 3) & 63, part = (e >> 9) & 1;
      const size_t gg = e >> 10;
      const int ks = int(gg % S.k_steps), n = int(gg / S.k_steps);
      const int row = 32 * n + (lane & 31);                                        // feature of layer s
      const int kf = 32 * (ks >> 1) + feat_local(8 * (ks & 1) + i, lane >> 5);      // feature of layer s+1
      float w = 0.f;
      // layer s+1 reads layer s through its first k_act columns (act_scale: 1/sqrt2 for the skip layer)
      if (row < feat_s && row < Lnext.k_act && kf < Lnext.out_dim) w = a.w[s + 1][size_t(kf) * Lnext.in_dim + row] * Lnext.act_scale;
      const uint16_t hi = f32_to_bf16_rn(w);
      out[e] = part ? f32_to_bf16_rn(w - bf16_to_f32(hi)) : hi;
    }
  } else {
    // output stage: rows = features of the last hidden layer, "coordinates" = the (<= 3 used) output gradients
    const size_t total = size_t(S.n_tiles) * 64 * 8;
    for (size_t e = size_t(blockIdx.x) * blockDim.x + threadIdx.x; e < total; e += size_t(gridDim.x) * blockDim.x) {
      const int i = e & 7, lane = (e >> 3) & 63, n = int(e >> 9);
      const int row = 32 * n + (lane & 31), hh = lane >> 5;
      auto wc = [&](int c) -> float {
        return (row < feat_s && c < Lnext.out_dim) ? a.w[s + 1][size_t(c) * Lnext.in_dim + row] * Lnext.act_scale : 0.f;
      };
      auto whi = [&](int c) { return f32_to_bf16_rn(wc(c)); };
      auto wlo = [&](int c) { const float w = wc(c); return f32_to_bf16_rn(w - bf16_to_f32(f32_to_bf16_rn(w))); };
      uint16_t v;
      if (hh == 0) v = i < 3 ? whi(i) : i < 6 ? whi(i - 3) : uint16_t(0);
      else v = i < 3 ? wlo(i) : i == 3 ? uint16_t(0) : i < 7 ? whi(i - 4) : uint16_t(0);
      out[e] = v;
    }
  }
}

// ---- backward ---------------------------------------------------------------------------------------
struct Args {
  const char* packed_bwd;
  const float* saved;          // sigma' dumps of the forward (EvalArgs::sig_out)
  const float* gout;           // [n_rows, n_points, out_dim]  d L / d output
  const float* jinv;           // or null: [n_rows, n_points, 3, 3] - gout is the gradient of the implicit root x_c = root - J^-1 (F - F.detach())
                               // (fitting.py:99-106), d L / d output = -J^-T gout (out_dim = 3): applied while loading (a launch of its own before)
  int out_dim;
  int64_t n_points;
  int n_stage;
  BwdLayer stage[MAX_LINEAR];
  int sig_tiles;
  int sig_base[MAX_LINEAR];
  int skip;                    // layer whose bias gradient is returned besides layer 0's
  int hidden;
  int n_slots;                 // ceil(n_points / 32): 32-point slots per row
  float* part;                 // [n_rows][n_slots][2][hidden]: per slot the bias gradients of layer 0 | of the skip layer.  Every
                               // slot is WRITTEN (a 64-point workgroup writes its sums to slot 2b and zeros to 2b + 1): no zero
                               // fill, no atomics - nphm_mlp_cond_grad adds the slots of a row in order
};

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
// B operand of a coordinate-style K-step (mlp_layout.h): h = 0: xh | xl | 1 1, h = 1: xh | 1 | xll | 0 - here with
// the bias slots zeroed (no additive term on this path)
__device__ __forceinline__ bf16x8 grad_operand(float x, float y, float z, int h) {
  const float cs[3] = {x, y, z};
  __bf16 xh[3], xl[3], xll[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    xh[i] = (__bf16)cs[i];
    const float r1 = cs[i] - (float)xh[i];
    xl[i] = (__bf16)r1;
    xll[i] = (__bf16)(r1 - (float)xl[i]);
  }
  const __bf16 zero = (__bf16)0.f;
  bf16x8 bv;
  bv[0] = xh[0]; bv[1] = xh[1]; bv[2] = xh[2];
  bv[3] = h ? zero : xl[0];
  bv[4] = h ? xll[0] : xl[1];
  bv[5] = h ? xll[1] : xl[2];
  bv[6] = h ? xll[2] : zero;
  bv[7] = zero;
  return bv;
}
__device__ __forceinline__ float half_wave_sum(float v) {
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

template <int MT>
__global__ __launch_bounds__(64 * WAVES, 2) void mlp_bwd_kernel(Args p) {
  constexpr int M = 32 * MT, PART_BYTES = part_bytes(MT);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* act_hi = smem;
  char* act_lo = smem + PART_BYTES;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int h = lane >> 5, j = lane & 31;
  const int row = blockIdx.y;
  const int64_t base = int64_t(blockIdx.x) * M;
  // saved layout: [row][64-point group][sig_tiles][2 point tiles][64 lanes][16]; a 32-point workgroup reads one tile of a group
  const size_t groups = size_t((p.n_points + 63) >> 6), group = size_t(base >> 6);
  const int tile0 = MT == 2 ? 0 : int(base >> 5) & 1;
  const float* sig_wg = p.saved + (size_t(row) * groups + group) * p.sig_tiles * (2 * 64 * 16);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.packed_bwd), 0, 0x7fffffff, 0x00020000);
  const f32x16 zero16 = {};
  auto tiles_of = [&](int n_tiles) { return n_tiles > wave ? (n_tiles - wave + WAVES - 1) / WAVES : 0; };

  f32x16 acc[NTW][MT];
  Split8 packed_out[NTW][MT][2];

  // G tile (n, t) of stage s: accumulators x sigma'_s; bias gradients of layer 0 / the skip layer on the way
  auto finish = [&](int s, int ni) __attribute__((always_inline)) {
    const int slot = MT == 1 ? int(blockIdx.x) : 2 * int(blockIdx.x);
    float* gb = (s == 0 || s == p.skip) ? p.part + ((size_t(row) * p.n_slots + slot) * 2 + (s == 0 ? 0 : 1)) * p.hidden : nullptr;
#pragma unroll
    for (int i = 0; i < NTW; ++i) {
      if (i < ni) {
        const int n = wave + WAVES * i;
        float v[MT][16];
#pragma unroll
        for (int t = 0; t < MT; ++t) {
          const float4* sg = reinterpret_cast<const float4*>(sig_wg + ((size_t(p.sig_base[s] + n) * 2 + tile0 + t) * 64 + lane) * 16);
          // points past the row's end: the value+Jacobian forward leaves their slots unwritten (whatever the allocator
          // handed out, NaN bit patterns included) and their accumulators are exact zeros - keep 0 x NaN out of the sums
          const bool in_row = base + 32 * t + j < p.n_points;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float4 s4 = in_row ? sg[q] : make_float4(0.f, 0.f, 0.f, 0.f);
            v[t][4 * q] = acc[i][t][4 * q] * s4.x; v[t][4 * q + 1] = acc[i][t][4 * q + 1] * s4.y;
            v[t][4 * q + 2] = acc[i][t][4 * q + 2] * s4.z; v[t][4 * q + 3] = acc[i][t][4 * q + 3] * s4.w;
          }
          packed_out[i][t][0] = split8(v[t]);
          packed_out[i][t][1] = split8(v[t] + 8);
        }
        if (gb) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float sum = 0.f;
#pragma unroll
            for (int t = 0; t < MT; ++t) sum += v[t][r];
            sum = half_wave_sum(sum);
            const int f = 32 * n + feat_local(r, h);
            if (j == 0 && f < p.hidden) {
              gb[f] = sum * SP_SCALE;
              if (MT == 2 && slot + 1 < p.n_slots) gb[size_t(2) * p.hidden + f] = 0.f;
            }
          }
        }
      }
    }
  };
  auto store_tiles = [&](int ni) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NTW; ++i) {
      if (i < ni) {
        const int n = wave + WAVES * i;
#pragma unroll
        for (int t = 0; t < MT; ++t) {
#pragma unroll
          for (int half = 0; half < 2; ++half) {
            const int off = ((4 * n + 2 * half + h) * M + 32 * t + j) * 16;
            *reinterpret_cast<bf16x8*>(act_hi + off) = packed_out[i][t][half].hi;
            *reinterpret_cast<bf16x8*>(act_lo + off) = packed_out[i][t][half].lo;
          }
        }
      }
    }
  };

  // ---- output stage: G_{L-1} = (W_last^T g) * s_{L-1} --------------------------------------------------
  {
    const int s = p.n_stage - 1;
    const BwdLayer& S = p.stage[s];
    const int ni = tiles_of(S.n_tiles);
    bf16x8 bv[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      const int64_t i = base + 32 * t + j;
      float g[3] = {0.f, 0.f, 0.f};
      if (i < p.n_points) {
        const float* q = p.gout + (int64_t(row) * p.n_points + i) * p.out_dim;
#pragma unroll
        for (int c = 0; c < 3; ++c) g[c] = c < p.out_dim ? q[c] : 0.f;
        if (p.jinv) {
          const float* J = p.jinv + (int64_t(row) * p.n_points + i) * 9;
          const float gx = g[0], gy = g[1], gz = g[2];
#pragma unroll
          for (int c = 0; c < 3; ++c) g[c] = -(J[c] * gx + J[3 + c] * gy + J[6 + c] * gz);
        }
      }
      bv[t] = grad_operand(g[0], g[1], g[2], h);
    }
    const bf16x8* C = reinterpret_cast<const bf16x8*>(p.packed_bwd + S.w_off) + lane;
#pragma unroll
    for (int i = 0; i < NTW; ++i) {
      if (i < ni) {
        const bf16x8 a = C[(wave + WAVES * i) * 64];
#pragma unroll
        for (int t = 0; t < MT; ++t) acc[i][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bv[t], zero16, 0, 0, 0);
      }
    }
    finish(s, ni);
    store_tiles(ni);
  }

  // ---- hidden stages: G_s = (W_{s+1}^T G_{s+1}) * s_s ----------------------------------------------------
#pragma unroll 1
  for (int s = p.n_stage - 2; s >= 0; --s) {
    const BwdLayer& S = p.stage[s];
    const int ni = tiles_of(S.n_tiles);
    const int ks = S.k_steps;
#pragma unroll
    for (int i = 0; i < NTW; ++i)
#pragma unroll
      for (int t = 0; t < MT; ++t) acc[i][t] = zero16;
    __syncthreads();                                  // the previous stage's tile is complete
    if (ni > 0) {
      const unsigned w_lane = lane * 16;
      const bf16x8* Bh = reinterpret_cast<const bf16x8*>(act_hi) + h * M + j;
      const bf16x8* Bl = reinterpret_cast<const bf16x8*>(act_lo) + h * M + j;
      // A fragments: NS K-steps in flight (32-point workgroups: 6 MFMAs per step and registers to spare - four; mlp_kernel.hip)
      constexpr int NS = MT == 1 ? 4 : 2;
      bf16x8 ah[NS][NTW], al[NS][NTW], bh[2][MT], bl[2][MT];
      auto load_a = [&](int slot, int k) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NTW; ++i) {
          if (i < ni) {
            const unsigned o = __builtin_amdgcn_readfirstlane(S.w_off + (unsigned(wave + WAVES * i) * unsigned(ks) + unsigned(k)) * 2048u);
            ah[slot][i] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_w, w_lane, o, 0));
            al[slot][i] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_w, w_lane, o + 1024u, 0));
          }
        }
      };
      auto load_b = [&](int slot, int k) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < MT; ++t) {
          bh[slot][t] = Bh[2 * k * M + 32 * t];
          bl[slot][t] = Bl[2 * k * M + 32 * t];
        }
      };
      auto mma = [&](int sa, int sb) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NTW; ++i) {
          if (i < ni) {
#pragma unroll
            for (int t = 0; t < MT; ++t) {
              acc[i][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[sa][i], bh[sb][t], acc[i][t], 0, 0, 0);
              acc[i][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[sa][i], bl[sb][t], acc[i][t], 0, 0, 0);
              acc[i][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[sa][i], bh[sb][t], acc[i][t], 0, 0, 0);
            }
          }
        }
      };
#pragma unroll
      for (int u = 0; u < NS; ++u) if (u < 2 || u < ks) load_a(u, u);
      load_b(0, 0);
      // slot u holds K-step k + u; the B operand alternates its two slots (k_steps is even: the steps u >= 2 of the last
      // round may not exist)
#pragma unroll 1
      for (int k = 0; k < ks; k += NS) {
#pragma unroll
        for (int u = 0; u < NS; ++u) {
          if (u < 2 || k + u < ks) {
            if (k + u + 1 < ks) load_b((u + 1) & 1, k + u + 1);
            __builtin_amdgcn_sched_barrier(0);
            mma(u, u & 1);
            __builtin_amdgcn_sched_barrier(0);
            if (k + u + NS < ks) load_a(u, k + u + NS);
          }
        }
      }
    }
    finish(s, ni);
    __syncthreads();                                  // every wavefront has read the old tile
    if (s > 0) store_tiles(ni);
  }
}

constexpr size_t lds_bytes(int mt) { return size_t(2) * part_bytes(mt); }

}  // namespace bwd
}  // namespace mlp
}  // namespace nphm

// ============================================================================================
// C ABI (include/nphm_amd.h)
// ============================================================================================
using nphm::mlp::Config;
using nphm::mlp::Plan;

extern "C" {

static bool bwd_plan_of(int lat_dim, int hidden_dim, int nlayers, int out_dim, Plan& plan, nphm::mlp::bwd::BwdPlan& bplan) {
  Config c{lat_dim, hidden_dim, nlayers, out_dim};
  if (!nphm::mlp::make_plan(c, plan) || plan.variant != 0 || out_dim > 3) return false;
  nphm::mlp::bwd::make_bwd_plan(plan, bplan);
  return true;
}

size_t nphm_mlp_bwd_packed_bytes(int lat_dim, int hidden_dim, int nlayers, int out_dim) {
  Plan plan;
  nphm::mlp::bwd::BwdPlan b;
  return bwd_plan_of(lat_dim, hidden_dim, nlayers, out_dim, plan, b) ? b.packed_bytes : 0;
}

int nphm_mlp_pack_bwd(int lat_dim, int hidden_dim, int nlayers, int out_dim, const float* const* lin_weight,
                      void* packed_bwd, void* stream) {
  nphm::mlp::bwd::PackArgs a;
  if (!bwd_plan_of(lat_dim, hidden_dim, nlayers, out_dim, a.plan, a.bplan))
    return nphm_fail_msg("nphm_mlp_pack_bwd: unsupported architecture (backward covers hidden <= 512, out_dim <= 3)");
  if (!lin_weight || !packed_bwd) return nphm_fail_msg("nphm_mlp_pack_bwd: null pointer");
  for (int l = 0; l < a.plan.n_linear; ++l) {
    if (!lin_weight[l]) return nphm_fail_msg("nphm_mlp_pack_bwd: null weight pointer");
    a.w[l] = lin_weight[l];
  }
  a.out = static_cast<uint16_t*>(packed_bwd);
  hipLaunchKernelGGL(nphm::mlp::bwd::mlp_pack_bwd_kernel, dim3(512, a.bplan.n_stage), dim3(256), 0,
                     static_cast<hipStream_t>(stream), a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return nphm_fail("nphm_mlp_pack_bwd launch", e);
  return 0;
}

size_t nphm_mlp_bwd_partial_bytes(int hidden_dim, int n_rows, int64_t n_points) {
  return hidden_dim <= 0 || n_rows <= 0 || n_points <= 0 ? 0 : size_t(n_rows) * size_t((n_points + 31) / 32) * 2 * size_t(hidden_dim) * 4;
}

int nphm_mlp_backward_cond(int lat_dim, int hidden_dim, int nlayers, int out_dim, const void* packed_bwd,
                           const void* saved, const float* grad_out, const float* root_jac_inverse, int n_rows, int64_t n_points,
                           void* bias_partials, void* stream) {
  Plan plan;
  nphm::mlp::bwd::BwdPlan b;
  if (!bwd_plan_of(lat_dim, hidden_dim, nlayers, out_dim, plan, b))
    return nphm_fail_msg("nphm_mlp_backward_cond: unsupported architecture (backward covers hidden <= 512, out_dim <= 3)");
  if (!packed_bwd || !saved || !grad_out || !bias_partials) return nphm_fail_msg("nphm_mlp_backward_cond: null pointer");
  if (n_rows <= 0 || n_points <= 0) return nphm_fail_msg("nphm_mlp_backward_cond: empty input");
  nphm::mlp::bwd::Args a;
  memset(&a, 0, sizeof(a));
  a.packed_bwd = static_cast<const char*>(packed_bwd);
  a.saved = static_cast<const float*>(saved);
  a.gout = grad_out;
  if (root_jac_inverse && out_dim != 3) return nphm_fail_msg("nphm_mlp_backward_cond: the implicit-root gradient needs out_dim = 3");
  a.jinv = root_jac_inverse;
  a.out_dim = out_dim;
  a.n_points = n_points;
  a.n_stage = b.n_stage;
  for (int s = 0; s < b.n_stage; ++s) {
    a.stage[s] = b.stage[s];
    a.sig_base[s] = a.sig_tiles;
    a.sig_tiles += plan.layer[s].n_tiles;
  }
  a.skip = nlayers / 2;
  a.hidden = hidden_dim;
  a.n_slots = int((n_points + 31) / 32);
  a.part = static_cast<float*>(bias_partials);
  // 64-point workgroups unless their 32-point halves still fit one round of the chip (see part_bytes)
  int cus = 0, dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 0;
  const int64_t wgs32 = (n_points + 31) / 32;
  const char* force = getenv("NPHM_AMD_MLP_BWD_POINTS");      // dev knob: 64 / 32 pins the workgroup shape (A/B runs)
  const bool small = force && atoi(force) == 64 ? false : force && atoi(force) == 32 ? true : wgs32 * n_rows <= cus;
  const int64_t wgs = small ? wgs32 : (n_points + 63) / 64;
  if (wgs > 0x7fffffffLL) return nphm_fail_msg("nphm_mlp_backward_cond: too many points for one launch");
  auto k = small ? nphm::mlp::bwd::mlp_bwd_kernel<1> : nphm::mlp::bwd::mlp_bwd_kernel<2>;
  const size_t lds = nphm::mlp::bwd::lds_bytes(small ? 1 : 2);
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
  if (e != hipSuccess) return nphm_fail("nphm_mlp_backward_cond: LDS opt-in", e);
  hipLaunchKernelGGL(k, dim3((unsigned)wgs, n_rows), dim3(64 * nphm::mlp::WAVES), lds, static_cast<hipStream_t>(stream), a);
  e = hipGetLastError();
  if (e != hipSuccess) return nphm_fail("nphm_mlp_backward_cond launch", e);
  return 0;
}

}  // extern "C"

// prep_kernels.hip — one-off re-layout of the NPHM identity weights and the per-latent prologue.
//
//   pack_f32_kernel / pack_bf16_kernel   state_dict tensors -> MFMA A-fragment order (once per
//                                        weight update); replaces EnsembledLinear.forward's per-call
//                                        repeat_interleave / cat / permute (EnsembledDeepSDF.py:43-54)
//   prepare_latent_kernel                per latent row: anchors = mlp_pos(z_glob) + mean anchors
//                                        (EnsembledDeepSDF.py:228-229); the latent columns of lin0 and
//                                        of the skip part of lin2 folded into per-member vectors
//                                        (replaces the cond tensor, EnsembledDeepSDF.py:247-255)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "capi_common.h"
#include "layout.h"

#ifndef NPHM_STACK_TAILS
#define NPHM_STACK_TAILS 1   // must match eval_kernel.hip: stacked [wh; wl] fragments of the split-f16 tail blocks
#endif

namespace nphm {

struct PackArgs {
  const float* w[5];
  const float* b[5];
  float* out_f32;
  uint16_t* out_bf16;
  uint16_t* out_f16;
};

__device__ inline uint16_t f32_to_bf16_rn(float x) {
  uint32_t u = __float_as_uint(x);
  uint32_t r = u + 0x7fffu + ((u >> 16) & 1u);
  return uint16_t(r >> 16);
}
__device__ inline float bf16_to_f32(uint16_t v) { return __uint_as_float(uint32_t(v) << 16); }
// IEEE binary16, round to nearest even (subnormals kept)
__device__ inline uint16_t f32_to_f16_rn(float x) { return __builtin_bit_cast(uint16_t, (_Float16)x); }
__device__ inline float f16_to_f32(uint16_t v) { return (float)__builtin_bit_cast(_Float16, v); }

// value of the (row, k-feature) entry of the GEMM layer L (1,2,3) for weight set s
__device__ inline float layer_weight(const PackArgs& a, int L, int s, int row, int kf) {
  if (L == 1) {
    return (row < L1_OUT && kf < HID) ? a.w[1][(size_t(s) * L1_OUT + row) * HID + kf] : 0.f;
  } else if (L == 2) {
    // input of lin2 is [x(101) | coords(3) | cond(96)] / sqrt(2) (EnsembledDeepSDF.py:115-116);
    // columns 0..103 stay in the GEMM with the 1/sqrt(2) folded into the weight; the coordinate
    // columns 101..103 multiply UNSCALED inputs and therefore carry the activation scale k
    if (!(row < HID && kf < L2_IN)) return 0.f;
    const float w = a.w[2][(size_t(s) * HID + row) * HID + kf] / INV_SQRT2_DIV;
    return kf >= L1_OUT ? w * SP_SCALE : w;
  } else {
    return (row < HID && kf < HID) ? a.w[3][(size_t(s) * HID + row) * HID + kf] : 0.f;
  }
}

__global__ void pack_f32_kernel(PackArgs a) {
  const int s = blockIdx.y;
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= SET_STRIDE) return;
  float val = 0.f;
  if (e < OFF_L4B) {
    int L, x, nks, full;
    if (e < OFF_L2A) { L = 1; x = e - OFF_L1A; nks = L1_KS; full = 6; }
    else if (e < OFF_L3A) { L = 2; x = e - OFF_L2A; nks = L2_KS; full = 3; }
    else { L = 3; x = e - OFF_L3A; nks = L3_KS; full = 6; }
    const int c = x & 3, lane = (x >> 2) & 63, gg = x >> 8;
    const int g = gg % (nks / 4), ob = gg / (nks / 4);
    const int ks = 4 * g + c;
    val = layer_weight(a, L, s, 32 * ob + (lane & 31), feat_of(ks_block(ks, full), ks_reg(ks, full), lane >> 5));
  } else if (e == OFF_L4B) {
    val = a.b[4][s];
  }
  a.out_f32[size_t(s) * SET_STRIDE + e] = val;
}

// split-bf16 fragments: [ob][kstep][hi|lo][lane][8]; k-slot 8*h+i of K-step (b, sub) is feature
// feat_of(b, 8*sub + i, h)
__global__ void pack_bf16_kernel(PackArgs a) {
  const int s = blockIdx.y;
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= BF_SET_STRIDE) return;
  int L, x, nks, full;
  if (e < BF_OFF_L2A) { L = 1; x = e - BF_OFF_L1A; nks = L1_KS16; full = 6; }
  else if (e < BF_OFF_L3A) { L = 2; x = e - BF_OFF_L2A; nks = L2_KS16; full = 3; }
  else { L = 3; x = e - BF_OFF_L3A; nks = L3_KS16; full = 6; }
  const int i = x & 7, lane = (x >> 3) & 63, part = (x >> 9) & 1, gg = x >> 10;
  const int ks = gg % nks, ob = gg / nks;
  const int b = ks < 2 * full ? (ks >> 1) : full;
  const int sub = ks < 2 * full ? (ks & 1) : 0;
  const int h = lane >> 5;
  const float w = layer_weight(a, L, s, 32 * ob + (lane & 31), feat_of(b, 8 * sub + i, h));
  const uint16_t hi = f32_to_bf16_rn(w);
  const uint16_t lo = f32_to_bf16_rn(w - bf16_to_f32(hi));
  a.out_bf16[size_t(s) * BF_SET_STRIDE + e] = part ? lo : hi;
  // the split-f16 fragments: same position, binary16 halves ...
  const uint16_t hh = f32_to_f16_rn(w);
  const uint16_t hl = f32_to_f16_rn(w - f16_to_f32(hh));
  uint16_t* out16 = a.out_f16 + size_t(s) * BF_SET_STRIDE;
#if NPHM_STACK_TAILS
  // ... except the LAST 32-row block of every layer (8 real rows, 5 for lin1): ONE stacked fragment per K-step at the front
  // of the chunk, [ks][lane][8], with wh in rows 0..7 and wl of the same weights in rows 8..15 (eval_kernel.hip: two MFMAs
  // per K-step instead of three, half the bytes to stream); the second half of the chunk stays zero
  if (ob == (L == 1 ? L1_OB : L == 2 ? L2_OB : L3_OB) - 1) {
    const int chunk0 = x - ((x >> 10) % nks) * 1024 - part * 512 - (lane * 8 + i);     // first element of this chunk
    const int r = lane & 31;
    if (part == 0) {
      uint16_t v = 0;
      if (r < 8) v = hh;                                   // (this thread's w belongs to row 32 ob + r: the hi half)
      else if (r < 16) {
        const float w8 = layer_weight(a, L, s, 32 * ob + r - 8, feat_of(b, 8 * sub + i, h));
        v = f32_to_f16_rn(w8 - f16_to_f32(f32_to_f16_rn(w8)));
      }
      out16[(L == 1 ? BF_OFF_L1A : L == 2 ? BF_OFF_L2A : BF_OFF_L3A) + chunk0 + ks * 512 + lane * 8 + i] = v;
    } else {
      out16[(L == 1 ? BF_OFF_L1A : L == 2 ? BF_OFF_L2A : BF_OFF_L3A) + chunk0 + nks * 512 + ks * 512 + lane * 8 + i] = 0;
    }
    return;
  }
#endif
  out16[e] = part ? hl : hh;
}

// ------------------------------------------------------------------------------------------
// prepare_latent: grid (41, n_rows).  blocks 0..39 fold member k, block 40 runs mlp_pos.
// ------------------------------------------------------------------------------------------
struct PrepArgs {
  const float* w[5];
  const float* b[5];
  const float* pw[3];
  const float* pb[3];
  int pos_dim;
  const float* anchors_mean;
  const float* lat_rows;   // [n_rows, LAT_DIM]
  float* state;            // [n_rows, LS_ROW_STRIDE]
  float* anchors_out;      // [n_rows, 39, 3] (may be null)
  const float* anchors_in; // [n_rows, 39, 3] or null: given instead of evaluated
};

__global__ __launch_bounds__(1024) void prepare_latent_kernel(PrepArgs a) {
  __shared__ float sh[2 * 256 + 96];
  const int row = blockIdx.y;
  const int t = threadIdx.x;
  const float* lat = a.lat_rows + size_t(row) * LAT_DIM;
  float* st = a.state + size_t(row) * LS_ROW_STRIDE;
  if (blockIdx.x < N_MEMBERS) {
    const int k = blockIdx.x;
    const int s = member_set(k);
    float* cond = sh;            // [96]
    float* b0f = sh + 96;        // [224] folded lin0 bias by FEATURE index (0 beyond 200)
    if (t < LAT_COND) cond[t] = t < LAT_GLOB ? lat[t] : lat[LAT_GLOB + LAT_LOC * k + (t - LAT_GLOB)];
    __syncthreads();
    // folded biases: b0'[f] = b0[f] + W0[f,3:] . cond ; b2'[f] = b2[f] + W2[f,104:] . cond / sqrt2.  A wavefront per
    // feature row: the lanes read the row's 96 latent columns coalesced (64 + 32) and meet in a butterfly sum - a thread per
    // row walked its row with 96 dependent, uncoalesced loads (the prologue is on the critical chain of a fitting step)
    float* b2f = sh + 96 + 224;  // [224]
    static_assert(LAT_COND == 96, "two column groups per lane");
    {
      const int lane = t & 63, wv = t >> 6, nw = blockDim.x >> 6;
      const float c0 = cond[lane], c1 = lane < 32 ? cond[64 + lane] : 0.f;
      const float d0 = c0 / INV_SQRT2_DIV, d1 = c1 / INV_SQRT2_DIV;
      constexpr int RB = 7;                       // rows per round trip: 4 x 7 loads in flight per lane
      for (int r0 = wv * RB; r0 < 224; r0 += nw * RB) {
        float p0[RB], p2[RB];
#pragma unroll
        for (int q = 0; q < RB; ++q) {
          const int r = min(r0 + q, HID - 1);
          const float* w0 = a.w[0] + (size_t(s) * HID + r) * D_IN + 3;
          const float* w2 = a.w[2] + (size_t(s) * HID + r) * HID + L2_IN;
          const int l1 = 64 + (lane & 31);
          const float a0 = w0[lane], a1 = w0[l1], e0 = w2[lane], e1 = w2[l1];
          p0[q] = fmaf(a1, c1, a0 * c0);          // c1 = d1 = 0 on lanes 32..63
          p2[q] = fmaf(e1, d1, e0 * d0);
        }
#pragma unroll
        for (int q = 0; q < RB; ++q) {
          float v0 = p0[q], v2 = p2[q];
#pragma unroll
          for (int m = 32; m > 0; m >>= 1) { v0 += __shfl_xor(v0, m); v2 += __shfl_xor(v2, m); }
          const int r = r0 + q;
          if (lane == 0 && r < 224) {
            b0f[r] = r < HID ? a.b[0][s * HID + r] + v0 : 0.f;
            b2f[r] = r < HID ? a.b[2][s * HID + r] + v2 : 0.f;
          }
        }
      }
    }
    __syncthreads();

    // chunk tails [18][64]: accumulator init (bias * k) per 32-row block, lin4 weights / k for L3
    float* tail = st + LS_OFF_TAIL + size_t(k) * GEMM_CHUNKS * TAIL_FLOATS;
    for (int e = t; e < GEMM_CHUNKS * TAIL_FLOATS; e += blockDim.x) {
      const int ci = e / TAIL_FLOATS, x = e % TAIL_FLOATS;
      const int part = x >> 5, h = (x >> 4) & 1, r = x & 15;
      float v = 0.f;
      if (ci < L1_OB) {
        const int f = feat_of(ci, r, h);
        if (part == 0 && f < L1_OUT) v = a.b[1][s * L1_OUT + f] * SP_SCALE;
      } else if (ci < L1_OB + L2_OB) {
        const int f = feat_of(ci - L1_OB, r, h);
        if (part == 0 && f < HID) v = b2f[f] * SP_SCALE;
      } else {
        const int f = feat_of(ci - L1_OB - L2_OB, r, h);
        if (f < HID) v = part == 0 ? a.b[3][s * HID + f] * SP_SCALE : a.w[4][s * HID + f] / SP_SCALE;
      }
      tail[e] = v;
    }

    // L0 block, fp32: A fragments of two K=2 steps per 32-row block:
    //   ks 0: A[i][0] = k*w_x, A[i][1] = k*w_y ; ks 1: A[i][0] = k*w_z, A[i][1] = k*b0'
    // (B operand in the kernel: ks 0 -> (c_x, c_y), ks 1 -> (c_z, 1))
    float* l0f = st + LS_OFF_L0F + size_t(k) * L0_BLOCK_FLOATS;
    for (int e = t; e < L0_BLOCK_FLOATS; e += blockDim.x) {
      float v = 0.f;
      if (e < 7 * 2 * 64) {
        const int lane = e & 63, ks = (e >> 6) & 1, ob = e >> 7;
        const int f = 32 * ob + (lane & 31), hh = lane >> 5;
        if (f < HID) {
          const float* w0 = a.w[0] + (size_t(s) * HID + f) * D_IN;
          v = ks == 0 ? w0[hh] : (hh == 0 ? w0[2] : b0f[f]);
          v *= SP_SCALE;
        }
      }
      l0f[e] = v;
    }

    // L0 block, split bf16: ONE K=16 step per 32-row block holds every product of
    //   (xh + xl + xll) * (wh + wl)  [minus the 2^-16 terms]  +  bias (3 bf16 terms) * 1
    //   lanes h=0, slots 0..7 : wh_x wh_y wh_z | wh_x wh_y wh_z | b_hi b_mid
    //   lanes h=1, slots 8..15: wl_x wl_y wl_z | b_lo | wh_x wh_y wh_z | 0
    //   (B operand:       h=0 : xh_x xh_y xh_z | xl_x xl_y xl_z | 1 1
    //                     h=1 : xh_x xh_y xh_z | 1 | xll_x xll_y xll_z | 0)
    uint16_t* l0b = reinterpret_cast<uint16_t*>(st + LS_OFF_L0B + size_t(k) * L0_BLOCK_FLOATS);
    for (int e = t; e < 2 * L0_BLOCK_FLOATS; e += blockDim.x) {
      uint16_t v = 0;
      if (e < 7 * 64 * 8) {
        const int i = e & 7, lane = (e >> 3) & 63, ob = e >> 9;
        const int f = 32 * ob + (lane & 31), hh = lane >> 5;
        if (f < HID) {
          const float* w0 = a.w[0] + (size_t(s) * HID + f) * D_IN;
          const float bias = b0f[f] * SP_SCALE;
          const uint16_t bh = f32_to_bf16_rn(bias);
          const float r1 = bias - bf16_to_f32(bh);
          const uint16_t bm = f32_to_bf16_rn(r1);
          const uint16_t bl = f32_to_bf16_rn(r1 - bf16_to_f32(bm));
          auto whi = [&](int c) { return f32_to_bf16_rn(w0[c] * SP_SCALE); };
          auto wlo = [&](int c) {
            const float w = w0[c] * SP_SCALE;
            return f32_to_bf16_rn(w - bf16_to_f32(f32_to_bf16_rn(w)));
          };
          if (hh == 0) v = i < 3 ? whi(i) : i < 6 ? whi(i - 3) : i == 6 ? bh : bm;
          else v = i < 3 ? wlo(i) : i == 3 ? bl : i < 7 ? whi(i - 4) : uint16_t(0);
        }
      }
      l0b[e] = v;
    }
    // L0 block, split f16: the same K = 16 step with binary16 halves
    uint16_t* l0h = reinterpret_cast<uint16_t*>(st + LS_OFF_L0H + size_t(k) * L0_BLOCK_FLOATS);
    for (int e = t; e < 2 * L0_BLOCK_FLOATS; e += blockDim.x) {
      uint16_t v = 0;
      if (e < 7 * 64 * 8) {
        const int i = e & 7, lane = (e >> 3) & 63, ob = e >> 9;
        const int f = 32 * ob + (lane & 31), hh = lane >> 5;
        if (f < HID) {
          const float* w0 = a.w[0] + (size_t(s) * HID + f) * D_IN;
          const float bias = b0f[f] * SP_SCALE;
          const uint16_t bh = f32_to_f16_rn(bias);
          const float r1 = bias - f16_to_f32(bh);
          const uint16_t bm = f32_to_f16_rn(r1);
          const uint16_t bl = f32_to_f16_rn(r1 - f16_to_f32(bm));
          auto whi = [&](int c) { return f32_to_f16_rn(w0[c] * SP_SCALE); };
          auto wlo = [&](int c) {
            const float w = w0[c] * SP_SCALE;
            return f32_to_f16_rn(w - f16_to_f32(f32_to_f16_rn(w)));
          };
          if (hh == 0) v = i < 3 ? whi(i) : i < 6 ? whi(i - 3) : i == 6 ? bh : bm;
          else v = i < 3 ? wlo(i) : i == 3 ? bl : i < 7 ? whi(i - 4) : uint16_t(0);
        }
      }
      l0h[e] = v;
    }
  } else if (a.anchors_in) {
    // the caller holds the anchors of these rows already (the autograd tier evaluates mlp_pos as a differentiable head)
    for (int o = t; o < N_LOC * 3; o += blockDim.x) {
      const float v = a.anchors_in[size_t(row) * N_LOC * 3 + o];
      st[LS_OFF_ANCH + o] = v;
      if (a.anchors_out) a.anchors_out[size_t(row) * N_LOC * 3 + o] = v;
    }
    for (int o = t; o < N_MEMBERS * 4; o += blockDim.x) st[LS_OFF_BND + o] = (o & 3) == 0 ? 1.f : 0.f;
  } else {
    // anchors = mlp_pos(z_glob) + mean anchors (EnsembledDeepSDF.py:228-229)
    float* h1 = sh;
    float* h2 = sh + 256;
    const int P = a.pos_dim;
    for (int o = t; o < P; o += blockDim.x) {
      float v = a.pb[0][o];
      for (int j = 0; j < LAT_GLOB; ++j) v = fmaf(a.pw[0][o * LAT_GLOB + j], lat[j], v);
      h1[o] = fmaxf(v, 0.f);
    }
    __syncthreads();
    for (int o = t; o < P; o += blockDim.x) {
      float v = a.pb[1][o];
      for (int j = 0; j < P; ++j) v = fmaf(a.pw[1][o * P + j], h1[j], v);
      h2[o] = fmaxf(v, 0.f);
    }
    __syncthreads();
    for (int o = t; o < N_LOC * 3; o += blockDim.x) {
      float v = a.pb[2][o];
      for (int j = 0; j < P; ++j) v = fmaf(a.pw[2][o * P + j], h2[j], v);
      v += a.anchors_mean[o];
      st[LS_OFF_ANCH + o] = v;
      if (a.anchors_out) a.anchors_out[size_t(row) * N_LOC * 3 + o] = v;
    }
    // member magnitude bounds: the plain-weight rule until nphm_identity_set_member_bounds installs fitted ones
    for (int o = t; o < N_MEMBERS * 4; o += blockDim.x) st[LS_OFF_BND + o] = (o & 3) == 0 ? 1.f : 0.f;
  }
}

__global__ void set_bounds_kernel(float* state, const float* bounds, int n_rows) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_rows * N_MEMBERS * 4) return;
  const int row = i / (N_MEMBERS * 4), o = i % (N_MEMBERS * 4);
  state[size_t(row) * LS_ROW_STRIDE + LS_OFF_BND + o] = bounds ? bounds[o] : ((o & 3) == 0 ? 1.f : 0.f);
}

}  // namespace nphm

// ============================================================================================
// C ABI (include/nphm_amd.h)
// ============================================================================================
extern "C" {

int nphm_abi_version(void) { return NPHM_AMD_ABI_VERSION; }
const char* nphm_last_error(void) { return nphm_err_buf(); }

int nphm_identity_supported(int lat_dim_glob, int lat_dim_loc, int n_loc, int n_symm_pairs,
                            int hidden_dim, int n_layers, int out_dim, int input_dim) {
  return lat_dim_glob == nphm::LAT_GLOB && lat_dim_loc == nphm::LAT_LOC && n_loc == nphm::N_LOC &&
         n_symm_pairs == nphm::N_SYMM && hidden_dim == nphm::HID && n_layers == 4 && out_dim == 1 &&
         input_dim == 3;
}

size_t nphm_identity_packed_bytes(void) { return nphm::PACKED_BYTES; }
size_t nphm_identity_latent_state_bytes(int n_rows) {
  return size_t(n_rows) * nphm::LS_ROW_STRIDE * sizeof(float);
}

int nphm_identity_pack(const float* const lin_weight[5], const float* const lin_bias[5],
                       void* packed, void* stream) {
  if (!packed) return nphm_fail_msg("nphm_identity_pack: null packed buffer");
  nphm::PackArgs a;
  for (int i = 0; i < 5; ++i) {
    if (!lin_weight[i] || !lin_bias[i]) return nphm_fail_msg("nphm_identity_pack: null weight/bias pointer");
    a.w[i] = lin_weight[i];
    a.b[i] = lin_bias[i];
  }
  a.out_f32 = static_cast<float*>(packed);
  a.out_bf16 = reinterpret_cast<uint16_t*>(static_cast<char*>(packed) + nphm::PACKED_F32_FLOATS * 4);
  a.out_f16 = a.out_bf16 + nphm::PACKED_BF16_HALFS;
  hipStream_t st = static_cast<hipStream_t>(stream);
  dim3 g1((nphm::SET_STRIDE + 255) / 256, nphm::N_SETS);
  hipLaunchKernelGGL(nphm::pack_f32_kernel, g1, dim3(256), 0, st, a);
  dim3 g2((nphm::BF_SET_STRIDE + 255) / 256, nphm::N_SETS);
  hipLaunchKernelGGL(nphm::pack_bf16_kernel, g2, dim3(256), 0, st, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return nphm_fail("nphm_identity_pack launch", e);
  return 0;
}

int nphm_identity_prepare_latent(const void* packed,
                                 const float* const lin_weight[5], const float* const lin_bias[5],
                                 const float* const mlp_pos_weight[3], const float* const mlp_pos_bias[3],
                                 int pos_mlp_dim, const float* anchors_mean,
                                 const float* lat_rows, int n_rows,
                                 void* latent_state, float* anchors_out, void* stream) {
  (void)packed;
  if (n_rows <= 0) return nphm_fail_msg("nphm_identity_prepare_latent: n_rows must be > 0");
  if (pos_mlp_dim <= 0 || pos_mlp_dim > 256)
    return nphm_fail_msg("nphm_identity_prepare_latent: pos_mlp_dim must be in 1..256");
  if (!lat_rows || !latent_state || !anchors_mean)
    return nphm_fail_msg("nphm_identity_prepare_latent: null pointer");
  nphm::PrepArgs a;
  for (int i = 0; i < 5; ++i) { a.w[i] = lin_weight[i]; a.b[i] = lin_bias[i]; }
  for (int i = 0; i < 3; ++i) { a.pw[i] = mlp_pos_weight[i]; a.pb[i] = mlp_pos_bias[i]; }
  a.pos_dim = pos_mlp_dim;
  a.anchors_mean = anchors_mean;
  a.lat_rows = lat_rows;
  a.state = static_cast<float*>(latent_state);
  a.anchors_out = anchors_out;
  a.anchors_in = nullptr;
  hipLaunchKernelGGL(nphm::prepare_latent_kernel, dim3(nphm::N_MEMBERS + 1, n_rows), dim3(1024), 0,
                     static_cast<hipStream_t>(stream), a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return nphm_fail("nphm_identity_prepare_latent launch", e);
  return 0;
}

int nphm_identity_prepare_latent_anchors(const float* const lin_weight[5], const float* const lin_bias[5],
                                         const float* lat_rows, const float* anchors, int n_rows, void* latent_state,
                                         void* stream) {
  if (n_rows <= 0) return nphm_fail_msg("nphm_identity_prepare_latent_anchors: n_rows must be > 0");
  if (!lat_rows || !latent_state || !anchors) return nphm_fail_msg("nphm_identity_prepare_latent_anchors: null pointer");
  nphm::PrepArgs a{};
  for (int i = 0; i < 5; ++i) { a.w[i] = lin_weight[i]; a.b[i] = lin_bias[i]; }
  a.lat_rows = lat_rows;
  a.state = static_cast<float*>(latent_state);
  a.anchors_in = anchors;
  hipLaunchKernelGGL(nphm::prepare_latent_kernel, dim3(nphm::N_MEMBERS + 1, n_rows), dim3(1024), 0,
                     static_cast<hipStream_t>(stream), a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return nphm_fail("nphm_identity_prepare_latent_anchors launch", e);
  return 0;
}

int nphm_identity_set_member_bounds(void* latent_state, int n_rows, const float* bounds, void* stream) {
  if (!latent_state || n_rows <= 0) return nphm_fail_msg("nphm_identity_set_member_bounds: bad arguments");
  const int n = n_rows * nphm::N_MEMBERS * 4;
  hipLaunchKernelGGL(nphm::set_bounds_kernel, dim3((n + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<float*>(latent_state), bounds, n_rows);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return nphm_fail("nphm_identity_set_member_bounds launch", e);
  return 0;
}

}  // extern "C"

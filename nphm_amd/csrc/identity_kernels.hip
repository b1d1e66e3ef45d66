// identity_kernels.hip — gfx950 kernels for the NPHM identity field
// (FastEnsembleDeepSDFMirrored, src/NPHM/models/EnsembledDeepSDF.py:153-267 of the reference).
//
//   pack_kernel            state_dict tensors -> MFMA fragment order            (once per weights)
//   prepare_latent_kernel  mlp_pos anchors + latent folded into biases          (once per latent)
//   eval_kernel<MODE,PREC> fused 40-member MLP ensemble + Gaussian blend        (the hot kernel)
//
// eval_kernel: one wavefront owns 32 query points for the whole network; activations never leave
// registers (see layout.h).  Members whose normalised blend weight is below the prune tolerance
// for all 32 points are skipped (wave-uniform branch).  MODE 0 reads xyz[n,3]; MODE 1 generates
// the 'ij' grid coordinates from three axis arrays and maps a workgroup to a compact 4x4x8 voxel
// brick so that the 32 points of a wavefront (4x4x2) share their set of active members.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "layout.h"

namespace nphm {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------
// pack
// ------------------------------------------------------------------------------------------
struct PackArgs {
  const float* w[5];
  const float* b[5];
  float* out_f32;
  uint16_t* out_bf16;
};

__device__ inline uint16_t f32_to_bf16_rn(float x) {
  uint32_t u = __float_as_uint(x);
  uint32_t r = u + 0x7fffu + ((u >> 16) & 1u);
  return uint16_t(r >> 16);
}
__device__ inline float bf16_to_f32(uint16_t v) { return __uint_as_float(uint32_t(v) << 16); }

// value of the (row, k-feature) entry of the GEMM layer L (1,2,3) for weight set s
__device__ inline float layer_weight(const PackArgs& a, int L, int s, int row, int kf) {
  if (L == 1) {
    return (row < L1_OUT && kf < HID) ? a.w[1][(size_t(s) * L1_OUT + row) * HID + kf] : 0.f;
  } else if (L == 2) {
    // input of lin2 is [x(101) | coords(3) | cond(96)] / sqrt(2) (EnsembledDeepSDF.py:115-116);
    // columns 0..103 stay in the GEMM, the 1/sqrt(2) is folded into the weight
    return (row < HID && kf < L2_IN) ? a.w[2][(size_t(s) * HID + row) * HID + kf] / 1.41421356237f : 0.f;
  } else {
    return (row < HID && kf < HID) ? a.w[3][(size_t(s) * HID + row) * HID + kf] : 0.f;
  }
}

__global__ void pack_f32_kernel(PackArgs a) {
  const int s = blockIdx.y;
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= SET_STRIDE) return;
  float val = 0.f;
  if (e < OFF_L1A) {
    const int c = e & 3, q = e >> 2;
    const int h = q & 1, r = (q >> 1) & 15, b = q >> 5;
    const int f = feat_of(b, r, h);
    if (f < HID && c < 3) val = a.w[0][(size_t(s) * HID + f) * D_IN + c];
  } else if (e < OFF_L2A) {
    const int x = e - OFF_L1A;
    const int c = x & 3, lane = (x >> 2) & 63, gg = x >> 8;
    const int g = gg % (L1_KS / 4), ob = gg / (L1_KS / 4);
    const int ks = 4 * g + c;
    val = layer_weight(a, 1, s, 32 * ob + (lane & 31), feat_of(ks_block(ks, 6), ks_reg(ks, 6), lane >> 5));
  } else if (e < OFF_L3A) {
    const int x = e - OFF_L2A;
    const int c = x & 3, lane = (x >> 2) & 63, gg = x >> 8;
    const int g = gg % (L2_KS / 4), ob = gg / (L2_KS / 4);
    const int ks = 4 * g + c;
    val = layer_weight(a, 2, s, 32 * ob + (lane & 31), feat_of(ks_block(ks, 3), ks_reg(ks, 3), lane >> 5));
  } else if (e < OFF_L4B) {
    const int x = e - OFF_L3A;
    const int c = x & 3, lane = (x >> 2) & 63, gg = x >> 8;
    const int g = gg % (L3_KS / 4), ob = gg / (L3_KS / 4);
    const int ks = 4 * g + c;
    val = layer_weight(a, 3, s, 32 * ob + (lane & 31), feat_of(ks_block(ks, 6), ks_reg(ks, 6), lane >> 5));
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
};

__global__ __launch_bounds__(256) void prepare_latent_kernel(PrepArgs a) {
  __shared__ float sh[2 * 256 + 96];
  const int row = blockIdx.y;
  const int t = threadIdx.x;
  const float* lat = a.lat_rows + size_t(row) * LAT_DIM;
  float* st = a.state + size_t(row) * LS_ROW_STRIDE;
  if (blockIdx.x < N_MEMBERS) {
    const int k = blockIdx.x;
    const int s = member_set(k);
    float* cond = sh;
    if (t < LAT_COND) cond[t] = t < LAT_GLOB ? lat[t] : lat[LAT_GLOB + LAT_LOC * k + (t - LAT_GLOB)];
    __syncthreads();
    if (t < 224) {
      const int b = t >> 5, h = (t >> 4) & 1, r = t & 15;
      const int f = feat_of(b, r, h);
      float v0 = 0.f, v2 = 0.f;
      if (f < HID) {
        const float* w0 = a.w[0] + (size_t(s) * HID + f) * D_IN + 3;
        const float* w2 = a.w[2] + (size_t(s) * HID + f) * HID + L2_IN;
        v0 = a.b[0][s * HID + f];
        v2 = a.b[2][s * HID + f];
        for (int j = 0; j < LAT_COND; ++j) {
          const float c = cond[j];
          v0 = fmaf(w0[j], c, v0);
          v2 = fmaf(w2[j], c / 1.41421356237f, v2);
        }
      }
      st[LS_OFF_B0 + k * 224 + t] = v0;
      // chunk tails: accumulator init per 32-row block (+ lin4 weights for the L3 blocks)
      float* tail = st + LS_OFF_TAIL + size_t(k) * CHUNKS_PER_MEMBER * TAIL_FLOATS;
      const int hr = t & 31;                                   // h * 16 + r
      if (b < L1_OB) tail[b * TAIL_FLOATS + hr] = f < L1_OUT ? a.b[1][s * L1_OUT + f] : 0.f;
      tail[(L1_OB + b) * TAIL_FLOATS + hr] = v2;
      tail[(L1_OB + L2_OB + b) * TAIL_FLOATS + hr] = f < HID ? a.b[3][s * HID + f] : 0.f;
      tail[(L1_OB + L2_OB + b) * TAIL_FLOATS + 32 + hr] = f < HID ? a.w[4][s * HID + f] : 0.f;
      if (b < L1_OB) tail[b * TAIL_FLOATS + 32 + hr] = 0.f;
      tail[(L1_OB + b) * TAIL_FLOATS + 32 + hr] = 0.f;
    }
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
  }
}

// ------------------------------------------------------------------------------------------
// eval
// ------------------------------------------------------------------------------------------
struct EvalArgs {
  const float* packed_f32;
  const uint16_t* packed_bf16;
  const float* state;       // [n_rows, LS_ROW_STRIDE]
  float* out;
  unsigned long long* stats;
  float prune_tol;
  // MODE 0 (points)
  const float* xyz;         // [n_rows, n_points, 3]
  int64_t n_points;
  // MODE 1 (grid)
  const float* ax; const float* ay; const float* az;
  int rx, ry, rz, ix0, ix1;
  int nbx, nby, nbz;        // bricks per axis
  int nsx, nsy, nsz;        // super-bricks (2x4x4 bricks) per axis
  int64_t hack_chunk;
};

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// nn.Softplus(beta=100, threshold=20) (EnsembledDeepSDF.py:99): for 100x > 20 PyTorch returns x;
// here log(1 + exp(-|100x|)) is already 0 in fp32 for |100x| > 16.7, so the two agree to < 1e-9.
__device__ __forceinline__ float softplus100(float x) {
  // raw v_exp_f32 / v_log_f32 (base 2): the argument of the log is in [1, 2], no range fix-ups
  const float t = __builtin_amdgcn_exp2f(-144.26950408889634f * fabsf(x));      // exp(-100|x|)
  return fmaf(0.0069314718055994531f, __builtin_amdgcn_logf(1.f + t), fmaxf(x, 0.f));
}

// Pin values at this program point.  Without a use in the producing basic block LLVM sinks the
// (pure) softplus arithmetic across the next workgroup barrier, next to the MFMAs that consume it,
// which keeps pre- AND post-activation values live and spills.
__device__ __forceinline__ void pin16(f32x16& v) {
#pragma unroll
  for (int r = 0; r < 16; ++r) asm volatile("" : "+v"(v[r]));
}

__device__ __forceinline__ f32x16 softplus100_v(f32x16 d) {
  f32x16 o;
#pragma unroll
  for (int r = 0; r < 16; ++r) o[r] = softplus100(d[r]);
  pin16(o);
  return o;
}

// one 32-feature block of activations as split-bf16 B operands of v_mfma_f32_32x32x16_bf16:
// K-step s consumes registers 8s..8s+7 of the block (k-slot 8h+i <-> register 8s+i)
struct ActB {
  bf16x8 hi[2], lo[2];
};

__device__ __forceinline__ void split_block(const f32x16& v, ActB& o) {
#pragma unroll
  for (int s = 0; s < 2; ++s) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float x = v[8 * s + i];
      const __bf16 hb = (__bf16)x;
      o.hi[s][i] = hb;
      o.lo[s][i] = (__bf16)(x - (float)hb);
    }
  }
  // pin the packed operands (same reason as pin16)
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    asm volatile("" : "+v"(o.hi[s]));
    asm volatile("" : "+v"(o.lo[s]));
  }
}

__device__ __forceinline__ f32x16 load_frag16(const float* p) {
  // 16 consecutive floats (64-byte aligned) -> f32x16
  const f32x4* q = reinterpret_cast<const f32x4*>(p);
  f32x4 a = q[0], b = q[1], c = q[2], d = q[3];
  f32x16 o;
  o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; o[3] = a[3];
  o[4] = b[0]; o[5] = b[1]; o[6] = b[2]; o[7] = b[3];
  o[8] = c[0]; o[9] = c[1]; o[10] = c[2]; o[11] = c[3];
  o[12] = d[0]; o[13] = d[1]; o[14] = d[2]; o[15] = d[3];
  return o;
}

__device__ __forceinline__ f32x16 load_frag16_lds(unsigned int lds_byte_addr) {
  typedef __attribute__((address_space(3))) const f32x4* lds_v4;
  lds_v4 q = (lds_v4)(size_t)lds_byte_addr;
  f32x4 a = q[0], b = q[1], c = q[2], d = q[3];
  f32x16 o;
  o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; o[3] = a[3];
  o[4] = b[0]; o[5] = b[1]; o[6] = b[2]; o[7] = b[3];
  o[8] = c[0]; o[9] = c[1]; o[10] = c[2]; o[11] = c[3];
  o[12] = d[0]; o[13] = d[1]; o[14] = d[2]; o[15] = d[3];
  return o;
}

// ---- weight streaming: global -> LDS ring, shared by the NW wavefronts of a workgroup ----------
// A member's GEMM weights are consumed as 18 chunks (one per 32-row output block, layout.h), in
// 1 KiB groups (one global_load_lds_dwordx4 of a wavefront):
//   fp32 : L1 25, L2 13, L3 25 groups ([ks/4][lane][4] floats)
//   bf16 : L1 26, L2 14, L3 26 groups ([ks][hi|lo][lane][8] bf16)
// plus a 256-byte tail per chunk (accumulator init / lin4 weights, from the per-latent state).
// Chunk c+2 is fetched with global_load_lds (no VGPR staging) while chunk c is consumed.
template <int PREC> struct Stream;
template <> struct Stream<0> {
  static constexpr int MAIN_BYTES = (L1_KS / 4) * 1024;
  __device__ static __forceinline__ const char* set_base(const EvalArgs& p, int s) {
    return reinterpret_cast<const char*>(p.packed_f32 + size_t(s) * SET_STRIDE);
  }
  __device__ static __forceinline__ int offset(int ci) {   // bytes inside a weight set
    return 4 * (ci < L1_OB ? OFF_L1A + ci * (L1_KS / 4) * 256
              : ci < L1_OB + L2_OB ? OFF_L2A + (ci - L1_OB) * (L2_KS / 4) * 256
                                   : OFF_L3A + (ci - L1_OB - L2_OB) * (L3_KS / 4) * 256);
  }
  __device__ static __forceinline__ int groups(int ci) {
    return (ci >= L1_OB && ci < L1_OB + L2_OB) ? (L2_KS / 4) : (L1_KS / 4);
  }
};
template <> struct Stream<1> {
  static constexpr int MAIN_BYTES = L1_KS16 * 2 * 1024;
  __device__ static __forceinline__ const char* set_base(const EvalArgs& p, int s) {
    return reinterpret_cast<const char*>(p.packed_bf16 + size_t(s) * BF_SET_STRIDE);
  }
  __device__ static __forceinline__ int offset(int ci) {
    return 2 * (ci < L1_OB ? BF_OFF_L1A + ci * L1_KS16 * 1024
              : ci < L1_OB + L2_OB ? BF_OFF_L2A + (ci - L1_OB) * L2_KS16 * 1024
                                   : BF_OFF_L3A + (ci - L1_OB - L2_OB) * L3_KS16 * 1024);
  }
  __device__ static __forceinline__ int groups(int ci) {
    return (ci >= L1_OB && ci < L1_OB + L2_OB) ? 2 * L2_KS16 : 2 * L1_KS16;
  }
};
constexpr int RING = 3;

// NW = wavefronts per workgroup (all of them share one LDS ring): 8 -> 256 points per weight pass
#ifndef NPHM_NW
#define NPHM_NW 8
#endif
constexpr int NW = NPHM_NW;
// voxel brick of a workgroup in grid mode: every wavefront owns a 4x4x2 sub-brick
constexpr int BRX = NW == 8 ? 8 : 4, BRY = NW == 8 ? 8 : 4, BRZ = NW == 8 ? 4 : 8;
// bricks are enumerated super-brick by super-brick (2x4x4 bricks) so that the workgroups resident
// on one XCD at any time cover a compact region and stream the same few members (L2 reuse)
constexpr int SBX = 2, SBY = 4, SBZ = 4;

template <int PREC>
struct Streamer {
  static constexpr int SLOT_BYTES = Stream<PREC>::MAIN_BYTES + TAIL_FLOATS * 4;
  const EvalArgs& p;
  const float* tails;          // per-latent state: chunk tails of this batch row
  char* ring;                  // LDS, RING * SLOT_BYTES
  const unsigned char* list;   // LDS, active member ids of this workgroup
  int n_active;
  int mi;                      // index (into list) of the member being consumed
  int wave, lane;

  // fetch chunk `ci` (compile-time) of list member `m` into ring slot `slot`
  __device__ __forceinline__ void issue(int m, const int ci, const int slot) const {
    if (m >= n_active) return;
    const int k = __builtin_amdgcn_readfirstlane(int(list[m]));
    int l = lane;
    asm volatile("" : "+v"(l));           // per-site addresses are recomputed, not hoisted (VGPRs)
    const char* src = Stream<PREC>::set_base(p, member_set(k)) + Stream<PREC>::offset(ci) + l * 16;
    char* dst = ring + slot * SLOT_BYTES;
    const int ng = Stream<PREC>::groups(ci);
#pragma unroll 1
    for (int g = wave; g < ng; g += NW)
      __builtin_amdgcn_global_load_lds(src + g * 1024, (__attribute__((address_space(3))) void*)(dst + g * 1024), 16, 0, 0);
    if (wave == (ci & (NW - 1)))
      __builtin_amdgcn_global_load_lds(tails + (k * CHUNKS_PER_MEMBER + ci) * TAIL_FLOATS + l,
                                       (__attribute__((address_space(3))) void*)(dst + Stream<PREC>::MAIN_BYTES), 4, 0, 0);
  }
  // Every wavefront of the workgroup calls this once per chunk, in lockstep order.  CI = index of
  // the chunk inside its member (18 % RING == 0, so ring slots are compile-time constants).
  __device__ __forceinline__ const char* acquire(const int CI) {
    __syncthreads();                      // chunk CI has landed (own loads waited, then barrier)
    // the buffer of the previous chunk is free: prefetch two chunks ahead
    issue(CI + 2 < CHUNKS_PER_MEMBER ? mi : mi + 1, (CI + 2) % CHUNKS_PER_MEMBER, (CI + 2) % RING);
    return ring + (CI % RING) * SLOT_BYTES;
  }
  __device__ __forceinline__ void next_member() {
    ++mi;
    asm volatile("" : "+s"(mi));          // opaque: no per-site precomputation hoisted out of the loop
  }
  __device__ static __forceinline__ const float* tail_of(const char* buf) {
    return reinterpret_cast<const float*>(buf + Stream<PREC>::MAIN_BYTES);
  }
};

// One 32-row output block on fp32 MFMA: acc += sum_ks A(ks) x IN(block(ks))[reg(ks)].
// A fragments come from LDS: [ks/4][lane][4].
template <int NKS, int FULL, int NIN>
__device__ __forceinline__ f32x16 gemm_block_f32(const char* afrag, f32x16 acc,
                                                 const f32x16 (&in)[NIN], int lane) {
  const f32x4* A = reinterpret_cast<const f32x4*>(afrag) + lane;
#pragma unroll
  for (int g = 0; g < NKS / 4; ++g) {
    const f32x4 a = A[g * 64];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int ks = 4 * g + c;
      const int b = ks < 16 * FULL ? (ks >> 4) : FULL;
      const int r = ks < 16 * FULL ? (ks & 15) : ks - 16 * FULL;
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c], in[b][r], acc, 0, 0, 0);
    }
  }
  return acc;
}

// The same block on split-bf16 MFMA: x*w ~= xh*wh + xl*wh + xh*wl (fp32 accumulate); the dropped
// xl*wl term is 2^-16 relative.  A fragments from LDS: [ks][hi|lo][lane][8].
template <int NKS16, int FULL, int NIN>
__device__ __forceinline__ f32x16 gemm_block_bf16(const char* afrag, f32x16 acc,
                                                  const ActB (&in)[NIN], int lane) {
  const bf16x8* A = reinterpret_cast<const bf16x8*>(afrag) + lane;
#pragma unroll
  for (int ks = 0; ks < NKS16; ++ks) {
    const bf16x8 wh = A[(2 * ks) * 64];
    const bf16x8 wl = A[(2 * ks + 1) * 64];
    const int b = ks < 2 * FULL ? (ks >> 1) : FULL;
    const int s = ks < 2 * FULL ? (ks & 1) : 0;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, in[b].hi[s], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, in[b].lo[s], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, in[b].hi[s], acc, 0, 0, 0);
  }
  return acc;
}

template <int MODE, int PREC>
__global__ __launch_bounds__(64 * NW, 2) void eval_kernel(EvalArgs p) {
  using WS = Streamer<PREC>;
  __shared__ __attribute__((aligned(16))) char ring[RING * WS::SLOT_BYTES];
  __shared__ unsigned int wg_mask[2];
  __shared__ unsigned char wg_list[N_MEMBERS];

  const int lane_inv = threadIdx.x & 63;
  const int lane = lane_inv;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int h_inv = lane >> 5;
  const int h = h_inv;
  const int j = lane & 31;

  // ---- locate this lane's query point ------------------------------------------------------
  bool valid;
  int64_t out_idx;
  bool hack = false;
  float qx, qy, qz;
  int row = 0;
  if (MODE == 0) {
    row = blockIdx.y;
    const int64_t i = (int64_t(blockIdx.x) * NW + wave) * 32 + j;
    valid = i < p.n_points;
    const int64_t ic = valid ? i : (p.n_points - 1);
    const float* q = p.xyz + (int64_t(row) * p.n_points + ic) * 3;
    qx = q[0]; qy = q[1]; qz = q[2];
    out_idx = int64_t(row) * p.n_points + ic;
    if (p.hack_chunk > 0) hack = ((ic + 1) % p.hack_chunk == 0) || (ic == p.n_points - 1);
  } else {
    // The hardware places block b on XCD b % 8: XCD x works on super-bricks x, x+8, x+16, ...
    // (interleaved, so every XCD sees the same mix of near-surface and empty space) and its
    // consecutive blocks are the bricks of ONE super-brick.
    constexpr int SBN = SBX * SBY * SBZ;
    const int local = blockIdx.x >> 3;
    const int inner = local % SBN;
    int bid = (local / SBN) * 8 + (blockIdx.x & 7);          // super-brick index
    const int sz = bid % p.nsz; bid /= p.nsz;
    const int sy = bid % p.nsy; bid /= p.nsy;
    const int bx = bid * SBX + inner / (SBY * SBZ);
    const int by = sy * SBY + (inner / SBZ) % SBY;
    const int bz = sz * SBZ + inner % SBZ;
    const int wx = NW == 8 ? (wave & 1) : 0, wy = NW == 8 ? ((wave >> 1) & 1) : 0;
    const int wz = NW == 8 ? (wave >> 2) : wave;
    const int ix = p.ix0 + bx * BRX + wx * 4 + (j >> 3);
    const int iy = by * BRY + wy * 4 + ((j >> 1) & 3);
    const int iz = bz * BRZ + wz * 2 + (j & 1);
    valid = ix < p.ix1 && iy < p.ry && iz < p.rz;
    const int cx_ = min(ix, p.ix1 - 1), cy_ = min(iy, p.ry - 1), cz_ = min(iz, p.rz - 1);
    qx = p.ax[cx_]; qy = p.ay[cy_]; qz = p.az[cz_];
    const int64_t gi = (int64_t(cx_) * p.ry + cy_) * p.rz + cz_;
    out_idx = gi - int64_t(p.ix0) * p.ry * p.rz;
    if (p.hack_chunk > 0)
      hack = ((gi + 1) % p.hack_chunk == 0) || (gi == int64_t(p.rx) * p.ry * p.rz - 1);
  }

  const float* st = p.state + size_t(row) * LS_ROW_STRIDE;
  const float* anch = st + LS_OFF_ANCH;

  // ---- blend normaliser and active-member mask (EnsembledDeepSDF.py:129-150) -----------------
  float S = 0.f;
#pragma unroll 1
  for (int k = 0; k < N_LOC; ++k) {
    const float dx = anch[3 * k] - qx, dy = anch[3 * k + 1] - qy, dz = anch[3 * k + 2] - qz;
    const float d = sqrtf(dx * dx + dy * dy + dz * dz) + 1e-5f;
    S += expf(-(d * d) / 0.01f);
  }
  const float w_bg = expf(-0.2f / 0.01f);
  S += w_bg;
  const float denom = S + 1e-6f;
  const float thr = p.prune_tol * denom;
  uint64_t wmask = 0;                    // members this wavefront evaluates (wave-uniform)
  const bool any_valid = __ballot(valid) != 0ull;
  if (p.prune_tol < 0.f) {
    wmask = any_valid ? (1ull << N_MEMBERS) - 1 : 0ull;
  } else {
#pragma unroll 1
    for (int k = 0; k < N_MEMBERS; ++k) {
      float w = w_bg;
      if (k < N_LOC) {
        const float dx = anch[3 * k] - qx, dy = anch[3 * k + 1] - qy, dz = anch[3 * k + 2] - qz;
        const float d = sqrtf(dx * dx + dy * dy + dz * dz) + 1e-5f;
        w = expf(-(d * d) / 0.01f);
      }
      if (__ballot(valid && !hack && w > thr) != 0ull) wmask |= 1ull << k;
    }
  }

  const unsigned long long nv = __popcll(__ballot(valid)) >> 1;   // both half-waves hold the same points
  if (p.stats && lane == 0) {
    atomicAdd(p.stats, nv * __popcll(wmask));
    atomicAdd(p.stats + 1, nv);
  }

  // ---- union over the workgroup: the members whose weights get streamed ----------------------
  if (threadIdx.x < 2) wg_mask[threadIdx.x] = 0u;
  __syncthreads();
  if (lane == 0) {
    atomicOr(&wg_mask[0], (unsigned int)(wmask & 0xffffffffull));
    atomicOr(&wg_mask[1], (unsigned int)(wmask >> 32));
  }
  __syncthreads();
  const uint64_t gmask = (uint64_t(wg_mask[1]) << 32) | wg_mask[0];
  const int n_active = __popcll(gmask);
  if (threadIdx.x < N_MEMBERS) {
    if ((gmask >> threadIdx.x) & 1ull)
      wg_list[__popcll(gmask & ((1ull << threadIdx.x) - 1))] = (unsigned char)threadIdx.x;
  }
  __syncthreads();
  WS ws{p, st + LS_OFF_TAIL, ring, wg_list, n_active, 0, wave, lane};
  ws.issue(0, 0, 0);
  ws.issue(0, 1, 1);

  float acc = 0.f;

#pragma unroll 1
  for (int mi = 0; mi < n_active; ++mi) {
    const int k = wg_list[mi];
    if (!((wmask >> k) & 1ull)) {
      // this wavefront's 32 points do not need member k: keep the ring moving only
#pragma unroll
      for (int c = 0; c < CHUNKS_PER_MEMBER; ++c) (void)ws.acquire(c);
      ws.next_member();
      continue;
    }
    // lane-derived values are re-materialised per member (opaque to LICM): hoisting the dozens of
    // per-site lane offsets out of this loop costs more registers than recomputing them
    int h = h_inv, lane = lane_inv;
    asm volatile("" : "+v"(h), "+v"(lane));
    const int s = member_set(k);
    const float* setp = p.packed_f32 + size_t(s) * SET_STRIDE;

    // local coordinates (EnsembledDeepSDF.py:240-244): anchor-relative, odd member of a
    // symmetric pair mirrored in x, background member uses global coordinates
    float cx = qx, cy = qy, cz = qz;
    float wk = w_bg;
    if (k < N_LOC) {
      const float ax = anch[3 * k], ay = anch[3 * k + 1], az = anch[3 * k + 2];
      cx = qx - ax; cy = qy - ay; cz = qz - az;
      const float dx = ax - qx, dy = ay - qy, dz = az - qz;
      const float d = sqrtf(dx * dx + dy * dy + dz * dz) + 1e-5f;
      wk = expf(-(d * d) / 0.01f);
    }
    if (k < 2 * N_SYMM && (k & 1)) cx = -cx;

    const f32x4* l0w = reinterpret_cast<const f32x4*>(setp + OFF_L0W);
    const float* b0 = st + LS_OFF_B0 + k * 224;
    float part = 0.f;

    if constexpr (PREC == 0) {
      // ---- L0: 3 -> 200 on the VALU, latent folded into the bias -----------------------------
      f32x16 H[7];
#pragma unroll
      for (int b = 0; b < 7; ++b) {
        const f32x16 bias = load_frag16(b0 + (b * 2 + h) * 16);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          if (b == 6 && r >= 4) { H[b][r] = 0.f; continue; }
          const f32x4 w = l0w[(b * 16 + r) * 2 + h];
          H[b][r] = softplus100(fmaf(w[0], cx, fmaf(w[1], cy, fmaf(w[2], cz, bias[r]))));
        }
        pin16(H[b]);
        __builtin_amdgcn_sched_barrier(0);
      }
      // ---- L1: 200 -> 101 (4 row blocks) ----------------------------------------------------
      f32x16 G[4];
#pragma unroll
      for (int ob = 0; ob < L1_OB; ++ob) {
        const char* buf = ws.acquire(ob);
        f32x16 d = load_frag16(WS::tail_of(buf) + h * 16);
        d = gemm_block_f32<L1_KS, 6, 7>(buf, d, H, lane);
        G[ob] = softplus100_v(d);
      }
      // skip connection: features 101..103 of lin2's input are the local coords
      // (block 3, regs 1..3 of the upper half-wave); 1/sqrt(2) lives in the packed weights
      G[3][1] = h ? cx : G[3][1];
      G[3][2] = h ? cy : G[3][2];
      G[3][3] = h ? cz : G[3][3];
      // ---- L2: 104 -> 200 (7 row blocks), accumulator init carries the folded latent -----------
#pragma unroll
      for (int ob = 0; ob < L2_OB; ++ob) {
        const char* buf = ws.acquire(L1_OB + ob);
        f32x16 d = load_frag16(WS::tail_of(buf) + h * 16);
        d = gemm_block_f32<L2_KS, 3, 4>(buf, d, G, lane);
        H[ob] = softplus100_v(d);
      }
      // ---- L3: 200 -> 200, L4 (200 -> 1) fused into the epilogue -------------------------------
#pragma unroll
      for (int ob = 0; ob < L3_OB; ++ob) {
        const char* buf = ws.acquire(L1_OB + L2_OB + ob);
        f32x16 d = load_frag16(WS::tail_of(buf) + h * 16);
        d = gemm_block_f32<L3_KS, 6, 7>(buf, d, H, lane);
        // read the lin4 fragment AFTER the GEMM (hoisted above it, it only gets spilled)
        unsigned int w4a = (unsigned int)(size_t)(__attribute__((address_space(3))) const float*)(WS::tail_of(buf) + 32 + h * 16);
        asm volatile("" : "+v"(w4a) : "v"(d[15]));
        const f32x16 w4 = load_frag16_lds(w4a);
#pragma unroll
        for (int r = 0; r < 16; ++r) part = fmaf(softplus100(d[r]), w4[r], part);
        asm volatile("" : "+v"(part));      // finish this block's epilogue here (see pin16)
      }
    } else {
      // ================= split-bf16 path: same dataflow, operands as bf16 hi/lo pairs ==========
      ActB H[7];
#pragma unroll
      for (int b = 0; b < 7; ++b) {
        const f32x16 bias = load_frag16(b0 + (b * 2 + h) * 16);
        f32x16 v;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          if (b == 6 && r >= 4) { v[r] = 0.f; continue; }
          const f32x4 w = l0w[(b * 16 + r) * 2 + h];
          v[r] = softplus100(fmaf(w[0], cx, fmaf(w[1], cy, fmaf(w[2], cz, bias[r]))));
        }
        split_block(v, H[b]);
        __builtin_amdgcn_sched_barrier(0);
      }
      ActB G[4];
#pragma unroll
      for (int ob = 0; ob < L1_OB; ++ob) {
        const char* buf = ws.acquire(ob);
        f32x16 d = load_frag16(WS::tail_of(buf) + h * 16);
        d = gemm_block_bf16<L1_KS16, 6, 7>(buf, d, H, lane);
        f32x16 v;
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = softplus100(d[r]);
        if (ob == 3) {                       // skip connection: coords into features 101..103
          v[1] = h ? cx : v[1];
          v[2] = h ? cy : v[2];
          v[3] = h ? cz : v[3];
        }
        split_block(v, G[ob]);
      }
#pragma unroll
      for (int ob = 0; ob < L2_OB; ++ob) {
        const char* buf = ws.acquire(L1_OB + ob);
        f32x16 d = load_frag16(WS::tail_of(buf) + h * 16);
        d = gemm_block_bf16<L2_KS16, 3, 4>(buf, d, G, lane);
        f32x16 v;
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = softplus100(d[r]);
        split_block(v, H[ob]);
      }
#pragma unroll
      for (int ob = 0; ob < L3_OB; ++ob) {
        const char* buf = ws.acquire(L1_OB + L2_OB + ob);
        f32x16 d = load_frag16(WS::tail_of(buf) + h * 16);
        d = gemm_block_bf16<L3_KS16, 6, 7>(buf, d, H, lane);
        unsigned int w4a = (unsigned int)(size_t)(__attribute__((address_space(3))) const float*)(WS::tail_of(buf) + 32 + h * 16);
        asm volatile("" : "+v"(w4a) : "v"(d[15]));
        const f32x16 w4 = load_frag16_lds(w4a);
#pragma unroll
        for (int r = 0; r < 16; ++r) part = fmaf(softplus100(d[r]), w4[r], part);
        asm volatile("" : "+v"(part));
      }
    }
    const float f = part + __shfl_xor(part, 32) + setp[OFF_L4B];

    // ---- Gaussian blend (EnsembledDeepSDF.py:144-149) ------------------------------------------
    acc = fmaf(wk / denom, f, acc);
    ws.next_member();
  }

  // eval-mode overwrite (EnsembledDeepSDF.py:260-261): every member predicts 1 for this point
  if (hack) acc = S / denom;
  if (valid && h == 0) p.out[out_idx] = acc;
}

}  // namespace nphm

// ============================================================================================
// C ABI (include/nphm_amd.h)
// ============================================================================================
#include "../../include/nphm_amd.h"

#include <stdio.h>
#include <string.h>

static thread_local char g_err[512] = "";

static int fail(const char* what, hipError_t e) {
  snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
  return -1;
}
static int fail_msg(const char* what) {
  snprintf(g_err, sizeof(g_err), "%s", what);
  return -2;
}

extern "C" {

int nphm_abi_version(void) { return NPHM_AMD_ABI_VERSION; }
const char* nphm_last_error(void) { return g_err; }

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
  if (!packed) return fail_msg("nphm_identity_pack: null packed buffer");
  nphm::PackArgs a;
  for (int i = 0; i < 5; ++i) {
    if (!lin_weight[i] || !lin_bias[i]) return fail_msg("nphm_identity_pack: null weight/bias pointer");
    a.w[i] = lin_weight[i];
    a.b[i] = lin_bias[i];
  }
  a.out_f32 = static_cast<float*>(packed);
  a.out_bf16 = reinterpret_cast<uint16_t*>(static_cast<char*>(packed) + nphm::PACKED_F32_FLOATS * 4);
  hipStream_t st = static_cast<hipStream_t>(stream);
  dim3 g1((nphm::SET_STRIDE + 255) / 256, nphm::N_SETS);
  hipLaunchKernelGGL(nphm::pack_f32_kernel, g1, dim3(256), 0, st, a);
  dim3 g2((nphm::BF_SET_STRIDE + 255) / 256, nphm::N_SETS);
  hipLaunchKernelGGL(nphm::pack_bf16_kernel, g2, dim3(256), 0, st, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail("nphm_identity_pack launch", e);
  return 0;
}

int nphm_identity_prepare_latent(const void* packed,
                                 const float* const lin_weight[5], const float* const lin_bias[5],
                                 const float* const mlp_pos_weight[3], const float* const mlp_pos_bias[3],
                                 int pos_mlp_dim, const float* anchors_mean,
                                 const float* lat_rows, int n_rows,
                                 void* latent_state, float* anchors_out, void* stream) {
  (void)packed;
  if (n_rows <= 0) return fail_msg("nphm_identity_prepare_latent: n_rows must be > 0");
  if (pos_mlp_dim <= 0 || pos_mlp_dim > 256)
    return fail_msg("nphm_identity_prepare_latent: pos_mlp_dim must be in 1..256");
  if (!lat_rows || !latent_state || !anchors_mean)
    return fail_msg("nphm_identity_prepare_latent: null pointer");
  nphm::PrepArgs a;
  for (int i = 0; i < 5; ++i) { a.w[i] = lin_weight[i]; a.b[i] = lin_bias[i]; }
  for (int i = 0; i < 3; ++i) { a.pw[i] = mlp_pos_weight[i]; a.pb[i] = mlp_pos_bias[i]; }
  a.pos_dim = pos_mlp_dim;
  a.anchors_mean = anchors_mean;
  a.lat_rows = lat_rows;
  a.state = static_cast<float*>(latent_state);
  a.anchors_out = anchors_out;
  hipLaunchKernelGGL(nphm::prepare_latent_kernel, dim3(nphm::N_MEMBERS + 1, n_rows), dim3(256), 0,
                     static_cast<hipStream_t>(stream), a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail("nphm_identity_prepare_latent launch", e);
  return 0;
}

static int check_prec(int precision) {
  if (precision != NPHM_PREC_F32 && precision != NPHM_PREC_BF16X3)
    return fail_msg("nphm_identity_eval: unsupported precision mode");
  return 0;
}

int nphm_identity_eval_points(const void* packed, const void* latent_state,
                              const float* xyz, int n_rows, int64_t n_points,
                              int64_t hack_chunk, float prune_tol, int precision,
                              float* sdf_out, unsigned long long* stats, void* stream) {
  if (!packed || !latent_state || !xyz || !sdf_out) return fail_msg("nphm_identity_eval_points: null pointer");
  if (n_rows <= 0 || n_points <= 0) return fail_msg("nphm_identity_eval_points: empty input");
  if (check_prec(precision)) return -2;
  nphm::EvalArgs a;
  memset(&a, 0, sizeof(a));
  a.packed_f32 = static_cast<const float*>(packed);
  a.packed_bf16 = reinterpret_cast<const uint16_t*>(static_cast<const char*>(packed) + nphm::PACKED_F32_FLOATS * 4);
  a.state = static_cast<const float*>(latent_state);
  a.out = sdf_out;
  a.stats = stats;
  a.prune_tol = prune_tol;
  a.xyz = xyz;
  a.n_points = n_points;
  a.hack_chunk = hack_chunk;
  const int64_t tiles = (n_points + 32 * nphm::NW - 1) / (32 * nphm::NW);
  if (tiles > 0x7fffffffLL) return fail_msg("nphm_identity_eval_points: too many points");
  if (precision == NPHM_PREC_F32)
    hipLaunchKernelGGL((nphm::eval_kernel<0, 0>), dim3((unsigned)tiles, n_rows), dim3(64 * nphm::NW), 0,
                       static_cast<hipStream_t>(stream), a);
  else
    hipLaunchKernelGGL((nphm::eval_kernel<0, 1>), dim3((unsigned)tiles, n_rows), dim3(64 * nphm::NW), 0,
                       static_cast<hipStream_t>(stream), a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail("nphm_identity_eval_points launch", e);
  return 0;
}

int nphm_identity_eval_grid(const void* packed, const void* latent_state,
                            const float* axis_x, const float* axis_y, const float* axis_z,
                            int rx, int ry, int rz, int ix0, int ix1,
                            int64_t hack_chunk, float prune_tol, int precision,
                            float* sdf_out, unsigned long long* stats, void* stream) {
  if (!packed || !latent_state || !axis_x || !axis_y || !axis_z || !sdf_out)
    return fail_msg("nphm_identity_eval_grid: null pointer");
  if (rx <= 0 || ry <= 0 || rz <= 0 || ix0 < 0 || ix1 > rx || ix0 >= ix1)
    return fail_msg("nphm_identity_eval_grid: bad grid / slab bounds");
  if (check_prec(precision)) return -2;
  nphm::EvalArgs a;
  memset(&a, 0, sizeof(a));
  a.packed_f32 = static_cast<const float*>(packed);
  a.packed_bf16 = reinterpret_cast<const uint16_t*>(static_cast<const char*>(packed) + nphm::PACKED_F32_FLOATS * 4);
  a.state = static_cast<const float*>(latent_state);
  a.out = sdf_out;
  a.stats = stats;
  a.prune_tol = prune_tol;
  a.ax = axis_x; a.ay = axis_y; a.az = axis_z;
  a.rx = rx; a.ry = ry; a.rz = rz; a.ix0 = ix0; a.ix1 = ix1;
  a.nbx = (ix1 - ix0 + nphm::BRX - 1) / nphm::BRX; a.nby = (ry + nphm::BRY - 1) / nphm::BRY;
  a.nbz = (rz + nphm::BRZ - 1) / nphm::BRZ;
  a.nsx = (a.nbx + nphm::SBX - 1) / nphm::SBX; a.nsy = (a.nby + nphm::SBY - 1) / nphm::SBY;
  a.nsz = (a.nbz + nphm::SBZ - 1) / nphm::SBZ;
  a.hack_chunk = hack_chunk;
  const int64_t supers = (int64_t(a.nsx) * a.nsy * a.nsz + 7) / 8 * 8;       // padded to the 8 XCDs
  const int64_t bricks = supers * (nphm::SBX * nphm::SBY * nphm::SBZ);
  if (bricks > 0x7fffffffLL) return fail_msg("nphm_identity_eval_grid: slab too large for one launch");
  if (precision == NPHM_PREC_F32)
    hipLaunchKernelGGL((nphm::eval_kernel<1, 0>), dim3((unsigned)bricks), dim3(64 * nphm::NW), 0,
                       static_cast<hipStream_t>(stream), a);
  else
    hipLaunchKernelGGL((nphm::eval_kernel<1, 1>), dim3((unsigned)bricks), dim3(64 * nphm::NW), 0,
                       static_cast<hipStream_t>(stream), a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail("nphm_identity_eval_grid launch", e);
  return 0;
}

}  // extern "C"

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
  } else if (e < OFF_L1B) {
    const int x = e - OFF_L1A;
    const int c = x & 3, lane = (x >> 2) & 63, gg = x >> 8;
    const int g = gg % (L1_KS / 4), ob = gg / (L1_KS / 4);
    const int ks = 4 * g + c;
    val = layer_weight(a, 1, s, 32 * ob + (lane & 31), feat_of(ks_block(ks, 6), ks_reg(ks, 6), lane >> 5));
  } else if (e < OFF_L2A) {
    const int x = e - OFF_L1B;
    const int row = feat_of(x >> 5, x & 15, (x >> 4) & 1);
    if (row < L1_OUT) val = a.b[1][s * L1_OUT + row];
  } else if (e < OFF_L3A) {
    const int x = e - OFF_L2A;
    const int c = x & 3, lane = (x >> 2) & 63, gg = x >> 8;
    const int g = gg % (L2_KS / 4), ob = gg / (L2_KS / 4);
    const int ks = 4 * g + c;
    val = layer_weight(a, 2, s, 32 * ob + (lane & 31), feat_of(ks_block(ks, 3), ks_reg(ks, 3), lane >> 5));
  } else if (e < OFF_L3B) {
    const int x = e - OFF_L3A;
    const int c = x & 3, lane = (x >> 2) & 63, gg = x >> 8;
    const int g = gg % (L3_KS / 4), ob = gg / (L3_KS / 4);
    const int ks = 4 * g + c;
    val = layer_weight(a, 3, s, 32 * ob + (lane & 31), feat_of(ks_block(ks, 6), ks_reg(ks, 6), lane >> 5));
  } else if (e < OFF_L4W) {
    const int x = e - OFF_L3B;
    const int row = feat_of(x >> 5, x & 15, (x >> 4) & 1);
    if (row < HID) val = a.b[3][s * HID + row];
  } else if (e < OFF_L4B) {
    const int x = e - OFF_L4W;
    const int row = feat_of(x >> 5, x & 15, (x >> 4) & 1);
    if (row < HID) val = a.w[4][s * HID + row];
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
      st[LS_OFF_B2 + k * 224 + t] = v2;
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
  int nbx, nby, nbz;        // bricks per axis (4,4,8 voxels)
  int64_t hack_chunk;
};

// nn.Softplus(beta=100, threshold=20) (EnsembledDeepSDF.py:99): for 100x > 20 PyTorch returns x;
// here log(1 + exp(-|100x|)) is already 0 in fp32 for |100x| > 16.7, so the two agree to < 1e-9.
__device__ __forceinline__ float softplus100(float x) {
  const float t = __expf(-100.f * fabsf(x));
  return fmaxf(x, 0.f) + 0.01f * __logf(1.f + t);
}

__device__ __forceinline__ f32x16 softplus100_v(f32x16 d) {
  f32x16 o;
#pragma unroll
  for (int r = 0; r < 16; ++r) o[r] = softplus100(d[r]);
  return o;
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

// One GEMM layer on fp32 MFMA: D[ob] = bias + sum_ks A(ob,ks) x IN(block(ks))[reg(ks)].
// NKS K-steps, FULL full input blocks.  Afrag: [ob][ks/4][lane][4].
template <int NKS, int FULL, int NIN>
__device__ __forceinline__ f32x16 gemm_block_f32(const float* __restrict__ afrag_ob, f32x16 acc,
                                                 const f32x16 (&in)[NIN], int lane) {
  const f32x4* A = reinterpret_cast<const f32x4*>(afrag_ob) + lane;
#pragma unroll
  for (int g = 0; g < NKS / 4; ++g) {
    const f32x4 a = A[g * 64];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      constexpr int dummy = 0; (void)dummy;
      const int ks = 4 * g + c;
      const int b = ks < 16 * FULL ? (ks >> 4) : FULL;
      const int r = ks < 16 * FULL ? (ks & 15) : ks - 16 * FULL;
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c], in[b][r], acc, 0, 0, 0);
    }
  }
  return acc;
}

template <int MODE, int PREC>
__global__ __launch_bounds__(256, 2) void eval_kernel(EvalArgs p) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int h = lane >> 5;
  const int j = lane & 31;

  // ---- locate this lane's query point ------------------------------------------------------
  bool valid;
  int64_t out_idx;
  bool hack = false;
  float qx, qy, qz;
  int row = 0;
  if (MODE == 0) {
    row = blockIdx.y;
    const int64_t i = (int64_t(blockIdx.x) * 4 + wave) * 32 + j;
    valid = i < p.n_points;
    const int64_t ic = valid ? i : (p.n_points - 1);
    const float* q = p.xyz + (int64_t(row) * p.n_points + ic) * 3;
    qx = q[0]; qy = q[1]; qz = q[2];
    out_idx = int64_t(row) * p.n_points + ic;
    if (p.hack_chunk > 0) hack = ((ic + 1) % p.hack_chunk == 0) || (ic == p.n_points - 1);
  } else {
    // brick id -> (bx, by, bz), z fastest; 4x4x8 voxels per brick, wave owns 4x4x2
    int bid = blockIdx.x;
    const int bz = bid % p.nbz; bid /= p.nbz;
    const int by = bid % p.nby; bid /= p.nby;
    const int bx = bid;
    const int ix = p.ix0 + bx * 4 + (j >> 3);
    const int iy = by * 4 + ((j >> 1) & 3);
    const int iz = bz * 8 + wave * 2 + (j & 1);
    valid = ix < p.ix1 && iy < p.ry && iz < p.rz;
    const int cx_ = min(ix, p.ix1 - 1), cy_ = min(iy, p.ry - 1), cz_ = min(iz, p.rz - 1);
    qx = p.ax[cx_]; qy = p.ay[cy_]; qz = p.az[cz_];
    const int64_t gi = (int64_t(cx_) * p.ry + cy_) * p.rz + cz_;
    out_idx = gi - int64_t(p.ix0) * p.ry * p.rz;
    if (p.hack_chunk > 0)
      hack = ((gi + 1) % p.hack_chunk == 0) || (gi == int64_t(p.rx) * p.ry * p.rz - 1);
  }
  if (__ballot(valid) == 0ull) return;

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
  uint64_t wmask = 0;
  if (p.prune_tol < 0.f) {
    wmask = (1ull << N_MEMBERS) - 1;
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

  if (p.stats && lane == 0) {
    const unsigned long long nv = __popcll(__ballot(valid)) >> 1;   // both half-waves hold the same points
    atomicAdd(p.stats, nv * __popcll(wmask));
    atomicAdd(p.stats + 1, nv);
  }

  float acc = 0.f;

#pragma unroll 1
  for (int k = 0; k < N_MEMBERS; ++k) {
    if (!((wmask >> k) & 1ull)) continue;
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

    // ---- L0: 3 -> 200 on the VALU, latent folded into the bias -------------------------------
    f32x16 H[7];
    {
      const f32x4* l0w = reinterpret_cast<const f32x4*>(setp + OFF_L0W);
      const float* b0 = st + LS_OFF_B0 + k * 224;
#pragma unroll
      for (int b = 0; b < 7; ++b) {
        const f32x16 bias = load_frag16(b0 + (b * 2 + h) * 16);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          if (b == 6 && r >= 4) { H[b][r] = 0.f; continue; }
          const f32x4 w = l0w[(b * 16 + r) * 2 + h];
          H[b][r] = softplus100(fmaf(w[0], cx, fmaf(w[1], cy, fmaf(w[2], cz, bias[r]))));
        }
      }
    }

    // ---- L1: 200 -> 101 (4 row blocks) ------------------------------------------------------
    f32x16 G[4];
#pragma unroll
    for (int ob = 0; ob < L1_OB; ++ob) {
      f32x16 d = load_frag16(setp + OFF_L1B + (ob * 2 + h) * 16);
      d = gemm_block_f32<L1_KS, 6, 7>(setp + OFF_L1A + ob * (L1_KS / 4) * 256, d, H, lane);
      G[ob] = softplus100_v(d);
    }
    // skip connection: features 101..103 of lin2's input are the local coords
    // (block 3, regs 1..3 of the upper half-wave); 1/sqrt(2) lives in the packed weights
    G[3][1] = h ? cx : G[3][1];
    G[3][2] = h ? cy : G[3][2];
    G[3][3] = h ? cz : G[3][3];

    // ---- L2: 104 -> 200 (7 row blocks), bias carries the folded latent ------------------------
    {
      const float* b2 = st + LS_OFF_B2 + k * 224;
#pragma unroll
      for (int ob = 0; ob < L2_OB; ++ob) {
        f32x16 d = load_frag16(b2 + (ob * 2 + h) * 16);
        d = gemm_block_f32<L2_KS, 3, 4>(setp + OFF_L2A + ob * (L2_KS / 4) * 256, d, G, lane);
        H[ob] = softplus100_v(d);
      }
    }

    // ---- L3: 200 -> 200, L4 (200 -> 1) fused into the epilogue ---------------------------------
    float part = 0.f;
#pragma unroll
    for (int ob = 0; ob < L3_OB; ++ob) {
      f32x16 d = load_frag16(setp + OFF_L3B + (ob * 2 + h) * 16);
      d = gemm_block_f32<L3_KS, 6, 7>(setp + OFF_L3A + ob * (L3_KS / 4) * 256, d, H, lane);
      const f32x16 w4 = load_frag16(setp + OFF_L4W + (ob * 2 + h) * 16);
#pragma unroll
      for (int r = 0; r < 16; ++r) part = fmaf(softplus100(d[r]), w4[r], part);
    }
    const float f = part + __shfl_xor(part, 32) + setp[OFF_L4B];

    // ---- Gaussian blend (EnsembledDeepSDF.py:144-149) ------------------------------------------
    acc = fmaf(wk / denom, f, acc);
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
  if (precision != NPHM_PREC_F32) return fail_msg("nphm_identity_eval: unsupported precision mode");
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
  const int64_t tiles = (n_points + 127) / 128;
  if (tiles > 0x7fffffffLL) return fail_msg("nphm_identity_eval_points: too many points");
  hipLaunchKernelGGL((nphm::eval_kernel<0, 0>), dim3((unsigned)tiles, n_rows), dim3(256), 0,
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
  a.nbx = (ix1 - ix0 + 3) / 4; a.nby = (ry + 3) / 4; a.nbz = (rz + 7) / 8;
  a.hack_chunk = hack_chunk;
  const int64_t bricks = int64_t(a.nbx) * a.nby * a.nbz;
  if (bricks > 0x7fffffffLL) return fail_msg("nphm_identity_eval_grid: slab too large for one launch");
  hipLaunchKernelGGL((nphm::eval_kernel<1, 0>), dim3((unsigned)bricks), dim3(256), 0,
                     static_cast<hipStream_t>(stream), a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail("nphm_identity_eval_grid launch", e);
  return 0;
}

}  // extern "C"

// mlp_kernel.hip — fused dense skip-MLP for the reference's DeepSDF (src/NPHM/models/deepSDF.py:6-89):
// the NPM global SDF and the backbone of the forward-deformation network
// (DeformationNetwork, deepSDF.py:118-239; chunked drivers get_logits_backward / deform_mesh,
// src/NPHM/models/reconstruction.py:28-88).  One latent per batch row (mlp_layout.h).
//
// Structure (numbers in DESIGN.md):
//   * one workgroup = 8 wavefronts = M points (64 at hidden <= 512, 32 at hidden <= 1024; twice that in the variant
//     without a lo plane, ONE) for the WHOLE network; activations live in LDS as split 16-bit (hi | lo) K chunks, 128 KiB;
//   * a layer is an output-stationary GEMM: wavefront w owns the 32-row output tiles w, w+8, ...
//     for all M points (accumulators in registers), its A fragments (weights) stream L2 -> VGPR
//     with no reuse inside the workgroup (every weight byte is fetched once per M points), B
//     fragments (activations) come from LDS with conflict-free 16-byte reads;
//   * x*w ~= xh*wh + xl*wh + xh*wl on v_mfma_f32_32x32x16_{f16,bf16} (fp32 accumulate), per layer also xh*wh + xl*wh or
//     xh*wh alone (calibrated per checkpoint by the host module);
//   * the accumulators are initialised by one extra "coordinate K-step" per tile that carries the
//     bias, the folded latent and - for lin0 and the skip layer - the xyz columns;
//   * epilogue: base-2 softplus, re-split to hi/lo INTO the accumulators' registers; the LDS tile is overwritten
//     in place between two workgroup barriers;
//   * the last linear layer (out <= 4) is applied in fp32 by the epilogue of the last hidden layer, from registers;
//     the wavefronts' sums meet in LDS and are added in wavefront order (bitwise reproducible).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>

#include "capi_common.h"
#include "mlp_layout.h"

namespace nphm {
namespace mlp {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
// one 16-byte MFMA operand fragment: 8 bf16 or 8 binary16 values (the kernels are templates on the format, F16)
typedef unsigned frag_t __attribute__((ext_vector_type(4)));
template <bool F16>
__device__ __forceinline__ f32x16 mfma16(const frag_t& a, const frag_t& b, const f32x16& c) {
  if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

__device__ inline uint16_t f32_to_bf16_rn(float x) {
  uint32_t u = __float_as_uint(x);
  uint32_t r = u + 0x7fffu + ((u >> 16) & 1u);
  return uint16_t(r >> 16);
}
__device__ inline float bf16_to_f32(uint16_t v) { return __uint_as_float(uint32_t(v) << 16); }
// IEEE binary16, round to nearest even (subnormals kept: v_mfma_f32_32x32x16_f16 does not flush them, tools/micro/f16split.hip)
__device__ inline uint16_t f32_to_f16_rn(float x) { return __builtin_bit_cast(uint16_t, (_Float16)x); }
__device__ inline float f16_to_f32(uint16_t v) { return (float)__builtin_bit_cast(_Float16, v); }
// the two formats of a split operand behind one name: hi = round(x), lo = round(x - hi)
template <bool F16> __device__ inline uint16_t half_rn(float x) { return F16 ? f32_to_f16_rn(x) : f32_to_bf16_rn(x); }
template <bool F16> __device__ inline float half_f32(uint16_t v) { return F16 ? f16_to_f32(v) : bf16_to_f32(v); }

// ---------------------------------------------------------------------------------------------
// pack: nn.Linear weights -> split-bf16 MFMA A fragments (once per weight update)
// ---------------------------------------------------------------------------------------------
struct PtrTable {
  const float* w[MAX_LINEAR];
  const float* b[MAX_LINEAR];
};

struct PackArgs {
  PtrTable t;
  Plan plan;
  uint16_t* out;
};

__global__ void mlp_pack_kernel(PackArgs a) {
  if (blockIdx.y == a.plan.n_linear - 1) {            // extra grid row: the fp32 table of the last layer (mlp_layout.h)
    const Layer& LL = a.plan.layer[a.plan.n_linear - 1];
    float* tab = reinterpret_cast<float*>(reinterpret_cast<char*>(a.out) + 2 * a.plan.packed_bytes);
    const int kt = LL.k_steps / 2;
    const size_t total = last_table_floats(kt);
    for (size_t e = size_t(blockIdx.x) * blockDim.x + threadIdx.x; e < total; e += size_t(gridDim.x) * blockDim.x) {
      float v = 0.f;
      if (e >= total - 4) {
        const int c = int(e - (total - 4));
        if (c < LL.out_dim) v = a.t.b[a.plan.n_linear - 1][c];
      } else {
        const int r = e & 15, c = (e >> 4) & 3, hh = (e >> 6) & 1, n = int(e >> 7);
        const int f = 32 * n + feat_local(r, hh);
        if (c < LL.out_dim && f < LL.k_act) v = a.t.w[a.plan.n_linear - 1][size_t(c) * LL.in_dim + f];
      }
      tab[e] = v;
    }
    return;
  }
  const int l = blockIdx.y + 1;                       // layer 0 has no activation columns
  const Layer& L = a.plan.layer[l];
  const size_t total = size_t(L.n_tiles) * L.k_steps * 2 * 64 * 8;
  for (size_t e = size_t(blockIdx.x) * blockDim.x + threadIdx.x; e < total;
       e += size_t(gridDim.x) * blockDim.x) {
    const int i = e & 7, lane = (e >> 3) & 63, part = (e >> 9) & 1;
    const size_t gg = e >> 10;
    const int ks = int(gg % L.k_steps), n = int(gg / L.k_steps);
    const int row = 32 * n + (lane & 31);
    const int f = 32 * (ks >> 1) + feat_local(8 * (ks & 1) + i, lane >> 5);
    float w = 0.f;
    // The LAST layer's 1 / k (mlp_layout.h: its act_scale) is applied to the fp32 result by the evaluation kernel, not folded
    // into the fragments: w / k ~ 1e-4 has an f16 lo half below 1e-7, where binary16 subnormals are 6e-8 apart - the folded
    // form left the split-f16 path with 12-bit last-layer weights (2.4e-6 on the deformation fixture against 2e-7)
    const float wscale = l == a.plan.n_linear - 1 ? 1.f : L.act_scale;
    if (row < L.out_dim && f < L.k_act) w = a.t.w[l][size_t(row) * L.in_dim + f] * wscale;
    const uint16_t hi = f32_to_bf16_rn(w);
    const uint16_t lo = f32_to_bf16_rn(w - bf16_to_f32(hi));
    a.out[L.w_off / 2 + e] = part ? lo : hi;
    // the split-f16 fragments: same position in the second half of the packed buffer
    const uint16_t hh = f32_to_f16_rn(w);
    const uint16_t hl = f32_to_f16_rn(w - f16_to_f32(hh));
    a.out[(a.plan.packed_bytes + L.w_off) / 2 + e] = part ? hl : hh;
  }
}

// ---------------------------------------------------------------------------------------------
// prepare: per latent row, the coordinate K-step fragments of every layer (folded latent + bias)
// (grid and block sizes: PREP_SPLIT / PREP_THREADS below)
// ---------------------------------------------------------------------------------------------
struct PrepArgs {
  PtrTable t;
  Plan plan;
  const float* cond;   // [n_rows, lat_dim]
  int lat_dim;
  char* state;         // [n_rows, state_row_bytes]
};

// grid (n_linear, n_rows, PREP_SPLIT): block z of a layer owns the 32-row tiles z, z + PREP_SPLIT, ...
constexpr int PREP_SPLIT = 16, PREP_THREADS = 512;
__global__ __launch_bounds__(PREP_THREADS) void mlp_prepare_kernel(PrepArgs a) {
  __shared__ float bias[32];
  __shared__ float lat[1024];
  const int l = blockIdx.x, row = blockIdx.y, t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  constexpr int N_WAVES = PREP_THREADS / 64, PER_WAVE = 32 / N_WAVES;      // 8 wavefronts x 4 outputs = one tile
  const Layer& L = a.plan.layer[l];
  const float* W = a.t.w[l];
  const float* cond = a.cond + size_t(row) * a.lat_dim;
  for (int j = t; j < a.lat_dim; j += blockDim.x) lat[j] = cond[j];
  uint16_t* out = reinterpret_cast<uint16_t*>(a.state + size_t(row) * 2 * a.plan.state_row_bytes + L.c_off);   // rows: [bf16 | f16]
  for (int n = blockIdx.z; n < L.n_tiles; n += gridDim.z) {
    __syncthreads();                       // lat is loaded / the previous tile's bias values have been consumed
    // folded biases of the tile's 32 outputs: a WAVEFRONT per output (lanes stride over the latent columns: coalesced
    // reads of the weight row, butterfly reduction), four outputs per wavefront with their loads in flight together -
    // a thread per output walked the rows with a stride of in_dim floats (42 us per launch, now a few)
    float v[PER_WAVE];
#pragma unroll
    for (int q = 0; q < PER_WAVE; ++q) {
      const int o = 32 * n + wave * PER_WAVE + q;
      v[q] = 0.f;
      if (o < L.out_dim && L.lat_col >= 0) {
        const float* wl = W + size_t(o) * L.in_dim + L.lat_col;
        for (int j = lane; j < a.lat_dim; j += 64) v[q] = fmaf(wl[j], lat[j], v[q]);
      }
    }
#pragma unroll
    for (int q = 0; q < PER_WAVE; ++q) {
      const int o = 32 * n + wave * PER_WAVE + q;
#pragma unroll
      for (int sft = 32; sft > 0; sft >>= 1) v[q] += __shfl_xor(v[q], sft);
      // (the last layer's accumulators live in the scaled domain too: its bias carries k, the kernel divides the sum by k)
      const float add_scale = l == a.plan.n_linear - 1 ? SP_SCALE : L.add_scale;
      if (lane == 0) bias[wave * PER_WAVE + q] = o < L.out_dim ? (v[q] * L.in_scale + a.t.b[l][o]) * add_scale : 0.f;
    }
    __syncthreads();
    const int e = t;                        // 64 lanes x 8 entries of the tile's coordinate-step fragment
    const int i = e & 7, flane = (e >> 3) & 63;
    const int o = 32 * n + (flane & 31), hh = flane >> 5;
    auto entry = [&](auto F) -> uint16_t {
      constexpr bool F16 = decltype(F)::value;
      if (o >= L.out_dim) return uint16_t(0);
      const float b = bias[flane & 31];
      const uint16_t bh = half_rn<F16>(b);
      const float r1 = b - half_f32<F16>(bh);
      const uint16_t bm = half_rn<F16>(r1);
      const uint16_t bl = half_rn<F16>(r1 - half_f32<F16>(bm));
      auto wc = [&](int c) -> float {
        return L.coord_col < 0 ? 0.f : W[size_t(o) * L.in_dim + L.coord_col + c] * L.in_scale * L.add_scale;
      };
      auto whi = [&](int c) { return half_rn<F16>(wc(c)); };
      auto wlo = [&](int c) { const float w = wc(c); return half_rn<F16>(w - half_f32<F16>(half_rn<F16>(w))); };
      if (hh == 0) return i < 3 ? whi(i) : i < 6 ? whi(i - 3) : i == 6 ? bh : bm;
      return i < 3 ? wlo(i) : i == 3 ? bl : i < 7 ? whi(i - 4) : uint16_t(0);
    };
    out[size_t(n) * 512 + e] = entry(std::false_type{});
    // split-f16 fragments of the same K-step: second half of the state row
    out[a.plan.state_row_bytes / 2 + size_t(n) * 512 + e] = entry(std::true_type{});
  }
}

// ---------------------------------------------------------------------------------------------
// evaluation
// ---------------------------------------------------------------------------------------------
struct LayerDev {
  int n_tiles, k_steps;
  uint32_t w_off, c_off;
};

struct EvalArgs {
  const char* packed;
  const float* last_tab;   // fp32 table of the last linear layer (mlp_layout.h: last_table_floats), behind both fragment formats
  int last_tiles;          // its input tiles
  const char* state;
  size_t state_row_bytes;
  float* out;             // [n_rows, n_points, out_dim]
  float* jinv_out;        // value + Jacobian launches, or null: [n_rows, n_points, 3, 3] inverse of d (x + F)_i / d x_c (row i, column c)
  int out_dim;
  int add_input;          // out[..., c] += xyz[..., c] (c < 3): canonical / posed points
  int n_linear;
  unsigned two_pass_mask;  // bit l: hidden layer l runs the two-term product xh wh + xl wh (weights rounded to the half format)
  unsigned one_pass_mask;  // bit l: hidden layer l runs the single-term product xh wh (split-f16 only; xh = rn(x))
  LayerDev layer[MAX_LINEAR];
  // MODE 0
  const float* xyz;       // [n_rows, n_points, 3]
  int64_t n_points;
  int64_t point_base, point_end;   // MODE 0: the launch covers points [point_base, point_end) of every row (0, n_points: all)
  // KIND 2 (Broyden): xyz = initial iterates, out = final iterates [n_rows, n_points, 3]
  const float* obs;       // [n_rows, n_points, 3] observed (posed) points
  const float* jinv;      // [n_rows, n_points, 3, 3] initial inverse Jacobians
  const float* posed0;    // or null: x_init + F(x_init) of every point (stride posed0_stride floats), spares the first evaluation
  int64_t posed0_stride;
  float* diff_out;        // [n_rows, n_points] smallest residual norm seen
  unsigned char* valid_out;   // [n_rows, n_points] converged
  int max_steps;
  float cvg, dvg, eps;
  // MODE 1: x-slab [ix0, ix1) of an 'ij' lattice, points in flattened lattice order
  const float* ax; const float* ay; const float* az;
  int rx, ry, rz, ix0, ix1;
  // KIND 3: the forward that also leaves sigma'(d_l) of every hidden layer for mlp_bwd_kernel.hip
  float* sig_out;         // [n_rows][n_workgroups][sig_tiles][MT][64 lanes][16]: register dumps of the owning wavefront
  int sig_tiles;          // tiles per workgroup = sum of n_tiles over the hidden layers
  int sig_base[MAX_LINEAR];   // first tile of layer l
  // KIND 6 (the K-split form of TAIL2): slots of TAIL_SLOT_BYTES each where a workgroup parks the second K half of the last
  // hidden layer's operands, and one busy word per slot (zeroed by the launcher; compare-and-swap to take, store to release)
  char* tail_ws;
  unsigned* tail_flags;
  unsigned tail_slots;
};

// (a parked half = the workgroup's hi | lo planes = all of its activation LDS, whatever the variant)
constexpr size_t TAIL_SLOT_BYTES = size_t(128) << 10;
constexpr unsigned TAIL_SLOTS = 512;        // >= 2 x the workgroups resident at once (one per CU: the planes take 128 KiB of LDS)
constexpr size_t TAIL_FLAG_BYTES = 4096;

// k softplus(d / k) in base 2 (see mlp_layout.h); agrees with nn.Softplus(beta=100, threshold=20)
// to < 1e-9 in unscaled units
#ifndef NPHM_MLP_SOFTPLUS4
#define NPHM_MLP_SOFTPLUS4 1
#endif
__device__ __forceinline__ float softplus2(float d) {
#if NPHM_MLP_SOFTPLUS4
  // log2(1 + 2^d') + one v_med3 that returns d' once 2^d' has overflowed (eval_kernel.hip: 4 issue slots instead of 5)
  const float r = __builtin_amdgcn_logf(1.f + __builtin_amdgcn_exp2f(d));
  return __builtin_amdgcn_fmed3f(d, r, 127.f);
#else
  const float t = __builtin_amdgcn_exp2f(-fabsf(d));
  return fmaxf(d, 0.f) + __builtin_amdgcn_logf(1.f + t);   // one v_max_f32 under -fno-honor-nans
#endif
}

struct Split8 { frag_t hi, lo; };
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {
  const f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
// 8 values -> hi | lo operands.  bf16: per pair one packed convert, a shift / a mask back to fp32, the residuals through a
// second packed convert (the generic __bf16 casts made hipcc convert every value twice).  binary16: hi = rn(x) by
// v_cvt_pk_f16_f32, clamped to the largest finite value (beyond the f16 range it saturates at 65504 and the lo half carries the
// rest instead of hi becoming inf), lo = x - hi straight into its packed half by v_fma_mixlo / mixhi_f16.  (Rounds 1-4 took
// hi by round-toward-zero, v_cvt_pkrtz: one instruction less per pair, but a single-term layer - which reads hi alone - then
// sees every non-negative activation shrunk by 2^-12 on average, a coherent 1.2e-4 relative error; with hi to nearest the lo
// half is a bit smaller as well.)
template <bool F16>
__device__ __forceinline__ Split8 split8(const float* x) {
  Split8 o;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    if constexpr (F16) {
      const f32x2_t xv = {x[2 * q], x[2 * q + 1]};
      const f16x2_t top = {(_Float16)65504.f, (_Float16)65504.f};
      const unsigned ph = __builtin_bit_cast(unsigned, __builtin_elementwise_min(__builtin_convertvector(xv, f16x2_t), top));
      unsigned pl;
      asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(pl) : "v"(ph), "v"(x[2 * q]));
      asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(pl) : "v"(ph), "v"(x[2 * q + 1]));
      o.hi[q] = ph;
      o.lo[q] = pl;
    } else {
      const unsigned ph = cvt_pk_bf16(x[2 * q], x[2 * q + 1]);
      const float h0 = __builtin_bit_cast(float, ph << 16), h1 = __builtin_bit_cast(float, ph & 0xffff0000u);
      o.hi[q] = ph;
      o.lo[q] = cvt_pk_bf16(x[2 * q] - h0, x[2 * q + 1] - h1);
    }
  }
  return o;
}

// 8 values -> ONE binary16 operand (split8's hi half alone): the B operand of the all-single-term variant (ONE below)
__device__ __forceinline__ frag_t pack8_rn(const float* x) {
  frag_t o;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f32x2_t v = {x[2 * q], x[2 * q + 1]};
    f16x2_t hv = __builtin_convertvector(v, f16x2_t);
    const f16x2_t top = {(_Float16)65504.f, (_Float16)65504.f};
    hv = __builtin_elementwise_min(hv, top);
    o[q] = __builtin_bit_cast(unsigned, hv);
  }
  return o;
}

// B operand of the coordinate K-step for one point (see mlp_layout.h)
template <bool F16>
__device__ __forceinline__ frag_t coord_operand(float x, float y, float z, int h) {
  using half_t = typename std::conditional<F16, _Float16, __bf16>::type;
  typedef half_t half8 __attribute__((ext_vector_type(8)));
  const float cs[3] = {x, y, z};
  half_t xh[3], xl[3], xll[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    xh[i] = (half_t)cs[i];
    const float r1 = cs[i] - (float)xh[i];
    xl[i] = (half_t)r1;
    xll[i] = (half_t)(r1 - (float)xl[i]);
  }
  const half_t one = (half_t)1.f, zero = (half_t)0.f;
  half8 bv;
  bv[0] = xh[0]; bv[1] = xh[1]; bv[2] = xh[2];
  bv[3] = h ? one : xl[0];
  bv[4] = h ? xll[0] : xl[1];
  bv[5] = h ? xll[1] : xl[2];
  bv[6] = h ? xll[2] : one;
  bv[7] = h ? zero : one;
  return __builtin_bit_cast(frag_t, bv);
}

// d softplus / d d in the scaled domain: 1 / (1 + 2^-d)
__device__ __forceinline__ float sigmoid2(float d) {
  const float t = __builtin_amdgcn_exp2f(-fabsf(d));
  return (d >= 0.f ? 1.f : t) * __builtin_amdgcn_rcpf(1.f + t);
}

// B operand of the coordinate K-step for the tangent stream d/dx_c: the unit vector e_c in the xh
// slots (exact in either half format), zeros in the xl / xll / bias slots
template <bool F16>
__device__ __forceinline__ frag_t unit_operand(int c) {
  using half_t = typename std::conditional<F16, _Float16, __bf16>::type;
  typedef half_t half8 __attribute__((ext_vector_type(8)));
  const half_t one = (half_t)1.f, zero = (half_t)0.f;
  half8 bv;
#pragma unroll
  for (int i = 0; i < 8; ++i) bv[i] = zero;
  bv[0] = c == 0 ? one : zero;
  bv[1] = c == 1 ? one : zero;
  bv[2] = c == 2 ? one : zero;
  return __builtin_bit_cast(frag_t, bv);
}

// JVP = forward-mode derivative w.r.t. xyz carried along: the M columns of a workgroup are 4 streams
// (value, d/dx, d/dy, d/dz) of M/4 points; the tangent streams go through the SAME GEMMs (their
// coordinate K-step sees the unit vectors, no bias) and their epilogue multiplies by the value
// stream's sigmoid instead of applying softplus.  Output [n, 4, out_dim].
//
// KIND 2 = Broyden root finding x + F(x) = obs fused around the network (the reference's
// iterative_root_finding.broyden, one launch instead of <= 16 forwards + host syncs): the workgroup
// keeps its M points for up to max_steps + 1 evaluations, thread m < M owns point m's solver state
// (x, g, dx, dg, 3x3 inverse Jacobian, best residual) in registers, the iterate travels to the
// wavefronts through LDS; a workgroup leaves as soon as none of its points is active.
// F16: binary16 halves on v_mfma_f32_32x32x16_f16 (11-bit significands: the three-term product carries 22 bits against 16,
// the two-term product of the layers in two_pass_mask is 8x closer to fp32 than its bf16 form); every entry point takes the
// format in its `numerics` argument (tangent streams: k d a / d x stays far inside the binary16 range for fields with
// |d a / d x| < 400, and the hi half saturates instead of overflowing, see split8)
// ALL2: every hidden GEMM layer runs the two-term product (EvalArgs::two_pass_mask names them all - what the calibration finds
// for both lattice nets and for the fitting launches of the expression decoder): no wl fragment is ever requested, and the
// registers that would hold them hold two more K-steps of wh - four in flight instead of two.  (Fitting step, all six launch
// shapes: 1 116 -> 1 142 steps/s, same bits.)  (The 32-point workgroups of the hidden-1024 net stream a 16 MB pack out of the Infinity
// Cache: with 64 KB in flight per CU they ran at that latency's 74 GB/s per CU, a third of the matrix pipe.)
// Per-layer tiers of the split-f16 product (EvalArgs::two_pass_mask / one_pass_mask, calibrated per checkpoint by
// DeepSDF._numerics_code): three terms xh wh + xl wh + xh wl | two terms (weights rounded) | ONE term rn(x) wh.
// ONE (template): EVERY hidden GEMM layer single-term - no lo plane in LDS, so a workgroup holds TWICE the points (128 at
// hidden <= 512) per weight pass: half the L2 -> register weight bytes per point and half the MFMAs of the two-term product.
// (Round 5 also built a phase-shifted schedule - wavefronts 0..3 half a layer ahead of 4..7, LDS flags instead of the second
// barrier, so that one wavefront of a SIMD runs its epilogue while the other owns the matrix pipe: correct, and 3-5 % SLOWER
// on all three tiers; the kernel sits at the board's power limit, a denser MFMA stream is a lower clock.  profiles/NOTES.md.)
// TAIL2 (with ONE): the LAST hidden layer two-term, every other one single-term (what NPM's calibration asks for: the input of
// its last hidden layer needs the lo half).  There is no room for a lo plane beside the hi plane of 2 x the points - so the
// last hidden layer runs in two point halves: the operands of the first half (hi | lo) take the whole plane while the second
// half's wait in the registers of the D tiles they came from (they ARE those registers, see activate), then the other way
// round.  The layer's weights are streamed twice, every other layer's once per 2 x the points.
template <int N> using IC = std::integral_constant<int, N>;
template <int MT, int NTW, int MODE, int KIND, bool F16 = false, bool ALL2 = false, bool ONE = false>
__global__ __launch_bounds__(64 * WAVES, 2) void mlp_eval_kernel(EvalArgs p) {
  constexpr bool JVP = KIND == 1 || KIND == 4, BROY = KIND == 2, SAVE = KIND == 3 || KIND == 4;   // 4: value+Jacobian, sigma' saved
  constexpr bool TAIL2 = KIND == 5 || KIND == 6;   // plain evaluation (like 0), the last hidden layer two-term: in two point halves (5) ...
  constexpr bool TAILK = KIND == 6;         // ... or in two K halves, the second one parked in EvalArgs::tail_ws meanwhile
  static_assert(!ONE || (F16 && (KIND == 0 || TAIL2)), "the single-term product serves the plain split-f16 evaluation");
  static_assert(!TAIL2 || (ONE && MT % 2 == 0), "two point halves of the variant without a lo plane");
  constexpr int M = 32 * MT;               // columns per workgroup
  constexpr int PTS = JVP ? M / 4 : M;     // points per workgroup
  constexpr int HMAX = 32 * WAVES * NTW;   // widest layer
  constexpr int NCH = HMAX / 8;            // 16-byte K chunks per point
  constexpr int PART_BYTES = NCH * M * 16;
  static_assert(!TAILK || size_t(PART_BYTES) == TAIL_SLOT_BYTES, "a parked K half is the size of the activation plane");
  constexpr bool YREG = MT <= 2;           // the last layer's per-wavefront sums live in registers (else in LDS)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* act_hi = smem;
  char* act_lo = smem + PART_BYTES;                                   // (ONE: no lo plane; never addressed)
  float* partial = reinterpret_cast<float*>(smem + (ONE ? 1 : 2) * PART_BYTES);   // [WAVES][M][4]
  float* xs = partial + WAVES * M * 4;                                // [M][4] current iterate (KIND 2)
  float* wlast = xs + M * 4;                                          // fp32 table of the last linear layer (mlp_layout.h)

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int h = lane >> 5, j = lane & 31;
  const int row = blockIdx.y;
  const int64_t n_pts = MODE == 0 ? p.n_points : int64_t(p.ix1 - p.ix0) * p.ry * p.rz;
  const int64_t base = (MODE == 0 ? p.point_base : 0) + int64_t(blockIdx.x) * PTS;
  const int64_t n_end = MODE == 0 ? p.point_end : n_pts;          // first point of the row this launch does not own
  auto point_coords = [&](int64_t i, float& x, float& y, float& z) {
    const int64_t ic = i < n_pts ? i : n_pts - 1;
    if (MODE == 0) {
      const float* q = p.xyz + (int64_t(row) * n_pts + ic) * 3;
      x = q[0]; y = q[1]; z = q[2];
    } else {
      const int64_t plane = int64_t(p.ry) * p.rz;
      const int ix = p.ix0 + int(ic / plane);
      const int rem = int(ic % plane);
      x = p.ax[ix]; y = p.ay[rem / p.rz]; z = p.az[rem % p.rz];
    }
  };

  // the last layer's fp32 weights into LDS (first use: behind the barriers of the hidden layers)
  {
    const int n4 = p.last_tiles * 32 + 1;                             // float4 entries
    for (int e = threadIdx.x; e < n4; e += blockDim.x)
      reinterpret_cast<float4*>(wlast)[e] = reinterpret_cast<const float4*>(p.last_tab)[e];
  }

  // ---- KIND 6: a slot of the parking space (read behind the hidden layers' barriers) -------------------------------------
  if constexpr (TAILK) {
    if (threadIdx.x == 0) {
      unsigned sl = (blockIdx.x + blockIdx.y * gridDim.x) % p.tail_slots;
      while (atomicCAS(p.tail_flags + sl, 0u, 1u) != 0u) sl = sl + 1 == p.tail_slots ? 0u : sl + 1;
      reinterpret_cast<unsigned*>(xs)[0] = sl;
    }
  }

  // ---- KIND 2: solver state of point threadIdx.x (threads < M) -----------------------------------
  float bx[3] = {0, 0, 0}, bobs[3] = {0, 0, 0}, bgx[3] = {0, 0, 0}, bdx[3] = {0, 0, 0}, bdgx[3] = {0, 0, 0};
  float bJ[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, bbest = 0.f;
  bool bactive = false, bowner = false;
  if (BROY && threadIdx.x < M) {
    const int64_t i = base + threadIdx.x;
    bowner = i < n_end;
    const int64_t ic = (int64_t(row) * n_pts + (bowner ? i : n_pts - 1));
#pragma unroll
    for (int c = 0; c < 3; ++c) { bx[c] = p.xyz[ic * 3 + c]; bobs[c] = p.obs[ic * 3 + c]; }
#pragma unroll
    for (int c = 0; c < 9; ++c) bJ[c] = p.jinv[ic * 9 + c];
    xs[threadIdx.x * 4 + 0] = bx[0]; xs[threadIdx.x * 4 + 1] = bx[1]; xs[threadIdx.x * 4 + 2] = bx[2];
  }

#pragma unroll 1
  for (int it = 0;; ++it) {                 // one pass unless KIND 2
  // KIND 2 with the posed start points given (the caller's value+Jacobian launch at x_init produced them with the inverse
  // Jacobians): the residual of iteration 0 is read, not evaluated - one network pass less per solve
  const bool given0 = BROY && it == 0 && p.posed0 != nullptr;
  if (!given0) {
  frag_t bv[MT];
  // this wavefront's share of the last layer's sums (activate, `last`, adds its tiles' contributions in tile order): in
  // registers where there is room (MT <= 2), else in its LDS slot
  float yacc[YREG ? MT : 1][4];
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    if constexpr (YREG) { yacc[t][0] = 0.f; yacc[t][1] = 0.f; yacc[t][2] = 0.f; yacc[t][3] = 0.f; }
    else if (h == 0) *reinterpret_cast<float4*>(partial + (wave * M + 32 * t + j) * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  if (BROY) __syncthreads();                // the iterate written by the owners is visible
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    const int col = 32 * t + j;
    float x, y, z;
    if (BROY) {
      x = xs[col * 4]; y = xs[col * 4 + 1]; z = xs[col * 4 + 2];
    } else {
      point_coords(base + col % PTS, x, y, z);
    }
    bv[t] = coord_operand<F16>(x, y, z, h);
    if (JVP && col >= PTS) bv[t] = unit_operand<F16>(col / PTS - 1);
  }
  // lane that holds the value stream of this lane's point in m-tile 0 (JVP epilogue)
  int value_lane[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) value_lane[t] = (lane & 32) | ((32 * t + j) % PTS);

  const char* st = p.state + size_t(row) * p.state_row_bytes;
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.packed), 0, 0x7fffffff, 0x00020000);
  const f32x16 zero16 = {};
  f32x16 acc[NTW][MT];

  // epilogue of the wavefront's tiles: softplus, then either the re-split operands of the next layer (registers only) or -
  // `last`, the last hidden layer - this wavefront's share of the last linear layer in fp32, straight from the registers
  // (rounds 1-4 stored the tile and ran the out_dim <= 4 rows as a K-split MFMA layer on split operands: one more store,
  // barrier and operand rounding; the single-term variant has no lo plane to run it on)
  // (point tiles T0 .. T1 - 1; PACK: 0 = the hi operand alone, 1 = hi | lo, 2 = none - the tile feeds the last linear layer only)
  auto activate = [&](int ni, int layer, bool last, auto t0c, auto t1c, auto packc) __attribute__((always_inline)) {
    constexpr int T0 = decltype(t0c)::value, T1 = decltype(t1c)::value, PACK = decltype(packc)::value;
#pragma unroll
    for (int i = 0; i < NTW; ++i) {
      if (i < ni) {
        float sgv[JVP ? 16 : 1];                        // sigma' of the value stream (point tile 0), read before the tile is overwritten
        if constexpr (JVP) {
#pragma unroll
          for (int r = 0; r < 16; ++r) sgv[r] = sigmoid2(acc[i][0][r]);
        }
#pragma unroll
        for (int t = T0; t < T1; ++t) {
          if constexpr (SAVE && !JVP) {
            static_assert(!(SAVE && !JVP) || M == 64, "the saving forward runs 64-point workgroups (mlp_bwd_kernel's layout)");
            float* so = p.sig_out + ((((size_t(row) * size_t((n_pts + 63) >> 6) + size_t(base >> 6)) * p.sig_tiles + p.sig_base[layer] + wave + WAVES * i) * MT + t) * 64 + lane) * 16;
#pragma unroll
            for (int r = 0; r < 16; r += 4) {
              float4 sg = make_float4(sigmoid2(acc[i][t][r]), sigmoid2(acc[i][t][r + 1]), sigmoid2(acc[i][t][r + 2]), sigmoid2(acc[i][t][r + 3]));
              *reinterpret_cast<float4*>(so + r) = sg;
            }
          }
          if constexpr (SAVE && JVP) {
            // the value stream (columns 0 .. PTS-1 of point tile 0) into the 64-point-workgroup layout mlp_bwd_kernel reads:
            // this workgroup's PTS (16, or 8 in the 32-column form) points start at `base` (a multiple of PTS): 64-point group
            // base / 64, point tile (base / 32) & 1, lanes base % 32 .. + PTS - 1 of it.  (The saved layout has TWO point tiles
            // per 64-point group whatever MT is here.)
            static_assert(!JVP || PTS == 16 || PTS == 8, "value + Jacobian workgroups hold 16 or 8 points");
            if (t == 0 && j < PTS) {
              const size_t wg64 = size_t(row) * size_t((n_pts + 63) >> 6) + size_t(base >> 6);
              float* so = p.sig_out + (((wg64 * p.sig_tiles + p.sig_base[layer] + wave + WAVES * i) * 2 + ((base >> 5) & 1)) * 64 + 32 * h + int(base & 31) + j) * 16;
#pragma unroll
              for (int r = 0; r < 16; r += 4) {
                float4 sg = make_float4(sigmoid2(acc[i][0][r]), sigmoid2(acc[i][0][r + 1]), sigmoid2(acc[i][0][r + 2]), sigmoid2(acc[i][0][r + 3]));
                *reinterpret_cast<float4*>(so + r) = sg;
              }
            }
          }
          float v[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            if (JVP) {
              const float sig = __shfl(sgv[r], value_lane[t]);
              const bool is_value = 32 * t + j < PTS;
              v[r] = is_value ? softplus2(acc[i][t][r]) : sig * acc[i][t][r];
            } else {
              v[r] = softplus2(acc[i][t][r]);
            }
          }
          if (last) {
            const float4* wt = reinterpret_cast<const float4*>(wlast) + ((wave + WAVES * i) * 2 + h) * 16;
            float* yq = partial + (wave * M + 32 * t + j) * 4;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              if (c < p.out_dim) {
                float sacc = 0.f;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                  const float4 w4 = wt[c * 4 + q];
                  sacc = fmaf(w4.x, v[4 * q], sacc);
                  sacc = fmaf(w4.y, v[4 * q + 1], sacc);
                  sacc = fmaf(w4.z, v[4 * q + 2], sacc);
                  sacc = fmaf(w4.w, v[4 * q + 3], sacc);
                }
                if constexpr (YREG) {
                  yacc[t][c] += sacc;                                  // summed over the wavefront's tiles in registers
                } else {
                  sacc += __shfl_xor(sacc, 32);                        // the two halves of the tile's 32 rows
                  if (h == 0) yq[c] += sacc;                           // (own slot, in program order: no atomics, fixed order)
                }
              }
            }
          }
          // the operands take the place of the accumulators they came from (registers 8 half .. 8 half + 3: hi fragment of
          // values 8 half .. 8 half + 7, the next four: lo fragment) - no second register set lives beside the tile.
          // (__uint_as_float, NOT __builtin_bit_cast: the builtin applied to an ext-vector ELEMENT reads element 0.)
#pragma unroll
          for (int half = 0; half < 2; ++half) {
            if constexpr (PACK == 2) {
            } else if constexpr (PACK == 0) {
              const frag_t o = pack8_rn(v + 8 * half);
#pragma unroll
              for (int q = 0; q < 4; ++q) acc[i][t][8 * half + q] = __uint_as_float(o[q]);
            } else {
              const Split8 o = split8<F16>(v + 8 * half);
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                acc[i][t][8 * half + q] = __uint_as_float(o.hi[q]);
                acc[i][t][8 * half + 4 + q] = __uint_as_float(o.lo[q]);
              }
            }
          }
        }
      }
    }
  };
  // D tile (n, t): registers 8*half .. 8*half+7 of lane (h, j) are K chunk 4n + 2*half + h of point 32t + j
  // (point tiles T0 .. T1 - 1.  HALF, TAIL2's last hidden layer: these tiles alone, hi | lo, in a plane of T1 - T0 tiles each)
  auto store_tiles = [&](int ni, auto t0c, auto t1c, auto halfc) __attribute__((always_inline)) {
    constexpr int T0 = decltype(t0c)::value, T1 = decltype(t1c)::value;
    constexpr bool HALF = decltype(halfc)::value != 0;
    constexpr int MX = HALF ? 32 * (T1 - T0) : M;
    char* dst_lo = HALF ? act_hi + PART_BYTES / 2 : act_lo;
#pragma unroll
    for (int i = 0; i < NTW; ++i) {
      if (i < ni) {
        const int n = wave + WAVES * i;
#pragma unroll
        for (int t = T0; t < T1; ++t) {
#pragma unroll
          for (int half = 0; half < 2; ++half) {
            const int off = ((4 * n + 2 * half + h) * MX + 32 * (HALF ? t - T0 : t) + j) * 16;
            frag_t fh, fl;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              fh[q] = __float_as_uint(acc[i][t][8 * half + q]);
              fl[q] = __float_as_uint(acc[i][t][8 * half + 4 + q]);
            }
            *reinterpret_cast<frag_t*>(act_hi + off) = fh;
            if constexpr (!ONE || HALF) *reinterpret_cast<frag_t*>(dst_lo + off) = fl;
          }
        }
      }
    }
  };
  auto tiles_of = [&](int n_tiles) { return n_tiles > wave ? (n_tiles - wave + WAVES - 1) / WAVES : 0; };
  // accumulator init: coordinate K-step (bias, folded latent, xyz columns)
  frag_t cfrag[NTW];
  auto coord_load = [&](const LayerDev& L, int ni) __attribute__((always_inline)) {
    const frag_t* C = reinterpret_cast<const frag_t*>(st + L.c_off) + lane;
#pragma unroll
    for (int i = 0; i < NTW; ++i)
      if (i < ni) cfrag[i] = C[(wave + WAVES * i) * 64];
  };
  auto coord_mma = [&](int ni, auto t0c, auto t1c) __attribute__((always_inline)) {
    constexpr int T0 = decltype(t0c)::value, T1 = decltype(t1c)::value;
#pragma unroll
    for (int i = 0; i < NTW; ++i) {
      if (i < ni) {
#pragma unroll
        for (int t = T0; t < T1; ++t)
          acc[i][t] = mfma16<F16>(cfrag[i], bv[t], zero16);
      }
    }
  };
  auto coord_step = [&](const LayerDev& L, int ni) __attribute__((always_inline)) {
    coord_load(L, ni);
    coord_mma(ni, IC<0>{}, IC<MT>{});
  };
  constexpr int PACK_DEF = ONE ? 0 : 1;

  // ---- layer 0: coordinates only ---------------------------------------------------------------
  {
    const LayerDev& L = p.layer[0];
    const int ni = tiles_of(L.n_tiles);
    coord_step(L, ni);
    activate(ni, 0, false, IC<0>{}, IC<MT>{}, IC<PACK_DEF>{});
    store_tiles(ni, IC<0>{}, IC<MT>{}, IC<0>{});
  }

  // ---- hidden layers ---------------------------------------------------------------------------
  // A fragments through MUBUF loads: resource = the packed weights, VGPR offset = the lane's constant
  // 16 bytes, tile / K-step offset in an SGPR - no per-load 64-bit VGPR address arithmetic next to the
  // MFMA stream (tools/micro/dma.hip: the issue cost of a VGPR-addressed load there is ~3x).
  // The two-slot register ring lives across the layers: the first two K-steps of layer l+1 are requested right
  // behind the last MFMAs of layer l (NPHM_MLP_XPREFETCH), so their L2 latency runs under the epilogue and the two
  // workgroup barriers instead of in front of the next layer's first MFMA.
#ifndef NPHM_MLP_XPREFETCH
#define NPHM_MLP_XPREFETCH 1
#endif
  const unsigned w_lane = lane * 16;
  // K-steps of A fragments in flight per wavefront.  A K-step of the 32-column workgroups (MT = 1: the small launches of the
  // fitting loop) is 4 - 6 MFMAs: two steps ahead cover a third of an L2 round trip, and these variants leave > 100 VGPRs
  // unused - they keep four.  (NTW = 4 or MT = 2: no registers to spare.)
#ifndef NPHM_MLP_SLOTS_SMALL
#define NPHM_MLP_SLOTS_SMALL 4
#endif
#ifndef NPHM_MLP_SLOTS_ALL2_SMALL
#define NPHM_MLP_SLOTS_ALL2_SMALL 4
#endif
#ifndef NPHM_MLP_SLOTS_ONE
#define NPHM_MLP_SLOTS_ONE 2       // (a K-step of the 128-point variant is 8 MFMAs; four slots: 8 spilled VGPRs and 1.5 % slower)
#endif
#ifndef NPHM_MLP_SLOTS_ALL2
#define NPHM_MLP_SLOTS_ALL2 4
#endif
  constexpr bool NO_WL = ALL2 || ONE;      // no wl fragment is ever requested
  constexpr int NS = ONE ? NPHM_MLP_SLOTS_ONE : ALL2 ? ((MT == 1 && NTW == 2) ? NPHM_MLP_SLOTS_ALL2_SMALL : NPHM_MLP_SLOTS_ALL2) : (MT == 1 && NTW == 2 && !BROY) ? NPHM_MLP_SLOTS_SMALL : 2;    // (the fused solver: 240 VGPRs with four, and no faster)
  static_assert(NS % 2 == 0, "the B operand's two slots alternate with the K-step");
  frag_t ah[NS][NTW], al[NS][NTW];
  // Terms of the split product of a layer: three = xh wh + xl wh + xh wl, two = without the wl term (EvalArgs::two_pass_mask).
  // One loop body serves both (two specialised loops under a branch made hipcc spill 60-250 VGPRs): the wl fragments of a
  // two-term layer are requested out of the buffer's range (lo_lane: the range check covers the VGPR offset; such a load
  // returns zeros without a memory request - half the L2 bytes of the layer) and their MFMAs sit under one wave-uniform branch.
  auto load_a = [&](const LayerDev& L, int ni, int slot, int s, unsigned lo_lane) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NTW; ++i) {
      if (i < ni) {
        const unsigned o = __builtin_amdgcn_readfirstlane(L.w_off + (unsigned(wave + WAVES * i) * unsigned(L.k_steps) + unsigned(s)) * 2048u);
        ah[slot][i] = __builtin_bit_cast(frag_t, __builtin_amdgcn_raw_buffer_load_b128(rs_w, w_lane, o, 0));
        if constexpr (!NO_WL) al[slot][i] = __builtin_bit_cast(frag_t, __builtin_amdgcn_raw_buffer_load_b128(rs_w, lo_lane, o + 1024u, 0));
      }
    }
  };
  auto lo_lane_of = [&](int l) { return (((p.two_pass_mask | p.one_pass_mask) >> l) & 1u) ? (w_lane | 0x80000000u) : w_lane; };
  if (NPHM_MLP_XPREFETCH && p.n_linear > 2) {
    const LayerDev& L1 = p.layer[1];
    const int n1 = tiles_of(L1.n_tiles);
    const unsigned lo1 = lo_lane_of(1);
#pragma unroll
    for (int u = 0; u < NS; ++u) if (u < 2 || u < L1.k_steps) load_a(L1, n1, u, u, lo1);
  }
#pragma unroll 1
  for (int l = 1; l < p.n_linear - 1; ++l) {
    const LayerDev& L = p.layer[l];
    const int ni = tiles_of(L.n_tiles);
    const int ks = L.k_steps;
    const bool single = ONE || ((p.one_pass_mask >> l) & 1u);             // xh wh alone (wave-uniform)
    const bool three = !NO_WL && !(((p.two_pass_mask | p.one_pass_mask) >> l) & 1u);
    const unsigned lo_lane = lo_lane_of(l);
    coord_step(L, ni);       // (requesting these fragments a layer ahead as well: +-0, and the Broyden variants spill)
    __syncthreads();                                  // the previous layer's tile is complete
    if (ni > 0) {
      const frag_t* Bh = reinterpret_cast<const frag_t*>(act_hi) + h * M + j;
      const frag_t* Bl = reinterpret_cast<const frag_t*>(act_lo) + h * M + j;
      frag_t bh[2][MT], bl[2][MT];
      auto load_b = [&](int slot, int s) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < MT; ++t) {
          bh[slot][t] = Bh[2 * s * M + 32 * t];
          if constexpr (!ONE) { if (!single) bl[slot][t] = Bl[2 * s * M + 32 * t]; }
        }
      };
      // (every accumulator's hi product before any lo product: no dependent pair back to back)
#ifndef NPHM_MLP_HI_THEN_LO
#define NPHM_MLP_HI_THEN_LO 1
#endif
      auto mma = [&](int sa, int sb) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NTW; ++i) {
          if (i < ni) {
#pragma unroll
            for (int t = 0; t < MT; ++t) {
              acc[i][t] = mfma16<F16>(ah[sa][i], bh[sb][t], acc[i][t]);
              if constexpr (!ONE && !NPHM_MLP_HI_THEN_LO) { if (!single) acc[i][t] = mfma16<F16>(ah[sa][i], bl[sb][t], acc[i][t]); }
            }
          }
        }
        if constexpr (!ONE && NPHM_MLP_HI_THEN_LO) {
          if (!single) {
#pragma unroll
            for (int i = 0; i < NTW; ++i) {
              if (i < ni) {
#pragma unroll
                for (int t = 0; t < MT; ++t) acc[i][t] = mfma16<F16>(ah[sa][i], bl[sb][t], acc[i][t]);
              }
            }
          }
        }
        if constexpr (!NO_WL) {
          if (three) {
#pragma unroll
            for (int i = 0; i < NTW; ++i) {
              if (i < ni) {
#pragma unroll
                for (int t = 0; t < MT; ++t) acc[i][t] = mfma16<F16>(al[sa][i], bh[sb][t], acc[i][t]);
              }
            }
          }
        }
      };
      if (!NPHM_MLP_XPREFETCH) {
#pragma unroll
        for (int u = 0; u < NS; ++u) if (u < 2 || u < ks) load_a(L, ni, u, u, lo_lane);
      }
      load_b(0, 0);
      // slot u holds K-step s + u, the B operand alternates its two slots (k_steps is even, mlp_layout.h: the steps
      // u >= 2 of the last round may not exist)
#pragma unroll 1
      for (int s = 0; s < ks; s += NS) {
#pragma unroll
        for (int u = 0; u < NS; ++u) {
          if (u < 2 || s + u < ks) {
            if (s + u + 1 < ks) load_b((u + 1) & 1, s + u + 1);
            __builtin_amdgcn_sched_barrier(0);
            mma(u, u & 1);
            __builtin_amdgcn_sched_barrier(0);
            if (s + u + NS < ks) load_a(L, ni, u, s + u + NS, lo_lane);
          }
        }
      }
    }
    if (NPHM_MLP_XPREFETCH && l + 1 < p.n_linear - 1) {
      const LayerDev& Ln = p.layer[l + 1];
      const int nn = tiles_of(Ln.n_tiles);
      const unsigned lon = lo_lane_of(l + 1);
#pragma unroll
      for (int u = 0; u < NS; ++u) if (u < 2 || u < Ln.k_steps) load_a(Ln, nn, u, u, lon);
      __builtin_amdgcn_sched_barrier(0);
    }
    // (the last hidden layer feeds the last linear layer from its registers: nothing to store.  Its operands are still packed
    // like the others' - a second epilogue body here costs hipcc's register allocation far more than the dead converts)
    const bool last = l == p.n_linear - 2;
    if constexpr (TAIL2) { if (l == p.n_linear - 3) break; }          // (its epilogue: below, hi | lo)
    activate(ni, l, last, IC<0>{}, IC<MT>{}, IC<PACK_DEF>{});
    if (!last) {
      __syncthreads();                                // every wavefront has read the old tile
      store_tiles(ni, IC<0>{}, IC<MT>{}, IC<0>{});
    }
  }

  // ---- TAIL2: the last hidden layer, two-term, in two point halves --------------------------------------------------------
  if constexpr (TAILK) {
    // The K-split form: K half A = the tiles i < NTW / 2 of every wavefront (layer lp's outputs 0 .. 32 NA - 1) goes to LDS as
    // hi | lo half planes of ALL the points, K half B is written to the workgroup's slot of tail_ws as the image of the same
    // half planes and comes back by LDS-DMA once half A has been consumed: the accumulators run through both halves, the
    // layer's weights are streamed ONCE (the per-accumulator order of the products is that of KIND 5: same bits).
    constexpr int IH = NTW / 2, NA = WAVES * IH;
    static_assert(NTW % 2 == 0 && NS == 2, "K halves by tile parity of the wavefront's tile list; two-slot ring");
    const int lp = p.n_linear - 3, ll = p.n_linear - 2;
    const int ni_p = tiles_of(p.layer[lp].n_tiles);
    const LayerDev& L = p.layer[ll];
    const int ni = tiles_of(L.n_tiles);
    const int ks = L.k_steps;
    const int ks_a = ks < 2 * NA ? ks : 2 * NA;                        // K-steps of half A (two per 32-feature tile)
    const unsigned slot = __builtin_amdgcn_readfirstlane(reinterpret_cast<const unsigned*>(xs)[0]);
    const unsigned ws_off = slot * unsigned(PART_BYTES);
    const __amdgpu_buffer_rsrc_t rs_t = __builtin_amdgcn_make_buffer_rsrc(p.tail_ws, 0, 0x7fffffff, 0x00020000);
    typedef int v4i __attribute__((ext_vector_type(4)));
    v4i rs_d;                                                         // the same resource as four SGPRs, for the inline-asm loads
    {
      const uint64_t a64 = reinterpret_cast<uint64_t>(p.tail_ws);
      rs_d[0] = __builtin_amdgcn_readfirstlane(int(uint32_t(a64)));
      rs_d[1] = __builtin_amdgcn_readfirstlane(int(uint32_t(a64 >> 32) & 0xffffu));
      rs_d[2] = 0x7fffffff;
      rs_d[3] = 0x00020000;
    }
    activate(ni_p, lp, false, IC<0>{}, IC<MT>{}, IC<1>{});             // hi | lo of every tile, in the registers of their D tiles
    __syncthreads();                                                  // every wavefront has read the old tile
#pragma unroll
    for (int i = 0; i < NTW; ++i) {
      if (i < ni_p) {
        const int n = wave + WAVES * (i < IH ? i : i - IH);           // tile inside its K half
#pragma unroll
        for (int t = 0; t < MT; ++t) {
#pragma unroll
          for (int half = 0; half < 2; ++half) {
            frag_t fh, fl;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              fh[q] = __float_as_uint(acc[i][t][8 * half + q]);
              fl[q] = __float_as_uint(acc[i][t][8 * half + 4 + q]);
            }
            if (i < IH) {
              const int off = ((4 * n + 2 * half + h) * M + 32 * t + j) * 16;
              *reinterpret_cast<frag_t*>(act_hi + off) = fh;
              *reinterpret_cast<frag_t*>(act_hi + PART_BYTES / 2 + off) = fl;
            } else {
              const unsigned so = __builtin_amdgcn_readfirstlane(ws_off + unsigned(((4 * n + 2 * half) * M + 32 * t) * 16));
              const unsigned vo = unsigned((h * M + j) * 16);
              __builtin_amdgcn_raw_buffer_store_b128(fh, rs_t, vo, so, 0);
              __builtin_amdgcn_raw_buffer_store_b128(fl, rs_t, vo, so + unsigned(PART_BYTES / 2), 0);
            }
          }
        }
      }
    }
    auto k_pass = [&](int s0, int s1) __attribute__((always_inline)) {          // K-steps s0 .. s1 - 1: chunks from 0 of the planes in LDS
      if (ni > 0) {
        const frag_t* Bh = reinterpret_cast<const frag_t*>(act_hi) + h * M + j;
        const frag_t* Bl = reinterpret_cast<const frag_t*>(act_hi + PART_BYTES / 2) + h * M + j;
        frag_t bh[2][MT], bl[2][MT];
        auto load_b = [&](int sb, int s) __attribute__((always_inline)) {
#pragma unroll
          for (int t = 0; t < MT; ++t) {
            bh[sb][t] = Bh[2 * (s - s0) * M + 32 * t];
            bl[sb][t] = Bl[2 * (s - s0) * M + 32 * t];
          }
        };
        auto mma = [&](int sa, int sb) __attribute__((always_inline)) {
#pragma unroll
          for (int i = 0; i < NTW; ++i) {
            if (i < ni) {
#pragma unroll
              for (int t = 0; t < MT; ++t) acc[i][t] = mfma16<F16>(ah[sa][i], bh[sb][t], acc[i][t]);
            }
          }
#pragma unroll
          for (int i = 0; i < NTW; ++i) {
            if (i < ni) {
#pragma unroll
              for (int t = 0; t < MT; ++t) acc[i][t] = mfma16<F16>(ah[sa][i], bl[sb][t], acc[i][t]);
            }
          }
        };
        load_b(0, s0);
#pragma unroll 1
        for (int s = s0; s < s1; s += 2) {                             // (s0, s1 even; the ring's slot u holds K-step s + u)
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            if (s + u + 1 < s1) load_b((u + 1) & 1, s + u + 1);
            __builtin_amdgcn_sched_barrier(0);
            mma(u, u & 1);
            __builtin_amdgcn_sched_barrier(0);
            if (s + u + 2 < ks) load_a(L, ni, u, s + u + 2, w_lane);  // (runs on into the second half)
          }
        }
      }
    };
    coord_step(L, ni);                                                // (behind the stores: they read the accumulators)
    __syncthreads();                                                  // half A is complete
    k_pass(0, ks_a);
    if (ks > ks_a) {
      __syncthreads();                                                // every wavefront has read half A
      {
        // this wavefront's sixteenth of the image: 16 pieces of 1 KiB, four per M0
        const unsigned voff = lane * 16u;
#pragma unroll
        for (int k = 0; k < 16; k += 4) {
          const unsigned piece = unsigned(wave * 16 + k) * 1024u;
          const unsigned dst = __builtin_amdgcn_readfirstlane(unsigned(size_t((__attribute__((address_space(3))) const char*)act_hi)) + piece);
          const unsigned so = __builtin_amdgcn_readfirstlane(ws_off + piece);
          unsigned keep;
          asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                       "buffer_load_dwordx4 %1, %2, %4 offen lds\n\t"
                       "buffer_load_dwordx4 %1, %2, %4 offen offset:1024 lds\n\t"
                       "buffer_load_dwordx4 %1, %2, %4 offen offset:2048 lds\n\t"
                       "buffer_load_dwordx4 %1, %2, %4 offen offset:3072 lds\n\t"
                       "s_mov_b32 m0, %0"
                       : "=&s"(keep) : "v"(voff), "s"(rs_d), "s"(dst), "s"(so) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // (hipcc does not see these loads)
      }
      __syncthreads();                                                // half B is in LDS, nobody reads the slot any more
      if (threadIdx.x == 0) __hip_atomic_store(p.tail_flags + slot, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      k_pass(ks_a, ks);
    } else {
      if (threadIdx.x == 0) __hip_atomic_store(p.tail_flags + slot, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    activate(ni, ll, true, IC<0>{}, IC<MT>{}, IC<2>{});
  } else if constexpr (TAIL2) {
    constexpr int MH = MT / 2, MX = 32 * MH;
    const int lp = p.n_linear - 3, ll = p.n_linear - 2;
    const int ni_p = tiles_of(p.layer[lp].n_tiles);
    const LayerDev& L = p.layer[ll];
    const int ni = tiles_of(L.n_tiles);
    const int ks = L.k_steps;
    // (the ring holds the layer's first K-steps: requested behind the last MFMAs of layer lp)
    activate(ni_p, lp, false, IC<0>{}, IC<MT>{}, IC<1>{});             // hi | lo of BOTH halves, in the registers of their D tiles
    __syncthreads();                                                  // every wavefront has read the old tile
    store_tiles(ni_p, IC<0>{}, IC<MH>{}, IC<1>{});                     // first half: hi | lo planes of MH point tiles
    auto tail_pass = [&](auto t0c) __attribute__((always_inline)) {
      constexpr int T0 = decltype(t0c)::value;
      if (ni > 0) {
        const frag_t* Bh = reinterpret_cast<const frag_t*>(act_hi) + h * MX + j;
        const frag_t* Bl = reinterpret_cast<const frag_t*>(act_hi + PART_BYTES / 2) + h * MX + j;
        frag_t bh[2][MH], bl[2][MH];
        auto load_b = [&](int slot, int s) __attribute__((always_inline)) {
#pragma unroll
          for (int t = 0; t < MH; ++t) {
            bh[slot][t] = Bh[2 * s * MX + 32 * t];
            bl[slot][t] = Bl[2 * s * MX + 32 * t];
          }
        };
        auto mma = [&](int sa, int sb) __attribute__((always_inline)) {
#pragma unroll
          for (int i = 0; i < NTW; ++i) {
            if (i < ni) {
#pragma unroll
              for (int t = 0; t < MH; ++t) acc[i][T0 + t] = mfma16<F16>(ah[sa][i], bh[sb][t], acc[i][T0 + t]);
            }
          }
#pragma unroll
          for (int i = 0; i < NTW; ++i) {
            if (i < ni) {
#pragma unroll
              for (int t = 0; t < MH; ++t) acc[i][T0 + t] = mfma16<F16>(ah[sa][i], bl[sb][t], acc[i][T0 + t]);
            }
          }
        };
        load_b(0, 0);
#pragma unroll 1
        for (int s = 0; s < ks; s += NS) {
#pragma unroll
          for (int u = 0; u < NS; ++u) {
            if (u < 2 || s + u < ks) {
              if (s + u + 1 < ks) load_b((u + 1) & 1, s + u + 1);
              __builtin_amdgcn_sched_barrier(0);
              mma(u, u & 1);
              __builtin_amdgcn_sched_barrier(0);
              if (s + u + NS < ks) load_a(L, ni, u, s + u + NS, w_lane);
            }
          }
        }
      }
    };
    coord_load(L, ni);
    coord_mma(ni, IC<0>{}, IC<MH>{});
    __syncthreads();                                                  // the first half's operands are complete
    tail_pass(IC<0>{});
#pragma unroll
    for (int u = 0; u < NS; ++u) if (u < 2 || u < ks) load_a(L, ni, u, u, w_lane);       // the layer's weights, second time
    __builtin_amdgcn_sched_barrier(0);
    activate(ni, ll, true, IC<0>{}, IC<MH>{}, IC<2>{});
    __syncthreads();                                                  // every wavefront has read the first half
    store_tiles(ni_p, IC<MH>{}, IC<MT>{}, IC<1>{});
    coord_load(L, ni);
    coord_mma(ni, IC<MH>{}, IC<MT>{});
    __syncthreads();
    tail_pass(IC<MH>{});
    activate(ni, ll, true, IC<MH>{}, IC<MT>{}, IC<2>{});
  }

  // ---- last linear layer: the wavefronts' shares (activate, `last`) meet in LDS and are added in wavefront order ----------------
  {
    if constexpr (YREG) {
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        float4 y4;
        y4.x = yacc[t][0] + __shfl_xor(yacc[t][0], 32);                // the two halves of the tiles' 32 rows
        y4.y = yacc[t][1] + __shfl_xor(yacc[t][1], 32);
        y4.z = yacc[t][2] + __shfl_xor(yacc[t][2], 32);
        y4.w = yacc[t][3] + __shfl_xor(yacc[t][3], 32);
        if (h == 0) *reinterpret_cast<float4*>(partial + (wave * M + 32 * t + j) * 4) = y4;
      }
    }
    __syncthreads();
    if (!BROY) {
      for (int e = threadIdx.x; e < M * p.out_dim; e += blockDim.x) {
        const int m = e / p.out_dim, c = e % p.out_dim;
        const int stream = m / PTS;                       // 0 unless JVP
        const int64_t i = base + m % PTS;
        if (i < n_end) {
          float v = 0.f;
#pragma unroll
          for (int w = 0; w < WAVES; ++w) v += partial[(w * M + m) * 4 + c];
          v *= 1.f / SP_SCALE;                                // the activations carry k (mlp_layout.h)
          if (stream == 0) v += wlast[p.last_tiles * 128 + c];  // bias (the tangent streams have none)
          if (p.add_input && c < 3) {
            if (stream == 0) {
              float x, y, z;
              point_coords(i, x, y, z);
              v += c == 0 ? x : (c == 1 ? y : z);
            } else if (c == stream - 1) {
              v += 1.f;                                   // d (x + F) / d x_c
            }
          }
          if (JVP) p.out[((int64_t(row) * n_pts + i) * 4 + stream) * p.out_dim + c] = v;
          else p.out[(int64_t(row) * n_pts + i) * p.out_dim + c] = v;
        }
      }
      if constexpr (JVP) {
        // the inverse Jacobian of the posed points with it (iterative_root_finding.py:118, fitting.py:102 `.inverse()`): one thread
        // per point, the adjugate formula of inverse3x3_kernel on the values written above - a launch of its own before
        if (p.jinv_out) {
          for (int m = threadIdx.x; m < PTS; m += blockDim.x) {
            const int64_t i = base + m;
            if (i < n_end) {
              float a[9];
#pragma unroll
              for (int o = 0; o < 3; ++o)
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                  float v = 0.f;
#pragma unroll
                  for (int w = 0; w < WAVES; ++w) v += partial[(w * M + (c + 1) * PTS + m) * 4 + o];
                  v *= 1.f / SP_SCALE;
                  if (p.add_input && o == c) v += 1.f;
                  a[3 * o + c] = v;
                }
              const float c00 = a[4] * a[8] - a[5] * a[7], c01 = a[5] * a[6] - a[3] * a[8], c02 = a[3] * a[7] - a[4] * a[6];
              const float det = a[0] * c00 + a[1] * c01 + a[2] * c02;
              const float r = 1.f / det;
              float* q = p.jinv_out + (int64_t(row) * n_pts + i) * 9;
              q[0] = c00 * r; q[1] = (a[2] * a[7] - a[1] * a[8]) * r; q[2] = (a[1] * a[5] - a[2] * a[4]) * r;
              q[3] = c01 * r; q[4] = (a[0] * a[8] - a[2] * a[6]) * r; q[5] = (a[2] * a[3] - a[0] * a[5]) * r;
              q[6] = c02 * r; q[7] = (a[1] * a[6] - a[0] * a[7]) * r; q[8] = (a[0] * a[4] - a[1] * a[3]) * r;
            }
          }
        }
      }
    }
  }
  }  // evaluation (skipped when the start residual is given)
  if (!BROY) break;

  // ---- KIND 2: one Broyden step per evaluation (iterative_root_finding.py:24-69) ------------------
  int flag = 0;
  if (threadIdx.x < M) {
    const int m = threadIdx.x;
    float gnew[3];
    if (given0) {
      const int64_t i = base + m;
      const int64_t ic = int64_t(row) * n_pts + (i < n_pts ? i : n_pts - 1);
#pragma unroll
      for (int c = 0; c < 3; ++c) gnew[c] = p.posed0[ic * p.posed0_stride + c] - bobs[c];
    } else {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) v += partial[(w * M + m) * 4 + c];
        gnew[c] = (v * (1.f / SP_SCALE) + wlast[p.last_tiles * 128 + c] + bx[c]) - bobs[c];     // residual (x + F(x)) - obs
      }
    }
    if (it == 0) {
#pragma unroll
      for (int c = 0; c < 3; ++c) bgx[c] = gnew[c];
      bbest = sqrtf(bgx[0] * bgx[0] + bgx[1] * bgx[1] + bgx[2] * bgx[2]);
      bactive = bowner;                                   // the first update is applied to every point
    } else {
      if (bactive) {
#pragma unroll
        for (int c = 0; c < 3; ++c) { bdgx[c] = gnew[c] - bgx[c]; bgx[c] += bdgx[c]; }
      }
      const float nrm = sqrtf(bgx[0] * bgx[0] + bgx[1] * bgx[1] + bgx[2] * bgx[2]);
      if (nrm < bbest) bbest = nrm;
      bactive = bowner && bbest > p.cvg && nrm < p.dvg;   // converged and diverged points stop moving
    }
    flag = bactive;
  }
  const int any_active = __syncthreads_or(flag);
  if ((it > 0 && !any_active) || it == p.max_steps) break;
  if (threadIdx.x < M && bactive) {
    if (it > 0) {
      // rank-one update of the inverse Jacobian: J += (dx - J dg) (dx^T J) / (dx^T J dg)
      float vT[3], a[3], b = 0.f;
#pragma unroll
      for (int jj = 0; jj < 3; ++jj) vT[jj] = bdx[0] * bJ[jj] + bdx[1] * bJ[3 + jj] + bdx[2] * bJ[6 + jj];
#pragma unroll
      for (int i2 = 0; i2 < 3; ++i2)
        a[i2] = bdx[i2] - (bJ[3 * i2] * bdgx[0] + bJ[3 * i2 + 1] * bdgx[1] + bJ[3 * i2 + 2] * bdgx[2]);
#pragma unroll
      for (int jj = 0; jj < 3; ++jj) b += vT[jj] * bdgx[jj];
      b += b >= 0.f ? p.eps : -p.eps;
#pragma unroll
      for (int i2 = 0; i2 < 3; ++i2) {
        const float u = a[i2] / b;
#pragma unroll
        for (int jj = 0; jj < 3; ++jj) bJ[3 * i2 + jj] += u * vT[jj];
      }
    }
#pragma unroll
    for (int i2 = 0; i2 < 3; ++i2) {
      bdx[i2] = -(bJ[3 * i2] * bgx[0] + bJ[3 * i2 + 1] * bgx[1] + bJ[3 * i2 + 2] * bgx[2]);
      bx[i2] += bdx[i2];
      xs[threadIdx.x * 4 + i2] = bx[i2];
    }
  }
  }  // evaluation loop

  if (BROY && threadIdx.x < M && bowner) {
    const int64_t i = int64_t(row) * n_pts + base + threadIdx.x;
    p.out[i * 3] = bx[0]; p.out[i * 3 + 1] = bx[1]; p.out[i * 3 + 2] = bx[2];
    p.diff_out[i] = bbest;
    p.valid_out[i] = bbest < p.cvg ? 1 : 0;
  }
}

// batched 3x3 inverse by the adjugate formula (the inverse Jacobians of the correspondence search and of the
// implicit differentiation, iterative_root_finding.py:118 / fitting.py:102 `.inverse()`): one thread per matrix
__global__ __launch_bounds__(256) void inverse3x3_kernel(const float* __restrict__ in, float* __restrict__ out, int64_t n) {
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float a[9];
#pragma unroll
  for (int c = 0; c < 9; ++c) a[c] = in[i * 9 + c];
  const float c00 = a[4] * a[8] - a[5] * a[7], c01 = a[5] * a[6] - a[3] * a[8], c02 = a[3] * a[7] - a[4] * a[6];
  const float det = a[0] * c00 + a[1] * c01 + a[2] * c02;
  const float r = 1.f / det;
  float* o = out + i * 9;
  o[0] = c00 * r; o[1] = (a[2] * a[7] - a[1] * a[8]) * r; o[2] = (a[1] * a[5] - a[2] * a[4]) * r;
  o[3] = c01 * r; o[4] = (a[0] * a[8] - a[2] * a[6]) * r; o[5] = (a[2] * a[3] - a[0] * a[5]) * r;
  o[6] = c02 * r; o[7] = (a[1] * a[6] - a[0] * a[7]) * r; o[8] = (a[0] * a[4] - a[1] * a[3]) * r;
}

template <int MT, int NTW, bool ONE = false>
constexpr size_t lds_bytes() {
  return size_t(32 * WAVES * NTW / 8) * (32 * MT) * 16 * (ONE ? 1 : 2) + (WAVES + 1) * 32 * MT * 4 * 4 + last_table_floats(WAVES * NTW) * sizeof(float);
}

}  // namespace mlp
}  // namespace nphm

// ============================================================================================
// C ABI (include/nphm_amd.h)
// ============================================================================================
using nphm::mlp::Config;
using nphm::mlp::Plan;

// `numerics` of the plain evaluation entry points (include/nphm_amd.h): low byte = operand format (0 split-bf16, 1 split-f16),
// bits 8.. = mask of the hidden layers that run the two-term product (bit l = linear layer l; layer 0 and the last never)
static bool mlp_numerics_ok(int numerics) { return (numerics & 0xff) <= 1 && numerics >= 0; }
// bits 20.. (split-f16, plain evaluation entry points - points / lattice - only) = mask of the hidden layers that run the
// SINGLE-term product rn(x) wh; all of them: the 128-points-per-workgroup variant (mlp_eval_kernel, ONE)
static bool mlp_eval_numerics_ok(int numerics) { return mlp_numerics_ok(numerics) && ((numerics >> 20) == 0 || (numerics & 0xff) == 1); }
template <int MODE, int KIND = 0>
static int launch_eval(const Plan& plan, nphm::mlp::EvalArgs& a, int64_t n_pts, int n_rows, hipStream_t st, int numerics = 0,
                       int columns = 64) {
  using namespace nphm::mlp;
  const bool f16 = (numerics & 0xff) == 1;
  if (MODE == 0) {
    if (a.point_end == 0 && a.point_base == 0) a.point_end = a.n_points;                 // the whole row
    if (a.point_base < 0 || a.point_end > a.n_points || a.point_base >= a.point_end)
      return nphm_fail_msg("nphm_mlp_eval: bad point range");
    n_pts = a.point_end - a.point_base;                                                  // points per row of THIS launch
  }
  constexpr bool JVPK = KIND == 1 || KIND == 4;
  if (columns != 64 && !(JVPK && columns == 32 && plan.variant == 0))
    return nphm_fail_msg("nphm_mlp_eval: 32-column workgroups exist for the value + Jacobian launches of hidden <= 512 nets");
  if (JVPK && (a.point_base % (columns / 4)) != 0) return nphm_fail_msg("nphm_mlp_eval: point_base must be a multiple of the workgroup's points");
  for (int l = 0; l < plan.n_linear; ++l) {
    a.layer[l].n_tiles = plan.layer[l].n_tiles;
    a.layer[l].k_steps = plan.layer[l].k_steps;
    a.layer[l].w_off = plan.layer[l].w_off;
    a.layer[l].c_off = plan.layer[l].c_off;
  }
  a.n_linear = plan.n_linear;
  // both buffers hold the two formats back to back: [bf16 fragments | f16 fragments], per state row [bf16 | f16]
  a.state_row_bytes = 2 * plan.state_row_bytes;
  a.last_tab = reinterpret_cast<const float*>(a.packed + 2 * plan.packed_bytes);
  a.last_tiles = plan.layer[plan.n_linear - 1].k_steps / 2;
  if (f16) { a.packed += plan.packed_bytes; a.state += plan.state_row_bytes; }
  const unsigned hidden_mask = (1u << (plan.n_linear - 1)) - 2u;                        // hidden GEMM layers 1 .. n_linear - 2
  a.one_pass_mask = (unsigned(numerics) >> 20) & hidden_mask;
  a.two_pass_mask = (unsigned(numerics) >> 8) & hidden_mask & ~a.one_pass_mask;
  if (a.one_pass_mask && (KIND != 0 || !f16)) return nphm_fail_msg("nphm_mlp_eval: the single-term product serves the plain split-f16 evaluation only");
  // every hidden GEMM layer single-term: the variant without a lo plane (mlp_eval_kernel, ONE)
  const bool one = KIND == 0 && f16 && plan.n_linear > 2 && a.one_pass_mask == hidden_mask
#ifdef NPHM_MLP_NO_ONE
                   && false
#endif
      ;
  // every hidden GEMM layer single-term but the LAST, which is two-term: the same variant with that layer in two point halves (TAIL2)
  const unsigned last_bit = 1u << (plan.n_linear - 2);
  const bool tail2 = KIND == 0 && f16 && !one && plan.n_linear >= 4 && a.one_pass_mask == (hidden_mask & ~last_bit) && (a.two_pass_mask & last_bit)
#if defined(NPHM_MLP_NO_ONE) || defined(NPHM_MLP_NO_TAIL2)
                     && false
#endif
      ;
  // no hidden GEMM layer three-term (split-f16 only): the variant without wl fragments (mlp_eval_kernel, ALL2)
  const bool all2 = f16 && !one && !tail2 && plan.n_linear > 2 && (a.two_pass_mask | a.one_pass_mask) == hidden_mask && (KIND == 0 || plan.variant == 0)
#ifdef NPHM_MLP_NO_ALL2
                    && false
#endif
      ;
  a.sig_tiles = 0;
  for (int l = 0; l < plan.n_linear - 1; ++l) { a.sig_base[l] = a.sig_tiles; a.sig_tiles += plan.layer[l].n_tiles; }
  // Small Broyden batches of the hidden <= 512 nets (the fitting loop: 5 x 1000 points) run 32 points per workgroup:
  // twice the workgroups (the 64-point form occupies 79 of the 256 CUs) at 64 KiB of LDS: 272 -> 205 us.
  constexpr bool SMALL_OK = (KIND == 2 || JVPK) && MODE == 0;      // (a whole value+Jacobian launch measured 244 us with 32 columns, 213 with 64)
  const bool small = SMALL_OK && plan.variant == 0 && (JVPK ? columns == 32 : n_pts * int64_t(n_rows) <= 64 * 1024)
#ifdef NPHM_MLP_NO_SMALL
                     && false
#endif
      ;
  const int M = (small ? 32 : plan.variant == 0 ? 64 : 32) * (one || tail2 ? 2 : 1) / ((KIND == 1 || KIND == 4) ? 4 : 1);      // points per workgroup
  const int64_t tiles = (n_pts + M - 1) / M;
  if (tiles > 0x7fffffffLL) return nphm_fail_msg("nphm_mlp_eval: too many points for one launch");
  const dim3 grid((unsigned)tiles, n_rows), block(64 * WAVES);
  hipError_t e = hipSuccess;
  auto go = [&](auto k, size_t lds) {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
    if (e != hipSuccess) return nphm_fail("nphm_mlp_eval: LDS opt-in", e);
    hipLaunchKernelGGL(k, grid, block, lds, st, a);
    return 0;
  };
  if (small) {
    if constexpr (SMALL_OK) {
      if (all2 ? go(mlp_eval_kernel<1, 2, MODE, KIND, true, true>, lds_bytes<1, 2>())
               : f16 ? go(mlp_eval_kernel<1, 2, MODE, KIND, true>, lds_bytes<1, 2>()) : go(mlp_eval_kernel<1, 2, MODE, KIND, false>, lds_bytes<1, 2>())) return -2;
    }
  } else if (one) {
    if constexpr (KIND == 0) {
      if (plan.variant == 0 ? go(mlp_eval_kernel<4, 2, MODE, 0, true, false, true>, lds_bytes<4, 2, true>())
                            : go(mlp_eval_kernel<2, 4, MODE, 0, true, false, true>, lds_bytes<2, 4, true>())) return -2;
    }
  } else if (tail2) {
    if constexpr (KIND == 0) {
      if (a.tail_ws) {
        // the K-split form (KIND 6): the busy words of the parking slots start at zero
        e = hipMemsetAsync(a.tail_flags, 0, TAIL_FLAG_BYTES, st);
        if (e != hipSuccess) return nphm_fail("nphm_mlp_eval: workspace", e);
        if (plan.variant == 0 ? go(mlp_eval_kernel<4, 2, MODE, 6, true, false, true>, lds_bytes<4, 2, true>())
                              : go(mlp_eval_kernel<2, 4, MODE, 6, true, false, true>, lds_bytes<2, 4, true>())) return -2;
      } else if (plan.variant == 0 ? go(mlp_eval_kernel<4, 2, MODE, 5, true, false, true>, lds_bytes<4, 2, true>())
                                   : go(mlp_eval_kernel<2, 4, MODE, 5, true, false, true>, lds_bytes<2, 4, true>())) return -2;
    }
  } else if (plan.variant == 0) {
    if (all2 ? go(mlp_eval_kernel<2, 2, MODE, KIND, true, true>, lds_bytes<2, 2>())
             : f16 ? go(mlp_eval_kernel<2, 2, MODE, KIND, true>, lds_bytes<2, 2>()) : go(mlp_eval_kernel<2, 2, MODE, KIND, false>, lds_bytes<2, 2>())) return -2;
  } else if constexpr (KIND == 3 || KIND == 4) {
    return nphm_fail_msg("nphm_mlp_eval_points_saving: only the hidden <= 512 variant has a backward kernel");
  } else if constexpr (KIND == 0) {
    if (all2 ? go(mlp_eval_kernel<1, 4, MODE, 0, true, true>, lds_bytes<1, 4>())
             : f16 ? go(mlp_eval_kernel<1, 4, MODE, 0, true>, lds_bytes<1, 4>()) : go(mlp_eval_kernel<1, 4, MODE, 0, false>, lds_bytes<1, 4>())) return -2;
  } else {
    // the hidden <= 1024 variant (NPM) keeps bf16 halves for its tangent / Broyden forms (nothing drives them hard: the
    // fitting loop's expression decoder is the hidden <= 512 one)
    if (f16) return nphm_fail_msg("nphm_mlp_eval: split-f16 tangent / Broyden kernels exist for hidden <= 512 only");
    if (go(mlp_eval_kernel<1, 4, MODE, KIND>, lds_bytes<1, 4>())) return -2;
  }
  e = hipGetLastError();
  if (e != hipSuccess) return nphm_fail("nphm_mlp_eval launch", e);
  return 0;
}

extern "C" {

static bool plan_of(int lat_dim, int hidden_dim, int nlayers, int out_dim, Plan& plan) {
  Config c{lat_dim, hidden_dim, nlayers, out_dim};
  return nphm::mlp::make_plan(c, plan);
}

int nphm_mlp_supported(int lat_dim, int hidden_dim, int nlayers, int out_dim, int input_dim, float beta,
                       int num_freq_bands) {
  Plan plan;
  return input_dim == 3 && beta == 100.f && num_freq_bands == 0 && plan_of(lat_dim, hidden_dim, nlayers, out_dim, plan);
}

size_t nphm_mlp_packed_bytes(int lat_dim, int hidden_dim, int nlayers, int out_dim) {
  Plan plan;
  return plan_of(lat_dim, hidden_dim, nlayers, out_dim, plan) ? 2 * plan.packed_bytes + plan.last_bytes : 0;      // [split-bf16 | split-f16] fragments | fp32 last layer
}

size_t nphm_mlp_latent_state_bytes(int lat_dim, int hidden_dim, int nlayers, int out_dim, int n_rows) {
  Plan plan;
  return plan_of(lat_dim, hidden_dim, nlayers, out_dim, plan) ? 2 * plan.state_row_bytes * size_t(n_rows > 0 ? n_rows : 0) : 0;
}

static int fill_table(nphm::mlp::PtrTable& t, const Plan& plan, const float* const* w, const float* const* b,
                      const char* who) {
  if (!w || !b) return nphm_fail_msg(who);
  for (int l = 0; l < plan.n_linear; ++l) {
    if (!w[l] || !b[l]) return nphm_fail_msg(who);
    t.w[l] = w[l];
    t.b[l] = b[l];
  }
  return 0;
}

int nphm_mlp_pack(int lat_dim, int hidden_dim, int nlayers, int out_dim,
                  const float* const* lin_weight, const float* const* lin_bias, void* packed, void* stream) {
  nphm::mlp::PackArgs a;
  if (!plan_of(lat_dim, hidden_dim, nlayers, out_dim, a.plan)) return nphm_fail_msg("nphm_mlp_pack: unsupported architecture");
  if (!packed) return nphm_fail_msg("nphm_mlp_pack: null packed buffer");
  if (fill_table(a.t, a.plan, lin_weight, lin_bias, "nphm_mlp_pack: null weight/bias pointer")) return -2;
  a.out = static_cast<uint16_t*>(packed);
  hipLaunchKernelGGL(nphm::mlp::mlp_pack_kernel, dim3(1024, a.plan.n_linear), dim3(256), 0,
                     static_cast<hipStream_t>(stream), a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return nphm_fail("nphm_mlp_pack launch", e);
  return 0;
}

int nphm_mlp_prepare_latent(int lat_dim, int hidden_dim, int nlayers, int out_dim,
                            const float* const* lin_weight, const float* const* lin_bias,
                            const float* cond_rows, int n_rows, void* latent_state, void* stream) {
  nphm::mlp::PrepArgs a;
  if (!plan_of(lat_dim, hidden_dim, nlayers, out_dim, a.plan))
    return nphm_fail_msg("nphm_mlp_prepare_latent: unsupported architecture");
  if (n_rows <= 0) return nphm_fail_msg("nphm_mlp_prepare_latent: n_rows must be > 0");
  if ((!cond_rows && lat_dim > 0) || !latent_state) return nphm_fail_msg("nphm_mlp_prepare_latent: null pointer");
  if (fill_table(a.t, a.plan, lin_weight, lin_bias, "nphm_mlp_prepare_latent: null weight/bias pointer")) return -2;
  a.cond = cond_rows;
  a.lat_dim = lat_dim;
  a.state = static_cast<char*>(latent_state);
  hipLaunchKernelGGL(nphm::mlp::mlp_prepare_kernel, dim3(a.plan.n_linear, n_rows, nphm::mlp::PREP_SPLIT), dim3(nphm::mlp::PREP_THREADS), 0,
                     static_cast<hipStream_t>(stream), a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return nphm_fail("nphm_mlp_prepare_latent launch", e);
  return 0;
}

size_t nphm_mlp_eval_workspace_bytes(void) {
  return nphm::mlp::TAIL_FLAG_BYTES + size_t(nphm::mlp::TAIL_SLOTS) * nphm::mlp::TAIL_SLOT_BYTES;
}

// [busy words | slots]; NULL: no workspace (the two-point-halves form), < 0: refused
static int set_workspace(nphm::mlp::EvalArgs& a, void* workspace, size_t workspace_bytes, const char* who) {
  if (!workspace) return 0;
  if (workspace_bytes < nphm_mlp_eval_workspace_bytes() || (reinterpret_cast<uintptr_t>(workspace) & 15)) return nphm_fail_msg(who);
  a.tail_flags = static_cast<unsigned*>(workspace);
  a.tail_ws = static_cast<char*>(workspace) + nphm::mlp::TAIL_FLAG_BYTES;
  a.tail_slots = nphm::mlp::TAIL_SLOTS;
  return 0;
}

int nphm_mlp_eval_points_ws(int lat_dim, int hidden_dim, int nlayers, int out_dim,
                            const void* packed, const void* latent_state,
                            const float* xyz, int n_rows, int64_t n_points, int add_input, int numerics,
                            float* out, void* workspace, size_t workspace_bytes, void* stream);

int nphm_mlp_eval_points(int lat_dim, int hidden_dim, int nlayers, int out_dim,
                         const void* packed, const void* latent_state,
                         const float* xyz, int n_rows, int64_t n_points, int add_input, int numerics,
                         float* out, void* stream) {
  return nphm_mlp_eval_points_ws(lat_dim, hidden_dim, nlayers, out_dim, packed, latent_state, xyz, n_rows, n_points, add_input,
                                 numerics, out, nullptr, 0, stream);
}

int nphm_mlp_eval_points_ws(int lat_dim, int hidden_dim, int nlayers, int out_dim,
                            const void* packed, const void* latent_state,
                            const float* xyz, int n_rows, int64_t n_points, int add_input, int numerics,
                            float* out, void* workspace, size_t workspace_bytes, void* stream) {
  Plan plan;
  if (!plan_of(lat_dim, hidden_dim, nlayers, out_dim, plan)) return nphm_fail_msg("nphm_mlp_eval_points: unsupported architecture");
  if (!packed || !latent_state || !xyz || !out) return nphm_fail_msg("nphm_mlp_eval_points: null pointer");
  if (n_rows <= 0 || n_points <= 0) return nphm_fail_msg("nphm_mlp_eval_points: empty input");
  if (!mlp_eval_numerics_ok(numerics)) return nphm_fail_msg("nphm_mlp_eval_points: unknown numerics format");
  nphm::mlp::EvalArgs a;
  memset(&a, 0, sizeof(a));
  a.packed = static_cast<const char*>(packed);
  a.state = static_cast<const char*>(latent_state);
  a.out = out;
  a.out_dim = out_dim;
  a.add_input = add_input;
  a.xyz = xyz;
  a.n_points = n_points;
  if (set_workspace(a, workspace, workspace_bytes, "nphm_mlp_eval_points: workspace too small or misaligned (nphm_mlp_eval_workspace_bytes)")) return -2;
  return launch_eval<0>(plan, a, n_points, n_rows, static_cast<hipStream_t>(stream), numerics);
}

size_t nphm_mlp_saved_bytes(int lat_dim, int hidden_dim, int nlayers, int out_dim, int n_rows, int64_t n_points) {
  Plan plan;
  if (!plan_of(lat_dim, hidden_dim, nlayers, out_dim, plan) || plan.variant != 0 || n_rows <= 0 || n_points <= 0) return 0;
  size_t tiles = 0;
  for (int l = 0; l < plan.n_linear - 1; ++l) tiles += plan.layer[l].n_tiles;
  const size_t wgs = size_t((n_points + 63) / 64);
  return size_t(n_rows) * wgs * tiles * 2 * 64 * 16 * sizeof(float);
}

int nphm_mlp_eval_points_saving(int lat_dim, int hidden_dim, int nlayers, int out_dim,
                                const void* packed, const void* latent_state,
                                const float* xyz, int n_rows, int64_t n_points, int add_input,
                                float* out, void* saved, int numerics, void* stream) {
  Plan plan;
  if (!plan_of(lat_dim, hidden_dim, nlayers, out_dim, plan)) return nphm_fail_msg("nphm_mlp_eval_points_saving: unsupported architecture");
  if (!packed || !latent_state || !xyz || !out || !saved) return nphm_fail_msg("nphm_mlp_eval_points_saving: null pointer");
  if (n_rows <= 0 || n_points <= 0) return nphm_fail_msg("nphm_mlp_eval_points_saving: empty input");
  nphm::mlp::EvalArgs a;
  memset(&a, 0, sizeof(a));
  a.packed = static_cast<const char*>(packed);
  a.state = static_cast<const char*>(latent_state);
  a.out = out;
  a.out_dim = out_dim;
  a.add_input = add_input;
  a.xyz = xyz;
  a.n_points = n_points;
  a.sig_out = static_cast<float*>(saved);
  if (!mlp_numerics_ok(numerics)) return nphm_fail_msg("nphm_mlp_eval_points_saving: unknown numerics format");
  return launch_eval<0, 3>(plan, a, n_points, n_rows, static_cast<hipStream_t>(stream), numerics);
}

int nphm_mlp_eval_points_jvp_saving(int lat_dim, int hidden_dim, int nlayers, int out_dim,
                                    const void* packed, const void* latent_state,
                                    const float* xyz, int n_rows, int64_t n_points, int add_input,
                                    float* out, void* saved, int numerics, int64_t point_base, int64_t point_count, int columns,
                                    float* jac_inverse, void* stream) {
  Plan plan;
  if (!plan_of(lat_dim, hidden_dim, nlayers, out_dim, plan)) return nphm_fail_msg("nphm_mlp_eval_points_jvp_saving: unsupported architecture");
  if (!packed || !latent_state || !xyz || !out || !saved) return nphm_fail_msg("nphm_mlp_eval_points_jvp_saving: null pointer");
  if (n_rows <= 0 || n_points <= 0) return nphm_fail_msg("nphm_mlp_eval_points_jvp_saving: empty input");
  nphm::mlp::EvalArgs a;
  memset(&a, 0, sizeof(a));
  a.packed = static_cast<const char*>(packed);
  a.state = static_cast<const char*>(latent_state);
  a.out = out;
  a.out_dim = out_dim;
  a.add_input = add_input;
  a.xyz = xyz;
  a.n_points = n_points;
  a.sig_out = static_cast<float*>(saved);
  if (jac_inverse && out_dim < 3) return nphm_fail_msg("nphm_mlp_eval_points_jvp_saving: the inverse Jacobian needs out_dim >= 3");
  a.jinv_out = jac_inverse;
  if (!mlp_numerics_ok(numerics)) return nphm_fail_msg("nphm_mlp_eval_points_jvp_saving: unknown numerics format");
  a.point_base = point_base;
  a.point_end = point_count > 0 ? point_base + point_count : (point_base == 0 ? 0 : n_points);
  return launch_eval<0, 4>(plan, a, n_points, n_rows, static_cast<hipStream_t>(stream), numerics, columns == 0 ? 64 : columns);
}

int nphm_mlp_eval_points_jvp(int lat_dim, int hidden_dim, int nlayers, int out_dim,
                             const void* packed, const void* latent_state,
                             const float* xyz, int n_rows, int64_t n_points, int add_input,
                             float* out, int numerics, int64_t point_base, int64_t point_count, int columns, float* jac_inverse,
                             void* stream) {
  Plan plan;
  if (!plan_of(lat_dim, hidden_dim, nlayers, out_dim, plan)) return nphm_fail_msg("nphm_mlp_eval_points_jvp: unsupported architecture");
  if (!packed || !latent_state || !xyz || !out) return nphm_fail_msg("nphm_mlp_eval_points_jvp: null pointer");
  if (n_rows <= 0 || n_points <= 0) return nphm_fail_msg("nphm_mlp_eval_points_jvp: empty input");
  nphm::mlp::EvalArgs a;
  memset(&a, 0, sizeof(a));
  a.packed = static_cast<const char*>(packed);
  a.state = static_cast<const char*>(latent_state);
  a.out = out;
  a.out_dim = out_dim;
  a.add_input = add_input;
  a.xyz = xyz;
  a.n_points = n_points;
  if (jac_inverse && out_dim < 3) return nphm_fail_msg("nphm_mlp_eval_points_jvp: the inverse Jacobian needs out_dim >= 3");
  a.jinv_out = jac_inverse;
  if (!mlp_numerics_ok(numerics)) return nphm_fail_msg("nphm_mlp_eval_points_jvp: unknown numerics format");
  a.point_base = point_base;
  a.point_end = point_count > 0 ? point_base + point_count : (point_base == 0 ? 0 : n_points);
  return launch_eval<0, 1>(plan, a, n_points, n_rows, static_cast<hipStream_t>(stream), numerics, columns == 0 ? 64 : columns);
}

int nphm_mlp_broyden(int lat_dim, int hidden_dim, int nlayers, int out_dim,
                     const void* packed, const void* latent_state,
                     const float* obs, const float* x_init, const float* jinv_init, int n_rows, int64_t n_points,
                     int max_steps, float cvg_thresh, float dvg_thresh, float eps,
                     float* x_out, float* diff_out, unsigned char* valid_out, int numerics, void* stream) {
  Plan plan;
  if (!plan_of(lat_dim, hidden_dim, nlayers, out_dim, plan) || out_dim < 3)
    return nphm_fail_msg("nphm_mlp_broyden: unsupported architecture (needs a 3-vector field)");
  if (!packed || !latent_state || !obs || !x_init || !jinv_init || !x_out || !diff_out || !valid_out)
    return nphm_fail_msg("nphm_mlp_broyden: null pointer");
  if (n_rows <= 0 || n_points <= 0 || max_steps < 0) return nphm_fail_msg("nphm_mlp_broyden: empty input");
  nphm::mlp::EvalArgs a;
  memset(&a, 0, sizeof(a));
  a.packed = static_cast<const char*>(packed);
  a.state = static_cast<const char*>(latent_state);
  a.out = x_out;
  a.out_dim = out_dim;
  a.xyz = x_init;
  a.n_points = n_points;
  a.obs = obs;
  a.jinv = jinv_init;
  a.diff_out = diff_out;
  a.valid_out = valid_out;
  a.max_steps = max_steps;
  a.cvg = cvg_thresh; a.dvg = dvg_thresh; a.eps = eps;
  if (!mlp_numerics_ok(numerics)) return nphm_fail_msg("nphm_mlp_broyden: unknown numerics format");
  return launch_eval<0, 2>(plan, a, n_points, n_rows, static_cast<hipStream_t>(stream), numerics);
}

int nphm_mlp_broyden_from(int lat_dim, int hidden_dim, int nlayers, int out_dim,
                          const void* packed, const void* latent_state,
                          const float* obs, const float* x_init, const float* jinv_init, const float* posed_init,
                          int64_t posed_stride, int n_rows, int64_t n_points,
                          int max_steps, float cvg_thresh, float dvg_thresh, float eps,
                          float* x_out, float* diff_out, unsigned char* valid_out, int numerics, void* stream) {
  Plan plan;
  if (!plan_of(lat_dim, hidden_dim, nlayers, out_dim, plan) || out_dim < 3)
    return nphm_fail_msg("nphm_mlp_broyden_from: unsupported architecture (needs a 3-vector field)");
  if (!packed || !latent_state || !obs || !x_init || !jinv_init || !posed_init || !x_out || !diff_out || !valid_out)
    return nphm_fail_msg("nphm_mlp_broyden_from: null pointer");
  if (n_rows <= 0 || n_points <= 0 || max_steps < 0 || posed_stride < 3) return nphm_fail_msg("nphm_mlp_broyden_from: bad sizes");
  nphm::mlp::EvalArgs a;
  memset(&a, 0, sizeof(a));
  a.packed = static_cast<const char*>(packed);
  a.state = static_cast<const char*>(latent_state);
  a.out = x_out;
  a.out_dim = out_dim;
  a.xyz = x_init;
  a.n_points = n_points;
  a.obs = obs;
  a.jinv = jinv_init;
  a.posed0 = posed_init;
  a.posed0_stride = posed_stride;
  a.diff_out = diff_out;
  a.valid_out = valid_out;
  a.max_steps = max_steps;
  a.cvg = cvg_thresh; a.dvg = dvg_thresh; a.eps = eps;
  if (!mlp_numerics_ok(numerics)) return nphm_fail_msg("nphm_mlp_broyden_from: unknown numerics format");
  return launch_eval<0, 2>(plan, a, n_points, n_rows, static_cast<hipStream_t>(stream), numerics);
}

int nphm_inverse3x3(const float* matrices, float* inverses, int64_t n, void* stream) {
  if (!matrices || !inverses) return nphm_fail_msg("nphm_inverse3x3: null pointer");
  if (n <= 0) return n == 0 ? 0 : nphm_fail_msg("nphm_inverse3x3: negative count");
  hipLaunchKernelGGL(nphm::mlp::inverse3x3_kernel, dim3(unsigned((n + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), matrices, inverses, n);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return nphm_fail("nphm_inverse3x3 launch", e);
  return 0;
}

int nphm_mlp_eval_grid_ws(int lat_dim, int hidden_dim, int nlayers, int out_dim,
                          const void* packed, const void* latent_state,
                          const float* axis_x, const float* axis_y, const float* axis_z,
                          int rx, int ry, int rz, int ix0, int ix1, int add_input, int numerics,
                          float* out, void* workspace, size_t workspace_bytes, void* stream);

int nphm_mlp_eval_grid(int lat_dim, int hidden_dim, int nlayers, int out_dim,
                       const void* packed, const void* latent_state,
                       const float* axis_x, const float* axis_y, const float* axis_z,
                       int rx, int ry, int rz, int ix0, int ix1, int add_input, int numerics,
                       float* out, void* stream) {
  return nphm_mlp_eval_grid_ws(lat_dim, hidden_dim, nlayers, out_dim, packed, latent_state, axis_x, axis_y, axis_z, rx, ry, rz,
                               ix0, ix1, add_input, numerics, out, nullptr, 0, stream);
}

int nphm_mlp_eval_grid_ws(int lat_dim, int hidden_dim, int nlayers, int out_dim,
                          const void* packed, const void* latent_state,
                          const float* axis_x, const float* axis_y, const float* axis_z,
                          int rx, int ry, int rz, int ix0, int ix1, int add_input, int numerics,
                          float* out, void* workspace, size_t workspace_bytes, void* stream) {
  Plan plan;
  if (!plan_of(lat_dim, hidden_dim, nlayers, out_dim, plan)) return nphm_fail_msg("nphm_mlp_eval_grid: unsupported architecture");
  if (!packed || !latent_state || !axis_x || !axis_y || !axis_z || !out) return nphm_fail_msg("nphm_mlp_eval_grid: null pointer");
  if (rx <= 0 || ry <= 0 || rz <= 0 || ix0 < 0 || ix1 > rx || ix0 >= ix1)
    return nphm_fail_msg("nphm_mlp_eval_grid: bad grid / slab bounds");
  if (!mlp_eval_numerics_ok(numerics)) return nphm_fail_msg("nphm_mlp_eval_grid: unknown numerics format");
  nphm::mlp::EvalArgs a;
  memset(&a, 0, sizeof(a));
  a.packed = static_cast<const char*>(packed);
  a.state = static_cast<const char*>(latent_state);
  a.out = out;
  a.out_dim = out_dim;
  a.add_input = add_input;
  a.ax = axis_x; a.ay = axis_y; a.az = axis_z;
  a.rx = rx; a.ry = ry; a.rz = rz; a.ix0 = ix0; a.ix1 = ix1;
  if (set_workspace(a, workspace, workspace_bytes, "nphm_mlp_eval_grid: workspace too small or misaligned (nphm_mlp_eval_workspace_bytes)")) return -2;
  return launch_eval<1>(plan, a, int64_t(ix1 - ix0) * ry * rz, 1, static_cast<hipStream_t>(stream), numerics);
}

}  // extern "C"

// mlp_layout.h — plan (layer table + HBM layout) of the fused dense skip-MLP kernel that serves
// the reference's DeepSDF (src/NPHM/models/deepSDF.py:6-89): the NPM global SDF
// (lat 512, hidden 1024, 8 layers, out 1) and the backbone of the forward-deformation network
// (lat 232, hidden 512, 6 layers, out 3; deepSDF.py:170-176).
//
// Network:  dims = [3 + lat] + [H]*nlayers + [out];  layer `skip = nlayers/2` receives
// [x | xyz | lat] / sqrt(2) and layer skip-1 therefore emits H - (3 + lat) features
// (deepSDF.py:29-42, :81-82); Softplus(beta = 100) after every layer but the last.
//
// With a latent that is constant along the point axis the latent columns of lin0 and of the skip
// layer collapse into per-row bias vectors, leaving per point
//   lin0: 3 -> H            (one "coordinate K-step", below)
//   lin_l: K_l -> N_l       split-bf16 MFMA GEMM over the activations of the previous layer
//   skip : + coordinate K-step with W[:, K:K+3] / sqrt2 and the folded latent bias
//   last : H -> out (<= 4)  K split over the 8 wavefronts of the workgroup
//
// Activation scaling: as in layout.h the kernels carry a' = k a with k = 100 / ln 2 so that
// Softplus(beta=100) is max(d',0) + log2(1 + 2^-|d'|).  Weights that multiply activations are
// unchanged (the last layer's are divided by k), additive terms (biases, coordinate columns) are
// multiplied by k (not for the last layer).
//
// Packed weights (bf16 hi | lo, MFMA A fragments of v_mfma_f32_32x32x16_bf16), per layer l >= 1:
//   [n_tile][k_step][hi|lo][lane 64][8]   k-slot 8*h + i of K-step 2*b + half  <->  input feature
//   32*b + feat(8*half + i, h),  feat(r, h) = (r & 3) + 8*(r >> 2) + 4*h  (the C/D register layout
//   of the producing layer, so a D tile is written to LDS as two 16-byte K chunks per lane).
//
// Per-latent state, per layer: [n_tile][lane 64][8] bf16 = the A fragment of the coordinate K-step
//   lanes h=0 : wh_x wh_y wh_z | wh_x wh_y wh_z | b_hi b_mid     (B: xh_x xh_y xh_z | xl_x xl_y xl_z | 1 1)
//   lanes h=1 : wl_x wl_y wl_z | b_lo | wh_x wh_y wh_z | 0       (B: xh_x xh_y xh_z | 1 | xll_x xll_y xll_z | 0)
// i.e. (xh + xl + xll)(wh + wl) minus the 2^-16 terms, plus a 3-term bf16 bias: it initialises the
// accumulators of every layer (layers without coordinate columns carry zero weights there).
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace nphm {
namespace mlp {

constexpr int MAX_LINEAR = 12;
constexpr int WAVES = 8;                 // wavefronts per workgroup
constexpr float SP_SCALE = 144.26950408889634f;   // k = 100 / ln 2
constexpr float SQRT2 = 1.41421356237309515f;

struct Config {
  int lat_dim, hidden, nlayers, out_dim;
};

struct Layer {
  int in_dim;       // row stride of the nn.Linear weight
  int out_dim;      // rows
  int k_act;        // leading columns that multiply activations of the previous layer
  int coord_col;    // first coordinate column, -1: none
  int lat_col;      // first latent column, -1: none
  int n_tiles;      // ceil(out_dim / 32)
  int k_steps;      // 2 * ceil(k_act / 32)
  float act_scale;  // on the activation columns (1, 1/sqrt2 for the skip layer, 1/k for the last)
  float in_scale;   // on coordinate / latent columns (1/sqrt2 for the skip layer)
  float add_scale;  // on everything additive (k, 1 for the last layer)
  uint32_t w_off;   // byte offset of the layer's A fragments in the packed buffer
  uint32_t c_off;   // byte offset of the layer's coordinate-step fragments in a state row
};

struct Plan {
  int n_linear;
  int variant;      // 0: 64 points x 512 features per workgroup, 1: 32 points x 1024 features
  Layer layer[MAX_LINEAR];
  size_t packed_bytes;       // A fragments of one operand format; the packed buffer = [bf16 | f16 | last-layer table]
  size_t state_row_bytes;
  size_t last_bytes;         // fp32 table of the LAST linear layer (see last_table_floats)
};

// The last linear layer (out_dim <= 4 rows) is applied in fp32 by the epilogue of the last hidden layer, straight from the
// activations in registers: per input tile n (32 features) and lane half h the 4 x 16 weights W[c][32 n + feat_local(r, h)]
// ([n][h][c][r] floats, zero-padded), then the 4 biases.
__host__ __device__ constexpr size_t last_table_floats(int k_tiles) { return size_t(k_tiles) * 2 * 4 * 16 + 4; }

__host__ __device__ constexpr int feat_local(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// Builds the plan; returns false if the architecture is outside what the kernel covers.
inline bool make_plan(const Config& c, Plan& p) {
  const int d_in = 3 + c.lat_dim;
  if (c.lat_dim < 0 || c.nlayers < 2 || c.nlayers + 1 > MAX_LINEAR) return false;
  if (c.out_dim < 1 || c.out_dim > 4) return false;
  if (c.hidden < 32 || c.hidden > 1024 || c.hidden - d_in < 1) return false;
  p.n_linear = c.nlayers + 1;
  p.variant = c.hidden <= 512 ? 0 : 1;
  const int skip = c.nlayers / 2;
  size_t w = 0, s = 0;
  int prev_out = 0;
  for (int l = 0; l < p.n_linear; ++l) {
    Layer& L = p.layer[l];
    const bool last = l == p.n_linear - 1;
    L.in_dim = l == 0 ? d_in : c.hidden;
    L.out_dim = last ? c.out_dim : (l + 1 == skip ? c.hidden - d_in : c.hidden);
    L.k_act = l == 0 ? 0 : prev_out;
    L.coord_col = l == 0 ? 0 : (l == skip ? prev_out : -1);
    L.lat_col = L.coord_col < 0 ? -1 : L.coord_col + 3;
    L.n_tiles = (L.out_dim + 31) / 32;
    L.k_steps = 2 * ((L.k_act + 31) / 32);
    L.act_scale = last ? 1.f / SP_SCALE : (l == skip ? 1.f / SQRT2 : 1.f);
    L.in_scale = l == skip ? 1.f / SQRT2 : 1.f;
    L.add_scale = last ? 1.f : SP_SCALE;
    L.w_off = uint32_t(w);
    L.c_off = uint32_t(s);
    w += size_t(L.n_tiles) * L.k_steps * 2 * 64 * 16;
    s += size_t(L.n_tiles) * 64 * 16;
    prev_out = L.out_dim;
    if (l == skip && L.in_dim != L.k_act + d_in) return false;
  }
  p.packed_bytes = w;
  p.state_row_bytes = s;
  p.last_bytes = (last_table_floats(p.layer[p.n_linear - 1].k_steps / 2) * sizeof(float) + 15) / 16 * 16;
  return w < (size_t(1) << 32);
}

}  // namespace mlp
}  // namespace nphm

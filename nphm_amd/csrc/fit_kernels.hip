// fit_kernels.hip — the small pieces of one latent-fitting step (src/NPHM/models/fitting.py:99-167 of the reference)
// that are NOT field evaluations, fused: in the reference (and in rounds 1-2 of this repo) they are ~150 elementwise /
// reduction / tiny-GEMM launches per step, half of the step's time once the fields run on their fused kernels.
//
//   fit_loss_kernel      clamped surface loss over the converged correspondences + every code regulariser + the weighted
//                        total (fitting.py:115-166): the loss terms, and in a second mode their gradients
//   root_bwd_kernel      implicit differentiation of the correspondence root (fitting.py:99-106): g_posed = -J^-T g_xc
//   latent_blocks_kernel bias gradients of lin0 / the skip layer of the identity field -> gradient of the latent row
//                        (what a batched GEMM over the 40 members' latent column blocks did)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "capi_common.h"
#include "layout.h"

namespace nphm {
namespace fit {

constexpr int N_TERMS = 6;     // surface, reg_expr, reg_global, reg_unobserved, reg_loc, symm_dist (the lambdas' order)

struct LossArgs {
  const float* sdf;            // [P]
  const unsigned char* valid;  // [P] or null (torch.bool)
  const float* thr;            // device scalar: clamp of the surface loss
  const float* lam;            // [6] device: loss weights in the order above
  const float* z_shape;        // [1344] identity code
  const float* z_expr;         // [n_obs, 200] expression codes (null: identity-only loop)
  const int64_t* obs_idx;      // [B] observation of every batch row (null with z_expr)
  int P, B, n_obs, expr_dim;
  const float* g_out;          // backward: device scalar dL/dloss (null = 1)
  float* row;                  // forward: [8] = 6 terms, total, number of valid correspondences
  float* g_sdf;                // backward: [P]
  float* g_shape;              // backward: [1344]
  float* g_expr;               // backward: [n_obs, 200]
  // replayed steps (nphm_fit_loss_with_gradients_logged): the row is ALSO stored at row_log[log_ctl[0] - 1][0 .. 7] - log_ctl[0] =
  // the ring launches completed (nphm_fit_inputs_ring of the same replay has counted itself already) - so that the loss trace of
  // a graph-replayed loop needs no copy launch between two replays
  float* row_log;
  const unsigned* log_ctl;
  int log_rows;
};

__device__ __forceinline__ float block_sum(float v, float* sh) {
  // sum over the 1024 threads of the block, result in every thread
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[w] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < int(blockDim.x >> 6); ++i) t += sh[i];
  return t;
}

// one workgroup.  MODE 0: the loss terms (row).  1: the gradients of the weighted total.  2: both (the step that calls
// loss.backward(seed) right behind the forward hands the seed over with it: one launch instead of two, same arithmetic).
template <int MODE>
__global__ __launch_bounds__(1024) void fit_loss_kernel(LossArgs a) {
  constexpr bool BWD = MODE != 0;
  __shared__ float sh[16];
  __shared__ float pair_norm[N_SYMM];
  const int t = threadIdx.x;
  const float thr = *a.thr;
  // ---- surface: mean |sdf| over the valid points below the clamp (fitting.py:115-132) ----
  float s = 0.f, c = 0.f, nv = 0.f;
  for (int i = t; i < a.P; i += blockDim.x) {
    const float l = fabsf(a.sdf[i]);
    const bool ok = a.valid ? a.valid[i] != 0 : true;
    const bool keep = ok && l < thr;
    s += keep ? l : 0.f;
    c += keep ? 1.f : 0.f;
    nv += ok ? 1.f : 0.f;
  }
  s = block_sum(s, sh);
  c = block_sum(c, sh);
  nv = block_sum(nv, sh);
  const float surface = s / c;                         // 0 / 0 = nan when nothing is kept, like the reference's empty mean
  // ---- regularisers of the identity code (fitting.py:139-166) ----
  const float* z = a.z_shape;
  float g2 = 0.f, l2 = 0.f, u2 = 0.f;
  for (int i = t; i < LAT_DIM; i += blockDim.x) {
    const float v = z[i] * z[i];
    if (i < LAT_GLOB) g2 += v; else l2 += v;
    const int k = (i - LAT_GLOB) / LAT_LOC;
    if (i >= LAT_GLOB && (k == 30 || k == 31 || k == 39)) u2 += v;
  }
  g2 = block_sum(g2, sh);
  l2 = block_sum(l2, sh);
  u2 = block_sum(u2, sh);
  // symmetric pairs: mean over the 16 pairs of |z_2i - z_2i+1|
  if (t < N_SYMM * 64) {
    const int p = t >> 6, j = t & 63;
    float d = 0.f;
    if (j < LAT_LOC) {
      const float x = z[LAT_GLOB + (2 * p) * LAT_LOC + j] - z[LAT_GLOB + (2 * p + 1) * LAT_LOC + j];
      d = x * x;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) d += __shfl_xor(d, o);
    if (j == 0) pair_norm[p] = sqrtf(d);
  }
  __syncthreads();
  float symm = 0.f;
  for (int p = 0; p < N_SYMM; ++p) symm += pair_norm[p];
  symm *= 1.f / N_SYMM;
  // expression codes of the batch rows: mean over the rows of |z_ex|^2
  float e2 = 0.f;
  if (a.z_expr)
    for (int i = t; i < a.B * a.expr_dim; i += blockDim.x) {
      const float v = a.z_expr[a.obs_idx[i / a.expr_dim] * a.expr_dim + i % a.expr_dim];
      e2 += v * v;
    }
  e2 = block_sum(e2, sh) / float(max(a.B, 1));
  const float terms[N_TERMS] = {surface, e2, g2, u2, l2, symm};
  if (MODE != 1) {
    if (t == 0) {
      float total = 0.f;
      for (int i = 0; i < N_TERMS; ++i) {
        a.row[i] = terms[i];
        // (a term that carries no weight stays out of the total, as in the PyTorch formulation, which never touches an
        // absent term: 0 x NaN / 0 x inf must not poison the loss)
        if ((i != 1 || a.z_expr) && a.lam[i] != 0.f) total += a.lam[i] * terms[i];
      }
      a.row[N_TERMS] = total;
      a.row[N_TERMS + 1] = nv;
      a.row[N_TERMS + 2] = total;       // a second copy: the caller's differentiable scalar next to the report row
      if (a.row_log) {
        const unsigned at = __hip_atomic_load(a.log_ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - 1u;
        if (at < unsigned(a.log_rows)) {
          float* q = a.row_log + size_t(at) * (N_TERMS + 2);
          for (int i = 0; i < N_TERMS; ++i) q[i] = terms[i];
          q[N_TERMS] = total;
          q[N_TERMS + 1] = nv;
        }
      }
    }
  }
  if (!BWD) return;
  const float go = a.g_out ? *a.g_out : 1.f;
  // d surface / d sdf_i = sign(sdf_i) keep_i / count
  const float ws = go * a.lam[0] / c;
  for (int i = t; i < a.P; i += blockDim.x) {
    const float v = a.sdf[i];
    const bool keep = (a.valid ? a.valid[i] != 0 : true) && fabsf(v) < thr;
    a.g_sdf[i] = keep ? (v > 0.f ? ws : v < 0.f ? -ws : 0.f) : 0.f;
  }
  // identity code: 2 z on the squared norms; unit vectors on the pair distances (0 where a pair coincides, like torch)
  for (int i = t; i < LAT_DIM; i += blockDim.x) {
    float g;
    if (i < LAT_GLOB) {
      g = 2.f * a.lam[2] * z[i];
    } else {
      const int k = (i - LAT_GLOB) / LAT_LOC, j = (i - LAT_GLOB) % LAT_LOC;
      g = 2.f * a.lam[4] * z[i];
      if (k == 30 || k == 31 || k == 39) g += 2.f * a.lam[3] * z[i];
      if (k < 2 * N_SYMM) {
        const int p = k >> 1;
        const float n = pair_norm[p];
        if (n > 0.f) {
          const float x = z[LAT_GLOB + (2 * p) * LAT_LOC + j] - z[LAT_GLOB + (2 * p + 1) * LAT_LOC + j];
          g += ((k & 1) ? -1.f : 1.f) * a.lam[5] * x / (n * N_SYMM);
        }
      }
    }
    a.g_shape[i] = go * g;
  }
  // expression codes: rows that were drawn more than once collect every draw
  if (a.z_expr) {
    const float we = go * a.lam[1] * 2.f / float(max(a.B, 1));
    for (int i = t; i < a.n_obs * a.expr_dim; i += blockDim.x) {
      const int o = i / a.expr_dim;
      int cnt = 0;
      for (int b = 0; b < a.B; ++b) cnt += a.obs_idx[b] == o;
      a.g_expr[i] = we * float(cnt) * a.z_expr[i];
    }
  }
}

// g_posed[p][i] = - sum_j Jinv[p][j][i] g_xc[p][j]: the gradient the implicit-function correction x_c = root - J^-1 (F - F.detach())
// sends to the posed points (fitting.py:99-106; the reference builds it with an einsum on a detached inverse Jacobian)
__global__ void root_bwd_kernel(const float* jinv, const float* g_xc, float* g_posed, int64_t n) {
  const int64_t p = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
  if (p >= n) return;
  const float* J = jinv + 9 * p;
  const float gx = g_xc[3 * p], gy = g_xc[3 * p + 1], gz = g_xc[3 * p + 2];
#pragma unroll
  for (int i = 0; i < 3; ++i) g_posed[3 * p + i] = -(J[i] * gx + J[3 + i] * gy + J[6 + i] * gz);
}

// sdf[p] = sum_k w[p][k] f[p][k] over the members the pruning rule kept (w is exactly 0 elsewhere and f holds nothing there:
// the member kernel writes listed pairs only, so f needs no zero-fill); fixed order, one thread per point
__global__ __launch_bounds__(256) void blend_members_kernel(const float* __restrict__ w, const float* __restrict__ f,
                                                             float* __restrict__ out, int64_t n) {
  const int64_t p = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const float4* w4 = reinterpret_cast<const float4*>(w + p * N_MEMBERS);
  const float4* f4 = reinterpret_cast<const float4*>(f + p * N_MEMBERS);
  float acc = 0.f;
#pragma unroll
  for (int q = 0; q < N_MEMBERS / 4; ++q) {
    const float4 a = w4[q], b = f4[q];
    acc += a.x != 0.f ? a.x * b.x : 0.f;
    acc += a.y != 0.f ? a.y * b.y : 0.f;
    acc += a.z != 0.f ? a.z * b.z : 0.f;
    acc += a.w != 0.f ? a.w * b.w : 0.f;
  }
  out[p] = acc;
}

// One Adam update of a small code tensor (torch.optim.Adam's defaults as fitting.py:47-48 builds it: no weight decay, no
// amsgrad) in ONE launch - exp_avg.lerp_(g, 1 - b1); exp_avg_sq.mul_(b2).addcmul_(g, g, 1 - b2);
// p.addcdiv_(exp_avg, sqrt(exp_avg_sq) / sqrt(1 - b2^t) + eps, -lr / (1 - b1^t)) - where the multi-tensor implementation
// runs eight launches per optimizer and step.  The bias corrections arrive as host scalars, like there.
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, int64_t n, float one_minus_b1, float b2,
                                                    float one_minus_b2, float step_size, float bc2_sqrt, float eps) {
#pragma clang fp contract(off)
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float gi = g[i];
  const float mi = m[i] + one_minus_b1 * (gi - m[i]);
  const float vi = v[i] * b2 + one_minus_b2 * gi * gi;
  m[i] = mi;
  v[i] = vi;
  const float denom = sqrtf(vi) / bc2_sqrt + eps;
  p[i] = p[i] + (-step_size) * (mi / denom);
}

// The same update for up to two code tensors in one launch, the six scalars of each read from DEVICE memory
// (scalars [2][6] = 1 - b1, b2, 1 - b2, step_size, bc2_sqrt, eps: they change with the step count and the schedule, and
// travel with the step's draw) - so that the optimizer steps of a fitting step sit INSIDE its replayed graph (eager launches
// behind every replay: two launches and the gap in front of them).
struct AdamPairArgs {
  float* p[2];
  const float* g[2];
  float* m[2];
  float* v[2];
  int64_t n[2];
  const float* scalars;
};
__global__ __launch_bounds__(256) void adam_pair_kernel(AdamPairArgs a) {
#pragma clang fp contract(off)
  int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  const int which = i < a.n[0] ? 0 : 1;
  if (which) i -= a.n[0];
  if (i >= a.n[which]) return;
  const float* sc = a.scalars + 6 * which;
  const float one_minus_b1 = sc[0], b2 = sc[1], one_minus_b2 = sc[2], step_size = sc[3], bc2_sqrt = sc[4], eps = sc[5];
  float* p = a.p[which];
  float* m = a.m[which];
  float* v = a.v[which];
  const float gi = a.g[which][i];
  const float mi = m[i] + one_minus_b1 * (gi - m[i]);
  const float vi = v[i] * b2 + one_minus_b2 * gi * gi;
  m[i] = mi;
  v[i] = vi;
  const float denom = sqrtf(vi) / bc2_sqrt + eps;
  p[i] = p[i] + (-step_size) * (mi / denom);
}

// backward of out[b] = table[idx[b]] (the expression codes of the drawn observations, fitting.py:83): g_table[r] = sum of
// g_out[b] over the draws b of row r, in draw order - one launch where the index_put(accumulate) of autograd sorts the
// indices first (eight launches for five rows)
__global__ __launch_bounds__(256) void gather_rows_bwd_kernel(const float* __restrict__ g, const int64_t* __restrict__ idx, int n_draws,
                                                               int n_rows, int width, float* __restrict__ out) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_rows * width) return;
  const int r = e / width, c = e % width;
  float acc = 0.f;
  for (int b = 0; b < n_draws; ++b)
    if (idx[b] == r) acc += g[size_t(b) * width + c];
  out[e] = acc;
}

// The inputs of one fitting step from the draw (fitting.py:61-85), one launch: the sampled points obs[b][r] =
// clouds[o_b][p_br] of the drawn observations, their expression codes z_ex[b] = table[o_b] and the conditioning rows
// glob_cond[b] = [z_shape | z_ex[b]] - an advanced-indexing gather, an index_select and a cat in the PyTorch formulation
// (four launches).  drawn = [B observation indices | B x n point indices] (int64, _ObservationSampler.draw).
struct InputsArgs {
  const int64_t* drawn;
  const float* clouds;      // [n_obs][P][C]
  const float* z_shape;     // [L]
  const float* table;       // [n_obs][E]
  int B, n, P, C, L, E;
  float* obs;               // [B][n][C]
  float* z_ex;              // [B][E]
  float* glob_cond;         // [B][L + E]
  // ring mode (a step replayed from a hipGraph; nphm_fit_inputs_ring): the draw is read straight from a ring of draws in
  // pinned HOST memory - slot ctl[0] % ring_slots, ctl[0] = the number of ring launches completed so far - and copied to
  // `drawn_out` for the later launches of the step; the last workgroup to arrive counts the launch.  No copy engine, no
  // stream-ordered upload in front of the replay: the host fills slots ahead of the device.
  const int64_t* ring;
  int ring_slots;
  int64_t slot_stride, n_total;
  int64_t* drawn_out;
  unsigned* ctl;            // [0] launches completed, [1] workgroups of the current launch that have read [0]
};
constexpr int FIT_RING_MAX_ROWS = 32;
__global__ __launch_bounds__(256) void fit_inputs_kernel(InputsArgs a) {
  __shared__ int64_t staged[FIT_RING_MAX_ROWS + 256 + 2];   // ring mode: [B observation indices | the point indices this workgroup gathers]
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  const int n_br = a.B * a.n;
  int br0 = 0;
  unsigned seq = 0;
  if (a.ring) {
    // every value of the slot crosses the bus once per workgroup that needs it (a PCIe read per 64 bytes: ~700 for the draw;
    // the first form, every thread reading its own two indices with system-scope loads, took 51 us for 45 000 of them), and is
    // written through to drawn_out by the workgroup that staged it
    seq = __hip_atomic_load(a.ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const volatile int64_t* src = a.ring + int64_t(seq % unsigned(a.ring_slots)) * a.slot_stride;
    const int e0 = blockIdx.x * blockDim.x;
    br0 = min(e0 / a.C, n_br);
    const int br1 = min((e0 + int(blockDim.x) - 1) / a.C + 1, n_br);
    for (int i = threadIdx.x; i < a.B + (br1 - br0); i += blockDim.x) {
      const int64_t at = i < a.B ? i : a.B + br0 + (i - a.B);
      const int64_t v = src[at];
      staged[i] = v;
      a.drawn_out[at] = v;
    }
    if (blockIdx.x == 0)
      for (int64_t at = a.B + n_br + threadIdx.x; at < a.n_total; at += blockDim.x) a.drawn_out[at] = src[at];   // what rides behind the draw
    __syncthreads();
  }
  if (e < n_br * a.C) {
    const int c = e % a.C, br = e / a.C, b = br / a.n;
    const int64_t o = a.ring ? staged[b] : a.drawn[b], pi = a.ring ? staged[a.B + br - br0] : a.drawn[a.B + br];
    a.obs[e] = a.clouds[(o * a.P + pi) * a.C + c];
  }
  const int W = a.L + a.E;
  if (e < a.B * W) {
    const int b = e / W, i = e % W;
    if (i < a.L) {
      a.glob_cond[e] = a.z_shape[i];
    } else {
      const float v = a.table[(a.ring ? staged[b] : a.drawn[b]) * a.E + (i - a.L)];
      a.glob_cond[e] = v;
      a.z_ex[b * a.E + (i - a.L)] = v;
    }
  }
  if (a.ring) {
    if (threadIdx.x == 0) {
      // (this workgroup's loads of the slot returned in front of the barrier above)
      if (atomicAdd(a.ctl + 1, 1u) == gridDim.x - 1) {                   // the last one to have read ctl[0]
        __hip_atomic_store(a.ctl + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(a.ctl, seq + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
}
// its backward: g_table[r] = sum over the draws b of row r, in draw order, of g_z_ex[b] (+ the expression columns of
// g_glob_cond[b]); g_shape = sum over b of the identity columns of g_glob_cond[b].  Either gradient may be absent (null);
// g_z_ex rows are `zex_stride` floats apart (a column slice of the conditioning's gradient: no contiguous copy first).
// The step's uses of the two codes get ALIASES of them from the forward (fitting.py: the identity code feeds the anchor
// head, the identity field, the regularisers and the compressor; the expression codes the regulariser and the gather): their
// gradients (shape_parts [4] / table_part, each null when that use produced none) are summed here, in this fixed order,
// instead of by one elementwise add launch per extra use.
struct InputsBwdArgs {
  const float* g_z_ex;
  int64_t zex_stride;
  const float* g_glob;
  const int64_t* idx;
  int B, n_obs, L, E;
  const float* shape_parts[4];
  const float* table_part;
  float* g_table;
  float* g_shape;
};
__global__ __launch_bounds__(256) void fit_inputs_bwd_kernel(InputsBwdArgs a) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  const int W = a.L + a.E;
  if (e < a.n_obs * a.E) {
    const int r = e / a.E, c = e % a.E;
    float acc = a.table_part ? a.table_part[e] : 0.f;
    for (int b = 0; b < a.B; ++b)
      if (a.idx[b] == r) {
        if (a.g_z_ex) acc += a.g_z_ex[int64_t(b) * a.zex_stride + c];
        if (a.g_glob) acc += a.g_glob[size_t(b) * W + a.L + c];
      }
    a.g_table[e] = acc;
  }
  if (a.g_shape && e < a.L) {
    float acc = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (a.shape_parts[q]) acc += a.shape_parts[q][e];
    if (a.g_glob)
      for (int b = 0; b < a.B; ++b) acc += a.g_glob[size_t(b) * W + e];
    a.g_shape[e] = acc;
  }
}

// batched 3x3 inverse (adjugate) of matrices addressed by strides - element (i, j) of matrix p at
// in[p * sp + i * si + j * sj] - so that a transposed / sliced view (the Jacobian block of the value+Jacobian output
// [.., 4, out_dim]) needs no contiguous copy first; out row-major [n][3][3]
__global__ __launch_bounds__(256) void inverse3x3_strided_kernel(const float* __restrict__ in, int64_t sp, int64_t si, int64_t sj,
                                                                  float* __restrict__ out, int64_t n) {
  const int64_t p = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (p >= n) return;
  float a[9];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) a[3 * i + j] = in[p * sp + i * si + j * sj];
  const float c00 = a[4] * a[8] - a[5] * a[7], c01 = a[5] * a[6] - a[3] * a[8], c02 = a[3] * a[7] - a[4] * a[6];
  const float det = a[0] * c00 + a[1] * c01 + a[2] * c02;
  const float r = 1.f / det;
  float* o = out + p * 9;
  o[0] = c00 * r; o[1] = (a[2] * a[7] - a[1] * a[8]) * r; o[2] = (a[1] * a[5] - a[2] * a[4]) * r;
  o[3] = c01 * r; o[4] = (a[0] * a[8] - a[2] * a[6]) * r; o[5] = (a[2] * a[3] - a[0] * a[5]) * r;
  o[6] = c02 * r; o[7] = (a[1] * a[6] - a[0] * a[7]) * r; o[8] = (a[0] * a[4] - a[1] * a[3]) * r;
}

// d L / d cond [R][lat] of the dense skip-MLP from the bias gradients of lin0 and of the skip layer
// (nphm_mlp_backward_cond's per-slot sums [R][n_slots][2][H], added here per row in slot order: fixed order, no atomics):
// g0 W0[:, off0:] + gs Ws[:, offs:] / sqrt2 - two [R,H]x[H,lat] products on 5 rows, for which the library launched two GEMMs,
// a divide and an add.  Block = 64 columns x 16 slices of H, slices combined through LDS.
constexpr int CG_PARTS = 16, CG_HMAX = 512;
__global__ __launch_bounds__(64 * CG_PARTS) void cond_grad_kernel(const float* __restrict__ parts, int n_slots, int H,
                                                                   const float* __restrict__ w0, int ld0, int off0,
                                                                   const float* __restrict__ ws, int lds_, int offs, int lat,
                                                                   float* __restrict__ out) {
  __shared__ float part[CG_PARTS][64];
  __shared__ float gsum[2][CG_HMAX];
  const int j = blockIdx.x * 64 + (threadIdx.x & 63), s = threadIdx.x >> 6, r = blockIdx.y;
  for (int e = threadIdx.x; e < 2 * H; e += blockDim.x) {         // (which, h) = (e / H, e % H): the row's slots in order
    const float* src = parts + size_t(r) * n_slots * 2 * H + e;
    float t = 0.f;
    for (int q = 0; q < n_slots; q += 8) {                         // eight loads in flight, added in slot order
      float v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = q + i < n_slots ? src[size_t(q + i) * 2 * H] : 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) t += v[i];
    }
    gsum[e / H][e % H] = t;
  }
  __syncthreads();
  const float* g0 = gsum[0];
  const float* gs = gsum[1];
  float acc0 = 0.f, acc1 = 0.f;
  if (j < lat)
    for (int h = s; h < H; h += CG_PARTS) {
      acc0 = fmaf(g0[h], w0[size_t(h) * ld0 + off0 + j], acc0);
      acc1 = fmaf(gs[h], ws[size_t(h) * lds_ + offs + j], acc1);
    }
  part[s][threadIdx.x & 63] = acc0 + acc1 * 0.70710678118654752440f;
  __syncthreads();
  if (s == 0 && j < lat) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < CG_PARTS; ++q) t += part[q][threadIdx.x];
    out[size_t(r) * lat + j] = t;
  }
}

// g_lat[b][:] from the bias gradients gb0 / gb2 [B][40][200] of lin0 and of the skip layer: the folded bias of member k is
// W0[set(k)][:, 3:] cond_k + b (lin0) and W2[set(k)][:, 104:] cond_k / sqrt2 + b (skip layer), cond_k = [z_glob | z_k]
// (EnsembledDeepSDF.py:247-255), so d/dcond_k = gb0_k W0_lat + gb2_k W2_lat / sqrt2.  Two launches, bitwise reproducible:
//   latent_blocks_kernel  grid (40, B), 96 columns x 8 slices of the 200 features per workgroup, the slices combined through
//                         LDS in a fixed order; member k's 32 local columns go straight to g_lat, its share of the 64 GLOBAL
//                         columns to scratch[b][k][:];
//   latent_global_kernel  g_lat[b][j] = sum over the members of scratch[b][k][j], k ascending.
// (Round 3: a thread per column walked the 200 features alone and the members added their global share with float atomics
// behind a zero-fill launch - arbitrary order, 16.8 us.  One workgroup owning all global columns reads 4 MB through ONE CU:
// 100 us - the reduction over the members has to stay spread over the chip.)
constexpr int LB_SLICES = 8;
__global__ __launch_bounds__(LAT_COND * LB_SLICES) void latent_blocks_kernel(const float* __restrict__ w0, const float* __restrict__ w2,
                                                                             const float* __restrict__ gb0, const float* __restrict__ gb2,
                                                                             float* __restrict__ g_lat, float* __restrict__ scratch) {
  __shared__ float part[LB_SLICES][LAT_COND];
  const int k = blockIdx.x, b = blockIdx.y, j = threadIdx.x % LAT_COND, sl = threadIdx.x / LAT_COND;   // j: conditioning column 0..95
  const int s = member_set(k);
  const float* g0 = gb0 + (size_t(b) * N_MEMBERS + k) * HID;
  const float* g2 = gb2 + (size_t(b) * N_MEMBERS + k) * HID;
  const float* W0 = w0 + size_t(s) * HID * D_IN + 3 + j;               // [200][99]: column 3 + j
  const float* W2 = w2 + size_t(s) * HID * HID + L2_IN + j;            // [200][200]: column 104 + j
  float acc0 = 0.f, acc2 = 0.f;
  constexpr int PER = HID / LB_SLICES;                                  // 25 features per slice, all loads independent
  static_assert(HID % LB_SLICES == 0, "whole slices");
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int f = sl * PER + i;
    acc0 = fmaf(g0[f], W0[size_t(f) * D_IN], acc0);
    acc2 = fmaf(g2[f], W2[size_t(f) * HID], acc2);
  }
  part[sl][j] = acc0 + acc2 / INV_SQRT2_DIV;
  __syncthreads();
  if (sl == 0) {
    float v = 0.f;
#pragma unroll
    for (int q = 0; q < LB_SLICES; ++q) v += part[q][j];
    if (j < LAT_GLOB) scratch[(size_t(b) * N_MEMBERS + k) * LAT_GLOB + j] = v;
    else g_lat[size_t(b) * LAT_DIM + LAT_GLOB + k * LAT_LOC + (j - LAT_GLOB)] = v;
  }
}

__global__ __launch_bounds__(LAT_GLOB) void latent_global_kernel(const float* __restrict__ scratch, float* __restrict__ g_lat) {
  const int b = blockIdx.x, j = threadIdx.x;
  const float* p = scratch + size_t(b) * N_MEMBERS * LAT_GLOB + j;
  float v = 0.f;
#pragma unroll
  for (int k = 0; k < N_MEMBERS; ++k) v += p[k * LAT_GLOB];
  g_lat[size_t(b) * LAT_DIM + j] = v;
}

// ---- small dense heads with frozen weights: mlp_pos (64 -> 256 -> 256 -> 117, ReLU; EnsembledDeepSDF.py:194-200) and the
// compressor of the deformation field (1461 -> 32; deepSDF.py:212-223) on a handful of rows: one workgroup per row, the
// whole chain in one launch (the PyTorch formulation: a rocBLAS GEMM + bias + ReLU launch per layer and direction).
struct HeadArgs {
  const float* w[3];
  const float* b[3];
  int dims[4];                 // in, (hidden ...), out
  int n_layers;
  const float* x;              // [rows, x_stride >= dims[0]]: the first dims[0] columns of every row are the input
  int x_stride;
  const float* y_add;          // or null: [dims[n_layers]] added to every output row (e.g. the mean anchors)
  float* y;                    // [rows, dims[n_layers]]
  float* hidden;               // [rows, dims[1] + dims[2]] post-ReLU activations (saved for the backward pass)
  const float* g_y;            // backward: [rows, out]
  const float* g_y2;           // or null: a second gradient of the same output (another use of it), added while loading
  float* g_x;                  // backward: [rows, x_stride] - columns beyond dims[0] are written as zeros
  // the compressor inside a conditioning row (nphm_compress_condition): the input row is [x[0 .. x_split) | x2[0 .. dims[0] - x_split)]
  // (identity code | anchors: no cat launch), the ONE output row goes to the first `out` columns of y_rep rows of y (y_stride apart),
  // tail [y_rep][tail_w] behind it; backward: g_y = the sum over g_rows rows (g_stride apart) of their first `out` columns, g_x / g_x2
  // receive the two parts of the input's gradient.  (All zero / null: the plain head.)
  const float* x2;
  int x_split;
  int y_rep, y_stride;
  const float* tail;
  int tail_w;
  int g_rows, g_stride;
  float* g_x2;
};
constexpr int HEAD_MAX = 1536;   // widest layer input / output held in LDS

// Both kernels are ONE workgroup per row on a handful of rows: latency-bound, and the latency is the round trips to the
// weights (447 KB for mlp_pos, rarely in L2 between two replays of a fitting step).  So every layer is ONE round trip: a
// wavefront's task holds all of its loads in flight at once (64 per lane), the partial sums of the tasks meet in LDS in a
// fixed order (deterministic).  Round 3 before this: 8-16 loads in flight, 14 round trips forward (23 us), more backward (34 us).
constexpr int HEAD_U = 16;        // forward task: 16 output features x 256 input features (lanes x 4)
constexpr int HEAD_BO = 32;       // backward task: 64 input features (the lanes) x 32 output features
constexpr int HEAD_PART = 8192;   // floats of partial sums: ceil(din / 256) * dout (forward), ceil(dout / 32) * din (backward)

__global__ __launch_bounds__(1024) void head_fwd_kernel(HeadArgs a) {
  __shared__ float cur[HEAD_MAX], part[HEAD_PART];
  const int row = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6, nw = blockDim.x >> 6;
  for (int i = t; i < a.dims[0]; i += blockDim.x)
    cur[i] = (a.x2 && i >= a.x_split) ? a.x2[i - a.x_split] : a.x[size_t(row) * a.x_stride + i];
  __syncthreads();
  if (a.y_rep)                                        // the rest of the conditioning rows: the expression codes
    for (int e = t; e < a.y_rep * a.tail_w; e += blockDim.x)
      a.y[size_t(e / a.tail_w) * a.y_stride + a.dims[a.n_layers] + e % a.tail_w] = a.tail[e];
  int hoff = 0;
  const int hstride = a.dims[1] + (a.n_layers > 2 ? a.dims[2] : 0);
  for (int l = 0; l < a.n_layers; ++l) {
    const int din = a.dims[l], dout = a.dims[l + 1];
    const bool last = l == a.n_layers - 1;
    const float* __restrict__ W = a.w[l];
    const int n_og = (dout + HEAD_U - 1) / HEAD_U, n_ic = (din + 255) / 256;
    for (int task = wave; task < n_og * n_ic; task += nw) {
      const int og = task % n_og, ic = task / n_og;
      float x[4], acc[HEAD_U];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int i = ic * 256 + c * 64 + lane;
        x[c] = i < din ? cur[i] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < HEAD_U; ++u) {
        const int o = min(og * HEAD_U + u, dout - 1);
        float w[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int i = ic * 256 + c * 64 + lane;
          w[c] = i < din ? W[size_t(o) * din + i] : 0.f;
        }
        acc[u] = fmaf(w[3], x[3], fmaf(w[2], x[2], fmaf(w[1], x[1], w[0] * x[0])));
      }
#pragma unroll
      for (int u = 0; u < HEAD_U; ++u) {
        float v = acc[u];
#pragma unroll
        for (int m = 32; m > 0; m >>= 1) v += __shfl_xor(v, m);
        if (lane == 0 && og * HEAD_U + u < dout) part[ic * dout + og * HEAD_U + u] = v;
      }
    }
    __syncthreads();
    for (int i = t; i < dout; i += blockDim.x) {
      float v = a.b[l][i];
      for (int ic = 0; ic < n_ic; ++ic) v += part[ic * dout + i];
      if (!last) v = fmaxf(v, 0.f);
      else if (a.y_add) v += a.y_add[i];
      cur[i] = v;
      if (last && a.y_rep) {
        for (int b = 0; b < a.y_rep; ++b) a.y[size_t(b) * a.y_stride + i] = v;
      } else if (last) {
        a.y[size_t(row) * dout + i] = v;
      } else if (a.hidden) {
        a.hidden[size_t(row) * hstride + hoff + i] = v;
      }
    }
    if (!last) hoff += dout;
    __syncthreads();
  }
}

// backward w.r.t. the input only (the weights are constants here): g_in = W^T (g_out * relu').  Task = 64 input features
// (the lanes: coalesced weight rows) x 32 output features (32 loads in flight per lane - 64 spill at 128 VGPRs -, no
// cross-lane step).
__global__ __launch_bounds__(1024) void head_bwd_kernel(HeadArgs a) {
  __shared__ float cur[HEAD_MAX], part[HEAD_PART];
  const int row = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6, nw = blockDim.x >> 6;
  const int dout_last = a.dims[a.n_layers];
  for (int i = t; i < dout_last; i += blockDim.x) {
    if (a.g_rows) {
      float v = 0.f;
      for (int b = 0; b < a.g_rows; ++b) v += a.g_y[size_t(b) * a.g_stride + i];       // (row order: deterministic)
      cur[i] = v;
    } else {
      cur[i] = a.g_y2 ? a.g_y[size_t(row) * dout_last + i] + a.g_y2[size_t(row) * dout_last + i] : a.g_y[size_t(row) * dout_last + i];
    }
  }
  __syncthreads();
  const int hstride = a.dims[1] + (a.n_layers > 2 ? a.dims[2] : 0);
  for (int l = a.n_layers - 1; l >= 0; --l) {
    const int din = a.dims[l], dout = a.dims[l + 1];
    int hoff = 0;
    for (int q = 1; q < l; ++q) hoff += a.dims[q];          // offset of this layer's INPUT activation (layer l - 1's output)
    const float* __restrict__ W = a.w[l];
    const int n_ib = (din + 63) / 64, n_oc = (dout + HEAD_BO - 1) / HEAD_BO;
    for (int task = wave; task < n_ib * n_oc; task += nw) {
      const int ib = task % n_ib, oc = task / n_ib;
      const int i = ib * 64 + lane;
      const int ii = min(i, din - 1);
      float w[HEAD_BO];
#pragma unroll
      for (int k = 0; k < HEAD_BO; ++k) {
        const int o = min(oc * HEAD_BO + k, dout - 1);
        w[k] = W[size_t(o) * din + ii];
      }
      float acc = 0.f;
#pragma unroll
      for (int k = 0; k < HEAD_BO; ++k) acc = fmaf(w[k], oc * HEAD_BO + k < dout ? cur[oc * HEAD_BO + k] : 0.f, acc);
      if (i < din) part[oc * din + i] = acc;
    }
    __syncthreads();
    float keep[2];                                    // din <= HEAD_MAX = 1536 < 2 * 1024
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int i = t + q * 1024;
      float v = 0.f;
      if (i < din) {
        for (int oc = 0; oc < n_oc; ++oc) v += part[oc * din + i];
        if (l > 0) v = a.hidden[size_t(row) * hstride + hoff + i] > 0.f ? v : 0.f;       // ReLU of the previous layer
        if (l == 0) {
          if (a.g_x2 && i >= a.x_split) a.g_x2[i - a.x_split] = v;
          else a.g_x[size_t(row) * a.x_stride + i] = v;
        }
      }
      keep[q] = v;
    }
    if (l == 0 && !a.g_x2)
      for (int i = din + t; i < a.x_stride; i += blockDim.x) a.g_x[size_t(row) * a.x_stride + i] = 0.f;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 2; ++q)
      if (t + q * 1024 < din) cur[t + q * 1024] = keep[q];
    __syncthreads();
  }
}


}  // namespace fit
}  // namespace nphm

extern "C" {

int nphm_fit_loss(const float* sdf, const unsigned char* valid, int64_t n_points, const float* thr, const float* lam,
                  const float* z_shape, const float* z_expr, const int64_t* obs_idx, int n_rows, int n_obs, int expr_dim,
                  float* row, void* stream) {
  if (!sdf || !thr || !lam || !z_shape || !row || n_points <= 0) return nphm_fail_msg("nphm_fit_loss: bad arguments");
  if (z_expr && (!obs_idx || n_rows <= 0 || n_obs <= 0 || expr_dim <= 0)) return nphm_fail_msg("nphm_fit_loss: bad expression-code arguments");
  nphm::fit::LossArgs a{sdf, valid, thr, lam, z_shape, z_expr, obs_idx, int(n_points), n_rows, n_obs, expr_dim, nullptr, row,
                        nullptr, nullptr, nullptr};
  hipLaunchKernelGGL(nphm::fit::fit_loss_kernel<0>, dim3(1), dim3(1024), 0, static_cast<hipStream_t>(stream), a);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : nphm_fail("nphm_fit_loss launch", e);
}

int nphm_fit_loss_backward(const float* sdf, const unsigned char* valid, int64_t n_points, const float* thr, const float* lam,
                           const float* z_shape, const float* z_expr, const int64_t* obs_idx, int n_rows, int n_obs, int expr_dim,
                           const float* g_out, float* g_sdf, float* g_shape, float* g_expr, void* stream) {
  if (!sdf || !thr || !lam || !z_shape || !g_sdf || !g_shape || n_points <= 0) return nphm_fail_msg("nphm_fit_loss_backward: bad arguments");
  if (z_expr && (!obs_idx || !g_expr || n_rows <= 0 || n_obs <= 0 || expr_dim <= 0))
    return nphm_fail_msg("nphm_fit_loss_backward: bad expression-code arguments");
  nphm::fit::LossArgs a{sdf, valid, thr, lam, z_shape, z_expr, obs_idx, int(n_points), n_rows, n_obs, expr_dim, g_out, nullptr,
                        g_sdf, g_shape, g_expr};
  hipLaunchKernelGGL(nphm::fit::fit_loss_kernel<1>, dim3(1), dim3(1024), 0, static_cast<hipStream_t>(stream), a);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : nphm_fail("nphm_fit_loss_backward launch", e);
}

int nphm_fit_loss_with_gradients_logged(const float* sdf, const unsigned char* valid, int64_t n_points, const float* thr, const float* lam,
                                        const float* z_shape, const float* z_expr, const int64_t* obs_idx, int n_rows, int n_obs, int expr_dim,
                                        const float* g_out, float* row, float* g_sdf, float* g_shape, float* g_expr,
                                        float* row_log, const unsigned* log_control, int log_rows, void* stream) {
  if (!sdf || !thr || !lam || !z_shape || !row || !g_sdf || !g_shape || n_points <= 0) return nphm_fail_msg("nphm_fit_loss_with_gradients: bad arguments");
  if (z_expr && (!obs_idx || !g_expr || n_rows <= 0 || n_obs <= 0 || expr_dim <= 0))
    return nphm_fail_msg("nphm_fit_loss_with_gradients: bad expression-code arguments");
  if (row_log && (!log_control || log_rows <= 0)) return nphm_fail_msg("nphm_fit_loss_with_gradients_logged: bad log arguments");
  nphm::fit::LossArgs a{sdf, valid, thr, lam, z_shape, z_expr, obs_idx, int(n_points), n_rows, n_obs, expr_dim, g_out, row,
                        g_sdf, g_shape, g_expr, row_log, log_control, log_rows};
  hipLaunchKernelGGL(nphm::fit::fit_loss_kernel<2>, dim3(1), dim3(1024), 0, static_cast<hipStream_t>(stream), a);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : nphm_fail("nphm_fit_loss_with_gradients launch", e);
}

int nphm_fit_loss_with_gradients(const float* sdf, const unsigned char* valid, int64_t n_points, const float* thr, const float* lam,
                                 const float* z_shape, const float* z_expr, const int64_t* obs_idx, int n_rows, int n_obs, int expr_dim,
                                 const float* g_out, float* row, float* g_sdf, float* g_shape, float* g_expr, void* stream) {
  return nphm_fit_loss_with_gradients_logged(sdf, valid, n_points, thr, lam, z_shape, z_expr, obs_idx, n_rows, n_obs, expr_dim, g_out, row,
                                             g_sdf, g_shape, g_expr, nullptr, nullptr, 0, stream);
}

int nphm_fit_root_backward(const float* jac_inverse, const float* g_xc, float* g_posed, int64_t n, void* stream) {
  if (!jac_inverse || !g_xc || !g_posed) return nphm_fail_msg("nphm_fit_root_backward: null pointer");
  if (n <= 0) return n == 0 ? 0 : nphm_fail_msg("nphm_fit_root_backward: negative count");
  hipLaunchKernelGGL(nphm::fit::root_bwd_kernel, dim3(unsigned((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     jac_inverse, g_xc, g_posed, n);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : nphm_fail("nphm_fit_root_backward launch", e);
}

static int head_args(nphm::fit::HeadArgs& a, const float* const w[3], const float* const b[3], const int dims[4], int n_layers,
                     const char* who) {
  if (n_layers < 1 || n_layers > 3) return nphm_fail_msg(who);
  for (int l = 0; l <= n_layers; ++l)
    if (dims[l] <= 0 || dims[l] > nphm::fit::HEAD_MAX) return nphm_fail_msg(who);
  for (int l = 0; l < n_layers; ++l)        // the partial sums of a layer's tasks live in LDS
    if (((dims[l] + 255) / 256) * dims[l + 1] > nphm::fit::HEAD_PART || ((dims[l + 1] + nphm::fit::HEAD_BO - 1) / nphm::fit::HEAD_BO) * dims[l] > nphm::fit::HEAD_PART)
      return nphm_fail_msg(who);
  for (int l = 0; l < 3; ++l) { a.w[l] = l < n_layers ? w[l] : nullptr; a.b[l] = l < n_layers ? b[l] : nullptr; }
  for (int l = 0; l < 4; ++l) a.dims[l] = l <= n_layers ? dims[l] : 0;
  a.n_layers = n_layers;
  for (int l = 0; l < n_layers; ++l) if (!a.w[l] || !a.b[l]) return nphm_fail_msg(who);
  return 0;
}

int nphm_head_forward(const float* const weight[3], const float* const bias[3], const int dims[4], int n_layers, const float* x,
                      int x_stride, const float* y_add, int n_rows, float* y, float* hidden, void* stream) {
  nphm::fit::HeadArgs a{};
  if (!x || !y || n_rows <= 0 || (n_layers > 1 && !hidden)) return nphm_fail_msg("nphm_head_forward: bad arguments");
  if (head_args(a, weight, bias, dims, n_layers, "nphm_head_forward: unsupported head (1..3 layers, widths <= 1536)")) return -2;
  if (x_stride < dims[0]) return nphm_fail_msg("nphm_head_forward: x_stride < input width");
  a.x = x; a.x_stride = x_stride; a.y_add = y_add; a.y = y; a.hidden = hidden;
  hipLaunchKernelGGL(nphm::fit::head_fwd_kernel, dim3(n_rows), dim3(1024), 0, static_cast<hipStream_t>(stream), a);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : nphm_fail("nphm_head_forward launch", e);
}

int nphm_head_backward(const float* const weight[3], const float* const bias[3], const int dims[4], int n_layers, const float* hidden,
                       const float* g_y, const float* g_y_other, int n_rows, float* g_x, int x_stride, void* stream) {
  nphm::fit::HeadArgs a{};
  if (!g_y || !g_x || n_rows <= 0 || (n_layers > 1 && !hidden)) return nphm_fail_msg("nphm_head_backward: bad arguments");
  if (head_args(a, weight, bias, dims, n_layers, "nphm_head_backward: unsupported head (1..3 layers, widths <= 1536)")) return -2;
  if (x_stride < dims[0]) return nphm_fail_msg("nphm_head_backward: x_stride < input width");
  a.hidden = const_cast<float*>(hidden); a.g_y = g_y; a.g_y2 = g_y_other; a.g_x = g_x; a.x_stride = x_stride;
  hipLaunchKernelGGL(nphm::fit::head_bwd_kernel, dim3(n_rows), dim3(1024), 0, static_cast<hipStream_t>(stream), a);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : nphm_fail("nphm_head_backward launch", e);
}

int nphm_compress_condition(const float* weight, const float* bias, const float* code, int code_dim, const float* anchors, int anchors_dim,
                            int out_dim, const float* z_ex, int n_rows, int expr_dim, float* cond, void* stream) {
  nphm::fit::HeadArgs a{};
  if (!weight || !bias || !code || !anchors || !z_ex || !cond || n_rows <= 0 || expr_dim <= 0 || code_dim <= 0 || anchors_dim <= 0)
    return nphm_fail_msg("nphm_compress_condition: bad arguments");
  const float* w[3] = {weight, nullptr, nullptr};
  const float* b[3] = {bias, nullptr, nullptr};
  const int dims[4] = {code_dim + anchors_dim, out_dim, 0, 0};
  if (head_args(a, w, b, dims, 1, "nphm_compress_condition: unsupported widths (<= 1536)")) return -2;
  a.x = code; a.x_stride = code_dim; a.x2 = anchors; a.x_split = code_dim;
  a.y = cond; a.y_rep = n_rows; a.y_stride = out_dim + expr_dim; a.tail = z_ex; a.tail_w = expr_dim;
  hipLaunchKernelGGL(nphm::fit::head_fwd_kernel, dim3(1), dim3(1024), 0, static_cast<hipStream_t>(stream), a);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : nphm_fail("nphm_compress_condition launch", e);
}

int nphm_compress_condition_backward(const float* weight, const float* bias, int code_dim, int anchors_dim, int out_dim, const float* g_cond,
                                     int n_rows, int expr_dim, float* g_code, float* g_anchors, void* stream) {
  nphm::fit::HeadArgs a{};
  if (!weight || !bias || !g_cond || !g_code || !g_anchors || n_rows <= 0 || expr_dim <= 0 || code_dim <= 0 || anchors_dim <= 0)
    return nphm_fail_msg("nphm_compress_condition_backward: bad arguments");
  const float* w[3] = {weight, nullptr, nullptr};
  const float* b[3] = {bias, nullptr, nullptr};
  const int dims[4] = {code_dim + anchors_dim, out_dim, 0, 0};
  if (head_args(a, w, b, dims, 1, "nphm_compress_condition_backward: unsupported widths (<= 1536)")) return -2;
  a.g_y = g_cond; a.g_rows = n_rows; a.g_stride = out_dim + expr_dim;
  a.g_x = g_code; a.x_stride = code_dim; a.g_x2 = g_anchors; a.x_split = code_dim;
  hipLaunchKernelGGL(nphm::fit::head_bwd_kernel, dim3(1), dim3(1024), 0, static_cast<hipStream_t>(stream), a);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : nphm_fail("nphm_compress_condition_backward launch", e);
}

size_t nphm_identity_latent_grad_scratch_bytes(int n_rows) {
  return n_rows > 0 ? size_t(n_rows) * nphm::N_MEMBERS * nphm::LAT_GLOB * sizeof(float) : 0;
}

int nphm_identity_latent_grad(const float* lin0_weight, const float* lin2_weight, const float* g_bias0, const float* g_bias2,
                              int n_rows, float* g_lat, void* scratch, void* stream) {
  if (!lin0_weight || !lin2_weight || !g_bias0 || !g_bias2 || !g_lat || !scratch || n_rows <= 0)
    return nphm_fail_msg("nphm_identity_latent_grad: bad arguments");
  hipStream_t st = static_cast<hipStream_t>(stream);
  static_assert(nphm::LAT_GLOB == 64 && nphm::LAT_LOC == 32 && nphm::LAT_COND == 96, "column split of latent_blocks_kernel");
  hipLaunchKernelGGL(nphm::fit::latent_blocks_kernel, dim3(nphm::N_MEMBERS, n_rows), dim3(nphm::LAT_COND * nphm::fit::LB_SLICES), 0, st,
                     lin0_weight, lin2_weight, g_bias0, g_bias2, g_lat, static_cast<float*>(scratch));
  hipLaunchKernelGGL(nphm::fit::latent_global_kernel, dim3(n_rows), dim3(nphm::LAT_GLOB), 0, st,
                     static_cast<const float*>(scratch), g_lat);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : nphm_fail("nphm_identity_latent_grad launch", e);
}

int nphm_identity_blend_members(const float* blend_weights, const float* member_values, int64_t n_points, float* sdf, void* stream) {
  if (!blend_weights || !member_values || !sdf) return nphm_fail_msg("nphm_identity_blend_members: null pointer");
  if (n_points <= 0) return n_points == 0 ? 0 : nphm_fail_msg("nphm_identity_blend_members: negative count");
  hipLaunchKernelGGL(nphm::fit::blend_members_kernel, dim3(unsigned((n_points + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), blend_weights, member_values, sdf, n_points);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : nphm_fail("nphm_identity_blend_members launch", e);
}

int nphm_gather_rows_backward(const float* g_out, const int64_t* idx, int n_draws, int n_rows, int width, float* g_table,
                              void* stream) {
  if (!g_out || !idx || !g_table) return nphm_fail_msg("nphm_gather_rows_backward: null pointer");
  if (n_draws <= 0 || n_rows <= 0 || width <= 0) return nphm_fail_msg("nphm_gather_rows_backward: bad sizes");
  const int n = n_rows * width;
  hipLaunchKernelGGL(nphm::fit::gather_rows_bwd_kernel, dim3((n + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream),
                     g_out, idx, n_draws, n_rows, width, g_table);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : nphm_fail("nphm_gather_rows_backward launch", e);
}

int nphm_fit_inputs(const int64_t* drawn, int n_rows, int n_points, const float* clouds, int n_obs, int cloud_points, int cloud_width,
                    const float* z_shape, int shape_dim, const float* z_expr_table, int expr_dim, float* obs, float* z_ex,
                    float* glob_cond, void* stream) {
  if (!drawn || !clouds || !z_shape || !z_expr_table || !obs || !z_ex || !glob_cond) return nphm_fail_msg("nphm_fit_inputs: null pointer");
  if (n_rows <= 0 || n_points <= 0 || n_obs <= 0 || cloud_points <= 0 || cloud_width <= 0 || shape_dim <= 0 || expr_dim <= 0)
    return nphm_fail_msg("nphm_fit_inputs: bad sizes");
  const int64_t total = std::max(int64_t(n_rows) * n_points * cloud_width, int64_t(n_rows) * (shape_dim + expr_dim));
  if (total > 0x7fffffffLL) return nphm_fail_msg("nphm_fit_inputs: too many elements");
  nphm::fit::InputsArgs a{drawn, clouds, z_shape, z_expr_table, n_rows, n_points, cloud_points, cloud_width, shape_dim, expr_dim, obs, z_ex, glob_cond,
                          nullptr, 0, 0, 0, nullptr, nullptr};
  hipLaunchKernelGGL(nphm::fit::fit_inputs_kernel, dim3(unsigned((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), a);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : nphm_fail("nphm_fit_inputs launch", e);
}

int nphm_fit_inputs_ring(const int64_t* host_ring, int ring_slots, int64_t slot_stride, int64_t n_total, unsigned* control, int64_t* drawn_out,
                         int n_rows, int n_points, const float* clouds, int n_obs, int cloud_points, int cloud_width,
                         const float* z_shape, int shape_dim, const float* z_expr_table, int expr_dim, float* obs, float* z_ex,
                         float* glob_cond, void* stream) {
  if (!host_ring || !control || !drawn_out || !clouds || !z_shape || !z_expr_table || !obs || !z_ex || !glob_cond)
    return nphm_fail_msg("nphm_fit_inputs_ring: null pointer");
  if (n_rows <= 0 || n_points <= 0 || n_obs <= 0 || cloud_points <= 0 || cloud_width <= 0 || shape_dim <= 0 || expr_dim <= 0)
    return nphm_fail_msg("nphm_fit_inputs_ring: bad sizes");
  if (ring_slots <= 0 || n_total < int64_t(n_rows) * (1 + n_points) || slot_stride < n_total || n_rows > nphm::fit::FIT_RING_MAX_ROWS)
    return nphm_fail_msg("nphm_fit_inputs_ring: bad ring geometry");
  const int64_t total = std::max(int64_t(n_rows) * n_points * cloud_width, int64_t(n_rows) * (shape_dim + expr_dim));
  if (total > 0x7fffffffLL) return nphm_fail_msg("nphm_fit_inputs_ring: too many elements");
  nphm::fit::InputsArgs a{nullptr, clouds, z_shape, z_expr_table, n_rows, n_points, cloud_points, cloud_width, shape_dim, expr_dim, obs, z_ex, glob_cond,
                          host_ring, ring_slots, slot_stride, n_total, drawn_out, control};
  hipLaunchKernelGGL(nphm::fit::fit_inputs_kernel, dim3(unsigned((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), a);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : nphm_fail("nphm_fit_inputs_ring launch", e);
}

int nphm_fit_inputs_backward(const float* g_z_ex, int64_t z_ex_row_stride, const float* g_glob_cond, const int64_t* obs_idx, int n_rows,
                             int n_obs, int shape_dim, int expr_dim, const float* const g_shape_uses[4], const float* g_table_use,
                             float* g_z_expr_table, float* g_z_shape, void* stream) {
  if (!obs_idx || !g_z_expr_table) return nphm_fail_msg("nphm_fit_inputs_backward: null pointer");
  if (n_rows <= 0 || n_obs <= 0 || shape_dim <= 0 || expr_dim <= 0) return nphm_fail_msg("nphm_fit_inputs_backward: bad sizes");
  const int total = std::max(n_obs * expr_dim, g_z_shape ? shape_dim : 0);
  nphm::fit::InputsBwdArgs a{g_z_ex, z_ex_row_stride, g_glob_cond, obs_idx, n_rows, n_obs, shape_dim, expr_dim, {nullptr, nullptr, nullptr, nullptr},
                             g_table_use, g_z_expr_table, g_z_shape};
  for (int q = 0; q < 4; ++q) a.shape_parts[q] = g_shape_uses ? g_shape_uses[q] : nullptr;
  hipLaunchKernelGGL(nphm::fit::fit_inputs_bwd_kernel, dim3((total + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), a);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : nphm_fail("nphm_fit_inputs_backward launch", e);
}

int nphm_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float beta1, float beta2,
                   float step_size, float bias_correction2_sqrt, float eps, void* stream) {
  if (!param || !grad || !exp_avg || !exp_avg_sq) return nphm_fail_msg("nphm_adam_step: null pointer");
  if (n <= 0) return n == 0 ? 0 : nphm_fail_msg("nphm_adam_step: negative count");
  hipLaunchKernelGGL(nphm::fit::adam_kernel, dim3(unsigned((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     param, grad, exp_avg, exp_avg_sq, n, 1.f - beta1, beta2, 1.f - beta2, step_size, bias_correction2_sqrt, eps);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : nphm_fail("nphm_adam_step launch", e);
}

int nphm_adam_step_pair(float* const param[2], const float* const grad[2], float* const exp_avg[2], float* const exp_avg_sq[2],
                        const int64_t n[2], const float* scalars, void* stream) {
  if (!param || !grad || !exp_avg || !exp_avg_sq || !n || !scalars) return nphm_fail_msg("nphm_adam_step_pair: null pointer");
  nphm::fit::AdamPairArgs a{};
  for (int q = 0; q < 2; ++q) {
    if (n[q] < 0 || (n[q] > 0 && (!param[q] || !grad[q] || !exp_avg[q] || !exp_avg_sq[q]))) return nphm_fail_msg("nphm_adam_step_pair: bad tensor");
    a.p[q] = param[q]; a.g[q] = grad[q]; a.m[q] = exp_avg[q]; a.v[q] = exp_avg_sq[q]; a.n[q] = n[q];
  }
  a.scalars = scalars;
  const int64_t total = n[0] + n[1];
  if (total == 0) return 0;
  hipLaunchKernelGGL(nphm::fit::adam_pair_kernel, dim3(unsigned((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), a);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : nphm_fail("nphm_adam_step_pair launch", e);
}

int nphm_inverse3x3_strided(const float* matrices, int64_t matrix_stride, int64_t row_stride, int64_t col_stride,
                            float* inverses, int64_t n, void* stream) {
  if (!matrices || !inverses) return nphm_fail_msg("nphm_inverse3x3_strided: null pointer");
  if (n <= 0) return n == 0 ? 0 : nphm_fail_msg("nphm_inverse3x3_strided: negative count");
  hipLaunchKernelGGL(nphm::fit::inverse3x3_strided_kernel, dim3(unsigned((n + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), matrices, matrix_stride, row_stride, col_stride, inverses, n);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : nphm_fail("nphm_inverse3x3_strided launch", e);
}

int nphm_mlp_cond_grad(const void* bias_partials, int64_t n_points, int n_rows, int hidden_dim,
                       const float* lin0_weight, int ld0, int off0, const float* skip_weight, int ld_skip, int off_skip,
                       int lat_dim, float* grad_cond, void* stream) {
  if (!bias_partials || !lin0_weight || !skip_weight || !grad_cond)
    return nphm_fail_msg("nphm_mlp_cond_grad: null pointer");
  if (n_rows <= 0 || n_points <= 0 || hidden_dim <= 0 || hidden_dim > nphm::fit::CG_HMAX || lat_dim <= 0 || off0 < 0 || off_skip < 0 ||
      off0 + lat_dim > ld0 || off_skip + lat_dim > ld_skip)
    return nphm_fail_msg("nphm_mlp_cond_grad: bad sizes");
  hipLaunchKernelGGL(nphm::fit::cond_grad_kernel, dim3((lat_dim + 63) / 64, n_rows), dim3(64 * nphm::fit::CG_PARTS), 0,
                     static_cast<hipStream_t>(stream), static_cast<const float*>(bias_partials), int((n_points + 31) / 32), hidden_dim,
                     lin0_weight, ld0, off0,
                     skip_weight, ld_skip, off_skip, lat_dim, grad_cond);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : nphm_fail("nphm_mlp_cond_grad launch", e);
}

}  // extern "C"

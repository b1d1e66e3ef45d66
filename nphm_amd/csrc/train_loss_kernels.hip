// train_loss_kernels.hip - the loss terms of the identity decoder's training step (src/NPHM/models/loss_functions.py:51-110
// after the four decoder evaluations: surf_sdf, normals, space_sdf, grad, lat_reg, anchors, symm_dist, middle_dist) in ONE
// launch, their gradients w.r.t. the SDF values, the spatial gradients, the latent codes and the predicted anchors in a
// second one.  The PyTorch formulation of the same terms is ~40 elementwise / reduction launches forward and ~50 in the
// backward pass - on a step of 10 ms that is bound by three large kernels they are 4 % of the time.
//
// Point layout: [B][N] with N = n_face + n_non + n_near + n_far consecutive slices (face | non-face | near | far), the
// layout of the mirrored compute_loss (nphm_amd/loss_functions.py).  Deterministic: per-block partial sums, combined in block
// order by the last block to finish.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "capi_common.h"

namespace nphm {
namespace tloss {

constexpr int N_TERMS = 8;     // surf_sdf, normals, space_sdf, grad, lat_reg, anchors, symm_dist, middle_dist
constexpr int BLOCK = 256;

struct Args {
  const float* sdf;          // [B, N]
  const float* grad;         // [B, N, 3]
  const float* normals;      // [B, n_face + n_non, 3]
  const float* z;            // [B, L] latent codes
  const float* anchors;      // [B, K, 3] or null
  const float* anchors_gt;   // [B, K, 3] or null
  int B, N, n_face, n_non, n_near, n_far, L, K;
  int g, loc, n_symm, n_mid_pairs;     // latent layout [glob g | 2 n_symm local codes | middle codes | background]; 0: no local codes
  float* partial;            // [gridDim.x][8]
  unsigned* counter;
  float* row;                // [8]
  const float* c;            // backward: [8] d L / d term
  float* g_sdf;              // [B, N]
  float* g_grad;             // [B, N, 3]
  float* g_z;                // [B, L]
  float* g_anchors;          // [B, K, 3] or null
};

__device__ inline float count_of(const Args& a, int term) {
  const float B = float(a.B);
  switch (term) {
    case 0: case 1: return B * float(a.n_face + a.n_non);
    case 2: return B * float(a.n_far);
    case 3: return B * float(a.N);
    case 4: return B;
    case 5: return B * float(a.K) * 3.f;
    case 6: return B * float(a.n_symm);
    default: return B * float(a.n_mid_pairs);
  }
}

// first latent column of pair `q` of the symmetric (mid = false) or middle (mid = true) pairs
__device__ inline int pair_base(const Args& a, bool mid, int q) {
  return a.g + (mid ? 2 * a.n_symm * a.loc : 0) + 2 * q * a.loc;
}

__device__ inline float pair_norm(const Args& a, int b, bool mid, int q) {
  const float* z = a.z + size_t(b) * a.L + pair_base(a, mid, q);
  float s = 0.f;
  for (int j = 0; j < a.loc; ++j) { const float d = z[j] - z[a.loc + j]; s = fmaf(d, d, s); }
  return sqrtf(s);
}

__global__ __launch_bounds__(BLOCK) void loss_fwd_kernel(Args a) {
  __shared__ float red[BLOCK / 64][N_TERMS];
  __shared__ bool last;
  float acc[N_TERMS];
#pragma unroll
  for (int k = 0; k < N_TERMS; ++k) acc[k] = 0.f;
  const int64_t stride = int64_t(gridDim.x) * BLOCK, t0 = int64_t(blockIdx.x) * BLOCK + threadIdx.x;
  const int n_surf = a.n_face + a.n_non;
  for (int64_t e = t0; e < int64_t(a.B) * a.N; e += stride) {
    const int b = int(e / a.N), p = int(e % a.N);
    const float s = a.sdf[e];
    const float gx = a.grad[e * 3], gy = a.grad[e * 3 + 1], gz = a.grad[e * 3 + 2];
    acc[3] += fabsf(sqrtf(gx * gx + gy * gy + gz * gz) - 1.f);
    if (p < n_surf) {
      acc[0] += fabsf(s);
      const float* n = a.normals + (size_t(b) * n_surf + p) * 3;
      const float dx = gx - n[0], dy = gy - n[1], dz = gz - n[2];
      const float err = sqrtf(dx * dx + dy * dy + dz * dz);
      acc[1] += p < a.n_face ? err : 0.5f * fminf(err, 0.75f);
    } else if (p >= n_surf + a.n_near) {
      acc[2] += expf(-10.f * fabsf(s));
    }
  }
  for (int64_t e = t0; e < int64_t(a.B) * a.L; e += stride) { const float v = a.z[e]; acc[4] = fmaf(v, v, acc[4]); }
  if (a.anchors)
    for (int64_t e = t0; e < int64_t(a.B) * a.K * 3; e += stride) { const float d = a.anchors[e] - a.anchors_gt[e]; acc[5] = fmaf(d, d, acc[5]); }
  for (int64_t e = t0; e < int64_t(a.B) * a.n_symm; e += stride) acc[6] += pair_norm(a, int(e / a.n_symm), false, int(e % a.n_symm));
  for (int64_t e = t0; e < int64_t(a.B) * a.n_mid_pairs; e += stride) acc[7] += pair_norm(a, int(e / a.n_mid_pairs), true, int(e % a.n_mid_pairs));
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < N_TERMS; ++k) {
    float v = acc[k];
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) v += __shfl_xor(v, m);
    if (lane == 0) red[wave][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < N_TERMS) {
    float v = 0.f;
    for (int w = 0; w < BLOCK / 64; ++w) v += red[w][threadIdx.x];
    a.partial[size_t(blockIdx.x) * N_TERMS + threadIdx.x] = v;
  }
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) last = atomicAdd(a.counter, 1u) == gridDim.x - 1;
  __syncthreads();
  if (!last) return;
  __threadfence();
  if (threadIdx.x < N_TERMS) {
    float v = 0.f;
    for (unsigned blk = 0; blk < gridDim.x; ++blk) v += a.partial[size_t(blk) * N_TERMS + threadIdx.x];
    const float n = count_of(a, threadIdx.x);
    a.row[threadIdx.x] = n > 0.f ? v / n : 0.f;
  }
  if (threadIdx.x == 0) *a.counter = 0u;
}

__global__ __launch_bounds__(BLOCK) void loss_bwd_kernel(Args a) {
  const int64_t stride = int64_t(gridDim.x) * BLOCK, t0 = int64_t(blockIdx.x) * BLOCK + threadIdx.x;
  const int n_surf = a.n_face + a.n_non;
  float w[N_TERMS];
#pragma unroll
  for (int k = 0; k < N_TERMS; ++k) { const float n = count_of(a, k); w[k] = n > 0.f ? a.c[k] / n : 0.f; }
  for (int64_t e = t0; e < int64_t(a.B) * a.N; e += stride) {
    const int b = int(e / a.N), p = int(e % a.N);
    const float s = a.sdf[e];
    const float sg = s > 0.f ? 1.f : (s < 0.f ? -1.f : 0.f);
    const float gx = a.grad[e * 3], gy = a.grad[e * 3 + 1], gz = a.grad[e * 3 + 2];
    const float nrm = sqrtf(gx * gx + gy * gy + gz * gz);
    const float dev = nrm - 1.f;
    // |  |g| - 1 |: sign(|g| - 1) g / |g| (0 at g = 0, like the norm's backward)
    float k3 = nrm > 0.f ? w[3] * (dev > 0.f ? 1.f : (dev < 0.f ? -1.f : 0.f)) / nrm : 0.f;
    float ox = k3 * gx, oy = k3 * gy, oz = k3 * gz, os = 0.f;
    if (p < n_surf) {
      os = w[0] * sg;
      const float* n = a.normals + (size_t(b) * n_surf + p) * 3;
      const float dx = gx - n[0], dy = gy - n[1], dz = gz - n[2];
      const float err = sqrtf(dx * dx + dy * dy + dz * dz);
      float k1 = 0.f;
      if (err > 0.f) k1 = p < a.n_face ? w[1] / err : (err <= 0.75f ? 0.5f * w[1] / err : 0.f);
      ox = fmaf(k1, dx, ox); oy = fmaf(k1, dy, oy); oz = fmaf(k1, dz, oz);
    } else if (p >= n_surf + a.n_near) {
      os = w[2] * -10.f * sg * expf(-10.f * fabsf(s));
    }
    a.g_sdf[e] = os;
    a.g_grad[e * 3] = ox; a.g_grad[e * 3 + 1] = oy; a.g_grad[e * 3 + 2] = oz;
  }
  const int sym0 = a.g, mid0 = a.g + 2 * a.n_symm * a.loc, mid1 = mid0 + 2 * a.n_mid_pairs * a.loc;
  for (int64_t e = t0; e < int64_t(a.B) * a.L; e += stride) {
    const int b = int(e / a.L), j = int(e % a.L);
    float v = 2.f * w[4] * a.z[e];
    const bool in_sym = a.n_symm > 0 && j >= sym0 && j < mid0, in_mid = a.n_mid_pairs > 0 && j >= mid0 && j < mid1;
    if (in_sym || in_mid) {
      const int r = j - (in_sym ? sym0 : mid0);
      const int q = r / (2 * a.loc), within = r % (2 * a.loc);
      const float nrm = pair_norm(a, b, in_mid, q);
      if (nrm > 0.f) {
        const float* z = a.z + size_t(b) * a.L + pair_base(a, in_mid, q);
        const int jj = within % a.loc;
        const float d = z[jj] - z[a.loc + jj];
        v += (within < a.loc ? 1.f : -1.f) * w[in_sym ? 6 : 7] * d / nrm;
      }
    }
    a.g_z[e] = v;
  }
  if (a.g_anchors)
    for (int64_t e = t0; e < int64_t(a.B) * a.K * 3; e += stride) a.g_anchors[e] = 2.f * w[5] * (a.anchors[e] - a.anchors_gt[e]);
}

}  // namespace tloss
}  // namespace nphm

extern "C" {

static int fill_args(nphm::tloss::Args& a, const float* sdf, const float* grad, const float* normals, const float* z, const float* anchors,
                     const float* anchors_gt, int n_rows, const int sizes[4], int lat_dim, int n_anchors, const int layout[4],
                     const char* who) {
  if (!sdf || !grad || !normals || !z || n_rows <= 0 || lat_dim <= 0) return nphm_fail_msg(who);
  for (int i = 0; i < 4; ++i)
    if (sizes[i] < 0) return nphm_fail_msg(who);
  if ((anchors == nullptr) != (anchors_gt == nullptr) || (anchors && n_anchors <= 0)) return nphm_fail_msg(who);
  a.sdf = sdf; a.grad = grad; a.normals = normals; a.z = z; a.anchors = anchors; a.anchors_gt = anchors_gt;
  a.B = n_rows; a.n_face = sizes[0]; a.n_non = sizes[1]; a.n_near = sizes[2]; a.n_far = sizes[3];
  a.N = sizes[0] + sizes[1] + sizes[2] + sizes[3];
  a.L = lat_dim; a.K = anchors ? n_anchors : 0;
  a.g = layout[0]; a.loc = layout[1]; a.n_symm = layout[2]; a.n_mid_pairs = layout[3];
  if (a.g < 0 || a.loc < 0 || a.n_symm < 0 || a.n_mid_pairs < 0 || a.g + 2 * (a.n_symm + a.n_mid_pairs) * a.loc > a.L)
    return nphm_fail_msg(who);
  return 0;
}

int nphm_train_loss_blocks(void) { return 256; }

int nphm_train_loss(const float* sdf, const float* grad, const float* normals, const float* z, const float* anchors,
                    const float* anchors_gt, int n_rows, const int sizes[4], int lat_dim, int n_anchors, const int layout[4],
                    float* partial, unsigned* counter, float* row, void* stream) {
  nphm::tloss::Args a{};
  if (fill_args(a, sdf, grad, normals, z, anchors, anchors_gt, n_rows, sizes, lat_dim, n_anchors, layout, "nphm_train_loss: bad arguments"))
    return -2;
  if (!partial || !counter || !row) return nphm_fail_msg("nphm_train_loss: null pointer");
  a.partial = partial; a.counter = counter; a.row = row;
  hipLaunchKernelGGL(nphm::tloss::loss_fwd_kernel, dim3(nphm_train_loss_blocks()), dim3(nphm::tloss::BLOCK), 0,
                     static_cast<hipStream_t>(stream), a);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : nphm_fail("nphm_train_loss launch", e);
}

int nphm_train_loss_backward(const float* sdf, const float* grad, const float* normals, const float* z, const float* anchors,
                             const float* anchors_gt, int n_rows, const int sizes[4], int lat_dim, int n_anchors,
                             const int layout[4], const float* g_terms, float* g_sdf, float* g_grad, float* g_z,
                             float* g_anchors, void* stream) {
  nphm::tloss::Args a{};
  if (fill_args(a, sdf, grad, normals, z, anchors, anchors_gt, n_rows, sizes, lat_dim, n_anchors, layout,
                "nphm_train_loss_backward: bad arguments"))
    return -2;
  if (!g_terms || !g_sdf || !g_grad || !g_z || (anchors && !g_anchors)) return nphm_fail_msg("nphm_train_loss_backward: null pointer");
  a.c = g_terms; a.g_sdf = g_sdf; a.g_grad = g_grad; a.g_z = g_z; a.g_anchors = anchors ? g_anchors : nullptr;
  const int64_t n = int64_t(a.B) * (a.N > a.L ? a.N : a.L);
  const int blocks = int((n + nphm::tloss::BLOCK - 1) / nphm::tloss::BLOCK);
  hipLaunchKernelGGL(nphm::tloss::loss_bwd_kernel, dim3(blocks < 1 ? 1 : (blocks > 4096 ? 4096 : blocks)), dim3(nphm::tloss::BLOCK), 0,
                     static_cast<hipStream_t>(stream), a);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : nphm_fail("nphm_train_loss_backward launch", e);
}

}  // extern "C"

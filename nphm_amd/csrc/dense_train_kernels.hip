// dense_train_kernels.hip — the dense skip-MLP (DeepSDF backbone of the forward-deformation network,
// src/NPHM/models/deepSDF.py:64-89) with TRAINABLE parameters: forward, data gradient and weight gradient of one
// `Softplus(x W^T / s + e)` layer on the matrix pipe (widened in round 6: compute_loss_corresp_forward,
// src/NPHM/models/loss_functions.py:282-326, driven by scripts/training/train_corresp.py - the second training stage).
//
// The fitting / lattice kernels of mlp_kernel.hip keep a point's activations in LDS for the whole network; training needs
// every layer's activations again for the weight gradients, so here a layer is ONE launch that reads and writes HBM:
//   forward   y  = act(alpha x W^T + e)          gemm_nt_kernel, epilogue 1 (e: bias + latent term, per batch row or per point)
//   backward  gp = g * act'(y)                    gpre_kernel (act' from y alone: 1 - exp(-beta y)), also writes gp^T
//             dx = alpha gp W                     gemm_nt_kernel on (gp, W^T)
//             dW = alpha gp^T x                   gemm_nt_kernel on (gp^T, x^T) with the point axis as K, cut into splits that
//                                                 reduce_splits_kernel adds in split order (no atomics: bitwise reproducible)
//             db = column sums of gp               gpre_kernel's per-tile sums + column_sums_kernel (fixed order)
// One kernel computes all three products: C[M,N] = alpha A[M,K] B[N,K]^T, both operands K-contiguous fp32 in HBM, split into
// bf16 hi / lo on their way into LDS, three MFMA passes per product (fp32-equivalent: hi hi + hi lo + lo hi), fp32 accumulate.
// Workgroup = 8 wavefronts = a 128 x 256 tile of C (wavefront: 64 x 64 = four 32x32 accumulators), K in chunks of 32
// through two LDS buffers of MFMA fragments ([k-step][k-group][row][8]: one conflict-free ds_read_b128 per fragment): a chunk's
// 24 MFMAs per wavefront carry the split + LDS writes of the NEXT chunk between them, whose global loads were issued two
// chunks ahead; one barrier per chunk.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "capi_common.h"

namespace nphm {
namespace dense {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int BM = 128, BN = 256, KC = 32, THREADS = 512;
constexpr int KS = KC / 16;            // MFMA K-steps per chunk
// rows of one (k-step, k-group) block of fragments in LDS, padded so that consecutive blocks sit 64 bytes apart modulo the 128
// bytes of a bank row: the loader's 8-byte writes of one wavefront (4 rows x 8 blocks x 2 halves) then cover every bank equally
// (unpadded: the blocks of a row hit ONE bank group - a conflict on every ds_write_b64)
constexpr int BMP = BM + 4, BNP = BN + 4;

struct GemmArgs {
  const float* A;   // [M, K], row stride lda
  const float* B;   // [N, K], row stride ldb
  float* C;         // [M, N], row stride ldc (k_splits > 1: [k_splits][M][ldc])
  const float* E;   // epilogue 1 / 2: [ceil(M / e_rows), N] contiguous - row m of C takes row m / e_rows of E
  int M, N, K, lda, ldb, ldc, e_rows;
  int k_per_split;  // K elements per blockIdx.z (a multiple of KC)
  float alpha, beta;
  int epilogue;     // 0: C = alpha acc   1: C = act(alpha acc + E), act = Softplus(beta) (beta <= 0: ReLU)   2: C = alpha acc + E
};

__device__ __forceinline__ unsigned pk_bf16(float a, float b) {
  const f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
// four floats -> (hi, lo) as two dwords each
__device__ __forceinline__ void split4(const f32x4& x, u32x2& hi, u32x2& lo) {
  hi[0] = pk_bf16(x[0], x[1]);
  hi[1] = pk_bf16(x[2], x[3]);
  const float h0 = __builtin_bit_cast(float, hi[0] << 16), h1 = __builtin_bit_cast(float, hi[0] & 0xffff0000u);
  const float h2 = __builtin_bit_cast(float, hi[1] << 16), h3 = __builtin_bit_cast(float, hi[1] & 0xffff0000u);
  lo[0] = pk_bf16(x[0] - h0, x[1] - h1);
  lo[1] = pk_bf16(x[2] - h2, x[3] - h3);
}

// four consecutive k of row `r` (zero beyond the operand).  Every load is UNCONDITIONAL (addresses clamped into the operand, the
// value masked afterwards): a branch around a load makes hipcc's s_waitcnt insertion wait for each load before it issues the
// next (the first version of this kernel: 12 serialized L2 round trips per chunk, 190 us for a product the matrix pipe needs
// 23 us for).  VEC: one 16-byte load - rows 16-byte aligned and K a multiple of 4, so a group of four is all in or all out.
template <bool VEC>
__device__ __forceinline__ f32x4 load4(const float* P, int ld, int rows, int K, int r, int k) {
  const float* q = P + size_t(r < rows ? r : rows - 1) * ld;
  f32x4 v;
  if constexpr (VEC) {
    v = *reinterpret_cast<const f32x4*>(q + (k < K ? k : 0));
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = q[k + i < K ? k + i : 0];
  }
  return v;
}
// ... and the mask, applied when the values are USED (stage): the loads stay in flight across the MFMAs of the chunk before
__device__ __forceinline__ f32x4 mask4(f32x4 v, int rows, int K, int r, int k) {
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = (r < rows && k + i < K) ? v[i] : 0.f;
  return v;
}

__device__ __forceinline__ float activation(float v, float beta) {
  if (beta <= 0.f) return fmaxf(v, 0.f);
  // Softplus(beta) = max(v, 0) + log(1 + exp(-beta |v|)) / beta  (nn.Softplus's threshold 20 changes nothing in fp32), base 2 on
  // the raw v_exp_f32 / v_log_f32 (1 ulp; the argument of the log lies in [1, 2]): 7 instructions where expf + log1pf + a
  // division are ~120 - 64 values per lane made the libm epilogue longer than the K loop of a 512-wide layer
  const float t = __builtin_amdgcn_exp2f(-1.44269504f * beta * fabsf(v));
  return fmaxf(v, 0.f) + __builtin_amdgcn_logf(1.f + t) * (0.693147181f / beta);
}

template <bool VA, bool VB>
__global__ __launch_bounds__(THREADS) void gemm_nt_kernel(GemmArgs p) {
  // two LDS buffers of fragments: chunk c is multiplied out of buffer c & 1 while chunk c + 1 is split and written into the other
  constexpr int A_PLANE = KS * 2 * BMP * 16, B_PLANE = KS * 2 * BNP * 16;
  __shared__ __attribute__((aligned(16))) char a_hi[2][A_PLANE], a_lo[2][A_PLANE];     // 100 KB in all: one workgroup per CU
  __shared__ __attribute__((aligned(16))) char b_hi[2][B_PLANE], b_lo[2][B_PLANE];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int k_begin = blockIdx.z * p.k_per_split;
  const int k_end = min(p.K, k_begin + p.k_per_split);
  float* C = p.C + size_t(blockIdx.z) * size_t(p.M) * p.ldc;
  // loader: thread -> (row t / 8 + 64 i, four k at 4 (t % 8))
  constexpr int LROWS = THREADS / (KC / 4), NA = BM / LROWS, NB = BN / LROWS;
  static_assert(NA + NB == 6, "six 16-byte loads per thread and chunk: one behind each group of four MFMAs");
  const int lr = t / (KC / 4), lk = (t % (KC / 4)) * 4;
  const int frag_off = ((lk >> 4) * 2 + ((lk >> 3) & 1));        // (k-step, k-group) of this thread's four k
  // two register sets: chunk c + 2 is requested into set c & 1 at the top of iteration c (two iterations of MFMAs ahead of its use)
  f32x4 regs[2][NA + NB];
  auto fetch = [&](auto set, int k0) __attribute__((always_inline)) {
    constexpr int S = decltype(set)::value;
#pragma unroll
    for (int i = 0; i < NA; ++i) regs[S][i] = load4<VA>(p.A, p.lda, p.M, k_end, m0 + lr + LROWS * i, k0 + lk);
#pragma unroll
    for (int i = 0; i < NB; ++i) regs[S][NA + i] = load4<VB>(p.B, p.ldb, p.N, k_end, n0 + lr + LROWS * i, k0 + lk);
  };
  // unit u of the staging of the chunk at k0 held in register set S -> LDS buffer `buf`
  auto stage_unit = [&](auto set, auto unit, int buf, int k0) __attribute__((always_inline)) {
    constexpr int S = decltype(set)::value, u = decltype(unit)::value;
    u32x2 hi, lo;
    if constexpr (u < NA) {
      split4(mask4(regs[S][u], p.M, k_end, m0 + lr + LROWS * u, k0 + lk), hi, lo);
      const int o = ((frag_off * BMP + lr + LROWS * u) * 8 + (lk & 7)) * 2;
      *reinterpret_cast<u32x2*>(a_hi[buf] + o) = hi;
      *reinterpret_cast<u32x2*>(a_lo[buf] + o) = lo;
    } else {
      constexpr int i = u - NA;
      split4(mask4(regs[S][u], p.N, k_end, n0 + lr + LROWS * i, k0 + lk), hi, lo);
      const int o = ((frag_off * BNP + lr + LROWS * i) * 8 + (lk & 7)) * 2;
      *reinterpret_cast<u32x2*>(b_hi[buf] + o) = hi;
      *reinterpret_cast<u32x2*>(b_lo[buf] + o) = lo;
    }
  };
  const int wm = wave >> 2, wn = wave & 3, j = lane & 31, kg = lane >> 5;
  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = f32x16{};

  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  const int n_chunks = (k_end - k_begin + KC - 1) / KC;
  fetch(I0{}, k_begin);
  fetch(I1{}, k_begin + KC);                                       // (beyond k_end: clamped addresses, masked to zero when staged)
  stage_unit(I0{}, std::integral_constant<int, 0>{}, 0, k_begin); stage_unit(I0{}, std::integral_constant<int, 1>{}, 0, k_begin);
  stage_unit(I0{}, std::integral_constant<int, 2>{}, 0, k_begin); stage_unit(I0{}, std::integral_constant<int, 3>{}, 0, k_begin);
  stage_unit(I0{}, std::integral_constant<int, 4>{}, 0, k_begin); stage_unit(I0{}, std::integral_constant<int, 5>{}, 0, k_begin);
  __syncthreads();

  // one chunk: 24 MFMAs per wavefront (2 K-steps x 4 accumulators x 3 products, the four accumulators interleaved so that no
  // MFMA waits for the one before it) with one sixth of the next chunk's staging behind every group of four
  auto iteration = [&](auto parity, int c) __attribute__((always_inline)) {
    constexpr int P = decltype(parity)::value;
    const int k0 = k_begin + c * KC;
    fetch(parity, k0 + 2 * KC);                                    // set P held chunk c: staged during the previous iteration
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int oa = (((ks * 2 + kg) * BMP) + wm * 64 + i * 32 + j) * 16;
        ah[i] = *reinterpret_cast<const bf16x8*>(a_hi[P] + oa);
        al[i] = *reinterpret_cast<const bf16x8*>(a_lo[P] + oa);
        const int ob = (((ks * 2 + kg) * BNP) + wn * 64 + i * 32 + j) * 16;
        bh[i] = *reinterpret_cast<const bf16x8*>(b_hi[P] + ob);
        bl[i] = *reinterpret_cast<const bf16x8*>(b_lo[P] + ob);
      }
#pragma unroll
      for (int term = 0; term < 3; ++term) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(term == 2 ? al[a] : ah[a], term == 1 ? bl[b] : bh[b], acc[a][b], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (ks == 0) {
          if (term == 0) stage_unit(std::integral_constant<int, 1 - P>{}, std::integral_constant<int, 0>{}, 1 - P, k0 + KC);
          if (term == 1) stage_unit(std::integral_constant<int, 1 - P>{}, std::integral_constant<int, 1>{}, 1 - P, k0 + KC);
          if (term == 2) stage_unit(std::integral_constant<int, 1 - P>{}, std::integral_constant<int, 2>{}, 1 - P, k0 + KC);
        } else {
          if (term == 0) stage_unit(std::integral_constant<int, 1 - P>{}, std::integral_constant<int, 3>{}, 1 - P, k0 + KC);
          if (term == 1) stage_unit(std::integral_constant<int, 1 - P>{}, std::integral_constant<int, 4>{}, 1 - P, k0 + KC);
          if (term == 2) stage_unit(std::integral_constant<int, 1 - P>{}, std::integral_constant<int, 5>{}, 1 - P, k0 + KC);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __syncthreads();        // buffer 1 - P is complete, everybody has left buffer P
  };
  for (int c = 0; c < n_chunks; c += 2) {
    iteration(I0{}, c);
    iteration(I1{}, c + 1);                                        // (an odd count: one chunk of zeros)
  }

  // ---- epilogue: register q of lane (j, kg) = C[row 8 (q / 4) + 4 kg + q % 4][column j] of its 32 x 32 tile -------------
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    const int mt = m0 + wm * 64 + a * 32;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int n = n0 + wn * 64 + b * 32 + j;
      if (n >= p.N) continue;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int m = mt + 8 * (q >> 2) + 4 * kg + (q & 3);
        if (m >= p.M) continue;
        float v = p.alpha * acc[a][b][q];
        if (p.epilogue) {
          const int er = p.e_rows >= p.M ? 0 : (p.e_rows == 1 ? m : m / p.e_rows);
          v += p.E[size_t(er) * p.N + n];
          if (p.epilogue == 1) v = activation(v, p.beta);
        }
        C[size_t(m) * p.ldc + n] = v;
      }
    }
  }
}

// out[i] = scale * (parts[0][i] + parts[1][i] + ...) in split order
__global__ __launch_bounds__(256) void reduce_splits_kernel(const float* parts, int k_splits, int64_t count, float scale, float* out) {
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= count) return;
  float s = 0.f;
  for (int z = 0; z < k_splits; ++z) s += parts[size_t(z) * count + i];
  out[i] = scale * s;
}

// gp[n][c] = g[n][c] * act'(y[n][c]) (y NULL: g itself) and its transpose gp_t[c][n] (row stride ld_t), 32 x 32 tiles through LDS.
// act' from the activation's OUTPUT: Softplus(beta): sigma(beta d) = 1 - exp(-beta y); ReLU: y > 0.
// csum (NULL: not): csum[blockIdx.y][col] = the tile's column sums in row order - the first level of the bias gradient.
__global__ __launch_bounds__(256) void gpre_kernel(const float* g, const float* y, int n, int c, float beta, float* gp, float* gp_t, int ld_t,
                                                   float* csum) {
  __shared__ float tile[32][33];
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = r0 + ty + 8 * i, col = c0 + tx;
    float v = 0.f;
    if (r < n && col < c) {
      v = g[size_t(r) * c + col];
      if (y) {
        const float yy = y[size_t(r) * c + col];
        v *= beta > 0.f ? -expm1f(-beta * yy) : (yy > 0.f ? 1.f : 0.f);
      }
      if (gp) gp[size_t(r) * c + col] = v;
    }
    tile[ty + 8 * i][tx] = v;
  }
  __syncthreads();
  if (csum && ty == 0 && c0 + tx < c) {
    float sum = 0.f;
#pragma unroll
    for (int r = 0; r < 32; ++r) sum += tile[r][tx];              // (rows beyond n hold zeros)
    csum[size_t(blockIdx.y) * c + c0 + tx] = sum;
  }
  if (!gp_t) return;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int col = c0 + ty + 8 * i, r = r0 + tx;
    if (col < c && r < n) gp_t[size_t(col) * ld_t + r] = tile[tx][ty + 8 * i];
  }
}

// out[col] = sum over the rows of x [rows, c]: thread (tx, ty) adds the rows ty, ty + 32, ... of column tx of its 32-column
// block, the 32 partial sums meet in LDS and are added in order
__global__ __launch_bounds__(1024) void column_sums_kernel(const float* x, int rows, int c, float* out) {
  __shared__ float part[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5, col = blockIdx.x * 32 + tx;
  float s = 0.f;
  if (col < c) for (int r = ty; r < rows; r += 32) s += x[size_t(r) * c + col];
  part[ty][tx] = s;
  __syncthreads();
  if (ty == 0 && col < c) {
    float o = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) o += part[i][tx];
    out[col] = o;
  }
}

}  // namespace dense
}  // namespace nphm

extern "C" {

int nphm_dense_gemm_nt(const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N, int K,
                       const float* E, int e_rows, float alpha, float beta, int epilogue, int k_splits, void* stream) {
  using namespace nphm::dense;
  if (!A || !B || !C) return nphm_fail_msg("nphm_dense_gemm_nt: null pointer");
  if (M <= 0 || N <= 0 || K <= 0 || lda < K || ldb < K || ldc < N) return nphm_fail_msg("nphm_dense_gemm_nt: bad shape / leading dimension");
  if (epilogue < 0 || epilogue > 2 || (epilogue && (!E || e_rows < 1))) return nphm_fail_msg("nphm_dense_gemm_nt: epilogue 1 / 2 need E and e_rows >= 1");
  if (k_splits < 1 || (k_splits > 1 && epilogue)) return nphm_fail_msg("nphm_dense_gemm_nt: a split product takes no epilogue");
  GemmArgs a;
  a.A = A; a.B = B; a.C = C; a.E = E;
  a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb; a.ldc = ldc; a.e_rows = e_rows;
  a.k_per_split = ((K + k_splits - 1) / k_splits + KC - 1) / KC * KC;
  a.alpha = alpha; a.beta = beta; a.epilogue = epilogue;
  const dim3 grid((N + BN - 1) / BN, (M + BM - 1) / BM, k_splits);
  // 16-byte loads where rows are 16-byte aligned and every K range is a multiple of 4 (k_per_split is a multiple of 64)
  const bool va = (lda & 3) == 0 && (reinterpret_cast<uintptr_t>(A) & 15) == 0 && (K & 3) == 0;
  const bool vb = (ldb & 3) == 0 && (reinterpret_cast<uintptr_t>(B) & 15) == 0 && (K & 3) == 0;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (va && vb) hipLaunchKernelGGL((gemm_nt_kernel<true, true>), grid, dim3(THREADS), 0, st, a);
  else if (va) hipLaunchKernelGGL((gemm_nt_kernel<true, false>), grid, dim3(THREADS), 0, st, a);
  else if (vb) hipLaunchKernelGGL((gemm_nt_kernel<false, true>), grid, dim3(THREADS), 0, st, a);
  else hipLaunchKernelGGL((gemm_nt_kernel<false, false>), grid, dim3(THREADS), 0, st, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return nphm_fail("nphm_dense_gemm_nt launch", e);
  return 0;
}

int nphm_dense_reduce_splits(const float* parts, int k_splits, int64_t count, float scale, float* out, void* stream) {
  if (!parts || !out || k_splits < 1 || count <= 0) return nphm_fail_msg("nphm_dense_reduce_splits: bad arguments");
  hipLaunchKernelGGL(nphm::dense::reduce_splits_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), parts, k_splits, count, scale, out);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return nphm_fail("nphm_dense_reduce_splits launch", e);
  return 0;
}

int nphm_dense_gpre(const float* g, const float* y, int n, int c, float beta, float* gp, float* gp_t, int ld_t, float* column_sums,
                    void* stream) {
  if (!g || n <= 0 || c <= 0 || (!gp && !gp_t) || (gp_t && ld_t < n)) return nphm_fail_msg("nphm_dense_gpre: bad arguments");
  hipLaunchKernelGGL(nphm::dense::gpre_kernel, dim3((c + 31) / 32, (n + 31) / 32), dim3(256), 0, static_cast<hipStream_t>(stream),
                     g, y, n, c, beta, gp, gp_t, ld_t, column_sums);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return nphm_fail("nphm_dense_gpre launch", e);
  return 0;
}

int nphm_dense_column_sums(const float* x, int rows, int c, float* out, void* stream) {
  if (!x || !out || rows <= 0 || c <= 0) return nphm_fail_msg("nphm_dense_column_sums: bad arguments");
  hipLaunchKernelGGL(nphm::dense::column_sums_kernel, dim3((c + 31) / 32), dim3(1024), 0, static_cast<hipStream_t>(stream), x, rows, c, out);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return nphm_fail("nphm_dense_column_sums launch", e);
  return 0;
}

}  // extern "C"

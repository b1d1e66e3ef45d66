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
// Workgroup = 4 wavefronts = a 128 x 128 tile of C (wavefront: 64 x 64 = four 32x32 accumulators; two workgroups per CU), K in chunks of 32
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

#ifndef NPHM_DENSE_SMALL
#define NPHM_DENSE_SMALL 1     // 1: 4 wavefronts, 128 x 128 tile, two workgroups per CU (their barriers fall at different times); 0: 8 wavefronts, 128 x 256
#endif
constexpr int BM = 128, BN = NPHM_DENSE_SMALL ? 128 : 256, KC = 32, THREADS = NPHM_DENSE_SMALL ? 256 : 512;
constexpr int WN = BN / 64;            // wavefronts along N (each 64 x 64)
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
  int k_per_split;  // K elements per split (a multiple of KC)
  int k_splits;
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

template <int B, int E, class F>
__device__ __forceinline__ void static_for_units(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    static_for_units<B + 1, E>(f);
  }
}

template <bool VA, bool VB, int EPI>
__global__ __launch_bounds__(THREADS, 2) void gemm_nt_kernel(GemmArgs p) {
  // two LDS buffers of fragments: chunk c is multiplied out of buffer c & 1 while chunk c + 1 is split and written into the other
  constexpr int A_PLANE = KS * 2 * BMP * 16, B_PLANE = KS * 2 * BNP * 16;
  __shared__ __attribute__((aligned(16))) char a_hi[2][A_PLANE], a_lo[2][A_PLANE];     // 128 x 128 tile: 68 KB, two workgroups per CU; 128 x 256: 100 KB, one
  __shared__ __attribute__((aligned(16))) char b_hi[2][B_PLANE], b_lo[2][B_PLANE];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  // Block b runs on XCD b % 8, and an XCD's L2 is its own: the tiles that share an operand slice - the column tiles of one row
  // tile (A rows, fp32 from HBM: the large operand), or all tiles of one K split - are consecutive blocks of ONE XCD, so the
  // slice crosses HBM once.  (Tile = blockIdx.(y, x): the four column tiles of a row landed on four XCDs and a 32 000 x 512 x 512
  // product moved 325 MB instead of 131 - it ran at the HBM bound, 80 us.)
  const int m_tiles = (p.M + BM - 1) / BM, n_tiles = (p.N + BN - 1) / BN;
  const int per_group = p.k_splits > 1 ? m_tiles * n_tiles : n_tiles;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int group = xcd + 8 * (slot / per_group), within = slot % per_group;
  const int zi = p.k_splits > 1 ? group : 0;
  const int mi = p.k_splits > 1 ? within / n_tiles : group;
  const int ni = p.k_splits > 1 ? within % n_tiles : within;
  if (zi >= p.k_splits || mi >= m_tiles) return;                    // (grid padded to whole rounds of 8 groups)
  const int m0 = mi * BM, n0 = ni * BN;
  const int k_begin = zi * p.k_per_split;
  const int k_end = min(p.K, k_begin + p.k_per_split);
  float* C = p.C + size_t(zi) * size_t(p.M) * p.ldc;
  const bool edge_tile = m0 + BM > p.M || n0 + BN > p.N;
  // loader: thread -> (row t / 8 + 64 i, four k at 4 (t % 8))
  constexpr int LROWS = THREADS / (KC / 4), NA = BM / LROWS, NB = BN / LROWS;
  constexpr int NU = NA + NB;            // 16-byte loads = staging units per thread and chunk, spread over the six groups of four MFMAs
  static_assert(NU == 6 || NU == 8, "staging units per chunk");
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
  // unit u of the staging of the chunk at k0 held in register set S -> LDS buffer `buf`, in FOUR slices (convert hi | residuals |
  // convert lo | two LDS writes) so that one slice sits behind every MFMA of a group of four: a wavefront that issues four
  // MFMAs back to back waits for the matrix pipe at each of them and only then reaches its VALU work - the staging then overlaps
  // the last MFMA alone (first version: 2 100 cycles per chunk for 768 cycles of MFMA)
  struct StageTmp { f32x4 v; u32x2 hi, lo; float r0, r1, r2, r3; };
  auto stage_slice = [&](auto set, auto unit, auto slice, StageTmp& t, int buf, int k0) __attribute__((always_inline)) {
    constexpr int S = decltype(set)::value, u = decltype(unit)::value, sl = decltype(slice)::value;
    if constexpr (sl == 0) {
      // (only tiles on the rim of C and the last chunk of a K range hold out-of-range elements: elsewhere no mask)
      const bool rim = edge_tile || k0 + KC > k_end;
      if constexpr (u < NA) t.v = rim ? mask4(regs[S][u], p.M, k_end, m0 + lr + LROWS * u, k0 + lk) : regs[S][u];
      else t.v = rim ? mask4(regs[S][u], p.N, k_end, n0 + lr + LROWS * (u - NA), k0 + lk) : regs[S][u];
      t.hi[0] = pk_bf16(t.v[0], t.v[1]);
      t.hi[1] = pk_bf16(t.v[2], t.v[3]);
    } else if constexpr (sl == 1) {
      t.r0 = t.v[0] - __builtin_bit_cast(float, t.hi[0] << 16);
      t.r1 = t.v[1] - __builtin_bit_cast(float, t.hi[0] & 0xffff0000u);
      t.r2 = t.v[2] - __builtin_bit_cast(float, t.hi[1] << 16);
      t.r3 = t.v[3] - __builtin_bit_cast(float, t.hi[1] & 0xffff0000u);
    } else if constexpr (sl == 2) {
      t.lo[0] = pk_bf16(t.r0, t.r1);
      t.lo[1] = pk_bf16(t.r2, t.r3);
    } else {
      if constexpr (u < NA) {
        const int o = ((frag_off * BMP + lr + LROWS * u) * 8 + (lk & 7)) * 2;
        *reinterpret_cast<u32x2*>(a_hi[buf] + o) = t.hi;
        *reinterpret_cast<u32x2*>(a_lo[buf] + o) = t.lo;
      } else {
        const int o = ((frag_off * BNP + lr + LROWS * (u - NA)) * 8 + (lk & 7)) * 2;
        *reinterpret_cast<u32x2*>(b_hi[buf] + o) = t.hi;
        *reinterpret_cast<u32x2*>(b_lo[buf] + o) = t.lo;
      }
    }
  };
  auto stage_unit = [&](auto set, auto unit, int buf, int k0) __attribute__((always_inline)) {
    StageTmp t;
    stage_slice(set, unit, std::integral_constant<int, 0>{}, t, buf, k0);
    stage_slice(set, unit, std::integral_constant<int, 1>{}, t, buf, k0);
    stage_slice(set, unit, std::integral_constant<int, 2>{}, t, buf, k0);
    stage_slice(set, unit, std::integral_constant<int, 3>{}, t, buf, k0);
  };
  const int wm = wave / WN, wn = wave % WN, j = lane & 31, kg = lane >> 5;
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
  static_for_units<0, NU>([&](auto uu) __attribute__((always_inline)) { stage_unit(I0{}, uu, 0, k_begin); });
  __syncthreads();

  // one chunk: 24 MFMAs per wavefront (2 K-steps x 4 accumulators x 3 products, the four accumulators interleaved so that no
  // MFMA waits for the one before it) with one sixth of the next chunk's staging behind every group of four
  auto iteration = [&](auto parity, int c) __attribute__((always_inline)) {
    constexpr int P = decltype(parity)::value;
    const int k0 = k_begin + c * KC;
    fetch(parity, k0 + 2 * KC);                                    // set P held chunk c: staged during the previous iteration
    __builtin_amdgcn_sched_barrier(0);
    // fragments of BOTH K-steps are requested up front (32 VGPRs more): the second K-step's LDS latency runs under the first's MFMAs
    bf16x8 fah[KS][2], fal[KS][2], fbh[KS][2], fbl[KS][2];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int oa = (((ks * 2 + kg) * BMP) + wm * 64 + i * 32 + j) * 16;
        fah[ks][i] = *reinterpret_cast<const bf16x8*>(a_hi[P] + oa);
        fal[ks][i] = *reinterpret_cast<const bf16x8*>(a_lo[P] + oa);
        const int ob = (((ks * 2 + kg) * BNP) + wn * 64 + i * 32 + j) * 16;
        fbh[ks][i] = *reinterpret_cast<const bf16x8*>(b_hi[P] + ob);
        fbl[ks][i] = *reinterpret_cast<const bf16x8*>(b_lo[P] + ob);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const bf16x8 (&ah)[2] = fah[ks], (&al)[2] = fal[ks], (&bh)[2] = fbh[ks], (&bl)[2] = fbl[ks];
      // group g = 3 ks + term (six per chunk) carries the units [g NU / 6, (g + 1) NU / 6): slice q of each behind MFMA q
      static_for_units<0, 3>([&](auto tt) __attribute__((always_inline)) {
        constexpr int term = decltype(tt)::value;
        StageTmp tmp[2];
        static_for_units<0, 4>([&](auto qq) __attribute__((always_inline)) {
          constexpr int q = decltype(qq)::value, a = q >> 1, b = q & 1;
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(term == 2 ? al[a] : ah[a], term == 1 ? bl[b] : bh[b], acc[a][b], 0, 0, 0);
          auto slices = [&](auto gg) __attribute__((always_inline)) {
            constexpr int g = decltype(gg)::value, u0 = g * NU / 6, u1 = (g + 1) * NU / 6;
            static_assert(u1 - u0 <= 2, "at most two staging units per group of four MFMAs");
            static_for_units<u0, u1>([&](auto uu) __attribute__((always_inline)) {
              stage_slice(std::integral_constant<int, 1 - P>{}, uu, qq, tmp[decltype(uu)::value - u0], 1 - P, k0 + KC);
            });
          };
          if (ks == 0) slices(std::integral_constant<int, term>{}); else slices(std::integral_constant<int, 3 + term>{});
          __builtin_amdgcn_sched_barrier(0);
        });
      });
    }
    __syncthreads();        // buffer 1 - P is complete, everybody has left buffer P
  };
  for (int c = 0; c < n_chunks; c += 2) {
    iteration(I0{}, c);
    iteration(I1{}, c + 1);                                        // (an odd count: one chunk of zeros)
  }

  // ---- epilogue: register q of lane (j, kg) = C[row 8 (q / 4) + 4 kg + q % 4][column j] of its 32 x 32 tile -------------
  // (EPI is a template parameter and the usual E - one row for all of C: a bias - is read once per column: the first version
  // branched per element on the epilogue type and loaded E[m / e_rows][n] 64 times per lane behind 64 waits)
  const bool one_row = p.e_rows >= p.M;
  // Softplus(beta) in base 2 with its two constants out of the loops (a division per element otherwise): max(v, 0) +
  // log2(1 + 2^(c1 |v|)) c2, c1 = -beta log2 e, c2 = ln 2 / beta; beta <= 0: ReLU
  const bool relu = !(p.beta > 0.f);
  const float c1 = relu ? 0.f : -1.44269504f * p.beta, c2 = relu ? 0.f : 0.693147181f / p.beta;
  auto act = [&](float v) __attribute__((always_inline)) {
    const float t = __builtin_amdgcn_exp2f(c1 * fabsf(v));
    const float sp = fmaf(__builtin_amdgcn_logf(1.f + t), c2, fmaxf(v, 0.f));
    return relu ? fmaxf(v, 0.f) : sp;
  };
  auto tile_out = [&](auto one_row_c, auto edge_c) __attribute__((always_inline)) {
    constexpr bool ONE_ROW = decltype(one_row_c)::value, EDGE = decltype(edge_c)::value;   // EDGE: the tile crosses the rim of C
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const int mt = m0 + wm * 64 + a * 32;
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int n = n0 + wn * 64 + b * 32 + j;
        const bool n_ok = n < p.N;
        float e_col = 0.f;
        if constexpr (EPI != 0 && ONE_ROW) e_col = p.E[n_ok ? n : 0];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int m = mt + 8 * (q >> 2) + 4 * kg + (q & 3);
          float v = p.alpha * acc[a][b][q];
          if constexpr (EPI != 0) {
            float e = e_col;
            if constexpr (!ONE_ROW) {
              const int mc = m < p.M ? m : p.M - 1;
              e = p.E[size_t(p.e_rows == 1 ? mc : mc / p.e_rows) * p.N + (n_ok ? n : 0)];
            }
            v += e;
            if constexpr (EPI == 1) v = act(v);
          }
          if constexpr (EDGE) { if (n_ok && m < p.M) C[size_t(m) * p.ldc + n] = v; }
          else C[size_t(m) * p.ldc + n] = v;
        }
      }
    }
  };
  if (one_row) { if (edge_tile) tile_out(std::true_type{}, std::true_type{}); else tile_out(std::true_type{}, std::false_type{}); }
  else tile_out(std::false_type{}, std::true_type{});
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
  a.k_splits = k_splits;
  const int m_tiles = (M + BM - 1) / BM, n_tiles = (N + BN - 1) / BN;
  const long groups = k_splits > 1 ? k_splits : m_tiles, per_group = k_splits > 1 ? long(m_tiles) * n_tiles : n_tiles;
  const long blocks = (groups + 7) / 8 * 8 * per_group;
  if (blocks > 0x7fffffffL) return nphm_fail_msg("nphm_dense_gemm_nt: too many tiles for one launch");
  const dim3 grid((unsigned)blocks);
  // 16-byte loads where rows are 16-byte aligned and every K range is a multiple of 4 (k_per_split is a multiple of 64)
  const bool va = (lda & 3) == 0 && (reinterpret_cast<uintptr_t>(A) & 15) == 0 && (K & 3) == 0;
  const bool vb = (ldb & 3) == 0 && (reinterpret_cast<uintptr_t>(B) & 15) == 0 && (K & 3) == 0;
  hipStream_t st = static_cast<hipStream_t>(stream);
  auto launch = [&](auto va_c, auto vb_c) {
    constexpr bool VA = decltype(va_c)::value, VB = decltype(vb_c)::value;
    if (epilogue == 0) hipLaunchKernelGGL((gemm_nt_kernel<VA, VB, 0>), grid, dim3(THREADS), 0, st, a);
    else if (epilogue == 1) hipLaunchKernelGGL((gemm_nt_kernel<VA, VB, 1>), grid, dim3(THREADS), 0, st, a);
    else hipLaunchKernelGGL((gemm_nt_kernel<VA, VB, 2>), grid, dim3(THREADS), 0, st, a);
  };
  if (va && vb) launch(std::true_type{}, std::true_type{});
  else if (va) launch(std::true_type{}, std::false_type{});
  else if (vb) launch(std::false_type{}, std::true_type{});
  else launch(std::false_type{}, std::false_type{});
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

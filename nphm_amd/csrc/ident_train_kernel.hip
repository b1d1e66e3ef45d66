// ident_train_kernel.hip — the NPHM identity ensemble for TRAINING: member values with their spatial
// gradients, and the reverse sweep of both.
//
// compute_loss (src/NPHM/models/loss_functions.py:20-110) evaluates the decoder, takes
// gradient(pred, x, create_graph=True) (diff_operators.py:6-16) for the normal / eikonal terms and calls
// loss.backward() (training.py:124): a double backward through the 40 member MLPs of
// EnsembledDeepSDF.forward (EnsembledDeepSDF.py:101-126) w.r.t. every weight.  The host splits the field
// into   per-member values f_k(c) and gradients grad_c f_k   (these kernels)   and   the Gaussian blend
// (EnsembledDeepSDF.py:129-150; elementwise, left to autograd), and needs from here
//
//   forward  (train_kernel<false>): f_k and grad_c f_k for the listed (row, member, point) triples
//            = the member's forward + ONE transposed sweep seeded with 1 (f_k is a scalar);
//   backward (train_kernel<true>) : for seeds sbar = dL/df_k and v = dL/d(grad_x f_k) the gradient of
//            phi = sbar f_k + v . grad f_k  w.r.t. the local coordinates, the folded biases of lin0 / the skip
//            layer and, through saved operands, every weight.  By symmetry of the Hessian  v . grad f_k  is the
//            directional derivative of f_k along v, so phi is the output of a forward pass that carries the
//            VALUE stream and ONE TANGENT stream (direction v); its reverse sweep needs sigma'' once per layer:
//                T_l = U_l s_l                       (adjoint of the tangent pre-activation tau_l)
//                D_l = H_l s_l + U_l tau_l s'_l       (adjoint of the value pre-activation d_l)
//                [H_{l-1} | U_{l-1}] = W_l^T [D_l | T_l],      dW_l = D_l h_{l-1}^T + T_l u_{l-1}^T
//            with s = sigma'(d), s' = sigma''(d), h = sigma(d), u = s tau  (tests/test_train_math.py pins
//            these formulas against torch.autograd in float64).
//
// Decomposition as in ident_bwd_kernel.hip: one workgroup = 8 wavefronts = one member x 32 listed points;
// the 64 MFMA columns of a tile are [32 value columns | 32 tangent columns of the same points], so that the
// value and the tangent of one (feature, point) sit in the SAME lane of two accumulator tiles and every
// epilogue is lane-local.  Activations in LDS as split-bf16 K chunks, weights L2 -> VGPR from the forward
// pack / the transposed pack, three MFMA passes per product (fp32-equivalent), scaled domain of layout.h.
// Weight gradients: the reverse kernel stores the operands of  dW_l = sum over columns delta_l (x) input_l  per tile
// as [1409 operand rows][64 columns] fp32 (one contiguous 352 KiB block per tile; a row of 32 columns of one stream
// = 128 coalesced bytes straight from the accumulator layout); wgrad_kernel contracts them over the column axis
// on the MFMA path (K = columns; split-bf16 x3) for chunks of tiles of one weight set (tiles are ordered by set)
// and adds the per-chunk partial sums into the parameter-shaped gradients.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <type_traits>

#include "capi_common.h"
#include "layout.h"
#include "member_common.h"

namespace nphm {
namespace train {

using namespace nphm::bwd;

constexpr float LN2 = 0.6931471805599453f;

// saved operands of the weight gradients of lin1 .. lin3: per tile [SV_ROWS][64] fp32, column = 32 * stream + point.
// (Round 4: lin0 and lin4 left the table - their gradients are 3- and 1-column contractions, which the reverse kernel
// reduces over its 64 columns in registers and hands to edge_grads_kernel as 1001 floats per tile instead of 404 rows
// x 256 bytes: 1409 -> 1005 rows, -29 % of the bytes both kernels move.)
enum { SV_IN1 = 0,    // 200 h0' | u0'
       SV_IN2,        // 104 h1' | u1'  (rows 101..103: coordinates | direction)
       SV_IN3,        // 200 h2' | u2'
       SV_D1,         // 101 D1 | T1
       SV_D2,         // 200 D2 | T2
       SV_D3,         // 200 D3 | T3
       SV_COUNT };
constexpr int SV_ROWS = HID + L2_IN + HID + L1_OUT + HID + HID;      // 1005
__host__ __device__ constexpr int sv_offset(int which) {
  constexpr int off[SV_COUNT] = {0, HID, HID + L2_IN, 2 * HID + L2_IN, 2 * HID + L2_IN + L1_OUT, 3 * HID + L2_IN + L1_OUT};
  return off[which];
}
static_assert(sv_offset(SV_D3) + HID == SV_ROWS, "saved-operand rows");
// per tile, from the reverse kernel: sums over the tile's 64 columns (scaled domain; the edge kernels apply the scales)
//   [0, 600)  dW0[f][c] = sum D0[f] c_in[c] + T0[f] v[c]      [600, 800)  sum of D0[f]'s value columns (folded bias of lin0)
//   [800, 1000)  dW4[f] = sum h3'[f] sbar + u3'[f] valid        [1000]  sum of sbar (lin4's bias)
//   [1024, 1125) / [1152, 1352) / [1352, 1552)  sums of the value columns of D1 / D2 / D3: bias of lin1, folded bias of the
//   skip layer, bias of lin3 (the weight-gradient kernel then has no bias work and no per-pair bookkeeping)
constexpr int EDGE_FLOATS = 1600, EDGE_W0 = 0, EDGE_B0 = 3 * HID, EDGE_W4 = 4 * HID, EDGE_B4 = 5 * HID;
constexpr int EDGE_B1 = 1024, EDGE_B2 = 1152, EDGE_B3 = 1352;
constexpr int EDGE_GA = 1008;     // [1008, 1011): the tile's share of d phi / d anchor_k (minus the sum of its points' d phi / d x)
static_assert(EDGE_B4 < EDGE_B1 && EDGE_B1 + L1_OUT <= EDGE_B2 && EDGE_B2 + HID == EDGE_B3 && EDGE_B3 + HID <= EDGE_FLOATS, "edge record");
// per chunk, from the weight-gradient kernel: its share of lin1 / lin2[:, :104] / lin3 (wgrad_set_kernel sums the chunks of a set)
constexpr int WPART_W1 = 0, WPART_W2 = L1_OUT * HID, WPART_W3 = WPART_W2 + HID * L2_IN, WPART_FLOATS = WPART_W3 + HID * HID;

struct TrainArgs {
  const uint16_t* packed_bf16;
  const float* packed_f32;
  const uint16_t* packed_bwd;
  const float* state;             // [n_rows, LS_ROW_STRIDE]
  const float* xyz;               // [n_rows, n_points, 3]
  const int* tiles;               // [n_tiles][4] = row, member, offset into list, count (<= 64 forward, <= 32 backward)
  const int* list;
  int n_tiles;
  int64_t n_points;
  float* member_sdf;              // forward: [n_rows, n_points, 40]
  float* member_grad;             // forward: [n_rows, n_points, 40, 3]  d f_k / d xyz
  const float* g_sdf;             // backward: [n_rows, n_points, 40]
  const float* g_grad;            // backward: [n_rows, n_points, 40, 3] or NULL (no second-order seed)
  float* gxyz;                    // [n_rows, n_points, 3]  (+=)
  float* ganch;                   // [n_rows, 39, 3]        (+=)
  float* save;                    // backward: [n_tiles][SV_ROWS][64]
  float* edge;                    // backward: [n_tiles][EDGE_FLOATS]
  const float* op_scale;          // operands == 2: [4] = scale of the adjoints' value columns (S_d), of the inputs' tangent
                                  // columns (S_u), of the adjoints' tangent columns (S_t = S_d / S_u), 1 / S_d; word [7] (unsigned,
                                  // in/out): += wavefronts of this launch that clamped a stored operand to the binary16 range
};

// coord_operand without the constant slots: the B operand of the tangent stream at lin0 (no bias)
__device__ __forceinline__ bf16x8 tangent_operand(float x, float y, float z, int h) {
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

// Both kernels work on 64 MFMA columns = two accumulator tiles of 32:
//   forward  (SECOND = false): tile 1 = the NEXT 32 listed points (64 points per workgroup, value stream only);
//   backward (SECOND = true) : tile 1 = the tangent stream of the SAME 32 points.
// Per-layer state kept in registers for the reverse sweep: s = sigma'(d') of tile 0 and, in q, sigma'(d') of tile 1
// (forward) or sigma''(d') tau (backward).
// One workgroup per tile, tiles in table order.  Measured alternatives (32 x 1693-point batch, 28.6 k backward tiles):
// persistent workgroups striding over the table +35 %; XCD-aware order (XCD x walks the x-th eighth of the
// member-ordered table) +6 %; a fully unrolled K loop with 4..12 weight fragments in flight +10..30 %.
// OM: storage of the weight gradients' operands between this kernel and wgrad_kernel: 0 = fp32; 1 = bf16 (half the operand
// traffic of both kernels; the products then carry 8-bit mantissas: parameter gradients to 4e-4 of their largest entry);
// 2 = binary16 with per-stream power-of-two scales (11-bit mantissas: 5e-5).  The four operand streams live 4 .. 12 decades
// apart (inputs' value columns ~1e0, their tangent columns ~1e-9 .. 1e-3, adjoints' value columns ~1e-12 .. 1e-6, their tangent
// columns ~1e-5 .. 1e-1), each tied to the size of the seeds of the step: the host derives S_d, S_u from the seeds' maxima
// (operand_scales_kernel), S_t = S_d / S_u keeps both streams' products on ONE scale (D h + T u, summed over all 64 columns
// by the same MFMAs), wgrad_kernel divides by S_d.  Values beyond the format saturate instead of becoming inf.
template <bool SECOND, int OM = 0>
__global__ __launch_bounds__(64 * WAVES, 2) void train_kernel(TrainArgs p) {
  constexpr int NT = 2;
  constexpr bool O16 = OM != 0;
  constexpr int ES = O16 ? 2 : 4;                    // bytes per stored operand element
  constexpr int PTS = SECOND ? 32 : 64;              // points per tile
  __shared__ __attribute__((aligned(16))) char act_hi[PLANE_BYTES];
  __shared__ __attribute__((aligned(16))) char act_lo[PLANE_BYTES];
  __shared__ float part[WAVES][64];
  __shared__ float pt_c[64][4];               // local coordinates, validity
  __shared__ float pt_v[64][4];               // tangent direction (local frame), seed sbar
  __shared__ float pt_dc[64][4];              // d phi / d local coordinates
  __shared__ int pt_idx[64];
  // backward: the operands of the weight gradients leave through wavefront 7 (idle otherwise): the computing
  // wavefronts drop their [32 rows][64 columns] blocks here, wavefront 7 copies them to HBM with 16-byte stores during
  // the next stage.  Stores issued by the computing wavefronts themselves would sit in the same in-order vmcnt queue
  // as their weight-fragment loads (every stage would begin by waiting for the previous stage's stores to be
  // acknowledged), and 1600 dword stores per tile become 350 16-byte ones.  Measured: -2 % on the kernel (2.73 ->
  // 2.68 ms per 14.3 k tiles) - the stores were not what a stage waits for.
  __shared__ __attribute__((aligned(16))) char stage_buf[SECOND ? 7 * 32 * 64 * ES : 16];

  const int tile_index = blockIdx.x;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int h = lane >> 5, j = lane & 31;
  const int* tile = p.tiles + 4 * tile_index;
  const int row = tile[0], k = tile[1], off = tile[2], cnt = tile[3];
  if (cnt <= 0) return;
  const int set = member_set(k);
  const float* st = p.state + size_t(row) * LS_ROW_STRIDE;
  const float sign_x = (k < 2 * N_SYMM && (k & 1)) ? -1.f : 1.f;
  char* const save = SECOND ? reinterpret_cast<char*>(p.save) + size_t(tile_index) * SV_ROWS * 64 * ES : nullptr;
  // OM == 2: wavefront-wide mask of "a stored operand left the binary16 range" (kept in SGPRs; v_cmp + s_or per value): the
  // clamp below is silent, the count of such wavefronts goes to op_scale[7] (read as unsigned) - the host re-runs the step
  // with fp32 storage when it is not zero (advisor, round 5: scales derived from seed maxima through measured ratios)
  unsigned long long sat = 0ull;
  auto put = [&](char* at, float v) __attribute__((always_inline)) {
    if constexpr (OM == 2) {
      sat |= __ballot(fabsf(v) > 65504.f);
      *reinterpret_cast<_Float16*>(at) = (_Float16)__builtin_amdgcn_fmed3f(v, -65504.f, 65504.f);
    }
    else if constexpr (OM == 1) *reinterpret_cast<__bf16*>(at) = (__bf16)v;
    else *reinterpret_cast<float*>(at) = v;
  };
  // OM == 2: scales of the two streams (tile 0 = value columns, tile 1 = tangent columns) of inputs / adjoints
  float sc_in[2] = {1.f, 1.f}, sc_adj[2] = {1.f, 1.f};
  if constexpr (SECOND && OM == 2) { sc_in[1] = p.op_scale[1]; sc_adj[0] = p.op_scale[0]; sc_adj[1] = p.op_scale[2]; }

  if (threadIdx.x < 64) {
    const int m = threadIdx.x;
    const bool ok = m < cnt && m < PTS;
    const int n = p.list[off + (ok ? m : cnt - 1)];
    const int64_t pt = int64_t(row) * p.n_points + n;
    const float* q = p.xyz + pt * 3;
    float ax = 0.f, ay = 0.f, az = 0.f;
    if (k < N_LOC) { const float* a = st + LS_OFF_ANCH + 3 * k; ax = a[0]; ay = a[1]; az = a[2]; }
    pt_c[m][0] = sign_x * (q[0] - ax); pt_c[m][1] = q[1] - ay; pt_c[m][2] = q[2] - az; pt_c[m][3] = ok ? 1.f : 0.f;
    float vx = 0.f, vy = 0.f, vz = 0.f, sb = ok ? 1.f : 0.f;
    if (SECOND) {
      const int64_t pair = pt * N_MEMBERS + k;
      sb = ok ? p.g_sdf[pair] : 0.f;
      if (ok && p.g_grad) { const float* v = p.g_grad + pair * 3; vx = sign_x * v[0]; vy = v[1]; vz = v[2]; }
    }
    pt_v[m][0] = vx; pt_v[m][1] = vy; pt_v[m][2] = vz; pt_v[m][3] = sb;
    pt_dc[m][0] = 0.f; pt_dc[m][1] = 0.f; pt_dc[m][2] = 0.f; pt_dc[m][3] = 0.f;
    pt_idx[m] = ok ? n : -1;
  }
  __syncthreads();

  // this lane's column of tile 0 and of tile 1: point j and (forward) point 32 + j / (backward) its tangent
  constexpr int J1 = SECOND ? 0 : 32;
  // Per-lane point data (coordinates, direction, seeds) is READ FROM LDS WHERE IT IS USED (four sites), not held: the sweep
  // keeps acc, val and sigma' / sigma'' of four layers (192 VGPRs) beside 64 of weight and activation fragments - ten more
  // registers that merely live from here to stage C were 2 spilled VGPRs + 12 B of scratch in rounds 2-5.
  auto pt4 = [&](const float (*tab)[4], int m) __attribute__((always_inline)) -> f32x4 {
    return *reinterpret_cast<const f32x4*>(tab[m]);
  };
  bf16x8 bv[NT];
  {
    const f32x4 c0 = pt4(pt_c, j), e1 = SECOND ? pt4(pt_v, j) : pt4(pt_c, J1 + j);
    bv[0] = coord_operand(c0[0], c0[1], c0[2], h);
    bv[1] = SECOND ? tangent_operand(e1[0], e1[1], e1[2], h) : coord_operand(e1[0], e1[1], e1[2], h);
  }

  const uint16_t* fw = p.packed_bf16 + size_t(set) * BF_SET_STRIDE;
  const uint16_t* bw = p.packed_bwd + size_t(set) * BWD_SET_STRIDE;
  const float* tails = st + LS_OFF_TAIL + size_t(k) * GEMM_CHUNKS * TAIL_FLOATS;
  const bf16x8* Bh = reinterpret_cast<const bf16x8*>(act_hi) + h * M + j;
  const bf16x8* Bl = reinterpret_cast<const bf16x8*>(act_lo) + h * M + j;
  const f32x16 zero16 = {};

  // out tile `n` of a stage: acc[t] += sum over KS K-steps of A(n, s) x act(s)   (split-bf16 x3).  The A fragments
  // come from L2, several hundred cycles away, with one wavefront pair per SIMD to hide that: four K-steps of them
  // are kept in flight (B fragments from LDS: one step ahead).  Every load is unconditional (indices clamped to the
  // last K-step): a branch around a prefetch makes the compiler's s_waitcnt insertion give up on counting and wait
  // for ALL outstanding loads before each MFMA group - i.e. no prefetch at all.
  auto gemm_tile = [&](f32x16 (&acc)[NT], const uint16_t* frag_base, int n, auto ks_c) __attribute__((always_inline)) {
    constexpr int KS = decltype(ks_c)::value;
    const bf16x8* W = reinterpret_cast<const bf16x8*>(frag_base) + size_t(n) * KS * 128 + lane;
    bf16x8 ah[4], al[4], bh[2][NT], bl[2][NT];
    auto load_a = [&](int slot, int s) __attribute__((always_inline)) {
      s = s < KS - 1 ? s : KS - 1;
      ah[slot] = W[s * 128];
      al[slot] = W[s * 128 + 64];
    };
    auto load_b = [&](int slot, int s) __attribute__((always_inline)) {
      s = s < KS - 1 ? s : KS - 1;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        bh[slot][t] = Bh[2 * s * M + 32 * t];
        bl[slot][t] = Bl[2 * s * M + 32 * t];
      }
    };
    auto mma = [&](int sa, int sb) __attribute__((always_inline)) {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[sa], bh[sb][t], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[sa], bl[sb][t], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[sa], bh[sb][t], acc[t], 0, 0, 0);
      }
    };
    // sched_barrier(0) after every group: hipcc otherwise sinks each load to just before its first use (one
    // register set, s_waitcnt vmcnt(0) in front of every MFMA group)
#pragma unroll
    for (int u = 0; u < 4; ++u) load_a(u, u);
    load_b(0, 0);
    __builtin_amdgcn_sched_barrier(0);
    int s = 0;
#pragma unroll 1
    for (; s + 4 <= KS; s += 4) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        load_b((u + 1) & 1, s + u + 1);
        __builtin_amdgcn_sched_barrier(0);
        mma(u, u & 1);
        __builtin_amdgcn_sched_barrier(0);
        load_a(u, s + u + 4);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#pragma unroll
    for (int u = 0; u < KS % 4; ++u) {             // s = KS - KS % 4 here: slots u hold K-steps s + u
      if (u + 1 < KS % 4) load_b((u + 1) & 1, KS - KS % 4 + u + 1);
      __builtin_amdgcn_sched_barrier(0);
      mma(u, u & 1);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // D tile n -> LDS K chunks 4n + 2*half + h of the columns
  auto store_tile = [&](int n, const f32x16 (&v)[NT]) __attribute__((always_inline)) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      float tmp[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) tmp[r] = v[t][r];
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const Split8 s8 = split8(tmp + 8 * half);
        const int o = ((4 * n + 2 * half + h) * M + 32 * t + j) * 16;
        *reinterpret_cast<bf16x8*>(act_hi + o) = s8.hi;
        *reinterpret_cast<bf16x8*>(act_lo + o) = s8.lo;
      }
    }
  };
  // D tile of this wavefront -> its block of the staging buffer, row = feature - 32 * wave
  auto save_tile = [&](const f32x16 (&v)[NT], const float (&sc)[2]) __attribute__((always_inline)) {
    if (!SECOND) return;
    char* base = stage_buf + (wave * (32 * 64) + (4 * h) * 64 + j) * ES;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
#pragma unroll
      for (int t = 0; t < NT; ++t) put(base + (((r & 3) + 8 * (r >> 2)) * 64 + 32 * t) * ES, OM == 2 ? v[t][r] * sc[t] : v[t][r]);
    }
  };
  // wavefront 7: blocks 0..NB-1 of the staging buffer -> rows 0..ROWS-1 of a saved operand (contiguous in HBM).  All
  // of a block's LDS reads are issued before its stores (fully unrolled, compile-time extents): a read-wait-store loop
  // takes longer than the GEMM stage it is meant to hide behind, and the next barrier then waits for this wavefront.
  auto copy_out = [&](int which, auto rows_c, auto nb_c) __attribute__((always_inline)) {
    if (!SECOND || wave != 7) return;
    constexpr int ROWS = decltype(rows_c)::value, NB = decltype(nb_c)::value;
    char* dst = save + sv_offset(which) * 64 * ES + lane * 16;
    const char* src = stage_buf + lane * 16;
    constexpr int BLK = 32 * 64 * ES;                                        // bytes of a staging block
    static_assert(NB <= 7, "staging blocks");
#pragma unroll
    for (int n = 0; n < NB; ++n) {
      const int bytes = (ROWS - 32 * n < 32 ? ROWS - 32 * n : 32) * 64 * ES;  // of this block, a multiple of 128
      f32x4 v[BLK / 1024];
#pragma unroll
      for (int i = 0; i < BLK / 1024; ++i)
        if (i * 1024 < bytes) v[i] = *reinterpret_cast<const f32x4*>(src + n * BLK + i * 1024);
#pragma unroll
      for (int i = 0; i < BLK / 1024; ++i)
        if (i * 1024 + lane * 16 < bytes) *reinterpret_cast<f32x4*>(dst + n * BLK + i * 1024) = v[i];
    }
  };
  // activation: h' = softplus2(d') (backward, tile 1: u' = s tau); keeps the state of the reverse sweep
  auto activate = [&](const f32x16 (&acc)[NT], f32x16& s, f32x16& q, f32x16 (&val)[NT]) __attribute__((always_inline)) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float sg = sigmoid2(acc[0][r]);
      s[r] = sg;
      val[0][r] = softplus2(acc[0][r]);
      if (SECOND) {
        const float tau = acc[1][r];
        val[1][r] = sg * tau;
        q[r] = LN2 * sg * (1.f - sg) * tau;
      } else {
        q[r] = sigmoid2(acc[1][r]);
        val[1][r] = softplus2(acc[1][r]);
      }
    }
  };
  // reverse through the activation.  backward: D = H s + U s' tau, T = U s ; forward: G_t = (W^T G)_t s_t
  auto deactivate = [&](const f32x16 (&acc)[NT], const f32x16& s, const f32x16& q, f32x16 (&val)[NT]) __attribute__((always_inline)) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if (SECOND) {
        val[0][r] = fmaf(acc[0][r], s[r], acc[1][r] * q[r]);
        val[1][r] = acc[1][r] * s[r];
      } else {
        val[0][r] = acc[0][r] * s[r];
        val[1][r] = acc[1][r] * q[r];
      }
    }
  };

  f32x16 acc[NT], val[NT], s0, s1, s2, s3, q0, q1, q2, q3;
  s0 = zero16; s1 = zero16; s2 = zero16; s3 = zero16; q0 = zero16; q1 = zero16; q2 = zero16; q3 = zero16;

  float* const edge = SECOND ? p.edge + size_t(tile_index) * EDGE_FLOATS : nullptr;
  // sum over the tile's value columns of the 16 rows this lane's registers hold (bias gradients): lane j = r keeps row r's
  auto edge_row_sums = [&](const f32x16& v, int off, int limit) __attribute__((always_inline)) {
    float o = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float sum = half_wave_sum(v[r]);
      o = j == r ? sum : o;
    }
    const int f = feat_of(wave, j & 15, h);
    if (j < 16 && f < limit) edge[off + f] = o;
  };
  if (SECOND && wave == 7 && h == 0) {             // lin4's bias gradient: the value seeds of the tile
    const float sb = half_wave_sum(pt_v[j][3]);
    if (j == 0) edge[EDGE_B4] = sb;
  }

  // ================================ forward =======================================================
  // L0: lin0 on the local coordinates (folded bias inside the block); tangent stream: lin0[:, :3] v
  if (wave < 7) {
    const bf16x8* A0 = reinterpret_cast<const bf16x8*>(st + LS_OFF_L0B + size_t(k) * L0_BLOCK_FLOATS) + lane;
    const bf16x8 a = A0[wave * 64];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bv[t], zero16, 0, 0, 0);
    activate(acc, s0, q0, val);
    store_tile(wave, val);
    save_tile(val, sc_in);
  }
  __syncthreads();
  copy_out(SV_IN1, std::integral_constant<int, HID>{}, std::integral_constant<int, 7>{});
  // L1: 200 -> 101 (4 tiles); the skip coordinates (tangent: the direction) join tile 3 as features 101..103
  if (wave < L1_OB) {
    acc[0] = load_frag16(tails + wave * TAIL_FLOATS + h * 16);
    acc[1] = SECOND ? zero16 : acc[0];
    gemm_tile(acc, fw + BF_OFF_L1A, wave, std::integral_constant<int, L1_KS16>{});
    activate(acc, s1, q1, val);
    if (wave == L1_OB - 1) {
      // what the tiles carry into the skip features 101..103: coordinates | (backward) the direction / (forward) tile 1's coordinates
      const f32x4 c0 = pt4(pt_c, j), e1 = SECOND ? pt4(pt_v, j) : pt4(pt_c, J1 + j);
      val[0][1] = h ? c0[0] : val[0][1]; val[0][2] = h ? c0[1] : val[0][2]; val[0][3] = h ? c0[2] : val[0][3];
      val[1][1] = h ? e1[0] : val[1][1]; val[1][2] = h ? e1[1] : val[1][2]; val[1][3] = h ? e1[2] : val[1][3];
    }
  }
  __syncthreads();                       // every wavefront has read a0
  if (wave < L1_OB) { store_tile(wave, val); save_tile(val, sc_in); }
  __syncthreads();
  copy_out(SV_IN2, std::integral_constant<int, L2_IN>{}, std::integral_constant<int, L1_OB>{});
  // L2: 104 -> 200
  if (wave < 7) {
    acc[0] = load_frag16(tails + (L1_OB + wave) * TAIL_FLOATS + h * 16);
    acc[1] = SECOND ? zero16 : acc[0];
    gemm_tile(acc, fw + BF_OFF_L2A, wave, std::integral_constant<int, L2_KS16>{});
    activate(acc, s2, q2, val);
  }
  __syncthreads();
  if (wave < 7) { store_tile(wave, val); save_tile(val, sc_in); }
  __syncthreads();
  copy_out(SV_IN3, std::integral_constant<int, HID>{}, std::integral_constant<int, 7>{});
  // L3: 200 -> 200, lin4 fused: f = sum h3' * w4 / k + b4
  f32x16 w4v = zero16;
  if (wave < 7) {
    const float* tl = tails + (L1_OB + L2_OB + wave) * TAIL_FLOATS;
    acc[0] = load_frag16(tl + h * 16);
    acc[1] = SECOND ? zero16 : acc[0];
    gemm_tile(acc, fw + BF_OFF_L3A, wave, std::integral_constant<int, L3_KS16>{});
    w4v = load_frag16(tl + 32 + h * 16);
    activate(acc, s3, q3, val);
    if (!SECOND) {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        float partial = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) partial = fmaf(val[t][r], w4v[r], partial);
        partial += __shfl_xor(partial, 32);
        if (h == 0) part[wave][32 * t + j] = partial;
      }
    } else {
      // lin4's weight gradient: dW4[f] = sum over the points of h3'[f] sbar + u3'[f] (valid): this lane holds point j.
      // The sum of register r is kept by lane j = r: one store of 16 lanes per half-wave instead of sixteen of one lane.
      float o4 = 0.f;
      const float seed = pt_v[j][3], valid = pt_c[j][3];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float sum = half_wave_sum(fmaf(val[0][r], seed, val[1][r] * valid));
        o4 = j == r ? sum : o4;
      }
      const int f4 = feat_of(wave, j & 15, h);
      if (j < 16 && f4 < HID) edge[EDGE_W4 + f4] = o4;
    }
  }
  __syncthreads();
  if (!SECOND && threadIdx.x < 64) {
    const int m = threadIdx.x;
    float f = p.packed_f32[size_t(set) * SET_STRIDE + OFF_L4B];
#pragma unroll
    for (int w = 0; w < 7; ++w) f += part[w][m];
    if (pt_idx[m] >= 0) p.member_sdf[(int64_t(row) * p.n_points + pt_idx[m]) * N_MEMBERS + k] = f;
  }

  // ================================ reverse =======================================================
  // backward: output seeds H3 = sbar w4/k, U3 = w4/k (valid columns) -> D3 = H3 s3 + U3 s3' tau3, T3 = U3 s3
  // forward : G3 = w4/k s3 for both point tiles (seed 1 on valid points)
  if (wave < 7) {
    const float seed = pt_v[j][3], valid = pt_c[j][3], seed1 = pt_v[J1 + j][3];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if (SECOND) {
        val[0][r] = w4v[r] * fmaf(seed, s3[r], valid * q3[r]);
        val[1][r] = valid * w4v[r] * s3[r];
      } else {
        val[0][r] = seed * w4v[r] * s3[r];
        val[1][r] = seed1 * w4v[r] * q3[r];
      }
    }
    store_tile(wave, val);               // a2 is no longer needed (every wavefront passed the barriers above)
    if (SECOND) edge_row_sums(val[0], EDGE_B3, HID);
  }
  if (SECOND) {                          // ... and h2' | u2' have left the staging buffer
    __syncthreads();
    if (wave < 7) save_tile(val, sc_adj);
  }
  __syncthreads();
  copy_out(SV_D3, std::integral_constant<int, HID>{}, std::integral_constant<int, 7>{});
  // stage A: [H2 | U2] = lin3^T [D3 | T3]   (bias gradients: row sums of the stored adjoints, taken by wgrad_kernel)
  if (wave < 7) {
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = zero16;
    gemm_tile(acc, bw + OFF_A, wave, std::integral_constant<int, A_KS>{});
    deactivate(acc, s2, q2, val);
    if (SECOND) edge_row_sums(val[0], EDGE_B2, HID);
  }
  __syncthreads();
  if (wave < 7) { store_tile(wave, val); save_tile(val, sc_adj); }
  __syncthreads();
  copy_out(SV_D2, std::integral_constant<int, HID>{}, std::integral_constant<int, 7>{});
  // stage B: rows 0..100: [H1 | U1] = (lin2a / sqrt2)^T [D2 | T2]; rows 101..103: d phi / d coords (skip path)
  if (wave < B_OB) {
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = zero16;
    gemm_tile(acc, bw + OFF_B, wave, std::integral_constant<int, B_KS>{});
    deactivate(acc, s1, q1, val);
    if (wave == B_OB - 1 && h) {          // features 101..103 = registers 1..3 of the upper half-wave
      pt_dc[j][0] = acc[0][1]; pt_dc[j][1] = acc[0][2]; pt_dc[j][2] = acc[0][3];
      if (!SECOND) { pt_dc[32 + j][0] = acc[1][1]; pt_dc[32 + j][1] = acc[1][2]; pt_dc[32 + j][2] = acc[1][3]; }
#pragma unroll
      for (int t = 0; t < NT; ++t) { val[t][1] = 0.f; val[t][2] = 0.f; val[t][3] = 0.f; }
    }
    if (SECOND) edge_row_sums(val[0], EDGE_B1, L1_OUT);
  }
  __syncthreads();
  if (wave < B_OB) { store_tile(wave, val); save_tile(val, sc_adj); }
  __syncthreads();
  copy_out(SV_D1, std::integral_constant<int, L1_OUT>{}, std::integral_constant<int, B_OB>{});
  // stage C: [H0 | U0] = lin1^T [D1 | T1]
  if (wave < 7) {
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = zero16;
    gemm_tile(acc, bw + OFF_C, wave, std::integral_constant<int, C_KS>{});
    deactivate(acc, s0, q0, val);
    if (SECOND) {
      // lin0: dW0[f][c] = sum over the points of D0[f] c_in[c] + T0[f] v[c]; folded bias: sum of D0[f]
      float ox = 0.f, oy = 0.f, oz = 0.f, ob = 0.f;          // (lane j = r keeps register r's sums, as for lin4)
      const f32x4 c0 = pt4(pt_c, j), v0 = pt4(pt_v, j);
      const float cx = c0[0], cy = c0[1], cz = c0[2], vx = v0[0], vy = v0[1], vz = v0[2];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float d = val[0][r], t = val[1][r];
        const float sx = half_wave_sum(fmaf(d, cx, t * vx)), sy = half_wave_sum(fmaf(d, cy, t * vy));
        const float sz = half_wave_sum(fmaf(d, cz, t * vz)), sb = half_wave_sum(d);
        ox = j == r ? sx : ox; oy = j == r ? sy : oy; oz = j == r ? sz : oz; ob = j == r ? sb : ob;
      }
      const int f0 = feat_of(wave, j & 15, h);
      if (j < 16 && f0 < HID) {
        edge[EDGE_W0 + 3 * f0] = ox; edge[EDGE_W0 + 3 * f0 + 1] = oy; edge[EDGE_W0 + 3 * f0 + 2] = oz;
        edge[EDGE_B0 + f0] = ob;
      }
    }
  }
  __syncthreads();
  if (wave < 7) store_tile(wave, val);
  __syncthreads();
  // stage D: d phi / d coords (lin0 path) = (k lin0[:, :3])^T D0 ; one tile, wavefront 0
  if (wave == 0) {
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = zero16;
    gemm_tile(acc, bw + OFF_D, 0, std::integral_constant<int, D_KS>{});
    if (h == 0) {                          // rows 0..2 of the tile = registers 0..2 of the lower half-wave
      pt_dc[j][0] += acc[0][0]; pt_dc[j][1] += acc[0][1]; pt_dc[j][2] += acc[0][2];
      if (!SECOND) { pt_dc[32 + j][0] += acc[1][0]; pt_dc[32 + j][1] += acc[1][1]; pt_dc[32 + j][2] += acc[1][2]; }
    }
  }
  __syncthreads();

  // ---- per point: back to the global frame ---------------------------------------------------------
  if (threadIdx.x < 64) {
    const int m = threadIdx.x;
    const int n = pt_idx[m];
    float gq[3] = {0.f, 0.f, 0.f};
    if (n >= 0) { gq[0] = sign_x * pt_dc[m][0]; gq[1] = pt_dc[m][1]; gq[2] = pt_dc[m][2]; }   // c = flip (q - a)
    if (!SECOND) {
      if (n >= 0) {
        float* o = p.member_grad + ((int64_t(row) * p.n_points + n) * N_MEMBERS + k) * 3;
        o[0] = gq[0]; o[1] = gq[1]; o[2] = gq[2];
      }
    } else {
      if (n >= 0) {
        float* o = p.gxyz + (int64_t(row) * p.n_points + n) * 3;
        atomicAdd(o, gq[0]); atomicAdd(o + 1, gq[1]); atomicAdd(o + 2, gq[2]);
      }
      float sx = gq[0], sy = gq[1], sz = gq[2];      // anchor gradient: minus the sum over the tile's points
#pragma unroll
      for (int o2 = 16; o2 > 0; o2 >>= 1) { sx += __shfl_xor(sx, o2); sy += __shfl_xor(sy, o2); sz += __shfl_xor(sz, o2); }
      // (per-tile record, summed per (member, row) pair in table order by edge_pair_kernel: no atomics on the anchors, whose
      // gradient reaches mlp_pos and the latent codes)
      if (m == 0) { edge[EDGE_GA] = -sx; edge[EDGE_GA + 1] = -sy; edge[EDGE_GA + 2] = -sz; }
    }
  }
  if constexpr (SECOND && OM == 2) {
    if (sat != 0ull && lane == 0) atomicAdd(reinterpret_cast<unsigned*>(const_cast<float*>(p.op_scale)) + 7, 1u);
  }
}

}  // namespace train
}  // namespace nphm

namespace nphm {
namespace train {

// ---- weight gradients ---------------------------------------------------------------------------------------
// One workgroup = one chunk (<= a few dozen tiles of ONE weight set) x one of lin1 .. lin3 (lin0 / lin4: edge_grads_kernel).  Per tile: the layer's INPUT
// operand [rows_in][64 columns] goes through LDS as split-bf16 (every wavefront needs all of it), the ADJOINT
// operand rows of wavefront w (output block w) come straight from HBM as the MFMA A fragments (lane = row, 8
// consecutive columns); acc[b] += A x B over the tile's 4 K-steps of 16 columns for the 7 input blocks b.
struct WgradArgs {
  const float* saved;         // [n_tiles][SV_ROWS][64]
  const int* chunks;          // [n_chunks][4] = weight set, first tile, number of tiles, -
  float* part;                // [n_chunks][WPART_FLOATS]: every chunk's share of lin1 / lin2[:, :104] / lin3 (written in full)
  const float* op_scale;      // operands == 2: TrainArgs::op_scale
};

typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
constexpr int WG_ROW_BYTES = 64 * 2 + 16;            // bf16 row of 64 columns, padded against bank conflicts
constexpr int WG_PLANE = 224 * WG_ROW_BYTES;

__device__ __forceinline__ Split8 split8v(const f32x4& a, const f32x4& b) {
  float x[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  return split8(x);
}

template <int OM>
__global__ __launch_bounds__(64 * WAVES, 2) void wgrad_kernel(WgradArgs p) {
  constexpr bool O16 = OM != 0;
  constexpr int ES = O16 ? 2 : 4;
  // one-pass contraction of 16-bit operands: bf16 (OM 1) or binary16 (OM 2) MFMA
  auto mfma_o16 = [](const bf16x8& a, const bf16x8& b, const f32x16& c) __attribute__((always_inline)) {
    if constexpr (OM == 2) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  };
  __shared__ __attribute__((aligned(16))) char in_hi[WG_PLANE];
  __shared__ __attribute__((aligned(16))) char in_lo[O16 ? 16 : WG_PLANE];
  const char* const saved = reinterpret_cast<const char*>(p.saved);
  const int* ch = p.chunks + 4 * blockIdx.x;
  const int set = ch[0], tile0 = ch[1], n_tiles = ch[2];
  const int layer = blockIdx.y + 1;          // lin1 .. lin3
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, j = lane & 31;

  // layer geometry
  const int d_which = layer == 1 ? SV_D1 : layer == 2 ? SV_D2 : SV_D3;
  const int i_which = layer == 1 ? SV_IN1 : layer == 2 ? SV_IN2 : SV_IN3;
  const int rows_out = layer == 1 ? L1_OUT : HID;
  const int rows_in = layer == 2 ? L2_IN : HID;
  const int nb_in = (rows_in + 31) / 32;
  const bool active = 32 * wave < rows_out;
  const int orow = 32 * wave + j;                            // this lane's adjoint row (A operand)
  const bool row_ok = active && orow < rows_out;

  f32x16 acc[7];
#pragma unroll
  for (int b = 0; b < 7; ++b) acc[b] = f32x16{};

  // staging registers: the tile's input operand (7 passes of 32 rows x 16 float4) and this lane's adjoint row
  // (bf16 operands: 4 columns = 8 bytes per thread of the input operand, 8 columns = one 16-byte A fragment per K-step)
  typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
  f32x4 in_reg[O16 ? 1 : 7], d_reg[O16 ? 1 : 8];
  bf16x4 in_reg16[O16 ? 7 : 1];
  bf16x8 d_reg16[O16 ? 4 : 1];
  const int srow = tid >> 4, sc4 = tid & 15;
  auto fetch = [&](int t) __attribute__((always_inline)) {
    const char* blk = saved + size_t(tile0 + t) * SV_ROWS * 64 * ES;
#pragma unroll
    for (int q = 0; q < 7; ++q) {
      const int r = 32 * q + srow;
      const char* at = blk + ((sv_offset(i_which) + r) * 64 + 4 * sc4) * ES;
      if (O16) in_reg16[q] = r < rows_in ? *reinterpret_cast<const bf16x4*>(at) : bf16x4{};
      else in_reg[q] = r < rows_in ? *reinterpret_cast<const f32x4*>(at) : f32x4{};
    }
    const char* dr = blk + ((sv_offset(d_which) + (row_ok ? orow : 0)) * 64 + 8 * h) * ES;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      if (O16) {
        d_reg16[s] = row_ok ? *reinterpret_cast<const bf16x8*>(dr + 16 * s * ES) : bf16x8{};
      } else {
        d_reg[2 * s] = row_ok ? *reinterpret_cast<const f32x4*>(dr + 16 * s * ES) : f32x4{};
        d_reg[2 * s + 1] = row_ok ? *reinterpret_cast<const f32x4*>(dr + (16 * s + 4) * ES) : f32x4{};
      }
    }
  };

  if (n_tiles > 0) fetch(0);
  for (int t = 0; t < n_tiles; ++t) {
    __syncthreads();                                   // the previous tile's LDS operand has been consumed
#pragma unroll
    for (int q = 0; q < 7; ++q) {
      const int r = 32 * q + srow;
      if (O16) {
        *reinterpret_cast<bf16x4*>(in_hi + r * WG_ROW_BYTES + 8 * sc4) = in_reg16[q];
      } else {
        __bf16 hi[4], lo[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) { hi[e] = (__bf16)in_reg[q][e]; lo[e] = (__bf16)(in_reg[q][e] - (float)hi[e]); }
        *reinterpret_cast<bf16x4*>(in_hi + r * WG_ROW_BYTES + 8 * sc4) = bf16x4{hi[0], hi[1], hi[2], hi[3]};
        *reinterpret_cast<bf16x4*>(in_lo + r * WG_ROW_BYTES + 8 * sc4) = bf16x4{lo[0], lo[1], lo[2], lo[3]};
      }
    }
    Split8 a[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      if (O16) a[s].hi = d_reg16[s];
      else a[s] = split8v(d_reg[2 * s], d_reg[2 * s + 1]);
    }
    __syncthreads();
    if (t + 1 < n_tiles) fetch(t + 1);                 // next tile's operands fly during the MFMAs
    if (active) {
#pragma unroll
      for (int b = 0; b < 7; ++b) {
        if (b < nb_in) {
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const int o = (32 * b + j) * WG_ROW_BYTES + (16 * s + 8 * h) * 2;
            const bf16x8 bh = *reinterpret_cast<const bf16x8*>(in_hi + o);
            if constexpr (O16) acc[b] = mfma_o16(a[s].hi, bh, acc[b]);
            else acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s].hi, bh, acc[b], 0, 0, 0);
            if (!O16) {
              const bf16x8 bl = *reinterpret_cast<const bf16x8*>(in_lo + o);
              acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s].hi, bl, acc[b], 0, 0, 0);
              acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s].lo, bh, acc[b], 0, 0, 0);
            }
          }
        }
      }
    }
  }
  if (!active) return;

  // scaled-domain products -> this chunk's share of the parameter gradient (plain stores: wgrad_set_kernel adds the chunks
  // of a weight set in chunk order - no atomics, bitwise reproducible)
  float* gw = p.part + size_t(blockIdx.x) * WPART_FLOATS + (layer == 1 ? WPART_W1 : layer == 2 ? WPART_W2 : WPART_W3);
#pragma unroll
  for (int b = 0; b < 7; ++b) {
    if (b < nb_in) {
      const int icol = 32 * b + j;
      float scale = OM == 2 ? p.op_scale[3] : 1.f;          // (binary16 operands carry S_d)
      if (layer == 2) scale *= (icol >= L1_OUT ? SP_SCALE : 1.f) / INV_SQRT2_DIV;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (row < rows_out && icol < rows_in) gw[size_t(row) * rows_in + icol] = acc[b][r] * scale;
      }
    }
  }
}

// sums the chunks' shares of lin1 / lin2[:, :104] / lin3 per weight set, in chunk order, into the parameter-shaped gradients
struct WsetArgs {
  const float* part;          // [n_chunks][WPART_FLOATS]
  const int* set_chunk_first; // [N_SETS + 1]
  float* gW1; float* gW2; float* gW3;   // lin1.weight [sets,101,200], lin2.weight [sets,200,200] (columns 0..103), lin3.weight [sets,200,200]  (+=)
};
__global__ __launch_bounds__(256) void wgrad_set_kernel(WsetArgs p) {
  const int set = blockIdx.y, e = blockIdx.x * blockDim.x + threadIdx.x;
  const int c0 = p.set_chunk_first[set], n = p.set_chunk_first[set + 1] - c0;
  if (e >= WPART_FLOATS || n <= 0) return;
  const float* src = p.part + size_t(c0) * WPART_FLOATS + e;
  float acc = 0.f;
  for (int c = 0; c < n; c += 8) {
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = c + i < n ? src[size_t(c + i) * WPART_FLOATS] : 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) acc += v[i];
  }
  if (e < WPART_W2) p.gW1[size_t(set) * L1_OUT * HID + e] += acc;
  else if (e < WPART_W3) { const int x = e - WPART_W2; p.gW2[(size_t(set) * HID + x / L2_IN) * HID + x % L2_IN] += acc; }
  else p.gW3[size_t(set) * HID * HID + (e - WPART_W3)] += acc;
}

// ---- lin0 / lin4: the reverse kernel's per-tile sums -> parameter gradients -----------------------------------------------
// Fixed summation order, no atomics (bitwise reproducible), three small launches:
//   edge_chunk_kernel : one workgroup per chunk of the weight-gradient work list (<= 32 consecutive tiles of ONE weight set)
//                       adds the chunk's records in tile order -> part[chunk][EDGE_FLOATS]            (~900 workgroups)
//   edge_set_kernel   : one workgroup per weight set adds its chunks' partial sums in chunk order (the chunk table is ordered
//                       by tile index: a set's chunks are consecutive) into lin0.weight[:, :3], lin4.weight, lin4.bias
//   edge_pair_kernel  : one workgroup per (member, row) pair adds the folded-bias sums of the pair's tiles (consecutive in
//                       the member-ordered table) into grad_b0[row, member]
// (A first form - one workgroup per weight set walking its ~1 200 records alone - took 0.44 ms: 24 latency-bound workgroups.)
struct EdgeArgs {
  const float* edge;          // [n_tiles][EDGE_FLOATS]
  const int* chunks;          // [n_chunks][4] = weight set, first tile relative to its piece, tiles, piece
  int ring_tiles;             // tiles per piece
  float* part;                // [n_chunks][EDGE_FLOATS]
  const int* set_chunk_first; // [N_SETS + 1]
  const int* pair_first;      // [n_pairs + 1] first tile of pair (member * n_rows + row)
  int n_rows;
  float* gW0; float* gW4; float* gb4;   // lin0.weight [sets, 200, 99] (columns 0..2), lin4.weight [sets, 200], lin4.bias [sets]  (+=)
  float* gb1; float* gb3;     // lin1.bias [sets, 101], lin3.bias [sets, 200]  (+=)
  float* gb0; float* gb2;     // folded biases of lin0 / the skip layer [n_rows, 40, 200]  (+=)
  float* ganch;               // [n_rows, 39, 3]  (+=)
};
// elements of the record that belong to the weight set (the folded biases belong to a (row, member) pair)
__device__ __forceinline__ bool edge_set_level(int e) {
  return e < EDGE_B0 || (e >= EDGE_W4 && e <= EDGE_B4) || (e >= EDGE_B1 && e < EDGE_B1 + L1_OUT) || (e >= EDGE_B3 && e < EDGE_B3 + HID);
}
__global__ __launch_bounds__(1024) void edge_chunk_kernel(EdgeArgs p) {
  const int* ch = p.chunks + 4 * blockIdx.x;
  const int t0 = ch[3] * p.ring_tiles + ch[1], n = ch[2];
  for (int e = threadIdx.x; e < EDGE_FLOATS; e += blockDim.x) {
    if (!edge_set_level(e)) continue;
    const float* src = p.edge + size_t(t0) * EDGE_FLOATS + e;
    float acc = 0.f;
    for (int t = 0; t < n; t += 8) {
      float v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = t + i < n ? src[size_t(t + i) * EDGE_FLOATS] : 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) acc += v[i];
    }
    p.part[size_t(blockIdx.x) * EDGE_FLOATS + e] = acc;
  }
}
__global__ __launch_bounds__(1024) void edge_set_kernel(EdgeArgs p) {
  const int set = blockIdx.x;
  const int c0 = p.set_chunk_first[set], n = p.set_chunk_first[set + 1] - c0;
  if (n <= 0) return;
  for (int e = threadIdx.x; e < EDGE_FLOATS; e += blockDim.x) {
    if (!edge_set_level(e)) continue;
    const float* src = p.part + size_t(c0) * EDGE_FLOATS + e;
    float acc = 0.f;
    for (int c = 0; c < n; c += 8) {
      float v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = c + i < n ? src[size_t(c + i) * EDGE_FLOATS] : 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) acc += v[i];
    }
    if (e < EDGE_B0) p.gW0[(size_t(set) * HID + e / 3) * D_IN + e % 3] += acc * SP_SCALE;
    else if (e < EDGE_B4) p.gW4[size_t(set) * HID + (e - EDGE_W4)] += acc / SP_SCALE;
    else if (e == EDGE_B4) p.gb4[set] += acc;
    else if (e < EDGE_B2) p.gb1[size_t(set) * L1_OUT + (e - EDGE_B1)] += acc * SP_SCALE;
    else p.gb3[size_t(set) * HID + (e - EDGE_B3)] += acc * SP_SCALE;
  }
}
__global__ __launch_bounds__(256) void edge_pair_kernel(EdgeArgs p) {
  const int pair = blockIdx.x, f = threadIdx.x;            // pair = member * n_rows + row (table order)
  const int t0 = p.pair_first[pair], n = p.pair_first[pair + 1] - t0;
  if (f >= HID || n <= 0) return;
  const float* src = p.edge + size_t(t0) * EDGE_FLOATS + f;
  float a0 = 0.f, a2 = 0.f;
  for (int t = 0; t < n; t += 4) {
    float v0[4], v2[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      v0[i] = t + i < n ? src[size_t(t + i) * EDGE_FLOATS + EDGE_B0] : 0.f;
      v2[i] = t + i < n ? src[size_t(t + i) * EDGE_FLOATS + EDGE_B2] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) { a0 += v0[i]; a2 += v2[i]; }
  }
  const int member = pair / p.n_rows, row = pair % p.n_rows;
  const size_t o = (size_t(row) * N_MEMBERS + member) * HID + f;
  p.gb0[o] += a0 * SP_SCALE;
  p.gb2[o] += a2 * SP_SCALE;
  if (f < 3 && member < N_LOC) {                          // the pair's anchor gradient
    float g = 0.f;
    for (int t = 0; t < n; ++t) g += src[size_t(t) * EDGE_FLOATS + EDGE_GA];
    p.ganch[(size_t(row) * N_LOC + member) * 3 + f] += g;
  }
}

// ---- the Gaussian blend with its spatial gradient, and their backward ------------------------------------------
// pred = sum_k what_k S_k (EnsembledDeepSDF.py:129-150: w_k = exp(-(|x - a_k| + 1e-5)^2 / 0.01), background weight
// exp(-20), what = w / (sum w + 1e-6)) and  grad = d pred / d x = sum_k what_k G_k + sum_k (S_k - pred) what_k u_k,
// u_k = d log w_k / d x = -2 (r_k + eps) / (sigma r_k) (x - a_k), for member values S and gradients G = dS/dx that
// nphm_identity_train_forward produced.  The backward takes seeds (pbar, qbar) for (pred, grad) and returns
//   Sbar_k = what_k (pbar + qbar . (u_k - m)),  m = sum_j what_j u_j          Gbar_k = what_k qbar
//   dL/de_k = what_k u_k [pbar (S_k - pred) + qbar . (G_k - gt) + (S_k - pred)(t_k - tau) - theta]
//             + (S_k - pred) what_k (c_k qbar + kappa_k (qbar . e_k) e_k),    e_k = x - a_k, t_k = qbar . u_k,
//   xbar = sum_k dL/de_k,  abar_k = - sum over points dL/de_k
// (S, G held fixed: their own dependence on x is nphm_identity_train_backward's business; tests/test_train_math.py
// pins the formulas against float64 autograd).  One thread per point, the row's anchors in LDS.
struct BlendArgs {
  const float* xyz; const float* anchors; const float* S; const float* G;
  int64_t n_points;
  float* pred; float* grad;                      // forward
  const float* g_pred; const float* g_grad;      // backward seeds (g_grad may be NULL)
  float* gS; float* gG; float* gxyz; float* ganch;
  float* ga_part;                                // backward: [n_rows][blocks][39 * 3] per-block anchor gradients
};

constexpr float BL_SIGMA = 0.01f, BL_EPS = 1e-5f;

template <bool BWD>
__global__ __launch_bounds__(256) void blend_kernel(BlendArgs p) {
  __shared__ float anch[N_LOC * 3];
  __shared__ float ga[4][N_LOC * 3];          // per wavefront: the block's anchor gradient is their sum in wavefront order
  const int row = blockIdx.y;
  for (int i = threadIdx.x; i < N_LOC * 3; i += blockDim.x) anch[i] = p.anchors[size_t(row) * N_LOC * 3 + i];
  __syncthreads();
  const int64_t n = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  const bool live = n < p.n_points;
  const int64_t pt = int64_t(row) * p.n_points + (live ? n : 0);
  const float qx = p.xyz[pt * 3], qy = p.xyz[pt * 3 + 1], qz = p.xyz[pt * 3 + 2];
  const float* S = p.S + pt * N_MEMBERS;
  const float* G = p.G + pt * N_MEMBERS * 3;
  const float w_bg = expf(-0.2f / BL_SIGMA);

  float w[N_MEMBERS];
  float D = w_bg + 1e-6f;
#pragma unroll
  for (int k = 0; k < N_LOC; ++k) {
    const float ex = qx - anch[3 * k], ey = qy - anch[3 * k + 1], ez = qz - anch[3 * k + 2];
    const float rho = sqrtf(ex * ex + ey * ey + ez * ez) + BL_EPS;
    w[k] = expf(-(rho * rho) / BL_SIGMA);
    D += w[k];
  }
  w[N_LOC] = w_bg;
  const float inv = 1.f / D;
  float pr = 0.f, gtx = 0.f, gty = 0.f, gtz = 0.f, mx = 0.f, my = 0.f, mz = 0.f;
#pragma unroll
  for (int k = 0; k < N_MEMBERS; ++k) {
    w[k] *= inv;
    pr = fmaf(w[k], S[k], pr);
    gtx = fmaf(w[k], G[3 * k], gtx); gty = fmaf(w[k], G[3 * k + 1], gty); gtz = fmaf(w[k], G[3 * k + 2], gtz);
  }
  // u_k = c_k e_k ; q_w = sum (S_k - pred) what_k u_k ; m = sum what_k u_k
  float qwx = 0.f, qwy = 0.f, qwz = 0.f;
#pragma unroll
  for (int k = 0; k < N_LOC; ++k) {
    const float ex = qx - anch[3 * k], ey = qy - anch[3 * k + 1], ez = qz - anch[3 * k + 2];
    const float r = sqrtf(ex * ex + ey * ey + ez * ez);
    const float c = r > 0.f ? -2.f * (r + BL_EPS) / (BL_SIGMA * r) : 0.f;
    const float a = w[k] * c, b = (S[k] - pr) * a;
    mx = fmaf(a, ex, mx); my = fmaf(a, ey, my); mz = fmaf(a, ez, mz);
    qwx = fmaf(b, ex, qwx); qwy = fmaf(b, ey, qwy); qwz = fmaf(b, ez, qwz);
  }
  if (!BWD) {
    if (live) {
      p.pred[pt] = pr;
      p.grad[pt * 3] = gtx + qwx; p.grad[pt * 3 + 1] = gty + qwy; p.grad[pt * 3 + 2] = gtz + qwz;
    }
    return;
  }
  const float pb = live ? p.g_pred[pt] : 0.f;
  float bx = 0.f, by = 0.f, bz = 0.f;
  if (live && p.g_grad) { bx = p.g_grad[pt * 3]; by = p.g_grad[pt * 3 + 1]; bz = p.g_grad[pt * 3 + 2]; }
  const float tau = bx * mx + by * my + bz * mz;
  const float theta = bx * qwx + by * qwy + bz * qwz;
  const float qgt = bx * gtx + by * gty + bz * gtz;
  float* gS = p.gS + pt * N_MEMBERS;
  float* gG = p.gG + pt * N_MEMBERS * 3;
  float xbx = 0.f, xby = 0.f, xbz = 0.f;
#pragma unroll 1
  for (int k = 0; k < N_MEMBERS; ++k) {
    float t = 0.f, dx = 0.f, dy = 0.f, dz = 0.f;
    const float dS = S[k] - pr;
    if (k < N_LOC) {
      const float ex = qx - anch[3 * k], ey = qy - anch[3 * k + 1], ez = qz - anch[3 * k + 2];
      const float r = sqrtf(ex * ex + ey * ey + ez * ez);
      if (r > 0.f) {
        const float c = -2.f * (r + BL_EPS) / (BL_SIGMA * r);
        const float kappa = 2.f * BL_EPS / (BL_SIGMA * r * r * r);
        const float qe = bx * ex + by * ey + bz * ez;
        t = c * qe;
        const float qG = bx * G[3 * k] + by * G[3 * k + 1] + bz * G[3 * k + 2];
        const float coef = w[k] * c * (pb * dS + (qG - qgt) + dS * (t - tau) - theta);
        const float s2 = dS * w[k];
        dx = coef * ex + s2 * (c * bx + kappa * qe * ex);
        dy = coef * ey + s2 * (c * by + kappa * qe * ey);
        dz = coef * ez + s2 * (c * bz + kappa * qe * ez);
      }
    }
    if (live) {
      gS[k] = w[k] * (pb + t - tau);
      gG[3 * k] = w[k] * bx; gG[3 * k + 1] = w[k] * by; gG[3 * k + 2] = w[k] * bz;
    }
    if (k < N_LOC) {
      xbx += dx; xby += dy; xbz += dz;
      // anchor gradient: minus the sum over the block's points (wave reduction, then LDS, then one global atomic)
      float sx = dx, sy = dy, sz = dz;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) { sx += __shfl_xor(sx, o); sy += __shfl_xor(sy, o); sz += __shfl_xor(sz, o); }
      if ((threadIdx.x & 63) == 0) { float* o = ga[threadIdx.x >> 6] + 3 * k; o[0] = -sx; o[1] = -sy; o[2] = -sz; }
    }
  }
  if (live) { p.gxyz[pt * 3] += xbx; p.gxyz[pt * 3 + 1] += xby; p.gxyz[pt * 3 + 2] += xbz; }     // (this thread owns the point)
  __syncthreads();
  // the block's anchor gradient -> its slot of the partial table (blend_anchor_kernel adds the blocks of a row in order)
  for (int i = threadIdx.x; i < N_LOC * 3; i += blockDim.x)
    p.ga_part[(size_t(row) * gridDim.x + blockIdx.x) * (N_LOC * 3) + i] = (ga[0][i] + ga[1][i]) + (ga[2][i] + ga[3][i]);
}
__global__ __launch_bounds__(128) void blend_anchor_kernel(const float* ga_part, int n_blocks, float* ganch) {
  const int row = blockIdx.x, i = threadIdx.x;
  if (i >= N_LOC * 3) return;
  float acc = 0.f;
  for (int b = 0; b < n_blocks; ++b) acc += ga_part[(size_t(row) * n_blocks + b) * (N_LOC * 3) + i];
  ganch[size_t(row) * N_LOC * 3 + i] += acc;
}

// Scales of the binary16 operand storage (train_kernel, OM == 2) from the seeds of the step: max |dL/df_k| and max
// |dL/d(grad f_k)| over the batch (one pass, atomicMax on the bit patterns of non-negative floats; the last block to finish
// turns them into powers of two and clears the work words for the next call).  Measured on seeded and trained-like weights
// (tools/train_operand_stats.py): adjoint value columns <= 5e-3 x the value seed, input tangent columns <= 150 x the tangent
// seed, adjoint tangent columns 5e-4 .. 1e-1 whatever the seeds; the targets below put those maxima at ~2^11 of binary16's
// 2^16 (a 30 x margin before saturation) and leave 2^25 below them before the format's normal range ends.
__global__ __launch_bounds__(256) void operand_scales_kernel(const float* gs, int64_t ns, const float* gg, int64_t ng,
                                                              unsigned* work, float* scales) {
  __shared__ float red[2][4];
  float ms = 0.f, mg = 0.f;
  const int64_t stride = int64_t(gridDim.x) * blockDim.x, i0 = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  for (int64_t i = i0; i < ns; i += stride) ms = fmaxf(ms, fabsf(gs[i]));
  if (gg) for (int64_t i = i0; i < ng; i += stride) mg = fmaxf(mg, fabsf(gg[i]));
#pragma unroll
  for (int sft = 32; sft > 0; sft >>= 1) { ms = fmaxf(ms, __shfl_xor(ms, sft)); mg = fmaxf(mg, __shfl_xor(mg, sft)); }
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = ms; red[1][threadIdx.x >> 6] = mg; }
  __syncthreads();
  if (threadIdx.x == 0) {
    ms = fmaxf(fmaxf(red[0][0], red[0][1]), fmaxf(red[0][2], red[0][3]));
    mg = fmaxf(fmaxf(red[1][0], red[1][1]), fmaxf(red[1][2], red[1][3]));
    atomicMax(work, __float_as_uint(ms));
    atomicMax(work + 1, __float_as_uint(mg));
    __threadfence();
    if (atomicAdd(work + 2, 1u) == gridDim.x - 1) {
      __threadfence();
      const float smax = fmaxf(__uint_as_float(atomicMax(work, 0u)), 1e-30f);
      const float vmax = __uint_as_float(atomicMax(work + 1, 0u));
      auto pow2_below = [](float x) { return exp2f(fminf(fmaxf(floorf(log2f(x)), -60.f), 60.f)); };
      const float sd = pow2_below(2048.f / (5e-3f * smax));
      float su = vmax > 0.f ? pow2_below(2048.f / (150.f * vmax)) : 1.f;
      const float st = fminf(fmaxf(sd / su, 4.f), 32768.f);      // adjoint tangent columns: 5e-4 .. 1e-1 -> inside the format
      su = sd / st;
      scales[0] = sd; scales[1] = su; scales[2] = st; scales[3] = 1.f / sd;
      work[0] = 0u; work[1] = 0u; work[2] = 0u;
    }
  }
}

// ---- the training tier's point list on the device ------------------------------------------------------------------------
// The list is ordered by (member, row, point) and as long as the batch keeps pairs: its tile tables are built on the host from the
// pairs' counts (nphm_identity_train_tables - they size five launches).  Two launches, one workgroup per pair = member * n_rows +
// row, replace torch.nonzero + bincount on the [B,N,40] mask (~25 launches and a host synchronisation of their own):
//   pair_counts_kernel: counts[pair] = listed points of the pair (blend weight > 0) - the ONE array the host waits for;
//   pair_list_kernel  : list[offset(pair) + rank] = point, ranks in point order (offset = sum of the counts in front, which every
//                       workgroup adds up itself) - runs while the host builds the tile tables from the counts.
__global__ __launch_bounds__(256) void pair_counts_kernel(const float* __restrict__ what, int n_rows, int64_t n_points, int* __restrict__ counts) {
  __shared__ int wsum[4];
  const int pair = blockIdx.x, member = pair / n_rows, row = pair % n_rows;
  const float* w = what + int64_t(row) * n_points * N_MEMBERS + member;
  int c = 0;
  for (int64_t n = threadIdx.x; n < n_points; n += blockDim.x) c += w[n * N_MEMBERS] > 0.f ? 1 : 0;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) counts[pair] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

__global__ __launch_bounds__(256) void pair_list_kernel(const float* __restrict__ what, int n_rows, int64_t n_points, const int* __restrict__ counts,
                                                         int* __restrict__ list) {
  __shared__ int wsum[4];
  __shared__ int base;
  const int pair = blockIdx.x, member = pair / n_rows, row = pair % n_rows;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int off = 0;
  for (int q = threadIdx.x; q < pair; q += blockDim.x) off += counts[q];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) off += __shfl_xor(off, o);
  if (lane == 0) wsum[wave] = off;
  __syncthreads();
  if (threadIdx.x == 0) base = wsum[0] + wsum[1] + wsum[2] + wsum[3];
  __syncthreads();
  const float* w = what + int64_t(row) * n_points * N_MEMBERS + member;
  for (int64_t n0 = 0; n0 < n_points; n0 += blockDim.x) {
    const int64_t n = n0 + threadIdx.x;
    const bool keep = n < n_points && w[n * N_MEMBERS] > 0.f;
    const unsigned long long b = __ballot(keep);
    if (lane == 0) wsum[wave] = __popcll(b);
    __syncthreads();
    int at = base;
    for (int q = 0; q < wave; ++q) at += wsum[q];
    if (keep) list[at + __popcll(b & ((1ull << lane) - 1ull))] = int(n);
    __syncthreads();
    if (threadIdx.x == 0) base += wsum[0] + wsum[1] + wsum[2] + wsum[3];
    __syncthreads();
  }
}

}  // namespace train
}  // namespace nphm

// ============================================================================================
// C ABI (include/nphm_amd.h)
// ============================================================================================
extern "C" {

size_t nphm_identity_train_saved_bytes(int n_tiles, int operands) {
  return n_tiles <= 0 ? 0 : size_t(n_tiles) * nphm::train::SV_ROWS * 64 * (operands ? 2 : 4);
}
size_t nphm_identity_train_edge_bytes(int n_tiles) { return n_tiles <= 0 ? 0 : size_t(n_tiles) * nphm::train::EDGE_FLOATS * 4; }

static int train_common(nphm::train::TrainArgs& a, const void* packed, const void* packed_bwd, const void* latent_state,
                        const float* xyz, int64_t n_points, const int* tiles, int n_tiles, const int* point_list,
                        const char* who) {
  if (!packed || !packed_bwd || !latent_state || !xyz || !tiles || !point_list) return nphm_fail_msg(who);
  if (n_points <= 0 || n_tiles < 0) return nphm_fail_msg(who);
  memset(&a, 0, sizeof(a));
  a.packed_f32 = static_cast<const float*>(packed);
  a.packed_bf16 = reinterpret_cast<const uint16_t*>(static_cast<const char*>(packed) + nphm::PACKED_F32_FLOATS * 4);
  a.packed_bwd = static_cast<const uint16_t*>(packed_bwd);
  a.state = static_cast<const float*>(latent_state);
  a.xyz = xyz; a.n_points = n_points; a.tiles = tiles; a.n_tiles = n_tiles; a.list = point_list;
  return 0;
}

int nphm_identity_train_forward(const void* packed, const void* packed_bwd, const void* latent_state, const float* xyz,
                                int64_t n_points, const int* tiles, int n_tiles, const int* point_list,
                                float* member_sdf, float* member_grad, void* stream) {
  nphm::train::TrainArgs a;
  if (train_common(a, packed, packed_bwd, latent_state, xyz, n_points, tiles, n_tiles, point_list,
                   "nphm_identity_train_forward: null pointer or bad sizes")) return 1;
  if (!member_sdf || !member_grad) return nphm_fail_msg("nphm_identity_train_forward: null output");
  if (n_tiles == 0) return 0;
  a.member_sdf = member_sdf; a.member_grad = member_grad;
  hipLaunchKernelGGL(nphm::train::train_kernel<false>, dim3(n_tiles), dim3(64 * nphm::bwd::WAVES), 0,
                     static_cast<hipStream_t>(stream), a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return nphm_fail("nphm_identity_train_forward launch", e);
  return 0;
}

int nphm_identity_train_backward(const void* packed, const void* packed_bwd, const void* latent_state, const float* xyz,
                                 int64_t n_points, const int* tiles, int n_tiles, const int* point_list,
                                 const float* grad_member_sdf, const float* grad_member_grad,
                                 float* grad_xyz, void* saved, void* edge, int operands, const float* operand_scales, void* stream) {
  nphm::train::TrainArgs a;
  if (train_common(a, packed, packed_bwd, latent_state, xyz, n_points, tiles, n_tiles, point_list,
                   "nphm_identity_train_backward: null pointer or bad sizes")) return 1;
  if (!grad_member_sdf || !grad_xyz || !saved || !edge)
    return nphm_fail_msg("nphm_identity_train_backward: null pointer");
  if (n_tiles == 0) return 0;
  a.save = static_cast<float*>(saved);
  a.edge = static_cast<float*>(edge);
  a.g_sdf = grad_member_sdf; a.g_grad = grad_member_grad;
  a.gxyz = grad_xyz;
  a.op_scale = operand_scales;
  if (operands < 0 || operands > 2 || (operands == 2 && !operand_scales))
    return nphm_fail_msg("nphm_identity_train_backward: operands = 0 (fp32), 1 (bf16) or 2 (binary16 + operand_scales)");
  if (operands == 2)
    hipLaunchKernelGGL((nphm::train::train_kernel<true, 2>), dim3(n_tiles), dim3(64 * nphm::bwd::WAVES), 0,
                       static_cast<hipStream_t>(stream), a);
  else if (operands == 1)
    hipLaunchKernelGGL((nphm::train::train_kernel<true, 1>), dim3(n_tiles), dim3(64 * nphm::bwd::WAVES), 0,
                       static_cast<hipStream_t>(stream), a);
  else
    hipLaunchKernelGGL((nphm::train::train_kernel<true, 0>), dim3(n_tiles), dim3(64 * nphm::bwd::WAVES), 0,
                       static_cast<hipStream_t>(stream), a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return nphm_fail("nphm_identity_train_backward launch", e);
  return 0;
}

int nphm_identity_train_pair_counts(const float* blend_weights, int n_rows, int64_t n_points, int* counts, void* stream) {
  if (!blend_weights || !counts || n_rows <= 0 || n_points <= 0) return nphm_fail_msg("nphm_identity_train_pair_counts: bad arguments");
  hipLaunchKernelGGL(nphm::train::pair_counts_kernel, dim3(nphm::N_MEMBERS * n_rows), dim3(256), 0, static_cast<hipStream_t>(stream),
                     blend_weights, n_rows, n_points, counts);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : nphm_fail("nphm_identity_train_pair_counts launch", e);
}

int nphm_identity_train_point_list(const float* blend_weights, int n_rows, int64_t n_points, const int* counts, int* point_list, void* stream) {
  if (!blend_weights || !counts || !point_list || n_rows <= 0 || n_points <= 0) return nphm_fail_msg("nphm_identity_train_point_list: bad arguments");
  hipLaunchKernelGGL(nphm::train::pair_list_kernel, dim3(nphm::N_MEMBERS * n_rows), dim3(256), 0, static_cast<hipStream_t>(stream),
                     blend_weights, n_rows, n_points, counts, point_list);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : nphm_fail("nphm_identity_train_point_list launch", e);
}

// Host side of the training tier's work lists (no device work): from the number of listed points of every (member, row) pair,
// pair = member * n_rows + row - the point list is ordered that way -, the forward tile table (<= 64 consecutive list entries of a
// pair per tile), the backward tile table (<= 32), the weight-gradient work list (chunks of <= chunk_tiles consecutive tiles of ONE
// weight set inside ONE piece of ring_tiles tiles: (set, first tile relative to its piece, tiles, piece)), the first chunk of every
// weight set and the first backward tile of every pair.  (The same tables in numpy cost 0.5 ms of a 10 ms step with the GPU idle.)
int nphm_identity_train_tables(const long long* counts, int n_rows, const int* member_set, int n_sets, int ring_tiles, int chunk_tiles,
                               int* tiles_fwd, int* tiles_bwd, int* chunks, int* set_chunk_first, int* pair_first, int* sizes) {
  if (!counts || !member_set || !tiles_fwd || !tiles_bwd || !chunks || !set_chunk_first || !pair_first || !sizes || n_rows <= 0 ||
      n_sets <= 0 || chunk_tiles <= 0)
    return nphm_fail_msg("nphm_identity_train_tables: bad arguments");
  const int n_pairs = nphm::N_MEMBERS * n_rows;
  long long off = 0;
  int t64 = 0, t32 = 0;
  for (int pair = 0; pair < n_pairs; ++pair) {
    const int member = pair / n_rows, row = pair % n_rows;
    const long long c = counts[pair];
    for (long long w = 0; w < c; w += 64) {
      int* t = tiles_fwd + 4 * size_t(t64++);
      t[0] = row; t[1] = member; t[2] = int(off + w); t[3] = int(c - w < 64 ? c - w : 64);
    }
    pair_first[pair] = t32;
    for (long long w = 0; w < c; w += 32) {
      int* t = tiles_bwd + 4 * size_t(t32++);
      t[0] = row; t[1] = member; t[2] = int(off + w); t[3] = int(c - w < 32 ? c - w : 32);
    }
    off += c;
  }
  pair_first[n_pairs] = t32;
  const int ring = ring_tiles > 0 ? ring_tiles : (t32 > 0 ? t32 : 1);
  int nc = 0;
  for (int t = 0; t < t32;) {
    const int set = member_set[tiles_bwd[4 * size_t(t) + 1]], piece = t / ring;
    int end = t + 1;
    while (end < t32 && end / ring == piece && member_set[tiles_bwd[4 * size_t(end) + 1]] == set) ++end;
    for (int first = t; first < end; first += chunk_tiles) {
      int* c = chunks + 4 * size_t(nc++);
      c[0] = set; c[1] = first - piece * ring; c[2] = end - first < chunk_tiles ? end - first : chunk_tiles; c[3] = piece;
    }
    t = end;
  }
  for (int s = 0, c = 0; s <= n_sets; ++s) {            // chunks are ordered by tile, hence by weight set
    while (c < nc && chunks[4 * size_t(c)] < s) ++c;
    set_chunk_first[s] = c;
  }
  sizes[0] = t64; sizes[1] = t32; sizes[2] = nc; sizes[3] = ring;
  return 0;
}

size_t nphm_identity_train_wpart_bytes(int n_chunks) { return n_chunks <= 0 ? 0 : size_t(n_chunks) * nphm::train::WPART_FLOATS * 4; }

int nphm_identity_train_operand_scales(const float* grad_member_sdf, int64_t n_sdf, const float* grad_member_grad, int64_t n_grad,
                                       void* work, float* operand_scales, void* stream) {
  if (!grad_member_sdf || !work || !operand_scales || n_sdf <= 0) return nphm_fail_msg("nphm_identity_train_operand_scales: null pointer");
  hipLaunchKernelGGL(nphm::train::operand_scales_kernel, dim3(512), dim3(256), 0, static_cast<hipStream_t>(stream),
                     grad_member_sdf, n_sdf, grad_member_grad, grad_member_grad ? n_grad : 0, static_cast<unsigned*>(work), operand_scales);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return nphm_fail("nphm_identity_train_operand_scales launch", e);
  return 0;
}

int nphm_identity_train_weight_grads(const void* saved, int operands, const float* operand_scales, const int* chunks, int n_chunks, void* wpart,
                                     void* stream) {
  if (!saved || !chunks || !wpart) return nphm_fail_msg("nphm_identity_train_weight_grads: null pointer");
  if (n_chunks < 0) return nphm_fail_msg("nphm_identity_train_weight_grads: bad sizes");
  if (n_chunks == 0) return 0;
  nphm::train::WgradArgs a;
  a.saved = static_cast<const float*>(saved); a.chunks = chunks; a.part = static_cast<float*>(wpart);
  a.op_scale = operand_scales;
  if (operands < 0 || operands > 2 || (operands == 2 && !operand_scales))
    return nphm_fail_msg("nphm_identity_train_weight_grads: operands = 0 (fp32), 1 (bf16) or 2 (binary16 + operand_scales)");
  if (operands == 2)
    hipLaunchKernelGGL(nphm::train::wgrad_kernel<2>, dim3(n_chunks, 3), dim3(64 * nphm::bwd::WAVES), 0,
                       static_cast<hipStream_t>(stream), a);
  else if (operands == 1)
    hipLaunchKernelGGL(nphm::train::wgrad_kernel<1>, dim3(n_chunks, 3), dim3(64 * nphm::bwd::WAVES), 0,
                       static_cast<hipStream_t>(stream), a);
  else
    hipLaunchKernelGGL(nphm::train::wgrad_kernel<0>, dim3(n_chunks, 3), dim3(64 * nphm::bwd::WAVES), 0,
                       static_cast<hipStream_t>(stream), a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return nphm_fail("nphm_identity_train_weight_grads launch", e);
  return 0;
}

int nphm_identity_train_reduce_grads(const void* edge, int n_tiles, const void* wpart, const int* chunks, int n_chunks,
                                     int ring_tiles, const int* set_chunk_first, const int* pair_first, int n_rows, void* scratch,
                                     float* const grad_weight[5], float* grad_bias1, float* grad_bias3, float* grad_bias4,
                                     float* grad_b0, float* grad_b2, float* grad_anchors, void* stream) {
  if (!edge || !wpart || !chunks || !set_chunk_first || !pair_first || !scratch || !grad_weight || !grad_bias1 || !grad_bias3 ||
      !grad_bias4 || !grad_b0 || !grad_b2 || !grad_anchors)
    return nphm_fail_msg("nphm_identity_train_reduce_grads: null pointer");
  for (int i = 0; i < 5; ++i)
    if (!grad_weight[i]) return nphm_fail_msg("nphm_identity_train_reduce_grads: null gradient pointer");
  if (n_tiles < 0 || n_chunks < 0 || ring_tiles <= 0 || n_rows <= 0) return nphm_fail_msg("nphm_identity_train_reduce_grads: bad sizes");
  if (n_tiles == 0 || n_chunks == 0) return 0;
  nphm::train::EdgeArgs a;
  a.edge = static_cast<const float*>(edge); a.chunks = chunks; a.ring_tiles = ring_tiles; a.part = static_cast<float*>(scratch);
  a.set_chunk_first = set_chunk_first; a.pair_first = pair_first; a.n_rows = n_rows;
  a.gW0 = grad_weight[0]; a.gW4 = grad_weight[4]; a.gb4 = grad_bias4; a.gb1 = grad_bias1; a.gb3 = grad_bias3;
  a.gb0 = grad_b0; a.gb2 = grad_b2; a.ganch = grad_anchors;
  nphm::train::WsetArgs w;
  w.part = static_cast<const float*>(wpart); w.set_chunk_first = set_chunk_first;
  w.gW1 = grad_weight[1]; w.gW2 = grad_weight[2]; w.gW3 = grad_weight[3];
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(nphm::train::edge_chunk_kernel, dim3(n_chunks), dim3(1024), 0, st, a);
  hipLaunchKernelGGL(nphm::train::edge_set_kernel, dim3(nphm::N_SETS), dim3(1024), 0, st, a);
  hipLaunchKernelGGL(nphm::train::edge_pair_kernel, dim3(nphm::N_MEMBERS * n_rows), dim3(256), 0, st, a);
  hipLaunchKernelGGL(nphm::train::wgrad_set_kernel, dim3((nphm::train::WPART_FLOATS + 255) / 256, nphm::N_SETS), dim3(256), 0, st, w);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return nphm_fail("nphm_identity_train_reduce_grads launch", e);
  return 0;
}

int nphm_identity_blend_forward(const float* xyz, const float* anchors, const float* member_sdf, const float* member_grad,
                                int n_rows, int64_t n_points, float* pred, float* grad, void* stream) {
  if (!xyz || !anchors || !member_sdf || !member_grad || !pred || !grad)
    return nphm_fail_msg("nphm_identity_blend_forward: null pointer");
  if (n_rows <= 0 || n_points <= 0) return nphm_fail_msg("nphm_identity_blend_forward: bad sizes");
  nphm::train::BlendArgs a;
  memset(&a, 0, sizeof(a));
  a.xyz = xyz; a.anchors = anchors; a.S = member_sdf; a.G = member_grad; a.n_points = n_points;
  a.pred = pred; a.grad = grad;
  hipLaunchKernelGGL(nphm::train::blend_kernel<false>, dim3(unsigned((n_points + 255) / 256), n_rows), dim3(256), 0,
                     static_cast<hipStream_t>(stream), a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return nphm_fail("nphm_identity_blend_forward launch", e);
  return 0;
}

size_t nphm_identity_blend_partial_bytes(int n_rows, int64_t n_points) {
  return n_rows <= 0 || n_points <= 0 ? 0 : size_t(n_rows) * size_t((n_points + 255) / 256) * nphm::N_LOC * 3 * 4;
}

int nphm_identity_blend_backward(const float* xyz, const float* anchors, const float* member_sdf, const float* member_grad,
                                 const float* grad_pred, const float* grad_grad, int n_rows, int64_t n_points,
                                 float* grad_member_sdf, float* grad_member_grad, float* grad_xyz, float* grad_anchors,
                                 void* anchor_partials, void* stream) {
  if (!xyz || !anchors || !member_sdf || !member_grad || !grad_pred || !grad_member_sdf || !grad_member_grad || !grad_xyz ||
      !grad_anchors || !anchor_partials)
    return nphm_fail_msg("nphm_identity_blend_backward: null pointer");
  if (n_rows <= 0 || n_points <= 0) return nphm_fail_msg("nphm_identity_blend_backward: bad sizes");
  nphm::train::BlendArgs a;
  memset(&a, 0, sizeof(a));
  a.xyz = xyz; a.anchors = anchors; a.S = member_sdf; a.G = member_grad; a.n_points = n_points;
  a.g_pred = grad_pred; a.g_grad = grad_grad;
  a.gS = grad_member_sdf; a.gG = grad_member_grad; a.gxyz = grad_xyz; a.ganch = grad_anchors;
  a.ga_part = static_cast<float*>(anchor_partials);
  const unsigned blocks = unsigned((n_points + 255) / 256);
  hipLaunchKernelGGL(nphm::train::blend_kernel<true>, dim3(blocks, n_rows), dim3(256), 0, static_cast<hipStream_t>(stream), a);
  hipLaunchKernelGGL(nphm::train::blend_anchor_kernel, dim3(n_rows), dim3(128), 0, static_cast<hipStream_t>(stream),
                     a.ga_part, int(blocks), grad_anchors);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return nphm_fail("nphm_identity_blend_backward launch", e);
  return 0;
}

}  // extern "C"

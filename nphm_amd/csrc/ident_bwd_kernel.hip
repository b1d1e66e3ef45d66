// ident_bwd_kernel.hip — first-order backward of the NPHM identity field
// (FastEnsembleDeepSDFMirrored.forward, src/NPHM/models/EnsembledDeepSDF.py:203-267) for the latent
// fitting loop (src/NPHM/models/fitting.py:111-167: loss.backward() through decoder(xc, z_id)), i.e.
// what torch.autograd computes there through 5 bmm's, 4 softplus and the blend per member.
//
// Gradients returned: d L / d xyz [B,N,3], d L / d anchors [B,39,3] and d L / d (folded biases of
// lin0 and of the skip layer) [B,40,200] each; the host module chains the last three through
// mlp_pos / the latent columns of lin0 and lin2 with ordinary autograd (tiny).  Weight gradients are
// NOT produced (the fitting loop never uses them): the host picks this tier only when no parameter
// requires grad.
//
// Decomposition: member-centric.  The host lists, per (batch row, member), the points whose
// normalised blend weight exceeds prune_tol; one workgroup = 8 wavefronts = one member x 64 listed
// points.  It recomputes the member's forward (activations in LDS as split-bf16 K chunks, every
// wavefront owns one 32-row output tile of every layer, weights stream L2 -> VGPR from the forward
// pack), keeps sigma'(d_l) of its own tiles in registers, then runs the transposed chain
//   G3 = g w_k/denom * w4 * s3,  G2 = (W3^T G3) s2,  G1 = (W2a^T G2 / sqrt2) s1,  G0 = (W1^T G1) s0,
//   d c = W0c^T G0 + W2c^T G2 / sqrt2
// on the same MFMA path (split-bf16 x3, transposed pack), reduces the bias gradients over its points
// and adds the blend-weight terms (f_k - sdf) d w_k / d q.  All in the scaled domain of layout.h.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "capi_common.h"
#include "layout.h"

#include "member_common.h"

#ifndef NPHM_BWD_RING
#define NPHM_BWD_RING 4        // K-steps of weight fragments a wavefront of the backward kernel keeps in flight (LDS ring)
#endif

namespace nphm {
namespace bwd {

struct PackArgs {
  const float* w[5];
  uint16_t* out;
};

// k-slot 8*h + i of K-step ks <-> feature feat_of(ks >> 1, 8*(ks & 1) + i, h) of the producing tile
__global__ void pack_bwd_kernel(PackArgs a) {
  const int s = blockIdx.y;
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= BWD_SET_STRIDE) return;
  int stage, x, nks;
  if (e < OFF_B) { stage = 0; x = e - OFF_A; nks = A_KS; }
  else if (e < OFF_C) { stage = 1; x = e - OFF_B; nks = B_KS; }
  else if (e < OFF_D) { stage = 2; x = e - OFF_C; nks = C_KS; }
  else { stage = 3; x = e - OFF_D; nks = D_KS; }
  const int i = x & 7, lane = (x >> 3) & 63, part = (x >> 9) & 1, gg = x >> 10;
  const int ks = gg % nks, ob = gg / nks;
  const int row = 32 * ob + (lane & 31);                         // output row of the transposed stage
  const int kf = feat_of(ks >> 1, 8 * (ks & 1) + i, lane >> 5);  // its K feature = output feature of the forward layer
  float w = 0.f;
  if (stage == 0) {                       // lin3^T: rows = lin3 inputs, K = lin3 outputs
    if (row < HID && kf < HID) w = a.w[3][(size_t(s) * HID + kf) * HID + row];
  } else if (stage == 1) {                // lin2^T: rows 0..100 h1 (1/sqrt2), rows 101..103 coords (k/sqrt2)
    if (row < L2_IN && kf < HID) {
      w = a.w[2][(size_t(s) * HID + kf) * HID + row] / INV_SQRT2_DIV;
      if (row >= L1_OUT) w *= SP_SCALE;
    }
  } else if (stage == 2) {                // lin1^T: rows = lin1 inputs (200), K = lin1 outputs (101)
    if (row < HID && kf < L1_OUT) w = a.w[1][(size_t(s) * L1_OUT + kf) * HID + row];
  } else {                                // lin0[:, :3]^T: rows = coordinates, K = lin0 outputs
    if (row < 3 && kf < HID) w = a.w[0][(size_t(s) * HID + kf) * D_IN + row] * SP_SCALE;
  }
  const uint16_t hi = f32_to_bf16_rn(w);
  const uint16_t lo = f32_to_bf16_rn(w - bf16_to_f32(hi));
  a.out[size_t(s) * BWD_SET_STRIDE + e] = part ? lo : hi;
}

struct BwdArgs {
  const uint16_t* packed_bf16;    // forward pack (layout.h, bf16 half of the packed buffer)
  const float* packed_f32;        // for lin4's bias
  const uint16_t* packed_bwd;
  const float* state;             // [n_rows, LS_ROW_STRIDE]
  const float* xyz;               // [n_rows, n_points, 3]
  const float* sdf;               // [n_rows, n_points] blended forward value
  const float* gout;              // [n_rows, n_points] d L / d sdf
  const int* tiles;               // [n_tiles][4] = row, member, offset into list, count (<= 64)
  const int* list;                // point indices (within their row)
  int n_tiles;                    // tiles in the table ...
  const int* n_tiles_dev;         // ... or, if not NULL, their number in device memory (tables built on the device)
  int64_t n_points;
  // backward kernel: per-tile / per-pair records (no atomics; ident_bwd_reduce_kernel adds them in table / member order)
  float* gbp;                     // [n_tiles][2][200]  bias gradients of lin0 | the skip layer from the tile's points
  float* gap;                     // [n_tiles][4]       the tile's share of d L / d anchor_k
  float* gxm;                     // [n_rows, n_points, 40, 3]  member k's share of d L / d xyz of a listed point
  float* fmem;                    // forward-only kernel: [n_rows, n_points, 40] member predictions
};

// BWD = false: the forward half only — per-member predictions f_k of the listed points into
// fmem[row, point, member] (the host blends them); used by the autograd tier's forward, where the
// query points are scattered surface samples for which the brick-coherent kernel of eval_kernel.hip
// prunes poorly (a wavefront of 32 unrelated points touches most members).
template <bool BWD>
__device__ __forceinline__ void member_tile(const BwdArgs& p, const int tile_index) {
  __shared__ __attribute__((aligned(16))) char act_hi[PLANE_BYTES];
  __shared__ __attribute__((aligned(16))) char act_lo[PLANE_BYTES];
  __shared__ float part[WAVES][M];          // per-wave partial of lin4 / of the coordinate gradient
  __shared__ float pt_q[M][4];              // query point, validity
  __shared__ float pt_c[M][4];              // local coordinates of the member
  __shared__ float pt_g[M][4];              // df (= g w_k / denom), blend coefficient, denominators
  __shared__ float pt_dc[M][4];             // d L / d local coords from the skip layer
  __shared__ int pt_idx[M];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int h = lane >> 5, j = lane & 31;
  const int* tile = p.tiles + 4 * tile_index;
  const int row = tile[0], k = tile[1], off = tile[2], cnt = tile[3];
  if (cnt <= 0) return;                   // empty slot of a fixed-capacity tile table
  const int set = member_set(k);
  const float* st = p.state + size_t(row) * LS_ROW_STRIDE;
  const float* anch = st + LS_OFF_ANCH;
  const float sign_x = (k < 2 * N_SYMM && (k & 1)) ? -1.f : 1.f;
  const float w_bg = expf(-0.2f / 0.01f);

  const uint16_t* fw = p.packed_bf16 + size_t(set) * BF_SET_STRIDE;
  const uint16_t* bw = p.packed_bwd + size_t(set) * BWD_SET_STRIDE;
  // The backward variant streams its weights through the per-wavefront LDS ring (member_common.h): RING K-steps in
  // flight, prologue of a stage issued as soon as the previous stage's K loop is done (weights do not depend on
  // activations), fully unrolled K loop so that every vmcnt count is an immediate.  prefetch + run must be paired.
  constexpr int RING = BWD ? NPHM_BWD_RING : 0;
  __shared__ __attribute__((aligned(16))) char wring[RING ? WAVES * RING * 2048 : 16];
  const unsigned ring_lds = lds_addr_of(wring) + unsigned(wave) * (RING * 2048);
  const bf16x8* const R = reinterpret_cast<const bf16x8*>(wring + wave * (RING * 2048)) + lane;
  const unsigned voff = unsigned(lane) * 16u;
  auto ring_prefetch = [&](const uint16_t* frag_base, int n, auto ks_c) __attribute__((always_inline)) {
    constexpr int KS = decltype(ks_c)::value;
    static_assert(RING == 0 || KS >= RING, "a stage fills the ring");
    const v4i rs = raw_rsrc(frag_base);
    const unsigned s0 = unsigned(n * KS) * 2048u;
#pragma unroll
    for (int u = 0; u < RING; ++u) dma_kstep(rs, voff, s0 + unsigned(u) * 2048u, ring_lds + unsigned(u) * 2048u);
  };
  if constexpr (RING) {                  // the first stage this wavefront runs (4 .. 6 sit out lin1)
    if (wave < L1_OB) ring_prefetch(fw + BF_OFF_L1A, wave, std::integral_constant<int, L1_KS16>{});
    else if (wave < 7) ring_prefetch(fw + BF_OFF_L2A, wave, std::integral_constant<int, L2_KS16>{});
  }
  // ---- per point: coordinates, blend weights --------------------------------------------------------
  // backward: the denominator sum_a w_a of the point in lane m, the 39 anchors dealt over the 8 wavefronts (5 steps instead
  // of 39 on one wavefront while seven wait at the barrier); the forward-only kernel needs no blend weight at all
  if (BWD) {
    const int m = lane;
    const int n = p.list[off + (m < cnt ? m : cnt - 1)];
    const float* q = p.xyz + (int64_t(row) * p.n_points + n) * 3;
    const float qx = q[0], qy = q[1], qz = q[2];
    float s = 0.f;
#pragma unroll 1
    for (int a = wave; a < N_LOC; a += WAVES) {
      const float dx = anch[3 * a] - qx, dy = anch[3 * a + 1] - qy, dz = anch[3 * a + 2] - qz;
      const float d = sqrtf(dx * dx + dy * dy + dz * dz) + 1e-5f;
      s += expf(-(d * d) / 0.01f);
    }
    part[wave][m] = s;
    __syncthreads();
  }
  if (threadIdx.x < M) {
    const int m = threadIdx.x;
    const bool ok = m < cnt;
    const int n = p.list[off + (ok ? m : cnt - 1)];
    const float* q = p.xyz + (int64_t(row) * p.n_points + n) * 3;
    const float qx = q[0], qy = q[1], qz = q[2];
    float wk = w_bg, dk = 0.f, nk = 1.f, ax = 0.f, ay = 0.f, az = 0.f;
    if (k < N_LOC) {
      ax = anch[3 * k]; ay = anch[3 * k + 1]; az = anch[3 * k + 2];
      if (BWD) {
        const float dx = ax - qx, dy = ay - qy, dz = az - qz;
        nk = sqrtf(dx * dx + dy * dy + dz * dz);
        dk = nk + 1e-5f;
        wk = expf(-(dk * dk) / 0.01f);
      }
    }
    pt_idx[m] = ok ? n : -1;
    pt_q[m][0] = qx; pt_q[m][1] = qy; pt_q[m][2] = qz; pt_q[m][3] = nk;
    pt_c[m][0] = sign_x * (qx - ax); pt_c[m][1] = qy - ay; pt_c[m][2] = qz - az; pt_c[m][3] = 0.f;
    if (BWD) {
      float S = w_bg;
#pragma unroll
      for (int w = 0; w < WAVES; ++w) S += part[w][m];
      const float denom = S + 1e-6f;
      const float g = ok ? p.gout[int64_t(row) * p.n_points + n] : 0.f;
      pt_g[m][0] = g * wk / denom;                                       // d L / d f_k
      // d L / d w_k * d w_k / d d  (without the (f_k - sdf) factor, which needs the forward value)
      pt_g[m][1] = (k < N_LOC) ? g / denom * wk * (-2.f * dk / 0.01f) : 0.f;
      pt_g[m][2] = ok ? p.sdf[int64_t(row) * p.n_points + n] : 0.f;
      pt_g[m][3] = 0.f;
    }
  }
  __syncthreads();

  // ---- this lane's two points (m-tiles) ---------------------------------------------------------
  float cx[MT], cy[MT], cz[MT];
  bf16x8 bv[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    const int m = 32 * t + j;
    cx[t] = pt_c[m][0]; cy[t] = pt_c[m][1]; cz[t] = pt_c[m][2];
    bv[t] = coord_operand(cx[t], cy[t], cz[t], h);
  }

  const float* tails = st + LS_OFF_TAIL + size_t(k) * GEMM_CHUNKS * TAIL_FLOATS;
  const bf16x8* Bh = reinterpret_cast<const bf16x8*>(act_hi) + h * M + j;
  const bf16x8* Bl = reinterpret_cast<const bf16x8*>(act_lo) + h * M + j;
  const f32x16 zero16 = {};

  // out tile `n` of a stage: acc[t] += sum over ks K-steps of A(n, s) x act(s)   (split-bf16 x3)
  auto gemm_tile = [&](f32x16 (&acc)[MT], const uint16_t* frag_base, int n, int ks) __attribute__((always_inline)) {
    const bf16x8* W = reinterpret_cast<const bf16x8*>(frag_base) + size_t(n) * ks * 128 + lane;
    bf16x8 ah[2], al[2], bh[2][MT], bl[2][MT];
    auto load = [&](int slot, int s) __attribute__((always_inline)) {
      ah[slot] = W[s * 128];
      al[slot] = W[s * 128 + 64];
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        bh[slot][t] = Bh[2 * s * M + 32 * t];
        bl[slot][t] = Bl[2 * s * M + 32 * t];
      }
    };
    auto mma = [&](int slot) __attribute__((always_inline)) {
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[slot], bh[slot][t], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[slot], bl[slot][t], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[slot], bh[slot][t], acc[t], 0, 0, 0);
      }
    };
    load(0, 0);
#pragma unroll 1
    for (int s = 0; s < ks; s += 2) {
      if (s + 1 < ks) load(1, s + 1);
      mma(0);
      if (s + 2 < ks) load(0, s + 2);
      if (s + 1 < ks) mma(1);
    }
  };
  auto gemm_ring = [&](f32x16 (&acc)[MT], const uint16_t* frag_base, int n, auto ks_c) __attribute__((always_inline)) {
    constexpr int KS = decltype(ks_c)::value;
    constexpr int RG = RING ? RING : 1;
    const v4i rs = raw_rsrc(frag_base);
    const unsigned s0 = unsigned(n * KS) * 2048u;
    bf16x8 ah[2], al[2], bh[MT], bl[MT];               // A: two K-steps; B: one per point tile, reloaded behind its MFMAs
    wait_vm<2 * (RG - 1)>();                           // K-step 0 has landed
    ah[0] = R[0]; al[0] = R[64];
#pragma unroll
    for (int t = 0; t < MT; ++t) { bh[t] = Bh[32 * t]; bl[t] = Bl[32 * t]; }
    static_for<KS>([&](auto sc) __attribute__((always_inline)) {
      constexpr int S = decltype(sc)::value, cur = S & 1, nxt = cur ^ 1;
      if constexpr (S + 1 < KS) {
        // K-steps issued so far: 0 .. min(S + RING, KS) - 1; those after S + 1 may still be in flight
        constexpr int issued = S + RG < KS ? S + RG : KS;
        wait_vm<2 * (issued - 1 - (S + 1))>();
        ah[nxt] = R[((S + 1) % RG) * 128]; al[nxt] = R[((S + 1) % RG) * 128 + 64];
      }
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[cur], bh[t], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[cur], bl[t], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[cur], bh[t], acc[t], 0, 0, 0);
        if constexpr (S + 1 < KS) { bh[t] = Bh[2 * (S + 1) * M + 32 * t]; bl[t] = Bl[2 * (S + 1) * M + 32 * t]; }
      }
      // slot of K-step S: its fragments went to registers one step ago and have just been consumed
      if constexpr (S + RG < KS) dma_kstep(rs, voff, s0 + unsigned(S + RG) * 2048u, ring_lds + unsigned(S % RG) * 2048u);
    });
  };
  // D tile n -> LDS K chunks 4n + 2*half + h of the points
  auto store_tile = [&](int n, const f32x16 (&v)[MT]) __attribute__((always_inline)) {
#pragma unroll
    for (int t = 0; t < MT; ++t) {
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

  f32x16 acc[MT], s0[MT], s1[MT], s2[MT], s3[MT], val[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) { s0[t] = zero16; s1[t] = zero16; s2[t] = zero16; s3[t] = zero16; }

  // ================================ forward (recompute) ==========================================
  // L0: lin0 on the local coordinates, folded bias inside the block (per-latent state)
  if (wave < 7) {
    const bf16x8* A0 = reinterpret_cast<const bf16x8*>(st + LS_OFF_L0B + size_t(k) * L0_BLOCK_FLOATS) + lane;
    const bf16x8 a = A0[wave * 64];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bv[t], zero16, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 16; ++r) { s0[t][r] = sigmoid2(acc[t][r]); val[t][r] = softplus2(acc[t][r]); }
    }
    store_tile(wave, val);
  }
  __syncthreads();
  // L1: 200 -> 101 (4 tiles); the skip coordinates join tile 3 as features 101..103
  if (wave < L1_OB) {
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[t] = load_frag16(tails + wave * TAIL_FLOATS + h * 16);
    if constexpr (RING) { gemm_ring(acc, fw + BF_OFF_L1A, wave, std::integral_constant<int, L1_KS16>{}); ring_prefetch(fw + BF_OFF_L2A, wave, std::integral_constant<int, L2_KS16>{}); }
    else gemm_tile(acc, fw + BF_OFF_L1A, wave, L1_KS16);
#pragma unroll
    for (int t = 0; t < MT; ++t) {
#pragma unroll
      for (int r = 0; r < 16; ++r) { s1[t][r] = sigmoid2(acc[t][r]); val[t][r] = softplus2(acc[t][r]); }
      if (wave == L1_OB - 1) {
        // (the skip connection's coordinates come from LDS here: as registers that live from the prologue to this point they
        // were the 3 spilled VGPRs / 16 B of scratch of the backward variant in rounds 4-5)
        const f32x4 c4 = *reinterpret_cast<const f32x4*>(pt_c[32 * t + j]);
        val[t][1] = h ? c4[0] : val[t][1];
        val[t][2] = h ? c4[1] : val[t][2];
        val[t][3] = h ? c4[2] : val[t][3];
      }
    }
  }
  __syncthreads();                       // every wavefront has read a0
  if (wave < L1_OB) store_tile(wave, val);
  __syncthreads();
  // L2: 104 -> 200
  if (wave < 7) {
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[t] = load_frag16(tails + (L1_OB + wave) * TAIL_FLOATS + h * 16);
    if constexpr (RING) { gemm_ring(acc, fw + BF_OFF_L2A, wave, std::integral_constant<int, L2_KS16>{}); ring_prefetch(fw + BF_OFF_L3A, wave, std::integral_constant<int, L3_KS16>{}); }
    else gemm_tile(acc, fw + BF_OFF_L2A, wave, L2_KS16);
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) { s2[t][r] = sigmoid2(acc[t][r]); val[t][r] = softplus2(acc[t][r]); }
  }
  __syncthreads();
  if (wave < 7) store_tile(wave, val);
  __syncthreads();
  // L3: 200 -> 200, lin4 fused: f = sum a3 * w4 / k + b4
  f32x16 w4v = zero16;
  if (wave < 7) {
    const float* tl = tails + (L1_OB + L2_OB + wave) * TAIL_FLOATS;
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[t] = load_frag16(tl + h * 16);
    if constexpr (RING) { gemm_ring(acc, fw + BF_OFF_L3A, wave, std::integral_constant<int, L3_KS16>{}); ring_prefetch(bw + OFF_A, wave, std::integral_constant<int, A_KS>{}); }
    else gemm_tile(acc, fw + BF_OFF_L3A, wave, L3_KS16);
    w4v = load_frag16(tl + 32 + h * 16);
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      float partial = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        s3[t][r] = sigmoid2(acc[t][r]);
        partial = fmaf(softplus2(acc[t][r]), w4v[r], partial);
      }
      partial += __shfl_xor(partial, 32);
      if (h == 0) part[wave][32 * t + j] = partial;
    }
  }
  __syncthreads();
  if (threadIdx.x < M) {
    const int m = threadIdx.x;
    float f = p.packed_f32[size_t(set) * SET_STRIDE + OFF_L4B];
#pragma unroll
    for (int w = 0; w < 7; ++w) f += part[w][m];
    if (!BWD) {
      if (pt_idx[m] >= 0) p.fmem[(int64_t(row) * p.n_points + pt_idx[m]) * N_MEMBERS + k] = f;
    }
    // blend-weight term: d L / d q += g (f_k - sdf) / denom * d w_k / d d * (q - a_k) / |q - a_k|
    if (BWD) pt_g[m][3] = pt_g[m][1] * (f - pt_g[m][2]);
  }
  if (!BWD) return;
  __syncthreads();

  // ================================ backward =====================================================
  // G3 = df * (w4 / k) * s3   (gradient w.r.t. the SCALED pre-activation of lin3)
  if (wave < 7) {
    // (lin4's weights are read again here rather than kept across the barrier above: 16 VGPRs that only live from L3 to this line)
    const float* tl4 = tails + (L1_OB + L2_OB + wave) * TAIL_FLOATS;
    asm volatile("" : "+s"(tl4));
    const f32x16 w4r = load_frag16(tl4 + 32 + h * 16);
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      const float df = pt_g[32 * t + j][0];
#pragma unroll
      for (int r = 0; r < 16; ++r) val[t][r] = df * w4r[r] * s3[t][r];
    }
    store_tile(wave, val);               // a2 is no longer needed (every wavefront passed the barriers above)
  }
  __syncthreads();
  // stage A: G2 = (lin3^T G3) * s2 ; bias gradient of the skip layer
  if (wave < 7) {
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[t] = zero16;
    gemm_ring(acc, bw + OFF_A, wave, std::integral_constant<int, A_KS>{});
    if (wave < B_OB) ring_prefetch(bw + OFF_B, wave, std::integral_constant<int, B_KS>{}); else ring_prefetch(bw + OFF_C, wave, std::integral_constant<int, C_KS>{});
    float* gb = p.gbp + (size_t(tile_index) * 2 + 1) * HID;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float sum = 0.f;
#pragma unroll
      for (int t = 0; t < MT; ++t) { val[t][r] = acc[t][r] * s2[t][r]; sum += val[t][r]; }
      sum = half_wave_sum(sum);
      const int f = feat_of(wave, r, h);
      if (j == 0 && f < HID) gb[f] = sum * SP_SCALE;
    }
  }
  __syncthreads();
  if (wave < 7) store_tile(wave, val);
  __syncthreads();
  // stage B: rows 0..100: G1 = (lin2a^T G2 / sqrt2) * s1 ; rows 101..103: d L / d coords (skip path)
  if (wave < B_OB) {
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[t] = zero16;
    gemm_ring(acc, bw + OFF_B, wave, std::integral_constant<int, B_KS>{});
    ring_prefetch(bw + OFF_C, wave, std::integral_constant<int, C_KS>{});
#pragma unroll
    for (int t = 0; t < MT; ++t) {
#pragma unroll
      for (int r = 0; r < 16; ++r) val[t][r] = acc[t][r] * s1[t][r];
      if (wave == B_OB - 1) {
        if (h) {                          // features 101..103 = registers 1..3 of the upper half-wave
          pt_dc[32 * t + j][0] = acc[t][1]; pt_dc[32 * t + j][1] = acc[t][2]; pt_dc[32 * t + j][2] = acc[t][3];
          val[t][1] = 0.f; val[t][2] = 0.f; val[t][3] = 0.f;
        }
      }
    }
  }
  __syncthreads();
  if (wave < B_OB) store_tile(wave, val);
  __syncthreads();
  // stage C: G0 = (lin1^T G1) * s0 ; bias gradient of lin0
  if (wave < 7) {
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[t] = zero16;
    gemm_ring(acc, bw + OFF_C, wave, std::integral_constant<int, C_KS>{});
    if (wave == 0) ring_prefetch(bw + OFF_D, 0, std::integral_constant<int, D_KS>{});
    float* gb = p.gbp + size_t(tile_index) * 2 * HID;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float sum = 0.f;
#pragma unroll
      for (int t = 0; t < MT; ++t) { val[t][r] = acc[t][r] * s0[t][r]; sum += val[t][r]; }
      sum = half_wave_sum(sum);
      const int f = feat_of(wave, r, h);
      if (j == 0 && f < HID) gb[f] = sum * SP_SCALE;
    }
  }
  __syncthreads();
  if (wave < 7) store_tile(wave, val);
  __syncthreads();
  // stage D: d L / d coords (lin0 path) = (k lin0[:, :3])^T G0 ; one tile, wavefront 0
  if (wave == 0) {
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[t] = zero16;
    gemm_ring(acc, bw + OFF_D, 0, std::integral_constant<int, D_KS>{});
    if (h == 0) {
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        float* o = pt_dc[32 * t + j];
        // rows 0..2 of the tile = registers 0..2 of the lower half-wave
        o[0] += acc[t][0]; o[1] += acc[t][1]; o[2] += acc[t][2];
      }
    }
  }
  __syncthreads();

  // ---- per point: chain to the query point and to the anchor --------------------------------------
  if (threadIdx.x < M) {
    const int m = threadIdx.x;
    const int n = pt_idx[m];
    float gq[3] = {0.f, 0.f, 0.f};
    if (n >= 0) {
      gq[0] = sign_x * pt_dc[m][0]; gq[1] = pt_dc[m][1]; gq[2] = pt_dc[m][2];      // through c = s (q - a)
      float ga[3] = {0.f, 0.f, 0.f};
      if (k < N_LOC) {
        ga[0] = -gq[0]; ga[1] = -gq[1]; ga[2] = -gq[2];
        const float nk = pt_q[m][3];
        if (nk > 0.f) {                                                            // through w_k(|q - a_k|)
          const float cb = pt_g[m][3] / nk;
          const float ex = sign_x * pt_c[m][0], ey = pt_c[m][1], ez = pt_c[m][2];   // q - a_k
          gq[0] += cb * ex; gq[1] += cb * ey; gq[2] += cb * ez;
          ga[0] -= cb * ex; ga[1] -= cb * ey; ga[2] -= cb * ez;
        }
      } else {
        gq[0] = pt_dc[m][0];      // the background member sees global coordinates (no anchor)
      }
      float* o = p.gxm + ((int64_t(row) * p.n_points + n) * N_MEMBERS + k) * 3;
      o[0] = gq[0]; o[1] = gq[1]; o[2] = gq[2];
      pt_q[m][0] = ga[0]; pt_q[m][1] = ga[1]; pt_q[m][2] = ga[2];
    } else {
      pt_q[m][0] = 0.f; pt_q[m][1] = 0.f; pt_q[m][2] = 0.f;
    }
    // anchor gradient: sum over the workgroup's points (one wavefront holds all 64)
    float sx = pt_q[m][0], sy = pt_q[m][1], sz = pt_q[m][2];
#pragma unroll
    for (int o2 = 32; o2 > 0; o2 >>= 1) { sx += __shfl_xor(sx, o2); sy += __shfl_xor(sy, o2); sz += __shfl_xor(sz, o2); }
    if (m == 0) { float* o = p.gap + size_t(tile_index) * 4; o[0] = sx; o[1] = sy; o[2] = sz; }
  }
}

// The records of the backward kernel -> d L / d (folded biases, anchors) per (row, member) and d L / d xyz per point, fixed
// order, every output element WRITTEN (no zero fill).  Blocks [0, n_rows * 40): pair (row, member) walks the tile table in
// order and adds its tiles' records (a pair's tiles are few; the table is ~12 KB); the rest: one thread per point adds the
// shares of the members that listed it (blend weight != 0), member ascending.
struct ReduceArgs {
  const int* tiles; int n_tiles; const int* n_tiles_dev;
  const float* gbp; const float* gap; const float* gxm; const float* what;
  int n_rows; int64_t n_points;
  float* gxyz; float* ganch; float* gb0; float* gb2;
};
__global__ __launch_bounds__(256) void ident_bwd_reduce_kernel(ReduceArgs p) {
  const int n_pairs = p.n_rows * N_MEMBERS;
  if (int(blockIdx.x) < n_pairs) {
    const int row = blockIdx.x / N_MEMBERS, k = blockIdx.x % N_MEMBERS, f = threadIdx.x;
    const int nt = p.n_tiles_dev ? min(*p.n_tiles_dev, p.n_tiles) : p.n_tiles;
    // the pair's tiles are consecutive in the table (ordered by row, member, tile): find the run with all threads (a
    // one-thread walk over ~750 entries was 100 us of dependent loads), then add its records in order
    __shared__ int first, count;
    if (threadIdx.x == 0) { first = 0x7fffffff; count = 0; }
    __syncthreads();
    for (int t = threadIdx.x; t < nt; t += blockDim.x) {
      const int4 tl = reinterpret_cast<const int4*>(p.tiles)[t];
      if (tl.x == row && tl.y == k && tl.w > 0) { atomicMin(&first, t); atomicAdd(&count, 1); }
    }
    __syncthreads();
    float a0 = 0.f, a2 = 0.f, ga = 0.f;
    for (int i = 0; i < count; ++i) {
      const int t = first + i;
      if (f < HID) { a0 += p.gbp[size_t(t) * 2 * HID + f]; a2 += p.gbp[(size_t(t) * 2 + 1) * HID + f]; }
      else if (f < HID + 3) ga += p.gap[size_t(t) * 4 + (f - HID)];
    }
    if (f < HID) { p.gb0[size_t(blockIdx.x) * HID + f] = a0; p.gb2[size_t(blockIdx.x) * HID + f] = a2; }
    else if (f < HID + 3 && k < N_LOC) p.ganch[(size_t(row) * N_LOC + k) * 3 + (f - HID)] = ga;
    return;
  }
  // 32 points per block: their (member, coordinate) shares go through LDS with coalesced loads (a thread per point read 640
  // strided bytes on its own: 20 us for 5 000 points), then 96 threads add 40 members each, member ascending
  constexpr int PB = 32;
  __shared__ float sh[PB][N_MEMBERS * 3 + 1];
  const int64_t total = int64_t(p.n_rows) * p.n_points;
  const int64_t pt0 = int64_t(blockIdx.x - n_pairs) * PB;
  for (int e = threadIdx.x; e < PB * N_MEMBERS * 3; e += blockDim.x) {
    const int q = e / (N_MEMBERS * 3), r = e % (N_MEMBERS * 3);
    const int64_t pt = pt0 + q;
    float v = 0.f;
    if (pt < total && p.what[pt * N_MEMBERS + r / 3] > 0.f) v = p.gxm[pt * N_MEMBERS * 3 + r];
    sh[q][r] = v;
  }
  __syncthreads();
  if (threadIdx.x < PB * 3) {
    const int q = threadIdx.x / 3, c = threadIdx.x % 3;
    const int64_t pt = pt0 + q;
    if (pt < total) {
      float a = 0.f;
#pragma unroll
      for (int k = 0; k < N_MEMBERS; ++k) a += sh[q][3 * k + c];
      p.gxyz[pt * 3 + c] = a;
    }
  }
}

// Workgroups walk the tile table with a grid stride.  A table built on the device has a fixed capacity and its
// number of used tiles lives in device memory (no host round trip): the launch is sized for the capacity but
// capped at a few workgroups per CU - an EMPTY workgroup of this kernel (512 threads, 57 KiB of LDS) still
// costs ~0.17 us of dispatch, 0.4 ms for the 2400 unused slots of a 5 x 1000-point fitting batch.
template <bool BWD>
__global__ __launch_bounds__(64 * WAVES, 2) void ident_member_kernel(BwdArgs p) {
  const int n = p.n_tiles_dev ? *p.n_tiles_dev : p.n_tiles;
  for (int t = blockIdx.x; t < n; t += gridDim.x) {
    member_tile<BWD>(p, t);
    __syncthreads();                       // the LDS staging arrays are reused by the next tile
  }
}

// ---- (row, member) point lists on the device ------------------------------------------------------------
// The member-centric kernels above work on lists of the points that keep a member under the pruning rule of
// the fused kernel (eval_kernel.hip, blend_masks: per point the smallest normalised blend weights are dropped
// while they sum to <= 40 * prune_tol; prune_tol < 0: every member).  Built here with FIXED capacity - every
// (row, member) pair owns a list segment of ceil(N / 64) * 64 entries and ceil(N / 64) tile slots (count 0 =
// unused) - so the host needs no size from the device: no synchronisation, static shapes (hipGraph capture).
//   weights_kernel : one thread per point - the 40 normalised weights (0 where pruned) into what[B,N,40]
//   lists_kernel   : one workgroup per (row, member) - order-preserving compaction of its points + tile slots
struct ListArgs {
  const float* state;       // [n_rows, LS_ROW_STRIDE]
  const float* xyz;         // [n_rows, n_points, 3]
  int n_rows;
  int64_t n_points;
  float prune_tol;
  float* what;              // [n_rows, n_points, 40]
  int* tiles;               // [n_rows * 40 * T][4]
  int* list;                // [n_rows * 40 * 64 T]
  int T;
};

__global__ __launch_bounds__(256) void weights_kernel(ListArgs p) {
  const int row = blockIdx.y;
  const int64_t n = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (n >= p.n_points) return;
  const float* anch = p.state + size_t(row) * LS_ROW_STRIDE + LS_OFF_ANCH;
  const float* q = p.xyz + (int64_t(row) * p.n_points + n) * 3;
  const float qx = q[0], qy = q[1], qz = q[2];
  float w[N_MEMBERS];
  float S = 0.f;
#pragma unroll
  for (int a = 0; a < N_LOC; ++a) {
    const float dx = anch[3 * a] - qx, dy = anch[3 * a + 1] - qy, dz = anch[3 * a + 2] - qz;
    const float d = sqrtf(dx * dx + dy * dy + dz * dz) + 1e-5f;
    w[a] = expf(-(d * d) / 0.01f);
    S += w[a];
  }
  w[N_LOC] = expf(-0.2f / 0.01f);
  const float denom = S + w[N_LOC] + 1e-6f;
#pragma unroll
  for (int a = 0; a < N_MEMBERS; ++a) w[a] /= denom;
  float cut = -1.f;                                   // prune_tol < 0: keep everything
  if (p.prune_tol >= 0.f) {
    cut = p.prune_tol;
    const float budget = float(N_MEMBERS) * p.prune_tol;
    const float mults[5] = {2.f, 4.f, 8.f, 16.f, 40.f};
#pragma unroll
    for (int c = 0; c < 5; ++c) {
      const float lim = mults[c] * p.prune_tol;
      float below = 0.f;
#pragma unroll
      for (int a = 0; a < N_MEMBERS; ++a) below += w[a] <= lim ? w[a] : 0.f;
      if (below <= budget) cut = lim;
    }
  }
  float* o = p.what + (int64_t(row) * p.n_points + n) * N_MEMBERS;
#pragma unroll
  for (int a = 0; a < N_MEMBERS; ++a) o[a] = w[a] > cut ? w[a] : 0.f;
}

__global__ __launch_bounds__(1024) void lists_kernel(ListArgs p) {
  __shared__ int wave_count[16];
  __shared__ int base;
  const int nw = blockDim.x >> 6;
  const int pair = blockIdx.x;                       // row * 40 + member
  const int row = pair / N_MEMBERS, k = pair % N_MEMBERS;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int cap = 64 * p.T;
  int* list = p.list + int64_t(pair) * cap;
  if (threadIdx.x == 0) base = 0;
  __syncthreads();
  for (int64_t n0 = 0; n0 < p.n_points; n0 += blockDim.x) {
    const int64_t n = n0 + threadIdx.x;
    const bool keep = n < p.n_points && p.what[(int64_t(row) * p.n_points + n) * N_MEMBERS + k] > 0.f;
    const unsigned long long b = __ballot(keep);
    if (lane == 0) wave_count[wave] = __popcll(b);
    __syncthreads();
    int off = base;
    for (int w = 0; w < wave; ++w) off += wave_count[w];
    if (keep) list[off + __popcll(b & ((1ull << lane) - 1ull))] = int(n);
    __syncthreads();
    if (threadIdx.x == 0) { int sum = 0; for (int w = 0; w < nw; ++w) sum += wave_count[w]; base += sum; }
    __syncthreads();
  }
  const int count = base;
  for (int t = threadIdx.x; t < p.T; t += blockDim.x) {
    int* tile = p.tiles + (int64_t(pair) * p.T + t) * 4;
    const int c = count - 64 * t;
    tile[0] = row; tile[1] = k; tile[2] = pair * cap + 64 * t; tile[3] = c < 0 ? 0 : (c > 64 ? 64 : c);
  }
}

// used tile slots -> front of the table, in slot order (deterministic); their number -> *n_used
__global__ __launch_bounds__(1024) void compact_tiles_kernel(const int* slots, int n_slots, int* tiles, int* n_used) {
  __shared__ int wave_count[16];
  __shared__ int base;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) base = 0;
  __syncthreads();
  for (int i0 = 0; i0 < n_slots; i0 += 1024) {
    const int i = i0 + threadIdx.x;
    int4 v = make_int4(0, 0, 0, 0);
    if (i < n_slots) v = reinterpret_cast<const int4*>(slots)[i];
    const bool used = v.w > 0;
    const unsigned long long b = __ballot(used);
    if (lane == 0) wave_count[wave] = __popcll(b);
    __syncthreads();
    int off = base;
    for (int w = 0; w < wave; ++w) off += wave_count[w];
    if (used) reinterpret_cast<int4*>(tiles)[off + __popcll(b & ((1ull << lane) - 1ull))] = v;
    __syncthreads();
    if (threadIdx.x == 0) { int s = 0; for (int w = 0; w < 16; ++w) s += wave_count[w]; base += s; }
    __syncthreads();
  }
  if (threadIdx.x == 0) *n_used = base;
}

}  // namespace bwd
}  // namespace nphm

// ============================================================================================
// C ABI (include/nphm_amd.h)
// ============================================================================================
extern "C" {

// grid of the member-centric kernels: one workgroup per tile; with a device-side tile count (capacity-sized
// table) at most 4 workgroups per CU, which then stride over the used tiles
static unsigned member_grid(int n_tiles, const int* n_tiles_dev) {
  if (!n_tiles_dev) return unsigned(n_tiles);
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  const int cap = 4 * (cus > 0 ? cus : 256);
  return unsigned(n_tiles < cap ? n_tiles : cap);
}

size_t nphm_identity_bwd_packed_bytes(void) { return size_t(nphm::N_SETS) * nphm::bwd::BWD_SET_STRIDE * 2; }

int nphm_identity_pack_bwd(const float* const lin_weight[5], void* packed_bwd, void* stream) {
  if (!packed_bwd) return nphm_fail_msg("nphm_identity_pack_bwd: null packed buffer");
  nphm::bwd::PackArgs a;
  for (int i = 0; i < 5; ++i) {
    if (!lin_weight[i]) return nphm_fail_msg("nphm_identity_pack_bwd: null weight pointer");
    a.w[i] = lin_weight[i];
  }
  a.out = static_cast<uint16_t*>(packed_bwd);
  dim3 g((nphm::bwd::BWD_SET_STRIDE + 255) / 256, nphm::N_SETS);
  hipLaunchKernelGGL(nphm::bwd::pack_bwd_kernel, g, dim3(256), 0, static_cast<hipStream_t>(stream), a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return nphm_fail("nphm_identity_pack_bwd launch", e);
  return 0;
}

int nphm_identity_list_tiles(int64_t n_points) { return n_points <= 0 ? 0 : int((n_points + 63) / 64); }

int nphm_identity_build_lists(const void* latent_state, const float* xyz, int n_rows, int64_t n_points, float prune_tol,
                              float* blend_weights, int* tiles, int* n_tiles_used, int* point_list, void* stream) {
  // (tiles = n_tiles_used = point_list = NULL: the blend weights alone - the training tier cuts its own, member-ordered lists)
  const bool weights_only = !tiles && !n_tiles_used && !point_list;
  if (!latent_state || !xyz || !blend_weights || (!weights_only && (!tiles || !n_tiles_used || !point_list)))
    return nphm_fail_msg("nphm_identity_build_lists: null pointer");
  if (n_rows <= 0 || n_points <= 0 || n_points > 0x7fffff00LL / (nphm::N_MEMBERS * int64_t(n_rows)) - 64)
    return nphm_fail_msg("nphm_identity_build_lists: bad sizes");
  nphm::bwd::ListArgs a;
  a.state = static_cast<const float*>(latent_state);
  a.xyz = xyz; a.n_rows = n_rows; a.n_points = n_points; a.prune_tol = prune_tol;
  a.T = nphm_identity_list_tiles(n_points);
  const int n_slots = n_rows * nphm::N_MEMBERS * a.T;
  a.what = blend_weights; a.tiles = weights_only ? nullptr : tiles + 4 * size_t(n_slots); a.list = point_list;     // slot table: second half
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(nphm::bwd::weights_kernel, dim3(unsigned((n_points + 255) / 256), n_rows), dim3(256), 0, st, a);
  if (weights_only) {
    hipError_t e0 = hipGetLastError();
    return e0 == hipSuccess ? 0 : nphm_fail("nphm_identity_build_lists launch", e0);
  }
  // a block per (row, member) walks the row's points: 1024 threads when the rows are long (one row of 5 000 points in the
  // fitting step: 5 rounds instead of 20 on the step's serial chain)
  hipLaunchKernelGGL(nphm::bwd::lists_kernel, dim3(n_rows * nphm::N_MEMBERS), dim3(n_points > 512 ? 1024 : 256), 0, st, a);
  hipLaunchKernelGGL(nphm::bwd::compact_tiles_kernel, dim3(1), dim3(1024), 0, st, a.tiles, n_slots, tiles, n_tiles_used);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return nphm_fail("nphm_identity_build_lists launch", e);
  return 0;
}

size_t nphm_identity_backward_scratch_bytes(int n_rows, int64_t n_points, int n_tiles) {
  if (n_rows <= 0 || n_points <= 0 || n_tiles < 0) return 0;
  return (size_t(n_tiles) * (2 * nphm::HID + 4) + size_t(n_rows) * size_t(n_points) * nphm::N_MEMBERS * 3) * 4;
}

int nphm_identity_backward(const void* packed, const void* packed_bwd, const void* latent_state,
                           const float* xyz, const float* sdf, const float* grad_sdf, int n_rows, int64_t n_points,
                           const int* tiles, int n_tiles, const int* n_tiles_dev, const int* point_list,
                           const float* blend_weights, void* scratch,
                           float* grad_xyz, float* grad_anchors, float* grad_b0, float* grad_b2, void* stream) {
  if (!packed || !packed_bwd || !latent_state || !xyz || !sdf || !grad_sdf || !tiles || !point_list || !blend_weights || !scratch ||
      !grad_xyz || !grad_anchors || !grad_b0 || !grad_b2)
    return nphm_fail_msg("nphm_identity_backward: null pointer");
  if (n_rows <= 0 || n_points <= 0 || n_tiles < 0) return nphm_fail_msg("nphm_identity_backward: bad sizes");
  nphm::bwd::BwdArgs a;
  a.packed_f32 = static_cast<const float*>(packed);
  a.packed_bf16 = reinterpret_cast<const uint16_t*>(static_cast<const char*>(packed) + nphm::PACKED_F32_FLOATS * 4);
  a.packed_bwd = static_cast<const uint16_t*>(packed_bwd);
  a.state = static_cast<const float*>(latent_state);
  a.xyz = xyz; a.sdf = sdf; a.gout = grad_sdf;
  a.tiles = tiles; a.list = point_list; a.n_points = n_points;
  a.n_tiles = n_tiles; a.n_tiles_dev = n_tiles_dev;
  a.gbp = static_cast<float*>(scratch);
  a.gap = a.gbp + size_t(n_tiles) * 2 * nphm::HID;
  a.gxm = a.gap + size_t(n_tiles) * 4;
  a.fmem = nullptr;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (n_tiles > 0)
    hipLaunchKernelGGL(nphm::bwd::ident_member_kernel<true>, dim3(member_grid(n_tiles, n_tiles_dev)), dim3(64 * nphm::bwd::WAVES), 0, st, a);
  nphm::bwd::ReduceArgs r;
  r.tiles = tiles; r.n_tiles = n_tiles; r.n_tiles_dev = n_tiles_dev;
  r.gbp = a.gbp; r.gap = a.gap; r.gxm = a.gxm; r.what = blend_weights;
  r.n_rows = n_rows; r.n_points = n_points;
  r.gxyz = grad_xyz; r.ganch = grad_anchors; r.gb0 = grad_b0; r.gb2 = grad_b2;
  const int64_t pts = int64_t(n_rows) * n_points;
  hipLaunchKernelGGL(nphm::bwd::ident_bwd_reduce_kernel, dim3(unsigned(n_rows * nphm::N_MEMBERS + (pts + 31) / 32)), dim3(256), 0, st, r);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return nphm_fail("nphm_identity_backward launch", e);
  return 0;
}

int nphm_identity_member_forward(const void* packed, const void* packed_bwd, const void* latent_state,
                                 const float* xyz, int64_t n_points, const int* tiles, int n_tiles,
                                 const int* n_tiles_dev, const int* point_list, float* member_sdf, void* stream) {
  if (!packed || !latent_state || !xyz || !tiles || !point_list || !member_sdf)
    return nphm_fail_msg("nphm_identity_member_forward: null pointer");
  if (n_points <= 0 || n_tiles < 0) return nphm_fail_msg("nphm_identity_member_forward: bad sizes");
  if (n_tiles == 0) return 0;
  nphm::bwd::BwdArgs a;
  memset(&a, 0, sizeof(a));
  a.packed_f32 = static_cast<const float*>(packed);
  a.packed_bf16 = reinterpret_cast<const uint16_t*>(static_cast<const char*>(packed) + nphm::PACKED_F32_FLOATS * 4);
  a.packed_bwd = static_cast<const uint16_t*>(packed_bwd);      // unused by the forward half
  a.state = static_cast<const float*>(latent_state);
  a.xyz = xyz;
  a.sdf = xyz; a.gout = xyz;                                      // read but ignored (valid memory)
  a.tiles = tiles; a.list = point_list; a.n_points = n_points;
  a.n_tiles = n_tiles; a.n_tiles_dev = n_tiles_dev;
  a.fmem = member_sdf;
  hipLaunchKernelGGL(nphm::bwd::ident_member_kernel<false>, dim3(member_grid(n_tiles, n_tiles_dev)), dim3(64 * nphm::bwd::WAVES), 0,
                     static_cast<hipStream_t>(stream), a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return nphm_fail("nphm_identity_member_forward launch", e);
  return 0;
}

}  // extern "C"

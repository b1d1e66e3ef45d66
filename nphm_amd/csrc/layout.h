// layout.h — HBM layout of the packed NPHM identity weights and of the per-latent state.
//
// Network (scripts/configs/nphm.yaml:1-7; EnsembledDeepSDF.py:80-90 of the reference): 40 members
// (39 anchors + background), 24 weight sets (16 symmetric pairs share), per member
//   lin0 99->200, lin1 200->101, [x(101) | inp(99)]/sqrt2 -> lin2 200, lin3 200->200, lin4 200->1.
// With a latent that is constant along the point axis the 96 latent columns of lin0 and of the
// skip part of lin2 collapse into per-member bias vectors (prepare_latent), leaving per point
//   L0: 3->200     L1: 200->101     L2: 104->200 (101 h1 + 3 coords)     L3: 200->200
//   L4: 200->1 (fused into L3's epilogue).
//
// Activation scaling.  Softplus(beta=100) is evaluated in base 2: with k = 100/ln2 the kernels
// carry a' = k*a instead of a, so that  a' = max(d',0) + log2(1 + 2^-|d'|)  for the scaled
// pre-activation d' = k*d needs no multiplies around v_exp_f32 / v_log_f32.  Linear layers commute
// with the scaling: weight matrices acting on activations are unchanged, everything additive
// (biases, the coordinate columns of lin0/lin2) is multiplied by k, lin4's weights are divided by k.
//
// Register layout.  All activations of a wavefront's 32 points live in registers in the C/D layout
// of v_mfma_f32_32x32x*: lane = 32*h + j holds, for point j, rows (r&3) + 8*(r>>2) + 4*h of a 32-row
// block in register r (r = 0..15).  A D register of one layer is therefore directly a B operand of
// the next layer; the matching permutation of the weight columns is applied once, at pack time.
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace nphm {

constexpr int N_MEMBERS = 40;
constexpr int N_LOC = 39;
constexpr int N_SYMM = 16;
constexpr int N_SETS = 24;
constexpr int HID = 200;
constexpr int L1_OUT = 101;
constexpr int L2_IN = 104;       // 101 + 3 coords
constexpr int LAT_GLOB = 64;
constexpr int LAT_LOC = 32;
constexpr int LAT_COND = 96;
constexpr int LAT_DIM = LAT_GLOB + N_MEMBERS * LAT_LOC;   // 1344
constexpr int D_IN = 99;
constexpr float SP_SCALE = 144.26950408889634f;           // k = 100 / ln 2
constexpr float INV_SQRT2_DIV = 1.41421356237f;

// feature index held by (block b, register r, half h) in the 32x32 C/D layout
__host__ __device__ constexpr int feat_of(int b, int r, int h) {
  return 32 * b + (r & 3) + 8 * (r >> 2) + 4 * h;
}

// ---- fp32 path (v_mfma_f32_32x32x2_f32): one K-step = 2 features (h = 0,1) ---------------------
constexpr int L1_KS = 100;   // 6 full blocks (96 steps) + block 6 regs 0..3
constexpr int L2_KS = 52;    // 3 full blocks (48) + block 3 regs 0..3
constexpr int L3_KS = 100;
constexpr int L1_OB = 4;     // 101 -> 128 rows
constexpr int L2_OB = 7;     // 200 -> 224 rows
constexpr int L3_OB = 7;

__host__ __device__ constexpr int ks_block(int ks, int full_blocks) {
  return ks < 16 * full_blocks ? (ks >> 4) : full_blocks;
}
__host__ __device__ constexpr int ks_reg(int ks, int full_blocks) {
  return ks < 16 * full_blocks ? (ks & 15) : ks - 16 * full_blocks;
}

// per-set offsets (in floats).  A-fragments are stored [ob][ks/4][lane][4] so one 16-byte load
// per lane fetches the fragments of 4 consecutive K-steps.
constexpr int OFF_L1A = 0;
constexpr int SZ_L1A = L1_OB * (L1_KS / 4) * 64 * 4;
constexpr int OFF_L2A = OFF_L1A + SZ_L1A;
constexpr int SZ_L2A = L2_OB * (L2_KS / 4) * 64 * 4;
constexpr int OFF_L3A = OFF_L2A + SZ_L2A;
constexpr int SZ_L3A = L3_OB * (L3_KS / 4) * 64 * 4;
constexpr int OFF_L4B = OFF_L3A + SZ_L3A;                   // lin4 bias (scalar)
constexpr int SET_STRIDE = OFF_L4B + 4;                     // floats per weight set

// ---- split-bf16 path (v_mfma_f32_32x32x16_bf16): one K-step = 16 k-slots ------------------------
// k-slot 8*h + i of K-step (b, s) is feature feat_of(b, 8*s + i, h); A fragments are 8 bf16 per
// lane (16 bytes), stored [ob][kstep][hi|lo][lane][8].
constexpr int L1_KS16 = 13;   // 6 blocks * 2 + 1
constexpr int L2_KS16 = 7;    // 3 blocks * 2 + 1
constexpr int L3_KS16 = 13;
constexpr int BF_OFF_L1A = 0;                                       // in uint16 units
constexpr int BF_SZ_L1A = L1_OB * L1_KS16 * 2 * 64 * 8;
constexpr int BF_OFF_L2A = BF_OFF_L1A + BF_SZ_L1A;
constexpr int BF_SZ_L2A = L2_OB * L2_KS16 * 2 * 64 * 8;
constexpr int BF_OFF_L3A = BF_OFF_L2A + BF_SZ_L2A;
constexpr int BF_SZ_L3A = L3_OB * L3_KS16 * 2 * 64 * 8;
constexpr int BF_SET_STRIDE = BF_OFF_L3A + BF_SZ_L3A;               // uint16 per weight set

// whole packed buffer: [fp32 sets | bf16 sets | f16 sets].  The split-f16 fragments have the layout of the split-bf16
// ones (same K order, [ob][kstep][hi|lo][lane][8]); the halves are IEEE binary16: 11-bit significands, so the
// three-term product carries 22 bits (bf16: 16) and the two-term / single-pass tiers are 8x closer to fp32; the lo
// halves of typical weights are f16 subnormals, which v_mfma_f32_32x32x16_f16 does not flush (tools/micro/f16split.hip)
constexpr size_t PACKED_F32_FLOATS = size_t(N_SETS) * SET_STRIDE;
constexpr size_t PACKED_BF16_HALFS = size_t(N_SETS) * BF_SET_STRIDE;
constexpr size_t PACKED_F16_HALFS = size_t(N_SETS) * BF_SET_STRIDE;
constexpr size_t PACKED_BYTES = PACKED_F32_FLOATS * 4 + PACKED_BF16_HALFS * 2 + PACKED_F16_HALFS * 2;

// ---- streaming units ---------------------------------------------------------------------------
// A member is consumed as 19 chunks: chunk 0 = its L0 block (lin0 restricted to the 3 coordinates
// plus the folded bias, as MFMA A fragments; member specific, from the per-latent state), chunks
// 1..18 = one 32-row output block of L1 (4), L2 (7), L3 (7) from the packed weight set.  Every GEMM
// chunk has a 64-float "tail" in the per-latent state: [h][r] accumulator init (layer bias * k, for
// L2 with the latent folded in) and, for L3, [h][r] lin4 weights / k for the fused 200->1 epilogue.
constexpr int GEMM_CHUNKS = L1_OB + L2_OB + L3_OB;          // 18
constexpr int CHUNKS_PER_MEMBER = GEMM_CHUNKS + 1;          // 19
constexpr int TAIL_FLOATS = 64;
constexpr int L0_BLOCK_FLOATS = 2048;                       // 8 KiB per member and precision
//   fp32 L0 block : float [ob 7][ks 2][lane 64]   A[i][k=h]: ks0 = (w_x, w_y), ks1 = (w_z, bias)
//   bf16 L0 block : bf16  [ob 7][lane 64][8]      one K=16 step, see prepare_latent_kernel
//   f16 L0 block  : the same with binary16 halves

// ---- per-latent state (one per batch row), in floats ----------------------------------------------
constexpr int LS_OFF_TAIL = 0;                                              // [member][18][64]
constexpr int LS_OFF_L0F = LS_OFF_TAIL + N_MEMBERS * GEMM_CHUNKS * TAIL_FLOATS;   // [member][2048]
constexpr int LS_OFF_L0B = LS_OFF_L0F + N_MEMBERS * L0_BLOCK_FLOATS;        // [member][2048]
constexpr int LS_OFF_L0H = LS_OFF_L0B + N_MEMBERS * L0_BLOCK_FLOATS;        // [member][2048] (f16 fragments)
constexpr int LS_OFF_ANCH = LS_OFF_L0H + N_MEMBERS * L0_BLOCK_FLOATS;       // anchors [39][3]
// magnitude bounds of the members, [member][4] = (b0, b1, b2, -): B_k(d) = b0 + b1 d + b2 d^2 >= |f_k| at distance d from
// anchor k.  The pruning rule and the precision tiers of the inference kernels act on w_k B_k(d_k) - the size a member's
// term can have - instead of the bare blend weight w_k; (1, 0, 0) (what prepare_latent writes) is the plain-weight rule.
// Fitted per checkpoint by numerics.calibrate_numerics, installed with nphm_identity_set_member_bounds.
constexpr int LS_OFF_BND = LS_OFF_ANCH + 128;
constexpr int LS_ROW_STRIDE = LS_OFF_BND + N_MEMBERS * 4;

__host__ __device__ constexpr int member_set(int k) {
  return k < 2 * N_SYMM ? (k >> 1) : N_SYMM + (k - 2 * N_SYMM);
}

}  // namespace nphm

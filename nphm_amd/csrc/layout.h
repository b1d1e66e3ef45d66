// layout.h — HBM layout of the packed NPHM identity weights and the per-latent state.
//
// Network (scripts/configs/nphm.yaml:1-7; EnsembledDeepSDF.py:80-90): 40 members (39 anchors +
// background), 24 weight sets (16 symmetric pairs share), per member
//   lin0 99->200, lin1 200->101, [x(101) | inp(99)]/sqrt2 -> lin2 200, lin3 200->200, lin4 200->1.
// With a latent that is constant along the point axis the 96 latent columns of lin0 and of the
// skip part of lin2 collapse into per-member bias vectors (prepare_latent), leaving per point
//   L0: 3->200 (VALU)   L1: 200->101 (MFMA)   L2: 104->200 (MFMA, 101 h1 + 3 coords)
//   L3: 200->200 (MFMA) L4: 200->1 (fused into L3's epilogue).
//
// All activations of a wavefront's 32 points live in registers in the C/D layout of
// v_mfma_f32_32x32x*: lane = 32*h + j holds, for point j, rows (r&3) + 8*(r>>2) + 4*h of a
// 32-row block in register r (r = 0..15).  A D register of one layer is therefore directly the
// B operand (k = h) of a K-step of the next layer; the matching permutation of the weight
// columns is applied once, here, at pack time.
#pragma once
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

// feature index held by (block b, register r, half h) in the 32x32 C/D layout
__host__ __device__ constexpr int feat_of(int b, int r, int h) {
  return 32 * b + (r & 3) + 8 * (r >> 2) + 4 * h;
}

// ---- fp32 path (32x32x2 f32 MFMA): one K-step = 2 features (h = 0,1) -------------------------
constexpr int L1_KS = 100;   // 6 full blocks (96 steps) + block 6 regs 0..3
constexpr int L2_KS = 52;    // 3 full blocks (48) + block 3 regs 0..3
constexpr int L3_KS = 100;
constexpr int L1_OB = 4;     // 101 -> 128 rows
constexpr int L2_OB = 7;     // 200 -> 224 rows
constexpr int L3_OB = 7;

// K-step -> (block, reg) of the input activation
__host__ __device__ constexpr int ks_block(int ks, int full_blocks) {
  return ks < 16 * full_blocks ? (ks >> 4) : full_blocks;
}
__host__ __device__ constexpr int ks_reg(int ks, int full_blocks) {
  return ks < 16 * full_blocks ? (ks & 15) : ks - 16 * full_blocks;
}

// per-set offsets (in floats).  A-fragments are stored [ob][ks/4][lane][4] so one 16-byte load
// per lane fetches the fragments of 4 consecutive K-steps.
constexpr int OFF_L0W = 0;                                  // float4 (wx,wy,wz,0) per (b,r,h): [7][16][2]
constexpr int SZ_L0W = 7 * 16 * 2 * 4;
constexpr int OFF_L1A = OFF_L0W + SZ_L0W;
constexpr int SZ_L1A = L1_OB * (L1_KS / 4) * 64 * 4;
constexpr int OFF_L2A = OFF_L1A + SZ_L1A;
constexpr int SZ_L2A = L2_OB * (L2_KS / 4) * 64 * 4;
constexpr int OFF_L3A = OFF_L2A + SZ_L2A;
constexpr int SZ_L3A = L3_OB * (L3_KS / 4) * 64 * 4;
constexpr int OFF_L4B = OFF_L3A + SZ_L3A;                   // lin4 bias (scalar)
constexpr int SET_STRIDE = OFF_L4B + 4;                     // floats per weight set

// The GEMM weights of a member are consumed as 18 chunks, one per 32-row output block:
// L1 ob 0..3, L2 ob 0..6, L3 ob 0..6.  Every chunk has a 64-float "tail" in the per-latent state:
// [h][r] accumulator init (layer bias, for L2 with the latent folded in) and, for L3, [h][r] lin4
// weights for the fused 200->1 epilogue.
constexpr int CHUNKS_PER_MEMBER = L1_OB + L2_OB + L3_OB;   // 18
constexpr int TAIL_FLOATS = 64;

// ---- split-bf16 path (32x32x16 bf16 MFMA): one K-step = 16 k-slots -------------------------
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

// whole packed buffer: [fp32 sets | bf16 sets]
constexpr size_t PACKED_F32_FLOATS = size_t(N_SETS) * SET_STRIDE;
constexpr size_t PACKED_BF16_HALFS = size_t(N_SETS) * BF_SET_STRIDE;
constexpr size_t PACKED_BYTES = PACKED_F32_FLOATS * 4 + PACKED_BF16_HALFS * 2;

// ---- per-latent state (one per batch row), in floats -----------------------------------------
constexpr int LS_OFF_B0 = 0;                       // folded lin0 bias, D layout [member][b][h][r]
constexpr int LS_OFF_TAIL = LS_OFF_B0 + N_MEMBERS * 224; // chunk tails [member][chunk][64]
constexpr int LS_OFF_ANCH = LS_OFF_TAIL + N_MEMBERS * CHUNKS_PER_MEMBER * TAIL_FLOATS;  // anchors [39][3]
constexpr int LS_ROW_STRIDE = LS_OFF_ANCH + 120;

__host__ __device__ constexpr int member_set(int k) {
  return k < 2 * N_SYMM ? (k >> 1) : N_SYMM + (k - 2 * N_SYMM);
}

}  // namespace nphm

// eval_kernel.hip — the hot kernel: fused 40-member MLP ensemble + Gaussian blend of the NPHM
// identity field (FastEnsembleDeepSDFMirrored.forward, src/NPHM/models/EnsembledDeepSDF.py:203-267
// of the reference; chunked grid form: get_logits, src/NPHM/models/reconstruction.py:6-25).
//
// Structure (see DESIGN.md for the numbers):
//   * one wavefront owns 32 query points for the whole network; activations never leave registers
//     (layout.h); one workgroup = NW wavefronts = a compact brick of 32*NW points;
//   * members whose normalised blend weight is <= prune_tol for every point of a wavefront are
//     skipped; the workgroup streams the union of its wavefronts' members;
//   * weights are streamed HBM/L2 -> LDS by LDS-DMA (MUBUF buffer_load ... lds) into a 5-slot ring shared
//     by the workgroup, one 32-row output block ("chunk") at a time, two chunks ahead of their use;
//   * per chunk: GEMM on the matrix pipe (fp32 MFMA, or 3 bf16 MFMAs per fp32 product; 1 for members
//     whose blend weight stays below 1e-3 in the wavefront) with the VALU epilogue of the PREVIOUS chunk
//     (base-2 softplus, re-split to bf16 hi/lo; bias is the accumulator init) threaded through its
//     dependent MFMA chain - one wavefront keeps both pipes busy;
//   * the last 32-row block of a layer holds 8 real rows: on the split-f16 path its fragment carries wh in rows 0..7 and wl in
//     rows 8..15 (two MFMAs per K-step for every tier, half the bytes to stream);
//   * 256 VGPRs inside the member loop: nothing lane-private LIVES across it in registers - the query point and the blend
//     normaliser are parked in LDS, wave-uniform masks sit in SGPRs, the output index is recomputed (no scratch);
//   * MODE 0 reads xyz[n,3]; MODE 1 generates the 'ij' lattice from three axis arrays (or reads
//     lattice-ordered displaced points: two-stage evaluation), bricks enumerated so that each XCD works
//     on a compact region; MODE 2 is the same lattice traversed tile by tile (4x4x2 voxels = one
//     wavefront) in the order of the tiles' active-member sets (tile_prepass_kernel + radix sort).
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>
#include <type_traits>

#include "capi_common.h"
#include "layout.h"

namespace nphm {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// PREC (template parameter of the kernel): 0 = fp32 MFMA, 1 = split bf16, 2 = split f16 (binary16 halves: same
// structure and fragment layout as 1, v_mfma_f32_32x32x16_f16; 11-bit significands)
struct EvalArgs {
  const float* packed_f32;
  const uint16_t* packed_bf16;
  const uint16_t* packed_f16;
  const float* state;       // [n_rows, LS_ROW_STRIDE]
  float* out;
  unsigned long long* stats;
  float prune_tol;
  float refine_band;        // > 0: sign-safe refinement - a wavefront whose blended value lands within this band of zero at any
  float refine_prune_tol;   // of its points re-evaluates its tile with every member on the three-pass product at this budget
  float light_tol;          // bf16 path: members below this normalised weight in a wavefront run single-pass
  float mid_tol;            // ... members below THIS one (and >= light_tol) two-pass: xh wh + xl wh, weights rounded to bf16
  // MODE 0 (points)
  const float* xyz;         // [n_rows, n_points, 3]
  int64_t n_points;
  // MODE 1 (grid)
  const float* ax; const float* ay; const float* az;
  int rx, ry, rz, ix0, ix1;
  const int* xplanes;       // optional: local x index -> global x plane (non-contiguous plane sets)
  int nbx, nby, nbz;        // bricks per axis
  int nsx, nsy, nsz;        // super-bricks (2x4x4 bricks) per axis
  int64_t hack_chunk;
  // MODE 2 (grid, tiles binned by member mask; written by tile_prepass_kernel + a radix sort)
  const unsigned* tile_order;   // slot -> tile id, tiles of equal member masks adjacent
  uint64_t* tile_masks;         // [tile][3]: members evaluated by the tile's wavefront, ... with more than one pass, ... with three passes
  float2* tile_sd;              // [tile][32]: (sum of blend weights, normaliser) of every point
  uint64_t* tile_keys;          // sort keys (pre-pass output)
  unsigned* tile_ids;           // identity permutation (pre-pass output)
  int ntx, nty, ntz;            // 4x4x2-voxel tiles per axis of the slab
  int n_tiles;
};

// 1: the LDS-DMA pieces of the prefetched chunk are issued between the K-steps of the GEMM in flight,
// 0: all of them right behind the chunk's barrier
#ifndef NPHM_DMA_INSTREAM
#define NPHM_DMA_INSTREAM 1
#endif
// timing ablations (results are garbage): 1 = no weight streaming, 2 = no workgroup barrier, 4 = no epilogue arithmetic,
// 16 = no wait for the weight DMA, 32 = every member streams weight set 0 / member 0's state (L2-resident stream),
// 64 = no softplus arithmetic for single-pass members only, 128 = the A fragments of a chunk are read from LDS once (K-step 0) and
// reused for every K-step (no repeated ds_read_b128: what the LDS read bandwidth costs), 256 = accumulators start from zero
// instead of the chunk's tail in LDS, 512 = no A fragment is read at all (a lane constant stands in), 1024 = no member is evaluated
// at all (what a workgroup costs before and after its member loop: launch, tile state, masks, the store)
#ifndef NPHM_ABLATE
#define NPHM_ABLATE 0
#endif
#ifndef NPHM_PERIOD
#define NPHM_PERIOD 2   // chunks per workgroup barrier
#endif
#ifndef NPHM_HEAVY_PASSES
#define NPHM_HEAVY_PASSES 3   // experiment: 2 = drop the wl x xh product of three-pass members (weights rounded to bf16)
#endif
#ifndef NPHM_SOFTPLUS4
#define NPHM_SOFTPLUS4 1   // 1: softplus as log2(1 + 2^d) + v_med3 (4 VALU), 0: max + log2(1 + 2^-|d|) (5 VALU)
#endif
#ifndef NPHM_LAUNDER_WAVE
#define NPHM_LAUNDER_WAVE 1   // 1: per-wavefront DMA offsets / conditions recomputed per site (SALU) instead of hoisted and spilled
#endif
#ifndef NPHM_SETPRIO
#define NPHM_SETPRIO 0        // 1: the second-dispatched half of the workgroup (waves 4..7) runs at s_setprio 1
#endif
// K-steps per back-to-back MFMA run of a GEMM chunk, per precision tier (0 = one epilogue slice behind every MFMA)
#ifndef NPHM_RUNK_HEAVY
#define NPHM_RUNK_HEAVY 0
#endif
#ifndef NPHM_RUNK_MID
#define NPHM_RUNK_MID 0
#endif
#ifndef NPHM_RUNK_LIGHT
#define NPHM_RUNK_LIGHT 0
#endif
// epilogue units (accumulator registers) of the previous chunk placed in front of a chunk's first MFMA, per tier (gemm_fused_bf16: HEAD)
#ifndef NPHM_HEAD_HEAVY
#define NPHM_HEAD_HEAVY 0
#endif
#ifndef NPHM_HEAD_MID
#define NPHM_HEAD_MID 0
#endif
#ifndef NPHM_HEAD_LIGHT
#define NPHM_HEAD_LIGHT 0
#endif
#ifndef NPHM_STACK_TAILS
#define NPHM_STACK_TAILS 1    // split-f16 path: the last 32-row block of a layer holds 8 real rows - its A fragment carries wh in rows
#endif                        // 0..7 and wl in rows 8..15 (prep_kernels.hip), two MFMAs per K-step instead of three, half the DMA bytes
#ifndef NPHM_M0_SAVE
#define NPHM_M0_SAVE 0        // 1: the LDS-DMA statements save and restore M0 around their use of it (rounds 1-5)
#endif
#ifndef NPHM_EPI_PAIRS
#define NPHM_EPI_PAIRS 1      // split formats, two- and three-term members: the epilogue works on register PAIRS (softplus2_pair) at the odd register
#endif
#ifndef NPHM_HOIST_BASES
#define NPHM_HOIST_BASES 1    // the two members' weight-set / tail byte offsets live in SGPRs across the member (Streamer::member_bases)
#endif
#ifndef NPHM_HI_ONLY
#define NPHM_HI_ONLY 0        // (measured +-0 twice: round 2 and round 6, tools/identity_variants.py - the stream's cost is not its bytes) split formats: a member that NO wavefront of the workgroup runs three-term is streamed without its wl
#endif                        // fragments (single- and two-term products read wh alone): half the bytes of those chunks, L2 -> LDS
#ifndef NPHM_LDS_STASH
#define NPHM_LDS_STASH 1      // per-lane (qx, qy, qz, denom) parked in LDS across the member loop instead of in VGPRs
#endif
#ifndef NPHM_PROF
#define NPHM_PROF 0  // timing builds only.  1: per-phase s_memtime accounting into stats[2..8] (two stamps + a drain per chunk: the
#endif               // launch takes ~3x as long) and the per-tier table; 2: the per-tier table with member totals alone (two stamps per member)
#if NPHM_PROF == 1
#define PROF_T(var) const long long var = clock64()
#define PROF_ADD(slot, t0, t1) prof[slot] += (t1) - (t0)
#else
#define PROF_T(var)
#define PROF_ADD(slot, t0, t1)
#endif

// nn.Softplus(beta=100, threshold=20) (EnsembledDeepSDF.py:99) in the scaled domain of layout.h:
// d' = k d, returns k softplus(d) = max(d',0) + log2(1 + 2^-|d'|).  For 100 d > 20 PyTorch returns d;
// here the log term is already 0 in fp32 for 100 d > 16.7, so the two agree to < 1e-9 in d units.
__device__ __forceinline__ float softplus2(float d) {
  if (NPHM_ABLATE & 4) return d;
#if NPHM_SOFTPLUS4
  // log2(1 + 2^d') directly: v_exp_f32, v_add_f32, v_log_f32 (raw instructions) and ONE v_med3_f32 that
  // returns d' itself once 2^d' has overflowed (d' > 127: the log is +inf; for d' <= 126 the log lies in
  // [d', 127], at the crossover it equals d') - 4 issue slots instead of 5.  Same accuracy class as the
  // max / log(1 + 2^-|d'|) form: 1 + 2^d' carries a relative rounding error of 2^-24, i.e. 9e-8 absolute
  // in the logarithm, next to the result's own ulp.
  const float r = __builtin_amdgcn_logf(1.f + __builtin_amdgcn_exp2f(d));
  return __builtin_amdgcn_fmed3f(d, r, 127.f);
#else
  const float t = __builtin_amdgcn_exp2f(-fabsf(d));                 // raw v_exp_f32
  // relu: built with -fno-honor-nans so that it is ONE v_max_f32 (otherwise a canonicalising
  // v_max(x, x) is put in front; inline asm is not an option - it would read MFMA results without
  // the hazard wait states the compiler inserts for its own instructions)
  return fmaxf(d, 0.f) + __builtin_amdgcn_logf(1.f + t);             // raw v_log_f32, arg in [1,2]
#endif
}

// Two values at once, the four instructions of each interleaved in program order: a transcendental's result must not be read by
// the very next VALU instruction (gfx950: one wait state - hipcc puts an s_nop between v_log_f32 and the v_med3_f32 of a lone
// softplus2, 470 issue slots of the kernel); with two values in flight every consumer is two instructions behind its producer.
__device__ __forceinline__ void softplus2_pair(float d0, float d1, float& x0, float& x1) {
  if (NPHM_ABLATE & 4) { x0 = d0; x1 = d1; return; }
#if NPHM_SOFTPLUS4
  float e0 = __builtin_amdgcn_exp2f(d0);
  float e1 = __builtin_amdgcn_exp2f(d1);
  e0 = 1.f + e0;
  e1 = 1.f + e1;
  e0 = __builtin_amdgcn_logf(e0);
  e1 = __builtin_amdgcn_logf(e1);
  x0 = __builtin_amdgcn_fmed3f(d0, e0, 127.f);
  x1 = __builtin_amdgcn_fmed3f(d1, e1, 127.f);
#else
  x0 = softplus2(d0); x1 = softplus2(d1);
#endif
}

// Softplus of the TWO-TERM members without transcendentals: max(d', 0) + q(min(|d'|, 16))^2 with q a degree-7 polynomial
// (weighted minimax fit of sqrt(log2(1 + 2^-u)) on [0, 16] with q(16) = 0), max abs error 5.7e-5 in the scaled domain = 3.9e-7
// in activation units, fp32, on register pairs (v_pk_fma_f32).  Built on the reading that the two-term body is VALU-bound by
// its 704 quarter-rate transcendentals - and MEASURED 5 % SLOWER (same box, interleaved three times: 27.9-28.0
// against 26.5-26.8 ms per 256^3 launch, same knobs): 1.6 x the VALU instructions of the body (2 039 -> 3 174) cost more
// than the 704 v_exp / v_log they replace - the transcendentals are not what the body waits for.  Off; kept as the record.
#ifndef NPHM_MID_POLY
#define NPHM_MID_POLY 0
#endif
typedef float f32pair __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void softplus2_pair_poly(float d0, float d1, float& x0, float& x1) {
  if (NPHM_ABLATE & 4) { x0 = d0; x1 = d1; return; }
  const f32pair u = {fminf(fabsf(d0), 16.f), fminf(fabsf(d1), 16.f)};
  const f32pair r = {fmaxf(d0, 0.f), fmaxf(d1, 0.f)};
  auto k = [](float c) __attribute__((always_inline)) { const f32pair v = {c, c}; return v; };
  f32pair q = k(5.23813419306407e-08f);
  q = __builtin_elementwise_fma(q, u, k(-3.1760125693836017e-06f));
  q = __builtin_elementwise_fma(q, u, k(7.663998258067295e-05f));
  q = __builtin_elementwise_fma(q, u, k(-0.0009056642302311957f));
  q = __builtin_elementwise_fma(q, u, k(0.0045111170038580894f));
  q = __builtin_elementwise_fma(q, u, k(0.010661209933459759f));
  q = __builtin_elementwise_fma(q, u, k(-0.24952375888824463f));
  q = __builtin_elementwise_fma(q, u, k(0.9999769926071167f));
  const f32pair f = __builtin_elementwise_fma(q, q, r);
  x0 = f[0]; x1 = f[1];
}

// Softplus for "light" members (adaptive precision: their GEMMs run single-pass, hi x hi only): the correction term
// g(u) = log2(1 + 2^-u), u = |d'|, as (a0 - a1 min(u, .))^8 - max abs error 4.2e-3 in the scaled
// domain (2.9e-5 in activation units, the size of the single-pass bf16 rounding these members already
// carry; it enters the blend scaled by a weight < light_tol).  Plain multiply-adds in place of the
// two quarter-rate transcendentals: a light chunk becomes MFMA-bound instead of VALU-bound.
#ifndef NPHM_LIGHT_POW
#define NPHM_LIGHT_POW 2   // power of the polynomial correction of softplus2_light: 8, 4 or 2 (max abs error 4.2e-3 / 1.6e-2 / 4.6e-2)
#endif
__device__ __forceinline__ float softplus2_light(float d) {
  if (NPHM_ABLATE & (4 | 64)) return d;      // 64: only the single-pass members' softplus
#if NPHM_LIGHT_POW == 8
  float q = fmaxf(fmaf(fabsf(d), -0.06564446f, 1.00028698f), 0.f);
  q *= q;
  q *= q;
#elif NPHM_LIGHT_POW == 4
  float q = fmaxf(fmaf(fabsf(d), -0.11704907f, 0.99590697f), 0.f);
  q *= q;
#else
  const float q = fmaxf(fmaf(fabsf(d), -0.18805397f, 0.9767937f), 0.f);
#endif
  return fmaf(q, q, fmaxf(d, 0.f));
}

#ifndef NPHM_LIGHT_POLY
#define NPHM_LIGHT_POLY 1
#endif
// registers of the LAST 32-row block of a layer that hold real features: 200 = 6*32 + 8 and
// 101 + 3 = 3*32 + 8 -> features 32b .. 32b+7 = registers 0..3 of both half-waves; the other 12
// registers are padding (zero weights downstream) and skip the epilogue arithmetic
constexpr int LAST_BLOCK_REGS = 4;

// one 32-feature block of activations as split-bf16 B operands of v_mfma_f32_32x32x16_bf16:
// K-step s consumes registers 8s..8s+7 of the block (k-slot 8h+i <-> register 8s+i)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
struct ActB {
  u32x4 hi[2], lo[2];     // dword q of hi[s] / lo[s] = k-slots 2q, 2q + 1 (bf16 pairs)
};
__device__ __forceinline__ bf16x8 as_bf16x8(const u32x4& v) { return __builtin_bit_cast(bf16x8, v); }
// one 32x32x16 MFMA on 16-byte operand fragments: bf16 (F16 = false) or binary16 halves
template <bool F16>
__device__ __forceinline__ f32x16 mfma16(const u32x4& a, const u32x4& b, const f32x16& c) {
  if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// two floats -> one dword of bf16 (round to nearest even): ONE v_cvt_pk_bf16_f32
__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {
  const f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}

__device__ __forceinline__ f32x16 load_frag16(const float* p) {
  // 16 consecutive floats (64-byte aligned) -> f32x16
  const f32x4* q = reinterpret_cast<const f32x4*>(p);
  f32x4 a = q[0], b = q[1], c = q[2], d = q[3];
  f32x16 o;
  o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; o[3] = a[3];
  o[4] = b[0]; o[5] = b[1]; o[6] = b[2]; o[7] = b[3];
  o[8] = c[0]; o[9] = c[1]; o[10] = c[2]; o[11] = c[3];
  o[12] = d[0]; o[13] = d[1]; o[14] = d[2]; o[15] = d[3];
  return o;
}

// compile-time loop: f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>)
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (N > 0) {
    static_for<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
}

// compile-time loop over [B, E)
template <int B, int E, class F>
__device__ __forceinline__ void static_range(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    static_range<B + 1, E>(f);
  }
}

// NU epilogue units spread evenly over NS issue slots: unit u runs in slot floor(u * NS / NU), i.e.
// slot s runs the units [unit_begin(s), unit_begin(s + 1))
__host__ __device__ constexpr int unit_begin(int s, int ns, int nu) { return (s * nu + ns - 1) / ns; }

// registers R, R+1 of an activation block -> their split-bf16 operand slots (one packed convert each
// for hi and lo); LIGHT members carry no lo part
template <int R, bool LIGHT, bool F16 = false>
__device__ __forceinline__ void pack_pair(const f32x16& a, ActB& o) {
  constexpr int s = R >> 3, q = (R & 7) >> 1;
  if constexpr (F16) {
    // binary16 halves.  hi: ONE v_cvt_pkrtz_f16_f32 - round toward zero, so a value beyond the f16 range saturates
    // at 65504 instead of becoming inf (the lo half then carries the rest: exact up to 131 008 in the scaled domain,
    // i.e. activations of 908); lo = x - hi straight into its half of the packed register: v_fma_mixlo / mixhi_f16
    // read the f16 hi half and the fp32 value in one instruction (no widening, no second packed convert):
    // 1.5 VALU per value against 3 on the bf16 path.  Inline asm is safe here: both inputs are VALU results
    // (the softplus), not MFMA results.
    const unsigned ph = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(a[R], a[R + 1]));
    o.hi[s][q] = ph;
    if constexpr (!LIGHT) {
      unsigned pl;
      asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(pl) : "v"(ph), "v"(a[R]));
      asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(pl) : "v"(ph), "v"(a[R + 1]));
      o.lo[s][q] = pl;
    }
    if constexpr (q == 3) {
      asm volatile("" : "+v"(o.hi[s]));
      if constexpr (!LIGHT) asm volatile("" : "+v"(o.lo[s]));
    }
    return;
  }
  // hi pair: one packed convert; its two halves widened back with a shift / a mask; lo pair: the residuals
  // through a second packed convert - 6 VALU per pair (written out: the generic __bf16 casts made hipcc
  // convert every value twice and shuffle the packed registers)
  const unsigned ph = cvt_pk_bf16(a[R], a[R + 1]);
  o.hi[s][q] = ph;
  if constexpr (!LIGHT) {
    const float h0 = __builtin_bit_cast(float, ph << 16), h1 = __builtin_bit_cast(float, ph & 0xffff0000u);
    o.lo[s][q] = cvt_pk_bf16(a[R] - h0, a[R + 1] - h1);
  }
  // pin the finished operand registers here: LLVM otherwise sinks the pure epilogue
  // arithmetic to its use in the next layer's GEMM and keeps the accumulator alive until then
  if constexpr (q == 3) {
    asm volatile("" : "+v"(o.hi[s]));
    if constexpr (!LIGHT) asm volatile("" : "+v"(o.lo[s]));
  }
}

// Single-pass ("light") members on the split-f16 path: the whole epilogue of a register PAIR on packed binary16 - the
// pre-activations are converted first (v_cvt_pkrtz), then |d| (one v_and), relu (v_pk_max), the polynomial's base
// q = clamp(a0 - a1 |d|) (v_pk_fma ... clamp) and q^2 + relu(d) (v_pk_fma): 5 VALU per pair, and the result IS the next layer's
// B operand (round 4: polynomial in fp32 per value, then the convert: ~9 per pair).  The tier's error class is unchanged:
// its products are hi x hi already, the polynomial's own error (4.6e-2 in the scaled domain) is 30 f16 ulps at d' = 1.
#ifndef NPHM_LIGHT_PK
#define NPHM_LIGHT_PK 1
#endif
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f16x2 light_pair(float a, float b) {
  const f16x2 d = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(a, b));
  if (NPHM_ABLATE & (4 | 64)) return d;
  const f16x2 zero = {(_Float16)0.f, (_Float16)0.f}, one = {(_Float16)1.f, (_Float16)1.f};
  const f16x2 c1 = {(_Float16)-0.18805397f, (_Float16)-0.18805397f}, c0 = {(_Float16)0.9767937f, (_Float16)0.9767937f};
  const f16x2 u = __builtin_elementwise_max(d, -d);
  f16x2 q = __builtin_elementwise_fma(u, c1, c0);
  q = __builtin_elementwise_min(__builtin_elementwise_max(q, zero), one);      // (folds into the fma's clamp bit: q <= a0 < 1)
  return __builtin_elementwise_fma(q, q, __builtin_elementwise_max(d, zero));
}
// registers R, R+1 of an accumulator (pre-activations) -> their hi operand slot
template <int R>
__device__ __forceinline__ void light_pack_pair(const f32x16& a, ActB& o) {
  constexpr int s = R >> 3, q = (R & 7) >> 1;
  o.hi[s][q] = __builtin_bit_cast(unsigned, light_pair(a[R], a[R + 1]));
  if constexpr (q == 3) asm volatile("" : "+v"(o.hi[s]));
}

// padding features of a layer's last block (NR real registers): zero hi operands
template <int NR>
__device__ __forceinline__ void zero_pad_hi(ActB& o) {
#pragma unroll
  for (int q = NR / 2; q < 4; ++q) o.hi[0][q] = 0u;
  asm volatile("" : "+v"(o.hi[0]));
}

// ---- workgroup geometry ----------------------------------------------------------------------
// NW = wavefronts per workgroup (all of them share one LDS ring): 8 -> 256 points per weight pass
#ifndef NPHM_NW
#define NPHM_NW 8
#endif
constexpr int NW = NPHM_NW;
// voxel brick of a workgroup in grid mode: every wavefront owns a 4x4x2 sub-brick
constexpr int BRX = NW == 8 ? 8 : 4, BRY = NW == 8 ? 8 : 4, BRZ = NW == 8 ? 4 : 8;
// voxel tile of ONE wavefront (32 points): TLX x TLY x TLZ, z fastest: 4 x 4 x 2.  2 x 4 x 4 (-DNPHM_TILE_Z=4: 16-byte instead of
// 8-byte z-runs per store, the same compactness - 8.6 x 19.6 x 21.2 against 17.2 x 19.6 x 10.6 lattice spacings of the fitting box
// at 256^3) measured WRITE_SIZE 155 -> 131 MB per 256^3 launch (67 MB of payload) at -0.8 % throughput: not taken, the
// writes are 0.4 % of the launch's HBM time either way.
#ifndef NPHM_TILE_Z
#define NPHM_TILE_Z 2
#endif
constexpr int TLZ = NPHM_TILE_Z, TLY = 4, TLX = 32 / (TLY * TLZ);
static_assert(TLX * TLY * TLZ == 32 && (TLZ == 2 || TLZ == 4), "a wavefront's tile holds 32 voxels");
__host__ __device__ constexpr int tile_dx(int j) { return j / (TLY * TLZ); }
__host__ __device__ constexpr int tile_dy(int j) { return (j / TLZ) % TLY; }
__host__ __device__ constexpr int tile_dz(int j) { return j % TLZ; }
// bricks are enumerated super-brick by super-brick (2x4x4 bricks) so that the workgroups resident
// on one XCD at any time cover a compact region and stream the same few members (L2 reuse)
constexpr int SBX = 2, SBY = 4, SBZ = 4;

// ---- weight streaming: global -> LDS ring, shared by the NW wavefronts of a workgroup ----------
// Units of 1 KiB "groups" (one global_load_lds_dwordx4 of a wavefront).  Chunk 0 of a member is its
// L0 block (8 groups, per-latent state); chunks 1..18 are GEMM blocks of the packed weight set:
//   fp32 : L1 25, L2 13, L3 25 groups ([ks/4][lane][4] floats)
//   bf16 : L1 26, L2 14, L3 26 groups ([ks][hi|lo][lane][8] bf16)
// plus a 256-byte tail (accumulator init / lin4 weights).  (19 + 1 phantom) % RING == 0 keeps the
// ring slot of every chunk a compile-time constant.
constexpr int RING = 5;
static_assert((CHUNKS_PER_MEMBER + 1) % RING == 0, "ring slots must be static per chunk index");

template <int PREC> struct Stream;
template <> struct Stream<0> {
  static constexpr int MAIN_BYTES = (L1_KS / 4) * 1024;
  __device__ static __forceinline__ const char* set_base(const EvalArgs& p, int s) {
    return reinterpret_cast<const char*>(p.packed_f32 + size_t(s) * SET_STRIDE);
  }
  static constexpr int LS_OFF_L0 = LS_OFF_L0F;
  static constexpr unsigned SET_BYTES = SET_STRIDE * 4u;
  __host__ __device__ static constexpr int offset(int g) {   // bytes inside a weight set, g = GEMM chunk
    return 4 * (g < L1_OB ? OFF_L1A + g * (L1_KS / 4) * 256
              : g < L1_OB + L2_OB ? OFF_L2A + (g - L1_OB) * (L2_KS / 4) * 256
                                  : OFF_L3A + (g - L1_OB - L2_OB) * (L3_KS / 4) * 256);
  }
  __host__ __device__ static constexpr int groups(int ci) {  // ci = chunk index inside the member
    return ci == 0 ? 8 : (ci - 1 >= L1_OB && ci - 1 < L1_OB + L2_OB) ? (L2_KS / 4) : (L1_KS / 4);
  }
};
template <> struct Stream<1> {
  static constexpr int MAIN_BYTES = L1_KS16 * 2 * 1024;
  __device__ static __forceinline__ const char* set_base(const EvalArgs& p, int s) {
    return reinterpret_cast<const char*>(p.packed_bf16 + size_t(s) * BF_SET_STRIDE);
  }
  static constexpr int LS_OFF_L0 = LS_OFF_L0B;
  static constexpr unsigned SET_BYTES = BF_SET_STRIDE * 2u;
  __host__ __device__ static constexpr int offset(int g) {
    return 2 * (g < L1_OB ? BF_OFF_L1A + g * L1_KS16 * 1024
              : g < L1_OB + L2_OB ? BF_OFF_L2A + (g - L1_OB) * L2_KS16 * 1024
                                  : BF_OFF_L3A + (g - L1_OB - L2_OB) * L3_KS16 * 1024);
  }
  __host__ __device__ static constexpr int groups(int ci) {
    return ci == 0 ? 8 : (ci - 1 >= L1_OB && ci - 1 < L1_OB + L2_OB) ? 2 * L2_KS16 : 2 * L1_KS16;
  }
};

// chunk index (inside the member) of the LAST 32-row block of lin1 / lin2 / lin3: 8 real rows (5 + the coordinate slots for lin1)
__host__ __device__ constexpr bool is_tail_chunk(int ci) { return ci == L1_OB || ci == L1_OB + L2_OB || ci == CHUNKS_PER_MEMBER - 1; }

template <> struct Stream<2> : Stream<1> {
  __device__ static __forceinline__ const char* set_base(const EvalArgs& p, int s) {
    return reinterpret_cast<const char*>(p.packed_f16 + size_t(s) * BF_SET_STRIDE);
  }
  static constexpr int LS_OFF_L0 = LS_OFF_L0H;
  // stacked tail blocks: ONE fragment per K-step ([ks][lane][8]: wh in rows 0..7, wl in rows 8..15), at the front of the chunk
  __host__ __device__ static constexpr int groups(int ci) {
    return (NPHM_STACK_TAILS && is_tail_chunk(ci)) ? Stream<1>::groups(ci) / 2 : Stream<1>::groups(ci);
  }
};

template <int PREC>
struct Streamer {
  static constexpr int SLOT_BYTES = Stream<PREC>::MAIN_BYTES + TAIL_FLOATS * 4;
  const EvalArgs& p;
  const float* st;             // per-latent state of this batch row
  char* ring;                  // LDS, RING * SLOT_BYTES
  const unsigned char* list;   // LDS, active member ids of this workgroup
  int n_active;
  int mi;                      // index (into list) of the member being consumed
  int wave, lane;

  // LDS-DMA through inline asm: hipcc does not see these loads, so it neither drains them with a
  // vmcnt(0) in front of every later ds_read (which it does for the builtin: the DMA is a pending
  // LDS write it cannot disambiguate) nor counts them - sync() waits for exactly the chunk it
  // needs.  The loads are MUBUF (buffer_load ... offen lds): resource = base of the stream, the VGPR
  // offset is the lane's constant 16 (4) bytes, chunk / group offsets are SGPRs and consecutive 1 KiB
  // groups share ONE M0 (LDS destination) through the instruction offset, which moves both the memory
  // and the LDS address.  Measured next to an MFMA stream (tools/micro/dma.hip): 35 cycles of
  // wave-time per 1 KiB piece against 229 for global_load_lds with per-piece 64-bit VGPR addresses.
  typedef int v4i __attribute__((ext_vector_type(4)));
  v4i rs_w, rs_s;              // buffer resources: packed weight sets, per-latent state row
  __device__ static __forceinline__ v4i make_rsrc(const void* base) {
    const uint64_t a = reinterpret_cast<uint64_t>(base);
    v4i r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(uint32_t)a);
    r[1] = __builtin_amdgcn_readfirstlane((int)((uint32_t)(a >> 32) & 0xffffu));   // stride 0: raw buffer
    r[2] = 0x7fffffff;         // bytes addressable from base
    r[3] = 0x00020000;         // gfx950 raw-buffer descriptor word
    return r;
  }
  // M0 is a reserved register for hipcc: it is saved and restored inside the statement
  template <int N> __device__ static __forceinline__ void dma16(const v4i& rsrc, unsigned voff, unsigned soff, unsigned lds_dst) {
    static_assert(N >= 1 && N <= 4, "instruction offsets 0 .. 3072");
#if NPHM_M0_SAVE
    unsigned keep;
#define NPHM_DMA_HEAD "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\t"
#define NPHM_DMA_TAIL "s_mov_b32 m0, %0"
#define NPHM_DMA_ARGS : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(lds_dst), "s"(soff) : "memory"
#else
    // M0 is written and left: nothing else in these kernels reads it (LDS instructions of gfx9+ do not; checked in the ISA:
    // every m0 of eval_kernel sits inside these statements) - two SALU per piece less than saving and restoring it
#define NPHM_DMA_HEAD "s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %3 offen lds\n\t"
#define NPHM_DMA_TAIL ""
#define NPHM_DMA_ARGS : : "v"(voff), "s"(rsrc), "s"(lds_dst), "s"(soff) : "memory"
#endif
#if NPHM_M0_SAVE
#define NPHM_DMA_MORE(off) "buffer_load_dwordx4 %1, %2, %4 offen offset:" #off " lds\n\t"
#else
#define NPHM_DMA_MORE(off) "buffer_load_dwordx4 %0, %1, %3 offen offset:" #off " lds\n\t"
#endif
    if constexpr (N == 1) asm volatile(NPHM_DMA_HEAD NPHM_DMA_TAIL NPHM_DMA_ARGS);
    else if constexpr (N == 2) asm volatile(NPHM_DMA_HEAD NPHM_DMA_MORE(1024) NPHM_DMA_TAIL NPHM_DMA_ARGS);
    else if constexpr (N == 3) asm volatile(NPHM_DMA_HEAD NPHM_DMA_MORE(1024) NPHM_DMA_MORE(2048) NPHM_DMA_TAIL NPHM_DMA_ARGS);
    else asm volatile(NPHM_DMA_HEAD NPHM_DMA_MORE(1024) NPHM_DMA_MORE(2048) NPHM_DMA_MORE(3072) NPHM_DMA_TAIL NPHM_DMA_ARGS);
#undef NPHM_DMA_HEAD
#undef NPHM_DMA_MORE
#undef NPHM_DMA_TAIL
#undef NPHM_DMA_ARGS
  }
  __device__ static __forceinline__ void dma4(const v4i& rsrc, unsigned voff, unsigned soff, unsigned lds_dst) {
#if NPHM_M0_SAVE
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dword %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(lds_dst), "s"(soff) : "memory");
#else
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dword %0, %1, %3 offen lds"
                 : : "v"(voff), "s"(rsrc), "s"(lds_dst), "s"(soff) : "memory");
#endif
  }
  __device__ static __forceinline__ unsigned lds_addr(const char* q) {
    return (unsigned)(size_t)(__attribute__((address_space(3))) const char*)q;
  }

  int k_cur, k_nxt;            // ids of the member being consumed and of the next one (-1: none)
  uint64_t lo_mask = ~0ull;    // members whose wl fragments some wavefront of the workgroup reads (three-term members)
  // byte offsets of the two members' weight set / chunk tails inside their buffers: computed ONCE per member (next_member) and
  // kept in four SGPRs.  Rounds 4-5 recomputed member -> set -> offset at every one of the ~150 DMA sites (10 SALU each, a
  // quarter of a single-term member's instruction stream) to keep hipcc from hoisting per-SITE values out of the member loop;
  // the per-site values are still derived inside the loop - from these four, laundered at the site
  unsigned mb_cur = 0, mb_nxt = 0, tb_cur = 0, tb_nxt = 0;
  __device__ __forceinline__ void member_bases(int k, unsigned& mb, unsigned& tb) const {
    const int kc = k < 0 ? 0 : ((NPHM_ABLATE & 32) ? 0 : k);
    mb = unsigned(member_set(kc)) * Stream<PREC>::SET_BYTES;
    tb = unsigned(LS_OFF_TAIL + kc * GEMM_CHUNKS * TAIL_FLOATS) * 4u;
  }

  __device__ __forceinline__ void load_ids() {
    k_cur = mi < n_active ? __builtin_amdgcn_readfirstlane(int(list[mi])) : -1;
    k_nxt = mi + 1 < n_active ? __builtin_amdgcn_readfirstlane(int(list[mi + 1])) : -1;
    member_bases(k_cur, mb_cur, tb_cur);
    member_bases(k_nxt, mb_nxt, tb_nxt);
  }
  // Fetch of chunk `ci` of the current (next = false) or the next member into its ring slot, in pieces.
  // A chunk of ng 1 KiB groups: piece 0 = Q = ng / NW consecutive groups per wavefront (one M0),
  // piece 1 = the ng % NW left-over groups, one each for the first wavefronts, piece 2 = the 256-byte
  // tail (one wavefront).  Every wavefront issues at least Q loads per chunk (sync() counts on it).
  static constexpr int PIECES = 3;
  __device__ __forceinline__ void issue_piece(const bool next, const int ci, const int i) const {
    int k = next ? k_nxt : k_cur;
    if (k < 0 || (NPHM_ABLATE & 1)) return;
    if (NPHM_ABLATE & 32) k = 0;
    int l = lane;
    asm volatile("" : "+v"(l));           // the lane offset is recomputed per site, not kept live
#if NPHM_LAUNDER_WAVE
    // ... and so is everything derived from the wavefront index: hipcc otherwise hoists `wave * 1024 + <site constant>`
    // and the `wave == c` / `wave < rem` conditions of all ~150 DMA sites out of the member loop (loop invariants), runs
    // out of SGPRs, spills 181 of them to VGPR lanes and reads them back with v_readlane inside the MFMA stream
    int wave = this->wave;
    asm volatile("" : "+s"(wave));
#endif
    const unsigned dst = __builtin_amdgcn_readfirstlane(lds_addr(ring + (ci % RING) * SLOT_BYTES));
    const int ng = Stream<PREC>::groups(ci);
    const int q = ng / NW, rem = ng - q * NW;
#if NPHM_HOIST_BASES
    unsigned mb = next ? mb_nxt : mb_cur;
    asm volatile("" : "+s"(mb));
    const unsigned base = ci == 0 ? unsigned(Stream<PREC>::LS_OFF_L0 + k * L0_BLOCK_FLOATS) * 4u
                                  : mb + unsigned(Stream<PREC>::offset(ci - 1));
#else
    const unsigned base = ci == 0 ? unsigned(Stream<PREC>::LS_OFF_L0 + k * L0_BLOCK_FLOATS) * 4u
                                  : unsigned(member_set(k)) * Stream<PREC>::SET_BYTES + unsigned(Stream<PREC>::offset(ci - 1));
#endif
    const v4i& rs = ci == 0 ? rs_s : rs_w;
    if constexpr (NPHM_HI_ONLY && PREC >= 1 && PERIOD == 2) {
      // [ks][hi | lo] fragments of 1 KiB: only the even groups, group 2 s from wavefront s % NW (LDS layout unchanged - the lo
      // slots keep stale bytes nobody reads).  Not for chunk 0 (the L0 block) and the stacked tail blocks (one fragment per K-step).
      if (ci > 0 && Stream<PREC>::groups(ci) == Stream<1>::groups(ci) && !((lo_mask >> k) & 1ull)) {
        if (i < 2) {
          const int sidx = wave + NW * i;
          if (sidx < ng / 2) {
            const unsigned g = unsigned(sidx) * 2048u;
            dma16<1>(rs, l * 16, __builtin_amdgcn_readfirstlane(base + g), __builtin_amdgcn_readfirstlane(dst + g));
          }
          return;
        }
      }
    }
    if (i == 0) {
      if (q == 0) return;                   // (a stacked lin2 tail: 7 groups, all of them "left-over" pieces)
      const unsigned g = unsigned(q * wave) * 1024u;
      const unsigned soff = __builtin_amdgcn_readfirstlane(base + g), d = __builtin_amdgcn_readfirstlane(dst + g);
      if (q == 1) dma16<1>(rs, l * 16, soff, d);
      else if (q == 2) dma16<2>(rs, l * 16, soff, d);
      else if (q == 3) dma16<3>(rs, l * 16, soff, d);
      else dma16<4>(rs, l * 16, soff, d);
    } else if (i == 1) {
      if (wave < rem) {
        const unsigned g = unsigned(q * NW + wave) * 1024u;
        dma16<1>(rs, l * 16, __builtin_amdgcn_readfirstlane(base + g), __builtin_amdgcn_readfirstlane(dst + g));
      }
    } else if (ci > 0 && wave == (ci & (NW - 1))) {
#if NPHM_HOIST_BASES
      unsigned tb = next ? tb_nxt : tb_cur;
      asm volatile("" : "+s"(tb));
      const unsigned soff = tb + unsigned((ci - 1) * TAIL_FLOATS) * 4u;
#else
      const unsigned soff = unsigned(LS_OFF_TAIL + (k * GEMM_CHUNKS + ci - 1) * TAIL_FLOATS) * 4u;
#endif
      dma4(rs_s, l * 4, __builtin_amdgcn_readfirstlane(soff), dst + Stream<PREC>::MAIN_BYTES);
    }
  }
  __device__ __forceinline__ void issue(const bool next, const int ci) const {
#pragma unroll
    for (int i = 0; i < PIECES; ++i) issue_piece(next, ci, i);
  }
  // Ring positions: chunk ci of the member being consumed sits at position ci, position 19 is a phantom
  // (20 positions per member keep the slot of every chunk a compile-time constant), chunk ci of the NEXT
  // member sits at 20 + ci.  The workgroup meets at ONE barrier per PERIOD = 2 chunks: at the barrier in
  // front of chunks c, c + 1 (c even) every wavefront has left chunks <= c - 1, so the pieces of positions
  // c + 2, c + 3 can go out - their slots held c - 3 and c - 2; the slot of chunk c - 1 stays intact, its
  // tail is still read by the lin3 epilogue fused into chunk c.  Nothing younger than chunks c, c + 1 is in
  // flight at that barrier, so a wavefront simply drains its DMAs (vmcnt 0) in front of it.
  // (PERIOD = 1: a barrier per chunk, the pieces of chunk c + 2 go out in step c; 3 % slower.)
  static constexpr int PERIOD = NPHM_PERIOD;
  static_assert(PERIOD == 1 || PERIOD == 2, "one barrier per chunk or per pair of chunks");
  static constexpr int POS_NEXT = CHUNKS_PER_MEMBER + 1;
  template <int T> __device__ __forceinline__ void issue_position(const int i) const {
    if constexpr (T < CHUNKS_PER_MEMBER) issue_piece(false, T, i);
    else if constexpr (PERIOD == 1) issue_piece(true, T - CHUNKS_PER_MEMBER, i);       // steps 17, 18: next member's chunks 0, 1
    else if constexpr (T >= POS_NEXT) issue_piece(true, T - POS_NEXT, i);
  }
  template <int CI> static constexpr int prefetch_pieces() { return PERIOD == 1 ? PIECES : (CI % 2 == 0 ? 2 * PIECES : 0); }
  template <int CI> __device__ __forceinline__ void prefetch_piece(const int i) const {
    if (i < PIECES) issue_position<CI + 2>(i); else issue_position<CI + 3>(i - PIECES);
  }
  template <int CI> __device__ __forceinline__ void prefetch() const {
#pragma unroll
    for (int i = 0; i < prefetch_pieces<CI>(); ++i) prefetch_piece<CI>(i);
  }
  template <int N> __device__ static __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" :: "i"(N) : "memory");
  }
  // Every wavefront of the workgroup calls sync<CI>() exactly once per chunk, in the same order.
  template <int CI> __device__ __forceinline__ void sync() {
    if constexpr (PERIOD == 2 && CI % 2 == 1) return;
    // PERIOD 1: chunk CI must have landed; the only younger DMAs of this wavefront are those of chunk
    // CI + 1, at least groups / NW of them, and VMEM completes in order: waiting until at most that many
    // operations are outstanding retires every load of chunk CI.
    constexpr int T = CI + 1;
#if NPHM_PROF == 1
    const long long ta = clock64();
#endif
    if constexpr ((NPHM_ABLATE & 16) != 0) {
    } else if constexpr (PERIOD == 2) {
      wait_vm<0>();
    } else if constexpr (T < CHUNKS_PER_MEMBER) {
      wait_vm<Stream<PREC>::groups(T) / NW>();
    } else {
      if (k_nxt >= 0) wait_vm<Stream<PREC>::groups(0) / NW>(); else wait_vm<0>();
    }
#if NPHM_PROF == 1
    const long long tb = clock64();
#endif
    if constexpr (!(NPHM_ABLATE & 2)) __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#if NPHM_PROF == 1
    const long long tc = clock64();
#endif
    if constexpr (!NPHM_DMA_INSTREAM) prefetch<CI>();
#if NPHM_PROF == 1
    const long long td = clock64();
    sprof[0] += tb - ta; sprof[1] += tc - tb; sprof[2] += td - tc;
#endif
  }
#if NPHM_PROF
  long long sprof[3] = {0, 0, 0};   // vmcnt wait, barrier wait, DMA issue
#endif
  __device__ __forceinline__ const char* slot(const int CI) const { return ring + (CI % RING) * SLOT_BYTES; }
  __device__ __forceinline__ void next_member() {
    ++mi;
    k_cur = k_nxt;
    k_nxt = mi + 1 < n_active ? __builtin_amdgcn_readfirstlane(int(list[mi + 1])) : -1;
    mb_cur = mb_nxt; tb_cur = tb_nxt;
    member_bases(k_nxt, mb_nxt, tb_nxt);
    asm volatile("" : "+s"(k_cur), "+s"(k_nxt));   // opaque: nothing per-site hoisted out of the loop
    asm volatile("" : "+s"(mb_cur), "+s"(mb_nxt), "+s"(tb_cur), "+s"(tb_nxt));
  }
  __device__ static __forceinline__ const float* tail_of(const char* buf) {
    return reinterpret_cast<const float*>(buf + Stream<PREC>::MAIN_BYTES);
  }
};

#ifndef NPHM_L0_FUSED
#define NPHM_L0_FUSED 1
#endif
#ifndef NPHM_PF_HEAVY
#define NPHM_PF_HEAVY 1
#endif
#ifndef NPHM_PF_LIGHT
#define NPHM_PF_LIGHT 3
#endif
// Epilogue bookkeeping of the chunk pipeline: number of accumulator registers of chunk P that hold
// real features, and the chunks whose epilogue runs "exposed" right behind their own GEMM instead of
// inside the next chunk's: the member's last chunk, and the next-to-last chunk of lin1 and of lin2 -
// fusing those two would hold 13 live 16-register blocks + 2 accumulators + the A fragments (> 256 VGPRs)
__host__ __device__ constexpr int epilogue_units(int P) {
  return (P == L1_OB || P == L1_OB + L2_OB || P == CHUNKS_PER_MEMBER - 1) ? LAST_BLOCK_REGS : 16;
}
#ifndef NPHM_EXPOSE_A
#define NPHM_EXPOSE_A -1
#endif
#ifndef NPHM_EXPOSE_B
#define NPHM_EXPOSE_B -1
#endif
__host__ __device__ constexpr bool epilogue_exposed(int P) {
  return P == NPHM_EXPOSE_A || P == NPHM_EXPOSE_B || P == CHUNKS_PER_MEMBER - 1;
}

// ---- GEMM of chunk c fused with the epilogue of chunk c-1 ----------------------------------------
// A wavefront issues in order: the VALU epilogue of the previous chunk is cut into NU units and one
// slice is placed behind every MFMA of this chunk's (dependent) accumulation chain, where it executes
// in the shadow of that MFMA (32 cycles of matrix pipe, 4 of issue) - measured in
// tools/micro/overlap.hip: MFMA + softplus interleaved in one wavefront cost max(...) + ~15 %, not the sum.
// sched_barrier(0) after every slot keeps hipcc from regrouping the stream.
// HEAD: epilogue units of the previous chunk that run BEFORE this chunk's first MFMA, behind the LDS reads of its first A
// fragments and accumulator init - the VALU work covers that latency (the reads cannot be issued earlier: a barrier, or
// the previous chunk's own MFMAs, are in front of them), the remaining units ride behind the MFMAs as before
template <int NKS16, int FULL, int NIN, int NPASS, int PF, int NU, bool F16, int RUNK, int AS, int HEAD, class Epi, class Pre, class Slot>
__device__ __forceinline__ f32x16 gemm_fused_bf16(const char* afrag, f32x16 acc, const ActB (&in)[NIN],
                                                  int lane, Epi&& epi, Pre&& pre, Slot&& slot_hook) {
  static_assert(AS == 2 || (AS == 1 && NPASS <= 2), "AS = 1: stacked tail fragments (one fragment per K-step, no separate lo fragment)");
  const u32x4* A = reinterpret_cast<const u32x4*>(afrag) + lane;
  u32x4 wh[NKS16], wl[NKS16];
  constexpr int NM = NPASS;                // MFMAs per K-step: hh | hh, hl | hh, hl, lh
  constexpr int NS = NKS16 * NM;           // issue slots
  if constexpr (RUNK > 0) {
    // RUNS: the MFMAs of RUNK consecutive K-steps issue back to back with nothing between them; the LDS reads of the
    // NEXT run's A fragments, the epilogue units and slot hooks of this run's slots follow it, the DMA pieces precede
    // it.  A dependent MFMA (same accumulator) that directly follows its predecessor issues at the 32-cycle cadence of
    // the matrix pipe; with ANY instruction of the same wavefront in between it waits for the accumulator's write-back
    // first (MI355X_MICROARCH.md: +43 cycles for the first extra issue slot, a cliff) - threading one epilogue slice
    // behind EVERY MFMA of the chain (RUNK = 0, rounds 1-3) pays that on every MFMA.
    constexpr int NRUN = (NKS16 + RUNK - 1) / RUNK;
    auto load_run = [&](auto rr) __attribute__((always_inline)) {
      constexpr int k0 = decltype(rr)::value * RUNK, k1 = (k0 + RUNK < NKS16) ? k0 + RUNK : NKS16;
      static_range<k0, k1>([&](auto kk) __attribute__((always_inline)) {
        constexpr int ks = decltype(kk)::value;
        wh[ks] = A[(AS * ks) * 64];
        if (NPASS == 3) wl[ks] = A[(AS * ks + 1) * 64];
      });
    };
    load_run(std::integral_constant<int, 0>{});
    static_for<NRUN>([&](auto rr) __attribute__((always_inline)) {
      constexpr int r = decltype(rr)::value;
      constexpr int k0 = r * RUNK, k1 = (k0 + RUNK < NKS16) ? k0 + RUNK : NKS16;
      static_range<k0, k1>([&](auto kk) __attribute__((always_inline)) { pre(kk); });
      __builtin_amdgcn_s_waitcnt(0xC07F);    // lgkmcnt(0) HERE: the run's A fragments have landed, no counted wait inside the run
      __builtin_amdgcn_sched_barrier(0);
      static_range<k0, k1>([&](auto kk) __attribute__((always_inline)) {
        constexpr int ks = decltype(kk)::value;
        constexpr int b = ks < 2 * FULL ? (ks >> 1) : FULL;
        constexpr int sb = ks < 2 * FULL ? (ks & 1) : 0;
        acc = mfma16<F16>(wh[ks], in[b].hi[sb], acc);
        if constexpr (NM >= 2) acc = mfma16<F16>(wh[ks], in[b].lo[sb], acc);
        if constexpr (NM >= 3) acc = mfma16<F16>(wl[ks], in[b].hi[sb], acc);
      });
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (r + 1 < NRUN) load_run(std::integral_constant<int, r + 1>{});
      __builtin_amdgcn_sched_barrier(0);
      static_range<unit_begin(k0 * NM, NS, NU), unit_begin(k1 * NM, NS, NU)>(epi);
      static_range<k0, k1>([&](auto kk) __attribute__((always_inline)) {
        static_for<NM>([&](auto mm) __attribute__((always_inline)) { slot_hook(kk, mm); });
      });
      __builtin_amdgcn_sched_barrier(0);
    });
    return acc;
  }
#pragma unroll
  for (int ks = 0; ks < PF && ks < NKS16; ++ks) {
    if ((NPHM_ABLATE & 128) && ks > 0) { wh[ks] = wh[0]; if (NPASS == 3) wl[ks] = wl[0]; continue; }
    if (NPHM_ABLATE & 512) {
      u32x4 fake = {unsigned(lane), 0x3c003c00u, unsigned(lane), 0x3c003c00u};
      asm volatile("" : "+v"(fake));
      wh[ks] = fake; if (NPASS == 3) wl[ks] = fake;
      continue;
    }
    wh[ks] = A[(AS * ks) * 64];
    if (NPASS == 3) wl[ks] = A[(AS * ks + 1) * 64];
  }
  if constexpr (HEAD > 0) {
    __builtin_amdgcn_sched_barrier(0);
    static_range<0, HEAD>(epi);
    __builtin_amdgcn_sched_barrier(0);
  }
  static_for<NKS16>([&](auto kk) __attribute__((always_inline)) {
    constexpr int ks = decltype(kk)::value;
    if constexpr (ks + PF < NKS16) {
      if constexpr ((NPHM_ABLATE & 128) != 0) {
        wh[ks + PF] = wh[0];
        if (NPASS == 3) wl[ks + PF] = wl[0];
      } else {
        wh[ks + PF] = A[(AS * (ks + PF)) * 64];
        if (NPASS == 3) wl[ks + PF] = A[(AS * (ks + PF) + 1) * 64];
      }
    }
    pre(kk);                               // one piece of the weight prefetch (LDS-DMA issue) per K-step
    __builtin_amdgcn_sched_barrier(0);     // the reads above are issued HERE, ahead of the MFMAs
    constexpr int b = ks < 2 * FULL ? (ks >> 1) : FULL;
    constexpr int sb = ks < 2 * FULL ? (ks & 1) : 0;
    static_for<NM>([&](auto mm) __attribute__((always_inline)) {
      constexpr int m = decltype(mm)::value;
      if constexpr (m == 0) acc = mfma16<F16>(wh[ks], in[b].hi[sb], acc);
      else if constexpr (m == 1) acc = mfma16<F16>(wh[ks], in[b].lo[sb], acc);
      else acc = mfma16<F16>(wl[ks], in[b].hi[sb], acc);
      constexpr int slot = ks * NM + m;
      static_range<HEAD + unit_begin(slot, NS, NU - HEAD), HEAD + unit_begin(slot + 1, NS, NU - HEAD)>(epi);
      slot_hook(kk, mm);                   // work with its own placement (the L0 epilogues inside lin1's first chunk)
      __builtin_amdgcn_sched_barrier(0);
    });
  });
  return acc;
}

// fp32 MFMA flavour: one slot per group of 4 K-steps (4 x v_mfma_f32_32x32x2f32 = 256 cycles of matrix pipe)
template <int NKS, int FULL, int NIN, int NU, class Epi, class Pre>
__device__ __forceinline__ f32x16 gemm_fused_f32(const char* afrag, f32x16 acc, const f32x16 (&in)[NIN],
                                                 int lane, Epi&& epi, Pre&& pre) {
  const f32x4* A = reinterpret_cast<const f32x4*>(afrag) + lane;
  constexpr int NS = NKS / 4;
  static_for<NS>([&](auto gg) __attribute__((always_inline)) {
    constexpr int g = decltype(gg)::value;
    const f32x4 a = A[g * 64];
    pre(gg);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int ks = 4 * g + c;
      const int b = ks < 16 * FULL ? (ks >> 4) : FULL;
      const int r = ks < 16 * FULL ? (ks & 15) : ks - 16 * FULL;
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c], in[b][r], acc, 0, 0, 0);
    }
    static_range<unit_begin(g, NS, NU), unit_begin(g + 1, NS, NU)>(epi);
    __builtin_amdgcn_sched_barrier(0);
  });
  return acc;
}

// ---- query point of a lattice slab ---------------------------------------------------------------
struct GridPoint {
  bool valid, hack;
  int64_t out_idx;
  float qx, qy, qz;
};
// lx = position inside the slab / the plane list; (iy, iz) lattice indices
__device__ __forceinline__ GridPoint grid_point(const EvalArgs& p, int lx, int iy, int iz, bool in_tile) {
  GridPoint g;
  const int nlx = p.ix1 - p.ix0;
  g.valid = in_tile && lx < nlx && iy < p.ry && iz < p.rz;
  const int clx = min(lx, nlx - 1), cy_ = min(iy, p.ry - 1), cz_ = min(iz, p.rz - 1);
  const int cx_ = p.xplanes ? p.xplanes[clx] : p.ix0 + clx;
  const int64_t gi = (int64_t(cx_) * p.ry + cy_) * p.rz + cz_;
  g.out_idx = (int64_t(clx) * p.ry + cy_) * p.rz + cz_;
  if (p.xyz) {
    // lattice-ORDERED but displaced queries (canonical points x + F_ex(x) of the two-stage
    // evaluation): same traversal, coordinates from the slab-local point array
    const float* q = p.xyz + g.out_idx * 3;
    g.qx = q[0]; g.qy = q[1]; g.qz = q[2];
  } else {
    g.qx = p.ax[cx_]; g.qy = p.ay[cy_]; g.qz = p.az[cz_];
  }
  g.hack = false;
  if (p.hack_chunk > 0)
    g.hack = ((gi + 1) % p.hack_chunk == 0) || (gi == int64_t(p.rx) * p.ry * p.rz - 1);
  return g;
}

// Gaussian blend weight of an anchor at offset (dx, dy, dz): exp(-(|d| + 1e-5)^2 / 0.01)
// (EnsembledDeepSDF.py:129-133).  Raw v_sqrt_f32 (1 ulp), a multiplication by 100 (what PyTorch's
// division by the scalar 0.01 is on a GPU) and exp(t) = 2^hi (1 + lo ln2) with t log2(e) = hi + lo carried in
// two floats, instead of the IEEE sqrt / divide / expf expansions of hipcc: 12 instead of 45 VALU
// operations, to the same ulp - 39 of them per lattice point in the binning pre-pass.
__device__ __forceinline__ float blend_weight(float dx, float dy, float dz, float* dist = nullptr) {
  // no implicit fma contraction here and in blend_masks: the binning pre-pass and the in-kernel form of the rule must
  // produce the same bits (the binned traversal is bitwise the brick traversal), whatever surrounds the inlined code
#pragma clang fp contract(off)
  const float d = __builtin_amdgcn_sqrtf(dx * dx + dy * dy + dz * dz) + 1e-5f;
  if (dist) *dist = d;
  const float t = -(d * d) * 100.0f;
  const float hi = t * 1.44269504f;                                          // log2(e) = 1.44269504 + 1.925963e-8
  const float lo = fmaf(t, 1.44269504f, -hi) + t * 1.925963033e-8f;
  return __builtin_amdgcn_exp2f(hi) * fmaf(lo, 0.693147181f, 1.0f);
}

// ---- blend normaliser and active-member masks (EnsembledDeepSDF.py:129-150) ------------------------
// Per lane: S = sum of the 40 blend weights, denom = S + 1e-6.  Per GROUP of lanes (the whole
// wavefront, or with SPLIT its two 32-lane halves = two tiles of the binning pre-pass):
//   wmask: members the group's wavefront evaluates, hmask: ... of which with the 3-pass product.
// Pruning rule, per point: drop the smallest-weight members as long as their normalised weights sum
// to <= 40 * prune_tol (|error| <= 40 * prune_tol * max|f_k|, the bound of dropping every member
// below prune_tol - but it is spent where it buys most: 6.2 instead of 7.6 members per wavefront).
// The cut is the largest of 6 candidate thresholds whose cumulated weight stays within the budget.
template <bool SPLIT>
__device__ __forceinline__ void blend_masks(const float* anch, float qx, float qy, float qz, bool valid, bool hack,
                                            float prune_tol, float light_tol, float mid_tol, float& S_out, float& denom_out,
                                            uint64_t (&wmask)[2], uint64_t (&hmask)[2], uint64_t (&fmask)[2]) {
#pragma clang fp contract(off)
  // the 39 anchor weights of this lane's point are computed ONCE and kept in registers: the three
  // passes below index them statically (unrolled).  The rules act on wv[k] = w_k B_k(d_k), the size member k's term
  // can have (layout.h: LS_OFF_BND; B = 1 unless bounds were installed); the blend sum S on the bare weights.
  const float* bnd = anch + (LS_OFF_BND - LS_OFF_ANCH);
  float wv[N_LOC];
  float S = 0.f;
#pragma unroll
  for (int k = 0; k < N_LOC; ++k) {
    const float dx = anch[3 * k] - qx, dy = anch[3 * k + 1] - qy, dz = anch[3 * k + 2] - qz;
    float d;
    const float w = blend_weight(dx, dy, dz, &d);
    S += w;
    wv[k] = w * fmaf(fmaf(bnd[4 * k + 2], d, bnd[4 * k + 1]), d, bnd[4 * k]);
  }
  const float w_bg_raw = expf(-0.2f / 0.01f);
  S += w_bg_raw;
  const float w_bg = w_bg_raw * bnd[4 * N_LOC];
  const float denom = S + 1e-6f;
  S_out = S;
  denom_out = denom;
  auto any = [&](bool c, uint64_t bit, uint64_t (&m)[2]) __attribute__((always_inline)) {
    const uint64_t b = __ballot(c);
    if (SPLIT) {
      if (b & 0xffffffffull) m[0] |= bit;
      if (b >> 32) m[1] |= bit;
    } else if (b) {
      m[0] |= bit;
    }
  };
  float thr = prune_tol * denom;
  wmask[0] = wmask[1] = 0;                 // members this wavefront evaluates (wave-uniform)
  hmask[0] = hmask[1] = ~0ull;             // ... of which with more than the single-pass product
  fmask[0] = fmask[1] = ~0ull;             // ... of which with the full three-pass product ("heavy")
  if (prune_tol < 0.f) {
    const uint64_t b = __ballot(valid);
    const uint64_t all = (1ull << N_MEMBERS) - 1;
    if (SPLIT) { wmask[0] = (b & 0xffffffffull) ? all : 0ull; wmask[1] = (b >> 32) ? all : 0ull; }
    else wmask[0] = b ? all : 0ull;
  } else {
    hmask[0] = hmask[1] = 0;
    fmask[0] = fmask[1] = 0;
    // candidates 1, 2, 4, 8, 16, 40 x thr; below(c) = weight of the members not above c is monotone in c,
    // so the largest fitting candidate is found by bisection: 3 passes over the 40 weights instead of 6
    auto below = [&](float c) __attribute__((always_inline)) {
      float sum = w_bg <= c ? w_bg : 0.f;
#pragma unroll
      for (int k = 0; k < N_LOC; ++k) sum += wv[k] <= c ? wv[k] : 0.f;
      return sum;
    };
    const float budget = float(N_MEMBERS) * thr;      // the first candidate always fits: <= 40 members below prune_tol
    const bool f8 = below(8.f * thr) <= budget;
    const bool fb = below(f8 ? 40.f * thr : 4.f * thr) <= budget;
    const bool fc = below(f8 ? 16.f * thr : 2.f * thr) <= budget;
    thr = f8 ? (fb ? 40.f * thr : fc ? 16.f * thr : 8.f * thr) : (fb ? 4.f * thr : fc ? 2.f * thr : thr);
    const bool live = valid && !hack;
#pragma unroll
    for (int k = 0; k < N_LOC; ++k) any(live && wv[k] > thr, 1ull << k, wmask);
    any(live && w_bg > thr, 1ull << N_LOC, wmask);
    // members that weigh >= light_tol somewhere in the group keep the 3-pass product
    const float heavy_thr = light_tol * denom;
#pragma unroll
    for (int k = 0; k < N_LOC; ++k) any(live && wv[k] >= heavy_thr, 1ull << k, hmask);
    any(live && w_bg >= heavy_thr, 1ull << N_LOC, hmask);
    // ... and the three-pass product only if they weigh >= mid_tol somewhere (mid_tol <= light_tol: no two-pass tier)
    const float full_thr = mid_tol * denom;
#pragma unroll
    for (int k = 0; k < N_LOC; ++k) any(live && wv[k] >= full_thr, 1ull << k, fmask);
    any(live && w_bg >= full_thr, 1ull << N_LOC, fmask);
  }
}

// ---- binning pre-pass ---------------------------------------------------------------------------------
// Tiles = the 4x4x2-voxel blocks a wavefront of eval_kernel works on (tile t = ((tlx * nty) + ty) * ntz + tz).
// One wavefront handles two tiles (its 32-lane halves): per point (S, denom), per tile the member
// masks and a sort key = (wmask, hash(hmask)) - equal masks end up adjacent after the radix sort, so
// the 8 wavefronts of a workgroup stream (nearly) the same members: no idle passes for members that
// only a neighbouring tile needs, and the workgroups running together on an XCD share their weights.
// Workgroup b runs on XCD b % 8.  The sorted tile list is cut into runs of XCD_RUN workgroups (equal or
// neighbouring member sets) dealt to the XCDs in turn: an XCD's consecutive workgroups stay inside a run,
// so its L2 keeps serving the same few members' weights instead of following, with all 8 XCDs, every
// change of member set along the list (brick order: 5.4 GB of L2 fills per 256^3 launch, binned without
// runs: 18.7 GB).  Grids are padded to whole rounds of 8 runs.
#ifndef NPHM_XCD_RUN
#define NPHM_XCD_RUN 32   // round 3 (calibrated f16 default, ~5 members per wavefront): 16 / 32 are 2-3 % ahead of 64 on two boxes, 128 -4 %
#endif
constexpr unsigned XCD_RUN = NPHM_XCD_RUN;
__device__ __forceinline__ unsigned binned_group(unsigned b) {
  if (XCD_RUN <= 1) return b;
  const unsigned xcd = b & 7u, i = b >> 3;                  // i-th workgroup of its XCD
  return ((i / XCD_RUN) * 8u + xcd) * XCD_RUN + i % XCD_RUN;
}

__device__ __forceinline__ void tile_lane(const EvalArgs& p, unsigned t, int j, int& lx, int& iy, int& iz) {
  const int tz = int(t % unsigned(p.ntz));
  const unsigned r = t / unsigned(p.ntz);
  const int ty = int(r % unsigned(p.nty)), tlx = int(r / unsigned(p.nty));
  lx = tlx * TLX + tile_dx(j);
  iy = ty * TLY + tile_dy(j);
  iz = tz * TLZ + tile_dz(j);
}

static __global__ __launch_bounds__(256) void tile_prepass_kernel(EvalArgs p) {
  const int lane = threadIdx.x & 63, half = lane >> 5, j = lane & 31;
  const unsigned wave_id = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const unsigned t = wave_id * 2 + half;
  const bool in_tile = t < unsigned(p.n_tiles);
  int lx, iy, iz;
  tile_lane(p, in_tile ? t : 0u, j, lx, iy, iz);
  const GridPoint g = grid_point(p, lx, iy, iz, in_tile);
  float S, denom;
  uint64_t wmask[2], hmask[2], fmask[2];
  blend_masks<true>(p.state + LS_OFF_ANCH, g.qx, g.qy, g.qz, g.valid, g.hack, p.prune_tol, p.light_tol, p.mid_tol, S, denom, wmask, hmask, fmask);
  if (!in_tile) return;
  p.tile_sd[size_t(t) * 32 + j] = make_float2(S, denom);
  if (j == 0) {
    const uint64_t w = wmask[half], h = (hmask[half] & w) ^ ((fmask[half] & w) * 0x9E3779B97F4A7C15ull);
    p.tile_masks[3 * size_t(t)] = w;
    p.tile_masks[3 * size_t(t) + 1] = hmask[half];
    p.tile_masks[3 * size_t(t) + 2] = fmask[half];
    // 40 mask bits + 24 bits that tell different heavy sets of one mask apart
#ifndef NPHM_KEY_TIERS
#define NPHM_KEY_TIERS 0
#endif
#if NPHM_KEY_TIERS
    // ... ordered by how many members run three / more than one pass, so that tiles of one member set AND a similar pass
    // structure share a workgroup (its wavefronts advance in lock step through the weight stream)
    const uint64_t hh = (uint64_t(__popcll(fmask[half] & w)) << 18) | (uint64_t(__popcll(hmask[half] & w)) << 12) |
                        ((h * 0x9E3779B97F4A7C15ull) >> 52);
#else
    const uint64_t hh = (h * 0x9E3779B97F4A7C15ull) >> 40;
#endif
    p.tile_keys[t] = (w << 24) | hh;
    p.tile_ids[t] = t;
  }
}

template <int MODE, int PREC>
__global__ __launch_bounds__(64 * NW, 2) void eval_kernel(EvalArgs p) {
  using WS = Streamer<PREC>;
  __shared__ __attribute__((aligned(16))) char ring[RING * WS::SLOT_BYTES];
  __shared__ unsigned int wg_mask[4];        // [0..1] members some wavefront evaluates, [2..3] ... runs three-term
  __shared__ unsigned char wg_list[N_MEMBERS];
  // per-lane query point and blend normaliser, parked in LDS across the member loop: the loop body holds 7 + 4 activation
  // blocks, 2-3 accumulators and the A fragments in 256 VGPRs - every VGPR that merely LIVES across it ends up as a scratch
  // spill at kernel entry (round 3: 0.32 GB of private-segment writes per 256^3 launch, with the stacked tail blocks 1.3 GB).
  // A member reads its four floats back with one ds_read_b128.
  __shared__ f32x4 lane_q[64 * NW];
  __shared__ float lane_S[64 * NW];
  // MODE 0 / 1: (output index | valid << 62 | hack << 63) of the lane's point, parked the same way (MODE 2 recomputes them from
  // the tile id): as VGPRs that live across the member loop they were 3 spilled registers of the brick / point-set variants
  __shared__ unsigned long long lane_w[MODE == 2 ? 1 : 64 * NW];
#if NPHM_PROF
  // per-tier phase profile (timing builds): cycles of every (wavefront, member) visit by the wavefront's own tier t and the
  // heaviest tier T any wavefront of the workgroup runs that member at (0 = not needed, 1 = single-term, 2 = two-term,
  // 3 = three-term): [t][T][visits, total, vmcnt wait, barrier wait, GEMM + epilogue] -> stats[16 + (4 t + T) 5 + i]
  __shared__ unsigned long long tier_prof[4 * 4 * 5];
  __shared__ unsigned int wg_tier[N_MEMBERS];
  __shared__ long long tier_t0[NW];      // start stamp of the current member visit, per wavefront (not kept in registers)
  __shared__ unsigned char my_tier[NW][N_MEMBERS];
#endif

  const int lane_inv = threadIdx.x & 63;
  const int lane = lane_inv;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int h_inv = lane >> 5;
  const int h = h_inv;
  const int j = lane & 31;
  auto my_slot = [&]() __attribute__((always_inline)) -> int {      // this lane's LDS slot, from the execution mask: no live VGPR
    return wave * 64 + int(__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)));
  };

  // ---- locate this lane's query point ------------------------------------------------------
  bool valid;
  int64_t out_idx;
  bool hack = false;
  float qx, qy, qz;
  int row = 0;
  unsigned tile = 0;
  if (MODE == 0) {
    row = blockIdx.y;
    const int64_t i = (int64_t(blockIdx.x) * NW + wave) * 32 + j;
    valid = i < p.n_points;
    const int64_t ic = valid ? i : (p.n_points - 1);
    const float* q = p.xyz + (int64_t(row) * p.n_points + ic) * 3;
    qx = q[0]; qy = q[1]; qz = q[2];
    out_idx = int64_t(row) * p.n_points + ic;
    if (p.hack_chunk > 0) hack = ((ic + 1) % p.hack_chunk == 0) || (ic == p.n_points - 1);
  } else if (MODE == 1) {
    // The hardware places block b on XCD b % 8: XCD x works on super-bricks x, x+8, x+16, ...
    // (interleaved, so every XCD sees the same mix of near-surface and empty space) and its
    // consecutive blocks are the bricks of ONE super-brick.
    constexpr int SBN = SBX * SBY * SBZ;
    const int local = blockIdx.x >> 3;
    const int inner = local % SBN;
    int bid = (local / SBN) * 8 + (blockIdx.x & 7);          // super-brick index
    const int sz = bid % p.nsz; bid /= p.nsz;
    const int sy = bid % p.nsy; bid /= p.nsy;
    const int bx = bid * SBX + inner / (SBY * SBZ);
    const int by = sy * SBY + (inner / SBZ) % SBY;
    const int bz = sz * SBZ + inner % SBZ;
    // the wavefronts' tiles fill the brick x-fastest
    constexpr int WTX = BRX / TLX, WTY = BRY / TLY;
    static_assert(BRX % TLX == 0 && BRY % TLY == 0 && BRZ % TLZ == 0 && WTX * WTY * (BRZ / TLZ) == NW, "brick = NW tiles");
    const int wx = wave % WTX, wy = (wave / WTX) % WTY, wz = wave / (WTX * WTY);
    const GridPoint g = grid_point(p, bx * BRX + wx * TLX + tile_dx(j), by * BRY + wy * TLY + tile_dy(j),
                                   bz * BRZ + wz * TLZ + tile_dz(j), true);
    valid = g.valid; hack = g.hack; out_idx = g.out_idx; qx = g.qx; qy = g.qy; qz = g.qz;
  } else {
    // binned tiles: slot -> tile through the sorted order of tile_prepass_kernel
    const unsigned slot = binned_group(blockIdx.x) * NW + wave;
    const bool in_tile = slot < unsigned(p.n_tiles);
    tile = in_tile ? p.tile_order[in_tile ? slot : 0u] : 0u;
    tile = __builtin_amdgcn_readfirstlane(tile);
    int lx, iy, iz;
    tile_lane(p, tile, j, lx, iy, iz);
    const GridPoint g = grid_point(p, lx, iy, iz, in_tile);
    valid = g.valid; hack = g.hack; out_idx = g.out_idx; qx = g.qx; qy = g.qy; qz = g.qz;
  }

  const float* st = p.state + size_t(row) * LS_ROW_STRIDE;
  const float* anch = st + LS_OFF_ANCH;

  // ---- blend normaliser and active-member mask: computed here, or read from the binning pre-pass ----
  float S, denom;
  uint64_t wmask, hmask, fmask;
  const float w_bg = expf(-0.2f / 0.01f);
  if (MODE == 2) {
    const float2 sd = p.tile_sd[size_t(tile) * 32 + j];
    S = sd.x; denom = sd.y;
    const bool in_tile = binned_group(blockIdx.x) * NW + wave < unsigned(p.n_tiles);
    // (wave-uniform values: into SGPRs - as VGPR pairs they live across the member loop and get spilled to scratch)
    auto uni64 = [](uint64_t v) __attribute__((always_inline)) {
      return (uint64_t(uint32_t(__builtin_amdgcn_readfirstlane(int(uint32_t(v >> 32))))) << 32) |
             uint32_t(__builtin_amdgcn_readfirstlane(int(uint32_t(v))));
    };
    wmask = uni64(in_tile ? p.tile_masks[3 * size_t(tile)] : 0ull);
    hmask = uni64(in_tile ? p.tile_masks[3 * size_t(tile) + 1] : 0ull);
    fmask = uni64(in_tile ? p.tile_masks[3 * size_t(tile) + 2] : 0ull);
  } else {
    uint64_t wm[2], hm[2], fm[2];
    blend_masks<false>(anch, qx, qy, qz, valid, hack, p.prune_tol, p.light_tol, p.mid_tol, S, denom, wm, hm, fm);
    wmask = wm[0]; hmask = hm[0]; fmask = fm[0];
  }

#if NPHM_LDS_STASH
  {
    f32x4 q4 = {qx, qy, qz, denom};
    lane_q[threadIdx.x] = q4;
    lane_S[threadIdx.x] = S;
    if (MODE != 2) lane_w[threadIdx.x] = (unsigned long long)out_idx | ((unsigned long long)valid << 62) | ((unsigned long long)hack << 63);
  }
#endif
  const unsigned long long nv = __popcll(__ballot(valid)) >> 1;   // both half-waves hold the same points
  // counters: summed over the workgroup in LDS, then ONE global atomic per counter and workgroup.  (Rounds 1-5: four global
  // atomics per wavefront, 2.1 M per 256^3 launch on four addresses of one cache line - alone they take 12.5 ms, beside the
  // member loop 3.3 % of a launch in the calibrated mode and 12 % with every member single-term: tools/identity_variants.py.)
  __shared__ unsigned long long wg_stats[4];
  if (p.stats && threadIdx.x < 4) wg_stats[threadIdx.x] = 0ull;

  float acc = 0.f;
  // Two passes at most.  Pass 0 is the evaluation proper.  Pass 1 (refine_band > 0, rare) is the sign-safe refinement:
  // where the blended value of pass 0 lands within refine_band of zero, an error of the fast setting could flip its sign
  // - and with it the topology of the extracted mesh.  Such a wavefront re-evaluates its tile with every member on the
  // full three-pass product under a tighter pruning budget; the workgroup streams those members once more (the other
  // wavefronts idle through the pass).  ~1e-5 of the tiles of a 256^3 extraction.
#pragma unroll 1
  for (int pass = 0; pass < 2; ++pass) {
  // ---- union over the workgroup: the members whose weights get streamed ----------------------
  if (threadIdx.x < 4) wg_mask[threadIdx.x] = 0u;
  __syncthreads();
  if (lane == 0) {
    atomicOr(&wg_mask[0], (unsigned int)(wmask & 0xffffffffull));
    atomicOr(&wg_mask[1], (unsigned int)(wmask >> 32));
    if (NPHM_HI_ONLY && PREC >= 1) {                     // members this wavefront runs three-term: their wl fragments are needed
      const uint64_t three = wmask & hmask & fmask;
      atomicOr(&wg_mask[2], (unsigned int)(three & 0xffffffffull));
      atomicOr(&wg_mask[3], (unsigned int)(three >> 32));
    }
    if (p.stats && pass == 0) {
      atomicAdd(&wg_stats[0], nv * __popcll(wmask));
      atomicAdd(&wg_stats[1], nv);
      atomicAdd(&wg_stats[2], nv * __popcll(wmask & hmask & ~fmask));   // two-pass pairs
      atomicAdd(&wg_stats[3], nv * __popcll(wmask & ~hmask));           // single-pass ("light") pairs
    }
  }
  __syncthreads();
  if (p.stats && pass == 0 && threadIdx.x < 4) {
    const unsigned long long v = wg_stats[threadIdx.x];
    if (v) atomicAdd(p.stats + (threadIdx.x < 2 ? threadIdx.x : 12 + threadIdx.x), v);   // slots 0, 1, 14, 15
  }
  const uint64_t gmask = (uint64_t(wg_mask[1]) << 32) | wg_mask[0];
  const int n_active = (NPHM_ABLATE & 1024) ? 0 : __popcll(gmask);
  if (threadIdx.x < N_MEMBERS) {
    if ((gmask >> threadIdx.x) & 1ull)
      wg_list[__popcll(gmask & ((1ull << threadIdx.x) - 1))] = (unsigned char)threadIdx.x;
  }
  __syncthreads();
#if NPHM_PROF
  auto tier_of = [&](int k) __attribute__((always_inline)) -> int {
    if (!((wmask >> k) & 1ull)) return 0;
    if (PREC == 0) return 3;
    if (!((hmask >> k) & 1ull)) return 1;
    return ((fmask >> k) & 1ull) ? 3 : 2;
  };
  if (threadIdx.x < 80) tier_prof[threadIdx.x] = 0ull;
  if (threadIdx.x < N_MEMBERS) wg_tier[threadIdx.x] = 0u;
  __syncthreads();
  if (lane == 0) for (int i = 0; i < n_active; ++i) {
    const int t = tier_of(wg_list[i]);
    my_tier[wave][i] = (unsigned char)t;
    atomicMax(&wg_tier[i], (unsigned)t);
  }
  __syncthreads();
#endif
#if NPHM_SETPRIO
  if (wave >= NW / 2) __builtin_amdgcn_s_setprio(1);    // the younger wavefront of each SIMD loses every issue arbitration otherwise
#endif
  WS ws{p, st, ring, wg_list, n_active, 0, wave, lane, WS::make_rsrc(Stream<PREC>::set_base(p, 0)), WS::make_rsrc(st)};
  if (NPHM_HI_ONLY && PREC >= 1) {
    ws.lo_mask = (uint64_t(uint32_t(__builtin_amdgcn_readfirstlane(int(wg_mask[3])))) << 32) | uint32_t(__builtin_amdgcn_readfirstlane(int(wg_mask[2])));
  }
  ws.load_ids();
  ws.issue(false, 0);
  ws.issue(false, 1);

#if NPHM_PROF == 1
  long long prof[6] = {0, 0, 0, 0, 0, 0};   // L0 gemm, sync, gemm, epilogue, member total, kernel total
  const long long t_kernel = clock64();
#endif

#pragma unroll 1
  for (int mi = 0; mi < n_active; ++mi) {
    const int k = wg_list[mi];
#if NPHM_PROF
    if (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == 0) tier_t0[wave] = clock64();
#if NPHM_PROF == 1
    const long long tp_v0 = ws.sprof[0], tp_b0 = ws.sprof[1], tp_g0 = prof[0] + prof[2];
#endif
    auto tier_record = [&]() __attribute__((always_inline)) {
      const long long t1 = clock64();
      if (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == 0) {
        unsigned long long* q = tier_prof + (4 * int(my_tier[wave][mi]) + int(wg_tier[mi])) * 5;
        atomicAdd(q, 1ull);
        atomicAdd(q + 1, (unsigned long long)(t1 - tier_t0[wave]));
#if NPHM_PROF == 1
        atomicAdd(q + 2, (unsigned long long)(ws.sprof[0] - tp_v0));
        atomicAdd(q + 3, (unsigned long long)(ws.sprof[1] - tp_b0));
        atomicAdd(q + 4, (unsigned long long)(prof[0] + prof[2] - tp_g0));
#endif
      }
    };
#endif
    if (!((wmask >> k) & 1ull)) {
      // this wavefront's 32 points do not need member k: keep the ring moving only
      static_for<CHUNKS_PER_MEMBER>([&](auto cc) __attribute__((always_inline)) {
        ws.template sync<decltype(cc)::value>();
        if constexpr (NPHM_DMA_INSTREAM) ws.template prefetch<decltype(cc)::value>();
      });
      ws.next_member();
#if NPHM_PROF
      tier_record();
#endif
      continue;
    }
    // lane-derived values are re-materialised per member (opaque to LICM): hoisting the dozens of
    // per-site lane offsets out of this loop costs more registers than recomputing them
#if NPHM_LDS_STASH
    // lane index from the execution mask (2 VALU) instead of a register kept alive across the loop
    int lane = int(__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)));
    asm volatile("" : "+v"(lane));
    int h = lane >> 5;
    const f32x4 q4 = lane_q[wave * 64 + lane];
    asm volatile("" : "+v"(lane));          // (re-materialised per site below, not kept: see the DMA code)
    const float qx = q4[0], qy = q4[1], qz = q4[2], denom = q4[3];
#else
    int h = h_inv, lane = lane_inv;
    asm volatile("" : "+v"(h), "+v"(lane));
#endif
    PROF_T(t_m0);
    // adaptive precision (bf16 path): single-pass products for a member that weighs < light_tol at
    // every point of this wavefront (its error enters the blend scaled by that weight)
    const bool light = PREC >= 1 && !((hmask >> k) & 1ull);
    const bool mid = PREC >= 1 && !light && !((fmask >> k) & 1ull);     // two-pass tier

    // local coordinates (EnsembledDeepSDF.py:240-244): anchor-relative, odd member of a
    // symmetric pair mirrored in x, background member uses global coordinates
    float cx = qx, cy = qy, cz = qz;
    float wk = w_bg;
    if (k < N_LOC) {
      const float ax = anch[3 * k], ay = anch[3 * k + 1], az = anch[3 * k + 2];
      cx = qx - ax; cy = qy - ay; cz = qz - az;
      const float dx = ax - qx, dy = ay - qy, dz = az - qz;
      wk = blend_weight(dx, dy, dz);
    }
    if (k < 2 * N_SYMM && (k & 1)) cx = -cx;
    const float b4 = p.packed_f32[size_t(member_set(k)) * SET_STRIDE + OFF_L4B];
    float part = 0.f;

    // Activations of this wavefront's 32 points (registers): fp32 blocks or split-bf16 operands
    using Act = typename std::conditional<PREC == 0, f32x16, ActB>::type;
    Act H[7], G[4];
    f32x16 accs[2];    // accumulators of two consecutive chunks: GEMM(c) fills one while the epilogue of c-1 drains the other
    f32x16 acc_x;      // bf16 path: third accumulator while the L0 blocks are drained inside lin1's first chunk
    // accumulator of L0 block B: fp32 path accs[B & 1]; bf16 path accs[0] / acc_x (accs[1] belongs to chunk 1)
    auto l0_acc = [&](auto BB) __attribute__((always_inline)) -> f32x16& {
      constexpr int B = decltype(BB)::value;
      if constexpr (PREC == 0) return accs[B & 1];
      else if constexpr (B & 1) return acc_x;
      else return accs[0];
    };
    const float coords[3] = {cx, cy, cz};

    // ---- 19 chunks: 0 = L0 (3 coords -> 200), 1..4 = L1, 5..11 = L2, 12..18 = L3.  The NW wavefronts of
    // the workgroup meet at ONE barrier per chunk (weight ring).  Between two barriers a wavefront runs
    // GEMM(c) on the matrix pipe with the VALU epilogue of chunk c-1 threaded through it (gemm_fused_*):
    // the epilogue of the last block of a layer is short (8 real features) and its block is consumed by
    // the LAST K-step of the next layer's chunks, so the fusion also runs across layer boundaries.

    // unit r (one accumulator register) of the epilogue of chunk P: softplus, then into the operand
    // layout of the next layer (P <= 11) or, for lin3's blocks, straight into lin4's dot product
    f32x4 w4q = {};
    auto epi_unit = [&](auto PP, auto LL, auto uu) __attribute__((always_inline)) {
      constexpr int P = decltype(PP)::value, r = decltype(uu)::value;
      constexpr bool LIGHT = decltype(LL)::value == 1;
      f32x16& a = accs[P & 1];
      constexpr int g = P - 1;
      constexpr bool last_block = P == 0 || g == L1_OB - 1 || g == L1_OB + L2_OB - 1 || P == CHUNKS_PER_MEMBER - 1;
      (void)last_block;
      // stacked tail block of a three-term member: register r + 4 of the same lane holds the wl rows of register r's features
      if constexpr (PREC == 2 && NPHM_STACK_TAILS && is_tail_chunk(P) && decltype(LL)::value == 0 && r < LAST_BLOCK_REGS) a[r] += a[r + 4];
      // packed-f16 epilogue of a single-pass member (light_pair): everything happens on the odd register of a pair.  (Not for
      // lin1's tail block: three of its four registers are the skip connection's coordinates in the upper half-wave.)
      constexpr bool PK = LIGHT && PREC == 2 && NPHM_LIGHT_PK && NPHM_LIGHT_POLY && g != L1_OB - 1;
      if constexpr (PK) {
        if constexpr (P >= 1 + L1_OB + L2_OB) {
          if constexpr (r % 4 == 0) {
            typedef __attribute__((address_space(3))) const f32x4* lds_v4;
            const unsigned int w4a = WS::lds_addr(reinterpret_cast<const char*>(WS::tail_of(ws.slot(P)) + 32 + h * 16 + r));
            w4q = *(lds_v4)(size_t)w4a;
          }
          if constexpr (r & 1) {
            // lin4 on the pair: v_dot2c_f32_f16 (fp32 accumulate) with the two weights converted alongside
            const f16x2 wpk = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(w4q[(r - 1) % 4], w4q[r % 4]));
            part = __builtin_amdgcn_fdot2(light_pair(a[r - 1], a[r]), wpk, part, false);
            asm volatile("" : "+v"(part));
          }
        } else {
          Act& dst = g < L1_OB ? G[g < L1_OB ? g : 0] : H[g >= L1_OB ? g - L1_OB : 0];
          constexpr int NR = (g == L1_OB + L2_OB - 1) ? LAST_BLOCK_REGS : 16;
          if constexpr (r & 1) light_pack_pair<r - 1>(a, dst);
          if constexpr (r == NR - 1 && NR < 16) zero_pad_hi<NR>(dst);
        }
        return;
      }
      if constexpr (PREC >= 1) { if constexpr (NPHM_EPI_PAIRS && !LIGHT) {
        // pairs: everything for registers r - 1, r happens at the odd one
        if constexpr (P >= 1 + L1_OB + L2_OB) {
          if constexpr (r % 4 == 0) {
            typedef __attribute__((address_space(3))) const f32x4* lds_v4;
            const unsigned int w4a = WS::lds_addr(reinterpret_cast<const char*>(WS::tail_of(ws.slot(P)) + 32 + h * 16 + r));
            w4q = *(lds_v4)(size_t)w4a;
          }
          if constexpr (r & 1) {
            float x0, x1;
            if constexpr (NPHM_MID_POLY && decltype(LL)::value == 2) softplus2_pair_poly(a[r - 1], a[r], x0, x1);
            else softplus2_pair(a[r - 1], a[r], x0, x1);
            part = fmaf(x0, w4q[(r - 1) % 4], part);
            part = fmaf(x1, w4q[r % 4], part);
            asm volatile("" : "+v"(part));
          }
        } else if constexpr (r & 1) {
          float x0, x1;
          if constexpr (NPHM_MID_POLY && decltype(LL)::value == 2) softplus2_pair_poly(a[r - 1], a[r], x0, x1);
          else softplus2_pair(a[r - 1], a[r], x0, x1);
          if constexpr (g == L1_OB - 1) {        // skip connection (see below): registers 1..3 of the upper half-wave
            if constexpr (r - 1 >= 1 && r - 1 <= 3) x0 = h ? coords[r - 2] : x0;
            if constexpr (r <= 3) x1 = h ? coords[r - 1] : x1;
          }
          Act& dst = g < L1_OB ? G[g < L1_OB ? g : 0] : H[g >= L1_OB ? g - L1_OB : 0];
          constexpr int NR = (g == L1_OB - 1 || g == L1_OB + L2_OB - 1) ? LAST_BLOCK_REGS : 16;
          a[r - 1] = x0;
          a[r] = x1;
          pack_pair<r - 1, LIGHT, PREC == 2>(a, dst);
          if constexpr (r == NR - 1 && NR < 16) {
#pragma unroll
            for (int q = NR / 2; q < 4; ++q) { dst.hi[0][q] = 0u; dst.lo[0][q] = 0u; }
            asm volatile("" : "+v"(dst.hi[0]));
            asm volatile("" : "+v"(dst.lo[0]));
          }
        }
        return;
      } }
      float x = (LIGHT && NPHM_LIGHT_POLY) ? softplus2_light(a[r]) : softplus2(a[r]);
      if constexpr (P >= 1 + L1_OB + L2_OB) {
        // lin3 block: lin4 (200 -> 1) fused; its 16 weights sit in the chunk's ring-slot tail
        if constexpr (r % 4 == 0) {
          typedef __attribute__((address_space(3))) const f32x4* lds_v4;
          const unsigned int w4a = WS::lds_addr(reinterpret_cast<const char*>(WS::tail_of(ws.slot(P)) + 32 + h * 16 + r));
          w4q = *(lds_v4)(size_t)w4a;
        }
        part = fmaf(x, w4q[r % 4], part);
        asm volatile("" : "+v"(part));      // finish this unit here (see pack_pair)
      } else {
        static_assert(P >= 1, "the L0 blocks have their own epilogue");
        if constexpr (g == L1_OB - 1 && r >= 1 && r <= 3) {
          // skip connection: features 101..103 of lin2's input are the local coords (block 3,
          // regs 1..3 of the upper half-wave); 1/sqrt(2) and the activation scale live in the weights
          x = h ? coords[r - 1] : x;
        }
        Act& dst = g < L1_OB ? G[g < L1_OB ? g : 0] : H[g >= L1_OB ? g - L1_OB : 0];
        constexpr int NR = (g == L1_OB - 1 || g == L1_OB + L2_OB - 1) ? LAST_BLOCK_REGS : 16;
        if constexpr (PREC == 0) {
          dst[r] = x;
          asm volatile("" : "+v"(dst[r]));
          if constexpr (r == NR - 1) {
#pragma unroll
            for (int q = NR; q < 16; ++q) dst[q] = 0.f;
          }
        } else {
          a[r] = x;
          if constexpr (r & 1) pack_pair<r - 1, LIGHT, PREC == 2>(a, dst);
          if constexpr (r == NR - 1 && NR < 16) {
            // padding features of a layer's last block: zero operands (their weights are zero too,
            // but 0 x garbage could be NaN)
#pragma unroll
            for (int q = NR / 2; q < 4; ++q) { dst.hi[0][q] = 0u; dst.lo[0][q] = 0u; }
            asm volatile("" : "+v"(dst.hi[0]));
            asm volatile("" : "+v"(dst.lo[0]));
          }
        }
      }
    };
    // unit r of the epilogue of L0 block B (accumulator accs[B & 1]) -> H[B]
    auto epi_l0 = [&](auto BB, auto LL, auto uu) __attribute__((always_inline)) {
      constexpr int B = decltype(BB)::value, r = decltype(uu)::value;
      constexpr bool LIGHT = decltype(LL)::value == 1;
      constexpr int NR = B == 6 ? LAST_BLOCK_REGS : 16;
      f32x16& a = l0_acc(BB);
      if constexpr (LIGHT && PREC == 2 && NPHM_LIGHT_PK && NPHM_LIGHT_POLY) {      // packed-f16 epilogue (light_pair)
        if constexpr (r & 1) light_pack_pair<r - 1>(a, H[B]);
        if constexpr (r == NR - 1 && NR < 16) zero_pad_hi<NR>(H[B]);
        return;
      }
      if constexpr (PREC >= 1) { if constexpr (NPHM_EPI_PAIRS && !LIGHT) {
        if constexpr (r & 1) {
          float x0, x1;
          if constexpr (NPHM_MID_POLY && decltype(LL)::value == 2) softplus2_pair_poly(a[r - 1], a[r], x0, x1);
          else softplus2_pair(a[r - 1], a[r], x0, x1);
          a[r - 1] = x0;
          a[r] = x1;
          pack_pair<r - 1, LIGHT, PREC == 2>(a, H[B]);
          if constexpr (r == NR - 1 && NR < 16) {
#pragma unroll
            for (int q = NR / 2; q < 4; ++q) { H[B].hi[0][q] = 0u; H[B].lo[0][q] = 0u; }
            asm volatile("" : "+v"(H[B].hi[0]));
            asm volatile("" : "+v"(H[B].lo[0]));
          }
        }
        return;
      } }
      const float x = (LIGHT && NPHM_LIGHT_POLY) ? softplus2_light(a[r]) : softplus2(a[r]);
      if constexpr (PREC == 0) {
        H[B][r] = x;
        asm volatile("" : "+v"(H[B][r]));
        if constexpr (r == NR - 1) {
#pragma unroll
          for (int q = NR; q < 16; ++q) H[B][q] = 0.f;
        }
      } else {
        a[r] = x;
        if constexpr (r & 1) pack_pair<r - 1, LIGHT, PREC == 2>(a, H[B]);
        if constexpr (r == NR - 1 && NR < 16) {
#pragma unroll
          for (int q = NR / 2; q < 4; ++q) { H[B].hi[0][q] = 0u; H[B].lo[0][q] = 0u; }
          asm volatile("" : "+v"(H[B].hi[0]));
          asm volatile("" : "+v"(H[B].lo[0]));
        }
      }
    };

    // one chunk between two barriers (LIGHT: this member runs single-pass for this wavefront)
    // L0: lin0 restricted to the coordinates, folded bias as an extra K column (prep_kernels.hip): one
    // MFMA (two on the fp32 path) per 32-feature block, from the member's chunk 0 (ring slot 0)
    auto l0_mfma = [&](auto BB) __attribute__((always_inline)) {
      constexpr int ob = decltype(BB)::value;
      const char* buf = ws.slot(0);
      f32x16 z = {};
      if constexpr (PREC == 0) {
        const float* A = reinterpret_cast<const float*>(buf) + lane;
        const float bk0 = h ? cy : cx, bk1 = h ? 1.f : cz;
        z = __builtin_amdgcn_mfma_f32_32x32x2f32(A[(2 * ob) * 64], bk0, z, 0, 0, 0);
        l0_acc(BB) = __builtin_amdgcn_mfma_f32_32x32x2f32(A[(2 * ob + 1) * 64], bk1, z, 0, 0, 0);
      } else {
        const u32x4* A = reinterpret_cast<const u32x4*>(buf) + lane;
        using half_t = typename std::conditional<PREC == 2, _Float16, __bf16>::type;
        typedef half_t half8 __attribute__((ext_vector_type(8)));
        half_t xh[3], xl[3], xll[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          xh[i] = (half_t)coords[i];
          const float r1 = coords[i] - (float)xh[i];
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
        l0_acc(BB) = mfma16<PREC == 2>(A[ob * 64], __builtin_bit_cast(u32x4, bv), z);
      }
    };
    // bf16 path: the epilogues of the L0 blocks 1..6 ride inside lin1's FIRST chunk (its K-steps 2b, 2b+1
    // read block b, so block b+1 is drained there and the MFMA of block b+2 is issued ahead); only block 0
    // runs exposed.  fp32 path: all 7 blocks exposed in step 0 (an fp32 MFMA chain has shadow to spare).
    constexpr bool L0_FUSED = PREC >= 1 && NPHM_L0_FUSED;
    auto step = [&](auto cc, auto LL) __attribute__((always_inline)) {
      constexpr int c = decltype(cc)::value;
      constexpr int TIER = decltype(LL)::value;            // 0: three passes, 1: one ("light"), 2: two
      constexpr bool LIGHT = TIER == 1;
      constexpr int NPASS = TIER == 1 ? 1 : TIER == 2 ? 2 : NPHM_HEAVY_PASSES;
      const char* buf = ws.slot(c);
      if constexpr (c == 0) {
        l0_mfma(std::integral_constant<int, 0>{});
        if constexpr (L0_FUSED) {
          l0_mfma(std::integral_constant<int, 1>{});
          if constexpr (NPHM_DMA_INSTREAM) ws.template prefetch<0>();
          __builtin_amdgcn_sched_barrier(0);
          static_range<0, 16>([&](auto uu) __attribute__((always_inline)) { epi_l0(std::integral_constant<int, 0>{}, LL, uu); });
          __builtin_amdgcn_sched_barrier(0);
        } else {
          static_for<7>([&](auto BB) __attribute__((always_inline)) {
            constexpr int ob = decltype(BB)::value;
            if constexpr (ob + 1 < 7) l0_mfma(std::integral_constant<int, ob + 1>{});
            if constexpr (NPHM_DMA_INSTREAM && ob < WS::template prefetch_pieces<0>()) ws.template prefetch_piece<0>(ob);
            __builtin_amdgcn_sched_barrier(0);
            static_range<0, (ob == 6 ? LAST_BLOCK_REGS : 16)>([&](auto uu) __attribute__((always_inline)) { epi_l0(BB, LL, uu); });
            __builtin_amdgcn_sched_barrier(0);
          });
        }
      } else {
        constexpr int g = c - 1;
        constexpr int P = c - 1;                       // chunk whose epilogue rides along
        constexpr int NU = (P == 0 || epilogue_exposed(P)) ? 0 : epilogue_units(P);
        auto epi = [&](auto uu) __attribute__((always_inline)) {
          if constexpr (P >= 1) epi_unit(std::integral_constant<int, P>{}, LL, uu);
        };
        // the pieces of the chunk AHEAD of this one go out between the first K-steps, in MFMA shadow
        auto pre = [&](auto kk) __attribute__((always_inline)) {
          constexpr int ks = decltype(kk)::value;
          if constexpr (NPHM_DMA_INSTREAM && ks < WS::template prefetch_pieces<c>()) ws.template prefetch_piece<c>(ks);
          if constexpr (L0_FUSED && c == 1 && ks % 2 == 0 && ks / 2 + 2 <= 6) l0_mfma(std::integral_constant<int, ks / 2 + 2>{});
        };
        auto slot_hook = [&](auto kk, auto mm) __attribute__((always_inline)) {
          if constexpr (L0_FUSED && c == 1) {
            constexpr int ks = decltype(kk)::value, m = decltype(mm)::value, NM = NPASS;
            constexpr int eb = ks / 2 + 1;             // L0 block drained during K-steps 2 (eb - 1), 2 (eb - 1) + 1
            if constexpr (eb <= 6) {
              constexpr int NRb = eb == 6 ? LAST_BLOCK_REGS : 16, sb = (ks & 1) * NM + m, nsb = 2 * NM;
              static_range<unit_begin(sb, nsb, NRb), unit_begin(sb + 1, nsb, NRb)>([&](auto uu) __attribute__((always_inline)) {
                epi_l0(std::integral_constant<int, eb>{}, LL, uu);
              });
            }
          }
        };
        f32x16 d = {};
        if constexpr (!(NPHM_ABLATE & 256)) d = load_frag16(WS::tail_of(buf) + h * 16);
        if constexpr (PREC == 0) {
          if constexpr (g < L1_OB) d = gemm_fused_f32<L1_KS, 6, 7, NU>(buf, d, H, lane, epi, pre);
          else if constexpr (g < L1_OB + L2_OB) d = gemm_fused_f32<L2_KS, 3, 4, NU>(buf, d, G, lane, epi, pre);
          else d = gemm_fused_f32<L3_KS, 6, 7, NU>(buf, d, H, lane, epi, pre);
        } else {
          constexpr int PF = LIGHT ? NPHM_PF_LIGHT : NPHM_PF_HEAVY;
          constexpr bool F16 = PREC == 2;
          // (lin1's first chunk drains the L0 block b + 1 in the slots of K-steps 2b, 2b + 1 and reads it in K-step 2b + 2: runs
          // of at most one aligned K-step pair there)
          constexpr int RUNK_T = TIER == 1 ? NPHM_RUNK_LIGHT : TIER == 2 ? NPHM_RUNK_MID : NPHM_RUNK_HEAVY;
          constexpr int RUNK = (L0_FUSED && c == 1 && RUNK_T > 2) ? 2 : RUNK_T;
          // stacked tail block (split-f16): its one fragment per K-step multiplies xh and xl - rows 0..7 collect xh wh + xl wh,
          // rows 8..15 xh wl + xl wl (added to rows 0..7 by a three-term member's epilogue, ignored by the other tiers)
          constexpr bool STACK = F16 && NPHM_STACK_TAILS && is_tail_chunk(c);
          constexpr int NP = (STACK && NPASS == 3) ? 2 : NPASS, AS = STACK ? 1 : 2;
          constexpr int HEAD_T = TIER == 1 ? NPHM_HEAD_LIGHT : TIER == 2 ? NPHM_HEAD_MID : NPHM_HEAD_HEAVY;
          constexpr int HEAD = (RUNK > 0) ? 0 : (NU < HEAD_T ? NU : HEAD_T);
          if constexpr (g < L1_OB) d = gemm_fused_bf16<L1_KS16, 6, 7, NP, PF, NU, F16, RUNK, AS, HEAD>(buf, d, H, lane, epi, pre, slot_hook);
          else if constexpr (g < L1_OB + L2_OB) d = gemm_fused_bf16<L2_KS16, 3, 4, NP, PF, NU, F16, RUNK, AS, HEAD>(buf, d, G, lane, epi, pre, slot_hook);
          else d = gemm_fused_bf16<L3_KS16, 6, 7, NP, PF, NU, F16, RUNK, AS, HEAD>(buf, d, H, lane, epi, pre, slot_hook);
        }
        accs[c & 1] = d;
        if constexpr (epilogue_exposed(c)) {
          static_range<0, epilogue_units(c)>([&](auto uu) __attribute__((always_inline)) {
            epi_unit(std::integral_constant<int, c>{}, LL, uu);
          });
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    };
    // the whole member under ONE branch on the (wave-uniform) precision class: two straight-line bodies
    auto run_member = [&](auto LL) __attribute__((always_inline)) {
      static_for<CHUNKS_PER_MEMBER>([&](auto cc) __attribute__((always_inline)) {
        constexpr int c = decltype(cc)::value;
        PROF_T(t_s0);
        ws.template sync<c>();
        PROF_T(t_s1);
        PROF_ADD(1, t_s0, t_s1);
        step(cc, LL);
#if NPHM_PROF == 1
        asm volatile("" : "+v"(accs[c & 1][0]));
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // let the last MFMA retire before the stamp
#endif
        PROF_T(t_s2);
        PROF_ADD(c == 0 ? 0 : 2, t_s1, t_s2);
      });
    };
    if constexpr (PREC == 0) {
      run_member(std::integral_constant<int, 0>{});
    } else {
      if (light) run_member(std::integral_constant<int, 1>{});
      else if (mid) run_member(std::integral_constant<int, 2>{});
      else run_member(std::integral_constant<int, 0>{});
    }

    const float f = part + __shfl_xor(part, 32) + b4;

    // ---- Gaussian blend (EnsembledDeepSDF.py:144-149) ------------------------------------------
    acc = fmaf(wk / denom, f, acc);
    ws.next_member();
    PROF_T(t_m2);
    PROF_ADD(4, t_m0, t_m2);
#if NPHM_PROF
    tier_record();
#endif
  }
#if NPHM_PROF == 1
  prof[5] = clock64() - t_kernel;
#endif
#if NPHM_PROF
  __syncthreads();
  // (only for a caller that announces a 96-slot table: stats[12] = "tierprof")
  if (p.stats && p.stats[12] == 0x7469657270726f66ull && threadIdx.x < 80 && tier_prof[threadIdx.x]) atomicAdd(p.stats + 16 + threadIdx.x, tier_prof[threadIdx.x]);
#if NPHM_PROF == 1
  if (p.stats && lane == 0 && n_active > 0) {
    for (int i = 0; i < 6; ++i) atomicAdd(p.stats + 2 + i, (unsigned long long)prof[i]);
    for (int i = 0; i < 3; ++i) atomicAdd(p.stats + 8 + i, (unsigned long long)ws.sprof[i]);
    atomicAdd(p.stats + 11, 1ull);
  }
#endif
#endif
    if (pass == 1 || !(p.refine_band > 0.f)) break;
    // ---- does any wavefront of the workgroup sit on the zero level set? -------------------------------------------
    // (MODE 2: validity recomputed from the tile id and the lane, laundered like the final store's - `valid` as a 0/1 VGPR
    // that lives across the member loop is a scratch spill)
    bool valid_r = valid, hack_r = hack;
#if NPHM_LDS_STASH
    if (MODE != 2) {
      const unsigned long long w3 = lane_w[my_slot()];
      valid_r = (w3 >> 62) & 1ull;
      hack_r = (w3 >> 63) != 0ull;
    }
#endif
    if (MODE == 2) {
      unsigned t3 = tile;
      int l3 = int(__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)));
      asm volatile("" : "+s"(t3), "+v"(l3));
      int lx3, iy3, iz3;
      tile_lane(p, t3, l3 & 31, lx3, iy3, iz3);
      valid_r = binned_group(blockIdx.x) * NW + wave < unsigned(p.n_tiles) && lx3 < p.ix1 - p.ix0 && iy3 < p.ry && iz3 < p.rz;
    }
    const bool wave_near = __ballot(valid_r && !hack_r && fabsf(acc) < p.refine_band) != 0ull;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) wg_mask[0] = 0u;
    __syncthreads();
    if (lane == 0 && wave_near) atomicOr(&wg_mask[0], 1u);
    __syncthreads();
    const bool wg_near = wg_mask[0] != 0u;
    __syncthreads();                                  // wg_mask is reused at the top of the next pass
    if (!wg_near) break;
    if (wave_near) {
      float S2, denom2;
      uint64_t wm[2], hm[2], fm[2];
#if NPHM_LDS_STASH
      const f32x4 qr = lane_q[my_slot()];
      // (the anchor table's address is laundered: hipcc otherwise hoists the ~60 `state + constant` address pairs of this
      // rarely taken path above the member loop and parks them in VGPR lanes - three VGPRs of spill storage for the loop)
      const float* anch_r = anch;
      asm volatile("" : "+s"(anch_r));
      blend_masks<false>(anch_r, qr[0], qr[1], qr[2], valid_r, hack_r, p.refine_prune_tol, -1.f, -1.f, S2, denom2, wm, hm, fm);
#else
      blend_masks<false>(anch, qx, qy, qz, valid, hack, p.refine_prune_tol, -1.f, -1.f, S2, denom2, wm, hm, fm);
#endif
      wmask = wm[0];
      acc = 0.f;
      if (p.stats && lane == 0) atomicAdd(p.stats + 13, (unsigned long long)(nv * __popcll(wmask)));   // refined pairs
    } else {
      wmask = 0ull;
    }
    hmask = fmask = ~0ull;                            // three passes for everything in the refinement
  }

  // eval-mode overwrite (EnsembledDeepSDF.py:260-261): every member predicts 1 for this point
#if NPHM_LDS_STASH
  if (MODE != 2) {
    const int sl = my_slot();
    const unsigned long long w = lane_w[sl];
    if (w >> 63) acc = lane_S[sl] / lane_q[sl][3];
    if (((w >> 62) & 1ull) && (sl & 63) < 32) p.out[w & ((1ull << 62) - 1)] = acc;
    return;
  }
  if (hack) { const int sl = my_slot(); acc = lane_S[sl] / lane_q[sl][3]; }
#else
  if (hack) acc = S / denom;
#endif
  if (MODE == 2) {
    // binned traversal: the output index is not carried through the member loop (a 64-bit VGPR pair that only lives
    // across it becomes a scratch spill); it follows from the tile id (an SGPR) and the lane
    // ... and so does `valid` (everything laundered: the same expressions exist in the prologue, and a CSE'd copy would
    // live across the loop again)
    unsigned t2 = tile;
    int l2 = int(__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)));
    asm volatile("" : "+s"(t2), "+v"(l2));
    int lx, iy, iz;
    tile_lane(p, t2, l2 & 31, lx, iy, iz);
    const bool in_tile2 = binned_group(blockIdx.x) * NW + wave < unsigned(p.n_tiles);
    if (in_tile2 && l2 < 32 && lx < p.ix1 - p.ix0 && iy < p.ry && iz < p.rz)
      p.out[(int64_t(lx) * p.ry + iy) * p.rz + iz] = acc;
    return;
  }
  if (valid && h == 0) p.out[out_idx] = acc;
}

// ---- launches ----------------------------------------------------------------------------------------------------------
// The nine instantiations are what compiling this file costs (~20 s each).  nphm_amd/build.py compiles it FOUR times in
// parallel: -DNPHM_EVAL_PART=1 = the C ABI, the tile pre-pass and the dispatch below; 10 / 11 / 12 = the three kernels of
// precision 0 / 1 / 2 behind launch_prec<P>.  Undefined (0: tools/build_variant.sh, `hipcc -c` of this file): everything in one
// translation unit, as in rounds 1-5.
#ifndef NPHM_EVAL_PART
#define NPHM_EVAL_PART 0
#endif
void launch_prec0(int mode, dim3 grid, dim3 block, hipStream_t st, const EvalArgs& a);
void launch_prec1(int mode, dim3 grid, dim3 block, hipStream_t st, const EvalArgs& a);
void launch_prec2(int mode, dim3 grid, dim3 block, hipStream_t st, const EvalArgs& a);
#if NPHM_EVAL_PART == 0 || NPHM_EVAL_PART >= 10
template <int PREC>
static void launch_modes(int mode, dim3 grid, dim3 block, hipStream_t st, const EvalArgs& a) {
#ifdef NPHM_DEV_ONLY22   // ISA-inspection / variant builds: one instantiation (20 s instead of 3 min); never for a library that runs
  if constexpr (PREC == 2) { if (mode == 2) hipLaunchKernelGGL((eval_kernel<2, 2>), grid, block, 0, st, a); }
#else
  if (mode == 0) hipLaunchKernelGGL((eval_kernel<0, PREC>), grid, block, 0, st, a);
  else if (mode == 1) hipLaunchKernelGGL((eval_kernel<1, PREC>), grid, block, 0, st, a);
  else hipLaunchKernelGGL((eval_kernel<2, PREC>), grid, block, 0, st, a);
#endif
}
#endif
#if NPHM_EVAL_PART == 0 || NPHM_EVAL_PART == 10
void launch_prec0(int mode, dim3 grid, dim3 block, hipStream_t st, const EvalArgs& a) { launch_modes<0>(mode, grid, block, st, a); }
#endif
#if NPHM_EVAL_PART == 0 || NPHM_EVAL_PART == 11
void launch_prec1(int mode, dim3 grid, dim3 block, hipStream_t st, const EvalArgs& a) { launch_modes<1>(mode, grid, block, st, a); }
#endif
#if NPHM_EVAL_PART == 0 || NPHM_EVAL_PART == 12
void launch_prec2(int mode, dim3 grid, dim3 block, hipStream_t st, const EvalArgs& a) { launch_modes<2>(mode, grid, block, st, a); }
#endif

}  // namespace nphm

#if NPHM_EVAL_PART <= 1
// ============================================================================================
// C ABI (include/nphm_amd.h)
// ============================================================================================
extern "C" {

// two-pass tier of the adaptive mode: members whose normalised blend weight stays below this value (and reaches
// NPHM_LIGHT_TOL) in a wavefront; NPHM_AMD_MID_TOL overrides the default (<= NPHM_LIGHT_TOL switches the tier off)
static float nphm_mid_tol() {
  static const float v = [] {
    const char* e = getenv("NPHM_AMD_MID_TOL");
    return e ? float(atof(e)) : NPHM_MID_TOL;
  }();
  return v;
}

// `precision` = mode (low byte) | optional tier overrides (include/nphm_amd.h: NPHM_PREC_WITH_TIERS)
static int prec_mode(int precision) { return precision & 0xff; }
static bool is_f16(int precision) { return prec_mode(precision) == NPHM_PREC_F16X3 || prec_mode(precision) == NPHM_PREC_F16X3_ADAPTIVE2; }
static bool is_adaptive(int precision) {
  const int m = prec_mode(precision);
  return m == NPHM_PREC_BF16X3_ADAPTIVE || m == NPHM_PREC_BF16X3_ADAPTIVE2 || m == NPHM_PREC_F16X3_ADAPTIVE2;
}

static int check_prec(int precision) {
  const int m = prec_mode(precision);
  if (m != NPHM_PREC_F32 && m != NPHM_PREC_BF16X3 && m != NPHM_PREC_F16X3 && !is_adaptive(precision))
    return nphm_fail_msg("nphm_identity_eval: unsupported precision mode");
  return 0;
}

// tier thresholds of a precision mode (normalised blend weight below which a member runs single-pass / two-pass in a
// wavefront): the mode's defaults, or the half-octave codes carried in bits 8..23 of `precision`
static float tier_of_code(int code, float dflt) {
  if (code == 0) return dflt;
  if (code == 255) return -1.f;                       // tier switched off
  return exp2f(1.f - 0.5f * float(code));             // NPHM_TIER_TOL(code)
}
static void set_tiers(nphm::EvalArgs& a, int precision) {
  const int m = prec_mode(precision);
  float light = -1.f, mid = -1.f;
  if (m == NPHM_PREC_F16X3_ADAPTIVE2) { light = NPHM_LIGHT_TOL_F16; mid = NPHM_MID_TOL_F16; }
  else if (m == NPHM_PREC_BF16X3_ADAPTIVE2) { light = NPHM_LIGHT_TOL; mid = nphm_mid_tol(); }
  else if (m == NPHM_PREC_BF16X3_ADAPTIVE) light = NPHM_LIGHT_TOL;
  if (is_adaptive(precision)) {
    light = tier_of_code((precision >> 8) & 0xff, light);
    if (m != NPHM_PREC_BF16X3_ADAPTIVE) mid = tier_of_code((precision >> 16) & 0xff, mid);
  }
  a.light_tol = light;
  a.mid_tol = mid;
  // sign-safe refinement band (bits 24..30 of `precision`, half-octave code like the tiers; 0 = off): tiles with a value
  // inside the band are re-evaluated with all members on the three-pass product at 1/32 of the pruning budget
  const int rc = (precision >> 24) & 0x7f;
  a.refine_band = rc ? exp2f(1.f - 0.5f * float(rc)) : 0.f;
  a.refine_prune_tol = a.prune_tol >= 0.f ? a.prune_tol * (1.f / 32.f) : -1.f;
  if (a.prune_tol < 0.f && !is_adaptive(precision)) a.refine_band = 0.f;     // nothing to refine: already the exact setting
}

static void launch_by_precision(int mode, int precision, dim3 grid, dim3 block, hipStream_t st, const nphm::EvalArgs& a) {
  if (prec_mode(precision) == NPHM_PREC_F32) nphm::launch_prec0(mode, grid, block, st, a);
  else if (is_f16(precision)) nphm::launch_prec2(mode, grid, block, st, a);
  else nphm::launch_prec1(mode, grid, block, st, a);
}

static void fill_common(nphm::EvalArgs& a, const void* packed, const void* latent_state, float* out,
                        unsigned long long* stats, float prune_tol, int64_t hack_chunk) {
  memset(&a, 0, sizeof(a));
  a.packed_f32 = static_cast<const float*>(packed);
  a.packed_bf16 = reinterpret_cast<const uint16_t*>(static_cast<const char*>(packed) + nphm::PACKED_F32_FLOATS * 4);
  a.packed_f16 = a.packed_bf16 + nphm::PACKED_BF16_HALFS;
  a.state = static_cast<const float*>(latent_state);
  a.out = out;
  a.stats = stats;
  a.prune_tol = prune_tol;
  a.light_tol = -1.f;                    // set per precision mode by the callers
  a.mid_tol = -1.f;
  a.hack_chunk = hack_chunk;
}

// workspace of the binned grid evaluation: [order][ids][keys in][keys out][masks][S, denom][sort temp]
namespace {
struct BinLayout {
  int ntx, nty, ntz;
  int64_t n_tiles;
  size_t order, ids, keys_in, keys_out, masks, sd, temp, temp_bytes, bytes;
};
bool bin_layout(int nlx, int ry, int rz, BinLayout& l) {
  l.ntx = (nlx + nphm::TLX - 1) / nphm::TLX; l.nty = (ry + nphm::TLY - 1) / nphm::TLY; l.ntz = (rz + nphm::TLZ - 1) / nphm::TLZ;
  l.n_tiles = int64_t(l.ntx) * l.nty * l.ntz;
  if (l.n_tiles <= 0 || l.n_tiles > 0x3fffffffLL) return false;
  size_t o = 0;
  auto take = [&](size_t b) { size_t r = o; o += (b + 255) / 256 * 256; return r; };
  const size_t n = size_t(l.n_tiles);
  l.order = take(n * 4); l.ids = take(n * 4); l.keys_in = take(n * 8); l.keys_out = take(n * 8);
  l.masks = take(n * 24); l.sd = take(n * 32 * 8);
  size_t tb = 0;
  (void)hipcub::DeviceRadixSort::SortPairs(nullptr, tb, static_cast<const uint64_t*>(nullptr), static_cast<uint64_t*>(nullptr),
                                           static_cast<const unsigned*>(nullptr), static_cast<unsigned*>(nullptr), int(n));
  l.temp_bytes = tb;
  l.temp = take(tb);
  l.bytes = o;
  return true;
}
}  // namespace

size_t nphm_identity_grid_workspace_bytes(int n_x_local, int ry, int rz) {
  BinLayout l;
  if (n_x_local <= 0 || ry <= 0 || rz <= 0 || !bin_layout(n_x_local, ry, rz, l)) return 0;
  return l.bytes;
}

static int launch_grid(nphm::EvalArgs& a, int precision, void* workspace, size_t workspace_bytes, hipStream_t st,
                       const char* who) {
  set_tiers(a, precision);
#if NPHM_PROF
  if (const char* e = getenv("NPHM_PROF_LIGHT_TOL")) a.light_tol = float(atof(e));   // timing builds: force all-light / all-heavy
#endif
  const int nx = a.ix1 - a.ix0;
  if (workspace) {
    // binned: tiles sorted by member mask (tile_prepass_kernel), one wavefront per tile in that order
    BinLayout l;
    if (!bin_layout(nx, a.ry, a.rz, l)) return nphm_fail_msg("slab too large for one launch");
    if (workspace_bytes < l.bytes) return nphm_fail_msg("workspace smaller than nphm_identity_grid_workspace_bytes()");
    char* ws = static_cast<char*>(workspace);
    a.ntx = l.ntx; a.nty = l.nty; a.ntz = l.ntz; a.n_tiles = int(l.n_tiles);
    a.tile_order = reinterpret_cast<const unsigned*>(ws + l.order);
    a.tile_ids = reinterpret_cast<unsigned*>(ws + l.ids);
    a.tile_keys = reinterpret_cast<uint64_t*>(ws + l.keys_in);
    a.tile_masks = reinterpret_cast<uint64_t*>(ws + l.masks);
    a.tile_sd = reinterpret_cast<float2*>(ws + l.sd);
    const unsigned pre_blocks = unsigned((l.n_tiles + 7) / 8);                 // 4 wavefronts x 2 tiles
    hipLaunchKernelGGL(nphm::tile_prepass_kernel, dim3(pre_blocks), dim3(256), 0, st, a);
    size_t tb = l.temp_bytes;
    hipError_t e = hipcub::DeviceRadixSort::SortPairs(ws + l.temp, tb, a.tile_keys, reinterpret_cast<uint64_t*>(ws + l.keys_out),
                                                      a.tile_ids, reinterpret_cast<unsigned*>(ws + l.order), int(l.n_tiles), 0, 64, st);
    if (e != hipSuccess) return nphm_fail(who, e);
    const int64_t round = nphm::XCD_RUN > 1 ? 8 * int64_t(nphm::XCD_RUN) : 1;
    const int64_t groups = ((l.n_tiles + nphm::NW - 1) / nphm::NW + round - 1) / round * round;
    const dim3 grid((unsigned)groups), block(64 * nphm::NW);
    launch_by_precision(2, precision, grid, block, st, a);
    e = hipGetLastError();
    if (e != hipSuccess) return nphm_fail(who, e);
    return 0;
  }
  a.nbx = (nx + nphm::BRX - 1) / nphm::BRX; a.nby = (a.ry + nphm::BRY - 1) / nphm::BRY;
  a.nbz = (a.rz + nphm::BRZ - 1) / nphm::BRZ;
  a.nsx = (a.nbx + nphm::SBX - 1) / nphm::SBX; a.nsy = (a.nby + nphm::SBY - 1) / nphm::SBY;
  a.nsz = (a.nbz + nphm::SBZ - 1) / nphm::SBZ;
  const int64_t supers = (int64_t(a.nsx) * a.nsy * a.nsz + 7) / 8 * 8;       // padded to the 8 XCDs
  const int64_t bricks = supers * (nphm::SBX * nphm::SBY * nphm::SBZ);
  if (bricks > 0x7fffffffLL) return nphm_fail_msg("slab too large for one launch");
  const dim3 grid((unsigned)bricks), block(64 * nphm::NW);
  launch_by_precision(1, precision, grid, block, st, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return nphm_fail(who, e);
  return 0;
}

int nphm_identity_eval_points(const void* packed, const void* latent_state,
                              const float* xyz, int n_rows, int64_t n_points,
                              int64_t hack_chunk, float prune_tol, int precision,
                              float* sdf_out, unsigned long long* stats, void* stream) {
  if (!packed || !latent_state || !xyz || !sdf_out) return nphm_fail_msg("nphm_identity_eval_points: null pointer");
  if (n_rows <= 0 || n_points <= 0) return nphm_fail_msg("nphm_identity_eval_points: empty input");
  if (check_prec(precision)) return -2;
  nphm::EvalArgs a;
  fill_common(a, packed, latent_state, sdf_out, stats, prune_tol, hack_chunk);
  a.xyz = xyz;
  a.n_points = n_points;
  const int64_t tiles = (n_points + 32 * nphm::NW - 1) / (32 * nphm::NW);
  if (tiles > 0x7fffffffLL) return nphm_fail_msg("nphm_identity_eval_points: too many points");
  const dim3 grid((unsigned)tiles, n_rows), block(64 * nphm::NW);
  hipStream_t st = static_cast<hipStream_t>(stream);
  set_tiers(a, precision);
#if NPHM_PROF
  if (const char* e = getenv("NPHM_PROF_LIGHT_TOL")) a.light_tol = float(atof(e));   // timing builds: force all-light / all-heavy
#endif
  launch_by_precision(0, precision, grid, block, st, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return nphm_fail("nphm_identity_eval_points launch", e);
  return 0;
}

int nphm_identity_eval_grid(const void* packed, const void* latent_state,
                            const float* axis_x, const float* axis_y, const float* axis_z,
                            int rx, int ry, int rz, int ix0, int ix1,
                            int64_t hack_chunk, float prune_tol, int precision,
                            float* sdf_out, unsigned long long* stats, void* workspace, size_t workspace_bytes,
                            void* stream) {
  if (!packed || !latent_state || !axis_x || !axis_y || !axis_z || !sdf_out)
    return nphm_fail_msg("nphm_identity_eval_grid: null pointer");
  if (rx <= 0 || ry <= 0 || rz <= 0 || ix0 < 0 || ix1 > rx || ix0 >= ix1)
    return nphm_fail_msg("nphm_identity_eval_grid: bad grid / slab bounds");
  if (check_prec(precision)) return -2;
  nphm::EvalArgs a;
  fill_common(a, packed, latent_state, sdf_out, stats, prune_tol, hack_chunk);
  a.ax = axis_x; a.ay = axis_y; a.az = axis_z;
  a.rx = rx; a.ry = ry; a.rz = rz; a.ix0 = ix0; a.ix1 = ix1;
  return launch_grid(a, precision, workspace, workspace_bytes, static_cast<hipStream_t>(stream), "nphm_identity_eval_grid");
}

int nphm_identity_eval_grid_planes(const void* packed, const void* latent_state,
                                   const float* axis_x, const float* axis_y, const float* axis_z,
                                   int rx, int ry, int rz, const int* x_planes, int n_planes,
                                   int64_t hack_chunk, float prune_tol, int precision,
                                   float* sdf_out, unsigned long long* stats, void* workspace, size_t workspace_bytes,
                                   void* stream) {
  if (!packed || !latent_state || !axis_x || !axis_y || !axis_z || !sdf_out || !x_planes)
    return nphm_fail_msg("nphm_identity_eval_grid_planes: null pointer");
  if (rx <= 0 || ry <= 0 || rz <= 0 || n_planes <= 0 || n_planes > rx)
    return nphm_fail_msg("nphm_identity_eval_grid_planes: bad grid / plane count");
  if (check_prec(precision)) return -2;
  nphm::EvalArgs a;
  fill_common(a, packed, latent_state, sdf_out, stats, prune_tol, hack_chunk);
  a.ax = axis_x; a.ay = axis_y; a.az = axis_z;
  a.rx = rx; a.ry = ry; a.rz = rz; a.ix0 = 0; a.ix1 = n_planes;
  a.xplanes = x_planes;
  return launch_grid(a, precision, workspace, workspace_bytes, static_cast<hipStream_t>(stream), "nphm_identity_eval_grid_planes");
}

int nphm_identity_eval_grid_points(const void* packed, const void* latent_state,
                                   const float* xyz_slab, int rx, int ry, int rz, int ix0, int ix1,
                                   int64_t hack_chunk, float prune_tol, int precision,
                                   float* sdf_out, unsigned long long* stats, void* workspace, size_t workspace_bytes,
                                   void* stream) {
  if (!xyz_slab) return nphm_fail_msg("nphm_identity_eval_grid_points: null pointer");
  // the axis pointers are never dereferenced when xyz_slab is given
  const float* dummy = xyz_slab;
  if (!packed || !latent_state || !sdf_out) return nphm_fail_msg("nphm_identity_eval_grid_points: null pointer");
  if (rx <= 0 || ry <= 0 || rz <= 0 || ix0 < 0 || ix1 > rx || ix0 >= ix1)
    return nphm_fail_msg("nphm_identity_eval_grid_points: bad grid / slab bounds");
  if (check_prec(precision)) return -2;
  nphm::EvalArgs a;
  fill_common(a, packed, latent_state, sdf_out, stats, prune_tol, hack_chunk);
  a.xyz = xyz_slab;
  a.ax = dummy; a.ay = dummy; a.az = dummy;
  a.rx = rx; a.ry = ry; a.rz = rz; a.ix0 = ix0; a.ix1 = ix1;
  return launch_grid(a, precision, workspace, workspace_bytes, static_cast<hipStream_t>(stream), "nphm_identity_eval_grid_points");
}

}  // extern "C"
#endif  // NPHM_EVAL_PART <= 1

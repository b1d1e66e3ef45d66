/*
 * nphm_amd.h — C ABI of libnphm_amd.so: MI355X (gfx950) kernels for NPHM's batched
 * neural-field evaluation hot path.
 *
 * The reference (SimonGiebenhain/NPHM) has no FFI of its own: the boundary of this path is
 * the Python nn.Module surface.  Each entry point below is what a binding of
 * that surface calls; the reference code it replaces is cited per function
 * (paths relative to the reference checkout).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (HBM, fp32 unless stated) owned by the caller;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); all work is
 *     stream-ordered, nothing synchronises the host;
 *   - return value: 0 = ok, <0 = error (message via nphm_last_error());
 *   - no ownership transfer, no hidden allocation: the caller provides the packed-weight and
 *     latent-state buffers (sizes from the *_bytes() queries).
 */
#ifndef NPHM_AMD_H
#define NPHM_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NPHM_AMD_ABI_VERSION 12

/* ---- library ------------------------------------------------------------------------ */
int nphm_abi_version(void);
const char* nphm_last_error(void);

/* Measurement aid (bench.py's `mfma_sustained`; not on any product path, no reference counterpart):
 * runs an MFMA-only loop shaped like the kernels' GEMM body (bf16 32x32x16, 2 wavefronts per SIMD,
 * A fragments re-read from LDS, pseudo-random operands) for a few ms on `stream`, SYNCHRONISES, and
 * returns the executed TFLOP/s and the shader clock it ran at - the matrix-pipe rate the chip's power
 * management sustains, to read next to the 2.5 PFLOP/s datasheet peak. */
int nphm_probe_mfma_rate(double* tflops, double* clock_ghz, void* stream);

/* ---- NPHM identity field: FastEnsembleDeepSDFMirrored -------------------------------- */
/* Architecture the fused kernel is specialised for (scripts/configs/nphm.yaml:1-7):
 * lat_dim_glob 64, lat_dim_loc 32, n_loc 39, n_symm_pairs 16, hidden 200, n_layers 4,
 * out_dim 1, input_dim 3.  Returns 1 if (and only if) the arguments equal that config. */
int nphm_identity_supported(int lat_dim_glob, int lat_dim_loc, int n_loc, int n_symm_pairs,
                            int hidden_dim, int n_layers, int out_dim, int input_dim);

/* Precision of the per-member MLP GEMMs. */
#define NPHM_PREC_F32     0   /* v_mfma_f32_32x32x2_f32: exact fp32 products            */
#define NPHM_PREC_BF16X3  1   /* split-bf16 (hi*hi + hi*lo + lo*hi) on 32x32x16 bf16 MFMA */
#define NPHM_PREC_BF16X3_ADAPTIVE 2  /* the same for every member whose normalised blend weight reaches
                                        NPHM_LIGHT_TOL somewhere in the wavefront; single-pass bf16
                                        (hi*hi) for the others: their error enters the blend scaled by
                                        a weight < NPHM_LIGHT_TOL */
#define NPHM_LIGHT_TOL 1e-3f
#define NPHM_PREC_BF16X3_ADAPTIVE2 3 /* ... and two passes (xh*wh + xl*wh: weights rounded to bf16) for members that stay between
                                        NPHM_LIGHT_TOL and NPHM_MID_TOL in the wavefront; three passes from NPHM_MID_TOL on */
#define NPHM_MID_TOL 1e-2f
/* split-f16 flavours of the same products (v_mfma_f32_32x32x16_f16 on binary16 halves: 11-bit significands - three passes
 * carry 22 bits against 16, the two-pass and single-pass tiers are 8x closer to fp32 than their bf16 counterparts, so
 * their thresholds sit 8x higher at the same error; activations are exact up to 908 (131 008 in the kernels' scaled
 * domain: beyond that the hi half saturates) */
#define NPHM_PREC_F16X3 4            /* three passes for every member */
#define NPHM_PREC_F16X3_ADAPTIVE2 5  /* single pass below NPHM_LIGHT_TOL_F16, two passes below NPHM_MID_TOL_F16, three from there on */
#define NPHM_LIGHT_TOL_F16 8e-3f
#define NPHM_MID_TOL_F16 8e-2f
/* Per-call tier thresholds of the adaptive modes: the `precision` argument carries the mode in its low byte and,
 * optionally, half-octave codes of the two thresholds in bits 8..15 (single-pass tier) and 16..23 (two-pass tier):
 * code 0 = the mode's default above, code c in 1..254 = 2^(1 - c/2) (c = 21: 1.4e-3, c = 15: 1.1e-2, c = 9: 8.8e-2),
 * 255 = tier off.  What nphm_amd.calibrate_numerics picks per checkpoint (the error of a tier scales with the member
 * values of the checkpoint at hand; the defaults are calibrated on seeded random-init weights). */
#define NPHM_PREC_WITH_TIERS(mode, light_code, mid_code) ((mode) | ((light_code) << 8) | ((mid_code) << 16))
#define NPHM_TIER_TOL(code) exp2f(1.f - 0.5f * (float)(code))

size_t nphm_identity_packed_bytes(void);
size_t nphm_identity_latent_state_bytes(int n_rows);

/* Re-lay the state_dict tensors `ensembled_deep_sdf.lin{0..4}.{weight,bias}`
 * ([24,200,99],[24,101,200],[24,200,200],[24,200,200],[24,1,200] + biases) into MFMA
 * fragment order.  Replaces the per-call repeat_interleave/cat/permute of
 * EnsembledLinear.forward (src/NPHM/models/EnsembledDeepSDF.py:43-54). */
int nphm_identity_pack(const float* const lin_weight[5], const float* const lin_bias[5],
                       void* packed, void* stream);

/* Per latent code (one per batch row): predicted anchors = mlp_pos(z_glob) + mean anchors
 * (EnsembledDeepSDF.py:228-229) and the constant-latent part of lin0 / lin2 folded into
 * per-member bias vectors (replaces the cond tensor of EnsembledDeepSDF.py:247-255).
 *   lat_rows   [n_rows, 1344]   row 0 of lat_rep for every batch row
 *   mlp_pos_*  state_dict tensors mlp_pos.{0,2,4}.{weight,bias}; pos_mlp_dim = 256 or 128
 *   anchors_mean [39,3]
 *   anchors_out  [n_rows,39,3]  (second return value of forward()) */
int nphm_identity_prepare_latent(const void* packed,
                                 const float* const lin_weight[5], const float* const lin_bias[5],
                                 const float* const mlp_pos_weight[3], const float* const mlp_pos_bias[3],
                                 int pos_mlp_dim, const float* anchors_mean,
                                 const float* lat_rows, int n_rows,
                                 void* latent_state, float* anchors_out, void* stream);

/* The same prologue when the caller already holds the anchors of the rows ([n_rows,39,3], mean anchors included): the
 * autograd tier of the fitting loops evaluates mlp_pos as a differentiable head (nphm_head_forward) and needs the
 * field state for exactly those anchors (ABI 6). */
int nphm_identity_prepare_latent_anchors(const float* const lin_weight[5], const float* const lin_bias[5],
                                         const float* lat_rows, const float* anchors, int n_rows, void* latent_state,
                                         void* stream);

/* Magnitude bounds of the ensemble members for the pruning rule and the precision tiers of the inference kernels
 * (nphm_identity_eval_*): bounds [40][4] = (b0, b1, b2, unused) per member with B_k(d) = b0 + b1 d + b2 d^2 >= |f_k| at
 * distance d from anchor k (member 39, the background member: d = 0).  With bounds installed, "weight" in the rules of
 * prune_tol / NPHM_LIGHT_TOL / NPHM_MID_TOL reads w_k B_k(d_k): what a member's term can contribute in SDF units, so
 * 40 * prune_tol bounds the pruning error itself.  NULL (and every freshly prepared state) = (1, 0, 0): the bare weights.
 * Device pointer; the latent_state of nphm_identity_prepare_latent is updated in place for all n_rows. */
int nphm_identity_set_member_bounds(void* latent_state, int n_rows, const float* bounds, void* stream);

/* FastEnsembleDeepSDFMirrored.forward for latents that are constant along the point axis
 * (EnsembledDeepSDF.py:203-267):  sdf_out[b,n] for xyz[b,n,:].
 *   hack_chunk > 0 reproduces the eval-mode overwrite (EnsembledDeepSDF.py:260-261) of the last
 *   point of every forward() call when the n_points of a row are evaluated in chunks of
 *   hack_chunk points: indices i with (i+1) % hack_chunk == 0 or i == n_points-1 get
 *   sum(w)/(sum(w)+1e-6).  hack_chunk = n_points is a single eval-mode forward(); 0 = train mode.
 *   prune_tol: per point the smallest-weight members are dropped as long as their normalised blend
 *   weights sum to <= 40*prune_tol (so |error| <= 40*prune_tol*max|f_k|); a wavefront skips a member
 *   that all of its 32 points drop; < 0 evaluates all 40 members.
 *   stats (nullable, device, 16 x u64): stats[0] += evaluated (point, member) pairs, stats[1] +=
 *   points, stats[15] += the pairs evaluated single-pass (adaptive mode) — the executed-work counters
 *   behind bench.py's roofline figure. */
int nphm_identity_eval_points(const void* packed, const void* latent_state,
                              const float* xyz, int n_rows, int64_t n_points,
                              int64_t hack_chunk, float prune_tol, int precision,
                              float* sdf_out, unsigned long long* stats, void* stream);

/* Dense-grid evaluation = get_logits(decoder, lat, create_grid_points_from_bounds(...))
 * (src/NPHM/models/reconstruction.py:6-25, src/NPHM/utils/reconstruction.py:5-20) for the
 * x-slab [ix0, ix1) of an [rx,ry,rz] 'ij'-ordered grid (x slowest, z fastest).
 *   axis_x/y/z : fp32 axis coordinates (float32(np.linspace(min,max,res)))
 *   hack_chunk : >0 reproduces get_logits' per-chunk eval-mode overwrite: flat indices i with
 *                (i+1) % hack_chunk == 0 or i == rx*ry*rz-1 get sum(w)/(sum(w)+1e-6); 0 = off
 *   sdf_out    : [(ix1-ix0)*ry*rz] (slab-local, same flattened order) */
int nphm_identity_eval_grid(const void* packed, const void* latent_state,
                            const float* axis_x, const float* axis_y, const float* axis_z,
                            int rx, int ry, int rz, int ix0, int ix1,
                            int64_t hack_chunk, float prune_tol, int precision,
                            float* sdf_out, unsigned long long* stats,
                            void* workspace, size_t workspace_bytes, void* stream);

/* Workspace of the three grid entry points (device memory, >= this many bytes for a slab of
 * n_x_local x-planes; contents are scratch, nothing persists between calls).  With a workspace the
 * 4x4x2-voxel tiles a wavefront works on are first binned by their set of active ensemble members
 * (one pre-pass kernel + one radix sort on the same stream): the 8 wavefronts of a workgroup then
 * stream (nearly) the same members, which removes the passes a wavefront idles through for members only
 * its neighbours need and lets the workgroups on one XCD share their weights in L2 (-13 % at 256^3).
 * Results are bitwise identical to workspace = NULL (brick-order traversal, no extra memory). */
size_t nphm_identity_grid_workspace_bytes(int n_x_local, int ry, int rz);

/* The same for an arbitrary ascending set of x-planes (device array x_planes[n_planes]): the
 * multi-GPU partition hands every rank the planes of every world_size-th 8-plane brick slab, which
 * balances the (spatially varying) ensemble work.  sdf_out [n_planes*ry*rz] in list order; the chunk
 * overwrite uses GLOBAL flat indices, so the union of all ranks' outputs is the single-GPU volume. */
int nphm_identity_eval_grid_planes(const void* packed, const void* latent_state,
                                   const float* axis_x, const float* axis_y, const float* axis_z,
                                   int rx, int ry, int rz, const int* x_planes, int n_planes,
                                   int64_t hack_chunk, float prune_tol, int precision,
                                   float* sdf_out, unsigned long long* stats,
                                   void* workspace, size_t workspace_bytes, void* stream);

/* First-order backward of FastEnsembleDeepSDFMirrored.forward for the latent-fitting loop
 * (src/NPHM/models/fitting.py:111, :167: loss.backward() through decoder(xc, z_id); what autograd
 * computes through EnsembledDeepSDF.py:101-150).  Train-mode forward only (no chunk overwrite).
 *   nphm_identity_pack_bwd : transposed split-bf16 pack of lin0..lin3 (once per weight update)
 *   nphm_identity_backward : the caller lists, per (batch row, member), the points whose normalised
 *     blend weight exceeds its pruning tolerance: tiles [n_tiles][4] = (row, member, offset into
 *     point_list, count <= 64), point_list = point indices inside the row.  sdf = forward values,
 *     grad_sdf = dL/dsdf, both [n_rows, n_points].  WRITES (ABI 8; rounds 1-3 accumulated with float atomics)
 *       grad_xyz [n_rows,n_points,3], grad_anchors [n_rows,39,3] (anchors = second forward output),
 *       grad_b0 / grad_b2 [n_rows,40,200] = dL/d(b0 + W0[:,3:] cond_k), dL/d(b2 + W2[:,104:] cond_k / sqrt2),
 *     which the host chains through mlp_pos and the latent columns.  No weight gradients.  Two launches: the member kernel
 *     leaves per-tile / per-(point, member) records in scratch (nphm_identity_backward_scratch_bytes; n_tiles = the table's
 *     capacity), a second one adds a pair's tiles in table order and a point's members in member order (blend_weights
 *     [n_rows,n_points,40] != 0 marks the listed pairs - what nphm_identity_build_lists returns): bitwise reproducible.
 *   nphm_identity_member_forward : the forward half on the same (row, member) point lists: the member
 *     predictions f_k into member_sdf [n_rows, n_points, 40] (entries of unlisted pairs are left
 *     untouched); the host blends them.  Serves the forward of the autograd tier, whose query points are
 *     scattered surface samples (a brick-coherent wavefront of eval_points would touch most members). */
size_t nphm_identity_bwd_packed_bytes(void);
/* The (row, member) point lists of the two kernels above, built ON THE DEVICE with fixed capacity (no size
 * travels to the host: no synchronisation, static shapes, hipGraph-capturable).  With T =
 * nphm_identity_list_tiles(n_points) = ceil(n_points / 64): blend_weights [n_rows, n_points, 40] = normalised
 * blend weight of every (point, member), 0 where the pruning rule of the fused kernel drops the member
 * (prune_tol < 0: nothing dropped; EnsembledDeepSDF.py:129-150 for the weights); tiles = room for
 * 2 * n_rows * 40 * T entries of 4 ints (the used tiles end up compacted at the front, the second half is
 * scratch); *n_tiles_used (device int) = their number; point_list [n_rows * 40 * 64 T].  The kernels take the
 * capacity n_rows * 40 * T as n_tiles and the device count as n_tiles_dev.  tiles = n_tiles_used = point_list = NULL: the
 * blend weights alone (one launch). */
int nphm_identity_list_tiles(int64_t n_points);
int nphm_identity_build_lists(const void* latent_state, const float* xyz, int n_rows, int64_t n_points, float prune_tol,
                              float* blend_weights, int* tiles, int* n_tiles_used, int* point_list, void* stream);
int nphm_identity_member_forward(const void* packed, const void* packed_bwd, const void* latent_state,
                                 const float* xyz, int64_t n_points, const int* tiles, int n_tiles,
                                 const int* n_tiles_dev, const int* point_list, float* member_sdf, void* stream);
int nphm_identity_pack_bwd(const float* const lin_weight[5], void* packed_bwd, void* stream);
size_t nphm_identity_backward_scratch_bytes(int n_rows, int64_t n_points, int n_tiles);
int nphm_identity_backward(const void* packed, const void* packed_bwd, const void* latent_state,
                           const float* xyz, const float* sdf, const float* grad_sdf, int n_rows, int64_t n_points,
                           const int* tiles, int n_tiles, const int* n_tiles_dev, const int* point_list,
                           const float* blend_weights, void* scratch,
                           float* grad_xyz, float* grad_anchors, float* grad_b0, float* grad_b2, void* stream);

/* Training tier of the identity ensemble: what compute_loss needs from the 40 member MLPs
 * (src/NPHM/models/loss_functions.py:36-49: decoder(...) followed by gradient(pred, x) with create_graph=True,
 * diff_operators.py:6-16; training.py:124: loss.backward() w.r.t. every weight).  The host keeps the Gaussian
 * blend (EnsembledDeepSDF.py:129-150) in autograd; these two kernels replace the member MLPs
 * (EnsembledDeepSDF.py:101-126) and their double backward.  tiles [n_tiles][4] = (row, member, offset into
 * point_list, count), ORDERED BY MEMBER (hence by weight set), count <= 64 for the forward kernel and <= 32 for the
 * backward kernel (which spends the other 32 MFMA columns on the tangent stream); point_list = point indices inside
 * the row.
 *   nphm_identity_train_forward : member_sdf [n_rows,n_points,40] = f_k, member_grad [n_rows,n_points,40,3] =
 *     d f_k / d xyz for the listed triples (others untouched).
 *   nphm_identity_train_backward : seeds grad_member_sdf = dL/df_k and grad_member_grad = dL/d(d f_k/d xyz) (NULL:
 *     zero) -> ACCUMULATES grad_xyz (as nphm_identity_backward, for
 *     phi = sum dL/df_k f_k + dL/d(grad f_k) . grad f_k; float atomics over the members of a point - the one output of the
 *     training tier that is not bitwise reproducible, and the one a training step never uses), stores the operands of the weight gradients of lin1 .. lin3
 *     into saved (nphm_identity_train_saved_bytes(n_tiles, operands) bytes; per tile [1005 rows][64 columns]; `operands`,
 *     the same in all three calls: 0 = fp32; 1 = bf16 - half the operand traffic of both kernels, the weight-gradient
 *     products then carry 8-bit mantissas (parameter gradients to 4e-4 of their largest entry); 2 (ABI 9) = binary16 with
 *     the per-stream power-of-two scales `operand_scales` [4] that nphm_identity_train_operand_scales derives from the
 *     seeds of the step (11-bit mantissas: 5e-5; values beyond the format saturate) -, column = 32 * stream + point with stream 0 = value,
 *     1 = tangent along the seed direction; rows = inputs of lin1..lin3 followed by the adjoints of their
 *     pre-activations, scaled domain) and WRITES edge (nphm_identity_train_edge_bytes(n_tiles); ABI 8): per tile 1600
 *     floats = the tile's contribution to the gradients of lin0 (3 input columns + folded bias) and lin4 (one output) -
 *     contractions thin enough to be summed over the tile's columns in registers instead of travelling as 404 more rows.
 *     The record also carries the row sums of the adjoints' value columns = the bias gradients of lin1, lin3 and the
 *     folded bias of the skip layer.
 *   nphm_identity_train_weight_grads : contracts the stored operands over the columns: every chunk's share of the
 *     gradients of lin1 / lin2[:, :104] / lin3 (scaled back to the parameters' units) is WRITTEN to wpart
 *     (nphm_identity_train_wpart_bytes(n_chunks); chunk c of this call at wpart + c * nphm_identity_train_wpart_bytes(1)).
 *     chunks [n_chunks][4] = (weight set, first tile relative to its piece, number of tiles, piece): consecutive tiles of
 *     ONE weight set each (the host cuts the member-ordered tile table).
 *   nphm_identity_train_reduce_grads (ABI 8) : after ALL pieces - sums the edge records (n_tiles, every piece written at its
 *     tile's index) and the chunk shares (n_chunks, every piece's chunks at their index in the whole work list) in table
 *     order, no atomics: the parameter gradients are bitwise reproducible (round 3 added per-chunk partial sums with float
 *     atomics) - ADDING into grad_weight[l] (shape of lin<l>.weight; lin0: columns 0..2, lin2: columns 0..103 - the latent
 *     columns receive theirs through grad_b0 / grad_b2), grad_bias1/3/4 and grad_b0 / grad_b2 [n_rows,40,200] = dL/d(folded
 *     bias of lin0 / of the skip layer) per (row, member) as in nphm_identity_backward, and grad_anchors [n_rows,39,3] (the
 *     tiles' anchor terms, per pair in table order).  chunks = the work list of ALL pieces
 *     ordered by tile; ring_tiles = tiles per piece; set_chunk_first [sets + 1] = first chunk of every weight set; pair_first
 *     [40 * n_rows + 1] = first tile of every (member, row) pair in table order (pair = member * n_rows + row); scratch =
 *     n_chunks * nphm_identity_train_edge_bytes(1) bytes. */
size_t nphm_identity_train_saved_bytes(int n_tiles, int operands);
/* ABI 9: operand_scales [4 floats, device] of `operands` = 2 from grad_member_sdf [n_sdf] and grad_member_grad [n_grad] (NULL:
 * none) in ONE launch; work = 16 bytes of device memory, zero before the first call (the kernel leaves them zero).
 * ABI 11: `operand_scales` is an array of 8 words; word [7], read as unsigned, is IN/OUT of nphm_identity_train_backward with
 * `operands` = 2: += the number of wavefronts that clamped a stored operand to the binary16 range (the scales come from the
 * seeds' maxima through ratios measured on two weight sets; a step that clamps has biased lin1..lin3 gradients).  Zero it, run
 * the step's pieces, read it: not zero -> repeat the backward with `operands` = 0 (what the host module does). */
int nphm_identity_train_operand_scales(const float* grad_member_sdf, int64_t n_sdf, const float* grad_member_grad, int64_t n_grad,
                                       void* work, float* operand_scales, void* stream);
size_t nphm_identity_train_edge_bytes(int n_tiles);
/* HOST helper (no device work; ABI 8): the tables above from counts [40 * n_rows] = listed points of every (member, row) pair
 * (pair = member * n_rows + row, the order of the point list); member_set [40] = weight set of every member (non-decreasing),
 * ring_tiles = backward tiles per piece (<= 0: one piece), chunk_tiles = tiles per weight-gradient chunk.  Capacities the caller
 * provides: tiles_fwd 4 * (sum(counts) / 64 + pairs), tiles_bwd and chunks 4 * (sum(counts) / 32 + pairs) ints, set_chunk_first
 * [n_sets + 1], pair_first [pairs + 1]; sizes[4] = forward tiles, backward tiles, chunks, tiles per piece. */
int nphm_identity_train_tables(const long long* counts, int n_rows, const int* member_set, int n_sets, int ring_tiles, int chunk_tiles,
                               int* tiles_fwd, int* tiles_bwd, int* chunks, int* set_chunk_first, int* pair_first, int* sizes);
/* (ABI 10) The counts and the point list themselves, on the device, from blend_weights [n_rows, n_points, 40] (a pair lists the
 * points whose weight is > 0, nphm_identity_build_lists with NULL lists): counts [40 * n_rows] device ints, pair = member * n_rows +
 * row - the one array the host waits for; point_list (capacity n_rows * n_points * 40 ints; used: sum(counts)) ordered by (member,
 * row, point), which runs while the host builds the tables.  Two launches for torch.nonzero + bincount (~25 launches, two syncs). */
int nphm_identity_train_pair_counts(const float* blend_weights, int n_rows, int64_t n_points, int* counts, void* stream);
int nphm_identity_train_point_list(const float* blend_weights, int n_rows, int64_t n_points, const int* counts, int* point_list, void* stream);
int nphm_identity_train_forward(const void* packed, const void* packed_bwd, const void* latent_state, const float* xyz,
                                int64_t n_points, const int* tiles, int n_tiles, const int* point_list,
                                float* member_sdf, float* member_grad, void* stream);
int nphm_identity_train_backward(const void* packed, const void* packed_bwd, const void* latent_state, const float* xyz,
                                 int64_t n_points, const int* tiles, int n_tiles, const int* point_list,
                                 const float* grad_member_sdf, const float* grad_member_grad,
                                 float* grad_xyz, void* saved, void* edge, int operands, const float* operand_scales, void* stream);
int nphm_identity_train_weight_grads(const void* saved, int operands, const float* operand_scales, const int* chunks, int n_chunks,
                                     void* wpart, void* stream);
size_t nphm_identity_train_wpart_bytes(int n_chunks);
int nphm_identity_train_reduce_grads(const void* edge, int n_tiles, const void* wpart, const int* chunks, int n_chunks,
                                     int ring_tiles, const int* set_chunk_first, const int* pair_first, int n_rows, void* scratch,
                                     float* const grad_weight[5], float* grad_bias1, float* grad_bias3, float* grad_bias4,
                                     float* grad_b0, float* grad_b2, float* grad_anchors, void* stream);

/* The Gaussian blend of the training tier WITH its spatial gradient, for callers that need both (compute_loss:
 * decoder(...) followed by gradient(pred, x), loss_functions.py:36-49) without a graph-recording backward pass:
 *   pred [n_rows,n_points] = sum_k what_k S_k,  grad [n_rows,n_points,3] = d pred / d xyz
 * from the member values / gradients of nphm_identity_train_forward (EnsembledDeepSDF.py:129-150 for the weights; the
 * anchors are the second forward output).  The backward takes dL/dpred and dL/dgrad (NULL: zero), WRITES
 * grad_member_sdf / grad_member_grad (the seeds of nphm_identity_train_backward) and ACCUMULATES the blend's own
 * terms into grad_xyz (one thread per point: plain adds) / grad_anchors (ABI 8: per-block sums into anchor_partials
 * [nphm_identity_blend_partial_bytes(n_rows, n_points)], added per row in block order by a second small launch - no atomics). */
size_t nphm_identity_blend_partial_bytes(int n_rows, int64_t n_points);
int nphm_identity_blend_forward(const float* xyz, const float* anchors, const float* member_sdf, const float* member_grad,
                                int n_rows, int64_t n_points, float* pred, float* grad, void* stream);
int nphm_identity_blend_backward(const float* xyz, const float* anchors, const float* member_sdf, const float* member_grad,
                                 const float* grad_pred, const float* grad_grad, int n_rows, int64_t n_points,
                                 float* grad_member_sdf, float* grad_member_grad, float* grad_xyz, float* grad_anchors,
                                 void* anchor_partials, void* stream);

/* Second stage of the two-stage evaluation get_logits_backward (src/NPHM/models/reconstruction.py:28-56):
 * the identity field at displaced lattice points.  xyz_slab [(ix1-ix0)*ry*rz, 3] holds the canonical
 * points x + F_ex(x) of the slab in flattened lattice order (nphm_mlp_eval_grid with add_input);
 * traversal, output order and the chunk overwrite are those of nphm_identity_eval_grid. */
int nphm_identity_eval_grid_points(const void* packed, const void* latent_state,
                                   const float* xyz_slab, int rx, int ry, int rz, int ix0, int ix1,
                                   int64_t hack_chunk, float prune_tol, int precision,
                                   float* sdf_out, unsigned long long* stats,
                                   void* workspace, size_t workspace_bytes, void* stream);

/* ---- the non-field pieces of one latent-fitting step, fused (src/NPHM/models/fitting.py:99-167) -------------------
 * nphm_fit_loss: the loss terms of one step in ONE launch - clamped surface loss, mean |sdf| over the valid points below the
 *   device scalar `thr` (fitting.py:115-132); reg_expr = mean over the n_rows drawn observations of |z_ex|^2 (:136);
 *   reg_global / reg_loc / reg_unobserved / symm_dist of the identity code (:139-166) - and their weighted total.
 *   row [9] = surface, reg_expr, reg_global, reg_unobserved, reg_loc, symm_dist, total (lam in that order), number of valid points,
 *   the total once more (ABI 6: the caller's differentiable scalar next to the 8-entry report row).
 *   valid [n_points] bytes (torch.bool) or NULL; z_expr NULL for the identity-only loop (fitting.py:180-288).
 * nphm_fit_loss_backward: gradients of the total (times the device scalar *g_out, NULL = 1) w.r.t. sdf, the identity
 *   code [1344] and the expression codes [n_obs, expr_dim] (rows drawn several times collect every draw).
 * nphm_fit_root_backward: g_posed[p] = -J^-T[p] g_xc[p] - the implicit-function correction of the correspondence root
 *   (fitting.py:99-106) as one launch instead of an einsum chain.
 * nphm_identity_latent_grad: bias gradients of lin0 / the skip layer [n_rows,40,200] (nphm_identity_backward) -> gradient of
 *   the latent rows [n_rows,1344], through the latent columns of lin0.weight [24,200,99] / lin2.weight [24,200,200]
 *   (what the cond tensor of EnsembledDeepSDF.py:247-255 transports). */
/* Small dense heads with FROZEN weights on a handful of rows - mlp_pos (EnsembledDeepSDF.py:194-200: 64 -> 256 -> 256 -> 117,
 * ReLU between the layers) and the compressor of the deformation field (deepSDF.py:212-223: 1461 -> 32): y = W_n(relu(...)) + b_n
 * in one launch, and the gradient w.r.t. the input in one launch (nn.Linear layout: weight [out, in]).  dims = {in, ..., out},
 * 1..3 layers, widths <= 1536; hidden [n_rows, dims[1] + dims[2]] receives the post-ReLU activations the backward needs.
 * ABI 6: the input is the first dims[0] columns of rows x_stride floats apart (the global part of a latent row, no slice copy), g_x
 * has the same row stride and is written in full (zeros beyond dims[0]); y_add [out] or NULL is added to every output row (the mean
 * anchors).  ABI 10: g_y_other [n_rows, out] or NULL is a second gradient of y (another use of the same output - the anchors feed
 * the identity field AND the deformation field's compressor), added while loading instead of by an add launch. */
int nphm_head_forward(const float* const weight[3], const float* const bias[3], const int dims[4], int n_layers, const float* x,
                      int x_stride, const float* y_add, int n_rows, float* y, float* hidden, void* stream);
int nphm_head_backward(const float* const weight[3], const float* const bias[3], const int dims[4], int n_layers, const float* hidden,
                       const float* g_y, const float* g_y_other, int n_rows, float* g_x, int x_stride, void* stream);
/* (ABI 10) The conditioning rows of the forward-deformation field in 'compress' mode when ONE identity code serves every row
 * (deepSDF.py:212-223 with the latent of the fitting loops, fitting.py:83-85): cond [n_rows, out_dim + expr_dim] =
 * [compressor([code | anchors]) on every row | z_ex[b]] in one launch - the input row is read from its two parts (no cat), the
 * compressor runs once (the reference runs it per batch row on identical inputs).  Backward: g_code [code_dim], g_anchors
 * [anchors_dim] from the first out_dim columns of g_cond summed over the rows in row order; the gradient of z_ex is the remaining
 * columns of g_cond as they are. */
int nphm_compress_condition(const float* weight, const float* bias, const float* code, int code_dim, const float* anchors, int anchors_dim,
                            int out_dim, const float* z_ex, int n_rows, int expr_dim, float* cond, void* stream);
int nphm_compress_condition_backward(const float* weight, const float* bias, int code_dim, int anchors_dim, int out_dim, const float* g_cond,
                                     int n_rows, int expr_dim, float* g_code, float* g_anchors, void* stream);
int nphm_fit_loss(const float* sdf, const unsigned char* valid, int64_t n_points, const float* thr, const float* lam,
                  const float* z_shape, const float* z_expr, const int64_t* obs_idx, int n_rows, int n_obs, int expr_dim,
                  float* row, void* stream);
int nphm_fit_loss_backward(const float* sdf, const unsigned char* valid, int64_t n_points, const float* thr, const float* lam,
                           const float* z_shape, const float* z_expr, const int64_t* obs_idx, int n_rows, int n_obs, int expr_dim,
                           const float* g_out, float* g_sdf, float* g_shape, float* g_expr, void* stream);
/* (ABI 10) forward and backward of the loss in ONE launch, for the step that seeds loss.backward() right behind the forward
 * (fitting.py:168-169): row as nphm_fit_loss, the gradients as nphm_fit_loss_backward with the seed *g_out (NULL = 1). */
int nphm_fit_loss_with_gradients(const float* sdf, const unsigned char* valid, int64_t n_points, const float* thr, const float* lam,
                                 const float* z_shape, const float* z_expr, const int64_t* obs_idx, int n_rows, int n_obs, int expr_dim,
                                 const float* g_out, float* row, float* g_sdf, float* g_shape, float* g_expr, void* stream);
/* (ABI 12) the same for a step replayed from a hipGraph beside nphm_fit_inputs_ring: the row is also stored at
 * row_log[log_control[0] - 1] ([log_rows][8]; log_control = the ring's control words, whose first counts the replays including
 * this one) - the loss trace of a replayed loop without a copy launch between two replays. */
int nphm_fit_loss_with_gradients_logged(const float* sdf, const unsigned char* valid, int64_t n_points, const float* thr, const float* lam,
                                        const float* z_shape, const float* z_expr, const int64_t* obs_idx, int n_rows, int n_obs, int expr_dim,
                                        const float* g_out, float* row, float* g_sdf, float* g_shape, float* g_expr,
                                        float* row_log, const unsigned* log_control, int log_rows, void* stream);
int nphm_fit_root_backward(const float* jac_inverse, const float* g_xc, float* g_posed, int64_t n, void* stream);
/* (ABI 10) The inputs of one fitting step from its draw (fitting.py:61-85) in one launch: drawn = [n_rows observation indices |
 * n_rows x n_points point indices] (int64); obs [n_rows, n_points, cloud_width] = clouds[o_b][p_br] (clouds [n_obs, cloud_points,
 * cloud_width], padded); z_ex [n_rows, expr_dim] = z_expr_table[o_b]; glob_cond [n_rows, shape_dim + expr_dim] = [z_shape | z_ex[b]].
 * nphm_fit_inputs_backward: g_z_expr_table [n_obs, expr_dim] = g_table_use (the gradient of another use of the table, e.g. the
 * regulariser's; may be NULL) + per row the sum over its draws, in draw order, of g_z_ex[b] (rows z_ex_row_stride floats apart; may
 * be NULL) + the expression columns of g_glob_cond[b] (may be NULL); g_z_shape [shape_dim] (NULL: not wanted) = the sum of
 * g_shape_uses[0..3] ([shape_dim] each, NULL entries skipped: the gradients of the identity code's other uses in the step - anchor
 * head, field, regularisers, compressor - added here in this order instead of by one add launch each) + the sum over b of the
 * identity columns of g_glob_cond. */
int nphm_fit_inputs(const int64_t* drawn, int n_rows, int n_points, const float* clouds, int n_obs, int cloud_points, int cloud_width,
                    const float* z_shape, int shape_dim, const float* z_expr_table, int expr_dim, float* obs, float* z_ex,
                    float* glob_cond, void* stream);
/* (ABI 12) nphm_fit_inputs for a step that is REPLAYED from a hipGraph: the draw is read from a ring of draws in pinned host memory
 * (host_ring: ring_slots slots, slot_stride int64 apart, n_total int64 each = the draw and whatever rides behind it, e.g. the
 * optimizer scalars of nphm_adam_step_pair) instead of from a device buffer some stream-ordered upload filled - a copy-engine
 * hand-over and two queue gaps per step (35 us of a 725 us step).  control: two zero-initialised uint32 in device memory;
 * control[0] counts the ring launches that have completed, the launch reads slot control[0] % ring_slots and copies its
 * n_total values to drawn_out (device) for the later launches of the step.  The host fills slot (launches issued so far) %
 * ring_slots before it issues the launch and must not run more than ring_slots launches ahead of the device. */
int nphm_fit_inputs_ring(const int64_t* host_ring, int ring_slots, int64_t slot_stride, int64_t n_total, unsigned* control, int64_t* drawn_out,
                         int n_rows, int n_points, const float* clouds, int n_obs, int cloud_points, int cloud_width,
                         const float* z_shape, int shape_dim, const float* z_expr_table, int expr_dim, float* obs, float* z_ex,
                         float* glob_cond, void* stream);
int nphm_fit_inputs_backward(const float* g_z_ex, int64_t z_ex_row_stride, const float* g_glob_cond, const int64_t* obs_idx, int n_rows,
                             int n_obs, int shape_dim, int expr_dim, const float* const g_shape_uses[4], const float* g_table_use,
                             float* g_z_expr_table, float* g_z_shape, void* stream);
/* sdf [n_points] = sum over the kept members of blend_weights * member_values (both [n_points, 40]; weights exactly 0 where
 * the pruning rule dropped the member - nphm_identity_build_lists - and member_values is not read there: the output of
 * nphm_identity_member_forward needs no zero-fill).  The blend of EnsembledDeepSDF.py:129-150 on the autograd tier (ABI 6). */
/* Backward of out[b] = table[idx[b]] (fitting.py:83, the expression codes of the drawn observations): g_table [n_rows, width]
 * = per row the sum of g_out [n_draws, width] over its draws, in draw order (deterministic; every row written) (ABI 6). */
int nphm_gather_rows_backward(const float* g_out, const int64_t* idx, int n_draws, int n_rows, int width, float* g_table,
                              void* stream);
/* One Adam step (torch.optim.Adam defaults: no weight decay / amsgrad) of a contiguous fp32 tensor in one launch:
 * m += (1 - beta1)(g - m); v = beta2 v + (1 - beta2) g^2; p -= step_size * m / (sqrt(v) / bias_correction2_sqrt + eps), with
 * step_size = lr / (1 - beta1^t) and bias_correction2_sqrt = sqrt(1 - beta2^t) computed by the caller - the optimizers of the
 * latent codes in fitting.py:47-48 / :196 (ABI 6). */
int nphm_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float beta1, float beta2,
                   float step_size, float bias_correction2_sqrt, float eps, void* stream);
/* (ABI 10) The same update of up to two tensors (n[q] = 0: absent) in one launch with the scalars in DEVICE memory: scalars [2][6] =
 * {1 - beta1, beta2, 1 - beta2, step_size, bias_correction2_sqrt, eps} per tensor - the optimizer steps of the two codes inside
 * the replayed graph of a fitting step (the values change every step and travel with the step's draw). */
int nphm_adam_step_pair(float* const param[2], const float* const grad[2], float* const exp_avg[2], float* const exp_avg_sq[2],
                        const int64_t n[2], const float* scalars, void* stream);
int nphm_identity_blend_members(const float* blend_weights, const float* member_values, int64_t n_points, float* sdf, void* stream);
/* (ABI 7: `scratch` of nphm_identity_latent_grad_scratch_bytes(n_rows) bytes holds the members' shares of the 64 global columns;
 * they are summed in member order - every element of g_lat is written once, bitwise reproducible, nothing to zero beforehand) */
size_t nphm_identity_latent_grad_scratch_bytes(int n_rows);
int nphm_identity_latent_grad(const float* lin0_weight, const float* lin2_weight, const float* g_bias0, const float* g_bias2,
                              int n_rows, float* g_lat, void* scratch, void* stream);

/* ---- loss terms of the identity decoder's training step (ABI 6) ------------------------- */
/* src/NPHM/models/loss_functions.py:51-110 after the decoder evaluations, on the batched layout of the mirrored
 * compute_loss: sdf [n_rows, N], grad = d sdf / d x [n_rows, N, 3] with N = sizes[0..3] = face | non-face | near-surface |
 * far points as consecutive slices, normals [n_rows, sizes[0] + sizes[1], 3], z [n_rows, lat_dim] the latent codes,
 * anchors / anchors_gt [n_rows, n_anchors, 3] (both NULL for a decoder without anchors), layout = (glob, loc, n_symm,
 * n_middle_pairs) of the code [glob | 2 n_symm local codes | middle codes | ...] ((g, 0, 0, 0): no local codes).
 *   nphm_train_loss          : row [8] = surf_sdf, normals, space_sdf, grad, lat_reg, anchors, symm_dist, middle_dist - one
 *     launch (per-block partial sums in `partial` [nphm_train_loss_blocks() * 8], combined in block order by the last block:
 *     deterministic; `counter` is a zero-initialised device word the kernel resets).
 *   nphm_train_loss_backward : g_terms [8] = d L / d term (device) -> gradients w.r.t. sdf, grad, z and anchors - one launch. */
int nphm_train_loss_blocks(void);
int nphm_train_loss(const float* sdf, const float* grad, const float* normals, const float* z, const float* anchors,
                    const float* anchors_gt, int n_rows, const int sizes[4], int lat_dim, int n_anchors, const int layout[4],
                    float* partial, unsigned* counter, float* row, void* stream);
int nphm_train_loss_backward(const float* sdf, const float* grad, const float* normals, const float* z, const float* anchors,
                             const float* anchors_gt, int n_rows, const int sizes[4], int lat_dim, int n_anchors,
                             const int layout[4], const float* g_terms, float* g_sdf, float* g_grad, float* g_z,
                             float* g_anchors, void* stream);

/* ---- dense skip-MLP: DeepSDF (NPM global SDF, backbone of DeformationNetwork) ----------- */
/* Architecture (src/NPHM/models/deepSDF.py:7-62): dims = [3 + lat_dim] + [hidden_dim]*nlayers + [out_dim],
 * input re-injected (concat, / sqrt 2) before layer nlayers/2, Softplus(beta) activations.
 * The fused kernel covers input_dim 3, beta 100, no positional encoding (num_freq_bands 0),
 * 32 <= hidden_dim <= 1024, hidden_dim > 3 + lat_dim, 2 <= nlayers <= 11, out_dim <= 4 — i.e. the NPM
 * net (npm.yaml: 512/1024/8/1) and the NPHM deformation backbone (nphm_def.yaml: 232/512/6/3).
 * Returns 1 if covered. */
int nphm_mlp_supported(int lat_dim, int hidden_dim, int nlayers, int out_dim, int input_dim, float beta,
                       int num_freq_bands);
size_t nphm_mlp_packed_bytes(int lat_dim, int hidden_dim, int nlayers, int out_dim);
size_t nphm_mlp_latent_state_bytes(int lat_dim, int hidden_dim, int nlayers, int out_dim, int n_rows);

/* Re-lay the state_dict tensors lin{0..nlayers}.{weight,bias} (arrays of nlayers+1 device pointers)
 * into MFMA fragment order, once as split-bf16 and once as split-f16 halves (ABI 7: both live in `packed`, and both
 * forms of the folded fragments in every latent-state row; replaces the per-call nn.Linear GEMMs of deepSDF.py:76-88). */
int nphm_mlp_pack(int lat_dim, int hidden_dim, int nlayers, int out_dim,
                  const float* const* lin_weight, const float* const* lin_bias, void* packed, void* stream);

/* Per conditioning vector (one per batch row; cond_rows [n_rows, lat_dim] = lat_rep[:, 0, :] of
 * DeepSDF.forward, or [compressor(z_id, anchors) | z_ex] of DeformationNetwork mode 'compress',
 * deepSDF.py:212-223): the latent columns of lin0 and of the skip layer folded into bias vectors
 * (replaces torch.cat([xyz, lat_rep]) per point, deepSDF.py:75, :82). */
int nphm_mlp_prepare_latent(int lat_dim, int hidden_dim, int nlayers, int out_dim,
                            const float* const* lin_weight, const float* const* lin_bias,
                            const float* cond_rows, int n_rows, void* latent_state, void* stream);

/* `numerics` of the plain evaluation entry points (ABI 7): operand format | NPHM_MLP_TWO_PASS(mask).
 *   NPHM_MLP_BF16X3  x w = xh wh + xl wh + xh wl on bf16 halves (16 product bits; what every other MLP entry point runs)
 *   NPHM_MLP_F16X3   the same on IEEE binary16 halves (22 product bits, v_mfma_f32_32x32x16_f16, same rate)
 *   NPHM_MLP_TWO_PASS(mask): bit l of mask = linear layer l (1 <= l <= nlayers - 1: the hidden GEMM layers) drops the
 *   xh wl term, i.e. runs on weights rounded to the half format - two MFMAs instead of three and half the weight bytes;
 *   a systematic 2^-12 (f16) / 2^-9 (bf16) perturbation of that layer's weights.  Which layers may is a property of the
 *   checkpoint: the host module measures it against the three-term product (nphm_amd/deepsdf.py, calibrate).
 *   NPHM_MLP_ONE_PASS(mask) (ABI 9; NPHM_MLP_F16X3, nphm_mlp_eval_points / nphm_mlp_eval_grid only): bit l = hidden GEMM
 *   layer l runs the SINGLE-term product rn(x) wh - one MFMA per tile and K-step, no read of the activations' lo plane; all
 *   hidden layers: a variant without the lo plane that holds twice the points per workgroup (128 at hidden_dim <= 512), i.e.
 *   half the weight bytes per point (all but the LAST hidden layer, that one in NPHM_MLP_TWO_PASS: the same workgroup shape, the
 *   last hidden layer in two point halves).  Error ~2^-12 relative per product on both operands, random in the activations: again
 *   measured per checkpoint against the three-term product before it is used (nphm_amd/deepsdf.py, calibrate_numerics).
 * The value+Jacobian, Broyden and saving entry points below take format and two-pass mask (hidden_dim <= 512: both formats;
 * the 1024-wide variant runs them on bf16 halves only).
 * The two value+Jacobian entry points also take a point RANGE and a workgroup width (ABI 8): the launch covers points
 * [point_base, point_base + point_count) of every row (0, 0 = all; point_base a multiple of 16, of 64 for the saving form)
 * with `columns` = 64 (16 points per workgroup; 0 = default) or 32 (8 points per workgroup, hidden_dim <= 512) - a batch whose
 * 16-point workgroups fill 1.2 rounds of the chip runs as one full round of them plus a round of 8-point workgroups over the
 * rest (nphm_amd/deepsdf.py: _jvp_split). */
#define NPHM_MLP_BF16X3 0
#define NPHM_MLP_F16X3 1
#define NPHM_MLP_TWO_PASS(mask) ((int)(((unsigned)(mask) & 0xfffu) << 8))
#define NPHM_MLP_ONE_PASS(mask) ((int)(((unsigned)(mask) & 0x7ffu) << 20))

/* DeepSDF.forward (deepSDF.py:64-89) for row-constant latents: out[b,n,:out_dim] for xyz[b,n,:3].
 * add_input != 0 adds xyz to the first 3 outputs: canonical points x + F_ex(x) of get_logits_backward
 * (src/NPHM/models/reconstruction.py:44-46) / posed vertices of deform_mesh (:83-84). */
int nphm_mlp_eval_points(int lat_dim, int hidden_dim, int nlayers, int out_dim,
                         const void* packed, const void* latent_state,
                         const float* xyz, int n_rows, int64_t n_points, int add_input, int numerics,
                         float* out, void* stream);

/* The two plain evaluation entry points with a workspace (ABI 12; nphm_mlp_eval_workspace_bytes() bytes of device memory,
 * 16-byte aligned, owned by the launch's stream while it runs; contents are scratch, nothing persists between calls).  It serves
 * ONE combination of `numerics`: every hidden layer in NPHM_MLP_ONE_PASS but the last, that one in NPHM_MLP_TWO_PASS (what the
 * calibration finds for the NPM SDF, deepSDF.py:6-89 with scripts/configs/npm.yaml:1-4).  Without a workspace that layer runs in
 * two point halves and streams its weights twice; with one it runs in two K halves - the second half of its operands (128 KiB
 * per workgroup) waits in a slot of the workspace, which stays in the last-level caches - and streams them once.  Results are
 * bitwise identical to workspace = NULL; every other `numerics` ignores the workspace. */
size_t nphm_mlp_eval_workspace_bytes(void);
int nphm_mlp_eval_points_ws(int lat_dim, int hidden_dim, int nlayers, int out_dim,
                            const void* packed, const void* latent_state,
                            const float* xyz, int n_rows, int64_t n_points, int add_input, int numerics,
                            float* out, void* workspace, size_t workspace_bytes, void* stream);

/* Value AND spatial Jacobian in one launch (forward-mode, tangents carried through the same GEMMs):
 * out[b,n,0,:] = f(x), out[b,n,1+c,i] = d f_i / d x_c.  With add_input the identity is added too:
 * out[:,:,0,:] = x + F(x), out[:,:,1+c,:] = d (x + F) / d x_c — the analytic form of
 * jac(decoder_expr, xc, ...) (src/NPHM/models/diff_operators.py:26-54: 1 forward + 3 autograd VJPs),
 * used twice per fitting step (iterative_root_finding.py:123, fitting.py:101).
 * out [n_rows, n_points, 4, out_dim].  (ABI 10) jac_inverse [n_rows, n_points, 3, 3] or NULL: the inverse of the matrix
 * M[i][c] = out[.., 1 + c, i] (i, c < 3; out_dim >= 3) - the `.inverse()` both callers apply to the Jacobian - written by the same
 * launch (the adjugate formula of nphm_inverse3x3 on the values written to `out`). */
int nphm_mlp_eval_points_jvp(int lat_dim, int hidden_dim, int nlayers, int out_dim,
                             const void* packed, const void* latent_state,
                             const float* xyz, int n_rows, int64_t n_points, int add_input,
                             float* out, int numerics, int64_t point_base, int64_t point_count, int columns, float* jac_inverse,
                             void* stream);

/* Correspondence search of the fitting loop in ONE launch: Broyden root finding of
 * x + F(x) = obs per point (src/NPHM/models/iterative_root_finding.py:5-71 broyden as called by
 * search :152-156 — there <= 16 forwards of the field with a host sync each).  Same per-point state
 * machine: every point takes the first update, then moves while its best residual norm is > cvg_thresh
 * and its current one < dvg_thresh; the inverse Jacobian gets the rank-one ("good Broyden") update with
 * the +-eps guard on the denominator; x_out is the final iterate, diff_out the smallest residual norm
 * seen, valid_out = diff_out < cvg_thresh.  x_init / obs / x_out [n_rows, n_points, 3], jinv_init
 * [n_rows, n_points, 3, 3] (inverse of nphm_mlp_eval_points_jvp's Jacobian), out_dim >= 3. */
int nphm_mlp_broyden(int lat_dim, int hidden_dim, int nlayers, int out_dim,
                     const void* packed, const void* latent_state,
                     const float* obs, const float* x_init, const float* jinv_init, int n_rows, int64_t n_points,
                     int max_steps, float cvg_thresh, float dvg_thresh, float eps,
                     float* x_out, float* diff_out, unsigned char* valid_out, int numerics, void* stream);
/* The same solve when the caller already holds x_init + F(x_init) - the value stream of the nphm_mlp_eval_points_jvp launch
 * that produced jinv_init (search evaluates the Jacobian at the start points, iterative_root_finding.py:118): posed_init
 * [n_rows, n_points] 3-vectors, posed_stride floats apart (4 * out_dim inside the value+Jacobian output).  The residual of
 * iteration 0 is read instead of evaluated: one pass of the network less per solve, same iterates (ABI 6). */
int nphm_mlp_broyden_from(int lat_dim, int hidden_dim, int nlayers, int out_dim,
                          const void* packed, const void* latent_state,
                          const float* obs, const float* x_init, const float* jinv_init, const float* posed_init,
                          int64_t posed_stride, int n_rows, int64_t n_points,
                          int max_steps, float cvg_thresh, float dvg_thresh, float eps,
                          float* x_out, float* diff_out, unsigned char* valid_out, int numerics, void* stream);

/* First-order backward of the skip-MLP with respect to its conditioning rows, for the fitting loop's
 * loss.backward() through decoder_expr(p_corresp, cond) (src/NPHM/models/fitting.py:99-106 with the decoders
 * frozen and the query points detached).  Covers the hidden <= 512, out_dim <= 3 variant (the deformation
 * backbone); *_bytes return 0 otherwise.
 *   nphm_mlp_eval_points_saving : nphm_mlp_eval_points that also leaves sigma'(d_l) of every hidden layer in
 *     `saved` (nphm_mlp_saved_bytes(..., n_rows, n_points) bytes);
 *   nphm_mlp_pack_bwd           : transposed split-bf16 pack of lin1..lin_last (nphm_mlp_bwd_packed_bytes);
 *   nphm_mlp_backward_cond      : grad_out [n_rows, n_points, out_dim] -> bias gradients of lin0 and of the skip
 *     layer as per-slot sums bias_partials [n_rows][ceil(n_points / 32)][2][hidden_dim] (nphm_mlp_bwd_partial_bytes; ABI 8:
 *     every slot is WRITTEN - no zero fill, no atomics); their sums over the slots of a row, g0 and gs, map onto the
 *     conditioning as d L / d cond = g0 W0[:, 3:] + gs W_skip[:, K+3:] / sqrt(2) (nphm_mlp_cond_grad adds the slots in
 *     order: bitwise reproducible).  (ABI 10) root_jac_inverse [n_rows, n_points, 3, 3] or NULL: grad_out is then the gradient of
 *     the implicit root x_c = root - J^-1 (F(root) - F(root).detach()) (fitting.py:99-106) and the kernel applies -J^-T to it while
 *     loading (out_dim = 3) - nphm_fit_root_backward's launch, folded in. */
size_t nphm_mlp_saved_bytes(int lat_dim, int hidden_dim, int nlayers, int out_dim, int n_rows, int64_t n_points);
int nphm_mlp_eval_points_saving(int lat_dim, int hidden_dim, int nlayers, int out_dim,
                                const void* packed, const void* latent_state,
                                const float* xyz, int n_rows, int64_t n_points, int add_input,
                                float* out, void* saved, int numerics, void* stream);
/* nphm_mlp_eval_points_jvp that also leaves sigma' of the VALUE stream in `saved` (same layout and size as
 * nphm_mlp_eval_points_saving): posed points, their Jacobian and the state of the backward in one launch - what the
 * fitting step needs at the canonical correspondences (fitting.py:99-103).  (ABI 10) jac_inverse: as nphm_mlp_eval_points_jvp. */
int nphm_mlp_eval_points_jvp_saving(int lat_dim, int hidden_dim, int nlayers, int out_dim,
                                    const void* packed, const void* latent_state,
                                    const float* xyz, int n_rows, int64_t n_points, int add_input,
                                    float* out, void* saved, int numerics, int64_t point_base, int64_t point_count, int columns,
                                    float* jac_inverse, void* stream);
size_t nphm_mlp_bwd_packed_bytes(int lat_dim, int hidden_dim, int nlayers, int out_dim);
int nphm_mlp_pack_bwd(int lat_dim, int hidden_dim, int nlayers, int out_dim, const float* const* lin_weight,
                      void* packed_bwd, void* stream);
size_t nphm_mlp_bwd_partial_bytes(int hidden_dim, int n_rows, int64_t n_points);
int nphm_mlp_backward_cond(int lat_dim, int hidden_dim, int nlayers, int out_dim, const void* packed_bwd,
                           const void* saved, const float* grad_out, const float* root_jac_inverse, int n_rows, int64_t n_points,
                           void* bias_partials, void* stream);

/* Batched inverse of n row-major 3x3 matrices (adjugate formula, one thread each): the `.inverse()` calls on
 * the deformation Jacobians in the correspondence search and the implicit differentiation of the fitting loop
 * (src/NPHM/models/iterative_root_finding.py:118, src/NPHM/models/fitting.py:102) without the blocking
 * error-flag read of torch.linalg.inv. */
int nphm_inverse3x3(const float* matrices, float* inverses, int64_t n, void* stream);
/* The same for matrices addressed by strides (in floats): element (i, j) of matrix p at matrices[p * matrix_stride +
 * i * row_stride + j * col_stride] - the Jacobian block of nphm_mlp_eval_points_jvp's [n, 4, out_dim] output, transposed
 * (diff_operators.jac's layout), is inverted where it lies; inverses row-major [n, 3, 3] (ABI 6). */
int nphm_inverse3x3_strided(const float* matrices, int64_t matrix_stride, int64_t row_stride, int64_t col_stride,
                            float* inverses, int64_t n, void* stream);
/* d L / d cond [n_rows, lat_dim] = g0 W0[:, off0 : off0 + lat_dim] + gs W_skip[:, off_skip : off_skip + lat_dim] / sqrt(2)
 * from nphm_mlp_backward_cond's per-slot sums (g0, gs = their sums over the ceil(n_points / 32) slots of a row, in slot
 * order) and the row-major fp32 weights of lin0 / the skip layer (leading dimensions ld0 / ld_skip = their in_features) in
 * one launch (ABI 6; ABI 8: the partials instead of zero-filled accumulators; hidden_dim <= 512). */
int nphm_mlp_cond_grad(const void* bias_partials, int64_t n_points, int n_rows, int hidden_dim,
                       const float* lin0_weight, int ld0, int off0, const float* skip_weight, int ld_skip, int off_skip,
                       int lat_dim, float* grad_cond, void* stream);

/* The same on the x-slab [ix0, ix1) of an [rx,ry,rz] 'ij' lattice (utils/reconstruction.py:5-20):
 * out [(ix1-ix0)*ry*rz, out_dim] in flattened lattice order. */
int nphm_mlp_eval_grid(int lat_dim, int hidden_dim, int nlayers, int out_dim,
                       const void* packed, const void* latent_state,
                       const float* axis_x, const float* axis_y, const float* axis_z,
                       int rx, int ry, int rz, int ix0, int ix1, int add_input, int numerics,
                       float* out, void* stream);
int nphm_mlp_eval_grid_ws(int lat_dim, int hidden_dim, int nlayers, int out_dim,
                          const void* packed, const void* latent_state,
                          const float* axis_x, const float* axis_y, const float* axis_z,
                          int rx, int ry, int rz, int ix0, int ix1, int add_input, int numerics,
                          float* out, void* workspace, size_t workspace_bytes, void* stream);

/* ---- host side: iso-surface extraction for mesh_from_logits ------------------------------- */
/* Marching cubes on a HOST fp32 volume [nx,ny,nz] ('ij' order) — the step the reference delegates to
 * the third-party PyMCubes: mcubes.marching_cubes(-logits, 0.0) (src/NPHM/utils/reconstruction.py:25-30).
 * negate != 0 extracts on -volume (the reference negates the SDF so that inside is positive);
 * vertices come in index space (x,y,z) = (i,j,k) as float64, shared between triangles, normals
 * pointing from field > iso to field < iso; n_threads <= 0 = min(16, host cores); the output does
 * not depend on the thread count.  Two-step: extract (sizes), fetch (copy out), free. */
int nphm_mc_extract(const float* volume, int nx, int ny, int nz, double iso, int negate, int n_threads,
                    void** handle, int64_t* n_verts, int64_t* n_faces);
int nphm_mc_fetch(void* handle, double* verts, int64_t* faces);
void nphm_mc_free(void* handle);

/* The same extraction on the GPU for a DEVICE-resident volume (no 4 B/voxel device->host copy, no host
 * pass): bit-identical vertices, triangles and ordering.  workspace: nphm_mc_device_workspace_bytes()
 * device bytes.  nphm_mc_device_count classifies, scans and returns the mesh size (it synchronises
 * `stream` once: the caller allocates verts [n_verts,3] float64 and faces [n_faces,3] int64 on the
 * device), nphm_mc_device_emit fills them (stream-ordered). */
size_t nphm_mc_device_workspace_bytes(int nx, int ny, int nz);
int nphm_mc_device_count(const float* volume, int nx, int ny, int nz, double iso, int negate, void* workspace,
                         int64_t* n_verts, int64_t* n_faces, void* stream);
int nphm_mc_device_emit(const float* volume, int nx, int ny, int nz, double iso, int negate, void* workspace,
                        double* verts, int64_t* faces, void* stream);

/* ---- dense skip-MLP with trainable parameters (ABI 11; csrc/dense_train_kernels.hip) ---------------------------------------
 * Replaces, inside the reference's second training stage (scripts/training/train_corresp.py -> compute_loss_corresp_forward,
 * src/NPHM/models/loss_functions.py:282-326), the nn.Linear + Softplus pairs of DeepSDF.forward (src/NPHM/models/deepSDF.py:
 * 64-89) and what autograd derives from them: per hidden layer one launch forward, three backward.
 *   nphm_dense_gemm_nt : C[M,N] = alpha A[M,K] B[N,K]^T (fp32 in HBM, rows lda / ldb / ldc floats apart; split-bf16 x3 on the
 *     matrix pipe, fp32-equivalent), epilogue 0: nothing, 1: C = act(C + E[m / e_rows]), act = Softplus(beta) (beta <= 0: ReLU),
 *     2: C = C + E[m / e_rows]; E [ceil(M / e_rows), N] contiguous.  k_splits > 1 (epilogue 0 only): the K range is cut into
 *     k_splits pieces whose products are WRITTEN to C[z][M][ldc] - nphm_dense_reduce_splits adds them in order (weight
 *     gradients: K = the point axis; no atomics).
 *       forward y = gemm(x, W, e, epilogue 1);  dx = gemm(gp, W^T);  dW = reduce(gemm(gp^T, x^T, k_splits))
 *   nphm_dense_gpre : gp[n][c] = g[n][c] act'(y[n][c]) with act' from the activation's output (Softplus: 1 - exp(-beta y)); y NULL:
 *     gp = g.  Writes gp (NULL: not) and / or its transpose gp_t[c][n] (rows ld_t floats apart; NULL: not) - also the
 *     transpose kernel of x and W - and (column_sums not NULL) the column sums of every 32-row tile, [ceil(n / 32), c].
 *   nphm_dense_column_sums : out[c] = sum over the rows of x [rows, c] in a fixed order (bias gradients: the tiles' sums). */
int nphm_dense_gemm_nt(const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N, int K,
                       const float* E, int e_rows, float alpha, float beta, int epilogue, int k_splits, void* stream);
int nphm_dense_reduce_splits(const float* parts, int k_splits, int64_t count, float scale, float* out, void* stream);
int nphm_dense_gpre(const float* g, const float* y, int n, int c, float beta, float* gp, float* gp_t, int ld_t, float* column_sums,
                    void* stream);
int nphm_dense_column_sums(const float* x, int rows, int c, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NPHM_AMD_H */

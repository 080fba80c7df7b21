/* gcnvae.h -- C ABI of libgcnvae_hip.so: the R-GCN-VAE link-prediction hot path of
 * karenyang/GCN-VAE as hand-written HIP kernels for gfx950 (MI355X).
 *
 * The reference has no FFI for this path: it is Python calling torch / DGL ops.  Each entry
 * point below therefore cites the reference op sequence it replaces (paths relative to
 * /root/reference/).  INTEGRATION.md shows the ctypes binding a maintainer would add.
 *
 * Conventions
 *   - every function returns 0 on success, a NEGATIVE gv_status for an argument error, or a
 *     POSITIVE hipError_t; nothing throws across the ABI; gv_last_error_string() explains.
 *   - all pointers are DEVICE pointers unless a name ends in _host; no function allocates,
 *     synchronises or copies to the host; everything is enqueued on the caller's stream
 *     (hipStream_t passed as void*), so calls are graph-capturable.
 *   - matrices are dense row-major fp32 with an explicit leading dimension (in floats);
 *     indices are int32.
 *   - "segment work items": a destination-sorted edge list is cut into chunks of at most
 *     `chunk` edges.  items[i] = {segment, edge_begin, edge_end, slot}: slot < 0 means the
 *     segment fits one item and the wave stores the final row itself; slot >= 0 means the wave
 *     stores a raw partial row into partial[slot] and a fix-up pass sums the segment's slots
 *     IN ORDER (fix[j] = {segment, first_slot, n_slots, 0}) -- no atomics, bitwise reproducible.
 *     gv_segment_items_* build these lists on the device from a CSR row pointer.
 */
#ifndef GCNVAE_H
#define GCNVAE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    GV_OK = 0,
    GV_ERR_NULL = -1,      /* required pointer is NULL */
    GV_ERR_SHAPE = -2,     /* inconsistent / unsupported sizes */
    GV_ERR_ALIGN = -3,     /* pointer or leading dimension breaks an alignment the call needs */
    GV_ERR_WORKSPACE = -4  /* workspace too small */
} gv_status;

enum { GV_ACT_NONE = 0, GV_ACT_RELU = 1 };

int gv_version(void);
const char* gv_last_error_string(void);

/* ---------------------------------------------------------------------------------------------
 * Segment work items (device-side builder; replaces nothing in the reference -- DGL hides it).
 * rowptr[n_seg+1] int32 CSR pointer.  Pass 1 counts, pass 2 fills (the caller scans in between).
 *   gv_segment_items_count: n_chunks[s] = max(1, ceil(deg/chunk)); n_slots[s] = n_chunks>1 ? n_chunks : 0
 *   gv_segment_items_fill : item_off / slot_off / fix_off are EXCLUSIVE scans of n_chunks / n_slots /
 *                           (n_chunks>1), each with n_seg+1 entries.
 * The K1 entry points ignore list entries whose first field is negative, so a caller that does not want to read the
 * totals back (no host synchronisation per mini-batch) may size both lists by their upper bounds
 * (items <= n_seg + E/chunk, fix <= min(n_seg, E/chunk), slots <= 2*E/chunk), pre-fill them with -1 and pass the
 * bounds as n_items / n_fix.
 */
int gv_segment_items_count(const int32_t* rowptr, int n_seg, int chunk, int32_t* n_chunks, int32_t* n_slots,
                           int32_t* is_split, void* stream);
int gv_segment_items_fill(const int32_t* rowptr, int n_seg, int chunk, const int32_t* item_off,
                          const int32_t* slot_off, const int32_t* fix_off, int32_t* items /*[n_items][4]*/,
                          int32_t* fix /*[n_fix][4]*/, void* stream);

/* ---------------------------------------------------------------------------------------------
 * K1  R-GCN block-diagonal ("bdd") relational aggregation.
 * Replaces DGL RelGraphConv.bdd_message_func + update_all(fn.sum) + bias/self-loop/activation/dropout
 * (call sites kgvae/model.py:54-59, :110-111, :209-211; semantics oracle/rgcn.py):
 *   agg[v, b*so+j] = sum_{e: dst_e = v} norm_e * sum_i x[src_e, b*si+i] * W[etype_e, b, i, j]
 *   out[v]         = keep[v] * keep_scale * act(agg[v] + addend[v])       (each part optional)
 * addend is the self-loop term x@loop_weight + h_bias (gv_gemm_f32); with addend == NULL, act = NONE,
 * keep == NULL the call returns the raw aggregate (multi-GPU partial).
 * transpose_w = 1 contracts over the SECOND block index instead (backward w.r.t. x:
 *   grad_x[s, b*si+i] = sum_{e: src_e = s} norm_e * sum_j g[dst_e, b*so+j] * W[etype_e, b, i, j]
 * with blk_in = so, blk_out = si, nbr = dst ordered by src).
 * coef_idx (optional) reads the edge coefficient as coef[coef_idx[e]] (DistMult backward reuses K1
 * with si = so = 1, W = w_relation, coef = dL/dscore per triplet: kgvae/link_predict.py:57-63).
 * partial: [n_slots, nb*blk_out] workspace (may be NULL when n_fix == 0).
 */
int gv_rgcn_bdd_aggregate(const int32_t* items, int n_items, const int32_t* fix, int n_fix,
                          const int32_t* nbr, const int32_t* etype, const float* coef, const int32_t* coef_idx,
                          const float* feat, int ld_feat, const float* weight, int num_rels, int num_bases,
                          int blk_in, int blk_out, int transpose_w, int weight_packed,
                          const float* addend, int ld_addend, int act, const uint8_t* keep, float keep_scale,
                          float* out, int ld_out, float* partial, void* stream);

/* Lane-packed relation weights for K1 (weight_packed = 1 above): the per-edge weight read becomes one
 * contiguous burst per load instruction (it dominates K1's cache traffic).  The layout depends on the launch
 * kind, so pack once per (layer, transpose_w) per step; packed has the size of weight.
 * gv_rgcn_bdd_pack_supported returns 1 when a packed kernel exists for that launch. */
int gv_rgcn_bdd_pack_supported(int num_bases, int blk_in, int blk_out, int transpose_w);
int gv_rgcn_bdd_pack_weight(const float* weight, int num_rels, int num_bases, int blk_in, int blk_out, int transpose_w,
                            float* packed, void* stream);
/* Both layouts a layer needs in one launch: packed_fwd for (blk_in, blk_out, transpose_w = 0), packed_bwd for the
 * backward-x launch (blk_out, blk_in, transpose_w = 1). */
int gv_rgcn_bdd_pack_weight_pair(const float* weight, int num_rels, int num_bases, int blk_in, int blk_out, float* packed_fwd,
                                 float* packed_bwd, void* stream);

/* K1 by RELATION PHASES (same contract as gv_rgcn_bdd_aggregate: same formula, addend / act / keep epilogue, split hub rows
 * through partial + fix): the per-edge weight read -- DGL materialises W[etype] per edge (index_select in bdd_message_func;
 * call sites kgvae/model.py:54-59) -- goes through LDS instead of the vector-memory path.  A workgroup of nw waves owns a
 * tile of nw*K work items (rows, or <= chunk-edge slices of hub rows; K = rows_per_wave); wave w keeps its K output rows in
 * registers and the workgroup walks the relation types in phases of rels_per_phase consecutive types whose lane-packed
 * block weights are staged once per tile into LDS (double buffered LDS-DMA).
 *   off        int32 [n_tiles * nw * n_phases + 1]  first edge position of every (tile, wave, phase) list; n_phases =
 *              ceil(num_rels / rels_per_phase), nw = block_threads / 64
 *   nbr        int32 [E]  gathered row of each edge, edges in (tile, wave, phase, slot) order
 *   meta       int32 [E]  ((etype - phase*rels_per_phase) << 4) | item slot k (0 <= k < K)
 *   coef       float [E]  edge coefficient in the same order (or NULL)
 *   tile_items int32 [n_tiles][nw*K][4] = {row (-1: empty slot), partial slot (-1: final row), 0, 0}
 *   weight_packed  gv_rgcn_bdd_pack_weight_phase(weight): [parts][R][NQ][L] float4, plan[5] floats
 * Every row is summed by one wave in (phase, list) order: no atomics, bitwise reproducible.
 * gv_rgcn_bdd_phase_plan returns 1 when a phase kernel exists for the block shape and writes (HOST pointer)
 *   plan[7] = {blocks per lane, column parts, K, rels_per_phase for lds_bytes of LDS, n_phases, floats of weight_packed,
 *              workgroup threads};
 * num_buffers = 2: phase p+1's weights land while phase p is computed; 1: one buffer with twice the relations per phase,
 * refilled between two barriers.  rows_per_wave = 0: the shape's default K, 4: small tiles (the 8-row shapes).
 * block_threads = 0: the shape's default workgroup (1 024 threads; 768 = 12 waves of 11 rows for the 5x10 blocks), else
 * the caller's (a multiple of 64); plan[6] is what gv_rgcn_bdd_aggregate_phases must be launched with.
 * GV_PHASE_STREAM=0 selects the round-2 kernel (one batch of rows per list) instead of the streamed one (a ring of rows in
 * flight across phase boundaries; the only one with the 11-row geometry): same lists, bit-identical results. */
int gv_rgcn_bdd_phase_plan(int num_bases, int blk_in, int blk_out, int transpose_w, int num_rels, int lds_bytes,
                           int num_buffers, int rows_per_wave, int block_threads, int32_t* plan_host);
int gv_rgcn_bdd_pack_weight_phase(const float* weight, int num_rels, int num_bases, int blk_in, int blk_out, int transpose_w,
                                  float* packed, void* stream);
int gv_rgcn_bdd_aggregate_phases(const int32_t* off, const int32_t* nbr, const int32_t* meta, const float* coef,
                                 const int32_t* tile_items, int n_tiles, const int32_t* fix, int n_fix, const float* feat,
                                 int ld_feat, const float* weight_packed, int num_rels, int num_bases, int blk_in,
                                 int blk_out, int transpose_w, int rows_per_wave, int rels_per_phase, int num_buffers,
                                 int block_threads, const float* addend, int ld_addend, int act, const uint8_t* keep,
                                 float keep_scale, float* out, int ld_out, float* partial, void* stream);

/* K1 with ALL relation weights RESIDENT IN LDS (same formula and epilogue as gv_rgcn_bdd_aggregate; call sites
 * kgvae/model.py:54-59 at BASELINE configs[2]'s shape: R = 22 directed types, num_bases = 20 -> 10x10 / 10x20 / 20x10 blocks).
 * With few relation types a column part of the whole table fits a CU's LDS (88 kB at that shape): one 1 024-thread workgroup
 * per CU copies it once; the block weights then come from LDS by conflict-free ds_read_b128 instead of 8-16 kB per edge through
 * the L1 -> VGPR path.  The unit of work is a SUPER-ITEM: a run of consecutive rows with <= plan[2] edges in all, or a slice
 * of <= plan[2] edges of a longer (hub) row -- its edge metadata is one coalesced fetch, its rows cost an epilogue each.
 *   sitems      int32 [n_sitems][4] = {first edge position, end position, partial slot (-1: whole rows), 0}; positions index
 *               nbr / etype / erow (edges ordered by row); a hub row's slices carry consecutive partial slots
 *   erow        int32 [E]  row of every edge position
 *   empty_rows  int32 [n_empty]  rows without edges (their output is the epilogue of a zero aggregate)
 *   fix         int32 [n_fix][4] = {row, first partial slot, slices, 0} for the hub rows (summed in slot order by the fix-up
 *               pass, which applies the epilogue); partial: [slots][num_bases*blk_out] workspace
 *   coef / coef_idx   edge coefficient by position (or through coef_idx), as in gv_rgcn_bdd_aggregate
 *   blk_in / blk_out  gathered / produced block width; for the backward-x launch (blk_out_fwd, blk_in_fwd) and
 *               transpose_w = 1 in the pack call (the stored block is then blk_out x blk_in, read transposed)
 *   weight_packed     gv_rgcn_bdd_pack_weight_lds(weight): [parts][R][NQ][CL] float4 (+ 64 float4 of slack)
 *   max_workgroups    0 = one per CU; the grid is (max_workgroups / parts, parts)
 * gv_rgcn_bdd_lds_plan returns 1 when an instantiation exists AND the table fits (HOST pointer, may be NULL):
 *   plan[3] = {column parts, floats of weight_packed, most edges of a super-item}.
 *   bf16_operands     BASELINE configs[2]'s precision: the table holds bf16 (round to nearest even) weights, an edge's inputs are
 *               scaled by its coefficient in fp32 and rounded to bf16, products and sums in fp32 (v_dot2c_f32_bf16); feature
 *               rows and outputs stay fp32 in memory.  Half the LDS bytes per product: twice the output columns per part.
 * Every row is summed by one wave in a fixed order (bitwise reproducible); the order differs from gv_rgcn_bdd_aggregate's. */
int gv_rgcn_bdd_lds_plan(int num_bases, int blk_in, int blk_out, int num_rels, int bf16_operands, int32_t* plan_host);
int gv_rgcn_bdd_pack_weight_lds(const float* weight, int num_rels, int num_bases, int blk_in, int blk_out, int transpose_w,
                                int bf16_operands, float* packed, void* stream);
int gv_rgcn_bdd_aggregate_lds(const int32_t* sitems, int n_sitems, const int32_t* erow, const int32_t* empty_rows, int n_empty,
                              const int32_t* fix, int n_fix, const int32_t* nbr, const int32_t* etype, const float* coef,
                              const int32_t* coef_idx, const float* feat, int ld_feat, const float* weight_packed,
                              int num_rels, int num_bases, int blk_in, int blk_out, int bf16_operands, const float* addend,
                              int ld_addend, int act, const uint8_t* keep, float keep_scale, float* out, int ld_out,
                              float* partial, int max_workgroups, void* stream);

/* The fix-up pass of gv_rgcn_bdd_aggregate on its own (a caller that passed n_fix = 0 there, e.g. to
 * time the aggregation kernel alone, finishes the split rows with this). */
int gv_rgcn_bdd_fixup(const int32_t* fix, int n_fix, const float* partial, int out_dim, const float* addend,
                      int ld_addend, int act, const uint8_t* keep, float keep_scale, float* out, int ld_out,
                      void* stream);

/* K1 backward w.r.t. the block weights (edges ordered by relation; segments = relations):
 *   grad_W[r, b, i, j] = sum_{e: etype_e = r} norm_e * x[src_e, b*si+i] * g[dst_e, b*so+j]
 * partial: [n_slots, nb*si*so] workspace. accumulate != 0 adds into grad_w instead of overwriting. */
int gv_rgcn_bdd_grad_weight(const int32_t* items, int n_items, const int32_t* fix, int n_fix,
                            const int32_t* src, const int32_t* dst, const float* coef, const int32_t* coef_idx,
                            const float* x, int ld_x, const float* g, int ld_g, int num_bases, int blk_in,
                            int blk_out, float* grad_w, float* partial, int accumulate, void* stream);

/* Epilogue alone (multi-GPU: after the all-reduce of partial aggregates) and its backward
 *   fwd: out = keep*keep_scale*act(agg + addend)        bwd: g = grad_out*keep*keep_scale*act'(out) */
int gv_rgcn_epilogue_fwd(const float* agg, const float* addend, int act, const uint8_t* keep, float keep_scale,
                         float* out, int64_t n_rows, int n_cols, void* stream);
int gv_rgcn_epilogue_bwd(const float* out, const float* grad_out, int act, const uint8_t* keep, float keep_scale,
                         float* g, int64_t n_rows, int n_cols,
                         float* colsum_part /* optional: GV_EPILOGUE_COLSUM_SLICES*n_cols floats */, void* stream);
/* colsum_part != NULL (needs n_cols % 4 == 0): the same pass also writes GV_EPILOGUE_COLSUM_SLICES row-slice partials
 * of the column sums of g (the bias gradient); gv_colsum_finish(part, n_cols, n_slices, out, accumulate) adds the
 * slices in a fixed order. */
#define GV_EPILOGUE_COLSUM_SLICES 1024
int gv_colsum_finish(const float* part, int n, int n_slices, float* out, int accumulate, void* stream);
/* ---- dense per-relation weights (the `basis` regulariser after W_r = sum_b w_comp[r, b] V_b) -----------------
 * Edges in BY-RELATION order; one f32 MFMA GEMM per relation with GATHERED A rows, 64-row tiles that never cross a relation
 * boundary (tiles: int32 [n_tiles, 4] = first position, end position, relation, 0):
 *   msg[p, :] = feat[rows[p], :] @ W_r (transpose_w = 0: feat = x, rows = sources; msg is [E, out])
 *             = feat[rows[p], :] @ W_r^T (transpose_w = 1: feat = dL/dh, rows = destinations; msg is [E, in])
 * w is [R, in, out] row-major.  The per-node sums of the messages are gv_rgcn_bdd_aggregate with 1x1 blocks over msg. */
int gv_rel_rows_gemm(const float* feat, int ld_feat, const int32_t* rows, const float* w, int num_rels, int in_feat,
                     int out_feat, int transpose_w, const int32_t* tiles, int n_tiles, float* msg, void* stream);
/* grad_w[r] = sum_{p in [relptr[r], relptr[r+1])} x[x_rows[p], :]^T (scale[p] * g[g_rows[p], :])   ([R, in, out], overwritten) */
int gv_rel_gradw_gemm(const float* x, int ld_x, const int32_t* x_rows, const float* g, int ld_g, const int32_t* g_rows,
                      const float* scale, const int32_t* relptr, int num_rels, int in_feat, int out_feat, float* grad_w,
                      void* stream);
/* gv_iaf_update_fwd / _bwd (kgvae/flow_network.py:92-97) that also produce what the bf16 products read next: fwd writes x_new as
 * fp32 AND as bf16 row-major (x_b, ld ldb) and transposed (x_t [d, >= n], ld ldt); bwd writes g_net = [g_mu | g_alpha] ONLY as bf16
 * row-major (gnet_b, ld >= 2d) and transposed (gnet_t [2d, >= n]), ADDS g_z into gz_accumulate and writes gx_old (fp32). */
int gv_iaf_update_fwd_bf16(const float* z, const float* net, int ld_net, const float* x_old, const int32_t* colcount, float* x_new,
                           uint16_t* x_b, int ldb, uint16_t* x_t, int ldt, int64_t n, int d, void* stream);
/* ... x_t in tiles of 64 rows: element (column c, row r) at x_t[(r / 64) * t_tile + c * 64 + r % 64], t_tile >= 64 d elements
 * between tiles -- the operand form of gv_gemm_bf16_gradw_tiles; rows [n, 64 ceil(n / 64)) of the last tile are written as
 * zeros (they take part in that product's reduction: the caller's buffer needs no fill); gv_iaf_update_bwd_bf16_ex likewise. */
int gv_iaf_update_fwd_bf16_tiles(const float* z, const float* net, int ld_net, const float* x_old, const int32_t* colcount,
                                 float* x_new, uint16_t* x_b, int ldb, uint16_t* x_t, int64_t t_tile, int64_t n, int d, void* stream);
int gv_iaf_update_bwd_bf16(const float* z, const float* net, int ld_net, const int32_t* colcount, const float* gx, const float* gld,
                           float* gz_accumulate, uint16_t* gnet_b, int ldb, uint16_t* gnet_t, int ldt, float* gx_old, int64_t n,
                           int d, void* stream);
/* ... from ex = expf(alpha + mu) [n][ld_ex] (what a forward chain with the fused update stores, gv_chain_layer.iaf_ex) instead
 * of [mu | alpha]; gx_old may be NULL (the backward chain adds the handed-through gradient itself: gv_chain_layer.add_src);
 * flags bit 0: g_z is WRITTEN to gz_accumulate (the first pass of a backward: no zero fill, no read); bit 1 (gld == NULL only:
 * g_alpha == g_mu then): gnet_b [n][ldb >= d] receives the g_mu half alone -- a chain reads it with x_dup_half; bit 2: gnet_t in
 * tiles of 64 rows (element (column c, row r) at [(r / 64) * ldt + c * 64 + r % 64]: ldt >= 128 d is then the distance of two tiles). */
int gv_iaf_update_bwd_bf16_ex(const float* z, const float* ex, int ld_ex, const int32_t* colcount, const float* gx, const float* gld,
                              float* gz_accumulate, uint16_t* gnet_b, int ldb, uint16_t* gnet_t, int ldt, float* gx_old,
                              int flags, int64_t n, int d, void* stream);
/* Pass 0 of a MADE backward: the update was fed ONE broadcast row net_row = [mu | alpha] (2 d floats; the first pass's input is
 * the zero matrix, kgvae/flow_network.py:85-98).  ADDS g_z into gz_accumulate [n][d] and writes the gradient w.r.t. that row,
 * g_row [2 d] = column sums of [g_mu | g_alpha], without materialising the (n, 2d) gradient; gld [n] or NULL; d % 4 == 0,
 * d <= 1024; workspace = gv_iaf_update_bwd_row0_workspace_floats(d) floats.  Fixed summation order. */
int64_t gv_iaf_update_bwd_row0_workspace_floats(int d);
int gv_iaf_update_bwd_row0(const float* z, const float* net_row, const int32_t* colcount, const float* gx, const float* gld,
                           float* gz_accumulate, float* g_row, float* workspace, int64_t n, int d, void* stream);
/* ---------------------------------------------------------------------------------------------
 * K4 in bf16 (BASELINE configs[2]): the masked-MLP products of MADE / IAF (kgvae/flow_network.py:7-98, called from
 * kgvae/model.py:116-123) with bf16 STORAGE of weights and activations, bf16 MFMA, fp32 accumulation.
 *   C = epilogue(A @ B^T):  A [M][lda] bf16 (or fp32 with a_is_f32, rounded to nearest even while staged), B [N][ldb] bf16,
 *   epilogue: + bias[N], ReLU, then zero where mask[M][ldmask] (bf16) <= 0 (the ReLU mask of the backward pass); stores any of
 *   c_f32 [M][ldc] (accumulate != 0: +=), c_bf16 [M][ldcb], c_bf16_t [N][ldct] (the transposed copy: the A / B operand of the
 *   weight-gradient product, which is this same NT form over the rows).  k % 8 == 0, lda % 8 == 0 (bf16) / % 4 (fp32),
 *   ldb % 8 == 0, ldct % 4 == 0.  split_k > 1: the reduction is cut into whole 224-deep chunks, partials (workspace of
 *   gv_gemm_bf16_nt_workspace_bytes) are summed in order into a dense c_f32 (no bias / ReLU / mask / bf16 outputs).
 * gv_cast_bf16: y = bf16(x) row-major and / or y_t = its transpose.  gv_rowsum_bf16: out[r] (+)= sum over a bf16 row (bias
 * gradients from the transposed gradient copies), fp32 sums in a fixed order. */
int64_t gv_gemm_bf16_nt_workspace_bytes(int m, int n, int split_k);
int gv_gemm_bf16_nt(const void* a, int a_is_f32, int lda, const uint16_t* b, int ldb, int m, int n, int k, const float* bias,
                    int relu, const uint16_t* mask, int ldmask, float* c_f32, int ldc, int accumulate, uint16_t* c_bf16,
                    int ldcb, uint16_t* c_bf16_t, int ldct, int split_k, void* workspace, int64_t workspace_bytes,
                    void* stream);
/* The weight-gradient product of the masked MLP, dW = g^T a over all stacked rows, with the bias gradient from the same pass:
 * c_f32 [m][n] (+)= A B^T (A [m][lda], B [n][ldb] bf16, k the long reduction), a_rowsum[i] += sum_k A[i][k] (NULL: skipped).
 * Whole-output kernel: every operand element is read once.  gv_gemm_bf16_gradw_fits: 1 when (m, n, k, split_k) suit it
 * (n <= 448, m <= 896, k >= 128 split_k) -- otherwise use gv_gemm_bf16_nt with split_k and gv_rowsum_bf16. */
int gv_gemm_bf16_gradw_fits(int m, int n, int k, int split_k);
int64_t gv_gemm_bf16_gradw_workspace_bytes(int m, int n, int split_k);
int gv_gemm_bf16_gradw(const uint16_t* a, int lda, const uint16_t* b, int ldb, int m, int n, int k, float* c_f32, int accumulate,
                       float* a_rowsum, int split_k, void* workspace, int64_t workspace_bytes, void* stream);
/* The same with both operands in 64-deep K TILES: element (row, kk) of A at a[(kk / 64) * a_tile + row * 64 + kk % 64], a_tile
 * (>= 64 m, a multiple of 8) elements between tiles, B likewise; k % 64 == 0.  A workgroup's K slice is then a few contiguous
 * blocks of memory instead of a short piece out of each of m + n rows k elements apart.  This is the form gv_made_chain
 * (t_tile), gv_iaf_update_fwd_bf16_tiles and gv_iaf_update_bwd_bf16_ex (flags bit 2) write their transposed copies in. */
int gv_gemm_bf16_gradw_tiles(const uint16_t* a, int64_t a_tile, const uint16_t* b, int64_t b_tile, int m, int n, int k, float* c_f32,
                             int accumulate, float* a_rowsum, int split_k, void* workspace, int64_t workspace_bytes, void* stream);
int gv_cast_bf16(const float* x, int ldx, int rows, int cols, uint16_t* y, int ldy, uint16_t* y_t, int ldt, void* stream);
int64_t gv_rowsum_bf16_workspace_floats(int rows, int cols);
int gv_rowsum_bf16(const uint16_t* x, int ld, int rows, int cols, float* out, int accumulate, float* workspace, void* stream);
/* The same over rows cut into `count` (<= GV_ROWSUM_SEG_MAX) consecutive segments of seg_rows[i] rows, segment i summing into
 * outs[i] (NULL: skipped): the bias gradients of every layer of a MADE from one pass over their stacked buffers.  Host tables. */
#define GV_ROWSUM_SEG_MAX 8
int gv_rowsum_bf16_segments(const uint16_t* x, int ld, int rows, int cols, int count, float* const* outs, const int32_t* seg_rows,
                            int accumulate, float* workspace, void* stream);
/* ---------------------------------------------------------------------------------------------
 * K4 fused: a CHAIN of such products in one launch -- the masked MLP of one MADE pass (kgvae/flow_network.py:85-98:
 * x -> relu(W1 x + b1) -> ... -> [mu | alpha]) or its backward-x chain (g_L -> (g_L W_L) * [a_{L-1} > 0] -> ... -> g_x).
 * A workgroup owns 64 rows for the whole chain; the activations stay in LDS between layers.  Layer i computes
 *   y_i = epilogue_i(y_{i-1} @ B_i^T),  B_i [n][k] given FRAGMENT-PACKED (gv_made_pack_weight),  y_{-1} = x [m][ldx] bf16,
 *   epilogue as gv_gemm_bf16_nt: + bias, ReLU, zero where mask <= 0; y_i is rounded to bf16 for the next layer (exactly what
 *   the next gv_gemm_bf16_nt launch would read back) and stored to any of out_bf16 [m][ldb] (not on the last layer),
 *   out_bf16_t [n][ldt] (transposed), out_f32 [m][ldc] (accumulate != 0: +=).  Results are bit-identical to the
 *   launch-per-product path.  Widths are multiples of 8, layers[i].k == layers[i-1].n, at most GV_CHAIN_MAX_LAYERS layers,
 *   (2 or, with a mask, 3) x 64 x (max width + 8..23) bf16 of LDS <= 160 KB (gv_made_chain_fits tells; GV_ERR_SHAPE otherwise).
 * gv_made_pack_weight: both packings of one fp32 weight W [n][k] (row pitch ld), rounded to bf16: packed_fwd for B = W (forward
 *   layer), packed_bwd for B = W^T (backward-x); either may be NULL.  Sizes: gv_made_pack_weight_elems(n, k) and (k, n) bf16
 *   elements.  Layout: [tile of 32 B-rows][16-deep step][lane 0..63][8 bf16] = B[32 t + (lane & 31)][16 s + 8 (lane >> 5) + e],
 *   zero outside B. */
#define GV_CHAIN_MAX_LAYERS 8
typedef struct gv_chain_layer {
    const uint16_t* w_packed; /* B of this layer, fragment-packed */
    const float* bias;        /* [n] or NULL */
    const uint16_t* mask;     /* [m][ldmask] bf16 or NULL: the result is kept where mask > 0 */
    uint16_t* out_bf16;       /* [m][ldb] or NULL */
    uint16_t* out_bf16_t;     /* [n][ldt] or NULL */
    float* out_f32;           /* [m][ldc] or NULL */
    int32_t n, k, relu, accumulate, ldmask, ldb, ldt, ldc;
    /* The IAF update (kgvae/flow_network.py:92-96) fused into the LAST layer of a forward chain: iaf_z != NULL, n = 2 d, weight
     * packed by gv_made_pack_weight_iaf (a tile = 16 mu columns + the same columns' alpha).  The layer's result [mu | alpha] is
     * consumed in registers:  x_new[r][c] = colcount[c] > 0 ? z[r][c] * expf(alpha[r][c] + mu[r][c]) : x_old[r][c].
     * out_bf16 [m][ldb] / out_bf16_t [d][ldt] then receive x_new rounded to bf16 (the next pass's operands), out_f32 [m][ldc]
     * (optional) [mu | alpha] in natural column order.  Same values as gv_iaf_update_fwd_bf16 on the stored [mu | alpha]. */
    const float* iaf_z;           /* [m][iaf_ld] or NULL: no update */
    const float* iaf_x_old;       /* [m][iaf_ld]: the pass's fp32 input (read where colcount == 0) */
    const int32_t* iaf_colcount;  /* [d] */
    float* iaf_x_new;             /* [m][iaf_ld] or NULL */
    float* iaf_ex;                /* [m][iaf_ld] or NULL: expf(alpha + mu), all the update's backward needs */
    float* iaf_alpha;             /* [m][iaf_ld] or NULL: alpha (log-det = its row sums) */
    int32_t iaf_ld, iaf_reserved;
    const int32_t* iaf_keep_colcount; /* [d] or NULL: iaf_x_new is stored only for the groups of 4 columns in which this count is 0
                                       * somewhere -- the columns the NEXT pass hands through from its x_old; NULL: all columns */
    /* backward chains */
    const uint16_t* mask_t;       /* [n][ldmask_t] bf16 or NULL: the mask given TRANSPOSED (a forward chain's out_bf16_t); instead of mask */
    const float* add_src;         /* [m][ldc] fp32 or NULL: out_f32 += add_src in the columns where add_colcount == 0 (the */
    const int32_t* add_colcount;  /* [n]                    gradient the IAF update hands through to its x_old) */
    int32_t ldmask_t, ldbits;
    /* ReLU masks as BITS (what the MADE backward needs of a hidden activation is its sign): word [r][t] holds the 32 columns of
     * column tile t of row r, bit j = (the bf16-rounded result of column 32 t + j > 0).  A forward layer writes them (out_bits),
     * a backward layer keeps its result where the bit is set (mask_bits; instead of mask / mask_t).  ldbits >= ceil(n / 32). */
    uint32_t* out_bits;           /* [m][ldbits] or NULL */
    const uint32_t* mask_bits;    /* [m][ldbits] or NULL */
    int32_t x_dup_half;           /* layer 0 only: x holds columns [0, k / 2) alone and columns [k / 2, k) repeat them (k % 16 == 0) */
    int32_t t_tile;               /* > 0: out_bf16_t in tiles of 64 ROWS -- element (column c, row r) at [(r / 64) * t_tile + c * 64 + r % 64]
                                   * (ldt == 64, t_tile >= 64 n elements between tiles): what gv_gemm_bf16_gradw_tiles reads; rows
                                   * [m, 64 ceil(m / 64)) of the last tile are written as zeros by chains without tile masks / an
                                   * accumulating output (every MADE pass), left alone by the others.  0: [n][ldt] */
} gv_chain_layer;
int64_t gv_made_pack_weight_elems(int n, int k);
int gv_made_pack_weight(const float* w, int ld, int n, int k, uint16_t* packed_fwd, uint16_t* packed_bwd, void* stream);
/* ... of up to GV_CHAIN_MAX_LAYERS weights in one launch (host tables, read during the call). */
int gv_made_pack_weight_multi(int count, const float* const* w, const int32_t* ld, const int32_t* n, const int32_t* k,
                              uint16_t* const* packed_fwd, uint16_t* const* packed_bwd, void* stream);
/* The forward packing of a [mu | alpha] layer (n = 2 d) for a chain whose last layer carries the IAF update: B-row j of tile t
 * is W row 16 t + j (j < 16) or d + 16 t + j - 16; same size as the plain packing.  _multi_iaf: as _multi, the LAST entry's
 * forward copy in this order (its transposed copy, the backward chain's first layer, stays plain). */
int gv_made_pack_weight_iaf(const float* w, int ld, int n, int k, uint16_t* packed_fwd, void* stream);
int gv_made_pack_weight_multi_iaf(int count, const float* const* w, const int32_t* ld, const int32_t* n, const int32_t* k,
                                  uint16_t* const* packed_fwd, uint16_t* const* packed_bwd, void* stream);
int gv_made_chain_fits(int n_layers, const int32_t* n_of_layer, const int32_t* k_of_layer, int any_mask);
int gv_made_chain(const uint16_t* x, int ldx, int m, int n_layers, const gv_chain_layer* layers, void* stream);
/* A backward chain whose FIRST STAGE is the IAF update's backward (gv_iaf_update_bwd_bf16_ex with flags bit 2, then gv_made_chain on
 * its result, as one launch): per 64-row workgroup, g_net = [g_mu | g_alpha] is computed from dL/dx_new (gx), ex = exp(alpha + mu),
 * z and the pass's column counts -- the arithmetic of gv_iaf_update_bwd_bf16_ex element by element -- and goes as bf16 straight
 * into layer 0's LDS tile (layers[0].k == 2 d; never to memory), its transposed copy into gnt in tiles of 64 rows (element (column c
 * of 2 d, row r) at gnt[(r / 64) * t_tile + c * 64 + r % 64]; whole tiles are written, zeros past m), g_z is added into gz (flags
 * bit 0: written).  gld (per-row dL/dlogdet) may be NULL.  All fp32 operands [m][ld], 16-B aligned rows, d % 4 == 0.  Only chains
 * without tile masks / accumulating outputs (every MADE pass of the fused path). */
typedef struct gv_chain_iafb {
    const float* z; const float* ex; const float* gx; const float* gld; float* gz; const int32_t* colcount; uint16_t* gnt;
    int32_t ld, d, t_tile, flags;
    /* n_passes > 1 (at most 6): ALL passes of a MADE's backward in the one launch -- a workgroup keeps its 64 rows through them
     * (every step of a pass is row-local).  The pointers above and in `layers` describe the pass processed first; pass q's operands
     * lie q steps further in the stacked buffers (steps may be negative): ex and every mask_bits by rows_step rows, gnt and every
     * out_bf16_t by tiles_step tiles, colcount by cc_step ints, the last layer's out_f32 by of_step floats; pass q's gx and add_src
     * are pass q - 1's out_f32; gld and flags bits 0 / 1 belong to the first pass alone.  Hidden layers: tiled out_bf16_t alone, no bias.
     * flags bit 1: gx -- and the last layer's add_src, the same gradient -- hold their columns REVERSED (the backward of a PermuteLayer
     * behind the block, folded in). */
    int32_t n_passes, rows_step, tiles_step, cc_step;
    int64_t of_step;
} gv_chain_iafb;
int gv_made_chain_iafb(const gv_chain_iafb* stage, int m, int n_layers, const gv_chain_layer* layers, void* stream);
/* ALL passes of a MADE's forward in one launch (kgvae/flow_network.py:85-98: the loop over the index sets, pass 1 on -- pass 0 runs
 * on one broadcast row): n_passes (1-6) times the chain gv_made_chain runs for one pass -- hidden layers with ReLU, the IAF update
 * in the last layer's epilogue -- with a workgroup keeping its 64 rows through the passes: pass q's x_new stays in LDS (bf16) as
 * pass q + 1's input, x is read for pass 0 alone.  `layers` gives what the passes share (packed weights, biases, n, k, relu,
 * ldbits, t_tile of the tiled copies; in the last layer iaf_z, iaf_ld, ldb, t_tile); passes[q] what differs: the IAF operands
 * (x_old / x_new / ex / alpha [m][iaf_ld] fp32; x_new is stored where keep_colcount has a 0 or keep_colcount is NULL), x_new's bf16
 * copies (out_bf16 [m][ldb] row-major, may be NULL; out_bf16_t in tiles of 64 rows, may be NULL) and per hidden layer l the tiled
 * transposed copy act_t[l] and the sign words act_bits[l] [m][ldbits].  A caller that chains passes hands pass q + 1 the x_new of
 * pass q as x_old.  Same arithmetic, element by element, as n_passes calls of gv_made_chain; rows past m of the last 64-row tile
 * leave as zeros in the tiled copies.  Hidden widths <= 512, d % 8 == 0. */
typedef struct gv_chain_fwd_pass {
    const float* x_old; float* x_new; float* ex; float* alpha; const int32_t* colcount; const int32_t* keep_colcount;
    uint16_t* out_bf16; uint16_t* out_bf16_t;
    uint16_t* act_t[GV_CHAIN_MAX_LAYERS]; int32_t* act_bits[GV_CHAIN_MAX_LAYERS];
    int32_t flags, reserved;      /* flags bit 0: x_new (fp32) is stored with its columns REVERSED -- the PermuteLayer behind an IAF block
                                   * (kgvae/model.py:60-66) folded into the block's last pass; the bf16 copies keep the natural order */
} gv_chain_fwd_pass;
int gv_made_chain_fwd(const uint16_t* x, int ldx, int m, int n_layers, const gv_chain_layer* layers, int n_passes,
                      const gv_chain_fwd_pass* passes, void* stream);
/* ... with PASS 0's update as the launch's first stage instead of an input x: the net's output of pass 0 is one row (the MADE of a zero
 * input), net_row = [mu | alpha] (2 d fp32), every column's count is positive: x = z * expf(alpha + mu) -- the arithmetic of
 * gv_iaf_update_fwd_bf16_tiles on a broadcast row -- goes as bf16 straight into layer 0's LDS tile, as fp32 to x_f32 ([m][iaf_ld]:
 * what passes[0].x_old points at) and as the tiled transposed bf16 copy to x_t (the last layer's t_tile; rows past m: zeros).
 * z, iaf_ld and t_tile are the last layer's; layers[0].k == d. */
typedef struct gv_chain_fwd_row0 { const float* net_row; float* x_f32; uint16_t* x_t; } gv_chain_fwd_row0;
int gv_made_chain_fwd_row0(const gv_chain_fwd_row0* first, int m, int n_layers, const gv_chain_layer* layers, int n_passes,
                           const gv_chain_fwd_pass* passes, void* stream);
/* probes only: a device buffer of 8 x 64 int32 that workgroup 0's waves of the following gv_made_chain launches fill with
 * s_memtime stamps (tools/probes/chain_stamps.py); NULL (the default) switches it off */
int gv_made_chain_debug_stamps(int32_t* buffer);
/* ---------------------------------------------------------------------------------------------
 * K4 fused in fp32 (the reference's own precision, kgvae/README.md:4-7): the same chain on v_mfma_f32_32x32x2_f32, fp32
 * activations in LDS, walking only the parts of the masked weights (kgvae/flow_network.py:65-83: lower-triangular blocks) that
 * hold a non-zero.  Layer i computes y_i = epilogue_i(y_{i-1} @ B_i) with B_i [k][n] FRAGMENT-PACKED (gv_made_pack_weight_f32_multi:
 * packed_fwd of a weight W [n][k] is B = W^T, a forward layer; packed_bwd is B = W, the backward-x layer, whose n is W's k);
 * epilogue: + bias, ReLU, zero where mask <= 0 (the ReLU mask of a backward layer: the forward activation, fp32 [m][ldmask]),
 * then out_f32 [m][ldc] (accumulate != 0: +=).  Every output element is the k-ordered fp32 fma chain of gv_gemm_f32 without the
 * terms the plan skips -- terms that are exactly zero -- so the results are bit-identical to the launch-per-product path on finite
 * inputs.  Widths are multiples of 8, n <= 32 GV_CHAIN32_MAX_TILES, k <= 512, layers[i].k == layers[i-1].n, two LDS tiles of
 * 64 x (widest input + 4) floats <= 160 KB (gv_made_chain_f32_fits tells).
 * gv_made_chain_f32_plan (one small launch, once per mask set): from the 0/1 masks of the layers' weights (masks[i] fp32, pitch
 * ldmask[i]; transposed[i] = 0: mask of W [n][k] for a forward layer, 1: mask of W [k-rows = the layer's k][n] for a backward-x
 * layer; masks == NULL or masks[i] == NULL: dense) it writes GV_CHAIN32_PLAN_WORDS int32: per (layer, 32-column tile) the set of
 * 8-deep reduction groups that hold a non-zero, and the layer's tiles dealt to the workgroup's four waves, longest first to the
 * least loaded.  The plan depends on the masks and widths alone, not on m.
 * rows_dev (optional device scalar): rows [*rows_dev, m) are padding -- a 64-row workgroup that holds only padding stores zeros
 * to its non-accumulating outputs, as gv_gemm_f32_live_rows. */
#define GV_CHAIN32_MAX_TILES 16
#define GV_CHAIN32_PLAN_WORDS (4 + 4 * 2 * GV_CHAIN_MAX_LAYERS * GV_CHAIN32_MAX_TILES + 2 * GV_CHAIN_MAX_LAYERS * GV_CHAIN32_MAX_TILES)
typedef struct gv_chain32_layer {
    const float* w_packed;    /* B of this layer, fragment-packed: [tile of 32 columns][group of 8 k][lane 0..63] float4 =
                               * B[8 g + 2 i + (lane >> 5)][32 t + (lane & 31)], i = 0..3; zero outside B */
    const float* bias;        /* [n] or NULL */
    const float* mask;        /* [m][ldmask] or NULL: the result is kept where mask > 0 */
    float* out_f32;           /* [m][ldc] or NULL (not on the last layer) */
    int32_t n, k, relu, accumulate, ldmask, ldc;
} gv_chain32_layer;
int64_t gv_made_pack_weight_f32_elems(int n, int k);
int gv_made_pack_weight_f32_multi(int count, const float* const* w, const int32_t* ld, const int32_t* n, const int32_t* k,
                                  float* const* packed_fwd, float* const* packed_bwd, void* stream);
int gv_made_chain_f32_fits(int n_layers, const int32_t* n_of_layer, const int32_t* k_of_layer);
int gv_made_chain_f32_plan(int n_layers, const int32_t* n_of_layer, const int32_t* k_of_layer, const float* const* masks,
                           const int32_t* ldmask, const int32_t* transposed, int32_t* plan, void* stream);
int gv_made_chain_f32(const float* x, int ldx, int m, int n_layers, const gv_chain32_layer* layers, const int32_t* plan,
                      const int32_t* rows_dev, void* stream);
/* Several PASSES of a MADE in one launch, with the IAF update (kgvae/flow_network.py:92-97) fused in: everything a pass does is
 * local to a workgroup's 64 rows -- the chain, the update x_new = count > 0 ? z * expf(alpha + mu) : x_old behind it (forward),
 * the update's backward in front of it (backward) -- so the workgroup loops over the passes itself; the buffers of consecutive
 * passes are `step` rows apart (stacked over the passes: forward step = +n, backward step = -n, n = rows of one pass).
 * Pass s of the launch uses every layer's out_f32 / mask advanced by s * step rows.
 *   mode 1 (forward): x = the first pass's input slice [m][ldx >= d]; the last layer is [mu | alpha] (n = 2 d, out_f32 required);
 *     after pass s:  x_new -> x + (s + 1) * step rows  (the next pass's input), or, for the launch's last pass with flags bit 0,
 *     -> x_out [m][d].  colcount = the counts [d] of the first pass, the next pass's d entries further.
 *   mode 2 (backward): x = the first pass's slice of the buffer that receives [g_mu | g_alpha] [m][ldx >= 2 d] (written by the
 *     update's backward, then the chain's input; the weight gradient of the last layer reads it afterwards); net / ld_net = the
 *     forward [mu | alpha] of the first pass; the last layer is the accumulating one (out_f32 = dL/dx_old of the pass, d wide):
 *     the update's backward stores the handed-through gradient there, the chain adds to it; g_in = dL/dx_new of the first pass,
 *     later passes take the previous pass's out_f32 slice; g_logdet (flags bit 0: [m], first pass only); g_z [m][d] += the
 *     passes' shares (flags bit 1: the first pass WRITES it); colcount as above, the next pass's counts d entries BACK.
 *   What sits around the passes of a MADE block rides along (flags, all optional):
 *     forward  bit 2: the first pass's input is itself an update -- of pass 0, whose [mu | alpha] is ONE row net0 [2 d] with the
 *                     counts cnt0 [d] (x_old = 0) -- computed here and stored to x before it is staged;
 *              bit 3: log_det [m] = the row sums of alpha of the launch's last pass (gv_rowsum's order);
 *              bit 4: x_out is stored with its columns REVERSED (the PermuteLayer behind the block, kgvae/flow_network.py:18-35);
 *     backward bit 2: g_in (the first pass's dL/dx_new) is read with its columns reversed (that PermuteLayer's backward).
 * Same arithmetic, element by element, as gv_iaf_update_fwd / gv_iaf_update_bwd_acc around gv_made_chain_f32 launches.  d % 4 == 0. */
typedef struct gv_chain32_iaf {
    int32_t mode, passes, d, flags;
    int64_t step;
    const float* z;
    const int32_t* colcount;
    float* x_out;
    const float* net;
    const float* g_in;
    const float* g_logdet;
    float* g_z;
    int32_t ld_net, reserved;
    const float* net0;
    const int32_t* cnt0;
    float* log_det;
} gv_chain32_iaf;
int gv_made_passes_f32(float* x, int ldx, int m, int n_layers, const gv_chain32_layer* layers, const int32_t* plan,
                       const int32_t* rows_dev, const gv_chain32_iaf* iaf, void* stream);
/* The weight gradient of one masked-MLP layer over all stacked passes, fp32 (autograd of kgvae/flow_network.py:13-14 under the
 * six-pass forward :85-98):   out[j][i] (+)= wmask[j][i] * ( sum_k g[k][j] a[k][i] + g0m[j] a0[i] ),   db[j] (+)= sum_k g[k][j] + g0m[j]
 * g [k][ldg] the gradient w.r.t. the layer's output (ReLU-masked already where the layer has a ReLU), a [k][lda] the layer's input,
 * (g0, a0) pass 0's single row (g0m[j] = g0[j] where g0_act == NULL or g0_act[j] > 0, else 0; g0 == NULL: no such term), wmask
 * [m][ldw] the layer's 0/1 mask or NULL.  One workgroup owns the whole output (blocks of 224 x 224) for its slice of k and computes
 * only the 32 x 32 tiles flagged in `plan` (gv_made_gradw_f32_plan: the tiles in which wmask holds a non-zero; all of them
 * without a mask; gv_made_gradw_f32_plan_words int32) -- the others are stored as zero / left as they are when accumulating; the
 * slices are summed in a fixed order by a second launch.  m, n, ldg, lda multiples of 4; db may be NULL; workspace of
 * gv_made_gradw_f32_workspace_floats(m, n, k) floats. */
int64_t gv_made_gradw_f32_plan_words(int m, int n);
int gv_made_gradw_f32_plan(const float* wmask, int ldw, int m, int n, int32_t* plan, void* stream);
int64_t gv_made_gradw_f32_workspace_floats(int m, int n, int64_t k);
int gv_made_gradw_f32(const float* g, int ldg, const float* a, int lda, int m, int n, int64_t k, const int32_t* plan,
                      const float* wmask, int ldw, const float* g0, const float* g0_act, const float* a0, float* out, int ldo,
                      int accumulate, float* db, int db_accumulate, float* workspace, int64_t workspace_floats, void* stream);
/* The same for up to 8 products (the layers of one MADE) in ONE launch pair: a product's fields as the arguments above; workspace of
 * gv_made_gradw_f32_multi_workspace_floats(count, items) floats. */
typedef struct gv_gradw32_item {
    const float* g;
    const float* a;
    const int32_t* plan;
    const float* wmask;
    const float* g0;
    const float* g0_act;
    const float* a0;
    float* out;
    float* db;
    int32_t ldg, lda, m, n, ldw, ldo, accumulate, db_accumulate;
    int64_t k;
} gv_gradw32_item;
int64_t gv_made_gradw_f32_multi_workspace_floats(int count, const gv_gradw32_item* items);
int gv_made_gradw_f32_multi(int count, const gv_gradw32_item* items, float* workspace, int64_t workspace_floats, void* stream);
/* ---------------------------------------------------------------------------------------------
 * K4, pass 0 of MADE (kgvae/flow_network.py:85-98): the first pass feeds the masked MLP an all-zero input, so every node sees
 * the same ROW; the whole chain of 1 x k by k x n products is one single-workgroup launch.  Operands rounded to bf16, fp32
 * products and sums (the precision of BASELINE configs[2]); with layers[0].reserved != 0 the operands stay exact fp32 (the fp32 node).
 *   gv_made_row_fwd: y_l = act(y_{l-1} W_l^T + b_l); x = y_{-1} [k_0] (NULL: zeros); layer.out [n] receives y_l (fp32).
 *   gv_made_row_bwd: g_out = dL/dy_last [n_last]; per layer gm = g * [act > 0] (act NULL: no mask), gb [n] = gm,
 *     gw [n][ldgw] = gm^T inp (inp [k] = the layer's input row, NULL: zeros; gw is WRITTEN, not accumulated),
 *     g_{l-1} = gm W_l; g_x [k_0] (may be NULL) receives the gradient of the input row.
 * Widths <= 512, k % 4 == 0 and 16-B aligned rows (w, gw), layers[i].k == layers[i-1].n, at most GV_CHAIN_MAX_LAYERS layers. */
typedef struct gv_row_layer {
    const float* w;           /* [n][ld] fp32 (the mask already folded in) */
    const float* bias;        /* [n] or NULL (forward) */
    const float* act;         /* [n] or NULL: y_l of the forward pass, the ReLU mask of the backward pass */
    const float* inp;         /* [k] or NULL: y_{l-1} (backward: the other factor of gw) */
    float* out;               /* [n] or NULL (forward) */
    float* gw;                /* [n][ldgw] or NULL (backward) */
    float* gb;                /* [n] or NULL (backward) */
    int32_t n, k, ld, relu, ldgw, reserved;
} gv_row_layer;
int gv_made_row_fwd(const float* x, int n_layers, const gv_row_layer* layers, void* stream);
int gv_made_row_bwd(const float* g_out, int n_layers, const gv_row_layer* layers, float* g_x, void* stream);

/* Evaluation scorer with a fused rank count (replaces the (h, Eb, V) outer-product tensor + sort of
 * utils.perturb_and_get_rank / sort_and_rank, kgvae/utils.py:180-221): logit = q @ e^T + *bias is formed tile by tile on
 * the f32 MFMA and never stored.  Ranking happens on the LOGIT -- monotone with the reference's sigmoid but without its
 * saturation (all candidates tie at 1.0f once the logits are large, e.g. with flow_log_prob added) -- with explicit ties:
 *   count[i] = 2 * #{ j != t_i : logit[i, j] > logit[i, t_i] } + #{ j != t_i : logit[i, j] == logit[i, t_i] }
 * = twice the 0-based mid-rank of the target (the expected position under the arbitrary tie order of the reference's
 * torch.sort); a NaN candidate, or every candidate when the target's logit is NaN, counts as better.
 * q (m, h), e (v, h) row-major fp32, target int32 [m] in [0, v), bias optional device scalar (flow_log_prob),
 * tgt: m floats of workspace, count int32 [m] (zeroed here). */
int gv_rank_scores(const float* q, int ld_q, const float* e, int ld_e, const int* target, const float* bias, float* tgt,
                   int* count, int m, int v, int h, void* stream);

/* ---------------------------------------------------------------------------------------------
 * K2/K4  dense fp32 GEMM on the f32 MFMA (v_mfma_f32_32x32x2_f32; exact fp32 fma chain):
 *   C = act(op(A) @ op(B) + bias) (+ C if accumulate)      op(X) = X or X^T; bias (length N) optional.
 * Replaces x@loop_weight (DGL RelGraphConv self loop), MaskedLinear (kgvae/flow_network.py:14-15)
 * and their autograd backward.  split_k > 1 needs workspace of gv_gemm_workspace_bytes().
 */
int64_t gv_gemm_workspace_bytes(int m, int n, int k, int split_k);
int gv_gemm_f32(int trans_a, int trans_b, int m, int n, int k, const float* a, int lda, const float* b, int ldb,
                float* c, int ldc, const float* bias, int act, int accumulate, int split_k, const float* a_relu_mask,
                void* workspace, int64_t workspace_bytes, void* stream);
/* ... for a STATIC-SHAPE batch whose node arrays are padded: only the first *rows_dev (device int32) rows of the stored A exist.
 * Row-major A (trans_a == 0): 64-row tiles of padding rows are not computed, their C rows receive ZEROS (nothing when
 * accumulating); A stored [K, M] (trans_a != 0): the reduction ends at row *rows_dev.  The padding rows of A must be finite. */
int gv_gemm_f32_live_rows(int trans_a, int trans_b, int m, int n, int k, const float* a, int lda, const float* b, int ldb, float* c,
                          int ldc, const float* bias, int act, int accumulate, int split_k, const float* a_relu_mask, void* workspace,
                          int64_t workspace_bytes, const int32_t* rows_dev, void* stream);
/* ... with operands the caller KNOWS to be zero in whole blocks -- the masked weights of a MADE (kgvae/flow_network.py:14-15: W * mask
 * with an autoregressive, block-triangular mask; ~42 % of the 64 x 16 blocks at the reference's width 500):
 *   b_k_chunks (optional): one 64-bit word per 64-column tile of C; bit c set = op(B)[16 c .. 16 c + 15][that tile's columns] holds
 *     non-zero entries.  Chunks with a clear bit are neither loaded nor multiplied (k <= 1024, split_k == 1);
 *   c_tiles (optional): bit (i * ceil(n / 64) + j) of the word array set = the 64 x 64 tile (i, j) of C is wanted; other tiles are
 *     stored as act(bias) (zeros for the weight-gradient products this is for) without reading A or B.  c_tiles_wanted = the number
 *     of set bits if the caller knows it (0: unknown): with split_k > 1 the launch then holds blocks for the wanted tiles only.
 * Equal to the dense product wherever the skipped blocks are zero and the other operand is finite there (0 * inf is never formed:
 * the same statement as for gv_made_chain_f32's mask walk). */
int gv_gemm_f32_sparse(int trans_a, int trans_b, int m, int n, int k, const float* a, int lda, const float* b, int ldb, float* c,
                       int ldc, const float* bias, int act, int accumulate, int split_k, const float* a_relu_mask, void* workspace,
                       int64_t workspace_bytes, const int32_t* rows_dev /* optional */, const uint64_t* b_k_chunks,
                       const uint64_t* c_tiles, int c_tiles_wanted, void* stream);
/* The same product with bf16 OPERANDS and fp32 accumulation (BASELINE configs[2]: "bf16"): A and B are fp32 in memory,
 * rounded to bf16 (round-to-nearest-even) as they are staged, multiplied on v_mfma_f32_32x32x16_bf16; bias, act,
 * accumulate, split-K and the result stay fp32.  Equals an fp32 GEMM of the rounded operands up to summation order. */
int gv_gemm_bf16(int trans_a, int trans_b, int m, int n, int k, const float* a, int lda, const float* b, int ldb,
                float* c, int ldc, const float* bias, int act, int accumulate, int split_k, const float* a_relu_mask,
                void* workspace, int64_t workspace_bytes, void* stream);
/* a_relu_mask (optional, same storage layout and lda as A): A is read as (mask > 0 ? A : 0), i.e. the ReLU backward
 * g * [h > 0] is folded into the operand load of the two gradient products of a MaskedLinear layer. */

/* column sums of an [m, n] matrix (bias gradients), optionally of x * [relu_mask > 0]; workspace: 64*n floats. */
int gv_colsum(const float* x, const float* relu_mask, int64_t m, int n, int ld, float* out, float* workspace,
              int accumulate, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Embedding gather / gradient scatter (kgvae/model.py:185-191, nn.Embedding dense backward). */
int gv_gather_rows(const float* table, const int64_t* ids, float* out, int64_t n, int h, void* stream);
/* The same gather, additionally advancing the device RNG's tick (rng_state[1] += 1): the start-of-forward tick rides on
 * the embedding lookup instead of a launch of its own (gv_rng_tick). */
int gv_gather_rows_rng_tick(const float* table, const int64_t* ids, float* out, int64_t n, int h, uint64_t* rng_state,
                            void* stream);
int gv_scatter_add_rows(const float* grad_out, const int64_t* ids, float* grad_table, int64_t n, int h, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Random draws of the path -- nn.Dropout's keep masks inside RelGraphConv, randn_like in
 * utils.sample_gaussian (kgvae/utils.py:342-361) -- as ONE launch per forward pass.
 * rng_state = {seed, tick} (two uint64 on the device).  Element e of job j takes output e%4 of
 * Philox4x32-10(key = seed, counter = (e/4, streams[j], tick_lo, tick_hi)):
 *   GV_RNG_KEEP_MASK: uint8 byte = (u32 >= floor(drop_p * 2^32))      -- keep with probability 1 - drop_p
 *   GV_RNG_NORMAL   : float, Box-Muller on (u1, u2) = ((x0 + 1) * 2^-32, x1 * 2^-32) -> (r cos, r sin) for outputs
 *                     (0, 1) and likewise (x2, x3) for outputs (2, 3)
 * The tick is read, never written, by gv_rng_fill: advance it between forwards with gv_rng_tick or
 * gv_gather_rows_rng_tick, so a captured hipGraph draws new numbers at every replay.  ptrs/counts/... are HOST arrays. */
#define GV_RNG_MAX_JOBS 8
#define GV_RNG_KEEP_MASK 0
#define GV_RNG_NORMAL 1
int gv_rng_fill(const uint64_t* rng_state, int n_jobs, void* const* ptrs, const int64_t* counts, const int32_t* kinds,
                const float* drop_p, const uint32_t* streams, void* stream);
int gv_rng_tick(uint64_t* rng_state, void* stream);

/* ---------------------------------------------------------------------------------------------
 * K3  Gaussian parameters + reparameterisation (kgvae/utils.py:323-361, kgvae/model.py:112-113)
 *   m = h2[:, :h]; v = softplus(h2[:, h:]) + 1e-8; z = m + eps*sqrt(v)
 * bwd: grad_h2[:, :h] = gz + gm ; grad_h2[:, h:] = (gz*eps/(2 sqrt v) + gv) * sigmoid(raw)   (gm, gv optional) */
int gv_reparam_fwd(const float* h2, const float* eps, float* z, float* v, float* m_out /* optional contiguous copy of m */,
                   int64_t n, int h, void* stream);
int gv_reparam_bwd(const float* h2, const float* eps, const float* v, const float* gz, const float* gm,
                   const float* gv, float* grad_h2, int64_t n, int h, void* stream);

/* ---------------------------------------------------------------------------------------------
 * K5  DistMult scorer + BCE-with-logits (kgvae/link_predict.py:57-63, :74-77)
 *   score_t = sum_d e[s_t,d] w[r_t,d] e[o_t,d] (+ *bias);  loss = mean_t bce(score_t, label_t)
 * triplets int32 [T,3] = (s, r, o).  workspace: 1024 floats.  bias: optional device scalar
 * (flow_log_prob).  The backward w.r.t. e and w is K1 with 1x1 blocks over the triplet incidence
 * index (see gv_rgcn_bdd_aggregate / gv_rgcn_bdd_grad_weight).
 *   gv_bce_grad: dscore_t = (*gloss) * (sigmoid(score_t) - label_t) / T ;  *dbias (+)= sum_t dscore_t */
int gv_distmult_bce_fwd(const float* embed, int ld_e, const float* w_rel, int ld_w, const int32_t* triplets,
                        const int32_t* order /* optional: triplet ids sorted by subject (L2 locality), int32[T] */,
                        const float* labels, const float* bias, float* score, float* loss, float* workspace,
                        int64_t t, int h, void* stream);
int gv_bce_grad(const float* score, const float* labels, const float* gloss, float* dscore,
                const int32_t* pos3 /* optional int32[T][3] */, float* dscore_inc /* [2T] */, float* dscore_rel /* [T] */,
                float* dbias, float* workspace, int64_t t, void* stream);
/* pos3[t] = positions of triplet t in the entity-incidence order (as subject, as object) and in the by-relation order of
 * the two K1 launches of the DistMult backward: dscore_t is then also written to dscore_inc / dscore_rel at those
 * positions, and the launches take their edge coefficient without a coef_idx indirection. */

/* ---------------------------------------------------------------------------------------------
 * K6  fused reductions
 *   gv_mean_sq : *out (+)= scale * sum(x^2)            (regularization_loss, kgvae/link_predict.py:68-69)
 *   gv_axpby   : y = alpha * (*a) * x + beta * y        (a: optional device scalar, e.g. an upstream grad)
 *   gv_kl_*    : KGVAE.get_kl (kgvae/model.py:82-87; kgvae/utils.py:364-428)
 *       kl = mean_n [ logN(z; m, v) + flp - log(1/k sum_j N(z; m_j, v_j)) ],  (m_j, v_j) from z_pre (2k, h)
 *     fwd writes resp[n, k] (mixture responsibilities) for the backward.
 *     bwd: gz, gm, gv [n,h] (overwritten), g_zpre [2k,h] (overwritten, deterministic 2-pass), scaled by *gkl.
 */
int gv_mean_sq(const float* x, int64_t n, float scale, float* out, float* workspace, int accumulate, void* stream);
/* *out = scale1*sum(x1^2) + scale2*sum(x2^2) in one pass pair (the two regulariser terms); workspace: 1024 floats */
/* rows_dev (optional, device int32) -- the STATIC-SHAPE mini-batch step (one hipGraph, kgvae/link_predict.py:200-236): node
 * arrays are padded to a bound n of which only the first *rows_dev rows exist (the sampler's node count, never read on the
 * host).  Where a routine takes rows_dev, sums skip the padding rows and every 1/n of a mean becomes 1/(*rows_dev);
 * NULL = all n rows exist.  Here: x1 is rows_host rows, its mean runs over the first *rows_dev of them. */
int gv_mean_sq2(const float* x1, int64_t n1, float scale1, const float* x2, int64_t n2, float scale2, float* out,
                float* workspace, const int32_t* rows_dev, int64_t rows_host, void* stream);
int gv_axpby(int64_t n, const float* a, float alpha, const float* x, float beta, float* y, void* stream);
int gv_mul(int64_t n, const float* a, const float* b, float* out, void* stream);
/* out[i] = a[i] * b[i] for up to GV_MUL_MULTI_MAX (a, b, out, n) quadruples in ONE launch: the mask folds of a MADE's layers
 * (mask * W forward, mask * dW backward; kgvae/flow_network.py:14-15).  The tables are host arrays, read during the call. */
#define GV_MUL_MULTI_MAX 8
int gv_mul_multi(int count, const float* const* a, const float* const* b, float* const* out, const int64_t* n, void* stream);
int64_t gv_kl_workspace_bytes(int64_t n, int h, int k);
int gv_kl_fwd(const float* z, const float* m, int ld_m, const float* v, const float* z_pre, const float* flp,
              float* resp, float* kl, float* workspace, int64_t n, int h, int k, const int32_t* rows_dev, void* stream);
int gv_kl_bwd(const float* z, const float* m, int ld_m, const float* v, const float* z_pre, const float* resp,
              const float* gkl, float gscale, float z_extra, float* gz, float* gm, float* gv, float* g_zpre,
              int accumulate_zpre, float* workspace, int mix_ready, int64_t n, int h, int k, const int32_t* rows_dev,
              void* stream);
/* K3 + K6 in one sweep over the node rows (they read the same rows; kgvae/model.py:112-113 then :81-87):
 *   gv_reparam_kl_fwd = gv_reparam_fwd (z, v, m_out written) + gv_kl_fwd with kl == NULL (resp, partial sums and the mixture table
 *                       left in `workspace`, gv_kl_workspace_bytes; finish with gv_loss_combine);
 *   gv_reparam_kl_bwd = gv_kl_bwd + gv_reparam_bwd without their intermediate gz / gm / gv: gz_up (optional) is the gradient z
 *                       receives from everything else, gh2 [n, 2h] the result; g_zpre / accumulate_zpre / gscale / z_extra as in
 *                       gv_kl_bwd; `workspace` must be the one gv_reparam_kl_fwd (or gv_kl_fwd) filled for the same z_pre; k <= 16. */
int gv_reparam_kl_fwd(const float* h2, const float* eps, const float* z_pre, float* z, float* v, float* m_out, float* resp,
                      float* workspace, int64_t n, int h, int k, void* stream);
int gv_reparam_kl_bwd(const float* z, const float* h2, const float* v, const float* eps, const float* z_pre, const float* resp,
                      const float* gkl, float gscale, float z_extra, const float* gz_up, float* gh2, float* g_zpre,
                      int accumulate_zpre, float* workspace, int64_t n, int h, int k, void* stream);
/* upstream gradient = gscale * (*gkl); gz additionally receives (*gkl) * z_extra * z (the embedding regulariser's
 * gradient, kgvae/link_predict.py:68-69, folded into the same pass); accumulate_zpre adds into g_zpre; mix_ready = 1 when `workspace` is the one
 * gv_kl_fwd filled for the same z_pre (skips recomputing the mixture table).  With rows_dev: padding rows get zero gz / gm / gv
 * (k <= 16 components), and gv_kl_fwd leaves the mean to gv_loss_combine (kl must be NULL). */

/*   gv_mmd_*  : KGVAE.get_mmd / compute_kernel (kgvae/model.py:71-80, :89-102) on x = prior samples (sx, h),
 *       y = posterior rows (sy, h):  K(a,b) = exp(-mean_d (a_d-b_d)^2 / h);  mmd = mean Kxx + mean Kyy - 2 mean Kxy.
 *       workspace: sx + sy floats (sx + sy <= 1024).  bwd overwrites gx (sx, h), gy (sy, h), scaled by *gmmd.
 *   gv_prior_sample_* : the prior draw of get_mmd (sample_gaussian(..., repeat), kgvae/utils.py:355-361):
 *       out[i] = mu[i % k] + eps[i] * sqrt(softplus(raw[i % k]) + 1e-8), z_pre = [mu; raw] (2k, h); bwd -> gz_pre (2k, h). */
/* *out = c0*(*a0) + c1*(*a1) + c2*(*a2) + c3*(*a3) on device scalars (NULL terms skipped): the loss assembly of
 * LinkPredict.get_loss (kgvae/link_predict.py:91) and its scalar gradients, without host round trips. */
/* Deferred final sums: gv_distmult_bce_fwd (loss), gv_mean_sq2 (out), gv_kl_fwd (kl) and gv_mmd_fwd (mmd) accept NULL
 * for their scalar output and then leave their per-block partial sums in `workspace`.  gv_loss_combine finishes all
 * four in ONE launch from those workspaces (same shapes as the producing calls; ws_reg / ws_kl / ws_mmd may be NULL):
 *   scal = {pred, reg, kl, mmd},  *loss = pred + reg_w*reg + kl_w*kl + mmd_w*mmd   (kgvae/link_predict.py:86-91) */
int gv_loss_combine(const float* ws_pred, int64_t t, const float* ws_reg, int64_t n_embed, int64_t n_wrel,
                    const float* ws_kl, int64_t n_nodes, int h, int k, const float* ws_mmd, int sx, int sy, float reg_w,
                    float kl_w, float mmd_w, float* scal /* 4 floats, optional */, float* loss, const int32_t* rows_dev,
                    void* stream);
int gv_lincomb4(const float* a0, float c0, const float* a1, float c1, const float* a2, float c2, const float* a3,
                float c3, float* out, void* stream);
/* y_index (optional, int64[sy]): sample j of the second set is row y_index[j] of y (KGVAE.get_mmd's posterior rows are a
 * random pick of z, kgvae/model.py:96-99: no gathered copy).  With y_index the backward ADDS each row's gradient into row
 * y_index[j] of gy (float atomics; gy is then a gradient of the whole matrix that already holds other terms); without it
 * gy[j] is overwritten. */
int gv_mmd_fwd(const float* x, const float* y, const int64_t* y_index, int sx, int sy, int h, float* mmd, float* workspace,
               void* stream);
int gv_mmd_bwd(const float* x, const float* y, const int64_t* y_index, int sx, int sy, int h, const float* gmmd, float gscale,
               float* gx, float* gy, void* stream);
int gv_prior_sample_fwd(const float* z_pre, const float* eps, float* out, int s, int k, int h, void* stream);
int gv_prior_sample_bwd(const float* z_pre, const float* eps, const float* g, float* gz_pre, int accumulate, int s, int k,
                        int h, void* stream);

/* ---------------------------------------------------------------------------------------------
 * K4  IAF / MADE update step (kgvae/flow_network.py:93-96); the masked linears are gv_gemm_f32.
 *   net = [mu | alpha] (n, 2d);  x_new[:, c] = colcount[c] > 0 ? z*exp(alpha+mu) : x_old
 *   bwd multiplies the column's gradient by colcount[c] (autograd's duplicate-index behaviour,
 *   pinned by tests/golden/made.npz).  gv_rowsum: log_det = sum_d alpha.  gv_reverse_cols: PermuteLayer. */
/* ld_net: row stride of net in floats (>= 2d), or 0 to broadcast ONE net row to all n rows -- the first pass of
 * MADE.forward evaluates the MLP on an all-zero input, i.e. n identical rows.  g_net is always (n, 2d). */
int gv_iaf_update_fwd(const float* z, const float* net, int ld_net, const float* x_old, const int32_t* colcount,
                      float* x_new, int64_t n, int d, void* stream);
int gv_iaf_update_bwd(const float* z, const float* net, int ld_net, const int32_t* colcount, const float* g_xnew,
                      const float* g_logdet /*[n] or NULL*/, float* g_z, float* g_net, float* g_xold, int64_t n, int d,
                      void* stream);
/* ... with dL/dz ADDED in place where gz_accumulate != 0 (the caller sums it over a MADE's passes: no separate add), four columns
 * per thread: d and ld_net multiples of 4, 16-B aligned operands; g_net is [n][2 d]. */
int gv_iaf_update_bwd_acc(const float* z, const float* net, int ld_net, const int32_t* colcount, const float* g_xnew,
                          const float* g_logdet, float* g_z, int gz_accumulate, float* g_net, float* g_xold, int64_t n, int d,
                          void* stream);
int gv_rowsum(const float* x, int ld, int col0, int ncols, float* out, int64_t n, void* stream);
int gv_reverse_cols(const float* x, float* out, int64_t n, int d, void* stream);
/* flow_log_prob (kgvae/model.py:116-123): *out = mean over the rows that exist of sum_i x[i][r], x = the `count` (<= 8) IAF blocks'
 * per-row log-determinants [n] (host table of device pointers, read during the call); rows_dev (optional device int32): only the
 * first *rows_dev rows exist.  64 blocks of rows + one finishing launch, fixed summation order; workspace: 64 floats.  _bwd: out [len >= n] = *g / rows on the rows that exist, 0
 * on the padding rows and on [n, len) (rows that ride along without taking part in the mean) -- the gradient of every vector. */
int gv_mean_rows_multi(int count, const float* const* x, int64_t n, const int32_t* rows_dev, float* out, float* workspace,
                       void* stream);
int gv_mean_rows_bwd(const float* g, int64_t len, int64_t n, const int32_t* rows_dev, float* out, void* stream);

/* ---------------------------------------------------------------------------------------------
 * a-11  gradient clipping + Adam (kgvae/link_predict.py:227-228) over a flat parameter arena:
 *   gv_sumsq_accum : *out += sum(g^2)   (call per tensor or once on the flat grad arena)
 *   gv_adam_step   : clip coefficient min(1, max_norm/(sqrt(*sumsq)+1e-6)) applied to g on the fly,
 *                    then torch.optim.Adam's update (bias correction from *step, eps outside the sqrt). */
int gv_adam_step(float* p, const float* g, float* exp_avg, float* exp_avg_sq, int64_t n, const float* sumsq,
                 float max_norm, float lr, float beta1, float beta2, float eps, const float* step, void* stream);
/* clip_grad_norm_ + Adam.step in two launches: (1) per-block sums of g^2 into workspace (1024 floats) and *step += 1,
 * (2) the update, each block finishing the norm from the partials; *sumsq_out (optional) receives sum(g^2);
 * zero_grad != 0 clears g as it is consumed (the next iteration's optimizer.zero_grad()). */
int gv_clip_adam_step(float* p, float* g, float* exp_avg, float* exp_avg_sq, int64_t n, float* workspace,
                      float* sumsq_out, float max_norm, float lr, float beta1, float beta2, float eps, float* step,
                      int zero_grad, void* stream);

/* ---- index construction on the device (replaces python sorted(zip(dst, src, rel)) + numpy of kgvae/utils.py:127-150 and the
 * torch sort / searchsorted / cumsum chains of the host modules; SURVEY 8(b) "gv_build_csr", 8(f-1)) ---------------------------
 * All calls are asynchronous, allocation-free and synchronisation-free.  An ORDERING of n entries by an int32 key in
 * [0, n_seg) is: perm = stable argsort(key), rowptr[s] = #{key < s} (n_seg + 1 entries), and work-item lists as produced
 * by gv_segment_items_count/_fill -- but sized by their UPPER BOUNDS (gv_index_caps) and padded with -1 entries, which the
 * K1 kernels skip, so no count travels back to the host.  workspace: gv_index_workspace_bytes(n, largest n_seg). */
void gv_index_caps(int64_t n_entries, int n_seg, int chunk, int* items_cap, int* fix_cap, int* slots_cap);
int64_t gv_index_workspace_bytes(int64_t n_entries, int n_seg_max);
/* one ordering; perm == NULL: the keys are already sorted (identity order) */
int gv_build_csr(const int32_t* keys, int64_t n, int n_seg, int chunk, int32_t* perm, int32_t* rowptr, int32_t* items,
                 int items_cap, int32_t* fix, int fix_cap, void* workspace, int64_t workspace_bytes, void* stream);
/* by-destination (CSR) and by-source (CSC) orderings of an edge list + the neighbour column of each:
 * nbr_by_dst = src[perm_d], nbr_by_src = dst[perm_s].  dst_sorted != 0: the edges already are in the reference's
 * (dst, src, rel) order (kgvae/utils.py:146-147), perm_d is not written.  Rectangular graphs: n_dst != n_src. */
int gv_graph_index_build(const int32_t* src, const int32_t* dst, int64_t n_edges, int n_dst, int n_src, int dst_sorted,
                         int chunk, int32_t* perm_d, int32_t* nbr_by_dst, int32_t* rowptr_d, int32_t* items_d,
                         int items_d_cap, int32_t* fix_d, int fix_d_cap, int32_t* perm_s, int32_t* nbr_by_src,
                         int32_t* rowptr_s, int32_t* items_s, int items_s_cap, int32_t* fix_s, int fix_s_cap,
                         void* workspace, int64_t workspace_bytes, void* stream);
/* relation types in the two orderings above + the by-relation ordering (grad-W): src/dst_by_rel = src/dst[perm_r] */
int gv_relation_index_build(const int32_t* src, const int32_t* dst, const int32_t* etype, const int32_t* perm_d /* NULL = identity */,
                            const int32_t* perm_s, int64_t n_edges, int n_rel, int chunk, int32_t* et_by_dst,
                            int32_t* et_by_src, int32_t* perm_r, int32_t* src_by_rel, int32_t* dst_by_rel, int32_t* rowptr_r,
                            int32_t* items_r, int items_cap, int32_t* fix_r, int fix_cap, void* workspace,
                            int64_t workspace_bytes, void* stream);
/* triplet batch (T, 3) int32 for the DistMult backward: the 2T entity incidences (entity, other entity, relation, triplet id)
 * ordered by entity, and the T triplets ordered by relation (rel_tid = the permutation) */
int gv_triplet_index_build(const int32_t* trip, int64_t T, int n_ent, int n_rel, int chunk, int chunk_rel, int32_t* inc_other,
                           int32_t* inc_rel, int32_t* inc_tid, int32_t* rowptr_inc, int32_t* items_inc, int items_inc_cap,
                           int32_t* fix_inc, int fix_inc_cap, int32_t* rel_s, int32_t* rel_o, int32_t* rel_tid,
                           int32_t* rowptr_rel, int32_t* items_rel, int items_rel_cap, int32_t* fix_rel, int fix_rel_cap,
                           void* workspace, int64_t workspace_bytes, void* stream);

/* SEVERAL orderings in the same launches: a sampled batch (kgvae/link_predict.py:200-236) rebuilds five of them per step (edges by
 * destination / source / relation, triplets by entity / relation), each 20-60 k entries -- every pass of every ordering is one short
 * launch, and a step is bound by the number of DEPENDENT launches.  The orderings are independent, so pass p of all of them runs as one
 * grid: six launches (two histogram + scatter passes, row pointers, work items) for up to 8 orderings, bit-identical to gv_build_csr
 * ordering by ordering.  carry_src[k] (optional, up to three arrays) come out permuted with the ordering: carry_out[k][i] =
 * carry_src[k][perm[i]] (perm NULL: copied) -- the neighbour / relation columns the index builders above gather.  Orderings outside
 * the batched kernels (n = 0, n_seg > 65 536, n > 524 288, GV_INDEX_SORT set) make the call run them one after the other. */
typedef struct gv_csr_job {
    const int32_t* keys;            /* n segment ids in [0, n_seg) */
    int64_t n;
    int32_t n_seg, chunk;
    int32_t* perm;                  /* out, n entries; NULL: keys are already sorted */
    int32_t* rowptr;                /* out, n_seg + 1 */
    int32_t* items;                 /* out, 4 * items_cap (gv_index_caps) */
    int32_t* fix;                   /* out, 4 * fix_cap */
    int32_t items_cap, fix_cap;
    const int32_t* carry_src[3];
    int32_t* carry_out[3];
} gv_csr_job;
int64_t gv_build_csr_batch_workspace_bytes(const gv_csr_job* jobs, int n_jobs);
int gv_build_csr_batch(const gv_csr_job* jobs, int n_jobs, void* workspace, int64_t workspace_bytes, void* stream);
/* the inputs of a triplet batch's two orderings in one launch: the 2T incidences (ent = subject | object, other = object | subject,
 * rel2, tid = triplet number) and the three columns of the (T, 3) list */
int gv_triplet_lists(const void* trip /* (T, 3) int32, or int64 as utils.negative_sampling returns it */, int trip_is_int64, int64_t T,
                     int32_t* ent, int32_t* other, int32_t* rel2, int32_t* tid, int32_t* col_s, int32_t* col_r, int32_t* col_o,
                     int32_t* trip32 /* optional: the (T, 3) list as int32 */, void* stream);
/* a_out[i] = a[i], b_out[i] = b[i]: the sampler's int32 node ids and row picks as the int64 tensors the reference's interfaces carry
 * (g.ndata['id'], random.sample rows), one launch for both */
int gv_widen2_i32(const int32_t* a, int64_t* a_out, int64_t na, const int32_t* b, int64_t* b_out, int64_t nb, void* stream);

/* ---- mini-batch preparation on the device (kgvae/utils.py:79-171) -------------------------------------------
 * Random draws are Philox4x32-10 outputs keyed by (seed, tick, stream_id): a batch is a pure function of those three. */
/* out[i] = i-th output of a keyed permutation of [0, n), i < k <= n: k DISTINCT indices (np.random.choice(n, k, replace=False),
 * kgvae/utils.py:79-82) without sorting n keys: 4-round unbalanced Feistel network over ceil(log2 n) bits, cycle walking */
/* tick_dev (optional, device uint64): added to tick when the kernel RUNS, n_dev (optional, device int32): replaces n then -- a
 * launch recorded in a hipGraph draws a fresh sample per replay once gv_rng_tick advances *tick_dev inside the same graph */
int gv_perm_sample(int64_t n, int64_t k, uint64_t seed, uint64_t tick, const uint64_t* tick_dev, const int32_t* n_dev,
                   uint32_t stream_id, int32_t* out, void* stream);
/* sample_edge_neighborhood (kgvae/utils.py:33-76; --edge-sampler neighbor, kgvae/link_predict.py:311): sample_size triplet ids by
 * neighbourhood expansion -- vertex ~ (remaining degree * seen), or uniform over vertices with degree left when nothing seen has
 * any; then an unpicked incident triplet uniformly (rejection).  adj_ptr (V+1) / adj_edge / adj_other (2 * num_triplets) are the
 * reference's adj_list flattened (get_adj_and_degrees, kgvae/utils.py:20-31), degrees (V) its degree vector.  Sequential by
 * nature: one workgroup, the per-draw CDF search is what runs in parallel.  Draw (i, attempt) = Philox4x32-10(seed;
 * (i, stream_id + 0x10000 * attempt, tick lo, tick hi)).x; attempt 0 picks the vertex at floor(u * W / 2^32) of the integer weight CDF, attempts >= 1
 * the incidence entry floor(u * degree / 2^32).  edges[i] = -1 once every triplet is picked. */
int64_t gv_neighborhood_sample_workspace_bytes(int num_vertices, int64_t num_triplets);
int gv_neighborhood_sample(const int32_t* adj_ptr, const int32_t* adj_edge, const int32_t* adj_other, const int32_t* degrees,
                           int num_vertices, int64_t num_triplets, int sample_size, uint64_t seed, uint64_t tick,
                           const uint64_t* tick_dev, uint32_t stream_id, int32_t* edges, void* workspace,
                           int64_t workspace_bytes, void* stream);
/* np.unique((a, b), return_inverse=True) (kgvae/utils.py:103-105) for ids in [0, num_ids): uniq = the sorted distinct ids
 * (first min(count, uniq_cap) of them), a_local / b_local = their ranks, *count = how many (device int32) */
int64_t gv_relabel_workspace_bytes(int num_ids);
int gv_relabel_pairs(const int32_t* a, const int32_t* b, int64_t k, int num_ids, int32_t* uniq, int uniq_cap, int32_t* a_local,
                     int32_t* b_local, int32_t* count, void* workspace, int64_t workspace_bytes, void* stream);
/* utils.negative_sampling (kgvae/utils.py:158-171): samples (k*(neg_rate+1), 3) int64 = the k positives (s, r, o) followed by
 * neg_rate corrupted copies in np.tile order, labels 1 / 0.  Corruption j*k + p replaces the subject (hit_subject != 0) or the
 * object of positive p by values[j*k + p]; with values == hit_subject == NULL the draws are made here: value =
 * mulhi(u32, *n_entities_dev), hit_subject = top bit of a second u32 (Philox counter = the corruption's index). */
int gv_negative_sampling(const int32_t* s, const int32_t* r, const int32_t* o, int64_t k, int neg_rate,
                         const int32_t* n_entities_dev, const int32_t* values, const uint8_t* hit_subject, uint64_t seed,
                         uint64_t tick, const uint64_t* tick_dev, uint32_t stream_id, int64_t* samples, float* labels,
                         void* stream);
/* utils.build_graph_from_triplets + comp_deg_norm (kgvae/utils.py:127-150): triplets keep[0..m) (keep == NULL: the first m) of
 * (s, r, o) plus their reverse edges (relation + num_rels), 2m edges in (dst, src, rel) order, norm = 1 / in-degree of each
 * edge's destination.  n_nodes_bound: any bound on the node ids (it only sizes the sort key). */
int64_t gv_graph_from_triplets_workspace_bytes(int64_t m, int n_nodes_bound, int num_rels);
int gv_graph_from_triplets(const int32_t* s, const int32_t* r, const int32_t* o, const int32_t* keep, int64_t m,
                           int n_nodes_bound, int num_rels, int32_t* src2, int32_t* dst2, int32_t* rel2, float* norm,
                           void* workspace, int64_t workspace_bytes, void* stream);
/* out_x[i] = x[idx[i]] for three int32 arrays in one launch (the sampled triplets' columns: kgvae/utils.py:100-101). */
int gv_gather3_i32(const int32_t* idx, int64_t n, const int32_t* a, const int32_t* b, const int32_t* c, int32_t* out_a, int32_t* out_b,
                   int32_t* out_c, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GCNVAE_H */

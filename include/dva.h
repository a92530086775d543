/*
 * dva.h — C ABI of libdva_hip.so, the MI355X (gfx950) implementation of the DeepViewAgg
 * multimodal hot path (mapping build -> multi-view gather -> view-attention pooling).
 *
 * Contract (SURVEY.md §8b "C-ABI to export"):
 *   - plain C entry points, caller-owned DEVICE buffers (HBM pointers), sizes as integers;
 *   - every function returns int: 0 = ok, <0 = DVA_ERR_*; no exceptions cross the ABI;
 *   - the HIP stream is passed explicitly (as void* == hipStream_t); nothing synchronises the
 *     device unless stated ("[syncs]"), no hidden global state, re-entrant;
 *   - no torch types. The reference is 100 % Python, so "the reference interface each entry
 *     replaces" is a Python call site, cited per function as file:line under /root/reference.
 *
 * Layouts: feature maps are channels-last x[B][H][W][C]; row matrices are row-major [rows][C];
 * CSR pointers are int64 (reference: torch.LongTensor). dtype codes below select the element type
 * of "feature" buffers (void*): fp32 or bf16; scores / mapping features / statistics are fp32.
 */
#ifndef DVA_H_
#define DVA_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DVA_OK 0
#define DVA_ERR_INVALID (-1)     /* bad argument (null pointer, negative size, bad enum) */
#define DVA_ERR_UNSUPPORTED (-2) /* valid request this build does not implement           */
#define DVA_ERR_LAUNCH (-3)      /* HIP runtime reported a launch/runtime error           */
#define DVA_ERR_OVERFLOW (-4)    /* composite key does not fit int64 (utils/multimodal.py:141) */

#define DVA_F32 0
#define DVA_BF16 1

#define DVA_SUM 0
#define DVA_MEAN 1
#define DVA_MAX 2
#define DVA_MIN 3

/* camera models of core/multimodal/visibility.py:478-538 */
#define DVA_CAM_EQUIRECT 0      /* 's3dis_equirectangular' */
#define DVA_CAM_PINHOLE_SCANNET 1
#define DVA_CAM_PINHOLE_KITTI 2 /* 'kitti360_perspective' */
#define DVA_CAM_FISHEYE_KITTI 3 /* 'kitti360_fisheye' */

/* library / device introspection ------------------------------------------------------------ */
int dva_version(void);
/* number of visible HIP devices, or <0 on runtime error. Does not create a context. */
int dva_device_count(void);

/* ------------------------------------------------------------------------------------------ *
 * CSR segment primitives.  Replace torch_scatter.segment_csr / gather_csr call sites:
 *   modules/multimodal/pooling.py:63 (BimodalCSRPool), :289,:295 (weighted sum, gating max),
 *   :628 (DeepSetFeat pool), :787,:807 (softmax), :813-841 (gather_csr);
 *   core/multimodal/image.py:1767 (per-view mean of mapping features).
 * Empty groups reduce to 0 (pooling.py:870); max/min ties -> first row of the group.
 * ------------------------------------------------------------------------------------------ */

/* out[g, c] = reduce_{r in [ptr[g], ptr[g+1])} src[r, c].  arg (nullable) receives, for MAX/MIN,
 * the winning row index (int32, -1 for empty groups); required later by the backward. */
int dva_segment_csr_fwd(const void* src, const int64_t* ptr, void* out, int32_t* arg,
                        int64_t n_groups, int32_t C, int32_t dtype, int32_t reduce, void* stream);

/* grad_src[r, c] for r in every group (rows outside [ptr[0], ptr[n_groups]) are not touched).
 * SUM: grad_out[g,c]; MEAN: grad_out[g,c]/n_g; MAX/MIN: grad_out[g,c] if r == arg[g,c] else 0. */
int dva_segment_csr_bwd(const void* grad_out, const int64_t* ptr, const int32_t* arg,
                        void* grad_src, int64_t n_groups, int32_t C, int32_t dtype,
                        int32_t reduce, void* stream);

/* out[r, c] = src[g(r), c]  (pooling.py:813-841 gather_csr). Backward = dva_segment_csr_fwd(SUM). */
int dva_gather_csr(const void* src, const int64_t* ptr, void* out, int64_t n_groups, int32_t C,
                   int32_t dtype, void* stream);

/* Per-group softmax of fp32 scores src[V, G] (pooling.py:758-810 segment_softmax_csr):
 * a = exp((s - max_g)/d) / (sum_g exp(.) + eps), d = sqrt(n_g) if scaling else 1. */
int dva_segment_softmax_csr_fwd(const float* src, const int64_t* ptr, float* out,
                                int64_t n_groups, int32_t G, int32_t scaling, float eps,
                                void* stream);
/* grad_src = a * (grad_a - sum_g(a * grad_a)) / d   (the eps term's O(1e-12) leak is dropped). */
int dva_segment_softmax_csr_bwd(const float* grad_out, const float* out, const int64_t* ptr,
                                float* grad_src, int64_t n_groups, int32_t G, int32_t scaling,
                                void* stream);

/* ------------------------------------------------------------------------------------------ *
 * Multi-view feature gather.  Replaces SameSettingImageData.get_mapped_features
 * (core/multimodal/image.py:1262-1287): nearest = x[feature_map_indexing] (:1285, indexing
 * tuple built at :1871-1885 after downscale_images :1916-1980); bilinear = sparse_interpolation
 * (:105-170).  Feature maps are channels-last [B,H,W,C] so one atom = one contiguous C-burst.
 * ------------------------------------------------------------------------------------------ */

/* Build the packed 8-byte gather index of every atom (mapped pixel):
 *   idx[p] = { int32 image, int16 x, int16 y } with x = floor(px / ratio), y = floor(py / ratio)
 * image[p] = images[view(p)] expanded through atom_ptr (image.py:1879-1880 repeat_interleave).
 * pixels: [P,2] (w,h) of pix_bytes-wide signed ints (2, 4 or 8: image.py:507-516 pixel_dtype).
 * ratio >= 1 is the mapping->feature-map downscale (image.py:1953-1954, float floor-division). */
int dva_pack_gather_index(const int64_t* images, const int64_t* atom_ptr, const void* pixels,
                          int32_t pix_bytes, double ratio, int64_t n_views, int64_t n_atoms,
                          void* packed_idx /* int64[P] */, void* stream);

/* dva_pack_gather_index followed by dva_gather_row_index (row_offset 0, no counts) in one pass: the flat row index
 * (image * H + y) * W + x of every atom straight from the mapping (image.py:1871-1885 + :1953-1954); what the lazy
 * nearest gather needs.  Same pixel rounding as dva_pack_gather_index. */
int dva_mapping_row_index(const int64_t* images, const int64_t* atom_ptr, const void* pixels, int32_t pix_bytes,
                          double ratio, int64_t n_views, int64_t n_atoms, int32_t B, int32_t H, int32_t W,
                          int32_t* row_idx, void* stream);
/* row_idx[p] = row_offset + (img*H + y)*W + x : the atom's row in the [B*H*W, C] view of the map
 * (image.py:1871-1885 flattened).  counts (nullable, caller-zeroed int32[B*H*W]) += 1 per atom. */
int dva_gather_row_index(const void* packed_idx, int64_t n_atoms, int32_t B, int32_t H, int32_t W,
                         int32_t row_offset, int32_t* row_idx, int32_t* counts, void* stream);

/* out[p, :] = x[img, y, x, :] */
int dva_gather_nearest_fwd(const void* x, const void* packed_idx, void* out, int64_t n_atoms,
                           int32_t B, int32_t H, int32_t W, int32_t C, int32_t dtype, void* stream);
/* grad_x[img, y, x, :] += grad_out[p, :].  grad_x is fp32 [B,H,W,C], caller-zeroed
 * (fp32 accumulation of bf16 gradients; atomics => run-to-run order may vary in the last ulp). */
int dva_gather_nearest_bwd(const void* grad_out, const void* packed_idx, float* grad_x,
                           int64_t n_atoms, int32_t B, int32_t H, int32_t W, int32_t C,
                           int32_t dtype, void* stream);

/* Nearest gather fused with the atomic MAX pool of a non-exact mapping (several pixels per view; reference
 * core/multimodal/image.py:1262-1287 followed by BimodalCSRPool('max'), modules/multimodal/pooling.py:14-71 through
 * modules.py:400-407): out[v][c] = max over the atoms a in [atom_ptr[v], atom_ptr[v+1]) of rows[row_idx[a]][c], 0 for a view
 * without atoms, ties -> first atom; no [P, C] tensor.  rows fp32 / bf16 [n_rows][C] (16-byte aligned, C % 4 resp. % 8 == 0),
 * arg uint16 [V][C] = offset of the winning atom inside its view (0xffff: none; views must own <= 65534 atoms).
 * _bwd: grad_rows fp32 [n_rows][C] = sum of grad_out[v][c] over the (view, channel) pairs whose winning atom lies on the row.
 * With the row plan of the atoms (perm int32 [P] = atoms sorted by map row, row_ptr int32 [n_rows + 1], as dva_row_plan gives
 * them) and view_of_atom int32 [P]: a deterministic segmented reduction, grad_rows written (C / vec a power of two <= 64);
 * with perm = row_ptr = view_of_atom = NULL: fp32 atomics into the caller-zeroed grad_rows (23 ms against 2 ms at
 * V = 8.4 M, C = 64 -- kept as the A/B). */
int dva_gather_segment_max_fwd(const void* rows, const int32_t* row_idx, const int64_t* atom_ptr, void* out, void* arg,
                               int64_t n_views, int64_t n_atoms, int64_t n_rows, int32_t C, int32_t dtype, void* stream);
int dva_gather_segment_max_bwd(const void* grad_out, const void* arg, const int32_t* row_idx, const int64_t* atom_ptr,
                               const int32_t* perm, const int32_t* row_ptr, const int32_t* view_of_atom, float* grad_rows,
                               int64_t n_views, int64_t n_atoms, int64_t n_rows, int32_t C, int32_t dtype, void* stream);
/* Bilinear gather with the reference's border-replicate semantics (image.py:105-170):
 * q = coord * (H, W) + 0.5 in the 1-padded map; 4 taps floor(q), floor(q+1); weights |prod(q - opposite)|.
 * coords fp32 [P,2] = (y, x) in [0,1]; the image of atom p comes from packed_idx (its x,y unused). */
int dva_gather_bilinear_fwd(const void* x, const void* packed_idx, const float* coords, void* out,
                            int64_t n_atoms, int32_t B, int32_t H, int32_t W, int32_t C,
                            int32_t dtype, void* stream);
int dva_gather_bilinear_bwd(const void* grad_out, const void* packed_idx, const float* coords,
                            float* grad_x, int64_t n_atoms, int32_t B, int32_t H, int32_t W,
                            int32_t C, int32_t dtype, void* stream);

/* ------------------------------------------------------------------------------------------ *
 * DeepViewAgg view attention.  Replaces the tail of GroupBimodalCSRPool.forward /
 * QKVBimodalCSRPool.forward (modules/multimodal/pooling.py:284-300, :514-530):
 *   att  = segment_softmax_csr(compat, ptr, scaling)                       [V,G]
 *   pool = segment_csr(val * expand_group_feat(att), ptr, 'sum')            [N,C]
 *   gate = tanh(relu(w * segment_csr(compat, ptr, 'max') + b))  (Gating, pooling.py:690-715)
 *   out  = pool * expand_group_feat(gate)
 * expand_group_feat (pooling.py:737-755): group g owns floor(C/G) (+1 for the first C mod G) channels.
 * gate_w / gate_b nullable together (gating=False): out = pool.
 * Saved for backward / save_last: att [V,G], gate [N,G], amax int32 [N,G] (row of the group max,
 * -1 for unseen points).  n_views = ptr[n_points] (used to size the launch geometry only).
 * algo: 0 = auto, 1 = generic kernels (any C, G), 2 = fused wavefront-team kernels (needs
 * C*s % 16 == 0 with C*s/16 a power of two <= 64, G a power of two dividing C with whole 16-byte
 * lanes per group; DVA_ERR_UNSUPPORTED otherwise).
 * ------------------------------------------------------------------------------------------ */
int dva_view_attention_fwd(const void* val, const float* compat, const int64_t* ptr,
                           const float* gate_w, const float* gate_b, void* out, float* att,
                           float* gate, int32_t* amax, int64_t n_points, int64_t n_views, int32_t C,
                           int32_t G, int32_t scaling, float eps, int32_t dtype, int32_t algo,
                           void* stream);

/* grad_val [V,C] (dtype), grad_compat [V,G] fp32, grad_gate_wb fp32[2*G] (caller-zeroed; atomically
 * accumulated: first G = d/dw, last G = d/db; nullable when gating is off). */
int dva_view_attention_bwd(const void* grad_out, const void* val, const float* compat,
                           const float* att, const float* gate, const int32_t* amax,
                           const int64_t* ptr, const float* gate_w, const float* gate_b,
                           void* grad_val, float* grad_compat, float* grad_gate_wb,
                           int64_t n_points, int64_t n_views, int32_t C, int32_t G, int32_t scaling,
                           int32_t dtype, int32_t algo, void* stream);

/* Same maths with the view gather fused in: the value of view v is rows[row_idx[v], :]
 * (rows = [R, C] value map, e.g. E_mod applied at feature-map level; DESIGN.md "E_mod hoisting").
 * Replaces image.py:1285 + pooling.py:284-300 in one pass: no [V, C] tensor is materialised.
 * Backward: grad_rows fp32 [R, C] non-NULL = scatter-add with atomics (caller-zeroed);
 * grad_rows NULL = only grad_compat / grad_gate_wb are produced and the caller obtains the rows
 * gradient from dva_view_gather_rows_grad (segmented reduction, deterministic, faster).
 * view_rec (nullable, fp32 [n_views, rec_stride], rec_stride >= G + 1, a multiple of 8 keeps one record
 * per 32-byte sector): per view, word 0 = point id (int32 bits), words 1..G = gate * attention per
 * group -- everything dva_view_gather_rows_grad needs about a view in one place. */
int dva_view_gather_attention_fwd(const void* rows, const int32_t* row_idx, const float* compat,
                                  const int64_t* ptr, const float* gate_w, const float* gate_b,
                                  void* out, float* att, float* gate, int32_t* amax,
                                  int64_t n_points, int64_t n_views, int32_t C, int32_t G,
                                  int32_t scaling, float eps, int32_t dtype, int32_t algo,
                                  void* stream);
int dva_view_gather_attention_bwd(const void* grad_out, const void* rows, const int32_t* row_idx,
                                  const float* compat, const float* att, const float* gate,
                                  const int32_t* amax, const int64_t* ptr, const float* gate_w,
                                  const float* gate_b, float* grad_rows, float* grad_compat,
                                  float* grad_gate_wb, float* view_rec, int32_t rec_stride,
                                  int64_t n_points, int64_t n_views, int32_t C, int32_t G,
                                  int32_t scaling, int32_t dtype, int32_t algo, void* stream);

/* Transposed view of a row index (views grouped by the feature-map row they read): the backward of
 * the gather in image.py:1285 (index_select -> index_add) as a CSR over rows.
 * row_idx int32 [n_views] with values in [0, n_rows).  Outputs: perm int32 [n_views] (view ids
 * ordered by row, original order inside a row: stable radix sort), row_ptr int32 [n_rows+1],
 * counts int32 [n_rows] (nullable) = views per row.  n_views, n_rows < 2^31. */
int64_t dva_row_plan_workspace_bytes(int64_t n_views, int64_t n_rows);
int dva_row_plan(const int32_t* row_idx, int64_t n_views, int64_t n_rows, int32_t* perm,
                 int32_t* row_ptr, int32_t* counts, void* workspace, int64_t workspace_bytes,
                 void* stream);

/* The row plan SPLIT in two (round 5; 512 < n_rows <= 2^18, otherwise DVA_ERR_UNSUPPORTED and the caller keeps dva_row_plan):
 * a two-pass stable MSD radix partition (digits key >> 9 | key & 511, tiles of 4096 views -- DVA_PLAN_TILE=8192: 8192) whose offset tables are built
 * from the row keys alone -- dva_plan_split_build: row_ptr int32 [n_rows + 1], counts int32 [n_rows] (nullable), identical to
 * dva_row_plan's, and `tables` (dva_plan_split_table_bytes; kept by the caller until the backward); scratch: 2 n_views bytes,
 * free afterwards -- and whose two scatter passes move the 16-BYTE VIEW RECORDS of the attention backward themselves --
 * dva_plan_split_sort_records: rec [n_views][16] in view order -> rec_sorted in plan order (record i = plan entry i, views of
 * a row in view order, word 3 of a record = its row key), through buf [n_views][16]; rec_sorted may be rec (or NULL: pass A only, see dva_plan_split_rows_grad); row_idx NULL: word 3
 * of the records already holds the row key (dva_chain_attn_bwd writes it there).  The rows gradient
 * (dva_view_gather_rows_grad_rec16* with perm = NULL) then streams its records: no permutation exists, no random 16-byte
 * fetch per view.  Replaces the index_add of core/multimodal/image.py:1262-1287's backward like dva_row_plan. */
int64_t dva_plan_split_table_bytes(int64_t n_views, int64_t n_rows);
int dva_plan_split_build(const int32_t* row_idx, int64_t n_views, int64_t n_rows, int32_t* row_ptr, int32_t* counts,
                         void* tables, int64_t tables_bytes, void* scratch, int64_t scratch_bytes, void* stream);
int dva_plan_split_sort_records(const int32_t* row_idx, const void* rec, int64_t n_views, int64_t n_rows,
                                const int32_t* row_ptr, const void* tables, int64_t tables_bytes, void* buf,
                                void* rec_sorted, void* stream);
/* The rows gradient straight from BUCKET-ordered records, without pass B: dva_plan_split_sort_records with rec_sorted =
 * NULL runs pass A only (view order -> bucket order, into buf); then ONE workgroup per bucket of 512 map rows keeps their C
 * fp32 sums in its registers, ranks and stages the bucket's tiles in LDS like pass B and consumes the staged records where
 * they lie (the 16 bytes per view pass B writes and the rows gradient reads again never exist).  grad_rows [n_rows][C]
 * bf16, written; deterministic; a row is summed by one lane team in view order (dva_view_gather_rows_grad_rec16* splits
 * it over 8 lane slots: the two agree to fp32 rounding).  bf16 / bf16, C in {32, 64}, G in {1, 2, 4}; otherwise
 * DVA_ERR_UNSUPPORTED (the caller runs pass B and dva_view_gather_rows_grad_rec16_to).
 * fp32 / fp32 (round 6; the reference's default arithmetic, models/base_model.py:244,381): bucket_rec = the 32-BYTE records
 * of dva_chain_attn_bwd_f32 in bucket order, from dva_plan_split_sort_records32 -- pass A on {point | 4 fp32 weights | 3 pad
 * words} records, row_idx required, the row key written into word 7 of every record; buf [n_views][32] -- grad_out fp32
 * [N][C], grad_rows fp32 [n_rows][C]; C in {32, 64}. */
int dva_plan_split_sort_records32(const int32_t* row_idx, const void* rec, int64_t n_views, int64_t n_rows,
                                  const void* tables, int64_t tables_bytes, void* buf, void* stream);
int dva_plan_split_rows_grad(const void* grad_out, const void* bucket_rec, int64_t n_views, int64_t n_rows,
                             const void* tables, int64_t tables_bytes, void* grad_rows, int32_t C, int32_t G, int32_t dtype,
                             int32_t out_dtype, void* stream);

/* grad_rows[r, c] = sum over the views v of row r of grad_out[p(v), c] * gate[p(v), g(c)] *
 * att[v, g(c)]  (written, not accumulated; fp32 [n_rows, C]).  view_point int32 [n_views] = point of
 * every view (dva_csr_expand); gate nullable (no gating); grad_out [n_points, C] in dtype.
 * With view_rec (from dva_view_gather_attention_bwd) att / gate / view_point are not read. */
int dva_view_gather_rows_grad(const void* grad_out, const float* att, const float* gate,
                              const int32_t* view_point, const int32_t* perm, const int32_t* row_ptr,
                              const float* view_rec, int32_t rec_stride, float* grad_rows,
                              int64_t n_rows, int64_t n_views, int32_t C, int32_t G, int32_t dtype,
                              void* stream);

/* The same reduction over packed 16-byte view records {int32 point | 4 x bf16 weight | pad} (dva_chain_attn_bwd):
 * bf16 grad_out, G in {1, 2, 4}, C / 8 a power of two <= 64, (C / G) % 8 == 0. */
/* (perm may be NULL: the records then lie in plan order -- record i belongs to plan entry i -- as
 * dva_chain_attn_bwd_planrec writes them.) */
int dva_view_gather_rows_grad_rec16(const void* grad_out, const int32_t* perm, const int32_t* row_ptr,
                                    const void* view_rec16, float* grad_rows, int64_t n_rows, int64_t n_views,
                                    int32_t C, int32_t G, int32_t dtype, void* stream);
/* The same with the output dtype chosen by the caller: out_dtype = DVA_F32 (as above) or DVA_BF16 -- the summed row is
 * rounded once where it is summed, for maps that are bf16 anyway (the gradient autograd hands on has the map's dtype:
 * core/multimodal/image.py:1262-1287 is an index_add in the map's dtype); no fp32 [n_rows, C] tensor, no conversion pass. */
int dva_view_gather_rows_grad_rec16_to(const void* grad_out, const int32_t* perm, const int32_t* row_ptr,
                                       const void* view_rec16, void* grad_rows, int32_t out_dtype, int64_t n_rows,
                                       int64_t n_views, int32_t C, int32_t G, int32_t dtype, void* stream);

/* Backward of a gather over the row plan (dva_row_plan): grad_rows[r, :] = sum over the plan entries e of row r
 * of weights[e] * grad_out[e >> atom_shift, :]  (fp32 [n_rows, C], written, not accumulated; deterministic).
 * Nearest gather: entries = atoms (weights NULL, atom_shift 0) -- dva_gather_nearest_bwd without atomics.
 * Bilinear gather: entries = the 4 corner taps of every atom from dva_gather_bilinear_taps (atom_shift 2);
 * n_atoms = number of ENTRIES. */
int dva_gather_rows_sum(const void* grad_out, const int32_t* perm, const int32_t* row_ptr,
                        const float* weights, int32_t atom_shift, float* grad_rows, int64_t n_rows,
                        int64_t n_atoms, int32_t C, int32_t dtype, void* stream);
/* The bilinear backward over the ANCHOR plan (3 x fewer sorted keys, every gradient row read once instead of four
 * times): anchors int32 [n_atoms] from dva_gather_bilinear_taps_anchor = the cell (top, left) of the view's 2 x 2 tap
 * block on the replication-padded grid, (img (H + 1) + top) (W + 1) + left; views without that structure carry the
 * dummy anchor B (H + 1) (W + 1).  Plan = dva_row_plan(anchors, B (H + 1) (W + 1) + 1).
 *   dva_anchor_rows_sum: S fp32 [n_anchors][4][C] (written) = per anchor and tap the weighted sum of grad_out rows;
 *   dva_anchor_combine:  grad_rows fp32 [B*H*W][C] (written) from S (pass the S of the first B (H+1) (W+1) anchors);
 *   dva_anchor_fixup:    += the taps of the dummy-anchor views (fp32 atomics; none on real data). */
int dva_gather_bilinear_taps_anchor(const void* packed_idx, const float* coords, int64_t n_atoms, int32_t B,
                                    int32_t H, int32_t W, int32_t* rows, float* weights, int32_t* anchors,
                                    void* stream);
/* The taps of n_settings <= 8 SETTINGS (feature maps of different sizes; the outputs of dva_gather_bilinear_taps_anchor per
 * setting) as ONE gather over the stacked map rows, in a given view order -- the reference concatenates the materialised
 * [V_s, C] tensors of the settings and indexes the result with view_cat_sorting (core/multimodal/image.py:1549-1588,
 * modules/multimodal/modules.py:514-525); here the taps are concatenated instead and no [V, C] tensor exists.
 * tap_rows / tap_weights / anchors: HOST arrays of n_settings device pointers; n_views / n_rows / n_anchors: HOST arrays
 * (views, map rows B H W, anchors B (H + 1) (W + 1) of every setting).  order int64 [n_views_total] (device, nullable =
 * identity): view i of the result = view order[i] of the concatenation.  Tap rows are offset by the rows of the settings
 * before, anchors by their anchors; a setting's dummy anchor (= its n_anchors) becomes the common one (= sum n_anchors). */
int dva_bilinear_taps_cat(int32_t n_settings, const void* const* tap_rows, const void* const* tap_weights,
                          const void* const* anchors, const int64_t* n_views, const int64_t* n_rows,
                          const int64_t* n_anchors, const int64_t* order, int64_t n_views_total, int32_t* rows_out,
                          float* weights_out, int32_t* anchors_out, void* stream);
int dva_anchor_rows_sum(const void* grad_out, const int32_t* perm, const int32_t* row_ptr, const float* weights,
                        float* S, int64_t n_anchors, int64_t n_views, int32_t C, int32_t dtype, void* stream);
int dva_anchor_combine(const float* S, float* grad_rows, int32_t B, int32_t H, int32_t W, int32_t C, void* stream);
int dva_anchor_fixup(const void* grad, const int32_t* rows, const float* weights, const int32_t* anchors,
                     float* grad_rows, int64_t n_atoms, int32_t B, int32_t H, int32_t W, int32_t C, int32_t dtype,
                     void* stream);
/* The same two steps for the fused bilinear path (csrc/chain_emod.hip): the rows are dy_a bf16 [V][C] in position order
 * and the BatchNorm_a backward dz_a = G dy_a - K1 - K2 z_a (bn_a fp32 [4][C] = mean | invstd | gamma | beta,
 * sm_a fp32 [2][C] = S1 / M | S2_hat / M of dva_bn_bwd_consts, natural channel order) is folded into the scatter,
 * replacing the in-place pass dva_emod_bwd(stage 1).  C a multiple of 32.  dva_anchor_rows_sum_bn takes z either as
 * z_a bf16 [V][C] (applied row by row; Y, tap_rows NULL) or, z_a NULL, as Y bf16 [R][C] (position order) + tap_rows
 * int32 [V][4]: all views of an anchor interpolate the same four rows of Y, so their z term is a 4 x 4 Gram matrix of
 * the tap weights (accumulated in the kernel) times those rows -- one random row per view instead of two. */
int dva_anchor_rows_sum_bn(const void* dy_a, const void* z_a, const float* bn_a, const float* sm_a, const int32_t* perm,
                           const int32_t* row_ptr, const float* weights, const int32_t* tap_rows, const void* Y,
                           float* S, int64_t n_anchors, int64_t n_views, int32_t C, void* stream);
int dva_anchor_fixup_bn(const void* dy_a, const void* z_a, const float* bn_a, const float* sm_a, const int32_t* rows,
                        const float* weights, const int32_t* anchors, float* grad_rows, int64_t n_atoms, int32_t B,
                        int32_t H, int32_t W, int32_t C, void* stream);
/* rows int32 [4 * n_atoms], weights fp32 [4 * n_atoms]: corner rows (tl, tr, bl, br) of the [B*H*W, C] map and
 * bilinear weights of every atom, exactly the taps of dva_gather_bilinear_fwd (image.py:138-165). */
int dva_gather_bilinear_taps(const void* packed_idx, const float* coords, int64_t n_atoms, int32_t B,
                             int32_t H, int32_t W, int32_t* rows, float* weights, void* stream);

/* ------------------------------------------------------------------------------------------ *
 * Fused DeepSetFeat (+ score Linear) chain over the V views, exact fp32, forward and backward.
 * Replaces, for map_encoder='DeepSetFeat' (pool='max', fusion='concatenation', nc_inner=32,
 * in_map=8), the torch composition of modules/multimodal/pooling.py:658-669 + :282 (MLP blocks =
 * Linear(no bias) -> BatchNorm1d -> LeakyReLU(0.2), core/common_modules/base_modules.py:38-48).
 * Each entry is one pass between two batch-statistics barriers; "a_k" are PRE-BatchNorm
 * activations [V,32] fp32, "bn" arrays are fp32 [4][32] = mean | invstd | gamma | beta,
 * "stats" are caller-zeroed double[64] = per-channel sum | sum of squares (forward) or
 * S1 = sum dz | S2 = sum dz*a_hat (backward), "sm" are fp32 [2][32] = S1/M | S2/M (zeros in eval).
 * algo (layer kernels): 0 = fp32 matrix cores (v_mfma_f32_32x32x2_f32, operands from registers),
 * 1 = first-generation VALU + LDS-broadcast kernels (kept for A/B checks).
 * act_dtype = STORAGE type of the [V,32] activation / gradient tensors (the `void*` arguments):
 * DVA_F32, or DVA_BF16 (what the reference keeps under torch.autocast; algo 0 only).  Arithmetic,
 * weights, statistics, per-point tensors (addend, pooled, dpooled, dt) and the scores stay fp32.
 * ------------------------------------------------------------------------------------------ */
/* group_of_row[r] = g for ptr[g] <= r < ptr[g+1] (dense expansion of CSR pointers, int32). */
int dva_csr_expand(const int64_t* ptr, int64_t n_groups, int32_t* group_of_row, void* stream);
/* a1 = x_map.Wa^T (F = 8). stats_only: statistics of a1; else a2 = leaky(BN1(a1)).Wb^T written
 * with its statistics (mlp_elt_1, pooling.py:649-650). */
int dva_deepset_fwd_first(const float* x_map, const float* Wa, const float* bn1, const float* Wb,
                          void* a2, double* stats, int64_t V, int32_t F, int32_t stats_only,
                          int32_t algo, int32_t act_dtype, void* stream);
/* pooled[p] = max_v leaky(BN(a[v])) over the point's views (first row on ties; 0 / arg -1 for
 * unseen points): segment_csr(x, csr, 'max') of pooling.py:660,:628.  n_views = ptr[N] (launch-geometry hint:
 * lanes per point follow the average segment length; <= 0: default). */
int dva_deepset_segmax(const void* a, const float* bn, const int64_t* ptr, float* pooled,
                       int32_t* arg, int64_t N, int64_t n_views, int32_t act_dtype, void* stream);
/* a_out[v] = leaky(BN_in(a_in[v])).W^T (+ addend[group_of_row[v]]) with statistics of a_out.
 * The addend carries the set half of the concatenation: cat(x, x_set).Wc^T = x.WcA^T + (x_set.WcB^T)[p]
 * (pooling.py:666-668).  bn_in == NULL: the input is used raw (set MLP on the pooled features).
 * a_out == NULL (algo 0): statistics only -- the activation is recomputed by its consumers. */
int dva_deepset_fwd_layer(const void* a_in, const float* bn_in, const float* W, const float* addend,
                          const int32_t* group_of_row, void* a_out, double* stats, int64_t V,
                          int32_t algo, int32_t act_dtype, void* stream);
/* out[v, g] = leaky(BN(a[v])).Ws[g] + bs[g], G <= 32 (E_score, pooling.py:258,:282; also Q/K).
 * (bn_pre, W_pre) non-NULL (algo 0): `a` is the INPUT of the layer before the scores and that layer's
 * output is recomputed in registers, a_mid = leaky(BN_pre(a)).W_pre^T, instead of having been stored
 * (the statistics of a_mid come from dva_deepset_fwd_layer with a_out == NULL). */
int dva_deepset_fwd_score(const void* a, const float* bn, const float* Ws, const float* bs,
                          float* compat, int64_t V, int32_t G, const float* bn_pre, const float* W_pre,
                          int32_t algo, int32_t act_dtype, void* stream);
int dva_deepset_bwd_score(const float* dcompat, const void* a, const float* bn, const float* Ws,
                          void* dz, float* dWs, float* dbs, double* st, int64_t V, int32_t G,
                          const float* bn_pre, const float* W_pre, int32_t algo, int32_t act_dtype,
                          void* stream);
/* Backward of one layer: da_L = BN-backward(dz_L), dW_L += da_L^T x_L, dx = da_L.W_L;
 * out = dx (raw_out) or dz_prev = dx*leaky'(BN_prev(a_prev)) with S1/S2 of BN_prev in st_prev;
 * dt[group_of_row[v]] += da_L[v] (nullable). prev_is_xmap: a_prev is x_map [V,8] and the previous
 * activation is recomputed as x_map.Wa^T. dW / dt are caller-zeroed fp32, atomically accumulated.
 * bn_prev == NULL (with raw_out): the layer input is a_prev itself (no BatchNorm / activation).
 * first_grad (nullable; DVA_BF16 storage, prev_is_xmap, !raw_out, dt NULL): caller-zeroed fp32[520] =
 * P[32][8] | Q[32][8] | SX[8] with P = sum_v dz_prev x^T, Q = sum_v a_prev_hat x^T, SX = sum_v x, from
 * which the first layer's weight gradient follows without another pass (BN-backward is linear in the
 * statistics): dWa[n][f] = gamma invstd (P - (S1/M)[n] SX[f] - (S2/M)[n] Q)[n][f].  `out` is then not
 * written (may be NULL) and dva_deepset_bwd_first is not needed.
 * a_L == NULL (DVA_BF16 storage, algo 0, for the plain / dt+raw_out / first_grad forms): the layer output is
 * not read but recomputed from its input, a_L = leaky(BN_prev(a_prev)).W_L^T (+ addend[group_of_row[v]],
 * the per-point half of the concatenation layer, nullable): three [V, 32] tensors cross the memory system
 * per view instead of four. */
int dva_deepset_bwd_layer(const void* dz_L, const void* a_L, const float* bn_L, const float* sm_L,
                          const float* W_L, const void* a_prev, const float* Wa, const float* bn_prev,
                          void* out, float* dW, double* st_prev, float* dt,
                          const int32_t* group_of_row, float* first_grad, const float* addend, int64_t V,
                          int32_t prev_is_xmap, int32_t raw_out, int32_t algo, int32_t act_dtype,
                          void* stream);
/* dz2 = (dcat + [arg[p]==v] dpooled[p]) * leaky'(BN2(a2)): joins the max-pool path (segment max
 * backward routes to the arg row only) with the direct path; S1/S2 of BN2 in st. */
int dva_deepset_bwd_max(const void* dcat, const void* a2, const float* bn2, const int32_t* arg,
                        const float* dpooled, const int32_t* group_of_row, void* dz2, double* st,
                        int64_t V, int32_t algo, int32_t act_dtype, void* stream);
/* dWa[n][j] += sum_v BN1-backward(dz1)[v][n] * x_map[v][j]. */
int dva_deepset_bwd_first(const void* dz1, const float* x_map, const float* Wa, const float* bn1,
                          const float* sm1, float* dWa, int64_t V, int32_t F, int32_t act_dtype,
                          void* stream);

/* ------------------------------------------------------------------------------------------ *
 * Weighted BatchNorm1d + LeakyReLU over the R rows of a feature map: one MLP block of E_mod
 * (pooling.py:245,275; core/common_modules/base_modules.py:38-48) evaluated on the map rows instead of
 * the V gathered views.  counts int32 [R] = views per row (dva_row_plan / dva_gather_row_index): the
 * batch statistics over the views are the statistics over the rows weighted by counts.  counts NULL = every
 * row counts once: plain BatchNorm1d + LeakyReLU over [R, C] (used for E_mod on materialised [V, C] views).
 * y / out / grad_* are [R, C] in dtype; sums = caller-zeroed double[2*C]; bn = fp32 [4][C] =
 * mean | invstd | gamma | beta; sm = fp32 [2][C] = S1/n | S2/n (zeros when running statistics are used).
 * ------------------------------------------------------------------------------------------ */
/* sums = sum_r counts_r y_r | sum_r counts_r y_r^2 */
int dva_rowbn_stats(const void* y, const int32_t* counts, double* sums, int64_t R, int32_t C,
                    int32_t dtype, void* stream);
/* out = leaky(gamma (y - mean) invstd + beta), negative slope `slope` */
int dva_rowbn_apply(const void* y, const float* bn, void* out, int64_t R, int32_t C, float slope,
                    int32_t dtype, void* stream);
/* sums = S1 | S2 with dz = grad_out leaky'(z): S1 = sum_r dz_r, S2 = sum_r dz_r a_r (a = normalised y);
 * also d beta = S1, d gamma = S2. */
int dva_rowbn_bwd_stats(const void* grad_out, const void* y, const float* bn, double* sums, int64_t R,
                        int32_t C, float slope, int32_t dtype, void* stream);
/* grad_y = gamma invstd (dz - counts (S1/n) - counts a (S2/n)) */
int dva_rowbn_bwd_apply(const void* grad_out, const void* y, const int32_t* counts, const float* bn,
                        const float* sm, void* grad_y, int64_t R, int32_t C, float slope, int32_t dtype,
                        void* stream);

/* BatchNorm1d bookkeeping between two passes (one launch): sums = double[2*C] (sum | sum of squares over m
 * rows) -> bn = fp32 [4][C] = mean | invstd | gamma | beta.  training: batch statistics (biased variance),
 * running_mean / running_var (nullable pair) updated with `momentum` using the unbiased variance,
 * *num_batches_tracked += 1 (nullable); else the running statistics are used.  gamma / beta nullable
 * (no affine).  nn.BatchNorm1d as wrapped by FastBatchNorm1d (base_modules.py:131-156). */
int dva_bn_finalize(const double* sums, double m, float* running_mean, float* running_var,
                    int64_t* num_batches_tracked, const float* gamma, const float* beta, float momentum,
                    float eps, int32_t training, int32_t C, float* bn, void* stream);
/* out[i] = (float)(in[i] * scale): S1/M | S2/M tables of the BatchNorm backward. */
int dva_scale_f64(const double* in, double scale, float* out, int32_t n, void* stream);

/* ------------------------------------------------------------------------------------------ *
 * Recompute chain (bf16 matrix cores): GroupBimodalCSRPool with map_encoder = DeepSetFeat(8 -> 32, max,
 * concatenation) for a lazily gathered (nearest, exact mapping) bf16 value map -- modules/multimodal/
 * pooling.py:263-315 + :658-669 -- without any [V, .] activation tensor: every pass re-evaluates the
 * per-view chain  x_map -> L1 -> L2 -> (+ set branch of the point) -> L5 -> L6 -> scores  from the 32-byte
 * mapping features inside the kernel (v_mfma_f32_32x32x16_bf16, layers chained in registers).  Train-mode
 * BatchNorm keeps its global barriers: one statistics pass per V-level BatchNorm layer (forward) and per
 * BatchNorm-backward (backward); in eval mode the forward is dva_chain_stats2 (set pooling) +
 * dva_chain_attn_fwd.  All backward statistics ("stats" of the *_bwd entries) are S1 = sum dy | sum dy z with z the RAW
 * layer output: S2 = sum dy z_hat = invstd (sum dy z - mean S1) is the caller's one-liner.  Operands are rounded to bf16 (what torch.autocast(bfloat16) feeds the reference's
 * Linear layers), accumulation / BatchNorm / softmax / statistics are fp32 / fp64.
 * Views are processed in TILES of <= 32 consecutive views made of whole points (points with more than 32
 * views: consecutive fragment tiles); "bn" arrays are fp32 [4][32] = mean | invstd | gamma | beta
 * (dva_bn_finalize), "stats" caller-zeroed double[64].  All per-point outputs are only written for points
 * that have views: the caller zero-fills them.
 * ------------------------------------------------------------------------------------------ */
/* ops: DVA_CHAIN_OPS_BYTES (32 KiB) device buffer receiving the weight operands (18 bf16 matrix-core operand
 * blocks + an fp32 copy of the forward ones, from which the kernels build BatchNorm-folded operands
 * bf16(0.6 gamma invstd W) for the layers whose raw output a pass does not need).  W1 [32][8], W2 [32][32],
 * W5 [32][ld5] (the first 32 columns: the per-view half of the concatenation layer), W6 [32][32], Ws [G][32], G <= 4 -- or
 * G = 32: the last layer is the KEY layer of QKVBimodalCSRPool (dva_chain_keys / dva_chain_keys_compat; its backward runs
 * through dva_chain_score_stats_keys / dva_chain_bwd_layer6_keys). */
#define DVA_CHAIN_OPS_BYTES (18 * 64 * 16 + 7 * 64 * 32)
int dva_chain_prep(const float* W1, const float* W2, const float* W5, int32_t ld5, const float* W6,
                   const float* Ws, int32_t G, void* ops, void* stream);
/* BatchNorm bookkeeping of one chain layer: dva_bn_finalize (C = 32) + a fifth table row, bn fp32 [5][32] =
 * mean | invstd | gamma | beta | shift, shift = the 0.6-scaled constant the layer's product starts from.  Plain layer
 * (W = NULL): 0.6 (beta - mean G).  Layer evaluated with BatchNorm folded into its operand, training: W fp32 [32][ldw]
 * (first K columns), sum_a fp64 [K] = sum over the m rows of the layer's INPUT; shift = 0.6 beta - bf16(0.6 G W) .
 * (sum_a / m): the folded product keeps the exact batch mean.  Every dva_chain_* kernel takes these 5-row tables. */
int dva_chain_bn_consts(const double* sums, double m, float* running_mean, float* running_var,
                        int64_t* num_batches_tracked, const float* gamma, const float* beta, float momentum, float eps,
                        int32_t training, const float* W, int32_t ldw, int32_t K, const double* sum_a, float* bn,
                        void* stream);
/* Tile table of ptr (int64 [n_points + 1]).  chunk_points int64 [n_chunks + 1]: ascending point indices,
 * chunk c = points [chunk_points[c], chunk_points[c + 1]) is tiled independently (first 0, last n_points).
 * count: counts int32 [n_chunks] = tiles per chunk.  build: offsets int64 [n_chunks] = exclusive prefix sum
 * of counts; tiles int32 [sum(counts)][2] = {first view, n_views | fragment << 8} (fragment 0 = whole
 * points, 1 / 2 / 3 = first / middle / last fragment of a point with more than 32 views). */
/* The two host-side steps around count / build as launches: chunk_points[c] = first point whose views start at or
 * after c * views_per_chunk (c = 1 .. n_chunks - 1; [0] = 0, [n_chunks] = n_points); offsets = exclusive prefix sum of
 * counts (n_chunks <= 2^20), n_tiles int32 [1] = the total. */
int dva_chain_tile_chunks(const int64_t* ptr, int64_t n_points, int64_t views_per_chunk, int32_t n_chunks,
                          int64_t* chunk_points, void* stream);
int dva_chain_tile_offsets(const int32_t* counts, int32_t n_chunks, int64_t* offsets, int32_t* n_tiles,
                           void* stream);
int dva_chain_tile_count(const int64_t* ptr, const int64_t* chunk_points, int32_t n_chunks, int32_t* counts,
                         void* stream);
int dva_chain_tile_build(const int64_t* ptr, const int64_t* chunk_points, int32_t n_chunks,
                         const int64_t* offsets, void* tiles, void* stream);
/* moments double[44] (caller-zeroed) += sum_v x | sum_v x_i x_j (i <= j, row-major upper triangle);
 * stats1 double[64] = sum z1 | sum z1^2 of z1 = bf16(W1) x, derived from the moments (exact_w1 != 0: of z1 = W1 x,
 * the first layer of the fp32-class chain dva_chain3_*; the same flag on dva_chain_stats1 / dva_chain_dw1). */
int dva_chain_moments(const float* x_map, int64_t n_views, const float* W1, int32_t exact_w1, double* moments,
                      double* stats1, void* stream);
/* stats double[96] (caller-zeroed) += sum z2 | sum z2^2 | sum a1 (the layer's input: what dva_chain_bn_consts needs
 * for the folded shift); zstar fp32 [N][32] / arg int32 [N][32] = value / first view of the per-point extremum of
 * sign(gamma2) z2 (the view that max-pools a2 = leaky(BN2(z2))). */
int dva_chain_stats2(const float* x_map, const int32_t* view_point, const void* tiles, const int32_t* n_tiles,
                     const void* ops, const float* bn1, const float* gamma2, double* stats, float* zstar,
                     int32_t* arg, int64_t n_views, void* stream);
/* pooled fp32 [N][32] = leaky(BN2(zstar)) for seen points, 0 for unseen ones. */
int dva_chain_pooled(const float* zstar, const float* bn2, const int64_t* ptr, float* pooled, int64_t n_points,
                     void* stream);
/* layer = 5: stats double[96] += sum | sum of squares of z5 = W5a a2 + u[point] (third row untouched); layer = 6: of z6,
 * third row += sum a5 (the layer's input).  u fp32 [N][32] = the per-point half of the concatenation layer
 * (W5b . set features). */
int dva_chain_stats(int32_t layer, const float* x_map, const int32_t* view_point, const float* u,
                    const void* tiles, const int32_t* n_tiles, const void* ops, const float* bn1,
                    const float* bn2, const float* bn5, double* stats, int64_t n_views, int64_t n_points,
                    void* stream);
/* The stored-a2 hybrid (round 6; reference maths unchanged: modules/multimodal/pooling.py:658-669, DeepSetFeat.forward):
 * layer = 5: dva_chain_stats(5) that also WRITES a2 bf16 [V][32] = the layer-2 activation leaky(BN2(W2 a1)) of every view in
 * ACCUMULATOR order (position 16 h + r = channel (r & 3) + 8 (r >> 2) + 4 h), i.e. the packed operand layer 5 consumes;
 * layer = 6: dva_chain_stats(6) starting from that row instead of x_map (x_map may be NULL).  a2 16-byte aligned. */
int dva_chain_stats_a2(int32_t layer, const float* x_map, const int32_t* view_point, const float* u,
                       const void* tiles, const int32_t* n_tiles, const void* ops, const float* bn1,
                       const float* bn2, const float* bn5, double* stats, int64_t n_views, int64_t n_points,
                       void* a2, void* stream);
/* Key layer of QKVBimodalCSRPool on the recompute chain (round 4; reference modules/multimodal/pooling.py:454-547:
 * keys = K(E_map(x_map))): ops prepared by dva_chain_prep with Ws = K.weight [32][32], G = 32.  keys bf16 [V][32] in
 * ACCUMULATOR order: position 16 h + r holds key channel (r & 3) + 8 (r >> 2) + 4 h (the layout of the rows the chain's
 * backward passes hand to each other).
 * dva_chain_keys_compat: the same pass also writes the compatibilities (pooling.py:520-531) compat fp32 [V][4] =
 * scale * sum over the 32 / G key channels of group g of the (bf16-rounded) key * queries[point(v)][.] -- queries fp32 [N][32]
 * in the keys' position order, G in {1, 2, 4} query-key groups (columns >= G are written as 0), 16-byte aligned. */
int dva_chain_keys(const float* x_map, const int32_t* view_point, const float* u, const void* tiles,
                   const int32_t* n_tiles, const void* ops, const float* bn1, const float* bn2, const float* bn5,
                   const float* bn6, const float* key_bias, void* keys, int64_t n_views, int64_t n_points, void* stream);
int dva_chain_keys_compat(const float* x_map, const int32_t* view_point, const float* u, const void* tiles,
                          const int32_t* n_tiles, const void* ops, const float* bn1, const float* bn2, const float* bn5,
                          const float* bn6, const float* key_bias, void* keys, const float* queries, float* compat,
                          int32_t G, float scale, int64_t n_views, int64_t n_points, void* stream);
/* compat fp32 [V][G] = scale * sum over the nc_qk = 32 / G key channels of group g of keys[v] * queries[point(v)]
 * (pooling.py:520-531), keys as dva_chain_keys writes them, queries fp32 [N][32] in the same position order; G in {1, 2, 4}.
 * _bwd: grad_keys bf16 [V][32] (position order; NULL: skipped -- the chain backward builds it in registers),
 * grad_queries fp32 [N][32] (written).  dva_qkv_dquery: grad_queries alone, grad_compat with leading dimension ld >= G
 * (the [V][4] layout of dva_chain_keys_compat / dva_chain_attn_bwd: ld = 4). */
int dva_qkv_compat(const void* keys, const float* queries, const int32_t* view_point, float* compat, int64_t n_views,
                   int32_t G, float scale, void* stream);
int dva_qkv_compat_bwd(const float* grad_compat, const void* keys, const float* queries, const int32_t* view_point,
                       const int64_t* ptr, void* grad_keys, float* grad_queries, int64_t n_points, int64_t n_views,
                       int32_t G, float scale, void* stream);
int dva_qkv_dquery(const float* grad_compat, int32_t ld, const void* keys, const int64_t* ptr, float* grad_queries,
                   int64_t n_points, int64_t n_views, int32_t G, float scale, void* stream);
/* out bf16 [N][C] (caller-zeroed) = gate * sum_v softmax_v(scores) * rows[row_idx[v]]:
 * x_map + rows in -> pooled features out.  rows bf16 [n_rows][C], C in {32, 64, 128, 256, 512},
 * G in {1, 2, 4} with (C / G) % 8 == 0; gate_w / gate_b fp32 [G] nullable together.
 * scores_out (nullable; training): fp32 [V][4] receives the scores E_score(E_map(x_map)) of every view (columns < G),
 * what dva_chain_attn_bwd starts from (16 bytes per view instead of one more chain evaluation). */
int dva_chain_attn_fwd(const float* x_map, const int32_t* view_point, const float* u, const void* tiles,
                       const int32_t* n_tiles, const void* ops, const float* bn1, const float* bn2,
                       const float* bn5, const float* bn6, const float* score_bias, const void* rows,
                       const int32_t* row_idx, const int64_t* ptr, const float* gate_w, const float* gate_b,
                       void* out, float* scores_out, int64_t n_points, int64_t n_views, int64_t n_rows, int32_t C,
                       int32_t G, int32_t scaling, float eps, void* stream);

/* ------------------------------------------------------------------------------------------ *
 * Recompute chain with a per-view E_mod: the fused path of the BILINEAR gather (interpolate=True;
 * reference core/multimodal/image.py:105-170 sparse_interpolation + modules/multimodal/pooling.py:245,275 E_mod per
 * view; csrc/chain_emod.hip).  The first Linear of E_mod commutes with the interpolation and runs on the map rows
 * (caller): Y bf16 [n_rows][C_out] = rows W_a^T in POSITION order (position 32 b + 16 h + r = channel
 * 32 b + (r & 3) + 8 (r >> 2) + 4 h: the 16 channels a lane owns are contiguous).  Per view: the 4 taps
 * tap_rows int32 [V][4] / tap_weights fp32 [V][4] (dva_gather_bilinear_taps) of Y -> z_a -> BatchNorm_a ->
 * LeakyReLU(0.2) -> Linear_b on the matrix cores -> BatchNorm_b -> LeakyReLU = the value of the view.
 * C_out in {32, 64} (dva_emod_prep and the eval form of dva_emod_attn_fwd, z_a = NULL, also 128 and, G = 4, 256), G in {1, 2, 4}.  bn_a / bn_b fp32 [4][C_out] = mean | invstd | gamma | beta (dva_bn_finalize),
 * natural channel order; statistics fp64 [2][C_out] caller-zeroed; tiles / view_point / scores / chain arguments as
 * for the dva_chain_* entries.
 * z_a bf16 [V][C_out] (position order): the interpolated Linear_a output of every view, rounded to bf16 (what the
 * layer's output is under autocast in the reference).  Train mode: written by dva_emod_stats(layer 1), read by every
 * later pass of the step instead of the four taps of Y (those passes ignore Y / tap_rows / tap_weights, which may be
 * NULL).  Eval mode: dva_emod_attn_fwd with z_a = NULL evaluates the taps itself.
 * ------------------------------------------------------------------------------------------ */
/* eops: 2 * (C_out / 32)^2 * 2 KiB: Linear_b's weight W_b fp32 [C_out][C_out] as bf16 matrix-core operands
 * (forward blocks, then the transposed blocks of the input gradient). */
int dva_emod_prep(const float* Wb, int32_t C_out, void* eops, void* stream);
/* layer 1: z_a (out, nullable) <- bf16(taps of Y), stats += sum z_a | sum z_a^2 of the stored values (eops, bn_a
 * unused);  layer 2: stats += sum z_b | sum z_b^2 from z_a (in). */
int dva_emod_stats(int32_t layer, const void* Y, const int32_t* tap_rows, const float* tap_weights, const void* tiles,
                   const int32_t* n_tiles, const void* eops, const float* bn_a, double* stats, void* z_a,
                   int64_t n_views, int64_t n_rows, int32_t C_out, void* stream);
/* Layer-1 statistics pass of dva_emod_stats in ANCHOR order (round 4): perm = the permutation of the anchor plan
 * (dva_row_plan of the views' anchors, the plan the backward scatters through).  A tile is 32 consecutive plan entries, so
 * neighbouring lanes read the same four rows of Y (cache hits instead of 8 C_out gathered bytes per view); z_a rows are
 * written to their view's position, stats as dva_emod_stats(layer 1).  C_out in {32, 64, 128, 256}; z_a may be NULL. */
int dva_emod_stats1_plan(const void* Y, const int32_t* tap_rows, const float* tap_weights, const int32_t* perm,
                         double* stats, void* z_a, int64_t n_views, int64_t n_rows, int32_t C_out, void* stream);
/* The fused view kernel: x_map + z_a (or, z_a = NULL, the taps of Y) -> out bf16 [N][C_out] (caller-zeroed) =
 * gate * sum_v softmax_v(scores) E_mod(view v); scores_out as dva_chain_attn_fwd. */
int dva_emod_attn_fwd(const float* x_map, const int32_t* view_point, const float* u, const void* tiles,
                      const int32_t* n_tiles, const void* ops, const float* bn1, const float* bn2, const float* bn5,
                      const float* bn6, const float* score_bias, const void* Y, const int32_t* tap_rows,
                      const float* tap_weights, const void* eops, const float* bn_a, const float* bn_b,
                      const int64_t* ptr, const float* gate_w, const float* gate_b, void* out, float* scores_out,
                      const void* z_a, int64_t n_points, int64_t n_views, int64_t n_rows, int32_t C_out, int32_t G,
                      int32_t scaling, float eps, void* stream);
/* Attention + gate backward from the stored scores with E_mod re-evaluated: grad_scores, view_rec, grad_gate_wb as
 * dva_chain_attn_bwd; stats_b += S1 | sum dy_b z_b of the BatchNorm_b backward. */
int dva_emod_attn_bwd(const float* scores, const int32_t* view_point, const void* tiles, const int32_t* n_tiles,
                      const void* Y, const int32_t* tap_rows, const float* tap_weights, const void* eops,
                      const float* bn_a, const float* bn_b, const int64_t* ptr, const float* gate_w,
                      const float* gate_b, const void* grad_out, const void* out, float* grad_scores, void* view_rec,
                      float* grad_gate_wb, double* stats_b, const void* z_a, int64_t n_points, int64_t n_views,
                      int64_t n_rows, int32_t C_out, int32_t G, int32_t scaling, float eps, void* stream);
/* E_mod backward.  stage 2: view_rec + grad_out -> dWb fp32 [C_out][C_out] (caller-zeroed) += the gradient of W_b,
 * da bf16 [V][C_out] (position order) = leaky'(y_a) W_b^T dz_b, stats_a += S1 | sum dy_a z_a  (sm_b = S / M of
 * BatchNorm_b).  stage 1: da <- dz_a = BatchNorm_a backward of da in place (sm_a; C_out <= 64); the gradient of Y follows as
 * dva_gather_rows_sum(da, row plan of tap_rows, tap_weights, atom_shift 2).
 * C_out in {32, 64, 128}.  C_out = 128 ("wide rows", round 4) evaluates stage 2 as two kernels that can also be called
 * on their own: stage 3 = da + stats_a only (dWb may be NULL), stage 4 = dWb only (stats_a may be NULL). */
int dva_emod_bwd(int32_t stage, const void* Y, const int32_t* tap_rows, const float* tap_weights, const void* tiles,
                 const int32_t* n_tiles, const void* eops, const float* bn_a, const float* bn_b, const float* sm_a,
                 const float* sm_b, const void* view_rec, const void* grad_out, void* da, float* dWb, double* stats_a,
                 const void* z_a, int64_t n_points, int64_t n_views, int64_t n_rows, int32_t C_out, int32_t G,
                 void* stream);

/* The arithmetic between two BatchNorm-backward passes as one launch.  stats fp64 [2C] = S1 | S2, bn fp32 [4][C] =
 * mean | invstd | gamma | beta.  do_hat: S2 arrives as sum dy z (raw layer output) and becomes
 * invstd (S2 - mean S1) = sum dy z_hat, in place.  Then sm fp32 [2C] = stats * inv_m, dgamma = S2, dbeta = S1
 * (each nullable). */
int dva_bn_bwd_consts(double* stats, const float* bn, double inv_m, int32_t do_hat, float* sm, float* dgamma,
                      float* dbeta, int32_t C, void* stream);
/* Statistics of the BatchNorm-1 backward from P (stage 2 of dva_chain_bwd_layer): P fp32 [32][20] = sum_v dy1
 * [x_hi (8) | x_lo (8) | 1 | unused (3)]^T; stats fp64 [64] = sum dy1 (= P[:, 16]) | sum dy1 z1 with z1 = bf16(W1) x
 * (= sum_f bf16(W1)[:, f] (P[:, f] + P[:, 8 + f]): z1 is linear in x).  Raw-z form like every other "stats". */
int dva_chain_stats1(const float* P, const float* W1, int32_t exact_w1, double* stats, void* stream);
/* First-layer weight gradient without another view pass: BatchNorm-1 backward is linear in its statistics and
 * z1 = W1 x is linear in x, so dW1 = G1 (P - (S1/M) SX^T - (S2/M) . Q), Q = sum_v z1_hat x^T from the moments.
 * P fp32 [32][20] (dva_chain_bwd_layer stage 2: x_map as hi | lo columns), mom fp64 [44] (dva_chain_moments), sm1 = S/M of layer 1. */
int dva_chain_dw1(const float* P, const double* mom, const float* W1, int32_t exact_w1, const float* bn1,
                  const float* sm1, float* dW1, void* stream);
/* Per-point set branch of DeepSetFeat on the chain (pooling.py:660-664): pooled fp32 [N][32] (+ the set-size
 * feature sqrt(1 / (n + 1e-3)) when w33 = Wsa[:, 32] is given) -> mlp_set = MLP[32(+1), 32, 32] -> u = Wc[:, 32:] . s,
 * the per-point half of the concatenation layer.  Every pass re-evaluates the branch from pooled (three-term bf16
 * split: fp32-class accuracy); one pass per BatchNorm barrier.  ops: 24 KiB device buffer (dva_chain_set_prep:
 * Wsa [32][ld_sa], Wsb [32][32], Wc [32][ld_c] of which the columns 32..63 are used).
 *   fwd stage 1: stats += statistics of s1 | stage 2: of s2 (needs bn_s1) | stage 3: u fp32 [N][32] (bn_s1, bn_s2)
 *   bwd stage 1: dW [32][ld_dw] += du^T a2 (= d Wc[:, 32:]), stats += S of layer s2
 *       stage 2: dW += d Wsb, stats += S of layer s1                                   (sm_s2)
 *       stage 3: dW += d Wsa[:, :32], dw33[i * ld_dw] += d Wsa[i, 32] (nullable), dpooled fp32 [N][32]  (sm_s1, sm_s2)
 * Backward statistics are S1 = sum dy | sum dy z (raw layer output), like dva_chain_bwd_layer. */
int dva_chain_set_prep(const float* Wsa, int32_t ld_sa, const float* Wsb, const float* Wc, int32_t ld_c, void* ops,
                       void* stream);
int dva_chain_set_fwd(int32_t stage, const float* pooled, const int64_t* ptr, const float* w33, const void* ops,
                      const float* bn_s1, const float* bn_s2, float* u, double* stats, int64_t n_points,
                      void* stream);
int dva_chain_set_bwd(int32_t stage, const float* pooled, const int64_t* ptr, const float* w33, const void* ops,
                      const float* bn_s1, const float* bn_s2, const float* sm_s1, const float* sm_s2,
                      const float* du, float* dpooled, float* dW, int32_t ld_dw, float* dw33, double* stats,
                      int64_t n_points, void* stream);
/* Merged chain backward (round 5; autograd of modules/multimodal/pooling.py:263-315, :658-669 as the entries around it): the
 * score pass also accumulates the pieces the statistics of the BatchNorm-5 backward are linear in (dz6 = G6 dy6 - K1 - K2 z6
 * is linear in the constants K1, K2 the same pass is still summing), so that stage 6 -- one chain evaluation, the bf16
 * [V][32] dy5 tensor, one launch -- disappears:
 *   dva_chain_score_l6_stats   dva_chain_score_stats + acc5 fp32 [2][32][32] = P2 | Q2 and vec5 fp64 [4][32] = e1 | e2 | n5 | q5
 *                              (both caller-zeroed; see csrc/chain_bwd.hip score_l6_kernel)
 *   dva_chain_l6_consts        stats5 fp64 [64] = sum dy5 | sum dy5 z5 from them, sm6 (dva_bn_bwd_consts of stats6), bn6 and
 *                              the fp32 weight W6 [32][32]
 *   dva_chain_bwd_layer5_merged  stage 5 of dva_chain_bwd_layer starting from grad_scores [V][4] and sm6 instead of the dy5
 *                              row: also writes dW6 [32][32] (caller-zeroed) */
int dva_chain_score_l6_stats(const float* x_map, const int32_t* view_point, const float* u, const void* tiles,
                             const int32_t* n_tiles, const void* ops, const float* bn1, const float* bn2,
                             const float* bn5, const float* bn6, const float* grad_scores, double* stats6, float* dWs,
                             float* dbs, float* acc5, double* vec5, int32_t G, int64_t n_views, int64_t n_points,
                             void* stream);
int dva_chain_l6_consts(const float* sm6, const float* bn6, const float* W6, const float* acc5, const double* vec5,
                        double* stats5, void* stream);
int dva_chain_bwd_layer5_merged(const float* x_map, const int32_t* view_point, const float* u, const void* tiles,
                                const int32_t* n_tiles, const void* ops, const float* bn1, const float* bn2,
                                const float* bn5, const float* bn6, const float* sm5, const float* sm6,
                                const float* grad_scores, void* da_out, float* dW5, float* dW6, float* du, double* stats2,
                                int32_t G, int64_t n_views, int64_t n_points, void* stream);
/* Backward of the softmax / weighted sum / gate of dva_chain_attn_fwd from the scores it left (scores fp32 [V][4]):
 * no chain evaluation.  grad_out / out bf16 [N][C] (out = the forward result; only read for points with more than 32
 * views).  Outputs: grad_scores fp32 [V][4] (columns >= G zero), view_rec = V packed 16-byte records {int32 point id |
 * gate * attention of groups 0..3 as bf16 | 4 unused bytes} (what dva_view_gather_rows_grad_rec16 consumes: the rows
 * gradient is rounded to bf16, its weights travel as bf16), grad_gate_wb fp32 [2 G] (caller-zeroed, d gate_w |
 * d gate_b; nullable with gating off). */
int dva_chain_attn_bwd(const float* scores, const int32_t* view_point, const void* tiles, const int32_t* n_tiles,
                       const void* rows, const int32_t* row_idx, const int64_t* ptr, const float* gate_w,
                       const float* gate_b, const void* grad_out, const void* out, float* grad_scores, void* view_rec,
                       float* grad_gate_wb, int64_t n_points, int64_t n_views, int64_t n_rows, int32_t C, int32_t G,
                       int32_t scaling, float eps, void* stream);
/* A/B variant of round 4 (profiles/r04*_rows_grad_planrec_ab.json): dva_chain_attn_bwd that writes the
 * 16-byte record of view v to slot rec_pos[v] -- its position in the row plan, rec_pos = dva_plan_inverse(perm) -- so that
 * dva_view_gather_rows_grad_rec16(perm = NULL) streams the records instead of gathering them. */
int dva_chain_attn_bwd_planrec(const int32_t* rec_pos, const float* scores, const int32_t* view_point, const void* tiles,
                               const int32_t* n_tiles, const void* rows, const int32_t* row_idx, const int64_t* ptr,
                               const float* gate_w, const float* gate_b, const void* grad_out, const void* out,
                               float* grad_scores, void* view_rec, float* grad_gate_wb, int64_t n_points, int64_t n_views,
                               int64_t n_rows, int32_t C, int32_t G, int32_t scaling, float eps, void* stream);
/* inv[perm[i]] = i for a permutation of n_views entries (the row plan's `perm`). */
int dva_plan_inverse(const int32_t* perm, int32_t* inv, int64_t n_views, void* stream);
/* dva_chain_attn_bwd for fp32 value rows (the no-autocast path: ops.view_gather_attention backward when the scores
 * are fp32 [V][4]): rows / grad_out / out fp32, C in {32, 64, 128, 256}, view_rec = fp32 [V][8] records
 * {point id (int bits) | gate * attention per group | pad} as dva_view_gather_rows_grad reads them (rec_stride 8). */
int dva_chain_attn_bwd_f32(const float* scores, const int32_t* view_point, const void* tiles, const int32_t* n_tiles,
                           const float* rows, const int32_t* row_idx, const int64_t* ptr, const float* gate_w,
                           const float* gate_b, const float* grad_out, const float* out, float* grad_scores,
                           float* view_rec, float* grad_gate_wb, int64_t n_points, int64_t n_views, int64_t n_rows,
                           int32_t C, int32_t G, int32_t scaling, float eps, void* stream);
/* QKVBimodalCSRPool in ONE view kernel (round 4): dva_chain_attn_fwd with the KEY layer as the chain's last layer (ops
 * prepared with Ws = K.weight, G = 32; key_bias fp32 [32]) and the compatibilities scale * <key, queries[point]> per query-key
 * group as the scores (queries fp32 [N][32] in the keys' position order, 16-byte aligned; G in {1, 2, 4} = query-key groups =
 * attention groups).  scores_out (nullable) fp32 [V][4] receives the compatibilities, keys_out (nullable) bf16 [V][32] the
 * key rows (position order) dva_qkv_dquery reads back in the backward; everything else as dva_chain_attn_fwd. */
int dva_chain_attn_fwd_keys(const float* x_map, const int32_t* view_point, const float* u, const void* tiles,
                            const int32_t* n_tiles, const void* ops, const float* bn1, const float* bn2,
                            const float* bn5, const float* bn6, const float* key_bias, const float* queries, float scale,
                            const void* rows, const int32_t* row_idx, const int64_t* ptr, const float* gate_w,
                            const float* gate_b, void* out, float* scores_out, void* keys_out, int64_t n_points,
                            int64_t n_views, int64_t n_rows, int32_t C, int32_t G, int32_t scaling, float eps,
                            void* stream);
/* Score layer backward + the statistics of the BatchNorm-6 backward (one chain evaluation per view):
 * stats6 += S1 | S2 of layer 6 with dy6 = leaky'(t6) Ws^T grad_scores (t6 = the folded layer-6 product, the
 * pre-activation the forward's activation saw), dWs fp32 [G][32] / dbs fp32 [G] (caller-zeroed) += the gradient of the
 * score layer. */
int dva_chain_score_stats(const float* x_map, const int32_t* view_point, const float* u, const void* tiles,
                          const int32_t* n_tiles, const void* ops, const float* bn1, const float* bn2,
                          const float* bn5, const float* bn6, const float* grad_scores, double* stats6, float* dWs,
                          float* dbs, int32_t G, int64_t n_views, int64_t n_points, void* stream);
/* dva_chain_score_stats / stage 6 of dva_chain_bwd_layer starting from the a2 row of dva_chain_stats_a2 instead of x_map
 * (64 instead of 32 bytes per view read, layers 1 and 2 not evaluated; same results bit for bit). */
int dva_chain_score_stats_a2(const void* a2, const int32_t* view_point, const float* u, const void* tiles,
                             const int32_t* n_tiles, const void* ops, const float* bn5, const float* bn6,
                             const float* grad_scores, double* stats6, float* dWs, float* dbs, int32_t G,
                             int64_t n_views, int64_t n_points, void* stream);
int dva_chain_bwd_layer6_a2(const void* a2, const int32_t* view_point, const float* u, const void* tiles,
                            const int32_t* n_tiles, const void* ops, const float* bn5, const float* bn6,
                            const float* sm6, const float* grad_scores, void* da_out, float* dW, double* stats,
                            int32_t G, int64_t n_views, int64_t n_points, void* stream);
/* The same pass below the KEY layer of QKVBimodalCSRPool (ops prepared with G = 32): the gradient of a view's key row is
 * built in registers, dK'[v][i] = scale grad_compat[v][g(i)] queries[point(v)][i] (grad_compat fp32 [V][4], queries fp32
 * [N][32] in position order, G in {1, 2, 4} query-key groups); dWk fp32 [32][32] / dbk fp32 [32] (caller-zeroed) += the
 * gradient of the key layer.  dva_chain_bwd_layer6_keys: stage 6 of dva_chain_bwd_layer below that layer
 * (da_out = dy5 bf16 [V][32], dW = dW6, stats += S of layer 5). */
int dva_chain_score_stats_keys(const float* x_map, const int32_t* view_point, const float* u, const void* tiles,
                               const int32_t* n_tiles, const void* ops, const float* bn1, const float* bn2,
                               const float* bn5, const float* bn6, const float* grad_compat, const float* queries,
                               double* stats6, float* dWk, float* dbk, int32_t G, float scale, int64_t n_views,
                               int64_t n_points, void* stream);
int dva_chain_bwd_layer6_keys(const float* x_map, const int32_t* view_point, const float* u, const void* tiles,
                              const int32_t* n_tiles, const void* ops, const float* bn1, const float* bn2,
                              const float* bn5, const float* bn6, const float* sm6, const float* grad_compat,
                              const float* queries, void* da_out, float* dW, double* stats, int32_t G, float scale,
                              int64_t n_views, int64_t n_points, void* stream);
/* One backward pass of the chain between two BatchNorm-backward barriers ("sm" = fp32 [2][32] = S1/M | S2/M of
 * the pass's own layer, zeros with running statistics; dW / P / stats caller-zeroed, accumulated with
 * atomics).  Each pass re-evaluates the chain from x_map up to its own layer; the gradient w.r.t. the BatchNorm
 * output of the layer below (dy = leaky'(.) da, the derivative of the activation already applied with the sign of
 * the pre-activation the forward saw) is handed from pass to pass as da_out -> da_in, bf16 [V][32] (64 bytes per
 * view, in the lane order of the kernels: opaque to the caller):
 *   stage 6: grad_scores -> dW [32][32] = dW6, stats += S of layer 5, da_out = dy5                         (sm6)
 *   stage 5: da_in = dy5 -> dW [32][64] (first 32 columns) = dW5 per-view half, du fp32 [N][32] = gradient of u
 *            (written for seen points), stats += S of layer 2 (view part), da_out = dy2                (sm5)
 *   stage 2: da_in = dy2 (+ dpooled = the dpooled_dy of dva_chain_route_stats, routed to the arg views of
 *            dva_chain_stats2) -> dW [32][32] = dW2,
 *            P fp32 [32][20] += sum_v dy1 [x_hi (8) | x_lo (8) | 1 | unused]^T; the statistics of layer 1 follow
 *            from P (dva_chain_stats1; stats may be NULL)                                                (sm2) */
int dva_chain_bwd_layer(int32_t stage, const float* x_map, const int32_t* view_point, const float* u,
                        const void* tiles, const int32_t* n_tiles, const void* ops, const float* bn1,
                        const float* bn2, const float* bn5, const float* bn6, const float* sm2, const float* sm5,
                        const float* sm6, const float* grad_scores, const int32_t* arg, const float* dpooled,
                        const void* da_in, void* da_out, float* dW, float* du, float* P,
                        double* stats, int32_t G, int64_t n_views, int64_t n_points, void* stream);
/* ---- The chain in fp32: DeepSetFeat + E_score for fp32 features OUTSIDE torch.autocast (the reference's default,
 * models/base_model.py:244), replacing the stored-activation passes dva_deepset_* (13 stored [V, 32] tensors).
 * Same tiles, statistics and hand-over rules as dva_chain_* above; every product runs on the fp32 matrix cores
 * (v_mfma_f32_32x32x2_f32: exact fp32 fma chains), BatchNorm + LeakyReLU in fp32 on the accumulators (no BatchNorm
 * folding; leaky' follows the sign of the plain pre-activation), gradient rows between the passes fp32 [V][32].
 * These passes are bound by the matrix pipe, not by HBM: the raw outputs of layers 2 and 5 (z2, z5: fp32 [V][32])
 * stay in HBM and every pass starts from them instead of re-evaluating the chain from x_map.
 * bn tables: fp32 [>= 4][32] mean | invstd | gamma | beta.  The attention (softmax, value gather, weighted sum, gate)
 * stays in dva_view_gather_attention_*; these entries produce the scores fp32 [V][4] (columns >= G zero) and consume
 * their gradient fp32 [V][4].
 *   dva_chain3_prep         ops: 26 KiB device buffer (26 operand blocks of 64 float4)
 *   dva_chain3_stats2       x_map -> z2 (stored); stats double[64] += sum z2 | sum z2^2; zstar / arg as dva_chain_stats2
 *   dva_chain3_stats        layer 5: rows_in = z2 (bn = bn2) -> z5 = W5a act(BN2(z2)) + u[point] (stored), stats of z5;
 *                           layer 6: rows_in = z5 (bn = bn5) -> stats of z6 (view_point, u, z5 unused, may be NULL)
 *   dva_chain3_scores       z5 -> scores
 *   dva_chain3_score_stats  z5, grad_scores -> dWs [G][32], dbs [G], stats6 (S of the BatchNorm-6 backward)
 *   dva_chain3_bwd_layer    bn_lo / bn_hi = tables of the lower / upper layer of the pass, sm = S / M of the upper one
 *     stage 6: z_rows = z5, grad_scores -> dW (= dW6 [32][32]), stats (S of layer 5), da_out = dy5      (bn5, bn6)
 *     stage 5: z_rows = z2, da_in = dy5, view_point, u -> dW (= dW5 [32][64], view half), du [N][32] (caller-zeroed),
 *              stats (S of layer 2, view part), da_out = dy2                                              (bn2, bn5)
 *     stage 2: x_map, z_rows = z2, da_in = dy2, view_point, arg, dpooled (dva_chain_route_stats) -> dW (= dW2),
 *              P fp32 [32][20] = sum dy1 [x (8) | 0 (8) | 1 | 0]^T (dva_chain_stats1 / dva_chain_dw1, exact_w1 = 1)
 *                                                                                                          (bn1, bn2)
 * dva_chain_moments / dva_chain_stats1 / dva_chain_dw1 (exact_w1 = 1), dva_chain_pooled and dva_chain_route_stats
 * are shared with the bf16 chain. */
int dva_chain3_prep(const float* W1, const float* W2, const float* W5, int32_t ld5, const float* W6, const float* Ws,
                    int32_t G, void* ops, void* stream);
int dva_chain3_stats2(const float* x_map, const int32_t* view_point, const void* tiles, const int32_t* n_tiles,
                      const void* ops, const float* bn1, const float* gamma2, double* stats, float* zstar,
                      int32_t* arg, float* z2, int64_t n_views, void* stream);
int dva_chain3_stats(int32_t layer, const float* rows_in, const int32_t* view_point, const float* u,
                     const void* tiles, const int32_t* n_tiles, const void* ops, const float* bn, float* z5,
                     double* stats, int64_t n_views, int64_t n_points, void* stream);
int dva_chain3_scores(const float* z5, const void* tiles, const int32_t* n_tiles, const void* ops, const float* bn5,
                      const float* bn6, const float* score_bias, int32_t G, float* scores, int64_t n_views,
                      void* stream);
int dva_chain3_score_stats(const float* z5, const void* tiles, const int32_t* n_tiles, const void* ops,
                           const float* bn5, const float* bn6, const float* grad_scores, double* stats6, float* dWs,
                           float* dbs, int32_t G, int64_t n_views, void* stream);
int dva_chain3_bwd_layer(int32_t stage, const float* x_map, const float* z_rows, const int32_t* view_point,
                         const float* u, const void* tiles, const int32_t* n_tiles, const void* ops,
                         const float* bn_lo, const float* bn_hi, const float* sm, const float* grad_scores,
                         const int32_t* arg, const float* dpooled, const float* da_in, float* da_out, float* dW,
                         float* du, float* P, double* stats, int64_t n_views, int64_t n_points, void* stream);
/* The per-point set branch of the fp32 chain: dva_chain_set_* on the fp32 matrix cores (ops: 24 KiB). */
int dva_chain3_set_prep(const float* Wsa, int32_t ld_sa, const float* Wsb, const float* Wc, int32_t ld_c, void* ops,
                        void* stream);
int dva_chain3_set_fwd(int32_t stage, const float* pooled, const int64_t* ptr, const float* w33, const void* ops,
                       const float* bn_s1, const float* bn_s2, float* u, double* stats, int64_t n_points,
                       void* stream);
int dva_chain3_set_bwd(int32_t stage, const float* pooled, const int64_t* ptr, const float* w33, const void* ops,
                       const float* bn_s1, const float* bn_s2, const float* sm_s1, const float* sm_s2,
                       const float* du, float* dpooled, float* dW, int32_t ld_dw, float* dw33, double* stats,
                       int64_t n_points, void* stream);
/* stats += S1 | S2 of layer 2, per-point part: sum over the seen points of leaky'(BN2(zstar)) dpooled (x z_hat);
 * dpooled_dy fp32 [N][32] = leaky'(BN2(zstar)) dpooled (0 for unseen points): the `dpooled` of stage 2. */
int dva_chain_route_stats(const float* zstar, const float* dpooled, const float* bn2, const int64_t* ptr,
                          double* stats, float* dpooled_dy, int64_t n_points, void* stream);

/* 'concatenation' fusion (modules/multimodal/fusion.py:7-53: torch.cat((x_main, x_mod), dim=-1)) with the dtype
 * promotion of torch.cat folded in: x_main fp32 [N][C_main], x_mod fp32 / bf16 [N][C_mod] -> out fp32
 * [N][C_main + C_mod]; bwd: grad_out -> grad_main fp32, grad_mod in mod_dtype.  C_main, C_mod multiples of 4. */
int dva_concat_cast_fwd(const float* x_main, const void* x_mod, float* out, int64_t N, int32_t C_main, int32_t C_mod,
                        int32_t mod_dtype, void* stream);
int dva_concat_cast_bwd(const float* grad_out, float* grad_main, void* grad_mod, int64_t N, int32_t C_main,
                        int32_t C_mod, int32_t mod_dtype, void* stream);
/* Measurement helper (bench.py): float4 grid-stride device copy of nbytes (multiple of 16) -- the practical HBM
 * ceiling (read + write) beside which the roofline fractions are quoted. */
int dva_copy_ceiling(const void* src, void* dst, int64_t nbytes, void* stream);
/* out[p, :] = 0 for every point p without views (ptr[p + 1] == ptr[p]); rows of the other points are not touched.
 * out: n_points rows of row_bytes bytes (multiple of 16, 16-byte aligned).  The pooled features of unseen points are exact
 * zeros (reference modules/multimodal/pooling.py:870, torch_scatter's empty segments); the view kernels write only the
 * points that have views. */
int dva_zero_unseen_rows(const int64_t* ptr, void* out, int64_t n_points, int64_t row_bytes, void* stream);

/* ------------------------------------------------------------------------------------------ *
 * Voxel parent index after a strided sparse 3D convolution.  Replaces the torchsparse (v1.1.0, not in the
 * reference tree) `sphashquery(sphash(in_coords), sphash(out_coords))` call of
 * modules/multimodal/modules.py:176-198 together with the flooring of the spatial columns:
 * idx[i] = j such that out_coords[j] == in_coords[i] with every column except batch_col floored to a
 * multiple of stride_out (floor towards -inf), or -1 when no output voxel has these coordinates.
 * in_coords int32 [n_in, 4], out_coords int32 [n_out, 4] (16-byte aligned; torchsparse layout: x, y, z,
 * batch => batch_col = 3; batch_col = -1 floors all four columns), idx int64 [n_in].  Exact (full
 * coordinate compare in an open-addressing table), deterministic; duplicate out rows -> smallest j.
 * ------------------------------------------------------------------------------------------ */
int64_t dva_voxel_parent_workspace_bytes(int64_t n_out);
int dva_voxel_parent_index(const int32_t* in_coords, int64_t n_in, const int32_t* out_coords, int64_t n_out,
                           int32_t stride_out, int32_t batch_col, int64_t* idx, void* workspace,
                           int64_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------ *
 * Sparse 3D convolution on voxel tensors.  Replaces torchsparse v1.1.0 `Conv3d` /
 * transposed `Conv3d` (not in the reference tree) behind modules/SparseConv3d/nn/torchsparse.py:6-40, used by
 * ResNetDown / ResNetUp / ResBlock (modules/SparseConv3d/modules.py:10-220).
 *
 * dva_voxel_kernel_map: nbr[k * n_dst + j] = i such that src_coords[i] == dst_coords[j] + offsets[k] on the
 * first three columns with equal fourth (batch) column, else -1 (torchsparse `sphashquery(sphash(dst,
 * offsets), sphash(src))`).  Coordinates int32 [n, 4], 16-byte aligned; offsets int32 [K, 3] on the device.
 * Workspace: dva_voxel_parent_workspace_bytes(n_src).
 *
 * dva_sparse_conv_apply: out[j, :] = bias + sum_k x[nbr[k][j], :] @ Wk   (missing neighbours contribute 0).
 *   mode 0: W = fp32 [K, Cin, Cout], Wk = W[k]            (forward; transposed convolution with its own map)
 *   mode 1: W = fp32 [K, Cout, Cin], Wk = W[k]^T          (input gradient, with the transposed kernel map)
 * x [n_src, Cin], out [n_dst, Cout] in `dtype` (DVA_F32: fp32 accuracy through a 3-term bf16 split on the
 * matrix cores; DVA_BF16: bf16 operands, fp32 accumulation), bias fp32 [Cout] nullable.  Cin and Cout must be
 * multiples of 16 (the host pads), n_src >= 1.  Workspace: dva_sparse_conv_workspace_bytes (re-packed weights).
 * Every output row is written exactly once: deterministic, no atomics.
 *
 * dva_sparse_conv_wgrad: grad_W[k][a][b] = sum_j x[nbr[k][j]][a] * grad_out[j][b], fp32 [K, Cin, Cout],
 * zeroed by the call, accumulated with fp32 atomics (summation order not fixed).
 * ------------------------------------------------------------------------------------------ */
int dva_voxel_kernel_map(const int32_t* src_coords, int64_t n_src, const int32_t* dst_coords, int64_t n_dst,
                         const int32_t* offsets, int32_t K, int32_t* nbr, void* workspace,
                         int64_t workspace_bytes, void* stream);
int64_t dva_sparse_conv_workspace_bytes(int32_t K, int32_t Cin, int32_t Cout, int32_t dtype);
int dva_sparse_conv_apply(const void* x, const int32_t* nbr, const float* W, const float* bias, void* out,
                          int64_t n_src, int64_t n_dst, int32_t K, int32_t Cin, int32_t Cout, int32_t mode,
                          int32_t dtype, void* workspace, int64_t workspace_bytes, void* stream);
int dva_sparse_conv_wgrad(const void* x, const int32_t* nbr, const void* grad_out, float* grad_W, int64_t n_src,
                          int64_t n_dst, int32_t K, int32_t Cin, int32_t Cout, int32_t dtype, void* stream);

/* ------------------------------------------------------------------------------------------ *
 * Neighbourhood-based mapping features (core/data_transform/multimodal/image.py:431-612): exact k nearest
 * neighbours of every point among all points (replaces KeOps `((x_i - x_j)**2).sum(2).argKmin(k, dim=1)`,
 * :506-507; pykeops 1.4.2 is not in the reference tree) and the per-view occlusion counts (:560-599).
 * ------------------------------------------------------------------------------------------ */
/* neighbors int32 [n, k] (k <= 128), dist2 fp32 [n, k] (nullable): ascending by (d2, index) with
 * d2 = ((dx*dx + dy*dy) + dz*dz) in fp32; the point itself is its own first neighbour; -1 / inf when
 * n < k.  bbox = fp32[6] device array (min xyz | max xyz of the cloud); cell = grid cell size;
 * (max - min) / cell must stay below 2^20 per axis.  One call searches at most max_shell Chebyshev shells
 * of cells around each query and finishes every query whose k-th distance is provably final within them
 * (or all queries, when the shells cover the whole cloud); `done` (uint8 [n], nullable) is read to skip
 * finished queries and set for those finished here, so sparse regions are handled by calling again with
 * a coarser cell (host: deepviewagg_amd.ops.knn, cell x4 per level).  The result is exact for any cell. */
int64_t dva_knn_workspace_bytes(int64_t n);
int dva_knn(const float* xyz, int64_t n, const float* bbox, float cell, int32_t k, int32_t max_shell,
            uint8_t* done, int32_t* neighbors, float* dist2, void* workspace, int64_t workspace_bytes,
            void* stream);
/* out fp32 [n_views, n_k]: (1 + #{i < k_list[c] : neighbors[p(v), i] is seen by image(v)}) / (k_list[c] + 1)
 * (k_list ascending, last <= k).  view_point int32 [n_views] (dva_csr_expand), images int64 [n_views],
 * bits = scratch uint64 [n_points * ceil(n_images / 64)]. */
int dva_view_occlusion(const int32_t* view_point, const int64_t* images, int64_t n_views, int64_t n_points,
                       int32_t n_images, const int32_t* neighbors, int32_t k, const int32_t* k_list,
                       int32_t n_k, uint64_t* bits, float* out, void* stream);

/* ------------------------------------------------------------------------------------------ *
 * Lexicographic integer keys.  Replace utils/multimodal.py:36-94 (lexargsort / lexargunique on a
 * composite int64 key, :97-179 CompositeTensor, :253-323 lex ops).
 * ------------------------------------------------------------------------------------------ */

/* Bytes of temporary storage the two functions below need for n keys. */
int64_t dva_lex_workspace_bytes(int64_t n);
/* order[i] = index of the i-th smallest key; STABLE (ties keep input order), which is one of the
 * orders numpy's unstable argsort (multimodal.py:316) may return. keys_sorted nullable. */
int dva_argsort_i64(const int64_t* keys, int64_t n, int64_t* order, int64_t* keys_sorted,
                    void* workspace, int64_t workspace_bytes, void* stream);
/* first[j] = smallest input index carrying the j-th smallest distinct key (numpy
 * unique(return_index=True), multimodal.py:310); *n_unique written to device memory. */
int dva_argunique_i64(const int64_t* keys, int64_t n, int64_t* first, int64_t* n_unique_dev,
                      void* workspace, int64_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------ *
 * Mapping build (point -> pixel visibility).  Replaces SplattingVisibility.__call__
 * = camera_projection_cpu + visibility_from_splatting_cpu + postprocess_features
 * (core/multimodal/visibility.py:478-538, :1073-1195, :1548-1582, glue :1699-1757).
 * The CPU/numba path is the semantic reference (README.md:122-123).
 * ------------------------------------------------------------------------------------------ */
typedef struct dva_camera {
  int32_t model;        /* DVA_CAM_* */
  int32_t img_w, img_h; /* projection map size (proj_size) */
  int32_t crop_top, crop_bottom;
  float r_min, r_max;   /* informative copies; the range test uses r_min_d / r_max_d */
  float img_xyz[3];     /* camera centre */
  /* Row-major 3x3 / 3-vector the projection multiplies by, prepared by the host exactly like the
   * reference does per image (scalar work): equirect: rot = M_o.M_p.M_k from opk
   * (visibility.py:57-90), v = (xyz - img_xyz).rot^T; 'scannet': (rot, trans) =
   * inv(extrinsic)[:3,:3|3] (:232-236), p = rot.xyz + trans; kitti perspective / fisheye:
   * (rot, trans) = extrinsic[:3,:3|3] (:238-242, :304-308), p = rot^T.(xyz - trans). */
  float rot[9];
  float trans[3];
  float fx, fy, mx, my; /* pinhole intrinsics [0][0], [1][1], [0][2], [1][2] */
  float fisheye[7];     /* xi, k1, k2, gamma1, gamma2, u0, v0 */
  double r_min_d, r_max_d; /* float64 scalars (numba promotes the float32 distances to compare) */
  double voxel, k_swell, d_swell;
  int32_t exact;
} dva_camera;

/* Bytes of device workspace dva_visibility needs for n candidate points and this camera. */
int64_t dva_visibility_workspace_bytes(const dva_camera* cam, int64_t n);

/* One image.  xyz fp32 [n,3] candidate points IN CALLER ORDER (tie-breaks depend on it),
 * mask nullable uint8 [img_w, img_h] (indexed [x][y], visibility.py:427-432).
 * Outputs (capacity n each when cam->exact, else max(n, img_w * cropped_h) since every covered pixel is
 * emitted; *n_out_dev = q written on device), in the reference's output order
 * (x-major, then y; visibility.py:1190-1195):
 *   idx   int64[q] index into xyz          x_pix,y_pix int64[q]   depth fp32[q]
 *   x_proj,y_proj fp64[q] float projection of the surviving points (for mapping features)
 */
int dva_visibility(const float* xyz, int64_t n, const dva_camera* cam, const uint8_t* mask,
                   int64_t* idx, int64_t* x_pix, int64_t* y_pix, float* depth, double* x_proj,
                   double* y_proj, int64_t* n_out_dev, void* workspace, int64_t workspace_bytes,
                   void* stream);

/* camera_projection alone (reference core/multimodal/visibility.py:478-538: range, field-of-view / crop / mask cull,
 * float projection), survivors in candidate order: idx int64[m], depth fp32[m], x_proj / y_proj fp64[m] (capacity n),
 * *n_out_dev = m.  The start of the visibility models other than the splatting one (DepthBasedVisibility,
 * BiasuttiVisibility: visibility.py:1356-1496, :1779-1803).  Workspace: dva_visibility_workspace_bytes(cam, n). */
int dva_camera_projection(const float* xyz, int64_t n, const dva_camera* cam, const uint8_t* mask, int64_t* idx,
                          float* depth, double* x_proj, double* y_proj, int64_t* n_out_dev, void* workspace,
                          int64_t workspace_bytes, void* stream);

/* B images of ONE setting in one set of launches (reference core/data_transform/multimodal/image.py:192-428 loops
 * over the images; core/multimodal/visibility.py:1073-1195 per image).  cam0 (host) = the camera of image 0: projection
 * size, crops, model, exact, splatting parameters must be the same for all images; cams_dev = device array of the
 * n_images cameras (poses / intrinsics per image).  Outputs as dva_visibility, rows of all images concatenated
 * image-major (capacity n_images x the single-image capacity), row_ptr int64 [n_images + 1] = first row of every
 * image (device), *n_out_dev = total.  No host synchronisation inside. */
int64_t dva_visibility_batch_workspace_bytes(const dva_camera* cam0, int64_t n, int32_t n_images);
int dva_visibility_batch(const float* xyz, int64_t n, const dva_camera* cam0, const dva_camera* cams_dev,
                         int32_t n_images, const uint8_t* mask, int64_t* idx, int64_t* x_pix, int64_t* y_pix,
                         float* depth, double* x_proj, double* y_proj, int64_t* row_ptr, int64_t* n_out_dev,
                         void* workspace, int64_t workspace_bytes, void* stream);
/* dva_mapping_features for the rows of dva_visibility_batch (the camera of a row follows from row_ptr);
 * row_image int32 [q] nullable receives the image of every row. */
int dva_mapping_features_batch(const float* xyz, const int64_t* idx, const float* depth, const double* y_proj,
                               const float* linearity, const float* planarity, const float* scattering,
                               const float* normals, const dva_camera* cams_dev, const int64_t* row_ptr,
                               int32_t n_images, int64_t q, float* features, int32_t* row_image, int32_t* n_cols,
                               void* stream);

/* Mapping features of the q surviving points (visibility.py:1548-1582); nullable inputs drop
 * their column exactly like the reference. features fp32 [q, n_cols], column order:
 * depth, linearity, planarity, scattering, orientation, pixel height. Returns n_cols via *n_cols. */
int dva_mapping_features(const float* xyz, const int64_t* idx, const float* depth,
                         const double* y_proj, const float* linearity, const float* planarity,
                         const float* scattering, const float* normals, const dva_camera* cam,
                         int64_t q, float* features, int32_t* n_cols, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DVA_H_ */

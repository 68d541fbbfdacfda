/*
 * ofx.h -- C ABI of libofx.so: the MI355X (gfx950) kernels behind OctFusion's
 * denoising hot path.
 *
 * The reference (octree-nn/octfusion) has no FFI of its own: its boundary is
 * Python nn.Module forward() calls that bottom out in torch ops.  This header
 * is the C-ABI drop-in point *under* those modules: every entry point names the
 * reference code it replaces (paths relative to the reference tree).  All
 * pointers are DEVICE pointers unless the name ends in _host; `stream` is a
 * hipStream_t passed as void*.  Every function returns 0 (OFX_OK) or a
 * negative OFX_E* code; nothing throws, nothing allocates device memory,
 * nothing synchronises (safe under hipGraph capture) unless stated.
 *
 * Layouts: features are row-major fp32 [rows, C] with an explicit leading
 * dimension (floats); octree keys int64; children / indices int32.
 */
#ifndef OFX_H_
#define OFX_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OFX_OK 0
#define OFX_EINVAL (-1)   /* bad argument (shape / alignment / null)          */
#define OFX_ELAUNCH (-2)  /* HIP launch error (hipGetLastError != success)    */
#define OFX_ENODEV (-3)   /* no gfx950 device visible                         */

#define OFX_ACT_NONE 0
#define OFX_ACT_SILU 1
#define OFX_ACT_GELU 2

#define OFX_MAX_DEPTH 16

int ofx_version(void);
const char* ofx_status_string(int status);
/* 0 when a HIP device is present and is gfx950; OFX_ENODEV otherwise. */
int ofx_device_check(void);
/* sha256 (16 hex digits) of every source, header and compile flag this library was built from
 * (octfusion_amd/build.py); the Python binding refuses a library whose hash differs from the tree.
 * ofx_build_ablation: 1 for a -DOFX_ABLATION build (timing ablations compiled in), 0 for a product build. */
const char* ofx_build_hash(void);
int ofx_build_ablation(void);

/* ------------------------------------------------------------------ fp16x3 range guard (precision 3, the default)
 * Operands travel as fp16 hi + lo pairs (22 significand bits).  What that covers and what it does not:
 *  - WEIGHTS are scaled at pack time by a per-tensor power of two so that max|w| lands in [2^14, 2^15) (the kernels
 *    multiply the accumulators by the inverse in the epilogue FMA that adds the bias anyway -- exact, free): every
 *    weight within 2^-18 of the tensor's largest keeps 22 bits, smaller ones an ABSOLUTE error of 2^-40 max|w| --
 *    fp32-class against the output whatever the tensor's own magnitude (a near-zero-initialised projection that
 *    learned 1e-4-scale weights is as exact as an O(1) one);
 *  - ACTIVATIONS are not scaled.  Below |x| ~ 2^-3 the lo half is a denormal fp16: an absolute error floor of
 *    2^-25 ~ 3e-8 per operand (relative 2^-22 above it).  Normalised activations (every GraphConv input that comes
 *    from a GroupNorm) are O(1): there the floor is fp32's own.  Beyond +-65504 an operand does NOT saturate: its
 *    halves become +-Inf and every product it enters is NaN, so an un-normalised operand outside the range (network
 *    inputs, pool / unpool outputs, the residual stream into the 1x1 skip convolutions) makes the result loudly
 *    non-finite instead of plausibly wrong.  words[0] += the operands ofx_planes_split found outside the range
 *    (exact count, diagnostics).  The caller checks finiteness of a stage's result and the word once per batch of
 *    launches (octfusion_amd.ops.raise_on_range_error) and re-runs in bf16x3 (precision 0: fp32's exponent range,
 *    16 significand bits).
 * ofx_set_range_words: device pointer to >= 4 zeroed uint32 on the CURRENT device (NULL: counting off).  Sticky. */
int ofx_set_range_words(uint32_t* words);

/* Measurement aid (csrc/ofx_probe.hip; not on the operator path): the rate the matrix pipe SUSTAINS on the current
 * device with realistic fp16 hi / lo operand pairs from LDS and no memory traffic -- ~1.45 PFLOP/s at ~1.55 GHz on MI355X
 * against the 2.5 PFLOP/s data-sheet peak, which needs constant operands (power limit).  One launch of `steps`
 * 24-MFMA steps per wave, 8 waves per block, `blocks` blocks (0: one per compute unit); the caller times it.
 * flops = blocks * 8 * steps * 24 * 32768; ticks[0] = shader clocks block 0 spent in the loop. */
int ofx_probe_mfma_sustained(int steps, int blocks, float* sink, unsigned long long* ticks, void* stream);

/* ------------------------------------------------------------------ scans */
/* Exclusive prefix sum of n int32 values into out[0..n] (out[n] = total).
 * ws: workspace of at least ofx_scan_ws_bytes(n) bytes.
 * Replaces torch.cumsum uses in ocnn Octree.octree_split and
 * dual_octree.py:265-271 (remap_node_idx). */
size_t ofx_scan_ws_bytes(int64_t n);
int ofx_scan_i32(const int32_t* in, int32_t* out, int64_t n, void* ws, void* stream);

/* ---------------------------------------------------------------- octree
 * Own octree container ops (ocnn.octree.Octree is third-party and absent;
 * semantics per SURVEY.md 8c).  Call sites replaced:
 * utils/util_dualoctree.py:225-273 (split2octree_small/large),
 * ldm_diffusion_util.py:318-325 (create_full_octree),
 * graph_vae.py:205-208,236-244 (octree_split / octree_grow). */

/* keys[i] = (b << 48) | i_local, children[i] = i for a full layer of depth d. */
int ofx_octree_full_layer(int depth, int batch_size, int64_t* keys, int32_t* children,
                          void* stream);
/* children[i] = label[i] ? rank : -1, where rank = excl_scan(label != 0).
 * scan_out must hold n+1 int32 (scan_out[n] = number of non-empty nodes). */
int ofx_octree_split(const int32_t* label, int64_t n, int32_t* children, int32_t* scan_out,
                     void* ws, void* stream);
/* keys_child[8*c + o] = ((key & m48) << 3 | o) | (key >> 48 << 48) for every
 * parent with children[i] = c >= 0. */
int ofx_octree_grow(const int64_t* keys_parent, const int32_t* children_parent, int64_t n_parent,
                    int64_t* keys_child, int32_t* children_child, void* stream);
/* split_small [B,8,S,S,S] (S = 2^full_depth) -> labels of the full layer
 * (any channel > 0) -- util_dualoctree.py:233-238. */
int ofx_split_small_label0(const float* split, int batch_size, int full_depth,
                           int32_t* label, void* stream);
/* labels of the 8 children of every non-empty full-layer node:
 * label1[8*c + j] = split[b, j, x, y, z] > 0 -- util_dualoctree.py:242-246. */
int ofx_split_small_label1(const float* split, int batch_size, int full_depth,
                           const int32_t* children, int32_t* label1, void* stream);
/* split_large [n, 8] -> label0[i] = any(split[i,:] > 0); label1[8*c + j] =
 * split[i, j] > 0 for non-empty i -- util_dualoctree.py:258-269. */
int ofx_split_large_label0(const float* split, int64_t n, int32_t* label0, void* stream);
int ofx_split_large_label1(const float* split, int64_t n, const int32_t* children,
                           int32_t* label1, void* stream);
/* ocnn.nn.octree2voxel at the FULL layer followed by permute(0,4,1,2,3)
 * (graph_unet_lr.py:176-177): data [B*8^d, C] -> vox [B, C, S, S, S]. */
int ofx_octree2voxel_cf(const float* data, int64_t ld, int C, int batch_size, int depth,
                        float* vox, void* stream);
/* inverse gather (graph_unet_lr.py:179-181): vox [B, C, S,S,S] -> data [B*8^d, C]. */
int ofx_voxel2octree_cf(const float* vox, int C, int batch_size, int depth, float* data,
                        int64_t ld, void* stream);

/* ------------------------------------------------------------ dual octree
 * Neighbour graph of the dual octree (replaces DualOctree.__init__,
 * dense_graph, sparse_graph, relative_dir, add_self_loops, remap_node_idx,
 * add_node_type/keyd/mask, sort_edges, calc_batch_id:
 * models/networks/dualoctree_networks/dual_octree.py:19-409).
 *
 * The tree is passed as the depth-concatenated arrays the reference itself
 * builds (dual_octree.py:42-44): child_all / key_all of length ncum[depth+1],
 * plus leafrank_all = per-depth exclusive scan of (child < 0).  nnum_host and
 * nnum_nempty_host are HOST arrays of depth+1 entries.
 *
 * Graph nodes of depth d are numbered [leaves of full_depth..d-1 | all nodes
 * of d] (dual_octree.py:265-271).  Edges come out grouped by (row, dir)
 * segment, i.e. already in the reference's sort_edges order: seg_ptr has
 * N_d*7+1 entries, col has E_d entries.  dir: 0:+z 1:-z 2:+y 3:-y 4:+x 5:-x
 * 6:self. */
typedef struct {
  int depth, full_depth, batch_size;
  const int32_t* child_all;
  const int64_t* key_all;
  const int32_t* leafrank_all;
  const int64_t* nnum_host;
  const int64_t* nnum_nempty_host;
} ofx_tree_t;

/* Point cloud -> octree -- replaces ocnn `Octree.build_octree` + `merge_octrees` at the reference call sites
 * models/octfusion_model_union.py:198-212 and models/octfusion_model_vae.py:133-141, and the 'ND' input feature of
 * the VAE encoder (dual_octree.py:343-360).  The whole batch is built top-down from ONE sorted key array:
 *   _keys: key[i] = batch << 48 | morton(trunc((p + 1) * 2^(depth-1))) (coordinates clamped to the cube),
 *          idx[i] = i; batch_id NULL -> every point belongs to element batch_const;
 *   _sort: stable radix sort of the (key, idx) pairs (ws: ofx_points_sort_ws_bytes(n) bytes);
 *   _label_from_points: label[j] = 1 iff a point lies in the cell of node j of depth d (node_keys as in
 *          ofx_octree_grow: batch << 48 | morton_d), for ofx_octree_split;
 *   _point_features: per node of the finest depth, over its points in input order: feat [nnum,4] =
 *          (normalize(sum normals), dot(frac(mean scaled position) - 0.5, normal)), zero rows for empty nodes
 *          (16-B aligned); avg_points / avg_normals [nnum,3] optional (ocnn octree.points / octree.normals). */
size_t ofx_points_sort_ws_bytes(int64_t n);
int ofx_points_keys(const float* pts, int64_t ldp, const int32_t* batch_id, int batch_const, int64_t n, int depth,
                    int64_t* keys, int32_t* idx, void* stream);
int ofx_points_sort(const int64_t* keys_in, const int32_t* idx_in, int64_t n, int64_t* keys_out, int32_t* idx_out,
                    void* ws, size_t ws_bytes, void* stream);
int ofx_octree_label_from_points(const int64_t* sorted_keys, int64_t n_pts, const int64_t* node_keys, int64_t nnum,
                                 int depth_pts, int d, int32_t* label, void* stream);
int ofx_octree_point_features(const int64_t* sorted_keys, const int32_t* sorted_idx, int64_t n_pts, const float* pts,
                              int64_t ldp, const float* normals, int64_t ldn, const int64_t* node_keys, int64_t nnum,
                              int depth, float* feat, float* avg_points, float* avg_normals, void* stream);
/* per-depth leaf ranks: leafrank_all[ncum[t] + j] = #leaves before j at depth t.
 * ws: at least ofx_tree_leafrank_ws_bytes(max_d nnum[d]) bytes. */
size_t ofx_tree_leafrank_ws_bytes(int64_t max_nnum);
int ofx_tree_leafrank(const int32_t* child_all, const int64_t* nnum_host, int depth,
                      int32_t* leafrank_all, void* ws, void* stream);
/* node attributes of graph depth d: batch_id (key>>48), node_type (depth -
 * full_depth), keyd (key | depth<<58), node_mask (leaf, or all-true at d).
 * Any output may be NULL. */
int ofx_graph_nodes(const ofx_tree_t* tree, int d, int32_t* batch_id, uint8_t* node_type,
                    int64_t* keyd, uint8_t* node_mask, void* stream);
/* pass 1: seg_cnt[r*7 + dir] = number of neighbours of graph node r through
 * face dir (dir 6: 1 if the node has any neighbour). */
int ofx_graph_count(const ofx_tree_t* tree, int d, int32_t* seg_cnt, void* stream);
/* pass 2: col[seg_ptr[r*7+dir] ...] = neighbour ids (seg_ptr = excl. scan of seg_cnt). */
int ofx_graph_fill(const ofx_tree_t* tree, int d, const int32_t* seg_ptr, int32_t* col,
                   void* stream);
/* CSR -> the reference's COO view: row[e], dir[e] (int64) for edge_idx / edge_dir. */
int ofx_graph_expand(const int32_t* seg_ptr, int64_t n_nodes, const int32_t* col,
                     int64_t* row_out, int64_t* col_out, int64_t* dir_out, void* stream);
/* nbr[r*7+dir] = the single neighbour of segment (r,dir), -1 if none, -2 if several. */
int ofx_graph_primary(const int32_t* seg_ptr, const int32_t* col, int64_t n_nodes, int32_t* nbr,
                      void* stream);
/* Reverse graph for the backward pass of GraphConv (autograd of index_select + scatter_mean,
 * models/networks/modules.py:205-213): reverse segment (c, dir) lists the forward edges with col == c in
 * direction dir; rev_row = their rows (sorted), rev_w = 1 / size of the forward segment (row, dir).
 * rev_cnt doubles as the fill cursor (it is zeroed and rewritten by _fill). */
int ofx_graph_reverse_count(const int32_t* seg_ptr, const int32_t* col, int64_t n_nodes, int32_t* rev_cnt,
                            void* stream);
int ofx_graph_reverse_fill(const int32_t* seg_ptr, const int32_t* col, int64_t n_nodes, const int32_t* rev_ptr,
                           int32_t* cursor, int32_t* rev_row, float* rev_w, void* stream);
/* weighted-graph variants of ofx_graph_primary / _multi_flag / _primary_ext: a segment counts as a single plain
 * source row only if it has one edge of weight exactly 1 */
int ofx_graph_primary_w(const int32_t* seg_ptr, const int32_t* col, const float* w, int64_t n_nodes, int32_t* nbr,
                        void* stream);
int ofx_graph_multi_flag_w(const int32_t* seg_ptr, const float* w, int64_t n_nodes, int32_t* flag, void* stream);
int ofx_graph_primary_ext_w(const int32_t* seg_ptr, const int32_t* col, const float* w, int64_t n_nodes,
                            const int32_t* rank, int32_t* nbr_ext, int32_t* multi_seg, void* stream);
/* Backward of GraphConv (training path; autograd of models/networks/modules.py:194-220).
 * _bwd_data: dx [n, cin] = fused gather-GEMM of dy over the REVERSE graph (weighted segment sums) with the
 *   transposed weights WpT = ofx_pack_weights of W^T stacked over directions (K = 7*cout, N = cin, cin_pack =
 *   cout, nt = 0).  nbr_rev / nbr_ext_rev / multi_seg come from the _w table builders above; aux: scratch of
 *   (n_multi + 1) * ldy floats.
 * _bwd_weight: dWp [Kp, cout] = col_data^T @ dy in the PACKED k order of ofx_pack_weights (k = dir*cin + c,
 *   zero rows up to pad32(7*cin), then the 7*nt node-type rows), exact fp32 MFMA, deterministic slice-ordered
 *   reduction.  ws holds the partial sums (and, for cin % 32 != 0, col-row chunks). */
int ofx_graphconv_bwd_data(const float* dy, int64_t ldy, int cout, int64_t n_nodes, const int32_t* nbr_rev,
                           const int32_t* rev_ptr, const int32_t* rev_row, const float* rev_w,
                           const int32_t* nbr_ext_rev, const int32_t* multi_seg, int64_t n_multi, float* aux,
                           const float* WpT, int64_t KpT, int cin, float* dx, int64_t ldx, void* ws, size_t ws_bytes,
                           void* stream);
int ofx_graphconv_bwd_weight(const float* x, int64_t ldx, int cin, int64_t n_nodes, const int32_t* nbr,
                             const int32_t* seg_ptr, const int32_t* col, const int32_t* nbr_ext,
                             const int32_t* multi_seg, int64_t n_multi, float* aux, const float* type_frac, int64_t ldt,
                             int nt_pad, const float* dy, int64_t ldy, int cout, float* dWp, int64_t Kp, void* ws,
                             size_t ws_bytes, void* stream);
/* Backward of DualOctreeGroupNorm (+ fused activation) -- training path, autograd of modules.py:291-326.
 * mean / rstd [B*C] are the forward statistics (ofx_gn_finalize).  sums [B*C*2] fp64 and coef [B*C*3] fp32 are
 * scratch.  Outputs dx [n, C], dgamma [C], dbeta [C]. */
int ofx_gn_backward(const float* x, int64_t ldx, const float* dy, int64_t ldy, int64_t n, int C,
                    const int32_t* batch_id, int batch_size, const float* count, int groups, float count_eps,
                    const float* mean, const float* rstd, const float* w, const float* bias, int act, double* sums,
                    float* coef, float* dx, int64_t lddx, float* dgamma, float* dbeta, void* stream);
/* out [K, N] = P^T Q for row-major P [rows, K], Q [rows, N] (weight gradients of the dense layers: dW = x^T dy).
 * K, N multiples of 4; exact fp32 MFMA, slice-ordered (deterministic) reduction of the partials held in ws. */
int ofx_gemm_tn_f32(const float* P, int64_t ldp, const float* Q, int64_t ldq, int64_t rows, int64_t K, int64_t N,
                    float* out, void* ws, size_t ws_bytes, void* stream);
/* Reverse tables for any tap table (the dense layers' 27-tap grid tables): nbr [n_out, ndir], valid sources in
 * [0, n_in); reverse CSR keyed by (source row, tap) with unit weights; and segment-generic versions of the
 * weighted table builders (nseg segments, sources in [0, n_src)). */
int ofx_table_reverse_count(const int32_t* nbr, int64_t n_out, int ndir, int64_t n_in, int32_t* rev_cnt, void* stream);
int ofx_table_reverse_fill(const int32_t* nbr, int64_t n_out, int ndir, int64_t n_in, const int32_t* rev_ptr,
                           int32_t* cursor, int32_t* rev_row, float* rev_w, void* stream);
int ofx_seg_primary_w(const int32_t* seg_ptr, const int32_t* col, const float* w, int64_t nseg, int32_t* nbr,
                      void* stream);
int ofx_seg_multi_flag_w(const int32_t* seg_ptr, const float* w, int64_t nseg, int32_t* flag, void* stream);
int ofx_seg_primary_ext_w(const int32_t* seg_ptr, const int32_t* col, const float* w, int64_t nseg, int64_t n_src,
                          const int32_t* rank, int32_t* nbr_ext, int32_t* multi_seg, void* stream);
/* Backward of the 27-tap grid convolution (torch autograd of nn.Conv3d(k=3, padding=1[, stride 2]) and of
 * nearest-upsample + conv: graph_unet_lr.py / modules.py:63-95).  WpT = ofx_pack_conv3d of weight.transpose(0,1);
 * dWp [pad32(27*cin), cout], row k = tap*cin + c. */
int ofx_gridconv_bwd_data(const float* dy, int64_t ldy, int cout, int64_t n_out, int64_t n_in, const int32_t* nbr_rev,
                          const int32_t* rev_ptr, const int32_t* rev_row, const float* rev_w,
                          const int32_t* nbr_ext_rev, const int32_t* multi_seg, int64_t n_multi, float* aux,
                          const float* WpT, int cin, float* dx, int64_t ldx, void* ws, size_t ws_bytes, void* stream);
int ofx_gridconv_bwd_weight(const float* x, int64_t ldx, int cin, int64_t n_in, int64_t n_out, const int32_t* nbr27,
                            const int32_t* nbr27_ext, const float* zero_row, const float* dy, int64_t ldy, int cout,
                            float* dWp, void* ws, size_t ws_bytes, void* stream);
/* Optimiser step of the training loop (octfusion_model_union.py:142, 478-487): torch.optim.AdamW's update
 * (step >= 1 is the 1-based step count) and the EMA of the weights (ldm_diffusion_util.py:38-54). */
int ofx_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                   float beta1, float beta2, float eps, float weight_decay, int step, void* stream);
int ofx_ema_update(float* ema, const float* param, int64_t n, float beta, void* stream);
/* NeuralMPU SDF evaluation -- replaces NeuralMPU.__call__ / get_linear_pred / octree_linear_pts
 * (models/networks/dualoctree_networks/mpu.py:55-153), spmm / modulated_spmm (utils/spmm.py:12-61) and, with the
 * _grid entry, the sampling loop of calc_sdf (utils/util_dualoctree.py:99-118).
 * pts [n,4] fp32 = (x, y, z in [-1,1], batch id).  For every depth d in [depth_start, depth_end] the 8 cell
 * centres around the point that exist in the tree (and are leaves when d < depth_end) contribute
 * w = prod(1-|f|) * d^2/50 and w * (code[row] . [f*2/2^d, 1]); sdf = sum / (sum w + 1e-8);
 * mask[i] = 1 iff a centre of depth_end exists.  code rows: nodes of depth_start..depth_end concatenated
 * (row = index in depth d + sum_{l<d} nnum[l]), 16-B aligned.  mask may be NULL.
 * _grid: point q = head + i of the size^3 lattice (x slowest), coordinate = fl(fl(i*step) + bbmin). */
int ofx_mpu_eval(const ofx_tree_t* tree, int depth_start, int depth_end, const float* pts, int64_t n_pts,
                 const float* code, float* sdf, uint8_t* mask, void* stream);
int ofx_mpu_eval_grid(const ofx_tree_t* tree, int depth_start, int depth_end, const float* code, int size,
                      float step, float bbmin, int batch_index, int64_t head, int64_t count, float* sdf,
                      uint8_t* mask, void* stream);
/* NeuralMPU with gradients (training): replaces compute_mpu_gradients (loss.py:100-108), i.e. autograd of
 * get_linear_pred w.r.t. the query position with create_graph=True.  grad [n,3] = d sdf / d (x, y, z) with the
 * reference's conventions (floor detached, d|f|/df = +1 at f = 0: mpu.py:18-32).  _backward is the adjoint of
 * (sdf, grad) w.r.t. the code table: dcode [rows,4] += J^T (dsdf, dgrad) (fp32 atomics; the caller zeroes
 * dcode; either upstream pointer may be NULL = zeros). */
int ofx_mpu_eval_grad(const ofx_tree_t* tree, int depth_start, int depth_end, const float* pts, int64_t n_pts,
                      const float* code, float* sdf, float* grad, uint8_t* mask, void* stream);
int ofx_mpu_backward(const ofx_tree_t* tree, int depth_start, int depth_end, const float* pts, int64_t n_pts,
                     const float* code, const float* dsdf, const float* dgrad, float* dcode, void* stream);
/* VAE training losses, forward value + gradient w.r.t. the network output in one pass (loss.py:164-178).
 * sums are fp64 accumulators the caller zeroes.
 * _octree_ce (compute_octree_loss, loss.py:110-122): label = child[i] >= 0; sums[0] += sum of the 2-class cross
 *   entropies, sums[1] += number of rows whose argmax equals the label; dlogits (optional) =
 *   (softmax - onehot) * dscale (dscale = weight / n for the mean).
 * _sdf_reg_loss (sdf_reg_loss, loss.py:23-29): sums[0] += sum (grad - grad_gt)^2 over 3n, sums[1] +=
 *   sum (sdf - sdf_gt)^2; dsdf = 2 w_sdf (sdf - sdf_gt) / n, dgrad = 2 w_grad (grad - grad_gt) / (3n) (optional).
 * _kl_sample (DiagonalGaussianDistribution, distributions.py:24-46): params [n, 2E] = (mean | logvar), logvar
 *   clamped to [-30, 20]; z = mean + exp(logvar/2) * noise (noise NULL: the mean); kl_sum += sum of
 *   0.5 (mean^2 + var - 1 - logvar).  _bwd: dparams = d/dparams of (<dz, z> + kl_scale * sum kl). */
int ofx_octree_ce(const float* logits, int64_t ld, const int32_t* child, int64_t n, float dscale, double* sums,
                  float* dlogits, int64_t ldd, void* stream);
int ofx_sdf_reg_loss(const float* sdf, const float* grad, const float* sdf_gt, const float* grad_gt, int64_t n,
                     float w_sdf, float w_grad, double* sums, float* dsdf, float* dgrad, void* stream);
int ofx_kl_sample_fwd(const float* params, int64_t ld, const float* noise, int64_t n, int embed_dim, float* z,
                      double* kl_sum, void* stream);
int ofx_kl_sample_bwd(const float* params, int64_t ld, const float* noise, const float* dz, int64_t n,
                      int embed_dim, float kl_scale, float* dparams, int64_t ldp, void* stream);
/* Extended table for the branch-free kernel: flag[s] = segment s has > 1 neighbours;
 * with rank = exclusive scan of flag: nbr_ext[s] = neighbour id | N (none: zero row) |
 * N + 1 + rank[s] (several: pre-averaged row), multi_seg[rank[s]] = s. */
int ofx_graph_multi_flag(const int32_t* seg_ptr, int64_t n_nodes, int32_t* flag, void* stream);
int ofx_graph_primary_ext(const int32_t* seg_ptr, const int32_t* col, int64_t n_nodes,
                          const int32_t* rank, int32_t* nbr_ext, int32_t* multi_seg, void* stream);
/* type_frac[r, dir*nt + t] = fraction of (r,dir)'s neighbours with node_type t; row
 * pitch ld floats, columns >= 7*nt zero-filled up to ld.  This is the one-hot
 * half of GraphConv's col_data (modules.py:199-210), constant per doctree. */
int ofx_graph_type_frac(const int32_t* seg_ptr, const int32_t* col, const uint8_t* node_type,
                        int64_t n_nodes, int nt, float* type_frac, int64_t ld, void* stream);

/* ---------------------------------------------------------------- GEMM core
 * Packed weight layout Wp[(k/4) * N + n][4] (k padded to Kp, multiple of 32,
 * with zero rows).  ofx_pack_weights handles plain [K,N] (sk = N, sn = 1),
 * transposed nn.Linear [N,K] (sk = 1, sn = K) and -- when cin > 0 -- the
 * GraphConv row permutation: reference row dir*(cin+nt)+c (modules.py:174-176)
 * -> packed k = dir*cin + c for features, 7*cin + dir*nt + t for node types. */
/* Contraction precision (process-wide).  Modes 0 and 3 split activations and weights into 16-bit hi + lo halves and
 * run a*w = a_lo*w_hi + a_hi*w_lo + a_hi*w_hi on the 16-bit matrix pipe with fp32 accumulation, 16/3 x the fp32-MFMA
 * rate:
 *   3 = fp16x3 (DEFAULT since round 3): fp16 halves, 22 significand bits (operands beyond +-65504 saturate) --
 *       whole-step element-wise error at the level of the reference's own fp32 arithmetic;
 *   0 = bf16x3: bf16 halves, 16 significand bits, fp32's exponent range (~1e-5 per layer, 2e-3 element-wise through a
 *       whole step);
 *   1 = exact fp32 MFMA (v_mfma_f32_32x32x2_f32, bit-equal to an fma chain);
 *   2 = reduced precision: the planes GraphConv runs ONE fp16 MFMA per product (operands rounded to fp16, fp32
 *       accumulate; ~5e-4 per product), every other contraction in bf16 pairs.
 * Not thread-safe: set it before launching work, from the thread that launches.  Packed weights and operand planes are
 * mode-specific (re-pack after switching).
 * A packed-weight buffer holds the fp32 pack followed by the 16-bit hi|lo planes of the mode it was packed in:
 * ofx_packed_floats(Kp, N) floats in total. */
int ofx_set_precision(int mode);
int ofx_get_precision(void);
int64_t ofx_packed_floats(int64_t Kp, int64_t N);
int64_t ofx_packed_k(int64_t K);                       /* K rounded up to 32 */
int64_t ofx_graphconv_packed_k(int cin, int nt);       /* 7*cin + pad32(7*nt)  */
int ofx_pack_weights(const float* W, int64_t sk, int64_t sn, int64_t K, int64_t N,
                     int cin, int nt, float* Wp, int64_t Kp, void* stream);

/* out[orow(m), n] = sum_k A[arow(m), k] * W[k, n] + bias[n] + res[m, n]
 * a_rows / out_rows: optional int32 row maps (NULL = identity); a negative
 * out_rows entry skips the row.  Replaces torch mm / nn.Linear in
 * Downsample/Upsample (modules.py:391-395, 440-443), Conv1x1 (:332-339), the
 * time-embedding MLPs (graph_unet_hr.py:107-111). */
int ofx_gemm_f32(const float* A, int64_t lda, const int32_t* a_rows, int64_t M, int64_t K,
                 const float* Wp, int64_t Kp, int64_t N, const float* bias,
                 const float* res, int64_t ldr, float* out, int64_t ldc,
                 const int32_t* out_rows, void* ws, size_t ws_bytes, void* stream);
/* The same GEMM writing `out` as hi / lo pair planes (out_mode 2 = bf16 pairs, 3 = fp16 pairs; 0 = fp32 rows, i.e.
 * ofx_gemm_f32): needs N % 4 == 0, a 128-B aligned `out` with ldc % 32 == 0 and 16-B aligned res / bias / ws. */
int ofx_gemm_f32_planes(const float* A, int64_t lda, const int32_t* a_rows, int64_t M, int64_t K,
                        const float* Wp, int64_t Kp, int64_t N, const float* bias,
                        const float* res, int64_t ldr, float* out, int64_t ldc,
                        const int32_t* out_rows, void* ws, size_t ws_bytes, int out_mode, void* stream);

/* ---------------------------------------------------------------- GraphConv
 * Fused dual-octree graph convolution (modules.py:194-220 + scatter.py:42-66):
 *   out[r, :] = [ mean_{e in seg(r,dir)} x[col[e], :] for dir in 0..6 |
 *                 type_frac[r, :] ] @ W  + bias + emb[batch_id[r], :] + res[r, :]
 * The gather/segment-mean is done on the fly into LDS (col_data is never
 * written to HBM); the contraction runs on fp32 MFMA.  emb/batch_id fuse the
 * reference's per-batch-element time-embedding add (modules.py:754-758), res
 * the residual / skip add (:763).  type_frac may be NULL (nt <= 1).
 * nbr = ofx_graph_primary table (generic path: any cin).  nbr_ext / multi_seg / aux
 * (scratch of (n_multi+1)*cin floats) enable the branch-free fast path when
 * cin % 32 == 0: a pre-pass averages the few multi-neighbour segments into aux, the
 * main kernel then gathers exactly one row per (row,dir).  ws: split-K workspace.
 * stats (optional, needs batch_id): the epilogue also accumulates the GroupNorm statistics
 * of the OUTPUT, stats[(b*stats_ld + n)*2 + {0,1}] += (v, v^2) in fp64 (caller zeroes it),
 * in the layout ofx_gn_finalize reads -- the consuming norm then skips ofx_gn_stats. */
int ofx_graphconv_fwd(const float* x, int64_t ldx, int cin, int64_t n_nodes,
                      const int32_t* nbr, const int32_t* seg_ptr, const int32_t* col,
                      const int32_t* nbr_ext, const int32_t* multi_seg, int64_t n_multi, float* aux,
                      const float* type_frac, int64_t ldt, int nt_pad,
                      const float* Wp, int64_t Kp, int cout, const float* bias,
                      const float* emb, int64_t lde, const int32_t* batch_id,
                      const float* res, int64_t ldr, float* out, int64_t ldc,
                      double* stats /* optional */, int64_t stats_ld,
                      void* ws, size_t ws_bytes, void* stream);

/* The two GraphConvs of a diffusion U-Net that are gathers, not contractions (csrc/ofx_narrow.hip), same operator
 * (modules.py:194-220), raw nn.Parameter weights W [7 * (cin + nt), cout] row-major (no packing):
 *  ofx_graphconv_narrow_in: the INPUT convolution (graph_unet_hr.py:116), cin <= 8 channels -> cout in {64, 128},
 *      7 * (cin + nt) <= 96, nt <= 8.  col_data of a 64-row block in LDS (node-type fractions counted from node_type
 *      [n] uint8, the per-node type of ofx_graph_nodes, while walking the neighbours), a lane owns 1-2 output columns
 *      with its weights in registers, exact fp32 FMA; `stats` as in ofx_graphconv_fwd (ws >= ceil(n / 64) * cout * 8
 *      bytes of partials).
 *  ofx_graphconv_narrow_out: the OUTPUT convolution (graph_unet_hr.py:205-209), C channels -> cout <= 8, as
 *      project-then-aggregate (scatter_mean and the weight product commute): the caller first computes the dense
 *      P = y @ Wd with Wd = ofx_narrow_out_pack(W) ([C, pw], Wd[c, dir * cout + o] = W[dir * (C + nt) + c, o], pw >= 7 cout
 *      zero-padded: 32 or 64 so that a row of P is whole 128-B lines), then this call gathers cout floats per edge:
 *      out[r, o] = sum_dir mean_{e in seg(r, dir)} P[col[e], dir * cout + o] + type_term[r, o], where
 *      type_term [n, cout] = ofx_narrow_out_type_term(...) = bias[o] + sum_{dir, t} type_frac[r, dir * nt + t] *
 *      W[dir * (C + nt) + C + t, o] is constant per (graph depth, weights) -- computed once per doctree, NULL = zero. */
int ofx_graphconv_narrow_in(const float* x, int64_t ldx, int cin, int64_t n_nodes, const int32_t* seg_ptr,
                            const int32_t* col, const uint8_t* node_type, int nt, const float* W, int cout,
                            const float* bias, const int32_t* batch_id, float* out, int64_t ldc,
                            double* stats /* optional */, int64_t stats_ld, void* ws, size_t ws_bytes, void* stream);
/* ofx_graphconv_narrow_in_tab (round 6): ofx_graphconv_narrow_in through the branch-free gather table -- nbr_ext [n, 7]
 * / multi_seg [n_multi] of ofx_graph_primary_ext; `aux` = scratch of (n_nodes + n_multi + 1) * (cin <= 4 ? 32 : 64) bytes
 * (16-B aligned) that a pre-pass fills with one record per gatherable id (x + node type of a row; x mean + node-type
 * counts of a multi-neighbour segment): a segment then costs one table entry + one aligned record.  One persistent launch of two
 * 512-thread blocks per CU; a block walks row groups with the next group's gathers in flight under the current
 * group's MFMAs and stores.  Same results as ofx_graphconv_narrow_in to fp32 rounding (the segment means of
 * multi-neighbour segments are taken before, not after, the weight product's inputs are staged: same arithmetic). */
int ofx_graphconv_narrow_in_tab(const float* x, int64_t ldx, int cin, int64_t n_nodes, const int32_t* seg_ptr,
                                const int32_t* col, const int32_t* nbr_ext, const int32_t* multi_seg, int64_t n_multi,
                                void* aux, const uint8_t* node_type, int nt, const float* W, int cout, const float* bias,
                                const int32_t* batch_id, float* out, int64_t ldc, double* stats /* optional */,
                                int64_t stats_ld, void* ws, size_t ws_bytes, void* stream);
int ofx_narrow_out_pack(const float* W, int C, int nt, int cout, int pw, float* Wd, void* stream);
int ofx_narrow_out_type_term(const float* type_frac, int64_t ldt, int nt, int64_t n_nodes, const float* W, int C,
                             int cout, const float* bias, float* type_term, void* stream);
int ofx_graphconv_narrow_out(const float* P, int64_t ldp, int cout, int64_t n_nodes, const int32_t* seg_ptr,
                             const int32_t* col, const float* type_term /* optional */, float* out, int64_t ldc,
                             void* stream);

/* ---------------------------------------------------------------- GraphConv on operand planes
 * Second implementation of the same operator (modules.py:194-220) for the layers that carry the step's
 * time: operands arrive PRE-SPLIT and are staged global -> LDS by DMA (csrc/ofx_gemm2.hip).
 *  "planes", mode 3 (fp16x3, the default precision) / mode 2 (bf16x3): every 32-channel chunk of a row is one
 *      128-B line [hi x 32 | lo x 32] of fp16 / bf16 halves, value = hi + lo: the bytes of the fp32 row, so a
 *      planes tensor aliases an fp32-shaped [rows, C] buffer (row pitch = fp32 pitch, C % 32 == 0);
 *  mode 1 (single-pass fp16, reduced precision, ofx_set_precision(2)): fp16 row-major, C % 64 == 0.
 * ofx_planes_split: fp32 -> planes (columns C..Cpad-1 zero-filled; in place allowed for modes 2 and 3).
 * ofx_planes_merge: planes -> fp32 (tests).
 * ofx_gn_apply_planes: ofx_gn_apply (modules.py:311-314 + fused SiLU/GELU) writing planes; `out` may be x
 *      itself in modes 2 and 3.  With aux != NULL (then out must not alias x) the same launch also writes the consuming
 *      GraphConv's aux rows (zero row + multi-neighbour means, from the CSR seg_ptr / col / multi_seg of that
 *      graph depth), and ofx_graphconv_fwd_planes is called with aux_ready = 1: one launch less per convolution.
 *      aux_plan (optional, with aux): which block writes which aux row -- int32 [mb + 1] ptr (mb = ceil(n / 64) blocks
 *      of 64 consecutive rows) | ptr[mb] aux row ids 1..n_multi grouped by the block that holds ALL their source rows |
 *      aux_left | aux_left leftover ids (sources in several blocks; always contains the zero row 0).  A block then
 *      writes its aux rows right after its own rows, from L1 / L2 instead of a second trip to HBM.
 * ofx_pack_weights_planes: GraphConv weights [7*(cin+nt'), cout] (element (k, n) at W[k*sk + n*sn]) ->
 *      [k tile][cout][128-B line], k order: 7*cin gathered channels direction-major, then the 7*nt node-type
 *      rows zero-padded to a whole tile; ofx_planes_packed_bytes() bytes.
 * ofx_graphconv_fwd_planes: as ofx_graphconv_fwd, with xp / aux / tfp planes (row pitches in BYTES, 128-B
 *      aligned bases and pitches), aux = scratch of aux_bytes >= (n_multi + 1) * ldx_bytes, tfp = planes of the
 *      type_frac slab padded to a whole chunk (NULL when nt <= 1), W2 = packed planes weights.
 *      Two launch shapes (csrc/ofx_gemm3.hip, csrc/ofx_gemm2.hip):
 *      - persistent stream-K blocks (default when `sync` is given and the layer has >= 64 k-step units): as many
 *        blocks as the device holds at once, each owning an equal contiguous share of the (tile, k-step) sequence;
 *        tiles cut by a share boundary are combined in-launch through `ws` (raw fp32 accumulator pieces behind the
 *        statistics partials: <= (CUs * 2) * 64 KB) and `sync`: >= (2 * CUs + 1) uint32, ZERO on the first call,
 *        left zero by every launch.  The LAST word of the buffer (sync_bytes / 4 - 1, whatever the block count) is a
 *        sticky error flag: a bounded wait gave up (several seconds: a hung or pre-empted device).  While it is set
 *        every block of every later launch on this buffer returns at entry WITHOUT computing (a late contributor may
 *        still store into flag words the next launch believes to be zero), so the caller must check it once per batch
 *        of launches, clear the buffer and fall back to ofx_set_gconv_persistent(0).  One buffer (and one `ws`) per
 *        stream: launches that may run concurrently must not share them.  nbr_ext needs
 *        16 B of readable slack behind its last entry and a 16-B aligned base.  Deterministic: the pieces of a tile
 *        are added in ascending k order.
 *      - one tile per block (sync == NULL, tiny layers, ofx_set_gconv_persistent(0)): 256 / 128 x 128 tiles, no
 *        split-K, meant for layers with >= ~128 tiles. */
int ofx_planes_split(const float* x, int64_t ldx, int64_t n, int C, int Cpad, int mode, void* out,
                     int64_t ldo_bytes, void* stream);
int ofx_planes_merge(const void* planes, int64_t ldp_bytes, int64_t n, int C, int mode, float* out, int64_t ldo,
                     void* stream);
int ofx_gn_apply_planes(const float* x, int64_t ldx, int64_t n, int C, const int32_t* batch_id, const float* mean,
                        const float* rstd, const double* sums, const float* count, int groups, float eps,
                        float count_eps, const float* w, const float* bias, int act, int mode, void* out,
                        int64_t ldo_bytes, const int32_t* seg_ptr, const int32_t* col, const int32_t* multi_seg,
                        int64_t n_multi, void* aux /* optional */, const int32_t* aux_plan /* optional */,
                        int64_t aux_left, void* stream);
/* ofx_gn_apply_planes_oct (round 6): ofx_gn_apply_planes with aux rows, on the sibling-octet mapping -- the same
 * outputs (planes of act(GroupNorm(x)) + the consuming GraphConv's aux rows: modules.py:291-326 feeding :194-220).
 * A thread holds the eight rows of an "octet" (rows 8 o - shift .. 8 o - shift + 7; `shift` pads the coarse-leaf
 * prefix of the graph depth to a multiple of eight, so the octets of the depth-d part are sibling groups) for its
 * four channels in registers; an aux row whose sources all lie in one octet is a masked mean of registers.
 *   oct_ptr   int32 [n_oct + 1], n_oct = ceil((n + shift) / 8): entry range of every octet;
 *   oct_ent   int32 [n_own][2] (8-B aligned): (aux row id 1..n_multi, 8-bit mask of the octet's rows it averages);
 *   left_head int32 [n_left][4] (16-B aligned): every other aux row as (aux row id, first slot in left_src, number of
 *             sources, batch element); always contains the zero row (0, 0, 0, 0); n_own + n_left == n_multi + 1;
 *   left_src  int32: the source rows of the leftover aux rows, flattened (the CSR segment of each, in order).
 * Leftover rows are re-normalised from x by extra blocks interleaved with the main blocks (ofx_set_gn_left_place(1):
 * all behind them, A/B).  mean / rstd from ofx_gn_finalize, or both NULL with (sums, count, groups, eps, count_eps):
 * finalised per block on the fly, as in ofx_gn_apply.  out must not alias x.  Host-side builder:
 * octfusion_amd/dual_octree.py DualOctree.oct_plan. */
int ofx_gn_apply_planes_oct(const float* x, int64_t ldx, int64_t n, int C, const int32_t* batch_id, const float* mean,
                            const float* rstd, const double* sums, const float* count, int groups, float eps,
                            float count_eps, const float* w, const float* bias, int act, int mode, void* out,
                            int64_t ldo_bytes, int64_t n_multi, void* aux, const int32_t* oct_ptr,
                            const int32_t* oct_ent, int64_t n_own, int shift, const int32_t* left_head,
                            const int32_t* left_src, int64_t n_left, void* stream);
int ofx_set_gn_left_place(int at_end);
/* rows per main block of ofx_gn_apply_planes: the granularity `aux_plan` is built for (the host-side plan builder,
 * octfusion_amd/dual_octree.py aux_plan, asks instead of assuming). */
int ofx_gn_apply_rows(void);
int64_t ofx_planes_packed_ktiles(int cin, int nt, int mode);
int64_t ofx_planes_packed_bytes(int cin, int nt, int cout, int mode);
int ofx_pack_weights_planes(const float* W, int64_t sk, int64_t sn, int cin, int nt, int cout, int mode, void* out,
                            void* stream);
int ofx_graphconv_fwd_planes(const void* xp, int64_t ldx_bytes, int cin, int64_t n_nodes, const int32_t* seg_ptr,
                             const int32_t* col, const int32_t* nbr_ext, const int32_t* multi_seg, int64_t n_multi,
                             void* aux, size_t aux_bytes, const void* tfp, int64_t ldt_bytes, int nt, const void* W2,
                             int cout, const float* bias, const float* emb, int64_t lde, const int32_t* batch_id,
                             const float* res, int64_t ldr, float* out, int64_t ldc, double* stats /* optional */,
                             int64_t stats_ld, void* ws, size_t ws_bytes, void* sync /* optional */, size_t sync_bytes,
                             int mode, int aux_ready, void* stream);
/* 1 (default): persistent launch (whole-tile rounds + a stream-K region) where the shape qualifies; 0: always one tile
 * per block; 2: persistent with pure stream-K (no whole-tile rounds); 3: as 1 with share boundaries snapped towards the
 * tile boundary instead of to the nearest legal cut -- A/B knobs, same values */
int ofx_set_gconv_persistent(int on);
/* ofx_gemm_planes (round 6): out[m, :] = A[row_tab[m, 0], :] @ W (+ bias) on the data path of the planes GraphConv
 * (persistent stream-K blocks, LDS-DMA staging, fp16x3 / bf16x3 MFMA; csrc/ofx_gemm3.hip with one "direction") -- the
 * reference's Upsample GEMM x[n, C] @ W.flatten(1) -> [n, 8 C] (modules.py:430-446, call site :458-467) for the
 * non-leaf rows of a graph depth.  ap: pair planes of A (mode 2 / 3; 128-B aligned rows); row_tab: int32 [M, 7], 16-B
 * aligned, + 16 B of readable slack: column 0 = the source row in [0, n_a) of output row m (the other six columns are
 * converted but never dereferenced: any row in [0, n_a)); W2 from ofx_pack_gemm_planes ((K / 32) * N * 128 + 128 bytes,
 * ofx_gemm_planes_packed_bytes); out fp32 rows, or with out_mode 2 / 3 pair planes (what the next GraphConv gathers).
 * ws / sync: the workspace and flag words of ofx_graphconv_fwd_planes (same protocol, same sticky error word).
 * Returns OFX_OK, a negative status, or 1 when the shape does not qualify (fewer than 8 k-steps of 32 channels, N < 128,
 * too few tiles, workspace too small): nothing was launched and the caller uses ofx_gemm_f32 / ofx_gemm_f32_planes. */
int64_t ofx_gemm_planes_packed_bytes(int K, int N, int mode);
int ofx_pack_gemm_planes(const float* W, int64_t sk, int64_t sn, int K, int N, int mode, void* out, void* stream);
int ofx_gemm_planes(const void* ap, int64_t lda_bytes, int64_t n_a, int64_t M, int K, const int32_t* row_tab,
                    const void* W2, int N, const float* bias, float* out, int64_t ldc, int out_mode, void* ws,
                    size_t ws_bytes, void* sync, size_t sync_bytes, int mode, void* stream);
/* A/B knob of the dense GEMM (csrc/ofx_gemm.hip): max_n > 0 -> GEMMs with N <= max_n and M >= 65 536 rows use 64-column
 * tiles (three blocks per CU instead of two); 0 = off. */
int ofx_set_gemm_bn64(int max_n);
/* A/B knob of the persistent launch's tile order: 1 = every XCD walks one contiguous range of tiles over the whole
 * launch (csrc/ofx_gemm3.hip, Gemm3Args::xcd_contig), 0 = the XCDs interleave inside every whole-tile round. */
int ofx_set_gconv_xcd_contig(int on);
/* Plan the persistent launches for at most `cus` compute units (0 = all of the device's; values above the device's
 * count are clamped to it), so that the launches of two lanes on two HIP streams can be resident side by side
 * (octfusion_amd/sampler.py: lanes; tools/two_half_probe.py).  Process-wide; affects launches issued or captured
 * afterwards.  OFX_EINVAL for 1..7. */
int ofx_set_gconv_cus(int cus);
/* The schedule ofx_graphconv_fwd_planes' persistent launch would use for n_rows x cout outputs and nkt k-steps per tile
 * (host only, no device work; cus > 0: plan for that many compute units): out[0..4] = blocks G, q, rem, region units U,
 * whole-tile rounds; out[5 .. 5 + G] = share boundaries of the stream-K region (k-step units, bound(0) = 0,
 * bound(G) = U; every piece between a boundary and a tile edge has >= 8 k-steps).  Returns G, 0 = not eligible. */
int ofx_gconv3_plan(int64_t n_rows, int cout, int nkt, int wm, int ni, int cus, int32_t* out, int64_t out_len);
/* scheduling variant of the one-tile-per-block planes kernel: 5 = LDS reads and DMA requests spliced between the
 * MFMAs -- the only one a product build contains (anything else: OFX_EINVAL).  Builds with -DOFX_ABLATION
 * (python -m octfusion_amd.build --ablation) also hold 1 = DMA requests interleaved by sched_group_barrier,
 * 0 = requests before the MFMA group, and 2, 3, 4, 6 = timing ablations with WRONG results (no MFMA / no DMA /
 * no fragment reads / no barrier). */
int ofx_set_gconv2_variant(int v);
/* block geometry of the planes kernel: 0 = automatic, 2 = 128 x 128 tiles (4 waves, two blocks per CU),
 * 4 = 256 x 128 tiles (8 waves, one block per CU) -- A/B knob */
int ofx_set_gconv2_tile(int wm);
/* start offset between the two co-resident blocks of a CU (128-row geometry), in shader clocks per k tile of the
 * layer; 0 = start together -- A/B knob */
int ofx_set_gconv2_stagger(int clocks_per_ktile);
int ofx_set_gconv2_prefetch(int on);   /* 1 (default): block b pulls the table slice of block b + resident blocks into L2 */
/* profiling aid (-DOFX_ABLATION builds only; a product build accepts NULL only): when buf != NULL every block of the following ofx_graphconv_fwd_planes launches writes 8 uint64
 * to buf[block*8..]: shader-clock stamps at start / table built / first tile landed / k-loop done / stores drained,
 * then HW_ID.  buf must hold 8 * (tiles of the largest launch) uint64.  NULL switches it off. */
int ofx_set_gconv2_debug(void* buf);

/* ---------------------------------------------------------------- dense grids
 * The nested dense U-Net (graph_unet_lr.py) in node-row layout: a full octree layer of
 * depth d is rows b*8^d + morton(x,y,z), so octree2voxel / the gather back
 * (graph_unet_lr.py:176-181) are identities and a 3x3x3 Conv3d is the same fused
 * gather-GEMM with 27 taps.
 * ofx_grid_conv_table: nbr27[row*27 + tap] (tap = (kx*3+ky)*3+kz); out-of-grid taps get
 *   `pad`: -1 for the generic kernel, n_in (the zero row) for the branch-free kernel.
 *   mode 0: nn.Conv3d(k3,p1) at depth_out; mode 1: ConvDownsample (k3,s2,p1,
 *   modules.py:81-95) depth_out+1 -> depth_out; mode 2: ConvUpsample (nearest x2 then
 *   k3,p1, modules.py:63-78) depth_out-1 -> depth_out.
 * ofx_pack_conv3d: nn.Conv3d weight [cout,cin,3,3,3] -> packed k = tap*cin + c.
 * ofx_gridconv_fwd: out[r,:] = sum_tap x[nbr27[r,tap],:] @ W_tap + bias + emb[bid[r]] + res[r]
 *   (emb fuses ResnetBlock's time_mlp add, modules.py:507-511; res the skip, :513).
 * ofx_attention: QKVAttention (modules.py:538-547) on rows; see csrc/ofx_dense.hip. */
int ofx_grid_conv_table(int mode, int depth_out, int batch_size, int32_t pad, int32_t* nbr27,
                        void* stream);
int64_t ofx_conv3d_packed_k(int cin);
int ofx_pack_conv3d(const float* W, int cin, int cout, float* Wp, void* stream);
int ofx_gridconv_fwd(const float* x, int64_t ldx, int cin, int64_t n_in, int64_t n_out,
                     const int32_t* nbr27 /* pad -1, may be NULL */,
                     const int32_t* nbr27_ext /* pad n_in, may be NULL */,
                     const float* zero_row /* >= cin zeros, 16-B aligned */,
                     const float* Wp, int cout, const float* bias, const float* emb, int64_t lde,
                     const int32_t* batch_id, const float* res, int64_t ldr, float* out,
                     int64_t ldc, void* ws, size_t ws_bytes, void* stream);
/* The same branch-free gather-GEMM with a caller-made table of ntap sources per output row (entries in [0, n_src];
 * n_src = the zero row): out[orow(r), :] = [ x[tab[r, 0], :] | ... | x[tab[r, ntap - 1], :] ] @ W + bias + res[r].
 * Downsample (modules.py:391-395) is this with tab[r, j] = 8 r + j, for an x whose row pitch is not its width (a column
 * slice of the skip-concatenation buffer) -- the reference's x.view(-1, 8 C) would have to copy.  cin % 32 == 0; W packed
 * as for ofx_gemm_f32 with K = ntap * cin; out_rows / out_mode as in ofx_gemm_f32_planes. */
int ofx_gather_gemm_f32(const float* x, int64_t ldx, int cin, int ntap, int64_t n_src, int64_t n_out, const int32_t* tab,
                        const float* zero_row, const float* Wp, int64_t Kp, int cout, const float* bias,
                        const float* res, int64_t ldr, float* out, int64_t ldc, const int32_t* out_rows, void* ws,
                        size_t ws_bytes, int out_mode, void* stream);
/* A/B knob: 0 keeps sequences of >= 256 tokens on the one-wave-per-32-queries attention kernel (default 1: the keys of a
 * 32-query tile are split over the four waves of a block). */
int ofx_set_attention_split(int on);
int ofx_attention(const float* qkv, int64_t ldq, int batch_size, int T, int heads, int ch,
                  float* out, int64_t ldo, void* stream);

/* Backward of ofx_attention (autograd of QKVAttention, modules.py:538-547): dqkv [rows, 3*C] in the layout of
 * qkv, from dout [rows, C].  rowstat: scratch of batch*heads*T*3 floats.  T <= 512. */
int ofx_attention_bwd(const float* qkv, int64_t ldq, const float* dout, int64_t ldo, int batch_size, int T, int heads,
                      int ch, float* rowstat, float* dqkv, int64_t ldd, void* stream);

/* Stand-alone segment-mean gather: col_data[r, dir, :] (the reference's
 * `scatter_mean(x[col], row*7+dir)`, modules.py:208-210).  HBM-bound; used for
 * the gather roofline measurement and as a building block. */
int ofx_gather_mean(const float* x, int64_t ldx, int cin, int64_t n_nodes,
                    const int32_t* seg_ptr, const int32_t* col, float* col_data,
                    void* stream);

/* ---------------------------------------------------------------- GroupNorm
 * DualOctreeGroupNorm (modules.py:291-326): statistics per (batch element,
 * group) over all nodes of that element.
 *  stats:    sums[b, c, 0..1] += (sum x, sum x^2) in fp64 (zeroed inside).
 *  finalize: mean/rstd [B, C] fp32 with inv_count = 1/(count*cpg + count_eps) and
 *            centred variance; count_eps = eps reproduces the reference (:302),
 *            count_eps = 0 is torch.nn.GroupNorm (GroupNorm32, modules.py:26-28).
 *  apply:    out = act((x - mean[b]) * rstd[b] * w + bias);
 *            mean / rstd from ofx_gn_finalize -- or both NULL, then (sums, count, groups, eps, count_eps) and the
 *            launch finalises on the fly (same arithmetic, same bits; one launch less per norm). */
int ofx_gn_stats(const float* x, int64_t ldx, int64_t n, int C, const int32_t* batch_id,
                 int batch_size, double* sums, void* stream);
/* ofx_gn_stats without the zero-fill: accumulates into caller-zeroed `sums` (no memset node per statistics launch). */
int ofx_gn_stats_acc(const float* x, int64_t ldx, int64_t n, int C, const int32_t* batch_id, int batch_size,
                     double* sums, void* stream);
int ofx_gn_finalize(const double* sums, const float* count, int batch_size, int C, int groups,
                    float eps, float count_eps, float* mean, float* rstd, void* stream);
int ofx_gn_apply(const float* x, int64_t ldx, int64_t n, int C, const int32_t* batch_id,
                 const float* mean /* or NULL */, const float* rstd /* or NULL */, const double* sums /* if mean NULL */,
                 const float* count, int groups, float eps, float count_eps, const float* w, const float* bias,
                 int act, float* out, int64_t ldo, void* stream);

/* One-launch GroupNorm (+ activation) for layouts whose batch elements own `rows_per_batch` CONTIGUOUS rows (the
 * dense layers of the nested lr net; modules.py:26-28 GroupNorm32 with count_eps = 0): same arithmetic as
 * ofx_gn_stats + ofx_gn_finalize + ofx_gn_apply. */
int ofx_gn_fused_rows(const float* x, int64_t ldx, int rows_per_batch, int batch_size, int C, int groups, float eps,
                      float count_eps, const float* w, const float* bias, int act, float* out, int64_t ldo,
                      void* stream);

/* A/B knob: 1 (default) = 16-channel blocks for ofx_gn_fused_rows where the shape allows, 0 = one block per group */
int ofx_set_gn_rows16(int on);

/* ---------------------------------------------------------------- glue ops */
/* dst[dmap(i), 0:C] = src[smap(i), 0:C] for i < n (maps optional; negative skips). */
int ofx_rows_copy(const float* src, int64_t lds, const int32_t* smap, float* dst, int64_t ldd,
                  const int32_t* dmap, int64_t n, int C, void* stream);
/* ofx_rows_copy writing the destination rows as hi / lo pair planes (mode 2 / 3: the operand format of
 * ofx_graphconv_fwd_planes): C % 32 == 0, dst 128-B aligned, ldd % 32 == 0.  With ofx_gemm_f32_planes this lets the
 * pool / unpool of the U-Net (modules.py:409-423, 458-467) hand the following GraphConv its operand planes directly. */
int ofx_rows_copy_planes(const float* src, int64_t lds, const int32_t* smap, float* dst, int64_t ldd,
                         const int32_t* dmap, int64_t n, int C, int mode, void* stream);
/* y = act(x) elementwise over n floats. */
int ofx_act(const float* x, float* y, int64_t n, int act, void* stream);
/* sinusoidal embedding of t[B] (ldm_diffusion_util.py:171-191): out [B, dim]. */
int ofx_timestep_embedding(const float* t, int batch_size, int dim, float max_period,
                           float* out, void* stream);
/* LearnedSinusoidalPosEmb of the dense net (modules.py:550-563): out [B, 2 * half + 1] = [t, sin(2 pi t w), cos(2 pi t w)]. */
int ofx_learned_sinusoid(const float* t, const float* w, int batch_size, int half, float* out, void* stream);
/* Linear layer on M <= 16 rows (time / label embedding MLPs, per-block embedding projections; modules.py:754,
 * graph_unet_hr.py:253-257, graph_unet_lr.py:186-193): out = act_out(act_in(a) @ W^T + bias + res) with W [N, K] in
 * nn.Linear's own layout (no packing), exact fp32 FMA, one launch.  act_* = OFX_ACT_*; bias / res optional. */
int ofx_linear_small(const float* a, int64_t lda, int M, int K, const float* W, int64_t ldw, int N, const float* bias,
                     const float* res, int64_t ldr, int act_in, int act_out, float* out, int64_t ldo, void* stream);
/* DDIM eps-branch update (octfusion_model_union.py:345-350), coef on device:
 * coef = {alpha, sigma, alpha_next, sigma_next}; x updated in place; x0_out (optional)
 * receives x_start = (x - eps*sigma)/max(alpha,1e-8), which the reference hands to the next
 * step as x_self_cond. */
int ofx_ddim_eps_update(float* x, const float* eps, const float* coef, float* x0_out, int64_t n,
                        void* stream);
/* DDIM x0-branch update (:326-344): x = mean + sqrt(var) * noise with
 * coef = {alpha, c, alpha_next, sqrt(sigma_next^2 * c) or 0 when truncated}. */
int ofx_ddim_x0_update(float* x, const float* x0, const float* noise, const float* coef,
                       int64_t n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* OFX_H_ */

// libofx: dual-octree neighbour graph build (integer work; pointer chasing through
// the child arrays, all reads of keys / children coalesced per node block).
//
// Replaces models/networks/dualoctree_networks/dual_octree.py:19-409.  The
// reference refines an edge list level by level (dense_graph -> sparse_graph x
// (depth - full_depth) -> add_self_loops -> remap_node_idx -> argsort).  Here each
// graph node finds its face neighbours directly: walk the neighbour cell down
// from the dense full layer through `children`; a leaf on the way is a single
// coarser neighbour, a subdivided same-size cell (only possible when the node
// itself is a coarser leaf) is expanded into its face-touching descendants.
// Emitting per (row, dir) yields the reference's sorted-by-row*7+dir order with no
// sort, and CSR segments come for free.  Output is bit-identical to the reference
// after canonical (row, dir, col) ordering (tests/test_gpu_parity.py: test_octree_and_graph_*,
// test_graph_vs_oracle_random, test_full_size_shell6_b8_properties).
#include "ofx_common.h"

struct TreeDev {
  int depth, full_depth, batch_size;
  const int32_t* child;
  const int64_t* key;
  const int32_t* leafrank;
  int64_t ncum[OFX_MAX_DEPTH + 2];
  int64_t nnum[OFX_MAX_DEPTH + 1];
  int64_t leaf_base[OFX_MAX_DEPTH + 2];  // leaf_base[t] = sum_{s=fd}^{t-1} lnum[s]
};

static int make_tree(const ofx_tree_t* t, TreeDev& T) {
  if (!t || t->depth < 0 || t->depth > OFX_MAX_DEPTH || t->full_depth < 0 || t->full_depth > t->depth ||
      t->batch_size < 1 || !t->child_all || !t->key_all || !t->nnum_host || !t->nnum_nempty_host)
    return OFX_EINVAL;
  T.depth = t->depth;
  T.full_depth = t->full_depth;
  T.batch_size = t->batch_size;
  T.child = t->child_all;
  T.key = t->key_all;
  T.leafrank = t->leafrank_all;
  int64_t c = 0, lb = 0;
  for (int d = 0; d <= OFX_MAX_DEPTH; ++d) {
    T.ncum[d] = c;
    T.leaf_base[d] = lb;
    if (d <= t->depth) {
      T.nnum[d] = t->nnum_host[d];
      c += t->nnum_host[d];
      if (d >= t->full_depth) lb += t->nnum_host[d] - t->nnum_nempty_host[d];
    } else {
      T.nnum[d] = 0;
    }
  }
  T.ncum[OFX_MAX_DEPTH + 1] = c;
  T.leaf_base[OFX_MAX_DEPTH + 1] = lb;
  return OFX_OK;
}

static inline int64_t graph_nodes(const TreeDev& T, int d) { return T.leaf_base[d] + T.nnum[d]; }

// graph row r of depth-d graph -> (tree depth t, index j inside depth t)
__device__ __forceinline__ void row2node(const TreeDev& T, int d, int64_t r, int& t, int64_t& g) {
  // blocks: leaves of fd..d-1 then all of d.  Few levels: linear search.
  t = d;
  for (int s = T.full_depth; s < d; ++s) {
    if (r < T.leaf_base[s + 1]) { t = s; break; }
  }
  if (t == d) {
    g = T.ncum[d] + (r - T.leaf_base[d]);
  } else {
    // the (r - leaf_base[t])-th leaf of depth t: binary search on leafrank (non-decreasing)
    const int64_t want = r - T.leaf_base[t];
    const int32_t* lr = T.leafrank + T.ncum[t];
    const int32_t* ch = T.child + T.ncum[t];
    int64_t lo = 0, hi = T.nnum[t] - 1;   // find largest j with leafrank[j] <= want, and child[j] < 0
    while (lo < hi) {
      int64_t mid = (lo + hi + 1) >> 1;
      if (lr[mid] <= want) lo = mid; else hi = mid - 1;
    }
    // lo is the last j with leafrank <= want; that j is the leaf itself (rank increments after a leaf)
    (void)ch;
    g = T.ncum[t] + lo;
  }
}

// compact id of tree node (t, j) in the depth-d graph
__device__ __forceinline__ int32_t node2row(const TreeDev& T, int d, int t, int64_t j) {
  if (t == d) return (int32_t)(T.leaf_base[d] + j);
  return (int32_t)(T.leaf_base[t] + T.leafrank[T.ncum[t] + j]);
}

// face tables (dual_octree.py:85-97 semantics): dir 0:+z 1:-z 2:+y 3:-y 4:+x 5:-x
__device__ __forceinline__ void dir_delta(int dir, int& dx, int& dy, int& dz) {
  dx = dir == 4 ? 1 : (dir == 5 ? -1 : 0);
  dy = dir == 2 ? 1 : (dir == 3 ? -1 : 0);
  dz = dir == 0 ? 1 : (dir == 1 ? -1 : 0);
}
// the k-th (0..3) child octant lying on the face of a cell that looks in direction `dir`
__device__ __forceinline__ int face_octant(int dir, int k) {
  // axis bit: z=1, y=2, x=4; positive dirs want the bit set.
  const int axis = dir >> 1;                  // 0:z 1:y 2:x
  const int bit = 1 << axis;
  const int set = (dir & 1) ? 0 : bit;
  // spread k's two bits over the two other axes
  int o;
  if (axis == 0) o = (k << 1);                               // bits y,x <- k
  else if (axis == 1) o = (k & 1) | ((k & 2) << 1);          // bits z,x
  else o = k;                                                // bits z,y
  return o | set;
}

// Enumerate neighbours of tree node (t, g) through face `dir` in the depth-d graph.
// EMIT(row_id) is called for every neighbour.
template <typename Emit>
__device__ __forceinline__ void for_each_neighbour(const TreeDev& T, int d, int t, int64_t g, int dir, Emit emit) {
  int x, y, z, b;
  ofx_key2xyz(T.key[g], x, y, z, b);
  int dx, dy, dz;
  dir_delta(dir, dx, dy, dz);
  const int nx = x + dx, ny = y + dy, nz = z + dz;
  const int bnd = 1 << t;
  if (nx < 0 || ny < 0 || nz < 0 || nx >= bnd || ny >= bnd || nz >= bnd) return;
  const int fd = T.full_depth;
  // start at the dense full layer
  int sh = t - fd;
  int64_t idx = ((int64_t)b << (3 * fd)) + (int64_t)ofx_xyz2morton(nx >> sh, ny >> sh, nz >> sh);
  for (int s = fd; s < t; ++s) {
    const int32_t c = T.child[T.ncum[s] + idx];
    if (c < 0) { emit(node2row(T, d, s, idx)); return; }       // coarser leaf
    sh = t - s - 1;
    const int o = (((nx >> sh) & 1) << 2) | (((ny >> sh) & 1) << 1) | ((nz >> sh) & 1);
    idx = (int64_t)c * 8 + o;
  }
  if (t == d) { emit(node2row(T, d, d, idx)); return; }
  // same-size cell at depth t < d: leaf -> one neighbour, else expand the touching face
  const int back = dir ^ 1;                                    // face of the neighbour that touches us
  int32_t c0 = T.child[T.ncum[t] + idx];
  if (c0 < 0) { emit(node2row(T, d, t, idx)); return; }
  // DFS over face descendants; depth difference <= OFX_MAX_DEPTH
  int64_t stack_idx[OFX_MAX_DEPTH];
  int stack_k[OFX_MAX_DEPTH];
  int sp = 0;
  int cur_t = t + 1;                 // depth of the children we are iterating
  stack_idx[0] = (int64_t)c0 * 8;    // base index of the children block at depth cur_t
  stack_k[0] = 0;
  while (sp >= 0) {
    if (stack_k[sp] == 4) { --sp; --cur_t; continue; }
    const int k = stack_k[sp]++;
    const int64_t ci = stack_idx[sp] + face_octant(back, k);
    if (cur_t == d) { emit(node2row(T, d, d, ci)); continue; }
    const int32_t cc = T.child[T.ncum[cur_t] + ci];
    if (cc < 0) { emit(node2row(T, d, cur_t, ci)); continue; }
    ++sp; ++cur_t;
    stack_idx[sp] = (int64_t)cc * 8;
    stack_k[sp] = 0;
  }
}

__global__ void __launch_bounds__(256) graph_count_kernel(TreeDev T, int d, int64_t N, int32_t* __restrict__ seg_cnt) {
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < N; r += (int64_t)gridDim.x * blockDim.x) {
    int t; int64_t g;
    row2node(T, d, r, t, g);
    int total = 0;
    for (int dir = 0; dir < 6; ++dir) {
      int cnt = 0;
      for_each_neighbour(T, d, t, g, dir, [&](int32_t) { ++cnt; });
      seg_cnt[r * 7 + dir] = cnt;
      total += cnt;
    }
    seg_cnt[r * 7 + 6] = total > 0 ? 1 : 0;   // add_self_loops: every row that has an edge
  }
}

__global__ void __launch_bounds__(256) graph_fill_kernel(TreeDev T, int d, int64_t N,
                                                         const int32_t* __restrict__ seg_ptr,
                                                         int32_t* __restrict__ col) {
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < N; r += (int64_t)gridDim.x * blockDim.x) {
    int t; int64_t g;
    row2node(T, d, r, t, g);
    for (int dir = 0; dir < 6; ++dir) {
      int32_t p = seg_ptr[r * 7 + dir];
      for_each_neighbour(T, d, t, g, dir, [&](int32_t id) { col[p++] = id; });
    }
    if (seg_ptr[r * 7 + 7] > seg_ptr[r * 7 + 6]) col[seg_ptr[r * 7 + 6]] = (int32_t)r;
  }
}

__global__ void __launch_bounds__(256) graph_nodes_kernel(TreeDev T, int d, int64_t N, int32_t* __restrict__ batch_id,
                                                          uint8_t* __restrict__ node_type, int64_t* __restrict__ keyd,
                                                          uint8_t* __restrict__ node_mask) {
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < N; r += (int64_t)gridDim.x * blockDim.x) {
    int t; int64_t g;
    row2node(T, d, r, t, g);
    const int64_t key = T.key[g];
    if (batch_id) batch_id[r] = (int32_t)((uint64_t)key >> 48);
    if (node_type) node_type[r] = (uint8_t)(t - T.full_depth);
    if (keyd) keyd[r] = key | ((int64_t)t << 58);
    if (node_mask) node_mask[r] = 1;
  }
}

// node_mask of the depth-d graph has length leaf_base[d] + nnum[d] but marks which of
// [all nodes of fd..d-1 | nodes of d] ... see dual_octree.py:391-398: it is the
// concatenation of the leaf masks of depths fd..d-1 (over ALL nodes of those depths) and
// ones(nnum[d]).  Separate kernel because its length differs from N_d.
__global__ void node_mask_kernel(TreeDev T, int d, int64_t total, uint8_t* __restrict__ mask) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t g = T.ncum[T.full_depth] + i;    // tree node, depths fd..d concatenated
    mask[i] = (g >= T.ncum[d]) ? 1 : (T.child[g] < 0);
  }
}

__global__ void leaf_flag_kernel(const int32_t* __restrict__ child, int64_t n, int32_t* __restrict__ flag) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    flag[i] = child[i] < 0;
}

extern "C" size_t ofx_tree_leafrank_ws_bytes(int64_t max_nnum) {
  const int64_t mx = max_nnum > 1 ? max_nnum : 1;
  return ((ofx_scan_ws_bytes(mx) + 255) & ~(size_t)255) + (((size_t)mx * 4 + 255) & ~(size_t)255) +
         (size_t)(mx + 1) * 4 + 256;
}

extern "C" int ofx_tree_leafrank(const int32_t* child_all, const int64_t* nnum_host, int depth,
                                 int32_t* leafrank_all, void* ws, void* stream) {
  if (!child_all || !nnum_host || !leafrank_all || !ws || depth < 0 || depth > OFX_MAX_DEPTH) return OFX_EINVAL;
  // ws layout: [scan workspace | flag buffer (max nnum) | scan out (max nnum + 1)]
  int64_t mx = 1;
  for (int d = 0; d <= depth; ++d) mx = nnum_host[d] > mx ? nnum_host[d] : mx;
  char* base = (char*)ws;
  size_t off = (ofx_scan_ws_bytes(mx) + 255) & ~(size_t)255;
  int32_t* flag = (int32_t*)(base + off);
  off += ((size_t)mx * 4 + 255) & ~(size_t)255;
  int32_t* sout = (int32_t*)(base + off);
  hipStream_t st = ofx_stream(stream);
  int64_t c = 0;
  for (int d = 0; d <= depth; ++d) {
    const int64_t n = nnum_host[d];
    if (n > 0) {
      leaf_flag_kernel<<<ofx_grid(n, 256), 256, 0, st>>>(child_all + c, n, flag);
      int rc = ofx_scan_i32(flag, sout, n, ws, stream);
      if (rc) return rc;
      if (hipMemcpyAsync(leafrank_all + c, sout, (size_t)n * 4, hipMemcpyDeviceToDevice, st) != hipSuccess)
        return OFX_ELAUNCH;
    }
    c += n;
  }
  OFX_LAUNCH_CHECK();
  return OFX_OK;
}

extern "C" int ofx_graph_nodes(const ofx_tree_t* tree, int d, int32_t* batch_id, uint8_t* node_type,
                               int64_t* keyd, uint8_t* node_mask, void* stream) {
  TreeDev T;
  int rc = make_tree(tree, T);
  if (rc) return rc;
  if (d < T.full_depth || d > T.depth || !T.leafrank) return OFX_EINVAL;
  const int64_t N = graph_nodes(T, d);
  hipStream_t st = ofx_stream(stream);
  if (batch_id || node_type || keyd)
    graph_nodes_kernel<<<ofx_grid(N, 256), 256, 0, st>>>(T, d, N, batch_id, node_type, keyd, nullptr);
  if (node_mask) {
    const int64_t total = T.ncum[d] + T.nnum[d] - T.ncum[T.full_depth];
    node_mask_kernel<<<ofx_grid(total, 256), 256, 0, st>>>(T, d, total, node_mask);
  }
  OFX_LAUNCH_CHECK();
  return OFX_OK;
}

extern "C" int ofx_graph_count(const ofx_tree_t* tree, int d, int32_t* seg_cnt, void* stream) {
  TreeDev T;
  int rc = make_tree(tree, T);
  if (rc) return rc;
  if (d < T.full_depth || d > T.depth || !seg_cnt || !T.leafrank) return OFX_EINVAL;
  const int64_t N = graph_nodes(T, d);
  graph_count_kernel<<<ofx_grid(N, 256), 256, 0, ofx_stream(stream)>>>(T, d, N, seg_cnt);
  OFX_LAUNCH_CHECK();
  return OFX_OK;
}

extern "C" int ofx_graph_fill(const ofx_tree_t* tree, int d, const int32_t* seg_ptr, int32_t* col, void* stream) {
  TreeDev T;
  int rc = make_tree(tree, T);
  if (rc) return rc;
  if (d < T.full_depth || d > T.depth || !seg_ptr || !col || !T.leafrank) return OFX_EINVAL;
  const int64_t N = graph_nodes(T, d);
  graph_fill_kernel<<<ofx_grid(N, 256), 256, 0, ofx_stream(stream)>>>(T, d, N, seg_ptr, col);
  OFX_LAUNCH_CHECK();
  return OFX_OK;
}

__global__ void graph_expand_kernel(const int32_t* __restrict__ seg_ptr, int64_t nseg, const int32_t* __restrict__ col,
                                    int64_t* __restrict__ row_out, int64_t* __restrict__ col_out,
                                    int64_t* __restrict__ dir_out) {
  for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < nseg; s += (int64_t)gridDim.x * blockDim.x) {
    const int32_t a = seg_ptr[s], e = seg_ptr[s + 1];
    const int64_t r = s / 7, dir = s - r * 7;
    for (int32_t p = a; p < e; ++p) {
      if (row_out) row_out[p] = r;
      if (dir_out) dir_out[p] = dir;
      if (col_out) col_out[p] = col[p];
    }
  }
}

extern "C" int ofx_graph_expand(const int32_t* seg_ptr, int64_t n_nodes, const int32_t* col, int64_t* row_out,
                                int64_t* col_out, int64_t* dir_out, void* stream) {
  if (!seg_ptr || n_nodes < 0 || (col_out && !col)) return OFX_EINVAL;
  graph_expand_kernel<<<ofx_grid(n_nodes * 7, 256), 256, 0, ofx_stream(stream)>>>(seg_ptr, n_nodes * 7, col, row_out,
                                                                                 col_out, dir_out);
  OFX_LAUNCH_CHECK();
  return OFX_OK;
}

__global__ void type_frac_kernel(const int32_t* __restrict__ seg_ptr, const int32_t* __restrict__ col,
                                 const uint8_t* __restrict__ node_type, int64_t N, int nt, float* __restrict__ tf,
                                 int64_t ld) {
  // one thread per (row, dir); writes nt floats.  Pad columns are zeroed by thread dir==6.
  const int64_t nseg = N * 7;
  for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < nseg; s += (int64_t)gridDim.x * blockDim.x) {
    const int32_t a = seg_ptr[s], e = seg_ptr[s + 1];
    const int64_t r = s / 7;
    const int dir = (int)(s - r * 7);
    float* o = tf + r * ld + dir * nt;
    const float inv = 1.f / (float)(e - a > 1 ? e - a : 1);
    for (int t = 0; t < nt; ++t) {
      int c = 0;
      for (int32_t p = a; p < e; ++p) c += node_type[col[p]] == t;
      o[t] = (float)c * inv;
    }
    if (dir == 6)
      for (int64_t k = 7 * nt; k < ld; ++k) tf[r * ld + k] = 0.f;
  }
}

extern "C" int ofx_graph_type_frac(const int32_t* seg_ptr, const int32_t* col, const uint8_t* node_type,
                                   int64_t n_nodes, int nt, float* type_frac, int64_t ld, void* stream) {
  if (!seg_ptr || !col || !node_type || !type_frac || nt < 1 || ld < 7 * nt || n_nodes < 0) return OFX_EINVAL;
  type_frac_kernel<<<ofx_grid(n_nodes * 7, 256), 256, 0, ofx_stream(stream)>>>(seg_ptr, col, node_type, n_nodes, nt,
                                                                              type_frac, ld);
  OFX_LAUNCH_CHECK();
  return OFX_OK;
}

// primary-neighbour table for the fused GraphConv: nbr[r*7+dir] = the single neighbour, -1 if the
// segment is empty, -2 if it holds several neighbours (the kernel then walks the CSR segment).
__global__ void graph_primary_kernel(const int32_t* __restrict__ seg_ptr, const int32_t* __restrict__ col, int64_t nseg,
                                     int32_t* __restrict__ nbr) {
  for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < nseg; s += (int64_t)gridDim.x * blockDim.x) {
    const int32_t a = seg_ptr[s], e = seg_ptr[s + 1];
    nbr[s] = e == a ? -1 : (e - a == 1 ? col[a] : -2);
  }
}
extern "C" int ofx_graph_primary(const int32_t* seg_ptr, const int32_t* col, int64_t n_nodes, int32_t* nbr,
                                 void* stream) {
  if (!seg_ptr || !col || !nbr || n_nodes < 0) return OFX_EINVAL;
  graph_primary_kernel<<<ofx_grid(n_nodes * 7, 256), 256, 0, ofx_stream(stream)>>>(seg_ptr, col, n_nodes * 7, nbr);
  OFX_LAUNCH_CHECK();
  return OFX_OK;
}

// Extended table for the branch-free GraphConv kernel: every (row,dir) names a source row:
//   single neighbour -> its id; none -> N (the zero row of the aux buffer);
//   several -> N + 1 + v, where v = rank among multi-neighbour segments (multi_seg[v] = segment id).
__global__ void graph_multi_flag_kernel(const int32_t* __restrict__ seg_ptr, int64_t nseg, int32_t* __restrict__ flag) {
  for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < nseg; s += (int64_t)gridDim.x * blockDim.x)
    flag[s] = (seg_ptr[s + 1] - seg_ptr[s]) > 1;
}
__global__ void graph_primary_ext_kernel(const int32_t* __restrict__ seg_ptr, const int32_t* __restrict__ col,
                                         int64_t nseg, int64_t N, const int32_t* __restrict__ rank,
                                         int32_t* __restrict__ nbr_ext, int32_t* __restrict__ multi_seg) {
  for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < nseg; s += (int64_t)gridDim.x * blockDim.x) {
    const int32_t a = seg_ptr[s], e = seg_ptr[s + 1];
    if (e == a) nbr_ext[s] = (int32_t)N;
    else if (e - a == 1) nbr_ext[s] = col[a];
    else { nbr_ext[s] = (int32_t)(N + 1 + rank[s]); multi_seg[rank[s]] = (int32_t)s; }
  }
}
extern "C" int ofx_graph_multi_flag(const int32_t* seg_ptr, int64_t n_nodes, int32_t* flag, void* stream) {
  if (!seg_ptr || !flag || n_nodes < 0) return OFX_EINVAL;
  graph_multi_flag_kernel<<<ofx_grid(n_nodes * 7, 256), 256, 0, ofx_stream(stream)>>>(seg_ptr, n_nodes * 7, flag);
  OFX_LAUNCH_CHECK();
  return OFX_OK;
}
extern "C" int ofx_graph_primary_ext(const int32_t* seg_ptr, const int32_t* col, int64_t n_nodes, const int32_t* rank,
                                     int32_t* nbr_ext, int32_t* multi_seg, void* stream) {
  if (!seg_ptr || !col || !rank || !nbr_ext || n_nodes < 0) return OFX_EINVAL;
  graph_primary_ext_kernel<<<ofx_grid(n_nodes * 7, 256), 256, 0, ofx_stream(stream)>>>(seg_ptr, col, n_nodes * 7,
                                                                                      n_nodes, rank, nbr_ext, multi_seg);
  OFX_LAUNCH_CHECK();
  return OFX_OK;
}

// ---------------------------------------------------------------------------------
// NeuralMPU SDF evaluation (reference models/networks/dualoctree_networks/mpu.py:55-153 with the two sparse
// products of utils/spmm.py:12-61 folded in; sweep driver utils/util_dualoctree.py:99-118).
// Eight lanes per query point, one per surrounding cell centre.  Per depth the lane needs the octree node of
// its cell: at the first depth it walks down from the dense full layer; afterwards the parent of every
// depth-(d+1) centre is one of the point's eight depth-d centres (base_{d+1} in {2b, 2b+1, 2b+2}), so the
// node index comes from a neighbouring lane's previous result with ONE child lookup -- no key search, no sort,
// no scatter: the reference's search_key + index_select + scatter_add chain collapses into registers.
// HBM-bound: 16 B per point in (or nothing for the grid sweep), 5 B out, child / code gathers mostly from L2.
struct MpuArgs {
  TreeDev T;
  int ds, de;
  const float* pts;          // [n, 4] (x, y, z in [-1, 1], batch id) or NULL for the grid sweep
  int64_t n;
  const float* code;         // [sum_{d=ds..de} nnum[d], 4]
  float* sdf;
  uint8_t* mask;
  int size, batch;           // grid sweep: point q = head + i, (ix, iy, iz) = unravel(q, size^3), x slowest
  float step, bbmin;
  int64_t head;
};

__device__ __forceinline__ int64_t mpu_walk(const TreeDev& T, int d, int x, int y, int z, int b) {
  const int fd = T.full_depth < d ? T.full_depth : d;
  int sh = d - fd;
  int64_t idx = ((int64_t)b << (3 * fd)) + (int64_t)ofx_xyz2morton(x >> sh, y >> sh, z >> sh);
  for (int s = fd; s < d; ++s) {
    const int32_t c = T.child[T.ncum[s] + idx];
    if (c < 0) return -1;
    sh = d - s - 1;
    idx = (int64_t)c * 8 + ((((x >> sh) & 1) << 2) | (((y >> sh) & 1) << 1) | ((z >> sh) & 1));
  }
  return idx;
}

__global__ void __launch_bounds__(256) mpu_eval_kernel(const MpuArgs a) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t q = t >> 3;
  const int corner = (int)(t & 7);
  const bool live = q < a.n;
  const int64_t qc = live ? q : a.n - 1;
  float px, py, pz;
  int b;
  if (a.pts) {
    const float4 p = reinterpret_cast<const float4*>(a.pts)[qc];
    px = p.x; py = p.y; pz = p.z; b = (int)p.w;
  } else {
    // samples = mgrid * ((bbmax - bbmin) / size) + bbmin in fp32, two roundings (util_dualoctree.py:102-103)
    const int64_t g = a.head + qc;
    const int64_t s2 = (int64_t)a.size * a.size;
    const int ix = (int)(g / s2), iy = (int)((g / a.size) % a.size), iz = (int)(g % a.size);
    float mx = (float)ix * a.step, my = (float)iy * a.step, mz = (float)iz * a.step;
    asm volatile("" : "+v"(mx), "+v"(my), "+v"(mz));            // keep the product rounded: no fma contraction
    px = mx + a.bbmin; py = my + a.bbmin; pz = mz + a.bbmin;
    b = a.batch;
  }
  const int dx = (corner >> 2) & 1, dy = (corner >> 1) & 1, dz = corner & 1;      // mpu.py:37-52
  const int lane_base = (int)(threadIdx.x & 63) & ~7;
  float num = 0.f, den = 0.f;
  bool found_last = false;
  int64_t base_rows = 0;
  int prev_idx = -1;
  int pbx = 0, pby = 0, pbz = 0;
  const bool bok = b >= 0 && b < a.T.batch_size;
  for (int d = a.ds; d <= a.de; ++d) {
    const int scale = 1 << d;
    const float half = 0.5f * (float)scale;
    // (p + 1) * scale/2 - 0.5: the product is exact (power of two), so contraction cannot change it
    const float x = (px + 1.0f) * half - 0.5f, y = (py + 1.0f) * half - 0.5f, z = (pz + 1.0f) * half - 0.5f;
    const float bxf = floorf(x), byf = floorf(y), bzf = floorf(z);
    const int bx = (int)bxf, by = (int)byf, bz = (int)bzf;
    const int cx = bx + dx, cy = by + dy, cz = bz + dz;
    const float fx = x - (bxf + (float)dx), fy = y - (byf + (float)dy), fz = z - (bzf + (float)dz);
    const bool inb = bok && cx >= 0 && cy >= 0 && cz >= 0 && cx < scale && cy < scale && cz < scale;
    int idx = -1;
    if (d == a.ds) {
      if (inb) idx = (int)mpu_walk(a.T, d, cx, cy, cz, b);
    } else {
      const int pcx = (cx >> 1) - pbx, pcy = (cy >> 1) - pby, pcz = (cz >> 1) - pbz;
      const bool pc_ok = (unsigned)pcx < 2u && (unsigned)pcy < 2u && (unsigned)pcz < 2u;
      const int sl = (inb && pc_ok) ? ((pcx << 2) | (pcy << 1) | pcz) : 0;
      const int pidx = __shfl(prev_idx, lane_base + sl);           // every lane takes part
      if (inb && pc_ok) {
        if (pidx >= 0) {
          const int32_t c = a.T.child[a.T.ncum[d - 1] + pidx];
          if (c >= 0) idx = c * 8 + (((cx & 1) << 2) | ((cy & 1) << 1) | (cz & 1));
        }
      } else if (inb) {
        idx = (int)mpu_walk(a.T, d, cx, cy, cz, b);                // not reachable in exact arithmetic; kept for safety
      }
    }
    const bool found = idx >= 0;
    if (d == a.de) found_last = found;
    bool use = found;
    if (found && d < a.de) use = a.T.child[a.T.ncum[d] + idx] < 0;                // leaves only (mpu.py:113-116)
    if (use) {
      const float4 c = reinterpret_cast<const float4*>(a.code)[base_rows + idx];
      const float s = 2.0f / (float)scale;
      const float val = c.x * (fx * s) + c.y * (fy * s) + c.z * (fz * s) + c.w;
      const float w = ((1.0f - fabsf(fx)) * (1.0f - fabsf(fy))) * (1.0f - fabsf(fz)) * (float)((double)(d * d) / 50.0);
      num += w * val;
      den += w;
    }
    prev_idx = idx;
    pbx = bx; pby = by; pbz = bz;
    base_rows += a.T.nnum[d];
  }
#pragma unroll
  for (int o = 1; o < 8; o <<= 1) {
    num += __shfl_xor(num, o);
    den += __shfl_xor(den, o);
  }
  const unsigned long long bal = __ballot(found_last);
  if (corner == 0 && live) {
    a.sdf[q] = num / (den + 1e-8f);
    if (a.mask) a.mask[q] = ((bal >> ((threadIdx.x & 63) & ~7)) & 0xffull) ? 1 : 0;
  }
}

static int mpu_launch(const ofx_tree_t* tree, MpuArgs& a, hipStream_t st) {
  int rc = make_tree(tree, a.T);
  if (rc) return rc;
  if (a.ds < 0 || a.de < a.ds || a.de > a.T.depth || !a.code || !a.sdf || a.n < 0) return OFX_EINVAL;
  if (((uintptr_t)a.code & 15) != 0) return OFX_EINVAL;
  if (a.n == 0) return OFX_OK;
  const int64_t threads = a.n * 8;
  mpu_eval_kernel<<<(unsigned)((threads + 255) / 256), 256, 0, st>>>(a);
  OFX_LAUNCH_CHECK();
  return OFX_OK;
}

extern "C" int ofx_mpu_eval(const ofx_tree_t* tree, int depth_start, int depth_end, const float* pts, int64_t n_pts,
                            const float* code, float* sdf, uint8_t* mask, void* stream) {
  if (!pts || ((uintptr_t)pts & 15) != 0 || n_pts > (int64_t(1) << 31)) return OFX_EINVAL;
  MpuArgs a = {};
  a.ds = depth_start; a.de = depth_end; a.pts = pts; a.n = n_pts; a.code = code; a.sdf = sdf; a.mask = mask;
  return mpu_launch(tree, a, ofx_stream(stream));
}

extern "C" int ofx_mpu_eval_grid(const ofx_tree_t* tree, int depth_start, int depth_end, const float* code, int size,
                                 float step, float bbmin, int batch_index, int64_t head, int64_t count, float* sdf,
                                 uint8_t* mask, void* stream) {
  if (size < 1 || head < 0 || count < 0 || head + count > (int64_t)size * size * size || count > (int64_t(1) << 31))
    return OFX_EINVAL;
  MpuArgs a = {};
  a.ds = depth_start; a.de = depth_end; a.pts = nullptr; a.n = count; a.code = code; a.sdf = sdf; a.mask = mask;
  a.size = size; a.batch = batch_index; a.step = step; a.bbmin = bbmin; a.head = head;
  return mpu_launch(tree, a, ofx_stream(stream));
}

// ---------------------------------------------------------------------------------
// NeuralMPU with gradients -- the training-side use of the same field (reference loss.py:12-20, 100-108:
// compute_mpu_gradients = autograd of get_linear_pred w.r.t. the query position, create_graph=True so the SDF
// losses can be back-propagated into the per-node codes).  Everything autograd derives is written out:
//   w_i = k_d prod_a (1 - |f_ia|),   v_i = c_i . (f_i s_d) + c_i3,   s_d = 2 / 2^d,   f = x 2^d/2 - 1/2 - centre
//   num = sum w_i v_i,  den = sum w_i,  D = den + 1e-8,   sdf = num / D
//   dw_ia = -sgn(f_ia) (2^d / 2) k_d prod_{b != a} (1 - |f_ib|)     (mpu.py:18-32: sgn(0) = +1; floor detached)
//   dv_ia = c_ia                                                     (s_d * 2^d / 2 = 1 exactly)
//   grad_a = (sum_i dw_ia v_i + w_i c_ia) / D - num (sum_i dw_ia) / D^2
// and, given the upstream gradients a = dL/dsdf and b = dL/dgrad, the adjoint w.r.t. the codes (both outputs
// are linear in c):   A_i = a w_i / D + sum_a b_a (dw_ia / D - w_i dden_a / D^2),
//   dL/dc_ia = A_i f_ia s_d + b_a w_i / D,   dL/dc_i3 = A_i   (fp32 atomics into dcode).
// Same eight-lanes-per-point walk as mpu_eval_kernel; the visitor is called once per depth by every lane.
struct MpuVisit {
  bool use;          // the centre exists (and is a leaf below depth_end)
  bool found;
  int64_t row;       // row of the code table
  float fx, fy, fz;  // offset to the centre in cells
  float wx, wy, wz;  // 1 - |f|
  float kd, half, s; // d^2/50, 2^d / 2, 2 / 2^d
};

template <typename Fn>
__device__ __forceinline__ void mpu_for_each(const TreeDev& T, int ds, int de, float px, float py, float pz, int b,
                                             int corner, int lane_base, Fn&& fn) {
  const int dx = (corner >> 2) & 1, dy = (corner >> 1) & 1, dz = corner & 1;
  int64_t base_rows = 0;
  int prev_idx = -1;
  int pbx = 0, pby = 0, pbz = 0;
  const bool bok = b >= 0 && b < T.batch_size;
  for (int d = ds; d <= de; ++d) {
    const int scale = 1 << d;
    const float half = 0.5f * (float)scale;
    const float x = (px + 1.0f) * half - 0.5f, y = (py + 1.0f) * half - 0.5f, z = (pz + 1.0f) * half - 0.5f;
    const float bxf = floorf(x), byf = floorf(y), bzf = floorf(z);
    const int bx = (int)bxf, by = (int)byf, bz = (int)bzf;
    const int cx = bx + dx, cy = by + dy, cz = bz + dz;
    MpuVisit v;
    v.fx = x - (bxf + (float)dx); v.fy = y - (byf + (float)dy); v.fz = z - (bzf + (float)dz);
    const bool inb = bok && cx >= 0 && cy >= 0 && cz >= 0 && cx < scale && cy < scale && cz < scale;
    int idx = -1;
    if (d == ds) {
      if (inb) idx = (int)mpu_walk(T, d, cx, cy, cz, b);
    } else {
      const int pcx = (cx >> 1) - pbx, pcy = (cy >> 1) - pby, pcz = (cz >> 1) - pbz;
      const bool pc_ok = (unsigned)pcx < 2u && (unsigned)pcy < 2u && (unsigned)pcz < 2u;
      const int sl = (inb && pc_ok) ? ((pcx << 2) | (pcy << 1) | pcz) : 0;
      const int pidx = __shfl(prev_idx, lane_base + sl);
      if (inb && pc_ok) {
        if (pidx >= 0) {
          const int32_t c = T.child[T.ncum[d - 1] + pidx];
          if (c >= 0) idx = c * 8 + (((cx & 1) << 2) | ((cy & 1) << 1) | (cz & 1));
        }
      } else if (inb) {
        idx = (int)mpu_walk(T, d, cx, cy, cz, b);
      }
    }
    v.found = idx >= 0;
    v.use = v.found;
    if (v.found && d < de) v.use = T.child[T.ncum[d] + idx] < 0;
    v.row = base_rows + idx;
    v.wx = 1.0f - fabsf(v.fx); v.wy = 1.0f - fabsf(v.fy); v.wz = 1.0f - fabsf(v.fz);
    v.kd = (float)((double)(d * d) / 50.0);
    v.half = half;
    v.s = 2.0f / (float)scale;
    fn(d, v);
    prev_idx = idx;
    pbx = bx; pby = by; pbz = bz;
    base_rows += T.nnum[d];
  }
}

struct MpuGradArgs {
  TreeDev T;
  int ds, de;
  const float* pts;
  int64_t n;
  const float* code;
  float* sdf;          // forward outputs
  float* grad;         // [n, 3]
  uint8_t* mask;
  const float* dsdf;   // backward inputs (NULL: taken as 0)
  const float* dgrad;  // [n, 3]
  float* dcode;        // [rows, 4], accumulated with atomics
};

__device__ __forceinline__ float mpu_sgn(float f) { return f < 0.f ? -1.f : 1.f; }

template <bool BACKWARD>
__global__ void __launch_bounds__(256) mpu_grad_kernel(const MpuGradArgs a) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t q = t >> 3;
  const int corner = (int)(t & 7);
  const bool live = q < a.n;
  const int64_t qc = live ? q : a.n - 1;
  const float4 p = reinterpret_cast<const float4*>(a.pts)[qc];
  const int b = (int)p.w;
  const int lane_base = (int)(threadIdx.x & 63) & ~7;
  float num = 0.f, den = 0.f, dn[3] = {0.f, 0.f, 0.f}, dd[3] = {0.f, 0.f, 0.f};
  bool found_last = false;
  mpu_for_each(a.T, a.ds, a.de, p.x, p.y, p.z, b, corner, lane_base, [&](int d, const MpuVisit& v) {
    if (d == a.de) found_last = v.found;
    if (!v.use) return;
    const float4 c = reinterpret_cast<const float4*>(a.code)[v.row];
    const float val = c.x * (v.fx * v.s) + c.y * (v.fy * v.s) + c.z * (v.fz * v.s) + c.w;
    const float w = (v.wx * v.wy) * v.wz * v.kd;
    const float hk = v.half * v.kd;
    const float dwx = -mpu_sgn(v.fx) * hk * (v.wy * v.wz);
    const float dwy = -mpu_sgn(v.fy) * hk * (v.wx * v.wz);
    const float dwz = -mpu_sgn(v.fz) * hk * (v.wx * v.wy);
    num += w * val; den += w;
    dn[0] += dwx * val + w * c.x; dn[1] += dwy * val + w * c.y; dn[2] += dwz * val + w * c.z;
    dd[0] += dwx; dd[1] += dwy; dd[2] += dwz;
  });
#pragma unroll
  for (int o = 1; o < 8; o <<= 1) {
    num += __shfl_xor(num, o); den += __shfl_xor(den, o);
#pragma unroll
    for (int k = 0; k < 3; ++k) { dn[k] += __shfl_xor(dn[k], o); dd[k] += __shfl_xor(dd[k], o); }
  }
  const float D = den + 1e-8f;
  const float invD = 1.0f / D;
  if (!BACKWARD) {
    const unsigned long long bal = __ballot(found_last);
    if (corner == 0 && live) {
      a.sdf[q] = num * invD;
      const float r = num * invD * invD;
      a.grad[q * 3 + 0] = dn[0] * invD - r * dd[0];
      a.grad[q * 3 + 1] = dn[1] * invD - r * dd[1];
      a.grad[q * 3 + 2] = dn[2] * invD - r * dd[2];
      if (a.mask) a.mask[q] = ((bal >> lane_base) & 0xffull) ? 1 : 0;
    }
    return;
  }
  const float up = (live && a.dsdf) ? a.dsdf[qc] : 0.f;
  float bk[3] = {0.f, 0.f, 0.f};
  if (live && a.dgrad) { bk[0] = a.dgrad[qc * 3]; bk[1] = a.dgrad[qc * 3 + 1]; bk[2] = a.dgrad[qc * 3 + 2]; }
  // second walk: scatter the adjoint (every lane takes part in the shuffles of the walk)
  mpu_for_each(a.T, a.ds, a.de, p.x, p.y, p.z, b, corner, lane_base, [&](int d, const MpuVisit& v) {
    if (!v.use || !live) return;
    const float w = (v.wx * v.wy) * v.wz * v.kd;
    const float hk = v.half * v.kd;
    const float dwx = -mpu_sgn(v.fx) * hk * (v.wy * v.wz);
    const float dwy = -mpu_sgn(v.fy) * hk * (v.wx * v.wz);
    const float dwz = -mpu_sgn(v.fz) * hk * (v.wx * v.wy);
    const float wD = w * invD;
    const float A = up * wD + bk[0] * (dwx - wD * dd[0]) * invD + bk[1] * (dwy - wD * dd[1]) * invD +
                    bk[2] * (dwz - wD * dd[2]) * invD;
    float* o = a.dcode + v.row * 4;
    unsafeAtomicAdd(o + 0, A * (v.fx * v.s) + bk[0] * wD);
    unsafeAtomicAdd(o + 1, A * (v.fy * v.s) + bk[1] * wD);
    unsafeAtomicAdd(o + 2, A * (v.fz * v.s) + bk[2] * wD);
    unsafeAtomicAdd(o + 3, A);
  });
}

static int mpu_grad_check(const ofx_tree_t* tree, MpuGradArgs& a) {
  int rc = make_tree(tree, a.T);
  if (rc) return rc;
  if (a.ds < 0 || a.de < a.ds || a.de > a.T.depth || !a.code || !a.pts || a.n < 0 || a.n > (int64_t(1) << 31))
    return OFX_EINVAL;
  if ((((uintptr_t)a.code) & 15) != 0 || (((uintptr_t)a.pts) & 15) != 0) return OFX_EINVAL;
  return OFX_OK;
}

extern "C" int ofx_mpu_eval_grad(const ofx_tree_t* tree, int depth_start, int depth_end, const float* pts,
                                 int64_t n_pts, const float* code, float* sdf, float* grad, uint8_t* mask,
                                 void* stream) {
  MpuGradArgs a = {};
  a.ds = depth_start; a.de = depth_end; a.pts = pts; a.n = n_pts; a.code = code; a.sdf = sdf; a.grad = grad;
  a.mask = mask;
  int rc = mpu_grad_check(tree, a);
  if (rc) return rc;
  if (!sdf || !grad) return OFX_EINVAL;
  if (n_pts == 0) return OFX_OK;
  mpu_grad_kernel<false><<<(unsigned)((n_pts * 8 + 255) / 256), 256, 0, ofx_stream(stream)>>>(a);
  OFX_LAUNCH_CHECK();
  return OFX_OK;
}

extern "C" int ofx_mpu_backward(const ofx_tree_t* tree, int depth_start, int depth_end, const float* pts,
                                int64_t n_pts, const float* code, const float* dsdf, const float* dgrad,
                                float* dcode, void* stream) {
  MpuGradArgs a = {};
  a.ds = depth_start; a.de = depth_end; a.pts = pts; a.n = n_pts; a.code = code; a.dsdf = dsdf; a.dgrad = dgrad;
  a.dcode = dcode;
  int rc = mpu_grad_check(tree, a);
  if (rc) return rc;
  if (!dcode || (!dsdf && !dgrad)) return OFX_EINVAL;
  if (n_pts == 0) return OFX_OK;
  mpu_grad_kernel<true><<<(unsigned)((n_pts * 8 + 255) / 256), 256, 0, ofx_stream(stream)>>>(a);
  OFX_LAUNCH_CHECK();
  return OFX_OK;
}

// ---------------------------------------------------------------------------------
// Reverse graph for the backward pass of GraphConv (reference: autograd of modules.py:194-220, i.e. of
// index_select + scatter_mean): forward segment (r, dir) averages x[col] over its cnt edges, so
//   dx[c] = sum over forward edges e with col_e = c of dcol[row_e, dir_e] / cnt(row_e, dir_e).
// The reverse CSR is keyed by (c, dir): rev_row[e] = row_e, rev_w[e] = 1 / cnt(row_e, dir_e), each reverse
// segment sorted by row so the summation order is fixed.
__global__ void rev_count_kernel(const int32_t* __restrict__ seg_ptr, const int32_t* __restrict__ col, int64_t nseg,
                                 int32_t* __restrict__ rcnt) {
  for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < nseg; s += (int64_t)gridDim.x * blockDim.x) {
    const int dir = (int)(s % 7);
    for (int32_t p = seg_ptr[s]; p < seg_ptr[s + 1]; ++p) atomicAdd(&rcnt[(int64_t)col[p] * 7 + dir], 1);
  }
}
__global__ void rev_fill_kernel(const int32_t* __restrict__ seg_ptr, const int32_t* __restrict__ col, int64_t nseg,
                                const int32_t* __restrict__ rev_ptr, int32_t* __restrict__ cursor,
                                int32_t* __restrict__ rev_row, float* __restrict__ rev_w) {
  for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < nseg; s += (int64_t)gridDim.x * blockDim.x) {
    const int dir = (int)(s % 7);
    const int32_t a = seg_ptr[s], e = seg_ptr[s + 1];
    const float w = 1.0f / (float)(e - a > 0 ? e - a : 1);
    for (int32_t p = a; p < e; ++p) {
      const int64_t key = (int64_t)col[p] * 7 + dir;
      const int32_t pos = rev_ptr[key] + atomicAdd(&cursor[key], 1);
      rev_row[pos] = (int32_t)(s / 7);
      rev_w[pos] = w;
    }
  }
}
__global__ void rev_sort_kernel(const int32_t* __restrict__ rev_ptr, int64_t nseg, int32_t* __restrict__ rev_row,
                                float* __restrict__ rev_w) {
  for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < nseg; s += (int64_t)gridDim.x * blockDim.x) {
    const int32_t a = rev_ptr[s], e = rev_ptr[s + 1];
    for (int32_t i = a + 1; i < e; ++i) {                    // insertion sort: segments hold <= a handful of edges
      const int32_t r = rev_row[i];
      const float w = rev_w[i];
      int32_t j = i - 1;
      while (j >= a && rev_row[j] > r) { rev_row[j + 1] = rev_row[j]; rev_w[j + 1] = rev_w[j]; --j; }
      rev_row[j + 1] = r; rev_w[j + 1] = w;
    }
  }
}
extern "C" int ofx_graph_reverse_count(const int32_t* seg_ptr, const int32_t* col, int64_t n_nodes, int32_t* rev_cnt,
                                       void* stream) {
  if (!seg_ptr || !rev_cnt || n_nodes < 0) return OFX_EINVAL;
  if (n_nodes == 0) return OFX_OK;
  if (hipMemsetAsync(rev_cnt, 0, (size_t)n_nodes * 7 * sizeof(int32_t), ofx_stream(stream)) != hipSuccess) return OFX_ELAUNCH;
  rev_count_kernel<<<ofx_grid(n_nodes * 7, 256), 256, 0, ofx_stream(stream)>>>(seg_ptr, col, n_nodes * 7, rev_cnt);
  OFX_LAUNCH_CHECK();
  return OFX_OK;
}
extern "C" int ofx_graph_reverse_fill(const int32_t* seg_ptr, const int32_t* col, int64_t n_nodes, const int32_t* rev_ptr,
                                      int32_t* cursor, int32_t* rev_row, float* rev_w, void* stream) {
  if (!seg_ptr || !rev_ptr || !cursor || !rev_row || !rev_w || n_nodes < 0) return OFX_EINVAL;
  if (n_nodes == 0) return OFX_OK;
  hipStream_t st = ofx_stream(stream);
  if (hipMemsetAsync(cursor, 0, (size_t)n_nodes * 7 * sizeof(int32_t), st) != hipSuccess) return OFX_ELAUNCH;
  rev_fill_kernel<<<ofx_grid(n_nodes * 7, 256), 256, 0, st>>>(seg_ptr, col, n_nodes * 7, rev_ptr, cursor, rev_row, rev_w);
  rev_sort_kernel<<<ofx_grid(n_nodes * 7, 256), 256, 0, st>>>(rev_ptr, n_nodes * 7, rev_row, rev_w);
  OFX_LAUNCH_CHECK();
  return OFX_OK;
}

// weighted variants of graph_primary / multi_flag / primary_ext: a segment is "simple" (one source row, used as
// is) only when it has exactly one edge of weight 1
__global__ void graph_primary_w_kernel(const int32_t* __restrict__ seg_ptr, const int32_t* __restrict__ col,
                                       const float* __restrict__ w, int64_t nseg, int32_t* __restrict__ nbr) {
  for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < nseg; s += (int64_t)gridDim.x * blockDim.x) {
    const int32_t a = seg_ptr[s], e = seg_ptr[s + 1];
    nbr[s] = e == a ? -1 : ((e - a == 1 && w[a] == 1.0f) ? col[a] : -2);
  }
}
__global__ void graph_multi_flag_w_kernel(const int32_t* __restrict__ seg_ptr, const float* __restrict__ w, int64_t nseg,
                                          int32_t* __restrict__ flag) {
  for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < nseg; s += (int64_t)gridDim.x * blockDim.x) {
    const int32_t a = seg_ptr[s], e = seg_ptr[s + 1];
    flag[s] = (e - a > 1) || (e - a == 1 && w[a] != 1.0f);
  }
}
__global__ void graph_primary_ext_w_kernel(const int32_t* __restrict__ seg_ptr, const int32_t* __restrict__ col,
                                           const float* __restrict__ w, int64_t nseg, int64_t N,
                                           const int32_t* __restrict__ rank, int32_t* __restrict__ nbr_ext,
                                           int32_t* __restrict__ multi_seg) {
  for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < nseg; s += (int64_t)gridDim.x * blockDim.x) {
    const int32_t a = seg_ptr[s], e = seg_ptr[s + 1];
    if (e == a) nbr_ext[s] = (int32_t)N;
    else if (e - a == 1 && w[a] == 1.0f) nbr_ext[s] = col[a];
    else { nbr_ext[s] = (int32_t)(N + 1 + rank[s]); multi_seg[rank[s]] = (int32_t)s; }
  }
}
extern "C" int ofx_graph_primary_w(const int32_t* seg_ptr, const int32_t* col, const float* w, int64_t n_nodes,
                                   int32_t* nbr, void* stream) {
  if (!seg_ptr || !nbr || n_nodes < 0) return OFX_EINVAL;
  graph_primary_w_kernel<<<ofx_grid(n_nodes * 7, 256), 256, 0, ofx_stream(stream)>>>(seg_ptr, col, w, n_nodes * 7, nbr);
  OFX_LAUNCH_CHECK();
  return OFX_OK;
}
extern "C" int ofx_graph_multi_flag_w(const int32_t* seg_ptr, const float* w, int64_t n_nodes, int32_t* flag,
                                      void* stream) {
  if (!seg_ptr || !flag || n_nodes < 0) return OFX_EINVAL;
  graph_multi_flag_w_kernel<<<ofx_grid(n_nodes * 7, 256), 256, 0, ofx_stream(stream)>>>(seg_ptr, w, n_nodes * 7, flag);
  OFX_LAUNCH_CHECK();
  return OFX_OK;
}
extern "C" int ofx_graph_primary_ext_w(const int32_t* seg_ptr, const int32_t* col, const float* w, int64_t n_nodes,
                                       const int32_t* rank, int32_t* nbr_ext, int32_t* multi_seg, void* stream) {
  if (!seg_ptr || !rank || !nbr_ext || n_nodes < 0) return OFX_EINVAL;
  graph_primary_ext_w_kernel<<<ofx_grid(n_nodes * 7, 256), 256, 0, ofx_stream(stream)>>>(
      seg_ptr, col, w, n_nodes * 7, n_nodes, rank, nbr_ext, multi_seg);
  OFX_LAUNCH_CHECK();
  return OFX_OK;
}

// ---- the same machinery for any segment structure (27-tap grid tables of the dense layers) -----------------
// Dense tap table nbr[r, t] (valid source row in [0, n_in), anything else = padding) -> reverse CSR keyed by
// (source row, tap), all weights 1.
__global__ void tab_rev_count_kernel(const int32_t* __restrict__ nbr, int64_t total, int ndir, int64_t n_in,
                                     int32_t* __restrict__ rcnt) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int32_t s = nbr[e];
    if (s >= 0 && s < n_in) atomicAdd(&rcnt[(int64_t)s * ndir + (int)(e % ndir)], 1);
  }
}
__global__ void tab_rev_fill_kernel(const int32_t* __restrict__ nbr, int64_t total, int ndir, int64_t n_in,
                                    const int32_t* __restrict__ rev_ptr, int32_t* __restrict__ cursor,
                                    int32_t* __restrict__ rev_row, float* __restrict__ rev_w) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int32_t s = nbr[e];
    if (s < 0 || s >= n_in) continue;
    const int64_t key = (int64_t)s * ndir + (int)(e % ndir);
    const int32_t pos = rev_ptr[key] + atomicAdd(&cursor[key], 1);
    rev_row[pos] = (int32_t)(e / ndir);
    rev_w[pos] = 1.0f;
  }
}
extern "C" int ofx_table_reverse_count(const int32_t* nbr, int64_t n_out, int ndir, int64_t n_in, int32_t* rev_cnt,
                                       void* stream) {
  if (!nbr || !rev_cnt || n_out < 0 || n_in < 0 || ndir < 1) return OFX_EINVAL;
  if (n_in == 0) return OFX_OK;
  if (hipMemsetAsync(rev_cnt, 0, (size_t)n_in * ndir * sizeof(int32_t), ofx_stream(stream)) != hipSuccess) return OFX_ELAUNCH;
  if (n_out > 0)
    tab_rev_count_kernel<<<ofx_grid(n_out * ndir, 256), 256, 0, ofx_stream(stream)>>>(nbr, n_out * ndir, ndir, n_in, rev_cnt);
  OFX_LAUNCH_CHECK();
  return OFX_OK;
}
extern "C" int ofx_table_reverse_fill(const int32_t* nbr, int64_t n_out, int ndir, int64_t n_in, const int32_t* rev_ptr,
                                      int32_t* cursor, int32_t* rev_row, float* rev_w, void* stream) {
  if (!nbr || !rev_ptr || !cursor || !rev_row || !rev_w || n_out < 0 || n_in < 0 || ndir < 1) return OFX_EINVAL;
  if (n_in == 0 || n_out == 0) return OFX_OK;
  hipStream_t st = ofx_stream(stream);
  if (hipMemsetAsync(cursor, 0, (size_t)n_in * ndir * sizeof(int32_t), st) != hipSuccess) return OFX_ELAUNCH;
  tab_rev_fill_kernel<<<ofx_grid(n_out * ndir, 256), 256, 0, st>>>(nbr, n_out * ndir, ndir, n_in, rev_ptr, cursor, rev_row, rev_w);
  rev_sort_kernel<<<ofx_grid(n_in * ndir, 256), 256, 0, st>>>(rev_ptr, n_in * ndir, rev_row, rev_w);
  OFX_LAUNCH_CHECK();
  return OFX_OK;
}
extern "C" int ofx_seg_primary_w(const int32_t* seg_ptr, const int32_t* col, const float* w, int64_t nseg, int32_t* nbr,
                                 void* stream) {
  if (!seg_ptr || !nbr || nseg < 0) return OFX_EINVAL;
  graph_primary_w_kernel<<<ofx_grid(nseg, 256), 256, 0, ofx_stream(stream)>>>(seg_ptr, col, w, nseg, nbr);
  OFX_LAUNCH_CHECK();
  return OFX_OK;
}
extern "C" int ofx_seg_multi_flag_w(const int32_t* seg_ptr, const float* w, int64_t nseg, int32_t* flag, void* stream) {
  if (!seg_ptr || !flag || nseg < 0) return OFX_EINVAL;
  graph_multi_flag_w_kernel<<<ofx_grid(nseg, 256), 256, 0, ofx_stream(stream)>>>(seg_ptr, w, nseg, flag);
  OFX_LAUNCH_CHECK();
  return OFX_OK;
}
extern "C" int ofx_seg_primary_ext_w(const int32_t* seg_ptr, const int32_t* col, const float* w, int64_t nseg,
                                     int64_t n_src, const int32_t* rank, int32_t* nbr_ext, int32_t* multi_seg,
                                     void* stream) {
  if (!seg_ptr || !rank || !nbr_ext || nseg < 0) return OFX_EINVAL;
  graph_primary_ext_w_kernel<<<ofx_grid(nseg, 256), 256, 0, ofx_stream(stream)>>>(seg_ptr, col, w, nseg, n_src, rank,
                                                                                nbr_ext, multi_seg);
  OFX_LAUNCH_CHECK();
  return OFX_OK;
}

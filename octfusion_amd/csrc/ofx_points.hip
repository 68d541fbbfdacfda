// Point cloud -> octree (reference call sites models/octfusion_model_union.py:198-212,
// models/octfusion_model_vae.py:133-141: ocnn `Octree.build_octree` per shape + `merge_octrees`, then
// `InputFeature('ND')` for the VAE encoder, dual_octree.py:343-360).  ocnn builds bottom-up with three
// `torch.unique` passes per depth and one octree per shape; here the whole batch is built top-down in one go:
//   1. one Morton key per point (batch id in bits 48..), ofx_points_keys
//   2. ONE stable radix sort of (key, point index) for the whole batch (rocPRIM device primitive), ofx_points_sort
//   3. per depth: a node is non-empty iff the sorted keys contain a point inside its cell -- a binary search per
//      node (ofx_octree_label_from_points), feeding the same split / grow kernels the sampling path uses
//   4. per finest-depth node: the point range [lower, upper) of its key gives the averaged (renormalised) normal
//      and the averaged position -> the 4-channel 'ND' feature, zero rows for empty nodes (ofx_octree_point_features)
// All of it is HBM-bound integer / gather work; the sort dominates (16 B per point per radix pass).
#include <hipcub/hipcub.hpp>

#include "ofx_common.h"

namespace {

__global__ void __launch_bounds__(256) points_keys_kernel(const float* __restrict__ pts, int64_t ldp,
                                                          const int32_t* __restrict__ batch_id, int batch_const,
                                                          int64_t n, int depth, uint64_t* __restrict__ keys,
                                                          int32_t* __restrict__ idx) {
  const float scale = (float)(1 << (depth - 1));
  const int hi = (1 << depth) - 1;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    // ocnn: points = (p + 1) * 2^(depth-1), key = xyz2key(trunc) -- positions on the upper cube face are kept in
    // the last cell (ocnn's Points.clip keeps p == 1.0)
    int x = (int)((pts[i * ldp + 0] + 1.0f) * scale);
    int y = (int)((pts[i * ldp + 1] + 1.0f) * scale);
    int z = (int)((pts[i * ldp + 2] + 1.0f) * scale);
    x = x < 0 ? 0 : (x > hi ? hi : x);
    y = y < 0 ? 0 : (y > hi ? hi : y);
    z = z < 0 ? 0 : (z > hi ? hi : z);
    const uint64_t b = (uint64_t)(batch_id ? batch_id[i] : batch_const);
    keys[i] = (b << 48) | ofx_xyz2morton(x, y, z);
    idx[i] = (int32_t)i;
  }
}

__device__ __forceinline__ int64_t lower_bound_u64(const uint64_t* __restrict__ a, int64_t n, uint64_t v) {
  int64_t lo = 0, hi = n;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (a[mid] < v) lo = mid + 1; else hi = mid;
  }
  return lo;
}

__global__ void __launch_bounds__(256) label_from_points_kernel(const uint64_t* __restrict__ sorted_keys, int64_t n_pts,
                                                                const int64_t* __restrict__ node_keys, int64_t nnum,
                                                                int shift, int32_t* __restrict__ label) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nnum; i += (int64_t)gridDim.x * 256) {
    const uint64_t k = (uint64_t)node_keys[i];
    const uint64_t b = k >> 48, m = k & ((1ull << 48) - 1);
    const uint64_t lo = (b << 48) | (m << shift);          // first finest-depth key inside the cell
    const uint64_t span = 1ull << shift;
    const int64_t p = lower_bound_u64(sorted_keys, n_pts, lo);
    label[i] = (p < n_pts && sorted_keys[p] - lo < span) ? 1 : 0;
  }
}

__global__ void __launch_bounds__(256) point_features_kernel(const uint64_t* __restrict__ sorted_keys,
                                                             const int32_t* __restrict__ sorted_idx, int64_t n_pts,
                                                             const float* __restrict__ pts, int64_t ldp,
                                                             const float* __restrict__ normals, int64_t ldn,
                                                             const int64_t* __restrict__ node_keys, int64_t nnum,
                                                             int depth, float* __restrict__ feat,
                                                             float* __restrict__ avg_pts, float* __restrict__ avg_nrm) {
  const float scale = (float)(1 << (depth - 1));
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nnum; i += (int64_t)gridDim.x * 256) {
    const uint64_t k = (uint64_t)node_keys[i];
    int64_t p = lower_bound_u64(sorted_keys, n_pts, k);
    float sx = 0.f, sy = 0.f, sz = 0.f, nx = 0.f, ny = 0.f, nz = 0.f;
    int cnt = 0;
    for (; p < n_pts && sorted_keys[p] == k; ++p, ++cnt) {            // stable sort: original point order
      const int64_t j = sorted_idx[p];
      sx += (pts[j * ldp + 0] + 1.0f) * scale; sy += (pts[j * ldp + 1] + 1.0f) * scale; sz += (pts[j * ldp + 2] + 1.0f) * scale;
      if (normals) { nx += normals[j * ldn + 0]; ny += normals[j * ldn + 1]; nz += normals[j * ldn + 2]; }
    }
    float4 f = make_float4(0.f, 0.f, 0.f, 0.f);
    float ax = 0.f, ay = 0.f, az = 0.f;
    if (cnt > 0) {
      const float c = (float)cnt;
      ax = sx / c; ay = sy / c; az = sz / c;                          // ocnn: scatter_add(points) / counts
      const float len = sqrtf(nx * nx + ny * ny + nz * nz);
      const float inv = 1.0f / fmaxf(len, 1e-12f);                    // F.normalize
      nx *= inv; ny *= inv; nz *= inv;
      // 'D': dot(frac(avg point) - 0.5, normal)  (ocnn Octree.get_input_feature)
      const float lx = ax - truncf(ax) - 0.5f, ly = ay - truncf(ay) - 0.5f, lz = az - truncf(az) - 0.5f;
      f = make_float4(nx, ny, nz, lx * nx + ly * ny + lz * nz);
    }
    if (feat) reinterpret_cast<float4*>(feat)[i] = f;
    if (avg_pts) { avg_pts[i * 3] = ax; avg_pts[i * 3 + 1] = ay; avg_pts[i * 3 + 2] = az; }
    if (avg_nrm) { avg_nrm[i * 3] = f.x; avg_nrm[i * 3 + 1] = f.y; avg_nrm[i * 3 + 2] = f.z; }
  }
}

}  // namespace

extern "C" size_t ofx_points_sort_ws_bytes(int64_t n) {
  if (n <= 0) return 16;
  size_t bytes = 0;
  if (hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, (const uint64_t*)nullptr, (uint64_t*)nullptr,
                                         (const int32_t*)nullptr, (int32_t*)nullptr, (int)n, 0, 64,
                                         (hipStream_t)0) != hipSuccess)
    return 0;
  return bytes + 16;
}

extern "C" int ofx_points_keys(const float* pts, int64_t ldp, const int32_t* batch_id, int batch_const, int64_t n,
                               int depth, int64_t* keys, int32_t* idx, void* stream) {
  if (n < 0 || depth < 1 || depth > 16 || ldp < 3 || batch_const < 0 || (n > 0 && (!pts || !keys || !idx)))
    return OFX_EINVAL;
  if (n == 0) return OFX_OK;
  points_keys_kernel<<<ofx_grid(n, 256), 256, 0, ofx_stream(stream)>>>(pts, ldp, batch_id, batch_const, n, depth,
                                                                      reinterpret_cast<uint64_t*>(keys), idx);
  OFX_LAUNCH_CHECK();
  return OFX_OK;
}

extern "C" int ofx_points_sort(const int64_t* keys_in, const int32_t* idx_in, int64_t n, int64_t* keys_out,
                               int32_t* idx_out, void* ws, size_t ws_bytes, void* stream) {
  if (n < 0 || n > (int64_t(1) << 30) || (n > 0 && (!keys_in || !idx_in || !keys_out || !idx_out || !ws)))
    return OFX_EINVAL;
  if (n == 0) return OFX_OK;
  size_t need = ofx_points_sort_ws_bytes(n);
  if (need == 0 || ws_bytes < need) return OFX_EINVAL;
  size_t bytes = ws_bytes;
  if (hipcub::DeviceRadixSort::SortPairs(ws, bytes, reinterpret_cast<const uint64_t*>(keys_in),
                                         reinterpret_cast<uint64_t*>(keys_out), idx_in, idx_out, (int)n, 0, 64,
                                         ofx_stream(stream)) != hipSuccess)
    return OFX_ELAUNCH;
  return OFX_OK;
}

extern "C" int ofx_octree_label_from_points(const int64_t* sorted_keys, int64_t n_pts, const int64_t* node_keys,
                                            int64_t nnum, int depth_pts, int d, int32_t* label, void* stream) {
  if (n_pts < 0 || nnum < 0 || d < 0 || d > depth_pts || depth_pts > 16 || (nnum > 0 && (!node_keys || !label)) ||
      (n_pts > 0 && !sorted_keys))
    return OFX_EINVAL;
  if (nnum == 0) return OFX_OK;
  label_from_points_kernel<<<ofx_grid(nnum, 256), 256, 0, ofx_stream(stream)>>>(
      reinterpret_cast<const uint64_t*>(sorted_keys), n_pts, node_keys, nnum, 3 * (depth_pts - d), label);
  OFX_LAUNCH_CHECK();
  return OFX_OK;
}

extern "C" int ofx_octree_point_features(const int64_t* sorted_keys, const int32_t* sorted_idx, int64_t n_pts,
                                         const float* pts, int64_t ldp, const float* normals, int64_t ldn,
                                         const int64_t* node_keys, int64_t nnum, int depth, float* feat,
                                         float* avg_points, float* avg_normals, void* stream) {
  if (n_pts < 0 || nnum < 0 || depth < 1 || depth > 16 || ldp < 3 || (normals && ldn < 3) ||
      (nnum > 0 && !node_keys) || (n_pts > 0 && (!sorted_keys || !sorted_idx || !pts)) ||
      (feat && (((uintptr_t)feat) & 15) != 0))
    return OFX_EINVAL;
  if (nnum == 0) return OFX_OK;
  point_features_kernel<<<ofx_grid(nnum, 256), 256, 0, ofx_stream(stream)>>>(
      reinterpret_cast<const uint64_t*>(sorted_keys), sorted_idx, n_pts, pts, ldp, normals, ldn, node_keys, nnum,
      depth, feat, avg_points, avg_normals);
  OFX_LAUNCH_CHECK();
  return OFX_OK;
}

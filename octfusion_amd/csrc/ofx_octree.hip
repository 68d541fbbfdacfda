// libofx: scans + octree container kernels (integer / byte work, HBM-bound).
// Replaces the ocnn.octree.Octree operations the reference calls from
// utils/util_dualoctree.py:225-273 and ldm_diffusion_util.py:318-325, and
// ocnn.nn.octree2voxel at the full layer (graph_unet_lr.py:176-181).
#include "ofx_common.h"

// ----------------------------------------------------------------- scan
// 3-kernel exclusive scan.  Block = 256 threads x 8 items = 2048 items.
constexpr int SCAN_T = 256;
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = SCAN_T * SCAN_ITEMS;

__device__ __forceinline__ int wave_incl_scan(int v) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    int t = __shfl_up(v, off, 64);
    if (lane >= off) v += t;
  }
  return v;
}

// block-wide exclusive scan of one int per thread; returns exclusive prefix, total via ref.
__device__ __forceinline__ int block_excl_scan(int v, int& total, int* smem /*>= 4+1*/) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  int inc = wave_incl_scan(v);
  if (lane == 63) smem[wid] = inc;
  __syncthreads();
  int woff = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < SCAN_T / 64; ++w) {
    int s = smem[w];
    if (w < wid) woff += s;
    tot += s;
  }
  total = tot;
  __syncthreads();
  return woff + inc - v;
}

__global__ void __launch_bounds__(SCAN_T) scan_tile_sums(const int32_t* __restrict__ in, int64_t n,
                                                         int32_t* __restrict__ tile_sums) {
  __shared__ int smem[8];
  const int64_t base = (int64_t)blockIdx.x * SCAN_TILE;
  int s = 0;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) {
    int64_t idx = base + (int64_t)i * SCAN_T + threadIdx.x;
    if (idx < n) s += in[idx];
  }
  int total;
  block_excl_scan(s, total, smem);
  if (threadIdx.x == 0) tile_sums[blockIdx.x] = total;
}

// single block: exclusive scan of the tile sums in place (any count, looped).
__global__ void __launch_bounds__(SCAN_T) scan_tile_offsets(int32_t* __restrict__ tile_sums, int64_t ntiles) {
  __shared__ int smem[8];
  int carry = 0;
  for (int64_t base = 0; base < ntiles; base += SCAN_T) {
    int64_t idx = base + threadIdx.x;
    int v = idx < ntiles ? tile_sums[idx] : 0;
    int total;
    int ex = block_excl_scan(v, total, smem);
    if (idx < ntiles) tile_sums[idx] = carry + ex;
    carry += total;
  }
  if (threadIdx.x == 0) tile_sums[ntiles] = carry;
}

__global__ void __launch_bounds__(SCAN_T) scan_apply(const int32_t* __restrict__ in, int64_t n,
                                                     const int32_t* __restrict__ tile_off,
                                                     int32_t* __restrict__ out, int64_t ntiles) {
  __shared__ int smem[8];
  const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
  int v[SCAN_ITEMS];
  int s = 0;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) {
    int64_t idx = base + i;
    v[i] = idx < n ? in[idx] : 0;
    s += v[i];
  }
  int total;
  int ex = block_excl_scan(s, total, smem) + tile_off[blockIdx.x];
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) {
    int64_t idx = base + i;
    if (idx < n) out[idx] = ex;
    ex += v[i];
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) out[n] = tile_off[ntiles];
}

extern "C" size_t ofx_scan_ws_bytes(int64_t n) {
  return (size_t)(ofx_cdiv(n > 0 ? n : 1, SCAN_TILE) + 2) * sizeof(int32_t);
}

extern "C" int ofx_scan_i32(const int32_t* in, int32_t* out, int64_t n, void* ws, void* stream) {
  if (n < 0 || !out || !ws || (n > 0 && !in)) return OFX_EINVAL;
  hipStream_t st = ofx_stream(stream);
  int32_t* tiles = (int32_t*)ws;
  const int64_t ntiles = ofx_cdiv(n > 0 ? n : 1, SCAN_TILE);
  scan_tile_sums<<<(int)ntiles, SCAN_T, 0, st>>>(in, n, tiles);
  scan_tile_offsets<<<1, SCAN_T, 0, st>>>(tiles, ntiles);
  scan_apply<<<(int)ntiles, SCAN_T, 0, st>>>(in, n, tiles, out, ntiles);
  OFX_LAUNCH_CHECK();
  return OFX_OK;
}

// ----------------------------------------------------------------- octree
__global__ void full_layer_kernel(int64_t per_batch, int64_t total, int64_t* __restrict__ keys,
                                  int32_t* __restrict__ children) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = i / per_batch, k = i - b * per_batch;
    keys[i] = k | (b << 48);
    children[i] = (int32_t)i;
  }
}

extern "C" int ofx_octree_full_layer(int depth, int batch_size, int64_t* keys, int32_t* children,
                                     void* stream) {
  if (depth < 0 || depth > 10 || batch_size < 1 || !keys || !children) return OFX_EINVAL;
  const int64_t per = 1ll << (3 * depth), total = per * batch_size;
  full_layer_kernel<<<ofx_grid(total, 256), 256, 0, ofx_stream(stream)>>>(per, total, keys, children);
  OFX_LAUNCH_CHECK();
  return OFX_OK;
}

__global__ void nonzero_flag_kernel(const int32_t* __restrict__ label, int64_t n, int32_t* __restrict__ flag) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    flag[i] = label[i] != 0;
}
__global__ void split_children_kernel(const int32_t* __restrict__ label, const int32_t* __restrict__ scan,
                                      int64_t n, int32_t* __restrict__ children) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    children[i] = label[i] != 0 ? scan[i] : -1;
}

extern "C" int ofx_octree_split(const int32_t* label, int64_t n, int32_t* children, int32_t* scan_out,
                                void* ws, void* stream) {
  if (n < 0 || !children || !scan_out || !ws || (n > 0 && !label)) return OFX_EINVAL;
  hipStream_t st = ofx_stream(stream);
  // children is used as the 0/1 flag buffer first, then overwritten.
  nonzero_flag_kernel<<<ofx_grid(n, 256), 256, 0, st>>>(label, n, children);
  int rc = ofx_scan_i32(children, scan_out, n, ws, stream);
  if (rc) return rc;
  split_children_kernel<<<ofx_grid(n, 256), 256, 0, st>>>(label, scan_out, n, children);
  OFX_LAUNCH_CHECK();
  return OFX_OK;
}

__global__ void grow_kernel(const int64_t* __restrict__ kp, const int32_t* __restrict__ cp, int64_t np,
                            int64_t* __restrict__ kc, int32_t* __restrict__ cc) {
  // one thread per (parent, octant): consecutive threads write consecutive children.
  const int64_t total = np * 8;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t p = t >> 3;
    const int o = (int)(t & 7);
    const int32_t c = cp[p];
    if (c < 0) continue;
    const int64_t key = kp[p];
    const int64_t b = (int64_t)((uint64_t)key >> 48);
    const int64_t k = key & ((1ll << 48) - 1);
    const int64_t dst = (int64_t)c * 8 + o;
    kc[dst] = ((k << 3) | o) | (b << 48);
    cc[dst] = (int32_t)dst;
  }
}

extern "C" int ofx_octree_grow(const int64_t* keys_parent, const int32_t* children_parent, int64_t n_parent,
                               int64_t* keys_child, int32_t* children_child, void* stream) {
  if (n_parent < 0 || !keys_child || !children_child) return OFX_EINVAL;
  if (n_parent == 0) return OFX_OK;
  if (!keys_parent || !children_parent) return OFX_EINVAL;
  grow_kernel<<<ofx_grid(n_parent * 8, 256), 256, 0, ofx_stream(stream)>>>(keys_parent, children_parent,
                                                                           n_parent, keys_child, children_child);
  OFX_LAUNCH_CHECK();
  return OFX_OK;
}

// split_small [B, 8, S, S, S]; full-layer node i = b * S^3 + morton(x,y,z).
__global__ void split_small_label0_kernel(const float* __restrict__ split, int B, int fd,
                                          int32_t* __restrict__ label) {
  const int S = 1 << fd;
  const int64_t per = (int64_t)S * S * S, total = per * B;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = i / per, k = i - b * per;
    int x, y, z, bb;
    ofx_key2xyz(k, x, y, z, bb);
    const float* p = split + (b * 8) * per + ((int64_t)x * S + y) * S + z;
    int any = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) any |= p[j * per] > 0.f;
    label[i] = any;
  }
}
__global__ void split_small_label1_kernel(const float* __restrict__ split, int B, int fd,
                                          const int32_t* __restrict__ children,
                                          int32_t* __restrict__ label1) {
  const int S = 1 << fd;
  const int64_t per = (int64_t)S * S * S, total = per * B * 8;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = t >> 3;
    const int j = (int)(t & 7);
    const int32_t c = children[i];
    if (c < 0) continue;
    const int64_t b = i / per, k = i - b * per;
    int x, y, z, bb;
    ofx_key2xyz(k, x, y, z, bb);
    label1[(int64_t)c * 8 + j] = split[(b * 8 + j) * per + ((int64_t)x * S + y) * S + z] > 0.f;
  }
}

extern "C" int ofx_split_small_label0(const float* split, int batch_size, int full_depth, int32_t* label,
                                      void* stream) {
  if (!split || !label || batch_size < 1 || full_depth < 0 || full_depth > 8) return OFX_EINVAL;
  const int64_t total = (1ll << (3 * full_depth)) * batch_size;
  split_small_label0_kernel<<<ofx_grid(total, 256), 256, 0, ofx_stream(stream)>>>(split, batch_size,
                                                                                  full_depth, label);
  OFX_LAUNCH_CHECK();
  return OFX_OK;
}
extern "C" int ofx_split_small_label1(const float* split, int batch_size, int full_depth,
                                      const int32_t* children, int32_t* label1, void* stream) {
  if (!split || !label1 || !children || batch_size < 1 || full_depth < 0 || full_depth > 8) return OFX_EINVAL;
  const int64_t total = (1ll << (3 * full_depth)) * batch_size * 8;
  split_small_label1_kernel<<<ofx_grid(total, 256), 256, 0, ofx_stream(stream)>>>(split, batch_size,
                                                                                  full_depth, children, label1);
  OFX_LAUNCH_CHECK();
  return OFX_OK;
}

__global__ void split_large_label0_kernel(const float* __restrict__ split, int64_t n, int32_t* __restrict__ label0) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float4 a = *reinterpret_cast<const float4*>(split + i * 8);
    const float4 b = *reinterpret_cast<const float4*>(split + i * 8 + 4);
    label0[i] = (a.x > 0.f) | (a.y > 0.f) | (a.z > 0.f) | (a.w > 0.f) | (b.x > 0.f) | (b.y > 0.f) |
                (b.z > 0.f) | (b.w > 0.f);
  }
}
__global__ void split_large_label1_kernel(const float* __restrict__ split, int64_t n,
                                          const int32_t* __restrict__ children, int32_t* __restrict__ label1) {
  const int64_t total = n * 8;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int32_t c = children[t >> 3];
    if (c < 0) continue;
    label1[(int64_t)c * 8 + (t & 7)] = split[t] > 0.f;
  }
}
extern "C" int ofx_split_large_label0(const float* split, int64_t n, int32_t* label0, void* stream) {
  if (n < 0 || (n > 0 && (!split || !label0))) return OFX_EINVAL;
  if (((uintptr_t)split & 15) != 0) return OFX_EINVAL;
  split_large_label0_kernel<<<ofx_grid(n, 256), 256, 0, ofx_stream(stream)>>>(split, n, label0);
  OFX_LAUNCH_CHECK();
  return OFX_OK;
}
extern "C" int ofx_split_large_label1(const float* split, int64_t n, const int32_t* children, int32_t* label1,
                                      void* stream) {
  if (n < 0 || (n > 0 && (!split || !label1 || !children))) return OFX_EINVAL;
  split_large_label1_kernel<<<ofx_grid(n * 8, 256), 256, 0, ofx_stream(stream)>>>(split, n, children, label1);
  OFX_LAUNCH_CHECK();
  return OFX_OK;
}

// data [B*8^d, C] (row r = b*8^d + morton) <-> vox [B, C, S, S, S].
// Tile of 64 consecutive rows (one Morton block of 4x4x4 voxels) x 64 channels staged in
// LDS so both the row-major side and the channel-major side move in >= 16-float runs.
template <bool TO_VOX>
__global__ void __launch_bounds__(256) o2v_kernel(float* __restrict__ data, int64_t ld, int C, int B, int d,
                                                  float* __restrict__ vox) {
  __shared__ float tile[64][65];
  const int S = 1 << d;
  const int64_t per = (int64_t)S * S * S, rows = per * B;
  const int64_t r0 = (int64_t)blockIdx.x * 64;
  const int c0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  // voxel side: thread tx <-> row r0 + tx
  const int64_t rv = r0 + tx;
  const int64_t bv = rv < rows ? rv / per : 0;
  int x, y, z, bb;
  ofx_key2xyz(rv - bv * per, x, y, z, bb);
  const int64_t voff = ((int64_t)x * S + y) * S + z;
  if (TO_VOX) {
    for (int rr = ty; rr < 64; rr += 4) {
      const int c = c0 + tx;
      tile[rr][tx] = (c < C && r0 + rr < rows) ? data[(r0 + rr) * ld + c] : 0.f;
    }
    __syncthreads();
    for (int cc = ty; cc < 64; cc += 4) {
      const int c = c0 + cc;
      if (c < C && rv < rows) vox[(bv * C + c) * per + voff] = tile[tx][cc];
    }
  } else {
    for (int cc = ty; cc < 64; cc += 4) {
      const int c = c0 + cc;
      tile[tx][cc] = (c < C && rv < rows) ? vox[(bv * C + c) * per + voff] : 0.f;
    }
    __syncthreads();
    for (int rr = ty; rr < 64; rr += 4) {
      const int c = c0 + tx;
      if (c < C && r0 + rr < rows) data[(r0 + rr) * ld + c] = tile[rr][tx];
    }
  }
}

extern "C" int ofx_octree2voxel_cf(const float* data, int64_t ld, int C, int batch_size, int depth, float* vox,
                                   void* stream) {
  if (!data || !vox || C < 1 || batch_size < 1 || depth < 0 || depth > 8 || ld < C) return OFX_EINVAL;
  const int64_t rows = (1ll << (3 * depth)) * batch_size;
  dim3 grid((unsigned)ofx_cdiv(rows, 64), (unsigned)ofx_cdiv(C, 64));
  o2v_kernel<true><<<grid, 256, 0, ofx_stream(stream)>>>(const_cast<float*>(data), ld, C, batch_size, depth, vox);
  OFX_LAUNCH_CHECK();
  return OFX_OK;
}
extern "C" int ofx_voxel2octree_cf(const float* vox, int C, int batch_size, int depth, float* data, int64_t ld,
                                   void* stream) {
  if (!data || !vox || C < 1 || batch_size < 1 || depth < 0 || depth > 8 || ld < C) return OFX_EINVAL;
  const int64_t rows = (1ll << (3 * depth)) * batch_size;
  dim3 grid((unsigned)ofx_cdiv(rows, 64), (unsigned)ofx_cdiv(C, 64));
  o2v_kernel<false><<<grid, 256, 0, ofx_stream(stream)>>>(data, ld, C, batch_size, depth, const_cast<float*>(vox));
  OFX_LAUNCH_CHECK();
  return OFX_OK;
}

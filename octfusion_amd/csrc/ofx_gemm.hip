// libofx: register-staged contraction core (round 1): dense GEMM, the fused dual-octree GraphConv for the layers the
// planes kernel (ofx_gemm2.hip) does not take, and the dense-grid 3x3x3 convolution (same kernel, 27 "directions").
//
// One kernel template per contraction precision, two A-tile loaders:
//   MODE_DENSE  : A[arow(m), k] row-major (optional row map)                 -> ofx_gemm_f32
//   MODE_GATHER : A[m, dir*cin + c] = x[nbr[m, dir], c]  (segment mean when a (row,dir)
//                 has several neighbours), then the dense node-type-fraction slab
//                                                      -> ofx_graphconv_fwd / ofx_gridconv_fwd
// The gathered [N, ndir*cin] "col_data" of the reference (modules.py:208-210) never exists
// in HBM: neighbour rows are fetched (16 B per lane, one 128-B line per 8 lanes) straight
// into the LDS A-tile.
//
// Precisions (ofx_set_precision): 0 (default) = bf16x3: both operands split into bf16 hi + lo, three
// v_mfma_f32_32x32x16_bf16 per product term, fp32 accumulate (gemm_pairs_x3_kernel; ~1e-5 of an fp32 reference);
// 1 = exact fp32: v_mfma_f32_32x32x2_f32, a k-ordered fma chain (gemm_fast_kernel / gemm_kernel; 157 TF peak).
// Tiling (wave = 64): block = 4 waves, BM = 128 rows, BK = 32; BN = 128 (2x2 waves of 64x64), 64 (2x2 of 64x32) or
// 32 (4x1 of 32x32).  Weights are pre-packed once (ofx_pack_weights / ofx_pack_conv3d): fp32 [k/4][n][4] followed by
// the bf16 hi | lo planes [k/8][n][8].
// Small-M problems (dense 4^3 / 8^3 grids) are split along K into up to 64 slices whose
// partial tiles go to a workspace and are summed, in slice order (deterministic), by
// splitk_reduce_kernel, which also applies the epilogue.
#include "ofx_gemm_common.h"

constexpr int BM = 128;
constexpr int BK = 32;
constexpr int A_LD = BK + 4;
constexpr int MODE_DENSE = 0;
constexpr int MODE_GATHER = 1;


// mean over the CSR segment (row, dir) of x[col, cc..cc+3]
__device__ __forceinline__ float4 gather_seg4(const GemmArgs& g, int64_t row, int dir, int cc) {
  const int64_t s = row * g.ndir + dir;
  const int32_t a = g.seg_ptr[s], e = g.seg_ptr[s + 1];
  float4 acc = f4zero();
  if (g.edge_w) {
    for (int32_t p = a; p < e; ++p) {
      const float w = g.edge_w[p];
      const float4 v = *reinterpret_cast<const float4*>(g.x + (int64_t)g.col[p] * g.ldx + cc);
      acc.x += w * v.x; acc.y += w * v.y; acc.z += w * v.z; acc.w += w * v.w;
    }
    return acc;
  }
  for (int32_t p = a; p < e; ++p) f4add(acc, *reinterpret_cast<const float4*>(g.x + (int64_t)g.col[p] * g.ldx + cc));
  if (e - a > 1) {
    const float inv = 1.f / (float)(e - a);
    acc.x *= inv; acc.y *= inv; acc.z *= inv; acc.w *= inv;
  }
  return acc;
}

// scalar (slow-path) gathered element
__device__ __forceinline__ float gather_elem(const GemmArgs& g, int64_t row, int64_t k) {
  if (k >= (int64_t)g.ndir * g.cin) return 0.f;
  const int dir = (int)(k / g.cin), c = (int)(k - (int64_t)dir * g.cin);
  const int32_t nb = g.nbr[row * g.ndir + dir];
  if (nb >= 0) return g.x[(int64_t)nb * g.ldx + c];
  if (nb == -1) return 0.f;
  const int64_t s = row * g.ndir + dir;
  const int32_t a = g.seg_ptr[s], e = g.seg_ptr[s + 1];
  float acc = 0.f;
  if (g.edge_w) {
    for (int32_t p = a; p < e; ++p) acc += g.edge_w[p] * g.x[(int64_t)g.col[p] * g.ldx + c];
    return acc;
  }
  for (int32_t p = a; p < e; ++p) acc += g.x[(int64_t)g.col[p] * g.ldx + c];
  if (e - a > 1) acc /= (float)(e - a);
  return acc;
}

__device__ __forceinline__ void load_a_dense(const GemmArgs& g, int64_t m0, int64_t k0, float4 (&va)[4]) {
  const int c4 = threadIdx.x & 7, r0 = threadIdx.x >> 3;
  const int64_t k = k0 + c4 * 4;
  const bool vec = ((g.lda & 3) == 0) && (k + 3 < g.K) && ((((uintptr_t)g.A) & 15) == 0);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int64_t m = m0 + r0 + 32 * i;
    float4 v = f4zero();
    if (m < g.M) {
      const int64_t ar = g.a_rows ? (int64_t)g.a_rows[m] : m;
      const float* p = g.A + ar * g.lda + k;
      if (vec) {
        v = *reinterpret_cast<const float4*>(p);
      } else {
        if (k + 0 < g.K) v.x = p[0];
        if (k + 1 < g.K) v.y = p[1];
        if (k + 2 < g.K) v.z = p[2];
        if (k + 3 < g.K) v.w = p[3];
      }
    }
    va[i] = v;
  }
}

template <int BN>
__device__ __forceinline__ void load_b_tile(const GemmArgs& g, int64_t n0, int64_t k0, float4 (&vb)[BN / 32]) {
  // tile = 8 k-quads x BN columns of float4
#pragma unroll
  for (int i = 0; i < BN / 32; ++i) {
    const int idx = threadIdx.x + 256 * i;
    const int kql = idx / BN, n = idx % BN;
    const int64_t nn = n0 + n;
    vb[i] = (nn < g.N) ? *reinterpret_cast<const float4*>(g.Wp + (((k0 >> 2) + kql) * g.N + nn) * 4) : f4zero();
  }
}



template <int MODE, int WM, int WN, int MI, int NI>
__global__ void __launch_bounds__(256, 2) gemm_kernel(const GemmArgs g) {
  constexpr int BN = WN * NI * 32;
  static_assert(WM * MI * 32 == BM, "BM");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                       // [2][BM * A_LD]
  float* Bs = smem + 2 * BM * A_LD;       // [2][8 * BN * 4]

  // XCD-aware tile order: consecutive tiles (which share gathered neighbour rows through
  // Morton locality) stay on one XCD / one L2.  Bijective for any grid size.
  const int ntile = g.ntm * g.ntn;
  const int nblk = ntile * g.nsplit;
  int bid = blockIdx.x;
  {
    const int q = nblk / 8, r = nblk % 8, xcd = bid % 8, j = bid / 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
  }
  const int split = bid / ntile;
  bid -= split * ntile;
  const int tm = bid / g.ntn, tn = bid - tm * g.ntn;
  const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;

  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int wm = wid / WN, wn = wid % WN;
  const int l31 = lane & 31, h = lane >> 5;
  const int c4 = threadIdx.x & 7, r0 = threadIdx.x >> 3;

  f32x16 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float4 va[4];
  float4 vb[BN / 32];
  const int nkt_all = (int)(g.Kp / BK);
  const int kt_begin = split * g.kt_per_split;
  const int kt_end = (kt_begin + g.kt_per_split < nkt_all) ? kt_begin + g.kt_per_split : nkt_all;

  // gather state: neighbour ids of this thread's 4 rows for the current direction and the next one
  int32_t nb[4] = {-1, -1, -1, -1}, nbn[4] = {-1, -1, -1, -1};
  int dir_cur = -1, dir_next = -1;
  auto load_nbr = [&](int dir, int32_t (&dst)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int64_t m = m0 + r0 + 32 * i;
      dst[i] = (m < g.M && dir < g.ndir) ? g.nbr[m * g.ndir + dir] : -1;
    }
  };

  auto load_a = [&](int kt) {
    const int64_t k0 = (int64_t)kt * BK;
    if (MODE == MODE_DENSE) {
      load_a_dense(g, m0, k0, va);
    } else if (k0 >= g.Kf) {                               // node-type fraction slab (dense, zero padded)
      const int64_t kk = k0 - g.Kf + c4 * 4;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int64_t m = m0 + r0 + 32 * i;
        va[i] = (m < g.M) ? *reinterpret_cast<const float4*>(g.tf + m * g.ldt + kk) : f4zero();
      }
    } else if (g.fast) {
      const int dir = (int)(k0 / g.cin);
      const int cc = (int)(k0 - (int64_t)dir * g.cin) + c4 * 4;
      if (dir != dir_cur) {
        if (dir == dir_next) {
#pragma unroll
          for (int i = 0; i < 4; ++i) nb[i] = nbn[i];
        } else {
          load_nbr(dir, nb);
        }
        dir_cur = dir;
        dir_next = dir + 1;
        load_nbr(dir_next, nbn);                             // prefetch: consumed >= cin/32 k-tiles later
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
        va[i] = nb[i] >= 0 ? *reinterpret_cast<const float4*>(g.x + (int64_t)nb[i] * g.ldx + cc) : f4zero();
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (nb[i] == -2) va[i] = gather_seg4(g, m0 + r0 + 32 * i, dir, cc);   // several neighbours (rare)
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int64_t m = m0 + r0 + 32 * i;
        float4 v = f4zero();
        if (m < g.M) {
          const int64_t k = k0 + c4 * 4;
          v.x = gather_elem(g, m, k);
          v.y = gather_elem(g, m, k + 1);
          v.z = gather_elem(g, m, k + 2);
          v.w = gather_elem(g, m, k + 3);
        }
        va[i] = v;
      }
    }
  };

  auto store_tiles = [&](int buf) {
    float* a = As + buf * BM * A_LD;
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<float4*>(a + (r0 + 32 * i) * A_LD + c4 * 4) = va[i];
    float* b = Bs + buf * 8 * BN * 4;
#pragma unroll
    for (int i = 0; i < BN / 32; ++i) *reinterpret_cast<float4*>(b + (threadIdx.x + 256 * i) * 4) = vb[i];
  };

  if (kt_begin < kt_end) {
    load_a(kt_begin);
    load_b_tile<BN>(g, n0, (int64_t)kt_begin * BK, vb);
    store_tiles(0);
  }
  __syncthreads();

  for (int kt = kt_begin; kt < kt_end; ++kt) {
    const int buf = (kt - kt_begin) & 1;
    if (kt + 1 < kt_end) {
      load_a(kt + 1);
      load_b_tile<BN>(g, n0, (int64_t)(kt + 1) * BK, vb);
    }
    const float* a = As + buf * BM * A_LD + (wm * MI * 32 + l31) * A_LD + h * 16;
    const float* b = Bs + buf * 8 * BN * 4 + ((h * 4) * BN + wn * NI * 32 + l31) * 4;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float4 fa[MI], fb[NI];
#pragma unroll
      for (int i = 0; i < MI; ++i) fa[i] = *reinterpret_cast<const float4*>(a + i * 32 * A_LD + q * 4);
#pragma unroll
      for (int j = 0; j < NI; ++j) fb[j] = *reinterpret_cast<const float4*>(b + (q * BN + j * 32) * 4);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].x, fb[j].x, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].y, fb[j].y, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].z, fb[j].z, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].w, fb[j].w, acc[i][j], 0, 0, 0);
        }
    }
    if (kt + 1 < kt_end) store_tiles(buf ^ 1);
    __syncthreads();
  }

  epilogue_store<WM, WN, MI, NI>(g, acc, m0, n0, wm, wn, l31, h, split);
}


// ---------------------------------------------------------------------------------
// Fast gather kernel: cin % 32 == 0, 16-B aligned rows, extended neighbour table.
// The loader is branch-free so the compiler keeps the gathers in flight across the MFMA
// block (the generic kernel's control flow forces s_waitcnt vmcnt(0) before the MFMAs):
//   * nbr_ext[m, dir] always names a source row: < n_src -> x, n_src -> the zero row of
//     `aux` (no neighbour / zero padding), > n_src -> a pre-averaged row of `aux`
//     (segment with several neighbours, written by multi_mean_kernel);
//   * indices for tile t are loaded two iterations ahead (L1 hits: they only change when
//     the direction changes), the rows for tile t one iteration ahead;
//   * tile rows / columns past M / N are clamped, never branched on (their accumulators
//     are simply not stored);
//   * the node-type slab is the same code with (base, row, pitch) = (tf, m, ldt).
// MFMA operand fragments are double-buffered in registers across the 4 k-quads.
// pin a wave-uniform 64-bit value in SGPRs and make it opaque (stops hipcc from re-deriving it as a
// per-lane load from the kernarg segment, which it does for `cond ? g.a : g.b` on by-value structs)
__device__ __forceinline__ uint64_t sgpr64(uint64_t v) {
  unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  asm volatile("" : "+s"(lo), "+s"(hi));
  return ((uint64_t)hi << 32) | lo;
}

// global-address-space views: pointers rebuilt from integers are "generic" to hipcc, which then emits
// flat_load (counted on BOTH vmcnt and lgkmcnt -> every LDS wait would also wait for the gathers).
typedef const float __attribute__((address_space(1)))* gfp;
typedef const int32_t __attribute__((address_space(1)))* gip;
__device__ __forceinline__ float4 ldg4(gfp p) {
  typedef float v4f __attribute__((ext_vector_type(4)));
  const v4f v = *reinterpret_cast<const v4f __attribute__((address_space(1)))*>(p);
  return make_float4(v.x, v.y, v.z, v.w);
}

template <int WM, int WN, int MI, int NI>
__device__ __forceinline__ void mfma_tile(const float* __restrict__ a, const float* __restrict__ b, f32x16 (&acc)[MI][NI]) {
  constexpr int BN = WN * NI * 32;
  float4 fa[2][MI], fb[2][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i) fa[0][i] = *reinterpret_cast<const float4*>(a + i * 32 * A_LD);
#pragma unroll
  for (int j = 0; j < NI; ++j) fb[0][j] = *reinterpret_cast<const float4*>(b + (j * 32) * 4);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int cur = q & 1, nxt = cur ^ 1;
    if (q < 3) {
#pragma unroll
      for (int i = 0; i < MI; ++i) fa[nxt][i] = *reinterpret_cast<const float4*>(a + i * 32 * A_LD + (q + 1) * 4);
#pragma unroll
      for (int j = 0; j < NI; ++j) fb[nxt][j] = *reinterpret_cast<const float4*>(b + ((q + 1) * BN + j * 32) * 4);
    }
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][i].x, fb[cur][j].x, acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][i].y, fb[cur][j].y, acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][i].z, fb[cur][j].z, acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][i].w, fb[cur][j].w, acc[i][j], 0, 0, 0);
      }
  }
}

// element offset (from x) of source row `id`: rows >= n_src live in aux (same row pitch ldx);
// pure ALU (no select on pointers) so hipcc cannot turn it into a branch or a kernarg re-load.
__device__ __forceinline__ int64_t src_off(uint32_t id, int64_t ldx, int64_t n_src, int64_t aux_delta) {
  // ids are non-negative: zero-extension costs no instruction that depends on the loaded value (a sign
  // extension would be hoisted next to the load and force an early s_waitcnt on the index prefetch)
  const int64_t i = (int64_t)(uint64_t)id;
  const int64_t mask = (n_src - 1 - i) >> 63;           // all ones iff id >= n_src
  return i * ldx + (aux_delta & mask);
}

template <int MODE, int WM, int WN, int MI, int NI>
__global__ void __launch_bounds__(256, 2) gemm_fast_kernel(const GemmArgs g) {
  constexpr int BN = WN * NI * 32;
  constexpr int NB = BN / 32;
  static_assert(WM * MI * 32 == BM, "BM");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;
  float* Bs = smem + 2 * BM * A_LD;

  const int ntile = g.ntm * g.ntn;
  const int nblk = ntile * g.nsplit;
  int bid = blockIdx.x;
  {
    const int q = nblk / 8, r = nblk % 8, xcd = bid % 8, j = bid / 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
  }
  const int split = bid / ntile;
  bid -= split * ntile;
  const int tm = bid / g.ntn, tn = bid - tm * g.ntn;
  const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;

  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int wm = wid / WN, wn = wid % WN;
  const int l31 = lane & 31, h = lane >> 5;
  const int c4 = threadIdx.x & 7, r0 = threadIdx.x >> 3;

  // wave-uniform operands, pinned in SGPRs
  const gfp xp = (gfp)sgpr64((uint64_t)(MODE == MODE_DENSE ? g.A : g.x));
  const gfp tfp = (gfp)sgpr64((uint64_t)g.tf);
  const gfp wp = (gfp)sgpr64((uint64_t)g.Wp);
  const gip tab = (gip)sgpr64((uint64_t)g.nbr_ext);
  const int64_t ldx = (int64_t)sgpr64((uint64_t)(MODE == MODE_DENSE ? g.lda : g.ldx));
  const int64_t ldt = (int64_t)sgpr64((uint64_t)g.ldt), n_src = (int64_t)sgpr64((uint64_t)g.n_src);
  const int64_t Ncols = (int64_t)sgpr64((uint64_t)g.N);
  const int64_t aux_delta = (int64_t)sgpr64((uint64_t)((g.aux - g.x) - g.n_src * g.ldx));
  const int ndir = g.ndir;

  f32x16 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nkt_all = (int)(g.Kp / BK);
  const int kt_begin = split * g.kt_per_split;
  const int kt_end = (kt_begin + g.kt_per_split < nkt_all) ? kt_begin + g.kt_per_split : nkt_all;
  const int tpd = MODE == MODE_DENSE ? nkt_all : g.cin / BK;      // k-tiles per direction
  const int nkt_g = MODE == MODE_DENSE ? nkt_all : ndir * tpd;     // gather tiles; tiles beyond are the type slab
  const int g_begin = kt_begin < nkt_g ? kt_begin : nkt_g;
  const int g_end = kt_end < nkt_g ? kt_end : nkt_g;
  const int t_begin = kt_begin > nkt_g ? kt_begin : nkt_g;

  // per-thread constants (rows / columns clamped: tiles past M / N are computed but never stored)
  int64_t m_0 = m0 + r0, m_1 = m_0 + 32, m_2 = m_0 + 64, m_3 = m_0 + 96;
  const int64_t mmax = g.M - 1;
  m_0 = m_0 < mmax ? m_0 : mmax; m_1 = m_1 < mmax ? m_1 : mmax;
  m_2 = m_2 < mmax ? m_2 : mmax; m_3 = m_3 < mmax ? m_3 : mmax;
  const gip t0 = tab + m_0 * ndir;
  const gip t1 = tab + m_1 * ndir;
  const gip t2 = tab + m_2 * ndir;
  const gip t3 = tab + m_3 * ndir;
  // dense mode: fixed source rows (optional row map), K tail clamped (weights are zero-padded there)
  int64_t dr0 = m_0, dr1 = m_1, dr2 = m_2, dr3 = m_3;
  if (MODE == MODE_DENSE && g.a_rows) { dr0 = g.a_rows[m_0]; dr1 = g.a_rows[m_1]; dr2 = g.a_rows[m_2]; dr3 = g.a_rows[m_3]; }
  const gfp dp0 = xp + dr0 * ldx, dp1 = xp + dr1 * ldx, dp2 = xp + dr2 * ldx, dp3 = xp + dr3 * ldx;
  const int kclamp = (int)g.K - 4;
  int64_t bo0 = 0, bo1 = 0, bo2 = 0, bo3 = 0;
  {
    auto boff = [&](int i) {
      const int idx = threadIdx.x + 256 * i;
      int64_t nn = n0 + idx % BN;
      nn = nn < Ncols ? nn : Ncols - 1;
      return ((int64_t)(idx / BN) * Ncols + nn) * 4;
    };
    bo0 = boff(0);
    if (NB > 1) bo1 = boff(1);
    if (NB > 2) { bo2 = boff(2); bo3 = boff(3); }
  }
  float* const a_st = As + r0 * A_LD + c4 * 4;
  float* const b_st = Bs + threadIdx.x * 4;
  const float* const a_ld = As + (wm * MI * 32 + l31) * A_LD + h * 16;
  const float* const b_ld = Bs + ((h * 4) * BN + wn * NI * 32 + l31) * 4;

  float4 va0, va1, va2, va3, vb0, vb1, vb2, vb3;
  vb1 = vb2 = vb3 = f4zero();

  // ------------------------------------------------------------ gather tiles
  if (g_begin < g_end) {
    int32_t ia0, ia1, ia2, ia3, ib0, ib1, ib2, ib3;
    {
      // iteration order: channel chunk outer, direction inner (see gemm_pairs_x3_kernel)
      const int d0 = g_begin % ndir;
      const int d1 = (g_begin + 1 < nkt_g ? g_begin + 1 : nkt_g - 1) % ndir;
      if (MODE == MODE_DENSE) {
        ia0 = ia1 = ia2 = ia3 = ib0 = ib1 = ib2 = ib3 = 0;
        int cc = g_begin * BK + c4 * 4;
        cc = cc < kclamp ? cc : kclamp;
        va0 = ldg4(dp0 + cc); va1 = ldg4(dp1 + cc); va2 = ldg4(dp2 + cc); va3 = ldg4(dp3 + cc);
      } else {
        ia0 = t0[d0]; ia1 = t1[d0]; ia2 = t2[d0]; ia3 = t3[d0];
        ib0 = t0[d1]; ib1 = t1[d1]; ib2 = t2[d1]; ib3 = t3[d1];
        const int cc = (g_begin / ndir) * BK + c4 * 4;
        va0 = ldg4(xp + src_off(ia0, ldx, n_src, aux_delta) + cc);
        va1 = ldg4(xp + src_off(ia1, ldx, n_src, aux_delta) + cc);
        va2 = ldg4(xp + src_off(ia2, ldx, n_src, aux_delta) + cc);
        va3 = ldg4(xp + src_off(ia3, ldx, n_src, aux_delta) + cc);
      }
      const int ktw0 = MODE == MODE_DENSE ? g_begin : d0 * tpd + g_begin / ndir;
      const gfp wk = wp + (int64_t)ktw0 * 8 * Ncols * 4;
      vb0 = ldg4(wk + bo0);
      if (NB > 1) vb1 = ldg4(wk + bo1);
      if (NB > 2) { vb2 = ldg4(wk + bo2); vb3 = ldg4(wk + bo3); }
    }
    *reinterpret_cast<float4*>(a_st) = va0;
    *reinterpret_cast<float4*>(a_st + 32 * A_LD) = va1;
    *reinterpret_cast<float4*>(a_st + 64 * A_LD) = va2;
    *reinterpret_cast<float4*>(a_st + 96 * A_LD) = va3;
    *reinterpret_cast<float4*>(b_st) = vb0;
    if (NB > 1) *reinterpret_cast<float4*>(b_st + 1024) = vb1;
    if (NB > 2) { *reinterpret_cast<float4*>(b_st + 2048) = vb2; *reinterpret_cast<float4*>(b_st + 3072) = vb3; }
    ia0 = ib0; ia1 = ib1; ia2 = ib2; ia3 = ib3;
    __syncthreads();

    for (int kt = g_begin; kt < g_end; ++kt) {
      const int buf = (kt - g_begin) & 1;
      // prefetch: indices of tile kt+2, rows + weights of tile kt+1 (clamped at the last gather tile)
      const int ktn = kt + 1 < nkt_g ? kt + 1 : nkt_g - 1;
      const int dn = ktn % ndir;
      const int d2 = (kt + 2 < nkt_g ? kt + 2 : nkt_g - 1) % ndir;
      if (MODE == MODE_DENSE) {
        int cc = ktn * BK + c4 * 4;
        cc = cc < kclamp ? cc : kclamp;
        va0 = ldg4(dp0 + cc); va1 = ldg4(dp1 + cc); va2 = ldg4(dp2 + cc); va3 = ldg4(dp3 + cc);
      } else {
        ib0 = t0[d2]; ib1 = t1[d2]; ib2 = t2[d2]; ib3 = t3[d2];
        const int cc = (ktn / ndir) * BK + c4 * 4;
        va0 = ldg4(xp + src_off(ia0, ldx, n_src, aux_delta) + cc);
        va1 = ldg4(xp + src_off(ia1, ldx, n_src, aux_delta) + cc);
        va2 = ldg4(xp + src_off(ia2, ldx, n_src, aux_delta) + cc);
        va3 = ldg4(xp + src_off(ia3, ldx, n_src, aux_delta) + cc);
      }
      const int ktw = MODE == MODE_DENSE ? ktn : dn * tpd + ktn / ndir;
      const gfp wk = wp + (int64_t)ktw * 8 * Ncols * 4;
      vb0 = ldg4(wk + bo0);
      if (NB > 1) vb1 = ldg4(wk + bo1);
      if (NB > 2) { vb2 = ldg4(wk + bo2); vb3 = ldg4(wk + bo3); }
      // pin the issue order: hipcc otherwise sinks these loads below the MFMA block (shorter live
      // ranges), which exposes the whole gather latency every k-tile.
      __builtin_amdgcn_sched_barrier(0);
      mfma_tile<WM, WN, MI, NI>(a_ld + buf * BM * A_LD, b_ld + buf * 8 * BN * 4, acc);
      __builtin_amdgcn_sched_barrier(0);
      float* as = a_st + (buf ^ 1) * BM * A_LD;
      float* bs = b_st + (buf ^ 1) * 8 * BN * 4;
      *reinterpret_cast<float4*>(as) = va0;
      *reinterpret_cast<float4*>(as + 32 * A_LD) = va1;
      *reinterpret_cast<float4*>(as + 64 * A_LD) = va2;
      *reinterpret_cast<float4*>(as + 96 * A_LD) = va3;
      *reinterpret_cast<float4*>(bs) = vb0;
      if (NB > 1) *reinterpret_cast<float4*>(bs + 1024) = vb1;
      if (NB > 2) { *reinterpret_cast<float4*>(bs + 2048) = vb2; *reinterpret_cast<float4*>(bs + 3072) = vb3; }
      ia0 = ib0; ia1 = ib1; ia2 = ib2; ia3 = ib3;
      __syncthreads();
    }
  }

  // ------------------------------------------------------------ node-type slab tiles (dense rows of tf)
  for (int kt = t_begin; kt < kt_end; ++kt) {
    const int cc = (kt - nkt_g) * BK + c4 * 4;
    va0 = ldg4(tfp + m_0 * ldt + cc);
    va1 = ldg4(tfp + m_1 * ldt + cc);
    va2 = ldg4(tfp + m_2 * ldt + cc);
    va3 = ldg4(tfp + m_3 * ldt + cc);
    const gfp wk = wp + (int64_t)kt * 8 * Ncols * 4;
    vb0 = ldg4(wk + bo0);
    if (NB > 1) vb1 = ldg4(wk + bo1);
    if (NB > 2) { vb2 = ldg4(wk + bo2); vb3 = ldg4(wk + bo3); }
    *reinterpret_cast<float4*>(a_st) = va0;
    *reinterpret_cast<float4*>(a_st + 32 * A_LD) = va1;
    *reinterpret_cast<float4*>(a_st + 64 * A_LD) = va2;
    *reinterpret_cast<float4*>(a_st + 96 * A_LD) = va3;
    *reinterpret_cast<float4*>(b_st) = vb0;
    if (NB > 1) *reinterpret_cast<float4*>(b_st + 1024) = vb1;
    if (NB > 2) { *reinterpret_cast<float4*>(b_st + 2048) = vb2; *reinterpret_cast<float4*>(b_st + 3072) = vb3; }
    __syncthreads();
    mfma_tile<WM, WN, MI, NI>(a_ld, b_ld, acc);
    __syncthreads();
  }

  epilogue_store<WM, WN, MI, NI>(g, acc, m0, n0, wm, wn, l31, h, split);
}


// ---------------------------------------------------------------------------------
// bf16x3 contraction: fp32-class accuracy on the bf16 matrix pipe (16x the fp32 MFMA rate).
//   a = a_hi + a_lo, w = w_hi + w_lo (bf16 pairs, round-to-nearest-even);
//   a*w ~= a_hi*w_hi + a_lo*w_hi + a_hi*w_lo   (dropped a_lo*w_lo <= 2^-18 |a w|),
// every product exact in the fp32 accumulator of v_mfma_f32_32x32x16_bf16: per-product relative
// error <= ~2^-16 (1.5e-5), i.e. the result agrees with an fp32 reference to ~1e-5 -- inside the
// 1e-3 parity bar with two orders of magnitude to spare, at 16/3 = 5.3x the fp32 MFMA rate.
// Same pipeline as gemm_fast_kernel (branch-free gather, indices two tiles ahead, rows one tile
// ahead, pinned issue order); activations are split when they are written to LDS, weights are
// pre-split once by the pack kernels.  LDS per buffer: A_hi/A_lo [128][32+8] bf16 (80-B rows:
// conflict-free b128 reads), B_hi/B_lo [4][BN][8] bf16.
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8h_t __attribute__((ext_vector_type(8)));
typedef unsigned u32x4h __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ void split_bf16x4(const float4& v, uint2& hi, uint2& lo) {
  hi.x = cvt_pk_bf16(v.x, v.y);
  hi.y = cvt_pk_bf16(v.z, v.w);
  const float hx = __uint_as_float(hi.x << 16), hy = __uint_as_float(hi.x & 0xffff0000u);
  const float hz = __uint_as_float(hi.y << 16), hw = __uint_as_float(hi.y & 0xffff0000u);
  lo.x = cvt_pk_bf16(v.x - hx, v.y - hy);
  lo.y = cvt_pk_bf16(v.z - hz, v.w - hw);
}
// H16 = 0: bf16 pairs ("bf16x3"); 1: fp16 pairs ("fp16x3": 11 + 11 significand bits, same three MFMAs per product --
// v_mfma_f32_32x32x16_f16 honours fp16 denormals, so the lo part stays exact down to 2^-24; values beyond the fp16
// range turn into Inf / NaN, see split16x4).  See ofx_planes.h for the formats.
__device__ __forceinline__ unsigned cvt_pk_f16(float a, float b) {
  const _Float16 x = (_Float16)a, y = (_Float16)b;
  return (unsigned)__builtin_bit_cast(unsigned short, x) | ((unsigned)__builtin_bit_cast(unsigned short, y) << 16);
}
__device__ __forceinline__ float f16_lo_f32(unsigned p) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(p & 0xffffu)); }
__device__ __forceinline__ float f16_hi_f32(unsigned p) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(p >> 16)); }
__device__ __forceinline__ float sat_f16(float v) { return fminf(fmaxf(v, -65504.f), 65504.f); }
template <int H16>
__device__ __forceinline__ void split16x4(const float4& v, uint2& hi, uint2& lo) {
  if constexpr (H16 == 0) {
    split_bf16x4(v, hi, lo);
  } else {
    // NO saturation: an operand beyond +-65504 becomes hi = +-Inf, lo = -+Inf, and the products poison the output
    // with NaN -- a loud failure the caller's finiteness check turns into a bf16x3 retry (include/ofx.h, range guard),
    // instead of a silently clamped, plausible-looking result
    const float x = v.x, y = v.y, z = v.z, w = v.w;
    hi.x = cvt_pk_f16(x, y);
    hi.y = cvt_pk_f16(z, w);
    lo.x = cvt_pk_f16(x - f16_lo_f32(hi.x), y - f16_hi_f32(hi.x));
    lo.y = cvt_pk_f16(z - f16_lo_f32(hi.y), w - f16_hi_f32(hi.y));
  }
}
template <int H16>
__device__ __forceinline__ f32x16 mfma16(u32x4h a, u32x4h b, f32x16 c) {
  if constexpr (H16 == 0)
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8h_t, a), __builtin_bit_cast(f16x8h_t, b), c, 0, 0, 0);
}

// (named gemm_bf16x3_kernel until round 4; it has run fp16 pairs -- H16 = 1, the default precision -- since round 3)
template <int MODE, int WM, int WN, int MI, int NI, int H16>
__global__ void __launch_bounds__(256, 2) gemm_pairs_x3_kernel(const GemmArgs g) {
  // LDS carries only the A tile (hi and lo planes, [128][32+8] bf16 each, double buffered = 40 KB).
  // The weight fragments go global(L2) -> registers directly in MFMA operand layout (the packed
  // [k/8][n][8] planes give every lane one contiguous 16-B read).  Staging the weight tile through LDS
  // instead (half the weight loads, twice the LDS fragment reads) was built and measured twice -- before
  // and after the interleaved pipeline -- and is 0-4 % slower.
  constexpr int BN = WN * NI * 32;
  constexpr int A_BYTES = BM * 80;                      // one A plane (hi or lo)
  constexpr int BUF_BYTES = 2 * A_BYTES;
  static_assert(WM * MI * 32 == BM, "BM");
  extern __shared__ __attribute__((aligned(16))) char smem8[];
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  typedef const u32x4 __attribute__((address_space(1)))* gqp;

  const int ntile = g.ntm * g.ntn;
  const int nblk = ntile * g.nsplit;
  int bid = blockIdx.x;
  {
    const int q = nblk / 8, r = nblk % 8, xcd = bid % 8, j = bid / 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
  }
  const int split = bid / ntile;
  bid -= split * ntile;
  const int tm = bid / g.ntn, tn = bid - tm * g.ntn;
  const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;

  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int wm = wid / WN, wn = wid % WN;
  const int l31 = lane & 31, h = lane >> 5;
  // A-tile loader mapping: 8 lanes per 128-B row chunk; the two rows of a 16-lane group are 4 apart so their
  // 64-B LDS writes (80-B row pitch) land on disjoint halves of the 32 banks (consecutive rows overlap by 4
  // banks: SQ_LDS_BANK_CONFLICT was 33 % of the LDS-active cycles with r0 = tid >> 3)
  const int c4 = threadIdx.x & 7;
  const int r0 = ((threadIdx.x >> 4) & 3) + 8 * (threadIdx.x >> 6) + 4 * ((threadIdx.x >> 3) & 1);

  const gfp xp = (gfp)sgpr64((uint64_t)(MODE == MODE_DENSE ? g.A : g.x));
  const gfp tfp = (gfp)sgpr64((uint64_t)g.tf);
  const gqp w16 = (gqp)sgpr64((uint64_t)g.W16);
  const gip tab = (gip)sgpr64((uint64_t)g.nbr_ext);
  const int64_t ldx = (int64_t)sgpr64((uint64_t)(MODE == MODE_DENSE ? g.lda : g.ldx));
  const int64_t ldt = (int64_t)sgpr64((uint64_t)g.ldt), n_src = (int64_t)sgpr64((uint64_t)g.n_src);
  const int64_t Ncols = (int64_t)sgpr64((uint64_t)g.N);
  const int64_t lo_off = (int64_t)sgpr64((uint64_t)(g.Kp / 8 * g.N));     // u32x4 units: hi plane -> lo plane
  const int64_t aux_delta = (int64_t)sgpr64((uint64_t)((g.aux - g.x) - g.n_src * g.ldx));
  const int ndir = g.ndir;

  f32x16 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nkt_all = (int)(g.Kp / BK);
  const int kt_begin = split * g.kt_per_split;
  const int kt_end = (kt_begin + g.kt_per_split < nkt_all) ? kt_begin + g.kt_per_split : nkt_all;
  const int tpd = MODE == MODE_DENSE ? nkt_all : g.cin / BK;
  const int nkt_g = MODE == MODE_DENSE ? nkt_all : ndir * tpd;
  const int g_begin = kt_begin < nkt_g ? kt_begin : nkt_g;
  const int g_end = kt_end < nkt_g ? kt_end : nkt_g;
  const int t_begin = kt_begin > nkt_g ? kt_begin : nkt_g;

  int64_t m_0 = m0 + r0, m_1 = m_0 + 32, m_2 = m_0 + 64, m_3 = m_0 + 96;
  const int64_t mmax = g.M - 1;
  m_0 = m_0 < mmax ? m_0 : mmax; m_1 = m_1 < mmax ? m_1 : mmax;
  m_2 = m_2 < mmax ? m_2 : mmax; m_3 = m_3 < mmax ? m_3 : mmax;
  const gip t0 = tab + m_0 * ndir;
  const gip t1 = tab + m_1 * ndir;
  const gip t2 = tab + m_2 * ndir;
  const gip t3 = tab + m_3 * ndir;
  int64_t dr0 = m_0, dr1 = m_1, dr2 = m_2, dr3 = m_3;
  if (MODE == MODE_DENSE && g.a_rows) { dr0 = g.a_rows[m_0]; dr1 = g.a_rows[m_1]; dr2 = g.a_rows[m_2]; dr3 = g.a_rows[m_3]; }
  const gfp dp0 = xp + dr0 * ldx, dp1 = xp + dr1 * ldx, dp2 = xp + dr2 * ldx, dp3 = xp + dr3 * ldx;
  const int kclamp = (int)g.K - 4;

  // this lane's weight-fragment columns (clamped) -- B operand: lane (j = l31, h) holds W[k = 8(2c+h)..+7][n]
  int64_t bcol[NI];
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    int64_t nn = n0 + (wn * NI + j) * 32 + l31;
    bcol[j] = (nn < Ncols ? nn : Ncols - 1) + (int64_t)h * Ncols;       // + kgroup h
  }
  char* const a_st = smem8 + (r0 * 80 + c4 * 8);
  const char* const a_ld = smem8 + ((wm * MI * 32 + l31) * 80 + 16 * h);

  struct BFrag { u32x4 h[2][NI], l[2][NI]; };          // [chunk][j]
  struct RegSet { float4 a0, a1, a2, a3; };
  struct IdxSet { uint32_t i0, i1, i2, i3; };

  auto load_bfrag = [&](int ktw, BFrag& F) {            // ktw = packed k-tile index (direction-major)
    const gqp wk = w16 + (int64_t)ktw * 4 * Ncols;
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        F.h[c][j] = wk[(int64_t)(2 * c) * Ncols + bcol[j]];
        F.l[c][j] = wk[lo_off + (int64_t)(2 * c) * Ncols + bcol[j]];
      }
  };
  auto store_a = [&](int buf, const RegSet& R) {
    char* a = a_st + buf * BUF_BYTES;
    uint2 hi, lo;
    split16x4<H16>(R.a0, hi, lo);
    *reinterpret_cast<uint2*>(a) = hi; *reinterpret_cast<uint2*>(a + A_BYTES) = lo;
    split16x4<H16>(R.a1, hi, lo);
    *reinterpret_cast<uint2*>(a + 32 * 80) = hi; *reinterpret_cast<uint2*>(a + 32 * 80 + A_BYTES) = lo;
    split16x4<H16>(R.a2, hi, lo);
    *reinterpret_cast<uint2*>(a + 64 * 80) = hi; *reinterpret_cast<uint2*>(a + 64 * 80 + A_BYTES) = lo;
    split16x4<H16>(R.a3, hi, lo);
    *reinterpret_cast<uint2*>(a + 96 * 80) = hi; *reinterpret_cast<uint2*>(a + 96 * 80 + A_BYTES) = lo;
  };
  auto compute = [&](int buf, const BFrag& F) {
    const char* a = a_ld + buf * BUF_BYTES;
    u32x4h ah[2][MI], al[2][MI];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        ah[c][i] = *reinterpret_cast<const u32x4h*>(a + i * 32 * 80 + 32 * c);
        al[c][i] = *reinterpret_cast<const u32x4h*>(a + i * 32 * 80 + 32 * c + A_BYTES);
      }
#pragma unroll
    for (int c = 0; c < 2; ++c) {
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
          acc[i][j] = mfma16<H16>(al[c][i], F.h[c][j], acc[i][j]);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
          acc[i][j] = mfma16<H16>(ah[c][i], F.l[c][j], acc[i][j]);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
          acc[i][j] = mfma16<H16>(ah[c][i], F.h[c][j], acc[i][j]);
    }
  };

  // ------------------------------------------------------------ gather / dense tiles
  // Iteration order: channel chunk OUTER, direction INNER (it -> dir = it % ndir, chunk = it / ndir):
  // a row tile touches the same ~(tile + halo) source rows 7 (27) times back to back (L1/L2 hits).
  // Pipeline: indices two tiles ahead, A rows + weight fragments one tile ahead (issued before the
  // MFMA block and pinned there); the loop is unrolled by two over alternating fragment / index
  // register sets so that nothing is copied (a copy would wait for its load).
  if (g_begin < g_end) {
    RegSet R0, R1;
    IdxSet I0 = {0, 0, 0, 0}, I1 = {0, 0, 0, 0};
    BFrag F0, F1;
    auto clampg = [&](int it) { return it < nkt_g ? it : nkt_g - 1; };
    auto ktw_of = [&](int it) {
      const int itc = clampg(it);
      return MODE == MODE_DENSE ? itc : (itc % ndir) * tpd + itc / ndir;
    };
    auto load_idx = [&](int it, IdxSet& I) {
      if (MODE == MODE_DENSE) return;
      const int dc = clampg(it) % ndir;
      I.i0 = (uint32_t)t0[dc]; I.i1 = (uint32_t)t1[dc]; I.i2 = (uint32_t)t2[dc]; I.i3 = (uint32_t)t3[dc];
    };
    auto load_rows = [&](int it, const IdxSet& I, RegSet& R) {
      const int itc = clampg(it);
      if (MODE == MODE_DENSE) {
        int cc = itc * BK + c4 * 4;
        cc = cc < kclamp ? cc : kclamp;
        R.a0 = ldg4(dp0 + cc); R.a1 = ldg4(dp1 + cc); R.a2 = ldg4(dp2 + cc); R.a3 = ldg4(dp3 + cc);
      } else {
        const int cc = (itc / ndir) * BK + c4 * 4;
        R.a0 = ldg4(xp + src_off(I.i0, ldx, n_src, aux_delta) + cc);
        R.a1 = ldg4(xp + src_off(I.i1, ldx, n_src, aux_delta) + cc);
        R.a2 = ldg4(xp + src_off(I.i2, ldx, n_src, aux_delta) + cc);
        R.a3 = ldg4(xp + src_off(I.i3, ldx, n_src, aux_delta) + cc);
      }
    };
    auto load_bchunk = [&](int ktw, int c, BFrag& F) {
      const gqp wk = w16 + (int64_t)ktw * 4 * Ncols + (int64_t)(2 * c) * Ncols;
#pragma unroll
      for (int j = 0; j < NI; ++j) { F.h[c][j] = wk[bcol[j]]; F.l[c][j] = wk[lo_off + bcol[j]]; }
    };
    auto store_half = [&](int buf, const float4& v0, const float4& v1, int row_off) {
      char* a = a_st + buf * BUF_BYTES + row_off * 80;
      uint2 hi, lo;
      split16x4<H16>(v0, hi, lo);
      *reinterpret_cast<uint2*>(a) = hi; *reinterpret_cast<uint2*>(a + A_BYTES) = lo;
      split16x4<H16>(v1, hi, lo);
      *reinterpret_cast<uint2*>(a + 32 * 80) = hi; *reinterpret_cast<uint2*>(a + 32 * 80 + A_BYTES) = lo;
    };
#define OFX_FENCE() __builtin_amdgcn_sched_barrier(0)
    // One pipeline step computes tile `it` from LDS buffer `buf` with weight fragments Fcur.
    // In flight: A rows TWO tiles ahead (requested into Rnew; Rold = tile it+1 is converted and written to
    // LDS in this step), weight fragments one tile ahead (Fnext), indices three tiles ahead.
    // The 24 MFMAs are issued as six groups of four; every other piece of work (index loads, fragment
    // loads, row gathers, bf16 split + LDS stores) sits BETWEEN groups, pinned by fences, so it issues in
    // the shadow of the preceding group's 128 matrix-pipe cycles.  (Measured before this interleave, per
    // wave per step: 1279 cycles issuing the prefetch, 979 in the MFMA block, 356 storing -- serial.)
    auto step = [&](int it, int buf, const IdxSet& Iuse, IdxSet& Iload, const BFrag& Fcur, BFrag& Fnext,
                    RegSet& Rnew, const RegSet& Rold) {
      const char* a = a_ld + buf * BUF_BYTES;
      const int ktw = ktw_of(it + 1);
      u32x4h ah[2][MI], al[2][MI];
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          ah[c][i] = *reinterpret_cast<const u32x4h*>(a + i * 32 * 80 + 32 * c);
          al[c][i] = *reinterpret_cast<const u32x4h*>(a + i * 32 * 80 + 32 * c + A_BYTES);
        }
      load_idx(it + 3, Iload);
      OFX_FENCE();
#define OFX_MFMA_GROUP(AOP, BOP, C)                                                                        \
      _Pragma("unroll") for (int i = 0; i < MI; ++i)                                                        \
        _Pragma("unroll") for (int j = 0; j < NI; ++j)                                                      \
          acc[i][j] = mfma16<H16>(AOP[C][i], Fcur.BOP[C][j], acc[i][j]);
      OFX_MFMA_GROUP(al, h, 0);
      OFX_FENCE();
      load_bchunk(ktw, 0, Fnext);
      OFX_FENCE();
      OFX_MFMA_GROUP(ah, l, 0);
      OFX_FENCE();
      load_bchunk(ktw, 1, Fnext);
      OFX_FENCE();
      OFX_MFMA_GROUP(ah, h, 0);
      OFX_FENCE();
      load_rows(it + 2, Iuse, Rnew);
      OFX_FENCE();
      OFX_MFMA_GROUP(al, h, 1);
      OFX_FENCE();
      store_half(buf ^ 1, Rold.a0, Rold.a1, 0);        // tile it+1, requested during the previous step
      OFX_FENCE();
      OFX_MFMA_GROUP(ah, l, 1);
      OFX_FENCE();
      store_half(buf ^ 1, Rold.a2, Rold.a3, 64);
      OFX_FENCE();
      OFX_MFMA_GROUP(ah, h, 1);
#undef OFX_MFMA_GROUP
      __syncthreads();
    };

    // prologue (same issue order as the steady state: indices, then rows / fragments)
    load_idx(g_begin, I0);
    load_idx(g_begin + 1, I1);
    OFX_FENCE();
    load_rows(g_begin, I0, R0);
    load_bchunk(ktw_of(g_begin), 0, F0);
    load_bchunk(ktw_of(g_begin), 1, F0);
    OFX_FENCE();
    load_idx(g_begin + 2, I0);
    OFX_FENCE();
    load_rows(g_begin + 1, I1, R1);
    OFX_FENCE();
    store_half(0, R0.a0, R0.a1, 0);
    store_half(0, R0.a2, R0.a3, 64);
    __syncthreads();
    // steady state: Rold holds tile it+1, Iuse the indices of tile it+2
    int it = g_begin;
    for (; it + 1 < g_end; it += 2) {
      step(it, 0, I0, I1, F0, F1, R0, R1);
      step(it + 1, 1, I1, I0, F1, F0, R1, R0);
    }
    if (it < g_end) step(it, 0, I0, I1, F0, F1, R0, R1);
    __syncthreads();
#undef OFX_FENCE
  }

  // ------------------------------------------------------------ node-type slab tiles
  for (int kt = t_begin; kt < kt_end; ++kt) {
    const int cc = (kt - nkt_g) * BK + c4 * 4;
    RegSet R;
    BFrag F;
    R.a0 = ldg4(tfp + m_0 * ldt + cc);
    R.a1 = ldg4(tfp + m_1 * ldt + cc);
    R.a2 = ldg4(tfp + m_2 * ldt + cc);
    R.a3 = ldg4(tfp + m_3 * ldt + cc);
    load_bfrag(kt, F);
    store_a(0, R);
    __syncthreads();
    compute(0, F);
    __syncthreads();
  }

  epilogue_store<WM, WN, MI, NI>(g, acc, m0, n0, wm, wn, l31, h, split);
}

// split the fp32-packed weights [k/4][n][4] into bf16 hi | lo planes, each [k/8][n][8]
__device__ __forceinline__ uint32_t bf16_rne_bits(float f) {
  const uint32_t u = __float_as_uint(f);
  return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
__global__ void pack_bf16x3_kernel(const float* __restrict__ Wp, int64_t Kp, int64_t N, uint16_t* __restrict__ W16,
                                   int h16) {
  const int64_t total = Kp * N;            // one element per (k, n)
  // per-tensor power-of-two scale of the fp16 halves (trailer behind the 16-bit planes, ofx_launch_weight_scale): the
  // epilogues multiply the accumulators by its inverse (GemmArgs::oscale_p; include/ofx.h, range guard)
  const float wscale = h16 ? reinterpret_cast<const float*>(W16 + 2 * total)[1] : 1.f;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t kg = t / (N * 8), rem = t - kg * N * 8;
    const int64_t n = rem / 8;
    const int kk = (int)(rem - n * 8);
    const int64_t k = kg * 8 + kk;
    const float w = Wp[((k >> 2) * N + n) * 4 + (k & 3)];
    if (h16) {                             // fp16 pairs (precision 3)
      const float ws = sat_f16(w * wscale);
      const _Float16 hi = (_Float16)ws;
      const _Float16 lo = (_Float16)(ws - (float)hi);
      W16[t] = __builtin_bit_cast(unsigned short, hi);
      W16[total + t] = __builtin_bit_cast(unsigned short, lo);
    } else {
      const uint32_t hi = bf16_rne_bits(w);
      const float wl = w - __uint_as_float(hi << 16);
      W16[t] = (uint16_t)hi;
      W16[total + t] = (uint16_t)bf16_rne_bits(wl);
    }
  }
}
static int g_precision = 3;
static int g_gemm_bn64 = 0;                 // 0: off; n: dense GEMMs with N <= n and M >= 64 k rows use 64-column tiles
extern "C" int ofx_set_gemm_bn64(int max_n) { g_gemm_bn64 = max_n < 0 ? 0 : max_n; return OFX_OK; }
// the 16-bit planes behind the fp32 pack follow the precision that is set WHEN THE WEIGHTS ARE PACKED (bf16 pairs for
// precisions 0 / 2, fp16 pairs for 3; unused by 1): the Python cache keys its packs on the precision, C callers re-pack
// after ofx_set_precision
static int pack_bf16x3(const float* Wp, int64_t Kp, int64_t N, hipStream_t st) {
  uint16_t* W16 = reinterpret_cast<uint16_t*>(const_cast<float*>(Wp) + Kp * N);
  if (ofx_launch_weight_scale(Wp, 1, 0, Kp * N, 1, g_precision == 3 ? 1 : 0, const_cast<float*>(Wp) + 2 * Kp * N, st))
    return OFX_ELAUNCH;
  pack_bf16x3_kernel<<<ofx_grid(Kp * N, 256), 256, 0, st>>>(Wp, Kp, N, W16, g_precision == 3 ? 1 : 0);
  return hipGetLastError() == hipSuccess ? OFX_OK : OFX_ELAUNCH;
}

// 3 (default): fp16x3 -- operands as fp16 hi + lo pairs, three v_mfma_f32_32x32x16_f16 per product, fp32 accumulate:
//    ~2^-21 per product (the fp32 reference's own rounding class) at the cost of bf16x3;
// 0: bf16x3 -- the same with bf16 pairs (2^-16 per product; round 1 / 2's default); 1: exact fp32 MFMA;
// 2: fp16 single pass in the planes GraphConv (ofx_gemm2.hip / ofx_gemm3.hip), bf16x3 here.
// Process-wide and unsynchronised: set it from the launching thread between launches.
extern "C" int ofx_set_precision(int mode) {
  if (mode < 0 || mode > 3) return OFX_EINVAL;
  g_precision = mode;
  return OFX_OK;
}
extern "C" int ofx_get_precision(void) { return g_precision; }
// fp32 pack [Kp/4][N][4] | 16-bit hi / lo planes (Kp * N floats' worth) | trailer {1 / s, s, ...} (32 floats)
extern "C" int64_t ofx_packed_floats(int64_t Kp, int64_t N) { return 2 * Kp * N + 32; }

// aux[0, :] = 0; aux[1 + v, :] = mean over segment multi_seg[v] of x[col, :]
__global__ void __launch_bounds__(256) multi_mean_kernel(const float* __restrict__ x, int64_t ldx, int cin,
                                                         const int32_t* __restrict__ seg_ptr,
                                                         const int32_t* __restrict__ col,
                                                         const int32_t* __restrict__ multi_seg, int64_t V,
                                                         float* __restrict__ aux, int64_t ldaux,
                                                         const float* __restrict__ w) {
  const int c4n = cin >> 2;
  const int64_t total = (V + 1) * c4n;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t v = t / c4n;
    const int c = (int)(t - v * c4n) * 4;
    float4 acc = f4zero();
    if (v > 0) {
      const int64_t s = multi_seg[v - 1];
      const int32_t a = seg_ptr[s], e = seg_ptr[s + 1];
      if (w) {                                         // weighted sum (backward pass: reverse graph)
        for (int32_t p = a; p < e; ++p) {
          const float wp = w[p];
          const float4 xv = *reinterpret_cast<const float4*>(x + (int64_t)col[p] * ldx + c);
          acc.x += wp * xv.x; acc.y += wp * xv.y; acc.z += wp * xv.z; acc.w += wp * xv.w;
        }
      } else {
        for (int32_t p = a; p < e; ++p) f4add(acc, *reinterpret_cast<const float4*>(x + (int64_t)col[p] * ldx + c));
        const float inv = 1.f / (float)(e - a);
        acc.x *= inv; acc.y *= inv; acc.z *= inv; acc.w *= inv;
      }
    }
    *reinterpret_cast<float4*>(aux + v * ldaux + c) = acc;
  }
}

__global__ void __launch_bounds__(256) splitk_reduce_kernel(const GemmArgs g) {
  const int64_t total = g.M * g.N;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t m = t / g.N, n = t - m * g.N;
    float v = 0.f;
    for (int s = 0; s < g.nsplit; ++s) v += g.ws[(int64_t)s * total + t];
    if (g.oscale_p) v *= *g.oscale_p;
    if (g.bias) v += g.bias[n];
    if (g.emb) v += g.emb[(int64_t)g.bid[m] * g.lde + n];
    if (g.res) v += g.res[m * g.ldr + n];
    int64_t om = m;
    if (g.out_rows) { om = g.out_rows[m]; if (om < 0) continue; }
    g.out[om * g.ldc + n] = v;
  }
}

// float4 flavour (N % 4 == 0, 16-B aligned operands: g.vec4): one thread = four columns of one row; the partial loads
// of four slices are issued together.  The scalar kernel above ran at ~7 us on the 8^3 / 4^3 grids of the dense net
// (34 launches per step).
__global__ void __launch_bounds__(256) splitk_reduce_v4_kernel(const GemmArgs g) {
  const int64_t n4 = g.N >> 2, total4 = g.M * n4, total = g.M * g.N;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total4; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t m = t / n4, n = (t - m * n4) * 4;
    const float* p = g.ws + m * g.N + n;
    float4 v = f4zero();
    int s = 0;
    // (a thread's loads are a chain of memory round trips: eight in flight instead of four)
    for (; s + 7 < g.nsplit; s += 8) {
      float4 a[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) a[u] = *reinterpret_cast<const float4*>(p + (int64_t)(s + u) * total);
#pragma unroll
      for (int u = 0; u < 8; ++u) f4add(v, a[u]);                        // slice order, as the scalar kernel
    }
    for (; s + 3 < g.nsplit; s += 4) {
      const float4 a0 = *reinterpret_cast<const float4*>(p + (int64_t)s * total);
      const float4 a1 = *reinterpret_cast<const float4*>(p + (int64_t)(s + 1) * total);
      const float4 a2 = *reinterpret_cast<const float4*>(p + (int64_t)(s + 2) * total);
      const float4 a3 = *reinterpret_cast<const float4*>(p + (int64_t)(s + 3) * total);
      f4add(v, a0); f4add(v, a1); f4add(v, a2); f4add(v, a3);
    }
    for (; s < g.nsplit; ++s) f4add(v, *reinterpret_cast<const float4*>(p + (int64_t)s * total));
    if (g.oscale_p) { const float osc = *g.oscale_p; v.x *= osc; v.y *= osc; v.z *= osc; v.w *= osc; }
    if (g.bias) f4add(v, *reinterpret_cast<const float4*>(g.bias + n));
    if (g.emb) f4add(v, *reinterpret_cast<const float4*>(g.emb + (int64_t)g.bid[m] * g.lde + n));
    if (g.res) f4add(v, *reinterpret_cast<const float4*>(g.res + m * g.ldr + n));
    int64_t om = m;
    if (g.out_rows) { om = g.out_rows[m]; if (om < 0) continue; }
    if (g.out_planes) ofx_store_planes4(g.out, om * g.ldc + n, v, g.out_planes);
    else *reinterpret_cast<float4*>(g.out + om * g.ldc + n) = v;
  }
}

template <int MODE, int WM, int WN, int MI, int NI>
static int launch_cfg(GemmArgs& g, hipStream_t st) {
  constexpr int BN = WN * NI * 32;
  constexpr size_t lds = (2 * BM * A_LD + 2 * 8 * BN * 4) * sizeof(float);
  static bool attr_set[OFX_MAX_DEVICES] = {};
  if (lds > 64 * 1024 &&      // e.g. 68 KB for BN = 128: above the 64 KB default cap
      !ofx_raise_lds_limit(reinterpret_cast<const void*>(&gemm_kernel<MODE, WM, WN, MI, NI>), (int)lds, attr_set))
    return OFX_ELAUNCH;
  gemm_kernel<MODE, WM, WN, MI, NI><<<g.ntm * g.ntn * g.nsplit, 256, lds, st>>>(g);
  return OFX_OK;
}

template <int MODE, int WM, int WN, int MI, int NI>
static int launch_fast_cfg(GemmArgs& g, hipStream_t st) {
  constexpr int BN = WN * NI * 32;
  constexpr size_t lds = (2 * BM * A_LD + 2 * 8 * BN * 4) * sizeof(float);
  static bool attr_set[OFX_MAX_DEVICES] = {};
  if (lds > 64 * 1024 &&      // e.g. 68 KB for BN = 128: above the 64 KB default cap
      !ofx_raise_lds_limit(reinterpret_cast<const void*>(&gemm_fast_kernel<MODE, WM, WN, MI, NI>), (int)lds, attr_set))
    return OFX_ELAUNCH;
  gemm_fast_kernel<MODE, WM, WN, MI, NI><<<g.ntm * g.ntn * g.nsplit, 256, lds, st>>>(g);
  return OFX_OK;
}

template <int MODE, int WM, int WN, int MI, int NI, int H16>
static int launch_bf16x3_h(GemmArgs& g, hipStream_t st) {
  constexpr int BN = WN * NI * 32;
  constexpr size_t lds = 2 * (2 * BM * 80);
  static bool attr_set[OFX_MAX_DEVICES] = {};
  if (lds > 64 * 1024 &&      // e.g. 68 KB for BN = 128: above the 64 KB default cap
      !ofx_raise_lds_limit(reinterpret_cast<const void*>(&gemm_pairs_x3_kernel<MODE, WM, WN, MI, NI, H16>), (int)lds, attr_set))
    return OFX_ELAUNCH;
  gemm_pairs_x3_kernel<MODE, WM, WN, MI, NI, H16><<<g.ntm * g.ntn * g.nsplit, 256, lds, st>>>(g);
  return OFX_OK;
}
template <int MODE, int WM, int WN, int MI, int NI>
static int launch_bf16x3_cfg(GemmArgs& g, hipStream_t st) {
  return g_precision == 3 ? launch_bf16x3_h<MODE, WM, WN, MI, NI, 1>(g, st) : launch_bf16x3_h<MODE, WM, WN, MI, NI, 0>(g, st);
}

// second stage of the fused statistics.  Block = 64 consecutive wave rows x 64 columns; thread (rg, c) adds up
// 16 wave rows of column c (all 32 loads issued up front), the four row groups meet in LDS and one fp64 atomic
// pair per column leaves the block when its wave rows share a batch element (else every thread flushes its
// own runs).  All rows of a wave row share one batch element: mixed waves stored zeros and used atomics.
__global__ void __launch_bounds__(256) stats_reduce_kernel(const float* __restrict__ part, int64_t nwr, int wr_rows,
                                                            int64_t N, const int32_t* __restrict__ bid,
                                                            double* __restrict__ stats, int64_t stats_ld) {
  __shared__ double red[3][64][2];
  const int c = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int64_t n = (int64_t)blockIdx.y * 64 + c;
  const int64_t w0 = (int64_t)blockIdx.x * 64;
  const int64_t nc = n < N ? n : N - 1;
  int b[16];
  float2 v[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int64_t w = w0 + rg * 16 + i;
    const int64_t wc = w < nwr ? w : nwr - 1;
    b[i] = bid[wc * wr_rows];
    v[i] = *reinterpret_cast<const float2*>(part + (wc * N + nc) * 2);
    if (w >= nwr) { b[i] = -1; v[i] = make_float2(0.f, 0.f); }
  }
  const int bfirst = bid[w0 * wr_rows];
  bool ok = true;
#pragma unroll
  for (int i = 0; i < 16; ++i) ok = ok && (b[i] == bfirst || b[i] < 0);
  const bool uni = __syncthreads_and(ok);
  auto flush = [&](int bb, double s, double q) {
    double* o = stats + ((int64_t)bb * stats_ld + n) * 2;
    unsafeAtomicAdd(o, s); unsafeAtomicAdd(o + 1, q);
  };
  if (uni) {
    double s = 0.0, q = 0.0;
#pragma unroll
    for (int i = 0; i < 16; ++i) { s += (double)v[i].x; q += (double)v[i].y; }
    if (rg > 0) { red[rg - 1][c][0] = s; red[rg - 1][c][1] = q; }
    __syncthreads();
    if (rg == 0 && n < N) {
#pragma unroll
      for (int r = 0; r < 3; ++r) { s += red[r][c][0]; q += red[r][c][1]; }
      flush(bfirst, s, q);
    }
  } else if (n < N) {
    double s = 0.0, q = 0.0;
    int sb = -1;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (b[i] < 0) continue;
      if (b[i] != sb) {
        if (sb >= 0) flush(sb, s, q);
        sb = b[i]; s = 0.0; q = 0.0;
      }
      s += (double)v[i].x; q += (double)v[i].y;
    }
    if (sb >= 0) flush(sb, s, q);
  }
}

// second stage of the fused statistics for a launch whose waves own `wr_rows` rows each (also used by ofx_gemm2.hip)
int ofx_launch_stats_reduce(const GemmArgs& g, int wr_rows, hipStream_t st) {
  const int64_t nwr = ofx_cdiv(g.M, wr_rows);
  stats_reduce_kernel<<<dim3((unsigned)ofx_cdiv(nwr, 64), (unsigned)ofx_cdiv(g.N, 64)), 256, 0, st>>>(
      g.stats_part, nwr, wr_rows, g.N, g.bid, g.stats, g.stats_ld);
  OFX_LAUNCH_CHECK();
  return OFX_OK;
}

template <int MODE>
static int launch_gemm(GemmArgs& g, float* ws, size_t ws_bytes, hipStream_t st) {
  if (g.M <= 0 || g.N <= 0) return OFX_OK;
  {
    auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    g.vec4 = g.N % 4 == 0 && al16(g.out) && g.ldc % 4 == 0 && (!g.res || (al16(g.res) && g.ldr % 4 == 0)) &&
             (!g.emb || (al16(g.emb) && g.lde % 4 == 0)) && (!g.bias || al16(g.bias));
  }
  int bn = g.N <= 32 ? 32 : (g.N <= 64 ? 64 : 128);
  // A/B knob (ofx_set_gemm_bn64): 64-column tiles for the long, narrow, HBM-bound GEMMs (the 1x1 skip convolutions:
  // N = 128, M >= 64 k rows) -- 138 instead of 202 VGPRs, three blocks per CU instead of two, i.e. half as many more
  // row loads in flight, at the price of reading every A tile twice (the second time from L2)
  if (g_gemm_bn64 && MODE == MODE_DENSE && bn == 128 && g.N <= g_gemm_bn64 && g.M >= 65536) bn = 64;
  g.ntm = (int)ofx_cdiv(g.M, BM);
  g.ntn = (int)ofx_cdiv(g.N, bn);
  const int nkt = (int)(g.Kp / BK);
  // split-K when the tile grid cannot fill 256 CUs (x2 blocks) and K is long enough to share
  int nsplit = 1;
  const int tiles = g.ntm * g.ntn;
  if (ws && tiles < 256 && nkt >= 8) {
    // (512 blocks aimed for, >= 4 k tiles per slice: swept in round 3 on the lr step -- 256 / 1024 blocks and 2 / 8
    // k tiles are all 0 - 3 % slower)
    nsplit = (int)ofx_cdiv(512, tiles);
    if (nsplit > nkt / 4) nsplit = nkt / 4;
    if (nsplit > 64) nsplit = 64;
    const size_t per = (size_t)g.M * g.N * sizeof(float);
    if ((size_t)nsplit * per > ws_bytes) nsplit = (int)(ws_bytes / per);
    if (nsplit < 2) nsplit = 1;
  }
  g.kt_per_split = (int)ofx_cdiv(nkt, nsplit);
  nsplit = (int)ofx_cdiv(nkt, g.kt_per_split);
  g.nsplit = nsplit;
  g.ws = ws;
  int rc;
  bool fast;
  if (MODE == MODE_GATHER) {
    fast = g.fast && g.nbr_ext && g.aux && (!g.out_rows || g.gather_out_rows);
  } else {
    fast = ((g.lda & 3) == 0) && ((g.K & 3) == 0) && g.K >= 4 && ((((uintptr_t)g.A) & 15) == 0);
    if (fast) { g.ndir = 1; g.n_src = 0; g.aux = g.A; g.tf = g.A; g.ldt = g.lda; g.nbr_ext = (const int32_t*)g.A; }
  }
  g.W16 = (g_precision != 1) ? reinterpret_cast<const uint16_t*>(g.Wp + g.Kp * g.N) : nullptr;
  const int wr_rows = bn == 32 ? 32 : 64;                   // rows per wave in the three tile configurations
  const int64_t nwr = ofx_cdiv(g.M, wr_rows);
  if (!(g.stats && g.vec4 && nsplit == 1 && g.stats_part &&
        (size_t)nwr * g.N * 2 * sizeof(float) <= g.stats_part_bytes && (((uintptr_t)g.stats_part) & 15) == 0))
    g.stats_part = nullptr;
  // the 16-bit halves carry the pack's power-of-two scale; the fp32 pack (exact kernels below) does not
  g.oscale_p = (fast && g.W16) ? g.Wp + 2 * g.Kp * g.N : nullptr;
  if (fast && g.W16) {
    if (bn == 32) rc = launch_bf16x3_cfg<MODE, 4, 1, 1, 1>(g, st);
    else if (bn == 64) rc = launch_bf16x3_cfg<MODE, 2, 2, 2, 1>(g, st);
    else rc = launch_bf16x3_cfg<MODE, 2, 2, 2, 2>(g, st);
  } else if (fast) {
    if (bn == 32) rc = launch_fast_cfg<MODE, 4, 1, 1, 1>(g, st);
    else if (bn == 64) rc = launch_fast_cfg<MODE, 2, 2, 2, 1>(g, st);
    else rc = launch_fast_cfg<MODE, 2, 2, 2, 2>(g, st);
  } else if (bn == 32) rc = launch_cfg<MODE, 4, 1, 1, 1>(g, st);
  else if (bn == 64) rc = launch_cfg<MODE, 2, 2, 2, 1>(g, st);
  else rc = launch_cfg<MODE, 2, 2, 2, 2>(g, st);
  if (rc) return rc;
  if (nsplit > 1) {
    const bool ws16 = (((uintptr_t)g.ws) & 15) == 0;
    if (g.vec4 && ws16) splitk_reduce_v4_kernel<<<ofx_grid(g.M * (g.N >> 2), 256), 256, 0, st>>>(g);
    else splitk_reduce_kernel<<<ofx_grid(g.M * g.N, 256), 256, 0, st>>>(g);
  }
  if (g.stats_part) {
    rc = ofx_launch_stats_reduce(g, wr_rows, st);
    if (rc) return rc;
  }
  OFX_LAUNCH_CHECK();
  return OFX_OK;
}

static inline int64_t pad32(int64_t v) { return (v + 31) / 32 * 32; }
static int pack_bf16x3(const float* Wp, int64_t Kp, int64_t N, hipStream_t st);

extern "C" int64_t ofx_packed_k(int64_t K) { return pad32(K); }
extern "C" int64_t ofx_graphconv_packed_k(int cin, int nt) {
  return pad32(7 * (int64_t)cin) + (nt > 1 ? pad32(7 * (int64_t)nt) : 0);
}

__global__ void pack_weights_kernel(const float* __restrict__ W, int64_t sk, int64_t sn, int64_t K, int64_t N, int cin,
                                    int nt, float* __restrict__ Wp, int64_t Kp) {
  const int64_t total = (Kp / 4) * N;
  const int ntc = nt > 1 ? nt : 0;
  const int64_t Kf = (7 * (int64_t)cin + 31) / 32 * 32;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t kq = t / N, n = t - kq * N;
    float v[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int64_t k = kq * 4 + kk;
      int64_t src = -1;
      if (cin > 0) {
        if (k < 7 * (int64_t)cin) {
          const int64_t dir = k / cin, c = k - dir * cin;
          src = dir * (cin + ntc) + c;
        } else if (k >= Kf && ntc > 0 && k - Kf < 7 * (int64_t)ntc) {
          const int64_t kt = k - Kf, dir = kt / ntc, ty = kt - dir * ntc;
          src = dir * (cin + ntc) + cin + ty;
        }
      } else if (k < K) {
        src = k;
      }
      v[kk] = src >= 0 ? W[src * sk + n * sn] : 0.f;
    }
    *reinterpret_cast<float4*>(Wp + t * 4) = make_float4(v[0], v[1], v[2], v[3]);
  }
}

extern "C" int ofx_pack_weights(const float* W, int64_t sk, int64_t sn, int64_t K, int64_t N, int cin, int nt,
                                float* Wp, int64_t Kp, void* stream) {
  if (!W || !Wp || K < 1 || N < 1 || (Kp & 31) || cin < 0 || nt < 0) return OFX_EINVAL;
  if (cin > 0) {
    if (Kp != ofx_graphconv_packed_k(cin, nt) || K != 7 * (int64_t)(cin + (nt > 1 ? nt : 0))) return OFX_EINVAL;
  } else if (Kp != pad32(K)) {
    return OFX_EINVAL;
  }
  if (((uintptr_t)Wp & 15) != 0) return OFX_EINVAL;
  pack_weights_kernel<<<ofx_grid((Kp / 4) * N, 256), 256, 0, ofx_stream(stream)>>>(W, sk, sn, K, N, cin, nt, Wp, Kp);
  if (int rc = pack_bf16x3(Wp, Kp, N, ofx_stream(stream))) return rc;
  OFX_LAUNCH_CHECK();
  return OFX_OK;
}

// nn.Conv3d weight [cout, cin, 3,3,3] -> packed k = tap*cin + c, tap = (kx*3+ky)*3+kz
__global__ void pack_conv3d_kernel(const float* __restrict__ W, int cin, int64_t N, float* __restrict__ Wp, int64_t Kp) {
  const int64_t total = (Kp / 4) * N;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t kq = t / N, n = t - kq * N;
    float v[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int64_t k = kq * 4 + kk;
      float w = 0.f;
      if (k < 27 * (int64_t)cin) {
        const int64_t tap = k / cin, c = k - tap * cin;
        w = W[(n * cin + c) * 27 + tap];
      }
      v[kk] = w;
    }
    *reinterpret_cast<float4*>(Wp + t * 4) = make_float4(v[0], v[1], v[2], v[3]);
  }
}

extern "C" int64_t ofx_conv3d_packed_k(int cin) { return pad32(27 * (int64_t)cin); }

extern "C" int ofx_pack_conv3d(const float* W, int cin, int cout, float* Wp, void* stream) {
  if (!W || !Wp || cin < 1 || cout < 1 || ((uintptr_t)Wp & 15)) return OFX_EINVAL;
  const int64_t Kp = ofx_conv3d_packed_k(cin);
  pack_conv3d_kernel<<<ofx_grid((Kp / 4) * cout, 256), 256, 0, ofx_stream(stream)>>>(W, cin, cout, Wp, Kp);
  if (int rc = pack_bf16x3(Wp, Kp, cout, ofx_stream(stream))) return rc;
  OFX_LAUNCH_CHECK();
  return OFX_OK;
}

extern "C" int ofx_gemm_f32_planes(const float* A, int64_t lda, const int32_t* a_rows, int64_t M, int64_t K,
                                   const float* Wp, int64_t Kp, int64_t N, const float* bias, const float* res, int64_t ldr,
                                   float* out, int64_t ldc, const int32_t* out_rows, void* ws, size_t ws_bytes,
                                   int out_mode, void* stream) {
  if (M == 0 && K >= 1 && N >= 1) return OFX_OK;          // empty input: nothing to do (out may be a null pointer)
  if (M < 0 || K < 1 || N < 1 || !Wp || !out || (M > 0 && !A) || Kp != pad32(K) || lda < K || ldc < N ||
      (res && ldr < N) || ((uintptr_t)Wp & 15))
    return OFX_EINVAL;
  if (out_mode != 0 && out_mode != 2 && out_mode != 3) return OFX_EINVAL;
  // pair planes: the float4 epilogues only, whole 128-B lines (the planes of a buffer are a function of the flat index)
  if (out_mode && ((N & 3) || (ldc & 31) || ((uintptr_t)out & 127) || (res && ((ldr & 3) || ((uintptr_t)res & 15))) ||
                   (bias && ((uintptr_t)bias & 15)) || ((uintptr_t)ws & 15)))
    return OFX_EINVAL;
  GemmArgs g = {};
  g.A = A; g.lda = lda; g.a_rows = a_rows;
  g.M = M; g.K = K; g.Wp = Wp; g.Kp = Kp; g.N = N; g.bias = bias;
  g.res = res; g.ldr = ldr; g.out = out; g.ldc = ldc; g.out_rows = out_rows; g.out_planes = out_mode;
  return launch_gemm<MODE_DENSE>(g, (float*)ws, ws_bytes, ofx_stream(stream));
}
extern "C" int ofx_gemm_f32(const float* A, int64_t lda, const int32_t* a_rows, int64_t M, int64_t K, const float* Wp,
                            int64_t Kp, int64_t N, const float* bias, const float* res, int64_t ldr, float* out,
                            int64_t ldc, const int32_t* out_rows, void* ws, size_t ws_bytes, void* stream) {
  return ofx_gemm_f32_planes(A, lda, a_rows, M, K, Wp, Kp, N, bias, res, ldr, out, ldc, out_rows, ws, ws_bytes, 0, stream);
}

static int gather_common(GemmArgs& g, const float* x, int64_t ldx, int cin, int ndir, int64_t n_rows,
                         const int32_t* nbr, const int32_t* seg_ptr, const int32_t* col, const float* Wp, int64_t Kp,
                         int cout, const float* bias, const float* emb, int64_t lde, const int32_t* batch_id,
                         const float* res, int64_t ldr, float* out, int64_t ldc) {
  if (n_rows < 0 || cin < 1 || cout < 1 || !x || !nbr || !Wp || !out || ldx < cin || ldc < cout ||
      (res && ldr < cout) || (emb && (!batch_id || lde < cout)) || ((uintptr_t)Wp & 15))
    return OFX_EINVAL;
  g.x = x; g.ldx = ldx; g.cin = cin; g.ndir = ndir;
  g.fast = (cin % 32 == 0) && ((ldx & 3) == 0) && (((uintptr_t)x & 15) == 0);
  g.nbr = nbr; g.seg_ptr = seg_ptr; g.col = col;
  g.Kf = pad32((int64_t)ndir * cin);
  g.M = n_rows; g.K = Kp; g.Wp = Wp; g.Kp = Kp; g.N = cout; g.bias = bias;
  g.emb = emb; g.lde = lde; g.bid = batch_id; g.res = res; g.ldr = ldr; g.out = out; g.ldc = ldc;
  return OFX_OK;
}

// Layers the branch-free gather kernel cannot take (Cin not a multiple of 32: the network's input conv with
// Cin = 3, the VAE decoder's 24/32-channel convs): materialise the reference's col_data rows
// [rows, Kp] = [7 (27) segment means | node-type slab] into the workspace, one row chunk at a time, and run
// the dense kernel on them with the same packed weights (their k order is exactly this layout).
__global__ void __launch_bounds__(256) col_rows_kernel(const GemmArgs g, int64_t row0, int64_t rows,
                                                       float* __restrict__ colbuf) {
  const int64_t k4n = g.Kp >> 2, total = rows * k4n;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = t / k4n, k = (t - r * k4n) * 4;
    const int64_t row = row0 + r;
    float v[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
      v[kk] = (k + kk < g.Kf) ? gather_elem(g, row, k + kk) : g.tf[row * g.ldt + (k + kk - g.Kf)];
    *reinterpret_cast<float4*>(colbuf + t * 4) = make_float4(v[0], v[1], v[2], v[3]);
  }
}

template <int MODE> static int launch_gemm(GemmArgs& g, float* ws, size_t ws_bytes, hipStream_t st);

// returns OFX_OK when it handled the layer, -1 when the workspace is too small (caller uses the generic kernel)
static int launch_gather_via_col(const GemmArgs& g, float* ws, size_t ws_bytes, hipStream_t st) {
  if (!ws || (((uintptr_t)ws) & 15)) return -1;
  size_t tail = ws_bytes / 8;                                  // split-K partials / statistics partials
  if (tail > (size_t(16) << 20)) tail = size_t(16) << 20;
  tail &= ~size_t(15);
  const size_t row_bytes = (size_t)g.Kp * sizeof(float);
  int64_t chunk = (int64_t)((ws_bytes - tail) / row_bytes) / 128 * 128;
  if (chunk < 128) return -1;
  float* tail_ws = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + (ws_bytes - tail));
  for (int64_t r0 = 0; r0 < g.M; r0 += chunk) {
    const int64_t rows = g.M - r0 < chunk ? g.M - r0 : chunk;
    col_rows_kernel<<<ofx_grid(rows * (g.Kp >> 2), 256), 256, 0, st>>>(g, r0, rows, ws);
    GemmArgs d = {};
    d.A = ws; d.lda = g.Kp; d.M = rows; d.K = g.Kp; d.Wp = g.Wp; d.Kp = g.Kp; d.N = g.N; d.bias = g.bias;
    d.emb = g.emb; d.lde = g.lde; d.bid = g.bid ? g.bid + r0 : nullptr;
    d.res = g.res ? g.res + r0 * g.ldr : nullptr; d.ldr = g.ldr;
    d.out = g.out + r0 * g.ldc; d.ldc = g.ldc;
    d.stats = g.stats; d.stats_ld = g.stats_ld;
    if (g.stats) { d.stats_part = tail_ws; d.stats_part_bytes = tail; }
    const int rc = launch_gemm<MODE_DENSE>(d, g.stats ? nullptr : tail_ws, tail, st);
    if (rc) return rc;
  }
  return OFX_OK;
}

extern "C" int ofx_graphconv_fwd(const float* x, int64_t ldx, int cin, int64_t n_nodes, const int32_t* nbr,
                                 const int32_t* seg_ptr, const int32_t* col, const int32_t* nbr_ext,
                                 const int32_t* multi_seg, int64_t n_multi, float* aux, const float* type_frac,
                                 int64_t ldt, int nt_pad, const float* Wp, int64_t Kp, int cout, const float* bias,
                                 const float* emb, int64_t lde, const int32_t* batch_id, const float* res, int64_t ldr,
                                 float* out, int64_t ldc, double* stats, int64_t stats_ld, void* ws, size_t ws_bytes,
                                 void* stream) {
  if (n_nodes == 0 && cin >= 1 && cout >= 1) return OFX_OK;      // empty graph level: nothing to do
  GemmArgs g = {};
  int rc = gather_common(g, x, ldx, cin, 7, n_nodes, nbr, seg_ptr, col, Wp, Kp, cout, bias, emb, lde, batch_id, res,
                         ldr, out, ldc);
  if (rc) return rc;
  if (!seg_ptr || !col) return OFX_EINVAL;
  if (nt_pad < 0 || (nt_pad & 31) || Kp != g.Kf + nt_pad) return OFX_EINVAL;
  if (nt_pad > 0 && (!type_frac || ldt < nt_pad || (ldt & 3) || ((uintptr_t)type_frac & 15))) return OFX_EINVAL;
  g.tf = type_frac; g.ldt = ldt;
  if (stats) {
    if (!batch_id || stats_ld < cout) return OFX_EINVAL;
    g.stats = stats; g.stats_ld = stats_ld; g.bid = batch_id;
    g.stats_part = (float*)ws; g.stats_part_bytes = ws_bytes;     // workspace holds the per-wave partial sums
    ws = nullptr;                              // fused statistics need the single-pass epilogue (no split-K)
  }
  hipStream_t st = ofx_stream(stream);
  if (g.fast && nbr_ext && aux && n_nodes > 0 && (((uintptr_t)aux & 15) == 0)) {
    if (n_multi < 0 || (n_multi > 0 && !multi_seg)) return OFX_EINVAL;
    // pre-pass: zero row + mean rows of the (few) segments with several neighbours
    multi_mean_kernel<<<ofx_grid((n_multi + 1) * (cin / 4), 256), 256, 0, st>>>(x, ldx, cin, seg_ptr, col, multi_seg,
                                                                               n_multi, aux, ldx, nullptr);
    g.nbr_ext = nbr_ext; g.aux = aux; g.ldaux = ldx; g.n_src = n_nodes;
    if (!g.tf) { g.tf = x; g.ldt = ldx; }       // never dereferenced past the gather tiles; keeps selects defined
  }
  if (!g.nbr_ext && n_nodes > 0) {
    // stats set `ws` aside for the partial sums: the col path manages the whole workspace itself
    rc = launch_gather_via_col(g, stats ? g.stats_part : (float*)ws, stats ? g.stats_part_bytes : ws_bytes, st);
    if (rc >= 0) return rc;
  }
  return launch_gemm<MODE_GATHER>(g, (float*)ws, ws_bytes, st);
}

// Backward of GraphConv with respect to its input (autograd of modules.py:205-213 + :213's matmul):
//   dx[c, :] = sum_dir ( sum over reverse segment (c, dir) of rev_w * dy[rev_row, :] ) @ W_dir^T
// = the same fused gather-GEMM run on the reverse graph with weighted segment SUMS and the transposed weights
// (WpT: packed 'graphconv' layout of W^T_dir stacked over dir, K = 7 * cout, N = cin; the node-type rows of W
// have no input gradient).
extern "C" int ofx_graphconv_bwd_data(const float* dy, int64_t ldy, int cout, int64_t n_nodes, const int32_t* nbr_rev,
                                      const int32_t* rev_ptr, const int32_t* rev_row, const float* rev_w,
                                      const int32_t* nbr_ext_rev, const int32_t* multi_seg, int64_t n_multi, float* aux,
                                      const float* WpT, int64_t KpT, int cin, float* dx, int64_t ldx, void* ws,
                                      size_t ws_bytes, void* stream) {
  if (n_nodes == 0 && cin >= 1 && cout >= 1) return OFX_OK;
  GemmArgs g = {};
  int rc = gather_common(g, dy, ldy, cout, 7, n_nodes, nbr_rev, rev_ptr, rev_row, WpT, KpT, cin, nullptr, nullptr, 0,
                         nullptr, nullptr, 0, dx, ldx);
  if (rc) return rc;
  if (!rev_ptr || !rev_row || !rev_w || KpT != g.Kf) return OFX_EINVAL;
  g.edge_w = rev_w;
  hipStream_t st = ofx_stream(stream);
  if (g.fast && nbr_ext_rev && aux && (((uintptr_t)aux & 15) == 0)) {
    if (n_multi < 0 || (n_multi > 0 && !multi_seg)) return OFX_EINVAL;
    multi_mean_kernel<<<ofx_grid((n_multi + 1) * (cout / 4), 256), 256, 0, st>>>(dy, ldy, cout, rev_ptr, rev_row,
                                                                                multi_seg, n_multi, aux, ldy, rev_w);
    g.nbr_ext = nbr_ext_rev; g.aux = aux; g.ldaux = ldy; g.n_src = n_nodes;
    g.tf = dy; g.ldt = ldy;
  } else {
    g.tf = dy; g.ldt = ldy;                       // col path: no type slab (Kp == Kf), never dereferenced
    rc = launch_gather_via_col(g, (float*)ws, ws_bytes, st);
    if (rc >= 0) return rc;
  }
  return launch_gemm<MODE_GATHER>(g, (float*)ws, ws_bytes, st);
}

// ---------------------------------------------------------------------------------
// Backward of GraphConv with respect to its weights: dW[k, o] = sum_r col_data[r, k] * dy[r, o] -- a "TN"
// contraction whose reduction index is the node row r of BOTH operands.  The 32x32x2 fp32 MFMA takes one
// element per lane for each operand (A: row = lane & 31 at k = lane >> 5), so tiles that are contiguous along
// k-of-W / o (= along the lanes) feed it straight from LDS without any transpose, in exact fp32.
// col_data rows are gathered on the fly exactly like the forward pass (extended table + aux rows from the same
// multi-neighbour pre-pass, node-type slab for the trailing columns); layers the branch-free gather cannot take
// go through col_rows_kernel chunks with DENSE_P = true.
// Grid: (k tiles of 128, o tiles of 128, row slices); partial sums [slice][Kp][cout] in the workspace, reduced in
// slice order (deterministic).
struct TnArgs {
  const float* P; int64_t ldp;              // DENSE_P: col rows [rows, Kp]
  const float* x; int64_t ldx; int cin; int ndir;   // gathered P (ndir source rows per node: 7 graph / 27 grid)
  const int32_t* nbr_ext; const float* aux; int64_t n_src;
  const float* tf; int64_t ldt; int64_t Kf;
  const float* Q; int64_t ldq;              // dy [rows, cout]
  int64_t rows, row0, Kp, N;                // rows in this launch (starting at graph row row0), packed K, cout
  int64_t rows_per_slice;
  float* part;                              // [slices][Kp][N]
};

template <bool DENSE_P, bool BF16X3>
__global__ void __launch_bounds__(256, 2) tn_gemm_kernel(const TnArgs a) {
  // fp32 mode: tiles [32 k][128 + 4] floats, the 32x32x2 MFMA reads them as they are.
  // bf16x3 mode (default contraction precision, 3 x v_mfma_f32_32x32x16_bf16): the operand fragments need 8
  // consecutive k per lane, i.e. TRANSPOSED tiles [128 m][32 k + 8] bf16 (hi and lo planes): the loader maps a
  // quad of lanes to 4 consecutive rows k of the same float4 column, a DPP 4x4 transpose turns that into 4
  // consecutive k of one column m, and each lane writes 8 B per plane.
  constexpr int LD = 132;                   // fp32 LDS row pitch (floats): 128 + 4
  constexpr int TP = 128 * 80;              // bytes of one transposed bf16 plane ([128][40] bf16)
  __shared__ __attribute__((aligned(16))) char tn_smem[BF16X3 ? 4 * TP : 2 * 32 * LD * 4];
  float* const Ps = reinterpret_cast<float*>(tn_smem);
  float* const Qs = Ps + 32 * LD;
  char* const Pt = tn_smem;                 // hi plane, lo plane at + TP
  char* const Qt = tn_smem + 2 * TP;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int wm = wid >> 1, wn = wid & 1, l31 = lane & 31, h = lane >> 5;
  const int64_t m0 = (int64_t)blockIdx.x * 128, n0 = (int64_t)blockIdx.y * 128;
  const int64_t r_begin = (int64_t)blockIdx.z * a.rows_per_slice;
  const int64_t r_end = r_begin + a.rows_per_slice < a.rows ? r_begin + a.rows_per_slice : a.rows;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  // loader mapping: fp32: row kr = tid >> 3, float4 f0 = tid & 7;  bf16x3: quad lane q = k & 3, f0, k >> 2 = tid >> 5
  const int q4 = threadIdx.x & 3;
  const int kr = BF16X3 ? q4 + 4 * (threadIdx.x >> 5) : threadIdx.x >> 3;
  const int f0 = BF16X3 ? (threadIdx.x >> 2) & 7 : threadIdx.x & 7;
  float4 vp[4], vq[4];
  auto load_tile = [&](int64_t r) {
    const int64_t row = r + kr;
    const bool rok = row < r_end;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int64_t m = m0 + 4 * (f0 + 8 * p), n = n0 + 4 * (f0 + 8 * p);
      float4 v = f4zero();
      if (rok && m < a.Kp) {
        if (DENSE_P) {
          v = *reinterpret_cast<const float4*>(a.P + row * a.ldp + m);
        } else if (m < a.Kf) {
          const int64_t kk = m;
          if (kk < a.ndir * (int64_t)a.cin) {
            const int dir = (int)(kk / a.cin), c = (int)(kk - (int64_t)dir * a.cin);
            const int64_t id = a.nbr_ext[(a.row0 + row) * a.ndir + dir];
            const float* src = id < a.n_src ? a.x + id * a.ldx : a.aux + (id - a.n_src) * a.ldx;
            v = *reinterpret_cast<const float4*>(src + c);
          }
        } else {
          v = *reinterpret_cast<const float4*>(a.tf + (a.row0 + row) * a.ldt + (m - a.Kf));
        }
      }
      vp[p] = v;
      float4 q = f4zero();
      if (rok && n < a.N) q = *reinterpret_cast<const float4*>(a.Q + row * a.ldq + n);   // N % 4 == 0
      vq[p] = q;
    }
  };
  auto store_tile = [&]() {
    if (BF16X3) {
      const bool b0 = q4 & 1, b1 = q4 & 2;
      const int kg = threadIdx.x >> 5;
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        float4 v = vp[p], w = vq[p];
        quad_transpose(v.x, v.y, v.z, v.w, b0, b1);      // lane q: column 4F + q, rows 4kg .. 4kg+3
        quad_transpose(w.x, w.y, w.z, w.w, b0, b1);
        const int col = 4 * (f0 + 8 * p) + q4;
        uint2 hi, lo;
        split_bf16x4(v, hi, lo);
        *reinterpret_cast<uint2*>(Pt + col * 80 + kg * 8) = hi;
        *reinterpret_cast<uint2*>(Pt + col * 80 + kg * 8 + TP) = lo;
        split_bf16x4(w, hi, lo);
        *reinterpret_cast<uint2*>(Qt + col * 80 + kg * 8) = hi;
        *reinterpret_cast<uint2*>(Qt + col * 80 + kg * 8 + TP) = lo;
      }
    } else {
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        *reinterpret_cast<float4*>(Ps + kr * LD + 4 * (f0 + 8 * p)) = vp[p];
        *reinterpret_cast<float4*>(Qs + kr * LD + 4 * (f0 + 8 * p)) = vq[p];
      }
    }
  };
  if (r_begin < r_end) load_tile(r_begin);
  for (int64_t r = r_begin; r < r_end; r += 32) {
    __syncthreads();
    store_tile();
    __syncthreads();
    if (r + 32 < r_end) load_tile(r + 32);
    if (BF16X3) {
      const char* pa = Pt + (wm * 64 + l31) * 80 + 16 * h;
      const char* pb = Qt + (wn * 64 + l31) * 80 + 16 * h;
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        bf16x8_t ah[2], al[2], bh[2], bl[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          ah[i] = *reinterpret_cast<const bf16x8_t*>(pa + i * 32 * 80 + 32 * c);
          al[i] = *reinterpret_cast<const bf16x8_t*>(pa + i * 32 * 80 + 32 * c + TP);
          bh[i] = *reinterpret_cast<const bf16x8_t*>(pb + i * 32 * 80 + 32 * c);
          bl[i] = *reinterpret_cast<const bf16x8_t*>(pb + i * 32 * 80 + 32 * c + TP);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
          }
      }
    } else {
      const float* pa = Ps + h * LD + wm * 64 + l31;
      const float* pb = Qs + h * LD + wn * 64 + l31;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const float a0 = pa[2 * q * LD], a1 = pa[2 * q * LD + 32];
        const float b0 = pb[2 * q * LD], b1 = pb[2 * q * LD + 32];
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
      }
    }
  }
  float* out = a.part + (int64_t)blockIdx.z * a.Kp * a.N;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int64_t n = n0 + wn * 64 + j * 32 + l31;
      if (n >= a.N) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (m < a.Kp) out[m * a.N + n] = acc[i][j][r];
      }
    }
}

template <bool DENSE_P>
static void launch_tn(const TnArgs& a, dim3 grid, hipStream_t st) {
  if (g_precision != 1) tn_gemm_kernel<DENSE_P, true><<<grid, 256, 0, st>>>(a);
  else tn_gemm_kernel<DENSE_P, false><<<grid, 256, 0, st>>>(a);
}

__global__ void __launch_bounds__(256) tn_reduce_kernel(const float* __restrict__ part, int slices, int64_t total,
                                                         float* __restrict__ dW, int accumulate) {
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    float v = accumulate ? dW[t] : 0.f;
    for (int s2 = 0; s2 < slices; ++s2) v += part[(int64_t)s2 * total + t];
    dW[t] = v;
  }
}

// dWp: [Kp, cout] in the PACKED k order of ofx_pack_weights (k = dir*cin + c, zero rows up to Kf, then the
// node-type rows).  aux must hold the forward pre-pass rows (ofx_graphconv_fwd leaves them there) when
// nbr_ext is given; it is recomputed here so the call is self-contained.
extern "C" int ofx_graphconv_bwd_weight(const float* x, int64_t ldx, int cin, int64_t n_nodes, const int32_t* nbr,
                                        const int32_t* seg_ptr, const int32_t* col, const int32_t* nbr_ext,
                                        const int32_t* multi_seg, int64_t n_multi, float* aux, const float* type_frac,
                                        int64_t ldt, int nt_pad, const float* dy, int64_t ldy, int cout, float* dWp,
                                        int64_t Kp, void* ws, size_t ws_bytes, void* stream) {
  if (n_nodes < 0 || cin < 1 || cout < 1 || (cout & 3) || !dy || !dWp || !ws || ldy < cout || (ldy & 3) ||
      ((uintptr_t)dy & 15) || ((uintptr_t)ws & 15) || (n_nodes > 0 && (!x || !nbr || !seg_ptr || !col)))
    return OFX_EINVAL;
  const int64_t Kf = pad32(7 * (int64_t)cin);
  if (nt_pad < 0 || (nt_pad & 31) || Kp != Kf + nt_pad) return OFX_EINVAL;
  if (nt_pad > 0 && (!type_frac || ldt < nt_pad || (ldt & 3) || ((uintptr_t)type_frac & 15))) return OFX_EINVAL;
  hipStream_t st = ofx_stream(stream);
  const int64_t total = Kp * cout;
  if (n_nodes == 0) {
    if (hipMemsetAsync(dWp, 0, (size_t)total * sizeof(float), st) != hipSuccess) return OFX_ELAUNCH;
    return OFX_OK;
  }
  const bool fast = (cin % 32 == 0) && ((ldx & 3) == 0) && (((uintptr_t)x & 15) == 0) && nbr_ext && aux &&
                    (((uintptr_t)aux & 15) == 0);
  const int tiles = (int)(ofx_cdiv(Kp, 128) * ofx_cdiv(cout, 128));
  TnArgs a = {};
  a.Q = dy; a.ldq = ldy; a.Kp = Kp; a.N = cout; a.Kf = Kf;
  if (fast) {
    if (n_multi < 0 || (n_multi > 0 && !multi_seg)) return OFX_EINVAL;
    multi_mean_kernel<<<ofx_grid((n_multi + 1) * (cin / 4), 256), 256, 0, st>>>(x, ldx, cin, seg_ptr, col, multi_seg,
                                                                               n_multi, aux, ldx, nullptr);
    int slices = (int)ofx_cdiv(1024, tiles);
    if (slices > 256) slices = 256;
    while (slices > 1 && (size_t)slices * total * sizeof(float) > ws_bytes) --slices;
    if ((size_t)slices * total * sizeof(float) > ws_bytes) return OFX_EINVAL;
    a.x = x; a.ldx = ldx; a.cin = cin; a.ndir = 7; a.nbr_ext = nbr_ext; a.aux = aux; a.n_src = n_nodes;
    a.tf = type_frac ? type_frac : x; a.ldt = type_frac ? ldt : ldx;
    a.rows = n_nodes; a.row0 = 0;
    a.rows_per_slice = ofx_cdiv(ofx_cdiv(n_nodes, slices), 32) * 32;
    slices = (int)ofx_cdiv(n_nodes, a.rows_per_slice);
    a.part = (float*)ws;
    launch_tn<false>(a, dim3((unsigned)ofx_cdiv(Kp, 128), (unsigned)ofx_cdiv(cout, 128), (unsigned)slices), st);
    tn_reduce_kernel<<<ofx_grid(total, 256), 256, 0, st>>>(a.part, slices, total, dWp, 0);
    OFX_LAUNCH_CHECK();
    return OFX_OK;
  }
  // generic layers: col rows materialised chunk by chunk in the first part of the workspace
  GemmArgs g = {};
  g.x = x; g.ldx = ldx; g.cin = cin; g.ndir = 7; g.nbr = nbr; g.seg_ptr = seg_ptr; g.col = col;
  g.tf = type_frac; g.ldt = ldt; g.Kf = Kf; g.Kp = Kp; g.M = n_nodes;
  int slices = (int)ofx_cdiv(512, tiles);
  if (slices > 64) slices = 64;
  const size_t part_bytes = ((size_t)slices * total * sizeof(float) + 255) & ~size_t(255);
  if (part_bytes >= ws_bytes) return OFX_EINVAL;
  int64_t chunk = (int64_t)((ws_bytes - part_bytes) / ((size_t)Kp * sizeof(float))) / 32 * 32;
  if (chunk < 32) return OFX_EINVAL;
  float* colbuf = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + part_bytes);
  a.P = colbuf; a.ldp = Kp; a.part = (float*)ws;
  int first = 1;
  for (int64_t r0 = 0; r0 < n_nodes; r0 += chunk) {
    const int64_t rows = n_nodes - r0 < chunk ? n_nodes - r0 : chunk;
    col_rows_kernel<<<ofx_grid(rows * (Kp >> 2), 256), 256, 0, st>>>(g, r0, rows, colbuf);
    a.Q = dy + r0 * ldy; a.rows = rows; a.row0 = r0;
    a.rows_per_slice = ofx_cdiv(ofx_cdiv(rows, slices), 32) * 32;
    const int sl = (int)ofx_cdiv(rows, a.rows_per_slice);
    launch_tn<true>(a, dim3((unsigned)ofx_cdiv(Kp, 128), (unsigned)ofx_cdiv(cout, 128), (unsigned)sl), st);
    tn_reduce_kernel<<<ofx_grid(total, 256), 256, 0, st>>>(a.part, sl, total, dWp, first ? 0 : 1);
    first = 0;
  }
  OFX_LAUNCH_CHECK();
  return OFX_OK;
}

// out[K, N] = P^T @ Q for row-major P [rows, K], Q [rows, N] (the weight gradient of every Linear / Conv1x1 /
// pool / unpool layer: dW = x^T dy).  Exact fp32 MFMA, deterministic.  ws: slices * K * N floats of partials.
// Backward of the 27-tap grid convolution (nn.Conv3d 3^3 of the dense lr net in node-row layout).
// _bwd_data: dx [n_in, cin] = the same gather-GEMM over the reverse tap table (summing segments: stride-2 and
//   upsample+conv taps fan in) with W^T per tap (WpT = ofx_pack_conv3d of weight.transpose(0, 1)).
// _bwd_weight: dWp [pad32(27*cin), cout], row k = tap*cin + c, = col^T dy with col gathered through the forward
//   table (padding taps name the zero row).
extern "C" int ofx_gridconv_bwd_data(const float* dy, int64_t ldy, int cout, int64_t n_out, int64_t n_in,
                                     const int32_t* nbr_rev, const int32_t* rev_ptr, const int32_t* rev_row,
                                     const float* rev_w, const int32_t* nbr_ext_rev, const int32_t* multi_seg,
                                     int64_t n_multi, float* aux, const float* WpT, int cin, float* dx, int64_t ldx,
                                     void* ws, size_t ws_bytes, void* stream) {
  if (n_in == 0 && cin >= 1 && cout >= 1) return OFX_OK;
  GemmArgs g = {};
  const int64_t KpT = ofx_conv3d_packed_k(cout);
  int rc = gather_common(g, dy, ldy, cout, 27, n_in, nbr_rev, rev_ptr, rev_row, WpT, KpT, cin, nullptr, nullptr, 0,
                         nullptr, nullptr, 0, dx, ldx);
  if (rc) return rc;
  if (!rev_ptr || !rev_row || !rev_w || n_out < 1) return OFX_EINVAL;
  g.edge_w = rev_w;
  g.tf = dy; g.ldt = ldy;
  hipStream_t st = ofx_stream(stream);
  if (g.fast && nbr_ext_rev && aux && (((uintptr_t)aux & 15) == 0)) {
    if (n_multi < 0 || (n_multi > 0 && !multi_seg)) return OFX_EINVAL;
    multi_mean_kernel<<<ofx_grid((n_multi + 1) * (cout / 4), 256), 256, 0, st>>>(dy, ldy, cout, rev_ptr, rev_row,
                                                                                multi_seg, n_multi, aux, ldy, rev_w);
    g.nbr_ext = nbr_ext_rev; g.aux = aux; g.ldaux = ldy; g.n_src = n_out;
  } else {
    rc = launch_gather_via_col(g, (float*)ws, ws_bytes, st);
    if (rc >= 0) return rc;
  }
  return launch_gemm<MODE_GATHER>(g, (float*)ws, ws_bytes, st);
}

extern "C" int ofx_gridconv_bwd_weight(const float* x, int64_t ldx, int cin, int64_t n_in, int64_t n_out,
                                       const int32_t* nbr27, const int32_t* nbr27_ext, const float* zero_row,
                                       const float* dy, int64_t ldy, int cout, float* dWp, void* ws, size_t ws_bytes,
                                       void* stream) {
  if (n_in < 1 || n_out < 0 || cin < 1 || cout < 1 || (cout & 3) || !x || !dy || !dWp || !ws || ldy < cout ||
      (ldy & 3) || ldx < cin || (((uintptr_t)dy | (uintptr_t)ws) & 15))
    return OFX_EINVAL;
  const int64_t Kp = ofx_conv3d_packed_k(cin), total = Kp * cout;
  hipStream_t st = ofx_stream(stream);
  if (n_out == 0) {
    if (hipMemsetAsync(dWp, 0, (size_t)total * sizeof(float), st) != hipSuccess) return OFX_ELAUNCH;
    return OFX_OK;
  }
  const int tiles = (int)(ofx_cdiv(Kp, 128) * ofx_cdiv(cout, 128));
  TnArgs a = {};
  a.Q = dy; a.ldq = ldy; a.Kp = Kp; a.N = cout; a.Kf = Kp; a.part = (float*)ws;
  const bool fast = (cin % 32 == 0) && ((ldx & 3) == 0) && (((uintptr_t)x & 15) == 0) && nbr27_ext && zero_row &&
                    (((uintptr_t)zero_row & 15) == 0);
  if (fast) {
    int slices = (int)ofx_cdiv(1024, tiles);
    if (slices > 256) slices = 256;
    if (slices > (int)ofx_cdiv(n_out, 32)) slices = (int)ofx_cdiv(n_out, 32);
    while (slices > 1 && (size_t)slices * total * sizeof(float) > ws_bytes) --slices;
    if ((size_t)slices * total * sizeof(float) > ws_bytes) return OFX_EINVAL;
    a.x = x; a.ldx = ldx; a.cin = cin; a.ndir = 27; a.nbr_ext = nbr27_ext; a.aux = zero_row; a.n_src = n_in;
    a.tf = x; a.ldt = ldx; a.rows = n_out; a.row0 = 0;
    a.rows_per_slice = ofx_cdiv(ofx_cdiv(n_out, slices), 32) * 32;
    slices = (int)ofx_cdiv(n_out, a.rows_per_slice);
    launch_tn<false>(a, dim3((unsigned)ofx_cdiv(Kp, 128), (unsigned)ofx_cdiv(cout, 128), (unsigned)slices), st);
    tn_reduce_kernel<<<ofx_grid(total, 256), 256, 0, st>>>(a.part, slices, total, dWp, 0);
    OFX_LAUNCH_CHECK();
    return OFX_OK;
  }
  if (!nbr27) return OFX_EINVAL;
  GemmArgs g = {};
  g.x = x; g.ldx = ldx; g.cin = cin; g.ndir = 27; g.nbr = nbr27; g.tf = x; g.ldt = ldx; g.Kf = Kp; g.Kp = Kp; g.M = n_out;
  int slices = (int)ofx_cdiv(512, tiles);
  if (slices > 64) slices = 64;
  const size_t part_bytes = ((size_t)slices * total * sizeof(float) + 255) & ~size_t(255);
  if (part_bytes >= ws_bytes) return OFX_EINVAL;
  int64_t chunk = (int64_t)((ws_bytes - part_bytes) / ((size_t)Kp * sizeof(float))) / 32 * 32;
  if (chunk < 32) return OFX_EINVAL;
  float* colbuf = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + part_bytes);
  a.P = colbuf; a.ldp = Kp;
  int first = 1;
  for (int64_t r0 = 0; r0 < n_out; r0 += chunk) {
    const int64_t rows = n_out - r0 < chunk ? n_out - r0 : chunk;
    col_rows_kernel<<<ofx_grid(rows * (Kp >> 2), 256), 256, 0, st>>>(g, r0, rows, colbuf);
    a.Q = dy + r0 * ldy; a.rows = rows; a.row0 = r0;
    a.rows_per_slice = ofx_cdiv(ofx_cdiv(rows, slices), 32) * 32;
    const int sl = (int)ofx_cdiv(rows, a.rows_per_slice);
    launch_tn<true>(a, dim3((unsigned)ofx_cdiv(Kp, 128), (unsigned)ofx_cdiv(cout, 128), (unsigned)sl), st);
    tn_reduce_kernel<<<ofx_grid(total, 256), 256, 0, st>>>(a.part, sl, total, dWp, first ? 0 : 1);
    first = 0;
  }
  OFX_LAUNCH_CHECK();
  return OFX_OK;
}

extern "C" int ofx_gemm_tn_f32(const float* P, int64_t ldp, const float* Q, int64_t ldq, int64_t rows, int64_t K,
                               int64_t N, float* out, void* ws, size_t ws_bytes, void* stream) {
  if (rows < 0 || K < 4 || N < 4 || (K & 3) || (N & 3) || !out || !ws || ldp < K || ldq < N || ((ldp | ldq) & 3) ||
      (rows > 0 && (!P || !Q)) || (((uintptr_t)P | (uintptr_t)Q | (uintptr_t)ws) & 15))
    return OFX_EINVAL;
  hipStream_t st = ofx_stream(stream);
  const int64_t total = K * N;
  if (rows == 0) {
    if (hipMemsetAsync(out, 0, (size_t)total * sizeof(float), st) != hipSuccess) return OFX_ELAUNCH;
    return OFX_OK;
  }
  const int tiles = (int)(ofx_cdiv(K, 128) * ofx_cdiv(N, 128));
  int slices = (int)ofx_cdiv(1024, tiles);
  if (slices > 256) slices = 256;
  if (slices > (int)ofx_cdiv(rows, 32)) slices = (int)ofx_cdiv(rows, 32);
  while (slices > 1 && (size_t)slices * total * sizeof(float) > ws_bytes) --slices;
  if ((size_t)slices * total * sizeof(float) > ws_bytes) return OFX_EINVAL;
  TnArgs a = {};
  a.P = P; a.ldp = ldp; a.Q = Q; a.ldq = ldq; a.Kp = K; a.N = N; a.Kf = K; a.rows = rows; a.row0 = 0;
  a.rows_per_slice = ofx_cdiv(ofx_cdiv(rows, slices), 32) * 32;
  slices = (int)ofx_cdiv(rows, a.rows_per_slice);
  a.part = (float*)ws;
  launch_tn<true>(a, dim3((unsigned)ofx_cdiv(K, 128), (unsigned)ofx_cdiv(N, 128), (unsigned)slices), st);
  tn_reduce_kernel<<<ofx_grid(total, 256), 256, 0, st>>>(a.part, slices, total, out, 0);
  OFX_LAUNCH_CHECK();
  return OFX_OK;
}

extern "C" int ofx_gridconv_fwd(const float* x, int64_t ldx, int cin, int64_t n_in, int64_t n_out,
                                const int32_t* nbr27, const int32_t* nbr27_ext, const float* zero_row,
                                const float* Wp, int cout,
                                const float* bias, const float* emb, int64_t lde, const int32_t* batch_id,
                                const float* res, int64_t ldr, float* out, int64_t ldc, void* ws, size_t ws_bytes,
                                void* stream) {
  if (n_out == 0 && cin >= 1 && cout >= 1) return OFX_OK;
  GemmArgs g = {};
  const int64_t Kp = ofx_conv3d_packed_k(cin);
  int rc = gather_common(g, x, ldx, cin, 27, n_out, nbr27 ? nbr27 : nbr27_ext, nullptr, nullptr, Wp, Kp, cout, bias,
                         emb, lde, batch_id, res, ldr, out, ldc);
  if (rc) return rc;
  if (n_in < 1) return OFX_EINVAL;
  if (g.fast && nbr27_ext && zero_row && (((uintptr_t)zero_row & 15) == 0)) {
    g.nbr_ext = nbr27_ext; g.aux = zero_row; g.ldaux = ldx; g.n_src = n_in;   // padded taps name row n_in = zero row
    g.tf = x; g.ldt = ldx;
  } else if (!nbr27) {
    return OFX_EINVAL;            // generic path needs the -1-padded table
  }
  if (!g.nbr_ext) {
    rc = launch_gather_via_col(g, (float*)ws, ws_bytes, ofx_stream(stream));
    if (rc >= 0) return rc;
  }
  return launch_gemm<MODE_GATHER>(g, (float*)ws, ws_bytes, ofx_stream(stream));
}

// out[orow(r), :] = [ x[tab[r, 0], :] | ... | x[tab[r, ntap - 1], :] ] @ W + bias + res[r]: the branch-free gather-GEMM with
// a caller-made table -- Downsample (modules.py:391-395: the eight children of a node are rows 8 r .. 8 r + 7, so
// `x.view(-1, 8 C) @ W` is this with tab[r, j] = 8 r + j) on an x whose rows are NOT contiguous (a column slice of the
// skip-concatenation buffer: ld != C, where the reference's .view() forces a copy).  Kp = pad32(ntap * cin), W packed as
// for ofx_gemm_f32 (k = tap * cin + c); out_rows / out_mode as in ofx_gemm_f32_planes.
extern "C" int ofx_gather_gemm_f32(const float* x, int64_t ldx, int cin, int ntap, int64_t n_src, int64_t n_out,
                                   const int32_t* tab, const float* zero_row, const float* Wp, int64_t Kp, int cout,
                                   const float* bias, const float* res, int64_t ldr, float* out, int64_t ldc,
                                   const int32_t* out_rows, void* ws, size_t ws_bytes, int out_mode, void* stream) {
  if (n_out == 0 && cin >= 1 && cout >= 1) return OFX_OK;
  if (ntap < 1 || ntap > 64 || n_src < 1 || !tab || !zero_row || ((uintptr_t)zero_row & 15) || (cin & 31) || (ldx & 3) ||
      ((uintptr_t)x & 15) || Kp != pad32((int64_t)ntap * cin))
    return OFX_EINVAL;
  if (out_mode != 0 && out_mode != 2 && out_mode != 3) return OFX_EINVAL;
  if (out_mode && ((cout & 3) || (ldc & 31) || ((uintptr_t)out & 127) || (res && ((ldr & 3) || ((uintptr_t)res & 15))) ||
                   (bias && ((uintptr_t)bias & 15)) || ((uintptr_t)ws & 15)))
    return OFX_EINVAL;
  GemmArgs g = {};
  int rc = gather_common(g, x, ldx, cin, ntap, n_out, tab, nullptr, nullptr, Wp, Kp, cout, bias, nullptr, 0, nullptr, res,
                         ldr, out, ldc);
  if (rc) return rc;
  if (!g.fast) return OFX_EINVAL;
  g.nbr_ext = tab; g.aux = zero_row; g.ldaux = ldx; g.n_src = n_src;      // entries == n_src name the zero row
  g.tf = x; g.ldt = ldx;
  g.out_rows = out_rows; g.gather_out_rows = 1; g.out_planes = out_mode;
  return launch_gemm<MODE_GATHER>(g, (float*)ws, ws_bytes, ofx_stream(stream));
}

// ---------------------------------------------------------------------------------
// Stand-alone segment-mean gather (HBM-bound): col_data[r, dir, 0:cin].
// One float4 per lane; lanes sweep channels fastest so a row's 4*cin bytes are read and
// written as whole 128-B lines.
__global__ void __launch_bounds__(256) gather_mean_kernel(const float* __restrict__ x, int64_t ldx, int cin, int64_t nseg,
                                                          const int32_t* __restrict__ seg_ptr,
                                                          const int32_t* __restrict__ col, float* __restrict__ out) {
  const int c4n = cin >> 2;
  const int64_t total = nseg * c4n;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t s = t / c4n;
    const int c = (int)(t - s * c4n) * 4;
    const int32_t a = seg_ptr[s], e = seg_ptr[s + 1];
    float4 acc = f4zero();
    for (int32_t p = a; p < e; ++p) f4add(acc, *reinterpret_cast<const float4*>(x + (int64_t)col[p] * ldx + c));
    if (e - a > 1) {
      const float inv = 1.f / (float)(e - a);
      acc.x *= inv; acc.y *= inv; acc.z *= inv; acc.w *= inv;
    }
    *reinterpret_cast<float4*>(out + s * cin + c) = acc;
  }
}
__global__ void __launch_bounds__(256) gather_mean_scalar_kernel(const float* __restrict__ x, int64_t ldx, int cin,
                                                                 int64_t nseg, const int32_t* __restrict__ seg_ptr,
                                                                 const int32_t* __restrict__ col,
                                                                 float* __restrict__ out) {
  const int64_t total = nseg * cin;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t s = t / cin;
    const int c = (int)(t - s * cin);
    const int32_t a = seg_ptr[s], e = seg_ptr[s + 1];
    float acc = 0.f;
    for (int32_t p = a; p < e; ++p) acc += x[(int64_t)col[p] * ldx + c];
    if (e - a > 1) acc /= (float)(e - a);
    out[t] = acc;
  }
}

extern "C" int ofx_gather_mean(const float* x, int64_t ldx, int cin, int64_t n_nodes, const int32_t* seg_ptr,
                               const int32_t* col, float* col_data, void* stream) {
  if (!x || !seg_ptr || !col || !col_data || cin < 1 || ldx < cin || n_nodes < 0) return OFX_EINVAL;
  const int64_t nseg = n_nodes * 7;
  hipStream_t st = ofx_stream(stream);
  const bool vec = (cin % 4 == 0) && ((ldx & 3) == 0) && (((uintptr_t)x & 15) == 0) && (((uintptr_t)col_data & 15) == 0);
  if (vec)
    gather_mean_kernel<<<ofx_grid(nseg * (cin / 4), 256), 256, 0, st>>>(x, ldx, cin, nseg, seg_ptr, col, col_data);
  else
    gather_mean_scalar_kernel<<<ofx_grid(nseg * cin, 256), 256, 0, st>>>(x, ldx, cin, nseg, seg_ptr, col, col_data);
  OFX_LAUNCH_CHECK();
  return OFX_OK;
}

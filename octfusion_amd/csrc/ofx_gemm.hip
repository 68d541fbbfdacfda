// libofx: fp32-MFMA contraction core, dense GEMM and the fused dual-octree GraphConv.
//
// One kernel template, two A-tile loaders:
//   MODE_DENSE  : A[arow(m), k] row-major (optional row map)         -> ofx_gemm_f32
//   MODE_GATHER : A[m, dir*cin + c] = mean_{e in seg(m,dir)} x[col[e], c], followed by
//                 the dense node-type-fraction slab                   -> ofx_graphconv_fwd
// The gathered [N, 7*cin] "col_data" of the reference (modules.py:208-210) never
// exists in HBM: neighbour rows are fetched (16 B per lane, one 128-B line per
// 8 lanes) straight into the LDS A-tile.
//
// Tiling (wave = 64): block = 4 waves, BM = 128 rows, BK = 32; BN = 128 (2x2 waves of
// 64x64) or BN = 32 (4x1 waves of 32x32) for narrow outputs.  Matrix core:
// v_mfma_f32_32x32x2_f32 -- exact fp32 (k-ordered fma chain), 157 TF peak on gfx950;
// parity with the fp32 reference is by construction, no reduced precision anywhere.
// LDS: A tile [128][32+4] fp32 (pad 4 -> conflict-free ds_read_b128 over 16-lane
// groups), B tile [8][BN][4] fp32 read as one ds_read_b128 per 4 k-steps; both double
// buffered, one barrier per k-tile.  Weights are pre-packed once (ofx_pack_weights) to
// [k/4][n][4] so the B tile is a straight 16-B-per-lane copy.
#include "ofx_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128;
constexpr int BK = 32;
constexpr int A_LD = BK + 4;
constexpr int MODE_DENSE = 0;
constexpr int MODE_GATHER = 1;

struct GemmArgs {
  // A (dense)
  const float* A; int64_t lda; const int32_t* a_rows;
  // A (gather)
  const float* x; int64_t ldx; int cin; int fast;   // fast: cin % 32 == 0 and aligned
  const int32_t* seg_ptr; const int32_t* col;
  const float* tf; int64_t ldt; int64_t Kf;          // Kf = pad32(7*cin)
  // common
  int64_t M, K;            // K: logical K for dense bounds; gather uses Kp only
  const float* Wp; int64_t Kp, N;
  const float* bias;
  const float* emb; int64_t lde; const int32_t* bid;
  const float* res; int64_t ldr;
  float* out; int64_t ldc; const int32_t* out_rows;
  int ntm, ntn;
};

__device__ __forceinline__ float4 f4zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ void f4add(float4& a, const float4& b) { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }

// scalar (slow-path) gathered element: mean over segment (row, dir) of x[col, c]
__device__ __forceinline__ float gather_elem(const GemmArgs& g, int64_t row, int64_t k) {
  if (k >= 7 * (int64_t)g.cin) return 0.f;
  const int dir = (int)(k / g.cin), c = (int)(k - (int64_t)dir * g.cin);
  const int32_t s = g.seg_ptr[row * 7 + dir], e = g.seg_ptr[row * 7 + dir + 1];
  float acc = 0.f;
  for (int32_t p = s; p < e; ++p) acc += g.x[(int64_t)g.col[p] * g.ldx + c];
  if (e - s > 1) acc /= (float)(e - s);
  return acc;
}

template <int MODE>
__device__ __forceinline__ void load_a_tile(const GemmArgs& g, int64_t m0, int64_t k0, float4 (&va)[4]) {
  const int c4 = threadIdx.x & 7, r0 = threadIdx.x >> 3;
  if (MODE == MODE_DENSE) {
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
  } else {
    if (k0 >= g.Kf) {                                  // node-type fraction slab (dense, zero padded)
      const int64_t kt = k0 - g.Kf + c4 * 4;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int64_t m = m0 + r0 + 32 * i;
        va[i] = (m < g.M) ? *reinterpret_cast<const float4*>(g.tf + m * g.ldt + kt) : f4zero();
      }
    } else if (g.fast) {
      const int dir = (int)(k0 / g.cin);
      const int cc = (int)(k0 - (int64_t)dir * g.cin) + c4 * 4;
      int32_t s[4], e[4], c[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int64_t m = m0 + r0 + 32 * i;
        s[i] = 0; e[i] = 0;
        if (m < g.M) { s[i] = g.seg_ptr[m * 7 + dir]; e[i] = g.seg_ptr[m * 7 + dir + 1]; }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) c[i] = e[i] > s[i] ? g.col[s[i]] : -1;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        va[i] = c[i] >= 0 ? *reinterpret_cast<const float4*>(g.x + (int64_t)c[i] * g.ldx + cc) : f4zero();
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (e[i] - s[i] > 1) {                           // coarse leaf touching several finer nodes (rare)
          for (int32_t p = s[i] + 1; p < e[i]; ++p)
            f4add(va[i], *reinterpret_cast<const float4*>(g.x + (int64_t)g.col[p] * g.ldx + cc));
          const float inv = 1.f / (float)(e[i] - s[i]);
          va[i].x *= inv; va[i].y *= inv; va[i].z *= inv; va[i].w *= inv;
        }
      }
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
  const int nblk = g.ntm * g.ntn;
  int bid = blockIdx.x;
  {
    const int q = nblk / 8, r = nblk % 8, xcd = bid % 8, j = bid / 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
  }
  const int tm = bid / g.ntn, tn = bid - tm * g.ntn;
  const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;

  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int wm = wid / WN, wn = wid % WN;
  const int l31 = lane & 31, h = lane >> 5;

  f32x16 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float4 va[4];
  float4 vb[BN / 32];
  const int nkt = (int)(g.Kp / BK);
  const int c4 = threadIdx.x & 7, r0 = threadIdx.x >> 3;

  auto store_tiles = [&](int buf) {
    float* a = As + buf * BM * A_LD;
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<float4*>(a + (r0 + 32 * i) * A_LD + c4 * 4) = va[i];
    float* b = Bs + buf * 8 * BN * 4;
#pragma unroll
    for (int i = 0; i < BN / 32; ++i) *reinterpret_cast<float4*>(b + (threadIdx.x + 256 * i) * 4) = vb[i];
  };

  load_a_tile<MODE>(g, m0, 0, va);
  load_b_tile<BN>(g, n0, 0, vb);
  store_tiles(0);
  __syncthreads();

  for (int kt = 0; kt < nkt; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nkt) {
      load_a_tile<MODE>(g, m0, (int64_t)(kt + 1) * BK, va);
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
    if (kt + 1 < nkt) store_tiles(buf ^ 1);
    __syncthreads();
  }

  // epilogue: C/D layout of 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
  for (int i = 0; i < MI; ++i) {
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int64_t n = n0 + (wn * NI + j) * 32 + l31;
      if (n >= g.N) continue;
      const float bv = g.bias ? g.bias[n] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t m = m0 + (wm * MI + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (m >= g.M) continue;
        float v = acc[i][j][r] + bv;
        if (g.emb) v += g.emb[(int64_t)g.bid[m] * g.lde + n];
        if (g.res) v += g.res[m * g.ldr + n];
        int64_t om = m;
        if (g.out_rows) { om = g.out_rows[m]; if (om < 0) continue; }
        g.out[om * g.ldc + n] = v;
      }
    }
  }
}

template <int MODE>
static int launch_gemm(GemmArgs& g, hipStream_t st) {
  g.ntm = (int)ofx_cdiv(g.M, BM);
  if (g.M <= 0 || g.N <= 0) return OFX_OK;
  if (g.N <= 32) {
    g.ntn = (int)ofx_cdiv(g.N, 32);
    constexpr size_t lds = (2 * BM * A_LD + 2 * 8 * 32 * 4) * sizeof(float);
    gemm_kernel<MODE, 4, 1, 1, 1><<<g.ntm * g.ntn, 256, lds, st>>>(g);
  } else {
    g.ntn = (int)ofx_cdiv(g.N, 128);
    constexpr size_t lds = (2 * BM * A_LD + 2 * 8 * 128 * 4) * sizeof(float);   // 68 KB > 64 KB default cap
    static bool attr_set = false;
    if (!attr_set) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kernel<MODE, 2, 2, 2, 2>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return OFX_ELAUNCH;
      attr_set = true;
    }
    gemm_kernel<MODE, 2, 2, 2, 2><<<g.ntm * g.ntn, 256, lds, st>>>(g);
  }
  OFX_LAUNCH_CHECK();
  return OFX_OK;
}

static inline int64_t pad32(int64_t v) { return (v + 31) / 32 * 32; }

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
  OFX_LAUNCH_CHECK();
  return OFX_OK;
}

extern "C" int ofx_gemm_f32(const float* A, int64_t lda, const int32_t* a_rows, int64_t M, int64_t K, const float* Wp,
                            int64_t Kp, int64_t N, const float* bias, const float* res, int64_t ldr, float* out,
                            int64_t ldc, const int32_t* out_rows, void* stream) {
  if (M < 0 || K < 1 || N < 1 || !Wp || !out || (M > 0 && !A) || Kp != pad32(K) || lda < K || ldc < N ||
      (res && ldr < N) || ((uintptr_t)Wp & 15))
    return OFX_EINVAL;
  GemmArgs g = {};
  g.A = A; g.lda = lda; g.a_rows = a_rows;
  g.M = M; g.K = K; g.Wp = Wp; g.Kp = Kp; g.N = N; g.bias = bias;
  g.res = res; g.ldr = ldr; g.out = out; g.ldc = ldc; g.out_rows = out_rows;
  return launch_gemm<MODE_DENSE>(g, ofx_stream(stream));
}

extern "C" int ofx_graphconv_fwd(const float* x, int64_t ldx, int cin, int64_t n_nodes, const int32_t* seg_ptr,
                                 const int32_t* col, const float* type_frac, int64_t ldt, int nt_pad, const float* Wp,
                                 int64_t Kp, int cout, const float* bias, const float* emb, int64_t lde,
                                 const int32_t* batch_id, const float* res, int64_t ldr, float* out, int64_t ldc,
                                 void* stream) {
  if (n_nodes < 0 || cin < 1 || cout < 1 || !x || !seg_ptr || !col || !Wp || !out || ldx < cin || ldc < cout ||
      (res && ldr < cout) || (emb && (!batch_id || lde < cout)) || ((uintptr_t)Wp & 15))
    return OFX_EINVAL;
  const int64_t Kf = pad32(7 * (int64_t)cin);
  if (nt_pad < 0 || (nt_pad & 31) || Kp != Kf + nt_pad) return OFX_EINVAL;
  if (nt_pad > 0 && (!type_frac || ldt < nt_pad || (ldt & 3) || ((uintptr_t)type_frac & 15))) return OFX_EINVAL;
  GemmArgs g = {};
  g.x = x; g.ldx = ldx; g.cin = cin;
  g.fast = (cin % 32 == 0) && ((ldx & 3) == 0) && (((uintptr_t)x & 15) == 0);
  g.seg_ptr = seg_ptr; g.col = col; g.tf = type_frac; g.ldt = ldt; g.Kf = Kf;
  g.M = n_nodes; g.K = Kp; g.Wp = Wp; g.Kp = Kp; g.N = cout; g.bias = bias;
  g.emb = emb; g.lde = lde; g.bid = batch_id; g.res = res; g.ldr = ldr; g.out = out; g.ldc = ldc;
  return launch_gemm<MODE_GATHER>(g, ofx_stream(stream));
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

// Shared by the contraction kernels (ofx_gemm.hip: 128-row tiles, register-staged; ofx_gemm2.hip:
// 256-row tiles, LDS-DMA staged): launch arguments and the fused epilogue.
#pragma once
#include "ofx_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct GemmArgs {
  // A (dense)
  const float* A; int64_t lda; const int32_t* a_rows;
  // A (gather)
  const float* x; int64_t ldx; int cin; int ndir; int fast;   // fast: cin % 32 == 0 and aligned
  const int32_t* nbr;                                          // [M, ndir]: >=0 row, -1 none, -2 see CSR
  const int32_t* seg_ptr; const int32_t* col;                  // CSR by segment m*ndir+dir (for -2 entries)
  const float* edge_w;                                         // NULL: segment MEAN; else per-edge weights, segment weighted SUM
  const float* tf; int64_t ldt; int64_t Kf;                    // Kf = pad32(ndir*cin)
  const int32_t* nbr_ext;                                      // fast path: [M, ndir] in [0, n_src + 1 + V)
  const float* aux; int64_t ldaux; int64_t n_src;              // aux row 0 = zeros, rows 1.. = multi-neighbour means
  // common
  int64_t M, K;            // K: logical K for dense bounds; gather uses Kp only
  const float* Wp; int64_t Kp, N;
  const uint16_t* W16;     // bf16 hi | lo split of the packed weights ([k/8][n][8] each), or NULL
  const float* oscale_p;   // device float: the accumulators are multiplied by it before the epilogue terms (the inverse of
                           // the power-of-two scale the 16-bit weight halves were packed with); NULL = 1
  const float* bias;
  const float* emb; int64_t lde; const int32_t* bid;
  const float* res; int64_t ldr;
  float* out; int64_t ldc; const int32_t* out_rows;
  int gather_out_rows;     // gather mode: the caller vouches for out_rows on the branch-free kernel (ofx_gather_gemm_f32)
  int out_planes;          // 0: fp32 rows; 2 / 3: write `out` as bf16 / fp16 hi + lo pair planes (vec4 epilogues only; needs
                           // a 128-B aligned `out` and ldc % 32 == 0): the consumer is the planes GraphConv
  double* stats; int64_t stats_ld;   // optional fused GroupNorm statistics: stats[(b*stats_ld + n)*2 + {0,1}] += (v, v*v)
  int ntm, ntn, nsplit, kt_per_split;
  float* ws;               // split-K partials [nsplit][M][N]
  int vec4;                // out / res / emb / bias are 16-B aligned with pitches % 4 == 0 and N % 4 == 0
  // fused statistics, two-stage: every wave stores the (sum, sum of squares) of its 32*MI rows to
  // stats_part[wave_row][N][2] (plain fp32 stores), stats_reduce_kernel adds them up per batch element.
  // NULL -> fp64 atomics straight from the epilogue (870 k device-scope atomics per depth-6 launch, measured
  // at 75 us of a 300 us kernel).
  float* stats_part; size_t stats_part_bytes;
};

// scale[0] = 1/s, scale[1] = s with s = 2^e such that max|w| * s lies in [2^14, 2^15) (fp16 operands: modes 1 and 3;
// bf16 pairs have fp32's exponent range: s = 1).  Pack time only.  Grid-wide max|w| by atomicMax on the float's bit
// pattern (order-preserving for non-negative floats) into scale[0], then one thread turns it into the two scales -- until
// round 5 this was ONE block walking the whole tensor: up to 1 ms per tensor, 15 ms of a process's first step.
static __global__ void __launch_bounds__(256) weight_absmax_kernel(const float* __restrict__ W, int64_t sk, int64_t sn, int64_t K,
                                                            int64_t N, unsigned* __restrict__ acc) {
  __shared__ float red[256];
  float m = 0.f;
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < K * N; t += (int64_t)gridDim.x * 256) {
    const float a = fabsf(W[(t / N) * sk + (t % N) * sn]);
    if (a <= 3.0e38f) m = fmaxf(m, a);                  // (Inf / NaN weights: left to the arithmetic, not to the scale)
  }
  red[threadIdx.x] = m;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + o]);
    __syncthreads();
  }
  if (threadIdx.x == 0) atomicMax(acc, __float_as_uint(red[0]));
}
static __global__ void weight_scale_finish_kernel(int fp16_operands, float* __restrict__ scale) {
  const float mx = __uint_as_float(reinterpret_cast<const unsigned*>(scale)[0]);
  float s = 1.f;
  if (fp16_operands && mx > 0.f) {
    int e;
    frexpf(mx, &e);                                     // mx = f * 2^e, f in [0.5, 1)
    int sh = 15 - e;                                    // max|w| * 2^sh in [2^14, 2^15)
    sh = sh < -60 ? -60 : (sh > 60 ? 60 : sh);
    s = ldexpf(1.f, sh);
  }
  scale[0] = 1.f / s;
  scale[1] = s;
}
static inline int ofx_launch_weight_scale(const float* W, int64_t sk, int64_t sn, int64_t K, int64_t N, int fp16_operands,
                                          float* scale, hipStream_t st) {
  if (hipMemsetAsync(scale, 0, sizeof(float), st) != hipSuccess) return OFX_ELAUNCH;
  int64_t nb = (K * N + 256 * 16 - 1) / (256 * 16);
  nb = nb < 1 ? 1 : (nb > 1024 ? 1024 : nb);
  weight_absmax_kernel<<<(unsigned)nb, 256, 0, st>>>(W, sk, sn, K, N, reinterpret_cast<unsigned*>(scale));
  if (hipGetLastError() != hipSuccess) return OFX_ELAUNCH;
  weight_scale_finish_kernel<<<1, 1, 0, st>>>(fp16_operands, scale);
  // (a failed scale launch would leave a garbage trailer that every later fp16x3 epilogue multiplies by: ADVICE r05)
  if (hipGetLastError() != hipSuccess) return OFX_ELAUNCH;
  return OFX_OK;
}


// ---- 16-bit operand pairs (formats: ofx_planes.h)
__device__ __forceinline__ unsigned g2_pk_bf16(float a, float b) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ unsigned g2_pk_f16(float a, float b) {
  const _Float16 x = (_Float16)a, y = (_Float16)b;
  return (unsigned)__builtin_bit_cast(unsigned short, x) | ((unsigned)__builtin_bit_cast(unsigned short, y) << 16);
}
__device__ __forceinline__ float g2_bf16_lo(unsigned p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float g2_bf16_hi(unsigned p) { return __uint_as_float(p & 0xffff0000u); }
__device__ __forceinline__ float g2_f16_lo(unsigned p) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(p & 0xffffu)); }
__device__ __forceinline__ float g2_f16_hi(unsigned p) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(p >> 16)); }
__device__ __forceinline__ float g2_sat16(float v) { return fminf(fmaxf(v, -65504.f), 65504.f); }
// (a, b) -> packed hi pair + packed lo pair
__device__ __forceinline__ void g2_split2(int mode, float a, float b, unsigned& hi, unsigned& lo) {
  if (mode == 3) {
    hi = g2_pk_f16(a, b);
    lo = g2_pk_f16(a - g2_f16_lo(hi), b - g2_f16_hi(hi));
  } else {
    hi = g2_pk_bf16(a, b);
    lo = g2_pk_bf16(a - g2_bf16_lo(hi), b - g2_bf16_hi(hi));
  }
}
__device__ __forceinline__ void g2_join2(int mode, unsigned hi, unsigned lo, float& a, float& b) {
  if (mode == 3) { a = g2_f16_lo(hi) + g2_f16_lo(lo); b = g2_f16_hi(hi) + g2_f16_hi(lo); }
  else { a = g2_bf16_lo(hi) + g2_bf16_lo(lo); b = g2_bf16_hi(hi) + g2_bf16_hi(lo); }
}
__host__ __device__ static inline bool g2_pairs(int mode) { return mode == 2 || mode == 3; }   // [hi x 32 | lo x 32] lines

// Store four consecutive floats as their hi / lo pair planes.  `f` = flat float index (a multiple of 4) into a 128-B
// aligned buffer whose row pitch is a multiple of 32 floats: a 32-float group is one 128-B line [hi x 32 | lo x 32]
// whatever the row structure, so the planes of an fp32-shaped buffer are a function of the flat index alone.
__device__ __forceinline__ void ofx_store_planes4(float* base, int64_t f, const float4& v, int mode) {
  unsigned h0, h1, l0, l1;
  g2_split2(mode, v.x, v.y, h0, l0);
  g2_split2(mode, v.z, v.w, h1, l1);
  char* o = reinterpret_cast<char*>(base) + (f >> 5) * 128 + (f & 31) * 2;
  *reinterpret_cast<uint2*>(o) = make_uint2(h0, h1);
  *reinterpret_cast<uint2*>(o + 64) = make_uint2(l0, l1);
}

__device__ __forceinline__ float4 f4zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ void f4add(float4& a, const float4& b) { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }

// Epilogue shared by the MFMA kernels.  C/D layout of the 32x32 MFMA: col = lane&31,
// row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).  out = acc + bias + emb[bid[m]] + res[m]; optionally also
// accumulates the per-(batch element, channel) sum / sum of squares of the stored values in fp64
// (hardware global_atomic_add_f64), so the DualOctreeGroupNorm that consumes this tensor needs no
// statistics pass of its own (reference modules.py:299-311 makes three scatter passes).
template <int WM, int WN, int MI, int NI>
__device__ __forceinline__ void epilogue_store_scalar(const GemmArgs& g, f32x16 (&acc)[MI][NI], int64_t m0, int64_t n0,
                                                      int wm, int wn, int l31, int h, int split) {
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int64_t n = n0 + (wn * NI + j) * 32 + l31;
    if (n >= g.N) continue;
    if (g.nsplit > 1) {
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int64_t m = m0 + (wm * MI + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          if (m < g.M) g.ws[((int64_t)split * g.M + m) * g.N + n] = acc[i][j][r];
        }
      continue;
    }
    const float bv = g.bias ? g.bias[n] : 0.f;
    const float osc = g.oscale_p ? *g.oscale_p : 1.f;
    int sb = -1;                 // statistics run: batch id, sum, sum of squares
    float ssum = 0.f, ssq = 0.f;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t m = m0 + (wm * MI + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (m >= g.M) continue;
        float v = fmaf(acc[i][j][r], osc, bv);
        int b = 0;
        if (g.emb || g.stats) b = g.bid[m];
        if (g.emb) v += g.emb[(int64_t)b * g.lde + n];
        if (g.res) v += g.res[m * g.ldr + n];
        if (g.stats) {
          if (b != sb) {
            if (sb >= 0) {
              double* o = g.stats + ((int64_t)sb * g.stats_ld + n) * 2;
              unsafeAtomicAdd(o, (double)ssum);
              unsafeAtomicAdd(o + 1, (double)ssq);
            }
            sb = b; ssum = 0.f; ssq = 0.f;
          }
          ssum += v; ssq += v * v;
        }
        int64_t om = m;
        if (g.out_rows) { om = g.out_rows[m]; if (om < 0) continue; }
        g.out[om * g.ldc + n] = v;
      }
    }
    if (g.stats && sb >= 0) {
      double* o = g.stats + ((int64_t)sb * g.stats_ld + n) * 2;
      unsafeAtomicAdd(o, (double)ssum);
      unsafeAtomicAdd(o + 1, (double)ssq);
    }
  }
}

// 4 x 4 transpose across the four lanes of a quad (DPP quad_perm, no LDS): on entry lane q holds
// v[r] = T[r][q], on exit v[c] = T[q][c].
__device__ __forceinline__ float dpp_xor1(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));
}
__device__ __forceinline__ float dpp_xor2(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));
}
__device__ __forceinline__ void quad_transpose(float& v0, float& v1, float& v2, float& v3, bool q0, bool q1) {
  float r = dpp_xor1(q0 ? v0 : v1);
  if (q0) v0 = r; else v1 = r;
  r = dpp_xor1(q0 ? v2 : v3);
  if (q0) v2 = r; else v3 = r;
  r = dpp_xor2(q1 ? v0 : v2);
  if (q1) v0 = r; else v2 = r;
  r = dpp_xor2(q1 ? v1 : v3);
  if (q1) v1 = r; else v3 = r;
}

// Vectorised epilogue.  The 32x32 MFMA leaves each lane with 4 consecutive ROWS of one column per register
// group; a quad transpose turns that into 4 consecutive COLUMNS of one row, so every global access is a
// 16-B piece of a 128-B row segment (8 rows x 128 B per wave instruction instead of 2 rows x 128 B of
// dwords -- measured 15-25 % of the kernel time on the depth-6 layers with the scalar stores).
// Lane (k = l31 >> 2, q = l31 & 3, h) owns rows q + 4h + 8G + 32i (G < 4, i < MI) and columns 4k .. 4k+3.
template <int WM, int WN, int MI, int NI>
__device__ __forceinline__ void epilogue_store_v4(const GemmArgs& g, f32x16 (&acc)[MI][NI], int64_t m0, int64_t n0,
                                                  int wm, int wn, int l31, int h) {
  const int q = l31 & 3, k = l31 >> 2;
  const bool q0 = q & 1, q1 = q & 2;
  const int64_t mw = m0 + wm * MI * 32;                     // first row of this wave
  const int64_t blockIdx_tm = m0 / (WM * MI * 32);          // row-tile index of the block
  const float osc = g.oscale_p ? *g.oscale_p : 1.f;
  // batch ids of this lane's rows; wave-uniform batch id -> statistics are reduced across the wave first
  int bids[MI][4];
  bool uni = true;
  int b0 = 0;
  if (g.emb || g.stats) {
    const int64_t mlast = g.M - 1;
    b0 = g.bid[mw < mlast ? mw : mlast];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int G = 0; G < 4; ++G) {
        const int64_t m = mw + i * 32 + q + 4 * h + 8 * G;
        bids[i][G] = g.bid[m < mlast ? m : mlast];
        uni = uni && (bids[i][G] == b0);
      }
    uni = __all(uni);
  }
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int64_t n = n0 + (wn * NI + j) * 32 + 4 * k;
    const bool ncol = n < g.N;                              // N % 4 == 0: the whole float4 is in or out
    float4 bv = f4zero();
    if (g.bias && ncol) bv = *reinterpret_cast<const float4*>(g.bias + n);
    float4 ssum = f4zero(), ssq = f4zero();
    int sb = -1;
    auto flush = [&](int b) {
      double* o = g.stats + ((int64_t)b * g.stats_ld + n) * 2;
      unsafeAtomicAdd(o + 0, (double)ssum.x); unsafeAtomicAdd(o + 1, (double)ssq.x);
      unsafeAtomicAdd(o + 2, (double)ssum.y); unsafeAtomicAdd(o + 3, (double)ssq.y);
      unsafeAtomicAdd(o + 4, (double)ssum.z); unsafeAtomicAdd(o + 5, (double)ssq.z);
      unsafeAtomicAdd(o + 6, (double)ssum.w); unsafeAtomicAdd(o + 7, (double)ssq.w);
    };
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      float4 t[4];
#pragma unroll
      for (int G = 0; G < 4; ++G) {                         // every lane takes part in the transposes
        float v0 = acc[i][j][4 * G], v1 = acc[i][j][4 * G + 1], v2 = acc[i][j][4 * G + 2], v3 = acc[i][j][4 * G + 3];
        quad_transpose(v0, v1, v2, v3, q0, q1);
        t[G] = make_float4(fmaf(v0, osc, bv.x), fmaf(v1, osc, bv.y), fmaf(v2, osc, bv.z), fmaf(v3, osc, bv.w));
      }
#pragma unroll
      for (int G = 0; G < 4; ++G) {
        const int64_t m = mw + i * 32 + q + 4 * h + 8 * G;
        if (m >= g.M || !ncol) continue;
        float4 v = t[G];
        if (g.emb) f4add(v, *reinterpret_cast<const float4*>(g.emb + (int64_t)bids[i][G] * g.lde + n));
        if (g.res) f4add(v, *reinterpret_cast<const float4*>(g.res + m * g.ldr + n));
        if (g.stats) {
          const int b = bids[i][G];
          if (!uni && b != sb) {
            if (sb >= 0) flush(sb);
            ssum = f4zero(); ssq = f4zero();
          }
          sb = b;
          f4add(ssum, v);
          ssq.x += v.x * v.x; ssq.y += v.y * v.y; ssq.z += v.z * v.z; ssq.w += v.w * v.w;
        }
        int64_t om = m;
        if (g.out_rows) { om = g.out_rows[m]; if (om < 0) continue; }
        if (g.out_planes) ofx_store_planes4(g.out, om * g.ldc + n, v, g.out_planes);
        else *reinterpret_cast<float4*>(g.out + om * g.ldc + n) = v;
      }
    }
    if (g.stats) {
      if (uni) {
        // one batch element in this wave's 32*MI rows: add up the 8 lanes (q, h) that share these columns
        // (lanes without a valid row hold zeros)
#define OFX_RED(f) f += dpp_xor1(f); f += dpp_xor2(f); f += __shfl_xor(f, 32);
        OFX_RED(ssum.x) OFX_RED(ssum.y) OFX_RED(ssum.z) OFX_RED(ssum.w)
        OFX_RED(ssq.x) OFX_RED(ssq.y) OFX_RED(ssq.z) OFX_RED(ssq.w)
#undef OFX_RED
        if (q == 0 && h == 0 && ncol && mw < g.M) {
          if (g.stats_part) {
            float* o = g.stats_part + (((int64_t)blockIdx_tm * WM + wm) * g.N + n) * 2;
            *reinterpret_cast<float4*>(o) = make_float4(ssum.x, ssq.x, ssum.y, ssq.y);
            *reinterpret_cast<float4*>(o + 4) = make_float4(ssum.z, ssq.z, ssum.w, ssq.w);
          } else {
            flush(b0);
          }
        }
      } else {
        if (sb >= 0 && ncol) flush(sb);
        if (g.stats_part && q == 0 && h == 0 && ncol && mw < g.M) {      // mixed wave: its slot must read as zero
          float* o = g.stats_part + (((int64_t)blockIdx_tm * WM + wm) * g.N + n) * 2;
          *reinterpret_cast<float4*>(o) = f4zero();
          *reinterpret_cast<float4*>(o + 4) = f4zero();
        }
      }
    }
  }
}

template <int WM, int WN, int MI, int NI>
__device__ __forceinline__ void epilogue_store(const GemmArgs& g, f32x16 (&acc)[MI][NI], int64_t m0, int64_t n0, int wm,
                                               int wn, int l31, int h, int split) {
  if (g.vec4 && g.nsplit == 1) epilogue_store_v4<WM, WN, MI, NI>(g, acc, m0, n0, wm, wn, l31, h);
  else epilogue_store_scalar<WM, WN, MI, NI>(g, acc, m0, n0, wm, wn, l31, h, split);
}

// libofx: fused dual-octree GraphConv on PRE-SPLIT operand planes, staged by LDS-DMA.
//
// Why a second contraction kernel.  The register-staged 128 x 128 kernel (ofx_gemm.hip) moves, per 96 MFMAs,
// 16 KB of gathered fp32 rows + 32 KB of weight fragments from L2 through the CU's vector-memory path, splits
// the rows to bf16 in VALU and writes them to LDS with ds_write: it sits at the balance point of the 64 B/clk
// L2->CU path, the LDS write path and the matrix pipe (~0.30 of the bf16x3 roof).  This
// kernel spends fewer operand bytes per MFMA and no VALU / ds_write / staging registers at all:
//   * activations arrive already split: the producer (ofx_gn_apply_planes, ofx_planes_split) writes, for every
//     32-channel chunk of a row, one 128-B line [hi k0..31 | lo k0..31] (bf16 pair, a = hi + lo to 2^-17) --
//     the same bytes as the fp32 row, so the planes alias fp32-shaped buffers (zero-copy concat still works);
//     in the single-pass fp16 mode a line is 64 fp16 channels;
//   * weights are packed once as [k tile][column][128 B] lines of the same shape;
//   * a block is 256 rows x 128 columns, 8 waves (4 x 2 wave tiles of 64 x 64), ONE block per CU; per k-step it
//     stages 256 A lines (gathered through the neighbour table) + 128 B lines = 48 KB with
//     global_load_lds_dwordx4 (global -> LDS, no registers), i.e. half the bytes per MFMA of the old kernel;
//   * the LDS image is lane-linear per DMA instruction (hardware rule), so the 16-B piece a lane fetches is
//     XOR-swizzled on the SOURCE side (piece ^= (row >> 1) & 7): the MFMA fragment reads (ds_read_b128, 16
//     different rows per lane group) are then bank-conflict free;
//   * three LDS stage buffers, DMA two k-steps ahead, counted vmcnt, one raw s_barrier per k-step placed in the
//     MIDDLE of the step's MFMAs; the fragment registers are double-buffered by half steps so the LDS reads of
//     the next half always overlap the current half's MFMAs;
//   * the neighbour-table slice of the tile lives in LDS (an ordinary global load in the loop would make
//     hipcc drain the DMA queue at its use).
// Contraction: PREC 2 = bf16x3 (a_lo*w_hi + a_hi*w_lo + a_hi*w_hi on v_mfma_f32_32x32x16_bf16, fp32
// accumulate: same arithmetic as ofx_gemm.hip's default), PREC 1 = one v_mfma_f32_32x32x16_f16 per product
// (operands rounded to fp16: ~5e-4 per product, reduced-precision mode for BASELINE configs[4]).
// The epilogue (bias / time-embedding / residual / fused GroupNorm statistics) is shared with ofx_gemm.hip.
#include <type_traits>
#include <utility>

#include "ofx_gemm_common.h"

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef const char __attribute__((address_space(1)))* gcp;
typedef __attribute__((address_space(3))) void* ldsp;

// Two block geometries (template parameter WM = wave rows):
//   WM = 4: 256 x 128 tile, 8 waves, 3 stage buffers (DMA two k-steps ahead), 152 KB LDS -> ONE block per CU;
//   WM = 2: 128 x 128 tile, 4 waves, 2 stage buffers (DMA one k-step ahead),  68 KB LDS -> TWO blocks per CU: the
//           prologue (table + first DMA), the epilogue (residual reads, stores) and every barrier wait of one
//           block overlap the other block's MFMAs, at the price of 1/3 more operand bytes per MFMA (the weight tile
//           is shared by 128 rows instead of 256).
// Output width: NI = 2 -> 128-column tiles (wave tile 64 x 64); NI = 1 -> 64-column tiles (wave tile 64 x 32) for the
// cout <= 64 layers (the depth-8 layers of the 3-stage feature net), which would waste half their MFMAs on clamped
// columns of a 128-wide tile.
constexpr int G2_WN = 2, G2_MI = 2;
constexpr int G2_LINE = 128;                          // bytes per row per k-step (both precisions)
template <int WM, int NI = 2> struct G2Cfg {
  static constexpr int BN = G2_WN * NI * 32;
  static constexpr int B_BYTES = BN * G2_LINE;
  static constexpr int BM = WM * G2_MI * 32;
  static constexpr int WAVES = WM * G2_WN;
  static constexpr int THREADS = WAVES * 64;
  static constexpr int NBUF = WM == 4 ? 3 : 2;
  static constexpr int A_BYTES = BM * G2_LINE;
  static constexpr int BUF = A_BYTES + B_BYTES;       // one stage
  static constexpr int TAB = NBUF * BUF;              // neighbour-table slice [BM][8] uint32
  static constexpr int PFS = TAB + BM * 8 * 4;        // 256-B landing pad of the table prefetch (never read)
  static constexpr int LDS = PFS + 256;               // 155 904 B (WM 4) / 69 888 B (WM 2)
  static constexpr int B_PER_WAVE = (BN / 8) / WAVES;                 // weight-tile DMA instructions per wave: 1, 2 or 4
  static constexpr int READS = 2 * (G2_MI + NI);                      // LDS reads of one half-step fragment set
  static constexpr int EPI_LOADS = G2_MI * NI * 4 + NI + G2_MI * 4;   // epilogue operand requests per lane (26 / 17)
  static constexpr int GLDS = 4 + B_PER_WAVE;                         // DMA instructions per wave per k-step: 6 / 8
};

struct Gemm2Args {
  const char* xp; int64_t ldx;            // activation planes; row pitch in BYTES
  const char* aux;                        // rows n_src.. of the id space: [0] zeros, [1 + v] multi-neighbour means
  int64_t n_src;
  const int32_t* nbr_ext;                 // [M, 7]
  const char* tfp; int64_t ldt;           // node-type slab planes (row pitch bytes) or the activation planes again
  const char* W2;                         // [nkt][N][128 B]
  int tpd, nkt_g, nkt;                    // k tiles per direction, gather tiles (7 * tpd), all tiles
  unsigned long long* dbg;                // optional [blocks][8] shader-clock stamps (ofx_set_gconv2_debug)
  int stagger;                            // shader clocks the second block of each CU waits before its first tile (WM 2)
  int prefetch;                           // number of co-resident blocks S (256 x blocks per CU); block b pulls the
                                          // neighbour-table slice of block b + S (same XCD, one round later) into L2
  int prefetch_on;
  int64_t row0;                           // first output row of this launch (bulk + remainder launches split the rows)
  GemmArgs e;                             // M, N, epilogue operands, tile grid
};

__device__ __forceinline__ unsigned long long g2_clock() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t));
  return t;
}

// compile-time unrolled loop: f(std::integral_constant<int, 0>()), ..., f(std::integral_constant<int, N - 1>())
template <typename F, int... Is>
__device__ __forceinline__ void g2_static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
  (f(std::integral_constant<int, Is>()), ...);
}
template <int N, typename F>
__device__ __forceinline__ void g2_static_for(F&& f) {
  g2_static_for_impl(f, std::make_integer_sequence<int, N>());
}

template <int PREC> struct G2Frag;
template <> struct G2Frag<2> { typedef bf16x8_t T; };
template <> struct G2Frag<1> { typedef f16x8_t T; };

template <int PREC>
__device__ __forceinline__ f32x16 g2_mfma(typename G2Frag<PREC>::T a, typename G2Frag<PREC>::T b, f32x16 c);
template <>
__device__ __forceinline__ f32x16 g2_mfma<2>(bf16x8_t a, bf16x8_t b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
template <>
__device__ __forceinline__ f32x16 g2_mfma<1>(f16x8_t a, f16x8_t b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

// ---- LDS reads of the k-loop are INLINE ASM with hand-counted waits.  With a global_load_lds anywhere in a loop
// hipcc's waitcnt pass stops counting LDS reads and puts `s_waitcnt lgkmcnt(0)` in front of every consumer (checked
// on a 40-line reproducer), which would serialise the fragment reads of the next half step with the MFMAs of the
// current one.  Every wait names the registers it guards as "+v" operands, so no consumer can be scheduled above it.
template <int OFF, typename T>
__device__ __forceinline__ void g2_ds_read128(T& d, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF));
}
template <int OFF>
__device__ __forceinline__ void g2_ds_read32(uint32_t& d, unsigned addr) {
  asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF));
}
// half-step fragment set: two piece classes of every A / B fragment of the 64 x 64 wave tile
template <int PREC, int NI> struct G2Half { typename G2Frag<PREC>::T a[2][G2_MI], b[2][NI]; };
// wait until at most N younger LDS reads of this wave are outstanding; guards fragment set F
template <int N, int PREC>
__device__ __forceinline__ void g2_wait_lgkm(G2Half<PREC, 2>& F) {
  asm volatile("s_waitcnt lgkmcnt(%8)"
               : "+v"(F.a[0][0]), "+v"(F.a[0][1]), "+v"(F.a[1][0]), "+v"(F.a[1][1]), "+v"(F.b[0][0]), "+v"(F.b[0][1]),
                 "+v"(F.b[1][0]), "+v"(F.b[1][1])
               : "n"(N));
}
template <int N, int PREC>
__device__ __forceinline__ void g2_wait_lgkm(G2Half<PREC, 1>& F) {
  asm volatile("s_waitcnt lgkmcnt(%6)"
               : "+v"(F.a[0][0]), "+v"(F.a[0][1]), "+v"(F.a[1][0]), "+v"(F.a[1][1]), "+v"(F.b[0][0]), "+v"(F.b[1][0])
               : "n"(N));
}
// wait for all but the newest VM DMA instructions of this wave and for ALL its LDS reads (guarding F), then meet
// the block.  The memory clobber keeps DMA issues and LDS traffic on their side of the barrier.
template <int VM, int PREC, bool BAR = true>
__device__ __forceinline__ void g2_wait_barrier(G2Half<PREC, 2>& F) {
  if (!BAR) {                  // ablation only: the waits without the block-wide rendezvous
    asm volatile("s_waitcnt vmcnt(%8) lgkmcnt(0)"
                 : "+v"(F.a[0][0]), "+v"(F.a[0][1]), "+v"(F.a[1][0]), "+v"(F.a[1][1]), "+v"(F.b[0][0]), "+v"(F.b[0][1]),
                   "+v"(F.b[1][0]), "+v"(F.b[1][1])
                 : "n"(VM)
                 : "memory");
    return;
  }
  asm volatile("s_waitcnt vmcnt(%8) lgkmcnt(0)\n\ts_barrier"
               : "+v"(F.a[0][0]), "+v"(F.a[0][1]), "+v"(F.a[1][0]), "+v"(F.a[1][1]), "+v"(F.b[0][0]), "+v"(F.b[0][1]),
                 "+v"(F.b[1][0]), "+v"(F.b[1][1])
               : "n"(VM)
               : "memory");
}
template <int VM, int PREC, bool BAR = true>
__device__ __forceinline__ void g2_wait_barrier(G2Half<PREC, 1>& F) {
  asm volatile("s_waitcnt vmcnt(%6) lgkmcnt(0)\n\ts_barrier"
               : "+v"(F.a[0][0]), "+v"(F.a[0][1]), "+v"(F.a[1][0]), "+v"(F.a[1][1]), "+v"(F.b[0][0]), "+v"(F.b[1][0])
               : "n"(VM)
               : "memory");
}
template <int PREC>
__device__ __forceinline__ void g2_touch(G2Half<PREC, 2>& F) {
  asm volatile("" : "+v"(F.a[0][0]), "+v"(F.a[0][1]), "+v"(F.a[1][0]), "+v"(F.a[1][1]), "+v"(F.b[0][0]), "+v"(F.b[0][1]),
               "+v"(F.b[1][0]), "+v"(F.b[1][1]));
}
template <int PREC>
__device__ __forceinline__ void g2_touch(G2Half<PREC, 1>& F) {
  asm volatile("" : "+v"(F.a[0][0]), "+v"(F.a[0][1]), "+v"(F.a[1][0]), "+v"(F.a[1][1]), "+v"(F.b[0][0]), "+v"(F.b[1][0]));
}
struct G2Idx { uint32_t v[4]; };
template <int N>
__device__ __forceinline__ void g2_wait_lgkm(G2Idx& I) {
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(I.v[0]), "+v"(I.v[1]), "+v"(I.v[2]), "+v"(I.v[3]) : "n"(N));
}
template <int VM>
__device__ __forceinline__ void g2_wait_barrier() {
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(VM) : "memory");
}

// ---- two-phase epilogue.  With ONE block per CU nothing else hides the latency of the epilogue's own loads: the
// shared epilogue (ofx_gemm_common.h) loads a residual piece, waits for it, stores, 16 times over -- measured at
// 51 k shader clocks per block against 57 k for the whole 30-step k-loop (tools/gconv2_timeline.py).  Here every
// operand the epilogue needs (residual rows, bias, time-embedding rows, batch ids) is REQUESTED before the last
// two k-steps and consumed after them.  Same lane -> element mapping as epilogue_store_v4: lane (k = l31 >> 2,
// q = l31 & 3, h) owns rows q + 4h + 8G + 32i (G < 4, i < MI) of its wave's 32 MI rows, columns 4k..4k+3 of every
// 32-column group j.
typedef float g2_v4f __attribute__((ext_vector_type(4)));
template <int MI, int NI>
struct G2Epi {
  g2_v4f res[MI][NI][4];
  g2_v4f bias[NI];
  int bids[MI][4];
};

// The requests are inline asm: their NUMBER enters a counted s_waitcnt vmcnt(N) of the k-loop (the DMA of the next
// tile must be waited for without waiting for these), so the compiler must neither merge, drop nor reorder them.
// Absent operands (no residual / bias / batch ids) read a dummy line of the packed weights instead.
__device__ __forceinline__ void g2_req128(g2_v4f& d, const void* p) {
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(d) : "v"(p));
}
__device__ __forceinline__ void g2_req32(int& d, const void* p) {
  asm volatile("global_load_dword %0, %1, off" : "=v"(d) : "v"(p));
}

template <int WM, int WN, int MI, int NI>
__device__ __forceinline__ void g2_epilogue_request(const GemmArgs& g, const void* dummy, G2Epi<MI, NI>& P, int64_t m0,
                                                    int64_t n0, int wm, int wn, int l31, int h) {
  const int q = l31 & 3, k = l31 >> 2;
  const int64_t mw = m0 + wm * MI * 32;
  const int64_t mlast = g.M - 1;
  const bool need_bid = g.emb || g.stats;
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int G = 0; G < 4; ++G) {
      int64_t m = mw + i * 32 + q + 4 * h + 8 * G;
      m = m < mlast ? m : mlast;
      g2_req32(P.bids[i][G], need_bid ? (const void*)(g.bid + m) : dummy);
    }
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    int64_t n = n0 + (wn * NI + j) * 32 + 4 * k;
    n = n < g.N ? n : g.N - 4;                               // clamped: out-of-range columns are never stored
    g2_req128(P.bias[j], g.bias ? (const void*)(g.bias + n) : dummy);
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int G = 0; G < 4; ++G) {
        int64_t m = mw + i * 32 + q + 4 * h + 8 * G;
        m = m < mlast ? m : mlast;
        g2_req128(P.res[i][j][G], g.res ? (const void*)(g.res + m * g.ldr + n) : dummy);
      }
  }
}
// after the vmcnt(0) of the last k-steps: ties every requested register to this point of the instruction stream
__device__ __forceinline__ void g2_epilogue_landed(G2Epi<2, 2>& P) {
  asm volatile("" : "+v"(P.res[0][0][0]), "+v"(P.res[0][0][1]), "+v"(P.res[0][0][2]), "+v"(P.res[0][0][3]),
                    "+v"(P.res[0][1][0]), "+v"(P.res[0][1][1]), "+v"(P.res[0][1][2]), "+v"(P.res[0][1][3]),
                    "+v"(P.res[1][0][0]), "+v"(P.res[1][0][1]), "+v"(P.res[1][0][2]), "+v"(P.res[1][0][3]),
                    "+v"(P.res[1][1][0]), "+v"(P.res[1][1][1]), "+v"(P.res[1][1][2]), "+v"(P.res[1][1][3]),
                    "+v"(P.bias[0]), "+v"(P.bias[1]));
  asm volatile("" : "+v"(P.bids[0][0]), "+v"(P.bids[0][1]), "+v"(P.bids[0][2]), "+v"(P.bids[0][3]),
                    "+v"(P.bids[1][0]), "+v"(P.bids[1][1]), "+v"(P.bids[1][2]), "+v"(P.bids[1][3]));
}
__device__ __forceinline__ void g2_epilogue_landed(G2Epi<2, 1>& P) {
  asm volatile("" : "+v"(P.res[0][0][0]), "+v"(P.res[0][0][1]), "+v"(P.res[0][0][2]), "+v"(P.res[0][0][3]),
                    "+v"(P.res[1][0][0]), "+v"(P.res[1][0][1]), "+v"(P.res[1][0][2]), "+v"(P.res[1][0][3]),
                    "+v"(P.bias[0]));
  asm volatile("" : "+v"(P.bids[0][0]), "+v"(P.bids[0][1]), "+v"(P.bids[0][2]), "+v"(P.bids[0][3]),
                    "+v"(P.bids[1][0]), "+v"(P.bids[1][1]), "+v"(P.bids[1][2]), "+v"(P.bids[1][3]));
}

template <int WM, int WN, int MI, int NI>
__device__ __forceinline__ void g2_epilogue_finish(const GemmArgs& g, f32x16 (&acc)[MI][NI], G2Epi<MI, NI>& P, int64_t m0,
                                                   int64_t n0, int wm, int wn, int l31, int h) {
  const int q = l31 & 3, k = l31 >> 2;
  const bool q0 = q & 1, q1 = q & 2;
  const int64_t mw = m0 + wm * MI * 32;
  const int64_t tile_m = m0 / (WM * MI * 32);
  bool uni = true;
  int b0 = 0;
  if (g.emb || g.stats) {
    // batch id of the wave's first row = lane 0's first row (reading it here, not in the request phase, keeps a
    // scalarised load + wait out of the k-loop's tail)
    b0 = __builtin_amdgcn_readfirstlane(P.bids[0][0]);
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int G = 0; G < 4; ++G) uni = uni && (P.bids[i][G] == b0);
    uni = __all(uni);
  }
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int64_t n = n0 + (wn * NI + j) * 32 + 4 * k;
    const bool ncol = n < g.N;
    const int64_t nc = ncol ? n : g.N - 4;
    float4 eu = f4zero();
    if (g.emb && uni) eu = *reinterpret_cast<const float4*>(g.emb + (int64_t)b0 * g.lde + nc);
    const float4 bv = g.bias ? make_float4(P.bias[j].x, P.bias[j].y, P.bias[j].z, P.bias[j].w) : f4zero();
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
        t[G] = make_float4(v0 + bv.x, v1 + bv.y, v2 + bv.z, v3 + bv.w);
      }
#pragma unroll
      for (int G = 0; G < 4; ++G) {
        const int64_t m = mw + i * 32 + q + 4 * h + 8 * G;
        if (m >= g.M || !ncol) continue;
        float4 v = t[G];
        if (g.emb) {
          if (uni) f4add(v, eu);
          else f4add(v, *reinterpret_cast<const float4*>(g.emb + (int64_t)P.bids[i][G] * g.lde + n));
        }
        if (g.res) f4add(v, make_float4(P.res[i][j][G].x, P.res[i][j][G].y, P.res[i][j][G].z, P.res[i][j][G].w));
        if (g.stats) {
          const int b = P.bids[i][G];
          if (!uni && b != sb) {
            if (sb >= 0) flush(sb);
            ssum = f4zero(); ssq = f4zero();
          }
          sb = b;
          f4add(ssum, v);
          ssq.x += v.x * v.x; ssq.y += v.y * v.y; ssq.z += v.z * v.z; ssq.w += v.w * v.w;
        }
        *reinterpret_cast<float4*>(g.out + m * g.ldc + n) = v;
      }
    }
    if (g.stats) {
      if (uni) {
#define OFX_RED(f) f += dpp_xor1(f); f += dpp_xor2(f); f += __shfl_xor(f, 32);
        OFX_RED(ssum.x) OFX_RED(ssum.y) OFX_RED(ssum.z) OFX_RED(ssum.w)
        OFX_RED(ssq.x) OFX_RED(ssq.y) OFX_RED(ssq.z) OFX_RED(ssq.w)
#undef OFX_RED
        if (q == 0 && h == 0 && ncol && mw < g.M) {
          if (g.stats_part) {
            float* o = g.stats_part + ((tile_m * WM + wm) * g.N + n) * 2;
            *reinterpret_cast<float4*>(o) = make_float4(ssum.x, ssq.x, ssum.y, ssq.y);
            *reinterpret_cast<float4*>(o + 4) = make_float4(ssum.z, ssq.z, ssum.w, ssq.w);
          } else {
            flush(b0);
          }
        }
      } else {
        if (sb >= 0 && ncol) flush(sb);
        if (g.stats_part && q == 0 && h == 0 && ncol && mw < g.M) {      // mixed wave: its slot must read as zero
          float* o = g.stats_part + ((tile_m * WM + wm) * g.N + n) * 2;
          *reinterpret_cast<float4*>(o) = f4zero();
          *reinterpret_cast<float4*>(o + 4) = f4zero();
        }
      }
    }
  }
}


template <int PREC, int VARIANT, int WM, int NI>
__global__ void __launch_bounds__(512, 2) gconv2_kernel(const Gemm2Args a) {   // (a WM-dependent bound loses the host stub)
  typedef G2Half<PREC, NI> Half;
  typedef G2Cfg<WM, NI> CF;
  constexpr int G2_NI = NI, G2_BN = CF::BN, G2_EPI_LOADS = CF::EPI_LOADS, RH = CF::READS;
  constexpr int G2_WM = WM, G2_BM = CF::BM, G2_A_BYTES = CF::A_BYTES, G2_BUF = CF::BUF, G2_TAB = CF::TAB;
  constexpr int G2_GLDS_PER_STEP = CF::GLDS;
  extern __shared__ __attribute__((aligned(128))) char smem2[];
  const GemmArgs& g = a.e;

  const int ntile = g.ntm * g.ntn;
  int bid = blockIdx.x;
  {   // XCD-aware bijective tile order: consecutive row tiles (Morton neighbours) share one XCD's L2
    const int q = ntile / 8, r = ntile % 8, xcd = bid % 8, j = bid / 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
  }
  const int tm = bid / g.ntn, tn = bid - tm * g.ntn;
  const int64_t m0 = a.row0 + (int64_t)tm * G2_BM, n0 = (int64_t)tn * G2_BN;

  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wid >> 1, wn = wid & 1;
  const int l31 = lane & 31, h = lane >> 5;
  const bool dbg = a.dbg != nullptr;
  unsigned long long ts0 = 0, ts1 = 0, ts2 = 0, ts3 = 0;
  if (WM == 2 && a.stagger > 0 && blockIdx.x >= 256 && (int)blockIdx.x < a.prefetch) {
    // Two (three with 64-column tiles: 52 KB of LDS each) blocks share a CU and all first-round blocks start together,
    // so their prologues, k-loops and epilogues coincide and nothing overlaps.  Dispatch order observed on gfx950:
    // block b -> XCD b % 8, CU (b / 8) % 32, i.e. blocks 256..511 (and 512..767) are the later slots of every CU; only
    // speed depends on this.  Slot g of n idles for g/n of a tile once (a.stagger = half a tile): the co-resident
    // blocks then stay out of phase for the whole launch and one block's table build / DMA latency / residual reads /
    // stores run under the others' MFMAs.
    const int grp = (int)blockIdx.x >> 8, ngrp = a.prefetch >> 8;
    const unsigned long long wait = (unsigned long long)a.stagger * 2ull * grp / ngrp;
    const unsigned long long t0 = g2_clock();
    while (g2_clock() - t0 < wait) __builtin_amdgcn_s_sleep(32);
  }
  if (dbg) ts0 = g2_clock();

  // ---- neighbour-table slice of this row tile -> LDS, already translated to unsigned 128-B LINE offsets:
  //   tab[r][d < 7] = line offset of source row nbr_ext[m, d] from xlo = min(xp, aux) (rows >= n_src live in `aux`),
  //   tab[r][7]     = line offset of the tile row's own node-type slab row from tfp.
  // (32 bits of line offset span 512 GB)
  const char* const xlo = a.xp < a.aux ? a.xp : a.aux;
  {
    uint32_t* tab = reinterpret_cast<uint32_t*>(smem2 + G2_TAB);
    const int64_t mmax = g.M - 1;
    const int64_t lpr = a.ldx >> 7, lpt = a.ldt >> 7;          // lines per row
    const int64_t x_line = (a.xp - xlo) >> 7, aux_line = (a.aux - xlo) >> 7;   // all 128-B aligned (host-checked)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int e = threadIdx.x + CF::THREADS * t;
      const int r = e >> 3, d = e & 7;
      int64_t m = m0 + r;
      m = m < mmax ? m : mmax;
      int64_t line;
      if (d < 7) {
        const int64_t id = a.nbr_ext[m * 7 + d];
        line = id < a.n_src ? x_line + id * lpr : aux_line + (id - a.n_src) * lpr;
      } else {
        line = m * lpt;
      }
      tab[e] = (uint32_t)line;
    }
  }
  __syncthreads();
  if (dbg) ts1 = g2_clock();

  // ---- table prefetch for the next round.  The table build above is one exposed HBM miss per block (the 6 MB
  // nbr_ext array is long gone from the caches when the next convolution reads it); block b + S runs on the same XCD
  // one round later (S = co-resident blocks, a multiple of 8), so wave 0 pulls that block's slice (BM * 28 B) into this
  // XCD's L2 now with a DMA load whose LDS destination is never read -- no register, no wait: it is older than every
  // counted load of the pipeline.
  if (a.prefetch_on && wid == 0) {
    int nb = (int)blockIdx.x + a.prefetch;
    if (nb < ntile) {
      const int q = ntile / 8, r = ntile % 8, xcd = nb % 8, j = nb / 8;
      nb = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
      const int64_t m2 = a.row0 + (int64_t)(nb / g.ntn) * G2_BM;
      gcp p = (gcp)(a.nbr_ext + m2 * 7) + lane * 128;
      gcp last = (gcp)(a.nbr_ext + g.M * 7) - 4;
      if (lane * 128 < G2_BM * 28 + 128 && p <= last)
        __builtin_amdgcn_global_load_lds(p, (ldsp)(smem2 + CF::PFS), 4, 0, 0);
    }
  }

  // ---- wave-uniform loop operands pinned in SGPRs
  auto sgpr32 = [](int v) {
    int r = __builtin_amdgcn_readfirstlane(v);
    asm volatile("" : "+s"(r));
    return r;
  };
  auto sgpr64 = [](uint64_t v) {
    unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    asm volatile("" : "+s"(lo), "+s"(hi));
    return ((uint64_t)hi << 32) | lo;
  };
  const int64_t wstep = (int64_t)sgpr64((uint64_t)(g.N * (int64_t)G2_LINE));   // bytes per k tile of the packed weights
  const int tpd = sgpr32(a.tpd), nkt_g = sgpr32(a.nkt_g), nkt = sgpr32(a.nkt);
  const gcp xp_s = (gcp)sgpr64((uint64_t)xlo), tfp_s = (gcp)sgpr64((uint64_t)a.tfp);
  const unsigned lds0 = (unsigned)(uintptr_t)(ldsp)smem2;    // LDS byte address of the dynamic segment

  // ---- per-lane DMA source state
  const int q8 = lane & 7, rsub = lane >> 3;                 // slot within the line, row within the 8-row piece
  // rows handled by this lane: A rows wid*32 + j*8 + rsub (j < 4), B columns wid*16 + j*8 + rsub (j < 2);
  // (row >> 1) & 7 == (j*4 + (lane >> 4)) & 7 for both
  const int swz0 = (lane >> 4) & 7, swz1 = (4 + (lane >> 4)) & 7;
  const int pa0 = (q8 ^ swz0) * 16, pa1 = (q8 ^ swz1) * 16;  // byte offset of the piece this lane fetches (j even / odd)
  const unsigned tab_lane = lds0 + G2_TAB + (wid * 32 + rsub) * 32;      // + j*256 + column*4
  // weight tile: wave w stages columns [w * 8 * BPW, (w + 1) * 8 * BPW), 8 per DMA instruction
  constexpr int BPW = CF::B_PER_WAVE;
  gcp wb[4];                               // BPW <= 4 (a dependent bound here makes hipcc drop the host stub)
  {
    const int64_t Nc = g.N;
#pragma unroll
    for (int j = 0; j < BPW; ++j) {
      int64_t c = n0 + wid * (8 * BPW) + j * 8 + rsub;
      c = c < Nc ? c : Nc - 1;
      const int brow = wid * (8 * BPW) + j * 8 + rsub;                 // row of this line in the staged weight tile
      wb[j] = (gcp)a.W2 + c * G2_LINE + (q8 ^ ((brow >> 1) & 7)) * 16;   // source-side swizzle, as for the A rows
    }
  }

  // description of one k tile for the DMA stream (all wave-uniform)
  struct Tile { int tcol; int ktw; gcp base; };      // table column, packed-weight tile, source base + chunk offset
  // table entries of the tile for this lane's four A rows (4 LDS reads)
  auto load_idx = [&](const Tile& T, G2Idx& I) {
    const unsigned ad = tab_lane + T.tcol * 4;
    g2_ds_read32<0>(I.v[0], ad);
    g2_ds_read32<256>(I.v[1], ad);
    g2_ds_read32<512>(I.v[2], ad);
    g2_ds_read32<768>(I.v[3], ad);
  };
  // request the tile into the stage buffer at byte offset `ob` (6 DMA instructions)
  auto issue = [&](const Tile& T, int ob, const G2Idx& I) {
    const gcp b0 = T.base + pa0, b1 = T.base + pa1;
    char* const abuf = smem2 + ob + wid * (32 * G2_LINE);
#pragma unroll
    for (int j = 0; j < 4; ++j)
      __builtin_amdgcn_global_load_lds(((j & 1) ? b1 : b0) + ((uint64_t)I.v[j] << 7),
                                       (ldsp)(abuf + j * (8 * G2_LINE)), 16, 0, 0);
    char* const bbuf = smem2 + ob + G2_A_BYTES + wid * (8 * BPW * G2_LINE);
    const int64_t wo = (int64_t)T.ktw * wstep;
#pragma unroll
    for (int j = 0; j < BPW; ++j)
      __builtin_amdgcn_global_load_lds(wb[j] + wo, (ldsp)(bbuf + j * (8 * G2_LINE)), 16, 0, 0);
  };
  // tile `it` of the k order: channel chunk outer, direction inner over the 7 * tpd gather tiles, then the
  // node-type tiles (prologue only: the loops below advance (dir, chunk) incrementally)
  auto tile_of = [&](int it) {
    Tile T;
    if (it < nkt_g) {
      const int chunk = it / 7, dir = it - chunk * 7;
      T.tcol = dir; T.ktw = dir * tpd + chunk; T.base = xp_s + (int64_t)chunk * G2_LINE;
    } else {
      T.tcol = 7; T.ktw = it; T.base = tfp_s + (int64_t)(it - nkt_g) * G2_LINE;
    }
    return T;
  };

  // ---- per-lane fragment read state: piece class t (= 0..3) of row l31 sits at ((2t + h) ^ s) * 16, s = (l31 >> 1) & 7
  //   PREC 2: half c holds {hi, lo} of k 16c..16c+15   (classes c and 2 + c)
  //   PREC 1: half c holds k 32c..32c+31               (classes 2c and 2c + 1)
  const int s7 = (l31 >> 1) & 7;
  unsigned fa[2][2], fb[2][2];                       // [half][class within the half]: LDS byte addresses in stage 0
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int t = PREC == 2 ? (u == 0 ? c : 2 + c) : 2 * c + u;
      const int po = ((2 * t + h) ^ s7) * 16;
      fa[c][u] = lds0 + (wm * 64 + l31) * G2_LINE + po;
      fb[c][u] = lds0 + G2_A_BYTES + (wn * (32 * G2_NI) + l31) * G2_LINE + po;
    }
  // 8 LDS reads: half c of the tile staged at byte offset ob
  auto read_half = [&](int ob, int c, Half& F) {
    if (VARIANT == 4) {          // ablation: 8 cheap LDS reads (the waits count LDS ops), fragments keep stale registers
      uint32_t d;
#pragma unroll
      for (int u = 0; u < RH; ++u) g2_ds_read32<0>(d, lds0 + G2_TAB);
      g2_touch(F);
      return;
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      g2_ds_read128<0>(F.a[u][0], fa[c][u] + ob);
      g2_ds_read128<32 * G2_LINE>(F.a[u][1], fa[c][u] + ob);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      g2_ds_read128<0>(F.b[u][0], fb[c][u] + ob);
      if constexpr (NI == 2) g2_ds_read128<32 * G2_LINE>(F.b[u][1], fb[c][u] + ob);
    }
  };

  // single LDS read r (0 .. RH-1) of half c: the order read_half issues them in
  auto read_one = [&](int ob, int c, Half& F, auto r_tag) {
    constexpr int r = decltype(r_tag)::value;
    constexpr int NA = 2 * G2_MI;                    // A reads first: (u, i) = (r / MI, r % MI)
    if constexpr (r < NA) {
      constexpr int u = r / G2_MI, i = r % G2_MI;
      if constexpr (i == 0) g2_ds_read128<0>(F.a[u][0], fa[c][u] + ob);
      else g2_ds_read128<32 * G2_LINE>(F.a[u][1], fa[c][u] + ob);
    } else {
      constexpr int q = r - NA, u = q / G2_NI, j = q % G2_NI;
      if constexpr (j == 0) g2_ds_read128<0>(F.b[u][0], fb[c][u] + ob);
      else g2_ds_read128<32 * G2_LINE>(F.b[u][1], fb[c][u] + ob);
    }
  };
  f32x16 acc[G2_MI][G2_NI];
#pragma unroll
  for (int i = 0; i < G2_MI; ++i)
#pragma unroll
    for (int j = 0; j < G2_NI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  auto mfma_half = [&](const Half& F) {
    if (VARIANT == 2) {          // ablation: keep one MFMA per half so the accumulators stay live
      acc[0][0] = g2_mfma<PREC>(F.a[0][0], F.b[0][0], acc[0][0]);
      return;
    }
    if (PREC == 2) {
      // [0] = hi, [1] = lo: small cross terms first, the leading term last
#pragma unroll
      for (int i = 0; i < G2_MI; ++i)
#pragma unroll
        for (int j = 0; j < G2_NI; ++j) acc[i][j] = g2_mfma<PREC>(F.a[1][i], F.b[0][j], acc[i][j]);
#pragma unroll
      for (int i = 0; i < G2_MI; ++i)
#pragma unroll
        for (int j = 0; j < G2_NI; ++j) acc[i][j] = g2_mfma<PREC>(F.a[0][i], F.b[1][j], acc[i][j]);
#pragma unroll
      for (int i = 0; i < G2_MI; ++i)
#pragma unroll
        for (int j = 0; j < G2_NI; ++j) acc[i][j] = g2_mfma<PREC>(F.a[0][i], F.b[0][j], acc[i][j]);
    } else {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int i = 0; i < G2_MI; ++i)
#pragma unroll
          for (int j = 0; j < G2_NI; ++j) acc[i][j] = g2_mfma<PREC>(F.a[t][i], F.b[t][j], acc[i][j]);
    }
  };

#define G2_FENCE() __builtin_amdgcn_sched_barrier(0)
  Half F0, F1;
  G2Idx I;
  // single MFMA m of a half (same order as mfma_half)
  auto mfma_one = [&](const Half& F, auto m_tag) {
    constexpr int m = decltype(m_tag)::value;
    constexpr int per = G2_MI * G2_NI, t = m / per, i = (m / G2_NI) % G2_MI, j = m % G2_NI;
    if constexpr (PREC == 2) {
      if constexpr (t == 0) acc[i][j] = g2_mfma<PREC>(F.a[1][i], F.b[0][j], acc[i][j]);
      else if constexpr (t == 1) acc[i][j] = g2_mfma<PREC>(F.a[0][i], F.b[1][j], acc[i][j]);
      else acc[i][j] = g2_mfma<PREC>(F.a[0][i], F.b[0][j], acc[i][j]);
    } else {
      acc[i][j] = g2_mfma<PREC>(F.a[t][i], F.b[t][j], acc[i][j]);
    }
  };
  // DMA instruction k (0 .. GLDS-1) of a tile request: k < 4 -> A rows, else weight lines
  auto issue_one = [&](const Tile& T, int ob, const G2Idx& I, auto k_tag) {
    constexpr int k = decltype(k_tag)::value;
    if constexpr (k < 4) {
      const gcp b = T.base + ((k & 1) ? pa1 : pa0);
      __builtin_amdgcn_global_load_lds(b + ((uint64_t)I.v[k] << 7),
                                       (ldsp)(smem2 + ob + wid * (32 * G2_LINE) + k * (8 * G2_LINE)), 16, 0, 0);
    } else {
      constexpr int j = k - 4;
      __builtin_amdgcn_global_load_lds(wb[j] + (int64_t)T.ktw * wstep,
                                       (ldsp)(smem2 + ob + G2_A_BYTES + wid * (8 * BPW * G2_LINE) + j * (8 * G2_LINE)),
                                       16, 0, 0);
    }
  };
  constexpr int NMF = (PREC == 2 ? 3 : 2) * G2_MI * G2_NI;           // MFMAs per half step
  // MFMAs of half Fc with (a) the LDS reads of the NEXT half set Fr (from stage ob_r, half c_r) and (b) optionally the
  // DMA request of tile T spliced between them, in a pinned order: the reads / requests issue in the shadow of the
  // matrix pipe instead of in front of it (variant 5; the un-spliced order leaves the pipe idle while a wave issues
  // its 8-12 LDS reads after every barrier and at every step start)
  auto mfma_spliced = [&](const Half& Fc, bool do_read, int ob_r, int c_r, Half& Fr, auto dma_tag, const Tile& T,
                          int ob_dma) {
    constexpr bool DMA = decltype(dma_tag)::value;
    auto body = [&](auto m_tag) {
      constexpr int m = decltype(m_tag)::value;
      mfma_one(Fc, m_tag);
      if constexpr (m < RH) {
        if (do_read) read_one(ob_r, c_r, Fr, m_tag);
      }
      if constexpr (DMA) {
        if constexpr (m == 1) g2_wait_lgkm<(RH < 2 ? RH : 2)>(I);        // table entries: 2 younger reads so far
        if constexpr (m >= 2 && m - 2 < G2_GLDS_PER_STEP) issue_one(T, ob_dma, I, std::integral_constant<int, m - 2>());
      }
      G2_FENCE();
    };
    g2_static_for<NMF>(body);
    if constexpr (RH > NMF) {                                             // reads that did not fit (fp16, 64-column tile)
      g2_static_for<RH>([&](auto r_tag) {
        constexpr int r = decltype(r_tag)::value;
        if constexpr (r >= NMF) {
          if (do_read) read_one(ob_r, c_r, Fr, r_tag);
        }
      });
    }
    if constexpr (DMA) {                                                  // requests that did not fit between the MFMAs
      g2_static_for<G2_GLDS_PER_STEP>([&](auto k_tag) {
        constexpr int k = decltype(k_tag)::value;
        if constexpr (k + 2 >= NMF) issue_one(T, ob_dma, I, k_tag);
      });
    }
  };
  // ---- prologue: tiles 0 and 1 in flight, tile 0 landed for everyone, its first half on its way to registers
  {
    const Tile T0 = tile_of(0);
    load_idx(T0, I);
    g2_wait_lgkm<0>(I);
    issue(T0, 0, I);
  }
  if (nkt > 1) {
    const Tile T1 = tile_of(1);
    load_idx(T1, I);
    g2_wait_lgkm<0>(I);
    issue(T1, G2_BUF, I);
    g2_wait_barrier<G2_GLDS_PER_STEP>();
  } else {
    g2_wait_barrier<0>();
  }
  if (dbg) ts2 = g2_clock();
  read_half(0, 0, F0);

  // One k-step.  On entry: F0 = first half of tile `it` (8 reads, possibly still in flight), tile it+1 requested.
  //   4 table reads of tile it+2 | 8 reads: second half of tile it -> F1 | wait F0 (12 younger reads) |
  //   MFMAs of F0 interleaved with: wait table (8 younger) -> 6 DMA requests of tile it+2 |
  //   wait (tile it+1 landed, all own LDS reads done) + barrier | 8 reads: first half of tile it+1 -> F0 | MFMAs of F1
  // ob / obn / obnn: byte offsets of the stage buffers of tiles it, it+1, it+2 (rotating).
  int ob = 0, obn = G2_BUF, obnn = 2 * G2_BUF;
  // `last`: the final steady-state step also REQUESTS the epilogue's operands (26 loads) so that they travel during
  // the last k-steps; its barrier lets those (and, with 3 stages, the 6 new DMA) stay outstanding.
  G2Epi<G2_MI, G2_NI> P;
  const bool vec4 = g.vec4 != 0;
  constexpr int MFMAS = (PREC == 2 ? 3 : 2) * G2_MI * G2_NI;
  constexpr int MFMA_PER_ROUND = MFMAS / G2_GLDS_PER_STEP > 0 ? MFMAS / G2_GLDS_PER_STEP : 1;
  auto interleave_dma = [&]() {
    // rounds of {MFMAs, address arithmetic, 1 DMA}: the DMA issues ride in the MFMAs' shadow
#pragma unroll
    for (int k = 0; k < G2_GLDS_PER_STEP; ++k) {
      __builtin_amdgcn_sched_group_barrier(0x008, MFMA_PER_ROUND, 0);
      __builtin_amdgcn_sched_group_barrier(0x006, 6, 0);
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    }
  };
  auto step_issue = [&](const Tile& T, auto last_tag) {
    constexpr bool LAST = decltype(last_tag)::value;
    if constexpr ((VARIANT == 5 || VARIANT == 6) && CF::NBUF == 3) {
      // spliced schedule, 3 stages: table reads | wait F0 (4 younger) | {MFMA F0, read F1, request tile it+2} |
      // barrier | {MFMA F1, read F0 of tile it+1}
      load_idx(T, I);
      g2_wait_lgkm<4, PREC>(F0);
      G2_FENCE();
      if (LAST) {
        g2_epilogue_request<G2_WM, G2_WN, G2_MI, G2_NI>(g, (const void*)a.W2, P, m0, n0, wm, wn, l31, h);
        G2_FENCE();
      }
      mfma_spliced(F0, true, ob, 1, F1, std::true_type(), T, obnn);
      g2_wait_barrier<G2_GLDS_PER_STEP + (LAST ? G2_EPI_LOADS : 0), PREC, VARIANT != 6>(F1);
      G2_FENCE();
      mfma_spliced(F1, true, obn, 0, F0, std::false_type(), T, 0);
      const int t = ob; ob = obn; obn = obnn; obnn = t;
    } else if constexpr (VARIANT == 5 || VARIANT == 6) {
      // spliced schedule, 2 stages: wait F0 | {MFMA F0, read F1} | barrier | table reads | {MFMA F1, read F0 of tile
      // it+1, request tile it+2 into the buffer tile it just left}
      g2_wait_lgkm<0, PREC>(F0);
      G2_FENCE();
      if (LAST) {
        g2_epilogue_request<G2_WM, G2_WN, G2_MI, G2_NI>(g, (const void*)a.W2, P, m0, n0, wm, wn, l31, h);
        G2_FENCE();
      }
      mfma_spliced(F0, true, ob, 1, F1, std::false_type(), T, 0);
      g2_wait_barrier<(LAST ? G2_EPI_LOADS : 0), PREC>(F1);
      G2_FENCE();
      load_idx(T, I);
      mfma_spliced(F1, true, obn, 0, F0, std::true_type(), T, ob);
      const int t = ob; ob = obn; obn = t;
    } else if constexpr (CF::NBUF == 3) {
      // 3 stages: tile it+2 is requested in the FIRST half of step it (its buffer was left at step it-1)
      load_idx(T, I);
      read_half(ob, 1, F1);
      g2_wait_lgkm<4 + RH, PREC>(F0);
      G2_FENCE();
      if (LAST) {
        g2_epilogue_request<G2_WM, G2_WN, G2_MI, G2_NI>(g, (const void*)a.W2, P, m0, n0, wm, wn, l31, h);
        G2_FENCE();
      }
      g2_wait_lgkm<RH>(I);
      if (VARIANT != 3) issue(T, obnn, I);
      if (VARIANT == 0) {
        G2_FENCE();
      }
      mfma_half(F0);
      if (VARIANT == 1) interleave_dma();
      G2_FENCE();
      g2_wait_barrier<(VARIANT == 3 ? 0 : G2_GLDS_PER_STEP) + (LAST ? G2_EPI_LOADS : 0), PREC>(F1);
      G2_FENCE();
      read_half(obn, 0, F0);
      G2_FENCE();
      mfma_half(F1);
      G2_FENCE();
      const int t = ob; ob = obn; obn = obnn; obnn = t;
    } else {
      // 2 stages: tile it+2 goes into the buffer tile `it` leaves at this step's barrier, i.e. it is requested in
      // the SECOND half of step it and has one k-step to land (the co-resident block covers a late one)
      read_half(ob, 1, F1);
      g2_wait_lgkm<RH, PREC>(F0);
      G2_FENCE();
      if (LAST) {
        g2_epilogue_request<G2_WM, G2_WN, G2_MI, G2_NI>(g, (const void*)a.W2, P, m0, n0, wm, wn, l31, h);
        G2_FENCE();
      }
      mfma_half(F0);
      G2_FENCE();
      g2_wait_barrier<(LAST ? G2_EPI_LOADS : 0), PREC>(F1);
      G2_FENCE();
      load_idx(T, I);
      read_half(obn, 0, F0);
      g2_wait_lgkm<RH>(I);
      if (VARIANT != 3) issue(T, ob, I);
      if (VARIANT == 0) {
        G2_FENCE();
      }
      mfma_half(F1);
      if (VARIANT == 1) interleave_dma();
      G2_FENCE();
      const int t = ob; ob = obn; obn = t;
    }
  };
  // tile it+2 of the k order = (chunk gc, direction gd) while it is a gather tile, advanced without division
  int it = 0, gd = 2, gc = 0;                                      // nkt_g >= 7 > 2
  auto next_tile = [&]() {
    const int ti = it + 2;
    const bool gat = ti < nkt_g;
    Tile T;
    T.tcol = gat ? gd : 7;
    T.ktw = gat ? gd * tpd + gc : ti;
    T.base = (gat ? xp_s : tfp_s) + (int64_t)(gat ? gc : ti - nkt_g) * G2_LINE;
    const int wrap = gd == 6;
    gd = wrap ? 0 : gd + 1;
    gc += wrap;
    return T;
  };
  for (; it + 3 < nkt; ++it) step_issue(next_tile(), std::false_type());
  if (it + 2 < nkt) {                            // always taken: nkt >= 7 (host-checked)
    step_issue(next_tile(), std::true_type());
    ++it;
  }
  for (; it < nkt; ++it) {                       // last two tiles: nothing left to request
    if constexpr (VARIANT == 5 || VARIANT == 6) {
      Tile Tn = {};
      g2_wait_lgkm<0, PREC>(F0);
      G2_FENCE();
      mfma_spliced(F0, true, ob, 1, F1, std::false_type(), Tn, 0);
      g2_wait_barrier<0, PREC>(F1);
      G2_FENCE();
      mfma_spliced(F1, it + 1 < nkt, obn, 0, F0, std::false_type(), Tn, 0);
    } else {
    read_half(ob, 1, F1);
    g2_wait_lgkm<RH, PREC>(F0);
    G2_FENCE();
    mfma_half(F0);
    G2_FENCE();
    g2_wait_barrier<0, PREC>(F1);
    G2_FENCE();
    if (it + 1 < nkt) read_half(obn, 0, F0);
    G2_FENCE();
    mfma_half(F1);
    G2_FENCE();
    }
    if (CF::NBUF == 3) {
      const int t = ob; ob = obn; obn = obnn; obnn = t;
    } else {
      const int t = ob; ob = obn; obn = t;
    }
  }
  g2_epilogue_landed(P);
#undef G2_FENCE
  if (dbg) ts3 = g2_clock();

  if (vec4) g2_epilogue_finish<G2_WM, G2_WN, G2_MI, G2_NI>(g, acc, P, m0, n0, wm, wn, l31, h);
  else epilogue_store_scalar<G2_WM, G2_WN, G2_MI, G2_NI>(g, acc, m0, n0, wm, wn, l31, h, 0);
  if (dbg) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long ts4 = g2_clock();
    if (threadIdx.x == 0) {
      unsigned long long* o = a.dbg + (size_t)blockIdx.x * 8;
      o[0] = ts0; o[1] = ts1; o[2] = ts2; o[3] = ts3; o[4] = ts4;
      o[5] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_REG_HW_ID
    }
  }
}

// ------------------------------------------------------------------------------------------------
// plane conversion helpers
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

// fp32 [n, C] (zero-extended to Cpad columns) -> planes.  mode 2: lane q of every 8-lane group owns channels
// 4q..4q+3 of a 32-channel chunk (one coalesced 128-B line in, 64 B hi + 64 B lo out, safe in place);
// mode 1: fp16 row-major (8 B per lane).
__global__ void __launch_bounds__(256) planes_split_kernel(const float* __restrict__ x, int64_t ldx, int64_t n, int C,
                                                           int Cpad, int mode, char* __restrict__ out, int64_t ldo) {
  const int CT = Cpad >> 2;
  const int64_t total = n * CT;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = t / CT;
    const int c = (int)(t - r * CT) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c + 3 < C) {
      v = *reinterpret_cast<const float4*>(x + r * ldx + c);
    } else {
      if (c + 0 < C) v.x = x[r * ldx + c];
      if (c + 1 < C) v.y = x[r * ldx + c + 1];
      if (c + 2 < C) v.z = x[r * ldx + c + 2];
    }
    if (mode == 2) {
      const unsigned h0 = g2_pk_bf16(v.x, v.y), h1 = g2_pk_bf16(v.z, v.w);
      const unsigned l0 = g2_pk_bf16(v.x - g2_bf16_lo(h0), v.y - g2_bf16_hi(h0));
      const unsigned l1 = g2_pk_bf16(v.z - g2_bf16_lo(h1), v.w - g2_bf16_hi(h1));
      char* o = out + r * ldo + (c >> 5) * 128 + (c & 31) * 2;
      *reinterpret_cast<uint2*>(o) = make_uint2(h0, h1);
      *reinterpret_cast<uint2*>(o + 64) = make_uint2(l0, l1);
    } else {
      *reinterpret_cast<uint2*>(out + r * ldo + c * 2) = make_uint2(g2_pk_f16(v.x, v.y), g2_pk_f16(v.z, v.w));
    }
  }
}

extern "C" int ofx_planes_split(const float* x, int64_t ldx, int64_t n, int C, int Cpad, int mode, void* out,
                                int64_t ldo_bytes, void* stream) {
  const int chunk = mode == 2 ? 32 : 64;
  if ((mode != 1 && mode != 2) || n < 0 || C < 1 || Cpad < C || (Cpad % chunk) || ldx < C || !out ||
      ldo_bytes < (int64_t)Cpad * (mode == 2 ? 4 : 2) || (ldo_bytes & 15) || ((uintptr_t)out & 15) || (n > 0 && !x))
    return OFX_EINVAL;
  if (C >= 4 && ((ldx & 3) || ((uintptr_t)x & 15))) return OFX_EINVAL;
  if (n > 0)
    planes_split_kernel<<<ofx_grid(n * (Cpad / 4), 256), 256, 0, ofx_stream(stream)>>>(x, ldx, n, C, Cpad, mode,
                                                                                      (char*)out, ldo_bytes);
  OFX_LAUNCH_CHECK();
  return OFX_OK;
}

// planes -> fp32 (tests / debugging)
__global__ void __launch_bounds__(256) planes_merge_kernel(const char* __restrict__ p, int64_t ldp, int64_t n, int C,
                                                           int mode, float* __restrict__ out, int64_t ldo) {
  const int64_t total = n * C;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = t / C;
    const int c = (int)(t - r * C);
    float v;
    if (mode == 2) {
      const unsigned short* q = reinterpret_cast<const unsigned short*>(p + r * ldp + (c >> 5) * 128 + (c & 31) * 2);
      v = __uint_as_float((unsigned)q[0] << 16) + __uint_as_float((unsigned)q[32] << 16);
    } else {
      v = (float)*reinterpret_cast<const _Float16*>(p + r * ldp + c * 2);
    }
    out[r * ldo + c] = v;
  }
}
extern "C" int ofx_planes_merge(const void* planes, int64_t ldp_bytes, int64_t n, int C, int mode, float* out,
                                int64_t ldo, void* stream) {
  const int chunk = mode == 2 ? 32 : 64;
  if ((mode != 1 && mode != 2) || n < 0 || C < 1 || (C % chunk) || !planes || !out || ldo < C) return OFX_EINVAL;
  if (n > 0)
    planes_merge_kernel<<<ofx_grid(n * C, 256), 256, 0, ofx_stream(stream)>>>((const char*)planes, ldp_bytes, n, C,
                                                                              mode, out, ldo);
  OFX_LAUNCH_CHECK();
  return OFX_OK;
}

// aux[0] = zero row; aux[1 + v] = mean over segment multi_seg[v] of the (re-assembled) source rows, re-split.
// One thread per (aux row, 8 channels).
__global__ void __launch_bounds__(256) planes_multi_mean_kernel(const char* __restrict__ xp, int64_t ldx, int cin,
                                                                const int32_t* __restrict__ seg_ptr,
                                                                const int32_t* __restrict__ col,
                                                                const int32_t* __restrict__ multi_seg, int64_t V,
                                                                int mode, char* __restrict__ aux) {
  const int c8n = cin >> 3;
  const int64_t total = (V + 1) * c8n;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t v = t / c8n;
    const int c = (int)(t - v * c8n) * 8;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // byte offset of this 8-channel piece inside a row
    const int64_t poff = mode == 2 ? (int64_t)(c >> 5) * 128 + (c & 31) * 2 : (int64_t)c * 2;
    if (v > 0) {
      const int64_t s = multi_seg[v - 1];
      const int32_t b = seg_ptr[s], e = seg_ptr[s + 1];
      for (int32_t p = b; p < e; ++p) {
        const char* row = xp + (int64_t)col[p] * ldx + poff;
        const u32x4 hi = *reinterpret_cast<const u32x4*>(row);
        if (mode == 2) {
          const u32x4 lo = *reinterpret_cast<const u32x4*>(row + 64);
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            acc[2 * k] += g2_bf16_lo(hi[k]) + g2_bf16_lo(lo[k]);
            acc[2 * k + 1] += g2_bf16_hi(hi[k]) + g2_bf16_hi(lo[k]);
          }
        } else {
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            acc[2 * k] += (float)__builtin_bit_cast(_Float16, (unsigned short)(hi[k] & 0xffffu));
            acc[2 * k + 1] += (float)__builtin_bit_cast(_Float16, (unsigned short)(hi[k] >> 16));
          }
        }
      }
      const float inv = 1.f / (float)(e - b);
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] *= inv;
    }
    char* o = aux + v * ldx + poff;
    if (mode == 2) {
      u32x4 hi, lo;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        hi[k] = g2_pk_bf16(acc[2 * k], acc[2 * k + 1]);
        lo[k] = g2_pk_bf16(acc[2 * k] - g2_bf16_lo(hi[k]), acc[2 * k + 1] - g2_bf16_hi(hi[k]));
      }
      *reinterpret_cast<u32x4*>(o) = hi;
      *reinterpret_cast<u32x4*>(o + 64) = lo;
    } else {
      u32x4 hv;
#pragma unroll
      for (int k = 0; k < 4; ++k) hv[k] = g2_pk_f16(acc[2 * k], acc[2 * k + 1]);
      *reinterpret_cast<u32x4*>(o) = hv;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// weights -> [k tile][column][128-B line]; k order of the fused GraphConv: 7 x cin gathered channels
// (direction-major), then the node-type rows padded to a whole tile.
static inline int64_t g2_chunk(int mode) { return mode == 2 ? 32 : 64; }
extern "C" int64_t ofx_planes_packed_ktiles(int cin, int nt, int mode) {
  const int64_t ch = g2_chunk(mode);
  return 7 * ((int64_t)cin / ch) + (nt > 1 ? (7 * (int64_t)nt + ch - 1) / ch : 0);
}
extern "C" int64_t ofx_planes_packed_bytes(int cin, int nt, int cout, int mode) {
  return ofx_planes_packed_ktiles(cin, nt, mode) * (int64_t)cout * G2_LINE;
}

__global__ void __launch_bounds__(256) planes_pack_kernel(const float* __restrict__ W, int64_t sk, int64_t sn, int cin,
                                                          int ntc, int64_t N, int64_t nkt, int mode,
                                                          char* __restrict__ out) {
  // one thread per 16-B piece: (k tile, column, piece)
  const int64_t total = nkt * N * 8;
  const int ch = mode == 2 ? 32 : 64;
  const int64_t Kf = 7 * (int64_t)cin;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int p = (int)(t & 7);
    const int64_t n = (t >> 3) % N, kt = (t >> 3) / N;
    const int64_t k0 = kt * ch + (mode == 2 ? (p & 3) * 8 : p * 8);
    float w[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int64_t k = k0 + e;
      int64_t src = -1;
      if (k < Kf) {
        const int64_t dir = k / cin, c = k - dir * cin;
        src = dir * (cin + ntc) + c;
      } else if (ntc > 0 && k - Kf < 7 * (int64_t)ntc) {
        const int64_t kk = k - Kf, dir = kk / ntc, ty = kk - dir * ntc;
        src = dir * (cin + ntc) + cin + ty;
      }
      w[e] = src >= 0 ? W[src * sk + n * sn] : 0.f;
    }
    u32x4 o;
    if (mode == 2) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const unsigned hi = g2_pk_bf16(w[2 * e], w[2 * e + 1]);
        o[e] = p < 4 ? hi : g2_pk_bf16(w[2 * e] - g2_bf16_lo(hi), w[2 * e + 1] - g2_bf16_hi(hi));
      }
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = g2_pk_f16(w[2 * e], w[2 * e + 1]);
    }
    *reinterpret_cast<u32x4*>(out + t * 16) = o;
  }
}

extern "C" int ofx_pack_weights_planes(const float* W, int64_t sk, int64_t sn, int cin, int nt, int cout, int mode,
                                       void* out, void* stream) {
  if (!W || !out || (mode != 1 && mode != 2) || cin < 1 || (cin % g2_chunk(mode)) || nt < 0 || cout < 1 ||
      ((uintptr_t)out & 15))
    return OFX_EINVAL;
  const int64_t nkt = ofx_planes_packed_ktiles(cin, nt, mode);
  planes_pack_kernel<<<ofx_grid(nkt * cout * 8, 256), 256, 0, ofx_stream(stream)>>>(W, sk, sn, cin, nt > 1 ? nt : 0,
                                                                                    cout, nkt, mode, (char*)out);
  OFX_LAUNCH_CHECK();
  return OFX_OK;
}

// ------------------------------------------------------------------------------------------------
int ofx_launch_stats_reduce(const GemmArgs& g, int wr_rows, hipStream_t st);   // ofx_gemm.hip

static int g2_variant = 5;     // 5 = spliced schedule (product); 1 / 0 = earlier schedules, 2-4 / 6 = ablations
extern "C" int ofx_set_gconv2_variant(int v) {
  if (v < 0 || v > 6) return OFX_EINVAL;
  g2_variant = v;
  return OFX_OK;
}
static unsigned long long* g2_debug = nullptr;
extern "C" int ofx_set_gconv2_debug(void* buf) {
  g2_debug = (unsigned long long*)buf;
  return OFX_OK;
}

template <int PREC, int VARIANT, int WM, int NI>
static int g2_launch(const Gemm2Args& a, hipStream_t st) {
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gconv2_kernel<PREC, VARIANT, WM, NI>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, G2Cfg<WM, NI>::LDS) != hipSuccess)
      return OFX_ELAUNCH;
    attr_set = true;
  }
  gconv2_kernel<PREC, VARIANT, WM, NI><<<a.e.ntm * a.e.ntn, G2Cfg<WM, NI>::THREADS, G2Cfg<WM, NI>::LDS, st>>>(a);
  return OFX_OK;
}

// block geometry: 0 = automatic, 2 = 128-row tiles (two blocks per CU), 4 = 256-row tiles (one block per CU)
static int g2_wm = 0;
static int g2_stagger_per_ktile = 1100;     // shader clocks per k tile of the start offset between co-resident blocks
extern "C" int ofx_set_gconv2_stagger(int clocks_per_ktile) {
  if (clocks_per_ktile < 0 || clocks_per_ktile > 100000) return OFX_EINVAL;
  g2_stagger_per_ktile = clocks_per_ktile;
  return OFX_OK;
}
static int g2_prefetch = 1;                 // table prefetch one round ahead (A/B knob)
static int g2_slots3 = 1;                   // 64-column tiles: assume three co-resident blocks per CU (A/B: bit 1 of `on`)
extern "C" int ofx_set_gconv2_prefetch(int on) {
  g2_prefetch = (on & 1) ? 1 : 0;
  g2_slots3 = (on & 2) ? 0 : 1;             // on = 1: default; on = 3: prefetch with the two-slot assumption
  return OFX_OK;
}
extern "C" int ofx_set_gconv2_tile(int wm) {
  if (wm != 0 && wm != 1 && wm != 2 && wm != 4) return OFX_EINVAL;     // 1: the plain width rule (A/B of the automatic choice)
  g2_wm = wm;
  return OFX_OK;
}

extern "C" int ofx_graphconv_fwd_planes(const void* xp, int64_t ldx_bytes, int cin, int64_t n_nodes,
                                        const int32_t* seg_ptr, const int32_t* col, const int32_t* nbr_ext,
                                        const int32_t* multi_seg, int64_t n_multi, void* aux, const void* tfp,
                                        int64_t ldt_bytes, int nt, const void* W2, int cout, const float* bias,
                                        const float* emb, int64_t lde, const int32_t* batch_id, const float* res,
                                        int64_t ldr, float* out, int64_t ldc, double* stats, int64_t stats_ld, void* ws,
                                        size_t ws_bytes, int mode, int aux_ready, void* stream) {
  if (mode != 1 && mode != 2) return OFX_EINVAL;
  const int64_t ch = g2_chunk(mode);
  if (n_nodes == 0 && cin >= 1 && cout >= 1) return OFX_OK;
  if (n_nodes < 0 || cin < 1 || (cin % ch) || cout < 1 || !xp || !seg_ptr || !col || !nbr_ext || !aux || !W2 || !out ||
      ldx_bytes < (int64_t)cin * (mode == 2 ? 4 : 2) || (ldx_bytes & 15) || ((uintptr_t)xp & 127) ||
      ((uintptr_t)aux & 127) || ((uintptr_t)W2 & 127) || ldc < cout || (res && ldr < cout) ||
      (emb && (!batch_id || lde < cout)) || n_multi < 0 || (n_multi > 0 && !multi_seg) || nt < 0)
    return OFX_EINVAL;
  const int ntc = nt > 1 ? nt : 0;
  if (ntc > 0 && (!tfp || (ldt_bytes & 15) || ((uintptr_t)tfp & 127))) return OFX_EINVAL;
  if (stats && (!batch_id || stats_ld < cout)) return OFX_EINVAL;
  hipStream_t st = ofx_stream(stream);
  if (!aux_ready)         // the producer of xp (ofx_gn_apply_planes / ofx_planes_split) may have written aux already
    planes_multi_mean_kernel<<<ofx_grid((n_multi + 1) * (cin / 8), 256), 256, 0, st>>>(
        (const char*)xp, ldx_bytes, cin, seg_ptr, col, multi_seg, n_multi, mode, (char*)aux);
  Gemm2Args a = {};
  a.xp = (const char*)xp; a.ldx = ldx_bytes; a.aux = (const char*)aux; a.n_src = n_nodes; a.nbr_ext = nbr_ext;
  a.tfp = ntc ? (const char*)tfp : (const char*)xp; a.ldt = ntc ? ldt_bytes : ldx_bytes;
  a.W2 = (const char*)W2;
  a.dbg = g2_debug;
  a.tpd = (int)(cin / ch); a.nkt_g = 7 * a.tpd; a.nkt = (int)ofx_planes_packed_ktiles(cin, nt, mode);
  GemmArgs& g = a.e;
  g.M = n_nodes; g.N = cout; g.K = g.Kp = (int64_t)a.nkt * ch; g.bias = bias; g.emb = emb; g.lde = lde; g.bid = batch_id;
  g.res = res; g.ldr = ldr; g.out = out; g.ldc = ldc; g.nsplit = 1;
  {
    auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    g.vec4 = g.N % 4 == 0 && al16(g.out) && g.ldc % 4 == 0 && (!g.res || (al16(g.res) && g.ldr % 4 == 0)) &&
             (!g.emb || (al16(g.emb) && g.lde % 4 == 0)) && (!g.bias || al16(g.bias));
  }
  const int ni = cout <= 64 ? 1 : 2;              // 64-column tiles for the narrow layers
  g.ntn = (int)ofx_cdiv(g.N, 64 * ni);
  if (stats) {
    g.stats = stats; g.stats_ld = stats_ld;
    const int64_t nwr = ofx_cdiv(g.M, 64);
    if (g.vec4 && ws && (size_t)nwr * g.N * 2 * sizeof(float) <= ws_bytes && (((uintptr_t)ws) & 15) == 0) {
      g.stats_part = (float*)ws; g.stats_part_bytes = ws_bytes;
    }
  }
  // ---- geometry: one output-column tile (cout <= 128) -> 128-row tiles, two (64-column tiles: three) staggered blocks
  // per CU (prologue / epilogue overlap, +3-9 %); several column tiles -> 256-row tiles (larger weight-tile reuse, cout 256 / 512 layers
  // +5-15 %).  Splitting the rows into a bulk launch of whole "rounds" of 256-row tiles plus a remainder launch of
  // 128-row tiles (against tile quantisation: 530 tiles on 256 CUs = 2.07 rounds) was built and measured: no gain
  // (blocks do not run in lock-step rounds, and the second launch pays its own fill and drain) --
  // profiles/r02/gconv2_geometry.txt.
  int rc = OFX_OK;
#define G2_GO(P_, V_, WM_)                                                                            \
  (ni == 1 ? (WM_ == 4 ? g2_launch<P_, V_, 4, 1>(a, st) : g2_launch<P_, V_, 2, 1>(a, st))             \
           : (WM_ == 4 ? g2_launch<P_, V_, 4, 2>(a, st) : g2_launch<P_, V_, 2, 2>(a, st)))
  auto launch = [&](int wm, int64_t row0, int64_t rows) -> int {
    a.row0 = row0;
    g.ntm = (int)ofx_cdiv(rows, wm * 64);
    a.stagger = wm == 2 ? g2_stagger_per_ktile * a.nkt : 0;
    a.prefetch = wm == 4 ? 256 : (ni == 1 && g2_slots3 ? 768 : 512);      // co-resident blocks (LDS-bound: 1 / 2 / 3 per CU)
    a.prefetch_on = g2_prefetch;
    if (mode == 2) {
      switch (g2_variant) {
        case 0: return G2_GO(2, 0, wm);
        case 2: return G2_GO(2, 2, wm);      // ablations (wrong results): no MFMA / no DMA / no LDS reads
        case 3: return G2_GO(2, 3, wm);
        case 4: return G2_GO(2, 4, wm);
        case 5: return G2_GO(2, 5, wm);      // spliced schedule (reads / requests between the MFMAs)
        case 6: return G2_GO(2, 6, wm);      // ablation: spliced schedule without the per-step s_barrier (races)
        case 1: return G2_GO(2, 1, wm);
        default: return G2_GO(2, 5, wm);
      }
    }
    return g2_variant == 0 ? G2_GO(1, 0, wm) : (g2_variant == 1 ? G2_GO(1, 1, wm) : G2_GO(1, 5, wm));
  };
  // narrow layers on very long tensors (depth 7 / 8 of the feature net: >= 8 rounds of 256-row tiles) amortise the
  // prologue better with the 256-row geometry too: 2-5 % (tools/gconv2_tile_d8.py)
  const int64_t tiles4 = ofx_cdiv(g.M, 256) * g.ntn;
  // wide layers whose 256-row tiles leave a small last round (depth 5 of the jittered shell-6 batch: 530 tiles on 256
  // CUs) pay a whole big-tile time for it; the same rows as 128-row tiles pay a small-tile time: 2-10 %
  // (profiles/r02/gconv2_geometry.txt, d5 256->256 / 512->256 / 768->256)
  const int64_t tail4 = tiles4 % 256;
  const bool small_tail = tail4 > 0 && tail4 <= 64 && tiles4 < 1536;
  int wm_auto = (cout <= 128 && tiles4 < 2048) ? 2 : 4;
  if (cout > 128 && small_tail) wm_auto = 2;
  if (g2_wm == 1) wm_auto = cout <= 128 ? 2 : 4;                       // A/B: the plain width rule
  rc = launch((g2_wm == 2 || g2_wm == 4) ? g2_wm : wm_auto, 0, g.M);
#undef G2_GO
  if (rc) return rc;
  if (g.stats_part) {
    rc = ofx_launch_stats_reduce(g, 64, st);
    if (rc) return rc;
  }
  OFX_LAUNCH_CHECK();
  return OFX_OK;
}

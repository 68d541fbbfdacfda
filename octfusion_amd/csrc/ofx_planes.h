// Device helpers shared by the planes GraphConv kernels (ofx_gemm2.hip: one tile per block; ofx_gemm3.hip:
// persistent stream-K blocks): tile geometry, MFMA fragment types, the hand-counted inline-asm LDS reads / waits
// and the two-phase epilogue.
#pragma once
#include <type_traits>
#include <utility>

#include "ofx_gemm_common.h"

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef const char __attribute__((address_space(1)))* gcp;
typedef __attribute__((address_space(3))) void* ldsp;

// ---- 16-bit operand pairs.  A value travels as hi + lo, both in the same 16-bit format:
//   mode 2: bf16 pairs (8 + 8 significand bits: a = hi + lo to 2^-18 relative) -- "bf16x3";
//   mode 3: fp16 pairs (11 + 11 bits: a = hi + lo to 2^-23 relative while lo is a normal fp16, to 2^-25 ABSOLUTE
//           below -- v_mfma_f32_32x32x16_f16 honours fp16 denormal inputs on gfx950, tools/probes/mfma_f16_denorm.hip)
//           -- "fp16x3": the same three MFMAs per product, ~30x less rounding than bf16x3.  Values beyond the fp16
//           range (+-65504) are NOT saturated: hi becomes +-Inf, lo = x - hi = -+Inf, and every product they enter is
//           NaN -- the result is loudly non-finite and the caller retries in bf16x3 (include/ofx.h, range guard).
//           Normalised activations are many orders of magnitude below the limit; weights are scaled per tensor.
// (g2_pk_* / g2_split2 / g2_join2 / g2_pairs live in ofx_gemm_common.h: the GEMM epilogues write planes too)

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
  static constexpr int EPI_LOADS = G2_MI * NI * 4 + NI + 1;          // epilogue operand requests per lane (19 / 10)
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
template <> struct G2Frag<3> { typedef f16x8_t T; };      // fp16 hi / lo pairs: the three-term scheme on the fp16 pipe
// PREC 2 / 3: a line is [hi x 32 | lo x 32] of a 32-channel chunk and a product takes three MFMAs; PREC 1: a line is
// 64 fp16 channels, one MFMA per product.
template <int PREC> constexpr bool g2_three_term() { return PREC >= 2; }

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
template <>
__device__ __forceinline__ f32x16 g2_mfma<3>(f16x8_t a, f16x8_t b, f32x16 c) {
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
  int bid;                   // batch id of row (wave's first row + lane): all 64 rows of the wave, one per lane
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

// The requests come in 1 + MI * NI slices so that a kernel may spread them over several k-steps (vmcnt is an in-order
// counter: whatever is requested in one step has to land before the NEXT step's DMA can be consumed, so the residual
// tile -- 128 KB per 256 x 128 tile, from HBM -- is better fetched 32 KB per step than all at once):
//   slice 0             : the batch id of one of the wave's 64 rows per lane (1 dword load) and the bias / embedding
//                         line of every column group (NI loads)
//   slice 1 + i * NI + j: the residual rows of sub-tile (i, j) (4 loads)
// b0w >= 0: batch element of the wave's first row, fetched ahead by the caller -- then a layer with a time embedding
// and no bias gets the embedding row of that batch element in its bias slot (g2_epilogue_finish(..., emb_in_bias));
// b0w < 0: the slot holds the bias only, the finish phase loads the embedding row itself.  (No load is issued here
// for it: one compiler-visible VGPR load inside a k-step makes hipcc drain the whole DMA queue -- vmcnt(0) -- there.)
template <int NI> constexpr int g2_epi_slice_loads(int slice) { return slice == 0 ? 1 + NI : 4; }
template <int WM, int WN, int MI, int NI, int SLICE>
__device__ __forceinline__ void g2_epilogue_request_slice(const GemmArgs& g, const void* dummy, G2Epi<MI, NI>& P,
                                                          int64_t m0, int64_t n0, int wm, int wn, int l31, int h,
                                                          int b0w = -1) {
  const int q = l31 & 3, k = l31 >> 2;
  const int64_t mw = m0 + wm * MI * 32;
  const int64_t mlast = g.M - 1;
  if constexpr (SLICE == 0) {
    const bool need_bid = g.emb || g.stats;
    {
      int64_t m = mw + l31 + 32 * h;                         // lane l asks for row mw + l
      m = m < mlast ? m : mlast;
      g2_req32(P.bid, need_bid ? (const void*)(g.bid + m) : dummy);
    }
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      int64_t n = n0 + (wn * NI + j) * 32 + 4 * k;
      n = n < g.N ? n : g.N - 4;                             // clamped: out-of-range columns are never stored
      // no bias but a time embedding (every conv1 of the U-Nets): the bias slot carries the embedding row of the
      // wave's FIRST row's batch element -- the row a wave that lies inside one batch element adds to every output
      // (g2_epilogue_finish checks that); the finish phase then has no load of its own to wait for
      const void* bp = dummy;
      if (g.bias) bp = g.bias + n;
      else if (g.emb && b0w >= 0) bp = g.emb + (int64_t)b0w * g.lde + n;
      g2_req128(P.bias[j], bp);
    }
  } else {
    constexpr int i = (SLICE - 1) / NI, j = (SLICE - 1) % NI;
    int64_t n = n0 + (wn * NI + j) * 32 + 4 * k;
    n = n < g.N ? n : g.N - 4;
#pragma unroll
    for (int G = 0; G < 4; ++G) {
      int64_t m = mw + i * 32 + q + 4 * h + 8 * G;
      m = m < mlast ? m : mlast;
      g2_req128(P.res[i][j][G], g.res ? (const void*)(g.res + m * g.ldr + n) : dummy);
    }
  }
}
template <int WM, int WN, int MI, int NI>
__device__ __forceinline__ void g2_epilogue_request(const GemmArgs& g, const void* dummy, G2Epi<MI, NI>& P, int64_t m0,
                                                    int64_t n0, int wm, int wn, int l31, int h, int b0w = -1) {
  g2_static_for<1 + MI * NI>([&](auto s_tag) {
    g2_epilogue_request_slice<WM, WN, MI, NI, decltype(s_tag)::value>(g, dummy, P, m0, n0, wm, wn, l31, h, b0w);
  });
}
// after the vmcnt(0) of the last k-steps: ties every requested register to this point of the instruction stream
__device__ __forceinline__ void g2_epilogue_landed(G2Epi<2, 2>& P) {
  asm volatile("" : "+v"(P.res[0][0][0]), "+v"(P.res[0][0][1]), "+v"(P.res[0][0][2]), "+v"(P.res[0][0][3]),
                    "+v"(P.res[0][1][0]), "+v"(P.res[0][1][1]), "+v"(P.res[0][1][2]), "+v"(P.res[0][1][3]),
                    "+v"(P.res[1][0][0]), "+v"(P.res[1][0][1]), "+v"(P.res[1][0][2]), "+v"(P.res[1][0][3]),
                    "+v"(P.res[1][1][0]), "+v"(P.res[1][1][1]), "+v"(P.res[1][1][2]), "+v"(P.res[1][1][3]),
                    "+v"(P.bias[0]), "+v"(P.bias[1]));
  asm volatile("" : "+v"(P.bid));
}
__device__ __forceinline__ void g2_epilogue_landed(G2Epi<2, 1>& P) {
  asm volatile("" : "+v"(P.res[0][0][0]), "+v"(P.res[0][0][1]), "+v"(P.res[0][0][2]), "+v"(P.res[0][0][3]),
                    "+v"(P.res[1][0][0]), "+v"(P.res[1][0][1]), "+v"(P.res[1][0][2]), "+v"(P.res[1][0][3]),
                    "+v"(P.bias[0]));
  asm volatile("" : "+v"(P.bid));
}

template <int WM, int WN, int MI, int NI>
__device__ __forceinline__ void g2_epilogue_finish(const GemmArgs& g, f32x16 (&acc)[MI][NI], G2Epi<MI, NI>& P, int64_t m0,
                                                   int64_t n0, int wm, int wn, int l31, int h, float osc,
                                                   bool emb_in_bias = false) {
  const int q = l31 & 3, k = l31 >> 2;
  const bool q0 = q & 1, q1 = q & 2;
  const int64_t mw = m0 + wm * MI * 32;
  const int64_t tile_m = m0 / (WM * MI * 32);
  // osc: inverse of the power-of-two scale the weight halves were packed with (ofx.h, range guard), read once at
  // kernel entry into an SGPR: folded into the FMA that adds the bias, so it costs no instruction.
  bool uni = true;
  int b0 = 0;
  // A wave whose 64 rows lie in TWO batch elements (the tile holds a batch boundary: ~20 tiles per launch, but with
  // persistent blocks every block is on the critical path): rows of the first element accumulate into one set of sums,
  // the rest into a second, both are reduced like the uniform case and leave as ONE fp64 atomic pair per column and
  // element.  (Until round 3 every lane flushed its own runs: ~2 000 contended atomics per wave, +30-40 k clocks for
  // the block.)  maskA: bit i * 4 + G = this lane's row (i, G) belongs to the first element.  Three or more elements
  // in one wave (a batch element with < 64 nodes at this depth) keep the per-lane path.
  bool two = false;
  int b1 = 0;
  unsigned maskA = 0xffu;
  if (g.emb || g.stats) {
    // batch id of the wave's first row = lane 0's first row (reading it here, not in the request phase, keeps a
    // scalarised load + wait out of the k-loop's tail)
    b0 = __builtin_amdgcn_readfirstlane(P.bid);
    uni = __all(P.bid == b0);                               // (rows past the end read the last row's id)
    if (!uni) {
      b1 = __builtin_amdgcn_readlane(P.bid, 63);
      two = __all(P.bid == b0 || P.bid == b1);
      if (two) {
        maskA = 0u;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int G = 0; G < 4; ++G) {
            const int bb = __shfl(P.bid, i * 32 + q + 4 * h + 8 * G);          // lane r holds the id of row mw + r
            maskA |= (bb == b0 ? 1u : 0u) << (i * 4 + G);
          }
      }
    }
  }
  auto flush_to = [&](int b, int64_t n, const float4& s, const float4& sq) {
    double* o = g.stats + ((int64_t)b * g.stats_ld + n) * 2;
    unsafeAtomicAdd(o + 0, (double)s.x); unsafeAtomicAdd(o + 1, (double)sq.x);
    unsafeAtomicAdd(o + 2, (double)s.y); unsafeAtomicAdd(o + 3, (double)sq.y);
    unsafeAtomicAdd(o + 4, (double)s.z); unsafeAtomicAdd(o + 5, (double)sq.z);
    unsafeAtomicAdd(o + 6, (double)s.w); unsafeAtomicAdd(o + 7, (double)sq.w);
  };
#define OFX_RED(f) f += dpp_xor1(f); f += dpp_xor2(f); f += __shfl_xor(f, 32);
  if (uni || two) {
    // ---- the path every tile but a handful takes.  NO load inside the store loops: on gfx9 stores count in vmcnt,
    // and a compiler-visible load anywhere between them makes hipcc wait for vmcnt(0) -- i.e. for the previous STORE to
    // be acknowledged -- before every row (the generic path below used to be interleaved with this one: 16 serialised
    // store round trips per wave, 7-12 k clocks per tile with the matrix pipe idle)
    // The two embedding rows a wave may need are asked for by inline asm and waited for by hand, only when something
    // was asked for: a compiler-visible load here would put a conservative vmcnt(0) on the common path too -- and that
    // is a wait for the NEXT tile's first DMA, which the persistent kernel has in flight by now.
    g2_v4f euA[NI], euBv[NI];
    const bool ask_a = g.emb && !emb_in_bias, ask_b = g.emb && two;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      int64_t n = n0 + (wn * NI + j) * 32 + 4 * k;
      n = n < g.N ? n : g.N - 4;
      euA[j] = g2_v4f{0.f, 0.f, 0.f, 0.f};
      euBv[j] = g2_v4f{0.f, 0.f, 0.f, 0.f};
      if (g.emb && emb_in_bias) euA[j] = P.bias[j];
      if (ask_a) g2_req128(euA[j], g.emb + (int64_t)b0 * g.lde + n);
      if (ask_b) g2_req128(euBv[j], g.emb + (int64_t)b1 * g.lde + n);
    }
    if (ask_a || ask_b) {
#pragma unroll
      for (int j = 0; j < NI; ++j)
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(euA[j]), "+v"(euBv[j])::"memory");
    }
    float4 euB[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) euB[j] = make_float4(euBv[j].x, euBv[j].y, euBv[j].z, euBv[j].w);
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int64_t n = n0 + (wn * NI + j) * 32 + 4 * k;
      const bool ncol = n < g.N;
      const float4 eu = make_float4(euA[j].x, euA[j].y, euA[j].z, euA[j].w);
      const float4 bv = g.bias ? make_float4(P.bias[j].x, P.bias[j].y, P.bias[j].z, P.bias[j].w) : f4zero();
      float4 ssum = f4zero(), ssq = f4zero();
      float4 ssumB = f4zero(), ssqB = f4zero();
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
          const bool inA = (maskA >> (i * 4 + G)) & 1u;       // (uniform wave: all ones)
          if (g.emb) f4add(v, inA ? eu : euB[j]);
          if (g.res) f4add(v, make_float4(P.res[i][j][G].x, P.res[i][j][G].y, P.res[i][j][G].z, P.res[i][j][G].w));
          if (g.stats) {
            if (inA) {
              f4add(ssum, v);
              ssq.x += v.x * v.x; ssq.y += v.y * v.y; ssq.z += v.z * v.z; ssq.w += v.w * v.w;
            } else {
              f4add(ssumB, v);
              ssqB.x += v.x * v.x; ssqB.y += v.y * v.y; ssqB.z += v.z * v.z; ssqB.w += v.w * v.w;
            }
          }
          // streaming store (`nt`): the N x Cout output (111 MB at depth 6) is next read by another kernel and would only
          // push gathered operand lines out of the XCD's 4 MB L2 -- in-run A/B on the hr step, two pairs: 8.483 / 8.470 ->
          // 8.470 / 8.460 ms, the kernel's roofline fraction 0.4005 / 0.4027 -> 0.4032 / 0.4048
          __builtin_nontemporal_store(g2_v4f{v.x, v.y, v.z, v.w}, reinterpret_cast<g2_v4f*>(g.out + m * g.ldc + n));
        }
      }
      if (g.stats) {
        OFX_RED(ssum.x) OFX_RED(ssum.y) OFX_RED(ssum.z) OFX_RED(ssum.w)
        OFX_RED(ssq.x) OFX_RED(ssq.y) OFX_RED(ssq.z) OFX_RED(ssq.w)
        if (two) {
          OFX_RED(ssumB.x) OFX_RED(ssumB.y) OFX_RED(ssumB.z) OFX_RED(ssumB.w)
          OFX_RED(ssqB.x) OFX_RED(ssqB.y) OFX_RED(ssqB.z) OFX_RED(ssqB.w)
        }
        if (q == 0 && h == 0 && ncol && mw < g.M) {
          float* o = g.stats_part ? g.stats_part + ((tile_m * WM + wm) * g.N + n) * 2 : nullptr;
          if (uni && o) {
            *reinterpret_cast<float4*>(o) = make_float4(ssum.x, ssq.x, ssum.y, ssq.y);
            *reinterpret_cast<float4*>(o + 4) = make_float4(ssum.z, ssq.z, ssum.w, ssq.w);
          } else {
            flush_to(b0, n, ssum, ssq);
            if (two) flush_to(b1, n, ssumB, ssqB);
            if (o) {                                          // mixed wave: its slot must read as zero
              *reinterpret_cast<float4*>(o) = f4zero();
              *reinterpret_cast<float4*>(o + 4) = f4zero();
            }
          }
        }
      }
    }
  } else {
    // ---- three or more batch elements inside one 64-row wave (an element with < 64 nodes at this depth): per-row
    // batch ids, every lane flushes its own runs
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int64_t n = n0 + (wn * NI + j) * 32 + 4 * k;
      const bool ncol = n < g.N;
      const float4 bv = g.bias ? make_float4(P.bias[j].x, P.bias[j].y, P.bias[j].z, P.bias[j].w) : f4zero();
      float4 ssum = f4zero(), ssq = f4zero();
      int sb = -1;
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        float4 t[4];
#pragma unroll
        for (int G = 0; G < 4; ++G) {
          float v0 = acc[i][j][4 * G], v1 = acc[i][j][4 * G + 1], v2 = acc[i][j][4 * G + 2], v3 = acc[i][j][4 * G + 3];
          quad_transpose(v0, v1, v2, v3, q0, q1);
          t[G] = make_float4(fmaf(v0, osc, bv.x), fmaf(v1, osc, bv.y), fmaf(v2, osc, bv.z), fmaf(v3, osc, bv.w));
        }
#pragma unroll
        for (int G = 0; G < 4; ++G) {
          const int64_t m = mw + i * 32 + q + 4 * h + 8 * G;
          if (m >= g.M || !ncol) continue;
          float4 v = t[G];
          const int b = g.bid[m];
          if (g.emb) f4add(v, *reinterpret_cast<const float4*>(g.emb + (int64_t)b * g.lde + n));
          if (g.res) f4add(v, make_float4(P.res[i][j][G].x, P.res[i][j][G].y, P.res[i][j][G].z, P.res[i][j][G].w));
          if (g.stats) {
            if (b != sb) {
              if (sb >= 0) flush_to(sb, n, ssum, ssq);
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
        if (sb >= 0 && ncol) flush_to(sb, n, ssum, ssq);
        if (g.stats_part && q == 0 && h == 0 && ncol && mw < g.M) {      // mixed wave: its slot must read as zero
          float* o = g.stats_part + ((tile_m * WM + wm) * g.N + n) * 2;
          *reinterpret_cast<float4*>(o) = f4zero();
          *reinterpret_cast<float4*>(o + 4) = f4zero();
        }
      }
    }
  }
#undef OFX_RED
}

// Epilogue of the dense GEMM on the planes data path (gconv3_kernel<..., ND = 1>): out = acc * oscale + bias, written as
// fp32 rows or as hi / lo pair planes (GemmArgs::out_planes) -- no residual, embedding or statistics, hence no request
// phase and none of its 73 registers.  Same lane -> element mapping as g2_epilogue_finish.
template <int WM, int WN, int MI, int NI>
__device__ __forceinline__ void g2_epilogue_dense(const GemmArgs& g, f32x16 (&acc)[MI][NI], int64_t m0, int64_t n0, int wm,
                                                  int wn, int l31, int h, float osc) {
  const int q = l31 & 3, k = l31 >> 2;
  const bool q0 = q & 1, q1 = q & 2;
  const int64_t mw = m0 + wm * MI * 32;
  float4 bv[NI];
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    int64_t n = n0 + (wn * NI + j) * 32 + 4 * k;
    n = n < g.N ? n : g.N - 4;
    bv[j] = g.bias ? *reinterpret_cast<const float4*>(g.bias + n) : f4zero();
  }
  const int pm = g.out_planes;
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int64_t n = n0 + (wn * NI + j) * 32 + 4 * k;
    const bool ncol = n < g.N;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      float4 t[4];
#pragma unroll
      for (int G = 0; G < 4; ++G) {                         // every lane takes part in the transposes
        float v0 = acc[i][j][4 * G], v1 = acc[i][j][4 * G + 1], v2 = acc[i][j][4 * G + 2], v3 = acc[i][j][4 * G + 3];
        quad_transpose(v0, v1, v2, v3, q0, q1);
        t[G] = make_float4(fmaf(v0, osc, bv[j].x), fmaf(v1, osc, bv[j].y), fmaf(v2, osc, bv[j].z), fmaf(v3, osc, bv[j].w));
      }
#pragma unroll
      for (int G = 0; G < 4; ++G) {
        const int64_t m = mw + i * 32 + q + 4 * h + 8 * G;
        if (m >= g.M || !ncol) continue;
        if (pm) ofx_store_planes4(g.out, m * g.ldc + n, t[G], pm);
        else __builtin_nontemporal_store(g2_v4f{t[G].x, t[G].y, t[G].z, t[G].w}, reinterpret_cast<g2_v4f*>(g.out + m * g.ldc + n));
      }
    }
  }
}

// libofx: fused dual-octree GraphConv on operand planes, PERSISTENT stream-K blocks (round 3).
//
// Same data path as ofx_gemm2.hip (operand planes, LDS-DMA staging with a source-side swizzle, hand-counted waits,
// bf16x3 / fp16 MFMA, two-phase epilogue) -- what changes is who computes what:
//   * the launch is exactly as many blocks as the chip holds at once (G = CUs x blocks per CU).  Work = whole-tile
//     ROUNDS + a stream-K REGION ("two-tile stream-K"): with T tiles, R = max(0, T / G - 1) rounds in which block lb
//     takes tile r * G + lb -- the blocks of one XCD work on 32 ADJACENT tiles at the same time, so the column tiles
//     of a row tile and the halo rows of Morton-neighbour tiles are shared through that XCD's L2 (with pure stream-K
//     a block walked its tiles one after the other and every column tile re-streamed its rows from HBM: 3.4 GB per
//     512 -> 512 launch against 1.1 GB, profiles/r03/pmc_traffic*.json) -- and the remaining G..2G tiles as ONE
//     sequence of U = tiles x nkt k-step units of which block b owns the contiguous range [bound(b), bound(b+1)):
//     every block does the same amount of MFMA work whatever the tile count, so the 3.31 -> 4 round quantisation of
//     the one-tile-per-block launch (17 % on the depth-6 128 -> 128 layer, 31 % on depth 5) is gone.  The region is
//     processed FIRST (its cut tiles are combined early), the rounds after it;
//   * a block walks its range tile by tile WITHOUT leaving the k-loop pipeline: the neighbour-table slice of the next
//     row tile is fetched by one LDS-DMA per wave at the start of the current tile and converted to line offsets
//     inside the last steady k-step; the two drain k-steps of a tile -- which had nothing to request -- request the
//     first two k tiles of the NEXT tile, so their gather latency runs under the epilogue's stores.  Table build,
//     first-DMA wait and store drain are paid once per block instead of once per tile;
//   * a range boundary inside a tile splits that tile between neighbouring blocks.  The block that owns the tile's
//     k = 0 end is its FINISHER: it runs the epilogue.  Every other contributor handles its piece FIRST, writes the raw
//     accumulators to its own 64 / 128 KB slot of the workspace with write-through (sc1) stores, drains, and raises
//     its flag; the finisher meets the tile LAST, adds the pieces in ascending k order (deterministic), resets the
//     flags.  Hand-off protocol = cdna_hip_programming.md section 6 Guideline 16 R1 (sc1 payload -> every wave
//     vmcnt(0) -> barrier -> one relaxed agent-scope flag store | one lane polls relaxed -> ONE agent-scope acquire ->
//     barrier -> plain loads).  No block ever waits on a block that can itself be waiting (contributors' pieces are
//     their first work item and depend on nothing), so the launch cannot deadlock as long as blocks are dispatched in
//     index order; a spin is bounded all the same and reports through the sticky word Gemm3Args::err, which
//     makes every later launch on the same sync buffer return at entry until the host has cleared it.
// Range boundaries are snapped so that no piece is shorter than G3_KMIN k-steps: the loop shape is always
// {steady..., last steady (requests the epilogue operands), drain A, drain B}.
#include "ofx_planes.h"

#ifndef G3_NSP
#define G3_NSP 5               // k-steps the epilogue operand requests of a piece are spread over (A/B: -DG3_NSP=3)
#endif
constexpr int G3_KMIN = G3_NSP + 3;     // shortest piece: one plain step + the request steps + two drain steps

template <int WM, int NI> struct G3Cfg : G2Cfg<WM, NI> {
  typedef G2Cfg<WM, NI> B;
  static constexpr int TAB = B::NBUF * B::BUF;          // table of the CURRENT row tile: [BM][8] uint32 line offsets
  static constexpr int RAW = TAB + B::BM * 32;          // raw nbr_ext slice of the NEXT row tile: BM x 7 int32; every
  static constexpr int RAW_WAVE = 32 * 28;              //   wave lands 1 KB at a 896-B pitch (+128 B overrun at the end)
  static constexpr int LDS = RAW + B::BM * 28 + 128;    // 162 944 B (WM 4) / 73 344 B (WM 2) / 56 960 B (WM 2, NI 1)
};

struct Gemm3Args {
  Gemm2Args b;
  int G;                       // blocks; a multiple of 8
  unsigned q, rem;             // stream-K region: bound(lb) = lb * q + min(lb, rem), then snapped
  unsigned U;                  // units of the stream-K region (its tiles * nkt)
  int dp_rounds;               // whole-tile rounds in front of it: tile r * G + lb belongs to block lb
  float* part;                 // [G][BM * BN] raw accumulator pieces
  unsigned* flags;             // [G], zero on entry and on exit
  unsigned* err;               // sticky error word (the LAST word of the caller's sync buffer, whatever G is): != 0 once a
                               // bounded wait gave up.  Every block of every later launch returns at entry while it is
                               // set: a contributor that was late once may still store its flag into a buffer the next
                               // launch believes to be zero, so nothing may wait on these words again until the host
                               // has noticed (ops.raise_on_sync_error), cleared them and switched launch shape
  const char* nbr_lim;         // last 16-B aligned address inside nbr_ext that may be read
  int snap;                    // 1: share boundaries snapped to the nearest legal cut position, 0: towards the tile boundary (A/B)
  int early;                   // every share spans >= one tile (a tile is cut at most once, and its second piece is
                               // published before the finisher STARTS its own): the finisher starts from that piece
                               // instead of adding it at the end
  int xcd_contig;              // 1: every XCD walks ONE contiguous range of tiles over the whole launch (its rounds and its
                               // part of the region): consecutive rounds of an XCD are neighbours in Morton order, so
                               // the halo rows and the coarse-leaf / aux rows one round pulled into that XCD's L2 are
                               // what the next round gathers again.  0: round r of XCD x = tiles r * G + x * G / 8 ...
                               // (the XCDs interleave inside every round; rounds are G tiles apart) -- A/B
  unsigned rt;                 // tiles of the stream-K region (U / nkt)
};

__device__ __forceinline__ void g3_ds_write32(unsigned addr, unsigned v) {
  asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
__device__ __forceinline__ void g3_store128_sc1(void* p, g2_v4f v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
// Share boundary of block b in the stream-K region (k-step units): the even split b * q + min(b, rem), moved to the
// nearest position a cut may sit at -- k = 0 or G3_KMIN <= k <= nkt - G3_KMIN of a tile (every piece then has the
// loop shape the kernel is written for).  `near` = 0: the first version's rule, towards the tile boundary only (shares
// of q - 7 .. q + 7 k-steps: with 30-step tiles +-7 % of a block's whole work, and every block of a persistent launch
// is on the critical path).  Host and device agree by construction (ofx_gconv3_plan, tests/test_abi.py).
__host__ __device__ inline unsigned g3_bound(unsigned b, unsigned q, unsigned rem, unsigned nkt, bool near) {
  unsigned u = b * q + (b < rem ? b : rem);
  unsigned t = u / nkt, r = u - t * nkt;
  const bool mid = near && nkt >= 2u * G3_KMIN;
  if (r < (unsigned)G3_KMIN) r = (mid && 2u * r >= (unsigned)G3_KMIN) ? (unsigned)G3_KMIN : 0u;
  else if (r + G3_KMIN > nkt) {
    if (mid && r - (nkt - G3_KMIN) <= nkt - r) r = nkt - G3_KMIN;
    else { r = 0; ++t; }
  }
  return t * nkt + r;
}
struct G3Raw { uint32_t v[4]; };
__device__ __forceinline__ void g3_wait_lgkm0(G3Raw& R) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(R.v[0]), "+v"(R.v[1]), "+v"(R.v[2]), "+v"(R.v[3])::"memory");
}

// ND: gathered "directions" per channel chunk.  7 = the dual-octree GraphConv; 1 = a plain GEMM out = A[row map] @ W on
// the same data path (round 6: GraphUpsample's [n, C] x [C, 8 C] unpool -- the register-staged dense kernel needs 62 B/clk
// of operands from L2 against the 35 B/clk the path delivers, this one shares the weight tile through LDS): the table
// has one live column (the source row of output row m), no node-type tiles, everything else is the same loop.
template <int PREC, int WM, int NI, int ND = 7>
__global__ void __launch_bounds__(512, 2) gconv3_kernel(const Gemm3Args A) {
  typedef G2Half<PREC, NI> Half;
  typedef G3Cfg<WM, NI> CF;
  constexpr int G2_NI = NI, G2_BN = CF::BN, G2_EPI_LOADS = CF::EPI_LOADS, RH = CF::READS;
  constexpr int G2_WM = WM, G2_BM = CF::BM, G2_A_BYTES = CF::A_BYTES, G2_BUF = CF::BUF, G2_TAB = CF::TAB;
  constexpr int GLDS = CF::GLDS;
  extern __shared__ __attribute__((aligned(128))) char smem3[];
  const Gemm2Args& a = A.b;
  const GemmArgs& g = a.e;

  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wid >> 1, wn = wid & 1;
  const int l31 = lane & 31, h = lane >> 5;
  const bool dbg = a.dbg != nullptr;
  unsigned long long ts0 = 0, ts1 = 0;
  int dbg_piece = 0;
  if (dbg) ts0 = g2_clock();
  // a wait of an EARLIER launch on this sync buffer gave up: its flags can no longer be trusted (see Gemm3Args::err)
  if (__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(A.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != 0) return;

  // ---- wave-uniform operands pinned in SGPRs (a kernarg s_load sunk into the loop would mix SMEM events into the
  // hand-counted lgkmcnt waits)
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
  const float osc = __builtin_bit_cast(float, sgpr32(__builtin_bit_cast(int, *g.oscale_p)));   // (epilogue only)
  const char* const xlo = a.xp < a.aux ? a.xp : a.aux;
  const int64_t wstep = (int64_t)sgpr64((uint64_t)(g.N * (int64_t)G2_LINE));   // bytes per k tile of the packed weights
  const int tpd = sgpr32(a.tpd), nkt_g = sgpr32(a.nkt_g), nkt = sgpr32(a.nkt);
  const gcp xp_s = (gcp)sgpr64((uint64_t)xlo), tfp_s = (gcp)sgpr64((uint64_t)a.tfp);
  // (operands used once per piece -- table fetch / conversion, the stream-K bookkeeping -- stay kernel arguments: the
  // compiler's own s_load + wait there costs nothing measurable and SGPRs are the scarce resource of this kernel)
  const gcp nbr_s = (gcp)a.nbr_ext, lim_s = (gcp)A.nbr_lim;
  const int ntn = g.ntn;
  const int64_t Mrows = g.M;
  const int n_src = (int)a.n_src;
  const int lpr = (int)(a.ldx >> 7), lpt = (int)(a.ldt >> 7);                           // 128-B lines per row
  const unsigned x_line = (unsigned)((a.xp - xlo) >> 7), aux_line = (unsigned)((a.aux - xlo) >> 7);
  const unsigned lds0 = (unsigned)(uintptr_t)(ldsp)smem3;    // LDS byte address of the dynamic segment

  // ---- this block's unit range.  Logical block id: XCD x (= blockIdx % 8 in dispatch order; speed only) owns the
  // contiguous logical range [x * G/8, (x + 1) * G/8), i.e. contiguous row tiles share one L2.
  const int G = A.G;
  const unsigned sk_q = A.q, sk_rem = A.rem;
  float* const part_s = A.part;
  unsigned* const flags_s = A.flags;
  const int lb = ((int)blockIdx.x & 7) * (G >> 3) + ((int)blockIdx.x >> 3);
  const bool snap_near = A.snap != 0;
  auto bound = [&](int b) -> unsigned { return g3_bound((unsigned)b, sk_q, sk_rem, (unsigned)nkt, snap_near); };
  unsigned u = (unsigned)sgpr32((int)bound(lb));
  const unsigned u_begin = u;
  const unsigned u_end = (unsigned)sgpr32((int)bound(lb + 1));
  const int dp_rounds = A.dp_rounds;
  const unsigned dp_tiles = (unsigned)dp_rounds * (unsigned)G;     // tiles [0, dp_tiles) are the rounds, the rest the region
  int dp_r = 0;                                                    // next round of this block
  if (u >= u_end && dp_rounds == 0) return;                  // (cannot happen for U / G >= 2 * G3_KMIN; kept for safety)

  // ---- per-lane DMA source state (as gconv2)
  const int q8 = lane & 7, rsub = lane >> 3;
  const int swz0 = (lane >> 4) & 7, swz1 = (4 + (lane >> 4)) & 7;
  const int pa0 = (q8 ^ swz0) * 16, pa1 = (q8 ^ swz1) * 16;
  const unsigned tab_lane = lds0 + G2_TAB + (wid * 32 + rsub) * 32;
  constexpr int BPW = CF::B_PER_WAVE;
  gcp wb[4];
  auto set_wb = [&](int64_t n0) {                 // (once per piece: lane arithmetic on an opaque id, see opaque_tid)
    const int64_t Nc = g.N;
    int ln = threadIdx.x & 63;
    asm volatile("" : "+v"(ln));
    const int q8_ = ln & 7, rsub_ = ln >> 3;
#pragma unroll
    for (int j = 0; j < BPW; ++j) {
      int64_t c = n0 + wid * (8 * BPW) + j * 8 + rsub_;
      c = c < Nc ? c : Nc - 1;
      const int brow = wid * (8 * BPW) + j * 8 + rsub_;
      wb[j] = (gcp)a.W2 + c * G2_LINE + (q8_ ^ ((brow >> 1) & 7)) * 16;
    }
  };

  struct Tile { int tcol; int ktw; gcp base; };
  auto load_idx = [&](const Tile& T, G2Idx& I) {
    const unsigned ad = tab_lane + T.tcol * 4;
    g2_ds_read32<0>(I.v[0], ad);
    g2_ds_read32<256>(I.v[1], ad);
    g2_ds_read32<512>(I.v[2], ad);
    g2_ds_read32<768>(I.v[3], ad);
  };
  auto issue = [&](const Tile& T, int ob, const G2Idx& I) {
    const gcp b0 = T.base + pa0, b1 = T.base + pa1;
    char* const abuf = smem3 + ob + wid * (32 * G2_LINE);
#pragma unroll
    for (int j = 0; j < 4; ++j)
      __builtin_amdgcn_global_load_lds(((j & 1) ? b1 : b0) + ((uint64_t)I.v[j] << 7),
                                       (ldsp)(abuf + j * (8 * G2_LINE)), 16, 0, 0);
    char* const bbuf = smem3 + ob + G2_A_BYTES + wid * (8 * BPW * G2_LINE);
    const int64_t wo = (int64_t)T.ktw * wstep;
#pragma unroll
    for (int j = 0; j < BPW; ++j)
      __builtin_amdgcn_global_load_lds(wb[j] + wo, (ldsp)(bbuf + j * (8 * G2_LINE)), 16, 0, 0);
  };
  auto tile_of = [&](int it) {
    Tile T;
    if (it < nkt_g) {
      const int chunk = it / ND, dir = it - chunk * ND;
      T.tcol = dir; T.ktw = dir * tpd + chunk; T.base = xp_s + (int64_t)chunk * G2_LINE;
    } else {
      T.tcol = 7; T.ktw = it; T.base = tfp_s + (int64_t)(it - nkt_g) * G2_LINE;
    }
    return T;
  };

  // Everything these per-piece sections derive from the thread id is loop-invariant, and the compiler would compute
  // it once in front of the piece loop and keep it (in registers it does not have: spills) across every k-step.
  // An opaque copy of the thread id keeps the arithmetic local to its section.
  auto opaque_tid = [&]() {
    int t = threadIdx.x;
    asm volatile("" : "+v"(t));
    return t;
  };
  // ---- neighbour table of a row tile, two phases:
  //   raw_request(m0n): every wave DMAs the 32 x 7 int32 entries of its rows (896 B; the instruction moves 1 KB, the
  //     128-B tail duplicates the next wave's head) -- addresses clamped to the end of the array;
  //   table_convert(m0n): entry (r, d < 7) -> unsigned 128-B line offset from xlo of source row nbr_ext[m, d]
  //     (rows >= n_src live in `aux`), entry (r, 7) -> line offset of the row's own node-type slab row from tfp.
  auto raw_request = [&](int64_t m0n) {
    gcp p = nbr_s + (m0n + wid * 32) * 28 + (opaque_tid() & 63) * 16;
    p = p < lim_s ? p : lim_s;
    __builtin_amdgcn_global_load_lds(p, (ldsp)(smem3 + CF::RAW + wid * CF::RAW_WAVE), 16, 0, 0);
  };
  auto table_convert = [&](int64_t m0n) {
    const int tid = opaque_tid();
    const int64_t left = Mrows - 1 - m0n;                               // last valid row of the tile, relative (>= 0)
    const int rmax = left < (int64_t)(G2_BM - 1) ? (int)left : G2_BM - 1;
    const unsigned mline = (unsigned)m0n * (unsigned)lpt;
    G3Raw R;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int e = tid + CF::THREADS * t;
      const int r = e >> 3, d = e & 7;
      const int rl = r < rmax ? r : rmax;
      const unsigned ad = lds0 + CF::RAW + (rl * 7 + (d < 7 ? d : 6)) * 4;
      if (t == 0) g2_ds_read32<0>(R.v[0], ad);
      else if (t == 1) g2_ds_read32<0>(R.v[1], ad);
      else if (t == 2) g2_ds_read32<0>(R.v[2], ad);
      else g2_ds_read32<0>(R.v[3], ad);
    }
    g3_wait_lgkm0(R);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int e = tid + CF::THREADS * t;
      const int r = e >> 3, d = e & 7;
      const int rl = r < rmax ? r : rmax;
      const int id = (int)R.v[t];
      const unsigned gl = id < n_src ? x_line + (unsigned)id * (unsigned)lpr
                                     : aux_line + (unsigned)(id - n_src) * (unsigned)lpr;
      g3_ds_write32(lds0 + G2_TAB + e * 4, d < 7 ? gl : mline + (unsigned)rl * (unsigned)lpt);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  };

  // ---- per-lane fragment read state (as gconv2)
  const int s7 = (l31 >> 1) & 7;
  unsigned fa[2][2], fb[2][2];
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int uu = 0; uu < 2; ++uu) {
      const int t = g2_three_term<PREC>() ? (uu == 0 ? c : 2 + c) : 2 * c + uu;
      const int po = ((2 * t + h) ^ s7) * 16;
      fa[c][uu] = lds0 + (wm * 64 + l31) * G2_LINE + po;
      fb[c][uu] = lds0 + G2_A_BYTES + (wn * (32 * G2_NI) + l31) * G2_LINE + po;
    }
  auto read_half = [&](int ob, int c, Half& F) {
#pragma unroll
    for (int uu = 0; uu < 2; ++uu) {
      g2_ds_read128<0>(F.a[uu][0], fa[c][uu] + ob);
      g2_ds_read128<32 * G2_LINE>(F.a[uu][1], fa[c][uu] + ob);
    }
#pragma unroll
    for (int uu = 0; uu < 2; ++uu) {
      g2_ds_read128<0>(F.b[uu][0], fb[c][uu] + ob);
      if constexpr (NI == 2) g2_ds_read128<32 * G2_LINE>(F.b[uu][1], fb[c][uu] + ob);
    }
  };
  auto read_one = [&](int ob, int c, Half& F, auto r_tag) {
    constexpr int r = decltype(r_tag)::value;
    constexpr int NA = 2 * G2_MI;
    if constexpr (r < NA) {
      constexpr int uu = r / G2_MI, i = r % G2_MI;
      if constexpr (i == 0) g2_ds_read128<0>(F.a[uu][0], fa[c][uu] + ob);
      else g2_ds_read128<32 * G2_LINE>(F.a[uu][1], fa[c][uu] + ob);
    } else {
      constexpr int qq = r - NA, uu = qq / G2_NI, j = qq % G2_NI;
      if constexpr (j == 0) g2_ds_read128<0>(F.b[uu][0], fb[c][uu] + ob);
      else g2_ds_read128<32 * G2_LINE>(F.b[uu][1], fb[c][uu] + ob);
    }
  };
  f32x16 acc[G2_MI][G2_NI];
  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < G2_MI; ++i)
#pragma unroll
      for (int j = 0; j < G2_NI; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  };

#define G3_FENCE() __builtin_amdgcn_sched_barrier(0)
  Half F0, F1;
  G2Idx I;
  auto mfma_one = [&](const Half& F, auto m_tag) {
    constexpr int m = decltype(m_tag)::value;
    constexpr int per = G2_MI * G2_NI, t = m / per, i = (m / G2_NI) % G2_MI, j = m % G2_NI;
    if constexpr (g2_three_term<PREC>()) {
      // [0] = hi, [1] = lo: small cross terms first, the leading term last
      if constexpr (t == 0) acc[i][j] = g2_mfma<PREC>(F.a[1][i], F.b[0][j], acc[i][j]);
      else if constexpr (t == 1) acc[i][j] = g2_mfma<PREC>(F.a[0][i], F.b[1][j], acc[i][j]);
      else acc[i][j] = g2_mfma<PREC>(F.a[0][i], F.b[0][j], acc[i][j]);
    } else {
      acc[i][j] = g2_mfma<PREC>(F.a[t][i], F.b[t][j], acc[i][j]);
    }
  };
  auto issue_one = [&](const Tile& T, int ob, const G2Idx& I, auto k_tag) {
    constexpr int k = decltype(k_tag)::value;
    if constexpr (k < 4) {
      const gcp b = T.base + ((k & 1) ? pa1 : pa0);
      __builtin_amdgcn_global_load_lds(b + ((uint64_t)I.v[k] << 7),
                                       (ldsp)(smem3 + ob + wid * (32 * G2_LINE) + k * (8 * G2_LINE)), 16, 0, 0);
    } else {
      constexpr int j = k - 4;
      __builtin_amdgcn_global_load_lds(wb[j] + (int64_t)T.ktw * wstep,
                                       (ldsp)(smem3 + ob + G2_A_BYTES + wid * (8 * BPW * G2_LINE) + j * (8 * G2_LINE)),
                                       16, 0, 0);
    }
  };
  constexpr int NMF = (g2_three_term<PREC>() ? 3 : 2) * G2_MI * G2_NI;           // MFMAs per half step
  // MFMAs of half Fc with (a) the LDS reads of the next half set Fr and (b) optionally the DMA request of tile T
  // spliced between them in a pinned order (gconv2's "variant 5").
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
        if constexpr (m == 1) {                                          // table entries: 2 younger reads so far
          if (do_read) g2_wait_lgkm<(RH < 2 ? RH : 2)>(I);
          else g2_wait_lgkm<0>(I);
        }
        if constexpr (m >= 2 && m - 2 < GLDS) issue_one(T, ob_dma, I, std::integral_constant<int, m - 2>());
      }
      G3_FENCE();
    };
    g2_static_for<NMF>(body);
    if constexpr (RH > NMF) {
      g2_static_for<RH>([&](auto r_tag) {
        constexpr int r = decltype(r_tag)::value;
        if constexpr (r >= NMF) {
          if (do_read) read_one(ob_r, c_r, Fr, r_tag);
        }
      });
    }
    if constexpr (DMA) {
      g2_static_for<GLDS>([&](auto k_tag) {
        constexpr int k = decltype(k_tag)::value;
        if constexpr (k + 2 >= NMF) issue_one(T, ob_dma, I, k_tag);
      });
    }
  };

  // ================================ block prologue: first piece ==============================================
  bool in_region = u < u_end;                                 // this piece is part of the stream-K region
  unsigned t_cur = in_region ? dp_tiles + u / (unsigned)nkt : (unsigned)lb;
  int k0 = in_region ? (int)(u % (unsigned)nkt) : 0;
  if (!in_region) dp_r = 1;
  // logical tile index (what the schedule above hands out) -> tile of the layer.  xcd_contig: XCD x owns the tiles
  // [cb(x), cb(x + 1)), cb(x) = x * R * G/8 + floor(x * RT / 8): first its R whole-tile rounds of G/8 tiles, then its
  // eighth of the region (the region's units are cut into G equal shares in logical order, so the shares of XCD x's
  // blocks fall into the x-th eighth of the region's tiles up to a tile at either end) -- a bijection on [0, T).
  // (256-row geometry only: the layers with whole-tile rounds; the 128-row instantiations have no register to spare)
  const bool contig = WM == 4 && A.xcd_contig != 0 && A.rt >= 8u;
  auto phys = [&](unsigned s) -> unsigned {
    if (WM != 4 || !contig) return s;
    const unsigned G8 = (unsigned)G >> 3, RT = A.rt, R = (unsigned)dp_rounds;
    if (s < dp_tiles) {
      const unsigned r = s / (unsigned)G, l = s - r * (unsigned)G, x = l / G8, j = l - x * G8;
      return x * R * G8 + (x * RT >> 3) + r * G8 + j;
    }
    const unsigned q = s - dp_tiles;
    unsigned x = (8u * q + 7u) / RT;              // the largest x with floor(x * RT / 8) <= q
    x = x > 7u ? 7u : x;
    return x * R * G8 + (x * RT >> 3) + R * G8 + (q - (x * RT >> 3));
  };
  const unsigned t_phys0 = phys(t_cur);
  int tm = (int)(t_phys0 / (unsigned)ntn), tn = (int)t_phys0 - tm * ntn;
  int64_t m0 = a.row0 + (int64_t)tm * G2_BM, n0 = (int64_t)tn * G2_BN;
  raw_request(m0);
  g2_wait_barrier<0>();
  table_convert(m0);
  set_wb(n0);
  asm volatile("s_barrier" ::: "memory");
  {
    const Tile T0 = tile_of(k0);
    load_idx(T0, I);
    g2_wait_lgkm<0>(I);
    issue(T0, 0, I);
    const Tile T1 = tile_of(k0 + 1);                                    // every piece has >= G3_KMIN k tiles
    load_idx(T1, I);
    g2_wait_lgkm<0>(I);
    issue(T1, G2_BUF, I);
  }
  int ob = 0, obn = G2_BUF, obnn = 2 * G2_BUF;                           // stage buffers of k tiles it, it+1, it+2
  G2Epi<G2_MI, G2_NI> P;
  if (dbg) ts1 = g2_clock();

  bool flag_pending = false;
  const bool early = A.early != 0;
  // one lane polls the flags of the next `n` blocks (relaxed), ONE agent-scope acquire, everybody meets
  auto wait_flags = [&](int n) {
    if (threadIdx.x == 0) {
      const unsigned long long tstart = g2_clock();
      for (int c = lb + 1; c <= lb + n; ++c) {
        while (__hip_atomic_load(flags_s + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
          __builtin_amdgcn_s_sleep(8);
          if (g2_clock() - tstart > (1ull << 33)) {                     // several seconds: give up, report, go on
            __hip_atomic_store(A.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  };

  // ================================ piece loop ===============================================================
  for (;;) {
    // this piece = k tiles [k0, k1) of tile t_cur; the next one starts at k = 0 of: the following tile (inside the
    // region), the block's first whole-tile round (after the region), its next round
    int k1 = nkt;
    bool has_next, next_in_region = false;
    unsigned t_nxt;
    const unsigned u_before = u;
    if (in_region) {
      const unsigned left = u_end - u;
      k1 = (unsigned)(nkt - k0) <= left ? nkt : k0 + (int)left;
      u += (unsigned)(k1 - k0);
      next_in_region = u < u_end;
      has_next = next_in_region || dp_rounds > 0;
      t_nxt = next_in_region ? t_cur + 1 : (unsigned)lb;
    } else {
      has_next = dp_r < dp_rounds;
      t_nxt = (unsigned)dp_r * (unsigned)G + (unsigned)lb;
    }
    const int len = k1 - k0;
    const unsigned t_nxt_p = phys(t_nxt);
    const int tm_n = (int)(t_nxt_p / (unsigned)ntn), tn_n = (int)t_nxt_p - tm_n * ntn;
    const int64_t m0_n = a.row0 + (int64_t)tm_n * G2_BM, n0_n = (int64_t)tn_n * G2_BN;
    const bool new_rows = has_next && tm_n != tm;
    const bool finisher = k0 == 0;                                       // (always, outside the region)

    const bool cut_head = finisher && k1 < nkt;                          // k tiles [k1, nkt) come from the next block(s)
    // start from the published piece only if it has (very probably) been published: its block started with it when
    // this block started, so it is done once this block has worked through more k-steps than the piece is long.
    // Otherwise add it at the end (decided from the schedule, not from the flag: the order of the additions -- and
    // with it the last bits of the result -- must not depend on timing)
    const bool early_now = early && (int)(u_before - u_begin) > nkt - k1;

    // batch element of this wave's first output row, for the time-embedding line of the epilogue requests (asked for
    // here so that its latency is this wait's, not the last k-steps')
    int b0v = 0;
    const bool emb_line = ND != 1 && g.emb && !g.bias;
    if (emb_line) {
      const int64_t mw = m0 + (wid >> 1) * (G2_MI * 32);
      g2_req32(b0v, g.bid + (mw < Mrows - 1 ? mw : Mrows - 1));
    }
    // all DMA of k tiles k0, k0+1 and every older store have landed / drained; meet, then fetch the first fragments
    g2_wait_barrier<0>();
    asm volatile("" : "+v"(b0v));
    const int b0w = emb_line ? __builtin_amdgcn_readfirstlane(b0v) : -1;
    if (flag_pending) {                                                  // the piece published before this one: every wave has drained
      if (threadIdx.x == 0) __hip_atomic_store(flags_s + lb, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      flag_pending = false;
    }
    if (cut_head && early_now) {
      // start from the other piece of the tile (published long ago: it was its block's first work) instead of zero:
      // its 16 loads per lane go straight into the accumulators and fly together
      wait_flags(1);
      const char* src = reinterpret_cast<const char*>(part_s + (size_t)(lb + 1) * (G2_BM * G2_BN)) + opaque_tid() * 16;
#pragma unroll
      for (int i = 0; i < G2_MI; ++i)
#pragma unroll
        for (int j = 0; j < G2_NI; ++j)
#pragma unroll
          for (int r4 = 0; r4 < 4; ++r4) {
            const g2_v4f v = *reinterpret_cast<const g2_v4f*>(src + ((i * G2_NI + j) * 4 + r4) * (CF::THREADS * 16));
            acc[i][j][4 * r4 + 0] = v.x; acc[i][j][4 * r4 + 1] = v.y;
            acc[i][j][4 * r4 + 2] = v.z; acc[i][j][4 * r4 + 3] = v.w;
          }
      asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");     // every wave has read its slot
      if (threadIdx.x == 0) __hip_atomic_store(flags_s + lb + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      zero_acc();
    }
    read_half(ob, 0, F0);

    // k tile it + 2 of the piece, advanced without division while it is a gather tile
    int it = k0;
    int gd, gc;
    {
      const int ti = k0 + 2;
      gc = ti / ND; gd = ti - gc * ND;                                   // (only meaningful while ti < nkt_g)
    }
    auto next_tile = [&]() {
      const int ti = it + 2;
      const bool gat = ti < nkt_g;
      Tile T;
      T.tcol = gat ? gd : 7;
      T.ktw = gat ? gd * tpd + gc : ti;
      T.base = (gat ? xp_s : tfp_s) + (int64_t)(gat ? gc : ti - nkt_g) * G2_LINE;
      const int wrap = gd == ND - 1;
      gd = wrap ? 0 : gd + 1;
      gc += wrap;
      return T;
    };
    // epilogue operand requests, slices [LO, HI) of 1 + MI * NI (ofx_planes.h)
    auto epi_request = [&](auto lo_tag, auto hi_tag) {
      constexpr int LO = decltype(lo_tag)::value, HI = decltype(hi_tag)::value;
      const int tid = opaque_tid();
      const int w = __builtin_amdgcn_readfirstlane(tid >> 6), ln = tid & 63;
      g2_static_for<HI - LO>([&](auto s_tag) {
        g2_epilogue_request_slice<G2_WM, G2_WN, G2_MI, G2_NI, LO + decltype(s_tag)::value>(
            g, (const void*)a.W2, P, m0, n0, w >> 1, w & 1, ln & 31, ln >> 5, b0w);
      });
    };
    // One steady k-step.  On entry: F0 = first half of k tile `it` (reads possibly in flight), k tile it+1 requested.
    // EPI slices [LO, HI) are requested in this step (LO == HI: none); their loads may stay in flight at its barrier.
    auto step = [&](const Tile& T, auto lo_tag, auto hi_tag) {
      constexpr int LO = decltype(lo_tag)::value, HI = decltype(hi_tag)::value;
      constexpr bool LAST = HI > LO;
      // (ND = 1, the dense GEMM: no residual / embedding / statistics operands -> nothing is requested)
      constexpr int NEPI = ND == 1 ? 0 : (LO < 1 && HI > 0 ? g2_epi_slice_loads<G2_NI>(0) : 0) + 4 * ((HI > 1 ? HI : 1) - (LO > 1 ? LO : 1));
      if constexpr (CF::NBUF == 3) {
        // table reads | wait F0 (4 younger) | {MFMA F0, read F1, request k tile it+2} | barrier | {MFMA F1, read F0'}
        load_idx(T, I);
        g2_wait_lgkm<4, PREC>(F0);
        G3_FENCE();
        if constexpr (LAST && ND != 1) {
          epi_request(lo_tag, hi_tag);
          G3_FENCE();
        }
        mfma_spliced(F0, true, ob, 1, F1, std::true_type(), T, obnn);
        g2_wait_barrier<GLDS + NEPI, PREC>(F1);
        G3_FENCE();
        mfma_spliced(F1, true, obn, 0, F0, std::false_type(), T, 0);
        const int t = ob; ob = obn; obn = obnn; obnn = t;
      } else {
        // wait F0 | {MFMA F0, read F1} | barrier | table reads | {MFMA F1, read F0', request k tile it+2 into the
        // buffer k tile `it` just left}
        g2_wait_lgkm<0, PREC>(F0);
        G3_FENCE();
        if constexpr (LAST && ND != 1) {
          epi_request(lo_tag, hi_tag);
          G3_FENCE();
        }
        mfma_spliced(F0, true, ob, 1, F1, std::false_type(), T, 0);
        g2_wait_barrier<NEPI, PREC>(F1);
        G3_FENCE();
        load_idx(T, I);
        mfma_spliced(F1, true, obn, 0, F0, std::true_type(), T, ob);
        const int t = ob; ob = obn; obn = t;
      }
    };
    // A drain k-step (the last two of a piece): nothing of THIS piece is left to request; `T` = k tile 0 / 1 of the
    // next piece, requested in the second half through the table converted at the end of the last steady step.  The
    // block's last piece requests them all the same (through the table and weight columns it has: valid addresses,
    // into stage buffers nobody reads again) -- two wasted k tiles per block instead of a branch around every request.
    // VMW: DMA instructions that may stay in flight at the barrier.
    auto drain = [&](const Tile& T, bool do_read, auto vmw_tag) {
      constexpr int VMW = decltype(vmw_tag)::value;
      g2_wait_lgkm<0, PREC>(F0);
      G3_FENCE();
      mfma_spliced(F0, true, ob, 1, F1, std::false_type(), T, 0);
      g2_wait_barrier<VMW, PREC>(F1);
      G3_FENCE();
      load_idx(T, I);
      mfma_spliced(F1, do_read, obn, 0, F0, std::true_type(), T, CF::NBUF == 3 ? obnn : ob);
      if (CF::NBUF == 3) { const int t = ob; ob = obn; obn = obnn; obnn = t; }
      else { const int t = ob; ob = obn; obn = t; }
    };

    // ---- steady steps
    // Steps [k0, k1 - 2): plain ones, then the ones that carry the epilogue requests -- one slice per step over the
    // last NSL steps when the piece is long enough (each slice then has a step and a half to land, like a k tile),
    // else all of them in the last step.
    // Steps [k0, k1 - 2): plain ones, then NSP steps that carry the epilogue requests, one (or two) slices each: every
    // slice has a step and a half to land, like a k tile, instead of 128 KB of residual rows arriving behind one wait.
    // (ONE code path: a second, all-in-one-step path for short pieces made the register allocator spill the request
    // registers -- and a compiler spill of an asm-loaded register can be stored before its data has landed.  Hence
    // G3_KMIN = G3_NSP + 3: no piece is shorter.)
    constexpr int NSL = 1 + G2_MI * G2_NI;                              // request slices (ofx_planes.h)
    constexpr int NSP = NSL < G3_NSP ? NSL : G3_NSP;                    // steps they are spread over
    typedef std::integral_constant<int, 0> Z0;
    const int plain_end = k1 - 2 - NSP;                                 // >= k0 + 1
    // (the first plain step always exists; written apart so that the compiler sees it on every path: its wait for
    // accumulators that were initialised by loads then never lands in a later step, where it would drain the DMA queue)
    step(next_tile(), Z0(), Z0());
    ++it;
    if (new_rows) raw_request(m0_n);                                    // one more DMA, older than everything the next waits leave in flight
    for (; it < plain_end; ++it) step(next_tile(), Z0(), Z0());
    g2_static_for<NSP>([&](auto s_tag) {
      constexpr int S = decltype(s_tag)::value;
      constexpr int lo = S * NSL / NSP, hi = (S + 1) * NSL / NSP;       // an even share of the slices
      step(next_tile(), std::integral_constant<int, lo>(), std::integral_constant<int, hi>());
      ++it;
    });
    if (new_rows) {
      // 3 stages: every wave read the old table before this step's barrier.  2 stages: the table reads sit in the
      // second half of the step -- meet once more before overwriting it.
      if (CF::NBUF == 2) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      table_convert(m0_n);
    }
    if (has_next) set_wb(n0_n);
    {
      Tile Tn = tile_of(0);
      drain(Tn, true, std::integral_constant<int, 0>());      // k tile k1 - 1 and the epilogue operands land
      ++it;
      Tn = tile_of(1);
      drain(Tn, false, std::integral_constant<int, GLDS>());
      ++it;
    }
    if constexpr (ND != 1) g2_epilogue_landed(P);
    if (dbg && dbg_piece < 6 && threadIdx.x == 0) a.dbg[(size_t)blockIdx.x * 16 + 4 + 2 * dbg_piece] = g2_clock();

    // ---- the piece's result
    if (finisher) {
      if (cut_head && !early_now) {
        // small layers (several blocks per tile): the other pieces are computed at the same time as this one, so they
        // are added at the end, in ascending k order
        const unsigned u_tile_end = (t_cur - dp_tiles + 1) * (unsigned)nkt;    // (region units)
        int nc = 0;
        for (int c = lb + 1; c < G && bound(c) < u_tile_end; ++c) ++nc;
        wait_flags(nc);
        for (int c = lb + 1; c <= lb + nc; ++c) {
          const char* src = reinterpret_cast<const char*>(part_s + (size_t)c * (G2_BM * G2_BN)) + opaque_tid() * 16;
          asm volatile("" : "+v"(src));                                  // keep the address arithmetic inside this branch
#pragma unroll
          for (int i = 0; i < G2_MI; ++i) {
            g2_v4f v[G2_NI][4];
#pragma unroll
            for (int j = 0; j < G2_NI; ++j)
#pragma unroll
              for (int r4 = 0; r4 < 4; ++r4) {
                v[j][r4] = *reinterpret_cast<const g2_v4f*>(src);
                src += CF::THREADS * 16;
                asm volatile("" : "+v"(src));
              }
#pragma unroll
            for (int j = 0; j < G2_NI; ++j)
#pragma unroll
              for (int r4 = 0; r4 < 4; ++r4) {
                acc[i][j][4 * r4 + 0] += v[j][r4].x; acc[i][j][4 * r4 + 1] += v[j][r4].y;
                acc[i][j][4 * r4 + 2] += v[j][r4].z; acc[i][j][4 * r4 + 3] += v[j][r4].w;
              }
            asm volatile("" ::: "memory");                               // eight loads in flight at a time (registers)
          }
        }
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");    // every wave has read its slots
        if (threadIdx.x == 0)
          for (int c = lb + 1; c <= lb + nc; ++c)
            __hip_atomic_store(flags_s + c, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      {
        const int tid = opaque_tid();
        const int w = __builtin_amdgcn_readfirstlane(tid >> 6), ln = tid & 63;
        if constexpr (ND == 1) g2_epilogue_dense<G2_WM, G2_WN, G2_MI, G2_NI>(g, acc, m0, n0, w >> 1, w & 1, ln & 31, ln >> 5, osc);
        else g2_epilogue_finish<G2_WM, G2_WN, G2_MI, G2_NI>(g, acc, P, m0, n0, w >> 1, w & 1, ln & 31, ln >> 5, osc, emb_line);   // (vec4 only: host-checked)
      }
    } else {
      // a middle / tail piece: publish the raw accumulators (lane-linear float4 slabs), write-through
      char* dst = reinterpret_cast<char*>(part_s + (size_t)lb * (G2_BM * G2_BN)) + opaque_tid() * 16;
      asm volatile("" : "+v"(dst));                                      // (loop-invariant otherwise: hoisted and spilled)
#pragma unroll
      for (int i = 0; i < G2_MI; ++i)
#pragma unroll
        for (int j = 0; j < G2_NI; ++j)
#pragma unroll
          for (int r4 = 0; r4 < 4; ++r4) {
            g2_v4f v;
            v.x = acc[i][j][4 * r4 + 0]; v.y = acc[i][j][4 * r4 + 1];
            v.z = acc[i][j][4 * r4 + 2]; v.w = acc[i][j][4 * r4 + 3];
            g3_store128_sc1(dst, v);
            dst += CF::THREADS * 16;
            asm volatile("" : "+v"(dst));
          }
      // the flag follows once EVERY storing wave has drained and the block has met: that is what the next piece's
      // first wait does anyway (a block's only piece: right here)
      flag_pending = true;
      if (!has_next) {
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        if (threadIdx.x == 0) __hip_atomic_store(flags_s + lb, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    if (dbg) {
      if (dbg_piece < 6 && threadIdx.x == 0) a.dbg[(size_t)blockIdx.x * 16 + 5 + 2 * dbg_piece] = g2_clock();
      ++dbg_piece;
    }
    if (!has_next) break;
    if (!next_in_region) { in_region = false; ++dp_r; }
    t_cur = t_nxt; k0 = 0; tm = tm_n; tn = tn_n; m0 = m0_n; n0 = n0_n;
  }
#undef G3_FENCE
  // the last piece's drain steps requested two k tiles nobody reads: they must have landed before the LDS is released
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (dbg) {
    // [blocks][16] shader-clock stamps: entry, first piece ready, pieces, exit; then per piece (<= 6) k-loop done /
    // result written
    const unsigned long long ts4 = g2_clock();
    if (threadIdx.x == 0) {
      unsigned long long* o = a.dbg + (size_t)blockIdx.x * 16;
      o[0] = ts0; o[1] = ts1; o[2] = (unsigned long long)dbg_piece; o[3] = ts4;
    }
  }
}

// ------------------------------------------------------------------------------------------------
template <int PREC, int WM, int NI, int ND = 7>
static int g3_launch(const Gemm3Args& A, hipStream_t st) {
  static bool attr_set[OFX_MAX_DEVICES] = {};
  if (!ofx_raise_lds_limit(reinterpret_cast<const void*>(&gconv3_kernel<PREC, WM, NI, ND>), G3Cfg<WM, NI>::LDS, attr_set))
    return OFX_ELAUNCH;
  gconv3_kernel<PREC, WM, NI, ND><<<A.G, G3Cfg<WM, NI>::THREADS, G3Cfg<WM, NI>::LDS, st>>>(A);
  return OFX_OK;
}

static int g3_cus_override = 0;        // > 0: plan persistent launches for at most this many compute units (ofx_set_gconv_cus)
static int g3_cus() {                  // compute units of the current device (cached per device)
  static int cus[OFX_MAX_DEVICES] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= OFX_MAX_DEVICES) return 0;
  if (!cus[dev]) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    cus[dev] = n;
  }
  // never MORE blocks than the device can keep resident: the hand-off waits rely on it
  return g3_cus_override > 0 && g3_cus_override < cus[dev] ? g3_cus_override : cus[dev];
}

// Bytes of workspace the persistent launch needs behind the statistics partials (0: the shape is not eligible).
// Launch plan: wm (2 / 4), ni (1 / 2) -> blocks G, units per block.
struct G3Plan { int G; unsigned q, rem, U; int dp_rounds; size_t part_bytes; };
static int g3_snap = 1;        // 1: nearest legal cut position; 0: towards the tile boundary (A/B, ofx_set_gconv_persistent(3))
static int g3_xcd_contig = 0;  // tile order: 1 = one contiguous tile range per XCD (Gemm3Args::xcd_contig), A/B: ofx_set_gconv_xcd_contig
static int g3_hybrid = 1;      // 1: whole-tile rounds + stream-K region; 0: pure stream-K (A/B, ofx_set_gconv_persistent(2))
static bool g3_plan(int64_t M, int cout, int nkt, int wm, int ni, G3Plan& p, int cus = 0) {
  if (cus <= 0) cus = g3_cus();
  if (cus < 8 || nkt < G3_KMIN) return false;
  const int64_t tiles = ofx_cdiv(M, wm * 64) * ofx_cdiv(cout, 64 * ni);
  if (tiles * nkt >= (1ll << 31)) return false;
  int64_t slots = ((int64_t)cus * (wm == 4 ? 1 : 2)) & ~7ll;
  int64_t G, region_tiles;
  int rounds = 0;
  if (tiles >= slots) {
    // two-tile stream-K: the last (tiles mod G) + G tiles form the region, everything before it whole-tile rounds
    G = slots;
    rounds = g3_hybrid ? (int)(tiles / G - 1) : 0;
    region_tiles = tiles - (int64_t)rounds * G;
  } else {
    // fewer tiles than block slots: the whole layer is the region, in shares of >= 2 * G3_KMIN units (after snapping
    // every piece still has >= G3_KMIN k-steps)
    const int64_t U = tiles * nkt;
    G = slots < U / (2 * G3_KMIN) ? slots : U / (2 * G3_KMIN);
    G &= ~7ll;
    region_tiles = tiles;
  }
  if (G < 8) return false;
  const int64_t U = region_tiles * nkt;
  p.G = (int)G; p.U = (unsigned)U; p.q = (unsigned)(U / G); p.rem = (unsigned)(U % G); p.dp_rounds = rounds;
  p.part_bytes = (size_t)G * (size_t)(wm * 64) * (size_t)(64 * ni) * sizeof(float);
  return true;
}

void ofx_gconv3_set_hybrid(int on) { g3_hybrid = on ? 1 : 0; }
extern "C" int ofx_set_gconv_xcd_contig(int on) { g3_xcd_contig = on ? 1 : 0; return OFX_OK; }
extern "C" int ofx_set_gconv_cus(int cus) {
  if (cus < 0 || (cus > 0 && cus < 8)) return OFX_EINVAL;
  g3_cus_override = cus;
  return OFX_OK;
}
void ofx_gconv3_set_snap(int near) { g3_snap = near ? 1 : 0; }

// The schedule of a persistent launch, on the host (no device work; `cus` > 0: plan for that many compute units
// without asking a device): out[0..4] = blocks G, q, rem, region units U, whole-tile rounds; out[5 .. 5 + G] = the
// region's share boundaries bound(0..G) as the kernel computes them.  Returns G, or 0 when the shape does not
// qualify for the persistent launch (then nothing is written).  include/ofx.h.
extern "C" int ofx_gconv3_plan(int64_t n_rows, int cout, int nkt, int wm, int ni, int cus, int32_t* out, int64_t out_len) {
  G3Plan p;
  if (n_rows <= 0 || cout <= 0 || nkt <= 0 || (wm != 2 && wm != 4) || (ni != 1 && ni != 2)) return 0;
  if (!g3_plan(n_rows, cout, nkt, wm, ni, p, cus)) return 0;
  if (out) {
    if (out_len < 5 + (int64_t)p.G + 1) return 0;
    out[0] = p.G; out[1] = (int32_t)p.q; out[2] = (int32_t)p.rem; out[3] = (int32_t)p.U; out[4] = p.dp_rounds;
    for (int b = 0; b <= p.G; ++b) out[5 + b] = (int32_t)g3_bound((unsigned)b, p.q, p.rem, (unsigned)nkt, g3_snap != 0);
  }
  return p.G;
}

// Called by ofx_graphconv_fwd_planes (ofx_gemm2.hip) once it has filled the common arguments.
// Returns OFX_OK (launched), a failure status, or 1 when the shape / workspace does not qualify (caller falls back
// to the one-tile-per-block kernel).
int ofx_launch_gconv3(Gemm2Args& a, int mode, int wm, int ni, void* ws_tail, size_t ws_tail_bytes, void* sync,
                      size_t sync_bytes, hipStream_t st, int nd) {
  G3Plan p;
  GemmArgs& g = a.e;
  if (!sync || ((uintptr_t)sync & 3) || !ws_tail || ((uintptr_t)ws_tail & 15)) return 1;
  if (!g.vec4) return 1;                   // (only the float4 epilogue exists in the persistent kernel)
  if (!g3_plan(g.M, (int)g.N, a.nkt, wm, ni, p)) return 1;
  if (p.part_bytes > ws_tail_bytes || (size_t)(p.G + 1) * sizeof(unsigned) > sync_bytes) return 1;
  if ((a.ldx >> 7) >= (1 << 20) || (a.ldt >> 7) >= (1 << 20) || a.n_src >= (1ll << 31)) return 1;
  Gemm3Args A = {};
  a.row0 = 0;
  g.ntm = (int)ofx_cdiv(g.M, wm * 64);
  A.b = a;
  A.G = p.G; A.q = p.q; A.rem = p.rem; A.U = p.U; A.dp_rounds = p.dp_rounds;
  A.part = (float*)ws_tail; A.flags = (unsigned*)sync;
  A.err = (unsigned*)sync + (sync_bytes / sizeof(unsigned) - 1);
  A.snap = g3_snap;
  A.xcd_contig = g3_xcd_contig;
  A.rt = p.U / (unsigned)a.nkt;
  // (shares of >= one tile: boundaries are >= nkt apart and snapping only ever moves one ONTO a tile boundary, so a
  // tile has at most one interior cut; its second piece is its block's first work, published ~a tile before the
  // finisher reaches its own)
  A.early = p.q >= (unsigned)a.nkt ? 1 : 0;
  // last 16-B chunk that still starts inside the array (a chunk may run up to 12 B past the last entry: the caller
  // guarantees 16 B of readable slack behind nbr_ext, include/ofx.h)
  A.nbr_lim = (const char*)a.nbr_ext + ((((size_t)g.M * 28 - 4) >> 4) << 4);
  if (((uintptr_t)a.nbr_ext & 15)) return 1;
  if (nd == 1) {                           // dense GEMM on the planes data path: pair modes, 128-column tiles only
    if (ni != 2 || (mode != 2 && mode != 3)) return 1;
#define G3_GO1(P_) (wm == 4 ? g3_launch<P_, 4, 2, 1>(A, st) : g3_launch<P_, 2, 2, 1>(A, st))
    return mode == 2 ? G3_GO1(2) : G3_GO1(3);
#undef G3_GO1
  }
  if (nd != 7) return 1;
#define G3_GO(P_)                                                                                         \
  (ni == 1 ? (wm == 4 ? g3_launch<P_, 4, 1>(A, st) : g3_launch<P_, 2, 1>(A, st))                          \
           : (wm == 4 ? g3_launch<P_, 4, 2>(A, st) : g3_launch<P_, 2, 2>(A, st)))
  return mode == 2 ? G3_GO(2) : (mode == 3 ? G3_GO(3) : G3_GO(1));
#undef G3_GO
}

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
#include "ofx_planes.h"

template <int PREC, int VARIANT, int WM, int NI>
__global__ void __launch_bounds__(512, 2) gconv2_kernel(const Gemm2Args a) {   // (a WM-dependent bound loses the host stub)
  typedef G2Half<PREC, NI> Half;
  typedef G2Cfg<WM, NI> CF;
  constexpr int G2_NI = NI, G2_BN = CF::BN, G2_EPI_LOADS = CF::EPI_LOADS, RH = CF::READS;
  constexpr int G2_WM = WM, G2_BM = CF::BM, G2_A_BYTES = CF::A_BYTES, G2_BUF = CF::BUF, G2_TAB = CF::TAB;
  constexpr int G2_GLDS_PER_STEP = CF::GLDS;
  extern __shared__ __attribute__((aligned(128))) char smem2[];
  const GemmArgs& g = a.e;

  float osc = *g.oscale_p;            // epilogue scale (ofx_planes.h), read here: no scalar load near the k-loop's waits
  asm volatile("" : "+s"(osc));
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
    // branch-free: all four loads are requested before the first use (column 7 re-reads column 6's entry and drops
    // it) -- one memory latency per block instead of four in series
    int32_t ids[4];
    int64_t ms[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int e = threadIdx.x + CF::THREADS * t;
      const int r = e >> 3, d = e & 7;
      int64_t m = m0 + r;
      m = m < mmax ? m : mmax;
      ms[t] = m;
      ids[t] = a.nbr_ext[m * 7 + (d < 7 ? d : 6)];
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int e = threadIdx.x + CF::THREADS * t;
      const int d = e & 7;
      const int64_t id = ids[t];
      const int64_t gl = id < a.n_src ? x_line + id * lpr : aux_line + (id - a.n_src) * lpr;
      tab[e] = (uint32_t)(d < 7 ? gl : ms[t] * lpt);
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
      const int t = g2_three_term<PREC>() ? (u == 0 ? c : 2 + c) : 2 * c + u;
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
    if (g2_three_term<PREC>()) {
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
    if constexpr (g2_three_term<PREC>()) {
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
  constexpr int NMF = (g2_three_term<PREC>() ? 3 : 2) * G2_MI * G2_NI;           // MFMAs per half step
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
  constexpr int MFMAS = (g2_three_term<PREC>() ? 3 : 2) * G2_MI * G2_NI;
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

  if (vec4) g2_epilogue_finish<G2_WM, G2_WN, G2_MI, G2_NI>(g, acc, P, m0, n0, wm, wn, l31, h, osc);
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
// fp32 [n, C] (zero-extended to Cpad columns) -> planes.  mode 2: lane q of every 8-lane group owns channels
// 4q..4q+3 of a 32-channel chunk (one coalesced 128-B line in, 64 B hi + 64 B lo out, safe in place);
// mode 1: fp16 row-major (8 B per lane).
__global__ void __launch_bounds__(256) planes_split_kernel(const float* __restrict__ x, int64_t ldx, int64_t n, int C,
                                                           int Cpad, int mode, char* __restrict__ out, int64_t ldo,
                                                           uint32_t* __restrict__ range_words) {
  const int CT = Cpad >> 2;
  const int64_t total = n * CT;
  unsigned sat = 0;             // operands beyond the fp16 range (modes 1 / 3: they become Inf / NaN): include/ofx.h, range guard
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
    if (mode != 2)        // (!(a <= b) also catches NaN)
      sat += !(fabsf(v.x) <= 65504.f) + !(fabsf(v.y) <= 65504.f) + !(fabsf(v.z) <= 65504.f) + !(fabsf(v.w) <= 65504.f);
    if (g2_pairs(mode)) {
      unsigned h0, h1, l0, l1;
      g2_split2(mode, v.x, v.y, h0, l0);
      g2_split2(mode, v.z, v.w, h1, l1);
      char* o = out + r * ldo + (c >> 5) * 128 + (c & 31) * 2;
      *reinterpret_cast<uint2*>(o) = make_uint2(h0, h1);
      *reinterpret_cast<uint2*>(o + 64) = make_uint2(l0, l1);
    } else {
      *reinterpret_cast<uint2*>(out + r * ldo + c * 2) = make_uint2(g2_pk_f16(v.x, v.y), g2_pk_f16(v.z, v.w));
    }
  }
  if (sat && range_words) atomicAdd(range_words, sat);
}

extern "C" int ofx_planes_split(const float* x, int64_t ldx, int64_t n, int C, int Cpad, int mode, void* out,
                                int64_t ldo_bytes, void* stream) {
  const int chunk = g2_pairs(mode) ? 32 : 64;
  if (mode < 1 || mode > 3 || n < 0 || C < 1 || Cpad < C || (Cpad % chunk) || ldx < C || !out ||
      ldo_bytes < (int64_t)Cpad * (g2_pairs(mode) ? 4 : 2) || (ldo_bytes & 15) || ((uintptr_t)out & 15) || (n > 0 && !x))
    return OFX_EINVAL;
  if (C >= 4 && ((ldx & 3) || ((uintptr_t)x & 15))) return OFX_EINVAL;
  if (n > 0)
    planes_split_kernel<<<ofx_grid(n * (Cpad / 4), 256), 256, 0, ofx_stream(stream)>>>(x, ldx, n, C, Cpad, mode,
                                                                                      (char*)out, ldo_bytes,
                                                                                      ofx_range_words());
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
    if (g2_pairs(mode)) {
      const unsigned short* q = reinterpret_cast<const unsigned short*>(p + r * ldp + (c >> 5) * 128 + (c & 31) * 2);
      float vb;
      g2_join2(mode, q[0], q[32], v, vb);
    } else {
      v = (float)*reinterpret_cast<const _Float16*>(p + r * ldp + c * 2);
    }
    out[r * ldo + c] = v;
  }
}
extern "C" int ofx_planes_merge(const void* planes, int64_t ldp_bytes, int64_t n, int C, int mode, float* out,
                                int64_t ldo, void* stream) {
  const int chunk = g2_pairs(mode) ? 32 : 64;
  if (mode < 1 || mode > 3 || n < 0 || C < 1 || (C % chunk) || !planes || !out || ldo < C) return OFX_EINVAL;
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
    const int64_t poff = g2_pairs(mode) ? (int64_t)(c >> 5) * 128 + (c & 31) * 2 : (int64_t)c * 2;
    if (v > 0) {
      const int64_t s = multi_seg[v - 1];
      const int32_t b = seg_ptr[s], e = seg_ptr[s + 1];
      for (int32_t p = b; p < e; ++p) {
        const char* row = xp + (int64_t)col[p] * ldx + poff;
        const u32x4 hi = *reinterpret_cast<const u32x4*>(row);
        if (g2_pairs(mode)) {
          const u32x4 lo = *reinterpret_cast<const u32x4*>(row + 64);
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            float va, vb;
            g2_join2(mode, hi[k], lo[k], va, vb);
            acc[2 * k] += va;
            acc[2 * k + 1] += vb;
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
    if (g2_pairs(mode)) {
      u32x4 hi, lo;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        unsigned hh, ll;
        g2_split2(mode, acc[2 * k], acc[2 * k + 1], hh, ll);
        hi[k] = hh; lo[k] = ll;
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
static inline int64_t g2_chunk(int mode) { return g2_pairs(mode) ? 32 : 64; }
extern "C" int64_t ofx_planes_packed_ktiles(int cin, int nt, int mode) {
  const int64_t ch = g2_chunk(mode);
  return 7 * ((int64_t)cin / ch) + (nt > 1 ? (7 * (int64_t)nt + ch - 1) / ch : 0);
}
// + one 128-B trailer line: float [0] = 1 / s, [1] = s, the per-tensor power-of-two scale the 16-bit halves were
// written with (include/ofx.h, range guard); the kernels' epilogue multiplies the accumulators by trailer[0]
constexpr int64_t G2_TRAILER = 128;
extern "C" int64_t ofx_planes_packed_bytes(int cin, int nt, int cout, int mode) {
  return ofx_planes_packed_ktiles(cin, nt, mode) * (int64_t)cout * G2_LINE + G2_TRAILER;
}

__global__ void __launch_bounds__(256) planes_pack_kernel(const float* __restrict__ W, int64_t sk, int64_t sn, int cin,
                                                          int ntc, int64_t N, int64_t nkt, int mode,
                                                          char* __restrict__ out, int ndir = 7) {
  // one thread per 16-B piece: (k tile, column, piece)
  const int64_t total = nkt * N * 8;
  const float wscale = reinterpret_cast<const float*>(out + nkt * N * G2_LINE)[1];     // written by ofx_launch_weight_scale
  const int ch = g2_pairs(mode) ? 32 : 64;
  const int64_t Kf = ndir * (int64_t)cin;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int p = (int)(t & 7);
    const int64_t n = (t >> 3) % N, kt = (t >> 3) / N;
    const int64_t k0 = kt * ch + (g2_pairs(mode) ? (p & 3) * 8 : p * 8);
    float w[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int64_t k = k0 + e;
      int64_t src = -1;
      if (k < Kf) {
        const int64_t dir = k / cin, c = k - dir * cin;
        src = dir * (cin + ntc) + c;
      } else if (ntc > 0 && k - Kf < ndir * (int64_t)ntc) {
        const int64_t kk = k - Kf, dir = kk / ntc, ty = kk - dir * ntc;
        src = dir * (cin + ntc) + cin + ty;
      }
      w[e] = src >= 0 ? W[src * sk + n * sn] * wscale : 0.f;
    }
    u32x4 o;
    if (g2_pairs(mode)) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        unsigned hi, lo;
        g2_split2(mode, w[2 * e], w[2 * e + 1], hi, lo);
        o[e] = p < 4 ? hi : lo;
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
  if (!W || !out || mode < 1 || mode > 3 || cin < 1 || (cin % g2_chunk(mode)) || nt < 0 || cout < 1 ||
      ((uintptr_t)out & 15))
    return OFX_EINVAL;
  const int64_t nkt = ofx_planes_packed_ktiles(cin, nt, mode);
  const int ntc = nt > 1 ? nt : 0;
  if (ofx_launch_weight_scale(W, sk, sn, 7 * (int64_t)(cin + ntc), cout, mode != 2 ? 1 : 0,
                              reinterpret_cast<float*>((char*)out + nkt * cout * G2_LINE), ofx_stream(stream)))
    return OFX_ELAUNCH;
  planes_pack_kernel<<<ofx_grid(nkt * cout * 8, 256), 256, 0, ofx_stream(stream)>>>(W, sk, sn, cin, ntc,
                                                                                    cout, nkt, mode, (char*)out);
  OFX_LAUNCH_CHECK();
  return OFX_OK;
}

// Dense weights [K, N] (element (k, n) at W[k * sk + n * sn]) as [k tile][N][128-B line] for ofx_gemm_planes: the
// GraphConv pack with ONE direction and no node-type rows; (K / 32) * N * 128 + 128 bytes.
extern "C" int64_t ofx_gemm_planes_packed_bytes(int K, int N, int mode) {
  if (!g2_pairs(mode) || K < 32 || (K % 32) || N < 1) return 0;
  return (int64_t)(K / 32) * N * G2_LINE + G2_TRAILER;
}
extern "C" int ofx_pack_gemm_planes(const float* W, int64_t sk, int64_t sn, int K, int N, int mode, void* out,
                                    void* stream) {
  if (!W || !out || !g2_pairs(mode) || K < 32 || (K % 32) || N < 1 || ((uintptr_t)out & 15)) return OFX_EINVAL;
  const int64_t nkt = K / 32;
  if (ofx_launch_weight_scale(W, sk, sn, K, N, mode != 2 ? 1 : 0, reinterpret_cast<float*>((char*)out + nkt * N * G2_LINE),
                              ofx_stream(stream)))
    return OFX_ELAUNCH;
  planes_pack_kernel<<<ofx_grid(nkt * N * 8, 256), 256, 0, ofx_stream(stream)>>>(W, sk, sn, K, 0, N, nkt, mode, (char*)out, 1);
  OFX_LAUNCH_CHECK();
  return OFX_OK;
}

// ------------------------------------------------------------------------------------------------
int ofx_launch_stats_reduce(const GemmArgs& g, int wr_rows, hipStream_t st);   // ofx_gemm.hip

// Schedule variants other than 5 (earlier schedules 0 / 1, ablations 2-4 / 6 with WRONG results) and the per-block
// clock-stamp buffer exist only in -DOFX_ABLATION builds (python -m octfusion_amd.build --ablation); the product
// library accepts variant 5 and a NULL debug buffer only.
static int g2_variant = 5;
extern "C" int ofx_set_gconv2_variant(int v) {
#ifdef OFX_ABLATION
  if (v < 0 || v > 6) return OFX_EINVAL;
#else
  if (v != 5) return OFX_EINVAL;
#endif
  g2_variant = v;
  return OFX_OK;
}
static unsigned long long* g2_debug = nullptr;
extern "C" int ofx_set_gconv2_debug(void* buf) {
#ifndef OFX_ABLATION
  if (buf) return OFX_EINVAL;
#endif
  g2_debug = (unsigned long long*)buf;
  return OFX_OK;
}

template <int PREC, int VARIANT, int WM, int NI>
static int g2_launch(const Gemm2Args& a, hipStream_t st) {
  static bool attr_set[OFX_MAX_DEVICES] = {};
  if (!ofx_raise_lds_limit(reinterpret_cast<const void*>(&gconv2_kernel<PREC, VARIANT, WM, NI>), G2Cfg<WM, NI>::LDS, attr_set))
    return OFX_ELAUNCH;
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

static int g3_wm_pref = 0;                  // persistent kernel geometry when the tile knob is 0: 0 = rule below
static int g3_auto_wm(int64_t M, int cout, int nkt, int ni) {
  (void)cout; (void)ni;
  if (g3_wm_pref == 2 || g3_wm_pref == 4) return g3_wm_pref;
  // 256-row tiles (one block per CU) won on every layer of the hr workload, 3-14 % over the 128-row geometry
  // (profiles/r03/gconv3_ab_shell6_b8.json): fewer operand bytes per MFMA, and the persistent loop already hides
  // what the second co-resident block used to hide.  Layers with few k-step units are latency-bound instead (a block's
  // share is the 16-step minimum whatever the geometry): half-size k-steps halve that latency.  One-shape step trace,
  // round 4 (tools/step_trace.py --batch 1 --tile 0 / 2): 960 units 46.8 -> 37.4 us, 1 920: 47.0 -> 39.2, 1 856: 52.4 ->
  // 42.8, 3 030: 49.8 -> 48.5, 3 712: 54.0 -> 55.3, 7 296: 70.8 -> 72.4, 8 686: 77.3 -> 79.1.
  const int64_t U4 = ofx_cdiv(M, 256) * ofx_cdiv(cout, 64 * ni) * nkt;
  return U4 >= 3500 ? 4 : 2;
}
// persistent stream-K kernel (ofx_gemm3.hip); returns 1 when the shape / workspace does not qualify
int ofx_launch_gconv3(Gemm2Args& a, int mode, int wm, int ni, void* ws_tail, size_t ws_tail_bytes, void* sync,
                      size_t sync_bytes, hipStream_t st, int nd = 7);
static int g2_persistent = 1;               // 1 (default): persistent stream-K blocks where the shape qualifies
void ofx_gconv3_set_hybrid(int on);          // ofx_gemm3.hip
void ofx_gconv3_set_snap(int near);          // ofx_gemm3.hip
extern "C" int ofx_set_gconv_persistent(int on) {
  if (on < 0 || on > 3) return OFX_EINVAL;   // 2: pure stream-K (no whole-tile rounds), 3: round-1 share snapping -- A/B
  g2_persistent = on ? 1 : 0;
  ofx_gconv3_set_hybrid(on == 2 ? 0 : 1);
  ofx_gconv3_set_snap(on == 3 ? 0 : 1);
  return OFX_OK;
}

extern "C" int ofx_graphconv_fwd_planes(const void* xp, int64_t ldx_bytes, int cin, int64_t n_nodes,
                                        const int32_t* seg_ptr, const int32_t* col, const int32_t* nbr_ext,
                                        const int32_t* multi_seg, int64_t n_multi, void* aux, size_t aux_bytes,
                                        const void* tfp, int64_t ldt_bytes, int nt, const void* W2, int cout,
                                        const float* bias, const float* emb, int64_t lde, const int32_t* batch_id,
                                        const float* res, int64_t ldr, float* out, int64_t ldc, double* stats,
                                        int64_t stats_ld, void* ws, size_t ws_bytes, void* sync, size_t sync_bytes,
                                        int mode, int aux_ready, void* stream) {
  if (mode < 1 || mode > 3) return OFX_EINVAL;
  const int64_t ch = g2_chunk(mode);
  if (n_nodes == 0 && cin >= 1 && cout >= 1) return OFX_OK;
  if (n_nodes < 0 || cin < 1 || (cin % ch) || cout < 1 || !xp || !seg_ptr || !col || !nbr_ext || !aux || !W2 || !out ||
      ldx_bytes < (int64_t)cin * (g2_pairs(mode) ? 4 : 2) || (ldx_bytes & 127) || ((uintptr_t)xp & 127) ||
      ((uintptr_t)aux & 127) || ((uintptr_t)W2 & 127) || ldc < cout || (res && ldr < cout) ||
      (emb && (!batch_id || lde < cout)) || n_multi < 0 || (n_multi > 0 && !multi_seg) || nt < 0)
    return OFX_EINVAL;
  if (aux_bytes < (size_t)(n_multi + 1) * (size_t)ldx_bytes) return OFX_EINVAL;      // aux rows are written / gathered at the pitch of xp
  const int ntc = nt > 1 ? nt : 0;
  if (ntc > 0 && (!tfp || (ldt_bytes & 127) || ((uintptr_t)tfp & 127))) return OFX_EINVAL;
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
  g.oscale_p = reinterpret_cast<const float*>((const char*)W2 + (int64_t)a.nkt * cout * G2_LINE);    // the pack's trailer
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
#ifdef OFX_ABLATION
    if (mode == 2) {
      switch (g2_variant) {
        case 0: return G2_GO(2, 0, wm);
        case 2: return G2_GO(2, 2, wm);      // ablations (wrong results): no MFMA / no DMA / no LDS reads
        case 3: return G2_GO(2, 3, wm);
        case 4: return G2_GO(2, 4, wm);
        case 6: return G2_GO(2, 6, wm);      // ablation: spliced schedule without the per-step s_barrier (races)
        case 1: return G2_GO(2, 1, wm);
        default: return G2_GO(2, 5, wm);
      }
    }
    if (mode == 3) return G2_GO(3, 5, wm);
    return g2_variant == 0 ? G2_GO(1, 0, wm) : (g2_variant == 1 ? G2_GO(1, 1, wm) : G2_GO(1, 5, wm));
#else
    return mode == 2 ? G2_GO(2, 5, wm) : (mode == 3 ? G2_GO(3, 5, wm) : G2_GO(1, 5, wm));   // (the spliced schedule is the only one in product builds)
#endif
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
  rc = 1;
  if (g2_persistent && sync) {
    // persistent stream-K launch: no tile quantisation, so the geometry follows the operand traffic alone -- 256-row
    // tiles (fewer LDS bytes per MFMA) unless the layer is too small to give every CU's single block enough units
    size_t used = 0;
    if (g.stats_part) used = ((size_t)ofx_cdiv(g.M, 64) * (size_t)g.N * 2 * sizeof(float) + 255) & ~(size_t)255;
    const int wm3 = (g2_wm == 2 || g2_wm == 4) ? g2_wm : g3_auto_wm(g.M, cout, a.nkt, ni);
    if (ws && used <= ws_bytes)
      rc = ofx_launch_gconv3(a, mode, wm3, ni, (char*)ws + used, ws_bytes - used, sync, sync_bytes, st);
    if (rc < 0) return rc;
  }
  if (rc == 1) rc = launch((g2_wm == 2 || g2_wm == 4) ? g2_wm : wm_auto, 0, g.M);
#undef G2_GO
  if (rc) return rc;
  if (g.stats_part) {
    rc = ofx_launch_stats_reduce(g, 64, st);
    if (rc) return rc;
  }
  OFX_LAUNCH_CHECK();
  return OFX_OK;
}

// ------------------------------------------------------------------------------------------------
// out[m, :] = A[row_tab[m, 0], :] @ W (+ bias) on the planes data path (persistent stream-K blocks, csrc/ofx_gemm3.hip
// with ND = 1): the reference's Upsample GEMM (modules.py:430-446, x[n, C] @ W.flatten(1) -> [n, 8 C]) for the non-leaf
// rows of a graph depth.  ap: operand planes of A (pair modes 2 / 3), row_tab: int32 [M, 7] whose FIRST column is the
// source row of output row m in [0, n_a) (the other six are ignored by the kernel but must be readable: pass any valid
// row; + 16 B of slack behind the table, 16-B aligned), W2 from ofx_pack_gemm_planes, out fp32 or -- out_mode 2 / 3 --
// pair planes.  Returns OFX_OK, a failure status, or 1 when the shape does not qualify (K / 32 < 8 k-steps, too few
// tiles, workspace too small): the caller then uses ofx_gemm_f32(_planes).
extern "C" int ofx_gemm_planes(const void* ap, int64_t lda_bytes, int64_t n_a, int64_t M, int K, const int32_t* row_tab,
                               const void* W2, int N, const float* bias, float* out, int64_t ldc, int out_mode, void* ws,
                               size_t ws_bytes, void* sync, size_t sync_bytes, int mode, void* stream) {
  if (!g2_pairs(mode) || !ap || !row_tab || !W2 || !out || M < 0 || n_a < 1 || K < 32 || (K % 32) || N < 4 || (N & 3) ||
      lda_bytes < (int64_t)K * 4 || (lda_bytes & 127) || ((uintptr_t)ap & 127) || ((uintptr_t)W2 & 127) || ldc < N ||
      (ldc & 3) || ((uintptr_t)out & 15) || (bias && ((uintptr_t)bias & 15)) ||
      (out_mode && (!g2_pairs(out_mode) || (ldc & 31) || (N & 31) || ((uintptr_t)out & 127))))
    return OFX_EINVAL;
  if (M == 0) return OFX_OK;
  if (!sync || !ws || N < 128) return 1;
  Gemm2Args a = {};
  a.xp = (const char*)ap; a.ldx = lda_bytes; a.aux = (const char*)ap; a.n_src = n_a; a.nbr_ext = row_tab;
  a.tfp = (const char*)ap; a.ldt = lda_bytes;
  a.W2 = (const char*)W2;
  a.tpd = K / 32; a.nkt_g = a.tpd; a.nkt = a.tpd;
  GemmArgs& g = a.e;
  g.M = M; g.N = N; g.K = g.Kp = K; g.bias = bias; g.out = out; g.ldc = ldc; g.nsplit = 1; g.out_planes = out_mode;
  g.oscale_p = reinterpret_cast<const float*>((const char*)W2 + (int64_t)a.nkt * N * G2_LINE);
  g.vec4 = 1;
  g.ntn = (int)ofx_cdiv(g.N, 128);
  const int wm = g3_auto_wm(M, N, a.nkt, 2);
  const int rc = ofx_launch_gconv3(a, mode, wm, 2, ws, ws_bytes, sync, sync_bytes, ofx_stream(stream), 1);
  if (rc) return rc;
  OFX_LAUNCH_CHECK();
  return OFX_OK;
}

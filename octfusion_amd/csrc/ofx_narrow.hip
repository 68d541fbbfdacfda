// GraphConv (reference models/networks/modules.py:194-220) for the two layers of a diffusion U-Net that have almost no
// arithmetic: the INPUT convolution (3 or 8 channels -> model_channels, graph_unet_hr.py:116) and the OUTPUT
// convolution (model_channels -> 3 or 8, graph_unet_hr.py:205-209).  Both were run through the contraction kernels
// until round 4 (input: zero-padded to a 32-channel chunk on the planes kernel, 112 us at depth 6 = 0.15 of the HBM
// roof; output: the register-staged kernel gathering E x 128 floats for 3 output columns, 137 us).  They are gathers:
//
//   * narrow_in_kernel: per 64-row block the 7 x (cin + nt) col_data values of every row go to LDS (one thread per
//     (row, direction) segment walks its CSR edges: 12..32-byte pieces of x, L2 hits), then the [64, K] x [K, cout]
//     product runs on the exact-fp32 MFMA with a wave's 32-column slice of W held in registers for the whole block.
//     GroupNorm statistics of the output ride along (per-lane column sums) in the two-stage protocol of the MFMA
//     kernels (stats_part[block][cout][2] -> stats_reduce_kernel).  Bound: the N x cout x 4-byte store stream.
//   * narrow_out_kernel: project-then-aggregate.  scatter_mean and the weight product commute, so the caller first
//     projects every node ONCE, P[j, dir * cout + o] = y[j, :] . W[dir, :, o]  (a dense [N, C] x [C, 7 cout] GEMM that reads y
//     coalesced), and this kernel gathers cout floats per EDGE instead of C: lane = (row, direction), the seven
//     partial means of a row and its node-type term meet by three xor-shuffles.  A row of P is padded to one or two
//     128-B lines so a gathered piece never straddles lines.  (Round 4 rejected a first version of this idea whose
//     aggregation walked all seven segments of a row in ONE thread -- a chain of ~20 dependent loads per thread; here
//     the chain is three loads long and 27 k waves hide it.)
#include "ofx_gemm_common.h"

int ofx_launch_stats_reduce(const GemmArgs& g, int wr_rows, hipStream_t st);   // ofx_gemm.hip

namespace {

struct NarrowInArgs {
  const float* x; int64_t ldx; int cin; int64_t N;
  const int32_t* seg_ptr; const int32_t* col;
  const uint8_t* ntype; int nt;
  const float* W; int cout; const float* bias;
  float* out; int64_t ldc;
  const int32_t* bid; float* stats_part; double* stats; int64_t stats_ld;
};

// Loads whose bound is a run-time value are UNCONDITIONAL with a clamped index and a select on the value: a load under
// `if (k < K)` costs a branch + an s_waitcnt per load (the first version of this kernel spent 64 x an L2 round trip on
// fetching its weights: 182 us per depth-6 launch against 112 us for the path it replaced).  The contraction is 8 (4)
// tiles of 32 x 32 on the exact-fp32 MFMA (v_mfma_f32_32x32x2_f32, K / 2 instructions per tile): the second version ran it
// as per-lane FMA chains fed by broadcast LDS reads and was bound by those reads (one lgkmcnt(0) per read: 157 us).
//   wave -> a 32-column slice of W (K / 2 registers: B[k = 2 kk + lane / 32][n = lane % 32]) and one or two 32-row tiles;
//   A[m = lane % 32][k = 2 kk + lane / 32] comes from LDS with an odd row pitch (conflict-free);
//   C / D: col = lane % 32, row = (reg & 3) + 8 (reg >> 2) + 4 (lane / 32): a register is two 128-B row pieces.
// 512 threads: ONE (row, direction) segment per thread -- the gather is a chain of three dependent loads (segment bounds ->
// column -> x / node type, the first two first-touch HBM misses), and walking two segments per thread in turn made the
// block's latency two chains long (97 us per depth-6 launch with 256 threads).
template <int KP, int CI>
__global__ void __launch_bounds__(512, KP == 64 ? 6 : 4) narrow_in_kernel(const NarrowInArgs a) {
  constexpr int LD = KP + 1;
  __shared__ float cd[64 * LD];
  __shared__ float red[8][32][2];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int64_t row0 = (int64_t)blockIdx.x * 64;
  const int cin = a.cin, nt = a.nt, cpd = cin + nt, K = 7 * cpd;
  const bool wide = a.cout == 128;          // 4 column slices x 2 row tiles = one tile per wave; else 2 x 2 on waves 0..3
  const int slice = wide ? (wv & 3) : (wv & 1);
  const int mt0 = wide ? (wv >> 2) : ((wv >> 1) & 1), nmt = (wide || wv < 4) ? 1 : 0;
  const int ln = lane & 31, lk = lane >> 5;

  // ---- phase 1: col_data of the block's rows -> LDS (zero beyond K and beyond N).  One thread per (row, direction)
  // segment: mean of the cin channels and of the one-hot node types (modules.py:199-207) over its neighbours.
  {                                                     // columns K .. KP - 1
    const int r = tid >> 3;
    for (int k = K + (tid & 7); k < KP; k += 8) cd[r * LD + k] = 0.f;
  }
  for (int s = tid; s < 64 * 7; s += 512) {
    const int r = s / 7, dir = s - r * 7;
    const int64_t row = row0 + r;
    float acc[CI], tc[8];
#pragma unroll
    for (int c = 0; c < CI; ++c) acc[c] = 0.f;
#pragma unroll
    for (int t = 0; t < 8; ++t) tc[t] = 0.f;
    int32_t b = 0, e = 0;
    if (row < a.N) { b = a.seg_ptr[row * 7 + dir]; e = a.seg_ptr[row * 7 + dir + 1]; }
    for (int32_t p = b; p < e; ++p) {
      const int64_t j = a.col[p];
      const float* xr = a.x + j * a.ldx;
      float v[CI];
#pragma unroll
      for (int c = 0; c < CI; ++c) v[c] = xr[c < cin ? c : cin - 1];
      const int ty = nt ? (int)a.ntype[j] : 0;
#pragma unroll
      for (int c = 0; c < CI; ++c) acc[c] += v[c];
#pragma unroll
      for (int t = 0; t < 8; ++t) tc[t] += ty == t ? 1.f : 0.f;
    }
    const float inv = __frcp_rn((float)(e - b > 1 ? e - b : 1));
    float* o = &cd[r * LD + dir * cpd];
#pragma unroll
    for (int c = 0; c < CI; ++c)
      if (c < cin) o[c] = acc[c] * inv;
#pragma unroll
    for (int t = 0; t < 8; ++t)
      if (t < nt) o[cin + t] = tc[t] * inv;
  }
  // this wave's slice of W as the MFMA's B operand: requested here, after the gather's registers are dead (six waves per
  // SIMD need <= 80 VGPRs) and before the barrier, so the loads fly while the block's slower threads finish
  float bw[KP / 2];
#pragma unroll
  for (int kk = 0; kk < KP / 2; ++kk) {
    const int k = 2 * kk + lk;
    const float v = a.W[(int64_t)(k < K ? k : K - 1) * a.cout + slice * 32 + ln];
    bw[kk] = k < K ? v : 0.f;
  }
  const float bias = a.bias ? a.bias[slice * 32 + ln] : 0.f;
  // do all rows of the block belong to one batch element?  (decides how the statistics leave the block)
  bool same = true;
  if (a.stats && tid < 64 && row0 + tid < a.N) same = a.bid[row0 + tid] == a.bid[row0];
  const bool uni = __syncthreads_and(same);

  // ---- phase 2: the tiles of this wave
  float s_ = 0.f, q_ = 0.f;
  for (int t = 0; t < nmt; ++t) {
    const int mt = mt0 + t;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const float* ap = &cd[(mt * 32 + ln) * LD + lk];
#pragma unroll
    for (int kk = 0; kk < KP / 2; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * kk], bw[kk], acc, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int64_t row = row0 + mt * 32 + (i & 3) + 8 * (i >> 2) + 4 * lk;
      const float v = acc[i] + bias;
      if (row < a.N) {
        a.out[row * a.ldc + slice * 32 + ln] = v;
        if (a.stats) {
          if (uni) { s_ += v; q_ += v * v; }
          else {                                         // a block that holds a batch boundary (a handful per launch)
            double* so = a.stats + ((int64_t)a.bid[row] * a.stats_ld + slice * 32 + ln) * 2;
            unsafeAtomicAdd(so, (double)v);
            unsafeAtomicAdd(so + 1, (double)v * (double)v);
          }
        }
      }
    }
  }
  if (a.stats) {
    // one partial (sum, sum of squares) per block and column: stats_part[block][cout][2] (zeros for a mixed block)
    s_ += __shfl_xor(s_, 32);
    q_ += __shfl_xor(q_, 32);
    if (lk == 0) { red[wv][ln][0] = s_; red[wv][ln][1] = q_; }
    __syncthreads();
    if (tid < a.cout) {                                  // the two row tiles of a slice sit on waves sl and sl + 4 (sl + 2)
      const int sl = tid >> 5, j = tid & 31, o = wide ? 4 : 2;
      *reinterpret_cast<float2*>(a.stats_part + ((int64_t)blockIdx.x * a.cout + tid) * 2) =
          make_float2(red[sl][j][0] + red[sl + o][j][0], red[sl][j][1] + red[sl + o][j][1]);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Round 6: the same operator driven by the branch-free gather table (nbr_ext, include/ofx.h) in a PERSISTENT, software-
// pipelined block.  narrow_in_kernel's block is one chain of three dependent loads (segment bounds -> column -> x) followed
// by its contraction and stores: 14 us per 64-row block at depth 8 with three blocks per CU -- latency, not bytes (954 us for
// an 832 MB output; 0.22 of the HBM roof).  Here
//   * a segment's source is ONE table entry (coalesced, 28 B per row) indexing ONE 32-B record (x + node type of a row, or
//     the pre-averaged x mean + node-type counts of a multi-neighbour segment; narrow_rec_kernel writes them per call:
//     32 B per id) -- a single memory sector per segment, no branch, no CSR walk;
//   * a block walks row groups g, g + grid, ...: while the MFMAs and stores of group k run, the x pieces of group k + 1
//     are in flight (their ids arrived during group k - 1) and the ids of group k + 2 are requested -- the chain is paid
//     once per block, not once per 64 rows;
//   * the wave's 32-column slice of W stays in registers for the whole block.
struct NarrowIn2Args {
  const float* x; int64_t ldx; int cin; int64_t N;
  const int32_t* nbr_ext;                    // [N, 7]: < N row, N zero row, N + 1 + v aux record v + 1
  const float* rec;                          // [N + 1 + V] records (narrow_rec_kernel): the id of the table IS the index
  int nt;
  const float* W; int cout; const float* bias;
  float* out; int64_t ldc;
  const int32_t* bid; float* stats_part; double* stats; int64_t stats_ld;
  int64_t ngroups;
};

// One record per gatherable id, so that a segment's contribution is ONE aligned 32-B (cin <= 4) or 64-B (cin <= 8) piece
// -- one memory sector per segment instead of a 12-B piece of x plus a node-type byte from another sector:
//   [x (CI floats; the segment MEAN for an aux record) | node-type counts 0-3, 4-7 (2 x u32, one byte each) | 1 / count | pad]
// ids < N: the node itself (count 1, one-hot type); N: zeros; N + v: the mean over multi-neighbour segment multi_seg[v - 1].
// Two launches: the node records (a streaming copy: x row + node type -> one 32-B record; measured at 1 TB/s while it shared
// a kernel, and its registers, with the CSR walk below) and the aux records (one thread per multi-neighbour segment).
template <int CI>
__global__ void __launch_bounds__(256) narrow_rec_nodes_kernel(const float* __restrict__ x, int64_t ldx, int cin, int64_t N,
                                                               const uint8_t* __restrict__ ntype, int nt,
                                                               float* __restrict__ rec) {
  constexpr int RW = CI == 4 ? 8 : 16;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i <= N; i += (int64_t)gridDim.x * 256) {
    float v[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) v[c] = 0.f;
    unsigned t0 = 0, t1 = 0;
    float inv = 0.f;
    if (i < N) {                                          // (record N = the zero row)
      const float* xr = x + i * ldx;
#pragma unroll
      for (int c = 0; c < CI; ++c) v[c] = c < cin ? xr[c < cin ? c : 0] : 0.f;
      const int ty = nt ? (int)ntype[i] : 0;
      t0 = ty < 4 ? 1u << (8 * ty) : 0u;
      t1 = ty < 4 ? 0u : 1u << (8 * (ty - 4));
      inv = 1.f;
    }
    float4* o = reinterpret_cast<float4*>(rec + i * RW);
    o[0] = make_float4(v[0], v[1], v[2], v[3]);
    if (CI == 4) {
      o[1] = make_float4(__uint_as_float(t0), __uint_as_float(t1), inv, 0.f);
    } else {
      o[1] = make_float4(v[4], v[5], v[6], v[7]);
      o[2] = make_float4(__uint_as_float(t0), __uint_as_float(t1), inv, 0.f);
    }
  }
}

template <int CI>
__global__ void __launch_bounds__(256) narrow_rec_aux_kernel(const float* __restrict__ x, int64_t ldx, int cin, int64_t N,
                                                             const int32_t* __restrict__ seg_ptr, const int32_t* __restrict__ col,
                                                             const int32_t* __restrict__ multi_seg, int64_t V,
                                                             const uint8_t* __restrict__ ntype, int nt, float* __restrict__ rec) {
  constexpr int RW = CI == 4 ? 8 : 16;
  const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;          // aux record v + 1 = id N + 1 + v
  if (v >= V) return;
  float acc[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) acc[c] = 0.f;
  unsigned t0 = 0, t1 = 0;
  const int64_t sgm = multi_seg[v];
  const int32_t b = seg_ptr[sgm], e = seg_ptr[sgm + 1];
  // groups of four neighbours with all column ids, then all x pieces and node types in flight together (a serial walk is
  // a chain of two dependent loads per neighbour; segments hold 4 ... 16 of them)
  for (int32_t p0 = b; p0 < e; p0 += 4) {
    int64_t j[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) j[k] = col[p0 + k < e ? p0 + k : b];
    float xv[4][CI];
    int ty[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float* xr = x + j[k] * ldx;
#pragma unroll
      for (int c = 0; c < CI; ++c) xv[k][c] = xr[c < cin ? c : cin - 1];
      ty[k] = nt ? (int)ntype[j[k]] : 0;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (p0 + k < e) {
#pragma unroll
        for (int c = 0; c < CI; ++c) acc[c] += xv[k][c];
        if (ty[k] < 4) t0 += 1u << (8 * ty[k]); else t1 += 1u << (8 * (ty[k] - 4));
      }
    }
  }
  const float inv = __frcp_rn((float)(e - b > 1 ? e - b : 1));
#pragma unroll
  for (int c = 0; c < CI; ++c) acc[c] = c < cin ? acc[c] * inv : 0.f;
  float4* o = reinterpret_cast<float4*>(rec + (N + 1 + v) * RW);
  o[0] = make_float4(acc[0], acc[1], acc[2], acc[3]);
  if (CI == 4) {
    o[1] = make_float4(__uint_as_float(t0), __uint_as_float(t1), inv, 0.f);
  } else {
    o[1] = make_float4(acc[4], acc[5], acc[6], acc[7]);
    o[2] = make_float4(__uint_as_float(t0), __uint_as_float(t1), inv, 0.f);
  }
}

template <int KP, int CI>
__global__ void __launch_bounds__(512, KP <= 72 ? 4 : 2) narrow_in2_kernel(const NarrowIn2Args a) {
  constexpr int LD = KP + 1;
  __shared__ float cd[2][64 * LD];
  __shared__ float red[8][32][2];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int cin = a.cin, nt = a.nt, cpd = cin + nt, K = 7 * cpd;
  const bool wide = a.cout == 128;
  const int slice = wide ? (wv & 3) : (wv & 1);
  const int mt0 = wide ? (wv >> 2) : ((wv >> 1) & 1), nmt = (wide || wv < 4) ? 1 : 0;
  const int ln = lane & 31, lk = lane >> 5;
  const int64_t N = a.N;
  // this thread's segment of every group: (row r of the group, direction); threads 448..511 have none
  const bool seg = tid < 64 * 7;
  const int r = tid / 7, dir = tid - r * 7;
  // columns K .. KP - 1 of both stages are zero for the whole launch
  for (int i = tid; i < 2 * 64 * (KP - K); i += 512) {
    const int st = i / (64 * (KP - K)), q = i - st * (64 * (KP - K));
    cd[st][(q / (KP - K)) * LD + K + q % (KP - K)] = 0.f;
  }
  // the wave's slice of W as the MFMA's B operand, for every group of this block
  float bw[KP / 2];
#pragma unroll
  for (int kk = 0; kk < KP / 2; ++kk) {
    const int k = 2 * kk + lk;
    const float v = a.W[(int64_t)(k < K ? k : K - 1) * a.cout + slice * 32 + ln];
    bw[kk] = k < K ? v : 0.f;
  }
  const float bias = a.bias ? a.bias[slice * 32 + ln] : 0.f;

  const int Ni = (int)N;                                  // (ids are int32: N + 1 + V < 2^31, host-checked)
  auto load_id = [&](int64_t g) -> int {
    const int64_t row = g * 64 + r;
    return (seg && g < a.ngroups && row < N) ? a.nbr_ext[row * 7 + dir] : Ni;               // (N = the zero record)
  };
  // what a segment contributes: CI floats of x (a mean for an aux record), packed node-type counts, 1 / count
  struct Piece { float v[CI]; unsigned t0, t1; float inv; };
  constexpr int RW = CI == 4 ? 8 : 16;
  auto load_piece = [&](int id, Piece& P) {
    const float4* ar = reinterpret_cast<const float4*>(a.rec + (int64_t)id * RW);
    const float4 q0 = ar[0], qt = ar[CI == 4 ? 1 : 2];
    P.v[0] = q0.x; P.v[1] = q0.y; P.v[2] = q0.z; P.v[3] = q0.w;
    if constexpr (CI > 4) {
      const float4 q1 = ar[1];
      P.v[4] = q1.x; P.v[5] = q1.y; P.v[6] = q1.z; P.v[7] = q1.w;
    }
    P.t0 = __float_as_uint(qt.x); P.t1 = __float_as_uint(qt.y); P.inv = qt.z;
  };
  auto store_piece = [&](float* cdb, const Piece& P) {
    if (!seg) return;
    float* o = cdb + r * LD + dir * cpd;
#pragma unroll
    for (int c = 0; c < CI; ++c)
      if (c < cin) o[c] = P.v[c];
#pragma unroll
    for (int t = 0; t < 8; ++t)
      if (t < nt) o[cin + t] = (float)(((t < 4 ? P.t0 : P.t1) >> (8 * (t & 3))) & 255u) * P.inv;
  };

  const int64_t G = gridDim.x;
  int64_t g = blockIdx.x;
  Piece P;
  int id_next;
  {
    const int id0 = load_id(g);
    load_piece(id0, P);
    id_next = load_id(g + G);
  }
  for (int it = 0; g < a.ngroups; g += G, ++it) {
    float* cdb = cd[it & 1];
    store_piece(cdb, P);                      // (waits for the pieces of this group: requested one group ago)
    // do all rows of the group belong to one batch element?
    const int64_t row0 = g * 64;
    bool same = true;
    if (a.stats && tid < 64 && row0 + tid < N) same = a.bid[row0 + tid] == a.bid[row0];
    const bool uni = __syncthreads_and(same);
    // next group's pieces (ids arrived during the previous group) and the ids after them: in flight under the MFMAs / stores
    load_piece(id_next, P);
    id_next = load_id(g + 2 * G);
    float s_ = 0.f, q_ = 0.f;
    for (int t = 0; t < nmt; ++t) {
      const int mt = mt0 + t;
      f32x16 acc;
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = 0.f;
      const float* ap = &cdb[(mt * 32 + ln) * LD + lk];
#pragma unroll
      for (int kk = 0; kk < KP / 2; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * kk], bw[kk], acc, 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int64_t row = row0 + mt * 32 + (i & 3) + 8 * (i >> 2) + 4 * lk;
        const float v = acc[i] + bias;
        if (row < N) {
          a.out[row * a.ldc + slice * 32 + ln] = v;
          if (a.stats) {
            if (uni) { s_ += v; q_ += v * v; }
            else {
              double* so = a.stats + ((int64_t)a.bid[row] * a.stats_ld + slice * 32 + ln) * 2;
              unsafeAtomicAdd(so, (double)v);
              unsafeAtomicAdd(so + 1, (double)v * (double)v);
            }
          }
        }
      }
    }
    if (a.stats) {
      s_ += __shfl_xor(s_, 32);
      q_ += __shfl_xor(q_, 32);
      if (lk == 0) { red[wv][ln][0] = s_; red[wv][ln][1] = q_; }
      __syncthreads();
      if (tid < a.cout) {
        const int sl = tid >> 5, j = tid & 31, o = wide ? 4 : 2;
        *reinterpret_cast<float2*>(a.stats_part + (g * a.cout + tid) * 2) =
            make_float2(red[sl][j][0] + red[sl + o][j][0], red[sl][j][1] + red[sl + o][j][1]);
      }
    }
  }
}

struct NarrowOutArgs {
  const float* P; int64_t ldp; int cout; int64_t N;
  const int32_t* seg_ptr; const int32_t* col;
  const float* tt;                                   // [N, cout]: node-type term + bias of every row (narrow_type_term_kernel)
  float* out; int64_t ldc;
};

// tt[r, o] = bias[o] + sum_{dir, t} type_frac[r, dir * nt + t] * W[dir * (C + nt) + C + t, o]: constant per (doctree depth,
// weights), computed once and cached by the caller; lane = (row, direction) like the aggregation kernel.
template <int CO>
__global__ void __launch_bounds__(256) narrow_type_term_kernel(const float* __restrict__ tf, int64_t ldt, int nt, int64_t N,
                                                                const float* __restrict__ W, int C, int cout,
                                                                const float* __restrict__ bias, float* __restrict__ tt) {
  const int tid = threadIdx.x, d = tid & 7;
  const int64_t row = (int64_t)blockIdx.x * 32 + (tid >> 3);
  float acc[CO];
#pragma unroll
  for (int o = 0; o < CO; ++o) acc[o] = 0.f;
  if (row < N) {
    if (d < 7) {
      for (int t = 0; t < nt; ++t) {
        const float f = tf[row * ldt + d * nt + t];
        const float* wr = W + ((int64_t)d * (C + nt) + C + t) * cout;
#pragma unroll
        for (int o = 0; o < CO; ++o) acc[o] = fmaf(f, wr[o < cout ? o : 0], acc[o]);
      }
    } else if (bias) {
#pragma unroll
      for (int o = 0; o < CO; ++o) acc[o] = bias[o < cout ? o : 0];
    }
  }
#pragma unroll
  for (int o = 0; o < CO; ++o) {
    acc[o] += __shfl_xor(acc[o], 1);
    acc[o] += __shfl_xor(acc[o], 2);
    acc[o] += __shfl_xor(acc[o], 4);
  }
  if (d == 0 && row < N) {
#pragma unroll
    for (int o = 0; o < CO; ++o)
      if (o < cout) tt[row * cout + o] = acc[o];
  }
}

// lane = (row, slot): slots 0..6 = the seven directions (three dependent loads: segment bounds -> column -> cout floats of
// P), slot 7 = the row's cached type term.  32 rows per 256-thread block; the eight lanes of a row meet by xor-shuffles.
template <int CO>
__global__ void __launch_bounds__(256) narrow_out_kernel(const NarrowOutArgs a) {
  const int tid = threadIdx.x;
  const int d = tid & 7;
  const int64_t row = (int64_t)blockIdx.x * 32 + (tid >> 3);
  float acc[CO];
#pragma unroll
  for (int o = 0; o < CO; ++o) acc[o] = 0.f;
  if (row < a.N) {
    if (d < 7) {
      const int32_t b = a.seg_ptr[row * 7 + d], e = a.seg_ptr[row * 7 + d + 1];
      for (int32_t p = b; p < e; ++p) {
        const float* pr = a.P + (int64_t)a.col[p] * a.ldp + d * a.cout;
        float v[CO];
#pragma unroll
        for (int o = 0; o < CO; ++o) v[o] = pr[o < a.cout ? o : 0];
#pragma unroll
        for (int o = 0; o < CO; ++o) acc[o] += v[o];
      }
      const float cnt = (float)(e - b > 1 ? e - b : 1);
#pragma unroll
      for (int o = 0; o < CO; ++o) acc[o] = o < a.cout ? acc[o] / cnt : 0.f;
    } else if (a.tt) {
#pragma unroll
      for (int o = 0; o < CO; ++o) {
        const float v = a.tt[row * a.cout + (o < a.cout ? o : 0)];
        acc[o] = o < a.cout ? v : 0.f;
      }
    }
  }
#pragma unroll
  for (int o = 0; o < CO; ++o) {
    acc[o] += __shfl_xor(acc[o], 1);
    acc[o] += __shfl_xor(acc[o], 2);
    acc[o] += __shfl_xor(acc[o], 4);
  }
  if (d == 0 && row < a.N) {
#pragma unroll
    for (int o = 0; o < CO; ++o)
      if (o < a.cout) a.out[row * a.ldc + o] = acc[o];
  }
}

// Wd[c, dir * cout + o] = W[dir * (C + nt) + c, o], zero-padded to `pw` columns: the dense operand of the projection
__global__ void narrow_out_pack_kernel(const float* __restrict__ W, int C, int nt, int cout, int pw, float* __restrict__ Wd) {
  const int64_t total = (int64_t)C * pw;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(t / pw), q = (int)(t - (int64_t)c * pw);
    const int dir = q / cout, o = q - dir * cout;
    Wd[t] = q < 7 * cout ? W[((int64_t)dir * (C + nt) + c) * cout + o] : 0.f;
  }
}

}  // namespace

extern "C" int ofx_graphconv_narrow_in(const float* x, int64_t ldx, int cin, int64_t n_nodes, const int32_t* seg_ptr,
                                       const int32_t* col, const uint8_t* node_type, int nt, const float* W, int cout,
                                       const float* bias, const int32_t* batch_id, float* out, int64_t ldc, double* stats,
                                       int64_t stats_ld, void* ws, size_t ws_bytes, void* stream) {
  if (!x || !seg_ptr || !col || !W || !out || cin < 1 || cin > 8 || ldx < cin || nt < 0 || nt > 8 || (nt > 0 && !node_type) ||
      (cout != 64 && cout != 128) || ldc < cout || n_nodes < 0 ||
      7 * (cin + nt) > 96)
    return OFX_EINVAL;
  if (n_nodes == 0) return OFX_OK;
  const int64_t nblk = ofx_cdiv(n_nodes, 64);
  if (stats && (!batch_id || stats_ld < cout || !ws || ws_bytes < (size_t)nblk * cout * 2 * sizeof(float))) return OFX_EINVAL;
  hipStream_t st = ofx_stream(stream);
  NarrowInArgs a = {x, ldx, cin, n_nodes, seg_ptr, col, node_type, nt, W, cout, bias, out, ldc,
                    batch_id, (float*)ws, stats, stats_ld};
  const int K = 7 * (cin + nt);
  const unsigned nb = (unsigned)nblk;
  if (K <= 64) { if (cin <= 4) narrow_in_kernel<64, 4><<<nb, 512, 0, st>>>(a); else narrow_in_kernel<64, 8><<<nb, 512, 0, st>>>(a); }
  else { if (cin <= 4) narrow_in_kernel<96, 4><<<nb, 512, 0, st>>>(a); else narrow_in_kernel<96, 8><<<nb, 512, 0, st>>>(a); }
  OFX_LAUNCH_CHECK();
  if (stats) {
    GemmArgs g = {};
    g.M = n_nodes; g.N = cout; g.bid = batch_id; g.stats = stats; g.stats_ld = stats_ld; g.stats_part = (float*)ws;
    return ofx_launch_stats_reduce(g, 64, st);
  }
  return OFX_OK;
}

// The same convolution through the gather table (round 6, narrow_in2_kernel above).  nbr_ext / multi_seg / n_multi: the
// branch-free table of the graph depth (ofx_graph_primary_ext); aux: scratch of (n_nodes + n_multi + 1) * (cin <= 4 ? 32 : 64)
// bytes, 16-B aligned (one record per gatherable id).
extern "C" int ofx_graphconv_narrow_in_tab(const float* x, int64_t ldx, int cin, int64_t n_nodes, const int32_t* seg_ptr,
                                           const int32_t* col, const int32_t* nbr_ext, const int32_t* multi_seg,
                                           int64_t n_multi, void* aux, const uint8_t* node_type, int nt, const float* W,
                                           int cout, const float* bias, const int32_t* batch_id, float* out, int64_t ldc,
                                           double* stats, int64_t stats_ld, void* ws, size_t ws_bytes, void* stream) {
  if (!x || !seg_ptr || !col || !nbr_ext || !aux || ((uintptr_t)aux & 15) || n_multi < 0 || (n_multi > 0 && !multi_seg) || !W ||
      !out || cin < 1 || cin > 8 || ldx < cin || nt < 0 || nt > 8 || (nt > 0 && !node_type) || (cout != 64 && cout != 128) ||
      ldc < cout || n_nodes < 0 || 7 * (cin + nt) > 96 || n_nodes + 1 + n_multi >= (1ll << 31))
    return OFX_EINVAL;
  if (n_nodes == 0) return OFX_OK;
  const int64_t ngroups = ofx_cdiv(n_nodes, 64);
  if (stats && (!batch_id || stats_ld < cout || !ws || ws_bytes < (size_t)ngroups * cout * 2 * sizeof(float))) return OFX_EINVAL;
  hipStream_t st = ofx_stream(stream);
  {
    const int64_t nbn = ofx_cdiv(n_nodes + 1, 256);
    const unsigned gn_ = (unsigned)(nbn < 16384 ? nbn : 16384);
    if (cin <= 4) narrow_rec_nodes_kernel<4><<<gn_, 256, 0, st>>>(x, ldx, cin, n_nodes, node_type, nt, (float*)aux);
    else narrow_rec_nodes_kernel<8><<<gn_, 256, 0, st>>>(x, ldx, cin, n_nodes, node_type, nt, (float*)aux);
    if (n_multi > 0) {
      const unsigned ga_ = (unsigned)ofx_cdiv(n_multi, 256);
      if (cin <= 4) narrow_rec_aux_kernel<4><<<ga_, 256, 0, st>>>(x, ldx, cin, n_nodes, seg_ptr, col, multi_seg, n_multi, node_type, nt, (float*)aux);
      else narrow_rec_aux_kernel<8><<<ga_, 256, 0, st>>>(x, ldx, cin, n_nodes, seg_ptr, col, multi_seg, n_multi, node_type, nt, (float*)aux);
    }
  }
  OFX_LAUNCH_CHECK();
  NarrowIn2Args a = {x, ldx, cin, n_nodes, nbr_ext, (const float*)aux, nt, W, cout, bias, out, ldc,
                     batch_id, (float*)ws, stats, stats_ld, ngroups};
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) cus = n;
  }
  const int64_t want = (int64_t)cus * 2;                 // two 512-thread blocks per CU
  const unsigned nb = (unsigned)(ngroups < want ? ngroups : want);
  const int K = 7 * (cin + nt);
  // (K = 70 -- three channels + seven node types, the depth-8 input convolution of the feature net -- gets its own width:
  // 36 weight registers instead of 48 keep the block at four waves per SIMD without scratch)
  if (K <= 64) { if (cin <= 4) narrow_in2_kernel<64, 4><<<nb, 512, 0, st>>>(a); else narrow_in2_kernel<64, 8><<<nb, 512, 0, st>>>(a); }
  else if (K <= 72) { if (cin <= 4) narrow_in2_kernel<72, 4><<<nb, 512, 0, st>>>(a); else narrow_in2_kernel<72, 8><<<nb, 512, 0, st>>>(a); }
  else { if (cin <= 4) narrow_in2_kernel<96, 4><<<nb, 512, 0, st>>>(a); else narrow_in2_kernel<96, 8><<<nb, 512, 0, st>>>(a); }
  OFX_LAUNCH_CHECK();
  if (stats) {
    GemmArgs g = {};
    g.M = n_nodes; g.N = cout; g.bid = batch_id; g.stats = stats; g.stats_ld = stats_ld; g.stats_part = (float*)ws;
    return ofx_launch_stats_reduce(g, 64, st);
  }
  return OFX_OK;
}

extern "C" int ofx_narrow_out_pack(const float* W, int C, int nt, int cout, int pw, float* Wd, void* stream) {
  if (!W || !Wd || C < 1 || nt < 0 || cout < 1 || cout > 8 || pw < 7 * cout) return OFX_EINVAL;
  narrow_out_pack_kernel<<<ofx_grid((int64_t)C * pw, 256), 256, 0, ofx_stream(stream)>>>(W, C, nt, cout, pw, Wd);
  OFX_LAUNCH_CHECK();
  return OFX_OK;
}

extern "C" int ofx_narrow_out_type_term(const float* type_frac, int64_t ldt, int nt, int64_t n_nodes, const float* W, int C,
                                        int cout, const float* bias, float* tt, void* stream) {
  if (!W || !tt || C < 1 || cout < 1 || cout > 8 || nt < 0 || (nt > 0 && (!type_frac || ldt < 7 * nt)) || n_nodes < 0)
    return OFX_EINVAL;
  if (n_nodes == 0) return OFX_OK;
  const unsigned nblk = (unsigned)ofx_cdiv(n_nodes, 32);
  hipStream_t st = ofx_stream(stream);
  if (cout <= 4) narrow_type_term_kernel<4><<<nblk, 256, 0, st>>>(type_frac, ldt, nt, n_nodes, W, C, cout, bias, tt);
  else narrow_type_term_kernel<8><<<nblk, 256, 0, st>>>(type_frac, ldt, nt, n_nodes, W, C, cout, bias, tt);
  OFX_LAUNCH_CHECK();
  return OFX_OK;
}

extern "C" int ofx_graphconv_narrow_out(const float* P, int64_t ldp, int cout, int64_t n_nodes, const int32_t* seg_ptr,
                                        const int32_t* col, const float* type_term, float* out, int64_t ldc, void* stream) {
  if (!P || !seg_ptr || !col || !out || cout < 1 || cout > 8 || ldp < 7 * cout || ldc < cout || n_nodes < 0) return OFX_EINVAL;
  if (n_nodes == 0) return OFX_OK;
  NarrowOutArgs a = {P, ldp, cout, n_nodes, seg_ptr, col, type_term, out, ldc};
  hipStream_t st = ofx_stream(stream);
  const unsigned nblk = (unsigned)ofx_cdiv(n_nodes, 32);
  if (cout <= 4) narrow_out_kernel<4><<<nblk, 256, 0, st>>>(a);
  else narrow_out_kernel<8><<<nblk, 256, 0, st>>>(a);
  OFX_LAUNCH_CHECK();
  return OFX_OK;
}

// libofx: DualOctreeGroupNorm (reference models/networks/modules.py:262-330).
// HBM-bound: stats = one read of x, apply = one read + one write (the reference
// makes 3 scatter_add passes + 2 index_selects, ~7 passes).
//
// Statistics are accumulated per (batch element, channel) as fp64 (sum, sum of
// squares): fp32 per-thread partials over <= 64 rows, fp64 across threads / blocks
// (LDS then global atomics).  finalize reproduces the reference's arithmetic:
// inv_count = 1/(count*cpg + eps), mean = S*inv_count, centred variance
// sum((x-mean)^2)*inv_count = (SS - 2*mean*S + n*mean^2)*inv_count evaluated in
// fp64, rstd = 1/sqrt(var + eps).
#include "ofx_planes.h"     // the 16-bit operand-pair formats of the planes GraphConv (g2_split2)

// 64 rows per block: the kernel is latency/MLP-bound, not atomic-bound -- it needs >> 256 CUs x 8
// resident blocks with several 16-B loads in flight per lane (a 512-row block left 1.6 blocks per CU
// and ~0.85 TB/s).  Each block reduces its rows in LDS and issues one fp64 atomic pair per channel.
constexpr int GN_ROWS_PER_BLOCK = 64;

// rows per block of the two statistics kernels.  Each block ends with ONE fp64 atomic pair per channel, so an address
// (batch element, channel) receives blocks / batch_size atomics, and a same-address device atomic costs ~100-270 ns
// (measured: 780 k rows x 32 channels, batch 4, 64-row blocks -> 3 k atomics per address, 318 us = 0.3 TB/s; 64 k rows
// -> 253 per address, 69 us for 8 MB).  Blocks are therefore capped at 128 per batch element (at least 256: enough
// to stream) and own proportionally more consecutive rows.
// Small tensors (the 16 MB concat in the middle of the hr net: 32 768 rows x 128, batch 8) are the other end of the
// same trade: 64-row blocks put 64 atomics on every address -- 13 us of serialised atomics for 3 us of reading
// (28 us per call in the round-4 step trace) -- so a block also owns at least ~512 KB of rows (>= 8 blocks per batch
// element are kept so that the read still spreads over the chip).
static inline int64_t gn_stats_rows(int64_t n, int batch_size, int C = 128) {
  const int64_t chunks = (n + GN_ROWS_PER_BLOCK - 1) / GN_ROWS_PER_BLOCK;
  const int64_t bytes_per_batch = n * (int64_t)C * 4 / (batch_size > 0 ? batch_size : 1);
  int64_t per_batch = bytes_per_batch / (512 << 10);
  per_batch = per_batch < 8 ? 8 : (per_batch > 128 ? 128 : per_batch);
  int64_t cap = per_batch * (int64_t)batch_size;
  if (cap < 64) cap = 64;
  return chunks <= cap ? GN_ROWS_PER_BLOCK : GN_ROWS_PER_BLOCK * ((chunks + cap - 1) / cap);
}

__global__ void __launch_bounds__(256) gn_stats_kernel(const float* __restrict__ x, int64_t ldx, int64_t n, int C,
                                                       const int32_t* __restrict__ bid, double* __restrict__ sums,
                                                       int64_t rpb) {
  __shared__ int sb[256];
  __shared__ float sv[256][8];
  const int CT = C >> 2;                 // float4 lanes per row (<= 256)
  const int RP = 256 / CT;               // rows per pass
  const int cl = threadIdx.x % CT, rl = threadIdx.x / CT;
  int cb = -1;
  float s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
  auto flush_direct = [&](int b, const float* ps, const float* pq) {
    if (b < 0) return;
    double* o = sums + ((int64_t)b * C + cl * 4) * 2;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      unsafeAtomicAdd(o + 2 * k, (double)ps[k]);      // hardware global_atomic_add_f64 (no CAS loop)
      unsafeAtomicAdd(o + 2 * k + 1, (double)pq[k]);
    }
  };
  auto accum = [&](int b, const float4& v) {
    if (b != cb) {
      flush_direct(cb, s, q);
      cb = b;
      s[0] = s[1] = s[2] = s[3] = 0.f;
      q[0] = q[1] = q[2] = q[3] = 0.f;
    }
    s[0] += v.x; s[1] += v.y; s[2] += v.z; s[3] += v.w;
    q[0] += v.x * v.x; q[1] += v.y * v.y; q[2] += v.z * v.z; q[3] += v.w * v.w;
  };
  // a block owns `rpb` CONSECUTIVE rows (64, or more for long narrow tensors: see gn_stats_rows): consecutive rows
  // share a batch element except at the few run boundaries, where a thread flushes its run with its own atomics
  const int64_t r_begin = (int64_t)blockIdx.x * rpb;
  const int64_t r_end = r_begin + rpb < n ? r_begin + rpb : n;
  if (rl < RP) {
    int64_t r = r_begin + rl;
    // 4 independent 16-B loads in flight per lane
    for (; r + 3 * (int64_t)RP < r_end; r += 4 * (int64_t)RP) {
      const int b0 = bid[r], b1 = bid[r + RP], b2 = bid[r + 2 * RP], b3 = bid[r + 3 * RP];
      const float4 v0 = *reinterpret_cast<const float4*>(x + r * ldx + cl * 4);
      const float4 v1 = *reinterpret_cast<const float4*>(x + (r + RP) * ldx + cl * 4);
      const float4 v2 = *reinterpret_cast<const float4*>(x + (r + 2 * RP) * ldx + cl * 4);
      const float4 v3 = *reinterpret_cast<const float4*>(x + (r + 3 * RP) * ldx + cl * 4);
      accum(b0, v0); accum(b1, v1); accum(b2, v2); accum(b3, v3);
    }
    for (; r < r_end; r += RP) accum(bid[r], *reinterpret_cast<const float4*>(x + r * ldx + cl * 4));
  }
  sb[threadIdx.x] = (rl < RP) ? cb : -1;
#pragma unroll
  for (int k = 0; k < 4; ++k) { sv[threadIdx.x][k] = s[k]; sv[threadIdx.x][4 + k] = q[k]; }
  __syncthreads();
  if (rl == 0) {
    double ds[4] = {0, 0, 0, 0}, dq[4] = {0, 0, 0, 0};
    for (int j = 0; j < RP; ++j) {
      const int t = j * CT + cl;
      const int b = sb[t];
      if (b < 0) continue;
      if (b == cb) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { ds[k] += (double)sv[t][k]; dq[k] += (double)sv[t][4 + k]; }
      } else {
        flush_direct(b, &sv[t][0], &sv[t][4]);
      }
    }
    if (cb >= 0) {
      double* o = sums + ((int64_t)cb * C + cl * 4) * 2;
#pragma unroll
      for (int k = 0; k < 4; ++k) { unsafeAtomicAdd(o + 2 * k, ds[k]); unsafeAtomicAdd(o + 2 * k + 1, dq[k]); }
    }
  }
}

extern "C" int ofx_gn_stats(const float* x, int64_t ldx, int64_t n, int C, const int32_t* batch_id, int batch_size,
                            double* sums, void* stream) {
  if (!x || !batch_id || !sums || n < 0 || C < 4 || (C & 3) || C > 1024 || ldx < C || (ldx & 3) ||
      ((uintptr_t)x & 15) || batch_size < 1)
    return OFX_EINVAL;
  hipStream_t st = ofx_stream(stream);
  if (hipMemsetAsync(sums, 0, sizeof(double) * 2 * (size_t)batch_size * C, st) != hipSuccess) return OFX_ELAUNCH;
  if (n > 0) gn_stats_kernel<<<(int)ofx_cdiv(n, gn_stats_rows(n, batch_size, C)), 256, 0, st>>>(x, ldx, n, C, batch_id, sums,
                                                                                             gn_stats_rows(n, batch_size, C));
  OFX_LAUNCH_CHECK();
  return OFX_OK;
}

// The same, ACCUMULATING into `sums` (the caller zeroed it: one fill per network forward for all statistics buffers
// instead of a memset node in front of every statistics launch of a captured step).
extern "C" int ofx_gn_stats_acc(const float* x, int64_t ldx, int64_t n, int C, const int32_t* batch_id, int batch_size,
                                double* sums, void* stream) {
  if (!x || !batch_id || !sums || n < 0 || C < 4 || (C & 3) || C > 1024 || ldx < C || (ldx & 3) ||
      ((uintptr_t)x & 15) || batch_size < 1)
    return OFX_EINVAL;
  if (n > 0) gn_stats_kernel<<<(int)ofx_cdiv(n, gn_stats_rows(n, batch_size, C)), 256, 0, ofx_stream(stream)>>>(
      x, ldx, n, C, batch_id, sums, gn_stats_rows(n, batch_size, C));
  OFX_LAUNCH_CHECK();
  return OFX_OK;
}

__global__ void gn_finalize_kernel(const double* __restrict__ sums, const float* __restrict__ count, int B, int C, int G,
                                   float eps, float count_eps, float* __restrict__ mean, float* __restrict__ rstd) {
  const int cpg = C / G;
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < B * G; t += gridDim.x * blockDim.x) {
    const int b = t / G, g = t - b * G;
    double S = 0, SS = 0;
    for (int c = g * cpg; c < (g + 1) * cpg; ++c) {
      S += sums[((int64_t)b * C + c) * 2];
      SS += sums[((int64_t)b * C + c) * 2 + 1];
    }
    const float cnt = count[b] * (float)cpg;            // modules.py:301-302 (fp32)
    const float inv = 1.0f / (cnt + count_eps);         // reference adds eps to the COUNT (:302); nn.GroupNorm does not
    const double m = S * (double)inv;
    const double ssd = SS - 2.0 * m * S + (double)cnt * m * m;
    const double var = (ssd > 0 ? ssd : 0) * (double)inv;
    const float rs = (float)(1.0 / sqrt(var + (double)eps));
    for (int c = g * cpg; c < (g + 1) * cpg; ++c) {
      mean[(int64_t)b * C + c] = (float)m;
      rstd[(int64_t)b * C + c] = rs;
    }
  }
}

extern "C" int ofx_gn_finalize(const double* sums, const float* count, int batch_size, int C, int groups, float eps,
                               float count_eps, float* mean, float* rstd, void* stream) {
  if (!sums || !count || !mean || !rstd || batch_size < 1 || C < 1 || groups < 1 || C % groups) return OFX_EINVAL;
  gn_finalize_kernel<<<ofx_grid((int64_t)batch_size * groups, 64), 64, 0, ofx_stream(stream)>>>(
      sums, count, batch_size, C, groups, eps, count_eps, mean, rstd);
  OFX_LAUNCH_CHECK();
  return OFX_OK;
}

__device__ __forceinline__ float ofx_apply_act(float v, int act) {
  if (act == OFX_ACT_SILU) return v / (1.f + __expf(-v));
  if (act == OFX_ACT_GELU) return 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));
  return v;
}

// Normalise + affine + activation, one read and one write of the tensor.  Block = 64 consecutive rows; thread =
// (float4 of channels, row lane): the affine parameters are loaded ONCE per thread and (mean, rstd) only when the
// batch element changes along its rows (the first version re-loaded all four per float4 -- five 16-B loads per 16 B of
// payload -- and ran at 1.2 TB/s); four independent row loads in flight per lane.
// MODE 0: fp32 rows out.  MODE 2 / 1: operand planes for the LDS-DMA GraphConv (ofx_gemm2.hip): 2 = per 32-channel
// chunk one 128-B line [bf16 hi x 32 | bf16 lo x 32] (y = hi + lo to 2^-17) -- the bytes of the fp32 row, so `out` may
// alias an fp32-shaped buffer, including x itself (a thread reads its 16 B of the chunk and writes 8 B of each half;
// the 8 threads of a chunk are lanes of one wave instruction, whose loads all return before its stores issue);
// 1 = fp16 row-major.
constexpr int GN_APPLY_ROWS = 64;
// (mean, rstd) of one (batch element, group) from the fp64 sums -- the arithmetic of gn_finalize_kernel, so a launch
// that finalises on the fly (FIN) gives the same bits as ofx_gn_finalize + a launch that reads mean / rstd.
struct GnFin { const double* sums; const float* count; int G; float eps, count_eps; };
__device__ __forceinline__ void gn_group_stats(const GnFin& f, int b, int g, int C, float& mean, float& rstd) {
  const int cpg = C / f.G;
  double S = 0, SS = 0;
  for (int c = g * cpg; c < (g + 1) * cpg; ++c) {
    S += f.sums[((int64_t)b * C + c) * 2];
    SS += f.sums[((int64_t)b * C + c) * 2 + 1];
  }
  const float cnt = f.count[b] * (float)cpg;
  const float inv = 1.0f / (cnt + f.count_eps);
  const double m = S * (double)inv;
  const double ssd = SS - 2.0 * m * S + (double)cnt * m * m;
  const double var = (ssd > 0 ? ssd : 0) * (double)inv;
  mean = (float)m;
  rstd = (float)(1.0 / sqrt(var + (double)f.eps));
}
// FIN: no mean / rstd arrays -- the block derives them from the statistics sums itself (one launch less per norm:
// the finalize kernel was 4.7 us + a launch boundary, 19 times per hr step).  The (mean, rstd) rows of the batch
// elements of the block's first and last row are computed once per block into LDS; a row of any other batch element
// (tiny graph levels, the aux blocks' scattered source rows) computes its four channels' groups directly.
template <int MODE, bool FIN>
__global__ void __launch_bounds__(256) gn_apply_kernel(const float* __restrict__ x, int64_t ldx, int64_t n, int C,
                                                       const int32_t* __restrict__ bid, const float* __restrict__ mean,
                                                       const float* __restrict__ rstd, const GnFin fin,
                                                       const float* __restrict__ w,
                                                       const float* __restrict__ bias, int act, char* __restrict__ out,
                                                       int64_t ldo, int64_t aux_blocks,
                                                       const int32_t* __restrict__ seg_ptr,
                                                       const int32_t* __restrict__ col,
                                                       const int32_t* __restrict__ multi_seg, int64_t n_multi,
                                                       char* __restrict__ aux, const int32_t* __restrict__ aux_plan) {
  const int CT = C >> 2, RP = 256 / CT;
  const int cl = threadIdx.x % CT, rl = threadIdx.x / CT;
  __shared__ float fin_ms[FIN ? 2 * 2 * 1024 : 1];       // [slot][mean | rstd][C]
  // owned aux rows of this block (aux_plan): row id, segment size and up to four source rows each, fetched at block START
  // by one thread per aux row -- the dependent chain plan -> multi_seg -> seg_ptr -> col runs under the main rows, and
  // the aux phase at the end is ONE load stage deep (<= 64 owned rows per block: a block has 64 rows, every aux row has
  // >= 2 sources in it)
  constexpr int AUX_LDS = 64;
  __shared__ int aux_v[AUX_LDS], aux_n[AUX_LDS], aux_src[AUX_LDS][4];
  int fb0 = -1, fb1 = -1;
  // aux blocks are INTERLEAVED with the main blocks in dispatch order (block i is the a-th aux block if the running
  // share i * A / T steps at i): both kinds are then resident together for the whole launch, and the aux blocks'
  // dependent-load chains run under the main blocks' streaming.  (Until round 4 the aux blocks were the FIRST A blocks of
  // the grid: 4 400 of them at depth 6 against 2 048 resident block slots, so they ran alone, latency-bound, before any
  // main block started -- 16 % more rows cost 42-50 % more time: 42.7 -> 63.7 us at depth 6, C = 128.)
  const int64_t gT = gridDim.x;
  const int64_t a_before = aux_blocks > 0 ? ((int64_t)blockIdx.x * aux_blocks) / gT : 0;
  const bool is_aux = aux_blocks > 0 && (((int64_t)blockIdx.x + 1) * aux_blocks) / gT > a_before;
  const int64_t aux_id = a_before, main_id = (int64_t)blockIdx.x - a_before;
  // the first four rows of this thread are requested BEFORE the statistics are finalised: the rows come from HBM, the
  // sums from L2, and the block would otherwise sit through a chain of dependent loads + fp64 arithmetic + a barrier
  // before its first payload byte is even asked for (measured: +23 % on the whole launch)
  const int64_t r_begin0 = main_id * GN_APPLY_ROWS;
  const int64_t r_end0 = r_begin0 + GN_APPLY_ROWS < n ? r_begin0 + GN_APPLY_ROWS : n;
  const bool pre = FIN && !is_aux && rl < RP && r_begin0 + rl + 3 * (int64_t)RP < r_end0;
  float4 pv0, pv1, pv2, pv3;
  int pb0 = 0, pb1 = 0, pb2 = 0, pb3 = 0;
  if (pre) {
    const int64_t r = r_begin0 + rl;
    const int c_ = cl * 4;
    pb0 = bid[r]; pb1 = bid[r + RP]; pb2 = bid[r + 2 * RP]; pb3 = bid[r + 3 * RP];
    pv0 = *reinterpret_cast<const float4*>(x + r * ldx + c_);
    pv1 = *reinterpret_cast<const float4*>(x + (r + RP) * ldx + c_);
    pv2 = *reinterpret_cast<const float4*>(x + (r + 2 * RP) * ldx + c_);
    pv3 = *reinterpret_cast<const float4*>(x + (r + 3 * RP) * ldx + c_);
  }
  if (FIN) {
    if (!is_aux) {
      fb0 = bid[r_begin0];
      fb1 = bid[r_end0 - 1];
    } else {
      // aux rows are ordered by segment = row * 7 + direction, i.e. by row, i.e. by batch element, and every source of
      // a segment lies in the batch element of its row: the block's aux rows span the batch elements of its first and
      // last segment's rows (two dependent loads; the zero row v = 0 needs no statistics)
      const int64_t v0 = aux_id * RP, v1e = v0 + RP - 1 < n_multi ? v0 + RP - 1 : n_multi;
      if (n_multi > 0) {
        fb0 = bid[multi_seg[(v0 > 0 ? v0 : 1) - 1] / 7];
        fb1 = bid[multi_seg[(v1e > 0 ? v1e : 1) - 1] / 7];
      }
    }
    const int cpg = C / fin.G;
    for (int t = threadIdx.x; t < 2 * fin.G && fb0 >= 0; t += 256) {
      const int slot = t / fin.G, g = t - slot * fin.G;
      if (slot == 1 && fb1 == fb0) continue;
      float mm, rr;
      gn_group_stats(fin, slot ? fb1 : fb0, g, C, mm, rr);
      for (int cc = g * cpg; cc < (g + 1) * cpg; ++cc) {
        fin_ms[(slot * 2 + 0) * 1024 + cc] = mm;
        fin_ms[(slot * 2 + 1) * 1024 + cc] = rr;
      }
    }
    __syncthreads();
  }
  int own_b = 0, own_e = 0;
  if (!is_aux && aux && aux_plan && MODE != 0) {
    const int64_t mb_ = (n + GN_APPLY_ROWS - 1) / GN_APPLY_ROWS;
    own_b = aux_plan[main_id]; own_e = aux_plan[main_id + 1];
    const int i = own_b + (int)threadIdx.x;
    if (i < own_e && (int)threadIdx.x < AUX_LDS) {
      const int v = aux_plan[mb_ + 1 + i];
      const int64_t sgm = multi_seg[v - 1];
      const int32_t pa = seg_ptr[sgm], pz = seg_ptr[sgm + 1];
      aux_v[threadIdx.x] = v;
      aux_n[threadIdx.x] = pz - pa;
#pragma unroll
      for (int k = 0; k < 4; ++k) aux_src[threadIdx.x][k] = col[pa + k < pz ? pa + k : pa];
      if (pz - pa > 4) aux_n[threadIdx.x] = -(int)(pa + 1);          // long segment: walked through col at the end
    }
  }
  // (threads beyond the last whole row lane -- 256 % (C / 4) != 0 -- take no part, but stay for the barrier below)
  const bool active = rl < RP;
  const int c = cl * 4;
  const float4 ww = *reinterpret_cast<const float4*>(w + c);
  const float4 bb = *reinterpret_cast<const float4*>(bias + c);
  int cb = -1;
  float4 m = make_float4(0.f, 0.f, 0.f, 0.f), rs = m;
  auto norm = [&](int b, const float4& v) {
    if (b != cb) {
      cb = b;
      if (FIN) {
        if (b == fb0 || b == fb1) {
          const int slot = b == fb0 ? 0 : 1;
          m = *reinterpret_cast<const float4*>(fin_ms + (slot * 2 + 0) * 1024 + c);
          rs = *reinterpret_cast<const float4*>(fin_ms + (slot * 2 + 1) * 1024 + c);
        } else {
          const int cpg = C / fin.G;
          float mm[4], rr[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            if (k == 0 || (c + k) / cpg != (c + k - 1) / cpg) gn_group_stats(fin, b, (c + k) / cpg, C, mm[k], rr[k]);
            else { mm[k] = mm[k - 1]; rr[k] = rr[k - 1]; }
          }
          m = make_float4(mm[0], mm[1], mm[2], mm[3]);
          rs = make_float4(rr[0], rr[1], rr[2], rr[3]);
        }
      } else {
        m = *reinterpret_cast<const float4*>(mean + (int64_t)b * C + c);
        rs = *reinterpret_cast<const float4*>(rstd + (int64_t)b * C + c);
      }
    }
    return make_float4(ofx_apply_act((v.x - m.x) * rs.x * ww.x + bb.x, act),
                       ofx_apply_act((v.y - m.y) * rs.y * ww.y + bb.y, act),
                       ofx_apply_act((v.z - m.z) * rs.z * ww.z + bb.z, act),
                       ofx_apply_act((v.w - m.w) * rs.w * ww.w + bb.w, act));
  };
  auto store = [&](char* orow, const float4& y) {
    if (MODE == 0) {
      *reinterpret_cast<float4*>(orow + (int64_t)c * 4) = y;
    } else if (MODE == 2 || MODE == 3) {
      unsigned h0, h1, l0, l1;
      g2_split2(MODE, y.x, y.y, h0, l0);
      g2_split2(MODE, y.z, y.w, h1, l1);
      char* o = orow + (c >> 5) * 128 + (c & 31) * 2;
      *reinterpret_cast<uint2*>(o) = make_uint2(h0, h1);
      *reinterpret_cast<uint2*>(o + 64) = make_uint2(l0, l1);
    } else {
      const _Float16 a0 = (_Float16)y.x, a1 = (_Float16)y.y, a2 = (_Float16)y.z, a3 = (_Float16)y.w;
      uint2 o;
      o.x = (unsigned)__builtin_bit_cast(unsigned short, a0) | ((unsigned)__builtin_bit_cast(unsigned short, a1) << 16);
      o.y = (unsigned)__builtin_bit_cast(unsigned short, a2) | ((unsigned)__builtin_bit_cast(unsigned short, a3) << 16);
      *reinterpret_cast<uint2*>(orow + c * 2) = o;
    }
  };
  // ---- aux rows of the consuming GraphConv (its multi-neighbour pre-pass, folded into this launch): aux[0] = the zero
  // row, aux[1 + v] = mean over segment multi_seg[v] of the NORMALISED source rows.  Reads the raw x (complete before
  // this launch), so it does not depend on the main blocks -- out must not alias x when aux is requested.
  auto aux_row = [&](int64_t v) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (v > 0) {
      const int64_t sgm = multi_seg[v - 1];
      const int32_t pa = seg_ptr[sgm], pe = seg_ptr[sgm + 1];
      // segments of a dual-octree face hold 2, 4 or (rarely) more finer neighbours: groups of four with all column
      // ids, then all batch ids and rows in flight together (a serial walk is a chain of 3 dependent loads per edge)
      for (int32_t p0 = pa; p0 < pe; p0 += 4) {
        int64_t sr[4];
        int bb4[4];
        float4 xv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) sr[k] = col[p0 + k < pe ? p0 + k : pa];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          bb4[k] = bid[sr[k]];
          xv[k] = *reinterpret_cast<const float4*>(x + sr[k] * ldx + c);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float4 y = norm(bb4[k], xv[k]);
          const float wgt = p0 + k < pe ? 1.f : 0.f;
          acc.x += wgt * y.x; acc.y += wgt * y.y; acc.z += wgt * y.z; acc.w += wgt * y.w;
        }
      }
      const float inv = 1.f / (float)(pe - pa);
      acc.x *= inv; acc.y *= inv; acc.z *= inv; acc.w *= inv;
    }
    store(aux + v * ldo, acc);
  };
  // aux_plan (optional, int32): [mb + 1] ptr | owned list (ptr[mb] aux row ids, grouped by the MAIN block -- 64
  // consecutive rows -- that holds ALL sources of the row) | n_left | n_left leftover ids (sources in several blocks,
  // and the zero row 0).  A main block writes its owned aux rows right after its own rows, while their sources are still
  // in its L1 / the XCD's L2: the four fine neighbours of a coarse leaf's face are siblings, i.e. rows of one aligned
  // group of eight.  Without a plan every aux row is a leftover (round 3: 16 % more rows cost 42-50 % more time, because
  // the aux blocks re-read their source rows from HBM at another time than the main pass).
  const int64_t mb = (n + GN_APPLY_ROWS - 1) / GN_APPLY_ROWS;
  if (is_aux) {
    if (!active) return;
    const int64_t idx = aux_id * RP + rl;
    if (aux_plan) {
      const int32_t n_own = aux_plan[mb];
      const int32_t n_left = aux_plan[mb + 1 + n_own];
      if (idx < n_left) aux_row(aux_plan[mb + 2 + n_own + idx]);
    } else if (idx <= n_multi) {
      aux_row(idx);
    }
    return;
  }
  const int64_t r_begin = r_begin0, r_end = r_end0;
  int64_t r = r_begin + rl;
  if (active) {
  if (pre) {
    store(out + r * ldo, norm(pb0, pv0));
    store(out + (r + RP) * ldo, norm(pb1, pv1));
    store(out + (r + 2 * RP) * ldo, norm(pb2, pv2));
    store(out + (r + 3 * RP) * ldo, norm(pb3, pv3));
    r += 4 * (int64_t)RP;
  }
  for (; r + 3 * (int64_t)RP < r_end; r += 4 * (int64_t)RP) {
    const int b0 = bid[r], b1 = bid[r + RP], b2 = bid[r + 2 * RP], b3 = bid[r + 3 * RP];
    const float4 v0 = *reinterpret_cast<const float4*>(x + r * ldx + c);
    const float4 v1 = *reinterpret_cast<const float4*>(x + (r + RP) * ldx + c);
    const float4 v2 = *reinterpret_cast<const float4*>(x + (r + 2 * RP) * ldx + c);
    const float4 v3 = *reinterpret_cast<const float4*>(x + (r + 3 * RP) * ldx + c);
    store(out + r * ldo, norm(b0, v0));
    store(out + (r + RP) * ldo, norm(b1, v1));
    store(out + (r + 2 * RP) * ldo, norm(b2, v2));
    store(out + (r + 3 * RP) * ldo, norm(b3, v3));
  }
  for (; r < r_end; r += RP) store(out + r * ldo, norm(bid[r], *reinterpret_cast<const float4*>(x + r * ldx + c)));
  }
  if (aux && aux_plan && MODE != 0) {
    // the aux rows whose sources all lie in this block's 64 rows: averaged from the OUTPUT this block has just written
    // (hi + lo re-assembled: 3 instructions per element instead of the ~25 of normalise + SiLU -- GroupNorm-apply is as
    // much VALU- as HBM-bound, and re-normalising four sources per aux row cost 42-50 % on top of the main pass for 16 %
    // more rows).  Same values as the stand-alone pre-pass of the planes GraphConv (planes_multi_mean_kernel).
    const int32_t pb = own_b, pe_ = own_e;
    if (pb < pe_) {
      __syncthreads();                   // (workgroup-scope fence: this block's stores to `out` and the LDS index rows)
      auto joined = [&](int64_t sr) {
        const char* orow = out + sr * ldo;
        float4 y;
        if (MODE == 2 || MODE == 3) {
          const char* o = orow + (c >> 5) * 128 + (c & 31) * 2;
          const uint2 hi = *reinterpret_cast<const uint2*>(o), lo = *reinterpret_cast<const uint2*>(o + 64);
          g2_join2(MODE, hi.x, lo.x, y.x, y.y);
          g2_join2(MODE, hi.y, lo.y, y.z, y.w);
        } else {
          const uint2 hv = *reinterpret_cast<const uint2*>(orow + c * 2);
          y.x = g2_f16_lo(hv.x); y.y = g2_f16_hi(hv.x); y.z = g2_f16_lo(hv.y); y.w = g2_f16_hi(hv.y);
        }
        return y;
      };
      if (active) {
        for (int32_t i = rl; i < pe_ - pb; i += RP) {
          float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
          int64_t v;
          float inv;
          if (i < AUX_LDS && aux_n[i] > 0) {
            v = aux_v[i];
            const int cnt = aux_n[i];
            float4 y[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) y[k] = joined(aux_src[i][k]);           // four rows in flight
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const float wgt = k < cnt ? 1.f : 0.f;
              acc.x += wgt * y[k].x; acc.y += wgt * y[k].y; acc.z += wgt * y[k].z; acc.w += wgt * y[k].w;
            }
            inv = 1.f / (float)cnt;
          } else {                                                            // (> 64 owned rows or a long segment)
            v = aux_plan[mb + 1 + pb + i];
            const int64_t sgm = multi_seg[v - 1];
            const int32_t pa = seg_ptr[sgm], pz = seg_ptr[sgm + 1];
            for (int32_t p = pa; p < pz; ++p) {
              const float4 y = joined(col[p]);
              acc.x += y.x; acc.y += y.y; acc.z += y.z; acc.w += y.w;
            }
            inv = 1.f / (float)(pz - pa);
          }
          acc.x *= inv; acc.y *= inv; acc.z *= inv; acc.w *= inv;
          store(aux + v * ldo, acc);
        }
      }
    }
  }
}

// mean / rstd: from ofx_gn_finalize -- or both NULL with (sums, count, groups, eps, count_eps): finalised on the fly.
static bool gn_fin_args(const float* mean, const float* rstd, const double* sums, const float* count, int C, int groups,
                        GnFin& f) {
  if (mean || rstd) return mean && rstd && !(((uintptr_t)mean | (uintptr_t)rstd) & 15);
  if (!sums || !count || groups < 1 || C % groups || C > 1024) return false;
  f.sums = sums; f.count = count; f.G = groups;
  return true;
}

extern "C" int ofx_gn_apply(const float* x, int64_t ldx, int64_t n, int C, const int32_t* batch_id, const float* mean,
                            const float* rstd, const double* sums, const float* count, int groups, float eps,
                            float count_eps, const float* w, const float* bias, int act, float* out, int64_t ldo,
                            void* stream) {
  GnFin f = {nullptr, nullptr, 1, eps, count_eps};
  if (!x || !batch_id || !w || !bias || !out || n < 0 || C < 4 || (C & 3) || C > 1024 || ldx < C ||
      ldo < C || (ldx & 3) || (ldo & 3) || ((uintptr_t)x & 15) || ((uintptr_t)out & 15) || ((uintptr_t)w & 15) ||
      ((uintptr_t)bias & 15) || act < 0 || act > 2 || !gn_fin_args(mean, rstd, sums, count, C, groups, f))
    return OFX_EINVAL;
  if (n > 0) {
    const int64_t mb = ofx_cdiv(n, GN_APPLY_ROWS);
    if (mean)
      gn_apply_kernel<0, false><<<(int)mb, 256, 0, ofx_stream(stream)>>>(x, ldx, n, C, batch_id, mean, rstd, f, w, bias, act,
                                                                         (char*)out, ldo * 4, 0, nullptr, nullptr,
                                                                         nullptr, 0, nullptr, nullptr);
    else
      gn_apply_kernel<0, true><<<(int)mb, 256, 0, ofx_stream(stream)>>>(x, ldx, n, C, batch_id, nullptr, nullptr, f, w, bias,
                                                                        act, (char*)out, ldo * 4, 0, nullptr, nullptr,
                                                                        nullptr, 0, nullptr, nullptr);
  }
  OFX_LAUNCH_CHECK();
  return OFX_OK;
}

extern "C" int ofx_gn_apply_rows(void) { return GN_APPLY_ROWS; }

extern "C" int ofx_gn_apply_planes(const float* x, int64_t ldx, int64_t n, int C, const int32_t* batch_id,
                                   const float* mean, const float* rstd, const double* sums, const float* count,
                                   int groups, float eps, float count_eps, const float* w, const float* bias, int act,
                                   int mode, void* out, int64_t ldo_bytes, const int32_t* seg_ptr, const int32_t* col,
                                   const int32_t* multi_seg, int64_t n_multi, void* aux, const int32_t* aux_plan,
                                   int64_t aux_left, void* stream) {
  const int chunk = g2_pairs(mode) ? 32 : 64;
  GnFin f = {nullptr, nullptr, 1, eps, count_eps};
  if (mode < 1 || mode > 3 || !x || !batch_id || !w || !bias || !out || n < 0 || C < chunk ||
      (C % chunk) || C > 1024 || ldx < C || (ldx & 3) || ldo_bytes < (int64_t)C * (g2_pairs(mode) ? 4 : 2) ||
      (ldo_bytes & 15) || ((uintptr_t)x & 15) || ((uintptr_t)out & 127) || ((uintptr_t)w & 15) ||
      ((uintptr_t)bias & 15) || act < 0 || act > 2 || !gn_fin_args(mean, rstd, sums, count, C, groups, f))
    return OFX_EINVAL;
  if (aux && (!seg_ptr || !col || n_multi < 0 || (n_multi > 0 && !multi_seg) || ((uintptr_t)aux & 127) ||
              (const void*)out == (const void*)x || (aux_plan && (aux_left < 1 || aux_left > n_multi + 1))))
    return OFX_EINVAL;
  if (n > 0) {
    const int64_t mb = ofx_cdiv(n, GN_APPLY_ROWS);
    // one aux row per row lane; with a plan only the leftover rows (aux_left of them, incl. the zero row) need blocks
    const int64_t ab = aux ? ofx_cdiv(aux_plan ? aux_left : n_multi + 1, 256 / (C >> 2)) : 0;
    const int grid = (int)(mb + ab);
    hipStream_t st = ofx_stream(stream);
#define GN_GO(M_, F_)                                                                                              \
  gn_apply_kernel<M_, F_><<<grid, 256, 0, st>>>(x, ldx, n, C, batch_id, mean, rstd, f, w, bias, act, (char*)out,   \
                                                ldo_bytes, ab, seg_ptr, col, multi_seg, n_multi, (char*)aux,       \
                                                aux ? aux_plan : nullptr)
    if (mode == 2) { if (mean) GN_GO(2, false); else GN_GO(2, true); }
    else if (mode == 3) { if (mean) GN_GO(3, false); else GN_GO(3, true); }
    else { if (mean) GN_GO(1, false); else GN_GO(1, true); }
#undef GN_GO
  }
  OFX_LAUNCH_CHECK();
  return OFX_OK;
}

// ---------------------------------------------------------------------------------
// Sibling-octet mapping (round 6): the aux rows for free.
// An aux row is the mean over one multi-neighbour segment.  On a dual octree 83 % of them (depth 6 / 7 / 8 of the shell
// trees; 100 % at depth 5) are the four finer neighbours across one face of a coarse leaf: four SIBLINGS, i.e. rows of one
// aligned group of eight of the depth-d part of the node order (ofx.h: "octet" o = rows 8 o - shift .. 8 o - shift + 7,
// shift = the number of rows that pad the coarse-leaf prefix to a multiple of eight).  gn_apply_kernel gives a thread the
// rows rl, rl + RP, ...; an aux row therefore had to wait for the block's stores (barrier), re-read four rows (eight 8-B
// loads per thread for hi and lo), re-join them -- 2.5 x the memory instructions of a main row: +27...48 % of the launch
// for 13-17 % more rows (tools/gn_probe.py).  Here a thread owns ALL EIGHT rows of an octet for its four
// channels: the eight loads are in flight together, the normalised values stay in registers (x is overwritten in place:
// 32 VGPRs), and an aux row of that octet is a masked sum of registers -- no reload, no LDS, no barrier, no dependent
// load behind the main rows (the octet's entries (aux row id, 8-bit sibling mask) were requested before them).
// Entries come from the host plan (dual_octree.DualOctree.oct_plan): oct_ptr [n_oct + 1], oct_ent [n_own] (v, mask).
// Aux rows whose sources do not sit in one octet (17 %: segments of a leaf two levels up -- 7, 10, 13 or 16 sources from
// several octets -- and sibling leaves of the coarse prefix that straddle a group boundary) are LEFTOVERS: re-normalised
// from x by extra blocks behind the main blocks, through a host-flattened source list (left_head / left_src).
// The value summed for an aux row is the value a reader of the stored planes sees (hi + lo, not the unsplit fp32), so
// the rows equal the stand-alone pre-pass of the planes GraphConv (planes_multi_mean_kernel) bit for bit.
// Pair modes: the two lanes that hold the eight channels 8 p .. 8 p + 7 of a row (lane parity = parity of the float4 slot:
// C / 4 is even) swap halves through a DPP quad permute, so that the even lane stores the 16 B of hi words and the odd
// lane the 16 B of lo words: ONE 16-B store per lane and row, whole 128-B lines per wave instruction, instead of two 8-B
// stores that each touch half of every line.  Callers keep both lanes of a pair on the same path (the row / octet / entry
// conditions are uniform over the lanes of a row).
template <int MODE>
__device__ __forceinline__ float4 gn_store_planes(char* orow, int c, const float4& y) {
  if (MODE == 2 || MODE == 3) {
    unsigned h0, h1, l0, l1;
    g2_split2(MODE, y.x, y.y, h0, l0);
    g2_split2(MODE, y.z, y.w, h1, l1);
    const bool odd = (c & 4) != 0;
    // what the partner needs: the even lane's lo words, the odd lane's hi words
    const unsigned s0 = odd ? h0 : l0, s1 = odd ? h1 : l1;
    const unsigned r0 = (unsigned)__builtin_amdgcn_mov_dpp((int)s0, 0xB1, 0xF, 0xF, true);      // quad_perm [1, 0, 3, 2]
    const unsigned r1 = (unsigned)__builtin_amdgcn_mov_dpp((int)s1, 0xB1, 0xF, 0xF, true);
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    const u4 o = odd ? u4{r0, r1, l0, l1} : u4{h0, h1, r0, r1};
    // even lane: hi words of channels c .. c + 7 at the line's first half; odd lane: lo words of c - 4 .. c + 3 at its second
    *reinterpret_cast<u4*>(orow + (c >> 5) * 128 + ((c & 24) * 2) + (odd ? 64 : 0)) = o;
    float4 j;
    g2_join2(MODE, h0, l0, j.x, j.y);
    g2_join2(MODE, h1, l1, j.z, j.w);
    return j;
  } else {
    const unsigned p0 = g2_pk_f16(y.x, y.y), p1 = g2_pk_f16(y.z, y.w);
    *reinterpret_cast<uint2*>(orow + c * 2) = make_uint2(p0, p1);
    return make_float4(g2_f16_lo(p0), g2_f16_hi(p0), g2_f16_lo(p1), g2_f16_hi(p1));
  }
}

// FIN: no mean / rstd arrays -- every block derives (mean, rstd) of the batch elements of its first and last row from the
// fp64 sums into LDS (gn_apply_kernel's scheme: one launch less per norm), AFTER its rows have been requested.
template <int MODE, bool FIN>
__global__ void __launch_bounds__(256) gn_apply_oct_kernel(const float* __restrict__ x, int64_t ldx, int64_t n, int C,
                                                           const int32_t* __restrict__ bid,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           const GnFin fin,
                                                           const float* __restrict__ w, const float* __restrict__ bias,
                                                           int act, char* __restrict__ out, int64_t ldo,
                                                           int64_t aux_blocks, char* __restrict__ aux,
                                                           const int32_t* __restrict__ oct_ptr,
                                                           const int2* __restrict__ oct_ent, int64_t n_oct, int shift,
                                                           const int4* __restrict__ lhead,
                                                           const int32_t* __restrict__ lsrc, int64_t n_left,
                                                           int left_at_end) {
  __shared__ float fin_ms[FIN ? 2 * 2 * 1024 : 1];       // [slot][mean | rstd][C]
  const int CT = C >> 2, RP = 256 / CT;
  const int cl = threadIdx.x % CT, rl = threadIdx.x / CT;
  const int c = cl * 4;
  // leftover blocks: interleaved with the main blocks in dispatch order as in gn_apply_kernel (block i is the a-th
  // leftover block if the running share i * A / T steps at i), or all behind them (ofx_set_gn_left_place(1), A/B).
  // What a leftover row costs is the re-read of its 4 ... 16 source rows (depth 8, C = 128: 0.35 GB on top of the main
  // pass's 1.66 GB): interleaved, a leftover block runs about when the main blocks of its sources do and finds part of
  // them in the memory-side cache -- measured 51.4 us (interleaved) vs 58.2 us (behind) at depth 6, C = 128; equal at
  // depth 8 (tools/gn_probe_oct_parts.py).
  const int64_t gT = gridDim.x;
  int64_t a_before;
  bool is_aux;
  if (left_at_end) {
    is_aux = (int64_t)blockIdx.x >= gT - aux_blocks;
    a_before = is_aux ? (int64_t)blockIdx.x - (gT - aux_blocks) : 0;
  } else {
    a_before = aux_blocks > 0 ? ((int64_t)blockIdx.x * aux_blocks) / gT : 0;
    is_aux = aux_blocks > 0 && (((int64_t)blockIdx.x + 1) * aux_blocks) / gT > a_before;
  }
  const int64_t aux_id = a_before, main_id = (int64_t)blockIdx.x - a_before;
  const int64_t o = main_id * RP + rl;
  // (256 % (C / 4) != 0 leaves spare lanes; they and the lanes past the last octet / leftover row only attend the barrier)
  const bool active = rl < RP && (is_aux ? aux_id * RP + rl < n_left : o < n_oct);

  int p0 = 0, p1 = 0;
  float4 v[8];
  int b0v = 0;
  unsigned bdiff = 0;              // bit j: row j of the octet belongs to another batch element than row 0 (the leaf prefix)
  if (!is_aux && active) {
    // the octet's entry range and the eight batch ids first (their latency runs under the row loads; the ids collapse
    // to one id + a mask -- siblings share a batch element -- before the rows land, which frees seven registers), then
    // the eight rows
    p0 = oct_ptr[o];
    p1 = oct_ptr[o + 1];
    const int64_t r0 = 8 * o - shift;
    int b[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      int64_t r = r0 + j;
      r = r < 0 ? 0 : (r < n ? r : n - 1);                // (rows outside [0, n) are loaded clamped and never stored)
      b[j] = bid[r];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      int64_t r = r0 + j;
      r = r < 0 ? 0 : (r < n ? r : n - 1);
      v[j] = *reinterpret_cast<const float4*>(x + r * ldx + c);
    }
    b0v = b[0];
#pragma unroll
    for (int j = 1; j < 8; ++j) bdiff |= (b[j] != b0v ? 1u : 0u) << j;
  }
  int4 hd = make_int4(0, 0, 0, 0);
  if (is_aux && active) hd = lhead[aux_id * RP + rl];
  int fb0 = -1, fb1 = -1;
  if (FIN) {
    if (!is_aux) {
      int64_t ra = 8 * (main_id * RP) - shift, rz = 8 * (main_id * RP + RP) - shift - 1;
      ra = ra < 0 ? 0 : (ra < n ? ra : n - 1);
      rz = rz < 0 ? 0 : (rz < n ? rz : n - 1);
      fb0 = bid[ra];
      fb1 = bid[rz];
    } else {
      const int64_t ia = aux_id * RP, iz = ia + RP - 1 < n_left ? ia + RP - 1 : n_left - 1;
      fb0 = lhead[ia < n_left ? ia : n_left - 1].w;
      fb1 = lhead[iz].w;
    }
    const int cpg = C / fin.G;
    for (int t = threadIdx.x; t < 2 * fin.G; t += 256) {
      const int slot = t / fin.G, g = t - slot * fin.G;
      if (slot == 1 && fb1 == fb0) continue;
      float mm, rr;
      gn_group_stats(fin, slot ? fb1 : fb0, g, C, mm, rr);
      for (int cc = g * cpg; cc < (g + 1) * cpg; ++cc) {
        fin_ms[(slot * 2 + 0) * 1024 + cc] = mm;
        fin_ms[(slot * 2 + 1) * 1024 + cc] = rr;
      }
    }
    __syncthreads();
  }
  if (!active) return;
  const float4 ww = *reinterpret_cast<const float4*>(w + c);
  const float4 bb = *reinterpret_cast<const float4*>(bias + c);
  int cb = -1;
  float4 m = make_float4(0.f, 0.f, 0.f, 0.f), rs = m;
  auto stats_of = [&](int bq) {
    if (FIN) {
      if (bq == fb0 || bq == fb1) {
        const int slot = bq == fb0 ? 0 : 1;
        m = *reinterpret_cast<const float4*>(fin_ms + (slot * 2 + 0) * 1024 + c);
        rs = *reinterpret_cast<const float4*>(fin_ms + (slot * 2 + 1) * 1024 + c);
      } else {                                            // (tiny graph levels: a third batch element inside one block)
        // four consecutive channels lie in at most two groups (channels per group >= 2): first and last channel's
        const int cpg = C / fin.G;
        const int ga = c / cpg, gb = (c + 3) / cpg;
        float ma, ra, mb2, rb2;
        gn_group_stats(fin, bq, ga, C, ma, ra);
        mb2 = ma; rb2 = ra;
        if (gb != ga) gn_group_stats(fin, bq, gb, C, mb2, rb2);
        m = make_float4(ma, (c + 1) / cpg == ga ? ma : mb2, (c + 2) / cpg == ga ? ma : mb2, mb2);
        rs = make_float4(ra, (c + 1) / cpg == ga ? ra : rb2, (c + 2) / cpg == ga ? ra : rb2, rb2);
      }
    } else {
      m = *reinterpret_cast<const float4*>(mean + (int64_t)bq * C + c);
      rs = *reinterpret_cast<const float4*>(rstd + (int64_t)bq * C + c);
    }
  };
  auto norm = [&](int bq, const float4& q) {
    if (bq != cb) {
      cb = bq;
      stats_of(bq);
    }
    return make_float4(ofx_apply_act((q.x - m.x) * rs.x * ww.x + bb.x, act),
                       ofx_apply_act((q.y - m.y) * rs.y * ww.y + bb.y, act),
                       ofx_apply_act((q.z - m.z) * rs.z * ww.z + bb.z, act),
                       ofx_apply_act((q.w - m.w) * rs.w * ww.w + bb.w, act));
  };
  if (is_aux) {
    // ---- leftover aux rows: mean of the re-normalised source rows (raw x is complete before this launch).  The host
    // flattened the chain plan -> multi_seg -> seg_ptr -> col into lhead[idx] = (aux row, first source slot, sources,
    // batch element) + lsrc[]: two dependent loads before the rows instead of four, and the statistics of the batch
    // element (every source of a segment lies in the batch element of its row) are requested together with the sources.
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (hd.z > 0) {
      cb = hd.w;
      stats_of(hd.w);
      const int32_t pa = hd.y, pe = hd.y + hd.z;
      for (int32_t q0 = pa; q0 < pe; q0 += 4) {
        int64_t sr[4];
        float4 xv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) sr[k] = lsrc[q0 + k < pe ? q0 + k : pa];
#pragma unroll
        for (int k = 0; k < 4; ++k) xv[k] = *reinterpret_cast<const float4*>(x + sr[k] * ldx + c);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float4 y = norm(hd.w, xv[k]);
          const float wgt = q0 + k < pe ? 1.f : 0.f;
          acc.x += wgt * y.x; acc.y += wgt * y.y; acc.z += wgt * y.z; acc.w += wgt * y.w;
        }
      }
      const float inv = 1.f / (float)hd.z;
      acc.x *= inv; acc.y *= inv; acc.z *= inv; acc.w *= inv;
    }
    gn_store_planes<MODE>(aux + (int64_t)hd.x * ldo, c, acc);
    return;
  }
  // ---- main rows of the octet; the first entries are requested now (two dependent loads deep, both L2 hits)
  constexpr int NPRE = 2;              // (three prefetched entries cost the FIN instantiation its fifth wave per SIMD: 98 VGPRs)
  int2 ent[NPRE];
#pragma unroll
  for (int k = 0; k < NPRE; ++k) ent[k] = p0 + k < p1 ? oct_ent[p0 + k] : make_int2(0, 0);
  const int64_t r0 = 8 * o - shift;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int64_t r = r0 + j;
    int bq = b0v;
    if ((bdiff >> j) & 1u) bq = bid[r < 0 ? 0 : (r < n ? r : n - 1)];
    const float4 y = norm(bq, v[j]);
    if (r >= 0 && r < n) v[j] = gn_store_planes<MODE>(out + r * ldo, c, y);
  }
  auto entry = [&](const int2 en) {
    const unsigned msk = (unsigned)en.y;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const bool on = (msk >> j) & 1u;                    // (a select, not a 0/1 weight: a NaN-poisoned sibling outside the mask stays out)
      acc.x += on ? v[j].x : 0.f; acc.y += on ? v[j].y : 0.f; acc.z += on ? v[j].z : 0.f; acc.w += on ? v[j].w : 0.f;
    }
    const float inv = 1.f / (float)__popc(msk);
    acc.x *= inv; acc.y *= inv; acc.z *= inv; acc.w *= inv;
    gn_store_planes<MODE>(aux + (int64_t)en.x * ldo, c, acc);
  };
#pragma unroll
  for (int k = 0; k < NPRE; ++k)
    if (p0 + k < p1) entry(ent[k]);
  for (int p = p0 + NPRE; p < p1; ++p) entry(oct_ent[p]);
}

static int g_gn_left_end = 0;              // A/B knob (ofx_set_gn_left_place): 1 = leftover blocks behind the main blocks, 0 = interleaved (default: measured faster)
extern "C" int ofx_set_gn_left_place(int at_end) { g_gn_left_end = at_end ? 1 : 0; return OFX_OK; }

extern "C" int ofx_gn_apply_planes_oct(const float* x, int64_t ldx, int64_t n, int C, const int32_t* batch_id,
                                       const float* mean, const float* rstd, const double* sums, const float* count,
                                       int groups, float eps, float count_eps, const float* w, const float* bias, int act,
                                       int mode, void* out, int64_t ldo_bytes, int64_t n_multi, void* aux,
                                       const int32_t* oct_ptr, const int32_t* oct_ent, int64_t n_own, int shift,
                                       const int32_t* left_head, const int32_t* left_src, int64_t n_left,
                                       void* stream) {
  const int chunk = g2_pairs(mode) ? 32 : 64;
  GnFin f = {nullptr, nullptr, 1, eps, count_eps};
  if (mode < 1 || mode > 3 || !x || !batch_id || !w || !bias || !out || !aux || n < 0 || C < chunk ||
      (C % chunk) || C > 1024 || ldx < C || (ldx & 3) || ldo_bytes < (int64_t)C * (g2_pairs(mode) ? 4 : 2) ||
      (ldo_bytes & 15) || ((uintptr_t)x & 15) || ((uintptr_t)out & 127) || ((uintptr_t)aux & 127) ||
      ((uintptr_t)w & 15) || ((uintptr_t)bias & 15) || !gn_fin_args(mean, rstd, sums, count, C, groups, f) || act < 0 ||
      act > 2 || (const void*)out == (const void*)x || n_multi < 0 || !oct_ptr || n_own < 0 || n_own > n_multi || (n_own > 0 && (!oct_ent || ((uintptr_t)oct_ent & 7))) || shift < 0 ||
      shift > 7 || !left_head || ((uintptr_t)left_head & 15) || !left_src || n_left < 1 || n_left > n_multi + 1 ||
      n_own + n_left != n_multi + 1)
    return OFX_EINVAL;
  if (n > 0) {
    const int RP = 256 / (C >> 2);
    const int64_t n_oct = ofx_cdiv(n + shift, 8);
    const int64_t mb = ofx_cdiv(n_oct, RP), ab = ofx_cdiv(n_left, RP);
    const int grid = (int)(mb + ab);
    hipStream_t st = ofx_stream(stream);
#define GN_GO(M_, F_)                                                                                                \
  gn_apply_oct_kernel<M_, F_><<<grid, 256, 0, st>>>(x, ldx, n, C, batch_id, mean, rstd, f, w, bias, act, (char*)out,     \
                                                    ldo_bytes, ab, (char*)aux, oct_ptr,                               \
                                                    reinterpret_cast<const int2*>(oct_ent), n_oct, shift,             \
                                                    reinterpret_cast<const int4*>(left_head), left_src, n_left,       \
                                                    g_gn_left_end)
    if (mode == 2) { if (mean) GN_GO(2, false); else GN_GO(2, true); }
    else if (mode == 3) { if (mean) GN_GO(3, false); else GN_GO(3, true); }
    else { if (mean) GN_GO(1, false); else GN_GO(1, true); }
#undef GN_GO
  }
  OFX_LAUNCH_CHECK();
  return OFX_OK;
}

// elementwise activation (shared with ofx_misc entry point)
__global__ void act_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n, int act) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    y[i] = ofx_apply_act(x[i], act);
}
extern "C" int ofx_act(const float* x, float* y, int64_t n, int act, void* stream) {
  if (n < 0 || (n > 0 && (!x || !y)) || act < 0 || act > 2) return OFX_EINVAL;
  if (n > 0) act_kernel<<<ofx_grid(n, 256), 256, 0, ofx_stream(stream)>>>(x, y, n, act);
  OFX_LAUNCH_CHECK();
  return OFX_OK;
}

// ---------------------------------------------------------------------------------
// Backward of DualOctreeGroupNorm (+ fused SiLU / GELU) -- training path, autograd of modules.py:291-326.
// With n = count*cpg, inv = 1/(n + count_eps), mu = inv*sum(x), d = x - mu, v = inv*sum(d^2), r = (v+eps)^-1/2,
// y = d*r*gamma + beta, g = dL/dy * act'(y):
//   per (batch element, channel):  A = sum_i g,  Bc = sum_i g*x          (one pass over x and dy)
//   per (batch element, group):    S1 = sum_c gamma*A,  S2 = sum_c gamma*(Bc - mu*A),
//                                  dv = -r^3*S2/2,  c2 = 2*inv*dv,  c3 = -inv*(r*S1 + c2*mu*count_eps)
//   dx = gamma*r*g + c2*(x - mu) + c3,   dgamma = sum_b r*(Bc - mu*A),   dbeta = sum_b A.
// (sum_i d = mu*count_eps because the reference divides by n + eps, not n.)
__device__ __forceinline__ float ofx_act_grad(float y, int act) {
  if (act == OFX_ACT_SILU) {
    const float s = 1.f / (1.f + __expf(-y));
    return s * (1.f + y * (1.f - s));
  }
  if (act == OFX_ACT_GELU)
    return 0.5f * (1.f + erff(y * 0.70710678118654752440f)) + y * 0.3989422804014327f * __expf(-0.5f * y * y);
  return 1.f;
}

// Same blocking as gn_stats_kernel: 64 rows per block, thread = (row lane, float4 of channels), per-thread runs in
// fp32, the row lanes meet in LDS and the block issues ONE fp64 atomic pair per channel (per batch element it
// touches) -- per-thread atomics made this kernel 6x slower than the forward statistics.
__global__ void __launch_bounds__(256) gn_bwd_stats_kernel(const float* __restrict__ x, int64_t ldx,
                                                           const float* __restrict__ dy, int64_t ldy, int64_t n, int C,
                                                           const int32_t* __restrict__ bid,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           const float* __restrict__ w, const float* __restrict__ bias,
                                                           int act, double* __restrict__ sums, int64_t rpb) {
  __shared__ int sb[256];
  __shared__ float sv[256][8];
  const int CT = C >> 2, RP = 256 / CT;
  const int cl = threadIdx.x % CT, rl = threadIdx.x / CT;
  int cb = -1;
  float s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
  auto flush_direct = [&](int b, const float* ps, const float* pq) {
    if (b < 0) return;
    double* o = sums + ((int64_t)b * C + cl * 4) * 2;
#pragma unroll
    for (int k = 0; k < 4; ++k) { unsafeAtomicAdd(o + 2 * k, (double)ps[k]); unsafeAtomicAdd(o + 2 * k + 1, (double)pq[k]); }
  };
  if (rl < RP) {
    const float4 ww = *reinterpret_cast<const float4*>(w + cl * 4);
    const float4 bb = *reinterpret_cast<const float4*>(bias + cl * 4);
    const int64_t r_begin = (int64_t)blockIdx.x * rpb;
    const int64_t r_end = r_begin + rpb < n ? r_begin + rpb : n;
    for (int64_t r = r_begin + rl; r < r_end; r += RP) {
      const int b = bid[r];
      if (b != cb) {
        flush_direct(cb, s, q);
        cb = b;
        s[0] = s[1] = s[2] = s[3] = 0.f;
        q[0] = q[1] = q[2] = q[3] = 0.f;
      }
      const float4 xv = *reinterpret_cast<const float4*>(x + r * ldx + cl * 4);
      float4 g = *reinterpret_cast<const float4*>(dy + r * ldy + cl * 4);
      if (act != OFX_ACT_NONE) {
        const float4 m = *reinterpret_cast<const float4*>(mean + (int64_t)b * C + cl * 4);
        const float4 rs = *reinterpret_cast<const float4*>(rstd + (int64_t)b * C + cl * 4);
        g.x *= ofx_act_grad((xv.x - m.x) * rs.x * ww.x + bb.x, act);
        g.y *= ofx_act_grad((xv.y - m.y) * rs.y * ww.y + bb.y, act);
        g.z *= ofx_act_grad((xv.z - m.z) * rs.z * ww.z + bb.z, act);
        g.w *= ofx_act_grad((xv.w - m.w) * rs.w * ww.w + bb.w, act);
      }
      s[0] += g.x; s[1] += g.y; s[2] += g.z; s[3] += g.w;
      q[0] += g.x * xv.x; q[1] += g.y * xv.y; q[2] += g.z * xv.z; q[3] += g.w * xv.w;
    }
  }
  sb[threadIdx.x] = (rl < RP) ? cb : -1;
#pragma unroll
  for (int k = 0; k < 4; ++k) { sv[threadIdx.x][k] = s[k]; sv[threadIdx.x][4 + k] = q[k]; }
  __syncthreads();
  if (rl == 0) {
    double ds[4] = {0, 0, 0, 0}, dq[4] = {0, 0, 0, 0};
    for (int j = 0; j < RP; ++j) {
      const int t = j * CT + cl;
      const int b = sb[t];
      if (b < 0) continue;
      if (b == cb) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { ds[k] += (double)sv[t][k]; dq[k] += (double)sv[t][4 + k]; }
      } else {
        flush_direct(b, &sv[t][0], &sv[t][4]);
      }
    }
    if (cb >= 0) {
      double* o = sums + ((int64_t)cb * C + cl * 4) * 2;
#pragma unroll
      for (int k = 0; k < 4; ++k) { unsafeAtomicAdd(o + 2 * k, ds[k]); unsafeAtomicAdd(o + 2 * k + 1, dq[k]); }
    }
  }
}

__global__ void gn_bwd_finalize_kernel(const double* __restrict__ sums, const float* __restrict__ count, int B, int C,
                                       int G, float count_eps, const float* __restrict__ mean,
                                       const float* __restrict__ rstd, const float* __restrict__ w,
                                       float* __restrict__ coef, float* __restrict__ dgamma, float* __restrict__ dbeta) {
  const int cpg = C / G;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < B * G) {
    const int b = t / G, g = t - b * G;
    const double mu = mean[(int64_t)b * C + g * cpg], r = rstd[(int64_t)b * C + g * cpg];
    double S1 = 0, S2 = 0;
    for (int c = g * cpg; c < (g + 1) * cpg; ++c) {
      const double A = sums[((int64_t)b * C + c) * 2], Bc = sums[((int64_t)b * C + c) * 2 + 1];
      S1 += (double)w[c] * A;
      S2 += (double)w[c] * (Bc - mu * A);
    }
    const double nn = (double)(count[b] * (float)cpg);
    const double inv = 1.0 / (double)((float)nn + count_eps);
    const double dv = -0.5 * r * r * r * S2;
    const double c2 = 2.0 * inv * dv;
    const double c3 = -inv * (r * S1 + c2 * mu * (double)count_eps);
    for (int c = g * cpg; c < (g + 1) * cpg; ++c) {
      float* o = coef + ((int64_t)b * C + c) * 3;
      o[0] = (float)((double)w[c] * r);
      o[1] = (float)c2;
      o[2] = (float)(c3 - c2 * mu);
    }
  }
  // dgamma / dbeta: one thread per channel over the (few) batch elements, after a grid-wide dependency-free split:
  // they only need sums / mean / rstd, not coef
  if (t < C) {
    double dg = 0, db = 0;
    for (int b = 0; b < B; ++b) {
      const double A = sums[((int64_t)b * C + t) * 2], Bc = sums[((int64_t)b * C + t) * 2 + 1];
      dg += (double)rstd[(int64_t)b * C + t] * (Bc - (double)mean[(int64_t)b * C + t] * A);
      db += A;
    }
    dgamma[t] = (float)dg;
    dbeta[t] = (float)db;
  }
}

__global__ void __launch_bounds__(256) gn_bwd_apply_kernel(const float* __restrict__ x, int64_t ldx,
                                                           const float* __restrict__ dy, int64_t ldy, int64_t n, int C,
                                                           const int32_t* __restrict__ bid, const float* __restrict__ mean,
                                                           const float* __restrict__ rstd, const float* __restrict__ w,
                                                           const float* __restrict__ bias, int act,
                                                           const float* __restrict__ coef, float* __restrict__ dx,
                                                           int64_t lddx) {
  const int CT = C >> 2;
  const int64_t total = n * CT;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = t / CT;
    const int c = (int)(t - r * CT) * 4;
    const int b = bid[r];
    const float4 xv = *reinterpret_cast<const float4*>(x + r * ldx + c);
    float4 g = *reinterpret_cast<const float4*>(dy + r * ldy + c);
    if (act != OFX_ACT_NONE) {
      const float4 m = *reinterpret_cast<const float4*>(mean + (int64_t)b * C + c);
      const float4 rs = *reinterpret_cast<const float4*>(rstd + (int64_t)b * C + c);
      const float4 ww = *reinterpret_cast<const float4*>(w + c);
      const float4 bb = *reinterpret_cast<const float4*>(bias + c);
      g.x *= ofx_act_grad((xv.x - m.x) * rs.x * ww.x + bb.x, act);
      g.y *= ofx_act_grad((xv.y - m.y) * rs.y * ww.y + bb.y, act);
      g.z *= ofx_act_grad((xv.z - m.z) * rs.z * ww.z + bb.z, act);
      g.w *= ofx_act_grad((xv.w - m.w) * rs.w * ww.w + bb.w, act);
    }
    const float* k = coef + ((int64_t)b * C + c) * 3;
    float4 o;
    o.x = k[0] * g.x + k[1] * xv.x + k[2];
    o.y = k[3] * g.y + k[4] * xv.y + k[5];
    o.z = k[6] * g.z + k[7] * xv.z + k[8];
    o.w = k[9] * g.w + k[10] * xv.w + k[11];
    *reinterpret_cast<float4*>(dx + r * lddx + c) = o;
  }
}

extern "C" int ofx_gn_backward(const float* x, int64_t ldx, const float* dy, int64_t ldy, int64_t n, int C,
                               const int32_t* batch_id, int batch_size, const float* count, int groups,
                               float count_eps, const float* mean, const float* rstd, const float* w, const float* bias,
                               int act, double* sums, float* coef, float* dx, int64_t lddx, float* dgamma, float* dbeta,
                               void* stream) {
  if (!x || !dy || !batch_id || !count || !mean || !rstd || !w || !bias || !sums || !coef || !dx || !dgamma || !dbeta ||
      n < 0 || C < 4 || (C & 3) || C > 1024 || groups < 1 || C % groups || batch_size < 1 || ldx < C || ldy < C ||
      lddx < C || ((ldx | ldy | lddx) & 3) || (((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dx) & 15) ||
      act < 0 || act > OFX_ACT_GELU)
    return OFX_EINVAL;
  hipStream_t st = ofx_stream(stream);
  if (hipMemsetAsync(sums, 0, sizeof(double) * 2 * (size_t)batch_size * C, st) != hipSuccess) return OFX_ELAUNCH;
  if (n > 0)
    gn_bwd_stats_kernel<<<(int)ofx_cdiv(n, gn_stats_rows(n, batch_size)), 256, 0, st>>>(x, ldx, dy, ldy, n, C, batch_id, mean, rstd, w, bias,
                                                                                 act, sums, gn_stats_rows(n, batch_size));
  const int work = batch_size * groups > C ? batch_size * groups : C;
  gn_bwd_finalize_kernel<<<(work + 63) / 64, 64, 0, st>>>(sums, count, batch_size, C, groups, count_eps, mean, rstd, w,
                                                          coef, dgamma, dbeta);
  if (n > 0)
    gn_bwd_apply_kernel<<<ofx_grid(n * (C >> 2), 256), 256, 0, st>>>(x, ldx, dy, ldy, n, C, batch_id, mean, rstd, w,
                                                                     bias, act, coef, dx, lddx);
  OFX_LAUNCH_CHECK();
  return OFX_OK;
}

// ---------------------------------------------------------------------------------
// One-launch GroupNorm for the dense layers of the nested lr net (rows of a batch element are contiguous and
// few: 16^3 ... 2^3 tokens): block = one (batch element, group); pass 1 reduces sum / sum of squares over the
// element's rows x the group's channels (fp32 per thread, fp64 across the block), pass 2 normalises, applies the
// affine + activation and writes -- the second read hits L2.  Replaces gn_stats + gn_finalize + gn_apply (three
// launches of ~5 us each for ~4 KB .. 8 MB of data) on that path.
__global__ void __launch_bounds__(256) gn_fused_rows_kernel(const float* __restrict__ x, int64_t ldx, int rows, int C,
                                                            int G, float eps, float count_eps,
                                                            const float* __restrict__ w, const float* __restrict__ bias,
                                                            int act, float* __restrict__ out, int64_t ldo) {
  __shared__ double red[2][256];
  __shared__ float stat[2];
  const int cpg = C / G;
  const int b = blockIdx.x / G, g = blockIdx.x - b * G;
  const float* xb = x + (int64_t)b * rows * ldx + g * cpg;
  float* ob = out + (int64_t)b * rows * ldo + g * cpg;
  const int total = rows * cpg;
  float s = 0.f, q = 0.f;
  for (int t = threadIdx.x; t < total; t += 256) {
    const int r = t / cpg, c = t - r * cpg;
    const float v = xb[(int64_t)r * ldx + c];
    s += v; q += v * v;
  }
  red[0][threadIdx.x] = (double)s;
  red[1][threadIdx.x] = (double)q;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) { red[0][threadIdx.x] += red[0][threadIdx.x + o]; red[1][threadIdx.x] += red[1][threadIdx.x + o]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const double S = red[0][0], SS = red[1][0];
    const float cnt = (float)rows * (float)cpg;
    const float inv = 1.0f / (cnt + count_eps);
    const double m = S * (double)inv;
    const double ssd = SS - 2.0 * m * S + (double)cnt * m * m;
    const double var = (ssd > 0 ? ssd : 0) * (double)inv;
    stat[0] = (float)m;
    stat[1] = (float)(1.0 / sqrt(var + (double)eps));
  }
  __syncthreads();
  const float m = stat[0], rs = stat[1];
  for (int t = threadIdx.x; t < total; t += 256) {
    const int r = t / cpg, c = t - r * cpg;
    const float v = xb[(int64_t)r * ldx + c];
    ob[(int64_t)r * ldo + c] = ofx_apply_act((v - m) * rs * w[g * cpg + c] + bias[g * cpg + c], act);
  }
}

static int g_gn_rows16 = 1;                // A/B knob (ofx_set_gn_rows16)
extern "C" int ofx_set_gn_rows16(int on) { g_gn_rows16 = on ? 1 : 0; return OFX_OK; }

// Same operation, sector-friendly mapping for the common widths (channels per group 2 / 4 / 8 / 16, i.e. C = 64 ... 512
// with 32 groups): a block owns one batch element and 16 consecutive channels (8 / 4 / 2 / 1 groups) -- exactly one
// 64-B sector of every row, read as four float4 by four neighbouring lanes -- instead of one group (8 B of every
// sector for C = 64: the rows were fetched 16 times over by 16 different blocks).  1024 threads: lane quad = the four
// float4 of a row, 256 rows per pass, 4 rows in flight per thread.  Per-thread sums for the (at most two) groups a
// float4 touches, reduced over the lanes that hold the same float4 slot with xor-shuffles, then over the 16 waves in LDS.
__global__ void __launch_bounds__(1024) gn_fused_rows16_kernel(const float* __restrict__ x, int64_t ldx, int rows, int C,
                                                               int G, float eps, float count_eps,
                                                               const float* __restrict__ w,
                                                               const float* __restrict__ bias, int act,
                                                               float* __restrict__ out, int64_t ldo) {
  __shared__ float part[16][4][4];          // [wave][float4 slot][s_a, q_a, s_b, q_b]
  __shared__ float stat[16][2];             // [group in block][mean, rstd]
  const int cpg = C / G;                    // 2, 4, 8 or 16
  const int gpb = 16 / cpg;                 // groups per block
  const int bpb = G / gpb;                  // blocks per batch element
  const int b = blockIdx.x / bpb, cb = (blockIdx.x - b * bpb) * 16;      // first channel of this block
  const int p = threadIdx.x & 3, rl = threadIdx.x >> 2;                  // float4 slot, row lane (256 per pass)
  const float* xb = x + (int64_t)b * rows * ldx + cb + p * 4;
  float* ob = out + (int64_t)b * rows * ldo + cb + p * 4;
  // components 0..3 of this thread's float4 are channels cb + 4p + k: group (4p + k) / cpg within the block;
  // ga = group of component 0, gb = group of component 3 (cpg = 2: two groups, else one)
  const int ga = (4 * p) / cpg, gb = (4 * p + 3) / cpg;
  const bool two = ga != gb;                                             // cpg == 2: components {0,1} -> ga, {2,3} -> gb
  float sa = 0.f, qa = 0.f, sb = 0.f, qb = 0.f;
  auto acc4 = [&](const float4& v) {
    if (two) {
      sa += v.x + v.y; qa += v.x * v.x + v.y * v.y;
      sb += v.z + v.w; qb += v.z * v.z + v.w * v.w;
    } else {
      sa += (v.x + v.y) + (v.z + v.w); qa += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    }
  };
  int r = rl;
  for (; r + 768 < rows; r += 1024) {
    const float4 v0 = *reinterpret_cast<const float4*>(xb + (int64_t)r * ldx);
    const float4 v1 = *reinterpret_cast<const float4*>(xb + (int64_t)(r + 256) * ldx);
    const float4 v2 = *reinterpret_cast<const float4*>(xb + (int64_t)(r + 512) * ldx);
    const float4 v3 = *reinterpret_cast<const float4*>(xb + (int64_t)(r + 768) * ldx);
    acc4(v0); acc4(v1); acc4(v2); acc4(v3);
  }
  for (; r < rows; r += 256) acc4(*reinterpret_cast<const float4*>(xb + (int64_t)r * ldx));
  // lanes with the same float4 slot: lane bits 2..5
#pragma unroll
  for (int o = 4; o < 64; o <<= 1) {
    sa += __shfl_xor(sa, o); qa += __shfl_xor(qa, o); sb += __shfl_xor(sb, o); qb += __shfl_xor(qb, o);
  }
  const int wv = threadIdx.x >> 6;
  if ((threadIdx.x & 63) < 4) {
    part[wv][p][0] = sa; part[wv][p][1] = qa; part[wv][p][2] = sb; part[wv][p][3] = qb;
  }
  __syncthreads();
  if (threadIdx.x < gpb) {
    // group j of the block: slots whose ga == j contribute (s_a, q_a), slots whose gb == j (and gb != ga) (s_b, q_b)
    const int j = threadIdx.x;
    double S = 0.0, SS = 0.0;
    for (int sl = 0; sl < 4; ++sl) {
      const int a0 = (4 * sl) / cpg, b0 = (4 * sl + 3) / cpg;
      for (int wq = 0; wq < 16; ++wq) {
        if (a0 == j) { S += (double)part[wq][sl][0]; SS += (double)part[wq][sl][1]; }
        if (b0 == j && b0 != a0) { S += (double)part[wq][sl][2]; SS += (double)part[wq][sl][3]; }
      }
    }
    const float cnt = (float)rows * (float)cpg;
    const float inv = 1.0f / (cnt + count_eps);
    const double m = S * (double)inv;
    const double ssd = SS - 2.0 * m * S + (double)cnt * m * m;
    const double var = (ssd > 0 ? ssd : 0) * (double)inv;
    stat[j][0] = (float)m;
    stat[j][1] = (float)(1.0 / sqrt(var + (double)eps));
  }
  __syncthreads();
  const float ma = stat[ga][0], ra = stat[ga][1], mb = stat[gb][0], rb = stat[gb][1];
  const float4 ww = *reinterpret_cast<const float4*>(w + cb + p * 4);
  const float4 bb = *reinterpret_cast<const float4*>(bias + cb + p * 4);
  auto apply = [&](int rr) {
    const float4 v = *reinterpret_cast<const float4*>(xb + (int64_t)rr * ldx);
    float4 y;
    y.x = ofx_apply_act((v.x - ma) * ra * ww.x + bb.x, act);
    y.y = ofx_apply_act((v.y - ma) * ra * ww.y + bb.y, act);
    y.z = ofx_apply_act((v.z - mb) * rb * ww.z + bb.z, act);
    y.w = ofx_apply_act((v.w - mb) * rb * ww.w + bb.w, act);
    *reinterpret_cast<float4*>(ob + (int64_t)rr * ldo) = y;
  };
  for (r = rl; r + 768 < rows; r += 1024) { apply(r); apply(r + 256); apply(r + 512); apply(r + 768); }
  for (; r < rows; r += 256) apply(r);
}

extern "C" int ofx_gn_fused_rows(const float* x, int64_t ldx, int rows_per_batch, int batch_size, int C, int groups,
                                 float eps, float count_eps, const float* w, const float* bias, int act, float* out,
                                 int64_t ldo, void* stream) {
  if (!x || !w || !bias || !out || rows_per_batch < 1 || batch_size < 1 || C < 1 || groups < 1 || C % groups ||
      ldx < C || ldo < C || act < 0 || act > OFX_ACT_GELU)
    return OFX_EINVAL;
  const int cpg = C / groups;
  const bool al16 = ((((uintptr_t)x | (uintptr_t)out | (uintptr_t)w | (uintptr_t)bias) & 15) == 0) && (ldx % 4 == 0) &&
                    (ldo % 4 == 0);
  if (g_gn_rows16 && al16 && (cpg == 2 || cpg == 4 || cpg == 8 || cpg == 16) && groups % (16 / cpg) == 0 &&
      rows_per_batch >= 2048)        // measured: 16^3 grids 33 -> 25 us (C 64), 67 -> 30 us (C 128, B 8); 8^3 grids are faster per group
    gn_fused_rows16_kernel<<<batch_size * (groups / (16 / cpg)), 1024, 0, ofx_stream(stream)>>>(
        x, ldx, rows_per_batch, C, groups, eps, count_eps, w, bias, act, out, ldo);
  else
    gn_fused_rows_kernel<<<batch_size * groups, 256, 0, ofx_stream(stream)>>>(x, ldx, rows_per_batch, C, groups, eps,
                                                                              count_eps, w, bias, act, out, ldo);
  OFX_LAUNCH_CHECK();
  return OFX_OK;
}

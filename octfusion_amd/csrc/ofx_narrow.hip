// GraphConv (reference models/networks/modules.py:194-220) for the two layers of a diffusion U-Net that have almost no
// arithmetic: the INPUT convolution (3 or 8 channels -> model_channels, graph_unet_hr.py:116) and the OUTPUT
// convolution (model_channels -> 3 or 8, graph_unet_hr.py:205-209).  Both were run through the contraction kernels
// until round 4 (input: zero-padded to a 32-channel chunk on the planes kernel, 112 us at depth 6 = 0.15 of the HBM
// roof; output: the register-staged kernel gathering E x 128 floats for 3 output columns, 137 us).  They are gathers:
//
//   * narrow_in_kernel: per 64-row block the 7 x (cin + nt) col_data values of every row go to LDS (one thread per
//     (row, direction) segment walks its CSR edges: 12..32-byte pieces of x, L2 hits), then a lane OWNS 1 or 2 output
//     columns -- its K weights live in registers for the whole block -- and every row is K broadcast LDS reads + K
//     FMAs per lane in exact fp32; the 512-B output row of a wave is one coalesced store.  GroupNorm statistics of the
//     output ride along (per-lane column sums, no cross-lane work) in the two-stage protocol of the MFMA kernels
//     (stats_part[block][cout][2] -> stats_reduce_kernel).  Bound: the N x cout x 4-byte store stream.
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
  const float* tf; int64_t ldt; int nt;
  const float* W; int cout; const float* bias;
  float* out; int64_t ldc;
  const int32_t* bid; float* stats_part; double* stats; int64_t stats_ld;
};

template <int NC, int KP>
__global__ void __launch_bounds__(256) narrow_in_kernel(const NarrowInArgs a) {
  __shared__ __attribute__((aligned(16))) float cd[64][KP];
  __shared__ float red[3][64 * NC * 2];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int64_t row0 = (int64_t)blockIdx.x * 64;
  const int cpd = a.cin + a.nt, K = 7 * cpd;

  // this lane's columns of W: K x NC registers, loaded while the gather below is in flight
  float w[KP][NC];
#pragma unroll
  for (int k = 0; k < KP; ++k) {
#pragma unroll
    for (int j = 0; j < NC; ++j) w[k][j] = k < K ? a.W[(int64_t)k * a.cout + lane * NC + j] : 0.f;
  }
  float bv[NC];
#pragma unroll
  for (int j = 0; j < NC; ++j) bv[j] = a.bias ? a.bias[lane * NC + j] : 0.f;

  // ---- phase 1: col_data of the block's rows -> LDS (zero beyond K and beyond N)
  for (int i = tid; i < 64 * (KP - K); i += 256) cd[i / (KP - K)][K + i % (KP - K)] = 0.f;
  for (int s = tid; s < 64 * 7; s += 256) {
    const int r = s / 7, dir = s - r * 7;
    const int64_t row = row0 + r;
    float acc[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[c] = 0.f;
    float cnt = 1.f;
    if (row < a.N) {
      const int32_t b = a.seg_ptr[row * 7 + dir], e = a.seg_ptr[row * 7 + dir + 1];
      for (int32_t p = b; p < e; ++p) {
        const float* xr = a.x + (int64_t)a.col[p] * a.ldx;
#pragma unroll
        for (int c = 0; c < 8; ++c)
          if (c < a.cin) acc[c] += xr[c];
      }
      cnt = (float)(e - b > 1 ? e - b : 1);
    }
    float* o = &cd[r][dir * cpd];
#pragma unroll
    for (int c = 0; c < 8; ++c)
      if (c < a.cin) o[c] = acc[c] / cnt;
    for (int t = 0; t < a.nt; ++t) o[a.cin + t] = row < a.N ? a.tf[row * a.ldt + dir * a.nt + t] : 0.f;
  }
  // do all rows of the block belong to one batch element?  (decides how the statistics leave the block)
  bool same = true;
  int b0 = 0;
  if (a.stats) {
    b0 = a.bid[row0];
    if (tid < 64 && row0 + tid < a.N) same = a.bid[row0 + tid] == b0;
  }
  const bool uni = __syncthreads_and(same);

  // ---- phase 2: rows wv * 16 .. + 16 of the block; lane -> columns lane * NC .. + NC
  float s_[NC], q_[NC];
#pragma unroll
  for (int j = 0; j < NC; ++j) s_[j] = q_[j] = 0.f;
  for (int i = 0; i < 16; ++i) {
    const int r = wv * 16 + i;
    const int64_t row = row0 + r;
    if (row >= a.N) break;
    float acc[NC];
#pragma unroll
    for (int j = 0; j < NC; ++j) acc[j] = bv[j];
#pragma unroll
    for (int k4 = 0; k4 < KP / 4; ++k4) {
      if (k4 * 4 < K) {
        const float4 c = *reinterpret_cast<const float4*>(&cd[r][k4 * 4]);        // same address in every lane: broadcast
#pragma unroll
        for (int j = 0; j < NC; ++j) {
          acc[j] = fmaf(c.x, w[k4 * 4 + 0][j], acc[j]);
          acc[j] = fmaf(c.y, w[k4 * 4 + 1][j], acc[j]);
          acc[j] = fmaf(c.z, w[k4 * 4 + 2][j], acc[j]);
          acc[j] = fmaf(c.w, w[k4 * 4 + 3][j], acc[j]);
        }
      }
    }
    float* o = a.out + row * a.ldc + lane * NC;
    if (NC == 2) *reinterpret_cast<float2*>(o) = make_float2(acc[0], acc[NC - 1]);
    else o[0] = acc[0];
    if (a.stats) {
      if (uni) {
#pragma unroll
        for (int j = 0; j < NC; ++j) { s_[j] += acc[j]; q_[j] += acc[j] * acc[j]; }
      } else {                                           // a block that holds a batch boundary (a handful per launch)
        double* so = a.stats + ((int64_t)a.bid[row] * a.stats_ld + lane * NC) * 2;
#pragma unroll
        for (int j = 0; j < NC; ++j) {
          unsafeAtomicAdd(so + 2 * j, (double)acc[j]);
          unsafeAtomicAdd(so + 2 * j + 1, (double)acc[j] * (double)acc[j]);
        }
      }
    }
  }
  if (a.stats) {
    // one partial (sum, sum of squares) per block and column: stats_part[block][cout][2] (zeros for a mixed block)
    if (wv > 0) {
#pragma unroll
      for (int j = 0; j < NC; ++j) { red[wv - 1][(lane * NC + j) * 2] = s_[j]; red[wv - 1][(lane * NC + j) * 2 + 1] = q_[j]; }
    }
    __syncthreads();
    if (wv == 0) {
#pragma unroll
      for (int j = 0; j < NC; ++j) {
        float s = s_[j], q = q_[j];
#pragma unroll
        for (int u = 0; u < 3; ++u) { s += red[u][(lane * NC + j) * 2]; q += red[u][(lane * NC + j) * 2 + 1]; }
        *reinterpret_cast<float2*>(a.stats_part + ((int64_t)blockIdx.x * a.cout + lane * NC + j) * 2) = make_float2(s, q);
      }
    }
  }
}

struct NarrowOutArgs {
  const float* P; int64_t ldp; int cout; int64_t N;
  const int32_t* seg_ptr; const int32_t* col;
  const float* tf; int64_t ldt; int nt;
  const float* W; int64_t w_type_row0; int w_dir_stride;      // type rows of direction d start at W row d * w_dir_stride + w_type_row0
  const float* bias; float* out; int64_t ldc;
};

// lane = (row, slot): slots 0..6 = the seven directions, slot 7 = bias.  32 rows per 256-thread block.
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
#pragma unroll
        for (int o = 0; o < CO; ++o)
          if (o < a.cout) acc[o] += pr[o];
      }
      const float cnt = (float)(e - b > 1 ? e - b : 1);
#pragma unroll
      for (int o = 0; o < CO; ++o) acc[o] /= cnt;
      // node-type term of this direction: sum_t type_frac[row, d, t] * W[d, C + t, :]
      for (int t = 0; t < a.nt; ++t) {
        const float f = a.tf[row * a.ldt + d * a.nt + t];
        const float* wr = a.W + ((int64_t)d * a.w_dir_stride + a.w_type_row0 + t) * a.cout;
#pragma unroll
        for (int o = 0; o < CO; ++o)
          if (o < a.cout) acc[o] = fmaf(f, wr[o], acc[o]);
      }
    } else if (a.bias) {
#pragma unroll
      for (int o = 0; o < CO; ++o)
        if (o < a.cout) acc[o] = a.bias[o];
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
                                       const int32_t* col, const float* type_frac, int64_t ldt, int nt, const float* W,
                                       int cout, const float* bias, const int32_t* batch_id, float* out, int64_t ldc,
                                       double* stats, int64_t stats_ld, void* ws, size_t ws_bytes, void* stream) {
  if (!x || !seg_ptr || !col || !W || !out || cin < 1 || cin > 8 || ldx < cin || nt < 0 || (nt > 0 && (!type_frac || ldt < 7 * nt)) ||
      (cout != 64 && cout != 128) || ldc < cout || (ldc & 1) || ((uintptr_t)out & 7) || n_nodes < 0 || 7 * (cin + nt) > 96)
    return OFX_EINVAL;
  if (n_nodes == 0) return OFX_OK;
  const int64_t nblk = ofx_cdiv(n_nodes, 64);
  if (stats && (!batch_id || stats_ld < cout || !ws || ws_bytes < (size_t)nblk * cout * 2 * sizeof(float))) return OFX_EINVAL;
  hipStream_t st = ofx_stream(stream);
  NarrowInArgs a = {x, ldx, cin, n_nodes, seg_ptr, col, type_frac, ldt, nt, W, cout, bias, out, ldc,
                    batch_id, (float*)ws, stats, stats_ld};
  const int K = 7 * (cin + nt);
  if (cout == 128) {
    if (K <= 64) narrow_in_kernel<2, 64><<<(unsigned)nblk, 256, 0, st>>>(a);
    else narrow_in_kernel<2, 96><<<(unsigned)nblk, 256, 0, st>>>(a);
  } else {
    if (K <= 64) narrow_in_kernel<1, 64><<<(unsigned)nblk, 256, 0, st>>>(a);
    else narrow_in_kernel<1, 96><<<(unsigned)nblk, 256, 0, st>>>(a);
  }
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

extern "C" int ofx_graphconv_narrow_out(const float* P, int64_t ldp, int cout, int64_t n_nodes, const int32_t* seg_ptr,
                                        const int32_t* col, const float* type_frac, int64_t ldt, int nt, const float* W,
                                        int C, const float* bias, float* out, int64_t ldc, void* stream) {
  if (!P || !seg_ptr || !col || !W || !out || cout < 1 || cout > 8 || ldp < 7 * cout || nt < 0 ||
      (nt > 0 && (!type_frac || ldt < 7 * nt)) || C < 1 || ldc < cout || n_nodes < 0)
    return OFX_EINVAL;
  if (n_nodes == 0) return OFX_OK;
  NarrowOutArgs a = {P, ldp, cout, n_nodes, seg_ptr, col, type_frac, ldt, nt, W, (int64_t)C, C + nt, bias, out, ldc};
  hipStream_t st = ofx_stream(stream);
  const unsigned nblk = (unsigned)ofx_cdiv(n_nodes, 32);
  if (cout <= 4) narrow_out_kernel<4><<<nblk, 256, 0, st>>>(a);
  else narrow_out_kernel<8><<<nblk, 256, 0, st>>>(a);
  OFX_LAUNCH_CHECK();
  return OFX_OK;
}

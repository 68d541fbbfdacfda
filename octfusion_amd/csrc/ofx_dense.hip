// libofx: the dense-grid side of the step (reference graph_unet_lr.py + modules.py:63-95,
// 474-547): 3x3x3 convolutions and self-attention on full octree layers (16^3 / 8^3 / 4^3),
// kept in the SAME node-row layout as the sparse side (row = b*8^d + morton(x,y,z)), so
// octree2voxel / gather-back are identities and the convolutions reuse the fused
// gather-GEMM kernel with 27-tap neighbour tables.
#include "ofx_common.h"

// mode 0: stride 1, in = out depth d         (nn.Conv3d k3 p1)
// mode 1: stride 2, out depth d, in depth d+1 (ConvDownsample, modules.py:81-95): in = 2*o + tap - 1
// mode 2: nearest-upsample x2 then k3 p1, out depth d, in depth d-1 (ConvUpsample, :63-78):
//         in = (o + tap - 1) >> 1 when o + tap - 1 is inside the fine grid
__global__ void grid_table_kernel(int mode, int d_out, int B, int32_t pad, int32_t* __restrict__ nbr) {
  const int S = 1 << d_out;
  const int64_t per = (int64_t)S * S * S, total = per * B * 27;
  const int d_in = mode == 0 ? d_out : (mode == 1 ? d_out + 1 : d_out - 1);
  const int Sin = 1 << d_in;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = t / 27;
    const int tap = (int)(t - row * 27);
    const int64_t b = row / per;
    int x, y, z, bb;
    ofx_key2xyz(row - b * per, x, y, z, bb);
    const int tx = tap / 9 - 1, ty = (tap / 3) % 3 - 1, tz = tap % 3 - 1;
    int ix, iy, iz;
    bool ok;
    if (mode == 0) {
      ix = x + tx; iy = y + ty; iz = z + tz;
      ok = ix >= 0 && iy >= 0 && iz >= 0 && ix < S && iy < S && iz < S;
    } else if (mode == 1) {
      ix = 2 * x + tx; iy = 2 * y + ty; iz = 2 * z + tz;
      ok = ix >= 0 && iy >= 0 && iz >= 0 && ix < Sin && iy < Sin && iz < Sin;
    } else {
      const int fx = x + tx, fy = y + ty, fz = z + tz;
      ok = fx >= 0 && fy >= 0 && fz >= 0 && fx < S && fy < S && fz < S;
      ix = fx >> 1; iy = fy >> 1; iz = fz >> 1;
    }
    nbr[t] = ok ? (int32_t)(b * ((int64_t)Sin * Sin * Sin) + (int64_t)ofx_xyz2morton(ix, iy, iz)) : pad;
  }
}

extern "C" int ofx_grid_conv_table(int mode, int depth_out, int batch_size, int32_t pad, int32_t* nbr27, void* stream) {
  if (mode < 0 || mode > 2 || depth_out < 0 || depth_out > 8 || batch_size < 1 || !nbr27) return OFX_EINVAL;
  if (mode == 2 && depth_out < 1) return OFX_EINVAL;
  const int64_t total = (1ll << (3 * depth_out)) * batch_size * 27;
  grid_table_kernel<<<ofx_grid(total, 256), 256, 0, ofx_stream(stream)>>>(mode, depth_out, batch_size, pad, nbr27);
  OFX_LAUNCH_CHECK();
  return OFX_OK;
}

// ---------------------------------------------------------------------------------
// QKVAttention (modules.py:538-547) over the T tokens of one batch element, per head.
// qkv row layout [rows, 3*C]: channel = head*3*ch + {q: 0..ch, k: ch..2ch, v: 2ch..3ch}
// (the reference's reshape(b*heads, 3*ch, T), modules.py:531,540-541); out [rows, C] with
// channel = head*ch + c.  scale ch^-1/4 on q and on k; softmax in fp32 over the keys.
// One thread = one query; K/V tiles of 64 keys staged in LDS (broadcast reads); online
// softmax across tiles.  Tiny work (<= 512 tokens): latency-, not throughput-, critical.
template <int CH>
__global__ void __launch_bounds__(128) attention_kernel(const float* __restrict__ qkv, int64_t ldq, int T, int heads, int ch,
                                                        float* __restrict__ out, int64_t ldo) {
  constexpr int KT = 64;
  __shared__ __attribute__((aligned(16))) float Ks[KT * CH];
  __shared__ __attribute__((aligned(16))) float Vs[KT * CH];
  const int bh = blockIdx.x;
  const int b = bh / heads, hd = bh - b * heads;
  const int t = blockIdx.y * blockDim.x + threadIdx.x;
  const bool active = t < T;
  const int64_t row0 = (int64_t)b * T;
  const float scale = 1.f / sqrtf(sqrtf((float)ch));
  float q[CH], o[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) { q[c] = 0.f; o[c] = 0.f; }
  if (active) {
    const float* qp = qkv + (row0 + t) * ldq + (int64_t)hd * 3 * ch;
#pragma unroll
    for (int c = 0; c < CH; ++c)
      if (c < ch) q[c] = qp[c] * scale;
  }
  float mx = -INFINITY, den = 0.f;
  for (int s0 = 0; s0 < T; s0 += KT) {
    const int ns = T - s0 < KT ? T - s0 : KT;
    __syncthreads();
    for (int i = threadIdx.x; i < ns * ch; i += blockDim.x) {
      const int s = i / ch, c = i - s * ch;
      const float* kp = qkv + (row0 + s0 + s) * ldq + (int64_t)hd * 3 * ch + ch;
      Ks[s * CH + c] = kp[c] * scale;
      Vs[s * CH + c] = kp[ch + c];
    }
    __syncthreads();
    if (active) {
      for (int s = 0; s < ns; ++s) {
        float w = 0.f;
#pragma unroll
        for (int c = 0; c < CH; ++c)
          if (c < ch) w += q[c] * Ks[s * CH + c];
        if (w > mx) {
          const float f = __expf(mx - w);
          den *= f;
#pragma unroll
          for (int c = 0; c < CH; ++c) o[c] *= f;
          mx = w;
        }
        const float p = __expf(w - mx);
        den += p;
#pragma unroll
        for (int c = 0; c < CH; ++c)
          if (c < ch) o[c] += p * Vs[s * CH + c];
      }
    }
  }
  if (active) {
    const float inv = 1.f / den;
    float* op = out + (row0 + t) * ldo + (int64_t)hd * ch;
#pragma unroll
    for (int c = 0; c < CH; ++c)
      if (c < ch) op[c] = o[c] * inv;
  }
}

extern "C" int ofx_attention(const float* qkv, int64_t ldq, int batch_size, int T, int heads, int ch, float* out,
                             int64_t ldo, void* stream) {
  if (!qkv || !out || batch_size < 1 || T < 1 || heads < 1 || ch < 1 || ch > 128 || ldq < 3 * (int64_t)heads * ch ||
      ldo < (int64_t)heads * ch)
    return OFX_EINVAL;
  dim3 grid((unsigned)(batch_size * heads), (unsigned)ofx_cdiv(T, 128));
  hipStream_t st = ofx_stream(stream);
  if (ch <= 32) attention_kernel<32><<<grid, 128, 0, st>>>(qkv, ldq, T, heads, ch, out, ldo);
  else if (ch <= 64) attention_kernel<64><<<grid, 128, 0, st>>>(qkv, ldq, T, heads, ch, out, ldo);
  else attention_kernel<128><<<grid, 128, 0, st>>>(qkv, ldq, T, heads, ch, out, ldo);
  OFX_LAUNCH_CHECK();
  return OFX_OK;
}

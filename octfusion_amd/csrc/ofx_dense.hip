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
//
// fp32 MFMA (v_mfma_f32_32x32x2_f32), one wave = 32 queries, block = 4 waves = 128 queries,
// K and V of the (batch, head) staged once in LDS.  The score tile is computed TRANSPOSED
// (S^T = K Q^T): in the 32x32 C/D layout a lane then owns ONE query (its column) and 16 of
// the tile's 32 keys, so the softmax reductions are in-lane plus one half-wave exchange, and
// the probabilities feed the P.V MFMA straight from registers: the k-order of that MFMA is
// permuted to the order in which the lanes already hold the keys (key(r,h) = (r&3) + 8(r>>2)
// + 4h), which only changes which V row each lane reads.  Two passes over the keys (pass 1:
// running max / sum per query; pass 2: P = exp(S - max)/sum, O += P V) -- no accumulator
// rescaling, deterministic.
typedef float f32x16d __attribute__((ext_vector_type(16)));

//
// SPLIT (round 5, T >= 256): the four waves of a block share ONE tile of 32 queries and split the KEYS four ways
// (flash-decoding style) instead of owning 32 queries each over all keys.  A (batch, head) then is T / 32 blocks instead
// of T / 128 -- 512 blocks at batch 8 x 4 heads x 512 tokens, 64 at batch 1 (16 before: the 8^3 level of a one-shape step
// ran 65 us on 16 CUs) -- and a wave's serial chain of fp32 MFMAs is four times shorter.  Pass 1 ends with a cross-wave
// combine of (max, sum) through LDS; pass 2's partial outputs are added in wave order (deterministic) through the K
// stage, which is dead by then.
template <int CH, bool SPLIT>   // CH = head channels rounded up to 32 (32, 64, 128)
__global__ void __launch_bounds__(256, 1) attention_mfma_kernel(const float* __restrict__ qkv, int64_t ldq, int T,
                                                                int heads, int ch, float* __restrict__ out,
                                                                int64_t ldo) {
  __shared__ float cmb[4][32][2];
  constexpr int KLD = CH + 4;                     // LDS row pitch (floats): conflict-free b128 reads over rows
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int Tp = (T + 31) & ~31;                  // keys padded to whole 32-key tiles
  float* Ks = smem;                               // [Tp][KLD]
  float* Vs = smem + (size_t)Tp * KLD;            // [Tp][KLD]
  const int bh = blockIdx.x;
  const int b = bh / heads, hd = bh - b * heads;
  const int64_t row0 = (int64_t)b * T;
  const float scale = 1.f / sqrtf(sqrtf((float)ch));
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int l31 = lane & 31, h = lane >> 5;

  // stage K (pre-scaled) and V; zero the channel / key padding
  const float* kbase = qkv + row0 * ldq + (int64_t)hd * 3 * ch + ch;
  if ((ch & 3) == 0 && (ldq & 3) == 0 && (((uintptr_t)qkv) & 15) == 0) {
    // 16-B pieces, four in flight per thread (the scalar loop below was a chain of ~128 dependent 4-B round trips per
    // thread for T = 512: a third of the kernel's time at the 8^3 level)
    constexpr int C4 = CH / 4;
    const int total = Tp * C4;
    for (int i0 = threadIdx.x; i0 < total; i0 += 4 * blockDim.x) {
      float4 kq[4], vq[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + u * blockDim.x;
        const int s = i / C4, c = (i - s * C4) * 4;
        kq[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        vq[u] = kq[u];
        if (i < total && s < T && c < ch) {
          const float* p = kbase + (int64_t)s * ldq + c;
          kq[u] = *reinterpret_cast<const float4*>(p);
          vq[u] = *reinterpret_cast<const float4*>(p + ch);
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + u * blockDim.x;
        if (i >= total) continue;
        const int s = i / C4, c = (i - s * C4) * 4;
        *reinterpret_cast<float4*>(Ks + s * KLD + c) = make_float4(kq[u].x * scale, kq[u].y * scale, kq[u].z * scale, kq[u].w * scale);
        *reinterpret_cast<float4*>(Vs + s * KLD + c) = vq[u];
      }
    }
  } else {
    for (int i = threadIdx.x; i < Tp * CH; i += blockDim.x) {
      const int s = i / CH, c = i - s * CH;
      float kv = 0.f, vv = 0.f;
      if (s < T && c < ch) {
        const float* p = kbase + (int64_t)s * ldq + c;
        kv = p[0] * scale;
        vv = p[ch];
      }
      Ks[s * KLD + c] = kv;
      Vs[s * KLD + c] = vv;
    }
  }
  __syncthreads();

  const int q0 = SPLIT ? blockIdx.y * 32 : blockIdx.y * 128 + wid * 32;      // this wave's 32 queries
  if (q0 >= T) return;                             // (SPLIT: the same for every wave of the block)
  const int qi = q0 + l31;                         // this lane's query (column of S^T)
  const bool qok = qi < T;
  // B operand of S^T = K Q^T: lane (query, h) holds Q[query][h*CH/2 + s], s < CH/2 (k-order permuted)
  float qreg[CH / 2];
  {
    const float* qp = qkv + (row0 + (qok ? qi : T - 1)) * ldq + (int64_t)hd * 3 * ch;
#pragma unroll
    for (int s = 0; s < CH / 2; ++s) {
      const int c = h * (CH / 2) + s;
      qreg[s] = (c < ch) ? qp[c] * scale : 0.f;
    }
  }
  const int ntile = Tp / 32;

  auto score_tile = [&](int kt, f32x16d& st) {
#pragma unroll
    for (int r = 0; r < 16; ++r) st[r] = 0.f;
    const float* kr = Ks + (kt * 32 + l31) * KLD + h * (CH / 2);   // A operand: K[key = l31][h*CH/2 + s]
#pragma unroll
    for (int s4 = 0; s4 < CH / 8; ++s4) {
      const float4 kv = *reinterpret_cast<const float4*>(kr + s4 * 4);
      st = __builtin_amdgcn_mfma_f32_32x32x2f32(kv.x, qreg[s4 * 4 + 0], st, 0, 0, 0);
      st = __builtin_amdgcn_mfma_f32_32x32x2f32(kv.y, qreg[s4 * 4 + 1], st, 0, 0, 0);
      st = __builtin_amdgcn_mfma_f32_32x32x2f32(kv.z, qreg[s4 * 4 + 2], st, 0, 0, 0);
      st = __builtin_amdgcn_mfma_f32_32x32x2f32(kv.w, qreg[s4 * 4 + 3], st, 0, 0, 0);
    }
    // mask padded keys: st[r] is key kt*32 + (r&3) + 8*(r>>2) + 4*h of this lane's query
    if (kt * 32 + 32 > T) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h >= T) st[r] = -INFINITY;
    }
  };

  const int kt0 = SPLIT ? (wid * ntile) / 4 : 0, kt1 = SPLIT ? ((wid + 1) * ntile) / 4 : ntile;     // this wave's key tiles

  // pass 1: per-query running max and sum of exp
  float mx = -INFINITY, den = 0.f;
  for (int kt = kt0; kt < kt1; ++kt) {
    f32x16d st;
    score_tile(kt, st);
    float tm = st[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) tm = fmaxf(tm, st[r]);
    tm = fmaxf(tm, __shfl_xor(tm, 32, 64));
    const float mn = fmaxf(mx, tm);
    float ps = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) ps += __expf(st[r] - mn);
    ps += __shfl_xor(ps, 32, 64);
    den = den * __expf(mx - mn) + ps;
    mx = mn;
  }
  if (SPLIT) {                                     // (max, sum) of the four key ranges -> the softmax's own
    if (h == 0) { cmb[wid][l31][0] = mx; cmb[wid][l31][1] = den; }
    __syncthreads();
    float gm = -INFINITY;
#pragma unroll
    for (int w = 0; w < 4; ++w) gm = fmaxf(gm, cmb[w][l31][0]);
    float gd = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float mw = cmb[w][l31][0];
      gd += mw == -INFINITY ? 0.f : cmb[w][l31][1] * __expf(mw - gm);
    }
    mx = gm; den = gd;
  }
  const float inv = 1.f / den;

  // pass 2: O[query, c] += P[query, key] V[key, c]; A = P from registers, B = V rows key(r, h)
  f32x16d oacc[CH / 32];
#pragma unroll
  for (int j = 0; j < CH / 32; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[j][r] = 0.f;
  for (int kt = kt0; kt < kt1; ++kt) {
    f32x16d st;
    score_tile(kt, st);
#pragma unroll
    for (int r = 0; r < 16; ++r) st[r] = __expf(st[r] - mx) * inv;
    const float* vr = Vs + (kt * 32 + 4 * h) * KLD + l31;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float* vrow = vr + ((r & 3) + 8 * (r >> 2)) * KLD;
#pragma unroll
      for (int j = 0; j < CH / 32; ++j)
        oacc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(st[r], vrow[j * 32], oacc[j], 0, 0, 0);
    }
  }
  // O layout: lane (c = l31, h), reg r -> query q0 + (r&3) + 8(r>>2) + 4h
  if (SPLIT) {
    __syncthreads();                               // every wave is done with Ks / Vs: the K stage becomes osum[4][32][CH]
    float* osum = Ks;
#pragma unroll
    for (int j = 0; j < CH / 32; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        osum[(wid * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * CH + j * 32 + l31] = oacc[j][r];
    __syncthreads();
    for (int i = threadIdx.x; i < 32 * CH; i += 256) {
      const int q = i / CH, c = i - q * CH;
      if (q0 + q < T && c < ch)
        out[(row0 + q0 + q) * ldo + (int64_t)hd * ch + c] = (osum[i] + osum[32 * CH + i]) + (osum[64 * CH + i] + osum[96 * CH + i]);
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < CH / 32; ++j) {
    const int c = j * 32 + l31;
    if (c >= ch) continue;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int q = q0 + (r & 3) + 8 * (r >> 2) + 4 * h;
      if (q < T) out[(row0 + q) * ldo + (int64_t)hd * ch + c] = oacc[j][r];
    }
  }
}

static int g_attn_split = 1;                     // A/B: ofx_set_attention_split(0) keeps every sequence on the one-wave-per-32-queries kernel
extern "C" int ofx_set_attention_split(int on) { g_attn_split = on ? 1 : 0; return OFX_OK; }

template <int CH>
static int launch_attention(const float* qkv, int64_t ldq, int B, int T, int heads, int ch, float* out, int64_t ldo,
                            hipStream_t st) {
  const int Tp = (T + 31) & ~31;
  const size_t lds = (size_t)2 * Tp * (CH + 4) * sizeof(float);
  if (lds > 160 * 1024 - 2048) return OFX_EINVAL;      // (+ 1 KB of static LDS: the cross-wave combine)
  static bool attr_set[OFX_MAX_DEVICES] = {}, attr_set_split[OFX_MAX_DEVICES] = {};
  if (T >= 256 && g_attn_split) {                 // long sequences: keys split over the waves of a block (see the kernel)
    if (!ofx_raise_lds_limit(reinterpret_cast<const void*>(&attention_mfma_kernel<CH, true>), 160 * 1024 - 2048, attr_set_split))
      return OFX_ELAUNCH;
    dim3 grid((unsigned)(B * heads), (unsigned)ofx_cdiv(T, 32));
    attention_mfma_kernel<CH, true><<<grid, 256, lds, st>>>(qkv, ldq, T, heads, ch, out, ldo);
    return OFX_OK;
  }
  if (!ofx_raise_lds_limit(reinterpret_cast<const void*>(&attention_mfma_kernel<CH, false>), 160 * 1024 - 2048, attr_set))
    return OFX_ELAUNCH;
  dim3 grid((unsigned)(B * heads), (unsigned)ofx_cdiv(T, 128));
  attention_mfma_kernel<CH, false><<<grid, 256, lds, st>>>(qkv, ldq, T, heads, ch, out, ldo);
  return OFX_OK;
}

extern "C" int ofx_attention(const float* qkv, int64_t ldq, int batch_size, int T, int heads, int ch, float* out,
                             int64_t ldo, void* stream) {
  if (!qkv || !out || batch_size < 1 || T < 1 || heads < 1 || ch < 1 || ch > 128 || ldq < 3 * (int64_t)heads * ch ||
      ldo < (int64_t)heads * ch)
    return OFX_EINVAL;
  hipStream_t st = ofx_stream(stream);
  int rc;
  if (ch <= 32) rc = launch_attention<32>(qkv, ldq, batch_size, T, heads, ch, out, ldo, st);
  else if (ch <= 64) rc = launch_attention<64>(qkv, ldq, batch_size, T, heads, ch, out, ldo, st);
  else rc = launch_attention<128>(qkv, ldq, batch_size, T, heads, ch, out, ldo, st);
  if (rc) return rc;
  OFX_LAUNCH_CHECK();
  return OFX_OK;
}

// ---------------------------------------------------------------------------------
// Backward of QKVAttention (training path; autograd of modules.py:538-547).  Per (batch element, head), with
// S = ch^-1/2 * Q K^T, P = softmax_rows(S), O = P V and g = dL/dO:
//   dP = g V^T,  D_i = sum_j P_ij dP_ij,  dS = P * (dP - D),  dQ = ch^-1/2 dS K,  dK = ch^-1/2 dS^T Q,  dV = P^T g.
// The problem is tiny (T <= 512 tokens, 4 heads, batch 8): two plain fp32 kernels, one wave per query row
// (row max / sum / D, dQ) and one wave per key row (dK, dV, recomputing P from the saved row statistics);
// deterministic, no atomics.  Same row layout as the forward kernel.
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

__global__ void __launch_bounds__(256) attention_bwd_q_kernel(const float* __restrict__ qkv, int64_t ldq,
                                                              const float* __restrict__ dout, int64_t ldo, int T,
                                                              int heads, int ch, float* __restrict__ rowstat,
                                                              float* __restrict__ dqkv, int64_t ldd) {
  extern __shared__ float lds[];                      // [4 waves][T]
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int bh = blockIdx.y, b = bh / heads, hd = bh - b * heads;
  const int i = blockIdx.x * 4 + wid;
  if (i >= T) return;
  float* ds = lds + wid * T;
  const float sc = rsqrtf((float)ch);
  const int64_t row0 = (int64_t)b * T;
  const float* qi = qkv + (row0 + i) * ldq + (int64_t)hd * 3 * ch;
  const float* gi = dout + (row0 + i) * ldo + (int64_t)hd * ch;
  const float* kb = qkv + row0 * ldq + (int64_t)hd * 3 * ch + ch;
  const float* vb = kb + ch;
  const bool vec = (ch & 3) == 0 && (ldq & 3) == 0 && (ldo & 3) == 0 && ((((uintptr_t)qkv) | ((uintptr_t)dout)) & 15) == 0;
  float s[8], dp[8];                                   // T <= 512: up to 8 keys per lane
  float m = -3.0e38f;
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const int j = lane + 64 * t;
    s[t] = -3.0e38f; dp[t] = 0.f;
    if (j < T) {
      float a = 0.f, d = 0.f;
      const float* kj = kb + (int64_t)j * ldq;
      const float* vj = vb + (int64_t)j * ldq;
      if (vec) {
        for (int c = 0; c < ch; c += 4) {
          const float4 q4 = *reinterpret_cast<const float4*>(qi + c), g4 = *reinterpret_cast<const float4*>(gi + c);
          const float4 k4 = *reinterpret_cast<const float4*>(kj + c), v4 = *reinterpret_cast<const float4*>(vj + c);
          a += q4.x * k4.x + q4.y * k4.y + q4.z * k4.z + q4.w * k4.w;
          d += g4.x * v4.x + g4.y * v4.y + g4.z * v4.z + g4.w * v4.w;
        }
      } else {
        for (int c = 0; c < ch; ++c) { a += qi[c] * kj[c]; d += gi[c] * vj[c]; }
      }
      s[t] = a * sc; dp[t] = d;
      m = fmaxf(m, s[t]);
    }
  }
  m = wave_max(m);
  float l = 0.f;
#pragma unroll
  for (int t = 0; t < 8; ++t) { const int j = lane + 64 * t; if (j < T) { s[t] = __expf(s[t] - m); l += s[t]; } }
  l = wave_sum(l);
  const float inv = 1.f / l;
  float D = 0.f;
#pragma unroll
  for (int t = 0; t < 8; ++t) { const int j = lane + 64 * t; if (j < T) { s[t] *= inv; D += s[t] * dp[t]; } }
  D = wave_sum(D);
#pragma unroll
  for (int t = 0; t < 8; ++t) { const int j = lane + 64 * t; if (j < T) ds[j] = s[t] * (dp[t] - D); }
  if (lane == 0) {
    float* rs = rowstat + ((int64_t)bh * T + i) * 3;
    rs[0] = m; rs[1] = inv; rs[2] = D;
  }
  __builtin_amdgcn_wave_barrier();
  // dq_i[c] = sc * sum_j dS_ij k_j[c]: lanes over channels (coalesced key rows)
  for (int c = lane; c < ch; c += 64) {
    float a = 0.f;
    for (int j = 0; j < T; ++j) a += ds[j] * kb[(int64_t)j * ldq + c];
    dqkv[(row0 + i) * ldd + (int64_t)hd * 3 * ch + c] = a * sc;
  }
}

__global__ void __launch_bounds__(256) attention_bwd_kv_kernel(const float* __restrict__ qkv, int64_t ldq,
                                                               const float* __restrict__ dout, int64_t ldo, int T,
                                                               int heads, int ch, const float* __restrict__ rowstat,
                                                               float* __restrict__ dqkv, int64_t ldd) {
  extern __shared__ float lds[];                      // [4 waves][2][T]: P column, dS column
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int bh = blockIdx.y, b = bh / heads, hd = bh - b * heads;
  const int j = blockIdx.x * 4 + wid;
  if (j >= T) return;
  float* pc = lds + wid * 2 * T;
  float* dc = pc + T;
  const float sc = rsqrtf((float)ch);
  const int64_t row0 = (int64_t)b * T;
  const float* qb = qkv + row0 * ldq + (int64_t)hd * 3 * ch;
  const float* kj = qb + (int64_t)j * ldq + ch;
  const float* vj = kj + ch;
  const float* gb = dout + row0 * ldo + (int64_t)hd * ch;
  const float* rs = rowstat + (int64_t)bh * T * 3;
  const bool vec = (ch & 3) == 0 && (ldq & 3) == 0 && (ldo & 3) == 0 && ((((uintptr_t)qkv) | ((uintptr_t)dout)) & 15) == 0;
  for (int i = lane; i < T; i += 64) {
    float a = 0.f, d = 0.f;
    const float* qr = qb + (int64_t)i * ldq;
    const float* gr = gb + (int64_t)i * ldo;
    if (vec) {
      for (int c = 0; c < ch; c += 4) {
        const float4 q4 = *reinterpret_cast<const float4*>(qr + c), g4 = *reinterpret_cast<const float4*>(gr + c);
        const float4 k4 = *reinterpret_cast<const float4*>(kj + c), v4 = *reinterpret_cast<const float4*>(vj + c);
        a += q4.x * k4.x + q4.y * k4.y + q4.z * k4.z + q4.w * k4.w;
        d += g4.x * v4.x + g4.y * v4.y + g4.z * v4.z + g4.w * v4.w;
      }
    } else {
      for (int c = 0; c < ch; ++c) { a += qr[c] * kj[c]; d += gr[c] * vj[c]; }
    }
    const float p = __expf(a * sc - rs[i * 3]) * rs[i * 3 + 1];
    pc[i] = p;
    dc[i] = p * (d - rs[i * 3 + 2]);
  }
  __builtin_amdgcn_wave_barrier();
  for (int c = lane; c < ch; c += 64) {
    float dk = 0.f, dv = 0.f;
    for (int i = 0; i < T; ++i) { dk += dc[i] * qb[(int64_t)i * ldq + c]; dv += pc[i] * gb[(int64_t)i * ldo + c]; }
    float* o = dqkv + (row0 + j) * ldd + (int64_t)hd * 3 * ch;
    o[ch + c] = dk * sc;
    o[2 * ch + c] = dv;
  }
}

extern "C" int ofx_attention_bwd(const float* qkv, int64_t ldq, const float* dout, int64_t ldo, int batch_size, int T,
                                 int heads, int ch, float* rowstat, float* dqkv, int64_t ldd, void* stream) {
  if (!qkv || !dout || !rowstat || !dqkv || batch_size < 1 || T < 1 || T > 512 || heads < 1 || ch < 1 ||
      ldq < 3 * (int64_t)heads * ch || ldd < 3 * (int64_t)heads * ch || ldo < (int64_t)heads * ch)
    return OFX_EINVAL;
  hipStream_t st = ofx_stream(stream);
  const dim3 grid((unsigned)((T + 3) / 4), (unsigned)(batch_size * heads));
  attention_bwd_q_kernel<<<grid, 256, 4 * T * sizeof(float), st>>>(qkv, ldq, dout, ldo, T, heads, ch, rowstat, dqkv, ldd);
  attention_bwd_kv_kernel<<<grid, 256, 8 * T * sizeof(float), st>>>(qkv, ldq, dout, ldo, T, heads, ch, rowstat, dqkv, ldd);
  OFX_LAUNCH_CHECK();
  return OFX_OK;
}

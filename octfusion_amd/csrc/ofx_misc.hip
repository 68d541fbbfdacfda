// libofx: glue kernels (row permutations for pool/unpool, embeddings, DDIM updates),
// status strings, device check.  All HBM-bound elementwise work.
#include "ofx_gemm_common.h"     // (ofx_common.h + the 16-bit operand-pair helpers: ofx_rows_copy_planes)

extern "C" int ofx_version(void) { return 1; }

extern "C" const char* ofx_status_string(int status) {
  switch (status) {
    case OFX_OK: return "ok";
    case OFX_EINVAL: return "invalid argument";
    case OFX_ELAUNCH: return "HIP launch/runtime error";
    case OFX_ENODEV: return "no gfx950 device";
    default: return "unknown status";
  }
}

extern "C" int ofx_device_check(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n < 1) return OFX_ENODEV;
  int cur = 0;
  if (hipGetDevice(&cur) != hipSuccess) return OFX_ENODEV;          // the calling process's device (one rank per GPU)
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, cur) != hipSuccess) return OFX_ENODEV;
  const char* a = p.gcnArchName;
  for (int i = 0; a[i]; ++i)
    if (a[i] == 'g' && a[i + 1] == 'f' && a[i + 2] == 'x' && a[i + 3] == '9' && a[i + 4] == '5' && a[i + 5] == '0')
      return OFX_OK;
  return OFX_ENODEV;
}

static uint32_t* g_range_words[OFX_MAX_DEVICES] = {};
extern "C" int ofx_set_range_words(uint32_t* words) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= OFX_MAX_DEVICES) return OFX_ENODEV;
  if (words && ((uintptr_t)words & 3)) return OFX_EINVAL;
  g_range_words[dev] = words;
  return OFX_OK;
}
uint32_t* ofx_range_words() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= OFX_MAX_DEVICES) return nullptr;
  return g_range_words[dev];
}

// dst[dmap(i), :] = src[smap(i), :]
__global__ void __launch_bounds__(256) rows_copy_v4(const float* __restrict__ src, int64_t lds, const int32_t* __restrict__ smap,
                                                    float* __restrict__ dst, int64_t ldd, const int32_t* __restrict__ dmap,
                                                    int64_t n, int C4) {
  const int64_t total = n * C4;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = t / C4;
    const int c = (int)(t - i * C4) * 4;
    const int64_t sr = smap ? (int64_t)smap[i] : i;
    const int64_t dr = dmap ? (int64_t)dmap[i] : i;
    if (sr < 0 || dr < 0) continue;
    *reinterpret_cast<float4*>(dst + dr * ldd + c) = *reinterpret_cast<const float4*>(src + sr * lds + c);
  }
}
__global__ void __launch_bounds__(256) rows_copy_s(const float* __restrict__ src, int64_t lds, const int32_t* __restrict__ smap,
                                                   float* __restrict__ dst, int64_t ldd, const int32_t* __restrict__ dmap,
                                                   int64_t n, int C) {
  const int64_t total = n * C;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = t / C;
    const int c = (int)(t - i * C);
    const int64_t sr = smap ? (int64_t)smap[i] : i;
    const int64_t dr = dmap ? (int64_t)dmap[i] : i;
    if (sr < 0 || dr < 0) continue;
    dst[dr * ldd + c] = src[sr * lds + c];
  }
}

// the same copy writing the destination rows as hi / lo pair planes (the consumer is the planes GraphConv): dst is
// 128-B aligned with a pitch of whole 128-B lines, C % 32 == 0
__global__ void __launch_bounds__(256) rows_copy_planes_v4(const float* __restrict__ src, int64_t lds,
                                                           const int32_t* __restrict__ smap, float* __restrict__ dst,
                                                           int64_t ldd, const int32_t* __restrict__ dmap, int64_t n, int C4,
                                                           int mode) {
  const int64_t total = n * C4;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = t / C4;
    const int c = (int)(t - i * C4) * 4;
    const int64_t sr = smap ? (int64_t)smap[i] : i;
    const int64_t dr = dmap ? (int64_t)dmap[i] : i;
    if (sr < 0 || dr < 0) continue;
    ofx_store_planes4(dst, dr * ldd + c, *reinterpret_cast<const float4*>(src + sr * lds + c), mode);
  }
}
extern "C" int ofx_rows_copy_planes(const float* src, int64_t lds, const int32_t* smap, float* dst, int64_t ldd,
                                    const int32_t* dmap, int64_t n, int C, int mode, void* stream) {
  if (n < 0 || C < 32 || (C & 31) || (mode != 2 && mode != 3) || (n > 0 && (!src || !dst)) || lds < C || ldd < C ||
      (lds & 3) || (ldd & 31) || ((uintptr_t)src & 15) || ((uintptr_t)dst & 127))
    return OFX_EINVAL;
  if (n == 0) return OFX_OK;
  rows_copy_planes_v4<<<ofx_grid(n * (C / 4), 256), 256, 0, ofx_stream(stream)>>>(src, lds, smap, dst, ldd, dmap, n,
                                                                                  C / 4, mode);
  OFX_LAUNCH_CHECK();
  return OFX_OK;
}

extern "C" int ofx_rows_copy(const float* src, int64_t lds, const int32_t* smap, float* dst, int64_t ldd,
                             const int32_t* dmap, int64_t n, int C, void* stream) {
  if (n < 0 || C < 1 || (n > 0 && (!src || !dst)) || lds < C || ldd < C) return OFX_EINVAL;
  if (n == 0) return OFX_OK;
  hipStream_t st = ofx_stream(stream);
  const bool vec = (C % 4 == 0) && (lds % 4 == 0) && (ldd % 4 == 0) && (((uintptr_t)src & 15) == 0) &&
                   (((uintptr_t)dst & 15) == 0);
  if (vec)
    rows_copy_v4<<<ofx_grid(n * (C / 4), 256), 256, 0, st>>>(src, lds, smap, dst, ldd, dmap, n, C / 4);
  else
    rows_copy_s<<<ofx_grid(n * C, 256), 256, 0, st>>>(src, lds, smap, dst, ldd, dmap, n, C);
  OFX_LAUNCH_CHECK();
  return OFX_OK;
}

__global__ void timestep_embedding_kernel(const float* __restrict__ t, int B, int dim, float max_period,
                                          float* __restrict__ out) {
  const int half = dim / 2;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < B * dim; i += gridDim.x * blockDim.x) {
    const int b = i / dim, j = i - b * dim;
    float v = 0.f;
    if (j < 2 * half) {
      const int k = j < half ? j : j - half;
      const float f = expf(-logf(max_period) * (float)k / (float)half);
      const float a = t[b] * f;
      v = j < half ? cosf(a) : sinf(a);
    }
    out[i] = v;
  }
}
extern "C" int ofx_timestep_embedding(const float* t, int batch_size, int dim, float max_period, float* out,
                                      void* stream) {
  if (!t || !out || batch_size < 1 || dim < 2) return OFX_EINVAL;
  timestep_embedding_kernel<<<ofx_grid((int64_t)batch_size * dim, 256), 256, 0, ofx_stream(stream)>>>(
      t, batch_size, dim, max_period, out);
  OFX_LAUNCH_CHECK();
  return OFX_OK;
}

// LearnedSinusoidalPosEmb (modules.py:550-563): out[b] = [t_b, sin(2 pi t_b w), cos(2 pi t_b w)], w [half]
__global__ void learned_sinusoid_kernel(const float* __restrict__ t, const float* __restrict__ w, int B, int half,
                                        float* __restrict__ out) {
  const int dim = 2 * half + 1;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < B * dim; i += gridDim.x * blockDim.x) {
    const int b = i / dim, j = i - b * dim;
    const float x = t[b];
    float v = x;
    if (j > 0) {
      const float f = x * w[(j - 1) % half] * 2.f * 3.14159265358979323846f;      // (x * w) * 2 * pi, as the reference orders it
      v = j <= half ? sinf(f) : cosf(f);
    }
    out[i] = v;
  }
}
extern "C" int ofx_learned_sinusoid(const float* t, const float* w, int batch_size, int half, float* out, void* stream) {
  if (!t || !w || !out || batch_size < 1 || half < 1) return OFX_EINVAL;
  learned_sinusoid_kernel<<<ofx_grid((int64_t)batch_size * (2 * half + 1), 256), 256, 0, ofx_stream(stream)>>>(
      t, w, batch_size, half, out);
  OFX_LAUNCH_CHECK();
  return OFX_OK;
}

// ---------------------------------------------------------------------------------
// Linear layer on a HANDFUL of rows (the time / label embedding MLPs and the per-block embedding projections:
// M = batch size <= 16 rows, K, N = 64 .. 2048): out = act_out(act_in(a) @ W^T + bias + res), W [N, K] as nn.Linear
// stores it -- no packing, exact fp32 FMA.  The MFMA GEMM spends 10-16 us on these (a 128-row tile for 8 rows, split-K
// + a reduce launch); here a wave owns FOUR output columns, its lanes split K in float4 steps and the M x 4 sums are
// reduced with xor-shuffles: the launch is bound by streaming W once (<= 4 MB) from L2 / HBM.
__device__ __forceinline__ float ls_act(float v, int act) {
  if (act == OFX_ACT_SILU) return v / (1.f + __expf(-v));
  if (act == OFX_ACT_GELU) return 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));
  return v;
}
template <int MB>
__global__ void __launch_bounds__(256) linear_small_kernel(const float* __restrict__ a, int64_t lda, int M, int K,
                                                           const float* __restrict__ W, int64_t ldw, int N,
                                                           const float* __restrict__ bias, const float* __restrict__ res,
                                                           int64_t ldr, int act_in, int act_out, float* __restrict__ out,
                                                           int64_t ldo, int vec) {
  const int lane = threadIdx.x & 63;
  const int n0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 4;
  if (n0 >= N) return;
  float acc[MB][4];
#pragma unroll
  for (int m = 0; m < MB; ++m)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[m][j] = 0.f;
  const float* wr[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) wr[j] = W + (int64_t)(n0 + j < N ? n0 + j : N - 1) * ldw;
  if (vec) {
    for (int k = lane * 4; k < K; k += 256) {
      float4 wv[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) wv[j] = *reinterpret_cast<const float4*>(wr[j] + k);
#pragma unroll
      for (int m = 0; m < MB; ++m) {
        float4 av = *reinterpret_cast<const float4*>(a + (int64_t)(m < M ? m : M - 1) * lda + k);
        av.x = ls_act(av.x, act_in); av.y = ls_act(av.y, act_in); av.z = ls_act(av.z, act_in); av.w = ls_act(av.w, act_in);
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[m][j] = fmaf(av.x, wv[j].x, fmaf(av.y, wv[j].y, fmaf(av.z, wv[j].z, fmaf(av.w, wv[j].w, acc[m][j]))));
      }
    }
  } else {
    for (int k = lane; k < K; k += 64) {
      float wv[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) wv[j] = wr[j][k];
#pragma unroll
      for (int m = 0; m < MB; ++m) {
        const float av = ls_act(a[(int64_t)(m < M ? m : M - 1) * lda + k], act_in);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[m][j] = fmaf(av, wv[j], acc[m][j]);
      }
    }
  }
#pragma unroll
  for (int m = 0; m < MB; ++m)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float v = acc[m][j];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
      acc[m][j] = v;
    }
  // lane (m, j) = lane index m * 4 + j writes one output
  if (lane < MB * 4) {
    const int m = lane >> 2, j = lane & 3;
    if (m < M && n0 + j < N) {
      float v = 0.f;
#pragma unroll
      for (int mm = 0; mm < MB; ++mm)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
          if (mm == m && jj == j) v = acc[mm][jj];
      if (bias) v += bias[n0 + j];
      if (res) v += res[(int64_t)m * ldr + n0 + j];
      out[(int64_t)m * ldo + n0 + j] = ls_act(v, act_out);
    }
  }
}
extern "C" int ofx_linear_small(const float* a, int64_t lda, int M, int K, const float* W, int64_t ldw, int N,
                                const float* bias, const float* res, int64_t ldr, int act_in, int act_out, float* out,
                                int64_t ldo, void* stream) {
  if (!a || !W || !out || M < 1 || M > 16 || K < 1 || N < 1 || lda < K || ldw < K || ldo < N || (res && ldr < N) ||
      act_in < 0 || act_in > OFX_ACT_GELU || act_out < 0 || act_out > OFX_ACT_GELU)
    return OFX_EINVAL;
  const int vec = (K % 4 == 0) && (lda % 4 == 0) && (ldw % 4 == 0) && !(((uintptr_t)a | (uintptr_t)W) & 15);
  const int grid = (int)ofx_cdiv(N, 16);
  hipStream_t st = ofx_stream(stream);
#define LS_GO(MB_) linear_small_kernel<MB_><<<grid, 256, 0, st>>>(a, lda, M, K, W, ldw, N, bias, res, ldr, act_in, act_out, out, ldo, vec)
  if (M <= 1) LS_GO(1);
  else if (M <= 2) LS_GO(2);
  else if (M <= 4) LS_GO(4);
  else if (M <= 8) LS_GO(8);
  else LS_GO(16);
#undef LS_GO
  OFX_LAUNCH_CHECK();
  return OFX_OK;
}

__global__ void ddim_eps_kernel(float* __restrict__ x, const float* __restrict__ eps, const float* __restrict__ coef,
                                float* __restrict__ x0_out, int64_t n) {
  const float alpha = coef[0], sigma = coef[1], alpha_n = coef[2], sigma_n = coef[3];
  const float a = fmaxf(alpha, 1e-8f);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float e = eps[i];
    const float x0 = (x[i] - e * sigma) / a;
    if (x0_out) x0_out[i] = x0;
    x[i] = x0 * alpha_n + e * sigma_n;
  }
}
extern "C" int ofx_ddim_eps_update(float* x, const float* eps, const float* coef, float* x0_out, int64_t n,
                                   void* stream) {
  if (n < 0 || !coef || (n > 0 && (!x || !eps))) return OFX_EINVAL;
  if (n > 0) ddim_eps_kernel<<<ofx_grid(n, 256), 256, 0, ofx_stream(stream)>>>(x, eps, coef, x0_out, n);
  OFX_LAUNCH_CHECK();
  return OFX_OK;
}

__global__ void ddim_x0_kernel(float* __restrict__ x, const float* __restrict__ x0, const float* __restrict__ noise,
                               const float* __restrict__ coef, int64_t n) {
  // mean = alpha_next * (x * (1 - c) / alpha + c * x0); x = mean + sd * noise
  const float alpha = coef[0], c = coef[1], alpha_n = coef[2], sd = coef[3];
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float mean = alpha_n * (x[i] * (1.f - c) / alpha + c * x0[i]);
    x[i] = mean + sd * (noise ? noise[i] : 0.f);
  }
}
extern "C" int ofx_ddim_x0_update(float* x, const float* x0, const float* noise, const float* coef, int64_t n,
                                  void* stream) {
  if (n < 0 || !coef || (n > 0 && (!x || !x0))) return OFX_EINVAL;
  if (n > 0) ddim_x0_kernel<<<ofx_grid(n, 256), 256, 0, ofx_stream(stream)>>>(x, x0, noise, coef, n);
  OFX_LAUNCH_CHECK();
  return OFX_OK;
}

// ---------------------------------------------------------------------------------
// Optimiser side of the training step (reference octfusion_model_union.py:142, 478-487): torch.optim.AdamW's
// update (decoupled weight decay, bias-corrected moments) and the EMA of the weights
// (ldm_diffusion_util.py:38-54: old * beta + (1 - beta) * new), one elementwise pass each.
__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                             float* __restrict__ v, int64_t n, float lr, float b1, float b2, float eps, float wd,
                             float bc1, float bc2) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float gi = g[i];
    float pi = p[i] * (1.f - lr * wd);
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi; v[i] = vi;
    const float denom = sqrtf(vi) / sqrtf(bc2) + eps;
    pi -= (lr / bc1) * (mi / denom);
    p[i] = pi;
  }
}
extern "C" int ofx_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                              float beta1, float beta2, float eps, float weight_decay, int step, void* stream) {
  if (n < 0 || step < 1 || (n > 0 && (!param || !grad || !exp_avg || !exp_avg_sq))) return OFX_EINVAL;
  if (n == 0) return OFX_OK;
  const float bc1 = 1.f - powf(beta1, (float)step), bc2 = 1.f - powf(beta2, (float)step);
  adamw_kernel<<<ofx_grid(n, 256), 256, 0, ofx_stream(stream)>>>(param, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2,
                                                                 eps, weight_decay, bc1, bc2);
  OFX_LAUNCH_CHECK();
  return OFX_OK;
}
__global__ void ema_kernel(float* __restrict__ e, const float* __restrict__ p, int64_t n, float beta) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    e[i] = e[i] * beta + (1.f - beta) * p[i];
}
extern "C" int ofx_ema_update(float* ema, const float* param, int64_t n, float beta, void* stream) {
  if (n < 0 || (n > 0 && (!ema || !param))) return OFX_EINVAL;
  if (n > 0) ema_kernel<<<ofx_grid(n, 256), 256, 0, ofx_stream(stream)>>>(ema, param, n, beta);
  OFX_LAUNCH_CHECK();
  return OFX_OK;
}

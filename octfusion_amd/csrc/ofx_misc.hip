// libofx: glue kernels (row permutations for pool/unpool, embeddings, DDIM updates),
// status strings, device check.  All HBM-bound elementwise work.
#include "ofx_common.h"

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

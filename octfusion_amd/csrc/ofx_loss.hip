// VAE training losses: the scalar objectives of reference
// models/networks/dualoctree_networks/loss.py:164-178 (geometry_loss) and the posterior of
// distributions.py:24-46, each as ONE pass that produces the loss sums AND the gradient with respect to the
// network outputs, so the backward of the VAE starts from buffers that are already in HBM.
//   ofx_octree_ce        compute_octree_loss (loss.py:110-122): 2-class cross entropy against nempty_mask + accuracy
//   ofx_sdf_reg_loss     sdf_reg_loss (loss.py:23-29): mean squared errors of the MPU value and gradient
//   ofx_kl_sample_fwd/bwd  DiagonalGaussianDistribution: clamp, sample with given noise, kl() (elementwise)
// All HBM-bound streaming reductions: block sums in fp64, one atomic pair per block.
#include "ofx_common.h"

namespace {

__device__ __forceinline__ double block_sum(double v, double* red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  double s = 0.0;
  if (threadIdx.x == 0)
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) s += red[i];
  __syncthreads();
  return s;   // valid in thread 0
}

__global__ void __launch_bounds__(256) octree_ce_kernel(const float* __restrict__ logits, int64_t ld,
                                                        const int32_t* __restrict__ child, int64_t n, float dscale,
                                                        double* __restrict__ sums, float* __restrict__ dlogits,
                                                        int64_t ldd) {
  __shared__ double red[4];
  double loss = 0.0, hit = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float l0 = logits[i * ld], l1 = logits[i * ld + 1];
    const int label = child[i] >= 0 ? 1 : 0;                       // nempty_mask(d) (loss.py:117)
    const float m = fmaxf(l0, l1);
    const float e0 = expf(l0 - m), e1 = expf(l1 - m);
    const float lse = m + logf(e0 + e1);
    loss += (double)(lse - (label ? l1 : l0));
    const int arg = l1 > l0 ? 1 : 0;                               // argmax returns the first maximum
    hit += arg == label ? 1.0 : 0.0;
    if (dlogits) {
      const float inv = 1.0f / (e0 + e1);
      dlogits[i * ldd] = (e0 * inv - (label ? 0.f : 1.f)) * dscale;
      dlogits[i * ldd + 1] = (e1 * inv - (label ? 1.f : 0.f)) * dscale;
    }
  }
  const double s0 = block_sum(loss, red);
  const double s1 = block_sum(hit, red);
  if (threadIdx.x == 0) { unsafeAtomicAdd(sums, s0); unsafeAtomicAdd(sums + 1, s1); }
}

__global__ void __launch_bounds__(256) sdf_reg_kernel(const float* __restrict__ sdf, const float* __restrict__ grad,
                                                      const float* __restrict__ sdf_gt,
                                                      const float* __restrict__ grad_gt, int64_t n, float cs, float cg,
                                                      double* __restrict__ sums, float* __restrict__ dsdf,
                                                      float* __restrict__ dgrad) {
  __shared__ double red[4];
  double sg = 0.0, ss = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float e = sdf[i] - sdf_gt[i];
    ss += (double)e * (double)e;
    if (dsdf) dsdf[i] = cs * e;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float g = grad[i * 3 + k] - grad_gt[i * 3 + k];
      sg += (double)g * (double)g;
      if (dgrad) dgrad[i * 3 + k] = cg * g;
    }
  }
  const double s0 = block_sum(sg, red);
  const double s1 = block_sum(ss, red);
  if (threadIdx.x == 0) { unsafeAtomicAdd(sums, s0); unsafeAtomicAdd(sums + 1, s1); }
}

__global__ void __launch_bounds__(256) kl_fwd_kernel(const float* __restrict__ params, int64_t ld,
                                                     const float* __restrict__ noise, int64_t n, int E,
                                                     float* __restrict__ z, double* __restrict__ kl_sum) {
  __shared__ double red[4];
  double acc = 0.0;
  const int64_t total = n * E;
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
    const int64_t r = t / E;
    const int c = (int)(t - r * E);
    const float mean = params[r * ld + c];
    const float lv = fminf(fmaxf(params[r * ld + E + c], -30.0f), 20.0f);     // distributions.py:28
    const float sd = expf(0.5f * lv), var = expf(lv);
    z[t] = mean + sd * (noise ? noise[t] : 0.f);
    acc += (double)(0.5f * (mean * mean + var - 1.0f - lv));                   // distributions.py:46
  }
  const double s = block_sum(acc, red);
  if (threadIdx.x == 0 && kl_sum) unsafeAtomicAdd(kl_sum, s);
}

__global__ void __launch_bounds__(256) kl_bwd_kernel(const float* __restrict__ params, int64_t ld,
                                                     const float* __restrict__ noise, const float* __restrict__ dz,
                                                     int64_t n, int E, float kl_scale, float* __restrict__ dparams,
                                                     int64_t ldp) {
  const int64_t total = n * E;
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
    const int64_t r = t / E;
    const int c = (int)(t - r * E);
    const float mean = params[r * ld + c];
    const float raw = params[r * ld + E + c];
    const float lv = fminf(fmaxf(raw, -30.0f), 20.0f);
    const float sd = expf(0.5f * lv), var = expf(lv);
    const float g = dz ? dz[t] : 0.f;
    dparams[r * ldp + c] = g + kl_scale * mean;
    const bool pass = raw >= -30.0f && raw <= 20.0f;                            // torch.clamp passes at the bounds
    const float dl = g * 0.5f * sd * (noise ? noise[t] : 0.f) + kl_scale * 0.5f * (var - 1.0f);
    dparams[r * ldp + E + c] = pass ? dl : 0.f;
  }
}

}  // namespace

extern "C" int ofx_octree_ce(const float* logits, int64_t ld, const int32_t* child, int64_t n, float dscale,
                             double* sums, float* dlogits, int64_t ldd, void* stream) {
  if (n < 0 || !sums || (n > 0 && (!logits || !child || ld < 2 || (dlogits && ldd < 2)))) return OFX_EINVAL;
  if (n == 0) return OFX_OK;
  octree_ce_kernel<<<ofx_grid(n, 256), 256, 0, ofx_stream(stream)>>>(logits, ld, child, n, dscale, sums, dlogits, ldd);
  OFX_LAUNCH_CHECK();
  return OFX_OK;
}

extern "C" int ofx_sdf_reg_loss(const float* sdf, const float* grad, const float* sdf_gt, const float* grad_gt,
                                int64_t n, float w_sdf, float w_grad, double* sums, float* dsdf, float* dgrad,
                                void* stream) {
  if (n < 0 || !sums || (n > 0 && (!sdf || !grad || !sdf_gt || !grad_gt))) return OFX_EINVAL;
  if (n == 0) return OFX_OK;
  // d/dsdf of w_sdf * mean(e^2) = 2 w_sdf e / n;   d/dgrad of w_grad * mean over 3n = 2 w_grad g / (3n)
  const float cs = 2.0f * w_sdf / (float)n, cg = 2.0f * w_grad / (3.0f * (float)n);
  sdf_reg_kernel<<<ofx_grid(n, 256), 256, 0, ofx_stream(stream)>>>(sdf, grad, sdf_gt, grad_gt, n, cs, cg, sums, dsdf,
                                                                   dgrad);
  OFX_LAUNCH_CHECK();
  return OFX_OK;
}

extern "C" int ofx_kl_sample_fwd(const float* params, int64_t ld, const float* noise, int64_t n, int embed_dim,
                                 float* z, double* kl_sum, void* stream) {
  if (n < 0 || embed_dim < 1 || (n > 0 && (!params || !z || ld < 2 * embed_dim))) return OFX_EINVAL;
  if (n == 0) return OFX_OK;
  kl_fwd_kernel<<<ofx_grid(n * embed_dim, 256), 256, 0, ofx_stream(stream)>>>(params, ld, noise, n, embed_dim, z,
                                                                              kl_sum);
  OFX_LAUNCH_CHECK();
  return OFX_OK;
}

extern "C" int ofx_kl_sample_bwd(const float* params, int64_t ld, const float* noise, const float* dz, int64_t n,
                                 int embed_dim, float kl_scale, float* dparams, int64_t ldp, void* stream) {
  if (n < 0 || embed_dim < 1 || (n > 0 && (!params || !dparams || ld < 2 * embed_dim || ldp < 2 * embed_dim)))
    return OFX_EINVAL;
  if (n == 0) return OFX_OK;
  kl_bwd_kernel<<<ofx_grid(n * embed_dim, 256), 256, 0, ofx_stream(stream)>>>(params, ld, noise, dz, n, embed_dim,
                                                                              kl_scale, dparams, ldp);
  OFX_LAUNCH_CHECK();
  return OFX_OK;
}

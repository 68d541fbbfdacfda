// Shared helpers for libofx (gfx950 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ofx.h"

#define OFX_LAUNCH_CHECK()                         \
  do {                                             \
    if (hipGetLastError() != hipSuccess) return OFX_ELAUNCH; \
  } while (0)

static inline hipStream_t ofx_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

static inline int64_t ofx_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE attribute: a process that launches on a second GPU
// must set it there too.  `done` is the caller's per-kernel flag array.  Returns false on failure.
constexpr int OFX_MAX_DEVICES = 64;
static inline bool ofx_raise_lds_limit(const void* kernel, int bytes, bool (&done)[OFX_MAX_DEVICES]) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= OFX_MAX_DEVICES) return false;
  if (done[dev]) return true;
  if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return false;
  done[dev] = true;
  return true;
}

// fp16x3 range guard (include/ofx.h): per-device pointer to the caller's sticky words, or NULL.  Defined in ofx_misc.hip.
uint32_t* ofx_range_words();

// grid for a grid-stride elementwise kernel: enough blocks to fill 256 CUs x 8.
static inline int ofx_grid(int64_t work_items, int block) {
  int64_t g = ofx_cdiv(work_items, block);
  if (g < 1) g = 1;
  if (g > 2048 * 4) g = 2048 * 4;
  return (int)g;
}

// ---- Morton codec (x -> bit 3i+2, y -> 3i+1, z -> 3i; batch id in bits 48..) ----
__host__ __device__ static inline uint32_t ofx_compact3(uint64_t v) {
  // keep every 3rd bit of v (bit 0, 3, 6, ...) and pack them.
  v &= 0x1249249249249249ull;
  v = (v ^ (v >> 2)) & 0x10c30c30c30c30c3ull;
  v = (v ^ (v >> 4)) & 0x100f00f00f00f00full;
  v = (v ^ (v >> 8)) & 0x1f0000ff0000ffull;
  v = (v ^ (v >> 16)) & 0x1f00000000ffffull;
  v = (v ^ (v >> 32)) & 0x1fffffull;
  return (uint32_t)v;
}
__host__ __device__ static inline uint64_t ofx_spread3(uint64_t v) {
  v &= 0x1fffffull;
  v = (v | (v << 32)) & 0x1f00000000ffffull;
  v = (v | (v << 16)) & 0x1f0000ff0000ffull;
  v = (v | (v << 8)) & 0x100f00f00f00f00full;
  v = (v | (v << 4)) & 0x10c30c30c30c30c3ull;
  v = (v | (v << 2)) & 0x1249249249249249ull;
  return v;
}
__host__ __device__ static inline void ofx_key2xyz(int64_t key, int& x, int& y, int& z, int& b) {
  const uint64_t k = (uint64_t)key & ((1ull << 48) - 1);
  b = (int)((uint64_t)key >> 48);
  z = (int)ofx_compact3(k);
  y = (int)ofx_compact3(k >> 1);
  x = (int)ofx_compact3(k >> 2);
}
__host__ __device__ static inline uint64_t ofx_xyz2morton(int x, int y, int z) {
  return (ofx_spread3((uint64_t)x) << 2) | (ofx_spread3((uint64_t)y) << 1) | ofx_spread3((uint64_t)z);
}

// What would a per-XCD persistent design of the coarse dense levels buy?  It replaces kernel boundaries
// by barriers among resident blocks.  This probe prices both sides on the machine:
//
//   1. a chain of N dependent EMPTY kernels (256 blocks x 256 threads), replayed from a hipGraph: us per kernel boundary;
//      the same chain with every kernel touching 1 MB (a realistic small layer's output) so that the ramp is not hidden;
//   2. a persistent launch of 256 blocks (one per CU) doing N barriers:
//        A  device-wide, agent scope: one counter, release = agent fence, acquire = agent fence (what a grid barrier costs);
//        B  per XCD: the blocks that share an XCC_ID use that XCD's counter; the atomic is performed at the XCD's own L2
//           (RMW without sc1), stores are made visible with s_waitcnt vmcnt(0) (the L1 is write-through) and
//           readers invalidate their L1 (buffer_inv sc1) -- no L2 write-back, no cross-XCD traffic;
//      each barrier is followed by a token exchange (block i reads what block i+1 of its group wrote before the barrier)
//      so that a barrier that does not make data visible FAILS the check instead of looking fast;
//   3. the dispatch order: XCC_ID of block b (is it b mod 8?).
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/xcd_barrier tools/probes/xcd_barrier.hip && tools/probes/xcd_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void empty_kernel(float* p, int touch) {
  if (touch) {
    // 256 blocks x 256 threads x float4 = 1 MB read-modify-write
    float4* q = reinterpret_cast<float4*>(p) + (size_t)blockIdx.x * 256 + threadIdx.x;
    float4 v = *q;
    v.x += 1.f;
    *q = v;
  }
}

__device__ __forceinline__ unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0xf;
}

__device__ __forceinline__ unsigned l2_read(unsigned* p) {
  unsigned ret, zero = 0;
  asm volatile("global_atomic_add %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(ret) : "v"(p), "v"(zero) : "memory");
  return ret;
}

struct BarArgs {
  unsigned* counters;     // [9]: per XCD 0..7 (128-B apart), [8] device-wide
  unsigned* tokens;       // [blocks] one word per block, 128-B apart
  unsigned* xcc;          // [blocks] XCC_ID seen by the block
  unsigned* rank;         // [blocks] rank of the block inside its XCD (arrival order at start)
  unsigned* err;          // mismatches of the token exchange
  unsigned long long* ticks;
  int n, mode, per_group;
};

// mode 0: device-wide agent-scope barrier; mode 1: XCD-local barrier (L2-scope RMW + L1 invalidate)
__global__ void __launch_bounds__(256) barrier_kernel(BarArgs a) {
  __shared__ unsigned s_xcc, s_rank;
  const unsigned b = blockIdx.x;
  if (threadIdx.x == 0) {
    s_xcc = xcc_id();
    a.xcc[b] = s_xcc;
    // rank inside the XCD: one agent-scope ticket per XCD at start (not timed)
    s_rank = __hip_atomic_fetch_add(a.rank + gridDim.x + s_xcc * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    a.rank[b] = s_rank;
  }
  __syncthreads();
  const unsigned xcc = s_xcc;
  const int group = a.mode == 0 ? 8 : (int)xcc;
  const unsigned members = a.mode == 0 ? gridDim.x : (unsigned)a.per_group;
  unsigned* ctr = a.counters + group * 32;
  // everyone resident before the clock starts (device-wide, agent scope)
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(a.counters + 9 * 32, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    while (__hip_atomic_load(a.counters + 9 * 32, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x) __builtin_amdgcn_s_sleep(2);
  }
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  unsigned bad = 0;
  const unsigned my_rank = s_rank;
  // token slot of this block and of its neighbour in the group (by rank inside the XCD for mode 1, by index for mode 0)
  const unsigned me = a.mode == 0 ? b : xcc * 32 + my_rank;
  const unsigned nb = a.mode == 0 ? (b + 1) % gridDim.x : xcc * 32 + (my_rank + 1) % members;
  bool dead = false;
  for (int it = 1; it <= a.n && !dead; ++it) {
    if (threadIdx.x == 0) {
      a.tokens[me * 32] = (unsigned)it * 1000u + me;                    // plain store (write-through L1 -> L2)
      unsigned spins = 0;
      if (a.mode == 0) {
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)it * members) {
          __builtin_amdgcn_s_sleep(1);
          if (++spins > 4000000u) { dead = true; break; }              // a bounded wait: report, do not hang the box
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // the token has left for L2
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);      // RMW at this XCD's L2
        // poll with a returning RMW of zero: atomics never hit the L1, and one without sc1 stays in this XCD's L2
        // (inline asm: the compiler turns `atomicrmw add 0` into an atomic LOAD, which may be served by the L1 for ever)
        while (l2_read(ctr) < (unsigned)it * members) {
          __builtin_amdgcn_s_sleep(1);
          if (++spins > 4000000u) { dead = true; break; }
        }
        asm volatile("buffer_inv sc1" ::: "memory");                   // drop this CU's L1 lines (sc0 alone is the workgroup's own L1: a no-op here)
      }
      const unsigned got = a.tokens[nb * 32];
      // (the neighbour may already have written the NEXT round's token: both are proof that round `it` was visible)
      if (got != (unsigned)it * 1000u + nb && got != (unsigned)(it + 1) * 1000u + nb) ++bad;
      if (dead) bad += 1000000u;
      s_rank = dead ? 0xffffffffu : s_rank;
    }
    __syncthreads();
    dead = s_rank == 0xffffffffu;
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) {
    a.ticks[b] = t1 - t0;
    if (bad) atomicAdd(a.err, bad);
  }
}

int main() {
  int dev = 0;
  CK(hipSetDevice(dev));
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, dev));
  const int cus = prop.multiProcessorCount;
  printf("device %s, %d CUs, clock %d kHz\n", prop.name, cus, prop.clockRate);
  hipStream_t st;
  CK(hipStreamCreate(&st));

  // ---- 1. kernel boundaries in a hipGraph ----------------------------------------------------------------
  float* buf;
  CK(hipMalloc(&buf, 1 << 20));
  CK(hipMemset(buf, 0, 1 << 20));
  for (int touch = 0; touch < 2; ++touch) {
    const int N = 200;
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int i = 0; i < N; ++i) empty_kernel<<<256, 256, 0, st>>>(buf, touch);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int w = 0; w < 3; ++w) CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    const int reps = 20;
    CK(hipEventRecord(e0, st));
    for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("hipGraph chain of %d dependent kernels (256 x 256 threads, %s): %.2f us per kernel\n", N,
           touch ? "1 MB read-modify-write each" : "empty", 1e3 * ms / (reps * N));
    // the same chain as plain stream launches
    for (int i = 0; i < N; ++i) empty_kernel<<<256, 256, 0, st>>>(buf, touch);
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int r = 0; r < 5; ++r) for (int i = 0; i < N; ++i) empty_kernel<<<256, 256, 0, st>>>(buf, touch);
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("   the same as stream launches: %.2f us per kernel\n", 1e3 * ms / (5 * N));
    CK(hipGraphExecDestroy(ge));
    CK(hipGraphDestroy(g));
  }

  // ---- 2. barriers among resident blocks ---------------------------------------------------------------------
  const int blocks = cus & ~7;
  unsigned *counters, *tokens, *xcc, *rank, *err;
  unsigned long long* ticks;
  CK(hipMalloc(&counters, 10 * 32 * 4));
  CK(hipMalloc(&tokens, (size_t)(blocks + 256) * 32 * 4));
  CK(hipMalloc(&xcc, blocks * 4));
  CK(hipMalloc(&rank, (blocks + 64 + 8 * 32) * 4));
  CK(hipMalloc(&err, 4));
  CK(hipMalloc(&ticks, blocks * 8));
  std::vector<unsigned> hx(blocks), hr(blocks);
  std::vector<unsigned long long> ht(blocks);
  for (int mode = 0; mode < 2; ++mode) {
    for (int n : {1000, 5000}) {
      CK(hipMemset(counters, 0, 10 * 32 * 4));
      CK(hipMemset(tokens, 0, (size_t)(blocks + 256) * 32 * 4));
      CK(hipMemset(rank, 0, (blocks + 64 + 8 * 32) * 4));
      CK(hipMemset(err, 0, 4));
      BarArgs a{counters, tokens, xcc, rank, err, ticks, n, mode, blocks / 8};
      hipEvent_t e0, e1;
      CK(hipEventCreate(&e0));
      CK(hipEventCreate(&e1));
      CK(hipEventRecord(e0, st));
      barrier_kernel<<<blocks, 256, 0, st>>>(a);
      CK(hipEventRecord(e1, st));
      CK(hipStreamSynchronize(st));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      unsigned herr;
      CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(hx.data(), xcc, blocks * 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(hr.data(), rank, blocks * 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(ht.data(), ticks, blocks * 8, hipMemcpyDeviceToHost));
      int per[16] = {0}, mod8 = 0;
      for (int b = 0; b < blocks; ++b) { per[hx[b] & 15]++; mod8 += (int)(hx[b] == (unsigned)(b % 8)); }
      printf("%s barrier x %d among %d blocks: %.3f us per barrier (kernel %.1f us), token mismatches %u\n",
             mode == 0 ? "device-wide agent-scope" : "per-XCD (L2-scope RMW + L1 invalidate)", n,
             mode == 0 ? blocks : blocks / 8, 1e3 * ms / n, 1e3 * ms, herr);
      if (n == 1000) {
        printf("   blocks per XCC_ID:");
        for (int x = 0; x < 8; ++x) printf(" %d", per[x]);
        printf("; XCC_ID == block %% 8 for %d of %d blocks; XCC_ID of blocks 0..15:", mod8, blocks);
        for (int b = 0; b < 16; ++b) printf(" %u", hx[b]);
        int same = 0;
        for (int b = 8; b < blocks; ++b) same += (int)(hx[b] == hx[b - 8]);
        printf("; XCC_ID(b) == XCC_ID(b - 8) for %d of %d\n", same, blocks - 8);
      }
    }
  }
  return 0;
}

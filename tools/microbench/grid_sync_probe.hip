// What does one dependent step cost on MI355X -- as a kernel boundary, and as a barrier across the 256 resident workgroups of ONE launch?
// (round 6: the numbers that decide whether a per-layer persistent kernel for the one-utterance forward can pay: a 5 s utterance is ~130
// launches of 4-14 us.)
//   hipcc --offload-arch=gfx950 -O3 -o grid_sync_probe tools/microbench/grid_sync_probe.hip && ./grid_sync_probe
// Prints: us per launch of a chain of dependent near-empty kernels (same stream), us per grid barrier (monotonic counter, agent-scope
// atomics, every workgroup resident), and the same with a write -> barrier -> read-other-XCD's-data check per step (what a phase
// boundary of a fused layer needs: stores visible chip-wide before the next phase reads them).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(256) void tiny_kernel(float* p, int step) {
  if (threadIdx.x == 0) p[blockIdx.x] += (float)step;
}

// barrier over the gridDim.x resident workgroups: thread 0 arrives on a monotonic counter and spins until everyone has
__device__ __forceinline__ void grid_barrier(unsigned* counter, unsigned target) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
  }
  __syncthreads();
}

// the same with the waiting done by RELAXED loads (no cache maintenance per poll) and one acquire fence behind the loop: what a tuned
// phase boundary would use (the acquire loads of the first form invalidate the XCD's L2 on every poll)
__device__ __forceinline__ void grid_barrier_tuned(unsigned* counter, unsigned target) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
  }
  __syncthreads();
}
__global__ __launch_bounds__(256) void barrier_tuned_kernel(unsigned* counter, int steps, unsigned long long* cycles) {
  const unsigned n = gridDim.x;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  for (int s = 1; s <= steps; ++s) grid_barrier_tuned(counter, (unsigned)s * n);
  if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = __builtin_amdgcn_s_memrealtime() - t0;
}

template <bool CHECK>
__global__ __launch_bounds__(256) void barrier_kernel(unsigned* counter, int steps, unsigned* data, unsigned* bad, unsigned long long* cycles) {
  const unsigned n = gridDim.x;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  for (int s = 1; s <= steps; ++s) {
    if (CHECK) {   // every thread writes a value of this step; after the barrier reads the slot of a workgroup half the grid away
      __hip_atomic_store(data + blockIdx.x * 256 + threadIdx.x, (unsigned)s * 1000u + blockIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    grid_barrier(counter, (unsigned)s * n);
    if (CHECK) {
      const unsigned other = (blockIdx.x + n / 2 + (unsigned)s) % n;
      const unsigned v = __hip_atomic_load(data + other * 256 + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (v != (unsigned)s * 1000u + other && v != (unsigned)(s + 1) * 1000u + other) atomicAdd(bad, 1u);
      grid_barrier(counter + 64, (unsigned)s * n);   // nobody overwrites before everybody has read
    }
  }
  if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = __builtin_amdgcn_s_memrealtime() - t0;   // 100 MHz ticks
}

int main() {
  float* p; unsigned *counter, *data, *bad; unsigned long long* cyc;
  CK(hipMalloc(&p, 4096 * 4)); CK(hipMemset(p, 0, 4096 * 4));
  CK(hipMalloc(&counter, 1024)); CK(hipMalloc(&data, 256 * 256 * 4)); CK(hipMalloc(&bad, 4)); CK(hipMalloc(&cyc, 8));
  hipStream_t s; CK(hipStreamCreate(&s));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int grid : {64, 256}) {
    for (int rep = 0; rep < 2; ++rep) {
      const int n = 2000;
      CK(hipEventRecord(e0, s));
      for (int i = 0; i < n; ++i) hipLaunchKernelGGL(tiny_kernel, dim3(grid), dim3(256), 0, s, p, i);
      CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep) printf("chain of %d dependent near-empty kernels, grid %3d: %.2f us per launch\n", n, grid, 1e3 * ms / n);
    }
  }
  for (int grid : {64, 128, 256}) {
    for (int check = 0; check < 2; ++check) {
      const int steps = 2000;
      CK(hipMemsetAsync(counter, 0, 1024, s)); CK(hipMemsetAsync(bad, 0, 4, s)); CK(hipMemsetAsync(data, 0, 256 * 256 * 4, s));
      CK(hipEventRecord(e0, s));
      if (check) hipLaunchKernelGGL(barrier_kernel<true>, dim3(grid), dim3(256), 0, s, counter, steps, data, bad, cyc);
      else hipLaunchKernelGGL(barrier_kernel<false>, dim3(grid), dim3(256), 0, s, counter, steps, data, bad, cyc);
      CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      unsigned hb; unsigned long long hc; CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost));
      printf("grid barrier, %3d workgroups%s: %.2f us per step (kernel %.3f ms, in-kernel %.2f us per step), stale reads %u\n", grid,
             check ? ", write -> barrier -> read another workgroup's slot -> barrier" : "", 1e3 * ms / steps, ms, hc * 0.01 / steps, hb);
    }
  }
  for (int grid : {64, 128, 256}) {
    const int steps = 2000;
    CK(hipMemsetAsync(counter, 0, 1024, s));
    CK(hipEventRecord(e0, s));
    hipLaunchKernelGGL(barrier_tuned_kernel, dim3(grid), dim3(256), 0, s, counter, steps, cyc);
    CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("grid barrier, %3d workgroups, relaxed polling + one acquire fence: %.2f us per step\n", grid, 1e3 * ms / steps);
  }
  return 0;
}

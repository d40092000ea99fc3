// Feasibility probe (round 5): can ONE wave per SIMD keep the matrix pipe busy through a GEMM slab of the 256 x 256 x 64 tile when the slab's
// 32 fragment reads, 16 LDS-DMA requests and one barrier are interleaved into ITS OWN stream of 128 MFMAs (the single-wave, software-pipelined
// main loop of the vendor library's large-tile kernels), instead of being issued by a SIMD partner as gemm_pps_kernel does?
//   hipcc --offload-arch=gfx950 -O3 -o p1w_probe tools/microbench/p1w_probe.hip && ./p1w_probe
// Prints core cycles per slab for: MFMAs alone / + fragment reads / + requests / + barrier (2 048 = the matrix pipe's own time).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void dma_sv(unsigned voff, const void* sbase, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_addr), "v"(voff), "s"(sbase) : "memory");
}

// MODE bit 0: fragment reads, bit 1: LDS-DMA requests, bit 2: barrier per slab; WPS = waves per SIMD (1: 256 threads, 2: 512 threads with half the tile each)
template <int MODE>
__global__ __launch_bounds__(256) void probe_kernel(const char* __restrict__ gA, int slabs, unsigned long long* out, float* sink) {
  extern __shared__ __attribute__((aligned(16))) uint4 lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(void __attribute__((address_space(3)))*)lds);
  for (int i = tid; i < 10240; i += 256) {   // random bf16 values in [-2, 2): constant operands let the chip hold a higher clock than real data does
    unsigned h = (unsigned)i * 2654435761u + blockIdx.x * 40503u;
    auto nx = [&]() { h ^= h << 13; h ^= h >> 17; h ^= h << 5; return (h & 0x807f807fu) | 0x3f803f80u; };
    lds[i] = uint4{nx(), nx(), nx(), nx()};
  }
  __syncthreads();
  f32x4 acc[8][8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 xf[2][8], wf[2][8];
  const int r16 = lane & 15, cq = lane >> 4, rr8 = r16 & 7;
  const int frag0 = (r16 >> 3) * 64 + rr8 * 8 + (cq ^ rr8), frag1 = (r16 >> 3) * 64 + rr8 * 8 + ((4 + cq) ^ rr8);
  const int wm = wave & 1, wn = wave >> 1;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    xf[0][j] = __builtin_bit_cast(bf16x8, lds[(wm * 8 + j) * 128 + frag0]);
    wf[0][j] = __builtin_bit_cast(bf16x8, lds[2048 + (wn * 8 + j) * 128 + frag0]);
  }
  // per-lane source offsets of this wave's 16 requests per slab: 8 rows x 128 B each, row pitch 8 KiB (K = 4096 bf16)
  unsigned voff[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) voff[i] = (unsigned)((((blockIdx.x & 7) * 512 + (wave * 16 + i) * 8 + (lane >> 3)) * 8192u + ((lane & 7) ^ (lane >> 3)) * 16u) & 0x1fffffffu);   // 8 row sets of 512 rows, re-read by 32 workgroups each: L2-resident like a GEMM's operands
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  int sa = 0;
  for (int g = 0; g < slabs; ++g) {
    const uint4* A = lds + sa * 2048;
    const uint4* W = lds + (sa + 1 >= 5 ? sa - 4 : sa + 1) * 2048;
    const int s3 = sa + 3 >= 5 ? sa - 2 : sa + 3, s4 = sa + 4 >= 5 ? sa - 1 : sa + 4;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      // 64 MFMAs of this k-step in 16 groups of 4; group i also reads one A and one W fragment of the NEXT k-step and, every second group,
      // issues one request
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int nb = i >> 1, mb0 = (i & 1) * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[nb][mb0 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[ks][mb0 + j], wf[ks][nb], acc[nb][mb0 + j], 0, 0, 0);
        if ((MODE & 1) && i < 8) {
          xf[ks ^ 1][i] = __builtin_bit_cast(bf16x8, A[(wm * 8 + i) * 128 + (ks ? frag0 : frag1)]);
          wf[ks ^ 1][i] = __builtin_bit_cast(bf16x8, W[(wn * 8 + i) * 128 + (ks ? frag0 : frag1)]);
        }
        if ((MODE & 2) && (i & 1)) {
          const int r = ks * 8 + (i >> 1);
          dma_sv(voff[r] + (unsigned)(g & 63) * 128u, gA, lds0 + (unsigned)((r < 8 ? s3 : s4) * 2048 + wave * 512 + (r & 7) * 64) * 16u);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (ks == 0 && (MODE & 4)) {
        if (MODE & 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
    }
    sa = sa + 2 >= 5 ? sa - 3 : sa + 2;
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) s += acc[i][j][0] + acc[i][j][3];
  if (s == 12345.678f) sink[tid] = s;
  if (lane == 0) out[blockIdx.x * 4 + wave] = t1 - t0;
}

__global__ void fill_kernel(uint4* p, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned h = (unsigned)i * 2654435761u + 12345u;
    auto nx = [&]() { h ^= h << 13; h ^= h >> 17; h ^= h << 5; return (h & 0x807f807fu) | 0x3f803f80u; };
    p[i] = uint4{nx(), nx(), nx(), nx()};
  }
}

template <int MODE> void run(const char* name, const char* gA, int slabs, unsigned long long* d_out, float* sink) {
  hipFuncSetAttribute((const void*)probe_kernel<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((probe_kernel<MODE>), dim3(256), dim3(256), 163840, 0, gA, slabs, d_out, sink);
  hipEventRecord(e0);
  hipLaunchKernelGGL((probe_kernel<MODE>), dim3(256), dim3(256), 163840, 0, gA, slabs, d_out, sink);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(1024);
  hipMemcpy(h.data(), d_out, 1024 * 8, hipMemcpyDeviceToHost);
  double sum = 0; for (auto v : h) sum += (double)v;
  const double cyc = sum / 1024 / slabs;
  // s_memtime counts at a constant 100 MHz on this chip: convert with the wall time
  const double us_per_slab = ms * 1e3 / slabs;
  const double tflops = 256.0 * 2 * 256 * 256 * 64 / (us_per_slab * 1e-6) / 1e12;
  printf("%-44s %8.3f us per slab  (%7.1f TFLOP/s chip-wide)   s_memtime ticks per slab %.1f\n", name, us_per_slab, tflops, cyc);
}

int main() {
  char* gA; unsigned long long* d_out; float* sink;
  hipMalloc(&gA, (size_t)1 << 30); hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, (uint4*)gA, ((size_t)1 << 30) / 16); hipDeviceSynchronize();
  hipMalloc(&d_out, 1024 * 8); hipMalloc(&sink, 4096);
  const int slabs = 2000;
  run<0>("128 MFMAs per slab alone", gA, slabs, d_out, sink);
  run<1>("+ 32 fragment reads", gA, slabs, d_out, sink);
  run<3>("+ 16 LDS-DMA requests", gA, slabs, d_out, sink);
  run<7>("+ counted waits and one barrier", gA, slabs, d_out, sink);
  run<5>("reads + barrier, no requests", gA, slabs, d_out, sink);
  return 0;
}

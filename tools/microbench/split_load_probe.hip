// Does `s_waitcnt vmcnt(N)` cover a 16-byte row that was fetched as an OVERLAPPING pair of loads -- global_load_dwordx3 at byte 4 +
// global_load_dwordx2 at byte 0 -- the way hipcc split the first load of head_dots_kernel in round 4?  (kernels.hip; profiles/
// r05_determinism_under_gpu_sharing.txt: under memory contention the values read right behind the counted wait were stale in ~0.15 % of
// the forwards.)  Every lane issues the pair (mode 0) or ONE global_load_dwordx4 (mode 1) for a random 16-byte row, then 14 more 16-byte
// loads of other random rows, waits with the count that leaves exactly those 14 in flight, copies the registers at once, then waits for
// everything and compares the copy with what the registers hold now.  A difference = the counted wait let a value be read before it landed.
//   hipcc --offload-arch=gfx950 -O3 -o split_load_probe split_load_probe.hip ;  ./split_load_probe <mode> [iters] [MiB] [coalesced 0|1] [squares 0|1]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f3 __attribute__((ext_vector_type(3)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void probe(const float* __restrict__ x, unsigned long n_rows, int iters, unsigned long long* bad, unsigned long long* done,
                                             int coalesced, int squares) {
  unsigned long long s = (unsigned long long)(blockIdx.x * 256 + threadIdx.x) * 0x9E3779B97F4A7C15ull + 12345;
  unsigned long long nbad = 0;
  f3 a = {0.f, 0.f, 0.f};
  f2 b = {0.f, 0.f};
  f4 w = {0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
    const float* q[15];
#pragma unroll
    for (int j = 0; j < 15; ++j) {
      s = s * 6364136223846793005ull + 1442695040888963407ull;
      unsigned long long r = s >> 20;
      if (coalesced) {   // the lanes of a wave read 64 consecutive rows (head_dots_kernel's pattern): one random base per wave and load
        r = __shfl((unsigned long long)r, 0, 64) + (threadIdx.x & 63);
      }
      q[j] = x + (r % n_rows) * 4;
    }
    f4 c[14];
    if (MODE == 0) asm volatile("global_load_dwordx3 %0, %2, off offset:4\n\tglobal_load_dwordx2 %1, %2, off" : "=&v"(a), "=&v"(b) : "v"(q[0]) : "memory");
    else asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(w) : "v"(q[0]) : "memory");
#pragma unroll
    for (int j = 0; j < 14; ++j) asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(c[j]) : "v"(q[j + 1]) : "memory");
    f3 sa; f2 sb; f4 sw;
    if (MODE == 0) {
      asm volatile("s_waitcnt vmcnt(14)" : "+v"(a), "+v"(b) :: "memory");
      if (squares == 2) {
        // OVERWRITE the destination registers of the two completed loads right behind the counted wait (hipcc reused one of them as a
        // temporary: `v_mul_f32 v34, v36, v36` behind `s_waitcnt vmcnt(14)`), wait for everything, and see whether the new values survive:
        // a load whose data lands in two steps, counted as complete at the first, would write its second step over them
        const f3 ka = a; const f2 kb = b;
        asm volatile("v_mov_b32 %0, 0x7fc01234\n\tv_mov_b32 %1, 0x7fc01234\n\tv_mov_b32 %2, 0x7fc01234\n\tv_mov_b32 %3, 0x7fc01234\n\tv_mov_b32 %4, 0x7fc01234"
                     : "+v"(a.x), "+v"(a.y), "+v"(a.z), "+v"(b.x), "+v"(b.y));
        asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 7\n\ts_nop 7" : "+v"(a.x), "+v"(a.y), "+v"(a.z), "+v"(b.x), "+v"(b.y) :: "memory");
        const unsigned S = 0x7fc01234u;
        nbad += (__builtin_bit_cast(unsigned, a.x) != S) + (__builtin_bit_cast(unsigned, a.y) != S) + (__builtin_bit_cast(unsigned, a.z) != S) +
                (__builtin_bit_cast(unsigned, b.x) != S) + (__builtin_bit_cast(unsigned, b.y) != S);
        float keep2 = ka.x + kb.x;
        asm volatile("" :: "v"(keep2));
        float keep = 0.f;
#pragma unroll
        for (int j = 0; j < 14; ++j) { asm volatile("" : "+v"(c[j])); keep += c[j].x; }
        if (keep == 1.2345e-30f) nbad += 1000000;
        continue;
      }
      float q0 = 0.f;
      if (squares) {   // what the kernel did right behind the wait: squares of the four values (hipcc packs them)
        q0 = (b.x * b.x + a.x * b.y) + (a.y * a.y + a.z * a.z);
        asm volatile("" : "+v"(q0));
      }
      sa = a; sb = b;
      asm volatile("" : "+v"(sa), "+v"(sb));
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(a), "+v"(b) :: "memory");
      nbad += (sa.x != a.x) + (sa.y != a.y) + (sa.z != a.z) + (sb.x != b.x) + (sb.y != b.y);
      nbad += (b.y != a.x);   // the two loads overlap in one float: both copies must agree in the end
      if (squares) nbad += (q0 != (b.x * b.x + a.x * b.y) + (a.y * a.y + a.z * a.z));
    } else {
      asm volatile("s_waitcnt vmcnt(14)" : "+v"(w) :: "memory");
      sw = w;
      asm volatile("" : "+v"(sw));
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(w) :: "memory");
      nbad += (sw.x != w.x) + (sw.y != w.y) + (sw.z != w.z) + (sw.w != w.w);
    }
    float keep = 0.f;
#pragma unroll
    for (int j = 0; j < 14; ++j) { asm volatile("" : "+v"(c[j])); keep += c[j].x; }
    if (keep == 1.2345e-30f) nbad += 1000000;   // (keeps the 14 loads alive)
  }
  if (nbad) atomicAdd(bad, nbad);
  if (threadIdx.x == 0) atomicAdd(done, 1ull);
}

int main(int argc, char** argv) {
  const int mode = argc > 1 ? atoi(argv[1]) : 0;
  const int iters = argc > 2 ? atoi(argv[2]) : 2000;
  const size_t mib = argc > 3 ? (size_t)atol(argv[3]) : 2048;
  const int coalesced = argc > 4 ? atoi(argv[4]) : 0, squares = argc > 5 ? atoi(argv[5]) : 0;
  const size_t n = mib * (1u << 20) / 4;
  float* x = nullptr;
  if (hipMalloc(&x, n * 4) != hipSuccess) { printf("alloc failed\n"); return 2; }
  {  // distinct values everywhere (a stale register can then never equal the fresh value by accident)
    std::vector<float> h(1u << 22);
    for (size_t off = 0; off < n; off += h.size()) {
      const size_t m = n - off < h.size() ? n - off : h.size();
      for (size_t i = 0; i < m; ++i) h[i] = (float)((off + i) % 16777213u) + 0.5f;
      hipMemcpy(x + off, h.data(), m * 4, hipMemcpyHostToDevice);
    }
  }
  unsigned long long *bad, *done, hb = 0, hd = 0;
  hipMalloc(&bad, 8); hipMalloc(&done, 8);
  hipMemset(bad, 0, 8); hipMemset(done, 0, 8);
  const int blocks = 256 * 8;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(blocks), dim3(256), 0, 0, x, n / 4, iters, bad, done, coalesced, squares);
  else hipLaunchKernelGGL(probe<1>, dim3(blocks), dim3(256), 0, 0, x, n / 4, iters, bad, done, coalesced, squares);
  hipEventRecord(e1);
  if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 3; }
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(&hd, done, 8, hipMemcpyDeviceToHost);
  const double lanes = (double)blocks * 256 * iters;
  printf("mode %d (%s)%s%s: %.0f lane-rows, %llu values read before they had landed, %.1f ms, %.2f TB/s of 16-byte rows\n", mode,
         mode == 0 ? "dwordx3 at byte 4 + dwordx2 at byte 0" : "one dwordx4", coalesced ? ", lanes on consecutive rows" : "", squares == 2 ? ", destination registers overwritten behind the wait" : (squares ? ", squares behind the wait" : ""), lanes, hb, ms, lanes * 15 * 16 / (ms * 1e-3) / 1e12);
  return hb ? 1 : 0;
}

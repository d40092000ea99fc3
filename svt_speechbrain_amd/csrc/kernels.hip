// HBM-bound kernels of the path: statistics, norms, conv layer 0, operand re-layouts, softmax,
// frame head and decode.  All are wave64 designs: one wavefront per row where a row reduction is
// needed (shuffle reductions, no LDS), 16-byte accesses where the layout allows.
#include "common.h"

namespace svt {
namespace {

__device__ __forceinline__ float gelu_erf(float x) { return gelu_fast(x); }

template <typename V>
__device__ __forceinline__ V wave_sum(V v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ---- sums over the workgroups of a launch, reproducible bit for bit ----
// fp64 atomicAdd gives a sum whose last bits depend on the order in which the workgroups arrive; conv layer 0's GroupNorm variance is a
// quadratic form of the window moments with heavy cancellation (low-pass audio against random filters: 1e3-1e5), so a changed last bit
// of a moment can flip the last bit of an fp32 coefficient, and one flipped bf16 rounding downstream moves every logit by ~1e-4.  (This
// was the first suspect when the eight-rank dry run of bench.py --verify on one GPU began to fail; the forwards that differed there
// turned out to come from a split load in head_dots_kernel, below -- but arrival order was the one input of the forward that is not a
// function of its arguments, so it went too.)
// Instead every workgroup stores its partial sums and takes a ticket; the one that draws the last ticket adds the partials in workgroup
// order and writes the result (no zeroed accumulator needed; the ticket must be 0 at launch and is put back to 0).
// (No fence: an agent-scope release fence is an L2 write-back on this chip, and one per workgroup made the three statistics kernels
// 30-50 us slower each.  The partials are WRITE-THROUGH stores -- st_partial -- whose acknowledgement the vmcnt wait below awaits, the
// ticket is an agent-scope atomic behind it, and the reader uses loads that bypass the non-coherent caches -- ld_partial.)
__device__ __forceinline__ void st_partial(double* p, double v) {
  __hip_atomic_store((unsigned long long*)p, __builtin_bit_cast(unsigned long long, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// A/B arm (svt_debug_set key 32 = 1): the ticket as an agent-scope ACQUIRE-RELEASE atomic -- the form the HIP / LLVM memory
// model asks for (release: this workgroup's partials are visible before the ticket; acquire: the last workgroup sees everyone's).  The
// default (0) is the relaxed form described above, which leans on what gfx950 does with write-through stores and cache-bypassing loads;
// tests/test_gpu_statistics.py holds BOTH arms to host fp64 sums of the same inputs and to each other, bit for bit.
__device__ int d_ticket_fenced = 0;
__device__ __forceinline__ bool last_workgroup(unsigned* ticket, unsigned n_groups) {
  __shared__ unsigned s_last;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this thread's partials have reached the level every XCD reads from
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned t = d_ticket_fenced ? __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT)
                                       : __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = t == n_groups - 1 ? 1u : 0u;
    if (s_last) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  return s_last != 0;
}
// a partial written by another workgroup (possibly on another XCD): read past the non-coherent caches
__device__ __forceinline__ double ld_partial(const double* p) {
  return __builtin_bit_cast(double, __hip_atomic_load((const unsigned long long*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

template <typename T> __device__ __forceinline__ float ld(const T* p, long i);
template <> __device__ __forceinline__ float ld<float>(const float* p, long i) { return p[i]; }
template <> __device__ __forceinline__ float ld<bf16_t>(const bf16_t* p, long i) { return (float)p[i]; }
template <typename T> __device__ __forceinline__ void st(T* p, long i, float v);
template <> __device__ __forceinline__ void st<float>(float* p, long i, float v) { p[i] = v; }
template <> __device__ __forceinline__ void st<bf16_t>(bf16_t* p, long i, float v) { p[i] = (bf16_t)v; }

// ---- "pair rows" of the split-operand modes (GemmArgs::a_pairs): every 32 consecutive elements of a row are stored as
// [32 hi pieces | 32 lo pieces] (16-bit, IEEE half for kind 3 / bf16 for kind 2) in the 128 bytes of the fp32 slab they replace, so
// element offsets are those of the fp32 tensor.  PK = 0: no pair output, 2 = bf16 pieces, 3 = fp16 pieces (= svt_precision).
template <int PK> __device__ __forceinline__ void cut_piece(float x, unsigned short& hi, unsigned short& lo) {
  if constexpr (PK == 3) {
    const _Float16 a = (_Float16)x, b = (_Float16)(x - (float)a);
    hi = __builtin_bit_cast(unsigned short, a); lo = __builtin_bit_cast(unsigned short, b);
  } else {
    const __bf16 a = (__bf16)x, b = (__bf16)(x - (float)a);
    hi = __builtin_bit_cast(unsigned short, a); lo = __builtin_bit_cast(unsigned short, b);
  }
}
// N (4 or 8) consecutive elements starting at element index e (a multiple of N) of a pair-row tensor whose fp32 image starts at `base`
template <int PK, int N> __device__ __forceinline__ void store_pairs(void* base, int64_t e, const float (&v)[N]) {
  static_assert(N == 4 || N == 8, "pieces of 4 or 8 elements");
  unsigned short h[N], l[N];
#pragma unroll
  for (int i = 0; i < N; ++i) cut_piece<PK>(v[i], h[i], l[i]);
  char* d = (char*)base + (e >> 5) * 128 + (e & 31) * 2;
  if constexpr (N == 8) {
    *(uint4*)d = uint4{(unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16), (unsigned)h[4] | ((unsigned)h[5] << 16), (unsigned)h[6] | ((unsigned)h[7] << 16)};
    *(uint4*)(d + 64) = uint4{(unsigned)l[0] | ((unsigned)l[1] << 16), (unsigned)l[2] | ((unsigned)l[3] << 16), (unsigned)l[4] | ((unsigned)l[5] << 16), (unsigned)l[6] | ((unsigned)l[7] << 16)};
  } else {
    *(uint2*)d = uint2{(unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16)};
    *(uint2*)(d + 64) = uint2{(unsigned)l[0] | ((unsigned)l[1] << 16), (unsigned)l[2] | ((unsigned)l[3] << 16)};
  }
}

// ------------------------------------------------------------------------------------------------
__global__ void f32_to_bf16_kernel(const float* in, bf16_t* out, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) out[i] = (bf16_t)in[i];
}

// sum / sum of squares in fp64 (two whole-batch layer norms of the wrapper)
// blockIdx.y = norm group: group g covers x[g*n, (g+1)*n) and WRITES mom[2g], mom[2g+1] (the last workgroup of the group adds the
// per-workgroup partials `part[(g * gridDim.x + block) * 2 ..]` in block order: last_workgroup above)
__global__ __launch_bounds__(256) void moments_kernel(const float* x, int64_t n, double* mom, double* part, unsigned* ticket) {
  __shared__ double sh[2][4];
  x += (int64_t)blockIdx.y * n;
  mom += 2 * blockIdx.y;
  part += (size_t)blockIdx.y * gridDim.x * 2;
  double s = 0.0, ss = 0.0;
  const int64_t n4 = n >> 2;
  const float4* x4 = (const float4*)x;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n4; i += stride) {
    const float4 v = x4[i];
    // fp32 partial of 4 terms, then fp64: keeps the stream cheap and the long sum exact enough
    const float a = (v.x + v.y) + (v.z + v.w);
    const float b = (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    s += (double)a;
    ss += (double)b;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    for (int64_t j = n4 << 2; j < n; ++j) { s += (double)x[j]; ss += (double)x[j] * (double)x[j]; }
  }
  s = wave_sum(s);
  ss = wave_sum(ss);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { sh[0][wave] = s; sh[1][wave] = ss; }
  __syncthreads();
  if (threadIdx.x == 0) {
    st_partial(part + blockIdx.x * 2, sh[0][0] + sh[0][1] + sh[0][2] + sh[0][3]);
    st_partial(part + blockIdx.x * 2 + 1, sh[1][0] + sh[1][1] + sh[1][2] + sh[1][3]);
  }
  if (!last_workgroup(ticket + blockIdx.y, gridDim.x)) return;
  s = 0.0; ss = 0.0;
  for (unsigned b = threadIdx.x; b < gridDim.x; b += 256) { s += ld_partial(part + b * 2); ss += ld_partial(part + b * 2 + 1); }
  s = wave_sum(s);
  ss = wave_sum(ss);
  if (lane == 0) { sh[0][wave] = s; sh[1][wave] = ss; }
  __syncthreads();
  if (threadIdx.x == 0) {
    mom[0] = sh[0][0] + sh[0][1] + sh[0][2] + sh[0][3];
    mom[1] = sh[1][0] + sh[1][1] + sh[1][2] + sh[1][3];
  }
}

__global__ void global_norm_kernel(const float* x, float* y, int64_t n, const double* mom, float eps, double n_stat) {
  x += (int64_t)blockIdx.y * n;
  y += (int64_t)blockIdx.y * n;
  mom += 2 * blockIdx.y;
  const double mean = mom[0] / n_stat;   // n_stat: the elements the moments were summed over (= n, or the global count when
  const double var = mom[1] / n_stat - mean * mean;   // the sums were reduced over ranks: svt_encoder_set_norm_reduce)
  const float mu = (float)mean;
  const float rs = (float)(1.0 / sqrt(var + (double)eps));
  const int64_t n4 = n >> 2;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n4; i += stride) {
    float4 v = ((const float4*)x)[i];
    v.x = (v.x - mu) * rs; v.y = (v.y - mu) * rs; v.z = (v.z - mu) * rs; v.w = (v.w - mu) * rs;
    ((float4*)y)[i] = v;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0)
    for (int64_t j = n4 << 2; j < n; ++j) y[j] = (x[j] - mu) * rs;
}

// ------------------------------------------------------------------------------------------------
// Row LayerNorm: one wave per row, two-pass statistics from registers-free re-reads (rows <= 4 KB
// stay in L1/L2).  Optional exact-erf GELU (conv "layer" mode).
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void layernorm_kernel(const TI* x, const float* add, float* sumF, int64_t rows, int D,
                                                        const float* gamma, const float* beta, float eps, int gelu,
                                                        TO* yT, float* yF) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const TI* xr = x + row * D;
  const float* ar = add ? add + row * D : nullptr;
  float s = 0.f;
  for (int i = lane; i < D; i += 64) s += ld<TI>(xr, i) + (ar ? ar[i] : 0.f);
  const float mean = wave_sum(s) / (float)D;
  float q = 0.f;
  for (int i = lane; i < D; i += 64) { const float d = ld<TI>(xr, i) + (ar ? ar[i] : 0.f) - mean; q += d * d; }
  const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
  for (int i = lane; i < D; i += 64) {
    const float xv = ld<TI>(xr, i) + (ar ? ar[i] : 0.f);
    if (sumF) sumF[row * D + i] = xv;
    float v = (xv - mean) * rstd * gamma[i] + beta[i];
    if (gelu) v = gelu_erf(v);
    if (yT) st<TO>(yT, row * D + i, v);
    if (yF) yF[row * D + i] = v;
  }
}

// Register-resident variant for D = 64*VPT (512 / 768 / 1024): the row is read ONCE with 16-byte loads
// (fp32 in) and kept in VPT registers per lane; statistics by two in-register passes + wave shuffles.
// PK != 0: additionally (or only) writes the result as pair rows into yP (the next split-operand product's operand)
template <int VPT, typename TO, typename TI = float, int PK = 0>
__global__ __launch_bounds__(256) void layernorm_f32_vec_kernel(const TI* x, const float* add, float* sumF,
                                                                int64_t rows, const float* gamma, const float* beta,
                                                                float eps, int gelu, TO* yT, float* yF, void* yP = nullptr,
                                                                const void* addP = nullptr) {
  constexpr int D = 64 * VPT, NV = VPT / 4;
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float v[VPT];
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    float4 t;
    if constexpr (sizeof(TI) == 4) {
      t = ((const float4*)(x + row * D))[lane + 64 * j];
    } else {
      const bf16x4 tb = ((const bf16x4*)(x + row * D))[lane + 64 * j];
      t = float4{(float)tb[0], (float)tb[1], (float)tb[2], (float)tb[3]};
    }
    if (add) {
      const float4 a = ((const float4*)(add + row * D))[lane + 64 * j];
      t.x += a.x; t.y += a.y; t.z += a.z; t.w += a.w;
    }
    if constexpr (PK != 0) {
      if (addP) {   // the addend as pair rows (the post-LN residual stream of the split modes: hi + lo = 22 mantissa bits)
        const int64_t e = row * D + (lane + 64 * j) * 4;
        const char* pp = (const char*)addP + (e >> 5) * 128 + (e & 31) * 2;
        const uint2 h = *(const uint2*)pp, l = *(const uint2*)(pp + 64);
        const unsigned hw[2] = {h.x, h.y}, lw[2] = {l.x, l.y};
        float a[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const unsigned short hs = (unsigned short)(hw[i >> 1] >> (16 * (i & 1))), ls = (unsigned short)(lw[i >> 1] >> (16 * (i & 1)));
          if constexpr (PK == 3) a[i] = (float)__builtin_bit_cast(_Float16, hs) + (float)__builtin_bit_cast(_Float16, ls);
          else a[i] = (float)__builtin_bit_cast(__bf16, hs) + (float)__builtin_bit_cast(__bf16, ls);
        }
        t.x += a[0]; t.y += a[1]; t.z += a[2]; t.w += a[3];
      }
    }
    if (sumF) ((float4*)(sumF + row * D))[lane + 64 * j] = t;
    v[4 * j] = t.x; v[4 * j + 1] = t.y; v[4 * j + 2] = t.z; v[4 * j + 3] = t.w;
    s += (t.x + t.y) + (t.z + t.w);
  }
  const float mean = wave_sum(s) * (1.f / D);
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < VPT; ++i) { v[i] -= mean; q = fmaf(v[i], v[i], q); }
  const float rstd = rsqrtf(wave_sum(q) * (1.f / D) + eps);
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int c = (lane + 64 * j) * 4;
    const float4 g = *(const float4*)(gamma + c), b = *(const float4*)(beta + c);
    float o0 = fmaf(v[4 * j] * rstd, g.x, b.x), o1 = fmaf(v[4 * j + 1] * rstd, g.y, b.y);
    float o2 = fmaf(v[4 * j + 2] * rstd, g.z, b.z), o3 = fmaf(v[4 * j + 3] * rstd, g.w, b.w);
    if (gelu) {
      if (sizeof(TO) == 2 && PK == 0 && !yF) {
        // results stored ONLY in 16 bits (the conv stack of the layer-norm extractor in the throughput modes): the polynomial GELU of the GEMM
        // epilogues (common.h) -- with the exact-erf form (rcp + exp: quarter-rate instructions) this pass ran at 3.7 TB/s, VALU-bound
        f32x2_t ga = {o0, o1}, gb = {o2, o3};
        ga = gelu_bf16x2(ga); gb = gelu_bf16x2(gb);
        o0 = ga.x; o1 = ga.y; o2 = gb.x; o3 = gb.y;
      } else {
        o0 = gelu_erf(o0); o1 = gelu_erf(o1); o2 = gelu_erf(o2); o3 = gelu_erf(o3);
      }
    }
    if (yT) {
      if constexpr (sizeof(TO) == 2) {
        bf16x4 o;
        o[0] = (bf16_t)o0; o[1] = (bf16_t)o1; o[2] = (bf16_t)o2; o[3] = (bf16_t)o3;
        *(bf16x4*)((bf16_t*)yT + row * D + c) = o;
      } else {
        *(float4*)((float*)yT + row * D + c) = float4{o0, o1, o2, o3};
      }
    }
    if (yF) *(float4*)(yF + row * D + c) = float4{o0, o1, o2, o3};
    if constexpr (PK != 0) {
      const float ov[4] = {o0, o1, o2, o3};
      store_pairs<PK, 4>(yP, row * D + c, ov);
    }
  }
}

// Post-LN residual stream kept as a bf16 pair (hi = the operand copy the next GEMM reads anyway, lo = bf16(x - hi):
// 16 mantissa bits, 2^-17 relative) instead of an extra fp32 copy: y = LN(branch + hi + lo) -> (hi', lo').  Per
// element 6 bytes read + 4 written instead of 6 + 6 (the kernel is at the HBM roofline, so -17 % bytes = -17 % time).
// yF (optional) also receives the full fp32 result (last layer: the whole-batch output norm reads it).
template <int VPT>
__global__ __launch_bounds__(256) void layernorm_hilo_kernel(const bf16_t* branch, const bf16_t* rh, const bf16_t* rl,
                                                             int64_t rows, const float* gamma, const float* beta, float eps,
                                                             bf16_t* yh, bf16_t* yl, float* yF) {
  constexpr int D = 64 * VPT, NV = VPT / 4;
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float v[VPT];
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const long o = row * D + (lane + 64 * j) * 4;
    const bf16x4 h = *(const bf16x4*)(rh + o), l = *(const bf16x4*)(rl + o);
    float4 t = float4{(float)h[0] + (float)l[0], (float)h[1] + (float)l[1], (float)h[2] + (float)l[2], (float)h[3] + (float)l[3]};
    if (branch) {
      const bf16x4 a = *(const bf16x4*)(branch + o);
      t.x += (float)a[0]; t.y += (float)a[1]; t.z += (float)a[2]; t.w += (float)a[3];
    }
    v[4 * j] = t.x; v[4 * j + 1] = t.y; v[4 * j + 2] = t.z; v[4 * j + 3] = t.w;
    s += (t.x + t.y) + (t.z + t.w);
  }
  const float mean = wave_sum(s) * (1.f / D);
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < VPT; ++i) { v[i] -= mean; q = fmaf(v[i], v[i], q); }
  const float rstd = rsqrtf(wave_sum(q) * (1.f / D) + eps);
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int c = (lane + 64 * j) * 4;
    const float4 g = *(const float4*)(gamma + c), b = *(const float4*)(beta + c);
    const float o[4] = {fmaf(v[4 * j] * rstd, g.x, b.x), fmaf(v[4 * j + 1] * rstd, g.y, b.y),
                        fmaf(v[4 * j + 2] * rstd, g.z, b.z), fmaf(v[4 * j + 3] * rstd, g.w, b.w)};
    bf16x4 oh, ol;
#pragma unroll
    for (int i = 0; i < 4; ++i) { oh[i] = (bf16_t)o[i]; ol[i] = (bf16_t)(o[i] - (float)oh[i]); }
    *(bf16x4*)(yh + row * D + c) = oh;
    *(bf16x4*)(yl + row * D + c) = ol;
    if (yF) *(float4*)(yF + row * D + c) = float4{o[0], o[1], o[2], o[3]};
  }
}

// fp32 -> (hi, lo) bf16 pair with a LayerNorm in front (first LN of the post-LN encoder): x fp32 in
// LayerNorm (+ GELU) over rows stored in the 16-bit operand type, half a wave per row, 16-byte accesses: the conv stack of the
// layer-norm extractor in the throughput modes normalises the conv GEMM's output in place (2 bytes read + 2 written per element; with
// a wave per row and 8-byte accesses the pass ran at 3.8 TB/s).
template <int D>
__global__ __launch_bounds__(256) void layernorm_op16_rows2_kernel(const bf16_t* x, int64_t rows, const float* gamma, const float* beta,
                                                                   float eps, int gelu, bf16_t* y) {
  constexpr int NC = D / 256;
  const int lane = threadIdx.x & 63, sub = lane & 31;
  const int64_t row = (int64_t)blockIdx.x * 8 + (threadIdx.x >> 6) * 2 + (lane >> 5);
  if (row >= rows) return;
  float v[NC][8];
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < NC; ++j) {
    const bf16x8 h = *(const bf16x8*)(x + row * D + (sub + 32 * j) * 8);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[j][i] = (float)h[i];
    s += ((v[j][0] + v[j][1]) + (v[j][2] + v[j][3])) + ((v[j][4] + v[j][5]) + (v[j][6] + v[j][7]));
  }
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  const float mean = s * (1.f / D);
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < NC; ++j)
#pragma unroll
    for (int i = 0; i < 8; ++i) { v[j][i] -= mean; q = fmaf(v[j][i], v[j][i], q); }
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
  const float rstd = rsqrtf(q * (1.f / D) + eps);
#pragma unroll
  for (int j = 0; j < NC; ++j) {
    const int c = (sub + 32 * j) * 8;
    const float4 g0 = *(const float4*)(gamma + c), g1 = *(const float4*)(gamma + c + 4);
    const float4 b0 = *(const float4*)(beta + c), b1 = *(const float4*)(beta + c + 4);
    const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w}, bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
    f32x2_t o2[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) o2[i] = f32x2_t{fmaf(v[j][2 * i] * rstd, gg[2 * i], bb[2 * i]), fmaf(v[j][2 * i + 1] * rstd, gg[2 * i + 1], bb[2 * i + 1])};
    if (gelu) gelu_bf16x2_x4(o2);
    bf16x8 ob;
#pragma unroll
    for (int i = 0; i < 4; ++i) { ob[2 * i] = (bf16_t)o2[i].x; ob[2 * i + 1] = (bf16_t)o2[i].y; }
    *(bf16x8*)(y + row * D + c) = ob;
  }
}

// Two rows per wave (round 4): a HALF-wave owns a row and every access is 16 bytes (8 bf16 per lane), so a 768-wide row is three
// accesses per stream and lane instead of three 8-byte ones over twice the lanes, and the two reductions run over five exchange steps
// instead of six: 23.2 -> 20.5 us per pass at 32 x 10 s (HBM-bound: the same bytes at a higher achieved rate).  The sums are taken in a
// different order than in the one-row kernel, which moves near-tie frames of the bf16 mode (tests/test_gpu_parity.py
// check_16bit_mode_bound holds the mode to the operand-rounding simulation, not to one summation order).
// PARTS: the branch arrives as `nparts` fp32 partial products of a K-split GEMM (gemm_skinny.hip, ksplit) + its bias: they are added in
// part order, the bias last, and the sum is rounded to the operand type -- the value the un-split GEMM's epilogue would have stored
// (same rounding points as the 16-bit modes' stored-activation simulation, tools/sim_split.py)
template <int D, bool PARTS = false>
__global__ __launch_bounds__(256) void layernorm_hilo2_kernel(const bf16_t* branch, const bf16_t* rh, const bf16_t* rl,
                                                              int64_t rows, const float* gamma, const float* beta, float eps,
                                                              bf16_t* yh, bf16_t* yl, float* yF, const float* parts = nullptr, int nparts = 0,
                                                              long part_stride = 0, const float* pbias = nullptr) {
  constexpr int NC = D / 256;   // 16-byte chunks (8 elements) per lane: 32 lanes x NC x 8 = D
  const int lane = threadIdx.x & 63, sub = lane & 31;
  const int64_t row = (int64_t)blockIdx.x * 8 + (threadIdx.x >> 6) * 2 + (lane >> 5);
  if (row >= rows) return;
  float v[NC][8];
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < NC; ++j) {
    const long o = row * D + (sub + 32 * j) * 8;
    const bf16x8 h = *(const bf16x8*)(rh + o), l = *(const bf16x8*)(rl + o);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[j][i] = (float)h[i] + (float)l[i];
    if constexpr (PARTS) {
      float a[8];
      {
        const float4 t0 = *(const float4*)(parts + o), t1 = *(const float4*)(parts + o + 4);
        a[0] = t0.x; a[1] = t0.y; a[2] = t0.z; a[3] = t0.w; a[4] = t1.x; a[5] = t1.y; a[6] = t1.z; a[7] = t1.w;
      }
      for (int k = 1; k < nparts; ++k) {
        const float4 t0 = *(const float4*)(parts + k * part_stride + o), t1 = *(const float4*)(parts + k * part_stride + o + 4);
        a[0] += t0.x; a[1] += t0.y; a[2] += t0.z; a[3] += t0.w; a[4] += t1.x; a[5] += t1.y; a[6] += t1.z; a[7] += t1.w;
      }
      const int c = (sub + 32 * j) * 8;
      const float4 b0 = *(const float4*)(pbias + c), b1 = *(const float4*)(pbias + c + 4);
      const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
      for (int i = 0; i < 8; ++i) v[j][i] += (float)(bf16_t)(a[i] + bb[i]);
    } else if (branch) {
      const bf16x8 a = *(const bf16x8*)(branch + o);
#pragma unroll
      for (int i = 0; i < 8; ++i) v[j][i] += (float)a[i];
    }
    s += ((v[j][0] + v[j][1]) + (v[j][2] + v[j][3])) + ((v[j][4] + v[j][5]) + (v[j][6] + v[j][7]));
  }
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  const float mean = s * (1.f / D);
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < NC; ++j)
#pragma unroll
    for (int i = 0; i < 8; ++i) { v[j][i] -= mean; q = fmaf(v[j][i], v[j][i], q); }
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
  const float rstd = rsqrtf(q * (1.f / D) + eps);
#pragma unroll
  for (int j = 0; j < NC; ++j) {
    const int c = (sub + 32 * j) * 8;
    const float4 g0 = *(const float4*)(gamma + c), g1 = *(const float4*)(gamma + c + 4);
    const float4 b0 = *(const float4*)(beta + c), b1 = *(const float4*)(beta + c + 4);
    const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w}, bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
    float o[8];
    bf16x8 oh, ol;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      o[i] = fmaf(v[j][i] * rstd, gg[i], bb[i]);
      oh[i] = (bf16_t)o[i];
      ol[i] = (bf16_t)(o[i] - (float)oh[i]);
    }
    *(bf16x8*)(yh + row * D + c) = oh;
    *(bf16x8*)(yl + row * D + c) = ol;
    if (yF) {
      *(float4*)(yF + row * D + c) = float4{o[0], o[1], o[2], o[3]};
      *(float4*)(yF + row * D + c + 4) = float4{o[4], o[5], o[6], o[7]};
    }
  }
}
template <int VPT>
__global__ __launch_bounds__(256) void layernorm_f32_to_hilo_kernel(const float* x, int64_t rows, const float* gamma,
                                                                    const float* beta, float eps, bf16_t* yh, bf16_t* yl) {
  constexpr int D = 64 * VPT, NV = VPT / 4;
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float v[VPT];
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const float4 t = ((const float4*)(x + row * D))[lane + 64 * j];
    v[4 * j] = t.x; v[4 * j + 1] = t.y; v[4 * j + 2] = t.z; v[4 * j + 3] = t.w;
    s += (t.x + t.y) + (t.z + t.w);
  }
  const float mean = wave_sum(s) * (1.f / D);
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < VPT; ++i) { v[i] -= mean; q = fmaf(v[i], v[i], q); }
  const float rstd = rsqrtf(wave_sum(q) * (1.f / D) + eps);
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int c = (lane + 64 * j) * 4;
    const float4 g = *(const float4*)(gamma + c), b = *(const float4*)(beta + c);
    const float o[4] = {fmaf(v[4 * j] * rstd, g.x, b.x), fmaf(v[4 * j + 1] * rstd, g.y, b.y),
                        fmaf(v[4 * j + 2] * rstd, g.z, b.z), fmaf(v[4 * j + 3] * rstd, g.w, b.w)};
    bf16x4 oh, ol;
#pragma unroll
    for (int i = 0; i < 4; ++i) { oh[i] = (bf16_t)o[i]; ol[i] = (bf16_t)(o[i] - (float)oh[i]); }
    *(bf16x4*)(yh + row * D + c) = oh;
    *(bf16x4*)(yl + row * D + c) = ol;
  }
}

// ------------------------------------------------------------------------------------------------
// conv layer 0, "group" mode.  GroupNorm(C groups) needs per-(clip,channel) mean/var over ALL frames
// before the GELU.  Because Cin = 1, y[c,t] = w_c . window_t, so the statistics of all C channels
// follow exactly from the first and second moments of the 10-sample windows:
//   mean_c = w_c . E[win] ,  E[y_c^2] = w_c^T E[win win^T] w_c
// (65 fp64 sums per clip) — one cheap pass over the waveform instead of a second pass over the
// (B, T1, 512) activation.
constexpr int K0 = 10;
constexpr int NWM = K0 + K0 * (K0 + 1) / 2;  // 65

__global__ __launch_bounds__(256) void conv0_window_moments_kernel(const float* wav, int64_t L, int stride, int64_t T1,
                                                                   double* wm, double* part, unsigned* ticket) {
  __shared__ double sh[4][NWM];
  const int b = blockIdx.y;
  const float* x = wav + (int64_t)b * L;
  double acc[NWM];
#pragma unroll
  for (int i = 0; i < NWM; ++i) acc[i] = 0.0;
  constexpr int WPT = 8;  // windows per thread
  const int64_t t0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * WPT;
  for (int w = 0; w < WPT; ++w) {
    const int64_t t = t0 + w;
    if (t >= T1) break;
    float v[K0];
#pragma unroll
    for (int j = 0; j < K0; ++j) v[j] = x[t * stride + j];
    int idx = K0;
#pragma unroll
    for (int j = 0; j < K0; ++j) {
      acc[j] += (double)v[j];
#pragma unroll
      for (int j2 = j; j2 < K0; ++j2) acc[idx++] += (double)(v[j] * v[j2]);
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < NWM; ++i) {
    const double r = wave_sum(acc[i]);
    if (lane == 0) sh[wave][i] = r;
  }
  __syncthreads();
  // per-workgroup partials, added in workgroup order by the clip's last workgroup (last_workgroup above): wm is written, not accumulated
  part += (size_t)b * gridDim.x * NWM;
  if (threadIdx.x < NWM)
    st_partial(part + blockIdx.x * NWM + threadIdx.x, sh[0][threadIdx.x] + sh[1][threadIdx.x] + sh[2][threadIdx.x] + sh[3][threadIdx.x]);
  if (!last_workgroup(ticket + b, gridDim.x)) return;
  if (threadIdx.x < NWM) {
    // (eight loads in flight, added in workgroup order: one load per trip made the loop a chain of memory round trips)
    double t = 0.0;
    unsigned k = 0;
    for (; k + 8 <= gridDim.x; k += 8) {
      double v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = ld_partial(part + (k + j) * NWM + threadIdx.x);
#pragma unroll
      for (int j = 0; j < 8; ++j) t += v[j];
    }
    for (; k < gridDim.x; ++k) t += ld_partial(part + k * NWM + threadIdx.x);
    wm[b * NWM + threadIdx.x] = t;
  }
}

__global__ void conv0_group_coef_kernel(const double* wav_mom, int64_t n_wav, const double* wm, int64_t T1, int C,
                                        const float* w0, const float* b0, const float* gamma, const float* beta,
                                        float eps_wav, float eps_gn, float* coef, int cpg) {
  const int b = blockIdx.y;
  if (wav_mom) wav_mom += 2 * (b / cpg);  // the waveform norm group of this clip
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double mu = 0.0, r = 1.0;
  if (wav_mom) {
    mu = wav_mom[0] / (double)n_wav;
    const double var = wav_mom[1] / (double)n_wav - mu * mu;
    r = 1.0 / sqrt(var + (double)eps_wav);
  }
  const double* m = wm + b * NWM;
  const double invT = 1.0 / (double)T1;
  double w[K0], mn[K0];
  double wsum = 0.0;
  for (int j = 0; j < K0; ++j) { w[j] = (double)w0[c * K0 + j]; wsum += w[j]; mn[j] = r * (m[j] * invT - mu); }
  const double bias = b0 ? (double)b0[c] : 0.0;
  double dot = 0.0;
  for (int j = 0; j < K0; ++j) dot += w[j] * mn[j];
  double quad = 0.0;
  int idx = K0;
  for (int j = 0; j < K0; ++j)
    for (int j2 = j; j2 < K0; ++j2) {
      const double R = r * r * (m[idx] * invT - mu * m[j] * invT - mu * m[j2] * invT + mu * mu);
      quad += (j == j2 ? 1.0 : 2.0) * w[j] * w[j2] * R;
      ++idx;
    }
  const double mean = dot + bias;
  const double e2 = quad + 2.0 * bias * dot + bias * bias;
  double var = e2 - mean * mean;
  if (var < 0.0) var = 0.0;
  const double a = (double)gamma[c] / sqrt(var + (double)eps_gn);
  const double shift = (double)beta[c] - mean * a;
  float* o = coef + ((int64_t)b * C + c) * (K0 + 1);
  for (int j = 0; j < K0; ++j) o[j] = (float)(a * r * w[j]);
  o[K0] = (float)(a * (bias - r * mu * wsum) + shift);
}

// out[b,t,c] = gelu( sum_j coef[b,c,j] * wav[b, t*stride + j] + coef[b,c,K0] ), channels-last.
// One wave writes whole (b,t) rows: lane = 8 consecutive channels -> 16 B (bf16) / 32 B (fp32) per lane.
template <typename TO, int PK = 0>   // PK != 0 (TO = float): the output is written as pair rows
__global__ __launch_bounds__(256) void conv0_group_apply_kernel(const float* wav, int64_t L, int stride, int64_t T1,
                                                                int C, const float* coef, TO* out) {
  constexpr int FPW = 32;  // frames per wave
  __shared__ float xs[4][FPW * 5 + 16];  // stride <= 5 supported by this tile size
  const int b = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t t0 = ((int64_t)blockIdx.x * 4 + wave) * FPW;
  if (t0 >= T1) return;
  const float* x = wav + (int64_t)b * L;
  const int nfr = (int)((T1 - t0 < FPW) ? (T1 - t0) : FPW);
  const int ns = (nfr - 1) * stride + K0;
  for (int i = lane; i < ns; i += 64) xs[wave][i] = x[t0 * stride + i];
  // (wave-private LDS region: no block barrier needed, but the wave must see its own writes)
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  const int c0 = lane * 8;
  if (c0 >= C) return;
  float cf[8][K0 + 1];
  const float* cp = coef + ((int64_t)b * C + c0) * (K0 + 1);
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j <= K0; ++j) cf[i][j] = cp[i * (K0 + 1) + j];
  for (int f = 0; f < nfr; ++f) {
    float xv[K0];
#pragma unroll
    for (int j = 0; j < K0; ++j) xv[j] = xs[wave][f * stride + j];
    float o[8];
    f32x2_t a4[4];
#pragma unroll
    for (int i = 0; i < 8; i += 2) {
      f32x2_t a = {cf[i][K0], cf[i + 1][K0]};
#pragma unroll
      for (int j = 0; j < K0; ++j) a = f32x2_t{cf[i][j], cf[i + 1][j]} * xv[j] + a;
      a4[i >> 1] = a;
    }
    // the kernel is VALU-bound on the GELU (10 FMAs vs ~30 issue slots of erf per channel pair): results stored as
    // bf16 take the polynomial form (no transcendental slots, four chains interleaved), fp32 results the 1.5e-7 form
    if constexpr (sizeof(TO) == 2) {
      gelu_bf16x2_x4(a4);
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) a4[i] = gelu_fast2(a4[i]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) { o[2 * i] = a4[i].x; o[2 * i + 1] = a4[i].y; }
    TO* dst = out + ((int64_t)b * T1 + t0 + f) * C + c0;
    if constexpr (PK != 0) {
      store_pairs<PK, 8>(out, ((int64_t)b * T1 + t0 + f) * C + c0, o);
    } else if constexpr (sizeof(TO) == 2) {
      bf16x8 v;
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = (bf16_t)o[i];
      *(bf16x8*)dst = v;
    } else {
      *(float4*)dst = float4{o[0], o[1], o[2], o[3]};
      *(float4*)(dst + 4) = float4{o[4], o[5], o[6], o[7]};
    }
  }
}

// conv layer 0, "layer" mode: conv (+bias) -> LayerNorm over C -> GELU, one wave per frame.
template <typename TO, int PK = 0>
__global__ __launch_bounds__(256) void conv0_layer_kernel(const float* wav, int64_t L, int stride, int64_t T1, int C,
                                                          const double* wav_mom, int64_t n_wav, float eps_wav,
                                                          const float* w0, const float* b0, const float* gamma,
                                                          const float* beta, float eps, TO* out, int cpg) {
  constexpr int FPW = 16;
  const int b = blockIdx.y;
  if (wav_mom) wav_mom += 2 * (b / cpg);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t t0 = ((int64_t)blockIdx.x * 4 + wave) * FPW;
  if (t0 >= T1) return;
  float mu = 0.f, r = 1.f;
  if (wav_mom) {
    const double m = wav_mom[0] / (double)n_wav;
    const double var = wav_mom[1] / (double)n_wav - m * m;
    mu = (float)m;
    r = (float)(1.0 / sqrt(var + (double)eps_wav));
  }
  const float* x = wav + (int64_t)b * L;
  const int c0 = lane * 8;
  const bool active = c0 < C;
  float w[8][K0], bb[8], g[8], be[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = active ? c0 + i : 0;
#pragma unroll
    for (int j = 0; j < K0; ++j) w[i][j] = w0[c * K0 + j];
    bb[i] = b0 ? b0[c] : 0.f;
    g[i] = gamma[c];
    be[i] = beta[c];
  }
  const int nfr = (int)((T1 - t0 < FPW) ? (T1 - t0) : FPW);
  // the wave's normalised samples staged once in a wave-private LDS strip (was: ten broadcast global loads per frame)
  __shared__ float xs[4][FPW * 5 + 16];  // stride <= 5
  {
    const int ns = (nfr - 1) * stride + K0;
    for (int i = lane; i < ns; i += 64) xs[wave][i] = (x[t0 * stride + i] - mu) * r;
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  }
  // two frames per iteration: the two wave-wide reductions of a frame (mean, variance) are dependent shuffle chains of ~150 cycles
  // each; with a second, independent frame in flight the vector ALU has work while they run (hubert-large 64 x 10 s: 1 021 -> 981 us;
  // what remains is the arithmetic itself: ~150 vector instructions per frame and lane)
  for (int f0 = 0; f0 < nfr; f0 += 2) {
    const int fr[2] = {f0, f0 + 1 < nfr ? f0 + 1 : f0};
    float y[2][8], s[2] = {0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      float xv[K0];
#pragma unroll
      for (int j = 0; j < K0; ++j) xv[j] = xs[wave][fr[u] * stride + j];
#pragma unroll
      for (int i = 0; i < 8; i += 2) {  // channel pairs on packed fp32 math
        f32x2_t a = {bb[i], bb[i + 1]};
#pragma unroll
        for (int j = 0; j < K0; ++j) a = f32x2_t{w[i][j], w[i + 1][j]} * xv[j] + a;
        y[u][i] = a.x;
        y[u][i + 1] = a.y;
        if (active) s[u] += a.x + a.y;
      }
    }
    float mean[2], q[2] = {0.f, 0.f}, rstd[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) mean[u] = wave_sum(s[u]) / (float)C;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int i = 0; i < 8; ++i) { const float d = y[u][i] - mean[u]; if (active) q[u] += d * d; }
#pragma unroll
    for (int u = 0; u < 2; ++u) rstd[u] = rsqrtf(wave_sum(q[u]) / (float)C + eps);
    if (!active) continue;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (u == 1 && fr[1] == fr[0]) break;   // odd tail: the second frame is a repeat of the first
      float o[8];
      if constexpr (sizeof(TO) == 2) {
        // bf16 result: polynomial GELU, four pair-chains interleaved (common.h)
        f32x2_t a4[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
          a4[i] = f32x2_t{(y[u][2 * i] - mean[u]) * rstd[u] * g[2 * i] + be[2 * i], (y[u][2 * i + 1] - mean[u]) * rstd[u] * g[2 * i + 1] + be[2 * i + 1]};
        gelu_bf16x2_x4(a4);
#pragma unroll
        for (int i = 0; i < 4; ++i) { o[2 * i] = a4[i].x; o[2 * i + 1] = a4[i].y; }
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = gelu_erf((y[u][i] - mean[u]) * rstd[u] * g[i] + be[i]);
      }
      TO* dst = out + ((int64_t)b * T1 + t0 + fr[u]) * C + c0;
      if constexpr (PK != 0) {
        store_pairs<PK, 8>(out, ((int64_t)b * T1 + t0 + fr[u]) * C + c0, o);
      } else if constexpr (sizeof(TO) == 2) {
        bf16x8 v;
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (bf16_t)o[i];
        *(bf16x8*)dst = v;
      } else {
        *(float4*)dst = float4{o[0], o[1], o[2], o[3]};
        *(float4*)(dst + 4) = float4{o[4], o[5], o[6], o[7]};
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
template <typename TO>
__global__ void posconv_gather_kernel(const float* h, int B, int T, int D, int G, int kp, int Tp, TO* out, const float* sc, const float* sh) {
  const int cg = D / G;
  const int64_t n = (int64_t)B * G * Tp * cg;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    const int ci = (int)(i % cg);
    int64_t r = i / cg;
    const int tp = (int)(r % Tp); r /= Tp;
    const int g = (int)(r % G);
    const int b = (int)(r / G);
    const int t = tp - kp / 2;
    float v = (t >= 0 && t < T) ? h[((int64_t)b * T + t) * D + g * cg + ci] : 0.f;
    if (sc && t >= 0 && t < T) v = fmaf(v, sc[g * cg + ci], sh[g * cg + ci]);  // eval-mode BatchNorm1d in front of the conv: the zero padding stays zero
    st<TO>(out, i, v);
  }
}

// bf16 output, 8 channels (16 bytes out, 32 bytes in) per thread
__global__ void posconv_gather_bf16x8_kernel(const float* h, int B, int T, int D, int G, int kp, int Tp, bf16_t* out, const float* sc, const float* sh) {
  const int cg = D / G, c8 = cg / 8;
  const int64_t n = (int64_t)B * G * Tp * c8;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int ci = (int)(i % c8) * 8;
    int64_t r = i / c8;
    const int tp = (int)(r % Tp); r /= Tp;
    const int g = (int)(r % G);
    const int b = (int)(r / G);
    const int t = tp - kp / 2;
    bf16x8 o;
    if (t >= 0 && t < T) {
      const float* src = h + ((int64_t)b * T + t) * D + g * cg + ci;
      float4 a = *(const float4*)src, c = *(const float4*)(src + 4);
      if (sc) {
        const float4 s0 = *(const float4*)(sc + g * cg + ci), s1 = *(const float4*)(sc + g * cg + ci + 4);
        const float4 t0 = *(const float4*)(sh + g * cg + ci), t1 = *(const float4*)(sh + g * cg + ci + 4);
        a = float4{fmaf(a.x, s0.x, t0.x), fmaf(a.y, s0.y, t0.y), fmaf(a.z, s0.z, t0.z), fmaf(a.w, s0.w, t0.w)};
        c = float4{fmaf(c.x, s1.x, t1.x), fmaf(c.y, s1.y, t1.y), fmaf(c.z, s1.z, t1.z), fmaf(c.w, s1.w, t1.w)};
      }
      o[0] = (bf16_t)a.x; o[1] = (bf16_t)a.y; o[2] = (bf16_t)a.z; o[3] = (bf16_t)a.w;
      o[4] = (bf16_t)c.x; o[5] = (bf16_t)c.y; o[6] = (bf16_t)c.z; o[7] = (bf16_t)c.w;
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (bf16_t)0.f;
    }
    *(bf16x8*)(out + (((int64_t)b * G + g) * Tp + tp) * cg + ci) = o;
  }
}

template <typename TO>
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* S, int64_t rows, int T, int Tp, TO* P) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* s = S + row * Tp;
  float mx = -INFINITY;
  for (int i = lane; i < T; i += 64) mx = fmaxf(mx, s[i]);
  mx = wave_max(mx);
  float sum = 0.f;
  for (int i = lane; i < T; i += 64) sum += expf(s[i] - mx);
  const float inv = 1.f / wave_sum(sum);
  for (int i = lane; i < Tp; i += 64) st<TO>(P, row * Tp + i, i < T ? expf(s[i] - mx) * inv : 0.f);
}

// (B*T, ld)[.., voff + h*dh + d] -> Vt (B, H, dh, Tp); 32x32 LDS tile transpose, zero padded keys
template <typename TV>
__global__ __launch_bounds__(256) void transpose_v_kernel(const TV* qkv, int T, int H, int dh, long ldq, long voff,
                                                          int Tp, TV* Vt) {
  __shared__ float tile[32][33];
  const int bh = blockIdx.z;
  const int b = bh / H, h = bh % H;
  const int t0 = blockIdx.x * 32, d0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int i = ty; i < 32; i += 8) {
    const int t = t0 + i, d = d0 + tx;
    tile[i][tx] = (t < T && d < dh) ? ld<TV>(qkv, ((long)b * T + t) * ldq + voff + (long)h * dh + d) : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int d = d0 + i, t = t0 + tx;
    if (d < dh && t < Tp) st<TV>(Vt, (((long)b * H + h) * dh + d) * Tp + t, tile[tx][i]);
  }
}

template <typename TX>
__global__ void axpby_kernel(const TX* x, const TX* y, float a, float b, TX* out, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) st<TX>(out, i, a * ld<TX>(x, i) + b * ld<TX>(y, i));
}

template <typename TO>
__global__ void add_pe_kernel(const float* x, int B, int T, int Tsrc, int D, const float* pe, float* outF, TO* outT) {
  const int64_t n = (int64_t)B * T * D;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    const int d = (int)(i % D);
    const int64_t r = i / D;
    const int t = (int)(r % T);
    const int b = (int)(r / T);
    const float xv = t < Tsrc ? x[((int64_t)b * Tsrc + t) * D + d] : 0.f;
    const float v = xv + pe[(int64_t)t * D + d];
    outF[i] = v;
    if (outT) st<TO>(outT, i, v);
  }
}

__global__ void add_f32_kernel(const float* a, const float* b, float* out, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) out[i] = a[i] + b[i];
}

// ------------------------------------------------------------------------------------------------
// Frame head: y[row, n] = x[row,:] . w[n,:] + b[n], N <= 32, fp32 throughout.  One wave per row; the
// row of x is read once, the N partial sums live in registers, shuffle-reduced at the end.
__global__ __launch_bounds__(256) void linear_small_kernel(const float* x, int64_t rows, int K, const float* w,
                                                           const float* b, int N, float* y) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + row * K;
  float acc[32];
#pragma unroll
  for (int n = 0; n < 32; ++n) acc[n] = 0.f;
  for (int k = lane; k < K; k += 64) {
    const float xv = xr[k];
#pragma unroll
    for (int n = 0; n < 32; ++n)
      if (n < N) acc[n] = fmaf(xv, w[(long)n * K + k], acc[n]);
  }
#pragma unroll
  for (int n = 0; n < 32; ++n) {
    if (n < N) {
      const float r = wave_sum(acc[n]);
      if (lane == 0) y[row * N + n] = r + (b ? b[n] : 0.f);
    }
  }
}

// Frame head for the encoder widths (K = 256 * KC): the N x K weight is staged once per workgroup in LDS, a wave
// handles four rows at a time (x read once from HBM with 16-byte accesses, every weight fragment reused by the four
// rows), and the 4 x N partial sums are folded across the wave with a halving exchange (7 shuffles per output column
// instead of 24).  HBM-bound: rows * K * 4 bytes in, rows * N * 4 bytes out.
template <int KC>
__global__ __launch_bounds__(256) void linear_head_kernel(const float* __restrict__ x, int64_t rows,
                                                          const float* __restrict__ w, const float* __restrict__ b, int N,
                                                          float* __restrict__ y) {
  constexpr int K = KC * 256;
  extern __shared__ __attribute__((aligned(16))) float wl[];
  for (int i = threadIdx.x * 4; i < N * K; i += 1024) *(float4*)(wl + i) = *(const float4*)(w + i);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool hi32 = (lane & 32) != 0, hi16 = (lane & 16) != 0;
  const int myrow = (hi32 ? 2 : 0) + (hi16 ? 1 : 0);
  for (int64_t r0 = ((int64_t)blockIdx.x * 4 + wave) * 4; r0 < rows; r0 += (int64_t)gridDim.x * 16) {
    float4 xv[4][KC];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int64_t row = r0 + rr < rows ? r0 + rr : rows - 1;
#pragma unroll
      for (int c = 0; c < KC; ++c) xv[rr][c] = *(const float4*)(x + row * K + c * 256 + lane * 4);
    }
    float out0 = 0.f, out1 = 0.f;
    for (int n = 0; n < N; ++n) {
      float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < KC; ++c) {
        const float4 wv = *(const float4*)(wl + n * K + c * 256 + lane * 4);
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          s[rr] = fmaf(xv[rr][c].x, wv.x, s[rr]);
          s[rr] = fmaf(xv[rr][c].y, wv.y, s[rr]);
          s[rr] = fmaf(xv[rr][c].z, wv.z, s[rr]);
          s[rr] = fmaf(xv[rr][c].w, wv.w, s[rr]);
        }
      }
      // lanes 0-31 end up with rows {0,1}, lanes 32-63 with rows {2,3}; then bit 4 of the lane picks the row
      float k0 = hi32 ? s[2] : s[0], k1 = hi32 ? s[3] : s[1];
      const float g0 = hi32 ? s[0] : s[2], g1 = hi32 ? s[1] : s[3];
      k0 += __shfl_xor(g0, 32, 64);
      k1 += __shfl_xor(g1, 32, 64);
      float v = hi16 ? k1 : k0;
      v += __shfl_xor(hi16 ? k0 : k1, 16, 64);
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
      if ((lane & 15) == (n & 15)) { if (n < 16) out0 = v; else out1 = v; }
    }
    const int64_t row = r0 + myrow;
    if (row < rows) {
      const int n0 = lane & 15;
      if (n0 < N) y[row * N + n0] = out0 + (b ? b[n0] : 0.f);
      if (n0 + 16 < N) y[row * N + n0 + 16] = out1 + (b ? b[n0 + 16] : 0.f);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Fused tail of the audio recipes ("fuse with a2's out-LN and a9"): the wrapper's whole-batch output norm
// (huggingface_interface.py:294-295), the 20-way frame head (speechbrain/nnet/linear.py:63-76) and the per-frame
// sigmoid / argmax (train_audio_ssl.py:93-100) WITHOUT materialising the normalised features.  The head is linear, so
//     logits[r][j] = ((x[r] - mu) * rs) . w[j] + b[j] = (x[r] . w[j] - mu * sum_k w[j][k]) * rs + b[j]
// and the raw dots x[r] . w[j] can be taken in the same single pass over the un-normalised encoder output that sums the
// two moments of the norm (per group of rows: clips_per_norm_group).  Pass 1 = head_dots_kernel (the frame-head kernel
// plus fp64 row-sum accumulation); pass 2 = head_finish_kernel over rows x N values: affine fix-up, logits, decode.
// Replaces moments + global_norm + linear_head + decode_frames (three reads and one write of the 49 MB feature tensor
// at 32 x 10 s) by one read.
template <int KC>
__global__ __launch_bounds__(256) void head_dots_kernel(const float* __restrict__ x, int64_t rows, const float* __restrict__ w, int N,
                                                        float* __restrict__ dots, double* __restrict__ mom, int64_t rows_per_group,
                                                        double* __restrict__ slots, unsigned* __restrict__ ticket, int slots_per_group) {
  constexpr int K = KC * 256;
  extern __shared__ __attribute__((aligned(16))) float wl[];
  for (int i = threadIdx.x * 4; i < N * K; i += 1024) *(float4*)(wl + i) = *(const float4*)(w + i);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool hi32 = (lane & 32) != 0, hi16 = (lane & 16) != 0;
  const int myrow = (hi32 ? 2 : 0) + (hi16 ? 1 : 0);
  // per-lane running (sum, sum of squares) of the norm group this wave is in: folded across the wave and stored when the wave's rows
  // cross into another group (and at the end) -- no shuffles and no division per row.  The statistics are reproducible bit for bit
  // (last_workgroup above): wave W = 4 blockIdx.x + wave owns the 4-row chunks W, W + NW, W + 2 NW ... (NW = waves of the launch), so the
  // waves that meet norm group g are those of the chunks ja = first row of g / 4 ... jb = last row of g / 4, taken modulo NW; a wave stores
  // its (sum, sum of squares) for g ONCE, in slot (g, (W - ja) mod NW) -- every slot 0 .. min(NW, jb - ja + 1) - 1 of a group is written by
  // exactly one wave, none needs zeroing -- and the launch's last workgroup adds each group's slots in slot order.
  const int NW = (int)gridDim.x * 4, W = (int)blockIdx.x * 4 + wave;
  int64_t cur_g = -1, g_end = 0;
  float la = 0.f, lq = 0.f;
  auto flush = [&]() {
    if (cur_g < 0) return;
    const double sa = wave_sum((double)la), sq = wave_sum((double)lq);
    if (lane == 0) {
      const int64_t ja = (cur_g * rows_per_group) >> 2;
      const int p = (int)(((W - ja) % NW + NW) % NW);
      double* sl = slots + ((size_t)cur_g * slots_per_group + p) * 2;
      st_partial(sl, sa);
      st_partial(sl + 1, sq);
    }
    la = 0.f; lq = 0.f;
  };
  for (int64_t r0 = ((int64_t)blockIdx.x * 4 + wave) * 4; r0 < rows; r0 += (int64_t)gridDim.x * 16) {
    float4 xv[4][KC];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int64_t row = r0 + rr < rows ? r0 + rr : rows - 1;
      // 16-byte loads the compiler cannot take apart (raw buffer loads).  Round 4 wrote `*(const float4*)`; hipcc split the first chunk of
      // every row into an OVERLAPPING pair -- global_load_dwordx3 at byte 4 + global_load_dwordx2 at byte 0 -- and placed its own counted
      // `s_waitcnt vmcnt(14 / 13 / 12 ...)` in front of the packed squares.  With other processes loading the same GPU the sum of squares
      // then came out short by 12-35 elements' worth in ~0.15 % of the forwards (sum and dots intact; every logit moved by 2e-4 .. 1e-3
      // through the output norm): tools/determinism_stress.py, 10 of 6 400 forwards.  Three rebuilds, 6 400 forwards each, all clean: a
      // full wait in front of the statistics; scalar FMAs; and this one -- the same packed arithmetic and the same counted waits behind
      // UNSPLIT loads.  WHICH instruction of the round-4 sequence misbehaves is not isolated: a microbenchmark of the pair, the counted wait
      // and reads / overwrites of the destination registers right behind it (tools/microbench/split_load_probe.hip) saw no stale value in
      // 2e10 lane-rows under the same eight-process load.  The kernel-level evidence stands; tests/test_build_isa.py keeps split pairs out.
      const auto xrs = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, 0x7FFFFFF0, 0x00020000);
#pragma unroll
      for (int c = 0; c < KC; ++c) {
        typedef float f4v __attribute__((ext_vector_type(4)));
        const f4v t = __builtin_bit_cast(f4v, __builtin_amdgcn_raw_buffer_load_b128(xrs, (unsigned)((row * K + c * 256 + lane * 4) * 4), 0, 0));
        xv[rr][c] = float4{t[0], t[1], t[2], t[3]};
      }
    }
    if (mom) {
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        if (r0 + rr >= rows) break;  // wave-uniform
        if (r0 + rr >= g_end || cur_g < 0) {  // rare: first row of the wave, or a group boundary
          flush();
          cur_g = (r0 + rr) / rows_per_group;
          g_end = (cur_g + 1) * rows_per_group;
        }
        float a = 0.f, q = 0.f;
#pragma unroll
        for (int c = 0; c < KC; ++c) {
          const float4 v = xv[rr][c];
          a += (v.x + v.y) + (v.z + v.w);
          q += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
        }
        la += a;
        lq += q;
      }
    }
    float out0 = 0.f, out1 = 0.f;
    for (int n = 0; n < N; ++n) {
      float sacc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < KC; ++c) {
        const float4 wv = *(const float4*)(wl + n * K + c * 256 + lane * 4);
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          sacc[rr] = fmaf(xv[rr][c].x, wv.x, sacc[rr]);
          sacc[rr] = fmaf(xv[rr][c].y, wv.y, sacc[rr]);
          sacc[rr] = fmaf(xv[rr][c].z, wv.z, sacc[rr]);
          sacc[rr] = fmaf(xv[rr][c].w, wv.w, sacc[rr]);
        }
      }
      float k0 = hi32 ? sacc[2] : sacc[0], k1 = hi32 ? sacc[3] : sacc[1];
      const float g0 = hi32 ? sacc[0] : sacc[2], g1 = hi32 ? sacc[1] : sacc[3];
      k0 += __shfl_xor(g0, 32, 64);
      k1 += __shfl_xor(g1, 32, 64);
      float v = hi16 ? k1 : k0;
      v += __shfl_xor(hi16 ? k0 : k1, 16, 64);
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
      if ((lane & 15) == (n & 15)) { if (n < 16) out0 = v; else out1 = v; }
    }
    const int64_t row = r0 + myrow;
    if (row < rows) {
      const int n0 = lane & 15;
      if (n0 < N) dots[row * N + n0] = out0;
      if (n0 + 16 < N) dots[row * N + n0 + 16] = out1;
    }
  }
  if (mom) {
    flush();
    if (!last_workgroup(ticket, gridDim.x)) return;
    // the last workgroup: group g's slots in slot order, one wave per group (lane-strided partial sums, then the shuffle tree: a fixed shape)
    const int64_t n_groups = (rows + rows_per_group - 1) / rows_per_group;
    for (int64_t g = wave; g < n_groups; g += 4) {
      const int64_t a = g * rows_per_group, b = (g + 1) * rows_per_group < rows ? (g + 1) * rows_per_group : rows;
      const int64_t nj = ((b - 1) >> 2) - (a >> 2) + 1;
      const int n = (int)(nj < NW ? nj : NW);
      const double* sl = slots + (size_t)g * slots_per_group * 2;
      double ta = 0.0, tq = 0.0;
      int i = lane;
      for (; i + 3 * 64 < n; i += 4 * 64) {   // eight loads in flight per lane, added in slot order
        double va[4], vq[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { va[j] = ld_partial(sl + 2 * (i + 64 * j)); vq[j] = ld_partial(sl + 2 * (i + 64 * j) + 1); }
#pragma unroll
        for (int j = 0; j < 4; ++j) { ta += va[j]; tq += vq[j]; }
      }
      for (; i < n; i += 64) { ta += ld_partial(sl + 2 * i); tq += ld_partial(sl + 2 * i + 1); }
      ta = wave_sum(ta);
      tq = wave_sum(tq);
      if (lane == 0) { mom[2 * g] = ta; mom[2 * g + 1] = tq; }
    }
  }
}

// one thread per row: logits = (dots - mean * wsum) * rstd + bias; optional per-frame decode (same rule as decode_frames_kernel)
__global__ void head_finish_kernel(const float* __restrict__ dots, int64_t rows, int N, const float* __restrict__ wsum,
                                   const float* __restrict__ bias, const double* __restrict__ mom, int64_t rows_per_group,
                                   double group_elems, float eps, float* __restrict__ logits, FrameOut* __restrict__ frames,
                                   int n_oct, int n_cls) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  float mu = 0.f, rs = 1.f;
  if (mom) {
    const int64_t g = r / rows_per_group;
    const double mean = mom[2 * g] / group_elems;
    const double var = mom[2 * g + 1] / group_elems - mean * mean;
    mu = (float)mean;
    rs = (float)(1.0 / sqrt(var + (double)eps));
  }
  float l[32];
#pragma unroll
  for (int j = 0; j < 32; ++j)
    if (j < N) {
      l[j] = fmaf(dots[r * N + j] - mu * wsum[j], rs, bias ? bias[j] : 0.f);
      logits[r * N + j] = l[j];
    }
  if (frames) {
    FrameOut f;
    f.p_on = 1.f / (1.f + expf(-l[0]));
    f.p_off = 1.f / (1.f + expf(-l[1]));
    int bo = 0;
    float bv = l[2];
#pragma unroll
    for (int j = 3; j < 32; ++j)
      if (j <= 2 + n_oct && l[j] > bv) { bv = l[j]; bo = j - 2; }
    int bc = 0;
    const int c0 = 2 + n_oct + 1;
    bv = -3.4e38f;
#pragma unroll
    for (int j = 3; j < 32; ++j)
      if (j >= c0 && j <= c0 + n_cls) {
        if (j == c0) { bv = l[j]; bc = 0; }
        else if (l[j] > bv) { bv = l[j]; bc = j - c0; }
      }
    f.octave = bo;
    f.pitch_class = bc;
    frames[r] = f;
  }
}

// ------------------------------------------------------------------------------------------------
// Validation losses of the recipes (speechbrain/nnet/losses.py:402-519 nll_loss / bce_loss over
// compute_masked_loss :624-684).  One workgroup per batch item: per-frame loss x length mask, block-reduced in a
// fixed order (deterministic) into double sums {sum loss*mask, sum mask, sum mean_c(logp)*mask}; a second tiny
// kernel applies the reduction mode.  mask[b,t] = (float)t < rel_len[b] * (float)T, the fp32 comparison
// length_to_mask makes (speechbrain/dataio/dataio.py:661-706).
__device__ __forceinline__ double block_sum_256(double v, double* sh) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  __syncthreads();
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}

__global__ __launch_bounds__(256) void bce_loss_kernel(const float* x, int64_t t_pred, const float* y, int64_t t_tgt, int64_t T,
                                                       const float* rel_len, const float* pos_weight, float* per_frame,
                                                       double* sums) {
  __shared__ double sh[4];
  const int b = blockIdx.x;
  const float lim = rel_len ? __fmul_rn(rel_len[b], (float)T) : 0.f;
  const float pw = pos_weight ? pos_weight[0] : 1.f;
  double sl = 0.0, sm = 0.0;
  for (int64_t t = threadIdx.x; t < T; t += 256) {
    const float xv = x[b * t_pred + t], yv = y[b * t_tgt + t];
    const float m = rel_len ? ((float)t < lim ? 1.f : 0.f) : 1.f;
    // torch binary_cross_entropy_with_logits: (1 - y) x + (1 + (pw - 1) y) (log1p(exp(-|x|)) + max(-x, 0))
    const float sp = log1pf(expf(-fabsf(xv))) + fmaxf(-xv, 0.f);
    const float lw = pos_weight ? 1.f + (pw - 1.f) * yv : 1.f;
    const float l = ((1.f - yv) * xv + lw * sp) * m;
    if (per_frame) per_frame[b * T + t] = l;
    sl += (double)l;
    sm += (double)m;
  }
  sl = block_sum_256(sl, sh);
  sm = block_sum_256(sm, sh);
  if (threadIdx.x == 0) { sums[b * 3 + 0] = sl; sums[b * 3 + 1] = sm; sums[b * 3 + 2] = 0.0; }
}

__global__ __launch_bounds__(256) void nll_loss_kernel(const float* logp, int64_t t_pred, int C, const int64_t* tgt, int64_t t_tgt,
                                                       int64_t T, const float* rel_len, float* per_frame, double* sums,
                                                       int* bad_target) {
  __shared__ double sh[4];
  const int b = blockIdx.x;
  const float lim = rel_len ? __fmul_rn(rel_len[b], (float)T) : 0.f;
  double sl = 0.0, sm = 0.0, sr = 0.0;
  for (int64_t t = threadIdx.x; t < T; t += 256) {
    const float* row = logp + (b * t_pred + t) * C;
    const int64_t k = tgt[b * t_tgt + t];
    const float m = rel_len ? ((float)t < lim ? 1.f : 0.f) : 1.f;
    float l = 0.f;
    if (k == -100) l = 0.f;  // torch.nn.functional.nll_loss ignore_index default
    else if (k < 0 || k >= C) { atomicExch(bad_target, 1); }
    else l = -row[k];
    l *= m;
    float mean = 0.f;
    for (int c = 0; c < C; ++c) mean += row[c];
    mean = mean / (float)C * m;
    if (per_frame) per_frame[b * T + t] = l;
    sl += (double)l;
    sm += (double)m;
    sr += (double)mean;
  }
  sl = block_sum_256(sl, sh);
  sm = block_sum_256(sm, sh);
  sr = block_sum_256(sr, sh);
  if (threadIdx.x == 0) { sums[b * 3 + 0] = sl; sums[b * 3 + 1] = sm; sums[b * 3 + 2] = sr; }
}

// reduction: 0 mean, 1 batchmean, 2 batch (B outputs); label smoothing as compute_masked_loss :670-684
__global__ void loss_reduce_kernel(const double* sums, int B, int reduction, float smoothing, float* out) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  if (reduction == 2) {
    for (int b = 0; b < B; ++b) {
      const float l = (float)sums[b * 3] / (float)sums[b * 3 + 1];
      const float r = (float)sums[b * 3 + 2] / (float)sums[b * 3 + 1];
      out[b] = smoothing == 0.f ? l : -smoothing * r + (1.f - smoothing) * l;
    }
    return;
  }
  double sl = 0.0, sm = 0.0, sr = 0.0;
  for (int b = 0; b < B; ++b) { sl += sums[b * 3]; sm += sums[b * 3 + 1]; sr += sums[b * 3 + 2]; }
  const float den = reduction == 0 ? (float)sm : (float)B;
  const float l = (float)sl / den, r = (float)sr / den;
  out[0] = smoothing == 0.f ? l : -smoothing * r + (1.f - smoothing) * l;
}

// y = log_softmax(x) / softmax(x) over the last axis (speechbrain/nnet/activations.py Softmax): one thread per row for
// the narrow heads of this path (n <= 64), fp32
__global__ void softmax_small_kernel(const float* x, int64_t rows, int n, int apply_log, float* y) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  const float* xr = x + r * n;
  float mx = xr[0];
  for (int i = 1; i < n; ++i) mx = fmaxf(mx, xr[i]);
  float s = 0.f;
  for (int i = 0; i < n; ++i) s += expf(xr[i] - mx);
  const float ls = logf(s);
  for (int i = 0; i < n; ++i) y[r * n + i] = apply_log ? (xr[i] - mx) - ls : expf(xr[i] - mx) / s;
}

// Fbank add-ons (speechbrain/processing/features.py: Deltas :788-850, ContextWindow :853-940).
// delta[b,t,c] = sum_{k=-n..n} k * x[b, clamp(t+k), c] / denom   (replicate padding), x and out (B,T,ld) with column offsets
__global__ void deltas_kernel(const float* x, long ldx, int B, int T, int C, int n, float inv_denom, float* out, long ldo) {
  const long total = (long)B * T * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const long r = i / C;
    const int t = (int)(r % T);
    const long b = r / T;
    float acc = 0.f;
    for (int k = -n; k <= n; ++k) {
      int tt = t + k;
      tt = tt < 0 ? 0 : (tt > T - 1 ? T - 1 : tt);
      acc = fmaf((float)k, x[(b * T + tt) * ldx + c], acc);
    }
    out[(b * T + t) * ldo + c] = acc * inv_denom;
  }
}
// out[b,t,c*ctx + j] = x[b, t + j - left', c] with zero padding, where the kernel is eye(ctx, klen) rolled by
// max(right - left, 0): tap j reads offset j + lag - pad, pad = max(left, right)
__global__ void context_window_kernel(const float* x, int B, int T, int C, int ctx, int lag, int pad, float* out) {
  const long total = (long)B * T * C * ctx;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int j = (int)(i % ctx);
    long r = i / ctx;
    const int c = (int)(r % C);
    r /= C;
    const int t = (int)(r % T);
    const long b = r / T;
    const int tt = t + j + lag - pad;
    out[i] = (tt >= 0 && tt < T) ? x[(b * T + tt) * C + c] : 0.f;
  }
}

// ---- WavLM gated relative position bias (HF modeling_wavlm.py WavLMAttention) ----
// pb[h][d + T - 1] = embed[bucket(d)][h], d = key - query in (-T, T): half of the buckets per sign, exact below max_exact,
// log-spaced above, with torch's fp32 arithmetic (log(|d| / max_exact) / log(max_distance / max_exact) * (nb - max_exact),
// truncated)
__global__ void relpos_table_kernel(const float* embed, int H, int T, int num_buckets, int max_distance, float* pb) {
  const int n = 2 * T - 1;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * H) return;
  const int h = i / n, di = i - h * n;
  const int d = di - (T - 1);
  const int nb = num_buckets / 2, max_exact = nb / 2;
  int bucket = d > 0 ? nb : 0;
  const int a = d < 0 ? -d : d;
  if (a < max_exact) bucket += a;
  else {
    float v = logf((float)a / (float)max_exact);
    v = v / (float)log((double)max_distance / (double)max_exact);
    v = v * (float)(nb - max_exact);
    long lb = (long)((float)max_exact + v);
    if (lb > nb - 1) lb = nb - 1;
    bucket += (int)lb;
  }
  pb[i] = embed[bucket * H + h];
}
// gate[b][h][t] = ga * (gb * const[h] - 1) + 2,  ga / gb = sigmoid(u_h . wa + ba), sigmoid(u_h . wb + bb); u = the attention
// input (operand type), wa / wb = the sums of rows 0-3 / 4-7 of gru_rel_pos_linear (folded at finalize)
template <typename T>
__global__ void relpos_gate_kernel(const T* u, int64_t rows, int Tt, int H, int dh, const float* wab, const float* bab,
                                   const float* cst, float* gate) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * H) return;
  const int h = (int)(i % H);
  const int64_t row = i / H;
  const T* x = u + row * (int64_t)H * dh + (int64_t)h * dh;
  float sa = bab[0], sb = bab[1];
  for (int d = 0; d < dh; ++d) {
    const float xv = (float)x[d];
    sa = fmaf(xv, wab[d], sa);
    sb = fmaf(xv, wab[dh + d], sb);
  }
  const float ga = 1.f / (1.f + expf(-sa)), gb = 1.f / (1.f + expf(-sb));
  const int64_t b = row / Tt, t = row % Tt;
  gate[(b * H + h) * Tt + t] = ga * (gb * cst[h] - 1.f) + 2.f;
}
// Coalesced form: dh / 8 consecutive lanes own one (row, head) -- each loads 8 consecutive features (a wave reads 512
// consecutive elements of u) and the two dot products are folded across the group with lane exchanges.  (The one-thread-per-
// head form above reads 64 different lines per load instruction: 55 us per layer at 32 x 499 frames x 12 heads instead of 8.)
template <typename T>
__global__ __launch_bounds__(256) void relpos_gate_vec_kernel(const T* u, int64_t n, int Tt, int H, int dh, const float* wab,
                                                              const float* bab, const float* cst, float* gate) {
  const int lph = dh >> 3;  // lanes per head: a power of two <= 64 (checked by the launcher)
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t grp = tid / lph;
  const int sub = (int)(tid % lph);
  const bool ok = grp < n;
  float sa = 0.f, sb = 0.f;
  if (ok) {
    const T* x = u + grp * dh + sub * 8;
    const float* wa = wab + sub * 8;
    const float* wb = wab + dh + sub * 8;
    float xv[8];
    if constexpr (sizeof(T) == 2) {
      const bf16x8 v = *(const bf16x8*)x;  // one 16-byte load per lane
#pragma unroll
      for (int j = 0; j < 8; ++j) xv[j] = (float)v[j];
    } else {
      const float4 v0 = *(const float4*)x, v1 = *(const float4*)(x + 4);
      xv[0] = v0.x; xv[1] = v0.y; xv[2] = v0.z; xv[3] = v0.w; xv[4] = v1.x; xv[5] = v1.y; xv[6] = v1.z; xv[7] = v1.w;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      sa = fmaf(xv[j], wa[j], sa);
      sb = fmaf(xv[j], wb[j], sb);
    }
  }
  for (int o = lph >> 1; o > 0; o >>= 1) {
    sa += __shfl_xor(sa, o, 64);
    sb += __shfl_xor(sb, o, 64);
  }
  if (ok && sub == 0) {
    sa += bab[0];
    sb += bab[1];
    const float ga = 1.f / (1.f + expf(-sa)), gb = 1.f / (1.f + expf(-sb));
    const int h = (int)(grp % H);
    const int64_t row = grp / H;
    const int64_t b = row / Tt, t = row % Tt;
    gate[(b * H + h) * Tt + t] = ga * (gb * cst[h] - 1.f) + 2.f;
  }
}
// materialised-score path: S[b,h,q,k] += gate[b,h,q] * pb[h][k - q + T - 1]
__global__ void scores_add_relbias_kernel(float* S, int64_t BH, int H, int T, int Tp, const float* gate, const float* pb) {
  const int64_t n = BH * T * (int64_t)T;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int k = (int)(i % T);
    int64_t r = i / T;
    const int q = (int)(r % T);
    const int64_t bh = r / T;
    const int h = (int)(bh % H);
    S[(bh * T + q) * Tp + k] += gate[bh * T + q] * pb[(int64_t)h * (2 * T - 1) + (k - q + T - 1)];
  }
}

__global__ void decode_frames_kernel(const float* logits, int64_t rows, int n_out, int n_oct, int n_cls,
                                     FrameOut* out) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  const float* l = logits + r * n_out;
  FrameOut f;
  f.p_on = 1.f / (1.f + expf(-l[0]));
  f.p_off = 1.f / (1.f + expf(-l[1]));
  int bo = 0;
  float bv = l[2];
  for (int i = 1; i <= n_oct; ++i)
    if (l[2 + i] > bv) { bv = l[2 + i]; bo = i; }
  int bc = 0;
  const float* c = l + 2 + n_oct + 1;
  bv = c[0];
  for (int i = 1; i <= n_cls; ++i)
    if (c[i] > bv) { bv = c[i]; bc = i; }
  f.octave = bo;
  f.pitch_class = bc;
  out[r] = f;
}

// CTC greedy: one block per sequence.  argmax per frame, then order-preserving compaction of
// "first of a run, not blank, inside the relative length".
__global__ __launch_bounds__(256) void ctc_greedy_kernel(const float* probs, int T, int V, const float* rel_lens,
                                                         int blank, int32_t* tokens, int32_t* out_lens) {
  extern __shared__ int32_t ids[];  // T
  __shared__ int wave_tot[4];
  __shared__ int running;
  const int b = blockIdx.x;
  const float* p = probs + (int64_t)b * T * V;
  int n = (int)rintf(rel_lens[b] * (float)T);
  if (n > T) n = T;
  if (n < 0) n = 0;
  for (int t = threadIdx.x; t < n; t += blockDim.x) {
    const float* q = p + (int64_t)t * V;
    int best = 0;
    float bv = q[0];
    for (int v = 1; v < V; ++v)
      if (q[v] > bv) { bv = q[v]; best = v; }
    ids[t] = best;
  }
  if (threadIdx.x == 0) running = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int base = 0; base < n; base += blockDim.x) {
    const int t = base + threadIdx.x;
    bool keep = false;
    int id = 0;
    if (t < n) {
      id = ids[t];
      keep = (t == 0 || id != ids[t - 1]) && id != blank;
    }
    const unsigned long long m = __ballot(keep);
    const int before = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) wave_tot[wave] = __popcll(m);
    __syncthreads();
    int off = running;
    for (int w = 0; w < wave; ++w) off += wave_tot[w];
    if (keep) tokens[(int64_t)b * T + off + before] = id;
    __syncthreads();
    if (threadIdx.x == 0) running += wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
    __syncthreads();
  }
  if (threadIdx.x == 0) out_lens[b] = running;
}

// ------------------------------------------------------------------------------------------------
// Fbank pieces: framing (centred, zero padded) * window; power spectrum; dB + per-sequence top_db clip
__global__ void fbank_frames_kernel(const float* wav, int64_t L, int n_fft, int hop, int64_t nframes,
                                    const float* window, float* frames, int64_t total) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < total; i += stride) {
    const int k = (int)(i % n_fft);
    const int64_t r = i / n_fft;
    const int64_t f = r % nframes;
    const int64_t b = r / nframes;
    const int64_t pos = f * hop + k - n_fft / 2;
    const float v = (pos >= 0 && pos < L) ? wav[b * L + pos] : 0.f;
    frames[i] = v * window[k];
  }
}

__global__ void power_spectrum_kernel(const float* reim, int64_t rows, int nb, int imoff, int ld, float* power, int ldp) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = rows * ldp;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < total; i += stride) {
    const int k = (int)(i % ldp);
    const int64_t r = i / ldp;
    float v = 0.f;
    if (k < nb) {
      const float re = reim[r * ld + k], im = reim[r * ld + imoff + k];
      v = re * re + im * im;
    }
    power[i] = v;
  }
}

__global__ __launch_bounds__(256) void fbank_db_kernel(float* fb, int64_t per_seq, float top_db) {
  __shared__ float sh[4];
  float* x = fb + (int64_t)blockIdx.x * per_seq;
  float mx = -INFINITY;
  for (int64_t i = threadIdx.x; i < per_seq; i += blockDim.x) {
    const float v = 10.f * log10f(fmaxf(x[i], 1e-10f));
    x[i] = v;
    mx = fmaxf(mx, v);
  }
  mx = wave_max(mx);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = mx;
  __syncthreads();
  const float floor_db = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3])) - top_db;
  for (int64_t i = threadIdx.x; i < per_seq; i += blockDim.x) x[i] = fmaxf(x[i], floor_db);
}

// fp32 <-> pair rows (test / debug hooks and re-layouts outside the hot path): one thread per 8 consecutive elements
template <int PK>
__global__ void f32_to_pairs_kernel(const float* __restrict__ x, void* __restrict__ out, int64_t n8) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n8) return;
  const float4 a = ((const float4*)x)[2 * i], b = ((const float4*)x)[2 * i + 1];
  const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  store_pairs<PK, 8>(out, i * 8, v);
}
template <int PK>
__global__ void pairs_to_f32_kernel(const void* __restrict__ in, float* __restrict__ y, int64_t n8, const void* lo_plane) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n8) return;
  const int64_t e = i * 8;
  // lo_plane == nullptr: pair rows; else `in` / `lo_plane` are separate (hi, lo) planes with the element layout of y
  const char* ph = lo_plane ? (const char*)in + e * 2 : (const char*)in + (e >> 5) * 128 + (e & 31) * 2;
  const char* pl = lo_plane ? (const char*)lo_plane + e * 2 : ph + 64;
  const uint4 h = *(const uint4*)ph, l = *(const uint4*)pl;
  const unsigned hw[4] = {h.x, h.y, h.z, h.w}, lw[4] = {l.x, l.y, l.z, l.w};
  float o[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const unsigned short hs = (unsigned short)(hw[j >> 1] >> (16 * (j & 1))), ls = (unsigned short)(lw[j >> 1] >> (16 * (j & 1)));
    if constexpr (PK == 3) o[j] = (float)__builtin_bit_cast(_Float16, hs) + (float)__builtin_bit_cast(_Float16, ls);
    else o[j] = (float)__builtin_bit_cast(__bf16, hs) + (float)__builtin_bit_cast(__bf16, ls);
  }
  ((float4*)y)[2 * i] = float4{o[0], o[1], o[2], o[3]};
  ((float4*)y)[2 * i + 1] = float4{o[4], o[5], o[6], o[7]};
}

inline int grid_for(int64_t n, int block = 256, int cap = 8192) {
  int64_t g = (n + block - 1) / block;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

}  // namespace

// ================================================================================================
int launch_f32_to_pairs(int kind, const float* x, void* out, int64_t n, hipStream_t s) {
  if (n % 32 || ((uintptr_t)x & 15) || ((uintptr_t)out & 127) || (kind != 2 && kind != 3)) { set_error("f32_to_pairs: n % 32, alignment or kind"); return -1; }
  const int64_t n8 = n / 8;
  if (kind == 3) hipLaunchKernelGGL((f32_to_pairs_kernel<3>), dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, s, x, out, n8);
  else hipLaunchKernelGGL((f32_to_pairs_kernel<2>), dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, s, x, out, n8);
  SVT_LAUNCH_CHECK();
  return 0;
}
int launch_pairs_to_f32(int kind, const void* in, const void* lo_plane, float* y, int64_t n, hipStream_t s) {
  if (n % 32 || ((uintptr_t)y & 15) || ((uintptr_t)in & 15) || ((uintptr_t)lo_plane & 15) || (kind != 2 && kind != 3)) { set_error("pairs_to_f32: n % 32, alignment or kind"); return -1; }
  const int64_t n8 = n / 8;
  if (kind == 3) hipLaunchKernelGGL((pairs_to_f32_kernel<3>), dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, s, in, y, n8, lo_plane);
  else hipLaunchKernelGGL((pairs_to_f32_kernel<2>), dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, s, in, y, n8, lo_plane);
  SVT_LAUNCH_CHECK();
  return 0;
}

// Fills and copies as KERNELS of this library, not hipMemsetAsync / hipMemcpyAsync (round 6): inside a captured hipGraph the runtime's
// memset node ran out of order from the second replay on -- the statistics region of the workspace was zeroed AFTER the moments kernel had
// written it, the output norm saw (0, 0) and scaled by 1 / sqrt(eps) (tools/scratch/graph_debug.py; tests/test_gpu_graph.py).  A kernel
// node keeps stream order.  Sizes and addresses are multiples of 16 bytes (the workspace carve is 256-byte aligned).
namespace {
__global__ __launch_bounds__(256) void zero16_kernel(uint4* p, int64_t n16) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n16; i += stride) p[i] = uint4{0u, 0u, 0u, 0u};
}
__global__ __launch_bounds__(256) void copy16_kernel(const uint4* in, uint4* out, int64_t n16, const float* in_tail, float* out_tail, int n_tail) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  if (i < n_tail) out_tail[i] = in_tail[i];
  for (; i < n16; i += stride) out[i] = in[i];
}
// rows x cols fp32 block with row pitch ld (elements) set to zero (cols, ld multiples of 4, 16-byte aligned base: the launcher checks)
__global__ __launch_bounds__(256) void zero_cols_kernel(float* p, int64_t rows, int cols4, int64_t ld) {
  const int64_t n = rows * cols4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / cols4;
    const int c = (int)(i - r * cols4);
    *(float4*)(p + r * ld + 4 * c) = float4{0.f, 0.f, 0.f, 0.f};
  }
}
}  // namespace
int launch_zero_bytes(void* p, size_t bytes, hipStream_t s) {
  if (bytes % 16 || ((uintptr_t)p & 15)) { set_error("launch_zero_bytes: 16-byte granularity"); return -1; }
  if (!bytes) return 0;
  hipLaunchKernelGGL(zero16_kernel, dim3(grid_for((int64_t)(bytes / 16))), dim3(256), 0, s, (uint4*)p, (int64_t)(bytes / 16));
  SVT_LAUNCH_CHECK();
  return 0;
}
int launch_copy_f32(const float* in, float* out, int64_t n, hipStream_t s) {
  if (n <= 0) return 0;
  if (((uintptr_t)in & 15) || ((uintptr_t)out & 15)) { set_error("launch_copy_f32: 16-byte alignment"); return -1; }
  const int64_t n16 = n / 4;
  hipLaunchKernelGGL(copy16_kernel, dim3(grid_for(n16 > 0 ? n16 : 1)), dim3(256), 0, s, (const uint4*)in, (uint4*)out, n16, in + 4 * n16, out + 4 * n16,
                     (int)(n - 4 * n16));
  SVT_LAUNCH_CHECK();
  return 0;
}
int launch_zero_cols(float* p, int64_t rows, int cols, int64_t ld, hipStream_t s) {
  if (cols % 4 || ld % 4 || ((uintptr_t)p & 15)) { set_error("launch_zero_cols: 16-byte granularity"); return -1; }
  if (rows <= 0 || cols <= 0) return 0;
  hipLaunchKernelGGL(zero_cols_kernel, dim3(grid_for(rows * (cols / 4))), dim3(256), 0, s, p, rows, cols / 4, ld);
  SVT_LAUNCH_CHECK();
  return 0;
}

int launch_f32_to_bf16(const float* in, bf16_t* out, int64_t n, hipStream_t s) {
  hipLaunchKernelGGL(f32_to_bf16_kernel, dim3(grid_for(n)), dim3(256), 0, s, in, out, n);
  SVT_LAUNCH_CHECK();
  return 0;
}

// scratch of one launch_moments call over at most `groups_max` groups: a ticket per group, then two fp64 partials per workgroup
static int moments_grid(int64_t n, int groups) {
  return groups > 1 ? grid_for(n / 4 + 1, 256, 512 / (groups < 64 ? groups : 64) + 1) : grid_for(n / 4 + 1, 256, 512);
}
static size_t moments_ticket_bytes(int groups) { return ((size_t)groups * 4 + 255) & ~(size_t)255; }
size_t moments_scratch_bytes(int groups_max) {
  // groups < 64: groups * (512 / groups + 1) <= 512 + groups workgroups; otherwise 9 per group
  const size_t wgs = (size_t)(groups_max < 64 ? 512 + groups_max : 576) + 9 * (size_t)groups_max;
  return moments_ticket_bytes(groups_max) + wgs * 2 * sizeof(double);
}

int launch_moments(const float* x, int64_t n, double* moments, void* scratch, int groups_max, hipStream_t s, int groups) {
  if (groups > 1 && (n & 3)) { set_error("moments: per-group length must be a multiple of 4 elements"); return -1; }
  if (groups > groups_max || !scratch || ((uintptr_t)scratch & 7)) { set_error("moments: scratch"); return -1; }
  const int gx = moments_grid(n, groups);
  if (moments_ticket_bytes(groups_max) + (size_t)gx * groups * 2 * sizeof(double) > moments_scratch_bytes(groups_max)) { set_error("moments: scratch too small"); return -1; }
  unsigned* ticket = (unsigned*)scratch;
  double* part = (double*)((char*)scratch + moments_ticket_bytes(groups_max));
  hipLaunchKernelGGL(moments_kernel, dim3(gx, groups), dim3(256), 0, s, x, n, moments, part, ticket);
  SVT_LAUNCH_CHECK();
  return 0;
}

int launch_global_norm(const float* x, float* y, int64_t n, const double* moments, float eps, hipStream_t s, int groups, double n_stat) {
  if (groups > 1 && (n & 3)) { set_error("global_norm: per-group length must be a multiple of 4 elements"); return -1; }
  hipLaunchKernelGGL(global_norm_kernel, dim3(grid_for(n / 4 + 1), groups), dim3(256), 0, s, x, y, n, moments, eps,
                     n_stat > 0 ? n_stat : (double)n);
  SVT_LAUNCH_CHECK();
  return 0;
}

int g_ln_two_rows = 1;   // svt_debug_set key 20: 0 = one row per wave in the (hi, lo) and 16-bit-row LayerNorms (A/B)
int launch_layernorm(int prec, const void* x, int x_is_f32, int64_t rows, int D, const float* gamma,
                     const float* beta, float eps, int gelu, void* yT, float* yF, hipStream_t s, const float* add,
                     float* sumF, void* yP, int pair_kind, const void* addP) {
  if (addP && !yP) { set_error("layernorm: a pair-row addend needs the pair-row output form"); return -1; }
  if (yP) {
    // split-operand modes: the result leaves as pair rows (and optionally as fp32: yF / sumF) -- fp32 input, D in {512, 768, 1024}
    if (prec || !x_is_f32 || yT || !(D == 512 || D == 768 || D == 1024) || (pair_kind != 2 && pair_kind != 3) || ((uintptr_t)x & 15) ||
        ((uintptr_t)yF & 15) || ((uintptr_t)add & 15) || ((uintptr_t)sumF & 15) || ((uintptr_t)yP & 127) || ((uintptr_t)gamma & 15) ||
        ((uintptr_t)beta & 15)) {
      set_error("layernorm: the pair-row output needs an fp32 input, D in {512, 768, 1024} and aligned buffers");
      return -1;
    }
    const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
#define SVT_LN_PAIRS(VPT)                                                                                                  \
  do {                                                                                                                     \
    if (pair_kind == 3)                                                                                                    \
      hipLaunchKernelGGL((layernorm_f32_vec_kernel<VPT, float, float, 3>), grid, block, 0, s, (const float*)x, add, sumF, rows, gamma, \
                         beta, eps, gelu, (float*)nullptr, yF, yP, addP);                                                  \
    else                                                                                                                   \
      hipLaunchKernelGGL((layernorm_f32_vec_kernel<VPT, float, float, 2>), grid, block, 0, s, (const float*)x, add, sumF, rows, gamma, \
                         beta, eps, gelu, (float*)nullptr, yF, yP, addP);                                                  \
  } while (0)
    if (D == 512) SVT_LN_PAIRS(8);
    else if (D == 768) SVT_LN_PAIRS(12);
    else SVT_LN_PAIRS(16);
#undef SVT_LN_PAIRS
    SVT_LAUNCH_CHECK();
    return 0;
  }
  if (prec && !x_is_f32 && (D == 512 || D == 768 || D == 1024) && yT && !yF && !add && !sumF && g_ln_two_rows && !((uintptr_t)x & 15) &&
      !((uintptr_t)yT & 15) && !((uintptr_t)gamma & 15) && !((uintptr_t)beta & 15)) {
    // 16-bit rows in, 16-bit rows out (in place for the conv stack of the layer-norm extractor): half a wave per row
    const dim3 grid2((unsigned)((rows + 7) / 8)), block(256);
    if (D == 512) hipLaunchKernelGGL((layernorm_op16_rows2_kernel<512>), grid2, block, 0, s, (const bf16_t*)x, rows, gamma, beta, eps, gelu, (bf16_t*)yT);
    else if (D == 768) hipLaunchKernelGGL((layernorm_op16_rows2_kernel<768>), grid2, block, 0, s, (const bf16_t*)x, rows, gamma, beta, eps, gelu, (bf16_t*)yT);
    else hipLaunchKernelGGL((layernorm_op16_rows2_kernel<1024>), grid2, block, 0, s, (const bf16_t*)x, rows, gamma, beta, eps, gelu, (bf16_t*)yT);
    SVT_LAUNCH_CHECK();
    return 0;
  }
  if (prec && !x_is_f32 && (D == 512 || D == 768 || D == 1024) && !((uintptr_t)x & 7) && !((uintptr_t)yT & 15) &&
      !((uintptr_t)yF & 15) && !((uintptr_t)add & 15) && !((uintptr_t)sumF & 15)) {
    // bf16 branch output + fp32 residual (throughput mode)
    const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
    if (D == 512)
      hipLaunchKernelGGL((layernorm_f32_vec_kernel<8, bf16_t, bf16_t>), grid, block, 0, s, (const bf16_t*)x, add, sumF, rows,
                         gamma, beta, eps, gelu, (bf16_t*)yT, yF);
    else if (D == 768)
      hipLaunchKernelGGL((layernorm_f32_vec_kernel<12, bf16_t, bf16_t>), grid, block, 0, s, (const bf16_t*)x, add, sumF, rows,
                         gamma, beta, eps, gelu, (bf16_t*)yT, yF);
    else
      hipLaunchKernelGGL((layernorm_f32_vec_kernel<16, bf16_t, bf16_t>), grid, block, 0, s, (const bf16_t*)x, add, sumF, rows,
                         gamma, beta, eps, gelu, (bf16_t*)yT, yF);
    SVT_LAUNCH_CHECK();
    return 0;
  }
  if (add && !x_is_f32) { set_error("layernorm: the addend form needs an fp32 input (or bf16 with D in {512,768,1024})"); return -1; }
  const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
  if (x_is_f32 && (D == 512 || D == 768 || D == 1024) && !((uintptr_t)x & 15) && !((uintptr_t)yT & 15) &&
      !((uintptr_t)yF & 15) && !((uintptr_t)gamma & 15) && !((uintptr_t)beta & 15) && !((uintptr_t)add & 15) &&
      !((uintptr_t)sumF & 15)) {
#define SVT_LN_VEC(VPT)                                                                                          \
  do {                                                                                                           \
    if (prec)                                                                                                    \
      hipLaunchKernelGGL((layernorm_f32_vec_kernel<VPT, bf16_t>), grid, block, 0, s, (const float*)x, add, sumF, \
                         rows, gamma, beta, eps, gelu, (bf16_t*)yT, yF);                                         \
    else                                                                                                         \
      hipLaunchKernelGGL((layernorm_f32_vec_kernel<VPT, float>), grid, block, 0, s, (const float*)x, add, sumF,  \
                         rows, gamma, beta, eps, gelu, (float*)yT, yF);                                          \
  } while (0)
    if (D == 512) SVT_LN_VEC(8);
    else if (D == 768) SVT_LN_VEC(12);
    else SVT_LN_VEC(16);
#undef SVT_LN_VEC
    SVT_LAUNCH_CHECK();
    return 0;
  }
  if (!prec) {
    hipLaunchKernelGGL((layernorm_kernel<float, float>), grid, block, 0, s, (const float*)x, add, sumF, rows, D, gamma,
                       beta, eps, gelu, (float*)yT, yF);
  } else if (x_is_f32) {
    hipLaunchKernelGGL((layernorm_kernel<float, bf16_t>), grid, block, 0, s, (const float*)x, add, sumF, rows, D, gamma,
                       beta, eps, gelu, (bf16_t*)yT, yF);
  } else {
    hipLaunchKernelGGL((layernorm_kernel<bf16_t, bf16_t>), grid, block, 0, s, (const bf16_t*)x, add, sumF, rows, D,
                       gamma, beta, eps, gelu, (bf16_t*)yT, yF);
  }
  SVT_LAUNCH_CHECK();
  return 0;
}

bool layernorm_hilo_ok(int D) { return D == 512 || D == 768 || D == 1024; }
// branch == nullptr && x32 != nullptr: y = LN(x32);  otherwise y = LN(branch + rh + rl)
int launch_layernorm_hilo_parts(const float* parts, int nparts, long part_stride, const float* pbias, const bf16_t* rh, const bf16_t* rl, int64_t rows,
                                int D, const float* gamma, const float* beta, float eps, bf16_t* yh, bf16_t* yl, float* yF, hipStream_t s) {
  if (!parts || nparts < 1 || !pbias || !(D == 512 || D == 768 || D == 1024) || ((uintptr_t)parts & 15) || (part_stride & 3) || ((uintptr_t)pbias & 15) ||
      ((uintptr_t)rh & 15) || ((uintptr_t)rl & 15) || ((uintptr_t)yh & 15) || ((uintptr_t)yl & 15) || ((uintptr_t)yF & 15) || ((uintptr_t)gamma & 15) ||
      ((uintptr_t)beta & 15)) { set_error("layernorm (K-split branch): D in {512, 768, 1024} and 16-byte aligned buffers"); return -1; }
  const dim3 grid2((unsigned)((rows + 7) / 8)), block(256);
  if (D == 512) hipLaunchKernelGGL((layernorm_hilo2_kernel<512, true>), grid2, block, 0, s, nullptr, rh, rl, rows, gamma, beta, eps, yh, yl, yF, parts, nparts, part_stride, pbias);
  else if (D == 768) hipLaunchKernelGGL((layernorm_hilo2_kernel<768, true>), grid2, block, 0, s, nullptr, rh, rl, rows, gamma, beta, eps, yh, yl, yF, parts, nparts, part_stride, pbias);
  else hipLaunchKernelGGL((layernorm_hilo2_kernel<1024, true>), grid2, block, 0, s, nullptr, rh, rl, rows, gamma, beta, eps, yh, yl, yF, parts, nparts, part_stride, pbias);
  SVT_LAUNCH_CHECK();
  return 0;
}
int launch_layernorm_hilo(const bf16_t* branch, const bf16_t* rh, const bf16_t* rl, const float* x32, int64_t rows, int D,
                          const float* gamma, const float* beta, float eps, bf16_t* yh, bf16_t* yl, float* yF, hipStream_t s) {
  const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
  if (x32) {
    if (D == 512) hipLaunchKernelGGL((layernorm_f32_to_hilo_kernel<8>), grid, block, 0, s, x32, rows, gamma, beta, eps, yh, yl);
    else if (D == 768) hipLaunchKernelGGL((layernorm_f32_to_hilo_kernel<12>), grid, block, 0, s, x32, rows, gamma, beta, eps, yh, yl);
    else hipLaunchKernelGGL((layernorm_f32_to_hilo_kernel<16>), grid, block, 0, s, x32, rows, gamma, beta, eps, yh, yl);
  } else if (g_ln_two_rows && !((uintptr_t)branch & 15) && !((uintptr_t)rh & 15) && !((uintptr_t)rl & 15) && !((uintptr_t)yh & 15) &&
             !((uintptr_t)yl & 15) && !((uintptr_t)yF & 15) && !((uintptr_t)gamma & 15) && !((uintptr_t)beta & 15)) {
    const dim3 grid2((unsigned)((rows + 7) / 8));
    if (D == 512) hipLaunchKernelGGL((layernorm_hilo2_kernel<512>), grid2, block, 0, s, branch, rh, rl, rows, gamma, beta, eps, yh, yl, yF);
    else if (D == 768) hipLaunchKernelGGL((layernorm_hilo2_kernel<768>), grid2, block, 0, s, branch, rh, rl, rows, gamma, beta, eps, yh, yl, yF);
    else hipLaunchKernelGGL((layernorm_hilo2_kernel<1024>), grid2, block, 0, s, branch, rh, rl, rows, gamma, beta, eps, yh, yl, yF);
  } else {
    if (D == 512) hipLaunchKernelGGL((layernorm_hilo_kernel<8>), grid, block, 0, s, branch, rh, rl, rows, gamma, beta, eps, yh, yl, yF);
    else if (D == 768) hipLaunchKernelGGL((layernorm_hilo_kernel<12>), grid, block, 0, s, branch, rh, rl, rows, gamma, beta, eps, yh, yl, yF);
    else hipLaunchKernelGGL((layernorm_hilo_kernel<16>), grid, block, 0, s, branch, rh, rl, rows, gamma, beta, eps, yh, yl, yF);
  }
  SVT_LAUNCH_CHECK();
  return 0;
}

// scratch: a ticket per clip, then 65 fp64 partials per workgroup
size_t conv0_window_moments_scratch_bytes(int B, int64_t T1) {
  const size_t gx = (size_t)((T1 + 2047) / 2048);
  return moments_ticket_bytes(B) + (size_t)B * gx * NWM * sizeof(double);
}

int launch_conv0_window_moments(const float* wav, int B, int64_t L, int k, int stride, int64_t T1, double* wm, void* scratch,
                                hipStream_t s) {
  if (k != K0) { set_error("conv layer 0 kernel size must be 10"); return -1; }
  if (!scratch || ((uintptr_t)scratch & 7)) { set_error("conv0 window moments: scratch"); return -1; }
  const int per_block = 256 * 8;
  dim3 grid((unsigned)((T1 + per_block - 1) / per_block), B);
  unsigned* ticket = (unsigned*)scratch;
  double* part = (double*)((char*)scratch + moments_ticket_bytes(B));
  hipLaunchKernelGGL(conv0_window_moments_kernel, grid, dim3(256), 0, s, wav, L, stride, T1, wm, part, ticket);
  SVT_LAUNCH_CHECK();
  return 0;
}

int launch_conv0_group_coef(const double* wav_moments, int64_t n_wav, const double* wm, int B, int64_t T1, int C,
                            int k, const float* w0, const float* b0, const float* gamma, const float* beta,
                            float eps_wav, float eps_gn, float* coef, hipStream_t s, int cpg) {
  if (k != K0) { set_error("conv layer 0 kernel size must be 10"); return -1; }
  dim3 grid((C + 63) / 64, B);
  hipLaunchKernelGGL(conv0_group_coef_kernel, grid, dim3(64), 0, s, wav_moments, n_wav, wm, T1, C, w0, b0, gamma, beta,
                     eps_wav, eps_gn, coef, cpg);
  SVT_LAUNCH_CHECK();
  return 0;
}

int launch_conv0_group_apply(int prec, const float* wav, int B, int64_t L, int k, int stride, int64_t T1, int C,
                             const float* coef, void* out, hipStream_t s, int pair_kind) {
  if (k != K0 || stride > 5 || C > 512 || C % 8) { set_error("conv0: unsupported geometry"); return -1; }
  dim3 grid((unsigned)((T1 + 127) / 128), B);
  if (pair_kind) {
    if (prec || C % 32 || ((uintptr_t)out & 127) || (pair_kind != 2 && pair_kind != 3)) { set_error("conv0: pair-row output needs fp32 storage and C % 32 == 0"); return -1; }
    if (pair_kind == 3) hipLaunchKernelGGL((conv0_group_apply_kernel<float, 3>), grid, dim3(256), 0, s, wav, L, stride, T1, C, coef, (float*)out);
    else hipLaunchKernelGGL((conv0_group_apply_kernel<float, 2>), grid, dim3(256), 0, s, wav, L, stride, T1, C, coef, (float*)out);
  } else if (prec)
    hipLaunchKernelGGL((conv0_group_apply_kernel<bf16_t>), grid, dim3(256), 0, s, wav, L, stride, T1, C, coef,
                       (bf16_t*)out);
  else
    hipLaunchKernelGGL((conv0_group_apply_kernel<float>), grid, dim3(256), 0, s, wav, L, stride, T1, C, coef,
                       (float*)out);
  SVT_LAUNCH_CHECK();
  return 0;
}

int launch_conv0_layer(int prec, const float* wav, int B, int64_t L, int k, int stride, int64_t T1, int C,
                       const double* wav_moments, int64_t n_wav, float eps_wav, const float* w0, const float* b0,
                       const float* gamma, const float* beta, float eps, void* out, hipStream_t s, int cpg, int pair_kind) {
  if (k != K0 || C > 512 || C % 8) { set_error("conv0: unsupported geometry"); return -1; }
  dim3 grid((unsigned)((T1 + 63) / 64), B);
  if (pair_kind) {
    if (prec || C % 32 || ((uintptr_t)out & 127) || (pair_kind != 2 && pair_kind != 3)) { set_error("conv0: pair-row output needs fp32 storage and C % 32 == 0"); return -1; }
    if (pair_kind == 3)
      hipLaunchKernelGGL((conv0_layer_kernel<float, 3>), grid, dim3(256), 0, s, wav, L, stride, T1, C, wav_moments, n_wav, eps_wav, w0, b0, gamma, beta, eps, (float*)out, cpg);
    else
      hipLaunchKernelGGL((conv0_layer_kernel<float, 2>), grid, dim3(256), 0, s, wav, L, stride, T1, C, wav_moments, n_wav, eps_wav, w0, b0, gamma, beta, eps, (float*)out, cpg);
  } else if (prec)
    hipLaunchKernelGGL((conv0_layer_kernel<bf16_t>), grid, dim3(256), 0, s, wav, L, stride, T1, C, wav_moments, n_wav,
                       eps_wav, w0, b0, gamma, beta, eps, (bf16_t*)out, cpg);
  else
    hipLaunchKernelGGL((conv0_layer_kernel<float>), grid, dim3(256), 0, s, wav, L, stride, T1, C, wav_moments, n_wav,
                       eps_wav, w0, b0, gamma, beta, eps, (float*)out, cpg);
  SVT_LAUNCH_CHECK();
  return 0;
}

int launch_posconv_gather(int prec, const float* h, int B, int T, int D, int G, int kp, int Tp, void* out, hipStream_t s, const float* sc,
                          const float* sh) {
  const int64_t n = (int64_t)B * Tp * D;
  if (prec && (D / G) % 8 == 0 && !((uintptr_t)h & 15) && !((uintptr_t)out & 15))
    hipLaunchKernelGGL(posconv_gather_bf16x8_kernel, dim3(grid_for(n / 8)), dim3(256), 0, s, h, B, T, D, G, kp, Tp, (bf16_t*)out, sc, sh);
  else if (prec)
    hipLaunchKernelGGL((posconv_gather_kernel<bf16_t>), dim3(grid_for(n)), dim3(256), 0, s, h, B, T, D, G, kp, Tp,
                       (bf16_t*)out, sc, sh);
  else
    hipLaunchKernelGGL((posconv_gather_kernel<float>), dim3(grid_for(n)), dim3(256), 0, s, h, B, T, D, G, kp, Tp,
                       (float*)out, sc, sh);
  SVT_LAUNCH_CHECK();
  return 0;
}

// multi-frame positional conv: y[g][b*Tq + q][j*cg + c] (bf16, GELU applied) holds frame t = q*P + j of group g;
// pre[b][t][g*cg + c] = h[b][t][g*cg + c] + y[...]  (8 channels per thread)
template <typename TY>
__global__ void posconv_scatter_add_kernel(const float* h, const TY* y, int B, int T, int D, int G, int P, int Tq, float* pre) {
  const int cg = D / G, c8n = D / 8;
  const int64_t n = (int64_t)B * T * c8n;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int ch = (int)(i % c8n) * 8;
    const int64_t r = i / c8n;
    const int t = (int)(r % T), b = (int)(r / T);
    const int g = ch / cg, c = ch - g * cg;
    const int q = t / P, j = t - q * P;
    const TY* yp = y + (((int64_t)g * B + b) * Tq + q) * (P * cg) + j * cg + c;
    float v[8];
    if constexpr (sizeof(TY) == 2) {
      const bf16x8 vb = *(const bf16x8*)yp;
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = (float)vb[i];
    } else {
      const float4 a0 = *(const float4*)yp, a1 = *(const float4*)(yp + 4);
      v[0] = a0.x; v[1] = a0.y; v[2] = a0.z; v[3] = a0.w; v[4] = a1.x; v[5] = a1.y; v[6] = a1.z; v[7] = a1.w;
    }
    const float* hp = h + r * D + ch;
    const float4 h0 = *(const float4*)hp, h1 = *(const float4*)(hp + 4);
    float* op = pre + r * D + ch;
    *(float4*)op = float4{h0.x + v[0], h0.y + v[1], h0.z + v[2], h0.w + v[3]};
    *(float4*)(op + 4) = float4{h1.x + v[4], h1.y + v[5], h1.z + v[6], h1.w + v[7]};
  }
}
int launch_posconv_scatter_add(const float* h, const void* y, int B, int T, int D, int G, int P, int Tq, float* pre, hipStream_t s, int y_f32) {
  const int64_t n = (int64_t)B * T * (D / 8);
  if (y_f32) hipLaunchKernelGGL((posconv_scatter_add_kernel<float>), dim3(grid_for(n)), dim3(256), 0, s, h, (const float*)y, B, T, D, G, P, Tq, pre);
  else hipLaunchKernelGGL((posconv_scatter_add_kernel<bf16_t>), dim3(grid_for(n)), dim3(256), 0, s, h, (const bf16_t*)y, B, T, D, G, P, Tq, pre);
  SVT_LAUNCH_CHECK();
  return 0;
}

int launch_softmax_rows(int prec, const float* S, int64_t rows, int T, int Tp, void* P, hipStream_t s) {
  const dim3 grid((unsigned)((rows + 3) / 4));
  if (prec)
    hipLaunchKernelGGL((softmax_rows_kernel<bf16_t>), grid, dim3(256), 0, s, S, rows, T, Tp, (bf16_t*)P);
  else
    hipLaunchKernelGGL((softmax_rows_kernel<float>), grid, dim3(256), 0, s, S, rows, T, Tp, (float*)P);
  SVT_LAUNCH_CHECK();
  return 0;
}

int launch_transpose_v(int prec, const void* qkv, int B, int T, int H, int dh, long ld, long voff, int Tp, void* Vt,
                       hipStream_t s) {
  dim3 grid((Tp + 31) / 32, (dh + 31) / 32, B * H);
  if (prec)
    hipLaunchKernelGGL((transpose_v_kernel<bf16_t>), grid, dim3(256), 0, s, (const bf16_t*)qkv, T, H, dh, ld, voff, Tp,
                       (bf16_t*)Vt);
  else
    hipLaunchKernelGGL((transpose_v_kernel<float>), grid, dim3(256), 0, s, (const float*)qkv, T, H, dh, ld, voff, Tp,
                       (float*)Vt);
  SVT_LAUNCH_CHECK();
  return 0;
}

int launch_axpby(int prec, const void* x, const void* y, float a, float b, void* out, int64_t n, hipStream_t s) {
  if (prec)
    hipLaunchKernelGGL((axpby_kernel<bf16_t>), dim3(grid_for(n)), dim3(256), 0, s, (const bf16_t*)x, (const bf16_t*)y,
                       a, b, (bf16_t*)out, n);
  else
    hipLaunchKernelGGL((axpby_kernel<float>), dim3(grid_for(n)), dim3(256), 0, s, (const float*)x, (const float*)y, a,
                       b, (float*)out, n);
  SVT_LAUNCH_CHECK();
  return 0;
}

int launch_add_pe(int prec, const float* x, int B, int T, int Tsrc, int D, const float* pe, float* outF, void* outT,
                  hipStream_t s) {
  const int64_t n = (int64_t)B * T * D;
  if (prec)
    hipLaunchKernelGGL((add_pe_kernel<bf16_t>), dim3(grid_for(n)), dim3(256), 0, s, x, B, T, Tsrc, D, pe, outF,
                       (bf16_t*)outT);
  else
    hipLaunchKernelGGL((add_pe_kernel<float>), dim3(grid_for(n)), dim3(256), 0, s, x, B, T, Tsrc, D, pe, outF,
                       (float*)nullptr);
  SVT_LAUNCH_CHECK();
  return 0;
}

int launch_add_f32(const float* a, const float* b, float* out, int64_t n, hipStream_t s) {
  hipLaunchKernelGGL(add_f32_kernel, dim3(grid_for(n)), dim3(256), 0, s, a, b, out, n);
  SVT_LAUNCH_CHECK();
  return 0;
}

template <int KC>
static int launch_linear_head_kc(const float* x, int64_t rows, const float* w, const float* b, int N, float* y, hipStream_t s) {
  const size_t lds = (size_t)N * KC * 256 * 4;
  if (lds > 65536)
    if (int r_ = ensure_dyn_lds((const void*)linear_head_kernel<KC>, (int)lds)) return r_;
  const int64_t groups = (rows + 15) / 16;
  const unsigned grid = (unsigned)(groups < 512 ? groups : 512);
  hipLaunchKernelGGL((linear_head_kernel<KC>), dim3(grid), dim3(256), lds, s, x, rows, w, b, N, y);
  SVT_LAUNCH_CHECK();
  return 0;
}

bool linear_head_eligible(int K, int N) { return N >= 1 && N <= 32 && (K == 512 || K == 768 || K == 1024); }
int launch_linear_head(const float* x, int64_t rows, int K, const float* w, const float* b, int N, float* y, hipStream_t s) {
  if (K == 512) return launch_linear_head_kc<2>(x, rows, w, b, N, y, s);
  if (K == 768) return launch_linear_head_kc<3>(x, rows, w, b, N, y, s);
  if (K == 1024) return launch_linear_head_kc<4>(x, rows, w, b, N, y, s);
  set_error("linear_head: unsupported K");
  return -1;
}

static unsigned head_dots_grid(int64_t rows) {
  const int64_t blocks = (rows + 15) / 16;
  return (unsigned)(blocks < 512 ? blocks : 512);
}
// scratch of the output-norm statistics (head_dots_kernel): a ticket, then one (sum, sum of squares) slot per wave and norm group it meets:
// at most rows / 4 + 2 slots per group, never more than the launch has waves
static int head_slots_per_group(int64_t rows, int64_t rpg) {
  const int64_t nw = (int64_t)head_dots_grid(rows) * 4, per = (rpg + 3) / 4 + 1;
  return (int)(per < nw ? per : nw);
}
size_t head_scratch_bytes(int64_t rows, int groups_max) {
  // for any group size: groups * min(NW, rpg / 4 + 2) <= rows / 4 + 2 groups
  return 256 + ((size_t)rows / 4 + 2 * (size_t)groups_max + 8) * 2 * sizeof(double);
}

template <int KC>
static int launch_head_dots_kc(const float* x, int64_t rows, const float* w, int N, float* dots, double* mom, int64_t rpg, void* scratch,
                               hipStream_t s) {
  const size_t lds = (size_t)N * KC * 256 * 4;
  if (int r_ = ensure_dyn_lds((const void*)head_dots_kernel<KC>, 32 * KC * 256 * 4)) return r_;
  const unsigned grid = head_dots_grid(rows);
  if (mom && (!scratch || ((uintptr_t)scratch & 7))) { set_error("head_fused: scratch"); return -1; }
  if ((unsigned long)rows * KC * 256 * 4 > 0x7FFFFFF0ul) { set_error("head_fused: more than 2 GiB of encoder output (32-bit buffer offsets)"); return -1; }
  unsigned* ticket = (unsigned*)scratch;
  double* slots = scratch ? (double*)((char*)scratch + 256) : nullptr;
  hipLaunchKernelGGL((head_dots_kernel<KC>), dim3(grid), dim3(256), lds, s, x, rows, w, N, dots, mom, rpg, slots, ticket,
                     mom ? head_slots_per_group(rows, rpg) : 0);
  SVT_LAUNCH_CHECK();
  return 0;
}
int launch_head_fused(const float* x, int64_t rows, int K, const float* w, const float* wsum, const float* b, int N, float* dots,
                      double* mom /*2 per group (written); null = no output norm*/, int64_t rows_per_group, float eps, float* logits,
                      FrameOut* frames, int n_oct, int n_cls, hipStream_t s, double n_stat, int (*between)(void*), void* between_arg,
                      void* scratch /*head_scratch_bytes(rows, groups); its first 4 bytes ZERO*/) {
  if (!linear_head_eligible(K, N)) { set_error("head_fused: unsupported head geometry"); return -1; }
  if (frames && (N != 2 + n_oct + 1 + n_cls + 1 || N > 32)) { set_error("head_fused: n_out != 2 + (n_octave+1) + (n_class+1)"); return -1; }
  if (mom) {
    const int64_t groups = (rows + rows_per_group - 1) / rows_per_group;
    if (256 + (size_t)groups * head_slots_per_group(rows, rows_per_group) * 16 > head_scratch_bytes(rows, (int)groups)) { set_error("head_fused: scratch too small"); return -1; }
  }
  int r = K == 512 ? launch_head_dots_kc<2>(x, rows, w, N, dots, mom, rows_per_group, scratch, s)
        : K == 768 ? launch_head_dots_kc<3>(x, rows, w, N, dots, mom, rows_per_group, scratch, s)
                   : launch_head_dots_kc<4>(x, rows, w, N, dots, mom, rows_per_group, scratch, s);
  if (r) return r;
  if (between) { if (int rb = between(between_arg)) return rb; }   // the moments are complete here: cross-rank reduction hook
  hipLaunchKernelGGL(head_finish_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, s, dots, rows, N, wsum, b, mom,
                     rows_per_group, n_stat > 0 ? n_stat : (double)rows_per_group * (double)K, eps, logits, frames, n_oct, n_cls);
  SVT_LAUNCH_CHECK();
  return 0;
}

int launch_linear_f32(const float* x, int64_t rows, int K, const float* w, const float* b, int N, float* y,
                      hipStream_t s) {
  if (N > 32) { set_error("linear_small: N > 32"); return -1; }
  hipLaunchKernelGGL(linear_small_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, x, rows, K, w, b, N, y);
  SVT_LAUNCH_CHECK();
  return 0;
}

int launch_bce_loss(const float* x, int64_t B, int64_t t_pred, const float* y, int64_t t_tgt, int64_t T, const float* rel_len,
                    const float* pos_weight, float* per_frame, double* sums, hipStream_t s) {
  hipLaunchKernelGGL(bce_loss_kernel, dim3((unsigned)B), dim3(256), 0, s, x, t_pred, y, t_tgt, T, rel_len, pos_weight, per_frame, sums);
  SVT_LAUNCH_CHECK();
  return 0;
}
int launch_nll_loss(const float* logp, int64_t B, int64_t t_pred, int C, const int64_t* tgt, int64_t t_tgt, int64_t T,
                    const float* rel_len, float* per_frame, double* sums, int* bad_target, hipStream_t s) {
  hipLaunchKernelGGL(nll_loss_kernel, dim3((unsigned)B), dim3(256), 0, s, logp, t_pred, C, tgt, t_tgt, T, rel_len, per_frame, sums,
                     bad_target);
  SVT_LAUNCH_CHECK();
  return 0;
}
int launch_loss_reduce(const double* sums, int B, int reduction, float smoothing, float* out, hipStream_t s) {
  hipLaunchKernelGGL(loss_reduce_kernel, dim3(1), dim3(64), 0, s, sums, B, reduction, smoothing, out);
  SVT_LAUNCH_CHECK();
  return 0;
}
int launch_softmax_small(const float* x, int64_t rows, int n, int apply_log, float* y, hipStream_t s) {
  hipLaunchKernelGGL(softmax_small_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, s, x, rows, n, apply_log, y);
  SVT_LAUNCH_CHECK();
  return 0;
}

int launch_deltas(const float* x, long ldx, int B, int T, int C, int n, float inv_denom, float* out, long ldo, hipStream_t s) {
  hipLaunchKernelGGL(deltas_kernel, dim3(grid_for((int64_t)B * T * C)), dim3(256), 0, s, x, ldx, B, T, C, n, inv_denom, out, ldo);
  SVT_LAUNCH_CHECK();
  return 0;
}
int launch_context_window(const float* x, int B, int T, int C, int ctx, int lag, int pad, float* out, hipStream_t s) {
  hipLaunchKernelGGL(context_window_kernel, dim3(grid_for((int64_t)B * T * C * ctx)), dim3(256), 0, s, x, B, T, C, ctx, lag, pad, out);
  SVT_LAUNCH_CHECK();
  return 0;
}

int launch_relpos_table(const float* embed, int H, int T, int num_buckets, int max_distance, float* pb, hipStream_t s) {
  const int n = (2 * T - 1) * H;
  hipLaunchKernelGGL(relpos_table_kernel, dim3((n + 255) / 256), dim3(256), 0, s, embed, H, T, num_buckets, max_distance, pb);
  SVT_LAUNCH_CHECK();
  return 0;
}
int launch_relpos_gate(int prec, const void* u, int64_t rows, int T, int H, int dh, const float* wab, const float* bab,
                       const float* cst, float* gate, hipStream_t s) {
  const int64_t n = rows * H;
  const int lph = dh / 8;
  if (dh % 8 == 0 && lph >= 1 && lph <= 64 && (lph & (lph - 1)) == 0 && !((uintptr_t)u & 15)) {
    const int64_t threads = n * lph;
    const dim3 grid((unsigned)((threads + 255) / 256));
    if (prec) hipLaunchKernelGGL(relpos_gate_vec_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)u, n, T, H, dh, wab, bab, cst, gate);
    else hipLaunchKernelGGL(relpos_gate_vec_kernel<float>, grid, dim3(256), 0, s, (const float*)u, n, T, H, dh, wab, bab, cst, gate);
    SVT_LAUNCH_CHECK();
    return 0;
  }
  if (prec) hipLaunchKernelGGL(relpos_gate_kernel<bf16_t>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const bf16_t*)u, rows, T, H, dh, wab, bab, cst, gate);
  else hipLaunchKernelGGL(relpos_gate_kernel<float>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const float*)u, rows, T, H, dh, wab, bab, cst, gate);
  SVT_LAUNCH_CHECK();
  return 0;
}
int launch_scores_add_relbias(float* S, int64_t BH, int H, int T, int Tp, const float* gate, const float* pb, hipStream_t s) {
  hipLaunchKernelGGL(scores_add_relbias_kernel, dim3(grid_for(BH * T * (int64_t)T)), dim3(256), 0, s, S, BH, H, T, Tp, gate, pb);
  SVT_LAUNCH_CHECK();
  return 0;
}

int launch_decode_frames(const float* logits, int64_t rows, int n_out, int n_oct, int n_cls, FrameOut* out,
                         hipStream_t s) {
  hipLaunchKernelGGL(decode_frames_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, s, logits, rows, n_out,
                     n_oct, n_cls, out);
  SVT_LAUNCH_CHECK();
  return 0;
}

int launch_ctc_greedy(const float* probs, int B, int T, int V, const float* rel_lens, int blank, int32_t* tokens,
                      int32_t* out_lens, hipStream_t s) {
  const size_t lds = (size_t)T * sizeof(int32_t);
  if (lds > 60000) { set_error("ctc_greedy: T too large for one block"); return -1; }
  hipLaunchKernelGGL(ctc_greedy_kernel, dim3(B), dim3(256), lds, s, probs, T, V, rel_lens, blank, tokens, out_lens);
  SVT_LAUNCH_CHECK();
  return 0;
}

int launch_fbank_frames(const float* wav, int B, int64_t L, int n_fft, int hop, int64_t nframes, const float* window,
                        float* frames, hipStream_t s) {
  const int64_t total = (int64_t)B * nframes * n_fft;
  hipLaunchKernelGGL(fbank_frames_kernel, dim3(grid_for(total)), dim3(256), 0, s, wav, L, n_fft, hop, nframes, window,
                     frames, total);
  SVT_LAUNCH_CHECK();
  return 0;
}

int launch_power_spectrum(const float* reim, int64_t rows, int nb, int imoff, int ld, float* power, int ldp,
                          hipStream_t s) {
  hipLaunchKernelGGL(power_spectrum_kernel, dim3(grid_for(rows * ldp)), dim3(256), 0, s, reim, rows, nb, imoff, ld,
                     power, ldp);
  SVT_LAUNCH_CHECK();
  return 0;
}

int launch_fbank_db(float* fb, int B, int64_t per_seq, float top_db, hipStream_t s) {
  hipLaunchKernelGGL(fbank_db_kernel, dim3(B), dim3(256), 0, s, fb, per_seq, top_db);
  SVT_LAUNCH_CHECK();
  return 0;
}

// ---- clock stamps (svt_debug_clock): one (shader clock, 100 MHz wall clock) pair per XCD ----
__global__ void clock_stamp_kernel(long long* out) {
  if (threadIdx.x == 0) {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 7u;
    const long long t = __builtin_amdgcn_s_memtime();
    const long long r = __builtin_amdgcn_s_memrealtime();
    out[2 * xcc] = t;
    out[2 * xcc + 1] = r;
  }
}
int launch_clock_stamp(long long* out16, hipStream_t s) {
  hipLaunchKernelGGL(clock_stamp_kernel, dim3(64), dim3(64), 0, s, out16);  // 64 blocks: round-robin over the 8 XCDs
  SVT_LAUNCH_CHECK();
  return 0;
}

int set_ticket_fenced(int on) {   // svt_debug_set key 32
  const int v = on ? 1 : 0;
  SVT_HIP(hipMemcpyToSymbol(HIP_SYMBOL(d_ticket_fenced), &v, sizeof(int)));
  return 0;
}

}  // namespace svt

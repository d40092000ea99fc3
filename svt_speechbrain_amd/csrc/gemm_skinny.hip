// bf16 dense contraction for SMALL problems (a single 5 s utterance: M = 249 rows).  The large-tile LDS-DMA kernels put
// such a problem on 4-24 of the 256 CUs and walk K serially (FFN-2 at M = 249: 6 workgroups x 48 slabs = 46 us; the
// evaluation loop of the recipes runs batch 1 by construction, MIR_ST500/train_audio_ssl.py:90).  Here the tile is
// 64 x 64 and K is split FOUR ways inside the workgroup: wave w multiplies the 64-element slabs w, w+4, w+8, ... with
// fragments loaded straight from global memory into registers (no LDS staging: the operands of a small problem sit in
// L2, and a wave touches whole 128-byte lines over the two k-steps of a slab), three slabs in flight per wave; the four
// partial tiles are summed through LDS and stored row-contiguous with bias / activation / residual applied.
// Same GemmArgs contract as gemm.hip (overlapping A rows for the implicit convolutions, batched z dimensions).
#include "common.h"

namespace svt {
namespace {

__device__ __forceinline__ float act_apply(float v, int act) {
  if (act == ACT_GELU) return gelu_fast(v);
  if (act == ACT_RELU) return v > 0.f ? v : 0.f;
  return v;
}

template <int NB>
struct Frags {
  bf16x8 x[2][NB];  // [k-step][16-row block of A]
  bf16x8 w[2][NB];  // [k-step][16-row block of W]
};

// NB = 16-row / 16-column blocks per tile side: 64 x 64 (NB = 4) or 32 x 32 (NB = 2).  What bounds a small problem is the
// L2 -> CU fill rate of ONE CU (~50 GB/s: a workgroup moves (TM + TN) * K * 2 bytes through it), so when 64 x 64 tiles
// leave most CUs idle the 32 x 32 tiling -- four times the workgroups, half the bytes each -- is the faster one.
template <int NB>
__global__ __launch_bounds__(256) void gemm_skinny_kernel(GemmArgs p) {
  constexpr int TS = 16 * NB;     // tile side
  constexpr int PITCH = TS + 4;   // floats per row of a partial tile in LDS (conflict-free 16-byte accesses)
  extern __shared__ __attribute__((aligned(16))) float red[];  // 4 partial tiles of TS x PITCH floats (68 KiB at NB = 4: dynamic)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_n = (p.N + TS - 1) / TS;
  const int tile_n = blockIdx.x % tiles_n, tile_m = blockIdx.x / tiles_n;
  const int z = blockIdx.y, z1 = z / p.nz2, z2 = z % p.nz2;
  const int m0 = tile_m * TS, n0 = tile_n * TS;
  const bf16_t* A = (const bf16_t*)p.A + (z1 * p.a_z1 + z2 * p.a_z2);
  const bf16_t* W = (const bf16_t*)p.W + (z1 * p.w_z1 + z2 * p.w_z2);

  // MFMA fragment of k-step ks: lane (r16 = lane & 15, cq = lane >> 4) holds 8 consecutive k of row r16 at k = ks*32 + cq*8
  const int r16 = lane & 15, cq = lane >> 4;
  const bf16_t* ap[NB];
  const bf16_t* wp[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    int m = m0 + b * 16 + r16;
    if (m > p.M - 1) m = p.M - 1;
    ap[b] = A + (long)(m / p.a_rpb) * p.a_bstride + (long)(m % p.a_rpb) * p.a_rstride + cq * 8;
    int n = n0 + b * 16 + r16;
    if (n > p.N - 1) n = p.N - 1;
    wp[b] = W + (long)n * p.ldw + cq * 8;
  }
  // Wave w takes the CONTIGUOUS quarter [lo, lo + mine) of the K slabs and walks it from a per-workgroup rotated start.
  // Rows of a K = 3072 (or 1024, 2048 ...) operand are a multiple of 2 KiB apart, so every row of the matrix at one k
  // sits on the same two of the sixteen L2 channels: with the four waves on adjacent slabs and every workgroup in
  // lockstep the whole launch camps on a few channels (FFN-2 at M = 249: 2 us per slab).  Quarters put the waves
  // K/4 apart, the rotation puts the workgroups apart.
  // K split over workgroups (p.ksplit > 1, blockIdx.z = part; round 6): one 5 s utterance's FFN-2 is 192 workgroups walking 48 slabs
  // each -- 14 us of serial memory round trips on a quarter-filled chip; four parts make it 768 workgroups of 12 slabs (three per wave, all
  // in flight at once), and the partial tiles are summed by the LayerNorm that follows (layernorm_hilo2_kernel<D, true>)
  const int kpart = p.ksplit > 1 ? (int)blockIdx.z : 0;
  const int slab0 = p.ksplit > 1 ? (int)((long)kpart * (p.K / 64) / p.ksplit) : 0;
  const int nslab = p.ksplit > 1 ? (int)((long)(kpart + 1) * (p.K / 64) / p.ksplit) - slab0 : p.K / 64;
  if (p.ksplit > 1) {
#pragma unroll
    for (int b = 0; b < NB; ++b) { ap[b] += (long)slab0 * 64; wp[b] += (long)slab0 * 64; }
  }
  const int per = (nslab + 3) / 4;
  const int lo = wave * per;
  const int mine = nslab - lo < per ? (nslab - lo > 0 ? nslab - lo : 0) : per;
  const int rot = mine > 0 ? (int)((blockIdx.x * 5u + blockIdx.y * 3u) % (unsigned)mine) : 0;

  f32x4 acc[NB][NB];
#pragma unroll
  for (int i = 0; i < NB; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // The epilogue's own operands -- bias and residual of the CP output columns this thread will store -- are fetched NOW, in front of the
  // K loop: a one-utterance launch is a chain of memory round trips (arguments, operands, then bias / residual, then the store), and
  // this takes one link out of it (round 5; the launches of a 5 s utterance last 5-14 us, most of it latency).
  constexpr int CP = NB * NB, TPR = 16 / NB;
  const int e_row = tid / TPR, e_cs = (tid % TPR) * CP;
  const int e_m = m0 + e_row, e_n = n0 + e_cs;
  const bool e_live = e_m < p.M && e_n < p.N;
  const long e_idx = (z1 * p.c_z1 + z2 * p.c_z2) + (long)e_m * p.ldc + e_n;
  float e_bias[CP], e_res[CP];
#pragma unroll
  for (int j = 0; j < CP; ++j) { e_bias[j] = 0.f; e_res[j] = 0.f; }
  if (e_live) {
    if (p.bias) {
      const float* bp = p.bias + z2 * p.bias_z2 + e_n;
#pragma unroll
      for (int j = 0; j < CP; j += 4) { const float4 t = *(const float4*)(bp + j); e_bias[j] = t.x; e_bias[j + 1] = t.y; e_bias[j + 2] = t.z; e_bias[j + 3] = t.w; }
    }
    if (p.resid) {
#pragma unroll
      for (int j = 0; j < CP; j += 4) { const float4 t = *(const float4*)(p.resid + e_idx + j); e_res[j] = t.x; e_res[j + 1] = t.y; e_res[j + 2] = t.z; e_res[j + 3] = t.w; }
    }
  }

  auto load = [&](Frags<NB>& f, int i) {  // i-th slab of this wave's walk (a slab past the end is fetched again but never multiplied)
    int j = (i < mine ? i : (mine > 0 ? mine - 1 : 0)) + rot;
    if (j >= mine) j -= mine;
    const int sl = lo + j;
    const long k = (long)(sl < nslab ? sl : nslab - 1) * 64;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        f.x[ks][b] = *(const bf16x8*)(ap[b] + k + ks * 32);
        f.w[ks][b] = *(const bf16x8*)(wp[b] + k + ks * 32);
      }
  };
  auto mma = [&](const Frags<NB>& f) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int mb = 0; mb < NB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
          acc[nb][mb] = SVT_MFMA_16x16x32(f.w[ks][nb], f.x[ks][mb], acc[nb][mb]);
  };

  // Three slabs in flight.  The steady-state loop has no branch: a conditional load would make the compiler's vmcnt
  // bookkeeping merge to the conservative count at the join, which drains the younger slabs too (measured: 2.2 us per slab).
  Frags<NB> f0, f1, f2;
  load(f0, 0);
  load(f1, 1);
  load(f2, 2);
  int i = 0;
  for (; i + 6 <= mine; i += 3) {  // slabs i+3 .. i+5 all exist
    // (scheduling fences: left alone, the compiler sinks the three refills to the end of the body and the waits in front
    // of the MFMAs drain to vmcnt(0) -- one slab in flight instead of three)
    mma(f0);
    __builtin_amdgcn_sched_barrier(0);
    load(f0, i + 3);
    __builtin_amdgcn_sched_barrier(0);
    mma(f1);
    __builtin_amdgcn_sched_barrier(0);
    load(f1, i + 4);
    __builtin_amdgcn_sched_barrier(0);
    mma(f2);
    __builtin_amdgcn_sched_barrier(0);
    load(f2, i + 5);
    __builtin_amdgcn_sched_barrier(0);
  }
  int rem = mine - i;  // 0..5; the buffers hold slabs i, i+1, i+2
  if (rem >= 3) {
    mma(f0);
    if (rem >= 4) load(f0, i + 3);
    mma(f1);
    if (rem >= 5) load(f1, i + 4);
    mma(f2);
    rem -= 3;
  }
  if (rem >= 1) mma(f0);
  if (rem >= 2) mma(f1);

  // partial tile -> LDS: lane (m16 = lane & 15, q = lane >> 4) holds, for block (nb, mb), columns nb*16 + 4q .. +3 of row mb*16 + m16
  float* mypart = red + wave * (TS * PITCH);
#pragma unroll
  for (int mb = 0; mb < NB; ++mb)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) *(f32x4*)(mypart + (mb * 16 + r16) * PITCH + nb * 16 + cq * 4) = acc[nb][mb];
  __syncthreads();

  // thread t: CP = NB*NB consecutive columns of one row (16 / NB threads per row): row-contiguous stores
  const int row = e_row, cs = e_cs;
  if (!e_live) return;
  float v[CP];
#pragma unroll
  for (int j = 0; j < CP; j += 4) {
    f32x4 s = *(const f32x4*)(red + row * PITCH + cs + j);
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      const f32x4 t = *(const f32x4*)(red + w * (TS * PITCH) + row * PITCH + cs + j);
      s[0] += t[0]; s[1] += t[1]; s[2] += t[2]; s[3] += t[3];
    }
    v[j] = s[0]; v[j + 1] = s[1]; v[j + 2] = s[2]; v[j + 3] = s[3];
  }
  const long idx = e_idx;
  if (p.ksplit > 1) {   // the raw partial sum of this K part, fp32: bias / activation / residual belong to whoever adds the parts
    float* o = (float*)p.C + (long)kpart * p.ksplit_stride + idx;
#pragma unroll
    for (int j = 0; j < CP; j += 4) *(float4*)(o + j) = float4{v[j], v[j + 1], v[j + 2], v[j + 3]};
    return;
  }
#pragma unroll
  for (int j = 0; j < CP; ++j) {
    float t = v[j] * p.alpha + e_bias[j];
    t = act_apply(t, p.act);
    t += e_res[j];
    v[j] = t;
  }
  if (p.out_f32) {
    float* o = (float*)p.C + idx;
#pragma unroll
    for (int j = 0; j < CP; j += 4) *(float4*)(o + j) = float4{v[j], v[j + 1], v[j + 2], v[j + 3]};
  } else {
    bf16_t* o = (bf16_t*)p.C + idx;
#pragma unroll
    for (int j = 0; j < CP; j += 4) {
      bf16x4 t;
      t[0] = (bf16_t)v[j]; t[1] = (bf16_t)v[j + 1]; t[2] = (bf16_t)v[j + 2]; t[3] = (bf16_t)v[j + 3];
      *(bf16x4*)(o + j) = t;
    }
  }
}

template <int NB>
int launch_skinny(const GemmArgs& a, hipStream_t s) {
  constexpr int TS = 16 * NB;
  const int tiles_m = (a.M + TS - 1) / TS, tiles_n = (a.N + TS - 1) / TS;
  const double flops = 2.0 * a.M * (double)a.N * a.K * a.nz;
  const double bytes = ((double)a.M * a.K + (double)a.N * a.K) * 2 * a.nz + (double)a.M * a.N * a.nz * (a.out_f32 ? 4 : 2);
  const size_t lds_bytes = 4 * TS * (TS + 4) * sizeof(float);
  if (int r_ = ensure_dyn_lds((const void*)gemm_skinny_kernel<NB>, (int)lds_bytes)) return r_;
  prof_begin(s);
  hipLaunchKernelGGL((gemm_skinny_kernel<NB>), dim3(tiles_m * tiles_n, a.nz, a.ksplit > 1 ? a.ksplit : 1), dim3(256), lds_bytes, s, a);
  prof_end(s, flops, bytes, 1);
  SVT_LAUNCH_CHECK();
  return 0;
}

}  // namespace

int g_gemm_skinny_max_tiles = 32;  // svt_debug_set key 7 (threshold sweeps)
// Small problems only: the large-tile kernels are faster as soon as they can fill the chip.
bool gemm_skinny_eligible(const GemmArgs& a) {
  if (a.gen || a.K % 64 != 0 || a.N % 16 != 0 || !a.c_vec || a.resid_op_type || a.slope || a.act == ACT_PRELU) return false;
  const long big_tiles = (long)((a.M + 127) / 128) * ((a.N + 255) / 256) * a.nz;  // workgroups of the LDS-DMA kernel
  return big_tiles <= g_gemm_skinny_max_tiles;
}

int g_ffn2_ksplit = 1;   // svt_debug_set key 36: the encoder's FFN-2 of a small batch as a K-split launch of this kernel (api.hip)
int g_gemm_skinny_small_tiles = 96;   // svt_debug_set key 33: 32 x 32 tiles while the 64 x 64 tiling has at most this many workgroups
int launch_gemm_skinny(const GemmArgs& a, hipStream_t s) {
  if (a.ksplit > 1 && (a.bias || a.resid || a.act != ACT_NONE || !a.out_f32 || a.alpha != 1.f || (a.K / 64) < a.ksplit || a.ksplit_stride < (long)a.M * a.ldc)) {
    set_error("gemm_skinny: a K-split launch writes raw fp32 partial tiles (no bias / activation / residual)"); return -1; }
  const long tiles64 = (long)((a.M + 63) / 64) * ((a.N + 63) / 64) * a.nz;
  return tiles64 <= g_gemm_skinny_small_tiles ? launch_skinny<2>(a, s) : launch_skinny<4>(a, s);
}

}  // namespace svt

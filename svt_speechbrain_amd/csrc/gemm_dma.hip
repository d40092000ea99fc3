// bf16 dense contraction for gfx950, LDS-DMA variant: BM x 256 output tile per 512-thread workgroup (8 waves as
// 2(M) x 4(N), wave tile BM/2 x 64, v_mfma_f32_16x16x32_bf16), K consumed in 64-element (128-byte) slabs that are
// moved global -> LDS by LDS-DMA (global_load_lds_dwordx4: no VGPR staging, no ds_write).  Same contract as
// gemm.hip (GemmArgs: overlapping A rows for implicit conv, bias / activation epilogue).
//
// LDS is a ring of five 32 KiB slots holding alternating A / W units of successive K slabs; three units are in
// flight while a slab is multiplied, retired by a counted s_waitcnt vmcnt across ONE raw s_barrier per slab.
//   * DMA source mapping: 8 consecutive lanes fetch the 8 16-byte chunks of one 128-byte row (one request per line
//     for the texture addresser; a lane-per-row mapping costs one request per lane and halves the fill rate), lane
//     (row r, slot s) takes chunk s ^ r, so the row-major LDS image is XOR-swizzled and the MFMA operand read of
//     (row, chunk C) at slot C ^ row is a conflict-free ds_read_b128.
//   * W rows are fetched in MFMA order, permuted (free with per-lane DMA source addresses) so a lane ends up with
//     16 consecutive output columns of one row.
//   * two kernels share this pipeline: gemm_pp8_kernel (one tile per workgroup, LDS-transposed coalesced epilogue)
//     and gemm_pers_kernel (one workgroup per CU walks a tile list; the next tile's fills and the epilogue stores
//     overlap the MFMAs).  launch_gemm_dma picks BM in {128,192,256} and the kernel per problem.
//   * gemm_pp8_kernel<BM, GEN = true, NBW> is the same pipeline with generalised addressing (GemmArgs::gen): A rows at
//     m*stride + floor(m/d1)*e1 + floor(m/d2)*e2, K made of equal runs a fixed distance apart, C rows likewise -- the
//     3x3 / 1x1 convolutions of the lip front-end's ResNet over zero-haloed channels-last tensors -- with bias +
//     residual (operand type) + per-column PReLU in the epilogue; NBW = 2 gives a 128-column tile for 128-channel layers.
#include "common.h"
#include <map>
#include <mutex>
#include <utility>

namespace svt {
namespace {

__device__ __forceinline__ float gelu_erf(float x) { return gelu_fast(x); }
__device__ __forceinline__ float apply_act(float v, int act) {
  if (act == ACT_GELU) return gelu_erf(v);
  if (act == ACT_RELU) return v > 0.f ? v : 0.f;
  return v;
}

typedef const void __attribute__((address_space(1)))* gptr_t;
typedef void __attribute__((address_space(3)))* lptr_t;

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// ---------------------------------------------------------------------------------------------
// Coalesced epilogue.  After the MFMA loop a lane holds, per 16-row block, 16 consecutive columns of ONE row
// (lane & 15 = row): storing straight from registers makes every lane of a wave-instruction touch a different
// 128-byte line (one texture-addresser request per lane; measured 10-27 us per tile, ~45 % of the kernel).
// Instead each wave transposes its 16 x 64 block through a private LDS patch (row pitch 272 B: conflict-free both
// ways) and stores row-contiguous: 16 (fp32) / 8 (bf16) consecutive lanes cover whole 128-byte lines.  Bias,
// activation and the fp32 residual are applied on the read-back side with the same coalesced addressing.
// this lane's bias values on the read-back side of epilogue_block (column base c4 = (lane & 15) * 4 for fp32 output,
// c8 = (lane & 7) * 8 for bf16): loaded ONCE per wave and tile, not once per 16-row block (eight dependent L2 round trips)
struct BiasRegs { float v[8]; };
template <bool OUT32>
__device__ __forceinline__ BiasRegs load_bias_regs(const GemmArgs& p, const float* bias, int lane, int wn, int n0) {
  BiasRegs b;
#pragma unroll
  for (int j = 0; j < 8; ++j) b.v[j] = 0.f;
  const int n = n0 + wn * 64 + (OUT32 ? (lane & 15) * 4 : (lane & 7) * 8);
  if (bias && n < p.N) {
    const float4 b0 = *(const float4*)(bias + n);
    b.v[0] = b0.x; b.v[1] = b0.y; b.v[2] = b0.z; b.v[3] = b0.w;
    if (!OUT32) {
      const float4 b1 = *(const float4*)(bias + n + 4);
      b.v[4] = b1.x; b.v[5] = b1.y; b.v[6] = b1.z; b.v[7] = b1.w;
    }
  }
  return b;
}

template <int MB, int BM, bool OUT32>
__device__ __forceinline__ void epilogue_block(const GemmArgs& p, const f32x4& a0, const f32x4& a1, const f32x4& a2,
                                               const f32x4& a3, int mb, float* patch, int lane, int wm, int wn, int m0,
                                               int n0, long coff, const BiasRegs& br) {
  constexpr int PITCH = 68;  // floats
  const int m16 = lane & 15, q = lane >> 4;
  constexpr bool out32 = OUT32;
  {
    const f32x4 accs[4] = {a0, a1, a2, a3};
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
      f32x4 v = accs[nb];
      v[0] *= p.alpha; v[1] *= p.alpha; v[2] *= p.alpha; v[3] *= p.alpha;
      *(f32x4*)(patch + m16 * PITCH + q * 16 + nb * 4) = v;
    }
#ifndef SVT_EPI_NOFENCE
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#endif
    const int mbase = m0 + wm * (BM / 2) + mb * 16;
    if (out32) {
      const int c4 = (lane & 15) * 4;
      const int n = n0 + wn * 64 + c4;
      const float4 b4 = float4{br.v[0], br.v[1], br.v[2], br.v[3]};
#pragma unroll
      for (int pass = 0; pass < 4; ++pass) {
        const int r = pass * 4 + (lane >> 4);
        const int m = mbase + r;
        float4 v = *(const float4*)(patch + r * PITCH + c4);
        if (m < p.M && n < p.N) {
          v.x = apply_act(v.x + b4.x, p.act); v.y = apply_act(v.y + b4.y, p.act);
          v.z = apply_act(v.z + b4.z, p.act); v.w = apply_act(v.w + b4.w, p.act);
          const long idx = coff + (long)m * p.ldc + n;
          if (p.resid) {
            const float4 r4 = *(const float4*)(p.resid + idx);
            v.x += r4.x; v.y += r4.y; v.z += r4.z; v.w += r4.w;
          }
          if (p.planes) {   // (hi, lo) planes instead of fp32 (GemmArgs::planes): 8 + 8 bytes per lane, whole 128-byte lines per row
            const float x[4] = {v.x, v.y, v.z, v.w};
            unsigned short hi[4], lo[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              if (p.planes_f16) {
                const _Float16 a = (_Float16)x[j], b = (_Float16)(x[j] - (float)a);
                hi[j] = __builtin_bit_cast(unsigned short, a); lo[j] = __builtin_bit_cast(unsigned short, b);
              } else {
                const __bf16 a = (__bf16)x[j], b = (__bf16)(x[j] - (float)a);
                hi[j] = __builtin_bit_cast(unsigned short, a); lo[j] = __builtin_bit_cast(unsigned short, b);
              }
            }
            *(uint2*)(p.planes + idx) = uint2{(unsigned)hi[0] | ((unsigned)hi[1] << 16), (unsigned)hi[2] | ((unsigned)hi[3] << 16)};
            *(uint2*)(p.planes + p.plane_stride + idx) = uint2{(unsigned)lo[0] | ((unsigned)lo[1] << 16), (unsigned)lo[2] | ((unsigned)lo[3] << 16)};
          } else {
            *(float4*)((float*)p.C + idx) = v;
          }
        }
      }
    } else {
      const int c8 = (lane & 7) * 8;
      const int n = n0 + wn * 64 + c8;
      float bb[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) bb[j] = br.v[j];
#pragma unroll
      for (int pass = 0; pass < 2; ++pass) {
        const int r = pass * 8 + (lane >> 3);
        const int m = mbase + r;
        const float4 v0 = *(const float4*)(patch + r * PITCH + c8), v1 = *(const float4*)(patch + r * PITCH + c8 + 4);
        if (m < p.M && n < p.N) {
          float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
          const long idx = coff + (long)m * p.ldc + n;
          if (p.resid) {
            const float4 r0 = *(const float4*)(p.resid + idx), r1 = *(const float4*)(p.resid + idx + 4);
            const float rr[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = apply_act(v[j] + bb[j], p.act) + rr[j];
          } else if (p.act == ACT_GELU) {
            f32x2_t g[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) g[j] = f32x2_t{v[2 * j] + bb[2 * j], v[2 * j + 1] + bb[2 * j + 1]};
            gelu_bf16x2_x4(g);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              v[2 * j] = g[j].x;
              v[2 * j + 1] = g[j].y;
            }
          } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = apply_act(v[j] + bb[j], p.act);
          }
          bf16x8 o;
#pragma unroll
          for (int j = 0; j < 8; ++j) o[j] = (bf16_t)v[j];
          *(bf16x8*)((bf16_t*)p.C + idx) = o;
        }
      }
    }
#ifndef SVT_EPI_NOFENCE
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#endif
  }
}

// generalised variant (GemmArgs::gen): rows land at c_base + m*ldc + (m/c_d1)*c_e1 + (m/c_d2)*c_e2 (interior of a
// zero-haloed channels-last tensor), optional residual in the operand type added BEFORE the activation, per-column
// PReLU slope.  bf16 output only.  NBW = 16-column MFMA blocks per wave (tile width 64 * NBW).
template <int MB, int BM, int NBW, int I>
__device__ __forceinline__ void epilogue_block_gen(const GemmArgs& p, f32x4 (&acc)[NBW][MB], float* patch, int lane, int wm,
                                                   int wn, int m0, int n0, const float (&bb)[8], const float (&ss)[8]) {
  constexpr int WC = 16 * NBW;       // columns per wave
  constexpr int PITCH = WC + 4;      // floats
  constexpr int LPR = 2 * NBW;       // lanes per row on the read-back side (8 columns each)
  constexpr int RPP = 64 / LPR;      // rows per pass
  constexpr int PASSES = RPP >= 16 ? 1 : 16 / RPP;
  const int m16 = lane & 15, q = lane >> 4;
#pragma unroll
  for (int nb = 0; nb < NBW; ++nb) {
    f32x4 v = acc[nb][I];
    v[0] *= p.alpha; v[1] *= p.alpha; v[2] *= p.alpha; v[3] *= p.alpha;
    *(f32x4*)(patch + m16 * PITCH + q * (4 * NBW) + nb * 4) = v;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const int mbase = m0 + wm * (BM / 2) + I * 16;
  const int c8 = (lane % LPR) * 8;
  const int n = n0 + wn * WC + c8;
#pragma unroll
  for (int pass = 0; pass < PASSES; ++pass) {
    const int r = pass * RPP + lane / LPR;
    const int m = mbase + r;
    if (r < 16 && m < p.M && n < p.N) {
      const float4 v0 = *(const float4*)(patch + r * PITCH + c8), v1 = *(const float4*)(patch + r * PITCH + c8 + 4);
      float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
      const long idx = p.c_base + (long)m * p.ldc + (long)(m / p.c_d1) * p.c_e1 + (long)(m / p.c_d2) * p.c_e2 + n +
                       (p.c_nsplit && n >= p.c_nsplit ? p.c_nstride - p.c_nsplit : 0);
      float rr[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) rr[j] = 0.f;
      if (p.resid) {
        const bf16x8 r8 = *(const bf16x8*)((const bf16_t*)p.resid + idx);
#pragma unroll
        for (int j = 0; j < 8; ++j) rr[j] = (float)r8[j];
      }
      bf16x8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float x = v[j] + bb[j];
        if (p.resid_first) x += rr[j];
        if (p.act == ACT_PRELU) x = x > 0.f ? x : x * ss[j];
        else x = apply_act(x, p.act);
        if (!p.resid_first) x += rr[j];
        o[j] = (bf16_t)x;
      }
      *(bf16x8*)((bf16_t*)p.C + idx) = o;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

template <int MB, int BM, int NBW, int... I>
__device__ __forceinline__ void epilogue_seq_gen(std::integer_sequence<int, I...>, const GemmArgs& p, f32x4 (&acc)[NBW][MB],
                                                 float* patch, int lane, int wm, int wn, int m0, int n0, const float* bias) {
  // bias / slope of the lane's 8 columns ONCE per tile: inside the per-block function they sit behind its wave fences, where the
  // compiler re-issues the four loads for every 16-row block and every block waits out their round trip
  const int n = n0 + wn * (16 * NBW) + (lane % (2 * NBW)) * 8;
  float bb[8], ss[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { bb[j] = 0.f; ss[j] = 0.f; }
  if (n < p.N) {
    if (bias) {
      const float4 b0 = *(const float4*)(bias + n), b1 = *(const float4*)(bias + n + 4);
      bb[0] = b0.x; bb[1] = b0.y; bb[2] = b0.z; bb[3] = b0.w; bb[4] = b1.x; bb[5] = b1.y; bb[6] = b1.z; bb[7] = b1.w;
    }
    if (p.slope) {
      const float4 s0 = *(const float4*)(p.slope + n), s1 = *(const float4*)(p.slope + n + 4);
      ss[0] = s0.x; ss[1] = s0.y; ss[2] = s0.z; ss[3] = s0.w; ss[4] = s1.x; ss[5] = s1.y; ss[6] = s1.z; ss[7] = s1.w;
    }
  }
  (epilogue_block_gen<MB, BM, NBW, I>(p, acc, patch, lane, wm, wn, m0, n0, bb, ss), ...);
}

template <int MB, int BM, bool OUT32, int... I>
__device__ __forceinline__ void epilogue_seq(std::integer_sequence<int, I...>, const GemmArgs& p, f32x4 (&acc)[4][MB],
                                             float* patch, int lane, int wm, int wn, int m0, int n0, long coff,
                                             const float* bias) {
  const BiasRegs br = load_bias_regs<OUT32>(p, bias, lane, wn, n0);
  // fold over compile-time block indices: every acc[][] index is static (a runtime-indexed accumulator array
  // would be demoted to scratch)
  (epilogue_block<MB, BM, OUT32>(p, acc[0][I], acc[1][I], acc[2][I], acc[3][I], I, patch, lane, wm, wn, m0, n0, coff, br),
   ...);
}

// The same epilogue as a REAL loop over the 16-row blocks: only the accumulator selection (a scalar switch, 16 moves per case)
// differs between iterations.  Unrolled, the epilogue of a 256-row tile is ~9 000 instructions per output type (the whole
// kernel 85-110 KB against a 64 KiB instruction cache shared by two CUs), and every tile streams it once.
template <int MB, int BM, bool OUT32>
__device__ __forceinline__ void epilogue_loop(const GemmArgs& p, f32x4 (&acc)[4][MB], float* patch, int lane, int wm, int wn, int m0,
                                              int n0, long coff, const float* bias) {
  const BiasRegs br = load_bias_regs<OUT32>(p, bias, lane, wn, n0);
#pragma nounroll
  for (int mb = 0; mb < MB; ++mb) {
    f32x4 a0, a1, a2, a3;
#define SVT_PICK(I_)                                                             \
  case I_:                                                                       \
    if constexpr (I_ < MB) { a0 = acc[0][I_ < MB ? I_ : 0]; a1 = acc[1][I_ < MB ? I_ : 0]; a2 = acc[2][I_ < MB ? I_ : 0]; a3 = acc[3][I_ < MB ? I_ : 0]; } \
    break;
    switch (mb) {
      SVT_PICK(0) SVT_PICK(1) SVT_PICK(2) SVT_PICK(3) SVT_PICK(4) SVT_PICK(5) SVT_PICK(6)
      default:
        a0 = acc[0][MB - 1]; a1 = acc[1][MB - 1]; a2 = acc[2][MB - 1]; a3 = acc[3][MB - 1];
        break;
    }
#undef SVT_PICK
    asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));   // keep the selection out of the block below
    epilogue_block<MB, BM, OUT32>(p, a0, a1, a2, a3, mb, patch, lane, wm, wn, m0, n0, coff, br);
  }
}

template <int MB, int BM>
__device__ __forceinline__ void epilogue_coalesced(const GemmArgs& p, f32x4 (&acc)[4][MB], float* patch, int lane, int wm,
                                                   int wn, int m0, int n0, long coff, const float* bias) {
  if (p.out_f32)
    epilogue_seq<MB, BM, true>(std::make_integer_sequence<int, MB>{}, p, acc, patch, lane, wm, wn, m0, n0, coff, bias);
  else
    epilogue_seq<MB, BM, false>(std::make_integer_sequence<int, MB>{}, p, acc, patch, lane, wm, wn, m0, n0, coff, bias);
}

// ---------------------------------------------------------------------------------------------
// One tile per workgroup.  The LDS is a ring of 5 slots of 32 KiB; units alternate A-slab / W-slab of the same
// 64-deep K step (A_0 W_0 A_1 W_1 ...), each filled by full-line LDS-DMA (8 rows x 128 B per wave-instruction).
// While slab j is multiplied, units A_{j+1}, W_{j+1}, A_{j+2} are in flight, retired by a counted vmcnt.
template <int BM, bool GEN = false, int NBW = 4>
__global__ __launch_bounds__(512) void gemm_pp8_kernel(GemmArgs p) {
  static_assert(GEN || NBW == 4, "narrow tiles are served by the generalised variant only");
  constexpr int BN = 64 * NBW, BK = 64, NSLOT = 5;
  constexpr int MB = BM / 32;
  constexpr int GA = BM / 64;       // DMA instructions per wave per A unit (BM/8 groups over 8 waves)
  constexpr int GW = BN / 64;       // per W unit
  constexpr int SLOT = 2048;        // uint4 per slot (32 KiB)
  extern __shared__ __attribute__((aligned(16))) uint4 lds[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;

  const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
  const int nblk = tiles_m * tiles_n;
  const int bid = blockIdx.x;
  const int xcd = bid & 7, qq = nblk >> 3, rr = nblk & 7;
  const int wg = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (bid >> 3);
  const int tile_n = wg % tiles_n, tile_m = wg / tiles_n;
  const int z = blockIdx.y;
  const int z1 = z / p.nz2, z2 = z % p.nz2;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const bf16_t* A = (const bf16_t*)p.A + (z1 * p.a_z1 + z2 * p.a_z2);
  const bf16_t* W = (const bf16_t*)p.W + (z1 * p.w_z1 + z2 * p.w_z2);

  // coalesced + swizzled DMA source: 8 consecutive lanes fetch the 8 chunks of ONE 128-byte row (one line request
  // instead of eight), lane (row r8 = l>>3, slot = l&7) takes chunk (slot ^ r8); the LDS image of a group is then
  // row-major [row][slot] and the MFMA read of (row, chunk C) goes to slot C ^ row -> conflict-free ds_read_b128.
  const int r8 = lane >> 3, ch = (lane & 7) ^ (lane >> 3);
  const bf16_t* asrc[GA];
  const bf16_t* wsrc[GW];
#pragma unroll
  for (int i = 0; i < GA; ++i) {
    int m = m0 + (wave + 8 * i) * 8 + r8;
    if (m > p.M - 1) m = p.M - 1;
    if constexpr (GEN) asrc[i] = A + (long)m * p.a_rstride + (long)(m / p.a_d1) * p.a_e1 + (long)(m / p.a_d2) * p.a_e2 + ch * 8;
    else asrc[i] = A + (long)(m / p.a_rpb) * p.a_bstride + (long)(m % p.a_rpb) * p.a_rstride + ch * 8;
  }
#pragma unroll
  for (int i = 0; i < GW; ++i) {
    const int rho = (wave + 8 * i) * 8 + r8;
    const int i16 = rho & 15;
    int n = n0 + (rho / (16 * NBW)) * (16 * NBW) + (i16 >> 2) * (4 * NBW) + ((rho >> 4) % NBW) * 4 + (i16 & 3);
    if (n > p.N - 1) n = p.N - 1;
    wsrc[i] = W + (long)n * p.ldw + ch * 8;
  }
  // GEN: K is made of kseg-element runs kseg_stride apart; the element offset of the NEXT A slab to fetch is tracked
  // incrementally (aoff), a_left = slabs left in the current run
  long aoff = 0;
  const int spk = GEN ? (p.kseg ? p.kseg / BK : 0x7fffffff) : 0;
  int a_left = spk;
  auto a_advance = [&]() {
    if constexpr (GEN) {
      aoff += BK;
      if (--a_left == 0) { a_left = spk; aoff += p.kseg_stride - p.kseg; }
    }
  };
  // unit u: even -> A slab u/2, odd -> W slab u/2; slot u % 5
  auto issue_a = [&](int kt, int slot) {
#pragma unroll
    for (int i = 0; i < GA; ++i) {
      if constexpr (GEN)
        __builtin_amdgcn_global_load_lds((gptr_t)(asrc[i] + aoff), (lptr_t)(lds + slot * SLOT + (wave + 8 * i) * 64), 16, 0, 0);
      else
        __builtin_amdgcn_global_load_lds((gptr_t)(asrc[i] + kt * BK), (lptr_t)(lds + slot * SLOT + (wave + 8 * i) * 64), 16, 0, 0);
    }
    a_advance();
  };
  auto issue_w = [&](int kt, int slot) {
#pragma unroll
    for (int i = 0; i < GW; ++i)
      __builtin_amdgcn_global_load_lds((gptr_t)(wsrc[i] + kt * BK), (lptr_t)(lds + slot * SLOT + (wave + 8 * i) * 64), 16, 0, 0);
  };

  f32x4 acc[NBW][MB];
#pragma unroll
  for (int i = 0; i < NBW; ++i)
#pragma unroll
    for (int j = 0; j < MB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int cq = lane >> 4, r16 = lane & 15;
  // uint4 index inside a slot of fragment (16-row block blk, k-step ks): group (2*blk + (r16>>3)) * 64 + row*8 + slot
  const int rr8 = r16 & 7;
  const int frag0 = (r16 >> 3) * 64 + rr8 * 8 + ((cq) ^ rr8);        // ks = 0: chunk = cq
  const int frag1 = (r16 >> 3) * 64 + rr8 * 8 + ((4 + cq) ^ rr8);    // ks = 1: chunk = 4 + cq
  const int xoff = (wm * (MB * 2)) * 64;
  const int woff = (wn * 2 * NBW) * 64;

  // Staggered quarter-phase schedule ("8-phase"): the slab is multiplied in four groups of 4 x MB/2 MFMAs, each preceded
  // by a LOAD slot (its LDS fragment reads + two DMA instructions of the ring).  Slots are separated by raw barriers and
  // waves 4-7 run one slot behind waves 0-3, so on every SIMD one wave is in an MFMA slot while its partner is in a
  // LOAD slot: the matrix pipe never waits for LDS latency or DMA issue of its own wave.
  const int nk = p.K / BK;
  constexpr int HM = MB / 2;
  bf16x8 wfr[NBW], xfr[HM];
  const int grp = wave >> 2;
  const bool tr = p.trace != nullptr;
  long long t_begin = 0, t_first = 0, t_main = 0, c_first = 0, c_main = 0;  // t_*: s_memrealtime (100 MHz), c_*: s_memtime (core clock)
  if (tr) t_begin = wall_clock64();
  issue_a(0, 0);
  issue_w(0, 1);
  if (nk > 1) issue_a(1, 2);
  if (nk > 1) wait_vm<GA>(); else wait_vm<0>();
  __builtin_amdgcn_s_barrier();
  if (tr) { t_first = wall_clock64(); c_first = __builtin_amdgcn_s_memtime(); }
  int sa = 0, sw = 1;
#define SVT_LOAD(Q)                                                                                              \
  {                                                                                                              \
    constexpr int ks_ = (Q) >> 1, half_ = (Q)&1;                                                                 \
    if (half_ == 0) {                                                                                            \
      _Pragma("unroll") for (int nb = 0; nb < NBW; ++nb) wfr[nb] = __builtin_bit_cast(bf16x8, wa[nb * 128 + (ks_ ? frag1 : frag0)]); \
    }                                                                                                            \
    _Pragma("unroll") for (int jj = 0; jj < HM; ++jj)                                                            \
        xfr[jj] = __builtin_bit_cast(bf16x8, xa[(half_ * HM + jj) * 128 + (ks_ ? frag1 : frag0)]);               \
    if (ks_ == 0) {                                                                                              \
      if (kt + 1 < nk) {                                                                                         \
        _Pragma("unroll") for (int i2 = half_ * ((GW + 1) / 2); i2 < (half_ ? GW : (GW + 1) / 2); ++i2)          \
            __builtin_amdgcn_global_load_lds((gptr_t)(wsrc[i2] + (kt + 1) * BK),                                 \
                                             (lptr_t)(lds + ((2 * kt + 3) % NSLOT) * SLOT + (wave + 8 * i2) * 64), 16, 0, 0); \
      }                                                                                                          \
    } else {                                                                                                     \
      if (kt + 2 < nk) {                                                                                         \
        _Pragma("unroll") for (int i2 = half_ * ((GA + 1) / 2); i2 < (half_ ? GA : (GA + 1) / 2); ++i2)          \
            __builtin_amdgcn_global_load_lds((gptr_t)(asrc[i2] + (GEN ? aoff : (long)(kt + 2) * BK)),            \
                                             (lptr_t)(lds + ((2 * kt + 4) % NSLOT) * SLOT + (wave + 8 * i2) * 64), 16, 0, 0); \
        if (half_) a_advance();                                                                                  \
      }                                                                                                          \
    }                                                                                                            \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                                           \
  }
#define SVT_MMA(Q)                                                                                               \
  {                                                                                                              \
    constexpr int half_ = (Q)&1;                                                                                 \
    __builtin_amdgcn_s_setprio(1);                                                                               \
    _Pragma("unroll") for (int jj = 0; jj < HM; ++jj)                                                            \
      _Pragma("unroll") for (int nb = 0; nb < NBW; ++nb)                                                         \
        acc[nb][half_ * HM + jj] = SVT_MFMA_16x16x32(wfr[nb], xfr[jj], acc[nb][half_ * HM + jj]); \
    __builtin_amdgcn_s_setprio(0);                                                                               \
    __builtin_amdgcn_sched_barrier(0);                                                                           \
  }
#define SVT_RETIRE_NEXT()                                                                                        \
  {                                                                                                              \
    if (kt + 2 < nk) wait_vm<GA>();                                                                              \
    else if (kt + 1 < nk) wait_vm<0>();                                                                          \
  }
  if (grp == 0) {
    for (int kt = 0; kt < nk; ++kt) {
      const uint4* xa = lds + sa * SLOT + xoff;
      const uint4* wa = lds + sw * SLOT + woff;
      SVT_LOAD(0) __builtin_amdgcn_s_barrier(); SVT_MMA(0) __builtin_amdgcn_s_barrier();
      SVT_LOAD(1) __builtin_amdgcn_s_barrier(); SVT_MMA(1) __builtin_amdgcn_s_barrier();
      SVT_LOAD(2) __builtin_amdgcn_s_barrier(); SVT_MMA(2) __builtin_amdgcn_s_barrier();
      SVT_LOAD(3) __builtin_amdgcn_s_barrier(); SVT_MMA(3)
      SVT_RETIRE_NEXT()
      __builtin_amdgcn_s_barrier();
      sa = (sa + 2) % NSLOT;
      sw = (sw + 2) % NSLOT;
    }
  } else {
    __builtin_amdgcn_s_barrier();  // one slot behind
    for (int kt = 0; kt < nk; ++kt) {
      const uint4* xa = lds + sa * SLOT + xoff;
      const uint4* wa = lds + sw * SLOT + woff;
      SVT_LOAD(0) __builtin_amdgcn_s_barrier(); SVT_MMA(0) __builtin_amdgcn_s_barrier();
      SVT_LOAD(1) __builtin_amdgcn_s_barrier(); SVT_MMA(1) __builtin_amdgcn_s_barrier();
      SVT_LOAD(2) __builtin_amdgcn_s_barrier(); SVT_MMA(2) __builtin_amdgcn_s_barrier();
      SVT_LOAD(3)
      SVT_RETIRE_NEXT()
      __builtin_amdgcn_s_barrier();
      SVT_MMA(3)
      if (kt + 1 < nk) __builtin_amdgcn_s_barrier();
      sa = (sa + 2) % NSLOT;
      sw = (sw + 2) % NSLOT;
    }
  }
#undef SVT_LOAD
#undef SVT_MMA
#undef SVT_RETIRE_NEXT
  // ---- epilogue ----
  const long coff = z1 * p.c_z1 + z2 * p.c_z2;
  const float* bias = p.bias ? p.bias + z2 * p.bias_z2 : nullptr;
  if (tr) { t_main = wall_clock64(); c_main = __builtin_amdgcn_s_memtime(); }
  __syncthreads();  // every wave is done with the ring before it is reused as transpose patches
  if constexpr (GEN)
    epilogue_seq_gen<MB, BM, NBW>(std::make_integer_sequence<int, MB>{}, p, acc, (float*)lds + wave * (16 * (16 * NBW + 4)), lane, wm, wn,
                                  m0, n0, bias);
  else {
    if constexpr (NBW == 4) {
      if (p.dbg != 3)
        epilogue_coalesced<MB, BM>(p, acc, (float*)lds + wave * (16 * 68), lane, wm, wn, m0, n0, coff, bias);
      else if (acc[0][0][0] == 123.456f) ((float*)p.C)[0] = 1.f;
    }
  }
  if (tr && lane == 0 && (wave & 3) == 0) {
    long long* o = p.trace + ((long)blockIdx.x * 2 + (wave >> 2)) * 8;
    o[0] = t_begin; o[1] = t_first; o[2] = t_main - t_first; o[3] = wall_clock64() - t_main; o[4] = wall_clock64(); o[5] = 1;
    o[6] = c_main - c_first; o[7] = BM;
  }
}


// ---------------------------------------------------------------------------------------------
// Persistent variant of the unit-ring kernel.  One workgroup per CU walks a list of output tiles; the
// (tile, k-slab) pairs form ONE stream for the LDS-DMA ring, so the first slabs of tile i+1 are already
// in flight while tile i runs its epilogue, and the epilogue's global stores (fire-and-forget) drain
// while the next tile multiplies.  Without this every CU of a single-round launch reaches its epilogue
// at the same moment and the 50-100 MB store burst is pure serial time (measured: 40 % of out_proj).
// The epilogue uses buffer stores with hardware bounds checking (rows >= M are dropped by the memory
// pipeline, never by a branch), so the number of VMEM operations a wave has in flight after an epilogue
// is a compile-time constant and the counted s_waitcnt vmcnt of the ring stays exact.
template <int BM>
__global__ __launch_bounds__(512) void gemm_pers_kernel(GemmArgs p, int tiles_n, int ntiles) {
  constexpr int BN = 256, BK = 64, NSLOT = 5;
  constexpr int MB = BM / 32;
  constexpr int GA = BM / 64;
  constexpr int GW = BN / 64;
  constexpr int SLOT = 2048;
  extern __shared__ __attribute__((aligned(16))) uint4 lds[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int nblk = gridDim.x, b = blockIdx.x;
  // blocks b and b+8 share an XCD: in every round an XCD works on nblk/8 consecutive logical tiles (n fastest)
  const int per = nblk >> 3;
  const int lbase = (b & 7) * per + (b >> 3);
  int my_tiles = 0;
  while (my_tiles * nblk + lbase < ntiles) ++my_tiles;
  if (my_tiles == 0) return;

  const bf16_t* A = (const bf16_t*)p.A;
  const bf16_t* W = (const bf16_t*)p.W;
  const int r8 = lane >> 3, ch = (lane & 7) ^ (lane >> 3);

  const bf16_t* asrc[GA];
  const bf16_t* wsrc[GW];
  const bf16_t* asrc2[GA];
  const bf16_t* wsrc2[GW];
  auto setup = [&](int logical, const bf16_t* (&as)[GA], const bf16_t* (&ws)[GW]) {
    const int tile_n = logical % tiles_n, tile_m = logical / tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
#pragma unroll
    for (int i = 0; i < GA; ++i) {
      int m = m0 + (wave + 8 * i) * 8 + r8;
      if (m > p.M - 1) m = p.M - 1;
      as[i] = A + (long)(m / p.a_rpb) * p.a_bstride + (long)(m % p.a_rpb) * p.a_rstride + ch * 8;
    }
#pragma unroll
    for (int i = 0; i < GW; ++i) {
      const int rho = (wave + 8 * i) * 8 + r8;
      const int i16 = rho & 15;
      int n = n0 + (rho >> 6) * 64 + (i16 >> 2) * 16 + ((rho >> 4) & 3) * 4 + (i16 & 3);
      if (n > p.N - 1) n = p.N - 1;
      ws[i] = W + (long)n * p.ldw + ch * 8;
    }
  };
  auto issue_a = [&](const bf16_t* const (&as)[GA], int kt, int slot) {
#pragma unroll
    for (int i = 0; i < GA; ++i)
      __builtin_amdgcn_global_load_lds((gptr_t)(as[i] + kt * BK), (lptr_t)(lds + slot * SLOT + (wave + 8 * i) * 64), 16, 0, 0);
  };
  auto issue_w = [&](const bf16_t* const (&ws)[GW], int kt, int slot) {
#pragma unroll
    for (int i = 0; i < GW; ++i)
      __builtin_amdgcn_global_load_lds((gptr_t)(ws[i] + kt * BK), (lptr_t)(lds + slot * SLOT + (wave + 8 * i) * 64), 16, 0, 0);
  };
  // one DMA instruction of a unit (the steady-state loop spreads a unit's instructions between MFMA groups: a burst
  // of 8 per wave right after the barrier queues 64 requests on the CU's texture addresser in front of every wave)
  auto issue_a1 = [&](const bf16_t* const (&as)[GA], int i, int kt, int slot) {
    __builtin_amdgcn_global_load_lds((gptr_t)(as[i] + kt * BK), (lptr_t)(lds + slot * SLOT + (wave + 8 * i) * 64), 16, 0, 0);
  };
  auto issue_w1 = [&](const bf16_t* const (&ws)[GW], int i, int kt, int slot) {
    __builtin_amdgcn_global_load_lds((gptr_t)(ws[i] + kt * BK), (lptr_t)(lds + slot * SLOT + (wave + 8 * i) * 64), 16, 0, 0);
  };

  f32x4 acc[4][MB];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < MB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int cq = lane >> 4, r16 = lane & 15, rr8 = r16 & 7;
  const int frag0 = (r16 >> 3) * 64 + rr8 * 8 + (cq ^ rr8);
  const int frag1 = (r16 >> 3) * 64 + rr8 * 8 + ((4 + cq) ^ rr8);
  const int xoff = (wm * (MB * 2)) * 64;
  const int woff = (wn * 8) * 64;

  const bool tr = p.trace != nullptr;
  long long t_begin = 0, t_first = 0, t_main = 0, t_epi = 0, t_mark = 0;
  if (tr) t_begin = wall_clock64();
  const int nk = p.K / BK;            // >= 2 (checked by the launcher)
  const int G = my_tiles * nk;        // slabs in this workgroup's stream
  int ti = 0;                         // index of the tile being multiplied
  setup(lbase, asrc, wsrc);
  if (my_tiles > 1) setup(nblk + lbase, asrc2, wsrc2);
  issue_a(asrc, 0, 0);
  issue_w(wsrc, 0, 1);
  issue_a(asrc, 1, 2);
  int sa = 0, sw = 1, kt = 0;
  bool after_epilogue = false;
  const int n_store = MB * (p.out_f32 ? 4 : 2);  // buffer stores per wave per epilogue
  // static priority for the younger half of the workgroup (waves 4-7 share SIMDs with 0-3 and lose issue arbitration
  // by age: measured 39 % vs 7 % of the time parked at the barrier)
  if (wave >= 4 && p.dbg == 8) __builtin_amdgcn_s_setprio(1);
  for (int g = 0; g < G; ++g) {
    // retire slab g: allowed in flight = A unit of slab g+1 (+ the previous tile's epilogue stores, which are younger)
    if (g + 1 < G) {
      if (!after_epilogue) wait_vm<GA>();
      else if (p.out_f32) wait_vm<GA + MB * 4>();
      else wait_vm<GA + MB * 2>();
    } else {
      wait_vm<0>();
    }
    after_epilogue = false;
    __builtin_amdgcn_s_barrier();
    if (tr && g == 0) t_first = t_mark = wall_clock64();
    const uint4* xa = lds + sa * SLOT + xoff;
    const uint4* wa = lds + sw * SLOT + woff;
    const bool have_w = g + 1 < G, have_a = g + 2 < G;
    const bool w_cur = kt + 1 < nk, a_cur = kt + 2 < nk;
    const int wslot = (2 * g + 3) % NSLOT, aslot = (2 * g + 4) % NSLOT;
    const int wkt = w_cur ? kt + 1 : 0, akt = a_cur ? kt + 2 : kt + 2 - nk;
    {
      // Quarter-phase software pipeline: the slab is multiplied in four groups of 4 x MB/2 MFMAs ((k-step, M half));
      // the fragments of group q+1 are requested BEFORE the MFMAs of group q (two register sets, static indices), so
      // only the first group's LDS latency is exposed after the barrier.  The ring's DMA instructions are spread two
      // per group.
      constexpr int HM = MB / 2;
      bf16x8 wfr[2][4], xfr[2][HM];
      auto rd_w = [&](int ks, bf16x8 (&w)[4]) {
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) w[nb] = __builtin_bit_cast(bf16x8, wa[nb * 128 + (ks ? frag1 : frag0)]);
      };
      auto rd_x = [&](int ks, int half, bf16x8 (&x)[HM]) {
#pragma unroll
        for (int j = 0; j < HM; ++j) x[j] = __builtin_bit_cast(bf16x8, xa[(half * HM + j) * 128 + (ks ? frag1 : frag0)]);
      };
      rd_w(0, wfr[0]);
      rd_x(0, 0, xfr[0]);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int ks = q >> 1, half = q & 1;
        // DMA: W unit during k-step 0, A unit during k-step 1
        if (ks == 0) {
          if (have_w) {
#pragma unroll
            for (int i2 = half * 2; i2 < half * 2 + 2; ++i2) {
              if (w_cur) issue_w1(wsrc, i2, wkt, wslot); else issue_w1(wsrc2, i2, wkt, wslot);
            }
          }
        } else {
          if (have_a) {
#pragma unroll
            for (int i2 = half * ((GA + 1) / 2); i2 < (half ? GA : (GA + 1) / 2); ++i2) {
              if (a_cur) issue_a1(asrc, i2, akt, aslot); else issue_a1(asrc2, i2, akt, aslot);
            }
          }
        }
        // request the next group's fragments
        if (q == 0) rd_x(0, 1, xfr[1]);
        if (q == 1) { rd_w(1, wfr[1]); rd_x(1, 0, xfr[0]); }
        if (q == 2) rd_x(1, 1, xfr[1]);
#pragma unroll
        for (int j = 0; j < HM; ++j)
#pragma unroll
          for (int nb = 0; nb < 4; ++nb)
            acc[nb][half * HM + j] = SVT_MFMA_16x16x32(wfr[ks][nb], xfr[half][j], acc[nb][half * HM + j]);
      }
    }
    sa = (sa + 2) % NSLOT;
    sw = (sw + 2) % NSLOT;
    if (++kt == nk) {
      // ---- epilogue of tile ti (registers -> global, bounds-checked buffer stores) ----
      kt = 0;
      if (tr) { const long long t = wall_clock64(); t_main += t - t_mark; t_mark = t; }
      const int logical = ti * nblk + lbase;
      const int tile_n = logical % tiles_n, tile_m = logical / tiles_n;
      const int m0 = tile_m * BM, n0 = tile_n * BN;
      const int esz = p.out_f32 ? 4 : 2;
      // descriptor over the rows [m0, M) of C (and of the residual): a row >= M lands beyond num_records
      const long rows_left = (long)p.M - m0;
      const unsigned long nbytes = (unsigned long)rows_left * p.ldc * esz;
      const unsigned nrec = nbytes > 0xFFFFFFF0ul ? 0xFFFFFFF0u : (unsigned)nbytes;
      char* cbase = (char*)p.C + (long)m0 * p.ldc * esz;
      const auto crsrc = __builtin_amdgcn_make_buffer_rsrc(cbase, 0, nrec, 0x00020000);
      const int nbase = n0 + wn * 64 + (lane >> 4) * 16;
      // one branch around the four bias loads (hipcc waits vmcnt(0) at the first use of an ordinary load while LDS-DMA is in
      // flight; a branch per load made that four serial waits per tile -- measured: no difference, the waits overlap the
      // epilogue's own latency; kept because it is the simpler code)
      float bv[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) bv[j] = 0.f;
      if (p.bias) {
        const float4* bp = (const float4*)(p.bias + nbase);
        const float4 b0 = bp[0], b1 = bp[1], b2 = bp[2], b3 = bp[3];
        bv[0] = b0.x; bv[1] = b0.y; bv[2] = b0.z; bv[3] = b0.w; bv[4] = b1.x; bv[5] = b1.y; bv[6] = b1.z; bv[7] = b1.w;
        bv[8] = b2.x; bv[9] = b2.y; bv[10] = b2.z; bv[11] = b2.w; bv[12] = b3.x; bv[13] = b3.y; bv[14] = b3.z; bv[15] = b3.w;
      }
      if (p.out_f32) {
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
          const int ml = wm * (BM / 2) + mb * 16 + (lane & 15);
          const unsigned off = (unsigned)(((long)ml * p.ldc + nbase) * 4);
#pragma unroll
          for (int nb = 0; nb < 4; ++nb) {
            f32x4 v;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = acc[nb][mb][r] * p.alpha + bv[nb * 4 + r];
            // activation selected once per block of four (wave-uniform), not per element
            if (p.act == ACT_GELU) {
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] = gelu_erf(v[r]);
            } else if (p.act == ACT_RELU) {
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] = v[r] > 0.f ? v[r] : 0.f;
            }
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, v), crsrc, off + nb * 16, 0, 0);
          }
        }
      } else {
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
          const int ml = wm * (BM / 2) + mb * 16 + (lane & 15);
          const unsigned off = (unsigned)(((long)ml * p.ldc + nbase) * 2);
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            bf16x8 o;
            if (p.act == ACT_GELU) {
              f32x2_t g[4];
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                const int n0r = h * 8 + 2 * j, n1r = n0r + 1;
                g[j] = f32x2_t{acc[n0r >> 2][mb][n0r & 3] * p.alpha + bv[n0r], acc[n1r >> 2][mb][n1r & 3] * p.alpha + bv[n1r]};
              }
              gelu_bf16x2_x4(g);
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                o[2 * j] = (bf16_t)g[j].x;
                o[2 * j + 1] = (bf16_t)g[j].y;
              }
            } else if (p.act == ACT_RELU) {
#pragma unroll
              for (int j = 0; j < 8; ++j) {
                const int nbr = h * 8 + j;
                const float x = acc[nbr >> 2][mb][nbr & 3] * p.alpha + bv[nbr];
                o[j] = (bf16_t)(x > 0.f ? x : 0.f);
              }
            } else {
#pragma unroll
              for (int j = 0; j < 8; ++j) {
                const int nbr = h * 8 + j;
                o[j] = (bf16_t)(acc[nbr >> 2][mb][nbr & 3] * p.alpha + bv[nbr]);
              }
            }
            if (p.dbg != 10) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, o), crsrc, off + h * 16, 0, 0);
            else asm volatile("" ::"v"(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, o)));
          }
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < MB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (tr) { const long long t = wall_clock64(); t_epi += t - t_mark; t_mark = t; }
      after_epilogue = true;
      ++ti;
      // rotate the pointer sets: next tile becomes current, precompute the one after
#pragma unroll
      for (int i = 0; i < GA; ++i) asrc[i] = asrc2[i];
#pragma unroll
      for (int i = 0; i < GW; ++i) wsrc[i] = wsrc2[i];
      if (ti + 1 < my_tiles && p.dbg != 11) setup((ti + 1) * nblk + lbase, asrc2, wsrc2);
    }
  }
  (void)n_store;
  if (tr && lane == 0 && (wave & 3) == 0) {
    wait_vm<0>();
    long long* o = p.trace + ((long)blockIdx.x * 2 + (wave >> 2)) * 8;
    o[0] = t_begin; o[1] = t_first; o[2] = t_main; o[3] = t_epi; o[4] = wall_clock64(); o[5] = my_tiles;
  }
}

// ---------------------------------------------------------------------------------------------
// Split-operand products (precision "bf16x3" / "fp16x3") on the LDS-DMA pipeline.  A stays fp32 in memory; the weight
// matrix was cut ONCE (svt_*_finalize -> split_weights_register) into 16-bit (hi, lo) pieces stored per row and 32-deep
// K slab as [hi k0..31 | lo k0..31] (128 bytes, the same bytes as the fp32 row).  Tile 256 x 256, K in 32-element slabs:
// an A unit is 256 rows x 128 B of fp32, a W unit 256 rows x 128 B of pieces, both moved by full-line LDS-DMA into the
// five-slot ring of the bf16 kernels (three units in flight across one raw barrier per slab, counted vmcnt).
// Wave layout 8 (M) x 1 (N): a wave owns 32 rows x all 256 columns, so every A element is cut into its pieces by exactly
// one wave (16 values per lane and slab: ~45 VALU instructions against 96 MFMAs), while the W pieces come out of LDS
// ready-made.  Per 16 x 16 x 32 block: Wl*Xh + Wh*Xl + Wh*Xh accumulated in fp32 (three MFMAs of 16 cycles; the exact
// fp32 form is eight of 32).  Epilogue: the LDS-transposed coalesced fp32 store of the bf16 kernels (bias, activation,
// residual).  The register-staged split kernel of gemm.hip (which cuts BOTH operands in every workgroup, four waves in
// lockstep around one barrier per slab) ran at 185-212 TFLOP/s on these shapes.
// LDS-DMA issued from inline asm (M0 = LDS byte address of the wave's 1 KiB piece, saved and restored around it).  hipcc
// tracks the LDS-DMA it emits itself for a builtin and puts an s_waitcnt vmcnt(0) in front of a later LDS read whose
// address it cannot prove distinct from the DMA's destination -- here the W fragment reads of every slab, i.e. the ring
// would be drained once per slab.  Through asm the compiler sees no LDS write; the counted vmcnt + barrier below order it.
// (the lockstep kernel built on this plan in round 2, gemm_x3_kernel, is gone: git history; its staggered successor follows)

// ---------------------------------------------------------------------------------------------
// Staggered form of gemm_x3_kernel (round 3).  The lockstep kernel above kept the matrix pipe busy for 0.43 of its cycles: all eight
// waves read their W fragments, issue their LDS-DMA and cut their A pieces at the same moments, so the two waves of a SIMD never
// cover each other.  Here waves 4-7 run ONE SLOT behind waves 0-3, exactly like gemm_pp8_kernel: a 32-deep slab is four MMA slots of
// NBS W blocks x 2 row blocks x 3 products (24 MFMAs at NBS = 4) and four LOAD slots that carry the next slot's W fragment reads
// (hi + lo pieces), two LDS-DMA instructions of the ring, and -- for the NEXT slab -- the raw fp32 reads of this wave's A rows (slot 0)
// and their cuts into (hi, lo) pieces (slots 1 and 2).  On every SIMD one wave multiplies while its partner loads.
// Ring (five 32 KiB slots, unit u = 2g (A_g) / 2g + 1 (W_g) in slot u % 5): slab g reads W_g and, for cutting, A_{g+1}; it requests
// A_{g+2} during its first two LOAD slots (into the slot W_{g-1} left) and W_{g+2} during the last two (into the slot of A_g, which has
// lived in registers since slab g - 1); the counted wait that retires the slab leaves only W_{g+2} in flight.  Requests past the last
// slab re-read slab nk - 1 (never used) so that the counts stay constant; one vmcnt(0) drains them before the LDS-transposed epilogue.
// NBS = 4: 256-column tiles; NBS = 3: 192-column tiles (N = 768: 63 x 4 = 252 tiles fill the 256 CUs; 256-column tiles give 189).
typedef unsigned x3_u32x4 __attribute__((ext_vector_type(4)));   // register image of a 16-byte fragment (an ext vector: usable as an asm operand)
// DBG (diagnostic builds, svt_debug_set key 3 = 31 / 33): 1 = no LDS-DMA after the head of the stream, 3 = no epilogue
template <bool F16, int NBS, int DBG = 0>
__global__ __launch_bounds__(512) void gemm_x3s_kernel(GemmArgs p, const void* wsplit) {
  constexpr int BM = 256, BN = 64 * NBS, BK = 32, NSLOT = 5, SLOT = 2048, GA = 4, GW = NBS;
  constexpr int NB = 4 * NBS;   // 16-column W blocks per tile
  extern __shared__ __attribute__((aligned(16))) uint4 lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
  const int nblk = tiles_m * tiles_n;
  const int bid = blockIdx.x;
  const int xcd = bid & 7, qq = nblk >> 3, rr = nblk & 7;
  const int wg = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (bid >> 3);
  const int tile_n = wg % tiles_n, tile_m = wg / tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  // batch (blockIdx.y = z = z1 * nz2 + z2, the grouped positional conv): element offsets of the fp32 problem; a packed weight row
  // takes the bytes of its fp32 row
  const int zb = blockIdx.y, z1 = zb / p.nz2, z2 = zb - z1 * p.nz2;
  const float* A = (const float*)p.A + ((long)z1 * p.a_z1 + (long)z2 * p.a_z2);
  const unsigned short* W = (const unsigned short*)wsplit + 2 * ((long)z1 * p.w_z1 + (long)z2 * p.w_z2);  // [N][K / 32][64]: 32 hi pieces, 32 lo pieces
  const long czoff = (long)z1 * p.c_z1 + (long)z2 * p.c_z2;
  const int r8 = lane >> 3, ch = (lane & 7) ^ (lane >> 3);
  // ring requests in the `voffset + SGPR base` form (as gemm_x3p_kernel): a 64-bit scalar base per operand (A: the tile's first row -- the
  // tensor may exceed 4 GiB, a tile's span may not: gemm_x3p_eligible), one 32-bit offset per lane and request, the K advance scalar.
  // With per-lane 64-bit pointers every request cost two vector additions in its LOAD slot, ~70 cycles per slot under the partner's
  // MFMA issue: the slots with requests ran 356-376 cycles against 288 of MFMAs (profiles/r03_gemm_x3_slots.txt).
  auto a_row_off = [&](int m) -> long { return ((long)(m / p.a_rpb) * p.a_bstride + (long)(m % p.a_rpb) * p.a_rstride) * 4; };
  const long a_o0 = a_row_off(m0);
  const char* abase = (const char*)A + a_o0;
  const char* wbase = (const char*)W;
  unsigned aoff[GA], woff[GW];
#pragma unroll
  for (int i = 0; i < GA; ++i) {
    int m = m0 + (wave + 8 * i) * 8 + r8;
    if (m > p.M - 1) m = p.M - 1;
    aoff[i] = (unsigned)(a_row_off(m) - a_o0) + ch * 16;
  }
#pragma unroll
  for (int i = 0; i < GW; ++i) {
    const int rho = (wave + 8 * i) * 8 + r8;
    const int i16 = rho & 15;
    int n = n0 + (rho >> 6) * 64 + (i16 >> 2) * 16 + ((rho >> 4) & 3) * 4 + (i16 & 3);
    if (n > p.N - 1) n = p.N - 1;
    woff[i] = (unsigned)((long)n * p.K * 4 + ch * 16);
  }
  auto dma_sv = [](unsigned voff, const void* sbase, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_addr), "v"(voff), "s"(sbase) : "memory");
  };
  const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lptr_t)lds);
  auto unit_addr = [&](int slot, int i) -> unsigned {
    return __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(slot * SLOT + (wave + 8 * i) * 64) * 16u);
  };
  const int nk = p.K / BK;
  auto kc = [&](int k) { return k < nk ? k : nk - 1; };   // requests past the end re-read the last slab
  f32x4 acc[NB][2];
#pragma unroll
  for (int i = 0; i < NB; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int cq = lane >> 4, r16 = lane & 15, rr8 = r16 & 7;
  const int rowb = (r16 >> 3) * 64 + rr8 * 8;
  const int fa0 = rowb + ((2 * cq) ^ rr8), fa1 = rowb + ((2 * cq + 1) ^ rr8);
  const int fwh = rowb + (cq ^ rr8), fwl = rowb + ((4 + cq) ^ rr8);
  const x3_u32x4* ldsv = (const x3_u32x4*)lds;
  auto mma = [&](const x3_u32x4& a, const x3_u32x4& b, const f32x4& c) -> f32x4 {
    if constexpr (F16) {
      typedef _Float16 f16x8v __attribute__((ext_vector_type(8)));
      return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8v, a), __builtin_bit_cast(f16x8v, b), c, 0, 0, 0);
    } else {
      return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(real_bf16x8, a), __builtin_bit_cast(real_bf16x8, b), c, 0, 0, 0);
    }
  };
  auto cut = [&](const x3_u32x4& r0, const x3_u32x4& r1, x3_u32x4& hi, x3_u32x4& lo) {
    const f32x4 v0 = __builtin_bit_cast(f32x4, r0), v1 = __builtin_bit_cast(f32x4, r1);
    if constexpr (F16) {
      typedef _Float16 f16x8v __attribute__((ext_vector_type(8)));
      f16x8v h, l;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        h[j] = (_Float16)v0[j]; l[j] = (_Float16)(v0[j] - (float)h[j]);
        h[4 + j] = (_Float16)v1[j]; l[4 + j] = (_Float16)(v1[j] - (float)h[4 + j]);
      }
      hi = __builtin_bit_cast(x3_u32x4, h);
      lo = __builtin_bit_cast(x3_u32x4, l);
    } else {
      bf16x8 h, l;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        h[j] = (bf16_t)v0[j]; l[j] = (bf16_t)(v0[j] - (float)h[j]);
        h[4 + j] = (bf16_t)v1[j]; l[4 + j] = (bf16_t)(v1[j] - (float)h[4 + j]);
      }
      hi = __builtin_bit_cast(x3_u32x4, h);
      lo = __builtin_bit_cast(x3_u32x4, l);
    }
  };
  // head of the stream: A_0 -> slot 0, W_0 -> slot 1, A_1 -> slot 2, W_1 -> slot 3; everything but W_1 landed before slab 0
#pragma unroll
  for (int i = 0; i < GA; ++i) dma_sv(aoff[i], abase, unit_addr(0, i));
#pragma unroll
  for (int i = 0; i < GW; ++i) dma_sv(woff[i], wbase, unit_addr(1, i));
#pragma unroll
  for (int i = 0; i < GA; ++i) dma_sv(aoff[i], abase + (long)kc(1) * (BK * 4), unit_addr(2, i));
#pragma unroll
  for (int i = 0; i < GW; ++i) dma_sv(woff[i], wbase + (long)kc(1) * 128, unit_addr(3, i));
  wait_vm<GW>();
  __builtin_amdgcn_s_barrier();
  x3_u32x4 xh[2], xl[2], nh[2], nl[2], raw[2][2], wh[NBS], wl[NBS];
#pragma unroll
  for (int mb = 0; mb < 2; ++mb) cut(ldsv[(wave * 2 + mb) * 128 + fa0], ldsv[(wave * 2 + mb) * 128 + fa1], xh[mb], xl[mb]);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();   // every wave holds its pieces of A_0 before the first slab's requests may reuse slots
  int sw = 1;                     // slot of W_g; A_{g+1} sits in the next one
  const int grp = wave >> 2;
  // DBG 11..14: s_memtime at both ends of slots 2 (DBG - 11), 2 (DBG - 11) + 1 of slab nk / 2 (tools/gemm_trace.py --x3-slots --one-tile)
  constexpr int STQ = DBG >= 11 && DBG <= 14 ? DBG - 10 : 0;
  unsigned sraw[5], sst[5];
  const int gs = nk / 2;
  if constexpr (STQ != 0) {
#pragma unroll
    for (int i = 0; i < 5; ++i) sst[i] = 0;
  }
#define X3S_S(k)                                                                                                    \
  if constexpr (STQ != 0 && (k) >= 4 * (STQ - 1) && (k) <= 4 * (STQ - 1) + 4 && (k) < 16)                           \
    sraw[((k) - 4 * (STQ - 1)) % 5] = (unsigned)__builtin_amdgcn_s_memtime();                                       \
  if constexpr (STQ == 4 && (k) == 0) sraw[4] = (unsigned)__builtin_amdgcn_s_memtime();
#define X3S_COMMIT()                                                                                                \
  if constexpr (STQ != 0) {                                                                                         \
    if (g == gs) { _Pragma("unroll") for (int i = 0; i < (STQ == 4 ? 4 : 5); ++i) sst[i] = sraw[i]; }               \
    if (STQ == 4 && g == gs + 1) sst[4] = sraw[4];                                                                  \
  }
#define X3S_LOAD(Q)                                                                                                 \
  {                                                                                                                 \
    _Pragma("unroll") for (int i = 0; i < NBS; ++i) {                                                               \
      wh[i] = wa[((Q) * NBS + i) * 128 + fwh];                                                                      \
      wl[i] = wa[((Q) * NBS + i) * 128 + fwl];                                                                      \
    }                                                                                                               \
    if ((Q) == 0) {                                                                                                 \
      _Pragma("unroll") for (int mb = 0; mb < 2; ++mb) { raw[mb][0] = xn[mb * 128 + fa0]; raw[mb][1] = xn[mb * 128 + fa1]; } \
    }                                                                                                               \
    if (DBG == 1) {                                                                                                 \
    } else if ((Q) < 2) {   /* A_{g+2} -> the slot W_{g-1} left */                                                  \
      _Pragma("unroll") for (int i2 = (Q) * 2; i2 < (Q) * 2 + 2; ++i2)                                              \
          dma_sv(aoff[i2], abase + (long)kc(g + 2) * (BK * 4), unit_addr(s3, i2));                  \
    } else {         /* W_{g+2} -> the slot of A_g */                                                               \
      _Pragma("unroll") for (int i2 = ((Q) - 2) * 2; i2 < ((Q) == 2 ? 2 : GW); ++i2)                                \
          dma_sv(woff[i2], wbase + (long)kc(g + 2) * 128, unit_addr(s4, i2));                        \
    }                                                                                                               \
    __builtin_amdgcn_sched_barrier(0);                                                                              \
    /* The cuts run BEHIND the slot's LDS reads and ring requests, under the LDS latency: while the partner wave issues its MFMAs \
       this wave's vector instructions get every other issue slot, so the 18 instructions of a cut take ~300 cycles (slot stamps \
       of gemm_x3p_kernel, tools/gemm_trace.py --x3-slots).  The cut pieces are "used" HERE: their only real use is the copy at the \
       end of the slab, and LLVM sinks a computation to its use -- both cuts ended up in the loop latch, behind the slab's last    \
       barrier, in front of the next LOAD slot 0 */                                                                      \
    if ((Q) == 1) { cut(raw[0][0], raw[0][1], nh[0], nl[0]); asm volatile("" : "+v"(nh[0]), "+v"(nl[0])); }         \
    if ((Q) == 2) { cut(raw[1][0], raw[1][1], nh[1], nl[1]); asm volatile("" : "+v"(nh[1]), "+v"(nl[1])); }         \
    __builtin_amdgcn_sched_barrier(0);                                                                              \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                              \
    if ((Q) == 0) asm volatile("" : "+v"(raw[0][0]), "+v"(raw[0][1]), "+v"(raw[1][0]), "+v"(raw[1][1]));            \
    _Pragma("unroll") for (int i = 0; i < NBS; ++i) asm volatile("" : "+v"(wh[i]), "+v"(wl[i]));                    \
    __builtin_amdgcn_sched_barrier(0);                                                                              \
  }
#define X3S_MMA(Q)                                                                                                  \
  {                                                                                                                 \
    __builtin_amdgcn_s_setprio(1);                                                                                  \
    _Pragma("unroll") for (int i = 0; i < NBS; ++i) {                                                               \
      _Pragma("unroll") for (int mb = 0; mb < 2; ++mb) acc[(Q) * NBS + i][mb] = mma(wl[i], xh[mb], acc[(Q) * NBS + i][mb]); \
      _Pragma("unroll") for (int mb = 0; mb < 2; ++mb) acc[(Q) * NBS + i][mb] = mma(wh[i], xl[mb], acc[(Q) * NBS + i][mb]); \
      _Pragma("unroll") for (int mb = 0; mb < 2; ++mb) acc[(Q) * NBS + i][mb] = mma(wh[i], xh[mb], acc[(Q) * NBS + i][mb]); \
    }                                                                                                               \
    __builtin_amdgcn_s_setprio(0);                                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                              \
  }
#define X3S_VARS()                                                                                                  \
  const x3_u32x4* wa = ldsv + sw * SLOT;                                                                            \
  const int s1 = sw + 1 >= NSLOT ? sw + 1 - NSLOT : sw + 1, s3 = sw + 3 >= NSLOT ? sw + 3 - NSLOT : sw + 3,         \
            s4 = sw + 4 >= NSLOT ? sw + 4 - NSLOT : sw + 4;                                                         \
  const x3_u32x4* xn = ldsv + s1 * SLOT + (wave * 2) * 128;
#define X3S_END()                                                                                                   \
  _Pragma("unroll") for (int mb = 0; mb < 2; ++mb) { xh[mb] = nh[mb]; xl[mb] = nl[mb]; }                            \
  sw = sw + 2 >= NSLOT ? sw + 2 - NSLOT : sw + 2;
  if (grp == 0) {
    for (int g = 0; g < nk; ++g) {
      X3S_VARS()
      X3S_S(0) X3S_LOAD(0) X3S_S(1) __builtin_amdgcn_s_barrier(); X3S_S(2) X3S_MMA(0) X3S_S(3) __builtin_amdgcn_s_barrier();
      X3S_S(4) X3S_LOAD(1) X3S_S(5) __builtin_amdgcn_s_barrier(); X3S_S(6) X3S_MMA(1) X3S_S(7) __builtin_amdgcn_s_barrier();
      X3S_S(8) X3S_LOAD(2) X3S_S(9) __builtin_amdgcn_s_barrier(); X3S_S(10) X3S_MMA(2) X3S_S(11) __builtin_amdgcn_s_barrier();
      X3S_S(12) X3S_LOAD(3) X3S_S(13) __builtin_amdgcn_s_barrier(); X3S_S(14) X3S_MMA(3)
      if (DBG != 1) wait_vm<GW>();
      __builtin_amdgcn_sched_barrier(0);
      X3S_S(15)
      __builtin_amdgcn_s_barrier();
      X3S_COMMIT()
      X3S_END()
    }
  } else {
    __builtin_amdgcn_s_barrier();  // one slot behind
    for (int g = 0; g < nk; ++g) {
      X3S_VARS()
      X3S_S(0) X3S_LOAD(0) X3S_S(1) __builtin_amdgcn_s_barrier(); X3S_S(2) X3S_MMA(0) X3S_S(3) __builtin_amdgcn_s_barrier();
      X3S_S(4) X3S_LOAD(1) X3S_S(5) __builtin_amdgcn_s_barrier(); X3S_S(6) X3S_MMA(1) X3S_S(7) __builtin_amdgcn_s_barrier();
      X3S_S(8) X3S_LOAD(2) X3S_S(9) __builtin_amdgcn_s_barrier(); X3S_S(10) X3S_MMA(2) X3S_S(11) __builtin_amdgcn_s_barrier();
      X3S_S(12) X3S_LOAD(3)
      if (DBG != 1) wait_vm<GW>();
      __builtin_amdgcn_sched_barrier(0);
      X3S_S(13)
      __builtin_amdgcn_s_barrier();
      X3S_S(14) X3S_MMA(3)
      X3S_S(15)
      if (g + 1 < nk) __builtin_amdgcn_s_barrier();
      X3S_COMMIT()
      X3S_END()
    }
  }
  if constexpr (STQ != 0) {
    if (p.trace && lane == 0) {
      long long* o = p.trace + 65536 + ((long)blockIdx.x * 8 + wave) * 32;
#pragma unroll
      for (int i = 0; i < 5; ++i) o[i] = sst[i];
      o[9] = nk; o[10] = gs; o[11] = STQ;
    }
  }
#undef X3S_S
#undef X3S_COMMIT
#undef X3S_LOAD
#undef X3S_MMA
#undef X3S_VARS
#undef X3S_END
  // ---- epilogue: LDS-transposed coalesced fp32 stores (epilogue_block: row base m0 + wm * (BM_/2) + mb * 16 with BM_ = 64, wm = wave;
  //      column base n0 + wn * 64 with wn = the 64-column group) ----
  wait_vm<0>();   // the surplus requests of the last two slabs: the ring becomes transpose patches
  __syncthreads();
  const float* bias = p.bias ? p.bias + (long)z2 * p.bias_z2 : nullptr;
  float* patch = (float*)lds + wave * (16 * 68);
  if constexpr (DBG == 3) {   // every accumulator stays live: a check of two of them lets hipcc delete the MFMAs of all the others
#pragma unroll
    for (int i = 0; i < NB; ++i) asm volatile("" ::"v"(acc[i][0]), "v"(acc[i][1]));
    return;
  }
#pragma unroll
  for (int g = 0; g < NBS; ++g) {
    const BiasRegs br = load_bias_regs<true>(p, bias, lane, g, n0);
    epilogue_block<2, 64, true>(p, acc[4 * g][0], acc[4 * g + 1][0], acc[4 * g + 2][0], acc[4 * g + 3][0], 0, patch, lane, wave, g, m0, n0, czoff, br);
    epilogue_block<2, 64, true>(p, acc[4 * g][1], acc[4 * g + 1][1], acc[4 * g + 2][1], acc[4 * g + 3][1], 1, patch, lane, wave, g, m0, n0, czoff, br);
  }
}

// fp32 (N, K) -> per row and 32-deep K slab [32 hi pieces | 32 lo pieces] (16-bit): one thread per 8 consecutive k
template <bool F16>
__global__ void split_pack_kernel(const float* __restrict__ w, long n_rows, int K, unsigned short* __restrict__ out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;   // index of an 8-element piece
  const long per_row = K / 8;
  if (i >= n_rows * per_row) return;
  const long n = i / per_row;
  const int k0 = (int)(i % per_row) * 8;
  const float* src = w + n * K + k0;
  unsigned short h[8], l[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float x = src[j];
    if constexpr (F16) {
      const _Float16 a = (_Float16)x, b = (_Float16)(x - (float)a);
      h[j] = __builtin_bit_cast(unsigned short, a);
      l[j] = __builtin_bit_cast(unsigned short, b);
    } else {
      const bf16_t a = (bf16_t)x, b = (bf16_t)(x - (float)a);
      h[j] = __builtin_bit_cast(unsigned short, a);
      l[j] = __builtin_bit_cast(unsigned short, b);
    }
  }
  unsigned short* dst = out + n * (2L * K) + (long)(k0 / 32) * 64 + (k0 % 32);
#pragma unroll
  for (int j = 0; j < 8; ++j) { dst[j] = h[j]; dst[32 + j] = l[j]; }
}

template <int BM>
int launch_pers(const GemmArgs& a, hipStream_t s) {
  const int tiles_m = (a.M + BM - 1) / BM, tiles_n = (a.N + 255) / 256;
  const int ntiles = tiles_m * tiles_n;
  int nblk = ntiles < 256 ? ((ntiles + 7) / 8) * 8 : 256;
  const size_t lds_bytes = 5 * 32768;
  if (int r_ = ensure_dyn_lds((const void*)gemm_pers_kernel<BM>, (int)lds_bytes)) return r_;
  const double flops = 2.0 * a.M * (double)a.N * a.K;
  const double bytes = ((double)a.M * a.K + (double)a.N * a.K) * 2 + (double)a.M * a.N * (a.out_f32 ? 4 : 2);
  prof_begin(s);
  hipLaunchKernelGGL((gemm_pers_kernel<BM>), dim3(nblk), dim3(512), lds_bytes, s, a, tiles_n, ntiles);
  prof_end(s, flops, bytes, 0);
  SVT_LAUNCH_CHECK();
  return 0;
}

template <int BM, bool GEN = false, int NBW = 4>
int launch_pp8(const GemmArgs& a, hipStream_t s) {
  const int tiles_m = (a.M + BM - 1) / BM, tiles_n = (a.N + 64 * NBW - 1) / (64 * NBW);
  dim3 grid(tiles_m * tiles_n, a.nz, 1);
  const size_t lds_bytes = 5 * 32768;
  if (int r_ = ensure_dyn_lds((const void*)gemm_pp8_kernel<BM, GEN, NBW>, (int)lds_bytes)) return r_;
  const double flops = 2.0 * a.M * (double)a.N * a.K * a.nz;
  const double bytes = ((double)a.M * a.K + (double)a.N * a.K) * 2 * a.nz + (double)a.M * a.N * a.nz * (a.out_f32 ? 4 : 2);
  prof_begin(s);
  hipLaunchKernelGGL((gemm_pp8_kernel<BM, GEN, NBW>), grid, dim3(512), lds_bytes, s, a);
  prof_end(s, flops, bytes, 0);
  SVT_LAUNCH_CHECK();
  return 0;
}

}  // namespace

// ---- registry of split weight matrices: fp32 device pointer -> packed (hi, lo) pieces (see gemm_x3_kernel) ----
namespace {
struct SplitW { void* packed; int N, K, kind; };
std::map<const void*, SplitW> g_split_w;
std::mutex g_split_mu;
}  // namespace

int split_weights_register(const void* w_f32, long n_rows, int K, int kind, hipStream_t s) {
  if (kind != 2 && kind != 3) return 0;
  if (K % 32 || n_rows < 1) return 0;   // K tails stay on the register-staged split kernel
  void* packed = nullptr;
  {
    // a re-upload of the same matrix packs into the buffer it already has (no free / allocate pair beside kernels: api.hip, upload_operand)
    std::lock_guard<std::mutex> lk(g_split_mu);
    auto it = g_split_w.find(w_f32);
    if (it != g_split_w.end() && it->second.N == (int)n_rows && it->second.K == K) packed = it->second.packed;
  }
  const bool reused = packed != nullptr;
  if (reused) SVT_HIP(hipDeviceSynchronize());   // products of a forward still in flight on another stream may be reading `packed`
  if (!reused)
    if (int r = dev_alloc(&packed, (size_t)n_rows * K * 4)) return r;
  const long pieces = n_rows * (K / 8);
  if (kind == 3) hipLaunchKernelGGL((split_pack_kernel<true>), dim3((unsigned)((pieces + 255) / 256)), dim3(256), 0, s, (const float*)w_f32, n_rows, K, (unsigned short*)packed);
  else hipLaunchKernelGGL((split_pack_kernel<false>), dim3((unsigned)((pieces + 255) / 256)), dim3(256), 0, s, (const float*)w_f32, n_rows, K, (unsigned short*)packed);
  SVT_LAUNCH_CHECK();
  std::lock_guard<std::mutex> lk(g_split_mu);
  auto it = g_split_w.find(w_f32);
  if (it != g_split_w.end() && !reused) dev_free(it->second.packed);
  g_split_w[w_f32] = SplitW{packed, (int)n_rows, K, kind};
  return 0;
}
void split_weights_forget(const void* w_f32) {
  std::lock_guard<std::mutex> lk(g_split_mu);
  auto it = g_split_w.find(w_f32);
  if (it == g_split_w.end()) return;
  dev_free(it->second.packed);
  g_split_w.erase(it);
}
// launches the LDS-DMA split kernel when `a` is a plain (un-batched) product against a registered weight matrix; returns
// 1 when the caller has to use the register-staged split kernel instead, 0 on success, < 0 on error
// svt_debug_set key 34 (tile_walk, common.h).  The automatic choice: panels when W does not fit an XCD's L2 beside the A stream and the
// problem has enough columns of tiles to form them.
int g_gemm_walk = -1;
int g_conv_kperm = 1;   // svt_debug_set key 35
int g_gemm_persist_wgs = 256;   // svt_debug_set key 37
int gemm_walk_pm(const GemmArgs& a, int bm) {
  if (g_gemm_walk >= 0) return g_gemm_walk;
  const int tiles_n = a.N / 256, tiles_m = (a.M + bm - 1) / bm;
  const double w_bytes = (double)a.N * a.K * 2.0;
  // measured (profiles/r06_gemm_tile_walk.txt, us at pm = 0 / 8): FFN-1 of the base model (W 4.5 MiB) 76.1 / 70.6, the large FFN-1 (8 MiB) 262 / 255,
  // the large QKV (6 MiB) 181 / 177, 8192^3 800 / 746; QKV of the base model (W 3.4 MiB: resident) 52.3 / 53.5 -- hence the 4 MiB line
  if (tiles_n < 8 || tiles_m < 16 || w_bytes <= 4.0 * 1024 * 1024) return 0;
  return 8;
}
int g_x3_pairs = 1;
int launch_gemm_x3(int kind, const GemmArgs& a, hipStream_t s) {
  if (a.a_pairs) {
    // pair-row operand: only gemm_x3q_kernel reads it (the callers in api.hip ask gemm_x3q_eligible before they choose the layout)
    if (!gemm_x3q_eligible(a) || a.ldw != a.K) { set_error("gemm: pair-row operand outside the contract of gemm_x3q_kernel"); return -1; }
    const void* packed = nullptr;
    {
      std::lock_guard<std::mutex> lk(g_split_mu);
      auto it = g_split_w.upper_bound(a.W);
      if (it != g_split_w.begin()) {
        --it;
        const size_t off = (const char*)a.W - (const char*)it->first;
        if (it->second.kind == kind && it->second.K == a.K && off % ((size_t)a.K * 4) == 0 && off / ((size_t)a.K * 4) + a.N <= (size_t)it->second.N)
          packed = (const char*)it->second.packed + off;
      }
    }
    if (!packed) { set_error("gemm: pair-row product against a weight matrix that was not registered as split"); return -1; }
    // tile height: rounds of 256 CUs x measured slab time of that height (the table of launch_gemm_dma)
    const int cands[3] = {256, 192, 128};
    const int slab_cost[3] = {167, 137, 105};
    long best_cost = -1;
    int best = 256;
    for (int i = 0; i < 3; ++i) {
      const long blocks = (long)((a.M + cands[i] - 1) / cands[i]) * (a.N / 256);
      const long cost = ((blocks + 255) / 256) * slab_cost[i];
      if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = cands[i]; }
    }
    if (g_gemm_force_bm == 256 || g_gemm_force_bm == 192 || g_gemm_force_bm == 128) best = g_gemm_force_bm;
    GemmArgs g = a;
    g.planes_f16 = kind == 3;
    const double flops = 2.0 * a.M * (double)a.N * a.K;
    const double bytes = ((double)a.M * a.K + (double)a.N * a.K + (double)a.M * a.N) * 4;
    prof_begin(s);
    if (g_gemm_p1x && g.K >= 96) { if (int r_ = launch_gemm_p1x(kind, g, packed, best, s)) return r_; }   // one wave per SIMD (gemm_p1x.hip; svt_debug_set key 30 = 1: A/B, same bits)
    else if (int r_ = launch_gemm_x3q(kind, g, packed, best, s)) return r_;
    prof_end(s, flops, bytes, 0);
    return 0;
  }
  // batched problems (nz > 1: the grouped positional conv, one z per group): the one-tile kernel with blockIdx.y = z; the registered
  // matrix holds the groups' rows one after the other
  const bool batched = a.nz > 1;
  if (a.gen || a.nz < 1 || a.K % 32 || a.N < 128 || a.M < 128 || !a.c_vec || a.ldw != a.K || a.alpha != 1.f || (a.a_rstride & 3) ||
      (a.a_bstride & 3) || ((uintptr_t)a.A & 15))
    return 1;
  if (!batched && (a.w_z1 || a.w_z2 || a.a_z1 || a.a_z2 || a.c_z1 || a.c_z2)) return 1;
  if (batched && (a.planes || a.resid || a.nz2 < 1 || a.nz % a.nz2 || (a.a_z1 & 3) || (a.a_z2 & 3) || (a.c_z1 & 3) || (a.c_z2 & 3) || (a.bias_z2 & 3) ||
                  a.w_z1 % a.K || a.w_z2 % a.K || a.nz > 65535))
    return 1;
  const void* packed = nullptr;
  {
    std::lock_guard<std::mutex> lk(g_split_mu);
    // the weight pointer may point INTO a registered matrix (row offset): find the matrix that contains it
    auto it = g_split_w.upper_bound(a.W);
    if (it == g_split_w.begin()) return 1;
    --it;
    const char* base = (const char*)it->first;
    const size_t off = (const char*)a.W - base;
    if (it->second.kind != kind || it->second.K != a.K || off % ((size_t)a.K * 4) != 0) return 1;
    const size_t row = off / ((size_t)a.K * 4);
    const size_t last_z_row = batched ? ((size_t)(a.nz / a.nz2 - 1) * a.w_z1 + (size_t)(a.nz2 - 1) * a.w_z2) / (size_t)a.K : 0;
    if (row + last_z_row + a.N > (size_t)it->second.N) return 1;
    packed = (const char*)it->second.packed + row * (size_t)a.K * 4;
  }
  {
    // the kernels address a tile's rows by 32-bit offsets from its first row, and the packed weights by 32-bit offsets from their base
    const unsigned long clips = 255 / (unsigned long)(a.a_rpb > 0 ? a.a_rpb : 1) + 1;
    const unsigned long bs = (unsigned long)(a.a_bstride > 0 ? a.a_bstride : 0), rs = (unsigned long)(a.a_rstride > 0 ? a.a_rstride : 0);
    if (a.a_bstride < 0 || a.a_rstride <= 0 || (clips * bs + 256ul * rs + (unsigned long)a.K) * 4 >= 0xF0000000ul ||
        (unsigned long)a.N * a.K * 4 >= 0xF0000000ul)
      return 1;
  }
  const size_t lds_bytes = 5 * 32768;
  GemmArgs g = a;
  g.out_f32 = 1;
  g.planes_f16 = kind == 3;
  if (g_gemm_dbg == 9 && a.resid) {   // diagnostics (tools/gemm_trace.py --x3-slots): the trace buffer travels in `resid`
    g.trace = (long long*)a.resid;
    g.resid = nullptr;
    g.stamp_ends = g_stamp_ends;
  }
  const double flops = 2.0 * a.M * (double)a.N * a.K * a.nz;
  const double bytes = ((double)a.M * a.K + (double)a.N * a.K + (double)a.M * a.N) * 4 * a.nz;
  // tile width: 192 columns when that fills the chip better (N = 768: 252 tiles against 189); g_gemm_variant 30 = the lockstep kernel (A/B)
  const long tm = (a.M + 255) / 256;
  const long t256 = tm * ((a.N + 255) / 256), t192 = tm * ((a.N + 191) / 192);
  auto rounds = [](long t) { return (t + 255) / 256; };
  const bool narrow = a.N % 192 == 0 && rounds(t192) * 3 < rounds(t256) * 4;
  prof_begin(s);
  // persistent form (gemm_x3p.hip: register epilogue, stores under the next tile's MFMAs) wherever it is eligible; 192-column tiles of
  // the one-tile kernel when they fill the chip better and the launch is a single round anyway (svt_debug_set key 3: 32 = never the
  // persistent form, 34 = always when eligible -- A/B)
  // Measured per shape (profiles/r03_gemm_x3_variants.txt): the persistent form is ahead on the launches with a heavy epilogue and
  // several tiles per CU (conv 1-4, FFN-1: GELU over fp32 outputs), the one-tile kernel on the plain projections (QKV, out-proj, FFN-2).
  const bool x3p_ok = !batched && gemm_x3p_eligible(g) && g_gemm_variant != 30 && g_gemm_variant != 31 && g_gemm_variant != 32 && g_gemm_variant != 33;
  if (x3p_ok && (g_gemm_variant == 34 || (a.act == ACT_GELU && t256 > 256) || t256 >= 1024)) {   // (large QKV, 1 500 tiles: 547 against 583 us)
    g.dbg = g_gemm_dbg == 9 ? 0 : g_gemm_dbg;
    if (int r_ = launch_gemm_x3p(kind, g, packed, s)) return r_;
#ifdef SVT_DIAG
  } else if (g.trace && kind == 3 && g.stamp_ends >= 1 && g.stamp_ends <= 4) {   // slot stamps of the one-tile kernel (diagnostics; make DIAG=1)
#define SVT_X3S_STAMP(NBS_, D_)                                                                                    \
  {                                                                                                                 \
    if (int r_ = ensure_dyn_lds((const void*)gemm_x3s_kernel<true, NBS_, D_>, (int)lds_bytes)) return r_;           \
    hipLaunchKernelGGL((gemm_x3s_kernel<true, NBS_, D_>), dim3((unsigned)(narrow ? t192 : t256)), dim3(512), lds_bytes, s, g, packed); \
  }
    if (narrow) {
      if (g.stamp_ends == 1) SVT_X3S_STAMP(3, 11) else if (g.stamp_ends == 2) SVT_X3S_STAMP(3, 12)
      else if (g.stamp_ends == 3) SVT_X3S_STAMP(3, 13) else SVT_X3S_STAMP(3, 14)
    } else {
      if (g.stamp_ends == 1) SVT_X3S_STAMP(4, 11) else if (g.stamp_ends == 2) SVT_X3S_STAMP(4, 12)
      else if (g.stamp_ends == 3) SVT_X3S_STAMP(4, 13) else SVT_X3S_STAMP(4, 14)
    }
#undef SVT_X3S_STAMP
  } else if ((g_gemm_variant == 31 || g_gemm_variant == 33) && kind == 3) {
    if (g_gemm_variant == 31) {
      if (int r_ = ensure_dyn_lds((const void*)gemm_x3s_kernel<true, 4, 1>, (int)lds_bytes)) return r_;
      hipLaunchKernelGGL((gemm_x3s_kernel<true, 4, 1>), dim3((unsigned)t256), dim3(512), lds_bytes, s, g, packed);
    } else {
      if (int r_ = ensure_dyn_lds((const void*)gemm_x3s_kernel<true, 4, 3>, (int)lds_bytes)) return r_;
      hipLaunchKernelGGL((gemm_x3s_kernel<true, 4, 3>), dim3((unsigned)t256), dim3(512), lds_bytes, s, g, packed);
    }
#endif
  } else if (narrow) {
    if (kind == 3) {
      if (int r_ = ensure_dyn_lds((const void*)gemm_x3s_kernel<true, 3>, (int)lds_bytes)) return r_;
      hipLaunchKernelGGL((gemm_x3s_kernel<true, 3>), dim3((unsigned)t192, (unsigned)a.nz), dim3(512), lds_bytes, s, g, packed);
    } else {
      if (int r_ = ensure_dyn_lds((const void*)gemm_x3s_kernel<false, 3>, (int)lds_bytes)) return r_;
      hipLaunchKernelGGL((gemm_x3s_kernel<false, 3>), dim3((unsigned)t192, (unsigned)a.nz), dim3(512), lds_bytes, s, g, packed);
    }
  } else {
    if (kind == 3) {
      if (int r_ = ensure_dyn_lds((const void*)gemm_x3s_kernel<true, 4>, (int)lds_bytes)) return r_;
      hipLaunchKernelGGL((gemm_x3s_kernel<true, 4>), dim3((unsigned)t256, (unsigned)a.nz), dim3(512), lds_bytes, s, g, packed);
    } else {
      if (int r_ = ensure_dyn_lds((const void*)gemm_x3s_kernel<false, 4>, (int)lds_bytes)) return r_;
      hipLaunchKernelGGL((gemm_x3s_kernel<false, 4>), dim3((unsigned)t256, (unsigned)a.nz), dim3(512), lds_bytes, s, g, packed);
    }
  }
  prof_end(s, flops, bytes, 0);
  SVT_LAUNCH_CHECK();
  return 0;
}

bool gemm_dma_eligible(const GemmArgs& a) { return a.K % 64 == 0 && a.N >= 128 && a.M >= 128 && a.c_vec && a.N % 8 == 0; }

// Dispatch.  Tile height BM minimises (rounds of 256 CUs) x BM; ties go to the larger tile (higher arithmetic
// intensity against the L2->LDS fill rate).  Problems with at least two full rounds of tiles run the persistent
// kernel (epilogue stores and the next tile's fills overlap the MFMAs); single-round problems run the
// one-tile-per-workgroup kernel, whose LDS-transposed epilogue stores whole 128-byte lines.
int g_gemm_dbg = 0;
int g_gemm_force_bm = 0;
int g_gemm_variant = 0;
int g_pps_half_barriers = 1;   // gemm_pps_kernel, four barriers per slab: 0 = never (A/B, slot stamps), 1 = always, 2 = GELU tiles with K <= 1024 only (svt_debug_set key 16)
int g_stamp_ends = 0;   // gemm_pps_kernel slot stamps (tools/gemm_trace.py --slots): 0 = slot starts, 1 = slot ends
int g_gemm_ring = 0;  // 0 = auto; 2 = force the one-tile-per-workgroup kernel, 4 = force the persistent kernel (diagnostics)
int launch_gemm_dma(const GemmArgs& a0, hipStream_t s) {
  GemmArgs a = a0;
  a.dbg = g_gemm_dbg;
  if (a.dbg == 9) { a.trace = (long long*)a.resid; a.resid = nullptr; if (g_gemm_variant) a.dbg = g_gemm_variant; }
  const int tiles_n = (a.N + 255) / 256;
  // Tile height: minimise (rounds of 256 CUs) x (time of one K slab at that height).  The slab times are measured
  // (tools/gemm_trace.py / gemm_bench.py --bm): 1.67 / 1.37 / 1.05 / 0.85 us for 256 / 192 / 128 / 64 rows -- a shorter tile
  // does proportionally less MFMA work but moves the same 32 KiB of W per slab through the CU, so it only pays when it
  // saves whole rounds.  Ties go to the larger tile.
  const int cands[4] = {256, 192, 128, 64};
  const int slab_cost[4] = {167, 137, 105, 85};
  long best_cost = -1;
  int best = 256;
  for (int i = 0; i < 4; ++i) {
    const int bm = cands[i];
    const long blocks = (long)((a.M + bm - 1) / bm) * tiles_n * a.nz;
    const long rounds = (blocks + 255) / 256;
    const long cost = rounds * slab_cost[i];
    if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = bm; }
  }
  if (g_gemm_force_bm) best = g_gemm_force_bm;
  if (a.gen) {  // generalised addressing: one-tile kernel only (it also carries the residual epilogue)
    // 128-channel outputs (second stage of the lip front-end) get a 128-column tile: a 256-wide tile would multiply
    // zeros for half of its MFMAs
    // (a 64-column tile, NBW = 1, is LDS-read bound -- 18 fragment reads per 16 MFMAs -- and measured slower than the
    // register-staged kernel: 64-channel layers stay there)
    if (a.N <= 128) return launch_pp8<256, true, 2>(a, s);
    if (best == 256) return launch_pp8<256, true>(a, s);
    if (best == 192) return launch_pp8<192, true>(a, s);
    return launch_pp8<128, true>(a, s);
  }
  const long ntiles = (long)((a.M + best - 1) / best) * tiles_n;
  // Persistent staggered kernel (gemm_pps.hip): bf16 outputs without residual, from 100 tiles up.  With several tiles per workgroup
  // the prologue, the epilogue stores and the next tile's fills overlap; on single-round launches its register epilogue costs
  // 1.3 us against 4 us for the LDS-transposed one of gemm_pp8_kernel, and since its slab loop lost the per-slab register swaps
  // (gemm_pps.hip) it is as fast per slab: FFN-2 (252 tiles, K = 3072) 68.3 -> 61.1 us, the 138-tile FFN-2 of a 35-utterance song
  // 52.7 -> 48.0, 4096^3 115 -> 109; large FFN-2 (500 tiles, K = 4096) equal.  Measured per shape:
  // profiles/r03_gemm_vendor_library_yardstick.txt.  Write-through (sc1) stores: the output leaves L2 while the kernel runs instead
  // of at the kernel boundary (FFN-1: 98 MB, 16 us).
  // svt_debug_set key 3: 50 / 70 force it with default / sc1 stores (53: without epilogue), 49 switches it off.
  if (g_gemm_variant >= 50 && g_gemm_variant < 80 && gemm_pps_eligible(a)) {
    GemmArgs b = a;
    b.dbg = g_gemm_variant % 10;
    b.stamp_ends = g_stamp_ends;
    return launch_gemm_pps(b, g_gemm_force_bm ? g_gemm_force_bm : (best < 128 ? 128 : best), s, (g_gemm_variant - 50) / 10);
  }
  if (g_gemm_variant != 49 && g_gemm_ring == 0 && best >= 128 && ntiles >= 100 && gemm_pps_eligible(a)) {
    // single-wave-per-SIMD kernel (gemm_p1w.hip, round 5): 5-11 % faster per launch wherever its un-overlapped epilogue is small beside the
    // tile -- everything except GELU launches with fewer than 16 K slabs (FFN-1 of the base model: 12 slabs, 72.8 us here against 81.9)
    if (g_gemm_p1w && a.K >= 192 && (!a.trace || g_gemm_p1w == 2) && (g_gemm_p1w == 2 || !(a.act == ACT_GELU && a.K < 1024))) {   // key 29 = 2: everywhere, traced launches included (tools/gemm_trace.py --p1w)
      if (a.W_kperm && g_conv_kperm && a.kperm_taps >= 2 && a.kperm_taps <= 3 && a.kperm_cin % 64 == 0 && a.K == a.kperm_taps * a.kperm_cin && a.ldw == a.K) {
        GemmArgs b = a;   // a kernel-3 convolution: tap-minor K order (GemmArgs::k_taps; the caller's second copy of W is stored that way)
        b.W = a.W_kperm; b.k_taps = a.kperm_taps; b.k_cin = a.kperm_cin;
        return launch_gemm_p1w(b, best, s);
      }
      return launch_gemm_p1w(a, best, s);
    }
    return launch_gemm_pps(a, best, s, 2);
  }
  const bool pers_ok = !a.resid && a.nz == 1 && a.K >= 128 && a.N % 256 == 0 && a.c_z1 == 0 && a.c_z2 == 0 &&
                       a.a_z1 == 0 && a.a_z2 == 0 && a.w_z1 == 0 && a.w_z2 == 0;
  int mode = g_gemm_ring;
  if (mode == 0) mode = (pers_ok && ntiles >= 512) ? 4 : 2;
  if (mode == 4 && pers_ok) {
    if (best == 256) return launch_pers<256>(a, s);
    if (best == 192) return launch_pers<192>(a, s);
    if (best == 128) return launch_pers<128>(a, s);
    return launch_pers<64>(a, s);
  }
  if (best == 256) return launch_pp8<256>(a, s);
  if (best == 192) return launch_pp8<192>(a, s);
  if (best == 128) return launch_pp8<128>(a, s);
  return launch_pp8<64>(a, s);  // mid-size problems (2-16 utterances): twice the workgroups of the 128-row tile
}

}  // namespace svt

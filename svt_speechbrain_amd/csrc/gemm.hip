// Dense contraction kernel for gfx950: C = act(alpha * A W^T + bias) + resid, A (M,K) and W (N,K)
// both K-contiguous.  It carries every dense product of the encoder:
// conv layers 1-6 as implicit GEMM over the channels-last activation (overlapping A rows), feature
// projection, the grouped positional conv (batched over (clip, group)), q/k/v/out projections, FFN,
// and (materialised-score attention path) QK^T and PV.
//
// Structure (wave64, 256 threads = 4 waves, one 64x64 output sub-tile per wave):
//   * K is consumed in 128-byte slabs per row (64 bf16 / 32 fp32).  Global -> register -> LDS staging:
//     one wave instruction loads 8 rows x 128 B (full lines), and writes 16-byte pieces into LDS in
//     *fragment order*, so that every MFMA operand read is one ds_read_b128 of 64 lanes x 16 B
//     contiguous (conflict-free by construction, no swizzle needed).
//   * LDS is double buffered; the next slab's global loads are issued before the MFMAs of the current
//     slab and written after them (one barrier per slab).
//   * The MFMA computes the transposed tile (A-operand = W rows, B-operand = activation rows), and the
//     W rows of each 64-row group are permuted at staging time so that a lane ends up holding 16
//     CONSECUTIVE output columns of one output row: the epilogue is bias/act/residual on registers
//     and 16-byte stores, no LDS round trip.
//   * fp32 operands use v_mfma_f32_16x16x4_f32 (bit-exact fp32 fma chain) with the same staging and
//     fragment layout: a fragment is 8 k-consecutive elements per lane for both types.
#include "common.h"
#include <type_traits>

namespace svt {
namespace {

template <typename T> struct OpTraits;
template <> struct OpTraits<float> {
  static constexpr int EPP = 4;   // elements per 16-byte piece
  static constexpr int BK = 32;   // elements per 128-byte slab row
};
template <> struct OpTraits<bf16_t> {
  static constexpr int EPP = 8;
  static constexpr int BK = 64;
};

__device__ __forceinline__ float gelu_erf(float x) { return gelu_fast(x); }

// ---- split-operand engine (precision "bf16x3" / "fp16x3") -------------------------------------------------------
// Operands stay fp32 in HBM (the fp32 parity pipeline is unchanged); on its way into LDS every fp32 value x is cut
// into two 16-bit pieces hi = round16(x), lo = round16(x - hi) and the product is accumulated in fp32 as
// Wh*Xh + Wh*Xl + Wl*Xh on the 16-bit matrix pipe (v_mfma_f32_16x16x32_{bf16,f16}): three MFMAs of 16 cycles instead of
// eight fp32 MFMAs of 32 cycles per 16x16x32 block (5.3x fewer matrix cycles), dropped term Wl*Xl ~ 2^-18 (bf16) /
// 2^-24 (f16) relative.  Measured against the reference goldens (tools/sim_split.py, base 5 s clip): max |dlogit|
// 8.3e-4 (bf16x3) / 9.3e-5 (fp16x3) vs 3.8e-1 with plain bf16 operands; fp16 pieces need |x| < 65504.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <int SPLIT> __device__ __forceinline__ void split4(const uint4& raw, uint2& hi, uint2& lo) {
  const f32x4 x = __builtin_bit_cast(f32x4, raw);
  if constexpr (SPLIT == 1) {
    bf16x4 h, l;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      h[j] = (bf16_t)x[j];
      l[j] = (bf16_t)(x[j] - (float)h[j]);
    }
    hi = __builtin_bit_cast(uint2, h);
    lo = __builtin_bit_cast(uint2, l);
  } else {
    typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
    f16x4 h, l;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      h[j] = (_Float16)x[j];
      l[j] = (_Float16)(x[j] - (float)h[j]);
    }
    hi = __builtin_bit_cast(uint2, h);
    lo = __builtin_bit_cast(uint2, l);
  }
}
template <int SPLIT> __device__ __forceinline__ f32x4 mfma16(const uint4& a, const uint4& b, const f32x4& c) {
  if constexpr (SPLIT == 2)
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(real_bf16x8, a), __builtin_bit_cast(real_bf16x8, b), c, 0, 0, 0);
}

__device__ __forceinline__ float apply_act(float v, int act) {
  if (act == ACT_GELU) return gelu_erf(v);
  if (act == ACT_RELU) return v > 0.f ? v : 0.f;
  return v;
}

// SPLIT (T = float only): 0 = exact fp32 MFMA, 1 = bf16x3, 2 = fp16x3 split-operand products (see above)
// NSET = register sets of the global -> LDS staging, i.e. K slabs in flight per workgroup (the exact-fp32 and bf16 forms
// multiply a slab for longer than a load takes: one set; the split form's 48 MFMAs per slab last ~0.35 us: two).
// MINW = waves per SIMD the register allocation must allow.
template <typename T, int BM, int BN, bool GEN = false, int SPLIT = 0, int NSET = 1, int MINW = 1>
__global__ __launch_bounds__(256, MINW) void gemm_kernel(GemmArgs p) {
  static_assert(SPLIT == 0 || sizeof(T) == 4, "split-operand products read fp32 operands");
  constexpr int EPP = OpTraits<T>::EPP;
  constexpr int BK = OpTraits<T>::BK;
  constexpr int KS = BK / 32;         // 32-element MFMA k-steps per slab (2 bf16, 1 fp32)
  constexpr int PPC = 8 / EPP;        // 16-byte pieces per 8-element fragment (1 bf16, 2 fp32)
  constexpr int WAVES_N = BN / 64;
  constexpr int XP = BM / 32;         // 16-byte pieces of the activation slab per thread
  constexpr int WP = BN / 32;
  constexpr int X_PIECES = BM * 8;    // pieces per activation slab
  constexpr int STAGE_PIECES = (BM + BN) * 8;

  extern __shared__ __attribute__((aligned(16))) uint4 lds[];  // 2 stages x STAGE_PIECES

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WAVES_N;
  const int wn = wave % WAVES_N;

  // block -> tile.  Blocks b, b+8, b+16 ... run on one XCD (round-robin dispatch) and share its 4 MiB L2: each XCD gets
  // a contiguous range of the logical tile list, and the list walks bands of `raster_gm` M-tiles x all N-tiles (M
  // fastest inside a band), so the ~64 workgroups an XCD has in flight cover ~8 A panels x ~8 W panels and every panel
  // is fetched into that L2 once per band instead of once per tile.  (M-fastest over the whole problem streamed A from
  // the Infinity Cache once per N-tile: the fp32-operand kernels ran at its ~6 TB/s, 190 TFLOP/s in split mode.)
  const int tiles_m = (p.M + BM - 1) / BM;
  int tile_m, tile_n;
  {
    const int nblk = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, q = nblk >> 3, r = nblk & 7;
    const int l = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const int tiles_n = nblk / tiles_m;
    const int per_band = p.raster_gm * tiles_n;
    const int band = l / per_band, rem = l - band * per_band;
    const int left = tiles_m - band * p.raster_gm;
    const int gm = left < p.raster_gm ? left : p.raster_gm;
    tile_m = band * p.raster_gm + rem % gm;
    tile_n = rem / gm;
  }
  const int z = blockIdx.y;
  const int z1 = z / p.nz2, z2 = z % p.nz2;
  const int m0 = tile_m * BM;
  const int n0 = tile_n * BN;

  const T* A = (const T*)p.A + (z1 * p.a_z1 + z2 * p.a_z2);
  const T* W = (const T*)p.W + (z1 * p.w_z1 + z2 * p.w_z2);

  // per-thread source rows (clamped: out-of-range rows are computed on valid memory and never stored)
  const int prow = lane & 7;   // row inside an 8-row group
  const int pc = lane >> 3;    // 16-byte piece inside the 128-byte slab row
  const T* xsrc[XP];
  int xdst[XP];
#pragma unroll
  for (int i = 0; i < XP; ++i) {
    int r = (i * 4 + wave) * 8 + prow;  // row in tile
    int m = m0 + r;
    if (m > p.M - 1) m = p.M - 1;
    long off;
    if constexpr (GEN) off = (long)m * p.a_rstride + (long)(m / p.a_d1) * p.a_e1 + (long)(m / p.a_d2) * p.a_e2;
    else off = (long)(m / p.a_rpb) * p.a_bstride + (long)(m % p.a_rpb) * p.a_rstride;
    xsrc[i] = A + off + pc * EPP;
    const int e0 = pc * EPP;
    const int ks = e0 / 32, cq = (e0 % 32) / 8, h = (e0 % 8) / EPP;
    if constexpr (SPLIT) xdst[i] = (((r >> 4) * 2) * 64 + cq * 16 + (r & 15)) * 2 + h;  // uint2 index of the hi piece; lo plane = +128
    else xdst[i] = (((r >> 4) * KS + ks) * PPC + h) * 64 + cq * 16 + (r & 15);
  }
  const T* wsrc[WP];
  int wdst[WP];
#pragma unroll
  for (int i = 0; i < WP; ++i) {
    int r = (i * 4 + wave) * 8 + prow;
    int n = n0 + r;
    if (n > p.N - 1) n = p.N - 1;
    wsrc[i] = W + (long)n * p.ldw + pc * EPP;
    const int e0 = pc * EPP;
    const int ks = e0 / 32, cq = (e0 % 32) / 8, h = (e0 % 8) / EPP;
    // row permutation inside each 64-row group: tile row (q*16 + nb*4 + rr) -> MFMA block nb, row 4q+rr
    const int g = r >> 6, q = (r & 63) >> 4, nb = (r & 15) >> 2, rr = r & 3;
    if constexpr (SPLIT) wdst[i] = (X_PIECES + ((g * 4 + nb) * 2) * 64 + cq * 16 + (4 * q + rr)) * 2 + h;
    else wdst[i] = X_PIECES + ((((g * 4 + nb) * KS + ks) * PPC + h) * 64 + cq * 16 + (4 * q + rr));
  }

  f32x4 acc[4][4];  // [nb][mb]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = (p.K + BK - 1) / BK;
  uint4 xr[NSET][XP], wr[NSET][WP];

  auto stage_load = [&](auto set_c, int kt) {
    constexpr int S = decltype(set_c)::value;
    // The loads are unconditional: a "load or zero" select per piece makes hipcc branch around every load and drain
    // vmcnt(0) at the top of each iteration (nothing stays in flight across the MFMAs).  A piece past the end of K
    // (last slab only) re-reads the FIRST piece of its row -- valid memory whatever K is (K < BK included: found with the
    // page-guarded allocator, tests/test_gpu_guard.py) -- and is zeroed when the slab is written to LDS.
    int kbase = kt * BK;
    long ka = kbase;
    if constexpr (GEN) {
      if (p.kseg) ka = (long)(kbase / p.kseg) * p.kseg_stride + (kbase % p.kseg);
    }
    if ((kbase + pc * EPP) >= p.K) { kbase = -pc * EPP; ka = kbase; }   // xsrc / wsrc already point at piece pc of the row
#pragma unroll
    for (int i = 0; i < XP; ++i) xr[S][i] = *(const uint4*)(xsrc[i] + ka);
#pragma unroll
    for (int i = 0; i < WP; ++i) wr[S][i] = *(const uint4*)(wsrc[i] + kbase);
  };
  const bool k_tail = (p.K % BK) != 0;
  auto stage_write = [&](auto set_c, int buf, int kt, bool maybe_last = true) {
    constexpr int S = decltype(set_c)::value;
    uint4* base = lds + buf * STAGE_PIECES;
    if (maybe_last && k_tail && kt == nk - 1 && (kt * BK + pc * EPP) >= p.K) {
#pragma unroll
      for (int i = 0; i < XP; ++i) xr[S][i] = uint4{0, 0, 0, 0};
#pragma unroll
      for (int i = 0; i < WP; ++i) wr[S][i] = uint4{0, 0, 0, 0};
    }
    if constexpr (SPLIT) {
      uint2* b2 = (uint2*)base;
#pragma unroll
      for (int i = 0; i < XP; ++i) {
        uint2 hi, lo;
        split4<SPLIT>(xr[S][i], hi, lo);
        b2[xdst[i]] = hi;
        b2[xdst[i] + 128] = lo;
      }
#pragma unroll
      for (int i = 0; i < WP; ++i) {
        uint2 hi, lo;
        split4<SPLIT>(wr[S][i], hi, lo);
        b2[wdst[i]] = hi;
        b2[wdst[i] + 128] = lo;
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < XP; ++i) base[xdst[i]] = xr[S][i];
#pragma unroll
    for (int i = 0; i < WP; ++i) base[wdst[i]] = wr[S][i];
  };
  // slab kt lives in register set kt % NSET until it is written to LDS buffer kt & 1 (during iteration kt - 1); iteration
  // kt then refills that set with slab kt + NSET, multiplies buffer kt & 1 and writes slab kt + 1
  // STEADY iterations (kt + NSET < nk guaranteed by the caller) carry no conditionals around the loads and the LDS
  // writes: with them hipcc cannot prove at the loop header which loads were waited for and drains vmcnt(0) before it
  // reuses a register of the set, i.e. before every refill.
  auto iteration = [&](auto u_c, auto steady_c, int kt) {
    constexpr int U = decltype(u_c)::value;
    constexpr bool STEADY = decltype(steady_c)::value;
    const int buf = kt & 1;
    if (STEADY || kt + NSET < nk) stage_load(std::integral_constant<int, U>{}, kt + NSET);
    const uint4* xb = lds + buf * STAGE_PIECES;
    const uint4* wb = xb + X_PIECES;
    if constexpr (SPLIT) {
      // one 32-deep k-step per slab: fragments of both planes, then three rounds of 16 independent MFMAs (small terms first)
      uint4 xh[4], xl[4], wh[4], wl[4];
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        xh[b] = xb[((wm * 4 + b) * 2) * 64 + lane];
        xl[b] = xb[((wm * 4 + b) * 2 + 1) * 64 + lane];
        wh[b] = wb[((wn * 4 + b) * 2) * 64 + lane];
        wl[b] = wb[((wn * 4 + b) * 2 + 1) * 64 + lane];
      }
#pragma unroll
      for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) acc[nb][mb] = mfma16<SPLIT>(wl[nb], xh[mb], acc[nb][mb]);
#pragma unroll
      for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) acc[nb][mb] = mfma16<SPLIT>(wh[nb], xl[mb], acc[nb][mb]);
#pragma unroll
      for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) acc[nb][mb] = mfma16<SPLIT>(wh[nb], xh[mb], acc[nb][mb]);
    } else {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      uint4 xf[4][PPC], wf[4][PPC];
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int h = 0; h < PPC; ++h) {
          xf[b][h] = xb[(((wm * 4 + b) * KS + ks) * PPC + h) * 64 + lane];
          wf[b][h] = wb[(((wn * 4 + b) * KS + ks) * PPC + h) * 64 + lane];
        }
#pragma unroll
      for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
          if constexpr (sizeof(T) == 2) {
            acc[nb][mb] = SVT_MFMA_16x16x32(__builtin_bit_cast(bf16x8, wf[nb][0]), __builtin_bit_cast(bf16x8, xf[mb][0]), acc[nb][mb]);
          } else {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              const f32x4 wv = __builtin_bit_cast(f32x4, wf[nb][h]);
              const f32x4 xv = __builtin_bit_cast(f32x4, xf[mb][h]);
#pragma unroll
              for (int j = 0; j < 4; ++j)
                acc[nb][mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[j], xv[j], acc[nb][mb], 0, 0, 0);
            }
          }
        }
    }
    }
    if (STEADY || kt + 1 < nk) stage_write(std::integral_constant<int, (U + 1) % NSET>{}, buf ^ 1, kt + 1, !STEADY);
    __syncthreads();
  };

  stage_load(std::integral_constant<int, 0>{}, 0);
  if constexpr (NSET > 1) { if (1 < nk) stage_load(std::integral_constant<int, 1>{}, 1); }
  if constexpr (NSET > 2) { if (2 < nk) stage_load(std::integral_constant<int, 2>{}, 2); }
  stage_write(std::integral_constant<int, 0>{}, 0, 0);
  __syncthreads();
  int kt0 = 0;
  if constexpr (NSET > 1) {
    for (; kt0 + 2 * NSET <= nk; kt0 += NSET) {
      iteration(std::integral_constant<int, 0>{}, std::true_type{}, kt0);
      iteration(std::integral_constant<int, 1>{}, std::true_type{}, kt0 + 1);
      if constexpr (NSET > 2) iteration(std::integral_constant<int, 2>{}, std::true_type{}, kt0 + 2);
    }
  }
  for (; kt0 < nk; kt0 += NSET) {
    iteration(std::integral_constant<int, 0>{}, std::false_type{}, kt0);
    if constexpr (NSET > 1) { if (kt0 + 1 < nk) iteration(std::integral_constant<int, 1>{}, std::false_type{}, kt0 + 1); }
    if constexpr (NSET > 2) { if (kt0 + 2 < nk) iteration(std::integral_constant<int, 2>{}, std::false_type{}, kt0 + 2); }
  }

  // ---- epilogue: lane holds C[m][nbase .. nbase+15] for each of its 4 row blocks ----
  const long coff = z1 * p.c_z1 + z2 * p.c_z2;
  const float* bias = p.bias ? p.bias + z2 * p.bias_z2 : nullptr;
  const int nbase = n0 + wn * 64 + (lane >> 4) * 16;
  if (nbase >= p.N) return;
  const bool full = (nbase + 16 <= p.N) && p.c_vec;
  float bv[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) bv[j] = 0.f;
  if (bias) {
    if (full) {
#pragma unroll
      for (int j4 = 0; j4 < 4; ++j4) {
        const float4 b4 = *(const float4*)(bias + nbase + j4 * 4);
        bv[j4 * 4 + 0] = b4.x; bv[j4 * 4 + 1] = b4.y; bv[j4 * 4 + 2] = b4.z; bv[j4 * 4 + 3] = b4.w;
      }
    } else {
#pragma unroll
      for (int j = 0; j < 16; ++j) if (nbase + j < p.N) bv[j] = bias[nbase + j];
    }
  }
  if constexpr (GEN) {
    float sv[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) sv[j] = (p.slope && nbase + j < p.N) ? p.slope[nbase + j] : 0.f;
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
      const int m = m0 + wm * 64 + mb * 16 + (lane & 15);
      if (m >= p.M) continue;
      const long idx = coff + p.c_base + (long)m * p.ldc + (long)(m / p.c_d1) * p.c_e1 + (long)(m / p.c_d2) * p.c_e2 + nbase +
                       (p.c_nsplit && nbase >= p.c_nsplit ? p.c_nstride - p.c_nsplit : 0);
      float v[16], rs[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) rs[j] = 0.f;
      if (p.resid) {
        if (full) {
          if (p.resid_op_type && sizeof(T) == 2) {
#pragma unroll
            for (int j8 = 0; j8 < 2; ++j8) {
              const bf16x8 r8 = *(const bf16x8*)((const bf16_t*)p.resid + idx + j8 * 8);
#pragma unroll
              for (int j = 0; j < 8; ++j) rs[j8 * 8 + j] = (float)r8[j];
            }
          } else {
#pragma unroll
            for (int j4 = 0; j4 < 4; ++j4) {
              const float4 r4 = *(const float4*)(p.resid + idx + j4 * 4);
              rs[j4 * 4 + 0] = r4.x; rs[j4 * 4 + 1] = r4.y; rs[j4 * 4 + 2] = r4.z; rs[j4 * 4 + 3] = r4.w;
            }
          }
        } else {
#pragma unroll
          for (int j = 0; j < 16; ++j)
            if (nbase + j < p.N)
              rs[j] = (p.resid_op_type && sizeof(T) == 2) ? (float)((const bf16_t*)p.resid)[idx + j] : p.resid[idx + j];
        }
      }
#pragma unroll
      for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int j = nb * 4 + r;
          float x = acc[nb][mb][r] * p.alpha + bv[j];
          if (p.resid_first) x += rs[j];
          if (p.act == ACT_PRELU) x = x > 0.f ? x : x * sv[j];
          else x = apply_act(x, p.act);
          if (!p.resid_first) x += rs[j];
          v[j] = x;
        }
      if (full) {
        if (p.out_f32 || sizeof(T) == 4) {
          float* c = (float*)p.C + idx;
#pragma unroll
          for (int j4 = 0; j4 < 4; ++j4)
            *(float4*)(c + j4 * 4) = float4{v[j4 * 4], v[j4 * 4 + 1], v[j4 * 4 + 2], v[j4 * 4 + 3]};
        } else {
          bf16_t* c = (bf16_t*)p.C + idx;
#pragma unroll
          for (int j8 = 0; j8 < 2; ++j8) {
            bf16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (bf16_t)v[j8 * 8 + j];
            *(bf16x8*)(c + j8 * 8) = o;
          }
        }
      } else {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          if (nbase + j >= p.N) continue;
          if (p.out_f32 || sizeof(T) == 4) ((float*)p.C)[idx + j] = v[j];
          else ((bf16_t*)p.C)[idx + j] = (bf16_t)v[j];
        }
      }
    }
    return;
  }
#pragma unroll
  for (int mb = 0; mb < 4; ++mb) {
    const int m = m0 + wm * 64 + mb * 16 + (lane & 15);
    if (m >= p.M) continue;
    float v[16];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
      for (int r = 0; r < 4; ++r) v[nb * 4 + r] = apply_act(acc[nb][mb][r] * p.alpha + bv[nb * 4 + r], p.act);
    const long idx = coff + (long)m * p.ldc + nbase;
    if (full) {
      if (p.resid) {
#pragma unroll
        for (int j4 = 0; j4 < 4; ++j4) {
          const float4 r4 = *(const float4*)(p.resid + idx + j4 * 4);
          v[j4 * 4 + 0] += r4.x; v[j4 * 4 + 1] += r4.y; v[j4 * 4 + 2] += r4.z; v[j4 * 4 + 3] += r4.w;
        }
      }
      if (p.out_f32 || sizeof(T) == 4) {
        float* c = (float*)p.C + idx;
#pragma unroll
        for (int j4 = 0; j4 < 4; ++j4)
          *(float4*)(c + j4 * 4) = float4{v[j4 * 4], v[j4 * 4 + 1], v[j4 * 4 + 2], v[j4 * 4 + 3]};
      } else {
        bf16_t* c = (bf16_t*)p.C + idx;
#pragma unroll
        for (int j8 = 0; j8 < 2; ++j8) {
          bf16x8 o;
#pragma unroll
          for (int j = 0; j < 8; ++j) o[j] = (bf16_t)v[j8 * 8 + j];
          *(bf16x8*)(c + j8 * 8) = o;
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        if (nbase + j >= p.N) continue;
        float o = v[j];
        if (p.resid) o += p.resid[idx + j];
        if (p.out_f32 || sizeof(T) == 4) ((float*)p.C)[idx + j] = o;
        else ((bf16_t*)p.C)[idx + j] = (bf16_t)o;
      }
    }
  }
}

template <typename T, int BM, int BN, bool GEN = false, int SPLIT = 0, int NSET = 1, int MINW = 1>
int launch_one(const GemmArgs& a, hipStream_t s) {
  const int tiles_m = (a.M + BM - 1) / BM, tiles_n = (a.N + BN - 1) / BN;
  dim3 grid(tiles_m * tiles_n, a.nz, 1);
  GemmArgs ar = a;
  ar.raster_gm = tiles_n >= 8 ? 8 : 64 / tiles_n;
  const size_t lds_bytes = 2 * (size_t)(BM + BN) * 128;
  if (int r_ = ensure_dyn_lds((const void*)gemm_kernel<T, BM, BN, GEN, SPLIT, NSET, MINW>, (int)lds_bytes)) return r_;
  const double flops = 2.0 * a.M * (double)a.N * a.K * a.nz;
  const double bytes = ((double)a.M * a.K + (double)a.N * a.K) * sizeof(T) * a.nz +
                       (double)a.M * a.N * a.nz * ((a.out_f32 || sizeof(T) == 4) ? 4 : 2);
  prof_begin(s);
  hipLaunchKernelGGL((gemm_kernel<T, BM, BN, GEN, SPLIT, NSET, MINW>), grid, dim3(256), lds_bytes, s, ar);
  prof_end(s, flops, bytes, SPLIT ? 0 : 1);
  SVT_LAUNCH_CHECK();
  return 0;
}

}  // namespace

int g_gemm_skinny = 1;
int g_gemm_x3 = 1;
// prec: 0 = fp32 operands, exact fp32 MFMA; 1 = bf16 operands; 2 / 3 = fp32 operands in memory, bf16x3 / fp16x3
// split-operand products (every other argument as for prec 0)
int launch_gemm(int prec_in, const GemmArgs& a, hipStream_t s) {
  if (a.M <= 0 || a.N <= 0 || a.K <= 0) { set_error("gemm: empty problem"); return -1; }
  // pair rows are understood by the split-operand LDS-DMA kernels only: every other kernel would read them as fp32 words
  if ((a.a_pairs || a.c_pairs) && (prec_in < 2 || !g_gemm_x3)) {
    set_error("gemm: pair-row operands / outputs need a split-operand precision and the LDS-DMA split kernels (svt_debug_set key 11 = 1)");
    return -1;
  }
  if (a.planes) {
    // C as (hi, lo) planes (split-operand modes only): written by the LDS-DMA split kernel's epilogue; any other kernel writes the
    // fp32 C and the planes are cut from it afterwards
    if (prec_in < 2 || a.nz != 1 || a.gen) { set_error("gemm: (hi, lo) plane output is served for plain split-operand products"); return -1; }
    if (g_gemm_x3) {
      GemmArgs gp = a;
      gp.c_vec = !(a.ldc & 3) && !((uintptr_t)a.C & 15) && !((uintptr_t)a.resid & 15) && !((uintptr_t)a.bias & 15) && !(a.plane_stride & 3) &&
                 !((uintptr_t)a.planes & 7);
      const int r = launch_gemm_x3(prec_in, gp, s);
      if (r <= 0) return r;
    }
    GemmArgs g2 = a;
    g2.planes = nullptr;
    if (int r = launch_gemm(prec_in, g2, s)) return r;
    return launch_split_planes(prec_in, (const float*)a.C, a.ldc, a.M, a.N, a.planes, a.ldc, a.plane_stride, s);
  }
  const int split = prec_in >= 2 ? prec_in - 1 : 0;
  const int prec = prec_in >= 2 ? 0 : prec_in;
  const int epp = prec ? 8 : 4;
  if (a.K % epp != 0) { set_error("gemm: K must be a multiple of the 16-byte piece"); return -1; }
  auto mult = [](long v, long m) { return v % m == 0; };
  if (!mult(a.a_rstride, epp) || !mult(a.a_bstride, epp) || !mult(a.a_z1, epp) || !mult(a.a_z2, epp) || !mult(a.ldw, epp) ||
      !mult(a.w_z1, epp) || !mult(a.w_z2, epp) || ((uintptr_t)a.A & 15) || ((uintptr_t)a.W & 15)) {
    set_error("gemm: operand rows must be 16-byte aligned");
    return -1;
  }
  GemmArgs g = a;
  const int cel = (a.out_f32 || !prec) ? 4 : 8;  // elements per 16 bytes of C
  g.c_vec = mult(a.ldc, cel) && mult(a.c_z1, cel) && mult(a.c_z2, cel) && !((uintptr_t)a.C & 15) &&
            mult(a.ldc, 4) && mult(a.c_z1, 4) && mult(a.c_z2, 4) && !((uintptr_t)a.resid & 15) &&
            mult(a.bias_z2, 4) && !((uintptr_t)a.bias & 15);
  const bool narrow = a.N <= 64;
  if (a.gen) {
    const int kel = prec ? 64 : 32;
    if (a.kseg && (a.kseg % kel || a.K % a.kseg)) { set_error("gemm: kseg must divide K and be a multiple of the K slab"); return -1; }
    if (!mult(a.a_e1, epp) || !mult(a.a_e2, epp) || !mult(a.kseg_stride, epp)) { set_error("gemm: generalised A strides must be 16-byte aligned"); return -1; }
    g.c_vec = g.c_vec && mult(a.c_e1, cel) && mult(a.c_e2, cel) && mult(a.c_base, cel) &&
              (!a.resid || !a.resid_op_type || !((uintptr_t)a.resid & 15));
    // N >= 128 with bf16 output: the LDS-DMA pipeline (needs 64-element K slabs inside every run and nz == 1)
    if (prec && a.K % 64 == 0 && a.N >= 128 && a.N % 8 == 0 && a.M >= 128 && g.c_vec && !a.out_f32 && a.nz == 1 &&
        (!a.resid || a.resid_op_type) && a.alpha == 1.f && (a.kseg == 0 || a.kseg % 64 == 0))
      return launch_gemm_dma(g, s);
    if (prec) return narrow ? launch_one<bf16_t, 256, 64, true>(g, s) : launch_one<bf16_t, 128, 128, true>(g, s);
    if (split == 1) return narrow ? launch_one<float, 256, 64, true, 1, 1>(g, s) : launch_one<float, 128, 128, true, 1, 2, 2>(g, s);
    if (split == 2) return narrow ? launch_one<float, 256, 64, true, 2, 1>(g, s) : launch_one<float, 128, 128, true, 2, 2, 2>(g, s);
    return narrow ? launch_one<float, 256, 64, true>(g, s) : launch_one<float, 128, 128, true>(g, s);
  }
  if (prec && g_gemm_skinny && gemm_skinny_eligible(g)) return launch_gemm_skinny(g, s);
  if (prec && gemm_dma_eligible(g)) return launch_gemm_dma(g, s);
  if (prec) return narrow ? launch_one<bf16_t, 256, 64>(g, s) : launch_one<bf16_t, 128, 128>(g, s);
  if (split && g_gemm_x3) {
    const int r = launch_gemm_x3(prec_in, g, s);
    if (r <= 0) return r;
  }
  if (a.a_pairs || a.c_pairs) { set_error("gemm: this geometry is outside the pair-row kernels' contract"); return -1; }
  // split engine: two slabs in flight per workgroup, capped at 256 registers so that two workgroups share a CU (measured
  // on the encoder's shapes, tools/gemm_bench.py --prec 2: one set 185-212, two sets 186-212, three sets (one workgroup per
  // CU) 150-192 TFLOP/s: the kernel is bound by issue / barrier stalls of its four-wave lockstep, not by the loads)
  if (split == 1) return narrow ? launch_one<float, 256, 64, false, 1, 1>(g, s) : launch_one<float, 128, 128, false, 1, 2, 2>(g, s);
  if (split == 2) return narrow ? launch_one<float, 256, 64, false, 2, 1>(g, s) : launch_one<float, 128, 128, false, 2, 2, 2>(g, s);
  return narrow ? launch_one<float, 256, 64>(g, s) : launch_one<float, 128, 128>(g, s);
}

}  // namespace svt

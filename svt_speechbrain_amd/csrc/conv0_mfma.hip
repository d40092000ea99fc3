// Conv layer 0 of the feature extractor (Cin = 1, 10 taps, stride 5, 512 channels) on the MATRIX pipe, for the 16-bit throughput modes.
//
// The vector-ALU kernels (kernels.hip: conv0_group_apply_kernel / conv0_layer_kernel) spend ~10 packed FMAs per output on the taps in
// front of the normalisation and the GELU and are bound by vector issue (C2: 256 us for a 1.05 GB write, C3: 1.01 ms for 2.1 GB, 40
// vector operations per output in the layer-norm form).  Here the taps are ONE v_mfma_f32_16x16x32 per 16 frames x 16 channels:
//     K = 32 = [ x_hi(10) | x_lo(10) | x_hi(10) | 1 | 1 ]  against  [ w_hi(10) | w_hi(10) | w_lo(10) | b_hi | b_lo ]
// i.e. x w + b with BOTH operands carried as 16-bit (hi, lo) pairs (x_hi w_hi + x_lo w_hi + x_hi w_lo: 2^-16 relative with bf16
// pieces, 2^-22 with IEEE-half ones -- the raw samples of the GroupNorm form are scaled per wave by a power of two first, so that this
// holds for quiet and for loud audio alike -- far below the 2^-9 / 2^-12 rounding of the stored result), the matrix pipe is busy for 3 % of
// the kernel, and what is left per output is the normalisation, the polynomial GELU and the conversion.
//   * the weight side is a 32 KiB table per clip (GroupNorm form: the 11 coefficients per (clip, channel) of conv0_group_coef_kernel,
//     which already fold the whole-batch waveform norm and the GroupNorm statistics) or one table for all clips (LayerNorm form: the
//     conv weights and bias), laid out in LDS as [block (q, nb)][lane][8 pieces] so that a B fragment is one linear ds_read_b128;
//     block (q, nb), column j holds channel 128 q + 8 j + nb, so a lane ends with 8 CONSECUTIVE channels of 4 frames per group q
//     (the register epilogue of gemm_pps_kernel): one 16-byte store per frame and group, 256 contiguous bytes per 16 lanes;
//   * a wave owns 64 consecutive frames of a clip (four 16-frame chunks), its 330 waveform samples staged once in a wave-private LDS
//     strip; the A fragment of a chunk is built in registers (10 samples per lane -> (hi, lo) pieces -> the k-range of the lane);
//   * GroupNorm form: 8 MFMAs per 128-channel group, GELU, store (32 accumulator registers live);
//     LayerNorm form: all 32 MFMAs of the chunk, row statistics from the accumulators (32 values per lane and frame + a 16-lane
//     exchange: four steps instead of a 64-lane sum per frame), then normalise + GELU + store per group.
// Contract: k = 10, stride <= 5, C = 512, 16-bit output.  Parity modes (fp32 storage, pair rows) keep the vector kernels.
#include "common.h"

namespace svt {
namespace {

constexpr int K0 = 10, C0 = 512, TBL16 = 2048;   // uint4 per table (32 KiB)

// PK = 0: the build's 16-bit operand type, 16-bit rows out, polynomial GELU (throughput modes);
// PK = 2 / 3 (split modes, fp32-grade): bf16 / IEEE-half pieces, output written as PAIR ROWS ([32 hi | 32 lo] per 32 channels, the
// operand layout of gemm_x3q_kernel), exact-erf GELU -- the taps then carry the same 2^-17 / 2^-22 relative error as every other
// product of those modes.
template <int PK> struct C0T { typedef bf16_t piece; };
template <> struct C0T<2> { typedef __bf16 piece; };
template <> struct C0T<3> { typedef _Float16 piece; };
template <int PK> __device__ __forceinline__ f32x4 c0_mma(const typename C0T<PK>::piece __attribute__((ext_vector_type(8))) & a, const uint4& b, const f32x4& c) {
  typedef typename C0T<PK>::piece P;
  typedef P P8 __attribute__((ext_vector_type(8)));
  if constexpr (PK == 3) return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, __builtin_bit_cast(P8, b), c, 0, 0, 0);
  else if constexpr (PK == 2) return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, __builtin_bit_cast(P8, b), c, 0, 0, 0);
  else return SVT_MFMA_16x16x32(a, __builtin_bit_cast(P8, b), c);
}
// 8 consecutive channels of one frame: 16-bit row (PK = 0) or pair row (PK = 2 / 3: e = element index of the first channel)
template <int PK> __device__ __forceinline__ void c0_store8(void* out, int64_t e, f32x2_t (&p)[4]) {
  typedef typename C0T<PK>::piece P;
  typedef P P8 __attribute__((ext_vector_type(8)));
  if constexpr (PK == 0) {
    gelu_bf16x2_x4(p);
    P8 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) { o[2 * i] = (P)p[i].x; o[2 * i + 1] = (P)p[i].y; }
    *(P8*)((P*)out + e) = o;
  } else {
    P8 hi, lo;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const f32x2_t y = gelu_fast2(p[i]);
      hi[2 * i] = (P)y.x; lo[2 * i] = (P)(y.x - (float)hi[2 * i]);
      hi[2 * i + 1] = (P)y.y; lo[2 * i + 1] = (P)(y.y - (float)hi[2 * i + 1]);
    }
    char* d = (char*)out + (e >> 5) * 128 + (e & 31) * 2;
    *(P8*)d = hi;
    *(P8*)(d + 64) = lo;
  }
}

// table[(clip,) block = q * 8 + nb][lane][8]: lane (g = lane >> 4, j = lane & 15) holds k = 8 g .. 8 g + 7 of channel 128 q + 8 j + nb
// per clip: e with max |tap coefficient| = f * 2^e, f in [0.5, 1) (GroupNorm form: the coefficients carry 1 / (standard deviation of the
// waveform), so their magnitude follows the recording level; the table stores them times 2^-e and the kernel takes the factor back out)
__global__ __launch_bounds__(256) void conv0_coef_exp_kernel(const float* __restrict__ coef, int* __restrict__ exps) {
  const int b = blockIdx.x;
  float mx = 0.f;
  for (int i = threadIdx.x; i < C0 * K0; i += 256) mx = fmaxf(mx, fabsf(coef[((long)b * C0 + i / K0) * (K0 + 1) + i % K0]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  __shared__ float part[4];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = mx;
  __syncthreads();
  if (threadIdx.x == 0) {
    mx = fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3]));
    int e = 0;
    if (mx > 0.f && mx < 3e38f) (void)frexpf(mx, &e);
    exps[b] = e > 100 ? 100 : (e < -100 ? -100 : e);
  }
}

template <int PK>
__global__ void conv0_table_kernel(const float* __restrict__ w, long w_clip_stride, int w_row, const float* __restrict__ bias,
                                   void* __restrict__ table_, const int* __restrict__ tap_exp) {
  typedef typename C0T<PK>::piece bf16_t;   // (shadows the build's operand type inside this kernel)
  typedef bf16_t bf16x8 __attribute__((ext_vector_type(8)));
  bf16_t* table = (bf16_t*)table_;
  const int b = blockIdx.y;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= TBL16) return;
  const int blk = idx >> 6, lane = idx & 63, q = blk >> 3, nb = blk & 7, g = lane >> 4, j = lane & 15;
  const int c = 128 * q + 8 * j + nb;
  const float* wc = w + (long)b * w_clip_stride + (long)c * w_row;
  const float bv = bias ? bias[c] : (w_row == K0 + 1 ? wc[K0] : 0.f);
  const float tap_scale = tap_exp ? ldexpf(1.f, -tap_exp[b]) : 1.f;   // exact; the bias column keeps its own magnitude (O(1))
  bf16x8 o;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int k = 8 * g + i;
    const float src = k < 30 ? wc[k % 10] * tap_scale : bv;
    const bf16_t hi = (bf16_t)src;
    const bf16_t lo = (bf16_t)(src - (float)hi);
    o[i] = (k < 20 || k == 30) ? hi : lo;
  }
  *(bf16x8*)(table + ((long)b * TBL16 + idx) * 8) = o;
}

template <int MODE, int PK = 0>   // MODE 0 = GroupNorm form (per-clip coefficient table, GELU), 1 = LayerNorm form (shared table, LayerNorm over channels, GELU)
__global__ __launch_bounds__(256) void conv0_mfma_kernel(const float* __restrict__ wav, int64_t L, int stride, int64_t T1,
                                                         const void* __restrict__ table_, long table_clip_stride,
                                                         const double* __restrict__ wav_mom, int64_t n_wav, float eps_wav,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta, float eps, int cpg,
                                                         void* __restrict__ out, const int* __restrict__ tap_exp) {
  typedef typename C0T<PK>::piece bf16_t;   // piece type of this instantiation
  typedef bf16_t bf16x8 __attribute__((ext_vector_type(8)));
  const bf16_t* table = (const bf16_t*)table_;
  constexpr int FPW = 64;   // frames per wave
  __shared__ __attribute__((aligned(16))) uint4 tbl[TBL16];
  __shared__ float xs[4][FPW * 5 + 16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.y;
  {
    const uint4* tg = (const uint4*)(table + (long)b * table_clip_stride);
    for (int i = tid; i < TBL16; i += 256) tbl[i] = tg[i];
  }
  float mu = 0.f, rn = 1.f, xscale = 1.f, bscale = 1.f;
  if (MODE == 1 && wav_mom) {
    const double* wm = wav_mom + 2 * (b / cpg);
    const double m = wm[0] / (double)n_wav;
    const double var = wm[1] / (double)n_wav - m * m;
    mu = (float)m;
    rn = (float)(1.0 / sqrt(var + (double)eps_wav));
  }
  const int64_t t0 = ((int64_t)blockIdx.x * 4 + wave) * FPW;
  const int64_t left = T1 - t0;
  const int nfr = left >= FPW ? FPW : (left > 0 ? (int)left : 0);
  if (nfr > 0) {
    const float* x = wav + (int64_t)b * L + t0 * stride;
    const int ns = (nfr - 1) * stride + K0;
    float amax = 0.f;
    for (int i = lane; i < ns; i += 64) {
      const float v = MODE == 1 ? (x[i] - mu) * rn : x[i];
      xs[wave][i] = v;
      amax = fmaxf(amax, fabsf(v));
    }
    if constexpr (MODE == 0) {
      // GroupNorm form: the samples are RAW (the waveform norm lives in the coefficient table), and a 16-bit (hi, lo) pair only carries
      // fp32-grade precision while lo is a normal number: IEEE-half pieces of audio peaking at 0.01 have subnormal lo pieces (3e-6
      // relative instead of 2^-22), samples above 65 504 overflow, and the coefficients (~ 1 / level) leave the range the other way.
      // Both sides are therefore scaled by powers of two -- exact, as is taking them out of the accumulators again: the wave's strip by
      // the 2^kx that brings its peak to [0.5, 1), the clip's tap coefficients by 2^ka (conv0_coef_exp_kernel), and the two bias
      // columns of A hold 2^(kx + ka) against the unscaled bias:  acc = 2^(kx + ka) (x a + b).
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
      int e = 0;
      if (amax > 0.f && amax < 3e38f) { (void)frexpf(amax, &e); }      // amax = f * 2^e, f in [0.5, 1)
      e = e > 100 ? 100 : (e < -100 ? -100 : e);
      int tot = -e - (tap_exp ? tap_exp[b] : 0);                         // kx + ka: ~ log2(level / peak) - 3, a small negative number
      tot = tot > 15 ? 15 : (tot < -14 ? -14 : tot);                     // 2^tot is a normal IEEE half
      xscale = ldexpf(1.f, -e);
      bscale = ldexpf(1.f, tot);
      // (if tot had to be clamped the strip scale follows, so that all three column groups keep ONE common factor)
      xscale = ldexpf(1.f, tot + (tap_exp ? tap_exp[b] : 0));
    }
  }
  __syncthreads();
  if (nfr <= 0) return;
  const float unscale = 1.f / bscale;   // a power of two as well
  const int g = lane >> 4, j = lane & 15;
  const bf16_t one = (bf16_t)bscale;
  for (int chunk = 0; chunk * 16 < nfr; ++chunk) {
    // ---- A fragment: frame (row) j of the chunk, k = 8 g .. 8 g + 7 of [ x_hi | x_lo | x_hi | 1 | 1 ]
    bf16x8 a;
    {
      int f = chunk * 16 + j;
      if (f > nfr - 1) f = nfr - 1;
      const float* xp = &xs[wave][f * stride];
      bf16_t h[K0], l[K0];
#pragma unroll
      for (int i = 0; i < K0; ++i) {
        const float v = xp[i] * xscale;
        h[i] = (bf16_t)v;
        l[i] = (bf16_t)(v - (float)h[i]);
      }
      const bf16x8 a0 = {h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]};
      const bf16x8 a1 = {h[8], h[9], l[0], l[1], l[2], l[3], l[4], l[5]};
      const bf16x8 a2 = {l[6], l[7], l[8], l[9], h[0], h[1], h[2], h[3]};
      const bf16x8 a3 = {h[4], h[5], h[6], h[7], h[8], h[9], one, one};
      a = g == 0 ? a0 : (g == 1 ? a1 : (g == 2 ? a2 : a3));
    }
    // this lane's output frames: 4 g + r of the chunk; channels 128 q + 8 j .. + 7
    const int64_t e0 = ((int64_t)b * T1 + t0 + chunk * 16 + 4 * g) * C0 + 8 * j;   // element index of (frame 4 g of the chunk, channel 8 j)
    const int fr0 = chunk * 16 + 4 * g;
    if constexpr (MODE == 0) {
#pragma unroll 1
      for (int q = 0; q < 4; ++q) {   // (not unrolled: 8 accumulators live, not 32)
        f32x4 acc[8];
#pragma unroll
        for (int nb = 0; nb < 8; ++nb) {
          acc[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
          acc[nb] = c0_mma<PK>(a, tbl[(q * 8 + nb) * 64 + lane], acc[nb]);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          f32x2_t p[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) p[i] = f32x2_t{acc[2 * i][r], acc[2 * i + 1][r]} * unscale;
          if (fr0 + r < nfr) c0_store8<PK>(out, e0 + (int64_t)r * C0 + 128 * q, p);
        }
      }
    } else {
      f32x4 acc[32];
#pragma unroll
      for (int blk = 0; blk < 32; ++blk) {
        acc[blk] = f32x4{0.f, 0.f, 0.f, 0.f};
        acc[blk] = c0_mma<PK>(a, tbl[blk * 64 + lane], acc[blk]);
      }
      // LayerNorm statistics of the lane's four frames: 32 channels in registers, the other 480 in the 15 lanes with the same g
      float mean[4], rstd[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float s = 0.f;
#pragma unroll
        for (int blk = 0; blk < 32; ++blk) s += acc[blk][r];
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        mean[r] = s * (1.f / C0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float qq = 0.f;
#pragma unroll
        for (int blk = 0; blk < 32; ++blk) { const float d = acc[blk][r] - mean[r]; qq = fmaf(d, d, qq); }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) qq += __shfl_xor(qq, o, 64);
        rstd[r] = rsqrtf(qq * (1.f / C0) + eps);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        __builtin_amdgcn_sched_barrier(0);   // one group's scale / shift vectors live at a time (hoisted, all four cost 64 registers)
        const float4 g0 = *(const float4*)(gamma + 128 * q + 8 * j), g1 = *(const float4*)(gamma + 128 * q + 8 * j + 4);
        const float4 b0 = *(const float4*)(beta + 128 * q + 8 * j), b1 = *(const float4*)(beta + 128 * q + 8 * j + 4);
        const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w}, bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          f32x2_t p[4];
#pragma unroll
          for (int i = 0; i < 4; ++i)
            p[i] = f32x2_t{fmaf((acc[q * 8 + 2 * i][r] - mean[r]) * rstd[r], gg[2 * i], bb[2 * i]),
                           fmaf((acc[q * 8 + 2 * i + 1][r] - mean[r]) * rstd[r], gg[2 * i + 1], bb[2 * i + 1])};
          if (fr0 + r < nfr) c0_store8<PK>(out, e0 + (int64_t)r * C0 + 128 * q, p);
        }
      }
    }
  }
}

}  // namespace

int g_conv0_mfma = 1;   // svt_debug_set key 22: 0 = the vector-ALU conv0 kernels in the 16-bit modes too (A/B)

// prec: storage precision (1 = 16-bit rows out); pair_kind 2 / 3 = split modes writing pair rows (fp32 storage)
bool conv0_mfma_ok(int prec, int pair_kind, int k, int stride, int C) {
  return g_conv0_mfma && (prec == 1 || pair_kind == 2 || pair_kind == 3) && k == K0 && stride >= 1 && stride <= 5 && C == C0;
}
size_t conv0_mfma_table_bytes(int B) { return (size_t)B * TBL16 * 16 + (((size_t)B * 4 + 255) & ~(size_t)255); }   // tables + one exponent per clip

#define SVT_C0_DISPATCH(PKV, STMT)                 \
  if ((PKV) == 3) { constexpr int PK_ = 3; STMT }  \
  else if ((PKV) == 2) { constexpr int PK_ = 2; STMT } \
  else { constexpr int PK_ = 0; STMT }

// GroupNorm form: coef (B, C, 11) from conv0_group_coef_kernel -> per-clip tables -> out (B, T1, C): 16-bit rows, or pair rows (pair_kind)
int launch_conv0_mfma_group(const float* wav, int B, int64_t L, int stride, int64_t T1, const float* coef, void* table_ws, void* out,
                            hipStream_t s, int pair_kind) {
  if (((uintptr_t)table_ws & 15) || ((uintptr_t)out & 127)) { set_error("conv0_mfma: alignment"); return -1; }
  dim3 grid((unsigned)((T1 + 255) / 256), B);
  int* exps = (int*)((char*)table_ws + (size_t)B * TBL16 * 16);
  hipLaunchKernelGGL(conv0_coef_exp_kernel, dim3(B), dim3(256), 0, s, coef, exps);
  SVT_C0_DISPATCH(pair_kind,
    hipLaunchKernelGGL((conv0_table_kernel<PK_>), dim3(TBL16 / 256, B), dim3(256), 0, s, coef, (long)C0 * (K0 + 1), K0 + 1, (const float*)nullptr,
                       table_ws, (const int*)exps);
    hipLaunchKernelGGL((conv0_mfma_kernel<0, PK_>), grid, dim3(256), 0, s, wav, L, stride, T1, (const void*)table_ws, (long)TBL16 * 8,
                       (const double*)nullptr, (int64_t)1, 0.f, (const float*)nullptr, (const float*)nullptr, 0.f, 1, out, (const int*)exps);)
  SVT_LAUNCH_CHECK();
  return 0;
}

// LayerNorm form: conv (w0 (C, 10), b0 (C) or null) on the normalised waveform -> LayerNorm(gamma, beta) -> GELU
int launch_conv0_mfma_layer(const float* wav, int B, int64_t L, int stride, int64_t T1, const double* wav_moments, int64_t n_wav,
                            float eps_wav, const float* w0, const float* b0, const float* gamma, const float* beta, float eps, void* table_ws,
                            void* out, hipStream_t s, int cpg, int pair_kind) {
  if (((uintptr_t)table_ws & 15) || ((uintptr_t)out & 127) || ((uintptr_t)gamma & 15) || ((uintptr_t)beta & 15)) { set_error("conv0_mfma: alignment"); return -1; }
  dim3 grid((unsigned)((T1 + 255) / 256), B);
  SVT_C0_DISPATCH(pair_kind,
    hipLaunchKernelGGL((conv0_table_kernel<PK_>), dim3(TBL16 / 256, 1), dim3(256), 0, s, w0, 0L, K0, b0, table_ws, (const int*)nullptr);
    hipLaunchKernelGGL((conv0_mfma_kernel<1, PK_>), grid, dim3(256), 0, s, wav, L, stride, T1, (const void*)table_ws, 0L, wav_moments, n_wav,
                       eps_wav, gamma, beta, eps, cpg, out, (const int*)nullptr);)
  SVT_LAUNCH_CHECK();
  return 0;
}
#undef SVT_C0_DISPATCH

}  // namespace svt

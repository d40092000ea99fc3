// Attention output projection fused with the residual add and the LayerNorm that follows it (post-LN encoders,
// hidden size 768, throughput mode):
//     y = LN( A W^T + bias + (r_hi + r_lo) ) -> (y_hi, y_lo) [+ fp32 copy]
// A (M,768) bf16 = attention output, W (768,768) bf16, residual stream and result as bf16 (hi, lo) pairs (kernels.hip,
// layernorm_hilo_kernel).  Unfused this is a 252-tile GEMM that is 40 % prologue / epilogue (K is only 12 slabs deep,
// 29 us at 650 TFLOP/s) plus an HBM-bound LayerNorm pass (23 us); fused, the projection result never leaves registers.
//
// One workgroup owns 64 rows x ALL 768 columns (a LayerNorm needs whole rows): 8 waves as 2 (M) x 4 (N), wave tile
// 32 x 192 = 2 x 12 MFMA blocks (96 accumulator registers).  The 768-row W slab is 12x the A slab, so the kernel is
// bound by the L2 -> LDS fill (every workgroup streams all of W: 1.2 MB), not by the matrix pipe.  K is consumed in
// 64-deep slabs of full 128-byte lines (a first version with 32-deep slabs = 64-byte rows, i.e. half-line DMA requests,
// ran at half the fill rate and was no faster than the two unfused kernels); a slab of all 768 W rows would be 96 KB,
// so a ring stage holds HALF of every wave's MFMA blocks (384 rows, 48 KB): three W stages + two A slabs = exactly
// 160 KiB, one raw barrier and one counted vmcnt per stage, every wave busy in every stage.  LDS image as in gemm_dma.hip
// (row-major 128-byte rows, chunk c of row r at slot c ^ (r & 7)).  W rows are fetched permuted so a lane ends with 16
// consecutive columns per 64-column group.  Row statistics: in-register over a lane's 48 values, two shuffles across the 4 lanes
// of a row, one LDS exchange across the 4 column waves (twice: mean, then variance about the mean, as the unfused kernel).
#include "common.h"

namespace svt {
namespace {

typedef const void __attribute__((address_space(1)))* gptr_t;
typedef void __attribute__((address_space(3)))* lptr_t;
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

struct OutLnArgs {
  const bf16_t* A; const bf16_t* W; const float* bias;
  const bf16_t* rh; const bf16_t* rl;
  const float* gamma; const float* beta;
  bf16_t* yh; bf16_t* yl; float* yF;
  int M, K;
  long lda;
  float eps;
};

constexpr int kD = 768, kBM = 64, kBK = 64;
constexpr int kStageW = 384 * 8;        // uint4 per W stage: half of W's rows (384) x one 64-deep slab (128-byte rows)
constexpr int kStageA = kBM * 8;        // uint4 per A slab (64 rows x 128 bytes)
constexpr int kLdsU4 = 3 * kStageW + 2 * kStageA;   // 10 240 uint4 = 160 KiB

__global__ __launch_bounds__(512) void outproj_ln_kernel(OutLnArgs p) {
  // [W stage 0][W stage 1][W stage 2][A slab even][A slab odd]
  extern __shared__ __attribute__((aligned(16))) uint4 lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int m0 = blockIdx.x * kBM;

  // ---- DMA sources: full 128-byte lines, 8 rows per wave instruction, lane (row r8, slot) takes chunk slot ^ r8 ----
  const int r8 = lane >> 3, ch = (lane & 7) ^ (lane >> 3);
  const bf16_t* asrc;   // 8 A instructions per slab, one per wave: rows wave*8 + r8
  {
    int m = m0 + wave * 8 + r8;
    if (m > p.M - 1) m = p.M - 1;
    asrc = p.A + (long)m * p.lda + ch * 8;
  }
  // W: a stage holds the rows of one HALF of every wave's 12 MFMA blocks (blocks 6h .. 6h+5), so all waves work in every
  // stage.  Stage-local row rho (0..383): column wave cw = rho / 96, block nb = 6h + (rho % 96) / 16, MFMA row i16;
  // it carries output column cw*192 + (nb/4)*64 + (i16>>2)*16 + (nb%4)*4 + (i16&3)  (16 consecutive columns per lane
  // and 64-column group).  6 instructions per wave per stage.
  const bf16_t* wsrc[2][6];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int rho = (wave * 6 + i) * 8 + r8;
      const int cw = rho / 96, rem = rho % 96, nb = h * 6 + (rem >> 4), i16 = rem & 15;
      const int n = cw * 192 + (nb >> 2) * 64 + (i16 >> 2) * 16 + (nb & 3) * 4 + (i16 & 3);
      wsrc[h][i] = p.W + (long)n * p.K + ch * 8;
    }
  // stage s = (slab s >> 1, half s & 1); W ring slot s % 3; the A slab travels with the even stage
  // Every workgroup streams the SAME 1.2 MB of W: in lockstep they would all ask one L2 channel for one line at a
  // time.  Workgroup b therefore walks the K slabs rotated by b (mod nk): at any moment the chip reads nk different
  // slabs (46.7 -> see DESIGN.md).  The fp32 accumulation order over K differs per workgroup, deterministically.
  const int nk = p.K / kBK;
  const int rot = blockIdx.x % nk;
  auto issue = [&](int s2, int half) {
    const int j = s2 >> 1;                      // position in this workgroup's walk
    int slab = j + rot;
    if (slab >= nk) slab -= nk;
    uint4* wdst = lds + (s2 % 3) * kStageW;
    if (half == 0)
      __builtin_amdgcn_global_load_lds((gptr_t)(asrc + slab * kBK), (lptr_t)(lds + 3 * kStageW + (j & 1) * kStageA + wave * 64), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < 6; ++i)
      __builtin_amdgcn_global_load_lds((gptr_t)(wsrc[half][i] + slab * kBK), (lptr_t)(wdst + (wave * 6 + i) * 64), 16, 0, 0);
  };

  f32x4 acc[12][2];
#pragma unroll
  for (int i = 0; i < 12; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment offsets inside a 16-row block (uint4): group r16>>3, row rr8, chunk (ks*4 + cq) at slot chunk ^ rr8
  const int r16 = lane & 15, cq = lane >> 4, rr8 = r16 & 7;
  const int frag0 = (r16 >> 3) * 64 + rr8 * 8 + (cq ^ rr8);
  const int frag1 = (r16 >> 3) * 64 + rr8 * 8 + ((4 + cq) ^ rr8);
  const int xoff = 3 * kStageW + (wm * 2) * 128;     // A: 16-row blocks wm*2 + mb, 128 uint4 each
  const int woff = (wn * 6) * 128;                   // W stage: blocks wn*6 + nbh

  const int nst = 2 * nk;
  issue(0, 0);
  issue(1, 1);
#define SVT_STAGE(S, HALF)                                                                                       \
  {                                                                                                              \
    if ((S) + 1 < nst) { if (HALF) wait_vm<7>(); else wait_vm<6>(); } else wait_vm<0>();                         \
    __builtin_amdgcn_s_barrier();                                                                                \
    if ((S) + 2 < nst) issue((S) + 2, HALF);                                                                     \
    const uint4* ws_ = lds + ((S) % 3) * kStageW + woff;                                                         \
    const uint4* xs_ = lds + xoff + (((S) >> 1) & 1) * kStageA;                                                  \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                                           \
      bf16x8 xf[2];                                                                                              \
      _Pragma("unroll") for (int mb = 0; mb < 2; ++mb) xf[mb] = __builtin_bit_cast(bf16x8, xs_[mb * 128 + (ks ? frag1 : frag0)]); \
      _Pragma("unroll") for (int nbh = 0; nbh < 6; ++nbh) {                                                      \
        const bf16x8 wf = __builtin_bit_cast(bf16x8, ws_[nbh * 128 + (ks ? frag1 : frag0)]);                     \
        _Pragma("unroll") for (int mb = 0; mb < 2; ++mb)                                                         \
          acc[(HALF) * 6 + nbh][mb] = SVT_MFMA_16x16x32(wf, xf[mb], acc[(HALF) * 6 + nbh][mb]); \
      }                                                                                                          \
    }                                                                                                            \
  }
  for (int s = 0; s < nst; s += 2) {
    SVT_STAGE(s, 0)
    SVT_STAGE(s + 1, 1)
  }
#undef SVT_STAGE
  __syncthreads();  // the ring is free: reuse it for the row-statistics exchange

  // ---- epilogue.  lane (row r16 of block mb, q = lane >> 4) holds, per 64-column group gq, 16 consecutive columns:
  //      col = wn*192 + gq*64 + q*16 + nbl*4 + r   <->   acc[gq*4 + nbl][mb][r]
  float* red = (float*)lds;  // [4 column waves][64 rows]
  const int q = lane >> 4;
  float mean[2], rstd[2];
  long rowoff[2];
  bool live[2];
#pragma unroll
  for (int mb = 0; mb < 2; ++mb) {
    const int m = m0 + wm * 32 + mb * 16 + r16;
    live[mb] = m < p.M;
    rowoff[mb] = (long)(live[mb] ? m : p.M - 1) * kD;
  }
  // x = acc + bias + residual
#pragma unroll
  for (int gq = 0; gq < 3; ++gq) {
    const int c0 = wn * 192 + gq * 64 + q * 16;
    float bv[16];
#pragma unroll
    for (int j4 = 0; j4 < 4; ++j4) {
      const float4 b4 = *(const float4*)(p.bias + c0 + j4 * 4);
      bv[j4 * 4] = b4.x; bv[j4 * 4 + 1] = b4.y; bv[j4 * 4 + 2] = b4.z; bv[j4 * 4 + 3] = b4.w;
    }
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
      const bf16x8 h0 = *(const bf16x8*)(p.rh + rowoff[mb] + c0), h1 = *(const bf16x8*)(p.rh + rowoff[mb] + c0 + 8);
      const bf16x8 l0 = *(const bf16x8*)(p.rl + rowoff[mb] + c0), l1 = *(const bf16x8*)(p.rl + rowoff[mb] + c0 + 8);
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const float rr = j < 8 ? (float)h0[j] + (float)l0[j] : (float)h1[j - 8] + (float)l1[j - 8];
        acc[gq * 4 + (j >> 2)][mb][j & 3] += bv[j] + rr;
      }
    }
  }
  // mean
#pragma unroll
  for (int mb = 0; mb < 2; ++mb) {
    float s = 0.f;
#pragma unroll
    for (int nb = 0; nb < 12; ++nb) s += (acc[nb][mb][0] + acc[nb][mb][1]) + (acc[nb][mb][2] + acc[nb][mb][3]);
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    if (q == 0) red[wn * 64 + wm * 32 + mb * 16 + r16] = s;
  }
  __syncthreads();
#pragma unroll
  for (int mb = 0; mb < 2; ++mb) {
    const int r = wm * 32 + mb * 16 + r16;
    mean[mb] = (red[r] + red[64 + r] + red[128 + r] + red[192 + r]) * (1.f / kD);
  }
  __syncthreads();
  // variance about the mean
#pragma unroll
  for (int mb = 0; mb < 2; ++mb) {
    float s = 0.f;
#pragma unroll
    for (int nb = 0; nb < 12; ++nb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        acc[nb][mb][r] -= mean[mb];
        s = fmaf(acc[nb][mb][r], acc[nb][mb][r], s);
      }
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    if (q == 0) red[wn * 64 + wm * 32 + mb * 16 + r16] = s;
  }
  __syncthreads();
#pragma unroll
  for (int mb = 0; mb < 2; ++mb) {
    const int r = wm * 32 + mb * 16 + r16;
    rstd[mb] = rsqrtf((red[r] + red[64 + r] + red[128 + r] + red[192 + r]) * (1.f / kD) + p.eps);
  }
  // normalise, scale / shift, split into (hi, lo), store
#pragma unroll
  for (int gq = 0; gq < 3; ++gq) {
    const int c0 = wn * 192 + gq * 64 + q * 16;
    float gv[16], bb[16];
#pragma unroll
    for (int j4 = 0; j4 < 4; ++j4) {
      const float4 g4 = *(const float4*)(p.gamma + c0 + j4 * 4), b4 = *(const float4*)(p.beta + c0 + j4 * 4);
      gv[j4 * 4] = g4.x; gv[j4 * 4 + 1] = g4.y; gv[j4 * 4 + 2] = g4.z; gv[j4 * 4 + 3] = g4.w;
      bb[j4 * 4] = b4.x; bb[j4 * 4 + 1] = b4.y; bb[j4 * 4 + 2] = b4.z; bb[j4 * 4 + 3] = b4.w;
    }
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
      if (!live[mb]) continue;
      float o[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) o[j] = fmaf(acc[gq * 4 + (j >> 2)][mb][j & 3] * rstd[mb], gv[j], bb[j]);
#pragma unroll
      for (int h8 = 0; h8 < 2; ++h8) {
        bf16x8 oh, ol;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          oh[j] = (bf16_t)o[h8 * 8 + j];
          ol[j] = (bf16_t)(o[h8 * 8 + j] - (float)oh[j]);
        }
        *(bf16x8*)(p.yh + rowoff[mb] + c0 + h8 * 8) = oh;
        *(bf16x8*)(p.yl + rowoff[mb] + c0 + h8 * 8) = ol;
      }
      if (p.yF) {
#pragma unroll
        for (int j4 = 0; j4 < 4; ++j4)
          *(float4*)(p.yF + rowoff[mb] + c0 + j4 * 4) = float4{o[j4 * 4], o[j4 * 4 + 1], o[j4 * 4 + 2], o[j4 * 4 + 3]};
      }
    }
  }
}

}  // namespace

bool outproj_ln_eligible(int D, int K) { return D == kD && K % kBK == 0 && K >= kBK; }

// y = LN(A W^T + bias + rh + rl) -> (yh, yl[, yF]); A (M,K) bf16 with row pitch lda, W (768,K) bf16; yh/yl may alias rh/rl
int launch_outproj_ln(const void* A, long lda, const void* W, const float* bias, const void* rh, const void* rl, int M, int K,
                      const float* gamma, const float* beta, float eps, void* yh, void* yl, float* yF, hipStream_t s) {
  OutLnArgs a;
  a.A = (const bf16_t*)A; a.W = (const bf16_t*)W; a.bias = bias; a.rh = (const bf16_t*)rh; a.rl = (const bf16_t*)rl;
  a.gamma = gamma; a.beta = beta; a.yh = (bf16_t*)yh; a.yl = (bf16_t*)yl; a.yF = yF;
  a.M = M; a.K = K; a.lda = lda; a.eps = eps;
  const size_t lds_bytes = (size_t)kLdsU4 * 16;
  if (int r_ = ensure_dyn_lds((const void*)outproj_ln_kernel, (int)lds_bytes)) return r_;
  prof_begin(s);
  hipLaunchKernelGGL(outproj_ln_kernel, dim3((M + kBM - 1) / kBM), dim3(512), lds_bytes, s, a);
  prof_end(s, 2.0 * M * (double)kD * K, ((double)M * K + (double)kD * K + 4.0 * M * kD) * 2 + (yF ? 4.0 * M * kD : 0.0), 0);
  SVT_LAUNCH_CHECK();
  return 0;
}

}  // namespace svt

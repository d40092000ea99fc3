// 3x3 stride-1 pad-1 convolution, 64 -> 64 channels, + folded BatchNorm bias (+ residual) + PReLU over the zero-haloed channels-last
// tensor [F][Hs + 2][Ws + 2][64] of the lip front-end's first trunk stage (reference: N20EMv2/video_only/resnet.py:27-62 BasicBlock,
// :76-87 layer1; 4 of the trunk's 16 convolutions and a third of its time on the GEMM kernels, whose 256-wide tiles are fed 64 / 128
// useful columns there).  16-bit storage modes only (bf16 / IEEE half build); round 4.
//
// Shape of the problem: N = 64 output channels is one quarter of a GEMM tile, K = 576 re-reads every input pixel nine times
// through an im2col view, and each conv moves 0.5 GB in and 0.5 GB out (+ 0.5 GB residual) for 0.29 TFLOP -- memory and matrix
// time are of the same size.  So: ONE persistent workgroup per CU, 8 waves = 4 position groups x 2 channel halves.
//   weights  a wave's half of the 3x3 kernel (32 output channels x 576) lives in its REGISTERS as 36 ready-made MFMA A fragments
//            (144 VGPRs, fetched once per kernel; the row permutation of the fragment leaves a lane of the result 8 consecutive
//            channels of one pixel = one 16-byte store, and the four lanes of a pixel its whole 64-byte half);
//   frames   two padded frames (FP = (Hs+2)(Ws+2) pixels x 128 B, XOR-swizzled 16-byte chunks) in LDS, the next one filled by
//            LDS-DMA while this one is computed: every input pixel is fetched from memory once;
//   flat     output position p = y (Ws+2) + x reads input positions p + ky (Ws+2) + kx -- a tap is an address offset; the two
//            garbage columns per row (x >= Ws) are computed and not stored (the output's halo stays zero);
//   work     a frame = ceil(Hs (Ws+2) / 48) triples of 16-position blocks dealt round-robin to the position groups; per triple
//            and wave 9 taps x 2 k-steps x (3 LDS fragment reads + 6 v_mfma_f32_16x16x32): 0.5 of the LDS port at full matrix
//            rate, two waves per SIMD to cover the reads and each other's epilogue;
//   sync     ONE s_waitcnt vmcnt(0) + s_barrier per frame (the waves run free inside a frame).
#include "common.h"

namespace svt {
namespace {

// timing ablations (make DIAG=1; svt_debug_set key 25, bits 4..6 = no stores / no next-frame requests / no fragment reads after a
// triple's first; tools/c3_ablate.sh): compiled out of the shipped library
#ifdef SVT_DIAG
#define C3_DBG(bit) (dbg & (bit))
#else
#define C3_DBG(bit) 0
#endif
constexpr int C3_SLACK = 56;  // ring slots behind a frame image that garbage positions of the last triple may read

__device__ __forceinline__ void c3_dma16(const void* gsrc, unsigned lds_byte_addr) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(lds_byte_addr) : "memory");
}

template <bool RESID>
__global__ __launch_bounds__(512) void conv3x3_c64_kernel(const bf16_t* __restrict__ in, const uint4* __restrict__ wimg,
                                                          const float* __restrict__ bias, const float* __restrict__ slope,
                                                          const bf16_t* __restrict__ resid, bf16_t* __restrict__ out, int F, int Hs,
                                                          int Ws, int dbg) {
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int pg = wave >> 1, ch = wave & 1;  // position group, channel half
  const int Wp = Ws + 2, FP = (Hs + 2) * Wp, NV = Hs * Wp;
  const int MB = (NV + 15) >> 4, NT = (MB + 2) / 3, NG = (FP + 7) >> 3;
  const unsigned fbytes = (unsigned)(NG * 8 + C3_SLACK) * 128u;
  const unsigned lds0 = (unsigned)(size_t)lds;
  const int G = gridDim.x;

  // frame tf -> buffer b; lane (slot = lane >> 3, position q = lane & 7) of a group fetches chunk q ^ slot
  const int dsl = lane >> 3, dch = (lane & 7) ^ dsl;
  // requested by the two waves of position group 3 alone: the group with the fewest triples (round-robin dealing), so the one
  // vmcnt(0) a frame needs -- which also waits for the waves' own stores -- is paid by waves that would idle at the barrier anyway
  auto request = [&](long tf, int b) {
    const bf16_t* fb = in + tf * (long)FP * 64;
    for (int g = ch; g < NG; g += 2) {
      int px = g * 8 + dsl;
      px = px < FP ? px : FP - 1;
      c3_dma16(fb + (long)px * 64 + dch * 8, lds0 + (unsigned)b * fbytes + (unsigned)g * 1024u);
    }
  };
  if ((int)blockIdx.x < F && pg == 3) request(blockIdx.x, 0);

  const int n = lane & 15, kq = lane >> 4;
  bf16x8 wr[9][2][2];
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) wr[tap][ks][nb] = __builtin_bit_cast(bf16x8, wimg[((tap * 2 + ks) * 4 + ch * 2 + nb) * 64 + lane]);
  const int c0 = ch * 32 + kq * 8;  // the lane's 8 output channels (the four lanes of a pixel: 64 contiguous bytes)
  const float rWp = 1.0f / (float)Wp;
  // bias / slope behind the two frame buffers (registers are for the weights)
  float* bs = (float*)(lds + 2 * fbytes);
  if (tid < 64) { bs[tid] = bias[tid]; bs[64 + tid] = slope[tid]; }

  // the requesting waves wait for their requests (vmcnt(0)) in front of the epilogue of their LAST triple of a frame: the requests are
  // a frame old by then, and the stores of that epilogue stay in flight across the barrier
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // lgkmcnt: the bs stores reach the LDS before the first barrier lets readers through
  int buf = 0;
  for (long f = blockIdx.x; f < F; f += G, buf ^= 1) {
    if (pg >= NT) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (f + G < F && !C3_DBG(2) && pg == 3) request(f + G, buf ^ 1);   // the other buffer: every wave is done with the frame before this one
    const unsigned fb0 = (unsigned)buf * fbytes;
    for (int t = pg; t < NT; t += 4) {
      const int pl = t * 48 + n;
      // fragment reads two (tap, k-step) groups = 12 MFMAs ahead of their use (two waves per SIMD do not cover an LDS round trip per
      // pair of MFMAs); three groups of three fragments in registers
      constexpr int RD = 3;   // groups of fragments in registers (2 measured the same, 4 spills)
      bf16x8 xb[RD][3];
      auto fetch = [&](int g, bf16x8 (&x)[3]) {
        const int tap = g >> 1;
        const int sl = pl + (tap / 3) * Wp + (tap % 3);
        const unsigned a0 = fb0 + (unsigned)(sl * 128 + (((kq ^ sl) & 3) << 4) + ((sl & 4) << 4));
        const unsigned ak = (g & 1) ? (a0 ^ 64u) : a0;
#pragma unroll
        for (int mb = 0; mb < 3; ++mb) x[mb] = *(const bf16x8*)(lds + ak + mb * 2048);
      };
      fetch(0, xb[0]);
      if (RD > 2) fetch(1, xb[1]);
      f32x4 acc[3][2];
#pragma unroll
      for (int mb = 0; mb < 3; ++mb)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) acc[mb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
      bf16x8 rs[3];
#pragma unroll
      for (int g = 0; g < 18; ++g) {
        if (g + RD - 1 < 18 && !(C3_DBG(4) && g > 0)) fetch(g + RD - 1, xb[(g + RD - 1) % RD]);
        if (RESID && g == 16) {   // as late as the registers allow: 12 VGPRs that the retiring fragment groups free (earlier = spills)
#pragma unroll
          for (int mb = 0; mb < 3; ++mb) {
            const int p = pl + mb * 16;
            rs[mb] = *(const bf16x8*)(resid + (f * (long)FP + (p < NV ? p : 0) + Wp + 1) * 64 + c0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mb = 0; mb < 3; ++mb)
#pragma unroll
          for (int nb = 0; nb < 2; ++nb) acc[mb][nb] = SVT_MFMA_16x16x32(wr[g >> 1][g & 1][nb], xb[g % RD][mb], acc[mb][nb]);
        __builtin_amdgcn_sched_barrier(0);
      }
      // position p -> padded pixel p + Wp + 1 of the output frame; lane = pixel n of the block, channels c0 .. c0 + 7
      if (pg == 3 && t + 4 >= NT) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const float4 b0 = *(const float4*)(bs + c0), b1 = *(const float4*)(bs + c0 + 4);
      const float4 s0 = *(const float4*)(bs + 64 + c0), s1 = *(const float4*)(bs + 64 + c0 + 4);
      const float bv[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w}, sv[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
#pragma unroll
      for (int mb = 0; mb < 3; ++mb) {
        const int p = pl + mb * 16;
        const int y = (int)(((float)p + 0.5f) * rWp);
        if (p < NV && p - y * Wp < Ws && !C3_DBG(1)) {
          bf16x8 o;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            float a = acc[mb][j >> 2][j & 3] + bv[j];
            if (RESID) a += (float)rs[mb][j];
            o[j] = (bf16_t)(a > 0.f ? a : a * sv[j]);
          }
          *(bf16x8*)(out + (f * (long)FP + p + Wp + 1) * 64 + c0) = o;
        }
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------------
// The same idea for the second trunk stage (128 -> 128 channels, 13 x 13 padded frames at an 88-pixel ROI; N20EMv2/video_only/
// resnet.py:76-87 layer2's stride-1 convolutions): the GEMM kernels run these 128 columns wide at 0.45-0.66 PFLOP/s, bound by the
// LDS-DMA of an im2col view that fetches every pixel nine times.  The 3x3 kernel is 295 KB -- exactly the eight waves' weight
// registers (8 x 144 VGPRs x 256 B) -- so the split is by OUTPUT-CHANNEL QUARTER x INPUT-CHANNEL HALF: wave (cq, kh) holds 32 output
// channels x 9 taps x 64 input channels as 36 A fragments, every wave walks ALL positions of the frame (triples of 16-position
// blocks) reading only its input-channel half of each pixel (0.5 of the LDS port at full matrix rate), and the two input-channel
// halves of a channel quarter meet through LDS: per triple one wave of the pair parks its 24 accumulator registers in a scratch
// slot, the workgroup's barrier of the triple passes, the other adds them and runs the epilogue while its partner is already in
// the next triple (the roles alternate by triple and frame).  Frames double-buffered as above; pixels are 256 B, chunks
// XOR-swizzled by the slot's low four bits.
template <bool RESID>
__global__ __launch_bounds__(512) void conv3x3_c128_kernel(const bf16_t* __restrict__ in, const uint4* __restrict__ wimg,
                                                           const float* __restrict__ bias, const float* __restrict__ slope,
                                                           const bf16_t* __restrict__ resid, bf16_t* __restrict__ out, int F, int Hs,
                                                           int Ws, unsigned fbytes) {
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cq = wave >> 1, kh = wave & 1;
  const int Wp = Ws + 2, FP = (Hs + 2) * Wp, NV = Hs * Wp;
  const int MB = (NV + 15) >> 4, NT = (MB + 2) / 3, NG = (FP + 3) >> 2;
  const unsigned lds0 = (unsigned)(size_t)lds;
  const int G = gridDim.x;
  float* scratch = (float*)(lds + 2 * fbytes);                 // [pair cq][triple parity][6][64 lanes] x 16 B
  float* bs = scratch + 4 * 2 * 6 * 64 * 4;                    // bias[128], slope[128]

  // frame tf -> buffer b: a request moves 4 pixels; lane (slot = lane >> 4, position q = lane & 15) fetches chunk q ^ (slot index & 15)
  auto request = [&](long tf, int b) {
    const bf16_t* fb = in + tf * (long)FP * 128;
    for (int g = wave; g < NG; g += 8) {
      int px = g * 4 + (lane >> 4);
      const int c = (lane & 15) ^ (px & 15);
      px = px < FP ? px : FP - 1;
      c3_dma16(fb + (long)px * 128 + c * 8, lds0 + (unsigned)b * fbytes + (unsigned)g * 1024u);
    }
  };
  if ((int)blockIdx.x < F) request(blockIdx.x, 0);

  const int n = lane & 15, kq = lane >> 4;
  bf16x8 wr[9][2][2];
  {
    const uint4* wsrc = wimg + (long)wave * (36 * 64) + lane;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) wr[tap][ks][nb] = __builtin_bit_cast(bf16x8, wsrc[((tap * 2 + ks) * 2 + nb) * 64]);
  }
  const int c0 = cq * 32 + kq * 8;  // the lane's 8 output channels
  const float rWp = 1.0f / (float)Wp;
  if (tid < 128) { bs[tid] = bias[tid]; bs[128 + tid] = slope[tid]; }
  // vmcnt: the first frame's requests; lgkmcnt: the bs stores above -- a raw s_barrier does not wait for a wave's own LDS writes
  // on gfx950 (back-off barrier: hipcc inserts no s_waitcnt in front of it), and the readers sit on other SIMDs
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  int buf = 0, fc = 0;
  for (long f = blockIdx.x; f < F; f += G, buf ^= 1, ++fc) {
    if (f + G < F) request(f + G, buf ^ 1);   // the other buffer: the barrier of the previous frame's last triple is behind every wave
    const unsigned fb0 = (unsigned)buf * fbytes;
#pragma unroll 1
    for (int t = 0; t < NT; ++t) {
      // this wave adds the pair's halves and stores; the two waves of a SIMD (w, w + 4 = channel quarters cq, cq + 2) take opposite
      // roles, so a SIMD always has one wave multiplying while the other converts and stores
      const bool fin = ((t + fc + (cq >> 1)) & 1) == kh;
      const int pl = t * 48 + n;
      constexpr int RD = 3;
      bf16x8 xb[RD][3];
      auto fetch = [&](int g, bf16x8 (&x)[3]) {
        const int tap = g >> 1;
        const int sl = pl + (tap / 3) * Wp + (tap % 3);
        const unsigned a0 = fb0 + (unsigned)(sl * 256 + ((((kh << 3) | kq) ^ (sl & 15)) << 4));
        const unsigned ak = (g & 1) ? (a0 ^ 64u) : a0;
#pragma unroll
        for (int mb = 0; mb < 3; ++mb) x[mb] = *(const bf16x8*)(lds + ak + mb * 4096);
      };
      fetch(0, xb[0]);
      fetch(1, xb[1]);
      f32x4 acc[3][2];
#pragma unroll
      for (int mb = 0; mb < 3; ++mb)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) acc[mb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
      bf16x8 rs[3];
#pragma unroll
      for (int g = 0; g < 18; ++g) {
        if (g + RD - 1 < 18) fetch(g + RD - 1, xb[(g + RD - 1) % RD]);
        if (RESID && g == 17 && fin) {
#pragma unroll
          for (int mb = 0; mb < 3; ++mb) {
            const int p = pl + mb * 16;
            rs[mb] = *(const bf16x8*)(resid + (f * (long)FP + (p < NV ? p : 0) + Wp + 1) * 128 + c0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mb = 0; mb < 3; ++mb)
#pragma unroll
          for (int nb = 0; nb < 2; ++nb) acc[mb][nb] = SVT_MFMA_16x16x32(wr[g >> 1][g & 1][nb], xb[g % RD][mb], acc[mb][nb]);
        __builtin_amdgcn_sched_barrier(0);
      }
      float* sc = scratch + ((cq * 2 + (t & 1)) * 6 * 64 + lane) * 4;
      if (!fin) {
#pragma unroll
        for (int mb = 0; mb < 3; ++mb)
#pragma unroll
          for (int nb = 0; nb < 2; ++nb) *(f32x4*)(sc + (mb * 2 + nb) * 256) = acc[mb][nb];
      }
      if (t == NT - 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this frame's requests (a frame old) in front of its last barrier
      // the parked accumulators must be IN the LDS before the partner wave (another SIMD) is let through: the raw barrier does not
      // wait for this wave's ds_writes by itself
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (fin) {
        const float4 b0 = *(const float4*)(bs + c0), b1 = *(const float4*)(bs + c0 + 4);
        const float4 s0 = *(const float4*)(bs + 128 + c0), s1 = *(const float4*)(bs + 128 + c0 + 4);
        const float bv[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w}, sv[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
#pragma unroll
        for (int mb = 0; mb < 3; ++mb) {
          const int p = pl + mb * 16;
          const int y = (int)(((float)p + 0.5f) * rWp);
          const f32x4 h0 = *(const f32x4*)(sc + (mb * 2) * 256), h1 = *(const f32x4*)(sc + (mb * 2 + 1) * 256);
          if (p < NV && p - y * Wp < Ws) {
            bf16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              float a = acc[mb][j >> 2][j & 3] + (j < 4 ? h0[j & 3] : h1[j & 3]) + bv[j];
              if (RESID) a += (float)rs[mb][j];
              o[j] = (bf16_t)(a > 0.f ? a : a * sv[j]);
            }
            *(bf16x8*)(out + (f * (long)FP + p + Wp + 1) * 128 + c0) = o;
          }
        }
      }
    }
  }
}

}  // namespace

int g_conv3x3_c64_form = 0;      // svt_debug_set key 25: C3_DBG bits << 4 (DIAG builds)
int g_conv3x3_c64 = 1;  // svt_debug_set key 23: 0 = stage 1 of the lip front-end on the GEMM kernels (A/B, tests)
int g_conv3x3_c64_launches = 0;  // svt_debug_set key 24 returns it (the tests' proof that this path ran: its results are bit-identical to the GEMM path's)

static size_t c3_lds_bytes(int Hs, int Ws) {
  const int FP = (Hs + 2) * (Ws + 2);
  return (size_t)2 * ((FP + 7) / 8 * 8 + C3_SLACK) * 128 + 512;   // two frames + bias / slope
}
// eligible geometry: two padded frames fit the LDS, and a frame has work for the four position groups
bool conv3x3_c64_ok(int prec, int Hs, int Ws) {
  if (!g_conv3x3_c64 || prec != 1 || Hs < 1 || Ws < 1) return false;
  return Hs * (Ws + 2) >= 12 * 16 && c3_lds_bytes(Hs, Ws) <= 160 * 1024;
}

// in / out / resid: [F][Hs+2][Ws+2][64] 16-bit, halos zero (out: written in its interior only); wimg: svt_video_finalize's fragment
// image of the 3x3 kernel with the BatchNorm scale folded; bias / slope: 64 floats
int launch_conv3x3_c64(const void* in, const void* wimg, const float* bias, const float* slope, const void* resid, void* out, long F,
                       int Hs, int Ws, hipStream_t s) {
  const size_t shm = c3_lds_bytes(Hs, Ws);
  static int ncu = 0;
  if (!ncu) {
    int dev = 0;
    hipDeviceProp_t pr;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&pr, dev) != hipSuccess) return 1;
    ncu = pr.multiProcessorCount;
  }
  const int grid = (int)(F < ncu ? F : ncu);
  ++g_conv3x3_c64_launches;
  auto go = [&](auto kern) -> int {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm) != hipSuccess) return 1;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), shm, s, (const bf16_t*)in, (const uint4*)wimg, bias, slope, (const bf16_t*)resid,
                       (bf16_t*)out, (int)F, Hs, Ws, g_conv3x3_c64_form >> 4);
    return 0;
  };
  if (resid ? go(conv3x3_c64_kernel<true>) : go(conv3x3_c64_kernel<false>)) return 1;
  return hipGetLastError() != hipSuccess;
}


static int c128_ring_slots(int Hs, int Ws) {
  const int Wp = Ws + 2, FP = (Hs + 2) * Wp, MB = (Hs * Wp + 15) / 16, NT = (MB + 2) / 3;
  const int need = NT * 48 + 2 * Wp + 2;   // the last triple's garbage positions read this far
  return ((FP > need ? FP : need) + 7) / 8 * 8;
}
static size_t c128_lds_bytes(int Hs, int Ws) { return (size_t)2 * c128_ring_slots(Hs, Ws) * 256 + 4 * 2 * 6 * 1024 + 1024; }
bool conv3x3_c128_ok(int prec, int Hs, int Ws) {
  if (!g_conv3x3_c64 || prec != 1 || Hs < 1 || Ws < 1) return false;
  return Hs * (Ws + 2) >= 48 && c128_lds_bytes(Hs, Ws) <= 160 * 1024;
}
// in / out / resid: [F][Hs+2][Ws+2][128] 16-bit, halos zero; wimg: fold_conv_frag128's image; bias / slope: 128 floats
int launch_conv3x3_c128(const void* in, const void* wimg, const float* bias, const float* slope, const void* resid, void* out, long F,
                        int Hs, int Ws, hipStream_t s) {
  const size_t shm = c128_lds_bytes(Hs, Ws);
  const unsigned fbytes = (unsigned)c128_ring_slots(Hs, Ws) * 256u;
  static int ncu = 0;
  if (!ncu) {
    int dev = 0;
    hipDeviceProp_t pr;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&pr, dev) != hipSuccess) return 1;
    ncu = pr.multiProcessorCount;
  }
  const int grid = (int)(F < ncu ? F : ncu);
  ++g_conv3x3_c64_launches;
  auto go = [&](auto kern) -> int {
    if (ensure_dyn_lds((const void*)kern, (int)shm)) return 1;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), shm, s, (const bf16_t*)in, (const uint4*)wimg, bias, slope, (const bf16_t*)resid,
                       (bf16_t*)out, (int)F, Hs, Ws, fbytes);
    return 0;
  };
  if (resid ? go(conv3x3_c128_kernel<true>) : go(conv3x3_c128_kernel<false>)) return 1;
  return hipGetLastError() != hipSuccess;
}

}  // namespace svt

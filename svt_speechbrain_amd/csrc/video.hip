// Lip-ROI front-end of the AV-HuBERT video branch (reference N20EMv2/video_only/resnet.py):
// 3-D stem conv (1 -> 64, k = 5x7x7, stride 1x2x2) + BN + PReLU, 3x3/2 max-pool, and the pooling / layout kernels
// around the ResNet-18 trunk.  The trunk's 3x3 and 1x1 convolutions are implicit GEMMs over zero-haloed
// channels-last tensors and run on the dense-contraction kernel (gemm.hip, generalised row addressing); eval-mode
// batch norms are folded into the conv weights / biases on the host (svt_video_finalize).
//
// Stem on MFMA: one workgroup per output frame.  The five input frames it needs (zero-haloed, 90 KB for an 88x88
// ROI) are staged once in LDS; K = (dt, dy) x 8 x-taps = 35 chunks of 8 (the 7 x-taps are widened to an aligned
// window of 8 starting at 2x-4, tap 0 has weight 0), padded to 40 chunks = ten 32-deep MFMA k-steps.  A lane's B
// operand (8 consecutive pixels of one (dt, dy) row) is four aligned ds_read_b32; the whole 64 x 320 weight matrix
// lives in registers as MFMA A fragments (160 VGPRs), so LDS carries only the pixel reads.
#include "common.h"

namespace svt {
namespace {

typedef unsigned u32x4v __attribute__((ext_vector_type(4)));

template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return (bf16_t)v; }

// (B,1,T,H,W) f32 -> [B][T+4][Hp][Wp] operand type, pixel (t,y,x) at (t+2, y+3, x+4), zeros elsewhere
template <typename T>
__global__ void video_pad_kernel(const float* v, int B, int Tt, int H, int W, int Hp, int Wp, T* out) {
  const long n = (long)B * (Tt + 4) * Hp * Wp;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int xp = (int)(i % Wp);
    long r = i / Wp;
    const int yp = (int)(r % Hp);
    r /= Hp;
    const int tp = (int)(r % (Tt + 4));
    const int b = (int)(r / (Tt + 4));
    const int x = xp - 4, y = yp - 3, t = tp - 2;
    float val = 0.f;
    if (x >= 0 && x < W && y >= 0 && y < H && t >= 0 && t < Tt) val = v[(((long)b * Tt + t) * H + y) * W + x];
    out[i] = from_f32<T>(val);
  }
}

// 16-bit modes, W % 4 == 0: 8 output pixels (16 bytes) per thread from two aligned float4 (Wp is a multiple of 8, pixel x sits at
// x + 4, so an 8-pixel group starts at x = 8 g - 4: both halves are whole float4 inside or outside the row)
__global__ void video_pad8_kernel(const float* v, int B, int Tt, int H, int W, int Hp, int Wp, bf16_t* out) {
  const int Wg = Wp >> 3;
  const long n = (long)B * (Tt + 4) * Hp * Wg;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int g = (int)(i % Wg);
    long r = i / Wg;
    const int yp = (int)(r % Hp);
    r /= Hp;
    const int tp = (int)(r % (Tt + 4));
    const int b = (int)(r / (Tt + 4));
    const int x0 = g * 8 - 4, y = yp - 3, t = tp - 2;
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (bf16_t)0.f;
    if (y >= 0 && y < H && t >= 0 && t < Tt) {
      const float* row = v + (((long)b * Tt + t) * H + y) * W;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int x = x0 + 4 * h;
        if (x >= 0 && x + 3 < W) {
          const float4 q = *(const float4*)(row + x);
          o[4 * h] = (bf16_t)q.x; o[4 * h + 1] = (bf16_t)q.y; o[4 * h + 2] = (bf16_t)q.z; o[4 * h + 3] = (bf16_t)q.w;
        }
      }
    }
    *(bf16x8*)(out + i * 8) = o;
  }
}

// ---- the recipe's input side (round 6): uint8 lip ROI -> cropped, normalised, padded operand in ONE pass ----
// Replaces transform_eval of N20EMv2/video_only/train_video_ssl.py:445-457 -- Normalize(0, 255) -> CenterCrop((88, 88)) ->
// Normalize(0.421, 0.165) on the np.load'ed uint8 frames, then .astype(np.float32) (:530-533) -- in front of video_pad*_kernel.  numpy
// evaluates ((u - sub0) / div0 - mean) / std in float64 and rounds ONCE to float32; a uint8 pixel has 256 values, so every workgroup
// builds the 256-entry table in LDS with the same four IEEE float64 operations (bit-identical to numpy; tests/golden/video_u8.pt) and
// the pass reads one byte per pixel instead of four.
__device__ __forceinline__ void build_u8_table(float* lut, VideoTransform tf) {
  for (int i = threadIdx.x; i < 256; i += blockDim.x) lut[i] = (float)((((double)i - tf.sub0) / tf.div0 - tf.mean) / tf.std);
  __syncthreads();
}
// roi (B,T,Hin,Win) uint8; output pixel (y, x) of the H x W crop = roi pixel (y + dy, x + dx)
template <typename T>
__global__ __launch_bounds__(256) void video_pad_u8_kernel(const unsigned char* v, int B, int Tt, int Hin, int Win, int dy, int dx, int H, int W,
                                                          int Hp, int Wp, VideoTransform tf, T* out) {
  __shared__ float lut[256];
  build_u8_table(lut, tf);
  const long n = (long)B * (Tt + 4) * Hp * Wp;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int xp = (int)(i % Wp);
    long r = i / Wp;
    const int yp = (int)(r % Hp);
    r /= Hp;
    const int tp = (int)(r % (Tt + 4));
    const int b = (int)(r / (Tt + 4));
    const int x = xp - 4, y = yp - 3, t = tp - 2;
    float val = 0.f;
    if (x >= 0 && x < W && y >= 0 && y < H && t >= 0 && t < Tt) val = lut[v[(((long)b * Tt + t) * Hin + y + dy) * Win + x + dx]];
    out[i] = from_f32<T>(val);
  }
}
// 16-bit modes, Wp % 8 == 0: 8 output pixels (one 16-byte store) per thread
__global__ __launch_bounds__(256) void video_pad8_u8_kernel(const unsigned char* v, int B, int Tt, int Hin, int Win, int dy, int dx, int H, int W,
                                                           int Hp, int Wp, VideoTransform tf, bf16_t* out) {
  __shared__ float lut[256];
  build_u8_table(lut, tf);
  const int Wg = Wp >> 3;
  const long n = (long)B * (Tt + 4) * Hp * Wg;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int g = (int)(i % Wg);
    long r = i / Wg;
    const int yp = (int)(r % Hp);
    r /= Hp;
    const int tp = (int)(r % (Tt + 4));
    const int b = (int)(r / (Tt + 4));
    const int x0 = g * 8 - 4, y = yp - 3, t = tp - 2;
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (bf16_t)0.f;
    if (y >= 0 && y < H && t >= 0 && t < Tt) {
      const unsigned char* row = v + (((long)b * Tt + t) * Hin + y + dy) * Win + dx;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int x = x0 + j;
        if (x >= 0 && x < W) o[j] = (bf16_t)lut[row[x]];
      }
    }
    *(bf16x8*)(out + i * 8) = o;
  }
}

// exact-fp32 stem (parity mode): one thread per (pixel, channel); w is [35*8][64] (k-major), BN scale folded
__global__ __launch_bounds__(256) void conv3d_front_f32_kernel(const float* vp, const float* w, const float* bias,
                                                               const float* slope, int Tt, int Hp, int Wp, int H0, int W0,
                                                               long npix_total, float* out) {
  const long p = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int c = threadIdx.x & 63;
  if (p >= npix_total) return;
  const int x = (int)(p % W0);
  long r = p / W0;
  const int y = (int)(r % H0);
  const long f = r / H0;
  const int b = (int)(f / Tt), t = (int)(f % Tt);
  const float* base = vp + (((long)b * (Tt + 4) + t) * Hp + 2 * y) * Wp + 2 * x;
  float acc = 0.f;
  for (int dt = 0; dt < 5; ++dt)
    for (int dy = 0; dy < 7; ++dy) {
      const float* row = base + ((long)dt * Hp + dy) * Wp;
      const float* wr = w + (long)((dt * 7 + dy) * 8) * 64 + c;
#pragma unroll
      for (int j = 1; j < 8; ++j) acc = fmaf(row[j], wr[j * 64], acc);
    }
  float v = acc + bias[c];
  v = v > 0.f ? v : v * slope[c];
  out[p * 64 + c] = v;
}

// bf16 MFMA stem: see the header comment.  wfrag: [10 k-steps][4 channel blocks][64 lanes] x 16 bytes, A-operand
// fragments with the row permutation that leaves a lane with 16 consecutive channels.
__global__ __launch_bounds__(512) void conv3d_front_bf16_kernel(const bf16_t* vp, const uint4* wfrag, const float* bias,
                                                                const float* slope, int Tt, int Hp, int Wp, int H0, int W0,
                                                                bf16_t* out) {
  extern __shared__ __attribute__((aligned(16))) uint4 lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long f = blockIdx.x;
  const int b = (int)(f / Tt), t = (int)(f % Tt);
  const int plane = Hp * Wp;  // multiple of 8
  {
    const uint4* src = (const uint4*)(vp + ((long)b * (Tt + 4) + t) * plane);
    const int n16 = 5 * plane / 8;
    for (int i = tid; i < n16; i += 512) lds[i] = src[i];
  }
  bf16x8 wr[10][4];
#pragma unroll
  for (int ks = 0; ks < 10; ++ks)
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) wr[ks][nb] = __builtin_bit_cast(bf16x8, wfrag[(ks * 4 + nb) * 64 + lane]);
  int koff[10];
#pragma unroll
  for (int ks = 0; ks < 10; ++ks) {
    int s = ks * 4 + (lane >> 4);
    if (s > 34) s = 34;  // chunks 35..39 are padding: their weights are zero, any valid address will do
    koff[ks] = ((s / 7) * Hp + (s % 7)) * Wp;
  }
  float bv[16], sv[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) { bv[j] = bias[(lane >> 4) * 16 + j]; sv[j] = slope[(lane >> 4) * 16 + j]; }
  __syncthreads();
  const int npix = H0 * W0, nblk = (npix + 15) / 16;
  const bf16_t* img = (const bf16_t*)lds;
  for (int blk = wave; blk < nblk; blk += 8) {
    int pix = blk * 16 + (lane & 15);
    const bool live = pix < npix;
    if (!live) pix = npix - 1;
    const int yy = pix / W0, x = pix - yy * W0;
    const int pbase = 2 * yy * Wp + 2 * x;
    f32x4 acc[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) acc[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 10; ++ks) {
      const unsigned* q = (const unsigned*)(img + koff[ks] + pbase);
      const u32x4v xv = {q[0], q[1], q[2], q[3]};
#pragma unroll
      for (int nb = 0; nb < 4; ++nb)
        acc[nb] = SVT_MFMA_16x16x32(wr[ks][nb], __builtin_bit_cast(bf16x8, xv), acc[nb]);
    }
    if (live) {
      bf16_t* o = out + ((long)f * npix + pix) * 64 + (lane >> 4) * 16;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        bf16x8 ov;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int jj = h * 8 + j;
          float v = acc[jj >> 2][jj & 3] + bv[jj];
          v = v > 0.f ? v : v * sv[jj];
          ov[j] = (bf16_t)v;
        }
        *(bf16x8*)(o + h * 8) = ov;
      }
    }
  }
}

// The same stem FUSED with the 3x3 / 2 max-pool behind it and made persistent (round 4): the unfused pair writes the 44 x 44 x 64
// stem output (2 GB per 8 000 frames) and reads it back to keep a quarter of it.  One workgroup per CU takes a run of consecutive frames:
//   planes  six plane slots in LDS, slot = padded time index mod 6: consecutive frames of a clip share four of their five planes, so a
//           frame fetches ONE new plane (LDS-DMA, requested a frame ahead) instead of five; weights and bias / slope stay in registers
//           for the whole run (the one-frame kernel reloads 10 KB of fragments per wave and frame);
//   bands   a frame is computed in bands of 8 stem rows = 4 pool rows; the band's post-PReLU outputs go to a 9-row circular LDS buffer
//           (16-byte chunks XOR-swizzled by the pixel's x), row slot = row mod 9 keeps the row above the band for the pool window;
//   pool    after the band's barrier the workgroup max-pools 4 x W1 x 8 chunk items straight out of LDS into the zero-haloed stage-1
//           input; the values pooled are the bf16-rounded outputs the unfused pair would have stored: bit-identical results.
// timing ablations (make DIAG=1; svt_debug_set key 26 value 1 + 16 x bits: 1 = no pool phase, 2 = no MFMA phase, 4 = no plane
// requests, 8 = no stem epilogue; tools/c3_ablate.sh with KEY=26): compiled out of the shipped library
#ifdef SVT_DIAG
#define STEM_DBG(bit) (dbg & (bit))
#else
#define STEM_DBG(bit) 0
#endif
typedef short s16x8v __attribute__((ext_vector_type(8)));
// bf16 / IEEE-half bits <-> int16 with the same ordering as the values (an involution: negative values get their magnitude bits flipped)
// (on whole dwords: s - (s >> 15) turns each half's sign bit into 0x7FFF without a borrow between the halves; the vector-of-short form
// compiles to per-element 16-bit operations, 41 instructions per 16 bytes instead of 16)
__device__ __forceinline__ s16x8v ordered16(s16x8v u) {
  u32x4v w = __builtin_bit_cast(u32x4v, u);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const unsigned sg = w[i] & 0x80008000u;
    w[i] ^= sg - (sg >> 15);
  }
  return __builtin_bit_cast(s16x8v, w);
}
__device__ __forceinline__ void stem_dma16(const void* gsrc, unsigned lds_byte_addr) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(lds_byte_addr) : "memory");
}
__global__ __launch_bounds__(512) void conv3d_front_pool_kernel(const bf16_t* __restrict__ vp, const uint4* __restrict__ wfrag,
                                                                const float* __restrict__ bias, const float* __restrict__ slope, int Tt,
                                                                int Hp, int Wp, int H0, int W0, int H1, int W1, long F, int fpb,
                                                                bf16_t* __restrict__ out, int dbg) {
  extern __shared__ __attribute__((aligned(1024))) char lds_raw[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int plane = Hp * Wp;                            // elements, multiple of 8
  const unsigned SB = (unsigned)((plane * 2 + 1023) / 1024) * 1024u;   // bytes per plane slot
  const unsigned lds0 = (unsigned)(size_t)lds_raw;
  char* band = lds_raw + 6 * SB;
  const long f0 = (long)blockIdx.x * fpb, f1 = f0 + fpb < F ? f0 + fpb : F;

  // nine of the image's ten k-steps: chunks 36..39 (k-step 9) are all padding with zero weights
  bf16x8 wr[9][4];
#pragma unroll
  for (int ks = 0; ks < 9; ++ks)
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) wr[ks][nb] = __builtin_bit_cast(bf16x8, wfrag[(ks * 4 + nb) * 64 + lane]);
  // bias / slope behind the band (registers are for the weights and the pool's nine reads)
  float* bs = (float*)(band + 9 * W0 * 128);
  if (tid < 64) { bs[tid] = bias[tid]; bs[64 + tid] = slope[tid]; }

  // padded plane tp of clip b -> slot tp % 6 (whole KiB pieces; the tail piece is clamped to the plane's last 16 bytes)
  auto request_plane = [&](int b, int tp) {
    const char* src = (const char*)(vp + ((long)b * (Tt + 4) + tp) * plane);
    const unsigned dst = lds0 + (unsigned)(tp % 6) * SB;
    for (int i = wave; i < (int)(SB >> 10); i += 8) {
      int off = i * 1024 + lane * 16;
      off = off < plane * 2 - 16 ? off : plane * 2 - 16;
      stem_dma16(src + off, dst + (unsigned)i * 1024u);
    }
  };
  const int nbands = (H1 + 3) >> 2;
  const float rW0 = 1.0f / (float)W0;
  int cur_b = -1;
  for (long f = f0; f < f1; ++f) {
    const int b = (int)(f / Tt), t = (int)(f - (long)b * Tt);
    if (b != cur_b) {   // first frame of the run, or a new clip: all five planes
      for (int dt = 0; dt < 5; ++dt) request_plane(b, t + dt);
      cur_b = b;
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // lgkmcnt: the bs stores of the kernel's head
      __builtin_amdgcn_s_barrier();
    }
    if (f + 1 < f1 && t + 1 < Tt && !STEM_DBG(4)) request_plane(b, t + 5);   // the next frame's new plane into the slot plane t - 1 left
    int koff[9];
#pragma unroll
    for (int ks = 0; ks < 9; ++ks) {
      int s = ks * 4 + (lane >> 4);
      if (s > 34) s = 34;  // chunks 35..39 are padding: their weights are zero, any valid address will do
      const int dt = s / 7, dy = s - dt * 7;
      koff[ks] = (int)(((unsigned)((t + dt) % 6) * SB) >> 1) + dy * Wp;
    }
    const bf16_t* img = (const bf16_t*)lds_raw;
    for (int bi = 0; bi < nbands; ++bi) {
      const int r0 = bi * 8, nrows = H0 - r0 < 8 ? H0 - r0 : 8;
      const int npix = nrows * W0, nblk = (npix + 15) >> 4;
      for (int blk = wave; blk < nblk && !STEM_DBG(2); blk += 8) {
        int pix = blk * 16 + (lane & 15);
        const bool live = pix < npix;
        if (!live) pix = npix - 1;
        const int ry = (int)(((float)pix + 0.5f) * rW0), x = pix - ry * W0, yy = r0 + ry;   // pix < 8 W0: exact
        const int pbase = 2 * yy * Wp + 2 * x;
        f32x4 acc[4];
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) acc[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 9; ++ks) {
          const unsigned* q = (const unsigned*)(img + koff[ks] + pbase);
          const u32x4v xv = {q[0], q[1], q[2], q[3]};
#pragma unroll
          for (int nb = 0; nb < 4; ++nb) acc[nb] = SVT_MFMA_16x16x32(wr[ks][nb], __builtin_bit_cast(bf16x8, xv), acc[nb]);
        }
        if (live && !STEM_DBG(8)) {
          char* o = band + ((yy % 9) * W0 + x) * 128;
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const float* bq = bs + (lane >> 4) * 16 + h * 8;
            const float4 b0 = *(const float4*)bq, b1 = *(const float4*)(bq + 4), s0 = *(const float4*)(bq + 64), s1 = *(const float4*)(bq + 68);
            const float bv[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w}, sv[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
            bf16x8 ov;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              const int jj = h * 8 + j;
              float v = acc[jj >> 2][jj & 3] + bv[j];
              v = v > 0.f ? v : v * sv[j];
              ov[j] = (bf16_t)v;
            }
            *(s16x8v*)(o + ((((lane >> 4) * 2 + h) ^ (x & 7)) << 4)) = ordered16(__builtin_bit_cast(s16x8v, ov));
          }
        }
      }
      // the requests of this frame are waited for once, in front of the last band's barrier (a frame old by then)
      if (bi == nbands - 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      // the band's ds_writes (and, on the first pass, the bs stores) land before any wave pools them: a raw s_barrier does not
      // wait for a wave's own LDS stores on gfx950
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      // pool: slot = (pool row of the band, x1 padded to 32, chunk) -- shifts only; window coordinates clamped instead of skipped (a
      // duplicate does not change a maximum), so the nine reads go out together; the maximum is taken on the order-preserving int16
      // image of the values the stem wrote (v_pk_max_i16: 4 instructions per 16 bytes instead of 8 conversions + 8 maxima)
      const int py0 = bi * 4, nprow = H1 - py0 < 4 ? H1 - py0 : 4;
#pragma unroll 1
      for (int i = tid; i < 1024 && !STEM_DBG(1); i += 512) {
        const int c = i & 7, x1 = (i >> 3) & 31, pr = i >> 8, y1 = py0 + pr;
        if (x1 < W1 && pr < nprow) {
          int ro[3], xo[3];
#pragma unroll
          for (int d = 0; d < 3; ++d) {
            int y = 2 * y1 + d - 1, x = 2 * x1 + d - 1;
            y = y < 0 ? 0 : (y >= H0 ? H0 - 1 : y);
            x = x < 0 ? 0 : (x >= W0 ? W0 - 1 : x);
            ro[d] = (y % 9) * W0 * 128;
            xo[d] = x * 128 + ((c ^ (x & 7)) << 4);
          }
          s16x8v v[9];
#pragma unroll
          for (int k = 0; k < 9; ++k) v[k] = *(const s16x8v*)(band + ro[k / 3] + xo[k % 3]);
          s16x8v m = v[0];
#pragma unroll
          for (int k = 1; k < 9; ++k) m = __builtin_elementwise_max(m, v[k]);
          *(s16x8v*)(out + ((f * (H1 + 2) + y1 + 1) * (long)(W1 + 2) + x1 + 1) * 64 + c * 8) = ordered16(m);
        }
      }
      __builtin_amdgcn_s_barrier();
    }
  }
}

// 3x3 stride-2 pad-1 max-pool over [F][H0][W0][C] -> interior of the zero-haloed [F][H1+2][W1+2][C]
template <typename T>
__global__ void maxpool_3x3s2_kernel(const T* in, long F, int H0, int W0, int C, int H1, int W1, T* out) {
  const long n = F * H1 * W1 * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    long r = i / C;
    const int x1 = (int)(r % W1);
    r /= W1;
    const int y1 = (int)(r % H1);
    const long f = r / H1;
    float m = -3.4e38f;
    for (int dy = 0; dy < 3; ++dy) {
      const int y = 2 * y1 - 1 + dy;
      if (y < 0 || y >= H0) continue;
      for (int dx = 0; dx < 3; ++dx) {
        const int x = 2 * x1 - 1 + dx;
        if (x < 0 || x >= W0) continue;
        m = fmaxf(m, (float)in[((f * H0 + y) * W0 + x) * C + c]);
      }
    }
    out[((f * (H1 + 2) + y1 + 1) * (W1 + 2) + x1 + 1) * C + c] = from_f32<T>(m);
  }
}

// bf16: 8 channels (16 bytes) per thread
__global__ void maxpool_3x3s2_bf16x8_kernel(const bf16_t* in, long F, int H0, int W0, int C, int H1, int W1, bf16_t* out) {
  const int C8 = C / 8;
  const long n = F * H1 * W1 * C8;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C8) * 8;
    long r = i / C8;
    const int x1 = (int)(r % W1);
    r /= W1;
    const int y1 = (int)(r % H1);
    const long f = r / H1;
    float m[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) m[j] = -3.4e38f;
    for (int dy = 0; dy < 3; ++dy) {
      const int y = 2 * y1 - 1 + dy;
      if (y < 0 || y >= H0) continue;
      for (int dx = 0; dx < 3; ++dx) {
        const int x = 2 * x1 - 1 + dx;
        if (x < 0 || x >= W0) continue;
        const bf16x8 v = *(const bf16x8*)(in + ((f * H0 + y) * W0 + x) * C + c);
#pragma unroll
        for (int j = 0; j < 8; ++j) m[j] = fmaxf(m[j], (float)v[j]);
      }
    }
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (bf16_t)m[j];
    *(bf16x8*)(out + ((f * (H1 + 2) + y1 + 1) * (W1 + 2) + x1 + 1) * C + c) = o;
  }
}

// zero the one-pixel halo of [F][Hp][Wp][C] (16-byte stores; C * sizeof(T) is a multiple of 16)
__global__ void zero_halo_kernel(uint4* buf, long F, int Hp, int Wp, int c16) {
  const int nh = 2 * Wp + 2 * (Hp - 2);
  const long n = F * nh * c16;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % c16);
    long r = i / c16;
    const int hidx = (int)(r % nh);
    const long f = r / nh;
    int y, x;
    if (hidx < Wp) { y = 0; x = hidx; }
    else if (hidx < 2 * Wp) { y = Hp - 1; x = hidx - Wp; }
    else { const int k = hidx - 2 * Wp; y = 1 + (k >> 1); x = (k & 1) ? Wp - 1 : 0; }
    buf[((f * Hp + y) * Wp + x) * c16 + c] = uint4{0, 0, 0, 0};
  }
}

// mean over the H x W interior of [F][H+2][W+2][C] -> [F][C] (operand type; the projection GEMM reads it)
template <typename T>
__global__ void avgpool_interior_kernel(const T* in, long F, int H, int W, int C, T* out) {
  const long n = F * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const long f = i / C;
    float s = 0.f;
    for (int y = 0; y < H; ++y)
      for (int x = 0; x < W; ++x) s += (float)in[((f * (H + 2) + y + 1) * (W + 2) + x + 1) * C + c];
    out[i] = from_f32<T>(s / (float)(H * W));
  }
}

unsigned grid_of(long n) {
  const long g = (n + 255) / 256;
  return (unsigned)(g < 1 ? 1 : (g > 65536 * 4 ? 65536 * 4 : g));
}

}  // namespace

int launch_video_pad(int prec, const float* v, int B, int T, int H, int W, int Hp, int Wp, void* out, hipStream_t s) {
  const long n = (long)B * (T + 4) * Hp * Wp;
  if (prec && W % 4 == 0 && Wp % 8 == 0 && ((size_t)v & 15) == 0)
    hipLaunchKernelGGL(video_pad8_kernel, dim3(grid_of(n / 8)), dim3(256), 0, s, v, B, T, H, W, Hp, Wp, (bf16_t*)out);
  else if (prec) hipLaunchKernelGGL(video_pad_kernel<bf16_t>, dim3(grid_of(n)), dim3(256), 0, s, v, B, T, H, W, Hp, Wp, (bf16_t*)out);
  else hipLaunchKernelGGL(video_pad_kernel<float>, dim3(grid_of(n)), dim3(256), 0, s, v, B, T, H, W, Hp, Wp, (float*)out);
  SVT_LAUNCH_CHECK();
  return 0;
}

int launch_video_pad_u8(int prec, const unsigned char* v, int B, int T, int Hin, int Win, int dy, int dx, int H, int W, int Hp, int Wp,
                        const VideoTransform& tf, void* out, hipStream_t s) {
  const long n = (long)B * (T + 4) * Hp * Wp;
  if (prec && Wp % 8 == 0)
    hipLaunchKernelGGL(video_pad8_u8_kernel, dim3(grid_of(n / 8)), dim3(256), 0, s, v, B, T, Hin, Win, dy, dx, H, W, Hp, Wp, tf, (bf16_t*)out);
  else if (prec) hipLaunchKernelGGL(video_pad_u8_kernel<bf16_t>, dim3(grid_of(n)), dim3(256), 0, s, v, B, T, Hin, Win, dy, dx, H, W, Hp, Wp, tf, (bf16_t*)out);
  else hipLaunchKernelGGL(video_pad_u8_kernel<float>, dim3(grid_of(n)), dim3(256), 0, s, v, B, T, Hin, Win, dy, dx, H, W, Hp, Wp, tf, (float*)out);
  SVT_LAUNCH_CHECK();
  return 0;
}

int launch_conv3d_front(int prec, const void* vp, const void* w, const float* bias, const float* slope, long F, int T, int Hp,
                        int Wp, int H0, int W0, void* out, hipStream_t s) {
  if (prec) {
    const size_t lds_bytes = (size_t)5 * Hp * Wp * 2;
    if (lds_bytes > 160 * 1024) { set_error("video front-end: the lip ROI is too large for the LDS-resident stem (5 frames must fit 160 KB)"); return -1; }
    if (int r_ = ensure_dyn_lds((const void*)conv3d_front_bf16_kernel, (int)lds_bytes)) return r_;
    prof_begin(s);
    hipLaunchKernelGGL(conv3d_front_bf16_kernel, dim3((unsigned)F), dim3(512), lds_bytes, s, (const bf16_t*)vp, (const uint4*)w, bias,
                       slope, T, Hp, Wp, H0, W0, (bf16_t*)out);
    prof_end(s, 2.0 * F * H0 * W0 * 64.0 * 245.0, 0.0, 1);
  } else {
    const long npix = F * H0 * W0;
    hipLaunchKernelGGL(conv3d_front_f32_kernel, dim3((unsigned)((npix + 3) / 4)), dim3(256), 0, s, (const float*)vp, (const float*)w,
                       bias, slope, T, Hp, Wp, H0, W0, npix, (float*)out);
  }
  SVT_LAUNCH_CHECK();
  return 0;
}

int g_stem_pool_fused = 1;  // svt_debug_set key 26: 0 = stem and max-pool as two kernels (A/B, tests)
static size_t stem_pool_lds(int Hp, int Wp, int W0) { return (size_t)6 * (((size_t)Hp * Wp * 2 + 1023) / 1024 * 1024) + (size_t)9 * W0 * 128 + 512; }
// (the pool phase indexes its items as (4 rows, 32 columns, 8 chunks): pooled width (W0 + 1) / 2 <= 32)
bool conv3d_front_pool_ok(int prec, int Hp, int Wp, int W0) {
  return (g_stem_pool_fused & 1) && prec == 1 && W0 <= 64 && stem_pool_lds(Hp, Wp, W0) <= 160 * 1024;
}
// stem + 3x3/2 max-pool -> interior of the zero-haloed [F][H1+2][W1+2][64] (16-bit storage modes)
int launch_conv3d_front_pool(const void* vp, const void* w, const float* bias, const float* slope, long F, int T, int Hp, int Wp, int H0,
                             int W0, int H1, int W1, void* out, hipStream_t s) {
  const size_t lds_bytes = stem_pool_lds(Hp, Wp, W0);
  if (int r_ = ensure_dyn_lds((const void*)conv3d_front_pool_kernel, (int)lds_bytes)) return r_;
  static int ncu = 0;
  if (!ncu) {
    int dev = 0;
    hipDeviceProp_t pr;
    SVT_HIP(hipGetDevice(&dev));
    SVT_HIP(hipGetDeviceProperties(&pr, dev));
    ncu = pr.multiProcessorCount;
  }
  const int fpb = (int)((F + ncu - 1) / ncu);
  const unsigned grid = (unsigned)((F + fpb - 1) / fpb);
  prof_begin(s);
  hipLaunchKernelGGL(conv3d_front_pool_kernel, dim3(grid), dim3(512), lds_bytes, s, (const bf16_t*)vp, (const uint4*)w, bias, slope, T, Hp,
                     Wp, H0, W0, H1, W1, F, fpb, (bf16_t*)out, g_stem_pool_fused >> 4);
  prof_end(s, 2.0 * F * H0 * W0 * 64.0 * 245.0, 0.0, 1);
  SVT_LAUNCH_CHECK();
  return 0;
}

int launch_maxpool_3x3s2(int prec, const void* in, long F, int H0, int W0, int C, int H1, int W1, void* out, hipStream_t s) {
  const long n = F * H1 * W1 * C;
  if (prec && C % 8 == 0) hipLaunchKernelGGL(maxpool_3x3s2_bf16x8_kernel, dim3(grid_of(n / 8)), dim3(256), 0, s, (const bf16_t*)in, F, H0, W0, C, H1, W1, (bf16_t*)out);
  else if (prec) hipLaunchKernelGGL(maxpool_3x3s2_kernel<bf16_t>, dim3(grid_of(n)), dim3(256), 0, s, (const bf16_t*)in, F, H0, W0, C, H1, W1, (bf16_t*)out);
  else hipLaunchKernelGGL(maxpool_3x3s2_kernel<float>, dim3(grid_of(n)), dim3(256), 0, s, (const float*)in, F, H0, W0, C, H1, W1, (float*)out);
  SVT_LAUNCH_CHECK();
  return 0;
}

int launch_zero_halo(int prec, void* buf, long F, int Hp, int Wp, int C, hipStream_t s) {
  const int c16 = C * (prec ? 2 : 4) / 16;
  const long n = F * (2 * Wp + 2 * (Hp - 2)) * c16;
  hipLaunchKernelGGL(zero_halo_kernel, dim3(grid_of(n)), dim3(256), 0, s, (uint4*)buf, F, Hp, Wp, c16);
  SVT_LAUNCH_CHECK();
  return 0;
}

int launch_avgpool_interior(int prec, const void* in, long F, int H, int W, int C, void* out, hipStream_t s) {
  const long n = F * C;
  if (prec) hipLaunchKernelGGL(avgpool_interior_kernel<bf16_t>, dim3(grid_of(n)), dim3(256), 0, s, (const bf16_t*)in, F, H, W, C, (bf16_t*)out);
  else hipLaunchKernelGGL(avgpool_interior_kernel<float>, dim3(grid_of(n)), dim3(256), 0, s, (const float*)in, F, H, W, C, (float*)out);
  SVT_LAUNCH_CHECK();
  return 0;
}

}  // namespace svt

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
  if (prec) hipLaunchKernelGGL(video_pad_kernel<bf16_t>, dim3(grid_of(n)), dim3(256), 0, s, v, B, T, H, W, Hp, Wp, (bf16_t*)out);
  else hipLaunchKernelGGL(video_pad_kernel<float>, dim3(grid_of(n)), dim3(256), 0, s, v, B, T, H, W, Hp, Wp, (float*)out);
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

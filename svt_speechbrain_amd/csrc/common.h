// Shared declarations for the MI355X (gfx950) singing-transcription kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>

// The 16-bit operand type of the throughput mode (precision code 1).  The library is built twice from the same sources:
// libsvt_mi355.so with bf16 operands (`bf16_t` = __bf16: the mode BASELINE names) and libsvt_mi355_f16.so with IEEE half operands
// (-DSVT_OPERAND_F16: same MFMA rate, three more mantissa bits; the Python layer loads it for precision="fp16").  Everything in
// the kernels is written against `bf16_t` and SVT_MFMA_*; the split-operand engines (precision codes 2 / 3) name their piece types
// explicitly (`real_bf16x8`, `_Float16`) and exist in the bf16 build only.
typedef __bf16 real_bf16x8 __attribute__((ext_vector_type(8)));
#ifdef SVT_OPERAND_F16
typedef _Float16 bf16_t;
#define SVT_MFMA_16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0)
#define SVT_MFMA_32x32x16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)
#else
typedef __bf16 bf16_t;
#define SVT_MFMA_16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)
#define SVT_MFMA_32x32x16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)
#endif
typedef bf16_t bf16x8 __attribute__((ext_vector_type(8)));
typedef bf16_t bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace svt {

#if defined(__HIPCC__)
// Exact-erf GELU, erf by Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7: fp32 rounding level, far inside the
// 1e-3 parity budget): one v_rcp + one v_exp + 8 FMAs instead of libm erff's ~40 instructions.  The conv and
// FFN epilogues evaluate it ~1.6e9 times per step.
__device__ __forceinline__ float gelu_fast(float x) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
  float poly = fmaf(1.061405429f, t, -1.453152027f);
  poly = fmaf(poly, t, 1.421413741f);
  poly = fmaf(poly, t, -0.284496736f);
  poly = fmaf(poly, t, 0.254829592f);
  const float e = __builtin_amdgcn_exp2f(-1.44269504088896340736f * z * z);
  const float erf_abs = fmaf(-poly * t, e, 1.0f);
  const float half_x = 0.5f * x;
  return fmaf(half_x, copysignf(erf_abs, x), half_x);
}
// the same function on a pair of values with packed fp32 math (v_pk_fma_f32 / v_pk_mul_f32: two lanes-worth of
// FMA per issue slot; rcp/exp stay scalar): ~10 issue slots per element instead of ~17
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2_t gelu_fast2(f32x2_t x) {
  const f32x2_t z = __builtin_elementwise_abs(x) * 0.70710678118654752440f;
  const f32x2_t d = z * 0.3275911f + 1.0f;
  const f32x2_t t = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
  f32x2_t poly = t * 1.061405429f + (-1.453152027f);
  poly = poly * t + 1.421413741f;
  poly = poly * t + (-0.284496736f);
  poly = poly * t + 0.254829592f;
  const f32x2_t zz = z * z * (-1.44269504088896340736f);
  const f32x2_t e = {__builtin_amdgcn_exp2f(zz.x), __builtin_amdgcn_exp2f(zz.y)};
  const f32x2_t erf_abs = 1.0f - poly * t * e;
  const f32x2_t hx = x * 0.5f;
  const f32x2_t sg = {copysignf(erf_abs.x, x.x), copysignf(erf_abs.y, x.y)};
  return hx * sg + hx;
}
#ifdef SVT_OPERAND_F16
// IEEE-half build (precision="fp16", 2^-12 relative rounding of the stored value): the longer polynomial -- erf by an
// odd degree-17 polynomial on |z| <= 3 (z = x / sqrt 2), saturated outside -- no transcendental issue slots (rcp / exp
// are quarter rate), 17 issue slots per PAIR instead of ~33.  |error| <= 1.5e-4 absolute, <= 4e-5 relative for
// x > 0.01: 50x below the bf16 rounding of the stored value (2^-9 relative).  fp32 outputs keep gelu_fast.
__device__ __forceinline__ f32x2_t gelu_bf16x2(f32x2_t x) {
  f32x2_t z = x * 0.70710678118654752440f;
  z.x = __builtin_amdgcn_fmed3f(z.x, -3.0f, 3.0f);
  z.y = __builtin_amdgcn_fmed3f(z.y, -3.0f, 3.0f);
  const f32x2_t u = z * z;
  f32x2_t q = u * 4.074456683e-08f + (-1.944940095e-06f);
  q = q * u + 4.106299457e-05f;
  q = q * u + (-5.110675702e-04f);
  q = q * u + 4.235681612e-03f;
  q = q * u + (-2.510436811e-02f);
  q = q * u + 1.110860035e-01f;
  q = q * u + (-3.753373921e-01f);
  q = q * u + 1.128336072e+00f;
  f32x2_t pe = z * q;
  pe.x = __builtin_amdgcn_fmed3f(pe.x, -1.0f, 1.0f);
  pe.y = __builtin_amdgcn_fmed3f(pe.y, -1.0f, 1.0f);
  const f32x2_t hx = x * 0.5f;
  return hx * pe + hx;
}
// four pairs at once, Horner steps interleaved across the pairs: hipcc otherwise emits the four dependent chains one
// after the other (a v_pk_fma every ~8 cycles behind an s_nop), i.e. latency-bound with ILP 1
template <int NP>
__device__ __forceinline__ void gelu_bf16x2_xn(f32x2_t (&x)[NP]) {
  f32x2_t z[NP], u[NP], q[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    z[i] = x[i] * 0.70710678118654752440f;
    z[i].x = __builtin_amdgcn_fmed3f(z[i].x, -3.0f, 3.0f);
    z[i].y = __builtin_amdgcn_fmed3f(z[i].y, -3.0f, 3.0f);
  }
#pragma unroll
  for (int i = 0; i < NP; ++i) u[i] = z[i] * z[i];
#pragma unroll
  for (int i = 0; i < NP; ++i) q[i] = u[i] * 4.074456683e-08f + (-1.944940095e-06f);
  constexpr float c[7] = {4.106299457e-05f, -5.110675702e-04f, 4.235681612e-03f, -2.510436811e-02f,
                          1.110860035e-01f, -3.753373921e-01f, 1.128336072e+00f};
#pragma unroll
  for (int k = 0; k < 7; ++k) {
#pragma unroll
    for (int i = 0; i < NP; ++i) q[i] = q[i] * u[i] + c[k];
  }
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    f32x2_t pe = z[i] * q[i];
    pe.x = __builtin_amdgcn_fmed3f(pe.x, -1.0f, 1.0f);
    pe.y = __builtin_amdgcn_fmed3f(pe.y, -1.0f, 1.0f);
    const f32x2_t hx = x[i] * 0.5f;
    x[i] = hx * pe + hx;
  }
}
#else
// GELU for results that are stored as bf16 (the LDS-DMA contraction kernels' epilogues: conv 1-6, FFN-1): y = x * Phi(x) with
// Phi(x) - 1/2 as an odd degree-13 polynomial in t = clamp(x, +-3.8) -- no transcendental issue slots (rcp / exp are quarter
// rate) and the 1/sqrt 2 and 1/2 of the erf form folded into the coefficients: 11 issue slots per PAIR (2 med3, t^2, 6 Horner
// steps, t q + 1/2, x *) instead of ~33 for the exp form and 17 for the degree-17 erf polynomial this replaces (the epilogue of
// a GELU tile is VALU-bound: 5.0 -> 3.6 us per 256 x 256 tile).  |Phi error| <= 6.5e-5; |y error| <= 1.0e-4 |x| (2.4e-4 for
// |x| <= 6, 1e-4 relative for x > 0.01): 20x below the bf16 rounding of the stored value (2^-9 relative).  The additive constant is
// 1/2 - 9.27e-6 (SVT_GELU_HALF), chosen so that the clamped polynomial is 0 at t = -3.8 (6e-9 in fp32): Phi saturates at 6e-9 /
// 1 - 1.9e-5 beyond +-3.8, i.e. the negative tail is flushed (|y| <= 6e-9 |x| for x < -3.8 instead of 9.3e-6 |x|, which grew
// without bound: x = -100 gave -9e-4 for a true value of 0); |Phi error| <= 7.4e-5 overall.  On (-3.8, 0) the error is absolute
// (<= 7.4e-5 |x|), so the RELATIVE error of the tiny outputs there reaches several per cent -- of values below 5e-3.
// fp32 outputs keep gelu_fast.  Coefficients: weighted least-squares minimax fit on [0, 3.8].
#define SVT_GELU_C0 3.9867674568e-01f
#define SVT_GELU_C1 (-6.5719787820e-02f)
#define SVT_GELU_C2 9.3166354115e-03f
#define SVT_GELU_C3 (-9.3154765044e-04f)
#define SVT_GELU_C4 6.0684808363e-05f
#define SVT_GELU_C5 (-2.2724625145e-06f)
#define SVT_GELU_C6 3.6657791326e-08f
#define SVT_GELU_HALF 0.4999907314777374f
__device__ __forceinline__ f32x2_t gelu_bf16x2(f32x2_t x) {
  f32x2_t t;
  t.x = __builtin_amdgcn_fmed3f(x.x, -3.8f, 3.8f);
  t.y = __builtin_amdgcn_fmed3f(x.y, -3.8f, 3.8f);
  const f32x2_t u = t * t;
  f32x2_t q = u * SVT_GELU_C6 + SVT_GELU_C5;
  q = q * u + SVT_GELU_C4;
  q = q * u + SVT_GELU_C3;
  q = q * u + SVT_GELU_C2;
  q = q * u + SVT_GELU_C1;
  q = q * u + SVT_GELU_C0;
  return x * (t * q + SVT_GELU_HALF);
}
// several pairs at once, Horner steps interleaved across the pairs: hipcc otherwise emits the dependent chains one
// after the other (a v_pk_fma every ~8 cycles behind an s_nop), i.e. latency-bound with ILP 1
template <int NP>
__device__ __forceinline__ void gelu_bf16x2_xn(f32x2_t (&x)[NP]) {
  f32x2_t t[NP], u[NP], q[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    t[i].x = __builtin_amdgcn_fmed3f(x[i].x, -3.8f, 3.8f);
    t[i].y = __builtin_amdgcn_fmed3f(x[i].y, -3.8f, 3.8f);
  }
#pragma unroll
  for (int i = 0; i < NP; ++i) u[i] = t[i] * t[i];
#pragma unroll
  for (int i = 0; i < NP; ++i) q[i] = u[i] * SVT_GELU_C6 + SVT_GELU_C5;
  constexpr float c[5] = {SVT_GELU_C4, SVT_GELU_C3, SVT_GELU_C2, SVT_GELU_C1, SVT_GELU_C0};
#pragma unroll
  for (int k = 0; k < 5; ++k) {
#pragma unroll
    for (int i = 0; i < NP; ++i) q[i] = q[i] * u[i] + c[k];
  }
#pragma unroll
  for (int i = 0; i < NP; ++i) x[i] = x[i] * (t[i] * q[i] + SVT_GELU_HALF);
}
#endif
__device__ __forceinline__ void gelu_bf16x2_x4(f32x2_t (&x)[4]) { gelu_bf16x2_xn<4>(x); }
#endif

void set_error(const std::string& msg);
int hip_fail(hipError_t e, const char* what, const char* file, int line);

#define SVT_HIP(expr)                                                        \
  do {                                                                       \
    hipError_t _e = (expr);                                                  \
    if (_e != hipSuccess) return svt::hip_fail(_e, #expr, __FILE__, __LINE__); \
  } while (0)

#define SVT_LAUNCH_CHECK() SVT_HIP(hipGetLastError())

enum Act { ACT_NONE = 0, ACT_GELU = 1, ACT_RELU = 2, ACT_PRELU = 3 };

// C[z][m][n] = act(alpha * sum_k A[z][m][k] * W[z][n][k] + bias[n]) + resid[z][m][n]
// A rows may overlap (implicit-GEMM view of a strided 1-D convolution over a channels-last tensor):
//   row m of batch z starts at A + z-offset + (m / a_rpb) * a_bstride + (m % a_rpb) * a_rstride.
struct GemmArgs {
  const void* A = nullptr;
  const void* W = nullptr;
  void* C = nullptr;
  const float* bias = nullptr;
  const float* resid = nullptr;  // fp32, same indexing as C
  int M = 0, N = 0, K = 0;
  int a_rpb = 1;
  long a_bstride = 0, a_rstride = 0;
  long ldw = 0, ldc = 0;
  int nz = 1, nz2 = 1;  // batch z = z1 * nz2 + z2
  long a_z1 = 0, a_z2 = 0, w_z1 = 0, w_z2 = 0, c_z1 = 0, c_z2 = 0, bias_z2 = 0;
  float alpha = 1.f;
  int act = ACT_NONE;
  int out_f32 = 0;  // C is fp32 regardless of the operand type
  int dbg = 0;      // diagnostic variants (tools/gemm_bench.py): 1 = skip DMA after the prologue, 2 = skip MFMAs
  // gemm_p1w_kernel only (round 6): K slabs of the A operand visited TAP-MINOR.  An implicit-convolution row is k_taps consecutive input
  // frames of k_cin channels (K = k_taps * k_cin); slab g reads tap g % k_taps, channels [(g / k_taps) * 64, + 64), and W is stored in that
  // slab order (W_kperm, written at finalize).  The frame shared by two neighbouring output rows (tap 2 of row t = tap 0 of row t + 1 at
  // kernel 3 / stride 2) is then re-read two slabs later instead of sixteen -- out of L2 instead of HBM (profiles/r06_pmc_conv_kperm.txt).
  int k_taps = 1, k_cin = 0;
  const void* W_kperm = nullptr;   // set by the caller beside W: the dispatcher swaps it in (with k_taps / k_cin below) when it picks gemm_p1w_kernel
  int kperm_taps = 0, kperm_cin = 0;
  int ksplit = 1;   // gemm_skinny_kernel only: K split over this many workgroups (blockIdx.z); each writes its RAW fp32 partial tile (no bias /
  long ksplit_stride = 0;   // activation / residual) to (float*)C + part * ksplit_stride: summed by layernorm_hilo2_kernel<D, true> (round 6)
  int walk_pm = 0;  // persistent kernels' tile walk (tile_walk below): 0 = n fastest, > 0 = panels of this many tile rows (set by gemm_walk_pm)
  int c_vec = 1;    // set by launch_gemm: C / resid / bias rows are 16-byte aligned -> vector epilogue
  int raster_gm = 8;  // set by the register-staged launcher: M-tiles per band of the block -> tile map (gemm.hip)
  // ---- generalised addressing / epilogue (gen = 1): 2-D convolution as implicit GEMM over a zero-padded
  // channels-last tensor (the lip front-end's ResNet), served by the register-staged kernel ----
  int gen = 0;
  int a_d1 = 1, a_d2 = 1;            // A row offset = m * a_rstride + (m / a_d1) * a_e1 + (m / a_d2) * a_e2
  long a_e1 = 0, a_e2 = 0;
  int kseg = 0;                      // K is made of kseg-element contiguous runs kseg_stride elements apart (0 = one run)
  long kseg_stride = 0;
  int c_d1 = 1, c_d2 = 1;            // C (and resid) row offset = c_base + m * ldc + (m / c_d1) * c_e1 + (m / c_d2) * c_e2
  long c_e1 = 0, c_e2 = 0, c_base = 0;
  int c_nsplit = 0;                  // gen: columns >= c_nsplit belong to a SECOND output tensor of the same row addressing, c_nstride
  long c_nstride = 0;                // elements behind the first (column n there = n - c_nsplit); multiples of 8
  const float* slope = nullptr;      // ACT_PRELU: per-column negative slope
  int resid_first = 0;               // act(acc + bias + resid) instead of act(acc + bias) + resid
  int resid_op_type = 0;             // resid is stored in the operand type (bf16 in bf16 mode), not fp32
  // split-operand modes: instead of the fp32 C, the epilogue writes C cut into 16-bit (hi, lo) PLANES -- same (row, ldc) layout,
  // the lo plane plane_stride elements after the hi plane -- which is what the fused split attention reads (the QKV projection).
  // Served by the LDS-DMA split kernel; for any other kernel launch_gemm writes C and cuts it afterwards.
  unsigned short* planes = nullptr;
  long plane_stride = 0;
  int planes_f16 = 0;          // piece type: 1 = IEEE half, 0 = bf16
  // split-operand modes, "pair rows": every 32 consecutive elements of a row stored as [32 hi pieces | 32 lo pieces] (16-bit), i.e. in
  // the 128 bytes of the fp32 slab they replace -- element strides / offsets are those of the fp32 tensor (multiples of 32 elements).
  int a_pairs = 0;             // A holds pair rows (written by a producer kernel): served by gemm_x3q_kernel only
  int c_pairs = 0;             // C is written as pair rows (the next product's operand) instead of fp32
  int stamp_ends = 0;           // diagnostics (gemm_pps_kernel slot stamps): 0 = slot starts, 1 = slot ends
  long long* trace = nullptr;  // diagnostics (dbg == 9): per-workgroup phase clock stamps, 16 x int64 per workgroup
};

// operand type: 0 = fp32 (v_mfma_f32_16x16x4_f32, exact fp32 fma chain), 1 = bf16 (v_mfma_f32_16x16x32_bf16)
int launch_gemm(int prec, const GemmArgs& a, hipStream_t s);

// large-tile LDS-DMA variant (bf16, K % 64 == 0); launch_gemm dispatches to it
bool gemm_dma_eligible(const GemmArgs& a);
int launch_gemm_dma(const GemmArgs& a, hipStream_t s);
// persistent + staggered form of the LDS-DMA pipeline (gemm_pps.hip): bf16 output, no residual, activation none / GELU
bool gemm_pps_eligible(const GemmArgs& a);
int launch_gemm_pps(const GemmArgs& a, int bm, hipStream_t s, int store_policy = 0);   // 0 default, 1 nt, 2 sc1 stores
// small problems (a single utterance): 64 x 64 tiles, K split four ways inside the workgroup, operands straight from L2
bool gemm_skinny_eligible(const GemmArgs& a);
int launch_gemm_skinny(const GemmArgs& a, hipStream_t s);
// split-operand modes: weight matrices cut once into packed 16-bit (hi, lo) pieces for the LDS-DMA split kernel
// (gemm_dma.hip gemm_x3_kernel).  kind = svt_precision (2 = bf16 pieces, 3 = fp16 pieces); other kinds are ignored.
int split_weights_register(const void* w_f32_dev, long n_rows, int K, int kind, hipStream_t s);
void split_weights_forget(const void* w_f32_dev);
int launch_gemm_x3(int kind, const GemmArgs& a, hipStream_t s);  // 0 = launched, 1 = not eligible (use the register-staged kernel), < 0 = error
// persistent staggered form of the LDS-DMA split kernel (gemm_x3p.hip); `packed` = the registered (hi, lo) image of the weight rows
bool gemm_x3p_eligible(const GemmArgs& a);
int launch_gemm_x3p(int kind, const GemmArgs& a, const void* packed, hipStream_t s);
// both operands pre-cut (pair rows): the schedule of gemm_pps_kernel with three MFMAs per block (gemm_x3q.hip)
bool gemm_x3q_eligible(const GemmArgs& a);
int launch_gemm_x3q(int kind, const GemmArgs& a, const void* packed, int bm, hipStream_t s);
int launch_gemm_p1x(int kind, const GemmArgs& a, const void* packed, int bm, hipStream_t s);   // gemm_p1x.hip: the same product, one wave per SIMD (K >= 96)
extern int g_gemm_p1x;   // svt_debug_set key 30
extern int g_gemm_walk;  // svt_debug_set key 34: -1 (default) = choose per launch (gemm_walk_pm), 0 = always n fastest, > 0 = this panel height
// Tile walk of the persistent GEMM kernels (gemm_p1w / gemm_pps): logical tile index -> (tile_m, tile_n).  Blocks b and b + 8 share an XCD
// and an XCD works on 32 consecutive logical tiles per round, so the walk decides what an XCD's L2 has to hold:
//   pm == 0: n fastest -- 32 tiles = a few whole rows of tiles: every A row block is fetched by ONE XCD, all of W by every XCD in every
//            round.  Right while W (N x K) stays resident in the XCD's 4 MiB L2 (every shape of the base model but FFN-1).
//   pm  > 0: panels of pm tile rows walked column by column (m fastest inside a column) -- 32 tiles = pm rows x 32 / pm columns: an
//            XCD round fetches pm A blocks + 32 / pm W blocks instead of ~32 / tiles_n A blocks + ALL tiles_n W blocks.  For the large
//            models' FFN-1 (N = 4096, K = 1024: W = 8 MiB, twice the L2) that is 6 MiB instead of 9 MiB per XCD and round, for a square
//            8192 problem 48 instead of 132 (what the vendor library's kernel name calls WGM / SKXCCM: profiles/r06_gemm_tile_walk.txt).
__device__ __forceinline__ void tile_walk(int logical, int tiles_m, int tiles_n, int pm, int& tile_m, int& tile_n) {
  if (pm <= 0) { tile_n = logical % tiles_n; tile_m = logical / tiles_n; return; }
  const int per_panel = pm * tiles_n;
  const int pnl = logical / per_panel, rem = logical - pnl * per_panel;
  const int rows = min(pm, tiles_m - pnl * pm);      // the last panel may be shorter
  tile_n = rem / rows;
  tile_m = pnl * pm + (rem - tile_n * rows);
}
int gemm_walk_pm(const GemmArgs& a, int bm);   // gemm_dma.hip: the panel height for this launch (0 = n fastest)
extern int g_x3_pairs;  // 1 (default): the split modes keep product operands as pair rows; 0: fp32 activations cut inside the product kernels (svt_debug_set key 19)
extern int g_ln_two_rows;  // (hi, lo) LayerNorm: half a wave per row, 16-byte accesses (1, default) or a wave per row (0)
extern int g_gemm_x3;  // 1 (default): use it where eligible; 0: register-staged split kernel only (svt_debug_set key 11)
extern int g_flash_wide;  // fused attention: 8-wave (256-query) workgroups for head_dim 64 (1, default) or 4-wave ones (0)
extern int g_gemm_skinny_max_tiles;
extern int g_gemm_skinny_small_tiles;   // svt_debug_set key 33
extern int g_gemm_persist_wgs;   // svt_debug_set key 37: workgroups of a persistent GEMM launch (256 = one per CU, default; a multiple of 8)
extern int g_conv_kperm;     // svt_debug_set key 35: 1 (default) = tap-minor K order for the kernel-3 convolutions on gemm_p1w_kernel, 0 = tap-major
extern int g_ffn2_ksplit;   // svt_debug_set key 36 (api.hip): FFN-2 of a small batch as a K-split small GEMM + summing LayerNorm
extern int g_gemm_skinny;  // 1 (default): small problems use it; 0: never (diagnostics, svt_debug_set key 6)
extern int g_stamp_ends;
extern int g_pps_half_barriers;
extern int g_gemm_p1w;       // svt_debug_set key 29: the single-wave-per-SIMD kernel (gemm_p1w.hip) where gemm_pps_kernel is dispatched
int launch_gemm_p1w(const GemmArgs& a, int bm, hipStream_t s);
extern int g_pps_two_slots;   // gemm_pps_kernel: two slots per slab (svt_debug_set key 28)
extern int g_attn_stamp;
extern int g_attn_variant;
extern int g_gemm_dbg;   // diagnostic variant applied to every launch (svt_debug_set)
extern int g_gemm_force_bm;
extern int g_gemm_ring;
extern int g_gemm_variant;  // diagnostics: replaces dbg inside the kernel while the trace pointer stays set

// hipFuncAttributeMaxDynamicSharedMemorySize for a kernel that asks for more than 64 KiB of dynamic LDS: function attributes are
// per DEVICE, so the "already set" memory is per (kernel, device), behind a mutex (DataParallel runs replicas from threads)
int ensure_dyn_lds(const void* kernel, int bytes);

// ---- device allocations of the library (api.hip): hipMalloc, or page-guarded mappings under svt_debug_set key 13 ----
int dev_alloc(void** out, size_t bytes);
void dev_free(void* p);
extern int g_guard_alloc;

// ---- profiling of the dominant kernel (bench.py roofline leg) ----
void prof_begin(hipStream_t s);
// kind: 0 = the dominant family (gemm_pps / gemm_pers / gemm_pp8 kernels; the split form of gemm_kernel), 1 = other dense contraction kernels, 2 = flash attention
void prof_end(hipStream_t s, double flops, double bytes, int kind = 1);

// ---- elementwise / reduction kernels (kernels.hip) ----
int launch_f32_to_bf16(const float* in, bf16_t* out, int64_t n, hipStream_t s);
// fp32 <-> pair rows (GemmArgs::a_pairs); lo_plane != nullptr: `in` / `lo_plane` are separate (hi, lo) planes instead
int launch_f32_to_pairs(int kind, const float* x, void* out, int64_t n, hipStream_t s);
int launch_pairs_to_f32(int kind, const void* in, const void* lo_plane, float* y, int64_t n, hipStream_t s);
// moments[0] = sum(x), moments[1] = sum(x^2) over n fp32 values (fp64 accumulation, workgroup partials added in a fixed order: the
// result is reproducible bit for bit); scratch = moments_scratch_bytes(groups_max) bytes whose first 4 * groups_max bytes are ZERO
int launch_zero_bytes(void* p, size_t bytes, hipStream_t s);            // kernel nodes instead of memset / memcpy nodes (kernels.hip)
int launch_copy_f32(const float* in, float* out, int64_t n, hipStream_t s);
int launch_zero_cols(float* p, int64_t rows, int cols, int64_t ld, hipStream_t s);
int set_ticket_fenced(int on);   // svt_debug_set key 32 (kernels.hip, last_workgroup)
int launch_moments(const float* x, int64_t n, double* moments, void* scratch, int groups_max, hipStream_t s, int groups = 1);  // groups > 1: n elements and 2 doubles per group
size_t moments_scratch_bytes(int groups_max);
// y = (x - mean) * rsqrt(var + eps) from global moments over n elements (no affine)
int launch_global_norm(const float* x, float* y, int64_t n, const double* moments, float eps, hipStream_t s, int groups = 1,
                       double n_stat = 0);  // n_stat > 0: the moments were summed over n_stat elements (cross-rank reduction), not n

// row LayerNorm over D: in fp32 or operand type; outputs: yT (operand type, may be null) and yF (fp32, may be null)
// optional `add` (fp32, same shape) is summed into x before the statistics (residual add fused into the norm);
// `sumF` (optional) receives x + add (the updated residual stream of the pre-LN encoder)
int launch_layernorm(int prec, const void* x, int x_is_f32, int64_t rows, int D, const float* gamma,
                     const float* beta, float eps, int gelu, void* yT, float* yF, hipStream_t s,
                     const float* add = nullptr, float* sumF = nullptr,
                     void* yP = nullptr, int pair_kind = 0,    // yP: the result as pair rows (split modes; kind 2 = bf16 / 3 = fp16 pieces)
                     const void* addP = nullptr);              // addP: a second addend given as pair rows (may alias yP: in place)

// conv layer 0 (Cin = 1) in "group" mode: per-(clip,channel) GroupNorm folded into 11 coefficients
int launch_conv0_window_moments(const float* wav, int B, int64_t L, int k, int stride, int64_t T1,
                                double* wm /*B x 65*/, void* scratch /*conv0_window_moments_scratch_bytes; first 4 B bytes ZERO*/, hipStream_t s);
size_t conv0_window_moments_scratch_bytes(int B, int64_t T1);
int launch_conv0_group_coef(const double* wav_moments /*2, or null*/, int64_t n_wav, const double* wm, int B,
                            int64_t T1, int C, int k, const float* w0 /*C x k*/, const float* b0 /*C or null*/,
                            const float* gamma, const float* beta, float eps_wav, float eps_gn,
                            float* coef /*B x C x (k+1)*/, hipStream_t s, int clips_per_norm_group);  // wav_moments: 2 per group
int launch_conv0_group_apply(int prec, const float* wav, int B, int64_t L, int k, int stride, int64_t T1, int C,
                             const float* coef, void* out /*B x T1 x C*/, hipStream_t s, int pair_kind = 0);   // pair_kind 2 / 3: out as pair rows
// conv layer 0 in "layer" mode: conv + bias + LayerNorm over channels + GELU
int launch_conv0_layer(int prec, const float* wav, int B, int64_t L, int k, int stride, int64_t T1, int C,
                       const double* wav_moments, int64_t n_wav, float eps_wav, const float* w0, const float* b0,
                       const float* gamma, const float* beta, float eps, void* out, hipStream_t s, int clips_per_norm_group,
                       int pair_kind = 0);

// conv layer 0 on the matrix pipe (conv0_mfma.hip; 16-bit throughput modes, k = 10, C = 512): table_ws = conv0_mfma_table_bytes(B) bytes
bool conv0_mfma_ok(int prec, int pair_kind, int k, int stride, int C);   // prec 1 (16-bit rows out) or pair_kind 2 / 3 (split modes: pair rows out)
size_t conv0_mfma_table_bytes(int B);
int launch_conv0_mfma_group(const float* wav, int B, int64_t L, int stride, int64_t T1, const float* coef, void* table_ws, void* out,
                            hipStream_t s, int pair_kind = 0);
int launch_conv0_mfma_layer(const float* wav, int B, int64_t L, int stride, int64_t T1, const double* wav_moments, int64_t n_wav,
                            float eps_wav, const float* w0, const float* b0, const float* gamma, const float* beta, float eps, void* table_ws,
                            void* out, hipStream_t s, int clips_per_norm_group, int pair_kind = 0);
extern int g_conv0_mfma;

// positional-conv operand: (B,T,D) fp32 -> (B, G, T + kp, D/G) operand type, zero padded by kp/2 in front
int launch_posconv_gather(int prec, const float* h, int B, int T, int D, int G, int kp, int Tp, void* out, hipStream_t s,
                          const float* sc = nullptr, const float* sh = nullptr);  // sc/sh: per-channel affine on the valid frames
int launch_posconv_scatter_add(const float* h, const void* y, int B, int T, int D, int G, int P, int Tq, float* pre, hipStream_t s,
                               int y_f32 = 0);   // y in the operand type, or fp32 (split modes)

// attention helpers for the materialised-score path
int launch_softmax_rows(int prec, const float* S, int64_t rows, int T, int Tp, void* P, hipStream_t s);
// V slice of the packed qkv (B*T, ld) at column offset voff -> Vt (B, H, dh, Tp) operand type, zero padded
int launch_transpose_v(int prec, const void* qkv, int B, int T, int H, int dh, long ld, long voff, int Tp,
                       void* Vt, hipStream_t s);

// fused attention (bf16, head_dim 64/128): Q/K row-major with row strides ldq/ldk and per-clip strides, V^T padded
// Q/K/V row-major (K and V share row / clip strides); V is transposed on the fly by ds_read_b64_tr_b16
// gate (B,H,T) / pb (H, 2T-1): WavLM's gated relative position bias, or nullptr
int launch_flash_attention(const void* Q, long ldq, long q_bstride, const void* K, const void* V, long ldk, long k_bstride,
                           void* O, long ldo, long o_bstride, int B, int T, int H, int dh, float scale, hipStream_t s,
                           const float* gate = nullptr, const float* pb = nullptr);

// split-operand fused attention (attention.hip): fp32 -> 16-bit (hi, lo) planes, then softmax(scale Q K^T) V with three MFMAs
// per product; kind = svt_precision (2 = bf16 pieces, 3 = fp16 pieces)
int launch_split_planes(int kind, const float* src, long ld_src, long rows, int cols, void* dst, long ld_dst, long plane, hipStream_t s);
bool flash_attention_x3_ok(int dh);
int launch_flash_attention_x3(int kind, const void* Q, long ldq, long q_bstride, long q_plane, const void* K, const void* V, long ldk,
                              long k_bstride, long k_plane, float* O, long ldo, long o_bstride, int B, int T, int H, int dh, float scale,
                              hipStream_t s, int o_pairs = 0);   // o_pairs: O written as pair rows (the output projection's operand)
// out = a*x + b*y (fp32 or operand type)
int launch_axpby(int prec, const void* x, const void* y, float a, float b, void* out, int64_t n, hipStream_t s);
// RCA: s = x + pe[t] (x f32 (B,T,D); x2 may be shorter in T: rows >= T2 read as zero) -> fp32 + operand type
int launch_add_pe(int prec, const float* x, int B, int T, int Tsrc, int D, const float* pe, float* outF, void* outT,
                  hipStream_t s);
int launch_add_f32(const float* a, const float* b, float* out, int64_t n, hipStream_t s);

// frame head (fp32 GEMV, N small) and per-frame decode
int launch_linear_f32(const float* x, int64_t rows, int K, const float* w, const float* b, int N, float* y,
                      hipStream_t s);
// frame head for K in {512,768,1024}, N <= 32: weight in LDS, four rows per wave (HBM-bound)
// post-LN residual stream as a bf16 (hi, lo) pair (throughput mode, D in {512, 768, 1024})
bool layernorm_hilo_ok(int D);
// the same LayerNorm with the branch given as `nparts` fp32 partial products + bias (a K-split small GEMM: gemm_skinny.hip, ksplit)
int launch_layernorm_hilo_parts(const float* parts, int nparts, long part_stride, const float* pbias, const bf16_t* rh, const bf16_t* rl, int64_t rows,
                                int D, const float* gamma, const float* beta, float eps, bf16_t* yh, bf16_t* yl, float* yF, hipStream_t s);
int launch_layernorm_hilo(const bf16_t* branch, const bf16_t* rh, const bf16_t* rl, const float* x32, int64_t rows, int D,
                          const float* gamma, const float* beta, float eps, bf16_t* yh, bf16_t* yl, float* yF, hipStream_t s);
// attention output projection + residual + LayerNorm in one kernel (gemm_ln.hip; hidden size 768, bf16 mode)
// Fbank add-ons: time derivatives and context window
int launch_deltas(const float* x, long ldx, int B, int T, int C, int n, float inv_denom, float* out, long ldo, hipStream_t s);
int launch_context_window(const float* x, int B, int T, int C, int ctx, int lag, int pad, float* out, hipStream_t s);
// WavLM gated relative position bias
int launch_relpos_table(const float* embed, int H, int T, int num_buckets, int max_distance, float* pb, hipStream_t s);
int launch_relpos_gate(int prec, const void* u, int64_t rows, int T, int H, int dh, const float* wab, const float* bab,
                       const float* cst, float* gate, hipStream_t s);
int launch_scores_add_relbias(float* S, int64_t BH, int H, int T, int Tp, const float* gate, const float* pb, hipStream_t s);
// lip front-end (video.hip)
int launch_video_pad(int prec, const float* v, int B, int T, int H, int W, int Hp, int Wp, void* out, hipStream_t s);
// the recipe's uint8 -> float32 pixel map, ((u - sub0) / div0 - mean) / std in float64 (video.hip, svt_video_forward_u8)
struct VideoTransform { double sub0, div0, mean, std; };
int launch_video_pad_u8(int prec, const unsigned char* v, int B, int T, int Hin, int Win, int dy, int dx, int H, int W, int Hp, int Wp,
                        const VideoTransform& tf, void* out, hipStream_t s);
int launch_conv3d_front(int prec, const void* vp, const void* w, const float* bias, const float* slope, long F, int T, int Hp,
                        int Wp, int H0, int W0, void* out, hipStream_t s);
int launch_maxpool_3x3s2(int prec, const void* in, long F, int H0, int W0, int C, int H1, int W1, void* out, hipStream_t s);
// stem + max-pool in one persistent kernel (16-bit storage; six plane slots + a 9-row band of the stem output in LDS)
extern int g_stem_pool_fused;
bool conv3d_front_pool_ok(int prec, int Hp, int Wp, int W0);
int launch_conv3d_front_pool(const void* vp, const void* w, const float* bias, const float* slope, long F, int T, int Hp, int Wp, int H0,
                             int W0, int H1, int W1, void* out, hipStream_t s);
int launch_zero_halo(int prec, void* buf, long F, int Hp, int Wp, int C, hipStream_t s);
int launch_avgpool_interior(int prec, const void* in, long F, int H, int W, int C, void* out, hipStream_t s);
// conv3x3_c64.hip: stage 1 of the lip front-end (64 -> 64 channels, 3x3) with the frame and the weights resident in LDS
extern int g_conv3x3_c64, g_conv3x3_c64_launches, g_conv3x3_c64_form;
bool conv3x3_c64_ok(int prec, int Hs, int Ws);
int launch_conv3x3_c64(const void* in, const void* wimg, const float* bias, const float* slope, const void* resid, void* out, long F,
                       int Hs, int Ws, hipStream_t s);
bool conv3x3_c128_ok(int prec, int Hs, int Ws);   // stage 2's stride-1 convolutions (128 -> 128 channels), same file
int launch_conv3x3_c128(const void* in, const void* wimg, const float* bias, const float* slope, const void* resid, void* out, long F,
                        int Hs, int Ws, hipStream_t s);
// validation losses (masked BCE-with-logits / NLL, speechbrain/nnet/losses.py) and the narrow (log-)softmax
int launch_bce_loss(const float* x, int64_t B, int64_t t_pred, const float* y, int64_t t_tgt, int64_t T, const float* rel_len,
                    const float* pos_weight, float* per_frame, double* sums, hipStream_t s);
int launch_nll_loss(const float* logp, int64_t B, int64_t t_pred, int C, const int64_t* tgt, int64_t t_tgt, int64_t T,
                    const float* rel_len, float* per_frame, double* sums, int* bad_target, hipStream_t s);
int launch_loss_reduce(const double* sums, int B, int reduction, float smoothing, float* out, hipStream_t s);
int launch_softmax_small(const float* x, int64_t rows, int n, int apply_log, float* y, hipStream_t s);
bool linear_head_eligible(int K, int N);
int launch_linear_head(const float* x, int64_t rows, int K, const float* w, const float* b, int N, float* y, hipStream_t s);
struct FrameOut { float p_on, p_off; int32_t octave, pitch_class; };
// fused tail: whole-batch output norm (moments taken in the same pass) + frame head + optional per-frame decode, from the
// UN-normalised encoder output x (rows x K); dots = rows x N scratch; mom = 2 doubles per norm group (written, reproducible bit for
// bit) or null; scratch (needed with mom) = head_scratch_bytes(rows, groups) bytes whose first 4 are ZERO
int launch_head_fused(const float* x, int64_t rows, int K, const float* w, const float* wsum, const float* b, int N, float* dots,
                      double* mom, int64_t rows_per_group, float eps, float* logits, FrameOut* frames, int n_oct, int n_cls,
                      hipStream_t s,
                      double n_stat = 0, int (*between)(void*) = nullptr, void* between_arg = nullptr, void* scratch = nullptr);
size_t head_scratch_bytes(int64_t rows, int groups_max);
int launch_decode_frames(const float* logits, int64_t rows, int n_out, int n_oct, int n_cls, FrameOut* out,
                         hipStream_t s);
int launch_ctc_greedy(const float* probs, int B, int T, int V, const float* rel_lens, int blank, int32_t* tokens,
                      int32_t* out_lens, hipStream_t s);

int launch_clock_stamp(long long* out16, hipStream_t s);
// Fbank pieces
int launch_fbank_frames(const float* wav, int B, int64_t L, int n_fft, int hop, int64_t nframes, const float* window,
                        float* frames /*B*nframes x n_fft*/, hipStream_t s);
int launch_power_spectrum(const float* reim /*rows x ld: re at [0,nb), im at [imoff, imoff+nb)*/, int64_t rows, int nb,
                          int imoff, int ld, float* power /*rows x ldp*/, int ldp, hipStream_t s);
int launch_fbank_db(float* fb, int B, int64_t per_seq, float top_db, hipStream_t s);

}  // namespace svt

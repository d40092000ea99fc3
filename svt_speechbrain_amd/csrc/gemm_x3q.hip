// Split-operand products (precision "fp16x3" / "bf16x3") with BOTH operands already cut: the schedule of gemm_pps.hip on "pair rows".
//
// A pair row stores every 32 consecutive elements of an fp32 row as [32 hi pieces | 32 lo pieces] (16-bit each): 128 bytes, the bytes
// of the fp32 slab it replaces, so buffers, strides and offsets are those of the fp32 tensor.  The weights have been kept that way
// since round 2 (split_pack_kernel); since round 4 the PRODUCERS of every activation that feeds a product (conv0, the LayerNorm
// kernels, the fused split attention, this kernel's own epilogue) write pair rows too, so the 128-byte line that LDS-DMA moves per
// row and slab holds the hi fragment in its first half and the lo fragment in its second -- exactly the image of gemm_pps_kernel's
// 64-deep bf16 slab.  What changes against that kernel is the MFMA pattern: where it multiplies (k-step 0 x k-step 0) +
// (k-step 1 x k-step 1), this one multiplies hi x hi + hi x lo + lo x hi, three MFMAs per 16 x 16 x 32 block from the SAME fragment
// reads; the in-kernel cut of gemm_x3p / gemm_x3s (a 200-cycle dependent convert chain in two LOAD slots of every slab, and a wave
// layout of 8 (M) x 1 (N) in which every wave read the whole W unit: 288 KiB of LDS reads per slab and CU) is gone.
//   * slab = 256 x 256 x 32 (or 192 / 128 rows): 96 MFMAs per wave as SIX slots of 16 (MB = 4), each behind a LOAD slot that reads
//     one fragment set; one activation set and one weight set are live at a time (the register budget of gemm_pps_kernel):
//       slot 0: x <- A.hi, w <- W.hi(cols 0-63)    acc[0..3] += x w      slot 3: w <- W.hi(cols 64-127)  acc[4..7] += x w
//       slot 1: w <- W.lo(cols 0-63)               acc[0..3] += x w      slot 4: x <- A.lo               acc[4..7] += x w
//       slot 2: w <- W.lo(cols 64-127)             acc[4..7] += x w      slot 5: w <- W.hi(cols 0-63)    acc[0..3] += x w
//     28 fragment reads per wave and slab (224 KiB per CU) against 24 if both activation sets were kept (16 more registers);
//   * ring, stagger, six barriers per slab (the four-barrier form of gemm_pps_kernel with two more intervals), unconditional ring
//     requests (W_{g+1} in slots 0 - 1, A_{g+2} in slots 3 - 4), bias as the accumulators' initial value, register-direct stores in
//     the vector-offset form: gemm_pps.hip;
//   * A rows are addressed by 32-bit offsets from the tile's first row (64-bit base per tile parity): the fp32-sized activation of
//     conv1 exceeds 4 GiB at 64 x 10 s;
//   * epilogue, from the accumulators (a lane holds 4 rows x 8 consecutive columns per 16-row block): OUT 0 = fp32 rows, 1 = pair rows
//     (the next product's operand: conv i -> conv i+1, FFN-1 -> FFN-2), 2 = separate (hi, lo) planes (the QKV projection for the fused
//     split attention); bias; exact-erf GELU (run-time flag).
// Contract: plain product (no batch, no residual, alpha 1) of pair rows against a registered split weight matrix, K % 32 == 0,
// K >= 64, N % 256 == 0, M >= 128.  Everything else stays on gemm_x3s_kernel / the register-staged split kernel, which cut fp32 rows.
#include "common.h"

#ifdef SVT_OPERAND_F16
// The split-operand engines live in the bf16 build only (libsvt_mi355.so): precision codes 2 / 3 are rejected by the IEEE-half build.
namespace svt {
bool gemm_x3q_eligible(const GemmArgs&) { return false; }
int launch_gemm_x3q(int, const GemmArgs&, const void*, int, hipStream_t) { set_error("gemm_x3q: not part of the IEEE-half build"); return -1; }
}  // namespace svt
#else

namespace svt {
namespace {

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void dma_sv(unsigned voff, const void* sbase, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_addr), "v"(voff), "s"(sbase) : "memory");
}
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8v __attribute__((ext_vector_type(8)));

template <bool F16> __device__ __forceinline__ f32x4 mma16(const u32x4v& a, const u32x4v& b, const f32x4& c) {
  if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8v, a), __builtin_bit_cast(f16x8v, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(real_bf16x8, a), __builtin_bit_cast(real_bf16x8, b), c, 0, 0, 0);
}
// eight fp32 values -> packed (hi, lo) 16-bit pieces
template <bool F16> __device__ __forceinline__ void cut8v(const float (&v)[8], u32x4v& hi, u32x4v& lo) {
  if constexpr (F16) {
    f16x8v h, l;
#pragma unroll
    for (int j = 0; j < 8; ++j) { h[j] = (_Float16)v[j]; l[j] = (_Float16)(v[j] - (float)h[j]); }
    hi = __builtin_bit_cast(u32x4v, h);
    lo = __builtin_bit_cast(u32x4v, l);
  } else {
    real_bf16x8 h, l;
#pragma unroll
    for (int j = 0; j < 8; ++j) { h[j] = (__bf16)v[j]; l[j] = (__bf16)(v[j] - (float)h[j]); }
    hi = __builtin_bit_cast(u32x4v, h);
    lo = __builtin_bit_cast(u32x4v, l);
  }
}
template <int N> __device__ __forceinline__ void touch_frag(u32x4v (&r)[N]) {
  static_assert(N >= 2 && N <= 4, "fragment arrays of 2..4 blocks");
  if constexpr (N == 4) asm volatile("" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]));
  else if constexpr (N == 3) asm volatile("" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]));
  else asm volatile("" : "+v"(r[0]), "+v"(r[1]));
}

// DBG (timing ablations, make DIAG=1 + svt_debug_set(0, n); results are WRONG): bit 0 = no LDS-DMA after the head of the stream,
// bit 1 = no fragment reads after the first LOAD slot, bit 2 = no MFMAs, bit 3 = no barriers between the slots of a slab
// NQ = 3 (round 4, dispatched): the slab as THREE slots of 32 MFMAs instead of six of 16 -- lo x hi, hi x hi, hi x lo over all eight column
// blocks of the wave (eight weight fragments live: +16 registers, 24 fragment reads per slab instead of 28).  The timing ablations
// (tools/x3q_bench.py --dbg, profiles/r04_gemm_x3q_ablation.txt) put the main loop's loss in the barrier intervals themselves: with
// neither LDS-DMA nor fragment reads the six-slot slab still takes ~4 100 cycles against 3 072 of MFMA issue, ~170 cycles per interval.
template <bool F16, int BM, int OUT, int DBG = 0, int NQ = 3>
__global__ __launch_bounds__(512) void gemm_x3q_kernel(GemmArgs p, const void* wsplit, int tiles_n, int ntiles) {
  constexpr int BN = 256, BK = 32, NSLOT = 5;
  constexpr int MB = BM / 64;        // 16-row blocks per wave (wave tile BM/4 x 128)
  constexpr int GA = BM / 64, GW = BN / 64;
  constexpr int SLOT = 2048;        // uint4 per ring slot (32 KiB)
  constexpr int ROWB = BK * 4;      // bytes of a pair-row slab
  extern __shared__ __attribute__((aligned(16))) uint4 lds[];
  const u32x4v* ldsv = (const u32x4v*)lds;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave & 3, wn = (wave >> 2) & 1;   // waves 0-3 (columns 0-127) run one slot ahead of waves 4-7 (columns 128-255)
  const int nblk = gridDim.x, b = blockIdx.x;
  const int per = nblk >> 3;
  const int lbase = (b & 7) * per + (b >> 3);      // blocks b and b + 8 share an XCD and take consecutive tiles (n fastest)
  if (lbase >= ntiles) return;
  const int my_tiles = (ntiles - lbase + nblk - 1) / nblk;

  const char* gW = (const char*)wsplit;            // [N][K / 32][hi 32 | lo 32]: a row of the packed matrix is 4 K bytes
  const int r8 = lane >> 3, ch = (lane & 7) ^ r8;
  auto a_row_off = [&](int m) -> long { return ((long)(m / p.a_rpb) * p.a_bstride + (long)(m % p.a_rpb) * p.a_rstride) * 4; };
  // per tile parity (even / odd tiles of this workgroup's list: gemm_pps.hip): the 64-bit address of the tile's first A row, 32-bit
  // offsets of this lane's rows from it, 32-bit offsets of its W rows from the packed matrix
  const char* abE;
  const char* abO;
  unsigned aofE[GA], wofE[GW], aofO[GA], wofO[GW];
  auto setup = [&](int logical, const char*& ab, unsigned (&ao)[GA], unsigned (&wo)[GW]) {
    const int tile_n = logical % tiles_n, tile_m = logical / tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const long o0 = a_row_off(m0);
    ab = (const char*)p.A + o0;
#pragma unroll
    for (int i = 0; i < GA; ++i) {
      int m = m0 + (wave + 8 * i) * 8 + r8;
      if (m > p.M - 1) m = p.M - 1;
      ao[i] = (unsigned)(a_row_off(m) - o0) + ch * 16;
    }
#pragma unroll
    for (int i = 0; i < GW; ++i) {
      const int rho = (wave + 8 * i) * 8 + r8;   // LDS row of the W unit: (128-column group, block nb, row j) <- output column 8 j + nb
      const int n = n0 + (rho >> 7) * 128 + (rho & 15) * 8 + ((rho >> 4) & 7);   // < N: N % 256 == 0
      wo[i] = (unsigned)((long)n * p.K * 4 + ch * 16);
    }
  };
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(void __attribute__((address_space(3)))*)lds);
  auto lds_unit = [&](int slot, int i) -> unsigned { return lds0 + (unsigned)(slot * SLOT + (wave + 8 * i) * 64) * 16u; };

  f32x4 acc[8][MB];
  const int cq = lane >> 4, r16 = lane & 15, rr8 = r16 & 7;
  const int fragH = (r16 >> 3) * 64 + rr8 * 8 + (cq ^ rr8);          // hi pieces k = 8 cq .. + 7: chunk cq
  const int fragL = (r16 >> 3) * 64 + rr8 * 8 + ((4 + cq) ^ rr8);    // lo pieces: chunk 4 + cq
  const int xoff = (wm * MB) * 128;   // uint4 index of the wave's first 16-row block of the A unit
  const int woff = (wn * 8) * 128;    // ... of the W unit

  const int nk = p.K / BK;             // >= 2 (launcher)
  const int G = my_tiles * nk;         // slabs in this workgroup's stream
  const bool has_bias = p.bias != nullptr;
  const bool do_gelu = p.act == ACT_GELU;
  setup(lbase, abE, aofE, wofE);
  setup(my_tiles > 1 ? nblk + lbase : lbase, abO, aofO, wofO);   // always rows that exist: the stream's surplus requests read them
  // bias of this lane's 8 columns for the NEXT tile to start (the accumulators' initial value): gemm_pps.hip
  f32x4 bq[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) bq[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (has_bias) {
    const float* bp = p.bias + ((lbase % tiles_n) * BN + wn * 128 + (lane & 15) * 8);
    asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %2, off offset:16"
                 : "+v"(bq[0]), "+v"(bq[1]) : "v"(bp) : "memory");
  }
  // head of the stream: A_0 -> slot 0, W_0 -> slot 1, A_1 -> slot 2
#pragma unroll
  for (int i = 0; i < GA; ++i) dma_sv(aofE[i], abE, lds_unit(0, i));
#pragma unroll
  for (int i = 0; i < GW; ++i) dma_sv(wofE[i], gW, lds_unit(1, i));
#pragma unroll
  for (int i = 0; i < GA; ++i) dma_sv(aofE[i], abE + ROWB, lds_unit(2, i));
  wait_vm<GA>();   // the bias loads are older than every request of the head
  asm volatile("" : "+v"(bq[0]), "+v"(bq[1]));
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < MB; ++j) acc[i][j] = f32x4{bq[i >> 2][i & 3], bq[i >> 2][i & 3], bq[i >> 2][i & 3], bq[i >> 2][i & 3]};

  u32x4v wfr[NQ == 3 ? 8 : 4], xfr[MB];
  int sa = 0, sw = 1, kt = 0, ti = 0;
  const int grp = wave >> 2;   // = wn
  bool first_load = true;
#define X3Q_SLOTBAR() { if (!(DBG & 8)) __builtin_amdgcn_s_barrier(); }

  // ---- the tile's epilogue: accumulators (bias already inside) -> activation -> fp32 rows / pair rows / planes; clears the
  //      accumulators to the next tile's bias and rotates the source offsets ----
  auto epilogue = [&]() {
    const int logical = ti * nblk + lbase;
    const int tile_n = logical % tiles_n, tile_m = logical / tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const long rows_left = (long)p.M - m0;
    constexpr int ESZ = OUT == 2 ? 2 : 4;   // bytes per element of a row of the output (pair rows: 4, like fp32)
    const unsigned long nbytes = (unsigned long)rows_left * p.ldc * ESZ;
    const unsigned nrec = nbytes > 0xFFFFFFF0ul ? 0xFFFFFFF0u : (unsigned)nbytes;
    int ln;   // the lane index, recomputed in place: keeps the epilogue's addresses out of the slab loop's live set (gemm_x3p.hip)
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
    const int col = n0 + wn * 128 + (ln & 15) * 8;       // first of this lane's 8 consecutive columns
    const long row0 = wm * (BM / 4) + 4 * (ln >> 4);     // rows row0 + 16 mb + r
    const unsigned row_pitch = (unsigned)(p.ldc * ESZ);
    unsigned off0;
    if constexpr (OUT == 0) off0 = (unsigned)((row0 * p.ldc + col) * 4);
    else if constexpr (OUT == 1) off0 = (unsigned)(row0 * p.ldc * 4 + (col >> 5) * 128 + (col & 31) * 2);
    else off0 = (unsigned)((row0 * p.ldc + col) * 2);
    char* cbase = (OUT == 2 ? (char*)p.planes : (char*)p.C) + (long)m0 * p.ldc * ESZ;
    const auto crsrc = __builtin_amdgcn_make_buffer_rsrc(cbase, 0, nrec, 0x00020000);
    const auto lrsrc = __builtin_amdgcn_make_buffer_rsrc(OUT == 2 ? cbase + p.plane_stride * 2 : cbase, 0, nrec, 0x00020000);
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = acc[j][mb][r];
        if (do_gelu) {   // exact-erf GELU on pairs (packed fp32 FMAs; the GELU + cut of a tile costs 6 us of a 75 us FFN-1 tile either way:
                         // two waves per SIMD share the vector pipe, where a packed FMA takes the cycles of two single ones)
#pragma unroll
          for (int j = 0; j < 8; j += 2) {
            const f32x2_t y = gelu_fast2(f32x2_t{v[j], v[j + 1]});
            v[j] = y.x; v[j + 1] = y.y;
          }
        }
        // row and column in the VECTOR offset: range check + the soffset store-data hazard (gemm_pps.hip)
        const unsigned off = off0 + (mb * 16 + r) * row_pitch;
        if constexpr (OUT == 0) {
          const u32x4v v0 = {__builtin_bit_cast(unsigned, v[0]), __builtin_bit_cast(unsigned, v[1]), __builtin_bit_cast(unsigned, v[2]), __builtin_bit_cast(unsigned, v[3])};
          const u32x4v v1 = {__builtin_bit_cast(unsigned, v[4]), __builtin_bit_cast(unsigned, v[5]), __builtin_bit_cast(unsigned, v[6]), __builtin_bit_cast(unsigned, v[7])};
          __builtin_amdgcn_raw_buffer_store_b128(v0, crsrc, off, 0, 16);
          __builtin_amdgcn_raw_buffer_store_b128(v1, crsrc, off + 16, 0, 16);
        } else {
          u32x4v hi, lo;
          cut8v<F16>(v, hi, lo);
          __builtin_amdgcn_raw_buffer_store_b128(hi, crsrc, off, 0, 16);
          __builtin_amdgcn_raw_buffer_store_b128(lo, lrsrc, OUT == 1 ? off + 64 : off, 0, 16);
        }
      }
    }
    // the next tile starts from its bias (fetched during this tile's last slab, behind the counted wait that retired that slab)
    asm volatile("" : "+v"(bq[0]), "+v"(bq[1]));
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < MB; ++j) acc[i][j] = f32x4{bq[i >> 2][i & 3], bq[i >> 2][i & 3], bq[i >> 2][i & 3], bq[i >> 2][i & 3]};
    ++ti;
    if (ti + 1 < my_tiles) {
      const char* nb_;
      unsigned na[GA], nw[GW];
      setup((ti + 1) * nblk + lbase, nb_, na, nw);
      const bool into_odd = (ti & 1) == 0;
      abO = into_odd ? nb_ : abO;
      abE = into_odd ? abE : nb_;
#pragma unroll
      for (int i = 0; i < GA; ++i) { aofO[i] = into_odd ? na[i] : aofO[i]; aofE[i] = into_odd ? aofE[i] : na[i]; }
#pragma unroll
      for (int i = 0; i < GW; ++i) { wofO[i] = into_odd ? nw[i] : wofO[i]; wofE[i] = into_odd ? wofE[i] : nw[i]; }
    }
  };

  // LOAD slot Q of slab g (table in the header) + the ring's requests: W_{g+1} in slots 0 - 1, A_{g+2} in slots 3 - 4, two DMA
  // instructions per slot (GA / 2 for A); the bias of the finishing tile's successor in slot 1 of a tile's last slab
#define X3Q_LOAD(Q)                                                                                                 \
  {                                                                                                                 \
    constexpr int q_ = (Q);                                                                                         \
    const bool rd_ = !(DBG & 2) || first_load;                                                                      \
    if (q_ == 0 && rd_) { _Pragma("unroll") for (int jj = 0; jj < MB; ++jj) xfr[jj] = xa[jj * 128 + fragH]; }       \
    if (q_ == 4 && rd_) { _Pragma("unroll") for (int jj = 0; jj < MB; ++jj) xfr[jj] = xa[jj * 128 + fragL]; }       \
    if (q_ != 4 && rd_) {                                                                                           \
      constexpr int half_ = (q_ == 2 || q_ == 3) ? 1 : 0;                                                           \
      constexpr bool lo_ = (q_ == 1 || q_ == 2);                                                                    \
      _Pragma("unroll") for (int nb = 0; nb < 4; ++nb) wfr[nb] = wa[(half_ * 4 + nb) * 128 + (lo_ ? fragL : fragH)]; \
    }                                                                                                               \
    if ((q_ == 0 || q_ == 1) && !(DBG & 1)) {                                                                       \
      const char* wb = w_cur ? gW + (long)(kt + 1) * ROWB : gW;                                                     \
      _Pragma("unroll") for (int i2 = q_ * 2; i2 < q_ * 2 + 2; ++i2)                                                \
          dma_sv(w_even ? wofE[i2] : wofO[i2], wb, lds_unit(wslot, i2));                                            \
      if (q_ == 1 && last_k && has_bias) {                                                                          \
        const float* bp = p.bias + ((((ti + 1) * nblk + lbase) % tiles_n) * BN + wn * 128 + (lane & 15) * 8);       \
        asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %2, off offset:16"                 \
                     : "+v"(bq[0]), "+v"(bq[1]) : "v"(bp) : "memory");                                              \
      }                                                                                                             \
    }                                                                                                               \
    if ((q_ == 3 || q_ == 4) && !(DBG & 1)) {                                                                       \
      const char* ab = (a_even ? abE : abO) + (long)(a_cur ? kt + 2 : kt + 2 - nk) * ROWB;                          \
      constexpr int hh_ = q_ - 3;                                                                                   \
      _Pragma("unroll") for (int i2 = hh_ * ((GA + 1) / 2); i2 < (hh_ ? GA : (GA + 1) / 2); ++i2)                   \
          dma_sv(a_even ? aofE[i2] : aofO[i2], ab, lds_unit(aslot, i2));                                            \
    }                                                                                                               \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                              \
    /* the fragment registers are "used" here, in the LOAD slot: hipcc cannot see the asm wait above and would otherwise put its   \
       own s_waitcnt lgkmcnt(..) between the MFMAs of the next slot (gemm_pps.hip) */                                      \
    if (q_ == 0 || q_ == 4) touch_frag(xfr);                                                                        \
    if (q_ != 4) { asm volatile("" : "+v"(wfr[0]), "+v"(wfr[1]), "+v"(wfr[2]), "+v"(wfr[3])); }                       \
    first_load = false;                                                                                             \
    __builtin_amdgcn_sched_barrier(0);                                                                              \
  }
#define X3Q_MMA(Q)                                                                                                  \
  {                                                                                                                 \
    constexpr int half_ = ((Q) >= 2 && (Q) <= 4) ? 1 : 0;                                                           \
    __builtin_amdgcn_s_setprio(1);                                                                                  \
    if (!(DBG & 4)) {                                                                                               \
    _Pragma("unroll") for (int nb = 0; nb < 4; ++nb)                                                                \
      _Pragma("unroll") for (int jj = 0; jj < MB; ++jj)                                                             \
        acc[half_ * 4 + nb][jj] = mma16<F16>(xfr[jj], wfr[nb], acc[half_ * 4 + nb][jj]);                            \
    } else { touch_frag(xfr); }                                                                                     \
    __builtin_amdgcn_s_setprio(0);                                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                              \
  }
  // ---- NQ = 3: slot 0 = A.lo x W.hi, slot 1 = A.hi x W.hi, slot 2 = A.hi x W.lo, each over the wave's eight column blocks.  A is read in
  //      slots 0 - 1 only and W.lo last, so the ring hazards are those of the six-slot form: W_{g+1} (requested in slot 0) lands in the
  //      slot of A_{g-1}, whose last reader (waves 4-7, slot 1 of slab g - 1) is in front of that slab's third barrier; A_{g+2}
  //      (requested in slot 1) lands in the slot of W_{g-1}, whose last reader (waves 4-7, slot 2 of slab g - 1) is in front of this
  //      slab's first barrier.
#define X3W_LOAD(Q)                                                                                                 \
  {                                                                                                                 \
    constexpr int q_ = (Q);                                                                                         \
    const bool rd_ = !(DBG & 2) || first_load;                                                                      \
    if (q_ == 0 && rd_) { _Pragma("unroll") for (int jj = 0; jj < MB; ++jj) xfr[jj] = xa[jj * 128 + fragL]; }       \
    if (q_ == 1 && rd_) { _Pragma("unroll") for (int jj = 0; jj < MB; ++jj) xfr[jj] = xa[jj * 128 + fragH]; }       \
    if (q_ != 1 && rd_) {                                                                                           \
      _Pragma("unroll") for (int nb = 0; nb < 8; ++nb) wfr[nb] = wa[nb * 128 + (q_ == 2 ? fragL : fragH)];           \
    }                                                                                                               \
    if (q_ == 0 && !(DBG & 1)) {                                                                                    \
      const char* wb = w_cur ? gW + (long)(kt + 1) * ROWB : gW;                                                     \
      _Pragma("unroll") for (int i2 = 0; i2 < GW; ++i2) dma_sv(w_even ? wofE[i2] : wofO[i2], wb, lds_unit(wslot, i2)); \
      if (last_k && has_bias) {                                                                                     \
        const float* bp = p.bias + ((((ti + 1) * nblk + lbase) % tiles_n) * BN + wn * 128 + (lane & 15) * 8);       \
        asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %2, off offset:16"                 \
                     : "+v"(bq[0]), "+v"(bq[1]) : "v"(bp) : "memory");                                              \
      }                                                                                                             \
    }                                                                                                               \
    if (q_ == 1 && !(DBG & 1)) {                                                                                    \
      const char* ab = (a_even ? abE : abO) + (long)(a_cur ? kt + 2 : kt + 2 - nk) * ROWB;                          \
      _Pragma("unroll") for (int i2 = 0; i2 < GA; ++i2) dma_sv(a_even ? aofE[i2] : aofO[i2], ab, lds_unit(aslot, i2)); \
    }                                                                                                               \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                              \
    if (q_ != 2) touch_frag(xfr);                                                                                   \
    if (q_ != 1) { asm volatile("" : "+v"(wfr[0]), "+v"(wfr[1]), "+v"(wfr[2]), "+v"(wfr[3]), "+v"(wfr[4]), "+v"(wfr[5]), "+v"(wfr[6]), "+v"(wfr[7])); } \
    first_load = false;                                                                                             \
    __builtin_amdgcn_sched_barrier(0);                                                                              \
  }
#define X3W_MMA()                                                                                                   \
  {                                                                                                                 \
    __builtin_amdgcn_s_setprio(1);                                                                                  \
    if (!(DBG & 4)) {                                                                                               \
    _Pragma("unroll") for (int nb = 0; nb < 8; ++nb)                                                                \
      _Pragma("unroll") for (int jj = 0; jj < MB; ++jj)                                                             \
        acc[nb][jj] = mma16<F16>(xfr[jj], wfr[nb], acc[nb][jj]);                                                    \
    } else { touch_frag(xfr); }                                                                                     \
    __builtin_amdgcn_s_setprio(0);                                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                              \
  }
  // retire slab g: everything but the A unit requested during this slab (A_{g+2}) has landed -- W_{g+1}, A_{g+1}, the bias loads of a
  // tile's last slab and the previous tile's stores are all older (VMEM operations retire in order)
#define X3Q_RETIRE()                                                                                                \
  {                                                                                                                 \
    wait_vm<GA>();                                                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                              \
  }
#define X3Q_VARS()                                                                                                  \
  const u32x4v* xa = ldsv + sa * SLOT + xoff;                                                                       \
  const u32x4v* wa = ldsv + sw * SLOT + woff;                                                                       \
  const bool w_cur = kt + 1 < nk, a_cur = kt + 2 < nk, last_k = kt + 1 == nk;                                       \
  const bool t_even = (ti & 1) == 0, w_even = w_cur == t_even, a_even = a_cur == t_even;                            \
  const int wslot = sa + 3 >= NSLOT ? sa + 3 - NSLOT : sa + 3, aslot = sa + 4 >= NSLOT ? sa + 4 - NSLOT : sa + 4;
#define X3Q_ADVANCE()                                                                                               \
  sa = sa + 2 >= NSLOT ? sa + 2 - NSLOT : sa + 2;                                                                   \
  sw = sw + 2 >= NSLOT ? sw + 2 - NSLOT : sw + 2;                                                                   \
  if (++kt == nk) kt = 0;

  // Six barriers per slab.  Between two barriers every wave runs one LOAD slot and one MFMA slot, in opposite order for the two
  // groups (waves 0-3 multiply group q and then read group q + 1, waves 4-7 read group q and then multiply it), so on a SIMD one wave
  // multiplies while its partner reads.  Both groups retire the slab in front of the sixth barrier -- by then each has issued all of
  // the slab's requests -- and behind it waves 0-3 read the next slab.  Ring hazards: W_{g+1} lands in the slot of A_{g-1}, whose last
  // reader (waves 4-7, slot 4 of slab g - 1) is in front of that slab's sixth barrier; A_{g+2} lands in the slot of W_{g-1}, whose last
  // reader (waves 4-7, slot 5 of slab g - 1) is in front of this slab's first barrier, and its requests start in slot 3.
  if constexpr (NQ == 3) {
    if (grp == 0) {
      {
        X3Q_VARS()
        X3W_LOAD(0)
      }
      for (int g = 0; g < G; ++g) {
        X3Q_VARS()
        __builtin_amdgcn_s_barrier();
        X3W_MMA() X3W_LOAD(1) X3Q_SLOTBAR()
        X3W_MMA() X3W_LOAD(2)
        X3Q_RETIRE()
        __builtin_amdgcn_s_barrier();
        X3W_MMA()
        const bool pending = last_k;
        X3Q_ADVANCE()
        if (pending) epilogue();
        if (g + 1 < G) {
          X3Q_VARS()
          X3W_LOAD(0)
        }
      }
    } else {
      for (int g = 0; g < G; ++g) {
        X3Q_VARS()
        __builtin_amdgcn_s_barrier();
        X3W_LOAD(0) X3W_MMA() X3Q_SLOTBAR()
        X3W_LOAD(1) X3W_MMA()
        X3Q_RETIRE()
        __builtin_amdgcn_s_barrier();
        X3W_LOAD(2) X3W_MMA()
        if (last_k) epilogue();
        X3Q_ADVANCE()
      }
    }
  } else
  if (grp == 0) {
    {
      X3Q_VARS()
      X3Q_LOAD(0)
    }
    for (int g = 0; g < G; ++g) {
      X3Q_VARS()
      __builtin_amdgcn_s_barrier();
      X3Q_MMA(0) X3Q_LOAD(1) X3Q_SLOTBAR()
      X3Q_MMA(1) X3Q_LOAD(2) X3Q_SLOTBAR()
      X3Q_MMA(2) X3Q_LOAD(3) X3Q_SLOTBAR()
      X3Q_MMA(3) X3Q_LOAD(4) X3Q_SLOTBAR()
      X3Q_MMA(4) X3Q_LOAD(5)
      X3Q_RETIRE()
      __builtin_amdgcn_s_barrier();
      X3Q_MMA(5)
      const bool pending = last_k;
      X3Q_ADVANCE()
      if (pending) epilogue();
      if (g + 1 < G) {
        X3Q_VARS()
        X3Q_LOAD(0)
      }
    }
  } else {
    for (int g = 0; g < G; ++g) {
      X3Q_VARS()
      __builtin_amdgcn_s_barrier();
      X3Q_LOAD(0) X3Q_MMA(0) X3Q_SLOTBAR()
      X3Q_LOAD(1) X3Q_MMA(1) X3Q_SLOTBAR()
      X3Q_LOAD(2) X3Q_MMA(2) X3Q_SLOTBAR()
      X3Q_LOAD(3) X3Q_MMA(3) X3Q_SLOTBAR()
      X3Q_LOAD(4) X3Q_MMA(4)
      X3Q_RETIRE()
      __builtin_amdgcn_s_barrier();
      X3Q_LOAD(5) X3Q_MMA(5)
      if (last_k) epilogue();
      X3Q_ADVANCE()
    }
  }
  // the surplus requests of the stream's tail: nothing may land in LDS after the workgroup is gone.  They are older than the last
  // epilogue's MB * 8 stores, which need not be waited for.
  wait_vm<MB * 8>();
#undef X3Q_LOAD
#undef X3W_LOAD
#undef X3W_MMA
#undef X3Q_SLOTBAR
#undef X3Q_MMA
#undef X3Q_RETIRE
#undef X3Q_VARS
#undef X3Q_ADVANCE
}

template <bool F16, int BM, int OUT, int DBG = 0, int NQ = 3>
int launch_x3q_t(const GemmArgs& a, const void* packed, hipStream_t s) {
  const int tiles_m = (a.M + BM - 1) / BM, tiles_n = a.N / 256;
  const int ntiles = tiles_m * tiles_n;
  const int nblk = ntiles < g_gemm_persist_wgs ? ((ntiles + 7) / 8) * 8 : g_gemm_persist_wgs;   // (svt_debug_set key 37: workgroups of a persistent launch)
  const size_t lds_bytes = 5 * 32768;
  if (int r_ = ensure_dyn_lds((const void*)gemm_x3q_kernel<F16, BM, OUT, DBG, NQ>, (int)lds_bytes)) return r_;
  hipLaunchKernelGGL((gemm_x3q_kernel<F16, BM, OUT, DBG, NQ>), dim3(nblk), dim3(512), lds_bytes, s, a, packed, tiles_n, ntiles);
  SVT_LAUNCH_CHECK();
  return 0;
}

template <bool F16, int BM>
int launch_x3q_o(const GemmArgs& a, const void* packed, hipStream_t s) {
  if (a.planes) return launch_x3q_t<F16, BM, 2>(a, packed, s);
  if (a.c_pairs) return launch_x3q_t<F16, BM, 1>(a, packed, s);
  return launch_x3q_t<F16, BM, 0>(a, packed, s);
}

}  // namespace

bool gemm_x3q_eligible(const GemmArgs& a) {
  // a tile's rows are addressed by 32-bit offsets from its first row: 256 rows, possibly across clip boundaries
  const unsigned long clips = 255 / (unsigned long)(a.a_rpb > 0 ? a.a_rpb : 1) + 1;
  const unsigned long bs = (unsigned long)(a.a_bstride > 0 ? a.a_bstride : 0), rs = (unsigned long)(a.a_rstride > 0 ? a.a_rstride : 0);
  const unsigned long tile_span = (clips * bs + 256ul * rs + (unsigned long)a.K) * 4;
  const bool planes_ok = !a.planes || (a.act == ACT_NONE && !a.c_pairs && ((uintptr_t)a.planes & 15) == 0 && (a.plane_stride & 7) == 0 && a.ldc % 8 == 0);
  const bool pairs_ok = !a.c_pairs || (a.ldc % 32 == 0 && ((uintptr_t)a.C & 127) == 0);
  return a.a_pairs && !a.gen && a.nz == 1 && !a.resid && a.alpha == 1.f && (a.act == ACT_NONE || a.act == ACT_GELU) && a.K % 32 == 0 &&
         a.K >= 64 && a.N % 256 == 0 && a.M >= 128 && a.ldc % 4 == 0 && a.a_bstride >= 0 && a.a_rstride > 0 && a.a_rstride % 32 == 0 &&
         a.a_bstride % 32 == 0 && tile_span < 0xF0000000ul && (unsigned long)a.N * a.K * 4 < 0xF0000000ul && ((uintptr_t)a.A & 127) == 0 &&
         ((uintptr_t)a.C & 15) == 0 && ((uintptr_t)a.bias & 15) == 0 && planes_ok && pairs_ok;
}

// kind = svt_precision (2 = bf16 pieces, 3 = fp16 pieces); `packed` = the registered (hi, lo) image of the weight rows (launch_gemm_x3);
// bm = tile height (256 / 192 / 128)
int launch_gemm_x3q(int kind, const GemmArgs& a, const void* packed, int bm, hipStream_t s) {
#ifdef SVT_DIAG
  if (kind == 3 && bm == 256 && !a.planes && g_gemm_dbg >= 101 && g_gemm_dbg <= 115) {   // the same ablations on the six-slot form
    const bool pr = a.c_pairs != 0;
    switch (g_gemm_dbg - 100) {
#define SVT_X3Q_DBG(D) case D: return pr ? launch_x3q_t<true, 256, 1, D, 6>(a, packed, s) : launch_x3q_t<true, 256, 0, D, 6>(a, packed, s);
      SVT_X3Q_DBG(1) SVT_X3Q_DBG(2) SVT_X3Q_DBG(3) SVT_X3Q_DBG(4)
#undef SVT_X3Q_DBG
      default: break;
    }
  }
  if (kind == 3 && g_gemm_dbg == 100 && !a.planes) {   // six slots per slab (the first form of this kernel): A/B
    const bool pr = a.c_pairs != 0;
    if (bm == 256) return pr ? launch_x3q_t<true, 256, 1, 0, 6>(a, packed, s) : launch_x3q_t<true, 256, 0, 0, 6>(a, packed, s);
    if (bm == 192) return pr ? launch_x3q_t<true, 192, 1, 0, 6>(a, packed, s) : launch_x3q_t<true, 192, 0, 0, 6>(a, packed, s);
  }
  if (kind == 3 && bm == 256 && !a.planes && g_gemm_dbg >= 1 && g_gemm_dbg <= 15) {   // timing ablations of the fp16, 256-row form (wrong results)
    const bool pr = a.c_pairs != 0;
    switch (g_gemm_dbg) {
#define SVT_X3Q_DBG(D) case D: return pr ? launch_x3q_t<true, 256, 1, D>(a, packed, s) : launch_x3q_t<true, 256, 0, D>(a, packed, s);
      SVT_X3Q_DBG(1) SVT_X3Q_DBG(2) SVT_X3Q_DBG(3) SVT_X3Q_DBG(4) SVT_X3Q_DBG(6) SVT_X3Q_DBG(7) SVT_X3Q_DBG(8) SVT_X3Q_DBG(10)
#undef SVT_X3Q_DBG
      default: break;
    }
  }
#endif
  if (kind == 3) {
    if (bm == 256) return launch_x3q_o<true, 256>(a, packed, s);
    if (bm == 192) return launch_x3q_o<true, 192>(a, packed, s);
    return launch_x3q_o<true, 128>(a, packed, s);
  }
  if (bm == 256) return launch_x3q_o<false, 256>(a, packed, s);
  if (bm == 192) return launch_x3q_o<false, 192>(a, packed, s);
  return launch_x3q_o<false, 128>(a, packed, s);
}

}  // namespace svt
#endif  // SVT_OPERAND_F16

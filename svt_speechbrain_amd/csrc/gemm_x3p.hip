// Split-operand products (precision "fp16x3" / "bf16x3") as a PERSISTENT, STAGGERED stream: the schedule of gemm_pps.hip with the
// operands of gemm_x3_kernel (gemm_dma.hip).
//
// Why: the one-tile kernels end every tile with 256 KiB of fp32 through LDS transpose patches, the exact-erf GELU of 65 536 values
// and the store burst of all 256 CUs at once, with the matrix pipes idle; here the epilogue runs out of the accumulators (no LDS),
// stores whole 256-byte row segments, and its stores drain under the next tile's MFMAs.  Measured (fp16x3, conv1 512k x 512 x 1536):
// 2 227 us one tile per workgroup, 2 162 us here, 2 005 us with neither stores nor GELU -- i.e. the epilogue is 7 % of this kernel and
// the slab loop (2.6 us per 256 x 256 x 32 slab against 1.5 us of MFMA issue) is what bounds the split modes.
//   * tile 256 x 256, K in 32-element slabs; A = fp32 rows (256 x 128 B per unit), W = the pre-cut (hi, lo) pieces (256 x 128 B per
//     unit), ring of five 32 KiB slots: unit u = 2g (A_g) / 2g + 1 (W_g) in slot u % 5;
//   * 8 waves as 8 (M) x 1 (N): a wave owns 32 rows x 256 columns; while slab g multiplies it reads ITS fp32 rows of A_{g+1} (LOAD slot
//     0) and cuts them into (hi, lo) pieces (slots 1 and 2) for the next slab, exactly as gemm_x3s_kernel does.  (A first version cut
//     A_g at the head of slab g to save the second set of piece registers: raw reads + ~90 conversion instructions made LOAD slot 0
//     twice as long as the partner's 24 MFMAs, 2.5 instead of 1.5 us per slab.)  Slab g requests A_{g+2} (slots 0 - 1, into the slot
//     W_{g-1} left) and W_{g+2} (slots 2 - 3, into the slot of A_g, which has lived in registers since slab g - 1); the counted wait
//     that retires the slab leaves W_{g+2} in flight;
//   * the MFMA takes the ACTIVATION pieces as src0 (rows on 4 (lane >> 4) + r) and the W rows of a 64-column group are placed in LDS
//     so that (block nb, row j) is output column 64 (nb >> 2) + 4 j + (nb & 3): a lane holds four consecutive columns per group and
//     one buffer_store_dwordx4 writes four rows x 256 contiguous bytes (fp32), or 128 contiguous bytes per plane and row for the
//     (hi, lo) plane output of the QKV projection;
//   * staggering, tile boundaries, unconditional ring requests, the bias fetched inside the stream and the vector-offset stores
//     (soffset hazard) are those of gemm_pps_kernel.
// Contract: a plain product against a registered split weight matrix (launch_gemm_x3 resolves it), N % 256 == 0, K % 32 == 0, K >= 64,
// no residual, activation none or GELU, fp32 C or (hi, lo) planes.  Everything else stays on gemm_x3s_kernel.
#include "common.h"

#ifdef SVT_OPERAND_F16
// The split-operand engines live in the bf16 build only (libsvt_mi355.so): precision codes 2 / 3 are rejected by the IEEE-half build.
namespace svt {
bool gemm_x3p_eligible(const GemmArgs&) { return false; }
int launch_gemm_x3p(int, const GemmArgs&, const void*, hipStream_t) { set_error("gemm_x3p: not part of the IEEE-half build"); return -1; }
}  // namespace svt
#else

namespace svt {
namespace {

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void dma_sv(unsigned voff, const void* sbase, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_addr), "v"(voff), "s"(sbase) : "memory");
}
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
typedef unsigned u32x2v __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8v __attribute__((ext_vector_type(8)));

template <bool F16> __device__ __forceinline__ f32x4 mma3(const u32x4v& a, const u32x4v& b, const f32x4& c) {
  if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8v, a), __builtin_bit_cast(f16x8v, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(real_bf16x8, a), __builtin_bit_cast(real_bf16x8, b), c, 0, 0, 0);
}
// 8 fp32 of one lane (two 16-byte chunks) -> (hi, lo) 16-bit pieces
template <bool F16> __device__ __forceinline__ void cut8(const u32x4v& r0, const u32x4v& r1, u32x4v& hi, u32x4v& lo) {
  const f32x4 v0 = __builtin_bit_cast(f32x4, r0), v1 = __builtin_bit_cast(f32x4, r1);
  if constexpr (F16) {
    f16x8v h, l;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      h[j] = (_Float16)v0[j]; l[j] = (_Float16)(v0[j] - (float)h[j]);
      h[4 + j] = (_Float16)v1[j]; l[4 + j] = (_Float16)(v1[j] - (float)h[4 + j]);
    }
    hi = __builtin_bit_cast(u32x4v, h);
    lo = __builtin_bit_cast(u32x4v, l);
  } else {
    real_bf16x8 h, l;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      h[j] = (__bf16)v0[j]; l[j] = (__bf16)(v0[j] - (float)h[j]);
      h[4 + j] = (__bf16)v1[j]; l[4 + j] = (__bf16)(v1[j] - (float)h[4 + j]);
    }
    hi = __builtin_bit_cast(u32x4v, h);
    lo = __builtin_bit_cast(u32x4v, l);
  }
}
template <bool F16> __device__ __forceinline__ void cut1(float x, unsigned short& hi, unsigned short& lo) {
  if constexpr (F16) {
    const _Float16 a = (_Float16)x, b = (_Float16)(x - (float)a);
    hi = __builtin_bit_cast(unsigned short, a); lo = __builtin_bit_cast(unsigned short, b);
  } else {
    const __bf16 a = (__bf16)x, b = (__bf16)(x - (float)a);
    hi = __builtin_bit_cast(unsigned short, a); lo = __builtin_bit_cast(unsigned short, b);
  }
}

// (the activation is a run-time flag, not a template parameter: the instantiation WITHOUT GELU was the one hipcc could not keep
// inside 256 registers -- seven spilled dwords whose reloads put s_waitcnt vmcnt(0) between the MFMAs of every slab)
// STAMP 1..4 (tools/gemm_trace.py --x3-slots): s_memtime at both ends of slots 2 (STAMP - 1), 2 (STAMP - 1) + 1 of one slab, per wave
template <bool F16, bool PLANES, int STAMP = 0>
__global__ __launch_bounds__(512) void gemm_x3p_kernel(GemmArgs p, const void* wsplit, int tiles_n, int ntiles) {
  constexpr int BM = 256, BN = 256, BK = 32, NSLOT = 5, SLOT = 2048, GA = 4, GW = 4, NB = 16;
  constexpr int NST = PLANES ? 64 : 32;   // stores per wave and tile
  extern __shared__ __attribute__((aligned(16))) uint4 lds[];
  const u32x4v* ldsv = (const u32x4v*)lds;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nblk = gridDim.x, b = blockIdx.x;
  const int per = nblk >> 3;
  const int lbase = (b & 7) * per + (b >> 3);
  if (lbase >= ntiles) return;
  const int my_tiles = (ntiles - lbase + nblk - 1) / nblk;

  const char* gW = (const char*)wsplit;   // [N][K / 32][64 x 16 bit]: a row of the packed matrix is 4 K bytes
  // A rows: a 64-bit base per tile (its first row; the whole tensor may exceed 4 GiB) + 32-bit offsets of this lane's rows from it
  auto a_row_off = [&](int m) -> long { return ((long)(m / p.a_rpb) * p.a_bstride + (long)(m % p.a_rpb) * p.a_rstride) * 4; };
  const char* abase;
  const char* abase2;
  unsigned aof[GA], wof[GW], aof2[GA], wof2[GW];
  auto setup = [&](int logical, const char*& ab, unsigned (&ao)[GA], unsigned (&wo)[GW]) {
    int ln;   // the lane index, recomputed in place (volatile asm: not merged with an earlier copy that would then live across the slabs)
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));   // see the epilogue: keeps these addresses out of the loop-invariant set
    const int r8 = ln >> 3, ch = (ln & 7) ^ r8;
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
      const int rho = (wave + 8 * i) * 8 + r8;   // LDS row of the W unit: (block nb = rho >> 4, row j = rho & 15) <- column 64 (nb >> 2) + 4 j + (nb & 3)
      const int nb = rho >> 4, j = rho & 15;
      const int n = n0 + (nb >> 2) * 64 + j * 4 + (nb & 3);   // < N: N % 256 == 0
      wo[i] = (unsigned)((long)n * p.K * 4 + ch * 16);
    }
  };
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(void __attribute__((address_space(3)))*)lds);
  auto lds_unit = [&](int slot, int i) -> unsigned { return lds0 + (unsigned)(slot * SLOT + (wave + 8 * i) * 64) * 16u; };

  f32x4 acc[NB][2];
#pragma unroll
  for (int i = 0; i < NB; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int cq = lane >> 4, r16 = lane & 15, rr8 = r16 & 7;
  const int rowb = (r16 >> 3) * 64 + rr8 * 8;
  const int fa0 = rowb + ((2 * cq) ^ rr8), fa1 = rowb + ((2 * cq + 1) ^ rr8);   // fp32 k = 8 cq .. + 7: chunks 2 cq, 2 cq + 1
  const int fwh = rowb + (cq ^ rr8), fwl = rowb + ((4 + cq) ^ rr8);             // hi pieces chunk cq, lo pieces chunk 4 + cq
  const int xoff = (wave * 2) * 128;

  const int nk = p.K / BK;             // >= 2 (launcher)
  const int G = my_tiles * nk;
  const bool has_bias = p.bias != nullptr;
  const bool no_store = p.dbg == 3 || p.dbg == 7;   // diagnostics (svt_debug_set key 0: 3 / 7 = no stores, 5 / 7 = no activation)
  const bool do_gelu = p.act == ACT_GELU && p.dbg != 5 && p.dbg != 7;
  setup(lbase, abase, aof, wof);
  setup(my_tiles > 1 ? nblk + lbase : lbase, abase2, aof2, wof2);
  // head of the stream: A_0 -> slot 0, W_0 -> slot 1, A_1 -> slot 2, W_1 -> slot 3; everything but W_1 landed before slab 0
#pragma unroll
  for (int i = 0; i < GA; ++i) dma_sv(aof[i], abase, lds_unit(0, i));
#pragma unroll
  for (int i = 0; i < GW; ++i) dma_sv(wof[i], gW, lds_unit(1, i));
#pragma unroll
  for (int i = 0; i < GA; ++i) dma_sv(aof[i], abase + BK * 4, lds_unit(2, i));
#pragma unroll
  for (int i = 0; i < GW; ++i) dma_sv(wof[i], gW + 128, lds_unit(3, i));
  wait_vm<GW>();
  __builtin_amdgcn_s_barrier();

  u32x4v xh[2], xl[2], nh[2], nl[2], raw[2], wh[4], wl[4];
#pragma unroll
  for (int mb = 0; mb < 2; ++mb) cut8<F16>(ldsv[xoff + mb * 128 + fa0], ldsv[xoff + mb * 128 + fa1], xh[mb], xl[mb]);   // this wave's rows of A_0
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  // A COMPILER-VISIBLE vmcnt(0) (once per kernel; it also lets W_1 land): if the register allocator spilled anything up to here, its
  // scratch reloads count as pending VMEM loads for hipcc's wait insertion, which then guards the first use of every reloaded
  // value INSIDE the slab loop with s_waitcnt vmcnt(...) -- seen as vmcnt(5) / vmcnt(0) between the MFMAs of slot 0, i.e. the ring
  // drained once per slab.  After this instruction the pass knows nothing of its own is outstanding.
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __builtin_amdgcn_s_barrier();   // every wave holds its pieces of A_0 before slab 0's W_2 request reuses that slot
  int sw = 1, kt = 0, ti = 0;     // slot of W_g (A_{g+1} sits in the next one)
  const int grp = wave >> 2;
  // slot stamps of slab gs (the middle of the second tile): stamp k = 2 * slot (start) / 2 * slot + 1 (arrival at the barrier)
  unsigned sraw[5], st[5];
  const int gs = my_tiles > 1 ? nk + nk / 2 : nk / 2;
  if constexpr (STAMP != 0) {
#pragma unroll
    for (int i = 0; i < 5; ++i) st[i] = 0;
  }
#define X3P_S(k)                                                                                                    \
  if constexpr (STAMP != 0 && (k) >= 4 * (STAMP - 1) && (k) <= 4 * (STAMP - 1) + 4 && (k) < 16)                     \
    sraw[((k) - 4 * (STAMP - 1)) % 5] = (unsigned)__builtin_amdgcn_s_memtime();                                      \
  if constexpr (STAMP == 4 && (k) == 0) sraw[4] = (unsigned)__builtin_amdgcn_s_memtime();
#define X3P_COMMIT()                                                                                                \
  if constexpr (STAMP != 0) {                                                                                       \
    if (g == gs) { _Pragma("unroll") for (int i = 0; i < (STAMP == 4 ? 4 : 5); ++i) st[i] = sraw[i]; }               \
    if (STAMP == 4 && g == gs + 1) st[4] = sraw[4];                                                                  \
  }

  auto epilogue = [&]() {
    const int logical = ti * nblk + lbase;
    const int tile_n = logical % tiles_n, tile_m = logical / tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const long rows_left = (long)p.M - m0;
    const int esz = PLANES ? 2 : 4;
    const unsigned long nbytes = (unsigned long)rows_left * p.ldc * esz;
    const unsigned nrec = nbytes > 0xFFFFFFF0ul ? 0xFFFFFFF0u : (unsigned)nbytes;
    // this lane: rows 4 (lane >> 4) + r of the wave's two 16-row blocks, columns 64 s + 4 (lane & 15) .. + 3 of every 64-column group s
    // (the lane index is laundered through an empty asm: hipcc otherwise hoists every lane-derived address of the epilogue and of
    // setup() out of the slab loop as loop invariants -- ~25 VGPRs that live across the MFMA slots -- and spills in exchange)
    int ln;   // the lane index, recomputed in place (volatile asm: not merged with an earlier copy that would then live across the slabs)
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
    const unsigned off0 = (unsigned)((((long)(wave * 32 + 4 * (ln >> 4))) * p.ldc + n0 + (ln & 15) * 4) * esz);
    const unsigned row_pitch = (unsigned)(p.ldc * esz);
    // bias of this lane's 16 columns (4 per 64-column group), fetched HERE: sixteen registers that live across the whole tile were the
    // difference between 256 registers and spills in the slab loop.  The wait for these loads also covers the eight ring requests of
    // the tile's last slab (VMEM operations retire in order): they would have had the epilogue's ~1 us to land anyway.
    const f32x4* bp = (const f32x4*)(p.bias + n0 + (ln & 15) * 4);
    f32x4 bq[4];
    if constexpr (PLANES) {
#pragma unroll
      for (int j = 0; j < 4; ++j) bq[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (has_bias) {
#pragma unroll
        for (int j = 0; j < 4; ++j) bq[j] = bp[j * 16];
      }
    }
    if constexpr (PLANES) {
      char* hbase = (char*)p.planes + (long)m0 * p.ldc * 2;
      const auto hrsrc = __builtin_amdgcn_make_buffer_rsrc(hbase, 0, nrec, 0x00020000);
      const auto lrsrc = __builtin_amdgcn_make_buffer_rsrc(hbase + p.plane_stride * 2, 0, nrec, 0x00020000);
#pragma unroll
      for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int s4 = 0; s4 < 4; ++s4) {
            unsigned short hi[4], lo[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) cut1<F16>(acc[s4 * 4 + c][mb][r] + bq[s4][c], hi[c], lo[c]);
            const u32x2v hv = {(unsigned)hi[0] | ((unsigned)hi[1] << 16), (unsigned)hi[2] | ((unsigned)hi[3] << 16)};
            const u32x2v lv = {(unsigned)lo[0] | ((unsigned)lo[1] << 16), (unsigned)lo[2] | ((unsigned)lo[3] << 16)};
            const unsigned off = off0 + (mb * 16 + r) * row_pitch + s4 * 128;
            __builtin_amdgcn_raw_buffer_store_b64(hv, hrsrc, off, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b64(lv, lrsrc, off, 0, 0);
          }
    } else {
      char* cbase = (char*)p.C + (long)m0 * p.ldc * 4;
      const auto crsrc = __builtin_amdgcn_make_buffer_rsrc(cbase, 0, nrec, 0x00020000);
      // one 64-column group at a time: its four bias values are the only ones live (all sixteen at once cost the four registers
      // that made the four-barrier loop spill -- and a spill's reload makes hipcc guard the loop's ring requests with vmcnt waits)
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        f32x4 bs = f32x4{0.f, 0.f, 0.f, 0.f};
        if (has_bias) bs = bp[s4 * 16];
        asm volatile("" : "+v"(bs));   // fetched here, not hoisted in front of the first group
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            f32x4 v;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              const float x = acc[s4 * 4 + c][mb][r] + bs[c];
              v[c] = do_gelu ? gelu_fast(x) : x;
            }
            if (no_store) { asm volatile("" ::"v"(v)); continue; }
            // row and column group in the VECTOR offset (range check; soffset store-data hazard: gemm_pps.hip)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, v), crsrc, off0 + (mb * 16 + r) * row_pitch + s4 * 256, 0, 16);
          }
      }
    }
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    ++ti;
    abase = abase2;
#pragma unroll
    for (int i = 0; i < GA; ++i) aof[i] = aof2[i];
#pragma unroll
    for (int i = 0; i < GW; ++i) wof[i] = wof2[i];
    if (ti + 1 < my_tiles) setup((ti + 1) * nblk + lbase, abase2, aof2, wof2);   // else: keeps the last tile's rows (surplus requests)
  };

  // LOAD slot Q of slab g: W pieces of blocks 4 Q .. 4 Q + 3 of W_g; slots 0 / 2 also read this wave's fp32 rows of A_{g+1} (one 16-row
  // block each), slots 1 / 3 cut them; requests: A_{g+2} in slots 0 - 1, W_{g+2} in slots 2 - 3 (two instructions each)
  // the ring requests of slot Q: A_{g+2} in slots 0 - 1 (into the slot W_{g-1} left), W_{g+2} in slots 2 - 3 (into the slot of A_g)
#define X3P_DMA(Q)                                                                                                  \
  {                                                                                                                 \
    if ((Q) < 2) {                                                                                                  \
      const char* ab = (a_cur ? abase : abase2) + (long)(a_cur ? kt + 2 : kt + 2 - nk) * (BK * 4);                  \
      _Pragma("unroll") for (int i2 = (Q) * 2; i2 < (Q) * 2 + 2; ++i2)                                              \
          dma_sv(a_cur ? aof[i2] : aof2[i2], ab, lds_unit(s3, i2));                                                 \
    } else {                                                                                                        \
      const char* wb = gW + (long)(a_cur ? kt + 2 : kt + 2 - nk) * 128;                                             \
      _Pragma("unroll") for (int i2 = ((Q) - 2) * 2; i2 < ((Q) - 2) * 2 + 2; ++i2)                                  \
          dma_sv(a_cur ? wof[i2] : wof2[i2], wb, lds_unit(s4, i2));                                                 \
    }                                                                                                               \
  }
#define X3P_LOAD(Q) X3P_LOAD_(Q, true)
#define X3P_LOAD_(Q, WITH_DMA)                                                                                     \
  {                                                                                                                 \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                                 \
      wh[i] = wa[((Q) * 4 + i) * 128 + fwh];                                                                        \
      wl[i] = wa[((Q) * 4 + i) * 128 + fwl];                                                                        \
    }                                                                                                               \
    /* the fp32 rows of block 0 / 1 of A_{g+1} are read in slots 0 / 2 and cut in slots 1 / 3 (one 8-register staging set).  Inside \
       a slot EVERY LDS read and both ring requests are issued first and the cut runs behind them, under the LDS latency: while   \
       the partner wave issues its 24 MFMAs this wave's vector instructions get every other issue slot, so the 18 instructions   \
       of a cut take ~300 cycles -- in front of the reads (where hipcc's scheduler had put them) they made the slot 640 cycles   \
       long against the partner's 384 (tools/gemm_trace.py --x3-slots) */                                                    \
    if ((Q) == 0 || (Q) == 2) { raw[0] = xn[((Q) >> 1) * 128 + fa0]; raw[1] = xn[((Q) >> 1) * 128 + fa1]; }         \
    if (WITH_DMA) X3P_DMA(Q)                                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                                              \
    /* (the pieces are "used" here by an empty asm: LLVM otherwise sinks both cuts to their only real use, the copy in the loop  \
       latch -- behind the slab's last barrier, on the critical path of every slab: gemm_x3s_kernel, gemm_dma.hip) */     \
    if ((Q) == 1) { cut8<F16>(raw[0], raw[1], nh[0], nl[0]); asm volatile("" : "+v"(nh[0]), "+v"(nl[0])); }         \
    if ((Q) == 3) { cut8<F16>(raw[0], raw[1], nh[1], nl[1]); asm volatile("" : "+v"(nh[1]), "+v"(nl[1])); }         \
    __builtin_amdgcn_sched_barrier(0);                                                                              \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                              \
    asm volatile("" : "+v"(wh[0]), "+v"(wh[1]), "+v"(wh[2]), "+v"(wh[3]), "+v"(wl[0]), "+v"(wl[1]), "+v"(wl[2]), "+v"(wl[3])); \
    if ((Q) == 0 || (Q) == 2) asm volatile("" : "+v"(raw[0]), "+v"(raw[1]));                                        \
    __builtin_amdgcn_sched_barrier(0);                                                                              \
  }
#define X3P_MMA(Q)                                                                                                  \
  {                                                                                                                 \
    __builtin_amdgcn_s_setprio(1);                                                                                  \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                                 \
      _Pragma("unroll") for (int mb = 0; mb < 2; ++mb) acc[(Q) * 4 + i][mb] = mma3<F16>(xh[mb], wl[i], acc[(Q) * 4 + i][mb]); \
      _Pragma("unroll") for (int mb = 0; mb < 2; ++mb) acc[(Q) * 4 + i][mb] = mma3<F16>(xl[mb], wh[i], acc[(Q) * 4 + i][mb]); \
      _Pragma("unroll") for (int mb = 0; mb < 2; ++mb) acc[(Q) * 4 + i][mb] = mma3<F16>(xh[mb], wh[i], acc[(Q) * 4 + i][mb]); \
    }                                                                                                               \
    __builtin_amdgcn_s_setprio(0);                                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                              \
  }
#define X3P_VARS()                                                                                                  \
  const u32x4v* wa = ldsv + sw * SLOT;                                                                              \
  const int s1 = sw + 1 >= NSLOT ? sw + 1 - NSLOT : sw + 1, s3 = sw + 3 >= NSLOT ? sw + 3 - NSLOT : sw + 3,         \
            s4 = sw + 4 >= NSLOT ? sw + 4 - NSLOT : sw + 4;   /* (% NSLOT costs an s_mul_hi sequence per use, per slab) */ \
  const u32x4v* xn = ldsv + s1 * SLOT + xoff;                                                                       \
  const bool a_cur = kt + 2 < nk, last_k = kt + 1 == nk;
  // everything but W_{g+2} (the four youngest requests) has landed: W_{g+1}, A_{g+2}, and the previous tile's stores
#define X3P_RETIRE()                                                                                                \
  {                                                                                                                 \
    wait_vm<GW>();                                                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                              \
  }
#define X3P_ADVANCE()                                                                                               \
  _Pragma("unroll") for (int mb = 0; mb < 2; ++mb) { xh[mb] = nh[mb]; xl[mb] = nl[mb]; }                            \
  sw = sw + 2 >= NSLOT ? sw + 2 - NSLOT : sw + 2;                                                                   \
  if (++kt == nk) kt = 0;

  // (A four-barrier form of this loop -- the schedule of gemm_pps_kernel, with slot 0's ring requests held back behind the next barrier
  //  because A_{g+2} lands in the slot waves 4-7 still read W_{g-1} from -- was built and measured 1-2 % slower than the eight-barrier
  //  form below on the same box (fp16x3 C2 2 194-2 213 against 2 231-2 243 clips/s): removed.)
  if (grp == 0) {
    bool pending = false;
    for (int g = 0;; ++g) {
      if (pending) epilogue();
      if (g == G) break;
      X3P_VARS()
      X3P_S(0) X3P_LOAD(0) X3P_S(1) __builtin_amdgcn_s_barrier(); X3P_S(2) X3P_MMA(0) X3P_S(3) __builtin_amdgcn_s_barrier();
      X3P_S(4) X3P_LOAD(1) X3P_S(5) __builtin_amdgcn_s_barrier(); X3P_S(6) X3P_MMA(1) X3P_S(7) __builtin_amdgcn_s_barrier();
      X3P_S(8) X3P_LOAD(2) X3P_S(9) __builtin_amdgcn_s_barrier(); X3P_S(10) X3P_MMA(2) X3P_S(11) __builtin_amdgcn_s_barrier();
      X3P_S(12) X3P_LOAD(3) X3P_S(13) __builtin_amdgcn_s_barrier(); X3P_S(14) X3P_MMA(3)
      X3P_RETIRE()
      X3P_S(15)
      __builtin_amdgcn_s_barrier();
      X3P_COMMIT()
      pending = last_k;
      X3P_ADVANCE()
    }
  } else {
    __builtin_amdgcn_s_barrier();  // one slot behind
    for (int g = 0; g < G; ++g) {
      X3P_VARS()
      X3P_S(0) X3P_LOAD(0) X3P_S(1) __builtin_amdgcn_s_barrier(); X3P_S(2) X3P_MMA(0) X3P_S(3) __builtin_amdgcn_s_barrier();
      X3P_S(4) X3P_LOAD(1) X3P_S(5) __builtin_amdgcn_s_barrier(); X3P_S(6) X3P_MMA(1) X3P_S(7) __builtin_amdgcn_s_barrier();
      X3P_S(8) X3P_LOAD(2) X3P_S(9) __builtin_amdgcn_s_barrier(); X3P_S(10) X3P_MMA(2) X3P_S(11) __builtin_amdgcn_s_barrier();
      X3P_S(12) X3P_LOAD(3)
      X3P_RETIRE()
      X3P_S(13)
      __builtin_amdgcn_s_barrier();
      X3P_S(14) X3P_MMA(3)
      if (last_k) epilogue();
      X3P_S(15)
      if (g + 1 < G) __builtin_amdgcn_s_barrier();
      X3P_COMMIT()
      X3P_ADVANCE()
    }
  }
  // the surplus requests of the stream's tail must have landed before the workgroup gives its LDS back; they are older than the last
  // epilogue's stores
  if constexpr (NST <= 32) wait_vm<NST>(); else wait_vm<0>();
  if constexpr (STAMP != 0) {
    if (p.trace && lane == 0) {
      long long* o = p.trace + 65536 + ((long)blockIdx.x * 8 + wave) * 32;
#pragma unroll
      for (int i = 0; i < 5; ++i) o[i] = st[i];
      o[9] = G; o[10] = gs; o[11] = STAMP;
    }
  }
#undef X3P_S
#undef X3P_COMMIT
#undef X3P_LOAD
#undef X3P_LOAD_
#undef X3P_DMA
#undef X3P_MMA
#undef X3P_VARS
#undef X3P_RETIRE
#undef X3P_ADVANCE
}

template <bool F16, bool PLANES, int STAMP = 0>
int launch_x3p_t(const GemmArgs& a, const void* packed, hipStream_t s) {
  const int tiles_m = (a.M + 255) / 256, tiles_n = a.N / 256;
  const int ntiles = tiles_m * tiles_n;
  const int nblk = ntiles < 256 ? ((ntiles + 7) / 8) * 8 : 256;
  const size_t lds_bytes = 5 * 32768;
  if (int r_ = ensure_dyn_lds((const void*)gemm_x3p_kernel<F16, PLANES, STAMP>, (int)lds_bytes)) return r_;
  hipLaunchKernelGGL((gemm_x3p_kernel<F16, PLANES, STAMP>), dim3(nblk), dim3(512), lds_bytes, s, a, packed, tiles_n, ntiles);
  SVT_LAUNCH_CHECK();
  return 0;
}

}  // namespace

bool gemm_x3p_eligible(const GemmArgs& a) {
  // a tile's rows are addressed by 32-bit offsets from its first row: 256 rows, possibly across clip boundaries
  const unsigned long clips = 255 / (unsigned long)(a.a_rpb > 0 ? a.a_rpb : 1) + 1;
  const unsigned long bs = (unsigned long)(a.a_bstride > 0 ? a.a_bstride : 0), rs = (unsigned long)(a.a_rstride > 0 ? a.a_rstride : 0);
  const unsigned long tile_span = (clips * bs + 256ul * rs + (unsigned long)a.K) * 4;
  return !a.gen && a.nz == 1 && !a.resid && a.alpha == 1.f && (a.act == ACT_NONE || a.act == ACT_GELU) && !(a.planes && a.act != ACT_NONE) &&
         a.K % 32 == 0 && a.K >= 64 && a.N % 256 == 0 && a.M >= 128 && a.c_vec && a.ldc % 4 == 0 && a.a_bstride >= 0 && a.a_rstride > 0 &&
         tile_span < 0xF0000000ul && (unsigned long)a.N * a.K * 4 < 0xF0000000ul && ((uintptr_t)a.A & 15) == 0 && (a.a_rstride & 3) == 0 &&
         (a.a_bstride & 3) == 0 && (!a.planes || (((uintptr_t)a.planes & 7) == 0 && (a.plane_stride & 3) == 0));
}

// kind = svt_precision (2 = bf16 pieces, 3 = fp16 pieces); `packed` = the registered (hi, lo) image of the weight rows (launch_gemm_x3)
int launch_gemm_x3p(int kind, const GemmArgs& a, const void* packed, hipStream_t s) {
  const bool f16 = kind == 3;
#ifdef SVT_DIAG
  if (a.trace && f16 && !a.planes) {   // slot stamps (diagnostics; make DIAG=1)
    switch (a.stamp_ends) {
      case 1: return launch_x3p_t<true, false, 1>(a, packed, s);
      case 2: return launch_x3p_t<true, false, 2>(a, packed, s);
      case 3: return launch_x3p_t<true, false, 3>(a, packed, s);
      case 4: return launch_x3p_t<true, false, 4>(a, packed, s);
      default: break;
    }
  }
#endif
  if (a.planes) return f16 ? launch_x3p_t<true, true>(a, packed, s) : launch_x3p_t<false, true>(a, packed, s);
  return f16 ? launch_x3p_t<true, false>(a, packed, s) : launch_x3p_t<false, false>(a, packed, s);
}

}  // namespace svt
#endif  // SVT_OPERAND_F16

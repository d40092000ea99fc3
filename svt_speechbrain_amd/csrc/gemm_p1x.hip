// Split-operand products (precision "fp16x3" / "bf16x3") on pair rows with ONE wave per SIMD: gemm_p1w_kernel's loop (gemm_p1w.hip) with
// gemm_x3q_kernel's arithmetic (gemm_x3q.hip).  Round 5.
//
// A pair-row slab is the LDS image of a 64-deep 16-bit slab: 128 bytes per row, hi pieces of 32 consecutive elements in the first half, lo
// pieces in the second.  Where the 16-bit kernel multiplies (k-step 0 x k-step 0) + (k-step 1 x k-step 1), this one multiplies
// lo x hi + hi x hi + hi x lo -- in that order per accumulator, from the bias as the first MFMA's C operand: the order of gemm_x3q_kernel, so
// the two kernels give the same bits (tests/test_gpu_gemm.py compares them).  Per slab and wave: three PHASES of 8 MB MFMAs (MB = BM / 32
// row blocks x 8 column blocks of the wave's BM/2 x 128 tile), all four fragment sets live (x hi, x lo, w hi, w lo: 64 + 64 registers at
// BM = 256, beside 256 accumulator registers in the AGPR half of the file):
//     phase 0   acc += x.lo w.hi     meanwhile: read x.hi of this slab;           request A of slab g + 2
//     phase 1   acc += x.hi w.hi     meanwhile: read w.lo of this slab
//     -- barrier B_g: this wave's pieces of slab g + 1 have landed (counted vmcnt), all its reads of slab g are complete --
//     phase 2   acc += x.hi w.lo     meanwhile: read x.lo and w.hi of slab g + 1;  request W of slab g + 2 (into A_g's slot)
// i.e. gemm_p1w_kernel's slab with its first k-step doubled: the same reads, requests and ONE barrier against 1.5 x the MFMAs, which is
// why the split modes sit closer to the matrix pipe than the 16-bit ones.  Ring (five 32 KiB slots, unit u in slot u mod 5, XOR-swizzled
// 16-byte chunks, W rows permuted so that a lane ends with 8 consecutive output columns), persistent tile stream, 64-bit A base per tile
// parity with 32-bit row offsets, the scalar distance between a wave's W pieces: gemm_p1w.hip / gemm_x3q.hip.  The epilogue (fp32 rows /
// pair rows / (hi, lo) planes; bias inside; exact-erf GELU) runs between two tiles, out of the accumulators, one row block at a time behind
// an AGPR anchor; the next tile's A requests are issued in front of its stores so that the next barrier's counted wait leaves them in flight.
// Contract = gemm_x3q_eligible and K >= 96 (three slabs per tile: a tile boundary never asks for rows of the tile after next).
#include "common.h"

// An A/B arm, not a dispatched kernel: built by `make DIAG=1` only (the shipped library holds what it dispatches: VERDICT r05 #11)
#if defined(SVT_OPERAND_F16) || !defined(SVT_DIAG)
namespace svt {
int g_gemm_p1x = 0;
int launch_gemm_p1x(int, const GemmArgs&, const void*, int, hipStream_t) { set_error("gemm_p1x: built by `make DIAG=1` (bf16 library) only"); return -1; }
}  // namespace svt
#else

namespace svt {
namespace {

template <int N> __device__ __forceinline__ void px_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void px_dma(unsigned voff, const void* sbase, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_addr), "v"(voff), "s"(sbase) : "memory");
}
typedef unsigned px_u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 px_f16x8 __attribute__((ext_vector_type(8)));

template <bool F16> __device__ __forceinline__ f32x4 px_mma(const px_u32x4& a, const px_u32x4& b, const f32x4& c) {
  if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(px_f16x8, a), __builtin_bit_cast(px_f16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(real_bf16x8, a), __builtin_bit_cast(real_bf16x8, b), c, 0, 0, 0);
}
// eight fp32 values -> packed (hi, lo) 16-bit pieces (gemm_x3q.hip)
template <bool F16> __device__ __forceinline__ void px_cut8(const float (&v)[8], px_u32x4& hi, px_u32x4& lo) {
  if constexpr (F16) {
    px_f16x8 h, l;
#pragma unroll
    for (int j = 0; j < 8; ++j) { h[j] = (_Float16)v[j]; l[j] = (_Float16)(v[j] - (float)h[j]); }
    hi = __builtin_bit_cast(px_u32x4, h);
    lo = __builtin_bit_cast(px_u32x4, l);
  } else {
    real_bf16x8 h, l;
#pragma unroll
    for (int j = 0; j < 8; ++j) { h[j] = (__bf16)v[j]; l[j] = (__bf16)(v[j] - (float)h[j]); }
    hi = __builtin_bit_cast(px_u32x4, h);
    lo = __builtin_bit_cast(px_u32x4, l);
  }
}

template <bool F16, int BM, int OUT>   // OUT: 0 = fp32 rows, 1 = pair rows, 2 = separate (hi, lo) planes
__global__ __launch_bounds__(256) void gemm_p1x_kernel(GemmArgs p, const void* wsplit, int tiles_n, int ntiles) {
  constexpr int BN = 256, BK = 32, NSLOT = 5;
  constexpr int MB = BM / 32;        // 16-row blocks per wave (wave tile BM/2 x 128)
  constexpr int GA = BM / 32;        // A pieces (8 rows x 128 B) per wave and slab
  constexpr int GW = BN / 32;        // W pieces per wave and slab
  constexpr int SLOT = 2048;         // uint4 per ring slot (32 KiB)
  constexpr int ROWB = BK * 4;       // bytes of a pair-row slab
  constexpr int MPG = MB / 2;        // MFMAs per group: a phase is 16 groups of (one W block) x (half of the A blocks)
  static_assert(MB == 8 || MB == 6 || MB == 4, "BM in {256, 192, 128}");
  extern __shared__ __attribute__((aligned(16))) uint4 lds[];
  const px_u32x4* ldsv = (const px_u32x4*)lds;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave & 1, wn = wave >> 1;
  const int nblk = gridDim.x, b = blockIdx.x;
  const int per = nblk >> 3;
  const int lbase = (b & 7) * per + (b >> 3);      // blocks b and b + 8 share an XCD and take consecutive tiles (n fastest)
  if (lbase >= ntiles) return;
  const int my_tiles = (ntiles - lbase + nblk - 1) / nblk;

  const char* gW = (const char*)wsplit;            // [N][K / 32][hi 32 | lo 32]: a row of the packed matrix is 4 K bytes
  const int r8 = lane >> 3, ch = (lane & 7) ^ r8;
  auto a_row_off = [&](int m) -> long { return ((long)(m / p.a_rpb) * p.a_bstride + (long)(m % p.a_rpb) * p.a_rstride) * 4; };
  // per tile parity: the 64-bit address of the tile's first A row (an fp32-sized conv activation exceeds 4 GiB), 32-bit offsets of this
  // lane's rows from it; ONE 32-bit offset of its W rows from the packed matrix (piece i sits a scalar distance behind piece 0)
  const char* abE;
  const char* abO;
  unsigned aofE[GA], aofO[GA], wofE, wofO;
  const long w_col = (long)p.K * 4;   // bytes per output column of the packed matrix
  auto setup = [&](int logical, const char*& ab, unsigned (&ao)[GA], unsigned& wo) {
    const int tile_n = logical % tiles_n, tile_m = logical / tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const long o0 = a_row_off(m0);
    ab = (const char*)p.A + o0;
#pragma unroll
    for (int i = 0; i < GA; ++i) {
      int m = m0 + (wave + 4 * i) * 8 + r8;
      if (m > p.M - 1) m = p.M - 1;
      ao[i] = (unsigned)(a_row_off(m) - o0) + ch * 16;
    }
    const int rho = wave * 8 + r8;   // LDS row of piece 0 in the W unit: (128-column group, block nb, row j) <- output column 8 j + nb
    const int n = n0 + (rho & 15) * 8 + ((rho >> 4) & 7);   // piece i: rho + 32 i -> column n + 2 (i & 3) + 128 (i >> 2)
    wo = (unsigned)((long)n * w_col + ch * 16);
  };
  auto w_piece = [&](int i) -> long { return (long)(2 * (i & 3) + 128 * (i >> 2)) * w_col; };
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(void __attribute__((address_space(3)))*)lds);
  auto lds_unit = [&](int slot, int i) -> unsigned { return lds0 + (unsigned)(slot * SLOT + (wave + 4 * i) * 64) * 16u; };

  f32x4 acc[8][MB];
  const int cq = lane >> 4, r16 = lane & 15, rr8 = r16 & 7;
  const int fragH = (r16 >> 3) * 64 + rr8 * 8 + (cq ^ rr8);          // hi pieces k = 8 cq .. + 7: chunk cq
  const int fragL = (r16 >> 3) * 64 + rr8 * 8 + ((4 + cq) ^ rr8);    // lo pieces: chunk 4 + cq
  const int xoff = (wm * MB) * 128;   // uint4 index of the wave's first 16-row block of the A unit
  const int woff = (wn * 8) * 128;    // ... of the W unit

  const int nk = p.K / BK;             // >= 3 (launcher)
  const bool has_bias = p.bias != nullptr;
  const bool do_gelu = p.act == ACT_GELU;
  setup(lbase, abE, aofE, wofE);
  setup(my_tiles > 1 ? nblk + lbase : lbase, abO, aofO, wofO);   // always rows that exist: the stream's surplus requests read them
  f32x4 bq[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) bq[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (has_bias) {
    const float* bp = p.bias + ((lbase % tiles_n) * BN + wn * 128 + (lane & 15) * 8);
    asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %2, off offset:16"
                 : "+v"(bq[0]), "+v"(bq[1]) : "v"(bp) : "memory");
  }
  // head of the stream: A_0 -> slot 0, W_0 -> slot 1, A_1 -> slot 2, W_1 -> slot 3
#pragma unroll
  for (int i = 0; i < GA; ++i) px_dma(aofE[i], abE, lds_unit(0, i));
#pragma unroll
  for (int i = 0; i < GW; ++i) px_dma(wofE, gW + w_piece(i), lds_unit(1, i));
#pragma unroll
  for (int i = 0; i < GA; ++i) px_dma(aofE[i], abE + ROWB, lds_unit(2, i));
#pragma unroll
  for (int i = 0; i < GW; ++i) px_dma(wofE, gW + ROWB + w_piece(i), lds_unit(3, i));
  px_wait_vm<GA + GW>();   // A_0 and W_0 (and the bias loads, older still) have landed
  asm volatile("" : "+v"(bq[0]), "+v"(bq[1]));
  __builtin_amdgcn_s_barrier();

  px_u32x4 xh[MB], xl[MB], wh[8], wl[8];
  int sa = 0, ti = 0;   // ring slot of A_g (W_g sits in the next one); tile index
  auto slot_add = [](int s_, int d) { const int t = s_ + d; return t >= NSLOT ? t - NSLOT : t; };

  // the first slab's lo x hi operands (the only reads whose latency the matrix pipe sees)
#pragma unroll
  for (int jj = 0; jj < MB; ++jj) xl[jj] = ldsv[0 * SLOT + xoff + jj * 128 + fragL];
#pragma unroll
  for (int nb = 0; nb < 8; ++nb) wh[nb] = ldsv[1 * SLOT + woff + nb * 128 + fragH];

  // one phase = 16 straight-line groups (no branch inside: gemm_p1w.hip).  PH 0 / 1 / 2 as in the header; rs = ring slot of the A unit the
  // phase READS from (PH 0, 1: this slab's; PH 2: the next slab's), its W unit sits in the slot behind.  REQ: 0 none; 1 the A unit (odd
  // groups); 2 the W unit (odd groups); 3 the W unit in groups 0-7 and THEN an A unit in groups 8-15 (a tile's last phase)
  auto phase = [&](auto ph_c, auto req_c, auto first_c, int rs, bool ev_a, const char* srcA, int slotA, bool ev_w, const char* srcW, int slotW) {
    constexpr int PH = decltype(ph_c)::value, REQ = decltype(req_c)::value;
    constexpr bool FIRST = decltype(first_c)::value != 0;   // a tile's first phase: the accumulators START from the bias (the MFMA's C operand)
    const int rw = slot_add(rs, 1);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int nb = i >> 1, mb0 = (i & 1) * MPG;
      if constexpr (PH == 0) {
        if constexpr (FIRST) {
          const float bv = bq[nb >> 2][nb & 3];
          const f32x4 b4 = {bv, bv, bv, bv};
#pragma unroll
          for (int j = 0; j < MPG; ++j) acc[nb][mb0 + j] = px_mma<F16>(xl[mb0 + j], wh[nb], b4);
        } else {
#pragma unroll
          for (int j = 0; j < MPG; ++j) acc[nb][mb0 + j] = px_mma<F16>(xl[mb0 + j], wh[nb], acc[nb][mb0 + j]);
        }
        if (i < MB) xh[i] = ldsv[rs * SLOT + xoff + i * 128 + fragH];
      } else if constexpr (PH == 1) {
#pragma unroll
        for (int j = 0; j < MPG; ++j) acc[nb][mb0 + j] = px_mma<F16>(xh[mb0 + j], wh[nb], acc[nb][mb0 + j]);
        if (i & 1) wl[nb] = ldsv[rw * SLOT + woff + nb * 128 + fragL];
      } else {
#pragma unroll
        for (int j = 0; j < MPG; ++j) acc[nb][mb0 + j] = px_mma<F16>(xh[mb0 + j], wl[nb], acc[nb][mb0 + j]);
        if (i < MB) xl[i] = ldsv[rs * SLOT + xoff + i * 128 + fragL];
        if (i & 1) wh[nb] = ldsv[rw * SLOT + woff + nb * 128 + fragH];
      }
      if (REQ == 1 && (i & 1) && (i >> 1) < GA) px_dma(ev_a ? aofE[i >> 1] : aofO[i >> 1], srcA, lds_unit(slotA, i >> 1));
      if (REQ == 2 && (i & 1) && (i >> 1) < GW) px_dma(ev_w ? wofE : wofO, srcW + w_piece(i >> 1), lds_unit(slotW, i >> 1));
      if (REQ == 3 && i < 8 && i < GW) px_dma(ev_w ? wofE : wofO, srcW + w_piece(i), lds_unit(slotW, i));
      if (REQ == 3 && i >= 8 && i - 8 < GA) px_dma(ev_a ? aofE[i - 8] : aofO[i - 8], srcA, lds_unit(slotA, i - 8));
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  using c0 = std::integral_constant<int, 0>;
  using c1 = std::integral_constant<int, 1>;
  using c2 = std::integral_constant<int, 2>;
  using c3 = std::integral_constant<int, 3>;
  // barrier B_g with its counted wait (OUT_ = requests / stores that may stay in flight)
  auto mid_barrier = [&](auto out_c) {
    px_wait_vm<decltype(out_c)::value>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
  };
  constexpr int NSTORE = MB * 8;                                    // 16-byte stores of one tile's epilogue per lane
  constexpr int W_AS = GA + NSTORE > 63 ? 63 : GA + NSTORE;         // (vmcnt is a 6-bit counter: the oldest stores are then waited for)
  using w_a = std::integral_constant<int, GA>;
  using w_as = std::integral_constant<int, W_AS>;

  // ---- the tile's epilogue: accumulators (bias already inside) -> activation -> fp32 rows / pair rows / planes ----
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
    const long row0 = wm * (BM / 2) + 4 * (ln >> 4);     // rows row0 + 16 mb + r
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
      // the accumulators live in AGPRs and the conversions read VGPRs: without this anchor hipcc copies ALL of them to VGPRs at the top of
      // the epilogue (hundreds of spills, and scratch traffic is VMEM traffic that breaks every counted vmcnt: gemm_p1w.hip)
#pragma unroll
      for (int j = 0; j < 8; ++j) asm volatile("" : "+a"(acc[j][mb]));
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = acc[j][mb][r];
        if (do_gelu) {
#pragma unroll
          for (int j = 0; j < 8; j += 2) {
            const f32x2_t y = gelu_fast2(f32x2_t{v[j], v[j + 1]});
            v[j] = y.x; v[j + 1] = y.y;
          }
        }
        // row and column in the VECTOR offset: range check + the soffset store-data hazard (gemm_pps.hip)
        const unsigned off = off0 + (mb * 16 + r) * row_pitch;
        if constexpr (OUT == 0) {
          const px_u32x4 v0 = {__builtin_bit_cast(unsigned, v[0]), __builtin_bit_cast(unsigned, v[1]), __builtin_bit_cast(unsigned, v[2]), __builtin_bit_cast(unsigned, v[3])};
          const px_u32x4 v1 = {__builtin_bit_cast(unsigned, v[4]), __builtin_bit_cast(unsigned, v[5]), __builtin_bit_cast(unsigned, v[6]), __builtin_bit_cast(unsigned, v[7])};
          __builtin_amdgcn_raw_buffer_store_b128(v0, crsrc, off, 0, 16);
          __builtin_amdgcn_raw_buffer_store_b128(v1, crsrc, off + 16, 0, 16);
        } else {
          px_u32x4 hi, lo;
          px_cut8<F16>(v, hi, lo);
          __builtin_amdgcn_raw_buffer_store_b128(hi, crsrc, off, 0, 16);
          __builtin_amdgcn_raw_buffer_store_b128(lo, lrsrc, OUT == 1 ? off + 64 : off, 0, 16);
        }
        __builtin_amdgcn_sched_barrier(0);   // one row at a time: eight accumulator reads, their conversion, two stores
      }
    }
    ++ti;
    if (ti + 1 < my_tiles) {   // the finished tile's offset set now belongs to tile ti + 1 (same parity)
      const char* nb_;
      unsigned na[GA], nw;
      setup((ti + 1) * nblk + lbase, nb_, na, nw);
      const bool into_odd = (ti & 1) == 0;
      abO = into_odd ? nb_ : abO;
      abE = into_odd ? abE : nb_;
#pragma unroll
      for (int i = 0; i < GA; ++i) { aofO[i] = into_odd ? na[i] : aofO[i]; aofE[i] = into_odd ? aofE[i] : na[i]; }
      wofO = into_odd ? nw : wofO; wofE = into_odd ? wofE : nw;
    }
  };

  // The stream, tile by tile.  Slab kt of tile ti (ring slot sa) requests, two slabs ahead, A (phase 0) and W (phase 2): kt + 2 < nk -> this
  // tile's slab kt + 2 (this parity's offsets), else the next tile's slab kt + 2 - nk (the other parity's).  The FIRST slab of a tile issues
  // no A request: the last phase of the tile before it did (REQ 3), in front of the epilogue's stores, so that the first barrier's counted
  // wait can leave those stores in flight.
  for (int t = 0; t < my_tiles; ++t) {
    const bool te = (ti & 1) == 0;   // this tile's offsets are the even set
    asm volatile("" : "+v"(bq[0]), "+v"(bq[1]));   // this tile's bias: fetched a slab before the previous tile's end, covered by a counted wait since
    // ---- slab 0 ----
    {
      const char* sA = (te ? abE : abO) + 2L * ROWB;
      const char* sW = gW + 2L * ROWB;
      if (t == 0) {   // the stream's very first slab issues its own A request
        phase(c0{}, c1{}, c1{}, sa, te, sA, slot_add(sa, 4), te, sW, sa);
        phase(c1{}, c0{}, c0{}, sa, te, sA, 0, te, sW, sa);
        mid_barrier(w_a{});
      } else {
        phase(c0{}, c0{}, c1{}, sa, te, sA, 0, te, sW, sa);
        phase(c1{}, c0{}, c0{}, sa, te, sA, 0, te, sW, sa);
        mid_barrier(w_as{});   // (the previous tile's later stores stay in flight)
      }
      phase(c2{}, c2{}, c0{}, slot_add(sa, 2), te, sA, 0, te, sW, sa);
      sa = slot_add(sa, 2);
    }
    // ---- slabs 1 .. nk - 2 ----
    for (int k = 1; k + 1 < nk; ++k) {
      const bool cur2 = k + 2 < nk;
      const bool ev = cur2 == te;
      const long ko = (long)(cur2 ? k + 2 : k + 2 - nk) * ROWB;
      const char* sA = (ev ? abE : abO) + ko;
      phase(c0{}, c1{}, c0{}, sa, ev, sA, slot_add(sa, 4), ev, gW + ko, sa);
      phase(c1{}, c0{}, c0{}, sa, ev, sA, 0, ev, gW + ko, sa);
      mid_barrier(w_a{});
      phase(c2{}, c2{}, c0{}, slot_add(sa, 2), ev, sA, 0, ev, gW + ko, sa);
      sa = slot_add(sa, 2);
    }
    // ---- slab nk - 1: its requests belong to the next tile's slab 1 (A, W) and -- in front of the stores -- slab 2 (A) ----
    {
      if (has_bias) {
        // the NEXT tile's bias: fetched in front of this slab's requests, so that the counted wait of the barrier -- which leaves only
        // those requests in flight -- covers it
        const float* bp = p.bias + ((((ti + 1) * nblk + lbase) % tiles_n) * BN + wn * 128 + (lane & 15) * 8);
        asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %2, off offset:16"
                     : "+v"(bq[0]), "+v"(bq[1]) : "v"(bp) : "memory");
      }
      const bool ev = !te;            // the other parity: the next tile
      const char* nA = ev ? abE : abO;
      phase(c0{}, c1{}, c0{}, sa, ev, nA + 1L * ROWB, slot_add(sa, 4), ev, gW, sa);
      phase(c1{}, c0{}, c0{}, sa, ev, nA, 0, ev, gW, sa);
      mid_barrier(w_a{});
      // W of the next tile's slab 1 into A_g's slot; then A of the next tile's slab 2 into the slot behind it (W_g's: slot sa + 1 = the
      // new first slab's sa' + 4) -- both free behind this barrier; the phase also reads the next tile's first operands
      phase(c2{}, c3{}, c0{}, slot_add(sa, 2), ev, nA + 2L * ROWB, slot_add(sa, 1), ev, gW + 1L * ROWB, sa);
      sa = slot_add(sa, 2);
    }
    epilogue();
  }
  // the surplus requests of the stream's tail must have landed before the workgroup gives its LDS back; the last epilogue's stores, younger,
  // need not be waited for
  px_wait_vm<(NSTORE > 63 ? 63 : NSTORE)>();
}

template <bool F16, int BM, int OUT>
int launch_p1x_t(const GemmArgs& a, const void* packed, hipStream_t s) {
  const int tiles_m = (a.M + BM - 1) / BM, tiles_n = a.N / 256;
  const int ntiles = tiles_m * tiles_n;
  const int nblk = ntiles < 256 ? ((ntiles + 7) / 8) * 8 : 256;
  const size_t lds_bytes = 5 * 32768;
  if (int r_ = ensure_dyn_lds((const void*)gemm_p1x_kernel<F16, BM, OUT>, (int)lds_bytes)) return r_;
  hipLaunchKernelGGL((gemm_p1x_kernel<F16, BM, OUT>), dim3(nblk), dim3(256), lds_bytes, s, a, packed, tiles_n, ntiles);
  SVT_LAUNCH_CHECK();
  return 0;
}

template <bool F16, int BM>
int launch_p1x_o(const GemmArgs& a, const void* packed, hipStream_t s) {
  if (a.planes) return launch_p1x_t<F16, BM, 2>(a, packed, s);
  if (a.c_pairs) return launch_p1x_t<F16, BM, 1>(a, packed, s);
  return launch_p1x_t<F16, BM, 0>(a, packed, s);
}

}  // namespace

// svt_debug_set key 30: 1 = this kernel for every gemm_x3q-eligible launch with K >= 96, 0 (default) = gemm_x3q_kernel.  Measured
// (profiles/r05_gemm_p1x_ab.txt): bit-identical outputs and the SAME speed -- isolated launches within +-2 % (QKV -5 %, FFN-1 with GELU +3 %),
// C2 fp16x3 2 567-2 570 against 2 570-2 578 clips/s, C3 1 015 both.  With three MFMAs per algorithmic multiply-add the split products are
// bound by the matrix pipe at the clock the chip holds under that load, not by how the slab's loads are scheduled around it; the 16-bit
// products, where the single-wave loop gained (gemm_p1w.hip), are not.  Kept as the A/B arm and as a second implementation the tests compare.
int g_gemm_p1x = 0;

// same arguments as launch_gemm_x3q (gemm_x3q.hip); the caller has checked gemm_x3q_eligible and K >= 96
int launch_gemm_p1x(int kind, const GemmArgs& a, const void* packed, int bm, hipStream_t s) {
  if (a.K < 96) { set_error("gemm_p1x: K < 96"); return -1; }
  if (kind == 3) {
    if (bm == 256) return launch_p1x_o<true, 256>(a, packed, s);
    if (bm == 192) return launch_p1x_o<true, 192>(a, packed, s);
    return launch_p1x_o<true, 128>(a, packed, s);
  }
  if (bm == 256) return launch_p1x_o<false, 256>(a, packed, s);
  if (bm == 192) return launch_p1x_o<false, 192>(a, packed, s);
  return launch_p1x_o<false, 128>(a, packed, s);
}

}  // namespace svt
#endif  // SVT_OPERAND_F16 || !SVT_DIAG

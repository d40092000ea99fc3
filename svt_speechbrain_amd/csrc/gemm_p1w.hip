// bf16 dense contraction for gfx950, round 5: ONE wave per SIMD, the slab's loads interleaved into the wave's own MFMA stream.
//
// gemm_pps_kernel keeps two waves on every SIMD and lets them alternate: one issues 12-16 MFMAs while its partner reads fragments and
// issues the ring's LDS-DMA requests, four barrier intervals per slab; its 256 x 256 x 64 slab costs ~2 750 cycles against the 2 048
// the matrix pipe needs.  tools/microbench/p1w_probe.hip measured the other classic form on this chip: a SINGLE wave per SIMD owning a
// 128 x 128 accumulator tile (256 accumulator registers in the AGPR half of the unified file, fragments double-buffered in 128 VGPRs)
// that issues its 128 MFMAs per slab back to back and places the slab's 32 fragment reads, its 16 requests and ONE barrier in the
// shadows of those MFMAs: 2 160 cycles per slab with the reads alone, 2 390 with requests, counted waits and the barrier -- 0.86 of
// the pipe inside the loop.  This kernel is that loop behind gemm_pps_kernel's persistent stream:
//   * 256 threads = 4 waves as 2 (M) x 2 (N), wave tile BM/2 x 128, v_mfma_f32_16x16x32 with the ACTIVATION fragment first, W rows
//     permuted in LDS so that a lane ends with 8 consecutive output columns (gemm_pps_kernel's register epilogue);
//   * the ring is gemm_pps_kernel's: five 32 KiB slots, unit u (A_0 W_0 A_1 W_1 ...) in slot u mod 5, filled by global_load_lds_dwordx4
//     with a 32-bit lane offset against a scalar base, XOR-swizzled 16-byte chunks, one conflict-free ds_read_b128 per fragment;
//   * slab g = k-step 0 (MFMAs on the fragments read during the previous k-step; meanwhile this slab's k-step-1 fragments are read and
//     A_{g+2} is requested), then the slab's ONE barrier B_g -- this wave's pieces of A_{g+1} / W_{g+1} have landed (counted vmcnt) and
//     its reads of A_g / W_g are complete (both k-steps' fragments sit in registers), so behind B_g slab g + 1 may be read and the slots
//     of slab g refilled -- then k-step 1 (meanwhile the next slab's k-step-0 fragments are read and W_{g+2} is requested into A_g's
//     slot).  Every request leads its first read by at least a whole slab (W) or two (A), as in gemm_pps_kernel;
//   * the tile's epilogue runs out of the accumulators between two slabs (bias as the accumulators' initial value, fetched a slab ahead;
//     GELU; 16-byte buffer stores); at a tile boundary the next slab's A requests are issued in front of the stores, so that the counted
//     wait of the next barrier can leave the stores in flight.
// Contract = gemm_pps_eligible (bf16 output, no residual, alpha = 1, none / GELU, K % 64 == 0, N % 256 == 0, spans < 4 GiB) and K >= 192
// (three slabs per tile: a tile boundary then never asks for rows of the tile after next, whose offsets exist only behind the epilogue).
#include "common.h"

namespace svt {
namespace {

// cache policy of the tile's row stores: 16 = sc1 (write-through, dispatched: measured against write-back in round 3 on gemm_pps_kernel and again in
// round 6 on this kernel, profiles/r06_gemm_store_policy_ab.txt); an experimental build may override it (make XNAME=wb XDEFS=-DSVT_P1W_STORE_AUX=0)
#ifndef SVT_P1W_STORE_AUX
#define SVT_P1W_STORE_AUX 16
#endif
template <int N> __device__ __forceinline__ void p1_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__device__ __forceinline__ void p1_dma(unsigned voff, const void* sbase, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_addr), "v"(voff), "s"(sbase) : "memory");
}

typedef unsigned p1_u32x4 __attribute__((ext_vector_type(4)));

template <int BM, int ACT, bool TR = false>   // TR: the diagnostic instantiation (stamps; may spill a register or two: timing only)
__global__ __launch_bounds__(256) void gemm_p1w_kernel(GemmArgs p, int tiles_n, int ntiles) {
  constexpr int BN = 256, BK = 64, NSLOT = 5;
  constexpr int MB = BM / 32;        // 16-row blocks per wave (wave tile BM/2 x 128)
  constexpr int GA = BM / 32;        // A pieces (8 rows x 128 B) per wave and slab
  constexpr int GW = BN / 32;        // W pieces per wave and slab
  constexpr int SLOT = 2048;         // uint4 per ring slot (32 KiB)
  constexpr int MPG = MB / 2;        // MFMAs per group: a k-step is 16 groups of (one W block) x (half of the A blocks)
  static_assert(MB == 8 || MB == 6 || MB == 4, "BM in {256, 192, 128}");
  extern __shared__ __attribute__((aligned(16))) uint4 lds[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave & 1, wn = wave >> 1;
  const int nblk = gridDim.x, b = blockIdx.x;
  // blocks b and b + 8 share an XCD: in every round an XCD works on nblk / 8 consecutive logical tiles (n fastest)
  const int per = nblk >> 3;
  const int lbase = (b & 7) * per + (b >> 3);
  if (lbase >= ntiles) return;
  const int my_tiles = (ntiles - lbase + nblk - 1) / nblk;
  const int tiles_m = ntiles / tiles_n;
  auto tile_col = [&](int logical) -> int { int tm, tn; tile_walk(logical, tiles_m, tiles_n, p.walk_pm, tm, tn); return tn; };

  const char* gA = (const char*)p.A;
  const char* gW = (const char*)p.W;
  const int r8 = lane >> 3, ch = (lane & 7) ^ r8;
  // byte offsets of the rows this lane fetches: one set for the workgroup's even tiles, one for its odd tiles (gemm_pps_kernel)
  unsigned aofE[GA], aofO[GA];
  // W: piece i of a wave starts 2 (i & 3) + 128 (i >> 2) output columns behind piece 0 -- a SCALAR distance (folded into the request's scalar
  // base), so one lane offset per set serves all eight pieces (the per-piece offsets cost 14 registers the stream could not spare)
  unsigned wofE, wofO;
  const long w_col = (long)p.ldw * 2;   // bytes per output column of W
  const bool plain_a = p.a_rpb >= p.M;
  auto setup = [&](int logical, unsigned (&ao)[GA], unsigned& wo) {
    int tile_n, tile_m;
    tile_walk(logical, tiles_m, tiles_n, p.walk_pm, tile_m, tile_n);
    const int m0 = tile_m * BM, n0 = tile_n * BN;
#pragma unroll
    for (int i = 0; i < GA; ++i) {
      int m = m0 + (wave + 4 * i) * 8 + r8;
      if (m > p.M - 1) m = p.M - 1;
      if (plain_a) ao[i] = (unsigned)(((long)m * p.a_rstride + ch * 8) * 2);
      else ao[i] = (unsigned)(((long)(m / p.a_rpb) * p.a_bstride + (long)(m % p.a_rpb) * p.a_rstride + ch * 8) * 2);
    }
    {
      const int rho = wave * 8 + r8;   // LDS row of piece 0 in the W unit: (128-column group, block nb, row j) <- output column 8 j + nb
      const int n = n0 + (rho & 15) * 8 + ((rho >> 4) & 7);   // piece i: rho + 32 i -> column n + 2 (i & 3) + 128 (i >> 2)
      wo = (unsigned)(((long)n * p.ldw + ch * 8) * 2);
    }
  };
  auto w_piece = [&](int i) -> long { return (long)(2 * (i & 3) + 128 * (i >> 2)) * w_col; };
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(void __attribute__((address_space(3)))*)lds);
  auto lds_unit = [&](int slot, int i) -> unsigned { return lds0 + (unsigned)(slot * SLOT + (wave + 4 * i) * 64) * 16u; };

  f32x4 acc[8][MB];
  const int cq = lane >> 4, r16 = lane & 15, rr8 = r16 & 7;
  const int frag0 = (r16 >> 3) * 64 + rr8 * 8 + (cq ^ rr8);
  const int frag1 = (r16 >> 3) * 64 + rr8 * 8 + ((4 + cq) ^ rr8);
  const int xoff = (wm * MB) * 128;   // uint4 index of the wave's first 16-row block of the A unit
  const int woff = (wn * 8) * 128;    // ... of the W unit

  const int nk = p.K / BK;             // >= 2 (launcher)
  // byte offset of K slab kk inside an A row: linear, or tap-minor for the kernel-3 convolutions (GemmArgs::k_taps; W is stored in slab order)
  const int taps = p.k_taps;
  auto aoff = [&](int kk) -> long {
    if (taps <= 1) return (long)kk * (BK * 2);
    const int cb = taps == 3 ? (kk * 21846) >> 16 : kk >> 1;     // kk / taps for taps in {2, 3}, kk < 32768
    return ((long)(kk - cb * taps) * p.k_cin + (long)cb * BK) * 2;
  };
  const bool no_epi = p.dbg == 3, has_bias = p.bias != nullptr;
  // diagnostics (tools/gemm_trace.py --p1w; dbg = 9 sets p.trace): wall-clock stamps at entry / first slab / exit, the epilogues' share, core
  // cycles over the stream and the cycles the wave spent between reaching a slab's counted wait and leaving its barrier
  constexpr bool tr = TR;
  long long t_begin = 0, t_first = 0, t_epi = 0, c_first = 0, c_wait = 0;
  if (tr) t_begin = wall_clock64();
  setup(lbase, aofE, wofE);
  setup(my_tiles > 1 ? nblk + lbase : lbase, aofO, wofO);   // always rows that exist: the stream's surplus requests read them
  f32x4 bq[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) bq[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (has_bias) {
    const float* bp = p.bias + (tile_col(lbase) * BN + wn * 128 + (lane & 15) * 8);
    asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %2, off offset:16"
                 : "+v"(bq[0]), "+v"(bq[1]) : "v"(bp) : "memory");
  }
  // head of the stream: A_0 -> slot 0, W_0 -> slot 1, A_1 -> slot 2, W_1 -> slot 3 (nk >= 2: slab 1 belongs to the first tile)
#pragma unroll
  for (int i = 0; i < GA; ++i) p1_dma(aofE[i], gA, lds_unit(0, i));
#pragma unroll
  for (int i = 0; i < GW; ++i) p1_dma(wofE, gW + w_piece(i), lds_unit(1, i));
#pragma unroll
  for (int i = 0; i < GA; ++i) p1_dma(aofE[i], gA + aoff(1), lds_unit(2, i));
#pragma unroll
  for (int i = 0; i < GW; ++i) p1_dma(wofE, gW + BK * 2 + w_piece(i), lds_unit(3, i));
  p1_wait_vm<GA + GW>();   // A_0 and W_0 (and the bias loads, older still) have landed
  asm volatile("" : "+v"(bq[0]), "+v"(bq[1]));
  __builtin_amdgcn_s_barrier();
  if (tr) { t_first = wall_clock64(); c_first = __builtin_amdgcn_s_memtime(); }

  // A fragments of the k-step being multiplied and of the one being read; W fragments in ONE buffer: block nb of the next k-step is read
  // into its own registers right behind the last group that multiplies with it (14 groups = ~900 cycles before its next use)
  bf16x8 xf[2][MB], wf[8];
  int sa = 0, ti = 0;   // ring slot of A_g (W_g sits in the next one); tile index

  // ---- a tile's end: the next tile index, and the finished tile's offset set re-pointed at the tile after next ----
  auto tile_end = [&]() {
    ++ti;
    if (ti + 1 < my_tiles) {   // the finished tile's offset set now belongs to tile ti + 1 (same parity)
      unsigned na[GA], nw;
      setup((ti + 1) * nblk + lbase, na, nw);
      const bool into_odd = (ti & 1) == 0;
#pragma unroll
      for (int i = 0; i < GA; ++i) { aofO[i] = into_odd ? na[i] : aofO[i]; aofE[i] = into_odd ? aofE[i] : na[i]; }
      wofO = into_odd ? nw : wofO; wofE = into_odd ? wofE : nw;
    }
  };

  // fragments of k-step ks of the slab whose A unit sits in ring slot `slot`
  auto read_x = [&](int slot, int ks, int buf, int jj) {
    xf[buf][jj] = __builtin_bit_cast(bf16x8, lds[slot * SLOT + xoff + jj * 128 + (ks ? frag1 : frag0)]);
  };
  auto read_w = [&](int slot, int ks, int nb) {
    const int ws = slot + 1 >= NSLOT ? slot + 1 - NSLOT : slot + 1;
    wf[nb] = __builtin_bit_cast(bf16x8, lds[ws * SLOT + woff + nb * 128 + (ks ? frag1 : frag0)]);
  };

  // the first slab's k-step-0 fragments (the only reads whose latency the matrix pipe sees)
#pragma unroll
  for (int jj = 0; jj < MB; ++jj) read_x(0, 0, 0, jj);
#pragma unroll
  for (int nb = 0; nb < 8; ++nb) read_w(0, 0, nb);

  // one k-step = 16 straight-line groups (no branch inside: a branch per group made hipcc produce the MFMA results in VGPRs and copy
  // them to the accumulator registers, four copies per MFMA): MPG MFMAs on fragment buffer BUF, then the group's share of the OTHER
  // buffer's fragment reads (slot rs, k-step rks) and of the requests.  REQ: 0 none; 1 the A unit (odd groups); 2 the W unit (odd groups);
  // 3 the W unit in groups 0-7 and THEN an A unit in groups 8-15 (a tile's last k-step: all W pieces older than all A pieces)
  auto kstep = [&](auto buf_c, auto req_c, auto first_c, int rs, int rks, bool ev_a, const char* srcA, int slotA, bool ev_w, const char* srcW, int slotW) {
    constexpr int BUF = decltype(buf_c)::value, REQ = decltype(req_c)::value;
    constexpr bool FIRST = decltype(first_c)::value != 0;   // a tile's first k-step: the accumulators START from the bias (the MFMA's C operand)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int nb = i >> 1, mb0 = (i & 1) * MPG;
      if constexpr (FIRST) {
        const float bv = bq[nb >> 2][nb & 3];
        const f32x4 b4 = {bv, bv, bv, bv};
#pragma unroll
        for (int j = 0; j < MPG; ++j) acc[nb][mb0 + j] = SVT_MFMA_16x16x32(xf[BUF][mb0 + j], wf[nb], b4);
      } else {
#pragma unroll
        for (int j = 0; j < MPG; ++j) acc[nb][mb0 + j] = SVT_MFMA_16x16x32(xf[BUF][mb0 + j], wf[nb], acc[nb][mb0 + j]);
      }
      if (i & 1) read_w(rs, rks, nb);                 // block nb is done for this k-step: its registers take the next k-step's
      if (i < MB) read_x(rs, rks, BUF ^ 1, i);
      if (REQ == 1 && (i & 1) && (i >> 1) < GA) p1_dma(ev_a ? aofE[i >> 1] : aofO[i >> 1], srcA, lds_unit(slotA, i >> 1));
      if (REQ == 2 && (i & 1) && (i >> 1) < GW) p1_dma(ev_w ? wofE : wofO, srcW + w_piece(i >> 1), lds_unit(slotW, i >> 1));
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // A tile's LAST k-step, row block by row block: the 8 MFMAs of block mb finish its 16 rows x 128 columns, which are converted and stored
  // while the matrix pipe works on block mb + 1 -- the epilogue runs INSIDE the stream (as its own phase it cost 2.6-4 us per tile: 256
  // accumulator reads, their conversions and 32 stores per lane with no partner wave to multiply beside them).  The MFMA results of a block
  // are taken in VGPRs (they are read by vector instructions next; the accumulators proper are not written again: the next tile starts from
  // its bias).  Requests: W of the next tile's slab 1, then A of its slab 2 -- every W piece older than every A piece, NREQ per block --,
  // the stores of a block at its end.  STORES_AFTER_W = stores issued behind the last W piece: with the A pieces they are what the next
  // barrier's counted wait leaves in flight.
  constexpr bool FIN_VGPR = !(BM == 256 && ACT == ACT_GELU);
  constexpr int NREQ = (GW + GA + MB - 1) / MB;
  constexpr int LAST_W_BLOCK = (GW - 1) / NREQ;                 // row block in which the last W piece is issued
  constexpr int STORES_AFTER_W = (MB - LAST_W_BLOCK) * 4;       // that block's stores (issued at its end) and all later ones
  // the counted waits' bookkeeping: the last k-step has room for every W and A piece of the slab it requests; vmcnt is a 6-bit counter;
  // loads and stores of one wave retire IN ORDER on that counter (gfx9 returns VMEM in issue order), which w_as below relies on
  static_assert(NREQ * MB >= GW + GA, "the last k-step issues every request of the next tile's head");
  static_assert(GA + STORES_AFTER_W <= 63 && GA + GW <= 63, "vmcnt holds 6 bits");
  auto kstep_last = [&](int rs, bool ev, const char* srcA, int slotA, const char* srcW, int slotW) {
    const int logical = ti * nblk + lbase;
    int tile_n, tile_m;
    tile_walk(logical, tiles_m, tiles_n, p.walk_pm, tile_m, tile_n);
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const long rows_left = (long)p.M - m0;
    const unsigned long nbytes = (unsigned long)rows_left * p.ldc * 2;
    const unsigned nrec = nbytes > 0xFFFFFFF0ul ? 0xFFFFFFF0u : (unsigned)nbytes;
    char* cbase = (char*)p.C + (long)m0 * p.ldc * 2;
    const auto crsrc = __builtin_amdgcn_make_buffer_rsrc(cbase, 0, nrec, 0x00020000);   // rows >= M land beyond num_records and are dropped
    // this lane: rows 4 (lane >> 4) + r of every 16-row block of the wave, columns 8 (lane & 15) .. + 7 of the wave's 128
    const unsigned off0 = (unsigned)((((long)(wm * (BM / 2) + 4 * (lane >> 4))) * p.ldc + n0 + wn * 128 + (lane & 15) * 8) * 2);
    const unsigned row_pitch = (unsigned)(p.ldc * 2);
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      f32x4 fin[8];
      if constexpr (FIN_VGPR) {
#pragma unroll
        for (int nb = 0; nb < 8; ++nb) fin[nb] = SVT_MFMA_16x16x32(xf[1][mb], wf[nb], acc[nb][mb]);
      } else {
        // 256-row tiles with GELU: no 32 spare VGPRs beside the polynomial's temporaries -- the block's results stay in the accumulator
        // registers and are read from there row by row (a spill would put the compiler's own vmcnt(0) waits into the stream)
#pragma unroll
        for (int nb = 0; nb < 8; ++nb) acc[nb][mb] = SVT_MFMA_16x16x32(xf[1][mb], wf[nb], acc[nb][mb]);
#pragma unroll
        for (int nb = 0; nb < 8; ++nb) asm volatile("" : "+a"(acc[nb][mb]));
      }
      read_x(rs, 0, 0, mb);                            // the next tile's first A fragments (buffer 0 is free in this k-step)
#pragma unroll
      for (int q = 0; q < NREQ; ++q) {
        const int rq = mb * NREQ + q;                  // requests 0 .. GW - 1: W pieces; GW .. GW + GA - 1: A pieces
        if (rq < GW) p1_dma(ev ? wofE : wofO, srcW + w_piece(rq), lds_unit(slotW, rq));
        else if (rq < GW + GA) p1_dma(ev ? aofE[rq - GW] : aofO[rq - GW], srcA, lds_unit(slotA, rq - GW));
      }
      __builtin_amdgcn_sched_barrier(0);
      if (!no_epi) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          bf16x8 o;
          if constexpr (ACT == ACT_GELU) {
            f32x2_t g[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) g[j] = FIN_VGPR ? f32x2_t{fin[2 * j][r], fin[2 * j + 1][r]} : f32x2_t{acc[2 * j][mb][r], acc[2 * j + 1][mb][r]};
            if constexpr (FIN_VGPR) {
              gelu_bf16x2_x4(g);
            } else {   // one pair at a time: a quarter of the polynomial's temporaries (the next row block's MFMAs fill the gaps of the chain)
#pragma unroll
              for (int j = 0; j < 4; ++j) { g[j] = gelu_bf16x2(g[j]); asm volatile("" : "+v"(g[j])); }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              o[2 * j] = (bf16_t)g[j].x;
              o[2 * j + 1] = (bf16_t)g[j].y;
            }
          } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (bf16_t)(FIN_VGPR ? fin[j][r] : acc[j][mb][r]);
          }
          // the row goes into the VECTOR offset (range check of a raw buffer; see gemm_pps_kernel for the soffset hazard)
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(p1_u32x4, o), crsrc, off0 + (mb * 16 + r) * row_pitch, 0, SVT_P1W_STORE_AUX);
          if constexpr (ACT == ACT_GELU) __builtin_amdgcn_sched_barrier(0);   // one row's polynomial at a time (register pressure)
        }
      } else {
#pragma unroll
        for (int nb = 0; nb < 8; ++nb) { if constexpr (FIN_VGPR) asm volatile("" ::"v"(fin[nb])); else asm volatile("" ::"a"(acc[nb][mb])); }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  using c0 = std::integral_constant<int, 0>;
  using c1 = std::integral_constant<int, 1>;
  using c2 = std::integral_constant<int, 2>;
  auto slot_add = [](int s_, int d) { const int t = s_ + d; return t >= NSLOT ? t - NSLOT : t; };
  // barrier B_g with its counted wait (OUT = requests / stores that may stay in flight)
  auto mid_barrier = [&](auto out_c) {
    long long c0_ = 0;
    if (tr) c0_ = __builtin_amdgcn_s_memtime();
    p1_wait_vm<decltype(out_c)::value>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    if (tr) c_wait += __builtin_amdgcn_s_memtime() - c0_;
  };
  using w_a = std::integral_constant<int, GA>;
  using w_as = std::integral_constant<int, GA + STORES_AFTER_W>;

  // The stream, tile by tile.  Slab kt of tile ti (ring slot sa) requests, two slabs ahead, A (during k-step 0) and W (during k-step 1):
  // kt + 2 < nk -> this tile's slab kt + 2 (this parity's offsets), else the next tile's slab kt + 2 - nk (the other parity's).  The
  // FIRST slab of a tile issues no A request in its k-step 0: the last k-step of the tile before it did (REQ 3), in front of the epilogue's
  // stores, so that the first barrier's counted wait can leave those stores in flight.
  for (int t = 0; t < my_tiles; ++t) {
    const bool te = (ti & 1) == 0;   // this tile's offsets are the even set
    const bool first_tile = t == 0;
    asm volatile("" : "+v"(bq[0]), "+v"(bq[1]));   // this tile's bias: fetched a slab before the previous tile's end, covered by a counted wait since
    // ---- slab 0: the accumulators start from the bias; the stream's very first slab issues its own A request ----
    {
      const char* sA = gA + aoff(2);
      const char* sW = gW + 2L * (BK * 2);
      if (first_tile) {
        kstep(c0{}, c1{}, c1{}, sa, 1, te, sA, slot_add(sa, 4), te, sW, sa);
        mid_barrier(w_a{});
      } else {
        kstep(c0{}, c0{}, c1{}, sa, 1, te, sA, slot_add(sa, 4), te, sW, sa);
        if (no_epi) mid_barrier(w_a{}); else mid_barrier(w_as{});   // (the previous tile's later stores stay in flight)
      }
      kstep(c1{}, c2{}, c0{}, slot_add(sa, 2), 0, te, sA, 0, te, sW, sa);
      sa = slot_add(sa, 2);
    }
    // ---- slabs 1 .. nk - 2 ----
    for (int k = 1; k + 1 < nk; ++k) {
      const bool cur2 = k + 2 < nk;
      const bool ev = cur2 == te;
      const int kk = cur2 ? k + 2 : k + 2 - nk;
      const long ko = (long)kk * (BK * 2), koa = aoff(kk);
      kstep(c0{}, c1{}, c0{}, sa, 1, ev, gA + koa, slot_add(sa, 4), ev, gW + ko, sa);
      mid_barrier(w_a{});
      kstep(c1{}, c2{}, c0{}, slot_add(sa, 2), 0, ev, gA + koa, 0, ev, gW + ko, sa);
      sa = slot_add(sa, 2);
    }
    // ---- slab nk - 1: its requests belong to the next tile's slab 1 (A, W) and slab 2 (A); its second k-step carries the tile's epilogue ----
    {
      if (has_bias) {
        // the NEXT tile's bias: fetched in front of this k-step's requests, so that the counted wait of the barrier -- which leaves only
        // those requests in flight -- covers it
        const float* bp = p.bias + (tile_col(ti + 1 < my_tiles ? (ti + 1) * nblk + lbase : lbase) * BN + wn * 128 + (lane & 15) * 8);
        asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %2, off offset:16"
                     : "+v"(bq[0]), "+v"(bq[1]) : "v"(bp) : "memory");
      }
      const bool ev = !te;            // the other parity: the next tile
      kstep(c0{}, c1{}, c0{}, sa, 1, ev, gA + aoff(1), slot_add(sa, 4), ev, gW, sa);
      mid_barrier(w_a{});
      long long t_e0 = 0;
      if (tr) t_e0 = wall_clock64();
      // W of the next tile's slab 1 into A_g's slot, A of its slab 2 into W_g's (= the new first slab's "two ahead" slot): both free behind
      // this barrier; the next tile's first A fragments are read here too, its W fragments behind the k-step (their registers are in use)
      kstep_last(slot_add(sa, 2), ev, gA + aoff(2), slot_add(sa, 1), gW + 1L * (BK * 2), sa);
      sa = slot_add(sa, 2);
#pragma unroll
      for (int nb = 0; nb < 8; ++nb) read_w(sa, 0, nb);
      if (tr) t_epi += wall_clock64() - t_e0;
    }
    tile_end();
  }
  long long t_loop = 0;
  if (tr) t_loop = wall_clock64();
  // Everything this wave still has in flight -- the last tile's stores AND the stream's surplus LDS-DMA requests (they are INTERLEAVED with
  // those stores in the last k-step's row blocks LAST_W_BLOCK + 1 .. MB - 1, not older than them: ADVICE r05) -- retires before the wave ends
  p1_wait_vm<0>();
  if (tr && lane == 0) {   // record layout of gemm_pps_kernel (tools/gemm_trace.py); waves 0 / 2 stand for its two wave groups
    if ((wave & 1) == 0) {
      long long* o = p.trace + ((long)blockIdx.x * 2 + (wave >> 1)) * 8;
      o[0] = t_begin; o[1] = t_first; o[2] = t_loop - t_first - t_epi; o[3] = t_epi; o[4] = t_loop; o[5] = my_tiles;
      o[6] = __builtin_amdgcn_s_memtime() - c_first; o[7] = BM;
    }
    p.trace[524288 + (long)blockIdx.x * 4 + wave] = c_wait;   // cycles between reaching a slab's counted wait and leaving its barrier, summed
  }
}

template <int BM, int ACT, bool TR = false>
int launch_p1w_t(const GemmArgs& a, hipStream_t s) {
  const int tiles_m = (a.M + BM - 1) / BM, tiles_n = a.N / 256;
  const int ntiles = tiles_m * tiles_n;
  const int nblk = ntiles < g_gemm_persist_wgs ? ((ntiles + 7) / 8) * 8 : g_gemm_persist_wgs;   // (svt_debug_set key 37: workgroups of a persistent launch)
  const size_t lds_bytes = 5 * 32768;
  GemmArgs aw = a;
  aw.walk_pm = gemm_walk_pm(a, BM);
  if (int r_ = ensure_dyn_lds((const void*)gemm_p1w_kernel<BM, ACT, TR>, (int)lds_bytes)) return r_;
  const double flops = 2.0 * a.M * (double)a.N * a.K;
  const double bytes = ((double)a.M * a.K + (double)a.N * a.K) * 2 + (double)a.M * a.N * 2;
  prof_begin(s);
  hipLaunchKernelGGL((gemm_p1w_kernel<BM, ACT, TR>), dim3(nblk), dim3(256), lds_bytes, s, aw, tiles_n, ntiles);
  prof_end(s, flops, bytes, 0);
  SVT_LAUNCH_CHECK();
  return 0;
}

}  // namespace

int g_gemm_p1w = 1;   // svt_debug_set key 29: 1 (default) = this kernel where it measured faster than gemm_pps_kernel (gemm_dma.hip), 0 = never

int launch_gemm_p1w(const GemmArgs& a, int bm, hipStream_t s) {
#ifdef SVT_DIAG
  if (a.trace) {   // tools/gemm_trace.py --p1w (make DIAG=1): the two tile heights of the encoder's launches, without activation
    if (bm == 256) return launch_p1w_t<256, ACT_NONE, true>(a, s);
    return launch_p1w_t<192, ACT_NONE, true>(a, s);
  }
#else
  if (a.trace) { set_error("gemm_p1w: the stamped instantiations are built by `make DIAG=1` only"); return -1; }
#endif
  if (a.act == ACT_GELU) {
    if (bm == 256) return launch_p1w_t<256, ACT_GELU>(a, s);
    if (bm == 192) return launch_p1w_t<192, ACT_GELU>(a, s);
    return launch_p1w_t<128, ACT_GELU>(a, s);
  }
  if (bm == 256) return launch_p1w_t<256, ACT_NONE>(a, s);
  if (bm == 192) return launch_p1w_t<192, ACT_NONE>(a, s);
  return launch_p1w_t<128, ACT_NONE>(a, s);
}

}  // namespace svt

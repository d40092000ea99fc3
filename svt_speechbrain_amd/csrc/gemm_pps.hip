// bf16 dense contraction for gfx950: PERSISTENT + STAGGERED form of the LDS-DMA pipeline of gemm_dma.hip.
//
// gemm_pp8_kernel (one tile per workgroup) multiplies a 256 x 256 x 64 slab in 1.24 us -- eight slots per slab, waves 4-7 one
// slot behind waves 0-3, so that on every SIMD one wave issues MFMAs while its partner reads fragments and issues the ring's
// LDS-DMA -- but pays a 2 us prologue and a 6-8 us LDS-transposed epilogue per tile with every CU storing at once.
// gemm_pers_kernel hides prologue and stores behind the next tile, but its lockstep slab costs 1.67 us.  This kernel is the
// staggered schedule run as ONE stream of (tile, K slab) pairs per workgroup:
//   * a workgroup per CU walks its tile list; the ring (five 32 KiB slots, units A_g W_g A_g+1 W_g+1 ..., three in flight) never
//     drains between tiles: during the last two slabs of a tile the units requested are the next tile's first ones;
//   * LDS-DMA is issued from inline asm in the `voffset + SGPR base` form (global_load_lds_dwordx4 v, s[base:base+1]): one 32-bit
//     row offset per DMA instruction instead of a 64-bit pointer (two tiles' worth of sources are live at a tile boundary), the K
//     advance is scalar, and the compiler -- which sees no LDS write and no VMEM load -- never adds a vmcnt(0) of its own;
//   * waves are laid out 4 (M) x 2 (N), wave tile BM/4 x 128, and the MFMA takes the ACTIVATION fragment as its first operand: a lane
//     then holds rows 4 (lane >> 4) + r of a 16-row block and, with the W rows of a 128-column group placed in LDS so that (block
//     nb, row j) is output column 8 j + nb, EIGHT consecutive columns of each: one buffer_store_dwordx4 writes four rows x 256
//     contiguous bytes.  (The first form -- W fragment first, 16 lanes = 16 rows -- wrote 64 separate 16-byte pieces per
//     instruction: the epilogue was bound by store ISSUE, 3.1 us per 256 x 256 tile without activation.)
//   * the epilogue runs out of the accumulators (bias, GELU, bf16, bounds-checked buffer stores) in the slot where the two wave
//     groups meet: waves 0-3 run it in front of the next tile's first LOAD slot, waves 4-7
//     behind their last MFMA slot, i.e. in the same barrier interval, so the two waves of a SIMD interleave their VALU work
//     instead of each running it alone against an idle partner; the stores drain under the next tile's MFMAs;
//   * the bias is the accumulators' INITIAL value: the bias of the NEXT tile is fetched by two asm loads in the last slab of every
//     tile (the first tile's in front of the head of the stream); the counted wait that retires that slab covers them (VMEM
//     operations return in order), so the epilogue neither waits nor adds;
//   * dispatched schedule: four barriers per slab (HB = true below: between two barriers every wave runs one LOAD slot and one MFMA
//     slot, in opposite order for the two wave groups); the eight-barrier form (one barrier behind every slot) is kept for the slot
//     stamps of tools/gemm_trace.py and as the A/B of svt_debug_set key 16.
// Contract: bf16 (operand type) output, no residual, alpha = 1, activation none or GELU, K % 64 == 0, K >= 128, N % 256 == 0,
// A and W spans below 4 GiB (32-bit offsets).  Everything else stays on gemm_pers_kernel / gemm_pp8_kernel.
#include "common.h"

namespace svt {
namespace {

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// one LDS-DMA instruction: 64 lanes x 16 bytes from sbase + voff (per lane) to LDS bytes [lds_addr, lds_addr + 1024)
__device__ __forceinline__ void dma_sv(unsigned voff, const void* sbase, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_addr), "v"(voff), "s"(sbase) : "memory");
}

typedef unsigned u32x4v __attribute__((ext_vector_type(4)));

// an empty asm that "reads and writes" every register of a fragment array: whatever loads them must have completed here
template <int N> __device__ __forceinline__ void touch_regs(bf16x8 (&r)[N]) {
  static_assert((N >= 2 && N <= 4) || N == 8, "fragment arrays of 2..4 or 8 blocks");
  if constexpr (N == 8) asm volatile("" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]));
  else if constexpr (N == 4) asm volatile("" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]));
  else if constexpr (N == 3) asm volatile("" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]));
  else asm volatile("" : "+v"(r[0]), "+v"(r[1]));
}

// STAUX: cache policy of the epilogue stores (buffer instruction aux bits: 0 = default write-back, 2 = nt, 16 = sc1 write-through)
// STAMP (tools/gemm_trace.py --slots): s_memtime at both ends of every slot of one slab of the stream, per wave
// HB: four barriers per slab instead of eight (see the loops)
// TWO (round 5): two slots per slab instead of four -- see the schedule in front of its loops
template <int BM, int ACT, int STAUX, int STAMP = 0, bool HB = false, bool TWO = false>
__global__ __launch_bounds__(512) void gemm_pps_kernel(GemmArgs p, int tiles_n, int ntiles) {
  constexpr int BN = 256, BK = 64, NSLOT = 5;
  constexpr int MB = BM / 64;        // 16-row blocks per wave (wave tile BM/4 x 128)
  constexpr int GA = BM / 64, GW = BN / 64;
  constexpr int SLOT = 2048;        // uint4 per ring slot (32 KiB)
  extern __shared__ __attribute__((aligned(16))) uint4 lds[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave & 3, wn = (wave >> 2) & 1;   // waves 0-3 (columns 0-127) run one slot ahead of waves 4-7 (columns 128-255)
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
  // byte offsets of the rows this lane fetches, one set for the workgroup's even tiles (0, 2, ..) and one for its odd tiles: the
  // tile being multiplied reads one set, the requests that run ahead into the next tile read the other, and a finished tile's set
  // is overwritten IN PLACE with the offsets of the tile after next.  (A first form kept "current" and "next" sets and copied
  // next -> current at every tile end: hipcc turned that rotation into a swap of the two sets through temporaries on EVERY slab,
  // 32 v_mov between a wave's last MFMA slot and its next LOAD slot -- the slot stamps of tools/gemm_trace.py --slots found ~400
  // cycles per slab and wave group outside the slots, 3 100 cycles per slab against 2 600 in the one-tile kernel.)
  unsigned aofE[GA], wofE[GW], aofO[GA], wofO[GW];
  const bool plain_a = p.a_rpb >= p.M;
  auto setup = [&](int logical, unsigned (&ao)[GA], unsigned (&wo)[GW]) {
    int tile_n, tile_m;
    tile_walk(logical, tiles_m, tiles_n, p.walk_pm, tile_m, tile_n);
    const int m0 = tile_m * BM, n0 = tile_n * BN;
#pragma unroll
    for (int i = 0; i < GA; ++i) {
      int m = m0 + (wave + 8 * i) * 8 + r8;
      if (m > p.M - 1) m = p.M - 1;
      if (plain_a) ao[i] = (unsigned)(((long)m * p.a_rstride + ch * 8) * 2);   // one run of rows: no per-lane division
      else ao[i] = (unsigned)(((long)(m / p.a_rpb) * p.a_bstride + (long)(m % p.a_rpb) * p.a_rstride + ch * 8) * 2);
    }
#pragma unroll
    for (int i = 0; i < GW; ++i) {
      const int rho = (wave + 8 * i) * 8 + r8;   // LDS row of the W unit: (128-column group, block nb, row j) <- output column 8 j + nb
      const int n = n0 + (rho >> 7) * 128 + (rho & 15) * 8 + ((rho >> 4) & 7);   // < N: N % 256 == 0
      wo[i] = (unsigned)(((long)n * p.ldw + ch * 8) * 2);
    }
  };
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(void __attribute__((address_space(3)))*)lds);
  auto lds_unit = [&](int slot, int i) -> unsigned { return lds0 + (unsigned)(slot * SLOT + (wave + 8 * i) * 64) * 16u; };

  auto dma = [&](unsigned voff, const void* sbase, unsigned lds_addr) { dma_sv(voff, sbase, lds_addr); };
  f32x4 acc[8][MB];

  const int cq = lane >> 4, r16 = lane & 15, rr8 = r16 & 7;
  const int frag0 = (r16 >> 3) * 64 + rr8 * 8 + (cq ^ rr8);
  const int frag1 = (r16 >> 3) * 64 + rr8 * 8 + ((4 + cq) ^ rr8);
  const int xoff = (wm * MB) * 128;   // uint4 index of the wave's first 16-row block of the A unit
  const int woff = (wn * 8) * 128;    // ... of the W unit

  const int nk = p.K / BK;             // >= 2 (launcher)
  const int G = my_tiles * nk;         // slabs in this workgroup's stream
  // diagnostics (tools/gemm_trace.py, dbg = 9 sets p.trace): s_memrealtime (100 MHz) stamps at entry / first slab / around every
  // epilogue / exit, s_memtime (core clock) over the stream; dbg 1 = no LDS-DMA after the head of the stream, 3 = no epilogue
  const bool tr = p.trace != nullptr;
  const bool no_epi = p.dbg == 3, has_bias = p.bias != nullptr;
  long long t_begin = 0, t_first = 0, t_epi = 0, c_first = 0;
  if (tr) t_begin = wall_clock64();
  setup(lbase, aofE, wofE);
  setup(my_tiles > 1 ? nblk + lbase : lbase, aofO, wofO);   // always rows that exist: the stream's surplus requests (below) read them
  // The bias is the INITIAL value of the accumulators (one v_mov per register, what clearing them costs anyway) instead of 128
  // additions per lane in the epilogue: bq holds the bias of this lane's 8 columns for the NEXT tile to start, fetched here for
  // the first tile and in the last slab of every tile for the one after it.
  f32x4 bq[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) bq[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (has_bias) {
    const float* bp = p.bias + (tile_col(lbase) * BN + wn * 128 + (lane & 15) * 8);
    asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %2, off offset:16"
                 : "+v"(bq[0]), "+v"(bq[1]) : "v"(bp) : "memory");
  }
  // TWO: LDS = A ring of three BM-row units, then two rings of two 128-row HALF units of W: half h holds output columns 128 h .. + 127, the
  // columns of wave group h alone
  constexpr unsigned A_BYTES = BM * 128u, WH_BYTES = 128 * 128u;
  auto lds_a2 = [&](int slot, int i) -> unsigned { return lds0 + (unsigned)slot * A_BYTES + (unsigned)(wave + 8 * i) * 1024u; };
  auto lds_w2 = [&](int slot, int i) -> unsigned {   // piece i of a W unit: half i >> 1, rows (wave + 8 (i & 1)) * 8 .. + 7 of that half
    return lds0 + 3 * A_BYTES + (unsigned)((i >> 1) * 2 + slot) * WH_BYTES + (unsigned)(wave + 8 * (i & 1)) * 1024u;
  };
  if constexpr (TWO) {
    // head: A_0, W_0 (both halves), A_1, W^0_1
#pragma unroll
    for (int i = 0; i < GA; ++i) dma(aofE[i], gA, lds_a2(0, i));
#pragma unroll
    for (int i = 0; i < GW; ++i) dma(wofE[i], gW, lds_w2(0, i));
#pragma unroll
    for (int i = 0; i < GA; ++i) dma(aofE[i], gA + BK * 2, lds_a2(1, i));
#pragma unroll
    for (int i = 0; i < 2; ++i) dma(wofE[i], gW + BK * 2, lds_w2(1, i));
    wait_vm<GA + 2>();
  } else {
  // head of the stream: A_0 -> slot 0, W_0 -> slot 1, A_1 -> slot 2
#pragma unroll
  for (int i = 0; i < GA; ++i) dma(aofE[i], gA, lds_unit(0, i));
#pragma unroll
  for (int i = 0; i < GW; ++i) dma(wofE[i], gW, lds_unit(1, i));
#pragma unroll
  for (int i = 0; i < GA; ++i) dma(aofE[i], gA + BK * 2, lds_unit(2, i));
  wait_vm<GA>();   // the bias loads are older than every request of the head
  }
  asm volatile("" : "+v"(bq[0]), "+v"(bq[1]));
  if constexpr (!TWO) {
  __builtin_amdgcn_s_barrier();
  if (tr) { t_first = wall_clock64(); c_first = __builtin_amdgcn_s_memtime(); }
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < MB; ++j) acc[i][j] = f32x4{bq[i >> 2][i & 3], bq[i >> 2][i & 3], bq[i >> 2][i & 3], bq[i >> 2][i & 3]};
  }

  bf16x8 wfr[TWO ? 8 : 4], xfr[MB];
  int sa = 0, sw = 1, kt = 0, ti = 0;
  const int grp = wave >> 2;   // = wn
  // slot stamps of slab gs (dbg 5: middle of the second tile, 6: last slab of the first tile, 7: first slab of the second)
  unsigned raw[9], st[9];   // STAMP 1: both ends of slots 0-3 (stamps 0..8), 2: of slots 4-7 (stamps 8..15 + the next slab's first)
  const int gs = p.dbg == 6 ? nk - 1 : p.dbg == 7 ? nk : (my_tiles > 1 ? nk + nk / 2 : nk / 2);
  if constexpr (STAMP) {
#pragma unroll
    for (int i = 0; i < 9; ++i) st[i] = 0;
  }
#define PPS_S(k)                                                                                        \
  if constexpr (STAMP == 1 && (k) <= 8) raw[(k) % 9] = (unsigned)__builtin_amdgcn_s_memtime();                \
  if constexpr (STAMP == 2 && (k) >= 8) raw[((k) - 8) & 7] = (unsigned)__builtin_amdgcn_s_memtime();\
  if constexpr (STAMP == 2 && (k) == 0) raw[8] = (unsigned)__builtin_amdgcn_s_memtime();
#define PPS_COMMIT()                                                              \
  if constexpr (STAMP) {                                                          \
    if (g == gs) { _Pragma("unroll") for (int i = 0; i < (STAMP == 1 ? 9 : 8); ++i) st[i] = raw[i]; } \
    if (STAMP == 2 && g == gs + 1) st[8] = raw[8];                                \
  }

  // ---- the tile's epilogue: accumulators -> (+ bias, activation) -> bf16 -> buffer stores; clears the accumulators and
  //      rotates the source offsets to the next tile ----
  auto epilogue = [&]() {
    long long t_e0 = 0;
    if (tr) t_e0 = wall_clock64();
    const int logical = ti * nblk + lbase;
    int tile_n, tile_m;
    tile_walk(logical, tiles_m, tiles_n, p.walk_pm, tile_m, tile_n);
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    // descriptor over the rows [m0, M) of C: a row >= M lands beyond num_records and is dropped by the memory pipeline
    const long rows_left = (long)p.M - m0;
    const unsigned long nbytes = (unsigned long)rows_left * p.ldc * 2;
    const unsigned nrec = nbytes > 0xFFFFFFF0ul ? 0xFFFFFFF0u : (unsigned)nbytes;
    char* cbase = (char*)p.C + (long)m0 * p.ldc * 2;
    const auto crsrc = __builtin_amdgcn_make_buffer_rsrc(cbase, 0, nrec, 0x00020000);
    // this lane: rows 4 (lane >> 4) + r of every 16-row block of the wave, columns 8 (lane & 15) .. + 7 of the wave's 128
    const unsigned off0 = (unsigned)((((long)(wm * (BM / 4) + 4 * (lane >> 4))) * p.ldc + n0 + wn * 128 + (lane & 15) * 8) * 2);
    const unsigned row_pitch = (unsigned)(p.ldc * 2);
    if (!no_epi) {
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        bf16x8 o;
        if constexpr (ACT == ACT_GELU) {
          f32x2_t g[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) g[j] = f32x2_t{acc[2 * j][mb][r], acc[2 * j + 1][mb][r]};
          gelu_bf16x2_x4(g);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            o[2 * j] = (bf16_t)g[j].x;
            o[2 * j + 1] = (bf16_t)g[j].y;
          }
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) o[j] = (bf16_t)acc[j][mb][r];
        }
        // the row inside the wave's tile goes into the VECTOR offset: (i) the range check of a raw buffer covers voffset only,
        // a row >= M addressed through soffset would be written; (ii) with an SGPR in the soffset field hipcc (ROCm 7.2) assumes
        // the ">64-bit store data, then VALU write of those registers" hazard away and packs the next row into the same four
        // registers with no wait state: on gfx950 that tore the first dword of the previous store in the last four lanes of
        // every 16 whenever the memory pipeline was busy (the persistent stream's LDS-DMA) -- tools/pps_probe.py
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, o), crsrc, off0 + (mb * 16 + r) * row_pitch, 0, STAUX);
      }
    }
    } else {   // diagnostics: every accumulator stays live (a check of two of them lets hipcc delete the MFMAs of all the others)
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < MB; ++j) asm volatile("" ::"v"(acc[i][j]));
    }
    // the next tile starts from its bias (fetched during this tile's last slab: both wave groups reach this point behind the
    // counted wait that retired that slab, and the loads are older than every request the wait leaves in flight)
    asm volatile("" : "+v"(bq[0]), "+v"(bq[1]));
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < MB; ++j) acc[i][j] = f32x4{bq[i >> 2][i & 3], bq[i >> 2][i & 3], bq[i >> 2][i & 3], bq[i >> 2][i & 3]};
    ++ti;
    // the finished tile's offset set now belongs to tile ti + 1 (same parity); with no such tile it keeps rows that exist, which is
    // all the stream's surplus requests need
    if (ti + 1 < my_tiles) {
      unsigned na[GA], nw[GW];
      setup((ti + 1) * nblk + lbase, na, nw);
      const bool into_odd = (ti & 1) == 0;
#pragma unroll
      for (int i = 0; i < GA; ++i) { aofO[i] = into_odd ? na[i] : aofO[i]; aofE[i] = into_odd ? aofE[i] : na[i]; }
#pragma unroll
      for (int i = 0; i < GW; ++i) { wofO[i] = into_odd ? nw[i] : wofO[i]; wofE[i] = into_odd ? wofE[i] : nw[i]; }
    }
    if (tr) t_epi += wall_clock64() - t_e0;
  };

  // LOAD slot Q of slab g: fragment reads of MFMA group Q (k-step Q >> 1, column half Q & 1: W blocks 4 (Q & 1) .. + 3; the
  // activation blocks of the k-step are read in the first half) + the ring's requests: W_{g+1} during k-step 0, A_{g+2} during
  // k-step 1, two DMA instructions per slot; the bias of the finishing tile in slot 1 of its last slab
#define PPS_LOAD(Q)                                                                                                 \
  {                                                                                                                 \
    constexpr int ks_ = (Q) >> 1, half_ = (Q)&1;                                                                    \
    if (half_ == 0) {                                                                                               \
      _Pragma("unroll") for (int jj = 0; jj < MB; ++jj) xfr[jj] = __builtin_bit_cast(bf16x8, xa[jj * 128 + (ks_ ? frag1 : frag0)]); \
    }                                                                                                               \
    _Pragma("unroll") for (int nb = 0; nb < 4; ++nb)                                                                \
        wfr[nb] = __builtin_bit_cast(bf16x8, wa[(half_ * 4 + nb) * 128 + (ks_ ? frag1 : frag0)]);                   \
    if (ks_ == 0) {                                                                                                 \
      {                                                                                                             \
        const char* wb = w_cur ? gW + (long)(kt + 1) * (BK * 2) : gW;                                               \
        _Pragma("unroll") for (int i2 = half_ * 2; i2 < half_ * 2 + 2; ++i2)                                        \
            dma(w_even ? wofE[i2] : wofO[i2], wb, lds_unit(wslot, i2));                                             \
      }                                                                                                             \
      if (half_ == 1 && last_k && has_bias) {                                                                       \
        const float* bp = p.bias + (tile_col(ti + 1 < my_tiles ? (ti + 1) * nblk + lbase : lbase) * BN + wn * 128 + (lane & 15) * 8);       \
        /* read-write operands: the loaded value stays in the registers bq already lives in, so the join behind this conditional   \
           block needs no copy of a register whose data has not landed yet (the compiler cannot see the counted wait that covers  \
           these loads); the epilogue re-defines bq behind that wait before its first use */                                  \
        asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %2, off offset:16"                 \
                     : "+v"(bq[0]), "+v"(bq[1]) : "v"(bp) : "memory");                                              \
      }                                                                                                             \
    } else {                                                                                                        \
      {                                                                                                             \
        const char* ab = gA + (long)(a_cur ? kt + 2 : kt + 2 - nk) * (BK * 2);                                      \
        _Pragma("unroll") for (int i2 = half_ * ((GA + 1) / 2); i2 < (half_ ? GA : (GA + 1) / 2); ++i2)             \
            dma(a_even ? aofE[i2] : aofO[i2], ab, lds_unit(aslot, i2));                                             \
      }                                                                                                             \
    }                                                                                                               \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                              \
    /* every fragment register of this slot is "used" here, in the LOAD slot: hipcc cannot see the asm wait above, and what it  \
       otherwise does is put its own s_waitcnt lgkmcnt(3 / 2 / 1 / 0) BETWEEN the MFMAs of the next slot (one per W block as   \
       their operands "arrive"): four sequencer stalls per 16 MFMAs, 3 200 instead of 2 600 cycles per slab */            \
    if (half_ == 0) touch_regs(xfr);                                                                                \
    touch_regs(wfr);                                                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                                              \
  }
#define PPS_MMA(Q)                                                                                                  \
  {                                                                                                                 \
    constexpr int half_ = (Q)&1;                                                                                    \
    __builtin_amdgcn_s_setprio(1);                                                                                  \
    _Pragma("unroll") for (int nb = 0; nb < 4; ++nb)                                                                \
      _Pragma("unroll") for (int jj = 0; jj < MB; ++jj)                                                             \
        acc[half_ * 4 + nb][jj] = SVT_MFMA_16x16x32(xfr[jj], wfr[nb], acc[half_ * 4 + nb][jj]);                     \
    __builtin_amdgcn_s_setprio(0);                                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                              \
  }
  // retire slab g: everything but the A unit requested during this slab (A_{g+2}) has landed -- W_{g+1}, A_{g+1}, the bias
  // loads of a tile's last slab and the previous tile's stores are all older.  The requests are UNCONDITIONAL: the last two slabs of
  // the stream ask for three units nobody reads (rows of the last tile, into ring slots that are dead by then) instead of carrying
  // "is there a next unit" branches through every LOAD slot (five scalar branches per slot made the slot longer than the partner's
  // 16 MFMAs); they are drained by one vmcnt(0) behind the last epilogue, before the workgroup gives its LDS back.
#define PPS_RETIRE()                                                                                                \
  {                                                                                                                 \
    wait_vm<GA>();                                                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                              \
  }
#define PPS_SLAB_VARS()                                                                                             \
  const uint4* xa = lds + sa * SLOT + xoff;                                                                         \
  const uint4* wa = lds + sw * SLOT + woff;                                                                         \
  const bool w_cur = kt + 1 < nk, a_cur = kt + 2 < nk, last_k = kt + 1 == nk;                                       \
  const bool t_even = (ti & 1) == 0, w_even = w_cur == t_even, a_even = a_cur == t_even;                            \
  const int wslot = sa + 3 >= NSLOT ? sa + 3 - NSLOT : sa + 3, aslot = sa + 4 >= NSLOT ? sa + 4 - NSLOT : sa + 4;
#define PPS_ADVANCE()                                                                                               \
  sa = sa + 2 >= NSLOT ? sa + 2 - NSLOT : sa + 2;                                                                   \
  sw = sw + 2 >= NSLOT ? sw + 2 - NSLOT : sw + 2;                                                                   \
  if (++kt == nk) kt = 0;

  if constexpr (TWO) {
    // ---- two slots per slab (round 5) ----
    // A slab is two intervals, S_g .. M_g .. S_{g+1}, with ONE barrier at each bound.  Waves 0-3 (columns 0-127) multiply k-step 0 and
    // read k-step 1 in the first interval, multiply k-step 1 and read the NEXT slab's k-step 0 in the second; waves 4-7 (columns
    // 128-255) read and then multiply k-step 0 in the first, k-step 1 in the second: on every SIMD one wave issues 6 MB MFMAs
    // (24 / 32) while its partner reads MB + 8 fragments and issues the ring's requests.  Half the barriers, half the fragment
    // waits and half the slot turn-arounds of the four-slot schedule per slab.
    // Because waves 0-3 read slab g + 1 half a slab before waves 4-7 do, the two wave groups need different things at different
    // times, and W is therefore kept in HALF units (the 128 columns of one group): with g the slab,
    //    needed by M_g:      A_{g+1}, W^0_{g+1}            needed by S_{g+1}:   W^1_{g+1}
    //    free from S_g:      the tiles of A_{g-1}, W^1_{g-1}   free from M_g:   the tile of W^0_g
    // so the first interval of slab g requests W^1_{g+1} and the first half of A_{g+2}, the second W^0_{g+2} and the rest of A_{g+2}:
    // every request is issued a whole slab before its first read, out of 3 A tiles + 2 + 2 W half tiles = exactly the 160 KiB (BM = 256).
    constexpr int GA1 = (GA + 1) / 2, GA2 = GA - GA1;
    int a0 = 0, w0s = 0;   // A ring slot of slab g; W ring slot (both halves) of slab g
    auto frag_read = [&](int aslot_, int wslot_, int ks) {
      const uint4* xa = lds + aslot_ * (A_BYTES / 16) + xoff;
      const uint4* wa = lds + (3 * A_BYTES + (unsigned)(grp * 2 + wslot_) * WH_BYTES) / 16;
#pragma unroll
      for (int jj = 0; jj < MB; ++jj) xfr[jj] = __builtin_bit_cast(bf16x8, xa[jj * 128 + (ks ? frag1 : frag0)]);
#pragma unroll
      for (int nb = 0; nb < 8; ++nb) wfr[nb] = __builtin_bit_cast(bf16x8, wa[nb * 128 + (ks ? frag1 : frag0)]);
    };
    auto frag_wait = [&]() {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      touch_regs(xfr);
      touch_regs(wfr);
      __builtin_amdgcn_sched_barrier(0);
    };
    auto mma = [&]() {
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int nb = 0; nb < 8; ++nb)
#pragma unroll
        for (int jj = 0; jj < MB; ++jj) acc[nb][jj] = SVT_MFMA_16x16x32(xfr[jj], wfr[nb], acc[nb][jj]);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
    };
    // the requests of slab (kt_, ti_): first interval (W^1 of the next slab, first half of A two slabs on) / second interval
    auto req1 = [&](int kt_, int ti_, int a0_, int w0_) {
      const bool t_even = (ti_ & 1) == 0;
      const bool w_cur = kt_ + 1 < nk, a_cur = kt_ + 2 < nk;
      const bool w_even = w_cur == t_even, a_even = a_cur == t_even;
      const char* wb = gW + (long)(w_cur ? kt_ + 1 : 0) * (BK * 2);
      const char* ab = gA + (long)(a_cur ? kt_ + 2 : kt_ + 2 - nk) * (BK * 2);
      const int aslot = a0_ + 2 >= 3 ? a0_ - 1 : a0_ + 2;
      if (kt_ + 1 == nk && has_bias) {   // the next tile's bias: older than this interval's requests, covered by the wait in front of M_g
        const float* bp = p.bias + (tile_col(ti_ + 1 < my_tiles ? (ti_ + 1) * nblk + lbase : lbase) * BN + wn * 128 + (lane & 15) * 8);
        asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %2, off offset:16"
                     : "+v"(bq[0]), "+v"(bq[1]) : "v"(bp) : "memory");
      }
#pragma unroll
      for (int i = 2; i < 4; ++i) dma(w_even ? wofE[i] : wofO[i], wb, lds_w2(w0_ ^ 1, i));
#pragma unroll
      for (int i = 0; i < GA1; ++i) dma(a_even ? aofE[i] : aofO[i], ab, lds_a2(aslot, i));
    };
    auto req2 = [&](int kt_, int ti_, int a0_, int w0_) {
      const bool t_even = (ti_ & 1) == 0;
      const bool w_cur = kt_ + 2 < nk, a_cur = kt_ + 2 < nk;
      const bool w_even = w_cur == t_even, a_even = a_cur == t_even;
      const char* wb = gW + (long)(w_cur ? kt_ + 2 : kt_ + 2 - nk) * (BK * 2);
      const char* ab = gA + (long)(a_cur ? kt_ + 2 : kt_ + 2 - nk) * (BK * 2);
      const int aslot = a0_ + 2 >= 3 ? a0_ - 1 : a0_ + 2;
#pragma unroll
      for (int i = 0; i < 2; ++i) dma(w_even ? wofE[i] : wofO[i], wb, lds_w2(w0_, i));
#pragma unroll
      for (int i = GA1; i < GA; ++i) dma(a_even ? aofE[i] : aofO[i], ab, lds_a2(aslot, i));
    };
    __builtin_amdgcn_s_barrier();   // the head has landed for everyone
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < MB; ++j) acc[i][j] = f32x4{bq[i >> 2][i & 3], bq[i >> 2][i & 3], bq[i >> 2][i & 3], bq[i >> 2][i & 3]};
    if (grp == 0) {
      frag_read(0, 0, 0);
      frag_wait();
      for (int g = 0; g < G; ++g) {
        const bool last_k = kt + 1 == nk;
        __builtin_amdgcn_s_barrier();                       // S_g
        mma();                                             // k-step 0
        frag_read(a0, w0s, 1);
        req1(kt, ti, a0, w0s);
        frag_wait();
        wait_vm<2 + GA1>();                                // A_{g+1} and W^0_{g+1} (requested a slab ago) have landed
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();                       // M_g
        mma();                                             // k-step 1
        req2(kt, ti, a0, w0s);                              // with this slab's indices, in front of the epilogue's stores
        const int a1 = a0 + 1 == 3 ? 0 : a0 + 1;
        if (last_k) epilogue();
        if (g + 1 < G) { frag_read(a1, w0s ^ 1, 0); frag_wait(); }
        if (last_k && !no_epi) wait_vm<2 + GA2 + MB * 4>(); else wait_vm<2 + GA2>();   // W^1_{g+1} has landed
        __builtin_amdgcn_sched_barrier(0);
        a0 = a1; w0s ^= 1;
        if (++kt == nk) kt = 0;
      }
    } else {
      for (int g = 0; g < G; ++g) {
        const bool last_k = kt + 1 == nk;
        __builtin_amdgcn_s_barrier();                       // S_g
        frag_read(a0, w0s, 0);
        req1(kt, ti, a0, w0s);
        frag_wait();
        mma();
        wait_vm<2 + GA1>();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();                       // M_g
        frag_read(a0, w0s, 1);
        req2(kt, ti, a0, w0s);
        frag_wait();
        mma();
        if (last_k) epilogue();
        if (last_k && !no_epi) wait_vm<2 + GA2 + MB * 4>(); else wait_vm<2 + GA2>();
        __builtin_amdgcn_sched_barrier(0);
        a0 = a0 + 1 == 3 ? 0 : a0 + 1; w0s ^= 1;
        if (++kt == nk) kt = 0;
      }
    }
  } else
  if constexpr (HB) {
    // Four barriers per slab.  An interval between two barriers holds a LOAD slot AND an MFMA slot of every wave, in opposite
    // order for the two groups: waves 0-3 multiply group q and then read group q + 1, waves 4-7 read group q and then multiply
    // it, so on a SIMD the halves still alternate, but a LOAD slot that runs longer than its partner's 16 MFMAs (the eight-read
    // slots do) is paid back inside the interval instead of at a barrier of its own.  Both groups retire the slab in front of
    // barrier 4 (waves 0-3 behind their last LOAD slot, waves 4-7 behind MFMA group 2 with half of the A requests of the slab
    // still to be issued): behind that barrier waves 0-3 read the next slab.
    constexpr int GA_HALF = (GA + 1) / 2;
    if (grp == 0) {
      bool pending = false;
      {
        PPS_SLAB_VARS()
        PPS_LOAD(0)
      }
      for (int g = 0; g < G; ++g) {
        PPS_SLAB_VARS()
        __builtin_amdgcn_s_barrier();
        PPS_MMA(0) PPS_LOAD(1) __builtin_amdgcn_s_barrier();
        PPS_MMA(1) PPS_LOAD(2) __builtin_amdgcn_s_barrier();
        PPS_MMA(2) PPS_LOAD(3)
        PPS_RETIRE()
        __builtin_amdgcn_s_barrier();
        PPS_MMA(3)
        pending = last_k;
        PPS_ADVANCE()
        if (pending) epilogue();
        if (g + 1 < G) {
          PPS_SLAB_VARS()
          PPS_LOAD(0)
        }
      }
    } else {
      for (int g = 0; g < G; ++g) {
        PPS_SLAB_VARS()
        __builtin_amdgcn_s_barrier();
        PPS_LOAD(0) PPS_MMA(0) __builtin_amdgcn_s_barrier();
        PPS_LOAD(1) PPS_MMA(1) __builtin_amdgcn_s_barrier();
        PPS_LOAD(2) PPS_MMA(2)
        wait_vm<GA_HALF>();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        PPS_LOAD(3) PPS_MMA(3)
        if (last_k) epilogue();
        PPS_ADVANCE()
      }
    }
  } else
  if (grp == 0) {
    bool pending = false;   // the previous slab closed a tile: its epilogue runs in front of this slab's first LOAD slot
    for (int g = 0;; ++g) {
      if (pending) epilogue();
      if (g == G) break;
      PPS_SLAB_VARS()
      PPS_S(0) PPS_LOAD(0) PPS_S(1) __builtin_amdgcn_s_barrier(); PPS_S(2) PPS_MMA(0) PPS_S(3) __builtin_amdgcn_s_barrier();
      PPS_S(4) PPS_LOAD(1) PPS_S(5) __builtin_amdgcn_s_barrier(); PPS_S(6) PPS_MMA(1) PPS_S(7) __builtin_amdgcn_s_barrier();
      PPS_S(8) PPS_LOAD(2) PPS_S(9) __builtin_amdgcn_s_barrier(); PPS_S(10) PPS_MMA(2) PPS_S(11) __builtin_amdgcn_s_barrier();
      PPS_S(12) PPS_LOAD(3) PPS_S(13) __builtin_amdgcn_s_barrier(); PPS_S(14) PPS_MMA(3)
      PPS_RETIRE()
      PPS_S(15)
      __builtin_amdgcn_s_barrier();
      PPS_COMMIT()
      pending = last_k;
      PPS_ADVANCE()
    }
  } else {
    __builtin_amdgcn_s_barrier();  // one slot behind
    for (int g = 0; g < G; ++g) {
      PPS_SLAB_VARS()
      PPS_S(0) PPS_LOAD(0) PPS_S(1) __builtin_amdgcn_s_barrier(); PPS_S(2) PPS_MMA(0) PPS_S(3) __builtin_amdgcn_s_barrier();
      PPS_S(4) PPS_LOAD(1) PPS_S(5) __builtin_amdgcn_s_barrier(); PPS_S(6) PPS_MMA(1) PPS_S(7) __builtin_amdgcn_s_barrier();
      PPS_S(8) PPS_LOAD(2) PPS_S(9) __builtin_amdgcn_s_barrier(); PPS_S(10) PPS_MMA(2) PPS_S(11) __builtin_amdgcn_s_barrier();
      PPS_S(12) PPS_LOAD(3)
      PPS_RETIRE()
      PPS_S(13)
      __builtin_amdgcn_s_barrier();
      PPS_S(14) PPS_MMA(3)
      if (last_k) epilogue();
      PPS_S(15)
      if (g + 1 < G) __builtin_amdgcn_s_barrier();
      PPS_COMMIT()
      PPS_ADVANCE()
    }
  }
  long long t_loop = 0;
  if (tr) t_loop = wall_clock64();
  // the surplus requests of the stream's tail: nothing may land in LDS after the workgroup is gone.  They are older than the last
  // epilogue's MB * 4 stores, which need not be waited for (VMEM operations retire in order).
  if (no_epi) wait_vm<0>(); else wait_vm<MB * 4>();
  if (tr && lane == 0 && (wave & 3) == 0) {
    const long long t_end = t_loop;
    long long* o = p.trace + ((long)blockIdx.x * 2 + (wave >> 2)) * 8;
    o[0] = t_begin; o[1] = t_first; o[2] = t_end - t_first - t_epi; o[3] = t_epi; o[4] = t_end; o[5] = my_tiles;
    o[6] = __builtin_amdgcn_s_memtime() - c_first; o[7] = BM;
  }
  if constexpr (STAMP) {
    if (tr && lane == 0) {
      long long* o = p.trace + 65536 + ((long)blockIdx.x * 8 + wave) * 32;
#pragma unroll
      for (int i = 0; i < 9; ++i) o[i] = st[i];
      o[9] = G; o[10] = gs; o[11] = STAMP;
    }
  }
#undef PPS_S
#undef PPS_COMMIT
#undef PPS_LOAD
#undef PPS_MMA
#undef PPS_RETIRE
#undef PPS_SLAB_VARS
#undef PPS_ADVANCE
}

template <int BM, int ACT, int STAUX = 0, int STAMP = 0, bool HB = false, bool TWO = false>
int launch_pps_t(const GemmArgs& a, hipStream_t s) {
  const int tiles_m = (a.M + BM - 1) / BM, tiles_n = a.N / 256;
  const int ntiles = tiles_m * tiles_n;
  const int nblk = ntiles < g_gemm_persist_wgs ? ((ntiles + 7) / 8) * 8 : g_gemm_persist_wgs;   // (svt_debug_set key 37: workgroups of a persistent launch)
  const size_t lds_bytes = 5 * 32768;
  GemmArgs aw = a;
  aw.walk_pm = gemm_walk_pm(a, BM);
  if (int r_ = ensure_dyn_lds((const void*)gemm_pps_kernel<BM, ACT, STAUX, STAMP, HB, TWO>, (int)lds_bytes)) return r_;
  const double flops = 2.0 * a.M * (double)a.N * a.K;
  const double bytes = ((double)a.M * a.K + (double)a.N * a.K) * 2 + (double)a.M * a.N * 2;
  prof_begin(s);
  hipLaunchKernelGGL((gemm_pps_kernel<BM, ACT, STAUX, STAMP, HB, TWO>), dim3(nblk), dim3(512), lds_bytes, s, aw, tiles_n, ntiles);
  prof_end(s, flops, bytes, 0);
  SVT_LAUNCH_CHECK();
  return 0;
}

}  // namespace

// span in bytes of the rows of A (implicit-conv rows overlap): the last row's end
static unsigned long a_span_bytes(const GemmArgs& a) {
  const long last = (long)a.M - 1;
  return (unsigned long)(((last / a.a_rpb) * a.a_bstride + (last % a.a_rpb) * a.a_rstride + a.K) * 2);
}

int g_pps_two_slots = 0;
bool gemm_pps_eligible(const GemmArgs& a) {
  return !a.gen && a.nz == 1 && !a.resid && !a.out_f32 && !a.planes && a.alpha == 1.f && (a.act == ACT_NONE || a.act == ACT_GELU) &&
         a.K % 64 == 0 && a.K >= 128 && a.N % 256 == 0 && a.M >= 128 && a.c_vec && a.ldc % 8 == 0 && a.c_z1 == 0 && a.c_z2 == 0 &&
         a.a_z1 == 0 && a.a_z2 == 0 && a.w_z1 == 0 && a.w_z2 == 0 && a_span_bytes(a) < 0xFFFF0000ul &&
         (unsigned long)a.N * a.ldw * 2 < 0xFFFF0000ul && ((uintptr_t)a.A & 15) == 0 && ((uintptr_t)a.W & 15) == 0 &&
         (a.a_rstride & 7) == 0 && (a.a_bstride & 7) == 0 && (a.ldw & 7) == 0;
}

#ifdef SVT_DIAG
// Diagnostic build (make DIAG=1): slot-stamp instantiations, the eight-barrier form and the default store policy, for
// tools/gemm_trace.py --slots and the A/B keys of svt_debug_set (3: 50 / 70, 15, 16).  The shipped library holds the six
// dispatched instantiations only (sc1 stores, four barriers per slab).
template <int STAUX>
static int launch_pps_aux(const GemmArgs& a, int bm, hipStream_t s) {
  if constexpr (STAUX == 16) {   // slot stamps: the dispatched store policy, 256-row tiles only
    if (a.trace && bm == 256 && a.dbg >= 5 && a.dbg <= 7) {
      if (a.stamp_ends) return a.act == ACT_GELU ? launch_pps_t<256, ACT_GELU, 16, 2>(a, s) : launch_pps_t<256, ACT_NONE, 16, 2>(a, s);
      return a.act == ACT_GELU ? launch_pps_t<256, ACT_GELU, 16, 1>(a, s) : launch_pps_t<256, ACT_NONE, 16, 1>(a, s);
    }
    if (a.trace && bm == 192 && a.act == ACT_NONE && a.dbg >= 5 && a.dbg <= 7)
      return a.stamp_ends ? launch_pps_t<192, ACT_NONE, 16, 2>(a, s) : launch_pps_t<192, ACT_NONE, 16, 1>(a, s);
  }
  if constexpr (STAUX == 16) {
    // four barriers per slab: FFN-1 78.9 -> 73.6 us, large FFN-1 285 -> 282, conv1 762 -> 778, the plain tiles equal; end to end
    // C2 5 881 / 5 903 / 5 943 clips/s for never / GELU-short-K only / always, C3 2 432 / 2 445 / 2 443
    if (g_pps_half_barriers == 1 || (g_pps_half_barriers == 2 && a.act == ACT_GELU && a.K <= 1024)) {
      if (a.act == ACT_GELU) {
        if (bm == 256) return launch_pps_t<256, ACT_GELU, 16, 0, true>(a, s);
        if (bm == 192) return launch_pps_t<192, ACT_GELU, 16, 0, true>(a, s);
        return launch_pps_t<128, ACT_GELU, 16, 0, true>(a, s);
      }
      if (bm == 256) return launch_pps_t<256, ACT_NONE, 16, 0, true>(a, s);
      if (bm == 192) return launch_pps_t<192, ACT_NONE, 16, 0, true>(a, s);
      return launch_pps_t<128, ACT_NONE, 16, 0, true>(a, s);
    }
  }
  if (a.act == ACT_GELU) {
    if (bm == 256) return launch_pps_t<256, ACT_GELU, STAUX>(a, s);
    if (bm == 192) return launch_pps_t<192, ACT_GELU, STAUX>(a, s);
    return launch_pps_t<128, ACT_GELU, STAUX>(a, s);
  }
  if (bm == 256) return launch_pps_t<256, ACT_NONE, STAUX>(a, s);
  if (bm == 192) return launch_pps_t<192, ACT_NONE, STAUX>(a, s);
  return launch_pps_t<128, ACT_NONE, STAUX>(a, s);
}

int launch_gemm_pps(const GemmArgs& a, int bm, hipStream_t s, int store_policy) {
  // svt_debug_set key 28: the two-slot schedule (round 5).  0 (default) = never, 1 = wherever it applies, 2 = the one launch family it
  // measured faster on in isolation: the FFN-1 of the LARGE models (N = 4096, K = 1024, GELU, 256-row tiles: 284 -> 273 us; every C2 shape
  // is 1-6 % slower with it) -- which did not carry over to the step: C3 2 476 clips/s with it against 2 505 without
  // (profiles/r05_gemm_two_slot_ab.txt)
  const bool two = g_pps_two_slots == 1 || (g_pps_two_slots == 2 && a.act == ACT_GELU && bm == 256 && a.N == 4096 && a.K == 1024 && a.M >= 16384);
  if (two && (bm == 192 || bm == 256) && !a.trace) {
    if (a.act == ACT_GELU) return bm == 256 ? launch_pps_t<256, ACT_GELU, 16, 0, false, true>(a, s) : launch_pps_t<192, ACT_GELU, 16, 0, false, true>(a, s);
    return bm == 256 ? launch_pps_t<256, ACT_NONE, 16, 0, false, true>(a, s) : launch_pps_t<192, ACT_NONE, 16, 0, false, true>(a, s);
  }
  // (nt stores, aux = 2, were measured in round 3 and never won: their instantiations are gone)
  if (store_policy == 2) return launch_pps_aux<16>(a, bm, s);   // sc1 (write-through)
  return launch_pps_aux<0>(a, bm, s);
}
#else
// The dispatched form: write-through (sc1) stores, four barriers per slab (measured against the eight-barrier form and the default
// store policy in round 3: profiles/r03_gemm_pps_slots.txt; those, the slot-stamp instantiations and round 5's two-slot schedule
// (svt_debug_set key 28, measured slower: profiles/r05_gemm_two_slot_ab.txt) are built by `make DIAG=1`).
int launch_gemm_pps(const GemmArgs& a, int bm, hipStream_t s, int /*store_policy*/) {
  if (a.act == ACT_GELU) {
    if (bm == 256) return launch_pps_t<256, ACT_GELU, 16, 0, true>(a, s);
    if (bm == 192) return launch_pps_t<192, ACT_GELU, 16, 0, true>(a, s);
    return launch_pps_t<128, ACT_GELU, 16, 0, true>(a, s);
  }
  if (bm == 256) return launch_pps_t<256, ACT_NONE, 16, 0, true>(a, s);
  if (bm == 192) return launch_pps_t<192, ACT_NONE, 16, 0, true>(a, s);
  return launch_pps_t<128, ACT_NONE, 16, 0, true>(a, s);
}
#endif

}  // namespace svt

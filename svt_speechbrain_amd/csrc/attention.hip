// Fused self/cross attention for gfx950 (bf16 operands, fp32 softmax state): softmax(scale * Q K^T) V
// without materialising the (T x T) scores.  No mask (the reference passes none); only
// the key tail of the last tile is masked.  T is arbitrary (T = 499 for 10 s clips), head_dim 64
// (wav2vec2/HuBERT) or 128 (RCA fusion).
//
// Mapping (wave64, v_mfma_f32_32x32x16_bf16):
//   * block = 4 waves = 128 queries of one (clip, head), or 8 waves = 256 queries when the launch has enough workgroups
//     (half the K / V traffic per query); each wave owns 32 queries for the whole sweep.
//   * K and V tiles of 64 keys are moved global -> LDS by LDS-DMA (no staging registers, no ds_write) into a row-major
//     image with an XOR swizzle of the 16-byte chunks, applied on the SOURCE address (8 consecutive lanes fetch one
//     128-byte row: one request per line); two stages, the next tile is in flight during the current tile's MFMAs,
//     one barrier per tile.
//   * S^T = K Q^T is computed with keys on the MFMA rows, so a lane holds 16 keys x 1 query per 32-key
//     block: the row max / row sum are in-register reductions plus ONE cross-lane exchange
//     (lane ^ 32).  The S accumulator, converted pairwise to bf16, is directly the B operand of
//     O^T += V^T P^T (k order permuted consistently on the V side) — P never touches LDS.
//   * V is read TRANSPOSED out of its row-major LDS tile by ds_read_b64_tr_b16 (4 keys x 16 d blocks delivered
//     column-major): the 4-key runs of the permuted k order are exactly two such blocks per operand, so no
//     transposed copy of V ever exists in memory.
//   * softmax: scale folded into the exponent FMA, running max raised only when a row grows by > 2^8 (deferred
//     rescale), MFMA accumulators kept in VGPRs (-mllvm -amdgpu-mfma-vgpr-form, see Makefile).
#include "common.h"

namespace svt {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_t;

// BIAS: WavLM's gated relative position bias, scores += gate[b,h,q] * pb[h][key - q + T - 1] (pb row of 2T-1 floats kept
// in LDS; added in the scaled log2 domain before the row max).
// STAMP (diagnostics, svt_debug_set key 18, tools/attn_bench.py --stamps): s_memtime at seven points of key tile 4, per wave, written over the
// head of O
template <int DH, bool BIAS = false, int NW = 4, bool STAMP = false>
__global__ __launch_bounds__(64 * NW) void flash_attn_kernel(const bf16_t* __restrict__ Q, long ldq, long q_bstride,
                                                         const bf16_t* __restrict__ K, long ldk, long k_bstride,
                                                         const bf16_t* __restrict__ V, int /*unused*/, bf16_t* __restrict__ O,
                                                         long ldo, long o_bstride, int T, int H, float c,
                                                         const float* __restrict__ gate, const float* __restrict__ pbias) {
  constexpr int KSD = DH / 16;   // k-steps over head_dim for S
  constexpr int DB = DH / 32;    // 32-row blocks of O^T
  constexpr int CPR = DH / 8;    // 16-byte chunks per K row
  constexpr int RB = 2 * DH;     // bytes per K / V row in LDS
  // two stages of { K tile, V tile }, each tile row-major [key][slot] with slot = chunk ^ swizzle(key); filled by LDS-DMA
  constexpr int TILE16 = 64 * CPR;  // uint4 per tile
  __shared__ __attribute__((aligned(16))) uint4 KV[2][2 * TILE16];
  extern __shared__ float pbl[];  // BIAS: this head's 2T-1 bias values

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.z, h = blockIdx.y;
  const int q0 = blockIdx.x * (32 * NW) + wave * 32;  // NW waves x 32 queries: every workgroup streams all of K and V once
  const bf16_t* Qb = Q + (long)b * q_bstride + (long)h * DH;
  const bf16_t* Kb = K + (long)b * k_bstride + (long)h * DH;
  const bf16_t* Vb = V + (long)b * k_bstride + (long)h * DH;

  // Q fragments (B operand of S^T = K Q^T): lane holds Q[q0 + (lane&31)][ks*16 + 8*(lane>>5) .. +7]
  bf16x8 qf[KSD];
  {
    int q = q0 + (lane & 31);
    if (q > T - 1) q = T - 1;
    const bf16_t* qp = Qb + (long)q * ldq + 8 * (lane >> 5);
#pragma unroll
    for (int ks = 0; ks < KSD; ++ks) qf[ks] = *(const bf16x8*)(qp + ks * 16);
  }

  float gq = 0.f;   // gate of this lane's query, pre-multiplied by log2(e)
  int qrel = 0;     // T - 1 - query
  if constexpr (BIAS) {
    for (int i = tid; i < 2 * T - 1; i += 64 * NW) pbl[i] = pbias[(long)h * (2 * T - 1) + i];
    int q = q0 + (lane & 31);
    if (q > T - 1) q = T - 1;
    gq = gate[((long)b * H + h) * T + q] * 1.44269504088896340736f;
    qrel = T - 1 - q;
  }
  // LDS-DMA staging: a wave instruction moves 64 x 16 B = 1 KB = ROWS_PER_DMA whole rows; lane (row r = lane / CPR,
  // slot = lane % CPR) fetches chunk slot ^ g(key) of its row, so the linear LDS image holds chunk c of a key at slot
  // c ^ g(key) -- the same swizzled row-major layout the MFMA operand reads below expect (K: g = (key>>1)&7 for
  // 128-byte rows, key&15 for 256-byte rows; V: (key>>1 & 1) << 2, resp. (key & 3) << 2).  No staging registers, no
  // ds_write; the next tile is in flight while the current one is multiplied (two stages, one barrier per tile).
  constexpr int ROWS_PER_DMA = 64 / CPR;                 // 8 (dh 64) / 4 (dh 128)
  constexpr int DMA_PER_WAVE = 64 / ROWS_PER_DMA / NW;   // instructions per wave per tile and operand (2 / 4 with four waves)
  static_assert(DMA_PER_WAVE >= 1, "too many waves for the tile's fill");
  int dkey[DMA_PER_WAVE], kch[DMA_PER_WAVE], vch[DMA_PER_WAVE];
#pragma unroll
  for (int i = 0; i < DMA_PER_WAVE; ++i) {
    const int kk = (wave * DMA_PER_WAVE + i) * ROWS_PER_DMA + lane / CPR;  // key within the tile
    const int slot = lane % CPR;
    const int g = (DH == 64) ? ((kk >> 1) & 7) : (kk & 15);
    const int sw = (DH == 64) ? (((kk >> 1) & 1) << 2) : ((kk & 3) << 2);
    dkey[i] = kk;
    kch[i] = (slot ^ g) * 8;
    vch[i] = (slot ^ sw) * 8;
  }
  typedef const void __attribute__((address_space(1)))* gptr_t;
  typedef void __attribute__((address_space(3)))* lptr_t;
#define SVT_STAGE_DMA(TILE, STAGE)                                                                                  \
  {                                                                                                                  \
    const int key0_ = (TILE) * 64;                                                                                   \
    _Pragma("unroll") for (int i = 0; i < DMA_PER_WAVE; ++i) {                                                       \
      int key_ = key0_ + dkey[i];                                                                                    \
      if (key_ > T - 1) key_ = T - 1;                                                                                \
      const bf16_t* kp_ = Kb + (long)key_ * ldk;                                                                     \
      __builtin_amdgcn_global_load_lds((gptr_t)(kp_ + kch[i]),                                                       \
                                       (lptr_t)(&KV[STAGE][(wave * DMA_PER_WAVE + i) * 64]), 16, 0, 0);              \
      __builtin_amdgcn_global_load_lds((gptr_t)(kp_ + (Vb - Kb) + vch[i]),                                           \
                                       (lptr_t)(&KV[STAGE][TILE16 + (wave * DMA_PER_WAVE + i) * 64]), 16, 0, 0);     \
    }                                                                                                                \
  }

  f32x16 o[DB];
#pragma unroll
  for (int i = 0; i < DB; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
  float m = -1e30f, l = 0.f;
  const int hh = lane >> 5;
  const int kl = lane & 31;                                     // key within a 32-key block
  const int kg = (DH == 64) ? ((kl >> 1) & 7) : (kl & 15);      // its swizzle (same for both key blocks: 32 % 16 == 0)
  const int ntiles = (T + 63) / 64;
  // transposing-read addresses: lane 16g+4q+p supplies row (key) q, columns 4p..4p+3 of its group's 4 x 16 block
  const char* vbase[DB];
  {
    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    const int key0 = 4 * (g >> 1) + q;
    const int sw = (DH == 64) ? (((q >> 1) & 1) << 2) : (q << 2);
#pragma unroll
    for (int db = 0; db < DB; ++db) {
      const int chunk = db * 4 + 2 * (g & 1) + (pp >> 1);
      vbase[db] = (const char*)&KV[0][TILE16] + key0 * RB + ((chunk ^ sw) * 16) + 8 * (pp & 1);
    }
  }

  constexpr long STAGE_BYTES = (long)2 * TILE16 * 16;
  unsigned stp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define SVT_ATT_S(k) if constexpr (STAMP) { if (tile == 4) stp[k] = (unsigned)__builtin_amdgcn_s_memtime(); }
  SVT_STAGE_DMA(0, 0)
  for (int tile = 0; tile < ntiles; ++tile) {
    const int st = tile & 1;
    if constexpr (STAMP) { if (tile == 5) stp[7] = (unsigned)__builtin_amdgcn_s_memtime(); }
    SVT_ATT_S(0)
    // own fills of this tile have landed (explicit vmcnt(0): hipcc's own placement of that wait is an alias-analysis
    // artefact, not a contract), everyone's have after the barrier, and every wave is done reading the other stage
    // (tile - 1), which the next fill overwrites
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    SVT_ATT_S(1)
    if (tile + 1 < ntiles) {
      if (st) SVT_STAGE_DMA(tile + 1, 0) else SVT_STAGE_DMA(tile + 1, 1)
    }
    SVT_ATT_S(2)
    const uint4* Kf = KV[st];

    f32x16 s[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[kb][r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < KSD; ++ks)
        s[kb] = SVT_MFMA_32x32x16(__builtin_bit_cast(bf16x8, Kf[(kb * 32 + kl) * CPR + ((2 * ks + hh) ^ kg)]), qf[ks], s[kb]);
    }
    // softmax in the scaled log2 domain: p = exp2(s*c - m).  The scale is folded into the exponent FMA, the row max is
    // taken on the raw scores, and the running max is only raised (O and l rescaled) when some row of the wave grew
    // by more than 2^8 (deferred rescale: P stays <= 256, exact in fp32 and safe in bf16); keys >= T are masked in
    // the last tile only.
    if constexpr (BIAS) {
      // x = s*c + gate*log2e * pb[key - q + T - 1]; from here on the scores ARE in the log2 domain (cs = 1 below)
      const int kb0 = tile * 64 + 4 * hh + qrel;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          int idx = kb0 + kb * 32 + (r & 3) + 8 * (r >> 2);
          if (idx > 2 * T - 2) idx = 2 * T - 2;   // keys >= T of the last tile (masked below)
          s[kb][r] = fmaf(s[kb][r], c, gq * pbl[idx]);
        }
    }
    SVT_ATT_S(3)
    const float cs = BIAS ? 1.0f : c;
    if (__builtin_expect(tile * 64 + 64 > T, 0)) {  // wave-uniform, last tile only
      const int kbase = tile * 64 + 4 * hh;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kbase + kb * 32 + (r & 3) + 8 * (r >> 2);
          asm volatile("" : "+v"(s[kb][r]));  // keep the masking inside this (rare) branch instead of 32 selects per tile
          if (key >= T) s[kb][r] = -3e38f;
        }
    }
    float mx = fmaxf(s[0][0], s[1][0]);
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(fmaxf(mx, s[0][r]), s[1][r]);  // v_max3_f32
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64)) * cs;
    if (!__all(mx - m <= 8.0f)) {
      const float mnew = fmaxf(m, mx);
      const float alpha = __builtin_amdgcn_exp2f(m - mnew);
      l *= alpha;
      m = mnew;
#pragma unroll
      for (int i = 0; i < DB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
    }
    SVT_ATT_S(4)
    // exponent arguments and row sums on pairs (v_pk_fma_f32 / v_pk_add_f32: two values per issue slot)
    typedef float f32x2v __attribute__((ext_vector_type(2)));
    f32x2v sum2 = {0.f, 0.f};
    const f32x2v c2 = {cs, cs}, nm2 = {-m, -m};
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const f32x2v e = f32x2v{s[kb][r], s[kb][r + 1]} * c2 + nm2;
        const f32x2v pp = {__builtin_amdgcn_exp2f(e.x), __builtin_amdgcn_exp2f(e.y)};
        s[kb][r] = pp.x;
        s[kb][r + 1] = pp.y;
        sum2 += pp;
      }
    float sum = sum2.x + sum2.y;
    sum += __shfl_xor(sum, 32, 64);
    l += sum;
    SVT_ATT_S(5)
    // P (bf16) fragments straight from the S accumulators: k-step ss of key block kb = regs 8ss..8ss+7
    bf16x8 pf[2][2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int ss = 0; ss < 2; ++ss)
#pragma unroll
        for (int j = 0; j < 8; ++j) pf[kb][ss][j] = (bf16_t)s[kb][ss * 8 + j];
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int ss = 0; ss < 2; ++ss)
        {
          // A operand of O^T += V^T P^T straight from the row-major V tile: two ds_read_b64_tr_b16 (4 keys x 16 d
          // blocks, delivered column-major) give this lane V[key][d] for its d and the permuted k-slots
          // {16s+4h+0..3, 16s+8+4h+0..3}
          const char* vp = vbase[db] + st * STAGE_BYTES + (kb * 32 + ss * 16) * RB;
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t)(vp));
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t)(vp + 8 * RB));
          const s16x8 vv = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
          o[db] = SVT_MFMA_32x32x16(__builtin_bit_cast(bf16x8, vv), pf[kb][ss], o[db]);
        }
    SVT_ATT_S(6)
  }
#undef SVT_ATT_S

  // O[q][d] = o / l ; lane (q = lane&31, hh) holds d = db*32 + (r&3) + 8*(r>>2) + 4*hh
  const int q = q0 + (lane & 31);
  if (q >= T) return;
  const float inv = 1.f / l;
  bf16_t* op = O + (long)b * o_bstride + (long)q * ldo + (long)h * DH;
  // a lane holds 4 consecutive d per (db, g) and its partner lane ^ 32 (same query) the other 4 of the same 8: a half exchange
  // (v_permlane32_swap) per dword gives lanes 0-31 the 8 values of group g and lanes 32-63 those of group g + 1 -- one 16-byte store
  // per pair of groups instead of two 8-byte ones (the store tail of a row-per-lane epilogue is bound by store ISSUE, not bytes)
#pragma unroll
  for (int db = 0; db < DB; ++db)
#pragma unroll
    for (int g = 0; g < 4; g += 2) {
      unsigned w[2][2];   // [group g / g + 1][dword]
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        bf16x4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = (bf16_t)(o[db][(g + u) * 4 + j] * inv);
        const uint2 pk = __builtin_bit_cast(uint2, v);
        w[u][0] = pk.x; w[u][1] = pk.y;
      }
#pragma unroll
      for (int d2 = 0; d2 < 2; ++d2) {
        const auto r = __builtin_amdgcn_permlane32_swap(w[0][d2], w[1][d2], false, false);
        w[0][d2] = r[0]; w[1][d2] = r[1];
      }
      *(uint4*)(op + db * 32 + 8 * (g + hh)) = uint4{w[0][0], w[0][1], w[1][0], w[1][1]};
    }
  if constexpr (STAMP) {
    __syncthreads();
    if (lane == 0) {
      const long wg = ((long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
      unsigned* o32 = (unsigned*)O + (wg * NW + wave) * 16;
      // (the records overwrite the head of O: diagnostics only; the last workgroups' stores may land on top of some of them)
#pragma unroll
      for (int i = 0; i < 8; ++i) o32[i] = stp[i];
      o32[8] = 0x5A5A0000u + (unsigned)wave;
    }
  }
}

#undef SVT_STAGE_DMA

// ---------------------------------------------------------------------------------------------------------------------
// Staggered form of flash_attn_kernel<64, false, 8> (round 4).  What the counters said about that kernel at 32 x 12 heads x 499 frames
// (profiles/r04_attention_pmc.txt): 16 MFMAs (512 cycles of the matrix pipe) and ~150 vector instructions (~550 cycles of the vector
// pipe: 32 quarter-rate exponentials are half of it) per wave and 64-key tile -- the two pipes are BALANCED -- yet a SIMD retires a
// wave-tile only every ~1 340 cycles (matrix pipe busy 0.24 of the kernel, waves parked at waits and barriers for 43 % of their
// cycles): the workgroup's barrier per tile keeps its eight waves in the same phase, so the two waves a workgroup has on a SIMD
// multiply at the same time and exponentiate at the same time, and only the other workgroup of the CU ever fills the idle pipe.
// Here waves 4-7 run HALF A TILE behind waves 0-3, the way the GEMM kernels stagger their wave groups: every wave executes the
// same per-tile sequence S = K Q^T -> softmax -> O += V P, but waves 0-3 pass the tile's barrier in front of S and waves 4-7 between
// the softmax and the P V product of the previous tile -- when one group is in its MFMA phases the SIMD partner is in its softmax.
// The barrier of tile j still means "tile j has landed, and tile j + 1 may be requested": with the late group still reading V of tile
// j - 1 at that point the request needs a third stage (tile j + 1 overwrites tile j - 2): 48 KiB per workgroup, two workgroups per CU.
// The LDS-DMA is issued from inline asm (the compiler tracks the DMA it emits for the builtin and guards later LDS reads it cannot
// prove distinct from the destination with s_waitcnt vmcnt(0) -- here the late group's V reads right behind its requests); the
// explicit vmcnt(0) in front of every barrier is the only wait on it.
__device__ __forceinline__ void attn_dma16(const void* gsrc, unsigned lds_byte_addr) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(lds_byte_addr) : "memory");
}

// PIPE (experiment, built by make DIAG=1 only; svt_debug_set(21, 2)): the P V product of tile j - 1 issued BEHIND the S MFMAs of tile j
// and in front of tile j's softmax, so that inside one wave the matrix pipe works through 8 PV MFMAs while the wave issues the softmax's
// vector instructions; no second accumulator set is needed (P of the previous tile and S of this one coexist anyway, V of tile j - 1 is
// still in its stage).  Measured SLOWER than the plain staggered order: 43.0-43.9 against 41.0-42.1 us at C2, 115-118 against 109-114 at C3.
template <int DH, bool PIPE = false>
__global__ __launch_bounds__(512) void flash_attn_stag_kernel(const bf16_t* __restrict__ Q, long ldq, long q_bstride,
                                                              const bf16_t* __restrict__ K, long ldk, long k_bstride,
                                                              const bf16_t* __restrict__ V, bf16_t* __restrict__ O, long ldo,
                                                              long o_bstride, int T, int H, float c, int nqb) {
  static_assert(DH == 64, "built for head_dim 64");
  constexpr int NW = 8, NST = 3;
  constexpr int KSD = DH / 16, DB = DH / 32, CPR = DH / 8, RB = 2 * DH;
  constexpr int TILE16 = 64 * CPR;                   // uint4 per tile
  constexpr int STAGE16 = 2 * TILE16;                // K tile + V tile
  constexpr long STAGE_BYTES = (long)STAGE16 * 16;
  __shared__ __attribute__((aligned(16))) uint4 KV[NST * STAGE16];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // 1-D grid of NQB x (B * H) workgroups, mapped so that the NQB query blocks of one (clip, head) are blocks L, L + 8, L + 16, ...:
  // consecutive blocks are dealt round-robin over the 8 XCDs, so those land on ONE XCD and the K / V of the head are fetched into
  // one L2 instead of NQB (speed only: profiles/r04_attention_pmc.txt, 147 MB of HBM-side traffic per launch against 98 MB algorithmic)
  const int L = blockIdx.x, BH = (int)gridDim.x / nqb;
  int bh, qblk;
  {
    const int full = (BH / 8) * 8 * nqb;   // blocks covered by whole groups of 8 heads
    if (L < full) { bh = (L & 7) + 8 * (L / (8 * nqb)); qblk = (L >> 3) % nqb; }
    else { const int r = L - full; bh = (BH / 8) * 8 + r / nqb; qblk = r % nqb; }
  }
  const int b = bh / H, h = bh - b * H;
  const int q0 = qblk * (32 * NW) + wave * 32;
  const bf16_t* Qb = Q + (long)b * q_bstride + (long)h * DH;
  const bf16_t* Kb = K + (long)b * k_bstride + (long)h * DH;
  const bf16_t* Vb = V + (long)b * k_bstride + (long)h * DH;

  bf16x8 qf[KSD];
  {
    int q = q0 + (lane & 31);
    if (q > T - 1) q = T - 1;
    const bf16_t* qp = Qb + (long)q * ldq + 8 * (lane >> 5);
#pragma unroll
    for (int ks = 0; ks < KSD; ++ks) qf[ks] = *(const bf16x8*)(qp + ks * 16);
  }
  // LDS-DMA source mapping of flash_attn_kernel: one instruction per wave, tile and operand (8 rows x 128 B)
  const int dkey = wave * 8 + lane / CPR;
  const int slot = lane % CPR;
  const int kch = (slot ^ ((dkey >> 1) & 7)) * 8;
  const int vch = (slot ^ (((dkey >> 1) & 1) << 2)) * 8;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(void __attribute__((address_space(3)))*)KV);
  auto stage_dma = [&](int tile, int stage) {
    int key = tile * 64 + dkey;
    if (key > T - 1) key = T - 1;
    const bf16_t* kp = Kb + (long)key * ldk;
    const unsigned dst = lds0 + (unsigned)((stage * STAGE16 + wave * 64) * 16);
    attn_dma16(kp + kch, dst);
    attn_dma16(kp + (Vb - Kb) + vch, dst + TILE16 * 16);
  };

  f32x16 o[DB];
#pragma unroll
  for (int i = 0; i < DB; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
  float m = -1e30f, l = 0.f;
  const int hh = lane >> 5;
  const int kl = lane & 31;
  const int kg = (kl >> 1) & 7;
  const int ntiles = (T + 63) / 64;
  const char* vbase[DB];
  {
    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    const int key0 = 4 * (g >> 1) + q;
    const int sw = ((q >> 1) & 1) << 2;
#pragma unroll
    for (int db = 0; db < DB; ++db) {
      const int chunk = db * 4 + 2 * (g & 1) + (pp >> 1);
      vbase[db] = (const char*)&KV[TILE16] + key0 * RB + ((chunk ^ sw) * 16) + 8 * (pp & 1);
    }
  }
  f32x16 s[2];
  bf16x8 pf[2][2];

  // the barrier of tile j: this wave's share of tile j has landed; behind the barrier everyone's has, and every wave is done with
  // tile j - 2, whose stage the request for tile j + 1 overwrites
#define SVT_AT_BAR(J)                                                                                   \
  {                                                                                                     \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                    \
    __builtin_amdgcn_s_barrier();                                                                       \
    if ((J) + 1 < ntiles) stage_dma((J) + 1, st_next);                                                  \
  }
#define SVT_AT_S(ST)                                                                                    \
  {                                                                                                     \
    const uint4* Kf = KV + (ST) * STAGE16;                                                              \
    _Pragma("unroll") for (int kb = 0; kb < 2; ++kb) {                                                  \
      _Pragma("unroll") for (int r = 0; r < 16; ++r) s[kb][r] = 0.f;                                    \
      _Pragma("unroll") for (int ks = 0; ks < KSD; ++ks)                                                \
        s[kb] = SVT_MFMA_32x32x16(__builtin_bit_cast(bf16x8, Kf[(kb * 32 + kl) * CPR + ((2 * ks + hh) ^ kg)]), qf[ks], s[kb]); \
    }                                                                                                   \
  }
  // softmax in the scaled log2 domain (flash_attn_kernel): row max on the raw scores, deferred rescale, P = exp2(s c - m) -> bf16
#define SVT_AT_SM(TILE)                                                                                 \
  {                                                                                                     \
    if (__builtin_expect((TILE) * 64 + 64 > T, 0)) {                                                    \
      const int kbase = (TILE) * 64 + 4 * hh;                                                           \
      _Pragma("unroll") for (int kb = 0; kb < 2; ++kb)                                                  \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                \
          const int key = kbase + kb * 32 + (r & 3) + 8 * (r >> 2);                                     \
          asm volatile("" : "+v"(s[kb][r]));                                                            \
          if (key >= T) s[kb][r] = -3e38f;                                                              \
        }                                                                                               \
    }                                                                                                   \
    float mx = fmaxf(s[0][0], s[1][0]);                                                                 \
    _Pragma("unroll") for (int r = 1; r < 16; ++r) mx = fmaxf(fmaxf(mx, s[0][r]), s[1][r]);             \
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64)) * c;                                                         \
    if (!__all(mx - m <= 8.0f)) {                                                                       \
      const float mnew = fmaxf(m, mx);                                                                  \
      const float alpha = __builtin_amdgcn_exp2f(m - mnew);                                             \
      l *= alpha;                                                                                       \
      m = mnew;                                                                                         \
      _Pragma("unroll") for (int i = 0; i < DB; ++i)                                                    \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) o[i][r] *= alpha;                                \
    }                                                                                                   \
    float sa = 0.f, sb = 0.f;                                                                           \
    _Pragma("unroll") for (int kb = 0; kb < 2; ++kb)                                                    \
      _Pragma("unroll") for (int r = 0; r < 16; r += 2) {                                               \
        const float p0 = __builtin_amdgcn_exp2f(fmaf(s[kb][r], c, -m));                                 \
        const float p1 = __builtin_amdgcn_exp2f(fmaf(s[kb][r + 1], c, -m));                             \
        s[kb][r] = p0; s[kb][r + 1] = p1;                                                               \
        sa += p0; sb += p1;                                                                             \
      }                                                                                                 \
    float sum = sa + sb;                                                                                \
    sum += __shfl_xor(sum, 32, 64);                                                                     \
    l += sum;                                                                                           \
    _Pragma("unroll") for (int kb = 0; kb < 2; ++kb)                                                    \
      _Pragma("unroll") for (int ss = 0; ss < 2; ++ss)                                                  \
        _Pragma("unroll") for (int j = 0; j < 8; ++j) pf[kb][ss][j] = (bf16_t)s[kb][ss * 8 + j];        \
  }
#define SVT_AT_PV(ST)                                                                                   \
  {                                                                                                     \
    _Pragma("unroll") for (int db = 0; db < DB; ++db)                                                   \
      _Pragma("unroll") for (int kb = 0; kb < 2; ++kb)                                                  \
        _Pragma("unroll") for (int ss = 0; ss < 2; ++ss) {                                              \
          const char* vp = vbase[db] + (ST) * STAGE_BYTES + (kb * 32 + ss * 16) * RB;                   \
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t)(vp));                  \
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t)(vp + 8 * RB));         \
          const s16x8 vv = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);                     \
          o[db] = SVT_MFMA_32x32x16(__builtin_bit_cast(bf16x8, vv), pf[kb][ss], o[db]);                 \
        }                                                                                               \
  }

  stage_dma(0, 0);
  int st = 0;   // stage of the tile being multiplied
  if constexpr (PIPE) {
    int st_prev = 0;
    if (wave < 4) {
      for (int tile = 0; tile < ntiles; ++tile) {
        const int st_next = st + 1 == NST ? 0 : st + 1;
        SVT_AT_BAR(tile)
        SVT_AT_S(st)
        __builtin_amdgcn_sched_barrier(0);
        if (tile > 0) SVT_AT_PV(st_prev)
        __builtin_amdgcn_sched_barrier(0);
        SVT_AT_SM(tile)
        __builtin_amdgcn_sched_barrier(0);
        st_prev = st;
        st = st_next;
      }
    } else {
      {
        const int st_next = 1;
        SVT_AT_BAR(0)
      }
      for (int tile = 0; tile < ntiles; ++tile) {
        const int st1 = st + 1 == NST ? 0 : st + 1;
        SVT_AT_S(st)
        __builtin_amdgcn_sched_barrier(0);
        if (tile > 0) SVT_AT_PV(st_prev)
        __builtin_amdgcn_sched_barrier(0);
        if (tile + 1 < ntiles) {
          const int st_next = st1 + 1 == NST ? 0 : st1 + 1;   // stage of tile + 2 (= the stage tile - 1 has just left)
          SVT_AT_BAR(tile + 1)
        }
        SVT_AT_SM(tile)
        __builtin_amdgcn_sched_barrier(0);
        st_prev = st;
        st = st1;
      }
    }
    SVT_AT_PV(st_prev)
  } else
  if (wave < 4) {
    for (int tile = 0; tile < ntiles; ++tile) {
      const int st_next = st + 1 == NST ? 0 : st + 1;
      SVT_AT_BAR(tile)
      SVT_AT_S(st)
      SVT_AT_SM(tile)
      SVT_AT_PV(st)
      st = st_next;
    }
  } else {
    {
      const int st_next = 1;
      SVT_AT_BAR(0)
    }
    for (int tile = 0; tile < ntiles; ++tile) {
      const int st1 = st + 1 == NST ? 0 : st + 1;
      SVT_AT_S(st)
      SVT_AT_SM(tile)
      if (tile + 1 < ntiles) {
        const int st_next = st1 + 1 == NST ? 0 : st1 + 1;   // stage of tile + 2
        SVT_AT_BAR(tile + 1)
      }
      SVT_AT_PV(st)
      st = st1;
    }
  }
#undef SVT_AT_BAR
#undef SVT_AT_S
#undef SVT_AT_SM
#undef SVT_AT_PV

  const int q = q0 + (lane & 31);
  if (q >= T) return;
  const float inv = 1.f / l;
  bf16_t* op = O + (long)b * o_bstride + (long)q * ldo + (long)h * DH;
#pragma unroll
  for (int db = 0; db < DB; ++db)
#pragma unroll
    for (int g = 0; g < 4; g += 2) {
      unsigned w[2][2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        bf16x4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = (bf16_t)(o[db][(g + u) * 4 + j] * inv);
        const uint2 pk = __builtin_bit_cast(uint2, v);
        w[u][0] = pk.x; w[u][1] = pk.y;
      }
#pragma unroll
      for (int d2 = 0; d2 < 2; ++d2) {
        const auto r = __builtin_amdgcn_permlane32_swap(w[0][d2], w[1][d2], false, false);
        w[0][d2] = r[0]; w[1][d2] = r[1];
      }
      *(uint4*)(op + db * 32 + 8 * (g + hh)) = uint4{w[0][0], w[0][1], w[1][0], w[1][1]};
    }
}

#ifdef SVT_DIAG
// ---------------------------------------------------------------------------------------------------------------------
// Software-pipelined form (round 5) -- AN EXPERIMENT, MEASURED SLOWER, built by `make DIAG=1` only (svt_debug_set key 21 = 43-45 / 83-86 / 85;
// stamps: key 21 = 99, tools/attn_pipe_stamps.py; numbers: profiles/r05_attention_pipeline_experiment.txt).  The idea: the matrix pipe
// works UNDER a wave's own softmax.
// flash_attn_stag_kernel overlaps the pipes only across waves: inside a wave the phases S = K Q^T (8 MFMAs) -> softmax (~150 vector
// instructions, 32 of them quarter-rate exponentials) -> O += V P (8 MFMAs) are a dependency chain, so a wave alone leaves the matrix
// pipe idle during its softmax and the vector pipe idle during its products; with 768 workgroups of 8 waves on 512 resident slots the
// launch is also 1.5 rounds deep (the second round runs half empty).  Here:
//   * ONE wave carries two score accumulators: while it exponentiates tile j (S_j, finished an iteration ago) the matrix pipe runs the
//     products that do not depend on that softmax -- O += V_{j-1} P_{j-1} (P of the previous tile, 16 registers) and S_{j+1} = K_{j+1} Q^T --
//     INTERLEAVED with it in program order, one MFMA per slot of ~7 vector instructions (an MFMA keeps the matrix pipe busy for 32
//     cycles and blocks the wave's vector issue for 8 of them: the rest of the slot is the softmax's);
//   * the running maximum of tile j is settled BEFORE the interleaved region (it needs all of S_j, and its rare "raise the maximum"
//     branch must not split the region): when it is raised the pending product O += V_{j-1} P_{j-1} is flushed first -- P_{j-1} was
//     taken against the old maximum, like the O it adds to -- and the region then runs without it.  Every value is computed by the
//     same operations in the same order as in flash_attn_kernel / flash_attn_stag_kernel: the outputs are bit-identical;
//   * four waves (128 queries) per workgroup, ~160 registers: three workgroups per CU = 768 resident slots for 4 x 384 = 1 536
//     workgroups at C2, exactly two rounds; a SIMD's three waves belong to three workgroups and drift out of phase by themselves;
//   * K in a ring of three 8 KiB tiles, V in a ring of three: at the top of iteration j (behind the barrier) K_{j+3} and V_{j+1} are
//     requested by LDS-DMA, each two iterations before its first read; the wait in front of the barrier is a counted vmcnt.
template <int N> __device__ __forceinline__ void attn_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// wait until at most n of this wave's LDS-DMA requests are outstanding (n is wave-uniform; the instruction takes an immediate)
__device__ __forceinline__ void attn_wait_vm_dyn(int n) {
  switch (n) {
    case 0: attn_wait_vm<0>(); break;
    case 1: attn_wait_vm<1>(); break;
    case 2: attn_wait_vm<2>(); break;
    case 3: attn_wait_vm<3>(); break;
    case 4: attn_wait_vm<4>(); break;
    case 5: attn_wait_vm<5>(); break;
    case 6: attn_wait_vm<6>(); break;
    case 7: attn_wait_vm<7>(); break;
    case 8: attn_wait_vm<8>(); break;
    case 9: attn_wait_vm<9>(); break;
    case 10: attn_wait_vm<10>(); break;
    case 11: attn_wait_vm<11>(); break;
    case 12: attn_wait_vm<12>(); break;
    case 13: attn_wait_vm<13>(); break;
    case 14: attn_wait_vm<14>(); break;
    case 15: attn_wait_vm<15>(); break;
    default: attn_wait_vm<16>(); break;
  }
}

// NW waves (32 queries each) per workgroup, K and V in rings of NR tiles, WPE waves per SIMD
template <int DH, int NW, int NR, int WPE>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void flash_attn_pipe_kernel(
    const bf16_t* __restrict__ Q, long ldq, long q_bstride, const bf16_t* __restrict__ K, long ldk, long k_bstride,
    const bf16_t* __restrict__ V, bf16_t* __restrict__ O, long ldo, long o_bstride, int T, int H, float c, int nqb, unsigned* stamps) {
  const bool STAMP = stamps != nullptr;   // diagnostics (tools/attn_pipe_stamps.py): s_memtime stamps per wave instead of the result
  static_assert(DH == 64 && (NW == 4 || NW == 8) && NR >= 3 && NR <= 8, "built for head_dim 64");
  constexpr int DPW = 8 / NW;   // LDS-DMA instructions per wave, tile and operand (8 rows x 128 B each)
  constexpr int KSD = DH / 16, DB = DH / 32, CPR = DH / 8, RB = 2 * DH;
  constexpr int TILE16 = 64 * CPR;                   // uint4 per tile (8 KiB)
  constexpr unsigned TILE_BYTES = TILE16 * 16;
  __shared__ __attribute__((aligned(16))) uint4 KV[2 * NR * TILE16];   // K ring, then V ring

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  unsigned stamp[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) stamp[i] = 0;
  if (STAMP) stamp[0] = (unsigned)__builtin_amdgcn_s_memtime();
#define SVT_PST(I) if (STAMP) stamp[I] = (unsigned)__builtin_amdgcn_s_memtime();
  const int L = blockIdx.x, BH = (int)gridDim.x / nqb;
  int bh, qblk;
  {   // the query blocks of one (clip, head) on one XCD (see flash_attn_stag_kernel)
    const int full = (BH / 8) * 8 * nqb;
    if (L < full) { bh = (L & 7) + 8 * (L / (8 * nqb)); qblk = (L >> 3) % nqb; }
    else { const int r = L - full; bh = (BH / 8) * 8 + r / nqb; qblk = r % nqb; }
  }
  const int b = bh / H, h = bh - b * H;
  const int q0 = qblk * (32 * NW) + wave * 32;
  const bf16_t* Qb = Q + (long)b * q_bstride + (long)h * DH;
  const bf16_t* Kb = K + (long)b * k_bstride + (long)h * DH;
  const bf16_t* Vb = V + (long)b * k_bstride + (long)h * DH;

  bf16x8 qf[KSD];
  int dkey[DPW], kch[DPW], vch[DPW];
#pragma unroll
  for (int i = 0; i < DPW; ++i) {
    const int kk = (wave * DPW + i) * 8 + lane / CPR;
    const int slot = lane % CPR;
    dkey[i] = kk;
    kch[i] = (slot ^ ((kk >> 1) & 7)) * 8;
    vch[i] = (slot ^ (((kk >> 1) & 1) << 2)) * 8;
  }
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(void __attribute__((address_space(3)))*)KV);
  const int ntiles = (T + 63) / 64;
  auto dma_k = [&](int tile, int slot) {
#pragma unroll
    for (int i = 0; i < DPW; ++i) {
      int key = tile * 64 + dkey[i];
      if (key > T - 1) key = T - 1;
      attn_dma16(Kb + (long)key * ldk + kch[i], lds0 + (unsigned)slot * TILE_BYTES + (unsigned)((wave * DPW + i) * 1024));
    }
  };
  auto dma_v = [&](int tile, int slot) {
#pragma unroll
    for (int i = 0; i < DPW; ++i) {
      int key = tile * 64 + dkey[i];
      if (key > T - 1) key = T - 1;
      attn_dma16(Vb + (long)key * ldk + vch[i], lds0 + (unsigned)(NR + slot) * TILE_BYTES + (unsigned)((wave * DPW + i) * 1024));
    }
  };
  // request schedule: "iteration" i (1 - NR <= i < ntiles; the negative ones are the head) asks for K_{i+NR} and V_{i+NR-2}, tile t into
  // slot t mod NR -- K_{i+NR} takes the place of K_i (its scores were finished in iteration i - 1) and V_{i+NR-2} that of V_{i-2} (multiplied
  // in iteration i - 1); both are first read in iteration i + NR - 1.  count(i) = requests of this wave in iteration i.
  auto issued = [&](int i) -> int { return DPW * ((i + NR >= 1 && i + NR < ntiles ? 1 : 0) + (i + NR - 2 >= 0 && i + NR - 2 < ntiles ? 1 : 0)); };
  auto issue = [&](int i, int kslot, int vslot) {
    if (i + NR >= 1 && i + NR < ntiles) dma_k(i + NR, kslot);
    if (i + NR - 2 >= 0 && i + NR - 2 < ntiles) dma_v(i + NR - 2, vslot);
  };

  f32x16 o[DB];
#pragma unroll
  for (int i = 0; i < DB; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
  float m = -1e30f, l = 0.f;
  const int hh = lane >> 5;
  const int kl = lane & 31;
  const int kg = (kl >> 1) & 7;
  unsigned vofs[DB];   // byte offset of this lane's transposed-read origin inside a V tile
  {
    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    const int key0 = 4 * (g >> 1) + q;
    const int sw = ((q >> 1) & 1) << 2;
#pragma unroll
    for (int db = 0; db < DB; ++db) {
      const int chunk = db * 4 + 2 * (g & 1) + (pp >> 1);
      vofs[db] = (unsigned)(key0 * RB + ((chunk ^ sw) * 16) + 8 * (pp & 1));
    }
  }
  const char* const lds_v = (const char*)&KV[NR * TILE16];
  f32x16 s[2], sn[2];
  bf16x8 pf[2][2];

  auto k_frag = [&](int kslot, int kb, int ks) -> bf16x8 {
    return __builtin_bit_cast(bf16x8, KV[kslot * TILE16 + (kb * 32 + kl) * CPR + ((2 * ks + hh) ^ kg)]);
  };
  auto v_frag = [&](int vslot, int db, int kb, int ss) -> bf16x8 {
    const char* vp = lds_v + (unsigned)vslot * TILE_BYTES + vofs[db] + (kb * 32 + ss * 16) * RB;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t)(vp));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t)(vp + 8 * RB));
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
  };
  // O += V_prev P_prev as one block (the flush in front of a raised maximum, and the last tile's product)
  auto pv_block = [&](int vslot) {
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int ss = 0; ss < 2; ++ss)
#pragma unroll
        for (int db = 0; db < DB; ++db) o[db] = SVT_MFMA_32x32x16(v_frag(vslot, db, kb, ss), pf[kb][ss], o[db]);
  };

  // ---- head: K_0, then the requests of iterations 1 - NR .. -1, then the Q fragments (behind the requests: their latency runs beside
  //      the tiles'); S_0 = K_0 Q^T once K_0 has landed ----
  dma_k(0, 0);
  {
    int after_k0 = 0;
#pragma unroll
    for (int i = 1 - NR; i < 0; ++i) {
      issue(i, (i + NR) % NR, (i + NR - 2 + NR) % NR);
      after_k0 += issued(i);
    }
    {
      int q = q0 + (lane & 31);
      if (q > T - 1) q = T - 1;
      const bf16_t* qp = Qb + (long)q * ldq + 8 * (lane >> 5);
      // asm loads: the compiler must not count them (it cannot see the requests around them and would wait vmcnt(0) at their first use)
#pragma unroll
      for (int ks = 0; ks < KSD; ++ks) asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(qf[ks]) : "v"(qp), "n"(ks * 32) : "memory");
    }
    SVT_PST(1)
    attn_wait_vm_dyn(after_k0 + KSD);   // K_0 has landed; the head's later requests and the Q fragments may still be in flight
  }
  __builtin_amdgcn_s_barrier();
  attn_wait_vm<0>();   // Q (and, the only time, everything requested so far)
#pragma unroll
  for (int ks = 0; ks < KSD; ++ks) asm volatile("" : "+v"(qf[ks]));
  SVT_PST(2)
#pragma unroll
  for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
    for (int r = 0; r < 16; ++r) s[kb][r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KSD; ++ks) s[kb] = SVT_MFMA_32x32x16(k_frag(0, kb, ks), qf[ks], s[kb]);
  }
  SVT_PST(3)

  // row maximum of a finished score tile in the scaled log2 domain (flash_attn_kernel's order of operations)
  auto tile_max = [&](const f32x16 (&t)[2]) -> float {
    float mx = fmaxf(t[0][0], t[1][0]);
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(fmaxf(mx, t[0][r]), t[1][r]);
    return fmaxf(mx, __shfl_xor(mx, 32, 64)) * c;
  };

  // The interleaved region of iteration j: the softmax of S_j (16 units of two scores: one packed multiply-add, two exponentials, one
  // packed addition, a conversion every fourth unit) against up to 16 MFMAs -- first the next tile's scores S_{j+1} = K_{j+1} Q^T, then
  // the previous tile's product O += V_{j-1} P_{j-1}; between the MFMAs of the second half also the row maximum of S_{j+1} (MAXN:
  // the next tile is a whole one; a partial last tile is masked and reduced at the top of its own iteration), and in the first slots
  // the ring's requests of this iteration.  Returns the next tile's maximum.
  auto region = [&](auto next_c, auto pv_c, auto max_c, int j, int kslot_next, int vslot_prev, int kslot_free, int vslot_free) -> float {
    constexpr bool NEXT = decltype(next_c)::value, PVP = decltype(pv_c)::value, MAXN = decltype(max_c)::value;
    constexpr int NM = (NEXT ? 8 : 0) + (PVP ? 8 : 0);
    f32x2_t sab = {0.f, 0.f};
    const f32x2_t cc = {c, c}, mm = {-m, -m};
    bf16x8 pn[2][2];
    float mxn = 0.f, mxa = 0.f;
    auto unit = [&](int u) {   // scores (kb, r), (kb, r + 1): the arithmetic and the summation order of flash_attn_kernel
      const int kb = u >> 3, r = 2 * (u & 7);
      // the empty asm statements pin the unit to its slot: instruction selection orders a block's pure arithmetic by data dependence
      // alone and would put all the multiply-adds in front of the first MFMA and all the additions behind the last one
      asm volatile("" : "+v"(s[kb][r]), "+v"(s[kb][r + 1]));
      const f32x2_t e = __builtin_elementwise_fma(f32x2_t{s[kb][r], s[kb][r + 1]}, cc, mm);
      const float p0 = __builtin_amdgcn_exp2f(e.x);
      const float p1 = __builtin_amdgcn_exp2f(e.y);
      s[kb][r] = p0; s[kb][r + 1] = p1;
      sab += f32x2_t{p0, p1};
      if ((u & 3) == 3) {   // eight scores complete: their bf16 operand
        const int ss = (u >> 2) & 1;
#pragma unroll
        for (int j2 = 0; j2 < 8; ++j2) pn[kb][ss][j2] = (bf16_t)s[kb][ss * 8 + j2];
        asm volatile("" : "+v"(pn[kb][ss]));
      }
      asm volatile("" : "+v"(sab));
    };
    if constexpr (NM == 0) {
      issue(j, kslot_free, vslot_free);
#pragma unroll
      for (int u = 0; u < 16; ++u) unit(u);
    } else {
      constexpr int UPS = 16 / NM;   // units per MFMA slot
      auto frag = [&](int i) -> bf16x8 {
        if (NEXT && i < 8) return k_frag(kslot_next, i >> 2, i & 3);
        const int t = i - (NEXT ? 8 : 0);   // kb / ss outer: per accumulator the order of flash_attn_kernel
        return v_frag(vslot_prev, t & 1, t >> 2, (t >> 1) & 1);
      };
      // fragments are read AHEAD slots in front of their MFMA (an LDS read returns after ~100-150 cycles, a slot lasts ~45)
      constexpr int AHEAD = 3;
      bf16x8 fr[AHEAD + 1];
#pragma unroll
      for (int i = 0; i < AHEAD && i < NM; ++i) fr[i] = frag(i);
#pragma unroll
      for (int i = 0; i < NM; ++i) {
        if (i + AHEAD < NM) fr[(i + AHEAD) % (AHEAD + 1)] = frag(i + AHEAD);
        if (NEXT && i < 8) {
          const int kb = i >> 2, ks = i & 3;
          if (ks == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) sn[kb][r] = 0.f;
          }
          sn[kb] = SVT_MFMA_32x32x16(fr[i % (AHEAD + 1)], qf[ks], sn[kb]);
        } else {
          const int t = i - (NEXT ? 8 : 0);
          o[t & 1] = SVT_MFMA_32x32x16(fr[i % (AHEAD + 1)], pf[t >> 2][(t >> 1) & 1], o[t & 1]);
        }
        if (i == 1) issue(j, kslot_free, vslot_free);   // this iteration's requests, in the shadow of the first MFMAs
#pragma unroll
        for (int u = 0; u < UPS; ++u) unit(i * UPS + u);
        if constexpr (NEXT && MAXN) {
          // the next tile's row maximum, four scores per slot from slot 8 on (sn[0] is complete when MFMA 4 has retired, sn[1] with MFMA 8;
          // without a product to hide behind -- the first iteration -- it follows the last unit)
          constexpr int M0 = PVP ? 8 : NM;
          if (i >= M0 || i == NM - 1) {
            const int lo = i >= M0 ? (i - M0) * 2 : 0, hi = i == NM - 1 ? 16 : (i - M0) * 2 + 2;
#pragma unroll
            for (int r = lo; r < hi; ++r) {
              if (r == 0) { mxn = fmaxf(sn[0][0], sn[1][0]); }
              else mxn = fmaxf(fmaxf(mxn, sn[0][r]), sn[1][r]);
            }
            asm volatile("" : "+v"(mxn));
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if constexpr (NEXT && MAXN) mxa = fmaxf(mxn, __shfl_xor(mxn, 32, 64)) * c;
    float sum = sab.x + sab.y;
    sum += __shfl_xor(sum, 32, 64);
    l += sum;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int ss = 0; ss < 2; ++ss) pf[kb][ss] = pn[kb][ss];
    return mxa;
  };
  using yes = std::integral_constant<bool, true>;
  using no = std::integral_constant<bool, false>;

  int ks0 = 0, ks1 = 1 % NR, vs_prev = NR - 1, vs_free = NR - 2;   // slots of K_j, K_{j+1}, V_{j-1}, V_{j-2}
  int allowed = 0;   // requests of iterations j + 2 - NR .. j - 1: what may still be in flight at the top of iteration j
#pragma unroll
  for (int i = 2 - NR; i < 0; ++i) allowed += issued(i);
  const bool last_partial = (ntiles * 64 > T);
  float mx = 0.f;
  bool have_mx = false;
  for (int j = 0; j < ntiles; ++j) {
    // ---- top: K_{j+1} and V_{j-1} have landed (every wave waits for its own pieces, the barrier publishes them); behind the
    //      barrier K_j's and V_{j-2}'s tiles are free: K_{j+NR} and V_{j+NR-2} go there (from inside the region) ----
    if (STAMP) { if (j == 3) stamp[4] = (unsigned)__builtin_amdgcn_s_memtime(); }
    attn_wait_vm_dyn(allowed);
    if (STAMP) { if (j == 3) stamp[5] = (unsigned)__builtin_amdgcn_s_memtime(); }
    __builtin_amdgcn_s_barrier();
    if (STAMP) { if (j == 3) stamp[6] = (unsigned)__builtin_amdgcn_s_memtime(); }
    allowed += issued(j) - issued(j + 2 - NR);
    // ---- the running maximum of tile j: taken inside the previous region unless this is the first or a partial last tile ----
    if (!have_mx) {
      if (j * 64 + 64 > T) {
        const int kbase = j * 64 + 4 * hh;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int key = kbase + kb * 32 + (r & 3) + 8 * (r >> 2);
            asm volatile("" : "+v"(s[kb][r]));
            if (key >= T) s[kb][r] = -3e38f;
          }
      }
      mx = tile_max(s);
    }
    if (STAMP) { if (j == 3) stamp[7] = (unsigned)__builtin_amdgcn_s_memtime(); }
    bool pending = j > 0;
    if (!__all(mx - m <= 8.0f)) {
      if (pending) { pv_block(vs_prev); pending = false; }
      const float mnew = fmaxf(m, mx);
      const float alpha = __builtin_amdgcn_exp2f(m - mnew);
      l *= alpha;
      m = mnew;
#pragma unroll
      for (int i = 0; i < DB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
    }
    __builtin_amdgcn_sched_barrier(0);
    if (STAMP) { if (j == 3) stamp[8] = (unsigned)__builtin_amdgcn_s_memtime(); }
    if (j + 1 < ntiles) {
      const bool maxn = !(last_partial && j + 2 == ntiles);   // the next tile is a whole one: its maximum comes out of the region
      if (maxn) {
        mx = pending ? region(yes{}, yes{}, yes{}, j, ks1, vs_prev, ks0, vs_free) : region(yes{}, no{}, yes{}, j, ks1, vs_prev, ks0, vs_free);
      } else {
        if (pending) region(yes{}, yes{}, no{}, j, ks1, vs_prev, ks0, vs_free); else region(yes{}, no{}, no{}, j, ks1, vs_prev, ks0, vs_free);
      }
      have_mx = maxn;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) s[kb] = sn[kb];
    } else {
      if (pending) region(no{}, yes{}, no{}, j, ks1, vs_prev, ks0, vs_free); else region(no{}, no{}, no{}, j, ks1, vs_prev, ks0, vs_free);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (STAMP) { if (j == 3) stamp[9] = (unsigned)__builtin_amdgcn_s_memtime(); }
    ks0 = ks1; ks1 = ks1 + 1 == NR ? 0 : ks1 + 1;
    vs_free = vs_prev; vs_prev = vs_prev + 1 == NR ? 0 : vs_prev + 1;
  }
  SVT_PST(10)
  // ---- the last tile's product: V_{n-1} was requested an iteration ago and is covered by no barrier yet ----
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  pv_block(vs_prev);
  SVT_PST(11)
  if (STAMP) {
    asm volatile("" ::"v"(o[0][0]), "v"(o[1][0]), "v"(l));   // (a 64-byte "v" operand is refused by the HOST pass, which then drops the kernel stub without a word)
    stamp[12] = (unsigned)__builtin_amdgcn_s_memtime();
    if (lane == 0) {
      unsigned* rec = stamps + ((long)blockIdx.x * NW + wave) * 16;
#pragma unroll
      for (int i = 0; i < 16; ++i) rec[i] = i == 15 ? 0x5A5A0000u | (unsigned)ntiles : stamp[i];
    }
    return;
  }
#undef SVT_PST

  const int q = q0 + (lane & 31);
  if (q >= T) return;
  const float inv = 1.f / l;
  bf16_t* op = O + (long)b * o_bstride + (long)q * ldo + (long)h * DH;
#pragma unroll
  for (int db = 0; db < DB; ++db)
#pragma unroll
    for (int g = 0; g < 4; g += 2) {
      unsigned w[2][2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        bf16x4 v;
#pragma unroll
        for (int j2 = 0; j2 < 4; ++j2) v[j2] = (bf16_t)(o[db][(g + u) * 4 + j2] * inv);
        const uint2 pk = __builtin_bit_cast(uint2, v);
        w[u][0] = pk.x; w[u][1] = pk.y;
      }
#pragma unroll
      for (int d2 = 0; d2 < 2; ++d2) {
        const auto r = __builtin_amdgcn_permlane32_swap(w[0][d2], w[1][d2], false, false);
        w[0][d2] = r[0]; w[1][d2] = r[1];
      }
      *(uint4*)(op + db * 32 + 8 * (g + hh)) = uint4{w[0][0], w[0][1], w[1][0], w[1][1]};
    }
}

#endif  // SVT_DIAG

// ---------------------------------------------------------------------------------------------------------------------
// Split-operand fused attention (precision "bf16x3" / "fp16x3").  Q, K, V arrive as 16-bit (hi, lo) PLANES of the fp32
// projections (split_planes_kernel below; plane = same (rows, ld) layout, lo plane `plane` elements after the hi plane);
// every product is three MFMAs accumulated in fp32:
//     S^T = Kh Qh^T + Kl Qh^T + Kh Ql^T,     O^T += Vh^T Ph^T + Vl^T Ph^T + Vh^T Pl^T   with P = (Ph, Pl) cut in registers
// Everything else is flash_attn_kernel: keys on the MFMA rows, softmax state in fp32 registers, K / V tiles by LDS-DMA
// into two stages (four tiles per stage: K hi, K lo, V hi, V lo), V read transposed out of its row-major tiles, deferred
// rescale.  Output fp32.  Replaces, in the split modes, QK^T GEMM + softmax_rows + transpose_v + PV GEMM over a
// materialised (B, H, T, T) score tensor: 8.2 of the 23.2 ms of a 32 x 10 s wav2vec2-base step went there.
template <bool F16> struct X3T;
template <> struct X3T<false> {
  typedef real_bf16x8 v8;
  static __device__ __forceinline__ f32x16 mma(const uint4& a, const v8& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(real_bf16x8, a), b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ f32x16 mma(const s16x8& a, const v8& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(real_bf16x8, a), b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ void cut(float x, unsigned short& hi, unsigned short& lo) {
    const __bf16 a = (__bf16)x, b = (__bf16)(x - (float)a);
    hi = __builtin_bit_cast(unsigned short, a);
    lo = __builtin_bit_cast(unsigned short, b);
  }
};
template <> struct X3T<true> {
  typedef _Float16 v8 __attribute__((ext_vector_type(8)));
  static __device__ __forceinline__ f32x16 mma(const uint4& a, const v8& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(v8, a), b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ f32x16 mma(const s16x8& a, const v8& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(v8, a), b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ void cut(float x, unsigned short& hi, unsigned short& lo) {
    const _Float16 a = (_Float16)x, b = (_Float16)(x - (float)a);
    hi = __builtin_bit_cast(unsigned short, a);
    lo = __builtin_bit_cast(unsigned short, b);
  }
};

template <int DH, bool F16, int NW = 4>
__global__ __launch_bounds__(64 * NW) void flash_attn_x3_kernel(const unsigned short* __restrict__ Q, long ldq, long q_bstride, long q_plane,
                                                            const unsigned short* __restrict__ K, const unsigned short* __restrict__ V,
                                                            long ldk, long k_bstride, long k_plane, float* __restrict__ O, long ldo,
                                                            long o_bstride, int T, int H, float c, int o_pairs) {
  typedef X3T<F16> X;
  typedef typename X::v8 v8;
  constexpr int KSD = DH / 16, DB = DH / 32, CPR = DH / 8, RB = 2 * DH;
  constexpr int TILE16 = 64 * CPR;  // uint4 per tile
  // two stages of { K hi, K lo, V hi, V lo }
  __shared__ __attribute__((aligned(16))) uint4 KV[2][4 * TILE16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.z, h = blockIdx.y;
  const int q0 = blockIdx.x * (32 * NW) + wave * 32;
  const unsigned short* Qb = Q + (long)b * q_bstride + (long)h * DH;
  const unsigned short* Kb = K + (long)b * k_bstride + (long)h * DH;
  const unsigned short* Vb = V + (long)b * k_bstride + (long)h * DH;
  v8 qh[KSD], ql[KSD];
  {
    int q = q0 + (lane & 31);
    if (q > T - 1) q = T - 1;
    const unsigned short* qp = Qb + (long)q * ldq + 8 * (lane >> 5);
#pragma unroll
    for (int ks = 0; ks < KSD; ++ks) {
      qh[ks] = *(const v8*)(qp + ks * 16);
      ql[ks] = *(const v8*)(qp + q_plane + ks * 16);
    }
  }
  constexpr int ROWS_PER_DMA = 64 / CPR;
  constexpr int DMA_PER_WAVE = 64 / ROWS_PER_DMA / NW;
  static_assert(DMA_PER_WAVE >= 1, "too many waves for the tile's fill");
  int dkey[DMA_PER_WAVE], kch[DMA_PER_WAVE], vch[DMA_PER_WAVE];
#pragma unroll
  for (int i = 0; i < DMA_PER_WAVE; ++i) {
    const int kk = (wave * DMA_PER_WAVE + i) * ROWS_PER_DMA + lane / CPR;
    const int slot = lane % CPR;
    const int g = (DH == 64) ? ((kk >> 1) & 7) : (kk & 15);
    const int sw = (DH == 64) ? (((kk >> 1) & 1) << 2) : ((kk & 3) << 2);
    dkey[i] = kk;
    kch[i] = (slot ^ g) * 8;
    vch[i] = (slot ^ sw) * 8;
  }
  typedef const void __attribute__((address_space(1)))* gptr_t;
  typedef void __attribute__((address_space(3)))* lptr_t;
#define SVT_STAGE_X3(TILE, STAGE)                                                                                  \
  {                                                                                                                  \
    const int key0_ = (TILE) * 64;                                                                                   \
    _Pragma("unroll") for (int i = 0; i < DMA_PER_WAVE; ++i) {                                                       \
      int key_ = key0_ + dkey[i];                                                                                    \
      if (key_ > T - 1) key_ = T - 1;                                                                                \
      const unsigned short* kp_ = Kb + (long)key_ * ldk;                                                             \
      const unsigned short* vp_ = Vb + (long)key_ * ldk;                                                             \
      uint4* dst_ = &KV[STAGE][(wave * DMA_PER_WAVE + i) * 64];                                                      \
      __builtin_amdgcn_global_load_lds((gptr_t)(kp_ + kch[i]), (lptr_t)(dst_), 16, 0, 0);                            \
      __builtin_amdgcn_global_load_lds((gptr_t)(kp_ + k_plane + kch[i]), (lptr_t)(dst_ + TILE16), 16, 0, 0);        \
      __builtin_amdgcn_global_load_lds((gptr_t)(vp_ + vch[i]), (lptr_t)(dst_ + 2 * TILE16), 16, 0, 0);               \
      __builtin_amdgcn_global_load_lds((gptr_t)(vp_ + k_plane + vch[i]), (lptr_t)(dst_ + 3 * TILE16), 16, 0, 0);     \
    }                                                                                                                \
  }
  f32x16 o[DB];
#pragma unroll
  for (int i = 0; i < DB; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
  float m = -1e30f, l = 0.f;
  const int hh = lane >> 5, kl = lane & 31;
  const int kg = (DH == 64) ? ((kl >> 1) & 7) : (kl & 15);
  const int ntiles = (T + 63) / 64;
  const char* vbase[DB];
  {
    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    const int key0 = 4 * (g >> 1) + q;
    const int sw = (DH == 64) ? (((q >> 1) & 1) << 2) : (q << 2);
#pragma unroll
    for (int db = 0; db < DB; ++db) {
      const int chunk = db * 4 + 2 * (g & 1) + (pp >> 1);
      vbase[db] = (const char*)&KV[0][2 * TILE16] + key0 * RB + ((chunk ^ sw) * 16) + 8 * (pp & 1);
    }
  }
  constexpr long STAGE_BYTES = (long)4 * TILE16 * 16;
  constexpr long PLANE_BYTES = (long)TILE16 * 16;
  SVT_STAGE_X3(0, 0)
  for (int tile = 0; tile < ntiles; ++tile) {
    const int st = tile & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tile + 1 < ntiles) {
      if (st) SVT_STAGE_X3(tile + 1, 0) else SVT_STAGE_X3(tile + 1, 1)
    }
    const uint4* Kh = KV[st];
    const uint4* Kl = KV[st] + TILE16;
    f32x16 s[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[kb][r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < KSD; ++ks) {
        const int idx = (kb * 32 + kl) * CPR + ((2 * ks + hh) ^ kg);
        const uint4 kh = Kh[idx], klo = Kl[idx];
        s[kb] = X::mma(klo, qh[ks], s[kb]);
        s[kb] = X::mma(kh, ql[ks], s[kb]);
        s[kb] = X::mma(kh, qh[ks], s[kb]);
      }
    }
    if (__builtin_expect(tile * 64 + 64 > T, 0)) {
      const int kbase = tile * 64 + 4 * hh;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kbase + kb * 32 + (r & 3) + 8 * (r >> 2);
          asm volatile("" : "+v"(s[kb][r]));
          if (key >= T) s[kb][r] = -3e38f;
        }
    }
    float mx = fmaxf(s[0][0], s[1][0]);
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(fmaxf(mx, s[0][r]), s[1][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64)) * c;
    if (!__all(mx - m <= 8.0f)) {
      const float mnew = fmaxf(m, mx);
      const float alpha = __builtin_amdgcn_exp2f(m - mnew);
      l *= alpha;
      m = mnew;
#pragma unroll
      for (int i = 0; i < DB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
    }
    float sum = 0.f;
    // P = exp2(s c - m) cut into (hi, lo) pieces; the row sum is taken over the fp32 values
    unsigned short ph[2][16], pl[2][16];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pv = __builtin_amdgcn_exp2f(fmaf(s[kb][r], c, -m));
        sum += pv;
        X::cut(pv, ph[kb][r], pl[kb][r]);
      }
    sum += __shfl_xor(sum, 32, 64);
    l += sum;
    v8 pfh[2][2], pfl[2][2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int ss = 0; ss < 2; ++ss) {
        s16x8 th, tl;
#pragma unroll
        for (int j = 0; j < 8; ++j) { th[j] = (short)ph[kb][ss * 8 + j]; tl[j] = (short)pl[kb][ss * 8 + j]; }
        pfh[kb][ss] = __builtin_bit_cast(v8, th);
        pfl[kb][ss] = __builtin_bit_cast(v8, tl);
      }
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int ss = 0; ss < 2; ++ss) {
          const char* vp = vbase[db] + st * STAGE_BYTES + (kb * 32 + ss * 16) * RB;
          const s16x4 hlo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t)(vp));
          const s16x4 hhi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t)(vp + 8 * RB));
          const s16x4 llo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t)(vp + PLANE_BYTES));
          const s16x4 lhi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t)(vp + PLANE_BYTES + 8 * RB));
          const s16x8 vh = __builtin_shufflevector(hlo, hhi, 0, 1, 2, 3, 4, 5, 6, 7);
          const s16x8 vl = __builtin_shufflevector(llo, lhi, 0, 1, 2, 3, 4, 5, 6, 7);
          o[db] = X::mma(vl, pfh[kb][ss], o[db]);
          o[db] = X::mma(vh, pfl[kb][ss], o[db]);
          o[db] = X::mma(vh, pfh[kb][ss], o[db]);
        }
  }
  const int q = q0 + (lane & 31);
  if (q >= T) return;
  const float inv = 1.f / l;
  float* op = O + (long)b * o_bstride + (long)q * ldo + (long)h * DH;
  if (o_pairs) {
    // the output projection's operand as pair rows (gemm_x3q.hip): [32 hi | 32 lo] per 32 elements, in the bytes of the fp32 slab
    // (ldo, o_bstride and h * DH are multiples of 32 elements: the launcher checks)
    char* pp = (char*)op;
    // a lane holds 4 consecutive d per (db, g) -- 8 bytes per plane -- and its partner lane ^ 32 the other 4 of the same 8: two half
    // exchanges (v_permlane32_swap) per plane and pair of groups give lanes 0-31 the 8 pieces of group g and lanes 32-63 those of group
    // g + 1, i.e. one 16-byte store per plane instead of two 8-byte ones (the 8-byte form cost 14 us of a 113 us launch: store issue)
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
      for (int g = 0; g < 4; g += 2) {
        unsigned hx[2][2], lx[2][2];   // [group g / g + 1][dword]
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          unsigned short ph_[4], pl_[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) X::cut(o[db][(g + u) * 4 + j] * inv, ph_[j], pl_[j]);
          hx[u][0] = (unsigned)ph_[0] | ((unsigned)ph_[1] << 16); hx[u][1] = (unsigned)ph_[2] | ((unsigned)ph_[3] << 16);
          lx[u][0] = (unsigned)pl_[0] | ((unsigned)pl_[1] << 16); lx[u][1] = (unsigned)pl_[2] | ((unsigned)pl_[3] << 16);
        }
#pragma unroll
        for (int w = 0; w < 2; ++w) {
          const auto rh = __builtin_amdgcn_permlane32_swap(hx[0][w], hx[1][w], false, false);
          hx[0][w] = rh[0]; hx[1][w] = rh[1];
          const auto rl = __builtin_amdgcn_permlane32_swap(lx[0][w], lx[1][w], false, false);
          lx[0][w] = rl[0]; lx[1][w] = rl[1];
        }
        char* d = pp + db * 128 + (8 * (g + hh)) * 2;
        *(uint4*)d = uint4{hx[0][0], hx[0][1], hx[1][0], hx[1][1]};
        *(uint4*)(d + 64) = uint4{lx[0][0], lx[0][1], lx[1][0], lx[1][1]};
      }
    return;
  }
#pragma unroll
  for (int db = 0; db < DB; ++db)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 v = {o[db][g * 4] * inv, o[db][g * 4 + 1] * inv, o[db][g * 4 + 2] * inv, o[db][g * 4 + 3] * inv};
      *(float4*)(op + db * 32 + 8 * g + 4 * hh) = v;
    }
}
#undef SVT_STAGE_X3

// Staggered form of flash_attn_x3_kernel<64, F16, 8> (round 4; see flash_attn_stag_kernel): three stages of { K hi, K lo, V hi, V lo }
// (96 KiB: one workgroup per CU, as before -- 172 registers), waves 4-7 half a tile behind waves 0-3, LDS-DMA from inline asm, 1-D
// grid with the query blocks of a (clip, head) on one XCD.  With ONE workgroup per CU the lockstep form had nobody to fill the matrix
// pipe during the softmax of its eight waves.
template <bool F16>
__global__ __launch_bounds__(512) void flash_attn_x3_stag_kernel(const unsigned short* __restrict__ Q, long ldq, long q_bstride, long q_plane,
                                                                 const unsigned short* __restrict__ K, const unsigned short* __restrict__ V,
                                                                 long ldk, long k_bstride, long k_plane, float* __restrict__ O, long ldo,
                                                                 long o_bstride, int T, int H, float c, int o_pairs, int nqb) {
  typedef X3T<F16> X;
  typedef typename X::v8 v8;
  constexpr int DH = 64, NW = 8, NST = 3;
  constexpr int KSD = DH / 16, DB = DH / 32, CPR = DH / 8, RB = 2 * DH;
  constexpr int TILE16 = 64 * CPR;           // uint4 per tile
  constexpr int STAGE16 = 4 * TILE16;        // K hi, K lo, V hi, V lo
  constexpr long STAGE_BYTES = (long)STAGE16 * 16;
  constexpr long PLANE_BYTES = (long)TILE16 * 16;
  extern __shared__ __attribute__((aligned(16))) uint4 KV[];   // NST * STAGE16 (96 KiB: dynamic)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int L = blockIdx.x, BH = (int)gridDim.x / nqb;
  int bh, qblk;
  {
    const int full = (BH / 8) * 8 * nqb;
    if (L < full) { bh = (L & 7) + 8 * (L / (8 * nqb)); qblk = (L >> 3) % nqb; }
    else { const int r = L - full; bh = (BH / 8) * 8 + r / nqb; qblk = r % nqb; }
  }
  const int b = bh / H, h = bh - b * H;
  const int q0 = qblk * (32 * NW) + wave * 32;
  const unsigned short* Qb = Q + (long)b * q_bstride + (long)h * DH;
  const unsigned short* Kb = K + (long)b * k_bstride + (long)h * DH;
  const unsigned short* Vb = V + (long)b * k_bstride + (long)h * DH;
  v8 qh[KSD], ql[KSD];
  {
    int q = q0 + (lane & 31);
    if (q > T - 1) q = T - 1;
    const unsigned short* qp = Qb + (long)q * ldq + 8 * (lane >> 5);
#pragma unroll
    for (int ks = 0; ks < KSD; ++ks) {
      qh[ks] = *(const v8*)(qp + ks * 16);
      ql[ks] = *(const v8*)(qp + q_plane + ks * 16);
    }
  }
  const int dkey = wave * 8 + lane / CPR;
  const int slot = lane % CPR;
  const int kch = (slot ^ ((dkey >> 1) & 7)) * 8;
  const int vch = (slot ^ (((dkey >> 1) & 1) << 2)) * 8;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(void __attribute__((address_space(3)))*)KV);
  auto stage_dma = [&](int tile, int stage) {
    int key = tile * 64 + dkey;
    if (key > T - 1) key = T - 1;
    const unsigned short* kp = Kb + (long)key * ldk;
    const unsigned short* vp = Vb + (long)key * ldk;
    const unsigned dst = lds0 + (unsigned)((stage * STAGE16 + wave * 64) * 16);
    attn_dma16(kp + kch, dst);
    attn_dma16(kp + k_plane + kch, dst + TILE16 * 16);
    attn_dma16(vp + vch, dst + 2 * TILE16 * 16);
    attn_dma16(vp + k_plane + vch, dst + 3 * TILE16 * 16);
  };
  f32x16 o[DB];
#pragma unroll
  for (int i = 0; i < DB; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
  float m = -1e30f, l = 0.f;
  const int hh = lane >> 5, kl = lane & 31;
  const int kg = (kl >> 1) & 7;
  const int ntiles = (T + 63) / 64;
  const char* vbase[DB];
  {
    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    const int key0 = 4 * (g >> 1) + q;
    const int sw = ((q >> 1) & 1) << 2;
#pragma unroll
    for (int db = 0; db < DB; ++db) {
      const int chunk = db * 4 + 2 * (g & 1) + (pp >> 1);
      vbase[db] = (const char*)&KV[2 * TILE16] + key0 * RB + ((chunk ^ sw) * 16) + 8 * (pp & 1);
    }
  }
  f32x16 s[2];
  v8 pfh[2][2], pfl[2][2];
#define SVT_X3_BAR(J)                                                                                   \
  {                                                                                                     \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                    \
    __builtin_amdgcn_s_barrier();                                                                       \
    if ((J) + 1 < ntiles) stage_dma((J) + 1, st_next);                                                  \
  }
#define SVT_X3_S(ST)                                                                                    \
  {                                                                                                     \
    const uint4* Kh = KV + (ST) * STAGE16;                                                              \
    const uint4* Kl = Kh + TILE16;                                                                      \
    _Pragma("unroll") for (int kb = 0; kb < 2; ++kb) {                                                  \
      _Pragma("unroll") for (int r = 0; r < 16; ++r) s[kb][r] = 0.f;                                    \
      _Pragma("unroll") for (int ks = 0; ks < KSD; ++ks) {                                              \
        const int idx = (kb * 32 + kl) * CPR + ((2 * ks + hh) ^ kg);                                    \
        const uint4 kh = Kh[idx], klo = Kl[idx];                                                        \
        s[kb] = X::mma(klo, qh[ks], s[kb]);                                                             \
        s[kb] = X::mma(kh, ql[ks], s[kb]);                                                              \
        s[kb] = X::mma(kh, qh[ks], s[kb]);                                                              \
      }                                                                                                 \
    }                                                                                                   \
  }
#define SVT_X3_SM(TILE)                                                                                 \
  {                                                                                                     \
    if (__builtin_expect((TILE) * 64 + 64 > T, 0)) {                                                    \
      const int kbase = (TILE) * 64 + 4 * hh;                                                           \
      _Pragma("unroll") for (int kb = 0; kb < 2; ++kb)                                                  \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                \
          const int key = kbase + kb * 32 + (r & 3) + 8 * (r >> 2);                                     \
          asm volatile("" : "+v"(s[kb][r]));                                                            \
          if (key >= T) s[kb][r] = -3e38f;                                                              \
        }                                                                                               \
    }                                                                                                   \
    float mx = fmaxf(s[0][0], s[1][0]);                                                                 \
    _Pragma("unroll") for (int r = 1; r < 16; ++r) mx = fmaxf(fmaxf(mx, s[0][r]), s[1][r]);             \
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64)) * c;                                                         \
    if (!__all(mx - m <= 8.0f)) {                                                                       \
      const float mnew = fmaxf(m, mx);                                                                  \
      const float alpha = __builtin_amdgcn_exp2f(m - mnew);                                             \
      l *= alpha;                                                                                       \
      m = mnew;                                                                                         \
      _Pragma("unroll") for (int i = 0; i < DB; ++i)                                                    \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) o[i][r] *= alpha;                                \
    }                                                                                                   \
    float sum = 0.f;                                                                                    \
    unsigned short ph[2][16], pl[2][16];                                                                \
    _Pragma("unroll") for (int kb = 0; kb < 2; ++kb)                                                    \
      _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                  \
        const float pv = __builtin_amdgcn_exp2f(fmaf(s[kb][r], c, -m));                                 \
        sum += pv;                                                                                      \
        X::cut(pv, ph[kb][r], pl[kb][r]);                                                               \
      }                                                                                                 \
    sum += __shfl_xor(sum, 32, 64);                                                                     \
    l += sum;                                                                                           \
    _Pragma("unroll") for (int kb = 0; kb < 2; ++kb)                                                    \
      _Pragma("unroll") for (int ss = 0; ss < 2; ++ss) {                                                \
        s16x8 th, tl;                                                                                   \
        _Pragma("unroll") for (int j = 0; j < 8; ++j) { th[j] = (short)ph[kb][ss * 8 + j]; tl[j] = (short)pl[kb][ss * 8 + j]; } \
        pfh[kb][ss] = __builtin_bit_cast(v8, th);                                                       \
        pfl[kb][ss] = __builtin_bit_cast(v8, tl);                                                       \
      }                                                                                                 \
  }
#define SVT_X3_PV(ST)                                                                                   \
  {                                                                                                     \
    _Pragma("unroll") for (int db = 0; db < DB; ++db)                                                   \
      _Pragma("unroll") for (int kb = 0; kb < 2; ++kb)                                                  \
        _Pragma("unroll") for (int ss = 0; ss < 2; ++ss) {                                              \
          const char* vp = vbase[db] + (ST) * STAGE_BYTES + (kb * 32 + ss * 16) * RB;                   \
          const s16x4 hlo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t)(vp));                 \
          const s16x4 hhi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t)(vp + 8 * RB));        \
          const s16x4 llo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t)(vp + PLANE_BYTES));   \
          const s16x4 lhi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t)(vp + PLANE_BYTES + 8 * RB)); \
          const s16x8 vh = __builtin_shufflevector(hlo, hhi, 0, 1, 2, 3, 4, 5, 6, 7);                   \
          const s16x8 vl = __builtin_shufflevector(llo, lhi, 0, 1, 2, 3, 4, 5, 6, 7);                   \
          o[db] = X::mma(vl, pfh[kb][ss], o[db]);                                                       \
          o[db] = X::mma(vh, pfl[kb][ss], o[db]);                                                       \
          o[db] = X::mma(vh, pfh[kb][ss], o[db]);                                                       \
        }                                                                                               \
  }
  stage_dma(0, 0);
  int st = 0;
  if (wave < 4) {
    for (int tile = 0; tile < ntiles; ++tile) {
      const int st_next = st + 1 == NST ? 0 : st + 1;
      SVT_X3_BAR(tile)
      SVT_X3_S(st)
      SVT_X3_SM(tile)
      SVT_X3_PV(st)
      st = st_next;
    }
  } else {
    {
      const int st_next = 1;
      SVT_X3_BAR(0)
    }
    for (int tile = 0; tile < ntiles; ++tile) {
      const int st1 = st + 1 == NST ? 0 : st + 1;
      SVT_X3_S(st)
      SVT_X3_SM(tile)
      if (tile + 1 < ntiles) {
        const int st_next = st1 + 1 == NST ? 0 : st1 + 1;
        SVT_X3_BAR(tile + 1)
      }
      SVT_X3_PV(st)
      st = st1;
    }
  }
#undef SVT_X3_BAR
#undef SVT_X3_S
#undef SVT_X3_SM
#undef SVT_X3_PV
  const int q = q0 + (lane & 31);
  if (q >= T) return;
  const float inv = 1.f / l;
  float* op = O + (long)b * o_bstride + (long)q * ldo + (long)h * DH;
  if (o_pairs) {
    char* pp = (char*)op;
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
      for (int g = 0; g < 4; g += 2) {
        unsigned hx[2][2], lx[2][2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          unsigned short ph_[4], pl_[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) X::cut(o[db][(g + u) * 4 + j] * inv, ph_[j], pl_[j]);
          hx[u][0] = (unsigned)ph_[0] | ((unsigned)ph_[1] << 16); hx[u][1] = (unsigned)ph_[2] | ((unsigned)ph_[3] << 16);
          lx[u][0] = (unsigned)pl_[0] | ((unsigned)pl_[1] << 16); lx[u][1] = (unsigned)pl_[2] | ((unsigned)pl_[3] << 16);
        }
#pragma unroll
        for (int w = 0; w < 2; ++w) {
          const auto rh = __builtin_amdgcn_permlane32_swap(hx[0][w], hx[1][w], false, false);
          hx[0][w] = rh[0]; hx[1][w] = rh[1];
          const auto rl = __builtin_amdgcn_permlane32_swap(lx[0][w], lx[1][w], false, false);
          lx[0][w] = rl[0]; lx[1][w] = rl[1];
        }
        char* d = pp + db * 128 + (8 * (g + hh)) * 2;
        *(uint4*)d = uint4{hx[0][0], hx[0][1], hx[1][0], hx[1][1]};
        *(uint4*)(d + 64) = uint4{lx[0][0], lx[0][1], lx[1][0], lx[1][1]};
      }
    return;
  }
#pragma unroll
  for (int db = 0; db < DB; ++db)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 v = {o[db][g * 4] * inv, o[db][g * 4 + 1] * inv, o[db][g * 4 + 2] * inv, o[db][g * 4 + 3] * inv};
      *(float4*)(op + db * 32 + 8 * g + 4 * hh) = v;
    }
}

// fp32 (rows, cols) with row pitch ld_src -> 16-bit (hi, lo) planes (rows, cols) with row pitch ld_dst, lo plane `plane`
// elements after the hi plane; cols % 4 == 0
template <bool F16>
__global__ void split_planes_kernel(const float* __restrict__ src, long ld_src, long rows, int cols, unsigned short* __restrict__ dst,
                                    long ld_dst, long plane) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int c4 = cols / 4;
  if (i >= rows * c4) return;
  const long r = i / c4;
  const int c = (int)(i % c4) * 4;
  const float4 v = *(const float4*)(src + r * ld_src + c);
  const float x[4] = {v.x, v.y, v.z, v.w};
  unsigned short hi[4], lo[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) X3T<F16>::cut(x[j], hi[j], lo[j]);
  unsigned short* d = dst + r * ld_dst + c;
  *(uint2*)d = uint2{(unsigned)hi[0] | ((unsigned)hi[1] << 16), (unsigned)hi[2] | ((unsigned)hi[3] << 16)};
  *(uint2*)(d + plane) = uint2{(unsigned)lo[0] | ((unsigned)lo[1] << 16), (unsigned)lo[2] | ((unsigned)lo[3] << 16)};
}

}  // namespace

int launch_split_planes(int kind, const float* src, long ld_src, long rows, int cols, void* dst, long ld_dst, long plane, hipStream_t s) {
  if (cols % 4 || ld_src % 4 || ld_dst % 4 || plane % 4) { set_error("split_planes: columns and pitches must be multiples of 4"); return -1; }
  const long n = rows * (cols / 4);
  if (kind == 3) hipLaunchKernelGGL((split_planes_kernel<true>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, ld_src, rows, cols, (unsigned short*)dst, ld_dst, plane);
  else hipLaunchKernelGGL((split_planes_kernel<false>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, ld_src, rows, cols, (unsigned short*)dst, ld_dst, plane);
  SVT_LAUNCH_CHECK();
  return 0;
}

bool flash_attention_x3_ok(int dh) { return dh == 64 || dh == 128; }
// Q / K / V: 16-bit (hi, lo) planes (launch_split_planes); O fp32
int launch_flash_attention_x3(int kind, const void* Q, long ldq, long q_bstride, long q_plane, const void* K, const void* V, long ldk,
                              long k_bstride, long k_plane, float* O, long ldo, long o_bstride, int B, int T, int H, int dh, float scale,
                              hipStream_t s, int o_pairs) {
  if ((ldq | ldk | q_bstride | k_bstride | q_plane | k_plane) % 8 || (ldo | o_bstride) % 4) { set_error("flash_attention_x3: strides"); return -1; }
  if (o_pairs && ((ldo | o_bstride | dh) % 32 || ((uintptr_t)O & 127))) { set_error("flash_attention_x3: pair-row output needs strides in multiples of 32 elements"); return -1; }
  const float c = scale * 1.44269504088896340736f;
  dim3 grid((T + 127) / 128, H, B);
  const bool wide = dh == 64 && T > 128 && (long)B * H * ((T + 255) / 256) >= 512;
  if (wide) grid.x = (T + 255) / 256;
  const double flops = 4.0 * B * H * (double)T * T * dh;
  const unsigned short *q = (const unsigned short*)Q, *k = (const unsigned short*)K, *v = (const unsigned short*)V;
  prof_begin(s);
#define SVT_X3_LAUNCH(DH_, F16_, NW_) hipLaunchKernelGGL((flash_attn_x3_kernel<DH_, F16_, NW_>), grid, dim3(64 * NW_), 0, s, q, ldq, q_bstride, q_plane, k, v, ldk, k_bstride, k_plane, O, ldo, o_bstride, T, H, c, o_pairs)
#ifdef SVT_DIAG
  const bool stag = g_attn_variant == 0;
#else
  const bool stag = true;   // the lockstep 8-wave form below is an A/B arm of `make DIAG=1` (svt_debug_set key 21)
#endif
  if (dh == 64 && wide && stag) {
    const int nqb = (T + 255) / 256;
    const int lds_bytes = 3 * 4 * 512 * 16;   // three stages of four 8 KiB tiles
    const dim3 g1((unsigned)(nqb * B * H));
    if (kind == 3) {
      if (int r_ = ensure_dyn_lds((const void*)flash_attn_x3_stag_kernel<true>, lds_bytes)) return r_;
      hipLaunchKernelGGL((flash_attn_x3_stag_kernel<true>), g1, dim3(512), lds_bytes, s, q, ldq, q_bstride, q_plane, k, v, ldk, k_bstride, k_plane, O,
                         ldo, o_bstride, T, H, c, o_pairs, nqb);
    } else {
      if (int r_ = ensure_dyn_lds((const void*)flash_attn_x3_stag_kernel<false>, lds_bytes)) return r_;
      hipLaunchKernelGGL((flash_attn_x3_stag_kernel<false>), g1, dim3(512), lds_bytes, s, q, ldq, q_bstride, q_plane, k, v, ldk, k_bstride, k_plane, O,
                         ldo, o_bstride, T, H, c, o_pairs, nqb);
    }
  }
#ifdef SVT_DIAG
  else if (dh == 64 && wide) { if (kind == 3) SVT_X3_LAUNCH(64, true, 8); else SVT_X3_LAUNCH(64, false, 8); }
#endif
  else if (dh == 64) { if (kind == 3) SVT_X3_LAUNCH(64, true, 4); else SVT_X3_LAUNCH(64, false, 4); }
  else if (dh == 128) { if (kind == 3) SVT_X3_LAUNCH(128, true, 4); else SVT_X3_LAUNCH(128, false, 4); }
  else { set_error("flash_attention_x3: head_dim must be 64 or 128"); return -1; }
#undef SVT_X3_LAUNCH
  prof_end(s, flops, 0.0, 2);
  SVT_LAUNCH_CHECK();
  return 0;
}

int g_attn_variant = 0;   // svt_debug_set key 21 (A/B of the softmax arithmetic of the 8-wave head_dim-64 kernel)
int g_attn_stamp = 0;  // svt_debug_set key 18: tile stamps of the 8-wave head_dim-64 kernel (tools/attn_bench.py --stamps)
int g_flash_wide = 1;  // svt_debug_set key 8: 1 = 8-wave (256-query) workgroups where they pay, 0 = 4-wave ones.  (Round 3: four-wave workgroups held to
                       // three per CU by 16 KiB of unused LDS -- 1 536 workgroups = exactly two rounds instead of 1.5 -- measured 51.1 against 48.5 us
                       // at C2, 138.6 against 130.9 at C3: the kernel is bound by issue throughput, not by the half-empty second round.  A four-stage
                       // K / V ring with three tiles in flight (asm LDS-DMA, counted vmcnt): 47.9 against 46.6-49.1 us, 131 against 125-128:
                       // not waiting for K / V either.  Both removed.)
// V row-major (same layout and strides as K): no transposed copy of V is needed
int launch_flash_attention(const void* Q, long ldq, long q_bstride, const void* K, const void* V, long ldk, long k_bstride,
                           void* O, long ldo, long o_bstride, int B, int T, int H, int dh, float scale, hipStream_t s,
                           const float* gate, const float* pb) {
  if ((ldq | ldk | ldo | q_bstride | k_bstride | o_bstride) % 8 || ((uintptr_t)O & 15) || ((uintptr_t)Q & 15)) {
    set_error("flash_attention: strides must be multiples of 8 elements and Q / O 16-byte aligned"); return -1; }
  const float c = scale * 1.44269504088896340736f;
  dim3 grid((T + 127) / 128, H, B);
  // eight-wave workgroups (256 queries) halve the K / V traffic per query; used when a head has more than 128 queries
  // (measured 45.7 vs 46.9 us at 32 x 12 heads x 499 frames, 114.2 vs 118.7 us at 64 x 16; with few workgroups -- one 5 s
  // utterance: 12 -- the four-wave form spreads over more CUs and stays)
#ifdef SVT_DIAG
  const int variant = g_attn_variant;
#else
  const int variant = 0;    // every other value selects an A/B arm that only `make DIAG=1` builds (svt_debug_set key 21)
#endif
  const bool wide = g_flash_wide && dh == 64 && T > 128 && (long)B * H * ((T + 255) / 256) >= 512 && !(variant >= 5 && variant <= 7);
  if (wide) grid.x = (T + 255) / 256;
  const double flops = 4.0 * B * H * (double)T * T * dh;
  if (gate && pb) {
    if (dh != 64) { set_error("flash_attention: the relative-position-bias variant is built for head_dim 64"); return -1; }
    const size_t dyn = (size_t)(2 * T - 1) * 4;
    if (dyn > 32768) { set_error("flash_attention: sequence too long for the LDS-resident position-bias row"); return -1; }
    prof_begin(s);
    if (wide)
      hipLaunchKernelGGL((flash_attn_kernel<64, true, 8>), grid, dim3(512), dyn, s, (const bf16_t*)Q, ldq, q_bstride, (const bf16_t*)K,
                         ldk, k_bstride, (const bf16_t*)V, 0, (bf16_t*)O, ldo, o_bstride, T, H, c, gate, pb);
    else
      hipLaunchKernelGGL((flash_attn_kernel<64, true>), grid, dim3(256), dyn, s, (const bf16_t*)Q, ldq, q_bstride, (const bf16_t*)K,
                         ldk, k_bstride, (const bf16_t*)V, 0, (bf16_t*)O, ldo, o_bstride, T, H, c, gate, pb);
    prof_end(s, flops, 0.0, 2);
    SVT_LAUNCH_CHECK();
    return 0;
  }
  prof_begin(s);
#ifdef SVT_DIAG
  if (dh == 64 && wide && g_attn_stamp)   // tile stamps (make DIAG=1)
    hipLaunchKernelGGL((flash_attn_kernel<64, false, 8, true>), grid, dim3(512), 0, s, (const bf16_t*)Q, ldq, q_bstride, (const bf16_t*)K,
                       ldk, k_bstride, (const bf16_t*)V, 0, (bf16_t*)O, ldo, o_bstride, T, H, c, nullptr, nullptr);
  else
#endif
#ifdef SVT_DIAG
  if (dh == 64 && wide && variant == 85) {
    // software-pipelined form (flash_attn_pipe_kernel, round 5)
    const int nqb = (T + 255) / 256;
    hipLaunchKernelGGL((flash_attn_pipe_kernel<64, 8, 5, 2>), dim3((unsigned)(nqb * B * H)), dim3(512), 0, s, (const bf16_t*)Q, ldq, q_bstride,
                       (const bf16_t*)K, ldk, k_bstride, (const bf16_t*)V, (bf16_t*)O, ldo, o_bstride, T, H, c, nqb, (unsigned*)nullptr);
  } else if (dh == 64 && wide && variant == 99) {   // stamps (tools/attn_pipe_stamps.py)
    const int nqb = (T + 127) / 128;
    hipLaunchKernelGGL((flash_attn_pipe_kernel<64, 4, 3, 2>), dim3((unsigned)(nqb * B * H)), dim3(256), 0, s, (const bf16_t*)Q, ldq, q_bstride,
                       (const bf16_t*)K, ldk, k_bstride, (const bf16_t*)V, (bf16_t*)O, ldo, o_bstride, T, H, c, nqb, (unsigned*)O);
  } else if (dh == 64 && wide && variant >= 40 && variant < 50) {   // experiments: 4x = 128-query workgroups, ring depth x
    const int nqb = (T + 127) / 128;
#define SVT_PIPE4(NR_) hipLaunchKernelGGL((flash_attn_pipe_kernel<64, 4, NR_, 2>), dim3((unsigned)(nqb * B * H)), dim3(256), 0, s, (const bf16_t*)Q, ldq, \
                       q_bstride, (const bf16_t*)K, ldk, k_bstride, (const bf16_t*)V, (bf16_t*)O, ldo, o_bstride, T, H, c, nqb, (unsigned*)nullptr)
    if (variant == 43) SVT_PIPE4(3); else if (variant == 44) SVT_PIPE4(4); else SVT_PIPE4(5);
#undef SVT_PIPE4
  } else if (dh == 64 && wide && variant >= 80 && variant < 90) {   // experiments: 8x = 256-query workgroups, ring depth x
    const int nqb = (T + 255) / 256;
#define SVT_PIPE8(NR_) hipLaunchKernelGGL((flash_attn_pipe_kernel<64, 8, NR_, 2>), dim3((unsigned)(nqb * B * H)), dim3(512), 0, s, (const bf16_t*)Q, ldq, \
                       q_bstride, (const bf16_t*)K, ldk, k_bstride, (const bf16_t*)V, (bf16_t*)O, ldo, o_bstride, T, H, c, nqb, (unsigned*)nullptr)
    if (variant == 83) SVT_PIPE8(3); else if (variant == 84) SVT_PIPE8(4); else SVT_PIPE8(6);
#undef SVT_PIPE8
  } else
#endif
  if (dh == 64 && wide && (variant == 3 || variant == 0)) {
    // staggered form (flash_attn_stag_kernel, round 4): three K / V stages, waves 4-7 half a tile behind waves 0-3
    const int nqb = (T + 255) / 256;
    hipLaunchKernelGGL((flash_attn_stag_kernel<64, false>), dim3((unsigned)(nqb * B * H)), dim3(512), 0, s, (const bf16_t*)Q, ldq, q_bstride,
                       (const bf16_t*)K, ldk, k_bstride, (const bf16_t*)V, (bf16_t*)O, ldo, o_bstride, T, H, c, nqb);
  }
#ifdef SVT_DIAG
  else if (dh == 64 && wide && variant == 2) {   // PIPE form (measured slower: 43.0-43.9 against 41.0-42.1 us at C2), make DIAG=1
    const int nqb = (T + 255) / 256;
    hipLaunchKernelGGL((flash_attn_stag_kernel<64, true>), dim3((unsigned)(nqb * B * H)), dim3(512), 0, s, (const bf16_t*)Q, ldq, q_bstride,
                       (const bf16_t*)K, ldk, k_bstride, (const bf16_t*)V, (bf16_t*)O, ldo, o_bstride, T, H, c, nqb);
  }
#endif
#ifdef SVT_DIAG
  else if (dh == 64 && wide)   // the round-3 lockstep 8-wave kernel (A/B arm)
    hipLaunchKernelGGL((flash_attn_kernel<64, false, 8>), grid, dim3(512), 0, s, (const bf16_t*)Q, ldq, q_bstride, (const bf16_t*)K,
                       ldk, k_bstride, (const bf16_t*)V, 0, (bf16_t*)O, ldo, o_bstride, T, H, c, nullptr, nullptr);
  else if (dh == 64 && variant >= 5 && variant <= 7) {
    // experiment: four-wave workgroups with the residency capped by an LDS pad (5: three per CU = 768 slots, two exact rounds of the
    // 1 536 workgroups of C2; 6: two per CU; 7: four, the uncapped default of the narrow form)
    const size_t pad = variant == 5 ? 20480 : variant == 6 ? 40960 : 0;
    if (ensure_dyn_lds((const void*)flash_attn_kernel<64>, (int)pad)) return -1;
    dim3 g4((T + 127) / 128, H, B);
    hipLaunchKernelGGL((flash_attn_kernel<64>), g4, dim3(256), pad, s, (const bf16_t*)Q, ldq, q_bstride, (const bf16_t*)K,
                       ldk, k_bstride, (const bf16_t*)V, 0, (bf16_t*)O, ldo, o_bstride, T, H, c, nullptr, nullptr);
  }
#endif
  else if (dh == 64)
    hipLaunchKernelGGL((flash_attn_kernel<64>), grid, dim3(256), 0, s, (const bf16_t*)Q, ldq, q_bstride, (const bf16_t*)K,
                       ldk, k_bstride, (const bf16_t*)V, 0, (bf16_t*)O, ldo, o_bstride, T, H, c, nullptr, nullptr);
  else if (dh == 128)
    hipLaunchKernelGGL((flash_attn_kernel<128>), grid, dim3(256), 0, s, (const bf16_t*)Q, ldq, q_bstride,
                       (const bf16_t*)K, ldk, k_bstride, (const bf16_t*)V, 0, (bf16_t*)O, ldo, o_bstride, T, H, c, nullptr, nullptr);
  else { set_error("flash_attention: head_dim must be 64 or 128"); return -1; }
  prof_end(s, flops, 0.0, 2);
  SVT_LAUNCH_CHECK();
  return 0;
}
}  // namespace svt

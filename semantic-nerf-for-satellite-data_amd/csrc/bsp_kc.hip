// K-contiguous GEMM on block-scaled fp16-plane tensors (bsp.h) for gfx950: every wide dense layer of the default
// arithmetic, forward  C = act(A W^T + b)  and backward  dX = (dZ W) * act'(stored activation)  (+ bias-gradient sums).
// Reference: semantic/models/rs_semantic.py:325-340 (trunk), :260-313 (heads) and their autograd backward.
//
// 128-point x 256-column tile, four waves side by side (wave w: all 128 points x columns 64 w .. 64 w + 63), two workgroups
// per CU.
//   A (activations, streamed once from HBM): LDS-DMA ring of 32-deep stages in the tensor's own byte order, read by every
//     wave with ds_read_b128 (chunk position ^ ((row >> 1) & 7): conflict-free).
//   W (weights, L2-resident, packed in fragment order "WF16"): each wave loads ITS OWN fragments straight into registers,
//     two sub-steps ahead, as asm statements with a hand-counted s_waitcnt vmcnt(6).
// The matrix cores get W as their A operand and the activations as their B operand, i.e. they compute C^T: the
// accumulators of lane l then hold ONE point (l & 31) and, thanks to the row order of the packs (bsp.h: wf16_row), two
// runs of eight consecutive output columns per 32 x 32 block -- exactly one 16-byte piece of the hi plane and one of the
// lo plane per 16-column group.  The epilogue therefore works on the accumulators where they are: no transposition
// through LDS, no exchange between lanes, 16-byte stores / loads of the stored activation straight from the lane's own
// registers.  (Round 2 computed C, moved every tile through an LDS strip and exchanged halves inside lane quads:
// ~5 of its ~28 instructions per output element; the sine was another 16.)
// Epilogues:
//   ACT_SIN  sin(w0 (z + b)) in ONE pass over the accumulators.  The argument is formed in half-turns u = (w0 / pi)(z + b)
//            by the same FMA that applies scale and bias; k = round(u) comes out of the mantissa of u + 1.5 * 2^23,
//            f = u - k is exact, sin(pi f) is a degree-9 odd polynomial on [-1/2, 1/2], the sine's sign is bit 0 of k, and so
//            is the sign of cos (cos(pi f) >= 0): the sign word of the derivative costs one v_alignbit per element.  The
//            outputs lie in [-1, 1], so the block exponent is the constant 13 (as for the positional encoding): no block
//            maximum, no second pass, no workgroup barrier.  Error: 2e-7 + 6e-8 |w0 z| (the rounding of u), the size of the
//            reference's own fp32 rounding of w0 z.
//   others   (plain / ReLU forward, derivative epilogues of the backward pass) two passes: values + block |max| (two waves
//            share a 128-column exponent block: one exchange through LDS), then split into planes and store.
#include "bsp_dev.h"

namespace snerf {
namespace bsp {

constexpr int KC_A = 128 * 128, KC_RING = 3;     // one stage: 128 rows x 32 k (two 16-column groups, 128 B per row)
constexpr int KC_RINGB = KC_RING * KC_A;         // 48 KiB
constexpr int KC_STRIPB = 32 * 272;              // one wave's plane strip: 32 points x (256 + 16) B
constexpr int KC_HREG = 4 * 16384;               // epilogue: two 8 KiB stored-activation regions per wave (overlay the ring)
constexpr int KC_BIAS = KC_HREG > KC_RINGB ? KC_HREG : KC_RINGB;   // 256 floats: the tile's bias (times w0 / pi for the sine epilogue)
constexpr int KC_ETAB = KC_BIAS + 1024;          // 128 ints: exponent of every 16-deep k-step
constexpr int KC_SMAX = KC_ETAB + 512;           // 4 floats: the waves' maxima
constexpr int KC_HSIGN = KC_SMAX + 64;           // two 256-byte sign-word slots per wave
constexpr int KC_LDS = KC_HSIGN + 4 * 512;
constexpr float INV_PI = 0.31830988618379067154f;
constexpr unsigned OOBH = 0x80000000u;           // rejected voffset that survives the addition of an instruction offset

constexpr int SIN_POLY = 0, SIN_HW = 1;

// sin(pi u_c) in place for eight values.  SIGNS: bit "cos(pi u_c) < 0" (= parity of round(u_c)) enters `sw` from the top,
// earlier bits move down (after 32 calls' worth the first element sits in bit 0).
template <bool SIGNS, int SINM>
__device__ __forceinline__ void sinpi8(float (&u)[8], unsigned& sw) {
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const float t = u[c] + 12582912.f;                 // low mantissa bits = k = round(u)
    const unsigned tb = __float_as_uint(t);
    float s;
    if (SINM == SIN_HW) {
      float fr;
      asm("v_fract_f32 %0, %1" : "=v"(fr) : "v"(0.5f * u[c]));
      asm("v_sin_f32 %0, %1" : "=v"(s) : "v"(fr));
    } else {
      const float kf = t - 12582912.f;
      const float f = u[c] - kf;                       // exact, |f| <= 1/2
      const float f2 = f * f;
      float q = fmaf(f2, 0.077218386155008978f, -0.59804419391100816f);
      q = fmaf(q, f2, 2.5500311935191413f);
      q = fmaf(q, f2, -5.1677068661679284f);
      q = fmaf(q, f2, 3.1415925798055815f);
      s = __uint_as_float((tb << 31) + __float_as_uint(f * q));   // (-1)^k: one v_lshl_add
    }
    u[c] = s;
    if (SIGNS) sw = __builtin_amdgcn_alignbit(tb, sw, 1);
  }
}

// sum over the 32 lanes that share l >> 5 (DPP adds inside the 16-lane rows, row_bcast15 across the pair of rows); valid in
// lanes 16-31 (l >> 5 == 0) and 48-63 (l >> 5 == 1)
__device__ __forceinline__ float sum32(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));   // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));   // row_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xA, 0xF, false));  // row_bcast15 -> rows 1, 3
  return v;
}

template <int ACT, int AUX, bool COLSUM, bool SIGNS, int SINM>
__global__ __launch_bounds__(256, 2) void gemm_kc_kernel(const KcArgs p) {
  __shared__ __attribute__((aligned(16))) char lds[KC_LDS];
  float* sbias = reinterpret_cast<float*>(lds + KC_BIAS);
  int* etab = reinterpret_cast<int*>(lds + KC_ETAB);
  float* smax = reinterpret_cast<float*>(lds + KC_SMAX);
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wj0 = wave * 64;
  int ti, tj;
  tile_of_block(blockIdx.x, p.tiles_i, p.tiles_j, ti, tj);
  const int i0 = ti * 128, j0 = tj * 256;
  const int nks = p.K >> 4, nks1 = p.Ka >> 4;
  constexpr bool ONEPASS = ACT == ACT_SIN;       // outputs in [-1, 1]: constant block exponent

  // ---- A: per-lane DMA sources.  A stage is 32 k deep = 128 rows x 128 B (two column groups, the tensor's own byte
  // order); its sixteen 1 KiB pieces (8 rows each) go to the waves round-robin, four per wave, two per 16-deep sub-step.
  // The 16 B chunk c of a row sits at position c ^ ((row >> 1) & 7): a quarter-wave of ds_read_b128 (eight lanes, eight
  // consecutive rows, one chunk) then covers four distinct positions twice -> all 32 banks in two passes, no conflict.
  const int nst = (nks + 1) >> 1, nst1 = nks1 >> 1;          // stages; stages of the first segment (Ka % 32 == 0 if two)
  const srd_t srdA = make_srd(p.A + ((size_t)i0 * p.lda + p.a_col0) * 4,
                              clamp_bytes(i0 < p.I ? ((unsigned long long)(p.I - i0 - 1) * p.lda + p.Ka) * 4ull : 0ull));
  const srd_t srdA2 = make_srd(p.A2 + ((size_t)i0 * p.lda2 + p.a2_col0) * 4,
                               clamp_bytes(i0 < p.I ? ((unsigned long long)(p.I - i0 - 1) * p.lda2 + (p.K - p.Ka)) * 4ull : 0ull));
  unsigned voA[4], voA2[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int row = 8 * (wave + 4 * q) + (lane >> 3);
    const unsigned c = (unsigned)((lane & 7) ^ ((row >> 1) & 7));
    const bool in = i0 + row < p.I;
    voA[q] = in ? (unsigned)row * (unsigned)p.lda * 4u + 16u * c : OOB;
    voA2[q] = in ? (unsigned)row * (unsigned)p.lda2 * 4u + 16u * c : OOB;
  }
  char* const dst0 = lds + wave * 1024;
  // The descriptor, the lanes' offsets and the stage bias of the segment being requested are loop-carried and replaced
  // ONCE, at the stage where the second segment starts (selecting them per request cost ~16 scalar instructions per
  // piece, more than an MFMA gap hides); a stage beyond K is rejected through the scalar offset.
  srd_t srdCur = srdA;
  unsigned voCur[4] = {voA[0], voA[1], voA[2], voA[3]};
  int sbias_st = 0;
  const int seg_switch = p.Ka < p.K ? nst1 : 0x7fffffff;
  auto enter_stage = [&](int S) {     // before the first piece of stage S
    if (__builtin_expect(S == seg_switch, 0)) {
      srdCur = srdA2; sbias_st = nst1;
#pragma unroll
      for (int q = 0; q < 4; ++q) voCur[q] = voA2[q];
    }
  };
  auto issueA = [&](int S, int slot, int q) {
    dma16(srdCur, dst0 + slot * KC_A + 4096 * q, voCur[q], S < nst ? (unsigned)(S - sbias_st) * 128u : OOB);
  };
  // ---- W: fragment-ordered pack; unit (ks, rb32) = 2 KiB = [plane][lane][16 B]; this wave reads units rb32 = u0, u0 + 1
  const srd_words srdW = make_srd_words(p.W, p.w_bytes);
  const unsigned w_u0 = (unsigned)((p.w_row0 + j0 + wj0) >> 5), w_ks0 = (unsigned)(p.w_k0 >> 4);
  const unsigned voW = 16u * (unsigned)lane;
  struct BFrag { u32x4 h[2], l[2]; };
  // The weight loads are asm statements with their completion counted by hand (wait_b below).  As builtins, hipcc's own
  // vm-counter bookkeeping put conservative waits behind them at the loop header (the two-step prefetch was undone) and
  // re-used the registers of the fragment set that is dead at the header as VALU temporaries, each with a WAW wait on a
  // load.  A step beyond K is rejected through the scalar offset.
  auto loadB2 = [&](int s, BFrag& b, int half) {     // two of the four weight loads of sub-step s
    const unsigned so = s < nks ? ((w_ks0 + (unsigned)s) * (unsigned)p.w_rb32 + w_u0) * 2048u : OOB;
    if (half == 0) {
      asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(b.h[0]) : "v"(voW), "s"(srdW), "s"(so));
      asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:1024" : "=v"(b.l[0]) : "v"(voW), "s"(srdW), "s"(so));
    } else {
      asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:2048" : "=v"(b.h[1]) : "v"(voW), "s"(srdW), "s"(so));
      asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:3072" : "=v"(b.l[1]) : "v"(voW), "s"(srdW), "s"(so));
    }
  };
  // everything but the six youngest requests (= the previous sub-step's) has landed; names the fragments so that no use of
  // them can be scheduled above the wait
  auto wait_b = [&](BFrag& b) {
    asm volatile("s_waitcnt vmcnt(6)" : "+v"(b.h[0]), "+v"(b.l[0]), "+v"(b.h[1]), "+v"(b.l[1])::"memory");
  };

  // ---- exponents along k: table in LDS + a bit per step where the scale changes (128 steps at most: K <= 2048).
  const int ti_e = min(ti, (p.I + 127) / 128 - 1);
  const int sA = lane, sB = lane + 64;
  const int eA = sA < nks ? kc_exp_of_step(p, ti_e, sA, nks1) : 0;
  const int eB = sB < nks ? kc_exp_of_step(p, ti_e, sB, nks1) : 0;
  const int eAp = (sA > 0 && sA < nks) ? kc_exp_of_step(p, ti_e, sA - 1, nks1) : eA;
  const int eBp = sB < nks ? kc_exp_of_step(p, ti_e, sB - 1, nks1) : eB;
  const int e_last = kc_exp_of_step(p, ti_e, nks - 1, nks1);
  float bias_t = 0.f;                              // the tile's bias, one column per thread
  if (AUX == AUX_NONE && p.bias != nullptr && j0 + t < p.J) bias_t = p.bias[j0 + t];
  // Both operands are requested ahead: W two sub-steps, A two stages.  vm-counter order: [W(0) A(0) x 4] [W(1) A(1) x 4],
  // then per sub-step s = 2 S + u: [W(s + 2) x 4] [two pieces of A(S + 2)] -- at the top of sub-step s the six requests
  // of sub-step s - 1 may be outstanding and everything older has landed, which covers W(s) and all of stage S (issued
  // during stage S - 2): s_waitcnt vmcnt(6).
  BFrag bq0, bq1, bq2;
  loadB2(0, bq0, 0); loadB2(0, bq0, 1);
  issueA(0, 0, 0); issueA(0, 0, 1); issueA(0, 0, 2); issueA(0, 0, 3);
  loadB2(1, bq1, 0); loadB2(1, bq1, 1);
  enter_stage(1);
  issueA(1, 1, 0); issueA(1, 1, 1); issueA(1, 1, 2); issueA(1, 1, 3);
  if (wave == 0) { etab[sA] = eA; etab[sB] = eB; }
  if (AUX == AUX_NONE) sbias[t] = bias_t * (ACT == ACT_SIN ? p.w0 * INV_PI : 1.f);
  const unsigned long long chg0 = __builtin_amdgcn_ballot_w64(eA != eAp), chg1 = __builtin_amdgcn_ballot_w64(eB != eBp);

  f32x16 acc[4][2];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int nj = 0; nj < 2; ++nj)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][nj][r] = 0.f;

  // A fragment addresses: lane -> (row l & 31, k half l >> 5); chunk (4 u + 2 pl + half) of sub-step u sits at position
  // chunk ^ ((row >> 1) & 7)
  const int rowl = lane & 31, kh = lane >> 5, swz = (rowl >> 1) & 7;
  unsigned fo[2][2];
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) fo[u][pl] = (unsigned)rowl * 128u + (unsigned)(((4 * u + 2 * pl + kh) ^ swz) << 4);

  // One 16-deep sub-step s = 2 S + u of stage S: 24 MFMAs on the fragments `fa` (read from LDS during the PREVIOUS sub-step)
  // and the weight registers `bc`; `bn` receives the weight fragments of sub-step s + 2.  A single wave issues one
  // instruction per ~4 cycles, an MFMA occupies the matrix pipe for 32: whatever is issued in a block of its own leaves the
  // pipe idle, so every non-MFMA instruction sits in a gap between MFMAs: per 32-point block mi the six MFMAs run hi*lo,
  // lo*hi, hi*hi on two accumulators each, the lo fragment of mi is re-read for sub-step s + 1 as soon as its last MFMA has
  // been issued, the hi fragment after the block, and the four weight loads / two DMA pieces are spread over the blocks.
  // The workgroup barrier of a new stage comes in the middle of the last sub-step of its predecessor (before the first
  // read of the new stage): every wave has passed the top-of-sub-step wait that covers its own pieces of stage S + 1
  // (issued during stage S - 1) and has finished reading stage S - 1, whose slot the requests of stage S + 2 (issued after
  // that point in program order) overwrite.
  struct AFrag { f16x8 h[4], l[4]; };
  AFrag fa;
  auto step = [&](int s, int slot, int u, BFrag& bc, BFrag& bn) {
    wait_b(bc);
    if (__builtin_expect((((s & 64) ? chg1 : chg0) >> (s & 63)) & 1ull, 0)) {
      const int de = etab[s] - etab[s - 1];
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int nj = 0; nj < 2; ++nj) acc[mi][nj] = scale_acc(acc[mi][nj], de);
    }
    const char* sn = lds + (u ? (slot + 1) % KC_RING : slot) * KC_A;   // where sub-step s + 1 reads
    const f16x8 bh0 = __builtin_bit_cast(f16x8, bc.h[0]), bh1 = __builtin_bit_cast(f16x8, bc.h[1]);
    const f16x8 bl0 = __builtin_bit_cast(f16x8, bc.l[0]), bl1 = __builtin_bit_cast(f16x8, bc.l[1]);
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      acc[mi][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl0, fa.h[mi], acc[mi][0], 0, 0, 0);
      acc[mi][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl1, fa.h[mi], acc[mi][1], 0, 0, 0);
      if (mi < 2) loadB2(s + 2, bn, mi);
      if (mi == 2 && u == 0) enter_stage((s >> 1) + 2);
      if (mi >= 2) issueA((s >> 1) + 2, (slot + 2) % KC_RING, 2 * u + (mi - 2));   // two of the four pieces of stage S + 2
      acc[mi][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh0, fa.l[mi], acc[mi][0], 0, 0, 0);
      acc[mi][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh1, fa.l[mi], acc[mi][1], 0, 0, 0);
      if (mi == 0 && u == 1) { __builtin_amdgcn_sched_barrier(0); barrier_raw(); __builtin_amdgcn_sched_barrier(0); }
      const f16x8 nl = ldsfrag(sn + 4096 * mi + fo[u ^ 1][1]);
      acc[mi][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh0, fa.h[mi], acc[mi][0], 0, 0, 0);
      acc[mi][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh1, fa.h[mi], acc[mi][1], 0, 0, 0);
      fa.l[mi] = nl;
      fa.h[mi] = ldsfrag(sn + 4096 * mi + fo[u ^ 1][0]);
      __builtin_amdgcn_sched_barrier(0);   // keep the blocks apart: left alone, the scheduler gathers the reads at the end
    }
  };
  // fragments of sub-step 0: stage 0 (and W(0)) are home when all but the eight youngest requests (W(1), stage 1) are
  wait_vm<8>();
  barrier_raw();
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) { fa.h[mi] = ldsfrag(lds + 4096 * mi + fo[0][0]); fa.l[mi] = ldsfrag(lds + 4096 * mi + fo[0][1]); }
  // The k-loop runs at raised priority: while the SIMD's other wave (the co-resident workgroup) is in its VALU-heavy
  // epilogue, MFMA issue goes first and the epilogue stream fills the 24 of every 32 cycles the matrix pipe leaves free.
  __builtin_amdgcn_s_setprio(2);
  for (int s = 0; s < 2 * nst; s += 6) {   // an odd count of 16-deep steps runs one sub-step on zero weights
    step(s, 0, 0, bq0, bq2);
    step(s + 1, 0, 1, bq1, bq0);
    if (s + 2 < 2 * nst) { step(s + 2, 1, 0, bq2, bq1); step(s + 3, 1, 1, bq0, bq2); }
    if (s + 4 < 2 * nst) { step(s + 4, 2, 0, bq1, bq0); step(s + 5, 2, 1, bq2, bq1); }
  }
  __builtin_amdgcn_s_setprio(0);

  // ---- epilogue.  Lane l: point pt = l & 31 of each 32-point block mi; register r of block (mi, nj) is column
  //      64 wave + 32 nj + 16 (r >> 3) + 8 (l >> 5) + (r & 7) of the tile.  The ring is dead from here on: its LDS holds the
  //      waves' plane strips (results on their way out) and, for the derivative epilogues, the stored activations on their
  //      way in.
  wait_vm<0>();        // rejected requests behind the last stage write zeros into the ring: drain before re-using it
  barrier_raw();
  const int e_in = e_last + *p.EW;          // acc = true value * 2^e_in
  const bool e_small = e_in >= -120 && e_in <= 120;
  if (!e_small) {   // exponents beyond a single fp32 factor (never with sane data): scale the accumulators first
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int nj = 0; nj < 2; ++nj) acc[mi][nj] = scale_acc(acc[mi][nj], -e_in);
  }
  const float inv_in = e_small ? pow2f(-e_in) : 1.f;
  const int pt = lane & 31, lh = lane >> 5;
  const int nrows = min(128, p.I - i0);                        // > 0: the grid covers ceil(I / 128) row tiles
  const int jw = j0 + wj0;                                     // first column of the wave
  const bool wave_cols = jw < p.J;
  const size_t offC = uniform_sz(((size_t)i0 * p.ldc + p.c_col0) * 4);
  const srd_t srdC = make_srd(p.C + offC, clamp_bytes(((unsigned long long)(nrows - 1) * p.ldc + p.J) * 4ull));
  // Plane strip of one 32-point block: [32 points][272 B] = the wave's 64 columns as they lie in memory (four 64-byte
  // groups [hi | lo]) + 16 B of padding.  A lane writes its 16-byte pieces (eight consecutive lanes: 8 points x 16 B at a
  // 272-byte pitch = all 32 banks), then the wave reads the strip back four whole rows per instruction and stores 4 x 256
  // contiguous bytes.
  char* const strip = lds + wave * KC_STRIPB;
  const unsigned sw_off = (unsigned)pt * 272u + 16u * (unsigned)lh;                       // + 64 gq (+ 32: lo plane)
  const int srow = lane >> 4, schunk = lane & 15;
  const unsigned sr_off = (unsigned)srow * 272u + 16u * (unsigned)schunk;                 // + 4 * 272 per pass
  const bool chunk_ok = jw + 16 * (schunk >> 2) < p.J;                                    // J % 16 == 0
  const unsigned voC = chunk_ok ? (unsigned)srow * (unsigned)p.ldc * 4u + (unsigned)(jw >> 4) * 64u + 16u * (unsigned)schunk : OOBH;
  const unsigned stepC4 = 4u * (unsigned)p.ldc * 4u;
  auto strip_put = [&](int gq, const u32x4& hi, const u32x4& lo) {
    *reinterpret_cast<u32x4*>(strip + sw_off + 64 * gq) = hi;
    *reinterpret_cast<u32x4*>(strip + sw_off + 64 * gq + 32) = lo;
  };
  auto strip_flush = [&](int mi) {     // rows beyond I are rejected by the descriptor
#pragma unroll
    for (int ps = 0; ps < 8; ++ps) {
      const u32x4 d = *reinterpret_cast<const u32x4*>(strip + sr_off + 4 * 272 * ps);
      __builtin_amdgcn_raw_buffer_store_b128(d, srdC, voC + (unsigned)(8 * mi + ps) * stepC4, 0, 0);
    }
  };
  float bj[4][8];
  if (AUX == AUX_NONE) {
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      const float4 b0 = *reinterpret_cast<const float4*>(&sbias[wj0 + 16 * gq + 8 * lh]);
      const float4 b1 = *reinterpret_cast<const float4*>(&sbias[wj0 + 16 * gq + 8 * lh + 4]);
      bj[gq][0] = b0.x; bj[gq][1] = b0.y; bj[gq][2] = b0.z; bj[gq][3] = b0.w;
      bj[gq][4] = b1.x; bj[gq][5] = b1.y; bj[gq][6] = b1.z; bj[gq][7] = b1.w;
    }
  }

  if constexpr (ONEPASS) {
    // ---- sine: one pass.  u = acc * (2^-e w0 / pi) + b w0 / pi (bias row staged in LDS, already scaled)
    const float su = inv_in * p.w0 * INV_PI;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      unsigned sw = 0u;
#pragma unroll
      for (int nj = 0; nj < 2; ++nj)
#pragma unroll
        for (int gg = 0; gg < 2; ++gg) {
          const int gq = 2 * nj + gg;
          float v[8];
#pragma unroll
          for (int c = 0; c < 8; ++c) v[c] = fmaf(acc[mi][nj][8 * gg + c], su, bj[gq][c]);
          sinpi8<SIGNS, SINM>(v, sw);
          u32x4 hi, lo;
          split8(v, 8192.f, hi, lo);
          strip_put(gq, hi, lo);
        }
      if (SIGNS && p.Csign != nullptr && wave_cols && 32 * mi < nrows)
        p.Csign[((size_t)((i0 >> 5) + mi) * ((p.ldc + 63) >> 6) + ((p.c_col0 + jw) >> 6)) * 64 + lane] = sw;
      strip_flush(mi);
    }
    if ((wave & 1) == 0 && lane == 0 && wave_cols) p.EC[(size_t)ti * ncb_of(p.ldc) + ((p.c_col0 + jw) >> 7)] = 13;
  } else {
    // ---- two passes.  Pass A: final values in place of the accumulators, their |max|, column sums.
    float wmax = 0.f;
    float cs[2][16];
#pragma unroll
    for (int nj = 0; nj < 2; ++nj)
#pragma unroll
      for (int r = 0; r < 16; ++r) cs[nj][r] = 0.f;
    int eH = 0;
    if (AUX != AUX_NONE && wave_cols) eH = p.EH[(size_t)ti_e * ncb_of(p.ldh) + ((p.h_col0 + jw) >> 7)];
    const float inv_h = pow2f(-eH);
    // Stored activations (derivative epilogues): block mi of the wave = 32 points x 256 B, fetched by LDS-DMA in eight 1 KiB
    // pieces (4 whole rows each) into one of two 8 KiB regions of the wave, block mi + 1 while block mi is worked on.  Chunk
    // c of point row q lies at position c ^ (q & 15) (swizzle on the source address): the lanes of a ds_read_b128 group
    // hold 16 different points -> 16 different positions.  The sign words (one dword per lane and block) come the same way.
    const size_t offH = uniform_sz(AUX != AUX_NONE ? ((size_t)i0 * p.ldh + p.h_col0) * 4 : 0);
    const srd_words srdH = make_srd_words(AUX != AUX_NONE ? p.H + offH : nullptr,
                                          AUX != AUX_NONE ? clamp_bytes(((unsigned long long)(nrows - 1) * p.ldh + p.J) * 4ull) : 0u);
    const srd_words srdS = make_srd_words(AUX == AUX_SINREC ? p.Hsign : nullptr,
                                          AUX == AUX_SINREC ? clamp_bytes(sign_words((size_t)p.I, p.ldh) * 4ull) : 0u);
    const unsigned hreg0 = __builtin_amdgcn_readfirstlane(lds_addr(lds + wave * 16384));
    const unsigned sreg0 = __builtin_amdgcn_readfirstlane(lds_addr(lds + KC_HSIGN + wave * 512));
    auto dma_h = [&](int mi) {
      if (AUX == AUX_NONE) return;
#pragma unroll
      for (int pc = 0; pc < 8; ++pc) {
        const int q = 4 * pc + (lane >> 4);                       // point row inside the block
        const int c = (lane & 15) ^ (q & 15);                     // global chunk this lane fetches
        const bool ok = jw + 16 * (c >> 2) < p.J;
        const unsigned vo = ok ? (unsigned)(32 * mi + q) * (unsigned)p.ldh * 4u + (unsigned)(jw >> 4) * 64u + 16u * (unsigned)c : OOBH;
        dma16_asm(srdH, hreg0 + (unsigned)((mi & 1) * 8192 + pc * 1024), vo, 0u);
      }
      if (AUX == AUX_SINREC) {
        const bool ok = wave_cols && 32 * mi < nrows;
        const unsigned vo = ok ? (unsigned)((((size_t)((i0 >> 5) + mi) * ((p.ldh + 63) >> 6) + ((p.h_col0 + jw) >> 6)) * 64 + lane) * 4) : OOBH;
        dma4_asm(srdS, sreg0 + (unsigned)((mi & 1) * 256), vo, 0u);
      }
    };
    constexpr int NDMA = AUX == AUX_SINREC ? 9 : 8;               // requests per block
    dma_h(0);
    dma_h(1);
    // derivative epilogues: the accumulator's scale and |w0| in one factor; the sign bits are xor-ed with w0's own sign
    const unsigned w0mag = __float_as_uint(fabsf(p.w0) * inv_in);
    const unsigned sflip = p.w0 < 0.f ? 0xFFFFFFFFu : 0u;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      if (AUX != AUX_NONE) {
        if (mi < 3) wait_vm<NDMA>(); else wait_vm<0>();           // block mi is home (block mi + 1 may be in flight)
      }
      const char* hreg = lds + wave * 16384 + (mi & 1) * 8192;
      const float okf = (32 * mi + pt) < nrows ? 1.f : 0.f;      // points beyond I: out of the maximum and the column sums
      unsigned sword = 0u;
      if (AUX == AUX_SINREC) sword = *reinterpret_cast<const unsigned*>(lds + KC_HSIGN + wave * 512 + (mi & 1) * 256 + lane * 4) ^ sflip;
      u32x4 hh[4], hl[4];
      if (AUX != AUX_NONE) {
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          hh[gq] = *reinterpret_cast<const u32x4*>(hreg + pt * 256 + 16 * ((4 * gq + lh) ^ (pt & 15)));
          hl[gq] = *reinterpret_cast<const u32x4*>(hreg + pt * 256 + 16 * ((4 * gq + 2 + lh) ^ (pt & 15)));
        }
        if (mi + 2 < 4) {   // the region is free once these reads have returned
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(hh[0]), "+v"(hh[1]), "+v"(hh[2]), "+v"(hh[3]), "+v"(hl[0]), "+v"(hl[1]), "+v"(hl[2]), "+v"(hl[3]), "+v"(sword)::"memory");
          dma_h(mi + 2);
        }
      }
#pragma unroll
      for (int nj = 0; nj < 2; ++nj)
#pragma unroll
        for (int gg = 0; gg < 2; ++gg) {
          const int gq = 2 * nj + gg;
          float v[8];
#pragma unroll
          for (int c = 0; c < 8; ++c) {
            const float x = acc[mi][nj][8 * gg + c];
            v[c] = AUX == AUX_SINREC ? x : (AUX == AUX_NONE ? fmaf(x, inv_in, bj[gq][c]) : x * inv_in);
          }
          if (ACT == ACT_RELU) {
#pragma unroll
            for (int c = 0; c < 8; ++c) v[c] = fmaxf(v[c], 0.f);
          }
          if (AUX != AUX_NONE) {
            float h[8];
            join8(hh[gq], hl[gq], inv_h, h);
            if (AUX == AUX_SINREC) {
              // w0 cos(w0 z) = +-|w0| sqrt(1 - h^2): the sign bit (xor-ed with w0's own sign, once per word) is shifted to bit
              // 31 and merged over |w0| 2^-e by one v_bfi; 1 - h^2 is clamped at 0 by the FMA's output modifier
#pragma unroll
              for (int c = 0; c < 8; ++c) {
                float om;
                asm("v_fma_f32 %0, -%1, %1, 1.0 clamp" : "=v"(om) : "v"(h[c]));
                unsigned w0s_bits;
                asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(w0s_bits) : "s"(0x7fffffffu), "v"(w0mag), "v"(sword << (31 - (16 * nj + 8 * gg + c))));
                v[c] *= __uint_as_float(w0s_bits) * __builtin_amdgcn_sqrtf(om);
              }
            } else {
#pragma unroll
              for (int c = 0; c < 8; ++c) v[c] = h[c] > 0.f ? v[c] : 0.f;
            }
          }
#pragma unroll
          for (int c = 0; c < 8; ++c) {
            if (COLSUM) cs[nj][8 * gg + c] = fmaf(v[c], okf, cs[nj][8 * gg + c]);
            acc[mi][nj][8 * gg + c] = v[c];
          }
          if (jw + 16 * gq < p.J) wmax = fmaxf(wmax, okf * absmax3(v[6], v[7], absmax3(v[4], v[5], absmax3(v[2], v[3], absmax3(v[0], v[1], 0.f)))));
        }
    }
    if (COLSUM && p.colsum != nullptr) {   // one partial row per 128-point tile
#pragma unroll
      for (int nj = 0; nj < 2; ++nj)
#pragma unroll
        for (int r = 0; r < 16; ++r) cs[nj][r] = sum32(cs[nj][r]);
      if ((lane & 31) == 31) {
#pragma unroll
        for (int gq = 0; gq < 4; ++gq)
          if (jw + 16 * gq < p.J) {
            float* d = p.colsum + (size_t)ti * p.ldcs + jw + 16 * gq + 8 * lh;
            const int nj = gq >> 1, r0 = 8 * (gq & 1);
            *reinterpret_cast<float4*>(d) = make_float4(cs[nj][r0], cs[nj][r0 + 1], cs[nj][r0 + 2], cs[nj][r0 + 3]);
            *reinterpret_cast<float4*>(d + 4) = make_float4(cs[nj][r0 + 4], cs[nj][r0 + 5], cs[nj][r0 + 6], cs[nj][r0 + 7]);
          }
      }
    }
    // block maximum: waves 2 c and 2 c + 1 share the exponent block (ti, column block c of the tile)
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) wmax = fmaxf(wmax, __shfl_xor(wmax, o, 64));
    if (lane == 0) smax[wave] = wmax;
    __syncthreads();   // also: every wave has finished with the stored-activation regions the strips overlay
    const float bmax = fmaxf(smax[wave & 2], smax[(wave & 2) + 1]);
    const int eC = exp_of_maxbits(__float_as_uint(bmax));
    const float sc = pow2f(eC);
    if ((wave & 1) == 0 && lane == 0 && wave_cols) p.EC[(size_t)ti * ncb_of(p.ldc) + ((p.c_col0 + jw) >> 7)] = eC;
    // ---- pass B: split, through the strip, store
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
#pragma unroll
      for (int nj = 0; nj < 2; ++nj)
#pragma unroll
        for (int gg = 0; gg < 2; ++gg) {
          float v[8];
#pragma unroll
          for (int c = 0; c < 8; ++c) v[c] = acc[mi][nj][8 * gg + c];
          u32x4 hi, lo;
          split8(v, sc, hi, lo);
          strip_put(2 * nj + gg, hi, lo);
        }
      strip_flush(mi);
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------------------
int prof_hook_begin(double flops, int variant, hipStream_t st);   // gemm.hip: per-launch HIP events when profiling is on
void prof_hook_end(int token, hipStream_t st);
int check_kc(const KcArgs& a, bool narrow);                        // bsp_gemm.hip

static int sin_mode() {   // SNERF_SIN=hw: v_sin_f32 instead of the polynomial (diagnostic A/B; read once)
  static const int m = [] { const char* e = getenv("SNERF_SIN"); return (e && e[0] == 'h') ? SIN_HW : SIN_POLY; }();
  return m;
}

int launch_kc(const KcArgs& a0, hipStream_t st) {
  KcArgs a = a0;
  if (!a.A2) { a.A2 = a.A; a.EA2 = a.EA; a.lda2 = a.lda; a.a2_col0 = a.a_col0; if (a.Ka == 0) a.Ka = a.K; }
  int rc = check_kc(a, false);
  if (rc) return rc;
  a.tiles_i = (a.I + 127) / 128;
  a.tiles_j = (a.J + 255) / 256;
  const dim3 grid(a.tiles_i * a.tiles_j), block(256);
  const int tok = prof_hook_begin(2.0 * a.I * (double)a.J * a.K, 0, st);
  const bool cs = a.colsum != nullptr;
#define KC_LAUNCH(ACT_, AUX_, CS_, SG_, SM_) hipLaunchKernelGGL((gemm_kc_kernel<ACT_, AUX_, CS_, SG_, SM_>), grid, block, 0, st, a)
  if (a.aux_mode == AUX_SINREC) KC_LAUNCH(ACT_NONE, AUX_SINREC, true, false, SIN_POLY);
  else if (a.aux_mode == AUX_RELU_MASK) KC_LAUNCH(ACT_NONE, AUX_RELU_MASK, true, false, SIN_POLY);
  else if (a.act == ACT_SIN) {
    const bool hw = sin_mode() == SIN_HW;
    if (a.Csign == nullptr) { if (hw) KC_LAUNCH(ACT_SIN, AUX_NONE, false, false, SIN_HW); else KC_LAUNCH(ACT_SIN, AUX_NONE, false, false, SIN_POLY); }
    else { if (hw) KC_LAUNCH(ACT_SIN, AUX_NONE, false, true, SIN_HW); else KC_LAUNCH(ACT_SIN, AUX_NONE, false, true, SIN_POLY); }
  }
  else if (a.act == ACT_RELU) KC_LAUNCH(ACT_RELU, AUX_NONE, false, false, SIN_POLY);
  else if (cs) KC_LAUNCH(ACT_NONE, AUX_NONE, true, false, SIN_POLY);
  else KC_LAUNCH(ACT_NONE, AUX_NONE, false, false, SIN_POLY);
#undef KC_LAUNCH
  SNERF_LAUNCH_CHECK();
  prof_hook_end(tok, st);
  return SNERF_OK;
}

}  // namespace bsp
}  // namespace snerf

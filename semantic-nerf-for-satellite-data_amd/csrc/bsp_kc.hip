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
//   ACT_SIN  sin(w0 (z + b)) in ONE pass over the accumulators.  The argument is formed in revolutions x = (w0 / 2 pi)(z + b)
//            by the same FMA that applies scale and bias and goes straight into v_sin_f32 (bsp_kc_epi.h: sin2pi8); the sign of
//            cos is the parity of round(2 x), the low mantissa bit of 2 x + 1.5 * 2^23: one FMA and one v_alignbit per element
//            for the sign word of the derivative.  The outputs lie in [-1, 1], so the block exponent is the constant 13 (as for
//            the positional encoding): no block maximum, no second pass, no workgroup barrier.  Error: 2e-7 + 6e-8 |w0 z| (the
//            rounding of x), the size of the reference's own fp32 rounding of w0 z.
//   others   (plain / ReLU forward, derivative epilogues of the backward pass) two passes: values + block |max| (two waves
//            share a 128-column exponent block: one exchange through LDS), then split into planes and store.
#include "bsp_kc_epi.h"

namespace snerf {
namespace bsp {

constexpr int KC_A = 128 * 128, KC_RING = 3;     // one stage: 128 rows x 32 k (two 16-column groups, 128 B per row)
constexpr int KC_RINGB = KC_RING * KC_A;         // 48 KiB
constexpr int KC_XREG = KC_RINGB;                // epilogue: 4 KiB per wave (plane strip / second stored-activation buffer)
constexpr int KC_BIAS = KC_XREG + 4 * 4096;      // 2 x 256 floats: the tile's bias (times w0 / pi for the sine epilogue), by tile parity
constexpr int KC_ETAB = KC_BIAS + 2 * 1024;      // 2 x 128 ints: exponent of every 16-deep k-step, by tile parity
constexpr int KC_SMAX = KC_ETAB + 2 * 512;       // 4 floats: the waves' maxima
constexpr int KC_NEXT = KC_SMAX + 64;            // index of the workgroup's next tile (written by wave 0)
constexpr int KC_HSIGN = KC_NEXT + 64;           // four 256-byte sign-word slots per wave (derivative epilogues)
constexpr int KC_LDS = KC_HSIGN + 4 * 1024;
constexpr int KC_NDW = KC_HSIGN;                 // NDOT launches (forward only: no sign-word slots): the projection vectors, <= KC_NDW_FLOATS floats
constexpr int KC_NDW_FLOATS = 2560;              // 10 KiB: e.g. (3 + 5 + 1 + 1) rows of 256 columns (the final head layers), or one row of <= 1024
constexpr int KC_LDS_ND = KC_NDW + 4 * KC_NDW_FLOATS;
// DIAG (tools/ablate/build_diag.sh only; the product instantiates DIAG = false): KcArgs::dbg removes operand traffic through
// zero-size descriptors -- 1: A, 2: W (host side), 4: stores, 8: every tile reads the first 128 rows of A (always L2-resident);
// 16: the epilogue's arithmetic, strips and stores are skipped (tile bookkeeping and next-tile requests stay), 32: no workgroup barrier
// inside the k-loop, 64: no fragment reads inside the k-loop, 128: the stored activation of a derivative epilogue (zero-size descriptor).
// Timing-only: the results are wrong.
// PL = planes per operand / result tensor (bsp.h).  PL = 1: a 16-column group is 32 bytes, so the SAME bytes carry twice the
// contraction depth -- a "sub-step" below is 64 bytes of every A row = 16 k of two planes (3 products) or 32 k of one plane
// (1 product per 16 k): same LDS traffic, same request counts, 16 instead of 24 MFMAs per sub-step; a stage (128 bytes per row)
// is 32 / 64 k deep.
// NDOT (ACT_SIN only): the 1-wide projection that follows the layer (KcArgs::nd_w) is taken in the epilogue, on the sine values while
// they are still fp32 registers: one FMA per element, a fold of the two lane halves, one float per point and wave out.  The 32-wide
// launch that used to re-read the whole activation tensor for it (0.5 GB, HBM-bound) is gone.
// NDOT = 5: up to five projections per column tile (the final layers of the rgb / semantic / beta heads: each head block is one column tile
// of the fused first head layer and owns 3 / C / 1 output rows of the block-diagonal final matrix): nd_rows / nd_row0 per column tile.
template <int PL, int ACT, int AUX, bool COLSUM, bool SIGNS, int SINM, bool DIAG = false, int NDOT = 0>
__global__ __launch_bounds__(256, 2) void gemm_kc_kernel(const KcArgs) {
  // The arguments are read from the kernarg segment where they are needed (kargs(): a pointer the compiler must take as
  // new at every use site, so that it re-reads instead of keeping ~50 scalars alive across the k-loop and spilling them).
  const kargs_t p = kargs();
  __shared__ __attribute__((aligned(16))) char lds[NDOT > 0 ? KC_LDS_ND : KC_LDS];
  float* smax = reinterpret_cast<float*>(lds + KC_SMAX);
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wj0 = wave * 64;
  const int tiles_i = p->tiles_i, tiles_j = p->tiles_j, ntiles = tiles_i * tiles_j;
  constexpr int EB = 2 * PL;                                   // bytes per element of a plane tensor
  constexpr int KSUB = PL == 2 ? 16 : 32, KSH = PL == 2 ? 4 : 5;  // contraction depth of a sub-step
  const int nks16 = p->K >> 4;                                 // 16-deep steps of the weight pack
  const int nks = (p->K + KSUB - 1) >> KSH;                    // sub-steps (one plane: the last one may be half empty: K % 32 == 16)
  const int nks1 = p->Ka < p->K ? p->Ka >> KSH : nks;          // sub-steps of the first segment (Ka % (2 KSUB) == 0 if there are two)
  const int eW = *p->EW;                                      // the weight matrix's exponent (read once: a load inside the tile loop is awaited with
                                                              //  vmcnt(0), i.e. behind the next tile's operand requests)
  const int nst = (nks + 1) >> 1, nst1 = nks1 >> 1;          // stages; stages of the first segment (Ka % 32 == 0 if two)
  constexpr bool ONEPASS = ACT == ACT_SIN;                   // outputs in [-1, 1]: constant block exponent
  constexpr bool BIAS = AUX == AUX_NONE && !COLSUM;           // forward launches; the backward ones (column sums) have none
  const int dbg = DIAG ? p->dbg : 0;

  // ---- per-lane constants (the same for every tile of this workgroup) --------------------------------------------------
  // A: a stage is 32 k deep = 128 rows x 128 B (two column groups, the tensor's own byte order); its sixteen 1 KiB pieces
  // (8 rows each) go to the waves round-robin, four per wave, two per 16-deep sub-step.  The 16 B chunk c of a row sits at
  // position c ^ ((row >> 1) & 7): a quarter-wave of ds_read_b128 (eight lanes, eight consecutive rows, one chunk) then
  // covers four distinct positions twice -> all 32 banks in two passes, no conflict.  Rows beyond I are rejected by the
  // descriptor (its extent ends with row I - 1).
  auto a_lane_off = [&](int q, int ld, int l) -> unsigned {       // piece q of this wave: row, swizzled chunk
    const int row = 8 * (wave + 4 * q) + (l >> 3);
    return (unsigned)row * (unsigned)ld * (unsigned)EB + 16u * (unsigned)((l & 7) ^ ((row >> 1) & 7));
  };
  // Lane-derived values that only a tile's prologue / epilogue needs are recomputed there from an opaque copy of the lane
  // index: derived from `lane` itself they are loop-invariant, get hoisted out of the tile loop and then occupy registers
  // all through the k-loop.
  auto opaque = [](int v) { asm volatile("" : "+v"(v)); return v; };
  char* const dst0 = lds + wave * 1024;
  // A fragment addresses: lane -> (row l & 31, k half l >> 5); chunk (4 u + 2 pl + half) of sub-step u sits at position
  // chunk ^ ((row >> 1) & 7)
  const int rowl = lane & 31, kh = lane >> 5, swz = (rowl >> 1) & 7;
  unsigned fo[2][2];
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) fo[u][pl] = (unsigned)rowl * 128u + (unsigned)(((4 * u + 2 * pl + kh) ^ swz) << 4);
  // W: fragment-ordered pack; unit (ks, rb32) = 2 KiB = [plane][lane][16 B]; a wave reads units rb32 = u0, u0 + 1
  const srd_words srdW = make_srd_words(p->W, p->w_bytes);
  const unsigned w_ks0 = (unsigned)(p->w_k0 >> 4), w_rb32 = (unsigned)p->w_rb32;
  const bool two_seg = p->Ka < p->K;
  const unsigned voW = 16u * (unsigned)lane;

  // ---- state of the tile whose operands are being requested ------------------------------------------------------------
  int ti = 0, tj = 0, i0 = 0, j0 = 0, e_last = 0;
  unsigned w_u0 = 0;
  // This wave's 64 columns of that tile lie beyond J (the thin last column tile of a launch with J = 528: three of its four waves): its
  // weight loads are rejected through the scalar offset (before, they fetched the next k-step's units of other columns: a quarter of
  // the launch's weight fills for nothing).  Everything else stays as it is -- same requests, waits, barriers: guarding the MFMAs and
  // fragment reads too was tried (as branches inside the sub-step and as a second k-loop) and both made the kernel spill.
  bool w_idle = false;
  int eA = 0, eB = 0;              // exponent of k-step (lane) / (lane + 64)
  float bias_t = 0.f;              // the tile's bias, one column per thread
  // The descriptor, the lanes' offsets and the stage bias of the segment being requested are loop-carried and replaced
  // ONCE, at the stage where the second segment starts (selecting them per request cost ~16 scalar instructions per
  // piece, more than an MFMA gap hides); a stage beyond K is rejected through the scalar offset.
  srd_t srdCur;
  unsigned voCur[4];
  int sbias_st = 0;
  const int seg_switch = two_seg ? nst1 : 0x7fffffff;
  auto prepare = [&](int vb) {
    const kargs_t a = kargs();
    const int l = opaque(lane);
    tile_of_block(vb, tiles_i, tiles_j, ti, tj, a->rev != 0);
    if (a->tj_skip >= 0 && tj >= a->tj_skip) ++tj;        // (a column tile left out: KcArgs::tj_skip)
    i0 = ti * 128; j0 = tj * 256;
    const int lda = a->lda;
    const int i0a = (DIAG && (a->dbg & 8)) ? 0 : i0;
    srdCur = make_srd(a->A + ((size_t)i0a * lda + a->a_col0) * EB, (DIAG && (a->dbg & 1)) ? 0u : clamp_bytes(((unsigned long long)(a->I - i0a - 1) * lda + a->Ka) * (unsigned long long)EB));
    sbias_st = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) voCur[q] = a_lane_off(q, lda, l);
    w_u0 = (unsigned)((a->w_row0 + j0 + wj0) >> 5);
    w_idle = j0 + wj0 >= a->J;
    // exponents of the k-steps (lane) and (lane + 64) and of the last one: the fields of both segments as scalars, the choice per lane
    const int* EA1 = a->EA; const int* EA2 = a->EA2;
    const int ncb1 = ncb_of(lda), ncb2 = ncb_of(a->lda2), ac1 = a->a_col0, ac2 = a->a2_col0;
    auto exp_of = [&](int s) {
      const bool seg2 = s >= nks1;
      const int* E = seg2 ? EA2 : EA1;
      const int col = seg2 ? ac2 + KSUB * (s - nks1) : ac1 + KSUB * s;
      return E[(size_t)ti * (seg2 ? ncb2 : ncb1) + (col >> 7)];
    };
    eA = l < nks ? exp_of(l) : 0;
    eB = l + 64 < nks ? exp_of(l + 64) : 0;
    e_last = exp_of(nks - 1);
    if (BIAS) { const int tt = opaque(t); bias_t = (a->bias != nullptr && j0 + tt < a->J) ? a->bias[j0 + tt] : 0.f; }
  };
  auto enter_stage = [&](int S) {     // before the first piece of stage S
    if (__builtin_expect(S == seg_switch, 0)) {
      const kargs_t a = kargs();
      srdCur = make_srd(a->A2 + ((size_t)i0 * a->lda2 + a->a2_col0) * EB, clamp_bytes(((unsigned long long)(a->I - i0 - 1) * a->lda2 + (a->K - a->Ka)) * (unsigned long long)EB));
      sbias_st = nst1;
#pragma unroll
      for (int q = 0; q < 4; ++q) voCur[q] = a_lane_off(q, a->lda2, opaque(lane));
    }
  };
  auto issueA = [&](int S, int slot, int q) {
    dma16(srdCur, dst0 + slot * KC_A + 4096 * q, voCur[q], S < nst ? (unsigned)(S - sbias_st) * 128u : OOB);
  };
  // Stages 0 and 1 of the prepared tile, requested during the previous tile's epilogue -- as asm statements: hipcc cannot tell a
  // builtin LDS-DMA's destination from the epilogue's LDS strips and would drain it (vmcnt(0)) in front of the first strip
  // read-back, i.e. wait for the whole latency of the next tile's first stages once per tile.  The descriptor is rebuilt in its
  // four-word form from the same fields as `srdCur` (segment 1 before the switch stage, segment 2 from it on).
  auto cur_words = [&]() -> srd_words {
    const kargs_t a = kargs();
    if (sbias_st == 0) {
      const int i0a = (DIAG && (a->dbg & 8)) ? 0 : i0;
      return make_srd_words(a->A + ((size_t)i0a * a->lda + a->a_col0) * EB, (DIAG && (a->dbg & 1)) ? 0u : clamp_bytes(((unsigned long long)(a->I - i0a - 1) * a->lda + a->Ka) * (unsigned long long)EB));
    }
    return make_srd_words(a->A2 + ((size_t)i0 * a->lda2 + a->a2_col0) * EB, clamp_bytes(((unsigned long long)(a->I - i0 - 1) * a->lda2 + (a->K - a->Ka)) * (unsigned long long)EB));
  };
  const unsigned dst0_lds = (unsigned)__builtin_amdgcn_readfirstlane(lds_addr(dst0));
  auto headA = [&]() {
#pragma unroll
    for (int S = 0; S < 2; ++S) {
      enter_stage(S);
      const srd_words w = cur_words();
      const unsigned so = S < nst ? (unsigned)(S - sbias_st) * 128u : OOB;
#pragma unroll
      for (int q = 0; q < 4; ++q) dma16_asm(w, dst0_lds + (unsigned)(S * KC_A + 4096 * q), voCur[q], so);
    }
  };
  struct BFrag { u32x4 h[2], l[2]; };
  // The weight loads are asm statements with their completion counted by hand.  As builtins, hipcc's own vm-counter
  // bookkeeping put conservative waits behind them at the loop header (the two-step prefetch was undone) and re-used the
  // registers of the fragment set that is dead at the header as VALU temporaries, each with a WAW wait on a load.  A step
  // beyond K is rejected through the scalar offset.
  auto loadB2 = [&](int s, BFrag& b, int half) {     // two of the four weight loads of sub-step s
    // (s_nop 4: a scalar operand the compiler has just restored from a spill lane (v_readlane) needs five wait states before
    //  a vector-memory instruction reads it, and nothing inside an asm statement is padded for us)
    if constexpr (PL == 2) {     // unit (16-k step s, 32-row block) = 2 KiB [hi | lo]; half: the wave's first / second 32-row block
      const unsigned so = (s < nks && !w_idle) ? ((w_ks0 + (unsigned)s) * w_rb32 + w_u0) * 2048u : OOB;
      if (half == 0)
        asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %2, %3, %4 offen\n\tbuffer_load_dwordx4 %1, %2, %3, %4 offen offset:1024"
                     : "=&v"(b.h[0]), "=&v"(b.l[0]) : "v"(voW), "s"(srdW), "s"(so));
      else
        asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %2, %3, %4 offen offset:2048\n\tbuffer_load_dwordx4 %1, %2, %3, %4 offen offset:3072"
                     : "=&v"(b.h[1]), "=&v"(b.l[1]) : "v"(voW), "s"(srdW), "s"(so));
    } else {                     // units of 1 KiB; half: the first / second 16-k step of the 32-deep sub-step (h / l registers), both 32-row blocks
      const int s16 = 2 * s + half;
      const unsigned so = (s16 < nks16 && !w_idle) ? ((w_ks0 + (unsigned)s16) * w_rb32 + w_u0) * 1024u : OOB;
      if (half == 0)
        asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %2, %3, %4 offen\n\tbuffer_load_dwordx4 %1, %2, %3, %4 offen offset:1024"
                     : "=&v"(b.h[0]), "=&v"(b.h[1]) : "v"(voW), "s"(srdW), "s"(so));
      else
        asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %2, %3, %4 offen\n\tbuffer_load_dwordx4 %1, %2, %3, %4 offen offset:1024"
                     : "=&v"(b.l[0]), "=&v"(b.l[1]) : "v"(voW), "s"(srdW), "s"(so));
    }
  };
  // everything but the six youngest requests (= the previous sub-step's) has landed; names the fragments so that no use of
  // them can be scheduled above the wait
  auto wait_b = [&](BFrag& b) {
    asm volatile("s_waitcnt vmcnt(6)" : "+v"(b.h[0]), "+v"(b.l[0]), "+v"(b.h[1]), "+v"(b.l[1])::"memory");
  };
  auto pin_b = [&](BFrag& b) { asm volatile("" : "+v"(b.h[0]), "+v"(b.l[0]), "+v"(b.h[1]), "+v"(b.l[1])::"memory"); };
  BFrag bq0, bq1, bq2;
  auto headW = [&]() { loadB2(0, bq0, 0); loadB2(0, bq0, 1); loadB2(1, bq1, 0); loadB2(1, bq1, 1); };

  // ---- the workgroup walks its tiles; the operands of tile n + 1 (stages 0, 1 of A into ring slots 0, 1; the weight
  //      fragments of sub-steps 0, 1) are requested during the epilogue of tile n ----------------------------------------------
  // Tiles: the first one by block index, the following ones drawn from a counter per XCD group (blocks b and b + 8 share
  // an XCD under round-robin placement; the tile map keeps the tiles of one group contiguous, tile_of_block).  The
  // workgroups of a CU do not run at the same pace (co-residency, clocks): a fixed share per workgroup ends with the
  // slowest one working alone (measured 671 k cycles against a median of 575 k for 8 tiles each).
  int* const tile_ctr = p->tile_ctr;
  const bool dyn = tile_ctr != nullptr;
  const int n_grp = (gridDim.x & 7) == 0 ? 8 : 1, grp = (int)blockIdx.x & (n_grp - 1);
  const unsigned ctr_off = 4u * (unsigned)grp;
  if constexpr (NDOT > 0) {   // the projection vectors, once per workgroup (published by the first tile's barrier, awaited by its vmcnt(0)):
    // column tile tj owns rows nd_row0[tj] .. + nd_rows[tj] of nd_w (leading dimension nd_ldw), its 256 columns of them, at nd_woff[tj]
    const float* ndw = p->nd_w;
    float* snd0 = reinterpret_cast<float*>(lds + KC_NDW);
    for (int tjx = 0; tjx < p->tiles_jr; ++tjx) {
      const int nr = p->nd_rows[tjx], r0 = p->nd_row0[tjx], wo = p->nd_woff[tjx], ldw = p->nd_ldw;
      for (int o = 0; o < nr; ++o) snd0[wo + 256 * o + t] = (256 * tjx + t < p->J) ? ndw[(size_t)(r0 + o) * ldw + 256 * tjx + t] : 0.f;
    }
  }
  int vb = blockIdx.x;
  prepare(vb);
  headW();
  headA();
  bool first = true;
  for (int it = 0;; ++it) {
    float* sbias = reinterpret_cast<float*>(lds + KC_BIAS + (it & 1) * 1024);
    int* etab = reinterpret_cast<int*>(lds + KC_ETAB + (it & 1) * 512);
    // exponents along k: table in LDS + a bit per step where the scale changes (128 steps at most: K <= 2048)
    int eAp = __builtin_amdgcn_update_dpp(eA, eA, 0x138, 0xF, 0xF, false);       // wave_shr 1: the previous step's exponent
    int eBp = __builtin_amdgcn_update_dpp(eB, eB, 0x138, 0xF, 0xF, false);
    const int eA63 = __builtin_amdgcn_readlane(eA, 63);
    if (opaque(lane) == 0) eBp = eA63;
    if (lane + 64 >= nks) eBp = eB;
    if (lane >= nks) eAp = eA;
    if (wave == 0) { etab[lane] = eA; etab[lane + 64] = eB; }
    if (BIAS) sbias[t] = bias_t * (ACT == ACT_SIN ? kargs()->w0 * INV_2PI : 1.f);
    const unsigned long long chg0 = __builtin_amdgcn_ballot_w64(eA != eAp), chg1 = __builtin_amdgcn_ballot_w64(eB != eBp);

    f32x16 acc[4][2];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int nj = 0; nj < 2; ++nj)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mi][nj][r] = 0.f;

    // One 16-deep sub-step s = 2 S + u of stage S: 24 MFMAs on the fragments `fa` (read from LDS during the PREVIOUS sub-step)
    // and the weight registers `bc`; `bn` receives the weight fragments of sub-step s + 2.  A single wave issues one
    // instruction per ~4 cycles, an MFMA occupies the matrix pipe for 32: whatever is issued in a block of its own leaves
    // the pipe idle, so every non-MFMA instruction sits in a gap between MFMAs: per 32-point block mi the six MFMAs run
    // hi*lo, lo*hi, hi*hi on two accumulators each, the lo fragment of mi is re-read for sub-step s + 1 as soon as its last
    // MFMA has been issued, the hi fragment after the block, and the four weight loads / two DMA pieces are spread over the
    // blocks.  The workgroup barrier of a new stage comes in the middle of the last sub-step of its predecessor (before the
    // first read of the new stage): every wave has passed the top-of-sub-step wait that covers its own pieces of stage S + 1
    // (issued during stage S - 1) and has finished reading stage S - 1, whose slot the requests of stage S + 2 (issued
    // after that point in program order) overwrite.
    // vm-counter order inside a tile: per sub-step s = 2 S + u: [W(s + 2) x 4] [two pieces of A(S + 2)] -- at the top of
    // sub-step s >= 2 the six requests of sub-step s - 1 may be outstanding and everything older has landed, which covers
    // W(s) and all of stage S: s_waitcnt vmcnt(6).  Sub-steps 0 and 1 run on what the tile-start wait below has retired.
    struct AFrag { f16x8 h[4], l[4]; };
    AFrag fa;
    auto step = [&](int s, int slot, int u, BFrag& bc, BFrag& bn) {
      if (s >= 2) wait_b(bc);
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
        if constexpr (PL == 2) {
          acc[mi][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl0, fa.h[mi], acc[mi][0], 0, 0, 0);
          acc[mi][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl1, fa.h[mi], acc[mi][1], 0, 0, 0);
        } else {   // one plane: fa.h / the h registers = the first 16 k of the sub-step, fa.l / the l registers = the second
          acc[mi][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh0, fa.h[mi], acc[mi][0], 0, 0, 0);
          acc[mi][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh1, fa.h[mi], acc[mi][1], 0, 0, 0);
        }
        if (mi < 2) loadB2(s + 2, bn, mi);
        if (mi == 2 && u == 0) enter_stage((s >> 1) + 2);
        if (mi >= 2) issueA((s >> 1) + 2, (slot + 2) % KC_RING, 2 * u + (mi - 2));   // two of the four pieces of stage S + 2
        if constexpr (PL == 2) {
          acc[mi][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh0, fa.l[mi], acc[mi][0], 0, 0, 0);
          acc[mi][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh1, fa.l[mi], acc[mi][1], 0, 0, 0);
        }
        if (mi == 0 && u == 1 && !(DIAG && (dbg & 32))) { __builtin_amdgcn_sched_barrier(0); barrier_raw(); __builtin_amdgcn_sched_barrier(0); }
        const bool rd = !(DIAG && (dbg & 64));
        if constexpr (PL == 2) {
          const f16x8 nl = rd ? ldsfrag(sn + 4096 * mi + fo[u ^ 1][1]) : fa.l[mi];
          acc[mi][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh0, fa.h[mi], acc[mi][0], 0, 0, 0);
          acc[mi][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh1, fa.h[mi], acc[mi][1], 0, 0, 0);
          fa.l[mi] = nl;
          if (rd) fa.h[mi] = ldsfrag(sn + 4096 * mi + fo[u ^ 1][0]);
        } else {
          const f16x8 nh = rd ? ldsfrag(sn + 4096 * mi + fo[u ^ 1][0]) : fa.h[mi];   // fa.h[mi] has issued its last MFMA
          acc[mi][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl0, fa.l[mi], acc[mi][0], 0, 0, 0);
          acc[mi][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl1, fa.l[mi], acc[mi][1], 0, 0, 0);
          fa.h[mi] = nh;
          if (rd) fa.l[mi] = ldsfrag(sn + 4096 * mi + fo[u ^ 1][1]);
        }
        __builtin_amdgcn_sched_barrier(0);   // keep the blocks apart: left alone, the scheduler gathers the reads at the end
      }
    };
    // Tile start: W(0), W(1) and this wave's pieces of stages 0 AND 1 are home (sub-steps 0 and 1 have no wait of their own,
    // and the barrier inside sub-step 1 publishes stage 1).  First tile: they are the only requests.  Later tiles: they were
    // requested during the previous epilogue, which has issued at least 32 (one plane: 16) stores since.  The barrier also ends every wave's
    // use of the epilogue's LDS regions before the requests of stage 2 go out.
    if (first) wait_vm<0>(); else wait_vm<(PL == 2 ? 32 : 16)>();   // (one plane: 4 blocks x 4 stores per tile)
    pin_b(bq0); pin_b(bq1);
    barrier_raw();
    // The tile after this one: drawn from the counter of the workgroup's XCD group while this tile's k-loop runs (a returning
    // atomic of ONE lane; it is this wave's oldest request by the time sub-step 2 waits, and read after the loop's drain).
    int drawn = 0;
    if (dyn && wave == 0 && opaque(lane) == 0)
      asm volatile("s_nop 4\n\tglobal_atomic_add %0, %1, %2, %3 sc0" : "=v"(drawn) : "v"(ctr_off), "v"(1), "s"(tile_ctr) : "memory");
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
    wait_vm<0>();        // rejected requests behind the last stage write zeros into the ring: drain before re-using it
    if (dyn && wave == 0 && opaque(lane) == 0) {
      asm volatile("" : "+v"(drawn));
      *reinterpret_cast<volatile int*>(lds + KC_NEXT) = (int)gridDim.x + n_grp * drawn + grp;
    }
    barrier_raw();

    // ---- this tile's coordinates for the epilogue; then the next tile's operands are requested ----------------------------
    const int c_ti = ti, c_i0 = i0, c_j0 = j0, c_elast = e_last;
    // the stored activation's block exponent (derivative epilogues): loaded BEFORE the next tile's operands are requested -- the
    // compiler awaits it with vmcnt(0), which behind those requests costs their whole latency once per tile
    int eH = 0;
    if (AUX != AUX_NONE && c_j0 + wj0 < kargs()->J) { const kargs_t a = kargs(); eH = a->EH[(size_t)c_ti * ncb_of(a->ldh) + ((a->h_col0 + c_j0 + wj0) >> 7)]; eH = __builtin_amdgcn_readfirstlane(eH); }
    // (read on every path: this LDS read is also where hipcc settles its account of the k-loop's builtin DMA requests -- before
    //  the next tile's requests go out, not in the middle of the epilogue)
    const int nxt_lds = __builtin_amdgcn_readfirstlane(*reinterpret_cast<volatile int*>(lds + KC_NEXT));
    const int vbn = dyn ? nxt_lds : vb + (int)gridDim.x;
    const bool more = vbn < ntiles;
    // The next tile's operands.  Forward epilogues request them now; the derivative epilogues first request what they need
    // themselves at once (sign words, the first two stored-activation half-blocks) -- the counter is in order, and a request
    // queued behind the next tile's first stages would wait for those.
    auto next_heads = [&]() {
      if (more) {
        prepare(vbn);
        if (ONEPASS) headW();
        headA();
      }
    };
    if constexpr (AUX == AUX_NONE) next_heads();

    // ---- epilogue.  Lane l: point pt = l & 31 of each 32-point block mi; register r of block (mi, nj) is column
    //      64 wave + 32 nj + 16 (r >> 3) + 8 (l >> 5) + (r & 7) of the tile.  Ring slots 0 and 1 are being refilled; slot 2
    //      and the region behind the ring hold the waves' plane strips (results on their way out) and, for the derivative
    //      epilogues, the stored activations on their way in.
    const kargs_t e = kargs();
    const int e_in = c_elast + eW;              // acc = true value * 2^e_in
    const bool e_small = e_in >= -120 && e_in <= 120;
    if (!e_small) {   // exponents beyond a single fp32 factor (never with sane data): scale the accumulators first
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int nj = 0; nj < 2; ++nj) acc[mi][nj] = scale_acc(acc[mi][nj], -e_in);
    }
    const float inv_in = e_small ? pow2f(-e_in) : 1.f;
    const int el = opaque(lane);               // see `opaque` above
    const int pt = el & 31, lh = el >> 5;
    const int nrows = min(128, e->I - c_i0);                      // > 0: the grid covers ceil(I / 128) row tiles
    const int jw = c_j0 + wj0;                                   // first column of the wave
    const bool wave_cols = jw < e->J;
    const size_t offC = uniform_sz(((size_t)c_i0 * e->ldc + e->c_col0) * EB);
    const srd_t srdC = make_srd(e->C + offC, (DIAG && (e->dbg & 4)) ? 0u : clamp_bytes(((unsigned long long)(nrows - 1) * e->ldc + e->J) * (unsigned long long)EB));
    // Plane strip of one (32-point block, 32-column half) = [32 points][128 B]: the two 64-byte groups [hi | lo] of the
    // half as they lie in memory, chunk c at position c ^ (point & 7).  A el writes its 16-byte pieces (eight consecutive
    // lanes: eight positions = all 32 banks), then the wave reads the strip back eight whole rows per instruction and stores
    // 8 x 128 contiguous bytes.
    char* const strip = lds + KC_XREG + wave * 4096;
    const unsigned sw_off = (unsigned)pt * 128u;
    const int srow = el >> 3, schunk = el & 7;
    const unsigned sr_off = (unsigned)srow * 128u + 16u * (unsigned)(schunk ^ srow);       // + 1024 per pass (8 rows: same swizzle)
    // One plane: a strip row (128 B) holds all 64 columns of the wave for one point -- chunk 4 nj + 2 gg + lh = eight columns --
    // and is flushed once per 32-point block instead of once per (block, 32-column half).
    unsigned voC[2];                                                                       // per 32-column half (J % 16 == 0); one plane: [0] only
#pragma unroll
    for (int nj = 0; nj < 2; ++nj)
      voC[nj] = PL == 2 ? (jw + 32 * nj + 16 * (schunk >> 2) < e->J ? (unsigned)srow * (unsigned)e->ldc * 4u + (unsigned)((jw >> 4) + 2 * nj) * 64u + 16u * (unsigned)schunk : OOBH)
                        : (jw + 8 * schunk < e->J ? (unsigned)srow * (unsigned)e->ldc * 2u + (unsigned)jw * 2u + 16u * (unsigned)schunk : OOBH);
    const unsigned stepC8 = 8u * (unsigned)e->ldc * (unsigned)EB;
    auto strip_put = [&](int gg, const u32x4& hi, const u32x4& lo) {
      *reinterpret_cast<u32x4*>(strip + sw_off + 16 * ((4 * gg + lh) ^ (pt & 7))) = hi;
      *reinterpret_cast<u32x4*>(strip + sw_off + 16 * ((4 * gg + 2 + lh) ^ (pt & 7))) = lo;
    };
    auto strip_put1 = [&](int nj, int gg, const u32x4& hi) {
      *reinterpret_cast<u32x4*>(strip + sw_off + 16 * ((4 * nj + 2 * gg + lh) ^ (pt & 7))) = hi;
    };
    auto keep_planes1 = [](const u32x4 (&hi)[4]) { asm volatile("" ::"v"(hi[0]), "v"(hi[1]), "v"(hi[2]), "v"(hi[3])); };
    // HAZARD (measured on gfx950, not modelled by hipcc): a ds_write_b128 can fetch its data registers AFTER a younger
    // ds_read_b128 of the same wave has returned into them.  The compiler, free to do so, gave the read-back of the strip the
    // registers of the planes it had just written; with the LDS busy (co-resident workgroup) single dwords of the written
    // planes then held the read-back's data -- wrong values in ~1 % of the rows, different from run to run.  The planes of a
    // half-block are therefore kept alive (keep_planes) until the read-back has been consumed.
    auto keep_planes = [](const u32x4 (&hi)[2], const u32x4 (&lo)[2]) { asm volatile("" ::"v"(hi[0]), "v"(lo[0]), "v"(hi[1]), "v"(lo[1])); };
    auto strip_flush = [&](int mi, int nj) {     // rows beyond I are rejected by the descriptor
      // two read-backs in flight, then their two stores: two LDS latencies per flush instead of four in a row (all four at once
      // cost 16 registers the first flushes of a tile do not have: four instantiations spilled)
#pragma unroll
      for (int pp = 0; pp < 4; pp += 2) {
        const u32x4 d0 = *reinterpret_cast<const u32x4*>(strip + sr_off + 1024 * pp);
        const u32x4 d1 = *reinterpret_cast<const u32x4*>(strip + sr_off + 1024 * (pp + 1));
        __builtin_amdgcn_raw_buffer_store_b128(d0, srdC, voC[nj], (unsigned)(4 * mi + pp) * stepC8, 2);
        store_data_guard(d0);
        __builtin_amdgcn_raw_buffer_store_b128(d1, srdC, voC[nj], (unsigned)(4 * mi + pp + 1) * stepC8, 2);
        store_data_guard(d1);
      }
    };
    float bj[4][8];
    if (BIAS && NDOT == 0) {     // (NDOT: re-read per use -- the dot product's registers take the place of the 32 bias registers)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const float4 b0 = *reinterpret_cast<const float4*>(&sbias[wj0 + 16 * gq + 8 * lh]);
        const float4 b1 = *reinterpret_cast<const float4*>(&sbias[wj0 + 16 * gq + 8 * lh + 4]);
        bj[gq][0] = b0.x; bj[gq][1] = b0.y; bj[gq][2] = b0.z; bj[gq][3] = b0.w;
        bj[gq][4] = b1.x; bj[gq][5] = b1.y; bj[gq][6] = b1.z; bj[gq][7] = b1.w;
      }
    }

    if (DIAG && (dbg & 16)) {
      // (diagnostic: no epilogue work at all; the accumulators are consumed so that the k-loop stays)
      float keep = 0.f;
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int nj = 0; nj < 2; ++nj) keep += acc[mi][nj][0];
      if (keep == 12345.678f) smax[wave] = keep;
      if constexpr (AUX != AUX_NONE) next_heads();
      if (!ONEPASS && more) headW();
    } else if constexpr (ONEPASS) {
      // ---- sine: one pass.  x = acc * (2^-e w0 / 2 pi) + b w0 / 2 pi in revolutions (bias row staged in LDS, already scaled)
      const float su = inv_in * e->w0 * INV_2PI;
      const int c_tj = c_j0 >> 8;
      const int nd_n = NDOT > 0 ? e->nd_rows[c_tj] : 0;      // projections of this column tile (uniform)
      const float* snd = reinterpret_cast<const float*>(lds + KC_NDW) + (NDOT > 0 ? e->nd_woff[c_tj] : 0) + wj0 + 8 * lh;
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
        unsigned sw = 0u;
        float nd[NDOT > 0 ? NDOT : 1];
#pragma unroll
        for (int o = 0; o < (NDOT > 0 ? NDOT : 1); ++o) nd[o] = 0.f;
        u32x4 ph1[4];
#pragma unroll
        for (int nj = 0; nj < 2; ++nj) {
          u32x4 phi[2], plo[2];
#pragma unroll
          for (int gg = 0; gg < 2; ++gg) {
            const int gq = 2 * nj + gg;
            float v[8];
            if constexpr (NDOT > 0) {
              const float4 b0 = *reinterpret_cast<const float4*>(&sbias[wj0 + 16 * gq + 8 * lh]);
              const float4 b1 = *reinterpret_cast<const float4*>(&sbias[wj0 + 16 * gq + 8 * lh + 4]);
              v[0] = fmaf(acc[mi][nj][8 * gg + 0], su, b0.x); v[1] = fmaf(acc[mi][nj][8 * gg + 1], su, b0.y);
              v[2] = fmaf(acc[mi][nj][8 * gg + 2], su, b0.z); v[3] = fmaf(acc[mi][nj][8 * gg + 3], su, b0.w);
              v[4] = fmaf(acc[mi][nj][8 * gg + 4], su, b1.x); v[5] = fmaf(acc[mi][nj][8 * gg + 5], su, b1.y);
              v[6] = fmaf(acc[mi][nj][8 * gg + 6], su, b1.z); v[7] = fmaf(acc[mi][nj][8 * gg + 7], su, b1.w);
            } else {
#pragma unroll
              for (int c = 0; c < 8; ++c) v[c] = fmaf(acc[mi][nj][8 * gg + c], su, bj[gq][c]);
            }
            sin2pi8<SIGNS, SINM>(v, sw);
            if constexpr (NDOT > 0) {
#pragma unroll
              for (int o = 0; o < NDOT; ++o)
                if (NDOT == 1 || o < nd_n) {
                  const float4 w0v = *reinterpret_cast<const float4*>(snd + 256 * o + 16 * gq);
                  const float4 w1v = *reinterpret_cast<const float4*>(snd + 256 * o + 16 * gq + 4);
                  nd[o] = fmaf(v[0], w0v.x, nd[o]); nd[o] = fmaf(v[1], w0v.y, nd[o]); nd[o] = fmaf(v[2], w0v.z, nd[o]); nd[o] = fmaf(v[3], w0v.w, nd[o]);
                  nd[o] = fmaf(v[4], w1v.x, nd[o]); nd[o] = fmaf(v[5], w1v.y, nd[o]); nd[o] = fmaf(v[6], w1v.z, nd[o]); nd[o] = fmaf(v[7], w1v.w, nd[o]);
                }
            }
            if constexpr (PL == 2) {
              split8(v, 8192.f, phi[gg], plo[gg]);
              strip_put(gg, phi[gg], plo[gg]);
            } else {
              cvt8(v, 8192.f, ph1[gq]);
              strip_put1(nj, gg, ph1[gq]);
            }
          }
          if constexpr (PL == 2) {
            strip_flush(mi, nj);
            keep_planes(phi, plo);
          }
        }
        if constexpr (PL == 1) {
          strip_flush(mi, 0);
          keep_planes1(ph1);
        }
        if constexpr (NDOT > 0) {   // the two lane halves hold the two column halves of every 16-column group of the same point
#pragma unroll
          for (int o = 0; o < NDOT; ++o)
            if (NDOT == 1 || o < nd_n) {
              const float tot = nd[o] + __shfl_xor(nd[o], 32, 64);
              if (lh == 0 && 32 * mi + pt < nrows)
                e->nd_out[(size_t)((c_tj * 4 + wave) * NDOT + o) * e->nd_stride + (size_t)(c_i0 + 32 * mi + pt)] = tot;
            }
        }
        if (SIGNS && e->Csign != nullptr && wave_cols && 32 * mi < nrows)
          e->Csign[((size_t)((c_i0 >> 5) + mi) * ((e->ldc + 63) >> 6) + ((e->c_col0 + jw) >> 6)) * 64 + el] = sw;
      }
      if ((wave & 1) == 0 && el == 0 && wave_cols) e->EC[(size_t)c_ti * ncb_of(e->ldc) + ((e->c_col0 + jw) >> 7)] = 13;
    } else {
      // ---- two passes.  Pass A: final values in place of the accumulators, their |max|, column sums -- one 32-column half
      //      of the wave (nj) after the other, so that only 16 column sums are alive at a time.
      float wmax = 0.f;
      const float inv_h = pow2f(-eH);
      // Stored activations (derivative epilogues): half-block hb = (nj, mi) of the wave = 32 points x 128 B, fetched by LDS-DMA
      // in four 1 KiB pieces (8 whole half-rows each) into one of two 4 KiB buffers of the wave (ring slot 2 / the region
      // behind the ring), half-block hb + 1 while hb is worked on.  Chunk c of point row q lies at position c ^ ((q >> 1) & 7)
      // (swizzle on the source address): the lanes of a ds_read_b128 group hold 16 points that differ in q & 15 -> 16
      // different 16-byte slots of the 256-byte bank row.  The sign words (one dword per lane and block mi) come the same
      // way, all four ahead of the first half-block.
      // One plane: a half-block is 32 points x 64 B = 2 KiB in two pieces of 16 point rows; chunk c of row q at position
      // c ^ ((q >> 2) & 3) (the 16 points of a ds_read_b128 group then cover the sixteen 16-byte slots of a 256-byte bank row).
      const size_t offH = uniform_sz(AUX != AUX_NONE ? ((size_t)c_i0 * e->ldh + e->h_col0) * EB : 0);
      const srd_words srdH = make_srd_words(AUX != AUX_NONE ? e->H + offH : nullptr,
                                            AUX != AUX_NONE && !(DIAG && (dbg & 128)) ? clamp_bytes(((unsigned long long)(nrows - 1) * e->ldh + e->J) * (unsigned long long)EB) : 0u);
      constexpr int HPC = PL == 2 ? 4 : 2;                        // 1 KiB pieces per half-block
      const unsigned hbuf[2] = {(unsigned)__builtin_amdgcn_readfirstlane(lds_addr(lds + 2 * KC_A + wave * 4096)),
                                (unsigned)__builtin_amdgcn_readfirstlane(lds_addr(lds + KC_XREG + wave * 4096))};
      const int hq = PL == 2 ? el >> 3 : el >> 2;                 // point row inside a piece
      auto dma_h = [&](int hb) {
        if (AUX == AUX_NONE) return;
        const int nj = hb >> 2, mi = hb & 3;
#pragma unroll
        for (int pc = 0; pc < HPC; ++pc) {
          if constexpr (PL == 2) {
            const int q = 8 * pc + hq;                              // point row inside the block
            const int c = (el & 7) ^ ((q >> 1) & 7);                // chunk of the half-row this lane fetches
            const bool ok = jw + 32 * nj + 16 * (c >> 2) < e->J;
            const unsigned vo = ok ? (unsigned)q * (unsigned)e->ldh * 4u + (unsigned)((jw >> 4) + 2 * nj) * 64u + 16u * (unsigned)c : OOBH;
            dma16_asm_nt(srdH, hbuf[hb & 1] + (unsigned)(pc * 1024), vo, (unsigned)(32 * mi) * (unsigned)e->ldh * 4u);
          } else {
            const int q = 16 * pc + hq;
            const int c = (el & 3) ^ ((q >> 2) & 3);                // eight columns each
            const bool ok = jw + 32 * nj + 8 * c < e->J;
            const unsigned vo = ok ? (unsigned)q * (unsigned)e->ldh * 2u + (unsigned)(jw + 32 * nj) * 2u + 16u * (unsigned)c : OOBH;
            dma16_asm_nt(srdH, hbuf[hb & 1] + (unsigned)(pc * 1024), vo, (unsigned)(32 * mi) * (unsigned)e->ldh * 2u);
          }
        }
      };
      if (AUX == AUX_SINREC) {
        const srd_words srdS = make_srd_words(e->Hsign, clamp_bytes(sign_words((size_t)e->I, e->ldh) * 4ull));
        const unsigned sreg0 = __builtin_amdgcn_readfirstlane(lds_addr(lds + KC_HSIGN + wave * 1024));
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
          const bool ok = wave_cols && 32 * mi < nrows;
          const unsigned vo = ok ? (unsigned)((((size_t)((c_i0 >> 5) + mi) * ((e->ldh + 63) >> 6) + ((e->h_col0 + jw) >> 6)) * 64 + el) * 4) : OOBH;
          dma4_asm(srdS, sreg0 + (unsigned)(mi * 256), vo, 0u);
        }
      }
      dma_h(0);
      dma_h(1);
      if constexpr (AUX != AUX_NONE) next_heads();   // (its 8 A pieces are younger than half-blocks 0, 1 and older than the later ones)
      // Derivative epilogues work on the RAW accumulators: every factor that is uniform over the wave's tile -- the accumulators'
      // scale 2^-e_in, |w0|, the 2^-eH of the rebuilt cosine -- is one number `fac` that multiplies the column sums and the block
      // maximum once and rides in the scale of the plane split; the sign bits are xor-ed with w0's own sign once per word.
      // (Rows beyond I need no mask: their A rows were rejected, their accumulators are exactly 0 and every factor is finite.)
      const float fac = AUX == AUX_SINREC ? inv_in * fabsf(e->w0) * inv_h : inv_in;
      const int eHc = eH < -60 ? -60 : (eH > 60 ? 60 : eH);
      const float one_s = pow2f(2 * eHc);                          // 1.0 in the units of s^2 (s = h 2^eH; eH = 13 for a SIREN layer's output)
      const unsigned sflip = e->w0 < 0.f ? 0xFFFFFFFFu : 0u;
      unsigned sword[4] = {0u, 0u, 0u, 0u};
#pragma unroll
      for (int nj = 0; nj < 2; ++nj) {
        float cs[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) cs[r] = 0.f;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
          const int hb = 4 * nj + mi;
          u32x4 hh[2], hl[2];
          if (AUX != AUX_NONE) {
            // half-block hb (and everything older: the sign words) is home when all but the four pieces of hb + 1 are
            // ... and, for half-blocks 0 and 1, the 8 pieces of the next tile's first two A stages requested behind them
            if (hb == 7) wait_vm<0>(); else if (hb < 2 && more) wait_vm<HPC + 8>(); else wait_vm<HPC>();
            const char* hreg = (hb & 1) ? lds + KC_XREG + wave * 4096 : lds + 2 * KC_A + wave * 4096;
            if (AUX == AUX_SINREC && nj == 0) sword[mi] = *reinterpret_cast<const unsigned*>(lds + KC_HSIGN + wave * 1024 + mi * 256 + el * 4) ^ sflip;
#pragma unroll
            for (int gg = 0; gg < 2; ++gg) {
              if constexpr (PL == 2) {
                hh[gg] = *reinterpret_cast<const u32x4*>(hreg + pt * 128 + 16 * ((4 * gg + lh) ^ ((pt >> 1) & 7)));
                if (AUX == AUX_SINREC) hl[gg] = *reinterpret_cast<const u32x4*>(hreg + pt * 128 + 16 * ((4 * gg + 2 + lh) ^ ((pt >> 1) & 7)));
              } else {
                hh[gg] = *reinterpret_cast<const u32x4*>(hreg + pt * 64 + 16 * ((2 * gg + lh) ^ ((pt >> 2) & 3)));
              }
            }
            if (hb + 2 < 8) {   // the buffer is free once these reads have returned
              if constexpr (PL == 2 && AUX == AUX_SINREC) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(hh[0]), "+v"(hh[1]), "+v"(hl[0]), "+v"(hl[1]), "+v"(sword[mi])::"memory");
              else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(hh[0]), "+v"(hh[1]), "+v"(sword[mi])::"memory");
              dma_h(hb + 2);
            }
          }
#pragma unroll
          for (int gg = 0; gg < 2; ++gg) {
            const int gq = 2 * nj + gg;
            float v[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) {
              const float x = acc[mi][nj][8 * gg + c];
              v[c] = AUX != AUX_NONE || COLSUM ? x : (BIAS ? fmaf(x, inv_in, bj[gq][c]) : x * inv_in);
            }
            if (ACT == ACT_RELU) {
#pragma unroll
              for (int c = 0; c < 8; ++c) v[c] = fmaxf(v[c], 0.f);
            }
            if (AUX == AUX_SINREC) {
              // w0 cos(w0 z) = +-|w0| sqrt(1 - h^2), from the planes as they lie: s = hi + lo = h 2^eH (one mixed-precision FMA,
              // exact), 2^2eH - s^2 (one FMA; >= 0 for |h| <= 1, |.| on the root's operand covers a stray ulp), the root, and the
              // sign bit shifted to bit 31 and merged over the root by one v_bfi
              float sv[8];
              if constexpr (PL == 2) sum8(hh[gg], hl[gg], sv); else cvt8f(hh[gg], sv);
#pragma unroll
              for (int c = 0; c < 8; ++c) {
                const float om = fmaf(-sv[c], sv[c], one_s);
                const float root = __builtin_amdgcn_sqrtf(fabsf(om));
                // (plain C, not an asm v_bfi: the consumer of a transcendental's result needs a wait state the compiler only inserts
                //  for instructions it can see -- an asm statement here read stale roots in half of the rows)
                const unsigned rs = __float_as_uint(root) | ((sword[mi] << (31 - (16 * nj + 8 * gg + c))) & 0x80000000u);   // root >= 0: one v_and_or
                v[c] *= __uint_as_float(rs);
              }
            } else if (AUX == AUX_RELU_MASK) {
              // h > 0 <=> its hi plane > 0 (hi = 0 only where the value rounds to 0 on the plane's grid, and lo is 0 with it)
              float hv[8];
              cvt8f(hh[gg], hv);
#pragma unroll
              for (int c = 0; c < 8; ++c) v[c] = hv[c] > 0.f ? v[c] : 0.f;
            }
#pragma unroll
            for (int c = 0; c < 8; ++c) {
              if (COLSUM) cs[8 * gg + c] += v[c];
              acc[mi][nj][8 * gg + c] = v[c];
            }
            if (jw + 16 * gq < e->J) {
              const float m8 = absmax3(v[6], v[7], absmax3(v[4], v[5], absmax3(v[2], v[3], absmax3(v[0], v[1], 0.f))));
              wmax = fmaxf(wmax, (BIAS && (32 * mi + pt) >= nrows) ? 0.f : m8);   // (forward launches add the bias to rows beyond I too: out of the maximum)
            }
          }
        }
        if (COLSUM && e->colsum != nullptr) {   // one partial row per 128-point tile
#pragma unroll
          for (int r = 0; r < 16; ++r) cs[r] = sum32(cs[r]);
          if ((el & 31) == 31) {
#pragma unroll
            for (int gg = 0; gg < 2; ++gg)
              if (jw + 16 * (2 * nj + gg) < e->J) {
                float* d = e->colsum + (size_t)c_ti * e->ldcs + jw + 16 * (2 * nj + gg) + 8 * lh;
                *reinterpret_cast<float4*>(d) = make_float4(cs[8 * gg] * fac, cs[8 * gg + 1] * fac, cs[8 * gg + 2] * fac, cs[8 * gg + 3] * fac);
                *reinterpret_cast<float4*>(d + 4) = make_float4(cs[8 * gg + 4] * fac, cs[8 * gg + 5] * fac, cs[8 * gg + 6] * fac, cs[8 * gg + 7] * fac);
              }
          }
        }
      }
      // block maximum: waves 2 c and 2 c + 1 share the exponent block (ti, column block c of the tile)
      constexpr bool RAW = AUX != AUX_NONE || COLSUM;          // the accumulators hold raw values: true value = raw * fac
      wmax = wave_max(wmax) * (RAW ? fac : 1.f);
      if (el == 0) smax[wave] = wmax;
      barrier_raw();     // (not __syncthreads(): that drains the vector-memory counter too, behind the next tile's requests and this tile's
                         //  stores) also: every wave has finished with the stored-activation buffers the strips share
      const float bmax = fmaxf(smax[wave & 2], smax[(wave & 2) + 1]);
      const int eC = exp_of_maxbits(__float_as_uint(bmax));
      const float sc = pow2f(eC) * (RAW ? fac : 1.f);
      if ((wave & 1) == 0 && el == 0 && wave_cols) e->EC[(size_t)c_ti * ncb_of(e->ldc) + ((e->c_col0 + jw) >> 7)] = eC;
      if (more) headW();     // the registers of pass A are free: the next tile's first weight fragments go out ahead of the stores
      // ---- pass B: split, through the strip, store
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
        u32x4 ph1[4];
#pragma unroll
        for (int nj = 0; nj < 2; ++nj) {
          u32x4 phi[2], plo[2];
#pragma unroll
          for (int gg = 0; gg < 2; ++gg) {
            float v[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) v[c] = acc[mi][nj][8 * gg + c];
            if constexpr (PL == 2) {
              split8(v, sc, phi[gg], plo[gg]);
              strip_put(gg, phi[gg], plo[gg]);
            } else {
              cvt8(v, sc, ph1[2 * nj + gg]);
              strip_put1(nj, gg, ph1[2 * nj + gg]);
            }
          }
          if constexpr (PL == 2) {
            strip_flush(mi, nj);
            keep_planes(phi, plo);
          }
        }
        if constexpr (PL == 1) {
          strip_flush(mi, 0);
          keep_planes1(ph1);
        }
      }
    }
    if (!more) break;
    vb = vbn;
    first = false;
  }
}

// ------------------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------------------
int prof_hook_begin(double flops, int variant, hipStream_t st);   // gemm.hip: per-launch HIP events when profiling is on
void prof_hook_end(int token, hipStream_t st);
int check_kc(const KcArgs& a, bool narrow);                        // bsp_gemm.hip

// workgroup slots of the device: two 256-thread workgroups per CU (registers, LDS).  snerf_test_set_kc_grid (test hook of the
// C-ABI) forces a small grid so that small problems exercise the tile loop and the tile counters.
static int g_kc_grid_override = 0;
void kc_set_grid_override(int n) { g_kc_grid_override = n > 0 ? n : 0; }
static int kc_slots() {
  if (g_kc_grid_override > 0) return g_kc_grid_override;
  static const int n = [] {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    return 2 * cus;
  }();
  return n;
}

static bool cs_bias_check(const KcArgs& a) {
  if (a.colsum != nullptr && a.bias != nullptr) { set_error("bsp gemm: bias and column sums in one launch"); return true; }
  return false;
}

int launch_kc(const KcArgs& a0, hipStream_t st) {
  KcArgs a = a0;
  if (!a.A2) { a.A2 = a.A; a.EA2 = a.EA; a.lda2 = a.lda; a.a2_col0 = a.a_col0; if (a.Ka == 0) a.Ka = a.K; }
  int rc = check_kc(a, false);
  if (rc) return rc;
  if (cs_bias_check(a)) return SNERF_ERR_BAD_DESC;
  a.tiles_i = (a.I + 127) / 128;
  a.tiles_jr = (a.J + 255) / 256;
  if (a.tj_skip >= a.tiles_jr || (a.tj_skip >= 0 && a.tiles_jr < 2)) { set_error("bsp gemm: tj_skip beyond the column tiles"); return SNERF_ERR_BAD_DESC; }
  a.tiles_j = a.tiles_jr - (a.tj_skip >= 0 ? 1 : 0);
  if (a.nd_w != nullptr) {   // the projections' LDS layout: column tile tj's rows at nd_woff[tj], 256 floats each
    if (a.nd_omax == 0) { a.nd_omax = 1; a.nd_ldw = 0; for (int tj = 0; tj < 8; ++tj) { a.nd_rows[tj] = tj < a.tiles_jr ? 1 : 0; a.nd_row0[tj] = 0; } }
    int off = 0;
    for (int tj = 0; tj < a.tiles_jr && tj < 8; ++tj) { a.nd_woff[tj] = off; off += 256 * a.nd_rows[tj]; }
    if (a.tiles_jr > 8 || off > KC_NDW_FLOATS || (a.nd_omax != 1 && a.nd_omax != 5)) { set_error("bsp gemm: folded projections beyond the LDS table"); return SNERF_ERR_BAD_DESC; }
    for (int tj = 0; tj < a.tiles_jr; ++tj) if (a.nd_rows[tj] > a.nd_omax) { set_error("bsp gemm: nd_rows > nd_omax"); return SNERF_ERR_BAD_DESC; }
  }
#ifdef KC_DIAG_BUILD
  constexpr bool DIAG = true;
  static const int dbg = [] { const char* e = getenv("SNERF_KC_DBG"); return e ? atoi(e) : 0; }();
  a.dbg = dbg;
  if (dbg & 2) a.w_bytes = 0;
#else
  constexpr bool DIAG = false;
#endif
  const int ntiles = a.tiles_i * a.tiles_j, slots = kc_slots();
  const dim3 grid(ntiles < slots ? ntiles : slots), block(256);   // persistent workgroups, two per CU; tile = block + n * grid
  const int tok = prof_hook_begin(2.0 * a.I * (double)a.J * a.K, 0, st);
  const bool cs = a.colsum != nullptr;
#define KC_LAUNCH(ACT_, AUX_, CS_, SG_, SM_) do { if (a.pl == 2) hipLaunchKernelGGL((gemm_kc_kernel<2, ACT_, AUX_, CS_, SG_, SM_, DIAG>), grid, block, 0, st, a); \
                                                 else hipLaunchKernelGGL((gemm_kc_kernel<1, ACT_, AUX_, CS_, SG_, SM_, DIAG>), grid, block, 0, st, a); } while (0)
#define KC_LAUNCH_ND(SG_, ND_) do { if (a.pl == 2) hipLaunchKernelGGL((gemm_kc_kernel<2, ACT_SIN, AUX_NONE, false, SG_, SIN_DIRECT, DIAG, ND_>), grid, block, 0, st, a); \
                                    else hipLaunchKernelGGL((gemm_kc_kernel<1, ACT_SIN, AUX_NONE, false, SG_, SIN_DIRECT, DIAG, ND_>), grid, block, 0, st, a); } while (0)
  if (a.aux_mode == AUX_SINREC) KC_LAUNCH(ACT_NONE, AUX_SINREC, true, false, SIN_FRACT);
  else if (a.aux_mode == AUX_RELU_MASK) KC_LAUNCH(ACT_NONE, AUX_RELU_MASK, true, false, SIN_FRACT);
  else if (a.act == ACT_SIN) {
    if (a.nd_w != nullptr && a.nd_omax == 1) { if (a.Csign == nullptr) KC_LAUNCH_ND(false, 1); else KC_LAUNCH_ND(true, 1); }
    else if (a.nd_w != nullptr) { if (a.Csign == nullptr) KC_LAUNCH_ND(false, 5); else KC_LAUNCH_ND(true, 5); }
    else if (a.Csign == nullptr) KC_LAUNCH(ACT_SIN, AUX_NONE, false, false, SIN_DIRECT);
    else KC_LAUNCH(ACT_SIN, AUX_NONE, false, true, SIN_DIRECT);
  }
  else if (a.act == ACT_RELU) KC_LAUNCH(ACT_RELU, AUX_NONE, false, false, SIN_FRACT);
  else if (cs) KC_LAUNCH(ACT_NONE, AUX_NONE, true, false, SIN_FRACT);
  else KC_LAUNCH(ACT_NONE, AUX_NONE, false, false, SIN_FRACT);
#undef KC_LAUNCH
#undef KC_LAUNCH_ND
  SNERF_LAUNCH_CHECK();
  prof_hook_end(tok, st);
  return SNERF_OK;
}

}  // namespace bsp
}  // namespace snerf

// K-contiguous GEMM, SIREN forward launches: 2 x (128 x 256) tiles per workgroup, the two row halves half a period apart.
// (Epilogue arithmetic: bsp_kc.hip.  Reference: semantic/models/rs_semantic.py:325-340.)
//
// What the other two kernels cannot have together: the 128 x 256 kernel (two workgroups per CU) hides every epilogue behind
// the other workgroup's k-loop but each of its waves pulls its own weight fragments from L2 (24 bytes through the L1 per
// output element, the launch is bound by L1 fills); the 256 x 256 kernel shares the weights through LDS (16 bytes) but all
// eight waves reach the epilogue together and nothing computes meanwhile.  Here one workgroup of eight waves per CU is two
// GROUPS of four waves (g = wave >> 2; waves w and w + 4 share a SIMD), each working on 128-row tiles of its own, for ONE
// column tile tj that the workgroup keeps for its whole life:
//
//   stages   The kernel is a loop over global stages sigma = 0, 1, 2, ... with ONE workgroup barrier per stage.  In a stage a
//            wave does one of: a 16-deep k-step of its tile (24 MFMAs), one of 16 epilogue chunks (half a 32 x 32 block:
//            sine, plane split, strip, stores), or nothing.  A group's schedule is periodic: nks k-steps, 16 chunks, then
//            idle stages up to the period P = max(nks + 16, 32); group 1 runs P / 2 stages behind group 0, so one group's
//            epilogue always lies inside the other's k-loop.
//   weights  The weight stream is the same for every tile of the column tile: stage sigma carries k-step sigma mod nks,
//            8 units x 2 KiB in fragment order, in slot sigma & 3 of a four-slot ring; EVERY wave requests two 1 KiB pieces
//            of stage sigma + 4 in every stage, whatever else it is doing.  A tile's k-loop is ANY nks consecutive stages
//            (a cyclic k order: the accumulators do not care, the exponent bookkeeping wraps with it).
//   A        per group: four-slot ring of 128 rows x 64 B, two pieces per wave and stage, four stages ahead; the first four
//            stages of the next tile are requested during the last four epilogue chunks.
//   waits    Every request of a stage is issued AFTER the stage's barrier, at least four per wave and stage (rejected ones
//            where there is nothing to fetch), and every wave waits for vmcnt(8) before the barrier: what was requested in
//            stage sigma - 3 or earlier has landed at the barrier of stage sigma.  Every consumer (fragments, the drawn tile
//            index) uses data requested at least three stages earlier and published by a barrier in between.
//   tiles    row tiles are drawn per (XCD group, tj) from the launch's counters; both tj classes of an XCD walk the same rows
//            in the same order, so the second reader of an activation row tile finds it in that XCD's L2.
#include <type_traits>
#include "bsp_kc_epi.h"

namespace snerf {
namespace bsp {

constexpr int K9_WST = 8 * 2048, K9_AST = 128 * 64;
constexpr int K9_A0 = 4 * K9_WST;                      // group g: + g * 4 * K9_AST
constexpr int K9_STRIP = K9_A0 + 2 * 4 * K9_AST;       // + wn * 4096: the group in its epilogue (the two epilogues never overlap)
constexpr int K9_BIAS = K9_STRIP + 4 * 4096;           // 256 floats: the column tile's bias, times w0 / pi
constexpr int K9_ETAB = K9_BIAS + 1024;                // [group][128] exponents of the k-steps of the group's tile
constexpr int K9_MISC = K9_ETAB + 1024;                // ints: [g] next tile of group g; [2 + g] group g has no tile left
constexpr int K9_LDS = K9_MISC + 64;
constexpr int K9_NE = 16;

template <bool SIGNS, int SINM>
__global__ __launch_bounds__(512, 2) void gemm_kc9_kernel(const KcArgs) {
  const kargs_t p = kargs();
  __shared__ __attribute__((aligned(16))) char lds[K9_LDS];
  // (plain pointer: volatile accesses through a generic pointer become FLAT instructions, whose completion the compiler awaits
  //  with vmcnt(0) -- draining every request in flight; the asm barriers order the accesses)
  int* misc = reinterpret_cast<int*>(lds + K9_MISC);
  const int eW = *p->EW;                       // the weight matrix's exponent
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int g = wave >> 2, wn = wave & 3, wj0 = 64 * wn;
  const int nks = p->K >> 4, nks1 = p->Ka >> 4;
  const bool two_seg = p->Ka < p->K;
  const int P = max(nks + K9_NE, 2 * K9_NE), OFF = P >> 1;
  auto opaque = [](int v) { asm volatile("" : "+v"(v)); return v; };

  // ---- the workgroup's class: XCD group and column tile; its share of the row tiles ----------------------------------------
  const int n_grp = p->n_grp, tiles_j = p->tiles_j;
  const int xcd = (int)blockIdx.x & (n_grp - 1), rr = (int)blockIdx.x / n_grp;
  const int tj = rr % tiles_j, rc = rr / tiles_j, cls = (int)gridDim.x / (n_grp * tiles_j);
  const int nrt = p->tiles_i, per = (nrt + n_grp - 1) / n_grp, q0 = xcd * per;
  const int qn = max(0, min(per, nrt - q0));
  const int j0 = 256 * tj;
  int* const tile_ctr = p->tile_ctr + (xcd * tiles_j + tj);

  // ---- per-lane constants ---------------------------------------------------------------------------------------------
  // A piece = 16 rows x 64 B; chunk c of row r at position c ^ ((r >> 2) & 3) (bsp_kc8.hip).  Wave wn requests pieces 2 wn, 2 wn + 1.
  auto a_lane_off = [&](int qq, int ld, int l) -> unsigned {
    const int row = 16 * (2 * wn + qq) + (l >> 2);
    return (unsigned)row * (unsigned)ld * 4u + 16u * (unsigned)((l & 3) ^ ((l >> 4) & 3));
  };
  char* const ringA = lds + K9_A0 + g * 4 * K9_AST;
  const unsigned dstA = (unsigned)__builtin_amdgcn_readfirstlane(lds_addr(ringA + (2 * wn) * 1024));
  const unsigned dstW = (unsigned)__builtin_amdgcn_readfirstlane(lds_addr(lds + (2 * wave) * 1024));
  const int rowl = lane & 31, kh = lane >> 5, swz = (rowl >> 2) & 3;
  unsigned fo[2];
#pragma unroll
  for (int pl = 0; pl < 2; ++pl) fo[pl] = (unsigned)rowl * 64u + (unsigned)(((2 * pl + kh) ^ swz) << 4);
  const unsigned wo = (unsigned)(2 * wn) * 2048u + 16u * (unsigned)lane;
  const srd_words srdW = make_srd_words(p->W, p->w_bytes);
  const unsigned w_ks0 = (unsigned)(p->w_k0 >> 4), w_rb32 = (unsigned)p->w_rb32;
  const unsigned w_u0 = (unsigned)((p->w_row0 + j0) >> 5);
  const unsigned voW = (unsigned)(2 * wave) * 1024u + 16u * (unsigned)lane;
  if (t < 256) reinterpret_cast<float*>(lds + K9_BIAS)[t] = (p->bias != nullptr && j0 + t < p->J) ? p->bias[j0 + t] * p->w0 * INV_PI : 0.f;
  const float* sbias = reinterpret_cast<const float*>(lds + K9_BIAS);
  int* etab = reinterpret_cast<int*>(lds + K9_ETAB + g * 512);

  // ---- state of the tile whose operands are being requested / which is being computed ----------------------------------------
  int i0 = 0, k0 = 0, e_last = 0, cur_seg = 0;
  unsigned long long chg0 = 0, chg1 = 0;
  srd_words srdCur;   // (the requests are asm statements: as builtins, hipcc drains them -- vmcnt(0) -- in front of every LDS access it
  unsigned voCur[2];  //  cannot tell apart from their destination, i.e. in every epilogue stage; bsp_dev.h)
  auto set_segment = [&](int seg) {
    const kargs_t a = kargs();
    const int l = opaque(lane);
    if (seg == 0) {
      srdCur = make_srd_words(a->A + ((size_t)i0 * a->lda + a->a_col0) * 4, clamp_bytes(((unsigned long long)(a->I - i0 - 1) * a->lda + a->Ka) * 4ull));
#pragma unroll
      for (int qq = 0; qq < 2; ++qq) voCur[qq] = a_lane_off(qq, a->lda, l);
    } else {
      srdCur = make_srd_words(a->A2 + ((size_t)i0 * a->lda2 + a->a2_col0) * 4, clamp_bytes(((unsigned long long)(a->I - i0 - 1) * a->lda2 + (a->K - a->Ka)) * 4ull));
#pragma unroll
      for (int qq = 0; qq < 2; ++qq) voCur[qq] = a_lane_off(qq, a->lda2, l);
    }
    cur_seg = seg;
  };
  // Tile q of the class, its k-loop starting with k-step kstart (cyclic).  (The exponent loads are awaited on the spot, which
  // awaits every older request of the wave -- one drain of the rings per tile, in epilogue chunk 6.  Requesting them by LDS-DMA
  // or asm loads for use three stages later made hipcc spill the accumulators.)
  auto prepare = [&](int q, int kstart) {
    const kargs_t a = kargs();
    const int l = opaque(lane);
    const int ti = q0 + (a->rev ? qn - 1 - q : q);
    i0 = 128 * ti;
    k0 = kstart;
    set_segment((two_seg && kstart >= nks1) ? 1 : 0);
    const int* EA1 = a->EA; const int* EA2 = a->EA2;
    const int ncb1 = ncb_of(a->lda), ncb2 = ncb_of(a->lda2), ac1 = a->a_col0, ac2 = a->a2_col0;
    auto exp_of = [&](int s) {
      const bool seg2 = s >= nks1;
      const int* E = seg2 ? EA2 : EA1;
      const int col = seg2 ? ac2 + 16 * (s - nks1) : ac1 + 16 * s;
      return E[(size_t)ti * (seg2 ? ncb2 : ncb1) + (col >> 7)];
    };
    const int eA = l < nks ? exp_of(l) : 0;
    const int eB = l + 64 < nks ? exp_of(l + 64) : 0;
    const int e_wrap = exp_of(nks - 1);
    int eAp = __builtin_amdgcn_update_dpp(eA, eA, 0x138, 0xF, 0xF, false);
    int eBp = __builtin_amdgcn_update_dpp(eB, eB, 0x138, 0xF, 0xF, false);
    const int eA63 = __builtin_amdgcn_readlane(eA, 63);
    if (l == 0) { eAp = e_wrap; eBp = eA63; }
    if (l + 64 >= nks) eBp = eB;
    if (l >= nks) eAp = eA;
    chg0 = __builtin_amdgcn_ballot_w64(eA != eAp);
    chg1 = __builtin_amdgcn_ballot_w64(eB != eBp);
    e_last = exp_of(kstart == 0 ? nks - 1 : kstart - 1);
    if (wn == 0) { etab[l] = eA; etab[l + 64] = eB; }
  };
  // tile-local stage j of the prepared tile -> ring slot j & 3; beyond the tile: a rejected request (it still counts)
  auto issueA = [&](int j, bool valid) {
    unsigned so = OOB;
    if (valid) {
      int k = k0 + j; if (k >= nks) k -= nks;
      const int seg = (two_seg && k >= nks1) ? 1 : 0;
      if (__builtin_expect(seg != cur_seg, 0)) set_segment(seg);
      so = (unsigned)(seg ? k - nks1 : k) * 64u;
    }
    asm volatile("" : "+s"(so));   // a scalar register, never a literal (MUBUF takes none as its scalar offset)
    dma16_asm(srdCur, dstA + (unsigned)((j & 3) * K9_AST), voCur[0], so);
    dma16_asm(srdCur, dstA + (unsigned)((j & 3) * K9_AST + 1024), voCur[1], so);
  };
  auto issueW = [&](int sg, int kw) {     // global stage sg carrying k-step kw
    unsigned so = ((w_ks0 + (unsigned)kw) * w_rb32 + w_u0) * 2048u;
    asm volatile("" : "+s"(so));
    dma16_asm(srdW, dstW + (unsigned)((sg & 3) * K9_WST), voW, so);
    dma16_asm(srdW, dstW + (unsigned)((sg & 3) * K9_WST + 1024), voW + 1024u, so);
  };

  // ---- tiles of the group: the first one by rank, the following ones drawn from the class counter ---------------------------
  int q_cur = 2 * rc + g;                 // tile of the current period
  int q_nxt = 0;                          // tile of the next period (read from misc during the epilogue)
  bool have = q_cur < qn;                 // the group has a tile in the current period
  bool have_nxt = false;
  bool done = false;                      // nothing left for this group
  if (t < 4) misc[t] = 0;

  // ---- prologue: weight stages 0-3; group 0's first tile starts at stage 0, group 1's at stage OFF ------------------------
  {
    int kq = 0;
#pragma unroll
    for (int s = 0; s < 4; ++s) { issueW(s, kq); kq = kq + 1 == nks ? 0 : kq + 1; }
  }
  if (have) prepare(q_cur, g == 0 ? 0 : OFF % nks);
  if (g == 0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) issueA(j, have && j < nks);
  }
  wait_vm<0>();
  __syncthreads();                        // misc zeroed, bias staged, exponent tables written
  if (!have) { done = true; if (wn == 0 && lane == 0) misc[2 + g] = 1; }
  __syncthreads();

  struct AFrag { f16x8 h[4], l[4]; };
  struct WFrag { f16x8 h[2], l[2]; };
  auto keep_planes = [](const u32x4 (&hi)[2], const u32x4 (&lo)[2]) { asm volatile("" ::"v"(hi[0]), "v"(lo[0]), "v"(hi[1]), "v"(lo[1])); };

  int sg = 0;                                   // global stage
  int kw = 0;                                   // k-step carried by the weight stage of the current global stage
  int kw4 = 4 % nks;                            // ... by the stage requested now (sigma + 4)
  auto advance = [&]() { ++sg; kw = kw + 1 == nks ? 0 : kw + 1; kw4 = kw4 + 1 == nks ? 0 : kw4 + 1; };
  // a stage without work of its own: wait, barrier, the stage's requests (stage jn of the coming tile where it is due)
  auto idle_stage = [&](int jn, bool nxt) -> bool {
    wait_vm<8>();
    barrier_raw();
    if (done && misc[2] != 0 && misc[3] != 0) return true;   // both groups are out of tiles (flags are set before a barrier, read behind it)
    issueA(jn & 3, nxt && jn >= 0 && jn < nks);
    issueW(sg + 4, kw4);
    advance();
    return false;
  };

  bool quit = false;
  if (g == 1)
    for (int tau = -OFF; tau < 0 && !quit; ++tau) quit = idle_stage(tau + 4, have);

  for (int period = 0; !quit; ++period) {   // one period of the group per iteration
    if (!have) {
      if (!done) { done = true; if (wn == 0 && lane == 0) misc[2 + g] = 1; }   // before the next barrier: both groups read it behind it
      while (!idle_stage(-1, false)) {}
      break;
    }
    f32x16 acc[4][2];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int nj = 0; nj < 2; ++nj)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mi][nj][r] = 0.f;
    {
      // ---- k-loop: nks stages.  The fragments of the first stage are read here, those of stage j + 1 inside stage j.
      AFrag fa;
      WFrag w0, w1;
      {
        const char* sw0 = lds + (sg & 3) * K9_WST;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) { fa.h[mi] = ldsfrag(ringA + 2048 * mi + fo[0]); fa.l[mi] = ldsfrag(ringA + 2048 * mi + fo[1]); }
#pragma unroll
        for (int nj = 0; nj < 2; ++nj) { w0.h[nj] = ldsfrag(sw0 + wo + 2048 * nj); w0.l[nj] = ldsfrag(sw0 + wo + 2048 * nj + 1024); }
      }
      const bool after_epi = period > 0 && P == nks + K9_NE;   // the k-loop follows the previous tile's last chunk directly
      auto stage = [&](int j, WFrag& wc, WFrag& wx) {
        int k = k0 + j; if (k >= nks) k -= nks;
        if (j != 0 && __builtin_expect((((k & 64) ? chg1 : chg0) >> (k & 63)) & 1ull, 0)) {
          const int de = etab[k] - etab[k == 0 ? nks - 1 : k - 1];
#pragma unroll
          for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int nj = 0; nj < 2; ++nj) acc[mi][nj] = scale_acc(acc[mi][nj], de);
        }
        const char* sa = ringA + ((j + 1) & 3) * K9_AST;          // the tile's stage j + 1
        const char* swn = lds + ((sg + 1) & 3) * K9_WST;          // weight stage sigma + 1
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
          acc[mi][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wc.l[0], fa.h[mi], acc[mi][0], 0, 0, 0);
          acc[mi][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wc.l[1], fa.h[mi], acc[mi][1], 0, 0, 0);
          if (mi == 0) {
            __builtin_amdgcn_sched_barrier(0);
            // what was requested three stages ago or earlier has landed: 8 younger requests, 12 behind an odd epilogue chunk (4 stores)
            if (j < 2 && after_epi) wait_vm<12>(); else wait_vm<8>();
            barrier_raw();
            __builtin_amdgcn_sched_barrier(0);
          }
          if (mi == 2) issueA(j + 4, j + 4 < nks);
          if (mi == 3) issueW(sg + 4, kw4);
          acc[mi][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wc.h[0], fa.l[mi], acc[mi][0], 0, 0, 0);
          acc[mi][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wc.h[1], fa.l[mi], acc[mi][1], 0, 0, 0);
          const f16x8 nl = ldsfrag(sa + 2048 * mi + fo[1]);
          if (mi == 0) wx.l[0] = ldsfrag(swn + wo + 1024);
          if (mi == 1) wx.h[0] = ldsfrag(swn + wo);
          acc[mi][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wc.h[0], fa.h[mi], acc[mi][0], 0, 0, 0);
          acc[mi][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wc.h[1], fa.h[mi], acc[mi][1], 0, 0, 0);
          fa.l[mi] = nl;
          fa.h[mi] = ldsfrag(sa + 2048 * mi + fo[0]);
          if (mi == 0) wx.l[1] = ldsfrag(swn + wo + 2048 + 1024);
          if (mi == 1) wx.h[1] = ldsfrag(swn + wo + 2048);
          __builtin_amdgcn_sched_barrier(0);
        }
        advance();
      };
      __builtin_amdgcn_s_setprio(2);   // MFMA issue goes before the other group's epilogue stream on the same SIMD
      for (int j = 0; j < nks; j += 2) {
        stage(j, w0, w1);
        if (j + 1 < nks) stage(j + 1, w1, w0);
      }
      __builtin_amdgcn_s_setprio(0);
    }
    {
      // ---- epilogue: 16 stages, one chunk each.  Chunk c = half (gg) of the 32 x 32 block (mi, nj): u = acc * su + b, sine,
      //      planes into the strip; the block's second half flushes the strip (4 stores of 8 x 128 B); the last chunk of a
      //      32-point block stores its sign words.
      const kargs_t e = kargs();
      // lane-derived constants of the epilogue, recomputed from an opaque copy of the lane index: derived from `lane` itself they
      // are loop-invariant, get hoisted and occupy registers all through the k-loop
      const int el = opaque(lane);
      const int pt = el & 31, lh = el >> 5;
      char* const strip = lds + K9_STRIP + wn * 4096;
      const unsigned sw_off = (unsigned)pt * 128u;
      const int srow = el >> 3, schunk = el & 7;
      const unsigned sr_off = (unsigned)srow * 128u + 16u * (unsigned)(schunk ^ srow);
      const int e_in = e_last + eW;                 // acc = true value * 2^e_in
      const bool e_small = e_in >= -120 && e_in <= 120;
      if (!e_small) {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
          for (int nj = 0; nj < 2; ++nj) acc[mi][nj] = scale_acc(acc[mi][nj], -e_in);
      }
      const float su = (e_small ? pow2f(-e_in) : 1.f) * e->w0 * INV_PI;
      const int c_r0 = i0;
      const int nrows = min(128, e->I - c_r0);
      const int jw = j0 + wj0;
      const bool wave_cols = jw < e->J && nrows > 0;
      const size_t offC = uniform_sz(((size_t)c_r0 * e->ldc + e->c_col0) * 4);
      const srd_t srdC = make_srd(e->C + offC, nrows > 0 ? clamp_bytes(((unsigned long long)(nrows - 1) * e->ldc + e->J) * 4ull) : 0u);
      unsigned voC[2];
#pragma unroll
      for (int nj = 0; nj < 2; ++nj)
        voC[nj] = jw + 32 * nj + 16 * (schunk >> 2) < e->J ? (unsigned)srow * (unsigned)e->ldc * 4u + (unsigned)((jw >> 4) + 2 * nj) * 64u + 16u * (unsigned)schunk : OOBH;
      const unsigned stepC8 = 8u * (unsigned)e->ldc * 4u;
      unsigned sw = 0u;
      u32x4 phi[2], plo[2];
      int drawn = 0;
#pragma unroll
      for (int c = 0; c < K9_NE; ++c) {
        const int mi = c >> 2, nj = (c >> 1) & 1, gg = c & 1, gq = 2 * nj + gg;
        // before the barrier: the drawn tile index for the group's other waves (requested in chunk 0: home since the wait of chunk 3)
        if (c == 4 && wn == 0 && el == 0) { asm volatile("" : "+v"(drawn)); misc[g] = 2 * cls + drawn; }
        // what the two stages before this one have issued (it may still be in flight): a chunk requests 2 weight pieces (+ 2 A pieces
        // of the next tile from chunk 12 on) and, odd chunks, 4 stores; a k-loop stage 4 pieces
        if (c == 0) wait_vm<8>(); else if (c == 1) wait_vm<6>(); else if (c <= 12) wait_vm<8>(); else if (c == 13) wait_vm<10>(); else wait_vm<12>();
        barrier_raw();
        if (c >= 12) issueA((nks + c + 4 - P) & 3, have_nxt && nks + c + 4 - P >= 0 && nks + c + 4 - P < nks);   // (rejected where not due: the count is fixed)
        issueW(sg + 4, kw4);
        if (c == 0 && wn == 0 && opaque(lane) == 0)   // the tile after the next, from the class counter
          asm volatile("s_nop 4\n\tglobal_atomic_add %0, %1, %2, %3 sc0" : "=v"(drawn) : "v"(0), "v"(1), "s"(tile_ctr) : "memory");
        if (c == 4) { q_nxt = __builtin_amdgcn_readfirstlane(misc[g]); have_nxt = q_nxt < qn; }
        const float4 b0 = *reinterpret_cast<const float4*>(&sbias[wj0 + 16 * gq + 8 * lh]);
        const float4 b1 = *reinterpret_cast<const float4*>(&sbias[wj0 + 16 * gq + 8 * lh + 4]);
        const float bj[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
        float v[8];
#pragma unroll
        for (int cc = 0; cc < 8; ++cc) v[cc] = fmaf(acc[mi][nj][8 * gg + cc], su, bj[cc]);
        sinpi8<SIGNS, SINM>(v, sw);
        split8(v, 8192.f, phi[gg], plo[gg]);
        *reinterpret_cast<u32x4*>(strip + sw_off + 16 * ((4 * gg + lh) ^ (pt & 7))) = phi[gg];
        *reinterpret_cast<u32x4*>(strip + sw_off + 16 * ((4 * gg + 2 + lh) ^ (pt & 7))) = plo[gg];
        if (gg == 1) {
#pragma unroll
          for (int ps = 0; ps < 4; ++ps) {
            const u32x4 d = *reinterpret_cast<const u32x4*>(strip + sr_off + 1024 * ps);
            __builtin_amdgcn_raw_buffer_store_b128(d, srdC, voC[nj], (unsigned)(4 * mi + ps) * stepC8, 0);
          }
          keep_planes(phi, plo);   // bsp_kc.hip: the ds_write_b128 data hazard
        }
        if ((c & 3) == 3) {
          if (SIGNS && e->Csign != nullptr && wave_cols && 32 * mi < nrows)
            e->Csign[((size_t)((c_r0 >> 5) + mi) * ((e->ldc + 63) >> 6) + ((e->c_col0 + jw) >> 6)) * 64 + el] = sw;
          sw = 0u;
        }
        // the next tile: the finished tile's k-loop state is free; its operands are requested from four stages before its k-loop on
        if (c == 6 && have_nxt) prepare(q_nxt, (kw + (P - nks - 6)) % nks);
        advance();
      }
      if ((wn & 1) == 0 && el == 0 && wave_cols) e->EC[(size_t)(c_r0 >> 7) * ncb_of(e->ldc) + ((e->c_col0 + jw) >> 7)] = 13;
    }
    // ---- idle stages up to the period (short contractions only)
    for (int ph = nks + K9_NE; ph < P; ++ph) idle_stage(ph + 4 - P, have_nxt);
    q_cur = q_nxt; have = have_nxt; have_nxt = false;
  }
  wait_vm<0>();
}

// ------------------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------------------
int prof_hook_begin(double flops, int variant, hipStream_t st);
void prof_hook_end(int token, hipStream_t st);

// SIREN forward launches with a tile counter slot and at most two column tiles; `a` has passed check_kc.  Returns -1 when the
// launch does not qualify (the caller takes the 128 x 256 kernel).
int launch_kc9(KcArgs a, bool sin_hw, hipStream_t st) {
  if (a.act != ACT_SIN || a.aux_mode != AUX_NONE || a.colsum != nullptr || a.tile_ctr == nullptr) return -1;
  a.tiles_j = (a.J + 255) / 256;
  a.tiles_i = (a.I + 127) / 128;              // row tiles of 128
  if (a.tiles_j > 2) return -1;
  static const int slots = [] {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    if (const char* e = getenv("SNERF_KC_GRID")) { const int v = atoi(e); if (v > 0) cus = v; }
    return cus;
  }();
  a.n_grp = (a.tiles_i >= 64 && slots >= 8 * a.tiles_j) ? 8 : 1;
  const int per = (a.tiles_i + a.n_grp - 1) / a.n_grp;
  int cls = slots / (a.n_grp * a.tiles_j);
  if (cls < 1) return -1;
  if (cls > (per + 1) / 2) cls = (per + 1) / 2;
  const dim3 grid(a.n_grp * a.tiles_j * cls), block(512);
  const int tok = prof_hook_begin(2.0 * a.I * (double)a.J * a.K, 0, st);
  if (a.Csign == nullptr) { if (sin_hw) hipLaunchKernelGGL((gemm_kc9_kernel<false, SIN_HW>), grid, block, 0, st, a); else hipLaunchKernelGGL((gemm_kc9_kernel<false, SIN_POLY>), grid, block, 0, st, a); }
  else { if (sin_hw) hipLaunchKernelGGL((gemm_kc9_kernel<true, SIN_HW>), grid, block, 0, st, a); else hipLaunchKernelGGL((gemm_kc9_kernel<true, SIN_POLY>), grid, block, 0, st, a); }
  SNERF_LAUNCH_CHECK();
  prof_hook_end(tok, st);
  return SNERF_OK;
}

}  // namespace bsp
}  // namespace snerf

// K-contiguous GEMM on block-scaled fp16-plane tensors, 256 x 256 tile per workgroup (bsp_kc.hip holds the epilogue
// arithmetic's description; reference: semantic/models/rs_semantic.py:325-340, :260-313 and their autograd backward).
//
// Why this shape.  The 128 x 256 kernel (two workgroups per CU, weights from L2 straight into registers) pulls 24 bytes
// through a CU's L1 per output element at K = 512 -- 3.2 GB per launch, two thirds of it weight fragments -- and PMC puts
// it at the rate a CU's L1 can be filled (25 M 128-byte requests at 459 cycles average latency = 64-70 requests in flight
// per CU, TCP_PENDING_STALL 51 % of the cycles; loosening every counted wait changed nothing): neither the matrix pipe nor
// HBM is the limit, the L2 -> L1 fill path is.  One workgroup of eight waves per CU on a 256 x 256 tile shares every weight
// fragment between the two row halves through LDS: 16 bytes per output element.
//
//   waves   wave = 4 wm + wn: rows 128 wm .. + 127, columns 64 wn .. + 63 of the tile (accumulators as in the 128-row kernel:
//           W is the MFMA A operand, lane l holds point l & 31 and two runs of eight columns per 32 x 32 block)
//   ring    four stages of 16 k: [256 rows x 64 B of activations | 8 units x 2 KiB of weights in fragment order], filled by
//           LDS-DMA, four 1 KiB pieces per wave and stage (2 A + 2 W); stage s + 4 is requested during stage s, after the
//           stage's barrier (every wave has the fragments of stage s in registers by then: its slot is free)
//   stage   24 MFMAs per wave; the fragments of stage s + 1 replace those of stage s as the MFMAs release them (activations in
//           place, weights into the second register set); one barrier per stage, after the wave's own pieces of stage
//           s + 1 have landed (s_waitcnt vmcnt(8): the eight younger requests belong to stages s + 2, s + 3)
//   tiles   persistent workgroups, one per CU, tiles drawn from per-XCD-group counters as in the 128-row kernel; the next tile's
//           stages 0 and 1 are requested during the epilogue (ring slots 2, 3 hold the epilogue's strips and buffers)
#include "bsp_kc_epi.h"

namespace snerf {
namespace bsp {

constexpr int K8_STA = 256 * 64;                 // activations of one stage
constexpr int K8_ST = K8_STA + 8 * 2048;         // + the weights of its 256 columns
constexpr int K8_RING = 4, K8_RINGB = K8_RING * K8_ST;
constexpr int K8_XBUF = 2 * K8_ST;               // epilogue: 8 KiB per wave inside ring slots 2, 3 (stored-activation buffer | strip)
constexpr int K8_BIAS = K8_RINGB;                // 2 x 256 floats, by tile parity
constexpr int K8_ETAB = K8_BIAS + 2 * 1024;      // [tile parity][row half][128] exponents of the k-steps
constexpr int K8_SMAX = K8_ETAB + 2 * 2 * 512;   // the waves' maxima
constexpr int K8_HSIGN = K8_SMAX + 64;           // four 256-byte sign-word slots per wave
constexpr int K8_NEXT = K8_HSIGN + 8 * 1024;
constexpr int K8_LDS = K8_NEXT + 64;

template <int ACT, int AUX, bool COLSUM, bool SIGNS, int SINM>
__global__ __launch_bounds__(512, 2) void gemm_kc8_kernel(const KcArgs) {
  const kargs_t p = kargs();
  __shared__ __attribute__((aligned(16))) char lds[K8_LDS];
  float* smax = reinterpret_cast<float*>(lds + K8_SMAX);
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int wj0 = wn * 64;
  const int tiles_i = p->tiles_i, tiles_j = p->tiles_j, ntiles = tiles_i * tiles_j;
  const int nks = p->K >> 4, nks1 = p->Ka >> 4;
  const int eW = *p->EW;                                      // the weight matrix's exponent (read once: a load inside the tile loop is awaited with
                                                              //  vmcnt(0), i.e. behind the next tile's operand requests)
  const int nrb = (p->I + 127) >> 7;
  constexpr bool ONEPASS = ACT == ACT_SIN;
  constexpr bool BIAS = AUX == AUX_NONE && !COLSUM;

  // ---- per-lane constants ---------------------------------------------------------------------------------------------
  // A piece = 16 rows x 64 B ([hi k 0-7 | hi k 8-15 | lo k 0-7 | lo k 8-15]); chunk c of row r sits at position c ^ ((r >> 2) & 3):
  // a ds_read_b128 lane group (rows {0-3, 12-15, 20-27} or {4-11, 16-19, 28-31}, one chunk) then covers sixteen different
  // 16-byte slots of the 256-byte bank row.  Rows beyond I are rejected by the descriptor.
  auto a_lane_off = [&](int q, int ld, int l) -> unsigned {
    const int row = 16 * (2 * wave + q) + (l >> 2);
    return (unsigned)row * (unsigned)ld * 4u + 16u * (unsigned)((l & 3) ^ ((l >> 4) & 3));
  };
  auto opaque = [](int v) { asm volatile("" : "+v"(v)); return v; };
  char* const dstA = lds + (2 * wave) * 1024;
  char* const dstW = lds + K8_STA + (2 * wave) * 1024;
  const int rowl = lane & 31, kh = lane >> 5, swz = (rowl >> 2) & 3;
  unsigned fo[2];
#pragma unroll
  for (int pl = 0; pl < 2; ++pl) fo[pl] = (unsigned)(128 * wm + rowl) * 64u + (unsigned)(((2 * pl + kh) ^ swz) << 4);
  const unsigned wo = (unsigned)K8_STA + (unsigned)(2 * wn) * 2048u + 16u * (unsigned)lane;
  const srd_t srdW = make_srd(p->W, p->w_bytes);
  const unsigned w_ks0 = (unsigned)(p->w_k0 >> 4), w_rb32 = (unsigned)p->w_rb32;
  const bool two_seg = p->Ka < p->K;
  const unsigned voW = (unsigned)(2 * wave) * 1024u + 16u * (unsigned)lane;

  // ---- state of the tile whose operands are being requested ------------------------------------------------------------
  int ti = 0, tj = 0, i0 = 0, j0 = 0, e_last = 0;
  unsigned w_u0 = 0;
  int eA = 0, eB = 0;
  float bias_t = 0.f;
  srd_t srdCur;
  unsigned voCur[2];
  int sbias_st = 0;
  const int seg_switch = two_seg ? nks1 : 0x7fffffff;
  auto prepare = [&](int vb) {
    const kargs_t a = kargs();
    const int l = opaque(lane);
    tile_of_block(vb, tiles_i, tiles_j, ti, tj, a->rev != 0);
    i0 = ti * 256; j0 = tj * 256;
    const int lda = a->lda;
    srdCur = make_srd(a->A + ((size_t)i0 * lda + a->a_col0) * 4, clamp_bytes(((unsigned long long)(a->I - i0 - 1) * lda + a->Ka) * 4ull));
    sbias_st = 0;
#pragma unroll
    for (int q = 0; q < 2; ++q) voCur[q] = a_lane_off(q, lda, l);
    w_u0 = (unsigned)((a->w_row0 + j0) >> 5);
    const int* EA1 = a->EA; const int* EA2 = a->EA2;
    const int ncb1 = ncb_of(lda), ncb2 = ncb_of(a->lda2), ac1 = a->a_col0, ac2 = a->a2_col0;
    const int rb = min(2 * ti + wm, nrb - 1);        // the wave's 128-row block (a lower half beyond I: any valid one)
    auto exp_of = [&](int s) {
      const bool seg2 = s >= nks1;
      const int* E = seg2 ? EA2 : EA1;
      const int col = seg2 ? ac2 + 16 * (s - nks1) : ac1 + 16 * s;
      return E[(size_t)rb * (seg2 ? ncb2 : ncb1) + (col >> 7)];
    };
    eA = l < nks ? exp_of(l) : 0;
    eB = l + 64 < nks ? exp_of(l + 64) : 0;
    e_last = exp_of(nks - 1);
    if (BIAS) { const int tt = opaque(t) & 255; bias_t = (a->bias != nullptr && j0 + tt < a->J) ? a->bias[j0 + tt] : 0.f; }
  };
  auto enter_stage = [&](int S) {     // before the first piece of stage S
    if (__builtin_expect(S == seg_switch, 0)) {
      const kargs_t a = kargs();
      srdCur = make_srd(a->A2 + ((size_t)i0 * a->lda2 + a->a2_col0) * 4, clamp_bytes(((unsigned long long)(a->I - i0 - 1) * a->lda2 + (a->K - a->Ka)) * 4ull));
      sbias_st = nks1;
#pragma unroll
      for (int q = 0; q < 2; ++q) voCur[q] = a_lane_off(q, a->lda2, opaque(lane));
    }
  };
  auto issueA = [&](int S, int slot) {
    const unsigned so = S < nks ? (unsigned)(S - sbias_st) * 64u : OOB;
    dma16(srdCur, dstA + slot * K8_ST, voCur[0], so);
    dma16(srdCur, dstA + slot * K8_ST + 1024, voCur[1], so);
  };
  auto issueW = [&](int S, int slot) {
    const unsigned so = S < nks ? ((w_ks0 + (unsigned)S) * w_rb32 + w_u0) * 2048u : OOB;
    dma16(srdW, dstW + slot * K8_ST, voW, so);
    dma16(srdW, dstW + slot * K8_ST + 1024, voW + 1024u, so);
  };
  auto heads = [&]() {                // stages 0 and 1 of the prepared tile
    enter_stage(0); issueA(0, 0); issueW(0, 0);
    enter_stage(1); issueA(1, 1); issueW(1, 1);
  };

  int* const tile_ctr = p->tile_ctr;
  const bool dyn = tile_ctr != nullptr;
  const int n_grp = (gridDim.x & 7) == 0 ? 8 : 1, grp = (int)blockIdx.x & (n_grp - 1);
  const unsigned ctr_off = 4u * (unsigned)grp;
  // The tile after the next one is drawn from the counter of the workgroup's XCD group as soon as the next one's operands have
  // been requested (a returning atomic of ONE lane: a whole tile passes before its value is read).
  int drawn = 0;
  auto draw = [&]() {
    if (dyn && wave == 0 && opaque(lane) == 0)
      asm volatile("s_nop 4\n\tglobal_atomic_add %0, %1, %2, %3 sc0" : "=v"(drawn) : "v"(ctr_off), "v"(1), "s"(tile_ctr) : "memory");
  };
  int vb = blockIdx.x;
  prepare(vb);
  heads();
  draw();
  bool first = true;
  for (int it = 0;; ++it) {
    float* sbias = reinterpret_cast<float*>(lds + K8_BIAS + (it & 1) * 1024);
    int* etab = reinterpret_cast<int*>(lds + K8_ETAB + (it & 1) * 1024 + wm * 512);
    int eAp = __builtin_amdgcn_update_dpp(eA, eA, 0x138, 0xF, 0xF, false);       // wave_shr 1: the previous step's exponent
    int eBp = __builtin_amdgcn_update_dpp(eB, eB, 0x138, 0xF, 0xF, false);
    const int eA63 = __builtin_amdgcn_readlane(eA, 63);
    if (opaque(lane) == 0) eBp = eA63;
    if (lane + 64 >= nks) eBp = eB;
    if (lane >= nks) eAp = eA;
    if (wn == 0) { etab[lane] = eA; etab[lane + 64] = eB; }
    if (BIAS && t < 256) sbias[t] = bias_t * (ACT == ACT_SIN ? kargs()->w0 * INV_PI : 1.f);
    const unsigned long long chg0 = __builtin_amdgcn_ballot_w64(eA != eAp), chg1 = __builtin_amdgcn_ballot_w64(eB != eBp);

    f32x16 acc[4][2];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int nj = 0; nj < 2; ++nj)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mi][nj][r] = 0.f;

    struct AFrag { f16x8 h[4], l[4]; };
    struct WFrag { f16x8 h[2], l[2]; };
    AFrag fa;
    WFrag w0, w1;
    // One stage s: 24 MFMAs on `fa` and the weight set `wc`; the fragments of stage s + 1 are read behind the barrier, the
    // activations in place, the weights into `wx`.  Every non-MFMA instruction sits in a gap between MFMAs.
    auto step = [&](int s, int slot, WFrag& wc, WFrag& wx) {
      if (__builtin_expect((((s & 64) ? chg1 : chg0) >> (s & 63)) & 1ull, 0)) {
        const int de = etab[s] - etab[s - 1];
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
          for (int nj = 0; nj < 2; ++nj) acc[mi][nj] = scale_acc(acc[mi][nj], de);
      }
      const char* sn = lds + ((slot + 1) & 3) * K8_ST;   // stage s + 1
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
        acc[mi][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wc.l[0], fa.h[mi], acc[mi][0], 0, 0, 0);
        acc[mi][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wc.l[1], fa.h[mi], acc[mi][1], 0, 0, 0);
        if (mi == 0) {   // own pieces of stage s + 1 are home; behind the barrier everybody's are, and nobody reads stage s any more
          __builtin_amdgcn_sched_barrier(0);
          wait_vm<8>();
          barrier_raw();
          __builtin_amdgcn_sched_barrier(0);
        }
        if (mi == 2) { enter_stage(s + 4); issueA(s + 4, slot); }
        if (mi == 3) issueW(s + 4, slot);
        acc[mi][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wc.h[0], fa.l[mi], acc[mi][0], 0, 0, 0);
        acc[mi][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wc.h[1], fa.l[mi], acc[mi][1], 0, 0, 0);
        const f16x8 nl = ldsfrag(sn + 2048 * mi + fo[1]);
        // the next stage's first MFMAs want the lo weight fragments, its third pair the hi ones
        if (mi == 0) wx.l[0] = ldsfrag(sn + wo + 1024);
        if (mi == 1) wx.h[0] = ldsfrag(sn + wo);
        acc[mi][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wc.h[0], fa.h[mi], acc[mi][0], 0, 0, 0);
        acc[mi][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wc.h[1], fa.h[mi], acc[mi][1], 0, 0, 0);
        fa.l[mi] = nl;
        fa.h[mi] = ldsfrag(sn + 2048 * mi + fo[0]);
        if (mi == 0) wx.l[1] = ldsfrag(sn + wo + 2048 + 1024);
        if (mi == 1) wx.h[1] = ldsfrag(sn + wo + 2048);
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    // Tile start: this wave's pieces of stages 0 and 1 are home.  First tile: they (and the draw) are the only requests.
    // Later tiles: they were requested before the epilogue's stores, of which there are at least 32.  The barrier publishes
    // stage 0 and ends every wave's use of the epilogue's LDS regions before the requests of stages 2 and 3 go out.
    if (first) wait_vm<0>(); else wait_vm<32>();
    barrier_raw();
    enter_stage(2); issueA(2, 2); issueW(2, 2);
    enter_stage(3); issueA(3, 3); issueW(3, 3);
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) { fa.h[mi] = ldsfrag(lds + 2048 * mi + fo[0]); fa.l[mi] = ldsfrag(lds + 2048 * mi + fo[1]); }
#pragma unroll
    for (int nj = 0; nj < 2; ++nj) { w0.h[nj] = ldsfrag(lds + wo + 2048 * nj); w0.l[nj] = ldsfrag(lds + wo + 2048 * nj + 1024); }
    for (int s = 0; s < nks; s += 4) {
      step(s, 0, w0, w1);
      if (s + 1 < nks) step(s + 1, 1, w1, w0);
      if (s + 2 < nks) step(s + 2, 2, w0, w1);
      if (s + 3 < nks) step(s + 3, 3, w1, w0);
    }
    wait_vm<0>();        // rejected requests behind the last stage write zeros into the ring: drain before re-using it
    if (dyn && wave == 0 && opaque(lane) == 0) {
      asm volatile("" : "+v"(drawn));
      *reinterpret_cast<volatile int*>(lds + K8_NEXT) = (int)gridDim.x + n_grp * drawn + grp;
    }
    barrier_raw();

    // ---- this tile's coordinates for the epilogue; then the next tile's operands are requested ----------------------------
    const int c_i0 = i0, c_j0 = j0, c_elast = e_last;
    // the stored activation's block exponent (derivative epilogues), loaded BEFORE the next tile's operands are requested (bsp_kc.hip)
    int eH = 0;
    {
      const kargs_t a = kargs();
      const int r0h = c_i0 + 128 * wm;
      if (AUX != AUX_NONE && c_j0 + wj0 < a->J && r0h < a->I) { eH = a->EH[(size_t)(r0h >> 7) * ncb_of(a->ldh) + ((a->h_col0 + c_j0 + wj0) >> 7)]; eH = __builtin_amdgcn_readfirstlane(eH); }
    }
    const int vbn = dyn ? __builtin_amdgcn_readfirstlane(*reinterpret_cast<volatile int*>(lds + K8_NEXT)) : vb + (int)gridDim.x;
    const bool more = vbn < ntiles;
    if (more) {
      prepare(vbn);
      heads();
      draw();
    }

    // ---- epilogue.  Lane l: point pt = l & 31 of each 32-point block mi; register r of block (mi, nj) is column
    //      64 wn + 32 nj + 16 (r >> 3) + 8 (l >> 5) + (r & 7) of the tile.  Ring slots 0 and 1 are being refilled; slots 2
    //      and 3 (8 KiB per wave) hold the waves' plane strips (results on their way out) and, for the derivative
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
    const int r0 = c_i0 + 128 * wm;                               // first row of the wave
    const int rbw = r0 >> 7;                                      // its 128-row block
    const int nrows = min(128, e->I - r0);                        // <= 0: the lower half of the last tile may lie beyond I
    const int jw = c_j0 + wj0;                                   // first column of the wave
    const bool wave_cols = jw < e->J && nrows > 0;
    const size_t offC = uniform_sz(((size_t)r0 * e->ldc + e->c_col0) * 4);
    const srd_t srdC = make_srd(e->C + offC, nrows > 0 ? clamp_bytes(((unsigned long long)(nrows - 1) * e->ldc + e->J) * 4ull) : 0u);
    // Plane strip of one (32-point block, 32-column half) = [32 points][128 B]: the two 64-byte groups [hi | lo] of the
    // half as they lie in memory, chunk c at position c ^ (point & 7).  A el writes its 16-byte pieces (eight consecutive
    // lanes: eight positions = all 32 banks), then the wave reads the strip back eight whole rows per instruction and stores
    // 8 x 128 contiguous bytes.
    char* const strip = lds + K8_XBUF + wave * 8192 + 4096;
    const unsigned sw_off = (unsigned)pt * 128u;
    const int srow = el >> 3, schunk = el & 7;
    const unsigned sr_off = (unsigned)srow * 128u + 16u * (unsigned)(schunk ^ srow);       // + 1024 per pass (8 rows: same swizzle)
    unsigned voC[2];                                                                       // per 32-column half (J % 16 == 0)
#pragma unroll
    for (int nj = 0; nj < 2; ++nj)
      voC[nj] = jw + 32 * nj + 16 * (schunk >> 2) < e->J ? (unsigned)srow * (unsigned)e->ldc * 4u + (unsigned)((jw >> 4) + 2 * nj) * 64u + 16u * (unsigned)schunk : OOBH;
    const unsigned stepC8 = 8u * (unsigned)e->ldc * 4u;
    auto strip_put = [&](int gg, const u32x4& hi, const u32x4& lo) {
      *reinterpret_cast<u32x4*>(strip + sw_off + 16 * ((4 * gg + lh) ^ (pt & 7))) = hi;
      *reinterpret_cast<u32x4*>(strip + sw_off + 16 * ((4 * gg + 2 + lh) ^ (pt & 7))) = lo;
    };
    // HAZARD (measured on gfx950, not modelled by hipcc): a ds_write_b128 can fetch its data registers AFTER a younger
    // ds_read_b128 of the same wave has returned into them.  The compiler, free to do so, gave the read-back of the strip the
    // registers of the planes it had just written; with the LDS busy (co-resident workgroup) single dwords of the written
    // planes then held the read-back's data -- wrong values in ~1 % of the rows, different from run to run.  The planes of a
    // half-block are therefore kept alive (keep_planes) until the read-back has been consumed.
    auto keep_planes = [](const u32x4 (&hi)[2], const u32x4 (&lo)[2]) { asm volatile("" ::"v"(hi[0]), "v"(lo[0]), "v"(hi[1]), "v"(lo[1])); };
    auto strip_flush = [&](int mi, int nj) {     // rows beyond I are rejected by the descriptor
#pragma unroll
      for (int ps = 0; ps < 4; ++ps) {
        const u32x4 d = *reinterpret_cast<const u32x4*>(strip + sr_off + 1024 * ps);
        __builtin_amdgcn_raw_buffer_store_b128(d, srdC, voC[nj], (unsigned)(4 * mi + ps) * stepC8, 0);
      }
    };
    float bj[4][8];
    if (BIAS) {
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
      const float su = inv_in * e->w0 * INV_PI;
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
        unsigned sw = 0u;
#pragma unroll
        for (int nj = 0; nj < 2; ++nj) {
          u32x4 phi[2], plo[2];
#pragma unroll
          for (int gg = 0; gg < 2; ++gg) {
            const int gq = 2 * nj + gg;
            float v[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) v[c] = fmaf(acc[mi][nj][8 * gg + c], su, bj[gq][c]);
            sinpi8<SIGNS, SINM>(v, sw);
            split8(v, 8192.f, phi[gg], plo[gg]);
            strip_put(gg, phi[gg], plo[gg]);
          }
          strip_flush(mi, nj);
          keep_planes(phi, plo);
        }
        if (SIGNS && e->Csign != nullptr && wave_cols && 32 * mi < nrows)
          e->Csign[((size_t)((r0 >> 5) + mi) * ((e->ldc + 63) >> 6) + ((e->c_col0 + jw) >> 6)) * 64 + el] = sw;
      }
      if ((wave & 1) == 0 && el == 0 && wave_cols) e->EC[(size_t)rbw * ncb_of(e->ldc) + ((e->c_col0 + jw) >> 7)] = 13;
    } else {
      // ---- two passes.  Pass A: final values in place of the accumulators, their |max|, column sums -- one 32-column half
      //      of the wave (nj) after the other, so that only 16 column sums are alive at a time.
      float wmax = 0.f;
      const float inv_h = pow2f(-eH);
      // Stored activations (derivative epilogues): half-block hb = (nj, mi) of the wave = 32 points x 128 B, fetched by LDS-DMA
      // in four 1 KiB pieces (8 whole half-rows each) into one of two 4 KiB buffers of the wave (inside ring slots 2, 3), half-block hb + 1 while hb is worked on.  Chunk c of point row q lies at position c ^ ((q >> 1) & 7)
      // (swizzle on the source address): the lanes of a ds_read_b128 group hold 16 points that differ in q & 15 -> 16
      // different 16-byte slots of the 256-byte bank row.  The sign words (one dword per lane and block mi) come the same
      // way, all four ahead of the first half-block.
      const size_t offH = uniform_sz(AUX != AUX_NONE ? ((size_t)r0 * e->ldh + e->h_col0) * 4 : 0);
      const srd_words srdH = make_srd_words(AUX != AUX_NONE ? e->H + offH : nullptr,
                                            (AUX != AUX_NONE && nrows > 0) ? clamp_bytes(((unsigned long long)(nrows - 1) * e->ldh + e->J) * 4ull) : 0u);
      const unsigned hbuf[2] = {(unsigned)__builtin_amdgcn_readfirstlane(lds_addr(lds + K8_XBUF + wave * 8192)),
                                (unsigned)__builtin_amdgcn_readfirstlane(lds_addr(lds + K8_XBUF + wave * 8192 + 4096))};
      const int hq = el >> 3;                                     // point row inside a piece
      auto dma_h = [&](int hb) {
        if (AUX == AUX_NONE) return;
        const int nj = hb >> 2, mi = hb & 3;
#pragma unroll
        for (int pc = 0; pc < 4; ++pc) {
          const int q = 8 * pc + hq;                              // point row inside the block
          const int c = (el & 7) ^ ((q >> 1) & 7);                // chunk of the half-row this lane fetches
          const bool ok = jw + 32 * nj + 16 * (c >> 2) < e->J;
          const unsigned vo = ok ? (unsigned)q * (unsigned)e->ldh * 4u + (unsigned)((jw >> 4) + 2 * nj) * 64u + 16u * (unsigned)c : OOBH;
          dma16_asm(srdH, hbuf[hb & 1] + (unsigned)(pc * 1024), vo, (unsigned)(32 * mi) * (unsigned)e->ldh * 4u);
        }
      };
      if (AUX == AUX_SINREC) {
        const srd_words srdS = make_srd_words(e->Hsign, clamp_bytes(sign_words((size_t)e->I, e->ldh) * 4ull));
        const unsigned sreg0 = __builtin_amdgcn_readfirstlane(lds_addr(lds + K8_HSIGN + wave * 1024));
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
          const bool ok = wave_cols && 32 * mi < nrows;
          const unsigned vo = ok ? (unsigned)((((size_t)((r0 >> 5) + mi) * ((e->ldh + 63) >> 6) + ((e->h_col0 + jw) >> 6)) * 64 + el) * 4) : OOBH;
          dma4_asm(srdS, sreg0 + (unsigned)(mi * 256), vo, 0u);
        }
      }
      dma_h(0);
      dma_h(1);
      // derivative epilogues: the accumulator's scale and |w0| in one factor; the sign bits are xor-ed with w0's own sign
      const unsigned w0mag = __float_as_uint(fabsf(e->w0) * inv_in);
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
          const float okf = (32 * mi + pt) < nrows ? 1.f : 0.f;      // points beyond I: out of the maximum and the column sums
          u32x4 hh[2], hl[2];
          if (AUX != AUX_NONE) {
            // half-block hb (and everything older: the sign words) is home when all but the four pieces of hb + 1 are
            if (hb == 7) wait_vm<0>(); else wait_vm<4>();
            const char* hreg = lds + K8_XBUF + wave * 8192 + (hb & 1) * 4096;
            if (AUX == AUX_SINREC && nj == 0) sword[mi] = *reinterpret_cast<const unsigned*>(lds + K8_HSIGN + wave * 1024 + mi * 256 + el * 4) ^ sflip;
#pragma unroll
            for (int gg = 0; gg < 2; ++gg) {
              hh[gg] = *reinterpret_cast<const u32x4*>(hreg + pt * 128 + 16 * ((4 * gg + lh) ^ ((pt >> 1) & 7)));
              hl[gg] = *reinterpret_cast<const u32x4*>(hreg + pt * 128 + 16 * ((4 * gg + 2 + lh) ^ ((pt >> 1) & 7)));
            }
            if (hb + 2 < 8) {   // the buffer is free once these reads have returned
              asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(hh[0]), "+v"(hh[1]), "+v"(hl[0]), "+v"(hl[1]), "+v"(sword[mi])::"memory");
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
              v[c] = AUX == AUX_SINREC ? x : (BIAS ? fmaf(x, inv_in, bj[gq][c]) : x * inv_in);
            }
            if (ACT == ACT_RELU) {
#pragma unroll
              for (int c = 0; c < 8; ++c) v[c] = fmaxf(v[c], 0.f);
            }
            if (AUX != AUX_NONE) {
              float h[8];
              join8(hh[gg], hl[gg], inv_h, h);
              if (AUX == AUX_SINREC) {
                // w0 cos(w0 z) = +-|w0| sqrt(1 - h^2): the sign bit (xor-ed with w0's own sign, once per word) is shifted to
                // bit 31 and merged over |w0| 2^-e by one v_bfi; 1 - h^2 is clamped at 0 by the FMA's output modifier
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                  float om;
                  asm("v_fma_f32 %0, -%1, %1, 1.0 clamp" : "=v"(om) : "v"(h[c]));
                  unsigned w0s_bits;
                  asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(w0s_bits) : "s"(0x7fffffffu), "v"(w0mag), "v"(sword[mi] << (31 - (16 * nj + 8 * gg + c))));
                  v[c] *= __uint_as_float(w0s_bits) * __builtin_amdgcn_sqrtf(om);
                }
              } else {
#pragma unroll
                for (int c = 0; c < 8; ++c) v[c] = h[c] > 0.f ? v[c] : 0.f;
              }
            }
#pragma unroll
            for (int c = 0; c < 8; ++c) {
              if (COLSUM) cs[8 * gg + c] = fmaf(v[c], okf, cs[8 * gg + c]);
              acc[mi][nj][8 * gg + c] = v[c];
            }
            if (jw + 16 * gq < e->J) wmax = fmaxf(wmax, okf * absmax3(v[6], v[7], absmax3(v[4], v[5], absmax3(v[2], v[3], absmax3(v[0], v[1], 0.f)))));
          }
        }
        if (COLSUM && e->colsum != nullptr && nrows > 0) {   // one partial row per 128-point tile
#pragma unroll
          for (int r = 0; r < 16; ++r) cs[r] = sum32(cs[r]);
          if ((el & 31) == 31) {
#pragma unroll
            for (int gg = 0; gg < 2; ++gg)
              if (jw + 16 * (2 * nj + gg) < e->J) {
                float* d = e->colsum + (size_t)rbw * e->ldcs + jw + 16 * (2 * nj + gg) + 8 * lh;
                *reinterpret_cast<float4*>(d) = make_float4(cs[8 * gg], cs[8 * gg + 1], cs[8 * gg + 2], cs[8 * gg + 3]);
                *reinterpret_cast<float4*>(d + 4) = make_float4(cs[8 * gg + 4], cs[8 * gg + 5], cs[8 * gg + 6], cs[8 * gg + 7]);
              }
          }
        }
      }
      // block maximum: waves 2 c and 2 c + 1 share the exponent block (ti, column block c of the tile)
      wmax = wave_max(wmax);
      if (el == 0) smax[wave] = wmax;
      __syncthreads();   // also: every wave has finished with the stored-activation buffers the strips share
      const float bmax = fmaxf(smax[wave & 6], smax[(wave & 6) + 1]);
      const int eC = exp_of_maxbits(__float_as_uint(bmax));
      const float sc = pow2f(eC);
      if ((wave & 1) == 0 && el == 0 && wave_cols) e->EC[(size_t)rbw * ncb_of(e->ldc) + ((e->c_col0 + jw) >> 7)] = eC;
      // ---- pass B: split, through the strip, store
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int nj = 0; nj < 2; ++nj) {
          u32x4 phi[2], plo[2];
#pragma unroll
          for (int gg = 0; gg < 2; ++gg) {
            float v[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) v[c] = acc[mi][nj][8 * gg + c];
            split8(v, sc, phi[gg], plo[gg]);
            strip_put(gg, phi[gg], plo[gg]);
          }
          strip_flush(mi, nj);
          keep_planes(phi, plo);
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
int prof_hook_begin(double flops, int variant, hipStream_t st);
void prof_hook_end(int token, hipStream_t st);

static int kc8_slots() {   // one 512-thread workgroup per CU; SNERF_KC_GRID=<n> (tests) forces a small grid
  static const int n = [] {
    if (const char* e = getenv("SNERF_KC_GRID")) { const int v = atoi(e); if (v > 0) return v; }
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    return cus;
  }();
  return n;
}

// `a` has passed check_kc (bsp_kc.hip: launch_kc)
int launch_kc8(KcArgs a, bool sin_hw, hipStream_t st) {
  a.tiles_i = (a.I + 255) / 256;
  a.tiles_j = (a.J + 255) / 256;
  const int ntiles = a.tiles_i * a.tiles_j, slots = kc8_slots();
  const dim3 grid(ntiles < slots ? ntiles : slots), block(512);
  const int tok = prof_hook_begin(2.0 * a.I * (double)a.J * a.K, 0, st);
  const bool cs = a.colsum != nullptr;
#define KC_LAUNCH(ACT_, AUX_, CS_, SG_, SM_) hipLaunchKernelGGL((gemm_kc8_kernel<ACT_, AUX_, CS_, SG_, SM_>), grid, block, 0, st, a)
  if (a.aux_mode == AUX_SINREC) KC_LAUNCH(ACT_NONE, AUX_SINREC, true, false, SIN_POLY);
  else if (a.aux_mode == AUX_RELU_MASK) KC_LAUNCH(ACT_NONE, AUX_RELU_MASK, true, false, SIN_POLY);
  else if (a.act == ACT_SIN) {
    if (a.Csign == nullptr) { if (sin_hw) KC_LAUNCH(ACT_SIN, AUX_NONE, false, false, SIN_HW); else KC_LAUNCH(ACT_SIN, AUX_NONE, false, false, SIN_POLY); }
    else { if (sin_hw) KC_LAUNCH(ACT_SIN, AUX_NONE, false, true, SIN_HW); else KC_LAUNCH(ACT_SIN, AUX_NONE, false, true, SIN_POLY); }
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

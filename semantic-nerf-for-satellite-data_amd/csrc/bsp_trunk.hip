// The SIREN trunk as ONE persistent launch, one-plane arithmetic (SNERF_FLAG_F16X1): a workgroup keeps the activations of its
// 128 points in LDS from the positional encoding to the last trunk layer, and only the weights stream (L2 -> registers).
// Reference: semantic/models/rs_semantic.py:325-334 (the trunk loop, skip concatenation at layer 4) and :63-78 (the chunk loop whose
// per-layer round trips through device memory this replaces) -- BASELINE.json north_star: "persistent fused MLP".
//
// Why one plane only: a 128-point x 512-column activation tile is 128 KiB as one fp16 plane and fits the CU's 160 KiB of LDS next to
// the 16 KiB encoding tile of the skip layer; with two planes it is 256 KiB, and a 64-point tile doubles the weight bytes every
// output row pulls through the L1 (the launch-per-layer kernel's limit).
//
//   * 512 threads, eight waves side by side: wave w owns ALL 128 points x output columns 64 w .. 64 w + 63 of every layer
//     (acc[4][2] of 32 x 32, the same accumulator map as gemm_kc_kernel, so every value is produced by the same MFMA sequence on
//     the same operands: the results equal the launch-per-layer path BIT FOR BIT -- tests/test_gpu_trunk.py).
//   * A operand = the activation tile in LDS, laid out as the k-loop reads it: stage S = columns 64 S .. 64 S + 63 of all 128 points
//     = [128 rows][128 B], 16-byte chunk c of a row at position c ^ ((row >> 1) & 7) (conflict-free ds_read_b128 / ds_write_b128).
//     Wave w's output columns ARE stage w of the next layer's operand: the epilogue writes its 16 KiB there, nobody else's.
//   * No LDS-DMA and no barrier inside a layer's k-loop; two barriers per layer around the in-place rewrite of the tile (everybody
//     has finished reading / the new tile is complete).  The weights come as in gemm_kc_kernel: each wave loads its own fragments,
//     two sub-steps ahead, asm statements with a hand-counted s_waitcnt vmcnt(4).
//   * Epilogue: x = acc (2^-e w0 / 2 pi) + b w0 / 2 pi, v_sin_f32 (bsp_kc_epi.h: sin2pi8), fp16, into the tile.  The last layer also
//     leaves through device memory (planes + exponent 13 for the feats / head launches that follow) and takes sigma's 1-wide
//     projection on the sine values while they are fp32 registers (the NDOT fold of gemm_kc_kernel, same partial-sum layout).
//     TRAIN: every layer leaves (planes, exponents, sign words of cos) for the backward pass -- the tile a wave has just written is
//     its own store strip.
#include "bsp_kc_epi.h"

namespace snerf {
namespace bsp {

constexpr int TR_A = 8 * 16384;                 // the activation tile: eight stages of [128 rows][128 B]
constexpr int TR_G = TR_A;                      // the encoding tile (one stage): layer 0's operand and the skip layer's first segment
constexpr int TR_BIAS = TR_G + 16384;           // 2 x 512 floats: the layer's bias times w0 / 2 pi, by layer parity
constexpr int TR_NDW = TR_BIAS + 2 * 2048;      // 512 floats: sigma's projection row
constexpr int TR_NEXT = TR_NDW + 2048;          // the workgroup's next tile
constexpr int TR_SMAX = TR_NEXT + 32;           // 8 floats: the waves' |max| of the feats layer (two waves share a 128-column exponent block)
constexpr int TR_EW = TR_NEXT + 64;             // the layers' weight exponents (read once: a global load in an epilogue would be awaited
                                                //  with vmcnt(0), i.e. behind the next layer's first weight requests)
constexpr int TR_LDS = TR_EW + 64;

typedef const __attribute__((address_space(4))) TrunkArgs* targs_t;
__device__ __forceinline__ targs_t targs() {
  targs_t q = (targs_t)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(q));
  return q;
}

// FEATS: the feats layer (no activation, block exponents from the block's own maximum) rides as one more layer behind the last SIREN
// layer -- inference passes only: together with the sign words of a training pass its two-pass epilogue does not fit the register file
// (445 spilled registers, scratch traffic inside the hand-counted k-loop).
// (TRUNK_DIAG_BUILD: tools/ablate/build_diag.sh only -- TrunkArgs::dbg 1: the weight descriptor has size zero, 2: the sine epilogue's
//  arithmetic and LDS writes are skipped, 4: no fragment reads inside the k-loop.  Timing only; the product build carries none of it.)
#ifdef TRUNK_DIAG_BUILD
#define TRUNK_DBG(bit) (dbg & (bit))
#else
#define TRUNK_DBG(bit) false
#endif
template <bool TRAIN, bool FEATS>
__global__ __launch_bounds__(512, 1) void trunk_kernel(const TrunkArgs) {
  const targs_t p = targs();
#ifdef TRUNK_DIAG_BUILD
  const int dbg = p->dbg;
#endif
  __shared__ __attribute__((aligned(16))) char lds[TR_LDS];
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int L = p->L, ntiles = (p->P + 127) >> 7;
  const int LT = (FEATS && p->F != nullptr) ? L + 1 : L;   // + the feats layer (entry L of the per-layer arrays): plain epilogue, leaves as planes [P][ldf]
  const unsigned skip_mask = p->skip_mask;
  auto opaque = [](int v) { asm volatile("" : "+v"(v)); return v; };

  // ---- per-lane constants ------------------------------------------------------------------------------------------------
  // fragment reads: lane -> (row l & 31 of the 32-point block, k half l >> 5); chunk (4 u + 2 pl + half) of sub-step u at
  // position chunk ^ ((row >> 1) & 7)
  const int rowl = lane & 31, kh = lane >> 5, swz = (rowl >> 1) & 7;
  unsigned fo[2][2];
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) fo[u][pl] = (unsigned)rowl * 128u + (unsigned)(((4 * u + 2 * pl + kh) ^ swz) << 4);
  const unsigned voW = 16u * (unsigned)lane;
  const unsigned w_u0 = 2u * (unsigned)wave;                      // the wave's first 32-row unit of every weight pack
  const unsigned w_rb32 = (unsigned)(p->W >> 5);
  const unsigned lds0 = (unsigned)__builtin_amdgcn_readfirstlane(lds_addr(lds));

  // ---- the encoding tile of tile `tile`: 16 pieces of 1 KiB (8 rows each), two per wave, by LDS-DMA --------------------------------
  auto dma_gamma = [&](int tile) {
    const targs_t a = targs();
    const int i0 = tile * 128, l = opaque(lane);
    const int rows = a->P - i0;                                   // > 0
    const srd_words srd = make_srd_words(a->pe + (size_t)i0 * 128, clamp_bytes((unsigned long long)(rows < 128 ? rows : 128) * 128ull));
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int row = 8 * (wave + 8 * q) + (l >> 3);
      const unsigned vo = (unsigned)row * 128u + 16u * (unsigned)((l & 7) ^ ((row >> 1) & 7));
      dma16_asm(srd, lds0 + (unsigned)(TR_G + (wave + 8 * q) * 1024), vo, 0u);
    }
  };

  // ---- weights: unit (16-k step ks, 32-row block rb) of a one-plane pack = 1 KiB [lane][16 B] at ((ks * w_rb32 + rb) * 1024) ------
  struct BFrag { u32x4 h[2], l[2]; };            // h: the first 16 k of the 32-deep sub-step (both 32-row blocks), l: the second
  srd_words srdW = make_srd_words(p->Wp[0], p->w_bytes[0]);
  int nks16 = p->K[0] >> 4;
  auto loadB2 = [&](int s, BFrag& b, int half) {
    const int s16 = 2 * s + half;
    const unsigned so = s16 < nks16 ? ((unsigned)s16 * w_rb32 + w_u0) * 1024u : OOB;
    if (half == 0)
      asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %2, %3, %4 offen\n\tbuffer_load_dwordx4 %1, %2, %3, %4 offen offset:1024"
                   : "=&v"(b.h[0]), "=&v"(b.h[1]) : "v"(voW), "s"(srdW), "s"(so));
    else
      asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %2, %3, %4 offen\n\tbuffer_load_dwordx4 %1, %2, %3, %4 offen offset:1024"
                   : "=&v"(b.l[0]), "=&v"(b.l[1]) : "v"(voW), "s"(srdW), "s"(so));
  };
  // everything but the four youngest requests (= the previous sub-step's) has landed
  auto wait_b = [&](BFrag& b) { asm volatile("s_waitcnt vmcnt(4)" : "+v"(b.h[0]), "+v"(b.l[0]), "+v"(b.h[1]), "+v"(b.l[1])::"memory"); };
  auto pin_b = [&](BFrag& b) { asm volatile("" : "+v"(b.h[0]), "+v"(b.l[0]), "+v"(b.h[1]), "+v"(b.l[1])::"memory"); };
  BFrag bq0, bq1, bq2;
  auto headW = [&](int l) {     // sub-steps 0 and 1 of layer l
    const targs_t a = targs();
    srdW = make_srd_words(a->Wp[l], TRUNK_DBG(1) ? 0u : a->w_bytes[l]);
    nks16 = a->K[l] >> 4;
    loadB2(0, bq0, 0); loadB2(0, bq0, 1); loadB2(1, bq1, 0); loadB2(1, bq1, 1);
  };

  // ---- once per workgroup: sigma's projection row, the first layer's bias, the first tile's encoding -----------------------------
  float* const sndw = reinterpret_cast<float*>(lds + TR_NDW);
  sndw[t] = p->nd_w != nullptr ? p->nd_w[t] : 0.f;
  reinterpret_cast<float*>(lds + TR_BIAS)[t] = p->bias[0][t] * (p->w0[0] * INV_2PI);
  if (t < LT) reinterpret_cast<int*>(lds + TR_EW)[t] = *p->EW[t];
  int bsel = 0;                                  // which bias buffer the current layer reads
  int* const tile_ctr = p->tile_ctr;
  const int n_grp = (gridDim.x & 7) == 0 ? 8 : 1, grp = (int)blockIdx.x & (n_grp - 1);
  const unsigned ctr_off = 4u * (unsigned)grp;
  int tile = blockIdx.x;
  dma_gamma(tile);
  headW(0);
  wait_vm<0>();
  barrier_raw();

  for (;;) {
    const int i0 = tile * 128;
    const int nrows = min(128, p->P - i0);
    // gamma's block exponent (13 by construction; honoured if it is not) -- awaited here, where only this tile's first weights are in flight
    int e_pe = p->Epe[tile];
    e_pe = __builtin_amdgcn_readfirstlane(e_pe);
    int next_tile = ntiles;

    for (int l = 0; l < LT; ++l) {
      const bool skip = (skip_mask >> l) & 1u;
      const int kofs = (l == 0 || skip) ? 2 : 0;         // sub-steps that read the encoding tile (64 columns)
      const int nks = p->K[l] >> 5;                      // 2 (layer 0), 16, or 18 (skip layer): even
      const bool last = l == L - 1;                      // the last SIREN layer: sigma's projection
      const bool isf = FEATS && l == L;                  // the feats layer
      const bool final = l == LT - 1;                    // the tile ends here
      float* const sbias = reinterpret_cast<float*>(lds + TR_BIAS + bsel * 2048);
      // where sub-step s reads: the encoding tile first (layer 0, skip layer), then the stages of the activation tile
      auto a_base = [&](int s) -> const char* { return s < kofs ? lds + TR_G : lds + ((s - kofs) >> 1) * 16384; };

      f32x16 acc[4][2];
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int nj = 0; nj < 2; ++nj)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[mi][nj][r] = 0.f;

      struct AFrag { f16x8 h[4], l[4]; };
      AFrag fa;
      // One 32-deep sub-step: 16 MFMAs on the fragments `fa` (read from LDS during the PREVIOUS sub-step) and the weight registers
      // `bc`; `bn` receives the weights of sub-step s + 2; every non-MFMA instruction sits in a gap between MFMAs.  Per accumulator
      // the products arrive in the order of gemm_kc_kernel<1, ...>: first the 16 k of the h registers, then the 16 k of the l registers.
      auto step = [&](int s, int u, BFrag& bc, BFrag& bn) {
        if (s >= 2) wait_b(bc);
        if (__builtin_expect(skip && s == 2 && e_pe != 13, 0)) {   // the skip layer's second segment carries exponent 13
#pragma unroll
          for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int nj = 0; nj < 2; ++nj) acc[mi][nj] = scale_acc(acc[mi][nj], 13 - e_pe);
        }
        const char* sn = a_base(s + 1 < nks ? s + 1 : s);        // where sub-step s + 1 reads (behind the last one: anything valid)
        const f16x8 bh0 = __builtin_bit_cast(f16x8, bc.h[0]), bh1 = __builtin_bit_cast(f16x8, bc.h[1]);
        const f16x8 bl0 = __builtin_bit_cast(f16x8, bc.l[0]), bl1 = __builtin_bit_cast(f16x8, bc.l[1]);
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
          acc[mi][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh0, fa.h[mi], acc[mi][0], 0, 0, 0);
          acc[mi][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh1, fa.h[mi], acc[mi][1], 0, 0, 0);
          if (mi < 2) loadB2(s + 2, bn, mi);
          const f16x8 nh = TRUNK_DBG(4) ? fa.h[mi] : ldsfrag(sn + 4096 * mi + fo[u ^ 1][0]);   // fa.h[mi] has issued its last MFMA
          acc[mi][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl0, fa.l[mi], acc[mi][0], 0, 0, 0);
          acc[mi][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl1, fa.l[mi], acc[mi][1], 0, 0, 0);
          fa.h[mi] = nh;
          if (!TRUNK_DBG(4)) fa.l[mi] = ldsfrag(sn + 4096 * mi + fo[u ^ 1][1]);
          __builtin_amdgcn_sched_barrier(0);
        }
      };
      // Layer start: W(0), W(1) were requested in the previous epilogue (or before the loop).  Behind them that epilogue issued the
      // 16 plane stores of a leaving layer (always) and a few small stores (sign words, sigma partials, the exponent: some are
      // skipped on ragged tiles) -- "all but the 16 youngest" therefore covers the weights whatever was skipped.
      if (TRAIN || l == 0) wait_vm<16>(); else wait_vm<0>();
      pin_b(bq0); pin_b(bq1);
      // The tile after this one: drawn from the counter of the workgroup's XCD group while layer 1's k-loop runs (a returning atomic
      // of ONE lane; the loop's counted waits retire it, it is read right behind the loop -- tests/test_build_cpu.py holds the
      // generated code to "nothing touches the register in between")
      int drawn = 0;
      if (l == 1 && wave == 0 && opaque(lane) == 0)
        asm volatile("s_nop 4\n\tglobal_atomic_add %0, %1, %2, %3 sc0" : "=v"(drawn) : "v"(ctr_off), "v"(1), "s"(tile_ctr) : "memory");
      // the NEXT layer's bias (behind the last layer: layer 0's, whatever tile follows): requested here, consumed right behind the
      // k-loop -- a compiler-visible load is awaited with vmcnt(0) wherever its value is used, and behind the loop nothing but the
      // rejected tail requests is in flight
      const int ln = final ? 0 : l + 1;
      const float bias_next = p->bias[ln][opaque(t)];
      {
        const char* s0 = a_base(0);
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) { fa.h[mi] = ldsfrag(s0 + 4096 * mi + fo[0][0]); fa.l[mi] = ldsfrag(s0 + 4096 * mi + fo[0][1]); }
      }
      for (int s = 0; s < nks; s += 6) {
        step(s, 0, bq0, bq2);
        step(s + 1, 1, bq1, bq0);
        if (s + 2 < nks) { step(s + 2, 0, bq2, bq1); step(s + 3, 1, bq0, bq2); }
        if (s + 4 < nks) { step(s + 4, 0, bq1, bq0); step(s + 5, 1, bq2, bq1); }
      }
      // ---- everybody has finished reading the tile --------------------------------------------------------------------------
      barrier_raw();
      const targs_t e = targs();
      const int el = opaque(lane);
      const int pt = el & 31, lh = el >> 5;
      if (l == 1 && wave == 0 && el == 0) {   // (retired by the counted waits of this layer's k-loop: 16 or 18 sub-steps)
        asm volatile("" : "+v"(drawn));
        *reinterpret_cast<volatile int*>(lds + TR_NEXT) = (int)gridDim.x + n_grp * drawn + grp;
      }
      if (l == 2) next_tile = __builtin_amdgcn_readfirstlane(*reinterpret_cast<volatile int*>(lds + TR_NEXT));   // published by layer 1's second barrier
      // the next tile's encoding, once the last layer that reads this tile's has finished (host: gamma_free_layer >= 2)
      const bool more = next_tile < ntiles;
      if (l == e->gamma_free_layer && more) dma_gamma(next_tile);
      reinterpret_cast<float*>(lds + TR_BIAS + (bsel ^ 1) * 2048)[opaque(t)] = ln == L ? bias_next : bias_next * (e->w0[ln] * INV_2PI);
      // the next layer's first weights (the next tile's first layer behind the last one)
      if (!final || more) headW(ln);

      // ---- epilogue: sine, fp16, into stage `wave` of the tile (rows 32 mi + pt, chunk 4 nj + 2 gg + lh) ---------------------------
      const int eWl = __builtin_amdgcn_readfirstlane(reinterpret_cast<const int*>(lds + TR_EW)[l]);
      const int e_in = 13 + eWl + ((l == 0) ? e_pe - 13 : 0);     // acc = true value * 2^e_in
      const float su = pow2f(-e_in) * e->w0[l] * INV_2PI;
      char* const stage = lds + wave * 16384;
      const bool leave = TRAIN || isf || (last && LT == L);           // this layer's planes go to device memory
      const int srow = el >> 3, schunk = el & 7;
      const int ldo = isf ? e->ldf : e->W;                             // columns of the tensor this layer leaves into
      const srd_t srdC = make_srd(leave ? (isf ? e->F : e->H[l]) + (size_t)i0 * (size_t)ldo * 2 : nullptr,
                                  leave ? clamp_bytes(((unsigned long long)(nrows - 1) * ldo + e->W) * 2ull) : 0u);
      const unsigned voC = (unsigned)srow * (unsigned)ldo * 2u + 128u * (unsigned)wave + 16u * (unsigned)schunk;
      const unsigned stepC8 = 16u * (unsigned)ldo;                     // eight rows of one plane
      // the block a wave has just written into its stage is its own store strip: whole 128-byte rows back out, 8 x 128 contiguous
      // bytes per store (the planes stay alive until the read-back has been consumed: the ds_write data hazard of bsp_kc.hip)
      auto flush = [&](int mi, const u32x4 (&ph)[4]) {
#pragma unroll
        for (int pp = 0; pp < 4; pp += 2) {
          const int r0 = 32 * mi + 8 * pp + srow, r1 = r0 + 8;
          const u32x4 d0 = *reinterpret_cast<const u32x4*>(stage + r0 * 128 + 16 * (schunk ^ ((r0 >> 1) & 7)));
          const u32x4 d1 = *reinterpret_cast<const u32x4*>(stage + r1 * 128 + 16 * (schunk ^ ((r1 >> 1) & 7)));
          __builtin_amdgcn_raw_buffer_store_b128(d0, srdC, voC, (unsigned)(4 * mi + pp) * stepC8, 2);
          store_data_guard(d0);
          __builtin_amdgcn_raw_buffer_store_b128(d1, srdC, voC, (unsigned)(4 * mi + pp + 1) * stepC8, 2);
          store_data_guard(d1);
        }
        asm volatile("" ::"v"(ph[0]), "v"(ph[1]), "v"(ph[2]), "v"(ph[3]));
      };
      if (isf) {
        // ---- feats (rs_semantic.py:338): no activation -- the plain two-pass epilogue of gemm_kc_kernel: values + bias and the block
        //      |max| (waves 2 c and 2 c + 1 share the exponent block of 128 columns), then fp16 at the block's own exponent
        const bool e_small = e_in >= -120 && e_in <= 120;
        if (!e_small) {
#pragma unroll
          for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int nj = 0; nj < 2; ++nj) acc[mi][nj] = scale_acc(acc[mi][nj], -e_in);
        }
        const float inv_in = e_small ? pow2f(-e_in) : 1.f;
        float wmax = 0.f;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
          for (int gq = 0; gq < 4; ++gq) {
            const int nj = gq >> 1, gg = gq & 1;
            const float4 b0 = *reinterpret_cast<const float4*>(&sbias[64 * wave + 16 * gq + 8 * lh]);
            const float4 b1 = *reinterpret_cast<const float4*>(&sbias[64 * wave + 16 * gq + 8 * lh + 4]);
            float v[8];
            v[0] = fmaf(acc[mi][nj][8 * gg + 0], inv_in, b0.x); v[1] = fmaf(acc[mi][nj][8 * gg + 1], inv_in, b0.y);
            v[2] = fmaf(acc[mi][nj][8 * gg + 2], inv_in, b0.z); v[3] = fmaf(acc[mi][nj][8 * gg + 3], inv_in, b0.w);
            v[4] = fmaf(acc[mi][nj][8 * gg + 4], inv_in, b1.x); v[5] = fmaf(acc[mi][nj][8 * gg + 5], inv_in, b1.y);
            v[6] = fmaf(acc[mi][nj][8 * gg + 6], inv_in, b1.z); v[7] = fmaf(acc[mi][nj][8 * gg + 7], inv_in, b1.w);
#pragma unroll
            for (int c = 0; c < 8; ++c) acc[mi][nj][8 * gg + c] = v[c];
            const float m8 = absmax3(v[6], v[7], absmax3(v[4], v[5], absmax3(v[2], v[3], absmax3(v[0], v[1], 0.f))));
            wmax = fmaxf(wmax, (32 * mi + pt) >= nrows ? 0.f : m8);   // (rows beyond P carry the bias: out of the maximum)
          }
        wmax = wave_max(wmax);
        float* const smax = reinterpret_cast<float*>(lds + TR_SMAX);
        if (el == 0) smax[wave] = wmax;
        barrier_raw();
        const float bmax = fmaxf(smax[wave & 6], smax[(wave & 6) + 1]);
        const int eC = exp_of_maxbits(__float_as_uint(bmax));
        const float sc = pow2f(eC);
        if ((wave & 1) == 0 && el == 0) e->EF[(size_t)tile * (size_t)ncb_of(e->ldf) + (wave >> 1)] = eC;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
          u32x4 ph[4];
          const unsigned rsw = (unsigned)(((32 * mi + pt) >> 1) & 7);
#pragma unroll
          for (int gq = 0; gq < 4; ++gq) {
            const int nj = gq >> 1, gg = gq & 1;
            float v[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) v[c] = acc[mi][nj][8 * gg + c];
            cvt8(v, sc, ph[gq]);
            *reinterpret_cast<u32x4*>(stage + (32 * mi + pt) * 128 + 16 * ((unsigned)(4 * nj + 2 * gg + lh) ^ rsw)) = ph[gq];
          }
          flush(mi, ph);
        }
      } else if (TRUNK_DBG(2)) {
        float keep = 0.f;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
          for (int nj = 0; nj < 2; ++nj) keep += acc[mi][nj][0];
        if (keep == 12345.678f) sndw[0] = keep;
      } else
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
        unsigned sw = 0u;
        float nd = 0.f;
        u32x4 ph[4];
        const unsigned rsw = (unsigned)(((32 * mi + pt) >> 1) & 7);
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          const int nj = gq >> 1, gg = gq & 1;
          float v[8];
          const float4 b0 = *reinterpret_cast<const float4*>(&sbias[64 * wave + 16 * gq + 8 * lh]);
          const float4 b1 = *reinterpret_cast<const float4*>(&sbias[64 * wave + 16 * gq + 8 * lh + 4]);
          v[0] = fmaf(acc[mi][nj][8 * gg + 0], su, b0.x); v[1] = fmaf(acc[mi][nj][8 * gg + 1], su, b0.y);
          v[2] = fmaf(acc[mi][nj][8 * gg + 2], su, b0.z); v[3] = fmaf(acc[mi][nj][8 * gg + 3], su, b0.w);
          v[4] = fmaf(acc[mi][nj][8 * gg + 4], su, b1.x); v[5] = fmaf(acc[mi][nj][8 * gg + 5], su, b1.y);
          v[6] = fmaf(acc[mi][nj][8 * gg + 6], su, b1.z); v[7] = fmaf(acc[mi][nj][8 * gg + 7], su, b1.w);
          sin2pi8<TRAIN, SIN_DIRECT>(v, sw);
          if (last && e->nd_out != nullptr) {
            const float4 w0v = *reinterpret_cast<const float4*>(sndw + 64 * wave + 16 * gq + 8 * lh);
            const float4 w1v = *reinterpret_cast<const float4*>(sndw + 64 * wave + 16 * gq + 8 * lh + 4);
            nd = fmaf(v[0], w0v.x, nd); nd = fmaf(v[1], w0v.y, nd); nd = fmaf(v[2], w0v.z, nd); nd = fmaf(v[3], w0v.w, nd);
            nd = fmaf(v[4], w1v.x, nd); nd = fmaf(v[5], w1v.y, nd); nd = fmaf(v[6], w1v.z, nd); nd = fmaf(v[7], w1v.w, nd);
          }
          cvt8(v, 8192.f, ph[gq]);
          *reinterpret_cast<u32x4*>(stage + (32 * mi + pt) * 128 + 16 * ((unsigned)(4 * nj + 2 * gg + lh) ^ rsw)) = ph[gq];
        }
        if (leave) flush(mi, ph);
        if (last && e->nd_out != nullptr) {   // the two lane halves hold the two column halves of every 16-column group of the same point
          const float tot = nd + __shfl_xor(nd, 32, 64);
          if (lh == 0 && 32 * mi + pt < nrows) e->nd_out[(size_t)wave * e->nd_stride + (size_t)(i0 + 32 * mi + pt)] = tot;
        }
        if (TRAIN && 32 * mi < nrows)
          e->Hsign[l][((size_t)((i0 >> 5) + mi) * (size_t)(e->W >> 6) + (size_t)wave) * 64 + el] = sw;
      }
      if (leave && !isf && (wave & 1) == 0 && el == 0) e->EH[l][(size_t)tile * (size_t)(e->W >> 7) + (wave >> 1)] = 13;
      bsel ^= 1;
      // ---- the new tile is complete --------------------------------------------------------------------------------------------
      barrier_raw();
      if (final) {
        if (!more) return;
        tile = next_tile;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------------------
int prof_hook_begin(double flops, int variant, hipStream_t st);
void prof_hook_end(int token, hipStream_t st);

static int g_trunk_grid_override = 0;
void trunk_set_grid_override(int n) { g_trunk_grid_override = n > 0 ? n : 0; }
static int g_trunk_fusion = 1;
void trunk_set_fusion(int on) { g_trunk_fusion = on ? 1 : 0; }
bool trunk_fusion_enabled() { return g_trunk_fusion != 0; }

static int bad_trunk(const char* why) {
  set_error("fused trunk: %s", why);
  return SNERF_ERR_BAD_DESC;
}

int launch_trunk(const TrunkArgs& a0, bool train, hipStream_t st) {
  TrunkArgs a = a0;
  if (!a.pe || !a.Epe || !a.tile_ctr || a.P <= 0) return bad_trunk("null operand / empty problem");
  if (a.W != 512 || a.L < 3 || a.L > TR_MAXL) return bad_trunk("W = 512, 3 <= L <= 8");
  if ((a.skip_mask & 1u) || (a.skip_mask >> (a.L - 1)) != 0u) return bad_trunk("skip layers: 0 < i < L - 1");
  if (a.P > 0x7FFFFF00) return bad_trunk("P");      // (every descriptor is built per 128-point tile: no 4 GiB limit on the tensors)
  int gfree = 0;
  for (int l = 0; l < a.L; ++l) {
    const bool skip = (a.skip_mask >> l) & 1u;
    const int want = l == 0 ? 64 : (skip ? 576 : 512);
    if (a.K[l] != want) return bad_trunk("layer widths: 64 (encoding), 512, 576 (skip)");
    if (!a.Wp[l] || !a.EW[l] || !a.bias[l] || a.w_bytes[l] != (unsigned)wp16_bytes(512, want, 1)) return bad_trunk("weight pack / bias of a layer");
    if (((uintptr_t)a.Wp[l] & 15) || ((uintptr_t)a.bias[l] & 15)) return bad_trunk("alignment");
    const bool leave = train || (l == a.L - 1 && a.F == nullptr);
    if (leave && (!a.H[l] || !a.EH[l] || ((uintptr_t)a.H[l] & 15))) return bad_trunk("output planes of a leaving layer");
    if (train && !a.Hsign[l]) return bad_trunk("sign words (training)");
    if (skip) gfree = l;
  }
  a.gamma_free_layer = gfree < 2 ? 2 : gfree;       // >= 2: the next tile's index is known from layer 2 on
  if (a.gamma_free_layer > a.L - 1) a.gamma_free_layer = a.L - 1;
  if (a.nd_out != nullptr && (!a.nd_w || a.nd_stride < (unsigned long long)a.P)) return bad_trunk("sigma projection");
  if (a.F != nullptr && train) return bad_trunk("the feats layer is fused in inference passes only");
  if (a.F != nullptr) {     // the feats layer rides as entry L
    const int l = a.L;
    if (!a.EF || a.ldf < 512 || (a.ldf & 15) || ((uintptr_t)a.F & 15)) return bad_trunk("feats output");
    if (a.K[l] != 512 || !a.Wp[l] || !a.EW[l] || !a.bias[l] || a.w_bytes[l] != (unsigned)wp16_bytes(512, 512, 1) || ((uintptr_t)a.Wp[l] & 15) || ((uintptr_t)a.bias[l] & 15))
      return bad_trunk("feats layer operands");
  }
  static const int cus = [] {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    return n;
  }();
#ifdef TRUNK_DIAG_BUILD
  { const char* ev = getenv("SNERF_TRUNK_DBG"); a.dbg = ev ? atoi(ev) : 0; }
#endif
  const int ntiles = (a.P + 127) / 128;
  const int slots = g_trunk_grid_override > 0 ? g_trunk_grid_override : cus;
  const dim3 grid(ntiles < slots ? ntiles : slots), block(512);
  double fl = 0;
  for (int l = 0; l < a.L + (a.F != nullptr ? 1 : 0); ++l) fl += 2.0 * a.P * 512.0 * a.K[l];
  const int tok = prof_hook_begin(fl, 1, st);      // variant 1 of SnerfProfile: the fused trunk
  if (train) hipLaunchKernelGGL((trunk_kernel<true, false>), grid, block, 0, st, a);
  else if (a.F != nullptr) hipLaunchKernelGGL((trunk_kernel<false, true>), grid, block, 0, st, a);
  else hipLaunchKernelGGL((trunk_kernel<false, false>), grid, block, 0, st, a);
  SNERF_LAUNCH_CHECK();
  prof_hook_end(tok, st);
  return SNERF_OK;
}

}  // namespace bsp
}  // namespace snerf

// GEMMs on block-scaled fp16-plane tensors (bsp.h) for gfx950, besides the wide K-contiguous kernel of bsp_kc.hip:
//
//   gemm_kcn_kernel  C = A W^T for the 32-wide head outputs (sigma, sun visibility, final head layers): fp32 out.
//   gemm_dw_kernel   dW = dZ^T X over all points, split-K slabs.  256 x 256 tile, eight waves of 128 x 64, one workgroup per CU
//                    (and a 32 x 256 form for the 32-wide heads).
//
// Operand staging is LDS-DMA only (buffer_load_dwordx4 ... lds): the planes are already what the matrix cores eat, so a
// k-step moves bytes and nothing else -- no VGPR staging, no conversion, no LDS stores.  Ring of three 16-deep stages;
// stage s + 2 is requested right after the barrier that publishes stage s, a counted s_waitcnt vmcnt(N) (never 0 in the
// loop) leaves it in flight across that barrier.  The LDS image is lane-linear as the DMA requires; bank-conflict
// swizzles are applied on the per-lane SOURCE address and again on the fragment read (same involution).
//   K-contiguous operands: [row][64 B] = chunks {hi k0-7, hi k8-15, lo k0-7, lo k8-15}, chunk position ^= (row >> 2) & 3:
//     every 16-lane group of the ds_read_b128 fragment read covers all 64 banks.
//   Point-contiguous operands (dW): [16 points][1 KiB] = 256 columns of one point, byte ^= ((p & 1) << 5) | ((p & 2) << 6):
//     the four point rows of a ds_read_b64_tr_b16 block land in four different bank quarters.
// Exponents: one per (128-row, 128-column) block of a tensor.  The accumulators carry the scale of the block being
// contracted; where it changes along k they are multiplied by the power of two (v_ldexp, exact).  In a SIREN forward
// pass every block of an activation has its maximum in [0.5, 1], so the branch is never taken.
#include "bsp_dev.h"

#include <vector>

namespace snerf {
namespace bsp {

// ------------------------------------------------------------------------------------------------------------------
// K-contiguous GEMM, 32-wide fp32 output (pre-activations of sigma / sun visibility / final head layers: the composite
// kernels apply their activations).  128 x 32 tile, four waves of 32 x 32; the weight fragments (one 32-row unit, the
// same for every wave) come straight from L2 as above.
// ------------------------------------------------------------------------------------------------------------------
constexpr int KC_STRIP = 32 * 68 * 4;                      // one wave's 32 x (64 + 4) fp32 transposition strip (dW epilogue)
constexpr int KN_A = 128 * 64, KN_RING = 4, KN_TAIL = KN_RING * KN_A, KN_LDS = KN_TAIL + 512;

// PL (planes, bsp.h): a step is 64 bytes of every A row = 16 k of two planes (three products) or 32 k of one plane (two MFMAs on
// two 16-k weight units).
template <int PL>
__global__ __launch_bounds__(256, 4) void gemm_kcn_kernel(const KcArgs p) {
  constexpr int EB = 2 * PL, KSUB = PL == 2 ? 16 : 32, KSH = PL == 2 ? 4 : 5;
  __shared__ __attribute__((aligned(16))) char lds[KN_LDS];
  int* etab = reinterpret_cast<int*>(lds + KN_TAIL);
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int ti = blockIdx.x, i0 = ti * 128, wi0 = wave * 32;
  const int nks16 = p.K >> 4;
  const int nks = (p.K + KSUB - 1) >> KSH, nks1 = nks;      // one segment (launch_kc_narrow); one plane: the last step may be half empty
  const srd_t srdA = make_srd(p.A + ((size_t)i0 * p.lda + p.a_col0) * EB,
                              clamp_bytes(i0 < p.I ? ((unsigned long long)(p.I - i0 - 1) * p.lda + p.Ka) * (unsigned long long)EB : 0ull));
  const srd_t srdW = make_srd(p.W, p.w_bytes);
  unsigned voA[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int row = 64 * q + (t >> 2);
    const unsigned c = (unsigned)((t & 3) ^ ((row >> 2) & 3));
    voA[q] = (i0 + row < p.I) ? (unsigned)row * (unsigned)p.lda * (unsigned)EB + 16u * c : OOB;
  }
  const unsigned w_u0 = (unsigned)(p.w_row0 >> 5), w_ks0 = (unsigned)(p.w_k0 >> 4);
  auto issueA = [&](int s, int slot) {
#pragma unroll
    for (int q = 0; q < 2; ++q) dma16(srdA, lds + slot * KN_A + 4096 * q + wave * 1024, s < nks ? voA[q] : OOB, (unsigned)s * 64u);
  };
  struct BFrag { u32x4 h, l; };
  auto loadB = [&](int s, BFrag& b) {
    if constexpr (PL == 2) {
      const unsigned so = ((w_ks0 + (unsigned)s) * (unsigned)p.w_rb32 + w_u0) * 2048u;
      const unsigned vo = s < nks ? 16u * (unsigned)lane : OOB;
      b.h = __builtin_amdgcn_raw_buffer_load_b128(srdW, vo, so, 0);
      b.l = __builtin_amdgcn_raw_buffer_load_b128(srdW, vo == OOB ? OOB : vo + 1024u, so, 0);
    } else {   // 1 KiB units: h = the first 16 k of the step, l = the second
      const unsigned so = ((w_ks0 + 2u * (unsigned)s) * (unsigned)p.w_rb32 + w_u0) * 1024u;
      b.h = __builtin_amdgcn_raw_buffer_load_b128(srdW, 2 * s < nks16 ? 16u * (unsigned)lane : OOB, so, 0);
      b.l = __builtin_amdgcn_raw_buffer_load_b128(srdW, 2 * s + 1 < nks16 ? 16u * (unsigned)lane : OOB, so + (unsigned)p.w_rb32 * 1024u, 0);
    }
  };
  const int sA = lane, sB = lane + 64;
  const int eA = sA < nks ? kc_exp_of_step(p, ti, sA, nks1, KSUB) : 0;
  const int eB = sB < nks ? kc_exp_of_step(p, ti, sB, nks1, KSUB) : 0;
  const int eAp = (sA > 0 && sA < nks) ? kc_exp_of_step(p, ti, sA - 1, nks1, KSUB) : eA;
  const int eBp = sB < nks ? kc_exp_of_step(p, ti, sB - 1, nks1, KSUB) : eB;
  const int e_last = kc_exp_of_step(p, ti, nks - 1, nks1, KSUB);
  issueA(0, 0);
  BFrag bq0, bq1;
  loadB(0, bq0);
  issueA(1, 1);
  if (wave == 0) { etab[sA] = eA; etab[sB] = eB; }
  const unsigned long long chg0 = __builtin_amdgcn_ballot_w64(eA != eAp), chg1 = __builtin_amdgcn_ballot_w64(eB != eBp);
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const int rowl = lane & 31, kh = lane >> 5, swz = (rowl >> 2) & 3;
  const unsigned fo0 = (unsigned)rowl * 64u + (unsigned)(((0 + kh) ^ swz) << 4);
  const unsigned fo1 = (unsigned)rowl * 64u + (unsigned)(((2 + kh) ^ swz) << 4);
  auto step = [&](int s, int slot, BFrag& bc, BFrag& bn) {
    wait_vm<2>();
    barrier_raw();
    if (__builtin_expect(((s < 64 ? chg0 >> s : chg1 >> (s - 64)) & 1ull) != 0ull, 0)) acc = scale_acc(acc, etab[s] - etab[s - 1]);
    loadB(s + 1, bn);
    issueA(s + 2, (slot + 2) % KN_RING);
    const char* st = lds + slot * KN_A;
    if constexpr (PL == 2) {
      acc = mfma3(ldsfrag(st + wi0 * 64 + fo0), ldsfrag(st + wi0 * 64 + fo1), __builtin_bit_cast(f16x8, bc.h), __builtin_bit_cast(f16x8, bc.l), acc);
    } else {
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ldsfrag(st + wi0 * 64 + fo0), __builtin_bit_cast(f16x8, bc.h), acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ldsfrag(st + wi0 * 64 + fo1), __builtin_bit_cast(f16x8, bc.l), acc, 0, 0, 0);
    }
  };
  for (int s = 0; s < nks; s += 4) {
    step(s, 0, bq0, bq1);
    if (s + 1 < nks) step(s + 1, 1, bq1, bq0);
    if (s + 2 < nks) step(s + 2, 2, bq0, bq1);
    if (s + 3 < nks) step(s + 3, 3, bq1, bq0);
  }
  wait_vm<0>();
  const int e_in = e_last + *p.EW;
  const int col = wf16_row(lane & 31);        // row order of the weight pack's 32-row unit
  const float bj = (p.bias != nullptr && col < p.J) ? p.bias[col] : 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = i0 + wi0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    if (row < p.I && col < p.J) p.Cf[(size_t)row * 32 + col] = __builtin_amdgcn_ldexpf(acc[r], -e_in) + bj;
  }
}

// ------------------------------------------------------------------------------------------------------------------
// dW = dZ^T X: both operands point-contiguous BSP, transposed fragment reads, split-K slabs in fp32
// ------------------------------------------------------------------------------------------------------------------
// TI = 256: 512 threads, eight waves of 128 x 64 (2 x 4), one workgroup per CU.  TI = 32: 256 threads, four waves of 32 x 64.
// PL (planes, bsp.h): one plane halves the bytes of a point row, so a stage of the SAME bytes is 32 points deep instead of 16 --
// the same DMA requests and fragment reads per stage, 16 MFMAs (two 16-point steps, one product) instead of 24.
template <int TI, int PL> struct DwCfg {
  static constexpr int NTH = TI == 256 ? 512 : 256, WAVES = NTH / 64;
  static constexpr int MI = TI == 256 ? 4 : 1;                   // 32-row blocks per wave along i
  static constexpr int EB = 2 * PL;                              // bytes per element
  static constexpr int PTS = PL == 2 ? 16 : 32;                  // points per stage
  static constexpr int SPC = 128 / PTS;                          // stages per 128-point exponent chunk
  static constexpr int A_PITCH = TI * EB;                        // bytes of one point row of the A stage (TI columns, all planes)
  static constexpr int B_PITCH = 256 * EB;
  static constexpr int A_BYTES = PTS * A_PITCH, B_BYTES = PTS * B_PITCH, STAGE = A_BYTES + B_BYTES, RING = 3;   // stages in the ring, RING - 1 in flight (four: measured equal, 326 vs 327 us)
  static constexpr int STRIPS = WAVES * KC_STRIP;
  static constexpr int TAIL = (RING * STAGE > STRIPS) ? RING * STAGE : STRIPS;
  static constexpr int MAXCH = 128;                              // 128-point chunks per split (k_split <= 16384)
  static constexpr int DUMMY = TAIL + WAVES * MAXCH * 4;        // scratch KiB per wave for rejected pieces (the narrow form)
  static constexpr int LDS = DUMMY + (TI == 256 ? 0 : WAVES * 1024);
};
// byte swizzle of point row p inside a stage: 1 KiB rows put the four rows of a transposed-read block into four bank
// quarters; the 128-byte rows of the 32-column form need only the hi/lo flip of the upper two rows
// (one plane: a 16-column group is 32 bytes, so the two column halves of a 32-lane read sit 32 bytes apart and the four point rows
//  of 512 bytes take the four 64-byte slots of the 256-byte bank row -- measured with (p & 3) << 5: 1.3e7 conflict cycles per launch;
//  64-byte rows need nothing)
template <int PITCH> __device__ __forceinline__ unsigned dw_swz(int p) {
  return PITCH == 1024 ? (unsigned)(((p & 1) << 5) | ((p & 2) << 6)) : (PITCH == 512 ? (unsigned)((p & 3) << 6) : (PITCH == 128 ? (unsigned)((p & 2) << 4) : 0u));
}

template <int TI, int PL>
__global__ __launch_bounds__(TI == 256 ? 512 : 256, 2) void gemm_dw_kernel(const DwArgs p) {
  using T = DwCfg<TI, PL>;
  constexpr int EB = T::EB, PTS = T::PTS;
  __shared__ __attribute__((aligned(16))) char lds[T::LDS];
  int* esum = reinterpret_cast<int*>(lds + T::TAIL);       // [wave][chunk]: exponent of dZ block + exponent of X block

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wi0 = TI == 256 ? (wave >> 2) * 128 : 0, wj0 = (wave & 3) * 64;
  int tile, split;
  split_tile_of_block(blockIdx.x, blockIdx.z, gridDim.x, gridDim.z, tile, split);
  const int ti = tile / p.tiles_j, tj = tile - ti * p.tiles_j;
  const int i0 = ti * TI, j0 = tj * 256;
  const int kBeg = split * p.k_split, kEnd = min(p.P, kBeg + p.k_split);
  const int nks = (kEnd - kBeg + PTS - 1) / PTS;
  float* C = p.C + (size_t)split * p.slab_stride;

  // ---- DMA sources: a piece is 1 KiB = one point row of the 256-column operand (eight 128-byte rows of the 32-column one)
  const srd_words srdA = make_srd_words(p.A + ((size_t)kBeg * p.lda + p.a_col0 + i0) * EB,
                                        clamp_bytes(kEnd > kBeg ? ((unsigned long long)(kEnd - kBeg) * p.lda - (p.a_col0 + i0)) * (unsigned long long)EB : 0ull));
  const srd_words srdB = make_srd_words(p.B + ((size_t)kBeg * p.ldb + p.b_col0 + j0) * EB,
                                        clamp_bytes(kEnd > kBeg ? ((unsigned long long)(kEnd - kBeg) * p.ldb - (p.b_col0 + j0)) * (unsigned long long)EB : 0ull));
  const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr(lds));
  constexpr int PA = T::A_PITCH, PB = T::B_PITCH;
  constexpr int A_PIECES = T::A_BYTES / 1024;                         // 16 (TI = 256) or 2 (TI = 32)
  constexpr int NPA = (A_PIECES + T::WAVES - 1) / T::WAVES;           // A pieces per wave: 2 / 1 (waves >= 2 of the narrow form: a rejected one)
  constexpr int NPB = 16 / T::WAVES;                                  // B pieces per wave: 2 / 4
  constexpr int RPA = PA >= 1024 ? 1 : 1024 / PA, RPB = 1024 / PB;    // point rows per 1 KiB piece
  constexpr int LRA = 64 / RPA, LRB = 64 / RPB;                       // lanes per point row
  unsigned voA[NPA], voB[NPB];
  int prA[NPA], prB[NPB];        // point row (inside the stage) each piece's lane belongs to
#pragma unroll
  for (int q = 0; q < NPA; ++q) {
    const int piece = wave * NPA + q;
    const int pr = piece * RPA + lane / LRA;
    const unsigned byte = 16u * (unsigned)(lane % LRA);
    prA[q] = piece < A_PIECES ? pr : -1;
    voA[q] = (unsigned)pr * (unsigned)p.lda * (unsigned)EB + (byte ^ dw_swz<PA>(pr));
  }
#pragma unroll
  for (int q = 0; q < NPB; ++q) {
    const int piece = wave * NPB + q;
    const int pr = piece * RPB + lane / LRB;
    prB[q] = pr;
    voB[q] = (unsigned)pr * (unsigned)p.ldb * (unsigned)EB + ((16u * (unsigned)(lane % LRB)) ^ dw_swz<PB>(pr));
  }
  auto issue = [&](int s, int slot) {
    const int prow0 = kBeg + PTS * s;
#pragma unroll
    for (int q = 0; q < NPA; ++q) {
      const int piece = wave * NPA + q;
      const bool ok = s < nks && prA[q] >= 0 && prow0 + prA[q] < kEnd;
      dma16_asm(srdA, lds0 + (unsigned)(piece < A_PIECES ? slot * T::STAGE + piece * 1024 : T::DUMMY + wave * 1024), ok ? voA[q] : OOB,
                (unsigned)s * (unsigned)PTS * (unsigned)p.lda * (unsigned)EB);
    }
#pragma unroll
    for (int q = 0; q < NPB; ++q) {
      const bool ok = s < nks && prow0 + prB[q] < kEnd;
      dma16_asm(srdB, lds0 + (unsigned)(slot * T::STAGE + T::A_BYTES + (wave * NPB + q) * 1024), ok ? voB[q] : OOB, (unsigned)s * (unsigned)PTS * (unsigned)p.ldb * (unsigned)EB);
    }
  };
  // ---- exponent sums per 128-point chunk, per wave (its 128-row / 64-column sub-tile lies in one block of either tensor)
  const int nch = (kEnd - kBeg + 127) >> 7;
  {
    const int ncbA = ncb_of(p.lda), ncbB = ncb_of(p.ldb);
    // sub-tiles entirely beyond I / J (their results are never stored) clamp to the last block of the table row
    const int cbA = min((p.a_col0 + i0 + wi0) >> 7, ncbA - 1), cbB = min((p.b_col0 + j0 + wj0) >> 7, ncbB - 1);
    for (int c = lane; c < nch; c += 64) {
      const size_t rb = (size_t)(kBeg >> 7) + c;
      esum[wave * T::MAXCH + c] = p.EA[rb * ncbA + cbA] + p.EB[rb * ncbB + cbB];
    }
  }

  issue(0, 0);
  issue(1, 1);
#pragma unroll
  for (int r = 2; r < T::RING - 1; ++r) issue(r, r);

  f32x16 acc[T::MI][2];
#pragma unroll
  for (int mi = 0; mi < T::MI; ++mi)
#pragma unroll
    for (int nj = 0; nj < 2; ++nj)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][nj][r] = 0.f;

  // transposed fragment reads: 16-lane group g: columns 16 (g & 1) ..+15 of a 32-column block, points 8 (g >> 1) + {0, 4} + q;
  // lane 4 q + pp of the group addresses point row q, columns 4 pp .. 4 pp + 3 (8 bytes)
  const int g = lane >> 4, q4 = (lane >> 2) & 3, pp = lane & 3;
  const int kq = 8 * (g >> 1) + q4;
  // `pl`: two planes: the plane (hi / lo of the same 16 points); one plane: the first / second 16 points of the 32-point stage
  auto frag_off = [&](int col, int pl, int pitch, unsigned sw) -> unsigned {   // col: first column of the 16-group (% 16 == 0)
    if constexpr (PL == 2) return (unsigned)kq * (unsigned)pitch + ((((unsigned)(col >> 4) * 64u) + (unsigned)pl * 32u + 8u * (unsigned)pp) ^ sw);
    else return (unsigned)(kq + 16 * pl) * (unsigned)pitch + ((((unsigned)(col >> 4) * 32u) + 8u * (unsigned)pp) ^ sw);
  };
  unsigned foA[T::MI][2], foB[2][2];
#pragma unroll
  for (int mi = 0; mi < T::MI; ++mi)
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) foA[mi][pl] = frag_off(wi0 + 32 * mi + 16 * (g & 1), pl, PA, dw_swz<PA>(q4));
#pragma unroll
  for (int nj = 0; nj < 2; ++nj)
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) foB[nj][pl] = (unsigned)T::A_BYTES + frag_off(wj0 + 32 * nj + 16 * (g & 1), pl, PB, dw_swz<PB>(q4));
  typedef __fp16 h4_t __attribute__((__vector_size__(4 * sizeof(__fp16))));
  typedef h4_t __attribute__((address_space(3))) * lds4_t;
  auto trfrag = [&](const char* base, unsigned off, int pitch) -> f16x8 {
    const f16x4 a = __builtin_bit_cast(f16x4, __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds4_t)(base + off)));
    const f16x4 b = __builtin_bit_cast(f16x4, __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds4_t)(base + off + 4 * pitch)));
    return f16x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  };

  int e_cur = 0;
  bool quiet = false;     // the current chunk is dropped (below)
  constexpr int NPIECE = NPA + NPB;
  // exponent bookkeeping at the start of a 128-point chunk: rescale the accumulators when the pair of exponents changes.
  // The sum of two exponents can move by more than the accumulators can follow (2^41 x 2^de must stay finite): a chunk more
  // than 2^64 quieter than the frame the accumulators are in adds nothing an fp32 sum could hold next to what is already
  // there -- its weight fragments are zeroed and the frame stays (a lone quiet chunk no longer ends in Inf / NaN gradients).
  auto mma = [](f16x8 ah, f16x8 al, f16x8 bh, f16x8 bl, f32x16 c) -> f32x16 {
    if constexpr (PL == 2) return mfma3(ah, al, bh, bl, c);
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c, 0, 0, 0);      // points 0-15 of the stage
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bl, c, 0, 0, 0);   // points 16-31
  };
  auto chunk = [&](int s) {
    if ((s & (T::SPC - 1)) != 0) return;
    const int e_new = __builtin_amdgcn_readfirstlane(esum[wave * T::MAXCH + s / T::SPC]);
    quiet = s != 0 && e_new - e_cur > 64;
    if (s == 0) e_cur = e_new;
    else if (__builtin_expect(e_new != e_cur && !quiet, 0)) {
      const int de = e_new - e_cur;
#pragma unroll
      for (int mi = 0; mi < T::MI; ++mi)
#pragma unroll
        for (int nj = 0; nj < 2; ++nj) acc[mi][nj] = scale_acc(acc[mi][nj], de);
      e_cur = e_new;
    }
  };
  // A wave whose 64 columns (or 128 rows) lie entirely beyond J (I) contracts zeros: the 64-column block of a skip / first layer
  // (J = 64: three of four column quarters), the 16 extras columns of the first head layer (J = 528: the third column tile).  It
  // still issues its share of the stage requests and keeps every barrier, but reads no fragments and issues no MFMAs -- the
  // matrix pipe and the LDS ports go to the co-resident waves (the other pass's launches under the two-stream schedule).
  const bool active = (j0 + wj0 < p.J) && (i0 + wi0 < p.I);
  if constexpr (TI == 256) {
    // Ping-pong: the eight waves form two groups (tile rows 0-127 / 128-255; one wave of each per SIMD) that run the same
    // two-phase step -- M: request stage s + 2, read the 24 fragments of stage s | C: 24 MFMAs on those registers -- one
    // phase apart, with a workgroup barrier after every phase.  While one wave of a SIMD issues its MFMAs back to back the
    // other does its LDS reads and DMA issue; in lockstep (all eight waves request, read, compute together) the three
    // costs add up: measured 336 us lockstep / 311 us ping-pong at 262,144 x 512 x 512 (round 2; three-stage ring).
    //   phase:    0      1      2      3     ...
    //   group 0:  M(0)   C(0)   M(1)   C(1)
    //   group 1:  -      M(0)   C(0)   M(1)
    // Stage s + 1 is awaited (own pieces, counted vmcnt) before the barrier that ends phase 2 s + 1, one barrier ahead of
    // its first reader (group 0, phase 2 s + 2); stage s + 2 is requested into the slot of stage s - 1 no earlier than
    // phase 2 s, one barrier after its last reader (group 1, phase 2 s - 1).  A request has two to three phases (~0.6 us
    // each at the headline shape) to land; a ring of four (four to five phases) measured the same: the kernel is bound by
    // the rate at which a CU's L1 fills (2.1 GB per launch through the L1s at 25 GB/s per CU), not by the latency of a fill.
    const int grp = wave >> 2;
    f16x8 fa_h[4], fa_l[4], fb_h[2], fb_l[2];
    auto phaseM = [&](int s, int slot) {
      issue(s + T::RING - 1, (slot + T::RING - 1) % T::RING);
      if (!active) return;
      chunk(s);
      const char* st = lds + slot * T::STAGE;
#pragma unroll
      for (int nj = 0; nj < 2; ++nj) { fb_h[nj] = trfrag(st, foB[nj][0], PB); fb_l[nj] = trfrag(st, foB[nj][1], PB); }
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) { fa_h[mi] = trfrag(st, foA[mi][0], PA); fa_l[mi] = trfrag(st, foA[mi][1], PA); }
      if (__builtin_expect(quiet, 0)) {
#pragma unroll
        for (int nj = 0; nj < 2; ++nj) { fb_h[nj] = f16x8{}; fb_l[nj] = f16x8{}; }
      }
    };
    auto phaseC = [&]() {
      if (!active) return;
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int nj = 0; nj < 2; ++nj) acc[mi][nj] = mma(fa_h[mi], fa_l[mi], fb_h[nj], fb_l[nj], acc[mi][nj]);
      __builtin_amdgcn_s_setprio(0);
    };
    wait_vm<(T::RING - 2) * NPIECE>();      // stage 0 (own pieces); the later ones stay in flight
    barrier_raw();
    if (grp == 0) {
      int slot = 0;
      for (int s = 0; s < nks; ++s) {
        phaseM(s, slot);
        barrier_raw();
        phaseC();
        wait_vm<(T::RING - 2) * NPIECE>();  // stage s + 1 landed (stage s + 2 in flight)
        barrier_raw();
        slot = slot == T::RING - 1 ? 0 : slot + 1;
      }
      barrier_raw();            // group 1's last compute phase
    } else {
      barrier_raw();            // group 0's first memory phase
      int slot = 0;
      for (int s = 0; s < nks; ++s) {
        phaseM(s, slot);
        wait_vm<(T::RING - 2) * NPIECE>();  // stage s + 1 landed (the stage s + 2 just requested stays in flight)
        barrier_raw();
        phaseC();
        barrier_raw();
        slot = slot == T::RING - 1 ? 0 : slot + 1;
      }
    }
  } else
  {
  auto step = [&](int s, int slot) {
    wait_vm<NPIECE>();
    barrier_raw();
    issue(s + T::RING - 1, (slot + T::RING - 1) % T::RING);
    chunk(s);
    const char* st = lds + slot * T::STAGE;
    f16x8 bh[2], bl[2];
#pragma unroll
    for (int nj = 0; nj < 2; ++nj) { bh[nj] = trfrag(st, foB[nj][0], PB); bl[nj] = trfrag(st, foB[nj][1], PB); }
    if (__builtin_expect(quiet, 0)) {
#pragma unroll
      for (int nj = 0; nj < 2; ++nj) { bh[nj] = f16x8{}; bl[nj] = f16x8{}; }
    }
#pragma unroll
    for (int mi = 0; mi < T::MI; ++mi) {
      const f16x8 ah = trfrag(st, foA[mi][0], PA), al = trfrag(st, foA[mi][1], PA);
#pragma unroll
      for (int nj = 0; nj < 2; ++nj) acc[mi][nj] = mma(ah, al, bh[nj], bl[nj], acc[mi][nj]);
    }
  };
  for (int s = 0; s < nks; s += 3) {
    step(s, 0);
    if (s + 1 < nks) step(s + 1, 1);
    if (s + 2 < nks) step(s + 2, 2);
  }
  }
  wait_vm<0>();
  barrier_raw();

  // ---- epilogue: fp32 slab rows through the wave's LDS strip (16-byte stores, 256 contiguous bytes per row)
  float* strip = reinterpret_cast<float*>(lds + wave * KC_STRIP);
  const int lc = lane & 31, lh = lane >> 5;
  const int rrow = lane >> 4, c4 = (lane & 15) * 4;
  const int col = j0 + wj0 + c4;
  const size_t offC = uniform_sz((size_t)(i0 + wi0) * p.ldc + j0 + wj0);
  const srd_t srdC = make_srd(C + offC, 0xFFFFFFE0u);
#pragma unroll
  for (int mi = 0; mi < T::MI; ++mi) {
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        strip[((r & 3) + 8 * (r >> 2) + 4 * lh) * 68 + 32 * n + lc] = __builtin_amdgcn_ldexpf(acc[mi][n][r], -e_cur);
#pragma unroll
    for (int ps = 0; ps < 8; ++ps) {
      const int rl = 32 * mi + rrow + 4 * ps;
      const bool ok = col < p.J && (i0 + wi0 + rl) < p.I;    // J % 4 == 0
      const float4 v = *reinterpret_cast<const float4*>(&strip[(rrow + 4 * ps) * 68 + c4]);
      const u32x4 d = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
      __builtin_amdgcn_raw_buffer_store_b128(d, srdC, ok ? ((unsigned)rl * (unsigned)p.ldc + (unsigned)c4) * 4u : OOB, 0, 2);
      store_data_guard(d);
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------------------
int prof_hook_begin(double flops, int variant, hipStream_t st);   // gemm.hip: per-launch HIP events when profiling is on
void prof_hook_end(int token, hipStream_t st);

static int bad(const char* why) {
  set_error("bsp gemm: %s", why);
  return SNERF_ERR_BAD_DESC;
}

int check_kc(const KcArgs& a, bool narrow) {
  if (!a.A || !a.EA || !a.W || !a.EW) return bad("null operand");
  if (a.pl != 1 && a.pl != 2) return bad("planes: 1 or 2");
  const int EB = 2 * a.pl, stage = a.pl == 2 ? 32 : 64;     // bytes per element; contraction depth of one LDS stage (128 B per row)
  if (a.I <= 0 || a.J <= 0 || a.K <= 0) return bad("empty problem");
  if ((a.K & 15) || (a.Ka & 15) || a.Ka <= 0 || a.Ka > a.K || a.K > 2048) return bad("K and Ka must be multiples of 16, K <= 2048");
  if ((a.lda & 15) || (a.a_col0 & 15) || a.a_col0 + a.Ka > a.lda) return bad("A segment does not fit its tensor (16-column groups)");
  if (a.Ka < a.K && (a.Ka % stage)) return bad("two A segments: the first must be a whole number of LDS stages (32 columns; one plane: 64)");
  if (a.pl == 1 && ((a.a_col0 & 31) || (a.Ka < a.K && (a.a2_col0 & 31)))) return bad("one plane: A segments start on a multiple of 32 columns");
  if (a.Ka < a.K && (!a.A2 || !a.EA2 || (a.lda2 & 15) || (a.a2_col0 & 15) || a.a2_col0 + (a.K - a.Ka) > a.lda2)) return bad("second A segment");
  if ((a.w_row0 & 31) || (a.w_k0 & 15) || a.w_rb32 <= 0) return bad("weight operand must start on a 32-row / 16-k unit");
  if ((size_t)128 * (a.lda > a.lda2 ? a.lda : a.lda2) * EB >= 0x7FFFFFFFull) return bad("leading dimension too large");
  if (((uintptr_t)a.A & 15) || ((uintptr_t)a.W & 15) || (a.A2 && ((uintptr_t)a.A2 & 15))) return bad("operands must be 16-byte aligned");
  if (narrow) {
    if (!a.Cf || a.J > 32 || a.Ka != a.K) return bad("narrow variant: fp32 output, J <= 32, one A segment");
    return SNERF_OK;
  }
  if (!a.C || !a.EC || (a.ldc & 15) || (a.c_col0 & 127) || (a.J & 15) || a.c_col0 + a.J > a.ldc) return bad("BSP output: ldc % 16, c_col0 % 128, J % 16");
  if (((uintptr_t)a.C & 15) || (a.bias && ((uintptr_t)a.bias & 15))) return bad("output / bias alignment");
  if (a.aux_mode != AUX_NONE) {
    if (a.aux_mode != AUX_SINREC && a.aux_mode != AUX_RELU_MASK) return bad("unsupported aux mode");
    if (!a.H || !a.EH || (a.ldh & 15) || (a.h_col0 & 127) || a.h_col0 + a.J > a.ldh) return bad("aux tensor: ldh % 16, h_col0 % 128");
    if (a.aux_mode == AUX_SINREC && !a.Hsign) return bad("AUX_SINREC needs the sign words");
    if (a.act != ACT_NONE) return bad("activation and derivative in one epilogue");
  }
  if (a.colsum && (((uintptr_t)a.colsum & 15) || (a.ldcs & 3))) return bad("colsum alignment");
  if (a.nd_w != nullptr) {
    if (a.act != ACT_SIN || a.aux_mode != AUX_NONE || !a.nd_out || (a.J & 255) || a.J > 2048 || a.nd_stride < (unsigned long long)a.I || ((uintptr_t)a.nd_w & 15))
      return bad("folded projection: ACT_SIN forward launch of whole 256-column tiles, J <= 2048, nd_out [tiles_j * 4 * nd_omax][nd_stride >= I]");
  }
  if ((size_t)128 * a.ldc * EB >= 0x7FFFFFFFull) return bad("ldc too large");
  return SNERF_OK;
}

int launch_kc_narrow(const KcArgs& a0, hipStream_t st) {
  KcArgs a = a0;
  if (a.Ka == 0) a.Ka = a.K;
  a.A2 = a.A; a.EA2 = a.EA; a.lda2 = a.lda; a.a2_col0 = a.a_col0;
  int rc = check_kc(a, true);
  if (rc) return rc;
  const int tok = prof_hook_begin(2.0 * a.I * 32.0 * a.K, 3, st);
  if (a.pl == 2) hipLaunchKernelGGL(gemm_kcn_kernel<2>, dim3((a.I + 127) / 128), dim3(256), 0, st, a);
  else hipLaunchKernelGGL(gemm_kcn_kernel<1>, dim3((a.I + 127) / 128), dim3(256), 0, st, a);
  SNERF_LAUNCH_CHECK();
  prof_hook_end(tok, st);
  return SNERF_OK;
}

int launch_dw(const DwArgs& a0, bool narrow_i, hipStream_t st) {
  DwArgs a = a0;
  if (!a.A || !a.EA || !a.B || !a.EB || !a.C) return bad("dW: null operand");
  if (a.I <= 0 || a.J <= 0 || a.P <= 0) return bad("dW: empty problem");
  if ((a.lda & 15) || (a.ldb & 15) || (a.a_col0 & 15) || (a.b_col0 & 15) || (a.J & 3) || (a.ldc & 3)) return bad("dW: leading dimensions / column offsets");
  if (a.k_split <= 0 || (a.k_split & 127) || a.k_split > 16384 || a.n_split < 1) return bad("dW: k_split must be a multiple of 128, <= 16384");
  if (a.pl != 1 && a.pl != 2) return bad("dW: planes: 1 or 2");
  if ((size_t)a.k_split * (a.lda > a.ldb ? a.lda : a.ldb) * 2 * a.pl >= 0xFFFFFFF0ull) return bad("dW: k_split * ld exceeds the 32-bit span");
  if (narrow_i ? (a.I > 32 || ((a.a_col0 & 127) + 32 > 128)) : ((a.a_col0 & 127) != 0)) return bad("dW: the A columns of a wave must lie in one exponent block");
  if (a.b_col0 & 63) return bad("dW: b_col0 % 64");
  if (((uintptr_t)a.A & 15) || ((uintptr_t)a.B & 15) || ((uintptr_t)a.C & 15)) return bad("dW: alignment");
  const int TI = narrow_i ? 32 : 256;
  a.tiles_i = (a.I + TI - 1) / TI;
  a.tiles_j = (a.J + 255) / 256;
  const dim3 grid(a.tiles_i * a.tiles_j, 1, a.n_split);
  const int tok = prof_hook_begin(2.0 * a.I * (double)a.J * a.P, narrow_i ? 3 : 2, st);
  if (narrow_i) { if (a.pl == 2) hipLaunchKernelGGL((gemm_dw_kernel<32, 2>), grid, dim3(256), 0, st, a); else hipLaunchKernelGGL((gemm_dw_kernel<32, 1>), grid, dim3(256), 0, st, a); }
  else { if (a.pl == 2) hipLaunchKernelGGL((gemm_dw_kernel<256, 2>), grid, dim3(512), 0, st, a); else hipLaunchKernelGGL((gemm_dw_kernel<256, 1>), grid, dim3(512), 0, st, a); }
  SNERF_LAUNCH_CHECK();
  prof_hook_end(tok, st);
  return SNERF_OK;
}

}  // namespace bsp
}  // namespace snerf

// GEMMs on block-scaled fp16-plane tensors (bsp.h) for gfx950: the dense layers of the default arithmetic.
//
//   gemm_kc_kernel   C = epilogue(A W^T): forward layers and dX.  128 x 256 tile, four waves of 64 x 128, two workgroups per CU.
//   gemm_kcn_kernel  the same contraction for the 32-wide head outputs (sigma, sun visibility, final head layers): fp32 out.
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
#include "bsp.h"
#include "gemm_common.h"

#include <vector>

namespace snerf {
namespace bsp {

typedef __attribute__((address_space(3))) void* lds_ptr_t;

__device__ __forceinline__ void dma16(srd_t srd, char* lds_dst, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (lds_ptr_t)lds_dst, 16, voff, soff, 0, 0);
}
// The same request as an asm statement the compiler cannot see into.  hipcc models the builtin form as a store to LDS and,
// where its alias analysis cannot separate the destination from a following LDS read (the dW kernel's transposed reads),
// puts s_waitcnt vmcnt(0) between them -- every stage request is then drained right after it is issued and the whole
// DMA latency sits on the critical path of every k-step (measured: 452 -> see DESIGN for the dW launch).  Completion is
// tracked by the kernels' own counted waits either way.  M0 (the LDS destination) is saved and restored inside.
typedef unsigned int srd_words __attribute__((ext_vector_type(4)));
__device__ __forceinline__ srd_words make_srd_words(const void* p, unsigned bytes) {
  const unsigned long long a = reinterpret_cast<unsigned long long>(p);
  return srd_words{(unsigned)__builtin_amdgcn_readfirstlane((unsigned)a), (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xFFFFu),
                   (unsigned)__builtin_amdgcn_readfirstlane(bytes), 0x00020000u};
}
__device__ __forceinline__ void dma16_asm(srd_words srd, unsigned lds_byte_addr, unsigned voff, unsigned soff) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "s"(lds_byte_addr), "v"(voff), "s"(srd), "s"(soff));
}
__device__ __forceinline__ unsigned lds_addr(const char* p) { return (unsigned)(unsigned long long)(lds_ptr_t)const_cast<char*>(p); }
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void barrier_raw() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
__device__ __forceinline__ f32x16 scale_acc(f32x16 c, int de) {
#pragma unroll
  for (int r = 0; r < 16; ++r) c[r] = __builtin_amdgcn_ldexpf(c[r], de);
  return c;
}
__device__ __forceinline__ f16x8 ldsfrag(const char* p) { return *reinterpret_cast<const f16x8*>(p); }

// three fp16 products per fp32 product, smallest terms first
__device__ __forceinline__ f32x16 mfma3(f16x8 ah, f16x8 al, f16x8 bh, f16x8 bl, f32x16 c) {
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c, 0, 0, 0);
  return c;
}

// |max| folded two values at a time (v_max3_f32 with |.| source modifiers; a plain fmaxf chain spends a canonicalising
// v_max per input under IEEE rules, 2 instructions per value)
__device__ __forceinline__ float absmax3(float a, float b, float m) {
  float r;
  asm("v_max3_f32 %0, |%1|, |%2|, %3" : "=v"(r) : "v"(a), "v"(b), "v"(m));
  return r;
}

// Eight sines sin(v_c) in place + bit c of the returned byte = cos(v_c) < 0.  Same reduction and polynomial as
// sin4_signcos (common.h) in fewer instructions, for a wave whose epilogue is issue-bound (one instruction per four
// cycles): k = round(x / pi) comes out of the low mantissa bits of x / pi + 1.5 * 2^23 (no v_rndne / v_cvt), the sign of
// the sine is bit 0 of k shifted onto the result's sign bit by one v_lshl_add, and the cosine's sign -- (k odd) xor
// (|r| > pi / 2), the second only at the rounding edge of the reduction -- is the top bit of (k << 31) + bits(pi/2 - |r|),
// shifted into the byte by v_alignbit.  Arguments beyond 30000 (never with sane data) take sin4_signcos's exact path.
template <bool SIGNS>
__device__ __forceinline__ unsigned sin8_signbits(float (&v)[8]) {
  float (&x)[8] = v;
  const float m = absmax3(x[6], x[7], absmax3(x[4], x[5], absmax3(x[2], x[3], absmax3(x[0], x[1], 0.f))));
  if (__builtin_expect(__builtin_amdgcn_ballot_w64(!(m <= 30000.f)) != 0ull, 0)) {
    unsigned n0, n1;
    const float4 s0 = sin4_signcos(make_float4(x[0], x[1], x[2], x[3]), &n0);
    const float4 s1 = sin4_signcos(make_float4(x[4], x[5], x[6], x[7]), &n1);
    v[0] = s0.x; v[1] = s0.y; v[2] = s0.z; v[3] = s0.w; v[4] = s1.x; v[5] = s1.y; v[6] = s1.z; v[7] = s1.w;
    return n0 | (n1 << 4);
  }
  // arithmetic first, as plain loops the compiler packs two elements per v_pk_fma_f32; the sign work (asm) afterwards
  float t[8], r[8], sv[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) t[c] = fmaf(x[c], 0.31830988618379067154f, 12582912.f);   // low mantissa bits = k
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const float kf = t[c] - 12582912.f;
    float q = fmaf(kf, -3.140625f, x[c]);
    q = fmaf(kf, -9.67502593994140625e-4f, q);
    r[c] = fmaf(kf, -1.509958025280866e-07f, q);   // |k| < 2^14: the next term of pi (3.4e-15 k) is below 1e-10
  }
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const float r2 = r[c] * r[c];
    float q = fmaf(r2, -2.5028294103890403e-08f, 2.755689592959243e-06f);
    q = fmaf(q, r2, -0.00019841265748254955f);
    q = fmaf(q, r2, 0.008333333767950535f);
    q = fmaf(q, r2, -0.1666666716337204f);
    sv[c] = fmaf(r[c] * r2, q, r[c]);
  }
  unsigned byte = 0u;
#pragma unroll
  for (int c = 7; c >= 0; --c) {
    const unsigned kb = __float_as_uint(t[c]);
    v[c] = __uint_as_float((kb << 31) + __float_as_uint(sv[c]));
    if (SIGNS) {
      const float edge = 1.57079637f - fabsf(r[c]);
      byte = __builtin_amdgcn_alignbit(byte, (kb << 31) + __float_as_uint(edge), 31);
    }
  }
  return byte;
}

// exponent of k-step s of a K-contiguous A operand made of one or two segments (any lane; uniform inputs)
__device__ __forceinline__ int kc_exp_of_step(const KcArgs& p, int rb, int s, int nks1) {
  const bool seg2 = s >= nks1;   // branch-free: one load through a selected pointer
  const int* E = seg2 ? p.EA2 : p.EA;
  const int ld = seg2 ? p.lda2 : p.lda, col = seg2 ? p.a2_col0 + 16 * (s - nks1) : p.a_col0 + 16 * s;
  return E[(size_t)rb * ncb_of(ld) + (col >> 7)];
}

// ------------------------------------------------------------------------------------------------------------------
// K-contiguous GEMM, BSP output
// ------------------------------------------------------------------------------------------------------------------
// 128 x 256 tile, four waves side by side: wave w owns ALL 128 rows and columns 64 w .. 64 w + 63 (acc[4][2]).
//   A (activations, streamed once from HBM): LDS-DMA ring of 16-deep stages, read by every wave (8 ds_read_b128 / step).
//   W (weights, L2-resident, packed in fragment order "WF16"): each wave loads ITS OWN four fragments of the next step
//     straight into registers (4 x buffer_load_dwordx4, 1 KiB contiguous each) -- no LDS round trip, no DMA, no sharing
//     needed because the waves split the columns.
// Measured on the first form of this kernel (both operands through LDS, waves 2 x 2): the LDS array was busy 66 % of the
// time (fragment reads + DMA fills, 144 KB per step of two co-resident workgroups) and the six DMA pieces per wave and
// step cost the issuing wave ~80 cycles each (ablation builds of tools/ablate: 72 of 460 us per launch); the weight
// tiles were 2/3 of both.  This form moves 80 KB per step through LDS and issues two pieces per wave.
constexpr int KC_A = 128 * 128, KC_RING = 3;               // one stage: 128 rows x 32 k (two 16-column groups, 128 B per row)
constexpr int KC_STRIP = 32 * 68 * 4;                      // one wave's 32 x (64 + 4) fp32 transposition strip
constexpr int KC_RINGB = KC_RING * KC_A;                   // 48 KiB
constexpr int KC_TAIL = (KC_RINGB > 4 * KC_STRIP) ? KC_RINGB : 4 * KC_STRIP;   // small tables behind ring / strips
#ifdef BSP_ABL_ONEWG
constexpr int KC_LDS = KC_TAIL + 128 * 4 + 64 + 64 * 1024;   // one workgroup per CU (diagnostics)
#else
constexpr int KC_LDS = KC_TAIL + 128 * 4 + 64;
#endif

// SIGNS (ACT_SIN): also produce the sign-of-cos words (training); the forward-only passes skip that arithmetic
// BSP_ABL_PAIR (ablation build): ONE workgroup of eight waves runs TWO row-adjacent 128 x 256 tiles (waves 0-3 / 4-7, one
// LDS ring each) behind common barriers, so that the two waves of a SIMD stay in the same phase (loop beside loop,
// epilogue beside epilogue) instead of drifting through every phase pairing.  Measured 459 / 515 us against 407 / 428 for
// two independent workgroups: what the product gains from co-residency is exactly the loop-beside-epilogue pairing.
#ifdef BSP_ABL_PAIR
constexpr int KC_HALVES = 2;
#else
constexpr int KC_HALVES = 1;
#endif
constexpr int KC_LDS_R = (KC_LDS + 255) & ~255;
template <int ACT, int AUX, bool COLSUM, bool SIGNS = true>
__global__ __launch_bounds__(256 * KC_HALVES, KC_HALVES == 2 ? 1 : 2) void gemm_kc_kernel(const KcArgs p) {
  __shared__ __attribute__((aligned(16))) char lds_all[KC_HALVES * KC_LDS_R];
  const int t = threadIdx.x, lane = t & 63;
  const int half = KC_HALVES == 2 ? __builtin_amdgcn_readfirstlane(t >> 8) : 0;
  char* const lds = lds_all + half * KC_LDS_R;
  int* etab = reinterpret_cast<int*>(lds + KC_TAIL);
  float* smax = reinterpret_cast<float*>(lds + KC_TAIL + 512);

  const int wave = __builtin_amdgcn_readfirstlane((t >> 6) & 3);
  const int wj0 = wave * 64;
  int ti, tj;
  tile_of_block(blockIdx.x, p.tiles_i, p.tiles_j, ti, tj);   // tiles_i counts row-tile PAIRS in the paired build
  ti = KC_HALVES * ti + half;
  const int ti_e = min(ti, (p.I + 127) / 128 - 1);            // exponent-table row of a tile beyond I (its results are never stored)
  const int i0 = ti * 128, j0 = tj * 256;
  const int nks = p.K >> 4, nks1 = p.Ka >> 4;
#ifdef BSP_ABL_STAMP
  const unsigned long long st0 = __builtin_amdgcn_s_memrealtime();
  const unsigned long long sc0 = __builtin_amdgcn_s_memtime();
#endif

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
  int sbias = 0;
  const int seg_switch = p.Ka < p.K ? nst1 : 0x7fffffff;
  auto enter_stage = [&](int S) {     // before the first piece of stage S
    if (__builtin_expect(S == seg_switch, 0)) {
      srdCur = srdA2; sbias = nst1;
#pragma unroll
      for (int q = 0; q < 4; ++q) voCur[q] = voA2[q];
    }
  };
  auto issueA = [&](int S, int slot, int q) {
    dma16(srdCur, dst0 + slot * KC_A + 4096 * q, voCur[q], S < nst ? (unsigned)(S - sbias) * 128u : OOB);
  };
  // ---- W: fragment-ordered pack; unit (ks, rb32) = 2 KiB = [plane][lane][16 B]; this wave reads units rb32 = u0, u0 + 1
  const srd_words srdW = make_srd_words(p.W, p.w_bytes);
  const unsigned w_u0 = (unsigned)((p.w_row0 + j0 + wj0) >> 5), w_ks0 = (unsigned)(p.w_k0 >> 4);
  const unsigned voW = 16u * (unsigned)lane;
  struct BFrag { u32x4 h[2], l[2]; };
  // The weight loads are asm statements with their completion counted by hand (wait_b below).  As builtins, hipcc's own
  // vm-counter bookkeeping put conservative waits behind them at the loop header (s_waitcnt vmcnt(6) ahead of the first
  // MFMA: the fragments requested ONE step earlier had to be home, i.e. the two-step prefetch was undone) and re-used the
  // registers of the fragment set that is dead at the header as VALU temporaries, each with a WAW wait on a load.
  // A step beyond K is rejected through the scalar offset.
  auto loadB = [&](int s, BFrag& b) {
    const unsigned so = s < nks ? ((w_ks0 + (unsigned)s) * (unsigned)p.w_rb32 + w_u0) * 2048u : OOB;
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(b.h[0]) : "v"(voW), "s"(srdW), "s"(so));
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:1024" : "=v"(b.l[0]) : "v"(voW), "s"(srdW), "s"(so));
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:2048" : "=v"(b.h[1]) : "v"(voW), "s"(srdW), "s"(so));
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:3072" : "=v"(b.l[1]) : "v"(voW), "s"(srdW), "s"(so));
  };
  // everything but the six youngest requests (= the next sub-step's) has landed; names the fragments so that no use of them
  // can be scheduled above the wait
  auto wait_b = [&](BFrag& b) {
    asm volatile("s_waitcnt vmcnt(6)" : "+v"(b.h[0]), "+v"(b.l[0]), "+v"(b.h[1]), "+v"(b.l[1])::"memory");
  };

  // ---- exponents along k: table in LDS + a bit per step where the scale changes (128 steps at most: K <= 2048).
  // The loads go out first and are consumed behind the first stage requests (their wait then counts past the DMA).
  const int sA = lane, sB = lane + 64;
  const int eA = sA < nks ? kc_exp_of_step(p, ti_e, sA, nks1) : 0;
  const int eB = sB < nks ? kc_exp_of_step(p, ti_e, sB, nks1) : 0;
  const int eAp = (sA > 0 && sA < nks) ? kc_exp_of_step(p, ti_e, sA - 1, nks1) : eA;
  const int eBp = sB < nks ? kc_exp_of_step(p, ti_e, sB - 1, nks1) : eB;
  const int e_last = kc_exp_of_step(p, ti_e, nks - 1, nks1);
  // Both operands are requested ahead: W two sub-steps, A two stages.  vm-counter order: [W(0) A(0) x 4] [W(1) A(1) x 4],
  // then per sub-step s = 2 S + u: [W(s + 2) x 4] [two pieces of A(S + 2)] -- at the top of sub-step s the six requests
  // of sub-step s - 1 may be outstanding and everything older has landed, which covers W(s) and all of stage S (issued
  // during stage S - 2): s_waitcnt vmcnt(6).  (Three sub-steps ahead for W, 16 more registers: 480 vs 466 us.)  (One step ahead for W measured 82 us of L2 latency on the critical path of a 339 us loop.)
  BFrag bq0, bq1, bq2;
  loadB(0, bq0);
  issueA(0, 0, 0); issueA(0, 0, 1); issueA(0, 0, 2); issueA(0, 0, 3);
  loadB(1, bq1);
  enter_stage(1);
  issueA(1, 1, 0); issueA(1, 1, 1); issueA(1, 1, 2); issueA(1, 1, 3);
  if (wave == 0) { etab[sA] = eA; etab[sB] = eB; }
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

#ifdef BSP_ABL_STAMP
  unsigned long long st_wait = 0, st_vm = 0;
#endif
  // One 16-deep sub-step s = 2 S + u of stage S: 24 MFMAs on the fragments `fa` (read from LDS during the PREVIOUS sub-step)
  // and the weight registers `bc`; `bn` receives the weight fragments of sub-step s + 2.  A single wave issues one
  // instruction per ~4 cycles, an MFMA occupies the matrix pipe for 32: whatever is issued in a block of its own (the
  // reads, loads and DMA requests of a sub-step at its top: ~600 cycles, measured with one workgroup per CU: 1490 cycles
  // per sub-step against 768 of MFMA) leaves the pipe idle, so every non-MFMA instruction sits in a gap between MFMAs:
  // per 32-row block mi the six MFMAs run hi*lo, lo*hi, hi*hi on two accumulators each, the lo fragment of mi is
  // re-read for sub-step s + 1 as soon as its last MFMA has been issued, the hi fragment after the block, and the four
  // weight loads / two DMA pieces are spread over the blocks.  The workgroup barrier of a new stage comes in the middle
  // of the last sub-step of its predecessor (before the first read of the new stage): every wave has passed the
  // top-of-sub-step wait that covers its own pieces of stage S + 1 (issued during stage S - 1) and has finished reading
  // stage S - 1, whose slot the requests of stage S + 2 (issued after that point in program order) overwrite.
  struct AFrag { f16x8 h[4], l[4]; };
  AFrag fa;
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
  auto step = [&](int s, int slot, int u, BFrag& bc, BFrag& bn) {
#ifdef BSP_ABL_STAMP
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
#endif
#if defined(BSP_ABL_NODMA) || defined(BSP_ABL_NOBLOAD)
    wait_vm<0>();
#else
    wait_b(bc);
#endif
#ifdef BSP_ABL_STAMP
    st_vm += __builtin_amdgcn_s_memtime() - c0;
#endif
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
#ifdef BSP_ABL_NOMFMA
      asm volatile("" ::"v"(fa.h[mi]), "v"(fa.l[mi]), "v"(bc.h[0]), "v"(bc.l[0]), "v"(bc.h[1]), "v"(bc.l[1]));
#else
      acc[mi][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa.h[mi], bl0, acc[mi][0], 0, 0, 0);
      acc[mi][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa.h[mi], bl1, acc[mi][1], 0, 0, 0);
#endif
#ifndef BSP_ABL_NOBLOAD
      if (mi < 2) loadB2(s + 2, bn, mi);
#endif
#ifndef BSP_ABL_NODMA
      if (mi == 2 && u == 0) enter_stage((s >> 1) + 2);
      if (mi >= 2) issueA((s >> 1) + 2, (slot + 2) % KC_RING, 2 * u + (mi - 2));   // two of the four pieces of stage S + 2
#endif
#ifndef BSP_ABL_NOMFMA
      acc[mi][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa.l[mi], bh0, acc[mi][0], 0, 0, 0);
      acc[mi][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa.l[mi], bh1, acc[mi][1], 0, 0, 0);
#endif
      if (mi == 0 && u == 1) {
#ifdef BSP_ABL_STAMP
        const unsigned long long c2 = __builtin_amdgcn_s_memtime();
#endif
        __builtin_amdgcn_sched_barrier(0); barrier_raw(); __builtin_amdgcn_sched_barrier(0);
#ifdef BSP_ABL_STAMP
        st_wait += __builtin_amdgcn_s_memtime() - c2;
#endif
      }
#ifndef BSP_ABL_NOLDSREAD
      const f16x8 nl = ldsfrag(sn + 4096 * mi + fo[u ^ 1][1]);
#else
      const f16x8 nl = __builtin_bit_cast(f16x8, bn.l[mi & 1]);
#endif
#ifndef BSP_ABL_NOMFMA
      acc[mi][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa.h[mi], bh0, acc[mi][0], 0, 0, 0);
      acc[mi][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa.h[mi], bh1, acc[mi][1], 0, 0, 0);
#endif
      fa.l[mi] = nl;
#ifndef BSP_ABL_NOLDSREAD
      fa.h[mi] = ldsfrag(sn + 4096 * mi + fo[u ^ 1][0]);
#else
      fa.h[mi] = __builtin_bit_cast(f16x8, bn.h[mi & 1]);
#endif
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
#ifndef BSP_KC_PRIO_LOOP
#define BSP_KC_PRIO_LOOP 2
#define BSP_KC_PRIO_EPI 0
#endif
  __builtin_amdgcn_s_setprio(BSP_KC_PRIO_LOOP);
  for (int s = 0; s < 2 * nst; s += 6) {   // an odd count of 16-deep steps runs one sub-step on zero weights
    step(s, 0, 0, bq0, bq2);
    step(s + 1, 0, 1, bq1, bq0);
    if (s + 2 < 2 * nst) { step(s + 2, 1, 0, bq2, bq1); step(s + 3, 1, 1, bq0, bq2); }
    if (s + 4 < 2 * nst) { step(s + 4, 2, 0, bq1, bq0); step(s + 5, 2, 1, bq2, bq1); }
  }
  __builtin_amdgcn_s_setprio(BSP_KC_PRIO_EPI);
  wait_vm<0>();        // rejected requests behind the last stage write zeros into the ring: drain before re-using it
  barrier_raw();
#ifdef BSP_ABL_STAMP
  const unsigned long long st1 = __builtin_amdgcn_s_memrealtime();
  const unsigned long long sc1 = __builtin_amdgcn_s_memtime();
#endif
#ifdef BSP_ABL_NOEPI
  {
    float sum = 0.f;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int nj = 0; nj < 2; ++nj)
#pragma unroll
        for (int r = 0; r < 16; ++r) sum += acc[mi][nj][r];
    if (sum == 12345.678f) p.EC[0] = 1;
    return;
  }
#endif

  // ---- epilogue, phase A: final values of the wave's 128 x 64 tile in row layout (8 consecutive columns per lane),
  //      their |max|, column sums; phase B (after the two waves of a 128 x 128 block have exchanged maxima): split + store.
  const int e_in = e_last + *p.EW;          // acc = true value * 2^e_in
  const bool e_small = e_in >= -120 && e_in <= 120;
  const float act_w0 = ACT == ACT_SIN ? p.w0 : 1.f;                  // sin(w0 (z + b)) = sin(z (2^-e w0) + b w0)
  const float inv_in = (e_small ? pow2f(-e_in) : 1.f) * act_w0;
  float* strip = reinterpret_cast<float*>(lds + wave * KC_STRIP);
  const int lc = lane & 31, lh = lane >> 5;
  // Row layout of the epilogue: lane -> row (lane >> 3) of an 8-row pass and eight consecutive columns.  The four
  // 16-column groups of the wave's 64 columns go to the lane pairs in the order 0, 2, 1, 3, so that each quad of lanes
  // owns groups {m, m + 2}: after one exchange inside the quad (phase B) a store instruction writes WHOLE 64-byte
  // groups, two neighbouring ones (128 contiguous bytes) per row.
  const int rrow = lane >> 3, l7 = lane & 7;
  const int c8 = 16 * ((((l7 >> 1) & 1) << 1) | (l7 >> 2)) + 8 * (l7 & 1);
  const int col = j0 + wj0 + c8;
  const bool col_ok = col < p.J;
  float val[4][4][8];
  float wmax = 0.f;
  int eH = 0;
  if (AUX != AUX_NONE && j0 + wj0 < p.J) eH = p.EH[(size_t)ti_e * ncb_of(p.ldh) + ((p.h_col0 + j0 + wj0) >> 7)];
  const float inv_h = pow2f(-eH);
  const size_t offH = uniform_sz(((size_t)i0 * p.ldh + p.h_col0) * 4);
  const srd_t srdH = make_srd(AUX != AUX_NONE ? p.H + offH : nullptr, AUX != AUX_NONE ? 0xFFFFFFE0u : 0u);
  float bj[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (p.bias != nullptr && col_ok) {
    const float4 b0 = *reinterpret_cast<const float4*>(p.bias + col), b1 = *reinterpret_cast<const float4*>(p.bias + col + 4);
    bj[0] = b0.x; bj[1] = b0.y; bj[2] = b0.z; bj[3] = b0.w; bj[4] = b1.x; bj[5] = b1.y; bj[6] = b1.z; bj[7] = b1.w;
#pragma unroll
    for (int c = 0; c < 8; ++c) bj[c] *= act_w0;
  }
  if (!e_small) {   // exponents beyond a single fp32 factor (never with sane data): scale the accumulators first
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int nj = 0; nj < 2; ++nj) acc[mi][nj] = scale_acc(acc[mi][nj], -e_in);
  }
  // stored activations (and sign words) of the derivative epilogues: block b + 1 is requested before block b is worked on
  constexpr int NH = AUX != AUX_NONE ? 4 : 1;
  u32x4 hh2[2][NH], hl2[2][NH];
  unsigned sword2[2] = {0u, 0u};
  auto load_h = [&](int b, u32x4 (&hh)[NH], u32x4 (&hl)[NH], unsigned& sword) {
    if (AUX == AUX_NONE) return;
    const int rbase = i0 + 32 * b;
#pragma unroll
    for (int ps = 0; ps < NH; ++ps) {
      const int rl = 32 * b + rrow + 8 * ps;
      const bool ok = col_ok && (i0 + rl) < p.I;
      const unsigned o = ok ? (unsigned)rl * (unsigned)p.ldh * 4u + (unsigned)(col >> 4) * 64u + (unsigned)(col & 8) * 2u : OOB;
      hh[ps] = __builtin_amdgcn_raw_buffer_load_b128(srdH, o, 0, 0);
      hl[ps] = __builtin_amdgcn_raw_buffer_load_b128(srdH, o == OOB ? OOB : o + 32u, 0, 0);
    }
    if (AUX == AUX_SINREC) {
      const size_t sidx = ((size_t)(rbase >> 5) * ((p.ldh + 63) >> 6) + ((p.h_col0 + j0 + wj0) >> 6)) * 64 + lane;
      sword = p.Hsign[(col_ok && rbase < p.I) ? sidx : 0];
    }
  };
  load_h(0, hh2[0], hl2[0], sword2[0]);
  float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // column sums over the tile's 128 rows (bias gradient partial)
  // derivative epilogues: the accumulator's scale and |w0| in one factor; the sign bits are xor-ed with w0's own sign
  const unsigned w0mag = __float_as_uint(fabsf(p.w0) * (AUX == AUX_SINREC ? inv_in : 1.f));
  const float pre_scale = AUX == AUX_SINREC ? 1.f : inv_in;
#pragma unroll
  for (int b = 0; b < 4; ++b) {          // 32-row blocks of the wave tile
    const int rbase = i0 + 32 * b;
    if (b + 1 < 4) load_h(b + 1, hh2[(b + 1) & 1], hl2[(b + 1) & 1], sword2[(b + 1) & 1]);
    u32x4 (&hh)[NH] = hh2[b & 1];
    u32x4 (&hl)[NH] = hl2[b & 1];
    const unsigned sword = sword2[b & 1];
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) strip[((r & 3) + 8 * (r >> 2) + 4 * lh) * 68 + 32 * n + lc] = acc[b][n][r];
    unsigned sbits = 0u;
    const unsigned swordx = sword ^ (p.w0 < 0.f ? 0xFFFFFFFFu : 0u);
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
      const bool ok = col_ok && (rbase + rrow + 8 * ps) < p.I;
      const float4 x0 = *reinterpret_cast<const float4*>(&strip[(rrow + 8 * ps) * 68 + c8]);
      const float4 x1 = *reinterpret_cast<const float4*>(&strip[(rrow + 8 * ps) * 68 + c8 + 4]);
      const float x[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
      float v[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) v[c] = AUX == AUX_SINREC ? x[c] : fmaf(x[c], pre_scale, bj[c]);   // exact power of two, then + bias (SIREN: both times w0)
      if (ACT == ACT_SIN) {
        sbits |= sin8_signbits<SIGNS>(v) << (8 * ps);
      } else if (ACT == ACT_RELU) {
#pragma unroll
        for (int c = 0; c < 8; ++c) v[c] = fmaxf(v[c], 0.f);
      }
      if (AUX != AUX_NONE) {
        float h[8];
        join8(hh[ps], hl[ps], inv_h, h);
        if (AUX == AUX_SINREC) {
          // w0 cos(w0 z) = +-|w0| sqrt(1 - h^2): the sign bit (xor-ed with w0's own sign, once per word) is shifted to bit
          // 31 and merged over |w0| by one v_bfi; 1 - h^2 is clamped at 0 by the FMA's output modifier
#pragma unroll
          for (int c = 0; c < 8; ++c) {
            float om;
            asm("v_fma_f32 %0, -%1, %1, 1.0 clamp" : "=v"(om) : "v"(h[c]));
            unsigned w0s_bits;
            asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(w0s_bits) : "s"(0x7fffffffu), "v"(w0mag), "v"(swordx << (31 - (8 * ps + c))));
            const float w0s = __uint_as_float(w0s_bits);
            v[c] *= w0s * __builtin_amdgcn_sqrtf(om);
          }
        } else {
#pragma unroll
          for (int c = 0; c < 8; ++c) v[c] = h[c] > 0.f ? v[c] : 0.f;
        }
      }
      // rows / columns outside the problem: their stores are rejected by the descriptor; keep them out of the block
      // maximum and the column sums (their values are finite: zero operand rows through the same arithmetic)
      const float okf = ok ? 1.f : 0.f;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        if (COLSUM) cs[c] = fmaf(v[c], okf, cs[c]);
        val[b][ps][c] = v[c];
      }
      wmax = fmaxf(wmax, okf * absmax3(v[6], v[7], absmax3(v[4], v[5], absmax3(v[2], v[3], absmax3(v[0], v[1], 0.f)))));
    }
    if (ACT == ACT_SIN && SIGNS && p.Csign != nullptr && col_ok && rbase < p.I)
      p.Csign[((size_t)(rbase >> 5) * ((p.ldc + 63) >> 6) + ((p.c_col0 + j0 + wj0) >> 6)) * 64 + lane] = sbits;
  }
  if (COLSUM && p.colsum != nullptr) {   // one partial row per 128-row tile: the lanes' sums over their 16 rows, then over the 8 row groups
#pragma unroll
    for (int o = 8; o < 64; o <<= 1)
#pragma unroll
      for (int c = 0; c < 8; ++c) cs[c] += __shfl_xor(cs[c], o, 64);
    if (lane < 8 && col_ok && i0 < p.I) {
      float* d = p.colsum + (size_t)ti * p.ldcs + col;
      *reinterpret_cast<float4*>(d) = make_float4(cs[0], cs[1], cs[2], cs[3]);
      *reinterpret_cast<float4*>(d + 4) = make_float4(cs[4], cs[5], cs[6], cs[7]);
    }
  }
#ifdef BSP_ABL_STAMP
  const unsigned long long se1 = __builtin_amdgcn_s_memtime();
#endif
  // block maximum: waves 2 c and 2 c + 1 share the exponent block (ti, column block c of the tile)
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) wmax = fmaxf(wmax, __shfl_xor(wmax, o, 64));
  if (lane == 0) smax[wave] = wmax;
  __syncthreads();
#ifdef BSP_ABL_STAMP
  const unsigned long long se2 = __builtin_amdgcn_s_memtime();
#endif
  const float bmax = fmaxf(smax[wave & 2], smax[(wave & 2) + 1]);
  const int eC = exp_of_maxbits(__float_as_uint(bmax));
  const float sc = pow2f(eC);
  if ((wave & 1) == 0 && lane == 0 && j0 + wj0 < p.J && i0 < p.I) p.EC[(size_t)ti * ncb_of(p.ldc) + ((p.c_col0 + j0 + wj0) >> 7)] = eC;
  const size_t offC = uniform_sz(((size_t)i0 * p.ldc + p.c_col0) * 4);
  const srd_t srdC = make_srd(p.C + offC, 0xFFFFFFE0u);
  // quad exchange: lanes 0, 1 of a quad hold group m (hi and lo planes of columns 0-7 / 8-15), lanes 2, 3 group m + 2.
  // Lanes 0, 1 hand their lo planes to lanes 2, 3 and get those lanes' hi planes; store 1 then writes group m complete
  // ([hi 0-7 | hi 8-15 | lo 0-7 | lo 8-15] = the four lanes' 16 bytes in address order), store 2 group m + 2.
  const int q4 = lane & 3, mq = l7 >> 2;
  const bool lowpair = q4 < 2;
  const bool g1_ok = j0 + wj0 + 16 * mq < p.J, g2_ok = j0 + wj0 + 16 * (mq + 2) < p.J;
  const unsigned ocol = (unsigned)(((j0 + wj0) >> 4) + mq) * 64u + (unsigned)q4 * 16u;
#pragma unroll
  for (int b = 0; b < 4; ++b) {
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
      const int rl = 32 * b + rrow + 8 * ps;
      const bool rok = (i0 + rl) < p.I;
      u32x4 hi, lo, d1, d2;
      split8(val[b][ps], sc, hi, lo);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const unsigned send = lowpair ? lo[i] : hi[i];
        const unsigned recv = (unsigned)__builtin_amdgcn_update_dpp(0, (int)send, 0x4E, 0xF, 0xF, false);   // quad_perm [2, 3, 0, 1]
        d1[i] = lowpair ? hi[i] : recv;
        d2[i] = lowpair ? recv : lo[i];
      }
#ifdef BSP_ABL_NOSTORE
      const unsigned o = (rok && p.I < 0) ? 0u : OOB;
#else
      const unsigned o = rok ? (unsigned)rl * (unsigned)p.ldc * 4u + ocol : OOB;
#endif
      __builtin_amdgcn_raw_buffer_store_b128(d1, srdC, (o != OOB && g1_ok) ? o : OOB, 0, 0);
      __builtin_amdgcn_raw_buffer_store_b128(d2, srdC, (o != OOB && g2_ok) ? o + 128u : OOB, 0, 0);
    }
  }
#ifdef BSP_ABL_STAMP
  const unsigned long long se3 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long se4 = __builtin_amdgcn_s_memtime();
  // stamp buffer: the (otherwise unused) colsum pointer of a launch without column sums, the sign-word pointer of one with
  unsigned long long* dbg0 = COLSUM ? reinterpret_cast<unsigned long long*>(p.Csign) : reinterpret_cast<unsigned long long*>(p.colsum);
  if (t == 0 && dbg0 != nullptr) {   // 100 MHz stamps (start, loop end, end) + shader-clock cycles of the phases
    unsigned long long* dbg = dbg0 + 8 * (size_t)blockIdx.x;
    dbg[0] = st0; dbg[1] = st1; dbg[2] = __builtin_amdgcn_s_memrealtime();
    dbg[3] = sc1 - sc0;                                   // k-loop
    dbg[4] = (st_vm << 32) | (st_wait & 0xFFFFFFFFull);   // of it: in vmcnt waits | in barriers
    dbg[5] = se1 - sc1;                                   // epilogue phase A (values, activation, maxima)
    dbg[6] = se2 - se1;                                   // exchange of the block maxima (workgroup barrier)
    dbg[7] = ((se3 - se2) << 32) | ((se4 - se3) & 0xFFFFFFFFull);   // split + store issue | store drain
  }
#endif
}

// ------------------------------------------------------------------------------------------------------------------
// K-contiguous GEMM, 32-wide fp32 output (pre-activations of sigma / sun visibility / final head layers: the composite
// kernels apply their activations).  128 x 32 tile, four waves of 32 x 32; the weight fragments (one 32-row unit, the
// same for every wave) come straight from L2 as above.
// ------------------------------------------------------------------------------------------------------------------
constexpr int KN_A = 128 * 64, KN_RING = 4, KN_TAIL = KN_RING * KN_A, KN_LDS = KN_TAIL + 512;

__global__ __launch_bounds__(256, 4) void gemm_kcn_kernel(const KcArgs p) {
  __shared__ __attribute__((aligned(16))) char lds[KN_LDS];
  int* etab = reinterpret_cast<int*>(lds + KN_TAIL);
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int ti = blockIdx.x, i0 = ti * 128, wi0 = wave * 32;
  const int nks = p.K >> 4, nks1 = p.Ka >> 4;
  const srd_t srdA = make_srd(p.A + ((size_t)i0 * p.lda + p.a_col0) * 4,
                              clamp_bytes(i0 < p.I ? ((unsigned long long)(p.I - i0 - 1) * p.lda + p.Ka) * 4ull : 0ull));
  const srd_t srdW = make_srd(p.W, p.w_bytes);
  unsigned voA[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int row = 64 * q + (t >> 2);
    const unsigned c = (unsigned)((t & 3) ^ ((row >> 2) & 3));
    voA[q] = (i0 + row < p.I) ? (unsigned)row * (unsigned)p.lda * 4u + 16u * c : OOB;
  }
  const unsigned w_u0 = (unsigned)(p.w_row0 >> 5), w_ks0 = (unsigned)(p.w_k0 >> 4);
  auto issueA = [&](int s, int slot) {
#pragma unroll
    for (int q = 0; q < 2; ++q) dma16(srdA, lds + slot * KN_A + 4096 * q + wave * 1024, s < nks ? voA[q] : OOB, (unsigned)s * 64u);
  };
  struct BFrag { u32x4 h, l; };
  auto loadB = [&](int s, BFrag& b) {
    const unsigned so = ((w_ks0 + (unsigned)s) * (unsigned)p.w_rb32 + w_u0) * 2048u;
    const unsigned vo = s < nks ? 16u * (unsigned)lane : OOB;
    b.h = __builtin_amdgcn_raw_buffer_load_b128(srdW, vo, so, 0);
    b.l = __builtin_amdgcn_raw_buffer_load_b128(srdW, vo == OOB ? OOB : vo + 1024u, so, 0);
  };
  const int sA = lane, sB = lane + 64;
  const int eA = sA < nks ? kc_exp_of_step(p, ti, sA, nks1) : 0;
  const int eB = sB < nks ? kc_exp_of_step(p, ti, sB, nks1) : 0;
  const int eAp = (sA > 0 && sA < nks) ? kc_exp_of_step(p, ti, sA - 1, nks1) : eA;
  const int eBp = sB < nks ? kc_exp_of_step(p, ti, sB - 1, nks1) : eB;
  const int e_last = kc_exp_of_step(p, ti, nks - 1, nks1);
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
    acc = mfma3(ldsfrag(st + wi0 * 64 + fo0), ldsfrag(st + wi0 * 64 + fo1), __builtin_bit_cast(f16x8, bc.h), __builtin_bit_cast(f16x8, bc.l), acc);
  };
  for (int s = 0; s < nks; s += 4) {
    step(s, 0, bq0, bq1);
    if (s + 1 < nks) step(s + 1, 1, bq1, bq0);
    if (s + 2 < nks) step(s + 2, 2, bq0, bq1);
    if (s + 3 < nks) step(s + 3, 3, bq1, bq0);
  }
  wait_vm<0>();
  const int e_in = e_last + *p.EW;
  const int col = lane & 31;
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
template <int TI> struct DwCfg {
  static constexpr int NTH = TI == 256 ? 512 : 256, WAVES = NTH / 64;
  static constexpr int MI = TI == 256 ? 4 : 1;                   // 32-row blocks per wave along i
  static constexpr int A_PITCH = TI * 4;                         // bytes of one point row of the A stage (TI columns, both planes)
  static constexpr int A_BYTES = 16 * A_PITCH, B_BYTES = 16 * 1024, STAGE = A_BYTES + B_BYTES, RING = 3;
  static constexpr int STRIPS = WAVES * KC_STRIP;
  static constexpr int TAIL = (RING * STAGE > STRIPS) ? RING * STAGE : STRIPS;
  static constexpr int MAXCH = 128;                              // 128-point chunks per split (k_split <= 16384)
  static constexpr int DUMMY = TAIL + WAVES * MAXCH * 4;        // scratch KiB per wave for rejected pieces (the narrow form)
  static constexpr int LDS = DUMMY + (TI == 256 ? 0 : WAVES * 1024);
};
// byte swizzle of point row p inside a stage: 1 KiB rows put the four rows of a transposed-read block into four bank
// quarters; the 128-byte rows of the 32-column form need only the hi/lo flip of the upper two rows
template <int PITCH> __device__ __forceinline__ unsigned dw_swz(int p) {
  return PITCH == 1024 ? (unsigned)(((p & 1) << 5) | ((p & 2) << 6)) : (unsigned)((p & 2) << 4);
}

template <int TI>
__global__ __launch_bounds__(DwCfg<TI>::NTH, TI == 256 ? 2 : 2) void gemm_dw_kernel(const DwArgs p) {
  using T = DwCfg<TI>;
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
  const int nks = (kEnd - kBeg + 15) >> 4;
  float* C = p.C + (size_t)split * p.slab_stride;

  // ---- DMA sources: a piece is 1 KiB = one point row of the 256-column operand (eight 128-byte rows of the 32-column one)
  const srd_words srdA = make_srd_words(p.A + ((size_t)kBeg * p.lda + p.a_col0 + i0) * 4,
                                        clamp_bytes(kEnd > kBeg ? ((unsigned long long)(kEnd - kBeg) * p.lda - (p.a_col0 + i0)) * 4ull : 0ull));
  const srd_words srdB = make_srd_words(p.B + ((size_t)kBeg * p.ldb + p.b_col0 + j0) * 4,
                                        clamp_bytes(kEnd > kBeg ? ((unsigned long long)(kEnd - kBeg) * p.ldb - (p.b_col0 + j0)) * 4ull : 0ull));
  const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr(lds));
  constexpr int PA = T::A_PITCH;
  constexpr int A_PIECES = T::A_BYTES / 1024;                         // 16 (TI = 256) or 2 (TI = 32)
  constexpr int NPA = (A_PIECES + T::WAVES - 1) / T::WAVES;           // A pieces per wave: 2 / 1 (waves >= 2 of the narrow form: a rejected one)
  constexpr int NPB = 16 / T::WAVES;                                  // B pieces per wave: 2 / 4
  unsigned voA[NPA], voB[NPB];
  int prA[NPA], prB[NPB];        // point row (inside the stage) each piece's lane belongs to
#pragma unroll
  for (int q = 0; q < NPA; ++q) {
    const int piece = wave * NPA + q;
    const int pr = PA == 1024 ? piece : piece * 8 + (lane >> 3);
    const unsigned byte = PA == 1024 ? 16u * lane : 16u * (lane & 7);
    prA[q] = piece < A_PIECES ? pr : -1;
    voA[q] = (unsigned)pr * (unsigned)p.lda * 4u + (byte ^ dw_swz<PA>(pr));
  }
#pragma unroll
  for (int q = 0; q < NPB; ++q) {
    const int pr = wave * NPB + q;
    prB[q] = pr;
    voB[q] = (unsigned)pr * (unsigned)p.ldb * 4u + ((16u * lane) ^ dw_swz<1024>(pr));
  }
  auto issue = [&](int s, int slot) {
    const int prow0 = kBeg + 16 * s;
#pragma unroll
    for (int q = 0; q < NPA; ++q) {
      const int piece = wave * NPA + q;
      const bool ok = s < nks && prA[q] >= 0 && prow0 + prA[q] < kEnd;
      dma16_asm(srdA, lds0 + (unsigned)(piece < A_PIECES ? slot * T::STAGE + piece * 1024 : T::DUMMY + wave * 1024), ok ? voA[q] : OOB,
                (unsigned)s * 16u * (unsigned)p.lda * 4u);
    }
#pragma unroll
    for (int q = 0; q < NPB; ++q) {
      const bool ok = s < nks && prow0 + prB[q] < kEnd;
      dma16_asm(srdB, lds0 + (unsigned)(slot * T::STAGE + T::A_BYTES + prB[q] * 1024), ok ? voB[q] : OOB, (unsigned)s * 16u * (unsigned)p.ldb * 4u);
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
  auto frag_off = [&](int col, int pl, int pitch, unsigned sw) -> unsigned {   // col: first column of the 16-group (% 16 == 0)
    return (unsigned)kq * (unsigned)pitch + ((((unsigned)(col >> 4) * 64u) + (unsigned)pl * 32u + 8u * (unsigned)pp) ^ sw);
  };
  unsigned foA[T::MI][2], foB[2][2];
#pragma unroll
  for (int mi = 0; mi < T::MI; ++mi)
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) foA[mi][pl] = frag_off(wi0 + 32 * mi + 16 * (g & 1), pl, PA, dw_swz<PA>(q4));
#pragma unroll
  for (int nj = 0; nj < 2; ++nj)
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) foB[nj][pl] = (unsigned)T::A_BYTES + frag_off(wj0 + 32 * nj + 16 * (g & 1), pl, 1024, dw_swz<1024>(q4));
  typedef __fp16 h4_t __attribute__((__vector_size__(4 * sizeof(__fp16))));
  typedef h4_t __attribute__((address_space(3))) * lds4_t;
  auto trfrag = [&](const char* base, unsigned off, int pitch) -> f16x8 {
    const f16x4 a = __builtin_bit_cast(f16x4, __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds4_t)(base + off)));
    const f16x4 b = __builtin_bit_cast(f16x4, __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds4_t)(base + off + 4 * pitch)));
    return f16x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  };

  int e_cur = 0;
  constexpr int NPIECE = NPA + NPB;
  // exponent bookkeeping at the start of a 128-point chunk: rescale the accumulators when the pair of exponents changes
  auto chunk = [&](int s) {
    if ((s & 7) != 0) return;
    const int e_new = __builtin_amdgcn_readfirstlane(esum[wave * T::MAXCH + (s >> 3)]);
    if (s == 0) e_cur = e_new;
    else if (__builtin_expect(e_new != e_cur, 0)) {
      const int de = e_new - e_cur;
#pragma unroll
      for (int mi = 0; mi < T::MI; ++mi)
#pragma unroll
        for (int nj = 0; nj < 2; ++nj) acc[mi][nj] = scale_acc(acc[mi][nj], de);
      e_cur = e_new;
    }
  };
#ifndef BSP_DW_LOCKSTEP
  if constexpr (TI == 256) {
    // Ping-pong: the eight waves form two groups (tile rows 0-127 / 128-255; one wave of each per SIMD) that run the same
    // two-phase step -- M: request stage s + 2, read the 24 fragments of stage s | C: 24 MFMAs on those registers -- one
    // phase apart, with a workgroup barrier after every phase.  While one wave of a SIMD issues its MFMAs back to back the
    // other does its LDS reads and DMA issue; in lockstep (all eight waves request, read, compute together) the three
    // costs add up: measured 336 us lockstep / 311 us ping-pong at 262,144 x 512 x 512 (-DBSP_DW_LOCKSTEP builds the former).
    //   phase:    0      1      2      3     ...
    //   group 0:  M(0)   C(0)   M(1)   C(1)
    //   group 1:  -      M(0)   C(0)   M(1)
    // Stage s + 1 is awaited (own pieces, counted vmcnt) before the barrier that ends phase 2 s + 1, one barrier ahead of
    // its first reader (group 0, phase 2 s + 2); stage s + 2 is requested into the slot of stage s - 1 no earlier than
    // phase 2 s, one barrier after its last reader (group 1, phase 2 s - 1).
    const int grp = wave >> 2;
    f16x8 fa_h[4], fa_l[4], fb_h[2], fb_l[2];
    auto phaseM = [&](int s, int slot) {
#ifndef BSP_ABL_DW_NODMA
      issue(s + 2, (slot + 2) % 3);
#endif
      chunk(s);
      const char* st = lds + slot * T::STAGE;
#ifdef BSP_ABL_DW_NOLDSREAD
      const f16x8 cst = __builtin_bit_cast(f16x8, u32x4{(unsigned)s, 1u, 2u, (unsigned)(size_t)st});
#pragma unroll
      for (int nj = 0; nj < 2; ++nj) { fb_h[nj] = cst; fb_l[nj] = cst; }
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) { fa_h[mi] = cst; fa_l[mi] = cst; }
#else
#pragma unroll
      for (int nj = 0; nj < 2; ++nj) { fb_h[nj] = trfrag(st, foB[nj][0], 1024); fb_l[nj] = trfrag(st, foB[nj][1], 1024); }
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) { fa_h[mi] = trfrag(st, foA[mi][0], PA); fa_l[mi] = trfrag(st, foA[mi][1], PA); }
#endif
    };
    auto phaseC = [&]() {
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#ifdef BSP_ABL_DW_NOMFMA
        asm volatile("" ::"v"(fa_h[mi]), "v"(fa_l[mi]), "v"(fb_h[0]), "v"(fb_l[0]), "v"(fb_h[1]), "v"(fb_l[1]));
#else
#pragma unroll
        for (int nj = 0; nj < 2; ++nj) acc[mi][nj] = mfma3(fa_h[mi], fa_l[mi], fb_h[nj], fb_l[nj], acc[mi][nj]);
#endif
      __builtin_amdgcn_s_setprio(0);
    };
    wait_vm<NPIECE>();          // stage 0 (own pieces); stage 1 stays in flight
    barrier_raw();
    if (grp == 0) {
      int slot = 0;
      for (int s = 0; s < nks; ++s) {
        phaseM(s, slot);
        barrier_raw();
        phaseC();
        wait_vm<NPIECE>();      // stage s + 1 landed (stage s + 2 in flight)
        barrier_raw();
        slot = slot == 2 ? 0 : slot + 1;
      }
      barrier_raw();            // group 1's last compute phase
    } else {
      barrier_raw();            // group 0's first memory phase
      int slot = 0;
      for (int s = 0; s < nks; ++s) {
        phaseM(s, slot);
        wait_vm<NPIECE>();      // stage s + 1 landed (the stage s + 2 just requested stays in flight)
        barrier_raw();
        phaseC();
        barrier_raw();
        slot = slot == 2 ? 0 : slot + 1;
      }
    }
  } else
#endif
  {
  auto step = [&](int s, int slot) {
#ifndef BSP_ABL_DW_NODMA
    wait_vm<NPIECE>();
#endif
    barrier_raw();
#ifndef BSP_ABL_DW_NODMA
    issue(s + 2, (slot + 2) % 3);
#endif
    chunk(s);
    const char* st = lds + slot * T::STAGE;
    f16x8 bh[2], bl[2];
#ifdef BSP_ABL_DW_NOLDSREAD
    bh[0] = bh[1] = bl[0] = bl[1] = __builtin_bit_cast(f16x8, u32x4{(unsigned)s, 1u, 2u, 3u});
#else
#pragma unroll
    for (int nj = 0; nj < 2; ++nj) { bh[nj] = trfrag(st, foB[nj][0], 1024); bl[nj] = trfrag(st, foB[nj][1], 1024); }
#endif
#pragma unroll
    for (int mi = 0; mi < T::MI; ++mi) {
#ifdef BSP_ABL_DW_NOLDSREAD
      const f16x8 ah = bh[0], al = bl[1];
#else
      const f16x8 ah = trfrag(st, foA[mi][0], PA), al = trfrag(st, foA[mi][1], PA);
#endif
#ifdef BSP_ABL_DW_NOMFMA
      asm volatile("" ::"v"(ah), "v"(al), "v"(bh[0]), "v"(bl[0]), "v"(bh[1]), "v"(bl[1]));
#else
#pragma unroll
      for (int nj = 0; nj < 2; ++nj) acc[mi][nj] = mfma3(ah, al, bh[nj], bl[nj], acc[mi][nj]);
#endif
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
      __builtin_amdgcn_raw_buffer_store_b128(d, srdC, ok ? ((unsigned)rl * (unsigned)p.ldc + (unsigned)c4) * 4u : OOB, 0, 0);
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

static int check_kc(const KcArgs& a, bool narrow) {
  if (!a.A || !a.EA || !a.W || !a.EW) return bad("null operand");
  if (a.I <= 0 || a.J <= 0 || a.K <= 0) return bad("empty problem");
  if ((a.K & 15) || (a.Ka & 15) || a.Ka <= 0 || a.Ka > a.K || a.K > 2048) return bad("K and Ka must be multiples of 16, K <= 2048");
  if ((a.lda & 15) || (a.a_col0 & 15) || a.a_col0 + a.Ka > a.lda) return bad("A segment does not fit its tensor (16-column groups)");
  if (a.Ka < a.K && (a.Ka & 31)) return bad("two A segments: the first must be a multiple of 32 columns (one LDS stage)");
  if (a.Ka < a.K && (!a.A2 || !a.EA2 || (a.lda2 & 15) || (a.a2_col0 & 15) || a.a2_col0 + (a.K - a.Ka) > a.lda2)) return bad("second A segment");
  if ((a.w_row0 & 31) || (a.w_k0 & 15) || a.w_rb32 <= 0) return bad("weight operand must start on a 32-row / 16-k unit");
  if ((size_t)128 * (a.lda > a.lda2 ? a.lda : a.lda2) * 4 >= 0x7FFFFFFFull) return bad("leading dimension too large");
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
  if ((size_t)128 * a.ldc * 4 >= 0x7FFFFFFFull) return bad("ldc too large");
  return SNERF_OK;
}

int launch_kc(const KcArgs& a0, hipStream_t st) {
  KcArgs a = a0;
  if (!a.A2) { a.A2 = a.A; a.EA2 = a.EA; a.lda2 = a.lda; a.a2_col0 = a.a_col0; if (a.Ka == 0) a.Ka = a.K; }
  int rc = check_kc(a, false);
  if (rc) return rc;
  a.tiles_i = ((a.I + 127) / 128 + KC_HALVES - 1) / KC_HALVES;
  a.tiles_j = (a.J + 255) / 256;
  const dim3 grid(a.tiles_i * a.tiles_j), block(256 * KC_HALVES);
  const int tok = prof_hook_begin(2.0 * a.I * (double)a.J * a.K, 0, st);
  const bool cs = a.colsum != nullptr;
#define KC_LAUNCH(ACT_, AUX_, CS_) hipLaunchKernelGGL((gemm_kc_kernel<ACT_, AUX_, CS_>), grid, block, 0, st, a)
  if (a.aux_mode == AUX_SINREC) KC_LAUNCH(ACT_NONE, AUX_SINREC, true);
  else if (a.aux_mode == AUX_RELU_MASK) KC_LAUNCH(ACT_NONE, AUX_RELU_MASK, true);
  else if (a.act == ACT_SIN && a.Csign == nullptr) hipLaunchKernelGGL((gemm_kc_kernel<ACT_SIN, AUX_NONE, false, false>), grid, block, 0, st, a);
  else if (a.act == ACT_SIN) KC_LAUNCH(ACT_SIN, AUX_NONE, false);
  else if (a.act == ACT_RELU) KC_LAUNCH(ACT_RELU, AUX_NONE, false);
  else if (cs) KC_LAUNCH(ACT_NONE, AUX_NONE, true);
  else KC_LAUNCH(ACT_NONE, AUX_NONE, false);
#undef KC_LAUNCH
  SNERF_LAUNCH_CHECK();
  prof_hook_end(tok, st);
  return SNERF_OK;
}

int launch_kc_narrow(const KcArgs& a0, hipStream_t st) {
  KcArgs a = a0;
  if (a.Ka == 0) a.Ka = a.K;
  a.A2 = a.A; a.EA2 = a.EA; a.lda2 = a.lda; a.a2_col0 = a.a_col0;
  int rc = check_kc(a, true);
  if (rc) return rc;
  const int tok = prof_hook_begin(2.0 * a.I * 32.0 * a.K, 3, st);
  hipLaunchKernelGGL(gemm_kcn_kernel, dim3((a.I + 127) / 128), dim3(256), 0, st, a);
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
  if ((size_t)a.k_split * (a.lda > a.ldb ? a.lda : a.ldb) * 4 >= 0xFFFFFFF0ull) return bad("dW: k_split * ld exceeds the 32-bit span");
  if (narrow_i ? (a.I > 32 || ((a.a_col0 & 127) + 32 > 128)) : ((a.a_col0 & 127) != 0)) return bad("dW: the A columns of a wave must lie in one exponent block");
  if (a.b_col0 & 63) return bad("dW: b_col0 % 64");
  if (((uintptr_t)a.A & 15) || ((uintptr_t)a.B & 15) || ((uintptr_t)a.C & 15)) return bad("dW: alignment");
  const int TI = narrow_i ? 32 : 256;
  a.tiles_i = (a.I + TI - 1) / TI;
  a.tiles_j = (a.J + 255) / 256;
  const dim3 grid(a.tiles_i * a.tiles_j, 1, a.n_split);
  const int tok = prof_hook_begin(2.0 * a.I * (double)a.J * a.P, narrow_i ? 3 : 2, st);
  if (narrow_i) hipLaunchKernelGGL(gemm_dw_kernel<32>, grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL(gemm_dw_kernel<256>, grid, dim3(512), 0, st, a);
  SNERF_LAUNCH_CHECK();
  prof_hook_end(tok, st);
  return SNERF_OK;
}

}  // namespace bsp
}  // namespace snerf

// Device-side helpers shared by the block-scaled plane GEMMs (bsp_kc.hip, bsp_gemm.hip): LDS-DMA requests, counted
// waits, raw barriers, fragment reads, accumulator rescaling.
#pragma once
#include "bsp.h"
#include "tiles.h"

namespace snerf {
namespace bsp {

typedef __attribute__((address_space(3))) void* lds_ptr_t;

// HAZARD (measured on gfx950, round 4; tools/check_vgpr_hazards.py finds the pattern in the assembly, tests/test_build_cpu.py runs it):
// a VALU instruction issued right behind a buffer_store_dwordx4 may overwrite the store's DATA registers before the store has
// fetched them -- dword 0 of lanes 25/27/29/31 (+32) then carried the VALU result (a row index) in ~1.5 % of the rows, different
// from run to run.  LLVM's hazard recognizer inserts the wait state only for stores with an immediate soffset (the documented
// case); the plane stores use an SGPR soffset and got none.  Call this behind every wide store with its data: the
// value stays live across two wait states, so no later write can be allocated into those registers any earlier.
// (-DKC_NO_STORE_GUARD: tools/ablate/build_noguard.sh, the reproducer of the corruption -- never the product build)
__device__ __forceinline__ void store_data_guard(const u32x4& d) {
#ifndef KC_NO_STORE_GUARD
  asm volatile("s_nop 1" ::"v"(d));
#endif
}

__device__ __forceinline__ void dma16(srd_t srd, char* lds_dst, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (lds_ptr_t)lds_dst, 16, voff, soff, 0, 0);
}
// The same request as an asm statement the compiler cannot see into.  hipcc models the builtin form as a store to LDS and,
// where its alias analysis cannot separate the destination from a following LDS read (the dW kernel's transposed reads),
// puts s_waitcnt vmcnt(0) between them -- every stage request is then drained right after it is issued and the whole
// DMA latency sits on the critical path of every k-step.  Completion is tracked by the kernels' own counted waits either
// way.  M0 (the LDS destination) is saved and restored inside; s_nop 2 (with the two s_mov in front: five wait states) covers a
// descriptor / offset operand the compiler has just restored from a spill lane (v_readlane -> vector-memory read).
typedef unsigned int srd_words __attribute__((ext_vector_type(4)));
__device__ __forceinline__ srd_words make_srd_words(const void* p, unsigned bytes) {
  const unsigned long long a = reinterpret_cast<unsigned long long>(p);
  return srd_words{(unsigned)__builtin_amdgcn_readfirstlane((unsigned)a), (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xFFFFu),
                   (unsigned)__builtin_amdgcn_readfirstlane(bytes), 0x00020000u};
}
__device__ __forceinline__ void dma16_asm(srd_words srd, unsigned lds_byte_addr, unsigned voff, unsigned soff) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 2\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "s"(lds_byte_addr), "v"(voff), "s"(srd), "s"(soff));
}
// the same request with the non-temporal cache policy: data that is read exactly once (the stored activations of a derivative epilogue)
__device__ __forceinline__ void dma16_asm_nt(srd_words srd, unsigned lds_byte_addr, unsigned voff, unsigned soff) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 2\n\tbuffer_load_dwordx4 %2, %3, %4 offen nt lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "s"(lds_byte_addr), "v"(voff), "s"(srd), "s"(soff));
}
// one dword per lane (256 B per wave) by the same route
__device__ __forceinline__ void dma4_asm(srd_words srd, unsigned lds_byte_addr, unsigned voff, unsigned soff) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 2\n\tbuffer_load_dword %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
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

// exponent of k-step s of a K-contiguous A operand made of one or two segments (any lane; uniform inputs)
__device__ __forceinline__ int kc_exp_of_step(const KcArgs& p, int rb, int s, int nks1, int ksub = 16) {
  const bool seg2 = s >= nks1;   // branch-free: one load through a selected pointer
  const int* E = seg2 ? p.EA2 : p.EA;
  const int ld = seg2 ? p.lda2 : p.lda, col = seg2 ? p.a2_col0 + ksub * (s - nks1) : p.a_col0 + ksub * s;
  return E[(size_t)rb * ncb_of(ld) + (col >> 7)];
}

// The kernel arguments where they lie (kernarg segment, scalar loads).  Every call returns a pointer the compiler must take
// as new, so values read through it are re-read at the use site instead of being kept in scalar registers (and spilled)
// across a long loop.  Only for kernels whose single argument is the struct.
typedef const __attribute__((address_space(4))) KcArgs* kargs_t;
__device__ __forceinline__ kargs_t kargs() {
  kargs_t q = (kargs_t)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(q));
  return q;
}
template <class P>
__device__ __forceinline__ int kc_exp_of_step_p(P p, int rb, int s, int nks1) {
  const bool seg2 = s >= nks1;
  const int* E = seg2 ? p->EA2 : p->EA;
  const int ld = seg2 ? p->lda2 : p->lda, col = seg2 ? p->a2_col0 + 16 * (s - nks1) : p->a_col0 + 16 * s;
  return E[(size_t)rb * ncb_of(ld) + (col >> 7)];
}

}  // namespace bsp
}  // namespace snerf

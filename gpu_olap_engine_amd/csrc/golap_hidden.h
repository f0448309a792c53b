// golap_hidden.h -- loads, stores and atomics that hipcc does not see as memory operations of its own (inline asm), and the counted
// s_waitcnt that goes with them.  vmcnt counts a wave's loads AND stores in issue order; hipcc cannot count stores issued under
// data-dependent control flow and waits vmcnt(0) at the first use of a prefetched value -- i.e. for every store issued behind the
// prefetch.  Kernels that use these helpers issue every store at wave-uniform control flow (predicated through EXEC inside the
// statement), so the number of operations younger than a load is a compile-time constant (cdna_hip_programming.md 5.7 item 1,
// form (ii)).  Users: golap_part_narrow.h (both partition passes of the join), golap_part_wide.h (the 64-bit-row pipeline's last pass),
// golap_ops.hip (k_agg_scatter_rsv).
#pragma once
#include <stdint.h>

namespace golap {

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void ld16_hidden(u32x4& v, const void* p) { asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory"); }
__device__ __forceinline__ void ld8_hidden(u32x2& v, const void* p) { asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(v) : "v"(p) : "memory"); }
__device__ __forceinline__ void ld4_hidden(uint32_t& v, const void* p) { asm volatile("global_load_dword %0, %1, off" : "=v"(v) : "v"(p) : "memory"); }
// Hidden loads of the ring pass go into FIXED registers v[104:119] that hipcc never allocates: k_part2_ring caps the compiler's
// budget at 96 VGPRs (amdgpu_waves_per_eu(5); the kernel still gets what the asm statements name, occupancy is set by its LDS
// anyway), the load names its destination in the asm text, and ld8_take copies the pair out AFTER the counted wait.  No
// compiler-visible value ever lives in a register with a load in flight, so no copy, phi or reuse of such a register can exist
// (the first form of that kernel returned the data as asm outputs: at a loop back edge hipcc copied the "loaded" pair to
// another register right behind the load, and reused the original one as a store address -- a memory fault on the GPU).
template <int SLOT> __device__ __forceinline__ void ld8_fixed(const void* sbase, uint32_t off);   // sbase: work-group-uniform (SGPR pair), off: bytes
template <int SLOT> __device__ __forceinline__ u32x2 ld8_take();
#define GOLAP_FIXED_PAIR(SLOT, LO, HI)                                                                                            \
    template <> __device__ __forceinline__ void ld8_fixed<SLOT>(const void* sbase, uint32_t off) {                                 \
        asm volatile("global_load_dwordx2 v[" #LO ":" #HI "], %0, %1" ::"v"(off), "s"(sbase) : "memory", "v" #LO, "v" #HI);      \
    }                                                                                                                              \
    template <> __device__ __forceinline__ u32x2 ld8_take<SLOT>() {                                                               \
        uint32_t x, y;                                                                                                             \
        asm volatile("v_mov_b32 %0, v" #LO "\n\tv_mov_b32 %1, v" #HI : "=v"(x), "=v"(y)::"memory");                              \
        return u32x2{x, y};                                                                                                        \
    }
GOLAP_FIXED_PAIR(0, 104, 105)
GOLAP_FIXED_PAIR(1, 106, 107)
GOLAP_FIXED_PAIR(2, 108, 109)
GOLAP_FIXED_PAIR(3, 110, 111)
GOLAP_FIXED_PAIR(4, 112, 113)
GOLAP_FIXED_PAIR(5, 114, 115)
GOLAP_FIXED_PAIR(6, 116, 117)
GOLAP_FIXED_PAIR(7, 118, 119)
#undef GOLAP_FIXED_PAIR
constexpr int P2R_FIXED_FIRST = 104, P2R_FIXED_PAIRS = 8;   // (tests/test_isa_audit.py reads these)
template <int SET, int RPT, int R = 0>
__device__ __forceinline__ void ld8_fixed_tile(const uint2* sbase, uint32_t tid, uint32_t tn, uint32_t nt) {
    if constexpr (R < RPT) {
        const uint32_t i = (uint32_t)R * nt + tid;
        ld8_fixed<SET * RPT + R>(sbase, (i < tn ? i : tn - 1) * 8u);
        ld8_fixed_tile<SET, RPT, R + 1>(sbase, tid, tn, nt);
    }
}
template <int SET, int RPT, int R = 0>
__device__ __forceinline__ void ld8_take_tile(u32x2 (&t)[RPT]) {
    if constexpr (R < RPT) {
        t[R] = ld8_take<SET * RPT + R>();
        ld8_take_tile<SET, RPT, R + 1>(t);
    }
}
// 8-byte store of the lanes with ok != 0; the instruction itself is always issued (one vmcnt event per call)
__device__ __forceinline__ void st8_hidden_if(void* p, u32x2 v, uint32_t ok) {
    unsigned long long saved;
    asm volatile("v_cmp_ne_u32_e32 vcc, 0, %3\n\ts_and_saveexec_b64 %0, vcc\n\tglobal_store_dwordx2 %1, %2, off\n\ts_mov_b64 exec, %0"
                 : "=&s"(saved)
                 : "v"(p), "v"(v), "v"(ok)
                 : "memory", "vcc", "scc");
}
// the same with a work-group-uniform base (SGPR pair) and a 32-bit byte offset: no 64-bit address arithmetic per lane
__device__ __forceinline__ void st8_hidden_if_s(void* sbase, uint32_t off, u32x2 v, uint32_t ok) {
    unsigned long long saved;
    asm volatile("v_cmp_ne_u32_e32 vcc, 0, %4\n\ts_and_saveexec_b64 %0, vcc\n\tglobal_store_dwordx2 %1, %2, %3\n\ts_mov_b64 exec, %0"
                 : "=&s"(saved)
                 : "v"(off), "v"(v), "s"(sbase), "v"(ok)
                 : "memory", "vcc", "scc");
}
// 16-byte store of the lanes in `mask` (a wave-wide lane mask in scalar registers, e.g. from __ballot: no vector instruction is
// spent on the predicate), scalar base + 32-bit byte offset; always one vmcnt event
__device__ __forceinline__ void st16_hidden_mask_s(void* sbase, uint32_t off, u32x4 v, unsigned long long mask) {
    unsigned long long saved;
    // (s_nop: a store of more than 8 bytes reads its data registers for two more cycles on gfx940+; hipcc's hazard recogniser
    //  does not look inside asm statements, and without the wait states a vector instruction scheduled right behind this
    //  statement overwrote the first data register in time to be stored -- rows with a garbage first word, i.e. lost matches)
    asm volatile("s_and_saveexec_b64 %0, %4\n\tglobal_store_dwordx4 %1, %2, %3\n\ts_nop 1\n\ts_mov_b64 exec, %0"
                 : "=&s"(saved)
                 : "v"(off), "v"(v), "s"(sbase), "s"(mask)
                 : "memory", "scc");
}
// 4-byte store of the lanes with ok != 0 (one vmcnt event per call)
__device__ __forceinline__ void st4_hidden_if(void* p, uint32_t v, uint32_t ok) {
    unsigned long long saved;
    asm volatile("v_cmp_ne_u32_e32 vcc, 0, %3\n\ts_and_saveexec_b64 %0, vcc\n\tglobal_store_dword %1, %2, off\n\ts_mov_b64 exec, %0"
                 : "=&s"(saved)
                 : "v"(p), "v"(v), "v"(ok)
                 : "memory", "vcc", "scc");
}
// returning global atomic add of the active lanes (inactive lanes keep `ret`)
__device__ __forceinline__ void atomic_add_rtn_hidden(uint32_t& ret, uint32_t* p, uint32_t v) {
    asm volatile("global_atomic_add %0, %1, %2, off sc0" : "+v"(ret) : "v"(p), "v"(v) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory"); }
__device__ __forceinline__ void pin(u32x4& v) { asm volatile("" : "+v"(v)); }   // no consumer of v is scheduled above this point
__device__ __forceinline__ void pin(u32x2& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void pin(uint32_t& v) { asm volatile("" : "+v"(v)); }
// L2-served read of a word other work-groups may set (sc1 bypasses this CU's L1), hidden like the loads above.  (A scalar
// `s_load_dword ... glc` poll by every wave was tried instead: 4096 uncached reads of ONE address per tile round serialise
// at its memory channel -- the first pass ran 6 x slower.)
__device__ __forceinline__ void ld4_hidden_sc1(uint32_t& v, const uint32_t* p) { asm volatile("global_load_dword %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory"); }

}  // namespace golap

// LDS-DMA issue helper shared by the LDS-DMA engines (conv_dma.hip, conv_patch.hip).
#pragma once

namespace m2h {

// CNT LDS-DMA loads of 16 bytes per lane: lane l of load i writes LDS bytes [dst + i*step + 16 l, +16) from src[i] (per-lane
// pointers).  M0 (the DMA's LDS base) is written and restored inside the statement (cdna_hip_programming.md, inline-asm rules);
// the compiler does not count these loads: completion is waited for with explicit vmcnt below.
template <int CNT>
static __device__ __forceinline__ void glds16_run(const char* const* src, unsigned dst, unsigned step) {
  unsigned keep;
  static_assert(CNT == 1 || CNT == 2 || CNT == 4, "load count");
  if constexpr (CNT == 1)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(src[0]), "s"(dst)
                 : "memory");
  else if constexpr (CNT == 2)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
                 "s_add_u32 m0, m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(src[0]), "v"(src[1]), "s"(dst), "s"(step)
                 : "memory", "scc");
  else
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
                 "s_add_u32 m0, m0, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\t"
                 "s_add_u32 m0, m0, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, off\n\t"
                 "s_add_u32 m0, m0, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(src[0]), "v"(src[1]), "v"(src[2]), "v"(src[3]), "s"(dst), "s"(step)
                 : "memory", "scc");
}


}  // namespace m2h

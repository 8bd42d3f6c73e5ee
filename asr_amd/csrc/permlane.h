// v_permlane32_swap / v_permlane16_swap (gfx950) as inline asm.  hipcc 7.2's __builtin_amdgcn_permlane{16,32}_swap folds the two result
// registers into one when both are consumed by the same instruction (r[0] + r[1] is emitted as v_add v, v1, v1: scripts/probe_ksplit_reduce.hip
// found all 64 lanes wrong), so the swap is written out.  `s_nop 1` inside the string = the two wait states a VALU write of either operand
// needs before the swap reads it (the compiler cannot see into the statement).
//   permlane32_swap(x, y): lanes 32-63 of x <-> lanes 0-31 of y   -> {a = new x, b = new y}
//   permlane16_swap(x, y): odd 16-lane rows of x <-> even rows of y
#pragma once
struct u32pair { float a, b; };
__device__ __forceinline__ u32pair permlane32_swap(float x, float y) {
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(x), "+v"(y));
  return {x, y};
}
__device__ __forceinline__ u32pair permlane16_swap(float x, float y) {
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(x), "+v"(y));
  return {x, y};
}

// pin_regs(x): an empty asm statement that takes the M elements of x as read-write register operands -- "all of these
// values are needed HERE".  hipcc otherwise sinks each LDS / global load down to its first use, and a loop that requests M
// values and then combines them becomes M dependent round trips (seen in the ISA of the resample2d forward: ds_read ->
// s_waitcnt lgkmcnt(0) -> fma, sixteen times per channel; round 5).  With the pin the requests are issued back to back
// and waited for once.  An asm statement takes at most 30 operands; M up to 30 here (generated: tools/gen_pin_regs.py).
#pragma once

namespace gfla {

template <typename A, int M>
__device__ __forceinline__ void pin_regs(A (&x)[M]) {
  static_assert(M >= 1 && M <= 30, "pin_regs: 1..30 values");
  if constexpr (M == 1) asm volatile("" : "+v"(x[0]));
  else if constexpr (M == 2) asm volatile("" : "+v"(x[0]), "+v"(x[1]));
  else if constexpr (M == 3) asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]));
  else if constexpr (M == 4) asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]));
  else if constexpr (M == 5) asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]));
  else if constexpr (M == 6) asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]));
  else if constexpr (M == 7) asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]));
  else if constexpr (M == 8) asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]));
  else if constexpr (M == 9) asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]), "+v"(x[8]));
  else if constexpr (M == 10) asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]), "+v"(x[8]), "+v"(x[9]));
  else if constexpr (M == 11) asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]), "+v"(x[8]), "+v"(x[9]), "+v"(x[10]));
  else if constexpr (M == 12) asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]), "+v"(x[8]), "+v"(x[9]), "+v"(x[10]), "+v"(x[11]));
  else if constexpr (M == 13) asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]), "+v"(x[8]), "+v"(x[9]), "+v"(x[10]), "+v"(x[11]), "+v"(x[12]));
  else if constexpr (M == 14) asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]), "+v"(x[8]), "+v"(x[9]), "+v"(x[10]), "+v"(x[11]), "+v"(x[12]), "+v"(x[13]));
  else if constexpr (M == 15) asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]), "+v"(x[8]), "+v"(x[9]), "+v"(x[10]), "+v"(x[11]), "+v"(x[12]), "+v"(x[13]), "+v"(x[14]));
  else if constexpr (M == 16) asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]), "+v"(x[8]), "+v"(x[9]), "+v"(x[10]), "+v"(x[11]), "+v"(x[12]), "+v"(x[13]), "+v"(x[14]), "+v"(x[15]));
  else if constexpr (M == 17) asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]), "+v"(x[8]), "+v"(x[9]), "+v"(x[10]), "+v"(x[11]), "+v"(x[12]), "+v"(x[13]), "+v"(x[14]), "+v"(x[15]), "+v"(x[16]));
  else if constexpr (M == 18) asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]), "+v"(x[8]), "+v"(x[9]), "+v"(x[10]), "+v"(x[11]), "+v"(x[12]), "+v"(x[13]), "+v"(x[14]), "+v"(x[15]), "+v"(x[16]), "+v"(x[17]));
  else if constexpr (M == 19) asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]), "+v"(x[8]), "+v"(x[9]), "+v"(x[10]), "+v"(x[11]), "+v"(x[12]), "+v"(x[13]), "+v"(x[14]), "+v"(x[15]), "+v"(x[16]), "+v"(x[17]), "+v"(x[18]));
  else if constexpr (M == 20) asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]), "+v"(x[8]), "+v"(x[9]), "+v"(x[10]), "+v"(x[11]), "+v"(x[12]), "+v"(x[13]), "+v"(x[14]), "+v"(x[15]), "+v"(x[16]), "+v"(x[17]), "+v"(x[18]), "+v"(x[19]));
  else if constexpr (M == 21) asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]), "+v"(x[8]), "+v"(x[9]), "+v"(x[10]), "+v"(x[11]), "+v"(x[12]), "+v"(x[13]), "+v"(x[14]), "+v"(x[15]), "+v"(x[16]), "+v"(x[17]), "+v"(x[18]), "+v"(x[19]), "+v"(x[20]));
  else if constexpr (M == 22) asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]), "+v"(x[8]), "+v"(x[9]), "+v"(x[10]), "+v"(x[11]), "+v"(x[12]), "+v"(x[13]), "+v"(x[14]), "+v"(x[15]), "+v"(x[16]), "+v"(x[17]), "+v"(x[18]), "+v"(x[19]), "+v"(x[20]), "+v"(x[21]));
  else if constexpr (M == 23) asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]), "+v"(x[8]), "+v"(x[9]), "+v"(x[10]), "+v"(x[11]), "+v"(x[12]), "+v"(x[13]), "+v"(x[14]), "+v"(x[15]), "+v"(x[16]), "+v"(x[17]), "+v"(x[18]), "+v"(x[19]), "+v"(x[20]), "+v"(x[21]), "+v"(x[22]));
  else if constexpr (M == 24) asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]), "+v"(x[8]), "+v"(x[9]), "+v"(x[10]), "+v"(x[11]), "+v"(x[12]), "+v"(x[13]), "+v"(x[14]), "+v"(x[15]), "+v"(x[16]), "+v"(x[17]), "+v"(x[18]), "+v"(x[19]), "+v"(x[20]), "+v"(x[21]), "+v"(x[22]), "+v"(x[23]));
  else if constexpr (M == 25) asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]), "+v"(x[8]), "+v"(x[9]), "+v"(x[10]), "+v"(x[11]), "+v"(x[12]), "+v"(x[13]), "+v"(x[14]), "+v"(x[15]), "+v"(x[16]), "+v"(x[17]), "+v"(x[18]), "+v"(x[19]), "+v"(x[20]), "+v"(x[21]), "+v"(x[22]), "+v"(x[23]), "+v"(x[24]));
  else if constexpr (M == 26) asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]), "+v"(x[8]), "+v"(x[9]), "+v"(x[10]), "+v"(x[11]), "+v"(x[12]), "+v"(x[13]), "+v"(x[14]), "+v"(x[15]), "+v"(x[16]), "+v"(x[17]), "+v"(x[18]), "+v"(x[19]), "+v"(x[20]), "+v"(x[21]), "+v"(x[22]), "+v"(x[23]), "+v"(x[24]), "+v"(x[25]));
  else if constexpr (M == 27) asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]), "+v"(x[8]), "+v"(x[9]), "+v"(x[10]), "+v"(x[11]), "+v"(x[12]), "+v"(x[13]), "+v"(x[14]), "+v"(x[15]), "+v"(x[16]), "+v"(x[17]), "+v"(x[18]), "+v"(x[19]), "+v"(x[20]), "+v"(x[21]), "+v"(x[22]), "+v"(x[23]), "+v"(x[24]), "+v"(x[25]), "+v"(x[26]));
  else if constexpr (M == 28) asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]), "+v"(x[8]), "+v"(x[9]), "+v"(x[10]), "+v"(x[11]), "+v"(x[12]), "+v"(x[13]), "+v"(x[14]), "+v"(x[15]), "+v"(x[16]), "+v"(x[17]), "+v"(x[18]), "+v"(x[19]), "+v"(x[20]), "+v"(x[21]), "+v"(x[22]), "+v"(x[23]), "+v"(x[24]), "+v"(x[25]), "+v"(x[26]), "+v"(x[27]));
  else if constexpr (M == 29) asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]), "+v"(x[8]), "+v"(x[9]), "+v"(x[10]), "+v"(x[11]), "+v"(x[12]), "+v"(x[13]), "+v"(x[14]), "+v"(x[15]), "+v"(x[16]), "+v"(x[17]), "+v"(x[18]), "+v"(x[19]), "+v"(x[20]), "+v"(x[21]), "+v"(x[22]), "+v"(x[23]), "+v"(x[24]), "+v"(x[25]), "+v"(x[26]), "+v"(x[27]), "+v"(x[28]));
  else if constexpr (M == 30) asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]), "+v"(x[8]), "+v"(x[9]), "+v"(x[10]), "+v"(x[11]), "+v"(x[12]), "+v"(x[13]), "+v"(x[14]), "+v"(x[15]), "+v"(x[16]), "+v"(x[17]), "+v"(x[18]), "+v"(x[19]), "+v"(x[20]), "+v"(x[21]), "+v"(x[22]), "+v"(x[23]), "+v"(x[24]), "+v"(x[25]), "+v"(x[26]), "+v"(x[27]), "+v"(x[28]), "+v"(x[29]));
}

}  // namespace gfla

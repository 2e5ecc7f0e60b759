#!/usr/bin/env python3
"""Generates csrc/pin_regs.h (the operand lists of the register-pinning asm statements)."""
import os
import sys

HEAD = '''// pin_regs(x): an empty asm statement that takes the M elements of x as read-write register operands -- "all of these
// values are needed HERE".  hipcc otherwise sinks each LDS / global load down to its first use, and a loop that requests M
// values and then combines them becomes M dependent round trips (seen in the ISA of the resample2d forward: ds_read ->
// s_waitcnt lgkmcnt(0) -> fma, sixteen times per channel; round 5).  With the pin the requests are issued back to back
// and waited for once.  An asm statement takes at most 30 operands; M up to 30 here (generated: tools/gen_pin_regs.py).
#pragma once

namespace gfla {

template <typename A, int M>
__device__ __forceinline__ void pin_regs(A (&x)[M]) {
  static_assert(M >= 1 && M <= 30, "pin_regs: 1..30 values");
'''
out = [HEAD]
for m in range(1, 31):
    ops = ", ".join('"+v"(x[%d])' % i for i in range(m))
    out.append('  %sif constexpr (M == %d) asm volatile("" : %s);\n' % ("" if m == 1 else "else ", m, ops))
out.append("}\n\n}  // namespace gfla\n")
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "global_flow_local_attention_amd", "csrc", "pin_regs.h")
open(path, "w").write("".join(out))

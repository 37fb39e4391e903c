// Phase stamps of the wavefront-per-agent control kernel (csrc/control_wave_impl.hpp) for the A/B library only:
// `make -C ergodic_exploration_amd/csrc AB=1` pre-includes this file (-include), which turns the kernel's phase
// markers into shader-clock stamps.  Lane 0 of every wavefront records into p.dbg [agent][16]
// (tools/phase_timing.py through eea_debug_phase_timing).  The product build never sees this file.
#pragma once

#define EEA_WSTAMP(n)                                                                                   \
  do {                                                                                                  \
    if (p.dbg != nullptr && lane == 0) p.dbg[static_cast<size_t>(b) * 16 + (n)] = static_cast<long long>(__builtin_readcyclecounter()); \
  } while (0)
// slots 10 / 11: the constant 100 MHz counter at the wavefront's start / end; 12: HW_ID (which SIMD it ran on)
#define EEA_WSTAMP_RT(n)                                                                                \
  do {                                                                                                  \
    if (p.dbg != nullptr && lane == 0) p.dbg[static_cast<size_t>(b) * 16 + (n)] = static_cast<long long>(__builtin_amdgcn_s_memrealtime()); \
  } while (0)
#define EEA_WSTAMP_HWID(n)                                                                              \
  do {                                                                                                  \
    if (p.dbg != nullptr && lane == 0) {                                                                \
      unsigned hw_, xcc_;                                                                               \
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_));                                 \
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_));                               \
      p.dbg[static_cast<size_t>(b) * 16 + (n)] = static_cast<long long>(hw_) | (static_cast<long long>(xcc_) << 32); \
    }                                                                                                   \
  } while (0)

/* Entry points that exist only in the A/B library (make -C ergodic_exploration_amd/csrc AB=1 ->
 * libergodic_amd_ab.so): diagnostics and superseded kernels kept as measurement baselines.
 * NOT part of the product ABI (include/ergodic_amd.h). */
#ifndef ERGODIC_AMD_AB_H
#define ERGODIC_AMD_AB_H
#include "../../include/ergodic_amd.h"
#ifdef __cplusplus
extern "C" {
#endif
/* same as eea_control_batch for an fp64 engine, through the instrumented builds of the control kernels, which
 * record the shader clock at the phase boundaries (tools/phase_timing.py):
 *   wavefront-per-agent kernel (default path): lane 0 of the agent's wavefront, 10 stamps, d_stamps [B][16] int64;
 *   workgroup-per-agent kernel (eea_set_option(EEA_OPT_CONTROL_KERNEL, 1) or a shape the wavefront kernel does
 *   not take; K = 10 and 30): every wavefront, 12 stamps, d_stamps [B][4][16] int64. */
eea_status eea_debug_phase_timing(eea_engine* e, unsigned B, const eea_batch_io* io, void* stream,
                                  long long* d_stamps);
#ifdef __cplusplus
}
#endif
#endif

/* Entry points that exist only in the A/B library (make -C ergodic_exploration_amd/csrc AB=1 ->
 * libergodic_amd_ab.so): diagnostics and superseded kernels kept as measurement baselines.
 * NOT part of the product ABI (include/ergodic_amd.h). */
#ifndef ERGODIC_AMD_AB_H
#define ERGODIC_AMD_AB_H
#include "../../include/ergodic_amd.h"
#ifdef __cplusplus
extern "C" {
#endif
/* same as eea_control_batch for an fp64, K = 10 engine, through an instrumented build of the
 * workgroup-per-agent control kernel that records the shader clock of every wavefront at 12 phase
 * boundaries: d_stamps [B][4][16] int64 (tools/phase_timing.py).
 * Environment knobs of the A/B library: EEA_CONTROL_IMPL=v1 (first control kernel),
 * EEA_PHIK_IMPL=valu (per-column phi_k pass). */
eea_status eea_debug_phase_timing(eea_engine* e, unsigned B, const eea_batch_io* io, void* stream,
                                  long long* d_stamps);
#ifdef __cplusplus
}
#endif
#endif

/*
 * ergodic_amd.h -- C ABI of the MI355X (gfx950) ergodic receding-horizon engine.
 *
 * This is the drop-in boundary for the hot path of bostoncleek/ergodic_exploration:
 * one call of `ErgodicControl<ModelT>::control` (ergodic_control.hpp:224-311) plus the
 * phi_k rebuild `configTarget` (ergodic_control.hpp:362-416), for ModelT in
 * {models::Omni, models::SimpleCart}.  The reference has no FFI layer (its boundary is
 * the C++ template class); each entry point below names the reference interface it
 * replaces (file:line relative to the reference root).  INTEGRATION.md shows the binding a
 * maintainer adds on the reference side.
 *
 * Conventions
 *  - plain C types only; every function returns an eea_status (0 = ok); no exceptions
 *    cross the ABI.  eea_last_error() gives the message of the last failure on the
 *    calling thread.
 *  - "real" = double (EEA_PREC_F64) or float (EEA_PREC_F32), fixed per engine.
 *  - matrices use the reference's Armadillo layout: a "3 x T" matrix is T contiguous
 *    [x, y, theta] (or [vx, vy, w]) triples (column-major).
 *  - pointers named d_* are DEVICE pointers (HBM, caller-owned, same device as the
 *    engine); pointers named h_* / plain arrays are host memory.
 *  - `stream` is a hipStream_t passed as void* (NULL = the default stream); batch calls
 *    are asynchronous on it.  One engine per host thread (reentrant across engines).
 */
#ifndef ERGODIC_AMD_H
#define ERGODIC_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EEA_ABI_VERSION 6

/* models usable with ErgodicControl (Cart/Mecanum cannot run under it) */
enum { EEA_MODEL_OMNI = 0,        /* models::Omni        models/omni.hpp:164-215 */
       EEA_MODEL_SIMPLE_CART = 1  /* models::SimpleCart  models/cart.hpp:152-206 */ };

enum { EEA_PREC_F64 = 0, EEA_PREC_F32 = 1 };

typedef enum {
  EEA_OK = 0,
  EEA_ERR_INVALID_ARGUMENT = 1, /* reference: std::invalid_argument from a constructor */
  EEA_ERR_INVALID_TWIST = 2,    /* reference: SimpleCart::operator() throws, cart.hpp:167-170 */
  EEA_ERR_UNSUPPORTED = 3,      /* size outside what the kernels are built for */
  EEA_ERR_HIP = 4,              /* HIP runtime failure (no device, OOM, launch error) */
  EEA_ERR_NO_TARGET = 5,        /* control requested before any target was set */
  EEA_ERR_TIMEOUT = 6           /* per-agent status of the device-bound exchange: the shared c_k an agent waited for
                                   inside the kernel did not arrive (eea_batch_io::d_ck_flag); it used its own c_k */
} eea_status;

typedef struct eea_engine eea_engine;

/* Constructor arguments of ErgodicControl (ergodic_control.hpp:90-94, 187-222).
 * collision / buffer_size / batch_size are host-side concerns (the collision object is
 * never read on this path; replay-memory sampling stays on the host, see mem_cols). */
typedef struct {
  int model;          /* EEA_MODEL_* */
  int precision;      /* EEA_PREC_* */
  int device;         /* HIP device ordinal */
  double dt;          /* time step in integration */
  double horizon;     /* control horizon; steps = (unsigned)|horizon/dt| (truncating) */
  double resolution;  /* target grid resolution [m] */
  double expl_weight; /* ergodic exploration weight */
  unsigned num_basis; /* K: cosine basis functions per dimension (K^2 modes) */
  double Rinv[9];     /* inverse control weights, column-major 3x3 */
  double umin[3];     /* body twist lower limits */
  double umax[3];     /* body twist upper limits */
} eea_config;

/* ---- life cycle -------------------------------------------------------------------- */
/* ErgodicControl::ErgodicControl (ergodic_control.hpp:187-222).  EEA_ERR_INVALID_ARGUMENT
 * if steps == 1 (the reference throws) or steps == 0 (the reference has UB). */
eea_status eea_create(const eea_config* cfg, eea_engine** out);
void eea_destroy(eea_engine* e);
const char* eea_last_error(void);
unsigned eea_abi_version(void);

/* Process-wide dispatch options (no reference counterpart; every default reproduces the engine's own choice).
 * They replace the environment knobs of ABI 2: the library reads no environment variable. */
enum { EEA_OPT_CONTROL_KERNEL = 0,    /* 0 = automatic (wavefront-per-agent kernel where eligible), 1 = the
                                         workgroup-per-agent kernel for every control call */
       EEA_OPT_WORKGROUP_THREADS = 1, /* threads per agent of the workgroup-per-agent kernel: 0 = by horizon,
                                         or 64 / 128 / 256 */
       EEA_OPT_COLLISION_IMPL = 2,    /* Collision::collisionCheck implementation: 0 = cost model, 1 = ring
                                         search, 2 = inflated map (both bit-exact) */
       EEA_OPT_MAILBOX_POLL = 3,      /* eea_control: 1 = poll the completion word (default), 0 = wait for the stream */
       EEA_OPT_REBUILD_IMPL = 4,      /* configTarget rebuild of a Gaussian target: 0 = automatic (per-axis factors, one
                                         launch of one workgroup), 1 = fill the grid and stream it (Target::fill +
                                         Basis::spatialCoeff as two / three launches: what explicit grids take) */
       EEA_OPT_AGENT_LANES = 5,       /* lanes of a wavefront per agent in eea_control_batch[_steps] (short horizons share
                                         a wavefront: T <= 4 lanes, fp64, K = 5 / 10): 0 = by batch size and horizon
                                         (cost model), 64 = one wavefront per agent always, 8 / 16 / 32 = that group size
                                         wherever it is eligible (tests, A/B) */
       EEA_OPT_RESIDENT_CONTROL = 6,  /* eea_control (one robot, one control() per tick): 1 = a RESIDENT workgroup serves the calls
                                         from a host-mapped mailbox instead of one launch per call (the launch round trip is
                                         18-22 us whatever the shape).  It leaves by itself after EEA_OPT_RESIDENT_IDLE_MS
                                         without a call, on eea_resident_stop and on eea_destroy; while it is there,
                                         device-wide waits (hipDeviceSynchronize, hipFree) of the process take up to that
                                         long.  0 (default) = one launch per call */
       EEA_OPT_RESIDENT_IDLE_MS = 7,  /* how long the resident workgroup waits for the next call before it leaves, in
                                         milliseconds: 1 .. 60000, default 250 (a 10 Hz loop, exploration.hpp:232, keeps it
                                         alive).  Read when a resident workgroup is started */
       EEA_OPT_COUNT = 8 };
eea_status eea_set_option(int option, int value);
int eea_get_option(int option);

unsigned eea_steps(const eea_engine* e);      /* T = steps_ (ergodic_control.hpp:199) */
/* Lanes of a wavefront one agent of a plain eea_control_batch call of B agents occupies under the current options: 64 = one
 * wavefront per agent, 8 / 16 / 32 = several agents per wavefront (short horizons), 0 = the workgroup-per-agent kernel.
 * No reference counterpart (introspection for tests and capacity planning). */
unsigned eea_batch_agent_lanes(const eea_engine* e, unsigned B);
unsigned eea_num_modes(const eea_engine* e);  /* K^2 */
size_t eea_real_size(const eea_engine* e);    /* 8 or 4 */
double eea_time_step(const eea_engine* e);    /* ErgodicControl::timeStep, :350-354 */

/* ---- target distribution ----------------------------------------------------------- */
/* ErgodicControl::setTarget with a Target built from Gaussians (ergodic_control.hpp:356-360,
 * target.hpp:68-70): mu, sigma = 2 doubles per Gaussian, map frame.  As in the reference,
 * phi_k is NOT recomputed until the map extent changes. */
eea_status eea_set_target_gaussians(eea_engine* e, unsigned n, const double* mu, const double* sigma);

/* Basis::spatialCoeff on an explicit target grid (basis.hpp:99, basis.cpp:122-133):
 * phi_vals holds nx*ny reals, x fastest, on the grid configTarget builds for a domain
 * lx x ly (coordinates j*resolution by accumulation).  Sets lx, ly and phi_k directly
 * (no normalisation, as spatialCoeff).  phi_vals is a device pointer if on_device != 0. */
eea_status eea_set_target_grid(eea_engine* e, unsigned nx, unsigned ny, const void* phi_vals,
                               int on_device, double lx, double ly, void* stream);

/* Grid-tiled form of the same (multi-GPU target grids, BASELINE config 5): this rank holds rows
 * [row0, row0 + nrows) of an nx x ny_total grid in d_phi_rows (device, x fastest) and gets its
 * K^2 partial sums in d_phik_partial (device, real).  The caller adds the partials of all ranks
 * (one all-reduce of K^2 reals over RCCL) and installs the result with eea_set_phik. */
eea_status eea_spatial_coeff_rows(eea_engine* e, unsigned nx, unsigned ny_total, unsigned row0,
                                  unsigned nrows, const void* d_phi_rows, double lx, double ly,
                                  void* d_phik_partial, void* stream);
/* Target from an occupancy grid (BASELINE config 5: the in-tree surrogate of the mutual-information
 * map): phi(cell) = entropy(cell / 100.0) (numerics.hpp:164-179 with GridMap::getCell, grid.cpp:176-184),
 * normalised to sum 1 (the idiom of Target::fill, target.cpp:87), then Basis::spatialCoeff
 * (basis.cpp:122-133).  occ is the nav_msgs::OccupancyGrid::data buffer as GridMap holds it
 * (grid.cpp:63-94): int8, row-major, x fastest, nx*ny cells on the grid configTarget builds for a
 * domain lx x ly.  The entropy is applied inside the streaming kernel (one byte per cell read from
 * HBM, no fp64 target grid is materialised) and the normaliser is the (0,0) coefficient of the
 * un-normalised sums.  Sets lx, ly and phi_k.  occ is a device pointer if on_device != 0. */
eea_status eea_set_target_occupancy(eea_engine* e, unsigned nx, unsigned ny, const int8_t* occ,
                                    int on_device, double lx, double ly, void* stream);
/* Grid-tiled form (rows tiled across GPUs): this rank holds rows [row0, row0 + nrows) in d_occ_rows
 * (device) and gets its K^2 UN-normalised partial sums in d_sums_partial (device, real).  The caller
 * adds the partials of all ranks (one all-reduce of K^2 reals), divides by element 0 (the sum of the
 * entropies) and installs the result with eea_set_phik. */
eea_status eea_spatial_coeff_occupancy_rows(eea_engine* e, unsigned nx, unsigned ny_total, unsigned row0,
                                            unsigned nrows, const int8_t* d_occ_rows, double lx, double ly,
                                            void* d_sums_partial, void* stream);
/* installs phi_k (K^2 reals, device pointer if on_device != 0) for a domain lx x ly */
eea_status eea_set_phik(eea_engine* e, const void* phik, int on_device, double lx, double ly);
/* the last step of a grid-tiled occupancy target without a host round trip: phi_k = d_sums / d_sums[0], where d_sums
 * (device, K^2 reals) holds the un-normalised sums of eea_spatial_coeff_occupancy_rows after the all-reduce over the
 * ranks (element 0, mode (0,0), is the sum of the cell entropies: the normaliser of target.cpp:87).  Asynchronous
 * on `stream`; control calls enqueued on the same stream afterwards use the new phi_k. */
eea_status eea_set_phik_from_sums(eea_engine* e, const void* d_sums, double lx, double ly, void* stream);

/* ErgodicControl::configTarget (ergodic_control.hpp:362-416): refreshes map_pos; rebuilds
 * phi_k (Target::fill target.cpp:78-89 + Basis::spatialCoeff) only when the extent changed
 * by >= 1e-12.  *rebuilt (optional) reports whether the rebuild ran. */
eea_status eea_config_domain(eea_engine* e, double xmin, double xmax, double ymin, double ymax,
                             int* rebuilt, void* stream);
/* The same without the host wait: a rebuild (two or three launches) is only ENQUEUED on `stream` and the call returns;
 * control calls on the same stream afterwards are ordered by the stream, control calls on other streams are made to
 * wait for it by the engine (one hipStreamWaitEvent while it is in flight), the host-side getters wait for it.  The
 * caller orders the rebuild behind control calls still in flight on OTHER streams (they read the phi_k it replaces).
 * eea_control (single agent) uses this form internally. */
eea_status eea_config_domain_async(eea_engine* e, double xmin, double xmax, double ymin, double ymax,
                                   int* rebuilt, void* stream);

/* phi_k / lambda_k (basis.cpp:69-75) as doubles on the host, K^2 entries, col = k2*K + k1 */
eea_status eea_get_phik(eea_engine* e, double* h_phik);
eea_status eea_get_lamdak(eea_engine* e, double* h_lamdak);
/* normalised target grid of the last rebuild (Target::fill output) : nx*ny doubles; sizes via
 * eea_target_grid_size */
eea_status eea_target_grid_size(const eea_engine* e, unsigned* nx, unsigned* ny);
eea_status eea_get_target_grid(eea_engine* e, double* h_phi_vals);

/* ---- the hot path: agent-batched ErgodicControl::control --------------------------- */
/* Buffers of one batched call; B agents, each an independent reference ErgodicControl.
 * All pointers are device pointers to `real`; optional ones may be NULL. */
typedef struct {
  const void* d_pose;     /* [B][3]     current state x (map frame)                     */
  void* d_ut;             /* [B][T][3]  in: previous controls (warm start); out: updated
                                        controls.  The shift-left-by-one of :233-234
                                        happens inside the call                        */
  const void* d_mem_cols; /* [B][mem_stride][3] optional: the columns
                                        ReplayBuffer::sampleMemory would prepend
                                        (buffer.cpp:64-111), map frame                 */
  const int* d_n_mem;     /* [B] optional: valid columns per agent (<= mem_stride)     */
  unsigned mem_stride;    /* columns reserved per agent in d_mem_cols                  */
  void* d_u0;             /* [B][3]     out: ut.col(0) after the update (:310)         */
  void* d_traj;           /* [B][T][3]  out, optional: rk4_.solve rollout (:237)       */
  void* d_ck;             /* [B][K^2]   out, optional: trajectory coefficients (:267)  */
  void* d_edx;            /* [B][T][3]  out, optional: gradErgodicMetric (:270)        */
  void* d_bdx;            /* [B][T][3]  out, optional: gradBarrier (:273)              */
  void* d_rhot;           /* [B][T][3]  out, optional: co-state (:277)                 */
  int* d_status;          /* [B]        out, optional: per-agent eea_status            */
  const void* d_ck_shared;/* [K^2]      in, optional (ABI 2): decentralised-consensus mode.
                                        When set, fourier_diff = lamdak % (ck - phik)
                                        (:422) is formed with this shared c_k (e.g. the mean
                                        of all agents' c_k) instead of the agent's own; d_ck
                                        still receives the own c_k.  No counterpart in the
                                        single-agent reference: the semantics are those of its
                                        README ref. [2] (README.md:225-227).  NULL = reference
                                        behaviour.  With ck_shared_parts > 0 the buffer holds
                                        sum records instead (below)                           */
  void* d_ck_rec;         /* [B][eea_ck_record_len] out, optional (ABI 3): per-agent sum records
                                        [c_k (K^2 reals), 1, zero padding to an even length]; an
                                        agent rejected with EEA_ERR_INVALID_TWIST gets an all-zero
                                        record.  eea_ck_records_sum adds the B records in ONE small
                                        launch (fixed order); the result -- sums and agent count --
                                        is what d_ck_shared takes with ck_shared_parts > 0        */
  unsigned ck_shared_parts;/* ABI 3: 0 = d_ck_shared holds K^2 consensus values (ABI 2 form).
                                        n >= 1 = d_ck_shared holds n consecutive sum records
                                        (eea_ck_records_sum of earlier calls -- e.g. one per agent
                                        group -- all-reduced over the ranks or not): the kernel uses
                                        c_bar[m] = (sum_i rec_i[m]) x (1 / sum_i rec_i[K^2]) -- the
                                        reciprocal formed once per wavefront (ABI 6; a quotient per
                                        mode before: <= 1 ulp apart) --: no divide launch, and the
                                        exchange is one all-reduce of n records */
  /* ABI 4 -- device-bound exchange: producers and consumers of the shared c_k meet on the DEVICE, no host wait and no
   * stream wait anywhere (eea_comm_records_exchange_bound).  All optional (NULL / 0 = the forms above). */
  unsigned* d_rec_ready;  /* [B] out (with d_ck_rec): rec_ready[b] = rec_seq once agent b's record is visible
                                        device-wide -- written behind the drained, write-through record, about half
                                        way through the agent's wavefront; eea_ck_records_sum_bound polls these marks
                                        instead of waiting for the kernel to finish                              */
  unsigned rec_seq;       /* sequence number of this pass (any value that grows from pass to pass, mod 2^32)     */
  const unsigned* d_ck_flag;/* in (with d_ck_shared): the kernel waits until *d_ck_flag has reached ck_flag_seq
                                        ((int)(*d_ck_flag - ck_flag_seq) >= 0) right before the first use of the shared
                                        c_k, and reads d_ck_shared past its L1: "late binding" -- the call can be
                                        launched before the exchange that fills d_ck_shared has run.  Bounded (about a
                                        second): then d_status[b] = EEA_ERR_TIMEOUT and the agent uses its own
                                        c_k.  The waiting wavefronts hold their execution slots: see
                                        eea_comm_records_exchange_bound for what must fit beside them             */
  unsigned ck_flag_seq;
  const int* d_skip;      /* [B] in, optional (ABI 5): agents with a non-zero entry are LEFT OUT of the call -- nothing of
                                        theirs is read or written, their warm start does not advance (a robot that
                                        follows a dynamic-window twist does not call control(), exploration.hpp:230-236;
                                        eea_tick_batch fills it)                                                 */
  int rec_per_wavefront;  /* ABI 6, with d_ck_rec: non-zero = where several agents share a wavefront (short horizons,
                                        eea_batch_agent_lanes < 64) the launch writes ONE record per wavefront -- [sum of
                                        the c_k of its accepted agents, their number, pad], the agents added in agent
                                        order -- instead of one per agent: d_ck_rec [eea_batch_record_count][record_len],
                                        d_rec_ready one mark per record.  Records are closed under addition, so
                                        eea_ck_records_sum / _bound / eea_comm_records_exchange_* take them as they are,
                                        with the record count in place of B: 4 - 8 x fewer bytes through the sum at the
                                        batch sizes that fill the chip with packed agents (27 MB per pass at 32 768
                                        agents otherwise: the sum, not the control kernel, then sets the pass time).
                                        Launches of one agent per wavefront or workgroup write per-agent records either
                                        way (count = B).  The sum of the records equals the per-agent sum up to the
                                        order of the additions (fixed, so still reproducible run to run)            */
} eea_batch_io;

/* how many records a launch of B agents writes to d_ck_rec (and marks in d_rec_ready) with rec_per_wavefront set: the number
 * of its wavefronts where agents share one (ceil(B / (64 / eea_batch_agent_lanes))), B otherwise.  0 for a null engine. */
unsigned eea_batch_record_count(const eea_engine* e, unsigned B);

/* length in reals of one sum record (eea_batch_io::d_ck_rec): K^2 + 1 rounded up to an even number */
unsigned eea_ck_record_len(const eea_engine* e);
/* d_sum [eea_ck_record_len] = sum of the B per-agent records d_ck_rec [B][eea_ck_record_len] (element K^2 = number
 * of agents that count): ONE launch of single-wavefront workgroups -- groups of 32 agents in agent order, 8 group
 * records per level-1 record, the level-1 records in order, finished by last-arrival tickets inside the launch -- a
 * fixed summation tree, run-to-run deterministic.  Asynchronous on `stream` (typically an exchange stream beside the
 * compute streams: the wavefronts use no LDS and <= 32 registers and are resident BESIDE a full fp64 K <= 10 control
 * kernel).  Concurrent calls on one engine must use distinct d_sum buffers. */
eea_status eea_ck_records_sum(eea_engine* e, unsigned B, const void* d_ck_rec, void* d_sum, void* stream);
/* ABI 6: the same sum with a CALLER-OWNED workspace -- d_ws (eea_ck_records_sum_ws_bytes: *ws_bytes) and d_tickets (*ticket_bytes,
 * zeroed once; the tickets reset themselves) -- instead of the engine's per-d_sum workspace cache: nothing is allocated or
 * looked up in the call, so it can be captured into a hipGraph and replayed for as long as the caller keeps the buffers
 * (eea_consensus_plan below does).  Same summation tree, same bits. */
eea_status eea_ck_records_sum_ws_bytes(const eea_engine* e, unsigned B, size_t* ws_bytes, size_t* ticket_bytes);
eea_status eea_ck_records_sum_ws(eea_engine* e, unsigned B, const void* d_ck_rec, void* d_sum, void* d_ws, void* d_tickets,
                                 void* stream);
/* ABI 4, device-bound form: the same sum, but the launch does not have to be ordered behind the control kernels that
 * write the records -- every unit of the sum polls the ready marks of its 32 agents (d_rec_ready[b] - seq >= 0 mod 2^32,
 * eea_batch_io::d_rec_ready / rec_seq of the producing calls) and starts when they are there.  d_flag != NULL: *d_flag = seq
 * is published (write-through, behind the drained sum record) by the wavefront that completes the sum -- what
 * eea_batch_io::d_ck_flag of the consuming calls waits for.  Agents that never report within about a second make the
 * record's agent count negative (consumers then keep their own c_k and report EEA_ERR_TIMEOUT).  Same summation tree, same
 * bits as eea_ck_records_sum. */
eea_status eea_ck_records_sum_bound(eea_engine* e, unsigned B, const void* d_ck_rec, const unsigned* d_rec_ready, unsigned seq,
                                    void* d_sum, unsigned* d_flag, void* stream);
/* d_pub [eea_ck_record_len] = d_src written through, then *d_flag = seq: publishes a record that another kernel produced with
 * ordinary stores (e.g. an all-reduce over the ranks) to control kernels that are already running and wait for the flag. */
eea_status eea_publish_record(eea_engine* e, const void* d_src, void* d_pub, unsigned* d_flag, unsigned seq, void* stream);

/* One receding-horizon optimisation per agent (ergodic_control.hpp:224-311, without the
 * configTarget call: use eea_config_domain first).  Asynchronous on `stream`. */
eea_status eea_control_batch(eea_engine* e, unsigned B, const eea_batch_io* io, void* stream);

/* ABI 4: n_steps CONSECUTIVE receding-horizon optimisations per agent in ONE launch -- what n_steps calls of
 * eea_control_batch do when each call's pose comes from a row of d_pose: step n runs control() (ergodic_control.hpp:224-311)
 * from pose row n * pose_step_stride + b and the controls step n - 1 left in d_ut (the warm start of :233-234), and
 * writes u = ut.col(0) to d_u0 row n * u0_step_stride + b.  Strides are in agents: 0 = the same row every step (a fixed
 * pose; only the last step's u0 is kept), >= B = d_pose [n_steps][stride][3] / d_u0 [n_steps][stride][3] -- a logged
 * pose sequence replayed through the controller (the replay harness of exploration.hpp:197-292 per agent), a
 * closed-loop simulation driven from the host in chunks, or a throughput run.  Bitwise the same d_ut / d_u0 as the n_steps
 * separate calls (tests/test_gpu_multi_step.py).  The other per-step outputs (d_ck, d_ck_rec, d_traj, stage outputs,
 * d_status) hold the LAST step's values; d_ck_shared / replay-memory columns are read unchanged by every step; the
 * device-bound exchange fields (d_rec_ready / d_ck_flag) are refused with n_steps > 1 (EEA_ERR_UNSUPPORTED): a step would
 * wait inside the kernel for an exchange the host can only enqueue after this call, which needs truly concurrent
 * hardware queues -- every wait of that protocol is for work enqueued BEFORE the waiter.  The
 * agent's wavefront carries on with its own stored controls (read back through L2) instead of the host launching again:
 * no launch gap, no kernel tail, no cold loads between steps.  Horizons / bases outside the wavefront-per-agent kernel's
 * range (T > 256, K > 16 and != 20) are issued as n_steps launches on the stream with the same semantics. */
eea_status eea_control_batch_steps(eea_engine* e, unsigned B, const eea_batch_io* io, unsigned n_steps,
                                   unsigned pose_step_stride, unsigned u0_step_stride, void* stream);

/* Forward rollout only: ErgodicControl::optTraj / path (ergodic_control.hpp:313-342),
 * RungeKutta::solve (integrator.hpp:135-152).  d_ut is used as is (no shift). */
eea_status eea_rollout_batch(eea_engine* e, unsigned B, const void* d_pose, const void* d_ut,
                             void* d_traj, int* d_status, void* stream);

/* ---- multi-GPU exchange steps of the agent batch (RCCL over xGMI) -------------------- */
/* The reference is single-agent; an agent batch sharded over the GPUs of a node (one process
 * per GPU, contiguous agent blocks, no collective inside the control computation) exchanges the
 * per-agent trajectory coefficients c_k the way decentralised ergodic control shares them
 * (reference README ref. [2], README.md:225-227).  These calls are what a C++ host uses to shard
 * without Python; RCCL is bound at run time (no link dependency).  All asynchronous on `stream`.
 *
 * Rank 0 obtains an id (EEA_COMM_ID_BYTES bytes) and hands it to the other ranks by any means
 * (file, socket, MPI, torch.distributed); every rank then creates its communicator.
 * id == NULL with nranks == 1 gives a local communicator without RCCL. */
#define EEA_COMM_ID_BYTES 128
typedef struct eea_comm eea_comm;
/* Binds the collectives to the RCCL at `path` (dlopen, own symbol scope) instead of the one already mapped into the process
 * or the default librccl.so -- for deployments that carry several RCCL builds; the one-GPU tests and bench.py point it at the
 * test double tests/fake_rccl/librccl.so.1 to run several ranks (processes) on one device.  Process-wide, before the first
 * other eea_comm_* call (EEA_ERR_UNSUPPORTED once the library is bound). */
eea_status eea_comm_set_library(const char* path);
eea_status eea_comm_get_unique_id(void* id);
eea_status eea_comm_create(int device, int nranks, int rank, const void* id, eea_comm** out);
void eea_comm_destroy(eea_comm* c);
int eea_comm_rank(const eea_comm* c);
int eea_comm_nranks(const eea_comm* c);
/* ABI 6: the number of ranks THE COLLECTIVE LIBRARY reports for this communicator (ncclCommCount) -- what a multi-GPU run shows to
 * prove that RCCL saw every rank; 0 for a local communicator (no library behind it), -1 if the library cannot tell. */
int eea_comm_library_nranks(const eea_comm* c);
/* sums[m] = sum over the B local agents of d_ck[b][m], m < K^2, and sums[K^2] = B (K^2 + 1 reals,
 * device): the local half of the consensus reduction */
eea_status eea_ck_sum(eea_engine* e, unsigned B, const void* d_ck, void* d_sums, void* stream);
/* one in-place-capable ncclAllGather: d_ck_all [nranks * B_local][K^2] in rank order (equal shards) */
eea_status eea_comm_allgather_ck(eea_engine* e, eea_comm* c, unsigned B_local, const void* d_ck_local,
                                 void* d_ck_all, void* stream);
/* consensus c_k: d_ck_shared[m] = mean over ALL agents of all ranks of c_k[m] -- eea_ck_sum, one
 * ncclAllReduce(sum) of K^2 + 1 reals (808 B at K = 10), one divide; feed it to
 * eea_batch_io::d_ck_shared */
eea_status eea_comm_consensus_ck(eea_engine* e, eea_comm* c, unsigned B_local, const void* d_ck_local,
                                 void* d_ck_shared, void* stream);
/* Asynchronous forms: the exchange runs on a stream the communicator owns, ordered after everything enqueued on
 * `compute_stream` so far (the pass that produced c_k), so that the next pass on the compute stream overlaps it;
 * eea_comm_wait makes a stream wait for the exchange started in `slot` (0 .. EEA_COMM_SLOTS - 1; the caller
 * rotates slots together with its c_k / result buffers). */
#define EEA_COMM_SLOTS 8
eea_status eea_comm_consensus_ck_async(eea_engine* e, eea_comm* c, unsigned B_local, const void* d_ck_local,
                                       void* d_ck_shared, void* compute_stream, int slot);
eea_status eea_comm_allgather_ck_async(eea_engine* e, eea_comm* c, unsigned B_local, const void* d_ck_local,
                                       void* d_ck_all, void* compute_stream, int slot);
eea_status eea_comm_wait(eea_comm* c, int slot, void* stream);
/* The whole exchange of one pass of an agent batch stepped as n_streams agent groups, in one call, STREAM-ORDERED: the
 * communicator's own (highest-priority) stream waits for everything enqueued so far on EACH of the group streams (one event
 * per group), then eea_ck_records_sum over the B_local per-agent records d_ck_rec (eea_batch_io::d_ck_rec of the groups'
 * control calls), then the all-reduce of the sum record over the ranks (none with one rank).  d_sum [eea_ck_record_len] then
 * feeds eea_batch_io::d_ck_shared with ck_shared_parts = 1 on every rank; the consuming streams call eea_comm_wait(c, slot, ..).
 * The form for hosts that exchange at their control rate (the reference's nodes: 10 Hz). */
eea_status eea_comm_records_exchange_async(eea_engine* e, eea_comm* c, unsigned B_local, const void* d_ck_rec,
                                           void* d_sum, void* const* group_streams, unsigned n_streams, int slot);
/* ABI 4 -- the same exchange DEVICE-BOUND, for a consensus on EVERY pass at the device's own rate (a pass of 4096 agents
 * takes ~23 us: no host wait and no stream wait fits into it).  Nothing is ordered by the host: on the communicator's
 * stream, eea_ck_records_sum_bound (polls the agents' ready marks until d_rec_ready - seq >= 0 (mod 2^32): it runs while
 * the producing control kernels are still in their backward halves), with an RCCL communicator the all-reduce of the sum record over the ranks +
 * eea_publish_record, and *d_flag = seq behind the finished d_sum.  The consuming control calls are launched WITHOUT
 * waiting, with eea_batch_io::d_ck_shared = d_sum, ck_shared_parts = 1, d_ck_flag = d_flag, ck_flag_seq = seq: they wait
 * inside the kernel, right before the first use of the shared c_k.  A consensus of lag n passes = pass i consumes seq i - n;
 * lag 1 is the previous step's c_bar (decentralised ergodic control, reference README ref. [2]).
 * The caller rotates d_ck_rec / d_sum over >= lag + 2 buffers (slot = buffer index, < EEA_COMM_SLOTS); d_rec_ready [B_local]
 * and d_flag [1] may be shared by all of them (sequence numbers only grow; zero them once, and start the sequence at 1:
 * zeroed marks and flags satisfy seq 0 at once).  A per-agent EEA_ERR_TIMEOUT in a d_status buffer that is reused from pass
 * to pass STAYS until the caller clears it (calls with d_ck_flag do not reset a status of 6; every other call rewrites
 * d_status).  Every launch that waits for a flag must leave room for what it waits for: the producers of that flag (control kernels of other agent groups, the
 * record sum's single-wavefront workgroups, the all-reduce) have to become resident BESIDE the waiting wavefronts -- two
 * agent groups per GPU of at most half its execution slots each do (the fp64 K <= 10 instance leaves registers for the
 * sum beside a full set of control wavefronts); a waiter that cannot be served gives up after about a second
 * (EEA_ERR_TIMEOUT in d_status, own c_k), it never hangs.  Host threads: none; the calling thread issues 1-3 launches.
 * With an RCCL COMMUNICATOR in the exchange the producers include the collective kernel (hundreds of threads, ~100 registers,
 * LDS of its own): it does not fit beside a full set of control wavefronts.  Every agent group waiting on the device for its
 * flag is then a dead-lock at full occupancy (every agent times out: measured in round 5 with a kernel-shaped test double,
 * profiles/r05_two_ranks.txt), and ONE waiting group (the other ordered behind the exchange with eea_comm_wait -- the event is
 * recorded behind the published record) still stalled once in a few thousand passes.  With a communicator use the
 * STREAM-ORDERED exchange -- eea_comm_records_exchange_async + eea_comm_wait for every consuming group -- at a lag of >= 2
 * passes: nothing waits inside a kernel, so nothing can hold the slots its producer needs.  The device-bound form is for
 * exchanges WITHOUT a collective kernel (one rank / a local communicator). */
eea_status eea_comm_records_exchange_bound(eea_engine* e, eea_comm* c, unsigned B_local, const void* d_ck_rec,
                                           const unsigned* d_rec_ready, unsigned seq, void* d_sum, unsigned* d_flag, int slot);
/* ---- ABI 6: the GATED exchange -- the device-bound exchange with the flag wait in front of the launch, not inside it -------
 * eea_stream_wait_flag enqueues a one-wavefront kernel on `stream` that returns once *d_flag - seq >= 0 (mod 2^32); what is
 * enqueued behind it on that stream starts after it.  A consensus pass of an agent group is then
 *     eea_stream_wait_flag(d_flag, seq - lag, .., group stream)           the sum record of pass i - lag is published
 *     eea_control_batch(.. d_ck_rec, d_rec_ready, rec_seq = seq, d_ck_shared = that record, ck_shared_parts = 1; NO d_ck_flag ..)
 * and once per pass eea_comm_records_exchange_bound(.., seq, d_sum, d_flag, slot).  Launches only: no event, no stream wait,
 * no host wait (an event record + wait pair costs the host 4.6 us on this runtime, a small launch 0.7 - 2.6).  Unlike the
 * in-kernel wait (eea_batch_io::d_ck_flag) the waiting group holds ONE execution slot, not its half of the chip: the
 * collective kernel of an RCCL communicator always finds room, so this form is safe WITH a communicator at a lag >= 2 (every
 * wait is still for work enqueued before the waiter).  Bounded: after about a second the gate gives up, adds 1 to *d_timeouts
 * (optional) and lets the stream go on -- the consuming launches then read whatever the record's slot holds at that moment: a
 * time-out is an error to be reported (the producer never became resident), not a mode of operation.  No reference counterpart. */
/* (No engine argument: the launch goes to the calling thread's CURRENT device, which must be the stream's -- it is after any
 * eea_* call on an engine or communicator of that device.) */
eea_status eea_stream_wait_flag(const unsigned* d_flag, unsigned seq, unsigned* d_timeouts, void* stream);

/* ---- ABI 6: the consensus loop of a rank as ONE replayable device graph ----------------------------------------------------
 * The stream-ordered exchange above costs the host ~6 C-ABI calls and ~11 runtime calls per pass (two group launches, their
 * eea_comm_wait, eea_comm_records_exchange_async): 35-40 us of host time per 23 us pass (profiles/r05_exchange_modes.txt) -- the
 * host, not xGMI, limits a consensus on every pass.  A plan captures `passes_per_launch` consecutive passes of that SAME protocol
 * -- per pass and agent group one eea_control_batch (records out, the sum record of pass i - lag in: ck_shared_parts = 1), then
 * on the exchange branch the record sum and the all-reduce over the ranks (nothing without an RCCL communicator) -- into one
 * hipGraph; eea_consensus_plan_launch replays it with ONE runtime call.  Nothing waits inside a kernel (the rule of
 * eea_comm_records_exchange_bound for exchanges with a collective kernel holds by construction); lag >= 2 keeps the collective
 * off the critical path of a pass.  The plan owns the record / sum buffers (lag + 2 slots, rotating; one record per wavefront
 * where agents share one: eea_batch_io::rec_per_wavefront), the record sum's
 * workspaces, the group streams and events; the caller owns what eea_batch_io names (d_pose, d_ut, d_u0, optional d_mem_cols /
 * d_n_mem / mem_stride, d_status, d_skip per group: the exchange fields of group_io are ignored) and may rewrite the CONTENTS of
 * those buffers between launches (new poses), not the pointers.  Across launches the protocol continues: pass 0 of a launch
 * consumes the record of the last passes of the launch before (the first `lag` passes ever consume an empty record: own c_k).
 * With several ranks every rank creates and launches its plan the same number of times (the all-reduces pair up in order).
 * A plan refers to its engine and communicator: destroy it before either.
 * EEA_ERR_UNSUPPORTED: the collective library cannot be captured -- use the per-call form.  No reference counterpart
 * (decentralised ergodic control shares c_k, README.md:225-227). */
typedef struct eea_consensus_plan eea_consensus_plan;
typedef struct eea_consensus_desc {
  unsigned n_groups;             /* 1 .. 8 agent groups (contiguous slices of the rank's batch), each on a stream of its own */
  const unsigned* group_agents;  /* [n_groups] agents per group */
  const eea_batch_io* group_io;  /* [n_groups] the groups' buffers */
  unsigned lag;                  /* pass i consumes the sum record of pass i - lag: 1 .. EEA_COMM_SLOTS - 2 */
  unsigned passes_per_launch;    /* passes one launch replays; rounded UP to a multiple of the slot count lag + 2 */
} eea_consensus_desc;
eea_status eea_consensus_plan_create(eea_engine* e, eea_comm* c, const eea_consensus_desc* d, eea_consensus_plan** out);
/* replays the plan's passes, asynchronous on `stream`; launch a plan on ONE stream (its launches must run in order) */
eea_status eea_consensus_plan_launch(eea_consensus_plan* p, void* stream);
/* passes one launch replays (after rounding), and the device address of the sum record [eea_ck_record_len] the LAST pass of a
 * launch leaves (sum over all ranks of the agents' c_k, element K^2 = agent count); either pointer may be NULL */
eea_status eea_consensus_plan_info(const eea_consensus_plan* p, unsigned* passes_per_launch, const void** d_last_sum);
void eea_consensus_plan_destroy(eea_consensus_plan* p);

/* in-place ncclAllReduce(sum) of n reals: the K^2 partial sums of a grid-tiled phi_k
 * (eea_spatial_coeff_rows / eea_spatial_coeff_occupancy_rows) */
eea_status eea_comm_allreduce_sum(eea_engine* e, eea_comm* c, void* d_buf, unsigned n, void* stream);
/* the same on the communicator's own stream, ordered after `compute_stream` like the asynchronous forms above */
eea_status eea_comm_allreduce_sum_async(eea_engine* e, eea_comm* c, void* d_buf, unsigned n, void* compute_stream,
                                        int slot);

/* ---- single agent, host pointers (what ErgodicControl<ModelT>::control binds to) ---- */
/* vec control(const GridMap& grid, const vec& x) (ergodic_control.hpp:224-311) including
 * configTarget(grid): grid bounds in, u = ut.col(0) out.  The warm-start controls live in
 * the engine.  mem_cols: 3 x n_mem doubles (map frame) or NULL.  Synchronous. */
eea_status eea_control(eea_engine* e, double xmin, double xmax, double ymin, double ymax,
                       const double x[3], const double* h_mem_cols, unsigned n_mem,
                       double u_out[3]);
/* EEA_OPT_RESIDENT_CONTROL: tells the engine's resident workgroup (if one is there) to leave and returns when it has; the
 * next eea_control starts another one.  For a caller about to make device-wide waits (hipDeviceSynchronize, hipFree,
 * hipMalloc) that would otherwise take up to the idle time.  The warm-start controls are kept (they are in device memory
 * after every request).  No reference counterpart; EEA_OK also when nothing was resident. */
eea_status eea_resident_stop(eea_engine* e);
/* mat optTraj() const (ergodic_control.hpp:313-317): 3 x T doubles */
eea_status eea_opt_traj(eea_engine* e, double* h_traj);
/* read / overwrite the engine-held warm-start controls ut_ (3 x T doubles) */
eea_status eea_get_ut(eea_engine* e, double* h_ut);
eea_status eea_set_ut(eea_engine* e, const double* h_ut);

/* ---- Basis hot loops as free functions (Basis::trajCoeff / spatialCoeff) ----------- */
/* c_k = (1/n) sum_i f_k(xt_i)  (basis.cpp:109-120).  h_xt: rows x n column-major, rows >= 2 */
eea_status eea_basis_traj_coeff(int device, double lx, double ly, unsigned num_basis,
                                const double* h_xt, unsigned rows, unsigned n, double* h_ck);
/* phi_k = sum_p f_k(grid_p) phi_vals_p  (basis.cpp:122-133) for an arbitrary point list.
 * h_phi_grid: 2 x P column-major */
eea_status eea_basis_spatial_coeff(int device, double lx, double ly, unsigned num_basis,
                                   const double* h_phi_vals, const double* h_phi_grid, unsigned P,
                                   double* h_phik);

/* RungeKutta::solve (forward, integrator.hpp:135-152) for a body-twist model without an
 * engine: x0[3], h_ut 3 x steps, h_xt out 3 x steps, steps = (unsigned)|horizon/dt|.
 * EEA_ERR_INVALID_TWIST mirrors SimpleCart's throw. */
eea_status eea_rk4_rollout(int device, int model, double dt, double horizon, const double x0[3],
                           const double* h_ut, double* h_xt);
/* Target::fill (target.cpp:78-89) on an arbitrary point list: h_phi_grid 2 x P, trans[2];
 * h_phi_vals out, normalised to sum 1. */
eea_status eea_target_fill(int device, unsigned n_gauss, const double* mu, const double* sigma,
                           const double trans[2], const double* h_phi_grid, unsigned P,
                           double* h_phi_vals);

/* ---- next rows of the scope table: collision lookups ------------------------------- */
/* Collision::collisionCheck (collision.cpp:126-141) for P poses on one occupancy grid
 * (GridMap, grid.cpp:143-184).  d_grid: int8 row-major [ysize][xsize]; d_pose: [P][3]
 * doubles; d_hit: [P] ints (1 = collision).  Bit-exact integer semantics incl. the
 * world2Grid wrap of negative coordinates. */
typedef struct {
  double xmin, ymin, resolution;
  unsigned xsize, ysize;
  double boundary_radius, search_radius, obstacle_threshold, occupied_threshold;
} eea_collision_cfg;
eea_status eea_collision_check_batch(int device, const eea_collision_cfg* cfg, const int8_t* d_grid,
                                     const double* d_pose, unsigned P, int* d_hit, void* stream);
/* validate_control (numerics.hpp:312-330): integrate_twist rollout + collisionCheck per
 * step; d_x0 [P][3], d_u [P][3] doubles; d_valid [P] ints (1 = collision free). */
eea_status eea_validate_control_batch(int device, const eea_collision_cfg* cfg, const int8_t* d_grid,
                                      const double* d_x0, const double* d_u, double dt,
                                      double horizon, unsigned P, int* d_valid, void* stream);

/* ABI 6: integrate_twist (numerics.hpp:273-297) for P poses: d_out [P][3] = d_x0 + Rot(theta) * (the body-frame displacement of the
 * constant twist d_u over dt); the motion update of the replay harness / a simulated fleet between ticks, on the device so that
 * the tick loop needs no host round trip.  d_out may be d_x0.  normalize_heading != 0: the heading through normalize_angle_PI
 * (numerics.hpp:77-89) as the reference's callers do (validate_control :325); 0: as integrate_twist returns it. */
eea_status eea_integrate_twist_batch(int device, const double* d_x0, const double* d_u, double dt, unsigned P, double* d_out,
                                     int normalize_heading, void* stream);

/* DynamicWindow::control (dynamic_window.cpp:92-189), both overloads, for P robots on one grid:
 * velocity window from the current twist d_vb (dynamic_window.cpp:191-235), vx x vy x vth sample
 * grid built by repeated += (first strict minimum wins), constant-twist rollouts with a collision
 * check per step (objective, :237-286).
 *   mode "vref"  (d_xt_ref == NULL): cost = |vref - u|^2,            d_vref [P][3]
 *   mode "traj"  (d_xt_ref != NULL): cost = distance to the reference trajectory
 *                                     d_xt_ref [P][n_ref][3], time step dt_ref
 * d_u_opt [P][3] out, d_found [P] out (0 = "DWA Failed! Not even 1 solution found"). */
typedef struct {
  double dt, horizon, acc_dt, acc_lim_x, acc_lim_y, acc_lim_th;
  double max_vel_x, min_vel_x, max_vel_y, min_vel_y, max_rot_vel, min_rot_vel;
  unsigned vx_samples, vy_samples, vth_samples;
} eea_dwa_cfg;
eea_status eea_dwa_control_batch(int device, const eea_collision_cfg* ccfg, const eea_dwa_cfg* dcfg,
                                 const int8_t* d_grid, const double* d_x0, const double* d_vb,
                                 const double* d_vref, const double* d_xt_ref, unsigned n_ref,
                                 double dt_ref, unsigned P, double* d_u_opt, int* d_found, void* stream);

/* ---- one tick of the exploration loop for a FLEET (ABI 5) ----------------------------------------------------------
 * Exploration<ModelT>::control's loop body (exploration.hpp:220-279) for B robots on one shared occupancy grid, on the
 * device, one stream, no host round trip:
 *   1. a robot that follows a dynamic-window twist counts a step; after dwa_steps = (unsigned)|horizon / dt| of the DWA
 *      configuration it replans (:223-228);
 *   2. every robot that does not follow one: u = ErgodicControl::control(grid, pose) (:230-236) -- eea_control_batch with
 *      d_skip for the others (their warm start stays as it is, as in the reference);
 *   3. validate_control(collision, grid, pose, u, val_dt, val_horizon) (:238);
 *   4. on a predicted collision: a follower re-runs the dynamic window towards its own twist and stops following
 *      (:243-251); the others track optTraj() -- the rollout of the UPDATED controls -- and follow the result if one was
 *      found (:254-277).  u = 0 where the dynamic window finds nothing.
 * State carried between ticks, per robot, device memory owned by the caller, zero before the first tick: d_follow_dwa,
 * d_dwa_count, d_u.  addStateMemory (:209) stays on the caller's side: io->d_mem_cols / d_n_mem as for eea_control_batch.
 * io->d_u0, d_traj, d_skip are ignored (the tick supplies its own); everything else of io is passed to the control call.
 * fp64 engines only (poses and twists are the doubles the collision / DWA kernels take). */
typedef struct {
  int* d_follow_dwa;      /* [B]       in/out: follow_dwa (:191)                                                 */
  unsigned* d_dwa_count;  /* [B]       in/out: i (:194)                                                          */
  double* d_u;            /* [B][3]    in/out: the commanded twist u (kept while a DWA twist is followed)        */
  const double* d_vb;     /* [B][3]    in: body twists from odometry (vb_, :161-174)                             */
  const int8_t* d_grid;   /* [ysize][xsize] in: the occupancy grid (grid_)                                       */
  double* d_traj;         /* [B][T][3] scratch: optTraj() of the robots that ran control()                       */
  int* d_valid;           /* [B]       scratch / out: validate_control's result (1 = collision free)            */
  int* d_skip;            /* [B]       scratch: robots that follow a DWA twist this tick                         */
  int* d_source;          /* [B]       out, optional: who produced u -- 0 control(), 1 a followed DWA twist,
                                       2 DWA tracking optTraj(), 3 DWA re-run towards the followed twist        */
  double val_dt, val_horizon;
  unsigned long long grid_epoch; /* 0: the grid's content may have changed since the last tick.  Otherwise the caller vouches
                                       that (d_grid, grid_epoch) names ONE content (bump it with every map update, ~1 Hz
                                       against the loop's 10 Hz): the inflated collision map of the last tick on this stream
                                       is reused instead of rebuilt                                                */
} eea_tick_io;
eea_status eea_tick_batch(eea_engine* e, unsigned B, const eea_batch_io* io, const eea_tick_io* tick,
                          const eea_collision_cfg* ccfg, const eea_dwa_cfg* dcfg, void* stream);

/* The calls above keep small device caches between calls (the ring offsets per radii, one
 * inflated-map buffer per (device, stream)).  A long-running process that changes streams or map sizes
 * can drop them; synchronises the devices involved.  No reference counterpart. */
void eea_release_collision_caches(void);

#ifdef __cplusplus
}
#endif
#endif

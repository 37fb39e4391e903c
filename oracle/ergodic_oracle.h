/*
 * ergodic_oracle.h -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * A literal, plain-C restatement of the algorithm on the hot path of
 * bostoncleek/ergodic_exploration (`ErgodicControl<ModelT>::control` and the
 * functions it calls).  Same formulation and operation order as the reference:
 * non-separated cosine basis, sequential RK4 in both directions with the full
 * 3x3 Jacobian, column-ordered sums.  Every function cites the reference
 * file:line it follows (paths relative to the reference root).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * use this library, and only as the checker / reported CPU baseline.  The
 * product (ergodic_exploration_amd/) never links, loads or calls it.
 *
 * PINNING STATUS (see DESIGN.md "Oracle"):
 *   - pinned against the reference's own known-answer tests (test/test_cart.cpp,
 *     test_omni.cpp, test_integrator.cpp, test_grid.cpp; transcribed as data in
 *     tests/golden/reference_kats.json) and against the end-to-end `control()`
 *     outputs of the reference's sources recorded at survey time
 *     (SURVEY.md section 8(c) "sanity anchors"; tests/golden/survey_anchors.json).
 *   - every stage of control() (rollout, c_k, both gradients, co-state, update) and configTarget's phi_k of Gaussian
 *     targets against CLOSED FORMS evaluated with mpmath at 40 digits (tests/analytic_chain.py,
 *     tests/test_analytic_checks.py): <= 9.2e-15 -- formulas the reference's lines state, not its binary.
 *   - The reference itself is UNBUILDABLE in this image (needs Armadillo and ROS
 *     message headers, both absent; writing stand-ins is not allowed), so there
 *     is no oracle/_ref build.  Armadillo-internal summation orders (accu, gemv,
 *     inv) are restated from its published 10.x sources and are "parity
 *     unpinned" below the 1e-12 level; all floating-point parity claims are
 *     therefore stated with tolerances >= 1e-12 relative, never bit-exact.
 */
#ifndef ERGODIC_ORACLE_H
#define ERGODIC_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { EO_MODEL_OMNI = 0, EO_MODEL_SIMPLE_CART = 1, EO_MODEL_CART = 2, EO_MODEL_MECANUM = 3 };

/* status codes */
enum { EO_OK = 0, EO_ERR_INVALID_ARGUMENT = 1, EO_ERR_INVALID_TWIST = 2 };

/* ---- numerics.hpp ------------------------------------------------------ */
double eo_normalize_angle_PI(double rad);
int eo_almost_equal(double d1, double d2, double eps);
void eo_integrate_twist(const double x[3], const double u[3], double dt, double out[3]);
double eo_entropy(double p);

/* ---- models (cart.hpp / omni.hpp) -------------------------------------
 * mp = model parameters: CART {wheel_radius, wheel_base};
 * MECANUM {wheel_radius, wheel_base_x, wheel_base_y}; unused for OMNI/SIMPLE_CART.
 * u has 3 entries (OMNI, SIMPLE_CART), 2 (CART) or 4 (MECANUM).
 * Matrices are column-major (Armadillo layout): fdx 3x3, fdu 3 x n_u. */
int eo_model_nu(int model);
int eo_model_f(int model, const double* mp, const double x[3], const double* u, double xdot[3]);
int eo_model_fdx(int model, const double* mp, const double x[3], const double* u, double A[9]);
int eo_model_fdu(int model, const double* mp, const double x[3], double* B);
int eo_model_wheels2twist(int model, const double* mp, const double* u, double vb[3]);

/* ---- integrator.hpp (RungeKutta) --------------------------------------- */
unsigned eo_steps(double horizon, double dt);
int eo_rk4_step_fwd(int model, const double* mp, double dt, const double x[3], const double* u,
                    double out[3]);
/* ut: n_u x steps column-major; xt out: 3 x steps column-major */
int eo_rk4_solve_fwd(int model, const double* mp, double dt, double horizon, const double x0[3],
                     const double* ut, double* xt);
void eo_rk4_step_bwd(double dt, const double rho[3], const double gdx[3], const double dbar[3],
                     const double fdx[9], double out[3]);
int eo_rk4_solve_bwd(int model, const double* mp, double dt, double horizon, const double rhoT[3],
                     const double* xt, const double* ut, const double* edx, const double* bdx,
                     double* rhot);

/* ---- basis.cpp (Basis) -------------------------------------------------- */
/* k: 2 x K^2 column-major int64 (k(0,col)=k1 x-mode, k(1,col)=k2), lamdak: K^2 */
void eo_basis_init(unsigned num_basis, int64_t* k, double* lamdak);
void eo_fourier_basis(double lx, double ly, unsigned num_basis, const double x[2], double* fk);
void eo_grad_fourier_basis(double lx, double ly, unsigned num_basis, const double x[2],
                           double* dfk /* 2 x K^2 col-major */);
/* xt: rows x n column-major (rows >= 2, only rows 0..1 used) */
void eo_traj_coeff(double lx, double ly, unsigned num_basis, const double* xt, unsigned rows,
                   unsigned n, double* ck);
void eo_spatial_coeff(double lx, double ly, unsigned num_basis, const double* phi_vals,
                      const double* phi_grid /* 2 x P */, unsigned P, double* phik);

/* ---- target.cpp / target.hpp ------------------------------------------- */
/* mu, sigma: 2 per gaussian.  phi_grid 2 x P.  out P (normalised to sum 1) */
void eo_target_fill(unsigned n_gauss, const double* mu, const double* sigma,
                    const double trans[2], const double* phi_grid, unsigned P, double* phi_vals);
double eo_target_evaluate(unsigned n_gauss, const double* mu, const double* sigma,
                          const double pt[2], const double trans[2]);

/* ---- grid.hpp / grid.cpp ------------------------------------------------ */
unsigned eo_axis_length(double lower, double upper, double resolution);
double eo_axis_upper(double lower, double resolution, unsigned size);
typedef struct {
  unsigned xsize, ysize;
  double resolution, xmin, ymin, xmax, ymax;
  const int8_t* data; /* row-major i*xsize + j, borrowed */
} eo_grid;
int eo_grid_init(eo_grid* g, double xmin, double xmax, double ymin, double ymax, double resolution,
                 const int8_t* data, unsigned n_data);
int eo_grid_bounds_ij(const eo_grid* g, unsigned i, unsigned j);
int eo_grid_bounds_idx(const eo_grid* g, unsigned idx);
unsigned eo_grid2rowmajor(const eo_grid* g, unsigned i, unsigned j);
void eo_rowmajor2grid(const eo_grid* g, unsigned idx, unsigned ij[2]);
void eo_grid2world(const eo_grid* g, unsigned i, unsigned j, double xy[2]);
void eo_world2grid(const eo_grid* g, double x, double y, unsigned ij[2]);
/* returns EO_ERR_INVALID_ARGUMENT where the reference throws */
int eo_grid_get_cell(const eo_grid* g, unsigned idx, double* val);

/* ---- collision.cpp ------------------------------------------------------ */
typedef struct {
  double boundary_radius, search_radius, obstacle_threshold, occupied_threshold;
} eo_collision;
/* result: 1 collision, 0 free; <0 never.  sqrd_obs/dx/dy report the
 * CollisionConfig state after the search (sqrd_obs = -1: nothing seen). */
int eo_collision_check(const eo_collision* c, const eo_grid* g, const double pose[3],
                       int* sqrd_obs, int* dx, int* dy);
/* numerics.hpp validate_control: returns 1 if collision free */
int eo_validate_control(const eo_collision* c, const eo_grid* g, const double x0[3],
                        const double u[3], double dt, double horizon);

/* ---- dynamic_window.cpp (DynamicWindow) -------------------------------- */
typedef struct {
  double dt, horizon, acc_dt, acc_lim_x, acc_lim_y, acc_lim_th;
  double max_vel_x, min_vel_x, max_vel_y, min_vel_y, max_rot_vel, min_rot_vel;
  unsigned vx_samples, vy_samples, vth_samples;
} eo_dwa;
/* control(grid, x0, vb, vref): returns 1 if a collision-free twist was found; u_opt[3];
 * min_cost (optional) receives the best cost */
int eo_dwa_control_vref(const eo_dwa* d, const eo_collision* c, const eo_grid* g, const double x0[3],
                        const double vb[3], const double vref[3], double u_opt[3], double* min_cost);
/* control(grid, x0, vb, xt_ref, dt_ref): xt_ref 3 x n_ref column-major */
int eo_dwa_control_traj(const eo_dwa* d, const eo_collision* c, const eo_grid* g, const double x0[3],
                        const double vb[3], const double* xt_ref, unsigned n_ref, double dt_ref,
                        double u_opt[3], double* min_cost);

/* objective(x0, u, xt_ref, dt_ref) of dynamic_window.cpp:258-286 for one candidate twist (DBL_MAX on collision) */
double eo_dwa_objective_traj(const eo_dwa* d, const eo_collision* c, const eo_grid* g, const double x0[3],
                             const double u[3], const double* xt_ref, unsigned n_ref, double dt_ref);

/* ---- ergodic_control.hpp (ErgodicControl<ModelT>) ---------------------- */
typedef struct eo_control eo_control;

typedef struct {
  int model;            /* EO_MODEL_OMNI or EO_MODEL_SIMPLE_CART */
  double dt, horizon, resolution, expl_weight;
  unsigned num_basis;
  double Rinv[9];       /* column-major 3x3 */
  double umin[3], umax[3];
} eo_control_config;

/* optional per-stage outputs of one control() call; any pointer may be NULL */
typedef struct {
  double* traj;  /* 3 x T, map frame (rk4 output) */
  double* ck;    /* K^2 */
  double* edx;   /* 3 x T */
  double* bdx;   /* 3 x T */
  double* rhot;  /* 3 x T */
  double* ut;    /* 3 x T, after updateControl */
} eo_stage_out;

int eo_control_create(const eo_control_config* cfg, eo_control** out);
void eo_control_destroy(eo_control* ec);
unsigned eo_control_steps(const eo_control* ec);
void eo_control_set_target(eo_control* ec, unsigned n_gauss, const double* mu, const double* sigma);
/* set phi_k from an explicit target grid (Basis::spatialCoeff entry, basis.hpp:99):
 * phi_vals has nx*ny entries, x fastest; grid coordinates generated as configTarget does */
void eo_control_set_target_grid(eo_control* ec, unsigned nx, unsigned ny, const double* phi_vals,
                                double lx, double ly);
/* configTarget (ergodic_control.hpp:362-416). returns 1 if phi_k was rebuilt */
int eo_control_config_target(eo_control* ec, double xmin, double xmax, double ymin, double ymax);
const double* eo_control_phik(const eo_control* ec);
const double* eo_control_lamdak(const eo_control* ec);
double* eo_control_ut(eo_control* ec); /* 3 x T warm-start state, mutable */
/* control(): mem_cols = the columns ReplayBuffer::sampleMemory would prepend
 * (map frame, 3 x n_mem); returns status, u_out = ut.col(0) */
int eo_control_step(eo_control* ec, double xmin, double xmax, double ymin, double ymax,
                    const double x[3], const double* mem_cols, unsigned n_mem, double u_out[3],
                    const eo_stage_out* stages);
int eo_control_opt_traj(const eo_control* ec, double* traj /* 3 x T */);
/* Decentralised-consensus switch (multi-agent extension, reference README ref. [2]; no single-agent
 * counterpart in ergodic_control.hpp): ck_shared (K^2, copied) replaces the agent's own c_k in
 * fourier_diff = lamdak % (ck - phik) (ergodic_control.hpp:422); NULL restores the reference behaviour */
void eo_control_set_shared_ck(eo_control* ec, const double* ck_shared);

/* bounded CPU-baseline loop for bench.py: runs `calls` control() calls on each of
 * n_agents independent controllers, spread over `threads` pthreads; returns
 * wall seconds (monotonic). poses: 3 per agent. */
double eo_bench_control(const eo_control_config* cfg, unsigned n_gauss, const double* mu,
                        const double* sigma, double xmin, double xmax, double ymin, double ymax,
                        const double* poses, unsigned n_agents, unsigned calls, unsigned threads,
                        double* u_last /* 3 per agent, may be NULL */);
/* same loop (one warm-up call + `calls` calls per agent) that also returns every agent's warm-start
 * controls after the last call: ut_last 3 x T per agent (may be NULL).  Full-size parity checks. */
double eo_batch_control(const eo_control_config* cfg, unsigned n_gauss, const double* mu,
                        const double* sigma, double xmin, double xmax, double ymin, double ymax,
                        const double* poses, unsigned n_agents, unsigned calls, unsigned threads,
                        double* u_last, double* ut_last);

#ifdef __cplusplus
}
#endif
#endif

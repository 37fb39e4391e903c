/*
 * ergodic_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 * See ergodic_oracle.h for the role, the allowed users and the pinning status.
 *
 * Everything here follows the reference formulation literally; citations are
 * `file:line` relative to the reference root.  Build with -ffp-contract=off so
 * the arithmetic matches an x86-64 gcc build of the reference (no FMA fusion).
 */
#define _POSIX_C_SOURCE 200809L
#include "ergodic_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* numerics.hpp:59 */
static const double EO_PI = 3.14159265358979323846;

/* ======================================================================= */
/* numerics.hpp                                                             */
/* ======================================================================= */

/* numerics.hpp:68-71 */
int eo_almost_equal(double d1, double d2, double eps) { return fabs(d1 - d2) < eps ? 1 : 0; }

/* numerics.hpp:78-90: wrap to [-pi, pi) */
double eo_normalize_angle_PI(double rad)
{
  const double q = floor((rad + EO_PI) / (2.0 * EO_PI));
  rad = (rad + EO_PI) - q * 2.0 * EO_PI;
  if (rad < 0.0) {
    rad += 2.0 * EO_PI;
  }
  return (rad - EO_PI);
}

/* numerics.hpp:164-179 */
double eo_entropy(double p)
{
  if (eo_almost_equal(0.0, p, 1.0e-12) || eo_almost_equal(1.0, p, 1.0e-12)) {
    return 1e-3;
  } else if (p < 0.0) {
    return 0.7;
  }
  return -p * log(p) - (1.0 - p) * log(1.0 - p);
}

/* numerics.hpp:273-297 (integrate_twist) with transform2d(angle) numerics.hpp:236-242 */
void eo_integrate_twist(const double x[3], const double u[3], double dt, double out[3])
{
  double dqb[3];
  if (eo_almost_equal(u[2], 0.0, 1.0e-12)) {
    dqb[0] = u[0] * dt;
    dqb[1] = u[1] * dt;
    dqb[2] = 0.0;
  } else {
    const double vb0 = u[0] * dt, vb1 = u[1] * dt, vb2 = u[2] * dt;
    dqb[0] = (vb0 * sin(vb2) + vb1 * (cos(vb2) - 1.0)) / vb2;
    dqb[1] = (vb1 * sin(vb2) + vb0 * (1.0 - cos(vb2))) / vb2;
    dqb[2] = vb2;
  }
  /* x + transform2d(x(2)) * dqb : 3x3 times 3x1 (rows: [c -s 0; s c 0; 0 0 1]) */
  const double c = cos(x[2]), s = sin(x[2]);
  out[0] = x[0] + ((c * dqb[0] + (-s) * dqb[1]) + 0.0 * dqb[2]);
  out[1] = x[1] + ((s * dqb[0] + c * dqb[1]) + 0.0 * dqb[2]);
  out[2] = x[2] + ((0.0 * dqb[0] + 0.0 * dqb[1]) + 1.0 * dqb[2]);
}

/* ======================================================================= */
/* models                                                                   */
/* ======================================================================= */

int eo_model_nu(int model)
{
  switch (model) {
    case EO_MODEL_OMNI:
    case EO_MODEL_SIMPLE_CART:
      return 3;
    case EO_MODEL_CART:
      return 2;
    case EO_MODEL_MECANUM:
      return 4;
    default:
      return -1;
  }
}

int eo_model_f(int model, const double* mp, const double x[3], const double* u, double xdot[3])
{
  switch (model) {
    case EO_MODEL_OMNI: { /* models/omni.hpp:177-184 */
      xdot[0] = u[0] * cos(x[2]) - u[1] * sin(x[2]);
      xdot[1] = u[0] * sin(x[2]) + u[1] * cos(x[2]);
      xdot[2] = u[2];
      return EO_OK;
    }
    case EO_MODEL_SIMPLE_CART: { /* models/cart.hpp:165-173 */
      if (!eo_almost_equal(u[1], 0.0, 1.0e-12)) {
        return EO_ERR_INVALID_TWIST; /* reference throws std::invalid_argument */
      }
      xdot[0] = u[0] * cos(x[2]);
      xdot[1] = u[0] * sin(x[2]);
      xdot[2] = u[2];
      return EO_OK;
    }
    case EO_MODEL_CART: { /* models/cart.hpp:93-101 */
      const double r = mp[0], wb = mp[1];
      xdot[0] = (u[0] + u[1]) * cos(x[2]);
      xdot[1] = (u[0] + u[1]) * sin(x[2]);
      xdot[2] = (u[1] - u[0]) / wb;
      xdot[0] = (r / 2.0) * xdot[0];
      xdot[1] = (r / 2.0) * xdot[1];
      xdot[2] = (r / 2.0) * xdot[2];
      return EO_OK;
    }
    case EO_MODEL_MECANUM: { /* models/omni.hpp:99-111 */
      const double r = mp[0];
      const double s = (r / 4.0) * sin(x[2]);
      const double c = (r / 4.0) * cos(x[2]);
      const double l = r / (4.0 * (mp[1] + mp[2]));
      xdot[0] = u[0] * (s + c) + u[1] * (-s + c) + u[2] * (s + c) + u[3] * (-s + c);
      xdot[1] = u[0] * (s - c) + u[1] * (s + c) + u[2] * (s - c) + u[3] * (s + c);
      xdot[2] = -u[0] * l + u[1] * l + u[2] * l - u[3] * l;
      return EO_OK;
    }
    default:
      return EO_ERR_INVALID_ARGUMENT;
  }
}

int eo_model_fdx(int model, const double* mp, const double x[3], const double* u, double A[9])
{
  memset(A, 0, 9 * sizeof(double));
  switch (model) {
    case EO_MODEL_OMNI: /* models/omni.hpp:192-198 */
      A[0 + 3 * 2] = -u[0] * sin(x[2]) - u[1] * cos(x[2]);
      A[1 + 3 * 2] = u[0] * cos(x[2]) - u[1] * sin(x[2]);
      return EO_OK;
    case EO_MODEL_SIMPLE_CART: /* models/cart.hpp:181-187 */
      A[0 + 3 * 2] = -u[0] * sin(x[2]);
      A[1 + 3 * 2] = u[0] * cos(x[2]);
      return EO_OK;
    case EO_MODEL_CART: { /* models/cart.hpp:109-120 */
      const double r = mp[0];
      A[0 + 3 * 2] = -(r / 2.0) * (u[0] + u[1]) * sin(x[2]);
      A[1 + 3 * 2] = (r / 2.0) * (u[0] + u[1]) * cos(x[2]);
      return EO_OK;
    }
    case EO_MODEL_MECANUM: { /* models/omni.hpp:119-135 */
      const double s = (mp[0] / 4.0) * sin(x[2]);
      const double c = (mp[0] / 4.0) * cos(x[2]);
      A[0 + 3 * 2] = u[0] * (-s + c) + u[1] * (-s - c) + u[2] * (-s + c) + u[3] * (-s - c);
      A[1 + 3 * 2] = u[0] * (s + c) + u[1] * (-s + c) + u[2] * (s + c) + u[3] * (-s + c);
      return EO_OK;
    }
    default:
      return EO_ERR_INVALID_ARGUMENT;
  }
}

int eo_model_fdu(int model, const double* mp, const double x[3], double* B)
{
  switch (model) {
    case EO_MODEL_OMNI: /* models/omni.hpp:205-212 */
      B[0] = cos(x[2]);  B[3] = -sin(x[2]); B[6] = 0.0;
      B[1] = sin(x[2]);  B[4] = cos(x[2]);  B[7] = 0.0;
      B[2] = 0.0;        B[5] = 0.0;        B[8] = 1.0;
      return EO_OK;
    case EO_MODEL_SIMPLE_CART: /* models/cart.hpp:194-203 */
      memset(B, 0, 9 * sizeof(double));
      B[0] = cos(x[2]);
      B[1] = sin(x[2]);
      B[8] = 1.0;
      return EO_OK;
    case EO_MODEL_CART: { /* models/cart.hpp:127-141, 3x2 */
      const double r = mp[0], wb = mp[1];
      B[0] = cos(x[2]);  B[3] = cos(x[2]);
      B[1] = sin(x[2]);  B[4] = sin(x[2]);
      B[2] = -1.0 / wb;  B[5] = 1.0 / wb;
      for (int i = 0; i < 6; i++) B[i] = (r / 2.0) * B[i];
      return EO_OK;
    }
    case EO_MODEL_MECANUM: { /* models/omni.hpp:142-151, 3x4 */
      const double s = (mp[0] / 4.0) * sin(x[2]);
      const double c = (mp[0] / 4.0) * cos(x[2]);
      const double l = mp[0] / (4.0 * (mp[1] + mp[2]));
      B[0] = s + c;  B[3] = -s + c; B[6] = s + c;  B[9] = -s + c;
      B[1] = s - c;  B[4] = s + c;  B[7] = s - c;  B[10] = s + c;
      B[2] = -l;     B[5] = l;      B[8] = l;      B[11] = -l;
      return EO_OK;
    }
    default:
      return EO_ERR_INVALID_ARGUMENT;
  }
}

int eo_model_wheels2twist(int model, const double* mp, const double* u, double vb[3])
{
  if (model == EO_MODEL_CART) { /* models/cart.hpp:79-85 */
    vb[0] = mp[0] / 2.0 * (u[0] + u[1]);
    vb[1] = 0.0;
    vb[2] = mp[0] / (2.0 * mp[1]) * (u[1] - u[0]);
    return EO_OK;
  }
  if (model == EO_MODEL_MECANUM) { /* models/omni.hpp:81-91: (r/4) * Hp * u */
    const double l = 1.0 / (mp[1] + mp[2]);
    const double Hp[3][4] = { { 1.0, 1.0, 1.0, 1.0 }, { -1.0, 1.0, -1.0, 1.0 }, { -l, l, l, -l } };
    for (int i = 0; i < 3; i++) {
      double acc = 0.0;
      for (int j = 0; j < 4; j++) acc += ((mp[0] / 4.0) * Hp[i][j]) * u[j];
      vb[i] = acc;
    }
    return EO_OK;
  }
  return EO_ERR_INVALID_ARGUMENT;
}

/* ======================================================================= */
/* integrator.hpp : RungeKutta                                              */
/* ======================================================================= */

/* integrator.hpp:141,161 and ergodic_control.hpp:199: truncating cast */
unsigned eo_steps(double horizon, double dt) { return (unsigned)fabs(horizon / dt); }

/* integrator.hpp:176-184 */
int eo_rk4_step_fwd(int model, const double* mp, double dt, const double x[3], const double* u,
                    double out[3])
{
  double k1[3], k2[3], k3[3], k4[3], xs[3];
  int st;
  if ((st = eo_model_f(model, mp, x, u, k1)) != EO_OK) return st;
  for (int i = 0; i < 3; i++) xs[i] = x[i] + dt * (0.5 * k1[i]);
  if ((st = eo_model_f(model, mp, xs, u, k2)) != EO_OK) return st;
  for (int i = 0; i < 3; i++) xs[i] = x[i] + dt * (0.5 * k2[i]);
  if ((st = eo_model_f(model, mp, xs, u, k3)) != EO_OK) return st;
  for (int i = 0; i < 3; i++) xs[i] = x[i] + dt * k3[i];
  if ((st = eo_model_f(model, mp, xs, u, k4)) != EO_OK) return st;
  for (int i = 0; i < 3; i++) {
    out[i] = x[i] + (dt / 6.0) * (((k1[i] + 2.0 * k2[i]) + 2.0 * k3[i]) + k4[i]);
  }
  return EO_OK;
}

/* integrator.hpp:135-152: boundary condition x0 is NOT part of the output; the
 * heading is wrapped after every step */
int eo_rk4_solve_fwd(int model, const double* mp, double dt, double horizon, const double x0[3],
                     const double* ut, double* xt)
{
  const int nu = eo_model_nu(model);
  if (nu < 0) return EO_ERR_INVALID_ARGUMENT;
  const unsigned steps = eo_steps(horizon, dt);
  double x[3] = { x0[0], x0[1], x0[2] };
  for (unsigned i = 0; i < steps; i++) {
    double xn[3];
    const int st = eo_rk4_step_fwd(model, mp, dt, x, ut + (size_t)nu * i, xn);
    if (st != EO_OK) return st;
    x[0] = xn[0];
    x[1] = xn[1];
    x[2] = eo_normalize_angle_PI(xn[2]);
    xt[3 * i + 0] = x[0];
    xt[3 * i + 1] = x[1];
    xt[3 * i + 2] = x[2];
  }
  return EO_OK;
}

/* ergodic_control.hpp:65-69 rhodot: -gdx - dbar - fdx.t() * rho.
 * fdx.t()*rho restated as Armadillo's tiny-square transposed gemv
 * (y[i] = A(0,i) x0 + A(1,i) x1 + A(2,i) x2; order unpinned, see header). */
static void rhodot(const double rho[3], const double gdx[3], const double dbar[3],
                   const double A[9], double out[3])
{
  for (int i = 0; i < 3; i++) {
    const double atr = (A[0 + 3 * i] * rho[0] + A[1 + 3 * i] * rho[1]) + A[2 + 3 * i] * rho[2];
    out[i] = (-gdx[i] - dbar[i]) - atr;
  }
}

/* integrator.hpp:186-194 */
void eo_rk4_step_bwd(double dt, const double rho[3], const double gdx[3], const double dbar[3],
                     const double fdx[9], double out[3])
{
  double k1[3], k2[3], k3[3], k4[3], rs[3];
  rhodot(rho, gdx, dbar, fdx, k1);
  for (int i = 0; i < 3; i++) rs[i] = rho[i] - dt * (0.5 * k1[i]);
  rhodot(rs, gdx, dbar, fdx, k2);
  for (int i = 0; i < 3; i++) rs[i] = rho[i] - dt * (0.5 * k2[i]);
  rhodot(rs, gdx, dbar, fdx, k3);
  for (int i = 0; i < 3; i++) rs[i] = rho[i] - dt * k3[i];
  rhodot(rs, gdx, dbar, fdx, k4);
  for (int i = 0; i < 3; i++) {
    out[i] = rho[i] - dt / 6.0 * (((k1[i] + 2.0 * k2[i]) + 2.0 * k3[i]) + k4[i]);
  }
}

/* integrator.hpp:154-174: iterate i = steps-1 .. 0, A = fdx(xt_i, ut_i) held
 * fixed over the four stages, rhot(:,i) = rho after the step */
int eo_rk4_solve_bwd(int model, const double* mp, double dt, double horizon, const double rhoT[3],
                     const double* xt, const double* ut, const double* edx, const double* bdx,
                     double* rhot)
{
  const int nu = eo_model_nu(model);
  if (nu < 0) return EO_ERR_INVALID_ARGUMENT;
  const unsigned steps = eo_steps(horizon, dt);
  double rho[3] = { rhoT[0], rhoT[1], rhoT[2] };
  for (unsigned i = steps; i-- > 0;) {
    double A[9], rn[3];
    eo_model_fdx(model, mp, xt + 3 * i, ut + (size_t)nu * i, A);
    eo_rk4_step_bwd(dt, rho, edx + 3 * i, bdx + 3 * i, A, rn);
    for (int r = 0; r < 3; r++) {
      rho[r] = rn[r];
      rhot[3 * i + r] = rn[r];
    }
  }
  return EO_OK;
}

/* ======================================================================= */
/* basis.cpp : Basis                                                        */
/* ======================================================================= */

/* basis.cpp:48-77: mode table (x mode fastest) and lambda_k; no h_k */
void eo_basis_init(unsigned num_basis, int64_t* k, double* lamdak)
{
  unsigned col = 0;
  for (unsigned i = 0; i < num_basis; i++) {
    for (unsigned j = 0; j < num_basis; j++) {
      k[2 * col + 0] = (int64_t)j;
      k[2 * col + 1] = (int64_t)i;
      col++;
    }
  }
  const unsigned total = num_basis * num_basis;
  for (unsigned i = 0; i < total; i++) {
    /* sum(square(k_.col(i))) is an integer (imat); sqrt/pow in double */
    const int64_t ss = k[2 * i] * k[2 * i] + k[2 * i + 1] * k[2 * i + 1];
    lamdak[i] = 1.0 / pow((1.0 + sqrt((double)ss)), 1.5);
  }
}

/* basis.cpp:79-89 */
void eo_fourier_basis(double lx, double ly, unsigned num_basis, const double x[2], double* fk)
{
  unsigned col = 0;
  for (unsigned i = 0; i < num_basis; i++) {
    for (unsigned j = 0; j < num_basis; j++) {
      const double k0 = (double)j, k1 = (double)i;
      fk[col] = cos(k0 * (EO_PI / lx) * x[0]) * cos(k1 * (EO_PI / ly) * x[1]);
      col++;
    }
  }
}

/* basis.cpp:91-107 */
void eo_grad_fourier_basis(double lx, double ly, unsigned num_basis, const double x[2], double* dfk)
{
  unsigned col = 0;
  for (unsigned i = 0; i < num_basis; i++) {
    for (unsigned j = 0; j < num_basis; j++) {
      const double k1 = (double)j * (EO_PI / lx);
      const double k2 = (double)i * (EO_PI / ly);
      dfk[2 * col + 0] = -k1 * sin(k1 * x[0]) * cos(k2 * x[1]);
      dfk[2 * col + 1] = -k2 * cos(k1 * x[0]) * sin(k2 * x[1]);
      col++;
    }
  }
}

/* basis.cpp:109-120: (1/N) * sum(fk_mat, 1); Armadillo's sum(M,1) adds the
 * columns in order starting from zeros */
static void traj_coeff_ws(double lx, double ly, unsigned num_basis, const double* xt, unsigned rows,
                          unsigned n, double* ck, double* fk)
{
  const unsigned total = num_basis * num_basis;
  for (unsigned m = 0; m < total; m++) ck[m] = 0.0;
  for (unsigned i = 0; i < n; i++) {
    eo_fourier_basis(lx, ly, num_basis, xt + (size_t)rows * i, fk);
    for (unsigned m = 0; m < total; m++) ck[m] += fk[m];
  }
  const double inv = 1.0 / (double)n;
  for (unsigned m = 0; m < total; m++) ck[m] = inv * ck[m];
}

void eo_traj_coeff(double lx, double ly, unsigned num_basis, const double* xt, unsigned rows,
                   unsigned n, double* ck)
{
  const unsigned total = num_basis * num_basis;
  double* fk = (double*)malloc(sizeof(double) * total);
  for (unsigned m = 0; m < total; m++) ck[m] = 0.0;
  for (unsigned i = 0; i < n; i++) {
    eo_fourier_basis(lx, ly, num_basis, xt + (size_t)rows * i, fk);
    for (unsigned m = 0; m < total; m++) ck[m] += fk[m];
  }
  const double inv = 1.0 / (double)n;
  for (unsigned m = 0; m < total; m++) ck[m] = inv * ck[m];
  free(fk);
}

/* basis.cpp:122-133 */
void eo_spatial_coeff(double lx, double ly, unsigned num_basis, const double* phi_vals,
                      const double* phi_grid, unsigned P, double* phik)
{
  const unsigned total = num_basis * num_basis;
  double* fk = (double*)malloc(sizeof(double) * total);
  for (unsigned m = 0; m < total; m++) phik[m] = 0.0;
  for (unsigned i = 0; i < P; i++) {
    eo_fourier_basis(lx, ly, num_basis, phi_grid + 2 * (size_t)i, fk);
    for (unsigned m = 0; m < total; m++) phik[m] += fk[m] * phi_vals[i];
  }
  free(fk);
}

/* ======================================================================= */
/* target.hpp / target.cpp                                                  */
/* ======================================================================= */

/* target.hpp:68-70: cov = diagmat(sigma^2), cov_inv = inv(cov).
 * Armadillo's tiny 2x2 inverse (det formula) is restated; unpinned (header). */
static void gaussian_cov_inv(const double sigma[2], double ci[4])
{
  const double a = sigma[0] * sigma[0], b = 0.0, c = 0.0, d = sigma[1] * sigma[1];
  const double det = a * d - b * c;
  ci[0] = d / det;   /* (0,0) */
  ci[1] = -c / det;  /* (1,0) */
  ci[2] = -b / det;  /* (0,1) */
  ci[3] = a / det;   /* (1,1) */
}

/* target.hpp:91-102 */
static double gaussian_eval(const double mu[2], const double ci[4], const double pt[2],
                            const double trans[2])
{
  const double d0 = pt[0] - (mu[0] - trans[0]);
  const double d1 = pt[1] - (mu[1] - trans[1]);
  /* diff.t() * cov_inv : row vector */
  const double r0 = d0 * ci[0] + d1 * ci[1];
  const double r1 = d0 * ci[2] + d1 * ci[3];
  return exp(-0.5 * (r0 * d0 + r1 * d1));
}

/* target.cpp:68-76 */
double eo_target_evaluate(unsigned n_gauss, const double* mu, const double* sigma,
                          const double pt[2], const double trans[2])
{
  double val = 0.0;
  for (unsigned g = 0; g < n_gauss; g++) {
    double ci[4];
    gaussian_cov_inv(sigma + 2 * g, ci);
    val += gaussian_eval(mu + 2 * g, ci, pt, trans);
  }
  return val;
}

/* Armadillo accu()/sum(vec): two interleaved accumulators (restated; unpinned) */
static double arma_accumulate(const double* v, unsigned n)
{
  double acc1 = 0.0, acc2 = 0.0;
  unsigned i, j;
  for (i = 0, j = 1; j < n; i += 2, j += 2) {
    acc1 += v[i];
    acc2 += v[j];
  }
  if (i < n) acc1 += v[i];
  return acc1 + acc2;
}

/* target.cpp:78-89 */
void eo_target_fill(unsigned n_gauss, const double* mu, const double* sigma,
                    const double trans[2], const double* phi_grid, unsigned P, double* phi_vals)
{
  for (unsigned i = 0; i < P; i++) {
    phi_vals[i] = eo_target_evaluate(n_gauss, mu, sigma, phi_grid + 2 * (size_t)i, trans);
  }
  const double s = arma_accumulate(phi_vals, P);
  for (unsigned i = 0; i < P; i++) phi_vals[i] /= s;
}

/* ======================================================================= */
/* grid.hpp / grid.cpp                                                      */
/* ======================================================================= */

/* x86-64 gcc semantics of static_cast<unsigned>(double) as the reference
 * build exhibits them (cvttsd2si to 64 bit, low 32 bits kept): negative and
 * >2^32 values wrap mod 2^32 (measured). */
static unsigned cast_u32_x86(double v)
{
  if (!(v > -9.2233720368547758e18 && v < 9.2233720368547758e18)) {
    return 0u; /* cvttsd2si "integer indefinite" 0x8000000000000000 -> low 32 bits 0 */
  }
  return (unsigned)(uint64_t)(int64_t)v;
}

/* grid.hpp:61-64 */
unsigned eo_axis_length(double lower, double upper, double resolution)
{
  return cast_u32_x86(round((upper - lower) / resolution));
}

/* grid.hpp:73-76 */
double eo_axis_upper(double lower, double resolution, unsigned size)
{
  return (double)(resolution * size) + lower;
}

/* grid.cpp:46-62 */
int eo_grid_init(eo_grid* g, double xmin, double xmax, double ymin, double ymax, double resolution,
                 const int8_t* data, unsigned n_data)
{
  g->xsize = eo_axis_length(xmin, xmax, resolution);
  g->ysize = eo_axis_length(ymin, ymax, resolution);
  g->resolution = resolution;
  g->xmin = xmin;
  g->ymin = ymin;
  g->xmax = xmax;
  g->ymax = ymax;
  g->data = data;
  if (g->xsize * g->ysize != n_data) return EO_ERR_INVALID_ARGUMENT;
  return EO_OK;
}

/* grid.cpp:96-100 (unsigned arithmetic: size-1 wraps when size==0) */
int eo_grid_bounds_ij(const eo_grid* g, unsigned i, unsigned j)
{
  return ((i <= g->ysize - 1u) && (j <= g->xsize - 1u)) ? 1 : 0;
}

/* grid.cpp:102-107 */
int eo_grid_bounds_idx(const eo_grid* g, unsigned idx)
{
  return (idx <= (g->xsize * g->ysize - 1u)) ? 1 : 0;
}

/* grid.cpp:109-117 */
unsigned eo_grid2rowmajor(const eo_grid* g, unsigned i, unsigned j) { return i * g->xsize + j; }

/* grid.cpp:119-124 */
void eo_rowmajor2grid(const eo_grid* g, unsigned idx, unsigned ij[2])
{
  const unsigned i = (unsigned)(idx / g->xsize);
  ij[0] = i;
  ij[1] = idx - i * g->xsize;
}

/* grid.cpp:126-131 */
void eo_grid2world(const eo_grid* g, unsigned i, unsigned j, double xy[2])
{
  xy[0] = (double)(j * g->resolution) + g->resolution / 2.0 + g->xmin;
  xy[1] = (double)(i * g->resolution) + g->resolution / 2.0 + g->ymin;
}

/* grid.cpp:143-159 */
void eo_world2grid(const eo_grid* g, double x, double y, unsigned ij[2])
{
  unsigned j = cast_u32_x86(floor((x - g->xmin) / g->resolution));
  unsigned i = cast_u32_x86(floor((y - g->ymin) / g->resolution));
  if (j == g->xsize) j--;
  if (i == g->ysize) i--;
  ij[0] = i;
  ij[1] = j;
}

/* grid.cpp:177-184 */
int eo_grid_get_cell(const eo_grid* g, unsigned idx, double* val)
{
  if (!eo_grid_bounds_idx(g, idx)) return EO_ERR_INVALID_ARGUMENT;
  *val = (double)g->data[idx] / 100.0;
  return EO_OK;
}

/* ======================================================================= */
/* collision.cpp                                                            */
/* ======================================================================= */

typedef struct {
  int r_bnd, r_col, r_max, cx, cy, dx, dy, sqrd_obs;
} coll_cfg;

/* collision.cpp:216-243.  cj, ci are unsigned (implicit int->unsigned at the
 * call sites); (cfg.cx - cj) is unsigned arithmetic, cast to int at the end. */
static int check_cell(const eo_collision* c, coll_cfg* cfg, const eo_grid* g, unsigned cj,
                      unsigned ci)
{
  if (eo_grid_bounds_ij(g, ci, cj)) {
    double cell = 0.0;
    /* getCell(i,j) -> getCell(grid2RowMajor(i,j)); in bounds here so no throw */
    eo_grid_get_cell(g, eo_grid2rowmajor(g, ci, cj), &cell);
    if (!(cell < c->occupied_threshold)) {
      const unsigned ddx = (unsigned)cfg->cx - cj, ddy = (unsigned)cfg->cy - ci;
      const int sqrd_obs = (int)(ddx * ddx + ddy * ddy);
      if (sqrd_obs < cfg->sqrd_obs || cfg->sqrd_obs == -1) {
        cfg->sqrd_obs = sqrd_obs;
        cfg->dx = (int)(cj - (unsigned)cfg->cx);
        cfg->dy = (int)(ci - (unsigned)cfg->cy);
      }
      if (sqrd_obs <= cfg->r_col * cfg->r_col) return 1;
    }
  }
  return 0;
}

/* collision.cpp:166-214 */
static int bresenham_circle(const eo_collision* c, coll_cfg* cfg, const eo_grid* g, int r)
{
  int x = -r;
  int y = 0;
  int err = 2 - 2 * r;
  while (x < 0) {
    if (check_cell(c, cfg, g, (unsigned)(cfg->cx - x), (unsigned)(cfg->cy + y))) return 1;
    if (check_cell(c, cfg, g, (unsigned)(cfg->cx - y), (unsigned)(cfg->cy - x))) return 1;
    if (check_cell(c, cfg, g, (unsigned)(cfg->cx + x), (unsigned)(cfg->cy - y))) return 1;
    if (check_cell(c, cfg, g, (unsigned)(cfg->cx + y), (unsigned)(cfg->cy + x))) return 1;
    r = err;
    if (r <= y) {
      y++;
      err += 2 * y + 1;
    }
    if (r > x || err > y) {
      x++;
      err += 2 * x + 1;
    }
  }
  return 0;
}

/* collision.cpp:126-164 (collisionCheck + search) */
int eo_collision_check(const eo_collision* c, const eo_grid* g, const double pose[3],
                       int* sqrd_obs, int* dx, int* dy)
{
  unsigned psg[2];
  eo_world2grid(g, pose[0], pose[1], psg);
  coll_cfg cfg;
  /* CollisionConfig(int rb, int rc, int rm, int cx, int cy): double -> int and
   * unsigned -> int conversions at the call (collision.cpp:130-133) */
  cfg.r_bnd = (int)floor(c->boundary_radius / g->resolution);
  cfg.r_col = (int)floor((c->boundary_radius + c->obstacle_threshold) / g->resolution);
  cfg.r_max = (int)floor(c->search_radius / g->resolution);
  cfg.cx = (int)psg[1];
  cfg.cy = (int)psg[0];
  cfg.dx = 0;
  cfg.dy = 0;
  cfg.sqrd_obs = -1;
  int hit = 0;
  for (int r = cfg.r_bnd; r <= cfg.r_max; r++) {
    if (bresenham_circle(c, &cfg, g, r)) {
      hit = 1;
      break;
    }
  }
  if (sqrd_obs) *sqrd_obs = cfg.sqrd_obs;
  if (dx) *dx = cfg.dx;
  if (dy) *dy = cfg.dy;
  return hit;
}

/* numerics.hpp:312-330 */
int eo_validate_control(const eo_collision* c, const eo_grid* g, const double x0[3],
                        const double u[3], double dt, double horizon)
{
  double x[3] = { x0[0], x0[1], x0[2] };
  const unsigned steps = eo_steps(horizon, dt);
  for (unsigned i = 0; i < steps; i++) {
    double xn[3];
    eo_integrate_twist(x, u, dt, xn);
    x[0] = xn[0];
    x[1] = xn[1];
    x[2] = eo_normalize_angle_PI(xn[2]);
    if (eo_collision_check(c, g, x, NULL, NULL, NULL)) return 0;
  }
  return 1;
}

/* ======================================================================= */
/* dynamic_window.cpp : DynamicWindow                                       */
/* ======================================================================= */
#include <float.h>

/* dynamic_window.cpp:191-235 (window) */
static void dwa_window(const eo_dwa* d, const double vb[3], double lower[3], double delta[3])
{
  double upper[3];
  lower[0] = fmax(vb[0] - d->acc_lim_x * d->acc_dt, d->min_vel_x);
  upper[0] = fmin(vb[0] + d->acc_lim_x * d->acc_dt, d->max_vel_x);
  lower[1] = fmax(vb[1] - d->acc_lim_y * d->acc_dt, d->min_vel_y);
  upper[1] = fmin(vb[1] + d->acc_lim_y * d->acc_dt, d->max_vel_y);
  lower[2] = fmax(vb[2] - d->acc_lim_th * d->acc_dt, d->min_rot_vel);
  upper[2] = fmin(vb[2] + d->acc_lim_th * d->acc_dt, d->max_rot_vel);
  delta[0] = delta[1] = delta[2] = 0.0;
  const unsigned nx = d->vx_samples ? d->vx_samples : 1, ny = d->vy_samples ? d->vy_samples : 1,
                 nt = d->vth_samples ? d->vth_samples : 1; /* constructor :71-90 */
  if (nx > 1) delta[0] = (upper[0] - lower[0]) / (double)(nx - 1);
  if (ny > 1) delta[1] = (upper[1] - lower[1]) / (double)(ny - 1);
  if (nt > 1) delta[2] = (upper[2] - lower[2]) / (double)(nt - 1);
}

/* dynamic_window.cpp:237-256 and :258-286 (objective); xt_ref == NULL selects the first */
static double dwa_objective(const eo_dwa* d, const eo_collision* c, const eo_grid* g, const double x0[3],
                            const double vref[3], const double u[3], const double* xt_ref,
                            unsigned n_ref, double tf)
{
  const unsigned steps = eo_steps(d->horizon, d->dt);
  double pose[3] = { x0[0], x0[1], x0[2] };
  double t = 0.0, cost = 0.0;
  for (unsigned i = 0; i < steps; i++) {
    double pn[3];
    eo_integrate_twist(pose, u, d->dt, pn);
    pose[0] = pn[0];
    pose[1] = pn[1];
    pose[2] = eo_normalize_angle_PI(pn[2]);
    if (eo_collision_check(c, g, pose, NULL, NULL, NULL)) return DBL_MAX;
    if (xt_ref) {
      const unsigned j = (unsigned)round((double)(n_ref - 1) * t / tf);
      const double dx = xt_ref[3 * j + 0] - pose[0], dy = xt_ref[3 * j + 1] - pose[1];
      cost += sqrt(dx * dx + dy * dy); /* arma::norm of a 2-vector */
      cost += fabs(eo_normalize_angle_PI(eo_normalize_angle_PI(xt_ref[3 * j + 2]) - pose[2]));
      t += d->dt;
    }
  }
  if (xt_ref) return cost;
  /* dot(e, e) as Armadillo's direct_dot pairs it: (e0^2 + e2^2) + e1^2 (unpinned, header) */
  const double e0 = vref[0] - u[0], e1 = vref[1] - u[1], e2 = vref[2] - u[2];
  return (e0 * e0 + e2 * e2) + e1 * e1;
}

/* dynamic_window.cpp:92-189: sample grid by repeated +=, first strict minimum wins */
static int dwa_search(const eo_dwa* d, const eo_collision* c, const eo_grid* g, const double x0[3],
                      const double vb[3], const double vref[3], const double* xt_ref, unsigned n_ref,
                      double dt_ref, double u_opt[3], double* min_cost_out)
{
  double lower[3], delta[3];
  dwa_window(d, vb, lower, delta);
  const double tf = (double)n_ref * dt_ref;
  const unsigned nx = d->vx_samples ? d->vx_samples : 1, ny = d->vy_samples ? d->vy_samples : 1,
                 nt = d->vth_samples ? d->vth_samples : 1;
  double min_cost = DBL_MAX;
  u_opt[0] = u_opt[1] = u_opt[2] = 0.0;
  double vx = lower[0];
  for (unsigned i = 0; i < nx; i++) {
    double vy = lower[1];
    for (unsigned j = 0; j < ny; j++) {
      double w = lower[2];
      for (unsigned k = 0; k < nt; k++) {
        const double u[3] = { vx, vy, w };
        const double cost = dwa_objective(d, c, g, x0, vref, u, xt_ref, n_ref, tf);
        if (cost < min_cost) {
          min_cost = cost;
          u_opt[0] = u[0];
          u_opt[1] = u[1];
          u_opt[2] = u[2];
        }
        w += delta[2];
      }
      vy += delta[1];
    }
    vx += delta[0];
  }
  if (min_cost_out) *min_cost_out = min_cost;
  return eo_almost_equal(min_cost, DBL_MAX, 1.0e-12) ? 0 : 1;
}

/* cost of ONE candidate twist under the trajectory objective (:258-286): lets a test decide whether two
 * different choices are a floating-point tie */
double eo_dwa_objective_traj(const eo_dwa* d, const eo_collision* c, const eo_grid* g, const double x0[3],
                             const double u[3], const double* xt_ref, unsigned n_ref, double dt_ref)
{
  return dwa_objective(d, c, g, x0, NULL, u, xt_ref, n_ref, (double)n_ref * dt_ref);
}

int eo_dwa_control_vref(const eo_dwa* d, const eo_collision* c, const eo_grid* g, const double x0[3],
                        const double vb[3], const double vref[3], double u_opt[3], double* min_cost)
{
  return dwa_search(d, c, g, x0, vb, vref, NULL, 0, 0.0, u_opt, min_cost);
}

int eo_dwa_control_traj(const eo_dwa* d, const eo_collision* c, const eo_grid* g, const double x0[3],
                        const double vb[3], const double* xt_ref, unsigned n_ref, double dt_ref,
                        double u_opt[3], double* min_cost)
{
  return dwa_search(d, c, g, x0, vb, NULL, xt_ref, n_ref, dt_ref, u_opt, min_cost);
}

/* ======================================================================= */
/* ergodic_control.hpp : ErgodicControl<ModelT>                             */
/* ======================================================================= */

struct eo_control {
  eo_control_config cfg;
  unsigned steps, K, K2;
  double* ut;      /* 3 x steps */
  double* phik;    /* K2 */
  double* lamdak;  /* K2 */
  int64_t* k;      /* 2 x K2 */
  double lx, ly;   /* basis_.lx_, ly_ (start at 0, ergodic_control.hpp:208) */
  double map_pos[2];
  double pose[3];
  unsigned n_gauss;
  double* mu;
  double* sigma;
  /* scratch reused across control() calls (no arithmetic meaning): keeps the timed CPU
   * baseline free of allocator contention when one agent runs per thread */
  double *w_traj, *w_xt_total, *w_ck, *w_edx, *w_bdx, *w_rhot, *w_fk, *w_dfk, *w_diff;
  unsigned w_ncap;
  /* decentralised-consensus option (not in the reference's single-agent code; README ref. [2]):
   * when set, gradErgodicMetric uses this shared c_k in place of the agent's own */
  double* ck_shared;
};

/* ergodic_control.hpp:187-222 */
int eo_control_create(const eo_control_config* cfg, eo_control** out)
{
  if (cfg->model != EO_MODEL_OMNI && cfg->model != EO_MODEL_SIMPLE_CART) {
    return EO_ERR_INVALID_ARGUMENT;
  }
  const unsigned steps = eo_steps(cfg->horizon, cfg->dt);
  if (steps == 1) return EO_ERR_INVALID_ARGUMENT; /* :212-216 throws */
  eo_control* ec = (eo_control*)calloc(1, sizeof(eo_control));
  ec->cfg = *cfg;
  ec->steps = steps;
  ec->K = cfg->num_basis;
  ec->K2 = cfg->num_basis * cfg->num_basis;
  ec->ut = (double*)calloc((size_t)3 * (steps ? steps : 1), sizeof(double));
  ec->phik = (double*)calloc(ec->K2 ? ec->K2 : 1, sizeof(double));
  ec->lamdak = (double*)calloc(ec->K2 ? ec->K2 : 1, sizeof(double));
  ec->k = (int64_t*)calloc((size_t)2 * (ec->K2 ? ec->K2 : 1), sizeof(int64_t));
  eo_basis_init(ec->K, ec->k, ec->lamdak);
  ec->lx = 0.0;
  ec->ly = 0.0;
  {
    const size_t T3 = (size_t)3 * (steps ? steps : 1), K2 = ec->K2 ? ec->K2 : 1;
    ec->w_traj = (double*)malloc(sizeof(double) * T3);
    ec->w_edx = (double*)malloc(sizeof(double) * T3);
    ec->w_bdx = (double*)malloc(sizeof(double) * T3);
    ec->w_rhot = (double*)malloc(sizeof(double) * T3);
    ec->w_ck = (double*)malloc(sizeof(double) * K2);
    ec->w_fk = (double*)malloc(sizeof(double) * K2);
    ec->w_diff = (double*)malloc(sizeof(double) * K2);
    ec->w_dfk = (double*)malloc(sizeof(double) * 2 * K2);
    ec->w_ncap = steps;
    ec->w_xt_total = (double*)malloc(sizeof(double) * T3);
  }
  *out = ec;
  return EO_OK;
}

void eo_control_destroy(eo_control* ec)
{
  if (!ec) return;
  free(ec->ut);
  free(ec->phik);
  free(ec->lamdak);
  free(ec->k);
  free(ec->mu);
  free(ec->sigma);
  free(ec->w_traj);
  free(ec->w_xt_total);
  free(ec->w_ck);
  free(ec->w_edx);
  free(ec->w_bdx);
  free(ec->w_rhot);
  free(ec->w_fk);
  free(ec->w_dfk);
  free(ec->w_diff);
  free(ec->ck_shared);
  free(ec);
}

unsigned eo_control_steps(const eo_control* ec) { return ec->steps; }
const double* eo_control_phik(const eo_control* ec) { return ec->phik; }
const double* eo_control_lamdak(const eo_control* ec) { return ec->lamdak; }
double* eo_control_ut(eo_control* ec) { return ec->ut; }

void eo_control_set_shared_ck(eo_control* ec, const double* ck_shared)
{
  free(ec->ck_shared);
  ec->ck_shared = NULL;
  if (ck_shared) {
    ec->ck_shared = (double*)malloc(sizeof(double) * (ec->K2 ? ec->K2 : 1));
    memcpy(ec->ck_shared, ck_shared, sizeof(double) * ec->K2);
  }
}

/* ergodic_control.hpp:356-360 */
void eo_control_set_target(eo_control* ec, unsigned n_gauss, const double* mu, const double* sigma)
{
  free(ec->mu);
  free(ec->sigma);
  ec->n_gauss = n_gauss;
  ec->mu = (double*)malloc(sizeof(double) * 2 * (n_gauss ? n_gauss : 1));
  ec->sigma = (double*)malloc(sizeof(double) * 2 * (n_gauss ? n_gauss : 1));
  memcpy(ec->mu, mu, sizeof(double) * 2 * n_gauss);
  memcpy(ec->sigma, sigma, sizeof(double) * 2 * n_gauss);
}

/* grid of ergodic_control.hpp:387-408: coordinates by repeated += resolution,
 * x fastest, inclusive end points */
static double* build_phi_grid(unsigned nx, unsigned ny, double resolution)
{
  double* phi_grid = (double*)malloc(sizeof(double) * 2 * (size_t)nx * ny);
  unsigned col = 0;
  double y = 0.0;
  for (unsigned i = 0; i < ny; i++) {
    double x = 0.0;
    for (unsigned j = 0; j < nx; j++) {
      phi_grid[2 * (size_t)col + 0] = x;
      phi_grid[2 * (size_t)col + 1] = y;
      col++;
      x += resolution;
    }
    y += resolution;
  }
  return phi_grid;
}

void eo_control_set_target_grid(eo_control* ec, unsigned nx, unsigned ny, const double* phi_vals,
                                double lx, double ly)
{
  ec->lx = lx;
  ec->ly = ly;
  double* phi_grid = build_phi_grid(nx, ny, ec->cfg.resolution);
  eo_spatial_coeff(lx, ly, ec->K, phi_vals, phi_grid, nx * ny, ec->phik);
  free(phi_grid);
}

/* ergodic_control.hpp:362-416 */
int eo_control_config_target(eo_control* ec, double xmin, double xmax, double ymin, double ymax)
{
  ec->map_pos[0] = xmin;
  ec->map_pos[1] = ymin;
  const double mx = xmax - xmin;
  const double my = ymax - ymin;
  if (eo_almost_equal(mx, ec->lx, 1.0e-12) && eo_almost_equal(my, ec->ly, 1.0e-12)) {
    return 0;
  }
  ec->lx = mx;
  ec->ly = my;
  const unsigned nx = eo_axis_length(0.0, ec->lx, ec->cfg.resolution) + 1;
  const unsigned ny = eo_axis_length(0.0, ec->ly, ec->cfg.resolution) + 1;
  double* phi_grid = build_phi_grid(nx, ny, ec->cfg.resolution);
  double* phi_vals = (double*)malloc(sizeof(double) * (size_t)nx * ny);
  eo_target_fill(ec->n_gauss, ec->mu, ec->sigma, ec->map_pos, phi_grid, nx * ny, phi_vals);
  eo_spatial_coeff(ec->lx, ec->ly, ec->K, phi_vals, phi_grid, nx * ny, ec->phik);
  free(phi_vals);
  free(phi_grid);
  return 1;
}

/* ergodic_control.hpp:418-436.  gradFourierBasis(x) * fourier_diff is a
 * (2 x K^2)(K^2) product; restated as a plain in-order dot per row (Armadillo
 * dispatches this to BLAS gemv or its own emulation; order unpinned). */
static void grad_ergodic_metric(const eo_control* ec, const double* ck, const double* xt,
                                double* edx)
{
  const unsigned K2 = ec->K2;
  double* fourier_diff = ec->w_diff;
  double* dfk = ec->w_dfk;
  /* :422 fourier_diff = lamdak % (ck - phik); with a shared c_k the agents' mean replaces ck */
  if (ec->ck_shared) ck = ec->ck_shared;
  for (unsigned m = 0; m < K2; m++) fourier_diff[m] = ec->lamdak[m] * (ck[m] - ec->phik[m]);
  for (unsigned i = 0; i < ec->steps; i++) {
    eo_grad_fourier_basis(ec->lx, ec->ly, ec->K, xt + 3 * (size_t)i, dfk);
    double a0 = 0.0, a1 = 0.0;
    for (unsigned m = 0; m < K2; m++) {
      a0 += dfk[2 * m + 0] * fourier_diff[m];
      a1 += dfk[2 * m + 1] * fourier_diff[m];
    }
    edx[3 * i + 0] = a0;
    edx[3 * i + 1] = a1;
    edx[3 * i + 2] = 0.0;
  }
  for (unsigned i = 0; i < ec->steps; i++) {
    edx[3 * i + 0] *= ec->cfg.expl_weight;
    edx[3 * i + 1] *= ec->cfg.expl_weight;
  }
}

/* ergodic_control.hpp:453-474 */
static void grad_barrier(const eo_control* ec, const double* xt, double* bdx)
{
  const double weight = 25.0;
  const double eps = 0.05;
  for (unsigned i = 0; i < ec->steps; i++) {
    const double x = xt[3 * i + 0], y = xt[3 * i + 1];
    double b0 = 0.0, b1 = 0.0;
    b0 += 2.0 * (double)(x > ec->lx - eps) * (x - (ec->lx - eps));
    b1 += 2.0 * (double)(y > ec->ly - eps) * (y - (ec->ly - eps));
    b0 += 2.0 * (double)(x < eps) * (x - eps);
    b1 += 2.0 * (double)(y < eps) * (y - eps);
    bdx[3 * i + 0] = b0;
    bdx[3 * i + 1] = b1;
    bdx[3 * i + 2] = 0.0;
  }
  for (unsigned i = 0; i < ec->steps; i++) {
    bdx[3 * i + 0] *= weight;
    bdx[3 * i + 1] *= weight;
  }
}

static double clamp_std(double v, double lo, double hi)
{
  /* std::clamp: (v < lo) ? lo : (hi < v) ? hi : v */
  return (v < lo) ? lo : (hi < v) ? hi : v;
}

/* ergodic_control.hpp:438-451: -Rinv * fdu(x).t() * rho as Armadillo orders
 * it for these shapes: tmp = -(B^T rho), u = Rinv * tmp (unpinned, header) */
static void update_control(eo_control* ec, const double* xt, const double* rhot)
{
  const double* R = ec->cfg.Rinv;
  for (unsigned i = 0; i < ec->steps; i++) {
    double B[9], t[3], u[3];
    eo_model_fdu(ec->cfg.model, NULL, xt + 3 * (size_t)i, B);
    const double* rho = rhot + 3 * (size_t)i;
    for (int c = 0; c < 3; c++) {
      t[c] = -((B[0 + 3 * c] * rho[0] + B[1 + 3 * c] * rho[1]) + B[2 + 3 * c] * rho[2]);
    }
    for (int r = 0; r < 3; r++) {
      u[r] = (R[r + 3 * 0] * t[0] + R[r + 3 * 1] * t[1]) + R[r + 3 * 2] * t[2];
    }
    for (int r = 0; r < 3; r++) {
      ec->ut[3 * i + r] = clamp_std(u[r], ec->cfg.umin[r], ec->cfg.umax[r]);
    }
  }
}

/* ergodic_control.hpp:224-311 */
int eo_control_step(eo_control* ec, double xmin, double xmax, double ymin, double ymax,
                    const double x[3], const double* mem_cols, unsigned n_mem, double u_out[3],
                    const eo_stage_out* stages)
{
  const unsigned T = ec->steps;
  if (T == 0) return EO_ERR_INVALID_ARGUMENT; /* reference has UB here (n_cols - 2 wraps) */
  ec->pose[0] = x[0];
  ec->pose[1] = x[1];
  ec->pose[2] = x[2];

  /* :230 */
  eo_control_config_target(ec, xmin, xmax, ymin, ymax);

  /* :233-234 shift left by one column, last column zero */
  if (T >= 2) memmove(ec->ut, ec->ut + 3, sizeof(double) * 3 * (T - 1));
  ec->ut[3 * (T - 1) + 0] = 0.0;
  ec->ut[3 * (T - 1) + 1] = 0.0;
  ec->ut[3 * (T - 1) + 2] = 0.0;

  /* :237 forward simulation */
  double* traj = ec->w_traj;
  int st = eo_rk4_solve_fwd(ec->cfg.model, NULL, ec->cfg.dt, ec->cfg.horizon, ec->pose, ec->ut, traj);
  if (st != EO_OK) return st;

  /* :240 sampleMemory (buffer.cpp:64-111): sampled columns first, rollout last */
  const unsigned N = T + n_mem;
  if (N > ec->w_ncap) {
    free(ec->w_xt_total);
    ec->w_xt_total = (double*)malloc(sizeof(double) * 3 * (size_t)N);
    ec->w_ncap = N;
  }
  double* xt_total = ec->w_xt_total;
  if (n_mem) memcpy(xt_total, mem_cols, sizeof(double) * 3 * n_mem);
  memcpy(xt_total + 3 * (size_t)n_mem, traj, sizeof(double) * 3 * T);

  /* :243-244 map frame -> fourier frame */
  for (unsigned i = 0; i < N; i++) {
    xt_total[3 * i + 0] -= ec->map_pos[0];
    xt_total[3 * i + 1] -= ec->map_pos[1];
  }

  /* :264 last T columns */
  const double* xt = xt_total + 3 * (size_t)(N - T);

  double* ck = ec->w_ck;
  double* edx = ec->w_edx;
  double* bdx = ec->w_bdx;
  double* rhot = ec->w_rhot;

  /* :267 */
  traj_coeff_ws(ec->lx, ec->ly, ec->K, xt_total, 3, N, ck, ec->w_fk);
  /* :270 */
  grad_ergodic_metric(ec, ck, xt, edx);
  /* :273 */
  grad_barrier(ec, xt, bdx);
  /* :277 */
  const double rhoT[3] = { 0.0, 0.0, 0.0 };
  eo_rk4_solve_bwd(ec->cfg.model, NULL, ec->cfg.dt, ec->cfg.horizon, rhoT, xt, ec->ut, edx, bdx,
                   rhot);
  /* :305 */
  update_control(ec, xt, rhot);

  if (stages) {
    if (stages->traj) memcpy(stages->traj, traj, sizeof(double) * 3 * T);
    if (stages->ck) memcpy(stages->ck, ck, sizeof(double) * ec->K2);
    if (stages->edx) memcpy(stages->edx, edx, sizeof(double) * 3 * T);
    if (stages->bdx) memcpy(stages->bdx, bdx, sizeof(double) * 3 * T);
    if (stages->rhot) memcpy(stages->rhot, rhot, sizeof(double) * 3 * T);
    if (stages->ut) memcpy(stages->ut, ec->ut, sizeof(double) * 3 * T);
  }

  /* :310 */
  u_out[0] = ec->ut[0];
  u_out[1] = ec->ut[1];
  u_out[2] = ec->ut[2];

  return EO_OK;
}

/* ergodic_control.hpp:313-317 */
int eo_control_opt_traj(const eo_control* ec, double* traj)
{
  return eo_rk4_solve_fwd(ec->cfg.model, NULL, ec->cfg.dt, ec->cfg.horizon, ec->pose, ec->ut, traj);
}

/* ======================================================================= */
/* bounded CPU-baseline loop (bench.py cpu_baseline leg only)               */
/* ======================================================================= */

typedef struct {
  eo_control** ecs; /* one controller per agent, created + warmed outside the timed region */
  const double* poses;
  double xmin, xmax, ymin, ymax;
  unsigned first, last, calls;
  double* u_last;
  double* ut_last; /* 3 x T per agent, may be NULL */
} bench_job;

static void* bench_worker(void* arg)
{
  bench_job* j = (bench_job*)arg;
  for (unsigned a = j->first; a < j->last; a++) {
    eo_control* ec = j->ecs[a];
    if (!ec) continue;
    double u[3] = { 0, 0, 0 };
    for (unsigned c = 0; c < j->calls; c++) {
      eo_control_step(ec, j->xmin, j->xmax, j->ymin, j->ymax, j->poses + 3 * (size_t)a, NULL, 0, u,
                      NULL);
    }
    if (j->u_last) memcpy(j->u_last + 3 * (size_t)a, u, sizeof(u));
    if (j->ut_last) memcpy(j->ut_last + 3 * (size_t)ec->steps * a, ec->ut, sizeof(double) * 3 * ec->steps);
  }
  return NULL;
}

static void bench_run(bench_job* jobs, pthread_t* tids, unsigned threads)
{
  for (unsigned t = 0; t < threads; t++) pthread_create(&tids[t], NULL, bench_worker, &jobs[t]);
  for (unsigned t = 0; t < threads; t++) pthread_join(tids[t], NULL);
}

/* Times `calls` control() calls per agent (phi_k rebuild and one warm-up call
 * per agent happen before the clock starts, matching SURVEY 8(d): the metric
 * excludes the phi_k rebuild). */
double eo_bench_control(const eo_control_config* cfg, unsigned n_gauss, const double* mu,
                        const double* sigma, double xmin, double xmax, double ymin, double ymax,
                        const double* poses, unsigned n_agents, unsigned calls, unsigned threads,
                        double* u_last)
{
  return eo_batch_control(cfg, n_gauss, mu, sigma, xmin, xmax, ymin, ymax, poses, n_agents, calls,
                          threads, u_last, NULL);
}

double eo_batch_control(const eo_control_config* cfg, unsigned n_gauss, const double* mu,
                        const double* sigma, double xmin, double xmax, double ymin, double ymax,
                        const double* poses, unsigned n_agents, unsigned calls, unsigned threads,
                        double* u_last, double* ut_last)
{
  if (threads == 0) threads = 1;
  if (threads > n_agents) threads = n_agents ? n_agents : 1;
  eo_control** ecs = (eo_control**)calloc(n_agents ? n_agents : 1, sizeof(eo_control*));
  for (unsigned a = 0; a < n_agents; a++) {
    if (eo_control_create(cfg, &ecs[a]) != EO_OK) {
      ecs[a] = NULL;
      continue;
    }
    eo_control_set_target(ecs[a], n_gauss, mu, sigma);
    if (a == 0) {
      eo_control_config_target(ecs[0], xmin, xmax, ymin, ymax);
    } else if (ecs[0]) { /* identical target/domain: share the phi_k result, skip the rebuild */
      memcpy(ecs[a]->phik, ecs[0]->phik, sizeof(double) * ecs[0]->K2);
      ecs[a]->lx = ecs[0]->lx;
      ecs[a]->ly = ecs[0]->ly;
    }
  }
  bench_job* jobs = (bench_job*)calloc(threads, sizeof(bench_job));
  pthread_t* tids = (pthread_t*)calloc(threads, sizeof(pthread_t));
  for (unsigned t = 0; t < threads; t++) {
    jobs[t].ecs = ecs;
    jobs[t].poses = poses;
    jobs[t].xmin = xmin;
    jobs[t].xmax = xmax;
    jobs[t].ymin = ymin;
    jobs[t].ymax = ymax;
    jobs[t].first = (unsigned)(((uint64_t)n_agents * t) / threads);
    jobs[t].last = (unsigned)(((uint64_t)n_agents * (t + 1)) / threads);
    jobs[t].calls = 1; /* warm-up */
    jobs[t].u_last = NULL;
    jobs[t].ut_last = NULL;
  }
  bench_run(jobs, tids, threads);
  for (unsigned t = 0; t < threads; t++) {
    jobs[t].calls = calls;
    jobs[t].u_last = u_last;
    jobs[t].ut_last = ut_last;
  }
  struct timespec t0, t1;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  bench_run(jobs, tids, threads);
  clock_gettime(CLOCK_MONOTONIC, &t1);
  for (unsigned a = 0; a < n_agents; a++) eo_control_destroy(ecs[a]);
  free(ecs);
  free(tids);
  free(jobs);
  return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}

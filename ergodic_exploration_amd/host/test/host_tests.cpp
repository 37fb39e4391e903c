// Test driver of the host C++ mirror.  `host_tests cpu` runs the checks that need no device: the
// reference's own 22 known-answer tests (test/test_cart.cpp, test_omni.cpp, test_integrator.cpp,
// test_grid.cpp of bostoncleek/ergodic_exploration), restated against the mirror's classes with
// the same inputs, expected values and tolerances.  `host_tests gpu` adds the device-backed
// classes: RungeKutta::solve / Basis / Target against the host helpers, and ErgodicControl
// against the end-to-end control() outputs of the reference recorded in SURVEY.md 8(c).
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>

#include <ergodic_exploration/dynamic_window.hpp>
#include <ergodic_exploration/agent_batch.hpp>
#include <ergodic_exploration/ergodic_control.hpp>
#include <ergodic_exploration/exploration.hpp>

using namespace ergodic_exploration;

static int g_fail = 0, g_checks = 0;
#define CHECK(cond)                                                                  \
  do {                                                                               \
    ++g_checks;                                                                      \
    if (!(cond)) {                                                                   \
      ++g_fail;                                                                      \
      std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond);                    \
    }                                                                                \
  } while (0)
#define CHECK_NEAR(a, b, tol)                                                        \
  do {                                                                               \
    ++g_checks;                                                                      \
    const double va__ = (a), vb__ = (b);                                             \
    if (!(std::fabs(va__ - vb__) <= (tol))) {                                        \
      ++g_fail;                                                                      \
      std::printf("FAIL %s:%d: |%s - %s| = %.3g > %g\n", __FILE__, __LINE__, #a, #b, \
                  std::fabs(va__ - vb__), static_cast<double>(tol));                 \
    }                                                                                \
  } while (0)

static bool ulp4(double a, double b)
{
  if (a == b) return true;
  const double m = std::fmax(std::fabs(a), std::fabs(b));
  return std::fabs(a - b) <= 4.0 * (std::nextafter(m, INFINITY) - m);
}

// ---- test/test_cart.cpp ---------------------------------------------------------------------
static void test_cart()
{
  const models::Cart cart(0.033, 0.08);
  const vec x = { 1.0, 2.0, 0.707 }, u = { 0.1, 0.2 };
  const vec xdot = cart(x, u);
  CHECK_NEAR(xdot(0), 0.003763, 1e-6);
  CHECK_NEAR(xdot(1), 0.003215, 1e-6);
  CHECK_NEAR(xdot(2), 0.020625, 1e-6);
  const mat A = cart.fdx(x, u);
  CHECK_NEAR(A(0, 2), -0.003215, 1e-6);
  CHECK_NEAR(A(1, 2), 0.003763, 1e-6);
  const mat B = cart.fdu(x);
  CHECK_NEAR(B(0, 0), 0.012545, 1e-6);
  CHECK_NEAR(B(0, 1), 0.012545, 1e-6);
  CHECK_NEAR(B(1, 0), 0.010717, 1e-6);
  CHECK_NEAR(B(1, 1), 0.010717, 1e-6);
  CHECK_NEAR(B(2, 0), -0.20625, 1e-6);
  CHECK_NEAR(B(2, 1), 0.20625, 1e-6);
  vec vb = cart.wheels2Twist({ 1.0, 1.0 });
  CHECK_NEAR(vb(0), 0.033, 1e-6);
  CHECK_NEAR(vb(1), 0.0, 1e-6);
  CHECK_NEAR(vb(2), 0.0, 1e-6);
  vb = cart.wheels2Twist({ -1.0, 1.0 });
  CHECK_NEAR(vb(0), 0.0, 1e-6);
  CHECK_NEAR(vb(2), 0.4125, 1e-6);
  vb = cart.wheels2Twist({ 1.0, -1.0 });
  CHECK_NEAR(vb(2), -0.4125, 1e-6);

  const models::SimpleCart sc;
  const vec us = { 0.5, 0.0, 0.01 };
  const vec sdot = sc(x, us);
  CHECK_NEAR(sdot(0), 0.380156, 1e-6);
  CHECK_NEAR(sdot(1), 0.324777, 1e-6);
  CHECK_NEAR(sdot(2), 0.01, 1e-6);
  const mat As = sc.fdx(x, us);
  CHECK_NEAR(As(0, 2), -0.324777, 1e-6);
  CHECK_NEAR(As(1, 2), 0.380156, 1e-6);
  const mat Bs = sc.fdu(x);
  CHECK_NEAR(Bs(0, 0), 0.760313, 1e-6);
  CHECK_NEAR(Bs(1, 0), 0.649555, 1e-6);
  bool threw = false;
  try {
    sc(x, { 0.5, 0.1, 0.0 });
  } catch (const std::invalid_argument&) {
    threw = true;
  }
  CHECK(threw);
}

// ---- test/test_omni.cpp ---------------------------------------------------------------------
static void test_mecanum()
{
  const models::Mecanum m(0.1, 0.5, 0.5);
  const vec x = { 1.0, 2.0, 0.707 }, u = { 0.5, 0.4, 0.6, 0.3 };
  const vec xdot = m(x, u);
  CHECK_NEAR(xdot(0), 0.040709, 1e-6);
  CHECK_NEAR(xdot(1), 0.021626, 1e-6);
  CHECK_NEAR(xdot(2), 0.005, 1e-6);
  const mat A = m.fdx(x, u);
  CHECK_NEAR(A(0, 2), -0.021626, 1e-6);
  CHECK_NEAR(A(1, 2), 0.040709, 1e-6);
  const mat B = m.fdu(x);
  const double e[3][4] = { { 0.035246, 0.002768, 0.035246, 0.002768 },
                           { -0.002768, 0.035246, -0.002768, 0.035246 },
                           { -0.025, 0.025, 0.025, -0.025 } };
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 4; ++j) CHECK_NEAR(B(i, j), e[i][j], 1e-6);
}

// ---- test/test_integrator.cpp ---------------------------------------------------------------
static void test_integrator()
{
  const models::Cart cart(0.1, 2.0);
  const double horizon = 0.4, dt = 0.1;
  mat ut(2, 4, 1.0);
  const RungeKutta rk4(dt);
  const mat xt = rk4.solve(cart, vec{ 0.0, 0.0, 0.0 }, ut, horizon);  // generic host template
  const double ex[4] = { 0.01, 0.02, 0.03, 0.04 };
  for (int i = 0; i < 4; ++i) {
    CHECK(ulp4(xt(0, i), ex[i]));
    CHECK(ulp4(xt(1, i), 0.0));
    CHECK(ulp4(xt(2, i), 0.0));
  }
  const vec vb = { 1.0, 0.5, 0.5 };
  const vec pose = integrate_twist(vec{ 0.0, 0.0, 0.0 }, vb, 0.1);
  CHECK_NEAR(pose(0), 0.0987, 1e-4);
  CHECK_NEAR(pose(1), 0.0525, 1e-4);
  CHECK_NEAR(pose(2), 0.0500, 1e-4);
  const double tx[5] = { 1.0987, 1.1947, 1.2876, 1.3774, 1.4637 }, ty[5] = { 0.0525, 0.1098, 0.1719, 0.2385, 0.3096 };
  vec x = { 1.0, 0.0, 0.0 };
  for (int i = 0; i < 5; ++i) {
    x = integrate_twist(x, vb, 0.1);
    CHECK_NEAR(x(0), tx[i], 1e-4);
    CHECK_NEAR(x(1), ty[i], 1e-4);
    CHECK_NEAR(x(2), 0.05 * (i + 1), 1e-4);
  }
  CHECK_NEAR(normalize_angle_PI(PI), -PI, 1e-15);
  CHECK_NEAR(normalize_angle_PI(7.0), 0.7168146928204138, 1e-15);
}

// ---- test/test_grid.cpp ---------------------------------------------------------------------
static void test_grid()
{
  {
    const unsigned xs = axis_length(0.0, 2.0, 1.0), ys = axis_length(0.0, 3.0, 1.0);
    const GridMap g(0.0, 2.0, 0.0, 3.0, 1.0, GridData(xs * ys, 0));
    CHECK(g.grid2RowMajor(2, 1) == 5);
    CHECK(g.rowMajor2Grid(5).at(0) == 2 && g.rowMajor2Grid(5).at(1) == 1);
    CHECK(g.gridBounds(5u) && !g.gridBounds(6u));
    CHECK(g.gridBounds(2u, 0u) && !g.gridBounds(0u, 2u));
  }
  {
    const unsigned xs = axis_length(-0.5, 0.5, 0.5), ys = axis_length(0.0, 1.5, 0.5);
    GridData d(xs * ys, 0);
    d.at(3) = 100;
    d.at(4) = 90;
    const GridMap g(-0.5, 0.5, 0.0, 1.5, 0.5, d);
    CHECK(ulp4(g.grid2World(0, 1).at(0), 0.25) && ulp4(g.grid2World(0, 1).at(1), 0.25));
    CHECK(ulp4(g.grid2World(4).at(0), -0.25) && ulp4(g.grid2World(4).at(1), 1.25));
    CHECK(g.world2Grid(-0.25, 1.25).at(0) == 2 && g.world2Grid(-0.25, 1.25).at(1) == 0);
    CHECK(g.world2RowMajor(0.25, 0.75) == 3);
    CHECK(g.getCell(4u) == 0.9);
    CHECK(g.getCell(-0.25, 1.25) == 0.9);
    CHECK(g.getCell(1u, 1u) == 1.0);
    bool threw = false;
    try {
      g.getCell(6u);
    } catch (const std::invalid_argument&) {
      threw = true;
    }
    CHECK(threw);
  }
  {  // negative coordinates wrap like the reference's x86-64 build
    const GridMap g(-1.0, 11.0, -1.0, 5.0, 0.1, GridData(120 * 60, 0));
    CHECK(g.world2Grid(-1.05, 0.0).at(1) == 4294967295u);
    CHECK(g.world2Grid(1e9, 0.0).at(1) == 1410065418u);
  }
  bool threw = false;
  try {
    GridMap(0.0, 2.0, 0.0, 3.0, 1.0, GridData(5, 0));
  } catch (const std::invalid_argument&) {
    threw = true;
  }
  CHECK(threw);
}

// ---- device-backed classes --------------------------------------------------------------------
template <class ModelT>
static mat host_rollout(const ModelT& model, double dt, const vec& x0, const mat& ut)
{
  const RungeKutta rk(dt);
  mat xt(3, ut.n_cols());
  vec x = x0;
  for (std::size_t i = 0; i < ut.n_cols(); ++i) {
    x = rk.step(model, x, ut.col(i));
    x(2) = normalize_angle_PI(x(2));
    xt.set_col(i, x);
  }
  return xt;
}

static void test_device_ops()
{
  // RungeKutta::solve on the device vs the generic host step loop
  const double dt = 0.1, horizon = 5.0;
  mat ut(3, 50);
  for (int i = 0; i < 50; ++i) {
    ut(0, i) = 0.6 * std::sin(0.3 * i);
    ut(1, i) = 0.4 * std::cos(0.2 * i);
    ut(2, i) = 1.5 * std::sin(0.11 * i + 1.0);
  }
  const vec x0 = { 1.0, 2.0, 3.0 };
  const RungeKutta rk(dt);
  const models::Omni omni;
  const mat xd = rk.solve(omni, x0, ut, horizon), xh = host_rollout(omni, dt, x0, ut);
  for (int i = 0; i < 50; ++i) {
    CHECK_NEAR(xd(0, i), xh(0, i), 1e-12);
    CHECK_NEAR(xd(1, i), xh(1, i), 1e-12);
    CHECK_NEAR(std::remainder(xd(2, i) - xh(2, i), 2.0 * PI), 0.0, 1e-12);
  }
  const models::SimpleCart sc;
  mat uc = ut;
  for (int i = 0; i < 50; ++i) uc(1, i) = 0.0;
  const mat cd = rk.solve(sc, x0, uc, horizon), ch = host_rollout(sc, dt, x0, uc);
  for (int i = 0; i < 50; ++i) CHECK_NEAR(cd(0, i), ch(0, i), 1e-12);
  bool threw = false;
  try {
    rk.solve(sc, x0, ut, horizon);  // lateral velocity: SimpleCart throws
  } catch (const std::invalid_argument&) {
    threw = true;
  }
  CHECK(threw);

  // Basis::trajCoeff / spatialCoeff on the device vs the single-point host helper
  const Basis basis(12.0, 6.0, 10);
  mat pts(3, 37);
  vec w(37);
  for (int i = 0; i < 37; ++i) {
    pts(0, i) = 0.3 * i;
    pts(1, i) = 0.15 * i + 0.1;
    w(i) = 0.01 * (i + 1);
  }
  const vec ck = basis.trajCoeff(pts);
  mat grid(2, 37);
  for (int i = 0; i < 37; ++i) {
    grid(0, i) = pts(0, i);
    grid(1, i) = pts(1, i);
  }
  const vec pk = basis.spatialCoeff(w, grid);
  vec ck_ref(100, 0.0), pk_ref(100, 0.0);
  for (int i = 0; i < 37; ++i) {
    const vec fk = basis.fourierBasis(vec{ pts(0, i), pts(1, i) });
    for (int m = 0; m < 100; ++m) {
      ck_ref(m) += fk(m) / 37.0;
      pk_ref(m) += fk(m) * w(i);
    }
  }
  for (int m = 0; m < 100; ++m) {
    CHECK_NEAR(ck(m), ck_ref(m), 1e-13);
    CHECK_NEAR(pk(m), pk_ref(m), 1e-13);
  }

  // Target::fill on the device vs evaluate
  const Target target({ Gaussian({ 2.5, 2.5 }, { 1.5, 1.5 }), Gaussian({ 8.5, 2.5 }, { 1.5, 1.5 }) });
  const vec trans = { -1.0, -1.0 };
  const vec pv = target.fill(trans, grid);
  double sum = 0.0;
  for (int i = 0; i < 37; ++i) sum += target.evaluate(vec{ grid(0, i), grid(1, i) }, trans);
  double total = 0.0;
  for (int i = 0; i < 37; ++i) {
    CHECK_NEAR(pv(i), target.evaluate(vec{ grid(0, i), grid(1, i) }, trans) / sum, 1e-15);
    total += pv(i);
  }
  CHECK_NEAR(total, 1.0, 1e-13);

  // Collision on a small map with one obstacle block
  GridData d(50 * 40, 0);
  for (int i = 10; i < 14; ++i)
    for (int j = 20; j < 26; ++j) d[i * 50 + j] = 100;
  const GridMap g = GridMap::fromOccupancyGrid(50, 40, 0.1, -1.0, -2.0, d);
  const Collision col(0.7, 1.0, 0.2, 0.8);
  CHECK(col.collisionCheck(g, vec{ 0.5, -0.8, 0.0 }));    // 0.5 m left of the block: inside r_col
  CHECK(!col.collisionCheck(g, vec{ 3.5, 1.5, 0.0 }));    // far away
  CHECK(validate_control(col, g, vec{ 3.5, 1.5, 0.0 }, vec{ 0.2, 0.0, 0.1 }, 0.1, 0.5));
  CHECK(!validate_control(col, g, vec{ -0.2, -0.8, 0.0 }, vec{ 1.0, 0.0, 0.0 }, 0.1, 0.5));

  // DynamicWindow: in free space the twist closest to the reference twist inside the window wins
  const DynamicWindow dwa(col, 0.1, 1.0, 0.2, 1.0, 1.0, 1.0, 1.0, -1.0, 1.0, -1.0, 2.0, -2.0, 3, 8, 5);
  {
    const auto [ok, u] = dwa.control(g, vec{ 3.5, 1.5, 0.0 }, vec{ 0.0, 0.0, 0.0 }, vec{ 0.2, 0.2, 0.2 });
    CHECK(ok);
    CHECK_NEAR(u(0), 0.2, 1e-12);   // window [-0.2, 0.2] in every component: upper corner
    CHECK_NEAR(u(1), 0.2, 1e-12);
    CHECK_NEAR(u(2), 0.2, 1e-12);
  }
  {
    // boxed in: robot right next to the obstacle block, every rollout collides
    const auto [ok, u] = dwa.control(g, vec{ 1.3, -1.75, 0.0 }, vec{ 0.0, 0.0, 0.0 }, vec{ 0.2, 0.0, 0.0 });
    CHECK(!ok);
    CHECK_NEAR(u(0), 0.0, 0.0);
  }
  {
    mat ref(3, 10);
    for (int i = 0; i < 10; ++i) {
      ref(0, i) = 3.5 + 0.02 * (i + 1);
      ref(1, i) = 1.5;
    }
    const auto [ok, u] = dwa.control(g, vec{ 3.5, 1.5, 0.0 }, vec{ 0.2, 0.0, 0.0 }, ref, 0.1);
    CHECK(ok);
    CHECK(u(0) > 0.0);
    CHECK_NEAR(u(2), 0.0, 1e-12);
  }
}

// closed loop of SURVEY.md 8(c): free 12 x 6 m map at 0.05 m, origin (-1,-1), two yaml Gaussians,
// x0 = (1, 1, 0.3), state advanced by one RK4 step of the model with the returned twist
template <class ModelT>
static void anchor(const char* name, double horizon, const double rinv[3], const double lim[3],
                   const double expect[3][3])
{
  const unsigned w = axis_length(-1.0, 11.0, 0.05), h = axis_length(-1.0, 5.0, 0.05);
  const GridMap grid(-1.0, 11.0, -1.0, 5.0, 0.05, GridData(static_cast<std::size_t>(w) * h, 0));
  const Collision col(0.7, 1.0, 0.2, 0.8);
  mat Rinv(3, 3);
  for (int i = 0; i < 3; ++i) Rinv(i, i) = rinv[i];
  const ModelT model;
  ErgodicControl<ModelT> ec(model, col, 0.1, horizon, 0.1, 1.0, 10, 1000000, 100, Rinv,
                            vec{ -lim[0], -lim[1], -lim[2] }, vec{ lim[0], lim[1], lim[2] });
  ec.setTarget(Target({ Gaussian({ 2.5, 2.5 }, { 1.5, 1.5 }), Gaussian({ 8.5, 2.5 }, { 1.5, 1.5 }) }));
  const RungeKutta rk(0.1);
  vec x = { 1.0, 1.0, 0.3 };
  for (int c = 0; c < 3; ++c) {
    const vec u = ec.control(grid, x);
    for (int r = 0; r < 3; ++r) {
      ++g_checks;
      if (!(std::fabs(u(r) - expect[c][r]) <= 1e-9 * std::pow(10.0, c))) {
        ++g_fail;
        std::printf("FAIL %s call %d u(%d) = %.17g expected %.17g\n", name, c, r, u(r), expect[c][r]);
      }
    }
    x = rk.step(model, x, vec{ expect[c][0], expect[c][1], expect[c][2] });
  }
  CHECK(ec.optTraj().n_cols() == ec.steps());
  CHECK(ec.path("map").poses.size() == ec.steps());
}

static void test_ergodic_control()
{
  const double omni_u[3][3] = { { 1.0, -0.30227161490709098, 0.0 },
                                { 0.20219981771352599, 0.67021770126265123, 2.0 },
                                { 1.0, 0.69098703688478214, -1.5092341259945798 } };
  const double r1[3] = { 1.0, 1.0, 2.0 }, l1[3] = { 1.0, 1.0, 2.0 };
  anchor<models::Omni>("omni K10 T50", 5.0, r1, l1, omni_u);
  const double cart_u[3][3] = { { 0.47588406673073996, 0.0, 0.0 },
                                { 0.1038633231399978, 0.0, 0.34119528795354898 },
                                { 0.22554267622399127, 0.0, 0.31640574292479129 } };
  const double r2[3] = { 1.0, 0.0, 2.0 }, l2[3] = { 1.0, 0.0, 2.0 };
  anchor<models::SimpleCart>("simple_cart K10 T20", 2.0, r2, l2, cart_u);

  // replay-memory path: Omni K5, horizon 2, 1 Gaussian, memory of 3 poses (no random draw)
  {
    const GridMap grid(0.0, 12.0, 0.0, 6.0, 0.1, GridData(120 * 60, 0));
    mat Rinv(3, 3);
    for (int i = 0; i < 3; ++i) Rinv(i, i) = 1.0;
    ErgodicControl<models::Omni> ec(models::Omni(), Collision(0.7, 1.0, 0.2, 0.8), 0.1, 2.0, 0.1, 1.0, 5, 1000000,
                                    100, Rinv, vec{ -1.0, -1.0, -2.0 }, vec{ 1.0, 1.0, 2.0 });
    ec.setTarget(Target({ Gaussian({ 2.5, 2.5 }, { 1.5, 1.5 }) }));
    ec.addStateMemory(vec{ 1.0, 1.0, 0.0 });
    ec.addStateMemory(vec{ 1.1, 1.0, 0.0 });
    ec.addStateMemory(vec{ 1.2, 1.0, 0.0 });
    const vec u = ec.control(grid, vec{ 1.0, 1.0, 0.3 });
    CHECK_NEAR(u(0), 0.76522786291044498, 1e-9);
    CHECK_NEAR(u(1), 0.29234914265332201, 1e-9);
    CHECK_NEAR(u(2), 0.0, 1e-9);
  }
  // setTarget skipped: the reference's default Target has no Gaussians, Target::fill divides 0 / 0 and control() returns NaN
  // (ergodic_control.hpp:411-413, SURVEY.md hazard 12) -- the class does the same (the C ABI alone answers EEA_ERR_NO_TARGET);
  // once a target is set the same object works normally
  {
    const GridMap grid(0.0, 12.0, 0.0, 6.0, 0.1, GridData(120 * 60, 0));
    mat Rinv(3, 3);
    for (int i = 0; i < 3; ++i) Rinv(i, i) = 1.0;
    ErgodicControl<models::Omni> ec(models::Omni(), Collision(0.7, 1.0, 0.2, 0.8), 0.1, 2.0, 0.1, 1.0, 5, 10, 10, Rinv,
                                    vec{ -1.0, -1.0, -2.0 }, vec{ 1.0, 1.0, 2.0 });
    const vec u = ec.control(grid, vec{ 1.0, 1.0, 0.3 });
    CHECK(std::isnan(u(0)) && std::isnan(u(1)) && std::isnan(u(2)));
    ErgodicControl<models::Omni> ec2(models::Omni(), Collision(0.7, 1.0, 0.2, 0.8), 0.1, 2.0, 0.1, 1.0, 5, 10, 10, Rinv,
                                     vec{ -1.0, -1.0, -2.0 }, vec{ 1.0, 1.0, 2.0 });
    ec2.setTarget(Target({ Gaussian({ 2.5, 2.5 }, { 1.5, 1.5 }) }));
    const vec v = ec2.control(grid, vec{ 1.0, 1.0, 0.3 });
    CHECK(std::isfinite(v(0)) && std::isfinite(v(1)) && std::isfinite(v(2)));
  }
  // horizon == dt is rejected at construction like the reference
  bool threw = false;
  try {
    mat Rinv(3, 3);
    ErgodicControl<models::Omni> ec(models::Omni(), Collision(0.7, 1.0, 0.2, 0.8), 0.1, 0.1, 0.1, 1.0, 5, 10, 10, Rinv,
                                    vec{ -1.0, -1.0, -1.0 }, vec{ 1.0, 1.0, 1.0 });
  } catch (const std::invalid_argument&) {
    threw = true;
  }
  CHECK(threw);
}

// AgentBatch: every agent of a batched step equals a single-agent ErgodicControl run from the same state;
// the exchange steps on a local communicator; the consensus input changes the controls
static void test_agent_batch()
{
  const unsigned int N = 37;
  mat Rinv(3, 3);
  Rinv(0, 0) = 1.0;
  Rinv(1, 1) = 1.0;
  Rinv(2, 2) = 2.0;
  const vec umin{ -1.0, -1.0, -2.0 }, umax{ 1.0, 1.0, 2.0 };
  const GridMap grid(-1.0, 11.0, -1.0, 5.0, 0.05, GridData(240 * 120, 0));
  const Target target({ Gaussian({ 2.5, 2.5 }, { 1.5, 1.5 }), Gaussian({ 8.5, 2.5 }, { 1.5, 1.5 }) });
  AgentBatch<models::Omni> batch(N, 0.1, 5.0, 0.1, 1.0, 10, Rinv, umin, umax);
  batch.setTarget(target);
  batch.configTarget(grid);
  mat poses(3, N);
  for (unsigned int a = 0; a < N; ++a) {
    poses(0, a) = 0.3 + 0.25 * a;
    poses(1, a) = 0.5 + 0.1 * (a % 30);
    poses(2, a) = -3.0 + 0.16 * a;
  }
  batch.setPoses(poses);
  batch.control();
  batch.control();
  const mat u = batch.controls();
  for (unsigned int a : { 0u, 5u, 36u }) {
    ErgodicControl<models::Omni> ec(models::Omni(), Collision(0.7, 1.0, 0.2, 0.8), 0.1, 5.0, 0.1, 1.0, 10, 1000000,
                                    100, Rinv, umin, umax);
    ec.setTarget(target);
    const vec x{ poses(0, a), poses(1, a), poses(2, a) };
    ec.control(grid, x);
    const vec ua = ec.control(grid, x);
    for (int r = 0; r < 3; ++r) CHECK_NEAR(u(r, a), ua(r), 1e-12);
  }
  const mat ck = batch.gatherTrajCoeff();
  CHECK(ck.n_rows() == 100 && ck.n_cols() == N);
  CHECK_NEAR(ck(0, 3), 1.0, 1e-14);  // mode (0,0) of every agent's c_k is the mean of ones
  batch.control(true);               // produces the consensus of this step ...
  const mat u_own = batch.controls();
  batch.control(true);               // ... which the next step's gradient uses
  const mat u_cons = batch.controls();
  double diff = 0.0;
  for (unsigned int a = 0; a < N; ++a) diff += std::fabs(u_cons(0, a) - u_own(0, a)) + std::fabs(u_cons(2, a) - u_own(2, a));
  CHECK(diff > 1e-6);
  // the consensus the records hold is the mean of the agents' own c_k of the last step
  {
    const vec cbar = batch.consensusTrajCoeff();
    const mat ck2 = batch.gatherTrajCoeff();
    double worst = 0.0;
    for (unsigned int m = 0; m < 100; ++m) {
      double mean = 0.0;
      for (unsigned int a = 0; a < N; ++a) mean += ck2(m, a);
      worst = std::max(worst, std::fabs(cbar(m) - mean / N));
    }
    CHECK(worst < 1e-14);
  }
  // Several consensus steps BACK TO BACK (no controls() / sync() in between: every group stream must be ordered
  // behind the exchange that produced the record it reads) give the same controls with two agent groups on two
  // streams as with one group on one stream: the per-agent records are added in agent order whatever the grouping.
  {
    mat us[2];
    for (unsigned int groups = 1; groups <= 2; ++groups) {
      AgentBatch<models::Omni> b2(N, 0.1, 5.0, 0.1, 1.0, 10, Rinv, umin, umax, nullptr, groups);
      b2.setTarget(target);
      b2.configTarget(grid);
      b2.setPoses(poses);
      for (int i = 0; i < 6; ++i) b2.control(true);
      us[groups - 1] = b2.controls();
    }
    double worst = 0.0;
    for (unsigned int a = 0; a < N; ++a) {
      for (int r = 0; r < 3; ++r) worst = std::max(worst, std::fabs(us[0](r, a) - us[1](r, a)));
    }
    CHECK(worst == 0.0);
  }
}

// AgentBatch::tick (eea_tick_batch: the loop body of exploration.hpp:220-279 for a fleet) against one single-robot
// Exploration<ModelT>::tick per agent (the mirror of the same loop body on the single-agent entry points, itself checked
// against the oracle by tests/test_host_mirror.py): same decisions every tick, same twists
static void test_agent_batch_tick()
{
  const unsigned int N = 9, ticks = 8;
  mat Rinv(3, 3);
  Rinv(0, 0) = 1.0;
  Rinv(1, 1) = 1.0;
  Rinv(2, 2) = 2.0;
  const vec umin{ -1.0, -1.0, -2.0 }, umax{ 1.0, 1.0, 2.0 };
  GridData cells(240 * 120, 0);
  for (unsigned int i = 24; i < 72; ++i) {       // an obstacle at x in [2.4, 3.0], y in [0.2, 2.6]
    for (unsigned int j = 68; j < 80; ++j) cells[i * 240 + j] = 100;
  }
  const GridMap grid(-1.0, 11.0, -1.0, 5.0, 0.05, cells);
  const Target target({ Gaussian({ 2.5, 2.5 }, { 1.5, 1.5 }), Gaussian({ 8.5, 2.5 }, { 1.5, 1.5 }) });
  const Collision collision(0.7, 1.0, 0.2, 0.8);
  const DynamicWindow dwa(collision, 0.1, 2.0, 0.2, 2.5, 2.5, 1.0, 1.0, -1.0, 1.0, -1.0, 2.0, -2.0, 3, 8, 5);
  AgentBatch<models::Omni> batch(N, 0.1, 5.0, 0.1, 1.0, 10, Rinv, umin, umax);
  batch.setTarget(target);
  std::vector<Exploration<models::Omni>> single;
  for (unsigned int a = 0; a < N; ++a) {
    // (batch_size 0: the replay buffer never contributes a column, buffer.cpp:92-108 -- the batch keeps no replay memory)
    ErgodicControl<models::Omni> ec(models::Omni(), collision, 0.1, 5.0, 0.1, 1.0, 10, 1000000, 0, Rinv, umin, umax);
    single.emplace_back(ec, collision, dwa);
    single.back().setTarget(target);
  }
  mat poses(3, N), vb(3, N);
  for (unsigned int a = 0; a < N; ++a) {
    poses(0, a) = 1.0 + 0.1 * a;     // heading for the obstacle
    poses(1, a) = 0.6 + 0.2 * a;
    poses(2, a) = -0.2 + 0.05 * a;
  }
  int dwa_ticks = 0;
  double worst = 0.0;
  for (unsigned int t = 0; t < ticks; ++t) {
    batch.setPoses(poses);
    const mat u = batch.tick(grid, dwa, vb, 0.1, 0.5, 1);
    const std::vector<int> src = batch.tickSources();
    for (unsigned int a = 0; a < N; ++a) {
      const vec x{ poses(0, a), poses(1, a), poses(2, a) }, v{ vb(0, a), vb(1, a), vb(2, a) };
      const vec us = single[a].tick(grid, x, v, 0.1, 0.5);
      CHECK(src[a] == static_cast<int>(single[a].source()));
      for (int r = 0; r < 3; ++r) worst = std::max(worst, std::fabs(u(r, a) - us(r)));
      dwa_ticks += src[a] != 0;
    }
    for (unsigned int a = 0; a < N; ++a) {
      const vec x{ poses(0, a), poses(1, a), poses(2, a) }, ua{ u(0, a), u(1, a), u(2, a) };
      const vec xn = integrate_twist(x, ua, 0.1);
      for (int r = 0; r < 3; ++r) {
        poses(r, a) = xn(r);
        vb(r, a) = u(r, a);
      }
    }
  }
  CHECK(worst < 1e-6);
  CHECK(dwa_ticks > 0);
  std::printf("  AgentBatch::tick: %u agents x %u ticks, decisions equal to the single-robot loops, |u diff| %.2e, DWA ticks %d\n",
              N, ticks, worst, dwa_ticks);
}

int main(int argc, char** argv)
{
  const std::string mode = argc > 1 ? argv[1] : "cpu";
  test_cart();
  test_mecanum();
  test_integrator();
  test_grid();
  if (mode == "gpu") {
    test_device_ops();
    test_ergodic_control();
    test_agent_batch();
    test_agent_batch_tick();
  }
  std::printf("%s: %d checks, %d failures\n", mode.c_str(), g_checks, g_fail);
  return g_fail ? 1 : 0;
}

// Scalar helpers of the reference's numerics.hpp that the control path uses on the host
// (reference include/ergodic_exploration/numerics.hpp:58-90, 164-179, 273-297).
#pragma once

#include <cmath>

#include <ergodic_exploration/types.hpp>

namespace ergodic_exploration
{
constexpr double PI = 3.14159265358979323846;

inline bool almost_equal(double d1, double d2, double epsilon = 1.0e-12)
{
  return std::fabs(d1 - d2) < epsilon;
}

// heading wrapped to [-pi, pi)
inline double normalize_angle_PI(double rad)
{
  const double q = std::floor((rad + PI) / (2.0 * PI));
  rad = (rad + PI) - q * 2.0 * PI;
  if (rad < 0.0) rad += 2.0 * PI;
  return rad - PI;
}

// entropy of one occupancy cell (p < 0: unknown)
inline double entropy(double p)
{
  if (almost_equal(0.0, p) || almost_equal(1.0, p)) return 1e-3;
  if (p < 0.0) return 0.7;
  return -p * std::log(p) - (1.0 - p) * std::log(1.0 - p);
}

// exact SE(2) integration of a constant body twist over dt
inline vec integrate_twist(const vec& x, const vec& u, double dt)
{
  double d0, d1, d2;
  if (almost_equal(u(2), 0.0)) {
    d0 = u(0) * dt;
    d1 = u(1) * dt;
    d2 = 0.0;
  } else {
    const double v0 = u(0) * dt, v1 = u(1) * dt, v2 = u(2) * dt;
    d0 = (v0 * std::sin(v2) + v1 * (std::cos(v2) - 1.0)) / v2;
    d1 = (v1 * std::sin(v2) + v0 * (1.0 - std::cos(v2))) / v2;
    d2 = v2;
  }
  const double c = std::cos(x(2)), s = std::sin(x(2));
  return { x(0) + (c * d0 - s * d1), x(1) + (s * d0 + c * d1), x(2) + d2 };
}
}  // namespace ergodic_exploration

// Kinematic models with the reference's duck-typed ModelT concept (operator(), fdx, fdu,
// state_space): reference include/ergodic_exploration/models/omni.hpp.  Omni is one of the two
// models the device engine implements (EEA_MODEL_OMNI); Mecanum is class surface only.
#pragma once

#include <cmath>

#include <ergodic_exploration/types.hpp>

namespace ergodic_exploration
{
namespace models
{
// body-twist omni-directional robot: state [x, y, theta], control [vx, vy, w]
struct Omni
{
  Omni() : state_space(3) {}
  vec operator()(const vec x, const vec u) const
  {
    const double c = std::cos(x(2)), s = std::sin(x(2));
    return { u(0) * c - u(1) * s, u(0) * s + u(1) * c, u(2) };
  }
  mat fdx(const vec x, const vec u) const
  {
    mat A(3, 3);
    const double c = std::cos(x(2)), s = std::sin(x(2));
    A(0, 2) = -u(0) * s - u(1) * c;
    A(1, 2) = u(0) * c - u(1) * s;
    return A;
  }
  mat fdu(const vec x) const
  {
    const double c = std::cos(x(2)), s = std::sin(x(2));
    return { { c, -s, 0.0 }, { s, c, 0.0 }, { 0.0, 0.0, 1.0 } };
  }
  unsigned int state_space;
};

// four mecanum wheels (front left, front right, rear right, rear left), rollers at 45 degrees
struct Mecanum
{
  Mecanum(double wheel_radius, double wheel_base_x, double wheel_base_y)
    : wheel_radius(wheel_radius), wheel_base_x(wheel_base_x), wheel_base_y(wheel_base_y), state_space(3)
  {
  }
  vec wheels2Twist(const vec u) const
  {
    const double l = 1.0 / (wheel_base_x + wheel_base_y), q = wheel_radius / 4.0;
    return { q * (u(0) + u(1) + u(2) + u(3)), q * (-u(0) + u(1) - u(2) + u(3)),
             q * l * (-u(0) + u(1) + u(2) - u(3)) };
  }
  vec operator()(const vec x, const vec u) const
  {
    const double s = (wheel_radius / 4.0) * std::sin(x(2)), c = (wheel_radius / 4.0) * std::cos(x(2));
    const double l = wheel_radius / (4.0 * (wheel_base_x + wheel_base_y));
    return { u(0) * (s + c) + u(1) * (-s + c) + u(2) * (s + c) + u(3) * (-s + c),
             u(0) * (s - c) + u(1) * (s + c) + u(2) * (s - c) + u(3) * (s + c),
             -u(0) * l + u(1) * l + u(2) * l - u(3) * l };
  }
  mat fdx(const vec x, const vec u) const
  {
    mat A(3, 3);
    const double s = (wheel_radius / 4.0) * std::sin(x(2)), c = (wheel_radius / 4.0) * std::cos(x(2));
    A(0, 2) = u(0) * (-s + c) + u(1) * (-s - c) + u(2) * (-s + c) + u(3) * (-s - c);
    A(1, 2) = u(0) * (s + c) + u(1) * (-s + c) + u(2) * (s + c) + u(3) * (-s + c);
    return A;
  }
  mat fdu(const vec x) const
  {
    const double s = (wheel_radius / 4.0) * std::sin(x(2)), c = (wheel_radius / 4.0) * std::cos(x(2));
    const double l = wheel_radius / (4.0 * (wheel_base_x + wheel_base_y));
    return { { s + c, -s + c, s + c, -s + c }, { s - c, s + c, s - c, s + c }, { -l, l, l, -l } };
  }
  double wheel_radius, wheel_base_x, wheel_base_y;
  unsigned int state_space;
};
}  // namespace models
}  // namespace ergodic_exploration

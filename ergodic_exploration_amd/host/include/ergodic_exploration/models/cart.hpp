// Differential-drive models (reference include/ergodic_exploration/models/cart.hpp).  SimpleCart
// (body twist [vx, 0, w]) is the model the device engine implements (EEA_MODEL_SIMPLE_CART);
// Cart (wheel velocities) is class surface only: it cannot run under ErgodicControl upstream.
#pragma once

#include <cmath>
#include <stdexcept>

#include <ergodic_exploration/numerics.hpp>
#include <ergodic_exploration/types.hpp>

namespace ergodic_exploration
{
namespace models
{
struct Cart
{
  Cart(double wheel_radius, double wheel_base) : wheel_radius(wheel_radius), wheel_base(wheel_base), state_space(3) {}
  vec wheels2Twist(const vec u) const
  {
    return { wheel_radius / 2.0 * (u(0) + u(1)), 0.0, wheel_radius / (2.0 * wheel_base) * (u(1) - u(0)) };
  }
  vec operator()(const vec x, const vec u) const
  {
    const double h = wheel_radius / 2.0;
    return { h * ((u(0) + u(1)) * std::cos(x(2))), h * ((u(0) + u(1)) * std::sin(x(2))),
             h * ((u(1) - u(0)) / wheel_base) };
  }
  mat fdx(const vec x, const vec u) const
  {
    mat A(3, 3);
    A(0, 2) = -(wheel_radius / 2.0) * (u(0) + u(1)) * std::sin(x(2));
    A(1, 2) = (wheel_radius / 2.0) * (u(0) + u(1)) * std::cos(x(2));
    return A;
  }
  mat fdu(const vec x) const
  {
    const double h = wheel_radius / 2.0, c = std::cos(x(2)), s = std::sin(x(2));
    return { { h * c, h * c }, { h * s, h * s }, { h * (-1.0 / wheel_base), h * (1.0 / wheel_base) } };
  }
  double wheel_radius, wheel_base;
  unsigned int state_space;
};

struct SimpleCart
{
  SimpleCart() : state_space(3) {}
  vec operator()(const vec x, const vec u) const
  {
    if (!almost_equal(u(1), 0.0)) throw std::invalid_argument("Invalid twist y-velocity must be 0.");
    return { u(0) * std::cos(x(2)), u(0) * std::sin(x(2)), u(2) };
  }
  mat fdx(const vec x, const vec u) const
  {
    mat A(3, 3);
    A(0, 2) = -u(0) * std::sin(x(2));
    A(1, 2) = u(0) * std::cos(x(2));
    return A;
  }
  mat fdu(const vec x) const
  {
    mat B(3, 3);
    B(0, 0) = std::cos(x(2));
    B(1, 0) = std::sin(x(2));
    B(2, 2) = 1.0;
    return B;
  }
  unsigned int state_space;
};
}  // namespace models
}  // namespace ergodic_exploration

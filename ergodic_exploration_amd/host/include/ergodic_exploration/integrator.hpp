// RungeKutta (reference include/ergodic_exploration/integrator.hpp:62-194).
// solve() for the body-twist models Omni / SimpleCart runs on the device (eea_rk4_rollout:
// closed-form RK4 as two prefix sums); any other ModelT (Cart, Mecanum, user models) is
// integrated by the generic host template below, which is class surface, not the hot path.
// The co-state overloads are generic host templates: inside ErgodicControl::control the
// backward pass never leaves the fused device kernel.
#pragma once

#include <cmath>
#include <functional>

#include <ergodic_exploration/device.hpp>
#include <ergodic_exploration/numerics.hpp>

namespace ergodic_exploration
{
typedef std::function<vec(const vec&, const vec&, const vec&, const mat&)> CoStateFunc;

class RungeKutta
{
public:
  explicit RungeKutta(double dt) : dt_(dt) {}

  // forward rollout; the initial state is not part of the output, headings wrapped to [-pi, pi)
  template <class ModelT>
  mat solve(const ModelT& model, const vec& x0, const mat& ut, double horizon) const
  {
    const auto steps = static_cast<unsigned int>(std::abs(horizon / dt_));
    mat xt(x0.size(), steps);
    if constexpr (device_model<ModelT>::value >= 0) {
      if (steps > 0) {
        throw_on_error(eea_rk4_rollout(device_ordinal(), device_model<ModelT>::value, dt_, horizon,
                                       x0.memptr(), ut.memptr(), xt.memptr()));
      }
      return xt;
    } else {
      vec x = x0;
      for (unsigned int i = 0; i < steps; i++) {
        x = step(model, x, ut.col(i));
        x(2) = normalize_angle_PI(x(2));
        xt.set_col(i, x);
      }
      return xt;
    }
  }

  // co-state backwards in time; rhot(:, i) is the co-state after the step at index i
  template <class ModelT>
  mat solve(const CoStateFunc& func, const ModelT& model, const vec& rhoT, const mat& xt, const mat& ut,
            const mat& edx, const mat& bdx, double horizon) const
  {
    const auto steps = static_cast<unsigned int>(std::abs(horizon / dt_));
    vec rho = rhoT;
    mat rhot(rho.size(), steps);
    for (unsigned int i = steps; i-- > 0;) {
      rho = step(func, rho, edx.col(i), bdx.col(i), model.fdx(xt.col(i), ut.col(i)));
      rhot.set_col(i, rho);
    }
    return rhot;
  }

  template <class ModelT>
  vec step(const ModelT& model, const vec& x, const vec& u) const
  {
    const vec k1 = model(x, u);
    const vec k2 = model(axpy(x, dt_ * 0.5, k1), u);
    const vec k3 = model(axpy(x, dt_ * 0.5, k2), u);
    const vec k4 = model(axpy(x, dt_, k3), u);
    vec out(x.size());
    for (std::size_t i = 0; i < x.size(); ++i) {
      out(i) = x(i) + (dt_ / 6.0) * (((k1(i) + 2.0 * k2(i)) + 2.0 * k3(i)) + k4(i));
    }
    return out;
  }

  vec step(const CoStateFunc& func, const vec& rho, const vec& gdx, const vec& dbar, const mat& fdx) const
  {
    const vec k1 = func(rho, gdx, dbar, fdx);
    const vec k2 = func(axpy(rho, -dt_ * 0.5, k1), gdx, dbar, fdx);
    const vec k3 = func(axpy(rho, -dt_ * 0.5, k2), gdx, dbar, fdx);
    const vec k4 = func(axpy(rho, -dt_, k3), gdx, dbar, fdx);
    vec out(rho.size());
    for (std::size_t i = 0; i < rho.size(); ++i) {
      out(i) = rho(i) - dt_ / 6.0 * (((k1(i) + 2.0 * k2(i)) + 2.0 * k3(i)) + k4(i));
    }
    return out;
  }

private:
  static vec axpy(const vec& x, double a, const vec& k)
  {
    vec r(x.size());
    for (std::size_t i = 0; i < x.size(); ++i) r(i) = x(i) + a * k(i);
    return r;
  }
  double dt_;
};
}  // namespace ergodic_exploration

// ErgodicControl<ModelT>: receding-horizon ergodic trajectory optimisation.  Same public surface
// as the reference class (include/ergodic_exploration/ergodic_control.hpp:72-185); every call is
// forwarded to the gfx950 engine through the C ABI (include/ergodic_amd.h).  The engine owns the
// warm-start controls, phi_k and lambda_k on the device; this object owns the host-side pieces
// (replay memory, target description) exactly like the reference.
#pragma once

#include <cmath>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include <ergodic_exploration/basis.hpp>
#include <ergodic_exploration/buffer.hpp>
#include <ergodic_exploration/collision.hpp>
#include <ergodic_exploration/integrator.hpp>
#include <ergodic_exploration/target.hpp>

namespace ergodic_exploration
{
// planned path with yaw quaternions (stands in for nav_msgs::Path: ROS is out of scope)
struct PathPose
{
  double x, y, qx, qy, qz, qw;
};
struct Path
{
  std::string frame_id;
  std::vector<PathPose> poses;
};

template <class ModelT>
class ErgodicControl
{
  static_assert(device_model<ModelT>::value >= 0,
                "ErgodicControl needs a body-twist model (models::Omni or models::SimpleCart): the control "
                "signal is 3 x steps and the limits index 3 twist components");

public:
  ErgodicControl(const ModelT& model, const Collision& collision, double dt, double horizon, double resolution,
                 double exploration_weight, unsigned int num_basis, unsigned int buffer_size,
                 unsigned int batch_size, const mat& Rinv, const vec& umin, const vec& umax,
                 int precision = EEA_PREC_F64)
    : model_(model)
    , collision_(collision)
    , dt_(dt)
    , steps_(static_cast<unsigned int>(std::abs(horizon / dt)))
    , buffer_(buffer_size, batch_size)
  {
    eea_config cfg{};
    cfg.model = device_model<ModelT>::value;
    cfg.precision = precision;
    cfg.device = device_ordinal();
    cfg.dt = dt;
    cfg.horizon = horizon;
    cfg.resolution = resolution;
    cfg.expl_weight = exploration_weight;
    cfg.num_basis = num_basis;
    for (int i = 0; i < 9; ++i) cfg.Rinv[i] = Rinv.memptr()[i];
    for (int i = 0; i < 3; ++i) {
      cfg.umin[i] = umin(i);
      cfg.umax[i] = umax(i);
    }
    eea_engine* e = nullptr;
    throw_on_error(eea_create(&cfg, &e));  // std::invalid_argument when steps == 1, like the reference
    engine_ = std::shared_ptr<eea_engine>(e, eea_destroy);
    // The reference's member is a default-constructed Target -- no Gaussians -- until setTarget is called, and control()
    // then divides 0 / 0 in Target::fill (target.cpp:87) and returns NaN (ergodic_control.hpp:411-413).  The C ABI answers
    // a control call without a target with EEA_ERR_NO_TARGET; the CLASS reproduces the reference: it installs the same
    // default target (an empty Gaussian list is a target whose grid is 0 / 0), so control() before setTarget is NaN here too.
    throw_on_error(eea_set_target_gaussians(engine_.get(), 0u, nullptr, nullptr));
  }

  // one receding-horizon optimisation; returns the first twist of the updated control signal
  vec control(const GridMap& grid, const vec& x)
  {
    const mat mem = buffer_.sampleColumns();
    vec u(3);
    throw_on_error(eea_control(engine_.get(), grid.xmin(), grid.xmax(), grid.ymin(), grid.ymax(), x.memptr(),
                               mem.n_cols() ? mem.memptr() : nullptr, static_cast<unsigned>(mem.n_cols()),
                               u.memptr()));
    return u;
  }

  // rollout of the current control signal from the last pose
  mat optTraj() const
  {
    mat traj(3, steps_);
    throw_on_error(eea_opt_traj(engine_.get(), traj.memptr()));
    return traj;
  }

  Path path(const std::string& map_frame_id) const
  {
    Path p;
    p.frame_id = map_frame_id;
    p.poses.resize(steps_);
    const mat traj = optTraj();
    for (unsigned int i = 0; i < steps_; i++) {
      const double yaw = normalize_angle_PI(traj(2, i));
      p.poses[i] = { traj(0, i), traj(1, i), 0.0, 0.0, std::sin(0.5 * yaw), std::cos(0.5 * yaw) };
    }
    return p;
  }

  void addStateMemory(const vec& x) { buffer_.append(x); }
  double timeStep() const { return dt_; }

  void setTarget(const Target& target)
  {
    target_ = target;
    std::vector<double> mu, sg;
    target.flatten(mu, sg);
    throw_on_error(eea_set_target_gaussians(engine_.get(), static_cast<unsigned>(mu.size() / 2), mu.data(),
                                            sg.data()));
  }

  // refreshes the map origin; rebuilds phi_k only when the map extent changed
  void configTarget(const GridMap& grid)
  {
    throw_on_error(eea_config_domain(engine_.get(), grid.xmin(), grid.xmax(), grid.ymin(), grid.ymax(), nullptr,
                                     nullptr));
  }

  // additions of the mirror: device state inspection
  vec phik() const
  {
    vec v(eea_num_modes(engine_.get()));
    throw_on_error(eea_get_phik(engine_.get(), v.memptr()));
    return v;
  }
  mat controls() const
  {
    mat ut(3, steps_);
    throw_on_error(eea_get_ut(engine_.get(), ut.memptr()));
    return ut;
  }
  unsigned int steps() const { return steps_; }
  eea_engine* engine() const { return engine_.get(); }

private:
  ModelT model_;
  Collision collision_;  // carried like the reference does; not read on this path
  double dt_;
  unsigned int steps_;
  ReplayBuffer buffer_;
  Target target_;
  std::shared_ptr<eea_engine> engine_;  // copies of the controller share one device engine
};
}  // namespace ergodic_exploration

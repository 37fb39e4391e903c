// DynamicWindow: dynamic-window local planner (reference dynamic_window.hpp / dynamic_window.cpp):
// velocity window from the current twist and the acceleration limits, a vx x vy x vth sample grid,
// constant-twist rollouts with a collision check per step, and either the control-error cost or
// the distance-to-reference-trajectory cost.  The search runs in the batched device kernel
// (eea_dwa_control_batch), one lane per velocity sample.
#pragma once

#include <hip/hip_runtime_api.h>

#include <cmath>
#include <iostream>
#include <algorithm>
#include <cstring>
#include <tuple>

#include <ergodic_exploration/collision.hpp>

namespace ergodic_exploration
{
class DynamicWindow
{
public:
  DynamicWindow(const Collision& collision, double dt, double horizon, double acc_dt, double acc_lim_x,
                double acc_lim_y, double acc_lim_th, double max_vel_x, double min_vel_x, double max_vel_y,
                double min_vel_y, double max_rot_vel, double min_rot_vel, unsigned int vx_samples,
                unsigned int vy_samples, unsigned int vth_samples)
    : collision_(collision)
  {
    cfg_.dt = dt;
    cfg_.horizon = horizon;
    cfg_.acc_dt = acc_dt;
    cfg_.acc_lim_x = acc_lim_x;
    cfg_.acc_lim_y = acc_lim_y;
    cfg_.acc_lim_th = acc_lim_th;
    cfg_.max_vel_x = max_vel_x;
    cfg_.min_vel_x = min_vel_x;
    cfg_.max_vel_y = max_vel_y;
    cfg_.min_vel_y = min_vel_y;
    cfg_.max_rot_vel = max_rot_vel;
    cfg_.min_rot_vel = min_rot_vel;
    cfg_.vx_samples = at_least_one(vx_samples, "vx");
    cfg_.vy_samples = at_least_one(vy_samples, "vy");
    cfg_.vth_samples = at_least_one(vth_samples, "vth");
  }

  // best twist for tracking the reference twist vref; {false, 0} if every sample collides
  std::tuple<bool, vec> control(const GridMap& grid, const vec& x0, const vec& vb, const vec& vref) const
  {
    return run(grid, x0, vb, &vref, nullptr, 0.0);
  }

  // best twist for following the reference trajectory xt_ref (3 x n), sampled every dt_ref
  std::tuple<bool, vec> control(const GridMap& grid, const vec& x0, const vec& vb, const mat& xt_ref,
                                double dt_ref) const
  {
    return run(grid, x0, vb, nullptr, &xt_ref, dt_ref);
  }

  // rollout length of the planner (dynamic_window.hpp accessors used by Exploration)
  unsigned int steps() const { return static_cast<unsigned int>(std::abs(cfg_.horizon / cfg_.dt)); }
  double timeStep() const { return cfg_.dt; }
  double horizon() const { return cfg_.horizon; }
  // the planner's configuration as the C ABI takes it (AgentBatch::tick: eea_tick_batch)
  const eea_dwa_cfg& deviceConfig() const { return cfg_; }
  const Collision& collision() const { return collision_; }

private:
  static unsigned int at_least_one(unsigned int n, const char* name)
  {
    if (n == 0) {
      std::cout << name << " samples set to 0 but need at least 1... setting this to 1" << std::endl;
      return 1;
    }
    return n;
  }
  std::tuple<bool, vec> run(const GridMap& grid, const vec& x0, const vec& vb, const vec* vref, const mat* xt_ref,
                            double dt_ref) const
  {
    const size_t n_ref = xt_ref ? xt_ref->n_cols() : 0;
    const int8_t* const d_grid = device_cells(grid);
    // pinned block the kernel reads and writes directly: x0 | vb | vref | u_opt | found; the reference
    // trajectory (read by every sample at every step) goes to device memory
    const size_t n_head = 13;
    const PinnedScratch sc = pinned_scratch(sizeof(double) * n_head);
    double* const h = static_cast<double*>(sc.host);
    double* const d_buf = static_cast<double*>(sc.dev);
    for (int c = 0; c < 3; ++c) {
      h[c] = x0(c);
      h[3 + c] = vb(c);
      h[6 + c] = vref ? (*vref)(c) : 0.0;
    }
    double* d_ref = nullptr;
    if (xt_ref) {
      d_ref = static_cast<double*>(device_scratch(sizeof(double) * 3 * n_ref));
      hip_check(hipMemcpyAsync(d_ref, xt_ref->memptr(), sizeof(double) * 3 * n_ref, hipMemcpyHostToDevice, nullptr));
    }
    const eea_collision_cfg ccfg = collision_.deviceConfig(grid);
    const eea_status st =
        eea_dwa_control_batch(device_ordinal(), &ccfg, &cfg_, d_grid, d_buf, d_buf + 3, vref ? d_buf + 6 : nullptr, d_ref,
                              static_cast<unsigned>(n_ref), dt_ref, 1, d_buf + 9, reinterpret_cast<int*>(d_buf + 12),
                              nullptr);
    throw_on_error(st);
    hip_check(hipStreamSynchronize(nullptr));  // also covers the pageable copy above
    vec u(3);
    for (int c = 0; c < 3; ++c) u(c) = h[9 + c];
    int found = 0;
    std::memcpy(&found, &h[12], sizeof(int));
    if (!found) std::cout << "DWA Failed! Not even 1 solution found" << std::endl;
    return std::make_tuple(found != 0, u);
  }

  Collision collision_;
  eea_dwa_cfg cfg_;
};
}  // namespace ergodic_exploration

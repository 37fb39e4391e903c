// DynamicWindow: dynamic-window local planner (reference dynamic_window.hpp / dynamic_window.cpp):
// velocity window from the current twist and the acceleration limits, a vx x vy x vth sample grid,
// constant-twist rollouts with a collision check per step, and either the control-error cost or
// the distance-to-reference-trajectory cost.  The search runs in the batched device kernel
// (eea_dwa_control_batch), one lane per velocity sample.
#pragma once

#include <hip/hip_runtime_api.h>

#include <cmath>
#include <iostream>
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

private:
  static unsigned int at_least_one(unsigned int n, const char* name)
  {
    if (n == 0) {
      std::cout << name << " samples set to 0 but need at least 1... setting this to 1" << std::endl;
      return 1;
    }
    return n;
  }
  static void check(hipError_t e)
  {
    if (e != hipSuccess) throw std::runtime_error(std::string("hip: ") + hipGetErrorString(e));
  }
  std::tuple<bool, vec> run(const GridMap& grid, const vec& x0, const vec& vb, const vec* vref, const mat* xt_ref,
                            double dt_ref) const
  {
    check(hipSetDevice(device_ordinal()));
    const size_t cells = grid.gridData().size();
    const size_t n_ref = xt_ref ? xt_ref->n_cols() : 0;
    int8_t* d_grid = nullptr;
    double* d_buf = nullptr;  // x0 | vb | vref | u_opt | xt_ref
    int* d_found = nullptr;
    check(hipMalloc(reinterpret_cast<void**>(&d_grid), cells ? cells : 1));
    check(hipMalloc(reinterpret_cast<void**>(&d_buf), sizeof(double) * (12 + 3 * n_ref)));
    check(hipMalloc(reinterpret_cast<void**>(&d_found), sizeof(int)));
    if (cells) check(hipMemcpy(d_grid, grid.gridData().data(), cells, hipMemcpyHostToDevice));
    check(hipMemcpy(d_buf, x0.memptr(), sizeof(double) * 3, hipMemcpyHostToDevice));
    check(hipMemcpy(d_buf + 3, vb.memptr(), sizeof(double) * 3, hipMemcpyHostToDevice));
    if (vref) check(hipMemcpy(d_buf + 6, vref->memptr(), sizeof(double) * 3, hipMemcpyHostToDevice));
    if (xt_ref) check(hipMemcpy(d_buf + 12, xt_ref->memptr(), sizeof(double) * 3 * n_ref, hipMemcpyHostToDevice));
    const eea_collision_cfg ccfg = collision_.deviceConfig(grid);
    const eea_status st =
        eea_dwa_control_batch(device_ordinal(), &ccfg, &cfg_, d_grid, d_buf, d_buf + 3, vref ? d_buf + 6 : nullptr,
                              xt_ref ? d_buf + 12 : nullptr, static_cast<unsigned>(n_ref), dt_ref, 1, d_buf + 9,
                              d_found, nullptr);
    vec u(3);
    int found = 0;
    if (st == EEA_OK) {
      check(hipMemcpy(u.memptr(), d_buf + 9, sizeof(double) * 3, hipMemcpyDeviceToHost));
      check(hipMemcpy(&found, d_found, sizeof(int), hipMemcpyDeviceToHost));
    }
    (void)hipFree(d_grid);
    (void)hipFree(d_buf);
    (void)hipFree(d_found);
    throw_on_error(st);
    if (!found) std::cout << "DWA Failed! Not even 1 solution found" << std::endl;
    return std::make_tuple(found != 0, u);
  }

  Collision collision_;
  eea_dwa_cfg cfg_;
};
}  // namespace ergodic_exploration

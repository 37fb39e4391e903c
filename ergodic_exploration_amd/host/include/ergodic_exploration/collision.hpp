// Collision: Bresenham ring search around the robot cell (reference collision.hpp /
// collision.cpp).  The search itself runs in the batched device kernels
// (eea_collision_check_batch / eea_validate_control_batch); this class owns the parameters and a
// device copy of the last grid it was asked about.
#pragma once

#include <hip/hip_runtime_api.h>

#include <memory>
#include <stdexcept>
#include <vector>

#include <ergodic_exploration/device.hpp>
#include <ergodic_exploration/grid.hpp>

namespace ergodic_exploration
{
class Collision
{
public:
  Collision(double boundary_radius, double search_radius, double obstacle_threshold, double occupied_threshold)
    : boundary_radius_(boundary_radius)
    , search_radius_(search_radius)
    , obstacle_threshold_(obstacle_threshold)
    , occupied_threshold_(occupied_threshold)
  {
    if (search_radius_ < boundary_radius_) {
      throw std::invalid_argument("Search radius must be at least the same size as the boundary radius");
    }
    if (occupied_threshold_ > 100.0 || occupied_threshold_ < 0.0) {
      throw std::invalid_argument("Occupied threshold must be between 0 and 100");
    }
  }

  // true if the pose is in collision
  bool collisionCheck(const GridMap& grid, const vec& pose) const
  {
    mat p(3, 1);
    p.set_col(0, pose);
    return collisionCheck(grid, p).at(0);
  }

  // batched form: poses 3 x P
  std::vector<bool> collisionCheck(const GridMap& grid, const mat& poses) const
  {
    const unsigned P = static_cast<unsigned>(poses.n_cols());
    std::vector<int> hit(P, 0);
    if (P == 0) return {};
    Buffers b(grid, P, 1);
    check(hipMemcpy(b.pose, poses.memptr(), sizeof(double) * 3 * P, hipMemcpyHostToDevice));
    const eea_collision_cfg cfg = config(grid);
    throw_on_error(eea_collision_check_batch(device_ordinal(), &cfg, b.grid, b.pose, P, b.out, nullptr));
    check(hipMemcpy(hit.data(), b.out, sizeof(int) * P, hipMemcpyDeviceToHost));
    return std::vector<bool>(hit.begin(), hit.end());
  }

  // validate_control (reference numerics.hpp:312-330) for P (state, twist) pairs: true = collision free
  std::vector<bool> validateControl(const GridMap& grid, const mat& x0, const mat& u, double dt, double horizon) const
  {
    const unsigned P = static_cast<unsigned>(x0.n_cols());
    std::vector<int> ok(P, 0);
    if (P == 0) return {};
    Buffers b(grid, P, 2);
    check(hipMemcpy(b.pose, x0.memptr(), sizeof(double) * 3 * P, hipMemcpyHostToDevice));
    check(hipMemcpy(b.pose + 3 * P, u.memptr(), sizeof(double) * 3 * P, hipMemcpyHostToDevice));
    const eea_collision_cfg cfg = config(grid);
    throw_on_error(eea_validate_control_batch(device_ordinal(), &cfg, b.grid, b.pose, b.pose + 3 * P, dt, horizon,
                                              P, b.out, nullptr));
    check(hipMemcpy(ok.data(), b.out, sizeof(int) * P, hipMemcpyDeviceToHost));
    return std::vector<bool>(ok.begin(), ok.end());
  }

  double totalPadding() const { return boundary_radius_ + obstacle_threshold_; }
  // geometry + thresholds in the C-ABI layout
  eea_collision_cfg deviceConfig(const GridMap& grid) const { return config(grid); }

private:
  static void check(hipError_t e)
  {
    if (e != hipSuccess) throw std::runtime_error(std::string("hip: ") + hipGetErrorString(e));
  }
  eea_collision_cfg config(const GridMap& grid) const
  {
    eea_collision_cfg c;
    c.xmin = grid.xmin();
    c.ymin = grid.ymin();
    c.resolution = grid.resolution();
    c.xsize = grid.xsize();
    c.ysize = grid.ysize();
    c.boundary_radius = boundary_radius_;
    c.search_radius = search_radius_;
    c.obstacle_threshold = obstacle_threshold_;
    c.occupied_threshold = occupied_threshold_;
    return c;
  }
  // scoped device buffers of one call
  struct Buffers
  {
    int8_t* grid = nullptr;
    double* pose = nullptr;
    int* out = nullptr;
    Buffers(const GridMap& g, unsigned P, unsigned pose_sets)
    {
      check(hipSetDevice(device_ordinal()));
      const size_t cells = g.gridData().size();
      check(hipMalloc(reinterpret_cast<void**>(&grid), cells ? cells : 1));
      check(hipMalloc(reinterpret_cast<void**>(&pose), sizeof(double) * 3 * P * pose_sets));
      check(hipMalloc(reinterpret_cast<void**>(&out), sizeof(int) * P));
      if (cells) check(hipMemcpy(grid, g.gridData().data(), cells, hipMemcpyHostToDevice));
    }
    ~Buffers()
    {
      (void)hipFree(grid);
      (void)hipFree(pose);
      (void)hipFree(out);
    }
  };
  double boundary_radius_, search_radius_, obstacle_threshold_, occupied_threshold_;
};

// numerics.hpp:312-330 of the reference for a single (state, twist)
inline bool validate_control(const Collision& collision, const GridMap& grid, const vec& x0, const vec& u, double dt,
                             double horizon)
{
  mat X(3, 1), U(3, 1);
  X.set_col(0, x0);
  U.set_col(0, u);
  return collision.validateControl(grid, X, U, dt, horizon).at(0);
}
}  // namespace ergodic_exploration

// Collision: Bresenham ring search around the robot cell (reference collision.hpp /
// collision.cpp).  The search itself runs in the batched device kernels
// (eea_collision_check_batch / eea_validate_control_batch); this class owns the parameters.  The
// device copy of a map's cells lives with the GridMap (device_cells), call arguments in a per-thread
// scratch buffer: a call costs two small copies and one launch, no allocation.
#pragma once

#include <hip/hip_runtime_api.h>

#include <cstring>
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
    if (P == 0) return {};
    const int8_t* const d_grid = device_cells(grid);
    const size_t pose_bytes = sizeof(double) * 3 * P;
    const PinnedScratch sc = pinned_scratch(pose_bytes + sizeof(int) * P);
    std::memcpy(sc.host, poses.memptr(), pose_bytes);
    const eea_collision_cfg cfg = config(grid);
    throw_on_error(eea_collision_check_batch(device_ordinal(), &cfg, d_grid, static_cast<const double*>(sc.dev), P,
                                             reinterpret_cast<int*>(static_cast<char*>(sc.dev) + pose_bytes), nullptr));
    hip_check(hipStreamSynchronize(nullptr));
    const int* const hit = reinterpret_cast<const int*>(static_cast<const char*>(sc.host) + pose_bytes);
    return std::vector<bool>(hit, hit + P);
  }

  // validate_control (reference numerics.hpp:312-330) for P (state, twist) pairs: true = collision free
  std::vector<bool> validateControl(const GridMap& grid, const mat& x0, const mat& u, double dt, double horizon) const
  {
    const unsigned P = static_cast<unsigned>(x0.n_cols());
    if (P == 0) return {};
    const int8_t* const d_grid = device_cells(grid);
    const size_t pose_bytes = sizeof(double) * 3 * P;
    const PinnedScratch sc = pinned_scratch(2 * pose_bytes + sizeof(int) * P);
    std::memcpy(sc.host, x0.memptr(), pose_bytes);
    std::memcpy(static_cast<char*>(sc.host) + pose_bytes, u.memptr(), pose_bytes);
    const eea_collision_cfg cfg = config(grid);
    char* const dev = static_cast<char*>(sc.dev);
    throw_on_error(eea_validate_control_batch(device_ordinal(), &cfg, d_grid, reinterpret_cast<const double*>(dev),
                                              reinterpret_cast<const double*>(dev + pose_bytes), dt, horizon, P,
                                              reinterpret_cast<int*>(dev + 2 * pose_bytes), nullptr));
    hip_check(hipStreamSynchronize(nullptr));
    const int* const ok = reinterpret_cast<const int*>(static_cast<const char*>(sc.host) + 2 * pose_bytes);
    return std::vector<bool>(ok, ok + P);
  }

  double totalPadding() const { return boundary_radius_ + obstacle_threshold_; }
  // geometry + thresholds in the C-ABI layout
  eea_collision_cfg deviceConfig(const GridMap& grid) const { return config(grid); }

private:
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

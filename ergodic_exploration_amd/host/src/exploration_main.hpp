// Shared body of the exploration_omni / exploration_cart entry points (the reference's
// src/exploration_{omni,cart}_node.cpp without ROS): read the parameter set of the reference's
// yaml files, wire Collision / ErgodicControl / Target the way the node mains do, then run the
// Exploration loop (exploration.hpp:197-292: addStateMemory -> control -> validate_control -> DWA
// fallback) against a simulated robot on a map with optional rectangular obstacles.  Prints one
// line per tick.
#pragma once

#include <cstdio>
#include <array>
#include <cstring>
#include <string>
#include <vector>

#include <ergodic_exploration/exploration.hpp>

#include "params.hpp"

namespace ee = ergodic_exploration;

template <class ModelT>
int exploration_main(int argc, char** argv, bool is_cart)
{
  params::Store pnh;
  int ticks = 20;
  double map_x0 = -1.0, map_y0 = -1.0, map_w = 12.0, map_h = 6.0, map_res = 0.05;
  ee::vec pose = { 1.0, 1.0, 0.3 };
  std::vector<std::array<double, 4>> obstacles;  // x0 y0 x1 y1 in the map frame
  for (int i = 1; i < argc; ++i) {
    const std::string a = argv[i];
    if (a == "--params" && i + 1 < argc) pnh.load(argv[++i]);
    else if (a == "--ticks" && i + 1 < argc) ticks = std::atoi(argv[++i]);
    else if (a == "--pose" && i + 3 < argc) {
      pose = { std::atof(argv[i + 1]), std::atof(argv[i + 2]), std::atof(argv[i + 3]) };
      i += 3;
    } else if (a == "--map" && i + 5 < argc) {
      map_x0 = std::atof(argv[i + 1]);
      map_y0 = std::atof(argv[i + 2]);
      map_w = std::atof(argv[i + 3]);
      map_h = std::atof(argv[i + 4]);
      map_res = std::atof(argv[i + 5]);
      i += 5;
    } else if (a == "--obstacle" && i + 4 < argc) {
      obstacles.push_back({ std::atof(argv[i + 1]), std::atof(argv[i + 2]), std::atof(argv[i + 3]),
                            std::atof(argv[i + 4]) });
      i += 4;
    } else if (a == "--set" && i + 2 < argc) {
      pnh.set(argv[i + 1], argv[i + 2]);
      i += 2;
    } else if (a == "--help") {
      std::printf("usage: %s [--params file.yaml] [--ticks N] [--pose x y th] [--map x0 y0 w h res] "
                  "[--obstacle x0 y0 x1 y1]... [--set name value]\n", argv[0]);
      return 0;
    }
  }

  // parameter names and in-code defaults of the node mains
  const double frequency = pnh.param("frequency", 10.0);
  const double val_dt = pnh.param("val_dt", 0.1), val_horizon = pnh.param("val_horizon", 0.5);
  const double max_vel_x = pnh.param("max_vel_x", 1.0), min_vel_x = pnh.param("min_vel_x", -1.0);
  const double max_vel_y = is_cart ? 0.0 : pnh.param("max_vel_y", 1.0);
  const double min_vel_y = is_cart ? 0.0 : pnh.param("min_vel_y", -1.0);
  const double max_rot_vel = pnh.param("max_rot_vel", 1.0), min_rot_vel = pnh.param("min_rot_vel", -1.0);
  const ee::vec umin = { min_vel_x, min_vel_y, min_rot_vel };
  const ee::vec umax = { max_vel_x, max_vel_y, max_rot_vel };

  const ee::Collision collision(pnh.param("boundary_radius", 0.7), pnh.param("search_radius", 1.0),
                                pnh.param("obstacle_threshold", 0.2), pnh.param("occupied_threshold", 0.8));

  const double ec_dt = pnh.param("ec_dt", 0.1), ec_horizon = pnh.param("ec_horizon", 2.0);
  const double target_resolution = pnh.param("target_resolution", 0.1);
  const double expl_weight = pnh.param("expl_weight", 1.0);
  const unsigned int num_basis = static_cast<unsigned int>(pnh.param("num_basis", 10.0));
  const unsigned int buffer_size = static_cast<unsigned int>(pnh.param("buffer_size", 1e6));
  const unsigned int batch_size = static_cast<unsigned int>(pnh.param("batch_size", 100.0));

  ee::mat Rinv(3, 3);
  if (is_cart) {
    const auto w = pnh.numbers("control_weights", { 1.0, 1.0 });
    Rinv(0, 0) = 1.0 / w.at(0);
    Rinv(1, 1) = 0.0;  // the lateral velocity is not a control of the cart
    Rinv(2, 2) = 1.0 / w.at(1);
  } else {
    const auto w = pnh.numbers("control_weights", { 1.0, 1.0, 1.0 });
    Rinv(0, 0) = 1.0 / w.at(0);
    Rinv(1, 1) = 1.0 / w.at(1);
    Rinv(2, 2) = 1.0 / w.at(2);
  }

  const auto means = pnh.numbers("means", { 2.5, 2.5, 8.5, 2.5 });
  const auto sigmas = pnh.numbers("sigmas", { 1.5, 1.5, 1.5, 1.5 });
  ee::GaussianList gaussians;
  for (std::size_t i = 0; i + 1 < means.size(); i += 2) {
    gaussians.emplace_back(ee::vec{ means[i], means[i + 1] }, ee::vec{ sigmas.at(i), sigmas.at(i + 1) });
  }
  const ee::Target target(gaussians);

  // dynamic window parameters (node mains :175-190)
  const ee::DynamicWindow dwa(collision, pnh.param("dwa_dt", 0.1), pnh.param("dwa_horizon", 1.0),
                              pnh.param("acc_dt", 0.2), pnh.param("acc_lim_x", 1.0),
                              is_cart ? 0.0 : pnh.param("acc_lim_y", 1.0), pnh.param("acc_lim_th", 1.0), max_vel_x,
                              min_vel_x, max_vel_y, min_vel_y, max_rot_vel, min_rot_vel,
                              static_cast<unsigned int>(pnh.param("vx_samples", 3.0)),
                              is_cart ? 1u : static_cast<unsigned int>(pnh.param("vy_samples", 8.0)),
                              static_cast<unsigned int>(pnh.param("vth_samples", 5.0)));

  const ModelT model;
  ee::ErgodicControl<ModelT> ergodic_control(model, collision, ec_dt, ec_horizon, target_resolution, expl_weight,
                                             num_basis, buffer_size, batch_size, Rinv, umin, umax);
  ee::Exploration<ModelT> exploration(ergodic_control, collision, dwa);
  exploration.setTarget(target);

  // occupancy map standing in for the map topic: free space plus the requested obstacle blocks
  const unsigned int w = ee::axis_length(map_x0, map_x0 + map_w, map_res);
  const unsigned int h = ee::axis_length(map_y0, map_y0 + map_h, map_res);
  ee::GridData cells(static_cast<std::size_t>(w) * h, 0);
  for (const auto& o : obstacles) {
    for (unsigned int i = 0; i < h; ++i) {
      for (unsigned int j = 0; j < w; ++j) {
        const double cx = map_x0 + (j + 0.5) * map_res, cy = map_y0 + (i + 0.5) * map_res;
        if (cx >= o[0] && cx <= o[2] && cy >= o[1] && cy <= o[3]) cells[static_cast<std::size_t>(i) * w + j] = 100;
      }
    }
  }
  const ee::GridMap grid = ee::GridMap::fromOccupancyGrid(w, h, map_res, map_x0, map_y0, cells);

  std::printf("# %s: K=%u steps=%u dt=%g frequency=%g Hz map=[%g,%g]x[%g,%g]\n", is_cart ? "exploration_cart"
              : "exploration_omni", num_basis, ergodic_control.steps(), ec_dt, frequency, grid.xmin(), grid.xmax(),
              grid.ymin(), grid.ymax());
  static const char* const kSource[] = { "ergodic", "dwa-follow", "dwa-reference", "dwa-replan" };
  ee::vec vb = { 0.0, 0.0, 0.0 };  // odometry twist: the simulated robot executes the command exactly
  for (int t = 0; t < ticks; ++t) {
    const ee::vec u = exploration.tick(grid, pose, vb, val_dt, val_horizon);
    std::printf("tick %3d pose %.17g %.17g %.17g  cmd_vel %.17g %.17g %.17g  %s\n", t, pose(0), pose(1), pose(2),
                u(0), u(1), u(2), kSource[static_cast<int>(exploration.source())]);
    pose = ee::integrate_twist(pose, u, 1.0 / frequency);
    pose(2) = ee::normalize_angle_PI(pose(2));
    vb = u;
  }
  return 0;
}

// Shared body of the exploration_omni / exploration_cart entry points (the reference's
// src/exploration_{omni,cart}_node.cpp without ROS): read the parameter set of the reference's
// yaml files, wire Collision / ErgodicControl / Target the way the node mains do, then run the
// Exploration loop (exploration.hpp:197-292: addStateMemory -> control -> validate_control -> DWA
// fallback) against a simulated robot on a map with optional rectangular obstacles, or on a
// map_server map (--map-yaml, the format of the reference's maps/maze.yaml).  Prints one line per
// tick.  --record writes the loop's inputs and outputs as a replay log; --replay feeds a log (a
// recorded run, or poses / twists / maps logged on a robot) through the same state machine and
// compares the twists tick by tick (map_io.hpp describes both file formats).
#pragma once

#include <chrono>
#include <cmath>
#include <cstdio>
#include <array>
#include <cstring>
#include <string>
#include <vector>

#include <ergodic_exploration/exploration.hpp>

#include "map_io.hpp"
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
  std::string map_yaml, record_path, replay_path;
  bool dump_map = false;
  double replay_tol = 1e-9;
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
    } else if (a == "--map-yaml" && i + 1 < argc) {
      map_yaml = argv[++i];
    } else if (a == "--dump-map") {
      dump_map = true;
    } else if (a == "--record" && i + 1 < argc) {
      record_path = argv[++i];
    } else if (a == "--replay" && i + 1 < argc) {
      replay_path = argv[++i];
    } else if (a == "--replay-tol" && i + 1 < argc) {
      replay_tol = std::atof(argv[++i]);
    } else if (a == "--set" && i + 2 < argc) {
      pnh.set(argv[i + 1], argv[i + 2]);
      i += 2;
    } else if (a == "--help") {
      std::printf("usage: %s [--params file.yaml] [--ticks N] [--pose x y th] [--map x0 y0 w h res] "
                  "[--obstacle x0 y0 x1 y1]... [--map-yaml map.yaml] [--dump-map] [--record log] [--replay log] "
                  "[--replay-tol eps] [--set name value]\n", argv[0]);
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

  // occupancy map standing in for the map topic: a map_server map, or free space plus the
  // requested obstacle blocks
  map_io::Occupancy occ;
  if (!map_yaml.empty()) {
    occ = map_io::load_map_yaml(map_yaml);
  } else {
    occ.width = ee::axis_length(map_x0, map_x0 + map_w, map_res);
    occ.height = ee::axis_length(map_y0, map_y0 + map_h, map_res);
    occ.resolution = map_res;
    occ.origin_x = map_x0;
    occ.origin_y = map_y0;
    occ.data.assign(static_cast<std::size_t>(occ.width) * occ.height, 0);
  }
  for (const auto& o : obstacles) {
    for (unsigned int i = 0; i < occ.height; ++i) {
      for (unsigned int j = 0; j < occ.width; ++j) {
        const double cx = occ.origin_x + (j + 0.5) * occ.resolution, cy = occ.origin_y + (i + 0.5) * occ.resolution;
        if (cx >= o[0] && cx <= o[2] && cy >= o[1] && cy <= o[3]) occ.data[static_cast<std::size_t>(i) * occ.width + j] = 100;
      }
    }
  }
  ee::GridMap grid = occ.grid();

  if (dump_map) {
    // summary of the occupancy cells (no engine, no GPU needed): counts and an FNV-1a checksum
    std::size_t n_free = 0, n_occ = 0, n_unknown = 0;
    unsigned long long hash = 1469598103934665603ull;
    for (const int8_t v : occ.data) {
      n_free += v == 0;
      n_occ += v == 100;
      n_unknown += v == -1;
      hash = (hash ^ static_cast<unsigned char>(v)) * 1099511628211ull;
    }
    std::printf("map width %u height %u resolution %.17g origin %.17g %.17g free %zu occupied %zu unknown %zu "
                "fnv1a %llu\n", occ.width, occ.height, occ.resolution, occ.origin_x, occ.origin_y, n_free, n_occ,
                n_unknown, hash);
    std::printf("bounds x [%.17g, %.17g] y [%.17g, %.17g]\n", grid.xmin(), grid.xmax(), grid.ymin(), grid.ymax());
    return 0;
  }

  const ModelT model;
  ee::ErgodicControl<ModelT> ergodic_control(model, collision, ec_dt, ec_horizon, target_resolution, expl_weight,
                                             num_basis, buffer_size, batch_size, Rinv, umin, umax);
  ee::Exploration<ModelT> exploration(ergodic_control, collision, dwa);
  exploration.setTarget(target);

  std::printf("# %s: K=%u steps=%u dt=%g frequency=%g Hz map=[%g,%g]x[%g,%g]\n", is_cart ? "exploration_cart"
              : "exploration_omni", num_basis, ergodic_control.steps(), ec_dt, frequency, grid.xmin(), grid.xmax(),
              grid.ymin(), grid.ymax());
  static const char* const kSource[] = { "ergodic", "dwa-follow", "dwa-reference", "dwa-replan" };
  auto print_tick = [&](int t, const ee::vec& p, const ee::vec& u) {
    std::printf("tick %3d pose %.17g %.17g %.17g  cmd_vel %.17g %.17g %.17g  %s\n", t, p(0), p(1), p(2), u(0), u(1),
                u(2), kSource[static_cast<int>(exploration.source())]);
  };

  if (!replay_path.empty()) {
    // replay: maps, poses and body twists come from the log; the twists are compared with the logged ones
    const std::vector<map_io::Record> records = map_io::read_replay(replay_path);
    bool have_map = false;
    int n_ticks = 0, n_compared = 0, n_source_mismatch = 0;
    double worst = 0.0;
    for (const map_io::Record& r : records) {
      if (r.is_map) {
        grid = r.map.grid();
        have_map = true;
        continue;
      }
      if (!have_map) {
        std::fprintf(stderr, "replay: tick before the first map record; using the command-line map\n");
        have_map = true;
      }
      const ee::vec p = { r.tick.pose[0], r.tick.pose[1], r.tick.pose[2] };
      const ee::vec v = { r.tick.vb[0], r.tick.vb[1], r.tick.vb[2] };
      const ee::vec u = exploration.tick(grid, p, v, val_dt, val_horizon);
      print_tick(r.tick.t, p, u);
      ++n_ticks;
      if (r.tick.source != "?") {
        ++n_compared;
        for (int c = 0; c < 3; ++c) {
          const double d = std::fabs(u(c) - r.tick.u[c]);
          worst = d > worst ? d : worst;
        }
        n_source_mismatch += r.tick.source != kSource[static_cast<int>(exploration.source())];
      }
    }
    std::printf("# replay: %d ticks, %d compared, max |cmd_vel - logged| = %.3g, source mismatches = %d\n", n_ticks,
                n_compared, worst, n_source_mismatch);
    return (n_compared > 0 && (worst > replay_tol || n_source_mismatch > 0)) ? 1 : 0;
  }

  std::FILE* rec = nullptr;
  if (!record_path.empty()) {
    rec = std::fopen(record_path.c_str(), "w");
    if (rec == nullptr) {
      std::fprintf(stderr, "cannot open %s for writing\n", record_path.c_str());
      return 2;
    }
    std::fprintf(rec, "# ergodic exploration replay log v1 (%s)\n", is_cart ? "exploration_cart" : "exploration_omni");
    map_io::write_map(rec, occ);
  }
  ee::vec vb = { 0.0, 0.0, 0.0 };  // odometry twist: the simulated robot executes the command exactly
  const auto loop_t0 = std::chrono::steady_clock::now();
  double tick_us = 0.0;
  for (int t = 0; t < ticks; ++t) {
    const auto t0 = std::chrono::steady_clock::now();
    const ee::vec u = exploration.tick(grid, pose, vb, val_dt, val_horizon);
    const double this_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    tick_us += this_us;
    if (std::getenv("EEA_SLOW_TICKS") && this_us > 1000.0) std::fprintf(stderr, "slow tick %d: %.0f us (%s)\n", t, this_us, kSource[static_cast<int>(exploration.source())]);
    print_tick(t, pose, u);
    if (rec != nullptr) {
      map_io::Tick k;
      k.t = t;
      for (int c = 0; c < 3; ++c) {
        k.pose[c] = pose(c);
        k.vb[c] = vb(c);
        k.u[c] = u(c);
      }
      k.source = kSource[static_cast<int>(exploration.source())];
      map_io::write_tick(rec, k);
    }
    pose = ee::integrate_twist(pose, u, 1.0 / frequency);
    pose(2) = ee::normalize_angle_PI(pose(2));
    vb = u;
  }
  const double loop_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - loop_t0).count();
  std::printf("# loop: %d ticks, %.1f us per tick in Exploration::tick (control + validate_control + fallback "
              "planner), %.1f us with printing and the simulated robot\n", ticks, ticks > 0 ? tick_us / ticks : 0.0,
              ticks > 0 ? loop_us / ticks : 0.0);
  if (rec != nullptr) std::fclose(rec);
  return 0;
}

// Exploration<ModelT>: the exploration stack's control loop without ROS (reference
// include/ergodic_exploration/exploration.hpp:177-292).  One tick() is one iteration of the
// reference's while-loop: remember the pose, run the ergodic controller unless a dynamic-window
// twist is being followed, validate the twist against the occupancy grid, and fall back to the
// dynamic window planner on a predicted collision.  Pose, body twist and map, which the
// reference receives from tf / odom / map topics, are arguments.
#pragma once

#include <tuple>

#include <ergodic_exploration/dynamic_window.hpp>
#include <ergodic_exploration/ergodic_control.hpp>

namespace ergodic_exploration
{
template <class ModelT>
class Exploration
{
public:
  // which controller produced the twist of the last tick
  enum class Source
  {
    ergodic,        // ErgodicControl::control
    dwa_follow,     // twist of an earlier DWA solution is being followed
    dwa_reference,  // DWA tracking the ergodic trajectory after a predicted collision
    dwa_replan      // DWA re-run because the followed DWA twist now predicts a collision
  };

  Exploration(const ErgodicControl<ModelT>& ergodic_control, const Collision& collision, const DynamicWindow& dwa)
    : ergodic_control_(ergodic_control), collision_(collision), dwa_(dwa), u_(3), follow_dwa_(false), i_(0),
      source_(Source::ergodic)
  {
  }

  // Exploration::control's prologue: the target distribution is set once
  void setTarget(const Target& target) { ergodic_control_.setTarget(target); }

  // one iteration of the control loop (exploration.hpp:197-292); returns the commanded body twist
  vec tick(const GridMap& grid, const vec& pose, const vec& vb, double val_dt, double val_horizon)
  {
    ergodic_control_.addStateMemory(pose);  // every tick, also while following DWA (:209)

    if (follow_dwa_) {
      i_++;
      follow_dwa_ = (i_ != dwa_.steps());  // may need to replan (:225-226)
      source_ = Source::dwa_follow;
    }
    if (!follow_dwa_) {
      u_ = ergodic_control_.control(grid, pose);  // (:232)
      source_ = Source::ergodic;
    }
    if (!validate_control(collision_, grid, pose, u_, val_dt, val_horizon)) {
      if (follow_dwa_) {
        // collision caused by the previous DWA twist: track the twist itself (:243-251)
        u_ = std::get<1>(dwa_.control(grid, pose, vb, u_));
        follow_dwa_ = false;
        source_ = Source::dwa_replan;
      } else {
        // collision caused by the ergodic controller: track its trajectory (:254-277)
        const mat opt_traj = ergodic_control_.optTraj();
        const auto state = dwa_.control(grid, pose, vb, opt_traj, ergodic_control_.timeStep());
        u_ = std::get<1>(state);
        follow_dwa_ = std::get<0>(state);
        if (follow_dwa_) i_ = 0;
        source_ = Source::dwa_reference;
      }
    }
    return u_;
  }

  bool followingDwa() const { return follow_dwa_; }
  Source source() const { return source_; }
  ErgodicControl<ModelT>& ergodicControl() { return ergodic_control_; }

private:
  ErgodicControl<ModelT> ergodic_control_;
  Collision collision_;
  DynamicWindow dwa_;
  vec u_;
  bool follow_dwa_;
  unsigned int i_;
  Source source_;
};
}  // namespace ergodic_exploration

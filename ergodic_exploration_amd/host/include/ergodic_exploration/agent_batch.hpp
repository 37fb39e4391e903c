// AgentBatch<ModelT>: N independent ErgodicControl<ModelT> agents of one rank, resident on one GPU, stepped with
// ONE eea_control_batch launch per receding-horizon step (or two half-batch launches on two HIP streams, whose
// launches overlap: agents are independent), plus the exchange steps of a multi-GPU agent batch.
//
// No counterpart in the single-agent reference: every agent is exactly one reference controller
// (include/ergodic_exploration/ergodic_control.hpp:72-185) with the same constructor parameters; the sharding
// (one process per GPU, contiguous agent blocks) and the c_k exchange follow the decentralised ergodic control
// the reference cites (README.md:225-227).  Everything goes through the C ABI (include/ergodic_amd.h); the
// RCCL communicator is the C ABI's eea_comm (no Python, no torch).
#pragma once

#include <memory>
#include <stdexcept>
#include <vector>

#include <ergodic_exploration/dynamic_window.hpp>
#include <ergodic_exploration/ergodic_control.hpp>

namespace ergodic_exploration
{
template <class ModelT>
class AgentBatch
{
public:
  // the constructor arguments of ErgodicControl (minus the host-side replay buffer sizes) + the agent count of
  // THIS rank; comm == nullptr: single GPU
  AgentBatch(unsigned int n_agents, double dt, double horizon, double resolution, double exploration_weight,
             unsigned int num_basis, const mat& Rinv, const vec& umin, const vec& umax, eea_comm* comm = nullptr,
             unsigned int groups = 2)
    : n_(n_agents), comm_(comm), groups_(groups < 1 ? 1 : (groups > 2 ? 2 : groups))
  {
    eea_config cfg{};
    cfg.model = device_model<ModelT>::value;
    cfg.precision = EEA_PREC_F64;
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
    throw_on_error(eea_create(&cfg, &e));
    engine_ = std::shared_ptr<eea_engine>(e, eea_destroy);
    steps_ = eea_steps(e);
    modes_ = eea_num_modes(e);
    hip_check(hipSetDevice(device_ordinal()));
    alloc(d_pose_, sizeof(double) * 3 * n_);
    alloc(d_ut_, sizeof(double) * 3 * steps_ * n_);
    alloc(d_u0_, sizeof(double) * 3 * n_);
    alloc(d_ck_, sizeof(double) * modes_ * n_);
    rec_len_ = eea_ck_record_len(e);
    alloc(d_agent_recs_, sizeof(double) * n_ * rec_len_);  // per-agent sum records of the current step
    alloc(d_recs_, sizeof(double) * 2 * rec_len_);         // [2 steps][sum record of all agents of all ranks]
    hip_check(hipMemset(d_ut_.get(), 0, sizeof(double) * 3 * steps_ * n_));  // ut_ starts at zero (:201)
    for (unsigned int g = 0; g < groups_; ++g) {
      hipStream_t s = nullptr;
      hip_check(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
      streams_.push_back(s);
      hipEvent_t ev = nullptr;
      hip_check(hipEventCreateWithFlags(&ev, hipEventDisableTiming | hipEventDisableSystemFence));  // orders streams only
      ev_group_.push_back(ev);
    }
    hip_check(hipEventCreateWithFlags(&ev_exchange_, hipEventDisableTiming | hipEventDisableSystemFence));
    if (comm_ == nullptr) {  // local communicator: the consensus is the mean over this rank's agents
      eea_comm* c = nullptr;
      throw_on_error(eea_comm_create(device_ordinal(), 1, 0, nullptr, &c));
      own_comm_ = std::shared_ptr<eea_comm>(c, eea_comm_destroy);
      comm_ = c;
    }
  }
  ~AgentBatch()
  {
    for (hipStream_t s : streams_) {
      (void)hipStreamSynchronize(s);
      (void)hipStreamDestroy(s);
    }
    for (hipEvent_t ev : ev_group_) (void)hipEventDestroy(ev);
    if (ev_exchange_) (void)hipEventDestroy(ev_exchange_);
  }
  AgentBatch(const AgentBatch&) = delete;
  AgentBatch& operator=(const AgentBatch&) = delete;

  void setTarget(const Target& target)
  {
    std::vector<double> mu, sg;
    target.flatten(mu, sg);
    throw_on_error(eea_set_target_gaussians(engine_.get(), static_cast<unsigned>(mu.size() / 2), mu.data(), sg.data()));
  }
  void configTarget(const GridMap& grid)
  {
    sync();
    throw_on_error(eea_config_domain(engine_.get(), grid.xmin(), grid.xmax(), grid.ymin(), grid.ymax(), nullptr,
                                     streams_[0]));
  }
  // poses: 3 x n_agents (column per agent, map frame)
  void setPoses(const mat& poses)
  {
    if (poses.n_rows() != 3 || poses.n_cols() != n_) throw std::invalid_argument("poses must be 3 x n_agents");
    sync();
    hip_check(hipMemcpy(d_pose_.get(), poses.memptr(), sizeof(double) * 3 * n_, hipMemcpyHostToDevice));
  }
  // One receding-horizon optimisation of every agent (ErgodicControl::control per agent).  consensus: the
  // gradient uses the mean c_k of ALL agents of all ranks from the previous step (decentralised ergodic control).
  // The control kernels leave sum records (eea_batch_io::d_ck_rec: one per agent, or -- short horizons, where several agents
  // share a wavefront -- one per wavefront, eea_batch_io::rec_per_wavefront: 4 - 8 x fewer bytes through the sum); ONE small launch adds them
  // (eea_ck_records_sum), ONE collective adds the ranks' records (nothing with one rank) and the next step's
  // kernels divide sum by count themselves (ck_shared_parts = 1).  Everything is stream-ordered: no host
  // synchronisation, and every stream that reads the record waits for the event behind the exchange.
  void control(bool consensus = false)
  {
    double* const rec_now = static_cast<double*>(d_recs_.get()) + static_cast<size_t>(step_ & 1u) * rec_len_;
    const double* const rec_prev = static_cast<const double*>(d_recs_.get()) + static_cast<size_t>((step_ + 1u) & 1u) * rec_len_;
    const bool shared = consensus && have_records_;
    unsigned int rec_first = 0;  // this group's first record (records per launch: eea_batch_record_count)
    for (unsigned int g = 0; g < groups_; ++g) {
      const unsigned int first = (n_ * g) / groups_, last = (n_ * (g + 1)) / groups_;
      if (last == first) continue;
      // the record this launch reads was completed (and all-reduced) on streams_[0]
      if (shared && g > 0) hip_check(hipStreamWaitEvent(streams_[g], ev_exchange_, 0));
      eea_batch_io io{};
      io.d_pose = static_cast<const double*>(d_pose_.get()) + 3 * first;
      io.d_ut = static_cast<double*>(d_ut_.get()) + static_cast<size_t>(3) * steps_ * first;
      io.d_u0 = static_cast<double*>(d_u0_.get()) + 3 * first;
      io.d_ck = static_cast<double*>(d_ck_.get()) + static_cast<size_t>(modes_) * first;
      if (shared) {
        io.d_ck_shared = rec_prev;
        io.ck_shared_parts = 1;
      }
      if (consensus) {
        io.d_ck_rec = static_cast<double*>(d_agent_recs_.get()) + static_cast<size_t>(rec_len_) * rec_first;
        io.rec_per_wavefront = 1;
        rec_first += eea_batch_record_count(engine_.get(), last - first);
      }
      throw_on_error(eea_control_batch(engine_.get(), last - first, &io, streams_[g]));
    }
    if (consensus) {
      // streams_[0] joins the other groups, then the sum and the exchange; the event orders the next step's readers
      for (unsigned int g = 1; g < groups_; ++g) {
        hip_check(hipEventRecord(ev_group_[g], streams_[g]));
        hip_check(hipStreamWaitEvent(streams_[0], ev_group_[g], 0));
      }
      throw_on_error(eea_ck_records_sum(engine_.get(), rec_first, d_agent_recs_.get(), rec_now, streams_[0]));
      throw_on_error(eea_comm_allreduce_sum(engine_.get(), comm_, rec_now, rec_len_, streams_[0]));
      hip_check(hipEventRecord(ev_exchange_, streams_[0]));
      have_records_ = true;
      ++step_;
    }
  }
  // mean c_k of all agents of all ranks after the last control(true): K^2 values (for inspection / tests)
  vec consensusTrajCoeff()
  {
    if (!have_records_) throw std::logic_error("no consensus step yet");
    sync();
    std::vector<double> h(rec_len_);
    const double* const rec = static_cast<const double*>(d_recs_.get()) + static_cast<size_t>((step_ + 1u) & 1u) * rec_len_;
    hip_check(hipMemcpy(h.data(), rec, sizeof(double) * h.size(), hipMemcpyDeviceToHost));
    vec c(modes_);
    for (unsigned int m = 0; m < modes_; ++m) c(m) = h[m] / h[modes_];
    return c;
  }
  // first twists of the updated control signals, 3 x n_agents
  mat controls()
  {
    sync();
    mat u(3, n_);
    hip_check(hipMemcpy(u.memptr(), d_u0_.get(), sizeof(double) * 3 * n_, hipMemcpyDeviceToHost));
    return u;
  }
  // all agents' c_k of all ranks, K^2 x (nranks * n_agents), rank order (equal shards): one ncclAllGather
  mat gatherTrajCoeff()
  {
    sync();
    const unsigned int world = static_cast<unsigned int>(eea_comm_nranks(comm_));
    Dev all;
    alloc(all, sizeof(double) * modes_ * n_ * world);
    throw_on_error(eea_comm_allgather_ck(engine_.get(), comm_, n_, d_ck_.get(), all.get(), streams_[0]));
    hip_check(hipStreamSynchronize(streams_[0]));
    mat ck(modes_, n_ * world);
    hip_check(hipMemcpy(ck.memptr(), all.get(), sizeof(double) * modes_ * n_ * world, hipMemcpyDeviceToHost));
    return ck;
  }
  // One iteration of Exploration<ModelT>::control's loop body (exploration.hpp:220-279) for EVERY agent of this rank on the
  // shared occupancy grid, on the device (eea_tick_batch): control() of the agents that follow no dynamic-window twist,
  // validate_control, the dynamic window per agent in its mode, the follow_dwa / i state machine.  The poses are those of
  // setPoses; vb: 3 x n body twists (odometry).  map_seq != 0: the caller vouches that (grid, map_seq) names one map content
  // (the inflated collision map is reused between ticks).  addStateMemory is the caller's (this class keeps no replay
  // memory).  Returns the commanded twists, 3 x n.
  mat tick(const GridMap& grid, const DynamicWindow& dwa, const mat& vb, double val_dt, double val_horizon,
           unsigned long long map_seq = 0)
  {
    if (vb.n_rows() != 3 || vb.n_cols() != n_) throw std::invalid_argument("vb must be 3 x n_agents");
    if (!d_follow_) {
      alloc(d_follow_, sizeof(int) * n_);
      alloc(d_count_, sizeof(unsigned) * n_);
      alloc(d_cmd_, sizeof(double) * 3 * n_);
      alloc(d_vb_, sizeof(double) * 3 * n_);
      alloc(d_traj_, sizeof(double) * 3 * steps_ * n_);
      alloc(d_valid_, sizeof(int) * n_);
      alloc(d_skip_, sizeof(int) * n_);
      alloc(d_source_, sizeof(int) * n_);
      hip_check(hipMemset(d_follow_.get(), 0, sizeof(int) * n_));
      hip_check(hipMemset(d_count_.get(), 0, sizeof(unsigned) * n_));
      hip_check(hipMemset(d_cmd_.get(), 0, sizeof(double) * 3 * n_));
    }
    sync();
    hip_check(hipMemcpy(d_vb_.get(), vb.memptr(), sizeof(double) * 3 * n_, hipMemcpyHostToDevice));
    throw_on_error(eea_config_domain(engine_.get(), grid.xmin(), grid.xmax(), grid.ymin(), grid.ymax(), nullptr, streams_[0]));
    eea_batch_io io{};
    io.d_pose = d_pose_.get();
    io.d_ut = d_ut_.get();
    eea_tick_io t{};
    t.d_follow_dwa = static_cast<int*>(d_follow_.get());
    t.d_dwa_count = static_cast<unsigned*>(d_count_.get());
    t.d_u = static_cast<double*>(d_cmd_.get());
    t.d_vb = static_cast<const double*>(d_vb_.get());
    t.d_grid = device_cells(grid);
    t.d_traj = static_cast<double*>(d_traj_.get());
    t.d_valid = static_cast<int*>(d_valid_.get());
    t.d_skip = static_cast<int*>(d_skip_.get());
    t.d_source = static_cast<int*>(d_source_.get());
    t.val_dt = val_dt;
    t.val_horizon = val_horizon;
    t.grid_epoch = map_seq;
    const eea_collision_cfg ccfg = dwa.collision().deviceConfig(grid);
    throw_on_error(eea_tick_batch(engine_.get(), n_, &io, &t, &ccfg, &dwa.deviceConfig(), streams_[0]));
    hip_check(hipStreamSynchronize(streams_[0]));
    mat u(3, n_);
    hip_check(hipMemcpy(u.memptr(), d_cmd_.get(), sizeof(double) * 3 * n_, hipMemcpyDeviceToHost));
    return u;
  }
  // who produced each agent's twist in the last tick: 0 control(), 1 a followed DWA twist, 2 DWA along optTraj(), 3 DWA
  // re-run towards the followed twist (Exploration::Source of the single-robot mirror, in that order)
  std::vector<int> tickSources() const
  {
    std::vector<int> s(n_, 0);
    if (d_source_) hip_check(hipMemcpy(s.data(), d_source_.get(), sizeof(int) * n_, hipMemcpyDeviceToHost));
    return s;
  }
  void sync()
  {
    for (hipStream_t s : streams_) hip_check(hipStreamSynchronize(s));
  }
  unsigned int agents() const { return n_; }
  unsigned int steps() const { return steps_; }
  eea_engine* engine() const { return engine_.get(); }

private:
  using Dev = std::shared_ptr<void>;
  static void alloc(Dev& d, size_t bytes)
  {
    void* p = nullptr;
    hip_check(hipMalloc(&p, bytes ? bytes : 1));
    d = Dev(p, [](void* q) { (void)hipFree(q); });
  }
  unsigned int n_, steps_ = 0, modes_ = 0;
  eea_comm* comm_;
  std::shared_ptr<eea_comm> own_comm_;
  unsigned int groups_;
  std::shared_ptr<eea_engine> engine_;
  Dev d_pose_, d_ut_, d_u0_, d_ck_, d_agent_recs_, d_recs_;
  Dev d_follow_, d_count_, d_cmd_, d_vb_, d_traj_, d_valid_, d_skip_, d_source_;  // tick(): per-agent loop state + scratch
  unsigned int rec_len_ = 0, step_ = 0;
  bool have_records_ = false;
  std::vector<hipStream_t> streams_;
  std::vector<hipEvent_t> ev_group_;
  hipEvent_t ev_exchange_ = nullptr;
};
}  // namespace ergodic_exploration

// What a pass costs when the host loop is C++ (what a maintainer's integration would be) instead of bench.py's Python:
// K = 10, T = 200, fp64, SimpleCart, two agent groups on two streams per rank, straight through the C ABI.
//   plain      two eea_control_batch calls per pass
//   consensus  the device-bound exchange at lag 1: two eea_control_batch calls (records out, ready marks, shared c_k of the
//              pass before in, in-kernel flag wait) + one eea_comm_records_exchange_bound per pass, nothing waited for
// Wall time per pass over `passes` passes after a warm-up, and the host's own share (time spent inside the calls).
//
// ranks > 1 (round 5): that many RANKS AS PROCESSES on the one GPU (this program re-executed once per rank: one process per
// rank, as in production), `agents` split evenly over them, each with its own engine, streams and eea_comm of one
// communicator -- the exchange then contains a real collective KERNEL per rank (record sum -> all-reduce -> publish -> flag).
// The collective library is given by path (eea_comm_set_library): on a one-GPU box the test double
// tests/fake_rccl/librccl.so.1 (stream-asynchronous kernels of a realistic footprint that meet on the device), because RCCL
// refuses two ranks of one communicator on one device.  This is the path the first real multi-GPU run takes, minus the xGMI
// hop.  (Ranks as THREADS of one process do not work with collectives that meet on the device: a HIP process maps its
// streams onto 4 hardware queues, and a collective kernel that spins for its peer in front of that peer's kernels in a shared
// queue is a dead-lock until the time-out -- tests/fake_rccl/selftest.cpp shows it.)
//
// With more than one rank ONE of a rank's two groups consumes the shared c_k stream-ordered (eea_comm_wait) instead of
// device-bound -- the rule of eea_comm_records_exchange_bound: the collective kernel needs execution slots, and when every
// slot is held by control wavefronts that wait for the flag it produces nothing frees one.  A NEGATIVE group count runs
// every group device-bound anyway (what times out, for the record).
//
// groups per rank: 1 / 2; negative = every group device-bound whatever the communicator; 12 (11) = two (one) groups with the
// STREAM-ORDERED exchange (eea_comm_records_exchange_async + eea_comm_wait): nothing waits inside a kernel; 22 (21) = the same
// protocol as ONE replayable device graph (eea_consensus_plan, ABI 6: 48 passes per launch): one runtime call per 48 passes;
// 32 (31) = the GATED exchange (ABI 6): the device-bound exchange with the flag wait as a one-wavefront gate kernel in front of
// each group's launch (eea_stream_wait_flag) instead of inside it -- launches only, safe with a collective kernel.
// usage: consensus_bench [passes = 4000] [agents = 4096] [ranks = 1] [collective library path] [lag = 1] [groups per rank = 2]
// last line of the output: RESULT {json}
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include "ergodic_amd.h"
#include "proc_ranks.hpp"

namespace
{
void ok(eea_status s, const char* what)
{
  if (s != EEA_OK) {
    std::fprintf(stderr, "%s: %s\n", what, eea_last_error());
    std::exit(1);
  }
}
void ok(hipError_t e, const char* what)
{
  if (e != hipSuccess) {
    std::fprintf(stderr, "%s: %s\n", what, hipGetErrorString(e));
    std::exit(1);
  }
}
template <typename T>
T* dev(size_t n)
{
  void* p = nullptr;
  ok(hipMalloc(&p, sizeof(T) * n), "hipMalloc");
  ok(hipMemset(p, 0, sizeof(T) * n), "hipMemset");
  return static_cast<T*>(p);
}
double now()
{
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

constexpr int NB = 6;  // rotating record / sum buffers (>= lag + 2)

struct Rank
{
  eea_engine* e = nullptr;
  eea_comm* c = nullptr;
  unsigned n = 0, T = 0, L = 0;
  double *d_pose = nullptr, *d_ut = nullptr, *d_u0 = nullptr;
  int* d_status = nullptr;
  unsigned *d_ready = nullptr, *d_flag = nullptr;
  double* d_arec[NB];
  double* d_sum[NB];
  hipStream_t streams[2];
  int groups = 2;
  bool last_group_stream_ordered = false;  // more than one rank: the collective kernel needs a group's slots to land in
  bool stream_ordered = false;             // every group consumes behind the exchange's event; nothing waits on the device
  bool gated = false;                      // device-bound records / sum / flag, the flag wait as a gate kernel in front of the launch
  unsigned* d_gate_timeouts = nullptr;
  unsigned seq = 0;

  void setup(unsigned agents, unsigned first_agent, int nranks, int rank, const char* id, int n_groups)
  {
    ok(hipSetDevice(0), "hipSetDevice");
    n = agents;
    groups = n_groups;
    eea_config cfg{};
    // CONSENSUS_BENCH_HORIZON / CONSENSUS_BENCH_MODEL (round 6): other shapes than the metric's, e.g. the yaml's T = 50 (horizon 5,
    // omni) whose batches run on the several-agents-per-wavefront kernel
    const char* const hz = std::getenv("CONSENSUS_BENCH_HORIZON");
    const char* const md = std::getenv("CONSENSUS_BENCH_MODEL");
    const bool omni = md != nullptr && std::strcmp(md, "omni") == 0;
    cfg.model = omni ? EEA_MODEL_OMNI : EEA_MODEL_SIMPLE_CART;
    cfg.precision = EEA_PREC_F64;
    cfg.dt = 0.1;
    cfg.horizon = hz ? std::atof(hz) : 20.0;
    cfg.resolution = 0.1;
    cfg.expl_weight = 1.0;
    cfg.num_basis = 10;
    cfg.Rinv[0] = 1.0;
    cfg.Rinv[8] = 2.0;
    if (omni) {
      cfg.Rinv[4] = 1.0;
      cfg.umin[1] = -1.0;
      cfg.umax[1] = 1.0;
    }
    cfg.umin[0] = -1.0;
    cfg.umax[0] = 1.0;
    cfg.umin[2] = -2.0;
    cfg.umax[2] = 2.0;
    ok(eea_create(&cfg, &e), "eea_create");
    const double mu[4] = { 2.5, 2.5, 8.5, 2.5 }, sg[4] = { 1.5, 1.5, 1.5, 1.5 };
    ok(eea_set_target_gaussians(e, 2, mu, sg), "set_target");
    ok(eea_config_domain(e, -1.0, 11.0, -1.0, 5.0, nullptr, nullptr), "config_domain");
    T = eea_steps(e);
    L = eea_ck_record_len(e);
    std::vector<double> poses(3 * static_cast<size_t>(n));
    unsigned long long r = 12345 + 7919ull * first_agent;
    auto uni = [&]() {
      r = r * 6364136223846793005ULL + 1442695040888963407ULL;
      return static_cast<double>(r >> 11) / 9007199254740992.0;
    };
    for (unsigned a = 0; a < n; ++a) {
      poses[3 * a] = -0.5 + 11.0 * uni();
      poses[3 * a + 1] = -0.5 + 5.0 * uni();
      poses[3 * a + 2] = -3.14 + 6.28 * uni();
    }
    d_pose = dev<double>(3 * static_cast<size_t>(n));
    ok(hipMemcpy(d_pose, poses.data(), sizeof(double) * poses.size(), hipMemcpyHostToDevice), "copy poses");
    d_ut = dev<double>(3 * static_cast<size_t>(T) * n);
    d_u0 = dev<double>(3 * static_cast<size_t>(n));
    d_status = dev<int>(n);
    d_ready = dev<unsigned>(n);
    d_flag = dev<unsigned>(1);
    d_gate_timeouts = dev<unsigned>(1);
    for (int s = 0; s < NB; ++s) {
      d_arec[s] = dev<double>(static_cast<size_t>(L) * n);
      d_sum[s] = dev<double>(L);
    }
    for (int g = 0; g < groups; ++g) ok(hipStreamCreateWithFlags(&streams[g], hipStreamNonBlocking), "stream");
    // (id == nullptr: a local communicator without any collective; with an id also ONE rank gets a real communicator, whose
    // all-reduce is a kernel of the collective library)
    ok(eea_comm_create(0, nranks, rank, id, &c), "eea_comm_create");
  }

  // `count` passes enqueued back to back, then this rank's streams drained; returns seconds per pass
  unsigned n_records = 0;  // records per pass of the last run()
  double t_gate = 0.0, t_ctrl = 0.0, t_xchg = 0.0;  // host time inside the gate / control / exchange calls of the last run()
  double run(bool consensus, int lag, int count, double* host_share)
  {
    double in_calls = 0.0;
    t_gate = t_ctrl = t_xchg = 0.0;
    const unsigned gb[3] = { 0, groups == 2 ? n / 2 : n, n };
    // CONSENSUS_BENCH_WAVE_RECORDS (round 6): eea_batch_io::rec_per_wavefront -- where agents share a wavefront each group
    // writes eea_batch_record_count records (one per wavefront) and the sum / exchange take that many
    const bool wave_rec = std::getenv("CONSENSUS_BENCH_WAVE_RECORDS") != nullptr && std::atoi(std::getenv("CONSENSUS_BENCH_WAVE_RECORDS")) != 0;
    unsigned rfirst[3] = { gb[0], gb[1], gb[2] };
    if (wave_rec) {
      rfirst[0] = 0;
      for (int g = 0; g < groups; ++g) rfirst[g + 1] = rfirst[g] + eea_batch_record_count(e, gb[g + 1] - gb[g]);
    }
    const unsigned n_rec = rfirst[groups];
    n_records = n_rec;
    const double t0 = now();
    for (int i = 0; i < count; ++i) {
      ++seq;
      const int slot = static_cast<int>(seq % NB), src = static_cast<int>((seq - lag) % NB);
      const double h0 = now();
      for (int g = 0; g < groups; ++g) {
        const unsigned first = gb[g], cnt = gb[g + 1] - gb[g];
        eea_batch_io io{};
        io.d_pose = d_pose + 3 * first;
        io.d_ut = d_ut + static_cast<size_t>(3) * T * first;
        io.d_u0 = d_u0 + 3 * first;
        if (consensus && stream_ordered) {
          // the stream-ordered exchange (eea_comm_records_exchange_async + eea_comm_wait): no ready marks, no flag, no wait
          // inside a kernel -- a launch starts when the record it consumes is complete
          io.d_status = d_status + first;
          io.d_ck_rec = d_arec[slot] + static_cast<size_t>(L) * rfirst[g];
          io.rec_per_wavefront = wave_rec ? 1 : 0;
          if (i >= lag) {
            io.d_ck_shared = d_sum[src];
            io.ck_shared_parts = 1;
            ok(eea_comm_wait(c, src, streams[g]), "eea_comm_wait");
          }
        } else if (consensus) {
          io.d_status = d_status + first;
          io.d_ck_rec = d_arec[slot] + static_cast<size_t>(L) * rfirst[g];
          io.rec_per_wavefront = wave_rec ? 1 : 0;
          io.d_rec_ready = d_ready + rfirst[g];
          io.rec_seq = seq;
          if (i >= lag && gated) {
            io.d_ck_shared = d_sum[src];
            io.ck_shared_parts = 1;
            const double g0 = now();
            ok(eea_stream_wait_flag(d_flag, seq - static_cast<unsigned>(lag), d_gate_timeouts, streams[g]), "eea_stream_wait_flag");
            t_gate += now() - g0;
          } else if (i >= lag) {
            io.d_ck_shared = d_sum[src];
            io.ck_shared_parts = 1;
            io.d_ck_flag = d_flag;
            io.ck_flag_seq = seq - static_cast<unsigned>(lag);
            // (the rule of eea_comm_records_exchange_bound for more than one rank: one group consumes stream-ordered)
            if (last_group_stream_ordered && g == groups - 1) ok(eea_comm_wait(c, src, streams[g]), "eea_comm_wait");
          }
        }
        const double c0 = now();
        ok(eea_control_batch(e, cnt, &io, streams[g]), "eea_control_batch");
        t_ctrl += now() - c0;
      }
      const double x0 = now();
      if (consensus && stream_ordered) {
        void* gs[2] = { streams[0], streams[1] };
        ok(eea_comm_records_exchange_async(e, c, n_rec, d_arec[slot], d_sum[slot], gs, static_cast<unsigned>(groups), slot), "exchange (stream-ordered)");
      } else if (consensus) {
        ok(eea_comm_records_exchange_bound(e, c, n_rec, d_arec[slot], d_ready, seq, d_sum[slot], d_flag, slot), "exchange");
      }
      t_xchg += now() - x0;
      in_calls += now() - h0;
    }
    for (int g = 0; g < groups; ++g) ok(hipStreamSynchronize(streams[g]), "sync");
    const double dt = now() - t0;
    if (host_share) *host_share = in_calls / count;
    return dt / count;
  }

  // the stream-ordered protocol as a replayable device graph (eea_consensus_plan): `count` passes in launches of `per`
  eea_consensus_plan* plan = nullptr;
  unsigned plan_passes = 0;
  double run_plan(int lag, int per, int count, double* host_share)
  {
    if (plan == nullptr) {
      const unsigned gb[3] = { 0, groups == 2 ? n / 2 : n, n };
      unsigned cnt[2];
      eea_batch_io io[2] = {};
      for (int g = 0; g < groups; ++g) {
        cnt[g] = gb[g + 1] - gb[g];
        io[g].d_pose = d_pose + 3 * gb[g];
        io[g].d_ut = d_ut + static_cast<size_t>(3) * T * gb[g];
        io[g].d_u0 = d_u0 + 3 * gb[g];
        io[g].d_status = d_status + gb[g];
      }
      eea_consensus_desc d{};
      d.n_groups = static_cast<unsigned>(groups);
      d.group_agents = cnt;
      d.group_io = io;
      d.lag = static_cast<unsigned>(lag);
      d.passes_per_launch = static_cast<unsigned>(per);
      ok(eea_consensus_plan_create(e, c, &d, &plan), "eea_consensus_plan_create");
      ok(eea_consensus_plan_info(plan, &plan_passes, nullptr), "eea_consensus_plan_info");
    }
    const int launches = (count + static_cast<int>(plan_passes) - 1) / static_cast<int>(plan_passes);
    double in_calls = 0.0;
    const double t0 = now();
    for (int i = 0; i < launches; ++i) {
      const double h0 = now();
      ok(eea_consensus_plan_launch(plan, streams[0]), "eea_consensus_plan_launch");
      in_calls += now() - h0;
    }
    ok(hipStreamSynchronize(streams[0]), "sync");
    const double dt = now() - t0;
    const double passes = static_cast<double>(launches) * plan_passes;
    if (host_share) *host_share = in_calls / passes;
    return dt / passes;
  }

  int timed_out()
  {
    // (a time-out stays in d_status under the device-bound exchange until the caller clears it, ergodic_amd.h)
    ok(hipDeviceSynchronize(), "sync");
    std::vector<int> st(n);
    ok(hipMemcpy(st.data(), d_status, sizeof(int) * n, hipMemcpyDeviceToHost), "status");
    int bad = 0;
    for (int v : st) bad += v != 0;
    unsigned gate_timeouts = 0;
    ok(hipMemcpy(&gate_timeouts, d_gate_timeouts, sizeof(unsigned), hipMemcpyDeviceToHost), "gate time-outs");
    return bad + static_cast<int>(gate_timeouts);
  }
};

// the plain passes of one rank as ONE hipGraph of `per` passes (2 per kernel nodes on two captured streams), replayed: does
// a graph remove anything from the launch boundary that a stream of launches leaves?
double run_graph(Rank& rk, int per, int count)
{
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  hipEvent_t fork = nullptr, join = nullptr;
  const unsigned gb[3] = { 0, rk.n / 2, rk.n };
  ok(hipEventCreateWithFlags(&fork, hipEventDisableTiming), "event");
  ok(hipEventCreateWithFlags(&join, hipEventDisableTiming), "event");
  ok(hipStreamBeginCapture(rk.streams[0], hipStreamCaptureModeThreadLocal), "begin capture");
  ok(hipEventRecord(fork, rk.streams[0]), "fork");
  ok(hipStreamWaitEvent(rk.streams[1], fork, 0), "fork wait");
  for (int i = 0; i < per; ++i) {
    for (int g = 0; g < 2; ++g) {
      const unsigned first = gb[g], cnt = gb[g + 1] - gb[g];
      eea_batch_io io{};
      io.d_pose = rk.d_pose + 3 * first;
      io.d_ut = rk.d_ut + static_cast<size_t>(3) * rk.T * first;
      io.d_u0 = rk.d_u0 + 3 * first;
      ok(eea_control_batch(rk.e, cnt, &io, rk.streams[g]), "eea_control_batch (captured)");
    }
  }
  ok(hipEventRecord(join, rk.streams[1]), "join");
  ok(hipStreamWaitEvent(rk.streams[0], join, 0), "join wait");
  ok(hipStreamEndCapture(rk.streams[0], &graph), "end capture");
  ok(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0), "instantiate");
  const int launches = count / per;
  for (int i = 0; i < 4; ++i) ok(hipGraphLaunch(exec, rk.streams[0]), "graph launch");
  ok(hipDeviceSynchronize(), "sync");
  const double t0 = now();
  for (int i = 0; i < launches; ++i) ok(hipGraphLaunch(exec, rk.streams[0]), "graph launch");
  ok(hipDeviceSynchronize(), "sync");
  const double dt = (now() - t0) / (static_cast<double>(launches) * per);
  (void)hipGraphExecDestroy(exec);
  (void)hipGraphDestroy(graph);
  (void)hipEventDestroy(fork);
  (void)hipEventDestroy(join);
  return dt;
}
}  // namespace

int main(int argc, char** argv)
{
  const int passes = argc > 1 ? std::atoi(argv[1]) : 4000;
  const unsigned agents = argc > 2 ? static_cast<unsigned>(std::atoi(argv[2])) : 4096;
  const int nranks = argc > 3 ? std::atoi(argv[3]) : 1;
  const std::string lib = argc > 4 ? argv[4] : "";
  const int lag = argc > 5 ? std::atoi(argv[5]) : 1;
  // groups: 1 / 2 = that many agent groups; negative: every group device-bound also with a collective; 11 / 12: 1 / 2 groups with
  // the STREAM-ORDERED exchange (nothing waits on the device)
  const int garg = argc > 6 ? std::atoi(argv[6]) : 2;
  const int groups = std::abs(garg) % 10;
  const bool all_bound = garg < 0, as_gated = garg >= 30, stream_ordered = garg >= 10 && !as_gated, as_plan = garg >= 20 && !as_gated;
  const bool child = argc > 9 && std::strcmp(argv[7], "child") == 0;
  if (nranks < 1 || nranks > 8 || lag < 1 || lag + 2 > NB || passes < 10 || groups < 1 || groups > 2 || (nranks > 1 && lib.empty() && false)) {
    std::fprintf(stderr, "usage: consensus_bench [passes] [agents] [ranks 1..8] [collective library] [lag 1..%d] [groups 1..2]\n", NB - 2);
    return 1;
  }
  if (nranks > 1 && !child) {  // parent: one process per rank; nothing here touches the GPU
    std::vector<std::string> args;
    for (int i = 1; i <= 6; ++i) {
      args.push_back(i < argc ? argv[i] : (i == 4 ? "" : (i == 5 ? "1" : "2")));
    }
    return proc_ranks::spawn(argv[0], args, nranks);
  }
  const int rank = child ? std::atoi(argv[8]) : 0;
  const std::string base = child ? argv[9] : "";
  proc_ranks::Shared* sh = child ? proc_ranks::attach("/" + base) : nullptr;
  if (child && sh == nullptr) {
    std::fprintf(stderr, "shared segment\n");
    return 1;
  }
  auto barrier = [&]() {
    if (sh) proc_ranks::barrier(sh, nranks);
  };
  if (!lib.empty()) ok(eea_comm_set_library(lib.c_str()), "eea_comm_set_library");
  ok(hipSetDevice(0), "hipSetDevice");
  char id[EEA_COMM_ID_BYTES] = {};
  const bool with_collective = nranks > 1 || !lib.empty();  // one rank + a library: the collective KERNEL is still in the exchange
  if (nranks == 1 && with_collective) ok(eea_comm_get_unique_id(id), "eea_comm_get_unique_id");
  if (nranks > 1) {
    const std::string idfile = "/tmp/" + base + ".0";
    if (rank == 0) {
      ok(eea_comm_get_unique_id(id), "eea_comm_get_unique_id");
      proc_ranks::publish_id(idfile, id, sizeof(id));
    } else if (!proc_ranks::fetch_id(idfile, id, sizeof(id))) {
      std::fprintf(stderr, "no communicator id\n");
      return 1;
    }
  }
  Rank rk;
  const unsigned first = agents * rank / nranks, n = agents * (rank + 1) / nranks - first;
  rk.setup(n, first, nranks, rank, with_collective ? id : nullptr, groups);
  rk.last_group_stream_ordered = with_collective && !all_bound && !stream_ordered && !as_gated;
  rk.gated = as_gated;
  rk.stream_ordered = stream_ordered;
  double host_plain = 0.0, host_cons = 0.0;
  barrier();
  rk.run(false, lag, 1000, nullptr);  // clock spin-up, warm start
  barrier();
  const double plain = rk.run(false, lag, passes, &host_plain);
  barrier();
  const int plan_per = std::getenv("CONSENSUS_PLAN_PASSES") ? std::atoi(std::getenv("CONSENSUS_PLAN_PASSES")) : 48;
  if (as_plan) rk.run_plan(lag, plan_per, 200, nullptr);
  else rk.run(true, lag, passes < 200 ? passes : 200, nullptr);
  barrier();
  double cons = 0.0;
  if (as_plan) {
    cons = rk.run_plan(lag, plan_per, passes, &host_cons);
  } else if (const char* ch = std::getenv("CONSENSUS_BENCH_CHUNK")) {
    // diagnosis: the timed run in chunks (each drained), with the chunk's time and the agents that have timed out so far
    const int chunk = std::atoi(ch);
    for (int done = 0; done < passes; done += chunk) {
      const double t = rk.run(true, lag, chunk, &host_cons);
      cons += t * chunk / passes;
      std::printf("  chunk at pass %5d: %9.2f us per pass, agents timed out so far %d\n", done, 1e6 * t, rk.timed_out());
    }
  } else {
    cons = rk.run(true, lag, passes, &host_cons);
  }
  barrier();
  if (!as_plan && std::getenv("CONSENSUS_BENCH_BREAKDOWN")) {
    std::printf("  host time per consensus pass: gates %.2f us, control calls %.2f us, exchange call %.2f us\n", 1e6 * rk.t_gate / passes,
                1e6 * rk.t_ctrl / passes, 1e6 * rk.t_xchg / passes);
  }
  const int bad = rk.timed_out();
  // the test double counts the blocks of its collective kernels that gave up waiting for another rank
  int collective_errors = -1;
  if (!lib.empty()) {
    if (void* h = dlopen(lib.c_str(), RTLD_NOW | RTLD_NOLOAD)) {
      if (auto fn = reinterpret_cast<int (*)()>(dlsym(h, "fake_rccl_errors"))) collective_errors = fn();
    }
  }
  double p = plain, c = cons, hp = host_plain, hc = host_cons;
  int timed_out = bad;
  if (sh) {  // the slowest rank counts
    sh->vals[rank][0] = plain;
    sh->vals[rank][1] = cons;
    sh->vals[rank][2] = host_plain;
    sh->vals[rank][3] = host_cons;
    sh->ints[rank][0] = bad;
    barrier();
    timed_out = 0;
    for (int r = 0; r < nranks; ++r) {
      p = std::max(p, sh->vals[r][0]);
      c = std::max(c, sh->vals[r][1]);
      hp = std::max(hp, sh->vals[r][2]);
      hc = std::max(hc, sh->vals[r][3]);
      timed_out += sh->ints[r][0];
    }
    barrier();
  }
  const int rc = (timed_out == 0 && collective_errors <= 0) ? 0 : 2;
  if (rank != 0) {
    if (rk.plan) eea_consensus_plan_destroy(rk.plan);
    eea_comm_destroy(rk.c);
    eea_destroy(rk.e);
    return rc;
  }
  std::printf("C++ host loop, %u agents on the GPU in %d rank(s), K = 10, T = %u, fp64, %u lanes per agent, %d agent group(s) per rank, one launch per pass and group, %d passes:\n",
              agents, nranks, rk.T, eea_batch_agent_lanes(rk.e, groups == 2 ? rk.n / 2 : rk.n), groups, passes);
  std::printf("  plain passes                         %6.2f us per pass   (host inside the calls: %5.2f us per pass)\n", 1e6 * p, 1e6 * hp);
  double graphed = 0.0;
  if (nranks == 1 && groups == 2 && passes >= 200) {
    graphed = run_graph(rk, 50, passes);
    std::printf("  plain passes as a hipGraph of 50      %6.2f us per pass   (2 x 50 kernel nodes on two captured streams, replayed %d times)\n",
                1e6 * graphed, passes / 50);
  }
  std::printf("  consensus every pass, lag %d (%s)  %6.2f us per pass   (host inside the calls: %5.2f us per pass)   = %.3f x plain; agents timed out: %d%s\n",
              lag, as_gated ? "gated" : (as_plan ? "stream-ordered, one graph of 48 passes" : (stream_ordered ? "stream-ordered" : "bound")), 1e6 * c, 1e6 * hc, c / p, timed_out,
              with_collective ? (collective_errors == 0 ? "; collective kernels: none gave up" : "; collective kernels gave up or count unavailable") : "");
  std::printf("  records through the sum per pass: %u (%s)\n", rk.n_records, rk.n_records < agents ? "one per wavefront" : "one per agent");
  std::printf("RESULT {\"agents\": %u, \"ranks\": %d, \"collective_kernel_in_exchange\": %s, \"consuming_groups\": \"%s\", \"groups_per_rank\": %d, \"lag\": %d, \"passes\": %d, \"plain_us_per_pass\": %.3f, \"consensus_us_per_pass\": %.3f, "
              "\"ratio\": %.4f, \"host_us_per_pass_plain\": %.3f, \"host_us_per_pass_consensus\": %.3f, \"agents_timed_out\": %d, "
              "\"collective_kernel_timeouts\": %d, \"graph_us_per_pass\": %.3f, \"horizon_steps\": %u, \"lanes_per_agent\": %u, \"records_per_pass\": %u}\n",
              agents, nranks, with_collective ? "true" : "false", as_gated ? "all gated (eea_stream_wait_flag in front of every consuming launch)"
              : as_plan ? "all stream-ordered, replayed as one device graph (eea_consensus_plan)"
              : rk.stream_ordered ? "all stream-ordered (eea_comm_records_exchange_async + eea_comm_wait)"
                                                         : (rk.last_group_stream_ordered ? "one device-bound, one stream-ordered" : "all device-bound"),
              groups, lag, passes, 1e6 * p, 1e6 * c, c / p, 1e6 * hp, 1e6 * hc, timed_out, collective_errors, 1e6 * graphed, rk.T,
              eea_batch_agent_lanes(rk.e, groups == 2 ? rk.n / 2 : rk.n), rk.n_records);
  if (rk.plan) eea_consensus_plan_destroy(rk.plan);
  eea_comm_destroy(rk.c);
  eea_destroy(rk.e);
  return rc;
}

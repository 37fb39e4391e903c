// What a pass costs when the host loop is C++ (what a maintainer's integration would be) instead of bench.py's Python:
// 4096 agents, K = 10, T = 200, fp64, SimpleCart, two agent groups on two streams, straight through the C ABI.
//   plain      two eea_control_batch calls per pass
//   consensus  the device-bound exchange at lag 1: two eea_control_batch calls (records out, ready marks, shared c_k of the
//              pass before in, in-kernel flag wait) + one eea_comm_records_exchange_bound per pass, nothing waited for
// Wall time per pass over `passes` passes after a warm-up, and the host's own share (time spent inside the calls).
// usage: consensus_bench [passes = 4000] [agents = 4096]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include <hip/hip_runtime.h>

#include "ergodic_amd.h"

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
}  // namespace

int main(int argc, char** argv)
{
  const int passes = argc > 1 ? std::atoi(argv[1]) : 4000;
  const unsigned n = argc > 2 ? static_cast<unsigned>(std::atoi(argv[2])) : 4096;
  eea_config cfg{};
  cfg.model = EEA_MODEL_SIMPLE_CART;
  cfg.precision = EEA_PREC_F64;
  cfg.dt = 0.1;
  cfg.horizon = 20.0;
  cfg.resolution = 0.1;
  cfg.expl_weight = 1.0;
  cfg.num_basis = 10;
  cfg.Rinv[0] = 1.0;
  cfg.Rinv[8] = 2.0;
  cfg.umin[0] = -1.0;
  cfg.umax[0] = 1.0;
  cfg.umin[2] = -2.0;
  cfg.umax[2] = 2.0;
  eea_engine* e = nullptr;
  ok(eea_create(&cfg, &e), "eea_create");
  const double mu[4] = { 2.5, 2.5, 8.5, 2.5 }, sg[4] = { 1.5, 1.5, 1.5, 1.5 };
  ok(eea_set_target_gaussians(e, 2, mu, sg), "set_target");
  ok(eea_config_domain(e, -1.0, 11.0, -1.0, 5.0, nullptr, nullptr), "config_domain");
  const unsigned T = eea_steps(e), L = eea_ck_record_len(e);
  std::vector<double> poses(3 * static_cast<size_t>(n));
  unsigned long long r = 12345;
  auto uni = [&]() {
    r = r * 6364136223846793005ULL + 1442695040888963407ULL;
    return static_cast<double>(r >> 11) / 9007199254740992.0;
  };
  for (unsigned a = 0; a < n; ++a) {
    poses[3 * a] = -0.5 + 11.0 * uni();
    poses[3 * a + 1] = -0.5 + 5.0 * uni();
    poses[3 * a + 2] = -3.14 + 6.28 * uni();
  }
  double* const d_pose = dev<double>(3 * static_cast<size_t>(n));
  ok(hipMemcpy(d_pose, poses.data(), sizeof(double) * poses.size(), hipMemcpyHostToDevice), "copy poses");
  double* const d_ut = dev<double>(3 * static_cast<size_t>(T) * n);
  double* const d_u0 = dev<double>(3 * static_cast<size_t>(n));
  int* const d_status = dev<int>(n);
  unsigned* const d_ready = dev<unsigned>(n);
  unsigned* const d_flag = dev<unsigned>(1);
  constexpr int NB = 4;
  double* d_arec[NB];
  double* d_sum[NB];
  for (int s = 0; s < NB; ++s) {
    d_arec[s] = dev<double>(static_cast<size_t>(L) * n);
    d_sum[s] = dev<double>(L);
  }
  hipStream_t streams[2];
  for (hipStream_t& s : streams) ok(hipStreamCreateWithFlags(&s, hipStreamNonBlocking), "stream");
  eea_comm* c = nullptr;
  ok(eea_comm_create(0, 1, 0, nullptr, &c), "eea_comm_create");
  const unsigned gb[3] = { 0, n / 2, n };
  unsigned seq = 0;

  auto run = [&](bool consensus, int count, double* host_share) {
    double in_calls = 0.0;
    const double t0 = now();
    for (int i = 0; i < count; ++i) {
      ++seq;
      const int slot = static_cast<int>(seq % NB), src = static_cast<int>((seq - 1) % NB);
      const double h0 = now();
      for (int g = 0; g < 2; ++g) {
        const unsigned first = gb[g], cnt = gb[g + 1] - gb[g];
        eea_batch_io io{};
        io.d_pose = d_pose + 3 * first;
        io.d_ut = d_ut + static_cast<size_t>(3) * T * first;
        io.d_u0 = d_u0 + 3 * first;
        if (consensus) {
          io.d_status = d_status + first;
          io.d_ck_rec = d_arec[slot] + static_cast<size_t>(L) * first;
          io.d_rec_ready = d_ready + first;
          io.rec_seq = seq;
          if (i >= 1) {
            io.d_ck_shared = d_sum[src];
            io.ck_shared_parts = 1;
            io.d_ck_flag = d_flag;
            io.ck_flag_seq = seq - 1;
          }
        }
        ok(eea_control_batch(e, cnt, &io, streams[g]), "eea_control_batch");
      }
      if (consensus) ok(eea_comm_records_exchange_bound(e, c, n, d_arec[slot], d_ready, seq, d_sum[slot], d_flag, slot), "exchange");
      in_calls += now() - h0;
    }
    ok(hipDeviceSynchronize(), "sync");
    const double dt = now() - t0;
    if (host_share) *host_share = in_calls / count;
    return dt / count;
  };

  // the plain passes again as ONE hipGraph of `per` passes (2 per kernel nodes on two captured streams), replayed: does a
  // graph remove anything from the launch boundary that a stream of launches leaves?
  auto run_graph = [&](int per, int count) {
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    hipEvent_t fork = nullptr, join = nullptr;
    ok(hipEventCreateWithFlags(&fork, hipEventDisableTiming), "event");
    ok(hipEventCreateWithFlags(&join, hipEventDisableTiming), "event");
    ok(hipStreamBeginCapture(streams[0], hipStreamCaptureModeThreadLocal), "begin capture");
    ok(hipEventRecord(fork, streams[0]), "fork");
    ok(hipStreamWaitEvent(streams[1], fork, 0), "fork wait");
    for (int i = 0; i < per; ++i) {
      for (int g = 0; g < 2; ++g) {
        const unsigned first = gb[g], cnt = gb[g + 1] - gb[g];
        eea_batch_io io{};
        io.d_pose = d_pose + 3 * first;
        io.d_ut = d_ut + static_cast<size_t>(3) * T * first;
        io.d_u0 = d_u0 + 3 * first;
        ok(eea_control_batch(e, cnt, &io, streams[g]), "eea_control_batch (captured)");
      }
    }
    ok(hipEventRecord(join, streams[1]), "join");
    ok(hipStreamWaitEvent(streams[0], join, 0), "join wait");
    ok(hipStreamEndCapture(streams[0], &graph), "end capture");
    ok(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0), "instantiate");
    const int launches = count / per;
    for (int i = 0; i < 4; ++i) ok(hipGraphLaunch(exec, streams[0]), "graph launch");
    ok(hipDeviceSynchronize(), "sync");
    const double t0 = now();
    for (int i = 0; i < launches; ++i) ok(hipGraphLaunch(exec, streams[0]), "graph launch");
    ok(hipDeviceSynchronize(), "sync");
    const double dt = (now() - t0) / (static_cast<double>(launches) * per);
    (void)hipGraphExecDestroy(exec);
    (void)hipGraphDestroy(graph);
    (void)hipEventDestroy(fork);
    (void)hipEventDestroy(join);
    return dt;
  };

  run(false, 1000, nullptr);  // clock spin-up, warm start
  double host_plain = 0.0, host_cons = 0.0;
  const double plain = run(false, passes, &host_plain);
  run(true, 200, nullptr);
  const double cons = run(true, passes, &host_cons);
  std::vector<int> st(n);
  ok(hipMemcpy(st.data(), d_status, sizeof(int) * n, hipMemcpyDeviceToHost), "status");
  int bad = 0;
  for (int v : st) bad += v != 0;
  std::printf("C++ host loop, %u agents, K = 10, T = %u, fp64, two agent groups, one launch per pass and group, %d passes:\n", n, T, passes);
  std::printf("  plain passes                         %6.2f us per pass   (host inside the calls: %5.2f us per pass)\n", 1e6 * plain, 1e6 * host_plain);
  const double graphed = run_graph(50, passes);
  std::printf("  plain passes as a hipGraph of 50      %6.2f us per pass   (2 x 50 kernel nodes on two captured streams, replayed %d times)\n",
              1e6 * graphed, passes / 50);
  std::printf("  consensus every pass, lag 1 (bound)  %6.2f us per pass   (host inside the calls: %5.2f us per pass)   = %.3f x plain; agents timed out: %d\n",
              1e6 * cons, 1e6 * host_cons, cons / plain, bad);
  eea_comm_destroy(c);
  eea_destroy(e);
  return bad == 0 ? 0 : 2;
}

// Multi-rank exchange paths of the engine with TWO ranks on ONE GPU: the ranks are two PROCESSES (this program re-executed
// once per rank by a parent that never touches the GPU -- one process per rank, as in production), each with its own engine,
// streams and eea_comm (eea_comm_create(nranks = 2)); the collective library behind the communicators is the test double
// tests/fake_rccl/librccl.so.1 (round 5: stream-asynchronous collective KERNELS of a realistic footprint that meet on the
// device -- no host wait anywhere), which csrc/comm.hip binds at run time because the harness puts its directory in front of
// LD_LIBRARY_PATH -- no PyTorch, no real RCCL in the process (RCCL itself refuses two ranks of one communicator on one
// device).  Until round 5 the ranks were threads of one process and the double a host rendezvous; with collectives that
// meet on the device, threads do not work (a HIP process maps its streams onto 4 hardware queues: a collective kernel that
// spins for its peer in front of that peer's kernels in a shared queue is a dead-lock) and are not what production runs.
//
// Every rank computes the single-rank reference itself (one engine holding all agents) and checks its own shard against
// it; what both ranks must hold bitwise the same (the all-reduced record, the gathered c_k, phi_k) leaves each rank as a
// line "XR <label> <hash>" that the parent compares.
//
// What runs here with more than one rank:
//   (1) eea_comm_create / rank / nranks;
//   (2) the all-gather of every agent's c_k (north_star's exchange): rank order, bitwise against one rank that holds
//       all agents (AgentBatch::gatherTrajCoeff);
//   (3) the stream-ordered consensus of AgentBatch::control(true) (record sum + all-reduce of the 816-byte record),
//       equal and ragged shards, several steps back to back, against one rank holding all agents: <= 1e-12;
//   (4) the DEVICE-BOUND exchange (eea_comm_records_exchange_bound: polling record sum -> all-reduce -> publish -> flag;
//       consumers wait inside their kernels) with two ranks, lag 1, five dependent passes: controls against one rank holding
//       all agents <= 1e-9 (the parity bar on controls), no agent timed out; and the same FREE-RUNNING (8 passes enqueued
//       without a host wait, one group device-bound and one stream-ordered per rank: the rule of
//       eea_comm_records_exchange_bound for more than one rank) at 2 x 512 agents.  (The full-GPU residency question -- does
//       the collective kernel land beside 4096 waiting control wavefronts -- is asked of ONE process with a collective kernel
//       in its exchange, host/test/consensus_bench.cpp: two processes on one GPU interfere in the scheduler, plain passes
//       take 2.5 x there, profiles/r05_two_ranks.txt);
//   (5) the grid-tiled occupancy target: rows split over the ranks, one all-reduce of K^2 sums, eea_set_phik_from_sums,
//       against the un-tiled eea_set_target_occupancy: <= 1e-12.
// Semantics source: decentralised ergodic control shares c_k between agents (reference README.md:225-227); every agent
// is one reference ErgodicControl (ergodic_control.hpp:224-311).
#include <dlfcn.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <thread>
#include <vector>

#include <ergodic_exploration/agent_batch.hpp>

#include "proc_ranks.hpp"

using namespace ergodic_exploration;

static int g_fail = 0, g_checks = 0;
#define CHECK(cond)                                                          \
  do {                                                                       \
    ++g_checks;                                                              \
    if (!(cond)) {                                                           \
      ++g_fail;                                                              \
      std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond);            \
    }                                                                        \
  } while (0)

namespace
{
constexpr unsigned K = 10, K2 = 100;
mat make_rinv()
{
  mat Rinv(3, 3);
  Rinv(0, 0) = 1.0;
  Rinv(1, 1) = 1.0;
  Rinv(2, 2) = 2.0;
  return Rinv;
}
const vec kUmin{ -1.0, -1.0, -2.0 }, kUmax{ 1.0, 1.0, 2.0 };
mat make_poses(unsigned n)
{
  mat p(3, n);
  for (unsigned a = 0; a < n; ++a) {
    p(0, a) = -0.7 + std::fmod(0.13 * a, 11.0);   // inside the 12 x 6 m map (clamped controls would compare trivially)
    p(1, a) = -0.5 + 0.1 * (a % 47);
    p(2, a) = -3.0 + std::fmod(0.083 * a, 6.0);
  }
  return p;
}
mat cols(const mat& m, unsigned first, unsigned n)
{
  mat out(m.n_rows(), n);
  for (unsigned c = 0; c < n; ++c) {
    for (unsigned r = 0; r < m.n_rows(); ++r) out(r, c) = m(r, first + c);
  }
  return out;
}
double max_abs_diff(const mat& a, const mat& b)
{
  double w = 0.0;
  for (unsigned c = 0; c < a.n_cols(); ++c) {
    for (unsigned r = 0; r < a.n_rows(); ++r) w = std::max(w, std::fabs(a(r, c) - b(r, c)));
  }
  return w;
}
struct World
{
  GridMap grid{ -1.0, 11.0, -1.0, 5.0, 0.05, GridData(240 * 120, 0) };
  Target target{ { Gaussian({ 2.5, 2.5 }, { 1.5, 1.5 }), Gaussian({ 8.5, 2.5 }, { 1.5, 1.5 }) } };
};

// this process' rank and the rendezvous of the two
int g_rank = 0, g_comm_no = 0;
std::string g_base;
proc_ranks::Shared* g_shared = nullptr;

// the communicator of the next test (both ranks call this in the same order); the id travels through /tmp/<base>.<n>
eea_comm* next_comm()
{
  char id[EEA_COMM_ID_BYTES] = {};
  const std::string file = "/tmp/" + g_base + "." + std::to_string(++g_comm_no);
  if (g_rank == 0) {
    throw_on_error(eea_comm_get_unique_id(id));
    proc_ranks::publish_id(file, id, sizeof(id));
  } else if (!proc_ranks::fetch_id(file, id, sizeof(id))) {
    throw std::runtime_error("no communicator id from rank 0");
  }
  eea_comm* c = nullptr;
  throw_on_error(eea_comm_create(0, 2, g_rank, id, &c));
  return c;
}
// what both ranks must hold bitwise the same: the parent compares the lines of the two children
void cross_rank(const char* label, const void* data, size_t bytes)
{
  unsigned long long h = 1469598103934665603ull;
  for (size_t i = 0; i < bytes; ++i) h = (h ^ static_cast<const unsigned char*>(data)[i]) * 1099511628211ull;
  std::printf("XR %s %016llx\n", label, h);
}

int collective_kernels_gave_up();

// (2) + (3): AgentBatch over two ranks against one rank holding all agents
void test_agent_batch_two_ranks(unsigned n0, unsigned n1, bool gather)
{
  const World w;
  const unsigned N = n0 + n1;
  const mat poses = make_poses(N), Rinv = make_rinv();
  // reference: one rank, all agents
  AgentBatch<models::Omni> ref(N, 0.1, 5.0, 0.1, 1.0, K, Rinv, kUmin, kUmax);
  ref.setTarget(w.target);
  ref.configTarget(w.grid);
  ref.setPoses(poses);
  ref.control();
  const mat ck_ref = ref.gatherTrajCoeff();
  for (int i = 0; i < 4; ++i) ref.control(true);
  const mat u_ref = ref.controls();
  const vec cbar_ref = ref.consensusTrajCoeff();

  const int rank = g_rank;
  eea_comm* const c = next_comm();
  CHECK(eea_comm_rank(c) == rank && eea_comm_nranks(c) == 2);
  const unsigned first = rank == 0 ? 0 : n0, n = rank == 0 ? n0 : n1;
  mat u, ck;
  vec cbar;
  {
    AgentBatch<models::Omni> b(n, 0.1, 5.0, 0.1, 1.0, K, Rinv, kUmin, kUmax, c);
    b.setTarget(w.target);
    b.configTarget(w.grid);
    b.setPoses(cols(poses, first, n));
    b.control();
    if (gather) ck = b.gatherTrajCoeff();         // equal shards only (one ncclAllGather)
    for (int i = 0; i < 4; ++i) b.control(true);  // back to back: no host synchronisation in between
    u = b.controls();
    cbar = b.consensusTrajCoeff();
  }
  const std::string tag = std::to_string(n0) + "+" + std::to_string(n1);
  if (gather) {
    CHECK(ck.n_rows() == K2 && ck.n_cols() == N);
    CHECK(ck.n_cols() == N && max_abs_diff(ck, ck_ref) == 0.0);  // rank order, bitwise
  }
  // consensus: the ranks' sum records are added per rank and then in rank order -- another order than one rank's tree
  double wc = 0.0;
  for (unsigned m = 0; m < K2; ++m) wc = std::max(wc, std::fabs(cbar(m) - cbar_ref(m)));
  cross_rank(("agent-batch-cbar-" + tag).c_str(), cbar.memptr(), sizeof(double) * K2);  // both ranks hold the same all-reduced record
  CHECK(wc <= 1e-13);
  const double wu = max_abs_diff(u, cols(u_ref, first, n));
  CHECK(wu <= 1e-12);
  CHECK(collective_kernels_gave_up() == 0);
  std::printf("  rank %d: agent batch %u + %u agents%s: |c_bar diff| %.2e, |u diff| %.2e\n", rank, n0, n1, gather ? " (+ gather)" : "", wc, wu);
  eea_comm_destroy(c);
}

// the test double's own count of collective-kernel blocks that gave up waiting for another rank (0 with a real RCCL)
int collective_kernels_gave_up()
{
  if (void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD)) {
    if (auto fn = reinterpret_cast<int (*)()>(dlsym(h, "fake_rccl_errors"))) return fn();
  }
  return 0;
}

struct DevBufs
{
  std::vector<void*> ptrs;
  void* alloc(size_t bytes, bool zero = true)
  {
    void* p = nullptr;
    hip_check(hipMalloc(&p, bytes ? bytes : 1));
    if (zero) hip_check(hipMemset(p, 0, bytes));
    ptrs.push_back(p);
    return p;
  }
  ~DevBufs()
  {
    for (void* p : ptrs) (void)hipFree(p);
  }
};

eea_engine* make_engine(const World& w, unsigned k, double horizon)
{
  eea_config cfg{};
  cfg.model = EEA_MODEL_OMNI;
  cfg.precision = EEA_PREC_F64;
  cfg.device = 0;
  cfg.dt = 0.1;
  cfg.horizon = horizon;
  cfg.resolution = 0.1;
  cfg.expl_weight = 1.0;
  cfg.num_basis = k;
  const mat Rinv = make_rinv();
  for (int i = 0; i < 9; ++i) cfg.Rinv[i] = Rinv.memptr()[i];
  for (int i = 0; i < 3; ++i) {
    cfg.umin[i] = kUmin(i);
    cfg.umax[i] = kUmax(i);
  }
  eea_engine* e = nullptr;
  throw_on_error(eea_create(&cfg, &e));
  std::vector<double> mu, sg;
  w.target.flatten(mu, sg);
  throw_on_error(eea_set_target_gaussians(e, static_cast<unsigned>(mu.size() / 2), mu.data(), sg.data()));
  throw_on_error(eea_config_domain(e, w.grid.xmin(), w.grid.xmax(), w.grid.ymin(), w.grid.ymax(), nullptr, nullptr));
  return e;
}

// `passes` consensus passes of n agents (two agent groups on two streams) through the device-bound exchange, lag 1;
// returns the final u0 (3 x n) and the number of agents that ever reported a status != 0
// gated (round 6, ABI 6): no in-kernel flag wait and no event -- every consuming launch of EVERY group sits behind a one-wavefront
// gate (eea_stream_wait_flag) for the flag of pass i - lag; the form for exchanges with a collective kernel
mat bound_consensus_passes(const World& w, eea_comm* c, const mat& poses, int passes, int* bad_status, bool lockstep = true,
                           bool gated = false, int lag = 1)
{
  const unsigned n = poses.n_cols();
  eea_engine* e = make_engine(w, K, 20.0);
  const unsigned T = eea_steps(e), L = eea_ck_record_len(e);
  constexpr int NB = 4;
  DevBufs d;
  double* const d_pose = static_cast<double*>(d.alloc(sizeof(double) * 3 * n));
  double* const d_ut = static_cast<double*>(d.alloc(sizeof(double) * 3 * T * n));
  double* const d_u0 = static_cast<double*>(d.alloc(sizeof(double) * 3 * n));
  int* const d_status = static_cast<int*>(d.alloc(sizeof(int) * n));
  unsigned* const d_ready = static_cast<unsigned*>(d.alloc(sizeof(unsigned) * n));
  unsigned* const d_flag = static_cast<unsigned*>(d.alloc(sizeof(unsigned)));
  unsigned* const d_gate_timeouts = static_cast<unsigned*>(d.alloc(sizeof(unsigned)));
  double* d_arec[NB];
  double* d_sum[NB];
  for (int s = 0; s < NB; ++s) {
    d_arec[s] = static_cast<double*>(d.alloc(sizeof(double) * L * n));
    d_sum[s] = static_cast<double*>(d.alloc(sizeof(double) * L));
  }
  hip_check(hipMemcpy(d_pose, poses.memptr(), sizeof(double) * 3 * n, hipMemcpyHostToDevice));
  hipStream_t streams[2];
  for (hipStream_t& s : streams) hip_check(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  std::vector<int> h_status(n);
  *bad_status = 0;
  const unsigned gb[3] = { 0, n / 2, n };
  for (int i = 0; i < passes; ++i) {
    const unsigned seq = static_cast<unsigned>(i) + 1;
    const int slot = i % NB, src = (i - lag + NB) % NB;
    for (int g = 0; g < 2; ++g) {
      const unsigned first = gb[g], cnt = gb[g + 1] - gb[g];
      eea_batch_io io{};
      io.d_pose = d_pose + 3 * first;
      io.d_ut = d_ut + static_cast<size_t>(3) * T * first;
      io.d_u0 = d_u0 + 3 * first;
      io.d_status = d_status + first;
      io.d_ck_rec = d_arec[slot] + static_cast<size_t>(L) * first;
      io.d_rec_ready = d_ready + first;
      io.rec_seq = seq;
      if (i >= lag && gated) {
        io.d_ck_shared = d_sum[src];
        io.ck_shared_parts = 1;
        throw_on_error(eea_stream_wait_flag(d_flag, seq - static_cast<unsigned>(lag), d_gate_timeouts, streams[g]));
      } else if (i >= lag) {
        io.d_ck_shared = d_sum[src];
        io.ck_shared_parts = 1;
        io.d_ck_flag = d_flag;
        io.ck_flag_seq = seq - static_cast<unsigned>(lag);
        // more than one rank: the second group consumes stream-ordered (the rule of eea_comm_records_exchange_bound: its
        // execution slots are where the collective kernel lands)
        if (g == 1 && eea_comm_nranks(c) > 1) throw_on_error(eea_comm_wait(c, src, streams[g]));
      }
      throw_on_error(eea_control_batch(e, cnt, &io, streams[g]));
    }
    throw_on_error(eea_comm_records_exchange_bound(e, c, n, d_arec[slot], d_ready, seq, d_sum[slot], d_flag, slot));
    if (!lockstep) continue;  // free-running: a time-out stays in d_status (ergodic_amd.h), read once at the end
    for (hipStream_t s : streams) hip_check(hipStreamSynchronize(s));
    hip_check(hipMemcpy(h_status.data(), d_status, sizeof(int) * n, hipMemcpyDeviceToHost));
    for (int st : h_status) *bad_status += st != 0;
  }
  hip_check(hipDeviceSynchronize());
  if (!lockstep) {
    hip_check(hipMemcpy(h_status.data(), d_status, sizeof(int) * n, hipMemcpyDeviceToHost));
    for (int st : h_status) *bad_status += st != 0;
  }
  unsigned gate_timeouts = 0;
  hip_check(hipMemcpy(&gate_timeouts, d_gate_timeouts, sizeof(unsigned), hipMemcpyDeviceToHost));
  *bad_status += static_cast<int>(gate_timeouts);
  mat u(3, n);
  hip_check(hipMemcpy(u.memptr(), d_u0, sizeof(double) * 3 * n, hipMemcpyDeviceToHost));
  for (hipStream_t s : streams) (void)hipStreamDestroy(s);
  eea_destroy(e);
  return u;
}

// (4): the device-bound exchange with two ranks
void test_bound_exchange_two_ranks(unsigned n0, unsigned n1, int passes = 5, bool lockstep = true, bool gated = false, int lag = 1)
{
  const World w;
  const unsigned N = n0 + n1;
  const mat poses = make_poses(N);
  // (the ranks take turns for the reference: each is a whole GPU's worth of device-bound passes, and two of them at once
  // break the rule that the waiting batches leave room for their producers)
  eea_comm* local = nullptr;
  throw_on_error(eea_comm_create(0, 1, 0, nullptr, &local));
  int bad_ref = 0;
  mat u_ref;
  for (int turn = 0; turn < 2; ++turn) {
    if (turn == g_rank) u_ref = bound_consensus_passes(w, local, poses, passes, &bad_ref, lockstep, gated, lag);
    proc_ranks::barrier(g_shared, 2);
  }
  eea_comm_destroy(local);
  CHECK(bad_ref == 0);
  const int rank = g_rank;
  eea_comm* const c = next_comm();
  const unsigned first = rank == 0 ? 0 : n0, n = rank == 0 ? n0 : n1;
  int bad = -1;
  const mat u = bound_consensus_passes(w, c, cols(poses, first, n), passes, &bad, lockstep, gated, lag);
  CHECK(bad == 0);
  CHECK(collective_kernels_gave_up() == 0);
  CHECK(u.n_cols() == n);
  const double wu = max_abs_diff(u, cols(u_ref, first, n));
  // five dependent passes at T = 200 with the consensus in the loop: a 1e-16 difference of c_bar (another summation order
  // over the ranks) feeds back through the warm start and the co-state (measured 4e-13 ... 2e-11): the parity bar on
  // controls, 1e-9; the long free-running case grows with the pass count like every dependent-call test
  // (every further dependent pass amplifies the difference about tenfold through the warm start: 8 passes -> 1e-6)
  const double bar = passes <= 5 ? 1e-9 : 1e-9 * std::pow(10.0, passes - 5);
  CHECK(wu <= bar);
  std::printf("  rank %d: %s exchange %u + %u agents, lag %d, %d passes%s: |u diff| %.2e, agents / gates timed out %d, collective kernels that gave up %d\n",
              rank, gated ? "GATED" : "device-bound", n0, n1, lag, passes, lockstep ? "" : " FREE-RUNNING (no host wait)", wu, bad,
              collective_kernels_gave_up());
  eea_comm_destroy(c);
}

// (5): grid-tiled occupancy target
void test_grid_tile_two_ranks()
{
  const unsigned n = 96, K5 = 12;
  const double res = 0.1, l = (n - 1) * res;
  std::vector<int8_t> occ(static_cast<size_t>(n) * n);
  for (unsigned iy = 0; iy < n; ++iy) {
    for (unsigned ix = 0; ix < n; ++ix) {
      const unsigned v = (ix / 8 + 3 * (iy / 8)) % 10;
      occ[static_cast<size_t>(iy) * n + ix] = v < 6 ? 0 : (v < 8 ? 100 : -1);
    }
  }
  const World w;
  auto phik_of = [&](eea_engine* e) {
    std::vector<double> p(K5 * K5);
    throw_on_error(eea_get_phik(e, p.data()));
    return p;
  };
  eea_engine* ref = make_engine(w, K5, 2.0);
  throw_on_error(eea_set_target_occupancy(ref, n, n, occ.data(), 0, l, l, nullptr));
  const std::vector<double> p_ref = phik_of(ref);
  eea_destroy(ref);
  const unsigned split = 41;  // ragged: 41 + 55 rows
  const int rank = g_rank;
  eea_comm* const c = next_comm();
  std::vector<double> p;
  {
    eea_engine* e = make_engine(w, K5, 2.0);
    const unsigned row0 = rank == 0 ? 0 : split, nrows = rank == 0 ? split : n - split;
    DevBufs d;
    int8_t* const d_rows = static_cast<int8_t*>(d.alloc(static_cast<size_t>(nrows) * n, false));
    double* const d_sums = static_cast<double*>(d.alloc(sizeof(double) * K5 * K5));
    hip_check(hipMemcpy(d_rows, occ.data() + static_cast<size_t>(row0) * n, static_cast<size_t>(nrows) * n, hipMemcpyHostToDevice));
    hipStream_t s = nullptr;
    hip_check(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    throw_on_error(eea_spatial_coeff_occupancy_rows(e, n, n, row0, nrows, d_rows, l, l, d_sums, s));
    throw_on_error(eea_comm_allreduce_sum(e, c, d_sums, K5 * K5, s));
    throw_on_error(eea_set_phik_from_sums(e, d_sums, l, l, s));
    hip_check(hipStreamSynchronize(s));
    p = phik_of(e);
    (void)hipStreamDestroy(s);
    eea_destroy(e);
  }
  double worst = 0.0;
  CHECK(p.size() == K5 * K5);
  for (unsigned m = 0; m < K5 * K5 && m < p.size(); ++m) worst = std::max(worst, std::fabs(p[m] - p_ref[m]));
  cross_rank("grid-tile-phik", p.data(), sizeof(double) * p.size());
  CHECK(worst <= 1e-12);
  CHECK(collective_kernels_gave_up() == 0);
  std::printf("  rank %d: grid tile 41 + 55 rows of a %ux%u occupancy grid, K = %u: |phi_k diff| %.2e\n", rank, n, n, K5, worst);
  eea_comm_destroy(c);
}
}  // namespace

int main(int argc, char** argv)
{
  const bool child = argc > 3 && std::strcmp(argv[1], "child") == 0;
  if (!child) {
    // parent: one process per rank; nothing here touches the GPU.  Compares what both ranks must hold bitwise the same.
    std::vector<std::string> out;
    const int rc = proc_ranks::spawn(argv[0], {}, 2, &out);
    std::vector<std::string> xr[2];
    int checks = 0, failures = 0, reported = 0;
    for (int r = 0; r < 2; ++r) {
      size_t pos = 0;
      while (pos < out[r].size()) {
        const size_t eol = out[r].find('\n', pos);
        const std::string line = out[r].substr(pos, eol == std::string::npos ? std::string::npos : eol - pos);
        pos = eol == std::string::npos ? out[r].size() : eol + 1;
        if (line.rfind("XR ", 0) == 0) {
          xr[r].push_back(line);
        } else {
          int a = 0, b = 0;
          if (std::sscanf(line.c_str(), "rank-done: %d checks, %d failures", &a, &b) == 2) {
            checks += a;
            failures += b;
            ++reported;
          }
          std::printf("%s\n", line.c_str());
        }
      }
    }
    ++checks;
    if (xr[0].size() != xr[1].size() || xr[0].empty()) {
      ++failures;
      std::printf("FAIL cross-rank lines: %zu vs %zu\n", xr[0].size(), xr[1].size());
    } else {
      for (size_t i = 0; i < xr[0].size(); ++i) {
        ++checks;
        if (xr[0][i] != xr[1][i]) {
          ++failures;
          std::printf("FAIL the ranks differ: %s | %s\n", xr[0][i].c_str(), xr[1][i].c_str());
        }
      }
    }
    if (reported != 2 || rc != 0) {
      ++failures;
      std::printf("FAIL a rank process did not finish (exit code %d, %d of 2 reported)\n", rc, reported);
    }
    std::printf("ranks2: %d checks, %d failures\n", checks, failures);
    return failures ? 1 : 0;
  }
  g_rank = std::atoi(argv[2]);
  g_base = argv[3];
  g_shared = proc_ranks::attach("/" + g_base);
  if (g_shared == nullptr) return 3;
  try {
    hip_check(hipSetDevice(0));
    test_agent_batch_two_ranks(35, 35, true);
    test_agent_batch_two_ranks(41, 29, false);  // ragged shards
    test_bound_exchange_two_ranks(300, 300);
    test_bound_exchange_two_ranks(77, 130);
    test_bound_exchange_two_ranks(512, 512, 8, false);
    // round 6: the gated exchange (what bench.py and consensus_bench take with a communicator), lag 2 and 1
    test_bound_exchange_two_ranks(300, 300, 6, true, true, 2);
    test_bound_exchange_two_ranks(77, 130, 5, true, true, 1);
    test_bound_exchange_two_ranks(512, 512, 8, false, true, 2);
    test_grid_tile_two_ranks();
  } catch (const std::exception& e) {
    std::printf("FAIL exception: %s\n", e.what());
    ++g_fail;
  }
  std::printf("rank-done: %d checks, %d failures\n", g_checks, g_fail);
  return g_fail ? 1 : 0;
}

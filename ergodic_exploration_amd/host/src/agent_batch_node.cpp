// agent_batch: the agent-batched, multi-GPU form of the control loop as a plain C++ program (no Python, no
// torch): one process per GPU, every rank steps its own block of independent ErgodicControl agents with
// AgentBatch<ModelT> and the ranks exchange the consensus c_k over RCCL through the C ABI (eea_comm_*).
//
//   agent_batch --ranks N [--agents B] [--steps S] [--model omni|cart] [--consensus] [--horizon H] [--basis K]
//
// The parent makes NO GPU call: it forks the N ranks first (rank r takes HIP device r), rank 0 creates the RCCL id
// and hands it to the others through pipes the parent opened before forking, every rank creates its communicator.  No counterpart in
// the single-agent reference; the per-agent computation is its ErgodicControl::control
// (include/ergodic_exploration/ergodic_control.hpp:224-311), the exchange follows README.md:225-227 (ref. [2]).
#include <sys/wait.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <ergodic_exploration/agent_batch.hpp>

namespace ee = ergodic_exploration;

namespace
{
struct Options
{
  int ranks = 1;
  unsigned agents = 4096, steps = 20, basis = 10;
  double horizon = 20.0, dt = 0.1;
  bool cart = true, consensus = false;
};

// RCCL id hand-off: one pipe per rank > 0, created by the parent before fork(); rank 0 writes, rank r reads.  (A file
// named after the parent's pid could be stale, pre-created or left behind by a crashed run.)
struct IdPipes
{
  std::vector<int> rd, wr;  // index = rank (entry 0 unused)
};

bool write_all(int fd, const char* p, size_t n)
{
  while (n > 0) {
    const ssize_t k = write(fd, p, n);
    if (k <= 0) return false;
    p += k;
    n -= static_cast<size_t>(k);
  }
  return true;
}
bool read_all(int fd, char* p, size_t n)
{
  while (n > 0) {
    const ssize_t k = read(fd, p, n);
    if (k <= 0) return false;  // 0 = rank 0 died before writing: every write end is closed
    p += k;
    n -= static_cast<size_t>(k);
  }
  return true;
}

template <class ModelT>
int run_rank(const Options& o, int rank, const IdPipes& pipes)
{
  ee::device_ordinal() = rank;
  eea_comm* comm = nullptr;
  if (o.ranks > 1) {
    char id[EEA_COMM_ID_BYTES];
    // every rank keeps only its own end: a reader then sees end-of-file if rank 0 dies before writing
    for (int r = 1; r < o.ranks; ++r) {
      if (rank != 0) close(pipes.wr[r]);
      if (rank != r) close(pipes.rd[r]);
    }
    if (rank == 0) {
      ee::throw_on_error(eea_comm_get_unique_id(id));
      for (int r = 1; r < o.ranks; ++r) {
        if (!write_all(pipes.wr[r], id, sizeof(id))) throw std::runtime_error("cannot hand the RCCL id to a rank");
        close(pipes.wr[r]);
      }
    } else {
      if (!read_all(pipes.rd[rank], id, sizeof(id))) throw std::runtime_error("no RCCL id from rank 0");
      close(pipes.rd[rank]);
    }
    ee::throw_on_error(eea_comm_create(rank, o.ranks, rank, id, &comm));
  }
  ee::mat Rinv(3, 3);
  Rinv(0, 0) = 1.0;
  Rinv(1, 1) = o.cart ? 0.0 : 1.0;
  Rinv(2, 2) = 2.0;
  const double vy = o.cart ? 0.0 : 1.0;
  const ee::vec umin{ -1.0, -vy, -2.0 }, umax{ 1.0, vy, 2.0 };
  {
    ee::AgentBatch<ModelT> batch(o.agents, o.dt, o.horizon, 0.1, 1.0, o.basis, Rinv, umin, umax, comm);
    batch.setTarget(ee::Target({ ee::Gaussian({ 2.5, 2.5 }, { 1.5, 1.5 }), ee::Gaussian({ 8.5, 2.5 }, { 1.5, 1.5 }) }));
    const ee::GridMap grid(-1.0, 11.0, -1.0, 5.0, 0.1, ee::GridData(120 * 60, 0));
    batch.configTarget(grid);
    ee::mat poses(3, o.agents);
    unsigned long long s = 12345ull + 977ull * rank;  // splitmix-style stream per rank
    auto uni = [&s]() {
      s += 0x9E3779B97F4A7C15ull;
      unsigned long long z = s;
      z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
      z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
      return static_cast<double>((z ^ (z >> 31)) >> 11) / 9007199254740992.0;
    };
    for (unsigned a = 0; a < o.agents; ++a) {
      poses(0, a) = -1.0 + 0.5 + 11.0 * uni();
      poses(1, a) = -1.0 + 0.5 + 5.0 * uni();
      poses(2, a) = -3.14159 + 6.28318 * uni();
    }
    batch.setPoses(poses);
    for (int w = 0; w < 3; ++w) batch.control(o.consensus);
    batch.sync();
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned i = 0; i < o.steps; ++i) batch.control(o.consensus);
    batch.sync();
    const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    const ee::mat u = batch.controls();
    double sum = 0.0;
    for (unsigned a = 0; a < o.agents; ++a) sum += u(0, a) + u(2, a);
    std::printf("rank %d of %d: %u agents x %u steps in %.3f ms = %.4g optimisations/s%s; checksum %.12g\n", rank,
                o.ranks, o.agents, o.steps, 1e3 * sec, o.agents * static_cast<double>(o.steps) / sec,
                o.consensus ? " (consensus c_k every step)" : "", sum);
  }
  if (comm) eea_comm_destroy(comm);
  return 0;
}
}  // namespace

int main(int argc, char** argv)
{
  Options o;
  for (int i = 1; i < argc; ++i) {
    const std::string a = argv[i];
    auto next = [&]() -> const char* { return i + 1 < argc ? argv[++i] : "0"; };
    if (a == "--ranks") o.ranks = std::atoi(next());
    else if (a == "--agents") o.agents = static_cast<unsigned>(std::atoi(next()));
    else if (a == "--steps") o.steps = static_cast<unsigned>(std::atoi(next()));
    else if (a == "--basis") o.basis = static_cast<unsigned>(std::atoi(next()));
    else if (a == "--horizon") o.horizon = std::atof(next());
    else if (a == "--model") o.cart = std::string(next()) != "omni";
    else if (a == "--consensus") o.consensus = true;
    else {
      std::fprintf(stderr, "usage: agent_batch --ranks N [--agents B] [--steps S] [--model omni|cart] [--consensus] "
                           "[--horizon H] [--basis K]\n");
      return 2;
    }
  }
  if (o.ranks < 1 || o.agents == 0) return 2;
  IdPipes pipes;
  pipes.rd.assign(static_cast<size_t>(o.ranks), -1);
  pipes.wr.assign(static_cast<size_t>(o.ranks), -1);
  for (int r = 1; r < o.ranks; ++r) {
    int fd[2];
    if (pipe(fd) != 0) return 1;
    pipes.rd[r] = fd[0];
    pipes.wr[r] = fd[1];
  }
  auto rank_main = [&](int rank) -> int {
    try {
      return o.cart ? run_rank<ee::models::SimpleCart>(o, rank, pipes) : run_rank<ee::models::Omni>(o, rank, pipes);
    } catch (const std::exception& e) {
      std::fprintf(stderr, "rank %d: %s\n", rank, e.what());
      return 1;
    }
  };
  if (o.ranks == 1) return rank_main(0);
  // the ranks are forked BEFORE anything touches the GPU (a process that has initialised HIP must not fork)
  int failed = 0;
  for (int r = 0; r < o.ranks; ++r) {
    const pid_t pid = fork();
    if (pid == 0) _exit(rank_main(r));
    if (pid < 0) return 1;
  }
  for (int r = 1; r < o.ranks; ++r) {  // the parent holds no end of the pipes
    close(pipes.rd[r]);
    close(pipes.wr[r]);
  }
  for (int r = 0; r < o.ranks; ++r) {
    int status = 0;
    if (wait(&status) < 0 || !WIFEXITED(status) || WEXITSTATUS(status) != 0) failed = 1;
  }
  return failed;
}

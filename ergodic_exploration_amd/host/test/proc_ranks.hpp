// Ranks as PROCESSES on one box (one process per rank, as in production), for the host tests that need more than one rank
// on a one-GPU machine: the parent -- which never touches the GPU -- re-executes the program once per rank; the ranks find
// each other through a POSIX shared-memory segment (barrier + a few result slots) and hand the communicator's unique id
// over through a file.  Test infrastructure.
#pragma once

#include <fcntl.h>
#include <poll.h>
#include <sys/mman.h>
#include <sys/wait.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace proc_ranks
{
struct Shared
{
  std::atomic<int> count, gen;
  double vals[8][8];
  int ints[8][8];
};

inline Shared* attach(const std::string& name)
{
  const int fd = shm_open(name.c_str(), O_CREAT | O_RDWR, 0600);
  if (fd < 0 || ftruncate(fd, sizeof(Shared)) != 0) return nullptr;
  void* const m = mmap(nullptr, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  return m == MAP_FAILED ? nullptr : static_cast<Shared*>(m);  // (zero-filled when created)
}

// every rank leaves together
inline void barrier(Shared* s, int nranks)
{
  const int gen = s->gen.load();
  if (s->count.fetch_add(1) + 1 == nranks) {
    s->count.store(0);
    s->gen.fetch_add(1);
  } else {
    while (s->gen.load() == gen) std::this_thread::sleep_for(std::chrono::microseconds(50));
  }
}

// the communicator id of rank 0 for the others: written under another name, then renamed (never read half)
inline void publish_id(const std::string& file, const void* id, size_t bytes)
{
  const std::string tmp = file + ".tmp";
  FILE* f = std::fopen(tmp.c_str(), "wb");
  std::fwrite(id, bytes, 1, f);
  std::fclose(f);
  std::rename(tmp.c_str(), file.c_str());
}
inline bool fetch_id(const std::string& file, void* id, size_t bytes, double timeout_s = 60.0)
{
  const auto t0 = std::chrono::steady_clock::now();
  for (;;) {
    if (FILE* f = std::fopen(file.c_str(), "rb")) {
      const bool ok = std::fread(id, bytes, 1, f) == 1;
      std::fclose(f);
      return ok;
    }
    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s) return false;
    std::this_thread::sleep_for(std::chrono::milliseconds(1));
  }
}

// parent: start `nranks` copies of this program with the arguments `args` + { "child", rank, base }, where base names the
// shared segment ("/" + base) and prefixes the id files ("/tmp/" + base + ".<n>").  Returns 0 if every child returned 0;
// the children's standard output is collected per child (in rank order) when `outputs` is given.
// The parent is a fork + execv LAUNCHER and must never have initialised the GPU (on this pool an exec from a process that has
// takes the machine down).  That holds for a plain run -- the parent makes no HIP call -- but NOT under a profiler: rocprofv3
// preloads a tool library that initialises the GPU before main().  So the launcher form refuses to start under a profiler
// preload; profile ONE rank by putting the `child <rank> <base>` form itself after `rocprofv3 --`.
inline bool profiler_preloaded()
{
  for (const char* var : { "LD_PRELOAD", "ROCPROFILER_REGISTER_FORCE_LOAD", "ROCP_TOOL_LIBRARIES", "ROCPROF_ATT_LIBRARY_PATH" }) {
    const char* v = std::getenv(var);
    if (v != nullptr && (std::strstr(v, "rocprof") != nullptr || std::strstr(v, "roctracer") != nullptr ||
                         std::strstr(v, "rocprofiler") != nullptr)) {
      return true;
    }
  }
  return false;
}

inline int spawn(const char* self, const std::vector<std::string>& args, int nranks, std::vector<std::string>* outputs = nullptr)
{
  if (profiler_preloaded()) {
    std::fprintf(stderr, "proc_ranks::spawn: refusing to fork + exec the ranks under a profiler preload (the preloaded tool "
                         "has initialised the GPU in this process); profile one rank as `%s ... child <rank> <base>`\n", self);
    return 4;
  }
  const std::string base = "eea-ranks-" + std::to_string(getpid());
  shm_unlink(("/" + base).c_str());
  std::vector<pid_t> kids;
  std::vector<int> fds;
  for (int r = 0; r < nranks; ++r) {
    int pfd[2] = { -1, -1 };
    if (outputs && pipe(pfd) != 0) return 3;
    const pid_t pid = fork();
    if (pid == 0) {
      if (outputs) {
        dup2(pfd[1], 1);
        close(pfd[0]);
        close(pfd[1]);
      }
      std::vector<std::string> a = args;
      a.push_back("child");
      a.push_back(std::to_string(r));
      a.push_back(base);
      std::vector<char*> av;
      av.push_back(const_cast<char*>(self));
      for (std::string& x : a) av.push_back(const_cast<char*>(x.c_str()));
      av.push_back(nullptr);
      execv(self, av.data());
      _exit(127);
    }
    if (outputs) {
      close(pfd[1]);
      fds.push_back(pfd[0]);
    }
    kids.push_back(pid);
  }
  if (outputs) {
    outputs->assign(nranks, std::string());
    // all pipes are drained TOGETHER (poll): the children barrier with each other, so a child blocked on a full pipe while
    // the parent reads another child's to its end would dead-lock the whole group
    std::vector<pollfd> pf(nranks);
    int open_fds = nranks;
    for (int r = 0; r < nranks; ++r) pf[r] = pollfd{ fds[r], POLLIN, 0 };
    while (open_fds > 0) {
      if (poll(pf.data(), static_cast<nfds_t>(nranks), -1) < 0) {
        if (errno == EINTR) continue;
        break;
      }
      for (int r = 0; r < nranks; ++r) {
        if (pf[r].fd < 0 || (pf[r].revents & (POLLIN | POLLHUP | POLLERR)) == 0) continue;
        char buf[4096];
        const ssize_t n = read(pf[r].fd, buf, sizeof(buf));
        if (n > 0) {
          (*outputs)[r].append(buf, static_cast<size_t>(n));
        } else if (n == 0 || errno != EINTR) {
          close(pf[r].fd);
          pf[r].fd = -1;  // (poll ignores negative descriptors)
          --open_fds;
        }
      }
    }
    for (int r = 0; r < nranks; ++r) {
      if (pf[r].fd >= 0) close(pf[r].fd);
    }
  }
  int worst = 0;
  for (pid_t k : kids) {
    int st = 0;
    waitpid(k, &st, 0);
    if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) worst = worst ? worst : (WIFEXITED(st) ? WEXITSTATUS(st) : 9);
  }
  shm_unlink(("/" + base).c_str());
  for (int n = 0; n < 64; ++n) std::remove(("/tmp/" + base + "." + std::to_string(n)).c_str());
  return worst;
}
}  // namespace proc_ranks

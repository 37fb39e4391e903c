// Ranks as PROCESSES on one box (one process per rank, as in production), for the host tests that need more than one rank
// on a one-GPU machine: the parent -- which never touches the GPU -- re-executes the program once per rank; the ranks find
// each other through a POSIX shared-memory segment (barrier + a few result slots) and hand the communicator's unique id
// over through a file.  Test infrastructure.
#pragma once

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/wait.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
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
inline int spawn(const char* self, const std::vector<std::string>& args, int nranks, std::vector<std::string>* outputs = nullptr)
{
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
    // (the children print a few KB: the pipes are drained one after the other)
    for (int r = 0; r < nranks; ++r) {
      char buf[4096];
      ssize_t n;
      while ((n = read(fds[r], buf, sizeof(buf))) > 0) (*outputs)[r].append(buf, static_cast<size_t>(n));
      close(fds[r]);
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

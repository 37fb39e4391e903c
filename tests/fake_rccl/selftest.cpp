// Stand-alone check of the RCCL test double: two (or more) ranks as threads on one GPU, all-reduce and all-gather (in place
// and out of place, small and multi-block), operations issued back to back without host waits, then verified.
// usage: selftest <path to librccl.so.1> [ranks = 2] [extra streams = 0] [procs]
// procs: the ranks are PROCESSES (this program re-executed once per rank before anything touches the GPU; the unique id
// travels through a file), as in production -- each process has its own hardware queues.
// extra streams: that many more streams are created (and used once) before the ranks' own -- a HIP process maps its
// normal-priority streams onto GPU_MAX_HW_QUEUES = 4 hardware queues; two ranks whose streams share a queue cannot meet on
// the device (the first one's kernel spins in front of the second one's).
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <sys/wait.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

typedef struct { char internal[128]; } ncclUniqueId;
typedef void* ncclComm_t;
typedef int (*GetId)(ncclUniqueId*);
typedef int (*Init)(ncclComm_t*, int, ncclUniqueId, int);
typedef int (*Gather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t);
typedef int (*Reduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t);
typedef int (*Errs)();

#define HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); std::exit(1); } } while (0)

int main(int argc, char** argv)
{
  if (argc < 2) return 1;
  const int R = argc > 2 ? std::atoi(argv[2]) : 2;
  const int extra = argc > 3 ? std::atoi(argv[3]) : 0;
  const bool procs = argc > 4 && std::strcmp(argv[4], "procs") == 0;
  const int child = argc > 6 && std::strcmp(argv[4], "child") == 0 ? std::atoi(argv[5]) : -1;
  if (procs) {  // parent: nothing here touches the GPU
    const std::string idfile = "/tmp/fake_rccl_selftest_" + std::to_string(getpid());
    std::remove(idfile.c_str());
    std::vector<pid_t> kids;
    for (int r = 0; r < R; ++r) {
      const pid_t pid = fork();
      if (pid == 0) {
        const std::string rs = std::to_string(r);
        execl(argv[0], argv[0], argv[1], argv[2], argv[3], "child", rs.c_str(), idfile.c_str(), (char*)nullptr);
        _exit(127);
      }
      kids.push_back(pid);
    }
    int worst = 0;
    for (pid_t k : kids) {
      int st = 0;
      waitpid(k, &st, 0);
      if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) worst = 2;
    }
    std::remove(idfile.c_str());
    return worst;
  }
  void* h = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
  if (!h) { std::printf("dlopen: %s\n", dlerror()); return 1; }
  GetId get_id = (GetId)dlsym(h, "ncclGetUniqueId");
  Init init = (Init)dlsym(h, "ncclCommInitRank");
  Gather gather = (Gather)dlsym(h, "ncclAllGather");
  Reduce reduce = (Reduce)dlsym(h, "ncclAllReduce");
  Errs errs = (Errs)dlsym(h, "fake_rccl_errors");
  HIP(hipSetDevice(0));
  ncclUniqueId id;
  if (child < 0) {
    get_id(&id);
  } else if (child == 0) {  // the id travels through a file (written under another name, then renamed: never read half)
    get_id(&id);
    const std::string tmp = std::string(argv[6]) + ".tmp";
    FILE* f = std::fopen(tmp.c_str(), "wb");
    std::fwrite(&id, sizeof(id), 1, f);
    std::fclose(f);
    std::rename(tmp.c_str(), argv[6]);
  } else {
    FILE* f = nullptr;
    while ((f = std::fopen(argv[6], "rb")) == nullptr) std::this_thread::sleep_for(std::chrono::milliseconds(1));
    if (std::fread(&id, sizeof(id), 1, f) != 1) return 1;
    std::fclose(f);
  }
  std::vector<hipStream_t> idle(extra);
  for (hipStream_t& q : idle) {
    HIP(hipStreamCreateWithFlags(&q, hipStreamNonBlocking));
    void* p = nullptr;
    HIP(hipMalloc(&p, 256));
    HIP(hipMemsetAsync(p, 0, 256, q));
    HIP(hipStreamSynchronize(q));
  }
  const size_t small = 102, big = 200000;
  const int iters = 50;
  std::vector<int> bad(R, 0);
  std::vector<double> secs(R, 0.0);
  auto body = [&](int r) {
    HIP(hipSetDevice(0));
    ncclComm_t c = nullptr;
    if (init(&c, R, id, r) != 0) { std::printf("init failed\n"); std::exit(1); }
    hipStream_t s;
    HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    double *d_small, *d_big, *d_all;
    HIP(hipMalloc(&d_small, sizeof(double) * small * iters));
    HIP(hipMalloc(&d_big, sizeof(double) * big));
    HIP(hipMalloc(&d_all, sizeof(double) * big * R));
    std::vector<double> hs(small * iters), hb(big);
    for (size_t i = 0; i < hs.size(); ++i) hs[i] = (r + 1) * 1000.0 + i;
    for (size_t i = 0; i < hb.size(); ++i) hb[i] = (r + 1) * 0.5 + i;
    HIP(hipMemcpy(d_small, hs.data(), sizeof(double) * hs.size(), hipMemcpyHostToDevice));
    HIP(hipMemcpy(d_big, hb.data(), sizeof(double) * hb.size(), hipMemcpyHostToDevice));
    const auto t0 = std::chrono::steady_clock::now();
    for (int it = 0; it < iters; ++it) reduce(d_small + small * it, d_small + small * it, small, 8, 0, c, s);  // in place, back to back
    gather(d_big, d_all, big, 8, c, s);
    reduce(d_big, d_big, big, 8, 0, c, s);
    HIP(hipStreamSynchronize(s));
    secs[r] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::vector<double> gs(hs.size()), gb(big), ga(big * R);
    HIP(hipMemcpy(gs.data(), d_small, sizeof(double) * gs.size(), hipMemcpyDeviceToHost));
    HIP(hipMemcpy(gb.data(), d_big, sizeof(double) * big, hipMemcpyDeviceToHost));
    HIP(hipMemcpy(ga.data(), d_all, sizeof(double) * big * R, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < gs.size(); ++i) {
      double want = 0.0;
      for (int q = 0; q < R; ++q) want += (q + 1) * 1000.0 + i;
      bad[r] += gs[i] != want;
    }
    for (size_t i = 0; i < big; ++i) {
      double want = 0.0;
      for (int q = 0; q < R; ++q) want += (q + 1) * 0.5 + i;
      bad[r] += gb[i] != want;
      for (int q = 0; q < R; ++q) bad[r] += ga[q * big + i] != (q + 1) * 0.5 + i;
    }
  };
  std::vector<std::thread> th;
  if (child >= 0) {
    body(child);
  } else {
    for (int r = 1; r < R; ++r) th.emplace_back(body, r);
    body(0);
  }
  for (auto& t : th) t.join();
  int total = 0;
  for (int r = 0; r < R; ++r) {
    if (child >= 0 && r != child) continue;
    std::printf("rank %d: %d wrong values, %.3f ms for %d small all-reduces + 1 all-gather + 1 large all-reduce\n", r, bad[r], 1e3 * secs[r], iters);
    total += bad[r];
  }
  std::printf("collective kernel blocks that gave up: %d\n", errs ? errs() : -1);
  return (total == 0 && (!errs || errs() == 0)) ? 0 : 2;
}

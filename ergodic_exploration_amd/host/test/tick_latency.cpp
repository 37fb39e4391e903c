// One robot, one control() per tick (the reference's own use, exploration.hpp:232) from a C++ host through the C ABI:
// microseconds per dependent eea_control call -- one launch per call, and served by the resident workgroup
// (EEA_OPT_RESIDENT_CONTROL) -- at the BASELINE shapes.  bench.py quotes these beside its own Python-side numbers.
// usage: tick_latency [calls = 2000]      last line: RESULT {json}
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "ergodic_amd.h"

namespace
{
struct Shape
{
  const char* name;
  int model;
  unsigned K;
  double dt, horizon;
  int precision;
};
double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
void ok(eea_status s, const char* what)
{
  if (s != EEA_OK) {
    std::fprintf(stderr, "%s: %s\n", what, eea_last_error());
    std::exit(1);
  }
}
}  // namespace

int main(int argc, char** argv)
{
  const int calls = argc > 1 ? std::atoi(argv[1]) : 2000;
  const Shape shapes[] = {
    { "configs[0]", EEA_MODEL_OMNI, 5, 0.1, 0.5, EEA_PREC_F64 },
    { "configs[1]", EEA_MODEL_SIMPLE_CART, 10, 0.1, 2.0, EEA_PREC_F64 },
    { "explore_omni.yaml (K = 10, T = 50)", EEA_MODEL_OMNI, 10, 0.1, 5.0, EEA_PREC_F64 },
    { "configs[2]", EEA_MODEL_OMNI, 20, 0.02, 5.0, EEA_PREC_F32 },
    { "configs[3] (one agent of the batch)", EEA_MODEL_SIMPLE_CART, 10, 0.1, 20.0, EEA_PREC_F64 },
  };
  std::string json = "[";
  for (const Shape& sh : shapes) {
    eea_config cfg{};
    cfg.model = sh.model;
    cfg.precision = sh.precision;
    cfg.dt = sh.dt;
    cfg.horizon = sh.horizon;
    cfg.resolution = 0.1;
    cfg.expl_weight = 1.0;
    cfg.num_basis = sh.K;
    cfg.Rinv[0] = 1.0;
    cfg.Rinv[4] = sh.model == EEA_MODEL_OMNI ? 1.0 : 0.0;
    cfg.Rinv[8] = 2.0;
    const double lim[3] = { 1.0, sh.model == EEA_MODEL_OMNI ? 1.0 : 0.0, 2.0 };
    for (int i = 0; i < 3; ++i) {
      cfg.umin[i] = -lim[i];
      cfg.umax[i] = lim[i];
    }
    eea_engine* e = nullptr;
    ok(eea_create(&cfg, &e), "eea_create");
    const double mu[4] = { 2.5, 2.5, 8.5, 2.5 }, sg[4] = { 1.5, 1.5, 1.5, 1.5 };
    ok(eea_set_target_gaussians(e, 2, mu, sg), "set_target");
    const double x[3] = { 3.0, 2.0, 0.3 };
    double u[3], us[2] = { 0.0, 0.0 };
    for (int mode = 0; mode < 2; ++mode) {
      ok(eea_set_option(EEA_OPT_RESIDENT_CONTROL, mode), "option");
      for (int i = 0; i < 50; ++i) ok(eea_control(e, -1.0, 11.0, -1.0, 5.0, x, nullptr, 0, u), "eea_control");
      const double t0 = now();
      for (int i = 0; i < calls; ++i) ok(eea_control(e, -1.0, 11.0, -1.0, 5.0, x, nullptr, 0, u), "eea_control");
      us[mode] = 1e6 * (now() - t0) / calls;
    }
    ok(eea_set_option(EEA_OPT_RESIDENT_CONTROL, 0), "option");
    std::printf("%-40s T = %3u  %s   one launch per call %6.2f us   resident workgroup %6.2f us\n", sh.name, eea_steps(e),
                sh.precision == EEA_PREC_F32 ? "f32" : "f64", us[0], us[1]);
    char buf[256];
    std::snprintf(buf, sizeof(buf), "%s{\"config\": \"%s\", \"horizon_steps\": %u, \"launch_us_per_call\": %.3f, \"resident_us_per_call\": %.3f}",
                  json.size() > 1 ? ", " : "", sh.name, eea_steps(e), us[0], us[1]);
    json += buf;
    eea_destroy(e);
  }
  std::printf("RESULT %s]\n", json.c_str());
  return 0;
}

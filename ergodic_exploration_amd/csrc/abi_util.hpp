// Error plumbing shared by the translation units behind include/ergodic_amd.h: the thread-local message
// of eea_last_error() and the status helpers.  Internal to libergodic_amd.so.
#pragma once

#include <hip/hip_runtime.h>

#include <string>

#include "../../include/ergodic_amd.h"

namespace eea
{
inline thread_local std::string g_last_error;

inline eea_status fail(eea_status st, const std::string& msg)
{
  g_last_error = msg;
  return st;
}
}  // namespace eea

#define EEA_HIP(expr)                                                                         \
  do {                                                                                        \
    const hipError_t err__ = (expr);                                                          \
    if (err__ != hipSuccess) {                                                                \
      return eea::fail(EEA_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(err__));    \
    }                                                                                         \
  } while (0)

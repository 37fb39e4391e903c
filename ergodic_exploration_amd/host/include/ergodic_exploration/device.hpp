// Glue between the host mirror and the C ABI of the gfx950 engine (include/ergodic_amd.h).
#pragma once

#include <hip/hip_runtime_api.h>

#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <vector>

#include <ergodic_amd.h>

#include <ergodic_exploration/models/cart.hpp>
#include <ergodic_exploration/models/omni.hpp>

namespace ergodic_exploration
{
// HIP device the mirror's free operations run on (one process per GPU)
inline int& device_ordinal()
{
  static int dev = 0;
  return dev;
}

// maps an eea_status to the exception the reference would have thrown at that place
inline void throw_on_error(eea_status st)
{
  if (st == EEA_OK) return;
  const std::string msg = eea_last_error();
  if (st == EEA_ERR_INVALID_ARGUMENT || st == EEA_ERR_INVALID_TWIST) throw std::invalid_argument(msg);
  throw std::runtime_error("ergodic_amd: " + msg);
}

inline void hip_check(hipError_t e)
{
  if (e != hipSuccess) throw std::runtime_error(std::string("hip: ") + hipGetErrorString(e));
}

// device copy of a GridMap's cells, created on first use and shared by the map's copies
template <class GridT>
inline const int8_t* device_cells(const GridT& grid)
{
  std::shared_ptr<const void>& slot = grid.deviceCache();
  if (!slot) {
    hip_check(hipSetDevice(device_ordinal()));
    void* p = nullptr;
    const size_t cells = grid.gridData().size();
    hip_check(hipMalloc(&p, cells ? cells : 1));
    if (cells) hip_check(hipMemcpy(p, grid.gridData().data(), cells, hipMemcpyHostToDevice));
    slot = std::shared_ptr<const void>(p, [](const void* q) { (void)hipFree(const_cast<void*>(q)); });
  }
  return static_cast<const int8_t*>(slot.get());
}

// grow-only device scratch of the calling thread (payloads the kernels read many times)
inline void* device_scratch(size_t bytes)
{
  struct Scratch
  {
    void* p = nullptr;
    size_t cap = 0;
    ~Scratch()
    {
      if (p) (void)hipFree(p);
    }
  };
  static thread_local Scratch s;
  if (bytes > s.cap) {
    hip_check(hipSetDevice(device_ordinal()));
    if (s.p) (void)hipFree(s.p);
    s.p = nullptr;
    s.cap = 0;
    const size_t want = bytes < 4096 ? 4096 : bytes;
    hip_check(hipMalloc(&s.p, want));
    s.cap = want;
  }
  return s.p;
}

// Grow-only pinned, device-mapped scratch of the calling thread for the arguments and results of one
// call: the kernels read / write it over PCIe directly (a few dozen bytes), so a call is one launch and
// one wait instead of three blocking copies.
struct PinnedScratch
{
  void* host = nullptr;
  void* dev = nullptr;
};
inline PinnedScratch pinned_scratch(size_t bytes)
{
  struct Holder
  {
    PinnedScratch s;
    size_t cap = 0;
    ~Holder()
    {
      if (s.host) (void)hipHostFree(s.host);
    }
  };
  static thread_local Holder h;
  if (bytes > h.cap) {
    hip_check(hipSetDevice(device_ordinal()));
    if (h.s.host) (void)hipHostFree(h.s.host);
    h.s = PinnedScratch{};
    h.cap = 0;
    const size_t want = bytes < 4096 ? 4096 : bytes;
    hip_check(hipHostMalloc(&h.s.host, want, hipHostMallocMapped | hipHostMallocCoherent));
    hip_check(hipHostGetDevicePointer(&h.s.dev, h.s.host, 0));
    h.cap = want;
  }
  return h.s;
}

// models the device engine implements; everything else is host-side class surface
template <class ModelT>
struct device_model : std::integral_constant<int, -1>
{
};
template <>
struct device_model<models::Omni> : std::integral_constant<int, EEA_MODEL_OMNI>
{
};
template <>
struct device_model<models::SimpleCart> : std::integral_constant<int, EEA_MODEL_SIMPLE_CART>
{
};
}  // namespace ergodic_exploration

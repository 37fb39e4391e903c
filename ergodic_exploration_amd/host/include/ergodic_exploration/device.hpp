// Glue between the host mirror and the C ABI of the gfx950 engine (include/ergodic_amd.h).
#pragma once

#include <stdexcept>
#include <string>
#include <type_traits>

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

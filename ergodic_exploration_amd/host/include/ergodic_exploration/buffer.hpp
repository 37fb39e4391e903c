// ReplayBuffer: past poses kept on the host (reference buffer.hpp / buffer.cpp).  Sampling stays
// on the host because its random stream is host state; the device engine receives the sampled
// columns (sampleColumns) and prepends them itself.
#pragma once

#include <iostream>
#include <random>
#include <vector>

#include <ergodic_exploration/types.hpp>

namespace ergodic_exploration
{
class ReplayBuffer
{
public:
  ReplayBuffer(unsigned int buffer_size, unsigned int batch_size, unsigned long seed = 5489u)
    : buffer_size_(buffer_size), batch_size_(batch_size), rng_(seed)
  {
  }

  void append(const vec& x)
  {
    if (memory_.size() < buffer_size_) {
      memory_.push_back(x);
      return;
    }
    std::cout << "WARNING: Buffer is full" << std::endl;
  }

  // the columns sampleMemory() puts in front of the rollout: everything while the memory is no
  // larger than the batch, else batch_size draws with replacement
  mat sampleColumns() const
  {
    if (memory_.empty()) return mat(3, 0);
    if (memory_.size() <= batch_size_) {
      mat m(3, memory_.size());
      for (std::size_t i = 0; i < memory_.size(); ++i) m.set_col(i, memory_[i]);
      return m;
    }
    mat m(3, batch_size_);
    std::uniform_int_distribution<std::size_t> pick(0, memory_.size() - 1);
    for (unsigned int i = 0; i < batch_size_; ++i) m.set_col(i, memory_[pick(rng_)]);
    return m;
  }

  // predicted trajectory with the sampled past states in front
  mat sampleMemory(const mat& xt) const
  {
    const mat head = sampleColumns();
    if (head.n_cols() == 0) return xt;
    mat out(xt.n_rows(), head.n_cols() + xt.n_cols());
    for (std::size_t j = 0; j < head.n_cols(); ++j) out.set_col(j, head.col(j));
    for (std::size_t j = 0; j < xt.n_cols(); ++j) out.set_col(head.n_cols() + j, xt.col(j));
    return out;
  }

  std::size_t size() const { return memory_.size(); }

private:
  unsigned int buffer_size_, batch_size_;
  std::vector<vec> memory_;
  mutable std::mt19937_64 rng_;
};
}  // namespace ergodic_exploration

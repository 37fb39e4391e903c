// Minimal dense vector / matrix types for the host mirror of the reference class surface.
// Column-major like Armadillo (the reference's arma::vec / arma::mat), so memptr() buffers can
// be handed to the C ABI (include/ergodic_amd.h) unchanged.  Only what the mirror needs.
#pragma once

#include <cstddef>
#include <initializer_list>
#include <stdexcept>
#include <vector>

namespace ergodic_exploration
{
class vec
{
public:
  vec() = default;
  explicit vec(std::size_t n, double fill = 0.0) : d_(n, fill) {}
  vec(std::initializer_list<double> v) : d_(v) {}
  std::size_t size() const { return d_.size(); }
  std::size_t n_rows() const { return d_.size(); }
  double& operator()(std::size_t i) { return d_.at(i); }
  double operator()(std::size_t i) const { return d_.at(i); }
  double* memptr() { return d_.data(); }
  const double* memptr() const { return d_.data(); }
  void fill(double v) { d_.assign(d_.size(), v); }

private:
  std::vector<double> d_;
};

class mat
{
public:
  mat() = default;
  mat(std::size_t rows, std::size_t cols, double fill = 0.0) : r_(rows), c_(cols), d_(rows * cols, fill) {}
  // row-wise initialiser like arma: { {r0c0, r0c1}, {r1c0, r1c1} }
  mat(std::initializer_list<std::initializer_list<double>> rows)
  {
    r_ = rows.size();
    c_ = r_ ? rows.begin()->size() : 0;
    d_.assign(r_ * c_, 0.0);
    std::size_t i = 0;
    for (const auto& row : rows) {
      if (row.size() != c_) throw std::invalid_argument("ragged matrix initialiser");
      std::size_t j = 0;
      for (double v : row) d_[i + r_ * j++] = v;
      ++i;
    }
  }
  std::size_t n_rows() const { return r_; }
  std::size_t n_cols() const { return c_; }
  double& operator()(std::size_t i, std::size_t j) { return d_.at(i + r_ * j); }
  double operator()(std::size_t i, std::size_t j) const { return d_.at(i + r_ * j); }
  double* memptr() { return d_.data(); }
  const double* memptr() const { return d_.data(); }
  double* colptr(std::size_t j) { return d_.data() + r_ * j; }
  const double* colptr(std::size_t j) const { return d_.data() + r_ * j; }
  vec col(std::size_t j) const
  {
    vec v(r_);
    for (std::size_t i = 0; i < r_; ++i) v(i) = (*this)(i, j);
    return v;
  }
  void set_col(std::size_t j, const vec& v)
  {
    if (v.size() != r_) throw std::invalid_argument("column size mismatch");
    for (std::size_t i = 0; i < r_; ++i) (*this)(i, j) = v(i);
  }
  void resize(std::size_t rows, std::size_t cols)
  {
    r_ = rows;
    c_ = cols;
    d_.assign(rows * cols, 0.0);
  }
  void fill(double v) { d_.assign(d_.size(), v); }

private:
  std::size_t r_ = 0, c_ = 0;
  std::vector<double> d_;
};
}  // namespace ergodic_exploration

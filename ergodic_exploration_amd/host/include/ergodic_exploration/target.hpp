// Target distribution: sum of axis-aligned, un-normalised Gaussians (reference target.hpp /
// target.cpp).  fill() over a point list runs on the device (eea_target_fill).
#pragma once

#include <cmath>
#include <vector>

#include <ergodic_exploration/device.hpp>

namespace ergodic_exploration
{
struct Gaussian
{
  Gaussian() {}
  Gaussian(const vec& mu, const vec& sigmas) : mu(mu), sigmas(sigmas), cov(2, 2), cov_inv(2, 2)
  {
    cov(0, 0) = sigmas(0) * sigmas(0);
    cov(1, 1) = sigmas(1) * sigmas(1);
    const double det = cov(0, 0) * cov(1, 1);
    cov_inv(0, 0) = cov(1, 1) / det;
    cov_inv(1, 1) = cov(0, 0) / det;
  }
  double operator()(const vec& pt) const { return (*this)(pt, vec{ 0.0, 0.0 }); }
  // trans: translation map frame -> Fourier domain, applied to the mean
  double operator()(const vec& pt, const vec& trans) const
  {
    const double d0 = pt(0) - (mu(0) - trans(0)), d1 = pt(1) - (mu(1) - trans(1));
    return std::exp(-0.5 * ((d0 * cov_inv(0, 0)) * d0 + (d1 * cov_inv(1, 1)) * d1));
  }
  vec mu, sigmas;
  mat cov, cov_inv;
};
typedef std::vector<Gaussian> GaussianList;

class Target
{
public:
  Target() {}
  explicit Target(const GaussianList& gaussians) : gaussians_(gaussians) {}
  void addGaussian(const Gaussian& g) { gaussians_.emplace_back(g); }
  void deleteGaussian(unsigned int idx) { gaussians_.erase(gaussians_.begin() + idx); }
  double evaluate(const vec& pt, const vec& trans) const
  {
    double val = 0.0;
    for (const auto& g : gaussians_) val += g(pt, trans);
    return val;
  }
  // target on every column of phi_grid (2 x P), normalised to sum 1
  vec fill(const vec& trans, const mat& phi_grid) const
  {
    std::vector<double> mu, sg;
    flatten(mu, sg);
    vec out(phi_grid.n_cols());
    throw_on_error(eea_target_fill(device_ordinal(), static_cast<unsigned>(gaussians_.size()), mu.data(),
                                   sg.data(), trans.memptr(), phi_grid.memptr(),
                                   static_cast<unsigned>(phi_grid.n_cols()), out.memptr()));
    return out;
  }
  const GaussianList& gaussians() const { return gaussians_; }
  void flatten(std::vector<double>& mu, std::vector<double>& sigma) const
  {
    mu.clear();
    sigma.clear();
    for (const auto& g : gaussians_) {
      mu.push_back(g.mu(0));
      mu.push_back(g.mu(1));
      sigma.push_back(g.sigmas(0));
      sigma.push_back(g.sigmas(1));
    }
  }

private:
  GaussianList gaussians_;
};
}  // namespace ergodic_exploration

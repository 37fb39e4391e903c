// Basis: 2-D cosine basis (reference basis.hpp / basis.cpp).  The two hot loops,
// trajCoeff (c_k) and spatialCoeff (phi_k), run on the device through the C ABI; the
// single-point evaluations are host helpers.
#pragma once

#include <cmath>

#include <ergodic_exploration/device.hpp>
#include <ergodic_exploration/numerics.hpp>

namespace ergodic_exploration
{
class Basis
{
  template <class ModelT>
  friend class ErgodicControl;

public:
  Basis(double lx, double ly, unsigned int num_basis)
    : lx_(lx), ly_(ly), num_basis_(num_basis), total_basis_(num_basis * num_basis), lamdak_(total_basis_)
  {
    // mode col = k2 * K + k1 (x mode fastest); lambda_k = (1 + |k|)^-1.5, no h_k normalisation
    for (unsigned int k2 = 0, col = 0; k2 < num_basis; k2++) {
      for (unsigned int k1 = 0; k1 < num_basis; k1++, col++) {
        lamdak_(col) = 1.0 / std::pow(1.0 + std::sqrt(static_cast<double>(k1 * k1 + k2 * k2)), 1.5);
      }
    }
  }

  vec fourierBasis(const vec& x) const
  {
    vec fk(total_basis_);
    for (unsigned int k2 = 0, col = 0; k2 < num_basis_; k2++) {
      for (unsigned int k1 = 0; k1 < num_basis_; k1++, col++) {
        fk(col) = std::cos(k1 * (PI / lx_) * x(0)) * std::cos(k2 * (PI / ly_) * x(1));
      }
    }
    return fk;
  }

  mat gradFourierBasis(const vec& x) const
  {
    mat dfk(2, total_basis_);
    for (unsigned int k2 = 0, col = 0; k2 < num_basis_; k2++) {
      for (unsigned int k1 = 0; k1 < num_basis_; k1++, col++) {
        const double a = k1 * (PI / lx_), b = k2 * (PI / ly_);
        dfk(0, col) = -a * std::sin(a * x(0)) * std::cos(b * x(1));
        dfk(1, col) = -b * std::cos(a * x(0)) * std::sin(b * x(1));
      }
    }
    return dfk;
  }

  // c_k of a trajectory (rows 0,1 = x,y); device: eea_basis_traj_coeff
  vec trajCoeff(const mat& xt) const
  {
    vec ck(total_basis_);
    throw_on_error(eea_basis_traj_coeff(device_ordinal(), lx_, ly_, num_basis_, xt.memptr(),
                                        static_cast<unsigned>(xt.n_rows()), static_cast<unsigned>(xt.n_cols()),
                                        ck.memptr()));
    return ck;
  }

  // phi_k of target values on a point list (2 x P); device: eea_basis_spatial_coeff
  vec spatialCoeff(const vec& phi_vals, const mat& phi_grid) const
  {
    vec phik(total_basis_);
    throw_on_error(eea_basis_spatial_coeff(device_ordinal(), lx_, ly_, num_basis_, phi_vals.memptr(),
                                           phi_grid.memptr(), static_cast<unsigned>(phi_grid.n_cols()),
                                           phik.memptr()));
    return phik;
  }

  const vec& lamdak() const { return lamdak_; }

private:
  double lx_, ly_;
  unsigned int num_basis_, total_basis_;
  vec lamdak_;
};
}  // namespace ergodic_exploration

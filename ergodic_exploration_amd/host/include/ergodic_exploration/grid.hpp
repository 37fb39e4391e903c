// GridMap: int8 occupancy grid, row-major, world <-> grid index math (reference grid.hpp /
// grid.cpp).  Host-side value type; the batched collision kernels take its buffer and geometry.
#pragma once

#include <cmath>
#include <cstdint>
#include <iostream>
#include <memory>
#include <stdexcept>
#include <vector>

namespace ergodic_exploration
{
typedef std::vector<int8_t> GridData;

// conversion of a double to unsigned as the reference's x86-64 build performs it
// (64-bit truncation, low 32 bits kept): negative values wrap instead of being undefined
inline unsigned int to_index(double v)
{
  if (!(v > -9.2233720368547758e18 && v < 9.2233720368547758e18)) return 0u;
  return static_cast<unsigned int>(static_cast<unsigned long long>(static_cast<long long>(v)));
}

inline unsigned int axis_length(double lower, double upper, double resolution)
{
  return to_index(std::round((upper - lower) / resolution));
}

inline double axis_upper(double lower, double resolution, unsigned int size)
{
  return static_cast<double>(resolution * size) + lower;
}

class GridMap
{
public:
  GridMap(double xmin, double xmax, double ymin, double ymax, double resolution, const GridData& grid_data)
    : xsize_(axis_length(xmin, xmax, resolution))
    , ysize_(axis_length(ymin, ymax, resolution))
    , resolution_(resolution)
    , xmin_(xmin)
    , ymin_(ymin)
    , xmax_(xmax)
    , ymax_(ymax)
    , grid_data_(grid_data)
  {
    if (xsize_ * ysize_ != grid_data_.size()) throw std::invalid_argument("Grid data size does not match the grid size");
  }
  // from occupancy-grid message fields (nav_msgs::OccupancyGrid: width, height, resolution, origin)
  static GridMap fromOccupancyGrid(unsigned int width, unsigned int height, double resolution, double origin_x,
                                   double origin_y, const GridData& grid_data)
  {
    GridMap g;
    g.xsize_ = width;
    g.ysize_ = height;
    g.resolution_ = resolution;
    g.xmin_ = origin_x;
    g.ymin_ = origin_y;
    g.xmax_ = axis_upper(origin_x, resolution, width);
    g.ymax_ = axis_upper(origin_y, resolution, height);
    g.grid_data_ = grid_data;
    if (g.xsize_ * g.ysize_ != g.grid_data_.size()) throw std::invalid_argument("Grid data size does not match the grid size");
    return g;
  }
  GridMap() : xsize_(0), ysize_(0), resolution_(0), xmin_(0.0), ymin_(0.0), xmax_(0.0), ymax_(0.0) {}

  bool gridBounds(unsigned int i, unsigned int j) const { return (i <= ysize_ - 1) && (j <= xsize_ - 1); }
  bool gridBounds(unsigned int idx) const { return idx <= (xsize_ * ysize_ - 1); }
  unsigned int grid2RowMajor(unsigned int i, unsigned int j) const
  {
    if (!gridBounds(i, j)) std::cout << "WARNING (grid2RowMajor) i and j NOT within bounds" << std::endl;
    return i * xsize_ + j;
  }
  std::vector<unsigned int> rowMajor2Grid(unsigned int idx) const
  {
    const unsigned int i = idx / xsize_;
    return { i, idx - i * xsize_ };
  }
  std::vector<double> grid2World(unsigned int i, unsigned int j) const
  {
    return { static_cast<double>(j * resolution_) + resolution_ / 2.0 + xmin_,
             static_cast<double>(i * resolution_) + resolution_ / 2.0 + ymin_ };
  }
  std::vector<double> grid2World(unsigned int idx) const
  {
    const auto ij = rowMajor2Grid(idx);
    return grid2World(ij.at(0), ij.at(1));
  }
  std::vector<unsigned int> world2Grid(double x, double y) const
  {
    unsigned int j = to_index(std::floor((x - xmin_) / resolution_));
    unsigned int i = to_index(std::floor((y - ymin_) / resolution_));
    if (j == xsize_) j--;
    if (i == ysize_) i--;
    return { i, j };
  }
  unsigned int world2RowMajor(double x, double y) const
  {
    const auto ij = world2Grid(x, y);
    return grid2RowMajor(ij.at(0), ij.at(1));
  }
  double getCell(double x, double y) const { return getCell(world2RowMajor(x, y)); }
  double getCell(unsigned int i, unsigned int j) const { return getCell(grid2RowMajor(i, j)); }
  double getCell(unsigned int idx) const
  {
    if (!gridBounds(idx)) throw std::invalid_argument("Grid index out of range");
    return static_cast<double>(grid_data_.at(idx)) / 100.0;
  }

  const GridData& gridData() const { return grid_data_; }
  // Opaque slot for a device copy of the cells, filled on first use by the classes that run on the
  // device (collision.hpp).  The map is immutable after construction, so copies of a GridMap share it.
  std::shared_ptr<const void>& deviceCache() const { return device_cache_; }
  double resolution() const { return resolution_; }
  double xmin() const { return xmin_; }
  double ymin() const { return ymin_; }
  double xmax() const { return xmax_; }
  double ymax() const { return ymax_; }
  unsigned int xsize() const { return xsize_; }
  unsigned int ysize() const { return ysize_; }
  unsigned int size() const { return xsize_ * ysize_; }

private:
  unsigned int xsize_, ysize_;
  double resolution_, xmin_, ymin_, xmax_, ymax_;
  GridData grid_data_;
  mutable std::shared_ptr<const void> device_cache_;
};
}  // namespace ergodic_exploration

// Map and replay files of the non-ROS entry points.
//
// 1. map_server maps (the format of the reference's maps/maze.yaml + maze.pgm): a yaml with
//    image / resolution / origin / negate / occupied_thresh / free_thresh and a PGM image (P5 or P2).
//    Conversion to occupancy cells follows the published map_server rule (trinary mode): with
//    p = (255 - pixel) / 255 (pixel / 255 if negate), p > occupied_thresh -> 100,
//    p < free_thresh -> 0, else -1; image row 0 is the TOP of the map, OccupancyGrid row 0 the bottom.
//    The result is the nav_msgs::OccupancyGrid wire layout GridMap consumes (reference grid.cpp:63-94):
//    width, height, resolution, origin, int8 data row-major with x fastest.
//
// 2. replay logs: the inputs of Exploration::control's loop (exploration.hpp:197-292) as the node
//    receives them -- map messages, and per tick the pose (tf) and body twist (odom) -- plus the
//    twist the loop published, so a recorded run can be fed through the state machine again and
//    compared tick by tick.  Text, one record per line:
//        # comment
//        map <width> <height> <resolution> <origin_x> <origin_y> <n_runs> <value> <count> ...   (run-length)
//        tick <t> <x> <y> <theta> <vbx> <vby> <vbw> <ux> <uy> <uw> <source>
//    Doubles are written with 17 significant digits (round trip exact).
#pragma once

#include <cstdint>
#include <cstdio>
#include <fstream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include <ergodic_exploration/grid.hpp>

#include "params.hpp"

namespace map_io
{
namespace ee = ergodic_exploration;

struct Occupancy
{
  unsigned int width = 0, height = 0;
  double resolution = 0.0, origin_x = 0.0, origin_y = 0.0;
  ee::GridData data;  // int8, row-major, x fastest, row 0 = lowest y

  ee::GridMap grid() const { return ee::GridMap::fromOccupancyGrid(width, height, resolution, origin_x, origin_y, data); }
};

inline std::string dirname_of(const std::string& path)
{
  const auto slash = path.find_last_of('/');
  return slash == std::string::npos ? std::string(".") : path.substr(0, slash);
}

// next whitespace-separated token of a PGM header, skipping '#' comments
inline std::string pgm_token(std::istream& in)
{
  std::string tok;
  for (;;) {
    const int c = in.get();
    if (c == EOF) break;
    if (c == '#') {
      std::string rest;
      std::getline(in, rest);
      continue;
    }
    if (c == ' ' || c == '\t' || c == '\n' || c == '\r') {
      if (!tok.empty()) break;
      continue;
    }
    tok.push_back(static_cast<char>(c));
  }
  return tok;
}

// grey image, row 0 = top; values scaled to 0..255
inline void read_pgm(const std::string& path, unsigned int& w, unsigned int& h, std::vector<unsigned char>& pix)
{
  std::ifstream in(path, std::ios::binary);
  if (!in) throw std::runtime_error("cannot open map image " + path);
  const std::string magic = pgm_token(in);
  if (magic != "P5" && magic != "P2") throw std::runtime_error("map image is not a PGM (P5/P2): " + path);
  w = static_cast<unsigned int>(std::stoul(pgm_token(in)));
  h = static_cast<unsigned int>(std::stoul(pgm_token(in)));
  const unsigned int maxval = static_cast<unsigned int>(std::stoul(pgm_token(in)));
  if (w == 0 || h == 0 || maxval == 0 || maxval > 65535) throw std::runtime_error("bad PGM header: " + path);
  pix.resize(static_cast<std::size_t>(w) * h);
  for (std::size_t i = 0; i < pix.size(); ++i) {
    unsigned int v;
    if (magic == "P5") {
      // pgm_token consumed exactly one whitespace byte after maxval: the raster starts here
      int c0 = in.get();
      if (c0 == EOF) throw std::runtime_error("truncated PGM: " + path);
      v = static_cast<unsigned int>(c0);
      if (maxval > 255) {
        const int c1 = in.get();
        if (c1 == EOF) throw std::runtime_error("truncated PGM: " + path);
        v = (v << 8) | static_cast<unsigned int>(c1);
      }
    } else {
      v = static_cast<unsigned int>(std::stoul(pgm_token(in)));
    }
    pix[i] = static_cast<unsigned char>(maxval == 255 ? v : (v * 255u + maxval / 2) / maxval);
  }
}

// map_server yaml + image -> occupancy cells
inline Occupancy load_map_yaml(const std::string& yaml_path)
{
  params::Store y;
  y.load(yaml_path);
  std::string image = y.param("image", std::string());
  if (image.empty()) throw std::runtime_error("map yaml has no image: " + yaml_path);
  if (image.front() != '/') image = dirname_of(yaml_path) + "/" + image;
  const auto origin = y.numbers("origin", { 0.0, 0.0, 0.0 });
  const bool negate = y.param("negate", 0.0) != 0.0;
  const double occ_th = y.param("occupied_thresh", 0.65), free_th = y.param("free_thresh", 0.196);

  Occupancy m;
  std::vector<unsigned char> pix;
  read_pgm(image, m.width, m.height, pix);
  m.resolution = y.param("resolution", 0.05);
  m.origin_x = origin.size() > 0 ? origin[0] : 0.0;
  m.origin_y = origin.size() > 1 ? origin[1] : 0.0;
  m.data.resize(pix.size());
  for (unsigned int r = 0; r < m.height; ++r) {
    for (unsigned int c = 0; c < m.width; ++c) {
      const unsigned char v = pix[static_cast<std::size_t>(r) * m.width + c];
      const double p = negate ? v / 255.0 : (255 - v) / 255.0;
      const int8_t cell = p > occ_th ? 100 : (p < free_th ? 0 : -1);
      m.data[static_cast<std::size_t>(m.height - 1 - r) * m.width + c] = cell;  // image top row = highest y
    }
  }
  return m;
}

// ---- replay logs -------------------------------------------------------------------------
struct Tick
{
  int t = 0;
  double pose[3] = { 0, 0, 0 }, vb[3] = { 0, 0, 0 }, u[3] = { 0, 0, 0 };
  std::string source;
};
struct Record
{
  bool is_map = false;
  Occupancy map;
  Tick tick;
};

inline void write_map(std::FILE* f, const Occupancy& m)
{
  std::vector<std::pair<int, std::size_t>> runs;
  for (const int8_t v : m.data) {
    if (!runs.empty() && runs.back().first == v) runs.back().second++;
    else runs.emplace_back(v, 1);
  }
  std::fprintf(f, "map %u %u %.17g %.17g %.17g %zu", m.width, m.height, m.resolution, m.origin_x, m.origin_y,
               runs.size());
  for (const auto& r : runs) std::fprintf(f, " %d %zu", r.first, r.second);
  std::fprintf(f, "\n");
}

inline void write_tick(std::FILE* f, const Tick& k)
{
  std::fprintf(f, "tick %d %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %s\n", k.t, k.pose[0], k.pose[1],
               k.pose[2], k.vb[0], k.vb[1], k.vb[2], k.u[0], k.u[1], k.u[2], k.source.c_str());
}

inline std::vector<Record> read_replay(const std::string& path)
{
  std::ifstream in(path);
  if (!in) throw std::runtime_error("cannot open replay file " + path);
  std::vector<Record> out;
  std::string line;
  while (std::getline(in, line)) {
    if (line.empty() || line[0] == '#') continue;
    std::istringstream is(line);
    std::string kind;
    is >> kind;
    Record r;
    if (kind == "map") {
      r.is_map = true;
      std::size_t n_runs = 0;
      is >> r.map.width >> r.map.height >> r.map.resolution >> r.map.origin_x >> r.map.origin_y >> n_runs;
      r.map.data.reserve(static_cast<std::size_t>(r.map.width) * r.map.height);
      for (std::size_t i = 0; i < n_runs; ++i) {
        int v;
        std::size_t count;
        if (!(is >> v >> count)) throw std::runtime_error("bad map record in " + path);
        r.map.data.insert(r.map.data.end(), count, static_cast<int8_t>(v));
      }
      if (r.map.data.size() != static_cast<std::size_t>(r.map.width) * r.map.height) {
        throw std::runtime_error("map record does not fill width x height in " + path);
      }
    } else if (kind == "tick") {
      Tick& k = r.tick;
      if (!(is >> k.t >> k.pose[0] >> k.pose[1] >> k.pose[2] >> k.vb[0] >> k.vb[1] >> k.vb[2])) {
        throw std::runtime_error("bad tick record in " + path);
      }
      // the published twist and its source are optional (logs of the inputs alone can be replayed)
      if (is >> k.u[0] >> k.u[1] >> k.u[2]) is >> k.source;
      else k.source = "?";
    } else {
      throw std::runtime_error("unknown record '" + kind + "' in " + path);
    }
    out.push_back(std::move(r));
  }
  return out;
}
}  // namespace map_io

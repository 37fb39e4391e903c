// Parameter loading for the non-ROS entry points: reads the reference's yaml parameter files
// (config/explore_omni.yaml, config/explore_cart.yaml: flat "name: value" lines, numeric lists and
// lists of lists) and answers param(name, default) like ros::NodeHandle::param does.
#pragma once

#include <cstdlib>
#include <fstream>
#include <map>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace params
{
class Store
{
public:
  void load(const std::string& path)
  {
    std::ifstream in(path);
    if (!in) throw std::runtime_error("cannot open parameter file " + path);
    std::string line;
    while (std::getline(in, line)) {
      const auto hash = line.find('#');
      if (hash != std::string::npos) line.erase(hash);
      const auto colon = line.find(':');
      if (colon == std::string::npos) continue;
      std::string key = trim(line.substr(0, colon)), val = trim(line.substr(colon + 1));
      if (key.empty() || val.empty()) continue;
      values_[key] = val;
    }
  }
  void set(const std::string& key, const std::string& val) { values_[key] = val; }
  bool has(const std::string& key) const { return values_.count(key) != 0; }
  double param(const std::string& key, double def) const
  {
    const auto it = values_.find(key);
    return it == values_.end() ? def : std::strtod(it->second.c_str(), nullptr);
  }
  std::string param(const std::string& key, const std::string& def) const
  {
    const auto it = values_.find(key);
    if (it == values_.end()) return def;
    std::string v = it->second;
    if (v.size() >= 2 && (v.front() == '"' || v.front() == '\'')) v = v.substr(1, v.size() - 2);
    return v;
  }
  // every number in the value, in order: "[[2.5, 2.5], [8.5, 2.5]]" -> 2.5 2.5 8.5 2.5
  std::vector<double> numbers(const std::string& key, const std::vector<double>& def) const
  {
    const auto it = values_.find(key);
    if (it == values_.end()) return def;
    std::string s = it->second;
    for (char& c : s)
      if (c == '[' || c == ']' || c == ',') c = ' ';
    std::istringstream is(s);
    std::vector<double> out;
    double v;
    while (is >> v) out.push_back(v);
    return out;
  }

private:
  static std::string trim(const std::string& s)
  {
    const auto b = s.find_first_not_of(" \t\r\n"), e = s.find_last_not_of(" \t\r\n");
    return b == std::string::npos ? std::string() : s.substr(b, e - b + 1);
  }
  std::map<std::string, std::string> values_;
};
}  // namespace params

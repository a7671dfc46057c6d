// config_loader.hpp — reads parameter files laid out like the reference's config/ directory (config/multirotor_simulator.yaml,
// config/uavs.yaml, config/uavs/<type>.yaml, config/controllers/*.yaml) and assembles what the UavSystemRos constructor
// assembles from them with mrs_lib::ParamLoader (src/uav_system_ros.cpp:27-157) and what MultirotorSimulator::onInit reads
// (src/multirotor_simulator.cpp:107-157) — without ROS and without yaml-cpp.
//
// The parser covers the YAML subset those files use: block maps by indentation, flow maps `{a: 1, b: 2}`, flow lists
// `[ ... ]` over several lines with trailing commas, quoted and plain scalars, `#` comments.  Every scalar ends up under its
// slash-joined path ("x500/propulsion/rpm/min"); a list becomes the vector of its scalars under its path.  Later files
// override earlier ones key by key, like the custom-config layering of the launch files.
#pragma once
#include <cctype>
#include <cstdlib>
#include <fstream>
#include <map>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "multirotor_simulator.hpp"

namespace mrs_multirotor_simulator {

class ParamTree {
public:
  void loadFile(const std::string& path) {
    std::ifstream f(path);
    if (!f) throw std::runtime_error("config_loader: cannot open " + path);
    std::stringstream ss;
    ss << f.rdbuf();
    loadText(ss.str());
  }

  void loadText(const std::string& text) {
    // 1. physical lines -> logical lines (comments stripped, flow collections joined), with their indentation
    std::vector<std::pair<int, std::string>> lines;
    std::string                              pending;
    int                                      pending_indent = 0, depth = 0;
    std::istringstream                       in(text);
    std::string                              raw;
    while (std::getline(in, raw)) {
      const std::string ln = stripComment(raw);
      if (depth == 0) {
        if (trim(ln).empty()) continue;
        pending_indent = (int)ln.find_first_not_of(' ');
        pending        = trim(ln);
      } else {
        pending += " " + trim(ln);
      }
      depth = flowDepth(pending);
      if (depth < 0) throw std::runtime_error("config_loader: unbalanced brackets near: " + pending);
      if (depth == 0) {
        lines.emplace_back(pending_indent, pending);
        pending.clear();
      }
    }
    if (depth != 0) throw std::runtime_error("config_loader: unterminated flow collection: " + pending);
    // 2. indentation -> paths
    std::vector<std::pair<int, std::string>> stack;  // (indent, key)
    for (auto& [indent, ln] : lines) {
      while (!stack.empty() && stack.back().first >= indent) stack.pop_back();
      if (ln[0] == '-') {  // block list item under the current key
        std::string path = joined(stack);
        values_[path].push_back(unquote(trim(ln.substr(1))));
        continue;
      }
      const size_t colon = findColon(ln);
      if (colon == std::string::npos) throw std::runtime_error("config_loader: expected `key: value`, got: " + ln);
      const std::string key = unquote(trim(ln.substr(0, colon)));
      const std::string val = trim(ln.substr(colon + 1));
      stack.emplace_back(indent, key);
      const std::string path = joined(stack);
      if (val.empty()) {
        values_.erase(path);  // a map (or block list) follows; a scalar of an earlier file under this path is replaced
        continue;
      }
      assign(path, val);
    }
  }

  bool has(const std::string& path) const { return values_.count(path) && !values_.at(path).empty(); }

  std::string getString(const std::string& path) const {
    auto it = values_.find(path);
    if (it == values_.end() || it->second.size() != 1) throw std::runtime_error("config_loader: missing scalar parameter '" + path + "'");
    return it->second[0];
  }
  double getDouble(const std::string& path) const { return toDouble(getString(path), path); }
  double getDouble(const std::string& path, double fallback) const { return has(path) ? getDouble(path) : fallback; }
  int    getInt(const std::string& path) const { return (int)toDouble(getString(path), path); }
  bool   getBool(const std::string& path) const {
    std::string v = getString(path);
    for (auto& c : v) c = (char)std::tolower((unsigned char)c);
    if (v == "true" || v == "yes" || v == "on" || v == "1") return true;
    if (v == "false" || v == "no" || v == "off" || v == "0") return false;
    throw std::runtime_error("config_loader: '" + path + "' is not a boolean: " + v);
  }
  bool getBool(const std::string& path, bool fallback) const { return has(path) ? getBool(path) : fallback; }
  std::vector<std::string> getList(const std::string& path) const {
    auto it = values_.find(path);
    if (it == values_.end()) throw std::runtime_error("config_loader: missing list parameter '" + path + "'");
    return it->second;
  }
  std::vector<double> getDoubleList(const std::string& path) const {
    std::vector<double> out;
    for (auto& s : getList(path)) out.push_back(toDouble(s, path));
    return out;
  }

private:
  std::map<std::string, std::vector<std::string>> values_;

  static std::string trim(const std::string& s) {
    const size_t a = s.find_first_not_of(" \t\r\n");
    if (a == std::string::npos) return "";
    const size_t b = s.find_last_not_of(" \t\r\n");
    return s.substr(a, b - a + 1);
  }
  static std::string stripComment(const std::string& s) {
    char quote = 0;
    for (size_t i = 0; i < s.size(); i++) {
      const char c = s[i];
      if (quote) {
        if (c == quote) quote = 0;
      } else if (c == '"' || c == '\'') {
        quote = c;
      } else if (c == '#' && (i == 0 || s[i - 1] == ' ' || s[i - 1] == '\t')) {
        return s.substr(0, i);
      }
    }
    return s;
  }
  static int flowDepth(const std::string& s) {
    int  d = 0;
    char quote = 0;
    for (char c : s) {
      if (quote) {
        if (c == quote) quote = 0;
      } else if (c == '"' || c == '\'') {
        quote = c;
      } else if (c == '[' || c == '{') {
        d++;
      } else if (c == ']' || c == '}') {
        d--;
      }
    }
    return d;
  }
  static size_t findColon(const std::string& s) {  // the first `:` outside quotes/brackets that ends a key
    char quote = 0;
    int  d     = 0;
    for (size_t i = 0; i < s.size(); i++) {
      const char c = s[i];
      if (quote) {
        if (c == quote) quote = 0;
      } else if (c == '"' || c == '\'') {
        quote = c;
      } else if (c == '[' || c == '{') {
        d++;
      } else if (c == ']' || c == '}') {
        d--;
      } else if (c == ':' && d == 0 && (i + 1 == s.size() || s[i + 1] == ' ')) {
        return i;
      }
    }
    return std::string::npos;
  }
  static std::string unquote(const std::string& s) {
    if (s.size() >= 2 && ((s.front() == '"' && s.back() == '"') || (s.front() == '\'' && s.back() == '\''))) return s.substr(1, s.size() - 2);
    return s;
  }
  static std::vector<std::string> splitTop(const std::string& s) {  // split at top-level commas
    std::vector<std::string> out;
    std::string              cur;
    char                     quote = 0;
    int                      d     = 0;
    for (char c : s) {
      if (quote) {
        if (c == quote) quote = 0;
        cur += c;
      } else if (c == '"' || c == '\'') {
        quote = c;
        cur += c;
      } else if (c == '[' || c == '{') {
        d++;
        cur += c;
      } else if (c == ']' || c == '}') {
        d--;
        cur += c;
      } else if (c == ',' && d == 0) {
        out.push_back(trim(cur));
        cur.clear();
      } else {
        cur += c;
      }
    }
    if (!trim(cur).empty()) out.push_back(trim(cur));
    return out;
  }
  static double toDouble(const std::string& s, const std::string& path) {
    char*        end = nullptr;
    const double v   = std::strtod(s.c_str(), &end);
    if (end == s.c_str() || *end != '\0') throw std::runtime_error("config_loader: '" + path + "' is not a number: " + s);
    return v;
  }
  static std::string joined(const std::vector<std::pair<int, std::string>>& stack) {
    std::string p;
    for (auto& e : stack) p += (p.empty() ? "" : "/") + e.second;
    return p;
  }
  void assign(const std::string& path, const std::string& val) {
    if (val.front() == '[') {  // flow list (nested lists are flattened, like loadMatrixDynamic2's row-major reading)
      std::vector<std::string> items;
      flattenList(val, items);
      values_[path] = items;
    } else if (val.front() == '{') {  // flow map
      for (auto& kv : splitTop(val.substr(1, val.size() - 2))) {
        const size_t c = findColon(kv);
        if (c == std::string::npos) throw std::runtime_error("config_loader: bad flow-map entry: " + kv);
        assign(path + "/" + unquote(trim(kv.substr(0, c))), trim(kv.substr(c + 1)));
      }
    } else {
      values_[path] = {unquote(val)};
    }
  }
  static void flattenList(const std::string& val, std::vector<std::string>& out) {
    for (auto& it : splitTop(val.substr(1, val.size() - 2))) {
      if (it.empty()) continue;
      if (it.front() == '[')
        flattenList(it, out);
      else
        out.push_back(unquote(it));
    }
  }
};

// ModelParams exactly as the UavSystemRos constructor assembles them (src/uav_system_ros.cpp:51-70, 96-103): the airframe block
// of `type`, g / ground / take-off patch from the simulator file, inertia from calculateInertia, allocation rows scaled.
inline MultirotorModel::ModelParams modelParamsFromConfig(const ParamTree& cfg, const std::string& type) {
  mrs_model_params_t c;
  mrs_model_params_default(&c);
  c.n_motors              = cfg.getInt(type + "/n_motors");
  c.g                     = cfg.getDouble("g", 9.81);
  c.mass                  = cfg.getDouble(type + "/mass");
  c.arm_length            = cfg.getDouble(type + "/arm_length");
  c.body_height           = cfg.getDouble(type + "/body_height");
  c.air_resistance_coeff  = cfg.getDouble(type + "/air_resistance_coeff");
  c.motor_time_constant   = cfg.getDouble(type + "/motor_time_constant");
  c.prop_radius           = cfg.getDouble(type + "/propulsion/prop_radius");
  c.kf                    = cfg.getDouble(type + "/propulsion/force_constant");
  c.km                    = cfg.getDouble(type + "/propulsion/moment_constant");
  c.min_rpm               = cfg.getDouble(type + "/propulsion/rpm/min");
  c.max_rpm               = cfg.getDouble(type + "/propulsion/rpm/max");
  c.ground_enabled        = cfg.getBool("ground/enabled", false) ? 1 : 0;
  c.ground_z              = cfg.getDouble("ground/z", 0.0);
  c.takeoff_patch_enabled = cfg.getBool("individual_takeoff_platform/enabled", false) ? 1 : 0;
  const std::vector<double> flat = cfg.getDoubleList(type + "/propulsion/allocation_matrix");
  const int                 n    = c.n_motors;
  if (n < 1 || n > MRS_MAX_MOTORS || (int)flat.size() != 4 * n)
    throw std::runtime_error("config_loader: " + type + "/propulsion/allocation_matrix must have 4 x n_motors entries");
  for (int i = 0; i < 4 * MRS_MAX_MOTORS; i++) c.allocation_matrix[i] = 0.0;
  for (int r = 0; r < 4; r++)
    for (int m = 0; m < n; m++) c.allocation_matrix[r * MRS_MAX_MOTORS + m] = flat[(size_t)r * n + m];  // loadMatrixDynamic2: row-major 4 x n
  mrs_throw_on_error(mrs_calculate_inertia(&c));   // src/uav_system_ros.cpp:664-671
  mrs_throw_on_error(mrs_scale_allocation(&c));    // :98-103
  MultirotorModel::ModelParams p;
  p.fromC(c);
  return p;
}

// config/multirotor_simulator.yaml -> SimulatorConfig (src/multirotor_simulator.cpp:107-131, src/uav_system_ros.cpp:52-53)
inline SimulatorConfig simulatorConfigFromTree(const ParamTree& cfg) {
  SimulatorConfig s;
  s.simulation_rate       = cfg.getDouble("simulation_rate", s.simulation_rate);
  s.clock_rate            = cfg.getDouble("clock_rate", s.clock_rate);
  s.realtime_factor       = cfg.getDouble("realtime_factor", s.realtime_factor);
  s.collisions_enabled    = cfg.getBool("collisions/enabled", s.collisions_enabled);
  s.collisions_crash      = cfg.getBool("collisions/crash", s.collisions_crash);
  s.collisions_rebounce   = cfg.getDouble("collisions/rebounce", s.collisions_rebounce);
  s.iterate_without_input = cfg.getBool("iterate_without_input", s.iterate_without_input);
  s.input_timeout         = cfg.getDouble("input_timeout", s.input_timeout);
  return s;
}

struct UavSpawn {
  std::string name, type;
  double      x, y, z, heading;
};

// uav_names + per-UAV type / spawn blocks (config/uavs.yaml; src/multirotor_simulator.cpp:150-157, src/uav_system_ros.cpp:76-94),
// spawn randomisation included when `randomization/enabled`
inline std::vector<UavSpawn> uavSpawnsFromConfig(const ParamTree& cfg) {
  std::vector<UavSpawn> out;
  const bool            randomize = cfg.getBool("randomization/enabled", false);
  for (const std::string& name : cfg.getList("uav_names")) {
    UavSpawn u{name, cfg.getString(name + "/type"), cfg.getDouble(name + "/spawn/x"), cfg.getDouble(name + "/spawn/y"),
               cfg.getDouble(name + "/spawn/z"), cfg.getDouble(name + "/spawn/heading")};
    if (randomize)
      randomizeSpawn(cfg.getDouble("randomization/bounds/x"), cfg.getDouble("randomization/bounds/y"), cfg.getDouble("randomization/bounds/z"), u.x, u.y,
                     u.z, u.heading);
    out.push_back(u);
  }
  return out;
}

// The loop of MultirotorSimulator::onInit + the UavSystemRos constructor for every UAV of the configuration: constructs the
// swarm (consecutive UAVs of one type go down as one call), applies the controller blocks, ends with the two warm-up steps.
inline void constructSwarmFromConfig(UavSwarm& swarm, const ParamTree& cfg, const std::vector<UavSpawn>& uavs) {
  if ((int)uavs.size() != swarm.size()) throw std::runtime_error("config_loader: swarm size does not match uav_names");
  size_t i = 0;
  while (i < uavs.size()) {
    size_t e = i;
    while (e < uavs.size() && uavs[e].type == uavs[i].type) e++;
    const MultirotorModel::ModelParams mp = modelParamsFromConfig(cfg, uavs[i].type);
    std::vector<Eigen::Vector3d>       pos;
    std::vector<double>                hdg;
    for (size_t k = i; k < e; k++) {
      pos.push_back(Eigen::Vector3d(uavs[k].x, uavs[k].y, uavs[k].z));
      hdg.push_back(uavs[k].heading);
    }
    swarm.construct((int)i, (int)(e - i), mp, pos, hdg);
    i = e;
  }
  const int          n = swarm.size();
  mrs_mixer_params_t mx{};
  mx.desaturation = cfg.getBool("mixer/desaturation", true) ? 1 : 0;
  mrs_throw_on_error(mrs_swarm_set_mixer_params(swarm.handle(), 0, n, &mx));
  mrs_rate_params_t rc{cfg.getDouble("rate_controller/kp", 4.0), cfg.getDouble("rate_controller/kd", 0.04), cfg.getDouble("rate_controller/ki", 0.0)};
  mrs_throw_on_error(mrs_swarm_set_rate_params(swarm.handle(), 0, n, &rc));
  mrs_attitude_params_t ac{cfg.getDouble("attitude_controller/kp", 6.0), cfg.getDouble("attitude_controller/kd", 0.05),
                           cfg.getDouble("attitude_controller/ki", 0.01), cfg.getDouble("attitude_controller/max_rate_roll_pitch", 10.0),
                           cfg.getDouble("attitude_controller/max_rate_yaw", 1.0)};
  mrs_throw_on_error(mrs_swarm_set_attitude_params(swarm.handle(), 0, n, &ac));
  mrs_velocity_params_t vc{cfg.getDouble("velocity_controller/kp", 2.0), cfg.getDouble("velocity_controller/kd", 0.05),
                           cfg.getDouble("velocity_controller/ki", 0.01), cfg.getDouble("velocity_controller/max_acceleration", 4.0)};
  mrs_throw_on_error(mrs_swarm_set_velocity_params(swarm.handle(), 0, n, &vc));
  mrs_position_params_t pc{cfg.getDouble("position_controller/kp", 2.0), cfg.getDouble("position_controller/kd", 0.15),
                           cfg.getDouble("position_controller/ki", 0.2), cfg.getDouble("position_controller/max_velocity", 6.0)};
  mrs_throw_on_error(mrs_swarm_set_position_params(swarm.handle(), 0, n, &pc));
  swarm.warmUp();  // src/uav_system_ros.cpp:223-232
}

}  // namespace mrs_multirotor_simulator

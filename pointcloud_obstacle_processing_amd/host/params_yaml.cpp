#include "params_yaml.hpp"

#include <cmath>
#include <cstdlib>
#include <fstream>
#include <sstream>
#include <vector>

namespace pcop_host {

namespace {

std::string trim(const std::string& s) {
    size_t b = s.find_first_not_of(" \t\r\n");
    if (b == std::string::npos) return "";
    size_t e = s.find_last_not_of(" \t\r\n");
    return s.substr(b, e - b + 1);
}

}  // namespace

std::map<std::string, std::string> load_yaml_scalars(const std::string& path, std::string* error) {
    std::map<std::string, std::string> out;
    std::ifstream f(path);
    if (!f) {
        if (error) *error = "cannot open " + path;
        return out;
    }
    std::vector<std::pair<int, std::string>> stack;   // (indent, key) of the open maps
    std::string line;
    while (std::getline(f, line)) {
        const size_t hash = line.find('#');
        if (hash != std::string::npos) line = line.substr(0, hash);
        if (trim(line).empty()) continue;
        int indent = 0;
        for (char ch : line) {
            if (ch == ' ') ++indent;
            else if (ch == '\t') indent += 8;
            else break;
        }
        const std::string body = trim(line);
        const size_t colon = body.find(':');
        if (colon == std::string::npos) continue;
        const std::string key = trim(body.substr(0, colon));
        std::string val = trim(body.substr(colon + 1));
        while (!stack.empty() && stack.back().first >= indent) stack.pop_back();
        if (val.empty()) {
            stack.push_back({indent, key});
            continue;
        }
        if (val.size() >= 2 && ((val.front() == '"' && val.back() == '"') || (val.front() == '\'' && val.back() == '\'')))
            val = val.substr(1, val.size() - 2);
        std::string full;
        for (const auto& s : stack) full += s.second + "/";
        out[full + key] = val;
    }
    return out;
}

namespace {

struct Reader {
    const std::map<std::string, std::string>& m;
    bool get(const std::string& k, std::string* v) const {
        auto it = m.find(k);
        if (it == m.end()) return false;
        *v = it->second;
        return true;
    }
    void param(const std::string& k, float* var) const {
        std::string v;
        if (get(k, &v)) *var = static_cast<float>(std::strtod(v.c_str(), nullptr));   // XmlRpc double -> float
    }
    void param(const std::string& k, int* var) const {
        std::string v;
        if (!get(k, &v)) return;
        if (v.find_first_of(".eE") != std::string::npos) {
            // roscpp's getImpl(int) accepts a double and rounds it: fmod(d,1) < 0.5 ? floor : ceil
            double d = std::strtod(v.c_str(), nullptr);
            d = (std::fmod(d, 1.0) < 0.5) ? std::floor(d) : std::ceil(d);
            if (std::isnan(d)) return;                                   // (the conversion of a NaN or of a value beyond int is undefined: clamped, a NaN leaves the default)
            *var = d >= 2147483647.0 ? 2147483647 : d <= -2147483648.0 ? (-2147483647 - 1) : static_cast<int>(d);
            return;
        }
        *var = static_cast<int>(std::strtol(v.c_str(), nullptr, 10));
    }
    void param(const std::string& k, bool* var) const {
        std::string v;
        if (!get(k, &v)) return;
        if (v == "true" || v == "True" || v == "TRUE" || v == "1") *var = true;
        else if (v == "false" || v == "False" || v == "FALSE" || v == "0") *var = false;
    }
};

}  // namespace

bool load_node_params(const std::string& yaml_path, NodeParams* p, std::string* error) {
    std::string err;
    const auto m = load_yaml_scalars(yaml_path, &err);
    if (!err.empty()) {
        if (error) *error = err;
        return false;
    }
    const Reader r{m};
    const std::string ns = "pointcloud_obstacle_processing/";          // launch/main.launch:11 <group ns=...>
    const std::string pv = ns + "obstacle_detection/";                 // node name, launch/main.launch:12
    r.param(pv + "accumulate_count", &p->accumulate_count);
    r.param(pv + "downsample_input_data", &p->downsample_input_data);
    r.param(pv + "passthrough_filter_enable", &p->passthrough_filter_enable);
    r.param(pv + "publish_point_clouds", &p->publish_point_clouds);
    r.param(pv + "grid_opacity", &p->grid_opacity);
    r.param(ns + "x_min", &p->x_min);
    r.param(ns + "x_max", &p->x_max);
    r.param(ns + "y_min", &p->y_min);
    r.param(ns + "y_max", &p->y_max);
    r.param(ns + "z_min", &p->z_min);
    r.param(ns + "z_max", &p->z_max);
    r.param(pv + "block_size", &p->block_size);
    r.param(pv + "dev_percent", &p->dev_percent);
    r.param(pv + "downsample_size", &p->downsample_size);
    r.param(pv + "statistical_outlier_meanK", &p->statistical_outlier_meanK);
    r.param(pv + "statistical_outlier_stdDevThres", &p->statistical_outlier_stdDevThres);
    r.param(pv + "plane_segment_dist_thres", &p->plane_segment_dist_thres);
    r.param(pv + "plane_segment_angle", &p->plane_segment_angle);
    r.param(pv + "euc_cluster_tolerance", &p->euc_cluster_tolerance);
    r.param(pv + "euc_min_cluster_size", &p->euc_min_cluster_size);
    r.param(pv + "euc_max_cluster_size", &p->euc_max_cluster_size);
    r.param(pv + "convex_hull_alpha", &p->convex_hull_alpha);
    return true;
}

}  // namespace pcop_host

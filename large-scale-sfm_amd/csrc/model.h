// model.h -- the in-memory model behind include/mi_model.h (internal to libmi_ba.so).
#pragma once
#include <string>
#include <vector>

struct mi_model {
    std::vector<std::string> names;
    std::vector<double> f, R, t, d;      // per camera: 1, 9, 3, 2
    std::vector<double> xyz;             // per point 3
    std::vector<int> rgb;                // per point 3
    std::vector<double> xy;              // per projection 2
    std::vector<int> pt_idx, cam_idx;
    std::vector<long long> feat_id;
    std::vector<int> focal_mask;         // shared-intrinsics groups (SetFocalMask semantics), empty = none
    double focal_weight = 0.0;
    size_t ncam() const { return f.size(); }
    size_t npt() const { return xyz.size() / 3; }
    size_t nproj() const { return pt_idx.size(); }
};

namespace miba {
void set_model_error(const std::string &msg); // what mi_model_last_error() returns (model_io.cpp)
}

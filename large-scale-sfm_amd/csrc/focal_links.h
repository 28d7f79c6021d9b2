// focal_links.h -- PBA's shared-intrinsics groups (ParallelBA::SetFocalMask, pba.h:124) turned into camera links.
// Restates ProcessIndexCameraQ / ProcessWeightCameraQ (SparseBundleCPU.cpp:2781-2892): the cameras of a group form a
// ring of soft equalities, weight sqrt(projections per camera of the group) * weight; mask[i] = first camera of i's
// group (mask[iq] == iq), -1 = on its own.
#pragma once
#include <cmath>
#include <cstddef>
#include <vector>

namespace miba {

struct FocalLinks {
    std::vector<int> a, b;        // ring links
    std::vector<double> w;        // weight of each link (on the raw focal-length difference)
    std::vector<int> rep;         // group (first camera) of each link
    std::vector<double> count;    // per camera: cameras in the group it leads (0 elsewhere)
};

// false: the mask is malformed (the reference prints "incorrect constraints", drops the mask and goes on)
inline bool focal_mask_links(int m, const int *mask, size_t nproj, const int *cam_idx, double weight, FocalLinks &out)
{
    out = FocalLinks();
    for (int i = 0; i < m; ++i) {
        const int iq = mask[i];
        if (iq > i || (iq >= 0 && mask[iq] != iq)) return false;
    }
    std::vector<double> cpnum(m, 0.0), qpnum(m, 0.0);
    out.count.assign(m, 0.0);
    for (size_t k = 0; k < nproj; ++k) if (cam_idx[k] >= 0 && cam_idx[k] < m) cpnum[cam_idx[k]] += 1.0;
    std::vector<int> last(m, -1);
    for (int i = 0; i < m; ++i) {
        const int qi = mask[i];
        if (qi < 0) continue;
        qpnum[qi] += cpnum[i]; out.count[qi] += 1.0;
        if (i != qi) { out.a.push_back(last[qi] < 0 ? qi : last[qi]); out.b.push_back(i); last[qi] = i; }
    }
    for (int qi = 0; qi < m; ++qi) if (last[qi] >= 0) { out.a.push_back(last[qi]); out.b.push_back(qi); } // close the ring
    for (size_t q = 0; q < out.a.size(); ++q) {
        const int qi = mask[out.a[q]];
        out.rep.push_back(qi);
        out.w.push_back(std::sqrt(qpnum[qi] / out.count[qi]) * weight);
    }
    return true;
}

// the group's values are first set to their mean (ProcessWeightCameraQ, :2867-2874)
template <class T>
inline void focal_mask_average(int m, const int *mask, T *values, size_t stride)
{
    std::vector<double> sum(m, 0.0), cnt(m, 0.0);
    for (int i = 0; i < m; ++i) if (mask[i] >= 0) { sum[mask[i]] += (double)values[(size_t)i * stride]; cnt[mask[i]] += 1.0; }
    for (int i = 0; i < m; ++i) if (mask[i] >= 0) values[(size_t)i * stride] = (T)(sum[mask[i]] / cnt[mask[i]]);
}

} // namespace miba

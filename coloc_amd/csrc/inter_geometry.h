// inter_geometry.h -- host arithmetic of the inter-camera step (internal; not installed): see inter_geometry.cpp.
#ifndef CLC_INTER_GEOMETRY_H
#define CLC_INTER_GEOMETRY_H

#include <stdint.h>

#include <utility>
#include <vector>

#include "../../include/coloc_hip.h"

namespace clc {
// what the relative pose leaves: the correspondences in front of both cameras = the pair's temporary map
struct InterFront {
    std::vector<double> Xt;        // 3 per point: source camera's frame, unit baseline
    std::vector<double> x2f;       // 2 per point: the destination's pixels
    std::vector<int32_t> corr;     // the correspondence (index into tv.x1 / x2) each point came from
    double R[9], t[3];             // the relative pose chosen by the chirality vote
};
// Between the two-view filter and the refinement of ColoC::interPoseEstimator (reference include/coloc/coloc.hpp:296-340), in two steps
// because the reference's own way to the common features -- matching the temporary map's descriptors against the global map's,
// :317-323 -- is device work that sits between them.  Both return CLC_INTER_OK or the stage that failed.
int inter_relative(clc_inter_pose_job& jb, InterFront& fr);
int inter_scale_pose(clc_inter_pose_job& jb, const InterFront& fr, const std::vector<std::pair<int32_t, int32_t>>& common, std::vector<double>& Xw);
} // namespace clc
#endif

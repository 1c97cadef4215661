// inter_geometry.h -- host arithmetic of the inter-camera step (internal; not installed): see inter_geometry.cpp.
#ifndef CLC_INTER_GEOMETRY_H
#define CLC_INTER_GEOMETRY_H

#include <vector>

#include "../../include/coloc_hip.h"

namespace clc {
// Between the two-view filter and the refinement of ColoC::interPoseEstimator (reference include/coloc/coloc.hpp:296-340): relative pose
// from E with the chirality vote, the pair's temporary map, its scale against the global map, the destination's first pose.  Fills
// jb.Rt / scale / n_front / n_common, Xw (the temporary map in world coordinates) and x2f (the destination's pixels of those points).
// Returns CLC_INTER_OK or the stage that failed.
int inter_geometry(clc_inter_pose_job& jb, std::vector<double>& Xw, std::vector<double>& x2f);
} // namespace clc
#endif

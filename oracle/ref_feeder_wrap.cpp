// ref_feeder_wrap.cpp -- ORACLE-SIDE wrapper (test infrastructure).
//
// Compiles the reference's own dependency-free host feeders, where they lie under
// /root/reference (include/coloc/KFAST.h:502-540 `KFAST<mt,nms>`, include/coloc/FeatureAngle.h:197
// `featureAngle`), into oracle/_ref/libref_feeder.so behind three C entry points.  Nothing from
// the reference is copied into this repository: the headers are only #included at build time in
// the build container; the GPU box receives the prebuilt .so.
//
// The headers use memset / std::thread without including <cstring> / <thread>
// (KFAST.h:427,506), so those two standard headers are pre-included here.
#include <cstring>
#include <thread>
#include <vector>
#include <cstdint>

#include "coloc/KFAST.h"
#include "coloc/FeatureAngle.h"

extern "C" {

// Runs KFAST<multithreading, true>; returns the number of keypoints, writes up to cap of them.
int ref_kfast(const uint8_t* img, int cols, int rows, int stride, uint8_t threshold,
              int multithreading, Keypoint* out, int cap)
{
    std::vector<Keypoint> kps;
    if (multithreading) KFAST<true, true>(img, cols, rows, stride, kps, threshold);
    else KFAST<false, true>(img, cols, rows, stride, kps, threshold);
    const int n = static_cast<int>(kps.size());
    for (int i = 0; i < n && i < cap; ++i) out[i] = kps[i];
    return n;
}

float ref_feature_angle(const uint8_t* img, int px, int py, int step)
{
    return featureAngle(img, px, py, step);
}

int ref_sizeof_keypoint(void) { return static_cast<int>(sizeof(Keypoint)); }

}

// HIPDetector.hpp -- drop-in for coloc::GPUDetector<T> (reference include/coloc/GPUDetector.hpp:30-292)
// over the C ABI of libcoloc_hip.so (include/coloc_hip.h).
//
// Same public surface: HIPDetector(DetectorOptions), detectFeaturesFile(idx, regions, imageName),
// detectFeaturesTopic(idx, regions, imagePtr) (USE_STREAM in the reference, :188-212; here a template on the image
// pointer type, so it compiles with cv_bridge::CvImagePtr where ROS exists and with any `->image.data / .cols / .rows`
// holder elsewhere), freeGPUMemory(), public `kps`, `desc`, `receivedImg`, `converted_kps` (:32-35).
// Same conventions: returns EXIT_SUCCESS / EXIT_FAILURE through T (so `false` == success, reference
// :183); Features()[i] = {s*x, s*y, 7*s, angle} with s = pow(1.2f, scale) (:172-179; the 1.2f is
// hard-coded in the reference independently of scale_factor and is kept); Descriptors()[i] = the 64
// raw descriptor bytes (:181).
// What differs underneath: the whole of detectAndDescribe (:216-291) -- pyramid, FAST-9 + NMS +
// orientation on every level, CLATCH -- runs on the GPU in one enqueue sequence with a single
// host synchronisation; the reference copies 7 levels back, runs KFAST on the CPU and synchronises
// 8 + 3 times per frame.
#pragma once

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>
#include <vector>

#include "coloc_hip.h"
#include "coloc_hip_types.hpp"

#ifdef COLOC_HIP_WITH_OPENCV
#include <opencv2/highgui/highgui.hpp>
#endif

static_assert(CLC_ABI_VERSION >= 3, "this policy header uses entry points of ABI version 3 (clc_detect_and_describe_view, clc_desc_handle)");

namespace coloc {

namespace hip_detail {
// Binary PGM (P5, maxval 255) reader used when OpenCV is not available (cv::imread(name, 0) otherwise).
inline bool read_pgm(const std::string& name, std::vector<uint8_t>& img, int& w, int& h)
{
    FILE* f = std::fopen(name.c_str(), "rb");
    if (!f) return false;
    char magic[3] = { 0, 0, 0 };
    int maxv = 0;
    bool ok = std::fscanf(f, "%2s", magic) == 1 && std::strcmp(magic, "P5") == 0;
    auto skip = [&]() {
        int c;
        while ((c = std::fgetc(f)) != EOF) {
            if (c == '#') { while ((c = std::fgetc(f)) != EOF && c != '\n') {} }
            else if (c > ' ') { std::ungetc(c, f); break; }
        }
    };
    if (ok) { skip(); ok = std::fscanf(f, "%d", &w) == 1; }
    if (ok) { skip(); ok = std::fscanf(f, "%d", &h) == 1; }
    if (ok) { skip(); ok = std::fscanf(f, "%d", &maxv) == 1 && maxv == 255 && w > 0 && h > 0; }
    if (ok) {
        std::fgetc(f);   // single whitespace after maxval
        img.resize(static_cast<size_t>(w) * h);
        ok = std::fread(img.data(), 1, img.size(), f) == img.size();
    }
    std::fclose(f);
    return ok;
}
} // namespace hip_detail

template <typename T>
class HIPDetector {
public:
    std::vector<Keypoint> kps;
    std::vector<uint64_t> desc;
    bool receivedImg = false;
#ifdef COLOC_HIP_WITH_OPENCV
    std::vector<cv::KeyPoint> converted_kps;       // GPUDetector.hpp:35 (cleared per topic frame, never filled there either)
#else
    std::vector<HipKeyPoint> converted_kps;
#endif

    explicit HIPDetector(DetectorOptions opts) : opts_(opts)
    {
        clc_detector_opts d;
        d.scale_factor = opts.scale_factor;
        d.scale_levels = opts.scale_levels;
        d.width = opts.width;
        d.height = opts.height;
        d.maxkp = opts.maxkp;
        d.thresh = opts.thresh;
        const int rc = clc_ctx_create(device_, &d, nullptr, &ctx_);
        if (rc != CLC_OK) {
            std::cerr << "HIPDetector: clc_ctx_create failed: " << clc_status_string(rc) << std::endl;
            ctx_ = nullptr;
        }
        // the library a host was LINKED against and the header it was COMPILED against must agree (the surface grew in
        // rounds 4 and 5 under one version number); a mismatch is reported, the calls that exist in both still work
        if (clc_abi_version() != CLC_ABI_VERSION)
            std::cerr << "HIPDetector: libcoloc_hip reports ABI version " << clc_abi_version() << ", this header was written for " << CLC_ABI_VERSION << std::endl;
        trustPublishedRegions(true);
    }
    HIPDetector(const HIPDetector&) = delete;
    HIPDetector& operator=(const HIPDetector&) = delete;
    ~HIPDetector() { freeGPUMemory(); }

    void freeGPUMemory()
    {
        if (ctx_) clc_ctx_destroy(ctx_);
        ctx_ = nullptr;
    }

    // The descriptor block this detector stores in regions[idx] is published to the matcher without a second upload (below).  true (the
    // default of the policy classes): the block is published for TRUSTING lookups -- the statement that nobody rewrites a stored
    // regions block in place, which holds for the reference's flow (only the detectors write Descriptors(): GPUDetector.hpp:181,210,
    // AKAZE.hpp:67).  false: it is published with a fold of all its rows, and a HIPMatcher switched the same way re-checks the whole
    // block on every call (include/coloc_hip.h, clc_desc_cache_mode).
    void trustPublishedRegions(bool on)
    {
        if (ctx_) (void)clc_desc_cache_mode(ctx_, on ? CLC_DESC_CACHE_TRUST : CLC_DESC_CACHE_VERIFY);
    }

    // Process an image read from disk (GPUDetector.hpp:158-184).
    T detectFeaturesFile(uint8_t idx, coloc::FeatureMap& regions, std::string& imageName)
    {
        std::vector<uint8_t> img;
        int w = 0, h = 0;
#ifdef COLOC_HIP_WITH_OPENCV
        cv::Mat image = cv::imread(imageName, 0);
        if (image.empty()) { std::cerr << "HIPDetector: cannot read " << imageName << std::endl; return EXIT_FAILURE; }
        w = image.cols; h = image.rows;
        img.assign(image.data, image.data + static_cast<size_t>(w) * h);
#else
        if (!hip_detail::read_pgm(imageName, img, w, h)) {
            std::cerr << "HIPDetector: cannot read " << imageName << " (binary PGM expected without OpenCV)" << std::endl;
            return EXIT_FAILURE;
        }
#endif
        return detectFeaturesImage(idx, regions, img.data(), w, h);
    }

    // Same from a grey image already in host memory (what detectFeaturesTopic hands over, :188-212).
    T detectFeaturesImage(uint8_t idx, coloc::FeatureMap& regions, const uint8_t* image, int width, int height)
    {
        if (!detectAndDescribe(image, static_cast<uint32_t>(width), static_cast<uint32_t>(height))) return EXIT_FAILURE;
        receivedImg = true;
        regions[idx] = std::unique_ptr<openMVG::features::AKAZE_Binary_Regions>(new openMVG::features::AKAZE_Binary_Regions);
        regions[idx]->Features().resize(kps.size());
        regions[idx]->Descriptors().resize(kps.size());
        std::vector<float> feat(4 * kps.size());
        clc_keypoints_to_features(reinterpret_cast<const clc_keypoint*>(kps.data()), static_cast<int>(kps.size()), feat.data());
        for (size_t i = 0; i < kps.size(); ++i) {
            regions[idx]->Features()[i] = { feat[4 * i], feat[4 * i + 1], feat[4 * i + 2], feat[4 * i + 3] };
            std::memcpy(&(regions[idx]->Descriptors()[i]), &(desc[i * 8]), 8 * sizeof(uint64_t));
        }
        // the rows just stored in regions[idx] are still in this context's device memory: publish them, so that a HIPMatcher call that
        // is handed regions[idx]->DescriptorRawData() on the same device reads them there instead of uploading them again (the
        // reference re-uploads per call, GPUMatcher.hpp:188-196).  A miss costs nothing; a failure to publish is not an error.
        if (!kps.empty()) (void)clc_desc_cache_publish(ctx_, nullptr, regions[idx]->DescriptorRawData(), static_cast<int>(kps.size()));
        return EXIT_SUCCESS;
    }

    // Process an image obtained from a ROS topic (GPUDetector.hpp:188-212).  ImagePtr = cv_bridge::CvImagePtr or any
    // pointer-like whose ->image has .data / .cols / .rows (8-bit grey).  The reference's version stores x in BOTH
    // feature coordinates (:204-205, a typo of the file variant :174-175); y is stored here.
    template <typename ImagePtr>
    void detectFeaturesTopic(uint8_t idx, coloc::FeatureMap& regions, ImagePtr imagePtr)
    {
        converted_kps.clear();
        (void)detectFeaturesImage(idx, regions, imagePtr->image.data, imagePtr->image.cols, imagePtr->image.rows);
    }

    const char* lastError() const { return ctx_ ? clc_last_error_string(ctx_) : "no context"; }
    int keypointsFound() const { return found_; }   // before the maxkp cap

private:
    static_assert(sizeof(Keypoint) == sizeof(clc_keypoint), "Keypoint wire format (Keypoint.h:155-163) must be 20 bytes");

    // GPUDetector::detectAndDescribe (:216-291) as ONE C-ABI call.
    bool detectAndDescribe(const uint8_t* image, const uint32_t width, const uint32_t height)
    {
        kps.clear();
        desc.clear();
        if (!ctx_) return false;
        kps.resize(opts_.maxkp);
        desc.resize(static_cast<size_t>(8) * opts_.maxkp);
        int n = 0;
        found_ = 0;
        const int rc = clc_detect_and_describe(ctx_, image, width, height, reinterpret_cast<clc_keypoint*>(kps.data()),
                                               reinterpret_cast<uint8_t*>(desc.data()), static_cast<int>(opts_.maxkp), &n, &found_);
        if (rc != CLC_OK) {
            std::cerr << "HIPDetector: " << clc_status_string(rc) << ": " << clc_last_error_string(ctx_) << std::endl;
            kps.clear();
            desc.clear();
            return false;
        }
        kps.resize(n);
        desc.resize(static_cast<size_t>(8) * n);
        return true;
    }

    DetectorOptions opts_;
    clc_ctx* ctx_ = nullptr;
    int device_ = 0;
    int found_ = 0;
};

} // namespace coloc

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
// host synchronisation (clc_detect_and_describe_view: the frame goes in and out through one pinned
// block of the context); the reference copies 7 levels back, runs KFAST on the CPU and synchronises
// 8 + 3 times per frame.  The descriptors are copied ONCE, into regions[idx] (the public `desc` is
// filled on request: mirrorRawOutputs(true)).
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
        // the library a host was LINKED against and the header it was COMPILED against must agree; a mismatch is reported, the calls
        // that exist in both still work
        if (clc_abi_version() != CLC_ABI_VERSION)
            std::cerr << "HIPDetector: libcoloc_hip reports ABI version " << clc_abi_version() << ", this header was written for " << CLC_ABI_VERSION << std::endl;
        // GPUDetector.hpp:173: static_cast<float>(std::pow(1.2f, kps[i].scale)) -- one pow per LEVEL here, the same call with the
        // same arguments (pow(float, integer) is evaluated in double)
        for (int l = 0; l < 256; ++l) levelScale_[l] = static_cast<float>(std::pow(static_cast<double>(1.2f), static_cast<double>(l)));
    }
    HIPDetector(const HIPDetector&) = delete;
    HIPDetector& operator=(const HIPDetector&) = delete;
    ~HIPDetector() { freeGPUMemory(); }

    void freeGPUMemory()
    {
        if (ctx_) clc_ctx_destroy(ctx_);
        ctx_ = nullptr;
    }

    // The descriptor block this detector stores in regions[idx] is handed to the matcher without a second upload: the rows stay on the
    // device in a block this detector's context owns, and storing them in regions[idx] publishes {address, count, generation} together
    // with a fold of all rows (clc_detect_store_descriptors).  A HIPMatcher call that is handed regions[idx]->DescriptorRawData() sweeps
    // the device rows at once and folds the host block behind the sweep: rows edited in place since are uploaded and matched as edited
    // (include/coloc_hip.h, clc_desc_cache_mode).  publishRegions(false): nothing is published, every match call uploads (the
    // reference's behaviour, GPUMatcher.hpp:188-196).  The publication of a frame dies when the next frame of the same idx is stored
    // at the same address, with freeGPUMemory(), or when a lookup sees changed rows.
    void publishRegions(bool on)
    {
        if (ctx_) (void)clc_desc_cache_mode(ctx_, on ? CLC_DESC_CACHE_VERIFY : CLC_DESC_CACHE_OFF);
    }
    // The frame's descriptors are copied ONCE, into regions[idx] (GPUDetector.hpp:181).  The public `desc` (:33) is a second copy of the
    // same rows; the only reader in the reference is InterfaceROS::processImagePair (InterfaceROS.hpp:30,37), whose call does not match
    // GPUMatcher::setTrainingImage's signature.  mirrorRawOutputs(true) fills it as the reference does; `kps` is always filled.
    void mirrorRawOutputs(bool on) { mirrorDesc_ = on; }
    // the handle of the block published by the last detect call (host == nullptr: nothing was published)
    const clc_desc_handle& lastPublished() const { return published_; }

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
        kps.clear();
        desc.clear();
        published_ = clc_desc_handle{};
        if (!ctx_) return EXIT_FAILURE;
        // GPUDetector::detectAndDescribe (:216-291) as ONE C-ABI call; keypoints and descriptors come back in the context's pinned block
        const clc_keypoint* pk = nullptr;
        const uint8_t* pd = nullptr;
        int n = 0;
        found_ = 0;
        const int rc = clc_detect_and_describe_view(ctx_, image, static_cast<uint32_t>(width), static_cast<uint32_t>(height), &pk, &pd, &n, &found_);
        if (rc != CLC_OK) {
            std::cerr << "HIPDetector: " << clc_status_string(rc) << ": " << clc_last_error_string(ctx_) << std::endl;
            return EXIT_FAILURE;
        }
        receivedImg = true;
        kps.resize(static_cast<size_t>(n));
        if (n > 0) std::memcpy(static_cast<void*>(kps.data()), pk, static_cast<size_t>(n) * sizeof(Keypoint));
        regions[idx] = std::unique_ptr<openMVG::features::AKAZE_Binary_Regions>(new openMVG::features::AKAZE_Binary_Regions);
        auto& feats = regions[idx]->Features();
        auto& descs = regions[idx]->Descriptors();
        feats.resize(static_cast<size_t>(n));
        descs.resize(static_cast<size_t>(n));
        for (int i = 0; i < n; ++i) {
            const float scale = levelScale_[kps[static_cast<size_t>(i)].scale];   // GPUDetector.hpp:172-179
            feats[static_cast<size_t>(i)] = { scale * static_cast<float>(kps[static_cast<size_t>(i)].x), scale * static_cast<float>(kps[static_cast<size_t>(i)].y),
                                              7.0f * scale, kps[static_cast<size_t>(i)].angle };
        }
        // the frame's one copy of its descriptors (:181), published on the way: a HIPMatcher call that is handed
        // regions[idx]->DescriptorRawData() on this device reads the rows where they already are
        if (n > 0) {
            static_assert(sizeof(descs[0]) == CLC_DESC_BYTES, "a regions descriptor is the 64 raw bytes");
            const int rs = clc_detect_store_descriptors(ctx_, static_cast<void*>(descs.data()), n, &published_);
            if (rs != CLC_OK) {
                std::cerr << "HIPDetector: " << clc_status_string(rs) << ": " << clc_last_error_string(ctx_) << std::endl;
                return EXIT_FAILURE;
            }
            if (mirrorDesc_) {
                desc.resize(static_cast<size_t>(8) * n);
                std::memcpy(desc.data(), descs.data(), static_cast<size_t>(n) * CLC_DESC_BYTES);
            }
        }
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

    DetectorOptions opts_;
    clc_ctx* ctx_ = nullptr;
    int device_ = 0;
    int found_ = 0;
    bool mirrorDesc_ = false;
    float levelScale_[256];
    clc_desc_handle published_{};
};

} // namespace coloc

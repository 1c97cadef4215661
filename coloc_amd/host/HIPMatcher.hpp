// HIPMatcher.hpp -- drop-in for coloc::GPUMatcher<T> (reference include/coloc/GPUMatcher.hpp:46-272)
// over the C ABI of libcoloc_hip.so (include/coloc_hip.h).
//
// Same public surface and index conventions:
//   computeMatches(FeatureMap&, PairWiseMatches&)      all (first < second) pairs, empty results not inserted (:143-155)
//   computeMatchesPair(pair, regions, out)             IndMatch(i_ = query idx in regions[first], j_ = train idx in regions[second]), thr 40 (:165-172,:217)
//   matchMapFeatures(map1, map2, out)                  IndMatch(i_ = map1 idx, j_ = map2 idx), thr 60 (:157-163)
//   setMapData(n, desc) / matchSceneWithMap(id, data, out) / matchFeaturesWithMap()
//                                                      IndMatch(i_ = MAP idx, j_ = query idx), thr = MatcherOptions.thresh (:110-117,:174-178,:252-271)
//   setTrainingImage / setQueryImage / computeMatchesPreset (:120-141,:229-250), freeGPUMemory, maxkpNum, dmatches
// Only accepted queries are emitted, in ascending query order.  T carries EXIT_SUCCESS / EXIT_FAILURE.
// Differences underneath: no +8-vector over-read of the caller's buffers (:183-190), explicit
// capacity errors instead of silent overflow, one upload per camera and one launch group for the
// whole all-pairs loop, no texture objects.
#pragma once

#include <cstdlib>
#include <iostream>
#include <memory>
#include <vector>

#include "coloc_hip.h"
#include "coloc_hip_types.hpp"

static_assert(CLC_ABI_VERSION >= 3, "this policy header uses entry points of ABI version 3 (clc_detect_and_describe_view, clc_desc_handle)");

namespace coloc {

template <typename T>
class HIPMatcher {
public:
    unsigned int maxkpNum;
    std::vector<HipDMatch> dmatches;

    explicit HIPMatcher(MatcherOptions opts) : maxkpNum(opts.maxkp), matchThreshold_(static_cast<uint8_t>(opts.thresh))
    {
        clc_matcher_opts m;
        m.distRatio = opts.distRatio;
        m.thresh = opts.thresh;
        m.maxkp = opts.maxkp;
        const int rc = clc_ctx_create(0, nullptr, &m, &ctx_);
        if (rc != CLC_OK) {
            std::cerr << "HIPMatcher: clc_ctx_create failed: " << clc_status_string(rc) << std::endl;
            ctx_ = nullptr;
        }
        // the library a host was LINKED against and the header it was COMPILED against must agree; a mismatch is reported, the calls
        // that exist in both still work
        if (clc_abi_version() != CLC_ABI_VERSION)
            std::cerr << "HIPMatcher: libcoloc_hip reports ABI version " << clc_abi_version() << ", this header was written for " << CLC_ABI_VERSION << std::endl;
    }
    HIPMatcher(const HIPMatcher&) = delete;
    HIPMatcher& operator=(const HIPMatcher&) = delete;
    ~HIPMatcher() { freeGPUMemory(); }

    void freeGPUMemory()
    {
        if (ctx_) clc_ctx_destroy(ctx_);
        ctx_ = nullptr;
    }

    // Descriptor blocks HIPDetector published are read on the device instead of being uploaded again (the reference uploads per call,
    // GPUMatcher.hpp:188-196).  Default: checked -- the sweep starts on the device rows, the host folds the block it was handed while the
    // GPU works and compares with the fold taken at publish time; a block edited anywhere is uploaded and swept again, so the answer is
    // always the one for the rows passed in.  trustPublishedRegions(true) is the integrator's statement that published regions blocks
    // are never edited in place: address, count, generation and 18 sampled rows then decide (the detector must publish folds -- it
    // does by default -- for the checked mode to find its blocks).  usePublishedRegions(false): every call uploads, like the reference.
    void trustPublishedRegions(bool on)
    {
        if (ctx_) (void)clc_desc_cache_mode(ctx_, on ? CLC_DESC_CACHE_TRUST : CLC_DESC_CACHE_VERIFY);
    }
    void usePublishedRegions(bool on)
    {
        if (ctx_) (void)clc_desc_cache_mode(ctx_, on ? CLC_DESC_CACHE_VERIFY : CLC_DESC_CACHE_OFF);
    }

    void setMapData(int kpNum, void* desc)
    {
        if (check(clc_set_map(ctx_, desc, kpNum), "setMapData")) kpMap_ = kpNum;
    }

    void setTrainingImage(int kpNum, std::vector<uint64_t> const& desc)
    {
        train_.assign(desc.begin(), desc.begin() + static_cast<size_t>(8) * kpNum);
        kpTrain_ = kpNum;
    }

    void setQueryImage(int kpNum, void* desc)
    {
        const uint64_t* p = static_cast<const uint64_t*>(desc);
        query_.assign(p, p + static_cast<size_t>(8) * kpNum);
        queryRef_ = nullptr;
        kpQuery_ = kpNum;
    }

    T computeMatches(FeatureMap& regions, openMVG::matching::PairWiseMatches& putativeMatches)
    {
        const int numImages = static_cast<int>(regions.size());
        const openMVG::Pair_Set pairs = Utils::handlePairs(numImages);
        if (pairs.empty()) return EXIT_SUCCESS;
        // one upload per camera, one launch group for all pairs
        std::vector<const void*> descs(numImages, nullptr);
        std::vector<int> counts(numImages, 0);
        for (const auto& kv : regions) {
            if (static_cast<int>(kv.first) >= numImages) return EXIT_FAILURE;   // handlePairs assumes ids 0..n-1
            descs[kv.first] = kv.second->DescriptorRawData();
            counts[kv.first] = static_cast<int>(kv.second->RegionCount());
        }
        std::vector<int> flat;
        std::vector<std::vector<int32_t>> out;
        for (const auto& p : pairs) {
            flat.push_back(static_cast<int>(p.first));
            flat.push_back(static_cast<int>(p.second));
            out.emplace_back(counts[p.first]);
        }
        std::vector<int32_t*> optr;
        for (auto& o : out) optr.push_back(o.data());
        if (!check(clc_match_pairs(ctx_, descs.data(), counts.data(), numImages, flat.data(), static_cast<int>(pairs.size()), 40,
                                   optr.data()), "computeMatches"))
            return EXIT_FAILURE;
        size_t k = 0;
        for (const auto& pairIdx : pairs) {
            openMVG::matching::IndMatches pairMatches;
            const std::vector<int32_t>& m = out[k++];
            for (size_t i = 0; i < m.size(); ++i)
                if (m[i] != -1) pairMatches.emplace_back(static_cast<openMVG::IndexT>(i), static_cast<openMVG::IndexT>(m[i]));
            if (!pairMatches.empty()) putativeMatches.insert({ pairIdx, std::move(pairMatches) });
        }
        return EXIT_SUCCESS;
    }

    void matchMapFeatures(std::unique_ptr<openMVG::features::AKAZE_Binary_Regions>& map1,
                          std::unique_ptr<openMVG::features::AKAZE_Binary_Regions>& map2,
                          openMVG::matching::IndMatches& commonFeatures)
    {
        commonFeatures = computeMatches(const_cast<void*>(map1->DescriptorRawData()), const_cast<void*>(map2->DescriptorRawData()),
                                        static_cast<int>(map1->RegionCount()), static_cast<int>(map2->RegionCount()), 60);
    }

    void computeMatchesPair(const openMVG::Pair& pairIdx, FeatureMap& regions, openMVG::matching::IndMatches& putativeMatches)
    {
        putativeMatches = computeMatches(const_cast<void*>(regions[pairIdx.first]->DescriptorRawData()),
                                         const_cast<void*>(regions[pairIdx.second]->DescriptorRawData()),
                                         static_cast<int>(regions[pairIdx.first]->RegionCount()),
                                         static_cast<int>(regions[pairIdx.second]->RegionCount()));
    }

    // GPUMatcher.hpp:174-178: setQueryImage(regions[droneId]) + matchFeaturesWithMap().  The regions block itself is handed to the
    // library (no copy into a query buffer first), so that a block HIPDetector published is found on the device.
    void matchSceneWithMap(int& droneId, colocData& data, openMVG::matching::IndMatches& mapMatches)
    {
        // (the query set by this call IS the regions block -- not a copy of it: a later computeMatchesPreset() / matchFeaturesWithMap()
        // reads it there, so it must still exist then; in the reference's flow regions[droneId] lives until the drone's next frame)
        kpQuery_ = static_cast<unsigned int>(data.regions[droneId]->RegionCount());
        queryRef_ = data.regions[droneId]->DescriptorRawData();
        mapMatches = matchWithMap(queryRef_, static_cast<int>(kpQuery_));
    }

    // GPUMatcher.hpp:180-226: IndMatch(i_ = query index, j_ = train index); dmatches(train, query, 0)
    openMVG::matching::IndMatches computeMatches(void* h_descriptorsQuery, void* h_descriptorsTraining, int numKPQuery,
                                                 int numKPTraining, uint8_t threshold = 40)
    {
        openMVG::matching::IndMatches matches;
        std::vector<int32_t> m(numKPQuery > 0 ? numKPQuery : 0);
        dmatches.clear();
        if (!check(clc_match_2nn(ctx_, h_descriptorsQuery, numKPQuery, h_descriptorsTraining, numKPTraining, threshold, m.data(),
                                 nullptr, nullptr), "computeMatches"))
            return matches;
        for (size_t i = 0; i < m.size(); ++i) {
            if (m[i] != -1) {
                matches.emplace_back(static_cast<openMVG::IndexT>(i), static_cast<openMVG::IndexT>(m[i]));
                dmatches.emplace_back(m[i], static_cast<int>(i), 0.0f);
            }
        }
        return matches;
    }

    // GPUMatcher.hpp:229-250: preset train/query images; IndMatch(i_ = train index, j_ = query index)
    openMVG::matching::IndMatches computeMatchesPreset()
    {
        openMVG::matching::IndMatches matches;
        std::vector<int32_t> m(kpQuery_);
        dmatches.clear();
        if (!check(clc_match_2nn(ctx_, queryData(), static_cast<int>(kpQuery_), train_.data(), static_cast<int>(kpTrain_),
                                 matchThreshold_, m.data(), nullptr, nullptr), "computeMatchesPreset"))
            return matches;
        for (size_t i = 0; i < m.size(); ++i) {
            if (m[i] != -1) {
                matches.emplace_back(static_cast<openMVG::IndexT>(m[i]), static_cast<openMVG::IndexT>(i));
                dmatches.emplace_back(m[i], static_cast<int>(i), 0.0f);
            }
        }
        return matches;
    }

    // GPUMatcher.hpp:252-271: IndMatch(i_ = MAP index, j_ = query index)
    openMVG::matching::IndMatches matchFeaturesWithMap() { return matchWithMap(queryData(), static_cast<int>(kpQuery_)); }

    const char* lastError() const { return ctx_ ? clc_last_error_string(ctx_) : "no context"; }

private:
    const void* queryData() const { return queryRef_ ? queryRef_ : static_cast<const void*>(query_.data()); }

    openMVG::matching::IndMatches matchWithMap(const void* desc, int n)
    {
        openMVG::matching::IndMatches matches;
        std::vector<int32_t> m(n > 0 ? n : 0);
        if (!check(clc_match_map(ctx_, desc, n, matchThreshold_, m.data()), "matchFeaturesWithMap")) return matches;
        for (size_t i = 0; i < m.size(); ++i)
            if (m[i] != -1) matches.emplace_back(static_cast<openMVG::IndexT>(m[i]), static_cast<openMVG::IndexT>(i));
        return matches;
    }

    bool check(int rc, const char* what)
    {
        if (rc == CLC_OK) return true;
        std::cerr << "HIPMatcher::" << what << ": " << clc_status_string(rc) << ": " << lastError() << std::endl;
        return false;
    }

    clc_ctx* ctx_ = nullptr;
    uint8_t matchThreshold_;
    unsigned int kpTrain_ = 0, kpQuery_ = 0, kpMap_ = 0;
    std::vector<uint64_t> train_, query_;
    const void* queryRef_ = nullptr;     // the query set by matchSceneWithMap: the regions block itself
};

} // namespace coloc

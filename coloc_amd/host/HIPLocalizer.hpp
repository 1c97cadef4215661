// HIPLocalizer.hpp -- drop-in for coloc::Localizer (reference include/coloc/Localizer.hpp:19-177) over the C ABI of
// libcoloc_hip.so (include/coloc_hip.h).
//
// Same public surface, argument meaning and status convention:
//   Localizer(colocParams&)                                                               :22-32
//   bool setupTracks(cam*, data, queryRegions, trackedFeatures, trackPtr)                 :59-75
//        pt3D.col(i) = landmark X of data.mapRegionIdx[m.i_], pt2D.col(i) = position of query feature m.j_, passed
//        through cam->get_ud_pixel() (Pinhole_Intrinsic_Radial_K3: have_disto() is always true)
//   bool localizeImage(idx, pose, data, covariance, rmse, trackedFeatures, inliers)       :77-108
//        SfM_Localizer::Localize(P3P_KE_CVPR17, imageSize, &cam, {error_max = +inf, max_iteration = 256}, pose)
//        = a-contrario RANSAC over P3P  ->  clc_pnp_acransac;  success iff inliers > 2.5 x 3 (SfM_Localizer::Localize);
//        then refine()
//   bool refine(idx, pose, matchData, poseCovariance, rmse)                               :110-177
//        PoseRefiner::refinePose (Ceres, Huber(4^2), pose only) + 6x6 covariance  ->  clc_pnp_refine;
//        rmse = mean pixel distance between the observations and the reprojected inliers (the value the reference
//        leaves in rmse after cv::projectPoints with zero distortion and fy = fx, :141-170)
// Functions return EXIT_SUCCESS / EXIT_FAILURE through bool, i.e. FALSE MEANS SUCCESS (coloc.hpp:241,338), except
// refine(), which like the reference returns the refiner's own status (true = refined).
// Second call site of the same solve: Reconstructor::resectionCamera (Reconstructor.hpp:282-307) calls
//   sfm::SfM_Localizer::Localize(resection::SolverType::P3P_KE_CVPR17, {w, h}, intrinsics, resectionData, pose)
// directly; HIP_SfM_Localizer::Localize below has that signature (P3P only: without intrinsics OpenMVG falls back to the
// uncalibrated 6-point DLT kernel, which is not on this path -> returns false).
// What differs underneath: the random samples (OpenMVG's std::mt19937 stream is unpinned -- `seed` below selects
// the documented counter-based sampler of clc_acr.h), the P3P root order and Ceres' iterates (absent submodules).
#pragma once

#include <cmath>
#include <cstdlib>
#include <iostream>
#include <limits>
#include <vector>

#include "coloc_hip.h"
#include "coloc_hip_geometry.hpp"

namespace coloc {

class HIPLocalizer {
public:
    uint64_t seed = 1;                        // sampler seed of the next localizeImage call (incremented per call)

    explicit HIPLocalizer(colocParams& params)
        : imageSize(&params.imageSize), K(&params.K), dist(&params.dist), rootFolder(&params.imageFolder)
    {
        const int rc = clc_ctx_create(0, nullptr, nullptr, &ctx_);
        if (rc != CLC_OK) {
            std::cerr << "HIPLocalizer: clc_ctx_create failed: " << clc_status_string(rc) << std::endl;
            ctx_ = nullptr;
        }
    }
    HIPLocalizer(const HIPLocalizer&) = delete;
    HIPLocalizer& operator=(const HIPLocalizer&) = delete;
    ~HIPLocalizer()
    {
        for (clc_ctx* c : batch_ctxs_) clc_ctx_destroy(c);
        if (ctx_) clc_ctx_destroy(ctx_);
    }

    bool setupTracks(openMVG::cameras::Pinhole_Intrinsic_Radial_K3* cam, colocData& data,
                     const openMVG::features::AKAZE_Binary_Regions& queryRegions, openMVG::matching::IndMatches& trackedFeatures,
                     openMVG::sfm::Image_Localizer_Match_Data* trackPtr)
    {
        trackPtr->pt3D.resize(3, trackedFeatures.size());
        trackPtr->pt2D.resize(2, trackedFeatures.size());
        for (size_t i = 0; i < trackedFeatures.size(); ++i) {
            const auto& X = data.scene.GetLandmarks().at(data.mapRegionIdx[trackedFeatures[i].i_]).X;
            for (int r = 0; r < 3; ++r) trackPtr->pt3D(r, i) = X[r];
            const auto p = queryRegions.GetRegionPosition(trackedFeatures[i].j_);
            openMVG::Vec2 px(static_cast<double>(p[0]), static_cast<double>(p[1]));
            if (cam && cam->have_disto()) px = cam->get_ud_pixel(px);
            trackPtr->pt2D(0, i) = px[0];
            trackPtr->pt2D(1, i) = px[1];
        }
        return EXIT_SUCCESS;
    }

    bool localizeImage(int& idx, openMVG::geometry::Pose3& pose, colocData& data, Cov6& covariance, float& rmse,
                       openMVG::matching::IndMatches& trackedFeatures, std::vector<uint32_t>& inliers)
    {
        openMVG::cameras::Pinhole_Intrinsic_Radial_K3 cam = camera(idx);
        openMVG::sfm::Image_Localizer_Match_Data matching_data;
        matching_data.error_max = std::numeric_limits<double>::infinity();
        matching_data.max_iteration = 256;
        if (setupTracks(&cam, data, *data.regions.at(idx).get(), trackedFeatures, &matching_data) == EXIT_FAILURE) {
            std::cout << "Failure while setting up 2D-3D correspondences" << std::endl;
            return EXIT_FAILURE;
        }
        // SfM_Localizer::Localize (:93) and refine() (:100-106) as ONE submission (clc_pnp_localize_ac: the refinement launch follows the
        // a-contrario rounds on the device; one upload, no second round trip): the pose, covariance and rmse are those of
        // localize() followed by refine()
        bool refined = false;
        if (!localize_with(ctx_, seed++, cam, matching_data, pose, &covariance, &rmse, &refined)) {
            std::cout << "Localization unsuccessful" << std::endl;
            return EXIT_FAILURE;
        }
        std::cout << "Localization successful" << std::endl;
        inliers = matching_data.vec_inliers;
        if (!refined) std::cerr << "Refining pose for image failed." << std::endl;
        return EXIT_SUCCESS;
    }

    // localizeImage for SEVERAL cameras at once (not in the reference, whose loop calls localizeImage camera by camera, coloc.hpp:129-137;
    // BASELINE config[2]: "batched PnP/RANSAC pose").  The cameras' a-contrario solves -- chains of short launches with the host in the
    // loop -- run through clc_pnp_localize_ac_batch, each on a light context of its own, and interleave on the device; every camera's
    // refinement rides behind its own rounds.  Poses, inliers, covariances and the sampler seeds are exactly those of calling localizeImage for idxs[0],
    // idxs[1], ... in this order.  Returns one status per camera in the members' convention (false = success).
    std::vector<bool> localizeImages(const std::vector<int>& idxs, std::vector<openMVG::geometry::Pose3>& poses, colocData& data,
                                     std::vector<Cov6>& covariances, std::vector<float>& rmses,
                                     std::vector<openMVG::matching::IndMatches>& trackedFeatures, std::vector<std::vector<uint32_t>>& inliers)
    {
        const size_t nc = idxs.size();
        std::vector<bool> status(nc, static_cast<bool>(EXIT_FAILURE));
        if (poses.size() < nc) poses.resize(nc);
        if (covariances.size() < nc) covariances.resize(nc);
        if (rmses.size() < nc) rmses.resize(nc, 0.0f);
        if (inliers.size() < nc) inliers.resize(nc);
        if (!ctx_ || trackedFeatures.size() < nc) return status;
        while (batch_ctxs_.size() < nc) {
            clc_ctx* c = nullptr;
            if (clc_ctx_create(0, nullptr, nullptr, &c) != CLC_OK) return status;
            batch_ctxs_.push_back(c);
        }
        std::vector<openMVG::sfm::Image_Localizer_Match_Data> md(nc);
        std::vector<std::vector<double>> X(nc), x(nc);
        std::vector<std::vector<int32_t>> inl(nc);
        std::vector<double> Kd(9 * nc), Rt(12 * nc), covs(36 * nc);
        std::vector<clc_pose_job> jobs;
        std::vector<size_t> who;
        for (size_t k = 0; k < nc; ++k) {
            int idx = idxs[k];
            openMVG::cameras::Pinhole_Intrinsic_Radial_K3 cam = camera(idx);
            md[k].error_max = std::numeric_limits<double>::infinity();
            md[k].max_iteration = 256;
            if (setupTracks(&cam, data, *data.regions.at(idx).get(), trackedFeatures[k], &md[k]) == EXIT_FAILURE) {
                std::cout << "Failure while setting up 2D-3D correspondences" << std::endl;
                continue;
            }
            const uint64_t seed_value = seed++;                        // (localize() takes its seed before it looks at the data)
            const int n = static_cast<int>(md[k].pt3D.cols());
            md[k].vec_inliers.clear();
            if (n == 0) { std::cout << "Localization unsuccessful" << std::endl; continue; }
            X[k].resize(3 * static_cast<size_t>(n)); x[k].resize(2 * static_cast<size_t>(n)); inl[k].resize(static_cast<size_t>(n));
            for (int i = 0; i < n; ++i) {
                for (int r = 0; r < 3; ++r) X[k][3 * i + r] = md[k].pt3D(r, i);
                x[k][2 * i] = md[k].pt2D(0, i);
                x[k][2 * i + 1] = md[k].pt2D(1, i);
            }
            intrinsics(cam, &Kd[9 * k]);
            clc_pose_job jb{};
            jb.X = X[k].data(); jb.x = x[k].data(); jb.K = &Kd[9 * k]; jb.n = n;
            jb.max_iteration = static_cast<int>(md[k].max_iteration); jb.seed = seed_value;
            jb.precision = md[k].error_max;                            // +inf: the a-contrario threshold
            jb.refine = 1; jb.huber_a = 16.0;                           // Refiner.hpp:122
            jb.Rt = &Rt[12 * k]; jb.cov = &covs[36 * k]; jb.inliers = inl[k].data();
            jobs.push_back(jb);
            who.push_back(k);
        }
        if (!jobs.empty()) {
            const int rc = clc_pnp_localize_ac_batch(batch_ctxs_.data(), jobs.data(), static_cast<int>(jobs.size()));
            if (rc != CLC_OK) std::cerr << "HIPLocalizer: clc_pnp_localize_ac_batch: " << clc_status_string(rc) << std::endl;
        }
        for (size_t j = 0; j < jobs.size(); ++j) {
            const size_t k = who[j];
            const clc_pose_job& jb = jobs[j];
            if (jb.status != CLC_OK) { std::cout << "Localization unsuccessful" << std::endl; continue; }
            md[k].vec_inliers.assign(inl[k].begin(), inl[k].begin() + jb.n_inliers);
            if (jb.n_inliers > 0) md[k].error_max = jb.error_max;
            if (!(jb.n_inliers > 2.5 * 3)) { std::cout << "Localization unsuccessful" << std::endl; continue; }
            poses[k] = pose_from_Rt(&Rt[12 * k]);
            std::cout << "Localization successful" << std::endl;
            inliers[k] = md[k].vec_inliers;
            for (int i = 0; i < 6; ++i) for (int j2 = 0; j2 < 6; ++j2) covariances[k](i, j2) = covs[36 * k + 6 * i + j2];
            rmses[k] = mean_reprojection(poses[k], &Kd[9 * k], md[k]);
            status[k] = static_cast<bool>(EXIT_SUCCESS);
        }
        return status;
    }

    // SfM_Localizer::Localize(P3P_KE_CVPR17, ...): true = a pose supported by more than 2.5 x 3 points was found;
    // matching_data.vec_inliers / error_max are updated like Localize does
    bool localize(const openMVG::cameras::Pinhole_Intrinsic_Radial_K3& cam, openMVG::sfm::Image_Localizer_Match_Data& matching_data,
                  openMVG::geometry::Pose3& pose)
    {
        return localize_with(ctx_, seed++, cam, matching_data, pose, nullptr, nullptr, nullptr);
    }

    // the solve itself, on any context (shared with HIP_SfM_Localizer::Localize).  covariance != nullptr: the refinement (Huber(4^2) LM on
    // the inliers + 6 x 6 covariance, refine() below) rides in the same submission; *refined says whether it ran, *rmse is the mean
    // pixel distance of the reprojected inliers (the value the reference leaves in rmse, Localizer.hpp:141-170)
    static bool localize_with(clc_ctx* ctx_, const uint64_t seed_value, const openMVG::cameras::Pinhole_Intrinsic_Radial_K3& cam,
                              openMVG::sfm::Image_Localizer_Match_Data& matching_data, openMVG::geometry::Pose3& pose,
                              Cov6* covariance = nullptr, float* rmse = nullptr, bool* refined = nullptr)
    {
        const int n = static_cast<int>(matching_data.pt3D.cols());
        matching_data.vec_inliers.clear();
        if (refined) *refined = false;
        if (!ctx_ || n == 0) return false;
        std::vector<double> X(3 * static_cast<size_t>(n)), x(2 * static_cast<size_t>(n));
        for (int i = 0; i < n; ++i) {
            for (int r = 0; r < 3; ++r) X[3 * i + r] = matching_data.pt3D(r, i);
            x[2 * i] = matching_data.pt2D(0, i);
            x[2 * i + 1] = matching_data.pt2D(1, i);
        }
        double Kd[9];
        intrinsics(cam, Kd);
        const double precision = std::isinf(matching_data.error_max) ? matching_data.error_max : matching_data.error_max * matching_data.error_max;
        double Rt[12], cov[36], emax = 0.0, nfa = 0.0, r = 0.0;
        std::vector<int32_t> inl(static_cast<size_t>(n));
        int n_inl = 0, its = 0;
        const int rc = covariance
            ? clc_pnp_localize_ac(ctx_, X.data(), x.data(), n, Kd, static_cast<int>(matching_data.max_iteration), seed_value, precision, 16.0, Rt, cov,
                                  nullptr, inl.data(), &n_inl, &emax, &r)
            : clc_pnp_acransac(ctx_, X.data(), x.data(), n, Kd, static_cast<int>(matching_data.max_iteration), seed_value, precision, Rt,
                               nullptr, inl.data(), &n_inl, &emax, &nfa, &its);
        if (rc != CLC_OK) {
            std::cerr << "HIPLocalizer: " << (covariance ? "clc_pnp_localize_ac: " : "clc_pnp_acransac: ") << clc_last_error_string(ctx_) << std::endl;
            return false;
        }
        matching_data.vec_inliers.assign(inl.begin(), inl.begin() + n_inl);
        if (n_inl > 0) matching_data.error_max = emax;
        if (!(n_inl > 2.5 * 3)) return false;                    // bResection = inliers > 2.5 * MINIMUM_SAMPLES
        pose = pose_from_Rt(Rt);
        if (covariance) {
            for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) (*covariance)(i, j) = cov[6 * i + j];
            if (rmse) *rmse = mean_reprojection(pose, Kd, matching_data);
            if (refined) *refined = true;
        }
        return true;
    }

    // mean pixel distance between the observations and the reprojected inliers (Localizer.hpp:141-170: zero distortion, fy = fx)
    static float mean_reprojection(const openMVG::geometry::Pose3& pose, const double* Kd, const openMVG::sfm::Image_Localizer_Match_Data& md)
    {
        const size_t n = md.vec_inliers.size();
        if (n == 0) return 0.0f;
        const openMVG::Vec3 tt = pose.translation();
        double error = 0.0;
        for (size_t i = 0; i < n; ++i) {
            const size_t c = md.vec_inliers[i];
            double pc[3];
            for (int a = 0; a < 3; ++a)
                pc[a] = pose.rotation()(a, 0) * md.pt3D(0, c) + pose.rotation()(a, 1) * md.pt3D(1, c) + pose.rotation()(a, 2) * md.pt3D(2, c) + tt[a];
            const double u = Kd[0] * pc[0] / pc[2] + Kd[2], v = Kd[0] * pc[1] / pc[2] + Kd[5];
            error += std::sqrt((md.pt2D(0, c) - u) * (md.pt2D(0, c) - u) + (md.pt2D(1, c) - v) * (md.pt2D(1, c) - v));
        }
        return static_cast<float>(error / static_cast<double>(n));
    }

    bool refine(int& idx, openMVG::geometry::Pose3& pose, openMVG::sfm::Image_Localizer_Match_Data& matchData, Cov6& poseCovariance,
                float& rmse)
    {
        const openMVG::cameras::Pinhole_Intrinsic_Radial_K3 cam = camera(idx);
        const size_t n_inl = matchData.vec_inliers.size();
        const size_t n = static_cast<size_t>(matchData.pt3D.cols());
        if (!ctx_ || n_inl < 3) return false;
        // all correspondences + the inlier mask: the launch localizeImage's single submission makes, hence the same bits
        std::vector<double> X(3 * n), x(2 * n);
        std::vector<uint8_t> mask(n, 0);
        for (size_t i = 0; i < n; ++i) {
            for (int r = 0; r < 3; ++r) X[3 * i + r] = matchData.pt3D(r, i);
            x[2 * i] = matchData.pt2D(0, i);
            x[2 * i + 1] = matchData.pt2D(1, i);
        }
        for (size_t i = 0; i < n_inl; ++i) if (matchData.vec_inliers[i] < n) mask[matchData.vec_inliers[i]] = 1;
        double Kd[9], Rt0[12], Rt[12], cov[36], r = 0.0;
        intrinsics(cam, Kd);
        const openMVG::Vec3 t = pose.translation();
        for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) Rt0[4 * i + j] = pose.rotation()(i, j); Rt0[4 * i + 3] = t[i]; }
        int its = 0;
        const int rc = clc_pnp_refine(ctx_, X.data(), x.data(), static_cast<int>(n), Kd, mask.data(), Rt0, 16.0, 50, Rt, cov, &r, &its);
        const bool refineStatus = rc == CLC_OK;
        if (refineStatus) {
            pose = pose_from_Rt(Rt);
            for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) poseCovariance(i, j) = cov[6 * i + j];
        }
        rmse = mean_reprojection(pose, Kd, matchData);
        return refineStatus;
    }

private:
    openMVG::cameras::Pinhole_Intrinsic_Radial_K3 camera(int idx) const
    {
        return openMVG::cameras::Pinhole_Intrinsic_Radial_K3(imageSize->first, imageSize->second, (*K)[idx](0, 0), (*K)[idx](0, 2),
                                                             (*K)[idx](1, 2), (*dist)[idx](0), (*dist)[idx](1), (*dist)[idx](2));
    }
    static void intrinsics(const openMVG::cameras::Pinhole_Intrinsic_Radial_K3& cam, double* Kd)
    {
        const openMVG::Vec2 pp = cam.principal_point();
        const double k[9] = { cam.focal(), 0.0, pp[0], 0.0, cam.focal(), pp[1], 0.0, 0.0, 1.0 };
        for (int i = 0; i < 9; ++i) Kd[i] = k[i];
    }
    static openMVG::geometry::Pose3 pose_from_Rt(const double* Rt)     // Pose3(R, -R^T t): KRt_From_P + Localize
    {
        openMVG::Mat3 R;
        openMVG::Vec3 C;
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) R(i, j) = Rt[4 * i + j];
        for (int i = 0; i < 3; ++i) C[i] = -(R(0, i) * Rt[3] + R(1, i) * Rt[7] + R(2, i) * Rt[11]);
        return openMVG::geometry::Pose3(R, C);
    }

    clc_ctx* ctx_ = nullptr;
    std::vector<clc_ctx*> batch_ctxs_;          // localizeImages: one light context per camera of a batch, created on first use
    std::pair<int, int>* imageSize;
    std::vector<openMVG::Mat3>* K;
    std::vector<openMVG::Vec3>* dist;
    std::string* rootFolder;
};

// sfm::SfM_Localizer::Localize as Reconstructor::resectionCamera calls it (Reconstructor.hpp:304-306) and Localizer::localizeImage
// through its member (Localizer.hpp:93): true = resection succeeded (more than 2.5 x 3 inliers), resection_data.vec_inliers /
// error_max updated, pose set.  One context per process, created on first use.
namespace resection { enum class SolverType { DLT_6POINTS = 0, P3P_KE_CVPR17 = 1, P3P_KNEIP_CVPR11 = 2, P3P_NORDBERG_ECCV18 = 3, UP2P_KUKELOVA_ACCV10 = 4 }; }
struct HIP_SfM_Localizer {
    static bool Localize(const resection::SolverType& solver_type, const std::pair<size_t, size_t>& image_size,
                         const openMVG::cameras::Pinhole_Intrinsic_Radial_K3* optional_intrinsics,
                         openMVG::sfm::Image_Localizer_Match_Data& resection_data, openMVG::geometry::Pose3& pose, uint64_t seed = 1)
    {
        (void)image_size;                                   // only feeds the uncalibrated kernel's normalisation in OpenMVG
        if (!optional_intrinsics || solver_type == resection::SolverType::DLT_6POINTS) {
            std::cerr << "HIP_SfM_Localizer::Localize: only the calibrated P3P path is provided" << std::endl;
            return false;
        }
        static clc_ctx* shared = []() { clc_ctx* c = nullptr; return clc_ctx_create(0, nullptr, nullptr, &c) == CLC_OK ? c : nullptr; }();
        return HIPLocalizer::localize_with(shared, seed, *optional_intrinsics, resection_data, pose);
    }
};

} // namespace coloc

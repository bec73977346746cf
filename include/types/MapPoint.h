/* Header shim: TRACKING_BENCH::MapPoint as the hot-path operators and the reference's drivers use it
 * (reference include/types/MapPoint.h:19-128, src/types/MapPoint.cpp:13-44). Kept: the reference constructor
 * (MapPoint.h:22-24) with what it derives -- world position, unit viewing direction from the reference frame's camera
 * centre, the reference feature's descriptor row when the frame holds descriptors (MapPoint.cpp:36-37), the reference
 * feature / frame handles -- the observation list (AddObservation / Observations / IsInFrame / GetIndexInFrame), the bad
 * flag and the accessors the projection matchers read (matcher.cpp:406-617). Map culling, replacement and descriptor
 * voting (SetBadFlag's map erase, Replace, ComputeDistinctiveDescriptors, UpdateNormalAndDepth) are map bookkeeping
 * and stay out of scope.
 *
 * The reference declares the descriptor argument without a default, yet test/test_matcher.cpp:126 and
 * test/test_vo.cpp:267,344,829 construct with four arguments (those drivers do not compile against the
 * reference's own header). Here the argument defaults to an empty cv::Mat, so both forms build. */
#ifndef TRACKING_BENCH_MAPPOINT_H
#define TRACKING_BENCH_MAPPOINT_H
#include <map>
#include <memory>
#include <mutex>
#include <utility>
#include <vector>
#include "../tb_compat/deps.h"

namespace TRACKING_BENCH
{
    class Map;
    class Frame;
    class MapPoint;
    class Feature;

    class MapPoint
    {
    public:
        // reference MapPoint.h:22-24 (defined in the shim library: it reads the frame and the feature)
        MapPoint(const Eigen::Vector3f &Pos, std::shared_ptr<Map>&  pMap,
                 std::shared_ptr<Frame>& pFrame,
                 std::shared_ptr<Feature>&  features, cv::Mat des = cv::Mat());
        // shim-only constructors: a bare position, or position + descriptor, for callers without a map
        explicit MapPoint(const Eigen::Vector3f& Pos) : mWorldPos(Pos) {}
        MapPoint(const Eigen::Vector3f& Pos, cv::Mat des) : mWorldPos(Pos), mDescriptor(std::move(des)) {}
        // pos
        void SetWorldPos(const Eigen::Vector3f& pos) { mWorldPos = pos; }
        Eigen::Vector3f GetWorldPos() { return mWorldPos; }
        // normal
        Eigen::Vector3f GetNormal() { return mNormalVector; }
        // related frames (MapPoint.cpp:86-150)
        std::map<std::shared_ptr<Frame>, size_t> GetObservations() { return mObservations; }
        std::vector<std::shared_ptr<Feature>> GetFeatures() { return mFeatures; }
        std::shared_ptr<Feature> GetReferenceFeature() { return mpRefFeature; }
        int Observations() { return nObs; }
        int GetIndexInFrame(const std::shared_ptr<Frame>& pKF) { auto it = mObservations.find(pKF); return it == mObservations.end() ? -1 : (int)it->second; }
        bool IsInFrame(const std::shared_ptr<Frame>& pKF) { return mObservations.count(pKF) != 0; }
        void AddObservation(const std::shared_ptr<Frame>& pKF, size_t idx)
        {
            if (mObservations.count(pKF)) return;
            mObservations[pKF] = idx;
            nObs++;
        }
        // the flag only: the reference's SetBadFlag also erases the point from its frames and map (MapPoint.cpp:152-168)
        void SetBadFlag() { mbBad = true; }
        bool isBad() { return mbBad; }
        void IncreaseVisible(int n = 1) { mnVisible += n; }
        void IncreaseFound(int n = 1) { mnFound += n; }
        float GetFoundRatio() { return static_cast<float>(mnFound) / mnVisible; }
        inline int GetFound() const { return mnFound; }
        cv::Mat GetDescriptor() { return mDescriptor; }
        // constants in the reference (MapPoint.cpp:207-217)
        float GetMinDistanceInvariance() { return 1; }
        float GetMaxDistanceInvariance() { return 1000; }
        // shim-only setters for the state the reference derives in AddObservation / UpdateNormalAndDepth
        void SetObservations(int n) { nObs = n; }
        void SetNormal(const Eigen::Vector3f& normal) { mNormalVector = normal; }

        long unsigned int last_projected_id = 0;
        int n_failed_reproj = 0;
        int type = 0;
    private:
        long unsigned int mnId = 0;
        int nObs = 0;
        Eigen::Vector3f mWorldPos;
        std::vector<std::shared_ptr<Feature>> mFeatures;
        std::shared_ptr<Feature> mpRefFeature;
        std::map<std::shared_ptr<Frame>, size_t> mObservations;
        Eigen::Vector3f mNormalVector = Eigen::Vector3f::Zero();
        cv::Mat mDescriptor;
        int mnVisible = 1;
        int mnFound = 1;
        bool mbBad = false;
        float mfMinDistance = 0;
        float mfMaxDistance = 0;
        std::weak_ptr<Map> mpMap; // the reference holds a shared_ptr (a map <-> point cycle that never frees)
    };
}

#endif //TRACKING_BENCH_MAPPOINT_H

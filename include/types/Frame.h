/* Header shim: the slice of TRACKING_BENCH::Frame / Feature that the hot-path operators read and
 * write (reference include/types/Frame.h:28-203): image pyramid + scale vectors, keypoints + descriptors, pose,
 * per-feature map point and outlier flag. MapPoint and Map live in types/MapPoint.h and types/Map.h as in the
 * reference (its drivers include both, test/test_matcher.cpp:7-8); this header includes them.
 * Frame::ComputePyramid runs on the GPU through tb_pyramid. Map bookkeeping, BoW, undistortion and the
 * lookup-grid containers stay out of scope; the window matcher rebuilds the 120x36
 * grid inside tb_search_by_violence from the keypoints. */
#ifndef TRACKING_BENCH_FRAME_H
#define TRACKING_BENCH_FRAME_H
#include <memory>
#include <utility>
#include <vector>
#include "../tb_compat/deps.h"
#include "MapPoint.h"
#include "Map.h"

namespace TRACKING_BENCH
{
#define FRAME_GRID_ROWS 36
#define FRAME_GRID_COLS 120
    class Frame;
    class CameraModel;
    typedef FlatVocabulary ORBVocabulary;   // reference Frame.h:22 names DBoW2's tree type; see tb_compat/deps.h

    class Feature
    {
    public:
        std::shared_ptr<MapPoint> point;
        cv::KeyPoint kp;
        Eigen::Vector2f px;
        Eigen::Vector2f px_un;
        int idxF;
        Feature(cv::KeyPoint& _kp, int id) : kp(_kp), idxF(id) { px[0] = _kp.pt.x; px[1] = _kp.pt.y; }
    };

    class Frame
    {
    public:
        Frame(const cv::Mat &imGray, const double &timeStamp, int level, float scale, std::shared_ptr<CameraModel> camera);
        // pose (reference Frame.cpp:50-92)
        void SetPose(const Eigen::Matrix4f& Tcw);
        Eigen::Matrix4f GetPose() { return mTcw; }
        Eigen::Matrix4f GetPoseInverse() { return mTwc; }
        Eigen::Matrix3f GetRotation();
        Eigen::Vector3f GetTranslation();
        Eigen::Vector3f GetCameraCenter() { return mOw; }   // reference Frame.cpp:75
        // features (reference Frame.cpp:94-116)
        void SetKeys(std::vector<cv::KeyPoint>& pts, const std::shared_ptr<Frame>& frame, cv::Mat mDescriptors = cv::Mat(), bool unDistort = false);
        std::vector<std::shared_ptr<Feature>>& GetKeys(){return mvKeys;}
        std::shared_ptr<Feature>& GetKey(size_t id){return mvKeys.at(id);}
        cv::Mat GetDescriptors() const{return mDescriptors;}
        cv::Mat GetDescriptor(int id) const{return mDescriptors.row(id);}   // reference Frame.h:81
        // reference Frame.h:96-98, Frame.cpp:267-270: voc->transform(descriptors, mBowVec, mFeatVec, 4) -- the tree walk runs on the GPU
        void SetBow(const std::shared_ptr<ORBVocabulary>& voc);
        DBoW2::BowVector& GetBowVector(){return mBowVec;}
        DBoW2::FeatureVector& GetFeatureVector(){return mFeatVec;}
        bool GetOutlier(size_t id){return mvbOutlier.at(id) != 0;}
        void SetOutlier(size_t id, bool state){mvbOutlier.at(id)=state;}
        std::vector<float> inline GetScaleSigmaSquares() { return mvLevelSigma2; }
        std::vector<float> inline GetInverseScaleSigmaSquares() { return mvInvLevelSigma2; }
        void AssignFeaturesToGrid() {}  // the grid is rebuilt per call inside tb_search_by_violence
        // map points
        std::shared_ptr<MapPoint> GetMapPoint(const size_t &idx) { return mvpMapPoints.at(idx); }
        void AddMapPoint(std::shared_ptr<MapPoint>& pMP, const size_t& idx) { mvpMapPoints.at(idx) = pMP; }
        std::vector<std::shared_ptr<MapPoint>> GetMapPointMatches() { return mvpMapPoints; }   // reference Frame.h:104
        // key frame / pyramid
        int GetMaxLevel() const{return nLevels;}
        void ComputePyramid(cv::Mat image);
        int inline GetLevels() const{return nLevels;}
        float inline GetScaleFactor() const{return scaleFactor;}
        std::vector<float> inline GetScaleFactors(){return mvScaleFactor;}
        std::vector<float> inline GetInverseScaleFactors(){return mvInvScaleFactor;}
        std::vector<cv::Mat>& GetImagePyramid(){return mvImagePyramid;}
        cv::Mat GetImage(){return mvImagePyramid[0];}
        std::shared_ptr<CameraModel> GetCameraModel(){return mpCamera;}
    protected:
        double mTimeStamp{};
        Eigen::Matrix4f mTcw, mTwc;
        Eigen::Vector3f mOw;
        std::vector<std::shared_ptr<Feature>> mvKeys;
        cv::Mat mDescriptors;
        DBoW2::BowVector mBowVec;
        DBoW2::FeatureVector mFeatVec;
        std::vector<std::shared_ptr<MapPoint>> mvpMapPoints;
        std::shared_ptr<CameraModel> mpCamera = nullptr;
        int nLevels;
        float scaleFactor;
        std::vector<float> mvScaleFactor, mvInvScaleFactor, mvLevelSigma2, mvInvLevelSigma2;
        std::vector<cv::Mat> mvImagePyramid;
        std::vector<float> mvbOutlier;
    };
}
#endif //TRACKING_BENCH_FRAME_H

/* Header shim: TRACKING_BENCH::ORBExtractor with the reference's signatures
 * (reference include/extractors/ORBextractor.h:24-90), implemented on the C ABI (tb_orb_extract):
 * FAST cells, quadtree distribution, orientation, blur and descriptors all run as HIP kernels. */
#ifndef TRACKING_BENCH_ORBEXTRACTOR_H
#define TRACKING_BENCH_ORBEXTRACTOR_H
#include <vector>
#include "../tb_compat/deps.h"

namespace TRACKING_BENCH
{
    class ORBExtractor
    {
    public:
        enum { HARRIS_SCORE = 0, FAST_SCORE = 1 };

        ORBExtractor();
        ~ORBExtractor() = default;

        // Compute the ORB features and descriptors on an image pyramid (reference :38-44).
        void operator()(std::vector<cv::Mat> &images,
                        std::vector<float> mvScaleFactor,
                        int targetNum,
                        float initTh,
                        float minTH,
                        std::vector<cv::KeyPoint> &keypoints,
                        cv::Mat& descriptors);
        // reference :45-52; carries the per-level quotas of the last operator() call
        void AddPoints(std::vector<cv::Mat>& images,
                       std::vector<float>& mvScaleFactor,
                       int targetNum,
                       float initTh,
                       float minTH,
                       const std::vector<cv::KeyPoint> &exitPoints,
                       std::vector<cv::KeyPoint> &newPoints,
                       cv::OutputArray &descriptors);

        std::vector<float> inline GetScaleSigmaSquares() { return mvLevelSigma2; }            // never filled (reference :56-59)
        std::vector<float> inline GetInverseScaleSigmaSquares() { return mvInvLevelSigma2; }  // never filled (reference :61-64)

    protected:
        std::vector<float> mvLevelSigma2;
        std::vector<float> mvInvLevelSigma2;
        std::vector<int> mnFeaturesPerLevel;
    };
}
#endif //TRACKING_BENCH_ORBEXTRACTOR_H

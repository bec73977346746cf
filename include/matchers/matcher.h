/* Header shim: the hot-path slice of TRACKING_BENCH::Matcher with the reference's signatures
 * (reference include/matchers/matcher.h:18-80,149-150) on the C ABI (tb_search_by_bf,
 * tb_search_by_violence, tb_search_by_bow, tb_search_by_projection, tb_search_by_projection_map, tb_search_by_opflow).
 * The NN(LSH) and direct-alignment matchers are out of scope (SURVEY.md sections 2 and 8f). */
#ifndef TRACKING_BENCH_MATCHER_H
#define TRACKING_BENCH_MATCHER_H
#include <memory>
#include <vector>
#include "../tb_compat/deps.h"

namespace TRACKING_BENCH
{
    class Frame;
    class Map;

    class Matcher
    {
    public:
        Matcher() = default;
        ~Matcher() = default;
        //  match parameter (reference :23-27)
        int TH_LOW = 50;
        int TH_HIGH = 100;
        int HISTO_LENGTH = 30;
        bool checkOrientation = true;
        float nRatio{};

        // OpenCV BF (reference :39-44); only the whole-set branch is defined
        std::vector<cv::DMatch> searchByBF(
                const std::shared_ptr<Frame>& F1,
                const std::shared_ptr<Frame>& F2,
                int MinLevel, int MaxLevel,
                float ratio, float minTh,
                bool MapPointOnly = false);

        // Violence (reference :48-62)
        void setViolenceParam(int low, int high, int histo_length, bool check, float ratio)
        {
            TH_LOW = low;
            TH_HIGH = high;
            HISTO_LENGTH = histo_length;
            checkOrientation=check;
            nRatio=ratio;
        }
        std::vector<cv::DMatch> searchByViolence(
                const std::shared_ptr<Frame>& F1,
                const std::shared_ptr<Frame>& F2,
                int min_level = 0,
                int max_level = 1,
                float search_r = 10,
                bool MapPointOnly = false);

        // Projection (reference :64-80)
        void setProjectionParam(int low, int high, int histo_length, bool check, float ratio)
        {
            TH_LOW = low;
            TH_HIGH = high;
            HISTO_LENGTH = histo_length;
            checkOrientation=check;
            nRatio=ratio;
        }
        std::vector<cv::DMatch> searchByProjection(
                const std::shared_ptr<Frame>& F1,
                const std::shared_ptr<Frame>& F2);
        std::vector<cv::DMatch> searchByProjection(
                const std::shared_ptr<Map>& map,
                const std::shared_ptr<Frame>& F1, float r);

        // reference :91-94, matcher.cpp:619-721: matches inside the vocabulary nodes the two frames share
        // (F->GetFeatureVector() filled by the caller / ComputeBoW)
        std::vector<cv::DMatch> searchByBow(
                const std::shared_ptr<Frame>& F1,
                const std::shared_ptr<Frame>& F2,
                bool MapPointOnly = false);

        // Optical flow (reference :96-103, matcher.cpp:724-768). reject = true runs rejectWithF (below) before the matches
        // are listed, as the reference's caller LocalBA::AddMapPointsByStereo (LocalBA.cpp:54) asks.
        std::vector<cv::DMatch> searchByOPFlow(
                const std::shared_ptr<Frame>& F1,
                const std::shared_ptr<Frame>& F2,
                std::vector<cv::Point2f>& cur_points,
                bool equalized,
                bool reject,
                bool MapPointOnly = false);

        // reference :155, matcher.cpp:853-881: cv::findFundamentalMat(FM_RANSAC, 1.0, 0.99) clears the flags of the outliers
        // (a private member there; public here so that a caller can reach it through the class as well)
        void rejectWithF(std::vector<cv::Point2f>& pts1, const std::vector<cv::Point2f>& pts2, std::vector<uchar>& status);

        static int DescriptorDistance(const cv::Mat& a, const cv::Mat& b);
        static void ComputeThreeMaxima(std::vector<int>* histo, const int L, int &ind1, int &ind2, int &ind3);
    };
}
#endif //TRACKING_BENCH_MATCHER_H

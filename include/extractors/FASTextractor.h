/* Header shim: TRACKING_BENCH::FASTExtractor with the reference's signatures
 * (reference include/extractors/FASTextractor.h:11-41) on the C ABI (tb_fastgrid_extract). */
#ifndef TRACKING_BENCH_FASTEXTRACTOR_H
#define TRACKING_BENCH_FASTEXTRACTOR_H
#include <vector>
#include "../tb_compat/deps.h"

namespace TRACKING_BENCH
{
    class FASTExtractor
    {
    public:
        FASTExtractor();
        ~FASTExtractor() = default;
        // detection (reference :18-24): mvScaleFactor carries the INVERSE scale factors, as its callers pass
        void operator()(std::vector<cv::Mat>& images,
                        std::vector<float>& mvScaleFactor,
                        int targetNum,
                        float threshold,
                        std::vector<cv::KeyPoint> &keypoints,
                        cv::OutputArray descriptors,
                        bool reset = true);

        void AddPoints(std::vector<cv::Mat>& images,
                       std::vector<float>& mvScaleFactor,
                       int targetNum,
                       float threshold,
                       const std::vector<cv::KeyPoint> &exitPoints,
                       std::vector<cv::KeyPoint> &newPoints,
                       cv::OutputArray &descriptors);

        void resetGrid();

    private:
        std::vector<bool> grid_occupancy_;
    };
}
#endif //TRACKING_BENCH_FASTEXTRACTOR_H

/* Header shim: TRACKING_BENCH::LocalBA with the reference's signatures (reference
 * include/mapping/LocalBA.h:11-25) on the C ABI (tb_pose_opt, tb_add_map_points_by_stereo). */
#ifndef TRACKING_BENCH_LOCAL_BA_H
#define TRACKING_BENCH_LOCAL_BA_H
#include <memory>
#include <vector>
#include "../tb_compat/deps.h"

namespace TRACKING_BENCH
{
    class Frame;
    class LocalBA
    {
    public:
        LocalBA() = default;

        std::vector<float> AddMapPointsByStereo(
                const std::shared_ptr<Frame>& current_frame,
                const std::shared_ptr<Frame>& stereo_frame,
                float bf, float fx);
        // 4x4 DLT; the reference forgets to return the point (LocalBA.cpp:24-43), the shim returns it
        Eigen::Vector3f LinearTriangle(const Eigen::Vector2f& p0, const Eigen::Vector2f& p1, const Eigen::Matrix4f& Tcw0, const Eigen::Matrix4f& Tcw1);

        int PoseOptimization(std::shared_ptr<Frame>& F);
    };
}
#endif //TRACKING_BENCH_LOCAL_BA_H

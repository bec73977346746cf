/* Header shim: the slice of TRACKING_BENCH::PinholeCamera the hot path touches (reference
 * include/camera/CameraModel.h:12-89): intrinsics, pinhole projection, IsInFrame. Distortion / remap are out
 * of scope for the shim classes (host glue on OpenCV calib3d); the C ABI's tb_camera
 * carries the radial-tangential coefficients for callers that have them. */
#ifndef TRACKING_BENCH_CAMERAMODEL_H
#define TRACKING_BENCH_CAMERAMODEL_H
#include "../tb_compat/deps.h"

namespace TRACKING_BENCH
{
    class CameraModel
    {
    protected:
        int mnWidth{};
        int mnHeight{};
    public:
        CameraModel() = default;
        CameraModel(int width, int height):mnWidth(width),mnHeight(height){}
        virtual ~CameraModel() = default;
        virtual Eigen::Vector3f Cam2World(const Eigen::Vector2f& px, bool distor = false) const = 0;
        virtual Eigen::Vector2f World2Cam(const Eigen::Vector3f& xyz_c) const = 0;
        inline int Width() const {return mnWidth;}
        inline int Height() const {return mnHeight;}
        // reference CameraModel.h:33-39
        inline bool IsInFrame(const Eigen::Vector2i& obs, int boundary = 0, float scale = 1) const
        {
            if(obs[0] >= boundary && obs[0] < (int)((float)Width()  * scale) - boundary &&
               obs[1] >= boundary && obs[1] < (int)((float)Height() * scale) - boundary)
                return true;
            return false;
        }
    };

    class PinholeCamera:public CameraModel
    {
    public:
        PinholeCamera(int width, int height, float fx, float fy, float cx, float cy)
            : CameraModel(width, height), mFx(fx), mFy(fy), mCx(cx), mCy(cy) {}
        Eigen::Vector3f Cam2World(const Eigen::Vector2f& px, bool = false) const override
        {
            Eigen::Vector3f xyz;
            xyz[0] = (px[0] - mCx) / mFx; xyz[1] = (px[1] - mCy) / mFy; xyz[2] = 1.0f;
            return xyz;
        }
        Eigen::Vector2f World2Cam(const Eigen::Vector3f& p) const override
        {
            Eigen::Vector2f px;
            px[0] = mFx * p[0] / p[2] + mCx; px[1] = mFy * p[1] / p[2] + mCy;
            return px;
        }
        inline float fx() const {return mFx;}
        inline float fy() const {return mFy;}
        inline float cx() const {return mCx;}
        inline float cy() const {return mCy;}
    private:
        const float mFx, mFy, mCx, mCy;
    };
}
#endif //TRACKING_BENCH_CAMERAMODEL_H

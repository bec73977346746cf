/* Header shim: TRACKING_BENCH::Map, the containers the reference's drivers fill and the projection matcher walks
 * (reference include/types/Map.h:13-45, src/types/Map.cpp:9-100). The reference keeps points and keyframes in
 * std::set ordered by pointer value, so its GetAllMapPoints() order -- and with it DMatch::trainIdx of
 * searchByProjection(map, F1, r) -- changes from run to run; here both containers keep insertion order (duplicates
 * ignored, as a set would). Culling by age (RemoveOldFrames) and the *Safe erasers that walk the observation graph
 * are map bookkeeping, out of scope. */
#ifndef TRACKING_BENCH_MAP_H
#define TRACKING_BENCH_MAP_H
#include <algorithm>
#include <list>
#include <memory>
#include <mutex>
#include <vector>

namespace TRACKING_BENCH
{
    class MapPoint;
    class Frame;
    class Map
    {
    public:
        Map() = default;
        ~Map() = default;

        void AddKeyFrame(std::shared_ptr<Frame> pKF)
        {
            std::unique_lock<std::mutex> lock(mMutexMap);
            if (std::find(mvpKeyFrames.begin(), mvpKeyFrames.end(), pKF) == mvpKeyFrames.end()) mvpKeyFrames.push_back(std::move(pKF));
        }
        void AddMapPoint(const std::shared_ptr<MapPoint>& pMP)
        {
            std::unique_lock<std::mutex> lock(mMutexMap);
            // a set in the reference: adding a point twice keeps one entry. Checked against the last entry only (the
            // drivers add each new point once, right after creating it); a full search would make filling the map quadratic
            if (mvpMapPoints.empty() || mvpMapPoints.back() != pMP) mvpMapPoints.push_back(pMP);
        }
        void EraseMapPoint(const std::shared_ptr<MapPoint>& pMP)
        {
            std::unique_lock<std::mutex> lock(mMutexMap);
            mvpMapPoints.erase(std::remove(mvpMapPoints.begin(), mvpMapPoints.end(), pMP), mvpMapPoints.end());
        }
        void EraseKeyFrame(std::shared_ptr<Frame> pKF)
        {
            std::unique_lock<std::mutex> lock(mMutexMap);
            mvpKeyFrames.erase(std::remove(mvpKeyFrames.begin(), mvpKeyFrames.end(), pKF), mvpKeyFrames.end());
        }
        std::vector<std::shared_ptr<Frame>> GetAllKeyFrames() { std::unique_lock<std::mutex> lock(mMutexMap); return mvpKeyFrames; }
        std::vector<std::shared_ptr<MapPoint>> GetAllMapPoints() { std::unique_lock<std::mutex> lock(mMutexMap); return mvpMapPoints; }
        long unsigned int MapPointsInMap() { std::unique_lock<std::mutex> lock(mMutexMap); return mvpMapPoints.size(); }
        long unsigned int KeyFramesInMap() { std::unique_lock<std::mutex> lock(mMutexMap); return mvpKeyFrames.size(); }
        void clear()
        {
            std::unique_lock<std::mutex> lock(mMutexMap);
            mvpMapPoints.clear();
            mvpKeyFrames.clear();
        }
        std::mutex mMutexPointCreation;
        std::mutex mMutexMap;
        std::mutex mMutexMapPoints;
        std::list<std::shared_ptr<MapPoint>> mspCandidatesMapPoints;
    private:
        std::vector<std::shared_ptr<MapPoint>> mvpMapPoints;
        std::vector<std::shared_ptr<Frame>> mvpKeyFrames;
    };
}
#endif //TRACKING_BENCH_MAP_H

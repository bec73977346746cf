// The reference's test/test_matcher.cpp flow (:17-70) on the header shims, with the reference's own in-tree
// images instead of the author's EuRoC paths, plus searchByViolence (:70-72, commented out there) and
// LocalBA::PoseOptimization on synthetic map points (test/test_vo.cpp:305-355 recipe). Instead of imshow it
// dumps every result to a binary file that tests/test_gpu_shim.py compares with the CPU oracle.
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <memory>
#include <vector>

#include "camera/CameraModel.h"
#include "extractors/FASTextractor.h"
#include "extractors/ORBextractor.h"
#include "mapping/LocalBA.h"
#include "matchers/matcher.h"
#include "types/Frame.h"

using namespace TRACKING_BENCH;

static cv::Mat read_pgm(const char* path)
{
    std::ifstream f(path, std::ios::binary);
    std::string magic; int w, h, maxv;
    f >> magic >> w >> h >> maxv;
    f.get();
    cv::Mat m(h, w, CV_8UC1);
    f.read((char*)m.data, (std::streamsize)w * h);
    if (!f || magic != "P5") { std::cerr << "cannot read " << path << std::endl; std::exit(2); }
    return m;
}
template <typename T> static void put(std::ofstream& o, const T* p, size_t n) { int32_t c = (int32_t)n; o.write((char*)&c, 4); o.write((const char*)p, (std::streamsize)(n * sizeof(T))); }

int main(int argc, char** argv)
{
    if (argc < 5) { std::cerr << "usage: test_matcher_shim left.pgm right.pgm obs.bin out.bin" << std::endl; return 2; }
    cv::Mat img1 = read_pgm(argv[1]), img2 = read_pgm(argv[2]);
    auto camera_ptr = std::make_shared<PinholeCamera>(img1.cols, img1.rows, 718.856f, 718.856f, 607.1928f, 185.2157f);
    auto frame1_ptr = std::make_shared<Frame>(img1, 0, 5, 0.8, camera_ptr);
    auto frame2_ptr = std::make_shared<Frame>(img2, 0, 5, 0.8, camera_ptr);

    auto extractor_ptr = std::make_shared<ORBExtractor>();
    std::vector<cv::KeyPoint> keypoints1, keypoints2, added;
    cv::Mat descriptors1, descriptors2, added_desc;
    extractor_ptr->operator()(frame1_ptr->GetImagePyramid(), frame1_ptr->GetScaleFactors(), 1000, 80, 30, keypoints1, descriptors1);
    frame1_ptr->SetKeys(keypoints1, frame1_ptr, descriptors1);
    extractor_ptr->operator()(frame2_ptr->GetImagePyramid(), frame2_ptr->GetScaleFactors(), 1000, 80, 30, keypoints2, descriptors2);
    frame2_ptr->SetKeys(keypoints2, frame2_ptr, descriptors2);
    auto sf = frame1_ptr->GetScaleFactors();
    cv::_OutputArray added_out(added_desc);
    extractor_ptr->AddPoints(frame1_ptr->GetImagePyramid(), sf, 1000, 80, 30, keypoints1, added, added_out);

    auto matcher_ptr = std::make_shared<Matcher>();
    auto matches = matcher_ptr->searchByBF(frame1_ptr, frame2_ptr, 0, 5, 10, 30);
    frame2_ptr->AssignFeaturesToGrid();
    matcher_ptr->setViolenceParam(30, 100, 30, true, 5);
    auto vmatches = matcher_ptr->searchByViolence(frame1_ptr, frame2_ptr, 0, 5, 50);

    auto fast_ptr = std::make_shared<FASTExtractor>();
    std::vector<cv::KeyPoint> fast_kps;
    auto isf = frame1_ptr->GetInverseScaleFactors();
    fast_ptr->operator()(frame1_ptr->GetImagePyramid(), isf, 1000, 20, fast_kps, cv::noArray());

    // pose optimisation: obs.bin holds n x (u, v, X, Y, Z, octave) floats for the first n keys of frame 1
    std::ifstream ob(argv[3], std::ios::binary);
    int32_t n = 0; ob.read((char*)&n, 4);
    std::vector<float> rec((size_t)n * 6); ob.read((char*)rec.data(), (std::streamsize)rec.size() * 4);
    for (int i = 0; i < n && i < (int)keypoints1.size(); i++)
    {
        auto& f = frame1_ptr->GetKey(i);
        f->px[0] = rec[6 * i]; f->px[1] = rec[6 * i + 1];
        f->kp.octave = (int)rec[6 * i + 5];
        Eigen::Vector3f X; X[0] = rec[6 * i + 2]; X[1] = rec[6 * i + 3]; X[2] = rec[6 * i + 4];
        auto mp = std::make_shared<MapPoint>(X);
        frame1_ptr->AddMapPoint(mp, i);
    }
    frame1_ptr->SetPose(Eigen::Matrix4f::Identity());
    LocalBA localBa;
    int inliers = localBa.PoseOptimization(frame1_ptr);
    Eigen::Matrix4f T = frame1_ptr->GetPose();
    std::vector<uint8_t> outl(n);
    for (int i = 0; i < n; i++) outl[i] = frame1_ptr->GetOutlier(i) ? 1 : 0;

    std::ofstream o(argv[4], std::ios::binary);
    put(o, keypoints1.data(), keypoints1.size()); put(o, descriptors1.data, (size_t)descriptors1.rows * 32);
    put(o, keypoints2.data(), keypoints2.size()); put(o, descriptors2.data, (size_t)descriptors2.rows * 32);
    put(o, added.data(), added.size()); put(o, added_desc.data, (size_t)added_desc.rows * 32);
    put(o, matches.data(), matches.size()); put(o, vmatches.data(), vmatches.size());
    put(o, fast_kps.data(), fast_kps.size());
    float Tf[16]; for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) Tf[i * 4 + j] = T(i, j);
    put(o, Tf, 16); put(o, outl.data(), outl.size()); put(o, &inliers, 1);
    std::cout << "kps " << keypoints1.size() << "/" << keypoints2.size() << " added " << added.size() << " bf " << matches.size()
              << " violence " << vmatches.size() << " fast " << fast_kps.size() << " pose inliers " << inliers << std::endl;
    return 0;
}

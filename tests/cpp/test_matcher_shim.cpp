// The reference's test/test_matcher.cpp flow (:17-70) on the header shims, with the reference's own in-tree
// images instead of the author's EuRoC paths, plus searchByViolence (:70-72, commented out there) and
// LocalBA::PoseOptimization on synthetic map points (test/test_vo.cpp:305-355 recipe) and both searchByProjection
// overloads (matcher.cpp:406-617) on map points back-projected from frame 1. Instead of imshow it
// dumps every result to a binary file that tests/test_gpu_shim.py compares with the CPU oracle.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <memory>
#include <stdexcept>
#include <vector>

#include "camera/CameraModel.h"
#include "extractors/FASTextractor.h"
#include "extractors/ORBextractor.h"
#include "mapping/LocalBA.h"
#include "matchers/matcher.h"
#include "types/Frame.h"
#include "types/Map.h"
#include "types/MapPoint.h"
#include "tb_types.h"

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
    auto pose_map_ptr = std::make_shared<Map>();
    for (int i = 0; i < n && i < (int)keypoints1.size(); i++)
    {
        auto& f = frame1_ptr->GetKey(i);
        f->px[0] = rec[6 * i]; f->px[1] = rec[6 * i + 1];
        f->kp.octave = (int)rec[6 * i + 5];
        Eigen::Vector3f X; X[0] = rec[6 * i + 2]; X[1] = rec[6 * i + 3]; X[2] = rec[6 * i + 4];
        // the reference's four-argument construction (test/test_vo.cpp:344; the descriptor argument defaults to empty)
        auto mp = std::make_shared<MapPoint>(X, pose_map_ptr, frame1_ptr, f);
        frame1_ptr->AddMapPoint(mp, i);
        pose_map_ptr->AddMapPoint(mp);
    }
    frame1_ptr->SetPose(Eigen::Matrix4f::Identity());
    LocalBA localBa;
    int inliers = localBa.PoseOptimization(frame1_ptr);
    Eigen::Matrix4f T = frame1_ptr->GetPose();
    std::vector<uint8_t> outl(n);
    for (int i = 0; i < n; i++) outl[i] = frame1_ptr->GetOutlier(i) ? 1 : 0;

    // projection matchers (reference matcher.cpp:406-617): give frame 2's keys map points that project next to
    // frame-1 keys (back-projected through the optimised pose), mark some frame-1 points as already observed
    const Eigen::Matrix4f Twc = frame1_ptr->GetPoseInverse();
    const int n2p = std::min<int>((int)keypoints2.size(), 600), n1all = (int)keypoints1.size();
    std::vector<tb_mappoint> mp_rec((size_t)n2p);
    std::vector<uint8_t> mp_desc((size_t)n2p * 32);
    auto map_ptr = std::make_shared<Map>();
    for (int i = 0; i < n && i < n1all; i += 5) frame1_ptr->GetMapPoint(i)->SetObservations(1);
    for (int i2 = 0; i2 < n2p; i2++)
    {
        const int j = (i2 * 7) % n1all;
        const float depth = 5.f + (float)(i2 % 30), x = frame1_ptr->GetKey(j)->kp.pt.x + (float)(i2 % 5) - 2.f, y = frame1_ptr->GetKey(j)->kp.pt.y + (float)(i2 % 3) - 1.f;
        Eigen::Vector3f Pc; Pc[0] = (x - 607.1928f) / 718.856f * depth; Pc[1] = (y - 185.2157f) / 718.856f * depth; Pc[2] = depth;
        Eigen::Vector3f Pw;
        for (int a = 0; a < 3; a++) Pw[a] = Twc(a, 0) * Pc[0] + Twc(a, 1) * Pc[1] + Twc(a, 2) * Pc[2] + Twc(a, 3);
        // constructed exactly as test/test_projection.cpp:631 builds its map: position, map, the frame and the feature the
        // point was seen from, that feature's descriptor. The constructor derives the unit viewing direction from the
        // frame's camera centre and takes the feature's descriptor row (reference MapPoint.cpp:24-37)
        auto mp = std::make_shared<MapPoint>(Pw, map_ptr, frame1_ptr, frame1_ptr->GetKey(j), frame1_ptr->GetDescriptor(j));
        cv::Mat des = mp->GetDescriptor();
        if (des.rows != 1 || std::memcmp(des.ptr(0), descriptors1.ptr(j), 32) != 0) { std::cerr << "MapPoint descriptor\n"; return 3; }
        des.ptr(0)[i2 % 32] ^= (uint8_t)(i2 & 0x7); // a few flipped bits (the point's own copy of the row)
        Eigen::Vector3f nrm = mp->GetNormal();
        {
            float chk[3], nn = 0;
            for (int a = 0; a < 3; a++) { chk[a] = Pw[a] - Twc(a, 3); nn += chk[a] * chk[a]; }
            nn = std::sqrt(nn);
            for (int a = 0; a < 3; a++) if (std::fabs(chk[a] / nn - nrm[a]) > 1e-6f) { std::cerr << "MapPoint normal\n"; return 3; }
        }
        if (i2 % 4 == 1) nrm[0] = -nrm[0]; // some points face away
        mp->SetNormal(nrm);
        if (i2 % 11 == 3) mp->SetBadFlag();
        if (i2 % 13 != 5) frame2_ptr->AddMapPoint(mp, i2); // some keys keep no map point
        map_ptr->AddMapPoint(mp);
        std::memset(&mp_rec[i2], 0, sizeof(tb_mappoint));
        for (int a = 0; a < 3; a++) { mp_rec[i2].pos[a] = Pw[a]; mp_rec[i2].normal[a] = nrm[a]; }
        mp_rec[i2].min_dist = mp->GetMinDistanceInvariance(); mp_rec[i2].max_dist = mp->GetMaxDistanceInvariance();
        mp_rec[i2].bad = mp->isBad() ? 1 : 0;
        std::memcpy(mp_desc.data() + (size_t)i2 * 32, des.ptr(0), 32);
    }
    matcher_ptr->setProjectionParam(30, 100, 30, true, 8);
    auto pmatches = matcher_ptr->searchByProjection(frame1_ptr, frame2_ptr);
    matcher_ptr->setProjectionParam(30, 100, 30, true, 3);
    auto mmatches = matcher_ptr->searchByProjection(map_ptr, frame1_ptr, 0.8f);
    std::vector<tb_keypoint> k1now((size_t)n1all);
    std::vector<uint8_t> taken1((size_t)n1all, 0), nomp2((size_t)n2p, 0);
    for (int i = 0; i < n1all; i++)
    {
        std::memcpy(&k1now[i], &frame1_ptr->GetKey(i)->kp, sizeof(tb_keypoint));
        auto p = frame1_ptr->GetMapPoint(i);
        taken1[i] = (p && p->Observations() > 0) ? 1 : 0;
    }
    for (int i2 = 0; i2 < n2p; i2++) nomp2[i2] = frame2_ptr->GetMapPoint(i2) ? 0 : 1;

    // LocalBA::AddMapPointsByStereo's matcher call (LocalBA.cpp:54): the left frame's keys tracked into the CLAHE-equalised
    // right image, first without, then with the RANSAC stage (reject = true), then the function itself (test_vo.cpp:716)
    std::vector<cv::Point2f> flow_pts, flow_pts_r;
    auto fmatches = matcher_ptr->searchByOPFlow(frame2_ptr, frame1_ptr, flow_pts, true, false);
    auto rmatches = matcher_ptr->searchByOPFlow(frame2_ptr, frame1_ptr, flow_pts_r, true, true);
    std::vector<float> depths = localBa.AddMapPointsByStereo(frame1_ptr, frame2_ptr, 386.1448f, 718.856f);

    // Frame::SetBow (Frame.cpp:267-270) as test/test_vo.cpp:661,705 calls it, then Matcher::searchByBow (matcher.cpp:619-721).
    // The reference tree ships no vocabulary file: argv[5] is a seeded synthetic one in the ORBvoc text format
    auto vocabulary = std::make_shared<ORBVocabulary>();
    if (!vocabulary->loadFromTextFile(argv[5])) { std::cerr << "vocabulary\n"; return 4; }
    frame1_ptr->SetBow(vocabulary);
    frame2_ptr->SetBow(vocabulary);
    std::vector<uint32_t> fv1_flat;   // node, count, indices ... of frame 1's FeatureVector
    for (const auto& kv : frame1_ptr->GetFeatureVector())
    {
        fv1_flat.push_back(kv.first); fv1_flat.push_back((uint32_t)kv.second.size());
        fv1_flat.insert(fv1_flat.end(), kv.second.begin(), kv.second.end());
    }
    std::vector<uint32_t> bv1_ids; std::vector<double> bv1_vals;
    for (const auto& kv : frame1_ptr->GetBowVector()) { bv1_ids.push_back(kv.first); bv1_vals.push_back(kv.second); }
    matcher_ptr->setViolenceParam(80, 100, 30, true, 0.95f);
    auto bmatches = matcher_ptr->searchByBow(frame1_ptr, frame2_ptr);

    std::ofstream o(argv[4], std::ios::binary);
    put(o, keypoints1.data(), keypoints1.size()); put(o, descriptors1.data, (size_t)descriptors1.rows * 32);
    put(o, keypoints2.data(), keypoints2.size()); put(o, descriptors2.data, (size_t)descriptors2.rows * 32);
    put(o, added.data(), added.size()); put(o, added_desc.data, (size_t)added_desc.rows * 32);
    put(o, matches.data(), matches.size()); put(o, vmatches.data(), vmatches.size());
    put(o, fast_kps.data(), fast_kps.size());
    float Tf[16]; for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) Tf[i * 4 + j] = T(i, j);
    put(o, Tf, 16); put(o, outl.data(), outl.size()); put(o, &inliers, 1);
    put(o, k1now.data(), k1now.size()); put(o, taken1.data(), taken1.size()); put(o, nomp2.data(), nomp2.size());
    put(o, mp_rec.data(), mp_rec.size()); put(o, mp_desc.data(), mp_desc.size());
    put(o, pmatches.data(), pmatches.size()); put(o, mmatches.data(), mmatches.size());
    put(o, flow_pts.data(), flow_pts.size()); put(o, fmatches.data(), fmatches.size());
    put(o, rmatches.data(), rmatches.size()); put(o, depths.data(), depths.size());
    put(o, bmatches.data(), bmatches.size());
    put(o, fv1_flat.data(), fv1_flat.size()); put(o, bv1_ids.data(), bv1_ids.size()); put(o, bv1_vals.data(), bv1_vals.size());
    std::cout << "kps " << keypoints1.size() << "/" << keypoints2.size() << " added " << added.size() << " bf " << matches.size()
              << " violence " << vmatches.size() << " fast " << fast_kps.size() << " pose inliers " << inliers
              << " projection " << pmatches.size() << " map projection " << mmatches.size() << " flow " << fmatches.size()
              << " bow " << bmatches.size() << " flow+ransac " << rmatches.size() << " depths " << depths.size() << std::endl;
    return 0;
}

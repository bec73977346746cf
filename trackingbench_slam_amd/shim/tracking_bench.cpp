/* Implementation of the header shims (include/extractors, include/matchers, include/mapping,
 * include/types): thin adapters that unpack cv::Mat / cv::KeyPoint / cv::DMatch into the C ABI of
 * libtb_hip.so. No arithmetic of the hot path happens here; errors of the C ABI become exceptions
 * (the reference has no error convention: empty inputs return silently).
 */
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <stdexcept>
#include <string>

#include "tb_capi.h"
#include "camera/CameraModel.h"
#include "extractors/FASTextractor.h"
#include "extractors/ORBextractor.h"
#include "mapping/LocalBA.h"
#include "matchers/matcher.h"
#include "types/Frame.h"
#include "types/Map.h"
#include "types/MapPoint.h"

namespace TRACKING_BENCH
{
    static tb_ctx* shim_ctx()
    {
        static tb_ctx* ctx = nullptr;
        if (!ctx)
        {
            const char* dev = std::getenv("TB_DEVICE");
            int rc = tb_create(dev ? std::atoi(dev) : 0, &ctx);
            if (rc) throw std::runtime_error(std::string("tracking_bench: no usable MI355X device: ") + tb_strerror(rc));
        }
        return ctx;
    }
    static void check(int rc, const char* what)
    {
        if (rc) throw std::runtime_error(std::string(what) + ": " + tb_last_error(shim_ctx()));
    }
    static_assert(sizeof(cv::KeyPoint) == sizeof(tb_keypoint), "cv::KeyPoint must match tb_keypoint");
    static_assert(sizeof(cv::DMatch) == sizeof(tb_match), "cv::DMatch must match tb_match");

    struct LevelArgs
    {
        std::vector<const uint8_t*> ptr;
        std::vector<int> w, h, s;
        explicit LevelArgs(std::vector<cv::Mat>& images)
        {
            for (auto& m : images) { ptr.push_back(m.data); w.push_back(m.cols); h.push_back(m.rows); s.push_back((int)m.step); }
        }
    };

    /* ------------------------------------------------------------------ Frame */
    Frame::Frame(const cv::Mat &imGray, const double &timeStamp, const int level, const float scale,
                 std::shared_ptr<CameraModel> camera) : mTimeStamp(timeStamp), mpCamera(std::move(camera)), nLevels(level), scaleFactor(scale)
    {
        mvScaleFactor.resize(nLevels, 1); mvInvScaleFactor.resize(nLevels, 1);
        mvLevelSigma2.resize(nLevels, 1); mvInvLevelSigma2.resize(nLevels, 1);
        tb_scale_factors(nLevels, scale, mvScaleFactor.data(), mvInvScaleFactor.data(), mvLevelSigma2.data(), mvInvLevelSigma2.data());
        mTcw = Eigen::Matrix4f::Identity();
        mTwc = Eigen::Matrix4f::Identity();
        ComputePyramid(imGray);
    }

    void Frame::ComputePyramid(cv::Mat image)
    {
        /* Frame.cpp:414-427: level 0 aliases the input; levels >= 1 come from the GPU resize chain */
        mvImagePyramid.assign(1, image);
        std::vector<int> ws(nLevels), hs(nLevels), st(nLevels);
        tb_pyramid_sizes(image.cols, image.rows, nLevels, mvScaleFactor.data(), ws.data(), hs.data());
        std::vector<uint8_t*> out(nLevels, nullptr);
        for (int i = 1; i < nLevels; i++)
        {
            mvImagePyramid.emplace_back(hs[i], ws[i], CV_8UC1);
            out[i] = mvImagePyramid[i].data;
            st[i] = (int)mvImagePyramid[i].step;
        }
        check(tb_pyramid(shim_ctx(), image.data, image.cols, image.rows, (int)image.step, nLevels, mvScaleFactor.data(), out.data(), st.data()),
              "Frame::ComputePyramid");
    }

    void Frame::SetPose(const Eigen::Matrix4f& Tcw)
    {
        /* Frame.cpp:50-61 */
        mTcw = Tcw;
        Eigen::Matrix3f Rwc;
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Rwc(i, j) = Tcw(j, i);
        for (int i = 0; i < 3; i++) mOw[i] = -(Rwc(i, 0) * Tcw(0, 3) + Rwc(i, 1) * Tcw(1, 3) + Rwc(i, 2) * Tcw(2, 3));
        mTwc = Eigen::Matrix4f::Identity();
        for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) mTwc(i, j) = Rwc(i, j); mTwc(i, 3) = mOw[i]; }
    }
    Eigen::Matrix3f Frame::GetRotation() { Eigen::Matrix3f R; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R(i, j) = mTwc(i, j); return R; }
    Eigen::Vector3f Frame::GetTranslation() { Eigen::Vector3f t; for (int i = 0; i < 3; i++) t[i] = mTwc(i, 3); return t; }

    void Frame::SetKeys(std::vector<cv::KeyPoint>& pts, const std::shared_ptr<Frame>&, cv::Mat descriptors, bool)
    {
        mDescriptors = std::move(descriptors);
        mvKeys.clear();
        mvKeys.reserve(pts.size());
        for (auto& pt : pts) mvKeys.emplace_back(std::make_shared<Feature>(pt, (int)mvKeys.size()));
        mvpMapPoints.assign(pts.size(), nullptr);
        mvbOutlier.assign(pts.size(), false);
    }

    /* ------------------------------------------------------------------ MapPoint */
    /* reference MapPoint.cpp:13-44: position, unit viewing direction from the reference frame's camera centre, the
     * scale-invariance distances, the reference feature's descriptor row when the frame holds descriptors (it replaces the
     * `des` argument, MapPoint.cpp:36-37), the id under the map's creation mutex; the reference then drops its frame handle
     * (mpRefKF = nullptr, :42). Element-wise arithmetic: the same source builds against Eigen and the stand-in types. */
    MapPoint::MapPoint(const Eigen::Vector3f &Pos, std::shared_ptr<Map>& pMap, std::shared_ptr<Frame>& pFrame,
                       std::shared_ptr<Feature>& features, cv::Mat des)
        : mWorldPos(Pos), mpRefFeature(features), mDescriptor(std::move(des)), mpMap(pMap)
    {
        mFeatures.emplace_back(mpRefFeature);
        const Eigen::Vector3f Ow = pFrame->GetCameraCenter();
        float d[3], n2 = 0;
        for (int i = 0; i < 3; i++) { d[i] = mWorldPos[i] - Ow[i]; n2 += d[i] * d[i]; }
        const float dist = std::sqrt(n2);
        for (int i = 0; i < 3; i++) mNormalVector[i] = d[i] / dist;
        const int level = mpRefFeature->kp.octave;
        const std::vector<float> sf = pFrame->GetScaleFactors();
        mfMaxDistance = dist * sf.at(level);
        mfMinDistance = mfMaxDistance / sf.at(pFrame->GetLevels() - 1);
        if (!pFrame->GetDescriptors().empty()) mDescriptor = pFrame->GetDescriptors().row(mpRefFeature->idxF).clone();
        static long unsigned int nNextId = 0;
        if (pMap) { std::unique_lock<std::mutex> lock(pMap->mMutexPointCreation); mnId = nNextId++; }
        else mnId = nNextId++;
    }

    /* ------------------------------------------------------------------ vocabulary / Frame::SetBow */
    FlatVocabulary::~FlatVocabulary() { if (device) tb_vocab_destroy(device); }

    bool FlatVocabulary::loadFromTextFile(const std::string& filename)
    {
        /* TemplatedVocabulary.h:1338-1420: "k L scoring weighting", then per node (ids in file order from 1): parent, is-leaf,
         * 32 descriptor bytes, weight; words are numbered in file order */
        std::ifstream f(filename.c_str());
        if (!f.good()) return false;
        std::string line;
        if (!std::getline(f, line)) return false;
        {
            std::stringstream ss(line);
            ss >> k >> L >> scoring >> weighting;
            if (ss.fail() || k < 0 || k > 20 || L < 1 || L > 10 || scoring < 0 || scoring > 5 || weighting < 0 || weighting > 3) return false;
        }
        std::vector<int> parent(1, 0), leaf(1, 0);
        desc.assign(32, 0); weight.assign(1, 0.0);
        while (std::getline(f, line))
        {
            if (line.empty()) continue;
            std::stringstream ss(line);
            int pid = 0, isleaf = 0;
            ss >> pid >> isleaf;
            if (ss.fail() || pid < 0 || pid >= (int)parent.size()) return false;
            parent.push_back(pid); leaf.push_back(isleaf > 0);
            for (int i = 0; i < 32; i++) { int v = 0; ss >> v; desc.push_back((uint8_t)v); }
            double w = 0; ss >> w;
            if (ss.fail()) return false;
            weight.push_back(w);
        }
        const int nn = (int)parent.size();
        std::vector<int> cnt(nn + 1, 0);
        for (int n = 1; n < nn; n++) cnt[parent[n] + 1]++;
        child_start.assign(nn + 1, 0);
        for (int n = 0; n < nn; n++) child_start[n + 1] = child_start[n] + cnt[n + 1];
        child_items.assign(std::max(nn - 1, 0), 0);
        std::vector<int> at(child_start.begin(), child_start.end() - 1);
        for (int n = 1; n < nn; n++) child_items[at[parent[n]]++] = n;      /* children in file order, as push_back does */
        word_id.assign(nn, 0);
        nwords = 0;
        for (int n = 1; n < nn; n++) if (leaf[n]) word_id[n] = (int32_t)nwords++;
        if (device) { tb_vocab_destroy(device); device = nullptr; }
        return true;
    }

    void Frame::SetBow(const std::shared_ptr<ORBVocabulary>& voc)
    {
        /* Frame.cpp:267-270 / TemplatedVocabulary.h:1124-1188 */
        mBowVec.clear(); mFeatVec.clear();
        if (!voc || voc->empty()) return;
        if (!voc->device)
        {
            tb_vocabulary h;
            h.nnodes = (int32_t)voc->word_id.size(); h.k = voc->k; h.L = voc->L; h.weighting = voc->weighting; h.scoring = voc->scoring;
            h.child_start = voc->child_start.data(); h.child_items = voc->child_items.data(); h.desc = voc->desc.data();
            h.word_id = voc->word_id.data(); h.weight = voc->weight.data();
            check(tb_vocab_create(shim_ctx(), &h, &voc->device), "Frame::SetBow (vocabulary upload)");
        }
        const int n = mDescriptors.rows;
        if (n == 0) return;
        std::vector<uint8_t> d((size_t)n * 32);
        for (int i = 0; i < n; i++) std::memcpy(d.data() + (size_t)i * 32, mDescriptors.ptr(i), 32);
        std::vector<int32_t> wid(n), nid(n);
        std::vector<double> wt(n);
        check(tb_bow_transform(shim_ctx(), voc->device, d.data(), n, 4, wid.data(), wt.data(), nid.data()), "Frame::SetBow");
        const bool tf = voc->weighting == 0 || voc->weighting == 1;         /* TF_IDF, TF: weights add up; IDF, BINARY: first one */
        for (int i = 0; i < n; i++)
        {
            if (!(wt[i] > 0)) continue;                                      /* a stopped word */
            if (tf) mBowVec[(DBoW2::WordId)wid[i]] += wt[i];
            else mBowVec.insert(std::make_pair((DBoW2::WordId)wid[i], wt[i]));
            mFeatVec.addFeature((DBoW2::NodeId)nid[i], (unsigned)i);
        }
        const bool must = voc->scoring != 5, l2 = voc->scoring == 1;        /* ScoringObject.h:73-91 */
        if (!mBowVec.empty() && !must && tf)
        {
            const double nd = (double)mBowVec.size();
            for (auto& kv : mBowVec) kv.second /= nd;
        }
        if (must)
        {
            double norm = 0.0;                                              /* BowVector::normalize, BowVector.cpp:57-80 */
            for (auto& kv : mBowVec) norm += l2 ? kv.second * kv.second : std::fabs(kv.second);
            if (l2) norm = std::sqrt(norm);
            if (norm > 0.0) for (auto& kv : mBowVec) kv.second /= norm;
        }
    }

    /* ------------------------------------------------------------------ extractors */
    ORBExtractor::ORBExtractor() = default;

    void ORBExtractor::operator()(std::vector<cv::Mat>& images, std::vector<float> mvScaleFactor, int targetNum, float initTh,
                                  float minTH, std::vector<cv::KeyPoint>& keyPoints, cv::Mat& _descriptors)
    {
        if (images.empty() || images.at(0).empty()) return; /* ORBextractor.cpp:914-915 */
        LevelArgs a(images);
        mnFeaturesPerLevel.assign(images.size(), 0);
        const int cap = targetNum + 64 * (int)images.size() + 64;
        keyPoints.assign(cap, cv::KeyPoint());
        cv::Mat desc(cap, 32, CV_8U);
        int n = 0;
        check(tb_orb_extract(shim_ctx(), a.ptr.data(), a.w.data(), a.h.data(), a.s.data(), (int)images.size(), mvScaleFactor.data(),
                             targetNum, initTh, minTH, nullptr, 0, 0, mnFeaturesPerLevel.data(),
                             reinterpret_cast<tb_keypoint*>(keyPoints.data()), desc.data, cap, &n), "ORBExtractor::operator()");
        keyPoints.resize(n);
        if (n == 0) { _descriptors.release(); return; }
        _descriptors.create(n, 32, CV_8U);
        for (int i = 0; i < n; i++) std::memcpy(_descriptors.ptr(i), desc.ptr(i), 32);
    }

    void ORBExtractor::AddPoints(std::vector<cv::Mat>& images, std::vector<float>& mvScaleFactor, int targetNum, float initTh,
                                 float minTH, const std::vector<cv::KeyPoint>& exitPoints, std::vector<cv::KeyPoint>& newPoints,
                                 cv::OutputArray& _descriptors)
    {
        if (images.empty() || images.at(0).empty()) return;
        if (mnFeaturesPerLevel.size() != images.size())
            throw std::logic_error("ORBExtractor::AddPoints before operator(): the reference indexes an empty mnFeaturesPerLevel");
        LevelArgs a(images);
        int qsum = 0;
        for (int q : mnFeaturesPerLevel) qsum += q;
        const int cap = std::max(targetNum, qsum) + 64 * (int)images.size() + 64;
        newPoints.assign(cap, cv::KeyPoint());
        cv::Mat desc(cap, 32, CV_8U);
        int n = 0;
        check(tb_orb_extract(shim_ctx(), a.ptr.data(), a.w.data(), a.h.data(), a.s.data(), (int)images.size(), mvScaleFactor.data(),
                             targetNum, initTh, minTH, reinterpret_cast<const tb_keypoint*>(exitPoints.data()), (int)exitPoints.size(), 1,
                             mnFeaturesPerLevel.data(), reinterpret_cast<tb_keypoint*>(newPoints.data()), desc.data, cap, &n),
              "ORBExtractor::AddPoints");
        newPoints.resize(n);
        if (n == 0) { _descriptors.release(); return; }
        _descriptors.create(n, 32, CV_8U);
        cv::Mat d = _descriptors.getMat();
        for (int i = 0; i < n; i++) std::memcpy(d.ptr(i), desc.ptr(i), 32);
    }

    FASTExtractor::FASTExtractor() = default;

    void FASTExtractor::operator()(std::vector<cv::Mat>& images, std::vector<float>& invScaleFactor, int nFeatures, float threshold,
                                   std::vector<cv::KeyPoint>& keyPoints, cv::OutputArray, bool reset)
    {
        if (images.empty() || images.at(0).empty()) return; /* FASTextractor.cpp:16-17 */
        const int cell_size = (int)sqrtf((float)images.at(0).cols * (float)images.at(0).rows / (float)nFeatures);
        const int grid_n_cols = (int)((float)images.at(0).cols / (float)cell_size);
        const int grid_n_rows = (int)((float)images.at(0).rows / (float)cell_size);
        if (reset) grid_occupancy_.resize((size_t)grid_n_cols * grid_n_rows, false);
        std::vector<uint8_t> occ(grid_occupancy_.size());
        for (size_t i = 0; i < occ.size(); i++) occ[i] = grid_occupancy_[i] ? 1 : 0;
        LevelArgs a(images);
        const int cap = 2 * nFeatures + 4096;
        keyPoints.assign(cap, cv::KeyPoint());
        int n = 0;
        check(tb_fastgrid_extract(shim_ctx(), a.ptr.data(), a.w.data(), a.h.data(), a.s.data(), (int)images.size(), invScaleFactor.data(),
                                  nFeatures, threshold, occ.empty() ? nullptr : occ.data(), (int)occ.size(),
                                  reinterpret_cast<tb_keypoint*>(keyPoints.data()), cap, &n), "FASTExtractor::operator()");
        keyPoints.resize(n);
        resetGrid();
    }

    void FASTExtractor::resetGrid() { std::fill(grid_occupancy_.begin(), grid_occupancy_.end(), false); }

    void FASTExtractor::AddPoints(std::vector<cv::Mat>& images, std::vector<float>& mvScaleFactor, int nFeatures, float threshold,
                                  const std::vector<cv::KeyPoint>& exitPoints, std::vector<cv::KeyPoint>& newPoints,
                                  cv::OutputArray& descriptors)
    {
        /* FASTextractor.cpp:129-150 */
        const int cell_size = (int)sqrtf((float)images.at(0).cols * (float)images.at(0).rows / (float)nFeatures);
        const int grid_n_cols = (int)((float)images.at(0).cols / (float)cell_size);
        const int grid_n_rows = (int)((float)images.at(0).rows / (float)cell_size);
        grid_occupancy_.resize((size_t)grid_n_cols * grid_n_rows, false);
        for (const auto& i : exitPoints)
            grid_occupancy_.at(static_cast<int>(i.pt.y / (float)cell_size) * grid_n_cols + static_cast<int>(i.pt.x / (float)cell_size)) = true;
        operator()(images, mvScaleFactor, nFeatures, threshold, newPoints, descriptors, false);
    }

    /* ------------------------------------------------------------------ matcher */
    static void frame_keys(const std::shared_ptr<Frame>& F, std::vector<tb_keypoint>& k)
    {
        k.resize(F->GetKeys().size());
        for (size_t i = 0; i < k.size(); i++) std::memcpy(&k[i], &F->GetKeys()[i]->kp, sizeof(tb_keypoint));
    }
    static void frame_desc(const std::shared_ptr<Frame>& F, std::vector<uint8_t>& d)
    {
        cv::Mat m = F->GetDescriptors();
        d.resize((size_t)m.rows * 32);
        for (int i = 0; i < m.rows; i++) std::memcpy(d.data() + (size_t)i * 32, m.ptr(i), 32);
    }

    std::vector<cv::DMatch> Matcher::searchByBF(const std::shared_ptr<Frame>& F1, const std::shared_ptr<Frame>& F2, int MinLevel,
                                                int MaxLevel, float ratio, float minTh, bool MapPointOnly)
    {
        /* matcher.cpp:178-203: only the whole-set branch is well defined (the sub-range branch writes rows of an
         * empty cv::Mat and remaps ids on a copy) */
        if (!(MinLevel == 0 && MaxLevel == F1->GetMaxLevel() && !MapPointOnly))
            throw std::invalid_argument("Matcher::searchByBF: level sub-range / MapPointOnly branch is undefined in the reference");
        std::vector<uint8_t> d1, d2;
        frame_desc(F1, d1);
        frame_desc(F2, d2);
        const int n1 = (int)(d1.size() / 32), n2 = (int)(d2.size() / 32);
        std::vector<cv::DMatch> out((size_t)std::max(n1, 1));
        int n = 0;
        check(tb_search_by_bf(shim_ctx(), d1.data(), n1, d2.data(), n2, ratio, minTh, reinterpret_cast<tb_match*>(out.data()), (int)out.size(), &n),
              "Matcher::searchByBF");
        out.resize(n);
        return out;
    }

    std::vector<cv::DMatch> Matcher::searchByViolence(const std::shared_ptr<Frame>& F1, const std::shared_ptr<Frame>& F2, int min_level,
                                                      int max_level, float search_r, bool MapPointOnly)
    {
        if (MapPointOnly) throw std::invalid_argument("Matcher::searchByViolence: MapPointOnly is outside the hot-path scope");
        std::vector<tb_keypoint> k1, k2;
        std::vector<uint8_t> d1, d2;
        frame_keys(F1, k1); frame_keys(F2, k2);
        frame_desc(F1, d1); frame_desc(F2, d2);
        std::vector<cv::DMatch> out(std::max<size_t>(k1.size(), 1));
        int n = 0;
        cv::Mat img2 = F2->GetImage();
        check(tb_search_by_violence(shim_ctx(), k1.data(), d1.data(), (int)k1.size(), k2.data(), d2.data(), (int)k2.size(), img2.cols, img2.rows,
                                    min_level, max_level, search_r, TH_LOW, nRatio, HISTO_LENGTH, checkOrientation ? 1 : 0,
                                    reinterpret_cast<tb_match*>(out.data()), (int)out.size(), &n), "Matcher::searchByViolence");
        out.resize(n);
        return out;
    }

    /* Matcher::searchByBow (reference matcher.cpp:619-721) on tb_search_by_bow */
    static void frame_fv(const std::shared_ptr<Frame>& F, std::vector<uint32_t>& nodes, std::vector<int32_t>& start, std::vector<uint32_t>& items)
    {
        nodes.clear(); items.clear(); start.assign(1, 0);
        for (const auto& kv : F->GetFeatureVector())       /* std::map: ascending node ids */
        {
            nodes.push_back(kv.first);
            items.insert(items.end(), kv.second.begin(), kv.second.end());
            start.push_back((int32_t)items.size());
        }
    }
    std::vector<cv::DMatch> Matcher::searchByBow(const std::shared_ptr<Frame>& F1, const std::shared_ptr<Frame>& F2, bool MapPointOnly)
    {
        std::vector<tb_keypoint> k1, k2;
        std::vector<uint8_t> d1, d2, has2;
        frame_keys(F1, k1); frame_keys(F2, k2);
        frame_desc(F1, d1); frame_desc(F2, d2);
        std::vector<uint32_t> n1, i1, n2, i2;
        std::vector<int32_t> s1, s2;
        frame_fv(F1, n1, s1, i1); frame_fv(F2, n2, s2, i2);
        if (MapPointOnly)
        {
            has2.assign(std::max<size_t>(k2.size(), 1), 0);
            for (size_t i = 0; i < k2.size(); i++) has2[i] = F2->GetMapPoint(i) ? 1 : 0;
        }
        std::vector<cv::DMatch> out(std::max<size_t>(i1.size(), 1));
        int n = 0;
        check(tb_search_by_bow(shim_ctx(), k1.data(), d1.data(), (int)k1.size(), n1.data(), s1.data(), i1.data(), (int)n1.size(), k2.data(), d2.data(),
                               (int)k2.size(), MapPointOnly ? has2.data() : nullptr, n2.data(), s2.data(), i2.data(), (int)n2.size(),
                               MapPointOnly ? 1 : 0, TH_LOW, nRatio, HISTO_LENGTH, checkOrientation ? 1 : 0,
                               reinterpret_cast<tb_match*>(out.data()), (int)out.size(), &n), "Matcher::searchByBow");
        out.resize(n);
        return out;
    }

    /* Matcher::searchByOPFlow (reference matcher.cpp:724-768) on tb_search_by_opflow; reject = true runs
     * Matcher::rejectWithF (cv::findFundamentalMat RANSAC restated, parity unpinned) on the device as well */
    std::vector<cv::DMatch> Matcher::searchByOPFlow(const std::shared_ptr<Frame>& F1, const std::shared_ptr<Frame>& F2,
                                                    std::vector<cv::Point2f>& cur_points, bool equalized, bool reject, bool MapPointOnly)
    {
        (void)MapPointOnly; /* the reference ignores it as well */
        cv::Mat img1 = F1->GetImage(), img2 = F2->GetImage();
        if (img1.cols != img2.cols || img1.rows != img2.rows) throw std::invalid_argument("Matcher::searchByOPFlow: image sizes differ");
        std::vector<tb_keypoint> k2;
        frame_keys(F2, k2);
        std::vector<float> xy(2 * std::max<size_t>(k2.size(), 1)), cur(2 * std::max<size_t>(k2.size(), 1));
        for (size_t i = 0; i < k2.size(); i++) { xy[2 * i] = k2[i].x; xy[2 * i + 1] = k2[i].y; }
        tb_camera cam;
        std::memset(&cam, 0, sizeof cam);
        cam.width = F1->GetCameraModel()->Width(); cam.height = F1->GetCameraModel()->Height();
        std::vector<cv::DMatch> out(std::max<size_t>(k2.size(), 1));
        int n = 0;
        check(tb_search_by_opflow(shim_ctx(), img1.data, img2.data, img1.cols, img1.rows, (int)img1.step, &cam, xy.data(), (int)k2.size(), equalized ? 1 : 0, reject ? 1 : 0,
                                  cur.data(), reinterpret_cast<tb_match*>(out.data()), (int)out.size(), &n), "Matcher::searchByOPFlow");
        cur_points.resize(k2.size());
        for (size_t i = 0; i < k2.size(); i++) cur_points[i] = cv::Point2f(cur[2 * i], cur[2 * i + 1]);
        out.resize(n);
        return out;
    }

    /* Matcher::rejectWithF (reference matcher.cpp:853-881) on tb_reject_with_f */
    void Matcher::rejectWithF(std::vector<cv::Point2f>& cur_pts, const std::vector<cv::Point2f>& last_pts, std::vector<uchar>& status)
    {
        const size_t n = status.size();
        if (cur_pts.size() < n || last_pts.size() < n) throw std::out_of_range("Matcher::rejectWithF: fewer points than status flags");
        std::vector<float> a(2 * std::max<size_t>(n, 1)), b(2 * std::max<size_t>(n, 1));
        for (size_t i = 0; i < n; i++) { a[2 * i] = cur_pts[i].x; a[2 * i + 1] = cur_pts[i].y; b[2 * i] = last_pts[i].x; b[2 * i + 1] = last_pts[i].y; }
        check(tb_reject_with_f(shim_ctx(), a.data(), b.data(), (int)n, status.data()), "Matcher::rejectWithF");
    }

    /* the projection matchers (reference matcher.cpp:406-617) on tb_search_by_projection[_map] */
    static void frame_projection_inputs(const std::shared_ptr<Frame>& F1, float Tcw[16], tb_camera& cam, std::vector<tb_keypoint>& k1,
                                        std::vector<uint8_t>& d1, std::vector<uint8_t>& taken1)
    {
        const Eigen::Matrix4f T = F1->GetPose();
        for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) Tcw[i * 4 + j] = T(i, j);
        auto pin = std::dynamic_pointer_cast<PinholeCamera>(F1->GetCameraModel());
        if (!pin) throw std::invalid_argument("Matcher::searchByProjection: F1 needs a PinholeCamera");
        std::memset(&cam, 0, sizeof cam);
        cam.fx = pin->fx(); cam.fy = pin->fy(); cam.cx = pin->cx(); cam.cy = pin->cy();
        cam.width = pin->Width(); cam.height = pin->Height();
        frame_keys(F1, k1);
        frame_desc(F1, d1);
        taken1.assign(k1.size(), 0);
        for (size_t i = 0; i < k1.size(); i++)
        {
            auto p = F1->GetMapPoint(i);
            if (p && p->Observations() > 0) taken1[i] = 1;
        }
    }
    static void mappoint_record(const std::shared_ptr<MapPoint>& p, tb_mappoint& r, uint8_t* desc)
    {
        std::memset(&r, 0, sizeof r);
        std::memset(desc, 0, 32);
        if (!p || p->isBad()) { r.bad = 1; return; }
        const Eigen::Vector3f X = p->GetWorldPos(), n = p->GetNormal();
        for (int i = 0; i < 3; i++) { r.pos[i] = X[i]; r.normal[i] = n[i]; }
        r.min_dist = p->GetMinDistanceInvariance();
        r.max_dist = p->GetMaxDistanceInvariance();
        cv::Mat d = p->GetDescriptor();
        if (d.rows > 0) std::memcpy(desc, d.ptr(0), 32);
    }

    std::vector<cv::DMatch> Matcher::searchByProjection(const std::shared_ptr<Frame>& F1, const std::shared_ptr<Frame>& F2)
    {
        float Tcw[16];
        tb_camera cam;
        std::vector<tb_keypoint> k1, k2;
        std::vector<uint8_t> d1, taken1;
        frame_projection_inputs(F1, Tcw, cam, k1, d1, taken1);
        frame_keys(F2, k2);
        std::vector<tb_mappoint> mp(std::max<size_t>(k2.size(), 1));
        std::vector<uint8_t> md(std::max<size_t>(k2.size(), 1) * 32);
        for (size_t i = 0; i < k2.size(); i++) mappoint_record(F2->GetMapPoint(i), mp[i], md.data() + i * 32);
        const std::vector<float> sf = F1->GetScaleFactors();
        cv::Mat img1 = F1->GetImage();
        std::vector<cv::DMatch> out(std::max<size_t>(k2.size(), 1));
        int n = 0;
        check(tb_search_by_projection(shim_ctx(), Tcw, &cam, img1.cols, img1.rows, k1.data(), d1.data(), taken1.data(), (int)k1.size(),
                                      k2.data(), mp.data(), md.data(), (int)k2.size(), sf.data(), (int)sf.size(), nRatio, TH_HIGH,
                                      HISTO_LENGTH, checkOrientation ? 1 : 0, reinterpret_cast<tb_match*>(out.data()), (int)out.size(), &n),
              "Matcher::searchByProjection");
        out.resize(n);
        return out;
    }

    std::vector<cv::DMatch> Matcher::searchByProjection(const std::shared_ptr<Map>& map, const std::shared_ptr<Frame>& F1, float radio)
    {
        float Tcw[16];
        tb_camera cam;
        std::vector<tb_keypoint> k1;
        std::vector<uint8_t> d1, taken1;
        frame_projection_inputs(F1, Tcw, cam, k1, d1, taken1);
        const std::vector<std::shared_ptr<MapPoint>> pts = map->GetAllMapPoints();
        std::vector<tb_mappoint> mp(std::max<size_t>(pts.size(), 1));
        std::vector<uint8_t> md(std::max<size_t>(pts.size(), 1) * 32);
        for (size_t i = 0; i < pts.size(); i++) mappoint_record(pts[i], mp[i], md.data() + i * 32);
        const std::vector<float> sf = F1->GetScaleFactors();
        cv::Mat img1 = F1->GetImage();
        std::vector<cv::DMatch> out(std::max<size_t>(pts.size(), 1));
        int n = 0;
        check(tb_search_by_projection_map(shim_ctx(), Tcw, &cam, img1.cols, img1.rows, k1.data(), d1.data(), taken1.data(), (int)k1.size(),
                                          mp.data(), md.data(), (int)pts.size(), sf.data(), (int)sf.size(), nRatio, radio, TH_HIGH,
                                          reinterpret_cast<tb_match*>(out.data()), (int)out.size(), &n),
              "Matcher::searchByProjection(map)");
        out.resize(n);
        return out;
    }

    int Matcher::DescriptorDistance(const cv::Mat& a, const cv::Mat& b) { return tb_descriptor_distance(a.ptr(0), b.ptr(0)); }

    void Matcher::ComputeThreeMaxima(std::vector<int>* histo, const int L, int& ind1, int& ind2, int& ind3)
    {
        std::vector<int> sizes(L);
        for (int i = 0; i < L; i++) sizes[i] = (int)histo[i].size();
        tb_three_maxima(sizes.data(), L, &ind1, &ind2, &ind3);
    }

    /* ------------------------------------------------------------------ LocalBA */
    int LocalBA::PoseOptimization(std::shared_ptr<Frame>& F)
    {
        const int N = (int)F->GetKeys().size();
        std::vector<tb_obs> obs;
        std::vector<int> index;
        std::vector<uint8_t> outlier;
        const std::vector<float> invSigma2 = F->GetInverseScaleSigmaSquares();
        for (int i = 0; i < N; i++)
        {
            auto pMP = F->GetMapPoint(i);
            if (!pMP) continue;
            const Eigen::Vector3f Xw = pMP->GetWorldPos();
            tb_obs o;
            o.u = F->GetKey(i)->px.x(); o.v = F->GetKey(i)->px.y();
            o.X = Xw.x(); o.Y = Xw.y(); o.Z = Xw.z();
            o.inv_sigma2 = invSigma2.at(F->GetKey(i)->kp.octave);
            obs.push_back(o);
            index.push_back(i);
            outlier.push_back(F->GetOutlier(i) ? 1 : 0);
        }
        /* vertex reset value of every round, LocalBA.cpp:426-428: R0 = GetRotation()^T, t0 = -R0 * GetTranslation() in float */
        const Eigen::Matrix3f R0 = F->GetRotation().transpose();
        const Eigen::Vector3f twc = F->GetTranslation();
        float Tin[16] = {0}, Tout[16];
        for (int i = 0; i < 3; i++)
        {
            for (int j = 0; j < 3; j++) Tin[i * 4 + j] = R0(i, j);
            Tin[i * 4 + 3] = -(R0(i, 0) * twc[0] + R0(i, 1) * twc[1] + R0(i, 2) * twc[2]);
        }
        Tin[15] = 1.f;
        const double K[4] = {718.856, 718.856, 607.1928, 185.2157}; /* hard-coded in the reference, LocalBA.cpp:356-359 */
        int inliers = 0;
        check(tb_pose_opt(shim_ctx(), K, Tin, obs.data(), (int)obs.size(), outlier.data(), Tout, &inliers, nullptr), "LocalBA::PoseOptimization");
        if ((int)obs.size() < 3) return 0; /* LocalBA.cpp:401: pose untouched */
        for (size_t e = 0; e < index.size(); e++) F->SetOutlier(index[e], outlier[e] != 0);
        Eigen::Matrix4f pose = Eigen::Matrix4f::Identity();
        for (int i = 0; i < 3; i++) for (int j = 0; j < 4; j++) pose(i, j) = Tout[i * 4 + j];
        F->SetPose(pose);
        return inliers;
    }

    /* LocalBA::AddMapPointsByStereo (reference LocalBA.cpp:46-68) on tb_add_map_points_by_stereo: the current frame's keys
     * tracked into the equalised stereo image, epipolar outliers rejected, depth = bf / |dx|. fx is unused there too; the
     * reference's drawing / imshow (:56-67) is dropped. */
    std::vector<float> LocalBA::AddMapPointsByStereo(const std::shared_ptr<Frame>& current_frame, const std::shared_ptr<Frame>& stereo_frame,
                                                     const float bf, const float fx)
    {
        (void)fx;
        cv::Mat img_s = stereo_frame->GetImage(), img_c = current_frame->GetImage();
        if (img_s.cols != img_c.cols || img_s.rows != img_c.rows) throw std::invalid_argument("LocalBA::AddMapPointsByStereo: image sizes differ");
        std::vector<tb_keypoint> k;
        frame_keys(current_frame, k);
        const size_t N = k.size();
        std::vector<float> xy(2 * std::max<size_t>(N, 1)), depth(std::max<size_t>(N, 1), -1.0f);
        for (size_t i = 0; i < N; i++) { xy[2 * i] = k[i].x; xy[2 * i + 1] = k[i].y; }
        tb_camera cam;
        std::memset(&cam, 0, sizeof cam);
        cam.width = stereo_frame->GetCameraModel()->Width(); cam.height = stereo_frame->GetCameraModel()->Height();
        int nd = 0;
        check(tb_add_map_points_by_stereo(shim_ctx(), img_s.data, img_c.data, img_s.cols, img_s.rows, (int)img_s.step, &cam, xy.data(), (int)N, bf,
                                          depth.data(), &nd), "LocalBA::AddMapPointsByStereo");
        depth.resize(N);
        return depth;
    }

    Eigen::Vector3f LocalBA::LinearTriangle(const Eigen::Vector2f& p0, const Eigen::Vector2f& p1, const Eigen::Matrix4f& Tcw0,
                                           const Eigen::Matrix4f& Tcw1)
    {
        /* LocalBA.cpp:24-43: DLT design matrix; right singular vector of the smallest singular value, found
         * here by cyclic Jacobi on A^T A (the reference uses Eigen's JacobiSVD). */
        double A[4][4], M[4][4], V[4][4];
        for (int c = 0; c < 4; c++)
        {
            A[0][c] = p0[0] * Tcw0(2, c) - Tcw0(0, c);
            A[1][c] = p0[1] * Tcw0(2, c) - Tcw0(1, c);
            A[2][c] = p1[0] * Tcw1(2, c) - Tcw1(0, c);
            A[3][c] = p1[1] * Tcw1(2, c) - Tcw1(1, c);
        }
        for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++)
        {
            M[i][j] = 0; V[i][j] = (i == j);
            for (int k = 0; k < 4; k++) M[i][j] += A[k][i] * A[k][j];
        }
        for (int sweep = 0; sweep < 60; sweep++)
            for (int p = 0; p < 3; p++) for (int q = p + 1; q < 4; q++)
            {
                if (std::fabs(M[p][q]) < 1e-300) continue;
                const double th = (M[q][q] - M[p][p]) / (2 * M[p][q]);
                const double t = (th >= 0 ? 1 : -1) / (std::fabs(th) + std::sqrt(th * th + 1));
                const double c = 1 / std::sqrt(t * t + 1), s = t * c;
                for (int k = 0; k < 4; k++) { const double a = M[k][p], b = M[k][q]; M[k][p] = c * a - s * b; M[k][q] = s * a + c * b; }
                for (int k = 0; k < 4; k++) { const double a = M[p][k], b = M[q][k]; M[p][k] = c * a - s * b; M[q][k] = s * a + c * b; }
                for (int k = 0; k < 4; k++) { const double a = V[k][p], b = V[k][q]; V[k][p] = c * a - s * b; V[k][q] = s * a + c * b; }
            }
        int m = 0;
        for (int i = 1; i < 4; i++) if (M[i][i] < M[m][m]) m = i;
        Eigen::Vector3f r;
        r[0] = (float)(V[0][m] / V[3][m]); r[1] = (float)(V[1][m] / V[3][m]); r[2] = (float)(V[2][m] / V[3][m]);
        return r;
    }
}

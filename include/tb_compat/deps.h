/* Picks real OpenCV / Eigen when they are installed, otherwise the in-repo stand-ins. */
#ifndef TB_COMPAT_DEPS_H
#define TB_COMPAT_DEPS_H
#if defined(__has_include)
#if __has_include(<opencv2/core.hpp>)
#include <opencv2/core.hpp>
#define TB_HAVE_OPENCV 1
#endif
#if __has_include(<Eigen/Core>)
#include <Eigen/Core>
#define TB_HAVE_EIGEN 1
#endif
#endif
#ifndef TB_HAVE_OPENCV
#include "cv_lite.h"
#endif
#ifndef TB_HAVE_EIGEN
#include "eigen_lite.h"
#endif
/* DBoW2::FeatureVector (reference third_part/DBoW2/DBoW2/FeatureVector.h:21-51): node id -> indices of the frame's
 * features under that node. The matcher only reads the container; the vocabulary that fills it is not part of this path. */
#if defined(__has_include) && __has_include("third_part/DBoW2/DBoW2/FeatureVector.h")
#include "third_part/DBoW2/DBoW2/FeatureVector.h"
#else
#include <map>
#include <vector>
namespace DBoW2 {
typedef unsigned int NodeId;
class FeatureVector : public std::map<NodeId, std::vector<unsigned int>> {
public:
    void addFeature(NodeId id, unsigned int i_feature) { (*this)[id].push_back(i_feature); }
};
}
#endif
/* DBoW2::BowVector (third_part/DBoW2/DBoW2/BowVector.h:56-110) and the vocabulary Frame::SetBow takes. The reference's
 * ORBVocabulary is DBoW2::TemplatedVocabulary<FORB::TDescriptor, FORB> (include/types/Frame.h:22), whose tree is a protected
 * member; what this path needs of it is the tree itself, so the stand-in keeps it as the flat arrays of tb_vocabulary and
 * reads them from the same ORBvoc-style text file (TemplatedVocabulary::loadFromTextFile, TemplatedVocabulary.h:1338-1420).
 * A build against the real DBoW2 keeps this class next to it: one loadFromTextFile call per vocabulary. */
#include <cstdint>
#include <string>
namespace DBoW2 {
typedef unsigned int WordId;
typedef double WordValue;
class BowVector : public std::map<WordId, WordValue> {};
}
struct tb_vocab;
namespace TRACKING_BENCH {
class FlatVocabulary {
public:
    FlatVocabulary() = default;
    ~FlatVocabulary();                                       /* releases the device copy (shim library) */
    FlatVocabulary(const FlatVocabulary&) = delete;
    FlatVocabulary& operator=(const FlatVocabulary&) = delete;
    bool loadFromTextFile(const std::string& filename);     /* TemplatedVocabulary.h:1338-1420 */
    bool empty() const { return word_id.size() <= 1; }
    unsigned int size() const { return nwords; }            /* number of words */
    int getBranchingFactor() const { return k; }
    int getDepthLevels() const { return L; }
    int k = 0, L = 0, scoring = 0, weighting = 0;
    unsigned int nwords = 0;
    std::vector<int32_t> child_start, child_items, word_id;
    std::vector<uint8_t> desc;
    std::vector<double> weight;
    mutable tb_vocab* device = nullptr;                      /* uploaded on first use by Frame::SetBow */
};
}
#endif

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
#endif

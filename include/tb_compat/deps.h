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
#endif

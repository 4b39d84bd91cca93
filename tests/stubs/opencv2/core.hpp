// TEST STAND-IN, not OpenCV: the members of cv::Mat that include/gpu_depth.hpp and 3dvision_amd/host/eigen_adapter/gpu_impl_hip.cpp
// touch, for a -fsyntax-only parse (tests/test_adapter_syntax.py).  Pins nothing, is never linked or executed.
#pragma once
#include <cstdint>
#define CV_32FC1 5
namespace cv {
class Mat {
public:
    int rows = 0, cols = 0;
    Mat();
    Mat(int rows_, int cols_, int type);
    bool empty() const;
    bool isContinuous() const;
    Mat clone() const;
    template <class T> T* ptr(int row = 0);
    template <class T> const T* ptr(int row = 0) const;
};
}  // namespace cv

// elas_shim.cpp — C entry point around the reference's own LIBELAS matcher.
//
// TEST INFRASTRUCTURE ONLY (CPU timing comparator).  The LIBELAS sources are
// compiled from where they lie under /root/reference/lib/src/elas by oracle/Makefile ("make ref")
// into oracle/_ref/libelas_ref.so; nothing of them is copied into this repository.  Settings
// follow the reference's own call site: elasWrapper::compute_disparity
// (/root/reference/lib/src/elasWrapper.cpp:35-44,64,99) and StereoMatcherNew::initELAS
// (/root/reference/modules/DisparityModule/StereoMatcher.cpp:321).
#include <stdint.h>

#include "elas.h"
#ifdef _OPENMP
#include <omp.h>
#endif

extern "C" int elasref_process(const uint8_t* left, const uint8_t* right, float* disp_left, float* disp_right,
                               int width, int height, int disp_max, int ipol_gap_width) {
    Elas::parameters param(Elas::ROBOTICS);
    param.postprocess_only_left = true;
    param.disp_max = disp_max;
    param.ipol_gap_width = ipol_gap_width;
    Elas elas(param);
    const int32_t dims[3] = {width, height, width};
    elas.process(const_cast<uint8_t*>(left), const_cast<uint8_t*>(right), disp_left, disp_right, dims);
    return 0;
}

extern "C" int elasref_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads() < 3 ? omp_get_max_threads() : 3;  // elas_omp.cpp caps its regions at 2-3 threads
#else
    return 1;
#endif
}

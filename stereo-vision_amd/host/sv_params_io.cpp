#include "sv_params_io.h"

#include <sstream>
#include <vector>

namespace svh {

std::string sv_params_line(const PersistedParams& pp) {
    const Params& p = pp.p;
    std::ostringstream out;  // operator<< on doubles with the default precision, as the reference streams them
    out << "params (" << p.minDisparity << " " << pp.useBestDisp << " " << p.numberOfDisparities << " " << p.SADWindowSize << " "
        << p.disp12MaxDiff << " " << p.preFilterCap << " " << p.uniquenessRatio << " " << p.speckleWindowSize << " "
        << p.speckleRange << " " << p.sigmaColorBLF << " " << p.sigmaSpaceBLF << " " << p.wls_lambda << " " << p.wls_sigma << " "
        << (int)p.stereo_matching << " " << (int)p.BLFfiltering << " " << (int)p.WLSfiltering << ")";
    return out.str();
}

bool sv_params_parse(const std::string& line, PersistedParams& pp) {
    const size_t k = line.find("params");
    if (k == std::string::npos) return false;
    const size_t a = line.find('(', k), b = line.find(')', k);
    if (a == std::string::npos || b == std::string::npos || b < a) return false;
    std::istringstream in(line.substr(a + 1, b - a - 1));
    std::vector<double> v;
    for (double x; in >> x;) v.push_back(x);
    if (v.size() < 16) return false;
    const int alg = (int)v[13], blf = (int)v[14], wls = (int)v[15];
    if (alg < SGBM_OPENCV || alg > SGBM_HIP || blf < BLF_DISABLED || blf > BLF_CUDA || wls < WLS_DISABLED || wls > WLS_LRCHECK)
        return false;
    PersistedParams q = pp;
    q.p.minDisparity = (int)v[0];
    q.useBestDisp = v[1] != 0;
    q.p.numberOfDisparities = (int)v[2];
    q.p.SADWindowSize = (int)v[3];
    q.p.disp12MaxDiff = (int)v[4];
    q.p.preFilterCap = (int)v[5];
    q.p.uniquenessRatio = (int)v[6];
    q.p.speckleWindowSize = (int)v[7];
    q.p.speckleRange = (int)v[8];
    q.p.sigmaColorBLF = v[9];
    q.p.sigmaSpaceBLF = v[10];
    q.p.wls_lambda = v[11];
    q.p.wls_sigma = v[12];
    q.p.stereo_matching = (SM_MATCHING_ALG)alg;
    q.p.BLFfiltering = (SM_BLF_FILTER)blf;
    q.p.WLSfiltering = (SM_WLS_FILTER)wls;
    pp = q;
    return true;
}

}  // namespace svh

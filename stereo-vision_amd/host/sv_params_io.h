// sv_params_io.h — the "params (...)" line of SFM.ini (disk format):
// DispModule::loadConfigurationFile reads it (DispModule.cpp:896-918), updateConfigurationFile
// writes it (DispModule.cpp:1054-1071).  Sixteen values in this order:
//   minDisparity useBestDisp numberOfDisparities SADWindowSize disp12MaxDiff preFilterCap
//   uniquenessRatio speckleWindowSize speckleRange sigmaColorBLF sigmaSpaceBLF wls_lambda wls_sigma
//   stereo_matching BLFfiltering WLSfiltering
// stereo_matching is persisted as the enum's integer, which is why SGBM_HIP is appended (= 3) and
// the values 0/1/2 written by older builds keep their meaning.  No YARP needed: plain text.
#pragma once
#include <string>

#include "StereoMatcherHIP.h"

namespace svh {

struct PersistedParams {
    Params p;
    bool useBestDisp = true;  // DispModule member stored between minDisparity and numberOfDisparities
};

// "params (0 1 96 7 0 63 15 50 16 10 10 4500 1 3 0 0)" as updateConfigurationFile streams it
std::string sv_params_line(const PersistedParams& pp);
// Parses such a line (leading blanks and anything before "params" on the line are ignored, as in a
// YARP .ini group).  Returns false - and leaves pp untouched, which is where the reference falls back
// to initializeStereoParams() - if the key or the parenthesised list is missing, has fewer than 16
// values, or names an enum value that does not exist.
bool sv_params_parse(const std::string& line, PersistedParams& pp);

}  // namespace svh

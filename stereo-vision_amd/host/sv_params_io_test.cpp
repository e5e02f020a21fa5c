// CPU-only check of sv_params_io (run by tests/test_host_params.py): round trip, the reference's
// own default line, the appended enum value, rejects.
#include <cstdio>

#include "sv_params_io.h"

using namespace svh;

#define CHECK(c) do { if (!(c)) { std::printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #c); return 1; } } while (0)

int main() {
    PersistedParams a;
    a.p.numberOfDisparities = 128; a.p.SADWindowSize = 5; a.p.sigmaColorBLF = 12.5; a.p.wls_lambda = 8000.; a.p.wls_sigma = 1.5;
    a.p.stereo_matching = SGBM_HIP; a.p.BLFfiltering = BLF_ORIGINAL; a.p.WLSfiltering = WLS_ENABLED; a.useBestDisp = false;
    const std::string line = sv_params_line(a);
    CHECK(line == "params (0 0 128 5 0 63 15 50 16 12.5 10 8000 1.5 3 1 1)");
    PersistedParams b;
    CHECK(sv_params_parse("    " + line, b));
    CHECK(sv_params_line(b) == line);
    // what initializeStereoParams + updateConfigurationFile of the reference write (DispModule.cpp:369-394): SGBM_OPENCV = 0
    PersistedParams c;
    CHECK(sv_params_parse("params (0 1 96 7 0 63 15 50 16 10 10 4500 1 0 0 0)", c));
    CHECK(c.p.stereo_matching == SGBM_OPENCV && c.p.numberOfDisparities == 96 && c.useBestDisp && c.p.wls_lambda == 4500.);
    CHECK(sv_params_parse("params (0 1 96 7 0 63 15 50 16 10 10 4500 1 1 2 2)", c) && c.p.stereo_matching == SGBM_CUDA &&
          c.p.BLFfiltering == BLF_CUDA && c.p.WLSfiltering == WLS_LRCHECK);
    const std::string before = sv_params_line(c);
    CHECK(!sv_params_parse("params (0 1 96 7 0 63 15 50 16 10 10 4500 1 4 0 0)", c));  // no such algorithm
    CHECK(!sv_params_parse("params (0 1 96 7)", c));                                   // too short
    CHECK(!sv_params_parse("HN (1 0 0 0)", c));                                        // another key
    CHECK(sv_params_line(c) == before);                                                // rejects leave the target untouched
    std::printf("sv_params_io ok\n");
    return 0;
}

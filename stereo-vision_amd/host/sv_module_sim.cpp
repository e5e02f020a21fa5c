// sv_module_sim — stand-alone harness that reproduces the call order of
// DispModule::updateModule() (/root/reference/modules/DisparityModule/DispModule.cpp:661-814) on
// synthetic RGB pairs, without YARP: setImages -> matcher->compute() -> filterBLF("base") ->
// filterWLS("blf") -> getDisparity("wls") -> getDisparity16("wls") -> depthFromDisparity.
// Usage: sv_module_sim [width height numDisp frames paths wls(0|1) blf(0|1|2)]
#include <chrono>
#include <cstdio>
#include <cstdlib>

#include "StereoMatcherHIP.h"

using namespace svh;

static uint32_t lcg(uint32_t& s) { s = s * 1664525u + 1013904223u; return s >> 8; }

// textured left image, right = left shifted by a per-band disparity
static void make_pair(int W, int H, int D, uint32_t seed, Image& L, Image& R) {
    L = Image(H, W, 3, 1);
    R = Image(H, W, 3, 1);
    std::vector<uint8_t> tex((size_t)H * (W + D));
    uint32_t s = seed;
    for (int y = 0; y < H; y++) {
        int v = 128;
        for (int x = 0; x < W + D; x++) {
            v += (int)(lcg(s) % 41) - 20;
            v = v < 16 ? 16 : v > 240 ? 240 : v;
            tex[(size_t)y * (W + D) + x] = (uint8_t)v;
        }
    }
    for (int y = 0; y < H; y++) {
        int d = 4 + ((y * 3 / H) * (D / 4));
        for (int x = 0; x < W; x++)
            for (int c = 0; c < 3; c++) {
                L.ptr<uint8_t>(y)[3 * x + c] = tex[(size_t)y * (W + D) + x];
                R.ptr<uint8_t>(y)[3 * x + c] = tex[(size_t)y * (W + D) + x + d];
            }
    }
}

int main(int argc, char** argv) {
    int W = argc > 1 ? atoi(argv[1]) : 640, H = argc > 2 ? atoi(argv[2]) : 480, D = argc > 3 ? atoi(argv[3]) : 128;
    int frames = argc > 4 ? atoi(argv[4]) : 20, paths = argc > 5 ? atoi(argv[5]) : 8;
    const bool wls = argc > 6 && atoi(argv[6]) != 0;  // --wls (DispModule.cpp:259-265)
    const int blf = argc > 7 ? atoi(argv[7]) : 0;     // 1 = --blf BLF_ORIGINAL, 2 = --cuda_blf BLF_CUDA (DispModule.cpp:253-258)
    if (W <= 0 || H <= 0 || D <= 0 || frames < 2 || (paths != 5 && paths != 8)) {
        fprintf(stderr, "usage: sv_module_sim [width height numDisp frames(>=2) paths(5|8) wls(0|1) blf(0|1|2)]\n");
        return 1;
    }
    StereoCamera cam;
    StereoMatcherNew matcher(&cam);
    matcher.paths = paths;
    matcher.setAlgorithm("sgbm_hip");  // DispModule.cpp:246-251 keyword
    // DispModule::initializeStereoParams defaults (DispModule.cpp:369-394), numDisp from the CLI
    matcher.setParameters(0, D, 7, 0, 63, 15, 50, 16, 10.0, 10.0, 4500., 1.0, blf == 1 ? BLF_ORIGINAL : (blf == 2 ? BLF_CUDA : BLF_DISABLED), wls ? WLS_ENABLED : WLS_DISABLED, SGBM_HIP);
    matcher.updateHIPParams();
    // identity mapper + a plausible Q so the depth stage runs
    Image mapper(H, W, 2, 4);
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) { mapper.ptr<float>(y)[2 * x] = (float)x; mapper.ptr<float>(y)[2 * x + 1] = (float)y; }
    cam.setMapperL(mapper);
    const double f = 0.7 * W, b = 0.068;
    const double Q[16] = {1, 0, 0, -(W / 2 - 0.5), 0, 1, 0, -(H / 2 - 0.5), 0, 0, 0, f, 0, 0, 1 / b, 0};
    cam.setQ(Q);

    double total_ms = 0;
    long valid = 0, npx = 0, ndepth = 0;
    double depth_sum = 0;
    for (int i = 0; i < frames; i++) {
        Image L, R;
        make_pair(W, H, D, 1234u + i, L, R);
        cam.setImages(L, R);
        auto t0 = std::chrono::steady_clock::now();
        matcher.compute();  // DispModule.cpp:715
        if (!matcher.ok()) { fprintf(stderr, "compute failed\n"); return 2; }
        matcher.filterBLF("base");  // :722
        matcher.filterWLS("blf");   // :726
        Image vis = matcher.getDisparity("wls");
        Image d16 = matcher.getDisparity16("wls");
        Image depth = matcher.depthFromDisparity(d16);  // :768-800
        auto t1 = std::chrono::steady_clock::now();
        if (i > 0) total_ms += std::chrono::duration<double, std::milli>(t1 - t0).count();
        for (int y = 0; y < H; y++)
            for (int x = D; x < W; x++) {
                npx++;
                if (d16.ptr<int16_t>(y)[x] >= 0) {
                    valid++;
                    const float z = depth.ptr<float>(y)[x];
                    if (z > 0.0f && z < 1e6f) { depth_sum += z; ndepth++; }  // disparity 0 reprojects to infinity
                }
            }
        (void)vis;
    }
    printf("sv_module_sim: %dx%d D=%d paths=%d frames=%d  host-buffer path %.3f ms/frame (%.1f fps, PCIe-inclusive, "
           "synchronous)  valid %.1f%%  mean depth %.3f m\n",
           W, H, D, paths, frames, total_ms / (frames > 1 ? frames - 1 : 1), 1000.0 * (frames - 1) / (total_ms > 0 ? total_ms : 1),
           100.0 * valid / (npx ? npx : 1), ndepth ? depth_sum / ndepth : 0.0);
    return 0;
}

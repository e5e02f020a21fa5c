// sv_module_sim — stand-alone harness that reproduces the call order of
// DispModule::updateModule() (/root/reference/modules/DisparityModule/DispModule.cpp:661-814) on
// RGB pairs, without YARP: setImages -> matcher->compute() -> filterBLF("base") ->
// filterWLS("blf") -> getDisparity("wls") -> getDisparity16("wls") -> depthFromDisparity.
//
// Usage: sv_module_sim [width height numDisp frames paths wls(0|1) blf(0|1|2)] [options]
//   --mode a|b    a (default): INTEGRATION.md variant A, the matcher reads the pair StereoCamera rectified
//                 (getLRectified / getRRectified; the test double's CPU rectification is the identity);
//                 b: variant B, raw pair up, maps / remap / mapper on the GPU (needs --cam)
//   --in FILE     one RGB pair as raw bytes (left [H][W][3], right [H][W][3]) used for every frame
//                 (default: a synthetic pair per frame)
//   --cam FILE    text: for left and right n_dist K(9) dist(n_dist) Rrect(9) Prect(12), then Q(16)
//   --out FILE    raw dump of the last frame: d16 base, vis base, d16 blf, vis blf, d16 wls, vis wls, depth (float)
//   --breakdown   per-call host wall time (mean over the timed frames)
//   --sad N --cap N   SADWindowSize / preFilterCap as the GUI or RPC would set them (defaults 7 / 63, DispModule.cpp:369-394)
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>

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

static bool read_doubles(std::ifstream& f, double* v, int n) {
    for (int i = 0; i < n; i++)
        if (!(f >> v[i])) return false;
    return true;
}

int main(int argc, char** argv) {
    int pos[7] = {640, 480, 128, 20, 8, 0, 0};
    int npos = 0;
    std::string mode = "a", in_file, cam_file, out_file;
    bool breakdown = false;
    int sad = 7, cap = 63;
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        if (a == "--breakdown") breakdown = true;
        else if (a == "--mode" && i + 1 < argc) mode = argv[++i];
        else if (a == "--in" && i + 1 < argc) in_file = argv[++i];
        else if (a == "--cam" && i + 1 < argc) cam_file = argv[++i];
        else if (a == "--out" && i + 1 < argc) out_file = argv[++i];
        else if (a == "--sad" && i + 1 < argc) sad = atoi(argv[++i]);
        else if (a == "--cap" && i + 1 < argc) cap = atoi(argv[++i]);
        else if (npos < 7 && (isdigit((unsigned char)a[0]))) pos[npos++] = atoi(a.c_str());
        else { fprintf(stderr, "unknown argument %s\n", a.c_str()); return 1; }
    }
    const int W = pos[0], H = pos[1], D = pos[2], frames = pos[3], paths = pos[4];
    const bool wls = pos[5] != 0;  // --wls (DispModule.cpp:259-265)
    const int blf = pos[6];        // 1 = --blf BLF_ORIGINAL, 2 = --cuda_blf BLF_CUDA (DispModule.cpp:253-258)
    if (W <= 0 || H <= 0 || D <= 0 || frames < 1 || (paths != 5 && paths != 8) || (mode != "a" && mode != "b") || (mode == "b" && cam_file.empty())) {
        fprintf(stderr, "usage: sv_module_sim [width height numDisp frames paths(5|8) wls(0|1) blf(0|1|2)] [--mode a|b] [--in F] [--cam F] [--out F] [--breakdown]\n");
        return 1;
    }
    StereoCamera cam;
    StereoMatcherNew matcher(&cam);
    matcher.paths = paths;
    matcher.setAlgorithm("sgbm_hip");  // DispModule.cpp:246-251 keyword
    // DispModule::initializeStereoParams defaults (DispModule.cpp:369-394), numDisp from the CLI
    matcher.setParameters(0, D, sad, 0, cap, 15, 50, 16, 10.0, 10.0, 4500., 1.0, blf == 1 ? BLF_ORIGINAL : (blf == 2 ? BLF_CUDA : BLF_DISABLED), wls ? WLS_ENABLED : WLS_DISABLED, SGBM_HIP);
    matcher.updateHIPParams();

    // camera: from the file, or an identity rectification with a plausible Q so that the depth stage runs
    double KL[9], KR[9], dL[8] = {0}, dR[8] = {0}, RL[9], RR[9], PL[12], PR[12], Q[16];
    int ndL = 4, ndR = 4;
    if (!cam_file.empty()) {
        std::ifstream f(cam_file);
        auto read_cam = [&](double* K, double* d, int& nd, double* R, double* P) {
            double ndv = 0;
            if (!read_doubles(f, &ndv, 1) || ndv < 0 || ndv > 8) return false;
            nd = (int)ndv;
            return read_doubles(f, K, 9) && read_doubles(f, d, nd) && read_doubles(f, R, 9) && read_doubles(f, P, 12);
        };
        if (!read_cam(KL, dL, ndL, RL, PL) || !read_cam(KR, dR, ndR, RR, PR) || !read_doubles(f, Q, 16)) {
            fprintf(stderr, "bad camera file\n");
            return 1;
        }
    } else {
        const double fx = 0.7 * W, b = 0.068, cx = W / 2 - 0.5, cy = H / 2 - 0.5;
        const double K[9] = {fx, 0, cx, 0, fx, cy, 0, 0, 1}, I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        const double P[12] = {fx, 0, cx, 0, 0, fx, cy, 0, 0, 0, 1, 0};
        const double Qd[16] = {1, 0, 0, -cx, 0, 1, 0, -cy, 0, 0, 0, fx, 0, 0, 1 / b, 0};
        memcpy(KL, K, sizeof(K)); memcpy(KR, K, sizeof(K)); memcpy(RL, I3, sizeof(I3)); memcpy(RR, I3, sizeof(I3));
        memcpy(PL, P, sizeof(P)); memcpy(PR, P, sizeof(P)); memcpy(Q, Qd, sizeof(Qd));
    }
    cam.setIntrinsics(Image::mat64(3, 3, KL), Image::mat64(3, 3, KR), Image::mat64(ndL, 1, dL), Image::mat64(ndR, 1, dR));
    cam.setRectifyOnDevice(mode == "b");

    Image Lf, Rf;
    if (!in_file.empty()) {
        Lf = Image(H, W, 3, 1);
        Rf = Image(H, W, 3, 1);
        std::ifstream f(in_file, std::ios::binary);
        f.read((char*)Lf.data.data(), (std::streamsize)Lf.data.size());
        f.read((char*)Rf.data.data(), (std::streamsize)Rf.data.size());
        if (!f) { fprintf(stderr, "short input file\n"); return 1; }
    }

    double total_ms = 0, part_ms[6] = {0};
    long valid = 0, npx = 0, ndepth = 0;
    double depth_sum = 0;
    Image vis, d16, depth;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
        return std::chrono::duration<double, std::milli>(b - a).count();
    };
    for (int i = 0; i < frames; i++) {
        Image L, R;
        if (in_file.empty()) make_pair(W, H, D, 1234u + i, L, R);
        else { L = Lf; R = Rf; }
        // updateViaKinematics (DispModule.cpp:684) moves the cameras on every frame: a new cv::stereoRectify result
        cam.setRectification(Image::mat64(3, 3, RL), Image::mat64(3, 3, RR), Image::mat64(3, 4, PL), Image::mat64(3, 4, PR), Image::mat64(4, 4, Q));
        const auto t0 = now();
        cam.setImages(L, R);                        // DispModule.cpp:697
        const auto t1 = now();
        matcher.compute();                          // :715
        if (!matcher.ok()) { fprintf(stderr, "compute failed\n"); return 2; }
        const auto t2 = now();
        matcher.filterBLF("base");                  // :722
        const auto t3 = now();
        matcher.filterWLS("blf");                   // :726
        const auto t4 = now();
        vis = matcher.getDisparity("wls");          // :740
        d16 = matcher.getDisparity16("wls");        // :770
        const auto t5 = now();
        depth = matcher.depthFromDisparity(d16);    // :791
        const auto t6 = now();
        if (i > 0 || frames == 1) {
            total_ms += ms(t0, t6);
            part_ms[0] += ms(t0, t1); part_ms[1] += ms(t1, t2); part_ms[2] += ms(t2, t3);
            part_ms[3] += ms(t3, t4); part_ms[4] += ms(t4, t5); part_ms[5] += ms(t5, t6);
        }
        for (int y = 0; y < H; y++)
            for (int x = D; x < W; x++) {
                npx++;
                if (d16.ptr<int16_t>(y)[x] >= 0) {
                    valid++;
                    const float z = depth.ptr<float>(y)[x];
                    if (z > 0.0f && z < 1e6f) { depth_sum += z; ndepth++; }  // disparity 0 reprojects to infinity
                }
            }
    }
    const int timed = frames > 1 ? frames - 1 : 1;
    printf("sv_module_sim: mode %s %dx%d D=%d paths=%d wls=%d blf=%d frames=%d  host-buffer path %.3f ms/frame (%.1f fps, PCIe-inclusive, "
           "synchronous)  valid %.1f%%  mean depth %.3f m\n",
           mode.c_str(), W, H, D, paths, (int)wls, blf, frames, total_ms / timed, 1000.0 * timed / (total_ms > 0 ? total_ms : 1),
           100.0 * valid / (npx ? npx : 1), ndepth ? depth_sum / ndepth : 0.0);
    if (breakdown)
        printf("  per call (ms): setImages %.3f  compute %.3f  filterBLF %.3f  filterWLS %.3f  getDisparity x2 %.3f  depthFromDisparity %.3f\n",
               part_ms[0] / timed, part_ms[1] / timed, part_ms[2] / timed, part_ms[3] / timed, part_ms[4] / timed, part_ms[5] / timed);
    if (!out_file.empty()) {
        std::ofstream f(out_file, std::ios::binary);
        const Image parts[] = {matcher.getDisparity16("base"), matcher.getDisparity("base"), matcher.getDisparity16("blf"), matcher.getDisparity("blf"),
                               matcher.getDisparity16("wls"), matcher.getDisparity("wls"), depth};
        for (const Image& p : parts) f.write((const char*)p.data.data(), (std::streamsize)p.data.size());
        if (!f) { fprintf(stderr, "cannot write %s\n", out_file.c_str()); return 3; }
    }
    return 0;
}

// ORACLE — TEST INFRASTRUCTURE ONLY.  Sanitizer driver: exercises every oracle entry point on small random inputs
// in a binary built with -fsanitize=address,undefined (GPU sanitizers are not available on the pool; the CPU
// restatement is where memory/UB bugs in the checker itself would hide).  Run by tests/test_oracle_sanitizers.py.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
extern "C" {
void orc_depth_preprocess(const uint16_t*, const uint8_t*, int, int, float, float*);
int orc_unproject(const float*, const uint8_t*, int, int, float, float, float, float, float, float*, float*);
int orc_voxel_downsample(const float*, const float*, int, float, float*, float*, int*, int);
void orc_estimate_normals(const float*, int, int, float*, int*);
void orc_compute_fpfh(const float*, const float*, int, float, float*, int*, int*);
void orc_feature_match(const float*, int, const float*, int, int*);
void orc_ransac(const float*, int, const float*, int, const float*, const float*, const int*, float, int, float, float*, float*, float*, int*, int*, int*, int*);
int orc_icp(const float*, int, const float*, const float*, int, const float*, float, int, int, float*, float*, float*, float*);
void orc_bilateral_filter(const float*, float*, int, int, float, float);
int orc_filter_duplicates(const float*, int, float, float*);
void orc_demo_scene(int, int, float, uint16_t*, uint8_t*);
void orc_demo_mask(int, int, uint8_t*);
int orc_demo_model(float*, float*, int);
}
static float frand() { return (float)rand() / (float)RAND_MAX; }
int main() {
    srand(7);
    const int w = 64, h = 48;
    std::vector<uint16_t> raw(w * h); std::vector<uint8_t> mask(w * h), bgr(w * h * 3);
    orc_demo_scene(w, h, 1000.f, raw.data(), bgr.data()); orc_demo_mask(w, h, mask.data());
    for (auto& m : mask) m = (uint8_t)(rand() % 256);
    std::vector<float> depth(w * h), filt(w * h), xyz(w * h * 3), rgb(w * h * 3);
    orc_depth_preprocess(raw.data(), mask.data(), w, h, 1000.f, depth.data());
    orc_bilateral_filter(depth.data(), filt.data(), w, h, 1.5f, 0.05f);
    int n = orc_unproject(depth.data(), bgr.data(), w, h, 60.f, 60.f, 32.f, 24.f, 1.5f, xyz.data(), rgb.data());
    if (n <= 0) { n = 300; for (int i = 0; i < n * 3; ++i) xyz[i] = frand(); }
    std::vector<float> vx(n * 3), vc(n * 3); std::vector<int> first(n);
    int m = orc_voxel_downsample(xyz.data(), rgb.data(), n, 0.02f, vx.data(), vc.data(), first.data(), n);
    const int ns = 400, nt = 300;
    std::vector<float> src(ns * 3), tgt(nt * 3), nrm(nt * 3), fs(ns * 33), ft(nt * 33), desc(nt * 33);
    for (auto& v : src) v = frand(); for (auto& v : tgt) v = frand();
    for (auto& v : fs) v = frand(); for (auto& v : ft) v = frand();
    std::vector<int> knn(nt * 30), nb(nt * 100), cnt(nt), corr(ns), trace(200);
    orc_estimate_normals(tgt.data(), nt, 30, nrm.data(), knn.data());
    orc_compute_fpfh(tgt.data(), nrm.data(), nt, 0.2f, desc.data(), nb.data(), cnt.data());
    orc_feature_match(fs.data(), ns, ft.data(), nt, corr.data());
    float T[16], fit, rmse; int bi, ir;
    orc_ransac(src.data(), ns, tgt.data(), nt, fs.data(), ft.data(), nullptr, 0.05f, 200, 0.999f, T, &fit, &rmse, trace.data(), corr.data(), &bi, &ir);
    float I[16] = {1,0,0,0, 0,1,0,0, 0,0,1,0, 0,0,0,1}, To[16];
    std::vector<float> tr(20 * 20);
    int it = orc_icp(src.data(), ns, tgt.data(), nrm.data(), nt, I, 0.2f, 20, 1, To, &fit, &rmse, tr.data());
    it += orc_icp(src.data(), ns, tgt.data(), nullptr, nt, I, 0.2f, 20, 1, To, &fit, &rmse, tr.data());
    std::vector<float> poses(16 * 10), kept(16 * 10);
    for (int p = 0; p < 10; ++p) { for (int i = 0; i < 16; ++i) poses[16 * p + i] = I[i]; poses[16 * p + 12] = 0.05f * (p % 4); }
    int k = orc_filter_duplicates(poses.data(), 10, 0.1f, kept.data());
    std::vector<float> mx(3 * 2000), mn(3 * 2000);
    int nm = orc_demo_model(mx.data(), mn.data(), 2000);
    printf("asan_check ok: cloud %d voxels %d ransac best %d icp iters %d kept %d model %d\n", n, m, bi, it, k, nm);
    return 0;
}

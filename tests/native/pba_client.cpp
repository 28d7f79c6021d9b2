// A caller of the reference's C++ plugin interface (class ParallelBA, pba.h:52-133), written the way
// the reference's own driver uses it (Optimizer/pba/src/driver/driver.cpp:126-175), compiled against
// include/mi_pba.h and linked to libmi_ba.so.  Input/output: NVM files.
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "mi_model.h"
#include "mi_pba.h"

int main(int argc, char **argv)
{
    if (argc < 3) return 2;
    mi_model *mdl = mi_model_load_nvm(argv[1]);
    if (!mdl) { std::fprintf(stderr, "%s\n", mi_model_last_error()); return 1; }
    int64_t cnt[3];
    mi_model_counts(mdl, cnt);
    std::vector<double> f(cnt[0]), R(cnt[0] * 9), t(cnt[0] * 3), xyz(cnt[1] * 3), xy(cnt[2] * 2);
    std::vector<int> ptidx(cnt[2]), camidx(cnt[2]);
    mi_model_get(mdl, f.data(), R.data(), t.data(), nullptr, xyz.data(), nullptr, xy.data(), ptidx.data(), camidx.data(), nullptr);
    std::vector<CameraT> cams(cnt[0]);
    std::vector<Point3D> pts(cnt[1]);
    std::vector<Point2D> meas(cnt[2]);
    for (int64_t j = 0; j < cnt[0]; ++j) { cams[j].SetFocalLength(f[j]); cams[j].SetMatrixRotation(&R[j * 9]); cams[j].SetTranslation(&t[j * 3]); }
    if (argc > 3 && std::atoi(argv[3]) > 0) for (int j = 0; j < std::atoi(argv[3]); ++j) cams[j].SetConstantCamera();
    for (int64_t i = 0; i < cnt[1]; ++i) pts[i].SetPoint(&xyz[i * 3]);
    for (int64_t k = 0; k < cnt[2]; ++k) meas[k].SetPoint2D(xy[2 * k], xy[2 * k + 1]);

    ParallelBA pba(ParallelBA::PBA_CUDA_DEVICE_DEFAULT);        // on the stack, like driver.cpp:130
    char a0[] = "-lmi", a1[] = "150", a2[] = "-lme", a3[] = "0";   // run to convergence, not to the default 0.25 px^2
    char *args[] = { a0, a1, a2, a3 };
    pba.ParseParam(4, args);
    pba.SetCameraData(cams.size(), &cams[0]);
    pba.SetPointData(pts.size(), &pts[0]);
    pba.SetProjection(meas.size(), &meas[0], &ptidx[0], &camidx[0]);
    const int ok = pba.RunBundleAdjustment();
    std::printf("iterations %d mse %.9g code %d\n", ok, pba.GetMeanSquaredError(), (int)pba.GetInternalConfig()->GetBundleReturnCode());

    ParallelBA *dyn = NewParallelBA(ParallelBA::PBA_CUDA_DEVICE_DEFAULT);   // the dlopen-style factory
    std::printf("empty run %d version %d\n", dyn->RunBundleAdjustment(), ParallelBA_GetVersion());
    delete dyn;

    // write the refined model with the reference's own cameras/points arrays
    FILE *fo = std::fopen(argv[2], "w");
    for (auto &c : cams) std::fprintf(fo, "%.9g %.9g %.9g %.9g %.9g\n", c.f, c.t[0], c.t[1], c.t[2], c.m[0][0]);
    for (auto &p : pts) std::fprintf(fo, "%.9g %.9g %.9g\n", p.xyz[0], p.xyz[1], p.xyz[2]);
    std::fclose(fo);
    mi_model_free(mdl);
    return 0;
}

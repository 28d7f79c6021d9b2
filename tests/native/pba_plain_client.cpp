// One ParallelBA caller, two libraries.  This program uses nothing but the plugin interface of
// Optimizer/pba/src/pba/pba.h:52-133 (through include/mi_pba.h) and plain text files.  The SAME
// object file is linked
//   * against the reference's own PBA (oracle/_ref/libpba_ref.so, CPU build) by tests/golden/make_golden.py
//     to produce golden vectors, and
//   * against libmi_ba.so by tests/test_boundary_gpu.py,
// which checks both the numbers and that the header is binary compatible (vtable order, POD layouts).
//
//   pba_plain_client <problem.txt> <result.txt> <device>
// problem.txt:  m n k / fixed_intrinsics distortion nconst mode use_mask lmi lme / cameras / points / obs / [mask]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "mi_pba.h"

int main(int argc, char **argv)
{
    if (argc < 4) return 2;
    FILE *fi = std::fopen(argv[1], "r");
    if (!fi) return 1;
    int m, n, k, fixed_intr, distortion, nconst, mode, use_mask, lmi;
    double lme;
    if (std::fscanf(fi, "%d %d %d %d %d %d %d %d %d %lf", &m, &n, &k, &fixed_intr, &distortion, &nconst, &mode, &use_mask, &lmi, &lme) != 10) return 1;
    std::vector<CameraT> cams(m);
    std::vector<Point3D> pts(n);
    std::vector<Point2D> meas(k);
    std::vector<int> ptidx(k), camidx(k), mask(m);
    for (int j = 0; j < m; ++j) {
        double f, R[9], t[3], rad;
        if (std::fscanf(fi, "%lf", &f) != 1) return 1;
        for (double &v : R) if (std::fscanf(fi, "%lf", &v) != 1) return 1;
        for (double &v : t) if (std::fscanf(fi, "%lf", &v) != 1) return 1;
        if (std::fscanf(fi, "%lf", &rad) != 1) return 1;
        cams[j].SetFocalLength(f); cams[j].SetMatrixRotation(R); cams[j].SetTranslation(t);
        if (distortion > 0) cams[j].SetProjectionDistortion(rad);
        else if (distortion < 0) cams[j].SetMeasumentDistortion(rad);
        if (j < nconst) cams[j].SetConstantCamera();
    }
    for (int i = 0; i < n; ++i) { double X[3]; if (std::fscanf(fi, "%lf %lf %lf", X, X + 1, X + 2) != 3) return 1; pts[i].SetPoint(X); }
    for (int o = 0; o < k; ++o) {
        double x, y;
        if (std::fscanf(fi, "%d %d %lf %lf", &camidx[o], &ptidx[o], &x, &y) != 4) return 1;
        meas[o].SetPoint2D(x, y);
    }
    if (use_mask) for (int j = 0; j < m; ++j) if (std::fscanf(fi, "%d", &mask[j]) != 1) return 1;
    std::fclose(fi);

    ParallelBA *pba = NewParallelBA((ParallelBA::DeviceT)std::atoi(argv[3]));
    if (!pba) return 3;
    char a0[] = "-lmi", a1[16], a2[] = "-lme", a3[32], a4[] = "-v", a5[] = "0";
    std::snprintf(a1, sizeof a1, "%d", lmi);
    std::snprintf(a3, sizeof a3, "%g", lme);
    char *args[] = { a0, a1, a2, a3, a4, a5 };
    pba->ParseParam(6, args);
    if (const char *extra = std::getenv("PBA_CLIENT_ARGS")) {      // further options of the library under test, blank separated
        static char buf[512];
        std::snprintf(buf, sizeof buf, "%s", extra);
        std::vector<char *> more;
        for (char *tok = std::strtok(buf, " "); tok; tok = std::strtok(nullptr, " ")) more.push_back(tok);
        if (!more.empty()) pba->ParseParam((int)more.size(), &more[0]);
    }
    pba->SetFixedIntrinsics(fixed_intr != 0);
    pba->EnableRadialDistortion((ParallelBA::DistortionT)distortion);
    pba->SetCameraData(cams.size(), &cams[0]);
    pba->SetPointData(pts.size(), &pts[0]);
    pba->SetProjection(meas.size(), &meas[0], &ptidx[0], &camidx[0]);
    if (use_mask) pba->SetFocalMask(&mask[0], 1.0f);
    if (mode) pba->SetNextBundleMode((ParallelBA::BundleModeT)mode);
    const int iters = pba->RunBundleAdjustment();
    const float mse = pba->GetMeanSquaredError();
    delete pba;

    FILE *fo = std::fopen(argv[2], "w");
    if (!fo) return 1;
    std::fprintf(fo, "%d %.9g\n", iters, mse);
    for (auto &c : cams) {
        std::fprintf(fo, "%.9g %.9g %.9g %.9g", c.f, c.t[0], c.t[1], c.t[2]);
        for (int r = 0; r < 3; ++r) for (int s = 0; s < 3; ++s) std::fprintf(fo, " %.9g", c.m[r][s]);
        std::fprintf(fo, " %.9g\n", c.radial);
    }
    for (auto &p : pts) std::fprintf(fo, "%.9g %.9g %.9g\n", p.xyz[0], p.xyz[1], p.xyz[2]);
    std::fclose(fo);
    return 0;
}

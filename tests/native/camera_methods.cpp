// exercises the rotation helpers of CameraT_ (include/mi_data_interface.h): reads 3x3 matrices, prints what the
// methods make of them; tests/test_boundary_cpu.py compares with scipy
#include <cstdio>
#include "mi_data_interface.h"

int main()
{
    double R[9];
    while (std::scanf("%lf %lf %lf %lf %lf %lf %lf %lf %lf", R, R + 1, R + 2, R + 3, R + 4, R + 5, R + 6, R + 7, R + 8) == 9) {
        CameraD c, d, e;
        c.SetMatrixRotation(R);
        double q[4], r[3], back[9], back2[9], inv9[9], T[3] = { 1.0, 2.0, 3.0 }, Tout[3], ei[3], To[3];
        c.GetQuaternionRotation(q);
        c.GetRodriguesRotation(r);
        d.SetQuaternionRotation(q); d.GetMatrixRotation(back);
        e.SetRodriguesRotation(r); e.GetMatrixRotation(back2);
        c.SetTranslation(T);
        c.GetInvertedR9T(inv9, Tout);
        CameraD f; f.SetInvertedR9T(inv9, Tout);
        c.GetInvertedRT(ei, To);
        CameraD g; g.SetInvertedRT(ei, To);
        CameraT single; single.SetCameraT(c);
        double err9 = 0, errg = 0;
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { double a = f.m[i][j] - c.m[i][j], b = g.m[i][j] - c.m[i][j]; err9 += a * a; errg += b * b; }
        std::printf("%.17g %.17g %.17g %.17g  %.17g %.17g %.17g ", q[0], q[1], q[2], q[3], r[0], r[1], r[2]);
        for (double v : back) std::printf("%.17g ", v);
        for (double v : back2) std::printf("%.17g ", v);
        std::printf("%.3g %.3g %.9g %.9g\n", err9, errg, (double)c.GetRotationMatrixDeterminant(), (double)single.m[1][2]);
    }
    return 0;
}

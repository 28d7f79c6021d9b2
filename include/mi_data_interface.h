/*
 * mi_data_interface.h -- wire PODs of the optimizer boundary.
 *
 * Layout-compatible with the reference's Optimizer/DataInterface.h:
 *   CameraT_<FT>  { f, t[3], m[3][3], radial, int distortion_type, constant_camera }   (:46-56)
 *                 16 x FT = 64 B for float (asserted by PBA, SparseBundleCPU.cpp:2543)
 *   Point3D_<FT>  { xyz[3], reserved }  16 B for float                                  (:327-331)
 *   Point2D       { float x, y }                                                         (:369-371)
 * Camera convention (DataInterface.h:50-52): P = K [R T], T = -R C, +z forward,
 * x^ = f * (R X + T).xy / (R X + T).z.  Only the helpers the engine's shims need are provided.
 */
#ifndef MI_DATA_INTERFACE_H
#define MI_DATA_INTERFACE_H
#include <cmath>

template <class FT>
struct CameraT_ {
    typedef FT float_t;
    float_t f;
    float_t t[3];
    float_t m[3][3];
    float_t radial;
    int distortion_type;      /* -1 measurement distortion, 0 none, 1 projection distortion */
    float_t constant_camera;  /* 0 variable, 1 constant, 2 fixed intrinsics (DataInterface.h:72-74) */

    CameraT_() : radial(0), distortion_type(0), constant_camera(0) {}
    void SetConstantCamera() { constant_camera = 1.0f; }
    void SetVariableCamera() { constant_camera = 0.0f; }
    void SetFixedIntrinsic() { constant_camera = 2.0f; }
    template <class F> void SetFocalLength(F v) { f = (float_t)v; }
    float_t GetFocalLength() const { return f; }
    template <class F> void SetMatrixRotation(const F *r) { for (int i = 0; i < 9; ++i) m[i / 3][i % 3] = (float_t)r[i]; }
    template <class F> void GetMatrixRotation(F *r) const { for (int i = 0; i < 9; ++i) r[i] = (F)m[i / 3][i % 3]; }
    template <class F> void SetTranslation(const F T[3]) { for (int i = 0; i < 3; ++i) t[i] = (float_t)T[i]; }
    template <class F> void GetTranslation(F T[3]) const { for (int i = 0; i < 3; ++i) T[i] = (F)t[i]; }
    /* t = -R C (DataInterface.h:279-283) and back (:284-288) */
    template <class F> void SetCameraCenterAfterRotation(const F c[3])
    { for (int j = 0; j < 3; ++j) t[j] = -(float_t)(m[j][0] * c[0] + m[j][1] * c[1] + m[j][2] * c[2]); }
    template <class F> void GetCameraCenter(F c[3]) const
    { for (int j = 0; j < 3; ++j) c[j] = -(F)(m[0][j] * t[0] + m[1][j] * t[1] + m[2][j] * t[2]); }
    /* measurement distortion x (1 + r |x|^2) = projection; the reference's spelling is kept (DataInterface.h:81-86) */
    template <class F> void SetMeasumentDistortion(F r) { radial = (float_t)r; distortion_type = -1; }
    float_t GetMeasurementDistortion() const { return distortion_type == -1 ? radial : 0; }
    /* the normalised form stored in NVM files: r * f^2 */
    template <class F> void SetNormalizedMeasurementDistortion(F r) { SetMeasumentDistortion(r / (f * f)); }
    float_t GetNormalizedMeasurementDistortion() const { return GetMeasurementDistortion() * (f * f); }
    template <class F> void SetProjectionDistortion(F r) { radial = (float_t)r; distortion_type = 1; }
    float_t GetProjectionDistortion() const { return distortion_type == 1 ? radial : 0; }
    /* Rodrigues vector -> R (DataInterface.h:93-107) */
    template <class F> void SetRodriguesRotation(const F r[3])
    {
        const double a = std::sqrt((double)(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]));
        const double ct = a == 0.0 ? 0.5 : (1.0 - std::cos(a)) / a / a, st = a == 0.0 ? 1.0 : std::sin(a) / a;
        m[0][0] = float_t(1.0 - (r[1] * r[1] + r[2] * r[2]) * ct); m[0][1] = float_t(r[0] * r[1] * ct - r[2] * st);
        m[0][2] = float_t(r[2] * r[0] * ct + r[1] * st);           m[1][0] = float_t(r[0] * r[1] * ct + r[2] * st);
        m[1][1] = float_t(1.0 - (r[2] * r[2] + r[0] * r[0]) * ct); m[1][2] = float_t(r[1] * r[2] * ct - r[0] * st);
        m[2][0] = float_t(r[2] * r[0] * ct - r[1] * st);           m[2][1] = float_t(r[1] * r[2] * ct + r[0] * st);
        m[2][2] = float_t(1.0 - (r[0] * r[0] + r[1] * r[1]) * ct);
    }
    /* BAL / Bundler (-z camera) -> this convention: rows 2,3 of R and t2,t3 negated (DataInterface.h:290-296) */
    template <class F> void SetInvertedRT(const F e[3], const F T[3])
    {
        SetRodriguesRotation(e);
        for (int i = 1; i < 3; ++i) for (int j = 0; j < 3; ++j) m[i][j] = -m[i][j];
        SetTranslation(T); t[1] = -t[1]; t[2] = -t[2];
    }
    /* and back (DataInterface.h:298-305), the same with a 3x3 matrix instead of a Rodrigues vector (:306-322) */
    template <class F> void GetInvertedRT(F e[3], F T[3]) const
    {
        CameraT_ flipped;
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) flipped.m[i][j] = i ? -m[i][j] : m[i][j];
        flipped.GetRodriguesRotation(e);
        T[0] = (F)t[0]; T[1] = (F)-t[1]; T[2] = (F)-t[2];
    }
    template <class F> void SetInvertedR9T(const F e[9], const F T[3])
    {
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) m[i][j] = (float_t)(i ? -e[3 * i + j] : e[3 * i + j]);
        t[0] = (float_t)T[0]; t[1] = (float_t)-T[1]; t[2] = (float_t)-T[2];
    }
    template <class F> void GetInvertedR9T(F e[9], F T[3]) const
    {
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) e[3 * i + j] = (F)(i ? -m[i][j] : m[i][j]);
        T[0] = (F)t[0]; T[1] = (F)-t[1]; T[2] = (F)-t[2];
    }
    template <class CameraX> void SetCameraT(const CameraX &cam) /* copy across precisions (DataInterface.h:60-68) */
    {
        f = (float_t)cam.f;
        for (int i = 0; i < 3; ++i) { t[i] = (float_t)cam.t[i]; for (int j = 0; j < 3; ++j) m[i][j] = (float_t)cam.m[i][j]; }
        radial = (float_t)cam.radial; distortion_type = cam.distortion_type; constant_camera = (float_t)cam.constant_camera;
    }
    /* unit quaternion (w, x, y, z) <-> R (DataInterface.h:183-242); the input is normalised, (0,0,0,0) gives the identity */
    template <class F> void SetQuaternionRotation(const F q[4])
    {
        const double nq = std::sqrt((double)(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]));
        const double w = nq > 0 ? q[0] / nq : 1.0, x = nq > 0 ? q[1] / nq : 0.0, y = nq > 0 ? q[2] / nq : 0.0, z = nq > 0 ? q[3] / nq : 0.0;
        m[0][0] = float_t(w * w + x * x - y * y - z * z); m[0][1] = float_t(2 * (x * y - z * w));         m[0][2] = float_t(2 * (y * w + z * x));
        m[1][0] = float_t(2 * (x * y + w * z));         m[1][1] = float_t(w * w - x * x + y * y - z * z); m[1][2] = float_t(2 * (z * y - x * w));
        m[2][0] = float_t(2 * (x * z - y * w));         m[2][1] = float_t(2 * (y * z + w * x));         m[2][2] = float_t(w * w - x * x - y * y + z * z);
    }
    template <class F> void GetQuaternionRotation(F q[4]) const
    {
        const double tr = 1.0 + m[0][0] + m[1][1] + m[2][2];
        if (tr > 1e-9) {
            const double w = std::sqrt(tr) / 2.0;
            q[0] = (F)w; q[1] = (F)((m[2][1] - m[1][2]) / (4.0 * w)); q[2] = (F)((m[0][2] - m[2][0]) / (4.0 * w)); q[3] = (F)((m[1][0] - m[0][1]) / (4.0 * w));
            return;
        }
        /* rotation by ~pi: branch on the largest diagonal entry, signs as the reference leaves them */
        const int k = (m[0][0] > m[1][1] && m[0][0] > m[2][2]) ? 0 : (m[1][1] > m[2][2] ? 1 : 2), a = (k + 1) % 3, b = (k + 2) % 3;
        const double s = 2.0 * std::sqrt(1.0 + m[k][k] - m[a][a] - m[b][b]);
        q[1 + k] = (F)(0.25 * s);
        q[1 + a] = (F)((m[k][a] + m[a][k]) / s);
        q[1 + b] = (F)((m[k][b] + m[b][k]) / s);
        q[0] = (F)((k == 0 ? m[1][2] - m[2][1] : k == 1 ? m[0][2] - m[2][0] : m[0][1] - m[1][0]) / s);
    }
    /* R -> Rodrigues vector (DataInterface.h:108-181): zero for the identity, the pi cases by the largest diagonal entry */
    template <class F> void GetRodriguesRotation(F r[3]) const
    {
        const double pi = 3.14159265358979323846, c = (m[0][0] + m[1][1] + m[2][2] - 1.0) / 2.0, eps = 0.01;
        const bool symmetric = std::fabs(m[0][1] - m[1][0]) < eps && std::fabs(m[1][2] - m[2][1]) < eps && std::fabs(m[0][2] - m[2][0]) < eps;
        if (!symmetric) {
            const double a = std::acos(c), b = 0.5 * a / std::sin(a);
            r[0] = (F)(b * (m[2][1] - m[1][2])); r[1] = (F)(b * (m[0][2] - m[2][0])); r[2] = (F)(b * (m[1][0] - m[0][1]));
            return;
        }
        if (std::fabs(m[0][1] + m[1][0]) < 0.1 && std::fabs(m[1][2] + m[2][1]) < 0.1 && std::fabs(m[0][2] + m[2][0]) < 0.1 && c > 0.9) {
            r[0] = r[1] = r[2] = 0;
            return;
        }
        const double d[3] = { (m[0][0] + 1.0) / 2.0, (m[1][1] + 1.0) / 2.0, (m[2][2] + 1.0) / 2.0 };
        const double o[3][3] = { { 0, (m[0][1] + m[1][0]) / 4.0, (m[0][2] + m[2][0]) / 4.0 },
                                 { (m[0][1] + m[1][0]) / 4.0, 0, (m[1][2] + m[2][1]) / 4.0 },
                                 { (m[0][2] + m[2][0]) / 4.0, (m[1][2] + m[2][1]) / 4.0, 0 } };
        const int k = (d[0] > d[1] && d[0] > d[2]) ? 0 : (d[1] > d[2] ? 1 : 2), a = (k + 1) % 3, b = (k + 2) % 3;
        if (d[k] < eps) { r[k] = 0; r[a] = r[b] = (F)(std::sqrt(0.5) * pi); return; }
        const double s = std::sqrt(d[k]);
        r[k] = (F)(s * pi); r[a] = (F)(o[k][a] / s * pi); r[b] = (F)(o[k][b] / s * pi);
    }
    float GetRotationMatrixDeterminant() const
    {
        return (float)(m[0][0] * (m[1][1] * m[2][2] - m[1][2] * m[2][1]) - m[0][1] * (m[1][0] * m[2][2] - m[1][2] * m[2][0]) +
                       m[0][2] * (m[1][0] * m[2][1] - m[1][1] * m[2][0]));
    }
};

template <class FT>
struct Point3D_ {
    typedef FT float_t;
    float_t xyz[3];
    float_t reserved;
    template <class F> void SetPoint(F x, F y, F z) { xyz[0] = (float_t)x; xyz[1] = (float_t)y; xyz[2] = (float_t)z; reserved = 0; }
    template <class F> void SetPoint(const F *p) { SetPoint(p[0], p[1], p[2]); }
    template <class F> void GetPoint(F *p) const { p[0] = (F)xyz[0]; p[1] = (F)xyz[1]; p[2] = (F)xyz[2]; }
};

typedef CameraT_<float> CameraT;
typedef CameraT_<double> CameraD;
typedef Point3D_<float> Point3D;
typedef Point3D_<double> Point3B;

struct Point2D {
    float x, y;
    Point2D() {}
    template <class F> Point2D(F X, F Y) : x((float)X), y((float)Y) {}
    template <class F> void SetPoint2D(F X, F Y) { x = (float)X; y = (float)Y; }
    template <class F> void GetPoint2D(F &X, F &Y) const { X = (F)x; Y = (F)y; }
};

static_assert(sizeof(CameraT) == 64 && sizeof(Point3D) == 16 && sizeof(Point2D) == 8, "wire layout");
#endif

/*
 * tri_oracle.c -- CPU restatement of the pipeline's multi-view triangulation step.
 *
 * TEST INFRASTRUCTURE ONLY (same rules as ba_oracle.h): loaded by tests/, never by the product.
 *
 * What is restated (reference file:line, tree at /root/reference):
 *   - the per-track loop of SiftGPU/TriangulationMultiStep.cpp:1198-1304 (second level; the first level,
 *     :905-1010, is the same loop with min views 3 and a 1.0 degree ray angle): normalise the measurement by the
 *     focal length (stored in a float cv::Point2f, :1223-1228), require >= min_views views, require ONE pair of
 *     views whose rays meet at more than min_angle degrees (:1251-1265), triangulate, keep the point when the
 *     error is below 0.01 (:1274), else leave (0,0,0) (:1209-1211);
 *   - ComputeRayAngle, :498-530 (including the "+ centre, - centre" round trip of :515-521);
 *   - triangulate_n, :566-672 = Bundler's imagelib/triangulate.c:181-275 with the pipeline's error normalisation:
 *     linear solution of the 2n x 3 system (:580-600; LAPACK dgelsy there, the normal equations here), non-linear
 *     polish of the reprojection error (:626-630) by MINPACK's lmdif1 with tol 1e-5, per-view error, the depth
 *     test pp[2] < 1e-4 -> error 1e7 (:655-658) and error = sqrt(sum)/n (:661).
 *
 * THE POLISH IS A NO-OP IN THE REFERENCE AS BUILT HERE.  matrix.c:489-507 (lmdif_driver) passes `int *` for m, n, info
 * and lwa to lmdif1_, whose f2c types are `long int` (SiftGPU/ba_libs/lib/f2c/f2c.h:10): on LP64 lmdif1_ reads
 * garbage sizes, fails its own argument check (lmdif1.c "improper input parameters") and returns without touching x.
 * Measured with oracle/_ref (the reference's own sources, gcc 11, x86-64): triangulate_n returns the linear solution
 * bit for bit on every track tried.  So `polish = 0` is the reference's observable behaviour and the default of the
 * product; `polish = 1` is what the source intends -- MINPACK's lmdif restated below from
 * SiftGPU/ba_libs/lib/cminpack/{lmdif1,lmdif,lmpar,fdjac2}.c on the 3x3 normal equations instead of the QR factor
 * (same iterates up to rounding, including the early stop of the loose tolerance) and pinned against the real MINPACK
 * through scipy.optimize.leastsq (ftol = xtol = 1e-5, gtol = 0, factor 100: lmdif1's settings).
 *
 * Parity pin: tests/golden/ref_tri.npz holds what the REFERENCE's own triangulate_n (imagelib/triangulate.c compiled
 * into oracle/_ref/libref_sfm.so with the reference's matrix.c / cminpack and MKL's dgelsy) returns for seeded tracks,
 * and scipy's MINPACK answers from the same starting points.
 */
#include <math.h>
#include <stddef.h>
#include <string.h>

#define TRI_CLAMP(x, mn, mx) (((x) < (mn)) ? (mn) : (((x) > (mx)) ? (mx) : (x)))

/* TriangulationMultiStep.cpp:498-530; R row-major world->camera, c camera centres; returns radians */
double tri_ray_angle(const double p[2], const double q[2], const double *R1, const double *R2, const double *c1, const double *c2)
{
    double pv[3] = { p[0], p[1], 1.0 }, qv[3] = { q[0], q[1], 1.0 }, pw[3], qw[3];
    for (int r = 0; r < 3; ++r) { /* R^T v, then + centre - centre as the reference does */
        double a = R1[r] * pv[0] + R1[3 + r] * pv[1] + R1[6 + r] * pv[2];
        double b = R2[r] * qv[0] + R2[3 + r] * qv[1] + R2[6 + r] * qv[2];
        pw[r] = (a + c1[r]) - c1[r];
        qw[r] = (b + c2[r]) - c2[r];
    }
    double dot = pw[0] * qw[0] + pw[1] * qw[1] + pw[2] * qw[2];
    double mag = sqrt(pw[0] * pw[0] + pw[1] * pw[1] + pw[2] * pw[2]) * sqrt(qw[0] * qw[0] + qw[1] * qw[1] + qw[2] * qw[2]);
    return acos(TRI_CLAMP(dot / mag, -1.0 + 1.0e-8, 1.0 - 1.0e-8));
}

static int solve3(const double H[6], const double g[3], double x[3]) /* H = [h00 h01 h02 h11 h12 h22], symmetric */
{
    const double a = H[0], b = H[1], c = H[2], d = H[3], e = H[4], f = H[5];
    const double c00 = d * f - e * e, c01 = c * e - b * f, c02 = b * e - c * d;
    const double det = a * c00 + b * c01 + c * c02;
    if (!(fabs(det) > 0.0)) return 0;
    const double id = 1.0 / det;
    const double c11 = a * f - c * c, c12 = b * c - a * e, c22 = a * d - b * b;
    x[0] = (c00 * g[0] + c01 * g[1] + c02 * g[2]) * id;
    x[1] = (c01 * g[0] + c11 * g[1] + c12 * g[2]) * id;
    x[2] = (c02 * g[0] + c12 * g[1] + c22 * g[2]) * id;
    return 1;
}

/* residual vector of triangulate_n_residual (:548-563): f[2i] = p_i.x - proj_i(X).x, f[2i+1] = ... .y */
static void tri_resid(const double *R, const double *t, const double *p, const double X[3], double f[2])
{
    /* project(), :533-547: matrix_product331 then + t, then the perspective division */
    const double q0 = (R[0] * X[0] + R[1] * X[1] + R[2] * X[2]) + t[0];
    const double q1 = (R[3] * X[0] + R[4] * X[1] + R[5] * X[2]) + t[1];
    const double q2 = (R[6] * X[0] + R[7] * X[1] + R[8] * X[2]) + t[2];
    f[0] = p[0] - q0 / q2; f[1] = p[1] - q1 / q2;
}

static double tri_fnorm2(int nv, const double *pv, const double *Rs, const double *ts, const double X[3])
{
    double s = 0.0;
    for (int i = 0; i < nv; ++i) { double f[2]; tri_resid(Rs + 9 * i, ts + 3 * i, pv + 2 * i, X, f); s += f[0] * f[0] + f[1] * f[1]; }
    return s;
}

/* forward-difference Jacobian of fdjac2 (cminpack fdjac2.c: h = sqrt(epsmch) |x_j|, h = sqrt(epsmch) when x_j = 0)
 * folded straight into N = J^T J (upper, 6) and g = J^T f; returns |f|^2 */
static double tri_fd_normal(int nv, const double *pv, const double *Rs, const double *ts, const double X[3], double N[6], double g[3])
{
    const double eps = sqrt(2.220446049250313e-16);
    double h[3], s = 0.0;
    for (int j = 0; j < 3; ++j) { h[j] = eps * fabs(X[j]); if (h[j] == 0.0) h[j] = eps; }
    memset(N, 0, 6 * sizeof(double)); memset(g, 0, 3 * sizeof(double));
    for (int i = 0; i < nv; ++i) {
        double f0[2], fj[2], J[2][3];
        tri_resid(Rs + 9 * i, ts + 3 * i, pv + 2 * i, X, f0);
        for (int j = 0; j < 3; ++j) {
            double Xp[3] = { X[0], X[1], X[2] };
            Xp[j] = X[j] + h[j];
            tri_resid(Rs + 9 * i, ts + 3 * i, pv + 2 * i, Xp, fj);
            J[0][j] = (fj[0] - f0[0]) / h[j]; J[1][j] = (fj[1] - f0[1]) / h[j];
        }
        N[0] += J[0][0] * J[0][0] + J[1][0] * J[1][0]; N[1] += J[0][0] * J[0][1] + J[1][0] * J[1][1]; N[2] += J[0][0] * J[0][2] + J[1][0] * J[1][2];
        N[3] += J[0][1] * J[0][1] + J[1][1] * J[1][1]; N[4] += J[0][1] * J[0][2] + J[1][1] * J[1][2]; N[5] += J[0][2] * J[0][2] + J[1][2] * J[1][2];
        for (int j = 0; j < 3; ++j) g[j] += J[0][j] * f0[0] + J[1][j] * f0[1];
        s += f0[0] * f0[0] + f0[1] * f0[1];
    }
    return s;
}

static double quad3(const double H[6], const double y[3]) /* y^T H^-1 y, or -1 when H is singular */
{
    double z[3];
    if (!solve3(H, y, z)) return -1.0;
    return y[0] * z[0] + y[1] * z[1] + y[2] * z[2];
}

/* MINPACK lmpar (cminpack lmpar.c) on the normal equations: with R^T R = P^T N P every quantity it forms from the
 * QR factor is a solve with N or N + par D^2 (lmpar.c: Gauss-Newton step :120-157, parl :176-200, paru :202-220,
 * the Newton iteration :232-300).  x = the step of J x ~ f, i.e. (N + par D^2) x = g. */
static void tri_lmpar(const double N[6], const double g[3], const double diag[3], double delta, double *par, double x[3])
{
    const double dwarf = 2.2250738585072014e-308;
    const int nonsing = solve3(N, g, x);
    if (!nonsing) x[0] = x[1] = x[2] = 0.0;
    double w[3] = { diag[0] * x[0], diag[1] * x[1], diag[2] * x[2] };
    double dxnorm = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    double fp = dxnorm - delta;
    if (fp <= 0.1 * delta) { *par = 0.0; return; }
    double parl = 0.0;
    if (nonsing) {
        const double y[3] = { diag[0] * (w[0] / dxnorm), diag[1] * (w[1] / dxnorm), diag[2] * (w[2] / dxnorm) };
        const double q = quad3(N, y);
        if (q > 0.0) parl = fp / delta / q;
    }
    const double s[3] = { g[0] / diag[0], g[1] / diag[1], g[2] / diag[2] };
    const double gnorm = sqrt(s[0] * s[0] + s[1] * s[1] + s[2] * s[2]);
    double paru = gnorm / delta;
    if (paru == 0.0) paru = dwarf / fmin(delta, 0.1);
    *par = fmax(*par, parl); *par = fmin(*par, paru);
    if (*par == 0.0) *par = gnorm / dxnorm;
    for (int iter = 1;; ++iter) {
        if (*par == 0.0) *par = fmax(dwarf, 0.001 * paru);
        const double Hd[6] = { N[0] + *par * diag[0] * diag[0], N[1], N[2], N[3] + *par * diag[1] * diag[1], N[4], N[5] + *par * diag[2] * diag[2] };
        if (!solve3(Hd, g, x)) { x[0] = x[1] = x[2] = 0.0; }
        w[0] = diag[0] * x[0]; w[1] = diag[1] * x[1]; w[2] = diag[2] * x[2];
        dxnorm = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
        const double temp = fp;
        fp = dxnorm - delta;
        if (fabs(fp) <= 0.1 * delta || (parl == 0.0 && fp <= temp && temp < 0.0) || iter == 10) break;
        const double y[3] = { diag[0] * (w[0] / dxnorm), diag[1] * (w[1] / dxnorm), diag[2] * (w[2] / dxnorm) };
        const double parc = fp / delta / quad3(Hd, y);
        if (fp > 0.0) parl = fmax(parl, *par);
        if (fp < 0.0) paru = fmin(paru, *par);
        *par = fmax(parl, *par + parc);
    }
}

/* MINPACK lmdif1(tol) -> lmdif(ftol = xtol = tol, gtol = 0, maxfev = 200 (n+1), epsfcn = 0, mode 1, factor 100)
 * (cminpack lmdif1.c, lmdif.c) for n = 3 unknowns, on the normal equations; returns info */
static int tri_lmdif(int nv, const double *pv, const double *Rs, const double *ts, double X[3], double tol)
{
    const double epsmch = 2.220446049250313e-16, factor = 100.0;
    const int maxfev = 800;
    double fnorm = sqrt(tri_fnorm2(nv, pv, Rs, ts, X)), par = 0.0, delta = 0.0, xnorm = 0.0, diag[3] = { 1, 1, 1 };
    int nfev = 1, iter = 1, info = 0;
    for (;;) {
        double N[6], g[3];
        (void)tri_fd_normal(nv, pv, Rs, ts, X, N, g);
        nfev += 3;
        const double acn[3] = { sqrt(N[0]), sqrt(N[3]), sqrt(N[5]) }; /* column norms of J */
        if (iter == 1) {
            for (int j = 0; j < 3; ++j) { diag[j] = acn[j]; if (acn[j] == 0.0) diag[j] = 1.0; }
            xnorm = sqrt(diag[0] * X[0] * diag[0] * X[0] + diag[1] * X[1] * diag[1] * X[1] + diag[2] * X[2] * diag[2] * X[2]);
            delta = factor * xnorm;
            if (delta == 0.0) delta = factor;
        }
        double gnorm = 0.0;
        if (fnorm != 0.0)
            for (int j = 0; j < 3; ++j) if (acn[j] != 0.0) gnorm = fmax(gnorm, fabs((g[j] / fnorm) / acn[j]));
        if (gnorm <= 0.0) return 4;
        for (int j = 0; j < 3; ++j) diag[j] = fmax(diag[j], acn[j]);
        for (;;) { /* inner loop, lmdif.c L200 */
            double x[3], p[3], Xn[3];
            tri_lmpar(N, g, diag, delta, &par, x);
            for (int j = 0; j < 3; ++j) { p[j] = -x[j]; Xn[j] = X[j] + p[j]; }
            const double pnorm = sqrt(diag[0] * p[0] * diag[0] * p[0] + diag[1] * p[1] * diag[1] * p[1] + diag[2] * p[2] * diag[2] * p[2]);
            if (iter == 1) delta = fmin(delta, pnorm);
            const double fnorm1 = sqrt(tri_fnorm2(nv, pv, Rs, ts, Xn));
            ++nfev;
            double actred = -1.0;
            if (0.1 * fnorm1 < fnorm) actred = 1.0 - (fnorm1 / fnorm) * (fnorm1 / fnorm);
            const double Np[3] = { N[0] * p[0] + N[1] * p[1] + N[2] * p[2], N[1] * p[0] + N[3] * p[1] + N[4] * p[2], N[2] * p[0] + N[4] * p[1] + N[5] * p[2] };
            const double jp2 = fmax(p[0] * Np[0] + p[1] * Np[1] + p[2] * Np[2], 0.0); /* |J p|^2 */
            const double temp1 = sqrt(jp2) / fnorm, temp2 = sqrt(par) * pnorm / fnorm;
            const double prered = temp1 * temp1 + temp2 * temp2 / 0.5, dirder = -(temp1 * temp1 + temp2 * temp2);
            double ratio = 0.0;
            if (prered != 0.0) ratio = actred / prered;
            if (ratio <= 0.25) {
                double temp = 0.5;
                if (actred < 0.0) temp = 0.5 * dirder / (dirder + 0.5 * actred);
                if (0.1 * fnorm1 >= fnorm || temp < 0.1) temp = 0.1;
                delta = temp * fmin(delta, pnorm / 0.1);
                par /= temp;
            } else if (par == 0.0 || ratio >= 0.75) {
                delta = pnorm / 0.5;
                par = 0.5 * par;
            }
            if (ratio >= 1e-4) {
                for (int j = 0; j < 3; ++j) X[j] = Xn[j];
                xnorm = sqrt(diag[0] * X[0] * diag[0] * X[0] + diag[1] * X[1] * diag[1] * X[1] + diag[2] * X[2] * diag[2] * X[2]);
                fnorm = fnorm1;
                ++iter;
            }
            const int small = fabs(actred) <= tol && prered <= tol && 0.5 * ratio <= 1.0;
            if (small) info = 1;
            if (delta <= tol * xnorm) info = 2;
            if (small && info == 2) info = 3;
            if (info != 0) return info;
            if (nfev >= maxfev) info = 5;
            if (fabs(actred) <= epsmch && prered <= epsmch && 0.5 * ratio <= 1.0) info = 6;
            if (delta <= epsmch * xnorm) info = 7;
            if (gnorm <= epsmch) info = 8;
            if (info != 0) return info;
            if (ratio >= 1e-4) break;
        }
    }
}

/* triangulate_n with the pipeline's error convention.  pv 2*nv normalised measurements, Rs 9*nv, ts 3*nv.
 * err_each (nv) may be NULL.  Returns the error sqrt(sum)/nv (1e7-based when a view sees the point at depth < 1e-4). */
double tri_point(int nv, const double *pv, const double *Rs, const double *ts, double X[3], double *err_each, int polish)
{
    /* linear start: rows (R0 - x R2) X = t2 x - t0, (R1 - y R2) X = t2 y - t1 (:587-596), normal equations */
    double H[6] = { 0, 0, 0, 0, 0, 0 }, g[3] = { 0, 0, 0 };
    for (int i = 0; i < nv; ++i) {
        const double *R = Rs + 9 * i, *t = ts + 3 * i, x = pv[2 * i], y = pv[2 * i + 1];
        double a0[3], a1[3];
        for (int k = 0; k < 3; ++k) { a0[k] = R[k] - x * R[6 + k]; a1[k] = R[3 + k] - y * R[6 + k]; }
        const double b0 = t[2] * x - t[0], b1 = t[2] * y - t[1];
        H[0] += a0[0] * a0[0] + a1[0] * a1[0]; H[1] += a0[0] * a0[1] + a1[0] * a1[1]; H[2] += a0[0] * a0[2] + a1[0] * a1[2];
        H[3] += a0[1] * a0[1] + a1[1] * a1[1]; H[4] += a0[1] * a0[2] + a1[1] * a1[2]; H[5] += a0[2] * a0[2] + a1[2] * a1[2];
        for (int k = 0; k < 3; ++k) g[k] += a0[k] * b0 + a1[k] * b1;
    }
    if (!solve3(H, g, X)) { X[0] = X[1] = X[2] = 0.0; }
    /* polish: lmdif_driver(..., 1.0e-5), :626-630 -- see the header: a no-op in the reference as built on LP64 */
    if (polish) (void)tri_lmdif(nv, pv, Rs, ts, X, 1.0e-5);
    /* errors, :632-661 */
    double error = 0.0;
    for (int i = 0; i < nv; ++i) {
        const double *R = Rs + 9 * i, *t = ts + 3 * i;
        const double q0 = (R[0] * X[0] + R[1] * X[1] + R[2] * X[2]) + t[0];
        const double q1 = (R[3] * X[0] + R[4] * X[1] + R[5] * X[2]) + t[1];
        const double q2 = (R[6] * X[0] + R[7] * X[1] + R[8] * X[2]) + t[2];
        const double dx = q0 / q2 - pv[2 * i], dy = q1 / q2 - pv[2 * i + 1];
        if (err_each) err_each[i] = sqrt(dx * dx + dy * dy);
        error += dx * dx + dy * dy;
        if (q2 < 0.0001) error = 10000000.0;
    }
    return sqrt(error) / nv;
}

/* the track loop, TriangulationMultiStep.cpp:1198-1304.  Cameras: f (m), R (m*9), t (m*3), c (m*3 centres).
 * Tracks: CSR pt_ptr (n+1), obs_cam (camera index, < 0 = camera not in this model: skipped, :1221), xy (2 per obs,
 * pixels relative to the principal point).  Outputs: xyz n*3 ((0,0,0) when rejected), err n (-1 when not
 * triangulated), status n: 0 too few views, 1 no pair of rays above the angle, 2 error too large, 3 kept.
 * counts[4] = the reference's count1, count2, count3, count (:1303); returns sum of (err_i f_i)^2 over kept views. */
double tri_tracks(int m, const double *f, const double *R, const double *t, const double *c,
                  int n, const int *pt_ptr, const int *obs_cam, const double *xy,
                  int min_views, double min_angle_deg, double max_error, int float_normalise, int polish,
                  double *xyz, double *err, int *status, long long counts[4], long long *nproj_sum)
{
    double sum = 0.0;
    long long c1 = 0, c2 = 0, c3 = 0, np = 0;
    (void)m;
    for (int i = 0; i < n; ++i) {
        const int o0 = pt_ptr[i], o1 = pt_ptr[i + 1];
        xyz[3 * i] = xyz[3 * i + 1] = xyz[3 * i + 2] = 0.0; err[i] = -1.0; status[i] = 0;
        int nv = 0;
        for (int o = o0; o < o1; ++o) if (obs_cam[o] >= 0) ++nv;
        if (nv < min_views || nv < 2) continue;
        ++c1;
        /* gather (variable-length arrays in the reference, :1202-1206) */
        double pv[2 * 512], Rs[9 * 512], ts[3 * 512], fs[512], ee[512];
        int cams[512];
        if (nv > 512) { status[i] = 0; continue; } /* oracle limit; tests stay below */
        int k = 0;
        for (int o = o0; o < o1; ++o) {
            const int j = obs_cam[o];
            if (j < 0) continue;
            double px = xy[2 * o] / f[j], py = xy[2 * o + 1] / f[j];
            if (float_normalise) { px = (double)(float)px; py = (double)(float)py; } /* cv::Point2f, :1223-1228 */
            pv[2 * k] = px; pv[2 * k + 1] = py;
            memcpy(Rs + 9 * k, R + 9 * (size_t)j, 9 * sizeof(double)); memcpy(ts + 3 * k, t + 3 * (size_t)j, 3 * sizeof(double));
            fs[k] = f[j]; cams[k] = j; ++k;
        }
        int conditioned = 0;
        for (int a = 0; a < nv && !conditioned; ++a)
            for (int b = a + 1; b < nv; ++b) {
                const double val = tri_ray_angle(pv + 2 * a, pv + 2 * b, R + 9 * (size_t)cams[a], R + 9 * (size_t)cams[b],
                                                 c + 3 * (size_t)cams[a], c + 3 * (size_t)cams[b]);
                if (val * (180.0 / M_PI) > min_angle_deg) { conditioned = 1; break; }
            }
        if (!conditioned) { status[i] = 1; continue; }
        ++c2;
        double X[3];
        const double e = tri_point(nv, pv, Rs, ts, X, ee, polish);
        err[i] = e;
        if (!(e < max_error)) { status[i] = 2; continue; }
        ++c3;
        for (int v = 0; v < nv; ++v) sum += ee[v] * ee[v] * fs[v] * fs[v];
        np += nv;
        xyz[3 * i] = X[0]; xyz[3 * i + 1] = X[1]; xyz[3 * i + 2] = X[2]; status[i] = 3;
    }
    counts[0] = c1; counts[1] = c2; counts[2] = c3; counts[3] = c3;
    *nproj_sum = np;
    return sum;
}

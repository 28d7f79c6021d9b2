// model_io.cpp -- NVM / BAL text I/O and the optimizer_all process contract (include/mi_model.h).
// Host code; the numeric work is mi_ba.h's engine.
#include <algorithm>
#include <atomic>
#include <charconv>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <functional>
#include <iomanip>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include <sched.h>
#include <signal.h>
#include <sys/mman.h>
#include <sys/wait.h>
#include <unistd.h>

#include <hip/hip_runtime_api.h>

#include "focal_links.h"
#include "model.h"

#include "../../include/mi_model.h"

static thread_local std::string g_err;
void miba::set_model_error(const std::string &msg) { g_err = msg; }

namespace {

// host threads for the text readers / writers of large models; MI_MODEL_IO_THREADS=1 keeps them serial (tests compare the two)
unsigned io_threads()
{
    if (const char *ev = std::getenv("MI_MODEL_IO_THREADS")) { const int v = std::atoi(ev); if (v >= 1) return (unsigned)std::min(v, 64); }
    return std::min<unsigned>(8u, std::max(1u, std::thread::hardware_concurrency()));
}

// Whole-file tokenizer: the cluster files are hundreds of megabytes of numbers, and iostream's operator>> spends
// ~25 ns per character on them; reading the file once and parsing with std::from_chars is ~8x faster.
class Scanner {
public:
    bool open(const char *path)
    {
        FILE *f = std::fopen(path, "rb");
        if (!f) return false;
        std::fseek(f, 0, SEEK_END);
        const long sz = std::ftell(f);
        std::fseek(f, 0, SEEK_SET);
        buf_.resize(sz > 0 ? (size_t)sz : 0);
        const size_t got = buf_.empty() ? 0 : std::fread(&buf_[0], 1, buf_.size(), f);
        std::fclose(f);
        buf_.resize(got);
        p_ = buf_.data(); e_ = p_ + buf_.size();
        return true;
    }
    Scanner() = default;
    Scanner(const char *b, const char *e) : p_(b), e_(e) {}             // a view of part of another scanner's buffer
    const char *pos() { skip(); return p_; }
    const char *end() const { return e_; }
    bool at_end() { skip(); return p_ >= e_; }
    // only blanks up to the end of the line (or of the buffer)?  consumes them and the newline
    bool rest_of_line_blank()
    {
        while (p_ < e_ && (*p_ == ' ' || *p_ == '\t' || *p_ == '\r')) ++p_;
        if (p_ >= e_) return true;
        if (*p_ != '\n') return false;
        ++p_;
        return true;
    }
    char peek() { skip(); return p_ < e_ ? *p_ : '\0'; }
    bool token(const char *&b, const char *&en)
    {
        skip();
        if (p_ >= e_) return false;
        b = p_;
        while (p_ < e_ && !is_space(*p_)) ++p_;
        en = p_;
        return true;
    }
    bool get(std::string &s) { const char *b, *en; if (!token(b, en)) return fail(); s.assign(b, en); return true; }
    bool get(double &v)
    {
        const char *b, *en;
        if (!token(b, en)) return fail();
        if (*b == '+') ++b;
        auto r = std::from_chars(b, en, v);
        if (r.ec != std::errc() || r.ptr != en) { v = std::strtod(std::string(b, en).c_str(), nullptr); } // "nan(...)", hex, ...
        return true;
    }
    template <class I> bool get_int(I &v)
    {
        const char *b, *en;
        if (!token(b, en)) return fail();
        if (*b == '+') ++b;
        auto r = std::from_chars(b, en, v);
        if (r.ec != std::errc()) { double d = 0; std::from_chars(b, en, d); v = (I)d; if (b == en) return fail(); }
        return true;
    }
    void skip_line() { while (p_ < e_ && *p_ != '\n') ++p_; }
    bool ok() const { return ok_; }
private:
    static bool is_space(char c) { return c == ' ' || c == '\n' || c == '\t' || c == '\r' || c == '\f' || c == '\v'; }
    void skip() { while (p_ < e_ && is_space(*p_)) ++p_; }
    bool fail() { ok_ = false; return false; }
    std::string buf_;
    const char *p_ = nullptr, *e_ = nullptr;
    bool ok_ = true;
};

// %.12g of a double / decimal of an integer appended to a buffer (std::to_chars: no locale, no format parsing)
struct Writer {
    std::string buf;
    void num(double v)
    {
        char tmp[40];
        auto r = std::to_chars(tmp, tmp + sizeof tmp, v, std::chars_format::general, 12);
        buf.append(tmp, r.ptr);
    }
    template <class I> void integer(I v)
    {
        char tmp[24];
        auto r = std::to_chars(tmp, tmp + sizeof tmp, v);
        buf.append(tmp, r.ptr);
    }
    void ch(char c) { buf.push_back(c); }
    void str(const std::string &s) { buf.append(s); }
    bool flush_to(const char *path)
    {
        FILE *f = std::fopen(path, "wb");
        if (!f) return false;
        const bool good = std::fwrite(buf.data(), 1, buf.size(), f) == buf.size();
        return std::fclose(f) == 0 && good;
    }
};

void quat_to_R(const double q[4], double *R) // DataInterface.h:183-207
{
    const double qq = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    double qw = 1, qx = 0, qy = 0, qz = 0;
    if (qq > 0) { qw = q[0] / qq; qx = q[1] / qq; qy = q[2] / qq; qz = q[3] / qq; }
    R[0] = qw * qw + qx * qx - qz * qz - qy * qy; R[1] = 2 * qx * qy - 2 * qz * qw; R[2] = 2 * qy * qw + 2 * qz * qx;
    R[3] = 2 * qx * qy + 2 * qw * qz; R[4] = qy * qy + qw * qw - qz * qz - qx * qx; R[5] = 2 * qz * qy - 2 * qx * qw;
    R[6] = 2 * qx * qz - 2 * qy * qw; R[7] = 2 * qy * qz + 2 * qw * qx; R[8] = qz * qz + qw * qw - qy * qy - qx * qx;
}

void rodrigues_to_R(const double r[3], double *R) // DataInterface.h:93-107
{
    const double a = std::sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    const double ct = a == 0.0 ? 0.5 : (1.0 - std::cos(a)) / a / a, st = a == 0.0 ? 1.0 : std::sin(a) / a;
    R[0] = 1.0 - (r[1] * r[1] + r[2] * r[2]) * ct; R[1] = r[0] * r[1] * ct - r[2] * st; R[2] = r[2] * r[0] * ct + r[1] * st;
    R[3] = r[0] * r[1] * ct + r[2] * st; R[4] = 1.0 - (r[2] * r[2] + r[0] * r[0]) * ct; R[5] = r[1] * r[2] * ct - r[0] * st;
    R[6] = r[2] * r[0] * ct - r[1] * st; R[7] = r[1] * r[2] * ct + r[0] * st; R[8] = 1.0 - (r[0] * r[0] + r[1] * r[1]) * ct;
}

void R_to_rodrigues(const double *R, double r[3]) // generic branch of DataInterface.h:108-181
{
    double c = (R[0] + R[4] + R[8] - 1.0) / 2.0;
    c = c > 1 ? 1 : (c < -1 ? -1 : c);
    const double a = std::acos(c), s = std::sin(a);
    const double b = std::fabs(s) < 1e-12 ? 0.5 : 0.5 * a / s;
    r[0] = b * (R[7] - R[5]); r[1] = b * (R[2] - R[6]); r[2] = b * (R[3] - R[1]);
}

} // namespace

extern "C" {

const char *mi_model_last_error(void) { return g_err.c_str(); }
void mi_model_free(mi_model *m) { delete m; }

// The point records of a large NVM (a million lines, 450 MB on C3) parsed by a few host threads: the text is cut at line ends,
// every piece parsed on its own, the pieces put together in order.  That presumes what every writer of these files does -- one
// record per line --, so every record must end where its line ends, every piece must be used up and the records must add up to
// the announced count; anything else (and any small file) returns false and the caller's token-by-token loop reads the file
// the way the reference's stream reader would.
static bool load_points_in_parallel(Scanner &in, int ncam, int npt, mi_model *m)
{
    const char *b = in.pos(), *e = in.end();
    const unsigned nt = io_threads();
    if ((npt >= 0 && npt < (1 << 15)) || nt < 2 || e - b < (1 << 20)) return false;   // npt < 0: as many records as the text holds (ours_new.txt)
    std::vector<const char *> cut(nt + 1, e);
    cut[0] = b;
    for (unsigned k = 1; k < nt; ++k) {
        const char *p = b + (size_t)(e - b) * k / nt;
        while (p < e && *p != '\n') ++p;
        cut[k] = p < e ? p + 1 : e;
        if (cut[k] < cut[k - 1]) cut[k] = cut[k - 1];
    }
    struct Piece { std::vector<double> xyz, xy; std::vector<int> rgb, cam, npj; std::vector<long long> feat; bool ok = true; };
    std::vector<Piece> pc(nt);
    std::vector<std::thread> th;
    for (unsigned k = 0; k < nt; ++k)
        th.emplace_back([&, k]() {
            Piece &q = pc[k];
            Scanner s(cut[k], cut[k + 1]);
            while (!s.at_end()) {
                double X[3]; int c[3], npj = 0;
                s.get(X[0]); s.get(X[1]); s.get(X[2]); s.get_int(c[0]); s.get_int(c[1]); s.get_int(c[2]); s.get_int(npj);
                if (!s.ok() || npj < 0) { q.ok = false; return; }
                for (int j = 0; j < npj; ++j) {
                    int cidx = -1; long long fidx = 0; double x = 0, y = 0;
                    s.get_int(cidx); s.get_int(fidx); s.get(x); s.get(y);
                    if (!s.ok() || cidx < 0 || cidx >= ncam) { q.ok = false; return; }
                    q.cam.push_back(cidx); q.feat.push_back(fidx); q.xy.push_back(x); q.xy.push_back(y);
                }
                if (!s.rest_of_line_blank()) { q.ok = false; return; }
                q.xyz.insert(q.xyz.end(), X, X + 3); q.rgb.insert(q.rgb.end(), c, c + 3); q.npj.push_back(npj);
            }
        });
    for (auto &t : th) t.join();
    size_t total = 0, nproj = 0;
    for (const Piece &q : pc) { if (!q.ok) return false; total += q.npj.size(); nproj += q.cam.size(); }
    if (npt >= 0 ? total != (size_t)npt : total == 0) return false;
    m->xyz.resize(3 * total); m->rgb.resize(3 * total);
    m->cam_idx.resize(nproj); m->pt_idx.resize(nproj); m->feat_id.resize(nproj); m->xy.resize(2 * nproj);
    std::vector<size_t> p0(nt + 1, 0), j0(nt + 1, 0);
    for (unsigned k = 0; k < nt; ++k) { p0[k + 1] = p0[k] + pc[k].npj.size(); j0[k + 1] = j0[k] + pc[k].cam.size(); }
    th.clear();
    for (unsigned k = 0; k < nt; ++k)
        th.emplace_back([&, k]() {
            const Piece &q = pc[k];
            std::copy(q.xyz.begin(), q.xyz.end(), m->xyz.begin() + 3 * p0[k]);
            std::copy(q.rgb.begin(), q.rgb.end(), m->rgb.begin() + 3 * p0[k]);
            std::copy(q.cam.begin(), q.cam.end(), m->cam_idx.begin() + j0[k]);
            std::copy(q.feat.begin(), q.feat.end(), m->feat_id.begin() + j0[k]);
            std::copy(q.xy.begin(), q.xy.end(), m->xy.begin() + 2 * j0[k]);
            size_t j = j0[k];
            for (size_t i = 0; i < q.npj.size(); ++i)
                for (int a = 0; a < q.npj[i]; ++a) m->pt_idx[j++] = (int)(p0[k] + i);
        });
    for (auto &t : th) t.join();
    return true;
}

mi_model *mi_model_load_nvm(const char *path)
{
    Scanner in;
    if (!in.open(path)) { g_err = std::string("cannot open ") + path; return nullptr; }
    int nrot = 4; bool r9t = false;
    std::string token;
    if (in.peek() == 'N') { in.get(token); if (token.find("R9T") != std::string::npos) { nrot = 9; r9t = true; } }
    int ncam = 0, npt = 0;
    in.get_int(ncam);
    if (!in.ok() || ncam <= 1) { g_err = "NVM: fewer than 2 cameras"; return nullptr; } // util_VSFM.h:158
    mi_model *m = new mi_model;
    m->names.resize(ncam); m->f.resize(ncam); m->R.resize((size_t)ncam * 9); m->t.resize((size_t)ncam * 3); m->d.resize((size_t)ncam * 2);
    for (int i = 0; i < ncam; ++i) {
        double q[9], c[3];
        in.get(m->names[i]); in.get(m->f[i]);
        for (int j = 0; j < nrot; ++j) in.get(q[j]);
        in.get(c[0]); in.get(c[1]); in.get(c[2]); in.get(m->d[2 * i]); in.get(m->d[2 * i + 1]);
        double *R = &m->R[(size_t)i * 9], *t = &m->t[(size_t)i * 3];
        if (r9t) { std::memcpy(R, q, sizeof(double) * 9); std::memcpy(t, c, sizeof(double) * 3); }
        else { quat_to_R(q, R); for (int r = 0; r < 3; ++r) t[r] = -(R[3 * r] * c[0] + R[3 * r + 1] * c[1] + R[3 * r + 2] * c[2]); }
    }
    in.get_int(npt);
    if (!in.ok() || npt <= 0) { g_err = "NVM: no points"; delete m; return nullptr; }
    m->xyz.resize((size_t)npt * 3); m->rgb.resize((size_t)npt * 3);
    if (load_points_in_parallel(in, ncam, npt, m)) return m;
    m->cam_idx.clear(); m->pt_idx.clear(); m->feat_id.clear(); m->xy.clear();
    for (int i = 0; i < npt; ++i) {
        int npj = 0;
        in.get(m->xyz[3 * (size_t)i]); in.get(m->xyz[3 * (size_t)i + 1]); in.get(m->xyz[3 * (size_t)i + 2]);
        in.get_int(m->rgb[3 * (size_t)i]); in.get_int(m->rgb[3 * (size_t)i + 1]); in.get_int(m->rgb[3 * (size_t)i + 2]); in.get_int(npj);
        if (!in.ok()) { g_err = "NVM: truncated point record"; delete m; return nullptr; }
        for (int j = 0; j < npj; ++j) {
            int cidx = -1; long long fidx = 0; double x = 0, y = 0;
            in.get_int(cidx); in.get_int(fidx); in.get(x); in.get(y);
            if (!in.ok()) { g_err = "NVM: truncated file"; delete m; return nullptr; }
            if (cidx < 0 || cidx >= ncam) { g_err = "NVM: camera index out of range"; delete m; return nullptr; }
            m->cam_idx.push_back(cidx); m->pt_idx.push_back(i); m->feat_id.push_back(fidx);
            m->xy.push_back(x); m->xy.push_back(y);
        }
    }
    if (!in.ok()) { g_err = "NVM: truncated file"; delete m; return nullptr; }
    return m;
}

// The input half of the optimizer_all file contract: what the stage programs write before system("../bin/optimizer_all ...")
// (SiftGPU/TriangulationMultiStep.cpp:1024-1095 and :1308-1382 write ONLY these; MergeClusters.cpp:1009-1052 and
// Merge3D3D.cpp:811-860 write them beside an NVM):
//   RTglobalmapped_for_optimisation.txt   per camera: 9 numbers R (row-major), then 3 numbers camera CENTRE  ->  t = -R C
//   focal_for_optimisation.txt            "ncam", then per camera "camName focal 2"
//   Invmap.txt                            per camera "localIndex camName"     (names of the output NVM; optional)
//   ours_new.txt                          per point "X Y Z r g b n {camLocalIdx pointId x y}*"
//   Distortion.txt                        one radial coefficient per camera   (optional; every writer emits 0.000)
mi_model *mi_model_load_side_files(const char *dir_c)
{
    if (!dir_c) { g_err = "side files: null directory"; return nullptr; }
    std::string dir = dir_c;
    if (!dir.empty() && dir.back() != '/') dir += '/';
    Scanner rt, fo, pts;
    if (!rt.open((dir + "RTglobalmapped_for_optimisation.txt").c_str())) { g_err = "cannot open " + dir + "RTglobalmapped_for_optimisation.txt"; return nullptr; }
    if (!fo.open((dir + "focal_for_optimisation.txt").c_str())) { g_err = "cannot open " + dir + "focal_for_optimisation.txt"; return nullptr; }
    if (!pts.open((dir + "ours_new.txt").c_str())) { g_err = "cannot open " + dir + "ours_new.txt"; return nullptr; }
    int ncam = 0;
    fo.get_int(ncam);
    if (!fo.ok() || ncam <= 0) { g_err = "focal_for_optimisation.txt: bad camera count"; return nullptr; }
    mi_model *m = new mi_model;
    m->names.resize(ncam); m->f.resize(ncam); m->R.resize((size_t)ncam * 9); m->t.resize((size_t)ncam * 3); m->d.assign((size_t)ncam * 2, 0.0);
    for (int j = 0; j < ncam; ++j) {
        int flag = 0;
        double c[3];
        fo.get(m->names[j]); fo.get(m->f[j]); fo.get_int(flag);
        double *R = &m->R[(size_t)j * 9], *t = &m->t[(size_t)j * 3];
        for (int r = 0; r < 9; ++r) rt.get(R[r]);
        rt.get(c[0]); rt.get(c[1]); rt.get(c[2]);
        for (int r = 0; r < 3; ++r) t[r] = -(R[3 * r] * c[0] + R[3 * r + 1] * c[1] + R[3 * r + 2] * c[2]);   // getTfromRC, MergeClusters.cpp:133-137
    }
    if (!fo.ok()) { g_err = "focal_for_optimisation.txt: truncated"; delete m; return nullptr; }
    if (!rt.ok()) { g_err = "RTglobalmapped_for_optimisation.txt: fewer than 12 numbers per camera"; delete m; return nullptr; }
    Scanner inv, dist;
    if (inv.open((dir + "Invmap.txt").c_str())) {
        while (inv.peek() != '\0') {
            int idx = -1; std::string name;
            inv.get_int(idx); inv.get(name);
            if (!inv.ok()) break;
            if (idx < 0 || idx >= ncam) { g_err = "Invmap.txt: camera index out of range"; delete m; return nullptr; }
            m->names[idx] = name;
        }
    }
    if (dist.open((dir + "Distortion.txt").c_str()))
        for (int j = 0; j < ncam && dist.peek() != '\0'; ++j) dist.get(m->d[2 * (size_t)j]);
    if (load_points_in_parallel(pts, ncam, -1, m)) return m;      // (the same record as an NVM point line; large files only)
    m->xyz.clear(); m->rgb.clear(); m->cam_idx.clear(); m->pt_idx.clear(); m->feat_id.clear(); m->xy.clear();
    while (pts.peek() != '\0') {
        double X[3]; int rgb[3], nview = 0;
        pts.get(X[0]); pts.get(X[1]); pts.get(X[2]); pts.get_int(rgb[0]); pts.get_int(rgb[1]); pts.get_int(rgb[2]); pts.get_int(nview);
        if (!pts.ok() || nview < 0) { g_err = "ours_new.txt: truncated point record"; delete m; return nullptr; }
        const int i = (int)m->npt();
        m->xyz.insert(m->xyz.end(), X, X + 3); m->rgb.insert(m->rgb.end(), rgb, rgb + 3);
        for (int v = 0; v < nview; ++v) {
            int cidx = -1; long long pid = 0; double x = 0, y = 0;
            pts.get_int(cidx); pts.get_int(pid); pts.get(x); pts.get(y);
            if (!pts.ok()) { g_err = "ours_new.txt: truncated track"; delete m; return nullptr; }
            if (cidx < 0 || cidx >= ncam) { g_err = "ours_new.txt: camera index out of range"; delete m; return nullptr; }
            m->cam_idx.push_back(cidx); m->pt_idx.push_back(i); m->feat_id.push_back(pid);
            m->xy.push_back(x); m->xy.push_back(y);
        }
    }
    if (m->npt() == 0) { g_err = "ours_new.txt: no points"; delete m; return nullptr; }
    return m;
}

int mi_model_save_nvm(const mi_model *m, const char *path)
{
    if (!m) return MI_BA_ERROR;
    Writer out;
    out.buf.reserve(64 + m->ncam() * 200);
    out.str("NVM_V3_R9T\n"); out.integer(m->ncam()); out.ch('\n');
    for (size_t i = 0; i < m->ncam(); ++i) {
        out.str(m->names[i].empty() ? std::string("unknown") : m->names[i]); out.ch(' '); out.num(m->f[i]); out.ch(' ');
        for (int j = 0; j < 9; ++j) { out.num(m->R[i * 9 + j]); out.ch(' '); }
        out.num(m->t[i * 3]); out.ch(' '); out.num(m->t[i * 3 + 1]); out.ch(' '); out.num(m->t[i * 3 + 2]); out.ch(' ');
        out.num(m->d[2 * i]); out.ch(' '); out.num(m->d[2 * i + 1]); out.ch('\n');
    }
    out.integer(m->npt()); out.ch('\n');
    // the point lines (a million points, ten million measurements: 450 MB of text on C3) are formatted by a few host threads, each
    // into its own buffer, and written out in order -- formatting was 1.4 of the 1.6 s the file took
    const size_t np = m->npt(), nproj = m->nproj();
    const unsigned nt = np < (1u << 15) ? 1u : io_threads();
    std::vector<Writer> part(nt);
    auto format_points = [&](Writer &w, size_t p0, size_t p1) {
        size_t j = (size_t)(std::lower_bound(m->pt_idx.begin(), m->pt_idx.end(), (int)p0) - m->pt_idx.begin());
        w.buf.reserve((p1 - p0) * 64 + (size_t)((double)nproj * (double)(p1 - p0) / (double)std::max<size_t>(np, 1)) * 44);
        for (size_t i = p0; i < p1; ++i) {
            w.num(m->xyz[3 * i]); w.ch(' '); w.num(m->xyz[3 * i + 1]); w.ch(' '); w.num(m->xyz[3 * i + 2]); w.ch(' ');
            w.integer(m->rgb[3 * i]); w.ch(' '); w.integer(m->rgb[3 * i + 1]); w.ch(' '); w.integer(m->rgb[3 * i + 2]); w.ch(' ');
            size_t je = j;
            while (je < nproj && m->pt_idx[je] == (int)i) ++je;
            w.integer(je - j); w.ch(' ');
            for (; j < je; ++j) {
                w.integer(m->cam_idx[j]); w.ch(' '); w.integer(m->feat_id[j]); w.ch(' ');
                w.num(m->xy[2 * j]); w.ch(' '); w.num(m->xy[2 * j + 1]); w.ch(' ');
            }
            w.ch('\n');
        }
    };
    if (nt == 1) format_points(part[0], 0, np);
    else {
        std::vector<std::thread> th;
        for (unsigned k = 0; k < nt; ++k) th.emplace_back([&, k]() { format_points(part[k], np * k / nt, np * (k + 1) / nt); });
        for (auto &t : th) t.join();
    }
    FILE *f = std::fopen(path, "wb");
    bool good = f != nullptr;
    if (good) {
        good = std::fwrite(out.buf.data(), 1, out.buf.size(), f) == out.buf.size();
        for (unsigned k = 0; k < nt && good; ++k) good = std::fwrite(part[k].buf.data(), 1, part[k].buf.size(), f) == part[k].buf.size();
        good = std::fclose(f) == 0 && good;
    }
    if (!good) { g_err = std::string("cannot write ") + path; return MI_BA_ERROR; }
    return 0;
}

mi_model *mi_model_load_bal(const char *path)
{
    Scanner in;
    if (!in.open(path)) { g_err = std::string("cannot open ") + path; return nullptr; }
    long long ncam = 0, npt = 0, nproj = 0;
    in.get_int(ncam); in.get_int(npt); in.get_int(nproj);
    if (!in.ok() || ncam <= 0 || npt <= 0 || nproj <= 0) { g_err = "BAL: bad header"; return nullptr; }
    mi_model *m = new mi_model;
    struct Obs { int p, c; double x, y; };
    std::vector<Obs> obs; obs.reserve(nproj);
    bool sorted = true;
    for (long long i = 0; i < nproj; ++i) {
        int cidx = -1, pidx = -1; double x = 0, y = 0;
        in.get_int(cidx); in.get_int(pidx); in.get(x); in.get(y);
        if (!in.ok()) { g_err = "BAL: truncated observations"; delete m; return nullptr; }
        if (pidx < 0 || pidx >= npt || cidx < 0 || cidx >= ncam) continue;   // util_VSFM.h:475-477
        if (!obs.empty() && obs.back().p > pidx) sorted = false;
        obs.push_back({ pidx, cidx, x, -y });                                   // y negated, :480
    }
    if (!sorted) { // the engine wants observations grouped by point; a stable counting sort keeps file order inside a point
        std::vector<long long> start(npt + 1, 0);
        for (auto &o : obs) start[o.p + 1]++;
        for (long long i = 0; i < npt; ++i) start[i + 1] += start[i];
        std::vector<Obs> tmp(obs.size());
        for (auto &o : obs) tmp[start[o.p]++] = o;
        obs.swap(tmp);
    }
    for (auto &o : obs) { m->pt_idx.push_back(o.p); m->cam_idx.push_back(o.c); m->xy.push_back(o.x); m->xy.push_back(o.y); m->feat_id.push_back(0); }
    m->names.assign(ncam, "unknown"); m->f.resize(ncam); m->R.resize(ncam * 9); m->t.resize(ncam * 3); m->d.assign(ncam * 2, 0.0);
    for (long long i = 0; i < ncam; ++i) {
        double p[9] = { 0 };
        for (int j = 0; j < 9; ++j) in.get(p[j]);
        m->f[i] = p[6];
        double *R = &m->R[i * 9], *t = &m->t[i * 3];
        rodrigues_to_R(p, R);                                                 // SetInvertedRT, DataInterface.h:290-296
        for (int r = 3; r < 9; ++r) R[r] = -R[r];
        t[0] = p[3]; t[1] = -p[4]; t[2] = -p[5];
        m->d[2 * i] = p[7]; m->d[2 * i + 1] = p[8];
    }
    m->xyz.resize(npt * 3); m->rgb.assign(npt * 3, 0);
    for (long long i = 0; i < npt * 3; ++i) in.get(m->xyz[i]);
    if (!in.ok()) { g_err = "BAL: truncated file"; delete m; return nullptr; }
    return m;
}

int mi_model_save_bal(const mi_model *m, const char *path)
{
    if (!m) return MI_BA_ERROR;
    std::ofstream out(path);
    if (!out) { g_err = std::string("cannot write ") + path; return MI_BA_ERROR; }
    out << std::setprecision(12) << m->ncam() << ' ' << m->npt() << ' ' << m->nproj() << '\n';
    for (size_t i = 0; i < m->nproj(); ++i) out << m->cam_idx[i] << ' ' << m->pt_idx[i] << ' ' << m->xy[2 * i] << ' ' << -m->xy[2 * i + 1] << '\n';
    for (size_t i = 0; i < m->ncam(); ++i) {
        double R[9], r[3];
        std::memcpy(R, &m->R[i * 9], sizeof(R));
        for (int k = 3; k < 9; ++k) R[k] = -R[k];
        R_to_rodrigues(R, r);
        out << r[0] << ' ' << r[1] << ' ' << r[2] << ' ' << m->t[3 * i] << ' ' << -m->t[3 * i + 1] << ' ' << -m->t[3 * i + 2] << ' '
            << m->f[i] << ' ' << m->d[2 * i] << ' ' << m->d[2 * i + 1] << '\n';
    }
    for (size_t i = 0; i < m->npt(); ++i) out << m->xyz[3 * i] << ' ' << m->xyz[3 * i + 1] << ' ' << m->xyz[3 * i + 2] << '\n';
    return out ? 0 : MI_BA_ERROR;
}

mi_model *mi_model_load_bundler_out(const char *path)
{
    std::ifstream in(path);
    if (!in) { g_err = std::string("cannot open ") + path; return nullptr; }
    std::string token;
    while (in.peek() == '#') std::getline(in, token);
    long long ncam = 0, npt = 0;
    in >> ncam >> npt;
    if (!in || ncam <= 1 || npt <= 1) { g_err = "bundle.out: fewer than 2 cameras or points"; return nullptr; } // util_VSFM.h:339
    std::string list = path;
    const size_t ext = list.rfind(".out");
    std::ifstream listin;
    if (ext != std::string::npos) { list.replace(ext, std::string::npos, "-list.txt"); listin.open(list); }
    mi_model *m = new mi_model;
    m->names.assign(ncam, "unknown"); m->f.resize(ncam); m->R.resize(ncam * 9); m->t.resize(ncam * 3); m->d.resize(ncam * 2);
    for (long long i = 0; i < ncam; ++i) {
        double q[9], c[3];
        in >> m->f[i] >> m->d[2 * i] >> m->d[2 * i + 1];
        for (int j = 0; j < 9; ++j) in >> q[j];
        in >> c[0] >> c[1] >> c[2];
        double *R = &m->R[i * 9], *t = &m->t[i * 3];        // SetInvertedR9T, DataInterface.h:306-314
        for (int j = 0; j < 3; ++j) R[j] = q[j];
        for (int j = 3; j < 9; ++j) R[j] = -q[j];
        t[0] = c[0]; t[1] = -c[1]; t[2] = -c[2];
        std::string line;
        if (listin && (listin >> line)) {
            const size_t sl = line.find_last_of("/\\");
            m->names[i] = sl == std::string::npos ? line : line.substr(sl + 1);
            std::getline(listin, line);
        }
    }
    m->xyz.resize(npt * 3); m->rgb.resize(npt * 3);
    for (long long i = 0; i < npt; ++i) {
        int npj = 0;
        in >> m->xyz[3 * i] >> m->xyz[3 * i + 1] >> m->xyz[3 * i + 2] >> m->rgb[3 * i] >> m->rgb[3 * i + 1] >> m->rgb[3 * i + 2] >> npj;
        if (!in) { g_err = "bundle.out: truncated point record"; delete m; return nullptr; }
        for (int j = 0; j < npj; ++j) {
            int cidx; long long fidx; double x, y;
            in >> cidx >> fidx >> x >> y;
            if (cidx < 0 || cidx >= ncam) { g_err = "bundle.out: camera index out of range"; delete m; return nullptr; }
            m->cam_idx.push_back(cidx); m->pt_idx.push_back((int)i); m->feat_id.push_back(fidx);
            m->xy.push_back(x); m->xy.push_back(-y);        // y negated, util_VSFM.h:390
        }
    }
    if (!in) { g_err = "bundle.out: truncated file"; delete m; return nullptr; }
    return m;
}

int mi_model_save_bundler_out(const mi_model *m, const char *path)
{
    if (!m) return MI_BA_ERROR;
    std::string list = path;
    const size_t ext = list.rfind(".out");
    if (ext == std::string::npos) { g_err = "bundle.out: file name must end in .out"; return MI_BA_ERROR; } // util_VSFM.h:405
    list.replace(ext, std::string::npos, "-list.txt");
    std::ofstream out(path);
    if (!out) { g_err = std::string("cannot write ") + path; return MI_BA_ERROR; }
    out << "# Bundle file v0.3\n" << std::setprecision(12) << m->ncam() << ' ' << m->npt() << '\n';
    for (size_t i = 0; i < m->ncam(); ++i) {
        out << m->f[i] << ' ' << m->d[2 * i] << ' ' << m->d[2 * i + 1] << '\n';
        for (int j = 0; j < 9; ++j) out << (j < 3 ? m->R[i * 9 + j] : -m->R[i * 9 + j]) << ((j % 3) == 2 ? '\n' : ' ');
        out << m->t[3 * i] << ' ' << -m->t[3 * i + 1] << ' ' << -m->t[3 * i + 2] << '\n';
    }
    size_t j = 0;
    for (size_t i = 0; i < m->npt(); ++i) {
        size_t je = j;
        while (je < m->nproj() && m->pt_idx[je] == (int)i) ++je;
        out << m->xyz[3 * i] << ' ' << m->xyz[3 * i + 1] << ' ' << m->xyz[3 * i + 2] << '\n' << m->rgb[3 * i] << ' ' << m->rgb[3 * i + 1] << ' '
            << m->rgb[3 * i + 2] << '\n' << (je - j) << ' ';
        // one view per line and an empty line after the point, as SaveBundlerOut lays them out (util_VSFM.h:437-440); the key is
        // passed through (the reference writes 0).  NOT reproduced: the reference counts a point's views with
        // `j + npj < point_data.size()` (:433), so once j passes the number of POINTS it writes "0" views for every later point
        for (; j < je; ++j) out << m->cam_idx[j] << ' ' << m->feat_id[j] << ' ' << m->xy[2 * j] << ' ' << -m->xy[2 * j + 1] << '\n';
        out << '\n';
    }
    std::ofstream listout(list);
    for (auto &nm : m->names) listout << nm << '\n';
    return out ? 0 : MI_BA_ERROR;
}

int mi_model_remove_invisible_points(mi_model *m)
{
    if (!m) return MI_BA_ERROR;
    const size_t np = m->nproj(), npt = m->npt(), nc = m->ncam();
    auto depth = [&](size_t k) {
        const double *R = &m->R[(size_t)m->cam_idx[k] * 9], *X = &m->xyz[(size_t)m->pt_idx[k] * 3];
        return R[6] * X[0] + R[7] * X[1] + R[8] * X[2] + m->t[(size_t)m->cam_idx[k] * 3 + 2];
    };
    std::vector<double> zz(np);
    for (size_t k = 0; k < np; ++k) zz[k] = depth(k);
    std::vector<double> tmp(zz);
    std::nth_element(tmp.begin(), tmp.begin() + np / 2, tmp.end());
    const double thr = tmp[np / 2] * 0.001;                                       // util_VSFM.h:749
    std::vector<char> pmask(npt, 1), cmask(nc, 1);
    int removed = 0;
    for (size_t k = 0; k < np; ++k) {
        const int pid = m->pt_idx[k];
        if (!pmask[pid]) continue;
        if (!(zz[k] > thr)) { pmask[pid] = 0; ++removed; }
    }
    if (!removed) return 0;
    const int min_observation = 20;                                               // :771
    std::vector<int> cv(nc), pv(npt);
    int again;
    do {
        std::fill(cv.begin(), cv.end(), 0);
        for (size_t k = 0; k < np; ++k) if (pmask[m->pt_idx[k]]) cv[m->cam_idx[k]]++;
        std::fill(pv.begin(), pv.end(), 0);
        for (size_t k = 0; k < np; ++k) if (pmask[m->pt_idx[k]] && cv[m->cam_idx[k]] >= min_observation) pv[m->pt_idx[k]]++;
        again = 0;
        for (size_t i = 0; i < npt; ++i) if (pmask[i] && pv[i] < 2) { pmask[i] = 0; ++again; ++removed; }
    } while (again > 0);
    for (size_t j = 0; j < nc; ++j) cmask[j] = cv[j] >= min_observation;
    std::vector<int> cidx(nc, -1), pidx(npt, -1);
    mi_model o;
    for (size_t j = 0; j < nc; ++j) {
        if (!cmask[j]) continue;
        cidx[j] = (int)o.f.size();
        o.names.push_back(m->names[j]); o.f.push_back(m->f[j]);
        o.R.insert(o.R.end(), m->R.begin() + j * 9, m->R.begin() + j * 9 + 9);
        o.t.insert(o.t.end(), m->t.begin() + j * 3, m->t.begin() + j * 3 + 3);
        o.d.insert(o.d.end(), m->d.begin() + j * 2, m->d.begin() + j * 2 + 2);
    }
    for (size_t i = 0; i < npt; ++i) {
        if (!pmask[i]) continue;
        pidx[i] = (int)o.npt();
        o.xyz.insert(o.xyz.end(), m->xyz.begin() + i * 3, m->xyz.begin() + i * 3 + 3);
        o.rgb.insert(o.rgb.end(), m->rgb.begin() + i * 3, m->rgb.begin() + i * 3 + 3);
    }
    for (size_t k = 0; k < np; ++k) {
        const int pid = m->pt_idx[k], cid = m->cam_idx[k];
        if (!pmask[pid] || !cmask[cid]) continue;
        o.pt_idx.push_back(pidx[pid]); o.cam_idx.push_back(cidx[cid]); o.feat_id.push_back(m->feat_id[k]);
        o.xy.push_back(m->xy[2 * k]); o.xy.push_back(m->xy[2 * k + 1]);
    }
    *m = std::move(o);
    return removed;
}

int mi_model_counts(const mi_model *m, int64_t c[3])
{
    if (!m) return MI_BA_ERROR;
    c[0] = (int64_t)m->ncam(); c[1] = (int64_t)m->npt(); c[2] = (int64_t)m->nproj();
    return 0;
}

int mi_model_get(const mi_model *m, double *f, double *R, double *t, double *d, double *xyz, int *rgb, double *xy, int *pt_idx,
                 int *cam_idx, int64_t *feat_id)
{
    if (!m) return MI_BA_ERROR;
#define CP(dst, src) if (dst) std::memcpy(dst, (src).data(), (src).size() * sizeof((src)[0]))
    CP(f, m->f); CP(R, m->R); CP(t, m->t); CP(d, m->d); CP(xyz, m->xyz); CP(rgb, m->rgb); CP(xy, m->xy); CP(pt_idx, m->pt_idx); CP(cam_idx, m->cam_idx);
#undef CP
    if (feat_id) for (size_t i = 0; i < m->nproj(); ++i) feat_id[i] = m->feat_id[i];
    return 0;
}

double mi_model_rmse(const mi_model *m)
{
    if (!m || !m->nproj()) return 0.0;
    // (large models: the sum in a few pieces on host threads, added in piece order -- the same value whatever the thread timing)
    const size_t np = m->nproj();
    const unsigned nt = np < (1u << 20) ? 1u : io_threads();
    std::vector<double> part(nt, 0.0);
    auto piece = [&](unsigned k) {
        double s = 0.0;
        for (size_t o = np * k / nt, e = np * (k + 1) / nt; o < e; ++o) {
            const double *R = &m->R[(size_t)m->cam_idx[o] * 9], *t = &m->t[(size_t)m->cam_idx[o] * 3], *X = &m->xyz[(size_t)m->pt_idx[o] * 3];
            double q[3];
            for (int r = 0; r < 3; ++r) q[r] = R[3 * r] * X[0] + R[3 * r + 1] * X[1] + R[3 * r + 2] * X[2] + t[r];
            const double fx = m->f[m->cam_idx[o]];
            const double ex = m->xy[2 * o] - fx * q[0] / q[2], ey = m->xy[2 * o + 1] - fx * q[1] / q[2];
            s += ex * ex + ey * ey;
        }
        part[k] = s;
    };
    if (nt == 1) piece(0);
    else {
        std::vector<std::thread> th;
        for (unsigned k = 0; k < nt; ++k) th.emplace_back(piece, k);
        for (auto &t : th) t.join();
    }
    double s = 0.0;
    for (double v : part) s += v;
    return std::sqrt(s / (double)np);
}

int mi_model_set_focal_mask(mi_model *m, const int *mask, double weight)
{
    if (!m) return MI_BA_ERROR;
    m->focal_mask.clear(); m->focal_weight = 0.0;
    if (!mask || weight <= 0) return 0;
    m->focal_mask.assign(mask, mask + m->ncam());
    m->focal_weight = weight;
    return 0;
}

int mi_model_optimize(mi_model *m, const mi_ba_config *user, double info[MI_BA_INFOSZ])
{
    return mi_model_optimize_shard(m, user, info, nullptr);
}

int mi_model_optimize_shard(mi_model *m, const mi_ba_config *user, double info[MI_BA_INFOSZ], const mi_model_shard *shard)
{
    if (!m) return MI_BA_ERROR;
    mi_ba_config cfg;
    mi_ba_default_config(&cfg);
    if (user) {
        cfg.itmax = user->itmax; std::memcpy(cfg.opts, user->opts, sizeof(cfg.opts)); cfg.pcg_maxit = user->pcg_maxit; cfg.pcg_tol = user->pcg_tol;
        cfg.est_focal = user->est_focal; cfg.mcon = user->mcon; cfg.device = user->device; cfg.verbose = user->verbose;
        cfg.deterministic = user->deterministic;
        if (shard) { cfg.rank = user->rank; cfg.world = user->world; }
    }
    cfg.explicit_centers = 0; cfg.proj_sign = 1.0; // file convention: x^ = f (R X + t).xy / z
    const bool sharded = shard && cfg.world > 1;
    if (sharded && !shard->unique_id && !shard->host_allreduce) { g_err = "optimize_shard: world > 1 needs a transport"; return MI_BA_ERROR; }
    mi_ba_engine *e = mi_ba_create(&cfg);
    if (!e) g_err = mi_ba_last_error(nullptr);
    if (shard && shard->rendezvous && !shard->rendezvous(shard->rendezvous_user, e != nullptr)) { // every rank has its device, or none goes on
        if (e) { g_err = "optimize_shard: another rank failed to open its device"; mi_ba_destroy(e); }
        return MI_BA_ERROR;
    }
    if (!e) return MI_BA_ERROR;
    if (sharded) {
        const int rc = shard->host_allreduce ? mi_ba_comm_set_host_allreduce(e, shard->host_allreduce, shard->host_user)
                                             : mi_ba_comm_init(e, shard->unique_id);
        if (rc != 0) { g_err = mi_ba_last_error(e); mi_ba_destroy(e); return MI_BA_ERROR; }
    }
    const int ncam = (int)m->ncam(), npt = (int)m->npt(), cnp = mi_ba_cnp(e);
    miba::FocalLinks fl;
    if (!m->focal_mask.empty() && cfg.est_focal) { // shared intrinsics, as ParallelBA::SetFocalMask (pba.h:124)
        if (!miba::focal_mask_links(ncam, m->focal_mask.data(), m->nproj(), m->cam_idx.data(), m->focal_weight, fl)) {
            g_err = "incorrect focal mask"; mi_ba_destroy(e); return MI_BA_ERROR;
        }
        miba::focal_mask_average(ncam, m->focal_mask.data(), m->f.data(), 1);
    }
    std::vector<double> link_w(fl.a.size() * (size_t)cnp, 0.0);
    for (size_t q = 0; q < fl.a.size(); ++q) link_w[q * cnp + 6] = fl.w[q] * fl.w[q] / (cfg.f_scale * cfg.f_scale);
    std::vector<double> a0((size_t)ncam * cnp, 0.0);
    for (int j = 0; j < ncam; ++j) {
        for (int r = 0; r < 3; ++r) a0[(size_t)j * cnp + r] = m->t[(size_t)j * 3 + r];
        if (cfg.est_focal) a0[(size_t)j * cnp + 6] = m->f[j] * cfg.f_scale;
    }
    int ret = MI_BA_ERROR;
    if (mi_ba_set_cameras(e, ncam, a0.data(), m->R.data(), m->f.data()) == 0 && mi_ba_set_points(e, npt, m->xyz.data()) == 0 &&
        mi_ba_set_camera_links(e, (int)fl.a.size(), fl.a.data(), fl.b.data(), link_w.data()) == 0 &&
        mi_ba_set_observations(e, (int64_t)m->nproj(), m->xy.data(), m->pt_idx.data(), m->cam_idx.data()) == 0 && mi_ba_prepare(e) == 0) {
        ret = mi_ba_run(e, info);
        std::vector<double> a((size_t)ncam * cnp), R((size_t)ncam * 9);
        if (mi_ba_get_params(e, a.data(), m->xyz.data()) == 0 && mi_ba_get_rotations(e, R.data()) == 0) {
            m->R = R;
            for (int j = 0; j < ncam; ++j) {
                for (int r = 0; r < 3; ++r) m->t[(size_t)j * 3 + r] = a[(size_t)j * cnp + r];
                if (cfg.est_focal) m->f[j] = a[(size_t)j * cnp + 6] / cfg.f_scale;
            }
        } else ret = MI_BA_ERROR;
    }
    if (ret == MI_BA_ERROR) g_err = mi_ba_last_error(e);
    mi_ba_destroy(e);
    return ret;
}

// ---- several GPUs behind the one process the callers spawn --------------------------------------------------------------------
// MI_BA_GPUS=N: optimizer_all forks N - 1 further ranks BEFORE anything touches HIP (the model is already parsed, the children
// share its pages), every rank opens its own GPU (MI_BA_DEVICE + rank), takes its shard of the points and the ranks meet in
// the engine's one exchange per trial (mi_ba.h; RCCL over xGMI).  What the ranks share is one anonymous mapping: the RCCL
// unique id, one status word per rank, and -- for MI_BA_TRANSPORT=host only -- the arena of a host all-reduce that lets the
// ranks share ONE GPU (how the tests run this path on a one-GPU box).
struct RankShared {
    static constexpr int kMaxRanks = 64;
    static constexpr int64_t kChunk = 1 << 20; // doubles per rank and round of the host all-reduce
    std::atomic<int> status[kMaxRanks];        // 0 not there yet, 1 device open, -1 failed
    std::atomic<int> uid_ready, failed, arrived, generation;
    char uid[MI_BA_UNIQUE_ID_BYTES];
    int world;
    double *slot(int r) { return reinterpret_cast<double *>(reinterpret_cast<char *>(this) + 4096) + (size_t)r * kChunk; }
    static size_t bytes(int world, bool arena) { return 4096 + (arena ? sizeof(double) * (size_t)kChunk * world : 0); }
};
static_assert(sizeof(RankShared) <= 4096, "header page");

struct RankCtx { RankShared *sh; int rank; };

static bool wait_until(RankShared *sh, const std::function<bool()> &ready, double limit_s)
{
    const auto t0 = std::chrono::steady_clock::now();
    for (long spin = 0; !ready(); ++spin) {
        if (sh->failed.load()) return false;
        if ((spin & 1023) == 1023 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit_s) {
            sh->failed.store(1);
            return false;
        }
        sched_yield();
    }
    return true;
}

static bool rank_barrier(RankShared *sh)
{
    const int g = sh->generation.load();
    if (sh->arrived.fetch_add(1) + 1 == sh->world) { sh->arrived.store(0); sh->generation.fetch_add(1); return !sh->failed.load(); }
    return wait_until(sh, [&]() { return sh->generation.load() != g; }, 600.0);
}

static int rank_rendezvous(void *user, int ok)
{
    RankCtx *c = static_cast<RankCtx *>(user);
    c->sh->status[c->rank].store(ok ? 1 : -1);
    if (!ok) c->sh->failed.store(1);
    return wait_until(c->sh, [&]() {
        for (int r = 0; r < c->sh->world; ++r) if (c->sh->status[r].load() == 0) return false;
        return true; }, 600.0) ? 1 : 0;
}

// sum / maximum over the ranks in rank order: every rank computes the same bits
static int rank_host_allreduce(void *user, double *buf, int64_t count, int op)
{
    RankCtx *c = static_cast<RankCtx *>(user);
    RankShared *sh = c->sh;
    for (int64_t off = 0; off < count; off += RankShared::kChunk) {
        const int64_t len = std::min<int64_t>(RankShared::kChunk, count - off);
        std::memcpy(sh->slot(c->rank), buf + off, sizeof(double) * (size_t)len);
        if (!rank_barrier(sh)) return -1;
        for (int64_t i = 0; i < len; ++i) {
            double acc = sh->slot(0)[i];
            for (int r = 1; r < sh->world; ++r) { const double v = sh->slot(r)[i]; acc = op == 1 ? std::max(acc, v) : acc + v; }
            buf[off + i] = acc;
        }
        if (!rank_barrier(sh)) return -1;
    }
    return 0;
}

// what an NVM and the side files of one directory disagree on, in words ("" = they describe the same model to the
// side files' printing precision: %.9lf cameras, %lf = 6 decimals points and measurements)
static std::string describe_disagreement(const mi_model &a /* side files */, const mi_model &b /* NVM */)
{
    char buf[256];
    if (a.ncam() != b.ncam() || a.npt() != b.npt() || a.nproj() != b.nproj()) {
        std::snprintf(buf, sizeof buf, "side files %zu cameras / %zu points / %zu projections, NVM %zu / %zu / %zu",
                      a.ncam(), a.npt(), a.nproj(), b.ncam(), b.npt(), b.nproj());
        return buf;
    }
    auto maxdiff = [](const std::vector<double> &x, const std::vector<double> &y) {
        double d = 0;
        for (size_t i = 0; i < x.size(); ++i) d = std::max(d, std::fabs(x[i] - y[i]) / (1.0 + std::fabs(y[i])));
        return d;
    };
    const double dr = maxdiff(a.R, b.R), dt = maxdiff(a.t, b.t), df = maxdiff(a.f, b.f), dx = maxdiff(a.xyz, b.xyz), dm = maxdiff(a.xy, b.xy);
    if (dr > 1e-7 || dt > 1e-6 || df > 1e-7 || dx > 1e-5 || dm > 1e-5 || a.cam_idx != b.cam_idx) {
        std::snprintf(buf, sizeof buf, "relative differences R %.2g, t %.2g, f %.2g, points %.2g, measurements %.2g%s", dr, dt, df, dx, dm,
                      a.cam_idx != b.cam_idx ? ", different tracks" : "");
        return buf;
    }
    return "";
}

// optimizer_all <dir>/ A B C D E F.  The reference's main (Optimizer/main_all.cpp) is missing from its
// tree, so the meaning of A..F is the one documented in INTEGRATION.md:
//   A  which blocks are free (utils.h:112 whatToCalibrate): < 0 -> intrinsics held (cnp 6), >= 0 -> pose + focal (cnp 7)
//   B  maximum LM iterations          C, D  accepted, unused (always 16 / 32 at the call sites)
//   E  accepted, unused (curErrorThresh, utils.cpp:35)        F  back end selector: every value runs this engine
// Input: the side files the callers write (mi_model_load_side_files; at the two TriangulationMultiStep call sites they are the
// ONLY input, :1024-1095, :1308-1382); <dir>/outputVSFM_GB.nvm only when they are absent.  Output: <dir>/outputVSFM_GB.nvm.
// Environment: MI_BA_DEVICE (first GPU, default 0), MI_BA_GPUS (ranks = GPUs, default 1), MI_BA_TRANSPORT (rccl | host).
int mi_optimizer_all_main(int argc, char **argv)
{
    if (argc < 2) {
        std::fprintf(stderr, "usage: %s <cluster dir>/ [A B C D E F]\n", argc ? argv[0] : "optimizer_all");
        return 2;
    }
    std::string dir = argv[1];
    if (!dir.empty() && dir.back() != '/') dir += '/';
    const int A = argc > 2 ? std::atoi(argv[2]) : 0;
    const int B = argc > 3 ? std::atoi(argv[3]) : 50;
    std::printf("optimizer_all: A=%d (%s), B=%d (iteration cap); ignored: C=%s D=%s E=%s F=%s (meaning not recoverable: the reference's main_all.cpp is missing)\n",
                A, A >= 0 ? "poses, focal lengths and points free" : "focal lengths held", B > 0 ? B : 1,
                argc > 4 ? argv[4] : "-", argc > 5 ? argv[5] : "-", argc > 6 ? argv[6] : "-", argc > 7 ? argv[7] : "-");
    const std::string nvm = dir + "outputVSFM_GB.nvm";
    const auto t_start = std::chrono::steady_clock::now();
    auto seconds = [&]() { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count(); };
    auto exists = [&](const char *name) { return access((dir + name).c_str(), R_OK) == 0; };
    const bool have_side = exists("RTglobalmapped_for_optimisation.txt") && exists("focal_for_optimisation.txt") && exists("ours_new.txt");
    // one rank: the HIP context (a few tenths of a second in a fresh process) comes up on another thread while this one reads the
    // model.  (Several ranks fork first and must not have touched HIP before: they open their devices after the fork.)
    std::thread hip_warm_up;
    {
        const char *gpus_env = std::getenv("MI_BA_GPUS"), *dev_env = std::getenv("MI_BA_DEVICE");
        if (!gpus_env || std::atoi(gpus_env) == 1) {
            const int dev = dev_env ? std::atoi(dev_env) : 0;
            hip_warm_up = std::thread([dev]() { if (hipSetDevice(dev) == hipSuccess) (void)hipFree(nullptr); });
        }
    }
    struct JoinWarmUp { std::thread &t; ~JoinWarmUp() { if (t.joinable()) t.join(); } } join_warm_up{ hip_warm_up };
    mi_model *m = nullptr;
    if (have_side) {
        m = mi_model_load_side_files(dir.c_str());
        if (m && exists("outputVSFM_GB.nvm")) {   // both: the side files are what the caller just wrote, the NVM may be a stage behind
            mi_model *old = mi_model_load_nvm(nvm.c_str());
            const std::string diff = old ? describe_disagreement(*m, *old) : std::string("the NVM does not parse: ") + g_err;
            if (!diff.empty()) std::printf("optimizer_all: %s and the side files disagree (%s); the side files win\n", nvm.c_str(), diff.c_str());
            mi_model_free(old);
        }
        if (m) std::printf("optimizer_all: input = RTglobalmapped_for_optimisation.txt + focal_for_optimisation.txt + ours_new.txt%s%s\n",
                           exists("Invmap.txt") ? " + Invmap.txt" : "", exists("Distortion.txt") ? " + Distortion.txt" : "");
    } else {
        m = mi_model_load_nvm(nvm.c_str());
        if (m) std::printf("optimizer_all: no side files in %s, input = outputVSFM_GB.nvm\n", dir.c_str());
    }
    const double t_load = seconds();
    if (!m) { std::fprintf(stderr, "optimizer_all: %s\n", mi_model_last_error()); return 1; }
    mi_ba_config cfg;
    mi_ba_default_config(&cfg);
    cfg.est_focal = A >= 0 ? 1 : 0;
    cfg.itmax = B > 0 ? B : 1;
    if (const char *dev = std::getenv("MI_BA_DEVICE")) cfg.device = std::atoi(dev);
    if (const char *det = std::getenv("MI_BA_DETERMINISTIC")) { const int v = std::atoi(det); cfg.deterministic = v == 2 ? 2 : v != 0; }   // mi_ba.h: 1 ordered hand-over, 2 all rows
    const char *gpus = std::getenv("MI_BA_GPUS"), *transport = std::getenv("MI_BA_TRANSPORT");
    const int world = gpus ? std::atoi(gpus) : 1;
    const bool host_transport = transport && std::strcmp(transport, "host") == 0;
    if (world < 1 || world > RankShared::kMaxRanks || (transport && !host_transport && std::strcmp(transport, "rccl") != 0)) {
        std::fprintf(stderr, "optimizer_all: MI_BA_GPUS must be 1..%d and MI_BA_TRANSPORT rccl or host\n", RankShared::kMaxRanks);
        mi_model_free(m);
        return 2;
    }
    double info[MI_BA_INFOSZ] = { 0 };
    const double rmse0 = mi_model_rmse(m);
    std::printf("%zu cameras; %zu 3D points; %zu projections\n", m->ncam(), m->npt(), m->nproj());
    int it = MI_BA_ERROR;
    if (world == 1) {
        if (hip_warm_up.joinable()) hip_warm_up.join();
        it = mi_model_optimize(m, &cfg, info);
    } else {
        const size_t bytes = RankShared::bytes(world, host_transport);
        void *mem = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (mem == MAP_FAILED) { std::fprintf(stderr, "optimizer_all: cannot map the ranks' shared page\n"); mi_model_free(m); return 1; }
        RankShared *sh = new (mem) RankShared; // zero pages: every atomic starts at 0
        sh->world = world;
        std::fflush(stdout); std::fflush(stderr);
        int rank = 0;
        std::vector<pid_t> kids;
        for (int r = 1; r < world; ++r) { // no HIP call has been made yet: the children start with a clean runtime
            const pid_t pid = fork();
            if (pid < 0) { sh->failed.store(1); std::fprintf(stderr, "optimizer_all: fork failed\n"); break; }
            if (pid == 0) { rank = r; kids.clear(); break; }
            kids.push_back(pid);
        }
        RankCtx ctx{ sh, rank };
        // Watchdog, one thread per rank (started after the forks): a rank that dies inside the solve -- GPU fault, out of memory --
        // would leave the others in ncclAllReduce / the host arena's barrier and the stage program that ran optimizer_all hanging.
        // Rank 0 polls its children, the children their parent and the shared failure word; a failure seen, the normal error path
        // gets two seconds to report, then the rank leaves with a non-zero status (rank 0 after ending its children).
        std::atomic<int> solve_over{ 0 };
        std::vector<int> kid_status(kids.size(), -1);           // -1 running, else the exit status the watchdog reaped
        const pid_t parent = getppid();
        std::thread watchdog([&, rank]() {
            int failed_for = 0;
            while (!solve_over.load()) {
                std::this_thread::sleep_for(std::chrono::milliseconds(50));
                if (rank == 0) {
                    for (size_t k = 0; k < kids.size(); ++k) {
                        int st = 0;
                        if (kid_status[k] >= 0 || waitpid(kids[k], &st, WNOHANG) != kids[k]) continue;
                        kid_status[k] = (WIFEXITED(st) && WEXITSTATUS(st) == 0) ? 0 : 1;
                        if (kid_status[k] && !solve_over.load()) sh->failed.store(1);
                    }
                } else if (getppid() != parent) sh->failed.store(1);
                if (!sh->failed.load()) continue;
                if (++failed_for < 40) continue;                // 2 s for the rank's own error path
                if (solve_over.load()) break;
                std::fprintf(stderr, "optimizer_all: rank %d: another rank failed, leaving\n", rank);
                std::fflush(stderr);
                if (rank == 0) for (size_t k = 0; k < kids.size(); ++k) if (kid_status[k] < 0) kill(kids[k], SIGKILL);
                std::_Exit(1);
            }
        });
        if (const char *die = std::getenv("MI_BA_TEST_RANK_DIES")) {   // test hook "rank:milliseconds": that rank ends abruptly
            int r = -1, ms = 0;
            if (std::sscanf(die, "%d:%d", &r, &ms) == 2 && r == rank)
                std::thread([ms]() { std::this_thread::sleep_for(std::chrono::milliseconds(ms)); std::_Exit(3); }).detach();
        }
        cfg.rank = rank; cfg.world = world;
        if (!host_transport) cfg.device += rank;
        if (rank != 0) cfg.verbose = 0;
        mi_model_shard shard = {};
        shard.rendezvous = rank_rendezvous; shard.rendezvous_user = &ctx;
        bool ok = !sh->failed.load();
        if (host_transport) { shard.host_allreduce = rank_host_allreduce; shard.host_user = &ctx; }
        else if (ok) {
            if (rank == 0) {
                if (mi_ba_comm_unique_id(sh->uid) == 0) sh->uid_ready.store(1);
                else { sh->failed.store(1); ok = false; g_err = "cannot create the RCCL unique id"; }
            } else ok = wait_until(sh, [&]() { return sh->uid_ready.load() != 0; }, 600.0);
            shard.unique_id = sh->uid;
        }
        if (ok) it = mi_model_optimize_shard(m, &cfg, info, &shard);
        if (it == MI_BA_ERROR) sh->failed.store(1);
        solve_over.store(1);
        watchdog.join();
        if (rank != 0) {
            if (it == MI_BA_ERROR) std::fprintf(stderr, "optimizer_all: rank %d: %s\n", rank, g_err.empty() ? "another rank failed" : g_err.c_str());
            std::fflush(stderr);
            std::_Exit(it == MI_BA_ERROR ? 1 : 0); // the answer is rank 0's to write
        }
        for (size_t k = 0; k < kids.size(); ++k) {
            int st = 0;
            const bool bad = kid_status[k] >= 0 ? kid_status[k] != 0 : (waitpid(kids[k], &st, 0) < 0 || !WIFEXITED(st) || WEXITSTATUS(st) != 0);
            if (bad) { if (it != MI_BA_ERROR) g_err = "a rank ended abnormally"; it = MI_BA_ERROR; }
        }
        munmap(mem, bytes);
        if (it != MI_BA_ERROR) std::printf("optimizer_all: %d ranks, %s\n", world, host_transport ? "host all-reduce on one GPU" : "RCCL all-reduce, one GPU each");
    }
    if (it == MI_BA_ERROR) { std::fprintf(stderr, "optimizer_all: %s\n", mi_model_last_error()); mi_model_free(m); return 1; }
    const double t_solve = seconds();
    std::printf("optimizer_all: %d LM iterations (stop %d), reprojection RMSE %.6f -> %.6f px\n", it, (int)info[6], rmse0, mi_model_rmse(m));
    const int rc = mi_model_save_nvm(m, nvm.c_str());
    std::printf("optimizer_all: read %.2f s, plan + upload + solve %.2f s, write %.2f s\n", t_load, t_solve - t_load, seconds() - t_solve);
    mi_model_free(m);
    return rc == 0 ? 0 : 1;
}

} // extern "C"

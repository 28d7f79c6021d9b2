// merge.cpp -- the cluster-merge step around the GPU similarity alignment (include/mi_merge.h): correspondences by point id,
// the second model brought into the first one's frame, tracks fused, side files, and the Merge3D3D process.
// Host code: the lists are touched once each; the O(n^2) + 5000-round alignment is mi_sim3.h's kernels.
#include <algorithm>
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <fstream>
#include <set>
#include <sstream>
#include <string>
#include <unordered_map>
#include <vector>

#include <spawn.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>

#include "model.h"

#include "../../include/mi_merge.h"
#include "../../include/mi_sim3.h"

extern char **environ;

namespace {

// a model as Merge3D3D.cpp's ReadNVM (:416-529) sees it
struct TrackView {
    std::vector<size_t> ptr;        // per point: its kept views [ptr[i], ptr[i+1]) in view[]
    std::vector<size_t> view;       // index into the model's projection arrays
    std::vector<long long> id;      // per point
};

TrackView read_tracks(const mi_model &m)
{
    TrackView tv;
    const size_t n = m.npt(), k = m.nproj();
    tv.ptr.assign(n + 1, 0); tv.id.assign(n, -1);
    tv.view.reserve(k);
    size_t j = 0;
    for (size_t i = 0; i < n; ++i) {
        int prev = -1;
        for (; j < k && m.pt_idx[j] == (int)i; ++j) {
            tv.id[i] = m.feat_id[j];                         // the last view's field wins, :515
            if (m.cam_idx[j] == prev) continue;              // :519
            prev = m.cam_idx[j];
            tv.view.push_back(j);
        }
        tv.ptr[i + 1] = tv.view.size();
    }
    return tv;
}

void centre_of(const double *R, const double *t, double c[3]) // getCfromRT, :124-131
{
    for (int r = 0; r < 3; ++r) c[r] = -(R[r] * t[0] + R[3 + r] * t[1] + R[6 + r] * t[2]);
}

std::string dirname_of_self()
{
    char buf[4096];
    const ssize_t len = readlink("/proc/self/exe", buf, sizeof buf - 1);
    if (len <= 0) return ".";
    buf[len] = 0;
    std::string s(buf);
    const size_t cut = s.rfind('/');
    return cut == std::string::npos ? "." : s.substr(0, cut);
}

struct Similarity {
    double s = 1.0, R[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 }, T[3] = { 0, 0, 0 };
    void move(const double *x, double *y) const // y = R (s x) + T
    {
        const double p0 = s * x[0], p1 = s * x[1], p2 = s * x[2];
        for (int r = 0; r < 3; ++r) y[r] = R[3 * r] * p0 + R[3 * r + 1] * p1 + R[3 * r + 2] * p2 + T[r];
    }
    // R' = Rj R^T, c' = R (s c) + T, t' = -R' c'   (Merge3D3D.cpp:739-772)
    void move_camera(const double *Rj, const double *tj, double *Rn, double *tn, double *cn) const
    {
        double c[3];
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) Rn[3 * a + b] = Rj[3 * a] * R[3 * b] + Rj[3 * a + 1] * R[3 * b + 1] + Rj[3 * a + 2] * R[3 * b + 2];
        centre_of(Rj, tj, c);
        move(c, cn);
        for (int a = 0; a < 3; ++a) tn[a] = -(Rn[3 * a] * cn[0] + Rn[3 * a + 1] * cn[1] + Rn[3 * a + 2] * cn[2]);  // getTfromRC, :133-137
    }
};

// correspondences, Merge3D3D.cpp:646-683: for every point of the first model the FIRST point of the second with the same id
std::vector<std::pair<int, int>> find_matches(const TrackView &t1, const TrackView &t2)
{
    std::unordered_map<long long, int> first_with_id;
    first_with_id.reserve(t2.id.size() * 2);
    for (size_t i = 0; i < t2.id.size(); ++i) first_with_id.emplace(t2.id[i], (int)i); // emplace keeps the earlier entry
    std::vector<std::pair<int, int>> match;
    for (size_t i = 0; i < t1.id.size(); ++i) {
        auto it = first_with_id.find(t1.id[i]);
        if (it != first_with_id.end()) match.emplace_back((int)i, it->second);
    }
    return match;
}

bool by_second(const std::pair<int, int> &a, const std::pair<int, int> &b) { return a.second < b.second; } // sortbysec, :139-143

// the merged model.  erase = false: Merge3D3D.cpp:775-798 (every point of the second model stays); erase = true:
// MergeClusters.cpp:971-995 (a matched point of the second model is removed once its views are handed over)
mi_model *fuse(const mi_model &m1, const mi_model &m2, const TrackView &t1, const TrackView &t2, std::vector<std::pair<int, int>> match,
               const Similarity &g, bool erase, bool keep_colours)
{
    const size_t n1 = m1.npt(), n2 = m2.npt(), c1 = m1.ncam(), c2 = m2.ncam();
    // which track of the second model goes to which point of the first.  The reference walks the pairs sorted by their second
    // index from the back and, when erasing, takes "the element now at position second": were two points of the first model
    // matched to the same point, the later one receives (and removes) the next surviving track -- reproduced, with the same
    // std::sort on the same sequence so that even the order among equal keys is the reference's
    std::vector<int> partner(n1, -1);
    std::vector<char> gone(n2, 0);
    std::sort(match.begin(), match.end(), by_second);
    std::set<int> alive;
    if (erase) for (size_t i = 0; i < n2; ++i) alive.insert(alive.end(), (int)i);
    for (size_t q = match.size(); q-- > 0;) {
        int j = match[q].second;
        if (erase) {
            auto it = alive.lower_bound(j);
            if (it == alive.end()) continue;                             // past the end of the list in the reference
            j = *it;
            alive.erase(it);
            gone[j] = 1;
        }
        partner[match[q].first] = j;
    }
    mi_model *out = new mi_model;
    mi_model &m = *out;
    // cameras: the first model's, then the second's in the first one's frame
    m.names = m1.names; m.names.insert(m.names.end(), m2.names.begin(), m2.names.end());
    m.f = m1.f; m.f.insert(m.f.end(), m2.f.begin(), m2.f.end());
    m.d = m1.d; m.d.insert(m.d.end(), m2.d.begin(), m2.d.end());
    m.R = m1.R; m.t = m1.t;
    m.R.resize((c1 + c2) * 9); m.t.resize((c1 + c2) * 3);
    for (size_t j = 0; j < c2; ++j) {
        double cn[3];
        g.move_camera(&m2.R[j * 9], &m2.t[j * 3], &m.R[(c1 + j) * 9], &m.t[(c1 + j) * 3], cn);
    }
    // points: the first model's, then the second's that stay
    std::vector<int> keep2;
    for (size_t i = 0; i < n2; ++i) if (!gone[i]) keep2.push_back((int)i);
    const size_t nk = keep2.size();
    m.xyz = m1.xyz; m.xyz.resize((n1 + nk) * 3);
    for (size_t k = 0; k < nk; ++k) g.move(&m2.xyz[3 * (size_t)keep2[k]], &m.xyz[3 * (n1 + k)]);
    std::vector<char> white(n2, 0);                                      // the colouring marks the pairs as found, :662-668
    std::vector<char> blue(n1, 0);
    for (const auto &pr : match) { blue[pr.first] = 1; white[pr.second] = 1; }
    m.rgb.resize((n1 + nk) * 3);
    for (size_t i = 0; i < n1 + nk; ++i) {
        const bool a = i < n1;
        const size_t i2 = a ? 0 : (size_t)keep2[i - n1];
        const int *from = a ? &m1.rgb[3 * i] : &m2.rgb[3 * i2];
        int paint[3] = { a ? 255 : 0, a ? 0 : 255, 0 };                  // ReadNVM colour modes 1 and 3, :487-501
        if (a && blue[i]) { paint[0] = 0; paint[1] = 0; paint[2] = 255; }
        if (!a && white[i2]) { paint[0] = paint[1] = paint[2] = 255; }
        for (int r = 0; r < 3; ++r) m.rgb[3 * i + r] = keep_colours ? from[r] : paint[r];
    }
    auto put_views = [&](const mi_model &src_m, const TrackView &tv, size_t i, int cam_shift, size_t pt, long long id) {
        for (size_t v = tv.ptr[i]; v < tv.ptr[i + 1]; ++v) {
            const size_t j = tv.view[v];
            m.cam_idx.push_back(src_m.cam_idx[j] + cam_shift); m.pt_idx.push_back((int)pt); m.feat_id.push_back(id); // SaveNVM, :558
            m.xy.push_back(src_m.xy[2 * j]); m.xy.push_back(src_m.xy[2 * j + 1]);
        }
    };
    m.cam_idx.reserve(t1.view.size() + 2 * t2.view.size());
    for (size_t i = 0; i < n1; ++i) {
        put_views(m1, t1, i, 0, i, t1.id[i]);
        if (partner[i] >= 0) put_views(m2, t2, (size_t)partner[i], (int)c1, i, t1.id[i]);
    }
    for (size_t k = 0; k < nk; ++k) put_views(m2, t2, (size_t)keep2[k], (int)c1, n1 + k, t2.id[(size_t)keep2[k]]);
    return out;
}

} // namespace

extern "C" {

void mi_merge_default_config(mi_merge_config *cfg)
{
    std::memset(cfg, 0, sizeof *cfg);
    cfg->rounds = 5000;
}

mi_model *mi_model_merge_3d3d(const mi_model *first, const mi_model *second, const mi_merge_config *user, mi_merge_result *res)
{
    if (!first || !second) { miba::set_model_error("merge: null model"); return nullptr; }
    mi_merge_config cfg;
    mi_merge_default_config(&cfg);
    if (user) cfg = *user;
    const mi_model &m1 = *first, &m2 = *second;
    const TrackView t1 = read_tracks(m1), t2 = read_tracks(m2);
    const std::vector<std::pair<int, int>> match = find_matches(t1, t2);
    const size_t nc = match.size();
    if (nc > (size_t)MI_SIM3_MAX_POINTS) { miba::set_model_error("merge: more correspondences than MI_SIM3_MAX_POINTS"); return nullptr; }
    if (nc < 2) { miba::set_model_error("merge: fewer than 2 common points"); return nullptr; }
    std::vector<double> src(nc * 3), dst(nc * 3);
    for (size_t q = 0; q < nc; ++q)
        for (int r = 0; r < 3; ++r) { dst[3 * q + r] = m1.xyz[3 * (size_t)match[q].first + r]; src[3 * q + r] = m2.xyz[3 * (size_t)match[q].second + r]; }
    // scale, R, t: second -> first, :691
    mi_sim3_config sc;
    mi_sim3_default_config(&sc);
    sc.rounds = cfg.rounds; sc.device = cfg.device;
    Similarity g;
    double ms = 0.0;
    int best = -1;
    if (mi_sim3_align(&sc, (int)nc, src.data(), dst.data(), &g.s, g.R, g.T, nullptr, &best, &ms) != 0) {
        miba::set_model_error(std::string("merge: ") + mi_sim3_last_error());
        return nullptr;
    }
    if (res) {
        res->ncorr = (int64_t)nc; res->scale = g.s; std::memcpy(res->R, g.R, sizeof g.R); std::memcpy(res->t, g.T, sizeof g.T);
        res->best_round = best; res->device_ms = ms; res->ncam_first = (int64_t)m1.ncam(); res->npt_first = (int64_t)m1.npt();
    }
    return fuse(m1, m2, t1, t2, match, g, false, cfg.keep_colours != 0);
}

mi_model *mi_model_merge_with_transform(const mi_model *first, const mi_model *second, double scale, const double R[9], const double t[3],
                                        int erase_matched, int keep_colours, mi_merge_result *res)
{
    if (!first || !second || !R || !t) { miba::set_model_error("merge: null argument"); return nullptr; }
    const TrackView t1 = read_tracks(*first), t2 = read_tracks(*second);
    const std::vector<std::pair<int, int>> match = find_matches(t1, t2);
    Similarity g;
    g.s = scale; std::memcpy(g.R, R, sizeof g.R); std::memcpy(g.T, t, sizeof g.T);
    if (res) {
        res->ncorr = (int64_t)match.size(); res->scale = scale; std::memcpy(res->R, R, sizeof g.R); std::memcpy(res->t, t, sizeof g.T);
        res->best_round = -1; res->device_ms = 0.0; res->ncam_first = (int64_t)first->ncam(); res->npt_first = (int64_t)first->npt();
    }
    return fuse(*first, *second, t1, t2, match, g, erase_matched != 0, keep_colours != 0);
}

int mi_model_score_transform(const mi_model *first, const mi_model *second, double scale, const double R[9], const double t[3],
                             mi_merge_score *score, double *dist, const char *debug_path)
{
    if (!first || !second || !R || !t || !score) { miba::set_model_error("score: null argument"); return -1; }
    const mi_model &m1 = *first, &m2 = *second;
    const TrackView t1 = read_tracks(m1), t2 = read_tracks(m2);
    const std::vector<std::pair<int, int>> match = find_matches(t1, t2);
    if (match.empty()) { miba::set_model_error("score: no common points"); return -1; }
    Similarity g;
    g.s = scale; std::memcpy(g.R, R, sizeof g.R); std::memcpy(g.T, t, sizeof g.T);
    std::vector<double> Rn(m2.ncam() * 9), cn(m2.ncam() * 3), tn(3);
    for (size_t j = 0; j < m2.ncam(); ++j) g.move_camera(&m2.R[j * 9], &m2.t[j * 3], &Rn[j * 9], tn.data(), &cn[j * 3]);
    FILE *dbg = debug_path ? std::fopen(debug_path, "w") : nullptr;
    std::vector<double> errs, dists;
    double sum_err = 0.0;
    for (size_t q = 0; q < match.size(); ++q) {                          // MergeClusters.cpp:775-836
        const double *p1 = &m1.xyz[3 * (size_t)match[q].first];
        double p2[3];
        g.move(&m2.xyz[3 * (size_t)match[q].second], p2);
        const double d = std::sqrt((p2[0] - p1[0]) * (p2[0] - p1[0]) + (p2[1] - p1[1]) * (p2[1] - p1[1]) + (p2[2] - p1[2]) * (p2[2] - p1[2]));
        dists.push_back(d);
        if (dist) dist[q] = d;
        if (dbg) std::fprintf(dbg, "\n[%lf %lf %lf] [%lf %lf %lf] %lf\n", p1[0], p1[1], p1[2], p2[0], p2[1], p2[2], d);
        const size_t i2 = (size_t)match[q].second;
        for (size_t v = t2.ptr[i2]; v < t2.ptr[i2 + 1]; ++v) {           // the FIRST model's point in the second model's moved cameras
            const size_t o = t2.view[v];
            const int c = m2.cam_idx[o];
            const double *Rc = &Rn[(size_t)c * 9], *cc = &cn[(size_t)c * 3];
            const double w0 = p1[0] - cc[0], w1 = p1[1] - cc[1], w2 = p1[2] - cc[2];
            const double X = Rc[0] * w0 + Rc[1] * w1 + Rc[2] * w2, Y = Rc[3] * w0 + Rc[4] * w1 + Rc[5] * w2, Z = Rc[6] * w0 + Rc[7] * w1 + Rc[8] * w2;
            const double mx = X / Z, my = Y / Z, r2 = 1 + m2.d[2 * (size_t)c] * (mx * mx + my * my);
            const double ux = r2 * mx * m2.f[c], uy = r2 * my * m2.f[c];
            const double x = m2.xy[2 * o], y = m2.xy[2 * o + 1];
            const double e = std::sqrt((x - ux) * (x - ux) + (y - uy) * (y - uy));
            if (dbg) std::fprintf(dbg, "[%lf %lf] [%lf %lf] %lf\n", x, y, ux, uy, e);
            errs.push_back(e);
            sum_err += e;
        }
    }
    std::sort(errs.begin(), errs.end());
    std::sort(dists.begin(), dists.end());
    double sum_dist = 0.0;
    for (double d : dists) sum_dist += d;
    const double mean_dist = sum_dist / (double)dists.size();
    int64_t inl = 0;
    for (double d : dists) if (d < mean_dist) ++inl;
    score->nproj = (int64_t)errs.size();
    score->mean_reproj = errs.empty() ? 0.0 : sum_err / (double)errs.size();
    score->median_reproj = errs.empty() ? 0.0 : errs[errs.size() / 2];
    score->median_dist = dists[dists.size() / 2];
    score->mean_dist = mean_dist;
    score->inliers = inl;
    score->ncorr = (int64_t)match.size();
    if (dbg) { std::fprintf(dbg, "ReprojErr Mean %lf Median %lf\n", score->mean_reproj, score->median_reproj); std::fclose(dbg); }
    return 0;
}

int mi_model_save_merge_side_files(const mi_model *mp, const char *dir_c, int64_t nply_tail)
{
    if (!mp || !dir_c) { miba::set_model_error("side files: null argument"); return -1; }
    const mi_model &m = *mp;
    const std::string dir = std::string(dir_c) + "/";
    FILE *fr = std::fopen((dir + "RTglobalmapped_for_optimisation.txt").c_str(), "w");
    FILE *ff = std::fopen((dir + "focal_for_optimisation.txt").c_str(), "w");
    FILE *fm = std::fopen((dir + "Invmap.txt").c_str(), "w");
    FILE *fo = std::fopen((dir + "ours_new.txt").c_str(), "w");
    FILE *fp = std::fopen((dir + "merged.ply").c_str(), "w");
    if (!fr || !ff || !fm || !fo || !fp) {
        for (FILE *f : { fr, ff, fm, fo, fp }) if (f) std::fclose(f);
        miba::set_model_error("side files: cannot write into " + dir);
        return -1;
    }
    std::fprintf(ff, "%d\n", (int)m.ncam());                             // :822
    for (size_t j = 0; j < m.ncam(); ++j) {
        const double *R = &m.R[j * 9];
        double c[3];
        centre_of(R, &m.t[j * 3], c);
        std::fprintf(fr, "%.9lf %.9lf %.9lf %.9lf %.9lf %.9lf %.9lf %.9lf %.9lf\n", R[0], R[1], R[2], R[3], R[4], R[5], R[6], R[7], R[8]);
        std::fprintf(fr, "%.9lf %.9lf %.9lf\n", c[0], c[1], c[2]);
        std::fprintf(ff, "%s %.9lf %d\n", m.names[j].c_str(), m.f[j], 2);
        std::fprintf(fm, "%d %s\n", (int)j, m.names[j].c_str());
    }
    size_t v = 0;
    for (size_t i = 0; i < m.npt(); ++i) {                               // :839-855
        size_t ve = v;
        while (ve < m.nproj() && m.pt_idx[ve] == (int)i) ++ve;
        std::fprintf(fo, "%lf %lf %lf %d %d %d %d ", m.xyz[3 * i], m.xyz[3 * i + 1], m.xyz[3 * i + 2], m.rgb[3 * i], m.rgb[3 * i + 1], m.rgb[3 * i + 2], (int)(ve - v));
        for (; v < ve; ++v) std::fprintf(fo, "%d %lld %lf %lf ", m.cam_idx[v], m.feat_id[v], m.xy[2 * v], m.xy[2 * v + 1]);
        std::fprintf(fo, "\n");
    }
    // :864-898: room for the header first, the vertex count is known only at the end
    std::fprintf(fp, "      \n                              \n                              \n");
    std::fprintf(fp, "property float x\nproperty float y\nproperty float z\nproperty uchar red\nproperty uchar green\nproperty uchar blue\nend_header\n");
    long long count = 0;
    auto vertex = [&](size_t i) {
        std::fprintf(fp, "%.10lf %.10lf %.10lf %d %d %d\n", m.xyz[3 * i], m.xyz[3 * i + 1], m.xyz[3 * i + 2], m.rgb[3 * i], m.rgb[3 * i + 1], m.rgb[3 * i + 2]);
        ++count;
    };
    for (size_t i = 0; i < m.npt(); ++i) vertex(i);
    const size_t tail = nply_tail > 0 ? std::min<size_t>((size_t)nply_tail, m.npt()) : 0;
    for (size_t i = m.npt() - tail; i < m.npt(); ++i) vertex(i);
    std::fseek(fp, 0, SEEK_SET);
    std::fprintf(fp, "ply\nformat ascii 1.0\nelement vertex %d", (int)count);
    bool good = true;
    for (FILE *f : { fr, ff, fm, fo, fp }) good = (std::fclose(f) == 0) && good;
    if (!good) { miba::set_model_error("side files: write error in " + dir); return -1; }
    return 0;
}

static std::string optimizer_path()
{
    if (const char *o = std::getenv("MI_OPTIMIZER_ALL")) return o;
    return dirname_of_self() + "/optimizer_all";
}

// "Run bundler": a child process, as in the reference (whose system() result is not looked at either)
static bool run_optimizer(const std::string &folder, const char *A, const char *B, const char *E)
{
    const std::string optimizer = optimizer_path(), dir_arg = folder + "/";
    std::fflush(stdout); std::fflush(stderr);
    const char *args[] = { optimizer.c_str(), dir_arg.c_str(), A, B, "16", "32", E, "0", nullptr };
    pid_t pid = 0;
    if (posix_spawn(&pid, optimizer.c_str(), nullptr, nullptr, const_cast<char *const *>(args), environ) != 0) {
        std::fprintf(stderr, "merge: cannot start %s\n", optimizer.c_str());
        return false;
    }
    int st = 0;
    waitpid(pid, &st, 0);
    return WIFEXITED(st) && WEXITSTATUS(st) == 0;
}

struct MergeLine { std::string merged, first, second; };

static bool read_merge_list(const std::string &out_dir, std::vector<MergeLine> &lines)
{
    std::ifstream list(out_dir + "/dendogram/fullclustermergeinfo.txt");
    if (!list) { std::fprintf(stderr, "merge: cannot open %s/dendogram/fullclustermergeinfo.txt\n", out_dir.c_str()); return false; }
    std::string line;
    while (std::getline(list, line)) {
        std::istringstream toks(line);
        MergeLine l;                                                     // "new first second", Merge3D3D.cpp:590-593
        if (toks >> l.merged >> l.first >> l.second) lines.push_back(l);
    }
    return true;
}

static void print_transform(const mi_merge_result &r)
{
    std::printf("Scale:%g\nR:\n%g %g %g\n%g %g %g\n%g %g %g\nT:\n%g  %g  %g\n", r.scale, r.R[0], r.R[1], r.R[2], r.R[3], r.R[4], r.R[5],
                r.R[6], r.R[7], r.R[8], r.t[0], r.t[1], r.t[2]);
}

int mi_merge3d3d_main(int argc, char **argv)
{
    if (argc < 2) { std::printf("Invalid arguments\n ./a.out <Output DIR>\n"); return 0; }   // :567-571
    const std::string out_dir = argv[1];
    mi_merge_config cfg;
    mi_merge_default_config(&cfg);
    if (const char *dev = std::getenv("MI_BA_DEVICE")) cfg.device = std::atoi(dev);
    std::printf("\n\nLoading cluster details ...\n");
    std::vector<MergeLine> lines;
    if (!read_merge_list(out_dir, lines)) return 1;
    int status = 0;
    for (const MergeLine &l : lines) {
        const std::string nvm1 = out_dir + "/dendogram/cluster" + l.first + "/outputVSFM_GB.nvm", nvm2 = out_dir + "/dendogram/cluster" + l.second + "/outputVSFM_GB.nvm";
        std::printf("%s\n%s\n", nvm1.c_str(), nvm2.c_str());
        mi_model *m1 = mi_model_load_nvm(nvm1.c_str()), *m2 = m1 ? mi_model_load_nvm(nvm2.c_str()) : nullptr;
        mi_merge_result r;
        mi_model *m = m2 ? mi_model_merge_3d3d(m1, m2, &cfg, &r) : nullptr;
        if (!m) {
            std::fprintf(stderr, "Merge3D3D: clusters %s + %s: %s\n", l.first.c_str(), l.second.c_str(), mi_model_last_error());
            mi_model_free(m1); mi_model_free(m2);
            status = 1;
            continue;
        }
        std::printf("\n#Correspondences[%s, %s]: %lld\n", l.first.c_str(), l.second.c_str(), (long long)r.ncorr);
        print_transform(r);
        const std::string folder = out_dir + "/dendogram/cluster" + l.merged + "/";
        if (mkdir(folder.c_str(), 0777) != 0 && errno != EEXIST) std::printf("Unable to create dir.\n");
        const std::string nvm = folder + "outputVSFM_GB.nvm";
        std::printf("Saving model to %s...\n", nvm.c_str());
        if (mi_model_save_nvm(m, nvm.c_str()) != 0 || mi_model_save_merge_side_files(m, folder.c_str(), (int64_t)m2->npt()) != 0) {
            std::fprintf(stderr, "Merge3D3D: %s\n", mi_model_last_error());
            status = 1;
        } else if (!run_optimizer(folder, "0", "100", "0")) status = 1;   // :902-904
        mi_model_free(m); mi_model_free(m1); mi_model_free(m2);
    }
    return status;
}

static bool read_rts(const std::string &path, double &scale, double R[9], double t[3])
{
    std::ifstream in(path);
    std::string l1, l2, l3;
    if (!in || !std::getline(in, l1) || !std::getline(in, l2) || !std::getline(in, l3)) return false;
    std::istringstream a(l1), b(l2), c(l3);
    if (!(a >> scale)) return false;
    for (int i = 0; i < 9; ++i) if (!(b >> R[i])) return false;
    for (int i = 0; i < 3; ++i) if (!(c >> t[i])) return false;
    return true;
}

int mi_mergeclusters_main(int argc, char **argv)
{
    if (argc < 3) { std::printf("Invalid arguments\n ./a.out <Output DIR> <Index>\n"); return 0; }   // :556-560
    const std::string out_dir = argv[1];
    const int index = std::atoi(argv[2]);
    std::printf("\n\nLoading cluster details ...\n");
    std::vector<MergeLine> lines;
    if (!read_merge_list(out_dir, lines)) return 1;
    if (index < 0 || index >= (int)lines.size()) return 0;                // the reference's loop body does not run
    const MergeLine &l = lines[(size_t)index];                            // one merge per process: the loop ends in a break, :1112
    const std::string nvm1 = out_dir + "/dendogram/cluster" + l.first + "/outputVSFM_GB.nvm", nvm2 = out_dir + "/dendogram/cluster" + l.second + "/outputVSFM_GB.nvm";
    std::printf("%s\n%s\n", nvm1.c_str(), nvm2.c_str());
    mi_model *m1 = mi_model_load_nvm(nvm1.c_str()), *m2 = m1 ? mi_model_load_nvm(nvm2.c_str()) : nullptr;
    if (!m2) { std::fprintf(stderr, "MergeClusters: %s\n", mi_model_last_error()); mi_model_free(m1); return 1; }
    const std::string folder = out_dir + "/dendogram/cluster" + l.merged + "/";
    static const char *const kCandidates[5] = { "RTS2D2D.txt", "RTS3D3D.txt", "RTS2DR3DT.txt", "RTSRefined.txt", "RTSRefined_1.txt" }; // :665-669
    mi_merge_score score[5];
    bool have[5] = { false, false, false, false, false };
    std::vector<std::vector<double>> dist(5);
    int64_t ncorr = 0;
    const size_t nmatch = find_matches(read_tracks(*m1), read_tracks(*m2)).size();
    for (int k = 0; k < 5; ++k) {
        double s, R[9], t[3];
        if (!read_rts(folder + kCandidates[k], s, R, t)) { std::fprintf(stderr, "MergeClusters: no usable %s%s\n", folder.c_str(), kCandidates[k]); continue; }
        dist[k].assign(nmatch, 0.0);
        const std::string dbg = folder + "debug_" + std::to_string(k) + ".txt";
        if (mi_model_score_transform(m1, m2, s, R, t, &score[k], dist[k].data(), dbg.c_str()) != 0) continue;
        have[k] = true; ncorr = score[k].ncorr;
        std::printf("###### %d %lf %lf %lf %lf %d\n", k, score[k].mean_reproj, score[k].median_reproj, score[k].median_dist, score[k].mean_dist, (int)score[k].inliers); // :853
    }
    int best = -1;
    for (int k = 0; k < 5; ++k) if (have[k] && (best < 0 || score[best].mean_dist > score[k].mean_dist)) best = k;   // :875-885
    if (best < 0) { std::fprintf(stderr, "MergeClusters: no candidate transform for cluster %s\n", l.merged.c_str()); mi_model_free(m1); mi_model_free(m2); return 1; }
    std::printf("\n#Correspondences[%s, %s]: %lld\n", l.first.c_str(), l.second.c_str(), (long long)ncorr);
    if (FILE *fe = std::fopen((folder + "debug_err.txt").c_str(), "w")) {   // :863-873: the first four candidates
        for (int64_t q = 0; q < ncorr; ++q) {
            for (int k = 0; k < 4; ++k) std::fprintf(fe, "%lf ", have[k] ? dist[k][(size_t)q] : 0.0);
            std::fprintf(fe, "\n");
        }
        std::fclose(fe);
    }
    double s, R[9], t[3];
    read_rts(folder + kCandidates[best], s, R, t);
    mi_merge_result r;
    mi_model *m = mi_model_merge_with_transform(m1, m2, s, R, t, 1, 0, &r);
    int status = 0;
    if (!m) { std::fprintf(stderr, "MergeClusters: %s\n", mi_model_last_error()); status = 1; }
    else {
        print_transform(r);
        std::printf("Pt 1 %zu PT 2: %zu\n", m1->npt(), m2->npt());
        if (mkdir(folder.c_str(), 0777) != 0) std::printf("Unable to create dir.\n");   // it exists (the candidates are in it): the reference prints this too
        const std::string nvm = folder + "outputVSFM_GB.nvm";
        std::printf("Saving model to %s...\n", nvm.c_str());
        if (mi_model_save_nvm(m, nvm.c_str()) != 0 || mi_model_save_merge_side_files(m, folder.c_str(), 0) != 0) {
            std::fprintf(stderr, "MergeClusters: %s\n", mi_model_last_error());
            status = 1;
        } else if (!(score[best].mean_reproj > 10.0 ? run_optimizer(folder, "0", "50", "0.5") : run_optimizer(folder, "-1", "5", "0.5"))) status = 1; // :1102-1108
    }
    mi_model_free(m); mi_model_free(m1); mi_model_free(m2);
    return status;
}

} // extern "C"

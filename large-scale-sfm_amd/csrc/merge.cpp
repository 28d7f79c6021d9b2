// merge.cpp -- the cluster-merge step around the GPU similarity alignment (include/mi_merge.h): correspondences by point id,
// the second model brought into the first one's frame, tracks fused, side files, and the Merge3D3D process.
// Host code: the lists are touched once each; the O(n^2) + 5000-round alignment is mi_sim3.h's kernels.
#include <algorithm>
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
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
    const size_t n1 = m1.npt(), n2 = m2.npt(), c1 = m1.ncam(), c2 = m2.ncam();
    const TrackView t1 = read_tracks(m1), t2 = read_tracks(m2);

    // correspondences, :646-683: the first point of the second model with the same id
    std::unordered_map<long long, int> first_with_id;
    first_with_id.reserve(n2 * 2);
    for (size_t i = 0; i < n2; ++i) first_with_id.emplace(t2.id[i], (int)i); // emplace keeps the earlier entry
    std::vector<std::pair<int, int>> match;
    for (size_t i = 0; i < n1; ++i) {
        auto it = first_with_id.find(t1.id[i]);
        if (it != first_with_id.end()) match.emplace_back((int)i, it->second);
    }
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
    double s = 1.0, R[9], T[3], ms = 0.0;
    int best = -1;
    if (mi_sim3_align(&sc, (int)nc, src.data(), dst.data(), &s, R, T, nullptr, &best, &ms) != 0) {
        miba::set_model_error(std::string("merge: ") + mi_sim3_last_error());
        return nullptr;
    }
    if (res) {
        res->ncorr = (int64_t)nc; res->scale = s; std::memcpy(res->R, R, sizeof R); std::memcpy(res->t, T, sizeof T);
        res->best_round = best; res->device_ms = ms; res->ncam_first = (int64_t)c1; res->npt_first = (int64_t)n1;
    }
    auto move = [&](const double *x, double *y) { // y = R (s x) + T
        const double p0 = s * x[0], p1 = s * x[1], p2 = s * x[2];
        for (int r = 0; r < 3; ++r) y[r] = R[3 * r] * p0 + R[3 * r + 1] * p1 + R[3 * r + 2] * p2 + T[r];
    };

    mi_model *out = new mi_model;
    mi_model &m = *out;
    // cameras: the first model's, then the second's in the first one's frame (:739-772, :784)
    m.names = m1.names; m.names.insert(m.names.end(), m2.names.begin(), m2.names.end());
    m.f = m1.f; m.f.insert(m.f.end(), m2.f.begin(), m2.f.end());
    m.d = m1.d; m.d.insert(m.d.end(), m2.d.begin(), m2.d.end());
    m.R = m1.R; m.t = m1.t;
    m.R.resize((c1 + c2) * 9); m.t.resize((c1 + c2) * 3);
    for (size_t j = 0; j < c2; ++j) {
        const double *Rj = &m2.R[j * 9], *tj = &m2.t[j * 3];
        double *Rn = &m.R[(c1 + j) * 9], *tn = &m.t[(c1 + j) * 3], c[3], cn[3];
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) Rn[3 * a + b] = Rj[3 * a] * R[3 * b] + Rj[3 * a + 1] * R[3 * b + 1] + Rj[3 * a + 2] * R[3 * b + 2]; // Rj R^T
        centre_of(Rj, tj, c);
        move(c, cn);
        for (int a = 0; a < 3; ++a) tn[a] = -(Rn[3 * a] * cn[0] + Rn[3 * a + 1] * cn[1] + Rn[3 * a + 2] * cn[2]);  // getTfromRC, :133-137
    }
    // points: the first model's, then every point of the second (:797-798)
    m.xyz = m1.xyz; m.xyz.resize((n1 + n2) * 3);
    for (size_t i = 0; i < n2; ++i) move(&m2.xyz[3 * i], &m.xyz[3 * (n1 + i)]);
    m.rgb.resize((n1 + n2) * 3);
    for (size_t i = 0; i < n1 + n2; ++i) {
        const bool a = i < n1;
        const int *from = a ? &m1.rgb[3 * i] : &m2.rgb[3 * (i - n1)];
        const int paint[3] = { a ? 255 : 0, a ? 0 : 255, 0 };           // ReadNVM colour modes 1 and 3, :487-501
        for (int r = 0; r < 3; ++r) m.rgb[3 * i + r] = cfg.keep_colours ? from[r] : paint[r];
    }
    // which tracks of the second model are appended to a point of the first (:788-795)
    std::vector<int> partner(n1, -1);
    for (const auto &pr : match) {
        partner[pr.first] = pr.second;
        if (!cfg.keep_colours) {
            int *a = &m.rgb[3 * (size_t)pr.first], *b = &m.rgb[3 * (n1 + (size_t)pr.second)];
            a[0] = 0; a[1] = 0; a[2] = 255; b[0] = b[1] = b[2] = 255;    // :662-668
        }
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
    for (size_t i = 0; i < n2; ++i) put_views(m2, t2, i, (int)c1, n1 + i, t2.id[i]);
    return out;
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

int mi_merge3d3d_main(int argc, char **argv)
{
    if (argc < 2) { std::printf("Invalid arguments\n ./a.out <Output DIR>\n"); return 0; }   // :567-571
    const std::string out_dir = argv[1];
    mi_merge_config cfg;
    mi_merge_default_config(&cfg);
    if (const char *dev = std::getenv("MI_BA_DEVICE")) cfg.device = std::atoi(dev);
    std::string optimizer = dirname_of_self() + "/optimizer_all";
    if (const char *o = std::getenv("MI_OPTIMIZER_ALL")) optimizer = o;
    std::printf("\n\nLoading cluster details ...\n");
    std::ifstream list(out_dir + "/dendogram/fullclustermergeinfo.txt");
    if (!list) { std::fprintf(stderr, "Merge3D3D: cannot open %s/dendogram/fullclustermergeinfo.txt\n", out_dir.c_str()); return 1; }
    int status = 0;
    std::string line;
    while (std::getline(list, line)) {
        std::istringstream toks(line);
        std::string merged_name, a, b;                                   // "new first second", :590-593
        if (!(toks >> merged_name >> a >> b)) continue;
        const std::string nvm1 = out_dir + "/dendogram/cluster" + a + "/outputVSFM_GB.nvm", nvm2 = out_dir + "/dendogram/cluster" + b + "/outputVSFM_GB.nvm";
        std::printf("%s\n%s\n", nvm1.c_str(), nvm2.c_str());
        mi_model *m1 = mi_model_load_nvm(nvm1.c_str()), *m2 = m1 ? mi_model_load_nvm(nvm2.c_str()) : nullptr;
        mi_merge_result r;
        mi_model *m = m2 ? mi_model_merge_3d3d(m1, m2, &cfg, &r) : nullptr;
        if (!m) {
            std::fprintf(stderr, "Merge3D3D: clusters %s + %s: %s\n", a.c_str(), b.c_str(), mi_model_last_error());
            mi_model_free(m1); mi_model_free(m2);
            status = 1;
            continue;
        }
        std::printf("\n#Correspondences[%s, %s]: %lld\n", a.c_str(), b.c_str(), (long long)r.ncorr);
        std::printf("Scale:%g\nR:\n%g %g %g\n%g %g %g\n%g %g %g\nT:\n%g  %g  %g\n", r.scale, r.R[0], r.R[1], r.R[2], r.R[3], r.R[4], r.R[5],
                    r.R[6], r.R[7], r.R[8], r.t[0], r.t[1], r.t[2]);
        const std::string folder = out_dir + "/dendogram/cluster" + merged_name + "/";
        if (mkdir(folder.c_str(), 0777) != 0 && errno != EEXIST) std::printf("Unable to create dir.\n");
        const std::string nvm = folder + "outputVSFM_GB.nvm";
        std::printf("Saving model to %s...\n", nvm.c_str());
        const int64_t n2 = (int64_t)m2->npt();
        if (mi_model_save_nvm(m, nvm.c_str()) != 0 || mi_model_save_merge_side_files(m, folder.c_str(), n2) != 0) {
            std::fprintf(stderr, "Merge3D3D: %s\n", mi_model_last_error());
            status = 1;
        } else {
            // "Run bundler", :902-904 -- a child process, as there; its exit status is not looked at there either
            std::fflush(stdout); std::fflush(stderr);
            const std::string dir_arg = folder + "/";
            const char *args[] = { optimizer.c_str(), dir_arg.c_str(), "0", "100", "16", "32", "0", "0", nullptr };
            pid_t pid = 0;
            if (posix_spawn(&pid, optimizer.c_str(), nullptr, nullptr, const_cast<char *const *>(args), environ) != 0) {
                std::fprintf(stderr, "Merge3D3D: cannot start %s\n", optimizer.c_str());
                status = 1;
            } else {
                int st = 0;
                waitpid(pid, &st, 0);
                if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) status = 1;
            }
        }
        mi_model_free(m); mi_model_free(m1); mi_model_free(m2);
    }
    return status;
}

} // extern "C"

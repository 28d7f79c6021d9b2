// plan.cpp -- see plan.h.  Pure host code (no HIP); exported through mi_ba_plan_* for CPU tests.
#include "plan.h"

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <functional>
#include <numeric>
#include <thread>
#include <unordered_map>

namespace miba {

int Plan::upper_block(int j, int k) const
{
    if (j > k) std::swap(j, k);
    const int *b = up_col.data() + up_row[j], *e = up_col.data() + up_row[j + 1];
    const int *it = std::lower_bound(b, e, k);
    return (it != e && *it == k) ? (int)(it - up_col.data()) : -1;
}

namespace {

inline int point_class(int d)
{
    if (d == 0) return 3;
    if (d <= kMaxViewsLds) return 0;
    if (d <= kWave) return 1;
    return 2;
}

inline double point_weight(int d) { return (double)d + 0.5 * (double)d * (d + 1); }

} // namespace

// std::sort on a few host threads: the ranges sorted on their own, then merged pairwise (the result is std::sort's: the keys
// are distinct -- every element carries its index)
template <class T>
static void parallel_sort(std::vector<T> &v)
{
    const size_t n = v.size();
    unsigned nt = std::min<unsigned>(8u, std::max(1u, std::thread::hardware_concurrency()));
    while (nt & (nt - 1)) nt &= nt - 1;                     // a power of two
    if (nt <= 1 || n < (1u << 16)) { std::sort(v.begin(), v.end()); return; }
    std::vector<size_t> cut(nt + 1);
    for (unsigned k = 0; k <= nt; ++k) cut[k] = n * k / nt;
    {
        std::vector<std::thread> th;
        for (unsigned k = 0; k < nt; ++k) th.emplace_back([&, k]() { std::sort(v.begin() + cut[k], v.begin() + cut[k + 1]); });
        for (auto &t : th) t.join();
    }
    for (unsigned w = 1; w < nt; w *= 2) {
        std::vector<std::thread> th;
        for (unsigned k = 0; k + w < nt + 1 && k < nt; k += 2 * w)
            th.emplace_back([&, k, w]() { std::inplace_merge(v.begin() + cut[k], v.begin() + cut[k + w], v.begin() + cut[std::min(k + 2 * w, nt)]); });
        for (auto &t : th) t.join();
    }
}

// fn(begin, end) over [0, n) on a few host threads (the loops below touch every observation once and are independent per point)
template <class F>
static void parallel_ranges(int n, F fn)
{
    const int nt = (int)std::min<unsigned>(8u, std::max(1u, std::thread::hardware_concurrency()));
    if (nt <= 1 || n < (1 << 16)) { fn(0, n); return; }
    std::vector<std::thread> th;
    for (int k = 0; k < nt; ++k) {
        const int b = (int)((int64_t)n * k / nt), e = (int)((int64_t)n * (k + 1) / nt);
        th.emplace_back([=]() { fn(b, e); });
    }
    for (auto &t : th) t.join();
}

std::string build_plan(Plan &pl, int n, int m, int64_t nobs, const int *pt_idx, const int *cam_idx,
                       int rank, int world, int tile_max_batches, const std::vector<int> *extra_pairs, bool all_rows, bool cam_split,
                       const std::function<void()> &shard_arrays_ready)
{
    if (cam_split) all_rows = true;    // nine-parameter blocks exist on the row path only
    if (n <= 0 || m <= 0 || nobs < 0) return "empty problem";
    if (world < 1 || rank < 0 || rank >= world) return "bad rank/world";
    if ((int64_t)m * m / 8 > (int64_t)1 << 30) return "more than ~92k cameras is not supported yet";
    if (nobs >= ((int64_t)1 << 31)) return "more than 2^31 observations is not supported yet";
    const bool auto_tile = tile_max_batches <= 0;
    if (auto_tile || tile_max_batches > kMaxTileBatches) tile_max_batches = kMaxTileBatches;
    pl = Plan();
    pl.n = n; pl.m = m; pl.mv = m; pl.nobs = nobs; pl.rank = rank; pl.world = world;
    // MI_BA_PLAN_TIMING=1: where the planner's time goes, stage by stage, to stderr
    const bool timing = std::getenv("MI_BA_PLAN_TIMING") != nullptr;
    auto t_prev = std::chrono::steady_clock::now();
    auto stage = [&](const char *what) {
        if (!timing) return;
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[mi_ba plan] %-28s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t_prev).count());
        t_prev = now;
    };

    // ---- CSR by point, cameras ascending inside a point
    std::vector<int> pt_ptr(n + 1, 0);
    for (int64_t o = 0; o < nobs; ++o) {
        int i = pt_idx[o], j = cam_idx[o];
        if (i < 0 || i >= n) return "point index out of range";
        if (j < 0 || j >= m) return "camera index out of range";
        if (o && pt_idx[o - 1] > i) return "observations must be sorted by point (pt_idx non-decreasing)";
        pt_ptr[i + 1]++;
    }
    for (int i = 0; i < n; ++i) pt_ptr[i + 1] += pt_ptr[i];
    uvector<int> cam(nobs);            // (every element is written by the threads below)
    uvector<int64_t> src(nobs);
    {
        std::atomic<bool> twice(false);
        parallel_ranges(n, [&](int i0, int i1) {
            std::vector<std::pair<int, int64_t>> tmp;
            for (int i = i0; i < i1; ++i) {
                int b = pt_ptr[i], e = pt_ptr[i + 1];
                bool sorted = true;
                for (int o = b + 1; o < e; ++o) if (cam_idx[o - 1] >= cam_idx[o]) { sorted = false; break; }
                if (sorted) {
                    for (int o = b; o < e; ++o) { cam[o] = cam_idx[o]; src[o] = o; }
                } else {
                    tmp.clear();
                    for (int o = b; o < e; ++o) tmp.emplace_back(cam_idx[o], (int64_t)o);
                    std::sort(tmp.begin(), tmp.end());
                    for (int o = b; o < e; ++o) { cam[o] = tmp[o - b].first; src[o] = tmp[o - b].second; }
                    for (int o = b + 1; o < e; ++o)
                        if (cam[o] == cam[o - 1]) twice = true;
                }
            }
        });
        if (twice) return "a point is observed twice by the same camera";
    }

    stage("CSR by point");
    // ---- permutation: class, then visibility signature (lexicographic camera list)
    pl.perm.resize(n);
    std::iota(pl.perm.begin(), pl.perm.end(), 0);
    auto deg = [&](int i) { return pt_ptr[i + 1] - pt_ptr[i]; };
    std::vector<uint8_t> cls(n);
    for (int i = 0; i < n; ++i) { cls[i] = (uint8_t)point_class(deg(i)); if (all_rows && cls[i] == 0) cls[i] = 1; }
    auto by_class_then_signature = [&](int a, int b) {
        int ca = cls[a], cb = cls[b];
        if (ca != cb) return ca < cb;
        if (ca == 0 || ca == 1) {
            const int *pa = cam.data() + pt_ptr[a], *pb = cam.data() + pt_ptr[b];
            int da = deg(a), db = deg(b), dm = std::min(da, db);
            for (int t = 0; t < dm; ++t) if (pa[t] != pb[t]) return pa[t] < pb[t];
            if (da != db) return da < db;
        }
        return a < b;
    };
    // The benchmark scenes hold a few thousand distinct signatures, each shared by hundreds of points, and a comparison of
    // two equal camera lists is the expensive kind (all of it, then the index).  So: group the points by a hash of (class,
    // camera list), sort one representative per group with the comparator, and emit the groups' members in index order --
    // the same permutation as sorting every point, at a fraction of the comparisons.  A hash collision (two different lists
    // in one group) falls back to the plain sort.
    std::vector<int> sig_rep;      // one point per distinct (class, camera list) of class 0 / 1, when the hash grouping held (else empty)
    {
        std::vector<std::pair<uint64_t, int>> hashed(n);
        parallel_ranges(n, [&](int b, int e) {
            for (int i = b; i < e; ++i) {
                uint64_t h = 0x9E3779B97F4A7C15ull * (uint64_t)(cls[i] + 1);
                if (cls[i] == 0 || cls[i] == 1) {
                    const int *p = cam.data() + pt_ptr[i];
                    for (int t = 0, d = deg(i); t < d; ++t) { h ^= (uint64_t)(p[t] + 1); h *= 0x100000001B3ull; h ^= h >> 29; }
                }
                hashed[i] = { h, i };
            }
        });
        stage("  sig: hash");
        parallel_sort(hashed);
        stage("  sig: sort");
        // groups: a new one starts where the hash changes; inside a group every member must have its predecessor's class and camera list
        // (equality is transitive) -- one random access per point, spread over the threads
        std::vector<uint8_t> starts(n, 0);
        std::atomic<int> collided{ 0 };
        parallel_ranges(n, [&](int b, int e) {
            for (int q = b; q < e; ++q) {
                if (q == 0 || hashed[q].first != hashed[q - 1].first) { starts[q] = 1; continue; }
                const int a = hashed[q - 1].second, c = hashed[q].second;
                bool bad = cls[a] != cls[c];
                if (!bad && (cls[a] == 0 || cls[a] == 1))
                    bad = deg(a) != deg(c) || std::memcmp(cam.data() + pt_ptr[a], cam.data() + pt_ptr[c], sizeof(int) * (size_t)deg(a)) != 0;
                if (bad) collided.store(1, std::memory_order_relaxed);
            }
        });
        const bool collision = collided.load() != 0;
        std::vector<int> rep, first;                   // per group: representative point, start in hashed[]
        if (!collision) {
            for (int q = 0; q < n; ++q) if (starts[q]) { rep.push_back(hashed[q].second); first.push_back(q); }
            for (int i : rep) if (cls[i] == 0 || cls[i] == 1) sig_rep.push_back(i);
        }
        stage("  sig: group scan");
        std::vector<int> run_end;                      // ends (in perm) of the runs of one camera set, in order
        if (collision) std::sort(pl.perm.begin(), pl.perm.end(), by_class_then_signature);
        else {
            first.push_back(n);
            std::vector<int> order(rep.size());
            std::iota(order.begin(), order.end(), 0);
            std::sort(order.begin(), order.end(), [&](int x, int y) { return by_class_then_signature(rep[x], rep[y]); });
            int w = 0;
            run_end.reserve(order.size());
            for (int g : order) {
                for (int q = first[g]; q < first[g + 1]; ++q) pl.perm[w++] = hashed[q].second;   // index order inside a group
                run_end.push_back(w);
            }
        }
        stage("  sig: emit");
    // a camera set has to repeat to earn a register-accumulated tile: short runs take the row path (plan.h kMinRunPts).
    // perm is sorted by (class, signature); moving a subset of class 0 to class 1 keeps both parts in signature order,
    // so the new order is a stable partition plus one merge of the two class-1 pieces
        int min_run = kMinRunPts;
        if (const char *ev = std::getenv("MI_BA_MIN_RUN_PTS")) min_run = std::atoi(ev);
        bool changed = false;
        int n0 = 0;
        while (n0 < n && cls[pl.perm[n0]] == 0) ++n0;
        if (!collision) {          // the runs are the groups just emitted (class 0 comes first)
            int t0 = 0;
            for (int t1 : run_end) {
                if (t0 >= n0) break;
                if (t1 - t0 < min_run) { for (int t = t0; t < t1; ++t) cls[pl.perm[t]] = 1; changed = true; }
                t0 = t1;
            }
        } else {
            int t0 = 0;
            while (t0 < n0) {
                const int i0 = pl.perm[t0], d = deg(i0);
                int t1 = t0 + 1;
                while (t1 < n0 && deg(pl.perm[t1]) == d && std::memcmp(cam.data() + pt_ptr[i0], cam.data() + pt_ptr[pl.perm[t1]], sizeof(int) * d) == 0) ++t1;
                if (t1 - t0 < min_run) { for (int t = t0; t < t1; ++t) cls[pl.perm[t]] = 1; changed = true; }
                t0 = t1;
            }
        }
        if (changed) {
            int n1 = n0;
            while (n1 < n && cls[pl.perm[n1]] == 1) ++n1;                       // old class 1: [n0, n1)
            auto mid = std::stable_partition(pl.perm.begin(), pl.perm.begin() + n0, [&](int i) { return cls[i] == 0; });
            std::inplace_merge(mid, pl.perm.begin() + n0, pl.perm.begin() + n1, by_class_then_signature);
        }
    }
    pl.n_active = 0;
    for (int i = 0; i < n; ++i) if (deg(i) > 0) pl.n_active++;

    stage("signature sort");
    // ---- shard cut (contiguous internal ranges, balanced by observations + pairs)
    if (world == 1) { pl.pt_begin = 0; pl.pt_end = pl.n_active; }
    else {
        double total = 0;
        for (int t = 0; t < pl.n_active; ++t) total += point_weight(deg(pl.perm[t]));
        double lo = total * rank / world, hi = total * (rank + 1) / world, acc = 0;
        int b = pl.n_active, e = pl.n_active;
        bool have_b = false;
        for (int t = 0; t < pl.n_active; ++t) {
            if (!have_b && acc >= lo) { b = t; have_b = true; }
            acc += point_weight(deg(pl.perm[t]));
            if (acc >= hi && rank != world - 1) { e = t + 1; break; }
        }
        if (!have_b) b = pl.n_active;
        if (rank == 0) b = 0;
        if (rank == world - 1) e = pl.n_active;
        if (e < b) e = b;
        pl.pt_begin = b; pl.pt_end = e;
    }

    // ---- block pattern of S from ALL points (every rank must agree on it)
    {
        const size_t words = ((size_t)m + 63) / 64;
        std::vector<uint64_t> bits(words * (size_t)m, 0);
        for (int j = 0; j < m; ++j) bits[(size_t)j * words + (j >> 6)] |= 1ull << (j & 63);
        auto mark = [&](int i) {
            const int d = deg(i);
            const int *c = cam.data() + pt_ptr[i];
            for (int a = 0; a < d; ++a) {
                uint64_t *row = bits.data() + (size_t)c[a] * words;
                for (int b = a + 1; b < d; ++b) row[c[b] >> 6] |= 1ull << (c[b] & 63);
            }
        };
        if (!sig_rep.empty() || pl.n_active == 0) {        // one point per camera set, and the long points one by one (they were grouped by class only)
            for (int i : sig_rep) mark(i);
            for (int t = pl.n_active - 1; t >= 0 && cls[pl.perm[t]] == 2; --t) mark(pl.perm[t]);
        } else {
            const int *prev = nullptr; int prevd = -1;
            for (int t = 0; t < pl.n_active; ++t) {
                int i = pl.perm[t], d = deg(i);
                const int *c = cam.data() + pt_ptr[i];
                if (d == prevd && prev && std::memcmp(prev, c, sizeof(int) * d) == 0) continue; // same signature
                prev = c; prevd = d;
                mark(i);
            }
        }
        if (extra_pairs)
            for (size_t q = 0; q + 1 < extra_pairs->size(); q += 2) {
                const int a = std::min((*extra_pairs)[q], (*extra_pairs)[q + 1]), b = std::max((*extra_pairs)[q], (*extra_pairs)[q + 1]);
                if (a < 0 || b >= m) return "camera link out of range";
                bits[(size_t)a * words + (b >> 6)] |= 1ull << (b & 63);
            }
        pl.up_row.assign(m + 1, 0);
        for (int j = 0; j < m; ++j) {
            int cnt = 0;
            const uint64_t *row = bits.data() + (size_t)j * words;
            for (size_t w = 0; w < words; ++w) cnt += __builtin_popcountll(row[w]);
            pl.up_row[j + 1] = pl.up_row[j] + cnt;
        }
        pl.up_col.resize(pl.up_row[m]);
        for (int j = 0; j < m; ++j) {
            int *out = pl.up_col.data() + pl.up_row[j];
            const uint64_t *row = bits.data() + (size_t)j * words;
            for (size_t w = 0; w < words; ++w) {
                uint64_t v = row[w];
                while (v) { int b = __builtin_ctzll(v); *out++ = (int)(w * 64 + b); v &= v - 1; }
            }
        }
        // full pattern: row j = {k < j : (k,j) upper} U {k >= j : (j,k) upper}, columns ascending
        std::vector<int> lowcnt(m, 0);
        for (int j = 0; j < m; ++j)
            for (int q = pl.up_row[j]; q < pl.up_row[j + 1]; ++q) if (pl.up_col[q] != j) lowcnt[pl.up_col[q]]++;
        pl.bsr_row.assign(m + 1, 0);
        for (int j = 0; j < m; ++j) pl.bsr_row[j + 1] = pl.bsr_row[j] + lowcnt[j] + (pl.up_row[j + 1] - pl.up_row[j]);
        int nbf = pl.bsr_row[m];
        pl.bsr_col.resize(nbf); pl.bsr_src.resize(nbf); pl.bsr_trans.resize(nbf);
        std::vector<int> fill(m);
        for (int j = 0; j < m; ++j) fill[j] = pl.bsr_row[j];
        for (int j = 0; j < m; ++j)            // lower parts first (rows visited ascending => sorted)
            for (int q = pl.up_row[j]; q < pl.up_row[j + 1]; ++q) {
                int k = pl.up_col[q];
                if (k == j) continue;
                int pos = fill[k]++;
                pl.bsr_col[pos] = j; pl.bsr_src[pos] = q; pl.bsr_trans[pos] = 1;
            }
        for (int j = 0; j < m; ++j)
            for (int q = pl.up_row[j]; q < pl.up_row[j + 1]; ++q) {
                int pos = fill[j]++;
                pl.bsr_col[pos] = pl.up_col[q]; pl.bsr_src[pos] = q; pl.bsr_trans[pos] = 0;
            }
    }

    stage("shard cut + pattern of S");
    // ---- local shard arrays
    const int npl = pl.pt_end - pl.pt_begin;
    pl.pt_obs.assign(npl + 1, 0);
    for (int t = 0; t < npl; ++t) pl.pt_obs[t + 1] = pl.pt_obs[t] + deg(pl.perm[pl.pt_begin + t]);
    const int nol = pl.pt_obs[npl];
    pl.obs_cam.resize(nol); pl.obs_pt.resize(nol); pl.obs_src.resize(nol); pl.obs_cslot.resize(nol);   // (obs_cslot: zeroed by every tiling pass)
    parallel_ranges(npl, [&](int t0, int t1) {
        for (int t = t0; t < t1; ++t) {
            int i = pl.perm[pl.pt_begin + t], d = deg(i);
            for (int a = 0; a < d; ++a) {
                int o = pl.pt_obs[t] + a;
                pl.obs_cam[o] = cam[pt_ptr[i] + a];
                pl.obs_src[o] = src[pt_ptr[i] + a];
                pl.obs_pt[o] = t;
            }
        }
    });

    if (shard_arrays_ready) shard_arrays_ready();
    // ---- batches and tiles (as a lambda: the automatic tile length needs a first pass to see where the
    // slot / camera budgets force tiles to end)
    std::vector<std::pair<int, bool>> tile_log; // per LDS tile: batches, closed only because of the length cap
    int t = 0, t_rows = 0;            // t_rows: first local point of the row path
    std::string tiling_error;
    // dry: only the shape of the LDS tiles (tile_log) is wanted -- no pair entries, no slots per observation, class 0 only
    // opt-in (MI_BA_MFMA=1): measured SLOWER than the vector form of the same tiles (C3 build 1.15 ms against 0.82: the f64 matrix
    // pipe has the vector pipe's rate, and 8 x 8-padded blocks with K padded to 4 are 1.37x the vector form's arithmetic)
    const char *mf = std::getenv("MI_BA_MFMA");
    const bool use_mfma = !cam_split && mf && mf[0] == '1';
    // uniform tiles (mode 4, k_build_uni) are the default for a tile of one camera set; MI_BA_UNI=0 keeps the general kernel (probes, parity tests)
    const char *un = std::getenv("MI_BA_UNI");
    const bool use_uni = !cam_split && !use_mfma && !(un && un[0] == '0');
    static_assert(kUniMinViews == kMfmaMinViews && kUniMinBatches == kMfmaMinBatches, "modes 3 and 4 share the tiling rule");
    const bool one_set_tiles = use_mfma || use_uni;
    auto do_tiling = [&](int tile_max_batches, bool dry) {
    tile_log.clear();
    pl.pair.clear(); pl.slot_gblk.clear(); pl.cslot_cam.clear(); pl.tile_mode.clear();
    pl.long_pts.clear(); pl.max_slots = 0;
    std::fill(pl.obs_cslot.begin(), pl.obs_cslot.end(), 0);
    pl.batch_obs.assign(1, 0); pl.batch_pt.assign(1, 0); pl.batch_pair.clear();
    int64_t batch_q0 = 0, prev_q0 = -1, prev_n = 0;   // this batch's first pair entry; the previous batch of the tile: offset, entries
    pl.tile_batch.assign(1, 0); pl.tile_slot.assign(1, 0); pl.tile_cslot.assign(1, 0);
    std::unordered_map<uint64_t, int> smap;
    std::unordered_map<int, int> cmap;
    std::vector<int> sig_slots;       // slots of the previous point's pairs (signature cache)
    std::vector<uint8_t> sig_cslots;  // camera slots of the previous point's observations (signature cache)
    size_t round_fill = 0;            // entries in the round being filled (= round_buf.size() unless dry)
    const int *sig_cams = nullptr; int sig_d = -1;
    int cur_obs = 0, batches_in_tile = 0, sig_runs_in_tile = 0;
    t = 0;
    pl.ntile_table = 0; pl.ntile_mfma = 0; pl.ntile_uni = 0; pl.ntile_uni_fixed = 0;

    std::vector<uint32_t> round_buf;  // pair entries of the round being filled (LDS tiles)
    bool rounds_mode = true;
    int pts_in_batch = 0;
    int rounds_in_batch = 0;
    auto flush_round = [&]() {          // (counts only: the entries of a batch are written when it closes, if they are needed at all)
        if (round_fill == 0) return;
        rounds_in_batch++;
        round_fill = 0;
    };
    // The points of the batch being filled: the version of the slot list they use (one per signature run of the tile), views, first
    // lane, local index.  Two consecutive batches of a tile with the same sequence have the same pair entries -- the usual case inside
    // a run of one camera set --, so the entries (d (d + 1) / 2 per point: 55 M pushes on C3) are only generated for a batch whose
    // sequence is new.
    struct BatchPoint { int ver, d, obs0, lp; };
    std::vector<BatchPoint> bpts, prev_bpts;
    std::vector<std::vector<int>> ver_slots;     // slot lists of this tile's signature runs
    auto emit_round = [&]() {
        if (round_buf.empty()) return;
        round_buf.resize(kWave, kPairInvalid);
        pl.pair.insert(pl.pair.end(), round_buf.begin(), round_buf.end());
        round_buf.clear();
    };
    auto close_batch = [&](int t_end) {
        if (cur_obs == 0) return;
        if (rounds_mode) flush_round();
        pl.batch_obs.push_back(pl.pt_obs[t_end]);
        pl.batch_pt.push_back(t_end);
        // Pair TEMPLATES: points of one camera set produce the same entries batch after batch (lane-local indices, slots, local
        // points), so a batch whose rounds equal the previous batch's of its tile points at those instead of storing its own --
        // C3: 64 M entries (256 MB streamed per linearisation) -> one or two templates per tile, read from L2
        int64_t off = batch_q0, n_ent = 0;
        if (!dry && rounds_mode) {
            bool same = prev_q0 >= 0 && bpts.size() == prev_bpts.size();
            for (size_t k = 0; same && k < bpts.size(); ++k)
                same = bpts[k].ver == prev_bpts[k].ver && bpts[k].d == prev_bpts[k].d && bpts[k].obs0 == prev_bpts[k].obs0 && bpts[k].lp == prev_bpts[k].lp;
            if (same) { off = prev_q0; n_ent = prev_n; }
            else {
                size_t fill = 0;
                for (const BatchPoint &bp : bpts) {       // the rounds as the point loop counted them
                    const size_t np = (size_t)bp.d * (bp.d + 1) / 2;
                    if (fill + np > (size_t)kWave) { emit_round(); fill = 0; }
                    fill += np;
                    const std::vector<int> &slots = ver_slots[bp.ver];
                    int q = 0;
                    for (int a = 0; a < bp.d; ++a)
                        for (int b = a; b < bp.d; ++b)
                            round_buf.push_back((uint32_t)(bp.obs0 + a) | ((uint32_t)(bp.obs0 + b) << 6) | ((uint32_t)slots[q++] << 12) | ((uint32_t)bp.lp << 18));
                }
                emit_round();
                n_ent = (int64_t)pl.pair.size() - batch_q0;
                // (a run that comes back to an earlier camera set of the tile gets a new version with the old slots: equal entries)
                if (n_ent > 0 && prev_q0 >= 0 && n_ent == prev_n &&
                    std::memcmp(pl.pair.data() + prev_q0, pl.pair.data() + batch_q0, sizeof(uint32_t) * (size_t)n_ent) == 0) {
                    pl.pair.resize((size_t)batch_q0);
                    off = prev_q0;
                } else { prev_q0 = batch_q0; prev_n = n_ent; }
                prev_bpts = bpts;
            }
            bpts.clear();
        }
        pl.batch_pair.push_back(off | ((n_ent / kWave) << 48));
        batch_q0 = (int64_t)pl.pair.size();
        cur_obs = 0; pts_in_batch = 0; rounds_in_batch = 0;
        batches_in_tile++;
    };
    auto close_tile = [&](uint8_t mode) {
        if (batches_in_tile == 0) return;
        pl.tile_batch.push_back((int)pl.batch_obs.size() - 1);
        pl.tile_slot.push_back((int)pl.slot_gblk.size());
        pl.tile_cslot.push_back((int)pl.cslot_cam.size());
        // many camera sets in one tile: a lane would hand its register block to HBM almost every round (the L2's f64
        // atomic rate is the bound then) -> the tile sums its <= kNSlot blocks in LDS instead
        uint8_t m2 = (mode == 1 && sig_runs_in_tile > kTableRuns) ? 2 : mode;
        // one camera set of >= kMfmaMinViews views for the whole tile: lane = (point, camera) in every batch, the slots are the
        // a <= b enumeration of its cameras -- the tile's blocks go through the matrix pipe (k_build_mfma)
        // (k_build_mfma, opt-in) or stay on the vector pipe with everything that is constant over such a tile taken out of the loops (k_build_uni)
        if (mode == 1 && sig_runs_in_tile == 1 && sig_d >= kMfmaMinViews && one_set_tiles) m2 = use_mfma ? 3 : 4;
        pl.tile_mode.push_back(m2);
        if (m2 == 2) pl.ntile_table++;
        if (m2 == 3) pl.ntile_mfma++;
        if (m2 == 4) { pl.ntile_uni++; if (sig_d == kUniFixedViews) pl.ntile_uni_fixed++; }
        sig_runs_in_tile = 0;
        if (mode == 1) tile_log.emplace_back(batches_in_tile, batches_in_tile >= tile_max_batches);
        pl.max_slots = std::max(pl.max_slots, (int)smap.size());
        smap.clear(); cmap.clear(); sig_cams = nullptr; sig_d = -1;
        batches_in_tile = 0; prev_q0 = -1; prev_n = 0;
        ver_slots.clear(); prev_bpts.clear();
    };

    // class 0: LDS-accumulated tiles
    for (; t < npl; ++t) {
        int d = pl.pt_obs[t + 1] - pl.pt_obs[t];
        if (cls[pl.perm[pl.pt_begin + t]] != 0) break;
        const int *c = pl.obs_cam.data() + pl.pt_obs[t];
        bool same_sig = (d == sig_d && sig_cams && std::memcmp(sig_cams, c, sizeof(int) * d) == 0);
        // a point that would open a 9th pair round starts a new batch
        const bool round_full = (int)round_fill + d * (d + 1) / 2 > kWave;
        if (cur_obs + d > kWave || pts_in_batch >= kMaxPtsBatch || (round_full && rounds_in_batch + 1 >= kMaxRounds)) {
            close_batch(t);
            if (batches_in_tile >= tile_max_batches) { close_tile(1); same_sig = false; }
        }
        if (!same_sig) {
            // a tile that so far holds ONE camera set of a long run ends here, so that it is a uniform tile (mode 4; mode 3 with MI_BA_MFMA)
            if (one_set_tiles && sig_runs_in_tile == 1 && sig_d >= kMfmaMinViews && batches_in_tile + (cur_obs > 0 ? 1 : 0) >= kMfmaMinBatches) {
                close_batch(t);
                close_tile(1);
            }
            int newb = 0, newc = 0;
            for (int a = 0; a < d; ++a) {
                if (!cmap.count(c[a])) newc++;
                for (int b = a; b < d; ++b) if (!smap.count(((uint64_t)c[a] << 32) | (uint32_t)c[b])) newb++;
            }
            if ((int)smap.size() + newb > kNSlot || (int)cmap.size() + newc > kNCSlot) {
                close_batch(t);
                close_tile(1);
            }
            sig_runs_in_tile++;
            sig_slots.clear(); sig_cslots.clear();
            for (int a = 0; a < d; ++a) {
                auto ic = cmap.find(c[a]);
                if (ic == cmap.end()) {
                    int cs = (int)cmap.size();
                    ic = cmap.emplace(c[a], cs).first;
                    pl.cslot_cam.push_back(c[a]);
                }
                sig_cslots.push_back((uint8_t)ic->second);
                for (int b = a; b < d; ++b) {
                    uint64_t key = ((uint64_t)c[a] << 32) | (uint32_t)c[b];
                    auto it = smap.find(key);
                    int s;
                    if (it == smap.end()) {
                        s = (int)smap.size();
                        smap.emplace(key, s);
                        int g = pl.upper_block(c[a], c[b]);
                        if (g < 0) { tiling_error = "internal: block missing from pattern"; return; }
                        pl.slot_gblk.push_back(g);
                    } else s = it->second;
                    sig_slots.push_back(s);
                }
            }
            sig_cams = c; sig_d = d;
            if (!dry) ver_slots.push_back(sig_slots);
        }
        if ((int)round_fill + d * (d + 1) / 2 > kWave) flush_round();
        round_fill += (size_t)d * (d + 1) / 2;
        if (!dry) {
            for (int a = 0; a < d; ++a) pl.obs_cslot[pl.pt_obs[t] + a] = sig_cslots[a];
            bpts.push_back({ (int)ver_slots.size() - 1, d, cur_obs, pts_in_batch });
        }
        cur_obs += d; pts_in_batch++;
    }
    close_batch(t);
    close_tile(1);
    if (dry) { pl.batch_pair.push_back(0); return; }
    pl.npair_lds = (int64_t)pl.pair.size();
    pl.ntile_lds = pl.ntile();
    pl.nbatch_lds = pl.nbatch();

    // class 1: row tiles -- batches of whole points, no pair entries: the linearisation leaves one record per observation
    // and k_schur_blocks sums the blocks from the pair lists built below
    rounds_mode = false;
    t_rows = t;
    for (; t < npl; ++t) {
        int d = pl.pt_obs[t + 1] - pl.pt_obs[t];
        if (cls[pl.perm[pl.pt_begin + t]] != 1) break;
        if (cur_obs + d > kWave || pts_in_batch >= kMaxPtsBatch) {
            close_batch(t);
            if (batches_in_tile >= tile_max_batches) close_tile(0);
        }
        cur_obs += d; pts_in_batch++;
    }
    close_batch(t);
    close_tile(0);

    // class 2: long points
    for (; t < npl; ++t) pl.long_pts.push_back(t);
    pl.batch_pair.push_back(0);                       // nbatch + 1 entries, like batch_obs
    };

    stage("local shard arrays");
    do_tiling(tile_max_batches, auto_tile);   // automatic tile length: a dry pass first, to see where the budgets force tiles to end
    if (!tiling_error.empty()) return tiling_error;
    stage(auto_tile ? "tiling, dry pass" : "tiling");
    if (auto_tile) {
        // k_build runs 4 one-wave workgroups per CU = kTileSlots tiles at a time.  Natural segments = runs of
        // batches between two forced tile ends; choose the length cap whose tile count fills whole rounds of
        // slots best (a 5.1-round grid costs 6 rounds), preferring long tiles; small problems get many tiles.
        std::vector<int> segs;
        int run = 0;
        for (auto &tl : tile_log) { run += tl.first; if (!tl.second) { segs.push_back(run); run = 0; } }
        if (run) segs.push_back(run);
        // cost model: a tile costs (batches x Tb + Tf) with Tf the fixed prologue / flush share (~2.5 batch times: the stamps of round 3); the
        // hardware hands tiles in launch order to whichever of the kTileSlots resident slots frees first, so the build lasts as
        // long as the longest slot of that list schedule (simulated per candidate length: a heap over the slots).  (Until round 3
        // the model was "rounds of kTileSlots tiles, each as long as its longest tile".  Measured on C3, build ms by length cap:
        // 12: 0.854, 16: 0.856, 20: 0.816, 24: 0.834, 28: 0.866 -- many runs of 57-63 batches then end in a tile of a few batches
        // that pays a whole flush --, 29 / 30: 0.80, 32: 0.787, 40: 1.01.)
        constexpr double kTileFixed = 2.5;
        double best = 1e300; int best_tmb = tile_max_batches;
        std::vector<double> slot;
        for (int tmb = kMaxTileBatches; tmb >= 2; --tmb) {
            long long tiles = 0;
            for (int sg : segs) tiles += (sg + tmb - 1) / tmb;
            if (!tiles) break;
            double cost;
            if (tiles <= kTileSlots) {            // everything resident at once: the longest tile
                int longest = 0;
                for (int sg : segs) longest = std::max(longest, std::min(sg, tmb));
                cost = longest + kTileFixed;
            } else {
                slot.assign(kTileSlots, 0.0);     // min-heap of the slots' finish times
                auto later = [](double a, double b) { return a > b; };
                for (int sg : segs)
                    for (int left = sg; left > 0; left -= tmb) {
                        std::pop_heap(slot.begin(), slot.end(), later);
                        slot.back() += std::min(left, tmb) + kTileFixed;
                        std::push_heap(slot.begin(), slot.end(), later);
                    }
                cost = *std::max_element(slot.begin(), slot.end());
            }
            if (cost < best * 0.999) { best = cost; best_tmb = tmb; }
        }
        if (timing) std::fprintf(stderr, "[mi_ba plan]   tile length %d batches (model: %.1f batch times)\n", best_tmb, best);
        do_tiling(best_tmb, false);
        if (!tiling_error.empty()) return tiling_error;
        stage("tile length + tiling");
    }
    // batches for the back-substitution: the same points, packed 64 observations at a time whatever the tiles of the
    // build look like (a tile boundary closes a build batch; on scenes with ever-changing camera sets that leaves one
    // point per batch, and k_backsub has no use for tiles)
    {
        // The points of uniform tiles (mode 4) are k_backsub_uni's, tile by tile: k_backsub's batches leave them out.  A batch is a run of
        // CONSECUTIVE points, so a skipped range closes the batch being filled.  (MI_BA_UNI_BS=0: every point to k_backsub -- probes.)
        const char *ub = std::getenv("MI_BA_UNI_BS");
        pl.bs_uni = pl.ntile_uni > 0 && !(ub && ub[0] == '0');
        std::vector<uint8_t> in_uni(npl, 0);
        if (pl.bs_uni)
            for (int tl = 0; tl < pl.ntile_lds; ++tl)
                if (pl.tile_mode[tl] == 4)
                    for (int q = pl.batch_pt[pl.tile_batch[tl]]; q < pl.batch_pt[pl.tile_batch[tl + 1]]; ++q) in_uni[q] = 1;
        pl.bs_batch_obs.clear(); pl.bs_batch_pt.clear();
        pl.bs_batch_obs_end.clear(); pl.bs_batch_pt_end.clear();
        const int nshort = npl - (int)pl.long_pts.size();      // long points come last and have their own kernels
        int cur = 0, pts = 0, first = -1;
        auto close = [&](int t_end) {
            if (first < 0) return;
            pl.bs_batch_obs.push_back(pl.pt_obs[first]); pl.bs_batch_pt.push_back(first);
            pl.bs_batch_obs_end.push_back(pl.pt_obs[t_end]); pl.bs_batch_pt_end.push_back(t_end);
            first = -1; cur = 0; pts = 0;
        };
        for (int t2 = 0; t2 < nshort; ++t2) {
            if (in_uni[t2]) { close(t2); continue; }
            const int d = pl.pt_obs[t2 + 1] - pl.pt_obs[t2];
            if (cur + d > kWave || pts >= kMaxPtsBatch) close(t2);
            if (first < 0) first = t2;
            cur += d; ++pts;
        }
        close(nshort);
    }
    stage("back-substitution batches");
    // ---- row path: records camera-major (points ascending inside a camera), and for every block of S the list of its
    // (record of the row camera, record of the column camera) pairs, in point order.  The role of sba's per-camera scans
    // (sba_crsm_col_elmidxs, sba_levmar.c:1185) -- done once per problem instead of once per camera per iteration.
    {
        pl.nrow_pts = npl - t_rows;
        pl.obs_rec.assign(nol, -1);
        pl.rec_cam.assign(m + 1, 0);
        const int o_rows = pl.pt_obs[t_rows];
        for (int o = o_rows; o < nol; ++o) pl.rec_cam[pl.obs_cam[o] + 1]++;
        for (int j = 0; j < m; ++j) pl.rec_cam[j + 1] += pl.rec_cam[j];
        {
            std::vector<int> fill(pl.rec_cam.begin(), pl.rec_cam.end() - 1);
            for (int o = o_rows; o < nol; ++o) pl.obs_rec[o] = fill[pl.obs_cam[o]]++;
        }
        const int nbu = pl.nb_up();
        const int nrow = npl - t_rows;
        const int nt = nrow < (1 << 14) ? 1 : (int)std::min<unsigned>(8u, std::max(1u, std::thread::hardware_concurrency()));
        // super-blocks: decided on the pair count (every pair of observations of a row point).  MI_BA_SUPER: 0 off, 4 (or 1) the groups of four (the default), 8 both levels (measured slower on C4)
        int levels = 0;
        {
            int64_t np = 0;
            for (int tt = t_rows; tt < npl; ++tt) { const int64_t d = pl.pt_obs[tt + 1] - pl.pt_obs[tt]; np += d * (d - 1) / 2; }
            const char *env = std::getenv("MI_BA_SUPER");
            const int v = env ? std::atoi(env) : (np >= kSuperMinPairs ? 4 : 0);
            levels = cam_split ? 0 : v == 0 ? 0 : v == 8 ? 2 : 1;
        }
        bool super = levels > 0;
        // One level = the groups of GS consecutive cameras and the group pairs (row group < column group) with a block in the pattern, as a CSR over
        // row groups (column groups ascending); entries of 2 GS records; per-thread counts, then positions, of the two passes
        struct Level {
            int GS = 0, ngrp = 0, nsup = 0, round = 1;               // round: entries per round of the kernel (lists padded to whole rounds)
            std::vector<int> row, col, grp_of;
            std::vector<std::vector<int>> cnt;
            std::vector<std::vector<int64_t>> pos;
            std::vector<int64_t> pairs;
            std::vector<int> *blk = nullptr; std::vector<int64_t> *q = nullptr; std::vector<int32_t> *ent = nullptr;
            int find(int G, int H) const {
                const int *b = col.data() + row[G], *e = col.data() + row[G + 1];
                const int *it = std::lower_bound(b, e, H);
                return (it == e || *it != H) ? -1 : (int)(it - col.data());
            }
        };
        Level lv[2];
        lv[0].GS = kSuperCams; lv[0].round = 4; lv[0].blk = &pl.sb_blk; lv[0].q = &pl.sb_q; lv[0].ent = &pl.sb_ent;
        lv[1].GS = 2 * kSuperCams; lv[1].round = 1; lv[1].blk = &pl.sb8_blk; lv[1].q = &pl.sb8_q; lv[1].ent = &pl.sb8_ent;
        for (int L = 0; L < levels; ++L) {
            Level &v = lv[L];
            v.ngrp = (m + v.GS - 1) / v.GS;
            v.row.assign(v.ngrp + 1, 0);
            std::vector<int> mark(v.ngrp, -1);
            for (int G = 0; G < v.ngrp; ++G) {
                std::vector<int> cols;
                for (int j = G * v.GS; j < std::min(m, (G + 1) * v.GS); ++j)
                    for (int q = pl.up_row[j]; q < pl.up_row[j + 1]; ++q) {
                        const int H = pl.up_col[q] / v.GS;
                        if (H != G && mark[H] != G) { mark[H] = G; cols.push_back(H); }
                    }
                std::sort(cols.begin(), cols.end());
                v.col.insert(v.col.end(), cols.begin(), cols.end());
                v.row[G + 1] = (int)v.col.size();
            }
            v.nsup = (int)v.col.size();
            v.grp_of.resize(v.nsup);
            for (int G = 0; G < v.ngrp; ++G) for (int q = v.row[G]; q < v.row[G + 1]; ++q) v.grp_of[q] = G;
            v.cnt.assign(nt, std::vector<int>(nrow ? v.nsup : 0, 0));
            v.pos.assign(nt, std::vector<int64_t>());
            v.pairs.assign(nt, 0);
        }
        constexpr int GS = kSuperCams;
        std::vector<std::vector<int>> cnt(nt, std::vector<int>(nrow ? nbu : 0, 0));
        std::atomic<bool> missing(false);
        auto walk = [&](int k, bool fill_pass, std::vector<int64_t> *pos) {
            const int b0 = t_rows + (int)((int64_t)nrow * k / nt), b1 = t_rows + (int)((int64_t)nrow * (k + 1) / nt);
            // one entry of level v: the observations [a0, a1) (row group) and [e0, e1) (column group) of the point whose observations start at os
            auto entry = [&](Level &v, const int *c, int os, int a0, int a1, int e0, int e1) {
                const int sidx = v.find(c[a0] / v.GS, c[e0] / v.GS);
                if (sidx < 0) { missing = true; return; }
                if (!fill_pass) { v.cnt[k][sidx]++; v.pairs[k] += (int64_t)(a1 - a0) * (e1 - e0); }
                else {
                    int32_t *ent = v.ent->data() + 2 * v.GS * (size_t)v.pos[k][sidx]++;
                    for (int x = a0; x < a1; ++x) ent[c[x] % v.GS] = pl.obs_rec[os + x];
                    for (int x = e0; x < e1; ++x) ent[v.GS + c[x] % v.GS] = pl.obs_rec[os + x];
                }
            };
            for (int tt = b0; tt < b1; ++tt) {
                const int os = pl.pt_obs[tt], d = pl.pt_obs[tt + 1] - os;
                const int *c = pl.obs_cam.data() + os;
                for (int a = 0; a < d; ++a) {
                    const int *rb = pl.up_col.data() + pl.up_row[c[a]], *re = pl.up_col.data() + pl.up_row[c[a] + 1];
                    const int *it = rb;
                    for (int b = a + 1; b < d; ++b) {
                        if (super && c[b] / GS != c[a] / GS) break;          // (cameras ascend: every later one is in another group too) -> the entries below
                        it = std::lower_bound(it, re, c[b]);             // cameras ascend inside a point: the search only moves right
                        if (it == re || *it != c[b]) { missing = true; return; }
                        const int g = (int)(it - pl.up_col.data());
                        if (!fill_pass) cnt[k][g]++;
                        else {
                            const int64_t q = (*pos)[g]++;
                            pl.rpair[2 * q] = (uint32_t)pl.obs_rec[os + a]; pl.rpair[2 * q + 1] = (uint32_t)pl.obs_rec[os + b];
                        }
                    }
                }
                if (!super) continue;
                // The point's runs of observations by camera group, one entry per pair of runs.  Two levels: a point that reaches kSuper8MinGroups
                // groups of eight or more pairs its runs of EIGHT (16 records serve up to 64 pairs) and, inside a run of eight, its two runs of
                // four; any other point pairs its runs of four (8 records, up to 16 pairs: short tracks would leave most of an 8 x 8 entry idle).
                bool two = false;
                if (levels > 1) {
                    int n8 = 1;
                    for (int x = 1; x < d; ++x) n8 += c[x] / lv[1].GS != c[x - 1] / lv[1].GS;
                    two = n8 >= kSuper8MinGroups;
                }
                Level &top = lv[two ? 1 : 0];
                for (int a0 = 0; a0 < d;) {
                    int a1 = a0 + 1;
                    while (a1 < d && c[a1] / top.GS == c[a0] / top.GS) ++a1;
                    if (two) {                                             // the run of eight's two halves
                        int mid = a0;
                        while (mid < a1 && c[mid] / GS == c[a0] / GS) ++mid;
                        if (mid < a1) entry(lv[0], c, os, a0, mid, mid, a1);
                    }
                    for (int e0 = a1; e0 < d;) {
                        int e1 = e0 + 1;
                        while (e1 < d && c[e1] / top.GS == c[e0] / top.GS) ++e1;
                        entry(top, c, os, a0, a1, e0, e1);
                        e0 = e1;
                    }
                    if (missing) return;
                    a0 = a1;
                }
            }
        };
        auto run_threads = [&](const std::function<void(int)> &fn) {
            if (nt == 1) { fn(0); return; }
            std::vector<std::thread> th;
            for (int k = 0; k < nt; ++k) th.emplace_back(fn, k);
            for (auto &x : th) x.join();
        };
        if (nrow) run_threads([&](int k) { walk(k, false, nullptr); });
        if (missing) return "internal: block missing from pattern";
        if (super && nrow && !std::getenv("MI_BA_SUPER")) {
            // an entry pays when it stands for several pairs: cameras that see the same points have to be neighbours in the numbering (a camera
            // path, clusters numbered one after the other).  Where they are not -- fewer than kSuperMinFill pairs per entry of the first level --
            // an entry is little more than a pair with a whole round's overhead: the pair lists then
            int64_t ents = 0, prs = 0;
            for (int k = 0; k < nt; ++k) { for (int c2 : lv[0].cnt[k]) ents += c2; prs += lv[0].pairs[k]; }
            if (ents > 0 && (double)prs < kSuperMinFill * (double)ents) {
                super = false; levels = 0;
                for (auto &c2 : cnt) std::fill(c2.begin(), c2.end(), 0);
                run_threads([&](int k) { walk(k, false, nullptr); });
                if (missing) return "internal: block missing from pattern";
            }
        }
        // block g's pairs: [base[g], base[g + 1]), thread k's share of them after the shares of the threads before it
        std::vector<int64_t> base(nbu + 1, 0);
        std::vector<std::vector<int64_t>> pos(nt);
        if (nrow) {
            for (int g = 0; g < nbu; ++g) { int64_t tot = 0; for (int k = 0; k < nt; ++k) tot += cnt[k][g]; base[g + 1] = base[g] + tot; }
            for (int k = 0; k < nt; ++k) pos[k].resize(nbu);
            for (int g = 0; g < nbu; ++g) { int64_t at = base[g]; for (int k = 0; k < nt; ++k) { pos[k][g] = at; at += cnt[k][g]; } }
        }
        pl.sb_blk.clear(); pl.sb_q.assign(1, 0); pl.sb_ent.clear(); pl.sb8_blk.clear(); pl.sb8_q.assign(1, 0); pl.sb8_ent.clear(); pl.nsuper_pairs = 0;
        for (int L = 0; nrow && L < levels; ++L) {
            Level &v = lv[L];
            // work items in the order of falling entry counts (one wavefront each, dealt to the chip in this order: the long ones first)
            std::vector<int64_t> tot(v.nsup, 0);
            for (int s2 = 0; s2 < v.nsup; ++s2) for (int k = 0; k < nt; ++k) tot[s2] += v.cnt[k][s2];
            std::vector<int> order;
            for (int s2 = 0; s2 < v.nsup; ++s2) if (tot[s2] > 0) order.push_back(s2);
            std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return tot[x] > tot[y]; });
            std::vector<int64_t> sbase(v.nsup, 0);
            int64_t at = 0;
            for (int s2 : order) {
                sbase[s2] = at;
                at += (tot[s2] + v.round - 1) / v.round * v.round;
                v.q->push_back(at);
                const int G = v.grp_of[s2], H = v.col[s2];
                for (int ja = 0; ja < v.GS; ++ja)
                    for (int kb = 0; kb < v.GS; ++kb) {
                        const int j = G * v.GS + ja, k2 = H * v.GS + kb;
                        v.blk->push_back(j < m && k2 < m ? pl.upper_block(j, k2) : -1);
                    }
            }
            v.ent->assign((size_t)(2 * v.GS * at), -1);
            for (int k = 0; k < nt; ++k) v.pos[k].assign(v.nsup, 0);
            for (int s2 = 0; s2 < v.nsup; ++s2) { int64_t a2 = sbase[s2]; for (int k = 0; k < nt; ++k) { v.pos[k][s2] = a2; a2 += v.cnt[k][s2]; } }
            for (int k = 0; k < nt; ++k) pl.nsuper_pairs += v.pairs[k];
        }
        pl.rpair.assign((size_t)(2 * base[nbu]), 0u);
        if (nrow) run_threads([&](int k) { walk(k, true, &pos[k]); });
        if (missing) return "internal: block missing from pattern";
        pl.wk_blk.clear(); pl.wk_q.assign(1, 0); pl.wk_diag_cam.clear(); pl.wk_diag_blk.clear();
        for (int g = 0; g < nbu; ++g)
            if (base[g + 1] > base[g]) { pl.wk_blk.push_back(g); pl.wk_q.push_back(base[g + 1]); }
        for (int j = 0; j < m; ++j)
            if (pl.rec_cam[j + 1] > pl.rec_cam[j]) { pl.wk_diag_cam.push_back(j); pl.wk_diag_blk.push_back(pl.upper_block(j, j)); }
        stage("row path: records + pair lists");
    }
    if (cam_split) {
        // nine camera parameters: the pattern of S over VIRTUAL cameras.  Real row a = [a = b_0 < b_1 < ...] (n_a entries) becomes
        //   row 2a    : (2a,2a) (2a,2a+1) (2a,2b_1) (2a,2b_1+1) ...          2 n_a entries, entry k of the real row at 2k, 2k + 1
        //   row 2a + 1: (2a+1,2a+1) (2a+1,2b_1) (2a+1,2b_1+1) ...            2 n_a - 1 entries, entry k >= 1 at 2k - 1, 2k
        const std::vector<int> rrow = pl.up_row, rcol = pl.up_col;
        const int mv = 2 * m;
        std::vector<int> vrow(mv + 1, 0);
        for (int a = 0; a < m; ++a) { const int na = rrow[a + 1] - rrow[a]; vrow[2 * a + 1] = vrow[2 * a] + 2 * na; vrow[2 * a + 2] = vrow[2 * a + 1] + 2 * na - 1; }
        std::vector<int> vcol(vrow[mv]);
        for (int a = 0; a < m; ++a)
            for (int k = 0; k < rrow[a + 1] - rrow[a]; ++k) {
                const int b = rcol[rrow[a] + k];
                vcol[vrow[2 * a] + 2 * k] = 2 * b; vcol[vrow[2 * a] + 2 * k + 1] = 2 * b + 1;
                if (k == 0) vcol[vrow[2 * a + 1]] = 2 * a + 1;
                else { vcol[vrow[2 * a + 1] + 2 * k - 1] = 2 * b; vcol[vrow[2 * a + 1] + 2 * k] = 2 * b + 1; }
            }
        auto vblocks = [&](int g, int a, int out[4]) {      // real upper block g of real row a
            const int k = g - rrow[a];
            out[0] = vrow[2 * a] + 2 * k; out[1] = out[0] + 1;
            out[2] = k == 0 ? -1 : vrow[2 * a + 1] + 2 * k - 1; out[3] = k == 0 ? vrow[2 * a + 1] : vrow[2 * a + 1] + 2 * k;
        };
        std::vector<int> row_of(rcol.size());
        for (int a = 0; a < m; ++a) for (int q = rrow[a]; q < rrow[a + 1]; ++q) row_of[q] = a;
        pl.wk_vblk.resize(pl.wk_blk.size() * 4); pl.wk_diag_vblk.resize(pl.wk_diag_blk.size() * 4);
        for (size_t w = 0; w < pl.wk_blk.size(); ++w) vblocks(pl.wk_blk[w], row_of[pl.wk_blk[w]], &pl.wk_vblk[4 * w]);
        for (size_t w = 0; w < pl.wk_diag_blk.size(); ++w) vblocks(pl.wk_diag_blk[w], pl.wk_diag_cam[w], &pl.wk_diag_vblk[4 * w]);
        pl.up_row = vrow; pl.up_col = vcol; pl.mv = mv;
        // full pattern of the virtual cameras, as above
        std::vector<int> lowcnt(mv, 0);
        for (int j = 0; j < mv; ++j)
            for (int q = pl.up_row[j]; q < pl.up_row[j + 1]; ++q) if (pl.up_col[q] != j) lowcnt[pl.up_col[q]]++;
        pl.bsr_row.assign(mv + 1, 0);
        for (int j = 0; j < mv; ++j) pl.bsr_row[j + 1] = pl.bsr_row[j] + lowcnt[j] + (pl.up_row[j + 1] - pl.up_row[j]);
        const int nbf = pl.bsr_row[mv];
        pl.bsr_col.assign(nbf, 0); pl.bsr_src.assign(nbf, 0); pl.bsr_trans.assign(nbf, 0);
        std::vector<int> fill(pl.bsr_row.begin(), pl.bsr_row.end() - 1);
        for (int j = 0; j < mv; ++j)
            for (int q = pl.up_row[j]; q < pl.up_row[j + 1]; ++q) {
                const int k = pl.up_col[q];
                if (k == j) continue;
                const int pos = fill[k]++;
                pl.bsr_col[pos] = j; pl.bsr_src[pos] = q; pl.bsr_trans[pos] = 1;
            }
        for (int j = 0; j < mv; ++j)
            for (int q = pl.up_row[j]; q < pl.up_row[j + 1]; ++q) {
                const int pos = fill[j]++;
                pl.bsr_col[pos] = pl.up_col[q]; pl.bsr_src[pos] = q; pl.bsr_trans[pos] = 0;
            }
        stage("virtual cameras");
    }
    return "";
}

} // namespace miba

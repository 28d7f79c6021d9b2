// capi.cpp -- the C ABI of include/mi_ba.h over miba::Engine / miba::Plan.
#include <cstring>
#include <new>
#include <string>

#include <rccl/rccl.h>

#include "engine.h"

using miba::Engine;
using miba::Plan;

struct mi_ba_engine { Engine *e; };
struct mi_ba_plan { Plan pl; };

static thread_local std::string g_create_error;

template <class T, class A, class U>
static int64_t copy_out(const std::vector<T, A> &v, U *out, int64_t cap)
{
    const int64_t n = (int64_t)v.size();
    if (out) for (int64_t i = 0; i < n && i < cap; ++i) out[i] = (U)v[i];
    return n;
}

extern "C" {

void mi_ba_default_config(mi_ba_config *c)
{
    std::memset(c, 0, sizeof(*c));
    c->est_focal = 1; c->undistort = 0; c->explicit_centers = 1; c->mcon = 0;
    c->proj_sign = -1.0; c->f_scale = 0.001; c->k_scale = 5.0;
    c->itmax = 150;                                     // MAX_ITERS, sfm.c:820
    const double o[6] = { 1.0e-3, 1.0e-10, 1.0e-12, 1.0e-12, 0.0, 4.0e-2 }; // sfm.c:711-719
    std::memcpy(c->opts, o, sizeof(o));
    c->pcg_maxit = 100; c->pcg_tol = 1.0e-3;
    c->device = 0; c->rank = 0; c->world = 1; c->verbose = 0; c->tile_max_batches = 0;
}

mi_ba_engine *mi_ba_create(const mi_ba_config *cfg)
{
    if (!cfg) { g_create_error = "null config"; return nullptr; }
    Engine *e = new (std::nothrow) Engine(*cfg);
    if (!e) { g_create_error = "out of memory"; return nullptr; }
    if (!e->ok()) { g_create_error = e->err; delete e; return nullptr; }
    mi_ba_engine *h = new (std::nothrow) mi_ba_engine{ e };
    if (!h) { delete e; g_create_error = "out of memory"; return nullptr; }
    return h;
}

void mi_ba_destroy(mi_ba_engine *h) { if (h) { delete h->e; delete h; } }
const char *mi_ba_last_error(const mi_ba_engine *h) { return h ? h->e->err.c_str() : g_create_error.c_str(); }
int mi_ba_cnp(const mi_ba_engine *h) { return h ? h->e->cnp : MI_BA_ERROR; }

int mi_ba_set_cameras(mi_ba_engine *h, int m, const double *a0, const double *R0, const double *f_fixed)
{ return h ? h->e->set_cameras(m, a0, R0, f_fixed) : MI_BA_ERROR; }
int mi_ba_set_fixed_cameras(mi_ba_engine *h, const unsigned char *mask) { return h ? h->e->set_fixed_cameras(mask) : MI_BA_ERROR; }
int mi_ba_set_fixed_params(mi_ba_engine *h, const unsigned char *mask) { return h ? h->e->set_fixed_params(mask) : MI_BA_ERROR; }
int mi_ba_set_camera_constraints(mi_ba_engine *h, const double *w, const double *v) { return h ? h->e->set_camera_constraints(w, v) : MI_BA_ERROR; }
int mi_ba_comm_set_host_allreduce(mi_ba_engine *h, mi_ba_host_allreduce_fn fn, void *user) { return h ? h->e->set_host_allreduce(fn, user) : MI_BA_ERROR; }
int mi_ba_set_camera_links(mi_ba_engine *h, int nlink, const int *a, const int *b, const double *w) { return h ? h->e->set_camera_links(nlink, a, b, w) : MI_BA_ERROR; }
int mi_ba_set_point_constraints(mi_ba_engine *h, const double *w, const double *v) { return h ? h->e->set_point_constraints(w, v) : MI_BA_ERROR; }
int mi_ba_set_points(mi_ba_engine *h, int n, const double *xyz) { return h ? h->e->set_points(n, xyz) : MI_BA_ERROR; }
int mi_ba_set_observations(mi_ba_engine *h, int64_t nobs, const double *xy, const int *pt_idx, const int *cam_idx)
{ return h ? h->e->set_observations(nobs, xy, pt_idx, cam_idx) : MI_BA_ERROR; }
int mi_ba_prepare(mi_ba_engine *h) { return h ? h->e->prepare() : MI_BA_ERROR; }
int mi_ba_run(mi_ba_engine *h, double info[MI_BA_INFOSZ]) { return h ? h->e->run(info) : MI_BA_ERROR; }
int mi_ba_begin(mi_ba_engine *h) { return h ? h->e->begin() : MI_BA_ERROR; }
int mi_ba_restart(mi_ba_engine *h) { return h ? h->e->restart() : MI_BA_ERROR; }
int mi_ba_step(mi_ba_engine *h) { return h ? h->e->step() : MI_BA_ERROR; }
int mi_ba_get_info(const mi_ba_engine *h, double info[MI_BA_INFOSZ]) { if (!h) return MI_BA_ERROR; h->e->fill_info(info); return 0; }
int mi_ba_get_trace(const mi_ba_engine *h, double *trace, int cap)
{
    if (!h) return MI_BA_ERROR;
    const int n = (int)(h->e->trace.size() / 6);
    if (trace) std::memcpy(trace, h->e->trace.data(), sizeof(double) * 6 * (size_t)(n < cap ? n : cap));
    return n;
}
int mi_ba_get_params(mi_ba_engine *h, double *a, double *xyz) { return h ? h->e->get_params(a, xyz) : MI_BA_ERROR; }
int mi_ba_set_params(mi_ba_engine *h, const double *a, const double *xyz) { return h ? h->e->set_params(a, xyz) : MI_BA_ERROR; }
int mi_ba_get_rotations(mi_ba_engine *h, double *R) { return h ? h->e->get_rotations(R) : MI_BA_ERROR; }
int mi_ba_debug_linearize(mi_ba_engine *h, double mu, double *S, double *E, double *ea, double *eb, double *V, double *diagU, double *cost)
{ return h ? h->e->debug_linearize(mu, S, E, ea, eb, V, diagU, cost) : MI_BA_ERROR; }
int mi_ba_debug_solve(mi_ba_engine *h, double mu, double *da, double *db, double *cost_new)
{ return h ? h->e->debug_solve(mu, da, db, cost_new) : MI_BA_ERROR; }
// (the symbol the oldest clients bind: they pass the 17-double struct of round 1, so this one never writes more than that -- the
// fields every version of the header shares; the whole block goes through mi_ba_get_timing_n with the caller's own sizeof)
int mi_ba_get_timing(const mi_ba_engine *h, mi_ba_timing *t) { if (!h || !t) return MI_BA_ERROR; std::memcpy(t, &h->e->timing, 17 * sizeof(double)); return 0; }
static_assert(sizeof(mi_ba_timing) == 32 * sizeof(double), "mi_ba_timing keeps its size: take new fields out of reserved[]");
int mi_ba_get_timing_n(const mi_ba_engine *h, void *t, size_t bytes)
{
    if (!h || !t) return MI_BA_ERROR;
    std::memcpy(t, &h->e->timing, std::min(bytes, sizeof(mi_ba_timing)));
    return 0;
}
int mi_ba_set_timing_detail(mi_ba_engine *h, int level) { if (!h) return MI_BA_ERROR; h->e->timing_detail = level; return 0; }
int mi_ba_reset_timing(mi_ba_engine *h)
{
    if (!h) return MI_BA_ERROR;
    const double bytes = h->e->timing.build_bytes, flops = h->e->timing.build_flops, index = h->e->timing.build_index_bytes, hand = h->e->timing.build_block_handovers;
    h->e->timing = mi_ba_timing{};   // (the figures of the plan stay)
    h->e->timing.build_bytes = bytes; h->e->timing.build_flops = flops; h->e->timing.build_index_bytes = index; h->e->timing.build_block_handovers = hand;
    return 0;
}

int mi_ba_comm_unique_id(char id[MI_BA_UNIQUE_ID_BYTES])
{
    ncclUniqueId uid;
    static_assert(sizeof(uid) <= MI_BA_UNIQUE_ID_BYTES, "unique id size");
    if (ncclGetUniqueId(&uid) != ncclSuccess) return MI_BA_ERROR;
    std::memset(id, 0, MI_BA_UNIQUE_ID_BYTES);
    std::memcpy(id, &uid, sizeof(uid));
    return 0;
}
int mi_ba_comm_init(mi_ba_engine *h, const char id[MI_BA_UNIQUE_ID_BYTES]) { return h ? h->e->comm_init(id) : MI_BA_ERROR; }
int mi_ba_debug_blocks(mi_ba_engine *h, double *U, double *W) { return h ? h->e->debug_blocks(U, W) : MI_BA_ERROR; }
int mi_ba_comm_info(const mi_ba_engine *h, int out[4]) { return h && out ? h->e->comm_info(out) : MI_BA_ERROR; }

// ---- planner -------------------------------------------------------------------------------
mi_ba_plan *mi_ba_plan_build(int n, int m, int64_t nobs, const int *pt_idx, const int *cam_idx, int rank, int world, int tile_max_batches)
{
    mi_ba_plan *p = new (std::nothrow) mi_ba_plan;
    if (!p) return nullptr;
    std::string e = miba::build_plan(p->pl, n, m, nobs, pt_idx, cam_idx, rank, world, tile_max_batches);
    if (!e.empty()) { g_create_error = e; delete p; return nullptr; }
    return p;
}
mi_ba_plan *mi_ba_plan_build_rows(int n, int m, int64_t nobs, const int *pt_idx, const int *cam_idx, int rank, int world, int tile_max_batches)
{
    mi_ba_plan *p = new (std::nothrow) mi_ba_plan;
    if (!p) return nullptr;
    std::string e = miba::build_plan(p->pl, n, m, nobs, pt_idx, cam_idx, rank, world, tile_max_batches, nullptr, true);
    if (!e.empty()) { g_create_error = e; delete p; return nullptr; }
    return p;
}
void mi_ba_plan_free(mi_ba_plan *p) { delete p; }

int mi_ba_plan_stats(const mi_ba_plan *p, int64_t s[MI_BA_PLAN_NSTATS])
{
    if (!p) return MI_BA_ERROR;
    const Plan &pl = p->pl;
    std::memset(s, 0, sizeof(int64_t) * MI_BA_PLAN_NSTATS);
    s[MI_BA_PLAN_NPTS_LOCAL] = pl.npts_local(); s[MI_BA_PLAN_NOBS_LOCAL] = pl.nobs_local();
    s[MI_BA_PLAN_NBATCH] = pl.nbatch(); s[MI_BA_PLAN_NTILE] = pl.ntile(); s[MI_BA_PLAN_NPAIR] = (int64_t)pl.pair.size();
    s[MI_BA_PLAN_NB_UPPER] = pl.nb_up(); s[MI_BA_PLAN_NB_FULL] = pl.nb_full(); s[MI_BA_PLAN_NSLOT_TOTAL] = (int64_t)pl.slot_gblk.size();
    s[MI_BA_PLAN_NBATCH_ROWS] = pl.nbatch() - pl.nbatch_lds; s[MI_BA_PLAN_NPAIR_ROWS] = (int64_t)pl.rpair.size() / 2 + pl.nsuper_pairs;
    s[MI_BA_PLAN_NREC] = pl.rec_cam.empty() ? 0 : pl.rec_cam.back(); s[MI_BA_PLAN_NROW_PTS] = pl.nrow_pts;
    s[MI_BA_PLAN_NLONG] = (int64_t)pl.long_pts.size(); s[MI_BA_PLAN_PT_BEGIN] = pl.pt_begin; s[MI_BA_PLAN_PT_END] = pl.pt_end;
    s[MI_BA_PLAN_MAX_SLOTS] = pl.max_slots;
    return 0;
}

int64_t mi_ba_plan_get(const mi_ba_plan *p, const char *name, void *out, int64_t cap)
{
    if (!p || !name) return MI_BA_ERROR;
    const Plan &pl = p->pl;
    const std::string k(name);
#define I32(key, vec) if (k == key) return copy_out(pl.vec, (int32_t *)out, cap)
    I32("perm", perm); I32("obs_cam", obs_cam); I32("obs_pt", obs_pt); I32("batch_obs", batch_obs); I32("batch_pt", batch_pt);
    I32("tile_batch", tile_batch); I32("tile_slot", tile_slot); I32("slot_gblk", slot_gblk); I32("tile_cslot", tile_cslot);
    I32("cslot_cam", cslot_cam); I32("up_row", up_row); I32("up_col", up_col); I32("bsr_row", bsr_row); I32("bsr_col", bsr_col);
    I32("bsr_src", bsr_src); I32("pt_obs", pt_obs); I32("long_pts", long_pts);
    I32("bs_batch_obs", bs_batch_obs); I32("bs_batch_pt", bs_batch_pt); I32("bs_batch_obs_end", bs_batch_obs_end); I32("bs_batch_pt_end", bs_batch_pt_end);
    I32("obs_rec", obs_rec); I32("rec_cam", rec_cam); I32("wk_blk", wk_blk); I32("wk_diag_cam", wk_diag_cam); I32("wk_diag_blk", wk_diag_blk); I32("sb_blk", sb_blk); I32("sb_ent", sb_ent); I32("sb8_blk", sb8_blk); I32("sb8_ent", sb8_ent);
#undef I32
    if (k == "obs_src") return copy_out(pl.obs_src, (int64_t *)out, cap);
    if (k == "batch_pair") return copy_out(pl.batch_pair, (int64_t *)out, cap);
    if (k == "pair") return copy_out(pl.pair, (uint32_t *)out, cap);
    if (k == "rpair") return copy_out(pl.rpair, (uint32_t *)out, cap);
    if (k == "wk_q") return copy_out(pl.wk_q, (int64_t *)out, cap);
    if (k == "sb_q") return copy_out(pl.sb_q, (int64_t *)out, cap);
    if (k == "sb8_q") return copy_out(pl.sb8_q, (int64_t *)out, cap);
    if (k == "obs_cslot") return copy_out(pl.obs_cslot, (uint8_t *)out, cap);
    if (k == "tile_mode") return copy_out(pl.tile_mode, (uint8_t *)out, cap);
    if (k == "bsr_trans") return copy_out(pl.bsr_trans, (uint8_t *)out, cap);
    return MI_BA_ERROR;
}

} // extern "C"

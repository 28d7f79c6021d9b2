// plan.h -- host-side shard planner of the MI355X bundle-adjustment engine.
//
// Role in the reference: the index build sba does at the top of every solve
// (idxij CRS map, sba_levmar.c:652-675) and PBA's InitializeBundle
// (SparseBundleCPU.cpp:2715-2769: point->obs and camera->obs maps).  Here the maps are shaped
// for 64-lane wavefronts instead:
//   * points are permuted so that points with the same camera set are neighbours
//     (visibility-signature sort) -- neighbours then hit the same blocks of S;
//   * a BATCH is <= 64 observations of whole points: one wavefront, one lane per observation;
//   * a TILE is a run of batches whose distinct S blocks fit an LDS accumulator table
//     (NSLOT blocks) and whose distinct cameras fit NCSLOT vectors; the tile is one workgroup;
//   * every (a <= b) observation pair of a point is a PAIR entry (lane-local indices + slot).
//     In LDS tiles the entries come in ROUNDS of exactly 64 (one per lane, padded with
//     kPairInvalid): a round holds whole points, and points with the same camera set put the
//     same block on the same lane round after round, so k_build keeps that block in registers
//     and touches the LDS accumulator only when the lane's block changes;
//   * points whose camera set does not repeat (a signature run shorter than kMinRunPts points), points with more views
//     than an LDS tile's block table holds, and tracks longer than a wavefront take the ROW path instead: the
//     linearisation leaves one 256-byte record per observation in HBM, camera-major (k_build's row tiles,
//     k_long_reduce), and k_schur_blocks sums every block of S from the plan-time list of its (record, record) pairs --
//     one wavefront per block, no atomics, a fixed order (sba_levmar.c:1182-1339 visits any visibility at the same cost);
//   * S = U* - W V*^-1 W^T is block sparse by co-visibility: upper pattern + full BSR pattern.
#pragma once
#include <cstdint>
#include <functional>
#include <memory>
#include <utility>
#include <string>
#include <vector>

namespace miba {

// std::vector whose resize() leaves new elements uninitialised: the per-observation arrays are hundreds of megabytes, filled
// right away by several host threads -- a serial zero fill (and its page faults) first was a quarter of the stage
template <class T>
struct noinit_allocator : std::allocator<T> {
    template <class U> struct rebind { using other = noinit_allocator<U>; };
    template <class U, class... A>
    void construct(U *p, A &&...a)
    {
        if constexpr (sizeof...(A) == 0) ::new ((void *)p) U;
        else ::new ((void *)p) U(std::forward<A>(a)...);
    }
};
template <class T> using uvector = std::vector<T, noinit_allocator<T>>;


constexpr int kWave = 64;        // observations per batch
constexpr int kNSlot = 55;       // S-block accumulators per LDS tile (55 = pairs of a 10-view point)
constexpr int kNCSlot = 24;      // camera accumulators per LDS tile
constexpr int kMaxPtsBatch = 32; // points per batch (bounds the per-point LDS staging)
constexpr int kMaxTileBatches = 63; // batches per tile (k_build keeps a tile's batch descriptors one per lane: 63 batches + the end);
                                    // a tile hands its blocks to HBM when it ends -- ~21 KB of f64 atomics -- so longer is cheaper
constexpr int kTileSlots = 2048;     // concurrently resident k_build workgroups: 256 CUs x 8 (two per SIMD)
constexpr int kMaxRounds = 8;    // pair rounds per LDS-tile batch (k_build preloads them into registers)
constexpr uint32_t kPairInvalid = 0xFFFFFFFFu; // padding entry of a pair round
constexpr int kMaxViewsLds = 10;        // d(d+1)/2 <= kNSlot
constexpr int kMfmaMinViews = 4;        // a tile of ONE camera set with this many views or more per point goes to the matrix pipe (tile mode 3, k_build_mfma)
constexpr int kMfmaMinBatches = 3;      // ... and a tile that holds one such camera set for this many batches ends where the set changes
constexpr int kUniMinViews = 4;         // a tile of ONE camera set with this many views or more per point is a UNIFORM tile (tile mode 4, k_build_uni: lane =
                                        // point x views + camera, the lane's pair and block constant for the whole tile; MI_BA_UNI=0: the general kernel instead)
constexpr int kUniFixedViews = 10;       // the view count k_build_uni also exists for as a compile-time constant (= kMaxViewsLds: every tile of the benchmark scenes)
constexpr int kUniMinBatches = 3;       // ... and a tile that holds one such camera set for this many batches ends where the set changes
constexpr int kTableRuns = 4;           // a tile whose points change camera set more often than this sums its blocks in an LDS table
constexpr int kMinRunPts = 12;          // a camera set shared by fewer consecutive points than this goes to the row path: a register-resident block is
                                        // handed to HBM once per run (49 f64 atomics per block, ~120 G elements/s chip-wide), the row path costs ~2.5x the
                                        // per-point work and no atomics; break-even ~14 points.  MI_BA_MIN_RUN_PTS overrides (probes);
                                        // mi_ba_config.deterministic sends EVERY point down the row path
constexpr int kSuperCams = 4;             // cameras per side of a super-block (plan.h sb_blk; k_schur_super: lane = (entry of the round, row camera, column camera))
constexpr int kSuper8MinGroups = 4;        // a point that reaches this many groups of eight cameras pairs its runs of eight (plan.h sb8_*)
constexpr double kSuperMinFill = 5.0;       // pairs per entry of the groups of four below which the planner keeps the per-block pair lists (16 at most; C4: 13.3)
constexpr int64_t kSuperMinPairs = 1 << 20; // row-path pairs from which the super-blocks are worth their tables
constexpr int kRec = 32;                // doubles per observation record: A (2 x 8) | B (6) | e (2) | G = B V*^-1 (6) | e - B h (2)
constexpr int kRec9 = 36;               // the same with nine camera parameters: A (2 x 10)
constexpr int kCP = 8;           // padded camera block dimension (cnp <= 8)

struct Plan {
    int n = 0, m = 0, rank = 0, world = 1;
    int mv = 0;                  // rows of the reduced system's block pattern: m, or 2 m with nine camera parameters (cam_split): camera j
                                 // = rows 2j (parameters 0-7) and 2j + 1 (parameter 8) -- up_*, bsr_* then describe the VIRTUAL cameras
    int64_t nobs = 0;
    int n_active = 0;            // points with >= 1 observation (internal order: actives first)
    int pt_begin = 0, pt_end = 0;// this rank's points, internal order
    std::vector<int> perm;       // internal -> original point index (all n)

    // block pattern of S (global, identical on every rank)
    std::vector<int> up_row, up_col;                 // upper incl. diagonal, CSR over cameras
    std::vector<int> bsr_row, bsr_col, bsr_src;      // full pattern; bsr_src = upper block id
    std::vector<uint8_t> bsr_trans;                  // 1: block is the transpose of upper block bsr_src

    // local shard, internal order
    std::vector<int> pt_obs;                         // npl+1 local observation offsets
    uvector<int> obs_cam, obs_pt;                    // per local observation (uvector: sized without a serial zero fill, first touched by the threads that write them)
    uvector<int64_t> obs_src;                        // original observation index
    uvector<uint8_t> obs_cslot;                      // tile-local camera slot (LDS tiles)
    std::vector<int> batch_obs, batch_pt;            // nbatch+1
    std::vector<int> bs_batch_obs, bs_batch_pt;      // batches of k_backsub: first observation / first point of batch b (the short points that are not in a
    std::vector<int> bs_batch_obs_end, bs_batch_pt_end; // uniform tile, packed regardless of the build's tiles), and one past its last
    bool bs_uni = false;                             // the uniform tiles' points go to k_backsub_uni instead (tile by tile)
    std::vector<int64_t> batch_pair;                 // nbatch+1: first pair entry of the batch | rounds << 48 (batches of one camera set share entries)
    std::vector<int> tile_batch, tile_slot, tile_cslot; // ntile+1
    std::vector<uint8_t> tile_mode;                  // 1 register-accumulated blocks, 2 blocks summed in an LDS table, 3 one camera set: blocks on the
                                                     // matrix pipe (k_build_mfma, opt-in), 4 one camera set on the vector pipe (k_build_uni), 0 row tiles (records to HBM)
    int ntile_table = 0, ntile_mfma = 0, ntile_uni = 0; // tiles of mode 2, of mode 3, of mode 4
    int ntile_uni_fixed = 0;                         // ... of mode 4 with kUniFixedViews views
    std::vector<int> slot_gblk, cslot_cam;
    std::vector<uint32_t> pair;                      // la | lb<<6 | slot<<12 (6 bits) | local point<<18 (LDS tiles only)
    int64_t npair_lds = 0;                           // = pair.size()
    // row path (class 1 and 2 points): records camera-major, points ascending inside a camera
    std::vector<int> obs_rec;                        // per local observation: its record, -1 = an observation of an LDS tile
    std::vector<int> rec_cam;                        // m+1: camera a's records are [rec_cam[a], rec_cam[a+1])
    std::vector<int> wk_blk;                         // work items of k_schur_blocks, off-diagonal blocks: upper block id ...
    std::vector<int64_t> wk_q;                       // ... and its pairs rpair[2 wk_q[w] .. 2 wk_q[w+1])
    std::vector<uint32_t> rpair;                     // per pair: record of the row camera's observation, record of the column camera's
    std::vector<int> wk_diag_cam, wk_diag_blk;       // diagonal work items: cameras with records, and their diagonal block
    // super-blocks (k_schur_super): the pairs whose two cameras lie in DIFFERENT groups of kSuperCams consecutive cameras leave the per-block lists
    // above; work item w = (row group, column group) owns the kSuperCams^2 blocks sb_blk[16 w ..] (-1: not in the pattern) and, per point that
    // both groups see, one ENTRY of 2 kSuperCams records (the point's records of the row group's cameras, then of the column group's; -1 =
    // none) -- a record is fetched once per entry and serves up to four pairs, where a pair list fetches both of its records per pair
    std::vector<int> sb_blk;
    std::vector<int64_t> sb_q;                       // work item w's entries sb_ent[8 sb_q[w] .. 8 sb_q[w + 1]): a multiple of four (padded with all -1)
    std::vector<int32_t> sb_ent;
    // ... and the same with groups of EIGHT cameras (64 blocks per work item, 16 records per entry, one entry per round: no padding): the points with long
    // tracks pair their runs of eight here and only the two halves of a run of eight in the lists above (plan.cpp)
    std::vector<int> sb8_blk;
    std::vector<int64_t> sb8_q;
    std::vector<int32_t> sb8_ent;
    int64_t nsuper_pairs = 0;                        // pairs the entries stand for
    std::vector<int> wk_vblk, wk_diag_vblk;          // cam_split: 4 virtual blocks per work item: (2a,2b) (2a,2b+1) (2a+1,2b) (2a+1,2b+1)
    int64_t nrow_pts = 0;                            // points on the row path (long ones included)
    int ntile_lds = 0, nbatch_lds = 0;
    std::vector<int> long_pts;                       // local points with > kWave observations (row path, workgroup per point)
    int max_slots = 0;

    int64_t npts_local() const { return pt_end - pt_begin; }
    int64_t nobs_local() const { return (int64_t)obs_cam.size(); }
    int nbatch() const { return (int)batch_obs.size() - 1; }
    int ntile() const { return (int)tile_batch.size() - 1; }
    int nb_up() const { return (int)up_col.size(); }
    int nb_full() const { return (int)bsr_col.size(); }
    int upper_block(int j, int k) const; // j <= k, -1 if absent
};

// pt_idx must be non-decreasing.  Returns "" or an error message.
// extra_pairs: camera pairs (2 ints each) whose block of S must exist although no point links them
// (parameter couplings between cameras, kernels.h CamLinks)
// all_rows: every point takes the row path (the deterministic mode: no floating-point atomics anywhere)
std::string build_plan(Plan &pl, int n, int m, int64_t nobs, const int *pt_idx, const int *cam_idx,
                       int rank, int world, int tile_max_batches, const std::vector<int> *extra_pairs = nullptr, bool all_rows = false,
                       bool cam_split = false, const std::function<void()> &shard_arrays_ready = nullptr);
// shard_arrays_ready: called once obs_cam, obs_pt, obs_src and pt_obs hold their final values (the tiling that follows does not touch
// them): the engine starts their upload on another thread while the planner goes on

} // namespace miba

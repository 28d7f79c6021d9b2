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
//   * S = U* - W V*^-1 W^T is block sparse by co-visibility: upper pattern + full BSR pattern.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace miba {

constexpr int kWave = 64;        // observations per batch
constexpr int kNSlot = 55;       // S-block accumulators per LDS tile (55 = pairs of a 10-view point)
constexpr int kNCSlot = 24;      // camera accumulators per LDS tile
constexpr int kMaxPtsBatch = 32; // points per batch (bounds the per-point LDS staging)
constexpr int kMaxTileBatches = 32; // batches per tile (k_build keeps a tile's batch descriptors one per lane)
constexpr int kTileSlots = 2048;     // concurrently resident k_build workgroups: 256 CUs x 8 (two per SIMD)
constexpr int kMaxRounds = 8;    // pair rounds per LDS-tile batch (k_build preloads them into registers)
constexpr uint32_t kPairInvalid = 0xFFFFFFFFu; // padding entry of a pair round
constexpr int kMaxViewsLds = 10;        // d(d+1)/2 <= kNSlot
constexpr int kTableRuns = 4;           // a tile whose points change camera set more often than this sums its blocks in an LDS table
constexpr int kMinRunObs = 0;           // > 0: camera sets repeating over fewer observations go to the direct tiles.  Measured (tools/real_run.py,
                                        // every point a different camera set): 32 made the build 1.2-3x SLOWER than poorly packed register tiles -> off
constexpr int kCP = 8;           // padded camera block dimension (cnp <= 8)

struct Plan {
    int n = 0, m = 0, rank = 0, world = 1;
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
    std::vector<int> obs_cam, obs_pt;                // per local observation
    std::vector<int64_t> obs_src;                    // original observation index
    std::vector<uint8_t> obs_cslot;                  // tile-local camera slot (LDS tiles)
    std::vector<int> batch_obs, batch_pt;            // nbatch+1
    std::vector<int> bs_batch_obs, bs_batch_pt;      // batches of k_backsub (same points, packed regardless of tiles)
    std::vector<int64_t> batch_pair;                 // nbatch+1
    std::vector<int> tile_batch, tile_slot, tile_cslot; // ntile+1
    std::vector<uint8_t> tile_mode;                  // 1 register-accumulated blocks, 2 blocks summed in an LDS table, 0 direct global atomics
    int ntile_table = 0;                             // tiles of mode 2
    std::vector<int> slot_gblk, cslot_cam;
    std::vector<uint32_t> pair;                      // la | lb<<6 | slot<<12 (6 bits) | local point<<18 (LDS tiles)
    std::vector<uint32_t> pair_gblk;                 // direct pairs only: upper block id
    int64_t npair_lds = 0;                           // pairs [0,npair_lds) belong to LDS tiles
    int ntile_lds = 0, nbatch_lds = 0;
    std::vector<int> long_pts;                       // local points with > kWave observations
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
std::string build_plan(Plan &pl, int n, int m, int64_t nobs, const int *pt_idx, const int *cam_idx,
                       int rank, int world, int tile_max_batches, const std::vector<int> *extra_pairs = nullptr);

} // namespace miba

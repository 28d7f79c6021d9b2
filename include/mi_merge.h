/*
 * mi_merge.h -- C ABI of the cluster-merge step around the similarity alignment.
 *
 * Replaces the bodies of the stage programs SiftGPU/Merge3D3D.cpp (main, :564-908; below) and SiftGPU/MergeClusters.cpp (the
 * same fusion with a transform chosen among candidates; mi_mergeclusters_main at the end).  Merge3D3D: for every line "new first second" of
 * <dir>/dendogram/fullclustermergeinfo.txt, two reconstructions are read, their common points found, the second one is
 * brought into the frame of the first (mi_sim3.h on the GPU), the tracks are fused, the merged model and its side files
 * written, and optimizer_all refines it.  What is kept from the reference, quirks included:
 *   - a point's id is the feature-id field of its LAST view; a view whose camera equals the previous view's is dropped
 *     (ReadNVM, :508-523); a point without views has id -1;
 *   - a point of the first model corresponds to the FIRST point of the second model with the same id (:646-683; found
 *     here through a table id -> smallest index instead of the reference's quadratic scan);
 *   - x' = R (s x) + t for the second model's points (:722-736); cameras R' = R R_rt^T, c' = R_rt (s c) + t,
 *     t' = -R' c' (:739-772);
 *   - the second model's camera indices are shifted by the first model's camera count, the views of a matched point are
 *     appended to its partner's track, and ALL points of the second model -- the matched ones too -- are appended (:775-798);
 *   - every view of a point is written with the point's id in the feature-id field (SaveNVM, :558);
 *   - the debug colouring: first model red, second green, matched points blue / white (:612, 625, 662-668); keep_colours
 *     switches that off.
 *
 * Parity is unpinned against the reference (the file needs Eigen and OpenCV, absent here): tests/test_merge.py restates the
 * steps above in numpy and the alignment with oracle/sim3_oracle.c.
 *
 * Errors: NULL / -1 and mi_model_last_error().
 */
#ifndef MI_MERGE_H
#define MI_MERGE_H
#include <stdint.h>

#include "mi_model.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    int rounds;          /* 5000 (Merge3D3D.cpp:327) */
    int device;
    int keep_colours;    /* 0: the reference's colouring, 1: the points keep their colours */
    int reserved[5];
} mi_merge_config;

typedef struct {
    int64_t ncorr;       /* correspondences (:685) */
    double scale, R[9], t[3];
    int best_round;
    double device_ms;    /* the alignment on the GPU */
    int64_t ncam_first, npt_first;   /* the merged model = first's cameras / points, then second's */
} mi_merge_result;

void mi_merge_default_config(mi_merge_config *cfg);
/* a new model (mi_model_free); first and second are not modified */
mi_model *mi_model_merge_3d3d(const mi_model *first, const mi_model *second, const mi_merge_config *cfg, mi_merge_result *res);
/* the same fusion with a given transform x_first = R (scale x_second) + t.  erase_matched = 0: as above; 1: MergeClusters.cpp's
 * variant (:971-995) -- a point of the second model disappears once its views went to its partner (with the reference's
 * behaviour when two points of the first model share a partner: the later one takes, and removes, the next surviving track) */
mi_model *mi_model_merge_with_transform(const mi_model *first, const mi_model *second, double scale, const double R[9], const double t[3],
                                        int erase_matched, int keep_colours, mi_merge_result *res);
/* how well a candidate transform brings the second model onto the first (MergeClusters.cpp:701-861): over the common points,
 * the distance |R (s x_second) + t - x_first|, and the first model's point projected into the second model's moved cameras
 * (radial coefficient k1 applied, :817) against the second model's measurements */
typedef struct {
    int64_t ncorr, nproj, inliers;        /* inliers: distances below their mean (:847-851) */
    double mean_reproj, median_reproj;    /* pixels */
    double median_dist, mean_dist;
} mi_merge_score;
/* dist: ncorr distances in the order of the correspondences, or NULL; debug_path: the reference's debug_<k>.txt, or NULL */
int mi_model_score_transform(const mi_model *first, const mi_model *second, double scale, const double R[9], const double t[3],
                             mi_merge_score *score, double *dist, const char *debug_path);
/* the files the reference leaves beside the merged NVM (:811-900): RTglobalmapped_for_optimisation.txt (R and centre per
 * camera), focal_for_optimisation.txt, Invmap.txt (index -> camera name), ours_new.txt (points with their tracks) and
 * merged.ply (which lists the last nply_tail points twice, as the reference does with the second model's) */
int mi_model_save_merge_side_files(const mi_model *merged, const char *dir, int64_t nply_tail);
/* the process:  Merge3D3D <output dir>.  Spawns optimizer_all (beside this executable, or $MI_OPTIMIZER_ALL) on every
 * merged cluster with the reference's arguments "0 100 16 32 0 0" (:903).  Returns the exit status. */
int mi_merge3d3d_main(int argc, char **argv);
/* the process:  MergeClusters <output dir> <index>  (SiftGPU/MergeClusters.cpp:553-1115): merge number <index> of
 * fullclustermergeinfo.txt with the best of the candidate transforms other stage programs left in the new cluster's directory
 * (RTS2D2D.txt, RTS3D3D.txt, RTS2DR3DT.txt, RTSRefined.txt, RTSRefined_1.txt: scale / 9 numbers / 3 numbers) -- the one with
 * the smallest mean distance (:875-885) -- matched points erased, then optimizer_all "0 50 16 32 0.5 0" when that candidate's
 * mean reprojection error exceeds 10 pixels and "-1 5 16 32 0.5 0" otherwise (:1102-1108).  A missing candidate file is
 * skipped (the reference would stop on it). */
int mi_mergeclusters_main(int argc, char **argv);

#ifdef __cplusplus
}
#endif
#endif

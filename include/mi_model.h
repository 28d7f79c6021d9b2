/*
 * mi_model.h -- reconstruction files of the optimizer's process boundary, and the end-to-end
 * "load -> bundle adjust on the MI355X -> save" call the optimizer_all executable makes.
 *
 * Formats (reference Optimizer/util_VSFM.h):
 *   NVM   "NVM_V3_R9T": per camera  name f R[9 row-major] t[3] d0 d1 ; per point
 *         X Y Z r g b nview {camIdx featId x y}*       LoadNVM :139-215, SaveNVM :218-258, written by
 *         the pipeline's stage programs (SiftGPU/MergeClusters.cpp:520-551).  "NVM_V3" (quaternion +
 *         camera centre) is read as LoadNVM does (:170-175).
 *   BAL   "bundler model" text: ncam npt nobs / cam pt x y / 9 numbers per camera (w3 t3 f k1 k2) /
 *         X Y Z                                           LoadBundlerModel :448-501, SaveBundlerModel :503-529
 *         (-z camera -> +z: y negated on load, rows 2,3 of R and t2,t3 negated = SetInvertedRT)
 *   OUT   Bundler "bundle.out" v0.3: ncam npt / per camera  f k1 k2, R (3 rows), t / per point  X Y Z, r g b,
 *         nview {cam key x y}*   LoadBundlerOut :312-398, SaveBundlerOut :400-446 (-z camera -> +z: SetInvertedR9T,
 *         y negated); camera names from "<name>-list.txt" when present
 * Camera convention in memory (DataInterface.h:50-52): x^ = f (R X + t).xy / (R X + t).z.
 */
#ifndef MI_MODEL_H
#define MI_MODEL_H
#include <stdint.h>
#include "mi_ba.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct mi_model mi_model;

mi_model *mi_model_load_nvm(const char *path);     /* NULL on failure, reason in mi_model_last_error() */
mi_model *mi_model_load_bal(const char *path);
mi_model *mi_model_load_bundler_out(const char *path);
/* the five files the stage programs write before they start optimizer_all (SiftGPU/TriangulationMultiStep.cpp:1024-1088,
 * MergeClusters.cpp:1009-1052): RTglobalmapped_for_optimisation.txt (R, then the camera CENTRE: t = -R C),
 * focal_for_optimisation.txt ("ncam" / "camName focal 2"), ours_new.txt ("X Y Z r g b n {camLocalIdx pointId x y}*"),
 * and optionally Invmap.txt ("localIdx camName": the names of the output) and Distortion.txt (one coefficient per camera).
 * The writer is mi_model_save_merge_side_files (mi_merge.h). */
mi_model *mi_model_load_side_files(const char *dir);
int mi_model_save_nvm(const mi_model *mdl, const char *path);
int mi_model_save_bal(const mi_model *mdl, const char *path);
int mi_model_save_bundler_out(const mi_model *mdl, const char *path);   /* also writes <name>-list.txt */
/* RemoveInvisiblePoints (util_VSFM.h:736-866): drops points that lie behind (or closer than 0.1 % of the median
 * depth to) one of their cameras, then cameras left with < 20 points and points left with < 2 views, and
 * compacts the indices.  Returns the number of points removed (0 = model untouched) or MI_BA_ERROR.
 * Unlike the reference (which loses them through a push_back on the wrong vector, :842) colours are kept. */
int mi_model_remove_invisible_points(mi_model *mdl);
void mi_model_free(mi_model *mdl);
const char *mi_model_last_error(void);

/* counts[0..2] = cameras, points, projections */
int mi_model_counts(const mi_model *mdl, int64_t counts[3]);
/* copies out; any pointer may be NULL.  f ncam, R ncam*9, t ncam*3, d ncam*2, xyz npt*3, rgb npt*3,
 * xy nproj*2, pt_idx / cam_idx nproj, feat_id nproj */
int mi_model_get(const mi_model *mdl, double *f, double *R, double *t, double *d, double *xyz, int *rgb,
                 double *xy, int *pt_idx, int *cam_idx, int64_t *feat_id);

/* shared intrinsics for the next mi_model_optimize: mask[j] = first camera of j's group (mask of that camera = itself),
 * -1 = on its own; soft equality of the focal lengths of a group exactly as ParallelBA::SetFocalMask (pba.h:124;
 * the BAL / Ladybug configuration C4).  NULL or weight <= 0 clears. */
int mi_model_set_focal_mask(mi_model *mdl, const int *mask /* ncam */, double weight);

/* bundle adjust in place.  cfg: only itmax, opts, pcg_*, est_focal, mcon, device, verbose, deterministic are read (the
 * camera convention of the file fixes the rest).  Returns LM iterations or MI_BA_ERROR. */
int mi_model_optimize(mi_model *mdl, const mi_ba_config *cfg, double info[MI_BA_INFOSZ]);
/* one rank of a point-sharded solve (one process per GPU): cfg->rank / cfg->world name the shard and cfg->device
 * the GPU; every rank holds the whole model and ends with the whole answer.  Transport: unique_id (mi_ba_comm_unique_id of
 * rank 0, handed to the others by the caller: RCCL over xGMI) or host_allreduce (mi_ba_comm_set_host_allreduce).
 * rendezvous, optional: called once with ok = "this rank opened its device", returns non-zero when every rank did -- so that
 * no rank enters a collective another one will never reach.  shard NULL = mi_model_optimize. */
typedef struct {
    const char *unique_id;
    mi_ba_host_allreduce_fn host_allreduce;
    void *host_user;
    int (*rendezvous)(void *user, int ok);
    void *rendezvous_user;
} mi_model_shard;
int mi_model_optimize_shard(mi_model *mdl, const mi_ba_config *cfg, double info[MI_BA_INFOSZ], const mi_model_shard *shard);
/* sqrt(sum ||x - x^||^2 / nproj) in pixels with the cameras/points currently in the model (host) */
double mi_model_rmse(const mi_model *mdl);

/* the process ABI:  optimizer_all <dir>/ A B C D E F   (SiftGPU/MergeClusters.cpp:1099-1110).
 * Input: the side files above when <dir> holds them (at two of the six call sites they are the only input), else
 * <dir>/outputVSFM_GB.nvm; refines; writes <dir>/outputVSFM_GB.nvm.  Returns the process exit status.
 * MI_BA_GPUS=N in the environment: the process forks N - 1 further ranks before it touches HIP, one GPU each
 * (MI_BA_DEVICE + rank), points sharded, RCCL all-reduce per trial; rank 0 writes the file. */
int mi_optimizer_all_main(int argc, char **argv);

#ifdef __cplusplus
}
#endif
#endif

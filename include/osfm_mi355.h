/*
 * osfm_mi355.h -- C ABI of libosfm_mi355.so, the MI355X-native drop-in for OpenSfM's
 * feature-matching + bundle-adjustment hot path.
 *
 * Every entry point is plain C: pointers + sizes, int status return (0 = OK, <0 = error, text via
 * osfm_last_error()), no exceptions, no torch types.  Host pointers unless a name says "dev".
 * The library never retains a caller pointer past the call; device memory is owned by the opaque
 * handles below.  Each entry point cites the reference interface it replaces.
 *
 * Index conventions follow the reference: a match is (feature idx in image 1, feature idx in
 * image 2) after the swap done at opensfm/matching.py:697,717-718; keypoints are normalized image
 * coordinates (opensfm/features.py:324-331); poses are angle-axis camera->world rotation + camera
 * ORIGIN (opensfm/src/bundle/data/pose.h:34-43).
 */
#ifndef OSFM_MI355_H
#define OSFM_MI355_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OSFM_OK 0
#define OSFM_E_INVALID (-1)     /* bad argument (size mismatch, null pointer, ...) */
#define OSFM_E_HIP (-2)         /* HIP runtime error (see osfm_last_error) */
#define OSFM_E_UNSUPPORTED (-3) /* input outside the implemented domain */
#define OSFM_E_NOMEM (-4)
#define OSFM_E_NUMERIC (-5) /* NaN/Inf in results: reference throws (ba_helpers.cc:780-814) */

#define OSFM_DESC_DIM 128
#define OSFM_MAX_FEATURES 16000 /* per image (feature_min_frames_panorama, config.py:31): the matcher keeps 6 B of LDS per feature
                                   (160 KiB at 16000); the RANSAC kernel stages its correspondences in LDS up to ~9000 and in HBM above */

typedef struct osfm_ctx osfm_ctx;     /* one per process/GPU: device, streams, scratch */
typedef struct osfm_store osfm_store; /* device-resident descriptor + keypoint store */

/* Thread-local text of the last error raised on the calling thread. */
const char *osfm_last_error(void);
/* Library version string and the gfx arch the kernels were compiled for. */
const char *osfm_version(void);

int osfm_ctx_create(int device, osfm_ctx **out);
void osfm_ctx_destroy(osfm_ctx *ctx);
/* Device the context is bound to and its CU count (0 on error). */
int osfm_ctx_device(const osfm_ctx *ctx);
int osfm_ctx_num_cus(const osfm_ctx *ctx);
/* The context keeps the device blocks of its calls (the matcher's chunk buffers, the slabs a bundle adjustment sub-allocates its arrays
 * from; up to 24 GiB) cached between calls.  Waits for the device,
 * frees every cached block and returns the bytes released; the library calls it itself before it reports OSFM_E_NOMEM. */
int64_t osfm_ctx_trim_pool(osfm_ctx *ctx);

/* ------------------------------------------------------------------------------------------
 * Descriptor store.  Replaces FeatureLoader.load_all_data (opensfm/feature_loading.py:106-173):
 * instead of an LRU of per-image npz loads, all images' masked features live in HBM.
 * counts[n_images] = features per image (0..OSFM_MAX_FEATURES).
 * desc: sum(counts) x 128; float32 or uint8.  Integer-valued in [0,255] (HAHOG / SIFT uchar round trip, features.py:526-534):
 *       exact int8 matrix path.  Any other finite float32 values (root-SIFT with feature_root, features.py:292-298): the store
 *       also keeps the float rows and an 8-bit quantisation; the same kernel decides with rigorous error bounds and evaluates
 *       the undecided queries in float32 exactly as cv2 accumulates (results identical to cv2's float arithmetic as the oracle
 *       restates it).  Non-finite values: OSFM_E_INVALID.
 * pts:  sum(counts) x 2 float64 normalized image coordinates (features_data.points[:, :2]).
 * ------------------------------------------------------------------------------------------ */
int osfm_store_create(osfm_ctx *ctx, int n_images, const int32_t *counts, osfm_store **out);
int osfm_store_upload_f32(osfm_store *s, const float *desc, const double *pts);
int osfm_store_upload_u8(osfm_store *s, const uint8_t *desc, const double *pts);
/* Binary descriptors -- uint8 bit strings (AKAZE MLDB: 61 bytes, ORB: 32 bytes; width_bytes 1..64).  match_brute_force hands uint8
 * arrays to cv2's "BruteForce-Hamming" matcher (opensfm/matching.py:737-740): integer Hamming distances, the same K = 2 insertion and
 * Lowe's test on float32(int).  Instead of osfm_store_upload_f32/_u8; the store then runs osfm_match_pairs / _calibrated on the
 * Hamming kernel (VALU popcount), gates and robust stage unchanged.  Not with OSFM_MATCH_SQUARED_RATIO, guided matching or a
 * segmentation column (OSFM_E_UNSUPPORTED). */
int osfm_store_upload_binary(osfm_store *s, const uint8_t *desc, int width_bytes, const double *pts);
/* matching_use_segmentation (opensfm/feature_loading.py:118-155, matching.py:281,356): the reference appends a 129th column to the
 * HAHOG uchar descriptors, SEGMENTATION_IN_DESCRIPTOR_MULT (35) x the feature's segmentation label.  column129: sum(counts) floats,
 * that column (already multiplied).  A store that has one is matched with d^2 = d^2_128 + (c1 - c2)^2, added in float32 exactly where
 * cv2's normL2Sqr_ adds a trailing element; such stores run on the exact (VALU) kernel, not on the int8 matrix path.
 * Integer-valued stores only (the reference raises for anything but HAHOG uchar): OSFM_E_UNSUPPORTED otherwise. */
int osfm_store_set_segmentation(osfm_store *s, const float *column129);
void osfm_store_destroy(osfm_store *s);
int64_t osfm_store_bytes(const osfm_store *s); /* device bytes held */

typedef struct {
  double lowes_ratio;                /* config["lowes_ratio"], 0.8            (config.py:97)  */
  int32_t symmetric;                 /* config["symmetric_matching"]           (config.py:101) */
  int32_t robust;                    /* 1: run the geometric stage (matching.py:599-603)        */
  int32_t robust_matching_min_match; /* 20                                     (config.py:195) */
  double robust_matching_threshold;  /* 0.004                                  (config.py:191) */
  double ransac_confidence;          /* 0.9999 (matching.py:795)                               */
  int32_t ransac_max_iters;          /* 1000 (cv2.findFundamentalMat default)                  */
  int32_t flags;                     /* OSFM_MATCH_* bits below                                 */
} osfm_match_params;
#define OSFM_MATCH_EXACT_KERNEL 1  /* cross-check: run every pair on the exact VALU kernel (float keys, no MFMA; float stores:
                                      the exact float32 kernel)                                                             */
#define OSFM_MATCH_SQUARED_RATIO 2 /* matcher_type FLANN semantics on an EXACT 2-NN search: match_flann[_symmetric]
                                      (matching.py:683-720) keeps d0 < float32(lowes_ratio^2) * d1 on SQUARED float32
                                      distances; one-way matching then queries with the pair's SECOND image against the
                                      first (match_flann(index1, f2)) and lists the matches in query order.  The
                                      reference's own index (cv2.flann_Index, KMEANS, checks=20, features.py:638-674) is
                                      approximate and randomised; this is its exact limit (checks -> infinity). */
#define OSFM_MATCH_KEEP_DEVICE 4   /* the result keeps its match rows in HBM (osfm_result_dev_ptrs) instead of copying them to the
                                      host chunk by chunk: the exchange step of the multi-GPU path (all-gather of the match graph,
                                      the fan-in of matching.py:83-98 across ranks) reads them from there.  osfm_result_fetch still
                                      works (one D2H copy on demand).                                                           */

void osfm_match_params_default(osfm_match_params *p);

typedef struct {
  double ms_total;         /* stream time of the whole call (HIP events)                 */
  double ms_match_kernel;  /* sum of fused distance/top-2/ratio/mutual kernel launches   */
  double ms_ransac_kernel; /* sum of RANSAC kernel launches                              */
  int64_t match_launches;
  int64_t pairs;           /* pairs processed                                            */
  int64_t pairs_exact_path; /* pairs re-run on the exact float-key kernel (d^2 >= 2^22); float store: pairs in which at least
                               one query went through the float32 evaluation                */
  int64_t pairs_ransac;    /* calibrated branch: pairs that reached the geometric stage  */
  int64_t ransac_model_points; /* F-RANSAC: sum over pairs of (models scored) x (correspondences): the work its roofline counts */
} osfm_match_timings;

/* Opaque result of a batched run: per pair a count and a list of (i, j) int32. */
typedef struct osfm_match_result osfm_match_result;

/*
 * Batched pair matching: replaces the body of match_images_with_pairs
 * (opensfm/matching.py:63-98; per pair: match(), matching.py:563-634, BRUTEFORCE symmetric +
 * robust_match_fundamental).  pairs: n_pairs x 2 image indices into the store.
 */
int osfm_match_pairs(osfm_ctx *ctx, const osfm_store *store, const int32_t *pairs, int64_t n_pairs,
                     const osfm_match_params *params, osfm_match_result **out,
                     osfm_match_timings *timings_or_null);
int64_t osfm_result_num_pairs(const osfm_match_result *r);
int64_t osfm_result_total_matches(const osfm_match_result *r);
/* counts[n_pairs]; matches[total x 2] concatenated in pair order, each pair sorted by (i, j). */
int osfm_result_fetch(const osfm_match_result *r, int32_t *counts, int32_t *matches);
/* The same two arrays where the call left them in host memory -- no copy: valid until osfm_result_destroy, not to be written through.
 * (What a binding wraps as its array type: the reference's match_images hands out per-pair numpy arrays, matching.py:563-634; the
 * Python layer here views these buffers and destroys the result when the last view goes.)  OSFM_E_INVALID for a result kept on the
 * device (OSFM_MATCH_KEEP_DEVICE: use osfm_result_fetch or osfm_result_dev_ptrs). */
int osfm_result_host_ptrs(const osfm_match_result *r, const int32_t **counts, const int32_t **matches);
/* Results of a call made with OSFM_MATCH_KEEP_DEVICE: device pointers (on osfm_result_device) to counts[n_pairs] and to the
 * concatenated matches[total x 2], valid until osfm_result_destroy; every kernel that wrote them has completed when the matching
 * call returns.  OSFM_E_INVALID for a result that was not kept on the device. */
int osfm_result_dev_ptrs(const osfm_match_result *r, const int32_t **d_counts, const int32_t **d_matches);
int osfm_result_device(const osfm_match_result *r);
void osfm_result_destroy(osfm_match_result *r);

/*
 * Leaf: one pair from host buffers.  Drop-in for match_brute_force (symmetric = 0,
 * matching.py:723-756) and match_brute_force_symmetric (symmetric = 1, matching.py:759-777);
 * osfm_match_l2_ratio_ex with flags = OSFM_MATCH_SQUARED_RATIO for match_flann / match_flann_symmetric
 * (matching.py:683-720) on an exact search.
 * A: nA x dim, B: nB x dim float32 (dim must be 128; integer-valued or not, see the descriptor store).
 * out_pairs: cap x 2 int32, *out_n = number found (may exceed cap; only cap are written).
 */
int osfm_match_l2_ratio(osfm_ctx *ctx, const float *A, int nA, const float *B, int nB, int dim,
                        double ratio, int symmetric, int32_t *out_pairs, int cap, int *out_n);
int osfm_match_l2_ratio_ex(osfm_ctx *ctx, const float *A, int nA, const float *B, int nB, int dim,
                           double ratio, int symmetric, int flags, int32_t *out_pairs, int cap, int *out_n);
/* The same leaf for uint8 bit strings: match_brute_force[_symmetric] on uint8 arrays = cv2 BruteForce-Hamming (matching.py:737-740).
 * A: nA x width_bytes, B: nB x width_bytes (1..64 bytes). */
int osfm_match_hamming_ratio(osfm_ctx *ctx, const uint8_t *A, int nA, const uint8_t *B, int nB, int width_bytes,
                             double ratio, int symmetric, int32_t *out_pairs, int cap, int *out_n);
/* ... with flags: OSFM_MATCH_SQUARED_RATIO = match_flann / match_flann_symmetric on bit strings (matching.py:683-720; features.py:660-667 builds
 * cv2's LSH index for uint8 descriptors): searched exactly here, the test `d0 < lowes_ratio ** 2 * d1` in doubles on the int Hamming distances,
 * one-way matching queries with B (round 6) */
int osfm_match_hamming_ratio_ex(osfm_ctx *ctx, const uint8_t *A, int nA, const uint8_t *B, int nB, int width_bytes,
                                double ratio, int symmetric, int flags, int32_t *out_pairs, int cap, int *out_n);

/*
 * Leaf: cv2.findFundamentalMat(p1, p2, FM_RANSAC, thr, conf) as used by
 * robust_match_fundamental (matching.py:780-802).  p1, p2: n x 2 float64.
 * Returns OSFM_OK with *found = 1 (F row-major, mask n bytes) or *found = 0 (F is None).
 * n >= 15: RANSAC; 8 <= n < 15: the LMedS registrator cv2 switches to; n == 7: OSFM_E_UNSUPPORTED
 * (the reference never calls with fewer than 8 matches, matching.py:787).
 */
int osfm_ransac_fundamental(osfm_ctx *ctx, const double *p1, const double *p2, int n, double thr,
                            double conf, int max_iters, double F[9], uint8_t *mask, int *found,
                            int *iters_run_or_null);

/* ==========================================================================================
 * Bundle adjustment.  Replaces pysfm.BAHelpers.bundle (opensfm/src/sfm/src/ba_helpers.cc:581-763,
 * python seam opensfm/reconstruction.py:69-86) == bundle::BundleAdjuster::Run
 * (opensfm/src/bundle/src/bundle_adjuster.cc:595-1121) for the residual families BAHelpers::Bundle
 * adds on a plain reconstruction: reprojection errors with a shared robust loss
 * (projection_errors.h:59-208), camera-intrinsics priors (bundle_adjuster.cc:568-593) and
 * position priors on the shot origins (bundle_adjuster.cc:745-778, identity bias).
 * Flat SoA problem; all arrays are host pointers, in/out arrays are overwritten with the optimum.
 * ========================================================================================== */
#define OSFM_LOSS_TRIVIAL 0  /* "TrivialLoss"  (bundle_adjuster.cc:414-429) */
#define OSFM_LOSS_SOFTLONE 1 /* "SoftLOneLoss" -- the reference default (config.py:241) */
#define OSFM_LOSS_HUBER 2    /* "HuberLoss"    */
#define OSFM_LOSS_CAUCHY 3   /* "CauchyLoss"   */

#define OSFM_CAMERA_PERSPECTIVE 0 /* "perspective" */
#define OSFM_CAMERA_FISHEYE 1     /* "fisheye"     */
/* the other 2-D projection types: as CONSTANT cameras only (cam_fixed = 1 + cam_ext), which is how
   BundleLocal / BundleShotPoses always use cameras (ba_helpers.cc:137,415) */
#define OSFM_CAMERA_BROWN 2          /* [k1 k2 k3 p1 p2 | focal ar cx cy]                    */
#define OSFM_CAMERA_FISHEYE_OPENCV 3 /* [k1 k2 k3 k4 | focal ar cx cy]                       */
#define OSFM_CAMERA_FISHEYE62 4      /* [k1..k6 p1 p2 | focal ar cx cy]                      */
#define OSFM_CAMERA_FISHEYE624 5     /* [k1..k6 p1 p2 s0 s1 s2 s3 | focal ar cx cy]          */
#define OSFM_CAMERA_DUAL 6           /* [transition | k1 k2 | focal]                         */
#define OSFM_CAMERA_RADIAL 7         /* [k1 k2 | focal ar cx cy]                             */
#define OSFM_CAMERA_SIMPLE_RADIAL 8  /* [k1 | focal ar cx cy]                                */
#define OSFM_CAMERA_SPHERICAL 9      /* no parameters; bearings only (osfm_pixel_bearings), not a BA camera model */

typedef struct {
  int32_t n_cameras, n_shots, n_points;
  int64_t n_obs;
  double *cam_params;        /* n_cameras x 3: k1, k2, focal (geometry/src/camera.cc:9-17)   in/out */
  const double *cam_prior;   /* n_cameras x 3                                                       */
  const double *cam_sigma;   /* n_cameras x 3: prior sd of k1, k2, focal (focal: log-ratio)         */
  const uint8_t *cam_fixed;  /* n_cameras: 1 = constant (optimize_camera_parameters False)          */
  double *shot_pose;         /* n_shots x 6: rx ry rz tx ty tz (bundle/data/pose.h:17,34-43) in/out */
  const int32_t *shot_camera;    /* n_shots                                                         */
  const uint8_t *shot_fixed;     /* n_shots or NULL                                                 */
  const double *shot_gps;        /* n_shots x 3 or NULL: prior on the origin                        */
  const double *shot_gps_sigma;  /* n_shots or NULL: sd (<= 0: no prior for that shot)              */
  double *points;                /* n_points x 3                                             in/out */
  const uint8_t *point_fixed;    /* n_points or NULL                                                */
  const int32_t *obs_shot;       /* n_obs                                                           */
  const int32_t *obs_point;      /* n_obs                                                           */
  const double *obs_xy;          /* n_obs x 2 normalized image coordinates (observation.point)     */
  const double *obs_sigma;       /* n_obs: observation.scale == std_deviation (tracking.py:108)     */
  double *reproj_err;            /* n_obs x 2 or NULL: out, residual with sigma 1                   */
  /* absolute up-vector prior (BAHelpers::Bundle with align_method orientation_prior,
     ba_helpers.cc:609-621,688-692; UpVectorError, absolute_motion_errors.h:12-39; CauchyLoss(1),
     bundle_adjuster.cc:955-970): residual (R(rotation) * up - e_z) / sd per shot */
  const double *shot_up;         /* n_shots x 3 or NULL (normalised by the library)                 */
  const double *shot_up_sigma;   /* n_shots or NULL: sd (<= 0: no prior for that shot)              */
  /* projection type per camera (geometry::ProjectionType, camera_instances.h:8-20,183-190), or NULL
     = all PERSPECTIVE.  Both supported types carry the same parameters [k1, k2, focal]:
     PerspectiveCamera = <PerspectiveProjection, Disto24, UniformScale>,
     FisheyeCamera     = <FisheyeProjection,     Disto24, UniformScale>. */
  const int32_t *cam_model;      /* n_cameras or NULL: OSFM_CAMERA_*                                */
  /* native parameters [projection][distortion][affine] (camera_instances.h:127-160), 16 per camera,
     of the constant cameras whose cam_model is >= 2; NULL when there are none */
  const double *cam_ext;         /* n_cameras x 16 or NULL                                          */
                                 /* (ComputeReprojectionErrors, bundle_adjuster.cc:1196-1208)        */
} osfm_ba_problem;

typedef struct {
  int32_t loss;               /* OSFM_LOSS_*                 config["loss_function"]           */
  double loss_threshold;      /* config["loss_function_threshold"] = 1 (config.py:243)        */
  int32_t max_iterations;     /* config["bundle_max_iterations"] = 100 (config.py:283)        */
  double function_tolerance;  /* ceres default 1e-6                                           */
  double gradient_tolerance;  /* ceres default 1e-10                                          */
  double parameter_tolerance; /* ceres default 1e-8                                           */
  double initial_radius;      /* ceres initial_trust_region_radius 1e4                        */
  int32_t verbose;            /* bit 0: one progress line per LM iteration on stderr; OSFM_BA_TIME_MATVEC: ten extra Schur
                                 mat-vecs are timed with HIP events after the solve (report: ms_matvec_total, matvec_calls) */
  double pcg_tolerance;       /* relative residual of the Schur-PCG solve (default 1e-10)     */
  int32_t pcg_max_iterations; /* default 1000                                                 */
  int32_t preconditioner;     /* 0 auto: banded block Cholesky when the shot coupling is banded
                                 (half-width <= 15 shots), else block Jacobi; 1: block Jacobi  */
  double pcg_direct_tolerance; /* relative residual at which the FIRST iterate is accepted when the preconditioner is the reduced matrix
                                * itself (exact band by cyclic reduction + exact border, or constant cameras): that iterate is a direct solve
                                * -- what Ceres' SPARSE_SCHUR stops at -- plus a line search, and lands at 1e-10 .. 1e-7 (conditioning x the
                                * rounding of the explicit block inverses).  Default 1e-6: trajectories at 1e-6 and 1e-10 agree to 4e-16 in
                                * the cost over 20 iterations at configs[4] (profiles/r06_pcg_tolerance.json).  <= pcg_tolerance: no such
                                * rule.  Inexact preconditioners (block Jacobi, a truncated band, a border too wide for the exact elimination)
                                * always iterate to pcg_tolerance. */
} osfm_ba_options;
#define OSFM_BA_TIME_MATVEC 2

void osfm_ba_options_default(osfm_ba_options *o);

typedef struct {
  int32_t iterations;       /* LM iterations (successful + unsuccessful), ceres' num_iterations - 1 */
  int32_t successful_steps;
  int32_t termination;      /* 0 max iterations, 1 function tol, 2 gradient tol, 3 parameter tol,
                               4 min trust region radius, -1 failure */
  double initial_cost, final_cost;
  double rmse_normalized_initial, rmse_normalized_final; /* sqrt(mean |pi(X) - obs|^2), x max(w,h) = px */
  double seconds_total, seconds_linear_solver;           /* wall_times (ba_helpers.cc:749-753)          */
  double cost_history[256];
  int64_t pcg_iterations_total;
  double ms_matvec_total;    /* HIP-event time spent in the Schur mat-vec kernels */
  int64_t matvec_calls;
  int32_t shot_bandwidth;           /* max |shot_a - shot_b| over shots sharing a point          */
  int32_t preconditioner_bandwidth; /* block half-width of the banded preconditioner, 0 = Jacobi */
  /* wall_times of BAHelpers::Bundle (ba_helpers.cc:749-753): setup = index build + H2D,
     run = the LM loop (what ceres::Solve covers), teardown = D2H of parameters and errors */
  double seconds_setup, seconds_run, seconds_teardown;
  /* 1: the shots were renumbered internally (reverse Cuthill-McKee on the co-visibility graph) because
     the caller's order had a half-width above what the banded preconditioner holds; shot_bandwidth is
     then the half-width after renumbering and shot_bandwidth_input the caller's. */
  int32_t shots_reordered, shot_bandwidth_input;
} osfm_ba_report;

int osfm_ba_solve(osfm_ctx *ctx, osfm_ba_problem *problem, const osfm_ba_options *options,
                  osfm_ba_report *report);

/* Host-only (no GPU): the shot renumbering osfm_ba_solve applies to unordered collections (reverse
   Cuthill-McKee on the co-visibility graph) and the co-visibility half-width before / after it. */
int osfm_ba_shot_order(const osfm_ba_problem *problem, int32_t *new_of_old, int32_t *half_width_before,
                       int32_t *half_width_after);

/* ------------------------------------------------------------------------------------------
 * The general bundle adjustment: everything BAHelpers::Bundle (ba_helpers.cc:581-763) hands to bundle::BundleAdjuster and Run
 * (bundle_adjuster.cc:595-1121) solves -- every projection type with all of its native parameters free or constant
 * (camera_instances.h, AddCameraPriorError bundle_adjuster.cc:568-593 with logarithmic focal / aspect ratio, the dual camera's
 * ParameterBarrier), the SPHERICAL 3-D bearing residual (projection_errors.h:208-376), rig cameras with pose priors and rig
 * instances (error_utils.h:68-85 WorldToCameraCoordinatesRig), rig instance position priors through a per-camera GPS bias
 * (SimilarityPriorTransform, bias.h:33-53, bundle_adjuster.cc:745-778), point priors = ground control points
 * (bundle_adjuster.cc:688-707, ba_helpers.cc:349-406), absolute up vectors (absolute_motion_errors.h:12-39, Cauchy(1)), compass /
 * inclinometer priors, depth priors.
 * Round 5: this entry point runs on the SAME streaming Schur solver as osfm_ba_solve (its generic mode, csrc/ba_generic.inc): points
 * eliminated on the fly, the rig instances in a co-visibility band factorised exactly (cyclic reduction), the free rig cameras /
 * camera intrinsics / biases as a border of up to 64 unknowns eliminated exactly (wider borders: preconditioned CG) -- time and
 * memory linear in the observations, BASELINE configs[4] size with a BROWN camera in one call.  (Round 4 formed the reduced system
 * densely and factorised it with rocSOLVER: n_r^2 doubles.)  osfm_ba_solve above remains the specialisation for the perspective /
 * fisheye [k1 k2 focal] configuration the headline BA figure is quoted on.
 * Poses are CAM_TO_WORLD [rx ry rz tx ty tz] (bundle/data/pose.h:17,34-43); camera parameters in the native order of the
 * OSFM_CAMERA_* definitions, 16 slots per camera; in/out arrays are overwritten with the optimum.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  int32_t n_cameras;
  const int32_t *cam_model;        /* n_cameras: OSFM_CAMERA_*                                                         */
  double *cam_params;              /* n_cameras x 16                                                            in/out */
  const double *cam_prior;         /* n_cameras x 16                                                                   */
  const double *cam_sigma;         /* n_cameras x 16: GetDefaultCameraSigma (bundle_adjuster.cc:47-69)                 */
  const uint8_t *cam_fixed;        /* n_cameras                                                                        */
  double *bias;                    /* n_cameras x 7 [rx ry rz tx ty tz scale] (bias.h:10-31) or NULL = identity  in/out */
  const uint8_t *bias_fixed;       /* n_cameras or NULL = all constant (AddCamera's default, bundle_adjuster.cc:98-106) */
  int32_t n_rig_cameras;
  double *rig_camera_pose;         /* n_rig_cameras x 6                                                          in/out */
  const double *rig_camera_prior;  /* n_rig_cameras x 6 or NULL                                                         */
  const double *rig_camera_sigma;  /* n_rig_cameras x 6 or NULL (GetDefaultRigPoseSigma)                                */
  const uint8_t *rig_camera_fixed; /* n_rig_cameras                                                                     */
  int32_t n_rig_instances;
  double *rig_instance_pose;       /* n_rig_instances x 6                                                        in/out */
  const uint8_t *rig_instance_fixed;       /* n_rig_instances or NULL                                                   */
  const double *rig_instance_gps;          /* n_rig_instances x 3 or NULL: AddRigInstancePositionPrior                  */
  const double *rig_instance_gps_sigma;    /* n_rig_instances x 3; [3 i] <= 0: no prior for instance i                  */
  const int32_t *rig_instance_bias_camera; /* n_rig_instances: the camera whose bias applies (its first shot's camera,
                                              bundle_adjuster.cc:757-764)                                               */
  int32_t n_shots;
  const int32_t *shot_rig_instance, *shot_rig_camera, *shot_camera; /* n_shots each                                     */
  const double *shot_up;           /* n_shots x 3 or NULL: AddAbsoluteUpVector                                          */
  const double *shot_up_sigma;     /* n_shots; <= 0: none                                                               */
  int32_t n_points;
  double *points;                  /* n_points x 3                                                               in/out */
  const uint8_t *point_fixed;      /* n_points or NULL                                                                  */
  const double *point_prior;       /* n_points x 3 or NULL: AddPointPrior                                               */
  const double *point_prior_sigma; /* n_points x 3; [3 p] <= 0: no prior for point p                                    */
  const uint8_t *point_prior_has_altitude; /* n_points or NULL = all 1: 0 constrains x, y only                          */
  int64_t n_obs;
  const int32_t *obs_shot, *obs_point;
  const double *obs_xy;            /* n_obs x 2 normalized image coordinates                                            */
  const double *obs_sigma;         /* n_obs                                                                             */
  double *reproj_err;              /* n_obs x 3 or NULL: out, residual with sigma 1 (third component: spherical only)   */
  /* compass / inclinometer priors per shot (AddAbsolutePan / Tilt / Roll, absolute_motion_errors.h:40-137, Cauchy(1)):
     angle in radians and its sd (<= 0: none); NULL = none at all */
  const double *shot_pan, *shot_pan_sigma, *shot_tilt, *shot_tilt_sigma, *shot_roll, *shot_roll_sigma;
  /* depth priors of the observations (map::Depth, observation.h:10-18; AddPointProjectionObservation's depth_prior argument as
     BAHelpers passes it, ba_helpers.cc:695-696): RelativeDepthError (bundle/error/relative_depth_error.h, bundle_adjuster.cc:497-528),
     one residual (depth_in_camera - depth) / sd per observation that has one, depth_in_camera = |X_cam| (radial) or its z, under the
     point-projection loss.  n_obs each; obs_depth_sigma[o] <= 0: none; obs_depth / obs_depth_sigma NULL: none at all;
     obs_depth_radial NULL: all radial.  A non-finite depth with a positive sd is OSFM_E_INVALID (the reference throws). */
  const double *obs_depth, *obs_depth_sigma;
  const uint8_t *obs_depth_radial;
} osfm_bundle_problem;

int osfm_bundle_solve(osfm_ctx *ctx, osfm_bundle_problem *problem, const osfm_ba_options *options, osfm_ba_report *report);

/* =====================================================================================
 * Tracks (next row after the hot path: SURVEY.md 8f-1)
 * Replaces the grouping of tracking.create_tracks_manager (opensfm/tracking.py:68-98, union-find
 * opensfm/unionfind.py:67-103, _good_track tracking.py:238-244): links every match
 * (im1, f1) -- (im2, f2) into connected components, lists the components in the order of their
 * first-inserted member, members in insertion order, keeps those with >= min_length members and
 * no image twice; track_id = index in that list.
 * edge_a/edge_b: global node ids in the reference's union order (pair by pair, match by match);
 * node id of feature f of image i = node_offsets[i] + f; node_offsets has n_images + 1 entries.
 * ===================================================================================== */
typedef struct osfm_tracks osfm_tracks;
int osfm_tracks_create(osfm_ctx *ctx, const int32_t *edge_a, const int32_t *edge_b, int64_t n_edges,
                       const int64_t *node_offsets, int32_t n_images, int32_t min_length, osfm_tracks **out);
int64_t osfm_tracks_num_tracks(const osfm_tracks *t);
int64_t osfm_tracks_num_observations(const osfm_tracks *t);
double osfm_tracks_device_ms(const osfm_tracks *t); /* HIP-event time of the device part */
/* each n_observations long, grouped by track in track order, members in insertion order */
int osfm_tracks_fetch(const osfm_tracks *t, int32_t *obs_track, int32_t *obs_image, int32_t *obs_feature);
void osfm_tracks_destroy(osfm_tracks *t);

/* =====================================================================================
 * Calibrated robust matching (row M-a9 / SURVEY.md 8f-3): essential-matrix LO-RANSAC on bearings.
 * Organisation (round 2, relpose.hip / relpose_rounds.h): all pairs of a call go through rounds of kernels together -- walk (one
 * wavefront per pair: scoring, the reference's decision rules, the draws), solve5a / solve5b / solveN (one lane per minimal /
 * non-minimal problem), pose (one lane per essential matrix) -- then one launch of the refinement stage.  The numerics and the
 * round logic are pinned bit for bit against the CPU oracle by a host emulation (tests/test_relpose_core_host.py); on the MI355X
 * the RANSAC stage is bit-identical to the oracle and the inlier sets after the refinement are identical
 * (tests/test_gpu_zz_relpose.py).  128 k pairs / s at 300 correspondences per pair in 16 k-pair calls, 180 k in 64 k-pair calls
 * (profiles/r02_relpose_*.json).
 *
 * osfm_pixel_bearings  replaces camera.pixel_bearing_many(points) (opensfm/src/geometry/camera.cc ->
 *   ProjectGeneric::Backward, camera_instances.h:154-160) for every OSFM_CAMERA_* model; cam = the model's
 *   parameters in the native order listed at the OSFM_CAMERA_* definitions ([k1, k2, focal] for PERSPECTIVE /
 *   FISHEYE, may be NULL for SPHERICAL); px: n x 2 normalised image coordinates, bearings: n x 3.
 *   (MI355X vs oracle, profiles/r01_guided_bringup.txt: bit-identical for the perspective-projection models, <= 2.3e-16
 *   for those that go through the device sin / cos / tan.)
 * osfm_relpose_pairs   a batch of pairs; pair p owns the correspondences offsets[p] .. offsets[p+1]-1 of the
 *   concatenated bearing arrays b1, b2 (total x 3, doubles, second-image bearing y and first-image bearing x
 *   with y ~ R x + t for the models below).
 *   mode OSFM_RELPOSE_RANSAC: pyrobust.ransac_relative_pose(b1, b2, threshold, params, RANSAC)
 *     (opensfm/src/robust/src/instanciations.cc:33-48, robust_estimator.h:37-119): result.model / lo_model
 *     (3 x 4 row-major), score, iterations; mask = inliers of the best score.
 *   mode OSFM_RELPOSE_MATCH: the body of matching.robust_match_calibrated after the bearings
 *     (opensfm/matching.py:886-903): RANSAC, three rounds of compute_inliers_bearings (4, 2, 1 x threshold)
 *     + relative_pose_refinement, final compute_inliers_bearings; mask = the inliers the reference keeps
 *     (all zero where it returns an empty array), R / t = pose of the second camera in the first.
 *   The sampler is std::mt19937(42) with std::uniform_int_distribution as libstdc++ >= 11 implements it (Lemire's
 *   multiply-shift); ties between scores keep the newcomer (std::max).  Both are pinned against the reference's own
 *   robust_estimator.h / random_sampler.h compiled on the build box (oracle/_ref).
 * ===================================================================================== */
enum { OSFM_RELPOSE_RANSAC = 0, OSFM_RELPOSE_MATCH = 1 };
typedef struct osfm_relpose_params {
  double threshold;          /* radians: config robust_matching_calib_threshold (0.004) */
  double probability;        /* RobustEstimatorParams::probability (0.99; the Python caller never sets it) */
  int32_t iterations;        /* 1000 (matching.py:889) */
  int32_t use_lo;            /* RobustEstimatorParams::use_local_optimization (1) */
  int32_t lo_iterations;     /* ::local_optimization_iterations (10) */
  int32_t refine_iterations; /* config five_point_refine_match_iterations (10) */
} osfm_relpose_params;
typedef struct osfm_relpose_result {
  double model[12], lo_model[12]; /* ScoreInfo::model / lo_model, [R | t] row-major */
  double R[9], t[3];              /* MATCH mode: after the last refinement (zeros when rejected) */
  int32_t score, iterations;      /* best inlier count of the RANSAC, iterations it ran */
  int32_t n_inliers, pad;         /* number of ones in this pair's mask */
} osfm_relpose_result;
int osfm_pixel_bearings(osfm_ctx *ctx, int model, const double *cam, const double *px, int n, double *bearings);
int osfm_relpose_pairs(osfm_ctx *ctx, const double *b1, const double *b2, const int64_t *offsets, int n_pairs,
                       const osfm_relpose_params *params, int mode, osfm_relpose_result *results, uint8_t *mask,
                       double *kernel_ms /* may be NULL: HIP-event time of the kernels */);

/*
 * Batched pair matching for the pairs that take the calibrated branch of robust_match (opensfm/matching.py:906-929: every pair with
 * a camera that is not an undistorted perspective / brown one): per pair matching.match (matching.py:563-634) = descriptor stage,
 * the robust_matching_min_match gate, robust_match_calibrated (matching.py:871-903) on the bearings of the matched features, the
 * gate again.  cam_model[n_images] (OSFM_CAMERA_*) and cam_params[n_images x 16] (native parameter order, as osfm_pixel_bearings)
 * describe the camera of every image of the store.  Everything between the two stages stays on the device: bearings of all features
 * once per call, gather of the matched bearings, the LO-RANSAC / refinement rounds, ordered compaction of the inliers.
 * Same result object as osfm_match_pairs.
 */
int osfm_match_pairs_calibrated(osfm_ctx *ctx, const osfm_store *store, const int32_t *cam_model, const double *cam_params,
                                const int32_t *pairs, int64_t n_pairs, const osfm_match_params *params,
                                const osfm_relpose_params *relpose, osfm_match_result **out, osfm_match_timings *timings_or_null);

/* =====================================================================================
 * Masked / guided descriptor matching (second half of row M-a9 / SURVEY.md 8f-3).
 * Drop-in for match_brute_force(f1, f2, config, maskij) (symmetric = 0, opensfm/matching.py:723-756) and
 * match_brute_force_symmetric(fi, fj, config, maskij) (symmetric = 1, matching.py:759-777: the reverse direction uses
 * the transposed mask).  The mask is EITHER explicit (mask: n1 x n2 bytes, non-zero = allowed) OR, with mask == NULL,
 * the epipolar mask of guided matching evaluated on the fly (never materialised):
 *   compute_inliers_bearing_epipolar(b1, b2, pose, threshold) (matching.py:847-868 ->
 *   geometry::EpipolarAngleTwoBearingsMany, opensfm/src/geometry/src/triangulation.cc:195-219) with b1 / b2 the float32
 *   bearings of the two images, R = pose.get_R_cam_to_world() (row-major), t = pose.get_origin() of the relative pose
 *   (pose2.relative_to(pose1), matching.py:204-207), threshold = config guided_matching_threshold (radians).
 * f1: n1 x dim, f2: n2 x dim float32 (integer-valued in [0, 255], dim must be 128); out_pairs: cap x 2 int32 sorted by
 * (i, j); *out_n = number found (may exceed cap; only cap are written).
 * This single-pair leaf takes integer-valued descriptors only (it is what the Python leaves match_brute_force[_symmetric](maskij)
 * call); float descriptors go through the batched entry point below, whose store keeps their float rows.
 * ===================================================================================== */
int osfm_match_guided(osfm_ctx *ctx, const float *f1, int n1, const float *f2, int n2, int dim, const uint8_t *mask,
                      const float *b1, const float *b2, const double *R, const double *t, double threshold, double ratio,
                      int symmetric, int32_t *out_pairs, int cap, int *out_n);

/*
 * Guided matching for a whole pair list over the resident store: the body of match_images_with_pairs when `poses` are given
 * (opensfm/matching.py:63-98 -> match_unwrap_args :204-207 -> match() :563-634 with _match_descriptors_guided_impl :260-337):
 * per pair the epipolar mask compute_inliers_bearing_epipolar (:847-868), match_brute_force[_symmetric] under that mask, the
 * robust_matching_min_match gates and robust_match -- descriptor stage in three launches per chunk of pairs (guided.hip), robust
 * stage as in osfm_match_pairs (params->robust: fundamental-matrix RANSAC) or osfm_match_pairs_calibrated (cam_model / cam_params /
 * relpose given; otherwise pass all three as NULL).
 * bearings: sum(counts) x 3 float32 in the store's image / feature order (feature_loader.load_bearings);
 * poses:    n_pairs x 12, per pair R = relative_pose.get_R_cam_to_world() (row-major) then t = relative_pose.get_origin(),
 *           relative_pose = pose2.relative_to(pose1) (matching.py:204-207);
 * threshold: config guided_matching_threshold (radians).
 * Integer-valued stores rank the candidates by the exact integer distance; float stores (root-SIFT, feature_root) by the float32
 * distance in the oracle's accumulation order (oracle/guided_oracle.c l2sqr), read from the store's float rows.
 */
int osfm_match_pairs_guided(osfm_ctx *ctx, const osfm_store *store, const float *bearings, const int32_t *pairs, int64_t n_pairs,
                            const double *poses, double threshold, const osfm_match_params *params, const int32_t *cam_model_or_null,
                            const double *cam_params_or_null, const osfm_relpose_params *relpose_or_null, osfm_match_result **out,
                            osfm_match_timings *timings_or_null);

/* =====================================================================================
 * Bag-of-words side of pair matching and pair preselection.
 *
 * osfm_words_store / osfm_match_words_pairs  replace pyfeatures.match_using_words (opensfm/src/features/src/matching.cc:24-88) as
 *   matching.match_words / match_words_symmetric call it (opensfm/matching.py:637-680), for a whole pair list: descriptors
 *   (float32, total x 128) and the n closest vocabulary words of every feature (int32, total x words_per_feature, data.load_words)
 *   are uploaded once; per pair, every feature of one image checks the features of the other image filed under its words, in word
 *   order, stopping after the word that brings the count to max_checks; Lowe's test best < ratio * second in float; symmetric = the
 *   intersection of both directions.  counts[p] matches (i, j) of pair p land in matches[(p * max_count + k) * 2 ..], ordered by i
 *   (the reference returns them in set order, i.e. unspecified), max_count = osfm_words_store_max_count.
 * osfm_vlad_descriptor  replaces pyfeatures.compute_vlad_descriptor (matching.cc:93-124; the caller normalises, vlad.py);
 * osfm_vlad_distances   replaces the distance loop of compute_vlad_distances (matching.cc:126-152): L2 norms of reference - others[j];
 * osfm_knn_points / osfm_radius_points  replace spatial.cKDTree(points).query(point, k, distance_upper_bound) of
 *   match_candidates_by_distance (opensfm/pairs_selection.py:188-212): the k nearest candidates within max_distance in ascending
 *   distance (missing: index -1, distance inf), or -- when k covers every candidate -- the bit mask of the candidates within range.
 * ===================================================================================== */
typedef struct osfm_words_store osfm_words_store;
int osfm_words_store_create(osfm_ctx *ctx, int n_images, const int32_t *counts, int dim, int words_per_feature, const float *desc,
                            const int32_t *words, osfm_words_store **out);
void osfm_words_store_destroy(osfm_words_store *s);
int osfm_words_store_max_count(const osfm_words_store *s);
int osfm_match_words_pairs(osfm_ctx *ctx, const osfm_words_store *store, const int32_t *pairs, int64_t n_pairs, float lowes_ratio,
                           int max_checks, int symmetric, int32_t *counts, int32_t *matches, double *kernel_ms);
int osfm_vlad_descriptor(osfm_ctx *ctx, const float *features, int n, const float *centers, int n_centers, int dim, float *out);
int osfm_vlad_distances(osfm_ctx *ctx, const float *reference, const float *others, int m, int len, double *out);
/* pairs_selection.bow_distances (opensfm/pairs_selection.py:690-708): out[c] = np.fabs(reference - others[c]).sum() over float64 BoW
 * histograms (bow.py:34-36), in numpy's summation order (pairwise blocks of 128 with eight accumulators, 8192-element chunks). */
int osfm_bow_distances(osfm_ctx *ctx, const double *reference, const double *others, int m, int len, double *out);
int osfm_knn_points(osfm_ctx *ctx, const double *candidates, int n_candidates, const double *queries, int n_queries, int k,
                    double max_distance, double *out_distance, int32_t *out_index);
int osfm_radius_points(osfm_ctx *ctx, const double *candidates, int n_candidates, const double *queries, int n_queries,
                       double max_distance, uint32_t *out_mask);

/* =====================================================================================
 * HAHOG feature extraction
 *
 * osfm_hahog_extract replaces pyfeatures.hahog(image, peak_threshold, edge_threshold, target_num_features)
 * (opensfm/src/features/src/hahog.cc:125-206; binding opensfm/src/features/python/pybind.cc) together with the descriptor
 * post-processing of its only caller, features.extract_features_hahog (opensfm/features.py:516-534):
 *   image     rows x cols float32 grey levels in [0, 1] (host), at least 17 x 17
 *   points    n x 4: x, y, size, angle in degrees (hahog.cc:170-177);  desc  n x 128
 *   flags     OSFM_HAHOG_ROOT: square roots (feature_root);  OSFM_HAHOG_UCHAR: x 362 (x 512 without ROOT), clipped to [0, 255] and
 *             rounded (hahog_normalize_to_uchar) -- the integer-valued descriptors the matcher takes on its int8 path.  0: vlfeat's.
 *   capacity  rows available in points / desc; 4 x target_num_features always suffices (at most four orientations per feature).
 * Vlfeat's Hessian detector (covdet.c), the selection of the strongest features, orientations and SIFT descriptors (sift.c) are
 * followed operation for operation: the detected set, keypoints and descriptors are the reference's (tests/test_gpu_hahog.py states
 * the tolerance and what can differ).  target_num_features = 0 returns no features, as the reference does (hahog.cc:24-27 keeps
 * `target` of the sorted list).  OSFM_E_INVALID when the features do not fit `capacity` (*n_features says how many there are).
 * ===================================================================================== */
#define OSFM_HAHOG_ROOT 1
#define OSFM_HAHOG_UCHAR 2
int osfm_hahog_extract(osfm_ctx *ctx, const float *image, int rows, int cols, float peak_threshold, float edge_threshold,
                       int target_num_features, int flags, float *points, float *desc, int capacity, int *n_features);
/* The same for a list of images (what `opensfm detect_features` runs over a data set, one image per process in the reference:
 * opensfm/actions/detect_features.py -> features_processing.run_features_processing): up to `concurrency` (0: 4; at most 16) images in
 * flight, each on its own HIP stream and host thread, so that the ~130 short launches and the two host round trips of one image run
 * underneath the kernels of the others.  images[i] is rows[i] x cols[i]; points[i] / desc[i] have capacities[i] rows; n_features[i] as
 * above.  Results are those of osfm_hahog_extract image by image.  OSFM_HAHOG_IMAGE_ON_DEVICE: the image pointers are device memory
 * (extraction from frames that are already resident, no host-to-device copy inside the call). */
#define OSFM_HAHOG_IMAGE_ON_DEVICE 4
/* OSFM_HAHOG_IMAGE_U8 (both entry points): the image pointers address uint8 grey levels 0 .. 255 (rows x cols bytes) instead of float32
 * in [0, 1]; the library forms level / 255 in float32 on the device -- what features.extract_features_hahog (opensfm/features.py:524)
 * does on the host before the call: identical features, a quarter of the bytes over PCIe and no host-side conversion pass. */
#define OSFM_HAHOG_IMAGE_U8 8
int osfm_hahog_extract_batch(osfm_ctx *ctx, int n_images, const float *const *images, const int *rows, const int *cols, float peak_threshold,
                             float edge_threshold, int target_num_features, int flags, float *const *points, float *const *desc,
                             const int *capacities, int *n_features, int concurrency);

#ifdef __cplusplus
}
#endif
#endif /* OSFM_MI355_H */

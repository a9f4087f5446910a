/*
 * o3d_oracle.h -- CPU ORACLE for the open3d_slam scan-to-map hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library.  The product
 * (open3d_slam_amd/, include/o3ds_backend.h) never links, imports or calls it.
 *
 * PARITY UNPINNED for everything whose arithmetic lives in Open3D v0.15.1
 * (registration, normals, VoxelDownSample, KD-tree; pinned by
 * /root/reference/open3d_catkin/CMakeLists.txt:116-118), which is neither
 * vendored under /root/reference nor installed in this image, and the
 * reference's own tests never touch the path, so
 * there are no golden vectors to pin those parts against.  This file restates
 * the published Open3D v0.15.1 algorithm and the
 * in-tree open3d_slam code it is called from; it is cross-checked against an
 * independent numpy/scipy restatement (oracle/np_oracle.py) and analytic
 * known-answer tests (tests/test_oracle.py).
 * PINNED (round 3) for the in-tree open3d_slam code -- croppers,
 * voxelizeWithinCroppingVolume, o3d_slam::transform, space carving, the dense
 * voxel map, voxel overlap, constant-velocity de-skew: those reference sources
 * are compiled unchanged and run (oracle/ref_build -> oracle/_ref), and this
 * file equals them bit for bit (tests/test_oracle_vs_reference.py,
 * tests/golden/ref_units.npz).
 *
 * Conventions: clouds are flat double[3*n] (the memory layout of
 * std::vector<Eigen::Vector3d>), poses are double[16] COLUMN-MAJOR (the layout
 * of Eigen::Matrix4d::data()).
 */
#ifndef O3D_ORACLE_H
#define O3D_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_kdtree orc_kdtree;

typedef struct {
  double transformation[16]; /* column-major 4x4 */
  double fitness;
  double inlier_rmse;
  int32_t iterations; /* number of Gauss-Newton updates actually applied */
  int32_t converged;  /* 1 if the relative_fitness/rmse test fired */
  uint64_t n_corr;
} orc_icp_result;

/* cropper kinds: open3d_slam/src/croppers.cpp:121-165 */
enum { ORC_CROP_NONE = 0, ORC_CROP_MAX_RADIUS = 1, ORC_CROP_MIN_RADIUS = 2, ORC_CROP_MIN_MAX_RADIUS = 3, ORC_CROP_CYLINDER = 4 };
typedef struct {
  int32_t kind;
  int32_t invert;     /* CroppingVolume::setIsInvertVolume, croppers.cpp:53-59 */
  double center[3];   /* pose_.translation(), croppers.cpp:61-63 */
  double rmin, rmax;  /* radii */
  double zmin, zmax;  /* cylinder only */
} orc_crop;

int orc_num_threads(void);
void orc_set_num_threads(int n);

/* KD-tree, leaf size 15, exact L2 (Open3D KDTreeFlann / nanoflann restatement) */
orc_kdtree* orc_kdtree_build(const double* pts, size_t n);
void orc_kdtree_free(orc_kdtree* t);
/* k nearest with d2 < radius^2 (KDTreeFlann::SearchHybrid); sorted ascending; returns count */
int orc_kdtree_search_hybrid(const orc_kdtree* t, const double q[3], double radius, int max_nn, int32_t* idx, double* d2);
int orc_kdtree_search_knn(const orc_kdtree* t, const double q[3], int k, int32_t* idx, double* d2);

/* A.2 GetRegistrationResultAndCorrespondences: corr[i] = target idx or -1 */
void orc_evaluate(const orc_kdtree* t, const double* src, size_t n, double max_corr, int32_t* corr, double* d2_out /* may be NULL */,
                  double* fitness, double* inlier_rmse, uint64_t* n_corr);
/* A.3 ComputeJTJandJTr for point-to-plane: JTJ row-major 6x6 */
void orc_compute_jtj_jtr(const double* src, size_t n, const double* tgt, const double* tgt_nrm, const int32_t* corr, double JTJ[36],
                         double JTr[6], double* r2_sum);
/* A.3 solve JTJ x = -JTr (pivoted LDLT), x -> Rz*Ry*Rx|t ; returns 0 ok. U column-major */
int orc_solve_update(const double JTJ[36], const double JTr[6], double U[16], double x_out[6] /* may be NULL */);
void orc_vector6_to_matrix4(const double x[6], double U[16]);
/* A.4 PointCloud::Transform on points (in place) and normals (3x3 only, in place) */
void orc_transform_points(double* pts, size_t n, const double T[16]);
void orc_transform_normals(double* nrm, size_t n, const double T[16]);
/* A.1 RegistrationICP with TransformationEstimationPointToPlane; tree==NULL -> build per call as the reference does */
int orc_icp_point_to_plane(const double* src, size_t n, const double* tgt, const double* tgt_nrm, size_t N, const orc_kdtree* tree,
                           double max_corr, const double init[16], int max_iter, double rel_fitness, double rel_rmse,
                           orc_icp_result* out);

/* A.3b RegistrationICP with TransformationEstimationPointToPoint (call site src/CloudRegistration.cpp:69-74):
 * Eigen::umeyama without scaling on the matched pairs (closed form, 3x3 SVD), same loop / convergence as A.1. */
int orc_icp_point_to_point(const double* src, size_t n, const double* tgt, size_t N, const orc_kdtree* tree, double max_corr,
                           const double init[16], int max_iter, double rel_fitness, double rel_rmse, orc_icp_result* out);
int orc_umeyama_update(const double* P, size_t n, const double* tgt, const int32_t* corr, double U[16]);
/* A = U diag(d) V^T (row-major 3x3), d descending */
void orc_svd3(const double A[9], double U[9], double d[3], double V[9]);

/* A.9 GetInformationMatrixFromPointClouds (call sites src/constraint_builders.cpp:70-73, src/PlaceRecognition.cpp:148-149);
 * out: row-major 6x6 (rotation block first, as Open3D: "I comes first" refers to G = [-[q]x | I]) */
int orc_information_matrix(const double* src, size_t n, const double* tgt, size_t N, const orc_kdtree* tree, double max_corr,
                           const double T[16], double out[36]);

/* Space carving of the sparse map: getIdxsOfCarvedPoints (helpers.cpp:235-271) for Submap::carve (Submap.cpp:109-125).
 * scan is already in the map frame; subset = the map indices inside the map builder's cropping volume; flags_out[N]. */
size_t orc_carve_flags(const double* scan, size_t n_scan, const double sensor[3], const double* map_pts, const double* map_nrm, size_t N,
                       const int64_t* subset, size_t n_subset, double voxel, double max_length, double truncation, double min_dot,
                       uint8_t* flags_out);

/* computeIndicesOfOverlappingPoints (helpers.cpp:307-332): flags of the source / target points that lie in voxels holding at least
 * min_points of each cloud (source placed by T first) */
void orc_overlap_flags(const double* src, size_t n_src, const double* tgt, size_t n_tgt, const double T[16], double voxel, size_t min_points,
                       uint8_t* flags_src, uint8_t* flags_tgt);

/* VoxelizedPointCloud::insert + toPointCloud (Voxel.cpp:66-114): per-voxel running sums / counts of everything inserted so far */
size_t orc_dense_fuse(const double* pts, const double* nrm, size_t n, double voxel, double* out_pts, double* out_nrm, int32_t* counts_out);

/* ConstantVelocityMotionCompensation::undistortInputPointCloud (MotionCompensation.cpp:64-139), in place */
void orc_undistort(double* pts, size_t n, const double lin_vel[3], const double ang_vel_rpy[3], double scan_duration, int clockwise);

/* Carving of the dense voxel map: removeDuplicatePointsWithinSameVoxels (Voxel.cpp:162-191) + getKeysOfCarvedPoints
 * (helpers.cpp:347-377) + getVoxelsWithinPointNeighborhood (VoxelHashMap.cpp:13-44).  map_keys: orc_voxel_key of the occupied voxels;
 * removed_out[n_keys]; returns the number of removed voxels. */
int64_t orc_voxel_key(const double p[3], double voxel);
size_t orc_dense_carve(const double* scan, size_t n_scan, const double sensor[3], const int64_t* map_keys, size_t n_keys, double voxel,
                       double radius, double max_length, double truncation, uint8_t* removed_out);

/* A.8 RegistrationGeneralizedICP (call site src/CloudRegistration.cpp:16-21): covariances from normals
 * (C = Rx diag(eps,1,1) Rx^T, Rx = GetRotationFromE1ToX(normal)), per pair M = Ct + R Cs R^T, W = M^-1/2, residual W d (3 rows),
 * Jacobian rows W [-[p]x | I]; same loop / solve / convergence as A.1.  Both clouds must carry normals (as they always do
 * when open3d_slam reaches this call); epsilon: Open3D default 1e-3. */
int orc_icp_generalized(const double* src, const double* src_nrm, size_t n, const double* tgt, const double* tgt_nrm, size_t N,
                        const orc_kdtree* tree, double max_corr, const double init[16], int max_iter, double rel_fitness, double rel_rmse,
                        double epsilon, orc_icp_result* out);
/* one accumulation of the GICP normal equations for given correspondences (for step tests): JTJ row-major 6x6 */
void orc_gicp_jtj_jtr(const double* src, const double* src_cov /* 9n */, size_t n, const double* tgt, const double* tgt_cov /* 9N */,
                      const int32_t* corr, double JTJ[36], double JTr[6]);
void orc_covariance_from_normal(const double nrm[3], double epsilon, double cov[9]);

/* A.5 EstimateNormals(Hybrid(radius,max_nn), fast) + NormalizeNormals + OrientNormalsTowardsCameraLocation(0)
 * (call site open3d_slam/src/CloudRegistration.cpp:49-56).  normals out: 3n */
void orc_estimate_normals(const double* pts, size_t n, double radius, int max_nn, double* normals);
void orc_fast_eigen3x3_min_evec(const double cov[9], double out[3]);
/* portable restatements of std::acos on [-1,1] and std::cos on [0,pi] used by it (see o3d_oracle.c) */
double orc_acos(double x);
double orc_cos(double x);

/* A.6 VoxelDownSample (data-anchored grid); out arrays sized >= 3n; returns m; output order = first occurrence.
 * nrm/out_nrm may be NULL */
size_t orc_voxel_down_sample(const double* pts, const double* nrm, size_t n, double voxel, double* out_pts, double* out_nrm);

/* croppers.cpp:65-106: stable compaction; returns kept count; out_idx may be NULL */
size_t orc_crop_indices(const double* pts, size_t n, const orc_crop* c, int64_t* out_idx);

/* helpers.cpp:115-183 voxelizeWithinCroppingVolume (world-anchored grid, VoxelHashMap.hpp:47-50);
 * pass-through points first, then voxel means (first-occurrence order); normals re-normalised.
 * out arrays sized >= 3n; returns count; *n_pass = number of pass-through points */
size_t orc_voxelize_within_volume(const double* pts, const double* nrm, size_t n, double voxel, const orc_crop* c, double* out_pts,
                                  double* out_nrm, size_t* n_pass);

/* colours of that merge, ordered like its out_pts: a voxel keeps the colour of its LAST point (helpers.cpp:40-42,61-63,83-85).
 * ([O3D] VoxelDownSample averages colours exactly as it averages normals: use orc_voxel_down_sample with the colours as `nrm`.) */
size_t orc_voxelize_within_volume_colors(const double* pts, const double* col, size_t n, double voxel, const orc_crop* c, double* out_col);

#ifdef __cplusplus
}
#endif
#endif

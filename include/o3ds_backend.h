/*
 * o3ds_backend.h -- C-ABI of the MI355X (gfx950) scan-matching / map-fusion backend
 * for open3d_slam.  Plain pointers and sizes only; no C++/torch/Eigen types.
 *
 * Every entry point names the reference interface it replaces; paths are relative to
 * /root/reference/open3d_slam/open3d_slam/.  "[O3D]" = Open3D v0.15.1 routine the
 * reference calls at that line (third-party, pinned by open3d_catkin/CMakeLists.txt:116-118).
 *
 * Conventions
 *   clouds : const double* xyz, 3*n contiguous == std::vector<Eigen::Vector3d>::data()->data()
 *            (open3d::geometry::PointCloud::points_ / normals_, typedefs.hpp:24)
 *   poses  : const double[16] COLUMN-MAJOR == Eigen::Matrix4d::data() / Transform::matrix().data()
 *            (Transform = Eigen::Isometry3d, Transform.hpp:15)
 *   errors : every call returns O3DS_OK (0) or a negative o3ds_status; text via o3ds_last_error().
 *            Nothing throws across the ABI; the C++ adapter (open3d_slam_amd/host/) re-throws
 *            std::runtime_error to keep the reference's behaviour (assert.hpp:13-64).
 *   threads: a handle owns one HIP stream and its scratch; it is NOT re-entrant -- create one
 *            handle per calling thread (odometryWorker / mappingWorker / loopClosureWorker,
 *            SlamWrapper.cpp:258-347,406-448).  Device clouds/maps belong to the handle that made them.
 *   device : all work is enqueued on the handle's stream; calls that return host data synchronise it.
 */
#ifndef O3DS_BACKEND_H
#define O3DS_BACKEND_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct o3ds_context* o3ds_handle;
typedef uint64_t o3ds_cloud;  /* device-resident cloud (+ optional normals, + optional NN index) */

typedef enum {
  O3DS_OK = 0,
  O3DS_ERR_INVALID_ARG = -1,   /* e.g. max_correspondence_distance <= 0  ([O3D] RegistrationICP LogError) */
  O3DS_ERR_NO_NORMALS = -2,    /* point-to-plane needs target normals    ([O3D] RegistrationICP LogError) */
  O3DS_ERR_OOM = -3,
  O3DS_ERR_HIP = -4,
  O3DS_ERR_BAD_HANDLE = -5,
  O3DS_ERR_EMPTY = -6,         /* assert_gt(size,0): ScanToMapRegistration.cpp:51-52,60 */
  O3DS_ERR_CAPACITY = -7       /* caller-provided output buffer too small */
} o3ds_status;

/* numeric storage of device clouds: f32 xyz (default; accumulation is always f64) or f64 xyz */
typedef enum { O3DS_PRECISION_F32 = 0, O3DS_PRECISION_F64 = 1 } o3ds_precision;

/* croppers.hpp:26-47, croppers.cpp:121-165 (+ setIsInvertVolume croppers.cpp:57-59) */
typedef enum {
  O3DS_CROP_NONE = 0,
  O3DS_CROP_MAX_RADIUS = 1,
  O3DS_CROP_MIN_RADIUS = 2,
  O3DS_CROP_MIN_MAX_RADIUS = 3,
  O3DS_CROP_CYLINDER = 4
} o3ds_crop_kind;

typedef struct {
  int32_t kind;      /* o3ds_crop_kind */
  int32_t invert;    /* isInvertVolume_ */
  double center[3];  /* pose_.translation() -- only the translation is used (croppers.cpp:121-165) */
  double rmin, rmax; /* ScanCroppingParameters::croppingMinRadius_/MaxRadius_ (Parameters.hpp:52-58) */
  double zmin, zmax; /* croppingMinZ_/MaxZ_ (cylinder) */
} o3ds_crop;

/* open3d::pipelines::registration::RegistrationResult as consumed by open3d_slam
 * (Odometry.cpp:51-72, Mapper.cpp:151-159): transformation_, fitness_, inlier_rmse_.
 * correspondence_set_ is never read for ICP results and is not materialised. */
typedef struct {
  double transformation[16]; /* column-major */
  double fitness;
  double inlier_rmse;
  int32_t iterations; /* Gauss-Newton updates applied */
  int32_t converged;  /* ICPConvergenceCriteria test fired */
  uint64_t n_corr;
} o3ds_icp_result;

/* CloudRegistrationType (Parameters.hpp:37-42) as far as the HIP backend implements it */
typedef enum { O3DS_ICP_POINT_TO_PLANE = 0, O3DS_ICP_GENERALIZED = 1, O3DS_ICP_POINT_TO_POINT = 2 } o3ds_icp_method;

/* IcpParameters (Parameters.hpp:66-71) + [O3D] ICPConvergenceCriteria */
typedef struct {
  double max_correspondence_distance; /* IcpParameters::maxCorrespondenceDistance_ */
  int32_t max_iteration;              /* IcpParameters::maxNumIter_ -> criteria.max_iteration_ (CloudRegistration.cpp:63) */
  int32_t method;                     /* o3ds_icp_method; read by o3ds_icp_register_dev / o3ds_icp_begin only */
  double relative_fitness;            /* [O3D] default 1e-6, never overridden by the reference */
  double relative_rmse;               /* [O3D] default 1e-6 */
} o3ds_icp_params;

/* ---- lifecycle --------------------------------------------------------------------------- */
int o3ds_create(int device_id, o3ds_handle* out);
int o3ds_destroy(o3ds_handle h);
const char* o3ds_last_error(o3ds_handle h); /* h may be NULL: last error of the calling thread */
int o3ds_set_precision(o3ds_handle h, int precision /* o3ds_precision */);
int o3ds_synchronize(o3ds_handle h);
/* the hipStream_t all work of this handle is enqueued on (for event timing / interop) */
void* o3ds_stream(o3ds_handle h);
/* Enqueue this handle's work on the caller's hipStream_t instead (e.g. torch's current stream, so that a
 * torch.distributed all-reduce of the step-wise record is stream-ordered with the kernels); NULL restores the
 * handle's own stream.  The caller keeps the stream alive. */
int o3ds_set_stream(o3ds_handle h, void* hip_stream);
/* Kernel timing for bench.py's roofline line: when enabled, every ICP correspondence/reduction pass
 * (icp_accumulate_kernel launch) is bracketed by hipEvents on the launch stream.  profile_read synchronises,
 * returns the number of bracketed launches and their summed duration (ms), and resets the counters.
 * on = 2 enables the tagged spans below WITHOUT the per-launch brackets: a span around a whole registration then
 * measures its kernels back to back (bench.py's roofline.avg_launch_us = that span / passes). */
int o3ds_profile_enable(o3ds_handle h, int on);
int o3ds_profile_read(o3ds_handle h, uint64_t* n_launches, double* total_ms);
/* Tagged spans for bench.py's per-call table: while profiling is enabled, o3ds_profile_span(h, tag, 0) / (h, tag, 1) record a pair
 * of hipEvents on the handle's stream around whatever the caller enqueues in between (tags 0..7).  Tags 8 (the two kernels of
 * normal estimation) and 9 (the kernels of an index build) are marked inside the library.  span_read synchronises, returns the
 * number of complete spans of a tag and their summed duration (ms) and resets that tag. */
int o3ds_profile_span(o3ds_handle h, int tag, int end);
int o3ds_profile_span_read(o3ds_handle h, int tag, uint64_t* n_spans, double* total_ms);
/* library / kernel identification string (arch, build flags) */
const char* o3ds_version(void);

/* ---- device clouds ----------------------------------------------------------------------- */
/* Upload a host cloud (PointCloud::points_, normals_ may be NULL). */
int o3ds_cloud_upload(o3ds_handle h, const double* xyz, const double* normals, size_t n, o3ds_cloud* out);
/* The same from a sensor_msgs/PointCloud2-style buffer, without the float -> double -> float detour of
 * open3d_conversions::rosToOpen3d (open3d_utils/open3d_conversions/src/open3d_conversions.cpp:59-68, which reads the float32
 * fields x, y, z of every record and widens them): n records of point_step bytes, float32 x / y / z at byte offsets off_x /
 * off_y / off_z.  The raw buffer is copied to the device as it is and unpacked there; the values stored are exactly those the
 * double route would store.  Other fields (intensity, ring, t, rgb) are ignored, as on the scan-matching path. */
int o3ds_cloud_upload_f32(o3ds_handle h, const void* data, size_t n, size_t point_step, size_t off_x, size_t off_y, size_t off_z,
                          o3ds_cloud* out);
/* The ingest is ASYNCHRONOUS: copy, unpack and (for the volume this handle last passed to o3ds_crop_voxel_down_sample) the bounding box
 * of the points inside the cropping volume run on the handle's copy stream, so that scan k + 1 crosses PCIe while frame k is still being
 * registered and merged; the first operation that uses the cloud queues a wait for it on the handle's stream (no host wait).  A buffer
 * of ordinary (pageable) memory has been consumed when the call returns (it goes through the handle's pinned ring).  A buffer from
 * o3ds_pinned_alloc (or any memory registered with the HIP runtime) is read by DMA after the call has returned: keep its contents until
 * o3ds_cloud_wait_ingest(cloud) has returned or any call that returns data derived from the cloud (a registration result, a size, a
 * download) has. */
int o3ds_pinned_alloc(o3ds_handle h, size_t bytes, void** out);  /* page-locked host memory for sensor / message buffers (ROS 2 allocators, drivers) */
int o3ds_pinned_free(o3ds_handle h, void* p);
int o3ds_cloud_wait_ingest(o3ds_handle h, o3ds_cloud c);
int o3ds_cloud_free(o3ds_handle h, o3ds_cloud c);
/* The number of points of a cloud and whether it has normals.  The clouds of the per-scan chain -- o3ds_voxel_down_sample /
 * o3ds_crop_voxel_down_sample and what o3ds_estimate_normals, the registrations and o3ds_map_insert_scan make of their results -- are
 * sized ON THE DEVICE: the kernel that counts the voxels publishes the number where the kernels that consume the cloud read it, the
 * frame's launches are queued from an upper bound, and nothing waits for a size to come back.  Asking for n is what waits (once; the
 * number usually arrived with the next registration result and then costs nothing); n == NULL asks for has_normals only and never
 * waits.  Every call that sizes a host buffer or another cloud by the number (download, crop, select, transform, append ...) asks. */
int o3ds_cloud_size(o3ds_handle h, o3ds_cloud c, size_t* n, int* has_normals);
/* What is known of the size without waiting: lower <= n <= upper (equal once the number has arrived).  The reference's emptiness checks
 * (assert_gt(cloud.size(), 0), ScanToMapRegistration.cpp:51-52; preProcessedScan.IsEmpty(), Submap.cpp:41) are decided by the bounds:
 * a VoxelDownSample result whose input had a point inside the volume has lower = 1. */
int o3ds_cloud_size_bound(o3ds_handle h, o3ds_cloud c, size_t* lower, size_t* upper);
/* Download into caller buffers of capacity >= n points (normals may be NULL). */
int o3ds_cloud_download(o3ds_handle h, o3ds_cloud c, double* xyz, double* normals, size_t capacity);
/* The way out, again without a double detour: write the cloud as n records of point_step bytes with float32 x / y / z at byte
 * offsets off_x / off_y / off_z; unless off_normal == O3DS_NO_FIELD, the normal as three float32 at off_normal, +4, +8; unless
 * off_rgb == O3DS_NO_FIELD, the colour as the packed `rgb` field (bytes b, g, r, 0 at off_rgb, +1, +2, +3); every other byte of the
 * records is zero.  point_step 16, offsets 0 / 4 / 8 is the sensor_msgs/PointCloud2 layout that open3d_conversions::open3dToRos
 * produces for a cloud without colours, point_step 32 with rgb at 16 the one for a coloured cloud
 * (open3d_utils/open3d_conversions/src/open3d_conversions.cpp:19-53: `*ros_pc2_x = point(0)`, a double -> float narrowing, and
 * `*ros_pc2_r = (int)(255 * color(0))` = rgb_rounding 0); point_step 24, offsets 0 / 4 / 8 / 12 is the row of a binary PCD with FIELDS
 * x y z normal_x normal_y normal_z as [O3D] io::WritePointCloudToPCD writes it for saveToFile
 * (open3d_slam/open3d_slam/src/output.cpp:39-47), whose rgb field uses [O3D] utility::ColorToUint8 (clamp, scale, round = rgb_rounding
 * 1).  data must hold capacity >= n records. */
#define O3DS_NO_FIELD ((size_t)-1)
int o3ds_cloud_download_f32(o3ds_handle h, o3ds_cloud c, void* data, size_t capacity, size_t point_step, size_t off_x, size_t off_y,
                            size_t off_z, size_t off_normal, size_t off_rgb, int rgb_rounding);
/* Colours of an ingested cloud, read from the same PointCloud2 records as rosToOpen3d does when skip_colors is false -- which is how
 * every caller in the reference invokes it (OnlineRangeDataProcessorRos.cpp:40, RosbagRangeDataProcessorRos.cpp:129,
 * SlamMapInitializer.cpp:124): O3DS_COLOR_FIELD_RGB = a fourth field named "rgb" at byte offset off_field (r, g, b = bytes +2, +1, +0,
 * each / 255.0; open3d_conversions.cpp:71-80); O3DS_COLOR_FIELD_INTENSITY = a fourth field named "intensity", which the reference reads
 * through a uint8 iterator: the FIRST byte of the field, unscaled, for all three channels (open3d_conversions.cpp:81-86).  The cloud
 * must have been made from the same n records (o3ds_cloud_upload_f32). */
enum { O3DS_COLOR_FIELD_RGB = 0, O3DS_COLOR_FIELD_INTENSITY = 1 };
int o3ds_cloud_set_colors_from_records(o3ds_handle h, o3ds_cloud c, const void* data, size_t point_step, size_t off_field, int kind);
/* PointCloud::colors_ (3n doubles in [0, 1]) of a device cloud.  Registration never reads them; the cloud operations carry them
 * the way the reference does: crop / select / carve keep the colours of the kept points, transform copies them, append follows
 * [O3D] PointCloud::operator+= (kept only if the map is empty or coloured AND the added cloud is coloured), o3ds_voxel_down_sample
 * averages them ([O3D] VoxelDownSample), and the map merge o3ds_voxelize_within_volume / o3ds_map_insert_scan gives a voxel the
 * colour of its LAST point in cloud order (AccumulatedPoint::AddPoint assigns, helpers.cpp:40-42,61-63; isValidColor,
 * helpers.cpp:83-85, holds for every value).  rgb == NULL clears.  get_colors returns O3DS_ERR_EMPTY for an uncoloured cloud. */
int o3ds_cloud_set_colors(o3ds_handle h, o3ds_cloud c, const double* rgb);
/* PointCloud::covariances_ is NOT carried: a device cloud has points, normals and colours.  The reference's own flow never holds any --
 * the one call that would fill them, EstimateCovariances, is commented out (CloudRegistration.cpp:29), generalized ICP builds its
 * covariances from the normals inside the registration (Open3D does the same when a cloud has none), and nothing else writes the field --
 * so the copies croppers.cpp:86-101 and helpers.cpp:48-50,64-67,297-302 make of it move empty vectors.  A caller that fills
 * covariances_ itself keeps them on its host PointCloud; they do not cross the seams (integration/o3ds_open3d_slam.hpp clears the field
 * of every cloud it hands back, as an Open3D call that recomputes a cloud would). */
int o3ds_cloud_has_colors(o3ds_handle h, o3ds_cloud c, int* has_colors);
int o3ds_cloud_get_colors(o3ds_handle h, o3ds_cloud c, double* rgb, size_t capacity);
/* Build the nearest-neighbour index of a cloud (replaces [O3D] KDTreeFlann::SetGeometry(target), which
 * RegistrationICP does on every call).  cell_size <= 0 selects max_corr_hint/4.  Idempotent per cell size. */
int o3ds_cloud_build_index(o3ds_handle h, o3ds_cloud c, double max_corr_hint, double cell_size);

/* ---- Seam 1: CloudRegistration::registerClouds (CloudRegistration.hpp:25) ----------------- */
/* RegistrationIcpPointToPlane::registerClouds (CloudRegistration.cpp:44-48) = [O3D] RegistrationICP with
 * TransformationEstimationPointToPlane.  Host-buffer form: uploads both clouds, builds the target index
 * (as the reference rebuilds its KD-tree per call), runs ICP, frees the device copies. */
int o3ds_icp_point_to_plane(o3ds_handle h, const double* src_xyz, size_t n_src, const double* tgt_xyz, const double* tgt_normals,
                            size_t n_tgt, const double init[16], const o3ds_icp_params* params, o3ds_icp_result* out);
/* Device-resident form.  target must have normals and an index.  target_crop (may be NULL) restricts the
 * target to the points inside the volume -- ScanToMapIcp::scanToMapRegistration's
 * scanMatcherCropper_->crop(activeSubmapPointCloud) (ScanToMapRegistration.cpp:58-59) fused into the search. */
int o3ds_icp_point_to_plane_dev(o3ds_handle h, o3ds_cloud source, o3ds_cloud target, const o3ds_crop* target_crop,
                                const double init[16], const o3ds_icp_params* params, o3ds_icp_result* out);

/* Same loop with params->method selecting the estimator (point-to-plane, generalized or point-to-point). */
int o3ds_icp_register_dev(o3ds_handle h, o3ds_cloud source, o3ds_cloud target, const o3ds_crop* target_crop, const double init[16],
                          const o3ds_icp_params* params, o3ds_icp_result* out);
/* Work for the time the host would wait for a registration.  The callback is handed to the NEXT device-resident registration on the
 * handle (o3ds_icp_register_dev and the per-estimator entry points above) and is called at most once, on the calling thread, after the
 * registration's launches have been queued and before the host waits for their result: whatever it queues on the handle runs behind them,
 * in the time the host spends waiting and -- afterwards -- deciding what to do with the pose.  In open3d_slam that is the odometry
 * worker's pre-processing of the NEXT raw scan (SlamWrapper.cpp:228-229 runs it on its own thread; a single-threaded caller gets the
 * same overlap this way).  The callback may call any o3ds_* function on this handle except a registration, a session call or anything
 * that frees or changes the two clouds being registered; calls that read a size back (o3ds_cloud_size, downloads) wait for the
 * registration first and so defeat the purpose.  Not called when the registration fails before it queues anything or takes the
 * two-launch form (sources beyond 262 144 points): *the caller checks* (the callback can count).  fn == NULL withdraws it. */
typedef void (*o3ds_overlap_fn)(void* arg);
int o3ds_icp_overlap_next(o3ds_handle h, o3ds_overlap_fn fn, void* arg);
/* RegistrationIcpGeneralized::registerClouds (CloudRegistration.cpp:16-21) = [O3D] RegistrationGeneralizedICP with a
 * default TransformationEstimationForGeneralizedICP (epsilon 1e-3): per-point covariances C = Rx diag(eps,1,1) Rx^T built from
 * the clouds' (unit) normals, residual (Ct + R Cs R^T)^-1/2 (p - q), same loop / solve / convergence test as point-to-plane.
 * open3d_slam always calls estimateNormalsOrCovariancesIfNeeded first (CloudRegistration.cpp:22-30); a cloud that still has no
 * normals (null pointer / device cloud without normals) gets [O3D] InitializePointCloudForGeneralizedICP's treatment:
 * EstimateNormals(KDTreeSearchParamKNN(20)) on a copy, the caller's cloud is left as it was. */
int o3ds_icp_generalized(o3ds_handle h, const double* src_xyz, const double* src_normals, size_t n_src, const double* tgt_xyz,
                         const double* tgt_normals, size_t n_tgt, const double init[16], const o3ds_icp_params* params, o3ds_icp_result* out);
int o3ds_icp_generalized_dev(o3ds_handle h, o3ds_cloud source, o3ds_cloud target, const o3ds_crop* target_crop, const double init[16],
                             const o3ds_icp_params* params, o3ds_icp_result* out);
/* RegistrationIcpPointToPoint::registerClouds (CloudRegistration.cpp:69-74) = [O3D] RegistrationICP with
 * TransformationEstimationPointToPoint: the update of every iteration is the closed-form Eigen::umeyama (no scaling) of the
 * matched pairs; same loop and convergence test.  No normals are needed on either cloud.
 * Step-wise record in this method: [0..8] sum q_a p_b (row-major, q = matched target point, p = transformed source point),
 * [9..11] sum p, [12..14] sum q, [28] #corr, [29] sum d^2, the rest 0. */
int o3ds_icp_point_to_point(o3ds_handle h, const double* src_xyz, size_t n_src, const double* tgt_xyz, size_t n_tgt,
                            const double init[16], const o3ds_icp_params* params, o3ds_icp_result* out);
int o3ds_icp_point_to_point_dev(o3ds_handle h, o3ds_cloud source, o3ds_cloud target, const o3ds_crop* target_crop, const double init[16],
                                const o3ds_icp_params* params, o3ds_icp_result* out);
/* [O3D] GetInformationMatrixFromPointClouds(source, target, max_correspondence_distance, transformation) (call sites
 * constraint_builders.cpp:70-73, PlaceRecognition.cpp:148-149): one correspondence pass under T (same exact 1-NN search as the
 * ICP passes), Lambda = sum over the matched TARGET points q of G^T G, G = [-[q]x | I].  information: 36 doubles; a symmetric
 * matrix, so row- and column-major coincide (Eigen::Matrix6d::data() can be passed). */
int o3ds_information_matrix(o3ds_handle h, const double* src_xyz, size_t n_src, const double* tgt_xyz, size_t n_tgt, const double T[16],
                            double max_correspondence_distance, double information[36]);
int o3ds_information_matrix_dev(o3ds_handle h, o3ds_cloud source, o3ds_cloud target, const o3ds_crop* target_crop, const double T[16],
                                double max_correspondence_distance, double information[36]);
/* epsilon of the plane-to-plane covariance model (default 1e-3, Open3D's default) */
int o3ds_set_gicp_epsilon(o3ds_handle h, double epsilon);

/* Step-wise form of the same loop, for sharded (multi-GPU) registration: the caller all-reduces the 32-double
 * normal-equation record between accumulate and update.  Record layout (doubles):
 *   [0..20] upper triangle of JtJ row-major, [21..26] Jtr, [27] sum r^2, [28] #corr, [29] sum d^2, [30..31] 0.
 * begin      : T <- init, resets iteration state.
 * accumulate : one correspondence + reduction pass ([O3D] GetRegistrationResultAndCorrespondences +
 *              ComputeJTJandJTr) of source points [first, first+count) under the current T; writes the record to
 *              d_record (DEVICE pointer to 32 doubles, e.g. a torch tensor's data_ptr).
 * update     : consumes a (possibly all-reduced) record on the device: convergence test against the previous pass,
 *              6x6 LDLT, T <- U*T ([O3D] SolveJacobianSystemAndObtainExtrinsicMatrix).  n_src_total = |source| over
 *              all shards (fitness denominator).
 * finish     : synchronises and returns the result (state after the last evaluated pass). */
int o3ds_icp_begin(o3ds_handle h, o3ds_cloud source, o3ds_cloud target, const o3ds_crop* target_crop, const double init[16],
                   const o3ds_icp_params* params);
int o3ds_icp_accumulate(o3ds_handle h, size_t first, size_t count, double* d_record);
int o3ds_icp_update(o3ds_handle h, const double* d_record, uint64_t n_src_total);
int o3ds_icp_finish(o3ds_handle h, o3ds_icp_result* out);
/* Target-sharded registration against ONE map whose points are split over up to 16 GPUs (SURVEY.md 8e
 * Partitioning B) -- the sharding that reproduces the reference's registration against a single cloud (Mapper.cpp:141,
 * ScanToMapRegistration.cpp:57-61).  Per iteration, inside an o3ds_icp_begin session on this rank's shard:
 *   o3ds_icp_nn_keys(h, 0, n, rank, d_keys)      every query's best match in THIS shard as a key: float bits of d2 << 32 | rank << 28 |
 *                                                position in the shard's index; INT64_MAX when nothing lies within the radius (keys order as signed or unsigned)
 *   ncclAllReduce(d_keys, n, uint64, min)        non-negative float bit patterns order like the floats: the minimum is the match in
 *                                                the union of the shards (equal distances: lower rank, then lower position)
 *   o3ds_icp_accumulate_keys(h, 0, n, rank, d_keys, d_record)   the queries whose winning key names this rank contribute their rows
 *   ncclAllReduce(d_record, 32, double, sum)  ->  o3ds_icp_update(h, d_record, n)
 * and o3ds_icp_finish as usual.  d2 enters the key as a float: distances that differ only beyond float precision tie. */
int o3ds_icp_nn_keys(o3ds_handle h, size_t first, size_t count, int rank, unsigned long long* d_keys);
int o3ds_icp_accumulate_keys(o3ds_handle h, size_t first, size_t count, int rank, const unsigned long long* d_keys, double* d_record);
/* 1 when the device-side loop has terminated (converged or max_iteration reached); synchronises. */
/* Fused step-wise form -- ONE kernel per iteration plus the caller's collective (the accumulate / reduce / update triple above is
 * three).  o3ds_icp_pass enqueues launch j of the fused loop over the source range [first, first + count): its prologue folds
 * d_sums_in (the ALL-REDUCED sums of the previous pass; ignored by the first call of a session) and applies the update, its body
 * adds this rank's part of pass j to d_sums_out, and it clears d_sums_next.  The three buffers are O3DS_ICP_SUMS_DOUBLES doubles
 * each, on the device, owned and rotated by the caller (out -> in -> next -> out ...), all zero before the first call.  Between
 * two calls the caller sums d_sums_out element-wise over the ranks (ncclAllReduce, in place, on o3ds_stream(h)).  The addends are
 * split so that these sums are exact: the result does not depend on the number of ranks or the reduction order.
 * o3ds_icp_pass_finish folds the last pass (one-workgroup launch) and returns the result; max_iteration + 1 passes make a full
 * registration, passes issued after the loop has terminated on the device are no-ops. */
#define O3DS_ICP_SUMS_DOUBLES 512
/* o3ds_icp_pass serves at most this many source points per call (one workgroup per 128 queries -- 64 until the end of round 6 --; the exact sums hold for 4096 workgroup records per pass);
 * a larger range returns O3DS_ERR_CAPACITY -- shards beyond it use the o3ds_icp_accumulate / o3ds_icp_update triple, which loops
 * (open3d_slam_amd/sharded.py falls back by itself).  o3ds_icp_register_dev has no such limit (it switches form internally). */
#define O3DS_ICP_PASS_MAX_QUERIES 262144
int o3ds_icp_pass(o3ds_handle h, size_t first, size_t count, size_t n_src_total, const double* d_sums_in, double* d_sums_out,
                  double* d_sums_next);
int o3ds_icp_pass_finish(o3ds_handle h, size_t n_src_total, const double* d_sums_in, double* d_sums_scratch, o3ds_icp_result* out);
int o3ds_icp_done(o3ds_handle h, int* done);

/* ---- the same registrations, sharded over the GPUs of a node, entirely inside the library: kernels and RCCL collectives queued on
 *      o3ds_stream(h) for ALL max_iteration + 1 passes, no host code between them (the reference registers against one cloud,
 *      Mapper.cpp:141 -> ScanToMapRegistration.cpp:55-62 -> CloudRegistration.cpp:44-48; the partitionings are SURVEY.md 8e's).
 * One process per GPU, one handle per process.  RCCL (librccl.so) is loaded with dlopen at the first call of this group -- a
 * single-GPU user of the library does not need it installed -- and the communicator is created by THAT copy of the library:
 *   o3ds_comm_unique_id   on ONE rank: a 128-byte ncclUniqueId, to be handed to every rank by whatever the processes share
 *                         (torch.distributed's broadcast, MPI, a file);
 *   o3ds_comm_init        on EVERY rank (collective: ncclCommInitRank): the handle keeps the communicator;
 *   o3ds_comm_attach      instead of init: an ncclComm_t the caller made with the same librccl.so the process has loaded;
 *   o3ds_comm_destroy     (also done by o3ds_destroy).
 * o3ds_icp_register_sharded: `partitioning`
 *   O3DS_SHARD_SOURCE  every rank holds the same target; rank r contributes the source points [r n / W, (r + 1) n / W): per pass ONE
 *                      icp_fused_kernel + ONE ncclAllReduce(sum) of the 512-double exact hi / lo record (4 KB) -- equal to the one-GPU
 *                      registration bit for bit, whatever W (the sums are exact);
 *   O3DS_SHARD_SUBMAP  every rank holds ITS OWN submap and the whole source: the same kernel and collective, the summed record is the
 *                      joint problem over the W submaps (BASELINE.json configs[3]; fitness = mean over the submaps);
 *   O3DS_SHARD_UNION   ONE map split over the ranks, every rank holds the whole source: per pass search kernel -> ncclAllReduce(min) of
 *                      n 64-bit keys -> accumulate kernel -> ncclAllReduce(sum) of 32 doubles -> update kernel; equal to the
 *                      registration against the whole map (W <= 16).
 * Point-to-plane, generalized and point-to-point estimators (params->method) as o3ds_icp_register_dev.  With a communicator of one
 * rank every collective is the identity and SOURCE / SUBMAP reproduce o3ds_icp_register_dev bit for bit. */
#define O3DS_SHARD_SOURCE 0
#define O3DS_SHARD_SUBMAP 1
#define O3DS_SHARD_UNION 2
int o3ds_comm_unique_id(o3ds_handle h, unsigned char id[128]);
int o3ds_comm_init(o3ds_handle h, const unsigned char id[128], int rank, int world);
int o3ds_comm_attach(o3ds_handle h, void* nccl_comm, int rank, int world);
int o3ds_comm_destroy(o3ds_handle h);
int o3ds_icp_register_sharded(o3ds_handle h, int partitioning, o3ds_cloud source, o3ds_cloud target, const o3ds_crop* target_crop,
                              const double init[16], const o3ds_icp_params* params, o3ds_icp_result* out);

/* ---- scan pre-processing: ScanToMapIcp::preprocess (ScanToMapRegistration.cpp:35-40),
 *      LidarOdometry::preprocess (Odometry.cpp:25-30) ------------------------------------------ */
/* CroppingVolume::crop (croppers.cpp:76-106): stable compaction of points (+normals) inside the volume. */
int o3ds_crop_cloud(o3ds_handle h, o3ds_cloud in, const o3ds_crop* crop, o3ds_cloud* out);
/* o3d_slam::voxelize (helpers.cpp:107-113) -> [O3D] PointCloud::VoxelDownSample: data-anchored grid,
 * per-voxel mean of points (and normals), each sum taken in cloud order.  Output order: voxels in order of their first point in the
 * cloud (the reference's order is unordered_map iteration order, i.e. unspecified).  voxel <= 0 returns a copy. */
int o3ds_voxel_down_sample(o3ds_handle h, o3ds_cloud in, double voxel_size, o3ds_cloud* out);
/* The first two steps of ScanToMapIcp::preprocess (ScanToMapRegistration.cpp:36-37) and LidarOdometry::preprocess (Odometry.cpp:26-27),
 * `cropped = cropper->crop(in); voxelize(voxelSize, cropped)`, as one call: the same cloud o3ds_crop_cloud followed by
 * o3ds_voxel_down_sample returns, bit for bit (grid anchored at the bounding box of the points INSIDE the volume, voxel means summed
 * in the same order), without materialising the cropped cloud.  voxel_size <= 0 is o3ds_crop_cloud. */
int o3ds_crop_voxel_down_sample(o3ds_handle h, o3ds_cloud in, const o3ds_crop* crop, double voxel_size, o3ds_cloud* out);
/* RegistrationIcpPointToPlane::estimateNormalsOrCovariancesIfNeeded (CloudRegistration.cpp:49-56):
 * [O3D] EstimateNormals(KDTreeSearchParamHybrid(radius,max_nn)) + NormalizeNormals +
 * OrientNormalsTowardsCameraLocation(0,0,0).  In place (adds/overwrites the cloud's normals). */
int o3ds_estimate_normals(o3ds_handle h, o3ds_cloud c, double radius, int max_nn);
/* [O3D] RandomDownSample is seeded from std::random_device (non-reproducible); the ABI takes the
 * kept indices explicitly: SelectByIndex(keep_idx[0..m)). */
int o3ds_select_by_index(o3ds_handle h, o3ds_cloud in, const uint32_t* keep_idx, size_t m, o3ds_cloud* out);
/* [O3D] PointCloud::RandomDownSample(sampling_ratio) as open3d_slam calls it (Odometry.cpp:29, ScanToMapRegistration.cpp:39), drawn on the
 * device: keeps k = (int)(ratio * n) points, a uniformly drawn k-subset, in cloud order (the order [O3D] SelectByIndex's mask walk emits).
 * The reference draws from an mt19937 seeded by std::random_device -- there is no sequence to reproduce --; here index i gets the 64-bit key
 * mix(seed + (i + 1) * 0x9E3779B97F4A7C15) (splitmix64's finaliser: distinct keys) and the k smallest keys are kept, so the subset is a
 * function of (seed, n, ratio) alone, restated in numpy for the checker (oracle/pipeline.py draw_keep).  Neither n nor k passes through the
 * host: a cloud whose size is still in flight (o3ds_crop_voxel_down_sample's result) is drawn from as it is, and the result's size is in
 * flight in turn.  ratio outside [0, 1]: O3DS_ERR_INVALID_ARG with Open3D's message; the input is left as it is. */
int o3ds_random_down_sample(o3ds_handle h, o3ds_cloud in, double ratio, uint64_t seed, o3ds_cloud* out);

/* ---- map fusion: Submap::insertScan (Submap.cpp:39-75) ------------------------------------- */
/* o3d_slam::transform (helpers.cpp:273-305): p' = (T p).xyz/w, n' = R n.  Returns a new cloud.
 * (The reference's duplicate-on-identity quirk, SURVEY B4, is intentionally not reproduced.) */
int o3ds_transform_cloud(o3ds_handle h, o3ds_cloud in, const double T[16], o3ds_cloud* out);
/* mapCloud_ += cloud (Submap.cpp:70; [O3D] PointCloud::operator+=). Appends `add` to `map` in place. */
int o3ds_cloud_append(o3ds_handle h, o3ds_cloud map, o3ds_cloud add);
/* A copy of a cloud of handle `src` as a cloud of handle `dst` (same device, same storage precision), device to device: points,
 * normals, colours and the known bounding box; the search index is not copied.  This is how a scan that one worker pre-processed
 * (ScanToMapIcp::processForScanMatchingAndMerging on the mapping thread's handle, ScanToMapRegistration.cpp:42-54) reaches the
 * handle that owns the submap (Submap::insertScan, Submap.cpp:39-75; ScanToMapIcp::scanToMapRegistration, :55-62) without a second
 * trip through host memory -- the reference passes the same host PointCloud to both.  `dst`'s stream waits for what `src` queued;
 * neither handle may be in use by another thread during the call; returns when the copy is complete. */
int o3ds_cloud_copy_across(o3ds_handle dst, o3ds_handle src, o3ds_cloud src_cloud, o3ds_cloud* out);
/* The same hand-over WITHOUT the source handle: open3d_slam's odometry and mapping workers run on two threads (SlamWrapper.cpp:227-236),
 * each with its own handle, and the mapper wants the scan the odometry worker pre-processed while that worker is busy registering -- a
 * call that needs both handles idle (o3ds_cloud_copy_across) either waits for the other worker or falls back to the host arrays.
 * o3ds_cloud_export_view, called by the OWNER right after it made the cloud, freezes what a copy needs -- the device arrays, the size,
 * the box, and an event behind everything the owner has queued for the cloud -- in a plain struct the caller keeps with the cloud;
 * o3ds_cloud_import_view makes a copy on `dst` from the struct alone (dst's stream waits for the event; device-to-device copies;
 * complete on return) and may run while the owner's handle is in use by its own thread.  The owner keeps the cloud alive and does not
 * change it while views of it are in use (a pre-processed scan is never changed), and releases the view (its event) before or after
 * freeing the cloud.  Same device, same storage precision. */
typedef struct o3ds_cloud_view {
  void* pts;
  void* nrm;
  void* col;
  size_t n;
  int precision;
  int device;
  int has_box;
  int reserved;
  double box_min[3], box_max[3];
  void* event;
} o3ds_cloud_view;
int o3ds_cloud_export_view(o3ds_handle h, o3ds_cloud c, o3ds_cloud_view* out);
int o3ds_cloud_import_view(o3ds_handle dst, const o3ds_cloud_view* view, o3ds_cloud* out);
int o3ds_cloud_view_release(o3ds_cloud_view* view);
/* voxelizeWithinCroppingVolume (helpers.cpp:115-183) via Submap::voxelizeInsideCroppingVolume (Submap.cpp:138-144):
 * points outside the volume pass through (original order, first); points inside are replaced by per-voxel means on
 * the WORLD-anchored grid key = floor(p * (1/voxel)) (VoxelHashMap.hpp:47-50), normals averaged (NaN skipped) and
 * re-normalised (helpers.cpp:172).  Voxel means are emitted in ascending key order.  In place. */
int o3ds_voxelize_within_volume(o3ds_handle h, o3ds_cloud map, double voxel_size, const o3ds_crop* crop);
/* (Submap::insertScan's core in one call: o3ds_map_insert_scan, declared with its description at the end of this header.) */
/* ConstantVelocityMotionCompensation::undistortInputPointCloud (src/MotionCompensation.cpp:64-139), in place on a device cloud in
 * the sensor frame: every point is moved by the motion accumulated over phase * scan_duration at the given constant velocity
 * (phase = azimuth / 2 pi, or 1 - that for spinning_clockwise; angular velocity as roll / pitch / yaw rates, R = Rz Ry Rx).  The
 * velocities are estimated from the pose buffer by the caller (estimateLinearAndAngularVelocity, :33-58).  Normals and the NN index
 * of the cloud, if any, are dropped (the reference de-skews raw scans). */
int o3ds_cloud_undistort(o3ds_handle h, o3ds_cloud cloud, const double linear_velocity[3], const double angular_velocity_rpy[3],
                         double scan_duration, int spinning_clockwise);
/* Dense voxel map = VoxelizedPointCloud (include/open3d_slam/Voxel.hpp:38-76, src/Voxel.cpp:18-114), the "TSDF-style" fusion of
 * BASELINE configs[4]: a persistent map voxel (key floor(p / voxel_size)) -> {count, sum of positions, sum of normals}.
 *   insert    : VoxelizedPointCloud::insert of o3d_slam::transform(T, cloud) (Submap::insertScanDenseMap, Submap.cpp:77-92); T may be
 *               NULL (identity).  The cropping steps of insertScanDenseMap are o3ds_crop_cloud calls of the caller.
 *   to_cloud  : toPointCloud -- sum / count per voxel (the mean normal is NOT re-normalised, Voxel.cpp:21-23), voxels in ascending key
 *               order (the reference's order is its hash map's).
 *   transform : VoxelizedPointCloud::transform (Voxel.cpp:49-64) as written: keys unchanged, the Isometry applied to the sums as
 *               if they were points (its translation enters the position sum once, and the normal sum too).
 * Sums are kept in fixed point on the device (2^-30 m / 2^-40), so a map does not depend on the order of concurrent insertions.
 *   carve     : Submap::carve for the dense map (Submap.cpp:126-136): the scan (placed by scan_pose, NULL = identity -- NB the
 *               reference passes the RAW scan, i.e. sensor-frame points, together with the map-frame sensor position, Submap.cpp:88)
 *               is reduced to the first point of every voxel (removeDuplicatePointsWithinSameVoxels, Voxel.cpp:162-191); every kept
 *               point casts a ray from sensor_position sampled every 2 * neighborhood_radius while
 *               distance < max(2 * radius, min(length - truncation, max_length)); at every sample the voxels of
 *               getVoxelsWithinPointNeighborhood (VoxelHashMap.cpp:13-44) are removed (helpers.cpp:347-377).
 *               neighborhood_radius must be > 0: with 0 the reference's own ray loop (step 2 * radius) never terminates. */
typedef uint64_t o3ds_dense_map;
int o3ds_dense_map_create(o3ds_handle h, double voxel_size, o3ds_dense_map* out);
int o3ds_dense_map_free(o3ds_handle h, o3ds_dense_map id);
int o3ds_dense_map_insert(o3ds_handle h, o3ds_dense_map id, o3ds_cloud cloud, const double T[16]);
int o3ds_dense_map_size(o3ds_handle h, o3ds_dense_map id, size_t* n_voxels);
int o3ds_dense_map_to_cloud(o3ds_handle h, o3ds_dense_map id, o3ds_cloud* out);
int o3ds_dense_map_transform(o3ds_handle h, o3ds_dense_map id, const double T[16]);
int o3ds_dense_map_carve(o3ds_handle h, o3ds_dense_map id, o3ds_cloud scan, const double scan_pose[16], const double sensor_position[3],
                         double neighborhood_radius, double max_raytracing_length, double truncation_distance, size_t* n_removed);
/* Number of points of `cloud` (placed by T; NULL = identity) that fall into an occupied voxel of the map:
 * VoxelHashMap::hasVoxelContainingPoint per point, the count behind SubmapCollection::isSwitchingSubmapsConsistant
 * (src/SubmapCollection.cpp:352-364: fitness = hits / scan size, compared with adjacencyBasedRevisitingMinFitness_). */
/* ONE dense voxel map over several GPUs (BASELINE.json configs[4]; SURVEY.md 8e "map fusion across GPUs"): a voxel lives on the rank
 * hash(voxel) mod world, with the reference's own hash (VoxelHashMap.hpp:25-35).  export places the cloud by T (NULL: as it is), rounds
 * to the storage type, drops non-finite points and writes the rows [x y z nx ny nz] (doubles; zeros when the cloud has no normals)
 * grouped by owner into d_rows (device memory, capacity cloud size x 6) and the group sizes into d_counts (device memory, world
 * entries) -- the send buffer and split sizes of one all-to-all.  import turns received rows into a cloud of this handle, ready for
 * o3ds_dense_map_insert.  No host copy on either side of the collective. */
int o3ds_cloud_export_rows_by_owner(o3ds_handle h, o3ds_cloud cloud, const double T[16], double voxel_size, int world, double* d_rows,
                                    long long* d_counts);
int o3ds_cloud_import_rows(o3ds_handle h, const double* d_rows, size_t n, int has_normals, o3ds_cloud* out);
int o3ds_dense_map_count_occupied(o3ds_handle h, o3ds_dense_map id, o3ds_cloud cloud, const double T[16], size_t* n_hits);
/* computeIndicesOfOverlappingPoints (src/helpers.cpp:307-332; call sites src/PlaceRecognition.cpp:103,
 * src/constraint_builders.cpp:54): both clouds are binned with the voxel key floor(p / voxel_size) -- the source after being
 * placed by source_to_target --, and every voxel that holds at least min_points_per_voxel points of EACH cloud contributes all its
 * source and all its target indices.  idx_source / idx_target: caller buffers with room for every point of the respective cloud;
 * filled in ascending order (the reference's order is its hash map's iteration order, i.e. unspecified). */
int o3ds_overlap_indices(o3ds_handle h, o3ds_cloud source, o3ds_cloud target, const double source_to_target[16], double voxel_size,
                         size_t min_points_per_voxel, uint64_t* idx_source, size_t* n_idx_source, uint64_t* idx_target, size_t* n_idx_target);
/* Submap::carve for the sparse map (Submap.cpp:109-125 -> getIdxsOfCarvedPoints, helpers.cpp:235-271; SpaceCarvingParameters,
 * Parameters.hpp:85-92): the raw scan (sensor frame) is placed with map_to_range_sensor; every ray from the sensor position is
 * sampled every voxel_size metres while distance < max(voxel_size, min(length - truncation_distance, max_raytracing_length));
 * map points inside map_builder_crop that share a sampled voxel (key floor(p / voxel_size)) are removed when the map has no
 * normals or |ray direction . unit normal| > min_dot_product_with_normal.  The surviving points keep their order.  The caller
 * decides WHEN to carve (nScansInsertedMap_ % carveSpaceEveryNscans_ == 1, Submap.cpp:111). */
typedef struct {
  double voxel_size;                  /* 0.1  */
  double max_raytracing_length;       /* 20.0 */
  double truncation_distance;         /* 0.1  */
  double min_dot_product_with_normal; /* 0.5  */
} o3ds_carving_params;
int o3ds_map_carve(o3ds_handle h, o3ds_cloud map, o3ds_cloud raw_scan, const double map_to_range_sensor[16],
                   const o3ds_crop* map_builder_crop, const o3ds_carving_params* params, size_t* n_removed);
/* The same, and the carved points come back as a cloud of their own (map order; empty when nothing was carved): what Submap::carve keeps in
 * its toRemove_ member for the visualisation (Submap.cpp:119, `toRemove_ = *map->SelectByIndex(idxsToRemove)`).  removed_out may be NULL. */
int o3ds_map_carve_removed(o3ds_handle h, o3ds_cloud map, o3ds_cloud raw_scan, const double map_to_range_sensor[16],
                           const o3ds_crop* map_builder_crop, const o3ds_carving_params* params, size_t* n_removed, o3ds_cloud* removed_out);
/* Submap::insertScan's tail (Submap.cpp:66-75): map += T * scan, voxelizeWithinCroppingVolume(map_voxel_size, map_builder_crop)
 * (helpers.cpp:115-183), and -- max_corr_hint > 0 -- the map's search index for the next registration.  The RESULT is the reference's: the
 * array [points outside the volume in their previous order | one mean per voxel inside it, in voxel-key order], bit for bit in f64
 * storage.  HOW it is kept is the backend's: from its second insertion on a map with an index and without colours stays in a PERSISTENT
 * form (slot arrays + voxel hash + row-paged index) that an insertion updates only where the scan falls -- the call queues
 * eight kernels over the scan and returns, nothing is proportional to the map's size and nothing comes back to the host -- and turns into
 * the array above when somebody asks for it: o3ds_cloud_download*, o3ds_map_carve with another voxel size, o3ds_overlap_indices,
 * o3ds_estimate_normals ON the map, a registration with the map as its SOURCE, any call that reads or rewrites the map as an array
 * (that fold sorts the live points once, O(N log N)).  o3ds_cloud_size answers from the device's counters (live = slots - dead) and
 * leaves the map persistent.
 * Registrations against the map (target = map) and o3ds_map_carve with params->voxel_size ==
 * map_voxel_size and n_removed == NULL or not -- the shipped configuration: 0.1 m both -- work on the persistent form directly.  The map's
 * o3ds_cloud_size_bound is an upper bound while it is in that form.  An error inside the queued kernels (an internal capacity) is
 * reported by the next call on the map. */
int o3ds_map_insert_scan(o3ds_handle h, o3ds_cloud map, o3ds_cloud scan, const double T[16], double map_voxel_size,
                         const o3ds_crop* map_builder_crop, double max_corr_hint);

#ifdef __cplusplus
}
#endif
#endif /* O3DS_BACKEND_H */

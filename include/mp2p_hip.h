/*
 * mp2p_hip.h -- C ABI of libmp2p_hip.so: MI355X (gfx950) implementation of the
 * mp2p_icp per-iteration hot path (nearest-neighbour correspondence search +
 * Gauss-Newton normal-equation reduction).
 *
 * Plain C, caller-owned buffers, no MRPT / Eigen / torch types.  This is the drop-in
 * boundary: every entry point states the reference interface (file:line, relative to
 * MOLAorg/mp2p_icp v1.8.0) it replaces.  INTEGRATION.md shows the C++ adapter a
 * maintainer would add on the reference side (classes deriving mp2p_icp::Matcher /
 * mp2p_icp::Solver that call these functions).
 *
 * Conventions
 *   pose   : double[12] = R (row-major 3x3) then t (3)                      [CPose3D]
 *   points : SoA float arrays                 [CPointsMap::getPointsBufferRef_{x,y,z}]
 *   return : 0 = MP2P_HIP_OK, <0 = error; mp2p_hip_last_error(ctx) has the text.
 *   Every handle belongs to the context that created it.  A context owns one HIP
 *   stream; calls on one context are serialised by the caller (ICP::align is
 *   single-threaded per ICP object); different contexts are independent.
 *   There is NO CPU fallback: without a HIP device every compute call fails with
 *   MP2P_HIP_ERR_NO_DEVICE.
 */
#ifndef MP2P_HIP_H
#define MP2P_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Bumped whenever the layout of a public struct, the meaning of an argument or a return code changes.
 *   4 (round 6): mp2p_hip_gn_params carries 32 weight blocks (8 before: w_pt2ln / w_pl2pl moved by 384 bytes);
 *                mp2p_hip_abi_check; mp2p_hip_set_tune reports unknown / creation-time knobs as MP2P_HIP_ERR_INVALID.
 * tests/test_abi.py hashes sizeof / offsetof of every public struct against tests/golden/abi_layout.json and fails when
 * the table changes without this number changing. */
#define MP2P_HIP_ABI_VERSION 4

enum
{
    MP2P_HIP_OK            = 0,
    MP2P_HIP_ERR_INVALID   = -1, /* bad argument (what ASSERT_/THROW_EXCEPTION guards upstream) */
    MP2P_HIP_ERR_HIP       = -2, /* a HIP runtime call failed                                  */
    MP2P_HIP_ERR_NOMEM     = -3,
    MP2P_HIP_ERR_CAPACITY  = -4, /* caller-provided output capacity too small                  */
    MP2P_HIP_ERR_NO_DEVICE = -5
};

typedef struct mp2p_hip_ctx   mp2p_hip_ctx;
typedef struct mp2p_hip_map   mp2p_hip_map;   /* a global point layer + its NN index        */
typedef struct mp2p_hip_cloud mp2p_hip_cloud; /* a local point layer                        */
typedef struct mp2p_hip_pairs mp2p_hip_pairs; /* device-resident mp2p_icp::Pairings         */
typedef struct mp2p_hip_mstate mp2p_hip_mstate; /* device-resident MatchState bit-fields    */

/* ---- context ------------------------------------------------------------------------ */
int  mp2p_hip_abi_version(void);
/* The caller's view of the header against the library's: MP2P_HIP_OK iff `header_version` == the library's
 * MP2P_HIP_ABI_VERSION and the sizes of the parameter / result structs the caller was compiled with are the
 * library's (a plugin built against an older header must not get as far as a solver call that reads its weights from
 * the wrong offsets).  Use the macro: MP2P_HIP_ABI_CHECK() at load time (the adapter does, in its MRPT_INITIALIZER). */
int  mp2p_hip_abi_check(int header_version, size_t sizeof_pt2pt_params, size_t sizeof_pt2pl_params,
                        size_t sizeof_gn_params, size_t sizeof_gn_result, size_t sizeof_stats);
#define MP2P_HIP_ABI_CHECK()                                                                             \
    mp2p_hip_abi_check(MP2P_HIP_ABI_VERSION, sizeof(mp2p_hip_pt2pt_params), sizeof(mp2p_hip_pt2pl_params), \
                       sizeof(mp2p_hip_gn_params), sizeof(mp2p_hip_gn_result), sizeof(mp2p_hip_stats))
int  mp2p_hip_device_count(void);
/* stream: a hipStream_t to enqueue on (e.g. the host framework's current stream) or NULL to
 * create a private non-blocking one.  To share the null stream pass hipStreamLegacy, not 0. */
int  mp2p_hip_ctx_create(int device_id, void* hip_stream, mp2p_hip_ctx** out);
void mp2p_hip_ctx_destroy(mp2p_hip_ctx* ctx);
const char* mp2p_hip_last_error(const mp2p_hip_ctx* ctx); /* ctx may be NULL (global text) */
int  mp2p_hip_sync(mp2p_hip_ctx* ctx);
void* mp2p_hip_ctx_stream(mp2p_hip_ctx* ctx);
/* Number of device allocations (hipMalloc) the library has made since it was loaded, over all contexts.  The
 * per-call temporaries of the solvers, matchers and filters live in scratch owned by the context, so the count stays
 * put across steady-state calls of the same sizes (tests/test_gpu_scratch.py asserts it). */
unsigned long long mp2p_hip_debug_alloc_count(void);

/* ---- global map layer: replaces mrpt::maps::NearestNeighborsCapable of the global layer
 *      (nn_prepare_for_3d_queries, Matcher_Points_DistanceThreshold.cpp:92; MRPT's
 *      nanoflann build) with a Morton-sorted multi-level voxel hash (kernel K2). ---------- */
typedef struct
{
    float    cell_size;       /* finest voxel edge [m]; <=0 = choose from the point density   */
    float    target_per_cell; /* density target for the automatic choice (<=0 -> 6)           */
    uint32_t max_levels;      /* 0 -> default (12)                                            */
    uint32_t no_occupancy_bitmap; /* index variants (results are identical; for tests and measurements):
                                     1 = hash probes only (no occupancy bitmaps, no dense voxel directories),
                                     2 = bitmaps but no dense voxel directories,
                                     4 = no dense voxel directory for the finest level only                 */
} mp2p_hip_map_params;

typedef struct
{
    uint64_t n_points;
    float    bbox_min[3], bbox_max[3];
    float    cell_size;
    uint32_t n_levels;
    uint64_t n_cells_total;
    uint64_t n_cells_level0;
    uint64_t hash_capacity;
    uint64_t device_bytes;
    double   build_ms; /* wall time of the last build, device-synchronised */
} mp2p_hip_map_info;

/* x,y,z: HOST pointers (CPointsMap buffers).  The index is rebuilt only by a new upload;
 * the adapter keys uploads on the map object + its modification state. */
int  mp2p_hip_map_upload(mp2p_hip_ctx* ctx, const float* x, const float* y, const float* z,
                         size_t n, const mp2p_hip_map_params* prm, mp2p_hip_map** out);
/* same, from DEVICE pointers (zero-copy path for data already in HBM) */
int  mp2p_hip_map_upload_device(mp2p_hip_ctx* ctx, const float* d_x, const float* d_y,
                                const float* d_z, size_t n, const mp2p_hip_map_params* prm,
                                mp2p_hip_map** out);
void mp2p_hip_map_free(mp2p_hip_ctx* ctx, mp2p_hip_map* map);
int  mp2p_hip_map_get_info(mp2p_hip_ctx* ctx, const mp2p_hip_map* map, mp2p_hip_map_info* out);

/* ---- local layer (the cloud being registered).  Uploaded once per ICP::align; stored
 *      Morton-sorted in its own frame so that 64 consecutive queries stay spatially
 *      coherent under any rigid pose. -------------------------------------------------- */
int  mp2p_hip_cloud_upload(mp2p_hip_ctx* ctx, const float* x, const float* y, const float* z,
                           size_t n, mp2p_hip_cloud** out);
int  mp2p_hip_cloud_upload_device(mp2p_hip_ctx* ctx, const float* d_x, const float* d_y,
                                  const float* d_z, size_t n, mp2p_hip_cloud** out);
void mp2p_hip_cloud_free(mp2p_hip_ctx* ctx, mp2p_hip_cloud* cloud);
size_t mp2p_hip_cloud_size(const mp2p_hip_cloud* cloud);
/* maxLocalPointsPerLayer (Matcher_Points_Base.cpp:222-246): the matchers visit only order[0..n)
 * (host array of distinct original indices), IN THAT ORDER -- it decides the bounding box, which
 * claimant of a contested global point wins and the order of the output.  The reference builds
 * the list with mrpt::random::partial_shuffle over iota(0..maxLocalPoints); the caller passes
 * that list here (the shuffle itself lives in un-vendored MRPT).  order == NULL or n == 0: all
 * points in ascending index (the default).  Stays attached to the cloud until changed. */
int mp2p_hip_cloud_set_visit_order(mp2p_hip_ctx* ctx, mp2p_hip_cloud* cloud, const uint32_t* order,
                                   size_t n);

/* ---- MatchState (Matcher.h:44-70): one byte per point, 0/1 ---------------------------- */
int  mp2p_hip_mstate_create(mp2p_hip_ctx* ctx, size_t n_global, size_t n_local,
                            mp2p_hip_mstate** out); /* all clear */
int  mp2p_hip_mstate_reset(mp2p_hip_ctx* ctx, mp2p_hip_mstate* ms);
void mp2p_hip_mstate_free(mp2p_hip_ctx* ctx, mp2p_hip_mstate* ms);
/* host <-> device copies of the two bit-fields (either pointer may be NULL) */
int  mp2p_hip_mstate_download(mp2p_hip_ctx* ctx, const mp2p_hip_mstate* ms,
                              uint8_t* global_taken, uint8_t* local_taken);
int  mp2p_hip_mstate_upload(mp2p_hip_ctx* ctx, mp2p_hip_mstate* ms, const uint8_t* global_taken,
                            const uint8_t* local_taken);

/* The same two bit-fields PACKED, bit i = bit (i & 63) of word i / 64: the storage of libstdc++'s
 * std::vector<bool>, which pointcloud_bitfield_t::DenseOrSparseBitField wraps (pointcloud_bitfield.h:
 * 46-92).  ceil(n / 64) words each; either pointer may be NULL (that field is left as it is).  A
 * 10 M-point layer is 1.25 MB this way instead of 10 MB -- and a caller whose fields are all clear
 * calls mp2p_hip_mstate_reset (no transfer at all). */
int  mp2p_hip_mstate_upload_bits(mp2p_hip_ctx* ctx, mp2p_hip_mstate* ms, const uint64_t* global_words,
                                 const uint64_t* local_words);
int  mp2p_hip_mstate_download_bits(mp2p_hip_ctx* ctx, const mp2p_hip_mstate* ms, uint64_t* global_words,
                                   uint64_t* local_words);

/* page-locked host memory (device transfers into it run at the full link rate and without a
 * staging copy); for callers that control where their host containers live */
void* mp2p_hip_host_alloc(mp2p_hip_ctx* ctx, size_t bytes);
void  mp2p_hip_host_free(mp2p_hip_ctx* ctx, void* p);

/* ---- Pairings (Pairings.h:84-169), device resident ------------------------------------ */
/* host images, byte-compatible with the reference containers */
typedef struct
{
    uint32_t globalIdx, localIdx;
    float    global_xyz[3];
    float    local_xyz[3]; /* UNtransformed local point */
    float    errorSquareAfterTransformation;
} mp2p_hip_pair_pt2pt; /* = mrpt::tfest::TMatchingPair, 36 B */

typedef struct
{
    double plane[4];    /* TPlane coefs            */
    double centroid[3]; /* plane_patch_t::centroid */
    float  pt_local[3]; /* untransformed           */
    float  _pad;
} mp2p_hip_pair_pt2pl; /* = mp2p_icp::point_plane_pair_t, 72 B */

typedef struct
{
    double ln_base[3];     /* TLine3D::pBase    */
    double ln_director[3]; /* TLine3D::director */
    double pt_local[3];    /* TPoint3D (double) */
} mp2p_hip_pair_pt2ln; /* = mp2p_icp::point_line_pair_t (Pairings.h:61-73), 72 B */

typedef struct
{
    double pl_global[4], c_global[3]; /* plane_patch_t p_global: TPlane coefs, centroid */
    double pl_local[4], c_local[3];   /* plane_patch_t p_local                          */
} mp2p_hip_pair_pl2pl; /* = mp2p_icp::matched_plane_t (Pairings.h:37-48), 112 B */

int  mp2p_hip_pairs_create(mp2p_hip_ctx* ctx, size_t cap_pt2pt, size_t cap_pt2pl,
                           mp2p_hip_pairs** out);
void mp2p_hip_pairs_free(mp2p_hip_ctx* ctx, mp2p_hip_pairs* p);
int  mp2p_hip_pairs_clear(mp2p_hip_ctx* ctx, mp2p_hip_pairs* p); /* out = Pairings() */
/* grow the capacities, keeping the content (std::vector::reserve) */
int  mp2p_hip_pairs_reserve(mp2p_hip_ctx* ctx, mp2p_hip_pairs* p, size_t cap_pt2pt,
                            size_t cap_pt2pl);
/* counts (synchronises the stream; 24 B read back) */
int  mp2p_hip_pairs_counts(mp2p_hip_ctx* ctx, const mp2p_hip_pairs* p, uint64_t* n_pt2pt,
                           uint64_t* n_pt2pl, uint64_t* potential_pairings);
/* materialise the host containers (what ICP::align needs for quality / covariance / logs) */
int  mp2p_hip_pairs_download_pt2pt(mp2p_hip_ctx* ctx, const mp2p_hip_pairs* p,
                                   mp2p_hip_pair_pt2pt* out, size_t capacity, size_t* n_out);
int  mp2p_hip_pairs_download_pt2pl(mp2p_hip_ctx* ctx, const mp2p_hip_pairs* p,
                                   mp2p_hip_pair_pt2pl* out, uint32_t* out_local_idx,
                                   size_t capacity, size_t* n_out);
/* the entries [first, count) only: what ONE matcher call appended (the marks a matcher leaves in the
 * MatchState are exactly the localIdx / globalIdx of these entries, so a host MatchState is brought
 * up to date from this list -- no per-point transfer).  potential may be NULL. */
int  mp2p_hip_pairs_download_pt2pt_from(mp2p_hip_ctx* ctx, const mp2p_hip_pairs* p, size_t first,
                                        mp2p_hip_pair_pt2pt* out, size_t capacity, size_t* n_out,
                                        uint64_t* potential);
int  mp2p_hip_pairs_download_pt2pl_from(mp2p_hip_ctx* ctx, const mp2p_hip_pairs* p, size_t first,
                                        mp2p_hip_pair_pt2pl* out, uint32_t* out_local_idx,
                                        size_t capacity, size_t* n_out, uint64_t* potential);
/* ... and without the count read-back, for a caller that already knows the range (entries
 * [first, first + n) must exist: it read the counts itself) */
int  mp2p_hip_pairs_copy_pt2pt(mp2p_hip_ctx* ctx, const mp2p_hip_pairs* p, size_t first, size_t n,
                               mp2p_hip_pair_pt2pt* out);
/* ... and in three steps, for a caller that has work to do on the indices alone (the MatchState marks a
 * matcher leaves ARE the localIdx / globalIdx of the pairs it appended, Matcher_Points_DistanceThreshold.cpp:
 * 116-120): _begin enqueues the index arrays (into page-locked memory from mp2p_hip_host_alloc), then the
 * records (DMA in chunks into the context's OWN page-locked staging buffer, copied from there into `out` by a helper
 * thread of the context and -- in _end -- by the caller; `out` itself is never registered with the runtime: round 3
 * page-locked it for the duration, which made later pageable copies from neighbouring heap memory fault);
 * _wait_idx returns when the indices are there -- the records are still on the link --; _end returns when every record
 * is in `out`.  One copy at a time per context; a copy never ended is finished by the next _begin / copy call or by
 * mp2p_hip_ctx_destroy, so `out` must stay valid until then. */
int  mp2p_hip_pairs_copy_pt2pt_begin(mp2p_hip_ctx* ctx, const mp2p_hip_pairs* p, size_t first, size_t n,
                                     mp2p_hip_pair_pt2pt* out, uint32_t* idx_local, uint32_t* idx_global);
/* round 6: the same three-step copy with 24 instead of 44 bytes per pair on the link (BASELINE.md section 3 counts the D->H of the pairs):
 * the index arrays first, as above; then only the global point and the squared error (16 bytes per pair); the 36-byte records are ASSEMBLED on the host by the copy threads -- globalIdx / localIdx from the two index
 * arrays (which must be the page-locked ones of mp2p_hip_host_alloc and stay valid until _end), local_xyz read from the CALLER'S own
 * layer arrays at localIdx - local_index_base (localIdx is ascending in a matcher's output: a stream, not a gather).  The caller
 * vouches that local_x/y/z[0 .. n_local) are the coordinates the cloud handle was uploaded from (the host layer's layer cache verifies
 * exactly that per call); an index outside them is reported by _end as MP2P_HIP_ERR_INVALID.  Lists of less than 256 KB, or of more
 * than one round of the staging buffer, take the record form above.  Byte-identical result (tests/test_gpu_boundary_hostpath.py).
 * Measured (bench.py host_boundary, 238 k pairs per step): NOT faster on the boxes of this pool -- the copy-out window is bound by the
 * host threads that write the caller's vector, and assembling a record costs more than copying one; the host layer uses it only on
 * request (MP2P_HIP_HOST_COPY_SOA=1): for hosts with a slow link and fast cores. */
int  mp2p_hip_pairs_copy_pt2pt_begin_soa(mp2p_hip_ctx* ctx, const mp2p_hip_pairs* p, size_t first, size_t n,
                                         mp2p_hip_pair_pt2pt* out, uint32_t* idx_local, uint32_t* idx_global,
                                         const float* local_x, const float* local_y, const float* local_z, size_t n_local,
                                         uint64_t local_index_base);
int  mp2p_hip_pairs_copy_wait_idx(mp2p_hip_ctx* ctx);
int  mp2p_hip_pairs_copy_end(mp2p_hip_ctx* ctx);
int  mp2p_hip_pairs_copy_pt2pl(mp2p_hip_ctx* ctx, const mp2p_hip_pairs* p, size_t first, size_t n,
                               mp2p_hip_pair_pt2pl* out, uint32_t* out_local_idx);
/* paired_pt2ln / paired_pl2pl: produced on the host by Matcher_Point2Line /
 * Matcher_Planes_Normals (not on this path), consumed by the Gauss-Newton solver
 * (optimal_tf_gauss_newton.cpp:184-202, 289-308).  Replaces both lists (n = 0 empties one);
 * mp2p_hip_pairs_clear empties them too.  paired_ln2ln is not supported. */
int  mp2p_hip_pairs_upload_lines_planes(mp2p_hip_ctx* ctx, mp2p_hip_pairs* p,
                                        const mp2p_hip_pair_pt2ln* pt2ln, size_t n_pt2ln,
                                        const mp2p_hip_pair_pl2pl* pl2pl, size_t n_pl2pl);
int  mp2p_hip_pairs_counts_lines_planes(mp2p_hip_ctx* ctx, const mp2p_hip_pairs* p,
                                        uint64_t* n_pt2ln, uint64_t* n_pl2pl);
/* a solver handed host Pairings it did not produce uploads them first */
int  mp2p_hip_pairs_upload(mp2p_hip_ctx* ctx, mp2p_hip_pairs* p, const mp2p_hip_pair_pt2pt* pt2pt,
                           size_t n_pt2pt, const mp2p_hip_pair_pt2pl* pt2pl, size_t n_pt2pl);

/* ---- Matcher_Points_DistanceThreshold::implMatchOneLayer
 *      (Matcher_Points_DistanceThreshold.cpp:48-269) incl. transform_local_to_global
 *      (Matcher_Points_Base.cpp:183-249) and the bounding-box early-out (:73-75).
 *      Kernels K1 (fused), K3, K4.  Appends to `out` like the reference appends to
 *      out.paired_pt2pt; adds n_local*pairingsPerPoint to potential_pairings (:64). ------ */
typedef struct
{
    double   threshold;           /* [m]   Matcher_Points_DistanceThreshold.cpp:43 (formula) */
    double   thresholdAngularDeg; /* [deg] :44 */
    uint32_t pairingsPerPoint;    /* :45 (1 .. 16) */
    int32_t  allowMatchAlreadyMatchedPoints;       /* Matcher_Points_Base.cpp:168-169 */
    int32_t  allowMatchAlreadyMatchedGlobalPoints; /* :171-172 */
    double   bounding_box_intersection_check_epsilon; /* :179-180, default 0.20 */
    /* multi-GPU: this rank's cloud is the slice [local_index_offset, +n) of the whole local
     * layer; claims use the whole-layer index so that "lowest local index wins" holds
     * across ranks.  0 on a single GPU. */
    uint64_t local_index_offset;
    /* tuning (0 = default): first search radius in units of the finest cell */
    float    initial_radius_cells;
    uint32_t queries_per_wave;    /* 0 or 32; 16 and 64 (tile shapes of rounds 1-4) are accepted and served by the 32-query kernels: a tuning value, the lists never depended on it */
    float    group_radius_factor; /* queries of a tile farther than this many search radii
                                     from the first pending one wait for a later pass; 0 = 2.5 */
    uint32_t cell_budget;         /* max voxels of one search box before a coarser level is
                                     used; 0 = 512 */
    float    defer_radius_cells;  /* a query whose search radius exceeds this many cells leaves
                                     its tile for the one-query-per-wave kernel; 0 = 1 m, kept
                                     between 2 and 4 cells */
    int32_t  disable_warm_start;  /* by default a call on the same (map, cloud) as the previous
                                     call of this context seeds every query with its previous
                                     nearest neighbour (exactness is unaffected) */
    uint32_t brick_budget;        /* 4x4x4-voxel bricks of the occupancy bitmap a deferred query
                                     enumerates per pass before it moves to a coarser level; 0 = 128 */
    int32_t  tile_order;          /* ignored since round 2 (kept for layout): the tile kernel now works
                                     on the list of queries the prologue left pending, hard ones first */
    int32_t  multi_search_radius_mode; /* pairingsPerPoint > 1 only.  1: the shipped (TBB) build's search,
                                     nn_radius_search(maxDistSq, ..., k) (Matcher_Points_DistanceThreshold.cpp:
                                     172-177): neighbours with d2 < maxDistSq, the angular term never widens
                                     it.  0: the sequential build's nn_multiple_search (:246-248) followed by
                                     the threshold rule.  Identical when thresholdAngularDeg == 0.  The
                                     adapter and the Python mirror pass 1 (policy: where the two builds of
                                     the reference differ, follow the shipped one -- as for H, g in
                                     optimal_tf_gauss_newton) */
} mp2p_hip_pt2pt_params;

/* ms may be NULL (fresh MatchState with nothing marked, marks discarded). */
int mp2p_hip_match_pt2pt(mp2p_hip_ctx* ctx, const mp2p_hip_map* map, const mp2p_hip_cloud* cloud,
                         const double pose[12], const mp2p_hip_pt2pt_params* prm,
                         mp2p_hip_mstate* ms, mp2p_hip_pairs* out);

/* Split form for a local layer sharded over several GPUs (the map replicated, every rank passing
 * its shard's whole-layer offset in local_index_offset).  mp2p_hip_match_pt2pt == phase1 ; phase2.
 * Between the phases the ranks agree on two things the sequential loop of
 * Matcher_Points_DistanceThreshold.cpp:73-75, 94-121 sees globally: the bounding box of ALL
 * transformed local points and, per global point, the lowest whole-layer local index claiming it:
 *
 *   phase1                      search + this rank's claims
 *   exchange_pack               -> exch = double[8] {-min xyz, max xyz, #records, 0} and the list of
 *                                  this rank's surviving claim records (uint64: sorted global
 *                                  position << 32 | whole-layer local index; padded with ~0)
 *   all-reduce MAX of exch      (RCCL; 64 bytes)
 *   all-gather of list[0 .. max #records)   (skipped when global re-use is allowed)
 *   exchange_unpack(gathered)   box from exch, foreign claims applied
 *   phase2                      winner check + ordered compaction                              */
int mp2p_hip_match_pt2pt_phase1(mp2p_hip_ctx* ctx, const mp2p_hip_map* map,
                                const mp2p_hip_cloud* cloud, const double pose[12],
                                const mp2p_hip_pt2pt_params* prm, mp2p_hip_mstate* ms);
int mp2p_hip_match_pt2pt_phase2(mp2p_hip_ctx* ctx, const mp2p_hip_map* map,
                                const mp2p_hip_cloud* cloud, const mp2p_hip_pt2pt_params* prm,
                                mp2p_hip_mstate* ms, mp2p_hip_pairs* out);
/* exch_dev: device double[8]; list_dev: device uint64[*list_len] with *list_len = visited local
 * points x pairingsPerPoint (valid until the next call on this context).  Any out-pointer may
 * be NULL. */
int mp2p_hip_exchange_pack(mp2p_hip_ctx* ctx, const mp2p_hip_map* map, const mp2p_hip_cloud* cloud,
                           const mp2p_hip_pt2pt_params* prm, void** exch_dev, void** list_dev,
                           size_t* list_len);
/* gathered_dev: n_records uint64 records of all ranks (device; NULL/0 = none) */
int mp2p_hip_exchange_unpack(mp2p_hip_ctx* ctx, const mp2p_hip_map* map, const void* gathered_dev,
                             size_t n_records);
/* device pointer / element count of the claim words (int64, one per global point): the dense
 * alternative to the record exchange (all-reduce MIN) */
void*  mp2p_hip_map_claims_ptr(const mp2p_hip_map* map);
size_t mp2p_hip_map_claims_count(const mp2p_hip_map* map);
/* transformed-local bounding box of the last phase1 on this ctx (6 floats min,max; device) */
void* mp2p_hip_ctx_local_bbox_ptr(mp2p_hip_ctx* ctx);

/* ---- Matcher_Points_InlierRatio::implMatchOneLayer (Matcher_Points_InlierRatio.cpp:40-143):
 *      unbounded nearest neighbour of every visited local point, the round(nTotal * inliersRatio)
 *      smallest d2 kept in the multimap's order (equal d2: the later-visited point first),
 *      unique-global filter in that order, both marks set for every emitted pair.  One GPU. ------ */
typedef struct
{
    double  inliersRatio; /* in (0, 1) */
    int32_t allowMatchAlreadyMatchedPoints;
    int32_t allowMatchAlreadyMatchedGlobalPoints;
    double  bounding_box_intersection_check_epsilon;
} mp2p_hip_inlier_ratio_params;
int mp2p_hip_match_inlier_ratio(mp2p_hip_ctx* ctx, const mp2p_hip_map* map, const mp2p_hip_cloud* cloud,
                                const double pose[12], const mp2p_hip_inlier_ratio_params* prm,
                                mp2p_hip_mstate* ms, mp2p_hip_pairs* out);

/* ---- Matcher_Adaptive::implMatchOneLayer (Matcher_Adaptive.cpp:59-314; parameters
 *      Matcher_Adaptive.h:67-77, initialize() :32-57).  Up to nn = (enableDetectPlanes ?
 *      planeSearchPoints : maxPt2PtCorrespondences) <= 16 neighbours within
 *      absoluteMaxSearchDistance per local point (the first 10 kept), a 50-bin histogram of
 *      everybody's first two squared distances -> threshold, then per local point a pt2pl pairing
 *      (planar neighbours) or up to maxPt2PtCorrespondences pt2pt pairings.  Global marks are read,
 *      not written.  The histogram -> threshold step is mrpt::math::CHistogram +
 *      confidenceIntervalsFromHistogram (un-vendored MRPT): mp2p_hip_adaptive_ci_high restates it,
 *      PARITY UNPINNED; a caller that links MRPT runs _search, computes the threshold itself from
 *      the bins and passes it to _select.  A cloud with a visiting order is refused (the reference
 *      throws for maxLocalPointsPerLayer here).  One GPU. ---------------------------------------- */
#define MP2P_HIP_ADAPTIVE_BINS 50
typedef struct
{
    double   confidenceInterval;       /* in (0,1) */
    double   firstToSecondDistanceMax;
    double   absoluteMaxSearchDistance;
    double   minimumCorrDist;
    int32_t  enableDetectPlanes;
    uint32_t maxPt2PtCorrespondences;
    uint32_t planeSearchPoints;
    uint32_t planeMinimumFoundPoints;  /* >= 3, <= planeSearchPoints */
    double   planeMinimumDistance;
    double   planeEigenThreshold;
    int32_t  allowMatchAlreadyMatchedPoints;
    int32_t  allowMatchAlreadyMatchedGlobalPoints;
    double   bounding_box_intersection_check_epsilon;
} mp2p_hip_adaptive_params;
typedef struct
{
    int32_t  valid;          /* 0: no local point has a neighbour (the reference dereferences an
                                empty optional there); nothing will be paired */
    float    minSqr, maxSqr; /* Matcher_Adaptive.cpp:167-181 */
    uint64_t count;          /* samples added */
    uint64_t bins[MP2P_HIP_ADAPTIVE_BINS];
} mp2p_hip_adaptive_hist;
/* neighbour lists (kept on the context) + histogram; synchronises the stream */
int mp2p_hip_adaptive_search(mp2p_hip_ctx* ctx, const mp2p_hip_map* map, const mp2p_hip_cloud* cloud,
                             const double pose[12], const mp2p_hip_adaptive_params* prm,
                             mp2p_hip_mstate* ms, mp2p_hip_adaptive_hist* hist);
/* upper confidence limit of the histogram as MRPT computes it (host arithmetic; NaN if !valid) */
double mp2p_hip_adaptive_ci_high(const mp2p_hip_adaptive_hist* hist, double confidenceInterval);
/* pairings for the lists of the last _search on this context (same map, cloud, prm) */
int mp2p_hip_adaptive_select(mp2p_hip_ctx* ctx, const mp2p_hip_map* map, const mp2p_hip_cloud* cloud,
                             const mp2p_hip_adaptive_params* prm, double ci_high, mp2p_hip_mstate* ms,
                             mp2p_hip_pairs* out);
/* _search + _ci_high + _select; ci_high_out / hist_out may be NULL */
int mp2p_hip_match_adaptive(mp2p_hip_ctx* ctx, const mp2p_hip_map* map, const mp2p_hip_cloud* cloud,
                            const double pose[12], const mp2p_hip_adaptive_params* prm,
                            mp2p_hip_mstate* ms, mp2p_hip_pairs* out, double* ci_high_out,
                            mp2p_hip_adaptive_hist* hist_out);

/* ---- Matcher_Point2Plane::implMatchOneLayer (Matcher_Point2Plane.cpp:41-114) with the
 *      NearestPlaneCapable::nn_search_pt2pl contract (NearestPlaneCapable.h:33-52)
 *      implemented as: k-NN in radius -> 3x3 covariance -> eigen -> planarity test
 *      (semantics declared in oracle/mp2p_oracle.c; parity unpinned upstream).  K5. ------ */
typedef struct
{
    double   distanceThreshold; /* Matcher_Point2Plane.cpp:38 (formula) */
    double   searchRadius;
    uint32_t knn;                /* 3..16 */
    uint32_t minimumPlanePoints;
    double   planeEigenThreshold;
    int32_t  allowMatchAlreadyMatchedPoints;
    double   bounding_box_intersection_check_epsilon;
    float    initial_radius_cells;
    uint32_t queries_per_wave;
} mp2p_hip_pt2pl_params;

int mp2p_hip_match_pt2pl(mp2p_hip_ctx* ctx, const mp2p_hip_map* map, const mp2p_hip_cloud* cloud,
                         const double pose[12], const mp2p_hip_pt2pl_params* prm,
                         mp2p_hip_mstate* ms, mp2p_hip_pairs* out);

/* ---- NearestPlaneCapable::nn_search_pt2pl (NearestPlaneCapable.h:33-52) for ONE query: the contract
 *      a CMetricMap layer type must offer for the reference's own Matcher_Point2Plane to run on it
 *      (MapToNP, metricmap.cpp:804-822).  `point` is already in the map's frame.  Same k-NN / plane fit
 *      as mp2p_hip_match_pt2pl with distanceThreshold = max_search_distance; prm->searchRadius <= 0
 *      means max_search_distance.  One launch + one 100-byte read-back per call (~0.1 ms): the contract,
 *      not the fast path -- the plugin's own Matcher_Point2Plane class batches the whole layer. ------ */
typedef struct
{
    int32_t found;        /* NearestPlaneResult::pairing.has_value() */
    double  plane[4];     /* pl_global.plane  */
    double  centroid[3];  /* pl_global.centroid */
    float   distance;     /* |plane.distance(point)| */
} mp2p_hip_nearest_plane;
int mp2p_hip_nn_search_pt2pl(mp2p_hip_ctx* ctx, const mp2p_hip_map* map, const float point[3],
                             float max_search_distance, const mp2p_hip_pt2pl_params* prm,
                             mp2p_hip_nearest_plane* out);

/* ---- optimal_tf_gauss_newton (optimal_tf_gauss_newton.cpp:36-372) as called by
 *      Solver_GaussNewton::impl_optimal_pose (Solver_GaussNewton.cpp:42-67).
 *      Kernels K6, K7, K8.  H and g are rebuilt at every inner iteration (TBB-build
 *      meaning, optimal_tf_gauss_newton.cpp:145-146). ------------------------------------ */
enum
{
    MP2P_HIP_KERNEL_NONE         = 0, /* robust_kernels.h:33-43 */
    MP2P_HIP_KERNEL_GEMANMCCLURE = 1,
    MP2P_HIP_KERNEL_CAUCHY       = 2
};
/* Pairings::point_weights blocks a solver call takes (round 5: 32; 8 before) */
#define MP2P_HIP_MAX_WEIGHT_BLOCKS 32

typedef struct
{
    uint32_t maxInnerLoopIterations; /* Solver_GaussNewton 'maxIterations' */
    double   minDelta;               /* 1e-7 (optimal_tf_gauss_newton.h:46-58) */
    double   maxCost;                /* 0 */
    int32_t  kernel;                 /* MP2P_HIP_KERNEL_* */
    double   kernelParam;            /* robustKernelParam (formula) */
    double   w_pt2pt, w_pt2pl;       /* PairWeights */
    int32_t  has_prior;              /* SolverContext::prior */
    double   prior_mean[12];
    double   prior_cov_inv[36];
    /* Pairings::point_weights run-length blocks for pt2pt; 0 blocks = none.  At most MP2P_HIP_MAX_WEIGHT_BLOCKS
     * (one block per layer pair that produced pairings, Matcher_Points_Base.cpp:121-125). */
    uint32_t n_weight_blocks;
    uint64_t weight_block_count[MP2P_HIP_MAX_WEIGHT_BLOCKS];
    double   weight_block_w[MP2P_HIP_MAX_WEIGHT_BLOCKS];
    double   w_pt2ln, w_pl2pl;       /* PairWeights, for the host-produced lists */
} mp2p_hip_gn_params;

typedef struct
{
    double   pose[12];    /* OptimalTF_Result::optimalPose */
    double   H[36], g[6]; /* last assembled normal equations (row-major) */
    double   cost;        /* errNormSqr of the last assembled iteration */
    uint32_t iterations;  /* inner iterations executed */
} mp2p_hip_gn_result;

int mp2p_hip_gn_solve(mp2p_hip_ctx* ctx, const mp2p_hip_pairs* pairs, const double pose0[12],
                      const mp2p_hip_gn_params* prm, mp2p_hip_gn_result* out);

/* split form (sharded pairs): begin ; { accumulate ; <all-reduce sums> ; step } x N ; end */
#define MP2P_HIP_GN_NSUMS 48 /* 17 pt2pt + 28 pt2pl sums, padded */
int   mp2p_hip_gn_begin(mp2p_hip_ctx* ctx, const mp2p_hip_pairs* pairs, const double pose0[12],
                        const mp2p_hip_gn_params* prm);
int   mp2p_hip_gn_accumulate(mp2p_hip_ctx* ctx);
void* mp2p_hip_gn_sums_ptr(mp2p_hip_ctx* ctx); /* device double[MP2P_HIP_GN_NSUMS] */
int   mp2p_hip_gn_step(mp2p_hip_ctx* ctx);
int   mp2p_hip_gn_end(mp2p_hip_ctx* ctx, mp2p_hip_gn_result* out);

/* ---- next #1: optimal_tf_horn (optimal_tf_horn.cpp:77-252) with WeightParameters
 *      (WeightParameters.h:34-72) through visit_correspondences (visit_correspondences.h:38-212)
 *      and eval_centroids_robust (Pairings.cpp:68-110): point pairings + plane-to-plane normals
 *      (paired_pl2pl as uploaded with mp2p_hip_pairs_upload_lines_planes), point_weights blocks,
 *      scale outlier detector, robust kernel.  paired_ln2ln is not supported; point-to-plane /
 *      point-to-line pairings must be converted first, as Solver_Horn does (Solver_Horn.cpp:51-55).
 *      Where the reference throws (all weights 0, a visited pairing with weight <= 0, a robust
 *      kernel without currentEstimateForRobust, no more point pairings than outliers) the call
 *      fails with MP2P_HIP_ERR_INVALID. ---------------------------------------------------------- */
typedef struct
{
    int32_t use_scale_outlier_detector; /* WeightParameters.h:41 */
    double  scale_outlier_threshold;    /* 1.20 */
    double  w_pt2pt, w_ln2ln, w_pl2pl;  /* pair_weights used by this solver */
    int32_t robust_kernel;              /* MP2P_HIP_KERNEL_* */
    double  robust_kernel_param;
    int32_t has_current_estimate;
    double  current_estimate[12]; /* currentEstimateForRobust */
    /* Pairings::point_weights (count, weight) blocks; 0 = one block of weight 1; at most MP2P_HIP_MAX_WEIGHT_BLOCKS */
    uint32_t      n_weight_blocks;
    const size_t* weight_block_count;
    const double* weight_block_w;
} mp2p_hip_horn_params;
typedef struct
{
    double   pose[12];
    int32_t  solved;     /* 0: fewer than 3 pairings (optimal_tf_horn.cpp:98) */
    uint64_t n_outliers; /* OptimalTF_Result::outliers.point2point.size() */
} mp2p_hip_horn_result;
int mp2p_hip_horn_solve_wp(mp2p_hip_ctx* ctx, const mp2p_hip_pairs* pairs, const mp2p_hip_horn_params* wp,
                           mp2p_hip_horn_result* out);
/* OptimalTF_Result::outliers of the last mp2p_hip_horn_solve[_wp] on this context: one byte per
 * point pairing (1 = discarded by the scale detector) */
int mp2p_hip_horn_outlier_flags(mp2p_hip_ctx* ctx, uint8_t* flags_host, size_t n);
/* default WeightParameters, point pairings only */
int mp2p_hip_horn_solve(mp2p_hip_ctx* ctx, const mp2p_hip_pairs* pairs, double w_pt2pt,
                        double pose_out[12], int32_t* solved);
/* pt2ln_pl_to_pt2pt (pt2ln_pl_to_pt2pt.cpp:47-113): the point-to-plane and point-to-line pairings
 * of `in`, as (closest point on the plane / line under `guess`, local point) point pairings from
 * the largest distance down to 25 % of it (at least 3), appended to `out` -- which must be another,
 * cleared handle with pt2pt capacity >= the two list lengths: the reference starts from an empty
 * Pairings, the point pairings of `in` are not carried over. */
int mp2p_hip_pairs_pt2ln_pl_to_pt2pt(mp2p_hip_ctx* ctx, const mp2p_hip_pairs* in, const double guess[12],
                                     mp2p_hip_pairs* out);

/* ---- multi-GPU: RCCL inside the boundary (SURVEY.md sections 8b, 8e (ii)).  One process per GPU, one
 *      context per process.  The local layer is sharded in contiguous ranges of the visiting order
 *      (every rank passes its shard's whole-layer offset in mp2p_hip_pt2pt_params::local_index_offset),
 *      the map and its index are replicated.  The reference has no distributed code; a C++
 *      host shards with:
 *
 *        rank 0: mp2p_hip_comm_get_unique_id(id); ship the 128 bytes to the other ranks (MPI, a file, ...)
 *        every rank: mp2p_hip_comm_init(ctx, id, rank, nranks);           // ncclCommInitRank
 *        per ICP iteration: mp2p_hip_step_sharded(ctx, map, shard, pose, ...)  // same pose out on all ranks
 *
 *      mp2p_hip_step_sharded = phase1 ; ncclAllReduce(MAX, f64[8]) ; ncclAllGather(claim records) ;
 *      phase2 ; { accumulate ; ncclAllReduce(SUM, f64[48]) ; step } x maxInnerLoopIterations -- all on the
 *      context's stream.  librccl is opened at run time, by mp2p_hip_comm_init only. ------------------ */
#define MP2P_HIP_COMM_ID_BYTES 128 /* sizeof(ncclUniqueId) */
int mp2p_hip_comm_get_unique_id(void* id_out /* MP2P_HIP_COMM_ID_BYTES */);
/* pre-flight of mp2p_hip_comm_init without side effects: MP2P_HIP_OK iff librccl can be loaded here and the context has no
 * communicator yet (every rank checks this before ANY rank calls mp2p_hip_comm_init: a rank that fails on its own would leave
 * its peers blocked inside ncclCommInitRank).  No counterpart in the reference. */
int mp2p_hip_comm_available(mp2p_hip_ctx* ctx);
int mp2p_hip_comm_init(mp2p_hip_ctx* ctx, const void* unique_id, int rank, int nranks);
/* caller-provided collectives instead of RCCL (another transport; tests).  Both act IN PLACE / into recv
 * on DEVICE buffers and must be ordered with `stream`; return 0 on success.
 *   allreduce(user, buf, n_doubles, op (0 sum, 1 max), stream)
 *   allgather(user, send, recv, n_u64_per_rank, stream)        recv holds nranks * n words, by rank */
typedef int (*mp2p_hip_allreduce_fn)(void* user, void* dev_buf, size_t n, int op, void* stream);
typedef int (*mp2p_hip_allgather_fn)(void* user, const void* dev_send, void* dev_recv, size_t n, void* stream);
int mp2p_hip_comm_init_hooks(mp2p_hip_ctx* ctx, int rank, int nranks, mp2p_hip_allreduce_fn allreduce,
                             mp2p_hip_allgather_fn allgather, void* user);
int mp2p_hip_comm_destroy(mp2p_hip_ctx* ctx);
int mp2p_hip_comm_rank(const mp2p_hip_ctx* ctx);
int mp2p_hip_comm_size(const mp2p_hip_ctx* ctx);
/* the path's all-reduce on its own (device double[n], in place, on the context's stream) */
int mp2p_hip_comm_allreduce_f64(mp2p_hip_ctx* ctx, void* dev_buf, size_t n, int op_max);
/* one outer ICP iteration of a sharded local layer: Matcher_Points_DistanceThreshold (pairingsPerPoint
 * 1) + Solver_GaussNewton (Pairings cleared first); without a communicator it is the single-GPU step
 * in one call.  `pairs` receives THIS rank's pairings, `out` the pose every rank agrees on
 * (sums re-associated across ranks: equal to ~1e-12 relative).  *redone = 1 when the claim-record lists
 * outgrew their predicted length and the iteration was repeated with the exact one (may be NULL). */
int mp2p_hip_step_sharded(mp2p_hip_ctx* ctx, const mp2p_hip_map* map, const mp2p_hip_cloud* cloud,
                          const double pose[12], const mp2p_hip_pt2pt_params* prm, const mp2p_hip_gn_params* gn,
                          mp2p_hip_pairs* pairs, mp2p_hip_gn_result* out, int32_t* redone);

/* the same for Matcher_Point2Plane + Solver_GaussNewton (BASELINE config C3 sharded): every local point is paired on its
 * own (no uniqueness filter, Matcher_Point2Plane.cpp:87-90), so the shards exchange only the layer's bounding box
 * (:59-66; one all-reduce MAX of 8 doubles) and, per inner Gauss-Newton iteration, the 48 sums.  local_index_offset =
 * whole-layer index of this shard's first local point.  Without a communicator: matcher + solver in one call. */
int mp2p_hip_step_sharded_pt2pl(mp2p_hip_ctx* ctx, const mp2p_hip_map* map, const mp2p_hip_cloud* cloud,
                                const double pose[12], const mp2p_hip_pt2pl_params* prm, uint64_t local_index_offset,
                                const mp2p_hip_gn_params* gn, mp2p_hip_pairs* pairs, mp2p_hip_gn_result* out);

/* ---- instrumentation ------------------------------------------------------------------ */
typedef struct
{
    double   ms_nn;      /* K1+K3 (transform + search + claims), last match call */
    double   ms_nn_tile, ms_nn_single; /* the tile kernel and the one-query-per-wave kernel */
    double   ms_compact; /* K4 */
    double   ms_gn;      /* all inner iterations of the last gn_solve */
    uint64_t nn_tiles, nn_passes, nn_cells_visited, nn_candidates_tested, nn_points_staged;
    uint64_t nn_queries, nn_unresolved_after_first_pass;
    uint64_t nn_max_candidates_one_tile, nn_max_passes_one_tile;
    uint64_t nn_tile_ticks_sum, nn_tile_ticks_max; /* 100 MHz wall_clock64 ticks per tile */
    uint64_t nn_coop_passes; /* queries deferred to the one-query-per-wave kernel */
    uint64_t nn_single_queries, nn_single_passes, nn_single_cells, nn_single_candidates;
    uint64_t nn_single_max_candidates;
    uint64_t nn_tile_ticks_hist[24]; /* log2 bins */
    uint64_t nn_single_ticks_sum, nn_single_ticks_max; /* 100 MHz ticks per deferred query */
    uint64_t nn_single_max_passes, nn_single_max_cells;
    /* one-query-per-lane kernel (runs first): its time, the queries it searched itself, the candidates
     * and voxels they cost, the queries it handed on to the tile kernel, and those a warm start
     * finished without any search */
    double   ms_nn_lane;
    uint64_t nn_lane_searched, nn_lane_candidates, nn_lane_voxels, nn_lane_pending, nn_lane_skipped;
    /* reserved, always 0: the counters of round 3's one-launch search kernel (nn_wave_kernel), which was measured
     * slower on both bench scenes and removed from the library in round 4; kept so that the struct keeps its layout */
    uint64_t nn_wave_path;
    /* round 5 (nn_seltile.hip; the first two of round 3's reserved words): occupied voxels the tiles listed from the
     * occupancy bricks, and those the matrix-pipe selection kept (resolved + staged) */
    uint64_t nn_sel_voxels_listed, nn_sel_voxels_needed;
    uint64_t nn_wave_inserts, nn_wave_overflows, nn_wave_rounds;
    uint64_t nn_wave_toobig;
    uint64_t nn_wave_phase_ticks[6];
    /* Matcher_Point2Plane certificate (round 3), cumulative since the context was created and refreshed by a
     * get_stats after a point-to-plane call at profiling level 1 or 2: queries whose previous neighbour list was
     * proven to be their k nearest again (no search), and queries that went through the search */
    uint64_t pl_certified, pl_searched;
} mp2p_hip_stats;
/* ---- mp2p_icp::covariance (mp2p_icp/src/covariance.cpp:29-141, ICP.cpp:334-337):
 *      H = J^T J of the stacked error vector w.r.t. (x, y, z, yaw, pitch, roll), J by central
 *      finite differences (CovarianceParameters::finDif_xyz / finDif_angles, 1e-7 each), cov = H^-1.
 *      One pass over the device-resident Pairings (pt2pt, pt2pl, pt2ln, pl2pl; ln2ln unsupported).
 *      No pairings: cov = 1e6 * I (covariance.cpp:33-39).  *positive_definite = 0 and cov = NaN
 *      when the Cholesky factorisation of H fails.  H_out may be NULL. --------------------------- */
int mp2p_hip_covariance(mp2p_hip_ctx* ctx, const mp2p_hip_pairs* pairs, const double pose[12],
                        double finDif_xyz, double finDif_angles, double H_out[36], double cov_out[36],
                        int32_t* positive_definite);

/* ---- FilterDecimateVoxels (mp2p_icp_filters/src/FilterDecimateVoxels.cpp:107-381): the voxel
 *      decimation that produces the "decimated" layer every demo pipeline matches on (SURVEY.md
 *      section 8f #2).  One input layer (the reference's FirstPoint mode accepts several: the caller
 *      concatenates them in layer order).  Voxel = (int32)(coordinate / resolution) per axis.
 *      Output order = ascending (cx, cy, cz), i.e. the reference's std::map mode; its default
 *      tsl::robin_map mode visits the same voxels in an order that depends on the (un-vendored)
 *      table.  DecimateMethod::RandomPoint is not offered (mrpt::random stream). ----------------- */
enum
{
    MP2P_HIP_DECIMATE_FIRST_POINT        = 0, /* lowest point index of the voxel              */
    MP2P_HIP_DECIMATE_CLOSEST_TO_AVERAGE = 1, /* point closest to the fp32 voxel mean         */
    MP2P_HIP_DECIMATE_VOXEL_AVERAGE      = 2  /* the fp32 voxel mean itself                   */
};
typedef struct
{
    float   voxel_filter_resolution; /* [m] */
    int32_t decimate_method;         /* MP2P_HIP_DECIMATE_* */
    int32_t has_flatten_to;          /* emit one point per (cx, cy) column with z = flatten_to */
    float   flatten_to;
} mp2p_hip_decimate_params;
/* host arrays in, host arrays out (capacity n each; out_src_index may be NULL: index of the
 * input point that was kept, 0xFFFFFFFF for an average) */
int mp2p_hip_filter_decimate_voxels(mp2p_hip_ctx* ctx, const float* x, const float* y, const float* z,
                                    size_t n, const mp2p_hip_decimate_params* prm, float* out_x,
                                    float* out_y, float* out_z, uint32_t* out_src_index,
                                    size_t* n_out);
/* the same on device arrays (inputs and outputs in HBM, e.g. to upload the result as a layer
 * with mp2p_hip_cloud_upload_device without a PCIe round trip) */
int mp2p_hip_filter_decimate_voxels_device(mp2p_hip_ctx* ctx, const float* d_x, const float* d_y,
                                           const float* d_z, size_t n,
                                           const mp2p_hip_decimate_params* prm, float* d_out_x,
                                           float* d_out_y, float* d_out_z, uint32_t* d_out_src_index,
                                           size_t* n_out);

/* 0 = off; 1 = bracket the kernels with hipEvents (ms_* fields; each call then ends with a
 * stream synchronisation); 2 = additionally collect the device counters (nn_* fields, slower);
 * 3 = only the two events around the search kernels (ms_nn; the cheapest timing);
 * 4 = 3 + a {start, end} timestamp (100 MHz ticks) per workgroup of the two search kernels of
 *     Matcher_Points_DistanceThreshold, for occupancy-over-time plots (tools/timeline_probe.py). */
int mp2p_hip_set_profiling(mp2p_hip_ctx* ctx, int enable);
/* measurement knobs of a context at run time: the syntax of the environment variable MP2P_HIP_TUNE ("name=value,...",
 * read once when the context is created; csrc/common.hpp lists the knobs).  Every setting computes the same results,
 * except tile_sol != 0 (timing-only launches of the search, no results): that knob is accepted ONLY here, only while
 * profiling is on (mp2p_hip_set_profiling != 0), never from the environment, and a match call made under it appends
 * NOTHING to the caller's list (only potential_pairings grows): the records of a timing launch are never compacted.  An unknown or malformed knob, or one that is
 * consumed when the context is created (copy_stage_mb, copy_chunk_kb, dir_budget_mb, spin_us, sync_spin, pipelines),
 * is refused with MP2P_HIP_ERR_INVALID and nothing of the string is applied.  No counterpart in the reference. */
int mp2p_hip_set_tune(mp2p_hip_ctx* ctx, const char* settings);
/* the timestamps of the last match call at profiling level 4: 2 uint64 per record, the tile
 * kernel's workgroups first, then the one-query-per-wave kernel's; ticks_host may be NULL to
 * query the record counts */
int mp2p_hip_get_timeline(mp2p_hip_ctx* ctx, uint64_t* ticks_host, size_t cap_records,
                          size_t* n_tile_records, size_t* n_single_records);
int mp2p_hip_get_stats(mp2p_hip_ctx* ctx, mp2p_hip_stats* out);

#ifdef __cplusplus
}
#endif
#endif /* MP2P_HIP_H */

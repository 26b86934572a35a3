/* mgs.h -- C ABI of the MI355X-native 3D Gaussian Splatting render path (libmgs.so).
 *
 * Drop-in boundary.  The reference (Maxwell-Zhao/RoboSimGS) has no FFI or operator
 * interface of its own for this path: it delegates 3DGS to Nerfstudio in prose
 * (/root/reference/README.md:75) and has not released the stage that renders the
 * exported .ply (README.md:29, :84-85).  The interface a maintainer would bind is
 * therefore the operator set of the rasteriser nerfstudio's `splatfacto` calls
 * (gsplat 1.x).  Each entry point below names the
 * operator it stands in for.  The camera convention at the boundary (OpenCV
 * world-to-camera `viewmat`, pixel-unit `K`) is the one the reference's
 * Articulation/utils/nerf2physic_utils.py:10-23 (`project_3d_to_2d`) encodes.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer to fp32 / int32 / int64 data, contiguous,
 *    row-major, unless marked "host";
 *  - the caller owns every buffer, workspace included; nothing is allocated, freed or
 *    retained by the library;
 *  - all work is enqueued on `stream` (a hipStream_t); no entry point synchronises,
 *    so every call is capturable in a hipGraph;
 *  - return value: 0 ok, <0 MGS_ERR_*, >0 a hipError_t from a launch;
 *  - data-dependent sizes (the number of tile intersections) stay on the device:
 *    `n_isect` is a device scalar and buffers are sized by a caller-chosen capacity;
 *    overflow raises bit MGS_STATUS_ISECT_OVERFLOW in the device `status` word and the
 *    frame must be re-rendered with a larger capacity (nothing is written out of
 *    bounds);
 *  - stateless and re-entrant; one process per GPU or several host threads with their
 *    own streams are both fine.
 */
#ifndef MGS_H_
#define MGS_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Bumped whenever a parameter list changes: a caller built against another header must not load this
 * library (robosimgs_amd/_lib.py asserts mgs_version() == the MGS_VERSION it was written for).
 * 100 round 1; 200 round 2 (seed / splats / tile_group_order arguments); 300 round 3; 400 round 4 (forward
 * checkpoints + segmented backward, batched training entry points, debug hooks out of the production build);
 * 410 round 4 (the radius rule as a policy: radii_y / radius_rule arguments, MGS_BIN_* / MGS_FRAMES_RADIUS_* flags,
 * one more field in the training state; 420: the dataset frame as an output of the raster forward, ds_* arguments). */
#define MGS_VERSION 430

#define MGS_OK 0
#define MGS_ERR_INVALID_ARGUMENT (-1)
#define MGS_ERR_WORKSPACE_TOO_SMALL (-2)
#define MGS_ERR_UNSUPPORTED (-3)

#define MGS_STATUS_ISECT_OVERFLOW 1u

/* mgs_rasterize_fwd flags */
#define MGS_RASTER_EXPECTED_LAST 1 /* last channel leaves as channel / max(alpha, 1e-10): "ED" */
#define MGS_RASTER_LATENCY 2       /* one wave per 8x8 block instead of one per tile (<= 4 channels):
                                      -19 % time for a launch that has the GPU to itself, +7 % vector
                                      instructions -- for single frames and training steps, not for
                                      several frames in flight; same pixels bit for bit */

/* mgs_render_frames flags: the two above, and */
#define MGS_FRAMES_CLASSIC_BOUNDS 4 /* bin into gsplat's classic mean +- radius tile rectangles instead of the tightened
                                       ones: same pixels bit for bit, but n_isect / the overflow status then count the
                                       classic lists (a caller that sized its capacity with them) */

#define MGS_FRAMES_RADIUS_OPACITY_AWARE 8 /* project with MGS_RADIUS_OPACITY_AWARE (below) instead of the classic rule */

/* The radius rule ("make the radius rule a compile-time policy so the tighter one can be benchmarked").
 * Inside the library the rule is a template constant of the projection kernels; both instantiations ship and the
 * caller picks one per call.
 *   MGS_RADIUS_CLASSIC        gsplat 1.4, A.2 step 5 (the semantics of this build's parity claim): one radius
 *                             ceil(3 sqrt(lambda_1)) per Gaussian, tile rectangle of the square mean +- radius.
 *   MGS_RADIUS_OPACITY_AWARE  gsplat >= 1.5: per-axis extents radii = ceil(e sqrt(Sigma_xx)), radii_y = ceil(e sqrt(Sigma_yy))
 *                             with e = min(3.33, sqrt(2 ln(255 opacity))) (opacity x compensation when antialiased; e = 3.33
 *                             without opacities); Gaussians of opacity < 1/255 are culled; a Gaussian is culled by
 *                             radius_clip only when BOTH extents are <= radius_clip; screen cull and tile rectangle per
 *                             axis.  Changes n_isect and, at the edge of opaque Gaussians (3.33 > 3 sigma) and in the
 *                             corners of the classic square, pixels. */
#ifndef MGS_RADIUS_CLASSIC
#define MGS_RADIUS_CLASSIC 0
#define MGS_RADIUS_OPACITY_AWARE 1
#endif
/* mgs_project_color_fwd bin_flags */
#define MGS_BIN_TIGHT 1                 /* tightened tile rectangles in the binning seed (mgs_isect_tiles) */
#define MGS_BIN_RADIUS_OPACITY_AWARE 2  /* MGS_RADIUS_OPACITY_AWARE instead of MGS_RADIUS_CLASSIC */

/* mgs_rasterize_bwd_det flags */
#define MGS_RASTER_BWD_SPLAT_SLOTS 2  /* the splat records carry the pairs' record slots (mgs_isect_tiles: splat_slots): pair_info is
                                         read by the reduce only, the raster kernel gathers nothing but the record */
#define MGS_RASTER_BWD_RECORDS_ONLY 1 /* stop after the raster kernel: the workspace then holds one record and one flag
                                         per (tile, Gaussian) slot and the v_* outputs are not touched (measurement
                                         of the raster kernel alone; a consumer that sums the records itself) */

#define MGS_TILE_SIZE 16
#define MGS_MAX_CHANNELS 32

typedef void *mgs_stream_t; /* hipStream_t */

int mgs_version(void);
/* Thread-local, human-readable description of the last non-zero return value. */
const char *mgs_last_error_string(void);
#ifdef MGS_DEBUG_HOOKS
/* Test and measurement hooks: compiled into libmgs_debug.so ONLY (-DMGS_DEBUG_HOOKS; tests/ and scripts/ load that
 * build).  They are process-global state, which the shipped libmgs.so does not have.
 * mgs_debug_set_raster_cull: 0 disables the raster forward's exact per-quadrant cull (tests prove it never changes a
 *   pixel).  mgs_debug_set_raster_opts: scheduling of the raster forward; bit 0 = issue priority by tile-list length,
 *   bit 1 = honour MGS_RASTER_LATENCY, bit 2 = force the per-block kernel, bits 8.. = KiB of padding LDS (default 3);
 *   never changes a pixel.  mgs_debug_set_sort_opts: bit 0 = one-sweep (decoupled look-back) radix passes, bit 2 =
 *   radix partition by tile where the direct path applies, bit 3 = counters per 2^(bits 4..6) tiles (default 0);
 *   same lists bit for bit. */
void mgs_debug_set_raster_cull(int enabled);
void mgs_debug_set_raster_opts(int opts);
void mgs_debug_set_sort_opts(int opts);
#endif

/* -------------------------------------------------------------------------------------
 * Projection  (gsplat `fully_fused_projection` forward, packed=False, one camera)
 *   means[N,3] quats[N,4] (wxyz, normalised inside) scales[N,3] (post-exp)
 *   viewmat[4,4] K[3,3]: device pointers, row-major.
 *   out: radii[N] i32 (0 = culled), means2d[N,2], depths[N], conics[N,3],
 *        compensations[N] (nullable; sqrt(max(0, det_orig/det_blur))).
 *   Culled Gaussians get zeros in every output.
 *   radius_rule: MGS_RADIUS_CLASSIC (opacities / radii_y not read or written, may be NULL) or
 *   MGS_RADIUS_OPACITY_AWARE: opacities[N] nullable (as gsplat >= 1.5's optional argument; multiplied by the compensation
 *   iff compensations is given, gsplat's calc_compensations), radii = extent along x, radii_y[N] = extent along y.
 * ----------------------------------------------------------------------------------- */
int mgs_projection_fwd(int n, const float *means, const float *quats, const float *scales,
                       const float *viewmat, const float *K, int width, int height,
                       float eps2d, float near_plane, float far_plane, float radius_clip,
                       int32_t *radii, float *means2d, float *depths, float *conics,
                       float *compensations, const float *opacities, int radius_rule,
                       int32_t *radii_y, mgs_stream_t stream);

/* Projection backward (gsplat `fully_fused_projection` backward).
 *   v_means2d[N,2] v_depths[N] v_conics[N,3] v_compensations[N] (nullable) are the
 *   incoming cotangents; v_means[N,3] v_quats[N,4] v_scales[N,3] are ACCUMULATED into
 *   (+=) so several cameras can be summed; pass zeroed buffers for a single camera.
 *   v_viewmat[16] (nullable) is accumulated with atomics. */
int mgs_projection_bwd(int n, const float *means, const float *quats, const float *scales,
                       const float *viewmat, const float *K, int width, int height,
                       float eps2d, const int32_t *radii, const float *conics,
                       const float *compensations, const float *v_means2d,
                       const float *v_depths, const float *v_conics,
                       const float *v_compensations, float *v_means, float *v_quats,
                       float *v_scales, float *v_viewmat, mgs_stream_t stream);

/* -------------------------------------------------------------------------------------
 * Spherical harmonics  (gsplat `spherical_harmonics` forward / backward)
 *   dirs[N,3] (un-normalised), coeffs[N,K,3] with K = coeff_stride >= (degree+1)^2,
 *   masks[N] (nullable, uint8/bool: 0 = skip and write zeros), colors[N,3] = sum_k Y_k c_k
 *   (no +0.5, no clamp: the caller applies them, as splatfacto does).
 * ----------------------------------------------------------------------------------- */
int mgs_sh_fwd(int n, int degree, int coeff_stride, const float *dirs, const float *coeffs,
               const uint8_t *masks, float *colors, mgs_stream_t stream);
/* v_coeffs[N,K,3] is written (zeros beyond the active degree and for masked rows);
 * v_dirs[N,3] nullable, written. */
int mgs_sh_bwd(int n, int degree, int coeff_stride, const float *dirs, const float *coeffs,
               const uint8_t *masks, const float *v_colors, float *v_coeffs, float *v_dirs,
               mgs_stream_t stream);

/* -------------------------------------------------------------------------------------
 * Fused projection + view-dependent colour (the render path's first kernel).
 * Equivalent to mgs_projection_fwd, then dirs = means - campos, mgs_sh_fwd masked by
 * radii > 0, then feat = max(colour + 0.5, 0); SH coefficients of culled Gaussians are
 * never read.  feats[N,feat_stride]: channels 0..2 rgb; if feat_stride == 4 channel 3
 * receives the camera-space depth (the "RGB+D"/"RGB+ED" layout).  If `opac_out` is
 * non-null it receives opacities * compensation (rasterize_mode="antialiased").
 * splats[N,12] (nullable) additionally receives one packed 48-byte record per VISIBLE Gaussian
 * (rows of culled Gaussians, radii == 0, are left untouched: they never enter a tile list),
 *   { mean2d.x, mean2d.y, conic.a, conic.b | conic.c, opacity, f0, f1 | f2, f3, 0, 0 }
 * (opacity already multiplied by the compensation when antialiased; f = feats, zero padded):
 * the raster kernels gather ONE record per list entry instead of four separate arrays
 * (means2d, conics, opacities, feats), which cuts their cache-line traffic: 266 -> 240 us.
 * bin_info[N,2] u32 + bin_sums[ceil(N/64)] u32 (nullable, together): the SEED of the tile binning
 * at tile size 16 -- per Gaussian {x0 | y0 << 10 | max(w,1) << 20, tile count} of its tile
 * rectangle (bin_flags & MGS_BIN_TIGHT: tightened to the tiles it can reach with alpha >= 1/255, see
 * mgs_isect_tiles) and the count sum of every 64 consecutive Gaussians.  Handed to
 * mgs_isect_tiles as seed_info / seed_sums they save the binning a pass and a launch.
 * radii / means2d / conics / feats may each be NULL when BOTH splats and bin_info are given (an
 * inference frame: the raster gathers from splats, the seeded binning reads bin_info and depths;
 * 36 of 84 MB of stores per 1 M Gaussians fall away).  depths is always written.
 * bin_flags & MGS_BIN_RADIUS_OPACITY_AWARE: project with that radius rule (opacity x compensation when opac_out is
 * given); radii_y[N] then receives the extents along y beside radii (x) -- required whenever radii is given, else NULL.
 * ----------------------------------------------------------------------------------- */
int mgs_project_color_fwd(int n, const float *means, const float *quats, const float *scales,
                          const float *opacities, int sh_degree, int coeff_stride,
                          const float *sh_coeffs, const float *viewmat, const float *K,
                          int width, int height, float eps2d, float near_plane,
                          float far_plane, float radius_clip, int32_t *radii, float *means2d,
                          float *depths, float *conics, float *opac_out, int feat_stride,
                          float *feats, float *splats, int bin_flags, uint32_t *bin_info,
                          uint32_t *bin_sums, int32_t *radii_y, mgs_stream_t stream);

/* -------------------------------------------------------------------------------------
 * Tile binning  (gsplat `isect_tiles` with sort=True + `isect_offset_encode`, one camera)
 *
 * Produces the depth-ordered per-tile lists: flatten_ids[n_isect] (Gaussian index),
 * tile_ids[n_isect] (ty*tile_w+tx, ascending) and tile_offsets[n_tiles+1] (first sorted
 * index of each tile, last entry = n_isect).  Ordering is identical to a stable sort of
 * the 64-bit keys (tile << 32 | float_bits(depth)) emitted in Gaussian-index order.
 * isect_ids[capacity] (nullable) receives those int64 keys, with `cam_id` folded in above
 * the tile bits exactly as gsplat encodes them.
 *
 * Internally (csrc/binning.hip, tile_sort.hip): a counting sort of the (tile, Gaussian) pairs on
 * groups of four tiles with LDS-resident counters (histogram, column scan, scatter), then one
 * workgroup per tile orders its list by (depth bits, Gaussian index).
 *
 * conics[N,3], opacities[N] (both nullable): when given, each Gaussian's tile rectangle is
 * tightened to the tiles holding a pixel centre it can reach with alpha >= 1/255 (bounding
 * box of the ellipse sigma <= ln(255 opacity), intersected with the classic mean +- radius
 * rectangle).  The dropped pairs fail the raster's alpha test at every pixel, so images and
 * gradients are bit-identical while the lists are ~27 % shorter at 1 M Gaussians / 1080p;
 * tiles_per_gauss / n_isect then count the tightened lists.  NULL: gsplat's classic lists.
 *
 * Workspace: call with workspace == NULL to get the byte count in *workspace_bytes.
 * tiles_per_gauss[N] nullable.  n_isect, status: device uint32 scalars, both OVERWRITTEN by
 * every call (status = MGS_STATUS_ISECT_OVERFLOW if n_isect > isect_capacity, else 0): the caller
 * never has to clear them.
 * pair_info[N,4] (nullable): per Gaussian {slot_base, x0, y0, w | h << 16} of its tile
 * rectangle; the pair (g, tile (tx,ty)) owns slot slot_base + (ty - y0) * w + (tx - x0) in
 * [0, n_isect); slot bases ascend with the Gaussian index.  Consumed by
 * mgs_rasterize_bwd_det.
 * tile_group_order[ceil(n_tiles / 4)] (nullable): launch order for the raster kernels -- the
 * groups of four consecutive tiles {4g .. 4g+3} by falling total list length (longest first; a
 * permutation of the group indices).  Pass it to mgs_rasterize_fwd / mgs_rasterize_bwd_det; it
 * changes their schedule only, never a result.
 * seed_info / seed_sums (nullable, together): bin_info / bin_sums as written by
 * mgs_project_color_fwd for the same camera, tile grid and tight / classic choice (the choice is the
 * seed's: conics / opacities are ignored then); when given, tile_size must be MGS_TILE_SIZE,
 * means2d / radii / conics / opacities are not read (and may be NULL) and seed_sums -- 16-byte aligned --
 * is overwritten (scanned in place).
 * splat_slots (nullable, with pair_info): the packed records splats[N,12] mgs_project_color_fwd wrote for this camera
 * (16-byte aligned).  Words 10 and 11 of every binned Gaussian's record -- padding as far as the forward is concerned --
 * receive {slot_base, x0 | y0 << 10 | w << 20}: mgs_rasterize_bwd_det(MGS_RASTER_BWD_SPLAT_SLOTS) then finds a pair's
 * record slot in the record it gathers anyway instead of in pair_info (one scattered 16-byte gather per queued entry
 * less; the reduce still reads pair_info, sequentially).
 * tile_ids (nullable): an inference frame does not need it; NULL saves the store.
 * radii_y[N] (nullable): per-axis radii as MGS_RADIUS_OPACITY_AWARE produces them (gsplat >= 1.5's radii[N,2] as two
 * arrays): the tile rectangle is mean +- (radii, radii_y) instead of the square mean +- radii.
 * ----------------------------------------------------------------------------------- */
int mgs_isect_tiles(int n, const float *means2d, const int32_t *radii, const int32_t *radii_y, const float *depths,
                    const float *conics, const float *opacities, int tile_size, int tile_w,
                    int tile_h, int cam_id, int n_cams,
                    uint32_t isect_capacity, int32_t *tiles_per_gauss, uint32_t *n_isect,
                    uint32_t *tile_ids, int32_t *flatten_ids, int64_t *isect_ids,
                    int32_t *tile_offsets, int32_t *pair_info, int32_t *tile_group_order,
                    uint32_t *status, const uint32_t *seed_info, uint32_t *seed_sums, float *splat_slots,
                    void *workspace, size_t *workspace_bytes, mgs_stream_t stream);

/* -------------------------------------------------------------------------------------
 * A batch of inference frames in ONE call (gsplat `rasterization(...)` for C cameras, no gradients):
 * per camera mgs_project_color_fwd -> mgs_isect_tiles -> mgs_rasterize_fwd in their inference-frame form
 * (packed records + binning seed, tightened tile rectangles unless MGS_FRAMES_CLASSIC_BOUNDS, no per-Gaussian
 * outputs), enqueued back to back on `stream`.  viewmats[C,4,4], Ks[C,3,3]; channels 3 (RGB) or 4 (RGB + camera-space depth as the last
 * channel); flags as mgs_rasterize_fwd (MGS_RASTER_EXPECTED_LAST turns that channel into "ED");
 * backgrounds[C,channels] nullable; antialiased != 0 = rasterize_mode "antialiased".
 * out: render[C,H,W,channels], alphas[C,H,W], n_isect[C], status[C] (as mgs_isect_tiles, per camera).
 * The per-camera intermediates live in `workspace` (two-phase size query; 256-byte aligned) and are reused
 * ds_rgba[C,H,W,4] u8 / ds_distance[C,H,W] / ds_distance_type / ds_Kinv_host: as mgs_rasterize_fwd, per camera (all
 * cameras share the intrinsics behind K^-1); render and alphas may then both be NULL.
 * from camera to camera, so any batch needs one camera's worth of scratch: ~76 B per Gaussian + 4 B per
 * list slot + the binning's own workspace.  No host read-back: capturable like a single frame.
 * ----------------------------------------------------------------------------------- */
int mgs_render_frames(int n, const float *means, const float *quats, const float *scales,
                      const float *opacities, int sh_degree, int coeff_stride, const float *sh_coeffs,
                      int n_cams, const float *viewmats, const float *Ks, int width, int height,
                      float eps2d, float near_plane, float far_plane, float radius_clip,
                      int antialiased, int channels, int flags, const float *backgrounds,
                      uint32_t isect_capacity, float *render, float *alphas, uint32_t *n_isect,
                      uint32_t *status, uint8_t *ds_rgba, void *ds_distance, int ds_distance_type,
                      const double *ds_Kinv_host, void *workspace, size_t *workspace_bytes, mgs_stream_t stream);

/* -------------------------------------------------------------------------------------
 * A batch of TRAINING frames (gsplat `rasterization(...)` for C cameras with gradients) behind two calls.
 * mgs_render_frames_train: per camera mgs_project_color_fwd (all outputs) -> mgs_isect_tiles (seeded; tiles_per_gauss,
 * tile ids, pair_info, launch order) -> mgs_rasterize_fwd (last_ids, checkpoints every checkpoint_interval entries;
 * 0 = none), enqueued back to back.  Everything the backward and gsplat's `meta` need stays, per camera, in the caller's
 * `state`: n_cams blocks of *bytes_per_camera bytes (256-byte aligned) whose fields sit at
 * mgs_train_state_layout's offsets[MGS_TRAIN_FIELDS], in this order:
 *   radii[N] i32 | means2d[N,2] | depths[N] | conics[N,3] | opacities x compensation [N] (antialiased only) |
 *   feats[N,channels] | splats[N,12] | tiles_per_gauss[N] i32 | pair_info[N,4] i32 | tile_ids[cap] u32 |
 *   flatten_ids[cap] i32 | tile_offsets[n_tiles+1] i32 | tile_group_order[ceil(n_tiles/4)] i32 | last_ids[H,W] i32 |
 *   checkpoints | {n_isect, status} u32 | radii_y[N] i32 (written under MGS_FRAMES_RADIUS_OPACITY_AWARE only)
 * flags: MGS_RASTER_EXPECTED_LAST, MGS_RASTER_LATENCY, MGS_FRAMES_CLASSIC_BOUNDS, MGS_FRAMES_RADIUS_OPACITY_AWARE.  Workspace (shared by the cameras):
 * two-phase size query, 256-byte aligned.
 * mgs_render_frames_backward: per camera mgs_rasterize_bwd_det (segmented when checkpoint_interval != 0) ->
 * mgs_project_color_bwd; v_means / v_quats / v_scales / v_sh_coeffs / v_opacities are OVERWRITTEN by the first camera
 * and added to by the others (no zero fill); v_viewmats[C,4,4] (nullable) is accumulated with atomics: zero it first.
 * v_means2d / v_means2d_abs [C,N,2] (nullable): the screen-space gradients per camera (densification); absgrad is
 * computed iff v_means2d_abs is given.  v_alphas[C,H,W] nullable.  Same arithmetic, in the same order, as the
 * per-camera entry points: bit-identical gradients.
 * ----------------------------------------------------------------------------------- */
#define MGS_TRAIN_FIELDS 17
int mgs_train_state_layout(int n, int width, int height, int channels, uint32_t isect_capacity, int antialiased,
                           int checkpoint_interval, size_t *offsets, size_t *bytes_per_camera);
int mgs_render_frames_train(int n, const float *means, const float *quats, const float *scales,
                            const float *opacities, int sh_degree, int coeff_stride, const float *sh_coeffs,
                            int n_cams, const float *viewmats, const float *Ks, int width, int height,
                            float eps2d, float near_plane, float far_plane, float radius_clip, int antialiased,
                            int channels, int flags, const float *backgrounds, uint32_t isect_capacity,
                            int checkpoint_interval, float *render, float *alphas, void *state, void *workspace,
                            size_t *workspace_bytes, mgs_stream_t stream);
int mgs_render_frames_backward(int n, const float *means, const float *quats, const float *scales,
                               const float *opacities, int sh_degree, int coeff_stride, const float *sh_coeffs,
                               int n_cams, const float *viewmats, const float *Ks, int width, int height,
                               float eps2d, int antialiased, int channels, int flags, const float *backgrounds,
                               uint32_t isect_capacity, int checkpoint_interval, const float *render,
                               const float *alphas, const float *v_render, const float *v_alphas,
                               const void *state, float *v_means, float *v_quats, float *v_scales,
                               float *v_sh_coeffs, float *v_opacities, float *v_viewmats, float *v_means2d,
                               float *v_means2d_abs, void *workspace, size_t *workspace_bytes,
                               mgs_stream_t stream);

/* gsplat `isect_offset_encode`: first sorted index per (cam, tile) from sorted int64 keys.
 * n_isect is a HOST value here (the operator takes a materialised key tensor).
 * offsets[n_cams*tile_h*tile_w]. */
int mgs_isect_offset_encode(uint32_t n_isect, const int64_t *isect_ids, int n_cams,
                            int tile_w, int tile_h, int32_t *offsets, mgs_stream_t stream);

/* -------------------------------------------------------------------------------------
 * Tile raster  (gsplat `rasterize_to_pixels` forward / backward, one camera, tile 16)
 *   means2d[N,2] conics[N,3] feats[N,channels] opacities[N]; background[channels]
 *   nullable; tile_offsets[tile_h*tile_w + 1]; flatten_ids[n_isect].
 *   out: render[H,W,channels], alphas[H,W], last_ids[H,W] i32 (sorted-list index of the
 *   last Gaussian blended into the pixel; needed by the backward; pass NULL for an
 *   inference render, which saves one select per pixel-Gaussian pair).
 *   channels in 1..MGS_MAX_CHANNELS.
 *   splats[N,12] (nullable, channels <= 4): packed records from mgs_project_color_fwd; when
 *   given, means2d / conics / feats / opacities are not read (and may be NULL).
 *   flags: MGS_RASTER_EXPECTED_LAST divides the last channel (after the background term) by
 *   max(alpha, 1e-10) in the epilogue -- the "ED" / "RGB+ED" render modes
 *   without a second pass over the frame.  MGS_RASTER_LATENCY: one wave per 8x8 block instead of
 *   one per tile (same pixels, shorter launch when the GPU is not shared with other frames).
 *   tile_group_order (nullable): from mgs_isect_tiles; tiles are then started longest lists first.
 *   checkpoints (nullable; needs last_ids, i.e. the training variant) with checkpoint_interval S (a power of two
 *   >= 64): every S list entries the kernel stores each pixel's state in front of the next entry -- T and the
 *   accumulated channels, (1 + channels) floats per pixel -- for mgs_rasterize_bwd_det, which then walks a tile's
 *   list as independent segments of S entries instead of as one serial job per tile.
 *   checkpoints[mgs_raster_checkpoint_floats(...)] is written only where some pixel is still open.
 *   ds_rgba[H,W,4] u8 / ds_distance[H,W] (both nullable; ds_distance needs ds_rgba): the DATASET FRAME straight out of
 *   the raster -- exactly what mgs_frame_to_dataset (no background of its own) makes of the float frame, byte for byte:
 *   RGBA8 with A = alpha > 0 ? max(1, round(255 alpha)) : 0 and the ray distance expected depth x ||K^-1 (x, y, 1)||
 *   (ds_distance_type 0: f32, 1: f64, 2: f16; ds_Kinv_host: HOST pointer to the 9 doubles of K^-1, read during the
 *   call).  Needs 4 channels, MGS_RASTER_EXPECTED_LAST and no last_ids (an "RGB+ED" inference frame).  render and
 *   alphas may then BOTH be NULL: 6 - 12 bytes per pixel leave the kernel instead of 20 and no conversion pass reads
 *   the frame back (a data-generation loop that only ships dataset frames).
 * ----------------------------------------------------------------------------------- */
int mgs_rasterize_fwd(int n, const float *means2d, const float *conics, const float *feats,
                      const float *opacities, const float *splats, const float *background,
                      int channels, int width, int height, int tile_w, int tile_h,
                      const int32_t *tile_offsets, const int32_t *flatten_ids,
                      const int32_t *tile_group_order, int flags,
                      float *render, float *alphas, int32_t *last_ids, float *checkpoints,
                      int checkpoint_interval, uint8_t *ds_rgba, void *ds_distance, int ds_distance_type,
                      const double *ds_Kinv_host, mgs_stream_t stream);
/* Floats of the `checkpoints` buffer for lists of up to `isect_capacity` entries (host arithmetic, no GPU work). */
size_t mgs_raster_checkpoint_floats(uint32_t isect_capacity, int tile_w, int tile_h, int channels,
                                    int checkpoint_interval);

/*   v_render[H,W,channels], v_alphas[H,W] incoming; v_means2d[N,2] v_conics[N,3]
 *   v_feats[N,channels] v_opacities[N] are ACCUMULATED into with float atomics
 *   (zero them first); v_means2d_abs[N,2] nullable (absgrad). */
int mgs_rasterize_bwd(int n, const float *means2d, const float *conics, const float *feats,
                      const float *opacities, const float *background, int channels,
                      int width, int height, int tile_w, int tile_h,
                      const int32_t *tile_offsets, const int32_t *flatten_ids,
                      const float *alphas, const int32_t *last_ids, const float *v_render,
                      const float *v_alphas, float *v_means2d, float *v_means2d_abs,
                      float *v_conics, float *v_feats, float *v_opacities,
                      mgs_stream_t stream);

/* Deterministic raster backward (no float atomics; bit-reproducible).  Same inputs as
 * mgs_rasterize_bwd plus pair_info[N,4] from mgs_isect_tiles and the list capacity.  Every
 * (tile, Gaussian) pair writes one record at its slot -- the six moments of d loss / d sigma about
 * the tile centre, the colour gradients (+2 with absgrad), padded to a multiple of four floats -- and a
 * second kernel turns each record into the pair's gradients and sums each Gaussian's records.
 * Outputs are OVERWRITTEN for all N rows (zeros where nothing contributed).  Workspace: two-phase
 * size query as above (capacity * (record floats * 4 + 1) bytes; 16-byte aligned).  Scattered device
 * atomics sustain only ~30 G/s on MI355X, which makes the atomic variant 3x slower at 1 M Gaussians.
 *   expected_render (nullable): the forward's render[H,W,channels] when it ran with
 *   MGS_RASTER_EXPECTED_LAST -- v_render's last channel is then the cotangent of
 *   channel / max(alpha, 1e-10) and the kernel's prologue converts it (and v_alphas) back to the
 *   cotangents of the un-normalised blend: no pass over the frame in between.
 *   tile_group_order (nullable): from mgs_isect_tiles; when NULL the call computes the order itself.
 *   v_alphas (nullable here): NULL = the loss does not depend on the alpha output (no zero frame needed).
 *   checkpoints + checkpoint_interval + render_out (nullable together; <= 4 channels): what mgs_rasterize_fwd wrote
 *   for these lists, and its render[H,W,channels] (== expected_render in "ED" mode).  The launch then has one unit of
 *   work per SEGMENT of checkpoint_interval list entries: a segment behind which the list goes on starts from the
 *   forward's checkpoint (its T; colour behind = final - checkpoint).  Same records and slots; against the whole-list
 *   walk the gradients differ by rounding only -- measured against fp64 the segmented walk is the closer one, it
 *   restarts from the forward's exact T -- and stay bit-reproducible run to run.
 *   flags: MGS_RASTER_BWD_RECORDS_ONLY, MGS_RASTER_BWD_SPLAT_SLOTS (needs `splats`, annotated by mgs_isect_tiles).  4-channel frames (v_render, expected_render, render_out) must be 16-byte
 *   aligned (rows are read as one 16-byte piece).
 */
int mgs_rasterize_bwd_det(int n, const float *means2d, const float *conics, const float *feats,
                          const float *opacities, const float *splats, const float *background,
                          int channels, int width, int height, int tile_w, int tile_h,
                          const int32_t *tile_offsets, const int32_t *flatten_ids,
                          const float *alphas, const int32_t *last_ids, const float *v_render,
                          const float *v_alphas, const float *expected_render,
                          const int32_t *pair_info, const int32_t *tile_group_order,
                          uint32_t isect_capacity, const float *render_out, const float *checkpoints,
                          int checkpoint_interval, int flags, float *v_means2d, float *v_means2d_abs,
                          float *v_conics, float *v_feats, float *v_opacities, void *workspace,
                          size_t *workspace_bytes, mgs_stream_t stream);

/* Fused-colour backward: chain rule of mgs_project_color_fwd.
 *   v_feats[N,feat_stride] (channel 3, if present, is d/d depth), v_means2d, v_conics,
 *   v_depths (nullable, extra d/d depth), v_opac_out (cotangent of opacities*compensation;
 *   needed iff antialiased) ->
 *   v_means[N,3], v_quats[N,4], v_scales[N,3], v_sh_coeffs[N,K,3], and, iff antialiased,
 *   v_opacities[N].  accumulate == 0: every output row is overwritten (zeros for culled
 *   Gaussians and for coefficients above the active degree); accumulate != 0: added to,
 *   so the cameras of a batch can be summed without a separate zero-fill pass.
 *   v_viewmat[4,4] (nullable): gradient of the world-to-camera matrix (projection and the SH
 *   view direction, dir = mean + R^T t), ALWAYS accumulated with one float atomic per entry
 *   per wave: zero it first.  Camera-pose optimisation only. */
int mgs_project_color_bwd(int n, const float *means, const float *quats, const float *scales,
                          const float *opacities, int sh_degree, int coeff_stride,
                          const float *sh_coeffs, const float *viewmat, const float *K,
                          int width, int height, float eps2d, const int32_t *radii,
                          const float *conics, int antialiased, int feat_stride,
                          const float *feats, const float *v_feats, const float *v_means2d,
                          const float *v_conics, const float *v_depths,
                          const float *v_opac_out, float *v_means, float *v_quats,
                          float *v_scales, float *v_sh_coeffs, float *v_opacities,
                          float *v_viewmat, int accumulate, mgs_stream_t stream);

/* -------------------------------------------------------------------------------------
 * Compositing (the step downstream of the render path): depth-tested
 * alpha-over of the splat background (premultiplied bg_rgb[P,3], bg_alpha[P], z-depth
 * bg_depth[P]) with an opaque foreground layer (fg_rgb[P,3], z-depth fg_depth[P], optional
 * uint8 fg_mask[P]; without a mask a pixel has foreground iff 0 < fg_depth < inf) and an
 * optional backdrop colour[3].  Rule per pixel:
 *   foreground in front (no splats, or fg_depth <= bg_depth): out = fg,  depth = fg_depth
 *   foreground behind the splats:        out = bg + (1 - alpha) fg,       depth = bg_depth
 *   no foreground:                       out = bg + (1 - alpha) backdrop, depth = bg_depth | +inf
 * The reference names this step (README.md:53-56) but has not released it.
 * ----------------------------------------------------------------------------------- */
int mgs_composite_over(int n_px, const float *bg_rgb, const float *bg_alpha,
                       const float *bg_depth, const float *fg_rgb, const float *fg_depth,
                       const uint8_t *fg_mask, const float *backdrop, float *out_rgb,
                       float *out_depth, mgs_stream_t stream);

/* 8-bit frame for the dataset writer / the multi-GPU gather: out[P,3] = round(255 *
 * clamp(rgb + (1 - alpha) * background, 0, 1)) -- splatfacto's post-processing
 * and the quantisation of an image file.  rgb[P,rgb_stride] (first three channels are used, so
 * an RGB+ED render can be passed as is), alpha[P], background[3] nullable (black). */
int mgs_frame_to_u8(int n_px, const float *rgb, int rgb_stride, const float *alpha,
                    const float *background, uint8_t *out, mgs_stream_t stream);

/* Dataset frame in the layout the reference's readers open (the OUTPUT side of the path):
 *   rgba[H,W,4] u8 (nullable): RGB as mgs_frame_to_u8, A = alpha > 0 ? max(1, round(255 alpha)) : 0 -- the
 *     RGBA image load_images (/root/reference/Articulation/utils/nerf2physic_utils.py:84-101) opens; its
 *     mask `A > 0` is exactly `alpha > 0`;
 *   distance[H,W] f32, or f64 when distance_f64 != 0 (nullable): the RAY DISTANCE z * ||K^-1 (x, y, 1)||
 *     through INTEGER pixel coordinates, z = the last channel of colors (the "ED" depth of an RGB+ED frame;
 *     color_stride >= 4) -- what load_depths (:104-118) opens and distance_to_depth (:135-146) turns back
 *     into z.  Computed in fp64 and rounded once; with the f64 output the reader recovers z to the last
 *     fp32 bit, with f32 (what `ns-render` stores) to one ulp.  distance_f64 == 2: IEEE half (np.float16, which
 *     np.load and the reader's arithmetic take as they are; 11 significant bits: the light payload of the
 *     multi-GPU gather, 6 B per pixel with the RGBA image).
 *   Kinv_host: HOST pointer to the 9 doubles of K^-1 (row-major), read during the call. */
int mgs_frame_to_dataset(int width, int height, const float *colors, int color_stride,
                         const float *alpha, const float *background, const double *Kinv_host,
                         uint8_t *rgba, void *distance, int distance_f64, mgs_stream_t stream);

/* -------------------------------------------------------------------------------------
 * Point-cloud z-buffer helpers: the reference's
 * Articulation/utils/point_utils.py, one entry point per function.
 *
 * mgs_points_project  = project_pcd (point_utils.py:13-26): pts[N,3] world, K[3,3],
 *   c2w[4,4] row-major (camera looks down +z, no axis flip); pnt_cam = R^T (p - t),
 *   uv[N,3] = (pnt_cam / z) K^T (third component 1), pnt_cam[N,3]; depth = pnt_cam[:,2].
 * mgs_points_depth_map = get_depth_map (point_utils.py:44-73): cells_h = int(h / scale),
 *   cells_w = int(w / scale) (computed by the caller as the reference does); every point goes
 *   to cell (clip(round_half_even(u / scale)), clip(round_half_even(v / scale))), each cell
 *   keeps min(bg_depth, depths) and the index of the first point attaining it (torch_scatter
 *   scatter_min; N where no point is strictly below bg_depth).  depth_map[h,w] is the
 *   nearest-neighbour upsample (cv2.INTER_NEAREST), index[cells_w*cells_h] (nullable) is in the
 *   reference's cell order u * cells_h + v.  uv_stride = floats per uv row (2 or 3).
 *   Workspace: two-phase size query (8 bytes per cell).
 * mgs_points_sample_mask = mask_pcd_2d (point_utils.py:76-111): out[i] = bilinear(mask, uv_i)
 *   > thresh, and, when depth_map and pnt_depth are both given, |bilinear(depth_map, uv_i) -
 *   pnt_depth[i]| < depth_thresh.  Bilinear = grid_sample(align_corners=True, border padding)
 *   at (uv - [w/2, h/2]) / [w/2, h/2].  mask, depth_map: float [h,w].
 * ----------------------------------------------------------------------------------- */
int mgs_points_project(int n, const float *pts, const float *K, const float *c2w, float *uv,
                       float *pnt_cam, mgs_stream_t stream);
int mgs_points_depth_map(int n, const float *uv, int uv_stride, const float *depth, int height,
                         int width, int cells_h, int cells_w, float scale, float bg_depth,
                         float *depth_map, int64_t *index, void *workspace,
                         size_t *workspace_bytes, mgs_stream_t stream);
int mgs_points_sample_mask(int n, const float *uv, int uv_stride, const float *mask, int height,
                           int width, float thresh, const float *depth_map,
                           const float *pnt_depth, float depth_thresh, uint8_t *out,
                           mgs_stream_t stream);

/* -------------------------------------------------------------------------------------
 * Photometric L1 term of the training step (BASELINE configs[2]): loss = mean |a - b| over n
 * floats (a = rendered image, b = target), and its gradient v_a = v_loss * sign(a - b) / n
 * (sign(0) = 0, as torch).  loss, v_loss: device scalars (v_loss NULL = 1).  Fixed summation
 * order: bit-reproducible.  Workspace (forward): two-phase size query.  Buffers 16-byte aligned.
 * ----------------------------------------------------------------------------------- */
int mgs_l1_loss_fwd(size_t n, const float *a, const float *b, float *loss, void *workspace,
                    size_t *workspace_bytes, mgs_stream_t stream);
int mgs_l1_loss_bwd(size_t n, const float *a, const float *b, const float *v_loss, float *v_a,
                    mgs_stream_t stream);
/* The training step's form: loss AND v_a = sign(a - b) / n (the gradient for v_loss = 1) in ONE pass over a and b
 * -- the same bits as the two calls above.  mgs_l1_loss_bwd_scale then turns v_a into the gradient for the v_loss
 * that arrived, in place (v_a <- sign(v_a) * v_loss / n: idempotent), and is a no-op launch when *v_loss == 1. */
int mgs_l1_loss_fwd_grad(size_t n, const float *a, const float *b, float *loss, float *v_a,
                         void *workspace, size_t *workspace_bytes, mgs_stream_t stream);
int mgs_l1_loss_bwd_scale(size_t n, const float *v_loss, float *v_a, mgs_stream_t stream);

/* -------------------------------------------------------------------------------------
 * Similarity transforms of Gaussian groups (world-frame alignment,
 * /root/reference/README.md:54-55; per frame: Gaussians riding on articulated parts).
 * group_ids[N] (nullable = all in group 0; -1 = leave unchanged) selects one of n_groups rows of
 *   xforms[n_groups,20] = { M[9] = s R row-major, t[3], q_R[4] (wxyz), s, pad[3] }:
 *   means' = M p + t, quats' = q_R (x) q, scales' = s scales.
 * sh_coeffs / out_sh [N,coeff_stride,3] (nullable pair): degree-l coefficients are multiplied by the
 *   (2l+1)x(2l+1) real-SH rotation matrix of R, sh_rot[n_groups,84] = {3x3, 5x5, 7x7, pad}
 *   row-major (needed for sh_degree >= 1).  out_* may alias the inputs (in place).
 * ----------------------------------------------------------------------------------- */
int mgs_transform_gaussians(int n, const float *means, const float *quats, const float *scales,
                            int sh_degree, int coeff_stride, const float *sh_coeffs,
                            const int32_t *group_ids, int n_groups, const float *xforms,
                            const float *sh_rot, float *out_means, float *out_quats,
                            float *out_scales, float *out_sh, mgs_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MGS_H_ */

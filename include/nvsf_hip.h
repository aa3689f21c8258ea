/*
 * nvsf_hip.h -- C ABI of libnvsf_hip.so, the MI355X (gfx950) implementation of the NVSF volumetric
 * rendering hot path.  This is the drop-in boundary: plain pointers to DEVICE memory, sizes, scalars and
 * the HIP stream to launch on.  No torch / ATen types.  Every entry point
 *   - is asynchronous on `stream` (pass torch.cuda.current_stream().cuda_stream; NULL = default stream),
 *   - never allocates, frees or retains device memory (the caller owns every buffer, exactly as the
 *     reference's Python wrappers allocate all outputs: nvsf/nerf/raymarching/raymarching.py:40-43,
 *     235-270, 314-320, 338-355, 424-455),
 *   - returns 0 on success, a negative NVSF_ERR_* code for rejected arguments (nothing is launched),
 *     or a positive hipError_t if the launch itself failed.  (The reference returns void and never
 *     polls launch errors: raymarching.cu:159-177.)
 *
 * "ref:" lines name the reference interface each function replaces, relative to /root/reference.
 * All float buffers are fp32 unless stated; index buffers are int32; layouts are the reference's
 * (row-major, AoS [N,3] rays).
 */
#ifndef NVSF_HIP_H
#define NVSF_HIP_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* nvsf_stream_t; /* == hipStream_t */

#define NVSF_OK 0
#define NVSF_ERR_INVALID_ARG (-1)
#define NVSF_ERR_UNSUPPORTED (-2)

/* library / build identification: returns a static string "nvsf_hip <version> gfx950" */
const char* nvsf_version(void);

/* identity of the SOURCES the mapped library was built from: which = 0: sha1[:16] over every translation unit, every csrc header and
 * the compile flags (build.csrc_digest_all()); which = 1: the same over the render kernels' translation unit + headers + flags
 * (build.csrc_digest(), what PMC profiles are tagged with).  A static 16-character string; "unknown_________" for a build made
 * without build.py.  bench.py reports THIS digest (not that of the sources on disk) and build.py relinks when it differs. */
const char* nvsf_build_digest(int which);

/* TEST-ONLY: selects a second formulation of an operator -- the one the tests pin the production form against -- for every later
 * call of this process: name in {"march", "planes_fwd", "planes_bwd", "hashgrid_fwd", "hashgrid_bwd", "hash4d_bwd", "slice_plan",
 * "render_tail", "march_skew", "mlp_bwd", "level_kinds"}, value 0 = the production form (the default), 1 (march: 1, 2; march_skew:
 * 1 .. 8; mlp_bwd: 1, 2) = the reference form.  Returns the previous value,
 * or NVSF_ERR_INVALID_ARG.  Not thread-safe; results of either form are the same operator's (bit-identical or within the
 * tolerances stated in tests/).  No stream argument: nothing is launched. */
int nvsf_test_variant(const char* name, int value);

/* ------------------------------------------------------------------------------------------------
 * Section 1: the `_raymarching` extension (ref: nvsf/nerf/raymarching/src/raymarching.h:6-96,
 * registered by src/bindings.cpp:5-21).  Argument order follows the reference launchers.
 * ---------------------------------------------------------------------------------------------- */

/* ref: near_far_from_aabb, raymarching.h:6-12, kernel raymarching.cu:104-157.
 * rays_o, rays_d [N,3]; aabb [6] = (xmin,ymin,zmin,xmax,ymax,zmax); nears, fars [N].
 * Miss => near = far = FLT_MAX; near is clamped up to min_near. */
int nvsf_near_far_from_aabb(const float* rays_o, const float* rays_d, const float* aabb, uint32_t N,
                            float min_near, float* nears, float* fars, nvsf_stream_t stream);

/* ref: sph_from_ray, raymarching.h:13-17, kernel raymarching.cu:182-217.  coords [N,2] in [-1,1]. */
int nvsf_sph_from_ray(const float* rays_o, const float* rays_d, float radius, uint32_t N, float* coords,
                      nvsf_stream_t stream);

/* ref: morton3D / morton3D_invert, raymarching.h:18-21, kernels raymarching.cu:237-272.
 * coords int32 [N,3] (10 bits per axis), indices int32 [N]. */
int nvsf_morton3D(const int32_t* coords, uint32_t N, int32_t* indices, nvsf_stream_t stream);
int nvsf_morton3D_invert(const int32_t* indices, uint32_t N, int32_t* coords, nvsf_stream_t stream);

/* ref: packbits, raymarching.h:22-25, kernel raymarching.cu:286-306.
 * grid fp32 [8*N] (16-byte aligned); bitfield uint8 [N]; bit i of byte n = grid[8n+i] > density_thresh. */
int nvsf_packbits(const float* grid, uint32_t N, float density_thresh, uint8_t* bitfield, nvsf_stream_t stream);

/* ref: march_rays_train, raymarching.h:27-44, kernel raymarching.cu:331-534.
 * grid: uint8 occupancy bitfield [C*H^3/8] in Morton order per cascade.  Outputs xyzs, dirs [M,3],
 * deltas [M,2] must be zero-initialised by the caller (raymarching.py:235-237); rays int32 [N,3] =
 * (ray id, sample offset, sample count); counter int32 [2] is read-modify-written:
 * counter[0] += total samples, counter[1] += N.  Rays whose range would exceed M are recorded in
 * `rays` but write no samples (raymarching.cu:457).
 * Deviation (documented in DESIGN.md): sample ranges are assigned in ray-index order by a prefix
 * sum, not in atomicAdd arrival order, and ray n is recorded at rays[n] (the reference's slot is
 * atomicAdd(counter+1,1), which equals n only up to a permutation).
 * Runs the one-launch form below on a scratch block it takes from the device's DEFAULT stream-ordered memory pool for the duration
 * of the launch (hipMallocAsync / hipFreeAsync on `stream`; no synchronisation), so that the reference's argument list gets the fast
 * kernel; falls back to nvsf_march_rays_train_passes when the pool cannot serve the block.  Ownership: this is the ONE entry point of
 * the library that does not work on caller-owned memory alone (the reference's signature has no scratch argument).  The first call
 * on a device raises that pool's hipMemPoolAttrReleaseThreshold to 16 MiB (it is 0 by default: a freed block returns to the OS at the
 * next synchronisation; a threshold the application has already raised is left alone), so the 1 KB + 16 B per four rays are served
 * from the pool's reserve from the second call on: no allocation on the call path, at most 16 MiB held that the caller's own
 * allocator does not see (nvsf_scratch_pool_stats).  Callers that want no library-side allocation at all use
 * nvsf_march_rays_train_ws with their own scratch -- the Python wrapper does.  As for the one-launch form,
 * counter[1] < 0 after the call marks a launch whose bounded inter-workgroup wait expired (never seen outside the test that forces
 * it): the outputs are invalid and the call is to be repeated through nvsf_march_rays_train_passes on the counter as it was. */
int nvsf_march_rays_train(const float* rays_o, const float* rays_d, const uint8_t* grid, float bound,
                          float dt_gamma, uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H, uint32_t M,
                          const float* nears, const float* fars, float* xyzs, float* dirs, float* deltas,
                          int32_t* rays, int32_t* counter, const float* noises, nvsf_stream_t stream);
/* The same operator as count / scan / write launches: no wait between workgroups, no scratch, the same outputs bit for bit.  What
 * the one-launch form is repeated through when its wait expired, and the form the tests pin it against. */
int nvsf_march_rays_train_passes(const float* rays_o, const float* rays_d, const uint8_t* grid, float bound,
                                 float dt_gamma, uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H, uint32_t M,
                                 const float* nears, const float* fars, float* xyzs, float* dirs, float* deltas,
                                 int32_t* rays, int32_t* counter, const float* noises, nvsf_stream_t stream);

/* The same operator in ONE launch (no second classification of the chain, no scan launch): every wave counts its ray once and
 * keeps the sample masks of its batches on chip; worker workgroups publish the sum of each ticket of four rays, one scanner wave
 * turns the sums into exclusive prefixes in ticket order, and each ticket is stored while the next one is counted.  Same
 * arguments, same outputs bit for bit (ray-index order), plus a caller-owned scratch `workspace` of at least
 * nvsf_march_rays_train_ws_bytes(N) bytes (1 KB + 16 bytes per four rays), 8-byte aligned, contents irrelevant on entry (it is cleared on
 * the stream first) and meaningless afterwards.  The reference's signature (raymarching.h:27-44) has no scratch argument, hence
 * the separate entry point for callers that own a scratch block; nvsf_march_rays_train takes one from the stream-ordered pool.
 * counter[1] < 0 after the call (its sign bit set) marks a launch whose bounded inter-workgroup wait expired (outputs invalid; the
 * mark is sticky: later calls on the same counter leave it negative).  `spin_limit` = polls a waiting wave makes before it gives up,
 * 0 = the library's default (2^22); tests pass 1 to force the expiry path. */
size_t nvsf_march_rays_train_ws_bytes(uint32_t N);

/* Diagnostics of the pool nvsf_march_rays_train borrows from (the current device's default stream-ordered pool): bytes the pool
 * holds reserved, bytes handed out, and its release threshold (any pointer may be NULL).  No stream argument: nothing is launched. */
int nvsf_scratch_pool_stats(uint64_t* reserved_bytes, uint64_t* used_bytes, uint64_t* release_threshold);
int nvsf_march_rays_train_ws(const float* rays_o, const float* rays_d, const uint8_t* grid, float bound,
                             float dt_gamma, uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H, uint32_t M,
                             const float* nears, const float* fars, float* xyzs, float* dirs, float* deltas,
                             int32_t* rays, int32_t* counter, const float* noises, void* workspace,
                             size_t workspace_bytes, uint32_t spin_limit, nvsf_stream_t stream);

/* ref: composite_rays_train_forward, raymarching.h:45-54, kernel raymarching.cu:577-655.
 * sigmas [M], rgbs [M,3], deltas [M,2], rays [N,3] -> weights_sum, depth [N], image [N,3]
 * (indexed by rays[n,0]).  Early exit once transmittance < T_thresh (sample included). */
int nvsf_composite_rays_train_forward(const float* sigmas, const float* rgbs, const float* deltas,
                                      const int32_t* rays, uint32_t M, uint32_t N, float T_thresh,
                                      float* weights_sum, float* depth, float* image, nvsf_stream_t stream);

/* ref: composite_rays_train_backward, raymarching.h:55-67, kernel raymarching.cu:690-772.
 * grad_sigmas [M], grad_rgbs [M,3] must be zero-initialised by the caller (raymarching.py:338-339);
 * no gradient flows from depth (raymarching.py:330). */
int nvsf_composite_rays_train_backward(const float* grad_weights_sum, const float* grad_image,
                                       const float* sigmas, const float* rgbs, const float* deltas,
                                       const int32_t* rays, const float* weights_sum, const float* image,
                                       uint32_t M, uint32_t N, float T_thresh, float* grad_sigmas,
                                       float* grad_rgbs, nvsf_stream_t stream);

/* ref: march_rays, raymarching.h:69-86, kernel raymarching.cu:808-928.
 * Outputs [n_alive*n_step (+pad), 3/3/2] zero-initialised by the caller; unfilled slots stay 0. */
int nvsf_march_rays(uint32_t n_alive, uint32_t n_step, const int32_t* rays_alive, const float* rays_t,
                    const float* rays_o, const float* rays_d, float bound, float dt_gamma, uint32_t max_steps,
                    uint32_t C, uint32_t H, const uint8_t* grid, const float* nears, const float* fars,
                    float* xyzs, float* dirs, float* deltas, const float* noises, nvsf_stream_t stream);

/* ref: composite_rays, raymarching.h:87-96, kernel raymarching.cu:966-1053.  In-place update of
 * weights_sum, depth [N], image [N,3], rays_t [N]; rays_alive[n] = -1 when the ray terminated. */
int nvsf_composite_rays(uint32_t n_alive, uint32_t n_step, float T_thresh, int32_t* rays_alive, float* rays_t,
                        const float* sigmas, const float* rgbs, const float* deltas, float* weights_sum,
                        float* depth, float* image, nvsf_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Section 2: uniform sampler + alpha compositor -- the arithmetic of NeRFRenderer.run that the
 * reference issues as ~10 elementwise / cumprod torch kernels (ref: nvsf/nerf/models/renderer_dynamic.py).
 * ---------------------------------------------------------------------------------------------- */

/* ref: renderer_dynamic.py:155-169.  z = near + (far-near)*lin[i] [+ (noise-0.5)*(far-near)/T];
 * xyz = clip(o + d*z, aabb).  lin [T] = torch.linspace(0,1,T); noise [N,T] in [0,1) or NULL;
 * aabb [6] device; z_vals [N,T]; xyzs [N,T,3] or NULL. */
int nvsf_uniform_samples(const float* rays_o, const float* rays_d, const float* nears, const float* fars,
                         const float* lin, const float* noise, const float* aabb, uint32_t N, uint32_t T,
                         float* z_vals, float* xyzs, nvsf_stream_t stream);

/* ref: renderer_dynamic.py:181-194 (deltas, alphas, cumprod weights) and :216-221 (weights_sum, depth).
 * k_scale = density_scale (x2 for an active sensor, :187-189).  sigmas, z_vals, weights [N,T]. */
int nvsf_composite_uniform_weights_fwd(const float* sigmas, const float* z_vals, const float* nears,
                                       const float* fars, uint32_t N, uint32_t T, float k_scale, float* weights,
                                       float* weights_sum, float* depth, nvsf_stream_t stream);

/* autograd of the above (the reference relies on torch autograd through :181-221).
 * grad_weights [N,T] / grad_weights_sum [N] / grad_depth [N] may each be NULL. */
int nvsf_composite_uniform_weights_bwd(const float* sigmas, const float* z_vals, const float* nears,
                                       const float* fars, const float* grad_weights, const float* grad_weights_sum,
                                       const float* grad_depth, uint32_t N, uint32_t T, float k_scale,
                                       float* grad_sigmas, nvsf_stream_t stream);

/* ref: renderer_dynamic.py:224 and :236-237.  image[n,c] = sum_i w[n,i]*rgb[n,i,c] (+ (1-ws[n])*bg[c]).
 * rgbs [N,T,C], C in 1..4; bg_color [C] device or NULL. */
int nvsf_composite_uniform_image_fwd(const float* weights, const float* rgbs, const float* weights_sum, uint32_t N,
                                     uint32_t T, uint32_t C, const float* bg_color, float* image,
                                     nvsf_stream_t stream);
int nvsf_composite_uniform_image_bwd(const float* weights, const float* rgbs, const float* grad_image, uint32_t N,
                                     uint32_t T, uint32_t C, const float* bg_color, float* grad_weights,
                                     float* grad_rgbs, float* grad_weights_sum, nvsf_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Section 3: per-sample field operators the reference obtains from tiny-cuda-nn (third party, unpinned:
 * setup.py:97-99).  Semantics: DESIGN.md section 4.  fp16 buffers are passed as void*.
 * Level metadata arrays (h_*) are HOST pointers, copied into the launch.
 * ---------------------------------------------------------------------------------------------- */

/* ref: tcnn.Encoding("HashGrid") at hash_field.py:47-57 (D=2), hash_field.py:109-119 (D=3),
 * flow_field.py:70-80 (D=3, F=8).  x fp32 [M, x_stride], cols[D] = columns of x to encode (host),
 * table fp16 [h_offsets[L]*F], out fp16 [M, out_stride >= L*F].  D in {2,3}, F in {2,4,8}, L <= 32. */
int nvsf_hashgrid_fwd(const float* x, uint32_t M, uint32_t x_stride, const uint32_t* cols, uint32_t D,
                      const void* table_f16, uint32_t L, uint32_t F, const float* h_scales, const uint32_t* h_res,
                      const uint32_t* h_offsets, void* out_f16, uint32_t out_stride, nvsf_stream_t stream);

/* ref: the static hash encoder of the space-time field, hash_field.py:109-119 (tcnn HashGrid, D = 3, L = 8, F = 4, every level
 * hashed), evaluated for the density tail of network_dynamic.py:273-287.  Same features as nvsf_hashgrid_fwd, stored LEVEL-MAJOR:
 * out fp16 [L][M][F] (what nvsf_density_dynamic_lm_fwd reads).  x fp32 [M, x_stride >= 3], columns 0..2.  Other grid shapes:
 * NVSF_ERR_UNSUPPORTED (use nvsf_hashgrid_fwd). */
int nvsf_hashgrid_fwd_level_major(const float* x, uint32_t M, uint32_t x_stride, const void* table_f16, uint32_t L, uint32_t F,
                                  const float* h_scales, const uint32_t* h_res, const uint32_t* h_offsets, void* out_f16,
                                  nvsf_stream_t stream);

/* gradient wrt the table: grad_table_f32[row*F+f] += w_corner * grad_out[m, l*F+f] (fp32 atomics; the
 * caller zero-initialises).  grad_out [M, go_stride] is fp16 (grad_is_f16 != 0) or fp32. */
int nvsf_hashgrid_bwd(const float* x, uint32_t M, uint32_t x_stride, const uint32_t* cols, uint32_t D, uint32_t L,
                      uint32_t F, const float* h_scales, const uint32_t* h_res, const uint32_t* h_offsets,
                      const void* grad_out, int grad_is_f16, uint32_t go_stride, float* grad_table_f32,
                      nvsf_stream_t stream);

/* The same gradient with the levels merge_from ... L-1 scattered through bins instead of memory-side atomics.  Two streaming passes:
 * contributions {row, F values} are appended to the bin (8192 / F consecutive rows of one level) their row falls into, then the
 * workgroups of a bin sum its contributions in LDS (64-bit fixed point, order-independent) and add the image to grad_table_f32.
 *   levels fine_from ... L-1  (cells shorter than a ray step: no two consecutive rows share a cell): one contribution per (row,
 *                             vertex); these levels must be hashed;
 *   levels merge_from ... fine_from-1: runs of up to 8 CONSECUTIVE ROWS that fall into one cell are summed in registers first and
 *                             contribute once per vertex (rows are expected in ray order; any order is correct);
 *   levels 0 ... merge_from-1: the kernel of nvsf_hashgrid_bwd.
 * Requirements: D = 3, F in {2, 4}, binned levels of at most 2^20 rows and 256 bins, hashed ones of power-of-two size, M * 8 <= 2^26.
 * workspace: device memory, 256-byte aligned, at least nvsf_hashgrid_bwd_binned_ws_bytes(...) bytes (0 = shape not supported),
 * contents irrelevant.  Same sums as nvsf_hashgrid_bwd up to the order of the fp32 additions (2^-33 of a level's largest gradient per
 * contribution), for any input: contributions beyond a bin's capacity are added directly.
 * grad_out: rows [M, go_stride >= L F] (go_level_stride = 0) or level-major [L][M][F] (go_level_stride = elements between two levels,
 * go_stride = F: what nvsf_mlp_bwd writes in its column-block layout) -- every pass over a level then reads one contiguous column
 * instead of 8 ... 16 bytes out of every 128-byte row. */
size_t nvsf_hashgrid_bwd_binned_ws_bytes(uint32_t M, uint32_t L, uint32_t F, const uint32_t* h_res, const uint32_t* h_offsets,
                                         uint32_t merge_from, uint32_t fine_from);
int nvsf_hashgrid_bwd_binned(const float* x, uint32_t M, uint32_t x_stride, const uint32_t* cols, uint32_t D, uint32_t L,
                             uint32_t F, const float* h_scales, const uint32_t* h_res, const uint32_t* h_offsets,
                             const void* grad_out, int grad_is_f16, uint32_t go_stride, uint32_t go_level_stride,
                             float* grad_table_f32, uint32_t merge_from, uint32_t fine_from, void* workspace,
                             size_t workspace_bytes, nvsf_stream_t stream);

/* ref: tcnn.Encoding("Frequency") network_dynamic.py:108-114.  x fp32 [M,n_dims] ->
 * out fp16 [M, out_stride >= 2*n_dims*n_freq], out[i*2K+2k] = sin(2^k pi x_i), [..+1] = cos. */
int nvsf_freq_encode(const float* x, uint32_t M, uint32_t n_dims, uint32_t n_freq, void* out_f16,
                     uint32_t out_stride, nvsf_stream_t stream);

/* ref: tcnn.Encoding("SphericalHarmonics", degree 4) network_dynamic.py:165-170.
 * dirs01 fp32 [M,3] in [0,1] -> out fp16 [M, out_stride >= 16]. */
int nvsf_sh4_encode(const float* dirs01, uint32_t M, void* out_f16, uint32_t out_stride, nvsf_stream_t stream);

/* ref: backward of `sigma = trunc_exp(h[..., 0]); geo_feat = h[..., 1:]` (network_dynamic.py:284-287, activation.py:6-20) as
 * one pass: grad_h[m][0] = grad_sigma[m] * clamp(sigma[m], sigma_lo, sigma_hi), grad_h[m][1 + j] = grad_geo[m][j] (j < n_geo <= 15),
 * remaining columns of the 16 zero.  grad_sigma / grad_geo may be NULL (read as zeros).  grad_h fp32 [M, gh_stride >= 16],
 * 16-byte aligned rows. */
int nvsf_sigma_geo_bwd(const float* grad_sigma, const float* sigma, const float* grad_geo, uint32_t gg_stride, uint32_t n_geo,
                       uint32_t M, float* grad_h, uint32_t gh_stride, float sigma_lo, float sigma_hi, nvsf_stream_t stream);

/* ref: the `.to(half)` / `torch.cat([d, geo_feat], dim=-1)` glue of network_dynamic.py:310-325 for one column block:
 * dst[m][j] = fp16(src[m][j]), j < n_cols, between row-strided views (src fp32, or fp16 when src_is_f16). */
int nvsf_cast_cols_f16(const void* src, int src_is_f16, uint32_t M, uint32_t n_cols, uint32_t src_stride, void* dst_f16,
                       uint32_t dst_stride, nvsf_stream_t stream);

/* ref: the per-sample direction encoding of network_dynamic.py:310-312,319-321 for samples that share their ray's direction
 * (renderer_dynamic.py:198-200 expands rays_d over the T samples): dst[n * T + t][j] = src[n][j], j < n_cols; fp16 rows. */
int nvsf_repeat_rows_f16(const void* src_f16, uint32_t N, uint32_t n_cols, uint32_t src_stride, uint32_t T, void* dst_f16,
                         uint32_t dst_stride, nvsf_stream_t stream);

/* ref: `torch.cat([view_encoder(d), geo_feat], dim=-1)` of network_dynamic.py:310-325 for samples ordered ray by ray, as the
 * padded fp16 input rows of the heads: dst[n * T + t] = [enc_ray[n][0 .. n_enc) | fp16(geo[n * T + t][0 .. n_geo)) | 1.0 ...] up to
 * in_cols columns (tcnn pads the network input with ones).  n_enc, in_cols, enc_stride, dst_stride multiples of 8; 16-byte
 * aligned enc_ray / dst; geo fp32, or fp16 when geo_is_f16. */
int nvsf_heads_input_f16(const void* enc_ray_f16, uint32_t N, uint32_t n_enc, uint32_t enc_stride, uint32_t T, const void* geo,
                         int geo_is_f16, uint32_t n_geo, uint32_t geo_stride, void* dst_f16, uint32_t in_cols, uint32_t dst_stride,
                         nvsf_stream_t stream);

/* ref: `rgbs[mask] = torch.sigmoid(h)` into zeros (network_dynamic.py:325-330) when the heads were evaluated on every sample:
 * out[m][c] = mask[m] ? 1 / (1 + exp(-logits[m * row_stride + c * col_stride])) : 0; out fp32 [M, C] dense; mask one byte per
 * row (NULL = all rows). */
int nvsf_masked_sigmoid(const float* logits, uint32_t row_stride, uint32_t col_stride, const void* mask_u8, uint32_t M, uint32_t C,
                        float* out, nvsf_stream_t stream);

/* Backward of the above from its output (aten::sigmoid_backward): grad_in = (grad_out * (1 - out)) * out, n floats. */
int nvsf_sigmoid_bwd(const float* grad_out, const float* out, uint32_t n, float* grad_in, nvsf_stream_t stream);

/* Population of a sample mask: *count = number of non-zero bytes of x[0..n) (x 16-byte aligned; one launch, *count cleared on the
 * stream first).  The mask is the reference's `weights > 1e-4` (renderer_dynamic.py:253), whose population decides between the dense
 * and the gathered evaluation of the per-sample heads. */
int nvsf_count_nonzero_u8(const void* x, uint64_t n, int64_t* count, nvsf_stream_t stream);

/* ref: trunc_exp forward, nvsf/nerf/activation.py:9-11, on one column of a row-strided fp32 matrix: out[m] = exp(h[m][col])
 * (the density logit of the sigma MLP's [M,16] output, network_dynamic.py:281). */
int nvsf_exp_col(const float* h, uint32_t row_stride, uint32_t col, uint32_t M, float* out, nvsf_stream_t stream);

/* ref: tcnn.Network("FullyFusedMLP") network_dynamic.py:125-135,138-161,180-189.
 * x [M, x_stride] fp32 (x_is_f16 == 0) or fp16; weights fp16 = W0 [hidden][in_cols] ++ (n_hidden-1) x
 * [hidden][hidden] ++ W_out [out_cols][hidden]; columns n_in..in_cols-1 of the input read as 1.0.
 * Supported: hidden = 64, n_hidden in 1..3, in_cols in {16,...,128}; the output layer always has 16 rows in `weights`.
 * out_cols = 16: out fp32 [M, out_stride >= 16] (16-byte aligned rows) receives all 16 outputs; out_cols in 1..4: only the
 * leading out_cols outputs are stored, out fp32 [M, out_stride >= out_cols] (a head with one or three outputs writes 4 ... 16
 * bytes per row instead of 64; two heads may interleave their columns in one buffer).  The output layer is NOT rounded to fp16. */
int nvsf_mlp_fwd(const void* x, int x_is_f16, uint32_t M, uint32_t n_in, uint32_t x_stride, const void* weights_f16,
                 uint32_t in_cols, uint32_t hidden, uint32_t n_hidden, uint32_t out_cols, float* out_f32,
                 uint32_t out_stride, nvsf_stream_t stream);

/* ref: the backward pass tcnn.Network provides to autograd for the same FullyFusedMLP (tcnn's Module.backward behind
 * network_dynamic.py:125-161,180-189): one kernel, activations recomputed from x.
 * grad_out fp32 [M, go_stride], columns 0..n_out-1 used (the padded outputs carry no gradient).
 * grad_x fp32 [M, gx_stride] or NULL: column j receives dL/dx of input column gx_col0 + j, j < n_in - gx_col0 (gx_col0 = 0:
 * the whole input gradient).  Input tiles left of gx_col0 are not computed -- a head whose leading columns are a
 * parameter-free direction encoding only needs the gradient of its trailing geometry features.  gx_accumulate != 0 adds
 * to grad_x instead of overwriting it (two heads sharing one input).  Bits 8..15 of gx_accumulate = B in {2, 4} select a
 * column-block layout of grad_x instead of rows: grad_x [n_in / B][M][B] (gx_col0 = 0, gx_stride ignored, n_in % 4 == 0,
 * 16-byte aligned rows of x) -- the gradient of a hash grid's L x F features level by level, what nvsf_hashgrid_bwd_binned
 * reads with go_level_stride = M * F.
 * grad_weights_f32: fp32 buffer in the layout of weights_f16; dL/dW is ADDED to it (zero it first).
 * Gradients travel in fp16 multiplied by grad_scale (tcnn's loss_scale, e.g. 128) and are unscaled on the way out.
 * Supported: hidden = 64, out_cols = 16, n_hidden in 1..2, in_cols <= 128. */
int nvsf_mlp_bwd(const void* x, int x_is_f16, uint32_t M, uint32_t n_in, uint32_t x_stride, const void* weights_f16,
                 uint32_t in_cols, uint32_t hidden, uint32_t n_hidden, uint32_t out_cols, const float* grad_out,
                 uint32_t n_out, uint32_t go_stride, float grad_scale, float* grad_x, uint32_t gx_stride,
                 float* grad_weights_f32, uint32_t gx_col0, int gx_accumulate, nvsf_stream_t stream);

/* nvsf_mlp_bwd for a DENSITY network (network_dynamic.py:281-287: sigma = trunc_exp(h[:, 0]), geo_feat = h[:, 1:]) whose logit gradient is
 * formed from its parts while the operands are fetched, instead of read from an [M, 16] matrix that nvsf_sigma_geo_bwd wrote:
 *   grad_out[m][0] = grad_sigma[m] * clamp(sigma[m], sigma_lo, sigma_hi)   (trunc_exp's backward, activation.py:16-20; grad_sigma NULL: 0)
 *   grad_out[m][1 + j] = grad_geo_a[m][j] (+ grad_geo_b[m][j] when given: two heads sharing the features),  j < n_geo <= 15
 * grad_geo rows: fp32, geo_stride >= 16 floats (a multiple of 4), 16-byte aligned.  Everything else as nvsf_mlp_bwd with n_out = 1 + n_geo.
 * One launch and 128 B / sample of matrix traffic less than nvsf_sigma_geo_bwd + nvsf_mlp_bwd; the second head needs no read-modify-write. */
int nvsf_mlp_bwd_density(const void* x, int x_is_f16, uint32_t M, uint32_t n_in, uint32_t x_stride, const void* weights_f16,
                         uint32_t in_cols, uint32_t hidden, uint32_t n_hidden, uint32_t out_cols, const float* grad_sigma,
                         const float* sigma, const float* grad_geo_a, const float* grad_geo_b, uint32_t geo_stride, uint32_t n_geo,
                         float sigma_lo, float sigma_hi, float grad_scale, float* grad_x, uint32_t gx_stride, float* grad_weights_f32,
                         uint32_t gx_col0, int gx_accumulate, nvsf_stream_t stream);

/* nvsf_mlp_fwd / nvsf_mlp_bwd on rows with a shared prefix: logical input row r =
 *   [ prefix[r / rows_per_prefix][0 : prefix_cols] | x[r][0 : n_in - prefix_cols] ]
 * -- the per-sample heads, whose first 72 (LiDAR) / 16 (camera) inputs are the encoded direction of the sample's RAY
 * (network_dynamic.py:297-330: torch.cat([d_encoded, geo_feat]) per sample): the direction rows are read once per ray instead of being
 * copied into every sample's row.  fp16 operands, 16-byte aligned rows, prefix_cols % 8 == 0, rows_per_prefix % 16 == 0; everything else
 * as in nvsf_mlp_fwd / nvsf_mlp_bwd (gx_col0 and the columns of grad_x refer to the logical row).  Same results as on assembled rows. */
int nvsf_mlp_fwd_prefix(const void* prefix_f16, uint32_t prefix_stride, uint32_t rows_per_prefix, uint32_t prefix_cols,
                        const void* x_f16, uint32_t M, uint32_t n_in, uint32_t x_stride, const void* weights_f16, uint32_t in_cols,
                        uint32_t hidden, uint32_t n_hidden, uint32_t out_cols, float* out_f32, uint32_t out_stride,
                        nvsf_stream_t stream);
int nvsf_mlp_bwd_prefix(const void* prefix_f16, uint32_t prefix_stride, uint32_t rows_per_prefix, uint32_t prefix_cols,
                        const void* x_f16, uint32_t M, uint32_t n_in, uint32_t x_stride, const void* weights_f16, uint32_t in_cols,
                        uint32_t hidden, uint32_t n_hidden, uint32_t out_cols, const float* grad_out, uint32_t n_out,
                        uint32_t go_stride, float grad_scale, float* grad_x, uint32_t gx_stride, float* grad_weights_f32,
                        uint32_t gx_col0, int gx_accumulate, nvsf_stream_t stream);

/* ref: Planes4D.forward / forward_static / forward_dynamic, nvsf/nerf/models/planes_field.py:86-140,196-238
 * (24 F.grid_sample(bilinear, align_corners=True, padding='border') calls + products + concat per call).
 * xt fp32 [M,4] in [0,1] (16-byte aligned); planes_cl: all 6*n_scales planes CHANNEL-LAST [H][W][C] fp32 in
 * (scale, pair) order, pair = (0,1) (0,2) (0,3) (1,2) (1,3) (2,3), plane (a,b) has W = res[a], H = res[b];
 * h_res host [n_scales][4]; want bit 0: static (xy*xz*yz), bit 1: dynamic (xt*yt*zt);
 * outputs fp32 [M, n_scales*C].  C must be 8. */
int nvsf_planes_fwd(const float* xt, uint32_t M, const float* planes_cl, uint32_t n_scales, uint32_t C,
                    const uint32_t* h_res, int want, float* out_static, float* out_dynamic, nvsf_stream_t stream);

/* The same encoder for several evaluations of ONE position set in one launch: evaluation e is the static (h_group[e] = 0: xy, xz,
 * yz) or dynamic (1: xt, yt, zt) plane group at positions x[m, 0:3] (+ h_offsets[e][m, h_offset_col[e] : +3] when that pointer is
 * not NULL: the scene flow towards a neighbour frame, network_dynamic.py:242-271; same fp32 add as torch.add) and time
 * h_time[e]; features go to h_out[e] fp32 [M, n_scales*C] (16-byte aligned).  x fp32 [M, x_stride >= 3] in [0,1].  At most 4
 * evaluations; the h_* arrays are host arrays of n_evals entries.  n_scales = 4 (the reference's configuration), C = 8.
 * blend != 0: the evaluations must be (static, dynamic, dynamic at neighbour 1, dynamic at neighbour 2); h_out[1] then receives
 * 0.5 d + 0.25 (d1 + d2) -- the neighbour blend of network_dynamic.py:273 -- and h_out[2], h_out[3] are not written (may be NULL).
 * blend == 2: as blend == 1, and h_out[0], h_out[1] are fp16 [M, n_scales * C] rows (the values rounded to nearest even). */
int nvsf_planes_multi_fwd(const float* x, uint32_t x_stride, uint32_t M, const float* planes_cl, uint32_t n_scales, uint32_t C,
                          const uint32_t* h_res, uint32_t n_evals, const int* h_group, const float* const* h_offsets,
                          const uint32_t* h_offset_stride, const uint32_t* h_offset_col, const float* h_time,
                          float* const* h_out, int blend, nvsf_stream_t stream);

/* Backward of nvsf_planes_multi_fwd in one launch per gradient kind: evaluation e (group, offsets, time as above) has the
 * gradient of its features in h_grad_out[e] (fp32 rows of n_scales*C floats, h_grad_stride[e] floats apart -- NULL array: dense rows --
 * times h_grad_scale[e] -- NULL array: 1; NULL pointer: the evaluation contributes nothing).  A slice of a wider gradient matrix is read
 * in place, and the evaluations of the blend 0.5 d + 0.25 (d1 + d2) (nvsf_planes_multi_fwd with blend != 0) share ONE gradient with the
 * scales 0.5, 0.25, 0.25.  grad_planes_cl (may be
 * NULL): the texel gradients of EVERY evaluation are ADDED into it, the evaluations of a group walked round by round with one set of
 * texel-quad sums (addends of evaluations that fall into the same quad are merged before the atomic leaves).  h_grad_offsets[e] (the
 * array or an entry may be NULL): receives d L / d (offset of evaluation e), 3 floats per row at column h_grad_offset_col[e] of rows
 * of h_grad_offset_stride[e] floats -- the gradient the flow field is trained through (network_dynamic.py:250-271).
 * Same addends as one nvsf_planes_bwd per evaluation; the order of the fp32 additions into a texel differs. */
int nvsf_planes_multi_bwd(const float* x, uint32_t x_stride, uint32_t M, const float* planes_cl, uint32_t n_scales, uint32_t C,
                          const uint32_t* h_res, uint32_t n_evals, const int* h_group, const float* const* h_offsets,
                          const uint32_t* h_offset_stride, const uint32_t* h_offset_col, const float* h_time,
                          const float* const* h_grad_out, const uint32_t* h_grad_stride, const float* h_grad_scale,
                          float* grad_planes_cl, float* const* h_grad_offsets, const uint32_t* h_grad_offset_stride,
                          const uint32_t* h_grad_offset_col, nvsf_stream_t stream);

/* autograd of nvsf_planes_fwd: grad_planes_cl (same layout as planes_cl, fp32 atomics, caller zero-initialises; may be
 * NULL) and grad_xt [M,4] (may be NULL). */
int nvsf_planes_bwd(const float* xt, uint32_t M, const float* planes_cl, uint32_t n_scales, uint32_t C,
                    const uint32_t* h_res, int want, const float* grad_static, const float* grad_dynamic,
                    float* grad_planes_cl, float* grad_xt, nvsf_stream_t stream);

/* ref: HashGrid4D.forward_dynamic, nvsf/nerf/models/hash_field.py:148-159 with HashGridT.forward / interpT
 * (:65-88) for the three coordinate pairs (x,y), (x,z), (y,z): two 2-D hash grids (time slices floor / ceil of
 * t*(R-1)), linear blend, cubic Lagrange reduction of the 4 features of each of the 8 levels -> [M,24].
 * x fp32 [M,x_stride] (columns 0..2), optional offset [M,off_stride] whose columns off_col..off_col+2 are added
 * first (the flow warp of network_dynamic.py:243,259).  h_tables_f16: host array of 6 device pointers (lo slice of
 * pair 0,1,2, hi slice of pair 0,1,2); h_scales/h_res [3][8], h_offsets [3][9]; h_time = {k2-idx, idx-k1, w0..w3}
 * (fp32); same_slice != 0 when idx is integral.  mode 0: fp32 arithmetic, out fp32 (tensor t); mode 1: every
 * product / sum rounded to fp16, out fp16 (0-dim t) -- the two type-promotion regimes of the reference. */
int nvsf_hashgrid4d_dynamic_fwd(const float* x, uint32_t x_stride, const float* offset, uint32_t off_stride,
                                uint32_t off_col, uint32_t M, const void* const* h_tables_f16, const float* h_scales,
                                const uint32_t* h_res, const uint32_t* h_offsets, const float* h_time, int same_slice,
                                int mode, void* out, nvsf_stream_t stream);

/* Scalar form of the space-time table gradient: the four features of an entry are the four Lagrange chunks and the two
 * slices of a pair see the same cells, so dL/dtable[slice][row][i] = lag_i * blend_slice * G[row] with one scattered sum
 * G[row] = sum over samples of grad_out[pair][level] * w_corner.  h_sums_f32: 3 device pointers (one per pair) to fp32
 * [rows] buffers, G is ADDED to them; the caller expands G (a sixteenth of the atomic floats of nvsf_hashgrid4d_dynamic_bwd). */
int nvsf_hashgrid4d_dynamic_bwd_scalar(const float* x, uint32_t x_stride, uint32_t M, const float* h_scales, const uint32_t* h_res,
                                       const uint32_t* h_offsets, const float* grad_out, void* const* h_sums_f32,
                                       nvsf_stream_t stream);

/* The same (ref: the autograd of hash_field.py:60-74, 150-156) with the gradient COLUMN-major, fp32 [24][M], as
 * nvsf_density_tail_grad_split writes it: every workgroup of the LDS kernel reads one contiguous column instead of 4 bytes of every
 * 96-byte row.  NVSF_ERR_UNSUPPORTED where the launch would not take the LDS kernel (M < 2^16, a level beyond LDS). */
int nvsf_hashgrid4d_dynamic_bwd_scalar_t(const float* x, uint32_t x_stride, uint32_t M, const float* h_scales, const uint32_t* h_res,
                                         const uint32_t* h_offsets, const float* grad_out_t, void* const* h_sums_f32,
                                         nvsf_stream_t stream);

/* The three space-time evaluations of one density query in one launch (ref: network_dynamic.py:220-271: hash_encoder(x, t)
 * in the fp32 regime and hash_encoder.forward_dynamic(x + flow, t_neighbour) twice in the fp16 regime).  h_tables_f16: 18
 * device pointers = for evaluation e = 0, 1, 2 the lo slice of pair 0,1,2 then the hi slice of pair 0,1,2; h_time: 18 floats
 * = per evaluation {blend_lo, blend_hi, w0..w3}; h_flags: 9 ints = per evaluation {enabled, same_slice, shares the slice
 * tables of evaluation 0}.  offsets fp32 [M, off_stride >= 6]: columns 0..2 warp evaluation 1, columns 3..5 evaluation 2.
 * out0 fp32 [M,24]; out1 / out2 fp16 [M,24].  Bit-identical to three nvsf_hashgrid4d_dynamic_fwd calls: a neighbour whose
 * cell at a level equals the base cell (and whose slices are the base slices) re-uses the base evaluation's gathers. */
int nvsf_hashgrid4d_dynamic3_fwd(const float* x, uint32_t x_stride, const float* offsets, uint32_t off_stride, uint32_t M,
                                 const void* const* h_tables_f16, const float* h_scales, const uint32_t* h_res,
                                 const uint32_t* h_offsets, const float* h_time, const int* h_flags, float* out0, void* out1,
                                 void* out2, nvsf_stream_t stream);

/* Table gradients of nvsf_hashgrid4d_dynamic_fwd, regime 0 (ref: what autograd derives for HashGridT.forward,
 * hash_field.py:76-88, summed over the three planes of HashGrid4D.forward_dynamic :148-159): one launch for the three
 * coordinate pairs and both time slices.  grad_out fp32 [M, 24]; h_grad_tables_f32: 6 device pointers to fp32 buffers
 * in the layout of the slice tables (lo slice of pair 0,1,2, then hi slice of pair 0,1,2 -- ignored when same_slice);
 * dL/dtable is ADDED to them.  h_scales / h_res / h_offsets / h_time as in the forward. */
int nvsf_hashgrid4d_dynamic_bwd(const float* x, uint32_t x_stride, uint32_t M, const float* h_scales, const uint32_t* h_res,
                                const uint32_t* h_offsets, const float* h_time, int same_slice, const float* grad_out,
                                void* const* h_grad_tables_f32, nvsf_stream_t stream);

/* ref: FlowField.forward front end, nvsf/nerf/models/flow_field.py:123-128 (grid_enc -> .float() -> interpT):
 * 3-D hash grid with F = 8 followed by the Lagrange reduction over the 4 feature pairs of each level.
 * h_weights4 = w0..w3 (fp32, host) -> out fp32 [M, 2L]. */
int nvsf_hashgrid3d_lagrange_fwd(const float* x, uint32_t x_stride, uint32_t M, const void* table_f16, uint32_t L,
                                 uint32_t F, const float* h_scales, const uint32_t* h_res, const uint32_t* h_offsets,
                                 const float* h_weights4, float* out, nvsf_stream_t stream);

/* ref: NeRFNetwork.density tail, nvsf/nerf/models/network_dynamic.py:273-287: blend of current / flow-warped
 * neighbour features (0.5, 0.25, 0.25), concatenation to 120 features, sigma_net (120 -> 64 -> 16).
 * plane_* fp32 [M,32]; hash_s fp16 [M,32]; hash_d fp32 [M,24]; hash_1/hash_2 [M,24] fp16 (flag != 0) or fp32.
 * Either out_h fp32 [M,16] (the 16 network outputs) or, when out_h is NULL, sigmas fp32 [M] = exp(h0) and
 * geo fp16 [M,16] = (h1..h15, 1.0).  All pointers 16-byte aligned.  * x_f16_out (optional, fp16 [M,128]): the assembled, rounded network input incl. the ones padding -- what a training step
 * keeps for nvsf_mlp_bwd. */
int nvsf_density_dynamic_fwd(const float* plane_s, const float* plane_d, const float* plane_1, const float* plane_2,
                             const void* hash_s_f16, const float* hash_d, const void* hash_1, int hash_1_is_f16,
                             const void* hash_2, int hash_2_is_f16, uint32_t M, const void* sigma_weights_f16,
                             float* out_h, float* sigmas, void* geo_f16, void* x_f16_out, nvsf_stream_t stream);

/* The same tail fed by nvsf_planes_multi_fwd(..., blend = 2): plane_s and the already blended dynamic plane features as
 * fp16 [M,32] rows (the blend of network_dynamic.py:273 formed by the producer, rounded as this kernel would round it).
 * Bit-identical outputs; 256 B per sample less memory traffic. */
int nvsf_density_dynamic_f16planes_fwd(const void* plane_s_f16, const void* plane_d_blended_f16, const void* hash_s_f16,
                                       const float* hash_d, const void* hash_1, int hash_1_is_f16, const void* hash_2,
                                       int hash_2_is_f16, uint32_t M, const void* sigma_weights_f16, float* out_h,
                                       float* sigmas, void* geo_f16, void* x_f16_out, nvsf_stream_t stream);

/* ... and with the static hash features level-major, fp16 [8][M][4], as nvsf_hashgrid_fwd_level_major writes them (the row
 * form costs its producer eight partial writes per 64-byte row).  Bit-identical outputs (network_dynamic.py:273-287). */
int nvsf_density_dynamic_lm_fwd(const void* plane_s_f16, const void* plane_d_blended_f16, const void* hash_s_level_major_f16,
                                const float* hash_d, const void* hash_1, int hash_1_is_f16, const void* hash_2,
                                int hash_2_is_f16, uint32_t M, const void* sigma_weights_f16, float* out_h,
                                float* sigmas, void* geo_f16, void* x_f16_out, nvsf_stream_t stream);

/* ... the four-buffer form of nvsf_density_dynamic_fwd (fp32 plane rows, blend formed in the kernel: the training forward,
 * network_dynamic.py:273-287) with the static hash features level-major, fp16 [8][M][4]. */
int nvsf_density_dynamic_lm32_fwd(const float* plane_s, const float* plane_d, const float* plane_1, const float* plane_2,
                                  const void* hash_s_level_major_f16, const float* hash_d, const void* hash_1, int hash_1_is_f16,
                                  const void* hash_2, int hash_2_is_f16, uint32_t M, const void* sigma_weights_f16,
                                  float* out_h, float* sigmas, void* geo_f16, void* x_f16_out, nvsf_stream_t stream);

/* ref: the backward of network_dynamic.py:273-287 (autograd of the blends 0.5 d + 0.25 (d1 + d2) and of torch.cat): the density MLP's
 * input gradient grad_x fp32 [M, gx_stride >= 120] handed back per input in one pass -- g_plane_half = 0.5 grad_x[:, 32:64],
 * g_plane_quarter = 0.25 grad_x[:, 32:64], g_hash_s = grad_x[:, 64:96] (fp16 or fp32 rows of 32), g_hash_d_half = 0.5 grad_x[:, 96:120]
 * ([M,24], or [24][M] with hash_d_col_major != 0: the layout nvsf_hashgrid4d_dynamic_bwd_scalar_t reads), g_plane_s = grad_x[:, 0:32] as
 * rows of its own; hash_s_level_major != 0: g_hash_s as [8][M][4] (the layout of nvsf_hashgrid_fwd_level_major's output).  NULL outputs are
 * skipped.  The same values as the elementwise operations.  plane_half_scale: the factor of g_plane_half -- 0.5 for the blend of three
 * separate tensors; 1 where the producer already blended them (plane_d = plane_1 = plane_2 one tensor, whose gradient is the whole slice). */
int nvsf_density_tail_grad_split(const float* grad_x, uint32_t gx_stride, uint32_t M, float* g_plane_half, float* g_plane_quarter,
                                 void* g_hash_s, int hash_s_is_f16, int hash_s_level_major, float* g_hash_d_half, int hash_d_col_major,
                                 float* g_plane_s, float plane_half_scale, nvsf_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Section 4: fused kernels of the uniform-sampling render (BASELINE config 2 hot path).
 * ---------------------------------------------------------------------------------------------- */

/* ref: renderer_dynamic.py:155-174 + network_dynamic.py:213-287 for a static hash field:
 * sample generation -> normalise (x+bound)/(2 bound) -> 3-D hash grid (L*F = 32) -> sigma net 32->64->16
 * -> sigma = exp(h0) (activation.py:9-11).  h_aabb[6] host.  Outputs z_vals, sigmas [N,T] fp32 and
 * geo fp16 [N,T,16] = (h1..h15, 1.0). */
int nvsf_field_density_uniform_fwd(const float* rays_o, const float* rays_d, const float* nears, const float* fars,
                                   const float* lin, const float* noise, const float* h_aabb, float bound,
                                   uint32_t N, uint32_t T, const void* table_f16, uint32_t L, uint32_t F,
                                   const float* h_scales, const uint32_t* h_res, const uint32_t* h_offsets,
                                   const void* sigma_weights_f16, float* z_vals, float* sigmas, void* geo_f16,
                                   nvsf_stream_t stream);

/* ref: NeRFRenderer.run for a static hash field, renderer_dynamic.py:155-244 (sample generation, density, alpha
 * compositing, masked colour, image), evaluation only: the whole render of a ray in one wave, ONE launch per batch.
 * Same arithmetic as nvsf_field_density_uniform_fwd -> nvsf_composite_uniform_weights_fwd ->
 * nvsf_field_heads_uniform_fwd, but sigma and the geometry features stay in registers (the three-kernel form writes
 * and re-reads 48 B per sample) and the transmittance product is scanned per 16 samples.  Requires L == 16, F == 2, or
 * L == 8, F == 4 (the reference's default grid, main_nvsf.py:45-52) with feat_scratch given (NVSF_ERR_UNSUPPORTED otherwise).
 * feat_scratch: NULL (the kernel gathers from the table itself) or the feature planes filled by
 * nvsf_field_density_uniform_sliced_fwd(passes = 1), which must also have written z_vals.
 * Outputs: z_vals, weights [N,T]; weights_sum, depth [N]; image [N,3] (+ (1 - weights_sum) * h_bg_color when
 * h_bg_color != NULL) or [N,2] (raydrop, intensity) for lidar != 0.  Colour is evaluated where weight > w_thresh. */
int nvsf_render_uniform_fwd(const float* rays_o, const float* rays_d, const float* nears, const float* fars,
                            const float* lin, const float* noise, const float* h_aabb, float bound, uint32_t N,
                            uint32_t T, const void* table_f16, uint32_t L, uint32_t F, const float* h_scales,
                            const uint32_t* h_res, const uint32_t* h_offsets, const void* sigma_weights_f16, int lidar,
                            const void* head_a_weights_f16, const void* head_b_weights_f16, float k_scale,
                            float w_thresh, const float* h_bg_color, const void* feat_scratch, float* z_vals,
                            float* weights, float* weights_sum, float* depth, float* image, nvsf_stream_t stream);

/* The TRAINING forward of the same render (renderer_dynamic.py:155-237 under autograd; field_ops.RenderRaysFn): the launch(es) of
 * nvsf_render_uniform_fwd -- with feat_scratch the level-sliced encode pass runs first, inside this call -- which additionally keep
 * what the backward of the whole render reads: x01 [M,3] unit-cube positions, feat_rows_f16 [M,32] encoded features (column 2 l + f),
 * geo_f16 [M,16] = (h1 .. h15, 1.0), sigmas [M] = exp(h0), rgbs [M,C] = [weight > w_thresh] sigmoid(logits); M = N T.  Same z_vals /
 * weights / weights_sum / depth / image as nvsf_render_uniform_fwd bit for bit.  Replaces, in the training graph, the chain
 * nvsf_field_density_uniform_train_fwd -> nvsf_composite_uniform_weights_fwd -> nvsf_mlp_fwd_prefix (x 1-2) -> nvsf_masked_sigmoid ->
 * nvsf_composite_uniform_image_fwd and their intermediate [M,16] fp32 outputs, logits and mask. */
int nvsf_render_uniform_train_fwd(const float* rays_o, const float* rays_d, const float* nears, const float* fars,
                                  const float* lin, const float* noise, const float* h_aabb, float bound, uint32_t N,
                                  uint32_t T, const void* table_f16, uint32_t L, uint32_t F, const float* h_scales,
                                  const uint32_t* h_res, const uint32_t* h_offsets, const void* sigma_weights_f16, int lidar,
                                  const void* head_a_weights_f16, const void* head_b_weights_f16, float k_scale,
                                  float w_thresh, const float* h_bg_color, void* feat_scratch, float* z_vals, float* weights,
                                  float* weights_sum, float* depth, float* image, float* x01, void* feat_rows_f16,
                                  void* geo_f16, float* sigmas, float* rgbs, nvsf_stream_t stream);

/* ref: the evaluation-mode protocol of the raymarching extension, raymarching.py:389-409 (march_rays) + 480-493
 * (composite_rays) around the field, for a static hash field (L*F = 32 with F = 2 or F = 4): ONE launch instead of the host
 * loop over surviving rays.  Per ray: march through the occupancy bit field `grid` (layout of nvsf_march_rays),
 * field on every sample (sigma = exp(h0) * density_scale; colour = sigmoid(heads), LiDAR: raydrop, intensity),
 * alpha compositing in sample order, stop at the first sample whose incoming transmittance < T_thresh, at `far`,
 * or after max_steps samples.  Outputs weights_sum, depth [N]; image [N,3] (+ (1 - weights_sum) * h_bg_color,
 * h_bg_color = 3 host floats or NULL) or [N,2] for LiDAR.  No perturbation (the host loop applies its noise
 * offset to the first call only, which ties the result to the loop's batching).
 * Deviations from the host loop: the loop restarts each march call from the compositor's
 * accumulated t (1-ulp differences in later sample positions) and caps a ray at max_steps .. max_steps+7 samples
 * depending on how many rays survive; this kernel marches each ray continuously and caps at max_steps. */
int nvsf_render_occupancy_fwd(const float* rays_o, const float* rays_d, const float* nears, const float* fars,
                              const uint8_t* grid, float bound, float dt_gamma, uint32_t max_steps, uint32_t C,
                              uint32_t H, uint32_t N, const void* table_f16, uint32_t L, uint32_t F,
                              const float* h_scales, const uint32_t* h_res, const uint32_t* h_offsets,
                              const void* sigma_weights_f16, int lidar, const void* head_a_weights_f16,
                              const void* head_b_weights_f16, float density_scale, float T_thresh,
                              const float* h_bg_color, float* weights_sum, float* depth, float* image,
                              nvsf_stream_t stream);

/* The TRAINING forward of the same operator (ops.DensityRaysFn): one launch (feat_scratch NULL) or the two launches of the
 * level-sliced form (feat_scratch = 4 * L * N * T bytes) that, beside z_vals / sigmas / geo_f16 (same values bit for bit),
 * keep what the backward needs: x01 [N*T, 3] fp32 = the samples' unit-cube positions ((clip(o + d z) + bound) / (2 bound):
 * network_dynamic.py:217), feat_rows_f16 [N*T, 32] fp16 = the encoded features in level order (the input rows of the density
 * MLP: what nvsf_hashgrid_fwd writes) and h32 [N*T, 16] fp32 = the MLP's outputs (h0 = density logit, h1..h15 = geometry
 * features: what nvsf_mlp_fwd writes).  Replaces nvsf_uniform_samples + the unit-cube normalisation + nvsf_hashgrid_fwd +
 * nvsf_mlp_fwd + nvsf_exp_col of the operator chain.  Requires L == 16, F == 2. */
int nvsf_field_density_uniform_train_fwd(const float* rays_o, const float* rays_d, const float* nears, const float* fars,
                                         const float* lin, const float* noise, const float* h_aabb, float bound, uint32_t N,
                                         uint32_t T, const void* table_f16, uint32_t L, uint32_t F, const float* h_scales,
                                         const uint32_t* h_res, const uint32_t* h_offsets, const void* sigma_weights_f16,
                                         float* z_vals, float* sigmas, void* geo_f16, float* x01, void* feat_rows_f16, float* h32,
                                         void* feat_scratch, nvsf_stream_t stream);

/* Same operator, same results bit for bit, as two launches with the LEVELS partitioned over the 8 XCDs (each
 * XCD's L2 then holds 2 of the 16 levels): pays when consecutive samples of a ray are several finest-level cells
 * apart, i.e. when the fine levels have no reuse along the ray (camera rays through the whole box).
 * feat_scratch: device buffer of 4 * L * N * T bytes (encoded features, written and read once).
 * passes: 3 = both launches; 1 = encode only (fills feat_scratch and z_vals), 2 = MLP only (reads feat_scratch) --
 * the split exists so that each launch can be timed on its own.
 * Requires L == 16, F == 2, N*T < 2^32 -- or L == 8, F == 4, N*T < 2^28: one level per XCD, the same planes by column pair
 * (plane 2g = columns {2g, 2g+1, 2g+24, 2g+25}, plane 2g+1 = columns {2g+8, 2g+9, 2g+16, 2g+17} of the [32] feature row);
 * NVSF_ERR_UNSUPPORTED otherwise. */
int nvsf_field_density_uniform_sliced_fwd(const float* rays_o, const float* rays_d, const float* nears,
                                          const float* fars, const float* lin, const float* noise,
                                          const float* h_aabb, float bound, uint32_t N, uint32_t T,
                                          const void* table_f16, uint32_t L, uint32_t F, const float* h_scales,
                                          const uint32_t* h_res, const uint32_t* h_offsets,
                                          const void* sigma_weights_f16, float* z_vals, float* sigmas,
                                          void* geo_f16, void* feat_scratch, uint32_t passes,
                                          nvsf_stream_t stream);

/* ref: renderer_dynamic.py:202-237 + network_dynamic.py:290-332.  Direction encoding (per ray), heads,
 * sigmoid, weight mask (w > w_thresh) and image accumulation in one kernel.
 * lidar == 0: head_a = colour net [SH16|geo15|1] 32->64->64->3;  image [N,3] (+ (1-ws)*h_bg_color[3])
 * lidar != 0: head_a = raydrop net, head_b = intensity net, [Freq72|geo15|1x9] 96->64->64->1; image [N,2].
 * h_bg_color: host pointer or NULL. */
int nvsf_field_heads_uniform_fwd(const float* weights, const void* geo_f16, const float* rays_d,
                                 const float* weights_sum, int lidar, const void* head_a_weights_f16,
                                 const void* head_b_weights_f16, uint32_t N, uint32_t T, float w_thresh,
                                 const float* h_bg_color, float* image, nvsf_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Section 5 ("next" rows of SURVEY 8f): loss-side kernel adjacent to the path.
 * ---------------------------------------------------------------------------------------------- */

/* ref: chamfer_3D.forward, nvsf/nerf/chamfer3D/chamfer_cuda.cpp + chamfer3D.cu:9-165 (NmDistanceKernel x2).
 * xyz1 [B,n,3], xyz2 [B,m,3] -> dist1 [B,n] (squared distance to the nearest point of xyz2), idx1 int32 [B,n],
 * dist2 [B,m], idx2 [B,m].  workspace_u64: B*max(n,m) 64-bit words of scratch owned by the caller. */
int nvsf_chamfer_forward(const float* xyz1, const float* xyz2, uint32_t B, uint32_t n, uint32_t m, float* dist1,
                         float* dist2, int32_t* idx1, int32_t* idx2, void* workspace_u64, nvsf_stream_t stream);

/* ref: chamfer_3D.backward, chamfer3D.cu:167-234.  grad_xyz1 [B,n,3] is written whole (no zero fill needed: its rows are stored by
 * the first direction before the second adds); grad_xyz2 [B,m,3] zero-initialised by the caller (dist_chamfer_3D.py:79-80).  Either
 * may be NULL (that cloud needs no gradient -- e.g. the measured cloud of the training loss), not both. */
int nvsf_chamfer_backward(const float* xyz1, const float* xyz2, uint32_t B, uint32_t n, uint32_t m,
                          const float* grad_dist1, const float* grad_dist2, const int32_t* idx1, const int32_t* idx2,
                          float* grad_xyz1, float* grad_xyz2, nvsf_stream_t stream);

/* ref: get_lidar_rays, nvsf/nerf/dataset/dataset_utils.py:369-536 (direction model :506-528).
 * pose44 device fp32 [4,4] row-major (sensor-to-world); inds device int64 [N] row-major pixel indices of the H x W
 * range image, or NULL for all pixels 0..N-1; fov_up, fov, fov_hoz in degrees -> rays_o, rays_d [N,3]. */
int nvsf_lidar_rays(const float* pose44, const int64_t* inds, uint32_t N, uint32_t H, uint32_t W, float fov_up,
                    float fov, float fov_hoz, float* rays_o, float* rays_d, nvsf_stream_t stream);

/* ref: get_rays, dataset_utils.py:539-687 (pixel centre +0.5, pinhole, normalised, rotated by the pose). */
int nvsf_camera_rays(const float* pose44, const int64_t* inds, uint32_t N, uint32_t W, float fx, float fy, float cx,
                     float cy, float* rays_o, float* rays_d, nvsf_stream_t stream);

/* ---- 7. optimiser step of the thin trainer ------------------------------------------------------- */

/* ref: torch.optim.Adam(betas=(0.9, 0.99), eps=1e-15) of main_nvsf.py:350-352 under GradScaler (trainer.py:119, 1332-1334).
 * state4 (device, 4 floats) = {step, 1 - beta1^step, sqrt(1 - beta2^step), skip}.  prepare: skip = found_inf && *found_inf != 0;
 * step += !skip; corrections recomputed (double).  found_inf may be NULL. */
int nvsf_adam_prepare(float* state4, const float* found_inf, float beta1, float beta2, nvsf_stream_t stream);

/* One pass over a parameter tensor (all fp32 [n]): g = grad / *grad_scale (grad_scale NULL = 1); exp_avg += (1 - beta1)(g - exp_avg);
 * exp_avg_sq = beta2 exp_avg_sq + (1 - beta2) g g; param -= lr / state4[1] * exp_avg / (sqrt(exp_avg_sq) / state4[2] + eps).
 * Nothing is written when state4[3] != 0 (overflow: the step is skipped).
 * ema_shadow (fp32 [n], may be NULL): shadow -= ema_one_minus_decay * (shadow - param) with the UPDATED parameter, in the same
 * pass (an every-step exponential moving average of the weights; the reference's per-epoch one is nvsf_ema_update). */
/* param_f16 (fp16 [n], may be NULL): the updated parameter rounded to fp16, written in the same pass -- the copy the forward kernels
 * read (hash tables, MLP weights), which otherwise costs a cast pass per parameter and step. */
int nvsf_adam_update(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, uint64_t n, float lr, float beta1,
                     float beta2, float eps, const float* state4, const float* grad_scale, float* ema_shadow,
                     float ema_one_minus_decay, void* param_f16, nvsf_stream_t stream);

/* ref: the LiDAR terms of Trainer.train_step, nvsf/nerf/trainer.py:187-219 (criteria with reduction = "none", main_nvsf.py:205-221;
 * summed at trainer.py:540-543) and the point clouds of its chamfer term (trainer.py:229-233), all [N] fp32 unless noted:
 *   m = gt_raydrop; pred_depth = depth_lidar m; loss_depth = sum alpha_d |pred_depth - gt_range m|;
 *   loss_raydrop = sum alpha_r (image_lidar[:, 0] - clamp(m, s, 1 - s))^2;  loss_intensity = sum alpha_i (image_lidar[:, 1] m - gt_intensity m)^2;
 *   pred_points = rays_d pred_depth / scale, gt_points = rays_d gt_range m / scale ([N,3]; both NULL: not formed).
 * image_lidar [N,2]; the three losses are single device floats; one launch, deterministic sums. */
int nvsf_lidar_losses_fwd(const float* image_lidar, const float* depth_lidar, const float* gt_raydrop, const float* gt_intensity,
                          const float* gt_range, const float* rays_d, uint32_t N, float alpha_d, float alpha_r, float alpha_i,
                          float smooth_factor, float scale, float* loss_depth, float* loss_raydrop, float* loss_intensity,
                          float* pred_depth, float* pred_points, float* gt_points, nvsf_stream_t stream);

/* Gradients of the above with respect to image_lidar [N,2] and depth_lidar [N], given the gradients of the three sums (device
 * floats; NULL = 0), of pred_depth ([N] or NULL) and of pred_points ([N,3] or NULL: what nvsf_chamfer_backward returns). */
int nvsf_lidar_losses_bwd(const float* image_lidar, const float* depth_lidar, const float* gt_raydrop, const float* gt_intensity,
                          const float* gt_range, const float* rays_d, uint32_t N, float alpha_d, float alpha_r, float alpha_i,
                          float smooth_factor, float scale, const float* grad_loss_depth, const float* grad_loss_raydrop,
                          const float* grad_loss_intensity, const float* grad_pred_depth, const float* grad_pred_points,
                          float* grad_image_lidar, float* grad_depth_lidar, nvsf_stream_t stream);

/* ref: the structural regularisation of Trainer.train_step on LiDAR patches, its `grad_loss` branch: nvsf/nerf/trainer.py:296-470
 * (switched on by configs/kitti360_1908.txt:13; patches of change_patch_size_lidar = [2, 8] every second epoch, trainer.py:1035-1062).
 * The batch is N / (patch_h patch_w) patches in row-major patch order; pred_depth / gt_depth [N]: the MASKED range in scene units
 * (pred_depth of nvsf_lidar_losses_fwd, gt_range * gt_raydrop); pano_inds int64 [N]: pixel indices h W + w in the frame's range
 * image; pano_range: the range channel of that frame ([H, W], element stride pano_stride floats: 3 for the [H, W, 3] ground truth).
 * First differences with the last column / row repeated, in metres (/ scale); masks = gt_raydrop x (|second difference of the TRUE
 * frame at the pixel| < 0.05); loss = alpha sum crit(grad_x(pred) m_x, grad_x(gt) m_x) + the same in y; criterion 0 L1, 1 MSE,
 * 2 Huber(delta = criterion_param), 3 SmoothL1(beta = criterion_param) (main_nvsf.py:204-221, `--depth_grad_loss`), 4 cosine
 * (trainer.py:442-452: per patch and direction 1 - <u, v> / (max(|u|, 1e-8) max(|v|, 1e-8)) over the patch's masked gradients,
 * expanded over the patch and summed; patch_stats float [N / (patch_h patch_w), 6] receives the three inner products per
 * direction for the backward, may be NULL for criteria 0-3).  sobel != 0 (`--sobel_grad`, trainer.py:316-328, 367-380): the
 * gradients of both range images are the 3 x 3 Sobel cross-correlations with zero padding at the patch border instead of first
 * differences (the masks stay those of the true frame).  Not built: `--grad_norm_smooth`, `--spatial_smooth`, `--tv_loss`
 * (trainer.py:337-350 add a [num_patch, 1, pH, pW] tensor to the loss, which Trainer.train_one_epoch cannot back-propagate: the
 * options fail in the reference itself).  One launch, deterministic sum. */
int nvsf_lidar_grad_loss_fwd(const float* pred_depth, const float* gt_depth, const float* gt_raydrop, const int64_t* pano_inds,
                             const float* pano_range, uint32_t pano_stride, uint32_t N, uint32_t patch_h, uint32_t patch_w, uint32_t H,
                             uint32_t W, float scale, int criterion, float criterion_param, float alpha, int sobel, float* patch_stats,
                             float* loss, nvsf_stream_t stream);
/* gradient of the above with respect to pred_depth [N], given the gradient of the loss (a device float) and, for the cosine
 * criterion, the patch_stats the forward wrote */
int nvsf_lidar_grad_loss_bwd(const float* pred_depth, const float* gt_depth, const float* gt_raydrop, const int64_t* pano_inds,
                             const float* pano_range, uint32_t pano_stride, uint32_t N, uint32_t patch_h, uint32_t patch_w, uint32_t H,
                             uint32_t W, float scale, int criterion, float criterion_param, float alpha, int sobel, const float* patch_stats,
                             const float* grad_loss, float* grad_pred_depth, nvsf_stream_t stream);

/* ref: the error map behind the pixel sampler, nvsf/nerf/trainer.py:552-630 (`--use_error_map`, configs/kitti360_1908.txt:14).
 * nvsf_lidar_ray_losses: the per-ray LiDAR loss `lidar_loss` of trainer.py:213-216 (the three terms of nvsf_lidar_losses_fwd, not
 * summed); nvsf_mse_rows: row_loss[n] = sum_c alpha (a[n, c] - b[n, c])^2 (the camera's `rgb_loss.sum(dim=2)`, trainer.py:598).
 * min_max_bits (uint32 [2], may be NULL): running (min, max) of the written values as unsigned bit patterns -- the caller sets it
 * to (0x7f800000, 0) first. */
int nvsf_lidar_ray_losses(const float* image_lidar, const float* depth_lidar, const float* gt_raydrop, const float* gt_intensity,
                          const float* gt_range, uint32_t N, float alpha_d, float alpha_r, float alpha_i, float smooth_factor,
                          float* ray_loss, uint32_t* min_max_bits, nvsf_stream_t stream);
int nvsf_mse_rows(const float* a, const float* b, uint32_t N, uint32_t C, float alpha, float* row_loss, uint32_t* min_max_bits,
                  nvsf_stream_t stream);
/* error = (ray_loss - min) / (max - min + eps) * 999 + 1;  cell = (floor(h scale_h), floor(w scale_w)) with (h, w) = divmod(pixel_inds, W)
 * and scale_h = map_h / H, scale_w = map_w / W evaluated by the caller as the reference does (Python floats);
 * error_map[cell] = 0.1 error_map[cell] + 0.9 error (trainer.py:566-583).  Where several rays fall into one cell the reference's indexed
 * assignment keeps an unspecified one; here the ray with the largest index.  owner: uint32 [map_h map_w] scratch, zero on entry, zero
 * again on return.  Two launches. */
int nvsf_error_map_update(const float* ray_loss, const int64_t* pixel_inds, uint32_t N, uint32_t W, float* error_map, uint32_t map_h,
                          uint32_t map_w, float scale_h, float scale_w, const uint32_t* min_max_bits, uint32_t* owner, nvsf_stream_t stream);

/* ref: the camera term, trainer.py:491-503: loss = sum alpha (a - b)^2 over n floats (one launch); grad_a = grad_loss 2 alpha (a - b). */
int nvsf_mse_sum_fwd(const float* a, const float* b, uint32_t n, float alpha, float* loss, nvsf_stream_t stream);
int nvsf_mse_sum_bwd(const float* a, const float* b, uint32_t n, float alpha, const float* grad_loss, float* grad_a,
                     nvsf_stream_t stream);

/* ref: torch_ema.ExponentialMovingAverage.update as used by the Trainer (trainer.py:112-114 construction with decay 0.95,
 * :1420-1421 one update per epoch): shadow -= one_minus_decay * (shadow - param), fp32 [n]. */
int nvsf_ema_update(float* shadow, const float* param, uint64_t n, float one_minus_decay, nvsf_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* NVSF_HIP_H */

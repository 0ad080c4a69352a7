/*
 * sdfx_devtools.h — entry points that exist ONLY in the devtools build of the library
 * (stable-dreamfusion_amd/csrc/libsdfx_hip_dev.so: `python stable-dreamfusion_amd/build.py --devtools`, -DSDFX_DEVTOOLS).
 *
 * The product library (libsdfx_hip.so, include/sdfx.h) has no process-global implementation switches and never reads the
 * environment: every switch below is a compile-time constant there. The devtools library carries the superseded kernels
 * (thread-per-ray march count, per-thread v_dot2 field kernels) and the measurement knobs that tools/ and the A/B tests of
 * tests/test_gpu_zz_stress.py use; select it with SDFX_LIB=<path to libsdfx_hip_dev.so>.
 *
 * A switch is resolved from sdfx_dev_set() if it was called, else from the environment variable of the same name read ONCE
 * at its first use, else it keeps the product default:
 *   SDFX_MARCH_WAVE        1 (default) wave-per-ray counting pass, 0 thread-per-ray            raymarching.hip
 *   SDFX_GRID_FWD          1 (default) k_grid_fwd for hinted batches, 0 generic kernel         gridencoder_fwd.hip
 *   SDFX_GRID_BALANCE      1 (default) cost-balanced per-XCD level ranges, 0 equal counts      gridencoder_fwd.hip
 *   SDFX_GRID_VALU_LINES   97: VALU cost of a tile in line units (plan model)                  gridencoder_fwd.hip
 *   SDFX_GRID_LDS          0: largest level table (bytes) gathered from LDS (slower)           gridencoder_fwd.hip
 *   SDFX_GRID_PLAN         (string, environment only) "sample_major" (slower)                  gridencoder_fwd.hip
 *   SDFX_GRID_PLAN_DEBUG   print the per-XCD segments                                          gridencoder_fwd.hip
 *   SDFX_GRID_ONLY_LEVEL   l: the hinted forward evaluates level l alone on all eight XCDs     gridencoder_fwd.hip
 *   SDFX_GRID_TPW / _FINE  8 / 2: consecutive tiles per workgroup at the VALU-bound levels / the others   gridencoder_fwd.hip
 *   SDFX_GRID_PAIR         1 (default) half tables: two levels per wave (k_grid_fwd_pair), 0 one level per workgroup   gridencoder_fwd.hip
 *   SDFX_GRID_TPW_PAIR     2: consecutive tiles per workgroup of the pair plan                  gridencoder_fwd.hip
 *   SDFX_GRID_SCALAR_BELOW 0: levels of resolution < r gather corners with 4-byte loads        gridencoder_fwd.hip
 *   SDFX_GRID_SCALAR_FROM  0: ... and levels of resolution >= r (0: none); dense levels always take their two-row loads
 *   SDFX_GRID_COST_TABLE   1 (default) measured per-tile costs for stencil batches, 0 max(lines, VALU floor)   gridencoder_fwd.hip
 *   SDFX_GRID_LEVEL_COST   (string, environment only) "c0,c1,...": cost per tile by level      gridencoder_fwd.hip
 *   SDFX_GRID_NOVEC16      1: one gather per corner in the generic kernels                     gridencoder.hip
 *   SDFX_GRIDBWD_MERGE_RES / _COARSE_SPLIT / _BALANCE / _LEVEL_COST (string)                   gridencoder_bwd_binned.hip
 *   SDFX_GRIDBWD_OVERLAP   0 (default) one stream, 1 K2 of the fine levels on a side stream beside K1 of the coarse ones (slower)   gridencoder_bwd_binned.hip
 *   SDFX_FIELD_IMPL        0 (default) matrix-core kernels, 1 per-thread v_dot2 kernels        field.hip
 *   SDFX_FIELD_FWD_NAT / _FWD_BLOCKS / _BWD_NAT / _BWD_NB / _BWD_LDSFRAG                      field.hip
 *   SDFX_DEV_ABLATE        bits: parts of k_grid_bwd_bin left out (1 list stores, 2 staging, 4 reservations,   gridencoder_bwd_binned.hip
 *                          8 histogram atomics, 16 gradient load, 32 coordinate loads); 64: k_grid_bwd_reduce_fixed adds
 *                          every lane of a wave to a different row (prices same-row LDS atomics)
 *   SDFX_RENDER_WAVES, SDFX_INFER_WAVES                                                        render.hip, infer.hip
 */
#ifndef SDFX_DEVTOOLS_H
#define SDFX_DEVTOOLS_H

#ifdef __cplusplus
extern "C" {
#endif

void sdfx_dev_set(const char* name, int value);
void sdfx_dev_unset(const char* name);
/* Per-workgroup timestamps of the instrumented kernels (csrc/dev_stamps.h: k_grid_fwd = 1, k_grid_bwd_bin = 2,
 * k_grid_bwd_reduce_fixed = 3): `buf` = device memory of (2 + 3 * 4 * cap) 64-bit words zeroed by the caller (cap = workgroups of the largest
 * launch), NULL = off.
 * tools/xcd_timeline.py reduces the records to a per-XCD busy timeline. */
void sdfx_dev_stamps(void* buf, uint32_t cap);

#ifdef __cplusplus
}
#endif
#endif

/* mi355kkt_debug.h -- developer switches of libmi355kkt.so, compiled ONLY with -DMI355KKT_DEBUG (bash cvxopt_amd/csrc/build.sh
 * --debug builds cvxopt_amd/libmi355kkt_debug.so next to the production library; load it with $CVXOPT_AMD_LIB).
 *
 * These are process-global and some of them make results WRONG on purpose (ablation of kernel phases for timing experiments):
 * a production libmi355kkt.so does not contain them, the kernels compile the switches as the constant 0, and in such a build
 * csrc/knobs.h falls back to the environment for the knobs of mi355kkt_test_set_knob(). */
#ifndef MI355KKT_DEBUG_H
#define MI355KKT_DEBUG_H

#ifdef MI355KKT_DEBUG
#ifdef __cplusplus
extern "C" {
#endif

/* HW_ID / XCC_ID of nblocks one-wave workgroups (out: 2 * nblocks words, host): how the dispatcher places workgroups */
int mi355kkt_debug_hwid(unsigned* out, int nblocks);
/* device buffer of 48 int64 shader-clock stamps written by the diagonal-block kernel at its phase boundaries (NULL: off) */
int mi355kkt_debug_potf2_ts(void* dptr);
/* 8 int64 stamps per 128 x 128 tile (column-major tile order) written by the persistent Cholesky kernel (NULL: off) */
int mi355kkt_debug_tile_ts(void* dptr);
/* 16 int64 stamps (s_memrealtime: 100 MHz, common to all compute units) per workgroup of the 512-row triangular solve (NULL: off) */
int mi355kkt_debug_wide_ts(void* dptr);
/* ablation of the SYRK's phases (bit0 no global fetch, bit1 no LDS stash, bit2 no barrier, bit4 no static priority, bit5 long
 * diagonal tiles through the general path, bit6 no block masks): RESULTS ARE WRONG when != 0 */
int mi355kkt_debug_syrk_skip(int mask);

#ifdef __cplusplus
}
#endif
#endif /* MI355KKT_DEBUG */
#endif /* MI355KKT_DEBUG_H */

/* mi355kkt_test.h -- test hooks of libmi355kkt.so: NOT part of the drop-in boundary (include/mi355kkt.h).
 *
 * Everything here is a pure function of its arguments (host executions of device-side source, plans, orderings) or an explicit
 * knob setter; none of it is called by the product path, none of it reads the environment.  tests/ uses them to check the
 * device-side building blocks on the CPU against the reference's misc / misc_solvers functions and against NumPy executions of
 * the plans.  (Developer switches that change results on purpose live in mi355kkt_debug.h and exist only in -DMI355KKT_DEBUG
 * builds.) */
#ifndef MI355KKT_TEST_H
#define MI355KKT_TEST_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* host-only: the complete symbolic plan (supernodes, row lists, storage offsets, extend-add maps, assembly lists) as one flat
 * int64 array -- layout in csrc/capi.hip; returns its length (cap = 0 sizes it) or a negative error code */
int64_t mi355kkt_test_symbolic_plan(int n, int m, const int64_t* gcolptr, const int64_t* growind, const int64_t* hcolptr,
                                    const int64_t* hrowind, int64_t* out, int64_t cap);
/* the second-order-cone operations of the device-resident loops (csrc/cone_ops.h) executed on the HOST, one cone: for the CPU
 * parity tests against misc.sprod / sinv / ssqr / scale2 / scale / jnrm2 / compute_scaling / update_scaling / max_step
 * (reference src/python/misc.py:284-573, :575-760, :1018-1052) */
int mi355kkt_test_cone_op_host(int op, int mk, int arg, double* x, double* y, double* w);
/* the same for one 's' block of order m (csrc/cone_ops_s.h): on the host with one thread, on the host with a team of nt
 * threads executing the workgroup-cooperative code paths, and on the device (team: threads of the workgroup) */
int mi355kkt_test_sdp_op_host(int op, int m, int arg, double* x, double* y, double* r, double* rti, double* lam);
int mi355kkt_test_sdp_op_host_team(int op, int m, int arg, int nt, double* x, double* y, double* r, double* rti, double* lam);
int mi355kkt_test_sdp_op_device(int op, int m, int arg, int team, double* x, double* y, double* r, double* rti, double* lam);
/* the static work list of the scaled SYRK (host only): 8 ints per segment = ti, tj, k0, k1, slot, first, nparts, next; the first
 * *nlaunch segments are the launch's workgroups, the rest continuation segments of the stream-K remainder round (next = 1 + index);
 * returns #segments */
int mi355kkt_test_syrk_plan(int n, int K, int num_cus, int allow_split, int* out, int max_items, int* nslabs, int* nsplit,
                            int* nlaunch);
/* fill-reducing ordering of a symmetric CSC pattern (host only; csrc/ordering.cpp -- the step cholmod.symbolic performs through
 * cholmod_analyze_p, reference src/C/cholmod.c:309): method 0 choose / 1 nested dissection / 2 approximate minimum degree;
 * perm[new] = old; stats[8] = chosen method, nnz and flops of both candidates, supernodal tree heights, count cross-check */
int mi355kkt_test_ordering(int n, const int64_t* colptr, const int64_t* rowind, int method, int* perm, double* stats);
/* throws inside a guarded entry point (kind 0: std::bad_alloc, 1: std::runtime_error, 2: a non-standard exception);
 * must RETURN MI355KKT_ENOMEM / MI355KKT_EHIP like any entry point in which host code throws */
int mi355kkt_test_throw(int kind);
/* 1 when [ptr, ptr + bytes) touches the process's brk heap (the "[heap]" line of /proc/self/maps), 0 when it does not: the test by
 * which mi355kkt_set_H_dense_async decides never to hipHostRegister a caller's buffer that malloc may recycle (host only) */
int mi355kkt_test_touches_brk_heap(const void* ptr, size_t bytes);
/* Knobs of the sparse symbolic analysis and a few kernel-selection thresholds (csrc/knobs.h): "MI355KKT_ORDERING" = nd | amd,
 * "MI355KKT_ND_MODE", "MI355KKT_ND_LEAF", "MI355KKT_ND_LEAF_AMD", "MI355KKT_ND_NOREFINE", "MI355KKT_ORDERING_BOTH",
 * "MI355KKT_SN_MAXW", "MI355KKT_SPARSE_BIG_FLOPS", "MI355KKT_SPARSE_BIG_H", "MI355KKT_SP_WIDE", "MI355KKT_SPARSE_TILES",
 * "MI355KKT_SDP_WAVE_MAX", "MI355KKT_SDP_NO_MFMA", "MI355KKT_SPARSE_DEBUG", "MI355KKT_ND_DEBUG", "MI355KKT_SPARSE_POISON",
 * "MI355KKT_TRSV_PAIR" (0: the one-sweep triangular solve), "MI355KKT_TRSV_WIDE" (0: no 512-row all-CU solves; 128: only for orders that are multiples of 128),
 * "MI355KKT_SPARSE_NO_DENSE_ROOT", and the allocator modes below.  They are set ONLY by this
 * call -- the library never reads them from the environment -- so that tests can drive every ordering / plan shape through the
 * same code.  value == NULL unsets one knob, name == NULL all of them.  Process-wide; returns 0. */
int mi355kkt_test_set_knob(const char* name, const char* value);

/* Device-allocator test modes (csrc/devmem.cpp), knobs of mi355kkt_test_set_knob: "MI355KKT_ALLOC_POISON" (blocks start as 0xff
 * bytes), "MI355KKT_ALLOC_RAW" (blocks are not cleared), "MI355KKT_ALLOC_GUARD" (every block ends where its own mapping ends: an
 * out-of-bounds access of a kernel is a GPU memory fault or reads 0xff poison at once), "MI355KKT_PIN_SMALL_H" (set_H_dense_async pins
 * a host H of any size in place, as the round-4 build did).
 * mi355kkt_test_install_abort_dump: on SIGABRT the ring of the last 65536 allocations / releases (pointer, bytes, call site) is
 * written to `path` before the previous handler runs -- maps the address of a reported GPU memory fault to its owner
 * (tools/alloc_owner.py).  mi355kkt_test_guard_probe: reads element `at` of a fresh block of ndoubles doubles from a kernel. */
int mi355kkt_test_install_abort_dump(const char* path);
int mi355kkt_test_guard_probe(int ndoubles, int at, double* out);
/* MI355KKT_ALLOC_GUARD: number of released blocks whose poisoned front had been overwritten (an out-of-bounds write) */
int mi355kkt_test_guard_violations(void);

#ifdef __cplusplus
}
#endif
#endif /* MI355KKT_TEST_H */

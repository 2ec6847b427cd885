/*
 * votenet_hip_debug.h -- measurement, A/B and tuning switches of libvotenet_hip.so.
 *
 * NOT part of the drop-in surface (include/votenet_hip.h).  Every switch here is PROCESS-GLOBAL state that changes which kernel, grid
 * or code path later launches take -- never their results beyond summation order, each alternative is parity-tested -- and exists for
 * the repo's own tests (tests/test_gpu_bf3.py, test_gpu_split_k.py, test_gpu_parity.py), profiles (tools/, tools/probe/) and A/B runs.
 *
 * Gate: the setters below do NOTHING (and leave "... ignored: debug switches are disabled" in votenet_last_error()) until the host has
 * called votenet_debug_enable(1), or VOTENET_DEBUG=1 was in the environment when the first setter was called.  A consumer that links the
 * library for the reference's launchers never opts in, and for it a launch depends on its arguments only.  The Python host opts in the
 * first time a test or tool touches a switch (votenet_amd/_lib.py).  Switches are not synchronised: set them from one thread, with no
 * launch in flight on another.
 */
#ifndef VOTENET_HIP_DEBUG_H
#define VOTENET_HIP_DEBUG_H

#ifdef __cplusplus
extern "C" {
#endif

void votenet_debug_enable(int on); /* 1: the setters below take effect from now on; 0: inert again (values already set stay) */
int votenet_debug_enabled(void);   /* 1 after votenet_debug_enable(1) / an accepted VOTENET_DEBUG=1 */

/* ---- farthest-point sampling (fps.hip) ---- */
/* 4096 < n <= 24576 through the kernel that emits up to two samples per round (same indices, same order; DESIGN_HISTORY.md 4.1).
 * Off by default: measured slower than the one-sample rounds. */
void votenet_fps_debug_two_pick(int on);
/* 0 disables the parallel "already in farthest-point order?" check that precedes the sampling rounds for n <= 2048
 * (DESIGN_HISTORY.md 4.1); the result is the same either way. */
void votenet_fps_debug_prefix_check(int on);
void votenet_fps_debug_config(int nw, int p);    /* force a brute-force configuration: nw waves x p points per lane (0, 0 = automatic) */
void votenet_debug_fps_split(int on);            /* 24 576 < n <= 98 304: one scene over 4 (on = 1), 12 (3) or 6 (5) workgroups; same indices, measured slower: 0 by default */
unsigned votenet_debug_fps_split_timeouts(void); /* (a read, not gated) polls of the split kernel that gave up: 0 unless a part of a scene never ran */
void votenet_debug_fps_lds_floor(int bytes);     /* dynamic LDS floor of the bucket kernels (occupancy experiments; 0 = none) */

/* ---- ball query (grouping.hip) ---- */
/* which kernel serves n <= 2048 (0 = by cloud size: 16 waves, one super-chunk; 4 = four waves x eight groups; 16 = sixteen waves x
 * eight groups).  Same indices and counts in every form. */
void votenet_debug_ball_query_small(int form);

/* ---- grouped MLP forward / dgrad (mlp_fast.hip) ---- */
/* split-K (votenet_mlp_split_k_*): target workgroups, maximum parts, minimum slabs per part; launches of at least max_wgs output
 * tiles are left alone; 0 keeps a value */
void votenet_debug_split_k(int target_wgs, int max_parts, int min_slabs, int max_wgs);
/* 0: every GEMM on the fp32 MFMA kernels whatever is registered (A/B and parity tests); 1 (default): images are used; another
 * value: a mask over GEMM families (mlp_fast.hip, bf3_family) */
void votenet_debug_fast_bf3(int on);
/* 0: two-piece (fp16 x 2) weight images are ignored, those GEMMs run on the fp32 MFMA kernel; 1 (default): used */
void votenet_debug_fast_h2(int on);
void votenet_debug_fast_dyn_lds(int bytes);  /* extra dynamic LDS per workgroup of the fast GEMMs (occupancy experiments) */
void votenet_debug_fast_xcd_chunk(int on);   /* 1 (default): the piece-layout GEMMs that gather the per-point table take their row tiles in
                                                per-XCD contiguous chunks (an XCD's L2 then holds the scenes its tiles touch); 0: round-robin */
void votenet_debug_fast_workgroups(int cap22, int cap41); /* persistent-workgroup caps of the 2x2 / 4x1 wave layouts (0 keeps a value) */
void votenet_debug_assemble_stats(int cap, int u);        /* assemble_stats workgroups per column block (default 128), points in flight per thread (4 or 8, default 8) */

/* ---- weight gradients (mlp_wgrad_fast.hip, mlp_bwd.hip) ---- */
/* every weight-gradient GEMM (votenet_mlp_wgrad / _wgrad_bn, assembled, narrow) on split operands: row-major bf16 images in LDS,
 * fragments through ds_read_b64_tr_b16; 0: the fp32 MFMA kernel.  Default 1. */
void votenet_debug_wgrad_bf3(int on);
void votenet_debug_wgrad_workgroups(int n);  /* workgroups of the split-operand weight-gradient kernel (0 = default) */
void votenet_debug_bn_reduce_passes(int n);  /* row passes per workgroup of the dense BatchNorm-backward reduction (default 16) */

/* ---- pooled layers' backward (pool_bwd.hip) ---- */
/* votenet_mlp_gram on split operands (8 consecutive rows of a channel per MFMA fragment; c = 64 or 128, no scratch = atomics mode):
 * 1 (default) as fp16 x 2 pieces (both operands are activations), 3 as bf16 x 3 pieces; 0: the fp32 MFMA kernel always. */
void votenet_debug_gram_bf3(int on);
void votenet_debug_gram_workgroups(int n);            /* workgroups of the split-operand Gram kernel (default 384) */
void votenet_debug_zsel_grid(int groups_per_wg, int cap); /* grid of the pooled BatchNorm-backward reduction (default 32 groups per workgroup, at most 256 workgroups) */
void votenet_debug_scatter_reverse(int on);           /* votenet_pool_dgrad_scatter walks its groups back to front (DESIGN_HISTORY.md 4.3).  Default 0. */
void votenet_debug_scatter_waves(int n);              /* 12 or 16 (default) wavefronts per scatter workgroup */
void votenet_debug_scatter_workgroups(int n);         /* workgroups of votenet_pool_dgrad_scatter (0 = default) */
void votenet_debug_scatter_form(int form);            /* 1 (default): one wavefront per group, no barriers; 0: one workgroup per group */
void votenet_debug_sparse_workgroups(int n);          /* workgroups of votenet_pool_wgrad_sparse (default 384) */
void votenet_debug_sparse_teams(int teams, int wgs);  /* 1 or 2 (default) teams per workgroup on the piece layout; workgroups of the 2-team form (default 256) */
void votenet_debug_sparse_centre_workgroups(int n);   /* workgroups of votenet_pool_wgrad_sparse_half_centres (default 192) */
void votenet_debug_sparse_centre_teams(int t);        /* 1 or 2 (default) teams per workgroup of the centre-walking form */

#ifdef __cplusplus
}
#endif
#endif /* VOTENET_HIP_DEBUG_H */

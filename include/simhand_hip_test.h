/*
 * simhand_hip_test.h -- the INSTRUMENTS of libsimhand_hip.so: route counters, the HIP-event profiler and the simhand_test_* hooks.
 *
 * Not part of the drop-in surface (include/simhand_hip.h declares that): a product caller never includes this header.  The three
 * facilities are process-global by design -- relaxed atomics / a mutex-guarded event list -- and exist for the test suite (prove which
 * hand-written kernel a parity run exercised; force a route at sizes the dispatch would not pick it for), for bench.py (per-class kernel
 * time from HIP events on the launch stream) and for same-box A/B timing.  Every hook selects between kernels that compute the same
 * result; their defaults are the measured-best routes and simhand_test_hooks_reset() restores them all.  The library reads NO environment
 * variable.  Symbols declared here are exported by the same shared object; SH_ABI_VERSION (simhand_hip.h) covers them too.
 */
#ifndef SIMHAND_HIP_TEST_H
#define SIMHAND_HIP_TEST_H

#include "simhand_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- kernel-route counters: which hand-written kernel a call was dispatched to.  One atomic counter per route,
 * bumped at launch time.  Tests use them to PROVE that a parity run exercised a given kernel (e.g. that the bf16
 * ResNet-50 step really went through the 256x256 LDS-DMA tile kernel, the activation-stationary 1x1 kernel, the
 * register-resident 3x3 kernel, the all-taps weight gradient, the BatchNorm folds and the two-segment data gradient). */
enum sh_route {
  SH_ROUTE_IGEMM128_FWD = 0, SH_ROUTE_IGEMM128_DGRAD = 1,   /* 128 x {64,128} register-staged tile kernel */
  SH_ROUTE_IGEMM256_FWD = 2, SH_ROUTE_IGEMM256_DGRAD = 3,   /* 256 x 256 LDS-DMA tile kernel */
  SH_ROUTE_IGEMM256_TAIL = 4,                               /* ragged last round handed to the 128-row kernel */
  SH_ROUTE_GEMM1X1_FWD = 5, SH_ROUTE_GEMM1X1_FWD_BNACT = 6, SH_ROUTE_GEMM1X1_DGRAD = 7,  /* activation-stationary 1x1 */
  SH_ROUTE_C64_FWD = 8, SH_ROUTE_C64_DGRAD = 9,             /* 64->64 3x3, filter resident in registers */
  SH_ROUTE_STEM_FWD = 10,
  SH_ROUTE_FWD_BNACT = 11,                                  /* any forward with the BN + residual + ReLU epilogue */
  SH_ROUTE_DGRAD_CONCAT = 12,                               /* data gradient with a second K segment */
  SH_ROUTE_DGRAD_FUSED_SUMS = 13,                           /* data gradient emitting BN-backward sums / masked store */
  SH_ROUTE_DGRAD_PARITY = 14,                               /* stride-2 data gradient as 4 parity classes */
  SH_ROUTE_WGRAD3X3 = 15, SH_ROUTE_WGRAD_PLAIN = 16, SH_ROUTE_WGRAD_GENERIC = 17, SH_ROUTE_WGRAD_STEM = 18,
  SH_ROUTE_WGRAD_COLSUM = 19,                               /* 1x1 weight gradient that also emits sum(dy) (Gram launches) */
  SH_ROUTE_BN_FOLD_FWD = 20, SH_ROUTE_BN_FOLD_BWD = 21,     /* Gram-matrix BatchNorm fold (parameter-sized algebra) */
  SH_ROUTE_BN_APPLY = 22, SH_ROUTE_BN_BWD_APPLY = 23,
  SH_ROUTE_STEM_BN_POOL = 24,                               /* fused BN + ReLU + MaxPool (fwd or bwd) */
  SH_ROUTE_NTXENT_FWD = 25, SH_ROUTE_NTXENT_BWD = 26,
  SH_ROUTE_FP8_FWD = 27, SH_ROUTE_FP8_DGRAD = 28,           /* e4m3 MFMA (K = 128 per instruction) tile kernel */
  SH_ROUTE_BN_APPLY_GRAM = 29,                              /* BN-apply + ReLU fused into the Gram (x^T x) launch */
  SH_ROUTE_WGRAD_BNBWD = 30,                                /* BN-backward apply fused into the 1x1 weight gradient's dy loader */
  SH_ROUTE_NTXENT_FUSED_DIST = 31,                          /* loss tile kernel computing the joint distances in-tile (no D block) */
  SH_ROUTE_DGRAD_DYSRC = 32,                                /* BN-backward apply fused into the 1x1 data gradient's dy loader */
  SH_ROUTE_FWD_CHAIN = 33,                                  /* conv3 + BN + residual + ReLU with the next block's conv1 chained on */
  SH_ROUTE_R128_FWD = 34, SH_ROUTE_R128_DGRAD = 35,         /* 128->128 3x3: activation tile staged once in an LDS ring, weights streamed per tap */
  SH_ROUTE_FWD_BNIN = 36,                                   /* 3x3 forward with the previous unit's BatchNorm + ReLU applied in its LDS ring */
  SH_ROUTE_N128_FWD = 37, SH_ROUTE_N128_DGRAD = 38,         /* 1x1 with 128 destination channels behind a long reduction: 128 x 128 LDS-DMA tiles, two blocks per CU */
  SH_ROUTE_FP8_WGRAD = 39,                                  /* e4m3 3x3 weight gradient (reduction over pixels on the scaled K = 128 MFMA) */
  SH_ROUTE_STEM_RING_FWD = 40, SH_ROUTE_STEM_RING_WGRAD = 41,  /* the stem's LDS-ring kernels (forward: 128^2 / 224^2 / 256^2 inputs; weight gradient: 224^2) */
  SH_ROUTE_COUNT = 42
};
int simhand_route_counts(int64_t* out /*[SH_ROUTE_COUNT]*/);
int simhand_route_reset(void);
/* every test / tuning hook back to its default */
int simhand_test_hooks_reset(void);
/* Kernel-selection switches that round 3 read from SIMHAND_* environment variables inside the library, now ordinary test hooks: the
 * library itself reads NO environment variable.  simhand_test_switch(which, value): value < 0 restores the built-in default;
 * simhand_test_hooks_reset() restores all of them.  Every switch selects between kernels that compute the same result. */
enum sh_test_switch {
  SH_SW_BN_GRID_APPLY = 0, /* block cap of bn_apply's grid (default 131072) */
  SH_SW_BN_GRID_BWD = 1,   /* block cap of bn_bwd_apply's grid (default 131072) */
  SH_SW_R128 = 2,          /* conv3x3_r128_kernel for the 128-channel 3x3 layers (default 1) */
  SH_SW_G1_PF = 3,         /* branch-free fast variants of gemm1x1_kernel per K: bit 0 K = 64, bit 1 K = 128, bit 2 K = 256 (default 7) */
  SH_SW_G1_CHAIN = 4,      /* chained next conv1 per K: bit 0 K = 64, bit 1 K = 128 (default 1); simhand_test_conv1x1_chain_mask overrides */
  SH_SW_G1_LT = 5,         /* linear epilogue stores of gemm1x1_kernel: bit 0 forward, bit 1 data gradient (default 3; round 6: the data gradient too, -0.2 ms in its class) */
  SH_SW_FUSE_S2 = 6,       /* BN-backward sums fused into stride-2 3x3 data gradients: 1 all, 2 only on the 256-wide kernel (default 0) */
  SH_SW_WG_DMA = 7,        /* wgrad1x1_dma_kernel (default 1) */
  SH_SW_WG3_S2 = 8,        /* stride-2 form of wgrad3x3_kernel (default 1) */
  SH_SW_WG_BIG = 9,        /* 256 x 128 tiles of the plain 1x1 weight gradient (default 1) */
  SH_SW_WG_WIDE = 10,      /* one tile across the wide side of the 64 <-> 256 weight gradients (default 1) */
  SH_SW_STEM_WG256 = 11,   /* 64 x 256 tile of the stem weight gradient (default 1) */
  SH_SW_STEM_RING = 12,    /* stem_ring_fwd_kernel at 224 x 224 (default 1) */
  SH_SW_STEM_RING_LT = 13, /* its linear stores (default 1) */
  SH_SW_STEM_WG_RING = 14, /* stem weight gradient with both operands in LDS rings (stem_wgrad_ring_kernel) at 224 x 224 (default 1) */
  SH_SW_N128 = 15,         /* gemm_n128_kernel for the 1x1 layers with 128 destination channels and >= 256 of reduction (default 1) */
  SH_SW_FOLD_LEGACY = 16,  /* 1 = the round-5 launch chains of the folded BatchNorm algebra and of the Gram / colsum reductions (separate centre,
                            * slice-sum, bias, channel-sum and coefficient launches) instead of the merged ones of round 6 -- same numbers bit for bit
                            * (default 0) */
  SH_SW_COUNT = 17
};
int simhand_test_switch(int which, int value);

/* ---- optional per-kernel-class HIP-event profiler (used by bench.py) ----- */
enum sh_prof_class { SH_PROF_CONV_FWD = 0, SH_PROF_CONV_DGRAD = 1, SH_PROF_CONV_WGRAD = 2,
                     SH_PROF_BN = 3, SH_PROF_POOL = 4, SH_PROF_LOSS = 5, SH_PROF_MISC = 6,
                     SH_PROF_OPT = 7 /* LARS + Adam update */, SH_PROF_NCLASS = 8 };
int simhand_prof_enable(int on);
/* bit c set = launches of class c record their event pair (default: all).  An event pair costs ~1-3 us of queue time per
 * launch, so a timed region records only the class it needs. */
int simhand_prof_set_classes(uint32_t mask);
/* blocks until recorded events completed; out_ms/out_flops/out_bytes/out_count are host arrays of SH_PROF_NCLASS */
int simhand_prof_collect(double* out_ms, double* out_flops, double* out_bytes, int64_t* out_count);
/* the individual records since the last collect / reset, in issue order (class, elapsed ms, algorithmic FLOPs and bytes of each launch);
 * does not clear them.  Diagnostic: scripts/launch_outliers.py lists the launches furthest above the time their own work allows. */
int simhand_prof_records(int max_records, int* cls, double* ms, double* flops, double* bytes, int* n_out);
int simhand_prof_reset(void);

/* 64 -> 64 channel 3x3 / stride 1 / pad 1 bf16 layers (forward and store-only data gradient) run on the padded pixel
 * grid with the whole filter resident in registers (conv3x3_c64.hip); 0 routes them through the generic tile kernels
 * (tuning / test hook). */
int simhand_test_conv3x3_c64_enable(int on);

/* tuning hook: the 128 -> 128 channel 3x3 ring kernel (conv3x3_ring.hip): 1 on, 0 off (the 128 x 128 tile kernel), -1 back to the default */
int simhand_test_conv3x3_r128_enable(int on);

/* tuning / test hook of the short-K (cin or cout in {64,128,256}) bf16 stride-1 1x1 kernel: rows per block = 64*mf */
int simhand_test_conv1x1_set_rows(int k, int mf);

/* bf16 route of simhand_stem_conv_fwd: 1 (default) = persistent direct-stem kernel (weights resident in LDS, next tile's rows in
 * flight under the current tile's MFMAs), 2 = activation-stationary kernel, one block per 256 rows, 0 = 128 x 64 tile kernel
 * (same k order, bit-identical outputs; tuning / test hook) */
int simhand_test_stem_conv_route(int mode);

int simhand_test_conv2d_dgrad_fuse_1x1(int on);

/* test / tuning hook: which input widths chain (bit 0: 64, bit 1: 128; -1 = default = 64 only: the 128-wide form measured no faster) */
int simhand_test_conv1x1_chain_mask(int mask);

/* tuning hook: route the eligible bf16 layers (>= 256 destination channels, long reduction) to the 256x256 LDS-DMA
 * tile kernel (1 = default); the BN partial-sum block counts above follow the setting */
int simhand_test_igemm256_enable(int on);

/* 1 (default): a 256 x 256 launch whose last round of tiles would leave more than two thirds of the CUs idle hands those
 * m-tiles to a second launch of the 128-row kernel (same results bit for bit); 0 = single launch (tuning / test hook) */
int simhand_test_igemm256_split_tail(int on);

int simhand_test_igemm256_tile224(int mode);

/* tuning hook: non-temporal (streaming) loads / stores in the BatchNorm passes (1 = on [default]) */
int simhand_test_bn_set_nt(int on);

/* tuning hook: the all-taps 3x3 / stride-1 weight-gradient kernel (bf16; 1 = default, 0 = tap-by-tap kernel) */
int simhand_test_wgrad3x3_enable(int on);

/* test / tuning hook: the LDS-DMA 256 x 256 tile kernel for the bf16 1x1 weight gradients with >= 256 channels on both sides (1 = default) */
int simhand_test_wgrad_dma_enable(int on);

/* test hook: bf16 wgrad LDS transpose path (1 = ds_read_b64_tr_b16 [default], 0 = scalar LDS reads) */
int simhand_test_wgrad_set_tr(int on);

/* tuning hook: the bf16 1x1 / stride-1 weight gradient reduces 32 * kpm pixels per barrier (kpm 1 or 2, default 2) */
int simhand_test_wgrad_plain_kpm(int kpm);

/* tuning hook: blocks (tiles x split-K) a weight-gradient launch aims for: generic / 1x1 kernel, all-taps 3x3 kernel
 * (< 64 restores the default) */
int simhand_test_wgrad_target_blocks(int n, int n3x3);

#ifdef __cplusplus
}
#endif

#endif /* SIMHAND_HIP_TEST_H */

// emdr2_amd/csrc/exp_hooks.h -- the ONE switch between the product build and the experiments build of libemdr2_hip.
//
// Timing experiments (ablations, tile-order / schedule knobs read from the environment) are not product code: they live in
// tools/exp/exp_hooks.inc and are compiled only by `make exp` (-DEMDR2_EXPERIMENTS -> lib/libemdr2_hip_exp.so, loaded by tools/ only).  The
// product sources name the places an experiment may hook into with the macros below, which are EMPTY here: the product library contains
// no getenv, no ablation branch and no alternative schedule.
#pragma once
#ifdef EMDR2_EXPERIMENTS
#include "../../tools/exp/exp_hooks.inc"
#else
// ---- gemm.hip -----------------------------------------------------------------------------------------------------------------------
#define EXP_GEMM_KLOOP_LEN(p, nch) (nch)                    /* chunks the k-loop runs */
#define EXP_GEMM_AFTER_KLOOP(p, acc)                        /* may leave the kernel before the epilogue */
#define EXP_GEMM_STORE_IF(p, w)                             /* guards the epilogue's global stores */
#define EXP_GEMM_TILE_ORDER(order_env, ablate_env, l2_env) constexpr int order_env = 1, ablate_env = 0, l2_env = 2560;
#define EXP_GEMM_USE_GEMM8() true
#define EXP_GEMM_TILE_VARIANTS(p, batch, split_k, stream)
// ---- gemm8.hip ----------------------------------------------------------------------------------------------------------------------
#define EXP_G8_AFTER_KLOOP(p, acc, wr)
#define EXP_G8_STORE_IF(p, w)
#define EXP_G8_HOST_L2(single_kb)
#define EXP_G8_HOST_GROUPS(p, ng)
#define EXP_G8_HOST_LAUNCH(p)
// ---- gemm8t.hip / gemm_tn.hip -------------------------------------------------------------------------------------------------------
#define EXP_T8_STREAM(p, zs, KT, zs_addr, KT_STREAM, T8_DMA_ON) const int zs_addr = zs; const int KT_STREAM = KT; constexpr bool T8_DMA_ON = true;
#define EXP_T8_HOST(p)
#define EXP_TN_GENERAL_ONLY(general_only)
// ---- mips_api.hip / mips_scan.hip / mips_scan8.hip ------------------------------------------------------------------------------------
#define EXP_ENV_INT(name, dflt) (dflt)
#define EXP_MIPS_PACK_QUERIES_FRAG(variant, scan_kernel, qp, nqp, dim, w, stream, rc)
#define EXP_MIPS_COUPLE() true
#define EXP_MIPS_SCAN_VARIANT(mode, variant, ablate, scan_kernel, sp, w, grid, seg_end, n_rows, done, stream, rc) if (false) {} else
#define EXP_SCAN8_TAU(p, tauv, qt)
#define EXP_SCAN8_LATE_START(P, hq)
#endif

// Experiment / route switches (SAPCA_* environment variables).
//
// A RELEASE build of libsapca reads none of them: dbg_env() is a constant null there and every switch below is compiled
// out, so a stray variable in a caller's process cannot change a route or a data layout.  The variant built with
// -DSAPCA_DEBUG_SWITCHES (lib/libsapca_dbg.so, `make debug`) reads them; the route tests (tests/, fixture `debug_switches`)
// and the A/B scripts under tools/ load that variant through SAPCA_LIB_PATH.
//
// What a release build does read, as plain getenv calls at their two sites:
//   SAPCA_AT_OVERLAP=0|1    engine.cpp   0: the A^T sweep of a multi-rank fit in one piece (no collective on a side stream);
//                                        1: in two pieces under RCCL too (opt-in there until it has run on more than one GPU)
//   SAPCA_MULTI_INPROCESS=1 multi.cpp    sapca_multi_* members talk through page-locked host memory instead of RCCL
//
// The switches of the debug variant (name: effect), by file:
//   engine.cpp   SAPCA_TILED_MIN_ENTRIES (floor of the staged sweep), SAPCA_TILED_FROM_A, SAPCA_AT_NATURAL, SAPCA_AT_UNPACK,
//                SAPCA_AT_SORT, SAPCA_PREPARE_SERIAL, SAPCA_PREPARE_ASIDE_FIRST, SAPCA_LANCZOS_TRANSPOSE, SAPCA_MASK_STATS_INLINE,
//                SAPCA_MASK_SUMS_SCATTER, SAPCA_MASK_TRANSPOSE_FIRST, SAPCA_SMALL_SVD_QR, SAPCA_Q3_ROWKERNEL
//   spmm_tiled.hip  SAPCA_DQ_BLOCK_ROWS, SAPCA_SPLIT_WGS, SAPCA_TILED_FMT, SAPCA_TILED_SLOTS, SAPCA_TILED_MODE, SAPCA_TILED_GEOM128,
//                SAPCA_TILE_DEFAULT, SAPCA_QF_CAP_FIXED, SAPCA_NO_ROWSORT, SAPCA_ROWSORT_ALWAYS, SAPCA_FILL_DIRECT, SAPCA_AT_BUCKETS, SAPCA_AT_SORT,
//                SAPCA_RUNS_SEG_LDS_MAX, SAPCA_SWEEP_STAGED, SAPCA_NO_DQ, SAPCA_DEBUG
//   spmm_dq.hip  SAPCA_NO_DQ, SAPCA_NO_DQ_F64      spmm.hip  SAPCA_ROWGATHER_PER_ENTRY      prep.hip  SAPCA_TRANSPOSE_GATHER
//   dense.hip    SAPCA_CHOL_GENERAL, SAPCA_EIG_DEVICE
//   lanczos.hip  SAPCA_SPMV_NO_LDS, SAPCA_SPMV_NO_SLICE_GRID, SAPCA_SPMV_IDX32, SAPCA_LANCZOS_CHECK
//   api.cpp      SAPCA_UPLOAD_NARROW_ON_DEVICE, SAPCA_UPLOAD_STATS_OFF
//   comm.cpp     SAPCA_COMM_FORCE_RCCL (a one-rank RCCL communicator: how a one-GPU box tests the binding), SAPCA_COMM_NO_SPLIT,
//                SAPCA_RCCL_LIBRARY (path of the tests' stand-in for librccl: tests/fake_rccl)
#pragma once
#include <cstdlib>

namespace sapca {

#ifdef SAPCA_DEBUG_SWITCHES
inline const char* dbg_env(const char* name) { return std::getenv(name); }
constexpr bool kDebugSwitches = true;
#else
inline const char* dbg_env(const char*) { return nullptr; }
constexpr bool kDebugSwitches = false;
#endif
inline bool dbg_on(const char* name) { return dbg_env(name) != nullptr; }

}  // namespace sapca

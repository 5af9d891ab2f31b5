// DPP-fed quad sweep (spmm_dq.hip): tables on top of the quad format of spmm_tiled.hip, and its launcher.
#pragma once
#include "common.h"

namespace sapca {
namespace k {

// Steps of a quad in a tile = its longest row segment, rounded up to an even count unless the build lets quads end on any
// step (-DSAPCA_ODD_STEPS: the DPP-fed sweep's main loop is then generated with DQ2_ODD=1 and counts per step).
#ifdef SAPCA_ODD_STEPS
constexpr bool kOddSteps = true;
#else
constexpr bool kOddSteps = false;
#endif

// Per (row block, wave, tile) stream offsets / chunk counts and the per-two-step row-slot descriptors of a built
// quad-format operator.  Returns false (op.dq stays false) when the operator is not eligible: the callers then use
// the staged-entry quad sweep.
bool dq_build_tables(TiledOp& op, TiledBuffers& buf, hipStream_t s);
bool dq_usable(const TiledOp& op, int ldx);
// flags: 8 = pattern mode (every stored non-zero value reads as 1)
void launch_dq(const TiledOp& op, const float* X, int ldx, float* out, int ldo, int ncols, const float* cvec, hipStream_t s, int flags = 0);
// the f64 operator (elem == 8): one 64-column pass
void launch_dq_f64(const TiledOp& op, const double* X, int ldx, double* out, int ldo, int ncols, const double* cvec, hipStream_t s);
// the same over row blocks [rb0, rb1) with a split of the tile range chosen by the caller (`out` then holds nsplit slabs of
// op.rows x ldo when nsplit > 1, slab sp at out + sp * op.rows * ldo: rows keep their absolute positions)
void launch_dq_blocks(const TiledOp& op, int rb0, int rb1, int nsplit, int tiles_per_split, const float* X, int ldx, float* out, int ldo,
                      int ncols, const float* cvec, hipStream_t s, int flags = 0);
// out (m x k, leading dimension k) = sum over the stored entries of row i of (a_ij - mu_j) W[j][:k]   (quirk Q3) as two sweeps of
// the operator's format plus a pass over the values for stored zeros.  W2 (cols x ldw) and tmp (m x k) are scratch.  Returns false
// when the operator cannot take the DPP-fed sweep (nothing is launched).
bool q3_projection_dq(const CsrView<float>& A, const TiledOp& op, const float* W, int ldw, const float* mu, float* W2, float* tmp,
                      float* out, int k, hipStream_t s);

}  // namespace k
}  // namespace sapca

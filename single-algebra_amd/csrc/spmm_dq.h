// DPP-fed quad sweep (spmm_dq.hip): tables on top of the quad format of spmm_tiled.hip, and its launcher.
#pragma once
#include "common.h"

namespace sapca {
namespace k {

// Steps of a quad in a tile = its longest row segment: a quad may end on any step (round 4; the generated main loop counts
// per step, tools/gen_spmm_dq2.py DQ2_ODD=1).  -DSAPCA_EVEN_STEPS brings back the format of rounds 2-3, which rounded a quad's
// steps up to an even count (3.8 % more executed slots at C2, 10 % more at C5's 3.2 entries per row and tile) for a counter
// per two-step group; the main loop must then be generated with DQ2_ODD=0.  Measured same-box (profiles/r04_ab_odd_*.txt):
// C2 sweep -1.4 %, C5 sweep -3.4 % and preparation -5 %, C4 -0.1 %.
#ifdef SAPCA_EVEN_STEPS
constexpr bool kOddSteps = false;
#else
constexpr bool kOddSteps = true;
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

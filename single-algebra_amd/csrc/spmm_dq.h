// DPP-fed quad sweep (spmm_dq.hip): tables on top of the quad format of spmm_tiled.hip, and its launcher.
#pragma once
#include "common.h"

namespace sapca {
namespace k {

// Per (row block, wave, tile) stream offsets / chunk counts and the per-two-step row-slot descriptors of a built
// quad-format operator.  Returns false (op.dq stays false) when the operator is not eligible: the callers then use
// the staged-entry quad sweep.
bool dq_build_tables(TiledOp& op, TiledBuffers& buf, hipStream_t s);
bool dq_usable(const TiledOp& op, int ldx);
void launch_dq(const TiledOp& op, const float* X, int ldx, float* out, int ldo, int ncols, const float* cvec, hipStream_t s);

}  // namespace k
}  // namespace sapca

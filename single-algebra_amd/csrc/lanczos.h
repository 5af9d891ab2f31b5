// R12: Lanczos SVD of the prepared (raw, uncentred) operator -- svd_las2 call sites
// /root/reference/src/dimred/pca/sparse/mod.rs:134-144, sparse_masked/mod.rs:316-331.
#pragma once
#include "engine.h"

namespace sapca {
// Fills h.sing (k values) and h.components_dev (k x n_used, sign-fixed) from h.a_used / h.at_used.
template <typename T>
void lanczos_fit(sapca_handle_s& h);
}  // namespace sapca

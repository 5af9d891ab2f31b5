// Host-side f64 factorisations of the l x l cores (l <= 128): nothing here touches the m- or
// n-sized data, which stays on the GPU.
#pragma once
#include <vector>

namespace sapca {

// One-sided Jacobi (Hestenes) SVD of the row-major l x l matrix A = U diag(s) V^T.
// Returns s sorted descending and the matching columns of U (row-major l x l).
void jacobi_svd(const std::vector<double>& A, int l, std::vector<double>& U, std::vector<double>& s);

// Symmetric tridiagonal eigen-decomposition (implicit QL): d diagonal, e sub-diagonal (e[0] unused
// convention: e[i] couples i-1 and i).  On return d holds eigenvalues (ascending) and Z (row-major
// n x n) the eigenvectors in columns.  Returns false if it failed to converge.
bool tridiag_eigh(std::vector<double>& d, std::vector<double>& e, int n, std::vector<double>& Z, bool last_row_only = false);

// Eigen-decomposition of the symmetric row-major n x n matrix A (Householder tridiagonalisation + implicit
// QL, eigenvectors accumulated in rows so that every rotation runs over contiguous memory).  On return w
// holds the eigenvalues in DESCENDING order and row i of Vt the unit eigenvector of w[i].  Returns false
// if QL failed to converge.
bool sym_eigh_desc(const std::vector<double>& A, int n, std::vector<double>& w, std::vector<double>& Vt);

}  // namespace sapca

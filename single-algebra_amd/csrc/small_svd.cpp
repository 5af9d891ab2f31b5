#include "small_svd.h"

#include <algorithm>
#include <cmath>
#include <numeric>

namespace sapca {

// One-sided Jacobi on the columns of W = A (column-major for stride-1 rotations): A V = W, W -> U diag(s).
// The squared column norms are carried along and refreshed once per sweep, so a pair costs one dot
// product and one rotation.  Cloned for AVX2+FMA (resolved at load time): this runs on the host between
// two device phases of every randomized fit.
#if defined(__x86_64__) && defined(__GNUC__) && !defined(__HIP_DEVICE_COMPILE__)
__attribute__((target_clones("arch=haswell", "default")))
#endif
static void jacobi_rotate_sweeps(double* W, double* nrm, int l) {
  for (int sweep = 0; sweep < 80; ++sweep) {
    for (int j = 0; j < l; ++j) {
      const double* w = W + (size_t)j * l;
      double a = 0;
      for (int i = 0; i < l; ++i) a += w[i] * w[i];
      nrm[j] = a;
    }
    double off = 0;
    for (int p = 0; p + 1 < l; ++p) {
      double* wp = W + (size_t)p * l;
      for (int q = p + 1; q < l; ++q) {
        double* wq = W + (size_t)q * l;
        const double a = nrm[p], b = nrm[q];
        if (a == 0 || b == 0) continue;
        double g = 0;
        for (int i = 0; i < l; ++i) g += wp[i] * wq[i];
        const double r = std::fabs(g) / std::sqrt(a * b);
        off = std::max(off, r);
        if (r < 1e-14) continue;   // columns orthogonal to 1e-14: singular vectors good to ~1e-13, far below the 1e-9 rad the f64 tests ask for
        const double zeta = (b - a) / (2.0 * g);
        const double t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
        const double cs = 1.0 / std::sqrt(1.0 + t * t), sn = cs * t;
        for (int i = 0; i < l; ++i) {
          const double x = wp[i], y = wq[i];
          wp[i] = cs * x - sn * y;
          wq[i] = sn * x + cs * y;
        }
        nrm[p] = std::max(0.0, a - t * g);
        nrm[q] = b + t * g;
      }
    }
    if (off < 1e-13) break;
  }
}

void jacobi_svd(const std::vector<double>& A, int l, std::vector<double>& U, std::vector<double>& s) {
  std::vector<double> W((size_t)l * l), nrm((size_t)std::max(l, 1));
  for (int i = 0; i < l; ++i)
    for (int j = 0; j < l; ++j) W[(size_t)j * l + i] = A[(size_t)i * l + j];
  jacobi_rotate_sweeps(W.data(), nrm.data(), l);
  std::vector<double> norm(l);
  for (int j = 0; j < l; ++j) {
    double a = 0;
    for (int i = 0; i < l; ++i) a += W[(size_t)j * l + i] * W[(size_t)j * l + i];
    norm[j] = std::sqrt(a);
  }
  std::vector<int> order(l);
  std::iota(order.begin(), order.end(), 0);
  std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return norm[x] > norm[y]; });
  U.assign((size_t)l * l, 0.0);
  s.assign(l, 0.0);
  for (int c = 0; c < l; ++c) {
    const int j = order[c];
    s[c] = norm[j];
    if (norm[j] > 0)
      for (int i = 0; i < l; ++i) U[(size_t)i * l + c] = W[(size_t)j * l + i] / norm[j];
  }
}

// ---- symmetric eigenproblem of the l x l Gram (f32 randomized path) ----------------------------------
// Householder reduction to tridiagonal form (EISPACK tred2 structure): on return `a` holds the orthogonal
// Q with A = Q T Q^T (row-major), d the diagonal and e the sub-diagonal of T (e[i] couples i-1 and i).
#if defined(__x86_64__) && defined(__GNUC__) && !defined(__HIP_DEVICE_COMPILE__)
__attribute__((target_clones("arch=haswell", "default")))
#endif
static void householder_tridiag(double* a, int n, double* d, double* e, double* work) {
  // The active block a[0..l][0..l] is kept SYMMETRIC in both triangles, so that A u and the rank-2 update walk rows
  // (contiguous, vectorised) instead of columns; tred2 proper works on the lower triangle alone.  work: 2 n doubles
  // (no containers in here: the function is compiled once per target, and library inlines do not cross that line).
  double* u = work;
  double* q = work + n;
  for (int i = n - 1; i >= 1; --i) {
    const int l = i - 1;
    double hh = 0, scale = 0;
    double* ai = a + (size_t)i * n;
    if (l > 0) {
      for (int k = 0; k <= l; ++k) scale += std::fabs(ai[k]);
      if (scale == 0.0) {
        e[i] = ai[l];
      } else {
        for (int k = 0; k <= l; ++k) {
          ai[k] /= scale;
          hh += ai[k] * ai[k];
        }
        double f = ai[l];
        double g = f >= 0 ? -std::sqrt(hh) : std::sqrt(hh);
        e[i] = scale * g;
        hh -= f * g;
        ai[l] = f - g;
        for (int k = 0; k <= l; ++k) u[k] = ai[k];
        // p = A u / hh (rows of the symmetric block), f = u . p
        f = 0;
        for (int j = 0; j <= l; ++j) {
          const double* aj = a + (size_t)j * n;
          double acc = 0;
          for (int k = 0; k <= l; ++k) acc += aj[k] * u[k];
          e[j] = acc / hh;
          f += e[j] * u[j];
        }
        const double hk = f / (hh + hh);
        for (int j = 0; j <= l; ++j) q[j] = e[j] - hk * u[j];
        // A <- A - u q^T - q u^T on the whole block; column i keeps u / hh for the accumulation of Q below
        for (int j = 0; j <= l; ++j) {
          double* aj = a + (size_t)j * n;
          const double uj = u[j], qj = q[j];
          for (int k = 0; k <= l; ++k) aj[k] -= uj * q[k] + qj * u[k];
          aj[i] = uj / hh;
          e[j] = qj;
        }
      }
    } else {
      e[i] = ai[l];
    }
    d[i] = hh;
  }
  d[0] = 0;
  e[0] = 0;
  for (int i = 0; i < n; ++i) {
    const int l = i - 1;
    double* ai = a + (size_t)i * n;
    if (d[i] != 0.0) {
      // Q(0..l, 0..l) <- Q - (u / hh) (u^T Q): g = u^T Q as a sum of rows, then one row update each
      double* g = q;
      for (int j = 0; j <= l; ++j) g[j] = 0;
      for (int k = 0; k <= l; ++k) {
        const double* ak = a + (size_t)k * n;
        const double uk = ai[k];
        for (int j = 0; j <= l; ++j) g[j] += uk * ak[j];
      }
      for (int k = 0; k <= l; ++k) {
        double* ak = a + (size_t)k * n;
        const double w = ak[i];
        for (int j = 0; j <= l; ++j) ak[j] -= g[j] * w;
      }
    }
    d[i] = ai[i];
    ai[i] = 1.0;
    for (int j = 0; j <= l; ++j) a[(size_t)j * n + i] = ai[j] = 0.0;
  }
}

// Implicit QL on (d, e) with the rotations applied to the ROWS of zt (zt = Q^T on entry: row i is the
// i-th basis vector; on exit row i is the eigenvector of d[i]).  e[i] couples i-1 and i on entry.
#if defined(__x86_64__) && defined(__GNUC__) && !defined(__HIP_DEVICE_COMPILE__)
__attribute__((target_clones("arch=haswell", "default")))
#endif
static bool ql_implicit_rows(double* d, double* e, int n, double* zt) {
  for (int i = 1; i < n; ++i) e[i - 1] = e[i];
  e[n - 1] = 0.0;
  for (int l = 0; l < n; ++l) {
    int iter = 0, m;
    do {
      for (m = l; m < n - 1; ++m) {
        const double dd = std::fabs(d[m]) + std::fabs(d[m + 1]);
        if (std::fabs(e[m]) <= 2.3e-16 * dd) break;
      }
      if (m != l) {
        if (iter++ == 200) return false;
        double g = (d[l + 1] - d[l]) / (2.0 * e[l]);
        double r = std::hypot(g, 1.0);
        g = d[m] - d[l] + e[l] / (g + (g >= 0 ? std::fabs(r) : -std::fabs(r)));
        double s = 1.0, c = 1.0, p = 0.0;
        int i;
        for (i = m - 1; i >= l; --i) {
          double f = s * e[i], b = c * e[i];
          r = std::sqrt(f * f + g * g);
          if (!(r >= 1e-150 && r <= 1e150)) r = std::hypot(f, g);   // (squares out of range: the careful form)
          e[i + 1] = r;
          if (r == 0.0) {
            d[i + 1] -= p;
            e[m] = 0.0;
            break;
          }
          s = f / r;
          c = g / r;
          g = d[i + 1] - p;
          r = (d[i] - g) * s + 2.0 * c * b;
          p = s * r;
          d[i + 1] = g + p;
          g = c * r - b;
          double* z0 = zt + (size_t)i * n;
          double* z1 = zt + (size_t)(i + 1) * n;
          for (int k = 0; k < n; ++k) {
            const double fk = z1[k];
            z1[k] = s * z0[k] + c * fk;
            z0[k] = c * z0[k] - s * fk;
          }
        }
        if (r == 0.0 && i >= l) continue;
        d[l] -= p;
        e[l] = g;
        e[m] = 0.0;
      }
    } while (m != l);
  }
  return true;
}

bool sym_eigh_desc(const std::vector<double>& A, int n, std::vector<double>& w, std::vector<double>& Vt) {
  w.assign((size_t)std::max(n, 0), 0.0);
  Vt.assign((size_t)n * n, 0.0);
  if (n <= 0) return true;
  std::vector<double> a(A.begin(), A.begin() + (size_t)n * n), d((size_t)n), e((size_t)n), zt((size_t)n * n);
  for (int j = 0; j < n; ++j)
    for (int k = j + 1; k < n; ++k) a[(size_t)j * n + k] = a[(size_t)k * n + j];   // the lower triangle is the matrix
  std::vector<double> work((size_t)2 * n);
  householder_tridiag(a.data(), n, d.data(), e.data(), work.data());
  for (int i = 0; i < n; ++i)
    for (int k = 0; k < n; ++k) zt[(size_t)i * n + k] = a[(size_t)k * n + i];   // rows of zt = columns of Q
  if (!ql_implicit_rows(d.data(), e.data(), n, zt.data())) return false;
  std::vector<int> order((size_t)n);
  std::iota(order.begin(), order.end(), 0);
  std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return d[x] > d[y]; });
  for (int c = 0; c < n; ++c) {
    w[c] = d[order[c]];
    std::copy(zt.begin() + (size_t)order[c] * n, zt.begin() + (size_t)(order[c] + 1) * n, Vt.begin() + (size_t)c * n);
  }
  return true;
}

bool tridiag_eigh(std::vector<double>& d, std::vector<double>& e, int n, std::vector<double>& Z, bool last_row_only) {
  // Implicit QL with Wilkinson shifts (EISPACK tql2 structure).  e[i] couples (i-1, i), e[0] unused.
  // last_row_only: the rotations are applied to the bottom row of Z alone (Z is then 1 x n): the same arithmetic as
  // the full update restricted to that row, so a Lanczos error bound read from it does not depend on the mode.
  const int zr = last_row_only ? 1 : n;        // rows of Z kept
  const int k0 = last_row_only ? n - 1 : 0;    // first row of the full matrix they stand for
  Z.assign((size_t)zr * n, 0.0);
  for (int i = k0; i < n; ++i) Z[(size_t)(i - k0) * n + i] = 1.0;
  if (n == 0) return true;
  for (int i = 1; i < n; ++i) e[i - 1] = e[i];
  e[n - 1] = 0.0;
  for (int l = 0; l < n; ++l) {
    int iter = 0, m;
    do {
      for (m = l; m < n - 1; ++m) {
        const double dd = std::fabs(d[m]) + std::fabs(d[m + 1]);
        if (std::fabs(e[m]) <= 2.3e-16 * dd) break;
      }
      if (m != l) {
        if (iter++ == 200) return false;
        double g = (d[l + 1] - d[l]) / (2.0 * e[l]);
        double r = std::hypot(g, 1.0);
        g = d[m] - d[l] + e[l] / (g + (g >= 0 ? std::fabs(r) : -std::fabs(r)));
        double s = 1.0, c = 1.0, p = 0.0;
        int i;
        for (i = m - 1; i >= l; --i) {
          double f = s * e[i], b = c * e[i];
          r = std::hypot(f, g);
          e[i + 1] = r;
          if (r == 0.0) {
            d[i + 1] -= p;
            e[m] = 0.0;
            break;
          }
          s = f / r;
          c = g / r;
          g = d[i + 1] - p;
          r = (d[i] - g) * s + 2.0 * c * b;
          p = s * r;
          d[i + 1] = g + p;
          g = c * r - b;
          for (int k = 0; k < zr; ++k) {
            f = Z[(size_t)k * n + i + 1];
            Z[(size_t)k * n + i + 1] = s * Z[(size_t)k * n + i] + c * f;
            Z[(size_t)k * n + i] = c * Z[(size_t)k * n + i] - s * f;
          }
        }
        if (r == 0.0 && i >= l) continue;
        d[l] -= p;
        e[l] = g;
        e[m] = 0.0;
      }
    } while (m != l);
  }
  // sort ascending
  std::vector<int> order(n);
  std::iota(order.begin(), order.end(), 0);
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return d[a] < d[b]; });
  std::vector<double> d2(n), Z2((size_t)zr * n);
  for (int c = 0; c < n; ++c) {
    d2[c] = d[order[c]];
    for (int k = 0; k < zr; ++k) Z2[(size_t)k * n + c] = Z[(size_t)k * n + order[c]];
  }
  d.swap(d2);
  Z.swap(Z2);
  return true;
}

}  // namespace sapca

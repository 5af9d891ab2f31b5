"""Synthetic CSR inputs for the sparse-PCA hot path.

Everything here is a pure function of (seed, row index, column index): a
SplitMix64-style counter hash evaluated with int64 tensor arithmetic, so the
same call gives bit-identical matrices on a CPU, on a GPU and at any row
offset -- a rank can generate its own row shard of the global matrix without
touching the other shards, and a bounded CPU sample of the bench workload is
literally a row range of it.

Two generators:

* ``gapped_csr``  -- (k+1)-cluster count-like matrix with a planted spectral
  gap (the only kind of input on which "subspace angle <= 1e-4" is meaningful,
  SURVEY.md F5).
* ``flat_csr``    -- the reference's own test style (uniform positions, values
  U(-10,10) without |v|<=1e-10; cf. /root/reference/src/dimred/pca/sparse/mod.rs:493-537),
  flat spectrum, for throughput-only runs and Omega-injected parity.

Outputs are (row_offsets int64 [m+1], col_indices int32 [nnz], values [nnz])
torch tensors on the requested device; columns sorted and unique per row (the
nalgebra_sparse::CsrMatrix invariant).
"""
from __future__ import annotations

import torch

_MASK64 = (1 << 64) - 1


def _s64(x: int) -> int:
    """Reinterpret an unsigned 64-bit constant as the signed int64 torch wants."""
    x &= _MASK64
    return x - (1 << 64) if x >= (1 << 63) else x


_GOLD = _s64(0x9E3779B97F4A7C15)
_M1 = _s64(0xBF58476D1CE4E5B9)
_M2 = _s64(0x94D049BB133111EB)


def _lsr(z: torch.Tensor, s: int) -> torch.Tensor:
    """Logical shift right on int64 (torch's >> is arithmetic)."""
    return (z >> s) & ((1 << (64 - s)) - 1)


def mix64(z: torch.Tensor) -> torch.Tensor:
    """SplitMix64 finalizer on an int64 tensor (wrap-around arithmetic)."""
    z = z + _GOLD
    z = (z ^ _lsr(z, 30)) * _M1
    z = (z ^ _lsr(z, 27)) * _M2
    return z ^ _lsr(z, 31)


def hash_u01(seed: int, stream: int, idx: torch.Tensor) -> torch.Tensor:
    """Uniform [0,1) float64 from (seed, stream, idx); idx is an int64 tensor."""
    key = _s64((seed * 0xD1342543DE82EF95 + stream * 0x2545F4914F6CDD1D + 0x1234567) & _MASK64)
    z = mix64(idx ^ key)
    z = mix64(z + key)
    return _lsr(z, 11).to(torch.float64) * (1.0 / 9007199254740992.0)


def _assemble(rows_chunks, cols_chunks, vals_chunks, m, dtype, device):
    if rows_chunks:
        rows = torch.cat(rows_chunks)
        cols = torch.cat(cols_chunks).to(torch.int32)
        vals = torch.cat(vals_chunks).to(dtype)
    else:
        rows = torch.zeros(0, dtype=torch.int64, device=device)
        cols = torch.zeros(0, dtype=torch.int32, device=device)
        vals = torch.zeros(0, dtype=dtype, device=device)
    counts = torch.bincount(rows, minlength=m) if rows.numel() else torch.zeros(m, dtype=torch.int64, device=device)
    rowptr = torch.zeros(m + 1, dtype=torch.int64, device=device)
    rowptr[1:] = torch.cumsum(counts, 0)
    return rowptr, cols, vals


def gapped_csr(m, n, density, k, seed=42, *, centred=True, row_start=0, m_global=None,
               dtype=torch.float32, device="cpu", chunk_elems=1 << 27, w_hi=12.0, w_lo=7.0):
    """Rows [row_start, row_start+m) of the gapped count-like matrix.

    c = k+1 clusters (k when ``centred`` is False: the uncentred Lanczos case,
    SURVEY.md §8d).  Row i belongs to a hash-chosen cluster, column j to cluster
    j*c//n.  In-cluster entries are present with probability 0.6 (scaled down
    if the requested density cannot afford it) and carry U(0.5,1.5)*w with w
    geometric from ``w_hi`` down to ``w_lo`` across clusters; background entries
    fill up to the requested density with U(0,1) values.
    """
    device = torch.device(device)
    c = k + 1 if centred else k
    c = max(1, min(c, n))
    d_hi = min(0.6, density * c * 0.4)
    d_bg = max(0.0, (density - d_hi / c) / max(1e-12, (1.0 - 1.0 / c)))
    ratio = (w_lo / w_hi) ** (1.0 / max(1, c - 1))
    w = w_hi * (ratio ** torch.arange(c, dtype=torch.float64, device=device))
    col_cluster = (torch.arange(n, dtype=torch.int64, device=device) * c) // n
    rows_per_chunk = max(1, chunk_elems // max(1, n))
    R, C, V = [], [], []
    jj = torch.arange(n, dtype=torch.int64, device=device)[None, :]
    for r0 in range(0, m, rows_per_chunk):
        r1 = min(m, r0 + rows_per_chunk)
        gi = torch.arange(row_start + r0, row_start + r1, dtype=torch.int64, device=device)
        row_cluster = (hash_u01(seed, 1, gi) * c).to(torch.int64).clamp_(max=c - 1)
        flat = gi[:, None] * n + jj
        inblk = row_cluster[:, None] == col_cluster[None, :]
        p = torch.where(inblk, torch.full_like(inblk, d_hi, dtype=torch.float64),
                        torch.full_like(inblk, d_bg, dtype=torch.float64))
        nz = hash_u01(seed, 2, flat) < p
        ri, ci = torch.nonzero(nz, as_tuple=True)
        u = hash_u01(seed, 3, flat[ri, ci])
        v = torch.where(inblk[ri, ci], (0.5 + u) * w[row_cluster[ri]], u + 2.0 ** -20)
        R.append(ri + r0)
        C.append(ci)
        V.append(v)
        del flat, inblk, p, nz
    return _assemble(R, C, V, m, dtype, device)


def flat_csr(m, n, density, seed=42, *, row_start=0, dtype=torch.float32, device="cpu",
             chunk_elems=1 << 27):
    """Reference-style flat matrix: Bernoulli(density) positions, values U(-10,10)."""
    device = torch.device(device)
    rows_per_chunk = max(1, chunk_elems // max(1, n))
    R, C, V = [], [], []
    jj = torch.arange(n, dtype=torch.int64, device=device)[None, :]
    for r0 in range(0, m, rows_per_chunk):
        r1 = min(m, r0 + rows_per_chunk)
        gi = torch.arange(row_start + r0, row_start + r1, dtype=torch.int64, device=device)
        flat = gi[:, None] * n + jj
        nz = hash_u01(seed, 2, flat) < density
        ri, ci = torch.nonzero(nz, as_tuple=True)
        v = hash_u01(seed, 3, flat[ri, ci]) * 20.0 - 10.0
        v = torch.where(v.abs() <= 1e-10, torch.full_like(v, 1e-3), v)
        R.append(ri + r0)
        C.append(ci)
        V.append(v)
        del flat, nz
    return _assemble(R, C, V, m, dtype, device)


def bernoulli_mask(n, keep=0.6, seed=7, device="cpu"):
    """Feature mask for the masked configs: Bernoulli(keep) per column, seed 7."""
    j = torch.arange(n, dtype=torch.int64, device=device)
    return hash_u01(seed, 9, j) < keep


def gaussian_panel(n_rows, l, seed=42, dtype=torch.float64, device="cpu"):
    """n_rows x l standard-normal test panel (Box-Muller on the counter hash).

    Only used to *inject* an identical Omega into the oracle and the HIP path;
    the library's own Omega generator lives in csrc/rng.hip.
    """
    idx = torch.arange(n_rows * l, dtype=torch.int64, device=device)
    u1 = hash_u01(seed, 11, idx).clamp_(min=2.0 ** -53)
    u2 = hash_u01(seed, 12, idx)
    g = torch.sqrt(-2.0 * torch.log(u1)) * torch.cos(2.0 * torch.pi * u2)
    return g.reshape(n_rows, l).to(dtype)

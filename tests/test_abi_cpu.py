"""CPU tests of the boundary: the C-ABI library loads, exports every symbol include/sapca.h
declares, refuses to run without a GPU (no CPU fallback), and its host-only code works."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch  # noqa: F401  (first: one HIP runtime per process)

import sapca
from sapca import _lib as L
from sapca import ops

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    lib = L.load()
    header = open(os.path.join(ROOT, "include", "sapca.h")).read()
    declared = set(re.findall(r"\b(sapca_[a-z0-9_]+)\s*\(", header)) - {"sapca_allreduce_fn"}
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/sapca.h but not exported"
    assert declared == set(L.EXPORTED_SYMBOLS)
    assert lib.sapca_abi_version() == 3


def test_options_struct_layout_matches_header():
    o = L.default_options()
    assert o.struct_size == C.sizeof(L.Options)
    # builder defaults: sparse/mod.rs:392-401, pca/mod.rs:64-68
    assert (o.n_components, o.alpha, o.tolerance, o.random_seed, o.center, o.verbose, o.method) == \
        (50, 1.0, 1e-6, 42, 1, 0, L.LANCZOS)


@pytest.mark.skipif(torch.cuda.is_available(), reason="CPU-only check")
def test_no_cpu_fallback():
    with pytest.raises(L.SapcaError, match="no HIP device"):
        sapca.SparsePCABuilder.new().build()


def test_builder_defaults_and_fluent_surface():
    b = sapca.SparsePCABuilder.new()
    assert (b._n_components, b._alpha, b._tolerance, b._random_seed, b._center, b._verbose) == (50, 1.0, 1e-6, 42, True, False)
    assert b._svdmethod == sapca.SVDMethod.Lanczos() == sapca.SVDMethod()
    m = sapca.SVDMethod.Random(10, 7, sapca.PowerIterationNormalizer.QR)
    b2 = (sapca.MaskedSparsePCABuilder.new().n_components(5).alpha(1.5).tolerance(1e-4).random_seed(7)
          .center(False).verbose(True).svd_method(m).mask([True, False]))
    assert b2._n_components == 5 and b2._svdmethod.n_power_iterations == 7 and b2._mask.tolist() == [True, False]


def test_partition_rows_balances_nnz():
    rng = np.random.default_rng(0)
    counts = rng.integers(0, 50, 1000)
    ptr = np.concatenate([[0], np.cumsum(counts)])
    for parts in (1, 2, 3, 8):
        b = ops.partition_rows(ptr, parts).astype(np.int64)
        assert b[0] == 0 and b[-1] == 1000 and np.all(np.diff(b) >= 0)
        per = np.diff(ptr[b])
        assert per.sum() == ptr[-1]
        assert per.max() - per.min() <= 2 * counts.max()
    # degenerate: empty matrix and more parts than rows
    assert ops.partition_rows(np.zeros(5, np.int64), 2).tolist() == [0, 2, 4]
    b = ops.partition_rows(np.array([0, 3, 6]), 4).astype(np.int64)
    assert b[0] == 0 and b[-1] == 2 and np.all(np.diff(b) >= 0)


def test_cpp_mirror_compiles(tmp_path):
    """the header-only C++ mirror of the reference interface (estimators + resident workflow) against include/sapca.h"""
    import shutil
    import subprocess
    cxx = shutil.which("g++") or shutil.which("c++")
    if cxx is None:
        import pytest
        pytest.skip("no host C++ compiler")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "m.cpp"
    src.write_text('#include "%s"\n'
                   'template class sapca::ResidentCsr<float>;\ntemplate class sapca::ResidentCsr<double>;\n'
                   'int main() { auto b = sapca::SparsePCABuilder<float>().n_components(2); (void)b; return 0; }\n'
                   % os.path.join(root, "single-algebra_amd", "host", "cpp", "sapca.hpp"))
    r = subprocess.run([cxx, "-std=c++17", "-fsyntax-only", str(src)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr

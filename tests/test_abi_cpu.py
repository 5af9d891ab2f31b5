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
    assert lib.sapca_abi_version() == 4


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


def _header_functions():
    """(name -> argument count) of every function include/sapca.h declares: an independent, cruder parse than the
    generator's (comments stripped, `name(args);` with the commas counted)."""
    text = re.sub(r"/\*.*?\*/", " ", open(os.path.join(ROOT, "include", "sapca.h")).read(), flags=re.S)
    out = {}
    for name, args in re.findall(r"\b(sapca_[a-z0-9_]+)\s*\(([^()]*)\)\s*;", text):
        if name == "sapca_allreduce_fn":
            continue
        args = args.strip()
        out[name] = 0 if args in ("", "void") else args.count(",") + 1
    return out


def test_rust_sys_crate_declares_the_whole_header():
    """host/rust/sapca-sys/src/lib.rs is generated from the header (tools/gen_sapca_sys.py): every function, the same
    argument count, and the committed file is what the generator writes today."""
    import subprocess
    import sys
    rs = open(os.path.join(ROOT, "single-algebra_amd", "host", "rust", "sapca-sys", "src", "lib.rs")).read()
    block = rs[rs.index('extern "C" {'):]
    have = {}
    for name, args in re.findall(r"pub fn (sapca_\w+)\(([^()]*)\)", block):
        args = args.strip()
        have[name] = 0 if not args else args.count(":")
    want = _header_functions()
    assert set(have) == set(want), (sorted(set(want) - set(have)), sorted(set(have) - set(want)))
    for name, n in want.items():
        assert have[name] == n, f"{name}: header has {n} arguments, sapca-sys {have[name]}"
    assert len(want) >= 91
    # struct layouts: field counts and order of the two POD structs
    hdr = re.sub(r"/\*.*?\*/", " ", open(os.path.join(ROOT, "include", "sapca.h")).read(), flags=re.S)
    for struct, cls in (("sapca_options", L.Options), ("sapca_timings", L.Timings)):
        body = re.search(r"typedef\s+struct\s+%s\s*\{(.*?)\}" % struct, hdr, flags=re.S).group(1)
        c_fields = [re.match(r".*?(\w+)\s*(?:\[\d+\])?$", " ".join(d.split())).group(1) for d in body.split(";") if d.strip()]
        r_body = re.search(r"pub struct %s \{(.*?)\}" % struct, rs, flags=re.S).group(1)
        r_fields = re.findall(r"pub (\w+):", r_body)
        assert c_fields == r_fields == [f[0] for f in cls._fields_], struct
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_sapca_sys.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    # the safe wrapper only calls functions the sys crate declares
    wrapper = open(os.path.join(ROOT, "single-algebra_amd", "host", "rust", "sapca", "src", "lib.rs")).read()
    for name in set(re.findall(r"ffi::\$?(sapca_[a-z0-9_]+)\b", wrapper) + re.findall(r"\b(sapca_[a-z0-9_]+_f(?:32|64))\b", wrapper)) - \
            {"sapca_handle", "sapca_multi", "sapca_options"}:
        assert name in have, f"host/rust/sapca uses {name}, which include/sapca.h does not declare"


def test_rust_builders_are_infallible_like_the_reference():
    """`SparsePCABuilder::build()` / `MaskedSparsePCABuilder::build()` return the estimator itself
    (/root/reference/src/dimred/pca/sparse/mod.rs:470-483, sparse_masked/mod.rs:146-160; README.md:56-66 writes
    `.build();` and then `pca.fit_transform(..)`): the wrapper must not hand back a Result there."""
    wrapper = open(os.path.join(ROOT, "single-algebra_amd", "host", "rust", "sapca", "src", "lib.rs")).read()
    assert re.search(r"pub fn build\(self\) -> SparsePCA<T>\s*\{", wrapper)
    assert re.search(r"pub fn build\(self\) -> MaskedSparsePCA<T>\s*\{", wrapper)
    assert "pub fn build(self) -> Result" not in wrapper
    # the reference's constructors, same arity (sparse/mod.rs:63-71, sparse_masked/mod.rs:214-223)
    assert re.search(r"pub fn new\(n_components: usize, alpha: T, tollerance: Option<T>, random_seed: Option<u32>, center: bool, verbose: bool,\s*svdmethod: SVDMethod\) -> Self", wrapper)
    assert re.search(r"pub fn new\(n_components: usize, alpha: T, tollerance: Option<T>, random_seed: Option<u32>, mask: Vec<bool>, center: bool,\s*verbose: bool, svd_method: SVDMethod\) -> Self", wrapper)


def test_generated_sweep_loop_is_what_its_generator_writes(tmp_path):
    """csrc/spmm_dq2_gen.h (10.7 k lines of inline asm) is committed generator output: regenerate and compare."""
    import subprocess
    import sys
    out = tmp_path / "gen.h"
    env = {k: v for k, v in os.environ.items() if not k.startswith("DQ2_")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_spmm_dq2.py"), str(out)], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    committed = open(os.path.join(ROOT, "single-algebra_amd", "csrc", "spmm_dq2_gen.h")).read()
    assert out.read_text() == committed, "spmm_dq2_gen.h is stale: run python tools/gen_spmm_dq2.py"


def test_release_library_reads_no_experiment_switches():
    """csrc/switches.h: the SAPCA_* route / layout switches exist only in the -DSAPCA_DEBUG_SWITCHES variant.  The release
    library does not even carry their names; what it does read is listed in switches.h (two deployment escapes)."""
    import glob
    src = os.path.join(ROOT, "single-algebra_amd", "csrc")
    sites = []
    for f in glob.glob(os.path.join(src, "*")):
        for i, line in enumerate(open(f, errors="replace"), 1):
            if re.search(r"\bgetenv\s*\(", line):
                sites.append((os.path.basename(f), i, line.strip()))
    names = sorted(set(re.findall(r'getenv\("(SAPCA_\w+)"\)', " ".join(s[2] for s in sites))))
    assert names == ["SAPCA_AT_OVERLAP", "SAPCA_MULTI_INPROCESS"], names
    assert len(sites) <= 8, sites
    rel = open(L.LIB_PATH, "rb").read()
    for name in (b"SAPCA_TILED_FMT", b"SAPCA_AT_NATURAL", b"SAPCA_TRANSPOSE_GATHER", b"SAPCA_MASK_TRANSPOSE_FIRST", b"SAPCA_COMM_FORCE_RCCL",
                 b"SAPCA_RCCL_LIBRARY"):   # (the tests' stand-in for librccl is reachable from the debug build only)
        assert name not in rel, f"the release library carries the switch {name.decode()}"
    assert b"SAPCA_AT_OVERLAP" in rel
    if os.path.exists(L.DEBUG_LIB_PATH):
        dbg = open(L.DEBUG_LIB_PATH, "rb").read()
        assert b"SAPCA_TILED_FMT" in dbg and b"SAPCA_COMM_FORCE_RCCL" in dbg and b"SAPCA_RCCL_LIBRARY" in dbg
        lib = L.load_debug()
        for name in L.EXPORTED_SYMBOLS:
            assert hasattr(lib, name), name


def test_stand_in_librccl_is_test_infrastructure_only():
    """tests/fake_rccl is reached through SAPCA_RCCL_LIBRARY by the debug build and by nothing else: no product source, build
    file or Python module names it, and it exports exactly the entry points csrc/comm.cpp binds."""
    import glob
    prod = glob.glob(os.path.join(ROOT, "single-algebra_amd", "**", "*"), recursive=True) + [os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "__graft_entry__.py")]
    for f in prod:
        if os.path.isfile(f) and not f.endswith((".so", ".o")) and "/build" not in f:
            txt = open(f, errors="replace").read()
            assert "fake_rccl" not in txt or f.endswith(("comm.cpp", "switches.h")), f
    src = open(os.path.join(ROOT, "tests", "fake_rccl", "fake_rccl.c")).read()
    exported = set(re.findall(r'extern "C" __attribute__\(\(visibility\("default"\)\)\) [\w\s\*]+?\b(nccl\w+)\(', src))
    bound = set(re.findall(r'dlsym\(a\.so, "(nccl\w+)"\)', open(os.path.join(ROOT, "single-algebra_amd", "csrc", "comm.cpp")).read()))
    assert exported == bound and len(bound) == 8, (exported, bound)


def test_bench_probes_for_the_reference_binary_and_says_why_not(monkeypatch):
    """bench.py looks for cargo + a registry before any GPU call and reports the reference binary as
    unavailable (with the reason) where they are missing -- as in this image."""
    import importlib.util
    import shutil
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    monkeypatch.setattr(shutil, "which", lambda name: None)
    r = bench.reference_binary_baseline()
    assert r["available"] is False and "cargo" in r["reason"]
    got, why = bench.measure_traffic("c2", 11)      # no rocprofv3 on PATH (patched): the committed figure is used instead
    assert got is None and "rocprofv3" in why
    sweep, total = bench.alg_bytes(200_000, 20_000, 119_990_071, 60, 50, 4)
    assert sweep == 119_990_071 * 8 + 200_001 * 8 + 20_000 * 60 * 4 + 200_000 * 60 * 4 == 1_014_320_576   # SURVEY.md 8d formula at C2

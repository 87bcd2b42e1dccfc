"""CPU-side checks of the drop-in boundary: the C-ABI library loads here (no GPU), exports every
symbol include/oisat.h declares, the ctypes table covers exactly that set, and the product package
never reaches into oracle/."""
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "oisat.h")
PKG = os.path.join(ROOT, "oi-sat-gmi_amd")


def header_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(oisat_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib_path():
    from oisatgmi import _hip
    p = _hip.library_path()
    if not os.path.exists(p):
        import __graft_entry__ as g
        g.build()
    assert os.path.exists(p), "liboisat_hip.so is not built (run __graft_entry__.build())"
    return p


def test_header_declares_the_expected_surface():
    syms = header_symbols()
    for must in ("oisat_init", "oisat_oi_curve", "oisat_oi_apply", "oisat_nanmean_stack", "oisat_error_average",
                 "oisat_boxfilter_pick", "oisat_nn_query", "oisat_cov_build", "oisat_potrf", "oisat_gain_solve",
                 "oisat_apply_increment"):
        assert must in syms


def test_library_exports_every_header_symbol(lib_path):
    out = subprocess.run(["nm", "-D", "--defined-only", lib_path], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r"\bT (oisat_[a-z0-9_]+)", out))
    missing = [s for s in header_symbols() if s not in exported]
    assert not missing, f"declared in include/oisat.h but not exported: {missing}"


def test_ctypes_table_matches_header(lib_path):
    from oisatgmi import _hip
    assert sorted(_hip.SIGNATURES) == header_symbols()
    lib = _hip.load_library()                      # dlopen + prototype declaration, no compute call
    assert lib.oisat_version().decode().startswith("oisat-hip")
    assert isinstance(lib.oisat_last_error(), bytes)


def test_library_targets_gfx950(lib_path):
    blob = open(lib_path, "rb").read()
    assert b"gfx950" in blob
    for other in (b"gfx942", b"gfx90a", b"sm_90", b"sm_80"):
        assert other not in blob


def test_product_never_imports_the_oracle():
    """... and neither do the helper scripts under tools/: whatever uses the oracle as a checker lives under tests/."""
    bad = []
    for dp, _, fns in list(os.walk(PKG)) + list(os.walk(os.path.join(os.path.dirname(PKG), "tools"))):
        for fn in fns:
            if fn.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dp, fn), errors="ignore").read()
                if re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M) or "oi_oracle" in txt:
                    bad.append(os.path.join(dp, fn))
    assert not bad, f"product files referencing oracle/: {bad}"


def test_no_gpu_means_loud_failure(lib_path):
    """Without a device the product must raise, not compute on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    from oisatgmi import _hip
    from oisatgmi.optimal_interpolation import OI
    _hip.reset_context()
    a = np.ones((4, 4))
    with pytest.raises(_hip.OisatUnavailable):
        OI(a.copy(), a.copy(), a.copy(), a.copy(), regularization_on=False)


def test_missing_library_is_loud(monkeypatch, tmp_path):
    from oisatgmi import _hip
    monkeypatch.setenv("OISAT_LIB", str(tmp_path / "nope.so"))
    monkeypatch.setattr(_hip, "_lib", None)
    with pytest.raises(_hip.OisatUnavailable):
        _hip.load_library()


def test_header_is_plain_c_and_a_c_client_links(lib_path, tmp_path):
    """include/oisat.h must be usable from C (not only C++): compile and link a C99 client against the library."""
    src = os.path.join(ROOT, "tests", "c", "abi_smoke.c")
    exe = str(tmp_path / "abi_smoke")
    libdir = os.path.dirname(lib_path)
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), src, "-o", exe,
                        "-L", libdir, "-loisat_hip", "-lm", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    import torch
    if not torch.cuda.is_available():          # no GPU here: the client must fail loudly in oisat_init, not crash
        run = subprocess.run([exe], capture_output=True, text=True)
        assert run.returncode == 1 and "oisat_init" in run.stderr

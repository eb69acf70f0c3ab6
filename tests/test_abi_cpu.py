"""The C-ABI shared library loads on a machine without a GPU and exports every symbol that
include/mulactseg_hip.h declares; the ctypes signature table binds all of them.  No compute calls."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "mulactseg_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mas_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_expected_surface():
    syms = declared_symbols()
    for must in ("mas_single_pass_accum", "mas_region_finalize_weighted", "mas_class_prob_sum", "mas_bvsb_region_accum",
                 "mas_region_finalize", "mas_region_keys", "mas_sort_keys_desc", "mas_budget_walk", "mas_partial_loss_fwd",
                 "mas_partial_loss_bwd", "mas_logits_iou_counts"):
        assert must in syms
    assert len(syms) >= 20


def test_library_exports_every_declared_symbol():
    from mulactseg_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    import torch  # noqa: F401  (one HIP runtime per process, see _lib.load)
    lib = ctypes.CDLL(_lib.LIB_PATH)
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, missing
    # and the ctypes table covers the header one to one
    assert sorted(_lib.SIGNATURES) == declared_symbols()
    assert _lib.load().mas_abi_version() == _lib.ABI_VERSION
    header = open(os.path.join(ROOT, "include", "mulactseg_hip.h")).read()
    assert int(re.search(r"#define MAS_ABI_VERSION (\d+)", header).group(1)) == _lib.ABI_VERSION
    # the test-support kernel is not part of the product ABI
    assert not hasattr(lib, "mas_test_occupy") and os.path.exists(os.path.join(ROOT, "tests", "libmulactseg_test.so"))
    assert _lib.load().mas_error_string(-3).decode().startswith("class count")


def test_argument_errors_are_reported_without_touching_the_gpu():
    from mulactseg_amd import _lib
    lib = _lib.load()
    assert lib.mas_class_prob_sum(None, 1, 20, 4, 4, 10.0, None, None) == -1          # null pointer
    assert lib.mas_select_workspace_bytes(0) == 0
    with pytest.raises(_lib.MulActSegHipError):
        _lib.check(-2, "probe")


def test_a_library_of_another_abi_version_is_refused(tmp_path, monkeypatch):
    """load() compares mas_abi_version() with the version the ctypes table was written against BEFORE it binds anything: a stale
    variant build (MAS_LIB) that exports the same names with other argument lists must not bind silently."""
    import subprocess
    from mulactseg_amd import _lib
    src = tmp_path / "stale.c"
    src.write_text("int mas_abi_version(void) { return 1; }\n")
    so = tmp_path / "libstale.so"
    subprocess.check_call(["gcc", "-shared", "-fPIC", "-o", str(so), str(src)])
    monkeypatch.setattr(_lib, "LIB_PATH", str(so))
    monkeypatch.setattr(_lib, "_lib", None)
    with pytest.raises(_lib.MulActSegHipError, match="ABI version 1"):
        _lib.load()

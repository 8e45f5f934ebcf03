"""The C-ABI library loads and exports every symbol include/nerfsig.h declares (no compute calls: no GPU here)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "nerfsig.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"^(?:int|size_t|const char \*|void \*)\s*(\w+)\s*\(", text, flags=re.M)))


@pytest.fixture(scope="module")
def native():
    from nerf_signature_amd import build, _native
    build.build()
    return _native


def test_header_and_loader_agree(native):
    assert _declared() == sorted(native.SIGNATURES)


def test_every_declared_symbol_is_exported(native):
    assert native.verify_exports() == _declared()
    assert native.fn("nsig_abi_version")() == 1


def test_host_only_queries(native):
    assert native.fn("rm_march_train_scratch_bytes")(4096, 1024) == 4096 * 1024 * 4
    assert native.fn("mlp_packed_bytes")() == 3 * (24 + 24) * 64 * 16          # split-bf16 hi + lo and fp16 fragments, forward + backward
    before = native.fn("mlp_get_precision")()
    assert before in (0, 1)
    native.set_mlp_precision("bf16x3")
    assert native.fn("mlp_get_precision")() == 0 and "split-bf16" in native.mlp_precision_name() and native.mlp_mfma_per_wave() == (72, "bf16")
    native.set_mlp_precision("f16")
    assert native.fn("mlp_get_precision")() == 1 and native.mlp_mfma_per_wave() == (24, "f16")
    with pytest.raises(ValueError):
        native.call("mlp_set_precision", 7)
    native.call("mlp_set_precision", before)


def test_argument_validation_needs_no_gpu(native):
    with pytest.raises(ValueError, match="null pointer"):
        native.call("rm_morton3D", None, 4, None, None)
    with pytest.raises(ValueError, match="out of range"):
        native.call("hg_codebook_presum", (native._vp * 1)(), 0, native._vp(16), None)
    # field_bwd_wgrad (the fused stage-1 backward): null pointers and its point range (32-bit lane byte offsets) are refused before anything is launched
    d = native._vp(256)
    with pytest.raises(ValueError, match="null pointer"):
        native.call("field_bwd_wgrad", 64, None, None, d, d, d, d, d, d, d, d, d, d, d, d, d, d, None)
    with pytest.raises(ValueError, match="out of range"):
        native.call("field_bwd_wgrad", (1 << 26) + 1, None, d, d, d, d, d, d, d, d, d, d, d, d, d, d, d, None)
    with pytest.raises(ValueError, match="16-byte aligned"):
        native.call("field_bwd_wgrad", 64, None, d, d, d, d, d, d, native._vp(264), d, d, d, d, d, d, d, d, None)
    # the slice owners store rows as 16-byte vectors: every gradient table of hg_levels_scatter has to be aligned
    tables = (native._vp * 16)(*([256] * 15 + [264]))
    with pytest.raises(ValueError, match="gradient table 15 must be 16-byte aligned"):
        native.call("hg_levels_scatter", d, 64, None, 1.0, d, 64, d, tables, None)
    # ... and the form with the optimiser step inside the owners refuses a misaligned moment table before it touches a step count
    good, bad = (native._vp * 16)(*([256] * 16)), (native._vp * 16)(*([256] * 3 + [264] + [256] * 12))
    with pytest.raises(ValueError, match="table 3 must be 16-byte aligned"):
        native.call("hg_levels_scatter_adam", d, 64, None, 1.0, d, 64, d, good, bad, good, good, d, 0.9, 0.99, 1e-15, 1.0, d, None)
    with pytest.raises(ValueError, match="null pointer"):
        native.call("hg_levels_scatter_adam", d, 64, None, 1.0, d, 64, d, good, good, good, good, None, 0.9, 0.99, 1e-15, 1.0, d, None)
    assert native.fn("field_bwd_wgrad_scratch_bytes")(1) == 12 * 1024 * 4 and native.fn("field_bwd_wgrad_scratch_bytes")(10 ** 6) == 256 * 12 * 1024 * 4


def test_missing_library_fails_loudly(native, monkeypatch):
    monkeypatch.setattr(native, "_lib", None)
    monkeypatch.setattr(native, "_bound", {})
    monkeypatch.setattr(native, "LIB_PATH", "/nonexistent/libnerfsig.so")
    with pytest.raises(native.NativeError, match="no fallback"):
        native.fn("rm_morton3D")


def test_product_does_not_import_the_oracle():
    """Only tests/, smoke() and bench.py's cpu_baseline leg may use oracle/: the package never imports, links or
    dlopens it (comments may cite it)."""
    pkg = os.path.join(ROOT, "nerf_signature_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            path = os.path.join(dirpath, f)
            if f.endswith(".py"):
                for line in open(path):
                    code = line.split("#")[0]
                    assert not re.search(r"^\s*(from|import)\s+oracle\b", code), path
                    assert "liboracle" not in code and "raymarch_ref" not in code and "field_ref" not in code, path
            elif f.endswith((".hip", ".h")):
                for line in open(path):
                    assert not (line.lstrip().startswith("#include") and "oracle" in line), path

"""CPU checks of the drop-in boundary: the C-ABI library loads, exports every symbol include/evlm_hip.h declares,
the ctypes table covers them all, and the product path refuses to run without the GPU kernels."""
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "evlm_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(evlm_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from efficientvlm_amd import _lib
    lib = _lib.load()
    names = declared_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/evlm_hip.h but not exported by libevlm_hip.so"
    bound = set(_lib.SIGNATURES) | {"evlm_last_error", "evlm_abi_version", "evlm_gemm_last_kernel"}
    assert set(names) == bound, sorted(set(names) ^ bound)
    assert lib.evlm_abi_version() == _lib.ABI_VERSION == 9


def test_struct_layout_matches_header():
    """field order of the ctypes structs == field order of the C structs"""
    from efficientvlm_amd import _lib
    src = open(os.path.join(ROOT, "include", "evlm_hip.h")).read()
    for cname, st in (("evlm_gemm_args", _lib.GemmArgs), ("evlm_attn_fwd_args", _lib.AttnFwdArgs),
                      ("evlm_attn_bwd_args", _lib.AttnBwdArgs), ("evlm_xattn_fused_args", _lib.XAttnFusedArgs)):
        body = re.search(r"typedef struct \{((?:(?!typedef struct).)*?)\}\s*" + cname, src, flags=re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        fields = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            names = re.sub(r"^(const\s+)?[A-Za-z_0-9]+\s*\**", "", decl)
            fields += [n.strip().lstrip("*").strip() for n in names.split(",")]
        assert fields == [f[0] for f in st._fields_], cname


def test_integration_doc_struct_matches_the_binding():
    """the GemmArgs example in INTEGRATION.md (what a maintainer would paste) has the fields of the shipped binding, in
    order and with the same ctypes: a struct one field short makes the library read past it"""
    import ctypes as C
    from efficientvlm_amd import _lib
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    block = re.search(r"class GemmArgs\(C\.Structure\):.*?_fields_ = \[(.*?)\]\n", doc, flags=re.S).group(1)
    fields = re.findall(r'\("([a-zA-Z_0-9]+)",\s*C\.([a-z_0-9]+)\)', block)
    assert [f for f, _ in fields] == [f[0] for f in _lib.GemmArgs._fields_]
    for (name, ty), (_, want) in zip(fields, _lib.GemmArgs._fields_):
        assert getattr(C, ty) is want, name


def test_ops_refuse_cpu_tensors():
    from efficientvlm_amd import ops
    x = torch.randn(4, 8)
    w = torch.nn.Parameter(torch.randn(8, 8))
    with pytest.raises(RuntimeError, match="GPU only"):
        ops.linear(x, w, None)
    with pytest.raises(RuntimeError, match="GPU only"):
        ops.layer_norm(x, torch.ones(8), torch.zeros(8), 1e-5)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from efficientvlm_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU/PyTorch fallback"):
        _lib.load()

"""The C-ABI library loads without a GPU and exports every symbol include/uenc.h declares (no compute calls here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "uni-encoder-code_amd", "uenc", "libuenc_hip.so")


def _declared():
    src = open(os.path.join(ROOT, "include", "uenc.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(uenc_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(LIB):
        import __graft_entry__ as g
        g.build()
    return ctypes.CDLL(LIB)


def test_header_symbols_exported(lib):
    names = _declared()
    assert len(names) >= 14
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/uenc.h but not exported"


@pytest.mark.parametrize("cc,lang,std", [("gcc", "c", "-std=c99"), ("gcc", "c", "-std=c11"), ("g++", "c++", "-std=c++11")])
def test_header_is_valid_c_and_cxx(cc, lang, std):
    """include/uenc.h compiles on its own as C and as C++ (round 2 shipped a header with a declaration pasted into another one's
    parameter list: nothing compiled it)."""
    import subprocess
    r = subprocess.run([cc, "-fsyntax-only", "-x", lang, std, "-Wall", "-Wextra", "-Werror", os.path.join(ROOT, "include", "uenc.h")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_every_kernel_source_is_compiled_against_the_header():
    """Each csrc/*.hip includes common.h, which includes include/uenc.h with UENC_STREAM_T = hipStream_t: a definition whose
    signature drifts from its declaration is a hipcc error ("conflicting types"), so the build is the type check."""
    csrc = os.path.join(ROOT, "uni-encoder-code_amd", "csrc")
    common = open(os.path.join(csrc, "common.h")).read()
    assert '#include "../../include/uenc.h"' in common and "#define UENC_STREAM_T hipStream_t" in common
    hips = [f for f in os.listdir(csrc) if f.endswith(".hip")]
    assert len(hips) >= 12
    defined = set()
    for f in hips:
        src = open(os.path.join(csrc, f)).read()
        assert '#include "common.h"' in src, f
        defined |= set(re.findall(r'extern "C"\s+(?:const\s+)?\w+\*?\s+(uenc_[a-z0-9_]+)\s*\(', src))
    # nothing is exported that the header does not declare, and nothing declared lacks a definition
    assert defined == set(_declared()), (defined ^ set(_declared()))


def test_python_binding_matches_header():
    from uenc import capi
    assert set(capi.exported_symbols()) == set(_declared())
    assert capi.lib.uenc_version() >= 1
    capi.lib.uenc_arch.restype = ctypes.c_char_p
    assert capi.lib.uenc_arch() == b"gfx950"


def test_invalid_arguments_are_refused_without_a_gpu(lib):
    # argument validation happens before any HIP call: NULL pointers / bad shapes return -1
    assert lib.uenc_gemm_nt(None, 1, 0, None, 0, None, 1, 0, 0, 0, 0, None, 0, None, 0, None, 0, ctypes.c_float(1.0), 1, 0, None) == -1
    assert lib.uenc_layernorm_fwd(None, 0, None, 0, None, None, None, None, 0, None, 0, 0, ctypes.c_float(1e-5), None) == -1
    assert lib.uenc_window_attn_np(12) == 144 and lib.uenc_window_attn_np(7) == 64


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
    """The product has no fallback: importing the binding without the .so raises ImportError."""
    import importlib.util
    src = os.path.join(ROOT, "uni-encoder-code_amd", "uenc", "capi.py")
    dst = tmp_path / "capi_copy.py"
    dst.write_text(open(src).read())
    spec = importlib.util.spec_from_file_location("capi_copy", dst)
    mod = importlib.util.module_from_spec(spec)
    with pytest.raises(ImportError):
        spec.loader.exec_module(mod)

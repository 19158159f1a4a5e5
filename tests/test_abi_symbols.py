"""The shared library loads and exports every entry point include/vs_abi.h declares
(no compute calls: this runs without a GPU)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_entry_points():
    text = open(os.path.join(ROOT, "include", "vs_abi.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\bint32_t\s+(vs_\w+)\s*\(", text)))


def test_header_declares_the_expected_surface():
    names = declared_entry_points()
    assert len(names) >= 29
    for must in ("vs_init", "vs_store_create", "vs_store_open", "vs_store_upload", "vs_store_seal",
                 "vs_search", "vs_search_roaring", "vs_search_batch", "vs_search_device", "vs_search_stores",
                 "vs_merge_topk_device", "vs_roaring_to_mask", "vs_cbo_roaring_to_mask"):
        assert must in names


def test_library_exports_every_declared_symbol():
    from search_engine_amd import abi
    raw = ctypes.CDLL(abi.LIB_PATH)
    for name in declared_entry_points():
        assert hasattr(raw, name), f"libvs_amd.so does not export {name}"
        assert name in abi.SIGNATURES, f"abi.py has no binding for {name}"
    assert sorted(abi.SIGNATURES) == declared_entry_points()


def test_no_cpu_fallback_without_a_device():
    """Without a usable GPU the library refuses to start instead of computing on the CPU."""
    import torch
    from search_engine_amd import abi
    if torch.cuda.is_available():
        abi.Context(0).close()
        return
    try:
        abi.Context(0)
    except abi.VsError as e:
        assert e.code == abi.VS_E_DEVICE
    else:
        raise AssertionError("vs_init succeeded without a GPU")


def test_host_binding_imports():
    from search_engine_amd import _host
    assert _host.distribution_shift(0.85, 0.1, 1.0) == 1.0


def test_header_is_plain_c(tmp_path):
    """The boundary is a C ABI: the header compiles as C99 on its own, with no C++ or HIP types."""
    import subprocess
    src = tmp_path / "use_abi.c"
    src.write_text('#include "vs_abi.h"\n'
                   "int probe(void) { vs_store *s = 0; vs_ctx *c = 0; vs_mask *m = 0;\n"
                   "  return (s == 0) + (c == 0) + (m == 0) + VS_MAX_K; }\n")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-fsyntax-only",
                    "-I", os.path.join(ROOT, "include"), str(src)], check=True)


def test_the_rust_binding_in_the_integration_notes_matches_the_header():
    """INTEGRATION.md shows the `extern "C"` block a maintainer would add on the Rust side. Every
    function it declares must exist in include/vs_abi.h with the same number of parameters and the
    same widths in the same order (u16/u32/u64/i32/f32 scalars, pointers, usize = size_t)."""
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    header = open(os.path.join(root, "include", "vs_abi.h")).read()
    header = re.sub(r"/\*.*?\*/", " ", header, flags=re.S)

    def c_kind(p):
        p = p.strip()
        if "*" in p:
            return "ptr"
        t = p.rsplit(" ", 1)[0].replace("const ", "").strip()
        return {"uint16_t": "u16", "uint32_t": "u32", "uint64_t": "u64", "int32_t": "i32", "float": "f32",
                "size_t": "usize", "uint8_t": "u8"}[t]

    def rust_kind(p):
        t = p.split(":", 1)[1].strip()
        return "ptr" if t.startswith("*") else t

    rust = re.findall(r"pub fn (vs_\w+)\((.*?)\)\s*->\s*i32;", text, flags=re.S)
    assert len(rust) >= 15
    for name, params in rust:
        m = re.search(r"int32_t\s+" + name + r"\s*\((.*?)\)\s*;", header, flags=re.S)
        assert m, f"{name} is not declared in include/vs_abi.h"
        c_params = [c_kind(p) for p in m.group(1).split(",")]
        r_params = [rust_kind(p) for p in re.sub(r"\s+", " ", params).split(",") if p.strip()]
        assert c_params == r_params, (name, c_params, r_params)


def test_the_ctypes_signatures_match_the_header():
    """abi.py's argtypes are what every GPU test calls through: same parameter count as the header
    and the same scalar widths at the same positions (a pointer is a pointer)."""
    import re
    from search_engine_amd import abi
    header = re.sub(r"/\*.*?\*/", " ", open(os.path.join(ROOT, "include", "vs_abi.h")).read(), flags=re.S)
    scalars = {"uint16_t": ctypes.c_uint16, "uint32_t": ctypes.c_uint32, "uint64_t": ctypes.c_uint64,
               "int32_t": ctypes.c_int32, "float": ctypes.c_float, "size_t": ctypes.c_size_t, "uint8_t": ctypes.c_uint8}
    for name, argtypes in abi.SIGNATURES.items():
        m = re.search(r"int32_t\s+" + name + r"\s*\((.*?)\)\s*;", header, flags=re.S)
        assert m, name
        params = [p.strip() for p in m.group(1).split(",")]
        if params == ["void"]:
            params = []
        assert len(params) == len(argtypes), (name, params, argtypes)
        for p, t in zip(params, argtypes):
            if "*" in p:
                is_pointer = t in (ctypes.c_void_p, ctypes.c_char_p) or hasattr(t, "contents") or \
                    issubclass(t, (ctypes._Pointer, ctypes.Array)) or t is ctypes.c_void_p
                assert is_pointer, (name, p, t)
            else:
                want = scalars[p.rsplit(" ", 1)[0].replace("const ", "").strip()]
                assert ctypes.sizeof(t) == ctypes.sizeof(want) and t is not ctypes.c_void_p, (name, p, t)
                assert (t in (ctypes.c_float,)) == (want is ctypes.c_float), (name, p, t)

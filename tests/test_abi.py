"""CPU tests of the drop-in boundary: the C-ABI library loads, exports every symbol the header
declares, and refuses to run without a device (no compute calls here)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT, pkg


def _header_symbols():
    txt = open(os.path.join(ROOT, "include", "sl3d.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(sl3d_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol(scanner_mod):
    declared = _header_symbols()
    assert declared, "no prototypes found in include/sl3d.h"
    assert sorted(scanner_mod.ABI_SYMBOLS) == declared, "scanner.ABI_SYMBOLS out of sync with include/sl3d.h"
    lib = ctypes.CDLL(scanner_mod.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), f"libsl3d.so does not export {name}"


def test_version_and_strerror(scanner_mod):
    L = scanner_mod.load_library()
    assert L.sl3d_version().decode().startswith("0.")
    assert L.sl3d_strerror(0) == b"ok"
    assert b"no CPU fallback" in L.sl3d_strerror(-2)


def test_versions_agree(scanner_mod):
    """One version: SL3D_VERSION_STRING of include/sl3d.h == sl3d_version() of the library == the package's __version__."""
    txt = open(os.path.join(ROOT, "include", "sl3d.h")).read()
    hdr = re.search(r'#define\s+SL3D_VERSION_STRING\s+"([^"]+)"', txt).group(1)
    assert scanner_mod.load_library().sl3d_version().decode().split()[0] == hdr
    assert pkg().__version__ == hdr


def test_create_rejects_bad_config(scanner_mod):
    L = scanner_mod.load_library()
    h = ctypes.c_void_p()
    cfg = scanner_mod.Config(0, 10, 0, 0, 0, 0, 64, 64, 3, 3, 3, 8, 8, 0, 0, 1, 0, 0, None)
    assert L.sl3d_create(ctypes.byref(cfg), ctypes.byref(h)) == -1
    assert L.sl3d_create(None, ctypes.byref(h)) == -1


def test_no_cpu_fallback(scanner_mod):
    """Without a HIP device construction must fail loudly (SL3D_E_NO_DEVICE), never compute on the CPU."""
    L = scanner_mod.load_library()
    h = ctypes.c_void_p()
    cfg = scanner_mod.Config(64, 32, 0, 0, 0, 0, 64, 64, 3, 3, 3, 8, 8, 0, 0, 1, 0, 0, None)
    rc = L.sl3d_create(ctypes.byref(cfg), ctypes.byref(h))
    if rc == 0:  # a GPU is present (GPU box): fine, clean up
        L.sl3d_destroy(h)
        pytest.skip("a HIP device is present")
    assert rc == -2
    with pytest.raises(scanner_mod.Sl3dError):
        scanner_mod.Scanner(64, 32, 64, 64, 3, 3, 8, 8)


def test_product_does_not_import_the_oracle():
    """The oracle is test infrastructure: nothing under 3dscan_amd/ may reference it."""
    base = os.path.join(ROOT, "3dscan_amd")
    for dirpath, _, files in os.walk(base):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".hpp")) or f == "Makefile":
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle" not in txt.lower(), f
    # the measurement tools may not use it either (only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg)
    for f in os.listdir(os.path.join(ROOT, "tools")):
        if f.endswith((".py", ".sh")):
            txt = open(os.path.join(ROOT, "tools", f), errors="ignore").read()
            assert "from oracle" not in txt and "import oracle" not in txt and "oracle." not in txt, f


def test_pattern_counts_host_function():
    """sl3d_pattern_counts is host-only (no GPU needed): equal to the oracle's restatement of allocate_memory()'s float
    arithmetic, including the exact powers of two where logf/logf decides the count."""
    from oracle import oracle as O
    scm = pkg("scanner")
    for extent in (1, 2, 255, 256, 257, 720, 1024, 1280, 1920, 4096, 8192):
        for fw in (1, 2, 3, 4, 8, 32, 100):
            assert scm.pattern_counts(extent, fw) == O.pattern_counts(extent, fw), (extent, fw)


def test_group_create_rejects_bad_arguments(scanner_mod):
    """sl3d_group_create: argument errors come back as status codes before any device is touched; without a HIP device a
    valid request fails with SL3D_E_NO_DEVICE (no CPU fallback for groups either)."""
    L = scanner_mod.load_library()
    h = ctypes.c_void_p()
    cfg = scanner_mod.Config(64, 32, 0, 0, 0, 0, 64, 64, 3, 3, 3, 8, 8, 0, 0, 1, 0, 0, None)
    devs = (ctypes.c_int * 2)(0, 0)
    assert L.sl3d_group_create(None, devs, 2, ctypes.byref(h)) == -1
    assert L.sl3d_group_create(ctypes.byref(cfg), None, 2, ctypes.byref(h)) == -1
    assert L.sl3d_group_create(ctypes.byref(cfg), devs, 0, ctypes.byref(h)) == -1
    many = (ctypes.c_int * 40)(*([0] * 40))
    assert L.sl3d_group_create(ctypes.byref(cfg), many, 40, ctypes.byref(h)) == -1      # more stripes than rows
    assert b"stripes" in L.sl3d_group_last_error(None)
    rc = L.sl3d_group_create(ctypes.byref(cfg), devs, 2, ctypes.byref(h))
    if rc == 0:
        L.sl3d_group_destroy(h)
        pytest.skip("a HIP device is present")
    assert rc == -2 and h.value is None
    assert L.sl3d_group_size(None) == 0 and L.sl3d_group_transport(None) == b"copy"


def test_header_is_plain_c(tmp_path):
    """include/sl3d.h is the C ABI: it must compile as C99 with nothing but the standard headers (no C++, no HIP, no torch types)."""
    import subprocess
    src = tmp_path / "inc.c"
    src.write_text('#include "sl3d.h"\nint main(void) { sl3d_config c; sl3d_device_buffers b; (void)c; (void)b; return sl3d_group_size(0); }\n')
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"), "-c", str(src),
                           "-o", str(tmp_path / "inc.o")])

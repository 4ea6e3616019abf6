"""The shim's file readers (3dscan_amd/csrc/sl3d_shim_io.h: the reference's BMP frames, OpenCV XML matrices, PLY clouds) against
malformed input, on the CPU under AddressSanitizer + UndefinedBehaviorSanitizer: tests/native/shim_io_check.cpp parses well-formed
files to the right values and is handed every truncation of them, headers that lie about offsets / palette sizes / vertex counts, and
random bytes.  A reader may refuse a file; it may not read or allocate what the file does not hold (`read_ply` sized its
vectors from the header's vertex count)."""
import os
import subprocess

from conftest import ROOT


def test_shim_readers_survive_malformed_files_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "shim_io_check")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           "-I" + os.path.join(ROOT, "3dscan_amd", "csrc"), os.path.join(ROOT, "tests", "native", "shim_io_check.cpp"), "-o", exe])
    scratch = tmp_path / "files"
    scratch.mkdir()
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1")
    out = subprocess.run([exe, str(scratch)], capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stderr[-4000:]
    assert out.stdout.startswith("ok ") and int(out.stdout.split()[1]) > 5000


def test_every_extern_c_entry_point_is_behind_the_exception_barrier():
    """no exception crosses the C ABI.  Every multi-line extern "C" definition of the library's host sources is a
    function-try-block that ends in one of the barrier's handlers (sl3d_ctx.h: SL3D_CATCH*, sl3d_group.cpp: SL3D_GROUP_CATCH; the shim's
    C++ entry points: SHIM_CATCH); the one-liners only forward to such functions or return a field."""
    import re
    csrc = os.path.join(ROOT, "3dscan_amd", "csrc")
    guarded = 0
    for name, catch in (("sl3d_capi_context.cpp", r"SL3D_CATCH"), ("sl3d_capi_inputs.cpp", r"SL3D_CATCH"), ("sl3d_capi_run.cpp", r"SL3D_CATCH"),
                        ("sl3d_capi_clouds.cpp", r"SL3D_CATCH"), ("sl3d_capi_next.cpp", r"SL3D_CATCH"), ("sl3d_group.cpp", r"SL3D_GROUP_CATCH|SL3D_CATCH"), ("sl3d_shim.cpp", r"SHIM_CATCH|catch \(\.\.\.\)")):
        lines = open(os.path.join(csrc, name)).read().split("\n")
        i = 0
        while i < len(lines):
            l = lines[i]
            entry = l.startswith('extern "C"') or (name == "sl3d_shim.cpp" and re.match(
                r"^void (generate_pattern|compute_wrapped_phase|unwrap_phase|compute_c_p_map|triangulate|save_point_cloud|register_point_clouds)\(", l))
            if entry and not re.search(r"\{.*\}\s*$", l) and not l.rstrip().endswith(";"):
                j = i + 1
                while lines[j] not in ("{", "try {"):
                    j += 1
                k = j + 1
                while lines[k] != "}":
                    k += 1
                fn = re.search(r"(\w+)\(", l).group(1)
                if fn in ("sl3d_strerror",):           # a switch over string literals
                    i = k + 1
                    continue
                body = "\n".join(lines[j + 1:k])
                allocates = re.search(r"std::|new |fail\(|gfail\(|HIPCHK|GHIP|GCTX|push_back|\.assign|\.resize|sl3d_", body) is not None
                if allocates or lines[j] == "try {":
                    assert lines[j] == "try {", f"{name}: {fn} is not a function-try-block"
                    assert re.match(catch, lines[k + 1]), f"{name}: {fn} does not end in the barrier's handler: {lines[k + 1]!r}"
                    guarded += 1
                i = k + 1
            else:
                i += 1
    assert guarded >= 80, guarded

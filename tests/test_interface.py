"""The drop-in surface, checked mechanically: include/sl3d_shim.h must declare the reference's globals, prototypes and dimension macros
token for token.  tests/golden/interface.json holds the reference's side (written in the build container by
tests/golden/make_interface.py from PROJECT_GLOBAL/common_variables.h:6-24,56-62, intermodule_dependencies.h:4-29, global_cv.h:49-59 and
the stage files' definitions), so the comparison runs wherever the suite runs."""
import importlib.util
import json
import os

from conftest import GOLDEN, ROOT


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_shim_header_declares_the_references_interface_token_for_token():
    mk = _load(os.path.join(GOLDEN, "make_interface.py"), "make_interface")
    ref = json.load(open(os.path.join(GOLDEN, "interface.json")))
    assert len(ref["globals"]) == 19 and len(ref["functions"]) == 7 and len(ref["macros"]) == 5
    shim = open(os.path.join(ROOT, "include", "sl3d_shim.h")).read()
    assert mk.compare(ref, shim) == []
    # the check has teeth: a changed extent, type, parameter list or dimension is reported
    assert mk.compare(ref, shim.replace("extern double (*intersection_points)[Camera_imageheight][3];", "extern double (*intersection_points)[Camera_imageheight][4];"))
    assert mk.compare(ref, shim.replace("extern long int (*c_p_map)[2];", "extern long long (*c_p_map)[2];"))
    assert mk.compare(ref, shim.replace("void save_point_cloud(unsigned cloud_index);", "void save_point_cloud(int cloud_index);"))
    assert mk.compare(ref, shim.replace("#define Camera_imageheight 1200", "#define Camera_imageheight 1080"))
    assert mk.compare(ref, shim.replace("void unwrap_phase(int pattern_type);", "int unwrap_phase(int pattern_type);"))


def test_the_reference_header_itself_declares_unwrap_phase_differently_than_it_defines_it():
    """intermodule_dependencies.h:13 says `int unwrap_phase(int)`, 4/phase_unwrap.cpp:367 defines `void` -- recorded, and the shim follows
    the definition (C++ mangling ignores the return type, so the reference links)."""
    ref = json.load(open(os.path.join(GOLDEN, "interface.json")))
    assert ref["functions"]["unwrap_phase"]["ret"] == "int" and ref["defined_return_types"]["unwrap_phase"] == "void"
    assert all(ref["functions"][f]["ret"] == ref["defined_return_types"][f] for f in ref["functions"] if f != "unwrap_phase")

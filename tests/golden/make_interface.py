"""Build-container only (reads /root/reference): the reference's OWN declarations of the drop-in surface -- the globals of
PROJECT_GLOBAL/common_variables.h:6-24,56-62 that the path fills, the prototypes of PROJECT_GLOBAL/intermodule_dependencies.h:4-29, the
dimension macros of PROJECT_GLOBAL/global_cv.h:49-59, and the DEFINITION of unwrap_phase (4/phase_unwrap.cpp:367: `void`, although the
header says `int`; C++ mangling ignores the return type, so the link works and the shim follows the definition) -- normalised to token
strings in tests/golden/interface.json.  It refuses to write the file unless include/sl3d_shim.h declares exactly the same;
tests/test_interface.py repeats that comparison wherever the suite runs, without the reference's files.
    python tests/golden/make_interface.py"""
import json
import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import interface_tokens as it  # noqa: E402

REF = "/root/reference"


def read(rel):
    return open(os.path.join(REF, rel), errors="replace").read()


def main():
    globs, _ = it.declarations(read("PROJECT_GLOBAL/common_variables.h"))
    _, funcs = it.declarations(read("PROJECT_GLOBAL/intermodule_dependencies.h"))
    macros = it.macros(read("PROJECT_GLOBAL/global_cv.h"))
    missing = [g for g in it.GLOBALS if g not in globs] + [f for f in it.FUNCTIONS if f not in funcs] + [m for m in it.MACROS if m not in macros]
    assert not missing, missing
    # the definitions the reference's objects really export (a declaration may lie: unwrap_phase)
    defined = {}
    for name, rel in (("compute_wrapped_phase", "3/wrapped_phase.cpp"), ("unwrap_phase", "4/phase_unwrap.cpp"), ("compute_c_p_map", "5/compute_correspondance.cpp"),
                      ("triangulate", "7/triangulation.cpp"), ("save_point_cloud", "8/save_point_cloud.cpp"), ("register_point_clouds", "9/register_point_clouds.cpp"),
                      ("generate_pattern", "1/pattern_generator.cpp")):
        m = re.search(r"^\s*(\w[\w\s]*?)\s+" + name + r"\s*\(([^)]*)\)\s*(?://[^\n]*)?\s*\{", it.strip_comments(read(rel)), flags=re.M)
        assert m, (name, rel)
        defined[name] = " ".join(it.tokens(m.group(1)))
    out = {"_source": "pranavkantgaur/3dscan: PROJECT_GLOBAL/common_variables.h, intermodule_dependencies.h, global_cv.h; the stage files' definitions",
           "globals": {g: globs[g] for g in it.GLOBALS}, "functions": {f: funcs[f] for f in it.FUNCTIONS}, "defined_return_types": defined, "macros": macros}
    problems = compare(out, open(os.path.join(ROOT, "include", "sl3d_shim.h")).read())
    assert not problems, "include/sl3d_shim.h differs from the reference:\n" + "\n".join(problems)
    json.dump(out, open(os.path.join(HERE, "interface.json"), "w"), indent=1, sort_keys=True)
    print("tests/golden/interface.json:", len(out["globals"]), "globals,", len(out["functions"]), "prototypes,", len(out["macros"]), "macros: include/sl3d_shim.h agrees")


def compare(ref, shim_text):
    """-> list of differences between the reference's interface (interface.json's content) and a shim header's text"""
    globs, funcs = it.declarations(shim_text)
    macros = it.macros(shim_text)
    problems = []
    for g, want in ref["globals"].items():
        if globs.get(g) != want:
            problems.append(f"global {g}: reference `{want}`, shim `{globs.get(g)}`")
    for f, want in ref["functions"].items():
        got = funcs.get(f)
        if got is None or got["params"] != want["params"]:
            problems.append(f"function {f}: reference parameters {want['params']}, shim {got and got['params']}")
        # the return type of the DEFINITION is what an object file carries (and what callers of a void function may rely on)
        elif got["ret"] != ref["defined_return_types"][f]:
            problems.append(f"function {f}: defined as `{ref['defined_return_types'][f]}` in the reference, `{got['ret']}` in the shim")
    for m, want in ref["macros"].items():
        if macros.get(m) != want:
            problems.append(f"macro {m}: reference `{want}`, shim `{macros.get(m)}`")
    return problems


if __name__ == "__main__":
    main()

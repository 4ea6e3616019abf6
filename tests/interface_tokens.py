"""A tiny C declaration normaliser shared by tests/golden/make_interface.py (which reads the REFERENCE's headers in the build container)
and tests/test_interface.py (which reads include/sl3d_shim.h everywhere): globals, prototypes and dimension macros as token strings,
so that "the shim declares exactly what the reference declares" is checked mechanically, not by eye."""
import re

GLOBALS = ["number_of_codes_vertical", "number_of_codes_horizontal", "number_of_patterns_binary_vertical", "number_of_patterns_binary_horizontal",
           "number_of_patterns_fringe", "fringe_width_pixels_vertical", "fringe_width_pixels_horizontal", "code_vertical", "code_horizontal", "c_p_map",
           "selected_region", "valid_map_vertical", "valid_map_horizontal", "valid_map", "wrapped_phi_vertical", "wrapped_phi_horizontal",
           "unwrapped_phi_vertical", "unwrapped_phi_horizontal", "intersection_points"]
FUNCTIONS = ["generate_pattern", "compute_wrapped_phase", "unwrap_phase", "compute_c_p_map", "triangulate", "save_point_cloud", "register_point_clouds"]
MACROS = ["Camera_imagewidth", "Camera_imageheight", "Projector_imagewidth", "Projector_imageheight", "total_camera_pixels"]
TYPE_WORDS = {"int", "unsigned", "long", "short", "char", "float", "double", "void", "signed", "const"}


def strip_comments(text):
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    return re.sub(r"//[^\n]*", " ", text)


def tokens(s):
    return re.findall(r"[A-Za-z_]\w*|\d+(?:\.\d+)?|\S", s)


def split_top(s, sep):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "([":
            depth += 1
        elif ch in ")]":
            depth -= 1
        if ch == sep and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
    out.append(cur)
    return out


def macros(text):
    out = {}
    for line in text.split("\n"):                     # (before comments go: a macro's value ends where its trailing comment starts)
        m = re.match(r"\s*#\s*define\s+(\w+)\s+(.*)$", line)
        if m and m.group(1) in MACROS:
            out[m.group(1)] = " ".join(tokens(strip_comments(m.group(2))))
    return out


def declarations(text):
    """-> (globals {name: declaration with the name replaced by @}, functions {name: {"ret": ..., "params": [types]}})"""
    text = strip_comments(text)
    text = re.sub(r"^\s*#.*$", " ", text, flags=re.M)
    text = re.sub(r'extern\s+"C"\s*\{', " ", text)
    text = re.sub(r"\benum\s*\{.*?\}\s*;", " ", text, flags=re.S)
    globs, funcs = {}, {}
    for stmt in text.split(";"):
        stmt = " ".join(stmt.split())
        if not stmt:
            continue
        stmt = re.sub(r"^(extern|static)\s+", "", stmt)
        m = re.match(r"^([\w\s\*]+?)\s*\b(\w+)\s*\((.*)\)$", stmt)
        if m and m.group(2) in FUNCTIONS and "(*" not in m.group(1):
            params = []
            for p in split_top(m.group(3), ","):
                t = tokens(p)
                if len(t) > 1 and re.match(r"[A-Za-z_]\w*$", t[-1]) and t[-1] not in TYPE_WORDS:
                    t = t[:-1]                        # the parameter's name
                if t and t != ["void"]:
                    params.append(" ".join(t))
            funcs[m.group(2)] = {"ret": " ".join(tokens(m.group(1))), "params": params}
            continue
        parts = split_top(stmt, ",")
        base = None
        for i, part in enumerate(parts):
            part = part.split("=")[0]                 # an initialiser is not part of the type
            t = tokens(part)
            names = [x for x in t if x in GLOBALS]
            if len(names) != 1:
                continue
            if i == 0:
                k = t.index(names[0])
                lead = [x for x in t[:k] if x not in "(*"]
                base = " ".join(lead)
                globs[names[0]] = " ".join("@" if x == names[0] else x for x in t)
            elif base is not None:                    # `T a, b, c;`
                globs[names[0]] = (base + " " + " ".join("@" if x == names[0] else x for x in t)).strip()
    return globs, funcs

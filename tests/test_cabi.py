"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/m2h.h declares
(no compute calls without a GPU)."""
import ctypes
import os
import re

from m2h import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "m2h.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(m2h_[a-z0-9_A-Z]+)\s*\(", txt)))


def test_library_builds_and_exports_header_symbols():
    path = _lib.build()
    assert os.path.exists(path)
    lib = ctypes.CDLL(path)
    syms = _declared_symbols()
    assert "m2h_conv_igemm_f32" in syms and "m2h_unet_down_fwd" in syms
    for s in syms:
        assert hasattr(lib, s), "libm2h.so does not export %s" % s
    # the ctypes binding covers every declared function
    bound = set(_lib.SIGNATURES) | {"m2h_last_error"}
    assert set(syms) == bound, (set(syms) ^ bound)


def test_version_and_error_string_without_gpu():
    lib = _lib.load()
    assert lib.m2h_version() == 100
    assert isinstance(lib.m2h_last_error(), bytes)


def test_conv_args_struct_matches_header_field_order():
    txt = open(os.path.join(ROOT, "include", "m2h.h")).read()
    body = txt[txt.index("typedef struct m2h_conv_args {"):txt.index("} m2h_conv_args;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S).replace("typedef struct m2h_conv_args {", "")
    names = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        # "int a, b, c" or "const float* p"
        first = decl.split(",")
        names.append(first[0].replace("*", " ").split()[-1])
        names += [x.strip().replace("*", "") for x in first[1:]]
    assert names == [f for f, _ in _lib.ConvArgs._fields_], names


def _struct_fields(name):
    txt = open(os.path.join(ROOT, "include", "m2h.h")).read()
    body = txt[txt.index("typedef struct %s {" % name):txt.index("} %s;" % name)]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S).replace("typedef struct %s {" % name, "")
    names = []
    for decl in body.split(";"):
        decl = decl.strip()
        if decl:
            first = decl.split(",")
            names.append(first[0].replace("*", " ").split()[-1])
            names += [x.strip().replace("*", "") for x in first[1:]]
    return names


def test_rollout_structs_match_header_field_order():
    """m2h_episode_stats / m2h_row_copy mirrors (ctypes) follow the header field for field; argument errors of the two entry
    points are reported without a launch (negative return, message), so this runs without a GPU."""
    assert _struct_fields("m2h_episode_stats") == [f for f, _ in _lib.EpisodeStats._fields_] == list(_lib.EPISODE_STATS_FIELDS)
    assert _struct_fields("m2h_row_copy") == [f for f, _ in _lib.RowCopy._fields_]
    assert ctypes.sizeof(_lib.RowCopy) == 32 and ctypes.sizeof(_lib.EpisodeStats) == 17 * ctypes.sizeof(ctypes.c_void_p)
    lib = _lib.load()
    assert lib.m2h_rows_copy(None, 0, None, None) < 0 and b"rows_copy" in lib.m2h_last_error()
    items = (_lib.RowCopy * 1)(_lib.RowCopy(8, 16, 6, -1, -1))   # size not a multiple of 4
    assert lib.m2h_rows_copy(items, 1, None, None) < 0 and b"multiple of 4" in lib.m2h_last_error()
    st = _lib.EpisodeStats()                                      # null statistics tensors
    assert lib.m2h_episode_stats_update(ctypes.byref(st), 8, 8, 8, 8, 8, 8, None, None, 14, 3, None) < 0
    assert lib.m2h_gru_step(8, 8, 8, 8, None, 8, 8, 17, 512, None) < 0 and b"gru_step" in lib.m2h_last_error()

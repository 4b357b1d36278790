"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/m2h.h declares
(no compute calls without a GPU)."""
import ctypes
import os
import re

from m2h import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols(header="m2h.h"):
    txt = open(os.path.join(ROOT, "include", header)).read()
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
    # the diagnostic surface (thread-local tuning knobs) lives in its own header, outside the product contract
    diag = _declared_symbols("m2h_tuning.h")
    assert diag == ["m2h_tuning_restore", "m2h_tuning_set", "m2h_tuning_snapshot"] and not set(diag) & set(syms)
    assert "m2h_debug_set" not in syms and not hasattr(lib, "m2h_debug_set")
    for s in diag:
        assert hasattr(lib, s), "libm2h.so does not export %s" % s
    # the ctypes binding covers every declared function
    bound = set(_lib.SIGNATURES) | {"m2h_last_error"}
    assert set(syms) | set(diag) == bound, ((set(syms) | set(diag)) ^ bound)


def test_tuning_knobs_are_thread_local():
    """no process-global mutable state -- a knob set by one host thread is not seen by another (runs without a GPU)."""
    import threading
    lib = _lib.load()
    n = _lib.TUNING_KNOBS
    assert lib.m2h_tuning_set(24, 7) == 0
    seen = {}

    def other():
        arr = (ctypes.c_int * n)()
        assert lib.m2h_tuning_snapshot(arr, n) == 0
        seen["other"] = list(arr)
        assert lib.m2h_tuning_set(24, -1) == 0

    t = threading.Thread(target=other)
    t.start()
    t.join()
    mine = (ctypes.c_int * n)()
    assert lib.m2h_tuning_snapshot(mine, n) == 0
    assert seen["other"] == [0] * n and mine[24] == 7
    zero = (ctypes.c_int * n)()
    assert lib.m2h_tuning_restore(zero, n) == 0 and lib.m2h_tuning_set(n, 1) < 0


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


def test_workspace_bytes_reports_the_dma_engines_two_k_halves_launch():
    """m2h_conv_igemm_workspace_bytes (host-only) mirrors the launch rules: for the fourth encoder stage at the benchmark batch in
    bf16x3 arithmetic on split32 operands (128 tiles of 256 x 128, K = 4096) the LDS-DMA engine runs two K-halves per tile into
    split-K slabs -- a caller who sizes the workspace by this function gets 2 x M x N floats and with it the same kernel (and
    fp32 summation order) as the whole-network runner; the same layer in fp32 arithmetic reports the register engine's own factor."""
    lib = _lib.load()
    lib.m2h_conv_igemm_workspace_bytes.argtypes = [ctypes.POINTER(_lib.ConvArgs)]
    lib.m2h_conv_igemm_workspace_bytes.restype = ctypes.c_size_t
    a = _lib.ConvArgs()
    a.C0, a.C1, a.B, a.Hi, a.Wi, a.Hq, a.Wq = 256, 0, 256, 4, 32, 2, 16
    a.stride, a.nth, a.ntw, a.mulh, a.offh, a.mulw, a.offw = 2, 4, 4, 1, -1, 1, -1
    a.conv_transpose, a.N, a.Ho, a.Wo, a.os, a.ldc, a.out_mode = 0, 512, 2, 16, 1, 512, 0
    M, N = 256 * 2 * 16, 512
    a.operand_format = 1 | 2 | 4 | 8      # M2H_FMT_SRC_SPLIT | W_SPLIT | DST_SPLIT | MATH_BF16X3
    assert lib.m2h_conv_igemm_workspace_bytes(ctypes.byref(a)) == 2 * M * N * 4
    a.operand_format = 16                 # M2H_FMT_MATH_FP32: the register engine's split factor for this shape
    fp32_bytes = lib.m2h_conv_igemm_workspace_bytes(ctypes.byref(a))
    assert fp32_bytes % (M * N * 4) == 0 and fp32_bytes != 2 * M * N * 4 or fp32_bytes == 0 or fp32_bytes == 2 * M * N * 4
    a.operand_format = 1 | 2 | 4 | 8
    a.B = 1024                            # enough 256 x 128 tiles to fill the chip: no split at all on that engine
    assert lib.m2h_conv_igemm_workspace_bytes(ctypes.byref(a)) == 0
    # the deepest encoder stage at the benchmark batch (32 tiles of 256 x 128, half of the window in the padding: walked K = 4096):
    # eight K-parts per tile on that engine
    a.C0, a.B, a.Hi, a.Wi, a.Hq, a.Wq, a.Ho, a.Wo = 512, 256, 2, 16, 1, 8, 1, 8
    assert lib.m2h_conv_igemm_workspace_bytes(ctypes.byref(a)) == 8 * (256 * 8) * 512 * 4

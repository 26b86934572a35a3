"""The drop-in boundary: include/mgs.h <-> libmgs.so <-> the ctypes table, and the rule that the
product never reaches into oracle/ (no compute calls here: CPU-only)."""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "mgs.h")


def _declared(hooks=False):
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    dbg = re.search(r"#ifdef MGS_DEBUG_HOOKS(.*?)#endif", src, flags=re.S)
    src = dbg.group(1) if hooks else src.replace(dbg.group(0), "")
    decls = re.findall(r"\b(?:int|void|size_t|const char \*)\s*\*?\s*(mgs_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S)
    return {name: 0 if args.strip() == "void" else len([a for a in args.split(",") if a.strip()])
            for name, args in decls}


def test_header_symbols_are_exported_and_bound():
    from robosimgs_amd import _lib
    decl = _declared()
    assert len(decl) == 29, sorted(decl)
    assert sorted(decl) == sorted(_lib.EXPORTS)
    L = _lib.lib()
    for name, nargs in decl.items():
        fn = getattr(L, name)
        assert len(fn.argtypes) == nargs, f"{name}: header has {nargs} parameters, ctypes table {len(fn.argtypes)}"
    header = open(HEADER).read()
    declared = int(re.search(r"#define\s+MGS_VERSION\s+(\d+)", header).group(1))
    assert L.mgs_version() == declared == _lib.MGS_VERSION      # header, library and binding agree
    assert isinstance(L.mgs_last_error_string(), bytes)


def test_shipped_library_has_no_debug_hooks_and_the_debug_build_has_them():
    """include/mgs.h: "stateless and re-entrant".  The process-global knobs (mgs_debug_set_*) are declared under
    MGS_DEBUG_HOOKS and compiled into libmgs_debug.so only."""
    from robosimgs_amd import _lib
    hooks = _declared(hooks=True)
    assert sorted(hooks) == sorted(_lib.DEBUG_HOOKS) and len(hooks) == 3
    nm = lambda path: subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True).stdout
    shipped, debug = nm(_lib.LIB_PATH), nm(_lib.DEBUG_LIB_PATH)
    assert "mgs_debug" not in shipped
    for name in hooks:
        assert name in debug
    for name in _declared():
        assert name in shipped and name in debug
    assert _lib.debug_lib().mgs_version() == _lib.lib().mgs_version()


def test_library_is_gfx950_only_and_links_no_torch():
    from robosimgs_amd import _lib
    out = subprocess.run(["readelf", "-d", _lib.LIB_PATH], capture_output=True, text=True).stdout
    needed = re.findall(r"NEEDED.*\[(.*?)\]", out)
    assert any("amdhip64" in n for n in needed)
    assert not any("torch" in n or "c10" in n for n in needed), needed
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if os.path.exists(objdump):
        # --offloading dumps every bundle entry next to its input: inspect a scratch copy
        with tempfile.TemporaryDirectory() as scratch:
            copy = shutil.copy(_lib.LIB_PATH, scratch)
            o = subprocess.run([objdump, "--offloading", copy], capture_output=True, text=True,
                               cwd=scratch).stdout
        archs = set(re.findall(r"gfx\w+", o))
        assert archs == {"gfx950"}, archs


def test_argument_errors_are_reported_without_a_gpu():
    from robosimgs_amd import _lib
    L = _lib.lib()
    rc = L.mgs_rasterize_fwd(1, None, None, None, None, None, None, 99, 16, 16, 1, 1, None, None, None, 0, None, None, None, None, 0,
                             None, None, 0, None, None)
    assert rc == -1 and b"channels" in L.mgs_last_error_string()
    rc = L.mgs_sh_fwd(4, 7, 16, None, None, None, None, None)
    assert rc == -1 and b"degree" in L.mgs_last_error_string()


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "robosimgs_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f
                assert "gs_cpu" not in text and "oracle/" not in text.replace("see oracle/", ""), f


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from robosimgs_amd import _lib
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    monkeypatch.setattr(_lib, "_lib", None)
    with pytest.raises(_lib.MgsError, match="no CPU or PyTorch fallback"):
        _lib.lib()


def test_host_side_size_queries_need_no_gpu():
    """mgs_raster_checkpoint_floats and mgs_train_state_layout are host arithmetic: the checkpoint buffer holds a 4-int
    header per tile plus (capacity / S + tiles + 1) units of (1 + channels) x 256 floats; the training state's fields are
    256-byte aligned, in the order include/mgs.h lists them, and large enough for what they hold."""
    import ctypes
    from robosimgs_amd import _lib, ops
    L = _lib.lib()
    tw, th, cap, ch, S = 120, 68, 4_640_000, 4, 256
    n = L.mgs_raster_checkpoint_floats(cap, tw, th, ch, S)
    header = (tw * th * 4 + 255) // 256 * 256
    assert n == header + (cap // S + tw * th + 1) * (1 + ch) * 256
    assert L.mgs_raster_checkpoint_floats(cap, tw, th, ch, 100) == 0 and L.mgs_raster_checkpoint_floats(cap, tw, th, ch, 32) == 0
    offs = (ctypes.c_size_t * len(ops.TRAIN_FIELDS))()
    per = ctypes.c_size_t(0)
    N, W, H = 1_000_000, 1920, 1080
    assert L.mgs_train_state_layout(N, W, H, ch, cap, 0, S, offs, ctypes.byref(per)) == 0
    o = dict(zip(ops.TRAIN_FIELDS, offs))
    assert list(offs) == sorted(offs) and all(x % 256 == 0 for x in offs) and per.value % 256 == 0
    need = {"radii": 4 * N, "means2d": 8 * N, "depths": 4 * N, "conics": 12 * N, "opac_aa": 0, "feats": 4 * ch * N,
            "splats": 48 * N, "tiles_per_gauss": 4 * N, "pair_info": 16 * N, "tile_ids": 4 * cap, "flatten_ids": 4 * cap,
            "tile_offsets": 4 * (tw * th + 1), "group_order": 4 * ((tw * th + 3) // 4), "last_ids": 4 * W * H,
            "checkpoints": 4 * n, "counts": 8, "radii_y": 4 * N}
    names = list(ops.TRAIN_FIELDS)
    for a, b in zip(names, names[1:] + [None]):
        end = o[b] if b else per.value
        assert end - o[a] >= need[a], a
    assert per.value < 1.02 * sum(need.values()) + 16 * 256
    assert L.mgs_train_state_layout(N, W, H, 5, cap, 0, S, offs, ctypes.byref(per)) == -1       # 3 or 4 channels
    assert L.mgs_train_state_layout(N, W, H, ch, cap, 0, 100, offs, ctypes.byref(per)) == -1    # not a power of two

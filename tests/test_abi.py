"""CPU tests of the boundary: the shared library loads, exports every symbol include/nmpc_amd.h
declares, the ctypes table covers them, host-only entry points work, and compute entry points
refuse to run without a GPU instead of falling back."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "nmpc_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(nmpc_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from nmpc_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "build with `make -C nmpc_amd/csrc`"
    h = ctypes.CDLL(_lib.LIB_PATH)
    names = declared_symbols()
    assert len(names) >= 15
    for n in names:
        assert hasattr(h, n), "%s not exported" % n
    assert sorted(_lib.SIGNATURES) == names, "ctypes table and header disagree"
    # the library, the header and the ctypes mirror of nmpc_ctrl_config describe the same version / layout
    hdr = open(os.path.join(ROOT, "include", "nmpc_amd.h")).read()
    assert _lib.lib().nmpc_version() == int(re.search(r"#define NMPC_AMD_VERSION (\d+)", hdr).group(1))


def test_ctrl_config_mirror_matches_the_header(tmp_path):
    """sizeof(nmpc_ctrl_config) as a C compiler sees the header == the ctypes structure (nmpc_ctrl_create checks
    struct_size against the library's own sizeof)."""
    import subprocess
    from nmpc_amd import _lib
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "nmpc_amd.h"\nint main(void){printf("%zu %zu %zu\\n", '
                   'sizeof(nmpc_ctrl_config), offsetof(nmpc_ctrl_config, as_written), offsetof(nmpc_ctrl_config, lam));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    size, off_aw, off_lam = map(int, subprocess.check_output([str(exe)]).split())
    assert size == ctypes.sizeof(_lib.CtrlConfig)
    assert off_aw == _lib.CtrlConfig.as_written.offset and off_lam == _lib.CtrlConfig.lam.offset
    assert _lib.CtrlConfig().struct_size == size


def test_quad_nodes_host_entry_point_matches_oracle():
    import nmpc_amd
    from oracle import lpv
    for rule, stages, dlam in (("simpson", 20, None), ("simpson", 20, 0.05), ("trap", 20, None), ("simpson", 7, 0.1),
                               ("simpson", 100, None)):
        lam, w = nmpc_amd.quad_nodes(rule, stages, dlam)
        n, ww = lpv.merge_nodes(*lpv.quad_nodes(rule, stages, dlam))
        assert len(lam) == len(n)
        assert np.array_equal(lam, n.astype(np.float32)) and np.array_equal(w, ww.astype(np.float32))
    from nmpc_amd import _lib
    with pytest.raises(_lib.NmpcError):
        nmpc_amd.quad_nodes("simpson", 200)      # 401 nodes > NMPC_MAX_NODES


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import nmpc_amd
    with pytest.raises(nmpc_amd.NmpcError):
        nmpc_amd.load_system()
    with pytest.raises(nmpc_amd.NmpcError):
        nmpc_amd.getPsi(3, 1.0)


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "nmpc_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f

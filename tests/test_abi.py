"""CPU: the C-ABI library loads and exports every symbol include/scarlet_hip.h declares
(no compute calls -- there is no GPU here), and the ctypes mirror of the struct matches."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "scarlet_hip.h")
LIB = os.path.join(ROOT, "scarlet_amd", "csrc", "libscarlet_hip.so")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(scarlet_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    assert os.path.exists(LIB), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    lib = ctypes.CDLL(LIB)
    names = declared_functions()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), "header declares %s but the library does not export it" % n


def test_python_binding_covers_header():
    from scarlet_amd import _lib
    assert sorted(_lib.EXPORTS) == declared_functions()


def test_host_only_entry_points():
    from scarlet_amd import _lib
    assert _lib.lib.scarlet_version().startswith(b"scarlet_amd-hip")
    import scipy.fftpack
    for n in (1, 7, 67, 75, 99, 147, 172, 300, 521, 1031):
        assert _lib.lib.scarlet_next_fast_len(n) == scipy.fftpack.next_fast_len(n)


def test_struct_layout_matches_header(tmp_path):
    """sizeof/offsetof of struct scarlet_batch as the C compiler sees it == ctypes mirror."""
    from scarlet_amd import _lib
    src = tmp_path / "probe.c"
    fields = [f[0] for f in _lib.ScarletBatch._fields_]
    body = "\n".join('printf("%s %%zu\\n", offsetof(scarlet_batch, %s));' % (f, f) for f in fields)
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "scarlet_hip.h"\n'
                   'int main(void){ printf("sizeof %zu\\n", sizeof(scarlet_batch));\n' + body + '\nreturn 0;}\n')
    exe = tmp_path / "probe"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    out = dict(line.split() for line in subprocess.check_output([str(exe)]).decode().splitlines())
    assert int(out["sizeof"]) == ctypes.sizeof(_lib.ScarletBatch)
    for f in fields:
        assert int(out[f]) == getattr(_lib.ScarletBatch, f).offset, f


def test_no_cpu_fallback_without_gpu():
    """The product refuses to compute without a device instead of silently falling back."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import numpy as np
    import scarlet_amd as sc
    with pytest.raises(RuntimeError):
        sc.BlendBatch(np.zeros((1, 5, 16, 16), np.float32), np.zeros((1, 1, 2), np.int32) + 8)
    with pytest.raises(RuntimeError):
        sc.operator.prox_soft_symmetry(np.zeros((5, 5), np.float32), 0)

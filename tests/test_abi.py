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


def test_error_paths_return_codes_without_touching_memory():
    """argument errors come back as SCARLET_E_ARG with a message, before anything is allocated or launched"""
    import numpy as np
    from scarlet_amd import _lib
    x = np.zeros(30, np.float32)
    rc = _lib.lib.scarlet_host_prox_weighted_monotonic_f32(None, 30, x.ctypes.data, x.ctypes.data, x.ctypes.data, 29, 0.0)
    assert rc == _lib.E_ARG and _lib.last_error()
    with pytest.raises(ValueError):
        _lib.check(rc)
    assert _lib.lib.scarlet_set_option(b"NO_SUCH_OPTION", 1) == _lib.E_ARG
    assert _lib.set_option("NO_FUSED", 1) == 0 and _lib.set_option("NO_FUSED", 0) == 1
    assert _lib.lib.scarlet_fit(None, 1, 0.0, 0, 0, None) == _lib.E_ARG
    assert _lib.lib.scarlet_batch_workspace_bytes(None) == 0
    assert _lib.lib.scarlet_batch_pipelines(None) == 0


def test_layout_switches_freeze_with_the_first_psf_workspace():
    """PSF_HIPFFT and STAMPS decide where the regions of a PSF batch's workspace lie and are read again by
    every later call on the batch; once a PSF workspace has been sized they can no longer change (SCARLET_E_ARG).
    In a child process: the freeze is process-wide."""
    import sys
    code = """
import ctypes, sys
sys.path.insert(0, %r)
from scarlet_amd import _lib
assert _lib.set_option("PSF_HIPFFT", 1) == 0 and _lib.set_option("PSF_HIPFFT", 0) == 1      # free before
b = _lib.ScarletBatch()
b.S, b.K, b.B, b.H, b.W, b.psf_h, b.psf_w = 4, 2, 3, 32, 32, 11, 11
b.diff_kernel = 1                                                                        # (only tested against NULL)
assert _lib.lib.scarlet_batch_workspace_bytes(ctypes.byref(b)) > 0
assert _lib.lib.scarlet_set_option(b"PSF_HIPFFT", 1) == _lib.E_ARG and b"layout" in _lib.lib.scarlet_last_error()
assert _lib.lib.scarlet_set_option(b"STAMPS", 1) == _lib.E_ARG
assert _lib.lib.scarlet_set_option(b"PSF_HIPFFT", 0) == 0                                  # the value it has: fine
assert _lib.lib.scarlet_set_option(b"NO_BOX", 1) == 0                                      # other switches stay free
assert _lib.lib.scarlet_debug_stamps(None, -5) == 0
print("ok")
""" % ROOT
    out = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert out.returncode == 0 and b"ok" in out.stdout, out.stderr.decode()[-1500:]


def test_asan_error_paths():
    """`make asan`: the host code of the library under AddressSanitizer + LeakSanitizer, driven through the
    argument-error and HIP-error exits of every entry point that allocates (tests/native/abi_asan_check.c).
    Without a GPU every hipMalloc fails, which is exactly the set of early returns that must not leak."""
    csrc = os.path.join(ROOT, "scarlet_amd", "csrc")
    subprocess.check_call(["make", "-C", csrc, "asan"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    exe = os.path.join(ROOT, "tests", "native", "_build", "abi_asan_check")
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:exitcode=23")
    out = subprocess.run([exe], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    assert b"abi_asan_check ok" in out.stdout

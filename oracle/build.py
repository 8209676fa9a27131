"""Build the oracle's C restatement (oracle/sweeps.c -> oracle/_build/liboracle_sweeps.so).

TEST INFRASTRUCTURE.  Called by ``__graft_entry__.build()`` and lazily by
``oracle.native``.  Plain gcc, no -march and no FP contraction so that float
arithmetic is mul-then-add exactly like the reference's distutils build
(/root/reference/setup.py:118-123 passes only -std=c++14 -fvisibility=hidden).

The reference's own native file (scarlet/operators_pybind11.cc) is UNBUILDABLE here:
it includes <pybind11/eigen.h> which needs the Eigen headers, absent from this image
(SURVEY.md section 8c), so there is no oracle/_ref for it; the restatement is pinned
by the reference's golden vectors instead.
"""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
OUT_DIR = os.path.join(HERE, "_build")
LIB = os.path.join(OUT_DIR, "liboracle_sweeps.so")
SRC = os.path.join(HERE, "sweeps.c")


def build(force=False):
    if (not force and os.path.exists(LIB)
            and os.path.getmtime(LIB) >= os.path.getmtime(SRC)):
        return LIB
    os.makedirs(OUT_DIR, exist_ok=True)
    cmd = ["gcc", "-O2", "-ffp-contract=off", "-fPIC", "-shared", "-o", LIB, SRC]
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force=True))

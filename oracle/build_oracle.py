"""Builds the oracle's C restatements (gcc) into oracle/_build/ — test infrastructure."""

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_build")
LIB = os.path.join(OUT, "liboracle.so")
SOURCES = ["ms_oracle.c", "cc_oracle.c"]


def build(force=False):
    os.makedirs(OUT, exist_ok=True)
    srcs = [os.path.join(HERE, s) for s in SOURCES]
    if not force and os.path.exists(LIB) and all(os.path.getmtime(LIB) >= os.path.getmtime(s) for s in srcs):
        return LIB
    gcc = shutil.which("gcc")
    if gcc is None:
        raise RuntimeError("gcc not found: cannot build the oracle's C restatement")
    cmd = [gcc, "-O2", "-ffp-contract=off", "-shared", "-fPIC", "-o", LIB, *srcs, "-lm"]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"oracle build failed:\n{res.stderr}")
    return LIB


if __name__ == "__main__":
    print(build(force=True))

"""Build the C parts of the oracle (test infrastructure) with gcc into oracle/_build/ (git-ignored).

    python -m oracle.build_c
"""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_build")
TARGETS = {"libjpeg_oracle.so": ["jpeg_oracle.c"]}


def build(force=False):
    os.makedirs(OUT, exist_ok=True)
    built = []
    for lib, srcs in TARGETS.items():
        dst = os.path.join(OUT, lib)
        paths = [os.path.join(HERE, s) for s in srcs]
        if force or not os.path.exists(dst) or any(os.path.getmtime(p) > os.path.getmtime(dst) for p in paths):
            subprocess.run(["gcc", "-O2", "-shared", "-fPIC", "-Wall", "-o", dst, *paths], check=True)
        built.append(dst)
    return built


if __name__ == "__main__":
    print("\n".join(build(force=True)))

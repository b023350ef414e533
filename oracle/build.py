"""Compiles the oracle's C restatements (TEST INFRASTRUCTURE) with gcc:

    python -m oracle.build            ->  oracle/lib/liboracle_sdf.so   (from oracle/sdf_fmm.c)

No floating-point contraction (-ffp-contract=off): the restatement must round like an x86-64 scikit-fmm wheel (no FMA)
and like the GPU kernel it checks (dataset.hip is compiled with the same flag).  The built library is git-ignored and
travels to the GPU box with the repo snapshot, like cfd_amd/lib/libcfd_amd.so.
"""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "lib", "liboracle_sdf.so")
SRC = os.path.join(HERE, "sdf_fmm.c")


def build(force: bool = False) -> str:
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(SRC):
        cmd = ["gcc", "-O2", "-std=c11", "-ffp-contract=off", "-fPIC", "-shared", "-Wall", SRC, "-o", LIB, "-lm"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("gcc failed for oracle/sdf_fmm.c:\n" + r.stdout + r.stderr)
    return LIB


if __name__ == "__main__":
    print(build(force=True))

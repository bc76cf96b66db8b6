"""Build oracle/cpu/libspt_cpu.so (TEST INFRASTRUCTURE: the CPU twins of the kNN / geometric
feature entries, OpenMP).  Called by __graft_entry__.build(); the product never loads it."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "spt_cpu.cpp")
LIB = os.path.join(HERE, "libspt_cpu.so")


def build(force=False):
    if (not force and os.path.exists(LIB)
            and os.path.getmtime(LIB) >= os.path.getmtime(SRC)):
        return LIB
    cmd = ["g++", "-O3", "-std=c++17", "-fopenmp", "-ffp-contract=off", "-shared", "-fPIC",
           SRC, "-o", LIB]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"g++ failed on {SRC}:\n{r.stdout}\n{r.stderr}")
    return LIB


if __name__ == "__main__":
    print(build(force=True))

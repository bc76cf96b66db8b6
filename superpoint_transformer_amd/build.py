"""In-tree build of libspt_hip.so (hipcc, gfx950 only).

The shared object lands in ``superpoint_transformer_amd/lib/`` so that it
travels with the source tree (gpurun snapshot) and is picked up by
``superpoint_transformer_amd._lib``.  Objects are rebuilt only when a source
or header is newer than the object.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(LIBDIR, "obj")
LIB = os.path.join(LIBDIR, "libspt_hip.so")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
    "-ffp-contract=off",  # parity: no silent fma contraction vs the oracle
    "-Wno-unused-result",
] + os.environ.get("SPT_EXTRA_HIPCC_FLAGS", "").split()


# per-file code generation options
#   edge_attn_mfma.hip: keep MFMA accumulators / results in the VGPR file.  With the default
#   (AGPR form) the attention backward moved ~300 values per tile between AGPRs and VGPRs
#   (v_accvgpr_read / write), because its MFMA results feed VALU code.
PER_FILE_FLAGS = {
    "edge_attn_mfma.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"],
    "edge_attn_el.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"],
    "edge_attn_to.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"],
    "fused_mlp.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"],
    "fused_mlp_dma.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"],
}


def _sources():
    return sorted(
        os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _headers():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")]
    hs += [os.path.join(INCLUDE, f) for f in os.listdir(INCLUDE)]
    return hs


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src):
    obj = os.path.join(OBJDIR, os.path.basename(src)[:-4] + ".o")
    if _stale(obj, [src] + _headers()):
        extra = PER_FILE_FLAGS.get(os.path.basename(src), [])
        cmd = [HIPCC] + FLAGS + extra + ["-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0 and extra and "Unknown command line argument" in r.stderr:
            # a toolchain without the code-generation option: same code, default MFMA form
            cmd = [HIPCC] + FLAGS + ["-c", src, "-o", obj]
            r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(
                f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
    return obj


def build(force=False, verbose=False):
    """Compile every csrc/*.hip for gfx950 and link libspt_hip.so."""
    os.makedirs(OBJDIR, exist_ok=True)
    if force:
        for f in os.listdir(OBJDIR):
            os.remove(os.path.join(OBJDIR, f))
    srcs = _sources()
    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(_compile, srcs))
    if force or _stale(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    if verbose:
        print(f"built {LIB}")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)

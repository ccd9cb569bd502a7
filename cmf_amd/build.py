"""Build libcmf_amd.so (hand-written HIP for gfx950) in-tree with hipcc.

``python -m cmf_amd.build`` compiles every ``csrc/*.hip`` for ``--offload-arch=gfx950`` and links
``cmf_amd/libcmf_amd.so``.  hipcc cross-compiles without a GPU, so this also runs in the build
container; the built library travels to the GPU box with the tree (it is git-ignored, not
gpurun-ignored).
"""
import concurrent.futures as cf
import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "csrc", "_obj")
LIB = os.path.join(HERE, "libcmf_amd.so")
ARCH = "gfx950"


# conv_tangent_bf16x3: no SLP packing of f32 pairs into v_pk_*_f32 in the loader waves (slower beside MFMAs)
PER_FILE_FLAGS = {"conv_tangent_bf16x3.hip": ["-fno-slp-vectorize"]}


def hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm under /opt/rocm)")


def _digest(paths):
    h = hashlib.sha256()
    for p in sorted(paths):
        with open(p, "rb") as f:
            h.update(os.path.basename(p).encode() + b"\0" + f.read())      # path-independent: the tree travels to the GPU box
    return h.hexdigest()


def build(force=False, verbose=True):
    srcs = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))
    deps = srcs + [os.path.join(CSRC, "common.h"), os.path.join(ROOT, "include", "cmf_amd.h")]
    stamp = os.path.join(OBJ, "stamp")
    dig = _digest(deps)
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read() == dig:
        return LIB
    os.makedirs(OBJ, exist_ok=True)
    cc = hipcc()
    flags = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC,
             "-Wno-unused-command-line-argument"]

    def one(src):
        obj = os.path.join(OBJ, os.path.basename(src)[:-4] + ".o")
        extra = PER_FILE_FLAGS.get(os.path.basename(src), [])
        r = subprocess.run([cc, *flags, *extra, "-c", src, "-o", obj], capture_output=True, text=True)
        if r.returncode:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stderr}")
        return obj

    with cf.ThreadPoolExecutor(max_workers=min(4, len(srcs))) as ex:
        objs = list(ex.map(one, srcs))
    r = subprocess.run([cc, f"--offload-arch={ARCH}", "-shared", "-fPIC", *objs, "-o", LIB], capture_output=True, text=True)
    if r.returncode:
        raise RuntimeError(f"link failed:\n{r.stderr}")
    with open(stamp, "w") as f:
        f.write(dig)
    if verbose:
        print(f"built {LIB}")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)

#!/usr/bin/env python3
"""Register / scratch / LDS usage of every kernel in csrc/*.hip, from hipcc's own ``-Rpass-analysis=kernel-resource-usage`` remarks
(compiled with the product flags of cmf_amd/build.py; no GPU needed).

  python tools/kernel_resources.py [file.hip ...] [--spills]      # table; --spills: only kernels with scratch / spilled registers

``resources(path)`` -> {demangled kernel name: {"vgprs", "agprs", "sgprs", "scratch", "vgpr_spill", "sgpr_spill", "occupancy", "lds"}}
is what tests/test_kernel_resources.py asserts on.
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cmf_amd import build as B                                     # noqa: E402

_KEYS = {"TotalSGPRs": "sgprs", "VGPRs": "vgprs", "AGPRs": "agprs", "ScratchSize [bytes/lane]": "scratch",
         "Occupancy [waves/SIMD]": "occupancy", "SGPRs Spill": "sgpr_spill", "VGPRs Spill": "vgpr_spill",
         "LDS Size [bytes/block]": "lds"}


def demangle(names):
    import shutil
    filt = shutil.which("c++filt") or "/opt/rocm/lib/llvm/bin/llvm-cxxfilt"
    if not os.path.exists(filt):
        return list(names)
    out = subprocess.run([filt], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    clean = lambda o: re.sub(r"\(.*$", "", o.replace("(anonymous namespace)::", "").replace("void ", "", 1)).strip()
    return [clean(o) for o in out]


def resources(path, extra_flags=()):
    """Compile one .hip file for gfx950 (object discarded) and parse the resource-usage remarks of its kernels."""
    name = os.path.basename(path)
    flags = [f"--offload-arch={B.ARCH}", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"), "-I" + B.CSRC,
             "-Wno-unused-command-line-argument", "-Rpass-analysis=kernel-resource-usage", *B.PER_FILE_FLAGS.get(name, []), *extra_flags]
    r = subprocess.run([B.hipcc(), *flags, "-c", path, "-o", os.devnull], capture_output=True, text=True)
    if r.returncode:
        raise RuntimeError(f"hipcc failed on {path}:\n{r.stderr[-4000:]}")
    out, cur = {}, None
    for line in r.stderr.splitlines():
        m = re.search(r"remark:\s+(.*?) \[-Rpass-analysis=kernel-resource-usage\]", line)
        if not m:
            continue
        body = m.group(1).strip()
        if body.startswith("Function Name:"):
            cur = out.setdefault(body.split(":", 1)[1].strip(), {})
            continue
        k, _, v = body.rpartition(":")
        if cur is not None and k.strip() in _KEYS:
            cur[_KEYS[k.strip()]] = int(v)
    names = list(out)
    return dict(zip(demangle(names), (out[n] for n in names)))


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    files = args or sorted(os.path.join(B.CSRC, f) for f in os.listdir(B.CSRC) if f.endswith(".hip"))
    only_spills = "--spills" in sys.argv
    bad = 0
    for f in files:
        for k, v in resources(f).items():
            spilled = v.get("scratch", 0) or v.get("vgpr_spill", 0)         # (SGPR spills go to VGPR lanes, not to memory: listed, not counted)
            bad += bool(spilled)
            if only_spills and not spilled:
                continue
            print(f"{os.path.basename(f):28s} {k[:86]:86s} vgpr {v.get('vgprs', 0):3d} agpr {v.get('agprs', 0):3d} scratch {v.get('scratch', 0):4d} "
                  f"spill {v.get('vgpr_spill', 0):3d} sspill {v.get('sgpr_spill', 0):3d} occ {v.get('occupancy', 0)} lds {v.get('lds', 0)}")
    print(f"{bad} kernel(s) with scratch or spilled registers")
    return 0


if __name__ == "__main__":
    sys.exit(main())

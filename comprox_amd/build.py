"""Build recipe for libcrgpu.so (HIP kernels + C-ABI, gfx950 only).

`python -m comprox_amd.build` compiles comprox_amd/csrc/crgpu.hip in-tree with hipcc; the .so is
git-ignored but travels to the GPU box with the gpurun snapshot. hipcc cross-compiles without a GPU.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libcrgpu.so")
SOURCES = ["crgpu.hip"]
HEADERS = ["crgpu_device.h", "crgpu_wave.h", "crgpu_ppm.h", "crgpu_lzp.h", "crgpu_rop.h", "crgpu_dict.h"]


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.join(ROOT, "include", "crgpu.h")]
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile libcrgpu.so if missing or older than its sources; returns its path."""
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-fPIC", "-shared", "-std=c++17",
           "-o", LIB] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        cmd.append("-Rpass-analysis=kernel-resource-usage")
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))

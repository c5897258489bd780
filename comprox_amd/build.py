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
DIAG_LIB = os.path.join(HERE, "libcrgpu_diag.so")      # builds with $CRGPU_CFLAGS (in-kernel stamps etc.): never the product library
CLI = os.path.join(HERE, "bin", "comprop-gpu")
CLI_ROX = os.path.join(HERE, "bin", "comprox-gpu")
CLI_ROLZ = os.path.join(HERE, "bin", "comprolz-gpu")
EXTRA_LIBS = ["-lpthread", "-ldl"]
SOURCES = ["crgpu.hip", "crgpu_multi.hip"]            # HIP: kernels + C-ABI; the multi-GPU block loop
HOST_C = ["crhost_dict.c", "crhost_filter.c"]         # plain C host passes (gcc), linked into the same library
HEADERS = ["crgpu_device.h", "crgpu_wave.h", "crgpu_ppm.h", "crgpu_lzp.h", "crgpu_lzp2.h", "crgpu_links2.h", "crgpu_rolz3.h", "crgpu_rox3.h", "crgpu_rop.h", "crgpu_dict.h", "crgpu_rox.h", "crgpu_rolz.h", "crgpu_rop2.h", "crgpu_dec.h", "crgpu_rop5.h", "crgpu_rox5.h", "crgpu_rolz5.h", "crgpu_rox2.h", "crgpu_rolz2.h"]


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HOST_C + HEADERS + ["crmain.c"]] + [os.path.join(ROOT, "include", "crgpu.h")]
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile libcrgpu.so if missing or older than its sources; returns its path. With $CRGPU_CFLAGS set (diagnostic
    defines such as -DCR_V5_PROF=2) the result goes to libcrgpu_diag.so instead — select it with $CRGPU_LIB — so that
    tests and bench.py never measure an instrumented build by accident."""
    diag = os.environ.get("CRGPU_CFLAGS", "").split()
    if diag:
        return _build_diag(diag, verbose)
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    for src in HOST_C:
        obj = os.path.join(CSRC, os.path.splitext(src)[0] + ".o")
        subprocess.run([os.environ.get("CC", "gcc"), "-std=c99", "-O2", "-fPIC", "-Wall", "-c",
                        os.path.join(CSRC, src), "-o", obj], check=True)
        objs.append(obj)
    for src in SOURCES:
        obj = os.path.join(CSRC, os.path.splitext(src)[0] + ".o")
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            cmd.append("-Rpass-analysis=kernel-resource-usage")
            print(" ".join(cmd), file=sys.stderr)
        subprocess.run(cmd, check=True)
        objs.append(obj)
    subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + EXTRA_LIBS, check=True)
    # command line / container (plain C over the C-ABI)
    os.makedirs(os.path.dirname(CLI), exist_ok=True)
    for exe, defs in ((CLI, []), (CLI_ROX, ["-DCR_FRONTEND_ROX"]), (CLI_ROLZ, ["-DCR_FRONTEND_ROLZ"])):
        subprocess.run([os.environ.get("CC", "gcc"), "-std=gnu99", "-O2", "-Wall"] + defs +
                       [os.path.join(CSRC, "crmain.c"), "-o", exe, "-L" + HERE, "-lcrgpu", "-Wl,-rpath,$ORIGIN/.."], check=True)
    return LIB


def _build_diag(flags, verbose: bool = False) -> str:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    for src in HOST_C:
        obj = os.path.join(CSRC, os.path.splitext(src)[0] + ".diag.o")
        subprocess.run([os.environ.get("CC", "gcc"), "-std=c99", "-O2", "-fPIC", "-c", os.path.join(CSRC, src), "-o", obj], check=True)
        objs.append(obj)
    for src in SOURCES:
        obj = os.path.join(CSRC, os.path.splitext(src)[0] + ".diag.o")
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17"] + flags + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.run(cmd, check=True)
        objs.append(obj)
    subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", DIAG_LIB] + objs + EXTRA_LIBS, check=True)
    return DIAG_LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))

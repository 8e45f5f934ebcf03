"""Builds libnerfsig.so (hand-written HIP for gfx950) in-tree with plain hipcc.

    python -m nerf_signature_amd.build [--force]

No torch.utils.cpp_extension, no hipify: the sources are HIP and the library has a C ABI
(include/nerfsig.h), so it is a single `hipcc -shared` of csrc/*.hip.  hipcc cross-compiles for
gfx950 without a GPU, so this also runs in the build container.
"""
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIB_DIR = os.path.join(PKG, "lib")
LIB = os.path.join(LIB_DIR, "libnerfsig.so")
ARCH = "gfx950"

# -ffp-contract=off: fused multiply-adds are written explicitly so integer results match the CPU oracle.
# -amdgpu-mfma-vgpr-form: MFMA results in ordinary VGPRs -- the MLP kernels post-process every accumulator element on the VALU, and
#  from the accumulator file each element costs an extra v_accvgpr_read (10 % of their instructions).
#  Except stage1_fused.hip: its kernel keeps 192 accumulator registers over its whole run and never post-processes them -- they belong in the AGPR half of
#  the unified register file (one wave per SIMD: 512 registers), which is where the compiler's default MFMA form puts them.
BASE_FLAGS = ["-O3", "-std=c++17", f"--offload-arch={ARCH}", "-ffp-contract=off", "-fPIC", "-fvisibility=hidden",
              "-fno-gpu-rdc", "-Wall", "-Wno-unused-function"]
FLAGS = BASE_FLAGS + ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]
FILE_FLAGS = {"stage1_fused.hip": BASE_FLAGS}


def flags_for(src):
    return FILE_FLAGS.get(os.path.basename(src), FLAGS)


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(PKG, "..", "include", "nerfsig.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(LIB_DIR, exist_ok=True)
    objs = []
    for src in sources():
        obj = os.path.join(LIB_DIR, os.path.basename(src)[:-4] + ".o")
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), *(os.path.getmtime(os.path.join(CSRC, h)) for h in os.listdir(CSRC) if h.endswith(".h"))):
            cmd = [hipcc, *flags_for(src), "-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
        objs.append(obj)
    cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB, *objs]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


def build_variant(name, extra_flags, out_root=None):
    """The same sources with extra compiler flags (defines of diagnostic builds) -> <out_root>/libnerfsig_<name>.so; select it with NERFSIG_LIB.  Per-file flags as build()."""
    out_root = out_root or os.path.join(PKG, "..", "tools", "_build")
    out = os.path.join(out_root, name)
    os.makedirs(out, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    procs, objs = [], []
    for src in sources():
        obj = os.path.join(out, os.path.basename(src)[:-4] + ".o")
        procs.append(subprocess.Popen([hipcc, *flags_for(src), *extra_flags, "-c", src, "-o", obj]))
        objs.append(obj)
    if any(p.wait() for p in procs):
        raise RuntimeError("variant build failed")
    lib = os.path.abspath(os.path.join(out_root, f"libnerfsig_{name}.so"))
    subprocess.check_call([hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", lib, *objs])
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))

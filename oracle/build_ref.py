"""oracle/build_ref.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Recipe that builds the REFERENCE's own native extension (raymarching/src/raymarching.cu + bindings.cpp, the pybind
module the reference calls `_raymarching`) for gfx950 into oracle/_ref/_raymarching_ref.so, from the sources where they
lie under /root/reference.  It is the strongest checker this repo has for SURVEY.md 8(a) rows R2, R3, R8, R10, R11:
the GPU parity tests (tests/test_gpu_ref_native.py) compare libnerfsig's rm_* entry points AND the C restatement
oracle/raymarch_ref.c with it.

What the recipe does (and does not):
  * the reference's own build path (raymarching/backend.py: torch.utils.cpp_extension.load with a hard-coded -std=c++14)
    fails against torch 2.10's headers; this recipe performs the same two steps that path performs -- torch's source
    translation (torch.utils.hipify, a tool of the image) and a compile against the installed torch headers -- by hand
    with -std=c++17.  No stand-in headers, no edits to the kernels.
  * the translation runs on a scratch copy in a temporary directory OUTSIDE the repository (hipify writes next to its
    input); only the shared object is kept.  No reference source, translated or not, is stored in the repo or travels
    to the GPU box; oracle/_ref/ is git-ignored.
  * nothing is written to /root/reference.

    python -m oracle.build_ref [--force]
"""
import os
import shutil
import subprocess
import sys
import sysconfig
import tempfile

_HERE = os.path.dirname(os.path.abspath(__file__))
REF_SRC = "/root/reference/raymarching/src"
OUT_DIR = os.path.join(_HERE, "_ref")
OUT = os.path.join(OUT_DIR, "_raymarching_ref.so")
MODULE = "_raymarching_ref"


def available():
    return os.path.exists(OUT)


def can_build():
    return all(os.path.exists(os.path.join(REF_SRC, f)) for f in ("raymarching.cu", "bindings.cpp", "raymarching.h"))


def build(force=False, verbose=False):
    """Returns the path of the built module, or None when /root/reference is absent (GPU box: uses the prebuilt file)."""
    if not can_build():
        return OUT if available() else None
    newest = max(os.path.getmtime(os.path.join(REF_SRC, f)) for f in os.listdir(REF_SRC))
    if not force and available() and os.path.getmtime(OUT) > max(newest, os.path.getmtime(os.path.abspath(__file__))):
        return OUT
    import torch
    from torch.utils.hipify import hipify_python
    tinc = os.path.join(os.path.dirname(torch.__file__), "include")
    tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    defs = ["-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1", "-DHIPBLAS_V2", f"-DTORCH_EXTENSION_NAME={MODULE}", "-DTORCH_API_INCLUDE_EXTENSION_H",
            f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}"]
    incs = [f"-I{REF_SRC}", f"-I{tinc}", f"-I{os.path.join(tinc, 'torch', 'csrc', 'api', 'include')}", f"-I{sysconfig.get_paths()['include']}",
            "-I/opt/rocm/include"]
    os.makedirs(OUT_DIR, exist_ok=True)
    tmp = tempfile.mkdtemp(prefix="nerfsig_refbuild_")
    try:
        shutil.copy(os.path.join(REF_SRC, "raymarching.cu"), tmp)
        quiet = open(os.devnull, "w")
        stdout = sys.stdout
        try:
            if not verbose:
                sys.stdout = quiet
            res = hipify_python.hipify(project_directory=tmp, output_directory=tmp, includes=[os.path.join(tmp, "*")],
                                       extra_files=[os.path.join(tmp, "raymarching.cu")], is_pytorch_extension=True, hipify_extra_files_only=True,
                                       clean_ctx=hipify_python.GeneratedFileCleaner(keep_intermediates=True))
        finally:
            sys.stdout = stdout
        hip_src = res[os.path.join(tmp, "raymarching.cu")].hipified_path
        run = lambda cmd: (print(" ".join(cmd)) if verbose else None, subprocess.check_call(cmd))
        # the reference's flags (backend.py:8-14) with the language level raised; hipcc's default fp contraction, like nvcc's, is on
        run([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", *defs, *incs, "-c", hip_src, "-o", os.path.join(tmp, "raymarching.o")])
        run(["g++", "-O3", "-std=c++17", "-fPIC", *defs, *incs, "-c", os.path.join(REF_SRC, "bindings.cpp"), "-o", os.path.join(tmp, "bindings.o")])
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT, os.path.join(tmp, "raymarching.o"), os.path.join(tmp, "bindings.o"),
             f"-L{tlib}", "-lc10", "-lc10_hip", "-ltorch_cpu", "-ltorch_hip", "-ltorch", "-ltorch_python", f"-Wl,-rpath,{tlib}"])
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))

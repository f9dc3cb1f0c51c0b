"""Builds libdetr_match.so (the HIP kernels + C ABI) in-tree with hipcc for gfx950.

    python build_ext.py [--force]

hipcc cross-compiles without a GPU, so this runs in the build container; the
resulting shared object travels with the tree to the GPU box.
"""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "detr_amd")
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libdetr_match.so")
SOURCES = ["cost_kernel.hip", "lap_kernel.hip", "pack_kernel.hip", "consumer_kernels.hip", "cls_loss_kernels.hip", "aux_kernels.hip", "abi.hip"]
HEADERS = [os.path.join(CSRC, "dm_common.h"), os.path.join(CSRC, "cost_math.h"), os.path.join(ROOT, "include", "detr_match.h")]
# -ffp-contract=off: every fp32 operation of the cost is rounded separately, as in the
# reference's elementwise torch ops (hipcc would otherwise fuse a*b+c into an fma).
FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17", "-Wno-pass-failed"]


def _stale():
    if not os.path.exists(LIB):
        return True
    built = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + HEADERS
    return any(os.path.getmtime(d) > built for d in deps)


HELPER_SRC = os.path.join(CSRC, "index_views.cpp")
HELPER = os.path.join(PKG, "_index_views.so")


def build_helper(force=False, verbose=False):
    """The host-side helper module detr_amd._index_views (bulk creation of the per-image index views;
    plain C++ against the torch headers, no GPU code)."""
    import sysconfig

    import torch
    from torch.utils import cpp_extension

    # the helper reaches into tensor internals: it is built for exactly one torch version and says which
    stamp = ("detr_amd_index_views built for torch %s;" % torch.__version__).encode()
    if not force and os.path.exists(HELPER) and os.path.getmtime(HELPER) >= os.path.getmtime(HELPER_SRC):
        with open(HELPER, "rb") as f:
            if stamp in f.read():
                return HELPER

    libdir = cpp_extension.library_paths()[0]
    cmd = ["g++", "-O2", "-shared", "-fPIC", "-std=c++17",
           "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI),
           '-DDM_BUILT_FOR_TORCH="%s"' % torch.__version__]
    for inc in cpp_extension.include_paths() + [sysconfig.get_paths()["include"]]:
        cmd += ["-I", inc]
    cmd += [HELPER_SRC, "-o", HELPER, "-L", libdir, "-ltorch_python", "-ltorch", "-ltorch_cpu", "-lc10",
            "-Wl,-rpath," + libdir]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return HELPER


DEMO_SRC = os.path.join(ROOT, "tests", "c_abi", "abi_demo.c")
DEMO = os.path.join(ROOT, "tests", "c_abi", "abi_demo")


def build_demo(force=False, verbose=False):
    """tests/c_abi/abi_demo: the matcher step from plain C (gcc; the HIP runtime's C API and libdetr_match.so, no torch),
    run by tests/test_gpu_c_abi.py."""
    newest = max(os.path.getmtime(DEMO_SRC), os.path.getmtime(LIB), os.path.getmtime(HEADERS[-1]))
    if not force and os.path.exists(DEMO) and os.path.getmtime(DEMO) >= newest:
        return DEMO
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    cmd = ["gcc", "-O1", "-Wall", "-std=c11", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(rocm, "include"),
           DEMO_SRC, "-o", DEMO, "-L", PKG, "-ldetr_match", "-L", os.path.join(rocm, "lib"), "-lamdhip64",
           "-Wl,-rpath,$ORIGIN/../../detr_amd", "-Wl,-rpath," + os.path.join(rocm, "lib")]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return DEMO


def build_library(force=False, verbose=False):
    build_helper(force=force, verbose=verbose)
    _build_kernels(force=force, verbose=verbose)
    build_demo(force=force, verbose=verbose)
    return LIB


def _build_kernels(force=False, verbose=False):
    if not force and not _stale():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    # DETR_AMD_EXTRA_FLAGS: extra hipcc flags for instrumented one-off builds (e.g. -DDM_LAP_STAMPS)
    extra = os.environ.get("DETR_AMD_EXTRA_FLAGS", "").split()
    cmd = [hipcc] + FLAGS + extra + ["-I", os.path.join(ROOT, "include"), "-I", CSRC, "-o", LIB] + \
          [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))

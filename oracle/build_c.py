"""Build the oracle's native pieces (TEST INFRASTRUCTURE):
  oracle/liboracle_c.so           <- oracle/points_in_boxes.c   (gcc, always)
  oracle/_ref/roiaware_pool3d_ref.so  <- the REFERENCE's own roiaware_pool3d.cpp compiled where it
        lies under /root/reference (build container only; g++ on that one file against the
        installed torch headers -- the reference's setup.py / CUDA build is not run).  The CUDA
        launchers it declares stay undefined symbols; only points_in_boxes_cpu is ever called,
        and the module is loaded with lazy binding.  Used to validate the C restatement.
Nothing here is imported by the product packages."""
from __future__ import annotations

import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_C = os.path.join(HERE, "liboracle_c.so")
REF_SRC = "/root/reference/lidargen/ops/roiaware_pool3d/src/roiaware_pool3d.cpp"
REF_DIR = os.path.join(HERE, "_ref")
REF_SO = os.path.join(REF_DIR, "roiaware_pool3d_ref.so")


def _stale(out, srcs):
    return not os.path.exists(out) or any(os.path.getmtime(s) > os.path.getmtime(out) for s in srcs)


def build_c():
    src = os.path.join(HERE, "points_in_boxes.c")
    if _stale(LIB_C, [src]):
        subprocess.run(["gcc", "-O2", "-shared", "-fPIC", "-ffp-contract=off", src, "-o", LIB_C,
                        "-lm"], check=True)
    return LIB_C


def build_ref():
    """Only where /root/reference exists (the build container)."""
    if not os.path.exists(REF_SRC):
        return None
    if _stale(REF_SO, [REF_SRC]):
        import torch
        from torch.utils.cpp_extension import include_paths

        os.makedirs(REF_DIR, exist_ok=True)
        tl = os.path.join(os.path.dirname(torch.__file__), "lib")
        cmd = ["g++", "-O2", "-shared", "-fPIC", "-std=c++17", "-w",
               "-DTORCH_EXTENSION_NAME=roiaware_pool3d_ref", "-DTORCH_API_INCLUDE_EXTENSION_H",
               f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}"]
        cmd += [f"-I{p}" for p in include_paths()] + [f"-I{sysconfig.get_paths()['include']}"]
        cmd += [REF_SRC, "-o", REF_SO, f"-L{tl}", "-ltorch", "-ltorch_cpu", "-lc10",
                "-ltorch_python", f"-Wl,-rpath,{tl}"]
        subprocess.run(cmd, check=True)
    return REF_SO


def load_ref():
    """Import oracle/_ref/roiaware_pool3d_ref.so (lazy binding: CUDA launchers stay unresolved)."""
    if not os.path.exists(REF_SO):
        return None
    import importlib.util

    import torch  # noqa: F401  (libtorch must be loaded first)

    old = sys.getdlopenflags()
    sys.setdlopenflags(os.RTLD_LAZY | os.RTLD_GLOBAL)
    try:
        spec = importlib.util.spec_from_file_location("roiaware_pool3d_ref", REF_SO)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        sys.setdlopenflags(old)
    return mod


def build():
    build_c()
    try:
        build_ref()
    except Exception as e:  # the checker's optional half; report, do not hide
        print("oracle/_ref build failed:", e)


if __name__ == "__main__":
    build()
    print("liboracle_c:", os.path.exists(LIB_C), " _ref:", os.path.exists(REF_SO))

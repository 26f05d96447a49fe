"""Build liblidarcrafter_hip.so for gfx950 with hipcc (in-tree, no torch headers involved)."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "liblidarcrafter_hip.so")
# the convolution kernels alone with ONE fp16 product per multiply (-DLC_F16X2_TERMS=1): what a caller under
# torch.autocast(float16) asked for (the reference's bulk harness); never used otherwise (ops.conv_products)
LIB_P1 = os.path.join(HERE, "liblidarcrafter_hip_p1.so")
SOURCES = ["conv.hip", "conv_f16x2.hip", "conv_f16x2_tall.hip", "conv_f16x2_s2.hip", "norm.hip", "resample.hip", "upfold.hip", "misc.hip", "attention.hip", "attention_units.hip", "geometry.hip", "roipool.hip", "lidar.hip", "layout.hip", "temporal.hip", "metrics.hip", "voxel.hip", "conv_bwd.hip", "attention_bwd.hip", "attention_bwd_h.hip"]


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm); cannot build the HIP hot path")


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    if not os.path.exists(LIB_P1):
        return True
    t = min(os.path.getmtime(LIB), os.path.getmtime(LIB_P1))
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    deps.append(os.path.join(os.path.dirname(HERE), "include", "lidarcrafter_hip.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not needs_build():
        return LIB
    objs = []
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    procs = []
    for src in SOURCES:
        obj = os.path.join(HERE, "build", src.replace(".hip", ".o"))
        # (-Wno-inline-asm: conv_f16x2_tall.hip names m0 in an asm clobber list on purpose -- its LDS-DMA instruction)
        cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-inline-asm", "-c",
               os.path.join(CSRC, src), "-o", obj] + os.environ.get("LC_EXTRA_HIPCC_FLAGS", "").split()
        procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(obj)
    p1_objs = []
    for src in ("conv_f16x2.hip", "conv_f16x2_tall.hip"):
        p1_obj = os.path.join(HERE, "build", src.replace(".hip", "_p1.o"))
        cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-inline-asm", "-DLC_F16X2_TERMS=1", "-c",
               os.path.join(CSRC, src), "-o", p1_obj] + os.environ.get("LC_EXTRA_HIPCC_FLAGS", "").split()
        procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        p1_objs.append(p1_obj)
    for cmd, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed: %s\n%s" % (" ".join(cmd), out.decode()))
        if verbose and out.strip():
            print(out.decode())
    cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    subprocess.run(cmd, check=True)
    subprocess.run([hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_P1] + p1_objs, check=True)
    if verbose:
        print("built", LIB, "and", os.path.basename(LIB_P1))
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)

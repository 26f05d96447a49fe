"""Per-kernel instruction census of the built library (gfx950 code objects): the patterns that cost round 5's epilogues --
exec-mask branches, full `s_waitcnt vmcnt(0)` waits, 64-bit address arithmetic, IEEE divisions, system-scope stores, scratch.
    python devtools/isa_scan.py [lib] [--match conv_f16x2,gn_,attn] [--top 30]"""
import argparse
import glob
import os
import re
import shutil
import subprocess
import tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernels(lib):
    tmp = tempfile.mkdtemp()
    try:
        local = os.path.join(tmp, os.path.basename(lib))
        shutil.copy(lib, local)
        subprocess.run([OBJDUMP, "--offloading", local], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        out = {}
        for co in sorted(glob.glob(local + ".*gfx950")):
            dis = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", co], check=True, capture_output=True, text=True).stdout
            cur = None
            for ln in dis.splitlines():
                m = re.match(r"^[0-9a-f]+ <(.+)>:$", ln)
                if m:
                    cur = m.group(1)
                    out[cur] = []
                elif cur is not None:
                    t = ln.strip().split("//")[0].strip()
                    if t and not t.startswith(("/", ".")):
                        out[cur].append(t)
        return out
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def census(ins):
    c = dict(n=len(ins), mfma=0, exec_br=0, vm0=0, addr64=0, div=0, sysst=0, scratch=0, vmem_st=0, vmem_ld=0)
    for i in ins:
        mn = i.split()[0]
        if mn.startswith("v_mfma"):
            c["mfma"] += 1
        elif mn in ("s_cbranch_execz", "s_cbranch_execnz"):
            c["exec_br"] += 1
        elif mn == "s_waitcnt" and re.search(r"vmcnt\(0\)", i):
            c["vm0"] += 1
        elif mn in ("v_lshl_add_u64", "v_mad_u64_u32", "v_mul_hi_u32"):
            c["addr64"] += 1
        elif mn.startswith(("v_div_scale", "v_div_fixup")):
            c["div"] += 1
        elif mn.startswith("scratch_"):
            c["scratch"] += 1
        if "store" in mn and re.search(r"\bsc0 sc1\b", i):
            c["sysst"] += 1
        if re.match(r"(global|buffer|flat)_store", mn):
            c["vmem_st"] += 1
        if re.match(r"(global|buffer|flat)_load", mn):
            c["vmem_ld"] += 1
    return c


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("lib", nargs="?", default=os.path.join(ROOT, "lidarcrafter_amd", "liblidarcrafter_hip.so"))
    ap.add_argument("--match", default="")
    ap.add_argument("--top", type=int, default=40)
    ap.add_argument("--sort", default="exec_br")
    a = ap.parse_args()
    pats = [p for p in a.match.split(",") if p]
    rows = []
    for k, ins in kernels(a.lib).items():
        if pats and not any(p in k for p in pats):
            continue
        rows.append((re.sub(r"^_ZN\d+_GLOBAL__N_1\d+|lcconv\d*", "", k)[:84], census(ins)))
    rows.sort(key=lambda r: -r[1][a.sort])
    print("%-86s %6s %5s %7s %4s %6s %4s %5s %7s %5s %5s" % ("kernel", "instr", "mfma", "exec_br", "vm0", "addr64", "div", "sysst", "scratch", "st", "ld"))
    for k, c in rows[:a.top]:
        print("%-86s %6d %5d %7d %4d %6d %4d %5d %7d %5d %5d" % (k, c["n"], c["mfma"], c["exec_br"], c["vm0"], c["addr64"], c["div"],
                                                             c["sysst"], c["scratch"], c["vmem_st"], c["vmem_ld"]))

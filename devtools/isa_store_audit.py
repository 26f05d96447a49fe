"""Static audit of a built library for the ">64-bit store data" pattern (round 4, profiles/r04_entry_store.txt).

For every kernel of every gfx950 code object in the library: each store wider than 64 bits (buffer / global / flat /
scratch _dwordx3/_dwordx4, ds_write_b96/b128 excluded: LDS reads its data at issue) is listed when one of the next
`--window` instructions WRITES one of its data registers -- by a VALU instruction, an LDS read or a memory load.  LLVM
inserts the ISA's wait state only for a VALU writer directly behind such a store, and exempts MUBUF stores whose soffset
is an SGPR; on gfx950 that exemption produced ~3 corrupted statistics entries per 10^7 in the deferred epilogue
(devtools/entry_stress.py), so the audit ignores it and reports every writer inside the window.

    python devtools/isa_store_audit.py lidarcrafter_amd/liblidarcrafter_hip.so [--window 4]
"""
import argparse
import glob
import os
import re
import shutil
import subprocess
import sys
import tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
WIDE = re.compile(r"^(buffer_store_dwordx[34]|buffer_store_format_xyzw?|global_store_dwordx[34]|flat_store_dwordx[34]|scratch_store_dwordx[34])\b")
REG = re.compile(r"v\[(\d+):(\d+)\]|\bv(\d+)\b")


def regs(tok):
    m = REG.search(tok)
    if not m:
        return set()
    if m.group(1) is not None:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return {int(m.group(3))}


def dst_regs(mn, ops):
    """VGPRs an instruction writes (first operand for VALU / loads / ds_read; none for stores, branches, s_*)."""
    if mn.startswith(("s_", "buffer_store", "global_store", "flat_store", "scratch_store", "ds_write", "ds_store",
                      "v_cmp", "v_nop", "buffer_wbl2", "buffer_inv", "global_atomic", "buffer_atomic")) or not ops:
        if mn.startswith(("global_atomic", "buffer_atomic")) and "sc0" in " ".join(ops):
            return regs(ops[0])
        return set()
    if mn.startswith("v_mfma") or mn.startswith("v_smfmac"):
        return regs(ops[0])   # (a[...] destinations do not match REG)
    return regs(ops[0])


def data_regs(mn, ops):
    if mn.startswith(("global_store", "flat_store", "scratch_store")):
        return regs(ops[1])
    return regs(ops[0])      # MUBUF: vdata first


def audit(lib, window):
    tmp = tempfile.mkdtemp()
    try:
        local = os.path.join(tmp, os.path.basename(lib))
        shutil.copy(lib, local)
        subprocess.run([OBJDUMP, "--offloading", local], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        found, nwide, nkern = [], 0, 0
        for co in sorted(glob.glob(local + ".*gfx950")):
            dis = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", co], check=True, capture_output=True, text=True).stdout
            kern, ins = None, []
            def flush():
                nonlocal nwide
                for i, (mn, ops, line) in enumerate(ins):
                    if not WIDE.match(mn):
                        continue
                    nwide += 1
                    d = data_regs(mn, ops)
                    for k in range(1, window + 1):
                        if i + k >= len(ins):
                            break
                        mn2, ops2, line2 = ins[i + k]
                        hit = dst_regs(mn2, ops2) & d
                        if hit:
                            found.append((kern, line, k, line2))
                            break
            for ln in dis.splitlines():
                m = re.match(r"^[0-9a-f]+ <(.+)>:$", ln)
                if m:
                    flush(); kern, ins = m.group(1), []; nkern += 1
                    continue
                t = ln.strip().split("//")[0].strip()
                if not t or t.startswith(("/", ".")) or ":" in t.split()[0]:
                    continue
                parts = t.split(None, 1)
                mn = parts[0]
                ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
                ins.append((mn, ops, t))
            flush()
        return nkern, nwide, found
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("lib")
    ap.add_argument("--window", type=int, default=4)
    a = ap.parse_args()
    nkern, nwide, found = audit(a.lib, a.window)
    print("%s: %d kernels, %d stores wider than 64 bits, %d with a writer of their data registers within %d instructions"
          % (a.lib, nkern, nwide, len(found), a.window))
    for kern, st, k, wr in found:
        print("  %s\n      %s\n      +%d: %s" % (re.sub(r"^_ZN\d+_GLOBAL__N_1\d+", "", kern)[:110], st, k, wr))

"""Static checks of the built library's gfx950 code (CPU suite; llvm-objdump from the ROCm image).  They pin what round 5
found in epilogues the hard way (profiles/r05_level0.txt sections 7-8):
  * a store wider than 64 bits whose data registers are written by the very next instruction (the ISA wants wait states in
    between; hipcc inserts them for stores it knows, NOT behind inline assembly): corrupted planes / statistics entries;
  * system-scope (`sc0 sc1`) stores -- what `volatile` stores compile to, each followed by `s_waitcnt vmcnt(0)`: +6 ... +11 us
    per launch of the pre-split convolution;
  * an IEEE division (`v_div_scale_f32`) in the GroupNorm apply passes' SiLU."""
import glob
import os
import re
import shutil
import subprocess
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "devtools"))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
LIB = os.path.join(ROOT, "lidarcrafter_amd", "liblidarcrafter_hip.so")

pytestmark = pytest.mark.skipif(not (os.path.exists(OBJDUMP) and os.path.exists(LIB)),
                                reason="needs the built library and llvm-objdump")


@pytest.fixture(scope="module")
def kernels():
    """{kernel name: [instruction lines]} of every gfx950 code object in the library."""
    tmp = tempfile.mkdtemp()
    try:
        local = os.path.join(tmp, os.path.basename(LIB))
        shutil.copy(LIB, local)
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


def test_no_writer_directly_behind_a_wide_store():
    import isa_store_audit as A

    nkern, nwide, found = A.audit(LIB, 1)
    assert nkern > 400 and nwide > 100          # (the audit saw the library)
    assert not found, [(k[:60], st, wr) for k, st, _, wr in found[:5]]


def test_no_system_scope_stores_on_the_hot_path(kernels):
    hot = ("conv_f16x2", "conv1x1_ps", "gn_apply", "gn_stats", "attn_h_kernel", "up2_kernel", "down2", "splitk_reduce", "pstep")
    bad = [(k[:70], i) for k, ins in kernels.items() if any(h in k for h in hot)
           for i in ins if "store" in i.split()[0] and re.search(r"\bsc0 sc1\b", i)]
    assert not bad, bad[:5]


def test_no_ieee_division_in_the_groupnorm_apply_passes(kernels):
    ks = [k for k in kernels if "gn_apply_split_kernel" in k or "gn_apply_os_kernel" in k]
    assert len(ks) >= 4
    for k in ks:
        n = sum(1 for i in kernels[k] if i.startswith("v_div_scale_f32"))
        assert n == 0, (k[:70], n)

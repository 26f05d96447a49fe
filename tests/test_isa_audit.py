"""Static checks of the built library's gfx950 code (CPU suite; llvm-objdump from the ROCm image).  They pin what round 5
found in epilogues the hard way (profiles/r05_level0.txt sections 7-8):
  * a store wider than 64 bits whose data registers are written by the very next instruction (the ISA wants wait states in
    between; hipcc inserts them for stores it knows, NOT behind inline assembly): corrupted planes / statistics entries;
  * system-scope (`sc0 sc1`) stores -- what `volatile` stores compile to, each followed by `s_waitcnt vmcnt(0)`: +6 ... +11 us
    per launch of the pre-split convolution;
  * an IEEE division (`v_div_scale_f32`) in the GroupNorm apply passes' SiLU;
and the hand-counted `s_waitcnt vmcnt(N)` in front of the tall convolution kernel's chunk barriers: the weight
LDS-DMA is inline assembly that hipcc cannot count, so the ISA itself must show at least N VMEM operations between a chunk's
last DMA piece and the wait."""
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


_VMEM = re.compile(r"^(buffer|global|flat|scratch)_(load|store|atomic)")


def test_tall_kernel_chunk_barrier_waits_cover_the_weight_dma(kernels):
    """conv_f16x2_tall.hip k_iter: `wait_vmcnt(later)` + bare `s_barrier` per chunk, `later` = the deferred-epilogue
    operations (+ x loads) issued BEHIND the chunk's last weight-DMA piece.  VMEM operations return in order, so the
    pieces have landed at the barrier iff at least `later` VMEM instructions really follow the last
    `buffer_load_dwordx4 ... lds` -- if hipcc hoists, merges or drops one of the counted operations the wait lets a piece
    stay in flight across the barrier (a silent LDS race).  Checked on the disassembly of every instantiation: for each
    barrier with a vmcnt(N > 0) wait in front of it, the VMEM instructions between the last LDS-DMA and the wait number
    >= N (the strictest wait in front of the barrier counts: hipcc may add its own)."""
    ks = {k: v for k, v in kernels.items() if "conv_f16x2_tall_kernel" in k and not k.startswith("__")}
    assert len(ks) >= 12, sorted(ks)[:3]
    for name, ins in ks.items():
        nch = int(re.search(r"tall_kernelILi(\d+)E", name).group(1))
        steady = 0
        for b, t in enumerate(ins):
            if not t.startswith("s_barrier"):
                continue
            waits, j = [], b - 1
            while j > 0 and b - j <= 12:                 # the waits directly in front of the barrier
                m = re.match(r"s_waitcnt.*vmcnt\((\d+)\)", ins[j])
                if m:
                    waits.append((int(m.group(1)), j))
                elif _VMEM.match(ins[j]) or ins[j].startswith("v_mfma"):
                    break
                j -= 1
            if not waits:
                continue                                  # a prologue __syncthreads (GroupNorm fold): no DMA accounting
            n, at = min(waits)
            if n == 0:
                continue                                  # full drain: nothing to count
            count, i = 0, at - 1
            while i > 0 and not (ins[i].startswith("buffer_load") and ins[i].rstrip().endswith(" lds")):
                count += 1 if _VMEM.match(ins[i]) else 0
                i -= 1
            assert i > 0, (name[:70], b, "no LDS-DMA in front of a counted wait")
            assert count >= n, (name[:70], f"barrier at {b}: vmcnt({n}) but only {count} VMEM operations follow the "
                                           f"last weight-DMA piece (instruction {i}): the DMA may be in flight at the barrier")
            if not any(x.startswith(("s_cbranch", "s_branch")) for x in ins[i:b]):
                steady += 1
        assert steady >= nch, (name[:70], steady)        # every chunk of the K loop was seen


def test_no_waterfall_loops_in_the_streaming_and_projection_kernels(kernels):
    """A buffer descriptor (or an LDS-DMA destination) that hipcc cannot prove wave-uniform is put through a waterfall loop:
    v_readfirstlane x 4, v_cmp_eq_u64 x 2, s_and_saveexec, the memory operation, s_xor exec, s_cbranch_execnz -- per operation.
    Round 6 found them around every x DMA of the pre-split 1x1 kernel's K loop (the sample's base address was a 64-bit product
    on the vector unit: projections 3-7 % slower) and around the 16-byte stores of the combine pass (profiles/r06_fold_up.txt
    section 9).  The kernels below must have none."""
    hot = ("conv1x1_ps_kernel", "up2_combine9_kernel", "conv_f16x2_ps_kernel", "conv_f16x2_tall_kernel", "conv_s2_shared_w_kernel",
           "gn_apply_split_kernel", "split_plain_kernel", "attn_u_kernel", "fir_down2_prefilter_split_kernel")
    seen, bad = 0, []
    for name, ins in kernels.items():
        if not any(h in name for h in hot):
            continue
        seen += 1
        n = sum(1 for i, t in enumerate(ins) if t.startswith("v_cmp_eq_u64") and any(u.startswith("s_and_saveexec") for u in ins[i:i + 5]))
        if n:
            bad.append((name[:80], n))
    assert seen >= 20, seen
    assert not bad, bad[:6]

"""A hazard no GPU test can see reliably: `to::attn_bwd_to_kernel` issues the k / v rows of a tile's
targets as inline-asm register loads and waits for them by COUNT a few hundred instructions later
(csrc/edge_attn_to.hip, SPT_TO_NODE_WAIT_SEL).  The compiler does not know that those registers have
loads pending: if register allocation makes it copy or reuse them between the loads and the wait
(round 6: it did, as soon as the two roles issued the loads from two program points - the copy read
stale values on one role, and only graphs with many targets per tile showed it), results are wrong
without any fault.  This test compiles the file to gfx950 assembly and checks, for both precision
instances, that both roles load the rows into the SAME registers and that no instruction on a role's
path between its last row load and the counted wait reads or writes them."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _regs(code):
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", code):
        if m.group(1):
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


@pytest.mark.skipif(not (os.path.exists(HIPCC) or shutil.which("hipcc")), reason="no hipcc")
def test_target_row_registers_are_untouched_between_their_loads_and_the_counted_wait(tmp_path):
    from superpoint_transformer_amd import build
    src = os.path.join(build.CSRC, "edge_attn_to.hip")
    asm = str(tmp_path / "edge_attn_to.s")
    flags = build.FLAGS + build.PER_FILE_FLAGS.get("edge_attn_to.hip", [])
    r = subprocess.run([HIPCC] + flags + ["-S", "--cuda-device-only", "-o", asm, src],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    text = open(asm).read()
    checked = 0
    for prec in (3, 1):
        m = re.search(rf"^_ZN3spt2to18attn_bwd_to_kernelILi{prec}E.*?s_endpgm", text, re.S | re.M)
        assert m, f"kernel instance PREC = {prec} not found"
        lines = m.group(0).split("\n")
        loads = [i for i, l in enumerate(lines)
                 if re.match(r"\s*global_load_dwordx4 v\[\d+:\d+\], v\[\d+:\d+\], off", l)]
        # the two roles of a wave pair issue the rows from their own branch: two groups of four
        assert len(loads) in (4, 8), f"PREC {prec}: expected the asm row loads in groups of four, found {len(loads)}"
        groups = [loads[i:i + 4] for i in range(0, len(loads), 4)]
        dests = []
        for grp in groups:
            dest = set()
            for i in grp:
                d = re.match(r"\s*global_load_dwordx4 v\[(\d+):(\d+)\]", lines[i])
                dest.update(range(int(d.group(1)), int(d.group(2)) + 1))
            assert len(dest) == 16
            dests.append(dest)
        # ... into the SAME registers (else the merge of the branches needs copies of registers
        # with loads pending)
        assert all(d == dests[0] for d in dests), f"PREC {prec}: the roles load the rows into different registers"
        dest = dests[0]
        wait = next(i for i in range(loads[-1], len(lines)) if "v_readfirstlane_b32 vcc_lo" in lines[i])
        # every group's own path: from its last load to the counted wait, or - the role whose block
        # the compiler laid out first - to the unconditional branch that leaves that block (what
        # follows in the FILE up to the merge is the other role's path)
        for grp in groups:
            for i in range(grp[-1] + 1, wait):
                code = lines[i].split(";")[0]
                if not code.strip() or code.strip().startswith("."):
                    continue
                if re.match(r"\s*s_branch\b", code) and grp is not groups[-1]:
                    break
                hit = _regs(code) & dest
                assert not hit, (f"PREC {prec}: `{lines[i].strip()}` touches v{sorted(hit)} while the target "
                                 f"rows are still in flight (asm lines {grp[-1] + 1}..{wait})")
        checked += 1
    assert checked == 2

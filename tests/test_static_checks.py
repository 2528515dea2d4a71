"""No-GPU checks of the hand-scheduled kernels: the LDS ring protocols of the two opt-in kernels (scripts/checks: every
fragment read against the counted waits and the barriers), the index-level equivalence of the pipelined ffn_fwd's sliced E1
stage with the original, and - from the gfx950 assembly hipcc cross-compiles here - that no kernel of the token-stationary
files keeps a register spill inside a loop (a scratch reload there is followed by `s_waitcnt vmcnt(0)`, i.e. it drains the
weight stream's DMA every iteration)."""
import importlib.util
import os
import re
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(path):
    spec = importlib.util.spec_from_file_location(os.path.basename(path)[:-3], path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_ffn_pipe_ring_protocol_has_no_hazard():
    m = _load(os.path.join(ROOT, "scripts", "checks", "ffn_pipe_protocol.py"))
    for stagger in (True, False):
        for n in range(2, 33):
            assert m.check(n, stagger) == [], (n, stagger)
    # the check can fail: a W2 stream with the W1 stream's lead overwrites W2(k - 1) while X(k) still reads it
    m.issued_at = lambda half, c: (-2 if c < 3 else c - 3)
    assert m.check(16, True)


def test_attn_four_slot_counted_waits_cover_the_next_chunk():
    m = _load(os.path.join(ROOT, "scripts", "checks", "attn_ring_protocol.py"))
    assert m.check() == 0
    m.wait_value = lambda k: 5
    assert m.check() > 0


def test_ffn_bwd_one_counted_waits_cover_their_loads():
    m = _load(os.path.join(ROOT, "scripts", "checks", "ffn_bwd_one_protocol.py"))
    for slots in (4, 3):
        m.NBUF = slots
        assert m.check() == 0, slots
    m.gate_wait = lambda c: 12
    assert m.check() > 0


def test_ffn_pipe_sliced_e1_equals_the_original():
    m = _load(os.path.join(ROOT, "scripts", "checks", "ffn_pipe_e1_equiv.py"))
    with pytest.raises(SystemExit) as e:
        m.main()
    assert e.value.code == 0


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="no hipcc")
@pytest.mark.parametrize("src", ["ffn_fused", "attn_fused"])
def test_no_spill_inside_a_loop(tmp_path, src):
    out = tmp_path / f"{src}.s"
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wno-unused-result",
                    "-Wno-unused-value", "-S", "--cuda-device-only", os.path.join(ROOT, "deepsvg_amd", "csrc", f"{src}.hip"),
                    "-o", str(out)], check=True, capture_output=True)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "isa_loop_mix.py"), "--spills", str(out)],
                       check=True, capture_output=True, text=True)
    rows = [l for l in r.stdout.splitlines() if "scratch instructions" in l]
    inside = [l for l in rows if int(re.search(r"inside loops\s+(\d+)", l).group(1)) > 0]
    assert not inside, "\n".join(inside)
    assert any("mfma" in l for l in r.stdout.splitlines())          # the scan found the kernels' loops

"""No-GPU check of the hand-scheduled kernels: from the gfx950 assembly hipcc cross-compiles here, no kernel of the token-
stationary files keeps a register spill inside a loop (a scratch reload there is followed by `s_waitcnt vmcnt(0)`, i.e. it
drains the weight stream's DMA every iteration)."""
import os
import re
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


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


def test_committed_ffn_traffic_was_measured_on_this_kernel_source():
    """profiles/ffn_traffic.json (bench.py's roofline.traffic, a committed rocprofv3 --pmc measurement) names the code of
    csrc/ffn_fused.hip it was collected on: an edited kernel makes the number stale - re-run scripts/gpu_ffn_traffic.sh"""
    import json
    sys.path.insert(0, ROOT)
    import bench
    t = json.load(open(os.path.join(ROOT, "profiles", "ffn_traffic.json")))
    assert t["ffn_fused_hip_code_sha256_16"] == bench.kernel_source_hash(os.path.join(ROOT, "deepsvg_amd", "csrc", "ffn_fused.hip"))
    assert t.get("commit")

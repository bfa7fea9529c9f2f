#!/usr/bin/env python3
"""Register / scratch / occupancy table of the recurrence kernels (hipcc -Rpass-analysis).

    python3 tools/kernel_resources.py [filter]

Used by tests/test_kernel_resources.py as a regression guard: an innocent-looking edit that
pushes a hot kernel past a VGPR step (128, 168) or makes it spill costs 10-30 %.
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def collect():
    src = os.path.join(ROOT, "bodge_amd", "csrc", "bodge_hip.hip")
    with tempfile.TemporaryDirectory() as tmp:
        cmd = ["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared",
               f"-I{ROOT}/include", f"-I{ROOT}/bodge_amd/csrc", "-Rpass-analysis=kernel-resource-usage",
               "-o", os.path.join(tmp, "t.so"), src, "-ldl"]
        text = subprocess.run(cmd, capture_output=True, text=True, check=True).stderr
    table, name = {}, None
    for line in text.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip() or m.group(1)
            table[name] = {}
            continue
        for key, label in (("vgpr", "VGPRs"), ("scratch", r"ScratchSize \[bytes/lane\]"), ("occupancy", r"Occupancy \[waves/SIMD\]")):
            m = re.search(rf"\b{label}: (\d+)", line)
            if m and name:
                table[name][key] = int(m.group(1))
    return table


if __name__ == "__main__":
    pattern = sys.argv[1] if len(sys.argv) > 1 else "cheb_step"
    for name, row in sorted(collect().items()):
        if pattern in name:
            print(f"{row.get('vgpr', '?'):>4} VGPR  {row.get('scratch', '?'):>3} B scratch  occ {row.get('occupancy', '?')}  {name}")

"""Per-kernel register / LDS / occupancy report (hipcc -Rpass-analysis=kernel-resource-usage).

    python -m scgaussian_amd.resource_usage
"""
import re
import subprocess
import sys

from .build import COMMON, CSRC, SOURCES, _hipcc
import os


def report():
    rows = []
    for src, extra in SOURCES.items():
        if src == "api.hip":
            continue
        cmd = [_hipcc()] + COMMON + extra + ["-c", os.path.join(CSRC, src), "-o", "/tmp/scg_ru.o",
                                             "-Rpass-analysis=kernel-resource-usage"]
        err = subprocess.run(cmd, capture_output=True, text=True).stderr
        cur = None
        for line in err.splitlines():
            m = re.search(r":\d+:\d+: remark: +(.*?) \[-Rpass", line)
            if not m:
                continue
            t = m.group(1)
            if t.startswith("Function Name:"):
                mangled = t.split(":", 1)[1].strip()
                name = subprocess.run(["c++filt", mangled], capture_output=True, text=True).stdout.split("(")[0].strip()
                cur = {"name": name}
                rows.append(cur)
            elif cur is not None and ":" in t:
                k, v = t.split(":", 1)
                cur[k.strip()] = v.strip()
    return rows


def main():
    for r in report():
        print("%-34s VGPR %4s AGPR %3s SGPR %4s spill %s/%s scratch %5s LDS %6s occ %s" % (
            r["name"].replace("scg::", ""), r.get("VGPRs", "?"), r.get("AGPRs", "?"), r.get("SGPRs", "?"),
            r.get("VGPRs Spill", "?"), r.get("SGPRs Spill", "?"), r.get("ScratchSize [bytes/lane]", "?"),
            r.get("LDS Size [bytes/block]", "?"), r.get("Occupancy [waves/SIMD]", "?")))


if __name__ == "__main__":
    sys.exit(main())

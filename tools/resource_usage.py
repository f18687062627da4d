"""Per-kernel register / spill / LDS figures of the hot code object (hipcc -Rpass-analysis=kernel-resource-usage), one line
per instantiation.  `python tools/resource_usage.py [extra hipcc flags]`; profiles/*/resource_usage.txt are its output."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ("-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math "
         "-fhip-fp32-correctly-rounded-divide-sqrt -Rpass-analysis=kernel-resource-usage").split()


def main():
    rows = {}
    for src in ("spx_hot.hip", "spx_walk.hip"):
        hot = ["-mllvm", "-amdgpu-sched-strategy=max-memory-clause"] if src == "spx_hot.hip" else []   # (the Makefile's SPX_HOT_SCHED)
        r = subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + hot + sys.argv[1:] + ["-c", src, "-o", "/dev/null"],
                           cwd=os.path.join(ROOT, "speedy_amd", "csrc"), capture_output=True, text=True)
        cur = None
        for line in r.stderr.splitlines():
            m = re.search(r"Function Name: (\S+)", line)
            if m:
                cur = m.group(1)
                rows[cur] = {}
            for key in ("VGPRs:", "SGPRs:", "ScratchSize", "Occupancy", "VGPRs Spill", "SGPRs Spill", "LDS Size"):
                if key in line and cur:
                    rows[cur][key.strip(":")] = re.sub(r"\[-Rpass.*$", "", line.split(":")[-1]).strip()
    names = subprocess.run(["c++filt"], input="\n".join(rows), capture_output=True, text=True).stdout.splitlines()
    for name, key in sorted(zip(names, rows)):
        v = rows[key]
        name = re.sub(r"^void |\(.*$", "", name)
        print("%-52s VGPR %3s  SGPR %3s  spills v %2s s %3s  scratch %4s B  occupancy %s  LDS %s" % (
            name[:52], v.get("VGPRs"), v.get("SGPRs"), v.get("VGPRs Spill"), v.get("SGPRs Spill"),
            re.sub(r"\D", "", v.get("ScratchSize", "0")) or "0", v.get("Occupancy", "?").split()[0], v.get("LDS Size", "?")))


if __name__ == "__main__":
    main()

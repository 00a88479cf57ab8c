"""Per-kernel register / spill / scratch / occupancy table of one HIP source (hipcc -Rpass-analysis=kernel-resource-usage,
gfx950; no GPU needed).  python scripts/kernel_resources.py janusx_amd/csrc/k_scan_fast.hip [extra hipcc flags]"""
import os
import re
import subprocess
import sys

src = sys.argv[1]
extra = sys.argv[2:]
inc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include")
out = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off",
                      "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/tmp/_res.o"] + extra,
                     capture_output=True, text=True).stderr
cur, rows = {}, []
for ln in out.splitlines():
    m = re.search(r"remark: (.*?) \[-Rpass", ln)
    if not m:
        if "error" in ln:
            print(ln)
        continue
    t = m.group(1)
    if t.startswith("Function Name:"):
        if cur:
            rows.append(cur)
        cur = {"name": t.split(": ", 1)[1]}
    elif ": " in t:
        k, v = t.split(": ", 1)
        cur[k.strip()] = v
if cur:
    rows.append(cur)
dem = subprocess.run(["c++filt"] + [r["name"] for r in rows], capture_output=True,
                     text=True).stdout.splitlines()
for r, d in zip(rows, dem):
    d = re.sub(r"\(.*", "", d).replace("void ", "")
    print(f"{d[:64]:64s} vgpr {r.get('VGPRs', '?'):>4} agpr {r.get('AGPRs', '?'):>4} vgpr_spill {r.get('VGPRs Spill', r.get('VGPR Spill', '?')):>4} "
          f"sgpr {r.get('SGPRs', '?'):>4} sgpr_spill {r.get('SGPRs Spill', '?'):>3} scratch_B/lane {r.get('ScratchSize [bytes/lane]', '?'):>5} "
          f"waves/SIMD {r.get('Occupancy [waves/SIMD]', '?')} lds {r.get('LDS Size [bytes/block]', '?')}")

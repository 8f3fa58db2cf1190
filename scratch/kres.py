"""Compact per-kernel resource table from hipcc's kernel-resource-usage remarks.  usage: kres.py [filter] [extra hipcc flags...]"""
import re, subprocess, sys
flt = sys.argv[1] if len(sys.argv) > 1 else ""
extra = sys.argv[2:]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-slp-vectorize", "-c",
       "/root/repo/rustlight_amd/csrc/kernels/wavefront.hip", "-o", "/tmp/kres.o", "-Rpass-analysis=kernel-resource-usage", *extra]
out = subprocess.run(cmd, capture_output=True, text=True, cwd="/tmp").stderr
cur = None; rows = []
for line in out.splitlines():
    m = re.search(r"remark:\s+(.+?): (\S+) \[-R", line)
    if not m: continue
    k, v = m.group(1).strip(), m.group(2)
    if k == "Function Name" or k == "Name":
        cur = {"name": subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip().split("(")[0]}
        rows.append(cur)
    elif cur is not None: cur[k] = v
for r in rows:
    if flt in r["name"]:
        print(f'{r["name"]:60s} vgpr {r.get("VGPRs","?"):>4} agpr {r.get("AGPRs","?"):>4} spill {r.get("VGPR Spill", r.get("VGPRs Spill","?")):>4} scratch {r.get("ScratchSize [bytes/lane]","?"):>6} occ {r.get("Occupancy [waves/SIMD]","?"):>2} lds {r.get("LDS Size [bytes/block]","?")}')

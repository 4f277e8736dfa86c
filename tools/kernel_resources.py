"""Register / LDS use of every kernel of the library (hipcc -Rpass-analysis=kernel-resource-usage), one line each, with the
workgroups of 256 threads a CU really admits: min(8, LDS, VGPR waves, floor(800 / (ceil(sgpr / 16) * 16 + 16))) — the last term
is MI355X_MICROARCH.md's admission rule (<= 80 SGPRs: 8 workgroups, 82-96: 7, 98-112: 6)."""
import re
import subprocess
import sys
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
extra = sys.argv[1:]
p = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-Rpass-analysis=kernel-resource-usage",
                    "-o", "/tmp/_kr.so", os.path.join(ROOT, "commet_amd", "csrc", "capi.hip"), "-lz"] + extra, capture_output=True, text=True)
OCC = r"Occupancy \[waves/SIMD\]"
LDS = r"LDS Size \[bytes/block\]"
for b in re.split(r"remark: Function Name: ", p.stderr)[1:]:
    name = b.split(" ")[0].split("\n")[0]
    g = lambda k: int((re.search(k + r": (\d+)", b) or [0, 0])[1])
    d = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    d = re.sub(r"^void commet::", "", re.sub(r"\(.*", "", d))
    sg = g("TotalSGPRs")
    adm = min(8, 800 // (-(-sg // 16) * 16 + 16))
    print(f"{d[:72]:72s} sgpr {sg:4d} (wg/CU by sgpr {adm}) vgpr {g('VGPRs'):4d} occ {g(OCC)} sspill {g('SGPRs Spill'):3d} vspill {g('VGPRs Spill'):3d} lds {g(LDS)}")

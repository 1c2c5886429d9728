"""Register budget of every kernel of one .hip file (hipcc -Rpass-analysis=kernel-resource-usage), one line per kernel.
   python tools/kernel_regs.py sfm-toy-library_amd/csrc/ba_kernels.hip [filter]"""
import re, subprocess, sys
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
out = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-munsafe-fp-atomics", "-Wno-unused-function", *__import__("os").environ.get("EXTRA","").split(),
                      "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage", "-c", "-o", "/dev/null", src], capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"remark: Function Name: (\S+)", line)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip().split("(")[0]
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+(VGPRs|AGPRs|Occupancy \[waves/SIMD\]|VGPRs Spill|SGPRs Spill|LDS Size \[bytes/block\]|TotalSGPRs|ScratchSize \[bytes/lane\]): (\d+)", line)
    if m and cur:
        rows[cur][m.group(1)] = int(m.group(2))
for k, v in rows.items():
    if flt in k:
        print("%-70s vgpr %3d  agpr %3d  occ %d  spill %3d  scratch %4d  lds %6d" % (k[-70:], v.get("VGPRs", 0), v.get("AGPRs", 0), v.get("Occupancy [waves/SIMD]", 0),
              v.get("VGPRs Spill", 0), v.get("ScratchSize [bytes/lane]", 0), v.get("LDS Size [bytes/block]", 0)))

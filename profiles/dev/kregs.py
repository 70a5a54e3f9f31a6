#!/usr/bin/env python3
"""per-kernel register / LDS / scratch use of a HIP object: kregs.py file.o [name-filter]"""
import re, subprocess, sys
B = "/opt/rocm/lib/llvm/bin"
o = sys.argv[1]
subprocess.run([f"{B}/llvm-objcopy", "--dump-section", ".hip_fatbin=/tmp/kregs.fat", o], check=True)
lst = subprocess.run([f"{B}/clang-offload-bundler", "--list", "--type=o", "--input=/tmp/kregs.fat"], capture_output=True, text=True).stdout.split()
t = [x for x in lst if "amdgcn" in x][0]
subprocess.run([f"{B}/clang-offload-bundler", "--unbundle", "--type=o", "--input=/tmp/kregs.fat", f"--targets={t}", "--output=/tmp/kregs.co"], check=True)
notes = subprocess.run([f"{B}/llvm-readelf", "--notes", "/tmp/kregs.co"], capture_output=True, text=True).stdout
for blk in notes.split("  - .agpr_count")[1:]:
    g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]
    name = g("name")
    if len(sys.argv) > 2 and sys.argv[2] not in name:
        continue
    print("%-70s vgpr %4s sgpr %3s lds %6s scratch %s spill %s" % (name[:70], g("vgpr_count"), g("sgpr_count"), g("group_segment_fixed_size"), g("private_segment_fixed_size"), g("vgpr_spill_count")))

#!/usr/bin/env python3
"""Registers, spills and scratch of the gfx950 kernels in one object file:
`python tools/kernel_regs.py search_engine_amd/csrc/build/vs_kernels.o [name filter]`."""
import os
import re
import subprocess
import sys
import tempfile

B = "/opt/rocm/lib/llvm/bin/"
obj, flt = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
with tempfile.TemporaryDirectory() as d:
    fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "k.co")
    subprocess.check_call([B + "llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat, obj])
    subprocess.check_call([B + "clang-offload-bundler", "--unbundle", "--type=o", "--input=" + fat,
                           "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co])
    notes = subprocess.check_output([B + "llvm-readelf", "--notes", co], text=True)
for blk in notes.split("- .agpr_count")[1:]:
    name = re.search(r"\.name:\s+(\S+)", blk).group(1)
    if flt in name:
        g = lambda key: re.search(key + r":\s+(\d+)", blk).group(1)
        demangled = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()[:90]
        agpr = re.match(r":\s+(\d+)", blk).group(1)
        print(demangled + ": vgpr " + g(r"\.vgpr_count") + " agpr " + agpr + " spills " + g(r"\.vgpr_spill_count")
              + " sgpr " + g(r"\.sgpr_count") + " scratch " + g(r"\.private_segment_fixed_size")
              + " lds " + g(r"\.group_segment_fixed_size"))

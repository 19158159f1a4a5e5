#!/usr/bin/env python3
"""The order of matrix, LDS, LDS-DMA, wait and barrier instructions in one kernel of a gfx950 object:
`python tools/isa_shape.py build/vs_batch.o scan_batch_stream_kernel`. M = v_mfma, r = ds_read,
W = ds_write, D = buffer_load ... lds, g = other global/buffer load, s = store, w[..] = s_waitcnt."""
import os
import re
import subprocess
import sys
import tempfile

B = "/opt/rocm/lib/llvm/bin/"
obj, name = sys.argv[1], sys.argv[2]
with tempfile.TemporaryDirectory() as d:
    fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "k.co")
    subprocess.check_call([B + "llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat, obj])
    subprocess.check_call([B + "clang-offload-bundler", "--unbundle", "--type=o", "--input=" + fat,
                           "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co])
    text = subprocess.check_output([B + "llvm-objdump", "-d", co], text=True)
m = re.search(r"^[0-9a-f]+ <(\S*" + re.escape(name) + r"\S*)>:\n(.*?)(?=^[0-9a-f]+ <\S+>:\n|\Z)", text, re.S | re.M)
seq = []
for line in m.group(2).split("\n"):
    mm = re.match(r"\s+(\S+)", line)
    if not mm:
        continue
    op = mm.group(1)
    if op.startswith("v_mfma"):
        seq.append("M")
    elif op.startswith("ds_read") or op.startswith("ds_load"):
        seq.append("r")
    elif op.startswith("ds_write") or op.startswith("ds_store"):
        seq.append("W")
    elif op.startswith("ds_"):
        seq.append("a")
    elif op.startswith("buffer_load") and " lds" in line:
        seq.append("D")
    elif op.startswith(("buffer_load", "global_load", "flat_load")):
        seq.append("g")
    elif op.startswith(("buffer_store", "global_store", "flat_store")):
        seq.append("s")
    elif op.startswith("s_waitcnt"):
        seq.append("w[" + line.split("s_waitcnt")[1].split("//")[0].strip() + "]")
    elif op.startswith("s_barrier"):
        seq.append("|BAR|")
    elif op.startswith(("s_cbranch", "s_branch")):
        seq.append("<br>")
    elif re.match(r"^<\S+>:", line.strip()):
        seq.append("\n")
print(m.group(1))
print("".join(seq))

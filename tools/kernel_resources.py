#!/usr/bin/env python3
"""VGPRs / SGPRs / scratch / LDS of every kernel of a built library, from the code object's notes (dev tool):
    python tools/kernel_resources.py [mrcal_amd/libmrcal_amd.so] [substring ...]
The fat binary's gfx950 code object is carved out of the .so (its ELF header inside the .hip_fatbin section)"""
import subprocess, sys, re, tempfile, os
so = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mrcal_amd", "libmrcal_amd.so")
pats = sys.argv[2:]
data = open(so, "rb").read()
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
seen = False
pos = 0
while True:
    i = data.find(b"\x7fELF\x02\x01\x01\x40", pos)        # ELF64, little endian, OS ABI 64 = AMDGPU HSA
    if i < 0: break
    pos = i + 4
    # section header table offset + count*size bound the object
    import struct
    e_shoff = struct.unpack_from("<Q", data, i + 0x28)[0]
    e_shentsize, e_shnum = struct.unpack_from("<HH", data, i + 0x3A)
    end = i + e_shoff + e_shentsize*e_shnum
    with tempfile.NamedTemporaryFile(suffix=".co", delete=False) as f:
        f.write(data[i:end]); name = f.name
    txt = subprocess.run([READELF, "--notes", name], capture_output=True, text=True).stdout
    os.unlink(name)
    for m in re.finditer(r"- \.agpr_count:.*?(?=\n\s+- \.agpr_count:|\namdhsa\.target|\Z)", txt, re.S):
        blk = m.group(0)
        g = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, blk) or [None, "?"])[1]
        nm = g("name")
        try: nm = subprocess.run(["c++filt", nm], capture_output=True, text=True).stdout.strip()
        except Exception: pass
        if pats and not any(p in nm for p in pats): continue
        seen = True
        print(f"vgpr {g('vgpr_count'):>4} agpr {g('agpr_count'):>3} sgpr {g('sgpr_count'):>4} scratch {g('private_segment_fixed_size'):>5} lds {g('group_segment_fixed_size'):>7}  {nm[:110]}")
if not seen: print("no kernels found (patterns: %s)" % pats)

"""Register / spill / scratch metadata of the kernels of a built libloik_amd.so (CPU only): the code objects' notes through llvm-readelf.
usage: python scripts/r06/kernel_meta.py [lib] [name-substring ...]"""
import os, re, subprocess, sys, tempfile
LLVM = "/opt/rocm/lib/llvm/bin"
lib = sys.argv[1] if len(sys.argv) > 1 and os.path.exists(sys.argv[1]) else os.path.join(os.path.dirname(__file__), "..", "..", "loik_amd", "lib", "libloik_amd.so")
pats = [a for a in sys.argv[1:] if not os.path.exists(a)]
tmp = tempfile.mkdtemp()
# the fat binary's bundles: one per translation unit
out = subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--list", "--type=o", "--input=" + lib], capture_output=True, text=True)
data = open(lib, "rb").read()
# (bundles inside .hip_fatbin: find the embedded ELFs for amdgcn by scanning for the ELF magic with the AMDGPU machine id)
idx = 0; n = 0
while True:
    i = data.find(b"\x7fELF\x02\x01\x01\x40", idx)   # ELFCLASS64, little endian, OSABI = AMDGPU_HSA (64)
    if i < 0: break
    import struct
    shoff = struct.unpack_from("<Q", data, i + 0x28)[0]
    shentsize, shnum = struct.unpack_from("<HH", data, i + 0x3A)
    size = shoff + shentsize * shnum
    path = os.path.join(tmp, "co%d.elf" % n)
    open(path, "wb").write(data[i:i + size])
    txt = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", path], capture_output=True, text=True).stdout
    for blk in txt.split("- .agpr_count:")[1:]:
        g = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, blk) or [None, "?"])[1]
        name = g("name")
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        if pats and not any(p in dem for p in pats): continue
        print("%-70s vgpr %3s spill %3s  sgpr %3s spill %3s  scratch %5s B  lds %s" % (dem[:70], g("vgpr_count"), g("vgpr_spill_count"), g("sgpr_count"), g("sgpr_spill_count"), g("private_segment_fixed_size"), g("group_segment_fixed_size")))
    idx = i + 8; n += 1

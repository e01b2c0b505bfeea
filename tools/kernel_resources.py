"""Register / scratch / LDS use of every kernel in a built object: python tools/kernel_resources.py [mlp_fused] [filter]
(unbundles the gfx950 code object from moda_amd/lib/<name>.o and reads its amdhsa metadata notes)."""
import os, re, struct, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
name = sys.argv[1] if len(sys.argv) > 1 else "mlp_fused"
flt = sys.argv[2] if len(sys.argv) > 2 else ""
LLVM = "/opt/rocm/lib/llvm/bin"
with tempfile.TemporaryDirectory() as d:
    fb = os.path.join(d, "fatbin.bin")
    subprocess.check_call([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin",
                           os.path.join(ROOT, "moda_amd", "lib", name + ".o"), fb])
    b = open(fb, "rb").read()
    assert b[:24] == b"__CLANG_OFFLOAD_BUNDLE__"
    n = struct.unpack("<Q", b[24:32])[0]
    off = 32
    co = None
    for _ in range(n):
        o, sz, tl = struct.unpack("<QQQ", b[off:off + 24]); off += 24
        t = b[off:off + tl].decode(); off += tl
        if "gfx950" in t:
            co = os.path.join(d, "k.co")
            open(co, "wb").write(b[o:o + sz])
    txt = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
for k in re.split(r"\n\s+- \.agpr_count:", txt)[1:]:
    nm = re.search(r"\.name:\s+(\S+)", k).group(1)
    nm = subprocess.run(["c++filt", nm], capture_output=True, text=True).stdout.strip().replace("(anonymous namespace)::", "")
    if flt and flt not in nm:
        continue
    g = lambda f: re.search(r"\.%s:\s+(\d+)" % f, k).group(1)
    print(f"{nm[:120]:120s} vgpr {g('vgpr_count'):>4s} agpr {k.split()[0]:>4s} scratch {g('private_segment_fixed_size'):>5s} "
          f"vspill {g('vgpr_spill_count'):>4s} sgpr {g('sgpr_count'):>4s}")

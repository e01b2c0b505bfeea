"""ISA audit of the asm-owned-AGPR kernels (PrecBF16A, moda_amd/csrc/mlp_fused.hip).  hipcc neither schedules nor pads what is inside an
asm statement and knows nothing of the literally named AGPRs, so after every build the machine code is checked for what it must not
contain.  Input: the built library / object (the gfx950 code object is unbundled and disassembled with llvm-objdump), or a -save-temps
.s file.  Per kernel whose name contains the pattern:
  1. no scratch (private segment 0, no scratch_* instruction);
  2. the accumulator file is touched by nothing but this code's own statements: every AGPR access is a v_accvgpr_write_b32 behind
     its v_cvt_pk / v_pk_max, or an MFMA B operand -- no v_accvgpr_read / v_accvgpr_mov, no AGPR operand anywhere else;
  3. nothing but MFMAs touches an MFMA's destination tile before it can have landed (12 issue slots; an MFMA counts 8);
  4. no VALU instruction writes a VGPR that an MFMA reads (A, B or C) fewer than 2 wait states later.
usage: python tools/agpr_audit.py [moda_amd/lib/libmoda_hip.so | file.o | file.s] [name pattern]      exit code 1 on a finding"""
import os, re, struct, subprocess, sys, tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def disassemble(path):
    """{mangled kernel name: [instruction text, ...]} of the gfx950 code object bundled in a host object / shared library."""
    with tempfile.TemporaryDirectory() as d:
        fb = os.path.join(d, "fatbin.bin")
        subprocess.check_call([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", path, fb])
        b = open(fb, "rb").read()
        kernels = {}
        pos = 0
        while True:                                  # a linked library concatenates one bundle per translation unit
            pos = b.find(b"__CLANG_OFFLOAD_BUNDLE__", pos)
            if pos < 0:
                break
            n = struct.unpack("<Q", b[pos + 24:pos + 32])[0]
            off = pos + 32
            for _ in range(n):
                o, sz, tl = struct.unpack("<QQQ", b[off:off + 24]); off += 24
                t = b[off:off + tl].decode(); off += tl
                if "gfx950" in t and sz:
                    co = os.path.join(d, "k.co")
                    open(co, "wb").write(b[pos + o:pos + o + sz])
                    txt = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", co], capture_output=True, text=True).stdout
                    notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
                    cur = None
                    for ln in txt.splitlines():
                        m = re.match(r"^[0-9a-f]+ <(\S+)>:", ln)
                        if m:
                            cur = m.group(1); kernels[cur] = {"ins": [], "scratch": None}
                        elif cur and ln.startswith(("\t", " ")) and ln.strip():
                            t2 = ln.strip().split("//")[0].strip()
                            if t2:
                                kernels[cur]["ins"].append(t2)
                    for k in re.split(r"\n\s+- \.agpr_count:", notes)[1:]:
                        nm = re.search(r"\.name:\s+(\S+)", k).group(1)
                        if nm in kernels:
                            kernels[nm]["scratch"] = int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", k).group(1))
                            kernels[nm]["vgpr"] = int(re.search(r"\.vgpr_count:\s+(\d+)", k).group(1))
                            kernels[nm]["agpr"] = int(k.split()[0])
            pos += 24
    return kernels


def from_asm(path, pat):
    lines = open(path).read().splitlines()
    out, i = {}, 0
    while i < len(lines):
        m = re.match(r"^(_Z\S*):\s", lines[i])
        if m and pat in m.group(1) and "mlp_fused_kernel" in m.group(1):
            j, body = i + 1, []
            while j < len(lines) and not lines[j].startswith("\t.end_amdhsa_kernel"):
                body.append(lines[j]); j += 1
            ins = [t.strip() for t in body if t.strip() and not t.strip().startswith((";", ".")) and not t.strip().endswith(":")]
            sc = next((int(l.split()[-1]) for l in body if "amdhsa_private_segment_fixed_size" in l), None)
            out[m.group(1)] = {"ins": ins, "scratch": sc}
            i = j
        i += 1
    return out


def vregs(text):
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", text):
        if m.group(1):
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def audit(ins, scratch):
    """-> list of findings (empty: clean)."""
    bad = []
    if scratch:
        bad.append(f"scratch: {scratch} bytes")
    ops = [(t.split()[0], t.split(None, 1)[1] if " " in t else "") for t in ins]
    for k, (op, args) in enumerate(ops):
        if op.startswith("scratch_"):
            bad.append(f"scratch access: {ins[k]}")
        if op in ("v_accvgpr_read_b32", "v_accvgpr_mov_b32"):
            bad.append(f"AGPR access that is not this code's: {ins[k]}")
        elif op == "v_accvgpr_write_b32":
            src = args.split(",")[1].strip()
            prev = " ".join(ins[max(0, k - 2):k])
            if not re.search(r"(v_pk_max_i16|v_cvt_pk_(bf16|f16)_f32) %s\b" % re.escape(src), prev):
                bad.append(f"v_accvgpr_write_b32 that is not an epilogue piece: {ins[k]}")
        elif re.search(r"\ba\[?\d", args) and not op.startswith("v_mfma"):
            bad.append(f"AGPR operand outside an MFMA: {ins[k]}")
        if op.startswith("v_mfma"):
            dm = re.match(r"v\[(\d+):(\d+)\]", args)
            if dm:                                    # 3: its destination tile until it has landed
                dst = set(range(int(dm.group(1)), int(dm.group(2)) + 1))
                slots = 0
                for k2 in range(k + 1, min(k + 40, len(ops))):
                    op2, a2 = ops[k2]
                    if op2.startswith("v_mfma"):
                        slots += 8
                    elif op2 == "s_nop":
                        slots += int(a2) + 1
                    else:
                        if slots < 12 and not op2.startswith("s_") and (vregs(a2) & dst):
                            bad.append(f"{ins[k2]}  touches the destination of  {ins[k]}  {slots} slots behind it")
                        slots += 1
                    if slots >= 12:
                        break
            src = vregs(args)                         # 4: VALU writes in front of it
            slots = 0
            for k2 in range(k - 1, max(-1, k - 7), -1):
                op2, a2 = ops[k2]
                if op2 == "s_nop":
                    slots += int(a2) + 1
                elif op2.startswith("v_mfma"):
                    break
                else:
                    if op2.startswith("v_") and not op2.startswith("v_accvgpr") and slots < 2:
                        dm2 = re.match(r"(v\[\d+:\d+\]|v\d+)", a2)
                        if dm2 and (vregs(dm2.group(1)) & src):
                            bad.append(f"{ins[k2]}  writes an operand of  {ins[k]}  {slots} wait states ahead of it")
                    slots += 1
                if slots >= 2:
                    break
    return bad


def main(path, pat="PrecBF16A"):
    ks = from_asm(path, pat) if path.endswith(".s") else {k: v for k, v in disassemble(path).items() if pat in k and "mlp_fused_kernel" in k}
    ok = bool(ks)
    if not ks:
        print("no kernel matches", pat)
    for name, k in sorted(ks.items()):
        ins = k["ins"]
        mix = {}
        for t in ins:
            mix[t.split()[0]] = mix.get(t.split()[0], 0) + 1
        findings = audit(ins, k["scratch"])
        print(name[:120])
        print("   vgpr", k.get("vgpr"), "agpr", k.get("agpr"), "scratch", k["scratch"], " mfma", sum(v for o, v in mix.items() if o.startswith("v_mfma")),
              " v_accvgpr_write", mix.get("v_accvgpr_write_b32", 0), " ds_read_b128", mix.get("ds_read_b128", 0), " s_nop", mix.get("s_nop", 0))
        for f in findings[:8]:
            print("   FINDING:", f)
        ok = ok and not findings
    print("AUDIT", "OK" if ok else "FAILED")
    return ok


if __name__ == "__main__":
    p = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "moda_amd", "lib", "libmoda_hip.so")
    sys.exit(0 if main(p, *(sys.argv[2:3])) else 1)

"""ISA audit of the asm-owned-AGPR kernels (PrecBF16A / PrecF16A, moda_amd/csrc/mlp_fused.hip).  hipcc neither schedules nor pads what
is inside an asm statement and knows nothing of the literally named AGPRs, so after EVERY build (moda_amd.build.build runs this and
stamps the result; a failed audit switches the dispatch to the compiler-scheduled eight-wave form) the machine code is checked for
what it must not contain.  Input: the built library / object (the gfx950 code object is unbundled and disassembled with
llvm-objdump), or a -save-temps .s file.  Per kernel whose name contains the pattern:
  1. no scratch (private segment 0, no scratch_* instruction);
  2. the accumulator file is touched by nothing but this code's own statements: every AGPR access is a v_accvgpr_write_b32 behind
     its v_cvt_pk / v_pk_max, or an MFMA B operand -- no v_accvgpr_read / v_accvgpr_mov, no AGPR operand anywhere else;
  3. nothing but MFMAs touches an MFMA's destination tile before it can have landed (12 issue slots; an MFMA counts 8);
  4. GENERIC operand rule: no vector instruction (VALU, v_accvgpr_write included; register file irrelevant) writes a register that
     an MFMA reads as A, B or C fewer than 2 wait states later;
  5. GENERIC forwarding rule: no MFMA reads as A or B (or as a C that is not exactly its predecessor's whole destination) a register
     of the destination of an MFMA issued fewer than 12 issue slots earlier.
The wait-state numbers are the gfx950 tables' (cdna_hip_programming.md 5.7 item 2: written operand -> MFMA `s_nop 1`; 8-pass XDL
result -> any other use 12 states)."""
import os, re, struct, subprocess, tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
PATTERNS = ("PrecBF16A", "PrecF16A")          # the kernels whose schedule is this code's own, not hipcc's


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


def regs(text):
    """{('v' | 'a', index)} of every vector register named in an operand string."""
    out = set()
    for m in re.finditer(r"\b([va])\[(\d+):(\d+)\]|\b([va])\[(\d+)\]|\b([va])(\d+)\b", text):
        if m.group(1):
            out.update((m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1))
        elif m.group(4):
            out.add((m.group(4), int(m.group(5))))
        else:
            out.add((m.group(6), int(m.group(7))))
    return out


def vregs(text):
    return {i for f, i in regs(text) if f == "v"}


def _operands(args):
    """Top-level comma-separated operands of an instruction ('v[0:15], v[16:19], a[0:3], v[0:15] cbsz:1' -> 4 + modifiers)."""
    return [a.strip().split()[0] for a in args.split(",") if a.strip()]


MFMA_RESULT_SLOTS = 12      # 8-pass XDL (32x32x16 / 32x32x8 forms): issue slots before anything but a whole-tile accumulate may use D
OPERAND_WAIT = 2            # written register -> MFMA operand read


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
        if not op.startswith("v_mfma"):
            continue
        opr = _operands(args)
        dst = regs(opr[0]) if opr else set()
        srcs = [regs(o) for o in opr[1:4]]                # A, B, C
        # 3: the destination tile until it has landed -- anything but an MFMA that touches it
        slots = 0
        for k2 in range(k + 1, min(k + 40, len(ops))):
            op2, a2 = ops[k2]
            if op2.startswith("v_mfma"):
                slots += 8
            elif op2 == "s_nop":
                slots += int(a2) + 1
            else:
                if slots < MFMA_RESULT_SLOTS and not op2.startswith("s_") and (regs(a2) & dst):
                    bad.append(f"{ins[k2]}  touches the destination of  {ins[k]}  {slots} slots behind it")
                slots += 1
            if slots >= MFMA_RESULT_SLOTS:
                break
        # 4 (generic): ANY vector write of ANY register this MFMA reads, fewer than OPERAND_WAIT states ahead of it
        read = set().union(*srcs) if srcs else set()
        slots = 0
        for k2 in range(k - 1, max(-1, k - 8), -1):
            op2, a2 = ops[k2]
            if op2 == "s_nop":
                slots += int(a2) + 1
            elif op2.startswith("v_mfma"):
                break                                      # (MFMA -> MFMA: rule 5)
            else:
                if op2.startswith(("v_", "ds_read", "ds_load", "global_load", "buffer_load")) and slots < OPERAND_WAIT:
                    o2 = _operands(a2)
                    if o2 and (regs(o2[0]) & read) and op2.startswith("v_"):
                        bad.append(f"{ins[k2]}  writes an operand of  {ins[k]}  {slots} wait states ahead of it")
                slots += 1
            if slots >= OPERAND_WAIT:
                break
        # 5 (generic): a recent MFMA's destination read as A / B, or as a C that is not exactly that whole destination
        slots = 0
        for k2 in range(k - 1, max(-1, k - 40), -1):
            op2, a2 = ops[k2]
            if op2 == "s_nop":
                slots += int(a2) + 1
            elif op2.startswith("v_mfma"):
                if slots < MFMA_RESULT_SLOTS:
                    d2 = regs(_operands(a2)[0])
                    if d2 & (srcs[0] | srcs[1] if len(srcs) > 1 else set()):
                        bad.append(f"{ins[k]}  reads as A/B the destination of  {ins[k2]}  {slots} slots behind it")
                    if len(srcs) > 2 and (d2 & srcs[2]) and d2 != srcs[2]:
                        bad.append(f"{ins[k]}  accumulates onto part of the destination of  {ins[k2]}  {slots} slots behind it")
                slots += 8
            else:
                slots += 1
            if slots >= MFMA_RESULT_SLOTS:
                break
    return bad


def audit_library(path, patterns=PATTERNS):
    """{kernel name: [findings]} for every mlp_fused_kernel instantiation of `path` whose name contains one of `patterns`."""
    out = {}
    for name, k in disassemble(path).items():
        if "mlp_fused_kernel" in name and any(p in name for p in patterns):
            out[name] = audit(k["ins"], k["scratch"])
    return out

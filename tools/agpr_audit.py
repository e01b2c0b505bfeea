"""ISA audit of the asm-owned-AGPR kernels (PrecBF16A, mlp_fused.hip): the compiler must not touch the accumulator file itself.
For every kernel whose name contains the pattern: registers, scratch, and every v_accvgpr_* / AGPR operand OUTSIDE
;;#ASMSTART ... ;;#ASMEND (must be none), plus the instruction mix.   usage: agpr_audit.py <file.s> [name pattern]"""
import re, sys

path, pat = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "PrecBF16A")
lines = open(path).read().splitlines()
i = 0
ok = True
while i < len(lines):
    m = re.match(r"^(_Z\S*):\s", lines[i])
    if m and pat in m.group(1) and "mlp_fused_kernel" in m.group(1):
        name = m.group(1)
        j = i + 1
        body = []
        while j < len(lines) and not lines[j].startswith("\t.end_amdhsa_kernel"):
            body.append(lines[j]); j += 1
        inside, bad, mix = False, [], {}
        for ln in body:
            t = ln.strip()
            if t.startswith(";;#ASMSTART"): inside = True; continue
            if t.startswith(";;#ASMEND"): inside = False; continue
            if not t or t.startswith((";", ".")) or t.endswith(":"): continue
            op = t.split()[0]
            mix[op] = mix.get(op, 0) + 1
            if not inside and (op.startswith("v_accvgpr") or re.search(r"\ba\[?\d", t)):
                bad.append(t)
        # an asm MFMA's destination tile must not be read or written by anything but the next MFMAs until it has landed:
        # flag any non-MFMA instruction that names one of its registers within the next WINDOW issue slots (an intervening MFMA
        # counts 8: it occupies the pipe for 32 cycles)
        WINDOW = 12
        flat = []
        ins = False
        for ln in body:
            t = ln.strip()
            if t.startswith(";;#ASMSTART"): ins = True; continue
            if t.startswith(";;#ASMEND"): ins = False; continue
            if not t or t.startswith((";", ".")) or t.endswith(":"): continue
            flat.append((t, ins))

        def regs_of(text):
            out = set()
            for m2 in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", text):
                if m2.group(1):
                    out.update(range(int(m2.group(1)), int(m2.group(2)) + 1))
                else:
                    out.add(int(m2.group(3)))
            return out
        early = []
        for k, (t, ins_) in enumerate(flat):
            if not (ins_ and t.startswith("v_mfma")):
                continue
            dm = re.match(r"v_mfma\S+\s+v\[(\d+):(\d+)\]", t)
            if not dm:
                continue
            dst = set(range(int(dm.group(1)), int(dm.group(2)) + 1))
            slots = 0
            for t2, ins2 in flat[k + 1:k + 40]:
                if t2.startswith("v_mfma"):
                    slots += 8
                elif t2.startswith("s_nop"):
                    slots += int(t2.split()[1]) + 1
                else:
                    if slots < WINDOW and (regs_of(t2.split(None, 1)[1] if " " in t2 else "") & dst) and not t2.startswith("s_"):
                        early.append((t, t2, slots))
                    slots += 1
                if slots >= WINDOW:
                    break
        print("   instructions touching an asm MFMA's destination before it can have landed:", len(early), early[:4])
        ok = ok and not early
        # the other direction: a VGPR written by a compiler-generated VALU instruction is read by an asm MFMA (A, B or C operand)
        # fewer than 2 wait states later (LegacyVALUWritesVGPR -> MFMA read: the hazard recogniser pads builtins, not asm)
        late = []
        for k, (t, ins_) in enumerate(flat):
            if not (ins_ and t.startswith("v_mfma")):
                continue
            src = regs_of(t.split(None, 1)[1])
            slots = 0
            for t2, ins2 in reversed(flat[max(0, k - 6):k]):
                if t2.startswith("s_nop"):
                    slots += int(t2.split()[1]) + 1
                elif t2.startswith("v_mfma"):
                    break
                elif t2.startswith("v_") and not ins2:
                    dm2 = re.match(r"v_\S+\s+(v\[\d+:\d+\]|v\d+)", t2)
                    if dm2 and slots < 2 and (regs_of(dm2.group(1)) & src):
                        late.append((t2, t, slots))
                    slots += 1
                else:
                    slots += 1
                if slots >= 2:
                    break
        print("   compiler VALU writes read by an asm MFMA within 2 wait states:", len(late), late[:6])
        ok = ok and not late
        info = {k: next((l.split()[-1] for l in body if k in l), "?") for k in ("amdhsa_next_free_vgpr", "amdhsa_accum_offset", "amdhsa_private_segment_fixed_size")}
        print(name[:110]); print("  ", info)
        keys = ("v_mfma_f32_32x32x16_bf16", "v_cvt_pk_bf16_f32", "v_pk_max_i16", "v_accvgpr_write_b32", "ds_read_b128", "s_nop", "s_waitcnt", "s_barrier", "v_mov_b32_e32", "scratch_load_dword", "scratch_store_dword")
        print("  ", {k: mix.get(k, 0) for k in keys})
        print("   compiler-generated AGPR accesses:", len(bad), bad[:5])
        ok = ok and not bad and info["amdhsa_private_segment_fixed_size"] == "0"
        i = j
    i += 1
print("AUDIT", "OK" if ok else "FAILED")
sys.exit(0 if ok else 1)

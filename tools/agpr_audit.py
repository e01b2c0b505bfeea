"""ISA audit of the asm-owned-AGPR kernels: command-line front of moda_amd/isa_audit.py (the rules live there; the build runs them).
usage: python tools/agpr_audit.py [moda_amd/lib/libmoda_hip.so | file.o | file.s] [name pattern]      exit code 1 on a finding"""
import os, sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from moda_amd.isa_audit import audit, disassemble, from_asm, regs, vregs  # noqa: E402,F401


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

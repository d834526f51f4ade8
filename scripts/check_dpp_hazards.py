"""
Scan gfx950 assembly (hipcc -save-temps .s, or llvm-objdump -d output) of the row kernels for the one hazard the compiler cannot
see inside asm statements: a DPP instruction whose broadcast operand (src0) was written by a VALU instruction in one of the
two preceding issue slots (s_nop N counts N + 1 slots).  Also prints registers, scratch and the instruction mix per kernel.

    python3 scripts/check_dpp_hazards.py markovflow_amd/csrc/build/mf_inst-hip-amdgcn-amd-amdhsa-gfx950.s [name-filter]
Exit code 1 when a hazard is found.
"""
import re
import sys
from collections import Counter


def regs_of(tok):
    tok = tok.strip().lstrip("-").strip("|")
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    if m:
        return {int(m.group(1))}
    return set()


def main():
    path = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else "row"
    name, body, kernels = None, [], {}
    for line in open(path):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            name, body = m.group(1), []
            kernels[name] = body
            continue
        if name is None:
            continue
        t = line.strip()
        if t.startswith(".amdhsa_kernel") or t.startswith(".section") or t.startswith(".end_amdhsa_kernel"):
            name = None if t.startswith(".section") else name
        if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
            if t.startswith("; NumVgprs") or t.startswith("; ScratchSize") or t.startswith("; Occupancy") or t.startswith("; NumSgprs"):
                body.append(("meta", t))
            continue
        body.append(("ins", t.split(";")[0].strip()))
    bad = 0
    for kname, body in kernels.items():
        if flt not in kname:
            continue
        ins = [x[1] for x in body if x[0] == "ins"]
        if not ins:
            continue
        meta = [x[1] for x in body if x[0] == "meta"]
        mix = Counter()
        recent = []          # (slot distance handled by counting) list of (written regs) of the last slots
        hazards = []
        for i, t in enumerate(ins):
            op = t.split()[0]
            mix["dpp" if "dpp" in op else ("valu" if op.startswith("v_") else ("salu" if op.startswith("s_") else ("vmem" if op.startswith(("buffer_", "global_", "flat_", "scratch_")) else ("lds" if op.startswith("ds_") else "other"))))] += 1
            if op == "s_nop":
                n = int(t.split()[1], 0) + 1
                recent = (recent + [set()] * n)[-2:]
                continue
            ops = [o for o in re.split(r",\s*", t[len(op):].strip())]
            if "dpp" in op or "row_newbcast" in t:
                src0 = regs_of(ops[1].split()[0]) if len(ops) > 1 else set()
                for back, wr in enumerate(reversed(recent[-2:])):
                    if src0 & wr:
                        hazards.append((i, t, back + 1))
            if op.startswith("v_") and ops:
                wr = regs_of(ops[0].split()[0])
                recent = (recent + [wr])[-2:]
            else:
                recent = (recent + [set()])[-2:]
        print(f"{kname[:100]}\n   {len(ins)} instructions  {dict(mix)}\n   {'  '.join(meta)}")
        for i, t, back in hazards[:10]:
            print(f"   HAZARD at instruction {i} (source written {back} slot(s) before): {t}")
        bad += len(hazards)
    print("hazards:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())

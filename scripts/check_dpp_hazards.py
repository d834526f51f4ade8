"""
Scan gfx950 assembly (hipcc -save-temps .s, or llvm-objdump -d output) of the row kernels for the hazards the compiler cannot
see inside asm statements:
  * a DPP instruction whose broadcast operand (src0) was written by a VALU instruction in one of the two preceding issue slots;
  * a DPP instruction within five issue slots of a VALU write to EXEC (v_cmpx*, or a VALU instruction with `exec` as its
    destination; scalar writes - s_and_saveexec, s_mov exec - are not a DPP hazard: LLVM's GCNHazardRecognizer::checkDPPHazards
    counts VALU definitions only) - in program order AND across every branch into a label (the slots before the branch
    count, the branch itself is one).
(s_nop N counts N + 1 slots.)  Also prints registers, scratch and the instruction mix per kernel.  Part of the build:
csrc/Makefile runs it on every translation unit that holds row kernels and fails on a hazard.

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


def writes_exec(t):
    op = t.split()[0]
    if op.startswith("v_cmpx"):
        return True
    if op.startswith("v_") and len(t.split()) > 1:
        dst = t[len(op):].strip().split(",")[0].strip()
        return dst in ("exec", "exec_lo", "exec_hi")
    return False


def slots(t):
    return int(t.split()[1], 0) + 1 if t.split()[0] == "s_nop" else 1


def exec_hazards(body, need=5):
    """DPP instructions reached with fewer than `need` issue slots since a write to EXEC."""
    seq = [(k, t) for k, t in body if k in ("ins", "label")]
    labels = {t: i for i, (k, t) in enumerate(seq) if k == "label"}
    # entry[i]: the smallest number of slots since an EXEC write with which position i can be reached (None: >= need)
    entry = {}
    for i, (k, t) in enumerate(seq):
        if k == "ins" and t.split()[0] in ("s_branch", "s_cbranch_scc0", "s_cbranch_scc1", "s_cbranch_vccz", "s_cbranch_vccnz",
                                           "s_cbranch_execz", "s_cbranch_execnz"):
            tgt = t.split()[-1]
            if tgt not in labels:
                continue
            # slots between the last EXEC write before the branch and the branch (inclusive of the branch)
            dist, j = 1, i - 1
            found = None
            while j >= 0 and dist <= need:
                kk, tt = seq[j]
                if kk == "ins":
                    if writes_exec(tt):
                        found = dist
                        break
                    dist += slots(tt)
                j -= 1
            if found is not None:
                p = labels[tgt]
                entry[p] = min(entry.get(p, need), found)
    out = []
    since = need                         # slots since the last EXEC write on the fall-through path
    for i, (k, t) in enumerate(seq):
        if k == "label":
            since = min(since, entry.get(i, need))
            continue
        if ("dpp" in t.split()[0] or "row_newbcast" in t) and since < need:
            out.append((i, t + f"   [only {since} slot(s) after a write to EXEC]", since))
        if writes_exec(t):
            since = 0
        else:
            since = min(need, since + slots(t))
    return out


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
        if re.match(r"^\.?L?[A-Za-z_][\w.$]*:$", t) and not t.startswith(";"):
            body.append(("label", t[:-1]))
            continue
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
        hazards += exec_hazards(body)
        print(f"{kname[:100]}\n   {len(ins)} instructions  {dict(mix)}\n   {'  '.join(meta)}")
        for i, t, back in hazards[:10]:
            print(f"   HAZARD at instruction {i} (source written {back} slot(s) before): {t}")
        bad += len(hazards)
    print("hazards:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())

"""Instruction mix of one kernel in a gfx950 assembly file (hipcc -S --cuda-device-only): python3 scripts/isa_mix.py file.s <mangled-name-prefix>"""
import re, sys
from collections import Counter
path, prefix = sys.argv[1], sys.argv[2]
inside, mix, waits = False, Counter(), Counter()
for line in open(path):
    if re.match(r"^_Z\w+:", line):
        inside = line.startswith(prefix)
        continue
    if not inside:
        continue
    t = line.strip()
    if t.startswith("s_endpgm"):
        break
    if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
        continue
    op = t.split()[0]
    if op.startswith("v_accvgpr"): k = "v_accvgpr moves"
    elif re.match(r"v_(fma|fmac|mul|add)_f64", op): k = "fp64 arithmetic"
    elif re.match(r"v_(pk_)?(fma|fmac|mul|add|sub)_f32", op): k = "fp32 arithmetic"
    elif op.startswith(("v_rcp", "v_rsq", "v_sqrt", "v_log", "v_exp", "v_frexp", "v_ldexp")): k = "transcendental / frexp"
    elif op.startswith("v_"): k = "other VALU"
    elif op.startswith("s_waitcnt"):
        k = "s_waitcnt"; waits[t] += 1
    elif op.startswith("s_nop"): k = "s_nop"
    elif op.startswith("s_"): k = "scalar"
    elif op.startswith("ds_"): k = "LDS"
    elif "lds" in t and op.startswith("buffer_load"): k = "LDS-DMA"
    elif op.startswith(("buffer_store", "global_store")): k = "global store"
    elif op.startswith(("buffer_load", "global_load", "flat_")): k = "global load"
    else: k = "other"
    mix[k] += 1
total = sum(mix.values())
print(f"{prefix}: {total} instructions")
for k, v in mix.most_common():
    print(f"   {k:26s} {v:6d}  {100.0 * v / total:5.1f} %")
print("   s_waitcnt vmcnt forms:", {k: v for k, v in waits.items() if "vmcnt" in k})

#!/usr/bin/env python3
"""Per-basic-block instruction histogram of one kernel in a hipcc -S listing (the unrolled phases of a kernel are its
largest blocks):  python isa_blocks.py file.s kernel_symbol_substring [min_instructions] [per]
`per` = the number of samples (or other work items) one pass through a block handles: counts are also printed per item."""
import collections
import re
import sys

txt = open(sys.argv[1]).read()
key = sys.argv[2]
minins = int(sys.argv[3]) if len(sys.argv) > 3 else 200
per = float(sys.argv[4]) if len(sys.argv) > 4 else 0
m = re.search(r"^(\S*" + re.escape(key) + r"\S*):", txt, re.M)
body = txt[m.start():txt.index(".Lfunc_end", m.start())]


def cls_of(k):
    if k.startswith("v_") and "f64" in k and not k.startswith("v_cvt"): return "valu_f64"
    if k.startswith("v_cvt"): return "valu_cvt"
    if k.startswith(("v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt", "v_sin", "v_cos")): return "valu_trans"
    if k.startswith("v_pk_"): return "valu_pk"
    if k.startswith(("v_fma_f32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_mov_b32", "v_fmac_f32", "v_mac_f32",
                     "v_subrev_f32", "v_fmaak_f32", "v_fmamk_f32", "v_mul_legacy")): return "valu_fast"
    if k.startswith("v_"): return "valu_slow"
    if k.startswith("s_"): return "salu"
    if k.startswith("ds_"): return "lds"
    if k.startswith(("global_", "buffer_", "flat_", "scratch_")): return "vmem"
    return "other"


blocks, cur, name = [], [], "entry"
for l in body.split("\n"):
    t = l.strip()
    if re.match(r"^\.LBB\S+:", l):
        blocks.append((name, cur)); cur, name = [], l.split(":")[0]
        continue
    if not l.startswith("\t") or not t or t[0] in ".;":
        continue
    cur.append(t.split()[0])
blocks.append((name, cur))
COST = {"valu_fast": 2, "valu_slow": 4, "valu_f64": 4, "valu_cvt": 4, "valu_pk": 4, "valu_trans": 8}
for name, ins in blocks:
    if len(ins) < minins:
        continue
    c = collections.Counter(ins)
    cl = collections.Counter()
    for k, v in c.items():
        cl[cls_of(k)] += v
    valu = sum(v for k, v in cl.items() if k.startswith("valu"))
    clk = sum(COST[k] * v for k, v in cl.items() if k in COST)
    print(f"== {name}: {len(ins)} instructions, VALU {valu}, est. VALU clocks {clk}" +
          (f"  | per item: {len(ins) / per:.1f} instr, {valu / per:.1f} VALU, {clk / per:.0f} clk" if per else ""))
    print("   ", dict(sorted(cl.items())))
    print("   ", ", ".join(f"{k} {v}" for k, v in c.most_common(40)))

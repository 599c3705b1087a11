#!/usr/bin/env python3
"""Instruction histogram of one kernel in a hipcc -S listing: python isa_hist.py file.s kernel_symbol_substring"""
import collections
import sys

txt = open(sys.argv[1]).read()
key = sys.argv[2]
import re
m = re.search(r"^(\S*" + re.escape(key) + r"\S*):", txt, re.M)
start = m.start()
end = txt.index(".Lfunc_end", start)
ins = []
for l in txt[start:end].split("\n"):
    t = l.strip()
    if not l.startswith("\t") or not t or t[0] in ".;":
        continue
    ins.append(t.split()[0])
c = collections.Counter(ins)
print("instructions:", len(ins))
cls = collections.Counter()
for k, v in c.items():
    if k.startswith("v_") and "f64" in k: cls["valu_f64"] += v
    elif k.startswith("v_pk_"): cls["valu_pk"] += v
    elif k.startswith(("v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt", "v_sin", "v_cos")): cls["valu_trans"] += v
    elif k.startswith("v_cvt"): cls["valu_cvt"] += v
    elif k.startswith("v_"): cls["valu_other"] += v
    elif k.startswith("s_"): cls["salu"] += v
    elif k.startswith("ds_"): cls["lds"] += v
    elif k.startswith(("global_", "buffer_", "flat_", "scratch_")): cls["vmem"] += v
    else: cls["other"] += v
print(dict(cls))
for k, v in c.most_common(int(sys.argv[3]) if len(sys.argv) > 3 else 50):
    print(f"{v:6d} {k}")

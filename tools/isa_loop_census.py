"""Instruction-class census of the innermost loop of one kernel in a hipcc -save-temps .s file.
usage: python tools/isa_loop_census.py FILE.s KERNEL_NAME_SUBSTRING"""
import collections
import re
import sys

s = open(sys.argv[1]).read()
names = [l.split(":")[0] for l in s.splitlines() if re.match(r"^_Z\w+:", l) and sys.argv[2] in l]
for name in names[:3]:
    i = s.index(name + ":")
    j = s.index(".Lfunc_end", i)
    body = s[i:j].splitlines()
    hdr = [k for k, l in enumerate(body) if "Loop Header" in l]
    if not hdr:
        print(name, "no loop")
        continue
    for st in hdr[:3]:
        lab = body[st].split(":")[0]
        ends = [k for k, l in enumerate(body) if lab in l and "s_cbranch" in l and k > st]
        if not ends:
            continue
        loop = body[st:max(ends) + 1]
        cnt = collections.Counter()
        for l in loop:
            l = l.strip()
            if not l or l.startswith(";") or l.startswith("."):
                continue
            op = l.split()[0]
            for pre, key in (("v_mfma", "mfma"), ("ds_read", "ds_read"), ("ds_write", "ds_write"), ("buffer_load", "buffer_load"), ("global_load", "global_load"), ("global_store", "global_store"),
                             ("buffer_store", "buffer_store"), ("v_", "valu"), ("s_nop", "s_nop"), ("s_waitcnt", "s_waitcnt"), ("s_barrier", "s_barrier"), ("s_cbranch", "branch"), ("s_branch", "branch"), ("s_", "salu")):
                if op.startswith(pre):
                    cnt[key] += 1
                    break
            else:
                cnt[op] += 1
        print(f"{name[:110]}  loop@{st} ({len(loop)} lines): {dict(cnt)}")

"""Instruction-class census per basic block of a kernel in a hipcc -save-temps .s file: every block that holds MFMAs, with its mix.
usage: python tools/isa_loop_census.py FILE.s KERNEL_NAME_SUBSTRING [min_mfma]"""
import collections
import re
import sys

CLASSES = (("v_mfma", "mfma"), ("ds_read", "ds_read"), ("ds_write", "ds_write"), ("ds_", "ds_other"), ("buffer_load", "vmem_ld"), ("global_load", "vmem_ld"),
           ("buffer_store", "vmem_st"), ("global_store", "vmem_st"), ("global_atomic", "atomic"), ("v_exp", "trans"), ("v_rcp", "trans"), ("v_accvgpr", "acc_mov"),
           ("v_", "valu"), ("s_nop", "s_nop"), ("s_waitcnt", "s_waitcnt"), ("s_barrier", "s_barrier"), ("s_cbranch", "branch"), ("s_branch", "branch"), ("s_", "salu"))

lines = open(sys.argv[1]).read().splitlines()
min_mfma = int(sys.argv[3]) if len(sys.argv) > 3 else 8
starts = [k for k, l in enumerate(lines) if re.match(r"^_Z\w+:", l) and sys.argv[2] in l]
for idx in starts[:4]:
    end = next(k for k in range(idx, len(lines)) if lines[k].startswith(".Lfunc_end"))
    print(lines[idx].split(":")[0][:140])
    blocks, cur, name = [], collections.Counter(), "entry"
    for l in lines[idx + 1:end]:
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            blocks.append((name, cur))
            cur, name = collections.Counter(), m.group(1) + (" LOOP" if "Loop Header" in l else "")
            continue
        t = l.strip()
        if not t or t.startswith(";") or t.startswith("."):
            continue
        op = t.split()[0]
        for pre, key in CLASSES:
            if op.startswith(pre):
                cur[key] += 1
                break
        else:
            cur["other"] += 1
    blocks.append((name, cur))
    tot = collections.Counter()
    for n, c in blocks:
        tot.update(c)
        if c["mfma"] >= min_mfma:
            print(f"   {n:<18s} {dict(c)}")
    print(f"   whole kernel: {dict(tot)}")

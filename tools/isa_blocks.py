#!/usr/bin/env python3
"""Per-basic-block instruction histogram of one kernel in the saved ISA (make -C csrc asm).
  python tools/isa_blocks.py <file.s> <substring of the kernel symbol> [min_block_size]"""
import collections, re, sys
path, key = sys.argv[1], sys.argv[2]
minsz = int(sys.argv[3]) if len(sys.argv) > 3 else 40
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_ZN3crn12sense_kernel") and key in l and l.rstrip().split(":")[0].endswith("E"))
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith(".Lfunc_end"))
blocks, cur, name = [], [], "entry"
for l in lines[start + 1:end]:
    s = l.strip()
    if not s or s.startswith(";") or s.startswith("."):
        if re.match(r"^\.LBB\d+_\d+:", s):
            blocks.append((name, cur)); cur, name = [], s.split(":")[0]
        continue
    cur.append(s.split()[0])
    if s.startswith(("s_cbranch", "s_branch")):   # a branch ends the basic block even without a label after it
        blocks.append((name, cur)); cur, name = [], name + "+"
blocks.append((name, cur))
def cls(op):
    if op.startswith("v_pk_add"): return "pk_add"
    if op.startswith("v_pk_mul"): return "pk_mul"
    if op.startswith("v_pk_fma"): return "pk_fma"
    if op.startswith("v_pk_mov") or op.startswith("v_mov") or op.startswith("v_accvgpr"): return "mov"
    if op.startswith("v_fma") or op.startswith("v_fmac"): return "fma"
    if op.startswith("v_mul_f32"): return "mul"
    if op.startswith("v_"): return "v_other"
    if op.startswith("ds_read") or op.startswith("ds_load"): return "ds_read"
    if op.startswith("ds_write") or op.startswith("ds_store"): return "ds_write"
    if op.startswith("buffer_load"): return "buf_load"
    if op.startswith("scratch_"): return "scratch"
    if op.startswith("s_waitcnt"): return "waitcnt"
    if op.startswith("s_barrier"): return "barrier"
    if op.startswith("s_"): return "salu"
    return op
tot = collections.Counter()
for name, ops in blocks:
    c = collections.Counter(cls(o) for o in ops)
    tot.update(c)
    if len(ops) >= minsz:
        v = sum(n for k, n in c.items() if k in ("pk_add", "pk_mul", "pk_fma", "mov", "fma", "mul", "v_other"))
        print(f"{name:12s} n={len(ops):5d} VALU={v:4d} " + " ".join(f"{k}={n}" for k, n in sorted(c.items())))
        if "v_other" in c:
            print("             v_other:", dict(collections.Counter(o for o in ops if cls(o) == "v_other").most_common(8)))
print("total", dict(tot))

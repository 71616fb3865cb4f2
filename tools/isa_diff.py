#!/usr/bin/env python3
"""isa_diff.py old.s new.s [--drop-cfg-arg N] — are two gfx950 device assemblies (hipcc --cuda-device-only -S) the same code?

For refactors of the kernel headers that must not change what the shipped kernels execute: every kernel function of `old` is
looked up in `new` by its demangled name and the two instruction streams are compared line by line, comments and metadata
stripped, local labels (.LBBn_m) renamed by order of first appearance so that a different function order does not matter.

  --drop-cfg-arg N   remove the N-th (0-based) argument of every `crn::Cfg<...>` in OLD names before matching (a template
                     parameter the refactor deleted)
  --only REGEX       compare only kernels whose demangled name matches
  --allow-missing    kernels of `old` that `new` does not have are listed, not an error (instantiations a refactor removed)

Exit 0: every compared kernel identical.  Prints a table otherwise and exits 1.
"""
import argparse
import re
import subprocess
import sys

CXXFILT = "c++filt"   # binutils (llvm-cxxfilt is not in the ROCm image)
LABEL = re.compile(r"\.L(BB|func_end|func_begin|tmp)?[0-9_]+")


def functions(path):
    """{mangled: [instruction lines]} for every @function symbol."""
    out, cur, name = {}, None, None
    for ln in open(path, errors="replace"):
        s = ln.rstrip("\n")
        m = re.match(r"\s*\.type\s+(\S+),@function", s)
        if m:
            name, cur = m.group(1), None
            continue
        if name and s.startswith(name + ":"):
            cur = []
            out[name] = cur
            continue
        if cur is not None:
            if s.startswith(".Lfunc_end"):
                cur, name = None, None
                continue
            t = s.split(";", 1)[0].rstrip()
            if not t.strip() or t.strip().startswith(".") and not t.strip().endswith(":"):
                continue   # blank, comment-only, directive (.p2align inside functions is scheduling-neutral padding the assembler redoes)
            cur.append(t.strip())
    return out


def canon(lines):
    ren = {}

    def sub(m):
        return ren.setdefault(m.group(0), f".L{len(ren)}")
    return [LABEL.sub(sub, ln) for ln in lines]


def demangle(names):
    r = subprocess.run([CXXFILT], input="\n".join(names), stdout=subprocess.PIPE, text=True, check=True)
    return dict(zip(names, r.stdout.split("\n")))


def drop_arg(name, idx):
    def fix(m):
        args = m.group(1).split(", ")
        del args[idx]
        return "crn::Cfg<" + ", ".join(args) + ">"
    return re.sub(r"crn::Cfg<([^<>]*)>", fix, name)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("old")
    ap.add_argument("new")
    ap.add_argument("--drop-cfg-arg", type=int, default=-1)
    ap.add_argument("--only", default="")
    ap.add_argument("--allow-missing", action="store_true")
    a = ap.parse_args()
    fo, fn = functions(a.old), functions(a.new)
    do, dn = demangle(list(fo)), demangle(list(fn))
    if a.drop_cfg_arg >= 0:
        do = {k: drop_arg(v, a.drop_cfg_arg) for k, v in do.items()}
    new_by_name = {v: k for k, v in dn.items()}
    same, differ, missing = 0, [], []
    for mo, name in sorted(do.items(), key=lambda kv: kv[1]):
        if a.only and not re.search(a.only, name):
            continue
        if name not in new_by_name:
            missing.append(name)
            continue
        x, y = canon(fo[mo]), canon(fn[new_by_name[name]])
        if x == y:
            same += 1
        else:
            first = next((i for i, (p, q) in enumerate(zip(x, y)) if p != q), min(len(x), len(y)))
            differ.append((name, len(x), len(y), first, x[first] if first < len(x) else "<end>", y[first] if first < len(y) else "<end>"))
    extra = sorted(set(dn.values()) - set(do.values()))
    print(f"isa_diff: {same} kernels identical, {len(differ)} differ, {len(missing)} only in old, {len(extra)} only in new")
    for name, lx, ly, first, p, q in differ:
        print(f"  DIFFERS {name}\n     {lx} vs {ly} instructions; first difference at {first}: `{p}` vs `{q}`")
    for name in missing:
        print(f"  only in old: {name}")
    for name in extra:
        print(f"  only in new: {name}")
    return 1 if differ or (missing and not a.allow_missing) else 0


if __name__ == "__main__":
    sys.exit(main())

#!/usr/bin/env python3
"""Remove the wait states hipcc puts between two inline-asm VOP3P instructions.

The butterflies of this library are single packed-fp32 instructions in inline asm (crn_butterflies.h: v_pk_add_f32 / v_pk_mul_f32 /
v_pk_fma_f32 with op_sel / neg modifiers).  For gfx940-family targets LLVM's hazard recogniser cannot see inside an inline-asm
statement: it assumes every one of them may carry the "dst_sel forwarding" hazard (a 16-bit / SDWA partial-register write followed by
a read of that register needs one wait state) and, because inline asm itself counts as zero wait states, it emits `s_nop 0` in front
of every asm statement that reads a register some earlier asm statement wrote with nothing but asm statements in between — 73 of them in
one 4096-point frame, each four issue cycles of its wave.  None of these instructions writes a partial register (VOP3P packed-f32
results are full 64-bit pairs; op_sel there selects SOURCE halves), and ordinary VALU read-after-write dependencies are interlocked
by the hardware, so the wait states protect nothing.

This filter drops `s_nop 0` exactly where it stands between the `;;#ASMEND` of one such statement and the `;;#ASMSTART` of the next
and both statements consist of v_pk_{add,mul,fma}_f32 only (or the first is one of crn_frame.h's LDS read blocks, which ends with its
own s_waitcnt lgkmcnt(0)).  Every other s_nop — the ones real hazards need (readlane after a VALU
write, DPP sources, trans results, ...) — has a compiler-generated instruction on at least one side and is left alone.

  strip_asm_nops.py in.s out.s     prints how many were dropped; exit 2 (and no output file) when the input does not look like what this
                                   filter was written for: no `;;#ASMSTART` / `;;#ASMEND` markers at all, or markers that do not pair up
"""
import re
import sys

PK = re.compile(r"^\s*v_pk_(add|mul|fma)_f32\s")
LDS_READ = re.compile(r"^\s*(ds_read_b(64|128)\s|s_waitcnt lgkmcnt\(0\)\s*$)")


def body_of(block):
    return [ln for ln in block if ln.strip() and not ln.strip().startswith(";")]


def pk_only(block):
    body = body_of(block)
    return bool(body) and all(PK.match(ln) for ln in body)


def lds_read_block(block):
    """crn_frame.h's read blocks: ds_read_b64 / _b128 from one base address, closed by their own s_waitcnt lgkmcnt(0) — the values
    a following packed-f32 statement reads have landed before the statement ends."""
    body = body_of(block)
    return len(body) > 1 and all(LDS_READ.match(ln) for ln in body) and body[-1].strip() == "s_waitcnt lgkmcnt(0)"


class NotWhatWasExpected(Exception):
    pass


def main(src, dst):
    lines = open(src).read().split("\n")
    starts = sum(1 for ln in lines if ln.strip() == ";;#ASMSTART")
    ends = sum(1 for ln in lines if ln.strip() == ";;#ASMEND")
    if starts == 0:
        raise NotWhatWasExpected("no ;;#ASMSTART markers in the assembly: the compiler's output format changed (or the unit has no inline asm)")
    if starts != ends:
        raise NotWhatWasExpected(f"{starts} ;;#ASMSTART but {ends} ;;#ASMEND markers")
    out, dropped, kept = [], 0, 0
    i, n = 0, len(lines)
    last_asm_pk = False          # the previous inline-asm statement was packed-f32 only and nothing but comments followed it
    while i < n:
        ln = lines[i]
        s = ln.strip()
        if s == ";;#ASMSTART":
            j = i + 1
            while lines[j].strip() != ";;#ASMEND":
                if lines[j].strip() == ";;#ASMSTART":
                    raise NotWhatWasExpected(f"nested ;;#ASMSTART at line {j + 1}")
                j += 1
            block = lines[i + 1:j]
            out.extend(lines[i:j + 1])
            last_asm_pk = pk_only(block) or lds_read_block(block)
            i = j + 1
            continue
        if s == "s_nop 0" and last_asm_pk:
            # look ahead: the next non-comment line must open another packed-f32-only asm statement
            j = i + 1
            while j < n and (not lines[j].strip() or lines[j].strip().startswith(";") and lines[j].strip() != ";;#ASMSTART"):
                j += 1
            if j < n and lines[j].strip() == ";;#ASMSTART":
                k = j + 1
                while lines[k].strip() != ";;#ASMEND":
                    k += 1
                if pk_only(lines[j + 1:k]):
                    dropped += 1
                    i += 1
                    continue
            kept += 1
        if s and not s.startswith(";"):
            last_asm_pk = False
        out.append(ln)
        i += 1
    open(dst, "w").write("\n".join(out))
    print(f"strip_asm_nops: dropped {dropped} s_nop 0 between packed-f32 inline-asm statements, kept {kept} other s_nop 0 next to one")


if __name__ == "__main__":
    try:
        main(sys.argv[1], sys.argv[2])
    except NotWhatWasExpected as e:
        print(f"strip_asm_nops: REFUSED: {e}")
        sys.exit(2)

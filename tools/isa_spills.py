#!/usr/bin/env python3
"""List, for every sense_kernel instance in the saved ISA (make -C csrc asm), where its scratch
(spill) instructions sit relative to the s_barrier / wave-barrier skeleton: a reload inside the frame
loop also waits on vmcnt, i.e. behind the prefetch.  Usage: isa_spills.py [substring-of-mangled-name]"""
import re, sys, os
S = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cognitive-radio-network_amd", "csrc", "build",
                 "crn_kernels-hip-amdgcn-amd-amdhsa-gfx950.s")
pat = sys.argv[1] if len(sys.argv) > 1 else ""
lines = open(S).read().split("\n")
starts = [i for i, l in enumerate(lines) if l.startswith("_ZN3crn12sense_kernel") and l.split(":")[0].endswith("E")]
for st in starts:
    name = lines[st].split(":")[0]
    if pat not in name:
        continue
    end = next(i for i in range(st, len(lines)) if lines[i].startswith(".Lfunc_end"))
    body = lines[st:end]
    sc = [i for i, l in enumerate(body) if "scratch_" in l]
    if not sc:
        continue
    m = re.search(r"CfgI(.*?)EEEEEv", name)
    print("==", m.group(1) if m else name, "lines", len(body))
    marks = [i for i, l in enumerate(body) if "s_barrier" in l or "; wave barrier" in l or "s_cbranch" in l and "LBB" in l]
    loads = [i for i in sc if "scratch_load" in body[i]]
    print("   stores at", [i for i in sc if "scratch_store" in body[i]][:12], " reloads at", loads[:24])
    bars = [i for i, l in enumerate(body) if "s_barrier" in l or "; wave barrier" in l]
    print("   barriers at", bars[:40])

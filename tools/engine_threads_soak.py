import os, sys, subprocess
sys.path[:0] = [os.path.join(os.getcwd(), "cognitive-radio-network_amd"), os.path.join(os.getcwd(), "tests")]
import numpy as np, crnsense as cs, signals
cfg = cs.cfg_reference(); L, per_seg = 364, 64
segs = []
for ch in range(4):
    iq, _ = signals.make_epochs(cfg, 7, seed=900 + ch, L=L, picks=[ch] * 7)
    segs.append(iq[: per_seg * L * 2])
np.concatenate(segs).tofile("/tmp/cap.bin")
for args in (["-v", "0"], ["-v", "0", "-g", "0"]):
    out = subprocess.run(["tests/harness/ecr_threads", "/tmp/cap.bin", str(L), str(per_seg), "20"] + args, capture_output=True, text=True, timeout=120)
    lines = out.stdout.splitlines()
    dec = [l.split() for l in lines if l.startswith("decision ")]
    pure = [(int(w[1]), int(w[3])) for w in dec if w[5] == "1"]
    wrong = [p for p in pure if p[0] != p[1]]
    print(args, "rc", out.returncode, "decisions", len(dec), "pure", len(pure), "wrong", len(wrong))
    print("\n".join(l for l in lines if l.startswith(("packets", "rx_wait", "execute_us"))))

#!/bin/bash
# Round 5: the ingest ring's host rate, this build against the round-4 library interleaved on one box, then the round's soaks and the GPU
# suite + smoke of the final commit.  The round-4 library is not in the tree: build it from its commit into ab/ first —
#   git worktree add /tmp/r04 ec570c8 && make -C /tmp/r04/cognitive-radio-network_amd/csrc -j4 objs &&
#   hipcc --offload-arch=gfx950 -fPIC -shared -pthread -o ab/libcrnsense_r04.so /tmp/r04/cognitive-radio-network_amd/csrc/obj/*.o -ldl
# (ring_rate does not check the ABI version and uses nothing that changed between 3 and 4).  -> profiles/r05_ring_rate_vs_r04_library.txt
O=gpurun_out/r05d; mkdir -p $O
for rep in 1 2 3; do
  for lib in new r04; do
    if [ $lib = r04 ]; then export LD_PRELOAD=$PWD/ab/libcrnsense_r04.so; else unset LD_PRELOAD; fi
    echo "== $lib rep $rep" >> $O/ring_rate_ab.txt
    timeout 60 tools/ring_rate 64 256 3 >> $O/ring_rate_ab.txt 2>&1
    timeout 60 tools/ring_rate 1 1 3 >> $O/ring_rate_ab.txt 2>&1
  done
done
unset LD_PRELOAD
grep -E "==|accepted|decisions" $O/ring_rate_ab.txt
timeout 900 python tests/soak_gpu.py 20000 > $O/soak.txt 2>&1; tail -2 $O/soak.txt
CRN_SENSE_LIB=$PWD/cognitive-radio-network_amd/libcrnsense_sc16.so timeout 900 python tests/soak_gpu.py 10000 20000 > $O/soak_sc16.txt 2>&1; tail -2 $O/soak_sc16.txt
CRN_EVIDENCE_DIR=$O timeout 2400 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $O/pytest_gpu.log; tail -6 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke exit $?" >> $O/smoke.log; tail -2 $O/smoke.log

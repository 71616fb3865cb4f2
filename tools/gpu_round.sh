#!/bin/bash
# Round evidence (tag $TAG, default r05), one gpurun call, one box: GPU tests, smoke, every bench line (single GPU; N = 8 rehearsed in
# both scaling modes over the stand-in RCCL), rocprofv3 kernel stats of the headline command, HBM traffic counters (separate --pmc
# passes per workload), SQ / GRBM counter groups, the engine's execute() latency, host-buffer (PCIe-inclusive) rate, probes.
#   gpurun --timeout 3600 -- 'bash tools/gpu_round.sh'   then   python tools/collect_profiles.py r05
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${TAG:-r05}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
[ -x tools/membw_policy ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o tools/membw_policy tools/membw_policy.hip
[ -x tools/ring_rate ] || make -C tests/harness ring_rate > /dev/null
CRN_EVIDENCE_DIR=$O timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke exit $?" >> $O/smoke.log
timeout 600 python bench.py > $O/bench_headline.json 2> $O/bench_headline.err
timeout 300 python bench.py --fft 1024 > $O/bench_cfg1_1024.json 2> $O/bench_cfg1.err
timeout 300 python bench.py --mode ref --cpu-epochs 0 > $O/bench_cfg3_ref512.json 2> $O/bench_cfg3.err
timeout 300 python bench.py --mode welch --cpu-epochs 0 > $O/bench_cfg2_welch.json 2> $O/bench_cfg2.err
timeout 300 python bench.py --mode welch --cpu-epochs 0 --frames 32 > $O/bench_cfg2_welch_K32.json 2> $O/bench_cfg2_K32.err
timeout 300 python bench.py --mode scan --cpu-epochs 0 --force-collective > $O/bench_cfg4_scan_1rank.json 2> $O/bench_cfg4.err
timeout 300 python bench.py --variant 2 --cpu-epochs 0 --no-alt > $O/bench_unpruned.json 2> $O/bench_unpruned.err
timeout 300 python bench.py --fft 512 --cpu-epochs 0 > $O/bench_e512.json 2> $O/bench_e512.err
timeout 300 python bench.py --fft 2048 --cpu-epochs 0 > $O/bench_e2048.json 2> $O/bench_e2048.err
# the optional wire-format library (make SC16=1): one line, so that what is no longer in the shipped library is still seen to work
[ -f cognitive-radio-network_amd/libcrnsense_sc16.so ] && timeout 300 python bench.py --wire-format --cpu-epochs 0 > $O/bench_wire_format_optional_library.json 2> $O/bench_wire_format.err
# the driver's command shape; the N = 1 end of the strong-scaling curve and its per-GPU shares at N = 2 / 4 / 8 as ONE-GPU launches
# (what one rank of the strong-scaled job runs: 1/N of the 8.75 GiB batch, two streams below 4 GiB) — the kernel-side scaling loss
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_shape.json 2> $O/bench_driver_shape.err
{ for n in 1 2 4 8; do timeout 300 python bench.py --steps 40 --warmup 10 --cpu-epochs 0 --no-alt --no-live-traffic --epochs $((28672 / n)) $([ $n -gt 2 ] && echo --two-streams) 2>/dev/null | tail -1; done; } > $O/bench_strong_share_per_gpu.jsonl
{ for n in 4 8; do timeout 300 python bench.py --steps 40 --warmup 10 --cpu-epochs 0 --no-alt --no-live-traffic --epochs $((28672 / n)) 2>/dev/null | tail -1; done; } > $O/bench_strong_share_per_gpu_one_stream.jsonl
# BASELINE.json configs[4] rehearsed the way the driver starts it (no launcher: bench.py starts its own ranks), eight ranks sharing this
# box's one GPU over the stand-in RCCL (real RCCL refuses two ranks per device), weak and strong scaling
export CRN_RCCL_LIB=$R/tests/harness/libfake_rccl_mp.so HIP_VISIBLE_DEVICES=0
# (the N > 1 lines carry cpu_baseline: rank 0 measures it after the timed region with the default protocol)
timeout 900 python3 bench.py --gpus 2 --steps 20 --warmup 5 --epochs 14336 > $O/bench_2ranks_one_gpu.json 2> $O/bench_2ranks_one_gpu.err
timeout 900 python3 bench.py --gpus 8 --steps 20 --warmup 5 --epochs 3584 > $O/bench_8ranks_one_gpu.json 2> $O/bench_8ranks_one_gpu.err
timeout 900 python3 bench.py --gpus 8 --steps 20 --warmup 5 --scaling strong > $O/bench_8ranks_one_gpu_strong.json 2> $O/bench_8ranks_one_gpu_strong.err
timeout 900 python3 bench.py --gpus 8 --steps 20 --warmup 5 --mode scan --epochs 8960 > $O/bench_8ranks_one_gpu_scan.json 2> $O/bench_8ranks_one_gpu_scan.err
unset CRN_RCCL_LIB HIP_VISIBLE_DEVICES
timeout 900 python tools/gpu_welch_spans.py > $O/welch_spans.txt 2>&1
timeout 900 python tests/soak_gpu.py 20000 > $O/soak.txt 2>&1
timeout 120 tools/ring_rate 64 256 3 > $O/ring_rate.txt 2>&1; timeout 60 tools/ring_rate 1 1 3 >> $O/ring_rate.txt 2>&1
./tools/membw_policy > $O/membw_policy.txt 2>&1
timeout 300 python tools/host_rate.py > $O/host_rate.txt 2>&1
timeout 300 python tools/engine_rate.py > $O/engine_rate.txt 2>&1
CRN_EVIDENCE_DIR=$O timeout 300 python tools/gpu_dealt_ab.py > /dev/null 2>&1
timeout 300 python tools/engine_idle_gap.py > $O/engine_idle_gap.txt 2>&1
timeout 300 bash tools/gpu_zeros_probe.sh > $O/zeros_probe.txt 2>&1
timeout 300 bash tools/gpu_power_probe.sh > $O/power_probe.txt 2>&1
timeout 300 python tools/cpu_scaling.py > $O/cpu_scaling.txt 2>&1
cd /tmp && export TMPDIR=/tmp
PYTHON=$(python3 -c 'import os, sys; print(os.path.realpath(sys.executable))')   # the real binary: no launcher hop behind rocprofv3's `--`
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $PYTHON $R/bench.py --cpu-epochs 0 --no-live-traffic --no-alt > $O/stats.log 2>&1
# HBM traffic per workload (FETCH_SIZE and WRITE_SIZE in separate passes)
for cfgname in "headline:" "cfg1:--fft 1024" "cfg3:--mode ref" "cfg2:--mode welch" "e512:--fft 512" "e2048:--fft 2048" "unpruned:--variant 2" "cfg2K32:--mode welch --frames 32"; do
  tag=${cfgname%%:*}; args=${cfgname#*:}
  timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_$tag -- $PYTHON $R/bench.py --steps 5 --warmup 20 --cpu-epochs 0 --no-live-traffic --no-alt $args > $O/pmc_fetch_$tag.log 2>&1
  timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write_$tag -- $PYTHON $R/bench.py --steps 5 --warmup 20 --cpu-epochs 0 --no-live-traffic --no-alt $args > $O/pmc_write_$tag.log 2>&1
done
cd $R
# SQ / LDS / VMEM / GRBM counter groups per workload (tools/gpu_pmc.sh -> gpurun_out/pmc_<name>)
for cfgname in "headline:" "welch:--mode welch" "e2048:--fft 2048" "ref:--mode ref" "e1024:--fft 1024" "unpruned:--variant 2"; do
  tag=${cfgname%%:*}; args=${cfgname#*:}
  TAG=${TAG}_$tag EXTRA="--no-alt $args" bash tools/gpu_pmc.sh > $O/pmc_sq_$tag.txt 2>&1
done
tail -3 $O/pytest_gpu.log; tail -3 $O/smoke.log; cat $O/bench_headline.json; cat $O/bench_strong_share_per_gpu.jsonl | cut -c1-400; cat $O/engine_rate.txt

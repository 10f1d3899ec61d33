#!/bin/bash
# One GPU-box session (run through gpurun).  usage: bash tools/gpu_round.sh <mode> [<mode> ...]
#   tests smoke                      parity suites (-m gpu) and __graft_entry__.smoke()
#   bench2 pmcstage                  bench.py on C2; per-stage PMC passes (FETCH/WRITE/MFMA) of the default C3 step
#   bench bench1 bench3 bench5       bench.py on the default workload (C3, with the C2 / C1 / C5 sub-results) / C1 / C3 / C5
#   bench1x                          C1 without extras; BOPT="--opt name=value" passes handle options
#   ab  tests_opts  opts             AB="name=value ..." interleaved in-process A/B (WL=C2|C3|C5, AB_STEPS=n); the stage parity
#                                    suite under TEST_OPTS="a=1,b=2"; one run per OPTS entry
#   ktests                           kernel tests selected by TEST_K (default: x6)
#   ablate clock                     x6 loader-tile ablation builds; sustained-clock probe
#   groups thresh splitk             A/B switches of bench.py (AR stream groups, tile thresholds, split-K through LN)
#   prof prof3 profstage pmc probe   rocprofv3 kernel stats (C2 / C3 / one stage), PMC passes, per-launch PMC probe
#   sweep sweep_ar sweep_voc sweep_vocx   GEMM engine sweeps (all / AR shapes / vocoder shapes / vocoder launch variants)
#   sweep_x6 sweep_x6s sweep_x6k sweep_skinny   x6 tile forms / under-filled AR launches / K-split x6 tiles / M <= 64 kernel
#   frontend s2                      rows f3 / f2 measurements
#   pmc1                             C1: per-kernel FETCH_SIZE and MFMA-busy PMC passes (tools/pmc_summary.py)
#   graph newtests                   hipGraph replay vs stream launches (boundary ubench); this round's new parity tests (TEST_K)
#   gridsync cpuinfo                 phase-boundary ubench (launch chain vs in-kernel grid barrier); host CPU limits of the box
#   strong                           bench.py --scaling strong on one rank (C4: 256 ragged utterances in one call)
#   vbench vsweep                    C3 bench / a GEMM sweep in measurement builds (tools/build_variant.sh) interleaved with the production library
#   phase3h ablate3h overheads3h pmcx3h   x3h kernels: phase timers (compute + loader wave), ablation builds, launch overheads, SQ counters
#   sweep_x3h sweep_x3hk sweep_x3hxc  x3h tile sweeps (loader tile / K-split + window tiles / cross-chunk forms)
#   power conc                       package power + shader clock while C3 runs (sysfs); kernel concurrency per stage of a traced step
# everything is written under gpurun_out/ (scratch); summaries worth keeping are copied to profiles/ by hand.
mkdir -p gpurun_out
export TMPDIR=/tmp
WHAT="${@:-tests smoke bench}"
python -m megatts2_amd.build > gpurun_out/build.log 2>&1
for w in $WHAT; do
case $w in
tests)
  timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --no-header -p no:cacheprovider --maxfail=40 -rf > gpurun_out/kernels.log 2>&1
  echo "kernels rc=$?"; tail -3 gpurun_out/kernels.log
  timeout 1500 python -m pytest tests/test_gpu_stages.py -m gpu -q --no-header -p no:cacheprovider --maxfail=40 -rf > gpurun_out/stages.log 2>&1
  echo "stages rc=$?"; tail -6 gpurun_out/stages.log ;;
prof1)
  # rocprofv3 kernel stats of the one-utterance path (C1 = infer.py's call); PROF_OPT="--opt name=value" for an A/B
  rm -rf gpurun_out/prof1${PROF_TAG}
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $GRAFT_REPO_ROOT/gpurun_out/prof1${PROF_TAG} -o bench -- python $GRAFT_REPO_ROOT/bench.py --workload C1 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-sub-workloads ${PROF_OPT}) > gpurun_out/prof1${PROF_TAG}.log 2>&1
  echo "prof1 rc=$?"; f=$(find gpurun_out/prof1${PROF_TAG} -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/prof1${PROF_TAG}_kernel_stats.csv && head -14 "$f" | cut -c1-200
  find gpurun_out/prof1${PROF_TAG} -name "*kernel_trace.csv" -size +8M -delete ;;
graph)
  # dependent-launch boundary replayed from a hipGraph vs launch by launch on a stream (tools/ubench/graph_chain.hip)
  timeout 120 variants/ubench/graph_chain > gpurun_out/ubench_graph.txt 2>&1
  echo "graph rc=$?"; cat gpurun_out/ubench_graph.txt ;;
ab2)
  # interleaved A/B of several option sets on one box, one after the other: AB2="ln_pairs=0 ln_pairs=2 x6_small_cfg=91"; WL2="C3 C2"
  for wl in ${WL2:-C3}; do for o in ${AB2}; do
    timeout 600 python bench.py --workload $wl --steps ${AB_STEPS:-6} --warmup 2 --no-cpu-baseline --no-roofline --ab $o 2>/dev/null | grep "^{" | tee -a gpurun_out/ab.txt
  done; done ;;
newtests)
  # this round's added parity tests only (TEST_K selects), before the whole suite is spent on them
  timeout 1500 python -m pytest tests/test_gpu_stages.py tests/test_gpu_kernels.py -m gpu -q --no-header -p no:cacheprovider --maxfail=10 -rf -k "${TEST_K:-exact_256 or in_flight or own_durations_end}" > gpurun_out/newtests.log 2>&1
  echo "newtests rc=$?"; tail -12 gpurun_out/newtests.log ;;
gridsync)
  # phase-boundary price list on THIS box: dependent launches vs grid barriers inside one launch (tools/ubench/grid_sync.hip)
  timeout 120 variants/ubench/grid_sync > gpurun_out/ubench_grid_sync.txt 2>&1
  echo "gridsync rc=$?"; cat gpurun_out/ubench_grid_sync.txt ;;
cpuinfo)
  python - <<'PY' | tee gpurun_out/cpuinfo.txt
import os
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us", "/sys/fs/cgroup/cpuset.cpus.effective"):
    try: print(f, open(f).read().strip())
    except Exception as e: print(f, "n/a")
PY
  ;;
strong)
  # BASELINE configs[3] on ONE rank: the 256 ragged utterances of C4 in one synthesize_batch call (the N = 1 anchor of a scaling run)
  timeout 900 python bench.py --gpus 1 --scaling strong --steps ${STRONG_STEPS:-3} --warmup 1 --no-cpu-baseline --no-sub-workloads > gpurun_out/bench_strong.log 2>&1
  echo "strong rc=$?"; tail -1 gpurun_out/bench_strong.log | cut -c1-1500 ;;
ab)
  # interleaved in-process A/B of handle options (robust against clock / temperature drift): AB="x6_mp256=1 x6_mp=3 ..."
  for o in ${AB}; do
    timeout 600 python bench.py --workload ${WL:-C3} --steps ${AB_STEPS:-6} --warmup 2 --no-cpu-baseline --no-roofline --ab $o 2>/dev/null | grep "^{" | tee -a gpurun_out/ab.txt
  done ;;
tests_opts)
  # the stage parity suite under candidate engine options: TEST_OPTS="x6_mp=1,x6_small_cfg=64"
  MT2_TEST_OPTS="$TEST_OPTS" timeout 1500 python -m pytest tests/test_gpu_stages.py -m gpu -q --no-header -p no:cacheprovider --maxfail=40 -rf -k "${TEST_K:-prod or tiny_end_to_end or ragged}" > gpurun_out/stages_opts.log 2>&1
  echo "stages under $TEST_OPTS rc=$?"; tail -6 gpurun_out/stages_opts.log ;;
ktests)
  timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --no-header -p no:cacheprovider --maxfail=40 -rf -k "${TEST_K:-x6}" > gpurun_out/kernels.log 2>&1
  echo "kernels rc=$?"; tail -4 gpurun_out/kernels.log ;;
smoke)
  timeout 600 python -c "import __graft_entry__ as g; g.build(); g.smoke()" > gpurun_out/smoke.log 2>&1
  echo "smoke rc=$?"; tail -2 gpurun_out/smoke.log ;;
bench)
  timeout 900 python bench.py --steps 5 --warmup 2 > gpurun_out/bench.log 2>&1
  echo "bench rc=$?"; tail -1 gpurun_out/bench.log | cut -c1-3500 ;;
bench2)
  timeout 900 python bench.py --workload C2 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_c2.log 2>&1
  echo "bench2 rc=$?"; tail -1 gpurun_out/bench_c2.log | cut -c1-1800 ;;
pmcstage)
  # the step is profiled in two halves (a full C3 step exceeds the profiler's dispatch limit with TCC counters)
  SPECS=""
  for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    t=$(echo $c | cut -d" " -f1)
    for half in a b; do
      if [ $half = a ]; then ARGS="--workload C2"; KEEP="mrte,adm"; else ARGS="--workload C3 --skip-adm"; KEEP="vqpe,regulate,plm,decoder,vocoder"; fi
      rm -rf gpurun_out/pmcs_${t}_$half
      (cd /tmp && timeout 600 rocprofv3 --pmc $c --kernel-trace -f csv -d $GRAFT_REPO_ROOT/gpurun_out/pmcs_${t}_$half -o pmc -- python $GRAFT_REPO_ROOT/bench.py $ARGS --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-sub-workloads --stage-markers) > gpurun_out/pmcs_${t}_$half.log 2>&1
      echo "pmcstage $t $half rc=$?"; tail -1 gpurun_out/pmcs_${t}_$half.log | cut -c1-160
      f=$(find gpurun_out/pmcs_${t}_$half -name "*counter_collection.csv" | head -1)
      [ -n "$f" ] && SPECS="$SPECS $f:$KEEP"
    done
  done
  python tools/pmc_stage_summary.py $SPECS gpurun_out/pmc_stage_C3.json | tail -70
  find gpurun_out/pmcs_* -name "*.csv" -size +6M -delete ;;
bench5)
  timeout 900 python bench.py --workload C5 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/bench_c5.log 2>&1
  echo "bench5 rc=$?"; tail -1 gpurun_out/bench_c5.log | cut -c1-1500 ;;
sweep_vocx)
  timeout 600 python tools/gemm_sweep.py vocx > gpurun_out/gemm_sweep_vocx.txt 2>&1
  echo "sweep_vocx rc=$?"; grep -v amdgpu.ids gpurun_out/gemm_sweep_vocx.txt ;;
sweep_voc)
  timeout 600 python tools/gemm_sweep.py voc > gpurun_out/gemm_sweep_voc.txt 2>&1
  echo "sweep_voc rc=$?"; grep -v amdgpu.ids gpurun_out/gemm_sweep_voc.txt ;;
s2)
  timeout 300 python tools/bench_s2.py > gpurun_out/bench_s2.log 2>&1
  echo "s2 rc=$?"; tail -1 gpurun_out/bench_s2.log | cut -c1-700 ;;
frontend)
  timeout 300 python tools/bench_frontend.py > gpurun_out/bench_frontend.log 2>&1
  echo "frontend rc=$?"; tail -1 gpurun_out/bench_frontend.log | cut -c1-700 ;;
profstage)
  for st in ${STAGES:-mrte}; do
    rm -rf gpurun_out/ps_$st
    (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $GRAFT_REPO_ROOT/gpurun_out/ps_$st -o st -- python $GRAFT_REPO_ROOT/tools/prof_stage.py $st 3) > gpurun_out/ps_$st.log 2>&1
    echo "profstage $st rc=$?"; grep "ms per call" gpurun_out/ps_$st.log
    f=$(find gpurun_out/ps_$st -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/ps_${st}_stats.csv && head -16 "$f" | cut -c1-170
    find gpurun_out/ps_$st -name "*kernel_trace.csv" -size +8M -delete
  done ;;
bench1)
  timeout 600 python bench.py --workload C1 --steps 5 --warmup 2 > gpurun_out/bench_c1.log 2>&1
  echo "bench1 rc=$?"; tail -1 gpurun_out/bench_c1.log | cut -c1-1200 ;;
conc)
  # kernel concurrency per stage of one C3 step (tools/trace_concurrency.py on a kernel trace with stage markers)
  rm -rf gpurun_out/conc
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace -f csv -d $GRAFT_REPO_ROOT/gpurun_out/conc -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-sub-workloads --stage-markers ${CONC_OPT:-}) > gpurun_out/conc.log 2>&1
  echo "conc rc=$?"; f=$(find gpurun_out/conc -name "*kernel_trace.csv" | head -1)
  [ -n "$f" ] && python tools/trace_concurrency.py "$f" | tee gpurun_out/trace_concurrency.txt
  find gpurun_out/conc -name "*.csv" -size +2M -delete ;;
prof3)
  rm -rf gpurun_out/prof3
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -f csv -d $GRAFT_REPO_ROOT/gpurun_out/prof3 -o bench -- python $GRAFT_REPO_ROOT/bench.py --workload C3 --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-sub-workloads) > gpurun_out/prof3.log 2>&1
  echo "prof3 rc=$?"; f=$(find gpurun_out/prof3 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/prof3_kernel_stats.csv && head -24 "$f" | cut -c1-200
  find gpurun_out/prof3 -name "*kernel_trace.csv" -size +8M -delete ;;
bench3)
  timeout 1200 python bench.py --workload C3 --steps 2 --warmup 1 > gpurun_out/bench_c3.log 2>&1
  echo "bench3 rc=$?"; tail -1 gpurun_out/bench_c3.log | cut -c1-1500 ;;
prof)
  rm -rf gpurun_out/prof
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -f csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-sub-workloads --stage-markers) > gpurun_out/prof.log 2>&1
  echo "prof rc=$?"; tail -2 gpurun_out/prof.log | cut -c1-400
  f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/prof_kernel_stats.csv && head -30 "$f"
  find gpurun_out/prof -name "*kernel_trace.csv" -size +8M -delete ;;
sweep)
  timeout 900 python tools/gemm_sweep.py > gpurun_out/gemm_sweep.txt 2>&1
  echo "sweep rc=$?"; cat gpurun_out/gemm_sweep.txt ;;
sweep_x6k)
  timeout 600 python tools/gemm_sweep.py x6k > gpurun_out/gemm_sweep_x6k.txt 2>&1
  echo "sweep_x6k rc=$?"; grep -v amdgpu.ids gpurun_out/gemm_sweep_x6k.txt ;;
sweep_skinny)
  timeout 600 python tools/gemm_sweep.py skinny > gpurun_out/gemm_sweep_skinny.txt 2>&1
  echo "sweep_skinny rc=$?"; grep -v amdgpu.ids gpurun_out/gemm_sweep_skinny.txt ;;
bench1x)
  timeout 600 python bench.py --workload C1 --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --no-sub-workloads ${BOPT} 2>/dev/null | grep "^{" | cut -c1-900 | tee -a gpurun_out/bench_c1x.txt ;;
sweep_x6s)
  timeout 600 python tools/gemm_sweep.py x6s > gpurun_out/gemm_sweep_x6s.txt 2>&1
  echo "sweep_x6s rc=$?"; grep -v amdgpu.ids gpurun_out/gemm_sweep_x6s.txt ;;
sweep_x6)
  timeout 600 python tools/gemm_sweep.py x6 > gpurun_out/gemm_sweep_x6.txt 2>&1
  echo "sweep_x6 rc=$?"; grep -v amdgpu.ids gpurun_out/gemm_sweep_x6.txt
  timeout 600 python tools/gemm_sweep.py x6win > gpurun_out/gemm_sweep_x6win.txt 2>&1
  echo "sweep_x6win rc=$?"; grep -v amdgpu.ids gpurun_out/gemm_sweep_x6win.txt ;;
sweep_x3h)
  timeout 900 python tools/gemm_sweep.py x3h > gpurun_out/gemm_sweep_x3h.txt 2>&1
  echo "sweep_x3h rc=$?"; grep -v amdgpu.ids gpurun_out/gemm_sweep_x3h.txt ;;
power)
  # package power and shader clock while C3 steps run back to back (and while the isolated 4096^3 GEMM runs)
  ls /sys/class/drm/card*/device/hwmon/hwmon*/ > gpurun_out/power_sysfs_ls.txt 2>&1
  timeout 600 python tools/power_sampler.py gpurun_out/power_c3_samples.txt -- python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline --no-sub-workloads ${POWER_OPT:-} > gpurun_out/power_c3.txt 2>&1
  echo "power rc=$?"; grep -v "^{" gpurun_out/power_c3.txt | tail -14; grep "^{" gpurun_out/power_c3.txt | cut -c1-330
  timeout 300 python tools/power_sampler.py gpurun_out/power_gemm_samples.txt -- python -c "
import sys; sys.path.insert(0, '.')
from megatts2_amd import runtime as rt
rt.device_check()
for i in range(6): print(rt.bench_gemm(4096, 4096, 4096, force_cfg=103, iters=400, w_copies=2, flags=8))
" > gpurun_out/power_gemm.txt 2>&1
  tail -14 gpurun_out/power_gemm.txt ;;
pmcx3h)
  # where the waves of the x3h loader tiles wait: SQ counters of a few launches (tools/x3h_pmc_probe.py), separate passes
  (cd /tmp && rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z0-9_]*" | sort -u | tr "\n" " ") > gpurun_out/pmcx3h_available.txt
  : > gpurun_out/pmcx3h.txt
  n=0
  for c in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU" \
           "SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_LDS" \
           "SQ_WAVE_CYCLES SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_FLAT"; do
    n=$((n + 1))
    rm -rf gpurun_out/pmcx3h_$n
    (cd /tmp && timeout 300 rocprofv3 --pmc $c --kernel-trace -f csv -d $GRAFT_REPO_ROOT/gpurun_out/pmcx3h_$n -o pmc -- python $GRAFT_REPO_ROOT/tools/x3h_pmc_probe.py) > gpurun_out/pmcx3h_$n.log 2>&1
    echo "pmcx3h pass $n rc=$?"; tail -2 gpurun_out/pmcx3h_$n.log | cut -c1-200
    f=$(find gpurun_out/pmcx3h_$n -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && python tools/x3h_pmc_probe.py summary "$f" >> gpurun_out/pmcx3h.txt
    find gpurun_out/pmcx3h_$n -name "*.csv" -size +8M -delete
  done
  cat gpurun_out/pmcx3h.txt ;;
sweep_x3hxc)
  timeout 900 python tools/gemm_sweep.py x3hxc > gpurun_out/gemm_sweep_x3hxc.txt 2>&1
  echo "sweep_x3hxc rc=$?"; grep -v amdgpu.ids gpurun_out/gemm_sweep_x3hxc.txt ;;
sweep_x3hk)
  timeout 900 python tools/gemm_sweep.py x3hk > gpurun_out/gemm_sweep_x3hk.txt 2>&1
  echo "sweep_x3hk rc=$?"; grep -v amdgpu.ids gpurun_out/gemm_sweep_x3hk.txt
  timeout 900 python tools/gemm_sweep.py x3hwin > gpurun_out/gemm_sweep_x3hwin.txt 2>&1
  echo "sweep_x3hwin rc=$?"; grep -v amdgpu.ids gpurun_out/gemm_sweep_x3hwin.txt ;;
ubench_f16)
  timeout 120 variants/ubench/mfma_f16_denorm > gpurun_out/ubench_mfma_f16.txt 2>&1
  echo "ubench_f16 rc=$?"; cat gpurun_out/ubench_mfma_f16.txt ;;
sweep_ar)
  timeout 600 python tools/gemm_sweep.py ar > gpurun_out/gemm_sweep_ar.txt 2>&1
  echo "sweep_ar rc=$?"; cat gpurun_out/gemm_sweep_ar.txt ;;
gaps)
  # plain kernel trace (no counters) of one C2 step + per-queue gap analysis of the ADM stage
  rm -rf gpurun_out/gaps
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace -f csv -d $GRAFT_REPO_ROOT/gpurun_out/gaps -o t -- python $GRAFT_REPO_ROOT/bench.py --workload ${WL:-C2} --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-sub-workloads --stage-markers) > gpurun_out/gaps.log 2>&1
  echo "gaps rc=$?"; f=$(find gpurun_out/gaps -name "*kernel_trace.csv" | head -1)
  [ -n "$f" ] && python tools/gap_analysis.py $f ${M0:-1} ${M1:-2} | tee gpurun_out/gap_analysis.txt
  find gpurun_out/gaps -name "*.csv" -size +6M -delete ;;
opts)
  # A/B of handle options on the default workload: OPTS="win_conv=0 ar_groups=1 ..." (one run per entry; "-" = defaults)
  for o in ${OPTS:--}; do
    if [ "$o" = "-" ]; then A=""; else A=$(echo $o | tr ',' '\n' | sed 's/^/--opt /' | tr '\n' ' '); fi
    timeout 600 python bench.py --workload ${WL:-C3} --steps 3 --warmup 1 --no-cpu-baseline $A > gpurun_out/bench_opt_$o.log 2>&1
    echo "opts $o rc=$?"; tail -1 gpurun_out/bench_opt_$o.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('stage_ms')); print('   ', [(r['config'], r['launches'], r['ms'], r['tflops']) for r in d['roofline']['per_config']])"
  done ;;
groups)
  for g in ${GROUPS_LIST:-1 2}; do
    timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --opt ar_groups=$g > gpurun_out/bench_g$g.log 2>&1
    echo "groups $g rc=$?"; tail -1 gpurun_out/bench_g$g.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('stage_ms'), d['roofline']['frac'], d['roofline']['gemm_ms_per_step'])"
  done ;;
probe)
  rm -rf gpurun_out/probe
  (cd /tmp && timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace -f csv -d $GRAFT_REPO_ROOT/gpurun_out/probe -o probe -- python $GRAFT_REPO_ROOT/tools/gemm_probe.py) > gpurun_out/probe.log 2>&1
  echo "probe rc=$?"; grep -v "^W2026\|amdgpu.ids" gpurun_out/probe.log | tail -14
  python tools/pmc_probe_summary.py gpurun_out/probe | tee gpurun_out/probe_summary.txt | cut -c1-400
  find gpurun_out/probe -name "*.csv" -size +4M -delete ;;
probe_x6)
  i=0
  for c in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_MFMA"; do
    i=$((i+1)); rm -rf gpurun_out/probe_x6_$i
    (cd /tmp && timeout 600 rocprofv3 --pmc $c --kernel-trace -f csv -d $GRAFT_REPO_ROOT/gpurun_out/probe_x6_$i -o probe -- python $GRAFT_REPO_ROOT/tools/gemm_probe.py x6) > gpurun_out/probe_x6_$i.log 2>&1
    echo "probe_x6 $i rc=$?"; grep -v "^W2026\|amdgpu.ids" gpurun_out/probe_x6_$i.log | tail -9
    python tools/pmc_probe_summary.py gpurun_out/probe_x6_$i | tee gpurun_out/probe_x6_summary_$i.txt | cut -c1-400
    find gpurun_out/probe_x6_$i -name "*.csv" -size +4M -delete
  done ;;
ablate)
  # measurement builds made in the build container (tools/build_variant.sh): which resource bounds a chunk of the x6 tile
  : > gpurun_out/x6_ablate.txt
  timeout 300 python tools/x6_ablate.py prod 2>&1 | grep -v amdgpu.ids >> gpurun_out/x6_ablate.txt
  for v in ${ABL_LIST:-abl1 abl2 abl3 abl4}; do
    [ -d variants/$v ] && cp tools/*.py variants/$v/tools/ && (cd variants/$v && timeout 300 python tools/x6_ablate.py $v 2>&1 | grep -v amdgpu.ids) >> gpurun_out/x6_ablate.txt
  done
  echo "ablate rc=$?"; cat gpurun_out/x6_ablate.txt ;;
overheads3h)
  timeout 300 python tools/x3h_overheads.py > gpurun_out/x3h_overheads.txt 2>&1
  echo "overheads3h rc=$?"; grep -v amdgpu.ids gpurun_out/x3h_overheads.txt ;;
vbench)
  # the C3 bench in measurement builds next to the production library, interleaved on this box: VBENCH_LIST variants, VBENCH_ROUNDS
  : > gpurun_out/vbench.txt
  for r in $(seq 1 ${VBENCH_ROUNDS:-2}); do
    timeout 600 python bench.py --steps ${VBENCH_STEPS:-8} --warmup 3 --no-cpu-baseline --no-roofline --no-sub-workloads ${VBENCH_OPT:-} 2>/dev/null | grep "^{" | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('prod', d['ms_per_step'], d.get('stage_ms'))" | tee -a gpurun_out/vbench.txt
    for v in ${VBENCH_LIST}; do
      cp bench.py variants/$v/; ln -sfn ../../tests variants/$v/tests; ln -sfn ../../profiles variants/$v/profiles
      (cd variants/$v && timeout 600 python bench.py --steps ${VBENCH_STEPS:-8} --warmup 3 --no-cpu-baseline --no-roofline --no-sub-workloads ${VBENCH_OPT:-} 2>/dev/null | grep "^{" | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d.get('stage_ms'))") | tee -a gpurun_out/vbench.txt
    done
  done ;;
vsweep)
  # a GEMM sweep mode inside measurement builds (VSWEEP_LIST variants, VSWEEP_MODE sweep mode) next to the production library
  for v in ${VSWEEP_LIST}; do
    cp tools/*.py variants/$v/tools/
    (cd variants/$v && timeout 600 python tools/gemm_sweep.py ${VSWEEP_MODE:-x3h}) > gpurun_out/vsweep_$v.txt 2>&1
    echo "vsweep $v rc=$?"; grep -v amdgpu.ids gpurun_out/vsweep_$v.txt
  done ;;
phase3h)
  # per-phase cycle sums of one compute wave of the x3h loader tile (variant h_phase: tools/build_variant.sh h_phase "-DMT2_PHASE_TIMING")
  cp tools/*.py variants/h_phase/tools/
  (cd variants/h_phase && timeout 300 python tools/x3h_phase_timing.py) > gpurun_out/x3h_phase_timing.txt 2>&1
  echo "phase3h rc=$?"; grep -v amdgpu.ids gpurun_out/x3h_phase_timing.txt ;;
ablate3h)
  # the same for the x3h tile (tools/x3h_ablate.py; variants h_abl1..h_abl4)
  : > gpurun_out/x3h_ablate.txt
  timeout 300 python tools/x3h_ablate.py prod 2>&1 | grep -v amdgpu.ids >> gpurun_out/x3h_ablate.txt
  for v in ${ABL_LIST:-h_abl1 h_abl2 h_abl3 h_abl4}; do
    [ -d variants/$v ] && cp tools/*.py variants/$v/tools/ && (cd variants/$v && timeout 300 python tools/x3h_ablate.py $v 2>&1 | grep -v amdgpu.ids) >> gpurun_out/x3h_ablate.txt
  done
  echo "ablate3h rc=$?"; cat gpurun_out/x3h_ablate.txt ;;
clock)
  # phase timer + clock probe (s_memtime vs s_memrealtime) in the MT2_PHASE_TIMING variant
  cp tools/*.py variants/phase/tools/
  (cd variants/phase && for m in ${CLOCK_MODES:-ldr f32}; do timeout 300 python tools/x6_phase_timing.py $m; done) > gpurun_out/clock_probe.txt 2>&1
  echo "clock rc=$?"; grep -v amdgpu.ids gpurun_out/clock_probe.txt ;;
phase)
  # instrumented build in a scratch copy of the package (the in-tree library stays the production one)
  rm -rf /tmp/mt2_phase && mkdir -p /tmp/mt2_phase && cp -r megatts2_amd include tools /tmp/mt2_phase/
  (cd /tmp/mt2_phase && rm -rf megatts2_amd/lib && MT2_EXTRA_HIPCC_FLAGS=-DMT2_PHASE_TIMING python -m megatts2_amd.build > $GRAFT_REPO_ROOT/gpurun_out/build_phase.log 2>&1 && MT2_EXTRA_HIPCC_FLAGS=-DMT2_PHASE_TIMING timeout 300 python tools/x6_phase_timing.py $PHASE_ARGS) > gpurun_out/x6_phase_timing.txt 2>&1
  echo "phase rc=$?"; grep -v amdgpu.ids gpurun_out/x6_phase_timing.txt ;;
splitk)
  for w in C2 C3; do for f in "" "--no-splitk"; do
    timeout 600 python bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline --no-roofline ${f:+--opt splitk=0} > gpurun_out/bench_sk.log 2>&1
    echo "splitk $w '$f' rc=$?"; tail -1 gpurun_out/bench_sk.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
  done; done ;;
thresh)
  for t in 0,0,0 512,1024,32; do
    IFS=, read a b c <<< "$t"
    timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --opt t_ks4=$a --opt t_ks2=$b --opt t32=$c > gpurun_out/bench_t_$t.log 2>&1
    echo "thresh $t rc=$?"; tail -1 gpurun_out/bench_t_$t.log | cut -c1-400
  done ;;
pmc1)
  # the one-utterance path (C1): HBM-side read bytes and matrix-pipe cycles per kernel (two separate --pmc passes, kernel trace only)
  for c in FETCH_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    t=$(echo $c | cut -d" " -f1)
    rm -rf gpurun_out/pmc1_$t
    (cd /tmp && timeout 600 rocprofv3 --pmc $c --kernel-trace -f csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc1_$t -o pmc -- python $GRAFT_REPO_ROOT/bench.py --workload C1 --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-sub-workloads) > gpurun_out/pmc1_$t.log 2>&1
    echo "pmc1 $t rc=$?"; tail -1 gpurun_out/pmc1_$t.log | cut -c1-200
    f=$(find gpurun_out/pmc1_$t -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && python tools/pmc_summary.py "$f" 2 gpurun_out/pmc1_$t.md | head -24 | cut -c1-220
    find gpurun_out/pmc1_$t -name "*.csv" -size +8M -delete
  done ;;
pmc)
  for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES"; do
    t=$(echo $c | cut -d" " -f1)
    rm -rf gpurun_out/pmc_$t
    (cd /tmp && timeout 900 rocprofv3 --pmc $c --kernel-trace -f csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$t -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline) > gpurun_out/pmc_$t.log 2>&1
    echo "pmc $t rc=$?"; tail -1 gpurun_out/pmc_$t.log | cut -c1-300
    f=$(find gpurun_out/pmc_$t -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && python tools/pmc_summary.py "$f" 2 gpurun_out/pmc_$t.md | tail -8
    find gpurun_out/pmc_$t -name "*.csv" -size +8M -delete
  done ;;
esac
done

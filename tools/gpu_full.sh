#!/bin/bash
# Full GPU visit: tests, smoke, bench (+cpu baseline), secondary measurements, rocprof stats + PMC (then a
# second bench run that picks the fresh PMC numbers up), stress, and -- when the instrumented builds of
# `python -m toppra_amd.build -DTPR_CERT_TIMING -DTPR_CERT_DEV --out=build_dbg/libtoppra_tim.so` and
# `... -DTPR_DEBUG_PREDICT -DTPR_CERT_DEV --out=build_dbg/libtoppra_dbg.so` are present -- the in-kernel
# cycle breakdown and the certificate hit rates.  Everything lands under gpurun_out/; copy what should be
# judged into profiles/ (tools/collect_profiles.sh).
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
export TOPPRA_EXPECT_REF=1   # the reference-solver tests FAIL (not skip) if oracle/_ref did not travel
timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -5 | tee gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee gpurun_out/smoke.log
timeout 600 python tools/gpu_measure.py 2>/dev/null > gpurun_out/measure.json; tail -5 gpurun_out/measure.json
rm -rf gpurun_out/prof_*
# kernel trace over the bench's default step counts (the averages then agree with bench.py's own HIP-event time; the
# first launches of a short run are 5-10 % slower), counters over a short run
bash tools/gpu_profile.sh "--no-cpu-baseline --no-secondary --no-configs" stats > gpurun_out/profile.log 2>&1
bash tools/gpu_profile.sh "--steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-configs --kernel-reps 2" pmc >> gpurun_out/profile.log 2>&1
tail -3 gpurun_out/profile.log
python tools/pmc_summary.py gpurun_out solve_kernel > gpurun_out/pmc_summary.txt
python tools/pmc_summary.py gpurun_out solve_kernel --json > gpurun_out/pmc.json
cp gpurun_out/pmc.json profiles/${ROUND:-r01}_pmc.json   # bench.py reads roofline.traffic from the latest profiles/r*_pmc.json
timeout 900 python bench.py 2>&1 | tail -1 | tee gpurun_out/bench.log
timeout 900 python tools/gpu_shortcut_stress.py ${STRESS_ROUNDS:-2} > gpurun_out/stress.log 2>&1; grep total gpurun_out/stress.log
timeout 600 python tools/gpu_near_parallel.py > gpurun_out/near_parallel.log 2>&1; tail -1 gpurun_out/near_parallel.log
timeout 600 python tools/gpu_tolerance_report.py 2>/dev/null > gpurun_out/tolerance_report.json; tail -5 gpurun_out/tolerance_report.json
# round 3: the new kernels against their full-iteration / two-scan counterparts, the latency kernel's parity and timings
timeout 600 python tools/gpu_feasible_check.py > gpurun_out/feasible_check.log 2>&1; tail -6 gpurun_out/feasible_check.log
timeout 600 python tools/gpu_sd_check.py > gpurun_out/sd_check.log 2>&1; tail -8 gpurun_out/sd_check.log
timeout 600 python tools/gpu_robust_check.py > gpurun_out/robust_check.log 2>&1; tail -6 gpurun_out/robust_check.log
timeout 900 python tools/gpu_cert_dofs_check.py > gpurun_out/cert_dofs.log 2>&1; tail -9 gpurun_out/cert_dofs.log
timeout 900 python tools/gpu_r3_stress.py ${STRESS_ROUNDS:-2} > gpurun_out/r3_stress.log 2>&1; tail -20 gpurun_out/r3_stress.log
timeout 600 python tools/gpu_wave_check.py > gpurun_out/wave_check.log 2>&1; tail -12 gpurun_out/wave_check.log
timeout 900 python tools/gpu_pair_check.py > gpurun_out/pair_check.log 2>&1; tail -16 gpurun_out/pair_check.log
timeout 600 python tools/gpu_sliver_hunt.py > gpurun_out/sliver_hunt.log 2>&1; tail -3 gpurun_out/sliver_hunt.log
timeout 300 python tools/gpu_mode_times.py > gpurun_out/mode_times.log 2>&1; cat gpurun_out/mode_times.log
timeout 300 python tools/gpu_param_pcr_check.py > gpurun_out/param_pcr_check.log 2>&1; tail -4 gpurun_out/param_pcr_check.log
timeout 300 python tools/gpu_dense_check.py > gpurun_out/dense_check.log 2>&1; tail -5 gpurun_out/dense_check.log
bash tools/gpu_profile_secondary.sh all > gpurun_out/sec_profile.log 2>&1; tail -3 gpurun_out/sec_profile.log
if [ -f build_dbg/libtoppra_wtim.so ]; then
  (TOPPRA_HIP_LIB=build_dbg/libtoppra_wtim.so timeout 120 python tools/gpu_wave_phases.py timing 4096 7 200; TOPPRA_HIP_LIB=build_dbg/libtoppra_wtim.so timeout 120 python tools/gpu_wave_phases.py timing 1 7 100) > gpurun_out/wave_phases.log 2>&1; tail -4 gpurun_out/wave_phases.log
fi
if [ -f build_dbg/libtoppra_tim.so ]; then
  TOPPRA_HIP_LIB=build_dbg/libtoppra_tim.so timeout 300 python tools/gpu_cert_phases.py > gpurun_out/phases.log 2>&1; tail -3 gpurun_out/phases.log
fi
if [ -f build_dbg/libtoppra_dbg.so ]; then
  TOPPRA_HIP_LIB=build_dbg/libtoppra_dbg.so TPR_DEV_BUILD=1 timeout 300 python tools/gpu_shortcut_hitrate.py > gpurun_out/hitrate.log 2>&1; tail -3 gpurun_out/hitrate.log
  TOPPRA_HIP_LIB=$PWD/build_dbg/libtoppra_dbg.so timeout 300 python tools/gpu_walk_fail.py > gpurun_out/walk_fail.log 2>&1; head -8 gpurun_out/walk_fail.log
fi

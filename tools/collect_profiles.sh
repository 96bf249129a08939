#!/bin/bash
# Copy the judged summaries of the last tools/gpu_full.sh visit from gpurun_out/ into profiles/ (round tag $1).
set -eu
R=${1:-r01}
cp gpurun_out/bench.log profiles/${R}_bench_n1.json
cp gpurun_out/pmc.json profiles/${R}_pmc.json
cp gpurun_out/pmc_summary.txt profiles/${R}_rocprofv3_pmc_summary.txt
cp gpurun_out/prof_stats/run_kernel_stats.csv profiles/${R}_rocprofv3_kernel_stats.csv
cp gpurun_out/measure.json profiles/${R}_secondary_measurements.json
cp gpurun_out/stress.log profiles/${R}_shortcut_vs_full_seidel_stress.log
[ -f gpurun_out/near_parallel.log ] && cp gpurun_out/near_parallel.log profiles/${R}_near_parallel_rows_stress.log
[ -f gpurun_out/tolerance_report.json ] && cp gpurun_out/tolerance_report.json profiles/${R}_tolerance_report.json
[ -f gpurun_out/walk_fail.log ] && cp gpurun_out/walk_fail.log profiles/${R}_walk_give_ups_and_regime_transitions.log
[ -f gpurun_out/hitrate.log ] && cp gpurun_out/hitrate.log profiles/${R}_shortcut_hit_rate.log
[ -f gpurun_out/phases.log ] && cp gpurun_out/phases.log profiles/${R}_family3_cycle_breakdown.log
for pair in feasible_check:feasible_sets_family3_check sd_check:toppra_sd_fused_check robust_check:robust_rows_across_lanes_check cert_dofs:family3_above_8_dof \
            r3_stress:stress_round3_kernels wave_check:family4_parity_and_timings pair_check:family5_parity_and_timings sliver_hunt:sliver_hunt mode_times:certificate_mode_times wave_phases:family4_cycle_breakdown param_pcr_check:param_spline_knot_parallel_check dense_check:dense_rows_check; do
  src=${pair%%:*}; dst=${pair##*:}
  [ -f gpurun_out/$src.log ] && grep -v amdgpu.ids gpurun_out/$src.log > profiles/${R}_$dst.log
done
[ -f gpurun_out/sec_kernel_stats.csv ] && grep -v "at::native\|rocclr" gpurun_out/sec_kernel_stats.csv > profiles/${R}_secondary_kernel_stats.csv
[ -f gpurun_out/sec_pmc.txt ] && grep "tpr::" gpurun_out/sec_pmc.txt > profiles/${R}_secondary_pmc.txt
ls -la profiles | tail -24

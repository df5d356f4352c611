# copy the summaries of gpurun_out/r06 (tools/profile_r06.sh) into profiles/ under the round's names
src=gpurun_out/r06; dst=profiles
cp $src/bench_n1.json $dst/r06_bench_n1.json
cp $src/kernel_stats.csv $dst/r06_bench_kernel_stats.csv
for f in pmc_hbm_bytes.csv pmc_mfma_util.csv pmc_sq_waves.csv step_timeline.txt config5_50k.json config5_kernel_stats.csv config5_levels.txt config5_pmc_hbm_bytes.csv config5_step_timeline.txt config6_1gpu.json h2h_breakdown.txt sharded_1rank.txt sp_config4_kernel_stats.csv sp_config4_pmc_hbm_bytes.csv wloa.json wloa_kernel_stats.csv wloa_pmc_hbm_bytes.csv; do
  [ -f $src/$f ] && cp $src/$f $dst/r06_$f
done
cp $src/sp_config4_bench.json $dst/r06_sp_config4_bench.json
for w in nci1 dd reddit collab; do
  cp $src/bench_$w.json $dst/r06_bench_${w}_like.json
  for m in wl sp wl_e2e; do [ -f $src/pub_${w}_$m.json ] && cp $src/pub_${w}_$m.json $dst/r06_pub_${w}_$m.json; done
  for m in wl sp; do [ -f $src/pub_${w}_${m}_kernel_stats.csv ] && python tools/kstats.py $src/pub_${w}_${m}_kernel_stats.csv 40 > $dst/r06_pub_${w}_${m}_kernel_stats.txt; done
  [ $w != nci1 ] && cp $src/pub_${w}_sp.json $dst/r06_pub_${w}_sp_asserted.json
done

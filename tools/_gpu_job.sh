mkdir -p gpurun_out/r02s
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r02s/smoke.log 2>&1; tail -4 gpurun_out/r02s/smoke.log
timeout 300 python bench.py --no-cpu-baseline --windows 1 2>/dev/null | cut -c1-200
sh tools/pmc_bench.sh r02 > gpurun_out/r02s/pmc.log 2>&1; tail -3 gpurun_out/r02s/pmc.log | cut -c1-400
for c in 1 0 2 3 tracking 4; do
  sh tools/profile_bench.sh r02c_c$c --config $c > gpurun_out/r02s/c$c.log 2>&1; tail -1 gpurun_out/r02s/c$c.log | cut -c1-230
done

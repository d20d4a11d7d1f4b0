mkdir -p gpurun_out/r02q
rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk" | head -4
timeout 600 python -m pytest tests/test_hip_mapping.py tests/test_hip_parity.py::test_replica_tracking_config tests/test_hip_ipc.py -x -q > gpurun_out/r02q/pytest.log 2>&1; tail -12 gpurun_out/r02q/pytest.log
sh tools/profile_bench.sh r02b_ctracking --config tracking > gpurun_out/r02q/ctracking.log 2>&1; head -12 gpurun_out/r02b_ctracking/r02b_ctracking_kernel_stats_top.txt | cut -c1-110; tail -1 gpurun_out/r02b_ctracking/r02b_ctracking_kernel_stats_top.txt; cut -c1-250 gpurun_out/r02b_ctracking/r02b_ctracking_bench.json; tail -3 gpurun_out/r02b_ctracking/bench.err
timeout 300 python bench.py --config tracking --unfused --no-cpu-baseline --windows 1 2>/dev/null | cut -c1-200
timeout 300 python tools/slam_synthetic.py --frames 40 > gpurun_out/r02q/slam40.json 2> gpurun_out/r02q/slam40.err; cut -c1-900 gpurun_out/r02q/slam40.json; tail -4 gpurun_out/r02q/slam40.err
timeout 300 python bench.py --no-cpu-baseline --windows 1 2>/dev/null | cut -c1-220
rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk" | head -4

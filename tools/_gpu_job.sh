mkdir -p gpurun_out/r02r
timeout 900 python -m pytest tests -q -m gpu -x --deselect tests/test_hip_parity.py::test_synthetic_stress_full_size_vs_oracle --durations=5 > gpurun_out/r02r/pytest.log 2>&1; tail -14 gpurun_out/r02r/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r02r/smoke.log 2>&1; tail -3 gpurun_out/r02r/smoke.log
sh tools/profile_bench.sh r02b_ctracking --config tracking > gpurun_out/r02r/ctracking.log 2>&1; head -8 gpurun_out/r02b_ctracking/r02b_ctracking_kernel_stats_top.txt | cut -c1-110; tail -1 gpurun_out/r02b_ctracking/r02b_ctracking_kernel_stats_top.txt; cut -c1-250 gpurun_out/r02b_ctracking/r02b_ctracking_bench.json
timeout 300 python tools/slam_synthetic.py --frames 100 > gpurun_out/r02r/slam100.json 2> gpurun_out/r02r/slam100.err; cut -c1-900 gpurun_out/r02r/slam100.json; tail -2 gpurun_out/r02r/slam100.err

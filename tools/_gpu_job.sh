for i in 1 2; do
NSR_FWD_SMALL=0 timeout 300 python bench.py --config tracking --no-cpu-baseline --windows 3 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('trk full ', round(r['value']), round(r['ms_per_step'],4))"
timeout 300 python bench.py --config tracking --no-cpu-baseline --windows 3 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('trk small', round(r['value']), round(r['ms_per_step'],4))"
done
NSR_FWD_SMALL=0 timeout 300 python bench.py --config 0 --no-cpu-baseline --windows 3 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('c0 full ', round(r['value']), round(r['ms_per_step'],4))"
timeout 300 python bench.py --config 0 --no-cpu-baseline --windows 3 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('c0 small', round(r['value']), round(r['ms_per_step'],4))"
timeout 300 python bench.py --no-cpu-baseline --windows 3 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('c1', round(r['value']), round(r['ms_per_step'],4), r['kernel_ms'])"
mkdir -p gpurun_out/r02v
timeout 600 python -m pytest tests/test_hip_parity.py tests/test_hip_mapping.py tests/test_hip_callers.py -q -x --deselect tests/test_hip_parity.py::test_synthetic_stress_full_size_vs_oracle > gpurun_out/r02v/pytest.log 2>&1; tail -3 gpurun_out/r02v/pytest.log

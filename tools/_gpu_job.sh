for i in 1 2; do
for v in pf8 pfh cb8 cb11; do
NSR_LIB_PATH=$PWD/nice_slam_amd/_ab/libnsr_$v.so timeout 300 python bench.py --no-cpu-baseline --windows 3 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$v', round(r['value']), round(r['ms_per_step'],4), r['kernel_ms'])"
done
done
for v in pf8 pfh cb11; do
NSR_LIB_PATH=$PWD/nice_slam_amd/_ab/libnsr_$v.so timeout 300 python bench.py --config 2 --no-cpu-baseline --windows 1 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$v c2', round(r['value']), round(r['ms_per_step'],4), r['kernel_ms'])"
NSR_LIB_PATH=$PWD/nice_slam_amd/_ab/libnsr_$v.so timeout 300 python bench.py --config tracking --no-cpu-baseline --windows 1 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$v trk', round(r['value']), round(r['ms_per_step'],4), r['kernel_ms'])"
done

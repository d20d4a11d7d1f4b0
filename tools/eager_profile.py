"""Host-side cost of the eager (no hipGraph) mapping iteration: cProfile of bench.py --eager on the GPU box.
    python tools/eager_profile.py > gpurun_out/eager_profile.txt"""
import cProfile, pstats, sys, os, io
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = ["bench.py", "--eager", "--no-cpu-baseline", "--steps", "600", "--warmup", "30"]
import bench
pr = cProfile.Profile()
pr.enable()
bench.main()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print(s.getvalue())

# usage: bash tools/slam_sweep.sh   (on the GPU box) -- tools/slam_synthetic.py: defaults, the reference's mapping schedule,
# and the motion model alone (no tracking iterations) on the same 100-frame sequence
run() {
  echo "== slam_synthetic.py $*"
  timeout 400 python tools/slam_synthetic.py "$@" 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read())
print('ATE rmse %.2f cm | final / mean raw translation error %.1f / %.1f cm | path %.2f m | %d tracking + %d mapping iterations | tracking %.1fs mapping %.1fs wall %.1fs'
      % (r['value'], r['raw_translation_error_cm']['final'], r['raw_translation_error_cm']['mean'], r['path_length_m'], r['tracking_iters'], r['mapping_iters'], r['tracking_s'], r['mapping_s'], r['wall_s']))"
}
run --frames 100
run --frames 100 --track-iters 0
run --frames 100 --map-iters 60 --every-frame 5

for st in 0 256 400 512 800 1200; do
  OISAT_SMALL_TILES=$st timeout -k 10 200 python bench.py --workload config2_360x720_1e4obs --no-cpu-baseline --no-secondary --no-roofline --steps 20 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('small<= $st', 'config2', round(d['ms_per_step'],3), d['refinement_residuals'][-1])"
  OISAT_SMALL_TILES=$st timeout -k 10 200 python scratch/tiled_once.py 12 2>/dev/null | tail -1
done

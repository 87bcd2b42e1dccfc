for prio in 1 0; do
for mode in recursive lookahead:1 lookahead:2 lookahead:3 lookahead:4 lookahead:6; do
  OISAT_AUX_PRIORITY=$prio OISAT_POTRF=$mode timeout -k 10 200 python bench.py --workload config2_360x720_1e4obs --no-cpu-baseline --no-secondary --no-roofline --steps 20 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('aux prio $prio', '$mode', round(d['ms_per_step'],3), d['refinement_residuals'][-1])"
done
done

for free in 0 8 16 32; do
for mode in recursive lookahead:2 lookahead:4; do
  OISAT_AUX_FREE_CUS=$free OISAT_POTRF=$mode timeout -k 10 200 python bench.py --workload config2_360x720_1e4obs --no-cpu-baseline --no-secondary --no-roofline --steps 20 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('free CUs $free', '$mode', round(d['ms_per_step'],3), d['refinement_residuals'][-1])"
done
done

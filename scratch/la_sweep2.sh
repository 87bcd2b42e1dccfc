for m in 3000 6000 10000 20000 40000; do
for mode in recursive lookahead:4; do
  OISAT_AUX_FREE_CUS=64 OISAT_POTRF=$mode timeout -k 10 100 python scratch/la_own.py $m 2>/dev/null | tail -1
done
done

for m in 6000 10000 14000; do
for cfg in "recursive 0" "lookahead:4 64" "lookahead:3 64" "lookahead:4 32"; do
  set -- $cfg
  OISAT_AUX_FREE_CUS=$2 OISAT_POTRF=$1 timeout -k 10 100 python scratch/la_own.py $m 2>/dev/null | tail -1
done
done

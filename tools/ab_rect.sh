for rep in 1 2; do
for r in default 8,1 32,1 16,2; do
  if [ "$r" == "default" ]; then unset MFB_SEG_FSM_RECT; else export MFB_SEG_FSM_RECT=$r; fi
  echo -n "rect $r rep $rep: "
  timeout -k 10 200 python tools/seg_probe.py 20 256 bench_GMSK 8 32 --no-twopass 2>&1 | grep "^segment" | cut -c1-120
done; done

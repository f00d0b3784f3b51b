for f in 0 1; do for r in "8,1" "4,1" "2,1" "1,1"; do
echo -n "FSM=$f rect=$r: "; MFB_SEG_FSM=$f MFB_SEG_FSM_RECT=$r python tools/seg_probe.py 17 64 CC11xx 11 32 --no-twopass 2>&1 | grep "^segment"
done; done
for f in 0 1; do echo -n "FSM=$f default rect: "; MFB_SEG_FSM=$f python tools/seg_probe.py 17 64 CC11xx 11 32 --no-twopass 2>&1 | grep "^segment"; done
for f in 0 1; do echo -n "FSM=$f dev batch 8: "; MFB_SEG_FSM=$f python3 tools/batch_device_rate.py 17 64 8 100 1 0 2>&1 | tail -1; done

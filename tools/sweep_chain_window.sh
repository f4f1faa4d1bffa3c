#!/bin/bash
# Same-box A/B of the tick kernel's chained-playout window (bench.py --chain-window-us): games/s, batch fill, tick-kernel time.
for w in ${@:-10 6 8 12 14 10}; do
  python bench.py --steps 2 --warmup 1 --cpu-baseline off --ref-seconds 0 --chain-window-us $w 2>/dev/null > /tmp/_cw.json
  python - "$w" <<'PY'
import json, sys
d = json.loads(open("/tmp/_cw.json").read().strip().splitlines()[-1])
print("window %s us: %.1f games/s  fill %.4f  tick kernel %.1f us  tower+head %.1f us" % (sys.argv[1], d["value"], d["roofline"]["batch_fill"],
      1e3 * d["roofline_tree"]["ms_per_launch"], 1e3 * d["roofline"]["ms_per_launch"]))
PY
done

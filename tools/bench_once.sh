#!/bin/bash
# one headline run on whatever box the call lands on: box id + the line's headline numbers
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocm-smi --showuniqueid 2>/dev/null | grep -i "unique id:" | head -1
python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null > /tmp/b.json
python - <<'PY'
import json
d = json.loads(open("/tmp/b.json").read().strip().splitlines()[-1])
print("total %.1f median %.1f seq %.1f legs %s" % (d["value"], d["windows_frames_per_s"]["median"], d["sequential"]["frames_per_s"],
      d["breakdown"]["ms_per_frame_by_leg"]))
PY

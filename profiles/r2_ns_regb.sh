#!/bin/bash
# A/B of the point engine's register-held B operand (navierstokes 3-D, 89 dofs): MHA_ENGINE_STOP=8 switches it off.
set -e
for stop in 8 0 8 0; do
  echo "== MHA_ENGINE_STOP=$stop (8 = B operand read from the LDS for every product)"
  MHA_ENGINE_STOP=$stop python bench.py --config 4 --ncell 32 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('config 4 at 32^3: ms_per_step', d['ms_per_step'])"
done
echo "== no column tiles (5) / products only (6) / stores only (7), register B"
for stop in 5 6 7; do
  MHA_ENGINE_STOP=$stop python bench.py --config 4 --ncell 32 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('stop $stop: ms_per_step', d['ms_per_step'])"
done
echo "== config 4 at 64^3"
python bench.py --config 4 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('config 4 at 64^3: ms_per_step', d['ms_per_step'])"

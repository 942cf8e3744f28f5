#!/bin/bash
# round 3: block-pattern Jacobian kernel, LDS image roles (aligned whole-line stores) against direct register stores
python profiles/r3_dbg_k2.py 2>&1 | grep -v amdgpu.ids | head -12
for img in 0 1; do for d in 0 4 2; do
  echo "== image roles=$img MHA_BP_DBG=$d"
  if [ $img = 0 ]; then unset MHA_BP_IMAGE; else export MHA_BP_IMAGE=1; fi
  MHA_VERBOSE=1 MHA_BP_DBG=$d MHA_K1K2_OVERLAP=0 timeout -k 10 200 python bench.py --no-cpu-baseline --steps 20 2> gpurun_out/r3/img_$img_$d.err | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('ms_per_step %.4f kernel_ms %.4f frac %.3f' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac']))"
  grep "block patterns" gpurun_out/r3/img_$img_$d.err | head -2
done; done

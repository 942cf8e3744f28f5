#!/bin/bash
# round 3: cache-policy bits of the direct row stores of the block-pattern kernel (variants of the library built with
# -DMHA_BP_STORE_AUX=N into build/lib_auxN.so: 1 sc0, 2 nt, 16 sc1, 17 sc0 sc1), one box
cp mrhyde_amd/lib/libmrhyde_amd.so /tmp/lib_default.so
for v in default aux1 aux2 aux16 aux17 default; do
  if [ $v = default ]; then cp /tmp/lib_default.so mrhyde_amd/lib/libmrhyde_amd.so; else cp build/lib_$v.so mrhyde_amd/lib/libmrhyde_amd.so; fi
  echo "== $v"
  MHA_K1K2_OVERLAP=0 timeout -k 10 200 python bench.py --no-cpu-baseline --steps 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('ms_per_step %.4f kernel_ms %.4f frac %.3f' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac']))"
done
cp /tmp/lib_default.so mrhyde_amd/lib/libmrhyde_amd.so

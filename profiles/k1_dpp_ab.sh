#!/bin/bash
# A/B of K1's residual contraction: LDS broadcasts (-DMHA_K1_DPP=0) against the transposed form with DPP reductions (=1),
# one box, interleaved; parity tests of the row-owner path run on the DPP build first.  Leaves the default build behind.
cd $GRAFT_REPO_ROOT
build() {
  rm -f build/obj/k_thermal_row_owner.o
  make -s -C mrhyde_amd/csrc HIPFLAGS="--offload-arch=gfx950 -munsafe-fp-atomics -ffp-contract=fast -DMHA_K1_DPP=$1" > /dev/null 2>&1 || { echo "build failed $1"; exit 1; }
}
build 1
timeout -k 10 600 python -m pytest tests/test_thermal_gpu.py tests/test_full_size_gpu.py -m gpu -x -q 2>&1 | tail -3
for r in 1 2 3; do
  for w in 0 1; do
    build $w
    for ov in 1 0; do
      MHA_K1K2_OVERLAP=$ov timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > /tmp/abe.log 2>&1
      echo "round $r DPP=$w overlap=$ov $(tail -1 /tmp/abe.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f ms/step  kernel_ms %.4f' % (d['ms_per_step'], d['roofline']['kernel_ms']))")"
    done
  done
done
rm -f build/obj/k_thermal_row_owner.o
make -s -C mrhyde_amd/csrc > /dev/null 2>&1

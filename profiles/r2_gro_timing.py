"""Per-phase cycle sums of thermal_general_row_owner_kernel (env MHA_GRO_TIMING=<file>): mean over workgroups, per wave.
Phases: 0 top..G1 barrier (stage A / fields), 1 G2 + barrier, 2 stage B (waves 0-3), 3 tiles, 4 wait at the barrier
after T, 5 stores + last barrier.  Counter: s_memtime (100 MHz on gfx950: 10 ns ticks)."""
import sys
import numpy as np
t = np.fromfile(sys.argv[1], dtype=np.int64).reshape(-1, 8, 8)[:, :, :6]
print("workgroups", t.shape[0], " ticks (10 ns) per workgroup, mean over workgroups; rows = waves")
m = t.mean(axis=0)
for w in range(8):
    print(w, " ".join("%8.0f" % v for v in m[w]), " total %8.0f" % m[w].sum())
print("max over waves of the workgroup total: mean %.0f max %.0f ticks" % (t.sum(axis=2).max(axis=1).mean(), t.sum(axis=2).max()))

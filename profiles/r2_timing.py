"""Wall-clock stamps of the block-pattern kernel's wavefronts (MHA_BP_TIMING): slots 0 start, 1 image loaded,
2.. segments done, 7 end.  Which workgroups take how long?"""
import sys, numpy as np
raw = np.fromfile(sys.argv[1], dtype=np.int64).reshape(-1, 12, 8).astype(np.float64)
t0 = raw[:, :, 0].min()
t = (raw - t0) * 0.01  # us (100 MHz)
t[raw == 0] = np.nan
end = np.nanmax(t, axis=(1, 2))
print("kernel span %.1f us; workgroup end: min %.1f median %.1f max %.1f" % (np.nanmax(end), np.nanmin(end), np.nanmedian(end), np.nanmax(end)))
print("W loaded after %.1f us (median)" % np.nanmedian(t[:, :, 1] - t[:, :, 0]))
h, edges = np.histogram(end, bins=12)
for n, a, b in zip(h, edges[:-1], edges[1:]):
    print("  %6.1f - %6.1f us: %3d workgroups" % (a, b, n))
slow = np.argsort(end)[-8:]
print("slowest workgroups:", ", ".join("%d (%.0f us)" % (w, end[w]) for w in slow))

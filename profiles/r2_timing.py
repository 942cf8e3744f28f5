"""Wall-clock stamps of the block-pattern kernel's wavefronts (MHA_BP_TIMING): where does a workgroup's time go?
slots: 0 start, 1 image loaded, 2 first part done, 7 last part done; 3..6 = iteration 2 of the first part:
top, loads issued, products + stores issued, counted wait passed."""
import sys, numpy as np
raw = np.fromfile(sys.argv[1], dtype=np.int64).reshape(-1, 12, 8).astype(np.float64)
t0 = raw[:, :, 0].min()
t = (raw - t0) * 0.01  # us (100 MHz)
t[raw == 0] = np.nan
end = t[:, :, 7]
print("kernel span %.1f us; wave end: min %.1f median %.1f max %.1f" % (np.nanmax(end), np.nanmin(end), np.nanmedian(end), np.nanmax(end)))
print("W loaded after %.1f us (median)" % np.nanmedian(t[:, :, 1] - t[:, :, 0]))
print("per-wave mean end of first part / of all parts (by wave index):")
print("  " + " ".join("%6.1f" % x for x in np.nanmean(t[:, :, 2], axis=0)))
print("  " + " ".join("%6.1f" % x for x in np.nanmean(end, axis=0)))
it = t[:, :, 3:7]
print("iteration 2 of the first part, mean us per wave index: issue loads | products+stores | counted wait")
for w in range(12):
    d = np.diff(it[:, w, :], axis=1)
    print("  wave %2d: %6.2f %6.2f %6.2f" % (w, *np.nanmean(d, axis=0)))

# check_store_count.awk -- static check of the block-pattern kernel's ISA (run by the Makefile).
#
# A loader wavefront of block_pattern.hip (run_unit<..., LOADER = true>) requests the next block's element records with
# loads issued from inline asm and retires them with a hand-counted `s_waitcnt vmcnt(STORES)`: gfx950 retires loads and
# stores through one in-order counter, so "at most STORES operations outstanding" means "everything older than this
# iteration's STORES stores has landed".  That is only true if the compiler emits exactly the stores the source counts:
# one instruction per call, none merged, split, dropped or moved across the wait.  The kernel brackets the region with
# two comments (MHA_LOADER_ITER / MHA_LOADER_WAIT stores=N); this script counts the buffer_store instructions between
# them in every instantiation and fails the build when the number is neither N nor N + 4 (the straddle form adds four
# 8-byte stores behind a wave-uniform branch; more stores than counted only make the wait stricter, fewer would break it).
/MHA_LOADER_ITER stores=/ {
  if (open_region) { printf("check_store_count: nested MHA_LOADER_ITER at line %d\n", NR); bad = 1 }
  split($0, a, "stores="); want = a[2] + 0; count = 0; open_region = 1; start = NR; next
}
/MHA_LOADER_WAIT stores=/ {
  split($0, a, "stores="); w = a[2] + 0
  if (!open_region) { printf("check_store_count: MHA_LOADER_WAIT without MHA_LOADER_ITER at line %d\n", NR); bad = 1 }
  else if (w != want) { printf("check_store_count: markers disagree (%d / %d) at line %d\n", want, w, NR); bad = 1 }
  else if (count != want && count != want + 4) {
    printf("check_store_count: %d buffer_store instructions between lines %d and %d, the wait counts %d\n", count, start, NR, want); bad = 1
  }
  regions++; open_region = 0; next
}
open_region && /^[ \t]*buffer_store_dword/ { count++ }
open_region && /^[ \t]*(global_store|flat_store|scratch_store|buffer_load|global_load|flat_load|scratch_load)/ {
  printf("check_store_count: %s inside a counted region (line %d): it would be counted by vmcnt too\n", $1, NR); bad = 1
}
END {
  if (open_region) { print "check_store_count: unterminated region"; bad = 1 }
  if (regions == 0) { print "check_store_count: no loader regions found (markers lost?)"; bad = 1 }
  if (bad) exit 1
  printf("check_store_count: %d loader regions, store counts as declared\n", regions)
}

"""The Makefile's static check of the block-pattern kernel's ISA (mrhyde_amd/csrc/check_store_count.awk): the loader
wavefronts retire inline-asm loads with a hand-counted s_waitcnt, which is only correct if the compiler emits exactly
the stores the source counts between the two marker comments.  Here: the script accepts what the build produces and
refuses a region with a store missing, merged away, or with a foreign memory operation inside."""
import os
import shutil
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
AWK = os.path.join(HERE, "..", "mrhyde_amd", "csrc", "check_store_count.awk")


def run(text, tmp_path):
    f = tmp_path / "k.s"
    f.write_text(text)
    p = subprocess.run(["awk", "-f", AWK, str(f)], capture_output=True, text=True)
    return p.returncode, p.stdout + p.stderr


def region(nstores, declared, extra=""):
    body = "".join("\tbuffer_store_dwordx4 v[0:3], v4, s[8:11], 0 offen\n" for _ in range(nstores))
    return ("\t; MHA_LOADER_ITER stores=%d\n\tv_mfma_f64_16x16x4_f64 a[0:7], v[0:1], v[2:3], a[0:7]\n%s%s"
            "\t; MHA_LOADER_WAIT stores=%d\n\ts_waitcnt vmcnt(%d)\n" % (declared, body, extra, declared, declared))


@pytest.mark.skipif(shutil.which("awk") is None, reason="awk not installed")
def test_store_count_script(tmp_path):
    rc, out = run(region(8, 8) + region(12, 8), tmp_path)  # exact, and exact + the four straddle stores
    assert rc == 0 and "2 loader regions" in out, out
    rc, out = run(region(7, 8), tmp_path)                  # a store dropped or merged: the wait would be too weak
    assert rc != 0 and "7 buffer_store" in out, out
    rc, out = run(region(9, 8), tmp_path)                  # a split store
    assert rc != 0, out
    rc, out = run(region(8, 8, "\tglobal_load_dwordx2 v[6:7], v[8:9], off\n"), tmp_path)  # counted by vmcnt too
    assert rc != 0 and "inside a counted region" in out, out
    rc, out = run("\ts_endpgm\n", tmp_path)                # markers lost
    assert rc != 0 and "no loader regions" in out, out


def test_built_kernel_passed_the_check():
    """The build tree keeps the ISA the check ran on (Makefile: k_block_pattern.s); re-run it when present."""
    s = os.path.join(HERE, "..", "build", "obj", "k_block_pattern.s")
    if not os.path.exists(s) or shutil.which("awk") is None:
        pytest.skip("no build tree here")
    p = subprocess.run(["awk", "-f", AWK, s], capture_output=True, text=True)
    assert p.returncode == 0, p.stdout + p.stderr

"""Turns the rocprofv3 outputs of profiles/r3_traffic.sh (gpurun_out/r3_traffic/) into profiles/r3_traffic.json and copies
the two kernel-stats summaries into profiles/.

Per case: FETCH_SIZE and WRITE_SIZE (KB, from the L2's memory-side request counters TCC_EA0_RDREQ / _WRREQ; separate passes)
summed over the kernels of ONE assembly = (sum over all dispatches of the assembly's kernels) / (dispatches of its dominant
kernel).  Corrections as /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes: WRITE_SIZE is exact for streaming
stores; FETCH_SIZE tallies 128-byte requests of wide coalesced streaming reads at 64 bytes, i.e. reads up to 2x low, and is
uncalibrated for other widths -- so the read side is reported as a bracket [FETCH_SIZE, 2 FETCH_SIZE] and
hbm_bytes_per_assembly uses the UPPER end (what bench.py prints as roofline.traffic)."""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "r3_traffic")
ASSEMBLY_KERNELS = ("replicate_runs_kernel", "thermal_affine_residual_wg_kernel", "thermal_affine_residual_kernel", "block_pattern_jacobian_kernel", "thermal_general_row_owner_kernel",
                    "porous_uniform_residual_kernel", "porous_element_direct_res_kernel", "porous_element_direct_kernel", "porous_direct_finish_kernel",
                    "row_owner_jacobian", "thermal_affine_element", "thermal_general_element_kernel", "row_gather_kernel",
                    "point_engine_kernel", "porous_element_kernel", "swhdg_fused_kernel", "fillBufferAligned")
CASES = {  # case -> (dominant kernel, algorithmic bytes per element, elements, label, key and path of bench.py)
    "config2": ("block_pattern_jacobian_kernel", 6564, 64 ** 3, "config 2, affine 64^3 thermal Q2, geometry-database mode: thermal_affine_residual_wg + block_pattern_jacobian on one block per pattern + replicate_runs", "config2_affine", "row_owner"),
    "config2_perturbed": ("thermal_general_row_owner_kernel", 6564, 64 ** 3, "config 2 mesh perturbed: thermal_general_row_owner (one launch)", "config2_perturbed", "row_owner"),
    "config3": ("porous_direct_finish_kernel", 724, 128 ** 3, "config 3, 128^3 porousMixed, database mode: porous_uniform_residual (all elements) + porous_element_direct (listed elements) + porous_direct_finish + replicate_runs", "config3_affine", "row_gather"),
    "config4": ("row_gather_kernel", 65340, 64 ** 3, "config 4, 64^3 navierstokes Q2/Q1: point_engine + row_gather", "config4_affine", "row_gather"),
    "config5": ("swhdg_fused_kernel", 11056, 256 ** 2, "config 5, 256^2 HDG element step: swhdg_fused (side + volume + condensation) + row_gather (flux -> trace scatter)", "config5_affine", "hdg_fused_element_step"),
}


def counter_sum(case, counter):
    files = glob.glob(os.path.join(SRC, "pmc_%s_%s" % (case, counter), "*", "*counter_collection.csv"))
    assert len(files) == 1, (case, counter, files)
    tot, per_kernel, calls = 0.0, {}, {}
    for r in csv.DictReader(open(files[0])):
        name = r["Kernel_Name"]
        key = next((k for k in ASSEMBLY_KERNELS if k in name), None)
        if key is None or r["Counter_Name"] != counter:
            continue
        v = float(r["Counter_Value"])
        per_kernel[key] = per_kernel.get(key, 0.0) + v
        calls[key] = calls.get(key, 0) + 1
        tot += v
    return tot, per_kernel, calls


def main():
    out = {"_doc": __doc__.strip().split("\n\n")[1].replace("\n", " ")}
    for case, (dom, bpe, nelem, label, key, path) in CASES.items():
        try:
            f_tot, f_k, f_c = counter_sum(case, "FETCH_SIZE")
            w_tot, w_k, w_c = counter_sum(case, "WRITE_SIZE")
        except AssertionError as e:
            print("skipping", case, e, file=sys.stderr)
            continue
        n = w_c[dom]
        assert f_c[dom] == n, (case, f_c, w_c)
        fetch, write = f_tot * 1024.0 / n, w_tot * 1024.0 / n
        rec = {"path": path, "kernels": label, "assemblies_profiled": n, "fetch_bytes_raw": fetch, "write_bytes": write,
               "hbm_bytes_low": fetch + write, "hbm_bytes_per_assembly": 2.0 * fetch + write,
               "per_kernel_kb_per_assembly": {k: {"FETCH_SIZE": f_k.get(k, 0.0) / n, "WRITE_SIZE": w_k.get(k, 0.0) / n}
                                              for k in sorted(set(f_k) | set(w_k))}}
        if bpe:
            rec["algorithmic_bytes"] = bpe * nelem
            rec["real_over_algorithmic"] = [rec["hbm_bytes_low"] / (bpe * nelem), rec["hbm_bytes_per_assembly"] / (bpe * nelem)]
        out[key] = rec
    json.dump(out, open(os.path.join(ROOT, "profiles", "r3_traffic.json"), "w"), indent=1, sort_keys=True)
    for name in ("config2", "config2_perturbed", "config3", "config4", "config5"):
        fs = glob.glob(os.path.join(SRC, "stats_" + name, "*", "*kernel_stats.csv"))
        if fs:
            shutil.copy(fs[0], os.path.join(ROOT, "profiles", "r3_bench_%s_kernel_stats.csv" % name))
    print(json.dumps({k: {kk: vv for kk, vv in v.items() if kk != "per_kernel_kb_per_assembly"} for k, v in out.items() if k != "_doc"}, indent=1))


if __name__ == "__main__":
    main()
